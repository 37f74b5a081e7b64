"""The training cycle of the reference's loop as hipGraphs.

Reference code/helpers/model.py:340-374: per clip `temporally_enhance_features` -> losses -> `backward`, and the optimiser
steps after every 2nd clip.  Run eagerly that is ~100 libsfvos launches per clip issued from Python and from autograd's
engine thread.  At the headline configuration (4,32) the GPU has 8.8 ms of work per clip and the host keeps up (a graph
gains 0.5 %), but the reference's DEFAULT configuration (1,1) (constants.py:7-8) has 1.6 ms of GPU work behind 2.4-2.6 ms
of host work per clip: launch-bound.  `GraphedStep` captures one hipGraph per position of the accumulation cycle (clip 1:
forward, loss, backward; clip 2: the same + all-reduce hook, SGD, zero_grad) and replays them: 2.3-2.6 -> 1.64 ms per clip
at (1,1) on MI355X, same numbers bit for bit (tests/test_gpu_graph.py).

What a graph fixes in place: every address (the clip buffer, parameters, BN buffers, optimiser state, the loss scalar) and
every host-side decision taken while capturing -- so
  * the caller copies each new clip into `clip.data` (the static input buffer) before calling the step;
  * the first graph of a cycle re-packs the weight images unconditionally, the others never do: parameters may change
    between cycles (load_state_dict, an eager step) but not in the middle of one;
  * capturing runs `warmup` whole cycles eagerly first (allocator, LDS attributes, packed images); model, BN buffers and
    optimiser state are put back afterwards, so constructing a GraphedStep does not train.
"""
import torch

from . import _lib
from .parallel import GradBucket


class GraphedStep(object):
    def __init__(self, model, optimizer, loss_fn, clip, accumulate=2, bucket=None, warmup=2, slow_offset=None):
        """model: SlowFastLayers in train mode (bf16 or fp32); optimizer: FusedSGD over its parameters (attached or not);
        loss_fn(merged) -> scalar tensor, made of capturable GPU work only (MSEProxyLoss, MaskBranch losses);
        clip: PackedClip whose `data` is the static input buffer; accumulate: clips per optimiser step (2 in the reference)."""
        if not clip.data.is_cuda:
            raise RuntimeError('GraphedStep runs on the GPU through libsfvos.so (no CPU fallback)')
        if not model.training:
            raise RuntimeError('GraphedStep captures the TRAINING cycle: call model.train() first')
        if accumulate < 1:
            raise ValueError('accumulate must be at least 1')
        self.model, self.opt, self.loss_fn, self.clip = model, optimizer, loss_fn, clip
        self.accumulate, self.slow_offset = int(accumulate), slow_offset
        self.bucket = bucket if bucket is not None else GradBucket(optimizer.flat_grad)
        if self.bucket.active:
            raise RuntimeError('GraphedStep does not capture the RCCL exchange: use the eager step with more than one rank')
        self.position = 0            # clip of the cycle the next call runs
        self._epoch = None           # weight epoch the packed images inside the graphs stand for
        self._graphs, self._losses = [], []
        self._capture(int(warmup))

    # one clip of the cycle, eagerly (this is what gets captured)
    def _clip_step(self, k):
        last = k == self.accumulate - 1
        loss = self.loss_fn(self.model.enhance_packed(self.clip, self.slow_offset))
        if last:
            self.bucket.arm()
        loss.backward()
        if last:
            self.bucket.finish()
            self.opt.step()
            self.opt.zero_grad()
        return loss.detach()

    def _capture(self, warmup):
        model, opt = self.model, self.opt
        dev = self.clip.data.device
        saved = {k: v.detach().clone() for k, v in model.state_dict().items()}
        saved_buf, saved_grad, saved_steps = opt.flat_buf.clone(), opt.flat_grad.clone(), opt._steps
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                for k in range(self.accumulate):
                    self._clip_step(k)
            # (opt._steps >= 1 from here on: the "first step" flag of sfvos_sgd_step is a captured constant, and with the
            # momentum buffer restored below -- zeros if no step had been taken -- buf = momentum * buf + g IS buf = g)
            for k in range(self.accumulate):
                if k == 0:
                    # the first graph of a cycle re-packs every weight image, whatever happened before it: mark the cached
                    # ones stale (the entries stay, so that the first miss refreshes them all in ONE batched launch)
                    for key, (_, img) in list(model._packs.items()):
                        model._packs[key] = (None, img)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=side):
                    loss = self._clip_step(k)
                self._graphs.append(g)
                self._losses.append(loss)
        torch.cuda.current_stream(dev).wait_stream(side)
        # capturing executed nothing, the warm-up cycles did: undo them (in place: the graphs hold these addresses)
        with torch.no_grad():
            for k, v in model.state_dict().items():
                v.copy_(saved[k])
            opt.flat_buf.copy_(saved_buf)
            opt.flat_grad.copy_(saved_grad)
        opt._steps = max(saved_steps, 1)
        _lib.bump_weight_epoch()     # eager code must not trust images packed from the warm-up weights
        self._epoch = _lib.weight_epoch()
        self.position = 0

    def __call__(self):
        """Runs the next clip of the cycle on the data now in `clip.data`; returns the (static) loss scalar of that clip --
        read it before the same position is replayed again."""
        k = self.position
        if k > 0 and _lib.weight_epoch() != self._epoch:
            raise RuntimeError('GraphedStep: the parameters changed outside the graphs in the middle of an accumulation '
                               'cycle (clip %d of %d); call reset() and restart the cycle' % (k + 1, self.accumulate))
        self._graphs[k].replay()
        self.position = (k + 1) % self.accumulate
        if self.position == 0:
            # the replayed SGD step moved the parameters behind Python's back: what the eager path cached is stale
            self.opt._steps += 1
            _lib.bump_weight_epoch()
        self._epoch = _lib.weight_epoch()
        return self._losses[k]

    def reset(self):
        """Drop a partly accumulated cycle: the next call is clip 1 again (gradients are zeroed)."""
        self.opt.zero_grad()
        self.position = 0
