"""Data-parallel plumbing for the hot path: one process per GPU, torch.distributed
(backend "nccl" == RCCL over xGMI on ROCm; "gloo" for the CPU tests).

The reference's SlowFast path is single-process; clips are independent units,
so they shard across ranks with ONE exchange step per optimiser step: the all-reduce(sum) of the
flat gradient bucket (3.9 M fp32 at (sp,fp)=(4,32)), averaged by world size.  The collective
runs on a side stream, layer by layer as backward finishes each layer's gradients (layer 3 and
the second lateral first), so all but the first layer's share overlaps with the rest of backward;
the optimiser waits for it.
BatchNorm statistics stay per replica (the reference has no SyncBN)."""
import os

import torch
import torch.distributed as dist


def init_distributed(backend=None):
    """Initialise the default process group from the torchrun environment (RANK, WORLD_SIZE,
    LOCAL_RANK, MASTER_ADDR, MASTER_PORT).  Returns (rank, world_size, local_rank)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 and not dist.is_initialized():
        if backend is None:  # SFVOS_DIST_BACKEND=gloo: rehearse the multi-rank control flow without RCCL
            backend = os.environ.get('SFVOS_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            torch.cuda.set_device(local % max(1, torch.cuda.device_count()))
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


class GradBucket(object):
    """All-reduce + average of one flat gradient tensor.

    `flat_grad` is FusedSGD.flat_grad (or any 1-D fp32 tensor the parameters' .grad alias).

    Whole-buffer use:  all_reduce()  (= start() + finish()) after backward.

    Overlapped with backward: arm() before the backward whose gradients are to be exchanged; the
    producer of the gradients then reports every finished contiguous range with segment_ready(lo, hi) -- FusedSGD.attach
    (module, bucket) makes SlowFastLayers' backward do that layer by layer, on the stream the weight-gradient kernels
    ran on -- and each range is all-reduced at once on a side stream (RCCL over xGMI) while backward continues with the
    earlier layers; finish() waits for the collectives, reduces whatever was not reported, and applies 1/world."""

    def __init__(self, flat_grad, group=None):
        self.flat = flat_grad
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self._works = []
        self._stream = torch.cuda.Stream(device=flat_grad.device) if flat_grad.is_cuda else None
        self._armed = False
        self._sent = []      # [lo, hi) ranges already handed to the collective since arm()

    # -- whole buffer -------------------------------------------------------------------------------------
    def start(self):
        if self.world == 1:
            return
        self._armed = False
        self._launch(0, self.flat.numel())

    def _launch(self, lo, hi, producer_stream=None):
        part = self.flat[lo:hi]
        if self._stream is not None:
            self._stream.wait_stream(producer_stream if producer_stream is not None else torch.cuda.current_stream())
            with torch.cuda.stream(self._stream):
                self._works.append(dist.all_reduce(part, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:
            self._works.append(dist.all_reduce(part, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        self._sent.append((lo, hi))

    # -- overlapped with backward -------------------------------------------------------------------------
    def arm(self):
        """The next backward produces the gradients to exchange: accept segment_ready() calls."""
        self._armed = self.world > 1
        self._sent = []

    @property
    def armed(self):
        return self._armed

    def segment_ready(self, lo, hi, producer_stream=None):
        """flat[lo:hi] is final (every kernel writing it has been enqueued on producer_stream / the current stream)."""
        if not self._armed or hi <= lo:
            return
        for a, b in self._sent:
            if lo < b and a < hi:
                raise RuntimeError('GradBucket: range [%d, %d) reported twice' % (lo, hi))
        self._launch(lo, hi, producer_stream)

    def finish(self):
        if self.world == 1:
            return
        if self._armed or self._sent:
            # whatever backward did not report (parameters outside the module, or no gradient sink attached)
            covered = sorted(self._sent)
            pos, gaps = 0, []
            for a, b in covered:
                if a > pos:
                    gaps.append((pos, a))
                pos = max(pos, b)
            if pos < self.flat.numel():
                gaps.append((pos, self.flat.numel()))
            for a, b in gaps:
                self._launch(a, b)
        self._armed = False
        self._sent = []
        for w in self._works:
            w.wait()
        self._works = []
        if self._stream is not None:
            torch.cuda.current_stream().wait_stream(self._stream)
            import ctypes
            from . import _lib
            _lib.call('sfvos_scale', ctypes.c_void_p(self.flat.data_ptr()), self.flat.numel(), 1.0 / self.world,
                      ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        else:
            self.flat.mul_(1.0 / self.world)  # gloo / CPU test path: host tensor, plumbing only

    def all_reduce(self):
        self.start()
        self.finish()


def shard_clips(num_clips, rank, world):
    """Clip i -> rank i mod world: the indices this rank processes."""
    return list(range(rank, num_clips, world))
