"""Data-parallel plumbing for the hot path: one process per GPU, torch.distributed
(backend "nccl" == RCCL over xGMI on ROCm; "gloo" for the CPU tests).

The reference's SlowFast path is single-process; clips are independent units,
so they shard across ranks with ONE exchange step per optimiser step: the all-reduce(sum) of the
flat gradient buffer (3.9 M fp32 at (sp,fp)=(4,32)), averaged by world size.  The collectives
run on a side stream while backward is still producing the earlier layers' gradients: the flat
buffer is cut into (at most) four contiguous buckets -- layer 3, layer 2, both laterals, layer 1, in
the order backward completes them (f3 s3 l2 f2 s2 l1 f1 s1; the laterals sit together at the end of
the reference's registration order, so their range is complete once conv_f2s1's backward has run,
behind layer 2) -- and each is all-reduced the moment backward has enqueued the last kernel writing
into it, so only the first layer's bucket cannot overlap; the optimiser waits for all of them.
xGMI is point to point: collectives of 0.3-7 MB keep the per-call RCCL launch latency (x4, not x8)
small against the transfer time while still hiding three of them behind backward.  The overlap is
verified on RCCL with ONE rank only (tests/test_gpu_rccl.py: librccl loads, the side-stream ordering
and finish() hold on the real backend); its effect on a multi-GPU step is the driver's measurement.
BatchNorm statistics stay per replica (the reference has no SyncBN)."""
import os

import torch
import torch.distributed as dist


def init_distributed(backend=None, force=False):
    """Initialise the default process group from the torchrun environment (RANK, WORLD_SIZE,
    LOCAL_RANK, MASTER_ADDR, MASTER_PORT).  Returns (rank, world_size, local_rank).
    force: create the process group even for ONE rank (bench.py --force-dist, tests/test_gpu_rccl.py: the RCCL path of
    the gradient exchange on a single GPU)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if force:
        os.environ.setdefault('MASTER_PORT', '29543')
    if (world > 1 or force) and not dist.is_initialized():
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

    def __init__(self, flat_grad, group=None, coalesce=True, force=False):
        """force: run the collectives even when the group has ONE rank (an all-reduce over one rank is the identity;
        this is how the RCCL path is exercised on a single GPU)."""
        self.flat = flat_grad
        self.group = group
        self.coalesce = coalesce   # False: every reported segment is its own collective (A/B)
        self._buckets = []         # [lo, hi) ranges all-reduced as ONE collective each once fully reported
        self._ready = []           # segments reported since arm() that have not been sent yet
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.active = self.world > 1 or (bool(force) and dist.is_initialized())
        self.collectives = 0       # all-reduce calls issued so far (tests / bench report)
        self._works = []
        self._stream = torch.cuda.Stream(device=flat_grad.device) if flat_grad.is_cuda else None
        self._armed = False
        self._sent = []      # [lo, hi) ranges already handed to the collective since arm()

    # -- whole buffer -------------------------------------------------------------------------------------
    def start(self):
        if not self.active:
            return
        self._armed = False
        self._launch(0, self.flat.numel())

    def _launch(self, lo, hi, producer_stream=None):
        part = self.flat[lo:hi]
        if self._stream is not None:
            producers = producer_stream if isinstance(producer_stream, (list, tuple)) else [producer_stream]
            for ps in producers:
                self._stream.wait_stream(ps if ps is not None else torch.cuda.current_stream())
            with torch.cuda.stream(self._stream):
                self._works.append(dist.all_reduce(part, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:
            self._works.append(dist.all_reduce(part, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        self.collectives += 1
        self._sent.append((lo, hi))

    # -- overlapped with backward -------------------------------------------------------------------------
    def set_buckets(self, ranges):
        """Disjoint [lo, hi) ranges of the flat buffer (FusedSGD.attach derives them from the module's layer
        table): segments reported inside a bucket are held back until they cover it, then sent as one collective."""
        ranges = sorted((int(a), int(b)) for a, b in ranges if b > a)
        for (a0, b0), (a1, b1) in zip(ranges, ranges[1:]):
            if a1 < b0:
                raise ValueError('GradBucket: buckets overlap')
        self._buckets = ranges

    def arm(self):
        """The next backward produces the gradients to exchange: accept segment_ready() calls."""
        self._armed = self.active
        self._sent = []
        self._ready = []

    @property
    def armed(self):
        return self._armed

    def segment_ready(self, lo, hi, producer_stream=None):
        """flat[lo:hi] is final (every kernel writing it has been enqueued on producer_stream / the current stream)."""
        if not self._armed or hi <= lo:
            return
        for a, b in self._sent + [(r[0], r[1]) for r in self._ready]:
            if lo < b and a < hi:
                raise RuntimeError('GradBucket: range [%d, %d) reported twice' % (lo, hi))
        bucket = None
        if self.coalesce:
            for a, b in self._buckets:
                if a <= lo and hi <= b:
                    bucket = (a, b)
        if bucket is None:
            self._launch(lo, hi, producer_stream)
            return
        self._ready.append((lo, hi, producer_stream))
        mine = sorted(r for r in self._ready if bucket[0] <= r[0] and r[1] <= bucket[1])
        pos = bucket[0]
        for a, b, _ in mine:
            if a != pos:
                return          # a hole: some layer of this bucket has not reported yet
            pos = b
        if pos != bucket[1]:
            return
        # complete: the collective waits for every stream that produced a piece of it
        self._ready = [r for r in self._ready if r not in mine]
        streams = []
        for _, _, st in mine:
            if st is not None and st not in streams:
                streams.append(st)
        self._launch(bucket[0], bucket[1], streams or None)

    def finish(self):
        if not self.active:
            return
        # whatever has not been handed to the collective yet (nothing at all when neither start() nor arm() ran:
        # finish() alone is then a whole-buffer all-reduce -- it never scales unreduced gradients)
        self._ready = []   # held-back segments of incomplete buckets go out with the gaps below
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
