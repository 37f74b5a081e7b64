"""Data-parallel plumbing for the hot path: one process per GPU, torch.distributed
(backend "nccl" == RCCL over xGMI on ROCm; "gloo" for the CPU tests).

The reference's SlowFast path is single-process; clips are independent units,
so they shard across ranks with ONE exchange step per optimiser step: the all-reduce(sum) of the
flat gradient bucket (3.9 M fp32 at (sp,fp)=(4,32)), averaged by world size.  The collective
is issued on a side stream as soon as backward has produced the last gradient and the optimiser
waits on its event, so it overlaps with whatever the caller runs next on the compute stream.
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
    start() launches the collective (side stream on GPU), finish() makes the current stream
    wait for it and applies the 1/world scaling."""

    def __init__(self, flat_grad, group=None):
        self.flat = flat_grad
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self._work = None
        self._stream = torch.cuda.Stream(device=flat_grad.device) if flat_grad.is_cuda else None
        self._event = None

    def start(self):
        if self.world == 1:
            return
        if self._stream is not None:
            self._stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._stream):
                self._work = dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        else:
            self._work = dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def finish(self):
        if self.world == 1:
            return
        if self._work is not None:
            self._work.wait()
            self._work = None
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
