"""CPU, world_size 2, gloo: the data-parallel exchange step of the hot path -- one all-reduce of
the flat gradient bucket every 2nd clip, averaged -- must equal the single-process gradient of
the same clips.  The conv arithmetic itself is out of scope here (no GPU): the
per-rank gradients are synthetic, the schedule and the bucket logic are the thing under test."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from sfvos_amd import GradBucket, init_distributed
    from sfvos_amd.parallel import shard_clips
    r, w, _ = init_distributed('gloo')
    assert (r, w) == (rank, world)
    n = 3890496 // 64  # a slice of the (4,32) bucket size keeps the test fast
    flat = torch.zeros(n)
    bucket = GradBucket(flat)
    clips = shard_clips(8, rank, world)
    results = []
    for step, clip in enumerate(clips):
        g = torch.Generator().manual_seed(1000 + clip)
        flat += torch.randn(n, generator=g)          # "backward" of this rank's clip accumulates
        if step % 2 == 1:                            # reference schedule: step every 2nd clip
            bucket.all_reduce()
            results.append(flat.clone())
            flat.zero_()
    q.put((rank, results))
    dist.barrier()
    dist.destroy_process_group()


def test_grad_bucket_allreduce_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    n = 3890496 // 64
    # single-process truth: mean over ranks of the sum of each rank's two clips
    for k in range(2):
        want = torch.zeros(n)
        for rank in range(world):
            for clip in (rank + world * (2 * k), rank + world * (2 * k + 1)):
                want += torch.randn(n, generator=torch.Generator().manual_seed(1000 + clip))
        want /= world
        for rank in range(world):
            assert torch.allclose(got[rank][k], want, atol=1e-6)
    assert torch.equal(got[0][0], got[1][0])
