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
    q.put((rank, [r.numpy() for r in results]))   # by value: the sender may exit before the parent reads
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
    got = {r: [torch.from_numpy(a) for a in v] for r, v in got.items()}
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


def _worker_segments(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from sfvos_amd import GradBucket, init_distributed
    init_distributed('gloo')
    n = 4096
    flat = torch.randn(n, generator=torch.Generator().manual_seed(50 + rank))
    local = flat.clone()
    bucket = GradBucket(flat)
    bucket.arm()
    assert bucket.armed
    # ranges reported in the order backward finishes its layers (last layer first), one range never reported
    for lo, hi in ((3000, 4096), (1200, 3000), (0, 700)):
        bucket.segment_ready(lo, hi)
    try:
        bucket.segment_ready(600, 800)
        overlap_refused = False
    except RuntimeError:
        overlap_refused = True
    bucket.finish()                       # reduces the unreported [700, 1200) too, then scales
    assert not bucket.armed
    q.put((rank, local.numpy(), flat.clone().numpy(), overlap_refused))
    dist.barrier()
    dist.destroy_process_group()


def test_grad_bucket_segments_overlap_api_world2():
    """arm() / segment_ready() / finish(): the per-layer exchange backward drives must give the same result as
    one all-reduce of the whole buffer, whatever the order and whatever is left unreported."""
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_segments, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    want = sum(torch.from_numpy(g[1]) for g in got) / world
    for rank, _, reduced, refused in got:
        assert refused
        assert torch.allclose(torch.from_numpy(reduced), want, atol=1e-6)


def _worker_buckets(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from sfvos_amd import GradBucket, init_distributed
    init_distributed('gloo')
    n = 4096
    flat = torch.randn(n, generator=torch.Generator().manual_seed(70 + rank))
    local = flat.clone()
    bucket = GradBucket(flat)
    bucket.set_buckets([(2000, 4096), (500, 2000)])       # [0, 500) belongs to no bucket
    bucket.arm()
    launched = []
    for lo, hi in ((3000, 4096), (500, 1200), (2000, 3000), (0, 300), (1200, 2000)):
        bucket.segment_ready(lo, hi)
        launched.append(sorted(bucket._sent))
    bucket.finish()
    # finish() with nothing armed or started must still reduce (never scale unreduced gradients)
    flat2 = torch.randn(n, generator=torch.Generator().manual_seed(90 + rank))
    local2 = flat2.clone()
    GradBucket(flat2).finish()
    q.put((rank, local.numpy(), flat.clone().numpy(), launched, local2.numpy(), flat2.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_grad_bucket_coalesced_buckets_world2():
    """Segments inside a bucket are held back until they cover it and then go out as ONE collective; segments outside
    every bucket go at once; finish() sends the rest.  Result = plain average, as for one whole-buffer all-reduce."""
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_buckets, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    want = sum(torch.from_numpy(g[1]) for g in got) / world
    want2 = sum(torch.from_numpy(g[4]) for g in got) / world
    for rank, _, reduced, launched, _, reduced2 in got:
        assert launched == [[], [], [(2000, 4096)], [(0, 300), (2000, 4096)], [(0, 300), (500, 2000), (2000, 4096)]]
        assert torch.allclose(torch.from_numpy(reduced), want, atol=1e-6)
        assert torch.allclose(torch.from_numpy(reduced2), want2, atol=1e-6)
