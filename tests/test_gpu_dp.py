"""GPU, two ranks sharing cuda:0, gloo transport (the collective library is not the thing under test; RCCL needs
one GPU per rank): the gradient exchange SlowFastLayers' backward drives through FusedSGD.attach(module, bucket) --
the flat gradient all-reduced in four coalesced buckets (or layer by layer) as soon as backward has produced them --
must leave the average of the ranks' local gradients in every rank's buffer; and bench.py's own step function, run at
world size 2, must leave both ranks with identical parameters equal to the single-process result on the same clips."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    try:
        _worker_body(rank, world, port, q)
    except BaseException:   # hand the traceback to the parent instead of dying silently
        import traceback
        q.put((rank, 'error', traceback.format_exc()))


def _worker_body(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK='0', SFVOS_DIST_BACKEND='gloo')
    import torch.distributed as dist
    from golden_util import SMALL_LEVELS, clip_inputs
    from oracle.closed_form import closed_form_state_dict
    from oracle.slowfast_ref import proxy_loss
    from sfvos_amd import FusedSGD, GradBucket, SlowFastLayers, init_distributed
    init_distributed()
    dev = torch.device('cuda:0')
    torch.cuda.set_device(dev)

    def local_and_exchanged(exchange, coalesce=True):
        m = SlowFastLayers(256, dev, 3, 7, precision='fp32')
        m.load_state_dict(closed_form_state_dict(m))
        m = m.to(dev).train()
        opt = FusedSGD(m.parameters())
        bucket = GradBucket(opt.flat_grad, coalesce=coalesce)
        opt.attach(m, bucket)
        opt.zero_grad()
        slow, fast = clip_inputs(3, 7, SMALL_LEVELS, rank, dev)     # rank r trains on clip r
        loss = proxy_loss(m.temporally_enhance_features(slow, fast))
        if exchange:
            bucket.arm()
        loss.backward()
        if exchange:
            order = list(bucket._sent)
            sent = sorted(order)
            if coalesce:   # (f1,s1) (f2,s2) (f3,s3) (l1,l2): four collectives that tile the whole buffer ...
                assert len(sent) == 4 and sent[0][0] == 0 and sent[-1][1] == opt.flat_grad.numel(), sent
                assert all(a[1] == b[0] for a, b in zip(sent, sent[1:])), sent
                # ... and go out in the order backward completes them: layer 3, layer 2, laterals, layer 1
                assert order == [sent[2], sent[1], sent[3], sent[0]], (order, sent)
            else:
                assert len(sent) >= 8, 'backward did not report its layers'
            bucket.finish()
        torch.cuda.synchronize()
        return opt.flat_grad.detach().cpu().clone()

    local = local_and_exchanged(False)
    reduced = local_and_exchanged(True)
    per_layer = local_and_exchanged(True, coalesce=False)
    assert torch.equal(reduced, per_layer)
    q.put((rank, local.numpy(), reduced.numpy()))   # by value: the sender may exit before the parent reads
    dist.barrier()
    dist.destroy_process_group()


def test_backward_driven_layerwise_allreduce_two_ranks():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in range(world)]
    for g in got:
        assert not isinstance(g[1], str), g[2]
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    got = [(r, torch.from_numpy(a), torch.from_numpy(b)) for r, a, b in got]
    want = sum(g[1] for g in got) / world
    scale = float(want.abs().max())
    assert scale > 0 and float((got[0][1] - got[1][1]).abs().max()) > 1e-3 * scale   # the ranks really differ
    for rank, _, reduced in got:
        assert float((reduced - want).abs().max()) <= 1e-6 * scale


# ---- bench.py's step function at world size 2 ---------------------------------------------------------------
def _clip_levels(idx, dev):
    from golden_util import SMALL_LEVELS
    from oracle.closed_form import closed_form_features
    fast = closed_form_features(7, SMALL_LEVELS, clip=20 + idx)
    return [fast[k].to(dev).permute(0, 2, 3, 1).unsqueeze(0).contiguous() for k in SMALL_LEVELS]   # [1,T,H,W,C]


def _bench_setup(dev, world_bucket):
    import bench
    from golden_util import SMALL_LEVELS
    from oracle.closed_form import closed_form_state_dict, closed_form_tensor
    from sfvos_amd import FusedSGD, GradBucket, MSEProxyLoss, PackedClip, SlowFastLayers
    m = SlowFastLayers(256, dev, 3, 7, precision='fp32')
    m.load_state_dict(closed_form_state_dict(m))
    m = m.to(dev).train()
    opt = FusedSGD(m.parameters(), lr=1e-2, momentum=0.9, weight_decay=1e-4)
    bucket = GradBucket(opt.flat_grad)
    opt.attach(m, bucket)
    opt.zero_grad()
    loss_fn = MSEProxyLoss({k: closed_form_tensor((1, 256, h, w), 'target/%s' % k, 4.0).to(dev)
                            for k, (h, w) in SMALL_LEVELS.items()})
    cur = {}
    step = bench.make_step(m, opt, bucket, loss_fn, lambda: m.enhance_packed(cur['clip']))

    def run(i, clip_idx):
        cur['clip'] = PackedClip.from_levels(_clip_levels(clip_idx, dev), keys=list(SMALL_LEVELS.keys()))
        return step(i)
    return m, opt, run


def _worker_bench(rank, world, port, q):
    try:
        os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                          LOCAL_RANK='0', SFVOS_DIST_BACKEND='gloo')
        import torch.distributed as dist
        from sfvos_amd import init_distributed
        init_distributed()
        dev = torch.device('cuda:0')
        torch.cuda.set_device(dev)
        m, opt, run = _bench_setup(dev, True)
        for i in range(4):                      # clip 2*i + rank -> rank; optimiser after i = 1 and 3
            run(i, 2 * i + rank)
        torch.cuda.synchronize()
        q.put((rank, opt.flat_param.detach().cpu().numpy()))
        dist.barrier()
        dist.destroy_process_group()
    except BaseException:
        import traceback
        q.put((rank, 'error', traceback.format_exc()))


def test_bench_step_at_world_size_two_equals_single_process():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_bench, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in range(world)]
    for g in got:
        assert not isinstance(g[1], str), g[2]
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    params = {r: torch.from_numpy(a) for r, a in got}
    assert torch.equal(params[0], params[1]), 'the ranks diverged'
    # single process, same clips: every optimiser step sees the average over ranks of each rank's two-clip sum
    dev = torch.device('cuda:0')
    m, opt, _ = _bench_setup(dev, False)
    from golden_util import SMALL_LEVELS
    from oracle.closed_form import closed_form_tensor
    from sfvos_amd import MSEProxyLoss, PackedClip, _lib
    import ctypes
    loss_fn = MSEProxyLoss({k: closed_form_tensor((1, 256, h, w), 'target/%s' % k, 4.0).to(dev)
                            for k, (h, w) in SMALL_LEVELS.items()})
    for k in range(2):
        for idx in (4 * k, 4 * k + 1, 4 * k + 2, 4 * k + 3):
            clip = PackedClip.from_levels(_clip_levels(idx, dev), keys=list(SMALL_LEVELS.keys()))
            loss_fn(m.enhance_packed(clip)).backward()
        _lib.call('sfvos_scale', ctypes.c_void_p(opt.flat_grad.data_ptr()), opt.flat_grad.numel(), 0.5,
                  ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        opt.step()
        opt.zero_grad()
    torch.cuda.synchronize()
    want = opt.flat_param.detach().cpu()
    scale = float(want.abs().max())
    assert float((params[0] - want).abs().max()) <= 1e-5 * scale
