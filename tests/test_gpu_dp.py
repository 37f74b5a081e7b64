"""GPU, two ranks sharing cuda:0, gloo transport (the collective library is not the thing under test; RCCL needs
one GPU per rank): the per-layer gradient exchange SlowFastLayers' backward drives through FusedSGD.attach(module,
bucket) -- every layer's slice of the flat gradient all-reduced as soon as backward has produced it -- must leave
the average of the ranks' local gradients in every rank's buffer."""
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

    def local_and_exchanged(exchange):
        m = SlowFastLayers(256, dev, 3, 7, precision='fp32')
        m.load_state_dict(closed_form_state_dict(m))
        m = m.to(dev).train()
        opt = FusedSGD(m.parameters())
        bucket = GradBucket(opt.flat_grad)
        opt.attach(m, bucket)
        opt.zero_grad()
        slow, fast = clip_inputs(3, 7, SMALL_LEVELS, rank, dev)     # rank r trains on clip r
        loss = proxy_loss(m.temporally_enhance_features(slow, fast))
        if exchange:
            bucket.arm()
        loss.backward()
        if exchange:
            assert len(bucket._sent) >= 8, 'backward did not report its layers'
            bucket.finish()
        torch.cuda.synchronize()
        return opt.flat_grad.detach().cpu().clone()

    local = local_and_exchanged(False)
    reduced = local_and_exchanged(True)
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
