"""GPU, ONE rank, backend nccl (= RCCL on ROCm): everything of the data-parallel gradient exchange short of xGMI.

The build box has one GPU, so the multi-GPU scaling curve is the driver's; what CAN be proven here is that the RCCL
branch of sfvos_amd.parallel works on the real backend: librccl loads and initialises a communicator, the bucket
all-reduces are issued `async_op` on the side HIP stream behind the streams that produced the gradients, finish() waits
for them before the optimiser, and the result of bench.py's own step function with the exchange switched on is
BIT-IDENTICAL to the same steps without it (an all-reduce over one rank is the identity, 1/world = 1).

The process group is created in a fresh child process (spawn) before that process makes any GPU call: the pytest
process itself has initialised the GPU long before."""
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


def _worker(port, q):
    try:
        os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0',
                          HSA_ENABLE_IPC_MODE_LEGACY='0')
        os.environ.pop('SFVOS_DIST_BACKEND', None)
        import torch.distributed as dist
        import bench
        from golden_util import SMALL_LEVELS
        from oracle.closed_form import closed_form_features, closed_form_state_dict, closed_form_tensor
        from sfvos_amd import FusedSGD, GradBucket, MSEProxyLoss, PackedClip, SlowFastLayers, init_distributed
        rank, world, _ = init_distributed(backend='nccl', force=True)     # before any GPU call of this process
        assert (rank, world) == (0, 1) and dist.is_initialized() and dist.get_backend() == 'nccl'
        dev = torch.device('cuda:0')
        torch.cuda.set_device(dev)

        def levels(idx):
            fast = closed_form_features(7, SMALL_LEVELS, clip=20 + idx)
            return [fast[k].to(dev).permute(0, 2, 3, 1).unsqueeze(0).contiguous() for k in SMALL_LEVELS]

        def run(force, coalesce=True):
            m = SlowFastLayers(256, dev, 3, 7, precision='fp32')
            m.load_state_dict(closed_form_state_dict(m))
            m = m.to(dev).train()
            opt = FusedSGD(m.parameters(), lr=1e-2, momentum=0.9, weight_decay=1e-4)
            bucket = GradBucket(opt.flat_grad, force=force, coalesce=coalesce)
            assert bucket.active == force
            opt.attach(m, bucket)
            opt.zero_grad()
            loss_fn = MSEProxyLoss({k: closed_form_tensor((1, 256, h, w), 'target/%s' % k, 4.0).to(dev)
                                    for k, (h, w) in SMALL_LEVELS.items()})
            cur = {}
            step = bench.make_step(m, opt, bucket, loss_fn, lambda: m.enhance_packed(cur['clip']))
            losses = []
            for i in range(4):       # optimiser (and exchange) after i = 1 and i = 3
                cur['clip'] = PackedClip.from_levels(levels(i), keys=list(SMALL_LEVELS.keys()))
                losses.append(float(step(i).detach()))
            torch.cuda.synchronize()
            return opt.flat_param.detach().cpu().clone(), losses, bucket.collectives

        p0, l0, c0 = run(False)
        p1, l1, c1 = run(True)
        p2, l2, c2 = run(True, coalesce=False)
        # a plain whole-buffer all-reduce on the default stream as well
        t = torch.arange(1024, dtype=torch.float32, device=dev)
        b = GradBucket(t, force=True)
        b.all_reduce()
        torch.cuda.synchronize()
        ok_plain = bool(torch.equal(t.cpu(), torch.arange(1024, dtype=torch.float32)))
        q.put(('ok', bool(torch.equal(p0, p1)), bool(torch.equal(p0, p2)), l0 == l1 == l2, c0, c1, c2, ok_plain,
               float(p1.abs().max())))
        dist.barrier()
        dist.destroy_process_group()
    except BaseException:
        import traceback
        q.put(('error', traceback.format_exc()))


def test_gradient_buckets_go_through_rccl_on_one_rank_bit_identically():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    p = ctx.Process(target=_worker, args=(_free_port(), q))
    p.start()
    got = q.get(timeout=600)
    p.join(300)
    assert got[0] == 'ok', got[1]
    _, same, same_per_layer, same_losses, c0, c1, c2, ok_plain, moved = got
    print('RCCL world-size-1: collectives issued off/coalesced/per-layer = %d/%d/%d' % (c0, c1, c2))
    assert c0 == 0, 'an inactive bucket must not issue collectives'
    assert c1 == 2 * 4, 'two optimiser steps x four coalesced buckets'
    assert c2 >= 2 * 8, 'two optimiser steps x one collective per layer'
    assert same and same_per_layer and same_losses, 'the exchange over one rank must be the identity, bit for bit'
    assert ok_plain and moved > 0
    assert p.exitcode == 0
