"""CPU (no GPU): host-side logic of the build -- planner tables, module surface/state-dict,
C-ABI symbol export -- and the loud failure when asked to compute without a GPU."""
import ctypes
import os
import re

import pytest
import torch

from golden_util import load_tables
from oracle.slowfast_ref import OracleSlowFastLayers

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_planner_kernel_sizes_match_reference_tables():
    from sfvos_amd import SlowFastPlan, calc_kernel_sizes
    t = load_tables()
    for p, ks in t['calc_kernel_sizes'].items():
        assert list(calc_kernel_sizes(int(p))) == ks
    for key, ks in t['kernel_sizes'].items():
        sp, fp = map(int, key.split('-'))
        plan = SlowFastPlan(256, sp, fp)
        assert list(plan.k_slow) == ks['slow'] and list(plan.k_fast) == ks['fast'] and list(plan.k_lat) == ks['lateral']
        assert plan.param_count() == t['param_counts'][key]
        # temporal extents collapse to exactly one frame
        assert plan.layer('s3').t_out == 1 and plan.layer('f3').t_out == 1
        assert plan.layer('l1').t_out == plan.layer('s1').t_out and plan.layer('l2').t_out == plan.layer('s2').t_out


def test_planner_rejects_impossible_configs():
    from sfvos_amd import SlowFastPlan
    with pytest.raises(ValueError):
        SlowFastPlan(256, 7, 1)      # lateral kernel would be < 1
    with pytest.raises(ValueError):
        SlowFastPlan(100, 1, 1)      # channels not a multiple of the MFMA tile
    with pytest.raises(ValueError):
        SlowFastPlan(256, 0, 3)


def test_flop_model_matches_survey():
    from sfvos_amd import SlowFastPlan, davis_pyramid
    P = sum(h * w for _, (h, w) in davis_pyramid())
    assert P == 85932
    plan = SlowFastPlan(256, 4, 32)
    assert abs(plan.forward_flops(P) / 1e9 - 4261) < 2
    assert abs(plan.train_flops(P) / 1e9 - 9260) < 4             # row a9
    assert abs(plan.train_flops(P, True) / 1e9 - 12783) < 6
    assert abs(plan.layer_flops(P)['f1'] / 1e9 - 3066.4) < 1
    assert abs(SlowFastPlan(256, 1, 1).forward_flops(P) / 1e9 - 257) < 1


@pytest.mark.parametrize('sp,fp', [(1, 1), (3, 7), (4, 32)])
def test_module_surface_matches_reference(sp, fp):
    from sfvos_amd import SlowFastLayers
    torch.manual_seed(63)
    m = SlowFastLayers(256, torch.device('cpu'), sp, fp)
    torch.manual_seed(63)
    o = OracleSlowFastLayers(256, torch.device('cpu'), sp, fp)
    sd, so = m.state_dict(), o.state_dict()
    assert list(sd.keys()) == list(so.keys())
    for k in sd:
        assert sd[k].shape == so[k].shape and sd[k].dtype == so[k].dtype
        assert torch.equal(sd[k], so[k]), 'default init differs for %s (RNG order)' % k
    assert [n for n, _ in m.named_parameters()] == [n for n, _ in o.named_parameters()]
    if (sp, fp) == (3, 7):
        t = load_tables()['state_dict']
        assert list(sd.keys()) == list(t.keys())
    m.load_state_dict(so, strict=True)
    assert m.training and not m.eval().training


def test_no_cpu_fallback():
    from sfvos_amd import FusedSGD, SlowFastLayers
    m = SlowFastLayers(256, torch.device('cpu'), 1, 1)
    x = torch.zeros(1, 256, 1, 4, 4)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        m(x, x)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        m.temporally_enhance_features([{'0': x[0].transpose(0, 1)}], [{'0': x[0].transpose(0, 1)}])
    opt = FusedSGD(m.parameters())
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        opt.step()


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, 'applying-slowfast-networks-to-video-object-segmentation_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle', src, re.M), f
                assert 'conv3d(' not in src.replace('sfvos_conv3d(', '').replace('_conv3d(', '') or f.endswith('.hip'), f


def test_cabi_library_exports_every_declared_symbol():
    from sfvos_amd import _lib
    header = open(os.path.join(ROOT, 'include', 'sfvos.h')).read()
    declared = set(re.findall(r'\b(sfvos_[a-z0-9_]+)\s*\(', header))
    declared -= {'sfvos_conv_desc'}
    assert declared == set(_lib.SIGNATURES.keys()), declared ^ set(_lib.SIGNATURES.keys())
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip('libsfvos.so not built (run `python __graft_entry__.py`)')
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    lib.sfvos_version.restype = ctypes.c_int
    assert lib.sfvos_version() >= _lib.ABI_REVISION
    # struct mirrors: the binder's sizes must be the ones the library was compiled with (sfvos_abi_sizes); _lib.load()
    # enforces the same at import time
    sizes = (ctypes.c_int * 7)()
    assert lib.sfvos_abi_sizes(sizes, 7) == 7
    assert list(sizes) == [ctypes.sizeof(_lib.ConvDesc), ctypes.sizeof(_lib.Pyramid), ctypes.sizeof(_lib.Levels),
                           ctypes.sizeof(_lib.MseTable), ctypes.sizeof(_lib.BnRunning), ctypes.sizeof(_lib.PackItem),
                           ctypes.sizeof(_lib.PlanarLevel)]
    assert ctypes.sizeof(_lib.BnRunning) == 48 and ctypes.sizeof(_lib.PackItem) == 40
    assert ctypes.sizeof(_lib.PlanarLevel) == 48
    assert ctypes.sizeof(_lib.Pyramid) == 68 and ctypes.sizeof(_lib.ConvDesc) == 152 and ctypes.sizeof(_lib.Levels) == 72
    assert _lib.ConvDesc().struct_size == 152
    _lib.load()


def test_integration_doc_shows_the_real_conv_desc():
    """INTEGRATION.md's ctypes mirror of sfvos_conv_desc must list exactly the fields of the binding in use."""
    from sfvos_amd import _lib
    doc = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    block = doc[doc.index('class ConvDesc(C.Structure):'):doc.index('lib.sfvos_conv3d.restype')]
    names = re.findall(r"\('([a-z_]+)',", block)
    assert names == [f[0] for f in _lib.ConvDesc._fields_], names


def test_bench_timed_region_never_touches_oracle():
    """bench.py may use oracle/ only inside cpu_baseline() (the reported CPU leg)."""
    src = open(os.path.join(ROOT, 'bench.py')).read()
    hits = [m.start() for m in re.finditer(r'^\s*(from|import)\s+oracle', src, re.M)]
    lo = src.index('def cpu_baseline(')
    hi = src.index('def pmc_traffic(')
    assert hits and all(lo < h < hi for h in hits), 'oracle imported outside cpu_baseline()'


def test_shard_clips():
    from sfvos_amd.parallel import shard_clips
    assert shard_clips(10, 0, 4) == [0, 4, 8] and shard_clips(10, 3, 4) == [3, 7]
    got = sorted(i for r in range(8) for i in shard_clips(37, r, 8))
    assert got == list(range(37))


def test_stale_conv_desc_is_refused_before_any_launch():
    """A binder that passes a shorter / older sfvos_conv_desc (INTEGRATION.md once showed one) gets an error,
    not a kernel launched on strides read from past the end of its struct.  Host logic only: no GPU needed."""
    from sfvos_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip('libsfvos.so not built (run `python __graft_entry__.py`)')
    lib = _lib.load()
    d = _lib.ConvDesc()
    d.dtype, d.batch, d.t_in, d.t_alloc, d.t_offset = _lib.BF16, 1, 7, 7, 0
    d.c_in, d.c_out, d.kt, d.taps, d.pad_t, d.ld_x, d.ld_y = 256, 32, 3, 9, 0, 256, 32
    d.pyr = _lib.make_pyramid([(12, 21)])
    assert lib.sfvos_conv3d_stat_rows(ctypes.byref(d), None) > 0
    assert lib.sfvos_conv3d_wgrad_workspace_bytes(ctypes.byref(d)) > 0
    d.struct_size -= 16
    assert lib.sfvos_conv3d_stat_rows(ctypes.byref(d), None) < 0
    assert b'struct_size' in lib.sfvos_last_error()
    assert lib.sfvos_conv3d_wgrad_workspace_bytes(ctypes.byref(d)) == 0


def test_mask_branch_surface_matches_torchvision_names():
    """MaskBranch mirrors roi_heads.mask_head / roi_heads.mask_predictor of torchvision's Mask R-CNN as the reference
    configures it (model.py:17-25): same state-dict keys and shapes as the torch-core restatement."""
    from oracle.mask_head_ref import OracleMaskBranch
    from sfvos_amd import MaskBranch
    m, o = MaskBranch(256, 2), OracleMaskBranch(256, 2)
    sm, so = m.state_dict(), o.state_dict()
    assert list(sm.keys()) == list(so.keys())
    assert 'mask_head.mask_fcn4.weight' in sm and 'mask_predictor.conv5_mask.weight' in sm
    assert tuple(sm['mask_predictor.conv5_mask.weight'].shape) == (256, 256, 2, 2)
    assert tuple(sm['mask_predictor.mask_fcn_logits.weight'].shape) == (2, 256, 1, 1)
    for k in sm:
        assert sm[k].shape == so[k].shape
    m.load_state_dict(so, strict=True)


def test_grad_bucket_force_runs_the_collectives_on_one_rank():
    """GradBucket(force=True) with a world-size-1 process group (here gloo on CPU; tests/test_gpu_rccl.py does the same
    with nccl = RCCL on the GPU box): the bucket is active, every reported range becomes a collective, the result of an
    all-reduce over one rank is the input, and without `force` one rank stays a no-op."""
    import torch.distributed as dist
    from sfvos_amd import GradBucket, init_distributed
    import socket
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    old = {k: os.environ.get(k) for k in ('MASTER_ADDR', 'MASTER_PORT', 'RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
    try:
        assert init_distributed(backend='gloo') == (0, 1, 0) and not dist.is_initialized()   # one rank, not forced: nothing
        assert init_distributed(backend='gloo', force=True) == (0, 1, 0) and dist.is_initialized()
        flat = torch.arange(4096, dtype=torch.float32)
        idle = GradBucket(flat.clone())
        assert not idle.active
        idle.arm(); assert not idle.armed
        idle.all_reduce(); assert idle.collectives == 0
        b = GradBucket(flat, force=True)
        assert b.active and b.world == 1
        b.set_buckets([(0, 1000), (1000, 4096)])
        b.arm()
        b.segment_ready(1000, 3000); assert b.collectives == 0      # bucket (1000, 4096) incomplete
        b.segment_ready(3000, 4096); assert b.collectives == 1
        b.segment_ready(0, 1000); assert b.collectives == 2
        b.finish()
        assert b.collectives == 2 and torch.equal(flat, torch.arange(4096, dtype=torch.float32))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def test_packed_clip_layouts_and_roi_align_host_logic():
    from sfvos_amd import MultiScaleRoIAlign, PackedClip, roi_align
    M = 2 * (3 * 4 + 2 * 2)
    assert PackedClip(torch.zeros(M, 256, dtype=torch.bfloat16), [(3, 4), (2, 2)], 1, 2).layout == 'ndhwc'
    assert PackedClip(torch.zeros(8, M, 32, dtype=torch.bfloat16), [(3, 4), (2, 2)], 1, 2).layout == 'grouped'
    c8 = PackedClip(torch.zeros(4, M, 64, dtype=torch.uint8), [(3, 4), (2, 2)], 1, 2)
    assert c8.layout == 'grouped8' and c8.channels == 256 and c8.window == 2
    with pytest.raises(ValueError):
        PackedClip(torch.zeros(4, M, 64, dtype=torch.bfloat16), [(3, 4), (2, 2)], 1, 2)   # 64-wide groups are e4m3 bytes
    with pytest.raises(ValueError):
        PackedClip(torch.zeros(M + 1, 256), [(3, 4), (2, 2)], 1, 2)
    # scales as torchvision infers them: the power of two nearest to feature size / image size
    pool = MultiScaleRoIAlign(['0', '1', '2', '3'], 14, 2)
    feats = [torch.zeros(1, 8, h, w) for h, w in ((192, 336), (96, 168), (48, 84), (24, 42))]
    pool.setup_scales(feats, [(749, 1333)])
    assert pool.scales == [0.25, 0.125, 0.0625, 0.03125] and (pool.k_min, pool.k_max) == (2, 5)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        roi_align(torch.zeros(1, 8, 4, 4), torch.zeros(1, 5), 7, 1.0, 2)
    with pytest.raises(NotImplementedError):
        MultiScaleRoIAlign(['0'], 14, 0)


def test_asm_mfma_kernels_pass_the_assembly_audit():
    """The wide conv kernel and the 16x16x32 weight gradients issue their MFMAs as inline asm with a tied accumulator
    (csrc/conv3d.hip, csrc/wgrad.hip); hipcc neither pads wait states around asm nor knows an MFMA sits inside, so what
    the kernels rely on is checked on the compiler's gfx950 assembly (tools/diag/audit_asm_mfma.py): no scratch, every
    accumulator tied, no other instruction on an accumulator register inside the MFMA loops, and the first readers
    behind them separated by the kernels' s_nop pairs."""
    import importlib.util
    import shutil
    if not (shutil.which('hipcc') or os.path.exists('/opt/rocm/bin/hipcc')):
        pytest.skip('hipcc not available')
    spec = importlib.util.spec_from_file_location('audit_asm_mfma', os.path.join(ROOT, 'tools', 'diag', 'audit_asm_mfma.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    n, bad = mod.main([])
    assert n >= 20 and bad == 0
