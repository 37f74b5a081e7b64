"""CPU: the oracle restatement reproduces the fixtures generated from the
reference's own SlowFastLayers class (oracle/make_golden.py)."""
import numpy as np
import pytest
import torch

from golden_util import (BIG_LEVELS, CONFIGS, SMALL_LEVELS, clip_inputs, load_case, load_tables, max_rel_err,
                         sample_idx)
from oracle.closed_form import closed_form_state_dict, hash_uniform
from oracle.slowfast_ref import (OracleSlowFastLayers, lateral_kernel_size, proxy_loss, sgd_step_,
                                 temporal_kernel_sizes)

TOL = 2e-5  # same ATen kernels on both sides; only thread-order noise is expected


def test_hash_uniform_is_stable():
    v = hash_uniform(5, 7)
    assert v.dtype == np.float32
    assert np.all(np.abs(v) <= 0.5)
    # known-answer: guards against accidental edits of the closed-form generator
    assert np.array_equal(v, hash_uniform(5, 7))
    assert len(np.unique(hash_uniform(1000, 1))) > 990


def test_kernel_size_tables():
    t = load_tables()
    for p, ks in t['calc_kernel_sizes'].items():
        assert list(temporal_kernel_sizes(int(p))) == ks
        assert sum(ks) == int(p) + 2
    for key, ks in t['kernel_sizes'].items():
        sp, fp = map(int, key.split('-'))
        m = OracleSlowFastLayers(256, torch.device('cpu'), sp, fp)
        assert list(m.kernel_sizes['slow']) == ks['slow']
        assert list(m.kernel_sizes['fast']) == ks['fast']
        assert list(m.kernel_sizes['lateral']) == ks['lateral']
        assert sum(p.numel() for p in m.parameters()) == t['param_counts'][key]


def test_published_param_count_differences():
    # final_report/chapters/Experiments.tex:20-24 (total params incl. Mask R-CNN)
    published = {'1-1': 45421851, '3-3': 46398747, '7-7': 48407835, '1-7': 45618459, '3-7': 46570779}
    t = load_tables()['param_counts']
    for a in published:
        for b in published:
            assert published[a] - published[b] == t[a] - t[b]


def test_state_dict_keys_and_order():
    t = load_tables()['state_dict']
    m = OracleSlowFastLayers(256, torch.device('cpu'), 3, 7)
    sd = m.state_dict()
    assert list(sd.keys()) == list(t.keys())
    for k, (shape, dtype) in t.items():
        assert list(sd[k].shape) == shape and str(sd[k].dtype) == dtype


def _run_oracle(sp, fp, levels, g):
    m = OracleSlowFastLayers(256, torch.device('cpu'), sp, fp)
    m.load_state_dict(closed_form_state_dict(m))
    m.eval()
    with torch.no_grad():
        slow, fast = clip_inputs(sp, fp, levels, 0, clips=g['clips'])
        out = m.temporally_enhance_features(slow, fast)
    for k, v in out.items():
        assert max_rel_err(v.numpy(), g['eval_out/%s' % k]) < TOL
        assert np.array_equal(v.argmax(1).numpy(), torch.from_numpy(g['eval_out/%s' % k]).argmax(1).numpy())
    m.train()
    params = list(m.parameters())
    bufs = [None] * len(params)
    for clip in (0, 1):
        slow, fast = clip_inputs(sp, fp, levels, clip, clips=g['clips'])
        out = m.temporally_enhance_features(slow, fast)
        loss = proxy_loss(out)
        loss.backward()
        assert abs(loss.item() - float(g['loss/%d' % clip])) < TOL * abs(float(g['loss/%d' % clip]))
        if clip == 0:
            for k, v in out.items():
                assert max_rel_err(v.detach().numpy(), g['out/0/%s' % k]) < TOL
    for key, p in m.named_parameters():
        gr = p.grad.reshape(-1)
        assert abs(gr.double().norm().item() - float(g['gnorm/%s' % key])) <= 1e-4 * float(g['gnorm/%s' % key]) + 1e-7
        ref = g['gsamp/%s' % key]
        assert np.abs(gr[sample_idx(gr.numel())].numpy() - ref).max() <= 1e-4 * np.abs(ref).max() + 1e-6
    for key, b in m.named_buffers():
        if key.endswith('num_batches_tracked'):
            assert int(b) == int(g['stat/%s' % key])
        else:
            assert max_rel_err(b.numpy(), g['stat/%s' % key]) < TOL
    sgd_step_(params, bufs)
    for key, p in m.named_parameters():
        v = p.detach().reshape(-1)
        assert abs(v.double().norm().item() - float(g['pnorm/%s' % key])) <= 1e-6 * float(g['pnorm/%s' % key])
        assert np.abs(v[sample_idx(v.numel())].numpy() - g['psamp/%s' % key]).max() < 1e-6


@pytest.mark.parametrize('sp,fp', CONFIGS)
def test_oracle_matches_reference_small(sp, fp):
    _run_oracle(sp, fp, SMALL_LEVELS, load_case(sp, fp, 'small'))


def test_oracle_matches_reference_big():
    _run_oracle(3, 7, BIG_LEVELS, load_case(3, 7, 'big'))


def test_lateral_formula_examples():
    # SURVEY.md 8a row a3
    def lat(sp, fp):
        ks, kf = temporal_kernel_sizes(sp), temporal_kernel_sizes(fp)
        l1, so, fo = lateral_kernel_size(sp, ks[0], fp, kf[0])
        l2, _, _ = lateral_kernel_size(so, ks[1], fo, kf[1])
        return l1, l2
    assert lat(3, 7) == (3, 2) and lat(1, 7) == (5, 3) and lat(4, 32) == (20, 11) and lat(4, 64) == (41, 21)
    assert lat(3, 3) == (1, 1)
