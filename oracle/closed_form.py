"""RNG-free, exactly reproducible test data (TEST INFRASTRUCTURE ONLY).

Values come from a 32-bit integer hash of the element index, so the golden
generator (build container, next to the reference) and the tests (anywhere)
rebuild bit-identical fp32 weights and inputs without sharing files and
without depending on any library's random stream.  (SURVEY.md 8c asks for
closed-form data; an integer hash is used instead of sin/cos so that no libm
rounding difference can leak into the vectors.)
"""
from collections import OrderedDict

import numpy as np
import torch

_M32 = np.uint64(0xFFFFFFFF)


def hash_uniform(n, salt):
    """n fp32 values in [-0.5, 0.5), exact multiples of 2**-24."""
    i = np.arange(n, dtype=np.uint64)
    h = (i * np.uint64(0x9E3779B1) + np.uint64(salt) * np.uint64(0x85EBCA77) + np.uint64(0x165667B1)) & _M32
    h ^= h >> np.uint64(15)
    h = (h * np.uint64(0x2C1B3C6D)) & _M32
    h ^= h >> np.uint64(12)
    h = (h * np.uint64(0x297A2D39)) & _M32
    h ^= h >> np.uint64(15)
    v = (h >> np.uint64(8)).astype(np.float64) / float(1 << 24) - 0.5
    return v.astype(np.float32)


def _salt(name):
    s = 0
    for ch in name.encode():
        s = (s * 131 + ch) & 0x7FFFFFFF
    return s


def closed_form_tensor(shape, name, scale=1.0, offset=0.0):
    n = int(np.prod(shape)) if len(shape) else 1
    v = hash_uniform(n, _salt(name)) * np.float32(scale) + np.float32(offset)
    return torch.from_numpy(v.reshape(shape).copy())


def closed_form_state_dict(module):
    """Deterministic values for every parameter/buffer of a SlowFastLayers-shaped
    module (reference or build), keyed by state-dict name.

    conv weights ~ U(-a, a) with a = sqrt(3 / fan_in) (unit-ish output variance),
    conv bias U(-0.1, 0.1), BN weight in [0.75, 1.25], BN bias U(-0.2, 0.2),
    running_mean U(-0.1, 0.1), running_var in [0.75, 1.25].
    """
    out = OrderedDict()
    for key, ref in module.state_dict().items():
        shape = tuple(ref.shape)
        if key.endswith('num_batches_tracked'):
            out[key] = torch.zeros((), dtype=torch.int64)
        elif key.endswith('running_mean'):
            out[key] = closed_form_tensor(shape, key, 0.2)
        elif key.endswith('running_var'):
            out[key] = closed_form_tensor(shape, key, 0.5, 1.0)
        elif key.startswith('bn_') and key.endswith('weight'):
            out[key] = closed_form_tensor(shape, key, 0.5, 1.0)
        elif key.startswith('bn_') and key.endswith('bias'):
            out[key] = closed_form_tensor(shape, key, 0.4)
        elif key.endswith('bias'):
            out[key] = closed_form_tensor(shape, key, 0.2)
        else:  # conv weight [Cout, Cin, kt, kh, kw]
            fan_in = shape[1] * shape[2] * shape[3] * shape[4]
            a = float(np.sqrt(3.0 / fan_in))
            out[key] = closed_form_tensor(shape, key, 2.0 * a)
    return out


LEVEL_KEYS = ['0', '1', '2', '3', 'pool']


def closed_form_features(num_frames, level_shapes, clip=0, channels=256, zero_frames=()):
    """One OrderedDict level-key -> [num_frames, channels, H, W] fp32, the
    shape SegmentationModel hands to temporally_enhance_features
    (reference code/helpers/model.py:336-340).  ``zero_frames`` are zeroed to mimic the
    reference's zero feature padding at sequence ends (model.py:215-225)."""
    feats = OrderedDict()
    for key, (h, w) in level_shapes.items():
        t = closed_form_tensor((num_frames, channels, h, w), 'feat/%s/%d' % (key, clip), 2.0)
        for z in zero_frames:
            t[z] = 0
        feats[key] = t
    return feats


def slice_slow(fast_feats, sp):
    """Centre ``sp`` frames of the fast window, as reference model.py:242-248 with
    image_feature_idx = fp // 2 (model.py:322,337)."""
    out = OrderedDict()
    for key, value in fast_feats.items():
        fp = value.shape[0]
        idx = fp // 2
        out[key] = value[idx - sp // 2: idx + (sp + 1) // 2]
    return out
