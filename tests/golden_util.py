"""Shared helpers for the parity tests: rebuild the closed-form inputs of
oracle/make_golden.py and compare a SlowFastLayers-shaped module's results with
the fixtures generated from the reference's own class."""
import json
import os
from collections import OrderedDict

import numpy as np
import torch

from oracle.closed_form import closed_form_features, closed_form_state_dict, slice_slow

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
CONFIGS = [(1, 1), (3, 3), (7, 7), (1, 7), (3, 7), (4, 32), (4, 64)]
SMALL_LEVELS = OrderedDict([('0', (12, 21)), ('pool', (6, 10))])
BIG_LEVELS = OrderedDict([('0', (24, 42))])
NSAMP = 64


def load_tables():
    with open(os.path.join(GOLDEN, 'tables.json')) as f:
        return json.load(f)


def load_case(sp, fp, tag='small'):
    return dict(np.load(os.path.join(GOLDEN, 'sf_%d_%d_%s.npz' % (sp, fp, tag))))


def sample_idx(numel):
    return (np.arange(NSAMP, dtype=np.int64) * 7919) % numel


def clip_inputs(sp, fp, levels, clip, device=None, clips=(0, 1)):
    """clip = 0/1: first / second accumulated clip of a fixture; `clips` = the fixture's `clips` entry (the
    closed-form clip ids oracle/make_golden.py picked for their ReLU margin).  The second clip has its
    first frame zeroed (mimics the reference's zero feature padding at sequence ends)."""
    zero = (0,) if (clip == 1 and fp > 1) else ()
    fast = closed_form_features(fp, levels, clip=int(clips[clip]), zero_frames=zero)
    if device is not None:
        fast = OrderedDict((k, v.to(device)) for k, v in fast.items())
    return [slice_slow(fast, sp)], [fast]


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def max_rel_err(a, b):
    """max |a-b| / max|b|  -- the '1e-3 relative' of BASELINE.json, taken against
    the tensor's own scale so near-zero logits do not blow the ratio up."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))
