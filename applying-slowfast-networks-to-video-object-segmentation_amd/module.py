"""SlowFastLayers -- drop-in for the reference class (code/helpers/model.py:30-165) whose
forward and backward run in the hand-written HIP kernels of libsfvos.so.

Same constructor, `forward`, `temporally_enhance_features`, parameter names, state-dict keys
and train/eval semantics as the reference; consumed the same way by SegmentationModel
(model.py:184,340), OsvosSegmentationModel (osvos/osvos_model.py:28,65), train.py:74-80.

All FPN levels of one temporally_enhance_features call are processed together: every conv / BN /
weight-gradient step is ONE libsfvos launch over the whole pyramid (include/sfvos.h, "pyramid
NDHWC"), with per-level BatchNorm statistics exactly as the reference's per-level calls produce.

PyTorch is used for device memory, streams and autograd bookkeeping only: every arithmetic
step of the path is a libsfvos call.  There is no CPU / eager fallback."""
import ctypes
import os
from collections import OrderedDict

import torch
from torch import nn

from . import _lib
from .plan import SlowFastPlan

_DT = {'fp32': (_lib.F32, torch.float32), 'bf16': (_lib.BF16, torch.bfloat16),
       # 'fp8': inference only; the four Cin = 256 convs (fast_conv1, slow_conv1-3: 99 % of the forward FLOPs) run on
       # e4m3 operands, the 32-channel layers and every conv RESULT stay bf16
       'fp8': (_lib.BF16, torch.bfloat16)}
# per-level coefficient rows of a BN layer: mean, rstd, scale, shift, var_unbiased, A, B, K, sum dz, sum dz*xhat
_CF_ROWS = 10
_MEAN, _RSTD, _SCALE, _SHIFT, _VARU, _CA, _CB, _CK, _SDZ, _SDZX = range(10)


def _ptr(t, elem_offset=0):
    return ctypes.c_void_p(t.data_ptr() + elem_offset * t.element_size())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _planar_levels(tensors):
    """sfvos_planar_level array for a list of [B,C,H,W] fp32 tensors (one per pyramid level)."""
    arr = (_lib.PlanarLevel * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i].ptr = t.data_ptr()
        arr[i].stride_t, arr[i].stride_c, arr[i].stride_h, arr[i].stride_w = t.stride(0), t.stride(1), t.stride(2), t.stride(3)
        arr[i].h, arr[i].w = t.shape[2], t.shape[3]
    return arr


class KernelTimer(object):
    """HIP-event timing of the libsfvos launches (recorded on the stream they are launched on).
    Enabled only by bench.py / profiling; costs two event records per region."""

    def __init__(self):
        self.records = []

    def reset(self):
        torch.cuda.synchronize()
        self.records = []

    class _Region(object):
        def __init__(self, timer, name):
            self.timer, self.name = timer, name

        def __enter__(self):
            self.start = torch.cuda.Event(enable_timing=True)
            self.end = torch.cuda.Event(enable_timing=True)
            self.start.record(torch.cuda.current_stream())

        def __exit__(self, *exc):
            self.end.record(torch.cuda.current_stream())
            self.timer.records.append((self.name, self.start, self.end))
            return False

    def region(self, name):
        return KernelTimer._Region(self, name)

    def summary(self):
        torch.cuda.synchronize()
        agg = {}
        for name, s, e in self.records:
            c, t = agg.get(name, (0, 0.0))
            agg[name] = (c + 1, t + s.elapsed_time(e))
        return {k: (c, t / c) for k, (c, t) in agg.items()}


class _NullRegion(object):
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NULL = _NullRegion()


def union_mask(masks, threshold=0.5):
    """Evaluation-side reducer of the reference (code/helpers/davis_evaluate.py:40-42): the union over the
    predicted instance masks [N,1,H,W] (fp32 probabilities, on the GPU) of `mask >= threshold` -> bool [H,W]."""
    if not masks.is_cuda:
        raise RuntimeError('union_mask runs on the GPU through libsfvos.so (no CPU fallback)')
    if masks.dim() != 4 or masks.shape[1] != 1:
        raise RuntimeError('union_mask expects [N,1,H,W], got %s' % (tuple(masks.shape),))
    m = masks if (masks.dtype == torch.float32 and masks.is_contiguous()) else masks.float().contiguous()
    H, W = m.shape[2], m.shape[3]
    out = torch.empty((H, W), dtype=torch.uint8, device=m.device)
    _lib.call('sfvos_mask_union', _ptr(m) if m.shape[0] else None, m.shape[0], H * W, float(threshold), _ptr(out),
              _stream())
    return out.bool()


class PackedClip(object):
    """Channels-last clips of a whole pyramid in the layout the kernels consume (a
    producer that emits NHWC frames hands them over without any layout pass).

    data : [M, C] tensor in the compute dtype, M = B * frames * sum_l(H_l*W_l) positions, level-major
           (position(l,b,t,h,w) as in include/sfvos.h) -- layout 'ndhwc'; or
           [C/32, M, 32] (bf16 only) -- layout 'grouped': the channels of a position are stored as 64-byte groups,
           group-major (sfvos_conv_desc.x_group_stride).  The first convs stream the input one 64-byte channel
           chunk at a time; in this layout every 128-byte line they touch is used whole (fewer L2 misses); or
           uint8 [C/64, M, 64] -- layout 'grouped8': OCP e4m3 bytes in 64-channel groups, quantised with
           SlowFastLayers.fp8_input_scale (SlowFastLayers.pack_fp8 builds it; precision='fp8', inference only).
    """

    def __init__(self, data, shapes, batch, frames, keys=None, pad=(0, 0)):
        """frames: frames per clip the buffer STORES.  pad = (before, after): zero frames in front of / behind them that
        complete the fast window (pad[0] + frames + pad[1] = fast_pathway_size) -- the reference's zero feature padding
        beyond the ends of a sequence (model.py:215-225), handed over "by pointer": the padding frames have no storage
        and are never read (sfvos_conv_desc.t_offset / t_alloc)."""
        self.data, self.shapes, self.batch, self.frames = data, [tuple(s) for s in shapes], batch, frames
        self.keys = list(keys) if keys is not None else [str(i) for i in range(len(shapes))]
        self.pad = (int(pad[0]), int(pad[1]))
        if self.pad[0] < 0 or self.pad[1] < 0 or frames < 1:
            raise ValueError('PackedClip: pad must be non-negative and at least one frame stored')
        M = batch * frames * sum(h * w for h, w in self.shapes)
        ok = (data.dim() == 2 and data.shape[0] == M) or (data.dim() == 3 and data.shape[1] == M and data.shape[2] == 32) \
            or (data.dim() == 3 and data.shape[1] == M and data.shape[2] == 64 and data.dtype == torch.uint8)
        if not ok:
            raise ValueError('PackedClip: data must be [%d, C] (ndhwc), [C/32, %d, 32] (grouped) or uint8 [C/64, %d, 64] '
                             '(e4m3 groups), got %s' % (M, M, M, tuple(data.shape)))

    @property
    def window(self):
        """Frames of the fast window this clip stands for (stored + zero padding)."""
        return self.pad[0] + self.frames + self.pad[1]

    @property
    def layout(self):
        if self.data.dim() == 3:
            return 'grouped8' if self.data.dtype == torch.uint8 else 'grouped'
        return 'ndhwc'

    @property
    def channels(self):
        return self.data.shape[0] * self.data.shape[2] if self.data.dim() == 3 else self.data.shape[1]

    @staticmethod
    def from_levels(levels, keys=None, layout='ndhwc', pad=(0, 0)):
        """levels: list of [B,T,H,W,C] tensors (one per FPN level, the T STORED frames) -> PackedClip (copies once)."""
        B, T = levels[0].shape[0], levels[0].shape[1]
        shapes = [tuple(x.shape[2:4]) for x in levels]
        if layout == 'grouped':
            C = levels[0].shape[-1]
            if C % 32 != 0 or levels[0].dtype != torch.bfloat16:
                raise ValueError("PackedClip layout 'grouped' needs bf16 data with C a multiple of 32")
            data = torch.cat([x.reshape(-1, C // 32, 32).permute(1, 0, 2) for x in levels], 1).contiguous()
        elif layout == 'ndhwc':
            data = torch.cat([x.reshape(-1, x.shape[-1]) for x in levels], 0)
        else:
            raise ValueError("PackedClip layout must be 'ndhwc' or 'grouped'")
        return PackedClip(data, shapes, B, T, keys, pad)


class _State(object):
    """What one forward leaves behind for its backward."""
    __slots__ = ('B', 'shapes', 'bufs', 'coef', 'train', 'dtype_name', 'slow_offset', 'x_frames', 'x_pad')


def _lv_total(lv):
    return sum(lv.m[i] for i in range(lv.n_levels))


class SlowFastLayers(nn.Module):
    def __init__(self, input_size, device, slow_pathway_size, fast_pathway_size, precision=None):
        super(SlowFastLayers, self).__init__()
        self.device = device
        self.slow_pathway_size = slow_pathway_size
        self.fast_pathway_size = fast_pathway_size
        self.plan = SlowFastPlan(input_size, slow_pathway_size, fast_pathway_size)
        shapes = self.plan.conv_shapes()
        bn_channels = {l.bn: l.c_out for l in self.plan.layers}
        # registration order == reference order (state-dict order and RNG order of the default init)
        for name in self.plan.module_order:
            if name in shapes:
                c_in, c_out, kt, kh, kw, bias = shapes[name]
                if bias:
                    mod = nn.Conv3d(in_channels=c_in, out_channels=c_out, kernel_size=(kt, kh, kw), padding=(0, 1, 1))
                else:
                    mod = nn.Conv3d(c_in, c_out, kernel_size=[kt, 1, 1], stride=[1, 1, 1], padding=[0, 0, 0],
                                    bias=False)
            else:
                mod = nn.BatchNorm3d(bn_channels[name])
            self.add_module(name, mod)
        self.precision = precision or os.environ.get('SFVOS_PRECISION', 'fp32')
        if self.precision not in _DT:
            raise ValueError("precision must be 'fp32', 'bf16' or 'fp8', got %r" % (self.precision,))
        self.fp8_input_scale = 32.0   # e4m3 activation scale of the input clip (|x| * scale must stay below 448)
        self.fp8_act_scale = 32.0     # ... of the slow pathway's concat buffers (post-ReLU BatchNorm outputs)
        self._fp8_sat = None          # device int: input elements that saturated in e4m3 since the last check
        self._packs = {}     # (conv name, kind, dtype) -> ((param version, data_ptr, epoch), packed tensor)
        self._timer = None
        self._timer_only = None
        self._side = None
        self.n_streams = int(os.environ.get('SFVOS_STREAMS', '2'))
        self.f1_alone = True   # with two streams: the side stream starts behind fast_conv1's forward launch

    def enable_kernel_timer(self, only=None):
        """HIP-event timing of the launches; `only`: an iterable of region names ('conv_fwd/f1', ...) -- every other
        launch goes out without event records (an event record between two kernels costs about a microsecond of
        GPU time, ~100 of them per step)."""
        self._timer = KernelTimer()
        self._timer_only = None if only is None else frozenset(only)
        return self._timer

    # ------------------------------------------------------------------ e4m3 activation scale (precision='fp8')
    def fp8_saturated(self, reset=True):
        """Input elements that exceeded the e4m3 range (|x| * fp8_input_scale > 448, stored as +-448) since the
        last call -- FPN features are unbounded conv outputs, so a fixed scale has to be checked against the data.
        Synchronises the device."""
        if self._fp8_sat is None:
            return 0
        n = int(self._fp8_sat.item())
        if reset:
            self._fp8_sat.zero_()
        return n

    def calibrate_fp8_scale(self, fast_features, headroom=2.0):
        """Set fp8_input_scale = 448 / (headroom * max|x|) from representative inputs (same structure as the
        fast_features argument of temporally_enhance_features: list of dict level -> [T,C,H,W] fp32 on the GPU)."""
        _lib.load()
        amax = None
        for feats in fast_features:
            for v in feats.values():
                if not v.is_cuda:
                    raise RuntimeError('calibrate_fp8_scale runs on the GPU through libsfvos.so (no CPU fallback)')
                s = v if v.dtype == torch.float32 else v.float()
                if amax is None:
                    amax = torch.zeros(1, dtype=torch.float32, device=s.device)
                _lib.call('sfvos_frames_absmax', _ptr(s), s.stride(0), s.stride(1), s.stride(2), s.stride(3),
                          s.shape[0], s.shape[1], s.shape[2], s.shape[3], _ptr(amax), _stream())
        m = float(amax.item()) if amax is not None else 0.0
        if m > 0.0:
            self.fp8_input_scale = 448.0 / (headroom * m)
        return self.fp8_input_scale

    def pack_fp8(self, fast_features):
        """Quantise a fast window into the resident e4m3 clip `enhance_packed` takes with precision='fp8': same argument
        as temporally_enhance_features' fast_features (list over clips of dict level -> [T,C,H,W] fp32 on the GPU);
        scale = self.fp8_input_scale (calibrate_fp8_scale), saturation counted (fp8_saturated)."""
        keys = list(fast_features[0].keys())
        ts = [(torch.stack([d[k] for d in fast_features]) if len(fast_features) > 1 else fast_features[0][k].unsqueeze(0))
              .transpose(1, 2) for k in keys]
        self._check_ready(ts[0])
        data = self._to_pyramid_fp8(ts, ts[0].shape[2])
        return PackedClip(data, [tuple(t.shape[3:]) for t in ts], ts[0].shape[0], ts[0].shape[2], keys)

    def _t(self, kind, layer):
        if self._timer is None:
            return _NULL
        name = '%s/%s' % (kind, layer)
        if self._timer_only is not None and name not in self._timer_only:
            return _NULL
        return self._timer.region(name)

    # ------------------------------------------------------------------ helpers
    def _check_ready(self, ref):
        if not ref.is_cuda:
            raise RuntimeError('sfvos_amd.SlowFastLayers runs only on an MI355X (gfx950) through libsfvos.so; '
                               'got a %s tensor. There is no CPU fallback.' % ref.device)
        _lib.load()
        w = self.fast_conv1.weight
        if w.device != ref.device:
            raise RuntimeError('module parameters are on %s but inputs on %s' % (w.device, ref.device))

    def _packed(self, layer, kind, dt_name):
        dt_id, tdt = _DT[dt_name]
        w = getattr(self, layer.conv).weight
        key = (layer.conv, kind, dt_name)
        hit = self._packs.get(key)
        tag = (w._version, w.data_ptr(), _lib.weight_epoch())
        if hit is not None and hit[0] == tag:
            return hit[1]
        # a miss: the weights changed (optimiser step).  Every image of this kind that has been used before is stale for
        # the same reason -- refresh them all in ONE launch (sfvos_pack_weights_batch) instead of one per layer
        todo = [(layer, key, w, tag)]
        for l2 in self.plan.layers:
            k2 = (l2.conv, kind, dt_name)
            if k2 == key or k2 not in self._packs:
                continue
            w2 = getattr(self, l2.conv).weight
            t2 = (w2._version, w2.data_ptr(), _lib.weight_epoch())
            if self._packs[k2][0] != t2 and w2.device == w.device:
                todo.append((l2, k2, w2, t2))
        todo = todo[:_lib.MAX_PACK_ITEMS]
        items = (_lib.PackItem * len(todo))()
        hold = []
        for i, (l2, k2, w2, t2) in enumerate(todo):
            wc = w2.detach()
            if wc.dtype != torch.float32 or not wc.is_contiguous():
                wc = wc.float().contiguous()
            img = torch.empty(w2.numel(), dtype=tdt, device=w2.device)
            hold.append(wc)
            items[i].w, items[i].packed = wc.data_ptr(), img.data_ptr()
            items[i].c_out, items[i].c_in, items[i].kt, items[i].taps = l2.c_out, l2.c_in, l2.kt, l2.taps
            items[i].dgrad = 0 if kind == 'fwd' else 1
            self._packs[k2] = (t2, img)
        _lib.call('sfvos_pack_weights_batch', items, len(todo), dt_id, _stream())
        return self._packs[key][1]

    def _packed_fp8(self, layer, act_scale):
        """(e4m3 weight image, [3][c_out] (bias, descale, weight scale) rows) of a 3x3 layer whose input was quantised
        with act_scale, cached like _packed."""
        conv = getattr(self, layer.conv)
        w = conv.weight
        key = (layer.conv, 'fwd8', float(act_scale))
        hit = self._packs.get(key)
        tag = (w._version, w.data_ptr(), _lib.weight_epoch(), None if conv.bias is None else conv.bias._version)
        if hit is not None and hit[0] == tag:
            return hit[1]
        packed = torch.empty(w.numel(), dtype=torch.uint8, device=w.device)
        bd = torch.empty((3, layer.c_out), dtype=torch.float32, device=w.device)
        wc = w.detach()
        if wc.dtype != torch.float32 or not wc.is_contiguous():
            wc = wc.float().contiguous()
        bc = None if conv.bias is None else conv.bias.detach().float().contiguous()
        _lib.call('sfvos_pack_weights_fp8', _ptr(wc), _ptr(bc) if bc is not None else None, _ptr(packed), _ptr(bd),
                  layer.c_out, layer.c_in, layer.kt, layer.taps, float(act_scale), _stream())
        self._packs[key] = (tag, (packed, bd))
        return packed, bd

    def _desc(self, layer, B, pyr, dt_id, ld_x, ld_y, t_alloc=None, t_offset=0, dgrad=False, accumulate=0):
        """ld_x: pitch of x in elements, or the x tensor itself ([M, ld] ndhwc / [G, M, 32] grouped)."""
        d = _lib.ConvDesc()
        d.x_group_stride = d.x_frame_stride = d.y_frame_stride = 0
        if torch.is_tensor(ld_x):
            if ld_x.dim() == 3:   # channel-group-major: [C/32][M][32] bf16 or [C/64][M][64] e4m3
                d.x_group_stride, ld_x = ld_x.shape[1] * ld_x.shape[2], ld_x.shape[2]
            else:
                ld_x = ld_x.shape[-1]
        d.dtype, d.batch, d.kt, d.taps, d.pyr = dt_id, B, layer.kt, layer.taps, pyr
        if dgrad:  # conv over dy producing dx: channels swapped, full temporal padding
            d.t_in, d.c_in, d.c_out, d.pad_t = layer.t_out, layer.c_out, layer.c_in, layer.kt - 1
            d.t_alloc, d.t_offset = layer.t_out, 0
        else:
            d.t_in, d.c_in, d.c_out, d.pad_t = layer.t_in, layer.c_in, layer.c_out, 0
            d.t_alloc, d.t_offset = (layer.t_in if t_alloc is None else t_alloc), t_offset
        d.ld_x, d.ld_y, d.accumulate = ld_x, ld_y, accumulate
        return d

    def _src_window(self, layer, slow_offset, x_frames=None, x_pad=0):
        """(buffer name, t_alloc, t_offset) of a layer's input.  The slow pathway's first conv reads its centre frames
        straight out of the fast clip when the caller's slow tensor aliases it; a clip that stores only x_frames of
        the fast window behind x_pad zero frames (PackedClip.pad) is addressed with a shifted window: the frames
        that fall outside its buffer are zero frames the kernels never read."""
        fp = self.plan.fp if x_frames is None else x_frames
        if layer.src == 'xs0' and slow_offset is not None:
            return 'xf0', fp, slow_offset - x_pad
        if layer.src == 'xf0' and x_frames is not None:
            return 'xf0', fp, -x_pad
        return layer.src, None, 0

    # ------------------------------------------------------------------ streams
    def _streams(self, dev):
        """(main, side): the fast pathway runs on the caller's current stream, the slow pathway (slow convs +
        laterals) on a side HIP stream; events order the lateral fusions.  The two pathways are independent
        between fusions, so HBM-bound BN passes of one overlap MFMA-bound convs of the other and launch tails
        are filled.  SFVOS_STREAMS=1 (or n_streams = 1) serialises everything on the current stream."""
        main = torch.cuda.current_stream(dev)
        if self.n_streams < 2:
            return main, None
        if self._side is None or self._side.device != dev:
            self._side = torch.cuda.Stream(device=dev)
        return main, self._side

    @staticmethod
    def _on_side(layer):
        return layer.name[0] in 'sl'

    # ------------------------------------------------------------------ forward engine (whole pyramid)
    def _engine_forward(self, shapes, B, xf0, xs0, slow_offset, keep, x_frames=None, x_pad=0):
        """xf0 / xs0: flat pyramid buffers [B*T*sum(HW), C] in the compute dtype (xs0 None when the slow
        clip aliases frames [slow_offset, slow_offset+sp) of the fast clip).
        Returns (list of merged [B,256,H,W] fp32 per level, state or None)."""
        plan = self.plan
        fp8 = self.precision == 'fp8'
        dt_name = 'bf16' if fp8 else self.precision
        dt_id, tdt = _DT[dt_name]
        dev = xf0.device
        L = len(shapes)
        if fp8 and (self.training or keep):
            raise RuntimeError("precision='fp8' is inference-only (eval mode, no autograd state): the e4m3 path has "
                               "no backward in this build")
        pix = sum(h * w for h, w in shapes)
        pyr = _lib.make_pyramid(shapes)
        bufs = {'xf0': xf0}
        if xs0 is not None:
            bufs['xs0'] = xs0
        train = self.training
        lib = _lib.load()
        main, side = self._streams(dev)

        # -- phase 1 (current stream): every buffer and workspace of the pass is allocated, and every weight
        # image packed, BEFORE the side stream is forked, and nothing is freed until the streams are joined,
        # so the caching allocator never hands memory still in use on one stream to the other.
        work = {}
        # e4m3 path: the Cin = 256 3x3 convs take e4m3 operands -- the input clip (64-channel groups) and the slow
        # pathway's concat buffers, which BN-apply writes as e4m3 [M][256] directly; conv results stay bf16
        fp8_bufs = ('cat1', 'cat2') if fp8 else ()
        if fp8 and (self._fp8_sat is None or self._fp8_sat.device != dev):
            self._fp8_sat = torch.zeros(1, dtype=torch.int32, device=dev)
        for l in plan.layers:
            for name in (l.raw, l.dst):
                if name not in bufs:
                    b = plan.buffers[name]
                    bufs[name] = torch.empty((B * b.frames * pix, b.channels),
                                             dtype=torch.uint8 if name in fp8_bufs else tdt, device=dev)
            sname, t_alloc, t_off = self._src_window(l, slow_offset, x_frames, x_pad)
            src = bufs[sname]
            lv = _lib.make_levels(shapes, B, l.t_out)
            if fp8 and l.taps == 9 and l.c_in % 64 == 0:   # (bias, descale) rows travel as `bias`
                d = self._desc(l, B, pyr, _lib.FP8, src, l.c_out, t_alloc, t_off)
                wp, bd = self._packed_fp8(l, self.fp8_input_scale if sname == 'xf0' else self.fp8_act_scale)
                w = dict(d=d, lv=lv, src=src, wp=wp, bias8=bd,
                         cf=torch.empty((L, _CF_ROWS, l.c_out), dtype=torch.float32, device=dev))
            else:
                d = self._desc(l, B, pyr, dt_id, src, l.c_out, t_alloc, t_off)
                w = dict(d=d, lv=lv, src=src, wp=self._packed(l, 'fwd', dt_name),
                         cf=torch.empty((L, _CF_ROWS, l.c_out), dtype=torch.float32, device=dev))
            if train:
                if min(lv.m[i] for i in range(L)) <= 1:
                    raise ValueError('Expected more than 1 value per channel when training, got input size %s'
                                     % str([B, l.c_out, l.t_out] + list(shapes[-1])))
                w['rows_pl'] = (ctypes.c_int * _lib.MAX_LEVELS)()
                rows = lib.sfvos_conv3d_stat_rows(ctypes.byref(d), w['rows_pl'])
                if rows <= 0:
                    _lib.check(rows if rows < 0 else -1, 'sfvos_conv3d_stat_rows')
                w['rows'] = rows
                w['part'] = torch.empty((rows, 2, l.c_out), dtype=torch.float32, device=dev)
            work[l.name] = w
        merged = [torch.empty((B, 256, H, W), dtype=torch.float32, device=dev) for (H, W) in shapes]

        # -- phase 2: launches
        if side is not None:
            side.wait_stream(main)
        ev = {}
        coef = {}
        order = list(plan.layers)
        if side is not None and self.f1_alone:
            # fast_conv1's forward (57 % of the forward's time, one workgroup per CU) first and ALONE: the side stream
            # starts behind it.  Interleaving slow_conv1's workgroups with it buys nothing (both fill every CU) and the
            # slow pathway then overlaps the short fast_conv2/3 + BN launches, whose tails it fills.
            order.sort(key=lambda l: l.name != 'f1')
        for l in order:
            conv, bn = getattr(self, l.conv), getattr(self, l.bn)
            w = work[l.name]
            d, lv, src, wp, cf = w['d'], w['lv'], w['src'], w['wp'], w['cf']
            raw, dst = bufs[l.raw], bufs[l.dst]
            cs = _CF_ROWS * l.c_out
            bias = _ptr(conv.bias.detach()) if conv.bias is not None else None
            if 'bias8' in w:
                bias = _ptr(w['bias8'])
            stream = side if (side is not None and self._on_side(l)) else main
            with torch.cuda.stream(stream):
                st = _stream()
                if side is not None:  # lateral fusions: the slow side consumes the fast pathway's activations
                    if l.name == 'l1' and 'f1' in ev:
                        stream.wait_event(ev['f1'])
                    if l.name == 'l2' and 'f2' in ev:
                        stream.wait_event(ev['f2'])
                run = None   # running-statistics update of this layer (training), folded into its BN-apply launch
                if train:
                    with self._t('conv_fwd', l.name):
                        _lib.call('sfvos_conv3d', ctypes.byref(d), _ptr(src), _ptr(wp), bias, _ptr(raw),
                                  _ptr(w['part']), st)
                    if side is not None and self.f1_alone and l.name == 'f1':
                        e0 = torch.cuda.Event()
                        e0.record(stream)
                        side.wait_event(e0)
                    _lib.call('sfvos_bn_finalize', _ptr(w['part']), L, w['rows_pl'], lv.m, _ptr(bn.weight.detach()),
                              _ptr(bn.bias.detach()), float(bn.eps), l.c_out, _ptr(cf[0, _MEAN]), _ptr(cf[0, _RSTD]),
                              _ptr(cf[0, _SCALE]), _ptr(cf[0, _SHIFT]), _ptr(cf[0, _VARU]), cs, st)
                    if bn.track_running_stats and bn.running_mean is not None:
                        # the reference runs the levels one after another: L consecutive momentum updates
                        # (num_batches_tracked += L), done by the first workgroup of the BN-apply launch below
                        if bn.momentum is None:
                            raise RuntimeError('BatchNorm momentum=None (cumulative average) is not supported')
                        run = _lib.BnRunning()
                        run.running_mean, run.running_var = bn.running_mean.data_ptr(), bn.running_var.data_ptr()
                        run.means, run.vars_unbiased = cf[0, _MEAN].data_ptr(), cf[0, _VARU].data_ptr()
                        run.num_batches_tracked = bn.num_batches_tracked.data_ptr()
                        run.n_updates, run.momentum = L, float(bn.momentum)
                        run = ctypes.byref(run)
                else:
                    with self._t('conv_fwd', l.name):
                        _lib.call('sfvos_conv3d', ctypes.byref(d), _ptr(src), _ptr(wp), bias, _ptr(raw), None, st)
                    _lib.call('sfvos_bn_eval_coeffs', _ptr(bn.weight.detach()), _ptr(bn.bias.detach()),
                              _ptr(bn.running_mean), _ptr(bn.running_var), float(bn.eps), l.c_out, _ptr(cf[0, _MEAN]),
                              _ptr(cf[0, _RSTD]), _ptr(cf[0, _SCALE]), _ptr(cf[0, _SHIFT]), st)
                    if L > 1:
                        cf[1:, :4] = cf[0, :4]  # same running statistics for every level (plumbing copy)
                with self._t('bn_apply', l.name):
                    if l.dst in fp8_bufs:
                        _lib.call('sfvos_bn_apply_fp8', _ptr(raw), l.c_out, _ptr(dst, l.dst_off), dst.shape[-1],
                                  ctypes.byref(lv), l.c_out, _ptr(cf[0, _SCALE]), _ptr(cf[0, _SHIFT]), cs,
                                  1 if l.relu else 0, float(self.fp8_act_scale), _ptr(self._fp8_sat), run, st)
                    else:
                        _lib.call('sfvos_bn_apply', _ptr(raw), l.c_out, _ptr(dst, l.dst_off), dst.shape[-1], dt_id,
                                  ctypes.byref(lv), l.c_out, _ptr(cf[0, _SCALE]), _ptr(cf[0, _SHIFT]), cs,
                                  1 if l.relu else 0, run, st)
                if side is not None and l.name in ('f1', 'f2'):
                    ev[l.name] = torch.cuda.Event()
                    ev[l.name].record(stream)
            coef[l.name] = cf
        if side is not None:
            main.wait_stream(side)

        # -- cat([slow224, fast32], 1).squeeze(2) (model.py:162) as the caller's NCHW fp32 tensors
        st = _stream()
        out = bufs['out']
        _lib.call('sfvos_pyramid_to_frames', _ptr(out), dt_id, _planar_levels(merged), len(merged), B, 256, 256, 0, st)
        if not keep:
            return merged, None
        state = _State()
        state.B, state.shapes, state.bufs, state.coef = B, shapes, bufs, coef
        state.train, state.dtype_name, state.slow_offset = train, dt_name, slow_offset
        state.x_frames, state.x_pad = x_frames, x_pad
        del bufs['out']
        return merged, state

    # ------------------------------------------------------------------ backward engine (whole pyramid)
    def _engine_backward(self, state, g_merged, need_slow, need_fast, need_param):
        plan = self.plan
        dt_name = state.dtype_name
        dt_id, tdt = _DT[dt_name]
        B, shapes, bufs, coef = state.B, state.shapes, state.bufs, state.coef
        dev = bufs['xf0'].device
        lib = _lib.load()
        pix = sum(h * w for h, w in shapes)
        pyr = _lib.make_pyramid(shapes)
        main, side = self._streams(dev)
        gb = {}  # gradient buffers w.r.t. activation buffers

        def galloc(name, zero=False):
            b = plan.buffers[name]
            fn = torch.zeros if zero else torch.empty
            gb[name] = fn((B * b.frames * pix, b.channels), dtype=tdt, device=dev)
            return gb[name]

        galloc('out', zero=any(g is None for g in g_merged))
        gs = [g if (g is None or (g.dtype == torch.float32 and g.is_contiguous())) else g.float().contiguous()
              for g in g_merged]
        if all(g is not None for g in gs):   # every level in one launch
            _lib.call('sfvos_frames_to_pyramid', _planar_levels(gs), len(gs), _ptr(gb['out']), dt_id, B, 256, 256,
                      _stream())
        else:
            off = 0
            for (H, W), g in zip(shapes, gs):
                if g is not None:
                    _lib.call('sfvos_frames_to_ndhwc', _ptr(g), 256 * H * W, H * W, W, 1, _ptr(gb['out'], off * 256),
                              dt_id, B, 256, H, W, 256, _stream())
                off += B * H * W

        # -- phase 1 (current stream): decide what runs, allocate every buffer / workspace, pack dgrad images
        any_param = any(need_param.values())
        # optional gradient sink (FusedSGD.attach): parameter gradients are written (first backward after
        # zero_grad()) or accumulated (later ones) straight into its flat buffer by the kernels that produce
        # them, and autograd gets None for them -- no temporary + `grad += tmp` launch per parameter.
        sink = getattr(self, '_grad_sink', None)
        sink_mode = sink.begin_direct(dev) if sink is not None else 0   # 0 off, 1 overwrite, 2 accumulate
        sink = sink if sink_mode else None
        sacc = 1 if sink_mode == 2 else 0
        direct = set()

        def sink_view(pname):
            mod, attr = pname.split('.')
            return sink.view_of(getattr(self, mod)._parameters[attr]) if sink is not None else None

        def gout(pname, shape, ok=True):
            t = sink_view(pname) if ok else None
            if t is None:
                return torch.empty(shape, dtype=torch.float32, device=dev)
            direct.add(pname)
            return t
        todo = []
        written = set()
        keepalive = []
        for l in reversed(plan.layers):
            conv = getattr(self, l.conv)
            first = l.src in ('xs0', 'xf0')
            need_in = (need_slow if l.src == 'xs0' else need_fast) if first else True
            need_w = need_param[l.conv + '.weight']
            need_b = conv.bias is not None and need_param[l.conv + '.bias']
            need_bn = need_param[l.bn + '.weight'] or need_param[l.bn + '.bias']
            if first and not (need_in or need_w or need_b or need_bn):
                continue
            if not first and not (any_param or need_slow or need_fast):
                continue
            lv = _lib.make_levels(shapes, B, l.t_out)
            rows = lib.sfvos_bn_bwd_rows(ctypes.byref(lv))
            w = dict(l=l, lv=lv, rows=rows, need_in=need_in, need_w=need_w, need_b=need_b,
                     part=torch.empty((rows, 2, l.c_out), dtype=torch.float32, device=dev),
                     # dgamma and dbeta leave one kernel with one accumulate flag: both direct or neither
                     dgamma=gout(l.bn + '.weight', (l.c_out,), sink_view(l.bn + '.bias') is not None),
                     dbeta=gout(l.bn + '.bias', (l.c_out,), sink_view(l.bn + '.weight') is not None),
                     dx=torch.empty((_lv_total(lv), l.c_out), dtype=tdt, device=dev))
            if need_b:
                w['bpart'] = torch.empty((rows, l.c_out), dtype=torch.float32, device=dev)
                w['db'] = gout(l.conv + '.bias', (l.c_out,))
            sname, t_alloc, t_off = self._src_window(l, state.slow_offset, state.x_frames, state.x_pad)
            w['src'] = bufs[sname]
            if need_w:
                d = self._desc(l, B, pyr, dt_id, w['src'], l.c_out, t_alloc, t_off)
                nbytes = lib.sfvos_conv3d_wgrad_workspace_bytes(ctypes.byref(d))
                if nbytes == 0:
                    _lib.check(-1, 'sfvos_conv3d_wgrad_workspace_bytes')
                w['wd'] = d
                w['ws'] = torch.empty(nbytes, dtype=torch.uint8, device=dev)
                w['gw'] = gout(l.conv + '.weight', tuple(conv.weight.shape))
            if need_in:
                if l.src not in gb:
                    galloc(l.src)
                w['acc'] = 1 if l.src in written else 0
                written.add(l.src)
                w['dd'] = self._desc(l, B, pyr, dt_id, l.c_out, gb[l.src].shape[-1], dgrad=True, accumulate=w['acc'])
                w['wpd'] = self._packed(l, 'dgrad', dt_name)
            todo.append(w)
            keepalive.append(w)

        # -- phase 2: launches.  Cross-pathway hand-offs: the fast side's data-gradient writes g(y_f2)/g(y_f1)
        # before the lateral's accumulates into it, and the next fast layer waits for that accumulate.
        if side is not None:
            side.wait_stream(main)
        ev = {}
        wait_for = {'l2': 'f3', 'f2': 'l2', 'l1': 'f2', 'f1': 'l1'}
        grads = {}
        for w in todo:
            l = w['l']
            conv, bn = getattr(self, l.conv), getattr(self, l.bn)
            lv, rows = w['lv'], w['rows']
            cf = coef[l.name]
            cs = _CF_ROWS * l.c_out
            dy = gb[l.dst]
            raw = bufs[l.raw]
            dx = w['dx']
            stream = side if (side is not None and self._on_side(l)) else main
            with torch.cuda.stream(stream):
                st = _stream()
                dep = wait_for.get(l.name)
                if side is not None and dep in ev:
                    stream.wait_event(ev[dep])
                treg = self._t('bn_bwd', l.name)
                treg.__enter__()
                _lib.call('sfvos_bn_bwd_reduce', _ptr(dy, l.dst_off), dy.shape[-1], _ptr(raw), l.c_out, dt_id,
                          ctypes.byref(lv), l.c_out, _ptr(cf[0, _SCALE]), _ptr(cf[0, _SHIFT]), _ptr(cf[0, _MEAN]),
                          _ptr(cf[0, _RSTD]), cs, 1 if l.relu else 0, _ptr(w['part']), st)
                _lib.call('sfvos_bn_bwd_finalize', _ptr(w['part']), ctypes.byref(lv), _ptr(bn.weight.detach()),
                          _ptr(cf[0, _MEAN]), _ptr(cf[0, _RSTD]), cs, l.c_out, 1 if state.train else 0,
                          _ptr(cf[0, _CA]), _ptr(cf[0, _CB]), _ptr(cf[0, _CK]), _ptr(cf[0, _SDZ]), _ptr(cf[0, _SDZX]), st)
                grads[l.bn + '.weight'], grads[l.bn + '.bias'] = w['dgamma'], w['dbeta']
                _lib.call('sfvos_bn_bwd_apply', _ptr(dy, l.dst_off), dy.shape[-1], _ptr(raw), l.c_out, _ptr(dx),
                          l.c_out, dt_id, ctypes.byref(lv), l.c_out, _ptr(cf[0, _SCALE]), _ptr(cf[0, _SHIFT]), cs,
                          1 if l.relu else 0, _ptr(cf[0, _CA]), _ptr(cf[0, _CB]), _ptr(cf[0, _CK]),
                          _ptr(w['bpart']) if w['need_b'] else None, _ptr(cf[0, _SDZ]), _ptr(cf[0, _SDZX]),
                          _ptr(w['dgamma']), _ptr(w['dbeta']), sacc if (l.bn + '.weight') in direct else 0, st)
                treg.__exit__(None, None, None)
                if w['need_b']:
                    _lib.call('sfvos_reduce_rows', _ptr(w['bpart']), rows, l.c_out, _ptr(w['db']),
                              sacc if (l.conv + '.bias') in direct else 0, st)
                    grads[l.conv + '.bias'] = w['db']
                if w['need_in']:  # data gradient first: the other pathway may be waiting for it
                    with self._t('conv_dgrad', l.name):
                        _lib.call('sfvos_conv3d', ctypes.byref(w['dd']), _ptr(dx), _ptr(w['wpd']), None,
                                  _ptr(gb[l.src]), None, st)
                    if side is not None:
                        ev[l.name] = torch.cuda.Event()
                        ev[l.name].record(stream)
                if w['need_w']:
                    with self._t('wgrad', l.name):
                        _lib.call('sfvos_conv3d_wgrad', ctypes.byref(w['wd']), _ptr(w['src']), _ptr(dx), _ptr(w['gw']),
                                  sacc if (l.conv + '.weight') in direct else 0, _ptr(w['ws']), st)
                    grads[l.conv + '.weight'] = w['gw']
                if sink is not None:  # this layer's slice of the flat gradient is complete: exchange may start
                    done = [q for q in list(conv.parameters()) + list(bn.parameters())]
                    if all((n_ in direct) for n_ in ([l.conv + '.weight', l.bn + '.weight', l.bn + '.bias']
                                                     + ([l.conv + '.bias'] if conv.bias is not None else []))):
                        sink.layer_done(done, stream)
        if side is not None:
            main.wait_stream(side)
        del keepalive  # workspaces are released only after the join
        for n in direct:  # already in the optimiser's buffer: nothing for autograd to accumulate
            grads[n] = None
        if sink is not None:
            sink.end_direct()
        return grads, gb

    # ------------------------------------------------------------------ input layout
    def _to_pyramid(self, tensors, frames, dt_id, tdt):
        """list over levels of [B,C,T,H,W] fp32 (any strides) -> flat pyramid buffer (model.py:157-158)."""
        B, C = tensors[0].shape[0], tensors[0].shape[1]
        pix = sum(t.shape[3] * t.shape[4] for t in tensors)
        M = B * frames * pix
        # bf16: channel-group-major ([C/32][M][32], see PackedClip) -- the layout the first convs read fastest
        grouped = tdt == torch.bfloat16 and C % 32 == 0 and os.environ.get('SFVOS_INPUT_LAYOUT', 'grouped') != 'ndhwc'
        flat = torch.empty((C // 32, M, 32) if grouped else (M, C), dtype=tdt, device=tensors[0].device)
        st = _stream()
        off = 0
        for t in tensors:
            s = t if t.dtype == torch.float32 else t.float()
            H, W = s.shape[3], s.shape[4]
            for b in range(B):
                pos = off + b * frames * H * W
                if grouped:
                    _lib.call('sfvos_frames_to_groups', _ptr(s[b]), s.stride(2), s.stride(1), s.stride(3), s.stride(4),
                              _ptr(flat, pos * 32), dt_id, frames, C, H, W, M * 32, st)
                else:
                    _lib.call('sfvos_frames_to_ndhwc', _ptr(s[b]), s.stride(2), s.stride(1), s.stride(3), s.stride(4),
                              _ptr(flat, pos * C), dt_id, frames, C, H, W, C, st)
            off += B * frames * H * W
        return flat

    def _to_pyramid_fp8(self, tensors, frames):
        """list over levels of [B,C,T,H,W] fp32 -> e4m3 clip [C/64][M][64] (sfvos_frames_to_groups_fp8)."""
        B, C = tensors[0].shape[0], tensors[0].shape[1]
        if C % 64 != 0:
            raise RuntimeError("precision='fp8' needs an input channel count that is a multiple of 64")
        pix = sum(t.shape[3] * t.shape[4] for t in tensors)
        M = B * frames * pix
        flat = torch.empty((C // 64, M, 64), dtype=torch.uint8, device=tensors[0].device)
        if self._fp8_sat is None or self._fp8_sat.device != flat.device:
            self._fp8_sat = torch.zeros(1, dtype=torch.int32, device=flat.device)
        st = _stream()
        off = 0
        for t in tensors:
            s = t if t.dtype == torch.float32 else t.float()
            H, W = s.shape[3], s.shape[4]
            for b in range(B):
                _lib.call('sfvos_frames_to_groups_fp8', _ptr(s[b]), s.stride(2), s.stride(1), s.stride(3), s.stride(4),
                          _ptr(flat, (off + b * frames * H * W) * 64), frames, C, H, W, M * 64,
                          float(self.fp8_input_scale), _ptr(self._fp8_sat), st)
            off += B * frames * H * W
        return flat

    def _slow_alias_offset(self, slow_list, fast_list):
        """k if every slow[l] is exactly frames [k, k+sp) of fast[l] (same storage and strides -- what
        SegmentationModel passes, model.py:336-338), else None."""
        k_all = None
        for s, f in zip(slow_list, fast_list):
            if s.dtype != f.dtype or s.stride() != f.stride() or s.shape[:2] != f.shape[:2] \
                    or s.shape[3:] != f.shape[3:]:
                return None
            if s.untyped_storage().data_ptr() != f.untyped_storage().data_ptr() or f.stride(2) == 0:
                return None
            delta = s.storage_offset() - f.storage_offset()
            if delta < 0 or delta % f.stride(2) != 0:
                return None
            k = delta // f.stride(2)
            if k + s.shape[2] > f.shape[2] or (k_all is not None and k != k_all):
                return None
            k_all = k
        return k_all

    # ------------------------------------------------------------------ reference API
    def _run(self, slow_list, fast_list):
        """lists (one entry per FPN level) of [B,C,T,H,W] tensors -> list of merged [B,256,H,W]."""
        self._check_ready(fast_list[0])
        plan = self.plan
        f0, s0 = fast_list[0], slow_list[0]
        if f0.shape[1] != plan.input_size or s0.shape[2] != plan.sp or f0.shape[2] != plan.fp:
            raise RuntimeError('expected %d channels and %d/%d slow/fast frames, got C=%d, %d/%d'
                               % (plan.input_size, plan.sp, plan.fp, f0.shape[1], s0.shape[2], f0.shape[2]))
        for s, f in zip(slow_list, fast_list):
            if s.shape[0] != f.shape[0] or s.shape[3:] != f.shape[3:] or f.shape[0] != f0.shape[0]:
                raise RuntimeError('slow and fast inputs disagree in batch or spatial size')
        if len(fast_list) > _lib.MAX_LEVELS:
            raise RuntimeError('at most %d pyramid levels per call, got %d' % (_lib.MAX_LEVELS, len(fast_list)))
        names = [n for n, _ in self.named_parameters()]
        params = [p for _, p in self.named_parameters()]
        keep = torch.is_grad_enabled() and (any(t.requires_grad for t in list(slow_list) + list(fast_list))
                                            or any(p.requires_grad for p in params))
        meta = dict(mode='frames', L=len(fast_list), keep=keep)
        return _SlowFastPyramidFn.apply(self, meta, names, *(list(slow_list) + list(fast_list) + params))

    def forward(self, slow, fast):
        """(slow [B,C,Ts,H,W], fast [B,C,Tf,H,W]) -> (slow [B,224,1,H,W], fast [B,32,1,H,W]) -- model.py:118-149."""
        merged = self._run([slow], [fast])[0]
        return merged[:, :224].unsqueeze(2), merged[:, 224:].unsqueeze(2)

    def temporally_enhance_features(self, slow_features, fast_features):
        """list(len B) of OrderedDict level -> [T,C,H,W]  ->  OrderedDict level -> [B,256,H,W] (model.py:151-165)."""
        slow_features = {k: [dic[k] for dic in slow_features] for k in slow_features[0]}
        fast_features = {k: [dic[k] for dic in fast_features] for k in fast_features[0]}
        keys = list(slow_features.keys())
        # model.py:157-158 stacks the B per-clip tensors; the reference always passes B = 1, where the stack is a pure
        # copy of the whole clip (2.8 GB read + 2.8 GB written at fp=32): a view of the caller's tensor does the same
        def batch(ts):
            return (ts[0].to(self.device).unsqueeze(0) if len(ts) == 1 else torch.stack(ts).to(self.device)).transpose(1, 2)
        slow_list = [batch(slow_features[k]) for k in keys]
        fast_list = [batch(fast_features[k]) for k in keys]
        merged = self._run(slow_list, fast_list)
        return OrderedDict(zip(keys, merged))

    def enhance_packed(self, clip, slow_offset=None):
        """PackedClip of the FAST window (channels-last, compute dtype) -> OrderedDict level -> [B,256,H,W].
        The slow pathway reads frames [slow_offset, slow_offset+sp) of the same clip; default = the
        centre frames, as SegmentationModel._slice_features takes them (model.py:242-248,322,337)."""
        self._check_ready(clip.data)
        plan = self.plan
        _, tdt = _DT[self.precision]
        if self.precision == 'fp8':   # the e4m3 clip of pack_fp8 (64-channel groups, quantised with fp8_input_scale)
            tdt = torch.uint8
            if clip.layout != 'grouped8' or clip.pad != (0, 0):
                raise RuntimeError("precision='fp8' takes the e4m3 clip SlowFastLayers.pack_fp8 builds (layout "
                                   "'grouped8', whole window stored), got layout %r" % clip.layout)
        if clip.window != plan.fp or clip.channels != plan.input_size or clip.data.dtype != tdt \
                or not clip.data.is_contiguous():
            raise RuntimeError('PackedClip must stand for %d frames (stored + zero padding) x %d channels, contiguous %s'
                               % (plan.fp, plan.input_size, tdt))
        if len(clip.shapes) > _lib.MAX_LEVELS:
            raise RuntimeError('at most %d pyramid levels per call' % _lib.MAX_LEVELS)
        if slow_offset is None:
            slow_offset = plan.fp // 2 - plan.sp // 2
        if slow_offset < 0 or slow_offset + plan.sp > plan.fp:
            raise RuntimeError('slow window [%d, %d) outside the fast clip' % (slow_offset, slow_offset + plan.sp))
        names = [n for n, _ in self.named_parameters()]
        params = [p for _, p in self.named_parameters()]
        keep = torch.is_grad_enabled() and (clip.data.requires_grad or any(p.requires_grad for p in params))
        meta = dict(mode='packed', L=len(clip.shapes), keep=keep, shapes=clip.shapes, B=clip.batch,
                    slow_offset=slow_offset, x_frames=clip.frames, x_pad=clip.pad[0])
        merged = _SlowFastPyramidFn.apply(self, meta, names, clip.data, *params)
        return OrderedDict(zip(clip.keys, merged))


class _SlowFastPyramidFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, meta, names, *tensors):
        L, keep = meta['L'], meta['keep']
        dt_id, tdt = _DT[module.precision]
        if meta['mode'] == 'frames':
            slow_list = [t.detach() for t in tensors[:L]]
            fast_list = [t.detach() for t in tensors[L:2 * L]]
            n_in = 2 * L
            B = fast_list[0].shape[0]
            shapes = [tuple(f.shape[3:]) for f in fast_list]
            slow_offset = module._slow_alias_offset(slow_list, fast_list)
            if module.precision == 'fp8':
                # the fast clip as e4m3 64-channel groups, read by fast_conv1 and (its centre frames) by slow_conv1
                xf0 = module._to_pyramid_fp8(fast_list, module.plan.fp)
                xs0 = None if slow_offset is not None else module._to_pyramid_fp8(slow_list, module.plan.sp)
            else:
                xf0 = module._to_pyramid(fast_list, module.plan.fp, dt_id, tdt)
                xs0 = None if slow_offset is not None else module._to_pyramid(slow_list, module.plan.sp, dt_id, tdt)
            ctx.in_meta = [(t.shape, t.dtype) for t in tensors[:n_in]]
        else:
            n_in = 1
            B, shapes, slow_offset = meta['B'], meta['shapes'], meta['slow_offset']
            xf0, xs0 = tensors[0].detach(), None
        x_frames, x_pad = meta.get('x_frames'), meta.get('x_pad', 0)
        if x_frames == module.plan.fp and x_pad == 0:
            x_frames = None   # the whole window is stored
        merged, state = module._engine_forward(shapes, B, xf0, xs0, slow_offset, keep, x_frames, x_pad)
        ctx.module, ctx.state, ctx.names, ctx.mode, ctx.n_in, ctx.L = module, state, names, meta['mode'], n_in, L
        return tuple(merged)

    @staticmethod
    def backward(ctx, *g_merged):
        module, state, names, L = ctx.module, ctx.state, ctx.names, ctx.L
        if state is None:
            raise RuntimeError('backward through a forward that ran without grad state')
        nig = ctx.needs_input_grad
        base = 3
        if ctx.mode == 'frames':
            need_slow = any(nig[base + i] for i in range(L))
            need_fast = any(nig[base + L + i] for i in range(L))
        else:
            need_slow = need_fast = bool(nig[base])
        need_param = {n: bool(nig[base + ctx.n_in + i]) for i, n in enumerate(names)}
        grads, gb = module._engine_backward(state, list(g_merged), need_slow, need_fast, need_param)
        ctx.state = None
        dt_id, _ = _DT[state.dtype_name]
        st = _stream()
        B, shapes = state.B, state.shapes
        out = [None, None, None]
        if ctx.mode == 'frames':
            g_in = [None] * (2 * L)
            for which, key, frames in ((0, 'xs0', module.plan.sp), (1, 'xf0', module.plan.fp)):
                if not (need_slow, need_fast)[which]:
                    continue
                gbuf = gb[key]
                C = gbuf.shape[-1]
                off = 0
                for i, (H, W) in enumerate(shapes):
                    shape, dtype = ctx.in_meta[which * L + i]
                    if nig[base + which * L + i]:
                        gi = torch.empty(shape, dtype=torch.float32, device=gbuf.device)
                        for b in range(B):
                            _lib.call('sfvos_ndhwc_to_frames', _ptr(gbuf, (off + b * frames * H * W) * C), dt_id,
                                      _ptr(gi[b]), gi.stride(2), gi.stride(1), gi.stride(3), gi.stride(4), frames, C,
                                      H, W, C, 0, st)
                        g_in[which * L + i] = gi.to(dtype)
                    off += B * frames * H * W
            out.extend(g_in)
        else:
            g = None
            if need_fast:
                g = gb['xf0']   # data gradients are pyramid NDHWC [M, C]; this buffer is not used again
                gs = gb['xs0']  # slow window gradient: added into frames [so, so+sp) of the fast clip's gradient
                sp, fp, so = module.plan.sp, module.plan.fp, state.slow_offset
                C = g.shape[-1]
                off_f = off_s = 0
                for (H, W) in shapes:
                    P = H * W
                    for b in range(B):
                        _lib.call('sfvos_add_inplace', _ptr(g, (off_f + (b * fp + so) * P) * C),
                                  _ptr(gs, (off_s + b * sp * P) * C), dt_id, sp * P * C, st)
                    off_f += B * fp * P
                    off_s += B * sp * P
                if state.x_frames is not None:   # the clip stores only part of the window: hand back those frames
                    parts, off_f = [], 0
                    for (H, W) in shapes:
                        P = H * W
                        v = g[off_f: off_f + B * fp * P].view(B, fp, P, C)
                        parts.append(v[:, state.x_pad: state.x_pad + state.x_frames].reshape(-1, C))
                        off_f += B * fp * P
                    g = torch.cat(parts, 0)
                if state.bufs['xf0'].dim() == 3:  # the clip came channel-group-major: hand its gradient back that way
                    g = g.view(g.shape[0], -1, 32).permute(1, 0, 2).contiguous()
            out.append(g)
        for n in names:
            out.append(grads.get(n) if need_param[n] else None)
        return tuple(out)
