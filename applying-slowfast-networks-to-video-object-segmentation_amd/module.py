"""SlowFastLayers -- drop-in for the reference class (code/helpers/model.py:30-165) whose
forward and backward run in the hand-written HIP kernels of libsfvos.so.

Same constructor, `forward`, `temporally_enhance_features`, parameter names, state-dict keys
and train/eval semantics as the reference; consumed the same way by SegmentationModel
(model.py:184,340), OsvosSegmentationModel (osvos/osvos_model.py:28,65), train.py:74-80.

PyTorch is used for device memory, streams and autograd bookkeeping only: every arithmetic
step of the path is a libsfvos call (see include/sfvos.h).  There is no CPU / eager fallback."""
import ctypes
import os
from collections import OrderedDict

import torch
from torch import nn

from . import _lib
from .plan import SlowFastPlan

_DT = {'fp32': (_lib.F32, torch.float32), 'bf16': (_lib.BF16, torch.bfloat16)}


def _ptr(t, elem_offset=0):
    return ctypes.c_void_p(t.data_ptr() + elem_offset * t.element_size())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


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


class _LevelState(object):
    """What one forward of one pyramid level leaves behind for its backward."""
    __slots__ = ('B', 'H', 'W', 'bufs', 'coef', 'train', 'dtype_name')


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
            raise ValueError("precision must be 'fp32' or 'bf16', got %r" % (self.precision,))
        self._packs = {}     # (conv name, kind, dtype) -> ((param version, data_ptr, epoch), packed tensor)
        self._zeros = None
        self._timer = None

    def enable_kernel_timer(self):
        self._timer = KernelTimer()
        return self._timer

    def _t(self, kind, layer, H, W):
        if self._timer is None:
            return _NULL
        return self._timer.region('%s/%s/%dx%d' % (kind, layer, H, W))

    # ------------------------------------------------------------------ helpers
    def _check_ready(self, ref):
        if not ref.is_cuda:
            raise RuntimeError('sfvos_amd.SlowFastLayers runs only on an MI355X (gfx950) through libsfvos.so; '
                               'got a %s tensor. There is no CPU fallback.' % ref.device)
        _lib.load()
        w = self.fast_conv1.weight
        if w.device != ref.device:
            raise RuntimeError('module parameters are on %s but inputs on %s' % (w.device, ref.device))

    def _zero_page(self, device):
        if self._zeros is None or self._zeros.device != device:
            self._zeros = torch.zeros(1024, dtype=torch.uint8, device=device)
        return self._zeros

    def _packed(self, layer, kind, dt_name):
        dt_id, tdt = _DT[dt_name]
        w = getattr(self, layer.conv).weight
        key = (layer.conv, kind, dt_name)
        hit = self._packs.get(key)
        tag = (w._version, w.data_ptr(), _lib.weight_epoch())
        if hit is not None and hit[0] == tag:
            return hit[1]
        packed = torch.empty(w.numel(), dtype=tdt, device=w.device)
        fn = 'sfvos_pack_weights_fwd' if kind == 'fwd' else 'sfvos_pack_weights_dgrad'
        wc = w.detach()
        if wc.dtype != torch.float32 or not wc.is_contiguous():
            wc = wc.float().contiguous()
        _lib.call(fn, _ptr(wc), _ptr(packed), dt_id, layer.c_out, layer.c_in, layer.kt, layer.taps, _stream())
        self._packs[key] = (tag, packed)
        return packed

    def _desc(self, layer, B, H, W, dt_id, ld_x, ld_y, dgrad=False, accumulate=0):
        d = _lib.ConvDesc()
        d.dtype, d.batch, d.h, d.w, d.kt, d.taps = dt_id, B, H, W, layer.kt, layer.taps
        if dgrad:  # conv over dy producing dx: channels swapped, full temporal padding
            d.t_in, d.c_in, d.c_out, d.pad_t = layer.t_out, layer.c_out, layer.c_in, layer.kt - 1
            d.x_batch_stride = layer.t_out * H * W * ld_x
            d.y_batch_stride = layer.t_in * H * W * ld_y
        else:
            d.t_in, d.c_in, d.c_out, d.pad_t = layer.t_in, layer.c_in, layer.c_out, 0
            d.x_batch_stride = layer.t_in * H * W * ld_x
            d.y_batch_stride = layer.t_out * H * W * ld_y
        d.ld_x, d.ld_y, d.accumulate = ld_x, ld_y, accumulate
        return d

    # ------------------------------------------------------------------ forward engine (one level)
    def _engine_forward(self, slow, fast, ndhwc_input, keep):
        """slow/fast: [B,C,T,H,W] fp32 (any strides) or, with ndhwc_input, [B,T,H,W,C] in the
        compute dtype.  Returns (merged [B,256,H,W] fp32, state or None)."""
        plan = self.plan
        dt_name = self.precision
        dt_id, tdt = _DT[dt_name]
        dev = fast.device
        st = _stream()
        zeros = self._zero_page(dev)
        if ndhwc_input:
            B, Tf, H, W, C = fast.shape
            Ts = slow.shape[1]
        else:
            B, C, Tf, H, W = fast.shape
            Ts = slow.shape[2]
        if C != plan.input_size or Ts != plan.sp or Tf != plan.fp:
            raise RuntimeError('expected %d channels and %d/%d slow/fast frames, got C=%d, %d/%d'
                               % (plan.input_size, plan.sp, plan.fp, C, Ts, Tf))
        if slow.shape[0] != B or tuple(slow.shape[-2:] if not ndhwc_input else slow.shape[2:4]) != (H, W):
            raise RuntimeError('slow and fast inputs disagree in batch or spatial size')
        bufs = {}

        def alloc(name):
            b = plan.buffers[name]
            bufs[name] = torch.empty((B, b.frames, H, W, b.channels), dtype=tdt, device=dev)
            return bufs[name]

        # -- layout: frames (NCHW stacks viewed as NCDHW, model.py:157-158) -> NDHWC
        for name, src in (('xs0', slow), ('xf0', fast)):
            if ndhwc_input:
                if src.dtype != tdt or not src.is_contiguous():
                    raise RuntimeError('NDHWC inputs must be contiguous %s' % tdt)
                bufs[name] = src
            else:
                s = src if src.dtype == torch.float32 else src.float()
                dst = alloc(name)
                T = s.shape[2]
                for b in range(B):
                    _lib.call('sfvos_frames_to_ndhwc', _ptr(s[b]), s.stride(2), s.stride(1), s.stride(3), s.stride(4),
                              _ptr(dst[b]), dt_id, T, C, H, W, C, st)

        train = self.training
        coef = {}
        for l in plan.layers:
            conv, bn = getattr(self, l.conv), getattr(self, l.bn)
            src = bufs[l.src]
            raw = alloc(l.raw)
            if l.dst not in bufs:
                alloc(l.dst)
            dst = bufs[l.dst]
            d = self._desc(l, B, H, W, dt_id, src.shape[-1], l.c_out)
            M = B * l.t_out * H * W
            wp = self._packed(l, 'fwd', dt_name)
            bias = _ptr(conv.bias.detach()) if conv.bias is not None else None
            cf = torch.empty((6, l.c_out), dtype=torch.float32, device=dev)  # mean rstd scale shift var_unb spare
            if train:
                if M <= 1:
                    raise ValueError('Expected more than 1 value per channel when training, got input size %s'
                                     % str([B, l.c_out, l.t_out, H, W]))
                rows = _lib.load().sfvos_conv3d_stat_rows(ctypes.byref(d))
                if rows <= 0:
                    _lib.check(rows if rows < 0 else -1, 'sfvos_conv3d_stat_rows')
                part = torch.empty((rows, 2, l.c_out), dtype=torch.float32, device=dev)
                with self._t('conv_fwd', l.name, H, W):
                    _lib.call('sfvos_conv3d', ctypes.byref(d), _ptr(src), _ptr(wp), bias, _ptr(raw), _ptr(part),
                              _ptr(zeros), st)
                _lib.call('sfvos_bn_finalize', _ptr(part), rows, M, _ptr(bn.weight.detach()), _ptr(bn.bias.detach()),
                          float(bn.eps), l.c_out, _ptr(cf[0]), _ptr(cf[1]), _ptr(cf[2]), _ptr(cf[3]), _ptr(cf[4]), st)
                if bn.track_running_stats and bn.running_mean is not None:
                    bn.num_batches_tracked.add_(1)
                    if bn.momentum is None:
                        raise RuntimeError('BatchNorm momentum=None (cumulative average) is not supported')
                    _lib.call('sfvos_bn_running_update', _ptr(bn.running_mean), _ptr(bn.running_var), _ptr(cf[0]),
                              _ptr(cf[4]), 1, l.c_out, float(bn.momentum), st)
            else:
                with self._t('conv_fwd', l.name, H, W):
                    _lib.call('sfvos_conv3d', ctypes.byref(d), _ptr(src), _ptr(wp), bias, _ptr(raw), None, _ptr(zeros),
                              st)
                _lib.call('sfvos_bn_eval_coeffs', _ptr(bn.weight.detach()), _ptr(bn.bias.detach()),
                          _ptr(bn.running_mean), _ptr(bn.running_var), float(bn.eps), l.c_out, _ptr(cf[2]),
                          _ptr(cf[3]), st)
                if keep:  # eval-mode backward needs (running_mean, rstd of running_var)
                    cf[0].copy_(bn.running_mean)
                    cf[1].copy_(torch.rsqrt(bn.running_var + bn.eps))
            with self._t('bn_apply', l.name, H, W):
                _lib.call('sfvos_bn_apply', _ptr(raw), l.c_out, _ptr(dst, l.dst_off), dst.shape[-1], dt_id, M, l.c_out,
                          _ptr(cf[2]), _ptr(cf[3]), 1 if l.relu else 0, st)
            coef[l.name] = cf

        # -- cat([slow224, fast32], 1).squeeze(2) (model.py:162) as the caller's NCHW fp32 tensor
        merged = torch.empty((B, 256, H, W), dtype=torch.float32, device=dev)
        _lib.call('sfvos_ndhwc_to_frames', _ptr(bufs['out']), dt_id, _ptr(merged), 256 * H * W, H * W, W, 1, B, 256,
                  H, W, 256, 0, st)
        if not keep:
            return merged, None
        state = _LevelState()
        state.B, state.H, state.W, state.bufs, state.coef = B, H, W, bufs, coef
        state.train, state.dtype_name = train, dt_name
        del bufs['out']
        return merged, state

    # ------------------------------------------------------------------ backward engine (one level)
    def _engine_backward(self, state, g_merged, need_slow, need_fast, need_param):
        plan = self.plan
        dt_name = state.dtype_name
        dt_id, tdt = _DT[dt_name]
        B, H, W, bufs, coef = state.B, state.H, state.W, state.bufs, state.coef
        dev = g_merged.device
        st = _stream()
        zeros = self._zero_page(dev)
        lib = _lib.load()
        g = g_merged if (g_merged.dtype == torch.float32 and g_merged.is_contiguous()) else g_merged.float().contiguous()

        gb = {}  # gradient buffers w.r.t. activation buffers

        def galloc(name):
            b = plan.buffers[name]
            gb[name] = torch.empty((B, b.frames, H, W, b.channels), dtype=tdt, device=dev)
            return gb[name]

        galloc('out')
        _lib.call('sfvos_frames_to_ndhwc', _ptr(g), 256 * H * W, H * W, W, 1, _ptr(gb['out']), dt_id, B, 256, H, W,
                  256, st)
        grads = {}
        written = set()
        any_param = any(need_param.values())
        for l in reversed(plan.layers):
            conv, bn = getattr(self, l.conv), getattr(self, l.bn)
            first = l.src in ('xs0', 'xf0')
            need_in = (need_slow if l.src == 'xs0' else need_fast) if first else True
            need_w = need_param[l.conv + '.weight']
            need_b = conv.bias is not None and need_param[l.conv + '.bias']
            need_bn = need_param[l.bn + '.weight'] or need_param[l.bn + '.bias']
            if first and not (need_in or need_w or need_b or need_bn):
                continue
            if not first and not (any_param or need_slow or need_fast):
                continue
            M = B * l.t_out * H * W
            cf = coef[l.name]
            dy = gb[l.dst]
            raw = bufs[l.raw]
            rows = lib.sfvos_bn_bwd_rows(M)
            part = torch.empty((rows, 2, l.c_out), dtype=torch.float32, device=dev)
            treg = self._t('bn_bwd', l.name, H, W)
            treg.__enter__()
            _lib.call('sfvos_bn_bwd_reduce', _ptr(dy, l.dst_off), dy.shape[-1], _ptr(raw), l.c_out, dt_id, M, l.c_out,
                      _ptr(cf[2]), _ptr(cf[3]), _ptr(cf[0]), _ptr(cf[1]), 1 if l.relu else 0, _ptr(part), st)
            dgamma = torch.empty(l.c_out, dtype=torch.float32, device=dev)
            dbeta = torch.empty(l.c_out, dtype=torch.float32, device=dev)
            abk = torch.empty((3, l.c_out), dtype=torch.float32, device=dev)
            _lib.call('sfvos_bn_bwd_finalize', _ptr(part), rows, M, _ptr(bn.weight.detach()), _ptr(cf[0]), _ptr(cf[1]),
                      l.c_out, 1 if state.train else 0, 0, _ptr(dgamma), _ptr(dbeta), _ptr(abk[0]), _ptr(abk[1]),
                      _ptr(abk[2]), st)
            grads[l.bn + '.weight'], grads[l.bn + '.bias'] = dgamma, dbeta
            dx = torch.empty((B, l.t_out, H, W, l.c_out), dtype=tdt, device=dev)
            bpart = torch.empty((rows, l.c_out), dtype=torch.float32, device=dev) if need_b else None
            _lib.call('sfvos_bn_bwd_apply', _ptr(dy, l.dst_off), dy.shape[-1], _ptr(raw), l.c_out, _ptr(dx), l.c_out,
                      dt_id, M, l.c_out, _ptr(cf[2]), _ptr(cf[3]), 1 if l.relu else 0, _ptr(abk[0]), _ptr(abk[1]),
                      _ptr(abk[2]), _ptr(bpart) if need_b else None, st)
            treg.__exit__(None, None, None)
            if need_b:
                db = torch.empty(l.c_out, dtype=torch.float32, device=dev)
                _lib.call('sfvos_reduce_rows', _ptr(bpart), rows, l.c_out, _ptr(db), 0, st)
                grads[l.conv + '.bias'] = db
            src = bufs[l.src]
            if need_w:
                d = self._desc(l, B, H, W, dt_id, src.shape[-1], l.c_out)
                nbytes = lib.sfvos_conv3d_wgrad_workspace_bytes(ctypes.byref(d))
                if nbytes == 0:
                    _lib.check(-1, 'sfvos_conv3d_wgrad_workspace_bytes')
                ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
                gw = torch.empty(conv.weight.shape, dtype=torch.float32, device=dev)
                with self._t('wgrad', l.name, H, W):
                    _lib.call('sfvos_conv3d_wgrad', ctypes.byref(d), _ptr(src), _ptr(dx), _ptr(gw), 0, _ptr(ws),
                              _ptr(zeros), st)
                grads[l.conv + '.weight'] = gw
            if need_in:
                if l.src not in gb:
                    galloc(l.src)
                gsrc = gb[l.src]
                acc = 1 if l.src in written else 0
                d = self._desc(l, B, H, W, dt_id, l.c_out, gsrc.shape[-1], dgrad=True, accumulate=acc)
                wp = self._packed(l, 'dgrad', dt_name)
                with self._t('conv_dgrad', l.name, H, W):
                    _lib.call('sfvos_conv3d', ctypes.byref(d), _ptr(dx), _ptr(wp), None, _ptr(gsrc), None, _ptr(zeros),
                              st)
                written.add(l.src)
            del dx
        return grads, gb

    # ------------------------------------------------------------------ reference API
    def _run_level(self, slow, fast, ndhwc_input=False):
        self._check_ready(fast)
        names = [n for n, _ in self.named_parameters()]
        params = [p for _, p in self.named_parameters()]
        # decided here: inside Function.forward grad mode is always off
        keep = torch.is_grad_enabled() and (slow.requires_grad or fast.requires_grad
                                            or any(p.requires_grad for p in params))
        return _SlowFastLevelFn.apply(self, (ndhwc_input, keep), names, slow, fast, *params)

    def forward(self, slow, fast):
        """(slow [B,C,Ts,H,W], fast [B,C,Tf,H,W]) -> (slow [B,224,1,H,W], fast [B,32,1,H,W]) -- model.py:118-149."""
        merged = self._run_level(slow, fast)
        return merged[:, :224].unsqueeze(2), merged[:, 224:].unsqueeze(2)

    def temporally_enhance_features(self, slow_features, fast_features):
        """list(len B) of OrderedDict level -> [T,C,H,W]  ->  OrderedDict level -> [B,256,H,W] (model.py:151-165)."""
        slow_features = {k: [dic[k] for dic in slow_features] for k in slow_features[0]}
        fast_features = {k: [dic[k] for dic in fast_features] for k in fast_features[0]}
        merged_features = OrderedDict()
        for key in slow_features.keys():
            s = torch.stack(slow_features[key]).to(self.device).transpose(1, 2)
            f = torch.stack(fast_features[key]).to(self.device).transpose(1, 2)
            merged_features[key] = self._run_level(s, f)
        return merged_features

    def temporally_enhance_features_ndhwc(self, slow_features, fast_features):
        """Same as temporally_enhance_features for producers that already hold channels-last clips:
        OrderedDict level -> [B,T,H,W,C] tensors in the compute dtype; skips the
        stack/transpose layout pass."""
        merged = OrderedDict()
        for key in slow_features.keys():
            merged[key] = self._run_level(slow_features[key], fast_features[key], ndhwc_input=True)
        return merged


class _SlowFastLevelFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, flags, names, slow, fast, *params):
        ndhwc_input, keep = flags
        merged, state = module._engine_forward(slow.detach(), fast.detach(), ndhwc_input, keep)
        ctx.module, ctx.state, ctx.names, ctx.ndhwc_input = module, state, names, ndhwc_input
        ctx.in_meta = (slow.shape, slow.dtype, fast.shape, fast.dtype)
        return merged

    @staticmethod
    def backward(ctx, g_merged):
        module, state, names = ctx.module, ctx.state, ctx.names
        if state is None:
            raise RuntimeError('backward through a forward that ran without grad state')
        need_slow, need_fast = ctx.needs_input_grad[3], ctx.needs_input_grad[4]
        need_param = {n: bool(ctx.needs_input_grad[5 + i]) for i, n in enumerate(names)}
        grads, gb = module._engine_backward(state, g_merged, need_slow, need_fast, need_param)
        ctx.state = None
        g_slow = g_fast = None
        dt_id, _ = _DT[state.dtype_name]
        st = _stream()
        for which, key in ((0, 'xs0'), (1, 'xf0')):
            if not (need_slow, need_fast)[which]:
                continue
            shape, dtype = ctx.in_meta[2 * which], ctx.in_meta[2 * which + 1]
            gbuf = gb[key]
            if ctx.ndhwc_input:
                gi = gbuf.to(dtype)
            else:
                B, C, T, H, W = shape
                gi = torch.empty(shape, dtype=torch.float32, device=gbuf.device)
                for b in range(B):
                    _lib.call('sfvos_ndhwc_to_frames', _ptr(gbuf[b]), dt_id, _ptr(gi[b]), gi.stride(2), gi.stride(1),
                              gi.stride(3), gi.stride(4), T, C, H, W, C, 0, st)
                gi = gi.to(dtype)
            if which == 0:
                g_slow = gi
            else:
                g_fast = gi
        out = [None, None, None, g_slow, g_fast]
        for i, n in enumerate(names):
            out.append(grads.get(n) if need_param[n] else None)
        return tuple(out)
