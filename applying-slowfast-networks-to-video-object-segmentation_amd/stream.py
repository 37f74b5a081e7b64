"""Sequence-mode inference with sliding-window reuse.

The reference evaluates a video frame by frame (code/helpers/model.py:316-340): for centre frame i it runs
`temporally_enhance_features` on the window of fp frames around i (zero frames beyond the ends of the sequence,
model.py:215-225), so consecutive centre frames share fp-1 input frames and every layer recomputes activations it
already produced for the previous centre frame.  In eval mode every layer of the path is shift-invariant in time
(valid temporal convs, BatchNorm with running statistics = a per-channel affine map, ReLU), so every activation can be
indexed on the absolute timeline of the video:

    F1[a] = f1(X[a .. a+kf1-1])     F2[a] = f2(F1[a .. a+kf2-1])       F3[a] = f3(F2[a .. a+kf3-1])
    C1[a] = cat(s1(X[a+so .. a+so+ks1-1]), l1(F1[a .. a+kl1-1]))        (so = fp//2 - sp//2)
    C2[a] = cat(s2(C1[a .. a+ks2-1]), l2(F2[a .. a+kl2-1]))             out[s] = cat(s3(C2[s .. s+ks3-1]), F3[s])

and the window starting at frame s (centre s + fp//2) needs exactly out[s].  When input frame n arrives, ONE new
frame of each of these streams becomes computable; `SlowFastStream.push` computes exactly those (8 convs with one
output frame each + their BatchNorm/ReLU) instead of the whole window: fast_conv1 does 1/T1f of its per-window work,
fast_conv2 1/T2f, the laterals and slow convs 1/T1s, 1/T2s.  The kernels are the ones the module uses; their inputs are
frame-major rings (`sfvos_conv_desc.x_frame_stride / y_frame_stride`): frame index a of a stream lives in slots a % T'
and a % T' + T' of a 2T'-slot ring, so every window of the last T' frames is contiguous.  When the frames of a video
are available in advance they are pushed `chunk` at a time (`push_many`): every layer then computes `chunk` output
frames per launch (T' = T + chunk - 1), which gives the kernels their multi-frame tiles and divides the launch count
per frame by `chunk`.  Outputs equal the module's eval-mode `temporally_enhance_features` on the same window
(tests/test_gpu_stream.py)."""
import ctypes
from collections import OrderedDict

import torch

from . import _lib
from .module import _CF_ROWS, _DT, _MEAN, _RSTD, _SCALE, _SHIFT, _ptr, _stream


class _Ring(object):
    """2T' frame slots of one whole-pyramid frame each ([positions per frame, channels]; the input ring of the bf16
    path is channel-group-major: [C/32][slots*positions][32])."""

    def __init__(self, period, positions, channels, dtype, device, grouped=False):
        self.T, self.FS, self.C, self.grouped = period, positions, channels, grouped
        shape = (channels // 32, 2 * period * positions, 32) if grouped else (2 * period * positions, channels)
        self.buf = torch.zeros(shape, dtype=dtype, device=device)

    def runs(self, first, count):
        """Slot runs that hold frames first .. first+count-1: [(slot, frame offset, length)] for both copies."""
        out = []
        s0 = first % self.T
        n1 = min(count, self.T - s0)
        for base in (0, self.T):
            out.append((base + s0, 0, n1))
            if n1 < count:
                out.append((base, n1, count - n1))
        return out


class SlowFastStream(object):
    def __init__(self, module, shapes, keys=None, chunk=1):
        """module: a sfvos_amd.SlowFastLayers on the GPU (used in eval mode); shapes: [(H, W)] of the FPN levels;
        chunk: most frames one push_many() call takes (1 = frame-by-frame, lowest latency)."""
        self.m = module
        self.plan = plan = module.plan
        self.shapes = [tuple(s) for s in shapes]
        self.keys = list(keys) if keys is not None else [str(i) for i in range(len(self.shapes))]
        if len(self.shapes) > _lib.MAX_LEVELS:
            raise RuntimeError('at most %d pyramid levels' % _lib.MAX_LEVELS)
        if chunk < 1:
            raise ValueError('chunk must be >= 1')
        w = module.fast_conv1.weight
        if not w.is_cuda:
            raise RuntimeError('SlowFastStream runs on the GPU through libsfvos.so (no CPU fallback)')
        _lib.load()
        self.chunk = G = int(chunk)
        self.dev = dev = w.device
        self.dt_name = 'bf16' if module.precision == 'fp8' else module.precision   # the stream runs fp8 modules as bf16
        self.dt_id, self.tdt = _DT[self.dt_name]
        self.FS = FS = sum(h * w_ for h, w_ in self.shapes)   # positions of one whole-pyramid frame (B = 1)
        self.lpos = []
        off = 0
        for h, w_ in self.shapes:
            self.lpos.append(off)
            off += h * w_
        self.pyr = _lib.make_pyramid(self.shapes)
        b = plan.buffers
        grouped = self.dt_name == 'bf16'
        self.rings = {
            'x': _Ring(plan.fp + G - 1, FS, plan.input_size, self.tdt, dev, grouped),
            'y_f1': _Ring(b['y_f1'].frames + G - 1, FS, 32, self.tdt, dev),
            'y_f2': _Ring(b['y_f2'].frames + G - 1, FS, 32, self.tdt, dev),
            'cat1': _Ring(b['cat1'].frames + G - 1, FS, 256, self.tdt, dev),
            'cat2': _Ring(b['cat2'].frames + G - 1, FS, 256, self.tdt, dev),
        }
        self.raw = {l.name: torch.empty((G * FS, l.c_out), dtype=self.tdt, device=dev) for l in plan.layers}
        self.out = torch.empty((G * FS, 256), dtype=self.tdt, device=dev)
        self.so = plan.fp // 2 - plan.sp // 2
        self.cf = {}          # layer -> eval coefficients [8, C] (identical for every level in eval mode)
        self.reset()
        self.refresh()

    # ------------------------------------------------------------------
    def refresh(self):
        """Re-read BatchNorm running statistics / affine parameters (call after loading a checkpoint); conv
        weights are re-packed on demand by the module's own cache."""
        st = _stream()
        for l in self.plan.layers:
            bn = getattr(self.m, l.bn)
            cf = torch.empty((1, _CF_ROWS, l.c_out), dtype=torch.float32, device=self.dev)
            _lib.call('sfvos_bn_eval_coeffs', _ptr(bn.weight.detach()), _ptr(bn.bias.detach()), _ptr(bn.running_mean),
                      _ptr(bn.running_var), float(bn.eps), l.c_out, _ptr(cf[0, _MEAN]), _ptr(cf[0, _RSTD]),
                      _ptr(cf[0, _SCALE]), _ptr(cf[0, _SHIFT]), st)
            self.cf[l.name] = cf

    def reset(self):
        """Start a new sequence."""
        self.n = 0                                                   # frames pushed so far
        self.done = {l.name: -1 for l in self.plan.layers}          # newest output frame index of each layer

    # ------------------------------------------------------------------
    def _write_input(self, feats, a):
        ring = self.rings['x']
        st = _stream()
        for slot, _, _ in ring.runs(a, 1):
            if feats is None:   # zero frame (padding beyond the ends of the sequence, model.py:215-225)
                if ring.grouped:
                    ring.buf[:, slot * self.FS:(slot + 1) * self.FS].zero_()
                else:
                    ring.buf[slot * self.FS:(slot + 1) * self.FS].zero_()
                continue
            for key, (H, W), lp in zip(self.keys, self.shapes, self.lpos):
                s = feats[key]
                if s.dim() == 4 and s.shape[0] == 1:
                    s = s[0]
                if tuple(s.shape) != (self.plan.input_size, H, W) or not s.is_cuda:
                    raise RuntimeError('level %s: expected a GPU tensor [%d,%d,%d], got %s on %s'
                                       % (key, self.plan.input_size, H, W, tuple(s.shape), s.device))
                s = s if s.dtype == torch.float32 else s.float()
                pos = slot * self.FS + lp
                if ring.grouped:
                    _lib.call('sfvos_frames_to_groups', _ptr(s), 0, s.stride(0), s.stride(1), s.stride(2),
                              _ptr(ring.buf, pos * 32), self.dt_id, 1, ring.C, H, W, ring.buf.shape[1] * 32, st)
                else:
                    _lib.call('sfvos_frames_to_ndhwc', _ptr(s), 0, s.stride(0), s.stride(1), s.stride(2),
                              _ptr(ring.buf, pos * ring.C), self.dt_id, 1, ring.C, H, W, ring.C, st)

    def _layer(self, name, src, src_shift, o_max, dst):
        """Output frames done[name]+1 .. o_max of layer `name` in one launch: window of kt + g - 1 frames of ring
        `src` starting at frame (first output + src_shift); result (BN + optional ReLU) -> those frames of ring
        `dst` (both copies) or the output buffer.  Returns the number of frames computed."""
        o0 = self.done[name] + 1
        g = o_max - o0 + 1
        if g <= 0:
            return 0
        assert g <= self.chunk
        l = self.plan.layer(name)
        ring = self.rings[src]
        conv = getattr(self.m, l.conv)
        d = _lib.ConvDesc()
        d.dtype, d.batch, d.kt, d.taps, d.pyr = self.dt_id, 1, l.kt, l.taps, self.pyr
        d.t_in, d.c_in, d.c_out, d.pad_t = l.kt + g - 1, l.c_in, l.c_out, 0
        d.t_alloc, d.t_offset = 2 * ring.T, (o0 + src_shift) % ring.T
        d.ld_y, d.accumulate = l.c_out, 0
        d.x_frame_stride = d.y_frame_stride = self.FS
        if ring.grouped:
            d.ld_x, d.x_group_stride = 32, ring.buf.shape[1] * 32
        else:
            d.ld_x, d.x_group_stride = ring.C, 0
        st = _stream()
        raw = self.raw[name]
        bias = _ptr(conv.bias.detach()) if conv.bias is not None else None
        _lib.call('sfvos_conv3d', ctypes.byref(d), _ptr(ring.buf), _ptr(self.m._packed(l, 'fwd', self.dt_name)), bias,
                  _ptr(raw), None, st)
        cf = self.cf[name]
        cs = _CF_ROWS * l.c_out
        if dst is None:
            targets = [(self.out, l.dst_off, 256, 0, g)]
        else:
            r = self.rings[dst]
            targets = [(r.buf, slot * self.FS * r.C + l.dst_off, r.C, foff, cnt) for slot, foff, cnt in r.runs(o0, g)]
        for buf, elem_off, ld, foff, cnt in targets:
            # eval-mode coefficients are the same for every level: a run of whole-pyramid frames is ONE "level"
            lv = _lib.make_levels([(1, self.FS)], 1, cnt)
            _lib.call('sfvos_bn_apply', _ptr(raw, foff * self.FS * l.c_out), l.c_out, _ptr(buf, elem_off), ld,
                      self.dt_id, ctypes.byref(lv), l.c_out, _ptr(cf[0, _SCALE]), _ptr(cf[0, _SHIFT]), cs,
                      1 if l.relu else 0, None, st)
        self.done[name] = o_max
        return g

    # ------------------------------------------------------------------
    def push_many(self, frames):
        """frames: list (1 .. chunk entries) of the next frames, each an OrderedDict level -> [256,H,W] (fp32, on the
        GPU) or None for a zero frame.  Returns the list of fused feature dicts (level -> [1,256,H,W] fp32) of the
        windows these frames complete, in order (empty while fewer than fp frames have been pushed)."""
        if self.m.training:
            raise RuntimeError('SlowFastStream is eval-mode inference: BatchNorm batch statistics are not shift-invariant')
        g_in = len(frames)
        if not 1 <= g_in <= self.chunk:
            raise ValueError('push_many takes 1..%d frames, got %d' % (self.chunk, g_in))
        p = self.plan
        kf, ks, (kl1, kl2) = p.k_fast, p.k_slow, p.k_lat
        T1s, T2s = p.buffers['cat1'].frames, p.buffers['cat2'].frames
        for i, f in enumerate(frames):
            self._write_input(f, self.n + i)
        self.n += g_in
        n = self.n - 1                       # newest input frame
        a_f1 = n - kf[0] + 1                 # newest computable frame of each stream
        a_f2 = a_f1 - kf[1] + 1
        a1 = n - p.fp + T1s
        a2 = n - p.fp + T2s
        a3 = n - p.fp + 1
        assert a_f1 - kl1 + 1 == a1 and a_f2 - kl2 + 1 == a2 and a1 - ks[1] + 1 == a2
        assert a2 - ks[2] + 1 == a3 and a_f2 - kf[2] + 1 == a3
        self._layer('f1', 'x', 0, a_f1, 'y_f1')
        self._layer('s1', 'x', self.so, a1, 'cat1')
        self._layer('l1', 'y_f1', 0, a1, 'cat1')
        self._layer('f2', 'y_f1', 0, a_f2, 'y_f2')
        self._layer('s2', 'cat1', 0, a2, 'cat2')
        self._layer('l2', 'y_f2', 0, a2, 'cat2')
        g_out = self._layer('s3', 'cat2', 0, a3, None)
        g_f3 = self._layer('f3', 'y_f2', 0, a3, None)
        assert g_out == g_f3
        st = _stream()
        results = []
        for j in range(g_out):
            merged = OrderedDict()
            for key, (H, W), lp in zip(self.keys, self.shapes, self.lpos):
                m = torch.empty((1, 256, H, W), dtype=torch.float32, device=self.dev)
                _lib.call('sfvos_ndhwc_to_frames', _ptr(self.out, (j * self.FS + lp) * 256), self.dt_id, _ptr(m),
                          256 * H * W, H * W, W, 1, 1, 256, H, W, 256, 0, st)
                merged[key] = m
            results.append(merged)
        return results

    def push(self, feats):
        """One frame: the fused features of the window it completes (centre frame = frames pushed - ceil(fp/2)), or
        None while fewer than fp frames have been pushed."""
        out = self.push_many([feats])
        return out[0] if out else None

    def run_sequence(self, frames):
        """frames: list (length N) of per-frame feature dicts.  Returns the N fused feature dicts the reference's
        per-frame loop computes (window of fp frames around each frame, zero frames beyond the ends)."""
        self.reset()
        fp = self.plan.fp
        seq = [None] * (fp // 2) + list(frames) + [None] * (fp - fp // 2 - 1)
        outs = []
        for i in range(0, len(seq), self.chunk):
            outs.extend(self.push_many(seq[i:i + self.chunk]))
        assert len(outs) == len(frames)
        return outs
