"""Host-side planning for the SlowFastLayers hot path: kernel-size tables, the layer graph and
the per-level buffer shapes.  Pure Python (no GPU, no torch) so it is unit-testable on CPU.

Mirrors reference code/helpers/model.py:37-69 (module table), :96-109 (kernel sizes),
:118-149 (graph)."""
from collections import namedtuple


def calc_kernel_sizes(pathway_size):
    """Three valid temporal kernels taking T from pathway_size to 1 (model.py:96-103)."""
    if pathway_size < 1:
        raise ValueError('pathway size must be >= 1, got %r' % (pathway_size,))
    div, rem = divmod(pathway_size, 3)
    if rem == 0:
        return (div, div + 1, div + 1)
    if rem == 1:
        return (div + 1, div + 1, div + 1)
    return (div + 1, div + 1, div + 2)


def calc_fuse_kernel_size(slow_in, slow_kernel, fast_in, fast_kernel):
    """Lateral kernel mapping fast's T' onto slow's T' with a valid conv (model.py:105-109)."""
    out_slow = slow_in - slow_kernel + 1
    out_fast = fast_in - fast_kernel + 1
    return out_fast - out_slow + 1, out_slow, out_fast


# One conv+BN(+ReLU) block of the graph.
#   src/dst name activation buffers; dst_off is the channel offset inside dst (concat by slice-write)
Layer = namedtuple('Layer', 'name conv bn src c_in c_out kt taps relu raw dst dst_off t_in t_out')
Buffer = namedtuple('Buffer', 'name frames channels')


class SlowFastPlan(object):
    """Everything that depends only on (input_size, sp, fp)."""

    def __init__(self, input_size, slow_pathway_size, fast_pathway_size):
        sp, fp = int(slow_pathway_size), int(fast_pathway_size)
        if input_size % 32 != 0:
            raise ValueError('input_size must be a multiple of 32 (MFMA tile), got %d' % input_size)
        self.input_size, self.sp, self.fp = input_size, sp, fp
        ks = calc_kernel_sizes(sp)
        kf = calc_kernel_sizes(fp)
        l1, so1, fo1 = calc_fuse_kernel_size(sp, ks[0], fp, kf[0])
        l2, so2, fo2 = calc_fuse_kernel_size(so1, ks[1], fo1, kf[1])
        if l1 < 1 or l2 < 1:
            raise ValueError('fast pathway (%d) too short for slow pathway (%d): lateral kernel < 1' % (fp, sp))
        self.k_slow, self.k_fast, self.k_lat = ks, kf, (l1, l2)
        so3, fo3 = so2 - ks[2] + 1, fo2 - kf[2] + 1
        assert so3 == 1 and fo3 == 1
        C = input_size
        self.layers = [
            Layer('s1', 'slow_conv1', 'bn_s1', 'xs0', C, 192, ks[0], 9, True, 'raw_s1', 'cat1', 0, sp, so1),
            Layer('f1', 'fast_conv1', 'bn_f1', 'xf0', C, 32, kf[0], 9, True, 'raw_f1', 'y_f1', 0, fp, fo1),
            Layer('l1', 'conv_f2s1', 'bn_f2s1', 'y_f1', 32, 64, l1, 1, True, 'raw_l1', 'cat1', 192, fo1, so1),
            Layer('s2', 'slow_conv2', 'bn_s2', 'cat1', 256, 192, ks[1], 9, True, 'raw_s2', 'cat2', 0, so1, so2),
            Layer('f2', 'fast_conv2', 'bn_f2', 'y_f1', 32, 32, kf[1], 9, True, 'raw_f2', 'y_f2', 0, fo1, fo2),
            Layer('l2', 'conv_f2s2', 'bn_f2s2', 'y_f2', 32, 64, l2, 1, True, 'raw_l2', 'cat2', 192, fo2, so2),
            Layer('s3', 'slow_conv3', 'bn_s3', 'cat2', 256, 224, ks[2], 9, False, 'raw_s3', 'out', 0, so2, 1),
            Layer('f3', 'fast_conv3', 'bn_f3', 'y_f2', 32, 32, kf[2], 9, False, 'raw_f3', 'out', 224, fo2, 1),
        ]
        self.buffers = {b.name: b for b in [
            Buffer('xs0', sp, C), Buffer('xf0', fp, C),
            Buffer('raw_s1', so1, 192), Buffer('raw_f1', fo1, 32), Buffer('raw_l1', so1, 64),
            Buffer('cat1', so1, 256), Buffer('y_f1', fo1, 32),
            Buffer('raw_s2', so2, 192), Buffer('raw_f2', fo2, 32), Buffer('raw_l2', so2, 64),
            Buffer('cat2', so2, 256), Buffer('y_f2', fo2, 32),
            Buffer('raw_s3', 1, 224), Buffer('raw_f3', 1, 32), Buffer('out', 1, 256)]}
        # the registration order of the reference (= state-dict and RNG-init order, model.py:47-67)
        self.module_order = ['fast_conv1', 'bn_f1', 'slow_conv1', 'bn_s1', 'fast_conv2', 'bn_f2', 'slow_conv2',
                             'bn_s2', 'fast_conv3', 'bn_f3', 'slow_conv3', 'bn_s3', 'conv_f2s1', 'bn_f2s1',
                             'conv_f2s2', 'bn_f2s2']

    def layer(self, name):
        for l in self.layers:
            if l.name == name:
                return l
        raise KeyError(name)

    def conv_shapes(self):
        """conv module name -> (c_in, c_out, kt, kh, kw, bias)."""
        out = {}
        for l in self.layers:
            k = 3 if l.taps == 9 else 1
            out[l.conv] = (l.c_in, l.c_out, l.kt, k, k, l.taps == 9)
        return out

    def param_count(self):
        n = 0
        for l in self.layers:
            n += l.c_in * l.c_out * l.kt * l.taps + (l.c_out if l.taps == 9 else 0) + 2 * l.c_out
        return n

    # ---- algorithmic work per clip -------------------------------------------------
    def forward_flops(self, positions):
        return sum(2.0 * l.c_in * l.c_out * l.kt * l.taps * l.t_out * positions for l in self.layers)

    def train_flops(self, positions, first_layer_dgrad=False):
        """fwd + wgrad for every conv + dgrad for the convs whose input needs a gradient."""
        fwd = self.forward_flops(positions)
        dgrad = sum(2.0 * l.c_in * l.c_out * l.kt * l.taps * l.t_out * positions for l in self.layers
                    if first_layer_dgrad or l.src not in ('xs0', 'xf0'))
        return 2.0 * fwd + dgrad

    def layer_flops(self, positions):
        return {l.name: 2.0 * l.c_in * l.c_out * l.kt * l.taps * l.t_out * positions for l in self.layers}

    def layer_bytes(self, positions, elt_bytes):
        """Compulsory bytes of each conv forward: read input once, write output once, read weights."""
        out = {}
        for l in self.layers:
            out[l.name] = (l.t_in * positions * l.c_in + l.t_out * positions * l.c_out
                           + l.c_in * l.c_out * l.kt * l.taps) * elt_bytes
        return out


def davis_pyramid():
    """(H, W) of the five FPN levels for a 480x854 DAVIS frame after torchvision's default
    transform (resize to 749x1333, pad to 768x1344) -- SURVEY.md 8 'P = 85 932'."""
    return [('0', (192, 336)), ('1', (96, 168)), ('2', (48, 84)), ('3', (24, 42)), ('pool', (12, 21))]
