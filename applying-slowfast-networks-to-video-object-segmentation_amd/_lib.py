"""ctypes binding of libsfvos.so -- the only path from Python to the HIP kernels.

There is no fallback: if the library is missing or a call fails, a RuntimeError is raised
(the product must never silently compute on another path)."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('SFVOS_LIB') or os.path.join(_HERE, 'csrc', 'libsfvos.so')  # SFVOS_LIB: A/B builds

F32, BF16, FP8 = 0, 1, 2
ABI_REVISION = 300   # the revision of include/sfvos.h this binding mirrors
MAX_LEVELS = 8

vp, i32, i64, f32 = C.c_void_p, C.c_int, C.c_int64, C.c_float


class Pyramid(C.Structure):
    """Mirror of sfvos_pyramid."""
    _fields_ = [('n_levels', i32), ('h', i32 * MAX_LEVELS), ('w', i32 * MAX_LEVELS)]


class Levels(C.Structure):
    """Mirror of sfvos_levels: positions per level of a flat pyramid buffer."""
    _fields_ = [('n_levels', i32), ('m', i64 * MAX_LEVELS)]


class ConvDesc(C.Structure):
    """Mirror of sfvos_conv_desc (include/sfvos.h).  struct_size is filled in by the constructor; the library
    rejects a descriptor whose size differs from its own struct (stale binding)."""
    _fields_ = [('struct_size', i32), ('dtype', i32), ('batch', i32), ('t_in', i32), ('t_alloc', i32), ('t_offset', i32), ('c_in', i32),
                ('c_out', i32), ('kt', i32), ('taps', i32), ('pad_t', i32), ('ld_x', i32), ('ld_y', i32),
                ('accumulate', i32), ('relu', i32), ('pyr', Pyramid), ('x_group_stride', i64), ('x_frame_stride', i64), ('y_frame_stride', i64)]


    def __init__(self, *args, **kw):
        super(ConvDesc, self).__init__(*args, **kw)
        self.struct_size = C.sizeof(ConvDesc)


class MseTable(C.Structure):
    """Mirror of sfvos_mse_table."""
    _fields_ = [('n', i32), ('out', vp * MAX_LEVELS), ('target', vp * MAX_LEVELS), ('grad', vp * MAX_LEVELS),
                ('numel', i64 * MAX_LEVELS)]


class BnRunning(C.Structure):
    """Mirror of sfvos_bn_running."""
    _fields_ = [('running_mean', vp), ('running_var', vp), ('means', vp), ('vars_unbiased', vp),
                ('num_batches_tracked', vp), ('n_updates', i32), ('momentum', f32)]


class PackItem(C.Structure):
    """Mirror of sfvos_pack_item."""
    _fields_ = [('w', vp), ('packed', vp), ('c_out', i32), ('c_in', i32), ('kt', i32), ('taps', i32), ('dgrad', i32)]


MAX_PACK_ITEMS = 16


class PlanarLevel(C.Structure):
    """Mirror of sfvos_planar_level."""
    _fields_ = [('ptr', vp), ('stride_t', i64), ('stride_c', i64), ('stride_h', i64), ('stride_w', i64), ('h', i32),
                ('w', i32)]


def make_pyramid(shapes):
    """shapes: list of (H, W) per level."""
    if not 1 <= len(shapes) <= MAX_LEVELS:
        raise ValueError('a pyramid has 1..%d levels, got %d' % (MAX_LEVELS, len(shapes)))
    p = Pyramid()
    p.n_levels = len(shapes)
    for l, (h, w) in enumerate(shapes):
        p.h[l], p.w[l] = int(h), int(w)
    return p


def make_levels(shapes, batch, frames):
    lv = Levels()
    lv.n_levels = len(shapes)
    for l, (h, w) in enumerate(shapes):
        lv.m[l] = int(batch) * int(frames) * int(h) * int(w)
    return lv


PD, PL, PM, PR = C.POINTER(ConvDesc), C.POINTER(Levels), C.POINTER(MseTable), C.POINTER(BnRunning)

# name -> (restype, argtypes); every symbol include/sfvos.h declares
SIGNATURES = {
    'sfvos_version': (i32, []),
    'sfvos_abi_sizes': (i32, [C.POINTER(i32), i32]),
    'sfvos_last_error': (C.c_char_p, []),
    'sfvos_check_device': (i32, []),
    'sfvos_frames_to_ndhwc': (i32, [vp, i64, i64, i64, i64, vp, i32, i32, i32, i32, i32, i32, vp]),
    'sfvos_frames_to_groups': (i32, [vp, i64, i64, i64, i64, vp, i32, i32, i32, i32, i32, i64, vp]),
    'sfvos_frames_to_groups_fp8': (i32, [vp, i64, i64, i64, i64, vp, i32, i32, i32, i32, i64, f32, vp, vp]),
    'sfvos_frames_absmax': (i32, [vp, i64, i64, i64, i64, i32, i32, i32, i32, vp, vp]),
    'sfvos_pack_weights_fp8': (i32, [vp, vp, vp, vp, i32, i32, i32, i32, f32, vp]),
    'sfvos_ndhwc_to_planar': (i32, [vp, i32, vp, i64, i32, i32, vp]),
    'sfvos_planar_to_ndhwc': (i32, [vp, vp, i32, i64, i32, i32, vp]),
    'sfvos_ndhwc_to_frames': (i32, [vp, i32, vp, i64, i64, i64, i64, i32, i32, i32, i32, i32, i32, vp]),
    'sfvos_pyramid_to_frames': (i32, [vp, i32, C.POINTER(PlanarLevel), i32, i32, i32, i32, i32, vp]),
    'sfvos_frames_to_pyramid': (i32, [C.POINTER(PlanarLevel), i32, vp, i32, i32, i32, i32, vp]),
    'sfvos_packed_weight_bytes': (C.c_size_t, [i32, i32, i32, i32, i32]),
    'sfvos_pack_weights_fwd': (i32, [vp, vp, i32, i32, i32, i32, i32, vp]),
    'sfvos_pack_weights_dgrad': (i32, [vp, vp, i32, i32, i32, i32, i32, vp]),
    'sfvos_pack_weights_batch': (i32, [C.POINTER(PackItem), i32, i32, vp]),
    'sfvos_conv3d_stat_rows': (i32, [PD, C.POINTER(i32)]),
    'sfvos_conv3d': (i32, [PD, vp, vp, vp, vp, vp, vp]),
    'sfvos_conv3d_wgrad_workspace_bytes': (C.c_size_t, [PD]),
    'sfvos_conv3d_wgrad': (i32, [PD, vp, vp, vp, i32, vp, vp]),
    'sfvos_bn_finalize': (i32, [vp, i32, C.POINTER(i32), C.POINTER(i64), vp, vp, f32, i32, vp, vp, vp, vp, vp, i32, vp]),
    'sfvos_bn_eval_coeffs': (i32, [vp, vp, vp, vp, f32, i32, vp, vp, vp, vp, vp]),
    'sfvos_bn_running_update': (i32, [vp, vp, vp, vp, i32, i32, i32, f32, vp, vp]),
    'sfvos_bn_apply': (i32, [vp, i32, vp, i32, i32, PL, i32, vp, vp, i32, i32, PR, vp]),
    'sfvos_bn_apply_fp8': (i32, [vp, i32, vp, i32, PL, i32, vp, vp, i32, i32, f32, vp, PR, vp]),
    'sfvos_bn_bwd_rows': (i32, [PL]),
    'sfvos_bn_bwd_reduce': (i32, [vp, i32, vp, i32, i32, PL, i32, vp, vp, vp, vp, i32, i32, vp, vp]),
    'sfvos_bn_bwd_finalize': (i32, [vp, PL, vp, vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, vp]),
    'sfvos_bn_bwd_apply': (i32, [vp, i32, vp, i32, vp, i32, i32, PL, i32, vp, vp, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp]),
    'sfvos_reduce_rows': (i32, [vp, i32, i32, vp, i32, vp]),
    'sfvos_sgd_step': (i32, [vp, vp, vp, i64, f32, f32, f32, i32, vp]),
    'sfvos_scale': (i32, [vp, i64, f32, vp]),
    'sfvos_add_inplace': (i32, [vp, vp, i32, i64, vp]),
    'sfvos_mse_loss_rows': (i32, [PM]),
    'sfvos_mse_loss': (i32, [PM, vp, vp, vp]),
    'sfvos_mse_loss_grad': (i32, [PM, vp, vp]),
    'sfvos_mask_union': (i32, [vp, i32, i64, f32, vp, vp]),
    'sfvos_pack_deconv2x2': (i32, [vp, vp, i32, i32, i32, vp]),
    'sfvos_deconv2x2_relu': (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]),
    'sfvos_mask_logits': (i32, [vp, i32, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp]),
    'sfvos_paste_masks': (i32, [vp, vp, i32, i32, i32, i32, i32, vp, vp]),
    'sfvos_relu_bwd_rows': (i32, [i64]),
    'sfvos_relu_bwd': (i32, [vp, vp, vp, i32, i64, i32, vp, vp]),
    'sfvos_mask_bce_loss': (i32, [vp, vp, vp, i32, i32, i32, vp, vp]),
    'sfvos_mask_bce_loss_grad': (i32, [vp, vp, vp, vp, i32, i32, i32, vp, vp]),
    'sfvos_mask_logits_bwd_rows': (i32, [i32, i32]),
    'sfvos_mask_logits_bwd': (i32, [vp, i32, vp, vp, i32, i32, i32, i32, i32, vp, vp, vp]),
    'sfvos_pack_deconv2x2_dgrad': (i32, [vp, vp, i32, i32, i32, vp]),
    'sfvos_deconv2x2_dgrad': (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    'sfvos_deconv2x2_wgrad_workspace_bytes': (C.c_size_t, [i32, i32, i32, i32, i32]),
    'sfvos_deconv2x2_wgrad': (i32, [vp, vp, i32, i32, i32, i32, i32, i32, vp, i32, vp, vp]),
    'sfvos_roi_levels': (i32, [vp, i32, i32, i32, f32, f32, f32, vp, vp]),
    'sfvos_roi_align_workspace_bytes': (C.c_size_t, [i32]),
    'sfvos_roi_align': (i32, [vp, i32, i32, i32, i32, vp, vp, i32, i32, f32, i32, i32, vp, vp, vp]),
    'sfvos_roi_align_bwd': (i32, [vp, i32, i32, i32, i32, vp, vp, i32, i32, f32, i32, i32, vp, vp, i32, vp]),
}

_lib = None


def load():
    """Load libsfvos.so (once).  Raises RuntimeError -- never falls back -- when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            'sfvos_amd: %s not found. Build it with `python __graft_entry__.py` (hipcc, gfx950); '
            'there is no CPU or PyTorch fallback for the SlowFastLayers kernels.' % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    # the binder's struct mirrors must have the library's sizes (a short struct would be read past its end)
    sizes = (i32 * 7)()
    n = lib.sfvos_abi_sizes(sizes, 7)
    mine = [C.sizeof(ConvDesc), C.sizeof(Pyramid), C.sizeof(Levels), C.sizeof(MseTable), C.sizeof(BnRunning),
            C.sizeof(PackItem), C.sizeof(PlanarLevel)]
    if lib.sfvos_version() < ABI_REVISION or n != 7 or list(sizes) != mine:
        raise RuntimeError('sfvos_amd: %s is ABI revision %d with struct sizes %s, this binding expects revision >= %d '
                           'and %s -- rebuild with `python __graft_entry__.py`' % (LIB_PATH, lib.sfvos_version(),
                                                                                   list(sizes)[:n], ABI_REVISION, mine))
    _lib = lib
    return lib


def check(rc, what=''):
    if rc != 0:
        msg = load().sfvos_last_error()
        raise RuntimeError('libsfvos %s failed (%d): %s' % (what, rc, msg.decode() if msg else ''))


def call(name, *args):
    """Invoke an int-returning entry point and raise on a non-zero status."""
    check(getattr(load(), name)(*args), name)


# Bumped by anything that rewrites parameters behind autograd's back (FusedSGD); the packed
# MFMA weight images cached by SlowFastLayers are keyed on it.
_weight_epoch = [0]


def bump_weight_epoch():
    _weight_epoch[0] += 1


def weight_epoch():
    return _weight_epoch[0]
