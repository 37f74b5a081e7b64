/*
 * sfvos.h -- C ABI of libsfvos.so: hand-written HIP (gfx950 / CDNA4) kernels for the
 * SlowFastLayers hot path of ChantalMP/Applying-SlowFast-networks-to-video-object-segmentation.
 *
 * The reference has no FFI layer; its boundary is the nn.Module surface of
 * `SlowFastLayers` (reference code/helpers/model.py:30-165).  The Python host
 * (sfvos_amd.SlowFastLayers) mirrors that surface and binds exactly the entry
 * points below through ctypes.  Each entry point names the
 * reference lines whose ATen dispatch it replaces.
 *
 * Conventions:
 *   - plain C; raw DEVICE pointers + explicit dims/strides; no torch types;
 *   - no allocation, no ownership transfer, no implicit device sync: every byte of
 *     workspace is caller-provided, every launch goes to the caller's hipStream_t;
 *   - re-entrant per stream, no thread-local or global mutable state except the
 *     per-thread last-error string;
 *   - return 0 on success, a negative SFVOS_E_* code otherwise; sfvos_last_error()
 *     gives the message for the calling thread.
 *
 * Pyramid layout ("pyramid NDHWC").  temporally_enhance_features (model.py:151-165) runs the
 * same 8 conv+BN blocks on every FPN level; here ONE launch covers all levels.  An activation
 * with T frames over a pyramid of L levels (H_l x W_l pixels) and B clips is one buffer of
 * M = B*T*sum_l(H_l*W_l) positions, level-major:
 *     position(l,b,t,h,w) = B*T*sum_{l'<l}(H_l'*W_l') + ((b*T + t)*H_l + h)*W_l + w
 *     element (.., c)      at base + position*ld + c
 * `ld` >= C is the per-position pitch in ELEMENTS, so a producer can write into a channel slice
 * of a wider buffer (this is how torch.cat at model.py:115,162 is eliminated).  BatchNorm
 * statistics stay per level (the reference normalises each level's call separately).
 */
#ifndef SFVOS_H
#define SFVOS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* sfvos_stream_t; /* hipStream_t */

enum { SFVOS_F32 = 0, SFVOS_BF16 = 1,
       SFVOS_FP8 = 2 /* OCP e4m3 operands; inference-only first step, see sfvos_conv_desc.dtype */ };

enum {
  SFVOS_OK = 0,
  SFVOS_E_ARG = -1,     /* bad argument / unsupported shape */
  SFVOS_E_LAUNCH = -2,  /* HIP launch error */
  SFVOS_E_NODEV = -3    /* no gfx950 device */
};

#define SFVOS_MAX_LEVELS 8

typedef struct sfvos_pyramid {
  int n_levels;
  int h[SFVOS_MAX_LEVELS];
  int w[SFVOS_MAX_LEVELS];
} sfvos_pyramid;

/* How a flat run of positions splits into levels: m[l] = B*T*H_l*W_l. */
typedef struct sfvos_levels {
  int n_levels;
  int64_t m[SFVOS_MAX_LEVELS];
} sfvos_levels;

/* ABI revision: 100 = round 1; 200 = struct_size in sfvos_conv_desc, no `zeros` argument, sfvos_abi_sizes,
 * sfvos_add_inplace, sfvos_mse_loss / _grad, mask-head entry points; 201 = sfvos_bn_running argument of the BN-apply
 * calls, mask-branch training entry points, sfvos_pack_weights_batch, sfvos_pyramid_to_frames / sfvos_frames_to_pyramid;
 * 300 (round 3) = sfvos_roi_levels / sfvos_roi_align / sfvos_roi_align_bwd (+ workspace query). */
int sfvos_version(void);
/* sizes[0..n) = sizeof(sfvos_conv_desc), sizeof(sfvos_pyramid), sizeof(sfvos_levels), sizeof(sfvos_mse_table),
 * sizeof(sfvos_bn_running), sizeof(sfvos_pack_item), sizeof(sfvos_planar_level) as THIS library was compiled; returns
 * how many entries exist (7).  A binder compares them with its own mirrors of the structs once at load time (a short struct would make the
 * library read strides from whatever follows it). */
int sfvos_abi_sizes(int* sizes, int n);
const char* sfvos_last_error(void);
/* 0 when the current HIP device is a gfx950; SFVOS_E_NODEV otherwise. */
int sfvos_check_device(void);

/* ---- layout (one level per call: the caller's tensors are separate per level) ------------- */

/* Frames fp32 [T][C][H][W] addressed through explicit ELEMENT strides (the reference's
 * stack().transpose(1,2) view is non-contiguous, model.py:157-158) -> NDHWC dst[T][H][W][ld].
 * Never writes the source. */
int sfvos_frames_to_ndhwc(const float* src, int64_t stride_t, int64_t stride_c, int64_t stride_h,
                          int64_t stride_w, void* dst, int dtype, int T, int C, int H, int W, int ld,
                          sfvos_stream_t stream);

/* Same source -> the channel-group-major layout of sfvos_conv_desc.x_group_stride (bf16 only): element
 * (t,h,w,c) at dst + (c/32)*group_stride + ((t*H+h)*W+w)*32 + c%32; C a multiple of 32. */
int sfvos_frames_to_groups(const float* src, int64_t stride_t, int64_t stride_c, int64_t stride_h,
                           int64_t stride_w, void* dst, int dtype, int T, int C, int H, int W,
                           int64_t group_stride, sfvos_stream_t stream);

/* NDHWC src[M][ld] (first C channels) -> planar fp32 dst[C][M]. */
int sfvos_ndhwc_to_planar(const void* src, int dtype, float* dst, int64_t M, int C, int ld,
                          sfvos_stream_t stream);

/* planar fp32 src[C][M] -> NDHWC dst[M][ld]. */
int sfvos_planar_to_ndhwc(const float* src, void* dst, int dtype, int64_t M, int C, int ld,
                          sfvos_stream_t stream);

/* NDHWC src[T][H][W][ld] (first C channels) accumulated (+=) or stored into fp32 frames
 * [T][C][H][W] with explicit strides: the fused output of model.py:162 as the caller's NCHW
 * tensor, and the input gradient of model.py:157-158 (reference code/osvos/osvos_model.py:50,64). */
int sfvos_ndhwc_to_frames(const void* src, int dtype, float* dst, int64_t stride_t, int64_t stride_c,
                          int64_t stride_h, int64_t stride_w, int T, int C, int H, int W, int ld,
                          int accumulate, sfvos_stream_t stream);

/* The two passes above for EVERY level of a pyramid in ONE launch (SlowFastLayers hands its five fused maps / takes
 * their gradients level by level: five launches of a few microseconds become one).  Level l: planar fp32 tensor
 * [T][C][h][w] addressed through element strides <-> the level's T*h*w positions of a level-major NDHWC pyramid buffer
 * with pitch ld (levels in order, level l at position T * sum_{l' < l} h_l' w_l'). */
typedef struct sfvos_planar_level {
  float* ptr;
  int64_t stride_t, stride_c, stride_h, stride_w;
  int h, w;
} sfvos_planar_level;
int sfvos_pyramid_to_frames(const void* src, int dtype, const sfvos_planar_level* levels, int n_levels, int T, int C,
                            int ld, int accumulate, sfvos_stream_t stream);
int sfvos_frames_to_pyramid(const sfvos_planar_level* levels, int n_levels, void* dst, int dtype, int T, int C, int ld,
                            sfvos_stream_t stream);

/* ---- weights ------------------------------------------------------------------------- */

/* Bytes of a packed weight image for a conv with these dims (same for fwd and dgrad packs). */
size_t sfvos_packed_weight_bytes(int dtype, int c_out, int c_in, int kt, int taps);

/* fp32 nn.Conv3d weight w[Cout][Cin][kt][kh][kw] (taps = kh*kw = 9 or 1; state-dict layout,
 * SURVEY.md 8b) -> the MFMA B-operand image consumed by sfvos_conv3d (forward). */
int sfvos_pack_weights_fwd(const float* w, void* packed, int dtype, int c_out, int c_in, int kt, int taps,
                           sfvos_stream_t stream);

/* Same weight -> the image for the data-gradient conv: channels swapped, all three kernel axes
 * flipped.  Feeding it to sfvos_conv3d with pad_t = kt-1 computes aten::convolution_backward's
 * grad_input. */
int sfvos_pack_weights_dgrad(const float* w, void* packed, int dtype, int c_out, int c_in, int kt, int taps,
                             sfvos_stream_t stream);

/* Several images in ONE launch (every layer of the model after an optimiser step): items[i] is one
 * sfvos_pack_weights_fwd (dgrad == 0) or sfvos_pack_weights_dgrad (dgrad != 0) call; 1 <= n <= SFVOS_MAX_PACK_ITEMS. */
#define SFVOS_MAX_PACK_ITEMS 16
typedef struct sfvos_pack_item {
  const float* w;
  void* packed;
  int c_out, c_in, kt, taps, dgrad;
} sfvos_pack_item;
int sfvos_pack_weights_batch(const sfvos_pack_item* items, int n, int dtype, sfvos_stream_t stream);

/* e4m3 operands (SFVOS_FP8).  Frames fp32 -> 64-channel groups of e4m3: element (t,h,w,c) = sat(src * scale) at byte
 * dst + (c/64)*group_stride + ((t*H+h)*W+w)*64 + c%64; C a multiple of 64. */
int sfvos_frames_to_groups_fp8(const float* src, int64_t stride_t, int64_t stride_c, int64_t stride_h,
                               int64_t stride_w, void* dst, int T, int C, int H, int W, int64_t group_stride,
                               float scale, int* sat_count, sfvos_stream_t stream);
/* sat_count (device int, may be NULL): += number of elements with |src * scale| > 448 (they saturate silently in
 * e4m3; FPN features are unbounded conv outputs, so a fixed scale must be checked against the data). */

/* *amax (device float holding a non-negative value, e.g. 0) = max(*amax, max |src|) over the strided fp32 frames:
 * calibration of the e4m3 activation scale (scale = 448 / amax with head-room).  Order-independent, deterministic. */
int sfvos_frames_absmax(const float* src, int64_t stride_t, int64_t stride_c, int64_t stride_h, int64_t stride_w,
                        int T, int C, int H, int W, float* amax, sfvos_stream_t stream);

/* conv weight fp32 [c_out][c_in][kt][kh][kw] -> e4m3 forward image for sfvos_conv3d (dtype SFVOS_FP8), quantised per
 * output channel: weight_scale[n] = 448 / max|w[n]|.  bias_descale: [3][c_out] floats: row 0 = bias (zeros when bias is
 * NULL), row 1 = 1 / (act_scale * weight_scale[n]), row 2 = weight_scale[n] (sfvos_conv3d reads rows 0 and 1).
 * c_in a multiple of 64. */
int sfvos_pack_weights_fp8(const float* w, const float* bias, void* packed, float* bias_descale, int c_out, int c_in,
                           int kt, int taps, float act_scale, sfvos_stream_t stream);

/* ---- convolution (replaces aten::convolution at model.py:112,120,124,132,136,144,147) ---- */

typedef struct sfvos_conv_desc {
  int struct_size; /* = sizeof(sfvos_conv_desc) of the binder's mirror; every entry point taking a desc rejects a mismatch */
  int dtype;       /* SFVOS_F32: f32 storage, exact-f32 MFMA; SFVOS_BF16: bf16 storage, f32 accumulate;
                    * SFVOS_FP8 (sfvos_conv3d only, forward 3x3 layers with c_in a multiple of 64): x and the weight image are OCP e4m3
                    * (x in 64-channel = 64-byte groups: x_group_stride > 0 counted in elements = bytes, or plain NDHWC
                    * with ld_x a multiple of 64), products on v_mfma_scale_f32_32x32x64_f8f6f4 (2x the bf16 rate),
                    * f32 accumulate, y stored as bf16; `bias` then points to [2][c_out] floats: row 0 the bias, row 1
                    * the per-output-channel de-quantisation factor 1/(activation scale * weight scale[n])
                    * (sfvos_pack_weights_fp8 writes both rows): y = acc * row1 + row0 */
  int batch;       /* clips B */
  int t_in;        /* input frames the conv sees: its window is frames [t_offset, t_offset + t_in) of the x buffer */
  int t_alloc;     /* frames per clip the x BUFFER holds.  Lets the slow pathway read its centre frames            */
  int t_offset;    /* (model.py:242-248) straight out of the fast clip's buffer.  Level-major x only: the window may
                    * stick out of [0, t_alloc) on either side (t_offset < 0, or t_offset + t_in > t_alloc); those
                    * frames are ZERO frames that are never read -- the reference's zero feature padding beyond the
                    * ends of a sequence (model.py:215-225) without materialising zeros */
  int c_in, c_out; /* multiples of 32 */
  int kt;          /* temporal taps */
  int taps;        /* 9 = 3x3 spatial, zero pad 1 ; 1 = 1x1 spatial, no pad */
  int pad_t;       /* zero frames each side in time: 0 (forward), kt-1 (data gradient) */
  int ld_x, ld_y;  /* per-position pitch of x / y in elements */
  int accumulate;  /* y += conv(x) instead of y = conv(x)  (gradient fan-in) */
  int relu;        /* y = max(conv(x) + bias, 0): the conv3x3 + ReLU blocks of the mask head (no statistics, no
                    * accumulate with it) */
  sfvos_pyramid pyr; /* spatial extents of the levels (stride 1: output spatial == input spatial) */
  int64_t x_group_stride; /* 0: x is pyramid NDHWC with pitch ld_x.  > 0 (bf16 only): x is stored as 64-byte
                           * channel groups, element (position, c) at (c/32)*x_group_stride + position*32 + c%32
                           * (ld_x ignored): every 128-byte line a kernel touches is used whole, so the
                           * temporal re-reads of the input hit in L2 instead of fetching half-used lines. */
  int64_t x_frame_stride; /* 0: x is level-major (all t_alloc frames of a level are consecutive).  > 0: x is a
                           * FRAME-major ring (sequence inference): frame t of every level sits at position
                           * t*x_frame_stride + B*sum_{l'<l} H_l'W_l' + (b*H_l+h)*W_l+w, so one frame of the whole
                           * pyramid is one contiguous slot; sfvos_conv3d only (no weight gradient). */
  int64_t y_frame_stride; /* the same for y: 0 level-major, > 0 frame-major (output frame t of every level at
                           * position t*y_frame_stride + ...): consecutive output frames are whole-pyramid slots. */
} sfvos_conv_desc;

/* Partial-statistics rows ([2][c_out] fp32 each, one per workgroup tile) sfvos_conv3d writes for
 * this desc: returns the total and, when rows_per_level != NULL, the count per level (rows are
 * level-major).  They are summed in fixed order by sfvos_bn_finalize -> deterministic. */
int sfvos_conv3d_stat_rows(const sfvos_conv_desc* d, int* rows_per_level);

/* y (=|+=) bias + sum_{dt,dh,dw,c} x[l][b][t_offset+t+dt-pad_t][h+dh-1][w+dw-1][c] * W over every level.
 * x: pyramid buffer with t_alloc frames; y: pyramid buffer with t_out = t_in + 2*pad_t - kt + 1 frames.
 * bias may be NULL.  stat_part may be NULL; otherwise it receives per-tile partial (sum, sum of
 * squares) of the values written.  Spatial / temporal zero padding is produced by the buffer descriptors' range
 * check (frames outside [0, t_in) and pixels outside the image are never read). */
int sfvos_conv3d(const sfvos_conv_desc* d, const void* x, const void* w_packed, const float* bias, void* y,
                 float* stat_part, sfvos_stream_t stream);

/* ---- weight gradient (aten::convolution_backward grad_weight / grad_bias) -------------- */

/* Workspace bytes for sfvos_conv3d_wgrad (fp32 split-K slabs). `d` describes the FORWARD conv. */
size_t sfvos_conv3d_wgrad_workspace_bytes(const sfvos_conv_desc* d);

/* grad_w[Cout][Cin][kt][kh][kw] (fp32, state-dict layout) (=|+=) sum over levels/clips/pixels of
 * dy[pos][n] * x[pos+shift][c].  x is the forward input (pyramid buffer, ld_x, t_alloc/t_offset),
 * dy the gradient w.r.t. the conv output (pyramid buffer with t_out frames, ld_y).
 * accumulate != 0 adds into grad_w (model.py:369-374 accumulates two clips before stepping). */
int sfvos_conv3d_wgrad(const sfvos_conv_desc* d, const void* x, const void* dy, float* grad_w, int accumulate,
                       void* workspace, sfvos_stream_t stream);

/* ---- batch norm (aten::native_batch_norm / _backward at model.py:113,121,...,148) -------
 * Per-level coefficient tables: quantity q of level l lives at q_ptr + l*coef_stride (elements). */

/* Reduce the partial rows of each level (rows_per_level, level-major) in fixed order; train statistics
 * over count_per_level values per channel: mean, biased var -> rstd = 1/sqrt(var+eps),
 * scale = gamma*rstd, shift = beta - mean*scale, plus the unbiased variance for the running stats. */
int sfvos_bn_finalize(const float* part, int n_levels, const int* rows_per_level, const int64_t* count_per_level,
                      const float* gamma, const float* beta, float eps, int C, float* mean, float* rstd, float* scale,
                      float* shift, float* save_var_unbiased, int coef_stride, sfvos_stream_t stream);

/* Eval mode: scale = gamma / sqrt(running_var + eps), shift = beta - running_mean * scale; also
 * mean = running_mean, rstd = 1/sqrt(running_var+eps) (for an eval-mode backward); one level row. */
int sfvos_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean,
                         const float* running_var, float eps, int C, float* mean, float* rstd, float* scale,
                         float* shift, sfvos_stream_t stream);

/* running = (1-momentum)*running + momentum*batch, applied for `n_updates` consecutive level rows
 * (means + l*coef_stride) in order: the reference updates once per FPN level (model.py:156-159);
 * num_batches_tracked (device int64 scalar, may be NULL) += n_updates. */
int sfvos_bn_running_update(float* running_mean, float* running_var, const float* means, const float* vars_unbiased,
                            int n_updates, int coef_stride, int C, float momentum, int64_t* num_batches_tracked,
                            sfvos_stream_t stream);

/* The running-statistics update of sfvos_bn_running_update folded into the BN-apply launch (its first workgroup does
 * it; one launch less per layer in training): means / vars_unbiased are coefficient-table rows, level l at
 * + l*coef_stride (the coef_stride of the apply call), n_updates consecutive rows applied in order. */
typedef struct sfvos_bn_running {
  float* running_mean;
  float* running_var;
  const float* means;
  const float* vars_unbiased;
  int64_t* num_batches_tracked; /* device int64 scalar += n_updates; may be NULL */
  int n_updates;
  float momentum;
} sfvos_bn_running;

/* y[m][0..C) = act(x[m][0..C) * scale_l + shift_l), act = ReLU when relu != 0 (model.py:114,122,...).
 * running (may be NULL): also update the running statistics (see sfvos_bn_running). */
int sfvos_bn_apply(const void* x, int ld_x, void* y, int ld_y, int dtype, const sfvos_levels* lv, int C,
                   const float* scale, const float* shift, int coef_stride, int relu, const sfvos_bn_running* running,
                   sfvos_stream_t stream);

/* The same with an e4m3 result (BASELINE config 5: the activation becomes the e4m3 operand of the next 3x3 conv):
 * y[m][c] = sat_e4m3(act(x*scale_l + shift_l) * act_scale); x bf16 with pitch ld_x (elements), y bytes with pitch ld_y;
 * C a multiple of 16; sat_count (device int, may be NULL) += values beyond +-448 before saturation. */
int sfvos_bn_apply_fp8(const void* x, int ld_x, void* y, int ld_y, const sfvos_levels* lv, int C, const float* scale,
                       const float* shift, int coef_stride, int relu, float act_scale, int* sat_count,
                       const sfvos_bn_running* running, sfvos_stream_t stream);

/* Rows of partials sfvos_bn_bwd_reduce / _apply write for these levels (level-major). */
int sfvos_bn_bwd_rows(const sfvos_levels* lv);

/* Pass 1 of BN(+ReLU) backward: dz = dy * (relu ? (x*scale+shift > 0) : 1);
 * part[row] = (sum dz, sum dz * xhat), xhat = (x-mean)*rstd, [2][C] per row. */
int sfvos_bn_bwd_reduce(const void* dy, int ld_dy, const void* x, int ld_x, int dtype, const sfvos_levels* lv, int C,
                        const float* scale, const float* shift, const float* mean, const float* rstd, int coef_stride,
                        int relu, float* part, sfvos_stream_t stream);

/* Finish pass 1, all levels in parallel: per level the sums (sum_dz, sum_dzx: coefficient-table rows, level l at
 * + l*coef_stride) and the three coefficients of pass 2:  dx = A*dz + B*x + K.  train != 0: batch-stat backward; else
 * eval (A = gamma*rstd, B = K = 0). */
int sfvos_bn_bwd_finalize(const float* part, const sfvos_levels* lv, const float* gamma, const float* mean,
                          const float* rstd, int coef_stride, int C, int train, float* coefA, float* coefB,
                          float* coefK, float* sum_dz, float* sum_dzx, sfvos_stream_t stream);

/* Pass 2: dx[m][c] = A*dz + B*x + K (dz as in pass 1), stored as dtype with pitch ld_dx;
 * bias_part (may be NULL) receives per-block partial sums of dx: sfvos_bn_bwd_rows(lv) rows of [C];
 * dgamma / dbeta (may be NULL) (=|+=) the per-level sums of pass 1 added in level order:
 * dgamma = sum_l sum dz*xhat, dbeta = sum_l sum dz. */
int sfvos_bn_bwd_apply(const void* dy, int ld_dy, const void* x, int ld_x, void* dx, int ld_dx, int dtype,
                       const sfvos_levels* lv, int C, const float* scale, const float* shift, int coef_stride,
                       int relu, const float* coefA, const float* coefB, const float* coefK, float* bias_part,
                       const float* sum_dz, const float* sum_dzx, float* dgamma, float* dbeta, int accumulate,
                       sfvos_stream_t stream);

/* out[c] (=|+=) sum_rows part[row][c]   (bias gradient from sfvos_bn_bwd_apply partials). */
int sfvos_reduce_rows(const float* part, int rows, int C, float* out, int accumulate, sfvos_stream_t stream);

/* ---- optimiser (torch.optim.SGD as built at reference train.py:80) ---------------------- */

/* g' = g + wd*p ; buf = first_step ? g' : momentum*buf + g' ; p -= lr*buf   over n fp32 elements. */
int sfvos_sgd_step(float* p, const float* g, float* momentum_buf, int64_t n, float lr, float momentum,
                   float weight_decay, int first_step, sfvos_stream_t stream);

/* x[i] *= s  (gradient averaging after the data-parallel all-reduce). */
int sfvos_scale(float* x, int64_t n, float s, sfvos_stream_t stream);

/* dst[i] += src[i] over n elements of `dtype` (n a multiple of 8): gradient fan-in where two consumers read the same
 * frames -- the slow pathway's window of the input clip (model.py:242-248: its gradient adds into frames
 * [k, k+sp) of the fast clip's gradient). */
int sfvos_add_inplace(void* dst, const void* src, int dtype, int64_t n, sfvos_stream_t stream);

/* ---- stand-in training loss (bench / tests) ---------------------------------------------------
 * The reference's loss is the sum of torchvision's RoI-head losses (model.py:346-368; torchvision is absent here and
 * out of scope); the benchmark and the parity fixtures use  loss = sum_l mean((out_l - target_l)^2)  over the fused
 * maps out_l [B,256,H_l,W_l] fp32 (oracle/slowfast_ref.py proxy_loss).  Two launches: value, gradient. */
typedef struct sfvos_mse_table {
  int n;                                   /* tensors (FPN levels), 1..SFVOS_MAX_LEVELS */
  const float* out[SFVOS_MAX_LEVELS];      /* contiguous fp32, numel[l] elements */
  const float* target[SFVOS_MAX_LEVELS];
  float* grad[SFVOS_MAX_LEVELS];           /* written by sfvos_mse_loss_grad only (may be NULL for sfvos_mse_loss) */
  int64_t numel[SFVOS_MAX_LEVELS];
} sfvos_mse_table;

/* fp32 partial rows sfvos_mse_loss needs in `part` for this table. */
int sfvos_mse_loss_rows(const sfvos_mse_table* t);
/* loss[0] = sum_l (1/numel_l) sum_i (out_l[i] - target_l[i])^2, reduced in a fixed order (f64 accumulate). */
int sfvos_mse_loss(const sfvos_mse_table* t, float* part, float* loss, sfvos_stream_t stream);
/* grad_l[i] = upstream * 2 (out_l[i] - target_l[i]) / numel_l; upstream: device scalar (autograd's grad_output) or
 * NULL for 1. */
int sfvos_mse_loss_grad(const sfvos_mse_table* t, const float* upstream, sfvos_stream_t stream);

/* ---- mask branch behind RoIAlign (model.py:17-25 MaskRCNNPredictor(256, 256, 2), model.py:346-347) -----
 * torchvision's MaskRCNNHeads / MaskRCNNPredictor / maskrcnn_inference / paste_masks_in_image on RoI features
 * [N,256,14,14] (RoIAlign itself is torchvision, out of scope).  The four conv3x3 + ReLU blocks are sfvos_conv3d calls
 * (kt = 1, taps = 9, relu = 1, batch = N RoIs, one 14x14 "level"); the rest: */

/* ConvTranspose2d(Cin, Cout, 2, 2, 0) weight [Cin][Cout][2][2] fp32 -> packed image (4*Cin*Cout elements of dtype). */
int sfvos_pack_deconv2x2(const float* w, void* packed, int dtype, int c_in, int c_out, sfvos_stream_t stream);
/* y[n][2i+a][2j+b][co] = act(bias[co] + sum_ci x[n][i][j][ci] * w[ci][co][a][b]); x: NHWC [n][h][w][c_in], y: NHWC
 * [n][2h][2w][c_out], both `dtype`; act = ReLU when relu != 0 (conv5_mask + relu of MaskRCNNPredictor). */
int sfvos_deconv2x2_relu(const void* x, const void* w_packed, const float* bias, void* y, int dtype, int n, int h, int w,
                         int c_in, int c_out, int relu, sfvos_stream_t stream);
/* mask_fcn_logits (conv1x1 c -> num_classes, weight [num_classes][c] fp32) on x: NHWC [n][positions][c]:
 * logits (may be NULL): [n][num_classes][positions] fp32; prob (may be NULL): [n][1][positions] = sigmoid of the logit
 * of class labels[n] (maskrcnn_inference). */
int sfvos_mask_logits(const void* x, int dtype, const float* w, const float* bias, const int64_t* labels, int n,
                      int positions, int c, int num_classes, float* logits, float* prob, sfvos_stream_t stream);
/* paste_masks_in_image: masks [n][1][mask_size][mask_size] fp32, boxes [n][4] fp32 (x1,y1,x2,y2) -> out
 * [n][1][img_h][img_w] fp32: masks zero-padded by `padding`, boxes expanded by (mask_size + 2 padding) / mask_size
 * and truncated to integers, bilinear resize (align_corners = false) to the box, zero outside it. */
int sfvos_paste_masks(const float* masks, const float* boxes, int n, int mask_size, int padding, int img_h, int img_w,
                      float* out, sfvos_stream_t stream);

/* ---- MultiScaleRoIAlign in front of the mask branch (model.py:346: roi_heads' mask_roi_pool =
 * MultiScaleRoIAlign(['0','1','2','3'], output_size 14, sampling_ratio 2) on the fused maps SlowFastLayers returns).
 * rois: [n][5] fp32 (image index in the batch, x1, y1, x2, y2), image coordinates.  torchvision's arithmetic, restated
 * (parity unpinned by the reference: torchvision is neither vendored nor installed). ---- */

/* LevelMapper: levels[r] = clamp(floor(canonical_level + log2(sqrt(box area) / canonical_scale) + eps), k_min, k_max) - k_min
 * (device int32 per RoI; torchvision: canonical_scale 224, canonical_level 4, eps 1e-6, k_min / k_max = -log2 of the first /
 * last feature level's scale). */
int sfvos_roi_levels(const float* rois, int n, int k_min, int k_max, float canonical_scale, float canonical_level,
                     float eps, int* levels, sfvos_stream_t stream);

/* Bytes of `workspace` for n RoIs (per-RoI tables of the separable bilinear parameters). */
size_t sfvos_roi_align_workspace_bytes(int n);

/* torchvision.ops.roi_align(feat, rois, pooled, spatial_scale, sampling_ratio, aligned = False) for the RoIs r with
 * levels[r] == level (levels == NULL: all of them): feat [B][C][H][W] fp32 contiguous; out [n][C][pooled][pooled] fp32
 * (rows of other levels are left untouched); sampling_ratio >= 1 and pooled * sampling_ratio <= 64. */
int sfvos_roi_align(const float* feat, int B, int C, int H, int W, const float* rois, const int* levels, int level, int n,
                    float spatial_scale, int pooled, int sampling_ratio, void* workspace, float* out,
                    sfvos_stream_t stream);

/* Its backward: dfeat [B][C][H][W] (=|+=) the gradient of the level's RoIs (dout [n][C][pooled][pooled]); a gather in a
 * fixed order (deterministic; torchvision scatters with atomics).  n == 0 writes zeros (accumulate == 0). */
int sfvos_roi_align_bwd(const float* dout, int B, int C, int H, int W, const float* rois, const int* levels, int level,
                        int n, float spatial_scale, int pooled, int sampling_ratio, void* workspace, float* dfeat,
                        int accumulate, sfvos_stream_t stream);

/* ---- training side of the mask branch (the reference trains roi_heads: only backbone and RPN are frozen,
 * model.py:176-179; losses.backward() at model.py:369 runs through torchvision's maskrcnn_loss and these modules).
 * The 3x3 convs: sfvos_conv3d with a sfvos_pack_weights_dgrad image (data gradient) and sfvos_conv3d_wgrad. ---- */

/* ReLU backward + bias gradient partials: dz[m][c] = a[m][c] > 0 ? dy[m][c] : 0 (a == NULL: dz = dy; dz may alias dy or
 * be NULL); col_part (may be NULL): sfvos_relu_bwd_rows(m) rows of [c] fp32, row r = column sums of dz over its 64
 * positions (sum them with sfvos_reduce_rows: the bias gradient of the conv that produced a).  dy, a, dz: [m][c] dtype. */
int sfvos_relu_bwd_rows(int64_t m);
int sfvos_relu_bwd(const void* dy, const void* a, void* dz, int dtype, int64_t m, int c, float* col_part,
                   sfvos_stream_t stream);

/* torchvision roi_heads.maskrcnn_loss behind the RoIAlign of the ground-truth masks:
 * loss[0] = mean_{n,p} BCEWithLogits(logits[n][labels[n]][p], targets[n][p]); logits [n][num_classes][positions] fp32,
 * targets [n][positions] fp32, labels int64; reduced in a fixed order (f64). */
int sfvos_mask_bce_loss(const float* logits, const int64_t* labels, const float* targets, int n, int num_classes,
                        int positions, float* loss, sfvos_stream_t stream);
/* dlogits[n][k][p] = k == labels[n] ? upstream * (sigmoid(logit) - target) / (n * positions) : 0;
 * upstream: device scalar (autograd's grad_output) or NULL for 1. */
int sfvos_mask_bce_loss_grad(const float* logits, const int64_t* labels, const float* targets, const float* upstream,
                             int n, int num_classes, int positions, float* dlogits, sfvos_stream_t stream);

/* Backward of mask_fcn_logits (conv1x1 c -> num_classes) and of the ReLU in front of it.  y: the deconv's post-ReLU
 * output, NHWC [n][positions][c] dtype; dlogits [n][num_classes][positions] fp32; w [num_classes][c] fp32.
 * dz[n][p][c] = (relu == 0 || y > 0) ? sum_k dlogits[n][k][p] w[k][c] : 0   (dtype, same shape as y);
 * part: sfvos_mask_logits_bwd_rows(n, positions) rows of [num_classes*c + num_classes + c] fp32 =
 * (grad of w | grad of the logits bias | column sums of dz = grad of the deconv bias) partials (sfvos_reduce_rows). */
int sfvos_mask_logits_bwd_rows(int n, int positions);
int sfvos_mask_logits_bwd(const void* y, int dtype, const float* dlogits, const float* w, int n, int positions, int c,
                          int num_classes, int relu, void* dz, float* part, sfvos_stream_t stream);

/* ConvTranspose2d(Cin, Cout, 2, 2, 0) backward.  Data gradient: weight [Cin][Cout][2][2] fp32 -> packed image, then
 * dx[n][i][j][ci] = sum_{a,b,co} dz[n][2i+a][2j+b][co] w[ci][co][a][b]; dz NHWC [n][2h][2w][c_out], dx NHWC
 * [n][h][w][c_in], both dtype. */
int sfvos_pack_deconv2x2_dgrad(const float* w, void* packed, int dtype, int c_in, int c_out, sfvos_stream_t stream);
int sfvos_deconv2x2_dgrad(const void* dz, const void* w_packed, void* dx, int dtype, int n, int h, int w, int c_in,
                          int c_out, sfvos_stream_t stream);
/* Weight gradient: grad_w[ci][co][a][b] (fp32, state-dict layout) (=|+=) sum_{n,i,j} x[n][i][j][ci] dz[n][2i+a][2j+b][co]
 * (exact-f32 MFMA, fixed-order split reduction); workspace: sfvos_deconv2x2_wgrad_workspace_bytes. */
size_t sfvos_deconv2x2_wgrad_workspace_bytes(int n, int h, int w, int c_in, int c_out);
int sfvos_deconv2x2_wgrad(const void* x, const void* dz, int dtype, int n, int h, int w, int c_in, int c_out,
                          float* grad_w, int accumulate, void* workspace, sfvos_stream_t stream);

/* ---- evaluation-side reducer (reference code/helpers/davis_evaluate.py:40-42) ------------ */

/* out[i] = OR over the n predicted masks of (masks[k][i] >= threshold), i < hw: the per-frame union the reference
 * builds with numpy before writing the PNG.  masks: n contiguous fp32 planes of hw elements ([N,1,H,W]);
 * out: hw bytes (0/1).  n may be 0 (all zeros).  Comparison and OR only: bit-exact with the reference. */
int sfvos_mask_union(const float* masks, int n, int64_t hw, float threshold, unsigned char* out,
                     sfvos_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SFVOS_H */
