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
 * Activation layout ("NDHWC"): element (t,h,w,c) of a clip lives at
 * base + ((t*H + h)*W + w)*ld + c, `ld` >= C being the per-position pitch in
 * ELEMENTS, so a producer can write into a channel slice of a wider buffer
 * (this is how torch.cat at model.py:115,162 is eliminated).
 */
#ifndef SFVOS_H
#define SFVOS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* sfvos_stream_t; /* hipStream_t */

enum { SFVOS_F32 = 0, SFVOS_BF16 = 1 };

enum {
  SFVOS_OK = 0,
  SFVOS_E_ARG = -1,     /* bad argument / unsupported shape */
  SFVOS_E_LAUNCH = -2,  /* HIP launch error */
  SFVOS_E_NODEV = -3    /* no gfx950 device */
};

int sfvos_version(void);
const char* sfvos_last_error(void);
/* 0 when the current HIP device is a gfx950; SFVOS_E_NODEV otherwise. */
int sfvos_check_device(void);

/* ---- layout -------------------------------------------------------------------------- */

/* Frames fp32 [T][C][H][W] addressed through explicit ELEMENT strides (the reference's
 * stack().transpose(1,2) view is non-contiguous, model.py:157-158) -> NDHWC dst[T][H][W][ld].
 * Never writes the source. */
int sfvos_frames_to_ndhwc(const float* src, int64_t stride_t, int64_t stride_c, int64_t stride_h,
                          int64_t stride_w, void* dst, int dtype, int T, int C, int H, int W, int ld,
                          sfvos_stream_t stream);

/* NDHWC src[M][ld] (first C channels) -> planar fp32 dst[C][M]  (the final
 * cat(...).squeeze(2) of model.py:162, emitted as the caller's NCHW tensor). */
int sfvos_ndhwc_to_planar(const void* src, int dtype, float* dst, int64_t M, int C, int ld,
                          sfvos_stream_t stream);

/* planar fp32 src[C][M] (e.g. the incoming NCHW gradient) -> NDHWC dst[M][ld]. */
int sfvos_planar_to_ndhwc(const float* src, void* dst, int dtype, int64_t M, int C, int ld,
                          sfvos_stream_t stream);

/* NDHWC src[T][H][W][ld] (first C channels) accumulated (+=) or stored into fp32 frames
 * [T][C][H][W] with explicit strides: the input gradient of model.py:157-158, needed only when
 * the caller's features require grad (reference code/osvos/osvos_model.py:50,64). */
int sfvos_ndhwc_to_frames(const void* src, int dtype, float* dst, int64_t stride_t, int64_t stride_c,
                          int64_t stride_h, int64_t stride_w, int T, int C, int H, int W, int ld,
                          int accumulate, sfvos_stream_t stream);

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

/* ---- convolution (replaces aten::convolution at model.py:112,120,124,132,136,144,147) ---- */

typedef struct sfvos_conv_desc {
  int dtype;       /* SFVOS_F32: f32 storage, exact-f32 MFMA; SFVOS_BF16: bf16 storage, f32 accumulate */
  int batch;       /* clips; x/y advance by x_batch_stride / y_batch_stride ELEMENTS */
  int t_in, h, w;  /* input frames, spatial extent (stride 1, output spatial == input spatial) */
  int c_in, c_out; /* multiples of 32 */
  int kt;          /* temporal taps */
  int taps;        /* 9 = 3x3 spatial, zero pad 1 ; 1 = 1x1 spatial, no pad */
  int pad_t;       /* zero frames each side in time: 0 (forward), kt-1 (data gradient) */
  int ld_x, ld_y;  /* per-position pitch of x / y in elements */
  int accumulate;  /* y += conv(x) instead of y = conv(x)  (gradient fan-in) */
  int64_t x_batch_stride, y_batch_stride;
} sfvos_conv_desc;

/* Number of [2][c_out] fp32 partial-statistics rows sfvos_conv3d writes for this desc
 * (one per workgroup tile; summed in fixed order by sfvos_bn_finalize -> deterministic). */
int sfvos_conv3d_stat_rows(const sfvos_conv_desc* d);

/* y[b][t][h][w][0..c_out) (=|+=) bias + sum_{dt,dh,dw,c} x[b][t+dt-pad_t][h+dh-1][w+dw-1][c] * W.
 * t_out = t_in + 2*pad_t - kt + 1.  bias may be NULL.  stat_part may be NULL; otherwise it
 * receives per-tile partial (sum, sum of squares) of the values written, over valid positions.
 * `zeros` is a caller-provided, zero-filled device buffer of >= 256 bytes (padding source). */
int sfvos_conv3d(const sfvos_conv_desc* d, const void* x, const void* w_packed, const float* bias, void* y,
                 float* stat_part, const void* zeros, sfvos_stream_t stream);

/* ---- weight gradient (aten::convolution_backward grad_weight / grad_bias) -------------- */

/* Workspace bytes for sfvos_conv3d_wgrad (fp32 split-K slabs). `d` describes the FORWARD conv. */
size_t sfvos_conv3d_wgrad_workspace_bytes(const sfvos_conv_desc* d);

/* grad_w[Cout][Cin][kt][kh][kw] (fp32, state-dict layout) (=|+=) sum_pos dy[pos][n] * x[pos+shift][c].
 * x is the forward input (NDHWC, ld_x), dy the gradient w.r.t. the conv output (NDHWC, ld_y).
 * accumulate != 0 adds into grad_w (model.py:369-374 accumulates two clips before stepping). */
int sfvos_conv3d_wgrad(const sfvos_conv_desc* d, const void* x, const void* dy, float* grad_w, int accumulate,
                       void* workspace, const void* zeros, sfvos_stream_t stream);

/* ---- batch norm (aten::native_batch_norm / _backward at model.py:113,121,...,148) ------- */

/* Reduce `rows` partial rows part[rows][2][C] in fixed order; train statistics over `count`
 * values per channel: mean, biased var -> rstd = 1/sqrt(var+eps), scale = gamma*rstd,
 * shift = beta - mean*scale.  save_mean/save_var_unbiased feed sfvos_bn_running_update. */
int sfvos_bn_finalize(const float* part, int rows, int64_t count, const float* gamma, const float* beta, float eps,
                      int C, float* mean, float* rstd, float* scale, float* shift, float* save_var_unbiased,
                      sfvos_stream_t stream);

/* Eval mode: scale = gamma / sqrt(running_var + eps), shift = beta - running_mean * scale. */
int sfvos_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean,
                         const float* running_var, float eps, int C, float* scale, float* shift,
                         sfvos_stream_t stream);

/* running = (1-momentum)*running + momentum*batch, applied for `n_updates` consecutive
 * (mean, unbiased var) rows in order (the reference updates once per FPN level, model.py:156-159). */
int sfvos_bn_running_update(float* running_mean, float* running_var, const float* means, const float* vars_unbiased,
                            int n_updates, int C, float momentum, sfvos_stream_t stream);

/* y[m][0..C) = act(x[m][0..C) * scale + shift), act = ReLU when relu != 0 (model.py:114,122,...). */
int sfvos_bn_apply(const void* x, int ld_x, void* y, int ld_y, int dtype, int64_t M, int C, const float* scale,
                   const float* shift, int relu, sfvos_stream_t stream);

/* Rows of [2][C] partials sfvos_bn_bwd_reduce writes for M positions. */
int sfvos_bn_bwd_rows(int64_t M);

/* Pass 1 of BN(+ReLU) backward: dz = dy * (relu ? (x*scale+shift > 0) : 1);
 * part[row] = (sum dz, sum dz * xhat), xhat = (x-mean)*rstd. */
int sfvos_bn_bwd_reduce(const void* dy, int ld_dy, const void* x, int ld_x, int dtype, int64_t M, int C,
                        const float* scale, const float* shift, const float* mean, const float* rstd, int relu,
                        float* part, sfvos_stream_t stream);

/* Finish pass 1: dgamma (+)= sum dz*xhat, dbeta (+)= sum dz, and the three per-channel
 * coefficients of pass 2:  dx = A*dz + B*x + K.  train != 0: batch-stat backward; else eval
 * (dx = dz * gamma * rstd_running, i.e. A = scale, B = K = 0). */
int sfvos_bn_bwd_finalize(const float* part, int rows, int64_t count, const float* gamma, const float* mean,
                          const float* rstd, int C, int train, int accumulate, float* dgamma, float* dbeta,
                          float* coefA, float* coefB, float* coefK, sfvos_stream_t stream);

/* Pass 2: dx[m][c] = A*dz + B*x + K (dz as in pass 1), stored as dtype with pitch ld_dx;
 * bias_part (may be NULL) receives per-block partial sums of dx: rows = sfvos_bn_bwd_rows(M), [C] each. */
int sfvos_bn_bwd_apply(const void* dy, int ld_dy, const void* x, int ld_x, void* dx, int ld_dx, int dtype, int64_t M,
                       int C, const float* scale, const float* shift, int relu, const float* coefA,
                       const float* coefB, const float* coefK, float* bias_part, sfvos_stream_t stream);

/* out[c] (=|+=) sum_rows part[row][c]   (bias gradient from sfvos_bn_bwd_apply partials). */
int sfvos_reduce_rows(const float* part, int rows, int C, float* out, int accumulate, sfvos_stream_t stream);

/* ---- optimiser (torch.optim.SGD as built at reference train.py:80) ---------------------- */

/* g' = g + wd*p ; buf = first_step ? g' : momentum*buf + g' ; p -= lr*buf   over n fp32 elements. */
int sfvos_sgd_step(float* p, const float* g, float* momentum_buf, int64_t n, float lr, float momentum,
                   float weight_decay, int first_step, sfvos_stream_t stream);

/* x[i] *= s  (gradient averaging after the data-parallel all-reduce). */
int sfvos_scale(float* x, int64_t n, float s, sfvos_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SFVOS_H */
