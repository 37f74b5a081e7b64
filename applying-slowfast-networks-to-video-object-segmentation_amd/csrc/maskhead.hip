// maskhead.hip -- the post-RoIAlign part of the Mask R-CNN mask branch the reference runs on the fused features
// (reference code/helpers/model.py:17-25 swaps in MaskRCNNPredictor(256, 256, 2), model.py:346-347
// calls roi_heads -> postprocess).  torchvision is third-party, un-vendored and absent here; the arithmetic is
// restated from its published modules (models/detection/mask_rcnn.py: MaskRCNNHeads, MaskRCNNPredictor;
// models/detection/roi_heads.py: maskrcnn_inference, expand_masks, expand_boxes, paste_mask_in_image):
//
//   RoI features [N,256,14,14] -> 4 x (conv3x3 256->256 + ReLU)        sfvos_conv3d (kt = 1, relu epilogue)
//                              -> ConvTranspose2d 2x2 stride 2 + ReLU   sfvos_deconv2x2_relu      (this file)
//                              -> conv1x1 256 -> num_classes, sigmoid, the box's label channel
//                                                                       sfvos_mask_logits         (this file)
//                              -> masks padded by 1, boxes expanded, bilinear resize to the box, paste into the image
//                                                                       sfvos_paste_masks         (this file)
//
// These tensors are tiny next to the SlowFast convs (1 GFLOP per RoI): the kernels here are plain MFMA / VALU code
// with operands straight from L2, not tuned pipelines.  The second half of the file is the training side (loss and
// the backward of these modules).
#include "elt_util.h"

namespace sfvos {

// ---- ConvTranspose2d(k = 2, stride 2, pad 0) + bias + ReLU -----------------------------------------------------
//   y[n][2i+a][2j+b][co] = relu(bias[co] + sum_ci x[n][i][j][ci] * w[ci][co][a][b])
// = one GEMM  Y[M = N*h*w][4*Cout] = X[M][Cin] * Wp[Cin][4*Cout]  with a scattering store.  Wave = 32 rows x 32 columns,
// K loop over Cin; packed weights wp[col = (a*2+b)*Cout + co][ci] (k-contiguous, like the conv images).
template <int DT>
__global__ __launch_bounds__(256) void deconv2x2_kernel(const char* __restrict__ x, const char* __restrict__ wp,
                                                        const float* __restrict__ bias, char* y, int M, int h, int w,
                                                        int c_in, int c_out, int relu) {
  typedef typename Elt<DT>::type T;
  constexpr int CE = Elt<DT>::CE, ES = 16 / CE;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const int row0 = blockIdx.x * 32, col0 = (blockIdx.y * 4 + wv) * 32;
  const int ncol = 4 * c_out;
  if (col0 >= ncol) return;
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  const int row = row0 + r < M ? row0 + r : M - 1;   // clamp: rows past M are computed and dropped
  const char* xa = x + (long long)row * c_in * ES;
  const char* wb = wp + (long long)(col0 + r) * c_in * ES;
  for (int k = 0; k < c_in; k += 2 * CE) {           // one MFMA step = chunks 2s + hh of A and B
    const u32x4 a = *(const u32x4*)(xa + (k + hh * CE) * ES);
    const u32x4 b = *(const u32x4*)(wb + (k + hh * CE) * ES);
    Mma<DT>::run(acc, a, b);
  }
  // C/D layout: column = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 hh
  const int col = col0 + r, ab = col / c_out, co = col - ab * c_out, da = ab >> 1, db = ab & 1;
  const float bv = bias ? bias[co] : 0.f;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int m = row0 + (e & 3) + 8 * (e >> 2) + 4 * hh;
    if (m < M) {
      const int n = m / (h * w), ij = m - n * (h * w), i = ij / w, j = ij - i * w;
      float v = acc[e] + bv;
      if (relu) v = fmaxf(v, 0.f);
      const long long pos = ((long long)n * 2 * h + 2 * i + da) * (2 * w) + 2 * j + db;
      ((T*)y)[pos * c_out + co] = Elt<DT>::from_f32(v);
    }
  }
}

// ConvTranspose2d weight [Cin][Cout][2][2] fp32 -> wp[(a*2+b)*Cout + co][ci]
template <int DT>
__global__ __launch_bounds__(256) void pack_deconv_kernel(const float* __restrict__ w, char* wp, int c_in, int c_out) {
  typedef typename Elt<DT>::type T;
  const long long total = 4ll * c_in * c_out;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int ci = (int)(i % c_in);
    const long long col = i / c_in;
    const int ab = (int)(col / c_out), co = (int)(col - (long long)ab * c_out);
    ((T*)wp)[i] = Elt<DT>::from_f32(w[((long long)ci * c_out + co) * 4 + ab]);
  }
}

// ---- mask_fcn_logits (conv1x1 C -> num_classes) + sigmoid + label select ---------------------------------------
// one wave per pixel: 64 lanes x C/64 channels, shuffle reduction.  logits [N][K][P] (optional, all classes),
// prob [N][1][P] = sigmoid(logit of class labels[n]) (maskrcnn_inference).
template <int DT>
__global__ __launch_bounds__(256) void mask_logits_kernel(const char* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ bias,
                                                          const long long* __restrict__ labels, int N, int P, int C,
                                                          int K, float* logits, float* prob) {
  typedef typename Elt<DT>::type T;
  const int lane = threadIdx.x & 63;
  const long long pix = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pix >= (long long)N * P) return;
  const int n = (int)(pix / P), p = (int)(pix - (long long)n * P);
  const T* xr = (const T*)x + pix * C;
  // a label outside [0, K) selects no channel: prob becomes NaN (torch raises an index error there; an
  // out-of-range label must never turn into an out-of-bounds access or an uninitialised result)
  const long long lab64 = labels ? labels[n] : 0;
  const int lab = (lab64 >= 0 && lab64 < K) ? (int)lab64 : -1;
  if (prob && lab < 0 && lane == 0) prob[(long long)n * P + p] = __builtin_nanf("");
  for (int k = 0; k < K; ++k) {
    if (!logits && k != lab) continue;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += Elt<DT>::to_f32(xr[c]) * w[(long long)k * C + c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    s += bias ? bias[k] : 0.f;
    if (lane == 0) {
      if (logits) logits[((long long)n * K + k) * P + p] = s;
      if (prob && k == lab) prob[(long long)n * P + p] = 1.f / (1.f + expf(-s));
    }
  }
}

// ---- paste_masks_in_image (roi_heads.py): masks zero-padded by `padding`, boxes expanded by (M + 2 pad) / M and
// truncated to integers, mask resized to the (w, h) = (x2 - x1 + 1, y2 - y1 + 1) box with bilinear interpolation
// (align_corners = False, torch's area_pixel source index: src = scale * (dst + 0.5) - 0.5 clamped at 0) and written
// into the image where the box overlaps it; zero elsewhere.
__global__ __launch_bounds__(256) void paste_masks_kernel(const float* __restrict__ masks, const float* __restrict__ boxes,
                                                          int N, int M, int padding, int H, int W, float* out) {
  const long long total = (long long)N * H * W;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int n = (int)(i / ((long long)H * W));
    const int yx = (int)(i - (long long)n * H * W), y = yx / W, xx = yx - y * W;
    const float* b = boxes + 4 * n;
    // expand_boxes: float arithmetic as torch does it, then .to(int64) (truncation toward zero)
    const float scale = (float)(M + 2 * padding) / (float)M;
    float w_half = (b[2] - b[0]) * 0.5f, h_half = (b[3] - b[1]) * 0.5f;
    const float x_c = (b[2] + b[0]) * 0.5f, y_c = (b[3] + b[1]) * 0.5f;
    w_half *= scale; h_half *= scale;
    const long long bx0 = (long long)(x_c - w_half), bx1 = (long long)(x_c + w_half);
    const long long by0 = (long long)(y_c - h_half), by1 = (long long)(y_c + h_half);
    long long bw = bx1 - bx0 + 1, bh = by1 - by0 + 1;
    bw = bw > 1 ? bw : 1; bh = bh > 1 ? bh : 1;
    const long long x_0 = bx0 > 0 ? bx0 : 0, x_1 = bx1 + 1 < W ? bx1 + 1 : W;
    const long long y_0 = by0 > 0 ? by0 : 0, y_1 = by1 + 1 < H ? by1 + 1 : H;
    float v = 0.f;
    if (xx >= x_0 && xx < x_1 && y >= y_0 && y < y_1) {
      const int S = M + 2 * padding;   // padded mask side
      const int dy = (int)(y - by0), dx = (int)(xx - bx0);   // position inside the resized mask
      if (dy < bh && dx < bw) {
        const float sy = (float)S / (float)bh, sx = (float)S / (float)bw;
        float fy = sy * ((float)dy + 0.5f) - 0.5f, fx = sx * ((float)dx + 0.5f) - 0.5f;
        fy = fy < 0.f ? 0.f : fy; fx = fx < 0.f ? 0.f : fx;
        const int y0i = (int)fy, x0i = (int)fx;
        const int y1i = y0i + (y0i < S - 1 ? 1 : 0), x1i = x0i + (x0i < S - 1 ? 1 : 0);
        const float ly = fy - (float)y0i, lx = fx - (float)x0i;
        auto at = [&](int py, int px) -> float {   // the zero-padded mask
          const int my = py - padding, mx = px - padding;
          return (my >= 0 && my < M && mx >= 0 && mx < M) ? masks[((long long)n * M + my) * M + mx] : 0.f;
        };
        v = (1.f - ly) * ((1.f - lx) * at(y0i, x0i) + lx * at(y0i, x1i)) +
            ly * ((1.f - lx) * at(y1i, x0i) + lx * at(y1i, x1i));
      }
    }
    out[i] = v;
  }
}


// =====================================================================================================================
// Training side of the mask branch (the reference trains roi_heads: only backbone and RPN are frozen, model.py:176-179):
// torchvision roi_heads.maskrcnn_loss = binary_cross_entropy_with_logits(mask_logits[arange(N), labels], targets)
// and the autograd of the modules above.  The 3x3 convs use sfvos_conv3d (data gradient, dgrad-packed weights) and
// sfvos_conv3d_wgrad; the rest is here.  Everything is reduced in a fixed order (deterministic, no atomics).

// ---- ReLU backward + column sums (bias gradient of the conv that produced `a`) -----------------------------------
//   dz[m][c] = a[m][c] > 0 ? dy[m][c] : 0 (a == NULL: dz = dy);  part[block][c] = sum over the block's rows of dz
constexpr int RB_ROWS = 64;  // rows per block
template <int DT>
// (dy is not __restrict__: the mask branch calls this in place, dz == dy; every element is read before it is written
// by the same thread)
__global__ __launch_bounds__(256) void relu_bwd_kernel(const char* dy, const char* __restrict__ a, char* dz,
                                                       long long M, int C, float* part) {
  constexpr int CE = Elt<DT>::CE, ES = 16 / CE;
  __shared__ float red[256][8];
  const int cpr = C / CE;              // 16-byte chunks per row (<= 64)
  const int rl = 256 / cpr;            // row lanes
  const int ch = threadIdx.x % cpr, rowl = threadIdx.x / cpr;
  const long long m0 = (long long)blockIdx.x * RB_ROWS;
  float acc[CE];
#pragma unroll
  for (int e = 0; e < CE; ++e) acc[e] = 0.f;
  if (rowl < rl)
    for (int i = rowl; i < RB_ROWS && m0 + i < M; i += rl) {
      const long long o = ((m0 + i) * C + ch * CE) * ES;
      float g[CE], v[CE];
      unpack<DT>(*(const u32x4*)(dy + o), g);
      if (a) {
        unpack<DT>(*(const u32x4*)(a + o), v);
#pragma unroll
        for (int e = 0; e < CE; ++e) g[e] = v[e] > 0.f ? g[e] : 0.f;
      }
#pragma unroll
      for (int e = 0; e < CE; ++e) acc[e] += g[e];
      if (dz) *(u32x4*)(dz + o) = pack<DT>(g);
    }
  if (part == nullptr) return;
#pragma unroll
  for (int e = 0; e < CE; ++e) red[threadIdx.x][e] = acc[e];
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    const int chunk = c / CE, e = c - chunk * CE;
    float t = 0.f;
    for (int k = 0; k < rl; ++k) t += red[k * cpr + chunk][e];
    part[(long long)blockIdx.x * C + c] = t;
  }
}

// ---- maskrcnn_loss: BCE with logits on the label's channel, mean over N x P ---------------------------------------
//   loss = mean( max(z, 0) - z t + log(1 + exp(-|z|)) ),  z = logits[n][labels[n]][p], t = targets[n][p]
__global__ __launch_bounds__(1024) void mask_bce_loss_kernel(const float* __restrict__ logits,
                                                             const long long* __restrict__ labels,
                                                             const float* __restrict__ targets, int N, int K, int P,
                                                             float* loss) {
  __shared__ double red[1024];
  const long long total = (long long)N * P;
  double s = 0.0;
  for (long long i = threadIdx.x; i < total; i += 1024) {
    const int n = (int)(i / P), p = (int)(i - (long long)n * P);
    const long long lab = labels[n];
    if (lab < 0 || lab >= K) { s += (double)__builtin_nanf(""); continue; }  // invalid label: the loss is NaN, nothing is read
    const float z = logits[((long long)n * K + (int)lab) * P + p], t = targets[i];
    s += (double)(fmaxf(z, 0.f) - z * t + log1pf(expf(-fabsf(z))));
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss[0] = (float)(red[0] / (double)total);
}

//   dlogits[n][k][p] = k == labels[n] ? upstream * (sigmoid(z) - t) / (N P) : 0
__global__ __launch_bounds__(256) void mask_bce_grad_kernel(const float* __restrict__ logits,
                                                            const long long* __restrict__ labels,
                                                            const float* __restrict__ targets,
                                                            const float* __restrict__ upstream, int N, int K, int P,
                                                            float* dlogits) {
  const long long total = (long long)N * K * P;
  const float up = (upstream ? upstream[0] : 1.f) / (float)((long long)N * P);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int p = (int)(i % P);
    const long long nk = i / P;
    const int k = (int)(nk % K), n = (int)(nk / K);
    float g = 0.f;
    if ((long long)k == labels[n]) {   // an out-of-range label matches no channel: zero gradient
      const float z = logits[i];
      g = up * (1.f / (1.f + expf(-z)) - targets[(long long)n * P + p]);
    }
    dlogits[i] = g;
  }
}

// ---- backward of mask_fcn_logits (conv1x1 C -> K) and of the ReLU in front of it ----------------------------------
//   dz[n][p][c] = y[n][p][c] > 0 ? sum_k dl[n][k][p] w[k][c] : 0          (gradient w.r.t. the deconv's pre-ReLU output)
//   part[block] = [ sum dl[k] y[c]  (K x C) | sum dl[k]  (K) | sum dz[c]  (C) ]   over the block's MLB_PIX pixels
// wave = 16 pixels, lane = channels lane, lane + 64, ...
constexpr int MLB_PIX = 64, MLB_MAXK = 8, MLB_MAXQ = 8;
template <int DT>
__global__ __launch_bounds__(256) void mask_logits_bwd_kernel(const char* __restrict__ y, const float* __restrict__ dl,
                                                              const float* __restrict__ w, int N, int P, int C, int K,
                                                              int relu, char* dz, float* part) {
  typedef typename Elt<DT>::type T;
  extern __shared__ float red[];  // [4 waves][K*C + K + C]
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int Q = C / 64, L = K * C + K + C;
  float aw[MLB_MAXK][MLB_MAXQ], ab[MLB_MAXK], cs[MLB_MAXQ], wr[MLB_MAXK][MLB_MAXQ];
#pragma unroll
  for (int k = 0; k < MLB_MAXK; ++k) {
    ab[k] = 0.f;
#pragma unroll
    for (int q = 0; q < MLB_MAXQ; ++q) {
      aw[k][q] = 0.f;
      wr[k][q] = (k < K && q < Q) ? w[(long long)k * C + lane + 64 * q] : 0.f;
    }
  }
#pragma unroll
  for (int q = 0; q < MLB_MAXQ; ++q) cs[q] = 0.f;
  const long long total = (long long)N * P;
  const long long pix0 = (long long)blockIdx.x * MLB_PIX + wv * (MLB_PIX / 4);
  for (int i = 0; i < MLB_PIX / 4; ++i) {
    const long long pix = pix0 + i;
    if (pix >= total) break;  // wave-uniform
    const int n = (int)(pix / P), p = (int)(pix - (long long)n * P);
    float g[MLB_MAXK];
#pragma unroll
    for (int k = 0; k < MLB_MAXK; ++k) g[k] = k < K ? dl[((long long)n * K + k) * P + p] : 0.f;
#pragma unroll
    for (int k = 0; k < MLB_MAXK; ++k) ab[k] += g[k];
#pragma unroll
    for (int q = 0; q < MLB_MAXQ; ++q) {
      if (q >= Q) break;
      const long long o = pix * C + lane + 64 * q;
      const float yv = Elt<DT>::to_f32(((const T*)y)[o]);
      float d = 0.f;
#pragma unroll
      for (int k = 0; k < MLB_MAXK; ++k) {
        aw[k][q] += g[k] * yv;
        d += g[k] * wr[k][q];
      }
      if (relu && !(yv > 0.f)) d = 0.f;
      cs[q] += d;
      ((T*)dz)[o] = Elt<DT>::from_f32(d);
    }
  }
  float* mine = red + wv * L;
#pragma unroll
  for (int k = 0; k < MLB_MAXK; ++k) {
    if (k >= K) break;
#pragma unroll
    for (int q = 0; q < MLB_MAXQ; ++q)
      if (q < Q) mine[k * C + lane + 64 * q] = aw[k][q];
    if (lane == 0) mine[K * C + k] = ab[k];
  }
#pragma unroll
  for (int q = 0; q < MLB_MAXQ; ++q)
    if (q < Q) mine[K * C + K + lane + 64 * q] = cs[q];
  __syncthreads();
  for (int i = threadIdx.x; i < L; i += 256)
    part[(long long)blockIdx.x * L + i] = (red[i] + red[L + i]) + (red[2 * L + i] + red[3 * L + i]);
}

// ---- ConvTranspose2d(2, 2, 0) backward ------------------------------------------------------------------------------
// data gradient:  dx[n][i][j][ci] = sum_{a,b,co} dz[n][2i+a][2j+b][co] w[ci][co][a][b]
// = GEMM dX[M][Cin] = G[M][4 Cout] * Wd[4 Cout][Cin], the rows of G gathered from the four sub-positions of a pixel;
// packed image wd[ci][(a*2+b)*Cout + co] (k-contiguous).  Wave = 32 rows x 32 columns as in the forward kernel.
template <int DT>
__global__ __launch_bounds__(256) void deconv2x2_dgrad_kernel(const char* __restrict__ dz, const char* __restrict__ wd,
                                                              char* dx, int M, int h, int w, int c_in, int c_out) {
  typedef typename Elt<DT>::type T;
  constexpr int CE = Elt<DT>::CE, ES = 16 / CE;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const int row0 = blockIdx.x * 32, col0 = (blockIdx.y * 4 + wv) * 32;
  if (col0 >= c_in) return;
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  const int row = row0 + r < M ? row0 + r : M - 1;   // clamp: rows past M are computed and dropped
  const int n = row / (h * w), ij = row - n * (h * w), i = ij / w, j = ij - i * w;
  const char* wb = wd + (long long)(col0 + r) * 4 * c_out * ES;
  for (int ab = 0; ab < 4; ++ab) {
    const long long pos = ((long long)n * 2 * h + 2 * i + (ab >> 1)) * (2 * w) + 2 * j + (ab & 1);
    const char* ga = dz + pos * c_out * ES;
    for (int k = 0; k < c_out; k += 2 * CE) {
      const u32x4 a = *(const u32x4*)(ga + (k + hh * CE) * ES);
      const u32x4 b = *(const u32x4*)(wb + (ab * c_out + k + hh * CE) * ES);
      Mma<DT>::run(acc, a, b);
    }
  }
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int m = row0 + (e & 3) + 8 * (e >> 2) + 4 * hh;
    if (m < M) ((T*)dx)[(long long)m * c_in + col0 + r] = Elt<DT>::from_f32(acc[e]);
  }
}

// ConvTranspose2d weight [Cin][Cout][2][2] fp32 -> wd[ci][(a*2+b)*Cout + co]
template <int DT>
__global__ __launch_bounds__(256) void pack_deconv_dgrad_kernel(const float* __restrict__ w, char* wd, int c_in, int c_out) {
  typedef typename Elt<DT>::type T;
  const long long total = 4ll * c_in * c_out;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int k = (int)(i % (4 * c_out)), ci = (int)(i / (4 * c_out));
    const int ab = k / c_out, co = k - ab * c_out;
    ((T*)wd)[i] = Elt<DT>::from_f32(w[((long long)ci * c_out + co) * 4 + ab]);
  }
}

// weight gradient:  dw[ci][co][a][b] = sum_{n,i,j} x[n][i][j][ci] dz[n][2i+a][2j+b][co]
// The reduction runs over positions, the slow index of both operands: the exact-f32 MFMA (32x32x2: one k per lane half)
// takes one element per lane, 32 lanes = 32 consecutive channels (coalesced), so no transpose is needed; bf16 operands
// are widened on load (products exact).  Wave = 32 ci x 32 co of one sub-position (a, b) over one split of the
// positions; WG_STEPS k-pairs are loaded ahead of their MFMAs.  part[split][(ci*Cout + co)*4 + ab].
constexpr int DWG_UN = 8;
template <int DT>
__global__ __launch_bounds__(256) void deconv2x2_wgrad_kernel(const char* __restrict__ x, const char* __restrict__ dz,
                                                              float* part, int M, int h, int w, int c_in, int c_out,
                                                              int per_split) {
  typedef typename Elt<DT>::type T;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const int tiles_co = c_out / 32;
  const int tile = blockIdx.x * 4 + wv;                  // (ci tile, ab, co tile)
  if (tile >= (c_in / 32) * 4 * tiles_co) return;
  const int ct = tile % tiles_co, ab = (tile / tiles_co) & 3, it = tile / (4 * tiles_co);
  const int m_begin = blockIdx.y * per_split, m_end = min(M, m_begin + per_split);
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  const T* xa = (const T*)x + it * 32 + r;
  const T* gb = (const T*)dz + ct * 32 + r;
  for (int m0 = m_begin; m0 < m_end; m0 += 2 * DWG_UN) {
    float av[DWG_UN], bv[DWG_UN];
#pragma unroll
    for (int u = 0; u < DWG_UN; ++u) {
      const int m = m0 + 2 * u + hh;
      const bool ok = m < m_end;
      const int mc = ok ? m : m_end - 1;
      const int n = mc / (h * w), ij = mc - n * (h * w), i = ij / w, j = ij - i * w;
      const long long pos = ((long long)n * 2 * h + 2 * i + (ab >> 1)) * (2 * w) + 2 * j + (ab & 1);
      const float a = Elt<DT>::to_f32(xa[(long long)mc * c_in]), b = Elt<DT>::to_f32(gb[pos * c_out]);
      av[u] = ok ? a : 0.f;
      bv[u] = b;
    }
#pragma unroll
    for (int u = 0; u < DWG_UN; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], acc, 0, 0, 0);
  }
  float* out = part + (long long)blockIdx.y * c_in * c_out * 4;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int ci = it * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh, co = ct * 32 + r;
    out[((long long)ci * c_out + co) * 4 + ab] = acc[e];
  }
}
}  // namespace sfvos

using namespace sfvos;

extern "C" int sfvos_pack_deconv2x2(const float* w, void* packed, int dtype, int c_in, int c_out,
                                    sfvos_stream_t stream) {
  SFVOS_REQUIRE(w && packed && c_in > 0 && c_out > 0 && c_in % 32 == 0 && c_out % 32 == 0,
                "pack_deconv2x2: channels must be positive multiples of 32");
  const unsigned grid = grid_for(4ll * c_in * c_out, 256);
  hipStream_t s = (hipStream_t)stream;
  DT_DISPATCH(dtype,
              hipLaunchKernelGGL(pack_deconv_kernel<SFVOS_F32>, dim3(grid), dim3(256), 0, s, w, (char*)packed, c_in, c_out),
              hipLaunchKernelGGL(pack_deconv_kernel<SFVOS_BF16>, dim3(grid), dim3(256), 0, s, w, (char*)packed, c_in, c_out));
  return check_launch("pack_deconv2x2");
}

extern "C" int sfvos_deconv2x2_relu(const void* x, const void* w_packed, const float* bias, void* y, int dtype, int n,
                                    int h, int w, int c_in, int c_out, int relu, sfvos_stream_t stream) {
  SFVOS_REQUIRE(x && w_packed && y && n > 0 && h > 0 && w > 0, "deconv2x2: bad argument");
  SFVOS_REQUIRE(c_in % 32 == 0 && c_out % 32 == 0 && c_in > 0 && c_out > 0, "deconv2x2: channels must be multiples of 32");
  const long long M = (long long)n * h * w;
  SFVOS_REQUIRE(M < (1ll << 31), "deconv2x2: too many positions");
  dim3 grid((unsigned)ceil_div64(M, 32), (unsigned)ceil_div(4 * c_out, 128));
  hipStream_t s = (hipStream_t)stream;
  DT_DISPATCH(dtype,
              hipLaunchKernelGGL(deconv2x2_kernel<SFVOS_F32>, grid, dim3(256), 0, s, (const char*)x, (const char*)w_packed,
                                 bias, (char*)y, (int)M, h, w, c_in, c_out, relu),
              hipLaunchKernelGGL(deconv2x2_kernel<SFVOS_BF16>, grid, dim3(256), 0, s, (const char*)x, (const char*)w_packed,
                                 bias, (char*)y, (int)M, h, w, c_in, c_out, relu));
  return check_launch("deconv2x2");
}

extern "C" int sfvos_mask_logits(const void* x, int dtype, const float* w, const float* bias, const int64_t* labels,
                                 int n, int positions, int c, int num_classes, float* logits, float* prob,
                                 sfvos_stream_t stream) {
  SFVOS_REQUIRE(x && w && n > 0 && positions > 0 && c > 0 && num_classes > 0 && (logits || prob),
                "mask_logits: bad argument");
  SFVOS_REQUIRE(!prob || labels || num_classes == 1, "mask_logits: prob needs the labels of the boxes");
  const long long pix = (long long)n * positions;
  hipStream_t s = (hipStream_t)stream;
  DT_DISPATCH(dtype,
              hipLaunchKernelGGL(mask_logits_kernel<SFVOS_F32>, dim3((unsigned)ceil_div64(pix, 4)), dim3(256), 0, s,
                                 (const char*)x, w, bias, (const long long*)labels, n, positions, c, num_classes, logits, prob),
              hipLaunchKernelGGL(mask_logits_kernel<SFVOS_BF16>, dim3((unsigned)ceil_div64(pix, 4)), dim3(256), 0, s,
                                 (const char*)x, w, bias, (const long long*)labels, n, positions, c, num_classes, logits, prob));
  return check_launch("mask_logits");
}

extern "C" int sfvos_paste_masks(const float* masks, const float* boxes, int n, int mask_size, int padding, int img_h,
                                 int img_w, float* out, sfvos_stream_t stream) {
  SFVOS_REQUIRE(masks && boxes && out && n > 0 && mask_size > 0 && padding >= 0 && img_h > 0 && img_w > 0,
                "paste_masks: bad argument");
  hipLaunchKernelGGL(paste_masks_kernel, dim3(grid_for((long long)n * img_h * img_w, 256 * 4)), dim3(256), 0,
                     (hipStream_t)stream, masks, boxes, n, mask_size, padding, img_h, img_w, out);
  return check_launch("paste_masks");
}

extern "C" int sfvos_relu_bwd_rows(int64_t m) { return m > 0 ? (int)ceil_div64(m, RB_ROWS) : 0; }

extern "C" int sfvos_relu_bwd(const void* dy, const void* a, void* dz, int dtype, int64_t m, int c, float* col_part,
                              sfvos_stream_t stream) {
  SFVOS_REQUIRE(dy && m > 0 && (dz || col_part), "relu_bwd: bad argument");
  const int ce = dtype == SFVOS_BF16 ? 8 : 4;
  SFVOS_REQUIRE(c > 0 && c % ce == 0 && c / ce <= 64 && 256 % (c / ce) == 0,
                "relu_bwd: C must be a power-of-two number (<= 64) of 16-byte chunks");
  const unsigned grid = (unsigned)ceil_div64(m, RB_ROWS);
  hipStream_t s = (hipStream_t)stream;
  DT_DISPATCH(dtype,
              hipLaunchKernelGGL(relu_bwd_kernel<SFVOS_F32>, dim3(grid), dim3(256), 0, s, (const char*)dy, (const char*)a,
                                 (char*)dz, (long long)m, c, col_part),
              hipLaunchKernelGGL(relu_bwd_kernel<SFVOS_BF16>, dim3(grid), dim3(256), 0, s, (const char*)dy, (const char*)a,
                                 (char*)dz, (long long)m, c, col_part));
  return check_launch("relu_bwd");
}

extern "C" int sfvos_mask_bce_loss(const float* logits, const int64_t* labels, const float* targets, int n,
                                   int num_classes, int positions, float* loss, sfvos_stream_t stream) {
  SFVOS_REQUIRE(logits && labels && targets && loss && n > 0 && num_classes > 0 && positions > 0,
                "mask_bce_loss: bad argument");
  hipLaunchKernelGGL(mask_bce_loss_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, logits,
                     (const long long*)labels, targets, n, num_classes, positions, loss);
  return check_launch("mask_bce_loss");
}

extern "C" int sfvos_mask_bce_loss_grad(const float* logits, const int64_t* labels, const float* targets,
                                        const float* upstream, int n, int num_classes, int positions, float* dlogits,
                                        sfvos_stream_t stream) {
  SFVOS_REQUIRE(logits && labels && targets && dlogits && n > 0 && num_classes > 0 && positions > 0,
                "mask_bce_loss_grad: bad argument");
  hipLaunchKernelGGL(mask_bce_grad_kernel, dim3(grid_for((long long)n * num_classes * positions, 256 * 4)), dim3(256), 0,
                     (hipStream_t)stream, logits, (const long long*)labels, targets, upstream, n, num_classes, positions,
                     dlogits);
  return check_launch("mask_bce_loss_grad");
}

extern "C" int sfvos_mask_logits_bwd_rows(int n, int positions) {
  return n > 0 && positions > 0 ? (int)ceil_div64((long long)n * positions, MLB_PIX) : 0;
}

extern "C" int sfvos_mask_logits_bwd(const void* y, int dtype, const float* dlogits, const float* w, int n, int positions,
                                     int c, int num_classes, int relu, void* dz, float* part, sfvos_stream_t stream) {
  SFVOS_REQUIRE(y && dlogits && w && dz && part && n > 0 && positions > 0, "mask_logits_bwd: bad argument");
  SFVOS_REQUIRE(c > 0 && c % 64 == 0 && c / 64 <= MLB_MAXQ && num_classes > 0 && num_classes <= MLB_MAXK,
                "mask_logits_bwd: C must be a multiple of 64 (<= %d), num_classes <= %d", 64 * MLB_MAXQ, MLB_MAXK);
  const unsigned grid = (unsigned)ceil_div64((long long)n * positions, MLB_PIX);
  const int lds = 4 * (num_classes * c + num_classes + c) * (int)sizeof(float);
  hipStream_t s = (hipStream_t)stream;
  DT_DISPATCH(dtype,
              hipLaunchKernelGGL(mask_logits_bwd_kernel<SFVOS_F32>, dim3(grid), dim3(256), lds, s, (const char*)y, dlogits,
                                 w, n, positions, c, num_classes, relu, (char*)dz, part),
              hipLaunchKernelGGL(mask_logits_bwd_kernel<SFVOS_BF16>, dim3(grid), dim3(256), lds, s, (const char*)y, dlogits,
                                 w, n, positions, c, num_classes, relu, (char*)dz, part));
  return check_launch("mask_logits_bwd");
}

extern "C" int sfvos_pack_deconv2x2_dgrad(const float* w, void* packed, int dtype, int c_in, int c_out,
                                          sfvos_stream_t stream) {
  SFVOS_REQUIRE(w && packed && c_in > 0 && c_out > 0 && c_in % 32 == 0 && c_out % 32 == 0,
                "pack_deconv2x2_dgrad: channels must be positive multiples of 32");
  const unsigned grid = grid_for(4ll * c_in * c_out, 256);
  hipStream_t s = (hipStream_t)stream;
  DT_DISPATCH(dtype,
              hipLaunchKernelGGL(pack_deconv_dgrad_kernel<SFVOS_F32>, dim3(grid), dim3(256), 0, s, w, (char*)packed, c_in, c_out),
              hipLaunchKernelGGL(pack_deconv_dgrad_kernel<SFVOS_BF16>, dim3(grid), dim3(256), 0, s, w, (char*)packed, c_in, c_out));
  return check_launch("pack_deconv2x2_dgrad");
}

extern "C" int sfvos_deconv2x2_dgrad(const void* dz, const void* w_packed, void* dx, int dtype, int n, int h, int w,
                                     int c_in, int c_out, sfvos_stream_t stream) {
  SFVOS_REQUIRE(dz && w_packed && dx && n > 0 && h > 0 && w > 0, "deconv2x2_dgrad: bad argument");
  SFVOS_REQUIRE(c_in % 32 == 0 && c_out % 32 == 0 && c_in > 0 && c_out > 0, "deconv2x2_dgrad: channels must be multiples of 32");
  const long long M = (long long)n * h * w;
  SFVOS_REQUIRE(M < (1ll << 29), "deconv2x2_dgrad: too many positions");
  dim3 grid((unsigned)ceil_div64(M, 32), (unsigned)ceil_div(c_in, 128));
  hipStream_t s = (hipStream_t)stream;
  DT_DISPATCH(dtype,
              hipLaunchKernelGGL(deconv2x2_dgrad_kernel<SFVOS_F32>, grid, dim3(256), 0, s, (const char*)dz,
                                 (const char*)w_packed, (char*)dx, (int)M, h, w, c_in, c_out),
              hipLaunchKernelGGL(deconv2x2_dgrad_kernel<SFVOS_BF16>, grid, dim3(256), 0, s, (const char*)dz,
                                 (const char*)w_packed, (char*)dx, (int)M, h, w, c_in, c_out));
  return check_launch("deconv2x2_dgrad");
}

static int deconv_wgrad_splits(long long M) {
  long long s = ceil_div64(M, 256);  // >= 256 positions per split
  return (int)(s < 1 ? 1 : s > 64 ? 64 : s);
}

extern "C" size_t sfvos_deconv2x2_wgrad_workspace_bytes(int n, int h, int w, int c_in, int c_out) {
  if (n <= 0 || h <= 0 || w <= 0 || c_in <= 0 || c_out <= 0) return 0;
  return (size_t)deconv_wgrad_splits((long long)n * h * w) * c_in * c_out * 4 * sizeof(float);
}

extern "C" int sfvos_deconv2x2_wgrad(const void* x, const void* dz, int dtype, int n, int h, int w, int c_in, int c_out,
                                     float* grad_w, int accumulate, void* workspace, sfvos_stream_t stream) {
  SFVOS_REQUIRE(x && dz && grad_w && workspace && n > 0 && h > 0 && w > 0, "deconv2x2_wgrad: bad argument");
  SFVOS_REQUIRE(c_in % 32 == 0 && c_out % 32 == 0 && c_in > 0 && c_out > 0, "deconv2x2_wgrad: channels must be multiples of 32");
  const long long M = (long long)n * h * w;
  SFVOS_REQUIRE(M < (1ll << 29), "deconv2x2_wgrad: too many positions");
  const int splits = deconv_wgrad_splits(M);
  int per_split = (int)ceil_div64(M, splits);
  per_split = (per_split + 1) & ~1;  // whole k-pairs
  const int tiles = (c_in / 32) * 4 * (c_out / 32);
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((unsigned)ceil_div(tiles, 4), (unsigned)splits);
  DT_DISPATCH(dtype,
              hipLaunchKernelGGL(deconv2x2_wgrad_kernel<SFVOS_F32>, grid, dim3(256), 0, s, (const char*)x, (const char*)dz,
                                 (float*)workspace, (int)M, h, w, c_in, c_out, per_split),
              hipLaunchKernelGGL(deconv2x2_wgrad_kernel<SFVOS_BF16>, grid, dim3(256), 0, s, (const char*)x, (const char*)dz,
                                 (float*)workspace, (int)M, h, w, c_in, c_out, per_split));
  int rc = check_launch("deconv2x2_wgrad");
  if (rc) return rc;
  return sfvos_reduce_rows((const float*)workspace, splits, c_in * c_out * 4, grad_w, accumulate, stream);
}
