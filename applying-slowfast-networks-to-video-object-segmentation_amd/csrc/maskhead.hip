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
// with operands straight from L2, not tuned pipelines.
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
  const int lab = labels ? (int)labels[n] : 0;
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
