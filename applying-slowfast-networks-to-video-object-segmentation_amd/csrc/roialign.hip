// roialign.hip -- MultiScaleRoIAlign between the fused feature maps and the mask branch: reference
// code/helpers/model.py:346 hands `slow_fast_features` to torchvision's roi_heads, whose mask path starts with
// MultiScaleRoIAlign(featmap_names = ['0','1','2','3'], output_size = 14, sampling_ratio = 2): a level mapper
// (k = floor(4 + log2(sqrt(area) / 224) + 1e-6) clamped to the feature levels) and torchvision.ops.roi_align
// (aligned = False) per level.  torchvision is third-party, not vendored and not installed here: the arithmetic below
// restates its published roi_align CPU/CUDA kernels (bilinear_interpolate, roi_align_forward / _backward) --
// PARITY UNPINNED BY THE REFERENCE, checked against oracle/roi_align_ref.py.
//
//   forward : out[r][c][ph][pw] = (1 / (g g)) sum_{iy, ix < g} bilinear(feat[b_r][c], y(ph, iy), x(pw, ix)),  g = sampling_ratio
//   backward: torchvision scatters every sample's gradient onto its four corner pixels with atomicAdd (run-to-run
//             different sums).  Here it is a GATHER: a thread owns a feature pixel (8 channels of it) and adds, RoI by
//             RoI and sample by sample in a fixed order, the contributions that land on it -- deterministic, no atomics.
//   Both read per-RoI tables of the separable bilinear parameters (row / column of each of the P g samples per axis),
//   written by a small preparation kernel: the rules of bilinear_interpolate (samples more than one pixel outside the
//   map contribute nothing; clamping at 0 and at the last row / column) are applied once per sample, not per channel.
#include "common.h"

namespace sfvos {

constexpr int ROI_MAX_SAMPLES = 64;   // P * sampling_ratio per axis

struct RoiSample {   // one sample position along one axis
  int lo, hi;        // the two rows (columns) it interpolates between
  float wlo, whi;    // their weights; both 0: the sample lies outside the map (contributes nothing)
};

// tab[r][axis][k], k = p * g + i
__global__ __launch_bounds__(128) void roi_prep_kernel(const float* __restrict__ rois, const int* __restrict__ levels,
                                                       int level, int n, float scale, int P, int g, int H, int W,
                                                       RoiSample* tab) {
  const int r = blockIdx.x;
  if (r >= n || (levels && levels[r] != level)) return;
  const int K = P * g;
  const int axis = threadIdx.x / ROI_MAX_SAMPLES, k = threadIdx.x % ROI_MAX_SAMPLES;
  if (k >= K) return;
  const float* roi = rois + 5 * r;
  // aligned = false: no half-pixel offset, extents forced to at least one pixel (roi_align_forward_kernel_impl)
  const float start = (axis == 0 ? roi[2] : roi[1]) * scale, end = (axis == 0 ? roi[4] : roi[3]) * scale;
  const float extent = fmaxf(end - start, 1.f);
  const float bin = extent / (float)P;
  const int p = k / g, i = k - p * g;
  float y = start + p * bin + ((float)i + .5f) * bin / (float)g;
  const int L = axis == 0 ? H : W;
  RoiSample s;
  if (y < -1.f || y > (float)L) {
    // outside: weights 0.  lo / hi stay conservative bounds for the backward kernel's row / column range test
    s.lo = s.hi = y < -1.f ? 0 : L - 1; s.wlo = s.whi = 0.f;
  } else {
    if (y <= 0.f) y = 0.f;
    int lo = (int)y, hi;
    if (lo >= L - 1) { hi = lo = L - 1; y = (float)lo; } else { hi = lo + 1; }
    const float l = y - (float)lo;
    s.lo = lo; s.hi = hi; s.wlo = 1.f - l; s.whi = l;
  }
  tab[((long long)r * 2 + axis) * ROI_MAX_SAMPLES + k] = s;
}

// one thread per output element, index order (pw, ph, c, r) as torchvision's kernel
__global__ __launch_bounds__(256) void roi_align_fwd_kernel(const float* __restrict__ feat, int B, int C, int H, int W,
                                                            const float* __restrict__ rois,
                                                            const int* __restrict__ levels, int level, int n, int P,
                                                            int g, const RoiSample* __restrict__ tab, float* out) {
  const long long total = (long long)n * C * P * P;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int pw = (int)(idx % P), ph = (int)((idx / P) % P);
    const int c = (int)((idx / ((long long)P * P)) % C), r = (int)(idx / ((long long)P * P * C));
    if (levels && levels[r] != level) continue;
    const int b = (int)rois[5 * r];
    float sum = 0.f;
    if (b >= 0 && b < B) {
      const float* f = feat + ((long long)b * C + c) * H * W;
      const RoiSample* ty = tab + ((long long)r * 2 + 0) * ROI_MAX_SAMPLES + ph * g;
      const RoiSample* tx = tab + ((long long)r * 2 + 1) * ROI_MAX_SAMPLES + pw * g;
      for (int iy = 0; iy < g; ++iy) {
        const RoiSample sy = ty[iy];
        for (int ix = 0; ix < g; ++ix) {
          const RoiSample sx = tx[ix];
          if ((sy.wlo == 0.f && sy.whi == 0.f) || (sx.wlo == 0.f && sx.whi == 0.f)) continue;  // outside: adds 0
          const float w1 = sy.wlo * sx.wlo, w2 = sy.wlo * sx.whi, w3 = sy.whi * sx.wlo, w4 = sy.whi * sx.whi;
          sum += w1 * f[sy.lo * W + sx.lo] + w2 * f[sy.lo * W + sx.hi] + w3 * f[sy.hi * W + sx.lo] +
                 w4 * f[sy.hi * W + sx.hi];
        }
      }
    }
    const float count = (float)(g * g > 1 ? g * g : 1);
    out[idx] = sum / count;
  }
}

// dfeat[b][c][h][w] (=|+=) sum over the RoIs of this level on image b, in index order, over their samples in (ky, kx)
// order, of dout[r][c][ky / g][kx / g] * wy * wx / (g g).  Thread = (b, h, w) x CG channels (blockIdx.y).
constexpr int ROI_CG = 8;
__global__ __launch_bounds__(256) void roi_align_bwd_kernel(const float* __restrict__ dout, int B, int C, int H, int W,
                                                            const float* __restrict__ rois,
                                                            const int* __restrict__ levels, int level, int n, int P,
                                                            int g, const RoiSample* __restrict__ tab, float* dfeat,
                                                            int accumulate) {
  const long long pix = (long long)blockIdx.x * 256 + threadIdx.x;
  if (pix >= (long long)B * H * W) return;
  const int w = (int)(pix % W), h = (int)((pix / W) % H), b = (int)(pix / ((long long)W * H));
  const int c0 = blockIdx.y * ROI_CG;
  const int K = P * g;
  const float count = (float)(g * g > 1 ? g * g : 1);
  float acc[ROI_CG];
#pragma unroll
  for (int j = 0; j < ROI_CG; ++j) acc[j] = 0.f;
  for (int r = 0; r < n; ++r) {
    if ((levels && levels[r] != level) || (int)rois[5 * r] != b) continue;
    const RoiSample* ty = tab + ((long long)r * 2 + 0) * ROI_MAX_SAMPLES;
    const RoiSample* tx = ty + ROI_MAX_SAMPLES;
    // the samples are monotonic along an axis: rows / columns outside [first.lo, last.hi] get nothing from this RoI
    if (h < ty[0].lo || h > ty[K - 1].hi || w < tx[0].lo || w > tx[K - 1].hi) continue;
    for (int ky = 0; ky < K; ++ky) {
      const RoiSample sy = ty[ky];
      const float wy = (h == sy.lo ? sy.wlo : 0.f) + (h == sy.hi ? sy.whi : 0.f);
      if (wy == 0.f) continue;
      for (int kx = 0; kx < K; ++kx) {
        const RoiSample sx = tx[kx];
        const float wx = (w == sx.lo ? sx.wlo : 0.f) + (w == sx.hi ? sx.whi : 0.f);
        if (wx == 0.f) continue;
        const float wgt = wy * wx;
        const float* d = dout + (((long long)r * C + c0) * P + ky / g) * P + kx / g;
#pragma unroll
        for (int j = 0; j < ROI_CG; ++j)
          if (c0 + j < C) acc[j] += d[(long long)j * P * P] * wgt / count;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < ROI_CG; ++j)
    if (c0 + j < C) {
      float* dst = dfeat + (((long long)b * C + c0 + j) * H + h) * W + w;
      *dst = accumulate ? *dst + acc[j] : acc[j];
    }
}

// torchvision.ops.poolers.LevelMapper: levels[r] = clamp(floor(lvl0 + log2(sqrt(area) / s0) + eps), k_min, k_max) - k_min
__global__ __launch_bounds__(256) void roi_levels_kernel(const float* __restrict__ rois, int n, int k_min, int k_max,
                                                         float s0, float lvl0, float eps, int* levels) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= n) return;
  const float* b = rois + 5 * r + 1;
  const float s = sqrtf((b[2] - b[0]) * (b[3] - b[1]));
  float t = floorf(lvl0 + log2f(s / s0) + eps);
  t = fminf(fmaxf(t, (float)k_min), (float)k_max);   // NaN (negative area) -> k_min, as clamp leaves nothing out of range
  levels[r] = (t == t ? (int)t : k_min) - k_min;
}

static int roi_check(int B, int C, int H, int W, int n, int pooled, int sr, const char* what) {
  SFVOS_REQUIRE(B >= 1 && C >= 1 && H >= 1 && W >= 1 && n >= 0, "%s: bad extent", what);
  SFVOS_REQUIRE(pooled >= 1 && sr >= 1 && pooled * sr <= ROI_MAX_SAMPLES,
                "%s: output_size %d x sampling_ratio %d must be positive with a product <= %d (adaptive sampling, "
                "sampling_ratio <= 0, is not implemented: the reference's mask pooler uses 2)", what, pooled, sr,
                ROI_MAX_SAMPLES);
  SFVOS_REQUIRE((long long)n * C * pooled * pooled < (1ll << 40) && (long long)B * H * W < (1ll << 31), "%s: too large", what);
  return SFVOS_OK;
}

}  // namespace sfvos

using namespace sfvos;

extern "C" size_t sfvos_roi_align_workspace_bytes(int n) {
  return (size_t)(n > 0 ? n : 1) * 2 * ROI_MAX_SAMPLES * sizeof(RoiSample);
}

extern "C" int sfvos_roi_levels(const float* rois, int n, int k_min, int k_max, float canonical_scale,
                                float canonical_level, float eps, int* levels, sfvos_stream_t stream) {
  SFVOS_REQUIRE(n >= 0 && k_min <= k_max, "roi_levels: bad arguments");
  if (n == 0) return SFVOS_OK;
  SFVOS_REQUIRE(rois && levels, "roi_levels: null pointer");
  hipLaunchKernelGGL(roi_levels_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, rois, n,
                     k_min, k_max, canonical_scale, canonical_level, eps, levels);
  return check_launch("roi_levels");
}

extern "C" int sfvos_roi_align(const float* feat, int B, int C, int H, int W, const float* rois, const int* levels,
                               int level, int n, float spatial_scale, int pooled, int sampling_ratio, void* workspace,
                               float* out, sfvos_stream_t stream) {
  if (int rc = roi_check(B, C, H, W, n, pooled, sampling_ratio, "roi_align")) return rc;
  if (n == 0) return SFVOS_OK;
  SFVOS_REQUIRE(feat && rois && workspace && out, "roi_align: null pointer");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(roi_prep_kernel, dim3((unsigned)n), dim3(2 * ROI_MAX_SAMPLES), 0, s, rois, levels, level, n,
                     spatial_scale, pooled, sampling_ratio, H, W, (RoiSample*)workspace);
  if (int rc = check_launch("roi_prep")) return rc;
  long long grid = ceil_div64((long long)n * C * pooled * pooled, 256);
  if (grid > 65536) grid = 65536;
  hipLaunchKernelGGL(roi_align_fwd_kernel, dim3((unsigned)grid), dim3(256), 0, s, feat, B, C, H, W, rois, levels, level, n,
                     pooled, sampling_ratio, (const RoiSample*)workspace, out);
  return check_launch("roi_align");
}

extern "C" int sfvos_roi_align_bwd(const float* dout, int B, int C, int H, int W, const float* rois, const int* levels,
                                   int level, int n, float spatial_scale, int pooled, int sampling_ratio,
                                   void* workspace, float* dfeat, int accumulate, sfvos_stream_t stream) {
  if (int rc = roi_check(B, C, H, W, n, pooled, sampling_ratio, "roi_align_bwd")) return rc;
  SFVOS_REQUIRE(dfeat && (n == 0 || (dout && rois && workspace)), "roi_align_bwd: null pointer");
  hipStream_t s = (hipStream_t)stream;
  if (n > 0) {
    hipLaunchKernelGGL(roi_prep_kernel, dim3((unsigned)n), dim3(2 * ROI_MAX_SAMPLES), 0, s, rois, levels, level, n,
                       spatial_scale, pooled, sampling_ratio, H, W, (RoiSample*)workspace);
    if (int rc = check_launch("roi_prep")) return rc;
  }
  // n == 0: the kernel writes zeros (or leaves an accumulating buffer as it is)
  const dim3 grid((unsigned)ceil_div64((long long)B * H * W, 256), (unsigned)ceil_div(C, ROI_CG));
  hipLaunchKernelGGL(roi_align_bwd_kernel, grid, dim3(256), 0, s, dout, B, C, H, W, rois, levels, level, n, pooled,
                     sampling_ratio, (const RoiSample*)workspace, dfeat, accumulate);
  return check_launch("roi_align_bwd");
}
