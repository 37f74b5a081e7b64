// elementwise.hip -- layout passes (reference model.py:157-158 stack/transpose, :162 cat/squeeze
// and their backward), weight packing, row reduction (bias gradient), SGD (train.py:80).
// NDHWC side: 16 bytes per lane per access.
#include "elt_util.h"

namespace sfvos {

// ---- layout: planar/strided fp32 <-> NDHWC ------------------------------------------------------
// tile = 64 positions x 64 channels through LDS; planar side coalesced along positions,
// NDHWC side 16 B per lane along channels.
struct PlanarView {
  long long st, sc, sh, sw;  // element strides of the fp32 planar tensor
  int T, C, H, W;
};

// destination element (position, c) at (c / gdiv) * gstride + position * ld + c % gdiv: plain NDHWC is gdiv = C_max
// (one group), the channel-group-major input layout is gdiv = ld = 32, gstride = positions * 32
// (bodies take the block coordinates as arguments: the *_multi kernels below run them for every level of a pyramid in
// one launch)
template <int DT>
__device__ __forceinline__ void to_ndhwc_body(const float* __restrict__ src, const PlanarView& v, char* dst, int ld,
                                              int gdiv, long long gstride, float qscale, int* sat_count, int bx, int by,
                                              int bz) {
  constexpr int CE = Elt<DT>::CE;
  __shared__ float tile[64][65];
  const int tid = threadIdx.x;
  const long long HW = (long long)v.H * v.W;
  const long long p0 = (long long)bx * 64;
  const int c0 = by * 64, t = bz;
  {
    const int px = tid & 63, cy = tid >> 6;
    const long long p = p0 + px;
    if (p < HW) {
      const int h = (int)(p / v.W), w = (int)(p - (long long)h * v.W);
      const float* s = src + t * v.st + h * v.sh + w * v.sw;
#pragma unroll 4
      for (int c = cy; c < 64; c += 4)
        if (c0 + c < v.C) tile[c][px] = s[(long long)(c0 + c) * v.sc];
    }
  }
  __syncthreads();
  constexpr int CPR = 64 / CE;  // chunks per tile row
  int sat = 0;
  for (int i = tid; i < 64 * CPR; i += 256) {
    const int px = i / CPR, ch = i - px * CPR;
    const long long p = p0 + px;
    if (p < HW && c0 + ch * CE < v.C) {
      float f[CE];
#pragma unroll
      for (int e = 0; e < CE; ++e) {
        f[e] = DT == SFVOS_FP8 ? tile[ch * CE + e][px] * qscale : tile[ch * CE + e][px];
        if (DT == SFVOS_FP8) sat += fabsf(f[e]) > 448.f ? 1 : 0;
      }
      const int c = c0 + ch * CE;
      *(u32x4*)(dst + ((c / gdiv) * gstride + ((long long)t * HW + p) * ld + c % gdiv) * (16 / CE)) = pack<DT>(f);
    }
  }
  if (DT == SFVOS_FP8 && sat_count != nullptr) {  // integer count: order-independent
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sat += __shfl_xor(sat, o);
    if ((tid & 63) == 0 && sat > 0) atomicAdd(sat_count, sat);
  }
}

template <int DT>
__global__ __launch_bounds__(256) void to_ndhwc_kernel(const float* __restrict__ src, PlanarView v, char* dst, int ld,
                                                       int gdiv, long long gstride, float qscale = 1.f,
                                                       int* sat_count = nullptr) {
  to_ndhwc_body<DT>(src, v, dst, ld, gdiv, gstride, qscale, sat_count, blockIdx.x, blockIdx.y, blockIdx.z);
}

// max |src| over strided fp32 frames (e4m3 activation-scale calibration); non-negative floats order like their bits
__global__ __launch_bounds__(256) void frames_absmax_kernel(const float* __restrict__ src, PlanarView v, float* amax) {
  const long long HW = (long long)v.H * v.W;
  const long long total = (long long)v.T * v.C * HW;
  float m = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    long long k = i;
    const long long p = k % HW; k /= HW;
    const int c = (int)(k % v.C);
    const int t = (int)(k / v.C);
    const int h = (int)(p / v.W), w = (int)(p - (long long)h * v.W);
    m = fmaxf(m, fabsf(src[t * v.st + c * v.sc + h * v.sh + w * v.sw]));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) atomicMax((unsigned*)amax, __builtin_bit_cast(unsigned, m));
}

// dst += src, 16 bytes per lane
template <int DT>
__global__ __launch_bounds__(256) void add_inplace_kernel(char* dst, const char* __restrict__ src, long long chunks) {
  constexpr int CE = Elt<DT>::CE;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < chunks; i += (long long)gridDim.x * 256) {
    float a[CE], b[CE];
    unpack<DT>(*(const u32x4*)(dst + i * 16), a);
    unpack<DT>(*(const u32x4*)(src + i * 16), b);
#pragma unroll
    for (int e = 0; e < CE; ++e) a[e] += b[e];
    *(u32x4*)(dst + i * 16) = pack<DT>(a);
  }
}

// ---- stand-in loss: sum_l mean((out_l - target_l)^2) (oracle/slowfast_ref.py proxy_loss) -------------
constexpr int MSE_ROWS_PER_LEVEL = 256;   // blocks (= partial rows) per tensor
struct MseTab {
  int n;
  const float* out[SFVOS_MAX_LEVELS]; const float* target[SFVOS_MAX_LEVELS]; float* grad[SFVOS_MAX_LEVELS];
  long long numel[SFVOS_MAX_LEVELS];
};

// grid = n * MSE_ROWS_PER_LEVEL; block b of level l sums a strided set of 1024-element runs of it in a fixed order
// (four runs in flight per trip: the loads of a trip are issued before the first use)
__global__ __launch_bounds__(256) void mse_partial_kernel(MseTab t, float* part) {
  __shared__ double red[256];
  const int l = blockIdx.x / MSE_ROWS_PER_LEVEL, b = blockIdx.x % MSE_ROWS_PER_LEVEL;
  const float* o = t.out[l];
  const float* g = t.target[l];
  const long long n = t.numel[l];
  const bool vec = (((size_t)o | (size_t)g) & 15) == 0;
  const long long stride = (long long)MSE_ROWS_PER_LEVEL * 1024;
  double s = 0.0;
  constexpr int UN = 4;
  for (long long i0 = ((long long)b * 256 + threadIdx.x) * 4; i0 < n; i0 += UN * stride) {
    f32x4 a[UN], c[UN];
    bool full[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const long long i = i0 + u * stride;
      full[u] = vec && i + 4 <= n;
      if (full[u]) { a[u] = *(const f32x4*)(o + i); c[u] = *(const f32x4*)(g + i); }
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const long long i = i0 + u * stride;
      if (full[u]) {
        float q = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = a[u][e] - c[u][e]; q += d * d; }
        s += (double)q;
      } else {
        for (long long k = i; k < n && k < i + 4; ++k) { const float d = o[k] - g[k]; s += (double)(d * d); }
      }
    }
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.x] = (float)(red[0] / (double)n);
}

// fixed-order sum of the partial rows: thread k adds rows k, k+256, ...; then a tree over the 256 threads
__global__ __launch_bounds__(256) void mse_final_kernel(const float* part, int rows, float* loss) {
  __shared__ double red[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < rows; i += 256) s += (double)part[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss[0] = (float)red[0];
}

__global__ __launch_bounds__(256) void mse_grad_kernel(MseTab t, const float* upstream) {
  const int l = blockIdx.y;
  const float* o = t.out[l];
  const float* g = t.target[l];
  float* d = t.grad[l];
  const long long n = t.numel[l];
  const float k = (upstream ? upstream[0] : 1.f) * 2.f / (float)n;
  for (long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (long long)gridDim.x * 1024) {
    if (i + 4 <= n && (((size_t)(o + i) | (size_t)(g + i) | (size_t)(d + i)) & 15) == 0) {
      const f32x4 a = *(const f32x4*)(o + i), c = *(const f32x4*)(g + i);
      f32x4 r;
#pragma unroll
      for (int e = 0; e < 4; ++e) r[e] = k * (a[e] - c[e]);
      *(f32x4*)(d + i) = r;
    } else {
      for (long long q = i; q < n && q < i + 4; ++q) d[q] = k * (o[q] - g[q]);
    }
  }
}

// bf16 fast path of the same pass for contiguous fp32 planes (stride_w = 1, stride_h = W, 16-byte aligned planes):
// tile = 128 positions x 64 channels.  Load: 16 bytes per lane along the positions (a wave reads two 512-byte runs of
// two channels); store: four consecutive lanes write the four 16-byte chunks of one position's 64-byte channel group,
// i.e. a wave writes 1 KiB runs.  LDS rows of 129 floats: the store phase reads 8 positions x 4 chunks per half-wave
// on 32 different banks.  (2.3 -> ~4.5 TB/s on the 2.8 GB fp32 -> 1.4 GB bf16 clip of the drop-in API.)
__device__ __forceinline__ void to_ndhwc_bf16_vec_body(const float* __restrict__ src, const PlanarView& v, char* dst,
                                                       int ld, int gdiv, long long gstride, int bx, int by, int bz) {
  constexpr int TP = 128, RS = 129;
  __shared__ float tile[64 * RS];
  const int tid = threadIdx.x;
  const long long HW = (long long)v.H * v.W;
  const long long p0 = (long long)bx * TP;
  const int c0 = by * 64, t = bz;
  f32x4 reg[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int idx = k * 256 + tid, c = idx >> 5, q = idx & 31;
    const long long p = p0 + 4 * q;
    f32x4 r = {0.f, 0.f, 0.f, 0.f};
    if (c0 + c < v.C && p < HW) r = *(const f32x4*)(src + t * v.st + (long long)(c0 + c) * v.sc + p);  // HW % 4 == 0
    reg[k] = r;
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int idx = k * 256 + tid, c = idx >> 5, q = idx & 31;
#pragma unroll
    for (int e = 0; e < 4; ++e) tile[c * RS + 4 * q + e] = reg[k][e];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int idx = k * 256 + tid;
    const int ch = idx & 3, pos = (idx >> 2) & (TP - 1), grp = idx >> 9;   // grp: 32-channel group of the tile (0/1)
    const long long p = p0 + pos;
    const int c = c0 + grp * 32 + ch * 8;
    if (p < HW && c < v.C) {
      float f[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = tile[(grp * 32 + ch * 8 + e) * RS + pos];
      *(u32x4*)(dst + ((c / gdiv) * gstride + ((long long)t * HW + p) * ld + c % gdiv) * 2) = pack<SFVOS_BF16>(f);
    }
  }
}

__global__ __launch_bounds__(256) void to_ndhwc_bf16_vec_kernel(const float* __restrict__ src, PlanarView v, char* dst,
                                                                int ld, int gdiv, long long gstride) {
  to_ndhwc_bf16_vec_body(src, v, dst, ld, gdiv, gstride, blockIdx.x, blockIdx.y, blockIdx.z);
}

static bool planes_vectorizable(const float* src, const PlanarView& v) {
  const long long HW = (long long)v.H * v.W;
  return v.sw == 1 && v.sh == v.W && HW % 4 == 0 && v.st % 4 == 0 && v.sc % 4 == 0 && ((size_t)src & 15) == 0 &&
         v.C % 8 == 0;
}

template <int DT>
__device__ __forceinline__ void from_ndhwc_body(const char* __restrict__ src, int ld, float* dst, const PlanarView& v,
                                                int accumulate, int bx, int by, int bz) {
  constexpr int CE = Elt<DT>::CE;
  __shared__ float tile[64][65];
  const int tid = threadIdx.x;
  const long long HW = (long long)v.H * v.W;
  const long long p0 = (long long)bx * 64;
  const int c0 = by * 64, t = bz;
  constexpr int CPR = 64 / CE;
  for (int i = tid; i < 64 * CPR; i += 256) {
    const int px = i / CPR, ch = i - px * CPR;
    const long long p = p0 + px;
    if (p < HW && c0 + ch * CE < v.C) {
      float f[CE];
      unpack<DT>(*(const u32x4*)(src + (((long long)t * HW + p) * ld + c0 + ch * CE) * (16 / CE)), f);
#pragma unroll
      for (int e = 0; e < CE; ++e) tile[ch * CE + e][px] = f[e];
    }
  }
  __syncthreads();
  const int px = tid & 63, cy = tid >> 6;
  const long long p = p0 + px;
  if (p < HW) {
    const int h = (int)(p / v.W), w = (int)(p - (long long)h * v.W);
    float* d = dst + t * v.st + h * v.sh + w * v.sw;
#pragma unroll 4
    for (int c = cy; c < 64; c += 4)
      if (c0 + c < v.C) {
        float* q = d + (long long)(c0 + c) * v.sc;
        *q = accumulate ? *q + tile[c][px] : tile[c][px];
      }
  }
}

template <int DT>
__global__ __launch_bounds__(256) void from_ndhwc_kernel(const char* __restrict__ src, int ld, float* dst, PlanarView v,
                                                         int accumulate) {
  from_ndhwc_body<DT>(src, ld, dst, v, accumulate, blockIdx.x, blockIdx.y, blockIdx.z);
}

// ---- the same passes for EVERY level of a pyramid in one launch (the module's output / output-gradient hand-over:
// five launches of a few microseconds each become one).  blockIdx.x enumerates the pixel tiles of all levels.
struct LevelViews {
  int n;
  PlanarView v[SFVOS_MAX_LEVELS];
  float* planar[SFVOS_MAX_LEVELS];
  long long pos[SFVOS_MAX_LEVELS];      // first position of the level in the pyramid buffer
  int blk_begin[SFVOS_MAX_LEVELS + 1];
};

__device__ __forceinline__ int level_of_block(const LevelViews& t, int bx) {
  int l = 0;
#pragma unroll
  for (int k = 1; k < SFVOS_MAX_LEVELS; ++k)
    if (k < t.n && bx >= t.blk_begin[k]) l = k;
  return l;
}

template <int DT>
__global__ __launch_bounds__(256) void from_ndhwc_multi_kernel(const char* __restrict__ src, int ld, LevelViews t,
                                                               int accumulate) {
  const int l = level_of_block(t, blockIdx.x);
  from_ndhwc_body<DT>(src + t.pos[l] * ld * (16 / Elt<DT>::CE), ld, t.planar[l], t.v[l], accumulate,
                      (int)blockIdx.x - t.blk_begin[l], blockIdx.y, blockIdx.z);
}

template <int DT>
__global__ __launch_bounds__(256) void to_ndhwc_multi_kernel(LevelViews t, char* dst, int ld) {
  const int l = level_of_block(t, blockIdx.x);
  to_ndhwc_body<DT>(t.planar[l], t.v[l], dst + t.pos[l] * ld * (16 / Elt<DT>::CE), ld, 1 << 30, 0ll, 1.f, nullptr,
                    (int)blockIdx.x - t.blk_begin[l], blockIdx.y, blockIdx.z);
}

__global__ __launch_bounds__(256) void to_ndhwc_bf16_vec_multi_kernel(LevelViews t, char* dst, int ld) {
  const int l = level_of_block(t, blockIdx.x);
  to_ndhwc_bf16_vec_body(t.planar[l], t.v[l], dst + t.pos[l] * ld * 2, ld, 1 << 30, 0ll,
                         (int)blockIdx.x - t.blk_begin[l], blockIdx.y, blockIdx.z);
}

// ---- evaluation: union of thresholded masks (davis_evaluate.py:40-42) ------------------------------
__global__ __launch_bounds__(256) void mask_union_kernel(const float* __restrict__ masks, int n, long long hw, float thr,
                                                         unsigned char* __restrict__ out) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < hw; i += (long long)gridDim.x * 256) {
    unsigned char any = 0;
    for (int k = 0; k < n; ++k) any |= (unsigned char)(masks[(long long)k * hw + i] >= thr);
    out[i] = any;
  }
}

// ---- e4m3 weight image (forward), per-output-channel scale ------------------------------------------
// block n: max |w[n][...]| -> weight scale 448/max; writes the [2][c_out] (bias, descale) rows
__global__ __launch_bounds__(256) void fp8_weight_scale_kernel(const float* __restrict__ w, const float* __restrict__ bias,
                                                               long long per_n, int c_out, float act_scale,
                                                               float* bias_descale, float* wscale) {
  __shared__ float red[256];
  const int n = blockIdx.x;
  float m = 0.f;
  for (long long i = threadIdx.x; i < per_n; i += 256) m = fmaxf(m, fabsf(w[(long long)n * per_n + i]));
  red[threadIdx.x] = m;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if ((int)threadIdx.x < k) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + k]);
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float ws = red[0] > 0.f ? 448.f / red[0] : 1.f;
    wscale[n] = ws;
    bias_descale[n] = bias ? bias[n] : 0.f;
    bias_descale[c_out + n] = 1.f / (act_scale * ws);
  }
}

// packed[cc][dt][tap][j][n][16]: c = cc*64 + j*16 + e (as pack_weights_kernel with CE = 16)
__global__ __launch_bounds__(256) void fp8_pack_weights_kernel(const float* __restrict__ w, const float* __restrict__ wscale,
                                                               char* packed, int c_out, int c_in, int kt, int taps) {
  const long long chunks = (long long)c_out * c_in * kt * taps / 16;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < chunks; i += (long long)gridDim.x * 256) {
    long long k = i;
    const int n = (int)(k % c_out); k /= c_out;
    const int j = (int)(k % 4); k /= 4;
    const int tap = (int)(k % taps); k /= taps;
    const int dt = (int)(k % kt); k /= kt;
    const int cc = (int)k;
    const float ws = wscale[n];
    float f[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int c = cc * 64 + j * 16 + e;
      f[e] = w[(((long long)n * c_in + c) * kt + dt) * taps + tap] * ws;
    }
    *(u32x4*)(packed + i * 16) = pack<SFVOS_FP8>(f);
  }
}

// ---- weight packing -----------------------------------------------------------------------------
// packed[cc][dt][tap][j][n][e]  (one 16-B chunk = CE reduction channels for one output channel n;
// a conv stage (cc, dt, tap group) is one contiguous block)
template <int DT>
__device__ __forceinline__ void pack_one(const float* __restrict__ w, char* packed, int c_out, int c_in, int kt, int taps,
                                         bool dgrad, long long i) {
  typedef typename Elt<DT>::type T;
  constexpr int CE = Elt<DT>::CE, CK = 4 * CE;
  // N = output channels of the conv this image feeds (its reduction channels are the other count)
  const int N = dgrad ? c_in : c_out;
  long long k = i;
  const int e = (int)(k % CE); k /= CE;
  const int n = (int)(k % N); k /= N;
  const int j = (int)(k % 4); k /= 4;
  const int tap = (int)(k % taps); k /= taps;
  const int dt = (int)(k % kt); k /= kt;
  const int cc = (int)k;
  const int c = cc * CK + j * CE + e;
  float val;
  if (dgrad)  // w[co = c][ci = n][kt-1-dt][taps-1-tap]
    val = w[(((long long)c * c_in + n) * kt + (kt - 1 - dt)) * taps + (taps - 1 - tap)];
  else  // w[co = n][ci = c][dt][tap]
    val = w[(((long long)n * c_in + c) * kt + dt) * taps + tap];
  ((T*)packed)[i] = Elt<DT>::from_f32(val);
}

template <int DT, bool DGRAD>
__global__ __launch_bounds__(256) void pack_weights_kernel(const float* __restrict__ w, char* packed, int c_out,
                                                           int c_in, int kt, int taps) {
  const long long total = (long long)c_out * c_in * kt * taps;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256)
    pack_one<DT>(w, packed, c_out, c_in, kt, taps, DGRAD, i);
}

// several images in ONE launch (all layers of a model after an optimiser step): block -> (item, 1024-element run)
struct PackTab {
  int n;
  const float* w[SFVOS_MAX_PACK_ITEMS];
  char* packed[SFVOS_MAX_PACK_ITEMS];
  int c_out[SFVOS_MAX_PACK_ITEMS], c_in[SFVOS_MAX_PACK_ITEMS], kt[SFVOS_MAX_PACK_ITEMS], taps[SFVOS_MAX_PACK_ITEMS];
  int dgrad[SFVOS_MAX_PACK_ITEMS];
  int blk_begin[SFVOS_MAX_PACK_ITEMS + 1];
};
constexpr int PACK_RUN = 1024;

template <int DT>
__global__ __launch_bounds__(256) void pack_weights_batch_kernel(PackTab t) {
  int it = 0;
  for (int k = 1; k < t.n; ++k)
    if ((int)blockIdx.x >= t.blk_begin[k]) it = k;
  const long long total = (long long)t.c_out[it] * t.c_in[it] * t.kt[it] * t.taps[it];
  const long long i0 = (long long)((int)blockIdx.x - t.blk_begin[it]) * PACK_RUN;
  for (int u = threadIdx.x; u < PACK_RUN; u += 256)
    if (i0 + u < total)
      pack_one<DT>(t.w[it], t.packed[it], t.c_out[it], t.c_in[it], t.kt[it], t.taps[it], t.dgrad[it] != 0, i0 + u);
}

__global__ __launch_bounds__(32 * RL) void reduce_rows_kernel(const float* part, int rows, int C, float* out,
                                                              int accumulate) {
  __shared__ double scratch[32 * RL];
  const int c_local = threadIdx.x & 31, sub = threadIdx.x >> 5, c = blockIdx.x * 32 + c_local;
  const double s = column_sum(part, rows, C, c < C ? c : 0, sub, scratch, c_local);
  if (sub == 0 && c < C) out[c] = (accumulate ? out[c] : 0.f) + (float)s;
}

// ---- optimiser ----------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sgd_kernel(float* p, const float* g, float* buf, long long n, float lr,
                                                  float momentum, float wd, int first) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float gg = g[i] + wd * p[i];
    const float b = first ? gg : momentum * buf[i] + gg;
    buf[i] = b;
    p[i] -= lr * b;
  }
}

__global__ __launch_bounds__(256) void scale_kernel(float* x, long long n, float s) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) x[i] *= s;
}

}  // namespace sfvos

using namespace sfvos;

extern "C" int sfvos_frames_to_ndhwc(const float* src, int64_t st, int64_t sc, int64_t sh, int64_t sw, void* dst,
                                     int dtype, int T, int C, int H, int W, int ld, sfvos_stream_t stream) {
  SFVOS_REQUIRE(src && dst && T > 0 && C > 0 && H > 0 && W > 0 && ld >= C, "frames_to_ndhwc: bad argument");
  SFVOS_REQUIRE(C % 8 == 0 && ld % 8 == 0, "frames_to_ndhwc: C and ld must be multiples of 8");
  PlanarView v{st, sc, sh, sw, T, C, H, W};
  dim3 grid((unsigned)ceil_div64((int64_t)H * W, 64), (unsigned)ceil_div(C, 64), (unsigned)T);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == SFVOS_BF16 && planes_vectorizable(src, v) && ((size_t)dst & 15) == 0) {
    dim3 gridv((unsigned)ceil_div64((int64_t)H * W, 128), (unsigned)ceil_div(C, 64), (unsigned)T);
    hipLaunchKernelGGL(to_ndhwc_bf16_vec_kernel, gridv, dim3(256), 0, s, src, v, (char*)dst, ld, 1 << 30, 0ll);
    return check_launch("frames_to_ndhwc");
  }
  DT_DISPATCH(dtype,
              hipLaunchKernelGGL(to_ndhwc_kernel<SFVOS_F32>, grid, dim3(256), 0, s, src, v, (char*)dst, ld, 1 << 30, 0ll),
              hipLaunchKernelGGL(to_ndhwc_kernel<SFVOS_BF16>, grid, dim3(256), 0, s, src, v, (char*)dst, ld, 1 << 30, 0ll));
  return check_launch("frames_to_ndhwc");
}

extern "C" int sfvos_frames_to_groups(const float* src, int64_t st, int64_t sc, int64_t sh, int64_t sw, void* dst,
                                      int dtype, int T, int C, int H, int W, int64_t group_stride,
                                      sfvos_stream_t stream) {
  SFVOS_REQUIRE(src && dst && T > 0 && C > 0 && H > 0 && W > 0, "frames_to_groups: bad argument");
  SFVOS_REQUIRE(dtype == SFVOS_BF16, "frames_to_groups: the channel-group-major layout is bf16 only");
  SFVOS_REQUIRE(C % 32 == 0 && group_stride >= (int64_t)T * H * W * 32 && group_stride % 8 == 0,
                "frames_to_groups: C must be a multiple of 32 and group_stride >= T*H*W*32");
  PlanarView v{st, sc, sh, sw, T, C, H, W};
  if (planes_vectorizable(src, v) && ((size_t)dst & 15) == 0) {
    dim3 gridv((unsigned)ceil_div64((int64_t)H * W, 128), (unsigned)ceil_div(C, 64), (unsigned)T);
    hipLaunchKernelGGL(to_ndhwc_bf16_vec_kernel, gridv, dim3(256), 0, (hipStream_t)stream, src, v, (char*)dst, 32, 32,
                       (long long)group_stride);
    return check_launch("frames_to_groups");
  }
  dim3 grid((unsigned)ceil_div64((int64_t)H * W, 64), (unsigned)ceil_div(C, 64), (unsigned)T);
  hipLaunchKernelGGL(to_ndhwc_kernel<SFVOS_BF16>, grid, dim3(256), 0, (hipStream_t)stream, src, v, (char*)dst, 32, 32,
                     (long long)group_stride);
  return check_launch("frames_to_groups");
}

extern "C" int sfvos_frames_to_groups_fp8(const float* src, int64_t st, int64_t sc, int64_t sh, int64_t sw, void* dst,
                                          int T, int C, int H, int W, int64_t group_stride, float scale,
                                          int* sat_count, sfvos_stream_t stream) {
  SFVOS_REQUIRE(src && dst && T > 0 && C > 0 && H > 0 && W > 0 && scale > 0.f, "frames_to_groups_fp8: bad argument");
  SFVOS_REQUIRE(C % 64 == 0 && group_stride >= (int64_t)T * H * W * 64 && group_stride % 16 == 0,
                "frames_to_groups_fp8: C must be a multiple of 64 and group_stride >= T*H*W*64");
  PlanarView v{st, sc, sh, sw, T, C, H, W};
  dim3 grid((unsigned)ceil_div64((int64_t)H * W, 64), (unsigned)ceil_div(C, 64), (unsigned)T);
  hipLaunchKernelGGL(to_ndhwc_kernel<SFVOS_FP8>, grid, dim3(256), 0, (hipStream_t)stream, src, v, (char*)dst, 64, 64,
                     (long long)group_stride, scale, sat_count);
  return check_launch("frames_to_groups_fp8");
}

extern "C" int sfvos_frames_absmax(const float* src, int64_t st, int64_t sc, int64_t sh, int64_t sw, int T, int C,
                                   int H, int W, float* amax, sfvos_stream_t stream) {
  SFVOS_REQUIRE(src && amax && T > 0 && C > 0 && H > 0 && W > 0, "frames_absmax: bad argument");
  PlanarView v{st, sc, sh, sw, T, C, H, W};
  hipLaunchKernelGGL(frames_absmax_kernel, dim3(grid_for((long long)T * C * H * W, 256 * 8)), dim3(256), 0,
                     (hipStream_t)stream, src, v, amax);
  return check_launch("frames_absmax");
}

extern "C" int sfvos_add_inplace(void* dst, const void* src, int dtype, int64_t n, sfvos_stream_t stream) {
  SFVOS_REQUIRE(dst && src && n > 0 && n % 8 == 0, "add_inplace: n must be a positive multiple of 8");
  SFVOS_REQUIRE((((size_t)dst | (size_t)src) & 15) == 0, "add_inplace: pointers must be 16-byte aligned");
  const long long chunks = n / (dtype == SFVOS_BF16 ? 8 : 4);
  hipStream_t s = (hipStream_t)stream;
  DT_DISPATCH(dtype,
              hipLaunchKernelGGL(add_inplace_kernel<SFVOS_F32>, dim3(grid_for(chunks, 256 * 4)), dim3(256), 0, s,
                                 (char*)dst, (const char*)src, chunks),
              hipLaunchKernelGGL(add_inplace_kernel<SFVOS_BF16>, dim3(grid_for(chunks, 256 * 4)), dim3(256), 0, s,
                                 (char*)dst, (const char*)src, chunks));
  return check_launch("add_inplace");
}

static int mse_table(const sfvos_mse_table* t, MseTab* m, bool need_grad, const char* what) {
  SFVOS_REQUIRE(t != nullptr && t->n >= 1 && t->n <= SFVOS_MAX_LEVELS, "%s: bad table", what);
  m->n = t->n;
  for (int l = 0; l < SFVOS_MAX_LEVELS; ++l) {
    const bool live = l < t->n;
    if (live)
      SFVOS_REQUIRE(t->out[l] && t->target[l] && t->numel[l] > 0 && (!need_grad || t->grad[l]),
                    "%s: level %d has a null pointer or no elements", what, l);
    m->out[l] = live ? t->out[l] : nullptr; m->target[l] = live ? t->target[l] : nullptr;
    m->grad[l] = live ? t->grad[l] : nullptr; m->numel[l] = live ? t->numel[l] : 0;
  }
  return SFVOS_OK;
}

extern "C" int sfvos_mse_loss_rows(const sfvos_mse_table* t) {
  if (t == nullptr || t->n < 1 || t->n > SFVOS_MAX_LEVELS) return SFVOS_E_ARG;
  return t->n * MSE_ROWS_PER_LEVEL;
}

extern "C" int sfvos_mse_loss(const sfvos_mse_table* t, float* part, float* loss, sfvos_stream_t stream) {
  MseTab m;
  int rc = mse_table(t, &m, false, "mse_loss");
  if (rc) return rc;
  SFVOS_REQUIRE(part && loss, "mse_loss: null pointer");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(mse_partial_kernel, dim3(m.n * MSE_ROWS_PER_LEVEL), dim3(256), 0, s, m, part);
  hipLaunchKernelGGL(mse_final_kernel, dim3(1), dim3(256), 0, s, (const float*)part, m.n * MSE_ROWS_PER_LEVEL, loss);
  return check_launch("mse_loss");
}

extern "C" int sfvos_mse_loss_grad(const sfvos_mse_table* t, const float* upstream, sfvos_stream_t stream) {
  MseTab m;
  int rc = mse_table(t, &m, true, "mse_loss_grad");
  if (rc) return rc;
  long long big = 0;
  for (int l = 0; l < m.n; ++l) big = m.numel[l] > big ? m.numel[l] : big;
  hipLaunchKernelGGL(mse_grad_kernel, dim3(grid_for(big, 1024, 1024u), m.n), dim3(256), 0, (hipStream_t)stream, m,
                     upstream);
  return check_launch("mse_loss_grad");
}

extern "C" int sfvos_pack_weights_fp8(const float* w, const float* bias, void* packed, float* bias_descale, int c_out,
                                      int c_in, int kt, int taps, float act_scale, sfvos_stream_t stream) {
  SFVOS_REQUIRE(w && packed && bias_descale && c_out > 0 && c_in > 0 && kt > 0 && (taps == 9 || taps == 1) &&
                act_scale > 0.f, "pack_weights_fp8: bad argument");
  SFVOS_REQUIRE(c_in % 64 == 0 && c_out % 32 == 0, "pack_weights_fp8: c_in must be a multiple of 64, c_out of 32");
  hipStream_t s = (hipStream_t)stream;
  float* wscale = bias_descale + 2 * (long long)c_out;  // row 2 of the caller's buffer
  const long long per_n = (long long)c_in * kt * taps;
  hipLaunchKernelGGL(fp8_weight_scale_kernel, dim3(c_out), dim3(256), 0, s, w, bias, per_n, c_out, act_scale,
                     bias_descale, wscale);
  const long long chunks = (long long)c_out * per_n / 16;
  hipLaunchKernelGGL(fp8_pack_weights_kernel, dim3(grid_for(chunks, 256)), dim3(256), 0, s, w, (const float*)wscale,
                     (char*)packed, c_out, c_in, kt, taps);
  return check_launch("pack_weights_fp8");
}

extern "C" int sfvos_planar_to_ndhwc(const float* src, void* dst, int dtype, int64_t M, int C, int ld,
                                     sfvos_stream_t stream) {
  SFVOS_REQUIRE(M > 0 && M < (1ll << 31), "planar_to_ndhwc: bad M");
  return sfvos_frames_to_ndhwc(src, 0, M, 0, 1, dst, dtype, 1, C, 1, (int)M, ld, stream);
}

extern "C" int sfvos_ndhwc_to_frames(const void* src, int dtype, float* dst, int64_t st, int64_t sc, int64_t sh,
                                     int64_t sw, int T, int C, int H, int W, int ld, int accumulate,
                                     sfvos_stream_t stream) {
  SFVOS_REQUIRE(src && dst && T > 0 && C > 0 && H > 0 && W > 0 && ld >= C, "ndhwc_to_frames: bad argument");
  SFVOS_REQUIRE(C % 8 == 0 && ld % 8 == 0, "ndhwc_to_frames: C and ld must be multiples of 8");
  PlanarView v{st, sc, sh, sw, T, C, H, W};
  dim3 grid((unsigned)ceil_div64((int64_t)H * W, 64), (unsigned)ceil_div(C, 64), (unsigned)T);
  hipStream_t s = (hipStream_t)stream;
  DT_DISPATCH(dtype,
              hipLaunchKernelGGL(from_ndhwc_kernel<SFVOS_F32>, grid, dim3(256), 0, s, (const char*)src, ld, dst, v,
                                 accumulate),
              hipLaunchKernelGGL(from_ndhwc_kernel<SFVOS_BF16>, grid, dim3(256), 0, s, (const char*)src, ld, dst, v,
                                 accumulate));
  return check_launch("ndhwc_to_frames");
}

static int make_level_views(const sfvos_planar_level* levels, int n_levels, int T, int C, int tile_px, LevelViews* t,
                            bool* vec, const char* what) {
  SFVOS_REQUIRE(levels && n_levels >= 1 && n_levels <= SFVOS_MAX_LEVELS && T > 0 && C > 0, "%s: bad argument", what);
  t->n = n_levels;
  long long pos = 0, blk = 0;
  *vec = true;
  for (int l = 0; l < SFVOS_MAX_LEVELS; ++l) {
    t->pos[l] = pos;
    if (l < n_levels) {
      const sfvos_planar_level& lv = levels[l];
      SFVOS_REQUIRE(lv.ptr && lv.h > 0 && lv.w > 0, "%s: level %d: bad view", what, l);
      t->v[l] = PlanarView{lv.stride_t, lv.stride_c, lv.stride_h, lv.stride_w, T, C, lv.h, lv.w};
      t->planar[l] = lv.ptr;
      *vec = *vec && planes_vectorizable(lv.ptr, t->v[l]);
      pos += (long long)T * lv.h * lv.w;
    } else {
      t->v[l] = PlanarView{0, 0, 0, 0, T, C, 1, 1};
      t->planar[l] = nullptr;
    }
  }
  // first pixel tile (of tile_px positions) of each level
  for (int l = 0; l < SFVOS_MAX_LEVELS; ++l) {
    t->blk_begin[l] = (int)blk;
    if (l < n_levels) blk += ceil_div64((long long)levels[l].h * levels[l].w, tile_px);
  }
  t->blk_begin[SFVOS_MAX_LEVELS] = (int)blk;
  SFVOS_REQUIRE(blk < (1ll << 31), "%s: grid out of range", what);
  return SFVOS_OK;
}

extern "C" int sfvos_pyramid_to_frames(const void* src, int dtype, const sfvos_planar_level* levels, int n_levels, int T,
                                       int C, int ld, int accumulate, sfvos_stream_t stream) {
  SFVOS_REQUIRE(src && ld >= C && C % 8 == 0 && ld % 8 == 0, "pyramid_to_frames: C and ld must be multiples of 8, ld >= C");
  LevelViews t;
  bool vec;
  int rc = make_level_views(levels, n_levels, T, C, 64, &t, &vec, "pyramid_to_frames");
  if (rc) return rc;
  dim3 grid((unsigned)t.blk_begin[SFVOS_MAX_LEVELS], (unsigned)ceil_div(C, 64), (unsigned)T);
  hipStream_t s = (hipStream_t)stream;
  DT_DISPATCH(dtype,
              hipLaunchKernelGGL(from_ndhwc_multi_kernel<SFVOS_F32>, grid, dim3(256), 0, s, (const char*)src, ld, t, accumulate),
              hipLaunchKernelGGL(from_ndhwc_multi_kernel<SFVOS_BF16>, grid, dim3(256), 0, s, (const char*)src, ld, t, accumulate));
  return check_launch("pyramid_to_frames");
}

extern "C" int sfvos_frames_to_pyramid(const sfvos_planar_level* levels, int n_levels, void* dst, int dtype, int T, int C,
                                       int ld, sfvos_stream_t stream) {
  SFVOS_REQUIRE(dst && ld >= C && C % 8 == 0 && ld % 8 == 0, "frames_to_pyramid: C and ld must be multiples of 8, ld >= C");
  LevelViews t;
  bool vec;
  int rc = make_level_views(levels, n_levels, T, C, 128, &t, &vec, "frames_to_pyramid");
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == SFVOS_BF16 && vec && ((size_t)dst & 15) == 0) {
    dim3 gridv((unsigned)t.blk_begin[SFVOS_MAX_LEVELS], (unsigned)ceil_div(C, 64), (unsigned)T);
    hipLaunchKernelGGL(to_ndhwc_bf16_vec_multi_kernel, gridv, dim3(256), 0, s, t, (char*)dst, ld);
    return check_launch("frames_to_pyramid");
  }
  rc = make_level_views(levels, n_levels, T, C, 64, &t, &vec, "frames_to_pyramid");
  if (rc) return rc;
  dim3 grid((unsigned)t.blk_begin[SFVOS_MAX_LEVELS], (unsigned)ceil_div(C, 64), (unsigned)T);
  DT_DISPATCH(dtype,
              hipLaunchKernelGGL(to_ndhwc_multi_kernel<SFVOS_F32>, grid, dim3(256), 0, s, t, (char*)dst, ld),
              hipLaunchKernelGGL(to_ndhwc_multi_kernel<SFVOS_BF16>, grid, dim3(256), 0, s, t, (char*)dst, ld));
  return check_launch("frames_to_pyramid");
}

extern "C" int sfvos_ndhwc_to_planar(const void* src, int dtype, float* dst, int64_t M, int C, int ld,
                                     sfvos_stream_t stream) {
  SFVOS_REQUIRE(M > 0 && M < (1ll << 31), "ndhwc_to_planar: bad M");
  return sfvos_ndhwc_to_frames(src, dtype, dst, 0, M, 0, 1, 1, C, 1, (int)M, ld, 0, stream);
}

extern "C" size_t sfvos_packed_weight_bytes(int dtype, int c_out, int c_in, int kt, int taps) {
  return (size_t)c_out * c_in * kt * taps * (dtype == SFVOS_BF16 ? 2 : 4);
}

static int pack_common(const float* w, void* packed, int dtype, int c_out, int c_in, int kt, int taps, bool dgrad,
                       sfvos_stream_t stream) {
  SFVOS_REQUIRE(w && packed, "pack_weights: null pointer");
  SFVOS_REQUIRE(c_out % 32 == 0 && c_in % 32 == 0 && kt >= 1 && (taps == 9 || taps == 1),
                "pack_weights: unsupported shape (c_out %d c_in %d kt %d taps %d)", c_out, c_in, kt, taps);
  const long long total = (long long)c_out * c_in * kt * taps;
  const unsigned grid = grid_for(total, 256);
  hipStream_t s = (hipStream_t)stream;
  if (dgrad)
    DT_DISPATCH(dtype,
                hipLaunchKernelGGL((pack_weights_kernel<SFVOS_F32, true>), dim3(grid), dim3(256), 0, s, w,
                                   (char*)packed, c_out, c_in, kt, taps),
                hipLaunchKernelGGL((pack_weights_kernel<SFVOS_BF16, true>), dim3(grid), dim3(256), 0, s, w,
                                   (char*)packed, c_out, c_in, kt, taps));
  else
    DT_DISPATCH(dtype,
                hipLaunchKernelGGL((pack_weights_kernel<SFVOS_F32, false>), dim3(grid), dim3(256), 0, s, w,
                                   (char*)packed, c_out, c_in, kt, taps),
                hipLaunchKernelGGL((pack_weights_kernel<SFVOS_BF16, false>), dim3(grid), dim3(256), 0, s, w,
                                   (char*)packed, c_out, c_in, kt, taps));
  return check_launch("pack_weights");
}

extern "C" int sfvos_pack_weights_fwd(const float* w, void* packed, int dtype, int c_out, int c_in, int kt, int taps,
                                      sfvos_stream_t stream) {
  return pack_common(w, packed, dtype, c_out, c_in, kt, taps, false, stream);
}
extern "C" int sfvos_pack_weights_dgrad(const float* w, void* packed, int dtype, int c_out, int c_in, int kt, int taps,
                                        sfvos_stream_t stream) {
  return pack_common(w, packed, dtype, c_out, c_in, kt, taps, true, stream);
}

extern "C" int sfvos_pack_weights_batch(const sfvos_pack_item* items, int n, int dtype, sfvos_stream_t stream) {
  SFVOS_REQUIRE(items && n >= 1 && n <= SFVOS_MAX_PACK_ITEMS, "pack_weights_batch: 1..%d items", SFVOS_MAX_PACK_ITEMS);
  PackTab t;
  t.n = n;
  long long blk = 0;
  for (int i = 0; i < SFVOS_MAX_PACK_ITEMS; ++i) {
    t.blk_begin[i] = (int)blk;
    if (i < n) {
      const sfvos_pack_item& it = items[i];
      SFVOS_REQUIRE(it.w && it.packed, "pack_weights_batch: item %d: null pointer", i);
      SFVOS_REQUIRE(it.c_out % 32 == 0 && it.c_in % 32 == 0 && it.c_out > 0 && it.c_in > 0 && it.kt >= 1 &&
                        (it.taps == 9 || it.taps == 1),
                    "pack_weights_batch: item %d: unsupported shape (c_out %d c_in %d kt %d taps %d)", i, it.c_out,
                    it.c_in, it.kt, it.taps);
      t.w[i] = it.w; t.packed[i] = (char*)it.packed; t.c_out[i] = it.c_out; t.c_in[i] = it.c_in; t.kt[i] = it.kt;
      t.taps[i] = it.taps; t.dgrad[i] = it.dgrad;
      blk += ceil_div64((long long)it.c_out * it.c_in * it.kt * it.taps, PACK_RUN);
      SFVOS_REQUIRE(blk < (1ll << 31), "pack_weights_batch: too many elements");
    } else {
      t.w[i] = nullptr; t.packed[i] = nullptr; t.c_out[i] = t.c_in[i] = t.kt[i] = t.taps[i] = 1; t.dgrad[i] = 0;
    }
  }
  t.blk_begin[SFVOS_MAX_PACK_ITEMS] = (int)blk;
  hipStream_t s = (hipStream_t)stream;
  DT_DISPATCH(dtype,
              hipLaunchKernelGGL(pack_weights_batch_kernel<SFVOS_F32>, dim3((unsigned)blk), dim3(256), 0, s, t),
              hipLaunchKernelGGL(pack_weights_batch_kernel<SFVOS_BF16>, dim3((unsigned)blk), dim3(256), 0, s, t));
  return check_launch("pack_weights_batch");
}

extern "C" int sfvos_mask_union(const float* masks, int n, int64_t hw, float threshold, unsigned char* out,
                                sfvos_stream_t stream) {
  SFVOS_REQUIRE(out && hw > 0 && n >= 0 && (masks || n == 0), "mask_union: bad argument");
  hipLaunchKernelGGL(mask_union_kernel, dim3(grid_for(hw, 256)), dim3(256), 0, (hipStream_t)stream, masks, n,
                     (long long)hw, threshold, out);
  return check_launch("mask_union");
}

extern "C" int sfvos_reduce_rows(const float* part, int rows, int C, float* out, int accumulate,
                                 sfvos_stream_t stream) {
  SFVOS_REQUIRE(part && out && rows > 0 && C > 0, "reduce_rows: bad argument");
  hipLaunchKernelGGL(reduce_rows_kernel, dim3(ceil_div(C, 32)), dim3(32 * RL), 0, (hipStream_t)stream, part, rows, C,
                     out, accumulate);
  return check_launch("reduce_rows");
}

extern "C" int sfvos_sgd_step(float* p, const float* g, float* buf, int64_t n, float lr, float momentum, float wd,
                              int first_step, sfvos_stream_t stream) {
  SFVOS_REQUIRE(p && g && buf && n > 0, "sgd_step: bad argument");
  hipLaunchKernelGGL(sgd_kernel, dim3(grid_for(n, 256 * 4)), dim3(256), 0, (hipStream_t)stream, p, g, buf,
                     (long long)n, lr, momentum, wd, first_step);
  return check_launch("sgd_step");
}

extern "C" int sfvos_scale(float* x, int64_t n, float s, sfvos_stream_t stream) {
  SFVOS_REQUIRE(x && n > 0, "scale: bad argument");
  hipLaunchKernelGGL(scale_kernel, dim3(grid_for(n, 256 * 4)), dim3(256), 0, (hipStream_t)stream, x, (long long)n, s);
  return check_launch("scale");
}
