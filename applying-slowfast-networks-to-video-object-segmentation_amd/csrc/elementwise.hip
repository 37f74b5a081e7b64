// elementwise.hip -- the HBM-bound pieces of the SlowFastLayers hot path:
//   layout passes (reference model.py:157-158 stack/transpose, :162 cat/squeeze),
//   weight packing, BatchNorm3d forward/backward (+ fused ReLU, model.py:113-114,121-122,...),
//   SGD (train.py:80).  All NDHWC, 16 bytes per lane per access.
#include "common.h"

namespace sfvos {

template <int DT> __device__ __forceinline__ void unpack(const u32x4& v, float* f);
template <> __device__ __forceinline__ void unpack<SFVOS_F32>(const u32x4& v, float* f) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    // copy the lane to a scalar first: hipcc (ROCm 7.2) mis-compiles __builtin_bit_cast applied
    // directly to an ext-vector element lvalue (every e reads element 0)
    const unsigned u = v[e];
    f[e] = __builtin_bit_cast(float, u);
  }
}
template <> __device__ __forceinline__ void unpack<SFVOS_BF16>(const u32x4& v, float* f) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const unsigned u = v[e];
    const unsigned lo = u << 16, hi = u & 0xffff0000u;
    f[2 * e] = __builtin_bit_cast(float, lo);
    f[2 * e + 1] = __builtin_bit_cast(float, hi);
  }
}
template <int DT> __device__ __forceinline__ u32x4 pack(const float* f);
template <> __device__ __forceinline__ u32x4 pack<SFVOS_F32>(const float* f) {
  u32x4 v;
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = __builtin_bit_cast(unsigned, f[e]);
  return v;
}
template <> __device__ __forceinline__ u32x4 pack<SFVOS_BF16>(const float* f) {
  u32x4 v;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const unsigned lo = __builtin_bit_cast(unsigned short, (__bf16)f[2 * e]);
    const unsigned hi = __builtin_bit_cast(unsigned short, (__bf16)f[2 * e + 1]);
    v[e] = lo | (hi << 16);
  }
  return v;
}

// ---- layout: planar/strided fp32 <-> NDHWC ------------------------------------------------------
// tile = 64 positions x 64 channels through LDS; planar side coalesced along positions,
// NDHWC side 16 B per lane along channels.
struct PlanarView {
  long long st, sc, sh, sw;  // element strides of the fp32 planar tensor
  int T, C, H, W;
};

template <int DT>
__global__ __launch_bounds__(256) void to_ndhwc_kernel(const float* __restrict__ src, PlanarView v, char* dst, int ld) {
  constexpr int CE = Elt<DT>::CE;
  __shared__ float tile[64][65];
  const int tid = threadIdx.x;
  const long long HW = (long long)v.H * v.W;
  const long long p0 = (long long)blockIdx.x * 64;
  const int c0 = blockIdx.y * 64, t = blockIdx.z;
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
  for (int i = tid; i < 64 * CPR; i += 256) {
    const int px = i / CPR, ch = i - px * CPR;
    const long long p = p0 + px;
    if (p < HW && c0 + ch * CE < v.C) {
      float f[CE];
#pragma unroll
      for (int e = 0; e < CE; ++e) f[e] = tile[ch * CE + e][px];
      *(u32x4*)(dst + (((long long)t * HW + p) * ld + c0 + ch * CE) * (16 / CE)) = pack<DT>(f);
    }
  }
}

template <int DT>
__global__ __launch_bounds__(256) void from_ndhwc_kernel(const char* __restrict__ src, int ld, float* dst, PlanarView v,
                                                         int accumulate) {
  constexpr int CE = Elt<DT>::CE;
  __shared__ float tile[64][65];
  const int tid = threadIdx.x;
  const long long HW = (long long)v.H * v.W;
  const long long p0 = (long long)blockIdx.x * 64;
  const int c0 = blockIdx.y * 64, t = blockIdx.z;
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

// ---- weight packing -----------------------------------------------------------------------------
// packed[cc][dt][tap][j][n][e]  (one 16-B chunk = CE reduction channels for one output channel n;
// a conv stage (cc, dt, tap group) is one contiguous block)
template <int DT, bool DGRAD>
__global__ __launch_bounds__(256) void pack_weights_kernel(const float* __restrict__ w, char* packed, int c_out,
                                                           int c_in, int kt, int taps) {
  typedef typename Elt<DT>::type T;
  constexpr int CE = Elt<DT>::CE, CK = 4 * CE;
  // N = output channels of the conv this image feeds, Kc = its reduction channels
  const int N = DGRAD ? c_in : c_out, Kc = DGRAD ? c_out : c_in;
  const long long total = (long long)N * Kc * kt * taps;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    long long k = i;
    const int e = (int)(k % CE); k /= CE;
    const int n = (int)(k % N); k /= N;
    const int j = (int)(k % 4); k /= 4;
    const int tap = (int)(k % taps); k /= taps;
    const int dt = (int)(k % kt); k /= kt;
    const int cc = (int)k;
    const int c = cc * CK + j * CE + e;
    float val;
    if (DGRAD)  // w[co = c][ci = n][kt-1-dt][taps-1-tap]
      val = w[(((long long)c * c_in + n) * kt + (kt - 1 - dt)) * taps + (taps - 1 - tap)];
    else  // w[co = n][ci = c][dt][tap]
      val = w[(((long long)n * c_in + c) * kt + dt) * taps + tap];
    ((T*)packed)[i] = Elt<DT>::from_f32(val);
  }
}

// ---- batch norm ---------------------------------------------------------------------------------
// Deterministic column sums of part[rows][ncol]: block = 32 channels x RL row lanes, each lane sums
// its strided rows (4 loads in flight), then lane 0 adds the RL lane sums in a fixed order.
constexpr int RL = 32;
__device__ __forceinline__ double column_sum(const float* part, int rows, int ncol, int col, int sub, double* scratch,
                                             int c_local) {
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  int rI = sub;
  for (; rI + 3 * RL < rows; rI += 4 * RL) {
    s0 += (double)part[(long long)rI * ncol + col];
    s1 += (double)part[(long long)(rI + RL) * ncol + col];
    s2 += (double)part[(long long)(rI + 2 * RL) * ncol + col];
    s3 += (double)part[(long long)(rI + 3 * RL) * ncol + col];
  }
  for (; rI < rows; rI += RL) s0 += (double)part[(long long)rI * ncol + col];
  scratch[sub * 32 + c_local] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  double tot = 0.0;
  if (sub == 0)
#pragma unroll
    for (int k = 0; k < RL; ++k) tot += scratch[k * 32 + c_local];
  __syncthreads();
  return tot;
}

// block = 32 channels x RL row lanes; grid = C/32
__global__ __launch_bounds__(32 * RL) void bn_finalize_kernel(const float* part, int rows, double count,
                                                          const float* gamma, const float* beta, float eps, int C,
                                                          float* mean, float* rstd, float* scale, float* shift,
                                                          float* var_unbiased) {
  __shared__ double scratch[32 * RL];
  const int c_local = threadIdx.x & 31, sub = threadIdx.x >> 5, c = blockIdx.x * 32 + c_local;
  const double s1 = column_sum(part, rows, 2 * C, c, sub, scratch, c_local);
  const double s2 = column_sum(part, rows, 2 * C, C + c, sub, scratch, c_local);
  if (sub == 0) {
    const double m = s1 / count;
    double var = s2 / count - m * m;
    if (var < 0.0) var = 0.0;
    const float r = (float)(1.0 / sqrt(var + (double)eps));
    mean[c] = (float)m;
    rstd[c] = r;
    const float sc = gamma[c] * r;
    scale[c] = sc;
    shift[c] = beta[c] - (float)m * sc;
    var_unbiased[c] = (float)(count > 1.0 ? var * count / (count - 1.0) : var);
  }
}

__global__ void bn_eval_coeffs_kernel(const float* gamma, const float* beta, const float* rm, const float* rv,
                                      float eps, int C, float* scale, float* shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) {
    const float sc = gamma[c] / sqrtf(rv[c] + eps);
    scale[c] = sc;
    shift[c] = beta[c] - rm[c] * sc;
  }
}

__global__ void bn_running_update_kernel(float* rm, float* rv, const float* means, const float* vars, int n, int C,
                                         float momentum) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) {
    float m = rm[c], v = rv[c];
    for (int i = 0; i < n; ++i) {
      m = (1.f - momentum) * m + momentum * means[(long long)i * C + c];
      v = (1.f - momentum) * v + momentum * vars[(long long)i * C + c];
    }
    rm[c] = m;
    rv[c] = v;
  }
}

template <int DT>
__global__ __launch_bounds__(256) void bn_apply_kernel(const char* __restrict__ x, int ld_x, char* y, int ld_y,
                                                       long long M, int C, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, int relu) {
  constexpr int CE = Elt<DT>::CE, ES = 16 / CE;
  const int cpr = C / CE;
  const long long total = M * cpr;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long m = i / cpr;
    const int c = (int)(i - m * cpr) * CE;
    float f[CE];
    unpack<DT>(*(const u32x4*)(x + (m * ld_x + c) * ES), f);
#pragma unroll
    for (int e = 0; e < CE; ++e) {
      float v = f[e] * scale[c + e] + shift[c + e];
      f[e] = relu ? fmaxf(v, 0.f) : v;
    }
    *(u32x4*)(y + (m * ld_y + c) * ES) = pack<DT>(f);
  }
}

// BN backward pass 1 / pass 2 share the thread layout: a block owns a strided set of position
// blocks; thread = (chunk of CE channels, row lane); per-channel partials reduced through LDS in
// a fixed order -> one deterministic partial row per block.
constexpr int BNB_THREADS = 256;
constexpr int BNB_POS = 512;  // positions per block iteration

template <int DT, bool APPLY>
__global__ __launch_bounds__(BNB_THREADS) void bn_bwd_kernel(
    const char* __restrict__ dy, int ld_dy, const char* __restrict__ x, int ld_x, char* dx, int ld_dx, long long M,
    int C, const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ mean,
    const float* __restrict__ rstd, int relu, const float* __restrict__ cA, const float* __restrict__ cB,
    const float* __restrict__ cK, float* part) {
  constexpr int CE = Elt<DT>::CE, ES = 16 / CE;
  const int cpr = C / CE;                    // chunks per position (<= 64 for C <= 256 bf16; <= 64 f32 needs C <= 256)
  const int rl = BNB_THREADS / cpr;          // row lanes per block
  const int ch = threadIdx.x % cpr, rowl = threadIdx.x / cpr;
  const int c = ch * CE;
  float a0[CE], a1[CE];
#pragma unroll
  for (int e = 0; e < CE; ++e) a0[e] = a1[e] = 0.f;
  float sc[CE], sh[CE], p0[CE], p1[CE], p2[CE];
#pragma unroll
  for (int e = 0; e < CE; ++e) {
    sc[e] = scale[c + e];
    sh[e] = shift[c + e];
    if (APPLY) { p0[e] = cA[c + e]; p1[e] = cB[c + e]; p2[e] = cK[c + e]; }
    else { p0[e] = mean[c + e]; p1[e] = rstd[c + e]; p2[e] = 0.f; }
  }
  if (rowl < rl) {
    for (long long pb = (long long)blockIdx.x * BNB_POS; pb < M; pb += (long long)gridDim.x * BNB_POS) {
      const long long pend = pb + BNB_POS < M ? pb + BNB_POS : M;
      for (long long m = pb + rowl; m < pend; m += rl) {
        float fdy[CE], fx[CE];
        unpack<DT>(*(const u32x4*)(dy + (m * ld_dy + c) * ES), fdy);
        unpack<DT>(*(const u32x4*)(x + (m * ld_x + c) * ES), fx);
        float out[CE];
#pragma unroll
        for (int e = 0; e < CE; ++e) {
          const float dz = (relu && !(fx[e] * sc[e] + sh[e] > 0.f)) ? 0.f : fdy[e];
          if (APPLY) {
            const float d = p0[e] * dz + p1[e] * fx[e] + p2[e];
            out[e] = d;
            a0[e] += d;
          } else {
            a0[e] += dz;
            a1[e] += dz * ((fx[e] - p0[e]) * p1[e]);
          }
        }
        if (APPLY) *(u32x4*)(dx + (m * ld_dx + c) * ES) = pack<DT>(out);
      }
    }
  }
  if (part == nullptr) return;
  __shared__ float red[2][BNB_THREADS][8];
#pragma unroll
  for (int e = 0; e < CE; ++e) {
    red[0][threadIdx.x][e] = a0[e];
    red[1][threadIdx.x][e] = a1[e];
  }
  __syncthreads();
  // thread -> one channel (and one of the two sums)
  const int nsum = APPLY ? 1 : 2;
  for (int i = threadIdx.x; i < nsum * C; i += BNB_THREADS) {
    const int which = i / C, cc = i - which * C;
    const int chunk = cc / CE, e = cc - chunk * CE;
    float s = 0.f;
    for (int k = 0; k < rl; ++k) s += red[which][k * cpr + chunk][e];
    part[((long long)blockIdx.x * nsum + which) * C + cc] = s;
  }
}

// dgamma/dbeta and the pass-2 coefficients.  block = 32 channels x 8 row lanes.
__global__ __launch_bounds__(32 * RL) void bn_bwd_finalize_kernel(const float* part, int rows, double count,
                                                              const float* gamma, const float* mean,
                                                              const float* rstd, int C, int train, int accumulate,
                                                              float* dgamma, float* dbeta, float* cA, float* cB,
                                                              float* cK) {
  __shared__ double scratch[32 * RL];
  const int c_local = threadIdx.x & 31, sub = threadIdx.x >> 5, c = blockIdx.x * 32 + c_local;
  const double sdz = column_sum(part, rows, 2 * C, c, sub, scratch, c_local);
  const double sdzx = column_sum(part, rows, 2 * C, C + c, sub, scratch, c_local);
  if (sub == 0) {
    if (dgamma) dgamma[c] = (accumulate ? dgamma[c] : 0.f) + (float)sdzx;
    if (dbeta) dbeta[c] = (accumulate ? dbeta[c] : 0.f) + (float)sdz;
    const double g = (double)gamma[c] * (double)rstd[c];
    if (train) {
      const double m1 = sdz / count, m2 = sdzx / count;
      cA[c] = (float)g;
      cB[c] = (float)(-g * m2 * (double)rstd[c]);
      cK[c] = (float)(g * m2 * (double)rstd[c] * (double)mean[c] - g * m1);
    } else {
      cA[c] = (float)g;
      cB[c] = 0.f;
      cK[c] = 0.f;
    }
  }
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

static inline unsigned grid_for(long long work_items, int per_block, unsigned cap = 2048u * 4u) {
  long long g = ceil_div64(work_items, per_block);
  if (g < 1) g = 1;
  if (g > (long long)cap) g = cap;
  return (unsigned)g;
}

}  // namespace sfvos

using namespace sfvos;

#define DT_DISPATCH(dtype, CALL_F32, CALL_BF16)                    \
  do {                                                             \
    if ((dtype) == SFVOS_F32) { CALL_F32; }                        \
    else if ((dtype) == SFVOS_BF16) { CALL_BF16; }                 \
    else { set_error("bad dtype %d", (int)(dtype)); return SFVOS_E_ARG; } \
  } while (0)

extern "C" int sfvos_frames_to_ndhwc(const float* src, int64_t st, int64_t sc, int64_t sh, int64_t sw, void* dst,
                                     int dtype, int T, int C, int H, int W, int ld, sfvos_stream_t stream) {
  SFVOS_REQUIRE(src && dst && T > 0 && C > 0 && H > 0 && W > 0 && ld >= C, "frames_to_ndhwc: bad argument");
  SFVOS_REQUIRE(C % 8 == 0 && ld % 8 == 0, "frames_to_ndhwc: C and ld must be multiples of 8");
  PlanarView v{st, sc, sh, sw, T, C, H, W};
  dim3 grid((unsigned)ceil_div64((int64_t)H * W, 64), (unsigned)ceil_div(C, 64), (unsigned)T);
  hipStream_t s = (hipStream_t)stream;
  DT_DISPATCH(dtype, hipLaunchKernelGGL(to_ndhwc_kernel<SFVOS_F32>, grid, dim3(256), 0, s, src, v, (char*)dst, ld),
              hipLaunchKernelGGL(to_ndhwc_kernel<SFVOS_BF16>, grid, dim3(256), 0, s, src, v, (char*)dst, ld));
  return check_launch("frames_to_ndhwc");
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

extern "C" int sfvos_bn_finalize(const float* part, int rows, int64_t count, const float* gamma, const float* beta,
                                 float eps, int C, float* mean, float* rstd, float* scale, float* shift,
                                 float* save_var_unbiased, sfvos_stream_t stream) {
  SFVOS_REQUIRE(part && gamma && beta && mean && rstd && scale && shift && save_var_unbiased, "bn_finalize: null");
  SFVOS_REQUIRE(rows > 0 && count > 0 && C > 0 && C % 32 == 0, "bn_finalize: bad rows/count/C");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(C / 32), dim3(32 * RL), 0, (hipStream_t)stream, part, rows, (double)count,
                     gamma, beta, eps, C, mean, rstd, scale, shift, save_var_unbiased);
  return check_launch("bn_finalize");
}

extern "C" int sfvos_bn_eval_coeffs(const float* gamma, const float* beta, const float* rm, const float* rv, float eps,
                                    int C, float* scale, float* shift, sfvos_stream_t stream) {
  SFVOS_REQUIRE(gamma && beta && rm && rv && scale && shift && C > 0, "bn_eval_coeffs: bad argument");
  hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3(ceil_div(C, 64)), dim3(64), 0, (hipStream_t)stream, gamma, beta, rm,
                     rv, eps, C, scale, shift);
  return check_launch("bn_eval_coeffs");
}

extern "C" int sfvos_bn_running_update(float* rm, float* rv, const float* means, const float* vars, int n, int C,
                                       float momentum, sfvos_stream_t stream) {
  SFVOS_REQUIRE(rm && rv && means && vars && n >= 0 && C > 0, "bn_running_update: bad argument");
  hipLaunchKernelGGL(bn_running_update_kernel, dim3(ceil_div(C, 64)), dim3(64), 0, (hipStream_t)stream, rm, rv, means,
                     vars, n, C, momentum);
  return check_launch("bn_running_update");
}

static int check_act(const char* what, int dtype, int64_t M, int C, int ld_a, int ld_b) {
  const int ce = dtype == SFVOS_BF16 ? 8 : 4;
  SFVOS_REQUIRE(dtype == SFVOS_F32 || dtype == SFVOS_BF16, "%s: bad dtype", what);
  SFVOS_REQUIRE(M > 0 && C > 0 && C % 32 == 0 && C <= 256, "%s: bad M/C (C must be a multiple of 32, <= 256)", what);
  SFVOS_REQUIRE(ld_a >= C && ld_b >= C && ld_a % ce == 0 && ld_b % ce == 0, "%s: bad pitch", what);
  return SFVOS_OK;
}

extern "C" int sfvos_bn_apply(const void* x, int ld_x, void* y, int ld_y, int dtype, int64_t M, int C,
                              const float* scale, const float* shift, int relu, sfvos_stream_t stream) {
  int rc = check_act("bn_apply", dtype, M, C, ld_x, ld_y);
  if (rc) return rc;
  SFVOS_REQUIRE(x && y && scale && shift, "bn_apply: null pointer");
  const int ce = dtype == SFVOS_BF16 ? 8 : 4;
  const unsigned grid = grid_for(M * (C / ce), 256 * 4);
  hipStream_t s = (hipStream_t)stream;
  DT_DISPATCH(dtype,
              hipLaunchKernelGGL(bn_apply_kernel<SFVOS_F32>, dim3(grid), dim3(256), 0, s, (const char*)x, ld_x,
                                 (char*)y, ld_y, (long long)M, C, scale, shift, relu),
              hipLaunchKernelGGL(bn_apply_kernel<SFVOS_BF16>, dim3(grid), dim3(256), 0, s, (const char*)x, ld_x,
                                 (char*)y, ld_y, (long long)M, C, scale, shift, relu));
  return check_launch("bn_apply");
}

extern "C" int sfvos_bn_bwd_rows(int64_t M) {
  int64_t r = ceil_div64(M, BNB_POS);
  if (r < 1) r = 1;
  if (r > 2048) r = 2048;
  return (int)r;
}

extern "C" int sfvos_bn_bwd_reduce(const void* dy, int ld_dy, const void* x, int ld_x, int dtype, int64_t M, int C,
                                   const float* scale, const float* shift, const float* mean, const float* rstd,
                                   int relu, float* part, sfvos_stream_t stream) {
  int rc = check_act("bn_bwd_reduce", dtype, M, C, ld_dy, ld_x);
  if (rc) return rc;
  SFVOS_REQUIRE(dy && x && scale && shift && mean && rstd && part, "bn_bwd_reduce: null pointer");
  const unsigned grid = (unsigned)sfvos_bn_bwd_rows(M);
  hipStream_t s = (hipStream_t)stream;
  DT_DISPATCH(dtype,
              hipLaunchKernelGGL((bn_bwd_kernel<SFVOS_F32, false>), dim3(grid), dim3(BNB_THREADS), 0, s,
                                 (const char*)dy, ld_dy, (const char*)x, ld_x, (char*)nullptr, 0, (long long)M, C,
                                 scale, shift, mean, rstd, relu, (const float*)nullptr, (const float*)nullptr,
                                 (const float*)nullptr, part),
              hipLaunchKernelGGL((bn_bwd_kernel<SFVOS_BF16, false>), dim3(grid), dim3(BNB_THREADS), 0, s,
                                 (const char*)dy, ld_dy, (const char*)x, ld_x, (char*)nullptr, 0, (long long)M, C,
                                 scale, shift, mean, rstd, relu, (const float*)nullptr, (const float*)nullptr,
                                 (const float*)nullptr, part));
  return check_launch("bn_bwd_reduce");
}

extern "C" int sfvos_bn_bwd_finalize(const float* part, int rows, int64_t count, const float* gamma, const float* mean,
                                     const float* rstd, int C, int train, int accumulate, float* dgamma, float* dbeta,
                                     float* coefA, float* coefB, float* coefK, sfvos_stream_t stream) {
  SFVOS_REQUIRE(part && gamma && mean && rstd && coefA && coefB && coefK, "bn_bwd_finalize: null pointer");
  SFVOS_REQUIRE(rows > 0 && count > 0 && C > 0 && C % 32 == 0, "bn_bwd_finalize: bad rows/count/C");
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C / 32), dim3(32 * RL), 0, (hipStream_t)stream, part, rows,
                     (double)count, gamma, mean, rstd, C, train, accumulate, dgamma, dbeta, coefA, coefB, coefK);
  return check_launch("bn_bwd_finalize");
}

extern "C" int sfvos_bn_bwd_apply(const void* dy, int ld_dy, const void* x, int ld_x, void* dx, int ld_dx, int dtype,
                                  int64_t M, int C, const float* scale, const float* shift, int relu,
                                  const float* coefA, const float* coefB, const float* coefK, float* bias_part,
                                  sfvos_stream_t stream) {
  int rc = check_act("bn_bwd_apply", dtype, M, C, ld_dy, ld_x);
  if (rc) return rc;
  SFVOS_REQUIRE(dy && x && dx && scale && shift && coefA && coefB && coefK, "bn_bwd_apply: null pointer");
  SFVOS_REQUIRE(ld_dx >= C && ld_dx % (dtype == SFVOS_BF16 ? 8 : 4) == 0, "bn_bwd_apply: bad ld_dx");
  const unsigned grid = (unsigned)sfvos_bn_bwd_rows(M);
  hipStream_t s = (hipStream_t)stream;
  DT_DISPATCH(dtype,
              hipLaunchKernelGGL((bn_bwd_kernel<SFVOS_F32, true>), dim3(grid), dim3(BNB_THREADS), 0, s,
                                 (const char*)dy, ld_dy, (const char*)x, ld_x, (char*)dx, ld_dx, (long long)M, C,
                                 scale, shift, (const float*)nullptr, (const float*)nullptr, relu, coefA, coefB, coefK,
                                 bias_part),
              hipLaunchKernelGGL((bn_bwd_kernel<SFVOS_BF16, true>), dim3(grid), dim3(BNB_THREADS), 0, s,
                                 (const char*)dy, ld_dy, (const char*)x, ld_x, (char*)dx, ld_dx, (long long)M, C,
                                 scale, shift, (const float*)nullptr, (const float*)nullptr, relu, coefA, coefB, coefK,
                                 bias_part));
  return check_launch("bn_bwd_apply");
}

extern "C" int sfvos_reduce_rows(const float* part, int rows, int C, float* out, int accumulate,
                                 sfvos_stream_t stream) {
  SFVOS_REQUIRE(part && out && rows > 0 && C > 0, "reduce_rows: bad argument");
  hipLaunchKernelGGL(reduce_rows_kernel, dim3(ceil_div(C, 32)), dim3(32 * RL), 0, (hipStream_t)stream, part, rows, C, out,
                     accumulate);
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
