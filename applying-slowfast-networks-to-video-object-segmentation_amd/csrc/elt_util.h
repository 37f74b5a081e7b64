// Helpers shared by the HBM-bound kernels (elementwise.hip, batchnorm.hip): 16-byte pack/unpack,
// deterministic column sums, dtype dispatch.
#pragma once
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

template <> __device__ __forceinline__ u32x4 pack<SFVOS_FP8>(const float* f) {  // 16 floats -> 16 e4m3 (saturating)
  u32x4 v;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    int w = __builtin_amdgcn_cvt_pk_fp8_f32(f[4 * e], f[4 * e + 1], 0, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(f[4 * e + 2], f[4 * e + 3], w, true);
    v[e] = (unsigned)w;
  }
  return v;
}

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

static inline unsigned grid_for(long long work_items, int per_block, unsigned cap = 2048u * 4u) {
  long long g = ceil_div64(work_items, per_block);
  if (g < 1) g = 1;
  if (g > (long long)cap) g = cap;
  return (unsigned)g;
}

#define DT_DISPATCH(dtype, CALL_F32, CALL_BF16)  /* f32 / bf16 only */                    \
  do {                                                             \
    if ((dtype) == SFVOS_F32) { CALL_F32; }                        \
    else if ((dtype) == SFVOS_BF16) { CALL_BF16; }                 \
    else { sfvos::set_error("bad dtype %d", (int)(dtype)); return SFVOS_E_ARG; } \
  } while (0)

}  // namespace sfvos
