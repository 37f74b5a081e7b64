// Shared device/host helpers for libsfvos (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/sfvos.h"

namespace sfvos {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short short4v;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

#define SFVOS_LDS __attribute__((address_space(3)))
#define SFVOS_GLOBAL __attribute__((address_space(1)))

// ---- element traits: everything on chip is organised in 16-byte chunks ---------------------
template <int DT> struct Elt;
template <> struct Elt<SFVOS_F32> {
  typedef float type;
  static constexpr int CE = 4;  // elements per 16-B chunk
  static __device__ __forceinline__ float to_f32(float v) { return v; }
  static __device__ __forceinline__ float from_f32(float v) { return v; }
};
template <> struct Elt<SFVOS_BF16> {
  typedef __bf16 type;
  static constexpr int CE = 8;
  static __device__ __forceinline__ float to_f32(__bf16 v) { return (float)v; }
  static __device__ __forceinline__ __bf16 from_f32(float v) { return (__bf16)v; }
};

template <> struct Elt<SFVOS_FP8> {   // OCP e4m3 bytes; conversions go through v_cvt_pk_fp8_f32 / v_cvt_f32_fp8
  typedef unsigned char type;
  static constexpr int CE = 16;
};
// element type a conv writes for a given operand type: fp8 operands -> bf16 results
template <int DT> struct YOf { static constexpr int DTY = DT == SFVOS_FP8 ? SFVOS_BF16 : DT; };
typedef __attribute__((ext_vector_type(8))) int i32x8;

// One MFMA "step" consumes one 16-B chunk per lane of A and of B: lane half h (lane>>5)
// holds chunk 2*step+h.  bf16: one 32x32x16; f32: four 32x32x2 (element e of both chunks
// pairs k = {chunk(2s)*4+e, chunk(2s+1)*4+e}); the k-sum is the same set either way.
template <int DT> struct Mma;
template <> struct Mma<SFVOS_BF16> {
  static __device__ __forceinline__ void run(f32x16& acc, const u32x4& a, const u32x4& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0,
                                                   0, 0);
  }
};
template <> struct Mma<SFVOS_F32> {
  static __device__ __forceinline__ void run(f32x16& acc, const u32x4& a, const u32x4& b) {
    f32x4 fa = __builtin_bit_cast(f32x4, a), fb = __builtin_bit_cast(f32x4, b);
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[e], fb[e], acc, 0, 0, 0);
  }
};

// async global -> LDS copy of 16 B per lane; LDS destination = wave-uniform base + lane*16.
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const SFVOS_GLOBAL void*)gsrc, (SFVOS_LDS void*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ u32x4 lds_read16(const void* p) { return *(const u32x4*)p; }

// LDS-DMA through a buffer descriptor, as one asm statement: buffer_load_dwordx4 ... offen lds.
// Why not __builtin_amdgcn_raw_ptr_buffer_load_lds: hipcc tracks that builtin as an LDS write and, when it cannot
// prove that the destination does not alias a following LDS read (dynamic ring slots), puts `s_waitcnt vmcnt(0)`
// right behind the copy -- a full memory round trip inside the MFMA loop.  The copies here are ordered by hand
// (one `s_waitcnt vmcnt(0)` + barrier per stage before anything reads what they wrote), so the statement is opaque
// to the compiler on purpose.  M0 (the LDS destination base) is saved and restored inside the statement.
//   base/records: descriptor of the source (wave-uniform); voff: this lane's byte offset, >= records reads zeros;
//   lds_offset: wave-uniform byte offset inside the workgroup's LDS allocation (kernels here have dynamic LDS only, so
//   `ptr - smem` is the LDS address), lane i lands at +16 i.
typedef __attribute__((ext_vector_type(4))) int i32x4;
__device__ __forceinline__ void lds_dma16(const void* base, int records, unsigned voff, unsigned lds_offset) {
  const unsigned long long b = (unsigned long long)base;
  i32x4 rs;
  rs[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
  rs[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(b >> 32) & 0xffffu));   // stride 0
  rs[2] = __builtin_amdgcn_readfirstlane(records);
  rs[3] = 0x00020000;
  const unsigned dst = __builtin_amdgcn_readfirstlane(lds_offset);
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %1\n\t"
      "s_nop 0\n\t"
      "buffer_load_dwordx4 %2, %3, 0 offen lds\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "s"(dst), "v"(voff), "s"(rs)
      : "memory");
}

// ---- host side -------------------------------------------------------------------------------
void set_error(const char* fmt, ...);
int check_launch(const char* what);
int current_device();   // hipGetDevice, -1 on error
int device_cu_count();  // compute units of the current device (256 when the query fails)

// hipFuncAttributeMaxDynamicSharedMemorySize must be raised once per (kernel, device): one instance per kernel
// template instantiation (function-local static), one atomic flag per device ordinal.
struct LdsAttrOnce {
  static constexpr int MAX_DEV = 64;
  int done[MAX_DEV] = {};
  int ensure(const void* kern, int lds_bytes, const char* what) {
    const int dev = current_device();
    if (dev >= 0 && dev < MAX_DEV && __atomic_load_n(&done[dev], __ATOMIC_ACQUIRE)) return SFVOS_OK;
    hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (e != hipSuccess) {
      set_error("%s: hipFuncSetAttribute(%d B LDS) failed: %s", what, lds_bytes, hipGetErrorString(e));
      return SFVOS_E_LAUNCH;
    }
    if (dev >= 0 && dev < MAX_DEV) __atomic_store_n(&done[dev], 1, __ATOMIC_RELEASE);
    return SFVOS_OK;
  }
};

// lateral.hip: dedicated kernel for the lateral (k x 1 x 1) data gradient; -1 = shape not covered
int lateral_dgrad_try(const sfvos_conv_desc* d, const void* x, const void* w_packed, void* y, hipStream_t stream);
// ... and for its forward (32 -> 64 channels): shape test, statistics rows (= workgroups) of a level, launch
bool lateral_fwd_applies(const sfvos_conv_desc* d);
int lateral_fwd_rows(const sfvos_conv_desc* d, int level);
int lateral_fwd_launch(const sfvos_conv_desc* d, const void* x, const void* w_packed, const float* bias, void* y,
                       float* stat_part, hipStream_t stream);

// lateral_wgrad.hip: weight gradient of the lateral convs; -1 / 0 = shape not covered
size_t lateral_wgrad_workspace_bytes(const sfvos_conv_desc* d);
int lateral_wgrad_try(const sfvos_conv_desc* d, const void* x, const void* dy, float* grad_w, int accumulate,
                      void* workspace, hipStream_t stream);
// wgrad.hip: grad_w[n][c][dt][tap] (=|+=) sum over the psplit slabs [n][dt][tap][c], fixed order
// wgrad_t1.hip: weight gradient of a kt x 3 x 3 conv 32 -> 32 with one output frame (fast_conv3); -1 / 0 = shape not covered
size_t wgrad_t1_workspace_bytes(const sfvos_conv_desc* d);
int wgrad_t1_try(const sfvos_conv_desc* d, const void* x, const void* dy, float* grad_w, int accumulate, void* workspace,
                 hipStream_t stream);
int launch_wgrad_reduce(const float* slab, int psplit, int c_out, int c_in, int kt, int taps, float* grad_w,
                        int accumulate, hipStream_t stream);

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

#define SFVOS_REQUIRE(cond, ...)   \
  do {                             \
    if (!(cond)) {                 \
      sfvos::set_error(__VA_ARGS__); \
      return SFVOS_E_ARG;          \
    }                              \
  } while (0)

}  // namespace sfvos
