// MFMA-only streams at the occupancy of the conv kernels (ONE workgroup per CU, LDS-limited): which rate does the matrix
// pipe sustain with 1, 2, 3 or 4 waves per SIMD, for v_mfma_f32_16x16x32_bf16 (16 accumulators per wave, the
// frame-split kernel's 6-MFMA groups) and v_mfma_f32_32x32x16_bf16 (9 accumulators, the wide kernel's 9-MFMA groups)?
// hipcc --offload-arch=gfx950 -O3 tools/diag/mfma_occ.hip -o /tmp/mfma_occ
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int SHAPE, int THREADS>
__global__ __launch_bounds__(THREADS) void k(const float* seed, float* out, int iters) {
  extern __shared__ char smem[];
  bf16x8 a[3], b[6];
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 8; ++j) {
      if (i < 3) a[i][j] = (__bf16)seed[(threadIdx.x * 37 + i * 8 + j) & 1023];
      b[i][j] = (__bf16)seed[(threadIdx.x * 53 + i * 8 + j + 7) & 1023];
    }
  float s = 0;
  if constexpr (SHAPE == 16) {
    f32x4 acc[8][2];
    for (int i = 0; i < 8; ++i)
      for (int n = 0; n < 2; ++n)
        for (int e = 0; e < 4; ++e) acc[i][n][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int t = 0; t < 30; ++t) {   // one "stage": 3 column shifts x 10 halo rows -> 144 MFMAs
        const int rr = t % 10;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int dh = 0; dh < 3; ++dh) {
          const int i = rr - dh;
          if (i >= 0 && i < 8) {
#pragma unroll
            for (int n = 0; n < 2; ++n)
              acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[t % 3], b[dh * 2 + n], acc[i][n], 0, 0, 0);
          }
        }
      }
    }
    for (int i = 0; i < 8; ++i)
      for (int n = 0; n < 2; ++n)
        for (int e = 0; e < 4; ++e) s += acc[i][n][e];
  } else if constexpr (SHAPE == 17 || SHAPE == 18) {
    // LDS-fed: the 16-wave layout (MT = 4 rows per wave: 18 steps of 1 A + 1 B read and 4 MFMAs = 72 MFMAs per stage)
    // resp. today's 8-wave layout (SHAPE 18: MT = 8: 30 steps, 1 A read + 0.6 B reads and 4.8 MFMAs = 144 per stage)
    constexpr int MT = SHAPE == 17 ? 4 : 8, ROWS = MT + 2, NA = 3 * ROWS;
    f32x4 acc[MT][2];
    for (int i = 0; i < MT; ++i)
      for (int n = 0; n < 2; ++n)
        for (int e = 0; e < 4; ++e) acc[i][n][e] = 0.f;
    const char* base = smem + (threadIdx.x & 63) * 16;
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
    // fill the 96 KB the loop reads with the seed's values as bf16 (random or zero operands)
    for (int i = threadIdx.x; i < 96 * 1024 / 2; i += THREADS) ((__bf16*)smem)[i] = (__bf16)seed[(i * 13 + 5) & 1023];
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
      u32x4 av[3], bw[2][3][2];
      auto lda = [&](int t) { av[t % 3] = *(const u32x4*)(base + ((t * 7 + it) & 63) * 1024); };
      auto ldb = [&](int dw) {
#pragma unroll
        for (int dh = 0; dh < 3; ++dh)
#pragma unroll
          for (int n = 0; n < 2; ++n) bw[dw & 1][dh][n] = *(const u32x4*)(base + 65536 + ((dw * 6 + dh * 2 + n + it) & 31) * 1024);
      };
      ldb(0); lda(0); lda(1);
#pragma unroll
      for (int t = 0; t < NA; ++t) {
        const int dw = t / ROWS, rr = t % ROWS;
        if (t + 2 < NA) {
          if ((t + 2) % ROWS == 0) ldb((t + 2) / ROWS);
          lda(t + 2);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int dh = 0; dh < 3; ++dh) {
          const int i = rr - dh;
          if (i >= 0 && i < MT) {
#pragma unroll
            for (int n = 0; n < 2; ++n)
              acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av[t % 3]),
                                                                  __builtin_bit_cast(bf16x8, bw[dw & 1][dh][n]), acc[i][n], 0, 0, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    for (int i = 0; i < MT; ++i)
      for (int n = 0; n < 2; ++n)
        for (int e = 0; e < 4; ++e) s += acc[i][n][e];
  } else {
    f32x16 acc[9];
    for (int i = 0; i < 9; ++i)
      for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int t = 0; t < 8; ++t) {    // 8 k-steps x 9 MFMAs (3 frames x 3 channel tiles) = 72 MFMAs = 144 of the small ones
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 9; ++i)
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i / 3], b[(t + i) % 6], acc[i], 0, 0, 0);
      }
    }
    for (int i = 0; i < 9; ++i)
      for (int e = 0; e < 16; ++e) s += acc[i][e];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + (smem[0] == 77 ? 1.f : 0.f);
}

template <int SHAPE, int THREADS>
void run(int lds, const float* seed, float* out) {
  const int iters = 400, blocks = 256 * 8;
  hipFuncSetAttribute((const void*)k<SHAPE, THREADS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<SHAPE, THREADS>), dim3(blocks), dim3(THREADS), lds, 0, seed, out, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k<SHAPE, THREADS>), dim3(blocks), dim3(THREADS), lds, 0, seed, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
  const double flops = (double)blocks * (THREADS / 64) * iters * (SHAPE == 17 ? 72 : 144) * (16.0 * 16 * 32 * 2);
  printf("  %s, %d waves/SIMD %s: %8.3f ms  %7.1f TF/s\n",
         SHAPE == 16 ? "16x16x32" : SHAPE == 17 ? "16x16x32 + LDS reads, 4 rows/wave" : SHAPE == 18 ? "16x16x32 + LDS reads, 8 rows/wave" : "32x32x16", THREADS / 256,
         lds ? "(1 WG/CU)" : "(no LDS limit)", ms, flops / ms / 1e9);
}

int main() {
  float* seed; float* out;
  hipMalloc(&seed, 4096); hipMalloc(&out, 16 << 20);
  float h[1024];
  srand(1);
  for (int z = 0; z < 2; ++z) {
    for (int i = 0; i < 1024; ++i) h[i] = z ? 0.f : (float)rand() / RAND_MAX * 2.f - 1.f;
    hipMemcpy(seed, h, 4096, hipMemcpyHostToDevice);
    printf(z ? "-- zero operands\n" : "-- random operands\n");
    const int L = 148 * 1024;
    run<16, 256>(L, seed, out); run<16, 512>(L, seed, out); run<16, 768>(L, seed, out); run<16, 1024>(L, seed, out);
    run<16, 512>(0, seed, out);
    run<18, 512>(L, seed, out); run<17, 1024>(L, seed, out); run<17, 768>(L, seed, out);
    run<32, 256>(L, seed, out); run<32, 512>(L, seed, out);
    run<32, 512>(0, seed, out);
  }
  return 0;
}
