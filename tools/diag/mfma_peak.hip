// Calibration: what bf16 MFMA rate does this chip sustain on random operands (no memory traffic)?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int SHAPE, int NACC>
__global__ __launch_bounds__(512) void k(const float* seed, float* out, int iters) {
  bf16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 8; ++j) {
      a[i][j] = (__bf16)seed[(threadIdx.x * 37 + i * 8 + j) & 1023];
      b[i][j] = (__bf16)seed[(threadIdx.x * 53 + i * 8 + j + 7) & 1023];
    }
  if constexpr (SHAPE == 32) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
      for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[u], b[(u + i) & 3], acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i)
      for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  } else {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i)
      for (int e = 0; e < 4; ++e) acc[i][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[u], b[(u + i) & 3], acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i)
      for (int e = 0; e < 4; ++e) s += acc[i][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  }
}

template <int SHAPE, int NACC>
void run(const char* name, int threads, int blocks, const float* seed, float* out) {
  const int iters = 4000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<SHAPE, NACC>), dim3(blocks), dim3(threads), 0, 0, seed, out, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<SHAPE, NACC>), dim3(blocks), dim3(threads), 0, 0, seed, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  const double flop_per_mfma = SHAPE == 32 ? 32.0 * 32 * 16 * 2 : 16.0 * 16 * 32 * 2;
  const double flops = (double)blocks * (threads / 64) * iters * 4 * NACC * flop_per_mfma;
  printf("%-34s threads %4d blocks %5d : %8.3f ms  %8.1f TF/s\n", name, threads, blocks, ms, flops / ms / 1e9);
}

int main() {
  float* seed; float* out;
  hipMalloc(&seed, 4096); hipMalloc(&out, 4 << 20);
  float h[1024];
  srand(1);
  for (int i = 0; i < 1024; ++i) h[i] = (float)rand() / RAND_MAX * 2.f - 1.f;
  hipMemcpy(seed, h, 4096, hipMemcpyHostToDevice);
  for (int rep = 0; rep < 2; ++rep) {
    run<32, 4>("32x32x16 bf16, 4 acc, 2 waves/SIMD", 512, 256 * 4, seed, out);
    run<32, 4>("32x32x16 bf16, 4 acc, 1 wave/SIMD", 256, 256 * 4, seed, out);
    run<16, 8>("16x16x32 bf16, 8 acc, 2 waves/SIMD", 512, 256 * 4, seed, out);
    run<16, 8>("16x16x32 bf16, 8 acc, 1 wave/SIMD", 256, 256 * 4, seed, out);
  }
  // zero operands (higher clock)
  for (int i = 0; i < 1024; ++i) h[i] = 0.f;
  hipMemcpy(seed, h, 4096, hipMemcpyHostToDevice);
  run<32, 4>("32x32x16 ZEROS, 2 waves/SIMD", 512, 256 * 4, seed, out);
  run<16, 8>("16x16x32 ZEROS, 2 waves/SIMD", 512, 256 * 4, seed, out);
  return 0;
}
