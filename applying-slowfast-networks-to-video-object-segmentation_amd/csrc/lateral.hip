// lateral.hip -- data gradient of the time-strided lateral convolution conv_f2s* (k x 1 x 1, 32 -> 64 channels,
// reference code/helpers/model.py:82-94,112): grad_input of aten::convolution_backward for that layer.
//
//   dx[t][px][c] (+)= sum_{f, n} dy[f][px][n] * w[n][c][t - f]        0 <= t - f < kt
//
// dy has T_slow (2-3) frames of 64 channels, dx has T_fast (12-22) frames of 32: every output frame is at most
// T_slow small GEMMs (K = 64) per pixel -- 23 GFLOP per clip against 0.28 GB of traffic, i.e. HBM-bound.  The
// generic conv kernel runs it as 12 nearly empty barrier stages per workgroup; here a wave owns 32 pixels for ALL
// output frames: the dy fragments of its pixels live in registers, weight fragments stream from L2, each output
// frame is read (accumulate), updated and written once.  No LDS, no barrier.
//
// MFMA roles (v_mfma_f32_16x16x32_bf16): D[c][px] = A[c][n] * B[n][px]; A = 16 output channels x 32 reduction
// channels of the packed data-gradient weight image (sfvos_pack_weights_dgrad: [chunk][dt][tap][j][c][8]: one
// 16-byte run per lane), B = 32 reduction channels x 16 pixels of dy (16 contiguous bytes per lane in NDHWC).
// A lane of D holds 4 consecutive channels of one pixel: 8-byte stores.
#include "common.h"

namespace sfvos {

struct LatLevels {
  int n;
  int HW[SFVOS_MAX_LEVELS];
  int wave_begin[SFVOS_MAX_LEVELS + 1];  // first 32-pixel wave tile of each level (tiles enumerate level, clip, px)
  long long xpos[SFVOS_MAX_LEVELS];      // first position of the level in the dy / dx pyramid buffers
  long long ypos[SFVOS_MAX_LEVELS];
};

struct LatArgs {
  const char* dy;   // conv "x": t_in frames, c_in channels, pitch ld_dy
  const char* wp;   // packed data-gradient image
  char* dx;         // conv "y": t_out = t_in + kt - 1 frames, 32 channels, pitch ld_dx
  int t_in, t_out, kt, ld_dy, ld_dx, accumulate, batch, n_waves;
  LatLevels lv;
};

// KS = c_in / 32 reduction chunks, TIN = frames of dy (compile-time: the fragments are a register array)
template <int KS, int TIN>
__global__ __launch_bounds__(256) void lateral_dgrad_kernel(LatArgs a) {
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tile = blockIdx.x * 4 + wv;
  if (tile >= a.n_waves) return;
  int lvl = 0;
#pragma unroll
  for (int l = 1; l < SFVOS_MAX_LEVELS; ++l)
    if (l < a.lv.n && tile >= a.lv.wave_begin[l]) lvl = l;
  const int HW = a.lv.HW[lvl];
  const int per_clip = (HW + 31) >> 5;
  const int k = tile - a.lv.wave_begin[lvl];
  const int b = k / per_clip, px0 = (k - b * per_clip) * 32;
  const int p16 = lane & 15, g = lane >> 4;

  // dy fragments: [frame][chunk][pixel half]
  u32x4 dyf[TIN][KS][2];
#pragma unroll
  for (int f = 0; f < TIN; ++f)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const int px = px0 + nt * 16 + p16;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (px < HW)
          v = *(const u32x4*)(a.dy + ((a.lv.xpos[lvl] + ((long long)b * a.t_in + f) * HW + px) * a.ld_dy + ks * 32 + g * 8) * 2);
        dyf[f][ks][nt] = v;
      }

  const char* wl = a.wp + (g * 32 + p16) * 16;  // lane part of a weight fragment address
  // this lane's 4-channel runs of output frame t: [pixel half][channel half]
  auto dst_of = [&](int t, int nt, int mt) {
    const int px = px0 + nt * 16 + p16;
    return (bf16x4*)(a.dx + ((a.lv.ypos[lvl] + ((long long)b * a.t_out + t) * HW + px) * a.ld_dx + mt * 16 + 4 * g) * 2);
  };
  const bool in0 = px0 + p16 < HW, in1 = px0 + 16 + p16 < HW;
  bf16x4 old[2][2];  // accumulate mode: frame t's previous contents, fetched one frame ahead of its use
  auto fetch_old = [&](int t) {
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        bf16x4 v = {(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
        if (a.accumulate && (nt ? in1 : in0)) v = *dst_of(t, nt, mt);
        old[nt][mt] = v;
      }
  };
  fetch_old(0);
  for (int t = 0; t < a.t_out; ++t) {
    bf16x4 cur[2][2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) cur[nt][mt] = old[nt][mt];
    if (t + 1 < a.t_out) fetch_old(t + 1);
    f32x4 acc[2][2];  // [channel half][pixel half]
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    // out[t] = sum_dt' in[t - (kt-1) + dt'] * Wimg[dt']  (the data-gradient image is already flipped in time)
#pragma unroll
    for (int f = 0; f < TIN; ++f) {
      const int dt = f + a.kt - 1 - t;
      if (dt < 0 || dt >= a.kt) continue;  // wave-uniform
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          const u32x4 w = *(const u32x4*)(wl + ((long long)((ks * a.kt + dt) * 4) * 32 + mt * 16) * 16);
#pragma unroll
          for (int nt = 0; nt < 2; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w),
                                                                  __builtin_bit_cast(bf16x8, dyf[f][ks][nt]),
                                                                  acc[mt][nt], 0, 0, 0);
        }
    }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      if (!(nt ? in1 : in0)) continue;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        f32x4 v = acc[mt][nt];
        const bf16x4 o0 = cur[nt][mt];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += (float)o0[e];  // zeros unless accumulating
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (__bf16)v[e];
        *dst_of(t, nt, mt) = o;
      }
    }
  }
}

// Called by sfvos_conv3d for the shapes this kernel covers; returns -1 when it does not apply.
int lateral_dgrad_try(const sfvos_conv_desc* d, const void* x, const void* w_packed, void* y, hipStream_t stream) {
  if (!(d->dtype == SFVOS_BF16 && d->taps == 1 && d->c_out == 32 && d->kt >= 1 && d->pad_t == d->kt - 1 &&
        (d->c_in == 32 || d->c_in == 64) && d->t_in >= 1 && d->t_in <= 3 && d->t_offset == 0 && d->t_alloc == d->t_in &&
        d->ld_x % 8 == 0 && d->ld_y % 4 == 0))
    return -1;
  LatArgs a;
  a.dy = (const char*)x; a.wp = (const char*)w_packed; a.dx = (char*)y;
  a.t_in = d->t_in; a.t_out = d->t_in + d->kt - 1; a.kt = d->kt; a.ld_dy = d->ld_x; a.ld_dx = d->ld_y;
  a.accumulate = d->accumulate; a.batch = d->batch;
  a.lv.n = d->pyr.n_levels;
  long long waves = 0, px = 0;
  for (int l = 0; l < SFVOS_MAX_LEVELS; ++l) {
    const bool live = l < a.lv.n;
    const int HW = live ? d->pyr.h[l] * d->pyr.w[l] : 1;
    a.lv.HW[l] = HW;
    a.lv.wave_begin[l] = (int)waves;
    a.lv.xpos[l] = (long long)d->batch * a.t_in * px;
    a.lv.ypos[l] = (long long)d->batch * a.t_out * px;
    if (live) {
      waves += (long long)d->batch * ceil_div(HW, 32);
      px += HW;
    }
  }
  a.lv.wave_begin[SFVOS_MAX_LEVELS] = (int)waves;
  if (waves <= 0 || waves >= (1ll << 30)) return -1;
  a.n_waves = (int)waves;
  const dim3 grid((unsigned)ceil_div64(waves, 4)), block(256);
  const int ks = d->c_in / 32;
#define SFVOS_LAT(KSv, TINv) \
  if (ks == KSv && d->t_in == TINv) { hipLaunchKernelGGL((lateral_dgrad_kernel<KSv, TINv>), grid, block, 0, stream, a); return check_launch("lateral_dgrad"); }
  SFVOS_LAT(2, 1) SFVOS_LAT(2, 2) SFVOS_LAT(2, 3) SFVOS_LAT(1, 1) SFVOS_LAT(1, 2) SFVOS_LAT(1, 3)
#undef SFVOS_LAT
  return -1;
}

}  // namespace sfvos
