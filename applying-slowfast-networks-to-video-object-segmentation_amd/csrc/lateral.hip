// lateral.hip -- forward and data gradient of the time-strided lateral convolution conv_f2s* (k x 1 x 1, 32 -> 64
// channels, reference code/helpers/model.py:82-94,112).  Data gradient (grad_input of aten::convolution_backward):
//
//   dx[t][px][c] (+)= sum_{f, n} dy[f][px][n] * w[n][c][t - f]        0 <= t - f < kt
//
// dy has T_slow (2-3) frames of 64 channels, dx has T_fast (12-22) frames of 32: every output frame is at most
// T_slow small GEMMs (K = 64) per pixel -- 23 GFLOP per clip against 0.28 GB of traffic, i.e. HBM-bound.  The
// generic conv kernel runs it as 12 nearly empty barrier stages per workgroup; here a wave owns 32 pixels for ALL
// output frames: the dy fragments of its pixels live in registers, the weight image (<= 80 KB) is copied into LDS
// once per workgroup (round 1 streamed the fragments from L2: 6 bytes of L2 traffic per byte of HBM traffic), each
// output frame is read (accumulate), updated and written once.  One barrier, after the copy.
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

constexpr int LAT_DG_PF = 3;       // output frames whose previous contents are in flight (accumulate mode)
constexpr int LAT_DG_MAX_LDS = 80 * 1024;   // weight image up to here: 8-wave workgroups, two per compute unit
constexpr int LAT_DG_BIG_LDS = 160 * 1024;  // ... up to here: 12-wave workgroups, one per compute unit
// MODE: 0 = weight fragments streamed from L2 (no LDS), 1 = image in LDS (<= 80 KB), 2 = image in LDS (<= 160 KB),
// 3 = all taps but the last in LDS (160 KB) and the last tap's fragments in registers: the (4,64) configuration's
// kt = 41 x 64 reduction channels is 164 KB, one tap more than a compute unit's LDS holds.
__host__ __device__ constexpr int lat_dg_nw(int mode) { return mode >= 2 ? 12 : 8; }

struct LatArgs {
  const char* dy;   // conv "x": t_in frames, c_in channels, pitch ld_dy
  const char* wp;   // packed data-gradient image
  char* dx;         // conv "y": t_out = t_in + kt - 1 frames, 32 channels, pitch ld_dx
  int t_in, t_out, kt, ld_dy, ld_dx, accumulate, batch, n_waves;
  int kl;           // taps of the weight image held in LDS (MODE 1, 2: kt; MODE 3: kt - 1)
  LatLevels lv;
};

// KS = c_in / 32 reduction chunks, TIN = frames of dy (compile-time: the fragments are a register array)
// WL: the weight image fits the LDS budget (<= 80 KB: kt <= 20 for 64 channels) and is staged there; else (the (4,64)
// configuration's kt = 41) the fragments stream from L2 as in round 1.
template <int KS, int TIN, bool ACC, int MODE>
__global__ __launch_bounds__(64 * lat_dg_nw(MODE)) void lateral_dgrad_kernel(LatArgs a) {
  constexpr bool WL = MODE >= 1, WREG = MODE == 3;
  constexpr int LAT_DG_NW = lat_dg_nw(MODE);
  extern __shared__ __attribute__((aligned(16))) char smem[];  // the weight image: [chunk][dt < kl][j][c][16 B]
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int KL = WL ? a.kl : a.kt;   // temporal stride of the image the fragments are read from
  // weight image -> LDS by LDS-DMA: wave wv copies the 1 KB pieces wv, wv + NW, ... (lane i lands at +16 i)
  if (WL)
    for (int q = wv; q < KS * KL * 2; q += LAT_DG_NW) {
      const int ks = q / (KL * 2);   // the taps [0, KL) of each reduction chunk are contiguous in the packed image
      glds16(a.wp + (ks * a.kt * 2 + (q - ks * KL * 2)) * 1024 + lane * 16, smem + q * 1024);
    }
  const bool live = (int)blockIdx.x * LAT_DG_NW + wv < a.n_waves;  // idle waves redo the last tile without storing
  const int tile = live ? blockIdx.x * LAT_DG_NW + wv : a.n_waves - 1;
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
        const int px = px0 + nt * 16 + p16 < HW ? px0 + nt * 16 + p16 : HW - 1;  // clamped: never stored
        dyf[f][ks][nt] =
            *(const u32x4*)(a.dy + ((a.lv.xpos[lvl] + ((long long)b * a.t_in + f) * HW + px) * a.ld_dy + ks * 32 + g * 8) * 2);
      }

  const char* wl = (WL ? (const char*)smem : a.wp) + (g * 32 + p16) * 16;  // lane part of a weight fragment address
  u32x4 wreg[WREG ? KS : 1][2];   // MODE 3: the fragments of tap kt - 1
  if (WREG) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
        wreg[ks][mt] = *(const u32x4*)(a.wp + (g * 32 + p16) * 16 + (long long)(((ks * a.kt + KL) * 4) * 32 + mt * 16) * 16);
  }
  // this lane's 4-channel runs of an output frame: [pixel half][channel half].  Loads of the previous contents are
  // unconditional (pixels beyond the level and frames beyond the clip clamped to the last valid one): a branch
  // around a load makes the compiler wait for every outstanding load before the next use.
  const bool in0 = live && px0 + p16 < HW, in1 = live && px0 + 16 + p16 < HW;
  const int pa = px0 + p16 < HW ? px0 + p16 : HW - 1, pb = px0 + 16 + p16 < HW ? px0 + 16 + p16 : HW - 1;
  const long long dframe = (long long)HW * a.ld_dx * 2;  // bytes
  char* const dc = a.dx + ((a.lv.ypos[lvl] + (long long)b * a.t_out * HW) * a.ld_dx + 4 * g) * 2;
  char* const dl0 = dc + (long long)pa * a.ld_dx * 2;
  char* const dl1 = dc + (long long)pb * a.ld_dx * 2;
  bf16x4 old[LAT_DG_PF][2][2];  // accumulate mode: previous contents of the next LAT_DG_PF frames
  auto fetch_old = [&](int t, bf16x4 (&o)[2][2]) {
    const int tc = t < a.t_out ? t : a.t_out - 1;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      o[0][mt] = *(const bf16x4*)(dl0 + tc * dframe + mt * 32);
      o[1][mt] = *(const bf16x4*)(dl1 + tc * dframe + mt * 32);
    }
  };
#pragma unroll
  for (int u = 0; u < LAT_DG_PF; ++u) {
    if (ACC) {
      fetch_old(u, old[u]);
    } else {
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) old[u][nt][mt] = bf16x4{(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
    }
  }
  if (WL) __syncthreads();  // weight image complete
  for (int t0 = 0; t0 < a.t_out; t0 += LAT_DG_PF) {
#pragma unroll
    for (int u = 0; u < LAT_DG_PF; ++u) {
      const int t = t0 + u;
      if (t < a.t_out) {  // wave-uniform
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
              u32x4 w;
              if (WREG && dt == KL) w = wreg[WREG ? ks : 0][mt];   // wave-uniform
              else w = *(const u32x4*)(wl + (long long)(((ks * KL + dt) * 4) * 32 + mt * 16) * 16);
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
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (__bf16)(v[e] + (float)old[u][nt][mt][e]);  // zeros unless accumulating
            *(bf16x4*)((nt ? dl1 : dl0) + t * dframe + mt * 32) = o;
          }
        }
      }
      if (ACC) fetch_old(t + LAT_DG_PF, old[u]);
    }
  }
}

// ---- forward of the lateral convolution (k x 1 x 1, 32 -> 64 channels, model.py:82-94,112) -------------------
//   y[f][px][n] = bias[n] + sum_{dt < kt, c < 32} x[f + dt][px][c] * w[n][c][dt]        f < t_out = t_in - kt + 1
// 21 GFLOP per clip against 0.15 GB: HBM-bound.  The generic conv kernel runs it as one barrier stage per temporal
// tap with two MFMAs per wave in it.  Here the WHOLE weight image (kt x 4 KB) is copied into LDS once per
// workgroup; a wave owns 32 pixels for all frames, streams its input frames once (16 bytes per lane, LAT_PF frames
// in flight) and keeps the t_out x 64-channel results of its pixels in registers: every input frame feeds the
// (at most t_out) output frames it belongs to.  One barrier after the weight copy, one for the statistics.
//
// MFMA roles: D[n][px] = A[n][c] * B[c][px]; A = 16 output channels x 32 input channels of the packed forward image
// (sfvos_pack_weights_fwd: [dt][j][n][8]: one 16-byte run per lane, 16 lanes = 256 contiguous bytes of LDS), B = 32
// channels x 16 pixels of x.  Per-channel (sum, sum of squares) of the fp32 results leave as ONE row per workgroup
// (BatchNorm statistics); workgroups are level-aligned so that a row belongs to one level.
struct LatFwdArgs {
  const char* x; const char* wp; const float* bias; char* y; float* stat_part;
  int t_in, t_alloc, t_offset, t_out, kt, ld_x, ld_y, batch, relu;
  int n;                               // levels
  int HW[SFVOS_MAX_LEVELS];
  int wg_begin[SFVOS_MAX_LEVELS + 1];  // first workgroup (= statistics row) of each level
  long long xpos[SFVOS_MAX_LEVELS], ypos[SFVOS_MAX_LEVELS];
};

// NW waves per workgroup; three waves per SIMD (at most 168 registers) so that the clip is ONE round of workgroups
// WREG: the weight image has ONE tap more than the 40 that 160 KB of LDS hold (the (4,64) configuration's kt = 41): the
// fragments of tap 40 stay in registers (16 per lane) and one input frame less is kept in flight.
template <int TOUT, int NW, bool WREG>
__global__ __launch_bounds__(64 * NW, 3) void lateral_fwd_kernel(LatFwdArgs a) {
  constexpr int LAT_PF = WREG ? 3 : 4;  // input frames in flight per wave
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [kt][4 chunks][64 n][16 B]; later [NW][2][64] floats
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int KL = WREG ? a.kt - 1 : a.kt;   // taps held in LDS
  // weight image -> LDS by LDS-DMA: wave wv copies the 1 KB pieces wv, wv + NW, ... (lane i lands at +16 i)
  for (int q = wv; q < KL * 4; q += NW) glds16(a.wp + q * 1024 + lane * 16, smem + q * 1024);
  int lvl = 0;
#pragma unroll
  for (int l = 1; l < SFVOS_MAX_LEVELS; ++l)
    if (l < a.n && (int)blockIdx.x >= a.wg_begin[l]) lvl = l;
  const int HW = a.HW[lvl];
  const int per_clip = (HW + 31) >> 5;
  const int tile = ((int)blockIdx.x - a.wg_begin[lvl]) * NW + wv;
  const bool live = tile < a.batch * per_clip;  // wave-uniform; idle waves still join the barriers
  const int b = live ? tile / per_clip : 0, px0 = live ? (tile - b * per_clip) * 32 : 0;
  const int p16 = lane & 15, g = lane >> 4;
  const bool in0 = live && px0 + p16 < HW, in1 = live && px0 + 16 + p16 < HW;

  f32x4 acc[TOUT][2][4];  // [output frame][pixel half][16-channel tile]
#pragma unroll
  for (int f = 0; f < TOUT; ++f)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) acc[f][nt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};

  // x loads are unconditional (a branch around a load makes the compiler wait for ALL loads before the next MFMA):
  // pixels beyond the level and frames beyond the clip are clamped to the last valid one; their results are
  // neither stored nor counted.
  const long long xframe = (long long)HW * a.ld_x * 2;  // bytes
  const int pa = px0 + p16 < HW ? px0 + p16 : HW - 1, pb = px0 + 16 + p16 < HW ? px0 + 16 + p16 : HW - 1;
  const char* xc = a.x + ((a.xpos[lvl] + ((long long)b * a.t_alloc + a.t_offset) * HW) * a.ld_x + g * 8) * 2;
  const char* xl0 = xc + (long long)pa * a.ld_x * 2;
  const char* xl1 = xc + (long long)pb * a.ld_x * 2;
  u32x4 xf[LAT_PF][2];
  auto load_x = [&](int t, u32x4 (&v)[2]) {
    const int tc = t < a.t_in ? t : a.t_in - 1;
    v[0] = *(const u32x4*)(xl0 + tc * xframe);
    v[1] = *(const u32x4*)(xl1 + tc * xframe);
  };
#pragma unroll
  for (int u = 0; u < LAT_PF; ++u) load_x(u, xf[u]);
  u32x4 wreg[WREG ? 4 : 1];   // the fragments of tap KL
  if (WREG) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) wreg[mt] = *(const u32x4*)(a.wp + (g * 64 + p16) * 16 + (KL * 256 + mt * 16) * 16);
  }
  __syncthreads();  // weight image complete
  const char* wl = smem + (g * 64 + p16) * 16;
  for (int t0 = 0; t0 < a.t_in; t0 += LAT_PF) {
#pragma unroll
    for (int u = 0; u < LAT_PF; ++u) {
      const int t = t0 + u;
      if (t < a.t_in) {  // wave-uniform
#pragma unroll
        for (int f = 0; f < TOUT; ++f) {
          const int dt = t - f;
          if (dt < 0 || dt >= a.kt) continue;  // wave-uniform
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) {
            u32x4 w;
            if (WREG && dt == KL) w = wreg[WREG ? mt : 0];   // wave-uniform
            else w = *(const u32x4*)(wl + (dt * 256 + mt * 16) * 16);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
              acc[f][nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w),
                                                                       __builtin_bit_cast(bf16x8, xf[u][nt]),
                                                                       acc[f][nt][mt], 0, 0, 0);
          }
        }
      }
      load_x(t + LAT_PF, xf[u]);
    }
  }

  // epilogue: + bias, optional ReLU, statistics of the fp32 values, 8-byte stores
  float s1[4][4], s2[4][4];  // [16-channel tile][e]: channel mt*16 + 4g + e
  f32x4 bias4[4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      s1[mt][e] = 0.f; s2[mt][e] = 0.f;
      bias4[mt][e] = a.bias ? a.bias[mt * 16 + 4 * g + e] : 0.f;
    }
  }
  char* yl = a.y + ((a.ypos[lvl] + (long long)b * a.t_out * HW + px0 + p16) * a.ld_y + 4 * g) * 2;
#pragma unroll
  for (int f = 0; f < TOUT; ++f)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      if (!(nt ? in1 : in0)) continue;
      char* dst = yl + ((long long)f * HW + nt * 16) * a.ld_y * 2;
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float v = acc[f][nt][mt][e] + bias4[mt][e];
          if (a.relu) v = fmaxf(v, 0.f);
          s1[mt][e] += v; s2[mt][e] += v * v;
          o[e] = (__bf16)v;
        }
        *(bf16x4*)(dst + mt * 32) = o;
      }
    }
  if (a.stat_part) {  // kernel argument: uniform
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int m = 1; m < 16; m <<= 1) {
          s1[mt][e] += __shfl_xor(s1[mt][e], m);
          s2[mt][e] += __shfl_xor(s2[mt][e], m);
        }
    __syncthreads();  // every wave is done with the weight image
    float* red = (float*)smem;  // [NW][2][64]
    if (p16 == 0) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          red[(wv * 2 + 0) * 64 + mt * 16 + 4 * g + e] = s1[mt][e];
          red[(wv * 2 + 1) * 64 + mt * 16 + 4 * g + e] = s2[mt][e];
        }
    }
    __syncthreads();
    if (tid < 128) {
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < NW; ++k) t += red[k * 128 + tid];
      a.stat_part[(long long)blockIdx.x * 128 + tid] = t;  // row = workgroup: [2][64]
    }
  }
}

constexpr int LAT_FWD_LDS_KT = 40;                  // taps 160 KB of LDS hold (4 KB each)
constexpr int LAT_FWD_MAX_KT = LAT_FWD_LDS_KT + 1;  // ... plus one tap in registers
// waves per workgroup: 12 waves per compute unit as 3 x 4 (weights <= 52 KB), 2 x 6 (<= 80 KB) or 1 x 12
static int lateral_fwd_nw(const sfvos_conv_desc* d) { return d->kt <= 13 ? 4 : d->kt <= 20 ? 6 : 12; }

// The shapes the forward kernel covers (make_plan asks: the statistics rows are per workgroup of THIS kernel).
bool lateral_fwd_applies(const sfvos_conv_desc* d) {
  const int t_out = d->t_in - d->kt + 1;
  return d->dtype == SFVOS_BF16 && d->taps == 1 && d->c_in == 32 && d->c_out == 64 && d->pad_t == 0 && d->kt >= 1 &&
         d->kt <= LAT_FWD_MAX_KT && t_out >= 1 && t_out <= 3 && d->accumulate == 0 && d->x_group_stride == 0 &&
         d->x_frame_stride == 0 && d->y_frame_stride == 0 && d->t_offset >= 0 && d->t_offset + d->t_in <= d->t_alloc &&
         d->ld_x % 8 == 0 && d->ld_y % 4 == 0;
}

// workgroups (= statistics rows) of level l
int lateral_fwd_rows(const sfvos_conv_desc* d, int l) {
  return (int)ceil_div64((long long)d->batch * ceil_div(d->pyr.h[l] * d->pyr.w[l], 32), lateral_fwd_nw(d));
}

int lateral_fwd_launch(const sfvos_conv_desc* d, const void* x, const void* w_packed, const float* bias, void* y,
                       float* stat_part, hipStream_t stream) {
  LatFwdArgs a;
  a.x = (const char*)x; a.wp = (const char*)w_packed; a.bias = bias; a.y = (char*)y; a.stat_part = stat_part;
  a.t_in = d->t_in; a.t_alloc = d->t_alloc; a.t_offset = d->t_offset; a.t_out = d->t_in - d->kt + 1; a.kt = d->kt;
  a.ld_x = d->ld_x; a.ld_y = d->ld_y; a.batch = d->batch; a.relu = d->relu;
  a.n = d->pyr.n_levels;
  long long wg = 0, px = 0;
  for (int l = 0; l < SFVOS_MAX_LEVELS; ++l) {
    const bool live = l < a.n;
    const int HW = live ? d->pyr.h[l] * d->pyr.w[l] : 1;
    a.HW[l] = HW;
    a.wg_begin[l] = (int)wg;
    a.xpos[l] = (long long)d->batch * a.t_alloc * px;
    a.ypos[l] = (long long)d->batch * a.t_out * px;
    if (live) {
      wg += lateral_fwd_rows(d, l);
      px += HW;
    }
  }
  a.wg_begin[SFVOS_MAX_LEVELS] = (int)wg;
  SFVOS_REQUIRE(wg > 0 && wg < (1ll << 30), "conv: grid out of range");
  const int nw = lateral_fwd_nw(d);
  const bool wreg = a.kt > LAT_FWD_LDS_KT;
  const int kl = wreg ? a.kt - 1 : a.kt;
  const int lds = kl * 4096 > nw * 512 ? kl * 4096 : nw * 512;
  const dim3 grid((unsigned)wg);
#define SFVOS_LATF(TOUTv, NWv, WREGv)                                                                       \
  if (a.t_out == TOUTv && nw == NWv && wreg == WREGv) {                                                     \
    auto kern = lateral_fwd_kernel<TOUTv, NWv, WREGv>;                                                      \
    static LdsAttrOnce once;                                                                                \
    if (int rc = once.ensure((const void*)kern, LAT_FWD_LDS_KT * 4096, "lateral_fwd")) return rc;           \
    hipLaunchKernelGGL(kern, grid, dim3(64 * NWv), lds, stream, a);                                         \
    return check_launch("lateral_fwd");                                                                     \
  }
  SFVOS_LATF(1, 4, false) SFVOS_LATF(2, 4, false) SFVOS_LATF(3, 4, false)
  SFVOS_LATF(1, 6, false) SFVOS_LATF(2, 6, false) SFVOS_LATF(3, 6, false)
  SFVOS_LATF(1, 12, false) SFVOS_LATF(2, 12, false) SFVOS_LATF(3, 12, false)
  SFVOS_LATF(1, 12, true) SFVOS_LATF(2, 12, true) SFVOS_LATF(3, 12, true)
#undef SFVOS_LATF
  return SFVOS_E_ARG;
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
  const int ks = d->c_in / 32;
  const int image = ks * d->kt * 2048;  // the packed data-gradient image
  // where the weight fragments come from: LDS (two small workgroups or one big one per compute unit), LDS + the last tap
  // in registers, or -- images beyond that -- streamed from L2
  int mode = 0;
  a.kl = d->kt;
  if (image <= LAT_DG_MAX_LDS) mode = 1;
  else if (image <= LAT_DG_BIG_LDS) mode = 2;
  else if (image - ks * 2048 <= LAT_DG_BIG_LDS) { mode = 3; a.kl = d->kt - 1; }
  const int nw = lat_dg_nw(mode);
  const dim3 grid((unsigned)ceil_div64(waves, nw)), block(64 * nw);
  const int lds = mode ? ks * a.kl * 2048 : 0;
#define SFVOS_LAT3(KSv, TINv, ACCv, MODEv)                                                         \
  if (ks == KSv && d->t_in == TINv && (d->accumulate != 0) == ACCv && mode == MODEv) {             \
    auto kern = lateral_dgrad_kernel<KSv, TINv, ACCv, MODEv>;                                      \
    static LdsAttrOnce once;                                                                       \
    if (int rc = once.ensure((const void*)kern, LAT_DG_BIG_LDS, "lateral_dgrad")) return rc;       \
    hipLaunchKernelGGL(kern, grid, block, lds, stream, a);                                         \
    return check_launch("lateral_dgrad");                                                          \
  }
#define SFVOS_LAT2(KSv, TINv, ACCv) \
  SFVOS_LAT3(KSv, TINv, ACCv, 0) SFVOS_LAT3(KSv, TINv, ACCv, 1) SFVOS_LAT3(KSv, TINv, ACCv, 2) SFVOS_LAT3(KSv, TINv, ACCv, 3)
#define SFVOS_LAT(KSv, TINv) SFVOS_LAT2(KSv, TINv, true) SFVOS_LAT2(KSv, TINv, false)
  SFVOS_LAT(2, 1) SFVOS_LAT(2, 2) SFVOS_LAT(2, 3) SFVOS_LAT(1, 1) SFVOS_LAT(1, 2) SFVOS_LAT(1, 3)
#undef SFVOS_LAT
#undef SFVOS_LAT2
#undef SFVOS_LAT3
  return -1;
}

}  // namespace sfvos
