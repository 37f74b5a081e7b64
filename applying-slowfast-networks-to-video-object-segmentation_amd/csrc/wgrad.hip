// wgrad.hip -- weight gradient of the kt x 3 x 3 / k x 1 x 1 NDHWC convolution
// (aten::convolution_backward grad_weight for reference code/helpers/model.py:72-78,83-92).
//
//   dW[n][c][dt][tap] = sum_{b,t,h,w} dy[b,t,h,w,n] * x[b,t+dt,h+dh-1,w+dw-1,c]
//
// GEMM view: D[n (32 rows)][c (32 cols)] += A[n][k = pixel] * B[k = pixel][c]; the reduction runs
// over pixels, so BOTH operands are "k-major" in NDHWC.  bf16: tiles sit in LDS as
// [pixel][32 channels] (64-B rows) and are read with ds_read_b64_tr_b16 (hardware transpose: a
// half-wave touches 4 consecutive 64-B rows = one 256-B bank row, conflict-free); f32: the exact
// 32x32x2 MFMA takes one scalar per lane, read with ds_read_b32 (32 consecutive floats per half).
//
// One workgroup (8 waves) owns NTN n-tiles x NTC c-tiles x DG temporal taps x all spatial taps;
// wave = one (n-tile, c-tile, dt) group with TAPS accumulator tiles.  It sweeps its share of the
// TH x 16 pixel tiles; for each tile it walks the input frames t: stage (tile, t) holds the halo
// tile of x[t] (double-buffered) and a ring of the DG+1 newest dy frames, so x[t] is paired with
// dy[t-dt] for all DG taps of the group from ONE load (x traffic / DG), and exactly one x tile +
// one dy frame are DMA'd per stage (global_load_lds, one barrier per stage).
// Split-K partials go to fp32 slabs, summed in a fixed order by wgrad_reduce_kernel
// (deterministic, no atomics), which also transposes into the state-dict layout
// [Cout][Cin][kt][kh][kw].
#include "common.h"

namespace sfvos {

struct WgradLevels {
  int n;
  int H[SFVOS_MAX_LEVELS], W[SFVOS_MAX_LEVELS], tiles_h[SFVOS_MAX_LEVELS], tiles_w[SFVOS_MAX_LEVELS];
  int tile_begin[SFVOS_MAX_LEVELS + 1];  // first pixel tile of each level (tiles enumerate level, clip, th, tw)
  long long xpos[SFVOS_MAX_LEVELS];      // first position of the level in the x / dy pyramid buffers
  long long ypos[SFVOS_MAX_LEVELS];
};

struct WgradArgs {
  const char* x;
  const char* dy;
  float* slab;
  const char* zeros;
  int t_in, t_alloc, t_offset, t_out, c_in, c_out, kt, ld_x, ld_y, batch;
  int n_blocks, c_blocks, dt_blocks, psplit;
  int ntiles;  // over all levels and clips
  WgradLevels lv;
};

template <int DT, int TAPS, int NTN, int NTC, int DG, int TH>
struct WgradCfg {
  static constexpr int CE = Elt<DT>::CE;
  static constexpr int SPP = 32 / CE;            // 16-B slots per pixel per 32-channel tile
  static constexpr int ROWB = SPP * 16;          // bytes per pixel row of a 32-channel tile
  static constexpr int HALO = TAPS == 9 ? 1 : 0;
  static constexpr int HR = TH + 2 * HALO, HC = 16 + 2 * HALO;
  static constexpr int NPOS = TH * 16, NHPOS = HR * HC;
  static constexpr int R = DG + 1;               // dy ring slots
  static constexpr int DY_SLOTS = NTN * NPOS * SPP;   // one dy frame
  static constexpr int X_SLOTS = NTC * NHPOS * SPP;   // one x halo tile
  static constexpr int DY_BYTES = DY_SLOTS * 16, X_BYTES = X_SLOTS * 16;
  static constexpr int LDS_BYTES = 2 * X_BYTES + R * DY_BYTES;
  static_assert(NTN * NTC * DG == 8, "one (n-tile, c-tile, dt) group per wave");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

__device__ __forceinline__ u32x2 tr_read(const char* p) {
  const short4v a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((SFVOS_LDS short4v*)p);
  return __builtin_bit_cast(u32x2, a);
}
__device__ __forceinline__ u32x4 join(const u32x2& lo, const u32x2& hi) {
  u32x4 v = {lo[0], lo[1], hi[0], hi[1]};
  return v;
}

template <int DT, int TAPS, int NTN, int NTC, int DG, int TH>
__global__ __launch_bounds__(512) void wgrad_kernel(WgradArgs a) {
  typedef WgradCfg<DT, TAPS, NTN, NTC, DG, TH> C;
  constexpr int CE = C::CE, ES = 16 / CE;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const xbase = smem;
  char* const dybase = smem + 2 * C::X_BYTES;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nt = wv % NTN, ct = (wv / NTN) % NTC, dg = wv / (NTN * NTC);
  const int r = lane & 31, hh = lane >> 5;

  int bid = blockIdx.x;
  const int ps = bid % a.psplit; bid /= a.psplit;
  const int nb = bid % a.n_blocks; bid /= a.n_blocks;
  const int cb = bid % a.c_blocks; bid /= a.c_blocks;
  const int db = bid;
  const int n_base = nb * NTN * 32, c_base = cb * NTC * 32, dt0 = db * DG;
  const int dt_live = min(DG, a.kt - dt0);      // temporal taps of this group that exist
  const int nfr = a.t_out + dt_live - 1;        // input frames per tile: t = dt0 .. dt0 + nfr - 1

  const int per = (a.ntiles + a.psplit - 1) / a.psplit;
  const int tile_begin = ps * per;
  const int tile_end = min(a.ntiles, tile_begin + per);
  const int S = max(0, tile_end - tile_begin) * nfr;

  f32x16 acc[TAPS];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

  const bool wave_live = (n_base + nt * 32 < a.c_out) && (c_base + ct * 32 < a.c_in) && (dg < dt_live);

  // stage s -> (tile, frame index fi); DMA of x[t = dt0 + fi] halo tile and dy frame fi
  auto issue = [&](int s) {
    const int tile = tile_begin + s / nfr, fi = s % nfr;
    int lvl = 0;
#pragma unroll
    for (int l = 1; l < SFVOS_MAX_LEVELS; ++l)
      if (l < a.lv.n && tile >= a.lv.tile_begin[l]) lvl = l;
    const int H = a.lv.H[lvl], W = a.lv.W[lvl];
    const long long HWp = (long long)H * W;
    int k = tile - a.lv.tile_begin[lvl];
    const int tw = k % a.lv.tiles_w[lvl]; k /= a.lv.tiles_w[lvl];
    const int th = k % a.lv.tiles_h[lvl]; k /= a.lv.tiles_h[lvl];
    const int b = k;
    const int h0 = th * TH, w0 = tw * 16;
    const int t = dt0 + fi;
    char* xb = xbase + (s & 1) * C::X_BYTES;
    char* dyb = dybase + (s % C::R) * C::DY_BYTES;
    const bool dy_ok = fi < a.t_out;
    const char* dyf = a.dy + (a.lv.ypos[lvl] + ((long long)b * a.t_out + fi) * HWp) * a.ld_y * ES;
#pragma unroll
    for (int it = 0; it < (C::DY_SLOTS + 511) / 512; ++it) {
      const int sl = it * 512 + tid;
      if (sl < C::DY_SLOTS) {
        const int j = sl % C::SPP, pos = (sl / C::SPP) % C::NPOS, tnt = sl / (C::SPP * C::NPOS);
        const int h = h0 + pos / 16, w = w0 + pos % 16, n = n_base + tnt * 32;
        const bool ok = dy_ok && h < H && w < W && n < a.c_out;
        const char* src = ok ? dyf + ((long long)(h * W + w) * a.ld_y + n + j * CE) * ES : a.zeros;
        glds16(src, dyb + (sl - lane) * 16);
      }
    }
    const bool x_ok = t < a.t_in;
    const char* xf = a.x + (a.lv.xpos[lvl] + ((long long)b * a.t_alloc + a.t_offset + t) * HWp) * a.ld_x * ES;
#pragma unroll
    for (int it = 0; it < (C::X_SLOTS + 511) / 512; ++it) {
      const int sl = it * 512 + tid;
      if (sl < C::X_SLOTS) {
        const int j = sl % C::SPP, hp = (sl / C::SPP) % C::NHPOS, tct = sl / (C::SPP * C::NHPOS);
        const int h = h0 + hp / C::HC - C::HALO, w = w0 + hp % C::HC - C::HALO, c = c_base + tct * 32;
        const bool ok = x_ok && (unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W && c < a.c_in;
        const char* src = ok ? xf + ((long long)(h * W + w) * a.ld_x + c + j * CE) * ES : a.zeros;
        glds16(src, xb + (sl - lane) * 16);
      }
    }
  };

  auto compute = [&](int s) {
    // this wave pairs x[t] with dy[t - dt0 - dg], which entered the ring dg stages ago
    const char* dyb = dybase + ((s - dg + C::R) % C::R) * C::DY_BYTES + nt * (C::NPOS * C::ROWB);
    const char* xb = xbase + (s & 1) * C::X_BYTES + ct * (C::NHPOS * C::ROWB);
    if constexpr (DT == SFVOS_BF16) {
      // lane -> (row q, 4-column group p) of its 16-lane group's 4x16 block; block rows k0..k0+3
      const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
      const int lane_off = (8 * (g >> 1) + q) * C::ROWB + (16 * (g & 1) + 4 * p) * 2;
      constexpr int TROWS = TAPS == 9 ? 3 : 1, TCOLS = TAPS == 9 ? 3 : 1;
      // step = (ty, dh): one A fragment per ty, TCOLS B fragments per step; the next step's
      // fragments are read while this step's MFMAs run (order pinned with sched_barrier)
      constexpr int PD = 3, NSTEP = TH * TROWS;
      u32x2 ar[PD][2], br[PD][TCOLS][2];
      auto load = [&](int step, int buf) {
        const int ty = step / TROWS, dh = step % TROWS;
        if (dh == 0) {
          const char* ap = dyb + ty * 16 * C::ROWB + lane_off;
          ar[ty % PD][0] = tr_read(ap);
          ar[ty % PD][1] = tr_read(ap + 4 * C::ROWB);
        }
#pragma unroll
        for (int dw = 0; dw < TCOLS; ++dw) {
          const char* bp = xb + ((ty + dh) * C::HC + dw) * C::ROWB + lane_off;
          br[buf][dw][0] = tr_read(bp);
          br[buf][dw][1] = tr_read(bp + 4 * C::ROWB);
        }
      };
#pragma unroll
      for (int step = 0; step < PD - 1 && step < NSTEP; ++step) load(step, step % PD);
#pragma unroll
      for (int step = 0; step < NSTEP; ++step) {
        if (step + PD - 1 < NSTEP) load(step + PD - 1, (step + PD - 1) % PD);
        __builtin_amdgcn_sched_barrier(0);
        const int ty = step / TROWS, dh = step % TROWS;
        const u32x4 av = join(ar[ty % PD][0], ar[ty % PD][1]);
#pragma unroll
        for (int dw = 0; dw < TCOLS; ++dw)
          Mma<SFVOS_BF16>::run(acc[dh * TCOLS + dw], av, join(br[step % PD][dw][0], br[step % PD][dw][1]));
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
      for (int ty = 0; ty < TH; ++ty) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
          const float av = *(const float*)(dyb + (ty * 16 + 2 * m + hh) * C::ROWB + r * 4);
#pragma unroll
          for (int tap = 0; tap < TAPS; ++tap) {
            const int dh = TAPS == 9 ? tap / 3 : 0, dw = TAPS == 9 ? tap % 3 : 0;
            const float bv = *(const float*)(xb + ((ty + dh) * C::HC + dw + 2 * m + hh) * C::ROWB + r * 4);
            acc[tap] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[tap], 0, 0, 0);
          }
        }
      }
    }
  };

  if (S > 0) issue(0);
  for (int s = 0; s < S; ++s) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (s + 1 < S) issue(s + 1);
    const int fo = s % nfr - dg;  // dy frame this wave pairs with x[t] at this stage
    if (wave_live && fo >= 0 && fo < a.t_out) compute(s);
  }

  // slab[ps][n][dt][tap][c]
  if (wave_live) {
    const int dt = dt0 + dg, c = c_base + ct * 32 + r;
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int n = n_base + nt * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
        a.slab[((((long long)ps * a.c_out + n) * a.kt + dt) * TAPS + tap) * a.c_in + c] = acc[tap][e];
      }
  }
}

// grad_w[n][c][dt][tap] (=|+=) sum_ps slab[ps][n][dt][tap][c]
// block = 64 consecutive c x 4 (n,dt,tap) rows: coalesced slab reads along c.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slab, int psplit, int c_out,
                                                           int c_in, int kt, int taps, float* grad_w, int accumulate) {
  const long long total = (long long)c_out * c_in * kt * taps;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    long long k = i;
    const int c = (int)(k % c_in); k /= c_in;
    const int tap = (int)(k % taps); k /= taps;
    const int dt = (int)(k % kt); k /= kt;
    const int n = (int)k;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int p = 0;
    for (; p + 4 <= psplit; p += 4) {
      s0 += slab[(long long)p * total + i];
      s1 += slab[(long long)(p + 1) * total + i];
      s2 += slab[(long long)(p + 2) * total + i];
      s3 += slab[(long long)(p + 3) * total + i];
    }
    for (; p < psplit; ++p) s0 += slab[(long long)p * total + i];
    const float s = (s0 + s1) + (s2 + s3);
    float* dst = grad_w + (((long long)n * c_in + c) * kt + dt) * taps + tap;
    *dst = accumulate ? *dst + s : s;
  }
}

struct WgradPlan {
  int cfg;  // 0: (1,2,4) 3x3 narrow-n ; 1: (2,2,2) 3x3 ; 2: (1,1,8) 3x3 c_in 32 ; 3: (2,1,4) 1x1
  int NTN, NTC, DG, TH;
  int n_blocks, c_blocks, dt_blocks, psplit, t_out, ntiles;
  WgradLevels lv;
};

static int make_wgrad_plan(const sfvos_conv_desc* d, WgradPlan* p) {
  SFVOS_REQUIRE(d != nullptr, "wgrad: null desc");
  SFVOS_REQUIRE(d->dtype == SFVOS_F32 || d->dtype == SFVOS_BF16, "wgrad: bad dtype");
  SFVOS_REQUIRE(d->taps == 9 || d->taps == 1, "wgrad: taps must be 9 or 1");
  SFVOS_REQUIRE(d->c_in % 32 == 0 && d->c_out % 32 == 0 && d->c_in > 0 && d->c_out > 0, "wgrad: channels % 32");
  SFVOS_REQUIRE(d->pad_t == 0, "wgrad: only forward convs (pad_t == 0) have a weight gradient here");
  const int ce = d->dtype == SFVOS_BF16 ? 8 : 4;
  SFVOS_REQUIRE(d->ld_x % ce == 0 && d->ld_y % ce == 0 && d->ld_x >= d->c_in && d->ld_y >= d->c_out, "wgrad: pitch");
  p->t_out = d->t_in - d->kt + 1;
  SFVOS_REQUIRE(p->t_out >= 1, "wgrad: kt > t_in");
  const bool f32 = d->dtype == SFVOS_F32;
  if (d->taps == 1) {
    p->cfg = 3; p->NTN = 2; p->NTC = 1; p->DG = 4;
  } else if (d->c_in <= 32) {
    p->cfg = 2; p->NTN = 1; p->NTC = 1; p->DG = 8;
  } else if (d->c_out <= 32) {
    p->cfg = 0; p->NTN = 1; p->NTC = 2; p->DG = 4;
  } else {
    p->cfg = 1; p->NTN = 2; p->NTC = 2; p->DG = 2;
  }
  p->TH = f32 ? 4 : 8;
  SFVOS_REQUIRE(d->pyr.n_levels >= 1 && d->pyr.n_levels <= SFVOS_MAX_LEVELS, "wgrad: n_levels out of range");
  SFVOS_REQUIRE(d->batch >= 1 && d->t_offset >= 0 && d->t_alloc >= d->t_offset + d->t_in, "wgrad: bad x window");
  WgradLevels& lv = p->lv;
  lv.n = d->pyr.n_levels;
  long long tiles = 0, px = 0;
  for (int l = 0; l < SFVOS_MAX_LEVELS; ++l) {
    const bool live = l < lv.n;
    const int H = live ? d->pyr.h[l] : 1, W = live ? d->pyr.w[l] : 1;
    SFVOS_REQUIRE(H >= 1 && W >= 1, "wgrad: level %d has bad extent", l);
    lv.H[l] = H; lv.W[l] = W;
    lv.tiles_h[l] = ceil_div(H, p->TH); lv.tiles_w[l] = ceil_div(W, 16);
    lv.tile_begin[l] = (int)tiles;
    lv.xpos[l] = (long long)d->batch * d->t_alloc * px;
    lv.ypos[l] = (long long)d->batch * p->t_out * px;
    if (live) {
      tiles += (long long)d->batch * lv.tiles_h[l] * lv.tiles_w[l];
      px += (long long)H * W;
    }
    SFVOS_REQUIRE(tiles < (1ll << 30), "wgrad: too many tiles");
  }
  lv.tile_begin[SFVOS_MAX_LEVELS] = (int)tiles;
  p->n_blocks = ceil_div(d->c_out, 32 * p->NTN);
  p->c_blocks = ceil_div(d->c_in, 32 * p->NTC);
  p->dt_blocks = ceil_div(d->kt, p->DG);
  p->ntiles = (int)tiles;
  const int col_blocks = p->n_blocks * p->c_blocks * p->dt_blocks;
  int ps = ceil_div(768, col_blocks);  // ~3 workgroups per CU over the launch
  if (ps > p->ntiles) ps = p->ntiles;
  if (ps < 1) ps = 1;
  const int per = ceil_div(p->ntiles, ps);  // no empty splits
  p->psplit = ceil_div(p->ntiles, per);
  return SFVOS_OK;
}

template <int DT, int TAPS, int NTN, int NTC, int DG, int TH>
static int launch_wgrad(const WgradArgs& a, long long grid, hipStream_t stream) {
  typedef WgradCfg<DT, TAPS, NTN, NTC, DG, TH> C;
  auto kern = wgrad_kernel<DT, TAPS, NTN, NTC, DG, TH>;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
    if (e != hipSuccess) {
      set_error("wgrad: hipFuncSetAttribute(%d B LDS) failed: %s", C::LDS_BYTES, hipGetErrorString(e));
      return SFVOS_E_LAUNCH;
    }
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), C::LDS_BYTES, stream, a);
  return check_launch("wgrad");
}

}  // namespace sfvos

using namespace sfvos;

extern "C" size_t sfvos_conv3d_wgrad_workspace_bytes(const sfvos_conv_desc* d) {
  WgradPlan p;
  if (make_wgrad_plan(d, &p) != SFVOS_OK) return 0;
  return (size_t)p.psplit * d->c_out * d->c_in * d->kt * d->taps * sizeof(float);
}

extern "C" int sfvos_conv3d_wgrad(const sfvos_conv_desc* d, const void* x, const void* dy, float* grad_w,
                                  int accumulate, void* workspace, const void* zeros, sfvos_stream_t stream) {
  WgradPlan p;
  int rc = make_wgrad_plan(d, &p);
  if (rc != SFVOS_OK) return rc;
  SFVOS_REQUIRE(x && dy && grad_w && workspace && zeros, "wgrad: null pointer");
  WgradArgs a;
  a.x = (const char*)x; a.dy = (const char*)dy; a.slab = (float*)workspace; a.zeros = (const char*)zeros;
  a.t_in = d->t_in; a.t_alloc = d->t_alloc; a.t_offset = d->t_offset; a.t_out = p.t_out; a.c_in = d->c_in;
  a.c_out = d->c_out; a.kt = d->kt;
  a.ld_x = d->ld_x; a.ld_y = d->ld_y; a.batch = d->batch;
  a.n_blocks = p.n_blocks; a.c_blocks = p.c_blocks;
  a.dt_blocks = p.dt_blocks; a.psplit = p.psplit;
  a.ntiles = p.ntiles; a.lv = p.lv;
  const long long grid = (long long)p.psplit * p.n_blocks * p.c_blocks * p.dt_blocks;
  hipStream_t s = (hipStream_t)stream;
  const bool bf = d->dtype == SFVOS_BF16;
  switch (p.cfg) {
    case 0: rc = bf ? launch_wgrad<SFVOS_BF16, 9, 1, 2, 4, 8>(a, grid, s) : launch_wgrad<SFVOS_F32, 9, 1, 2, 4, 4>(a, grid, s); break;
    case 1: rc = bf ? launch_wgrad<SFVOS_BF16, 9, 2, 2, 2, 8>(a, grid, s) : launch_wgrad<SFVOS_F32, 9, 2, 2, 2, 4>(a, grid, s); break;
    case 2: rc = bf ? launch_wgrad<SFVOS_BF16, 9, 1, 1, 8, 8>(a, grid, s) : launch_wgrad<SFVOS_F32, 9, 1, 1, 8, 4>(a, grid, s); break;
    default: rc = bf ? launch_wgrad<SFVOS_BF16, 1, 2, 1, 4, 8>(a, grid, s) : launch_wgrad<SFVOS_F32, 1, 2, 1, 4, 4>(a, grid, s); break;
  }
  if (rc != SFVOS_OK) return rc;
  const long long total = (long long)d->c_out * d->c_in * d->kt * d->taps;
  long long rgrid = ceil_div64(total, 256);
  if (rgrid > 8192) rgrid = 8192;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)rgrid), dim3(256), 0, s, (const float*)workspace, p.psplit,
                     d->c_out, d->c_in, d->kt, d->taps, grad_w, accumulate);
  return check_launch("wgrad_reduce");
}
