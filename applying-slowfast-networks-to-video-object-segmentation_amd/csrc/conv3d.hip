// conv3d.hip -- kt x 3 x 3 (pad 0,1,1) and k x 1 x 1 3D convolution, pyramid NDHWC, as an implicit
// GEMM on the gfx950 matrix cores.  Replaces aten::convolution at reference
// code/helpers/model.py:112,120,124,132,136,144,147; with a flipped weight image and
// pad_t = kt-1 the same kernel is aten::convolution_backward's grad_input.
// ONE launch covers every FPN level of temporally_enhance_features (model.py:156-163).
//
// Decomposition (one workgroup = 8 waves, 2 per SIMD):
//   output tile  = TT output frames x (TH rows x 32 px) x BN output channels, f32 accumulators
//   K loop       = for 64-byte channel chunk cc / for temporal tap dt / for tap group tg
//   temporal ring: the halo tiles ((TH+2) x 34 px x 64 B) of TT+1 consecutive input frames sit in
//                  LDS.  At temporal tap dt output frame j reads ring frame dt+j, so one weight
//                  slice W[dt][taps][cc] feeds all TT frames (weights are streamed ONCE per
//                  workgroup) and every ring frame is re-used by 9 spatial taps x up to TT
//                  temporal taps.  While tap dt computes, frame dt+TT is DMA'd into the free slot.
//   staging      : global_load_lds (LDS-DMA) 16 B per lane; weights double-buffered; one barrier
//                  per stage.  Out-of-image / out-of-clip pixels come from a zero page.
//   LDS images   : chunk-major [16B chunk][row][col] and [tap][chunk][n]: every ds_read_b128 of an
//                  MFMA operand covers 32 consecutive 16-B slots per half-wave -> conflict-free.
//   inner loop   : branch-free: per (tap, k-step) NT B-fragment + TT*MT A-fragment reads feed
//                  TT*MT*NT MFMAs (32x32x16 bf16, or 4 x 32x32x2 exact f32); the reads of step k+1
//                  are pinned ahead of the MFMAs of step k.
//   epilogue     : + bias, optional += y, store as dtype, per-channel (sum, sumsq) of the tile
//                  written as one deterministic partial row per workgroup (BN statistics).
#include <stdlib.h>

#include "common.h"

namespace sfvos {

struct ConvLevels {
  int n;
  int H[SFVOS_MAX_LEVELS], W[SFVOS_MAX_LEVELS], tiles_h[SFVOS_MAX_LEVELS], tiles_w[SFVOS_MAX_LEVELS];
  int wg_begin[SFVOS_MAX_LEVELS + 1];   // first workgroup of each level
  int row_begin[SFVOS_MAX_LEVELS + 1];  // first statistics row of each level
  long long xpos[SFVOS_MAX_LEVELS];     // first position of the level in the x / y pyramid buffers
  long long ypos[SFVOS_MAX_LEVELS];
};

struct ConvArgs {
  const char* x;
  const char* wp;
  const float* bias;
  char* y;
  float* stat_part;
  const char* zeros;
  int batch, t_in, t_alloc, t_offset, t_out, c_in, c_out, kt, pad_t, ld_x, ld_y, accumulate;
  int t_blocks, n_blocks;
  int debug;  // timing-only: bit0 skip compute, bit1 skip DMA after the first stage (results wrong)
  ConvLevels lv;
};

template <int DT, int TAPS, int TPS, int TT, int MT, int NT, int WS, int WN>
struct ConvCfg {
  static constexpr int NWAVES = WS * WN, NTHREADS = 64 * NWAVES;
  static constexpr int TH = WS * MT, TW = 32;
  static constexpr int HALO = (TAPS == 9) ? 1 : 0;
  static constexpr int HR = TH + 2 * HALO, HC = TW + 2 * HALO;
  static constexpr int BN = WN * NT * 32;
  static constexpr int R = TT + 1;                    // ring slots
  static constexpr int X_SLOTS = 4 * HR * HC;         // 16-B slots per ring frame
  static constexpr int X_BYTES = ((X_SLOTS * 16 + 255) / 256) * 256;
  static constexpr int W_SLOTS = TPS * 4 * BN;
  static constexpr int W_BYTES = W_SLOTS * 16;
  static constexpr int NTG = TAPS / TPS;
  static constexpr int LDS_BYTES = R * X_BYTES + 2 * W_BYTES;
  static_assert(TAPS % TPS == 0, "tap groups must tile the taps");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

template <int DT, int TAPS, int TPS, int TT, int MT, int NT, int WS, int WN>
__global__ __launch_bounds__(64 * WS * WN) void conv3d_kernel(ConvArgs a) {
  typedef ConvCfg<DT, TAPS, TPS, TT, MT, NT, WS, WN> C;
  typedef typename Elt<DT>::type T;
  constexpr int CE = Elt<DT>::CE, CK = 4 * CE, ES = 16 / CE;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const ring = smem;
  char* const wbase = smem + C::R * C::X_BYTES;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ws = wv % WS, wn = wv / WS;
  const int r = lane & 31, hh = lane >> 5;

  // workgroup -> (level, clip, frame block, channel block, pixel tile)
  int lvl = 0;
#pragma unroll
  for (int l = 1; l < SFVOS_MAX_LEVELS; ++l)
    if (l < a.lv.n && (int)blockIdx.x >= a.lv.wg_begin[l]) lvl = l;
  const int H = a.lv.H[lvl], W = a.lv.W[lvl], tiles_w = a.lv.tiles_w[lvl], tiles_h = a.lv.tiles_h[lvl];
  int bid = blockIdx.x - a.lv.wg_begin[lvl];
  const int tw = bid % tiles_w; bid /= tiles_w;
  const int th = bid % tiles_h; bid /= tiles_h;
  const int nb = bid % a.n_blocks; bid /= a.n_blocks;
  const int tb = bid % a.t_blocks; bid /= a.t_blocks;
  const int b = bid;
  const int h0 = th * C::TH, w0 = tw * C::TW, n0 = nb * C::BN, tb0 = tb * TT;

  const int NF = TT + a.kt - 1;  // input frames this workgroup touches: t = tb0 - pad_t + i
  const int ncc = a.c_in / CK;
  const int S = ncc * a.kt * C::NTG;
  const long long HWp = (long long)H * W;
  // frame 0 of this clip inside the x buffer (t_alloc frames per clip, the conv's window starts at t_offset)
  const char* xclip = a.x + (a.lv.xpos[lvl] + ((long long)b * a.t_alloc + a.t_offset) * HWp) * a.ld_x * ES;

  f32x16 acc[TT][MT][NT];
#pragma unroll
  for (int j = 0; j < TT; ++j)
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int q = 0; q < NT; ++q)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][i][q][e] = 0.f;

  // ---- DMA: one ring frame (halo tile of input frame i, channel chunk cc) ----------------------
  auto issue_frame = [&](int cc, int i) {
    const int t = tb0 - a.pad_t + i;
    const bool t_ok = (unsigned)t < (unsigned)a.t_in;
    char* xb = ring + ((cc * NF + i) % C::R) * C::X_BYTES;
    const char* xsrc = xclip + ((long long)t * HWp * a.ld_x + cc * CK) * ES;
#pragma unroll
    for (int it = 0; it < (C::X_SLOTS + C::NTHREADS - 1) / C::NTHREADS; ++it) {
      const int sl = it * C::NTHREADS + tid;
      if (sl < C::X_SLOTS) {
        const int col = sl % C::HC, rowj = sl / C::HC, row = rowj % C::HR, j = rowj / C::HR;
        const int h = h0 + row - C::HALO, w = w0 + col - C::HALO;
        const bool ok = t_ok && (unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W;
        const char* src = ok ? xsrc + ((long long)(h * W + w) * a.ld_x + j * CE) * ES : a.zeros;
        glds16(src, xb + (sl - lane) * 16);
      }
    }
  };
  // ---- DMA: weight slice of stage s = (cc, dt, tg): [TPS taps][4 chunks][BN] -------------------
  auto issue_w = [&](int cc, int dt, int tg, int s) {
    char* wb = wbase + (s & 1) * C::W_BYTES;
    const char* wsrc = a.wp + (((long long)(cc * a.kt + dt) * TAPS + tg * TPS) * 4 * a.c_out + n0) * 16;
#pragma unroll
    for (int it = 0; it < (C::W_SLOTS + C::NTHREADS - 1) / C::NTHREADS; ++it) {
      const int sl = it * C::NTHREADS + tid;
      if (sl < C::W_SLOTS) {
        const int tj = sl / C::BN, n = sl - tj * C::BN;  // tj = tap_local*4 + chunk
        if (n0 + n < a.c_out) glds16(wsrc + ((long long)tj * a.c_out + n) * 16, wb + (sl - lane) * 16);
      }
    }
  };

  // ---- one stage of MFMAs --------------------------------------------------------------------------
  auto compute = [&](int cc, int dt, int tg, int s) {
    const char* wb = wbase + (s & 1) * C::W_BYTES;
    const char* xf[TT];
#pragma unroll
    for (int j = 0; j < TT; ++j) xf[j] = ring + ((cc * NF + dt + j) % C::R) * C::X_BYTES;
    // k-step k = (tap_local, st): operand fragments of step k+1 are read from LDS while the MFMAs
    // of step k run (software pipeline in registers; the waits become counted, not lgkmcnt(0)).
    u32x4 bv[2][NT], av[2][TT][MT];
    auto load = [&](int k, int buf) {
#ifdef SFVOS_ABLATE  // timing-only builds (scratch): 1 = no A re-reads, 2 = no B re-reads, 3 = neither
      const bool skip_a = (SFVOS_ABLATE & 1) && k > 1, skip_b = (SFVOS_ABLATE & 2) && k > 1;
#else
      constexpr bool skip_a = false, skip_b = false;
#endif
      const int tp = k >> 1, st = k & 1;
      const int tap = tg * TPS + tp;
      const int dh = (TAPS == 9) ? tap / 3 : 0, dw = (TAPS == 9) ? tap - 3 * dh : 0;
#pragma unroll
      for (int q = 0; q < NT; ++q)
        if (!skip_b) bv[buf][q] = lds_read16(wb + (((tp * 4 + 2 * st + hh) * C::BN) + (wn * NT + q) * 32 + r) * 16);
#pragma unroll
      for (int j = 0; j < TT; ++j)
#pragma unroll
        for (int i = 0; i < MT; ++i)
          if (!skip_a)
            av[buf][j][i] = lds_read16(xf[j] + (((2 * st + hh) * C::HR + ws * MT + i + dh) * C::HC + r + dw) * 16);
    };
    load(0, 0);
#pragma unroll
    for (int k = 0; k < 2 * TPS; ++k) {
      if (k + 1 < 2 * TPS) load(k + 1, (k + 1) & 1);
      // pin the order: hipcc otherwise sinks each ds_read next to its MFMA and drains lgkmcnt(0)
      // before every MFMA (LDS-latency-bound stream)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < TT; ++j)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int q = 0; q < NT; ++q) Mma<DT>::run(acc[j][i][q], av[k & 1][j][i], bv[k & 1][q]);
    }
  };

  // ---- main loop -------------------------------------------------------------------------------------
  int s = 0;
  for (int cc = 0; cc < ncc; ++cc) {
    // chunk prologue: refill the ring with frames 0..TT-1 of this chunk.  All waves must have
    // finished the previous chunk's last stage before its live slots are overwritten.
    if (cc > 0) __syncthreads();
    for (int i = 0; i < TT; ++i) issue_frame(cc, i);
    if (cc == 0) issue_w(0, 0, 0, 0);
    for (int dt = 0; dt < a.kt; ++dt) {
      for (int tg = 0; tg < C::NTG; ++tg, ++s) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // stage s operands landed; everyone is done with stage s-1
        if (!(a.debug & 2)) {
          if (tg == 0 && dt + TT < NF) issue_frame(cc, dt + TT);  // slot freed by frame dt-1
          if (s + 1 < S) {
            int ntg = tg + 1, ndt = dt, ncc2 = cc;
            if (ntg == C::NTG) { ntg = 0; if (++ndt == a.kt) { ndt = 0; ++ncc2; } }
            issue_w(ncc2, ndt, ntg, s + 1);
          }
        }
        if (!(a.debug & 1)) compute(cc, dt, tg, s);
      }
    }
  }

  // ---- epilogue --------------------------------------------------------------------------------------
  float s1[NT], s2[NT];
#pragma unroll
  for (int q = 0; q < NT; ++q) s1[q] = s2[q] = 0.f;
  T* yclip = (T*)a.y + (a.lv.ypos[lvl] + (long long)b * a.t_out * HWp) * a.ld_y;
#pragma unroll
  for (int q = 0; q < NT; ++q) {
    const int n = n0 + (wn * NT + q) * 32 + r;
    const bool nok = n < a.c_out;
    const float bias = (a.bias && nok) ? a.bias[n] : 0.f;
#pragma unroll
    for (int j = 0; j < TT; ++j) {
      const int to = tb0 + j;
      if (to < a.t_out) {
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          const int h = h0 + ws * MT + i;
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int w = w0 + (e & 3) + 8 * (e >> 2) + 4 * hh;
            if (nok && h < H && w < W) {
              T* dst = yclip + ((long long)(to * H + h) * W + w) * a.ld_y + n;
              float v = acc[j][i][q][e] + bias;
              if (a.accumulate) v += Elt<DT>::to_f32(*dst);
              *dst = Elt<DT>::from_f32(v);
              s1[q] += v;
              s2[q] += v * v;
            }
          }
        }
      }
    }
  }
  if (a.stat_part) {
    __syncthreads();  // staging buffers are dead: reuse as reduction scratch
    float* red = (float*)smem;  // [NWAVES][NT][32][2]
#pragma unroll
    for (int q = 0; q < NT; ++q) {
      const float t1 = s1[q] + __shfl_xor(s1[q], 32), t2 = s2[q] + __shfl_xor(s2[q], 32);
      if (lane < 32) {
        red[((wv * NT + q) * 32 + lane) * 2 + 0] = t1;
        red[((wv * NT + q) * 32 + lane) * 2 + 1] = t2;
      }
    }
    __syncthreads();
    if (tid < C::BN && n0 + tid < a.c_out) {
      const int wn_ = tid / (NT * 32), q = (tid / 32) % NT, l = tid & 31;
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int k = 0; k < WS; ++k) {
        t1 += red[(((wn_ * WS + k) * NT + q) * 32 + l) * 2 + 0];
        t2 += red[(((wn_ * WS + k) * NT + q) * 32 + l) * 2 + 1];
      }
      const long long prow = a.lv.row_begin[lvl] + ((long long)(b * a.t_blocks + tb) * tiles_h + th) * tiles_w + tw;
      a.stat_part[(prow * 2 + 0) * a.c_out + n0 + tid] = t1;
      a.stat_part[(prow * 2 + 1) * a.c_out + n0 + tid] = t2;
    }
  }
}

// ---- host-side planning ----------------------------------------------------------------------------
struct ConvPlan {
  int family;  // 0 narrow (c_out <= 32), 1 mid (c_out == 64, 1x1), 2 wide
  int TT, NT, TH, BN;
  int t_blocks, n_blocks, t_out;
  ConvLevels lv;
};

static int pick_tt(int t_out, int max_tt) {
  // fewest frame blocks first, then the smallest TT that covers them
  const int nblk = ceil_div(t_out, max_tt);
  return ceil_div(t_out, nblk);
}

static int make_plan(const sfvos_conv_desc* d, ConvPlan* p) {
  SFVOS_REQUIRE(d != nullptr, "conv: null desc");
  SFVOS_REQUIRE(d->dtype == SFVOS_F32 || d->dtype == SFVOS_BF16, "conv: bad dtype %d", d->dtype);
  SFVOS_REQUIRE(d->taps == 9 || d->taps == 1, "conv: taps must be 9 or 1, got %d", d->taps);
  SFVOS_REQUIRE(d->c_in > 0 && d->c_in % 32 == 0 && d->c_out > 0 && d->c_out % 32 == 0,
                "conv: channels must be positive multiples of 32 (c_in %d, c_out %d)", d->c_in, d->c_out);
  SFVOS_REQUIRE(d->c_out <= 256, "conv: c_out %d > 256 unsupported", d->c_out);
  SFVOS_REQUIRE(d->batch >= 1 && d->t_in >= 1 && d->kt >= 1, "conv: bad extent");
  SFVOS_REQUIRE(d->t_offset >= 0 && d->t_alloc >= d->t_offset + d->t_in,
                "conv: x window [t_offset %d, +t_in %d) exceeds t_alloc %d", d->t_offset, d->t_in, d->t_alloc);
  SFVOS_REQUIRE(d->pad_t >= 0 && d->pad_t < d->kt + 1, "conv: bad pad_t %d", d->pad_t);
  SFVOS_REQUIRE(d->ld_x >= d->c_in && d->ld_y >= d->c_out, "conv: pitch smaller than channel count");
  const int ce = d->dtype == SFVOS_BF16 ? 8 : 4;
  SFVOS_REQUIRE(d->ld_x % ce == 0, "conv: ld_x %d must be a multiple of %d (16-byte chunks)", d->ld_x, ce);
  SFVOS_REQUIRE(d->pyr.n_levels >= 1 && d->pyr.n_levels <= SFVOS_MAX_LEVELS, "conv: n_levels %d out of [1,%d]",
                d->pyr.n_levels, SFVOS_MAX_LEVELS);
  p->t_out = d->t_in + 2 * d->pad_t - d->kt + 1;
  SFVOS_REQUIRE(p->t_out >= 1, "conv: kernel longer than padded input (t_in %d, kt %d, pad_t %d)", d->t_in, d->kt,
                d->pad_t);
  if (d->c_out <= 32) {
    p->family = 0; p->TT = pick_tt(p->t_out, 4); p->NT = 1; p->TH = 8; p->BN = 32;
  } else if (d->c_out == 64 && d->taps == 1) {
    p->family = 1; p->TT = pick_tt(p->t_out, 3); p->NT = 1; p->TH = 8; p->BN = 64;
  } else {
    // 12 accumulator tiles (NT 4 x TT 3) would spill: cap TT at 2 for the 256-wide image
    p->family = 2; p->NT = d->c_out <= 192 ? 3 : 4; p->TT = pick_tt(p->t_out, p->NT == 4 ? 2 : 3); p->TH = 4;
    p->BN = 64 * p->NT;
  }
  p->t_blocks = ceil_div(p->t_out, p->TT);
  p->n_blocks = ceil_div(d->c_out, p->BN);
  ConvLevels& lv = p->lv;
  lv.n = d->pyr.n_levels;
  long long wg = 0, rows = 0, px = 0;
  for (int l = 0; l < SFVOS_MAX_LEVELS; ++l) {
    const bool live = l < lv.n;
    const int H = live ? d->pyr.h[l] : 1, W = live ? d->pyr.w[l] : 1;
    SFVOS_REQUIRE(H >= 1 && W >= 1, "conv: level %d has bad extent %dx%d", l, H, W);
    lv.H[l] = H; lv.W[l] = W;
    lv.tiles_h[l] = ceil_div(H, p->TH); lv.tiles_w[l] = ceil_div(W, 32);
    lv.wg_begin[l] = (int)wg; lv.row_begin[l] = (int)rows;
    lv.xpos[l] = (long long)d->batch * d->t_alloc * px;
    lv.ypos[l] = (long long)d->batch * p->t_out * px;
    if (live) {
      wg += (long long)d->batch * p->t_blocks * p->n_blocks * lv.tiles_h[l] * lv.tiles_w[l];
      rows += (long long)d->batch * p->t_blocks * lv.tiles_h[l] * lv.tiles_w[l];
      px += (long long)H * W;
    }
    SFVOS_REQUIRE(wg < (1ll << 31) && rows < (1ll << 31), "conv: grid out of range");
  }
  lv.wg_begin[SFVOS_MAX_LEVELS] = (int)wg;
  lv.row_begin[SFVOS_MAX_LEVELS] = (int)rows;
  return SFVOS_OK;
}

template <int DT, int TAPS, int TPS, int TT, int MT, int NT, int WS, int WN>
static int launch(const ConvArgs& a, long long grid, hipStream_t stream) {
  typedef ConvCfg<DT, TAPS, TPS, TT, MT, NT, WS, WN> C;
  auto kern = conv3d_kernel<DT, TAPS, TPS, TT, MT, NT, WS, WN>;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
    if (e != hipSuccess) {
      set_error("conv: hipFuncSetAttribute(%d B LDS) failed: %s", C::LDS_BYTES, hipGetErrorString(e));
      return SFVOS_E_LAUNCH;
    }
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(C::NTHREADS), C::LDS_BYTES, stream, a);
  return check_launch("conv3d");
}

template <int DT, int TAPS>
static int dispatch(const ConvPlan& p, const ConvArgs& a, long long grid, hipStream_t s) {
#define SFVOS_CASE(F, TPSv, TTv, MTv, NTv, WSv, WNv) \
  if (p.family == F && p.TT == TTv && p.NT == NTv) return launch<DT, TAPS, TPSv, TTv, MTv, NTv, WSv, WNv>(a, grid, s);
  // narrow: 8 rows x 32 px x TT frames x 32 channels; wave = one row, all frames
  constexpr int NTPS = TAPS == 9 ? 9 : 1;
  SFVOS_CASE(0, NTPS, 1, 1, 1, 8, 1) SFVOS_CASE(0, NTPS, 2, 1, 1, 8, 1) SFVOS_CASE(0, NTPS, 3, 1, 1, 8, 1)
  SFVOS_CASE(0, NTPS, 4, 1, 1, 8, 1)
  if constexpr (TAPS == 1) {
    // mid (lateral 32->64): 8 rows x 32 px x TT frames x 64 channels
    SFVOS_CASE(1, 1, 1, 2, 1, 4, 2) SFVOS_CASE(1, 1, 2, 2, 1, 4, 2) SFVOS_CASE(1, 1, 3, 2, 1, 4, 2)
  } else {
    // wide: 4 rows x 32 px x TT frames x 192/256 channels, 3 taps per stage
    SFVOS_CASE(2, 3, 1, 1, 3, 4, 2) SFVOS_CASE(2, 3, 2, 1, 3, 4, 2) SFVOS_CASE(2, 3, 3, 1, 3, 4, 2)
    SFVOS_CASE(2, 3, 1, 1, 4, 4, 2) SFVOS_CASE(2, 3, 2, 1, 4, 4, 2)
  }
#undef SFVOS_CASE
  set_error("conv: no kernel instance for family %d TT %d NT %d taps %d", p.family, p.TT, p.NT, TAPS);
  return SFVOS_E_ARG;
}

}  // namespace sfvos

using namespace sfvos;

extern "C" int sfvos_conv3d_stat_rows(const sfvos_conv_desc* d, int* rows_per_level) {
  ConvPlan p;
  if (make_plan(d, &p) != SFVOS_OK) return SFVOS_E_ARG;
  if (rows_per_level)
    for (int l = 0; l < p.lv.n; ++l) rows_per_level[l] = p.lv.row_begin[l + 1] - p.lv.row_begin[l];
  return p.lv.row_begin[SFVOS_MAX_LEVELS];
}

extern "C" int sfvos_conv3d(const sfvos_conv_desc* d, const void* x, const void* w_packed, const float* bias, void* y,
                            float* stat_part, const void* zeros, sfvos_stream_t stream) {
  ConvPlan p;
  int rc = make_plan(d, &p);
  if (rc != SFVOS_OK) return rc;
  SFVOS_REQUIRE(x && w_packed && y && zeros, "conv: null pointer");
  SFVOS_REQUIRE(!(d->taps == 1 && p.family == 2), "conv: 1x1 conv with c_out > 64 has no kernel instance");
  ConvArgs a;
  a.x = (const char*)x; a.wp = (const char*)w_packed; a.bias = bias; a.y = (char*)y; a.stat_part = stat_part;
  a.zeros = (const char*)zeros;
  a.batch = d->batch; a.t_in = d->t_in; a.t_alloc = d->t_alloc; a.t_offset = d->t_offset; a.t_out = p.t_out;
  a.c_in = d->c_in; a.c_out = d->c_out; a.kt = d->kt;
  a.pad_t = d->pad_t; a.ld_x = d->ld_x; a.ld_y = d->ld_y; a.accumulate = d->accumulate;
  a.t_blocks = p.t_blocks; a.n_blocks = p.n_blocks;
  a.lv = p.lv;
  { const char* dbg = getenv("SFVOS_CONV_DEBUG"); a.debug = dbg ? atoi(dbg) : 0; }
  const long long grid = p.lv.wg_begin[SFVOS_MAX_LEVELS];
  SFVOS_REQUIRE(grid > 0, "conv: empty grid");
  hipStream_t s = (hipStream_t)stream;
  if (d->dtype == SFVOS_BF16)
    return d->taps == 9 ? dispatch<SFVOS_BF16, 9>(p, a, grid, s) : dispatch<SFVOS_BF16, 1>(p, a, grid, s);
  return d->taps == 9 ? dispatch<SFVOS_F32, 9>(p, a, grid, s) : dispatch<SFVOS_F32, 1>(p, a, grid, s);
}
