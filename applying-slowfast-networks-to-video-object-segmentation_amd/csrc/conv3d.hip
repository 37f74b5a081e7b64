// conv3d.hip -- kt x 3 x 3 (pad 0,1,1) and k x 1 x 1 3D convolution, pyramid NDHWC, as an implicit
// GEMM on the gfx950 matrix cores.  Replaces aten::convolution at reference
// code/helpers/model.py:112,120,124,132,136,144,147; with a flipped weight image and
// pad_t = kt-1 the same kernel is aten::convolution_backward's grad_input.
// ONE launch covers every FPN level of temporally_enhance_features (model.py:156-163).
//
// Decomposition (one workgroup = 8 waves, 2 per SIMD):
//   output tile  = TT output frames x (TH rows x 32 px) x BN output channels, f32 accumulators
//   K loop       = for 64-byte channel chunk cc / for temporal tap dt / for tap group tg
//   temporal ring: the halo tiles ((TH+2) x 34 px x 64 B) of consecutive input frames sit in LDS
//                  (2 TT slots in the wide kernel, TT+1 in the frame-split kernel).  At temporal
//                  tap dt output frame j reads ring frame dt+j, so one weight slice W[dt][taps][cc]
//                  feeds all TT frames (weights are streamed ONCE per workgroup) and every ring
//                  frame is re-used by 9 spatial taps x up to TT temporal taps.  While tap dt
//                  computes, frame dt+TT is DMA'd into a free slot; at the last tap of a channel
//                  chunk the wide kernel copies the next chunk's first TT frames instead.
//   staging      : buffer_load ... lds (LDS-DMA) 16 B per lane; weights double-buffered; one barrier
//                  per stage.  Out-of-image / out-of-clip pixels are zero-filled by the buffer
//                  descriptor's range check (never read).
//   LDS images   : chunk-major [16B chunk][row][col] and [tap][chunk][n]: every ds_read_b128 of an
//                  MFMA operand covers 32 consecutive 16-B slots per half-wave -> conflict-free.
//   inner loop   : branch-free: per (tap, k-step) NT B-fragment + TT*MT A-fragment reads feed
//                  TT*MT*NT MFMAs (4 x 32x32x2 exact f32, 32x32x64 e4m3); the reads of step k+1
//                  are pinned ahead of the MFMAs of step k.  bf16: 16x16x32 MFMAs, a tap = four
//                  quadrant steps (pixel half, channel half) sharing operand sets -- see M16 below.
//   epilogue     : + bias, optional += y, store as dtype, per-channel (sum, sumsq) of the tile
//                  written as one deterministic partial row per workgroup (BN statistics).
#include <stdlib.h>

#include "common.h"

// Diagnostic stamps (scratch builds with -DSFVOS_STAMP only; never in the shipped library).
#ifdef SFVOS_STAMP
#define SFVOS_STAMP_AT(idx)                                                                   \
  if (a.stamps && blockIdx.x == 300 && s < 128) {                                             \
    const unsigned long long tstamp = __builtin_amdgcn_s_memtime();                           \
    if (lane == 0) a.stamps[(wv * 128 + s) * 4 + (idx)] = tstamp;                             \
  }
#else
#define SFVOS_STAMP_AT(idx)
#endif

// Timing-only switches (SFVOS_CONV_DEBUG bits; results are WRONG with them) exist only in diagnostic builds
// (-DSFVOS_DIAG): in the shipped library every test folds to a constant, so no branch, no extra live range.
#ifdef SFVOS_DIAG
#define SFVOS_DBG(bit) ((a.debug & (bit)) != 0)
#else
#define SFVOS_DBG(bit) false
#endif

namespace sfvos {

struct ConvLevels {
  int n;
  int H[SFVOS_MAX_LEVELS], W[SFVOS_MAX_LEVELS], tiles_h[SFVOS_MAX_LEVELS], tiles_w[SFVOS_MAX_LEVELS];
  int wg_begin[SFVOS_MAX_LEVELS + 1];   // first workgroup of each level
  int row_begin[SFVOS_MAX_LEVELS + 1];  // first statistics row of each level
  long long xpos[SFVOS_MAX_LEVELS];     // first position of the level in the x / y pyramid buffers
  long long xbs[SFVOS_MAX_LEVELS];      // x: positions between consecutive clips / consecutive frames of the level
  long long xfs[SFVOS_MAX_LEVELS];      //    (level-major: t_alloc*HW, HW; frame-major ring: HW, x_frame_stride)
  long long ypos[SFVOS_MAX_LEVELS];
  long long ybs[SFVOS_MAX_LEVELS];      // y: positions between consecutive clips / frames (as xbs / xfs)
  long long yfs[SFVOS_MAX_LEVELS];
};

struct ConvArgs {
  const char* x;
  const char* wp;
  const float* bias;
  char* y;
  float* stat_part;
  int batch, t_in, t_alloc, t_offset, t_out, c_in, c_out, kt, pad_t, ld_x, ld_y, accumulate, relu;
  // x addressing: element (position, chunk cc of 64 bytes) at x + cc*x_chunk_bytes + position*x_pitch_bytes:
  // pyramid NDHWC = (64, ld_x*ES); channel-group-major input = (group stride in bytes, 64)
  long long x_chunk_bytes;
  int x_pitch_bytes;
  int t_blocks, n_blocks;  // frame blocks of THIS launch
  int t_first, t_end;      // output frames [t_first, t_end) of this launch (a layer may take two launches:
  int tb_offset, t_blocks_total;  // blocks of TT and of TT-1 frames so that no frame slot is padding)
  int debug;  // timing-only: bit0 skip compute, bit1 skip DMA after the first stage (results wrong)
  // frame-split kernel only: ONE launch holds the blocks of 4, 2 and 1 frames (part 0, 1, 2), big blocks first,
  // so the short workgroups fill the tail of the long ones; lv.wg_begin then counts per frame block.
  struct Part { int wgs, t_blocks, t_first, t_end, tb_offset; } part[3];
#ifdef SFVOS_STAMP
  unsigned long long* stamps;  // diagnostic build only: s_memtime stamps of one workgroup
#endif
  ConvLevels lv;
};

template <int DT, int TAPS, int TPS, int TT, int MT, int NT, int WS, int WN>
struct ConvCfg {
  static constexpr int NWAVES = WS * WN, NTHREADS = 64 * NWAVES;
  static constexpr int TH = WS * MT, TW = 32;
  static constexpr int HALO = (TAPS == 9) ? 1 : 0;
  static constexpr int HR = TH + 2 * HALO, HC = TW + 2 * HALO;
  static constexpr int BN = WN * NT * 32;
  // ring slots: the TT frames of a stage + the TT frames that are in flight -- one new frame per stage inside a channel
  // chunk, and ALL TT first frames of the next chunk during the last stage of a chunk (the ring used to be refilled at
  // every chunk boundary with all waves waiting: a memory round trip, 8 times per workgroup for c_in = 256)
  static constexpr int R = 2 * TT;
  static constexpr int X_SLOTS = 4 * HR * HC;         // 16-B slots per ring frame
  static constexpr int X_BYTES = ((X_SLOTS + 63) / 64) * 1024;  // whole 64-slot wave-pieces (the tail is padding)
  static constexpr int W_SLOTS = TPS * 4 * BN;
  static constexpr int W_BYTES = ((W_SLOTS + 63) / 64) * 1024;
  static constexpr int NTG = TAPS / TPS;
  static constexpr int LDS_BYTES = R * X_BYTES + 2 * W_BYTES;
  static_assert(TAPS % TPS == 0, "tap groups must tile the taps");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

// CIN: compile-time c_in (0 = runtime).  The 256-channel instances get their own symbol, so the dominant
// launches (fast_conv1 forward) are identifiable in a rocprofv3 kernel trace, and a constant chunk count.
// (the body is a __device__ function: the buffer-descriptor type it uses exists only in device compilation)
template <int DT, int TAPS, int TPS, int TT, int MT, int NT, int WS, int WN, int CIN>
__device__ __forceinline__ void conv3d_body(const ConvArgs& a) {
  typedef ConvCfg<DT, TAPS, TPS, TT, MT, NT, WS, WN> C;
  // F8 (e4m3 operands, SFVOS_FP8): a 16-byte chunk holds 16 channels, a pixel's 64-byte group 64; one
  // v_mfma_scale_f32_32x32x64_f8f6f4 consumes the whole group (chunks hh and 2+hh per lane half), so a tap is ONE
  // k-step instead of two; results leave as bf16, de-quantised per output channel (as in the frame-split kernel).
  constexpr bool F8 = DT == SFVOS_FP8;
  constexpr int YDT = YOf<DT>::DTY;
  typedef typename Elt<YDT>::type T;   // element type of y
  constexpr int CE = Elt<DT>::CE, CEY = Elt<YDT>::CE, CK = 4 * CE;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const ring = smem;
  char* const wbase = smem + C::R * C::X_BYTES;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ws = wv % WS, wn = wv / WS;
  const int r = lane & 31, hh = lane >> 5;
  // M16 (bf16): the MFMAs are v_mfma_f32_16x16x32_bf16 -- one instruction consumes a pixel's whole 64-byte channel group
  // (K = 32), a wave's 32 px x 32 ch tile is four 16 x 16 blocks [pixel half][channel half].  Same LDS images, same
  // number of 16-byte fragment reads per tap (two pixel halves + two channel halves instead of two k-steps of each
  // operand), half the accumulator updates per FLOP: under the board's power limit the cheaper instruction is
  // the faster kernel (timing-only build with two 16x16x32 on each 32x32x16's operands: -3 ... -6 % on the slow pathway).
#ifdef SFVOS_WIDE_M32  // A/B builds only: the 32x32x16 form
  constexpr bool M16 = false;
#else
  constexpr bool M16 = DT == SFVOS_BF16;
#endif
  const int p16 = lane & 15, kg = lane >> 4;

  // workgroup -> (level, clip, frame block, channel block, pixel tile)
  int lvl = 0;
#pragma unroll
  for (int l = 1; l < SFVOS_MAX_LEVELS; ++l)
    if (l < a.lv.n && (int)blockIdx.x >= a.lv.wg_begin[l]) lvl = l;
  const int H = a.lv.H[lvl], W = a.lv.W[lvl], tiles_w = a.lv.tiles_w[lvl], tiles_h = a.lv.tiles_h[lvl];
  // XCD-aware order: workgroups are dealt round-robin over the 8 XCDs (speed only, never
  // correctness), so the t_blocks frame blocks of one pixel tile -- which stream the same input
  // frames -- get ids that are equal mod 8 and adjacent in time: they share one XCD's L2.
  int bid = blockIdx.x - a.lv.wg_begin[lvl];
  const int per_group = 8 * a.t_blocks;
  const int grp = bid / per_group, rem = bid - grp * per_group;
  const int tb = rem >> 3;
  const int ntile = tiles_h * tiles_w, tgroups = (ntile + 7) >> 3;
  const int tile = (grp % tgroups) * 8 + (rem & 7);
  if (tile >= ntile) return;  // padding workgroup (whole workgroup, before any barrier)
  const int bn = grp / tgroups;
  const int nb = bn % a.n_blocks, b = bn / a.n_blocks;
  const int th = tile / tiles_w, tw = tile - th * tiles_w;
  const int h0 = th * C::TH, w0 = tw * C::TW, n0 = nb * C::BN, tb0 = a.t_first + tb * TT;

  const int ncc = (CIN ? CIN : a.c_in) / CK;
  const long long HWp = (long long)H * W;
  // frame 0 of this clip inside the x buffer (t_alloc frames per clip, the conv's window starts at t_offset)
  const long long xfs_bytes = a.lv.xfs[lvl] * a.x_pitch_bytes;  // bytes between consecutive frames of this level
  const char* xclip = a.x + (a.lv.xpos[lvl] + b * a.lv.xbs[lvl]) * a.x_pitch_bytes;

  f32x16 acc[M16 ? 1 : TT][M16 ? 1 : MT][M16 ? 1 : NT];
  f32x4 acc4[M16 ? TT : 1][M16 ? MT : 1][M16 ? NT : 1][2][2];  // M16: [pixel half][channel half] blocks of the tile
#pragma unroll
  for (int j = 0; j < TT; ++j)
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int q = 0; q < NT; ++q)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          if constexpr (M16) acc4[j][i][q][e >> 3][(e >> 2) & 1][e & 3] = 0.f;
          else acc[j][i][q][e] = 0.f;
        }

  // ---- staging: buffer_load ... lds with per-lane offsets fixed for the whole kernel (see conv3d_fs_kernel):
  // padding pixels / frames outside the clip are zero-filled by the descriptor's range check.  Pixel-major halo
  // image, slot = (row*HC + col)*4 + (chunk ^ ((col>>2)&3)): four consecutive lanes copy one pixel's 64-byte
  // run (coalesced) and the 16 lanes of a ds_read_b128 group still cover all 64 banks.
  constexpr int XWP = (C::X_SLOTS + 63) / 64, WWP = (C::W_SLOTS + 63) / 64;  // 64-slot wave-pieces
  constexpr int NWV = C::NWAVES;
  constexpr int NX = (XWP + NWV - 1) / NWV, NW = (WWP + NWV - 1) / NWV;
  constexpr unsigned OOB = 0x80000000u;
  unsigned xo[NX], wo[NW];
#pragma unroll
  for (int it = 0; it < NX; ++it) {
    const int sl = it * C::NTHREADS + tid;
    const int cq = sl & 3, rc = sl >> 2, col = rc % C::HC, row = rc / C::HC;
    const int j = cq ^ ((col >> 2) & 3);
    const int h = h0 + row - C::HALO, w = w0 + col - C::HALO;
    const bool ok = sl < C::X_SLOTS && (unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W;
    xo[it] = ok ? (unsigned)(((long long)h * W + w) * a.x_pitch_bytes + j * 16) : OOB;
  }
  int xsw[2][C::HALO ? 3 : 1];  // lane part of an A-fragment address: k-step st, column shift dw
#pragma unroll
  for (int st = 0; st < 2; ++st)
#pragma unroll
    for (int dw = 0; dw < (C::HALO ? 3 : 1); ++dw)
      xsw[st][dw] = ((r + dw) * 4 + ((2 * st + hh) ^ (((r + dw) >> 2) & 3))) * 16;
  // M16: lane = (MFMA row p16, 16-byte chunk kg); column shift dw.  Row p16 of the operand is pixel 4 (p16 & 3) + (p16 >> 2)
  // of the half (any assignment of pixels to rows works, the epilogue undoes it): with rows = consecutive pixels the
  // ds_read_b128 lane groups ({0-3, 12-15, 20-27}, ...) met each 16-byte slot twice (SQ_LDS_BANK_CONFLICT = a third of the
  // LDS cycles); with this 4 x 4 transpose every group covers all 64 banks for all three shifts (checked exhaustively
  // against the group table).  The second pixel half is 16 pixels = 1024 bytes further (16 pixels do not change the
  // swizzle term): a compile-time constant on the read
  int xsw16[C::HALO ? 3 : 1];
#pragma unroll
  for (int dw = 0; dw < (C::HALO ? 3 : 1); ++dw) {
    const int px = 4 * (p16 & 3) + (p16 >> 2) + dw;
    xsw16[dw] = (px * 4 + (kg ^ ((px >> 2) & 3))) * 16;
  }
#pragma unroll
  for (int it = 0; it < NW; ++it) {
    const int sl = it * C::NTHREADS + tid;
    const int tj = sl / C::BN, n = sl - tj * C::BN;  // tj = tap_local*4 + chunk
    wo[it] = (sl < C::W_SLOTS && n0 + n < a.c_out) ? (unsigned)((tj * a.c_out + n) * 16) : OOB;
  }
  const int lds_wave_off = wv * 1024;
  const int frame_bytes = (int)(HWp * a.x_pitch_bytes);

  struct Dma {
    const char* xsrc; char* xb; int xrec; bool do_x;
    const char* wsrc; char* wb; int wrec; bool do_w;
  };
  constexpr int NPIECE = NX + NW;
  auto wrap = [](int sl) { return sl >= C::R ? sl - C::R : sl; };
  auto prep_frame = [&](Dma& d, int cc, int i, int slot) {  // halo tile of input frame i, chunk cc -> ring slot
    const int t = tb0 - a.pad_t + i, ft = a.t_offset + t;  // frame of the conv's window / of the x buffer
    // outside the window (temporal padding) or outside the buffer (zero frames "by pointer", model.py:215-225):
    // an empty descriptor, the copy writes zeros
    const bool t_ok = (unsigned)t < (unsigned)a.t_in && (unsigned)ft < (unsigned)a.t_alloc;
    d.do_x = true;
    d.xrec = t_ok ? frame_bytes - (a.x_pitch_bytes > 64 ? cc * 64 : 0) : 0;
    d.xb = ring + slot * C::X_BYTES + lds_wave_off;
    d.xsrc = xclip + (long long)(t_ok ? ft : 0) * xfs_bytes + cc * a.x_chunk_bytes;
  };
  auto prep_w = [&](Dma& d, int cc, int dt, int tg, int s) {  // weight slice of stage s: [TPS taps][4 chunks][BN]
    d.do_w = true;
    d.wb = wbase + (s & 1) * C::W_BYTES + lds_wave_off;
    d.wsrc = a.wp + (((long long)(cc * a.kt + dt) * TAPS + tg * TPS) * 4 * a.c_out + n0) * 16;
    d.wrec = (TPS * 4 * a.c_out - n0) * 16;
  };
  auto piece = [&](const Dma& d, int p) {
    if (p < NX) {
      if (d.do_x && p * NWV + wv < XWP) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)d.xsrc, 0, d.xrec, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (SFVOS_LDS void*)(d.xb + p * (C::NTHREADS * 16)), 16, xo[p], 0, 0, 0);
      }
    } else {
      const int q = p - NX;
      if (d.do_w && q * NWV + wv < WWP) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)d.wsrc, 0, d.wrec, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (SFVOS_LDS void*)(d.wb + q * (C::NTHREADS * 16)), 16, wo[q], 0, 0, 0);
      }
    }
  };
  auto issue_all = [&](const Dma& d) {
#pragma unroll
    for (int p = 0; p < NPIECE; ++p) piece(d, p);
  };

  // ---- one stage of MFMAs (+ the next stage's DMA, interleaved) -------------------------------------
  auto compute = [&](int tg, int s, int fslot, const Dma& d) {
    // per-stage operand bases: lane part + ring slot / weight buffer; every read below adds only a
    // compile-time constant, which folds into the ds_read offset field (no VALU per read)
    const char* wbl = wbase + (s & 1) * C::W_BYTES + (hh * C::BN + wn * NT * 32 + r) * 16;
    const char* xfl[TT];
#pragma unroll
    for (int j = 0; j < TT; ++j) xfl[j] = ring + wrap(fslot + j) * C::X_BYTES + ws * MT * C::HC * 64;
    // k-step k = (tap_local, st): a PD-deep register pipeline -- the fragments of step k+PD-1 are read
    // from LDS while the MFMAs of step k run, so the wait in front of a step never covers reads that
    // were issued just before it (counted lgkmcnt, not lgkmcnt(0)).
    if constexpr (M16) {
      // A tap = four quadrant steps (pixel half ph, channel half nh) in the order (0,0) (0,1) (1,1) (1,0): consecutive steps
      // share one operand set, so each set -- A[ph]: TT x MT pixel fragments, B[nh]: NT weight fragments -- is read once per
      // tap.  Sets are read in the order they are needed, n = 4 tap + {A0, B0, B1, A1}, set n before the MFMAs of step
      // n - 1 - LA (pinned with sched_barrier: the wait in front of a step never covers reads issued just before it).
      //   LA = 1: two slots per operand -- B0[t] (live for the whole tap) and B1[t] swap slots from tap to tap, so every new
      //           set lands in the slot whose set died a step earlier: the register footprint of the old 2-deep pipeline;
      //   LA = 3: rings of 3 A / 4 B slots, for the tiles whose accumulators leave the registers.
      constexpr bool DEEP = TT * MT * NT * 16 + 4 * (3 * TT * MT + 4 * NT) <= 170;
      constexpr int LA = DEEP ? 3 : 1, NA = DEEP ? 3 : 2, NB = DEEP ? 4 : 2;
      constexpr int NSET = 4 * TPS, NQ = 4 * TPS;
      constexpr int PPQ = (NPIECE + NQ - 1) / NQ;  // DMA pieces per quadrant step
      const int dh = (TAPS == 9) ? tg : 0;
      const char* wbl16 = wbase + (s & 1) * C::W_BYTES + (kg * C::BN + wn * NT * 32 + p16) * 16;
      const char* xf16[TT];
#pragma unroll
      for (int j = 0; j < TT; ++j) xf16[j] = xfl[j] + dh * C::HC * 64;
      u32x4 as[NA][TT][MT], bs[NB][NT];
      auto slot_a = [](int t, int ph) { return (2 * t + ph) % NA; };
      auto slot_b = [](int t, int nh) { return NB == 2 ? ((t & 1) ^ nh) : (2 * t + nh) % NB; };
      auto load_set = [&](int n) {
        const int t = n >> 2, w = n & 3, dw = (TAPS == 9) ? t : 0;
        if (w == 0 || w == 3) {
          const int ph = w == 3;
#pragma unroll
          for (int j = 0; j < TT; ++j)
#pragma unroll
            for (int i = 0; i < MT; ++i) as[slot_a(t, ph)][j][i] = lds_read16(xf16[j] + xsw16[dw] + (ph * 1024 + i * C::HC * 64));
        } else {
          const int nh = w == 2;
#pragma unroll
          for (int q = 0; q < NT; ++q)
            bs[slot_b(t, nh)][q] = lds_read16(wbl16 + (t * 4 * C::BN + q * 32 + nh * 16) * 16);
        }
      };
#pragma unroll
      for (int n = 0; n <= LA && n < NSET; ++n) load_set(n);
#pragma unroll
      for (int k = 0; k < NQ; ++k) {
        if (k + 1 + LA < NSET) load_set(k + 1 + LA);
        __builtin_amdgcn_sched_barrier(0);
        const int t = k >> 2, ph = (k & 3) >> 1, nh = ((k & 3) == 1 || (k & 3) == 2) ? 1 : 0;
#pragma unroll
        for (int j = 0; j < TT; ++j)
#pragma unroll
          for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int q = 0; q < NT; ++q)
              // One asm statement per MFMA, accumulator tied ("+v"): through the builtin hipcc gives every 16x16x32 result
              // a NEW register quadruple (a 128-bit result has no tied form; 220 of 324 MFMAs of the 3 x 3 tile moved
              // their accumulator) and the 256-register tiles spilled 12-88 bytes per lane into the MFMA loop; tied they
              // need fewer registers than the 32x32x16 form did (233 against 247).  Wait states (the guide's asm rules):
              // the operands come from ds_read (hipcc counts those and waits in front of the statement), the only reader
              // of D inside the loop is the next MFMA of the same accumulator taking it whole as C (chain: none needed),
              // and the first reader outside is the epilogue, behind the nops that follow the main loop.
              asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0"
                           : "+v"(acc4[j][i][q][ph][nh]) : "v"(as[slot_a(t, ph)][j][i]), "v"(bs[slot_b(t, nh)][q]));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < PPQ; ++u)
          if (k * PPQ + u < NPIECE) piece(d, k * PPQ + u);
      }
      return;
    }
    constexpr int OPR = F8 ? 8 : 4;  // registers per operand fragment
    constexpr int PD = (!F8 && (WS * WN == 4 || TT * MT * NT * 16 + 3 * (NT + TT * MT) * OPR <= 192)) ? 3 : 2;  // registers
    constexpr int KS = F8 ? TPS : 2 * TPS;
    constexpr int PPS = (NPIECE + KS - 1) / KS;  // DMA pieces per k-step
    u32x4 bv[F8 ? 1 : PD][NT], av[F8 ? 1 : PD][TT][MT];
    i32x8 bv8[F8 ? PD : 1][NT], av8[F8 ? PD : 1][TT][MT];
    auto read8 = [&](const char* lo, const char* hi) {
      const i32x4 l = __builtin_bit_cast(i32x4, lds_read16(lo)), h = __builtin_bit_cast(i32x4, lds_read16(hi));
      return __builtin_shufflevector(l, h, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    auto load = [&](int k, int buf) {
#ifdef SFVOS_ABLATE  // timing-only builds (scratch): 1 = no A re-reads, 2 = no B re-reads, 3 = neither
      const bool skip_a = (SFVOS_ABLATE & 1) && k > 1, skip_b = (SFVOS_ABLATE & 2) && k > 1;
#else
      constexpr bool skip_a = false, skip_b = false;
#endif
      const int tp = F8 ? k : k >> 1, st = F8 ? 0 : k & 1;
      // 3x3: a tap group is one kernel row (TPS == 3): dh = tg (runtime), dw = tp (compile-time, indexes xsw)
      static_assert(TAPS == 1 || TPS == 3, "3x3 layers stage one kernel row per tap group");
      const int dh = (TAPS == 9) ? tg : 0, dw = (TAPS == 9) ? tp : 0;
      if constexpr (F8) {
#pragma unroll
        for (int q = 0; q < NT; ++q) {
          const char* wt = wbl + (tp * 4 * C::BN + q * 32) * 16;
          bv8[buf][q] = read8(wt, wt + 2 * C::BN * 16);
        }
#pragma unroll
        for (int j = 0; j < TT; ++j)
#pragma unroll
          for (int i = 0; i < MT; ++i) {
            const char* xr = xfl[j] + (i + dh) * C::HC * 64;
            av8[buf][j][i] = read8(xr + xsw[0][dw], xr + xsw[1][dw]);
          }
      } else {
#pragma unroll
        for (int q = 0; q < NT; ++q)
          if (!skip_b) bv[buf][q] = lds_read16(wbl + ((tp * 4 + 2 * st) * C::BN + q * 32) * 16);
#pragma unroll
        for (int j = 0; j < TT; ++j)
#pragma unroll
          for (int i = 0; i < MT; ++i)
            if (!skip_a)
              av[buf][j][i] = lds_read16(xfl[j] + xsw[st][dw] + (i + dh) * C::HC * 64);
      }
    };
#pragma unroll
    for (int k = 0; k < PD - 1 && k < KS; ++k) load(k, k % PD);
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      if (k + PD - 1 < KS) load(k + PD - 1, (k + PD - 1) % PD);
      // pin the order on both sides: hipcc otherwise sinks each ds_read next to its MFMA and drains
      // lgkmcnt(0) before every MFMA (LDS-latency-bound stream)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < TT; ++j)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int q = 0; q < NT; ++q) {
            if constexpr (F8)  // e4m3 x e4m3, block scales 2^0 (E8M0 127): the real scales are applied in the epilogue
              acc[j][i][q] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av8[k % PD][j][i], bv8[k % PD][q],
                                                                              acc[j][i][q], 0, 0, 0, 127, 0, 127);
            else
              Mma<F8 ? SFVOS_BF16 : DT>::run(acc[j][i][q], av[k % PD][j][i], bv[k % PD][q]);
          }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < PPS; ++u)
        if (k * PPS + u < NPIECE) piece(d, k * PPS + u);
    }
  };

  // ---- main loop -------------------------------------------------------------------------------------
  // Temporal taps that can touch a real frame for this frame block: with temporal zero padding
  // (data-gradient convs, pad_t = kt-1) most (dt, frame) pairs of a block are padding -- skip them.
  const int dt_lo = max(0, a.pad_t - tb0 - (TT - 1));
  const int dt_hi = min(a.kt - 1, a.pad_t - tb0 + a.t_in - 1);
  const int S = ncc * max(0, dt_hi - dt_lo + 1) * C::NTG;
  int s = 0;
  // The second-dispatched half of the workgroup loses VALU/LDS issue arbitration to its older SIMD
  // partner on every stage (stamps: it finishes its MFMAs ~30 % later and the older half then idles
  // at the barrier); one static priority raise evens them out.
  if (C::NWAVES == 8 && wv >= 4 && !SFVOS_DBG(32)) __builtin_amdgcn_s_setprio(1);
  int fslot = 0;  // ring slot of frame dt (the j = 0 frame of the stage); slots advance with the frames staged
  if (S > 0) {    // first chunk: its first TT frames and the first weight slice
    for (int i = 0; i < TT; ++i) {
      Dma d; d.do_w = false;
      prep_frame(d, 0, dt_lo + i, wrap(fslot + i));
      issue_all(d);
    }
    Dma d; d.do_x = false;
    prep_w(d, 0, dt_lo, 0, 0);
    issue_all(d);
  }
  for (int cc = 0; cc < ncc && S > 0; ++cc) {
    for (int dt = dt_lo; dt <= dt_hi; ++dt) {
      for (int tg = 0; tg < C::NTG; ++tg, ++s) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        SFVOS_STAMP_AT(3)
        if (!SFVOS_DBG(16)) __syncthreads();  // stage s operands landed; everyone is done with stage s-1
        SFVOS_STAMP_AT(0)
        Dma d; d.do_x = d.do_w = false;
        if (!SFVOS_DBG(2)) {
          // last temporal tap of a chunk: the first TT frames of the NEXT chunk go into the TT slots behind the live
          // ones (nothing else is copied into the ring at this tap), so they land while this tap's stages multiply
          if (tg == 0 && dt == dt_hi && cc + 1 < ncc && !SFVOS_DBG(4)) {
            for (int i = 0; i < TT; ++i) {
              Dma e; e.do_w = false;
              prep_frame(e, cc + 1, dt_lo + i, wrap(fslot + TT + i));
              issue_all(e);
            }
          }
          // next frame goes into the slot freed by frame dt-1; next stage's weights into the other buffer
          if (tg == 0 && dt < dt_hi && !SFVOS_DBG(4)) prep_frame(d, cc, dt + TT, wrap(fslot + TT));
          if (s + 1 < S && !SFVOS_DBG(8)) {
            int ntg = tg + 1, ndt = dt, ncc2 = cc;
            if (ntg == C::NTG) { ntg = 0; if (++ndt > dt_hi) { ndt = dt_lo; ++ncc2; } }
            prep_w(d, ncc2, ndt, ntg, s + 1);
          }
        }
        SFVOS_STAMP_AT(1)
        if (!SFVOS_DBG(1)) compute(tg, s, fslot, d);
        else issue_all(d);
        SFVOS_STAMP_AT(2)
      }
      fslot = wrap(fslot + 1);
    }
    fslot = wrap(fslot + TT - 1);  // the next chunk's first frame sits behind this chunk's last TT frames
  }

  if constexpr (M16) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // the last asm MFMAs' D -> the epilogue's VALU reads
  // ---- epilogue --------------------------------------------------------------------------------------
  // Each 32 px x 32 ch accumulator tile goes through a per-wave f32 scratch in LDS (the staging buffers
  // are dead) and leaves as 16-byte chunks: lane -> (pixel, 8 or 4 channels), so a pixel's 32-channel
  // segment is written (and, when accumulating, read) as whole 64/128-byte runs instead of 2-byte pieces.
  float s1[NT][M16 ? 2 : 1], s2[NT][M16 ? 2 : 1];  // M16: a lane's channel is nh * 16 + p16
#pragma unroll
  for (int q = 0; q < NT; ++q)
#pragma unroll
    for (int nh = 0; nh < (M16 ? 2 : 1); ++nh) s1[q][nh] = s2[q][nh] = 0.f;
  T* yclip = (T*)a.y + (a.lv.ypos[lvl] + b * a.lv.ybs[lvl]) * a.ld_y;
  const long long yfs = a.lv.yfs[lvl];
  __syncthreads();  // every wave has finished reading the ring / weight buffers
  float* scr = (float*)smem + wv * (32 * 33);  // [32 px][32 ch], rows padded to 33 floats
  constexpr int CPP = 32 / CEY;                 // 16-byte output chunks per pixel of a tile
#pragma unroll
  for (int q = 0; q < NT; ++q) {
    const int nbase = n0 + (wn * NT + q) * 32;
    if (nbase >= a.c_out) continue;  // wave-uniform
    const float bias = a.bias ? a.bias[nbase + r] : 0.f;
    const float desc = F8 ? a.bias[a.c_out + nbase + r] : 1.f;  // e4m3: [2][c_out] = (bias, de-quantisation factor)
    const float bias16[2] = {M16 && a.bias ? a.bias[nbase + p16] : 0.f, M16 && a.bias ? a.bias[nbase + 16 + p16] : 0.f};
#pragma unroll
    for (int j = 0; j < TT; ++j) {
      const int to = tb0 + j;
      if (to >= a.t_end) continue;  // wave-uniform
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int h = h0 + ws * MT + i;
        if (h >= H) continue;  // wave-uniform
        if constexpr (M16) {
          // block (ph, nh), element e of lane (p16, kg): MFMA row 4 kg + e = pixel 16 ph + 4 e + kg, channel 16 nh + p16
#pragma unroll
          for (int ph = 0; ph < 2; ++ph)
#pragma unroll
            for (int nh = 0; nh < 2; ++nh)
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const int px = 16 * ph + 4 * e + kg;
                float v = acc4[j][i][q][ph][nh][e] + bias16[nh];
                if (a.relu) v = fmaxf(v, 0.f);
                scr[px * 33 + 16 * nh + p16] = v;
                if (w0 + px < W) { s1[q][nh] += v; s2[q][nh] += v * v; }
              }
        } else {
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int px = (e & 3) + 8 * (e >> 2) + 4 * hh;
            float v = F8 ? acc[j][i][q][e] * desc + bias : acc[j][i][q][e] + bias;
            if (a.relu) v = fmaxf(v, 0.f);
            scr[px * 33 + r] = v;
            if (w0 + px < W) { s1[q][0] += v; s2[q][0] += v * v; }
          }
        }
        T* yrow = yclip + (to * yfs + (long long)h * W + w0) * a.ld_y + nbase;
#pragma unroll
        for (int it = 0; it < (32 * CPP) / 64; ++it) {
          const int idx = it * 64 + lane, px = idx / CPP, ch = (idx % CPP) * CEY;
          float f[CEY];
#pragma unroll
          for (int u = 0; u < CEY; ++u) f[u] = scr[px * 33 + ch + u];
          if (w0 + px < W) {
            T* dst = yrow + (long long)px * a.ld_y + ch;
            if (a.accumulate) {
              const u32x4 old = *(const u32x4*)dst;
              T oldv[CEY];
              __builtin_memcpy(oldv, &old, 16);
#pragma unroll
              for (int u = 0; u < CEY; ++u) f[u] += Elt<YDT>::to_f32(oldv[u]);
            }
            T outv[CEY];
#pragma unroll
            for (int u = 0; u < CEY; ++u) outv[u] = Elt<YDT>::from_f32(f[u]);
            u32x4 o;
            __builtin_memcpy(&o, outv, 16);
            *(u32x4*)dst = o;
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
      if constexpr (M16) {
#pragma unroll
        for (int nh = 0; nh < 2; ++nh) {  // the four lane groups hold four pixel quarters of channel 16 nh + p16
          float t1 = s1[q][nh] + __shfl_xor(s1[q][nh], 16), t2 = s2[q][nh] + __shfl_xor(s2[q][nh], 16);
          t1 += __shfl_xor(t1, 32); t2 += __shfl_xor(t2, 32);
          if (lane < 16) {
            red[((wv * NT + q) * 32 + 16 * nh + lane) * 2 + 0] = t1;
            red[((wv * NT + q) * 32 + 16 * nh + lane) * 2 + 1] = t2;
          }
        }
      } else {
        const float t1 = s1[q][0] + __shfl_xor(s1[q][0], 32), t2 = s2[q][0] + __shfl_xor(s2[q][0], 32);
        if (lane < 32) {
          red[((wv * NT + q) * 32 + lane) * 2 + 0] = t1;
          red[((wv * NT + q) * 32 + lane) * 2 + 1] = t2;
        }
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
      const long long prow = a.lv.row_begin[lvl] +
                             ((long long)(b * a.t_blocks_total + a.tb_offset + tb) * tiles_h + th) * tiles_w + tw;
      a.stat_part[(prow * 2 + 0) * a.c_out + n0 + tid] = t1;
      a.stat_part[(prow * 2 + 1) * a.c_out + n0 + tid] = t2;
    }
  }
}

template <int DT, int TAPS, int TPS, int TT, int MT, int NT, int WS, int WN, int CIN = 0>
__global__ __launch_bounds__(64 * WS * WN, (WS * WN) / 4) void conv3d_kernel(ConvArgs a) {
  conv3d_body<DT, TAPS, TPS, TT, MT, NT, WS, WN, CIN>(a);
}

// ---- frame-split kernel for kt x 3 x 3 convs with c_out <= 32 ----------------------------------------
// With one 32-channel output tile every pixel fragment feeds ONE MFMA, so in the generic kernel above the
// LDS, not the matrix pipe, paces such layers (1.25 ds_read_b128 per MFMA per wave).  Here the eight waves of
// a workgroup split the 8 rows x 32 px x TT frames x 32 channels tile by FRAME, ROW GROUP and K HALF:
//   wave (j, rs, kh): output frame j, rows rs*MT .. rs*MT+MT-1, 16-byte chunks {2kh, 2kh+1} of every 64-byte
//   channel group.  For each column shift dw it walks the MT+2 halo rows once: the fragment of halo row rr
//   serves the MFMAs of the three (output row i, dh) pairs with i + dh = rr, and the three weight fragments
//   (dh, dw) stay in registers for the whole walk -> (3 (MT+2) + 9) reads per 9 MT MFMAs (0.54 per MFMA at
//   MT = 8).  The two K halves of a tile are summed through LDS once, after the K loop; each partner then
//   finishes MT/2 of the rows (bias, store, statistics).
// Ring, LDS images, staging, XCD order and the statistics rows are those of conv3d_kernel.
template <int DT, int TT, int RS>
struct FsCfg {
  static constexpr int NWAVES = 8, NTHREADS = 512;
  static constexpr int MT = 8 / RS, HM = MT / 2, TH = 8, TW = 32, HR = TH + 2, HC = TW + 2, BN = 32;
  static constexpr int R = TT + 1;
  static constexpr int X_SLOTS = 4 * HR * HC;
  static constexpr int X_BYTES = ((X_SLOTS + 63) / 64) * 1024;  // whole 64-slot wave-pieces (the tail is padding)
  static constexpr int W_SLOTS = 9 * 4 * BN;
  static constexpr int W_BYTES = W_SLOTS * 16;
  static constexpr int XCH_BYTES = NWAVES * HM * 4096;  // K-half exchange: HM accumulator tiles per wave
  static constexpr int STAGE_BYTES = R * X_BYTES + 2 * W_BYTES;
  static constexpr int LDS_BYTES = STAGE_BYTES > XCH_BYTES ? STAGE_BYTES : XCH_BYTES;
  static_assert(TT * RS == 4 && HM >= 1, "eight waves = TT frames x RS row groups x 2 K halves");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

// M16 (bf16 only): the MFMAs are v_mfma_f32_16x16x32_bf16 -- one instruction consumes a pixel's whole 64-byte
// channel group, so the two wave halves split the 32 pixel columns instead of K (no exchange at the end).  Same
// cycles per FLOP as 32x32x16, but the chip holds a higher clock on this shape under load.
// (TAG: a second kernel that instantiates the same body needs its own specialisation -- host compilation rejects a
// second reference to one specialisation of this device-only template)
// STAT (input-stationary sweep; TT = 1, bf16): the data gradient of a conv with ONE output frame (fast_conv3: its input
// gradient has kt frames, each of which meets exactly one temporal tap of the single dy frame).  As single-frame blocks
// that is kt workgroups per pixel tile, each copying the same dy halo tile and one weight slice for one stage of MFMAs.
// Here ONE workgroup per pixel tile copies the dy tile once (ring slot 1; slot 0 is the epilogue's scratch), then walks
// the output frames: weight slice of the frame's tap (double-buffered), one stage, store the frame.
template <int DT, int TT, int RS, int CIN, bool M16, int TAG = 0, bool STAT = false>
__device__ __forceinline__ void fs_body(const ConvArgs& a, const int wg, const ConvArgs::Part& pt) {
  typedef FsCfg<DT, TT, RS> C;
  // F8 (e4m3 operands, SFVOS_FP8): v_mfma_scale_f32_32x32x64_f8f6f4 -- one instruction consumes a pixel's whole
  // 64-byte group (64 channels), the two wave halves split the wave's ROWS; results leave as bf16, de-quantised
  // per output channel.  Everything about staging is shared: a 16-byte chunk just holds 16 channels.
  constexpr bool F8 = DT == SFVOS_FP8;
  constexpr int YDT = YOf<DT>::DTY;
  typedef typename Elt<YDT>::type T;   // element type of y
  constexpr int CE = Elt<DT>::CE, CEY = Elt<YDT>::CE, CK = 4 * CE, MT = C::MT, HM = C::HM;
  static_assert(!M16 || DT == SFVOS_BF16, "16x16x32 is the bf16 path");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const ring = smem;
  char* const wbase = smem + C::R * C::X_BYTES;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kh = wv >> 2, jf = (wv & 3) / RS, rs = (wv & 3) % RS;  // half partners (wv, wv^4) share a SIMD
  const int r = lane & 31, hh = lane >> 5;   // 32x32x16 fragments: row r, K half hh
  const int p16 = lane & 15, g16 = lane >> 4;  // 16x16x32 fragments: row p16, 16-byte chunk g16

  int lvl = 0;
#pragma unroll
  for (int l = 1; l < SFVOS_MAX_LEVELS; ++l)
    if (l < a.lv.n && wg >= a.lv.wg_begin[l] * pt.t_blocks) lvl = l;
  const int H = a.lv.H[lvl], W = a.lv.W[lvl], tiles_w = a.lv.tiles_w[lvl], tiles_h = a.lv.tiles_h[lvl];
  int bid = wg - a.lv.wg_begin[lvl] * pt.t_blocks;  // XCD-aware order, as in conv3d_kernel
  const int per_group = 8 * pt.t_blocks;
  const int grp = bid / per_group, rem = bid - grp * per_group;
  const int tb = rem >> 3;
  const int ntile = tiles_h * tiles_w, tgroups = (ntile + 7) >> 3;
  const int tile = (grp % tgroups) * 8 + (rem & 7);
  if (tile >= ntile) return;
  const int b = grp / tgroups;
  const int th = tile / tiles_w, tw = tile - th * tiles_w;
  const int h0 = th * C::TH, w0 = tw * C::TW, tb0 = pt.t_first + tb * TT;

  const int NF = TT + a.kt - 1;
  const int ncc = (CIN ? CIN : a.c_in) / CK;
  const long long HWp = (long long)H * W;
  const long long xfs_bytes = a.lv.xfs[lvl] * a.x_pitch_bytes;  // bytes between consecutive frames of this level
  const char* xclip = a.x + (a.lv.xpos[lvl] + b * a.lv.xbs[lvl]) * a.x_pitch_bytes;

  f32x16 acc[M16 ? 1 : (F8 ? HM : MT)];   // 32x32 tiles: [row]
  f32x4 acc16[M16 ? MT : 1][2];  // 16x16 tiles: [row][channel half]
#pragma unroll
  for (int i = 0; i < (M16 ? 1 : (F8 ? HM : MT)); ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
#pragma unroll
  for (int i = 0; i < (M16 ? MT : 1); ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc16[i][0][e] = acc16[i][1][e] = 0.f;

  // ---- staging: buffer_load ... lds (LDS-DMA through a buffer descriptor).  A copy is (descriptor of the frame /
  // weight slice, per-lane byte offset fixed for the whole kernel): padding pixels and frames outside the clip
  // carry an out-of-range offset / an empty descriptor and the hardware range check writes zeros for them, so a
  // piece costs two scalar instructions and no vector ALU work.  Pieces are whole wave-instructions (64 slots);
  // wave w owns pieces w, w+8, w+16 of an image.
  constexpr int XWP = (C::X_SLOTS + 63) / 64, WWP = (C::W_SLOTS + 63) / 64;  // wave-pieces per frame / weight slice
  constexpr int NX = (XWP + 7) / 8, NW = (WWP + 7) / 8;
  static_assert(C::X_BYTES >= XWP * 1024 && C::W_BYTES >= WWP * 1024, "staging images hold whole wave-pieces");
  constexpr unsigned OOB = 0x80000000u;
  unsigned xo[NX], wo[NW];
#pragma unroll
  for (int it = 0; it < NX; ++it) {
    // pixel-major image: slot = (row*HC + col)*4 + (chunk ^ swz(col)).  Four consecutive lanes copy the four
    // 16-byte chunks of ONE pixel (a 64-byte run of global memory), so the copy is coalesced; the XOR spreads
    // the 16 lanes of a ds_read_b128 group over all 64 banks: swz = (col>>2)&3 for the 32x32x16 fragments (same
    // chunk, 16 columns), 2*((col>>2)&1) for the 16x16x32 ones (two chunks x 8 columns per group).
    const int sl = it * C::NTHREADS + tid;
    const int cq = sl & 3, rc = sl >> 2, col = rc % C::HC, row = rc / C::HC;
    const int j = cq ^ (M16 ? 2 * ((col >> 2) & 1) : ((col >> 2) & 3));
    const int h = h0 + row - 1, w = w0 + col - 1;
    const bool ok = sl < C::X_SLOTS && (unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W;
    xo[it] = ok ? (unsigned)(((long long)h * W + w) * a.x_pitch_bytes + j * 16) : OOB;
  }
  int xsw[3];  // lane part of an A-fragment address for column shift dw (chunk 2kh+hh of pixel column r+dw)
  int xsw8[2][3];  // F8: both halves of the 32-byte operand: chunks hh and 2+hh
#pragma unroll
  for (int dw = 0; dw < 3; ++dw) {
    const int col = M16 ? kh * 16 + p16 + dw : r + dw;
    xsw[dw] = M16 ? (col * 4 + (g16 ^ (2 * ((col >> 2) & 1)))) * 16 : (col * 4 + ((2 * kh + hh) ^ ((col >> 2) & 3))) * 16;
#pragma unroll
    for (int st = 0; st < 2; ++st) xsw8[st][dw] = ((r + dw) * 4 + ((2 * st + hh) ^ (((r + dw) >> 2) & 3))) * 16;
  }
#pragma unroll
  for (int it = 0; it < NW; ++it) {
    const int sl = it * C::NTHREADS + tid;
    const int tj = sl / C::BN, n = sl - tj * C::BN;  // tj = tap*4 + chunk
    wo[it] = (sl < C::W_SLOTS && n < a.c_out) ? (unsigned)((tj * a.c_out + n) * 16) : OOB;
  }
  const int lds_wave_off = wv * 1024;
  const int frame_bytes = (int)(HWp * a.x_pitch_bytes), wslice_bytes = 9 * 4 * a.c_out * 16;

  struct Dma {
    const char* xsrc; char* xb; int xrec; bool do_x;
    const char* wsrc; char* wb; bool do_w;
  };
  constexpr int NPIECE = NX + NW;
  auto wrap = [](int sl) { return sl >= C::R ? sl - C::R : sl; };
  auto prep_frame = [&](Dma& d, int cc, int i, int slot) {
    const int t = tb0 - a.pad_t + i, ft = a.t_offset + t;  // frame of the conv's window / of the x buffer
    const bool t_ok = (unsigned)t < (unsigned)a.t_in && (unsigned)ft < (unsigned)a.t_alloc;
    d.do_x = true;
    // frame outside the window or outside the buffer: empty descriptor, all zeros.  (pyramid NDHWC: the chunk
    // offset sits inside the pixel)
    d.xrec = t_ok ? frame_bytes - (a.x_pitch_bytes > 64 ? cc * 64 : 0) : 0;
    d.xb = ring + slot * C::X_BYTES + lds_wave_off;
    d.xsrc = xclip + (long long)(t_ok ? ft : 0) * xfs_bytes + cc * a.x_chunk_bytes;
  };
  auto prep_w = [&](Dma& d, int cc, int dt, int s) {  // weight slice of stage s: [9 taps][4 chunks][32]
    d.do_w = true;
    d.wb = wbase + (s & 1) * C::W_BYTES + lds_wave_off;
    d.wsrc = a.wp + ((long long)(cc * a.kt + dt) * 9 * 4 * a.c_out) * 16;
  };
  auto piece = [&](const Dma& d, int p) {
    if (p < NX) {
      if (d.do_x && p * 8 + wv < XWP) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)d.xsrc, 0, d.xrec, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (SFVOS_LDS void*)(d.xb + p * (C::NTHREADS * 16)), 16, xo[p], 0, 0, 0);
      }
    } else {
      const int q = p - NX;
      if (d.do_w && q * 8 + wv < WWP) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)d.wsrc, 0, wslice_bytes, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (SFVOS_LDS void*)(d.wb + q * (C::NTHREADS * 16)), 16, wo[q], 0, 0, 0);
      }
    }
  };
  auto issue_all = [&](const Dma& d) {
#pragma unroll
    for (int p = 0; p < NPIECE; ++p) piece(d, p);
  };

  // ---- stage bookkeeping: stage = (channel group cc, temporal tap dt); fslot = ring slot of frame dt --------
  const int dt_lo = max(0, a.pad_t - tb0 - (TT - 1));  // temporal taps that can touch a real frame (see conv3d_kernel)
  const int dt_hi = min(a.kt - 1, a.pad_t - tb0 + a.t_in - 1);
  const int S = ncc * max(0, dt_hi - dt_lo + 1);
  struct Stage { int cc, dt, fslot; };
  auto next_of = [&](const Stage& x) {
    Stage n = x;
    if (x.dt < dt_hi) { n.dt = x.dt + 1; n.fslot = wrap(x.fslot + 1); }
    else { n.cc = x.cc + 1; n.dt = dt_lo; n.fslot = (n.cc * NF + dt_lo) % C::R; }
    return n;
  };
  // the copy that runs DURING stage x (index s): the next ring frame of this chunk and the next stage's weights
  auto prep_stage = [&](Dma& d, const Stage& x, int s) {
    d.do_x = d.do_w = false;
    if (SFVOS_DBG(2)) return;
    if (x.dt < dt_hi) prep_frame(d, x.cc, x.dt + TT, wrap(x.fslot + TT));
    if (s + 1 < S) {
      const Stage n = next_of(x);
      prep_w(d, n.cc, n.dt, s + 1);
    }
  };

  // ---- one stage: 9 MT MFMAs per wave.  Its copy (descriptor d, prepared during the PREVIOUS stage) is issued
  // in the first steps, so it has most of the stage to land; the scalar work for the next stage's descriptor
  // runs after the first MFMA group, in the shadow of the matrix pipe, not between barrier and first MFMA.
  auto compute = [&](int s, const Stage& cur, const Dma& d, Dma& dn, const Stage& nx) {
    // F8: the wave owns rows [kh*HM, kh*HM+HM) of its row group; operand = chunks hh and 2+hh (32 bytes)
    constexpr int MTW = F8 ? HM : MT;
    const char* wbl = wbase + (s & 1) * C::W_BYTES +
                      (M16 ? g16 * C::BN + p16 : F8 ? hh * C::BN + r : (2 * kh + hh) * C::BN + r) * 16;
    const char* xfl = ring + wrap(cur.fslot + jf) * C::X_BYTES + (rs * MT + (F8 ? kh * HM : 0)) * C::HC * 64;
    constexpr int ROWS = MTW + 2, NA = 3 * ROWS, PDA = 3, NH = M16 ? 2 : 1;
    constexpr int PSTRIDE = 2 * NPIECE <= NA ? 2 : 1;  // copies go out in the first steps of the stage
    static_assert(NPIECE <= NA, "DMA pieces must fit the step count");
    u32x4 av[F8 ? 1 : PDA][1], bw[F8 ? 1 : 2][3][NH];
    i32x8 av8[F8 ? PDA : 1], bw8[F8 ? 2 : 1][3];   // F8: 32-byte operands = chunks hh and 2+hh, one register octet
    typedef __attribute__((ext_vector_type(4))) int i32x4;
    auto read8 = [&](const char* lo, const char* hi) {
      const i32x4 l = __builtin_bit_cast(i32x4, lds_read16(lo)), h = __builtin_bit_cast(i32x4, lds_read16(hi));
      return __builtin_shufflevector(l, h, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    auto load_a = [&](int t) {
      const int dw = t / ROWS, rr = t - dw * ROWS;
#ifdef SFVOS_FS_ABLATE  // timing-only builds: 1 = the pixel fragments of a stage are read once (t < PDA), 2 = the weight fragments once
      if ((SFVOS_FS_ABLATE & 1) && t >= PDA) return;
#endif
      if constexpr (F8) {
        av8[t % PDA] = read8(xfl + xsw8[0][dw] + rr * C::HC * 64, xfl + xsw8[1][dw] + rr * C::HC * 64);
      } else {
        av[t % PDA][0] = lds_read16(xfl + xsw[dw] + rr * C::HC * 64);
      }
    };
    auto load_b = [&](int dw) {
#ifdef SFVOS_FS_ABLATE
      if ((SFVOS_FS_ABLATE & 2) && dw > 0) {   // (real values in both buffers: a zero or undefined operand would falsify the power)
        if (dw == 1) {
#pragma unroll
          for (int dh = 0; dh < 3; ++dh) {
            if constexpr (F8) bw8[1][dh] = bw8[0][dh];
            else {
#pragma unroll
              for (int nh = 0; nh < NH; ++nh) bw[1][dh][nh] = bw[0][dh][nh];
            }
          }
        }
        return;
      }
#endif
#pragma unroll
      for (int dh = 0; dh < 3; ++dh) {
        if constexpr (F8) {
          const char* wt = wbl + ((dh * 3 + dw) * 4 * C::BN) * 16;
          bw8[dw & 1][dh] = read8(wt, wt + 2 * C::BN * 16);
        } else {
#pragma unroll
          for (int nh = 0; nh < NH; ++nh)  // M16: nh = channel half
            bw[dw & 1][dh][nh] = lds_read16(wbl + ((dh * 3 + dw) * 4 * C::BN + nh * 16) * 16);
        }
      }
    };
    load_b(0);
    load_a(0);
    load_a(1);
#pragma unroll
    for (int t = 0; t < NA; ++t) {
      const int dw = t / ROWS, rr = t - dw * ROWS;
      if (t + PDA - 1 < NA) {
        if ((t + PDA - 1) % ROWS == 0) load_b((t + PDA - 1) / ROWS);
        load_a(t + PDA - 1);
      }
      __builtin_amdgcn_sched_barrier(0);  // reads of step t+2 stay ahead of the MFMAs of step t
#pragma unroll
      for (int dh = 0; dh < 3; ++dh) {
        const int i = rr - dh;
        if (i >= 0 && i < MTW) {
          if constexpr (M16) {
#pragma unroll
            for (int nh = 0; nh < 2; ++nh)
              acc16[i][nh] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av[t % PDA][0]),
                                                                    __builtin_bit_cast(bf16x8, bw[dw & 1][dh][nh]),
                                                                    acc16[i][nh], 0, 0, 0);
          } else if constexpr (F8) {
            // e4m3 x e4m3 (cbsz = blgp = 0), block scales 2^0 (E8M0 127): the real scales are applied in the epilogue
            acc[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av8[t % PDA], bw8[dw & 1][dh], acc[i], 0, 0, 0, 127,
                                                                     0, 127);
          } else {
            Mma<DT>::run(acc[i], av[t % PDA][0], bw[dw & 1][dh][0]);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (t % PSTRIDE == 0 && t / PSTRIDE < NPIECE) piece(d, t / PSTRIDE);
      if (t == 1) {
        if constexpr (STAT) dn.do_x = dn.do_w = false;
        else if (s + 1 < S) prep_stage(dn, nx, s + 1);
        else dn.do_x = dn.do_w = false;
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

  float s1 = 0.f, s2 = 0.f;  // per-lane partial statistics of channel r (32x32) / channels p16, 16+p16 (16x16)
  float s1b = 0.f, s2b = 0.f;
  T* yclip = (T*)a.y + (a.lv.ypos[lvl] + b * a.lv.ybs[lvl]) * a.ld_y;
  const long long yfs = a.lv.yfs[lvl];
  // ---- epilogue, 16x16 tiles (bf16): the wave's 16 pixel columns x 32 channels of each row go through a per-wave
  // f32 scratch (the first 17 KB of LDS) and leave as one 16-byte chunk per lane (lane -> pixel lane/4, channels
  // 8*(lane%4)..+7)
  auto store_m16 = [&](const int to) {
    float* scr = (float*)smem + wv * (16 * 33);
    const float bias0 = a.bias ? a.bias[p16] : 0.f, bias1 = a.bias ? a.bias[16 + p16] : 0.f;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int h = h0 + rs * MT + i;
      if (h >= H) continue;  // wave-uniform
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int px = 4 * g16 + e;
        float v0 = acc16[i][0][e] + bias0, v1 = acc16[i][1][e] + bias1;
        if (a.relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
        scr[px * 33 + p16] = v0;
        scr[px * 33 + 16 + p16] = v1;
        if (w0 + kh * 16 + px < W) { s1 += v0; s2 += v0 * v0; s1b += v1; s2b += v1 * v1; }
      }
      const int px = lane >> 2, ch = (lane & 3) * 8;
      float f[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) f[u] = scr[px * 33 + ch + u];
      if (w0 + kh * 16 + px < W) {
        T* dst = yclip + (to * yfs + (long long)h * W + w0 + kh * 16 + px) * a.ld_y + ch;
        if (a.accumulate) {
          const u32x4 old = *(const u32x4*)dst;
          T oldv[8];
          __builtin_memcpy(oldv, &old, 16);
#pragma unroll
          for (int u = 0; u < 8; ++u) f[u] += Elt<YDT>::to_f32(oldv[u]);
        }
        T outv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) outv[u] = Elt<YDT>::from_f32(f[u]);
        u32x4 o;
        __builtin_memcpy(&o, outv, 16);
        *(u32x4*)dst = o;
      }
    }
  };

  if constexpr (STAT) {
    static_assert(TT == 1 && M16, "the input-stationary sweep is a single-frame bf16 body");
    // the window's only real frame (t = 0 = tb0 - pad_t + i for i = pad_t: tb0 is 0 here) -> ring slot 1, once
    {
      Dma f; f.do_w = false;
      prep_frame(f, 0, a.pad_t, 1);
      issue_all(f);
      Dma w; w.do_x = false;
      prep_w(w, 0, a.pad_t - pt.t_first, 0);
      issue_all(w);
    }
    int k = 0;
    for (int to = pt.t_first; to < pt.t_end; ++to, ++k) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();   // this frame's weight slice (and, the first time, the dy tile) landed; the other buffer is free
      Dma dw; dw.do_x = dw.do_w = false;
      if (to + 1 < pt.t_end) prep_w(dw, 0, a.pad_t - (to + 1), k + 1);   // output frame t meets tap pad_t - t only
      const Stage cur{0, a.pad_t - to, 1};
      Dma dn;
      compute(k, cur, dw, dn, cur);
      store_m16(to);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc16[i][0][e] = acc16[i][1][e] = 0.f;
    }
    return;   // (no statistics rows: a data gradient has none)
  }

  // ---- main loop ----------------------------------------------------------------------------------------
  int s = 0;
#ifdef SFVOS_STAMP
  if (a.stamps && blockIdx.x == 300 && lane == 0) {
    a.stamps[8 * 128 * 4 + wv * 4 + 0] = __builtin_amdgcn_s_memtime();
    a.stamps[8 * 128 * 4 + wv * 4 + 1] = __builtin_amdgcn_s_memrealtime();
  }
#endif
  Stage cur{0, dt_lo, dt_lo % C::R};
  Dma d; d.do_x = d.do_w = false;
  if (S > 0) {
    prep_stage(d, cur, 0);
    Dma w0; w0.do_x = false;
    prep_w(w0, 0, dt_lo, 0);
    issue_all(w0);
  }
  for (; s < S; ++s) {
    if (cur.dt == dt_lo) {
      // chunk prologue: refill the ring with the first TT frames of this chunk.  All waves must have
      // finished the previous chunk's last stage before its live slots are overwritten.
      if (cur.cc > 0) __syncthreads();
      for (int i = 0; i < TT; ++i) {
        Dma f; f.do_w = false;
        prep_frame(f, cur.cc, dt_lo + i, wrap(cur.fslot + i));
        issue_all(f);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    SFVOS_STAMP_AT(3)
    if (!SFVOS_DBG(16)) __syncthreads();
    SFVOS_STAMP_AT(0)
    const Stage nx = next_of(cur);
    Dma dn;
    SFVOS_STAMP_AT(1)
    if (!SFVOS_DBG(1)) compute(s, cur, d, dn, nx);
    else { issue_all(d); if (s + 1 < S) prep_stage(dn, nx, s + 1); else dn.do_x = dn.do_w = false; }
    SFVOS_STAMP_AT(2)
    d = dn;
    cur = nx;
  }

#ifdef SFVOS_STAMP
  if (a.stamps && blockIdx.x == 300 && lane == 0) {
    a.stamps[8 * 128 * 4 + wv * 4 + 2] = __builtin_amdgcn_s_memtime();
    a.stamps[8 * 128 * 4 + wv * 4 + 3] = __builtin_amdgcn_s_memrealtime();
  }
#endif
  const int to = tb0 + jf;
  __syncthreads();  // ring / weight buffers are dead
  if constexpr (M16) {
    if (to < pt.t_end) store_m16(to);
  } else {
// ---- K-half exchange: each partner hands the other the tiles it will not finish ----------------------
    f32x16 fin[HM];
    if constexpr (F8) {   // the wave already holds complete sums of its HM rows
#pragma unroll
      for (int ii = 0; ii < HM; ++ii) fin[ii] = acc[ii];
    } else {
      f32x4* mine = (f32x4*)(smem + wv * (HM * 4096)) + lane;
      const f32x4* theirs = (const f32x4*)(smem + (wv ^ 4) * (HM * 4096)) + lane;
      if (kh == 0) {
  #pragma unroll
        for (int ii = 0; ii < HM; ++ii)
  #pragma unroll
          for (int g = 0; g < 4; ++g)
            mine[(ii * 4 + g) * 64] = f32x4{acc[HM + ii][4 * g], acc[HM + ii][4 * g + 1], acc[HM + ii][4 * g + 2],
                                            acc[HM + ii][4 * g + 3]};
      } else {
  #pragma unroll
        for (int ii = 0; ii < HM; ++ii)
  #pragma unroll
          for (int g = 0; g < 4; ++g)
            mine[(ii * 4 + g) * 64] = f32x4{acc[ii][4 * g], acc[ii][4 * g + 1], acc[ii][4 * g + 2], acc[ii][4 * g + 3]};
      }
      __syncthreads();
  #pragma unroll
      for (int ii = 0; ii < HM; ++ii) {
        if (kh == 0) fin[ii] = acc[ii];
        else fin[ii] = acc[HM + ii];
  #pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 o = theirs[(ii * 4 + g) * 64];
          // fixed order: K half 0 + K half 1
  #pragma unroll
          for (int u = 0; u < 4; ++u) fin[ii][4 * g + u] = kh == 0 ? fin[ii][4 * g + u] + o[u] : o[u] + fin[ii][4 * g + u];
        }
      }
      __syncthreads();  // exchange area is dead: the per-wave transposition scratch below overlays it
    }
  
    // ---- epilogue: rows kh*HM .. kh*HM+HM-1 of the wave's frame (as conv3d_kernel's, one channel tile) ----
    float* scr = (float*)smem + wv * (32 * 33);
    constexpr int CPP = 32 / CEY;
    if (r < a.c_out && to < pt.t_end) {  // c_out is a multiple of 32: always true for r; kept for symmetry
      const float bias = a.bias ? a.bias[r] : 0.f;
      const float desc = F8 ? a.bias[a.c_out + r] : 1.f;  // e4m3: [2][c_out] = (bias, de-quantisation factor)
  #pragma unroll
      for (int ii = 0; ii < HM; ++ii) {
        const int h = h0 + rs * MT + kh * HM + ii;
        if (h >= H) continue;  // wave-uniform
  #pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int px = (e & 3) + 8 * (e >> 2) + 4 * hh;
          float v = F8 ? fin[ii][e] * desc + bias : fin[ii][e] + bias;
          if (a.relu) v = fmaxf(v, 0.f);
          scr[px * 33 + r] = v;
          if (w0 + px < W) { s1 += v; s2 += v * v; }
        }
        T* yrow = yclip + (to * yfs + (long long)h * W + w0) * a.ld_y;
  #pragma unroll
        for (int it = 0; it < (32 * CPP) / 64; ++it) {
          const int idx = it * 64 + lane, px = idx / CPP, ch = (idx % CPP) * CEY;
          float f[CEY];
  #pragma unroll
          for (int u = 0; u < CEY; ++u) f[u] = scr[px * 33 + ch + u];
          if (w0 + px < W) {
            T* dst = yrow + (long long)px * a.ld_y + ch;
            if (a.accumulate) {
              const u32x4 old = *(const u32x4*)dst;
              T oldv[CEY];
              __builtin_memcpy(oldv, &old, 16);
  #pragma unroll
              for (int u = 0; u < CEY; ++u) f[u] += Elt<YDT>::to_f32(oldv[u]);
            }
            T outv[CEY];
  #pragma unroll
            for (int u = 0; u < CEY; ++u) outv[u] = Elt<YDT>::from_f32(f[u]);
            u32x4 o;
            __builtin_memcpy(&o, outv, 16);
            *(u32x4*)dst = o;
          }
        }
      }
    }
}
  if (a.stat_part) {
    __syncthreads();
    float* red = (float*)smem;  // [8 waves][32][2]
    if constexpr (M16) {
      float t1 = s1 + __shfl_xor(s1, 16), t2 = s2 + __shfl_xor(s2, 16);
      float t1b = s1b + __shfl_xor(s1b, 16), t2b = s2b + __shfl_xor(s2b, 16);
      t1 += __shfl_xor(t1, 32); t2 += __shfl_xor(t2, 32); t1b += __shfl_xor(t1b, 32); t2b += __shfl_xor(t2b, 32);
      if (lane < 16) {
        red[(wv * 32 + lane) * 2 + 0] = t1;
        red[(wv * 32 + lane) * 2 + 1] = t2;
        red[(wv * 32 + 16 + lane) * 2 + 0] = t1b;
        red[(wv * 32 + 16 + lane) * 2 + 1] = t2b;
      }
    } else {
      const float t1 = s1 + __shfl_xor(s1, 32), t2 = s2 + __shfl_xor(s2, 32);
      if (lane < 32) {
        red[(wv * 32 + lane) * 2 + 0] = t1;
        red[(wv * 32 + lane) * 2 + 1] = t2;
      }
    }
    __syncthreads();
    if (tid < 32 && tid < a.c_out) {
      float u1 = 0.f, u2 = 0.f;
#pragma unroll
      for (int k = 0; k < C::NWAVES; ++k) {
        u1 += red[(k * 32 + tid) * 2 + 0];
        u2 += red[(k * 32 + tid) * 2 + 1];
      }
      const long long prow = a.lv.row_begin[lvl] +
                             ((long long)(b * a.t_blocks_total + pt.tb_offset + tb) * tiles_h + th) * tiles_w + tw;
      a.stat_part[(prow * 2 + 0) * a.c_out + tid] = u1;
      a.stat_part[(prow * 2 + 1) * a.c_out + tid] = u2;
    }
  }
}

template <int DT, int CIN = 0>
__global__ __launch_bounds__(512, 2) void conv3d_fs_kernel(ConvArgs a) {
  int wg = blockIdx.x;  // workgroup-uniform three-way split: blocks of 4 frames, then of 2, then of 1
#ifdef SFVOS_STAMP  // diagnostic build: per-workgroup timeline (start, end, HW_ID, XCC_ID)
  unsigned long long* tl = a.stamps ? a.stamps + 8 * 128 * 4 + 64 + 4ull * blockIdx.x : nullptr;
  if (tl && threadIdx.x == 0) {
    tl[0] = __builtin_amdgcn_s_memrealtime();
    tl[2] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    tl[3] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
  }
#endif
  constexpr bool M16 = DT == SFVOS_BF16;
  if (wg < a.part[0].wgs) fs_body<DT, 4, 1, CIN, M16>(a, wg, a.part[0]);
  else if (wg - a.part[0].wgs < a.part[1].wgs) fs_body<DT, 2, 2, CIN, M16>(a, wg - a.part[0].wgs, a.part[1]);
  else fs_body<DT, 1, 4, CIN, M16>(a, wg - a.part[0].wgs - a.part[1].wgs, a.part[2]);
#ifdef SFVOS_STAMP
  if (tl && threadIdx.x == 0) tl[1] = __builtin_amdgcn_s_memrealtime();
#endif
}

// Launches that consist of single-frame blocks only (T_out = 1: fast_conv3's forward): the 1-frame body alone needs
// 2 ring slots + 2 weight buffers = 80 KB and fewer registers, so TWO workgroups share a CU (4 waves per SIMD): each
// covers the other's memory round trip per stage (a 1-frame block prefetches one frame) and 362 workgroups fit one round.
template <int DT, int CIN = 0>
__global__ __launch_bounds__(512, 4) void conv3d_fs1_kernel(ConvArgs a) {
  constexpr bool M16 = DT == SFVOS_BF16;
  const int wg = blockIdx.x;
  fs_body<DT, 1, 4, CIN, M16, 1>(a, wg, a.part[2]);
}

// The input-stationary sweep (fs_body STAT): one workgroup per pixel tile walks all output frames of a data gradient whose
// input has a single frame.  80 KB of LDS, two workgroups per CU.
template <int DT, int CIN = 0>
__global__ __launch_bounds__(512, 4) void conv3d_fs1s_kernel(ConvArgs a) {
  constexpr bool M16 = DT == SFVOS_BF16;
  const int wg = blockIdx.x;
  fs_body<DT, 1, 4, CIN, M16, 2, true>(a, wg, a.part[2]);
}

// ---- host-side planning ----------------------------------------------------------------------------
struct ConvPlan {
  int stat;    // family 3: input-stationary sweep over the output frames (conv3d_fs1s_kernel)
  int family;  // 0 narrow 1x1 (c_out <= 32), 1 mid (c_out == 64, 1x1), 2 wide, 3 frame-split (c_out <= 32, 3x3),
               // 4 lateral forward (lateral.hip: its own kernel, one statistics row per workgroup)
  int TT, MT, NT, TH, BN;
  int t_blocks, n_blocks, t_out;  // t_blocks = total frame blocks
  int n_launch, l_tt[3], l_blocks[3], l_first[3];  // launches: blocks of l_tt frames starting at frame l_first
  ConvLevels lv;
};

// Frame blocks: fewest blocks of at most max_tt frames, sizes as even as possible and NO padded frame:
// rem blocks of base+1 frames (first launch) followed by nblk-rem blocks of base frames (second launch).
static void split_frames(int t_out, int max_tt, ConvPlan* p);

static void split_frames(int t_out, int max_tt, ConvPlan* p) {
  const int nblk = ceil_div(t_out, max_tt), base = t_out / nblk, rem = t_out % nblk;
  p->t_blocks = nblk;
  p->n_launch = 0;
  if (rem > 0) {
    p->l_tt[p->n_launch] = base + 1; p->l_blocks[p->n_launch] = rem; p->l_first[p->n_launch] = 0;
    ++p->n_launch;
  }
  p->l_tt[p->n_launch] = base; p->l_blocks[p->n_launch] = nblk - rem; p->l_first[p->n_launch] = rem * (base + 1);
  ++p->n_launch;
  p->TT = p->l_tt[0];
}

// Frame-split kernel: n4 blocks of 4 frames, n2 of 2, n1 of 1 per pixel tile, all in ONE launch, dispatched
// big blocks first.  Workgroups of a launch go to compute units as these free up, so the launch ends when the
// last unit drains: with equal blocks the final round can be nearly empty (362 tiles x 5 blocks = 7.07 rounds
// of 256 -> 8).  The split is chosen by simulating that greedy dispatch with relative block costs; trading a
// block of 4 for two of 2 costs some efficiency per frame but lets short blocks fill the tail.
static double greedy_makespan(long long units, const int n[3], const double cost[3], int cus) {
  double t[64]; long long k[64]; int g = 1;  // groups of units that free up at the same time
  t[0] = 0.0; k[0] = cus;
  for (int c = 0; c < 3; ++c) {
    long long left = units * n[c];
    while (left > 0) {
      int e = 0;
      for (int i = 1; i < g; ++i) if (t[i] < t[e]) e = i;
      const long long m = left < k[e] ? left : k[e];
      const double done = t[e] + cost[c];
      k[e] -= m; left -= m;
      if (k[e] == 0) { t[e] = t[g - 1]; k[e] = k[g - 1]; --g; }
      int j = 0;
      for (; j < g; ++j) if (t[j] == done) break;
      if (j == g) { if (g == 64) return 1e30; t[g] = done; k[g] = 0; ++g; }
      k[j] += m;
    }
  }
  double end = 0.0;
  for (int i = 0; i < g; ++i) if (k[i] > 0 && t[i] > end) end = t[i];
  return end;
}

static void split_frames_balanced(int t_out, long long units, ConvPlan* p) {
  static const double cost[3] = {1.0, 0.7, 0.5};  // measured relative workgroup times of 4 / 2 / 1 frame blocks
  const int cus = device_cu_count();
  int best[3] = {t_out / 4, (t_out % 4) / 2, t_out % 2};
  double best_t = greedy_makespan(units, best, cost, cus);
  for (int n4 = t_out / 4 - 1; n4 >= 0 && n4 >= t_out / 4 - 2; --n4) {
    const int rest = t_out - 4 * n4;
    const int cand[3] = {n4, rest / 2, rest % 2};
    const double t = greedy_makespan(units, cand, cost, cus);
    if (t < best_t * 0.995) { best_t = t; best[0] = cand[0]; best[1] = cand[1]; best[2] = cand[2]; }
  }
#ifdef SFVOS_DIAG  // tuning aid of diagnostic builds only: "n4,n2,n1" (must cover t_out exactly)
  if (const char* ov = getenv("SFVOS_FS_SPLIT")) {
    int v[3];
    if (sscanf(ov, "%d,%d,%d", &v[0], &v[1], &v[2]) == 3 && v[0] >= 0 && v[1] >= 0 && v[2] >= 0 &&
        4 * v[0] + 2 * v[1] + v[2] == t_out) { best[0] = v[0]; best[1] = v[1]; best[2] = v[2]; }
  }
#endif
  p->t_blocks = 0;
  p->n_launch = 3;  // parts of the single launch; empty parts have 0 blocks
  int first = 0;
  for (int c = 0; c < 3; ++c) {
    const int tt = 4 >> c;
    p->l_tt[c] = tt; p->l_blocks[c] = best[c]; p->l_first[c] = first;
    first += best[c] * tt;
    p->t_blocks += best[c];
  }
  p->TT = 4;
}

static int make_plan(const sfvos_conv_desc* d, ConvPlan* p) {
  SFVOS_REQUIRE(d != nullptr, "conv: null desc");
  p->stat = 0;
  SFVOS_REQUIRE(d->struct_size == (int)sizeof(sfvos_conv_desc),
                "conv: sfvos_conv_desc.struct_size is %d, this library's struct has %d bytes (stale binding?)",
                d->struct_size, (int)sizeof(sfvos_conv_desc));
  SFVOS_REQUIRE(d->dtype == SFVOS_F32 || d->dtype == SFVOS_BF16 || d->dtype == SFVOS_FP8, "conv: bad dtype %d", d->dtype);
  SFVOS_REQUIRE(d->taps == 9 || d->taps == 1, "conv: taps must be 9 or 1, got %d", d->taps);
  SFVOS_REQUIRE(d->c_in > 0 && d->c_in % 32 == 0 && d->c_out > 0 && d->c_out % 32 == 0,
                "conv: channels must be positive multiples of 32 (c_in %d, c_out %d)", d->c_in, d->c_out);
  SFVOS_REQUIRE(d->c_out <= 256, "conv: c_out %d > 256 unsupported", d->c_out);
  SFVOS_REQUIRE(d->batch >= 1 && d->t_in >= 1 && d->kt >= 1, "conv: bad extent");
  SFVOS_REQUIRE(d->t_alloc >= 1, "conv: t_alloc %d", d->t_alloc);
  // level-major x: the window [t_offset, t_offset + t_in) may stick out of the buffer's t_alloc frames on either side;
  // those frames are zeros (never read).  The frame-major ring is addressed modulo its slots by the caller: in range.
  SFVOS_REQUIRE(d->x_frame_stride == 0 || (d->t_offset >= 0 && d->t_alloc >= d->t_offset + d->t_in),
                "conv: ring window [t_offset %d, +t_in %d) exceeds its %d slots", d->t_offset, d->t_in, d->t_alloc);
  SFVOS_REQUIRE(d->t_offset > -(1 << 20) && d->t_offset < (1 << 20) && d->t_in < (1 << 20), "conv: window out of range");
  SFVOS_REQUIRE(d->pad_t >= 0 && d->pad_t < d->kt + 1, "conv: bad pad_t %d", d->pad_t);
  const int ce = d->dtype == SFVOS_BF16 ? 8 : d->dtype == SFVOS_FP8 ? 16 : 4;
  if (d->dtype == SFVOS_FP8)
    SFVOS_REQUIRE(d->taps == 9 && d->c_in % 64 == 0 && d->accumulate == 0 && d->pad_t == 0,
                  "conv: e4m3 operands are implemented for forward 3x3 convs with c_in %% 64 == 0 (no accumulate)");
  SFVOS_REQUIRE(d->ld_y >= d->c_out, "conv: pitch smaller than channel count");
  SFVOS_REQUIRE(d->x_frame_stride >= 0 && d->y_frame_stride >= 0, "conv: negative frame stride");
  if (d->x_group_stride != 0) {
    SFVOS_REQUIRE((d->dtype == SFVOS_BF16 || d->dtype == SFVOS_FP8) && d->x_group_stride > 0 &&
                  d->x_group_stride % 16 == 0,
                  "conv: the channel-group-major x layout needs bf16 / e4m3, stride a positive multiple of 16 elements");
  } else {
    SFVOS_REQUIRE(d->ld_x >= d->c_in, "conv: pitch smaller than channel count");
    SFVOS_REQUIRE(d->ld_x % ce == 0, "conv: ld_x %d must be a multiple of %d (16-byte chunks)", d->ld_x, ce);
  }
  SFVOS_REQUIRE(d->pyr.n_levels >= 1 && d->pyr.n_levels <= SFVOS_MAX_LEVELS, "conv: n_levels %d out of [1,%d]",
                d->pyr.n_levels, SFVOS_MAX_LEVELS);
  p->t_out = d->t_in + 2 * d->pad_t - d->kt + 1;
  SFVOS_REQUIRE(p->t_out >= 1, "conv: kernel longer than padded input (t_in %d, kt %d, pad_t %d)", d->t_in, d->kt,
                d->pad_t);
  constexpr int WNW = 2;  // waves across the channel dimension (mid / wide families)
  bool lateral_ok = true;
#ifdef SFVOS_DIAG
  lateral_ok = !getenv("SFVOS_NO_LATERAL_KERNEL");
#endif
  if (lateral_ok && lateral_fwd_applies(d)) {
    p->family = 4; split_frames(p->t_out, 3, p); p->MT = 1; p->NT = 1; p->TH = 8; p->BN = 64;
  } else if (d->c_out <= 32 && d->taps == 9) {
    long long units = 0;
    for (int l = 0; l < d->pyr.n_levels && l < SFVOS_MAX_LEVELS; ++l)
      units += (long long)d->batch * ceil_div(d->pyr.h[l] > 0 ? d->pyr.h[l] : 1, 8) * ceil_div(d->pyr.w[l] > 0 ? d->pyr.w[l] : 1, 32);
    p->family = 3; split_frames_balanced(p->t_out, units, p); p->NT = 1; p->TH = 8; p->BN = 32;
    if (d->t_in == 1 && d->pad_t == d->kt - 1 && d->kt > 1 && d->dtype == SFVOS_BF16) {
      // data gradient of a conv with ONE output frame (fast_conv3): every frame of dx meets exactly one temporal tap, so
      // a multi-frame block multiplies TT taps x TT frames of which TT are real.  One workgroup per pixel tile keeps
      // the dy tile and walks the t_out frames, one weight slice and one stage each (conv3d_fs1s_kernel; single-frame
      // blocks copied the same dy tile t_out times).
      p->stat = 1;
      p->l_blocks[0] = p->l_blocks[1] = 0; p->l_blocks[2] = 1;
      p->l_first[0] = p->l_first[1] = p->l_first[2] = 0;
      p->t_blocks = 1;
    }
  } else if (d->c_out <= 32) {
    p->family = 0; split_frames(p->t_out, 4, p); p->MT = 1; p->NT = 1; p->TH = 8; p->BN = 32;
  } else if (d->c_out == 64 && d->taps == 1) {
    p->family = 1; split_frames(p->t_out, 3, p); p->MT = 2; p->NT = 2 / WNW; p->TH = 8; p->BN = 64;
  } else {
    // accumulators: TT x MT x NT tiles of 16 registers per wave: at most 9 tiles per wave
    // (e4m3 operand fragments are 8 registers each: 128 output channels per workgroup and at most two frames per
    // block there -- the wider tiles spill)
    p->family = 2; p->NT = d->dtype == SFVOS_FP8 ? 2 : (d->c_out <= 192 ? 6 : 8) / WNW;
    // one output frame (slow_conv3's forward): 128-channel tiles.  2 ring slots + 2 x 24.5 KB of weights = 76 KB and 103
    // registers, so TWO workgroups share a CU and cover each other's per-stage weight copy (0.208 -> 0.198 ms)
    int nt2_min = 192;   // (192 channels on 128-channel tiles would multiply 25 % padding: the 192-channel tile stays)
#ifdef SFVOS_DIAG
    if (const char* ov = getenv("SFVOS_NT2_MIN")) nt2_min = atoi(ov);
#endif
    if (d->dtype == SFVOS_BF16 && p->t_out == 1 && d->pad_t == 0 && d->c_out > nt2_min) p->NT = 2;
    split_frames(p->t_out, d->dtype == SFVOS_FP8 ? 2 : (p->NT == 3 ? 3 : 2), p);
    p->MT = 1; p->TH = 4;
    p->BN = 32 * p->NT * WNW;
    // Data-gradient convs (pad_t = kt-1): most (output frame, temporal tap) pairs of a multi-frame block meet only
    // padding, but the block multiplies every tap any of its frames needs.  Count the (frame, tap) slots each blocking
    // multiplies; when single-frame blocks multiply fewer, use them -- with two pixel rows per wave (8 rows x 32 px x
    // BN per workgroup), which restores the weight re-use the lost frames gave (slow_conv3's data gradient: 2 slots
    // instead of 4, 0.28 -> 0.17 ms; slow_conv2's: 4 instead of 5, 0.36 -> 0.31 ms).
    auto slots = [&](int tb0, int tt) {
      const int lo = d->pad_t - tb0 - (tt - 1) > 0 ? d->pad_t - tb0 - (tt - 1) : 0;
      const int hi = d->kt - 1 < d->pad_t - tb0 + d->t_in - 1 ? d->kt - 1 : d->pad_t - tb0 + d->t_in - 1;
      return hi >= lo ? tt * (hi - lo + 1) : 0;
    };
    int cur = 0, single = 0;
    for (int li = 0; li < p->n_launch; ++li)
      for (int b = 0; b < p->l_blocks[li]; ++b) cur += slots(p->l_first[li] + b * p->l_tt[li], p->l_tt[li]);
    for (int t = 0; t < p->t_out; ++t) single += slots(t, 1);
    if (single < cur) {
      p->n_launch = 1; p->l_tt[0] = 1; p->l_blocks[0] = p->t_out; p->l_first[0] = 0; p->t_blocks = p->t_out; p->TT = 1;
      p->MT = 2; p->TH = 8;
    }
  }
  p->n_blocks = ceil_div(d->c_out, p->BN);
  ConvLevels& lv = p->lv;
  lv.n = d->pyr.n_levels;
  long long wg = 0, rows = 0, px = 0;
  for (int l = 0; l < SFVOS_MAX_LEVELS; ++l) {
    const bool live = l < lv.n;
    const int H = live ? d->pyr.h[l] : 1, W = live ? d->pyr.w[l] : 1;
    SFVOS_REQUIRE(H >= 1 && W >= 1, "conv: level %d has bad extent %dx%d", l, H, W);
    SFVOS_REQUIRE((long long)H * W * (d->x_group_stride ? 4 * ce : d->ld_x) * (16 / ce) < (1ll << 31),
                  "conv: level %d frame of %dx%d x pitch %d exceeds the 2 GiB per-frame offset range", l, H, W,
                  d->ld_x);
    lv.H[l] = H; lv.W[l] = W;
    lv.tiles_h[l] = ceil_div(H, p->TH); lv.tiles_w[l] = ceil_div(W, 32);
    lv.wg_begin[l] = (int)wg; lv.row_begin[l] = (int)rows;
    if (d->x_frame_stride > 0) {  // frame-major ring: frame t of every level sits at position t * x_frame_stride
      lv.xpos[l] = (long long)d->batch * px; lv.xbs[l] = (long long)H * W; lv.xfs[l] = d->x_frame_stride;
    } else {
      lv.xpos[l] = (long long)d->batch * d->t_alloc * px; lv.xbs[l] = (long long)d->t_alloc * H * W; lv.xfs[l] = (long long)H * W;
    }
    if (d->y_frame_stride > 0) {
      lv.ypos[l] = (long long)d->batch * px; lv.ybs[l] = (long long)H * W; lv.yfs[l] = d->y_frame_stride;
    } else {
      lv.ypos[l] = (long long)d->batch * p->t_out * px; lv.ybs[l] = (long long)p->t_out * H * W; lv.yfs[l] = (long long)H * W;
    }
    if (live) {
      // pixel tiles padded to groups of 8 (XCD-aware order in the kernel)
      wg += (long long)d->batch * p->n_blocks * ceil_div(lv.tiles_h[l] * lv.tiles_w[l], 8) * 8;  // x blocks per launch
      rows += (long long)d->batch * p->t_blocks * lv.tiles_h[l] * lv.tiles_w[l];
      px += (long long)H * W;
    }
    SFVOS_REQUIRE(wg < (1ll << 31) && rows < (1ll << 31), "conv: grid out of range");
  }
  lv.wg_begin[SFVOS_MAX_LEVELS] = (int)wg;
  lv.row_begin[SFVOS_MAX_LEVELS] = (int)rows;
  if (p->family == 4) {  // rows are the workgroups of the lateral kernel
    rows = 0;
    for (int l = 0; l <= SFVOS_MAX_LEVELS; ++l) {
      lv.row_begin[l] = (int)rows;
      if (l < lv.n) rows += lateral_fwd_rows(d, l);
    }
  }
  return SFVOS_OK;
}

template <int DT, int TAPS, int TPS, int TT, int MT, int NT, int WS, int WN, int CIN = 0>
static int launch(const ConvArgs& a, long long grid, hipStream_t stream) {
  typedef ConvCfg<DT, TAPS, TPS, TT, MT, NT, WS, WN> C;
  auto kern = conv3d_kernel<DT, TAPS, TPS, TT, MT, NT, WS, WN, CIN>;
  static LdsAttrOnce once;
  if (int rc = once.ensure((const void*)kern, C::LDS_BYTES, "conv")) return rc;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(C::NTHREADS), C::LDS_BYTES, stream, a);
  return check_launch("conv3d");
}

template <int DT, int CIN = 0>
static int launch_fs1s(const ConvArgs& a, long long grid, hipStream_t stream) {
  if constexpr (DT == SFVOS_BF16) {
    constexpr int LDS1 = FsCfg<DT, 1, 4>::LDS_BYTES;
    auto kern = conv3d_fs1s_kernel<DT, CIN>;
    static LdsAttrOnce once;
    if (int rc = once.ensure((const void*)kern, LDS1, "conv")) return rc;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), LDS1, stream, a);
    return check_launch("conv3d_fs1s");
  } else {
    set_error("conv: the input-stationary sweep is a bf16 kernel");
    return SFVOS_E_ARG;
  }
}

template <int DT, int CIN = 0>
static int launch_fs(const ConvArgs& a, long long grid, hipStream_t stream) {
  if constexpr (DT == SFVOS_BF16) if (a.part[0].wgs == 0 && a.part[1].wgs == 0) {  // single-frame blocks only
    constexpr int LDS1 = FsCfg<DT, 1, 4>::LDS_BYTES;
    auto kern1 = conv3d_fs1_kernel<DT, CIN>;
    static LdsAttrOnce once1;
    if (int rc = once1.ensure((const void*)kern1, LDS1, "conv")) return rc;
    hipLaunchKernelGGL(kern1, dim3((unsigned)grid), dim3(512), LDS1, stream, a);
    return check_launch("conv3d_fs1");
  }
  constexpr int LDS = FsCfg<DT, 4, 1>::LDS_BYTES;  // the largest of the three bodies
  static_assert(LDS >= FsCfg<DT, 2, 2>::LDS_BYTES && LDS >= FsCfg<DT, 1, 4>::LDS_BYTES, "LDS of the merged launch");
  auto kern = conv3d_fs_kernel<DT, CIN>;
  static LdsAttrOnce once;
  if (int rc = once.ensure((const void*)kern, LDS, "conv")) return rc;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), LDS, stream, a);
  return check_launch("conv3d_fs");
}

template <int DT, int TAPS>
static int dispatch(const ConvPlan& p, const ConvArgs& a, long long grid, hipStream_t s) {
#define SFVOS_CASE(F, TPSv, TTv, MTv, NTv, WSv, WNv) \
  if constexpr ((DT == SFVOS_FP8) == ((NTv) == 2 && (F) == 2))                                                        \
    if (p.family == F && p.TT == TTv && p.MT == MTv && p.NT == NTv) return launch<DT, TAPS, TPSv, TTv, MTv, NTv, WSv, WNv>(a, grid, s);
  // two waves per SIMD
  if constexpr (TAPS == 1) {
    // narrow 1x1 (lateral data gradient 64 -> 32): 8 rows x 32 px x TT frames x 32 channels; wave = one row
    SFVOS_CASE(0, 1, 1, 1, 1, 8, 1) SFVOS_CASE(0, 1, 2, 1, 1, 8, 1) SFVOS_CASE(0, 1, 3, 1, 1, 8, 1)
    SFVOS_CASE(0, 1, 4, 1, 1, 8, 1)
    // mid (lateral 32->64): 8 rows x 32 px x TT frames x 64 channels
    SFVOS_CASE(1, 1, 1, 2, 1, 4, 2) SFVOS_CASE(1, 1, 2, 2, 1, 4, 2) SFVOS_CASE(1, 1, 3, 2, 1, 4, 2)
  } else {
    // wide: 4 rows x 32 px x TT frames x 192/256 channels, 3 taps per stage
    SFVOS_CASE(2, 3, 1, 1, 3, 4, 2) SFVOS_CASE(2, 3, 2, 1, 3, 4, 2) SFVOS_CASE(2, 3, 3, 1, 3, 4, 2)
    SFVOS_CASE(2, 3, 1, 1, 4, 4, 2) SFVOS_CASE(2, 3, 2, 1, 4, 4, 2)
    // single-frame blocks (data gradients): 8 rows x 32 px x 192/256 channels, two rows per wave
    SFVOS_CASE(2, 3, 1, 2, 3, 4, 2) SFVOS_CASE(2, 3, 1, 2, 4, 4, 2)
    // e4m3 operands: 4 rows x 32 px x TT frames x 128 channels
    SFVOS_CASE(2, 3, 1, 1, 2, 4, 2) SFVOS_CASE(2, 3, 2, 1, 2, 4, 2)
    // bf16, one output frame: the same 128-channel tile, two workgroups per CU
    if constexpr (DT == SFVOS_BF16)
      if (p.family == 2 && p.TT == 1 && p.MT == 1 && p.NT == 2) return launch<DT, TAPS, 3, 1, 1, 2, 4, 2>(a, grid, s);
  }
#undef SFVOS_CASE
  set_error("conv: no kernel instance for family %d TT %d MT %d NT %d taps %d", p.family, p.TT, p.MT, p.NT, TAPS);
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
                            float* stat_part, sfvos_stream_t stream) {
  ConvPlan p;
  int rc = make_plan(d, &p);
  if (rc != SFVOS_OK) return rc;
  SFVOS_REQUIRE(x && w_packed && y, "conv: null pointer");
  SFVOS_REQUIRE(!(stat_part && d->accumulate), "conv: statistics are those of the conv result; not available with accumulate");
  SFVOS_REQUIRE(!(d->relu && (stat_part || d->accumulate)), "conv: the ReLU epilogue excludes statistics and accumulate");
  SFVOS_REQUIRE(d->ld_y % (d->dtype == SFVOS_F32 ? 4 : 8) == 0, "conv: ld_y %d must be a multiple of a 16-byte chunk", d->ld_y);
  SFVOS_REQUIRE(d->dtype != SFVOS_FP8 || bias != nullptr, "conv: e4m3 operands need the [2][c_out] (bias, descale) rows");
  SFVOS_REQUIRE(!(d->taps == 1 && p.family == 2), "conv: 1x1 conv with c_out > 64 has no kernel instance");
  bool lateral_ok = true;
#ifdef SFVOS_DIAG
  lateral_ok = !getenv("SFVOS_NO_LATERAL_KERNEL");
#endif
  if (!bias && !stat_part && d->taps == 1 && d->x_group_stride == 0 && d->x_frame_stride == 0 &&
      d->y_frame_stride == 0 && !d->relu && lateral_ok) {
    rc = lateral_dgrad_try(d, x, w_packed, y, (hipStream_t)stream);  // lateral data gradient: its own HBM-bound kernel
    if (rc >= 0) return rc;
  }
  if (p.family == 4) return lateral_fwd_launch(d, x, w_packed, bias, y, stat_part, (hipStream_t)stream);
  ConvArgs a;
  a.x = (const char*)x; a.wp = (const char*)w_packed; a.bias = bias; a.y = (char*)y; a.stat_part = stat_part;
  a.batch = d->batch; a.t_in = d->t_in; a.t_alloc = d->t_alloc; a.t_offset = d->t_offset; a.t_out = p.t_out;
  a.c_in = d->c_in; a.c_out = d->c_out; a.kt = d->kt;
  a.pad_t = d->pad_t; a.ld_x = d->ld_x; a.ld_y = d->ld_y; a.accumulate = d->accumulate; a.relu = d->relu;
  const int es = d->dtype == SFVOS_BF16 ? 2 : d->dtype == SFVOS_FP8 ? 1 : 4;  // bytes per element of x
  a.x_pitch_bytes = d->x_group_stride ? 64 : d->ld_x * es;
  a.x_chunk_bytes = d->x_group_stride ? d->x_group_stride * es : 64;
  a.n_blocks = p.n_blocks; a.t_blocks_total = p.t_blocks;
  a.debug = 0;
#ifdef SFVOS_DIAG
  { const char* dbg = getenv("SFVOS_CONV_DEBUG"); a.debug = dbg ? atoi(dbg) : 0; }
#endif
#ifdef SFVOS_STAMP
  { const char* sp = getenv("SFVOS_STAMP_PTR"); a.stamps = sp ? (unsigned long long*)strtoull(sp, nullptr, 0) : nullptr; }
#endif
  hipStream_t s = (hipStream_t)stream;
  int tb_offset = 0;
  if (p.family == 3) {  // one launch: blocks of 4, 2 and 1 frames as parts 0, 1, 2
    a.lv = p.lv;        // wg_begin counts per frame block; the kernel scales it by the part's block count
    a.t_blocks = a.t_first = a.t_end = a.tb_offset = 0;
    long long grid = 0;
    for (int c = 0; c < 3; ++c) {
      ConvArgs::Part& pt = a.part[c];
      pt.t_blocks = p.l_blocks[c]; pt.t_first = p.l_first[c]; pt.t_end = pt.t_first + p.l_tt[c] * pt.t_blocks;
      if (p.stat && c == 2) pt.t_end = p.t_out;   // the one "block" of the stationary sweep covers every output frame
      pt.tb_offset = tb_offset;
      tb_offset += pt.t_blocks;
      const long long wgs = (long long)p.lv.wg_begin[SFVOS_MAX_LEVELS] * pt.t_blocks;
      SFVOS_REQUIRE(wgs < (1ll << 31), "conv: grid out of range");
      pt.wgs = (int)wgs;
      grid += wgs;
    }
    SFVOS_REQUIRE(grid > 0 && grid < (1ll << 31), "conv: grid %lld out of range", grid);
    if (p.stat) return launch_fs1s<SFVOS_BF16>(a, grid, s);   // (make_plan sets it for bf16 only)
    if (d->dtype == SFVOS_BF16)
      return d->c_in == 256 ? launch_fs<SFVOS_BF16, 256>(a, grid, s) : launch_fs<SFVOS_BF16>(a, grid, s);
    if (d->dtype == SFVOS_FP8) return d->c_in == 256 ? launch_fs<SFVOS_FP8, 256>(a, grid, s) : launch_fs<SFVOS_FP8>(a, grid, s);
    return launch_fs<SFVOS_F32>(a, grid, s);
  }
  for (int c = 0; c < 3; ++c) a.part[c] = ConvArgs::Part{0, 0, 0, 0, 0};
  for (int li = 0; li < p.n_launch; ++li) {
    p.TT = p.l_tt[li];
    a.t_blocks = p.l_blocks[li]; a.t_first = p.l_first[li]; a.t_end = a.t_first + p.l_tt[li] * p.l_blocks[li];
    a.tb_offset = tb_offset;
    tb_offset += p.l_blocks[li];
    a.lv = p.lv;  // workgroup prefix of this launch: per-level unit count x its frame blocks
    for (int l = 0; l <= SFVOS_MAX_LEVELS; ++l) a.lv.wg_begin[l] = p.lv.wg_begin[l] * a.t_blocks;
    const long long grid = a.lv.wg_begin[SFVOS_MAX_LEVELS];
    SFVOS_REQUIRE(grid > 0 && grid < (1ll << 31), "conv: grid %lld out of range", grid);
    if (d->dtype == SFVOS_BF16)
      rc = d->taps == 9 ? dispatch<SFVOS_BF16, 9>(p, a, grid, s) : dispatch<SFVOS_BF16, 1>(p, a, grid, s);
    else if (d->dtype == SFVOS_FP8)
      rc = dispatch<SFVOS_FP8, 9>(p, a, grid, s);   // make_plan admits e4m3 for 3x3 convs only
    else
      rc = d->taps == 9 ? dispatch<SFVOS_F32, 9>(p, a, grid, s) : dispatch<SFVOS_F32, 1>(p, a, grid, s);
    if (rc != SFVOS_OK) return rc;
  }
  return SFVOS_OK;
}
