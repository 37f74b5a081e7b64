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
// bf16 3x3 layers run 16x16x32 MFMAs with K = 32 pixels from two frames or two tile rows (FPR below).
//
// One workgroup (8 waves) owns NTN n-tiles x NTC c-tiles x DG temporal taps x all spatial taps;
// wave = one (n-tile, c-tile, dt) group with TAPS accumulator tiles.  It sweeps its share of the
// TH x 16 pixel tiles; stage (tile, fo) pairs the dy frame fo with the x frames fo+dt, dt = dt0..dt0+DG-1:
// the halo tiles of the x frames live in a ring of R slots that is filled in the order the frames are
// needed -- across tile boundaries too -- and dy frames are double-buffered, so EVERY wave multiplies at
// EVERY stage (x traffic / DG, one new x tile + one dy frame per stage, one barrier per stage).
// Staging is buffer_load ... lds: pixels outside the image, channels past the tensor and the padding
// lanes of a piece carry an out-of-range offset and are zero-filled by the hardware range check.
// Split-K partials go to fp32 slabs, summed in a fixed order by wgrad_reduce_kernel
// (deterministic, no atomics), which also transposes into the state-dict layout
// [Cout][Cin][kt][kh][kw].
#include <stdlib.h>

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
  int t_in, t_alloc, t_offset, t_out, c_in, c_out, kt, ld_x, ld_y, batch;
  long long x_group_bytes;  // 0: x is pyramid NDHWC (pitch ld_x); else bytes between its 64-byte channel groups (bf16)
  int n_blocks, c_blocks, dt_blocks, psplit;
  int ntiles;  // over all levels and clips
  WgradLevels lv;
};

// NXS: x tiles a stage may copy ahead (2: one frame per stage is new when t_out > 1; DG: a conv with ONE output frame
// re-uses nothing between stages -- every stage needs DG new frames -- so the ring holds two stages, R = 2 DG)
// KS: waves that SHARE one (n-tile, c-tile, dt) group and split the tile's rows between them (summed through LDS at the
// end, in a fixed order).  fast_conv2 (kt = 11, one n-tile, one c-tile) on 8 taps per workgroup runs taps 8 + 3: five of
// sixteen wave slots idle; with 4 taps x 2 row halves it runs 4 + 4 + 3.
// FPR (bf16, 3x3, even t_out): FRAME-PAIRED stages on v_mfma_f32_16x16x32_bf16.  A stage is (tile, output frames fo, fo+1):
// K = 32 = the 16 pixels of a tile row in frame fo ++ the same 16 pixels in frame fo+1 (any assignment of the reduction
// index to k works as long as dy and x use the same one), so a fragment is one transposed read from each of two frame
// buffers -- the same tiles, the same bytes copied and read per frame as the 32x32x16 form, half the accumulator updates
// per FLOP.  Under the board's power limit (these kernels run at 1.96 GHz of 2.4) the cheaper instruction is
// the faster kernel: the same operand registers through this shape (timing-only build) took 7 % off fast_conv1's weight
// gradient.  Wave dt reads x ring frames q0+dt and q0+dt+1, so a stage keeps DG+1 frames and the ring advances two per
// stage: R >= DG + 3; 6-row tiles make that fit (and divide every DAVIS level height).  LDS rows stay [pixel][32 ch] with
// their 32-byte halves swapped on every second group of four rows (see the fragment addressing in wgrad_body).
// FPR = 2 (bf16, 3x3, any t_out): ROW-PAIRED 16x16x32 -- K = the 16 pixels of tile row r ++ those of row r + THK/2 of the
// SAME frame (second read = first + a constant), stages and ring as in the 32x32x16 form; 16 % more transposed reads (the
// row walk covers 6 pair rows x 3 shifts x 2 channel halves instead of 10 rows x 3 shifts), no other change.
template <int DT, int TAPS, int NTN, int NTC, int DG, int TH, int R, int NXS = 2, int KS = 1, int FPR = 0>
struct WgradCfg {
  static constexpr int CE = Elt<DT>::CE;
  static constexpr int SPP = 32 / CE;            // 16-B slots per pixel per 32-channel tile
  static constexpr int ROWB = SPP * 16;          // bytes per pixel row of a 32-channel tile
  static constexpr int HALO = TAPS == 9 ? 1 : 0;
  static constexpr int HR = TH + 2 * HALO, HC = 16 + 2 * HALO;
  static constexpr int NPOS = TH * 16, NHPOS = HR * HC;
  static constexpr int DY_SLOTS = NTN * NPOS * SPP;   // one dy frame
  static constexpr int XT_SLOTS = ((NHPOS * SPP + 63) / 64) * 64;  // one 32-channel x halo tile, padded to whole
  static constexpr int X_SLOTS = NTC * XT_SLOTS;                   // 64-slot wave-pieces (a piece = one channel group)
  static constexpr int DWP = (DY_SLOTS + 63) / 64, XWP = X_SLOTS / 64;
  static constexpr int FS = FPR == 1 ? 2 : 1;     // output frames per stage
  static constexpr int DYF_BYTES = DWP * 1024;    // one dy frame
  static constexpr int DY_BYTES = FS * DYF_BYTES, X_BYTES = XWP * 1024;
  static constexpr int LDS_BYTES = R * X_BYTES + 2 * DY_BYTES;
  static constexpr int THK = TH / KS;             // tile rows per wave
  static constexpr int PF = 3;                    // FPR: x fragments read ahead of their MFMAs
  static_assert(NTN * NTC * DG * KS == 8 && TH % KS == 0, "KS waves per (n-tile, c-tile, dt) group");
  static_assert(KS == 1 || (KS == 2 && DT == SFVOS_BF16 && TAPS == 9), "the row split is a bf16 3x3 configuration");
  static constexpr int RU = FPR == 2 ? THK / 2 : THK;   // rows a wave walks (row pairs: pair p = rows p and p + RU)
  static_assert(FPR != 1 || (DT == SFVOS_BF16 && TAPS == 9 && KS == 1 && NXS == 2 && R >= DG + 3),
                "frame pairs: bf16 3x3, a stage holds DG + 1 frames and two more are in flight");
  static_assert(FPR != 2 || (DT == SFVOS_BF16 && TAPS == 9 && THK % 2 == 0 && (RU * HC) % 8 == 0),
                "row pairs: bf16 3x3; the second row of a pair keeps the first one's swizzle state (a multiple of 8 LDS rows on)");
  static_assert(R > DG, "the ring holds the DG frames of a stage plus the one in flight");
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

// (the body is a __device__ function: the buffer-descriptor type it uses exists only in device compilation)
template <int DT, int TAPS, int NTN, int NTC, int DG, int TH, int R, int NXS, int KS, int FPR>
__device__ __forceinline__ void wgrad_body(const WgradArgs& a) {
  typedef WgradCfg<DT, TAPS, NTN, NTC, DG, TH, R, NXS, KS, FPR> C;
  constexpr int THK = C::THK;
  constexpr int CE = C::CE, ES = 16 / CE;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const xbase = smem;
  char* const dybase = smem + R * C::X_BYTES;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nt = wv % NTN, ct = (wv / NTN) % NTC, dg = (wv / (NTN * NTC)) % DG, ks = wv / (NTN * NTC * DG);
  const int r = lane & 31, hh = lane >> 5;

  // XCD-aware order (speed only, never correctness): workgroup ids are dealt round-robin over the 8 XCDs, so XCD x
  // runs ids x, x+8, ... in that order.  Logical work item L = ps * col + (cb * G + g): the col = c_blocks * G
  // workgroups of one pixel split ps read the same x / dy tiles.  XCD x takes the CONTIGUOUS range
  // [x*q + min(x,r), ...) of q or q+1 items (N = 8 q + r): every XCD gets the same number of workgroups (+-1) for
  // ANY split count, and the sharers of a tile sit on one XCD, adjacent in time.
  const int G = a.dt_blocks * a.n_blocks, col = G * a.c_blocks, N = a.psplit * col;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nq = N >> 3, nr = N & 7;
  if (slot >= nq + (xcd < nr ? 1 : 0)) return;  // padding workgroup (N not a multiple of 8)
  const int L = xcd * nq + min(xcd, nr) + slot;
  const int ps = L / col, ci = L - ps * col;
  const int cb = ci / G, g = ci - cb * G, nb = g % a.n_blocks, db = g / a.n_blocks;
  const int n_base = nb * NTN * 32, c_base = cb * NTC * 32, dt0 = db * DG;
  const int dt_live = min(DG, a.kt - dt0);      // temporal taps of this group that exist
  const int nxf = a.t_out + dt_live - 1;        // x frames per tile: t = dt0 .. dt0 + nxf - 1

  const int per = (a.ntiles + a.psplit - 1) / a.psplit;
  const int tile_begin = ps * per;
  const int tile_end = min(a.ntiles, tile_begin + per);
  const int ntile = max(0, tile_end - tile_begin);
  const int S = ntile * (a.t_out / C::FS);   // stages (tile, fo) / (tile, frame pair)
  const int QT = ntile * nxf;      // x tile loads, in the order they are needed

  // accumulators of the wave's 32 x 32 (n, c) tile, one per spatial tap: a 32x32x16 result tile, or (frame pairs) the
  // four 16x16x32 result blocks [2 nh + ch] -- element 4 blk + e' of the tile in both forms (slab store below)
  f32x16 acc[FPR ? 1 : TAPS];
  f32x4 acc4[FPR ? TAPS : 1][4];
  auto acc_get = [&](int t, int e) -> float { if constexpr (FPR) return acc4[t][e >> 2][e & 3]; else return acc[t][e]; };
  auto acc_add = [&](int t, int e, float v) { if constexpr (FPR) acc4[t][e >> 2][e & 3] += v; else acc[t][e] += v; };
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      if constexpr (FPR) acc4[t][e >> 2][e & 3] = 0.f; else acc[t][e] = 0.f;
    }

  const bool wave_live = (n_base + nt * 32 < a.c_out) && (c_base + ct * 32 < a.c_in) && (dg < dt_live);

  // ---- staging side.  The x loads run ahead of the dy loads (up to R frames, possibly into the next pixel
  // tile), so each stream keeps its own tile context: per-lane offsets inside a frame, frame 0, frame stride.
  constexpr int NDY = (C::DWP + 7) / 8, NXP = (C::XWP + 7) / 8;  // wave w copies wave-pieces w, w+8, ...
  constexpr unsigned OOB = 0x80000000u;
  unsigned dyo[NDY], xo[NXP];
  const char* dy_frame0 = a.dy; const char* x_frame0 = a.x;
  long long dy_fstride = 0, x_fstride = 0;
  int xi_q = 0, xi_tile = tile_begin - 1, xi_f = nxf;        // next x load: index, its tile, its frame in the tile
  int di_tile = tile_begin - 1, di_fo = a.t_out;             // next dy load
  const int lds_wave_off = wv * 1024;
  auto tile_geom = [&](int tile, int& lvl, int& b, int& h0, int& w0) {
    lvl = 0;
#pragma unroll
    for (int l = 1; l < SFVOS_MAX_LEVELS; ++l)
      if (l < a.lv.n && tile >= a.lv.tile_begin[l]) lvl = l;
    int k = tile - a.lv.tile_begin[lvl];
    const int tw = k % a.lv.tiles_w[lvl]; k /= a.lv.tiles_w[lvl];
    const int th = k % a.lv.tiles_h[lvl]; k /= a.lv.tiles_h[lvl];
    b = k; h0 = th * TH; w0 = tw * 16;
  };
  auto enter_tile_x = [&](int tile) {
    int lvl, b, h0, w0;
    tile_geom(tile, lvl, b, h0, w0);
    const int H = a.lv.H[lvl], W = a.lv.W[lvl];
    const long long HWp = (long long)H * W;
    const int pitch = a.x_group_bytes ? 64 : a.ld_x * ES;  // bytes per position
    x_fstride = HWp * pitch;
    x_frame0 = a.x + (a.lv.xpos[lvl] + (long long)b * a.t_alloc * HWp) * pitch;  // frame 0 of the clip's buffer
#pragma unroll
    for (int it = 0; it < NXP; ++it) {
      const int sl = it * 512 + tid;
      const int tct = sl / C::XT_SLOTS, ts = sl - tct * C::XT_SLOTS;
      const int jl = ts % C::SPP, hp = ts / C::SPP;
      // frame pairs: LDS row hp keeps its 16-byte chunks XOR-ed with 2 * ((hp >> 2) & 1) -- see the fragment reads
      const int j = FPR ? jl ^ (2 * ((hp >> 2) & 1)) : jl;
      const int h = h0 + hp / C::HC - C::HALO, w = w0 + hp % C::HC - C::HALO, c = c_base + tct * 32;
      const bool ok = sl < C::X_SLOTS && hp < C::NHPOS && (unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W &&
                      c < a.c_in;
      // the channel-group term of the grouped layout goes into the piece's base pointer (x_piece), not the offset
      xo[it] = ok ? (unsigned)(((long long)h * W + w) * pitch + (a.x_group_bytes ? 0 : c * ES) + j * 16) : OOB;
    }
  };
  auto enter_tile_dy = [&](int tile) {
    int lvl, b, h0, w0;
    tile_geom(tile, lvl, b, h0, w0);
    const int H = a.lv.H[lvl], W = a.lv.W[lvl];
    const long long HWp = (long long)H * W;
    dy_fstride = HWp * a.ld_y * ES;
    dy_frame0 = a.dy + (a.lv.ypos[lvl] + (long long)b * a.t_out * HWp) * a.ld_y * ES;
#pragma unroll
    for (int it = 0; it < NDY; ++it) {
      const int sl = it * 512 + tid;
      const int jl = sl % C::SPP, pos = (sl / C::SPP) % C::NPOS, tnt = sl / (C::SPP * C::NPOS);
      const int j = FPR ? jl ^ (2 * ((pos >> 2) & 1)) : jl;
      const int h = h0 + pos / 16, w = w0 + pos % 16, n = n_base + tnt * 32;
      const bool ok = sl < C::DY_SLOTS && h < H && w < W && n < a.c_out;
      dyo[it] = ok ? (unsigned)((((long long)h * W + w) * a.ld_y + n + j * CE) * ES) : OOB;
    }
  };
  struct Copy { const char* src; char* dst; int rec; };
  auto begin_x = [&](Copy& c) {  // next x tile of the load order -> ring slot xi_q % R
    if (xi_f == nxf) { xi_f = 0; enter_tile_x(++xi_tile); }
    const int ft = a.t_offset + dt0 + xi_f;  // frame of the x buffer; outside it: a zero frame (empty descriptor)
    const bool f_ok = (unsigned)ft < (unsigned)a.t_alloc;
    c.src = x_frame0 + (long long)(f_ok ? ft : 0) * x_fstride;
    c.rec = f_ok ? (int)x_fstride : 0;
    c.dst = xbase + (xi_q % R) * C::X_BYTES + lds_wave_off;
    ++xi_f; ++xi_q;
  };
  auto begin_dy = [&](Copy& c, int s, int f) {  // next dy frame = frame f (of FS) of stage s -> buffer s & 1
    if (di_fo == a.t_out) { di_fo = 0; enter_tile_dy(++di_tile); }
    c.src = dy_frame0 + (long long)di_fo * dy_fstride;
    c.rec = (int)dy_fstride;
    c.dst = dybase + (s & 1) * C::DY_BYTES + f * C::DYF_BYTES + lds_wave_off;
    ++di_fo;
  };
  auto x_piece = [&](const Copy& c, int p) {
    if (p * 8 + wv < C::XWP) {
      // wave-piece -> its 32-channel tile -> (grouped layout) that tile's channel group
      const long long goff = a.x_group_bytes * ((c_base >> 5) + (p * 8 + wv) / (C::XT_SLOTS / 64));
      lds_dma16(c.src + goff, c.rec, xo[p], (unsigned)(c.dst - smem) + p * 8192);
    }
  };
  auto dy_piece = [&](const Copy& c, int p) {
    if (p * 8 + wv < C::DWP) {
      lds_dma16(c.src, c.rec, dyo[p], (unsigned)(c.dst - smem) + p * 8192);
    }
  };
  auto load_x_now = [&]() {
    Copy c;
    begin_x(c);
#pragma unroll
    for (int p = 0; p < NXP; ++p) x_piece(c, p);
  };

  // ---- compute side: stage s multiplies dy buffer s&1 with the x frame in ring slot (q0 + dg) % R; while it
  // runs, up to two more x tiles (nx) and the next dy frame (ndy) are copied, piece by piece between MFMA groups.
  constexpr int TROWS = TAPS == 9 ? 3 : 1, TCOLS = TAPS == 9 ? 3 : 1;
  constexpr int NSTEP = THK * TROWS;
  constexpr int NCOPY = NXS * NXP + C::FS * NDY;   // copy slots of a stage: [x tile 0 pieces] ... [x tile NXS-1 pieces][dy pieces of each frame]
  static_assert(NCOPY + 2 <= 2 * NSTEP || DT != SFVOS_BF16, "the copies of a stage must fit between its MFMA steps");
  auto compute = [&](int s, int q0, int nx, bool ndy) {
    const char* dyb = dybase + (s & 1) * C::DY_BYTES + nt * (C::NPOS * C::ROWB);
    const char* xb = xbase + ((q0 + dg) % R) * C::X_BYTES + ct * (C::XT_SLOTS * 16);
    Copy cx, cd;
    // copy schedule by slot: x tile i = slot / NXP begins at slot i NXP, its pieces follow; then the dy frame
    auto copies = [&](int slot) {
#ifdef SFVOS_WG_ABLATE
      if ((SFVOS_WG_ABLATE & 4) && s > 0) {   // keep the cursors moving, copy nothing
        if (slot < NXS * NXP) { if (slot % NXP == 0 && slot / NXP < nx) begin_x(cx); }
        else if (slot < NCOPY && ndy && (slot - NXS * NXP) % NDY == 0) begin_dy(cd, s + 1, (slot - NXS * NXP) / NDY);
        return;
      }
#endif
      if (slot < NXS * NXP) {
        const int i = slot / NXP, pc = slot - i * NXP;
        if (i < nx) {
          if (pc == 0) begin_x(cx);
          x_piece(cx, pc);
        }
        return;
      }
      if (slot < NCOPY && ndy) {
        const int d = slot - NXS * NXP, f = d / NDY, pc = d - f * NDY;
        if (pc == 0) begin_dy(cd, s + 1, f);
        dy_piece(cd, pc);
      }
    };
    if (!wave_live) {  // nothing to multiply (tap / channel tile past the tensor): just feed the copies
#pragma unroll
      for (int slot = 0; slot < NCOPY; ++slot) copies(slot);
      return;
    }
    if constexpr (DT == SFVOS_BF16 && FPR) {
      // ---- frame pairs, 16x16x32 MFMAs: K = the 16 pixels of a tile row in frame fo ++ the same pixels in frame fo + 1.
      // Operand of a 16-channel half: lane (p16 = channel, g16 = k chunk) holds pixels 4 g16 .. +3 of the row in frame A
      // (first ds_read_b64_tr_b16) and in frame B (second): dy frames (fo, fo + 1), x ring frames (q0 + dt, q0 + dt + 1).
      // A 16-lane group reads 4 LDS rows x 32 bytes; the two groups of a half-wave (g16 = 0, 1 / 2, 3) read rows that are
      // 4 apart, which would collide on 64-byte rows (4 rows = 64 banks) -- so row r keeps its two 32-byte halves swapped
      // when (r >> 2) & 1 (the copies write it that way, enter_tile_*): rows 4 apart then always sit in different bank
      // halves.  Byte address of lane (g16, q, p), channel half h, rows from B on:
      //   64 (B + 4 g16 + q) + 32 (h ^ ((B + 4 g16 + q) >> 2 & 1)) + 8 p  =  (lx[B & 3] ^ 32 (((B >> 2) & 1) ^ h)) + 64 B:
      // four lane constants, one add per stage, frame and variant, one XOR where the half flips, immediates per read.
      const int g16 = lane >> 4, p16 = lane & 15, qq = p16 >> 2, pp = p16 & 3;
      // second read of a fragment: the next ring frame / dy frame (frame pairs) or RU tile rows further (row pairs)
      const int xoA = (int)(xb - smem) + ks * (THK * C::HC * C::ROWB);
      const int xoB = FPR == 1 ? (int)(xbase - smem) + ((q0 + dg + 1) % R) * C::X_BYTES + ct * (C::XT_SLOTS * 16)
                               : xoA + C::RU * C::HC * C::ROWB;
      constexpr int DYB = FPR == 1 ? C::DYF_BYTES : C::RU * 16 * C::ROWB;
      int xa[2][4];
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int lx = 64 * (4 * g16 + qq) + 8 * pp + 32 * ((g16 + ((v + qq) >> 2)) & 1);
        xa[0][v] = xoA + lx;
        xa[1][v] = xoB + lx;
      }
      const int ya = (int)(dyb - smem) + ks * (THK * 16 * C::ROWB) + 64 * (4 * g16 + qq) + 8 * pp + 32 * (g16 & 1);   // dy rows start at 16 ty: v = 0, par = 0
      // x fragments in issue order: f = (halo row rr, channel half ch of x, column shift dw); each serves the (tile row
      // rr - dh, vertical tap dh) pairs that meet its halo row, for both n halves: up to 6 MFMAs.  PF fragments in
      // flight; the dy rows roll through three slots -- row rr + 1 is read into the slot of row rr - 2 right after that
      // row's last MFMAs (fragment (rr, 1, 2), vertical tap 2).
      constexpr int RU = C::RU, NROW = RU + 2, NFR = NROW * 6;
      static_assert(NCOPY <= NFR, "one copy slot per x fragment");
      u32x4 ar[3][2], br[C::PF + 1];
      auto load_a = [&](int ty) {
#ifdef SFVOS_WG_ABLATE  // timing-only builds (wrong results): 1 = x fragments read PF + 1 times per stage, 2 = dy fragments three
        if ((SFVOS_WG_ABLATE & 2) && ty >= 3) return;   // times, 4 = no staging copies behind the first stage
#endif
#pragma unroll
        for (int nh = 0; nh < 2; ++nh) {
          const int y = (nh ? ya ^ 32 : ya) + ty * 16 * C::ROWB;
          ar[ty % 3][nh] = join(tr_read(smem + y), tr_read(smem + y + DYB));
        }
      };
      auto load_b = [&](int f) {
#ifdef SFVOS_WG_ABLATE
        if ((SFVOS_WG_ABLATE & 1) && f > C::PF) return;
#endif
        const int rr = f / 6, ch = (f / 3) & 1, dw = f % 3;
        const int rb = rr * C::HC + dw, v = rb & 3, flip = ((rb >> 2) & 1) ^ ch;   // row base of the fragment
        br[f % (C::PF + 1)] = join(tr_read(smem + (flip ? xa[0][v] ^ 32 : xa[0][v]) + rb * C::ROWB),
                                   tr_read(smem + (flip ? xa[1][v] ^ 32 : xa[1][v]) + rb * C::ROWB));
      };
      auto mma = [&](int f, int dh) {
        const int rr = f / 6, ch = (f / 3) & 1, dw = f % 3, ty = rr - dh;
        if (ty < 0 || ty >= RU) return;
#pragma unroll
        for (int nh = 0; nh < 2; ++nh)
          // block (n half, c half) of the 32 x 32 tile.  One asm statement per MFMA with the accumulator tied: through the
          // builtin hipcc gives every 128-bit MFMA result a new register quadruple (no tied form) and the rotation costs
          // ~36 registers.  Operands come from ds_read (hipcc waits in front of the statement); inside the loop D is
          // read only by the next MFMA of the same accumulator, whole, as C; the slab stores sit behind nops.
          asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0"
                       : "+v"(acc4[dh * 3 + dw][2 * nh + ch]) : "v"(ar[ty % 3][nh]), "v"(br[f % (C::PF + 1)]));
      };
      load_a(0);
      if (RU > 1) load_a(1);
#pragma unroll
      for (int f = 0; f < C::PF; ++f) load_b(f);
#pragma unroll
      for (int rr = 0; rr < NROW; ++rr)
#pragma unroll
        for (int ch = 0; ch < 2; ++ch)
#pragma unroll
          for (int dw = 0; dw < 3; ++dw) {
            const int f = (rr * 2 + ch) * 3 + dw;
            if (f + C::PF < NFR) load_b(f + C::PF);
            __builtin_amdgcn_sched_barrier(0);
            mma(f, 2);
            if (ch == 1 && dw == 2 && rr >= 1 && rr + 1 < RU) {   // row rr - 2 is dead (or never existed): fetch row rr + 1
              __builtin_amdgcn_sched_barrier(0);
              load_a(rr + 1);
              __builtin_amdgcn_sched_barrier(0);
            }
            mma(f, 1);
            mma(f, 0);
            __builtin_amdgcn_sched_barrier(0);
            copies(f);
            __builtin_amdgcn_sched_barrier(0);
          }
    } else if constexpr (DT == SFVOS_BF16) {
      // lane -> (row q, 4-column group p) of its 16-lane group's 4x16 block; block rows k0..k0+3
      const int gq = lane >> 4, i = lane & 15, qq = i >> 2, pp = i & 3;
      const int lane_off = (8 * (gq >> 1) + qq) * C::ROWB + (16 * (gq & 1) + 4 * pp) * 2;
      // every read below adds a compile-time constant (ds_read offset field); row split: this wave's THK rows
      const char* dyl = dyb + lane_off + ks * (THK * 16 * C::ROWB);
      const char* xl = xb + lane_off + ks * (THK * C::HC * C::ROWB);
      if constexpr (TAPS == 9) {
        // row walk: step = halo row rr of the x tile.  Its three column-shifted fragments are read ONCE and serve the
        // (output row ty, vertical tap dh) pairs with ty + dh = rr -- up to 9 MFMAs on 9 different accumulators;
        // the dy fragments of rows rr, rr-1, rr-2 stay in a 4-deep rolling buffer.  76 transposed reads per stage
        // instead of 160.  Fragments of step rr+1 are read while the MFMAs of step rr run (pinned order).
        constexpr int NROW = THK + 2;
        constexpr int CPS = (NCOPY + NROW - 1) / NROW;   // copy slots per row step
        u32x2 ar[4][2], br[2][3][2];
        auto load = [&](int rr) {
          if (rr < THK) {
            ar[rr & 3][0] = tr_read(dyl + rr * 16 * C::ROWB);
            ar[rr & 3][1] = tr_read(dyl + (rr * 16 + 4) * C::ROWB);
          }
#pragma unroll
          for (int dw = 0; dw < 3; ++dw) {
            br[rr & 1][dw][0] = tr_read(xl + (rr * C::HC + dw) * C::ROWB);
            br[rr & 1][dw][1] = tr_read(xl + (rr * C::HC + dw + 4) * C::ROWB);
          }
        };
        load(0);
#pragma unroll
        for (int rr = 0; rr < NROW; ++rr) {
          if (rr + 1 < NROW) load(rr + 1);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int dh = 0; dh < 3; ++dh) {
            const int ty = rr - dh;
            if (ty < 0 || ty >= THK) continue;
            const u32x4 av = join(ar[ty & 3][0], ar[ty & 3][1]);
#pragma unroll
            for (int dw = 0; dw < 3; ++dw)
              Mma<SFVOS_BF16>::run(acc[dh * 3 + dw], av, join(br[rr & 1][dw][0], br[rr & 1][dw][1]));
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int u = 0; u < CPS; ++u) copies(rr * CPS + u);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
      // step = (ty, dh): one A fragment per ty, TCOLS B fragments per step; a PD-deep register pipeline
      // reads the fragments of step+PD-1 while this step's MFMAs run (order pinned with sched_barrier)
      constexpr int PD = 3;
      static_assert(NCOPY <= NSTEP, "one copy slot per MFMA step in this path");
      u32x2 ar[PD][2], br[PD][TCOLS][2];
      auto load = [&](int step, int buf) {
        const int ty = step / TROWS, dh = step % TROWS;
        if (dh == 0) {
          ar[ty % PD][0] = tr_read(dyl + ty * 16 * C::ROWB);
          ar[ty % PD][1] = tr_read(dyl + (ty * 16 + 4) * C::ROWB);
        }
#pragma unroll
        for (int dw = 0; dw < TCOLS; ++dw) {
          br[buf][dw][0] = tr_read(xl + ((ty + dh) * C::HC + dw) * C::ROWB);
          br[buf][dw][1] = tr_read(xl + ((ty + dh) * C::HC + dw + 4) * C::ROWB);
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
        copies(step);
        __builtin_amdgcn_sched_barrier(0);
      }
      }
    } else {
#pragma unroll
      for (int slot = 0; slot < NCOPY; ++slot) copies(slot);
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

  // ---- main loop ----------------------------------------------------------------------------------------
  if (S > 0) {
#pragma unroll
    for (int f = 0; f < C::FS; ++f) {
      Copy c;
      begin_dy(c, 0, f);
#pragma unroll
      for (int p = 0; p < NDY; ++p) dy_piece(c, p);
    }
  }
  int q0 = 0, fo = 0;  // first x load of the stage (tile_local * nxf + fo), dy frame of the stage
  for (int s = 0; s < S; ++s) {
    const int need = q0 + dt_live - 1 + (C::FS - 1);  // last x load this stage reads
    if (xi_q <= need) {
      // not yet copied (start of the sweep, or a tile boundary the ring could not prefetch across): every wave
      // is done with the previous stage after this barrier, so the slots of its frames may be overwritten
      if (s > 0) __syncthreads();
#ifdef SFVOS_WG_ABLATE   // 8: the refill at a tile boundary copies nothing (cursors move)
      if ((SFVOS_WG_ABLATE & 8) && s > 0) { while (xi_q <= need) { Copy c; begin_x(c); } } else
#endif
      while (xi_q <= need) load_x_now();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // loads q < q0 + R overwrite frames that no stage >= s reads
    const int nx = max(0, min(NXS, min(QT, q0 + R) - xi_q));
    compute(s, q0, nx, s + 1 < S);
    q0 += C::FS;
    if ((fo += C::FS) == a.t_out) { fo = 0; q0 += dt_live - 1; }
  }

  if constexpr (FPR) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // the last asm MFMAs' D -> the VALU reads below
  if constexpr (KS == 2) {
    // the two row halves of a group: wave w + 4 hands its accumulators to wave w through LDS, five taps per round
    // (4 waves x 5 tiles x 4 KB = 80 KB), added as (rows 0..THK-1) + (rows THK..): a fixed order
    constexpr int TPR = 5;
    static_assert(4 * TPR * 4096 <= C::LDS_BYTES, "hand-over area");
    float* red = (float*)smem;
    __syncthreads();   // ring and dy buffers are dead
#pragma unroll
    for (int t0 = 0; t0 < TAPS; t0 += TPR) {
      if (ks == 1) {
#pragma unroll
        for (int t = t0; t < t0 + TPR && t < TAPS; ++t)
#pragma unroll
          for (int e = 0; e < 16; ++e) red[(((wv - 4) * TPR + (t - t0)) * 16 + e) * 64 + lane] = acc_get(t, e);
      }
      __syncthreads();
      if (ks == 0) {
#pragma unroll
        for (int t = t0; t < t0 + TPR && t < TAPS; ++t)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc_add(t, e, red[((wv * TPR + (t - t0)) * 16 + e) * 64 + lane]);
      }
      __syncthreads();
    }
  }
  // slab[ps][n][dt][tap][c]
  if (wave_live && ks == 0) {
    const int dt = dt0 + dg;
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        // 32x32x16: element e of lane (r, hh) = row (e & 3) + 8 (e >> 2) + 4 hh, column r;  16x16x32 blocks: element
        // 4 (2 nh + ch) + e' of lane (p16, g16) = row 16 nh + 4 g16 + e', column 16 ch + p16
        const int n = n_base + nt * 32 + (FPR ? 16 * (e >> 3) + 4 * (lane >> 4) + (e & 3) : (e & 3) + 8 * (e >> 2) + 4 * hh);
        const int c = c_base + ct * 32 + (FPR ? 16 * ((e >> 2) & 1) + (lane & 15) : r);
        a.slab[((((long long)ps * a.c_out + n) * a.kt + dt) * TAPS + tap) * a.c_in + c] = acc_get(tap, e);
      }
  }
}

template <int DT, int TAPS, int NTN, int NTC, int DG, int TH, int R, int NXS = 2, int KS = 1, int FPR = 0>
__global__ __launch_bounds__(512) void wgrad_kernel(WgradArgs a) {
  wgrad_body<DT, TAPS, NTN, NTC, DG, TH, R, NXS, KS, FPR>(a);
}

// grad_w[n][c][dt][tap] (=|+=) sum_ps slab[ps][n][dt][tap][c]
// block = 64 consecutive slab elements (16 lanes x 16-byte loads, coalesced along c) x 16 groups of splits: group g adds
// the splits g, g+16, g+32, ... in order (two running sums), the sixteen group sums are combined in a fixed order
// through LDS -- deterministic, and enough loads in flight to stream the slabs (the reduction of 256 slabs used to take as
// long as the kernel that wrote them).
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slab, int psplit, int c_out,
                                                           int c_in, int kt, int taps, float* grad_w, int accumulate) {
  __shared__ float red[16][64];
  const long long total = (long long)c_out * c_in * kt * taps;   // a multiple of 32 (c_in is)
  const int g = threadIdx.x >> 4, l = threadIdx.x & 15;
  for (long long base = (long long)blockIdx.x * 64; base < total; base += (long long)gridDim.x * 64) {
    const long long i = base + 4 * l;
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
    if (i < total) {
      int p = g;
      for (; p + 16 < psplit; p += 32) {
        s0 += *(const f32x4*)(slab + (long long)p * total + i);
        s1 += *(const f32x4*)(slab + (long long)(p + 16) * total + i);
      }
      for (; p < psplit; p += 16) s0 += *(const f32x4*)(slab + (long long)p * total + i);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) red[g][4 * l + e] = s0[e] + s1[e];
    __syncthreads();
    const long long j = base + threadIdx.x;
    if (threadIdx.x < 64 && j < total) {
      const int t = threadIdx.x;
      const float q0 = (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
      const float q1 = (red[4][t] + red[5][t]) + (red[6][t] + red[7][t]);
      const float q2 = (red[8][t] + red[9][t]) + (red[10][t] + red[11][t]);
      const float q3 = (red[12][t] + red[13][t]) + (red[14][t] + red[15][t]);
      const float sum = (q0 + q1) + (q2 + q3);
      long long k = j;
      const int c = (int)(k % c_in); k /= c_in;
      const int tap = (int)(k % taps); k /= taps;
      const int dt = (int)(k % kt); k /= kt;
      const int n = (int)k;
      float* dst = grad_w + (((long long)n * c_in + c) * kt + dt) * taps + tap;
      *dst = accumulate ? *dst + sum : sum;
    }
    __syncthreads();
  }
}

int launch_wgrad_reduce(const float* slab, int psplit, int c_out, int c_in, int kt, int taps, float* grad_w,
                        int accumulate, hipStream_t stream) {
  const long long total = (long long)c_out * c_in * kt * taps;
  long long rgrid = ceil_div64(total, 64);
  if (rgrid > 8192) rgrid = 8192;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)rgrid), dim3(256), 0, stream, slab, psplit, c_out, c_in, kt,
                     taps, grad_w, accumulate);
  return check_launch("wgrad_reduce");
}

struct WgradPlan {
  int cfg;  // 0: (1,2,4) 3x3 narrow-n ; 1: (2,2,2) 3x3 ; 2: (1,1,8) 3x3 c_in 32 (f32) / (1,1,4) x 2 row halves (bf16) ; 3: (2,1,4) 1x1 ;
            // 4: (1,1,8) 3x3 c_in 32 with ONE output frame (bf16): 4-row tiles, ring of two stages
  int NTN, NTC, DG, TH, FPR;   // FPR: 16x16x32 MFMAs, 1 = frame-paired stages (bf16 3x3, c_in >= 64, even t_out >= 4), 2 = row pairs
  int n_blocks, c_blocks, dt_blocks, psplit, t_out, ntiles;
  WgradLevels lv;
};

static int make_wgrad_plan(const sfvos_conv_desc* d, WgradPlan* p) {
  SFVOS_REQUIRE(d != nullptr, "wgrad: null desc");
  SFVOS_REQUIRE(d->struct_size == (int)sizeof(sfvos_conv_desc),
                "wgrad: sfvos_conv_desc.struct_size is %d, this library's struct has %d bytes (stale binding?)",
                d->struct_size, (int)sizeof(sfvos_conv_desc));
  SFVOS_REQUIRE(d->dtype == SFVOS_F32 || d->dtype == SFVOS_BF16, "wgrad: bad dtype");
  SFVOS_REQUIRE(d->taps == 9 || d->taps == 1, "wgrad: taps must be 9 or 1");
  SFVOS_REQUIRE(d->c_in % 32 == 0 && d->c_out % 32 == 0 && d->c_in > 0 && d->c_out > 0, "wgrad: channels % 32");
  SFVOS_REQUIRE(d->pad_t == 0, "wgrad: only forward convs (pad_t == 0) have a weight gradient here");
  SFVOS_REQUIRE(d->x_frame_stride == 0 && d->y_frame_stride == 0,
                "wgrad: frame-major ring buffers are an inference-only layout");
  const int ce = d->dtype == SFVOS_BF16 ? 8 : 4;
  SFVOS_REQUIRE(d->ld_y % ce == 0 && d->ld_y >= d->c_out, "wgrad: pitch");
  if (d->x_group_stride != 0)
    SFVOS_REQUIRE(d->dtype == SFVOS_BF16 && d->x_group_stride > 0 && d->x_group_stride % 8 == 0,
                  "wgrad: the channel-group-major x layout is bf16 only, stride a positive multiple of 8 elements");
  else
    SFVOS_REQUIRE(d->ld_x % ce == 0 && d->ld_x >= d->c_in, "wgrad: pitch");
  p->t_out = d->t_in - d->kt + 1;
  SFVOS_REQUIRE(p->t_out >= 1, "wgrad: kt > t_in");
  const bool f32 = d->dtype == SFVOS_F32;
  if (d->taps == 1) {
    p->cfg = 3; p->NTN = 2; p->NTC = 1; p->DG = 4;
  } else if (d->c_in <= 32) {
    p->cfg = 2; p->NTN = 1; p->NTC = 1; p->DG = f32 ? 8 : 4;   // bf16: 4 taps x 2 row halves
#ifdef SFVOS_DIAG
    if (getenv("SFVOS_WGRAD_KS1")) p->DG = 8;   // A/B: the 8-tap configuration
#endif
  } else if (d->c_out <= 32) {
    p->cfg = 0; p->NTN = 1; p->NTC = 2; p->DG = 4;
  } else {
    p->cfg = 1; p->NTN = 2; p->NTC = 2; p->DG = 2;
  }
  p->TH = f32 ? 4 : 8;
  p->FPR = 0;
  // the row-split configuration: 16-row tiles, 8 rows per wave (72 MFMAs per wave between barriers as in the other
  // configurations, 27 % less halo per pixel: fast_conv2 0.200 -> 0.184 ms; 8 taps per workgroup: 0.203 ms.  The same
  // decomposition with one c-tile per workgroup for fast_conv1 (c_in 256) was 3 % slower than (2 c-tiles, 4 taps): 2.58 vs
  // 2.51 ms, 104 vs 68 MB of slabs)
  if (p->cfg == 2 && !f32 && p->DG == 4) p->TH = 16;
  // bf16 3x3 layers with c_in >= 64 and an even number >= 4 of output frames: frame-paired stages on
  // v_mfma_f32_16x16x32_bf16, 6-row tiles (see WgradCfg): fast_conv1 2.46 -> 2.36 ms.  With t_out = 2 a tile is ONE stage
  // and every stage starts with the tile-boundary refill of the ring: slow_conv2 0.27 -> 0.30 ms, left on the 32x32x16 form
  if (!f32 && d->taps == 9 && (p->cfg == 0 || p->cfg == 1) && p->t_out % 2 == 0 && p->t_out >= 4) { p->FPR = 1; p->TH = 6; }
  // every other bf16 3x3 configuration of 8 rows per wave: row-paired 16x16x32 (see WgradCfg)
  else if (!f32 && d->taps == 9 && (p->cfg == 0 || p->cfg == 1 || (p->cfg == 2 && p->DG == 4 && p->TH == 16))) p->FPR = 2;
#ifdef SFVOS_DIAG
  if (getenv("SFVOS_WGRAD_M32") && p->FPR) { p->TH = p->FPR == 1 ? 8 : p->TH; p->FPR = 0; }   // A/B: the 32x32x16 configurations
  if (getenv("SFVOS_WGRAD_RPR") && p->FPR == 1) { p->TH = 8; p->FPR = 2; }                     // A/B: row pairs where frame pairs apply
  if (p->cfg == 2 && !f32 && p->DG == 4 && getenv("SFVOS_WGRAD_TH8")) p->TH = 8;   // A/B
#endif
  if (p->cfg == 2 && !f32 && p->t_out == 1 && d->kt > 1) {
    // fast_conv3 (12 frames -> 1): nothing is re-used between stages, every stage needs DG new x frames.  With the
    // two-tile look-ahead of the other configurations each stage waited a memory round trip for the other six.
    p->cfg = 4; p->TH = 4; p->DG = 8; p->FPR = 0;
  }
  SFVOS_REQUIRE(d->pyr.n_levels >= 1 && d->pyr.n_levels <= SFVOS_MAX_LEVELS, "wgrad: n_levels out of range");
  SFVOS_REQUIRE(d->batch >= 1 && d->t_alloc >= 1 && d->t_offset > -(1 << 20) && d->t_offset < (1 << 20) &&
                d->t_in < (1 << 20), "wgrad: bad x window");  // frames of the window outside [0, t_alloc) are zeros
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
  // Pixel split (split-K): N = col_blocks * psplit workgroups, one per compute unit at a time (LDS), dealt evenly over
  // the XCDs (see the kernel).  A split costs one fp32 slab written and re-read; too few splits leave units idle in the
  // last round.  Candidates: the split counts that fill k = 1..4 whole rounds of the chip as closely as possible.
  const int cus = device_cu_count() > 0 ? device_cu_count() : 256;
  const double t_mma = 2.0 * d->c_in * d->c_out * d->kt * d->taps * (double)p->t_out * px * d->batch / 1.0e15;
  const double t_slab = 2.0 * 4.0 * d->c_out * d->c_in * d->kt * d->taps / 3.0e12;  // per split: write + read
  int ps = 1;
  double best = 1e30;
  for (int k = 1; k <= 4; ++k) {
    int m = (cus * k) / col_blocks;
    if (m < 1) m = 1;
    if (m > p->ntiles) m = p->ntiles;
    const int per_m = ceil_div(p->ntiles, m);          // tiles per workgroup
    const int m_eff = ceil_div(p->ntiles, per_m);      // splits that are not empty
    const int rounds = ceil_div(m_eff * col_blocks, cus);
    // a round lasts as long as a workgroup: per_m tiles of the ntiles, on 1/col_blocks of the output
    const double cost = t_mma * (double)rounds * cus * per_m / ((double)p->ntiles * col_blocks) + m_eff * t_slab;
    if (cost < best) { best = cost; ps = m_eff; }
  }
#ifdef SFVOS_DIAG  // tuning aid of diagnostic builds only
  if (const char* ov = getenv("SFVOS_WGRAD_SPLIT")) ps = atoi(ov) > 0 ? atoi(ov) : ps;
#endif
  if (ps > p->ntiles) ps = p->ntiles;
  if (ps < 1) ps = 1;
  const int per = ceil_div(p->ntiles, ps);  // no empty splits
  p->psplit = ceil_div(p->ntiles, per);
  return SFVOS_OK;
}

template <int DT, int TAPS, int NTN, int NTC, int DG, int TH, int R, int NXS = 2, int KS = 1, int FPR = 0>
static int launch_wgrad(const WgradArgs& a, long long grid, hipStream_t stream) {
  typedef WgradCfg<DT, TAPS, NTN, NTC, DG, TH, R, NXS, KS, FPR> C;
  auto kern = wgrad_kernel<DT, TAPS, NTN, NTC, DG, TH, R, NXS, KS, FPR>;
  static LdsAttrOnce once;
  if (int rc = once.ensure((const void*)kern, C::LDS_BYTES, "wgrad")) return rc;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), C::LDS_BYTES, stream, a);
  return check_launch("wgrad");
}

}  // namespace sfvos

using namespace sfvos;

// The lateral k x 1 x 1 convs (32 -> 64 channels, bf16) have a kernel of their own (lateral_wgrad.hip: all kt taps per
// workgroup, x and dy read once), and so has the kt x 3 x 3 conv 32 -> 32 with one output frame (wgrad_t1.hip).  The workspace is sized for whichever of the two kernels needs more, so the generic one
// stays available as the fallback.
static bool lateral_wgrad_wanted(const sfvos_conv_desc* d) {
#ifdef SFVOS_DIAG
  if (getenv("SFVOS_NO_LATERAL_KERNEL")) return false;
#endif
  return d != nullptr && d->struct_size == (int)sizeof(sfvos_conv_desc);
}

extern "C" size_t sfvos_conv3d_wgrad_workspace_bytes(const sfvos_conv_desc* d) {
  WgradPlan p;
  if (make_wgrad_plan(d, &p) != SFVOS_OK) return 0;
  size_t n = (size_t)p.psplit * d->c_out * d->c_in * d->kt * d->taps * sizeof(float);
  if (lateral_wgrad_wanted(d)) {
    size_t m = lateral_wgrad_workspace_bytes(d);
    if (m > n) n = m;
    m = wgrad_t1_workspace_bytes(d);
    if (m > n) n = m;
  }
  return n;
}

extern "C" int sfvos_conv3d_wgrad(const sfvos_conv_desc* d, const void* x, const void* dy, float* grad_w,
                                  int accumulate, void* workspace, sfvos_stream_t stream) {
  WgradPlan p;
  int rc = make_wgrad_plan(d, &p);
  if (rc != SFVOS_OK) return rc;
  SFVOS_REQUIRE(x && dy && grad_w && workspace, "wgrad: null pointer");
  if (lateral_wgrad_wanted(d)) {
    rc = lateral_wgrad_try(d, x, dy, grad_w, accumulate, workspace, (hipStream_t)stream);
    if (rc >= 0) return rc;
    bool t1_ok = true;
#ifdef SFVOS_DIAG
    t1_ok = !getenv("SFVOS_NO_T1_KERNEL");
#endif
    if (t1_ok) {
      rc = wgrad_t1_try(d, x, dy, grad_w, accumulate, workspace, (hipStream_t)stream);   // fast_conv3: one tap per workgroup
      if (rc >= 0) return rc;
    }
  }
  WgradArgs a;
  a.x = (const char*)x; a.dy = (const char*)dy; a.slab = (float*)workspace;
  a.t_in = d->t_in; a.t_alloc = d->t_alloc; a.t_offset = d->t_offset; a.t_out = p.t_out; a.c_in = d->c_in;
  a.c_out = d->c_out; a.kt = d->kt;
  a.ld_x = d->ld_x; a.ld_y = d->ld_y; a.batch = d->batch;
  a.x_group_bytes = d->x_group_stride * 2;
  a.n_blocks = p.n_blocks; a.c_blocks = p.c_blocks;
  a.dt_blocks = p.dt_blocks; a.psplit = p.psplit;
  a.ntiles = p.ntiles; a.lv = p.lv;
  // padded to a multiple of 8 (XCD-aware order in the kernel; padding workgroups exit at once)
  const long long grid = ceil_div64((long long)p.psplit * p.n_blocks * p.c_blocks * p.dt_blocks, 8) * 8;
  hipStream_t s = (hipStream_t)stream;
  const bool bf = d->dtype == SFVOS_BF16;
  switch (p.cfg) {
    // last template argument: x ring slots (R >= 2 DG lets the ring prefetch across pixel-tile boundaries)
    case 0: rc = !bf ? launch_wgrad<SFVOS_F32, 9, 1, 2, 4, 4, 5>(a, grid, s)
                     : p.FPR == 1 ? launch_wgrad<SFVOS_BF16, 9, 1, 2, 4, 6, 7, 2, 1, 1>(a, grid, s)
                     : p.FPR == 2 ? launch_wgrad<SFVOS_BF16, 9, 1, 2, 4, 8, 5, 2, 1, 2>(a, grid, s)
                                  : launch_wgrad<SFVOS_BF16, 9, 1, 2, 4, 8, 5>(a, grid, s); break;
    case 1: rc = !bf ? launch_wgrad<SFVOS_F32, 9, 2, 2, 2, 4, 4>(a, grid, s)
                     : p.FPR == 1 ? launch_wgrad<SFVOS_BF16, 9, 2, 2, 2, 6, 5, 2, 1, 1>(a, grid, s)
                     : p.FPR == 2 ? launch_wgrad<SFVOS_BF16, 9, 2, 2, 2, 8, 5, 2, 1, 2>(a, grid, s)
                                  : launch_wgrad<SFVOS_BF16, 9, 2, 2, 2, 8, 5>(a, grid, s); break;
    case 2: rc = !bf ? launch_wgrad<SFVOS_F32, 9, 1, 1, 8, 4, 10>(a, grid, s)
                     : p.DG == 4 && p.TH == 16 && p.FPR == 2 ? launch_wgrad<SFVOS_BF16, 9, 1, 1, 4, 16, 6, 2, 2, 2>(a, grid, s)
                     : p.DG == 4 && p.TH == 16 ? launch_wgrad<SFVOS_BF16, 9, 1, 1, 4, 16, 6, 2, 2>(a, grid, s)
                     : p.DG == 4 ? launch_wgrad<SFVOS_BF16, 9, 1, 1, 4, 8, 7, 2, 2>(a, grid, s)
                                 : launch_wgrad<SFVOS_BF16, 9, 1, 1, 8, 8, 11>(a, grid, s); break;
    case 4: rc = launch_wgrad<SFVOS_BF16, 9, 1, 1, 8, 4, 16, 8>(a, grid, s); break;
    default: rc = bf ? launch_wgrad<SFVOS_BF16, 1, 2, 1, 4, 8, 8>(a, grid, s) : launch_wgrad<SFVOS_F32, 1, 2, 1, 4, 4, 8>(a, grid, s); break;
  }
  if (rc != SFVOS_OK) return rc;
  return launch_wgrad_reduce((const float*)workspace, p.psplit, d->c_out, d->c_in, d->kt, d->taps, grad_w, accumulate, s);
}
