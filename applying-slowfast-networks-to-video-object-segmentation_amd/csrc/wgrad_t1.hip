// wgrad_t1.hip -- weight gradient of a kt x 3 x 3 convolution 32 -> 32 channels with ONE output frame (fast_conv3,
// reference code/helpers/model.py:62-63,147: 12 frames -> 1 at (sp,fp) = (4,32), 22 -> 1 at (4,64); aten::convolution_backward
// grad_weight), bf16 operands:
//
//   dW[n][c][dt][dh][dw] = sum_{b, h, w} dy[b][h][w][n] * x[b][dt][h + dh - 1][w + dw - 1][c]
//
// 19 GFLOP against 72 MB per clip: HBM-bound.  The generic kernel (wgrad.hip) gives a workgroup 6-8 temporal taps, one per
// wave, and one slab [32][kt][9][32] of fp32 partial sums per PIXEL SPLIT: with the 124 splits it needs to fill the chip,
// 55 MB of slabs are written and read again for 72 MB of input.  Here a workgroup owns ONE temporal tap dt (= one x frame)
// and a share of the pixel tiles, its 8 waves split the tile's ROWS and are summed through LDS at the end, so the splits
// drop to cus / kt (21 for kt = 12: 9 MB of slabs).  The price: the kt workgroups of a share all read its dy tiles -- they
// run at the same time on one XCD (see the workgroup order in the kernel), and all but the first read hits in that XCD's L2.
//
//   stage   = one 16 x 16 pixel tile of a (level, clip) plane: the 18 x 18 halo tile of x frame dt (21 LDS-DMA pieces of
//             1 KB) and the dy tile (16 pieces) land in one slot of an LDS ring; pixels outside the plane are zero-filled by
//             the buffer descriptor's range check (= the conv's spatial zero padding).
//   ring    = 4 slots of 37 KB, 3 stages in flight, counted s_waitcnt vmcnt(n) in front of a stage (as lateral_wgrad.hip).
//   compute = wave w owns tile rows 2w, 2w+1: row walk over its 4 halo rows (as wgrad.hip): the three column-shifted x
//             fragments of a halo row serve the (row, dh) pairs that meet it -- 18 MFMAs (v_mfma_f32_32x32x16_bf16, K = the 16
//             pixels of a row) on 9 accumulator tiles; both operands are k-major in NDHWC -> ds_read_b64_tr_b16.
//   output  = the 8 waves' accumulators are added in wave order through LDS, tap by tap (fixed order: deterministic), and
//             leave as this workgroup's part [32][dt][9][32] of its share's slab; wgrad_reduce_kernel adds the shares.
#include <stdlib.h>

#include "common.h"

#ifdef SFVOS_DIAG   // timing-only switches of diagnostic builds (SFVOS_T1_DEBUG; results wrong)
#define SFVOS_T1_DBG(bit) ((a.debug & (bit)) != 0)
#else
#define SFVOS_T1_DBG(bit) false
#endif

namespace sfvos {

typedef __attribute__((ext_vector_type(2))) unsigned int t1_u32x2;

constexpr int T1_NW = 8;                       // waves per workgroup
constexpr int T1_TH = 16, T1_TW = 16;          // pixel tile
constexpr int T1_HC = T1_TW + 2;               // halo tile: 18 x 18 positions
constexpr int T1_XPOS = (T1_TH + 2) * T1_HC;   // 324 positions
constexpr int T1_XP = (T1_XPOS + 15) / 16;     // 21 pieces of 16 positions x 64 B
constexpr int T1_DP = T1_TH;                   // 16 dy pieces (one tile row each)
constexpr int T1_PIECES = T1_XP + T1_DP;       // 37
constexpr int T1_SLOT = T1_PIECES * 1024;      // 37 KB
constexpr int T1_R = 4;                        // ring slots
constexpr int T1_NPW = (T1_PIECES + T1_NW - 1) / T1_NW;   // at most 5 pieces per wave and stage
constexpr int T1_LDS = T1_R * T1_SLOT;         // 148 KB (>= the 64 KB of the final tap-by-tap sums)
static_assert(T1_LDS <= 160 * 1024 && T1_LDS >= 2 * T1_NW * 4096, "LDS budget");

struct T1Args {
  const char* x;
  const char* dy;
  float* slab;
  int t_alloc, t_offset, kt, ld_x, ld_y, batch;
  int ntiles, per, nshare;   // 16 x 16 tiles over all levels and clips; tiles per pixel share; shares (= slabs)
  int debug;                 // diagnostic builds only: 1 no fragment reads / MFMAs, 2 no x copies, 4 no dy copies, 8 no sums / slabs
  int n;                     // levels
  int H[SFVOS_MAX_LEVELS], W[SFVOS_MAX_LEVELS], tiles_h[SFVOS_MAX_LEVELS], tiles_w[SFVOS_MAX_LEVELS];
  int tile_begin[SFVOS_MAX_LEVELS + 1];
  long long xpos[SFVOS_MAX_LEVELS], ypos[SFVOS_MAX_LEVELS];
};

// s_waitcnt vmcnt(n) for a wave-uniform runtime n in [0, 10] (two younger stages of at most five pieces)
__device__ __forceinline__ void t1_wait_vmcnt(int n) {
#define SFVOS_VMW(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
  switch (n) {
    SFVOS_VMW(1) SFVOS_VMW(2) SFVOS_VMW(3) SFVOS_VMW(4) SFVOS_VMW(5) SFVOS_VMW(6) SFVOS_VMW(7) SFVOS_VMW(8) SFVOS_VMW(9)
    SFVOS_VMW(10)
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
#undef SFVOS_VMW
}

__global__ __launch_bounds__(64 * T1_NW) void wgrad_t1_kernel(T1Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  // workgroup -> (pixel share, temporal tap).  Workgroup ids are dealt round-robin over the 8 XCDs; XCD x takes the
  // CONTIGUOUS range of q or q + 1 of the N = nshare * kt logical items L = share * kt + dt (as wgrad.hip): every XCD gets
  // the same number of workgroups (+-1) -- with whole shares per XCD 21 shares x 12 taps put 36 workgroups on five XCDs of
  // 32 CUs and none on three -- and the taps of a share still sit on one XCD (two at a range boundary), adjacent in time.
  const int N = a.nshare * a.kt, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nq = N >> 3, nr = N & 7;
  if (slot >= nq + (xcd < nr ? 1 : 0)) return;   // padding workgroup (before any barrier)
  const int L = xcd * nq + min(xcd, nr) + slot;
  const int share = L / a.kt, dt = L - share * a.kt;
  const int tile_begin = share * a.per;
  const int S = max(0, min(a.ntiles, tile_begin + a.per) - tile_begin);   // stages of this workgroup

  // ---- copies.  Wave w copies pieces w, w + 8, ... of a stage: x halo pieces first (16 halo positions each, 4 lanes per
  // position), then the dy pieces (one tile row each).  Per piece the lane's position inside the tile is fixed for the
  // whole kernel; per stage it is shifted by the tile's origin, checked against the plane, and turned into a byte offset.
  int plh[T1_NPW], plw[T1_NPW];
#pragma unroll
  for (int i = 0; i < T1_NPW; ++i) {
    const int p = wv + i * T1_NW;
    if (p < T1_XP) {
      const int hp = p * 16 + (lane >> 2);
      plh[i] = hp < T1_XPOS ? hp / T1_HC - 1 : -(1 << 20);   // the padding positions of the last piece: never valid
      plw[i] = hp % T1_HC - 1;
    } else {
      plh[i] = p - T1_XP;
      plw[i] = lane >> 2;
    }
  }
  const int npw = (T1_PIECES - wv + T1_NW - 1) / T1_NW;   // 5 (waves 0-4) or 4
  const unsigned cj = (unsigned)(lane & 3) * 16;          // 16-byte chunk of the position's 64-byte channel run
  constexpr unsigned OOB = 0x80000000u;

  // Tile cursor: (level, clip, tile row, tile column) of the next stage to issue, advanced by one tile per stage; the level
  // tables are read with compile-time indices only (a dynamically indexed kernel argument is a scalar-memory round trip).
  int lvl = 0, cb = 0, th = 0, tw = 0, H = 1, W = 1, tiles_h = 1, tiles_w = 1;
  long long xpos = 0, ypos = 0;
  auto set_level = [&](int l) {
    lvl = l;
#pragma unroll
    for (int k = 0; k < SFVOS_MAX_LEVELS; ++k)
      if (k == l) { H = a.H[k]; W = a.W[k]; tiles_h = a.tiles_h[k]; tiles_w = a.tiles_w[k]; xpos = a.xpos[k]; ypos = a.ypos[k]; }
  };
  {
    int l0 = 0, begin = 0;
#pragma unroll
    for (int l = 1; l < SFVOS_MAX_LEVELS; ++l)
      if (l < a.n && tile_begin >= a.tile_begin[l]) { l0 = l; begin = a.tile_begin[l]; }
    set_level(l0);
    int k = tile_begin - begin;
    tw = k % tiles_w; k /= tiles_w;
    th = k % tiles_h; cb = k / tiles_h;
  }
  auto issue = [&](int s) {   // every piece of stage s (= the cursor's tile) that belongs to this wave -> slot s % R
    const int HW = H * W;
    const int h0 = th * T1_TH, w0 = tw * T1_TW;
    const char* xf = a.x + (xpos + ((long long)cb * a.t_alloc + a.t_offset + dt) * HW) * a.ld_x * 2;
    const char* yf = a.dy + (ypos + (long long)cb * HW) * a.ld_y * 2;
    const int xrec = HW * a.ld_x * 2, yrec = HW * a.ld_y * 2;
    const unsigned dst = (unsigned)((s % T1_R) * T1_SLOT);
#pragma unroll
    for (int i = 0; i < T1_NPW; ++i) {
      const int p = wv + i * T1_NW;   // wave-uniform
      if (p < T1_PIECES) {
        const int h = h0 + plh[i], w = w0 + plw[i];
        const bool ok = (unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W;
        if (SFVOS_T1_DBG(p < T1_XP ? 2 : 4)) continue;
        if (p < T1_XP)
          lds_dma16(xf, xrec, ok ? (unsigned)((h * W + w) * a.ld_x * 2) + cj : OOB, dst + p * 1024);
        else
          lds_dma16(yf, yrec, ok ? (unsigned)((h * W + w) * a.ld_y * 2) + cj : OOB, dst + p * 1024);
      }
    }
    if (++tw == tiles_w) {
      tw = 0;
      if (++th == tiles_h) {
        th = 0;
        if (++cb == a.batch) { cb = 0; set_level(lvl + 1 < a.n ? lvl + 1 : lvl); }
      }
    }
  };

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

  // transposed fragment of 16 consecutive [position][32 ch] rows (64 bytes each), as in wgrad.hip: lane -> (row q, 4-column
  // group p) of its 16-lane group's 4 x 16 block
  const int gq = lane >> 4, li = lane & 15, qq = li >> 2, pp = li & 3;
  const int lane_off = (8 * (gq >> 1) + qq) * 64 + (16 * (gq & 1) + 4 * pp) * 2;
  auto frag = [&](const char* rows) {
    const t1_u32x2 lo = __builtin_bit_cast(t1_u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((SFVOS_LDS short4v*)(rows + lane_off)));
    const t1_u32x2 hi = __builtin_bit_cast(t1_u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((SFVOS_LDS short4v*)(rows + lane_off + 4 * 64)));
    u32x4 v = {lo[0], lo[1], hi[0], hi[1]};
    return v;
  };

  for (int s = 0; s < T1_R - 1 && s < S; ++s) issue(s);
  for (int s = 0; s < S; ++s) {
    // stage s has landed once at most the pieces of the min(R - 2, S - 1 - s) younger stages are outstanding
    t1_wait_vmcnt(npw * min(T1_R - 2, S - 1 - s));
    __syncthreads();   // everybody's pieces of stage s are in LDS; everybody is done reading stage s - 1
    if (s + T1_R - 1 < S) issue(s + T1_R - 1);   // into the slot of stage s - 1
    if (SFVOS_T1_DBG(1)) continue;
    const char* xt = smem + (s % T1_R) * T1_SLOT + (2 * wv) * T1_HC * 64;   // halo row 2w of the x tile
    const char* yt = smem + (s % T1_R) * T1_SLOT + T1_XP * 1024 + (2 * wv) * 16 * 64;   // tile row 2w of dy
    u32x4 A[2], B[2][3];
    A[0] = frag(yt);
    A[1] = frag(yt + 16 * 64);
#pragma unroll
    for (int dw = 0; dw < 3; ++dw) B[0][dw] = frag(xt + dw * 64);
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      if (rr + 1 < 4) {
#pragma unroll
        for (int dw = 0; dw < 3; ++dw) B[(rr + 1) & 1][dw] = frag(xt + ((rr + 1) * T1_HC + dw) * 64);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int dh = 0; dh < 3; ++dh) {
        const int ty = rr - dh;
        if (ty < 0 || ty > 1) continue;
#pragma unroll
        for (int dw = 0; dw < 3; ++dw) Mma<SFVOS_BF16>::run(acc[dh * 3 + dw], A[ty], B[rr & 1][dw]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // ---- the eight row groups' sums, tap by tap through LDS (two 32-KB buffers): every wave writes its tile of tap t; then
  // wave w adds, in wave order, the eight tiles' elements e = 2w, 2w+1 and stores them into slab[share][n][dt][t][c] (one
  // wave doing a whole tap made the other seven wait at the next barrier: 9 x 128 dependent LDS reads in a row)
  __syncthreads();   // the ring is dead
  if (SFVOS_T1_DBG(8)) return;
  float* red = (float*)smem;
  const int r = lane & 31, hh = lane >> 5;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    float* buf = red + (t & 1) * (T1_NW * 1024);
#pragma unroll
    for (int e = 0; e < 16; ++e) buf[(wv * 16 + e) * 64 + lane] = acc[t][e];
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int e = 2 * wv + u;   // wave-uniform
      float sum = 0.f;
#pragma unroll
      for (int w = 0; w < T1_NW; ++w) sum += buf[(w * 16 + e) * 64 + lane];
      const int n = (e & 3) + 8 * (e >> 2) + 4 * hh;
      a.slab[((((long long)share * 32 + n) * a.kt + dt) * 9 + t) * 32 + r] = sum;
    }
  }
}

struct T1Plan {
  int ntiles, per, nshare;
  long long grid;
};

static bool wgrad_t1_plan(const sfvos_conv_desc* d, T1Plan* p) {
  if (!(d->dtype == SFVOS_BF16 && d->taps == 9 && d->c_in == 32 && d->c_out == 32 && d->pad_t == 0 && d->kt >= 2 &&
        d->t_in == d->kt && d->x_group_stride == 0 && d->x_frame_stride == 0 && d->y_frame_stride == 0 &&
        d->t_offset >= 0 && d->t_offset + d->t_in <= d->t_alloc && d->ld_x % 8 == 0 && d->ld_x >= 32 && d->ld_y % 8 == 0 &&
        d->ld_y >= 32 && d->batch >= 1 && d->pyr.n_levels >= 1 && d->pyr.n_levels <= SFVOS_MAX_LEVELS))
    return false;
  long long tiles = 0;
  for (int l = 0; l < d->pyr.n_levels; ++l) {
    if (d->pyr.h[l] < 1 || d->pyr.w[l] < 1) return false;
    if ((long long)d->pyr.h[l] * d->pyr.w[l] * (d->ld_x > d->ld_y ? d->ld_x : d->ld_y) * 2 >= (1ll << 31)) return false;
    tiles += (long long)d->batch * ceil_div(d->pyr.h[l], T1_TH) * ceil_div(d->pyr.w[l], T1_TW);
  }
  if (tiles < 1 || tiles >= (1ll << 30)) return false;
  const int cus = device_cu_count() > 0 ? device_cu_count() : 256;
  // shares: one round of workgroups (cus / kt shares x kt taps); every share costs one slab written and read again, so tiny
  // problems get fewer (at least 4 stages each)
  long long g = tiles / 4 > 0 ? tiles / 4 : 1;
  if (g > cus / d->kt) g = cus / d->kt > 0 ? cus / d->kt : 1;
#ifdef SFVOS_DIAG  // tuning aid of diagnostic builds only
  if (const char* ov = getenv("SFVOS_T1_SHARES")) { const long long v = atoll(ov); if (v >= 1 && v <= tiles) g = v; }
#endif
  p->per = (int)ceil_div64(tiles, g);
  p->nshare = (int)ceil_div64(tiles, p->per);
  p->ntiles = (int)tiles;
  p->grid = ceil_div64((long long)p->nshare * d->kt, 8) * 8;
  return true;
}

size_t wgrad_t1_workspace_bytes(const sfvos_conv_desc* d) {
  T1Plan p;
  if (!wgrad_t1_plan(d, &p)) return 0;
  return (size_t)p.nshare * 32 * 32 * d->kt * 9 * sizeof(float);
}

// -1: shape not covered (the caller falls back to the generic kernel)
int wgrad_t1_try(const sfvos_conv_desc* d, const void* x, const void* dy, float* grad_w, int accumulate, void* workspace,
                 hipStream_t stream) {
  T1Plan p;
  if (!wgrad_t1_plan(d, &p)) return -1;
  T1Args a;
  a.x = (const char*)x; a.dy = (const char*)dy; a.slab = (float*)workspace;
  a.t_alloc = d->t_alloc; a.t_offset = d->t_offset; a.kt = d->kt; a.ld_x = d->ld_x; a.ld_y = d->ld_y; a.batch = d->batch;
  a.ntiles = p.ntiles; a.per = p.per; a.nshare = p.nshare;
  a.debug = 0;
#ifdef SFVOS_DIAG
  if (const char* dbg = getenv("SFVOS_T1_DEBUG")) a.debug = atoi(dbg);
#endif
  a.n = d->pyr.n_levels;
  long long tiles = 0, px = 0;
  for (int l = 0; l < SFVOS_MAX_LEVELS; ++l) {
    const bool live = l < a.n;
    const int H = live ? d->pyr.h[l] : 1, W = live ? d->pyr.w[l] : 1;
    a.H[l] = H; a.W[l] = W;
    a.tiles_h[l] = ceil_div(H, T1_TH); a.tiles_w[l] = ceil_div(W, T1_TW);
    a.tile_begin[l] = (int)tiles;
    a.xpos[l] = (long long)d->batch * d->t_alloc * px;
    a.ypos[l] = (long long)d->batch * px;   // t_out = 1
    if (live) {
      tiles += (long long)d->batch * a.tiles_h[l] * a.tiles_w[l];
      px += (long long)H * W;
    }
  }
  a.tile_begin[SFVOS_MAX_LEVELS] = (int)tiles;
  static LdsAttrOnce once;
  if (int rc = once.ensure((const void*)wgrad_t1_kernel, T1_LDS, "wgrad_t1")) return rc;
  hipLaunchKernelGGL(wgrad_t1_kernel, dim3((unsigned)p.grid), dim3(64 * T1_NW), T1_LDS, stream, a);
  if (int rc = check_launch("wgrad_t1")) return rc;
  return launch_wgrad_reduce((const float*)workspace, p.nshare, 32, 32, d->kt, 9, grad_w, accumulate, stream);
}

}  // namespace sfvos
