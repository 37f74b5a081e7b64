// lateral_wgrad.hip -- weight gradient of the lateral time-strided convolution conv_f2s* (k x 1 x 1, 32 -> 64 channels,
// reference code/helpers/model.py:82-94,112; aten::convolution_backward grad_weight), bf16 operands:
//
//   dW[n][c][dt] = sum_{b, f < t_out, px} dy[b][f][px][n] * x[b][f + dt][px][c]            dt < kt = t_in - t_out + 1
//
// 21 GFLOP against 0.15 GB per clip at (sp,fp) = (4,32): HBM-bound.  The generic weight-gradient kernel (wgrad.hip) gives a
// workgroup 4 temporal taps, so dy is read kt/4 times and x 1.4 times; here ONE workgroup holds ALL kt taps (kt x 64 x 32
// fp32 accumulators spread over its 8 waves: wave w owns the 32-channel half n = w & 1 of taps [(w >> 1) NTW, + NTW)) and every
// byte of x and dy is read exactly once:
//
//   stage   = one tile of 16 consecutive positions of a (level, clip): its t_in x runs (16 px x 64 B = one 1-KB LDS-DMA piece
//             each) and its 2 t_out dy half-runs (16 px x 64 B of one 32-channel half) land in one slot of an LDS ring.
//             The reduction runs over pixels, so both MFMA operands are k-major in NDHWC: the slot keeps [pixel][32 ch] rows
//             and the fragments are read with ds_read_b64_tr_b16 (as in wgrad.hip).
//   ring    = R slots, R - 1 stages in flight: the copies of stage s + R - 1 are issued as soon as the barrier of stage s
//             has released the slot of stage s - 1, and the wait in front of a stage is a COUNTED s_waitcnt vmcnt(n) that
//             leaves the R - 2 younger stages in flight (a 1 x 1 conv has no halo: nothing is re-used between tiles, so
//             what matters is bytes in flight -- 80-140 KB per compute unit here -- not re-use).
//   compute = per wave and 16-position step NTW * t_out MFMAs (v_mfma_f32_32x32x16_bf16, K = the 16 pixels) on t_out dy
//             fragments and NTW + t_out - 1 x fragments.  (Round 3 first split the waves by tap only -- 8 tap blocks x
//             both channel halves: 24 tap slots for 20 taps, 16 for 11, twice the dy fragments per wave -- and was bound by
//             this part, not by memory: 28 us of fragment reads + MFMAs against 20 us of copies for conv_f2s1.)
//   output  = one fp32 slab [64][kt][32] per workgroup, summed in fixed order by wgrad_reduce (deterministic, no atomics).
#include <stdlib.h>

#include "common.h"

#ifdef SFVOS_DIAG
#define SFVOS_LWG_DBG(bit) ((a.debug & (bit)) != 0)
#else
#define SFVOS_LWG_DBG(bit) false
#endif

namespace sfvos {

typedef __attribute__((ext_vector_type(2))) unsigned int lw_u32x2;

struct LatWgArgs {
  const char* x;
  const char* dy;
  float* slab;
  int t_in, t_alloc, t_offset, t_out, kt, ld_x, ld_y, batch;
  int ntiles, per;  // 16 SUB-position tiles over all levels and clips; tiles per pixel share
  int nshare;       // pixel shares (= slabs): every share is swept by `groups` workgroups, one per tap group
  int groups, dtg;  // tap groups and taps per group: group g owns taps [g dtg, min(kt, (g+1) dtg)) and copies only the
                    // x frames they meet (dtg + t_out - 1 of the t_in) -- smaller slabs (kt x 8 KB per SHARE, written in
                    // pieces by its groups) for somewhat more input traffic, most of it hits in the XCD's L2
  int ring;         // R: slots of the LDS ring
  int debug;        // diagnostic builds only (timing, results wrong): 1 no fragment reads / MFMAs, 2 no copies, 4 no slabs
  int n;            // levels
  int HW[SFVOS_MAX_LEVELS];
  int tile_begin[SFVOS_MAX_LEVELS + 1];
  long long xpos[SFVOS_MAX_LEVELS], ypos[SFVOS_MAX_LEVELS];
};

constexpr int LWG_NW = 8;        // waves per workgroup
constexpr int LWG_MAX_NTW = 8;   // taps per wave (x one channel half): 128 accumulator registers; 4 tap blocks -> 32 taps per group
constexpr int LWG_MAX_RING = 6;

// s_waitcnt vmcnt(n) for a wave-uniform runtime n (the instruction takes an immediate)
__device__ __forceinline__ void wait_vmcnt(int n) {
#define SFVOS_VMW(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
  switch (n) {
    SFVOS_VMW(0) SFVOS_VMW(1) SFVOS_VMW(2) SFVOS_VMW(3) SFVOS_VMW(4) SFVOS_VMW(5) SFVOS_VMW(6) SFVOS_VMW(7)
    SFVOS_VMW(8) SFVOS_VMW(9) SFVOS_VMW(10) SFVOS_VMW(11) SFVOS_VMW(12) SFVOS_VMW(13) SFVOS_VMW(14) SFVOS_VMW(15)
    SFVOS_VMW(16) SFVOS_VMW(17) SFVOS_VMW(18) SFVOS_VMW(19) SFVOS_VMW(20) SFVOS_VMW(21) SFVOS_VMW(22) SFVOS_VMW(23)
    SFVOS_VMW(24) SFVOS_VMW(25) SFVOS_VMW(26) SFVOS_VMW(27) SFVOS_VMW(28) SFVOS_VMW(29) SFVOS_VMW(30) SFVOS_VMW(31)
    SFVOS_VMW(32) SFVOS_VMW(33) SFVOS_VMW(34) SFVOS_VMW(35) SFVOS_VMW(36) SFVOS_VMW(37) SFVOS_VMW(38) SFVOS_VMW(39)
    SFVOS_VMW(40) SFVOS_VMW(41) SFVOS_VMW(42) SFVOS_VMW(43) SFVOS_VMW(44) SFVOS_VMW(45) SFVOS_VMW(46) SFVOS_VMW(47)
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;   // more than the counter holds: drain
  }
#undef SFVOS_VMW
}

// SUB: 16-position sub-tiles per stage (a stage copies SUB KB contiguous bytes per x frame: longer DRAM bursts)
template <int NTW, int TOUT, int SUB>
__global__ __launch_bounds__(64 * LWG_NW) void lateral_wgrad_kernel(LatWgArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  // workgroup -> (pixel share, tap group): the groups of one share get ids that are equal mod 8 and adjacent, i.e. they
  // run at the same time on one XCD (speed only: the frames and dy tiles they share then hit in its L2)
  const int chunk = blockIdx.x / (8 * a.groups), within = blockIdx.x - chunk * (8 * a.groups);
  const int grp = within >> 3, share = chunk * 8 + (within & 7);
  if (share >= a.nshare) return;   // padding workgroup (before any barrier)
  const int f0 = grp * a.dtg;                           // first tap = first x frame of this group
  const int kt_g = min(a.dtg, a.kt - f0);               // its taps
  const int nf = kt_g + TOUT - 1;                       // x frames it needs: f0 .. f0 + nf - 1
  const int units = nf + 2 * TOUT;                      // x frames, then (f, n half) of dy
  const int pieces = units * SUB;                       // 1-KB pieces per stage: piece = unit * SUB + sub-tile
  const int slot_bytes = (a.dtg + 3 * TOUT - 1) * SUB * 1024;   // the same for every group (the largest)
  const int npw = (pieces - wv + LWG_NW - 1) / LWG_NW;  // pieces this wave copies per stage: wv, wv + 8, ...
  const int R = a.ring;
  const int tile_begin = share * a.per;
  const int S = max(0, min(a.ntiles, tile_begin + a.per) - tile_begin);   // stages of this workgroup

  // lane part of a copy: pixel lane / 4 of the tile, 16-byte chunk lane % 4 of its 64-byte run
  const unsigned xo = (unsigned)((lane >> 2) * a.ld_x * 2 + (lane & 3) * 16);
  const unsigned yo = (unsigned)((lane >> 2) * a.ld_y * 2 + (lane & 3) * 16);

  // Copy cursor: the stages are issued in order, so the geometry of the next tile is carried in scalars and advanced by
  // 16 positions per stage; only when a tile starts a new (level, clip) plane is it looked up in the level tables (the
  // dynamically indexed kernel arguments cost a scalar-memory round trip each: per piece and stage they were a third of
  // the kernel's time).
  int cur_tile = tile_begin;   // next tile to issue
  int left = 0;                // positions from that tile's first one to the end of its plane (<= 0: look the plane up)
  long long xb = 0, yb = 0;    // byte offsets of the tile in frame t_offset of x / frame 0 of dy
  long long xfs = 0, yfs = 0;  // bytes between consecutive frames of the plane
  auto enter = [&](int tile) {
    int lvl = 0;
#pragma unroll
    for (int l = 1; l < SFVOS_MAX_LEVELS; ++l)
      if (l < a.n && tile >= a.tile_begin[l]) lvl = l;
    const int HW = a.HW[lvl];
    const int per_clip = (HW + 16 * SUB - 1) / (16 * SUB);
    const int k = tile - a.tile_begin[lvl];
    const int b = k / per_clip, px0 = (k - b * per_clip) * 16 * SUB;
    left = HW - px0;
    xfs = (long long)HW * a.ld_x * 2;
    yfs = (long long)HW * a.ld_y * 2;
    xb = (a.xpos[lvl] + ((long long)b * a.t_alloc + a.t_offset + f0) * HW + px0) * a.ld_x * 2;
    yb = (a.ypos[lvl] + (long long)b * TOUT * HW + px0) * a.ld_y * 2;
  };
  auto issue = [&](int s) {   // every piece of stage s (= tile cur_tile) that belongs to this wave -> slot s % R
    if (left <= 0) enter(cur_tile);
    if (SFVOS_LWG_DBG(2)) { ++cur_tile; left -= 16 * SUB; return; }
    const unsigned dst = (unsigned)((s % R) * slot_bytes);
    const int xrec = left * a.ld_x * 2, yrec = left * a.ld_y * 2;   // the rest of the plane; beyond it: zero-filled
    for (int p = wv; p < pieces; p += LWG_NW) {
      const int u = p / SUB, sub = p - u * SUB;   // lanes of sub-tile `sub` start 16 sub positions into the tile
      if (u < nf) {
        lds_dma16(a.x + xb + u * xfs, xrec, xo + sub * 16 * a.ld_x * 2, dst + p * 1024);
      } else {
        const int q = u - nf, f = q >> 1, nh = q & 1;
        lds_dma16(a.dy + yb + f * yfs + nh * 64, yrec - nh * 64, yo + sub * 16 * a.ld_y * 2, dst + p * 1024);
      }
    }
    ++cur_tile;
    left -= 16 * SUB;
    xb += 16 * SUB * a.ld_x * 2;
    yb += 16 * SUB * a.ld_y * 2;
  };

  f32x16 acc[NTW];
#pragma unroll
  for (int d = 0; d < NTW; ++d)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[d][e] = 0.f;

  // transposed fragment of a [16 px][32 ch] tile (64-byte rows), as in wgrad.hip: lane -> (row q, 4-column group p) of its
  // 16-lane group's 4x16 block
  const int gq = lane >> 4, li = lane & 15, qq = li >> 2, pp = li & 3;
  const int lane_off = (8 * (gq >> 1) + qq) * 64 + (16 * (gq & 1) + 4 * pp) * 2;
  auto frag = [&](const char* tile) {
    const lw_u32x2 lo = __builtin_bit_cast(lw_u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((SFVOS_LDS short4v*)(tile + lane_off)));
    const lw_u32x2 hi = __builtin_bit_cast(lw_u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((SFVOS_LDS short4v*)(tile + lane_off + 4 * 64)));
    u32x4 v = {lo[0], lo[1], hi[0], hi[1]};
    return v;
  };

  const int nh = wv & 1;                   // this wave's 32-channel half of n
  const int dt0 = (wv >> 1) * NTW;         // first tap of this wave inside the group
  const int live = min(NTW, kt_g - dt0);   // taps of this wave that exist (<= 0: the wave only copies)

  for (int s = 0; s < R - 1 && s < S; ++s) issue(s);
  for (int s = 0; s < S; ++s) {
    // stage s has landed once at most the pieces of the min(R - 2, S - 1 - s) younger stages are outstanding
    wait_vmcnt(npw * min(R - 2, S - 1 - s));
    __syncthreads();   // everybody's pieces of stage s are in LDS; everybody is done reading stage s - 1
    if (s + R - 1 < S) issue(s + R - 1);   // into the slot of stage s - 1
    if (live > 0 && !SFVOS_LWG_DBG(1)) {
#pragma unroll
      for (int ks = 0; ks < SUB; ++ks) {   // one K step = 16 positions
        const char* slot = smem + (s % R) * slot_bytes + ks * 1024;
        u32x4 A[TOUT], B[NTW + TOUT - 1];
#pragma unroll
        for (int f = 0; f < TOUT; ++f) A[f] = frag(slot + (nf + 2 * f + nh) * SUB * 1024);
#pragma unroll
        for (int u = 0; u < NTW + TOUT - 1; ++u) B[u] = frag(slot + min(dt0 + u, nf - 1) * SUB * 1024);
#pragma unroll
        for (int d = 0; d < NTW; ++d)
          if (d < live) {   // wave-uniform
#pragma unroll
            for (int f = 0; f < TOUT; ++f) Mma<SFVOS_BF16>::run(acc[d], A[f], B[d + f]);
          }
      }
    }
  }

  // slab[share][n][dt][c]: this group's taps of its share's slab
  const int r = lane & 31, hh = lane >> 5;
#pragma unroll
  for (int d = 0; d < NTW; ++d)
    if (d < live && !SFVOS_LWG_DBG(4)) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int n = nh * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
        a.slab[(((long long)share * 64 + n) * a.kt + f0 + dt0 + d) * 32 + r] = acc[d][e];
      }
    }
}

struct LatWgPlan {
  int groups, dtg, ntw;   // tap groups, taps per group, taps per wave (4 tap blocks x 2 channel halves = 8 waves)
  int sub, ring;          // 16-position sub-tiles per stage, LDS ring slots
  int ntiles, per, nshare;
  long long grid;
  int lds;
};

// Shapes this kernel covers, and how it is launched for them.
static bool lateral_wgrad_plan(const sfvos_conv_desc* d, LatWgPlan* p) {
  const int t_out = d->t_in - d->kt + 1;
  if (!(d->dtype == SFVOS_BF16 && d->taps == 1 && d->c_in == 32 && d->c_out == 64 && d->pad_t == 0 && d->kt >= 1 &&
        t_out >= 1 && t_out <= 3 && d->x_group_stride == 0 && d->x_frame_stride == 0 &&
        d->y_frame_stride == 0 && d->t_offset >= 0 && d->t_offset + d->t_in <= d->t_alloc && d->ld_x % 8 == 0 &&
        d->ld_x >= 32 && d->ld_y % 8 == 0 && d->ld_y >= 64 && d->batch >= 1 && d->pyr.n_levels >= 1 &&
        d->pyr.n_levels <= SFVOS_MAX_LEVELS))
    return false;
  for (int l = 0; l < d->pyr.n_levels; ++l) {
    if (d->pyr.h[l] < 1 || d->pyr.w[l] < 1) return false;
    if ((long long)d->pyr.h[l] * d->pyr.w[l] * (d->ld_x > d->ld_y ? d->ld_x : d->ld_y) * 2 >= (1ll << 31)) return false;
  }
  const int cus = device_cu_count() > 0 ? device_cu_count() : 256;
  // Tap groups.  Measured inside the training step (conv_f2s1, kt = 20 / conv_f2s2, kt = 11; 1, 2, 4 groups): 52.7 / 48.6 /
  // 56.6 us and 32.6 / 34.4 / 42.2 us -- with many taps the halved slabs (kt x 8 KB per share, written and read again) pay
  // for the re-read frames, with few they do not.  One group up to 15 taps, two beyond; more only where a group would
  // otherwise need more than LWG_MAX_NTW taps per wave.
  int G = 0;
  for (int g = d->kt > 15 ? 2 : 1; g <= 8 && G == 0; g *= 2) {
    const int dtg = ceil_div(d->kt, g);
    if (ceil_div(d->kt, dtg) == g && ceil_div(dtg, LWG_NW / 2) <= LWG_MAX_NTW) G = g;
  }
  if (G == 0) return false;
#ifdef SFVOS_DIAG
  if (const char* ov = getenv("SFVOS_LWG_GROUPS")) {
    const int v = atoi(ov);
    if (v >= 1 && v <= d->kt && ceil_div(d->kt, ceil_div(d->kt, v)) == v && ceil_div(ceil_div(d->kt, v), LWG_NW / 2) <= LWG_MAX_NTW) G = v;
  }
#endif
  p->groups = G;
  p->dtg = ceil_div(d->kt, G);
  p->ntw = ceil_div(p->dtg, LWG_NW / 2);
  if (p->ntw > LWG_MAX_NTW) return false;
  const int units = p->dtg + 3 * t_out - 1;   // x frames of a group + the dy half-frames
  // 32-position tiles (2-KB bursts per frame) while two slots of them fit the LDS; else 16-position tiles
  int sub = 2 * 2 * units * 1024 <= 160 * 1024 ? 2 : 1;
#ifdef SFVOS_DIAG
  if (const char* ov = getenv("SFVOS_LWG_SUB")) { const int v = atoi(ov); if (v == 1 || (v == 2 && sub == 2)) sub = v; }
#endif
  const int slot = units * 1024 * sub;
  int R = (160 * 1024) / slot;
  if (R > LWG_MAX_RING) R = LWG_MAX_RING;
  if (R < 2) return false;
  // counted waits: (R - 2) stages x at most ceil(pieces / 8) pieces per wave must fit the 6-bit counter
  while (R > 2 && (R - 2) * ceil_div(units * sub, LWG_NW) > 47) --R;
  long long tiles = 0;
  for (int l = 0; l < d->pyr.n_levels; ++l) tiles += (long long)d->batch * ceil_div(d->pyr.h[l] * d->pyr.w[l], 16 * sub);
  if (tiles < 1 || tiles >= (1ll << 30)) return false;
  // shares: workgroups / groups, each with a contiguous run of the tiles; every share costs one slab of kt x 8 KB written
  // and read again, so tiny problems get fewer (at least 8 stages each)
  long long g = tiles / 8 > 0 ? tiles / 8 : 1;
  if (g > cus / G) g = cus / G > 0 ? cus / G : 1;
#ifdef SFVOS_DIAG  // tuning aids of diagnostic builds only
  if (const char* ov = getenv("SFVOS_LWG_WGS")) { const long long v = atoll(ov) / G; if (v >= 1 && v <= tiles) g = v; }
  if (const char* ov = getenv("SFVOS_LWG_RING")) { const int v = atoi(ov); if (v >= 2 && v * slot <= 160 * 1024) R = v; }
#endif
  p->per = (int)ceil_div64(tiles, g);
  p->nshare = (int)ceil_div64(tiles, p->per);
  p->ntiles = (int)tiles;
  p->ring = R;
  p->sub = sub;
  p->grid = ceil_div64(p->nshare, 8) * 8 * G;
  p->lds = R * slot;
  return true;
}

size_t lateral_wgrad_workspace_bytes(const sfvos_conv_desc* d) {
  LatWgPlan p;
  if (!lateral_wgrad_plan(d, &p)) return 0;
  return (size_t)p.nshare * 64 * 32 * d->kt * sizeof(float);
}

// -1: shape not covered (the caller falls back to the generic kernel)
int lateral_wgrad_try(const sfvos_conv_desc* d, const void* x, const void* dy, float* grad_w, int accumulate,
                      void* workspace, hipStream_t stream) {
  LatWgPlan p;
  if (!lateral_wgrad_plan(d, &p)) return -1;
  LatWgArgs a;
  a.x = (const char*)x; a.dy = (const char*)dy; a.slab = (float*)workspace;
  a.t_in = d->t_in; a.t_alloc = d->t_alloc; a.t_offset = d->t_offset; a.t_out = d->t_in - d->kt + 1; a.kt = d->kt;
  a.ld_x = d->ld_x; a.ld_y = d->ld_y; a.batch = d->batch;
  a.debug = 0;
#ifdef SFVOS_DIAG
  if (const char* dbg = getenv("SFVOS_LWG_DEBUG")) a.debug = atoi(dbg);
#endif
  a.ntiles = p.ntiles; a.per = p.per; a.nshare = p.nshare; a.groups = p.groups; a.dtg = p.dtg; a.ring = p.ring;
  a.n = d->pyr.n_levels;
  long long tiles = 0, px = 0;
  for (int l = 0; l < SFVOS_MAX_LEVELS; ++l) {
    const bool live = l < a.n;
    const int HW = live ? d->pyr.h[l] * d->pyr.w[l] : 1;
    a.HW[l] = HW;
    a.tile_begin[l] = (int)tiles;
    a.xpos[l] = (long long)d->batch * a.t_alloc * px;
    a.ypos[l] = (long long)d->batch * a.t_out * px;
    if (live) {
      tiles += (long long)d->batch * ceil_div(HW, 16 * p.sub);
      px += HW;
    }
  }
  a.tile_begin[SFVOS_MAX_LEVELS] = (int)tiles;
#define SFVOS_LWG(NTWv, TOUTv) SFVOS_LWGS(NTWv, TOUTv, 1) SFVOS_LWGS(NTWv, TOUTv, 2)
#define SFVOS_LWGS(NTWv, TOUTv, SUBv)                                                                \
  if (p.ntw == NTWv && a.t_out == TOUTv && p.sub == SUBv) {                                          \
    auto kern = lateral_wgrad_kernel<NTWv, TOUTv, SUBv>;                                             \
    static LdsAttrOnce once;                                                                         \
    if (int rc = once.ensure((const void*)kern, 160 * 1024, "lateral_wgrad")) return rc;             \
    hipLaunchKernelGGL(kern, dim3((unsigned)p.grid), dim3(64 * LWG_NW), p.lds, stream, a);           \
    if (int rc = check_launch("lateral_wgrad")) return rc;                                           \
    return launch_wgrad_reduce((const float*)workspace, p.nshare, 64, 32, d->kt, 1, grad_w, accumulate, stream); \
  }
#define SFVOS_LWG3(NTWv) SFVOS_LWG(NTWv, 1) SFVOS_LWG(NTWv, 2) SFVOS_LWG(NTWv, 3)
  SFVOS_LWG3(1) SFVOS_LWG3(2) SFVOS_LWG3(3) SFVOS_LWG3(4) SFVOS_LWG3(5) SFVOS_LWG3(6) SFVOS_LWG3(7) SFVOS_LWG3(8)
#undef SFVOS_LWG3
#undef SFVOS_LWG
#undef SFVOS_LWGS
  return -1;
}

}  // namespace sfvos
