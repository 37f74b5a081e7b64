// batchnorm.hip -- BatchNorm3d forward/backward (+ fused ReLU) over a whole pyramid buffer in one
// launch (reference model.py:113-114,121-122,125-126,133-134,137-138,145,148 and their autograd:
// aten::native_batch_norm / native_batch_norm_backward / relu_ / threshold_backward).
// Statistics and coefficients are per level (the reference normalises each FPN level's call
// separately); every reduction runs in a fixed order (deterministic, no atomics).
#include "elt_util.h"

namespace sfvos {

struct LevelTab {
  int n;
  long long mb[SFVOS_MAX_LEVELS + 1];   // first position of each level
  int blk_begin[SFVOS_MAX_LEVELS + 1];  // first block / partial row of each level (bwd kernels)
};

struct FinalizeTab {
  int n;
  int row_begin[SFVOS_MAX_LEVELS + 1];
  double count[SFVOS_MAX_LEVELS];
};

constexpr int BNB_THREADS = 256;
constexpr int BNB_POS = 512;      // positions per block iteration
constexpr int BNB_MAX_ROWS = 512;  // per level

static int make_level_tab(const sfvos_levels* lv, LevelTab* t, const char* what) {
  SFVOS_REQUIRE(lv != nullptr && lv->n_levels >= 1 && lv->n_levels <= SFVOS_MAX_LEVELS, "%s: bad level count", what);
  t->n = lv->n_levels;
  long long m = 0, blk = 0;
  for (int l = 0; l <= SFVOS_MAX_LEVELS; ++l) {
    t->mb[l] = m;
    t->blk_begin[l] = (int)blk;
    if (l < lv->n_levels) {
      SFVOS_REQUIRE(lv->m[l] > 0, "%s: level %d has no positions", what, l);
      m += lv->m[l];
      long long r = ceil_div64(lv->m[l], BNB_POS);
      if (r > BNB_MAX_ROWS) r = BNB_MAX_ROWS;
      blk += r;
    }
  }
  return SFVOS_OK;
}

__device__ __forceinline__ int level_of_pos(const LevelTab& t, long long m) {
  int lvl = 0;
#pragma unroll
  for (int l = 1; l < SFVOS_MAX_LEVELS; ++l)
    if (l < t.n && m >= t.mb[l]) lvl = l;
  return lvl;
}

// grid = (C/32, n_levels); block = 32 channels x RL row lanes
__global__ __launch_bounds__(32 * RL) void bn_finalize_kernel(const float* part, FinalizeTab ft, const float* gamma,
                                                              const float* beta, float eps, int C, float* mean,
                                                              float* rstd, float* scale, float* shift,
                                                              float* var_unbiased, int cs) {
  __shared__ double scratch[32 * RL];
  const int l = blockIdx.y;
  const int c_local = threadIdx.x & 31, sub = threadIdx.x >> 5, c = blockIdx.x * 32 + c_local;
  const float* p = part + (long long)ft.row_begin[l] * 2 * C;
  const int rows = ft.row_begin[l + 1] - ft.row_begin[l];
  const double count = ft.count[l];
  const double s1 = column_sum(p, rows, 2 * C, c, sub, scratch, c_local);
  const double s2 = column_sum(p, rows, 2 * C, C + c, sub, scratch, c_local);
  if (sub == 0) {
    const double m = s1 / count;
    double var = s2 / count - m * m;
    if (var < 0.0) var = 0.0;
    const float r = (float)(1.0 / sqrt(var + (double)eps));
    const long long o = (long long)l * cs + c;
    mean[o] = (float)m;
    rstd[o] = r;
    const float sc = gamma[c] * r;
    scale[o] = sc;
    shift[o] = beta[c] - (float)m * sc;
    var_unbiased[o] = (float)(count > 1.0 ? var * count / (count - 1.0) : var);
  }
}

__global__ void bn_eval_coeffs_kernel(const float* gamma, const float* beta, const float* rm, const float* rv,
                                      float eps, int C, float* mean, float* rstd, float* scale, float* shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) {
    const float r = 1.f / sqrtf(rv[c] + eps);
    const float sc = gamma[c] * r;
    mean[c] = rm[c];
    rstd[c] = r;
    scale[c] = sc;
    shift[c] = beta[c] - rm[c] * sc;
  }
}

__global__ void bn_running_update_kernel(float* rm, float* rv, const float* means, const float* vars, int n, int cs,
                                         int C, float momentum, long long* num_batches_tracked) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c == 0 && num_batches_tracked) *num_batches_tracked += n;  // one forward per level (model.py:156-159)
  if (c < C) {
    float m = rm[c], v = rv[c];
    for (int i = 0; i < n; ++i) {
      m = (1.f - momentum) * m + momentum * means[(long long)i * cs + c];
      v = (1.f - momentum) * v + momentum * vars[(long long)i * cs + c];
    }
    rm[c] = m;
    rv[c] = v;
  }
}

// the running-statistics update of bn_running_update_kernel as a prologue of the first workgroup of a BN-apply launch
struct RunUpd {
  float* rm; float* rv; const float* means; const float* vars; long long* nbt; int n; float momentum;
};

__device__ __forceinline__ void running_update_block0(const RunUpd& ru, int C, int cs) {
  if (blockIdx.x != 0 || ru.rm == nullptr) return;
  if (threadIdx.x == 0 && ru.nbt) *ru.nbt += ru.n;  // one forward per level (model.py:156-159)
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float m = ru.rm[c], v = ru.rv[c];
    for (int i = 0; i < ru.n; ++i) {
      m = (1.f - ru.momentum) * m + ru.momentum * ru.means[(long long)i * cs + c];
      v = (1.f - ru.momentum) * v + ru.momentum * ru.vars[(long long)i * cs + c];
    }
    ru.rm[c] = m;
    ru.rv[c] = v;
  }
}

static int make_run_upd(const sfvos_bn_running* r, RunUpd* ru, const char* what) {
  ru->rm = nullptr; ru->rv = nullptr; ru->means = nullptr; ru->vars = nullptr; ru->nbt = nullptr; ru->n = 0;
  ru->momentum = 0.f;
  if (r == nullptr) return SFVOS_OK;
  SFVOS_REQUIRE(r->running_mean && r->running_var && r->means && r->vars_unbiased && r->n_updates >= 0,
                "%s: bad sfvos_bn_running", what);
  ru->rm = r->running_mean; ru->rv = r->running_var; ru->means = r->means; ru->vars = r->vars_unbiased;
  ru->nbt = (long long*)r->num_batches_tracked; ru->n = r->n_updates; ru->momentum = r->momentum;
  return SFVOS_OK;
}

template <int DT>
__global__ __launch_bounds__(256) void bn_apply_kernel(const char* __restrict__ x, int ld_x, char* y, int ld_y,
                                                       LevelTab lt, int C, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, int cs, int relu, RunUpd ru) {
  constexpr int CE = Elt<DT>::CE, ES = 16 / CE;
  running_update_block0(ru, C, cs);
  const int cpr = C / CE;
  const long long total = lt.mb[SFVOS_MAX_LEVELS] * cpr;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long m = i / cpr;
    const int c = (int)(i - m * cpr) * CE;
    const long long co = (long long)level_of_pos(lt, m) * cs + c;
    float f[CE];
    unpack<DT>(*(const u32x4*)(x + (m * ld_x + c) * ES), f);
#pragma unroll
    for (int e = 0; e < CE; ++e) {
      float v = f[e] * scale[co + e] + shift[co + e];
      f[e] = relu ? fmaxf(v, 0.f) : v;
    }
    *(u32x4*)(y + (m * ld_y + c) * ES) = pack<DT>(f);
  }
}

// y = sat_e4m3(act(x * scale + shift) * act_scale): the slow pathway's concat buffers as e4m3 operands of the next
// 3x3 conv (BASELINE config 5).  x: bf16 [M][ld_x]; y: bytes [M][ld_y]; 16 channels (one 16-byte chunk) per lane.
__global__ __launch_bounds__(256) void bn_apply_fp8_kernel(const char* __restrict__ x, int ld_x, char* y, int ld_y,
                                                           LevelTab lt, int C, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, int cs, int relu,
                                                           float act_scale, int* sat_count, RunUpd ru) {
  running_update_block0(ru, C, cs);
  const int cpr = C / 16;
  const long long total = lt.mb[SFVOS_MAX_LEVELS] * cpr;
  int sat = 0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long m = i / cpr;
    const int c = (int)(i - m * cpr) * 16;
    const long long co = (long long)level_of_pos(lt, m) * cs + c;
    float f[16];
    unpack<SFVOS_BF16>(*(const u32x4*)(x + (m * ld_x + c) * 2), f);
    unpack<SFVOS_BF16>(*(const u32x4*)(x + (m * ld_x + c + 8) * 2), f + 8);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      float v = f[e] * scale[co + e] + shift[co + e];
      v = (relu ? fmaxf(v, 0.f) : v) * act_scale;
      sat += fabsf(v) > 448.f ? 1 : 0;
      f[e] = v;
    }
    *(u32x4*)(y + m * ld_y + c) = pack<SFVOS_FP8>(f);
  }
  if (sat_count != nullptr) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sat += __shfl_xor(sat, o);
    if ((threadIdx.x & 63) == 0 && sat > 0) atomicAdd(sat_count, sat);
  }
}

// BN backward pass 1 / pass 2 share the thread layout: a block belongs to ONE level and owns a
// strided set of 512-position runs of it; thread = (chunk of CE channels, row lane); per-channel
// partials reduced through LDS in a fixed order -> one deterministic partial row per block.
template <int DT, bool APPLY>
__global__ __launch_bounds__(BNB_THREADS) void bn_bwd_kernel(
    const char* __restrict__ dy, int ld_dy, const char* __restrict__ x, int ld_x, char* dx, int ld_dx, LevelTab lt,
    int C, const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ mean,
    const float* __restrict__ rstd, int cs, int relu, const float* __restrict__ cA, const float* __restrict__ cB,
    const float* __restrict__ cK, float* part, const float* __restrict__ sum_dz, const float* __restrict__ sum_dzx,
    float* dgamma, float* dbeta, int accumulate) {
  constexpr int CE = Elt<DT>::CE, ES = 16 / CE;
  int lvl = 0;
#pragma unroll
  for (int l = 1; l < SFVOS_MAX_LEVELS; ++l)
    if (l < lt.n && (int)blockIdx.x >= lt.blk_begin[l]) lvl = l;
  const int bi = blockIdx.x - lt.blk_begin[lvl], nblk = lt.blk_begin[lvl + 1] - lt.blk_begin[lvl];
  const long long m_begin = lt.mb[lvl], m_end = lt.mb[lvl + 1];
  const int cpr = C / CE;                    // chunks per position (<= 64)
  const int rl = BNB_THREADS / cpr;          // row lanes per block
  const int ch = threadIdx.x % cpr, rowl = threadIdx.x / cpr;
  const int c = ch * CE;
  const long long co = (long long)lvl * cs + c;
  float a0[CE], a1[CE];
#pragma unroll
  for (int e = 0; e < CE; ++e) a0[e] = a1[e] = 0.f;
  float sc[CE], sh[CE], p0[CE], p1[CE], p2[CE];
#pragma unroll
  for (int e = 0; e < CE; ++e) {
    sc[e] = scale[co + e];
    sh[e] = shift[co + e];
    if (APPLY) { p0[e] = cA[co + e]; p1[e] = cB[co + e]; p2[e] = cK[co + e]; }
    else { p0[e] = mean[co + e]; p1[e] = rstd[co + e]; p2[e] = 0.f; }
  }
  if (rowl < rl) {
    for (long long pb = m_begin + (long long)bi * BNB_POS; pb < m_end; pb += (long long)nblk * BNB_POS) {
      const long long pend = pb + BNB_POS < m_end ? pb + BNB_POS : m_end;
      // UN positions per trip: all their loads are issued before the first use (memory-level parallelism;
      // the per-channel sums still add the positions in ascending order)
      constexpr int UN = 4;
      for (long long m0 = pb + rowl; m0 < pend; m0 += (long long)UN * rl) {
        u32x4 rdy[UN], rx[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
          const long long m = m0 + (long long)u * rl;
          if (m < pend) {
            rdy[u] = *(const u32x4*)(dy + (m * ld_dy + c) * ES);
            rx[u] = *(const u32x4*)(x + (m * ld_x + c) * ES);
          }
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
          const long long m = m0 + (long long)u * rl;
          if (m >= pend) break;
          float fdy[CE], fx[CE];
          unpack<DT>(rdy[u], fdy);
          unpack<DT>(rx[u], fx);
          float out[CE];
#pragma unroll
          for (int e = 0; e < CE; ++e) {
            const float dz = (relu && !(fx[e] * sc[e] + sh[e] > 0.f)) ? 0.f : fdy[e];
            if (APPLY) {
              const float d = p0[e] * dz + p1[e] * fx[e] + p2[e];
              out[e] = d;
              a0[e] += d;
            } else {
              a0[e] += dz;
              a1[e] += dz * ((fx[e] - p0[e]) * p1[e]);
            }
          }
          if (APPLY) *(u32x4*)(dx + (m * ld_dx + c) * ES) = pack<DT>(out);
        }
      }
    }
  }
  if (APPLY && blockIdx.x == 0 && (dgamma || dbeta)) {
    // dgamma / dbeta: the per-level sums of pass 1 (bn_bwd_finalize), added in level order
    for (int cc = threadIdx.x; cc < C; cc += BNB_THREADS) {
      double tg = 0.0, tb = 0.0;
      for (int l = 0; l < lt.n; ++l) {
        tg += (double)sum_dzx[(long long)l * cs + cc];
        tb += (double)sum_dz[(long long)l * cs + cc];
      }
      if (dgamma) dgamma[cc] = (accumulate ? dgamma[cc] : 0.f) + (float)tg;
      if (dbeta) dbeta[cc] = (accumulate ? dbeta[cc] : 0.f) + (float)tb;
    }
  }
  if (part == nullptr) return;
  __shared__ float red[2][BNB_THREADS][8];
#pragma unroll
  for (int e = 0; e < CE; ++e) {
    red[0][threadIdx.x][e] = a0[e];
    red[1][threadIdx.x][e] = a1[e];
  }
  __syncthreads();
  const int nsum = APPLY ? 1 : 2;  // thread -> one channel (and one of the two sums)
  for (int i = threadIdx.x; i < nsum * C; i += BNB_THREADS) {
    const int which = i / C, cc = i - which * C;
    const int chunk = cc / CE, e = cc - chunk * CE;
    float s = 0.f;
    for (int k = 0; k < rl; ++k) s += red[which][k * cpr + chunk][e];
    part[((long long)blockIdx.x * nsum + which) * C + cc] = s;
  }
}

// Per level: the sums of pass 1 reduced in fixed order, and the pass-2 coefficients.  grid = (C/32, n_levels);
// block = 32 channels x RL row lanes.  (dgamma / dbeta = the per-level sums added over the levels: done by block 0 of
// the pass-2 kernel, so that this tiny kernel runs all levels in parallel instead of one after the other.)
__global__ __launch_bounds__(32 * RL) void bn_bwd_finalize_kernel(const float* part, LevelTab lt, const float* gamma,
                                                                  const float* mean, const float* rstd, int cs, int C,
                                                                  int train, float* cA, float* cB, float* cK,
                                                                  float* sum_dz, float* sum_dzx) {
  __shared__ double scratch[32 * RL];
  const int l = blockIdx.y;
  const int c_local = threadIdx.x & 31, sub = threadIdx.x >> 5, c = blockIdx.x * 32 + c_local;
  const float* p = part + (long long)lt.blk_begin[l] * 2 * C;
  const int rows = lt.blk_begin[l + 1] - lt.blk_begin[l];
  const double count = (double)(lt.mb[l + 1] - lt.mb[l]);
  const double sdz = column_sum(p, rows, 2 * C, c, sub, scratch, c_local);
  const double sdzx = column_sum(p, rows, 2 * C, C + c, sub, scratch, c_local);
  if (sub == 0) {
    const long long o = (long long)l * cs + c;
    sum_dz[o] = (float)sdz;
    sum_dzx[o] = (float)sdzx;
    const double g = (double)gamma[c] * (double)rstd[o];
    if (train) {
      const double m1 = sdz / count, m2 = sdzx / count;
      cA[o] = (float)g;
      cB[o] = (float)(-g * m2 * (double)rstd[o]);
      cK[o] = (float)(g * m2 * (double)rstd[o] * (double)mean[o] - g * m1);
    } else {
      cA[o] = (float)g;
      cB[o] = 0.f;
      cK[o] = 0.f;
    }
  }
}

static int check_act(const char* what, int dtype, int C, int ld_a, int ld_b) {
  const int ce = dtype == SFVOS_BF16 ? 8 : 4;
  SFVOS_REQUIRE(dtype == SFVOS_F32 || dtype == SFVOS_BF16, "%s: bad dtype", what);
  SFVOS_REQUIRE(C > 0 && C % 32 == 0 && C <= 256, "%s: C must be a multiple of 32, <= 256", what);
  SFVOS_REQUIRE(ld_a >= C && ld_b >= C && ld_a % ce == 0 && ld_b % ce == 0, "%s: bad pitch", what);
  return SFVOS_OK;
}

}  // namespace sfvos

using namespace sfvos;

extern "C" int sfvos_bn_finalize(const float* part, int n_levels, const int* rows_per_level,
                                 const int64_t* count_per_level, const float* gamma, const float* beta, float eps,
                                 int C, float* mean, float* rstd, float* scale, float* shift, float* save_var_unbiased,
                                 int coef_stride, sfvos_stream_t stream) {
  SFVOS_REQUIRE(part && rows_per_level && count_per_level && gamma && beta && mean && rstd && scale && shift &&
                    save_var_unbiased,
                "bn_finalize: null pointer");
  SFVOS_REQUIRE(n_levels >= 1 && n_levels <= SFVOS_MAX_LEVELS && C > 0 && C % 32 == 0 && coef_stride >= C,
                "bn_finalize: bad n_levels/C/coef_stride");
  FinalizeTab ft;
  ft.n = n_levels;
  int r = 0;
  for (int l = 0; l <= SFVOS_MAX_LEVELS; ++l) {
    ft.row_begin[l] = r;
    if (l < n_levels) {
      SFVOS_REQUIRE(rows_per_level[l] > 0 && count_per_level[l] > 0, "bn_finalize: level %d empty", l);
      r += rows_per_level[l];
      ft.count[l] = (double)count_per_level[l];
    } else if (l < SFVOS_MAX_LEVELS) {
      ft.count[l] = 1.0;
    }
  }
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(C / 32, n_levels), dim3(32 * RL), 0, (hipStream_t)stream, part, ft,
                     gamma, beta, eps, C, mean, rstd, scale, shift, save_var_unbiased, coef_stride);
  return check_launch("bn_finalize");
}

extern "C" int sfvos_bn_eval_coeffs(const float* gamma, const float* beta, const float* rm, const float* rv, float eps,
                                    int C, float* mean, float* rstd, float* scale, float* shift,
                                    sfvos_stream_t stream) {
  SFVOS_REQUIRE(gamma && beta && rm && rv && mean && rstd && scale && shift && C > 0, "bn_eval_coeffs: bad argument");
  hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3(ceil_div(C, 64)), dim3(64), 0, (hipStream_t)stream, gamma, beta, rm,
                     rv, eps, C, mean, rstd, scale, shift);
  return check_launch("bn_eval_coeffs");
}

extern "C" int sfvos_bn_running_update(float* rm, float* rv, const float* means, const float* vars, int n,
                                       int coef_stride, int C, float momentum, int64_t* num_batches_tracked,
                                       sfvos_stream_t stream) {
  SFVOS_REQUIRE(rm && rv && means && vars && n >= 0 && C > 0 && coef_stride >= C, "bn_running_update: bad argument");
  hipLaunchKernelGGL(bn_running_update_kernel, dim3(ceil_div(C, 64)), dim3(64), 0, (hipStream_t)stream, rm, rv, means,
                     vars, n, coef_stride, C, momentum, (long long*)num_batches_tracked);
  return check_launch("bn_running_update");
}

extern "C" int sfvos_bn_apply(const void* x, int ld_x, void* y, int ld_y, int dtype, const sfvos_levels* lv, int C,
                              const float* scale, const float* shift, int coef_stride, int relu,
                              const sfvos_bn_running* running, sfvos_stream_t stream) {
  int rc = check_act("bn_apply", dtype, C, ld_x, ld_y);
  if (rc) return rc;
  LevelTab lt;
  rc = make_level_tab(lv, &lt, "bn_apply");
  if (rc) return rc;
  RunUpd ru;
  rc = make_run_upd(running, &ru, "bn_apply");
  if (rc) return rc;
  SFVOS_REQUIRE(x && y && scale && shift && coef_stride >= C, "bn_apply: bad pointer / coef_stride");
  const int ce = dtype == SFVOS_BF16 ? 8 : 4;
  const unsigned grid = grid_for(lt.mb[SFVOS_MAX_LEVELS] * (C / ce), 256 * 4);
  hipStream_t s = (hipStream_t)stream;
  DT_DISPATCH(dtype,
              hipLaunchKernelGGL(bn_apply_kernel<SFVOS_F32>, dim3(grid), dim3(256), 0, s, (const char*)x, ld_x,
                                 (char*)y, ld_y, lt, C, scale, shift, coef_stride, relu, ru),
              hipLaunchKernelGGL(bn_apply_kernel<SFVOS_BF16>, dim3(grid), dim3(256), 0, s, (const char*)x, ld_x,
                                 (char*)y, ld_y, lt, C, scale, shift, coef_stride, relu, ru));
  return check_launch("bn_apply");
}

extern "C" int sfvos_bn_apply_fp8(const void* x, int ld_x, void* y, int ld_y, const sfvos_levels* lv, int C,
                                  const float* scale, const float* shift, int coef_stride, int relu, float act_scale,
                                  int* sat_count, const sfvos_bn_running* running, sfvos_stream_t stream) {
  LevelTab lt;
  int rc = make_level_tab(lv, &lt, "bn_apply_fp8");
  if (rc) return rc;
  RunUpd ru;
  rc = make_run_upd(running, &ru, "bn_apply_fp8");
  if (rc) return rc;
  SFVOS_REQUIRE(x && y && scale && shift && coef_stride >= C && act_scale > 0.f, "bn_apply_fp8: bad argument");
  SFVOS_REQUIRE(C > 0 && C % 16 == 0 && ld_x >= C && ld_x % 8 == 0 && ld_y >= C && ld_y % 16 == 0 &&
                    (((size_t)y) & 15) == 0,
                "bn_apply_fp8: C must be a multiple of 16, ld_x of 8 (bf16), ld_y of 16 (bytes), y 16-byte aligned");
  const unsigned grid = grid_for(lt.mb[SFVOS_MAX_LEVELS] * (C / 16), 256 * 4);
  hipLaunchKernelGGL(bn_apply_fp8_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const char*)x, ld_x, (char*)y,
                     ld_y, lt, C, scale, shift, coef_stride, relu, act_scale, sat_count, ru);
  return check_launch("bn_apply_fp8");
}

extern "C" int sfvos_bn_bwd_rows(const sfvos_levels* lv) {
  LevelTab lt;
  if (make_level_tab(lv, &lt, "bn_bwd_rows") != SFVOS_OK) return SFVOS_E_ARG;
  return lt.blk_begin[SFVOS_MAX_LEVELS];
}

extern "C" int sfvos_bn_bwd_reduce(const void* dy, int ld_dy, const void* x, int ld_x, int dtype,
                                   const sfvos_levels* lv, int C, const float* scale, const float* shift,
                                   const float* mean, const float* rstd, int coef_stride, int relu, float* part,
                                   sfvos_stream_t stream) {
  int rc = check_act("bn_bwd_reduce", dtype, C, ld_dy, ld_x);
  if (rc) return rc;
  LevelTab lt;
  rc = make_level_tab(lv, &lt, "bn_bwd_reduce");
  if (rc) return rc;
  SFVOS_REQUIRE(dy && x && scale && shift && mean && rstd && part && coef_stride >= C, "bn_bwd_reduce: bad argument");
  const unsigned grid = (unsigned)lt.blk_begin[SFVOS_MAX_LEVELS];
  hipStream_t s = (hipStream_t)stream;
  DT_DISPATCH(dtype,
              hipLaunchKernelGGL((bn_bwd_kernel<SFVOS_F32, false>), dim3(grid), dim3(BNB_THREADS), 0, s,
                                 (const char*)dy, ld_dy, (const char*)x, ld_x, (char*)nullptr, 0, lt, C, scale, shift,
                                 mean, rstd, coef_stride, relu, (const float*)nullptr, (const float*)nullptr,
                                 (const float*)nullptr, part, (const float*)nullptr, (const float*)nullptr,
                                 (float*)nullptr, (float*)nullptr, 0),
              hipLaunchKernelGGL((bn_bwd_kernel<SFVOS_BF16, false>), dim3(grid), dim3(BNB_THREADS), 0, s,
                                 (const char*)dy, ld_dy, (const char*)x, ld_x, (char*)nullptr, 0, lt, C, scale, shift,
                                 mean, rstd, coef_stride, relu, (const float*)nullptr, (const float*)nullptr,
                                 (const float*)nullptr, part, (const float*)nullptr, (const float*)nullptr,
                                 (float*)nullptr, (float*)nullptr, 0));
  return check_launch("bn_bwd_reduce");
}

extern "C" int sfvos_bn_bwd_finalize(const float* part, const sfvos_levels* lv, const float* gamma, const float* mean,
                                     const float* rstd, int coef_stride, int C, int train, float* coefA, float* coefB,
                                     float* coefK, float* sum_dz, float* sum_dzx, sfvos_stream_t stream) {
  LevelTab lt;
  int rc = make_level_tab(lv, &lt, "bn_bwd_finalize");
  if (rc) return rc;
  SFVOS_REQUIRE(part && gamma && mean && rstd && coefA && coefB && coefK && sum_dz && sum_dzx,
                "bn_bwd_finalize: null pointer");
  SFVOS_REQUIRE(C > 0 && C % 32 == 0 && coef_stride >= C, "bn_bwd_finalize: bad C/coef_stride");
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C / 32, lt.n), dim3(32 * RL), 0, (hipStream_t)stream, part, lt, gamma,
                     mean, rstd, coef_stride, C, train, coefA, coefB, coefK, sum_dz, sum_dzx);
  return check_launch("bn_bwd_finalize");
}

extern "C" int sfvos_bn_bwd_apply(const void* dy, int ld_dy, const void* x, int ld_x, void* dx, int ld_dx, int dtype,
                                  const sfvos_levels* lv, int C, const float* scale, const float* shift,
                                  int coef_stride, int relu, const float* coefA, const float* coefB,
                                  const float* coefK, float* bias_part, const float* sum_dz, const float* sum_dzx,
                                  float* dgamma, float* dbeta, int accumulate, sfvos_stream_t stream) {
  int rc = check_act("bn_bwd_apply", dtype, C, ld_dy, ld_x);
  if (rc) return rc;
  LevelTab lt;
  rc = make_level_tab(lv, &lt, "bn_bwd_apply");
  if (rc) return rc;
  SFVOS_REQUIRE(dy && x && dx && scale && shift && coefA && coefB && coefK && coef_stride >= C,
                "bn_bwd_apply: bad argument");
  SFVOS_REQUIRE(ld_dx >= C && ld_dx % (dtype == SFVOS_BF16 ? 8 : 4) == 0, "bn_bwd_apply: bad ld_dx");
  SFVOS_REQUIRE(!(dgamma || dbeta) || (sum_dz && sum_dzx), "bn_bwd_apply: dgamma / dbeta need the per-level sums");
  const unsigned grid = (unsigned)lt.blk_begin[SFVOS_MAX_LEVELS];
  hipStream_t s = (hipStream_t)stream;
  DT_DISPATCH(dtype,
              hipLaunchKernelGGL((bn_bwd_kernel<SFVOS_F32, true>), dim3(grid), dim3(BNB_THREADS), 0, s,
                                 (const char*)dy, ld_dy, (const char*)x, ld_x, (char*)dx, ld_dx, lt, C, scale, shift,
                                 (const float*)nullptr, (const float*)nullptr, coef_stride, relu, coefA, coefB, coefK,
                                 bias_part, sum_dz, sum_dzx, dgamma, dbeta, accumulate),
              hipLaunchKernelGGL((bn_bwd_kernel<SFVOS_BF16, true>), dim3(grid), dim3(BNB_THREADS), 0, s,
                                 (const char*)dy, ld_dy, (const char*)x, ld_x, (char*)dx, ld_dx, lt, C, scale, shift,
                                 (const float*)nullptr, (const float*)nullptr, coef_stride, relu, coefA, coefB, coefK,
                                 bias_part, sum_dz, sum_dzx, dgamma, dbeta, accumulate));
  return check_launch("bn_bwd_apply");
}
