// fe_solve.hip — fixed-effect trainer (include/gdmix_fe.h): one worker's shard resident in HBM, one L-BFGS
// evaluation = two streaming passes over it + a handful of small kernels; the L-BFGS driver is the same resumable
// compact-form step the team kernels use (re_lbfgs_compact.hpp), here with its state in HBM between kernels so
// that the caller can all-reduce [gradient, value] across workers in between
// (fixed_effect_lr_lbfgs_model.py:309-392: _train_model_fn; :394-430 _compute_loss_and_gradients; :635-643).
//
// The streaming passes (fe_stream_kernel) are HBM-bound and written for that: the non-zeros are cut into
// fixed blocks of FE_BLK consecutive entries whatever the row / column lengths are (the CSR arrays for
// X theta, the CSC copy for X'r), one workgroup per block; every lane loads 16 entries 256 apart (coalesced, all in
// flight at once), multiplies by the gathered vector element and parks the product in LDS; then the segments
// (rows / columns) that intersect the block are summed out of LDS — one thread per segment when they are short,
// one wavefront per segment when they are long — in entry order, so the result does not depend on the launch.
// A segment that crosses block boundaries leaves partial sums that a small second kernel adds up in block order.
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "re_internal.hpp"
#include "re_lbfgs_compact.hpp"
#include "../../include/gdmix_fe.h"

#include <new>

namespace gdmix {

#ifndef GDMIX_FE_BLK
#define GDMIX_FE_BLK 2048
#endif
constexpr int FE_BLK = GDMIX_FE_BLK;   // entries per workgroup of a streaming pass
constexpr int FE_THREADS = 256;
constexpr int FE_TILE_ROWS = 1 << 18;  // rows per tile of the CSC copy: 2 MiB of residuals, resident in an XCD's L2
// LDS index of product k: one pad double per 32 so that equal-length rows do not land on one bank
__device__ __forceinline__ int fe_slot(int k) { return k + (k >> 5); }
constexpr int FE_WAVES = FE_THREADS / WAVE;
constexpr int FE_DOT_BLOCKS = 512;
constexpr int FE_FIN_BLOCKS = 64;   // workgroups (= lanes of the final wavefront) that add up the per-block partial sums

struct FeDev {
  int n, d, ic, P, nblk, m;
  int64_t z, D;
  const int32_t* row_ptr;   // [n+1]
  const int32_t* csr_col;   // [z] local feature id
  const float* csr_val;
  const int32_t* col_ptr;   // [ntile*d+1] column copy cut into row tiles: segment t*d + c = column c, rows of tile t
  const int32_t* csc_row;   // [z] sorted by (tile, column, row)
  const float* csc_val;
  int ntile, nseg_c;        // row tiles, ntile * d
  const float *y, *o, *w;   // w may be NULL
  const int64_t* umap;      // [d] local -> global feature id
  double* xl;               // [d] x of the features present in this shard
  double* rs;               // [n] per-sample residual
  double* gl;               // [ntile*d] data gradient per (tile, local column)
  double* fg;               // [P + 1] global data gradient (intercept last), then the data value
  int32_t* own_r;           // [nblk+1] first row owned by a block of the CSR pass
  int32_t* own_c;           // [nblk+1] first column owned by a block of the CSC pass
  uint8_t *carry_r, *carry_c;   // [nblk]
  double *pf_r, *pl_r, *pf_c, *pl_c;                        // [nblk] partial sums of the first / last segment of a block
  double *loss_part, *rsum_part, *loss_fix, *rsum_fix;      // [nblk]
  double* acc_part;         // [FE_DOT_BLOCKS][TEAM_K]
  double* fin_part;         // [FE_FIN_BLOCKS][2]
  CompactState* state;
  CompactPlan* plan;
  CompactMats* mats;
  Work W;                   // global coefficient space, P each; ws / wy m*P
};

// first segment whose start position is >= pos (ptr non-decreasing, ptr[nseg] = z)
__device__ __forceinline__ int seg_lower_bound(const int32_t* __restrict__ ptr, int nseg, int64_t pos) {
  int lo = 0, hi = nseg;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if ((int64_t)ptr[mid] < pos) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// own[b] = first segment that starts in block b or later; carry[b] = the block begins inside a segment that
// started in an earlier block
__global__ void fe_own_kernel(const int32_t* __restrict__ ptr, int nseg, int nblk, int64_t z, int32_t* __restrict__ own,
                              uint8_t* __restrict__ carry) {
  for (int b = blockIdx.x * blockDim.x + threadIdx.x; b <= nblk; b += gridDim.x * blockDim.x) {
    const int64_t k0 = (int64_t)b * FE_BLK;
    const int o0 = (b == nblk) ? nseg : seg_lower_bound(ptr, nseg, k0);
    own[b] = o0;
    if (b < nblk) carry[b] = (b > 0 && k0 < z && (o0 >= nseg || (int64_t)ptr[o0] > k0)) ? 1 : 0;
  }
}

__global__ void fe_prepare_kernel(FeDev F) {
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < F.d; j += gridDim.x * blockDim.x) F.xl[j] = F.W.x[F.umap[j]];
}

// what is done with a finished segment sum
// HESS: the pass computes the diagonal of X~' D X~ instead of the gradient (fixed_effect_lr_lbfgs_model.py:271-296): the row pass
// leaves d_i = w_i rho_i (1 - rho_i), rho = sigmoid(logit), the column pass sums val^2 d_i
template <bool ROWS, bool HESS = false>
__device__ __forceinline__ void fe_emit(const FeDev& F, const SolveParams& o, int s, double sum, double xb, float fo, float fy, float fw,
                                        double& loss, double& rsum) {
  if (ROWS) {
    const double zi = sum + xb + (double)fo;
    const double yi = (double)fy;
    const double wi = (double)fw;
    double ri;
    if (HESS) {
      const double rho = sigmoid_full(zi);
      ri = wi * rho * (1.0 - rho);
    } else if (o.linear) {
      const double e = zi - yi;
      loss += wi * e * e;
      ri = 2.0 * wi * e;
    } else {
      loss += logistic_terms(zi, yi, wi, ri);
    }
    F.rs[s] = ri;
    rsum += ri;
  } else {
    F.gl[s] = sum;
  }
}

template <bool ROWS, bool HESS = false>
__global__ __launch_bounds__(FE_THREADS) void fe_stream_kernel(FeDev F, SolveParams o) {
  __shared__ double prod[FE_BLK + FE_BLK / 32];
  __shared__ double red[2][FE_WAVES];
  const int tid = threadIdx.x, lane = tid & (WAVE - 1), wv = tid >> 6;
  const int b = blockIdx.x;
  const int32_t* __restrict__ ptr = ROWS ? F.row_ptr : F.col_ptr;
  const int32_t* __restrict__ idx = ROWS ? F.csr_col : F.csc_row;
  const float* __restrict__ val = ROWS ? F.csr_val : F.csc_val;
  const double* __restrict__ vec = ROWS ? F.xl : F.rs;
  const int32_t* __restrict__ own = ROWS ? F.own_r : F.own_c;
  double* const pf = ROWS ? F.pf_r : F.pf_c;
  double* const pl = ROWS ? F.pl_r : F.pl_c;
  const uint8_t* __restrict__ cflag = ROWS ? F.carry_r : F.carry_c;
  const int64_t k0 = (int64_t)b * FE_BLK;
  const int64_t k1 = (k0 + FE_BLK < F.z) ? k0 + FE_BLK : F.z;
  const int cnt = (int)(k1 - k0);
  // Requests in dependency order, none waited for before it is needed: the block's descriptor, the streaming loads,
  // then (from the descriptor) what the first segment of this thread will need at the very end. A block costs three
  // memory round trips (descriptor | entries + segment data | gathers) instead of six in program order.
  const int o0 = own[b], o1 = own[b + 1];
  const bool carry = cflag[b] != 0;
  const double xb = (ROWS && F.ic) ? F.W.x[F.D] : 0.0;
  float v[FE_BLK / FE_THREADS];
  int c[FE_BLK / FE_THREADS];
#pragma unroll
  for (int q = 0; q < FE_BLK / FE_THREADS; ++q) {
    const int k = tid + q * FE_THREADS;
    const bool ok = k < cnt;
    v[q] = ok ? val[k0 + k] : 0.0f;
    c[q] = ok ? idx[k0 + k] : 0;
  }
  const int nwork = (o1 - o0) + (carry ? 1 : 0);
  const bool by_wave = nwork * 48 <= cnt;   // long segments: one wavefront each
  const int step = by_wave ? FE_WAVES : FE_THREADS;
  const int tfirst = by_wave ? wv : tid;
  int p0 = 0, p1 = 0;
  float fo = 0.0f, fy = 0.0f, fw = 1.0f;
  if (tfirst < nwork) {
    const bool is_carry = carry && tfirst == 0;
    const int s = is_carry ? o0 - 1 : o0 + tfirst - (carry ? 1 : 0);
    p0 = ptr[s];
    p1 = ptr[s + 1];
    if (ROWS) { fo = F.o[s]; fy = F.y[s]; if (F.w) fw = F.w[s]; }
  }
#pragma unroll
  for (int q = 0; q < FE_BLK / FE_THREADS; ++q) {
    const int k = tid + q * FE_THREADS;
    if (k < cnt) prod[fe_slot(k)] = (HESS && !ROWS) ? (double)v[q] * (double)v[q] * vec[c[q]] : (double)v[q] * vec[c[q]];
  }
  __syncthreads();
  double loss = 0.0, rsum = 0.0;
  for (int t = tfirst; t < nwork; t += step) {
    const bool is_carry = carry && t == 0;
    const int s = is_carry ? o0 - 1 : o0 + t - (carry ? 1 : 0);
    if (t != tfirst) {
      p0 = ptr[s];
      p1 = ptr[s + 1];
      if (ROWS) { fo = F.o[s]; fy = F.y[s]; fw = F.w ? F.w[s] : 1.0f; }
    }
    const int64_t a0 = is_carry ? k0 : (int64_t)p0;
    const int64_t a1full = (int64_t)p1;
    const int64_t a1 = a1full < k1 ? a1full : k1;
    const int lo = (int)(a0 - k0), hi = (int)(a1 - k0);
    double sum = 0.0;
    if (by_wave) {
      for (int k = lo + lane; k < hi; k += WAVE) sum += prod[fe_slot(k)];
      sum = wave_sum(sum);
    } else {
      for (int k = lo; k < hi; ++k) sum += prod[fe_slot(k)];
    }
    if (!by_wave || lane == 0) {
      if (is_carry) pf[b] = sum;                       // completed (or passed on) by fe_fix_kernel
      else if (a1full > k1) pl[b] = sum;               // continues in the next block
      else fe_emit<ROWS, HESS>(F, o, s, sum, xb, fo, fy, fw, loss, rsum);
    }
  }
  if (ROWS) {
    loss = wave_sum(loss);
    rsum = wave_sum(rsum);
    if (lane == 0) { red[0][wv] = loss; red[1][wv] = rsum; }
    __syncthreads();
    if (tid == 0) {
      double a = red[0][0], r = red[1][0];
#pragma unroll
      for (int w = 1; w < FE_WAVES; ++w) { a += red[0][w]; r += red[1][w]; }
      F.loss_part[b] = a;
      F.rsum_part[b] = r;
    }
  }
}

// segments that cross block boundaries: the block in which such a segment ends adds up its parts in block order
template <bool ROWS, bool HESS = false>
__global__ void fe_fix_kernel(FeDev F, SolveParams o) {
  const int32_t* __restrict__ ptr = ROWS ? F.row_ptr : F.col_ptr;
  const int32_t* __restrict__ own = ROWS ? F.own_r : F.own_c;
  const double* pf = ROWS ? F.pf_r : F.pf_c;
  const double* pl = ROWS ? F.pl_r : F.pl_c;
  const double xb = (ROWS && F.ic) ? F.W.x[F.D] : 0.0;
  for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < F.nblk; b += gridDim.x * blockDim.x) {
    double loss = 0.0, rsum = 0.0;
    const int64_t k0 = (int64_t)b * FE_BLK;
    const int64_t k1 = (k0 + FE_BLK < F.z) ? k0 + FE_BLK : F.z;
    const int o0 = own[b];
    const bool carry = (ROWS ? F.carry_r : F.carry_c)[b] != 0;
    if (carry) {
      const int s = o0 - 1;
      if ((int64_t)ptr[s + 1] <= k1) {   // ends here
        const int ob = (int)((int64_t)ptr[s] / FE_BLK);
        double t = pl[ob];
        for (int bb = ob + 1; bb <= b; ++bb) t += pf[bb];
        fe_emit<ROWS, HESS>(F, o, s, t, xb, ROWS ? F.o[s] : 0.0f, ROWS ? F.y[s] : 0.0f, (ROWS && F.w) ? F.w[s] : 1.0f, loss, rsum);
      }
    }
    if (ROWS) { F.loss_fix[b] = loss; F.rsum_fix[b] = rsum; }
  }
}

// local gradient into the global coefficient space; workgroup 0 also adds up the value and the intercept gradient
__global__ __launch_bounds__(FE_THREADS) void fe_finish_kernel(FeDev F) {
  __shared__ double red[2][FE_WAVES];
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < F.d; j += gridDim.x * blockDim.x) {
    double g = F.gl[j];
    for (int t = 1; t < F.ntile; ++t) g += F.gl[(size_t)t * F.d + j];   // tiles in row order
    F.fg[F.umap[j]] = g;
  }
  // the first FE_FIN_BLOCKS workgroups also add up a contiguous range of the per-block value / residual sums each
  if (blockIdx.x >= FE_FIN_BLOCKS) return;
  const int tid = threadIdx.x, lane = tid & (WAVE - 1), wv = tid >> 6;
  const int chunk = (F.nblk + FE_FIN_BLOCKS - 1) / FE_FIN_BLOCKS;
  const int b0 = blockIdx.x * chunk;
  const int b1 = (b0 + chunk < F.nblk) ? b0 + chunk : F.nblk;
  double a = 0.0, r = 0.0;
  for (int b = b0 + tid; b < b1; b += FE_THREADS) {
    a += F.loss_part[b]; a += F.loss_fix[b];
    r += F.rsum_part[b]; r += F.rsum_fix[b];
  }
  a = wave_sum(a);
  r = wave_sum(r);
  if (lane == 0) { red[0][wv] = a; red[1][wv] = r; }
  __syncthreads();
  if (tid == 0) {
    double sa = red[0][0], sr = red[1][0];
#pragma unroll
    for (int w = 1; w < FE_WAVES; ++w) { sa += red[0][w]; sr += red[1][w]; }
    F.fin_part[2 * blockIdx.x] = sa;
    F.fin_part[2 * blockIdx.x + 1] = sr;
  }
}

// one wavefront: the FE_FIN_BLOCKS range sums -> data value and intercept gradient
__global__ __launch_bounds__(WAVE) void fe_finish2_kernel(FeDev F) {
  const int lane = threadIdx.x;
  const double a = wave_sum(lane < FE_FIN_BLOCKS ? F.fin_part[2 * lane] : 0.0);
  const double r = wave_sum(lane < FE_FIN_BLOCKS ? F.fin_part[2 * lane + 1] : 0.0);
  if (lane == 0) {
    if (F.ic) F.fg[F.D] = r;
    F.fg[F.P] = a;
  }
}

// g = reduced data gradient + regulariser, and every product the driver needs (re_lbfgs_compact.hpp acc[] layout)
__global__ __launch_bounds__(FE_THREADS) void fe_dots_kernel(FeDev F, SolveParams o) {
  __shared__ double red[FE_WAVES][TEAM_K];
  const int tid = threadIdx.x, lane = tid & (WAVE - 1), wv = tid >> 6;
  const int col = F.state->col, head = F.state->head, m = o.m, P = F.P;
  double acc[TEAM_K];
#pragma unroll
  for (int k = 0; k < TEAM_K; ++k) acc[k] = 0.0;
  for (int j = blockIdx.x * blockDim.x + tid; j < P; j += gridDim.x * blockDim.x) {
    const bool reg = (j < F.D) || o.regularize_bias;   // the intercept is coefficient D
    const double xj = F.W.x[j];
    const double gj = F.fg[j] + (reg ? o.l2 * xj : 0.0);
    F.W.g[j] = gj;
    const double dj = F.W.d[j], rj = F.W.r[j];
    if (reg) acc[0] += xj * xj;
    acc[1] += gj * dj;
    acc[2] += gj * gj;
    const double yj = gj - rj;
    acc[3] += yj * yj;
    acc[4] += yj * gj;
    acc[TEAM_RD] += rj * dj;
    acc[TEAM_K - 1] = fmax(acc[TEAM_K - 1], fabs(gj));
    double2 h[TEAM_MCAP];   // all pairs requested before the first is used (see team_eval)
#pragma unroll
    for (int i = 0; i < TEAM_MCAP; ++i) {
      int sl = head + i;
      if (sl >= m) sl -= m;
      if (i >= m) sl = 0;
      h[i] = compact_hist(F.W, m, j)[sl * COMPACT_HIST_STRIDE];
    }
#pragma unroll
    for (int i = 0; i < TEAM_MCAP; ++i) {
      acc[5 + i] += (i < col ? h[i].x : 0.0) * yj;
      acc[5 + TEAM_MCAP + i] += (i < col ? h[i].y : 0.0) * yj;
    }
  }
  double mine = 0.0;
#pragma unroll
  for (int k = 0; k < TEAM_K; ++k) {
    const double t = (k == TEAM_K - 1) ? wave_max_nonneg(acc[k]) : wave_sum(acc[k]);
    if (lane == k) mine = t;
  }
  if (lane < TEAM_K) red[wv][lane] = mine;
  __syncthreads();
  if (tid < TEAM_K) {
    double s = red[0][tid];
#pragma unroll
    for (int w = 1; w < FE_WAVES; ++w) s = (tid == TEAM_K - 1) ? fmax(s, red[w][tid]) : s + red[w][tid];
    F.acc_part[(size_t)blockIdx.x * TEAM_K + tid] = s;
  }
}

// one workgroup: totals of the products, then the driver's decision
__global__ __launch_bounds__(FE_THREADS) void fe_step_kernel(FeDev F, SolveParams o, int dot_blocks, int32_t* status_out) {
  __shared__ double tot[TEAM_K];
  __shared__ CompactMats mats;
  const int tid = threadIdx.x;
  {
    // value v of block b by thread (b % 8) * 32 + v, eight partial totals per value, combined in order
    __shared__ double part8[8][32];
    const int v = tid & 31, g = tid >> 5;
    if (v < TEAM_K) {
      double s = 0.0;
      for (int b = g; b < dot_blocks; b += 8) {
        const double t = F.acc_part[(size_t)b * TEAM_K + v];
        s = (v == TEAM_K - 1) ? fmax(s, t) : s + t;
      }
      part8[g][v] = s;
    }
    __syncthreads();
    if (tid < TEAM_K) {
      double s = part8[0][tid];
#pragma unroll
      for (int k = 1; k < 8; ++k) s = (tid == TEAM_K - 1) ? fmax(s, part8[k][tid]) : s + part8[k][tid];
      tot[tid] = s;
    }
  }
  {
    const double* src = reinterpret_cast<const double*>(F.mats);
    double* dst = reinterpret_cast<double*>(&mats);
    for (int k = tid; k < (int)(sizeof(CompactMats) / sizeof(double)); k += FE_THREADS) dst[k] = src[k];
  }
  __syncthreads();
  double acc[TEAM_K];
#pragma unroll
  for (int k = 0; k < TEAM_K; ++k) acc[k] = tot[k];
  CompactState S = *F.state;
  CompactPlan plan;
  plan.action = CA_STOP; plan.col = S.col; plan.head = S.head; plan.stp = S.stp; plan.gamma = 1.0;
  const double f_new = F.fg[F.P] + 0.5 * o.l2 * acc[0];
  compact_advance(S, acc, f_new, o, mats, plan);
  __syncthreads();
  {
    double* dst = reinterpret_cast<double*>(F.mats);
    const double* src = reinterpret_cast<const double*>(&mats);
    for (int k = tid; k < (int)(sizeof(CompactMats) / sizeof(double)); k += FE_THREADS) dst[k] = src[k];
  }
  if (tid == 0) {
    *F.state = S;
    *F.plan = plan;
    *status_out = (plan.action == CA_STOP || plan.action == CA_STOP_RESTORE) ? S.status : -1;
  }
}

__global__ void fe_update_kernel(FeDev F, int m) {
  const CompactPlan plan = *F.plan;
  if (plan.action == CA_STOP) return;
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < F.P; j += gridDim.x * blockDim.x) {
    if (plan.action == CA_STOP_RESTORE) F.W.x[j] = F.W.t[j];
    else compact_update(plan, *F.mats, F.W, F.P, m, j);
  }
}

__global__ void fe_init_kernel(FeDev F, const double* __restrict__ theta0) {
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < F.P; j += gridDim.x * blockDim.x) {
    F.W.x[j] = theta0 ? theta0[j] : 0.0;
    F.W.d[j] = 0.0;
    F.W.r[j] = 0.0;
    F.W.g[j] = 0.0;
    F.W.t[j] = 0.0;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    CompactState S;
    compact_init(S);
    *F.state = S;
  }
}

// ---- row-tiled copy of the CSC arrays -----------------------------------------------------------------------------
// The column pass gathers the residual of every entry's row. With the columns stored whole, consecutive entries
// of a column are rows far apart and the residual vector (8 B x samples) does not fit an XCD's 4 MiB L2: every
// gather pulls a full line from the fabric (measured: 7.5 GB fetched for 1 GB of entries). Cutting the columns into
// row tiles of FE_TILE_ROWS and storing the entries tile-major (tile, column, row) keeps the residuals a pass touches
// at any time within 2 MiB; the per-(tile, column) sums are added up in tile order afterwards.
__global__ void fe_tile_count_kernel(const int32_t* __restrict__ col_ptr, const int32_t* __restrict__ csc_row, int d,
                                     uint32_t* __restrict__ key, uint32_t* __restrict__ perm, int32_t* __restrict__ cnt) {
  const int lane = threadIdx.x & (WAVE - 1);
  const int nw = (gridDim.x * blockDim.x) >> 6;
  for (int c = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; c < d; c += nw) {
    const int k0 = col_ptr[c], k1 = col_ptr[c + 1];
    for (int kb = k0; kb < k1; kb += WAVE) {   // wave-uniform trip count
      const int k = kb + lane;
      const bool in = k < k1;
      const int t = in ? csc_row[k] / FE_TILE_ROWS : -1;
      if (in) { key[k] = (uint32_t)t; perm[k] = (uint32_t)k; }
      // a column's entries are in row order, so equal tiles are runs of lanes: one atomic per run instead of one per entry
      // (same-address atomics serialise in L2: 8 ms of the 10 ms this build took)
      const int tp = __shfl_up(t, 1);
      const bool head = in && (lane == 0 || t != tp);
      const unsigned long long heads = __ballot(head);
      const unsigned long long valid = __ballot(in);
      if (head) {
        const unsigned long long above = (lane == WAVE - 1) ? 0ull : (heads >> (lane + 1)) << (lane + 1);
        const int end = above ? __ffsll((long long)above) - 1 : __popcll(valid);   // lanes in use are 0 .. popc(valid)-1
        atomicAdd(&cnt[(size_t)t * d + c], end - lane);
      }
    }
  }
}

__global__ void fe_tile_gather_kernel(const uint32_t* __restrict__ perm, const int32_t* __restrict__ csc_row,
                                      const float* __restrict__ csc_val, int64_t z, int32_t* __restrict__ trow,
                                      float* __restrict__ tval) {
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < z; k += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t src = perm[k];
    trow[k] = csc_row[src];
    tval[k] = csc_val[src];
  }
}

}  // namespace gdmix

using namespace gdmix;

// ---- scoring ----------------------------------------------------------------------------------------------------------
// logits of every sample of a raw shard under a global coefficient vector (intercept last): one thread per sample straight
// off the sample-major arrays the reader produced — no pack, no column copy; the coefficient vector (8 B x features) is
// gathered from L2. Sums in row order, starting from the intercept, as the packed scoring pass does.
__global__ __launch_bounds__(256) void fe_score_kernel(int64_t n, const int64_t* __restrict__ row_nnz_ptr,
                                                       const int64_t* __restrict__ col_global, const float* __restrict__ val,
                                                       const float* __restrict__ offset, const double* __restrict__ theta,
                                                       int64_t D, int ic, float* __restrict__ score, float* __restrict__ per_coord) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double acc = ic ? theta[D] : 0.0;
  if (row_nnz_ptr) {
    int64_t k = row_nnz_ptr[i];
    const int64_t k1 = row_nnz_ptr[i + 1];
    for (; k + 4 <= k1; k += 4) {
      const float v0 = val[k], v1 = val[k + 1], v2 = val[k + 2], v3 = val[k + 3];
      const double t0 = theta[col_global[k]], t1 = theta[col_global[k + 1]], t2 = theta[col_global[k + 2]], t3 = theta[col_global[k + 3]];
      acc += (double)v0 * t0;
      acc += (double)v1 * t1;
      acc += (double)v2 * t2;
      acc += (double)v3 * t3;
    }
    for (; k < k1; ++k) acc += (double)val[k] * theta[col_global[k]];
  }
  const double off = offset ? (double)offset[i] : 0.0;
  const double z = acc + off;
  score[i] = (float)z;
  per_coord[i] = (float)(z - off);
}

struct gdmix_fe_problem {
  gdmix_re_ctx* ctx;
  FeDev F;
  SolveParams o;
  void* pool;            // one device allocation carved into the arrays above
  void* tiled;           // row-tiled CSC copy (NULL when the shard has a single row tile)
  int32_t* status_dev;
  hipEvent_t ev[3];
  bool timed;
};

#define HIP_TRY(expr)                                                                   \
  do {                                                                                  \
    hipError_t _rc = (expr);                                                            \
    if (_rc != hipSuccess) {                                                            \
      set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_rc), __FILE__, __LINE__); \
      return GDMIX_RE_EHIP;                                                             \
    }                                                                                   \
  } while (0)

static size_t up256(size_t x) { return (x + 255) & ~(size_t)255; }

extern "C" {

GDMIX_API int gdmix_fe_create(gdmix_re_ctx* ctx, const gdmix_re_packed* b, int64_t num_features, const gdmix_re_opts* opts,
                              const double* theta0, gdmix_fe_problem** out, void* stream) {
  if (!out) { set_error("out is NULL"); return GDMIX_RE_EINVAL; }
  *out = nullptr;
  if (!ctx || !b || !opts) { set_error("NULL argument"); return GDMIX_RE_EINVAL; }
  if (b->E != 1) { set_error("the shard must be packed as one entity (E = %lld)", (long long)b->E); return GDMIX_RE_EINVAL; }
  if (opts->m < 1 || opts->m > TEAM_MCAP) { set_error("1 <= m <= %d", TEAM_MCAP); return GDMIX_RE_EINVAL; }
  if (opts->regularize_bias && !opts->has_intercept) { set_error("regularize_bias requires has_intercept"); return GDMIX_RE_EINVAL; }
  if (num_features < 1 || num_features > 0x7ffffff0ll) { set_error("bad num_features"); return GDMIX_RE_EINVAL; }
  if (b->Z > 0x7ffffff0ll || b->N > 0x7ffffff0ll) { set_error("shard exceeds 2^31 samples or non-zeros"); return GDMIX_RE_ERANGE; }
  hipStream_t s = static_cast<hipStream_t>(stream);
  gdmix_ctx_impl* ci = &ctx->impl;
  HIP_TRY(hipSetDevice(ci->device));
  gdmix_fe_problem* p = new (std::nothrow) gdmix_fe_problem();
  if (!p) { set_error("out of host memory"); return GDMIX_RE_ENOMEM; }
  p->ctx = ctx;
  p->pool = nullptr;
  p->tiled = nullptr;
  p->timed = false;
  for (auto& e : p->ev) e = nullptr;
  FeDev& F = p->F;
  const int ic = opts->has_intercept ? 1 : 0;
  F.n = (int)b->N; F.z = b->Z; F.d = (int)b->D; F.ic = ic; F.D = num_features; F.P = (int)num_features + ic; F.m = opts->m;
  F.ntile = (F.n + FE_TILE_ROWS - 1) / FE_TILE_ROWS;
  if (F.ntile < 1) F.ntile = 1;
  if ((int64_t)F.ntile * F.d > 0x7ffffff0ll) { set_error("row tiles x features exceeds 2^31"); delete p; return GDMIX_RE_ERANGE; }
  F.nseg_c = F.ntile * F.d;
  F.nblk = (int)((F.z + FE_BLK - 1) / FE_BLK);
  if (F.nblk < 1) F.nblk = 1;
  F.row_ptr = b->row_ptr; F.csr_col = b->csr_col; F.csr_val = b->csr_val; F.col_ptr = b->col_ptr; F.csc_row = b->csc_row;
  F.csc_val = b->csc_val; F.y = b->y; F.o = b->offset; F.w = b->weight; F.umap = b->unique_global;
  SolveParams& o = p->o;
  o.l2 = opts->l2; o.ftol = opts->ftol; o.pgtol = opts->pgtol; o.threshold = 0.0; o.regularize_bias = opts->regularize_bias;
  o.has_intercept = ic; o.m = opts->m; o.max_iter = opts->max_iter; o.maxfun = opts->maxfun; o.maxls = opts->maxls;
  o.variance_mode = 0; o.sum_loss = 1; o.linear = opts->linear ? 1 : 0;
  const size_t P = (size_t)F.P, nb = (size_t)F.nblk;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t r = off; off = up256(off + bytes); return r; };
  const size_t o_xl = take((size_t)(F.d + 1) * 8), o_rs = take((size_t)(F.n + 1) * 8), o_gl = take(((size_t)F.nseg_c + 1) * 8);
  const size_t o_fg = take((P + 1) * 8), o_ownr = take((nb + 1) * 4), o_ownc = take((nb + 1) * 4), o_cr = take(nb + 1), o_cc = take(nb + 1);
  const size_t o_part = take(nb * 8 * 8), o_acc = take((size_t)FE_DOT_BLOCKS * TEAM_K * 8), o_fin = take((size_t)FE_FIN_BLOCKS * 2 * 8);
  const size_t o_state = take(sizeof(CompactState)), o_plan = take(sizeof(CompactPlan)), o_mats = take(sizeof(CompactMats));
  const size_t o_vec = take(((size_t)5 * P + compact_hist_doubles((int64_t)P, opts->m)) * 8 + 16), o_status = take(64);
  hipError_t rc = hipMalloc(&p->pool, off);
  if (rc != hipSuccess) { set_error("hipMalloc(%zu) failed: %s", off, hipGetErrorString(rc)); delete p; return GDMIX_RE_ENOMEM; }
  char* base = static_cast<char*>(p->pool);
  rc = hipMemsetAsync(base, 0, off, s);
  if (rc != hipSuccess) { set_error("hipMemsetAsync failed: %s", hipGetErrorString(rc)); (void)hipFree(p->pool); delete p; return GDMIX_RE_EHIP; }
  F.xl = reinterpret_cast<double*>(base + o_xl); F.rs = reinterpret_cast<double*>(base + o_rs);
  F.gl = reinterpret_cast<double*>(base + o_gl); F.fg = reinterpret_cast<double*>(base + o_fg);
  F.own_r = reinterpret_cast<int32_t*>(base + o_ownr); F.own_c = reinterpret_cast<int32_t*>(base + o_ownc);
  F.carry_r = reinterpret_cast<uint8_t*>(base + o_cr); F.carry_c = reinterpret_cast<uint8_t*>(base + o_cc);
  double* part = reinterpret_cast<double*>(base + o_part);
  F.pf_r = part; F.pl_r = part + nb; F.pf_c = part + 2 * nb; F.pl_c = part + 3 * nb;
  F.loss_part = part + 4 * nb; F.rsum_part = part + 5 * nb; F.loss_fix = part + 6 * nb; F.rsum_fix = part + 7 * nb;
  F.acc_part = reinterpret_cast<double*>(base + o_acc);
  F.fin_part = reinterpret_cast<double*>(base + o_fin);
  F.state = reinterpret_cast<CompactState*>(base + o_state);
  F.plan = reinterpret_cast<CompactPlan*>(base + o_plan);
  F.mats = reinterpret_cast<CompactMats*>(base + o_mats);
  double* v = reinterpret_cast<double*>(base + o_vec);
  F.W.x = v; F.W.g = v + P; F.W.d = v + 2 * P; F.W.t = v + 3 * P; F.W.r = v + 4 * P;
  F.W.ws = v + 5 * P + ((5 * P) & 1);   // 16-byte aligned: the interleaved history (re_lbfgs_compact.hpp) is read with 16-byte loads
  F.W.wy = F.W.ws + (size_t)opts->m * P;
  F.W.rs = F.rs; F.W.alpha = nullptr; F.W.rho = nullptr; F.W.part = nullptr;
  p->status_dev = reinterpret_cast<int32_t*>(base + o_status);
  const int g = (int)((nb + 1 + 255) / 256);
  hipLaunchKernelGGL(fe_own_kernel, dim3(g), dim3(256), 0, s, F.row_ptr, F.n, F.nblk, F.z, F.own_r, F.carry_r);
  if (F.ntile > 1) {
    // build the row-tiled CSC copy: stable sort of the packed (column, row) order by tile, segment table by histogram + scan
    const size_t z = (size_t)F.z, ns = (size_t)F.nseg_c;
    size_t toff = 0;
    auto ttake = [&](size_t bytes) { size_t r = toff; toff = up256(toff + bytes); return r; };
    const size_t t_ptr = ttake((ns + 1) * 4), t_row = ttake(z * 4), t_val = ttake(z * 4);
    rc = hipMalloc(&p->tiled, toff);
    if (rc != hipSuccess) { set_error("hipMalloc(%zu) failed: %s", toff, hipGetErrorString(rc)); (void)hipFree(p->pool); delete p; return GDMIX_RE_ENOMEM; }
    char* tb = static_cast<char*>(p->tiled);
    int32_t* tptr = reinterpret_cast<int32_t*>(tb + t_ptr);
    int32_t* trow = reinterpret_cast<int32_t*>(tb + t_row);
    float* tval = reinterpret_cast<float*>(tb + t_val);
    unsigned bits = 1;
    while ((1 << bits) < F.ntile) ++bits;
    size_t sort_tmp = 0, scan_tmp = 0;
    rc = rocprim::radix_sort_pairs(nullptr, sort_tmp, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr,
                                   z, 0u, bits, s);
    if (rc == hipSuccess) rc = rocprim::exclusive_scan(nullptr, scan_tmp, (int32_t*)nullptr, (int32_t*)nullptr, 0, ns + 1, rocprim::plus<int32_t>(), s);
    void* tmp = nullptr;
    size_t woff = 0;
    auto wtake = [&](size_t bytes) { size_t r = woff; woff = up256(woff + bytes); return r; };
    const size_t w_key = wtake(z * 4), w_perm = wtake(z * 4), w_key2 = wtake(z * 4), w_perm2 = wtake(z * 4), w_cnt = wtake((ns + 1) * 4);
    const size_t w_lib = wtake(sort_tmp > scan_tmp ? sort_tmp : scan_tmp);
    if (rc == hipSuccess) rc = hipMalloc(&tmp, woff);
    if (rc != hipSuccess) { set_error("tiled column copy: %s", hipGetErrorString(rc)); (void)hipFree(p->tiled); (void)hipFree(p->pool); delete p; return GDMIX_RE_ENOMEM; }
    char* wb = static_cast<char*>(tmp);
    uint32_t* key = reinterpret_cast<uint32_t*>(wb + w_key);
    uint32_t* perm = reinterpret_cast<uint32_t*>(wb + w_perm);
    uint32_t* key2 = reinterpret_cast<uint32_t*>(wb + w_key2);
    uint32_t* perm2 = reinterpret_cast<uint32_t*>(wb + w_perm2);
    int32_t* cnt = reinterpret_cast<int32_t*>(wb + w_cnt);
    (void)hipMemsetAsync(cnt, 0, (ns + 1) * 4, s);
    hipLaunchKernelGGL(fe_tile_count_kernel, dim3(ci->num_cus * 8), dim3(256), 0, s, b->col_ptr, b->csc_row, F.d, key, perm, cnt);
    size_t lt = scan_tmp;
    rc = rocprim::exclusive_scan(wb + w_lib, lt, cnt, tptr, 0, ns + 1, rocprim::plus<int32_t>(), s);
    lt = sort_tmp;
    if (rc == hipSuccess) rc = rocprim::radix_sort_pairs(wb + w_lib, lt, key, key2, perm, perm2, z, 0u, bits, s);
    hipLaunchKernelGGL(fe_tile_gather_kernel, dim3(ci->num_cus * 16), dim3(256), 0, s, perm2, b->csc_row, b->csc_val, (int64_t)z, trow, tval);
    if (rc == hipSuccess) rc = hipStreamSynchronize(s);
    (void)hipFree(tmp);
    if (rc != hipSuccess) { set_error("tiled column copy: %s", hipGetErrorString(rc)); (void)hipFree(p->tiled); (void)hipFree(p->pool); delete p; return GDMIX_RE_EHIP; }
    F.col_ptr = tptr; F.csc_row = trow; F.csc_val = tval;
  }
  hipLaunchKernelGGL(fe_own_kernel, dim3(g), dim3(256), 0, s, F.col_ptr, F.nseg_c, F.nblk, F.z, F.own_c, F.carry_c);
  int gp = (int)((P + 255) / 256);
  if (gp > 1024) gp = 1024;
  hipLaunchKernelGGL(fe_init_kernel, dim3(gp), dim3(256), 0, s, F, theta0);
  rc = hipGetLastError();
  if (rc != hipSuccess) { set_error("launch failed: %s", hipGetErrorString(rc)); (void)hipFree(p->pool); delete p; return GDMIX_RE_EHIP; }
  *out = p;
  return GDMIX_RE_OK;
}

GDMIX_API void gdmix_fe_destroy(gdmix_fe_problem* p) {
  if (!p) return;
  for (auto& e : p->ev) if (e) (void)hipEventDestroy(e);
  if (p->pool) (void)hipFree(p->pool);
  if (p->tiled) (void)hipFree(p->tiled);
  delete p;
}

GDMIX_API double* gdmix_fe_reduce_buffer(gdmix_fe_problem* p, int64_t* count) {
  if (!p) return nullptr;
  if (count) *count = (int64_t)p->F.P + 1;
  return p->F.fg;
}

GDMIX_API int gdmix_fe_eval(gdmix_fe_problem* p, void* stream) {
  if (!p) { set_error("problem is NULL"); return GDMIX_RE_EINVAL; }
  hipStream_t s = static_cast<hipStream_t>(stream);
  const FeDev& F = p->F;
  if (!p->ev[0]) for (auto& e : p->ev) HIP_TRY(hipEventCreate(&e));
  HIP_TRY(hipMemsetAsync(F.fg, 0, ((size_t)F.P + 1) * 8, s));
  int gd = (F.d + 255) / 256;
  if (gd > 2048) gd = 2048;
  if (gd < 1) gd = 1;
  int gf = (F.nblk + 255) / 256;
  hipLaunchKernelGGL(fe_prepare_kernel, dim3(gd), dim3(256), 0, s, F);
  HIP_TRY(hipEventRecord(p->ev[0], s));
  hipLaunchKernelGGL((fe_stream_kernel<true>), dim3(F.nblk), dim3(FE_THREADS), 0, s, F, p->o);
  hipLaunchKernelGGL((fe_fix_kernel<true>), dim3(gf), dim3(256), 0, s, F, p->o);
  HIP_TRY(hipEventRecord(p->ev[1], s));
  hipLaunchKernelGGL((fe_stream_kernel<false>), dim3(F.nblk), dim3(FE_THREADS), 0, s, F, p->o);
  hipLaunchKernelGGL((fe_fix_kernel<false>), dim3(gf), dim3(256), 0, s, F, p->o);
  HIP_TRY(hipEventRecord(p->ev[2], s));
  hipLaunchKernelGGL(fe_finish_kernel, dim3(gd < FE_FIN_BLOCKS ? FE_FIN_BLOCKS : gd), dim3(FE_THREADS), 0, s, F);
  hipLaunchKernelGGL(fe_finish2_kernel, dim3(1), dim3(WAVE), 0, s, F);
  HIP_TRY(hipGetLastError());
  p->timed = true;
  return GDMIX_RE_OK;
}

GDMIX_API int gdmix_fe_hessian_diag(gdmix_fe_problem* p, const double* theta, void* stream) {
  if (!p) { set_error("problem is NULL"); return GDMIX_RE_EINVAL; }
  hipStream_t s = static_cast<hipStream_t>(stream);
  FeDev F = p->F;
  if (theta) F.W.x = const_cast<double*>(theta);   // the passes only read x
  HIP_TRY(hipMemsetAsync(F.fg, 0, ((size_t)F.P + 1) * 8, s));
  int gd = (F.d + 255) / 256;
  if (gd > 2048) gd = 2048;
  if (gd < 1) gd = 1;
  int gf = (F.nblk + 255) / 256;
  hipLaunchKernelGGL(fe_prepare_kernel, dim3(gd), dim3(256), 0, s, F);
  hipLaunchKernelGGL((fe_stream_kernel<true, true>), dim3(F.nblk), dim3(FE_THREADS), 0, s, F, p->o);
  hipLaunchKernelGGL((fe_fix_kernel<true, true>), dim3(gf), dim3(256), 0, s, F, p->o);
  hipLaunchKernelGGL((fe_stream_kernel<false, true>), dim3(F.nblk), dim3(FE_THREADS), 0, s, F, p->o);
  hipLaunchKernelGGL((fe_fix_kernel<false, true>), dim3(gf), dim3(256), 0, s, F, p->o);
  hipLaunchKernelGGL(fe_finish_kernel, dim3(gd < FE_FIN_BLOCKS ? FE_FIN_BLOCKS : gd), dim3(FE_THREADS), 0, s, F);
  hipLaunchKernelGGL(fe_finish2_kernel, dim3(1), dim3(WAVE), 0, s, F);   // fg[D] = sum_i d_i (the intercept's entry), fg[P] unused
  HIP_TRY(hipGetLastError());
  return GDMIX_RE_OK;
}

GDMIX_API int gdmix_fe_step(gdmix_fe_problem* p, void* stream, int32_t* status) {
  if (!p || !status) { set_error("NULL argument"); return GDMIX_RE_EINVAL; }
  hipStream_t s = static_cast<hipStream_t>(stream);
  const FeDev& F = p->F;
  int gp = (F.P + 255) / 256;
  int dot_blocks = gp < FE_DOT_BLOCKS ? gp : FE_DOT_BLOCKS;
  hipLaunchKernelGGL(fe_dots_kernel, dim3(dot_blocks), dim3(FE_THREADS), 0, s, F, p->o);
  hipLaunchKernelGGL(fe_step_kernel, dim3(1), dim3(FE_THREADS), 0, s, F, p->o, dot_blocks, p->status_dev);
  if (gp > 1024) gp = 1024;
  hipLaunchKernelGGL(fe_update_kernel, dim3(gp), dim3(256), 0, s, F, p->o.m);
  HIP_TRY(hipGetLastError());
  int32_t* hp = p->ctx->impl.host_pinned + 960;
  HIP_TRY(hipMemcpyAsync(hp, p->status_dev, sizeof(int32_t), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  *status = *hp;
  return GDMIX_RE_OK;
}

GDMIX_API int gdmix_fe_result(gdmix_fe_problem* p, double* theta, double* fval, double* gnorm, int32_t* nit, int32_t* nfev,
                              void* stream) {
  if (!p) { set_error("problem is NULL"); return GDMIX_RE_EINVAL; }
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (theta) HIP_TRY(hipMemcpyAsync(theta, p->F.W.x, (size_t)p->F.P * 8, hipMemcpyDeviceToDevice, s));
  CompactState S;
  HIP_TRY(hipMemcpyAsync(&S, p->F.state, sizeof(S), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  if (fval) *fval = S.f;
  if (gnorm) *gnorm = S.sbgnrm;
  if (nit) *nit = S.nit;
  if (nfev) *nfev = S.nfev;
  return GDMIX_RE_OK;
}

GDMIX_API int gdmix_fe_score(gdmix_re_ctx* ctx, int64_t n, const int64_t* row_nnz_ptr, const int64_t* col_global, const float* val,
                             const float* offset, const double* theta, int64_t num_features, int has_intercept, float* score,
                             float* per_coord, void* stream) {
  if (!ctx || n < 0 || !theta || !score || !per_coord || num_features < 0) { set_error("bad argument"); return GDMIX_RE_EINVAL; }
  if (row_nnz_ptr && (!col_global || !val)) { set_error("row_nnz_ptr without col_global / val"); return GDMIX_RE_EINVAL; }
  if (n == 0) return GDMIX_RE_OK;
  HIP_TRY(hipSetDevice(ctx->impl.device));
  const int64_t blocks = (n + 255) / 256;
  if (blocks > 0x7fffffffLL) { set_error("too many samples for one launch"); return GDMIX_RE_ERANGE; }
  hipLaunchKernelGGL(fe_score_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), n, row_nnz_ptr,
                     col_global, val, offset, theta, num_features, has_intercept ? 1 : 0, score, per_coord);
  HIP_TRY(hipGetLastError());
  return GDMIX_RE_OK;
}

GDMIX_API int gdmix_fe_last_eval_ms(gdmix_fe_problem* p, float* rows_ms, float* cols_ms) {
  if (!p || !p->timed) { set_error("no evaluation has been timed"); return GDMIX_RE_EINVAL; }
  HIP_TRY(hipEventSynchronize(p->ev[2]));
  if (rows_ms) HIP_TRY(hipEventElapsedTime(rows_ms, p->ev[0], p->ev[1]));
  if (cols_ms) HIP_TRY(hipEventElapsedTime(cols_ms, p->ev[1], p->ev[2]));
  return GDMIX_RE_OK;
}

}  // extern "C"
