// re_solve.hip — gfx950 kernels of the random-effect solve and score.
//
//   re_classify_kernel / re_order_kernel   bucket entities by the LDS footprint of their solve
//   re_solve_wave_kernel                   ONE WAVEFRONT PER ENTITY: the entity's (X, y, offset, weight)
//                                          block and all L-BFGS state live in LDS for the whole solve
//   re_solve_block_kernel                  one 256-thread workgroup per entity for blocks that do not
//                                          fit the LDS budget: X streams from HBM/L2 every evaluation,
//                                          L-BFGS state lives in a per-workgroup global scratch slot
//   re_score_kernel                        logits X~theta + offset, one thread per sample
#include "re_internal.hpp"
#include "re_solve_team.hpp"

namespace gdmix {

// ---------------------------------------------------------------------------------------------------
// classification
// ---------------------------------------------------------------------------------------------------
__global__ void re_classify_kernel(const int64_t* __restrict__ ent_row_ptr, const int64_t* __restrict__ ent_nnz_ptr,
                                   const int64_t* __restrict__ ent_feat_ptr, int64_t E, int ic, int m, bool has_w,
                                   ClassTable tab, int32_t* __restrict__ cls_out, int32_t* __restrict__ counts) {
  __shared__ int32_t local[GDMIX_RE_NUM_CLASSES];
  __shared__ int32_t tall_ge[TALL_ADAPT_STEPS], team_ge[TALL_TEAM_STEPS], mid_ge[TALL_MID_STEPS];
  if (threadIdx.x < GDMIX_RE_NUM_CLASSES) local[threadIdx.x] = 0;
  if (threadIdx.x < TALL_ADAPT_STEPS) tall_ge[threadIdx.x] = 0;
  if (threadIdx.x < TALL_TEAM_STEPS) team_ge[threadIdx.x] = 0;
  if (threadIdx.x < TALL_MID_STEPS) mid_ge[threadIdx.x] = 0;
  __syncthreads();
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += (int64_t)gridDim.x * blockDim.x) {
    const int n = (int)(ent_row_ptr[e + 1] - ent_row_ptr[e]);
    const int z = (int)(ent_nnz_ptr[e + 1] - ent_nnz_ptr[e]);
    const int d = (int)(ent_feat_ptr[e + 1] - ent_feat_ptr[e]);
    const int p = d + ic;
    int c = BLOCK_CLASS;
    // cheapest first: the group kernel with the fewest lanes and coefficient slots that holds the entity, smallest LDS bucket;
    // then the LDS-resident wavefront kernel (any m); the one-workgroup team kernel takes what is left
    const size_t wlds_bytes = wave_lds_bytes(p, n, z, d, m, has_w);
    for (int k = 0; k < BLOCK_CLASS; ++k) {
      if (tab.lds_bytes[k] <= 0) continue;
      const int kind = tab.kind[k];
      if (group_lanes(kind) > 0) {
        const int cap = group_lanes(kind) * group_epl(kind);
        if (m <= M_REG && p <= cap && n <= tab.ncap[k] && z <= tab.zcap[k]) { c = k; break; }
      } else if (kind == KIND_WLDS) {
        if (wlds_bytes <= (size_t)tab.lds_bytes[k]) { c = k; break; }
      }
    }
    if (tab.tall_min_n > 0 && m <= M_REG && p <= TALL_MAX_P && n >= tab.tall_min_n && !(tab.giant_nnz > 0 && z >= tab.giant_nnz)) c = (n >= tab.tall_split_n) ? TALL_CLASS
        : (tall_resident_bytes(1, tall_sets(1, p), d, n, z, has_w) <= (size_t)TALL_LEAN_ARENA ? TALL_L_CLASS : TALL_S_CLASS);
    else if (tab.giant_nnz > 0 && z >= tab.giant_nnz) c = GIANT_CLASS;
    else if (c == BLOCK_CLASS || (tab.team_nnz > 0 && z >= tab.team_nnz)) {
      // too large for a wavefront group: a team of CUs, sized by the non-zeros (streaming bandwidth)
      if (tab.team_nnz > 0 && z >= tab.team_nnz)
        c = (z >= 128 * tab.team_nnz) ? TEAM8_CLASS : ((z >= 8 * tab.team_nnz) ? TEAM32_CLASS : TEAM128_CLASS);
    }
    cls_out[e] = c;
    atomicAdd(&local[c], 1);
    if (c == TALL_S_CLASS && tab.tall_adapt_limit > 0 && n >= tall_adapt_n(0)) {
#pragma unroll
      for (int k = 0; k < TALL_ADAPT_STEPS; ++k)
        if (n >= tall_adapt_n(k)) atomicAdd(&tall_ge[k], 1);
    }
    if (c == TALL_S_CLASS && tab.tall_mid_n < 0 && n >= tall_mid_step(0)) {      // candidates of the mid class (class_base_kernel decides)
#pragma unroll
      for (int k = 0; k < TALL_MID_STEPS; ++k)
        if (n >= tall_mid_step(k)) atomicAdd(&mid_ge[k], 1);
    }
    if (c == TALL_S_CLASS && tab.tall_mid_n > 0 && n >= tab.tall_mid_n) atomicAdd(&mid_ge[0], 1);      // a fixed threshold: slot 0 counts them
    if (c == TALL_CLASS && tab.tall_team_n > 0 && n >= tab.tall_team_n) {   // candidates of the team class (class_base_kernel decides)
#pragma unroll
      for (int k = 0; k < TALL_TEAM_STEPS; ++k)
        if ((int64_t)n >= ((int64_t)tab.tall_team_n << k)) atomicAdd(&team_ge[k], 1);
    }
    if (c >= TEAM128_CLASS && c <= TEAM8_CLASS) {
      // work of the team tiers (non-zeros: total and largest entity), for the choice of the team size; rare entities
      atomicAdd(reinterpret_cast<unsigned long long*>(counts + 4 * GDMIX_RE_NUM_CLASSES) + c, (unsigned long long)z);
      atomicMax(counts + 3 * GDMIX_RE_NUM_CLASSES + c, z);
    }
  }
  __syncthreads();
  if (threadIdx.x < GDMIX_RE_NUM_CLASSES && local[threadIdx.x]) atomicAdd(&counts[threadIdx.x], local[threadIdx.x]);
  if (threadIdx.x < TALL_ADAPT_STEPS && tall_ge[threadIdx.x]) atomicAdd(&counts[3 * GDMIX_RE_NUM_CLASSES + threadIdx.x], tall_ge[threadIdx.x]);
  if (threadIdx.x < TALL_TEAM_STEPS && team_ge[threadIdx.x]) atomicAdd(&counts[3 * GDMIX_RE_NUM_CLASSES + TALL_TEAM_GE + threadIdx.x], team_ge[threadIdx.x]);
  if (threadIdx.x < TALL_MID_STEPS && mid_ge[threadIdx.x]) atomicAdd(&counts[3 * GDMIX_RE_NUM_CLASSES + TALL_MID_GE + threadIdx.x], mid_ge[threadIdx.x]);
}

// order[class_base[c] + k] = e. Position inside a class is by ticket: the launch order inside a class
// does not influence any entity's result (every entity is solved independently and deterministically),
// only which workgroup picks it up. Tickets are taken per workgroup (LDS histogram, then one global
// atomic per class per workgroup): per-entity global atomics on 8 addresses serialise in L2.
// split > 0 (class_base_kernel lowered the split of the tall classes for this batch): one-wavefront tall entities of at least
// `split` samples move to the eight-wavefront class here, in cls as well (the per-class times are attributed through it).
__global__ __launch_bounds__(256) void re_order_kernel(int32_t* __restrict__ cls, int64_t E,
                                                       const int32_t* __restrict__ class_base,
                                                       int32_t* __restrict__ cursor, int32_t* __restrict__ order,
                                                       const int64_t* __restrict__ ent_row_ptr, const int32_t* __restrict__ split_dev) {
  __shared__ int32_t cnt[GDMIX_RE_NUM_CLASSES], base[GDMIX_RE_NUM_CLASSES];
  const int split = *split_dev;
  const int team_from = split_dev[TALL_TEAM_SLOT - TALL_ADAPT_SLOT];   // > 0: eight-wavefront tall entities of at least this many samples get a team
  const int mid_from = split_dev[TALL_MID_SLOT - TALL_ADAPT_SLOT];     // > 0: one-wavefront tall entities of at least this many samples (below the split) go to the mid class
  const int64_t chunk = (int64_t)blockDim.x * 8;
  for (int64_t start = (int64_t)blockIdx.x * chunk; start < E; start += (int64_t)gridDim.x * chunk) {
    if (threadIdx.x < GDMIX_RE_NUM_CLASSES) cnt[threadIdx.x] = 0;
    __syncthreads();
    int c[8], pos[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int64_t e = start + (int64_t)k * blockDim.x + threadIdx.x;
      c[k] = (e < E) ? cls[e] : -1;
      if (split > 0 && c[k] == TALL_S_CLASS && ent_row_ptr[e + 1] - ent_row_ptr[e] >= split) { c[k] = TALL_CLASS; cls[e] = TALL_CLASS; }
      else if (team_from > 0 && c[k] == TALL_CLASS && ent_row_ptr[e + 1] - ent_row_ptr[e] >= team_from) { c[k] = TALL_T_CLASS; cls[e] = TALL_T_CLASS; }
      else if (mid_from > 0 && c[k] == TALL_S_CLASS && ent_row_ptr[e + 1] - ent_row_ptr[e] >= mid_from) { c[k] = TALL_M_CLASS; cls[e] = TALL_M_CLASS; }
      pos[k] = (c[k] >= 0) ? atomicAdd(&cnt[c[k]], 1) : 0;
    }
    __syncthreads();
    if (threadIdx.x < GDMIX_RE_NUM_CLASSES)
      base[threadIdx.x] = cnt[threadIdx.x] ? atomicAdd(&cursor[threadIdx.x], cnt[threadIdx.x]) : 0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int64_t e = start + (int64_t)k * blockDim.x + threadIdx.x;
      if (c[k] >= 0) order[class_base[c[k]] + base[c[k]] + pos[k]] = (int32_t)e;
    }
    __syncthreads();
  }
}

hipError_t launch_classify(const gdmix_re_packed* b, int ic, int m, const ClassTable& tab, int32_t* cls_tmp,
                           int32_t* counts_dev, hipStream_t s) {
  if (b->E == 0) return hipSuccess;
  int grid = (int)((b->E + 255) / 256);
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(re_classify_kernel, dim3(grid), dim3(256), 0, s, b->ent_row_ptr, b->ent_nnz_ptr,
                     b->ent_feat_ptr, b->E, ic, m, b->weight != nullptr, tab, cls_tmp, counts_dev);
  return hipGetLastError();
}

hipError_t launch_order(const gdmix_re_packed* b, int32_t* cls_tmp, const int32_t* class_base_dev,
                        int32_t* cursor_dev, hipStream_t s) {
  if (b->E == 0) return hipSuccess;
  int grid = (int)((b->E + 2047) / 2048);
  if (grid > 2048) grid = 2048;
  // (class_base_dev = counts + NUM_CLASSES: the chosen split sits two rows further)
  hipLaunchKernelGGL(re_order_kernel, dim3(grid), dim3(256), 0, s, cls_tmp, b->E, class_base_dev, cursor_dev,
                     b->order, b->ent_row_ptr, class_base_dev + 2 * GDMIX_RE_NUM_CLASSES + TALL_ADAPT_SLOT);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// shared epilogue: theta, thresholded theta, stats
// ---------------------------------------------------------------------------------------------------
template <class G>
__device__ __forceinline__ void write_results(G& grp, const OutDev& O, const SolveParams& o, int64_t e, int64_t c0,
                                              int p, const double* x, const SolveStats& st) {
  for (int j = grp.tid; j < p; j += grp.NT) {
    const double v = x[j];
    if (O.theta) O.theta[c0 + j] = v;
    // threshold_coefficients: |x| <= threshold -> 0.0, intercept included (util/model_utils.py:4-12)
    if (O.theta_thr) O.theta_thr[c0 + j] = (fabs(v) <= o.threshold) ? 0.0 : v;
  }
  if (grp.tid == 0) {
    if (O.fval) O.fval[e] = st.f;
    if (O.gnorm) O.gnorm[e] = st.gnorm;
    if (O.nit) O.nit[e] = st.nit;
    if (O.nfev) O.nfev[e] = st.nfev;
    if (O.status) O.status[e] = st.status;
  }
}

// ---------------------------------------------------------------------------------------------------
// four entities per wavefront, one per 16-lane DPP row (re_solve_quad.hpp)
// ---------------------------------------------------------------------------------------------------
// G lanes per entity: G = 16 -> four entities per wavefront ("quad"), G = 32 -> two ("pair").
#ifndef GDMIX_QUAD_WAVES_EPL4
#define GDMIX_QUAD_WAVES_EPL4 2
#endif
#ifndef GDMIX_QUAD_WAVES_EPL2
#define GDMIX_QUAD_WAVES_EPL2 2
#endif
// NCAP / ZCAP (sample / non-zero capacity of a row's LDS block) are template constants so that every LDS
// offset is an instruction immediate off one base register (runtime offsets cost ~12 VGPRs of addresses).
// G < 64: 64/G entities per wavefront; G = 64: one; G > 64: one entity per workgroup of G/64 wavefronts
// (cross-wave stage of every reduction through LDS + one barrier).
template <int G, int EPL, int NCAP, int ZCAP>
__global__ __launch_bounds__(G > WAVE ? G : WAVE)
__attribute__((amdgpu_waves_per_eu(EPL == 2 ? GDMIX_QUAD_WAVES_EPL2 : (EPL >= 3 ? GDMIX_QUAD_WAVES_EPL4 : 1)))) void re_solve_grp_kernel(
    BatchDev B, OutDev O, SolveParams o, const double* __restrict__ theta0, int begin, int count) {
  extern __shared__ __align__(16) unsigned char smem[];
  constexpr int NG = G >= WAVE ? 1 : WAVE / G;   // entities per workgroup
  constexpr int NWG = G > WAVE ? G / WAVE : 1;   // wavefronts per entity
  const int tid = threadIdx.x;
  const int row = (G >= WAVE) ? 0 : (tid / G);
  const int gl = (G >= WAVE) ? tid : (tid & (G - 1));
  const int slot = blockIdx.x * NG + row;
  const bool valid = slot < count;
  const int64_t e = valid ? (int64_t)B.order[begin + slot] : 0;
  const int ic = o.has_intercept ? 1 : 0;
  const int64_t r0 = B.ent_row_ptr[e], z0 = B.ent_nnz_ptr[e], f0 = B.ent_feat_ptr[e];
  const int n = valid ? (int)(B.ent_row_ptr[e + 1] - r0) : 0;
  const int nnz = valid ? (int)(B.ent_nnz_ptr[e + 1] - z0) : 0;
  const int d = valid ? (int)(B.ent_feat_ptr[e + 1] - f0) : 0;
  const int p = valid ? d + ic : 0;
  const int64_t c0 = f0 + e * ic;

  QuadLds L;
  L.q = quad_layout(G * EPL, NCAP, ZCAP, NWG, NG);
  L.base = smem + (size_t)row * L.q.bytes;
  L.hdr = L.base + (NWG > 1 ? (tid >> 6) * QUAD_HDR_BYTES : 0);
  L.has_w = B.weight != nullptr;
  XWave X;
  X.buf = reinterpret_cast<double*>(L.base + QUAD_HDR_BYTES * NWG);
  X.phase = 0;

  if (valid) {
    for (int k = gl; k < nnz; k += G) {
      L.csr()[k] = make_int2(B.csr_col[z0 + k], __float_as_int(B.csr_val[z0 + k]));
      L.csc()[k] = make_int2(B.csc_row[z0 + k], __float_as_int(B.csc_val[z0 + k]));
    }
    for (int i = gl; i < n; i += G) {
      L.y()[i] = B.y[r0 + i];
      L.o()[i] = B.offset[r0 + i];
      if (L.has_w) L.w()[i] = B.weight[r0 + i];
    }
    for (int i = gl; i <= n; i += G) L.row_ptr()[i] = B.row_ptr[r0 + e + i];
    for (int i = gl; i <= d; i += G) L.col_ptr()[i] = B.col_ptr[z0 + e + i];
  }
  // packed (start | len << 16) extents of the lane's first sample and of its coefficient slots
  unsigned rowc = 0, colc[EPL];
  if (gl < n) {
    const int k0 = B.row_ptr[r0 + e + gl], k1 = B.row_ptr[r0 + e + gl + 1];
    rowc = (unsigned)k0 | ((unsigned)(k1 - k0) << 16);
  }
  QuadState<EPL> V;
#pragma unroll
  for (int s = 0; s < EPL; ++s) {
    const int j = gl + G * s;
    colc[s] = 0;
    if (j < p && j >= ic) {
      const int k0 = B.col_ptr[z0 + e + (j - ic)], k1 = B.col_ptr[z0 + e + (j - ic) + 1];
      colc[s] = (unsigned)k0 | ((unsigned)(k1 - k0) << 16);
    }
    V.x[s] = (theta0 && j < p) ? theta0[c0 + j] : 0.0;
    V.g[s] = 0.0; V.d[s] = 0.0;
  }
  grp_fence<G>();
  SolveStats st;
  quad_solve<G, EPL, (NCAP > G)>(L, o, gl, n, p, ic, valid, rowc, colc, V, X, st);
  if (!valid) return;

#pragma unroll
  for (int s = 0; s < EPL; ++s) {
    const int j = gl + G * s;
    if (j < p) {
      const double v = V.x[s];
      if (O.theta) O.theta[c0 + j] = v;
      if (O.theta_thr) O.theta_thr[c0 + j] = (fabs(v) <= o.threshold) ? 0.0 : v;
    }
  }
  if (gl == 0) {
    if (O.fval) O.fval[e] = st.f;
    if (O.gnorm) O.gnorm[e] = st.gnorm;
    if (O.nit) O.nit[e] = st.nit;
    if (O.nfev) O.nfev[e] = st.nfev;
    if (O.status) O.status[e] = st.status;
  }
  if (o.variance_mode == GDMIX_RE_VAR_SIMPLE && O.variance) {
    // _compute_variance SIMPLE (binary_logistic_regression.py:175-180) with the final theta
    double* const xs = L.xs();
    double* const rs = L.rs();
    grp_fence<G>();
#pragma unroll
    for (int s = 0; s < EPL; ++s) {
      const int j = gl + G * s;
      if (j < p) xs[j] = V.x[s];
    }
    grp_fence<G>();
    const double x0 = ic ? xs[0] : 0.0;
    double dpart = 0.0;
    for (int i = gl; i < n; i += G) {
      const int k0 = L.row_ptr()[i], k1 = L.row_ptr()[i + 1];
      const double z = gather_dot(L.csr() + k0, k1 - k0, xs + ic, x0) + (double)L.o()[i];
      const double rho = sigmoid_full(z);
      const double di = rho * (1.0 - rho) * (L.has_w ? (double)L.w()[i] : 1.0);
      rs[i] = di;
      dpart += di;
    }
    const double dsum = grp_sum<G>(dpart, X);
    grp_fence<G>();
    const int first_reg = (ic && !o.regularize_bias) ? 1 : 0;
#pragma unroll
    for (int s = 0; s < EPL; ++s) {
      const int j = gl + G * s;
      if (j < p) {
        double h;
        if (ic && j == 0) {
          h = dsum;
        } else {
          h = 0.0;
          const int2* csc = L.csc();
          int k = (int)(colc[s] & 0xffffu);
          const int k1 = k + (int)(colc[s] >> 16);
          while (k < k1) {   // runs of equal sample = duplicates of one matrix cell
            const int rw = csc[k].x;
            double v = (double)__int_as_float(csc[k].y);
            ++k;
            while (k < k1 && csc[k].x == rw) { v += (double)__int_as_float(csc[k].y); ++k; }
            h += v * v * rs[rw];
          }
        }
        h += (j < first_reg) ? 0.0 : o.l2;
        O.variance[c0 + j] = 1.0 / (h + 1.0e-12);
      }
    }
  }
}

template <int G, int EPL, int NCAP, int ZCAP>
static hipError_t launch_quad_t(const BatchDev& B, const OutDev& O, const SolveParams& o, const double* theta0,
                                int begin, int count, hipStream_t s) {
  constexpr int NG = G >= WAVE ? 1 : WAVE / G;
  constexpr int NWG = G > WAVE ? G / WAVE : 1;
  const int row_lds_bytes = quad_layout(G * EPL, NCAP, ZCAP, NWG, NG).bytes;
  static DynLdsOnce lds_attr;
  if (hipError_t rc = lds_attr.set(reinterpret_cast<const void*>(re_solve_grp_kernel<G, EPL, NCAP, ZCAP>)); rc != hipSuccess) return rc;
  hipLaunchKernelGGL((re_solve_grp_kernel<G, EPL, NCAP, ZCAP>), dim3((count + NG - 1) / NG), dim3(G > WAVE ? G : WAVE),
                     (size_t)row_lds_bytes * NG, s, B, O, o, theta0, begin, count);
  return hipGetLastError();
}

hipError_t launch_solve_quad(int g, int epl, const BatchDev& B, const OutDev& O, const SolveParams& o, const double* theta0,
                             int begin, int count, int ncap, int zcap, hipStream_t s) {
  if (count <= 0) return hipSuccess;
#define GDMIX_GRP_CASE(GG, EE, NN, ZZ) \
  if (g == GG && epl == EE && ncap == NN && zcap == ZZ) return launch_quad_t<GG, EE, NN, ZZ>(B, O, o, theta0, begin, count, s);
  GDMIX_GRP_CASE(16, 2, 16, 64) GDMIX_GRP_CASE(16, 2, 32, 128) GDMIX_GRP_CASE(16, 2, 128, 512)
  GDMIX_GRP_CASE(16, 3, 16, 64) GDMIX_GRP_CASE(16, 3, 32, 128) GDMIX_GRP_CASE(16, 3, 128, 512)
  GDMIX_GRP_CASE(16, 4, 16, 64) GDMIX_GRP_CASE(16, 4, 32, 128) GDMIX_GRP_CASE(16, 4, 128, 512)
  GDMIX_GRP_CASE(32, 3, 32, 128) GDMIX_GRP_CASE(32, 3, 64, 256) GDMIX_GRP_CASE(32, 3, 256, 1024)
  GDMIX_GRP_CASE(32, 4, 32, 128) GDMIX_GRP_CASE(32, 4, 64, 256) GDMIX_GRP_CASE(32, 4, 256, 1024)
  GDMIX_GRP_CASE(64, 3, 64, 512) GDMIX_GRP_CASE(64, 3, 512, 2048) GDMIX_GRP_CASE(64, 4, 64, 512) GDMIX_GRP_CASE(64, 4, 512, 2048)
  GDMIX_GRP_CASE(128, 3, 128, 1024) GDMIX_GRP_CASE(128, 3, 1024, 3072) GDMIX_GRP_CASE(128, 4, 128, 1024) GDMIX_GRP_CASE(128, 4, 1024, 3072)
  GDMIX_GRP_CASE(256, 3, 256, 2048) GDMIX_GRP_CASE(256, 3, 2048, 4096) GDMIX_GRP_CASE(256, 4, 256, 2048) GDMIX_GRP_CASE(256, 4, 2048, 4096)
  GDMIX_GRP_CASE(512, 4, 512, 4096)
#undef GDMIX_GRP_CASE
  return hipErrorInvalidValue;
}

// ---------------------------------------------------------------------------------------------------
// one wavefront per entity, everything LDS-resident
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(WAVE) void re_solve_wave_kernel(BatchDev B, OutDev O, SolveParams o,
                                                             const double* __restrict__ theta0, int begin) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int lane = threadIdx.x;
  const int64_t e = B.order[begin + blockIdx.x];
  const int ic = o.has_intercept ? 1 : 0;
  const int64_t r0 = B.ent_row_ptr[e], z0 = B.ent_nnz_ptr[e], f0 = B.ent_feat_ptr[e];
  const int n = (int)(B.ent_row_ptr[e + 1] - r0);
  const int nnz = (int)(B.ent_nnz_ptr[e + 1] - z0);
  const int d = (int)(B.ent_feat_ptr[e + 1] - f0);
  const int p = d + ic;
  const int m = o.m;
  const int64_t c0 = f0 + e * ic;

  // carve-up: doubles first (8-byte aligned), then 4-byte arrays; must match wave_lds_bytes()
  double* dp = reinterpret_cast<double*>(smem);
  Work W;
  W.x = dp; dp += p;
  W.g = dp; dp += p;
  W.d = dp; dp += p;
  W.t = dp; dp += p;
  W.r = dp; dp += p;
  W.ws = dp; dp += (size_t)m * p;
  W.wy = dp; dp += (size_t)m * p;
  W.rs = dp; dp += n;
  W.alpha = dp; dp += m;
  W.rho = dp; dp += m;
  float* fp = reinterpret_cast<float*>(dp);
  float* s_csr_val = fp; fp += nnz;
  float* s_csc_val = fp; fp += nnz;
  float* s_y = fp; fp += n;
  float* s_o = fp; fp += n;
  float* s_w = nullptr;
  if (B.weight) { s_w = fp; fp += n; }
  int32_t* ip = reinterpret_cast<int32_t*>(fp);
  int32_t* s_csr_col = ip; ip += nnz;
  int32_t* s_csc_row = ip; ip += nnz;
  int32_t* s_row_ptr = ip; ip += n + 1;
  int32_t* s_col_ptr = ip; ip += d + 1;

  // stage the entity's block: every array is a contiguous slice of the packed batch -> coalesced reads
  for (int k = lane; k < nnz; k += WAVE) {
    s_csr_val[k] = B.csr_val[z0 + k];
    s_csr_col[k] = B.csr_col[z0 + k];
    s_csc_val[k] = B.csc_val[z0 + k];
    s_csc_row[k] = B.csc_row[z0 + k];
  }
  for (int i = lane; i < n; i += WAVE) {
    s_y[i] = B.y[r0 + i];
    s_o[i] = B.offset[r0 + i];
    if (s_w) s_w[i] = B.weight[r0 + i];
  }
  for (int i = lane; i <= n; i += WAVE) s_row_ptr[i] = B.row_ptr[r0 + e + i];
  for (int i = lane; i <= d; i += WAVE) s_col_ptr[i] = B.col_ptr[z0 + e + i];
  for (int j = lane; j < p; j += WAVE) W.x[j] = theta0 ? theta0[c0 + j] : 0.0;

  WaveGroup grp{lane};
  grp.sync();
  EntityView P{n, d, p, ic, s_row_ptr, s_csr_col, s_csr_val, s_col_ptr, s_csc_row, s_csc_val, s_y, s_o, s_w};
  SolveStats st;
  lbfgs_solve(grp, P, o, W, st);
  write_results(grp, O, o, e, c0, p, W.x, st);
  if (o.variance_mode == GDMIX_RE_VAR_SIMPLE && O.variance) variance_simple(grp, P, o, W, O.variance + c0);
}

hipError_t launch_solve_wave(const BatchDev& B, const OutDev& O, const SolveParams& o, const double* theta0,
                             int begin, int count, int lds_bytes, hipStream_t s) {
  if (count <= 0) return hipSuccess;
  static DynLdsOnce lds_attr;
  if (hipError_t rc = lds_attr.set(reinterpret_cast<const void*>(re_solve_wave_kernel)); rc != hipSuccess) return rc;
  hipLaunchKernelGGL(re_solve_wave_kernel, dim3(count), dim3(WAVE), (size_t)lds_bytes, s, B, O, o, theta0, begin);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// one workgroup per entity, state in a global scratch slot, X streamed from HBM/L2
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(WAVE* BLOCK_NW) void re_solve_block_kernel(BatchDev B, OutDev O, SolveParams o,
                                                                        const double* __restrict__ theta0,
                                                                        int begin, int count, double* scratch,
                                                                        size_t slot_doubles, int64_t max_p) {
  __shared__ double red[2 * BLOCK_NW];
  const int ic = o.has_intercept ? 1 : 0;
  const int m = o.m;
  double* slot = scratch + (size_t)blockIdx.x * slot_doubles;
  for (int idx = blockIdx.x; idx < count; idx += gridDim.x) {
    const int64_t e = B.order[begin + idx];
    const int64_t r0 = B.ent_row_ptr[e], z0 = B.ent_nnz_ptr[e], f0 = B.ent_feat_ptr[e];
    const int n = (int)(B.ent_row_ptr[e + 1] - r0);
    const int d = (int)(B.ent_feat_ptr[e + 1] - f0);
    const int p = d + ic;
    const int64_t c0 = f0 + e * ic;
    double* dp = slot;
    Work W;
    W.x = dp; dp += max_p;
    W.g = dp; dp += max_p;
    W.d = dp; dp += max_p;
    W.t = dp; dp += max_p;
    W.r = dp; dp += max_p;
    W.ws = dp; dp += (size_t)m * max_p;
    W.wy = dp; dp += (size_t)m * max_p;
    W.alpha = dp; dp += m;
    W.rho = dp; dp += m;
    W.rs = dp;
    BlockGroup<BLOCK_NW> grp{(int)threadIdx.x, red, 0};
    for (int j = grp.tid; j < p; j += grp.NT) W.x[j] = theta0 ? theta0[c0 + j] : 0.0;
    grp.sync();
    EntityView P{n, d, p, ic, B.row_ptr + r0 + e, B.csr_col + z0, B.csr_val + z0, B.col_ptr + z0 + e,
                 B.csc_row + z0, B.csc_val + z0, B.y + r0, B.offset + r0, B.weight ? B.weight + r0 : nullptr};
    SolveStats st;
    lbfgs_solve(grp, P, o, W, st);
    write_results(grp, O, o, e, c0, p, W.x, st);
    if (o.variance_mode == GDMIX_RE_VAR_SIMPLE && O.variance) variance_simple(grp, P, o, W, O.variance + c0);
    grp.sync();   // the slot is reused by the next entity
  }
}

// ---------------------------------------------------------------------------------------------------
// order of a class by size, largest first (the persistent team kernel hands entities out in this order so
// that the long solves start first). One workgroup, bitonic sort in LDS; ties by entity index, so the order
// and with it the launch are reproducible. Lists longer than SORT_CAP stay in ticket order.
// ---------------------------------------------------------------------------------------------------
constexpr int SORT_CAP = 16384;   // 128 KB of keys in LDS
constexpr int SORT_RANK_CAP = 4096;
__global__ __launch_bounds__(1024) void re_sort_class_kernel(int32_t* __restrict__ list, int count,
                                                             const int64_t* __restrict__ ent_nnz_ptr) {
  extern __shared__ __align__(16) unsigned char sort_smem[];
  unsigned long long* const key = reinterpret_cast<unsigned long long*>(sort_smem);
  int np2 = 1;
  while (np2 < count) np2 <<= 1;
  for (int i = threadIdx.x; i < np2; i += blockDim.x) {
    unsigned long long k = ~0ull;
    if (i < count) {
      const int64_t e = list[i];
      const unsigned long long z = (unsigned long long)(ent_nnz_ptr[e + 1] - ent_nnz_ptr[e]);
      k = ((0xffffffffull - (z & 0xffffffffull)) << 32) | (unsigned long long)(uint32_t)e;
    }
    key[i] = k;
  }
  __syncthreads();
  if (count <= SORT_RANK_CAP) {
    // short lists: the rank of a key is the number of smaller keys (keys are distinct: the entity index is part of them) — one
    // barrier instead of the ~60 of the bitonic network, whose stages each wait for the slowest wavefront of a CU the group kernels
    // of the same solve are using (1 746 entities of a MovieLens share: 325 us on the critical path of a 3 ms step).
    unsigned long long mine[SORT_RANK_CAP / 1024];
    int rank[SORT_RANK_CAP / 1024];
#pragma unroll
    for (int q = 0; q < SORT_RANK_CAP / 1024; ++q) {
      const int i = threadIdx.x + q * 1024;
      mine[q] = i < count ? key[i] : ~0ull;
      rank[q] = 0;
    }
    const int nq = (count + 1023) / 1024;      // rows of keys in use (uniform)
    for (int t = 0; t < count; ++t) {
      const unsigned long long other = key[t];   // the same address for every lane: one broadcast read
#pragma unroll
      for (int q = 0; q < SORT_RANK_CAP / 1024; ++q)
        if (q < nq) rank[q] += other < mine[q] ? 1 : 0;
    }
#pragma unroll
    for (int q = 0; q < SORT_RANK_CAP / 1024; ++q)
      if (threadIdx.x + q * 1024 < count) list[rank[q]] = (int32_t)(uint32_t)mine[q];
    return;
  }
  for (int k = 2; k <= np2; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = threadIdx.x; t < np2; t += blockDim.x) {
        const int partner = t ^ j;
        if (partner > t) {
          const unsigned long long a = key[t], b = key[partner];
          const bool up = (t & k) == 0;
          if ((a > b) == up) { key[t] = b; key[partner] = a; }
        }
      }
      __syncthreads();
    }
  }
  for (int i = threadIdx.x; i < count; i += blockDim.x) list[i] = (int32_t)(uint32_t)key[i];
}

void launch_sort_class(int32_t* list, int count, const int64_t* ent_nnz_ptr, hipStream_t s) {
  if (count <= 1 || count > SORT_CAP) return;
  static DynLdsOnce lds_attr;
  if (lds_attr.set(reinterpret_cast<const void*>(re_sort_class_kernel)) != hipSuccess) return;   // stays in ticket order: slower tail, same results
  int np2 = 1;
  while (np2 < count) np2 <<= 1;
  hipLaunchKernelGGL(re_sort_class_kernel, dim3(1), dim3(1024), (size_t)np2 * 8, s, list, count, ent_nnz_ptr);
}

// ---------------------------------------------------------------------------------------------------
// team kernels (re_solve_team.hpp): compact-form L-BFGS with X and the state vectors in HBM.
//   GRID = false  one workgroup per entity, entities strided over the grid, one scratch slot per workgroup
//   GRID = true   a persistent grid of one workgroup per CU split into `teams` teams (workgroup b belongs to
//                 team b % teams; placement only matters for speed); every team
//                 takes entities team, team + teams, ... of the class, one after another
// ---------------------------------------------------------------------------------------------------
#ifndef GDMIX_TEAM_WAVES_PER_EU
#define GDMIX_TEAM_WAVES_PER_EU 2
#endif
// Where the five p-vectors and the residuals of a one-workgroup entity live (round 4). The workgroup kernel holds one entity per CU
// (eight wavefronts of up to 256 VGPRs) and its static LDS is 41 KB, so ~119 KB of the CU's LDS are free: x, g, d (and x_old,
// g_old where they fit) and the per-sample residuals go there instead of the global scratch slot — a quarter of the reads and
// three quarters of the writes of the class, and the two gather passes (x by the rows, the residuals by
// the columns) become LDS gathers. The history stays in HBM. VEC = 0: everything in the slot (entities too large, and every
// multi-workgroup team: x and the residuals are read by all workgroups); 1: x + residuals (the two gather targets; p up to
// ~15 k); 3: x, g, d + residuals (p up to ~4.9 k); 5: + x_old, g_old (p up to ~3 k).
// Same arithmetic in the same order: results do not depend on the placement (test_block_kernel_*, test_team_lds_vectors_*).
extern __shared__ __attribute__((aligned(16))) double team_arena[];
__host__ __device__ inline int team_vec_level(int p, int n, int arena_doubles) {
  const long pp = (p + 1) & ~1, nn = (n + 1) & ~1;
  if (5 * pp + nn <= arena_doubles) return 5;
  if (3 * pp + nn <= arena_doubles) return 3;
  if (pp + nn <= arena_doubles) return 1;
  return 0;
}

template <int NW, int VEC>
__device__ __forceinline__ void team_entity(Team<NW>& tm, const EntityView& P, const SolveParams& o, Work W, const OutDev& O,
                                            const double* __restrict__ theta0, int64_t e, int64_t c0) {
  const int p = P.p;
  if (VEC >= 1) {
    const int pp = (p + 1) & ~1;
    double* a = team_arena;
    W.x = a; a += pp;
    if (VEC >= 3) { W.g = a; a += pp; W.d = a; a += pp; }
    if (VEC >= 5) { W.t = a; a += pp; W.r = a; a += pp; }
    W.rs = a;
  }
  for (int j = tm.tid; j < p; j += tm.NT) W.x[j] = theta0 ? theta0[c0 + j] : 0.0;
  SolveStats st;
  team_solve(tm, P, o, W, st);
  TeamAsGroup<NW> grp{tm, tm.tid, tm.NT};
  write_results(grp, O, o, e, c0, p, W.x, st);
  if (o.variance_mode == GDMIX_RE_VAR_SIMPLE && O.variance) variance_simple(grp, P, o, W, O.variance + c0);
  tm.sync();   // the slot (and the arena) is reused by the next entity
}

template <int NW, bool GRID>
__global__ __launch_bounds__(WAVE* NW) __attribute__((amdgpu_waves_per_eu(GDMIX_TEAM_WAVES_PER_EU))) void re_solve_team_kernel(BatchDev B, OutDev O, SolveParams o,
                                                                 const double* __restrict__ theta0, int begin, int count,
                                                                 double* scratch, size_t slot_doubles, int64_t max_p,
                                                                 TeamSync* gs, int teams, int arena_doubles, int xcd_barrier) {
  __shared__ TeamLds<NW> lds;
  static_assert(sizeof(TeamLds<NW>) % 16 == 0, "the dynamic arena behind the static LDS must stay 16-byte aligned");
  const int ic = o.has_intercept ? 1 : 0;
  const int m = o.m;
  Team<NW> tm;
  int team = 0;
  if (GRID) {
    team = (int)blockIdx.x % teams;
    tm.bid = blockIdx.x / teams;
    tm.nblocks = gridDim.x / teams;
    tm.tid = (int)(tm.bid * blockDim.x + threadIdx.x);
    tm.NT = (int)(tm.nblocks * blockDim.x);
    gs += team;
  } else {
    tm.tid = (int)threadIdx.x;
    tm.NT = WAVE * NW;
    tm.nblocks = 1;
    tm.bid = 0;
  }
  tm.wid = tm.tid >> 6;
  tm.nwaves = tm.NT >> 6;
  tm.lane = (int)threadIdx.x & (WAVE - 1);
  tm.gs = gs;
  tm.L = &lds;
  tm.epoch = 0;
  tm.phase = 0;
  tm.one_xcd = false;
  if (GRID && xcd_barrier) tm.team_placement();
  double* slot = scratch + (size_t)(GRID ? team : (int)blockIdx.x) * slot_doubles;
  for (int idx = (int)blockIdx.x;; idx += (int)gridDim.x) {
    if (GRID) {
      // teams take the next entity of the class (largest first, re_sort_class_kernel) as they become free
      if (tm.bid == 0 && threadIdx.x == 0)
        gs->cur = (int)__hip_atomic_fetch_add(&(gs - team)->next, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      tm.sync();
      idx = gs->cur;
    }
    if (idx >= count) break;
    const int64_t e = B.order[begin + idx];
    const int64_t r0 = B.ent_row_ptr[e], z0 = B.ent_nnz_ptr[e], f0 = B.ent_feat_ptr[e];
    const int n = (int)(B.ent_row_ptr[e + 1] - r0);
    const int d = (int)(B.ent_feat_ptr[e + 1] - f0);
    const int p = d + ic;
    const int64_t c0 = f0 + e * ic;
    double* dp = slot;
    Work W;
    W.x = dp; dp += max_p;
    W.g = dp; dp += max_p;
    W.d = dp; dp += max_p;
    W.t = dp; dp += max_p;
    W.r = dp; dp += max_p;
    dp += (reinterpret_cast<uintptr_t>(dp) >> 3) & 1;   // the history is read with 16-byte loads
    W.ws = dp; dp += (size_t)2 * m * ((max_p + 63) & ~(int64_t)63);   // tiles of 64 coefficients (re_lbfgs_compact.hpp)
    W.wy = nullptr;
    W.alpha = dp; dp += m;
    W.rho = dp; dp += m;
    W.part = dp; dp += TEAM_LONG_CAP * WAVE;
    W.rs = dp;
    EntityView P{n, d, p, ic, B.row_ptr + r0 + e, B.csr_col + z0, B.csr_val + z0, B.col_ptr + z0 + e,
                 B.csc_row + z0, B.csc_val + z0, B.y + r0, B.offset + r0, B.weight ? B.weight + r0 : nullptr};
    const int vec = GRID ? 0 : team_vec_level(p, n, arena_doubles);   // uniform over the workgroup
    if (!GRID && vec == 5) team_entity<NW, 5>(tm, P, o, W, O, theta0, e, c0);
    else if (!GRID && vec == 3) team_entity<NW, 3>(tm, P, o, W, O, theta0, e, c0);
    else if (!GRID && vec == 1) team_entity<NW, 1>(tm, P, o, W, O, theta0, e, c0);
    else team_entity<NW, 0>(tm, P, o, W, O, theta0, e, c0);
  }
}

template <int NW>
static hipError_t launch_team_block(const BatchDev& B, const OutDev& O, const SolveParams& o, const double* theta0,
                                    int begin, int count, double* scratch, size_t slot_doubles, int slots,
                                    int64_t max_p, hipStream_t s) {
  int grid = count < slots ? count : slots;
  // largest first (workgroups start in index order and take the next entity as they finish): a big entity started last
  // would be the tail of the launch
  launch_sort_class(const_cast<int32_t*>(B.order) + begin, count, B.ent_nnz_ptr, s);
  // the LDS the workgroup leaves free (one workgroup per CU: eight wavefronts at two per SIMD) holds the p-vectors of an
  // entity that fits (team_vec_level); GDMIX_TEAM_ARENA_KB=0 keeps everything in the scratch slot (A/B and tests)
  constexpr int kArenaMax = (160 * 1024 - (int)sizeof(TeamLds<NW>) - 64) & ~15;
  int arena_bytes = kArenaMax;
  if (const char* ev = getenv("GDMIX_TEAM_ARENA_KB")) {
    const int kb = atoi(ev);
    if (kb >= 0 && kb * 1024 < arena_bytes) arena_bytes = kb * 1024;
  }
  static DynLdsOnce lds_attr;   // (the limit counts the dynamic part only: static + dynamic must stay within the CU's 160 KB)
  if (hipError_t err = lds_attr.set(reinterpret_cast<const void*>(&re_solve_team_kernel<NW, false>), kArenaMax); err != hipSuccess) return err;
  hipLaunchKernelGGL((re_solve_team_kernel<NW, false>), dim3(grid), dim3(WAVE * NW), (size_t)arena_bytes, s, B, O, o, theta0, begin,
                     count, scratch, slot_doubles, max_p, static_cast<TeamSync*>(nullptr), 1, arena_bytes / 8, 0);
  return hipGetLastError();
}

hipError_t launch_solve_block(const BatchDev& B, const OutDev& O, const SolveParams& o, const double* theta0,
                              int begin, int count, double* scratch, size_t slot_doubles, int slots,
                              int64_t max_p, hipStream_t s) {
  if (count <= 0) return hipSuccess;
  if (o.m > TEAM_MCAP) {   // two-loop form, any m
    int grid = count < slots ? count : slots;
    hipLaunchKernelGGL(re_solve_block_kernel, dim3(grid), dim3(WAVE * BLOCK_NW), 0, s, B, O, o, theta0, begin, count,
                       scratch, slot_doubles, max_p);
    return hipGetLastError();
  }
  return launch_team_block<TEAM_BLOCK_NW>(B, O, o, theta0, begin, count, scratch, slot_doubles, slots, max_p, s);
}

hipError_t launch_solve_grid(const BatchDev& B, const OutDev& O, const SolveParams& o, const double* theta0,
                             int begin, int count, double* scratch, size_t slot_doubles, int64_t max_p,
                             void* sync_buf, int blocks, int teams, hipStream_t s) {
  if (count <= 0) return hipSuccess;
  if (teams < 1 || teams > TEAM_MAX_TEAMS) return hipErrorInvalidValue;
  blocks -= blocks % teams;
  if (blocks > TEAM_MAX_BLOCKS * teams) blocks = TEAM_MAX_BLOCKS * teams;
  if (blocks < teams) return hipErrorInvalidValue;
  // the persistent grid must be fully resident: one workgroup per CU, which the register budget must admit
  int per_cu = 0;
  hipError_t err = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, re_solve_team_kernel<TEAM_GRID_NW, true>,
                                                               WAVE * TEAM_GRID_NW, 0);
  if (err != hipSuccess) return err;
  if (per_cu < 1) return hipErrorLaunchOutOfResources;
  // the counters at the head of every team's TeamSync, in one call (a memset per team was 128 launches per tier)
  err = hipMemset2DAsync(sync_buf, sizeof(TeamSync), 0, 64, (size_t)teams, s);
  if (err != hipSuccess) return err;
  launch_sort_class(const_cast<int32_t*>(B.order) + begin, count, B.ent_nnz_ptr, s);
  // GDMIX_RE_XCD_BARRIER=0: every barrier with the full release (A/B and tests); default: measured per team (Team::team_placement)
  const char* xe = getenv("GDMIX_RE_XCD_BARRIER");
  const int xcd_barrier = (xe && atoi(xe) == 0) ? 0 : 1;
  int device = 0;
  (void)hipGetDevice(&device);
  ScopedGridGate gate(device, s);      // one persistent grid at a time per device, whatever context it comes from (re_internal.hpp)
  if (gate.status() != hipSuccess) return gate.status();
  hipLaunchKernelGGL((re_solve_team_kernel<TEAM_GRID_NW, true>), dim3(blocks), dim3(WAVE * TEAM_GRID_NW), 0, s, B, O, o,
                     theta0, begin, count, scratch, slot_doubles, max_p, static_cast<TeamSync*>(sync_buf), teams, 0, xcd_barrier);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// FULL variance: diag((X~' D X~ + (l2 + 1e-12) I - l2 e0 e0')^-1)   (binary_logistic_regression.py:181-187)
// One wavefront per entity, H and L^-1 in a global scratch slot (2 p^2 + p + n doubles). H is SPD, so the
// inverse comes from a Cholesky factor: diag(H^-1)_j = sum_i (L^-1)_ij^2 (the reference uses LU,
// np.linalg.inv; both agree to rounding on these well-conditioned matrices).
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void re_variance_full_kernel(BatchDev B, int64_t E, SolveParams o,
                                                               const double* __restrict__ theta,
                                                               double* __restrict__ variance, double* scratch,
                                                               size_t slot_doubles, int64_t max_p) {
  const int lane = threadIdx.x & (WAVE - 1);
  const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const int ic = o.has_intercept ? 1 : 0;
  double* slot = scratch + (size_t)wave0 * slot_doubles;
  for (int64_t e = wave0; e < E; e += nwaves) {
    const int64_t r0 = B.ent_row_ptr[e], z0 = B.ent_nnz_ptr[e], f0 = B.ent_feat_ptr[e];
    const int n = (int)(B.ent_row_ptr[e + 1] - r0);
    const int d = (int)(B.ent_feat_ptr[e + 1] - f0);
    const int p = d + ic;
    if (p > VAR_FULL_MAX_P) continue;         // wave-uniform: re_variance_big.hip takes these
    const int64_t c0 = f0 + e * ic;
    double* H = slot;                         // p x p, row-major; becomes L (lower)
    double* M = H + (size_t)max_p * max_p;    // p x p, L^-1
    double* xi = M + (size_t)max_p * max_p;   // p: dense row of X~
    const int32_t* rp = B.row_ptr + r0 + e;
    const double* th = theta + c0;
    for (int a = lane; a < p * p; a += WAVE) H[a] = 0.0;
    for (int j = lane; j < p; j += WAVE) xi[j] = 0.0;
    wave_mem_fence();
    for (int i = 0; i < n; ++i) {
      const int k0 = rp[i], k1 = rp[i + 1];
      // logit and D_i (every lane computes the same scalar)
      double acc = ic ? th[0] : 0.0;
      for (int k = k0; k < k1; ++k) acc += (double)B.csr_val[z0 + k] * th[ic + B.csr_col[z0 + k]];
      const double z = acc + (double)B.offset[r0 + i];
      const double rho = 1.0 / (1.0 + exp(-z));
      const double di = rho * (1.0 - rho) * (B.weight ? (double)B.weight[r0 + i] : 1.0);
      // dense row (duplicates summed, as toarray() does)
      if (lane == 0) {
        if (ic) xi[0] = 1.0;
        for (int k = k0; k < k1; ++k) xi[ic + B.csr_col[z0 + k]] += (double)B.csr_val[z0 + k];
      }
      wave_mem_fence();
      // H += di * xi xi' over the non-zero pattern of the row: entries = intercept + first occurrences
      const int m = (k1 - k0) + ic;
      for (int t = lane; t < m * m; t += WAVE) {
        const int ta = t / m, tb = t - ta * m;
        int ca = (ic && ta == 0) ? 0 : ic + B.csr_col[z0 + k0 + ta - ic];
        int cb = (ic && tb == 0) ? 0 : ic + B.csr_col[z0 + k0 + tb - ic];
        bool first = true;   // skip repeated occurrences of a column inside the row
        for (int k = k0; k < k0 + ta - ic; ++k) first &= (ic + B.csr_col[z0 + k] != ca) || (ic && ta == 0);
        for (int k = k0; k < k0 + tb - ic; ++k) first &= (ic + B.csr_col[z0 + k] != cb) || (ic && tb == 0);
        if (first) H[(size_t)ca * p + cb] += xi[ca] * di * xi[cb];
      }
      wave_mem_fence();
      if (lane == 0) {
        if (ic) xi[0] = 0.0;
        for (int k = k0; k < k1; ++k) xi[ic + B.csr_col[z0 + k]] = 0.0;
      }
      wave_mem_fence();
    }
    for (int j = lane; j < p; j += WAVE) {
      double add = o.l2 + 1.0e-12;
      if (j == 0 && ic && !o.regularize_bias) add -= o.l2;
      H[(size_t)j * p + j] += add;
    }
    wave_mem_fence();
    // right-looking Cholesky, lower triangle in place
    for (int j = 0; j < p; ++j) {
      const double ljj = sqrt(H[(size_t)j * p + j]);
      wave_mem_fence();
      for (int i = j + lane; i < p; i += WAVE) H[(size_t)i * p + j] = (i == j) ? ljj : H[(size_t)i * p + j] / ljj;
      wave_mem_fence();
      for (int i = j + 1 + lane; i < p; i += WAVE) {
        const double lij = H[(size_t)i * p + j];
        for (int k = j + 1; k <= i; ++k) H[(size_t)i * p + k] -= lij * H[(size_t)k * p + j];
      }
      wave_mem_fence();
    }
    // M = L^-1 column by column (lane per column), then diag(H^-1)_c = sum_i M[i][c]^2
    for (int c = lane; c < p; c += WAVE) {
      double ss = 0.0;
      for (int i = c; i < p; ++i) {
        double sum = (i == c) ? 1.0 : 0.0;
        for (int k = c; k < i; ++k) sum -= H[(size_t)i * p + k] * M[(size_t)k * p + c];
        const double mic = sum / H[(size_t)i * p + i];
        M[(size_t)i * p + c] = mic;
        ss += mic * mic;
      }
      variance[c0 + c] = ss;
    }
    wave_mem_fence();
  }
}

hipError_t launch_variance_full(const BatchDev& B, int64_t E, const SolveParams& o, const double* theta, double* variance,
                                double* scratch, size_t slot_doubles, int slots, int64_t max_p, hipStream_t s) {
  if (E <= 0) return hipSuccess;
  int waves = slots;
  if ((int64_t)waves > E) waves = (int)E;
  const int blocks = (waves + 3) / 4;
  // every wave of the grid owns one slot: grid = blocks * 4 waves <= slots is ensured by the caller
  hipLaunchKernelGGL(re_variance_full_kernel, dim3(blocks), dim3(256), 0, s, B, E, o, theta, variance, scratch,
                     slot_doubles, max_p);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// scoring: logits of every sample of the batch  (job_consumers.py:138-152, binary_logistic_regression.py:241-262)
// ---------------------------------------------------------------------------------------------------
// One thread per sample over the whole batch, whatever the entity sizes are (a Zipf partition has entities of one
// sample and of a million): a pure streaming pass, bound by HBM. The entity of a sample is found by bisection of
// ent_row_ptr — the wavefront's first and last sample by a 64-way search of the whole wavefront, each lane then within
// that range, which is a handful of entities or a single one. Adjacent lanes read adjacent rows, so the (value, column)
// loads of a wavefront fall on consecutive cache lines; the coefficient gathers of an entity stay in L1/L2.

// largest e in [lo, hi] with ptr[e] <= g (ptr[lo] <= g)
__device__ __forceinline__ int64_t entity_of_sample(const int64_t* __restrict__ ptr, int64_t lo, int64_t hi, int64_t g) {
  while (lo < hi) {
    const int64_t mid = (lo + hi + 1) >> 1;
    if (ptr[mid] <= g) lo = mid; else hi = mid - 1;
  }
  return lo;
}

// The same by a whole wavefront (uniform arguments, all lanes active): 64 probes per step instead of one, so a
// million entities take four dependent loads instead of twenty.
__device__ __forceinline__ int64_t wave_entity_of_sample(const int64_t* __restrict__ ptr, int64_t lo, int64_t hi, int64_t g, int lane) {
  while (lo < hi) {
    const int64_t step = (hi - lo + WAVE - 1) / WAVE;
    const int64_t probe = lo + (int64_t)(lane + 1) * step;
    const bool le = ptr[probe < hi ? probe : hi] <= g;     // non-decreasing in the lane index
    const int c = __popcll(__ballot(le));
    const int64_t below = lo + (int64_t)c * step;           // last probe that is <= g (lo itself when c == 0)
    const int64_t above = lo + (int64_t)(c + 1) * step;     // first probe that is > g
    const int64_t nhi = (c < WAVE && above <= hi) ? above - 1 : hi;
    lo = below < hi ? below : hi;
    hi = (c == WAVE) ? lo : nhi;
  }
  return lo;
}

__global__ __launch_bounds__(256) void re_score_kernel(BatchDev B, int64_t E, int64_t N, int ic, const double* __restrict__ theta,
                                                       const uint8_t* __restrict__ has_model,
                                                       float* __restrict__ logit, float* __restrict__ per_coord) {
  const int lane = threadIdx.x & (WAVE - 1);
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t gf = g - lane;
  if (gf >= N) return;
  const int64_t gl = (gf + WAVE - 1 < N) ? gf + WAVE - 1 : N - 1;
  const int64_t e_lo = wave_entity_of_sample(B.ent_row_ptr, 0, E - 1, gf, lane);
  const int64_t w_hi = (e_lo + WAVE < E) ? e_lo + WAVE : E - 1;   // 64 samples span at most 64 non-empty entities
  const int64_t e_hi = wave_entity_of_sample(B.ent_row_ptr, e_lo, (B.ent_row_ptr[w_hi] > gl) ? w_hi : E - 1, gl, lane);
  if (g >= N) return;
  const int64_t e = entity_of_sample(B.ent_row_ptr, e_lo, e_hi, g);
  const int64_t r0 = B.ent_row_ptr[e], z0 = B.ent_nnz_ptr[e];
  const double off = (double)B.offset[g];
  const bool model = has_model ? has_model[e] != 0 : true;
  double z = off;
  if (model) {
    const int64_t c0 = B.ent_feat_ptr[e] + e * ic;
    const int32_t* rp = B.row_ptr + r0 + e + (g - r0);
    const int k0 = rp[0], k1 = rp[1];
    const float* __restrict__ val = B.csr_val + z0;
    const int32_t* __restrict__ col = B.csr_col + z0;
    const double* __restrict__ th = theta + c0 + ic;
    double acc = ic ? theta[c0] : 0.0;
    int k = k0;
    for (; k + 4 <= k1; k += 4) {   // four gathers in flight; the sum keeps the row's order
      const float v0 = val[k], v1 = val[k + 1], v2 = val[k + 2], v3 = val[k + 3];
      const double t0 = th[col[k]], t1 = th[col[k + 1]], t2 = th[col[k + 2]], t3 = th[col[k + 3]];
      acc += (double)v0 * t0;
      acc += (double)v1 * t1;
      acc += (double)v2 * t2;
      acc += (double)v3 * t3;
    }
    for (; k < k1; ++k) acc += (double)val[k] * th[col[k]];
    z = acc + off;
  }
  logit[g] = (float)z;
  per_coord[g] = (float)(z - off);
}

hipError_t launch_score(const BatchDev& B, int64_t E, int64_t N, int ic, const double* theta, const uint8_t* has_model,
                        float* logit, float* per_coord, hipStream_t s) {
  if (E <= 0 || N <= 0) return hipSuccess;
  const int64_t blocks = (N + 255) / 256;
  if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
  hipLaunchKernelGGL(re_score_kernel, dim3((unsigned)blocks), dim3(256), 0, s, B, E, N, ic, theta, has_model, logit, per_coord);
  return hipGetLastError();
}

}  // namespace gdmix
