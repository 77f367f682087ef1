// re_solve_team.hpp — the solve for entities too large for one wavefront group: a TEAM of workgroups
// (one workgroup for the workgroup-per-entity class, every CU of the device for a giant) works on one
// entity with X, y and the L-BFGS vectors in HBM.
//
// What bounds these entities is not arithmetic but the number of team-wide synchronisation points and
// of dependent passes over p-vectors, so the limited-memory direction is evaluated in the compact
// (Byrd-Nocedal-Schnabel) form that L-BFGS-B itself uses (lbfgsb.f bmv/formk)
// instead of the two-loop recursion of the register kernels:
//     d = -H g,   H = gamma I + [S  gamma Y] [ R^-T (D + gamma Y'Y) R^-1   -R^-T ] [ S'       ]
//                                            [ -R^-1                        0    ] [ gamma Y' ]
//     R = triu(S'Y), D = diag(S'Y), gamma = 1/theta = s'y / y'y of the newest pair.
// All it needs from the p-vectors are the 2m dot products S'g and Y'g, which are accumulated in the same
// pass that forms g (one fused reduction of 2m+6 numbers); new rows of S'Y and Y'Y follow from the
// difference of these products at consecutive iterates. One objective evaluation then costs three
// team synchronisations (x ready -> rows -> columns+dots -> decision) whatever m is, and the direction
// is one elementwise pass. Same driver rules as the other kernels (scipy's loop around setulb, MINPACK-2
// dcsrch, pair acceptance, restart on line search failure: re_solve_core.hpp lbfgs_advance); results
// agree with them to rounding (different summation order), which tests/test_gpu_parity.py pins against
// the reference fixtures through the workgroup class.
#pragma once
#include "re_solve_core.hpp"

namespace gdmix {

constexpr int TEAM_MCAP = 10;                 // history pairs the compact path keeps accumulators for
constexpr int TEAM_K = 2 * TEAM_MCAP + 6;     // fused reduction width: sq, gd, gg, rr, gr, S'g, Y'g, max|g|
constexpr int TEAM_VEC = 32;                  // doubles per workgroup slot of the device-wide exchange buffer
constexpr int TEAM_MAX_BLOCKS = 256;           // workgroups per team
constexpr int TEAM_MAX_TEAMS = 32;
constexpr int TEAM_SHORT_COL = 16;            // tiles whose columns are all this short: one lane per column

// Exchange buffer of a multi-workgroup team (HBM).
struct TeamSync {
  unsigned count;    // barrier arrivals, monotonic inside a launch (zeroed by a memset node before it)
  unsigned abort;    // a workgroup gave up waiting (never expected; keeps a lost workgroup from hanging the GPU)
  unsigned pad[14];
  double vec[2][TEAM_MAX_BLOCKS][TEAM_VEC];
};

// LDS of one workgroup of a team.
template <int NW>
struct TeamLds {
  double red[2][NW][TEAM_VEC];
  double out[2][TEAM_VEC];
  double SY[TEAM_MCAP * TEAM_MCAP];   // s_i'y_k, chronological, i <= k used
  double YY[TEAM_MCAP * TEAM_MCAP];   // y_i'y_k
  double ap[TEAM_MCAP], bp[TEAM_MCAP];   // S'g_k, Y'g_k at the last accepted iterate
  double la[TEAM_MCAP], lb[TEAM_MCAP];
  double u[TEAM_MCAP], q[TEAM_MCAP];
  double sc[4];
};

template <int NW>
struct Team {
  int tid, NT;            // thread index / thread count of the whole team
  int wid, nwaves, lane;  // wavefront index / count of the whole team
  unsigned nblocks, bid;  // workgroups in the team, this workgroup's index
  TeamSync* gs;           // used when nblocks > 1
  TeamLds<NW>* L;
  unsigned epoch;
  int phase;

  __device__ __forceinline__ void block_sync() const { __syncthreads(); }

  // Device-wide barrier with release/acquire of everything written before it: the XCDs' L2s are not
  // coherent with each other and a CU's L1 is not refreshed by other CUs' stores, so one lane per workgroup
  // writes the XCD's dirty L2 lines back before arriving and invalidates this CU's L1 after the wait.
  __device__ __forceinline__ void device_barrier() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    ++epoch;
    if (threadIdx.x == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_fetch_add(&gs->count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned target = epoch * nblocks;
      unsigned spins = 0;
      uint64_t t0 = 0;
      while (__hip_atomic_load(&gs->count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(1);
        if ((++spins & 255u) == 0) {
          if (__hip_atomic_load(&gs->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
          const uint64_t now = wall_clock64();
          if (t0 == 0) t0 = now;
          else if (now - t0 > 400000000ull) {   // 4 s at 100 MHz
            __hip_atomic_store(&gs->abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
  }
  __device__ __forceinline__ void sync() {
    if (nblocks > 1) device_barrier();
    else __syncthreads();
  }
  __device__ __forceinline__ bool aborted() const {
    return nblocks > 1 && __hip_atomic_load(&gs->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
  }

  // Team-wide reduction of K numbers held by every thread: sums of v[0..K-2], maximum (of non-negative
  // values) of v[K-1]; every thread of the team gets the same totals. Fixed shape: lanes (DPP) -> waves of a
  // workgroup in index order -> workgroups strided over lanes, then the lane reduction again. Orders all
  // memory operations of the team like sync().
  template <int K>
  __device__ __forceinline__ void reduce(double (&v)[K]) {
    static_assert(K <= TEAM_VEC, "reduction wider than the exchange slot");
    const int ph = phase;
    phase ^= 1;
    double mine = 0.0;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const double t = (k == K - 1) ? wave_max_nonneg(v[k]) : wave_sum(v[k]);
      if (lane == k) mine = t;
    }
    const int w = threadIdx.x >> 6;
    if (lane < K) L->red[ph][w][lane] = mine;
    __syncthreads();
    if (threadIdx.x < K) {
      double s = L->red[ph][0][threadIdx.x];
#pragma unroll
      for (int ww = 1; ww < NW; ++ww) {
        const double t = L->red[ph][ww][threadIdx.x];
        s = (threadIdx.x == K - 1) ? fmax(s, t) : s + t;
      }
      if (nblocks > 1) gs->vec[ph][bid][threadIdx.x] = s;
      else L->out[ph][threadIdx.x] = s;
    }
    if (nblocks > 1) {
      device_barrier();
      for (int k = w; k < K; k += NW) {
        double s = 0.0;
        for (unsigned b = lane; b < nblocks; b += WAVE) {
          const double t = gs->vec[ph][b][k];
          s = (k == K - 1) ? fmax(s, t) : s + t;
        }
        s = (k == K - 1) ? wave_max_nonneg(s) : wave_sum(s);
        if (lane == 0) L->out[ph][k] = s;
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = L->out[ph][k];
  }
};

__device__ __forceinline__ int readlane_i(int v, int l) { return __builtin_amdgcn_readlane(v, l); }

// acc + sum_{k in [k0,k1)} val[k] * vec[idx[k]], added in index order. These entities stream from HBM/L2, where
// a dependent (index -> gather) chain costs two memory latencies: eight entries are loaded, then gathered, at a
// time, so a typical row (or short column) costs two latencies instead of two per entry.
__device__ __forceinline__ double gather_dot8(const float* __restrict__ val, const int32_t* __restrict__ idx,
                                              const double* __restrict__ vec, int k0, int k1, double acc) {
  for (int k = k0; k < k1; k += 8) {
    int c[8];
    float v[8];
    double xv[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const bool ok = k + q < k1;
      c[q] = ok ? idx[k + q] : 0;
      v[q] = ok ? val[k + q] : 0.0f;
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) xv[q] = (k + q < k1) ? vec[c[q]] : 0.0;
#pragma unroll
    for (int q = 0; q < 8; ++q)
      if (k + q < k1) acc += (double)v[q] * xv[q];
  }
  return acc;
}

// f, g and every dot product the driver needs, at W.x. acc[] layout: 0 sum x_j^2 over regularised j,
// 1 g'd, 2 g'g, 3 (g-r)'(g-r), 4 g'r, 5.. S_i'g, 5+MCAP.. Y_i'g (chronological i < col), K-1 max|g_j|.
template <int NW>
__device__ __forceinline__ double team_eval(Team<NW>& tm, const EntityView& P, const SolveParams& o, const Work& W,
                                            int col, int head, double (&acc)[TEAM_K]) {
  const int n = P.n, p = P.p, ic = P.ic, m = o.m;
  const double* __restrict__ x = W.x;
  // ---- rows: logits, per-sample loss and residual
  double pr[3] = {0.0, 0.0, 0.0};
  const double x0 = ic ? x[0] : 0.0;
  for (int i = tm.tid; i < n; i += tm.NT) {
    const double a = gather_dot8(P.csr_val, P.csr_col, x + ic, P.row_ptr[i], P.row_ptr[i + 1], x0);
    const double z = a + (double)P.o[i];
    const double yi = (double)P.y[i];
    const double wi = P.w ? (double)P.w[i] : 1.0;
    double ri;
    pr[0] += logistic_terms(z, yi, wi, ri);
    W.rs[i] = ri;
    pr[1] += ri;
  }
  tm.reduce(pr);   // also makes rs[] visible to the whole team
  const double loss = pr[0], rsum = pr[1];
  // ---- columns: X'r by tiles of 64 coefficients per wavefront, lane c of the tile ends up owning column c
  const int first_reg = (ic && !o.regularize_bias) ? 1 : 0;
  const double inv_n = 1.0 / (double)n;
#pragma unroll
  for (int k = 0; k < TEAM_K; ++k) acc[k] = 0.0;
  for (int tile = tm.wid; tile * WAVE < p; tile += tm.nwaves) {
    const int j = tile * WAVE + tm.lane;
    const bool valid = j < p;
    const bool feat = valid && !(ic && j == 0);
    int cb = 0, ce = 0;
    if (feat) { cb = P.col_ptr[j - ic]; ce = P.col_ptr[j - ic + 1]; }
    const int maxlen = (int)wave_max_nonneg((double)(ce - cb));
    double mine = 0.0;
    if (maxlen <= TEAM_SHORT_COL) {
      mine = gather_dot8(P.csc_val, P.csc_row, W.rs, cb, ce, 0.0);
    } else {
      // eight columns at a time, lanes striding each of them: one dependent gather chain per step
      const int ncol = (p - tile * WAVE) < WAVE ? (p - tile * WAVE) : WAVE;
      for (int c = 0; c < ncol; c += 8) {
        int b[8], len = 0;
        double s[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          b[q] = readlane_i(cb, c + q);
          const int l = readlane_i(ce, c + q) - b[q];
          len = l > len ? l : len;
          s[q] = 0.0;
        }
        for (int off = tm.lane; off < len; off += WAVE) {
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const int k = b[q] + off;
            if (k < readlane_i(ce, c + q)) s[q] += (double)P.csc_val[k] * W.rs[P.csc_row[k]];
          }
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const double t = wave_sum(s[q]);
          if (tm.lane == c + q) mine = t;
        }
      }
    }
    if (valid) {
      const double a = (ic && j == 0) ? rsum : mine;
      const double xj = x[j];
      const double gj = inv_n * (a + ((j < first_reg) ? 0.0 : o.l2 * xj));
      W.g[j] = gj;
      const double dj = W.d[j], rj = W.r[j];
      if (j >= first_reg) acc[0] += xj * xj;
      acc[1] += gj * dj;
      acc[2] += gj * gj;
      const double yj = gj - rj;
      acc[3] += yj * yj;
      acc[4] += gj * rj;
      acc[TEAM_K - 1] = fmax(acc[TEAM_K - 1], fabs(gj));
#pragma unroll
      for (int i = 0; i < TEAM_MCAP; ++i) {
        if (i < col) {
          int sl = head + i;
          if (sl >= m) sl -= m;
          acc[5 + i] += W.ws[(size_t)sl * p + j] * gj;
          acc[5 + TEAM_MCAP + i] += W.wy[(size_t)sl * p + j] * gj;
        }
      }
    }
  }
  tm.reduce(acc);
  return inv_n * (loss + 0.5 * o.l2 * acc[0]);
}

// The whole fmin_l_bfgs_b run for one entity by a team. Requires 1 <= o.m <= TEAM_MCAP. W.x holds theta0 on
// entry (visible to the team), theta on exit.
template <int NW>
__device__ void team_solve(Team<NW>& tm, const EntityView& P, const SolveParams& o, const Work& W, SolveStats& out) {
  const int p = P.p, m = o.m;
  TeamLds<NW>& L = *tm.L;
  int col = 0, head = 0, nit = 0, nfev = 0, ifun = 0, status = -1;
  bool first = true, iter0 = true;
  double theta = 1.0, f = 0.0, fold = 0.0, gdold = 0.0, stp = 0.0, sbgnrm = 0.0, gg_k = 0.0;
  LineSearch ls;
  double acc[TEAM_K];
  for (int j = tm.tid; j < p; j += tm.NT) { W.d[j] = 0.0; W.r[j] = 0.0; }
  tm.sync();
  for (;;) {
    const double f_new = team_eval(tm, P, o, W, col, head, acc);
    ++nfev;
    if (tm.aborted()) { status = GDMIX_RE_ST_ABORTED; break; }
    const double gd = acc[1], gg = acc[2], rr = acc[3], gr = acc[4];
    bool restore = false, store_pair = false, shift = false;
    double dr = 0.0;
    const double stp_prev = stp;
    if (first) {
      first = false;
      f = f_new;
      sbgnrm = acc[TEAM_K - 1];
      if (sbgnrm <= o.pgtol) { status = 0; break; }
    } else {
      f = f_new;
      const int task = dcsrch_step(ls, f_new, gd, stp);
      if (task == LS_FG) {
        ++ifun;
        if (ifun - 1 < o.maxls) {
          for (int j = tm.tid; j < p; j += tm.NT) W.x[j] = stp * W.d[j] + W.t[j];
          tm.sync();
          continue;
        }
        restore = true;   // iback >= maxls: back to the last iterate, forget the history
      } else {
        ++nit;
        iter0 = false;
        sbgnrm = acc[TEAM_K - 1];
        if (nit >= o.max_iter) { status = 2; break; }
        if (nfev > o.maxfun) { status = 3; break; }
        if (sbgnrm <= o.pgtol) { status = 0; break; }
        {
          const double ddum = fmax(fabs(fold), fmax(fabs(f), 1.0));
          if (fold - f <= o.ftol * ddum) { status = 1; break; }
        }
        double ddum;
        if (stp == 1.0) { dr = gd - gdold; ddum = -gdold; }
        else { dr = (gd - gdold) * stp; ddum = -gdold * stp; }
        store_pair = dr > EPSMCH * ddum;
      }
    }
    // ---- new search direction
    int slot = 0;          // history slot of the pair being stored
    double gg_cur = gg;    // g'g of the gradient the direction is built from
    if (restore) {
      if (col == 0) {
        for (int j = tm.tid; j < p; j += tm.NT) W.x[j] = W.t[j];
        tm.sync();
        f = fold;
        status = 4;
        break;
      }
      col = 0; head = 0; theta = 1.0;
      f = fold;
      gg_cur = gg_k;
    }
    if (store_pair) {
      if (col < m) { slot = head + col; if (slot >= m) slot -= m; ++col; }
      else { slot = head; ++head; if (head >= m) head = 0; shift = true; }
      theta = rr / dr;
    }
    const int cnew = col - 1;   // chronological index of the stored pair
    if (threadIdx.x == 0) {
      if (store_pair) {
        if (shift) {   // drop the oldest pair
          for (int i = 0; i + 1 < m; ++i) {
            for (int k = 0; k + 1 < m; ++k) {
              L.SY[i * TEAM_MCAP + k] = L.SY[(i + 1) * TEAM_MCAP + k + 1];
              L.YY[i * TEAM_MCAP + k] = L.YY[(i + 1) * TEAM_MCAP + k + 1];
            }
            L.ap[i] = L.ap[i + 1];
            L.bp[i] = L.bp[i + 1];
          }
        }
      }
      // current S'g, Y'g in chronological order after the shift; products with the new pair in closed form
#pragma unroll
      for (int i = 0; i < TEAM_MCAP; ++i) {
        const int src = shift ? i + 1 : i;
        if (src < TEAM_MCAP) { L.la[i] = acc[5 + src]; L.lb[i] = acc[5 + TEAM_MCAP + src]; }
      }
      if (store_pair) {
        for (int i = 0; i < cnew; ++i) {
          L.SY[i * TEAM_MCAP + cnew] = L.la[i] - L.ap[i];          // s_i'(g - g_k)
          const double yy = L.lb[i] - L.bp[i];                     // y_i'(g - g_k)
          L.YY[i * TEAM_MCAP + cnew] = yy;
          L.YY[cnew * TEAM_MCAP + i] = yy;
        }
        L.SY[cnew * TEAM_MCAP + cnew] = dr;
        L.YY[cnew * TEAM_MCAP + cnew] = rr;
        L.la[cnew] = stp_prev * gd;      // s'g,  s = stp d
        L.lb[cnew] = gg - gr;            // y'g,  y = g - g_k
      }
      const double gamma = 1.0 / theta;
      // q = R^-1 a
      for (int i = col - 1; i >= 0; --i) {
        double s = L.la[i];
        for (int k = i + 1; k < col; ++k) s -= L.SY[i * TEAM_MCAP + k] * L.q[k];
        L.q[i] = s / L.SY[i * TEAM_MCAP + i];
      }
      // u = R^-T ((D + gamma Y'Y) q - gamma b)
      for (int i = 0; i < col; ++i) {
        double s = L.SY[i * TEAM_MCAP + i] * L.q[i] - gamma * L.lb[i];
        for (int k = 0; k < col; ++k) s += gamma * L.YY[i * TEAM_MCAP + k] * L.q[k];
        for (int k = 0; k < i; ++k) s -= L.SY[k * TEAM_MCAP + i] * L.u[k];
        L.u[i] = s / L.SY[i * TEAM_MCAP + i];
      }
      double gdn = -gg_cur * (col > 0 ? gamma : 1.0);
      for (int i = 0; i < col; ++i) gdn += gamma * L.lb[i] * L.q[i] - L.la[i] * L.u[i];
      L.sc[0] = gdn;
      for (int i = 0; i < col; ++i) { L.ap[i] = L.la[i]; L.bp[i] = L.lb[i]; }
    }
    tm.block_sync();
    double gdn = L.sc[0];
    if (gdn >= 0.0) {   // not a descent direction (lnsrlb info = -4): steepest descent without history
      if (col == 0) { status = 4; break; }
      col = 0; head = 0; theta = 1.0;
      store_pair = false;
      gdn = -gg_cur;
    }
    gg_k = gg_cur;
    gdold = gdn;
    fold = f;
    stp = iter0 ? fmin(1.0 / sqrt(gg_cur), LS_STPMAX) : 1.0;
    dcsrch_start(ls, f, gdn, stp);
    ifun = 1;
    {
      const double gamma = 1.0 / theta;
      for (int j = tm.tid; j < p; j += tm.NT) {
        const double gj = restore ? W.r[j] : W.g[j];
        const double xj = restore ? W.t[j] : W.x[j];
        double sn = 0.0, yn = 0.0;
        if (store_pair) {
          sn = stp_prev * W.d[j];   // exact for stp == 1
          yn = W.g[j] - W.r[j];
          W.ws[(size_t)slot * p + j] = sn;
          W.wy[(size_t)slot * p + j] = yn;
        }
        double dj = -gj;
        if (col > 0) {
          double su = 0.0, yq = 0.0;
#pragma unroll
          for (int i = 0; i < TEAM_MCAP; ++i) {
            if (i < col) {
              int sl = head + i;
              if (sl >= m) sl -= m;
              const bool fresh = store_pair && i == cnew;
              const double si = fresh ? sn : W.ws[(size_t)sl * p + j];
              const double yi = fresh ? yn : W.wy[(size_t)sl * p + j];
              su += L.u[i] * si;
              yq += L.q[i] * yi;
            }
          }
          dj = gamma * (yq - gj) - su;
        }
        const double z = xj + dj;   // mainlb re-derives d from the subspace point
        dj = z - xj;
        W.d[j] = dj;
        W.t[j] = xj;
        W.r[j] = gj;
        W.x[j] = stp * dj + xj;
      }
    }
    tm.sync();
  }
  out.f = f;
  out.gnorm = sbgnrm;
  out.nit = nit;
  out.nfev = nfev;
  out.status = status;
}

// The team seen through the thread-group interface of re_solve_core.hpp (variance_simple, result epilogue).
template <int NW>
struct TeamAsGroup {
  Team<NW>& tm;
  int tid, NT;
  __device__ __forceinline__ double sum(double v) {
    double a[3] = {v, 0.0, 0.0};
    tm.reduce(a);
    return a[0];
  }
  __device__ __forceinline__ void sync() { tm.sync(); }
};

}  // namespace gdmix
