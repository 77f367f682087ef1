// re_solve_core.hpp — the per-entity L-BFGS solve, written once against a thread-group policy
// (WaveGroup: one wavefront per entity, state in LDS; BlockGroup: one workgroup per entity, state in
// a global scratch slot). Follows, step for step, what the reference executes per entity:
//   fit()                      gdmix-trainer/src/gdmix/models/custom/binary_logistic_regression.py:191-239
//   _loss / _gradient          :84-110 / :121-131   (fused here into one pass, one X~theta)
//   scipy fmin_l_bfgs_b        L-BFGS-B 3.0 unconstrained branch + scipy's python driver loop
//                              (restated on the CPU in oracle/re_oracle.c)
#pragma once
#include "re_device.hpp"

namespace gdmix {

struct SolveParams {
  double l2, ftol, pgtol, threshold;
  int regularize_bias, has_intercept, m, max_iter, maxfun, maxls, variance_mode;
  int sum_loss, linear;   // fixed-effect objective (team kernels only)
};

// One entity's data, pointers into LDS (wave kernel) or HBM (block kernel).
struct EntityView {
  int n, d, p, ic;
  const int32_t* row_ptr;   // [n+1]
  const int32_t* csr_col;   // [nnz] local
  const float* csr_val;
  const int32_t* col_ptr;   // [d+1]
  const int32_t* csc_row;   // [nnz]
  const float* csc_val;
  const float* y;
  const float* o;
  const float* w;           // may be nullptr
};

// Working vectors of one solve (LDS or global scratch).
struct Work {
  double* x;    // [p] current / trial point (holds theta0 on entry, theta on exit)
  double* g;    // [p]
  double* d;    // [p]
  double* t;    // [p] x_old
  double* r;    // [p] g_old, then y
  double* ws;   // [m*p] s history (slot-major)
  double* wy;   // [m*p] y history
  double* rs;   // [n] per-sample residual w*(sigma(z)-y)
  double* alpha;  // [m] two-loop coefficients (uniform; every thread stores the same value)
  double* rho;    // [m] 1/(s'y) per history slot (uniform)
  double* part;   // team kernels: [TEAM_LONG_CAP * 64] partial sums of split columns
};

constexpr double EPSMCH = 2.220446049250313e-16;

// f and g at W.x. Returns f; leaves g in W.g. One pass over the CSR copy for the logits, one pass over
// the CSC copy for X'r: both are ordered sums, no atomics.
template <class G>
__device__ __forceinline__ double eval_fg(G& grp, const EntityView& P, const SolveParams& o, const Work& W) {
  const int n = P.n, p = P.p, ic = P.ic;
  const double* __restrict__ x = W.x;
  double part = 0.0;   // per-thread partial of sum_i w_i*ce_i + (l2/2) * sum_j x_j^2
  double rpart = 0.0;  // per-thread partial of sum_i r_i (the intercept's gradient)
  const double x0 = ic ? x[0] : 0.0;
  for (int i = grp.tid; i < n; i += grp.NT) {
    double acc = x0;
    const int k1 = P.row_ptr[i + 1];
    for (int k = P.row_ptr[i]; k < k1; ++k) acc += (double)P.csr_val[k] * x[ic + P.csr_col[k]];
    const double z = acc + (double)P.o[i];
    const double yi = (double)P.y[i];
    const double wi = P.w ? (double)P.w[i] : 1.0;
    double ri;
    part += logistic_terms(z, yi, wi, ri);
    W.rs[i] = ri;
    rpart += ri;
  }
  const int first_reg = (ic && !o.regularize_bias) ? 1 : 0;
  double sq = 0.0;
  for (int j = first_reg + grp.tid; j < p; j += grp.NT) sq += x[j] * x[j];
  const double inv_n = 1.0 / (double)n;
  // cost.sum() + regulariser, added once (binary_logistic_regression.py:105-108)
  const double cost = grp.sum(part);
  const double f = inv_n * (cost + 0.5 * o.l2 * grp.sum(sq));
  const double rsum = grp.sum(rpart);   // also orders the rs[] writes before the reads below (block)
  grp.sync();
  for (int j = grp.tid; j < p; j += grp.NT) {
    double acc;
    if (ic && j == 0) {
      acc = rsum;
    } else {
      acc = 0.0;
      const int c = j - ic;
      const int k1 = P.col_ptr[c + 1];
      for (int k = P.col_ptr[c]; k < k1; ++k) acc += (double)P.csc_val[k] * W.rs[P.csc_row[k]];
    }
    const double reg = (j < first_reg) ? 0.0 : o.l2 * x[j];
    W.g[j] = inv_n * (acc + reg);
  }
  grp.sync();
  return f;
}

template <class G>
__device__ __forceinline__ double dot(G& grp, const double* a, const double* b, int p) {
  double s = 0.0;
  for (int j = grp.tid; j < p; j += grp.NT) s += a[j] * b[j];
  return grp.sum(s);
}

template <class G>
__device__ __forceinline__ double maxabs(G& grp, const double* a, int p) {
  double s = 0.0;
  for (int j = grp.tid; j < p; j += grp.NT) s = fmax(s, fabs(a[j]));
  return grp.max_nonneg(s);
}

struct SolveStats {
  double f, gnorm;
  int nit, nfev, status;
};

// Resumable state of one fmin_l_bfgs_b run (all values uniform across the cooperating threads). The
// workgroup kernel keeps it in registers; the device-wide path for giant entities keeps it in HBM between
// its kernels.
struct LbfgsState {
  LineSearch ls;
  double f, fold, gdold, stp, theta, sbgnrm;
  int col, head, nit, nfev, ifun, status, first, iter0;
  int counted, failed_at_t;   // nfev is scipy's funcalls (re_solve_quad.hpp, quad_solve): does the next evaluation count?
};

__device__ __forceinline__ void lbfgs_init(LbfgsState& T) {
  T.f = 0.0; T.fold = 0.0; T.gdold = 0.0; T.stp = 0.0; T.theta = 1.0; T.sbgnrm = 0.0;
  T.col = 0; T.head = 0; T.nit = 0; T.nfev = 0; T.ifun = 0; T.status = -1; T.first = 1; T.iter0 = 1;
  T.counted = 1; T.failed_at_t = 1;
}

// One step of the driver: f_new and W.g are the objective and gradient at the trial point W.x. Advances the
// algorithm until it needs another evaluation (next trial point left in W.x, T.status < 0) or stops
// (T.status >= 0, result in W.x). Same rules as scipy's loop + L-BFGS-B's mainlb/lnsrlb.
// Synchronisation: element j of every p-vector is only ever touched by thread j mod NT; W.x is synced before
// returning because the evaluation gathers it across threads.
template <class G>
__device__ void lbfgs_advance(G& grp, int p, const SolveParams& o, const Work& W, LbfgsState& T, double f_new) {
  const int m = o.m;
  double* alpha = W.alpha;
  double* rho = W.rho;
  T.nfev += T.counted ? 1 : 0;
  bool need_dir = false, restart = false;
  if (T.first) {
    T.first = 0;
    T.f = f_new;
    T.sbgnrm = maxabs(grp, W.g, p);
    if (T.sbgnrm <= o.pgtol) { T.status = 0; return; }
    need_dir = true;
  } else {
    T.f = f_new;
    const double gd = dot(grp, W.g, W.d, p);
    double stp = T.stp;
    const int task = dcsrch_step(T.ls, f_new, gd, stp);
    T.stp = stp;
    if (task == LS_FG) {
      ++T.ifun;
      if (T.ifun - 1 < o.maxls) {
        double mv = 0.0;
        for (int j = grp.tid; j < p; j += grp.NT) {
          const double xn = stp * W.d[j] + W.t[j];   // stp == 1: exactly t + d
          mv = (xn != W.x[j]) ? 1.0 : mv;
          W.x[j] = xn;
        }
        T.counted = grp.max_nonneg(mv) != 0.0;
        grp.sync();
        return;
      }
      restart = true;   // iback >= maxls
      need_dir = true;
    } else {
      // NEW_X: scipy's python loop first (nit / maxiter / maxfun), then mainlb's own tests
      ++T.nit;
      T.iter0 = 0;
      T.sbgnrm = maxabs(grp, W.g, p);
      if (T.nit >= o.max_iter) { T.status = 2; return; }
      if (T.nfev > o.maxfun) { T.status = 3; return; }
      if (T.sbgnrm <= o.pgtol) { T.status = 0; return; }
      {
        const double ddum = fmax(fabs(T.fold), fmax(fabs(T.f), 1.0));
        if (T.fold - T.f <= o.ftol * ddum) { T.status = 1; return; }
      }
      // pair update: s = stp*d, y = g - g_old; skipped when s'y <= epsmch * (-g_old'd*stp)
      double rrp = 0.0;
      for (int j = grp.tid; j < p; j += grp.NT) {
        const double yj = W.g[j] - W.r[j];
        W.r[j] = yj;
        rrp += yj * yj;
      }
      const double rr = grp.sum(rrp);
      double dr, ddum;
      if (stp == 1.0) { dr = gd - T.gdold; ddum = -T.gdold; }
      else { dr = (gd - T.gdold) * stp; ddum = -T.gdold * stp; }
      if (dr > EPSMCH * ddum) {
        int sl;
        if (T.col < m) { sl = T.head + T.col; if (sl >= m) sl -= m; ++T.col; }
        else { sl = T.head; ++T.head; if (T.head >= m) T.head = 0; }
        double* s = W.ws + (size_t)sl * p;
        double* yv = W.wy + (size_t)sl * p;
        for (int j = grp.tid; j < p; j += grp.NT) {
          s[j] = stp * W.d[j];   // exact for stp == 1
          yv[j] = W.r[j];
        }
        rho[sl] = 1.0 / dr;
        T.theta = rr / dr;
      }
      need_dir = true;
    }
  }
  while (need_dir) {
    if (restart) {
      double off = 0.0;   // the abandoned search's last trial: had it moved away from the iterate?
      for (int j = grp.tid; j < p; j += grp.NT) { off = (W.x[j] != W.t[j]) ? 1.0 : off; W.x[j] = W.t[j]; W.g[j] = W.r[j]; }
      T.failed_at_t = T.failed_at_t && (grp.max_nonneg(off) == 0.0);
      T.f = T.fold;
      restart = false;
      if (T.col == 0) { T.status = 4; grp.sync(); return; }
      T.col = 0; T.head = 0; T.theta = 1.0;
    }
    // direction: two-loop recursion over the stored pairs, H0 = (1/theta) I
    for (int j = grp.tid; j < p; j += grp.NT) W.d[j] = -W.g[j];
    if (T.col > 0) {
      for (int a = T.col - 1; a >= 0; --a) {
        int sl = T.head + a; if (sl >= m) sl -= m;
        const double* s = W.ws + (size_t)sl * p;
        const double* yv = W.wy + (size_t)sl * p;
        const double al = rho[sl] * dot(grp, s, W.d, p);
        alpha[a] = al;
        for (int j = grp.tid; j < p; j += grp.NT) W.d[j] -= al * yv[j];
      }
      const double h0 = 1.0 / T.theta;
      for (int j = grp.tid; j < p; j += grp.NT) W.d[j] *= h0;
      for (int a = 0; a < T.col; ++a) {
        int sl = T.head + a; if (sl >= m) sl -= m;
        const double* s = W.ws + (size_t)sl * p;
        const double* yv = W.wy + (size_t)sl * p;
        const double beta = rho[sl] * dot(grp, yv, W.d, p);
        const double c = alpha[a] - beta;
        for (int j = grp.tid; j < p; j += grp.NT) W.d[j] += c * s[j];
      }
    }
    // z = x + d ; d = z - x  (mainlb re-derives d from the subspace point); save x, g
    double dd = 0.0, gdp = 0.0;
    for (int j = grp.tid; j < p; j += grp.NT) {
      const double xj = W.x[j];
      const double z = xj + W.d[j];
      const double dj = z - xj;
      W.d[j] = dj;
      W.t[j] = xj;
      const double gj = W.g[j];
      W.r[j] = gj;
      dd += dj * dj;
      gdp += gj * dj;
    }
    const double dnorm = sqrt(grp.sum(dd));
    const double gd = grp.sum(gdp);
    T.gdold = gd;
    T.fold = T.f;
    if (gd >= 0.0) { restart = true; continue; }   // lnsrlb info = -4
    const double stp = T.iter0 ? fmin(1.0 / dnorm, LS_STPMAX) : 1.0;
    T.stp = stp;
    dcsrch_start(T.ls, T.f, gd, stp);
    T.ifun = 1;
    double mv = 0.0;
    for (int j = grp.tid; j < p; j += grp.NT) {
      const double xn = stp * W.d[j] + W.t[j];
      mv = (xn != W.t[j]) ? 1.0 : mv;
      W.x[j] = xn;
    }
    T.counted = (grp.max_nonneg(mv) != 0.0) || !T.failed_at_t;
    T.failed_at_t = 1;
    grp.sync();
    need_dir = false;
  }
}

// The whole fmin_l_bfgs_b run for one entity (wave-per-entity LDS kernel, workgroup-per-entity kernel).
template <class G>
__device__ void lbfgs_solve(G& grp, const EntityView& P, const SolveParams& o, const Work& W, SolveStats& out) {
  LbfgsState T;
  lbfgs_init(T);
  do {
    const double f = eval_fg(grp, P, o, W);
    lbfgs_advance(grp, P.p, o, W, T, f);
  } while (T.status < 0);
  out.f = T.f;
  out.gnorm = T.sbgnrm;
  out.nit = T.nit;
  out.nfev = T.nfev;
  out.status = T.status;
}

// _compute_variance, SIMPLE mode (binary_logistic_regression.py:175-180): 1/(sum_i X~_ij^2 D_i + l2*[j reg] + 1e-12),
// D_i = rho_i (1-rho_i) w_i. Duplicate (row, col) entries are summed before squaring, as the reference's
// toarray() does. W.rs is reused for D.
// SC1: the team's exchanged vectors (W.x, W.rs) are accessed through sc1 loads / stores (re_device.hpp, ld_x / st_x)
template <bool SC1 = false, class G>
__device__ __forceinline__ void variance_simple(G& grp, const EntityView& P, const SolveParams& o, const Work& W,
                                                double* var_out) {
  const int n = P.n, p = P.p, ic = P.ic;
  const double* __restrict__ x = W.x;
  const double x0 = ic ? ld_x<SC1>(x) : 0.0;
  double dpart = 0.0;
  for (int i = grp.tid; i < n; i += grp.NT) {
    double acc = x0;
    const int k1 = P.row_ptr[i + 1];
    for (int k = P.row_ptr[i]; k < k1; ++k) acc += (double)P.csr_val[k] * ld_x<SC1>(x + ic + P.csr_col[k]);
    const double z = acc + (double)P.o[i];
    const double rho = 1.0 / (1.0 + exp(-z));
    const double di = rho * (1.0 - rho) * (P.w ? (double)P.w[i] : 1.0);
    st_x<SC1>(W.rs + i, di);
    dpart += di;
  }
  const double dsum = grp.sum(dpart);
  grp.sync();
  const int first_reg = (ic && !o.regularize_bias) ? 1 : 0;
  for (int j = grp.tid; j < p; j += grp.NT) {
    double h;
    if (ic && j == 0) {
      h = dsum;
    } else {
      h = 0.0;
      const int c = j - ic;
      const int k1 = P.col_ptr[c + 1];
      int k = P.col_ptr[c];
      while (k < k1) {   // runs of equal row = duplicates of one matrix cell
        const int row = P.csc_row[k];
        double v = (double)P.csc_val[k];
        ++k;
        while (k < k1 && P.csc_row[k] == row) { v += (double)P.csc_val[k]; ++k; }
        h += v * v * ld_x<SC1>(W.rs + row);
      }
    }
    h += (j < first_reg) ? 0.0 : o.l2;
    var_out[j] = 1.0 / (h + 1.0e-12);
  }
  grp.sync();
}

}  // namespace gdmix
