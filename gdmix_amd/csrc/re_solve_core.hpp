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
  for (int i = grp.tid; i < n; i += G::NT) {
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
  for (int j = first_reg + grp.tid; j < p; j += G::NT) sq += x[j] * x[j];
  part += 0.5 * o.l2 * sq;
  const double inv_n = 1.0 / (double)n;
  const double f = inv_n * grp.sum(part);
  const double rsum = grp.sum(rpart);   // also orders the rs[] writes before the reads below (block)
  grp.sync();
  for (int j = grp.tid; j < p; j += G::NT) {
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
  for (int j = grp.tid; j < p; j += G::NT) s += a[j] * b[j];
  return grp.sum(s);
}

template <class G>
__device__ __forceinline__ double maxabs(G& grp, const double* a, int p) {
  double s = 0.0;
  for (int j = grp.tid; j < p; j += G::NT) s = fmax(s, fabs(a[j]));
  return grp.max_nonneg(s);
}

struct SolveStats {
  double f, gnorm;
  int nit, nfev, status;
};

// The whole fmin_l_bfgs_b run for one entity.
// Synchronisation: element j of every p-vector is only ever touched by thread j mod NT, so the vector
// updates need no barrier; the only cross-thread traffic is the x / rs gathers inside eval_fg.
template <class G>
__device__ void lbfgs_solve(G& grp, const EntityView& P, const SolveParams& o, const Work& W, SolveStats& out) {
  const int p = P.p, m = o.m;
  double* alpha = W.alpha;
  double* rho = W.rho;
  int col = 0, head = 0;
  double theta = 1.0;
  int nit = 0, nfev = 1, status = -1;
  bool iter0 = true;
  double f = eval_fg(grp, P, o, W);
  double sbgnrm = maxabs(grp, W.g, p);
  if (sbgnrm <= o.pgtol) status = 0;
  while (status < 0) {
    // ---- direction: two-loop recursion over the stored pairs, H0 = (1/theta) I ------------------
    for (int j = grp.tid; j < p; j += G::NT) W.d[j] = -W.g[j];
    if (col > 0) {
      for (int a = col - 1; a >= 0; --a) {
        int sl = head + a; if (sl >= m) sl -= m;
        const double* s = W.ws + (size_t)sl * p;
        const double* yv = W.wy + (size_t)sl * p;
        const double al = rho[sl] * dot(grp, s, W.d, p);
        alpha[a] = al;
        for (int j = grp.tid; j < p; j += G::NT) W.d[j] -= al * yv[j];
      }
      const double h0 = 1.0 / theta;
      for (int j = grp.tid; j < p; j += G::NT) W.d[j] *= h0;
      for (int a = 0; a < col; ++a) {
        int sl = head + a; if (sl >= m) sl -= m;
        const double* s = W.ws + (size_t)sl * p;
        const double* yv = W.wy + (size_t)sl * p;
        const double beta = rho[sl] * dot(grp, yv, W.d, p);
        const double c = alpha[a] - beta;
        for (int j = grp.tid; j < p; j += G::NT) W.d[j] += c * s[j];
      }
    }
    // z = x + d ; d = z - x  (mainlb re-derives d from the subspace point); save x, g
    double dd = 0.0, gdp = 0.0;
    for (int j = grp.tid; j < p; j += G::NT) {
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
    double gd = grp.sum(gdp);
    const double gdold = gd;
    double stp = iter0 ? fmin(1.0 / dnorm, LS_STPMAX) : 1.0;
    const double fold = f;
    bool restart = false;
    if (gd >= 0.0) {
      restart = true;   // lnsrlb info = -4
    } else {
      LineSearch S;
      dcsrch_start(S, f, gd, stp);
      int ifun = 0;
      for (;;) {
        ++ifun;
        if (ifun - 1 >= o.maxls) { restart = true; break; }
        if (stp == 1.0) {
          for (int j = grp.tid; j < p; j += G::NT) W.x[j] = W.t[j] + W.d[j];
        } else {
          for (int j = grp.tid; j < p; j += G::NT) W.x[j] = stp * W.d[j] + W.t[j];
        }
        grp.sync();
        f = eval_fg(grp, P, o, W);
        ++nfev;
        gd = dot(grp, W.g, W.d, p);
        if (dcsrch_step(S, f, gd, stp) != LS_FG) break;
      }
    }
    if (restart) {
      for (int j = grp.tid; j < p; j += G::NT) { W.x[j] = W.t[j]; W.g[j] = W.r[j]; }
      grp.sync();
      f = fold;
      if (col == 0) { status = 4; break; }
      col = 0; head = 0; theta = 1.0;
      continue;
    }
    // ---- NEW_X: scipy's python loop first (nit / maxiter / maxfun), then mainlb's own tests ------
    ++nit;
    iter0 = false;
    sbgnrm = maxabs(grp, W.g, p);
    if (nit >= o.max_iter) { status = 2; break; }
    if (nfev > o.maxfun) { status = 3; break; }
    if (sbgnrm <= o.pgtol) { status = 0; break; }
    {
      const double ddum = fmax(fabs(fold), fmax(fabs(f), 1.0));
      if (fold - f <= o.ftol * ddum) { status = 1; break; }
    }
    // ---- pair update: s = stp*d, y = g - g_old; skipped when s'y <= epsmch * (-g_old'd*stp) ------
    double rrp = 0.0;
    for (int j = grp.tid; j < p; j += G::NT) {
      const double yj = W.g[j] - W.r[j];
      W.r[j] = yj;
      rrp += yj * yj;
    }
    const double rr = grp.sum(rrp);
    double dr, ddum;
    if (stp == 1.0) { dr = gd - gdold; ddum = -gdold; }
    else { dr = (gd - gdold) * stp; ddum = -gdold * stp; }
    if (dr <= EPSMCH * ddum) continue;
    int sl;
    if (col < m) { sl = head + col; if (sl >= m) sl -= m; ++col; }
    else { sl = head; ++head; if (head >= m) head = 0; }
    double* s = W.ws + (size_t)sl * p;
    double* yv = W.wy + (size_t)sl * p;
    for (int j = grp.tid; j < p; j += G::NT) {
      s[j] = (stp == 1.0) ? W.d[j] : stp * W.d[j];
      yv[j] = W.r[j];
    }
    rho[sl] = 1.0 / dr;
    theta = rr / dr;
  }
  out.f = f;
  out.gnorm = sbgnrm;
  out.nit = nit;
  out.nfev = nfev;
  out.status = status;
}

// _compute_variance, SIMPLE mode (binary_logistic_regression.py:175-180): 1/(sum_i X~_ij^2 D_i + l2*[j reg] + 1e-12),
// D_i = rho_i (1-rho_i) w_i. Duplicate (row, col) entries are summed before squaring, as the reference's
// toarray() does. W.rs is reused for D.
template <class G>
__device__ __forceinline__ void variance_simple(G& grp, const EntityView& P, const SolveParams& o, const Work& W,
                                                double* var_out) {
  const int n = P.n, p = P.p, ic = P.ic;
  const double* __restrict__ x = W.x;
  const double x0 = ic ? x[0] : 0.0;
  double dpart = 0.0;
  for (int i = grp.tid; i < n; i += G::NT) {
    double acc = x0;
    const int k1 = P.row_ptr[i + 1];
    for (int k = P.row_ptr[i]; k < k1; ++k) acc += (double)P.csr_val[k] * x[ic + P.csr_col[k]];
    const double z = acc + (double)P.o[i];
    const double rho = 1.0 / (1.0 + exp(-z));
    const double di = rho * (1.0 - rho) * (P.w ? (double)P.w[i] : 1.0);
    W.rs[i] = di;
    dpart += di;
  }
  const double dsum = grp.sum(dpart);
  grp.sync();
  const int first_reg = (ic && !o.regularize_bias) ? 1 : 0;
  for (int j = grp.tid; j < p; j += G::NT) {
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
        h += v * v * W.rs[row];
      }
    }
    h += (j < first_reg) ? 0.0 : o.l2;
    var_out[j] = 1.0 / (h + 1.0e-12);
  }
  grp.sync();
}

}  // namespace gdmix
