// re_solve_wreg.hpp — the L-BFGS driver with one coefficient set per lane in registers (wreg_solve_ev), as the master wavefront
// of the tall kernel runs it (re_solve_tall.hip). Round 1's kernel around it — one wavefront per entity, X block in LDS — was
// superseded by the group kernels (re_solve_quad.hpp) and removed in round 4 with its thirteen size classes.
//
// Coefficient j of the entity lives in lane (j mod 64), slot (j div 64): x, g, d, x_old, g_old and
// the whole (s, y) history (M_REG = 10 pairs) are VGPRs. LDS holds only what lanes gather from each
// other — the trial point (for X~theta), the per-sample residual (for X'r) — plus the entity's X block
// (CSR and CSC copies as packed {index, value} pairs) and its y / offset / weight.
//
// The history is kept as a shift register (newest pair at index M_REG-1) so that every index into the
// register arrays is a compile-time constant; a shift costs 36*EPL v_mov per accepted iteration, noise
// next to the 2*cnt wave reductions of the two-loop recursion.
//
// Same algorithm, same stopping rules and the same accumulation order inside X~theta and X'r as
// re_solve_core.hpp / oracle/re_oracle.c (fit(), binary_logistic_regression.py:191-239).
#pragma once
#include "re_solve_core.hpp"

namespace gdmix {

constexpr int M_REG = 10;   // history pairs held in registers; larger m is served by the LDS kernel

__device__ __forceinline__ double uniform_d(double v) { return readlane0(v); }

// two sums in one pass: the DPP chains of a and b interleave, hiding each other's latency
__device__ __forceinline__ void wave_sum2(double& a, double& b) {
#define GDMIX_STEP2(CTRL, MASK)              \
  {                                          \
    const double ta = dpp_get0<CTRL, MASK>(a); \
    const double tb = dpp_get0<CTRL, MASK>(b); \
    a += ta;                                 \
    b += tb;                                 \
  }
  GDMIX_STEP2(0x111, 0xf) GDMIX_STEP2(0x112, 0xf) GDMIX_STEP2(0x114, 0xf) GDMIX_STEP2(0x118, 0xf)
  GDMIX_STEP2(0x142, 0xa) GDMIX_STEP2(0x143, 0xc)
#undef GDMIX_STEP2
  a = readlane63(a);
  b = readlane63(b);
}

// two sums and one max (of non-negative values) in one pass
__device__ __forceinline__ void wave_sum2_max(double& a, double& b, double& c) {
#define GDMIX_STEP3(CTRL, MASK)              \
  {                                          \
    const double ta = dpp_get0<CTRL, MASK>(a); \
    const double tb = dpp_get0<CTRL, MASK>(b); \
    const double tc = dpp_get0<CTRL, MASK>(c); \
    a += ta;                                 \
    b += tb;                                 \
    c = fmax(c, tc);                         \
  }
  GDMIX_STEP3(0x111, 0xf) GDMIX_STEP3(0x112, 0xf) GDMIX_STEP3(0x114, 0xf) GDMIX_STEP3(0x118, 0xf)
  GDMIX_STEP3(0x142, 0xa) GDMIX_STEP3(0x143, 0xc)
#undef GDMIX_STEP3
  a = readlane63(a);
  b = readlane63(b);
  c = readlane63(c);
}

template <int EPL>
struct WregState {
  double x[EPL], g[EPL], d[EPL], xo[EPL], go[EPL];
};

// f and g at the point held in `xt` (registers). g <- gradient, returns f. rsum_out: sum of residuals.
template <int EPL, class Eval>
__device__ __forceinline__ void wreg_solve_ev(double* const rho, double* const alpha, LineSearch* const lsp, const SolveParams& o,
                                              WregState<EPL>& V, SolveStats& out, Eval&& eval) {
  double S[M_REG][EPL], Y[M_REG][EPL];
#pragma unroll
  for (int a = 0; a < M_REG; ++a) {
#pragma unroll
    for (int s = 0; s < EPL; ++s) { S[a][s] = 0.0; Y[a][s] = 0.0; }
  }
  const int m = o.m;   // rho, alpha: every lane stores the same value (uniform LDS state)
  int cnt = 0;
  double theta = 1.0;
  int nit = 0, nfev = 0, status = -1, ifun = 0;
  bool iter0 = true, first = true;
  double f = 0.0, fold = 0.0, gd = 0.0, gdold = 0.0, rr = 0.0, stp = 0.0, sbgnrm = 0.0;
  for (;;) {
    // ---- f, g at the trial point; g'd, y'y and max|g| in one reduction pass (the first trial of a
    //      line search is accepted ~95% of the time, so y'y and max|g| are computed speculatively) -------
    bool counted;
    f = uniform_d(eval(V.x, V.g, first, counted));
    nfev += counted ? 1 : 0;
    {
      double a = 0.0, b = 0.0, c = 0.0;
#pragma unroll
      for (int s = 0; s < EPL; ++s) {
        a += V.g[s] * V.d[s];
        const double yj = V.g[s] - V.go[s];
        b += yj * yj;
        c = fmax(c, fabs(V.g[s]));
      }
      wave_sum2_max(a, b, c);
      gd = a; rr = b; sbgnrm = c;
    }
    bool restart = false;
    if (first) {
      first = false;
      if (sbgnrm <= o.pgtol) { status = 0; break; }
    } else {
      LineSearch LS = *lsp;
      const int task = dcsrch_step(LS, f, gd, stp);
      stp = uniform_d(stp);
      *lsp = LS;
      if (task == LS_FG) {
        ++ifun;
        if (ifun - 1 < o.maxls) {
          if (stp == 1.0) {
#pragma unroll
            for (int s = 0; s < EPL; ++s) V.x[s] = V.xo[s] + V.d[s];
          } else {
#pragma unroll
            for (int s = 0; s < EPL; ++s) V.x[s] = stp * V.d[s] + V.xo[s];
          }
          continue;
        }
        restart = true;   // iback >= maxls
      } else {
        // ---- NEW_X: scipy's python loop first (nit / maxiter / maxfun), then mainlb's own tests -----
        ++nit;
        iter0 = false;
        if (nit >= o.max_iter) { status = 2; break; }
        if (nfev > o.maxfun) { status = 3; break; }
        if (sbgnrm <= o.pgtol) { status = 0; break; }
        {
          const double ddum = fmax(fabs(fold), fmax(fabs(f), 1.0));
          if (fold - f <= o.ftol * ddum) { status = 1; break; }
        }
        double dr, ddum;
        if (stp == 1.0) { dr = gd - gdold; ddum = -gdold; }
        else { dr = (gd - gdold) * stp; ddum = -gdold * stp; }
        if (dr > EPSMCH * ddum) {
          // push (s, y): shift the register history down by one, newest at M_REG-1
#pragma unroll
          for (int a = 0; a < M_REG - 1; ++a) {
            rho[a] = rho[a + 1];
#pragma unroll
            for (int s = 0; s < EPL; ++s) { S[a][s] = S[a + 1][s]; Y[a][s] = Y[a + 1][s]; }
          }
#pragma unroll
          for (int s = 0; s < EPL; ++s) {
            S[M_REG - 1][s] = (stp == 1.0) ? V.d[s] : stp * V.d[s];
            Y[M_REG - 1][s] = V.g[s] - V.go[s];
          }
          rho[M_REG - 1] = uniform_d(1.0 / dr);
          theta = uniform_d(rr / dr);
          if (cnt < m) ++cnt;
        }
      }
    }
    // ---- new search direction (repeated from steepest descent after a line-search restart) -----------
    for (;;) {
      if (restart) {
#pragma unroll
        for (int s = 0; s < EPL; ++s) { V.x[s] = V.xo[s]; V.g[s] = V.go[s]; }
        f = fold;
        if (cnt == 0) { status = 4; break; }
        cnt = 0; theta = 1.0;
        restart = false;
      }
#pragma unroll
      for (int s = 0; s < EPL; ++s) V.d[s] = -V.g[s];
      if (cnt > 0) {
        // two-loop recursion over the last cnt pairs (indices M_REG-cnt .. M_REG-1, newest last)
#pragma unroll
        for (int a = M_REG - 1; a >= 0; --a) {
          if (a >= M_REG - cnt) {
            double t = 0.0;
#pragma unroll
            for (int s = 0; s < EPL; ++s) t += S[a][s] * V.d[s];
            const double al = uniform_d(rho[a] * wave_sum(t));
            alpha[a] = al;
#pragma unroll
            for (int s = 0; s < EPL; ++s) V.d[s] -= al * Y[a][s];
          }
        }
        const double h0 = 1.0 / theta;
#pragma unroll
        for (int s = 0; s < EPL; ++s) V.d[s] *= h0;
#pragma unroll
        for (int a = 0; a < M_REG; ++a) {
          if (a >= M_REG - cnt) {
            double t = 0.0;
#pragma unroll
            for (int s = 0; s < EPL; ++s) t += Y[a][s] * V.d[s];
            const double c = uniform_d(alpha[a] - rho[a] * wave_sum(t));
#pragma unroll
            for (int s = 0; s < EPL; ++s) V.d[s] += c * S[a][s];
          }
        }
      }
      // z = x + d ; d = z - x (mainlb re-derives d from the subspace point); save x, g
      double dd = 0.0, gdp = 0.0;
#pragma unroll
      for (int s = 0; s < EPL; ++s) {
        const double xj = V.x[s];
        const double z = xj + V.d[s];
        const double dj = z - xj;
        V.d[s] = dj;
        V.xo[s] = xj;
        V.go[s] = V.g[s];
        dd += dj * dj;
        gdp += V.g[s] * dj;
      }
      wave_sum2(dd, gdp);
      gd = gdp;
      gdold = gd;
      fold = f;
      if (gd >= 0.0) { restart = true; continue; }   // lnsrlb info = -4
      stp = iter0 ? uniform_d(fmin(1.0 / sqrt(dd), LS_STPMAX)) : 1.0;
      {
        LineSearch LS;
        dcsrch_start(LS, f, gd, stp);
        *lsp = LS;
      }
      ifun = 1;
      break;
    }
    if (status >= 0) break;
    if (stp == 1.0) {
#pragma unroll
      for (int s = 0; s < EPL; ++s) V.x[s] = V.xo[s] + V.d[s];
    } else {
#pragma unroll
      for (int s = 0; s < EPL; ++s) V.x[s] = stp * V.d[s] + V.xo[s];
    }
  }
  if (status == 4) {   // abnormal stop: report the restored gradient's norm
    double mx = 0.0;
#pragma unroll
    for (int s = 0; s < EPL; ++s) mx = fmax(mx, fabs(V.g[s]));
    sbgnrm = wave_max_nonneg(mx);
  }
  out.f = f;
  out.gnorm = sbgnrm;
  out.nit = nit;
  out.nfev = nfev;
  out.status = status;
}

}  // namespace gdmix
