// re_solve_quad.hpp — FOUR ENTITIES PER WAVEFRONT: each DPP row (16 lanes) solves one entity.
//
// Why: the register-resident wave kernel is VALU-issue bound (rocprof: SQ_ACTIVE_INST_VALU ~ 90% of the
// SIMD cycles) and about half of its VALU instructions are the fp64 cross-lane reductions of the two-loop
// recursion (6 DPP stages + readlane for every dot product, ~25 per iteration). With one entity per
// 16-lane row, a reduction is 4 within-row DPP stages (quad_perm x2, row_half_mirror, row_mirror), needs
// no cross-row traffic and no readlane, and — because every VALU instruction serves four entities — its
// cost per entity drops ~6x; all other per-wave work is shared by four entities as well.
//
// Coefficient j of an entity lives in lane (j mod 16) of its row, slot (j div 16); EPL slots per lane
// hold p <= 16*EPL coefficients. Values that are uniform per ENTITY (f, step, line-search state, ...) are
// ordinary per-lane values that happen to be equal inside a row: the butterfly reductions below give
// bit-identical results to all 16 lanes, so every branch on them takes whole rows. The four entities
// of a wave advance independently, one function evaluation per trip of the main loop; an entity that is
// still line-searching simply sits out the direction update of that trip.
//
// Same algorithm, stopping rules and accumulation order inside X~theta and X'r as re_solve_core.hpp /
// oracle/re_oracle.c (fit(), binary_logistic_regression.py:191-239).
#pragma once
#include "re_solve_wreg.hpp"

namespace gdmix {

constexpr int ROW = 16;   // lanes per entity

template <int CTRL>
__device__ __forceinline__ double dpp_row(double v) {
  int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, true);
  int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}

// dpp_ctrl: quad_perm:[1,0,3,2] = 0xB1, quad_perm:[2,3,0,1] = 0x4E, row_half_mirror = 0x141, row_mirror = 0x140
// Butterfly: every lane of the row ends with the same bits (each stage adds the same two partial sums).
__device__ __forceinline__ double row_sum(double v) {
  v += dpp_row<0xB1>(v);
  v += dpp_row<0x4E>(v);
  v += dpp_row<0x141>(v);
  v += dpp_row<0x140>(v);
  return v;
}

__device__ __forceinline__ void row_sum2(double& a, double& b) {
#define GDMIX_RSTEP2(CTRL)              \
  {                                     \
    const double ta = dpp_row<CTRL>(a); \
    const double tb = dpp_row<CTRL>(b); \
    a += ta;                            \
    b += tb;                            \
  }
  GDMIX_RSTEP2(0xB1) GDMIX_RSTEP2(0x4E) GDMIX_RSTEP2(0x141) GDMIX_RSTEP2(0x140)
#undef GDMIX_RSTEP2
}

// max of non-negative values via the bit pattern order of IEEE doubles (no canonicalising v_max needed)
__device__ __forceinline__ double max_nn(double a, double b) { return (a > b) ? a : b; }

__device__ __forceinline__ void row_sum2_max(double& a, double& b, double& c) {
#define GDMIX_RSTEP3(CTRL)              \
  {                                     \
    const double ta = dpp_row<CTRL>(a); \
    const double tb = dpp_row<CTRL>(b); \
    const double tc = dpp_row<CTRL>(c); \
    a += ta;                            \
    b += tb;                            \
    c = max_nn(c, tc);                  \
  }
  GDMIX_RSTEP3(0xB1) GDMIX_RSTEP3(0x4E) GDMIX_RSTEP3(0x141) GDMIX_RSTEP3(0x140)
#undef GDMIX_RSTEP3
}

// LDS bytes of ONE entity (row) in the quad kernel; a wavefront uses four of these
__host__ __device__ inline size_t quad_lds_bytes(int p, int n, int nnz, int d, bool has_w) {
  return wreg_lds_bytes(p, n, nnz, d, has_w);   // same carve-up as the register wave kernel
}

template <int EPL>
__device__ __forceinline__ double quad_eval(const WregLds& L, const SolveParams& o, int gl, int n, int p, int ic,
                                            const double (&xt)[EPL], double (&g)[EPL]) {
#pragma unroll
  for (int s = 0; s < EPL; ++s) {
    const int j = gl + ROW * s;
    if (j < p) L.xs[j] = xt[s];
  }
  wave_lds_fence();
  double part = 0.0, rpart = 0.0;
  const double x0 = ic ? L.xs[0] : 0.0;
  for (int i = gl; i < n; i += ROW) {
    double acc = x0;
    const int k1 = L.row_ptr[i + 1];
    for (int k = L.row_ptr[i]; k < k1; ++k) {
      const int2 cv = L.csr[k];
      acc += (double)__int_as_float(cv.y) * L.xs[ic + cv.x];
    }
    const double z = acc + (double)L.o[i];
    const double yi = (double)L.y[i];
    const double wi = L.w ? (double)L.w[i] : 1.0;
    const double e = exp(-fabs(z));
    const double ce = fmax(z, 0.0) - z * yi + log(1.0 + e);
    const double sig = (z >= 0.0) ? 1.0 / (1.0 + e) : e / (1.0 + e);
    const double ri = wi * (sig - yi);
    L.rs[i] = ri;
    part += wi * ce;
    rpart += ri;
  }
  const int first_reg = (ic && !o.regularize_bias) ? 1 : 0;
  double sq = 0.0;
#pragma unroll
  for (int s = 0; s < EPL; ++s) {
    const int j = gl + ROW * s;
    if (j >= first_reg && j < p) sq += xt[s] * xt[s];
  }
  part += 0.5 * o.l2 * sq;
  row_sum2(part, rpart);
  wave_lds_fence();
  const double inv_n = 1.0 / (double)n;
#pragma unroll
  for (int s = 0; s < EPL; ++s) {
    const int j = gl + ROW * s;
    double gj = 0.0;
    if (j < p) {
      double acc;
      if (ic && j == 0) {
        acc = rpart;
      } else {
        acc = 0.0;
        const int c = j - ic;
        const int k1 = L.col_ptr[c + 1];
        for (int k = L.col_ptr[c]; k < k1; ++k) {
          const int2 rv = L.csc[k];
          acc += (double)__int_as_float(rv.y) * L.rs[rv.x];
        }
      }
      const double reg = (j < first_reg) ? 0.0 : o.l2 * xt[s];
      gj = inv_n * (acc + reg);
    }
    g[s] = gj;
  }
  return inv_n * part;
}

// The solve of the (up to) four entities of a wave. `valid` marks rows that own an entity.
template <int EPL>
__device__ __forceinline__ void quad_solve(const WregLds& L, const SolveParams& o, int gl, int n, int p, int ic,
                                           bool valid, WregState<EPL>& V, SolveStats& out) {
  double S[M_REG][EPL], Y[M_REG][EPL];
#pragma unroll
  for (int a = 0; a < M_REG; ++a) {
#pragma unroll
    for (int s = 0; s < EPL; ++s) { S[a][s] = 0.0; Y[a][s] = 0.0; }
  }
  double* const rho = L.rho;      // per-entity uniform state, every lane of the row stores the same value
  double* const alpha = L.alpha;
  const int m = o.m;
  int cnt = 0;
  double theta = 1.0;
  int nit = 0, nfev = 0, ifun = 0;
  int status = valid ? -1 : 0;
  bool iter0 = true, first = true;
  double f = 0.0, fold = 0.0, gd = 0.0, gdold = 0.0, rr = 0.0, stp = 0.0, sbgnrm = 0.0;
  while (__any(status < 0)) {
    bool need_dir = false, restart = false;
    if (status < 0) {
      // ---- f, g at the trial point; g'd, y'y and max|g| in one reduction pass ------------------------
      f = quad_eval<EPL>(L, o, gl, n, p, ic, V.x, V.g);
      ++nfev;
      {
        double a = 0.0, b = 0.0, c = 0.0;
#pragma unroll
        for (int s = 0; s < EPL; ++s) {
          a += V.g[s] * V.d[s];
          const double yj = V.g[s] - V.go[s];
          b += yj * yj;
          c = max_nn(c, fabs(V.g[s]));
        }
        row_sum2_max(a, b, c);
        gd = a; rr = b; sbgnrm = c;
      }
      if (first) {
        first = false;
        if (sbgnrm <= o.pgtol) status = 0;
        else need_dir = true;
      } else {
        LineSearch LS = *L.ls;
        const int task = dcsrch_step(LS, f, gd, stp);
        *L.ls = LS;
        if (task == LS_FG) {
          ++ifun;
          if (ifun - 1 < o.maxls) {
#pragma unroll
            for (int s = 0; s < EPL; ++s) V.x[s] = stp * V.d[s] + V.xo[s];   // stp == 1: exactly xo + d
          } else {
            restart = true;   // iback >= maxls
            need_dir = true;
          }
        } else {
          // ---- NEW_X: scipy's python loop first (nit / maxiter / maxfun), then mainlb's own tests ---
          ++nit;
          iter0 = false;
          const double dmx = fmax(fabs(fold), fmax(fabs(f), 1.0));
          if (nit >= o.max_iter) status = 2;
          else if (nfev > o.maxfun) status = 3;
          else if (sbgnrm <= o.pgtol) status = 0;
          else if (fold - f <= o.ftol * dmx) status = 1;
          else {
            need_dir = true;
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
                S[M_REG - 1][s] = stp * V.d[s];   // exact for stp == 1
                Y[M_REG - 1][s] = V.g[s] - V.go[s];
              }
              rho[M_REG - 1] = 1.0 / dr;
              theta = rr / dr;
              if (cnt < m) ++cnt;
            }
          }
        }
      }
    }
    // ---- new search direction for the rows that need one (again after a line-search restart) ---------
    while (__any(need_dir)) {
      if (need_dir && restart) {
#pragma unroll
        for (int s = 0; s < EPL; ++s) { V.x[s] = V.xo[s]; V.g[s] = V.go[s]; }
        f = fold;
        restart = false;
        if (cnt == 0) { status = 4; need_dir = false; }
        else { cnt = 0; theta = 1.0; }
      }
      if (need_dir) {
#pragma unroll
        for (int s = 0; s < EPL; ++s) V.d[s] = -V.g[s];
      }
      // two-loop recursion over each row's last cnt pairs (indices M_REG-cnt .. M_REG-1, newest last)
#pragma unroll
      for (int a = M_REG - 1; a >= 0; --a) {
        const bool use = need_dir && (a >= M_REG - cnt);
        if (__any(use)) {
          if (use) {
            double t = 0.0;
#pragma unroll
            for (int s = 0; s < EPL; ++s) t += S[a][s] * V.d[s];
            const double al = rho[a] * row_sum(t);
            alpha[a] = al;
#pragma unroll
            for (int s = 0; s < EPL; ++s) V.d[s] -= al * Y[a][s];
          }
        }
      }
      if (need_dir && cnt > 0) {
        const double h0 = 1.0 / theta;
#pragma unroll
        for (int s = 0; s < EPL; ++s) V.d[s] *= h0;
      }
#pragma unroll
      for (int a = 0; a < M_REG; ++a) {
        const bool use = need_dir && (a >= M_REG - cnt);
        if (__any(use)) {
          if (use) {
            double t = 0.0;
#pragma unroll
            for (int s = 0; s < EPL; ++s) t += Y[a][s] * V.d[s];
            const double c = alpha[a] - rho[a] * row_sum(t);
#pragma unroll
            for (int s = 0; s < EPL; ++s) V.d[s] += c * S[a][s];
          }
        }
      }
      if (need_dir) {
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
        row_sum2(dd, gdp);
        gd = gdp;
        gdold = gd;
        fold = f;
        if (gd >= 0.0) {
          restart = true;   // lnsrlb info = -4: stay in this loop
        } else {
          stp = iter0 ? fmin(1.0 / sqrt(dd), LS_STPMAX) : 1.0;
          LineSearch LS;
          dcsrch_start(LS, f, gd, stp);
          *L.ls = LS;
          ifun = 1;
#pragma unroll
          for (int s = 0; s < EPL; ++s) V.x[s] = stp * V.d[s] + V.xo[s];
          need_dir = false;
        }
      }
    }
  }
  if (status == 4) {   // abnormal stop: report the restored gradient's norm
    double mx = 0.0;
#pragma unroll
    for (int s = 0; s < EPL; ++s) mx = max_nn(mx, fabs(V.g[s]));
    double d0 = 0.0, d1 = 0.0;
    row_sum2_max(d0, d1, mx);
    sbgnrm = mx;
  }
  out.f = f;
  out.gnorm = sbgnrm;
  out.nit = nit;
  out.nfev = nfev;
  out.status = status;
}

}  // namespace gdmix
