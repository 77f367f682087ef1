// re_lbfgs_compact.hpp — the driver of fmin_l_bfgs_b in compact (Byrd-Nocedal-Schnabel) form, as a resumable step:
// given the objective value and the fused products of one evaluation, decide what happens next (stop, another
// line-search trial, a new search direction) and leave the coefficients of the elementwise update. Shared by the
// team kernels (re_solve_team.hpp: state in registers, one kernel per solve) and the fixed-effect stepping kernels
// (fe_solve.hip: state in HBM between kernels, an all-reduce in between). Rules: scipy's loop around setulb,
// L-BFGS-B 3.0 mainlb / lnsrlb / matupd, MINPACK-2 dcsrch (re_device.hpp).
#pragma once
#include "re_solve_core.hpp"

namespace gdmix {

constexpr int TEAM_MCAP = 10;                 // history pairs the compact path keeps accumulators for
constexpr int TEAM_K = 2 * TEAM_MCAP + 7;     // fused reduction width: sq, gd, gg, yy, yg, S'y, Y'y, r'd, max|g|
constexpr int TEAM_RD = 2 * TEAM_MCAP + 5;    // acc index of r'd
constexpr int COMPACT_KD = TEAM_K + 2 * TEAM_MCAP;   // acc[] with the DIRECT products behind it: TEAM_K + i = S_i'g, TEAM_K + MCAP + i = Y_i'g
// acc[] layout, with y = g - r (r = the gradient at the last accepted iterate): 0 sum x_j^2 over regularised j, 1 g'd,
// 2 g'g, 3 y'y, 4 y'g, 5.. S_i'y, 5+MCAP.. Y_i'y (chronological i < col), K-2 r'd, K-1 max|g_j|.
// nfev is scipy's funcalls: ScalarFunction serves a point equal to the previously evaluated one from its cache without counting
// it (re_solve_quad.hpp, quad_solve). Whether the trial moved is known to the threads that formed it (compact_update_n returns
// it per coefficient); the team kernels pass it on as a flag next to their barrier (Team::moved_*), compact_advance takes it as
// `counted`. The fixed-effect stepping kernels count every evaluation.
// r'd is the slope of the line search at its start, taken from the direction as it is actually used (after mainlb's
// d = (x + d) - x): the small solve also yields g'd = -g'Hg algebraically, but the two differ once x dwarfs d (badly scaled
// entities; tests/golden exit_hard_*, exit_extreme_*), and L-BFGS-B's line search runs on the former (lnsrlb: gd = ddot(g, d)).
// It arrives with the first trial's products and replaces the algebraic value before the search takes its first decision.
// The products are taken with y, not with g: S'g and Y'g at the new gradient are the stored ones plus these, and the new
// column of S'Y / Y'Y is these — sums, where products with g would need differences of nearly equal numbers once the
// gradient changes little between iterates.
// ... and S'g and Y'g are ALSO taken directly (round 6; compact_advance's `sg`, 2 m more accumulators: COMPACT_KD in all): a stored s_i'g_k that is the sum of s_i'y over the iterations since the pair was made carries
// the rounding of every term at the size the gradient had THEN — eps |s_i| |y_k| each — which is |g_then| / |g_now| times what the
// direct product's is. A fit whose gradient falls by 10^3 - 10^4 inside the history window (tools/fuzz_fe.py case 6700230: squared
// loss, |g| 3 000 -> 0.9 in twelve iterations) left the reference's trajectory by 1e-5 at the first iterate the problem amplifies,
// where L-BFGS-B itself (direct products: cauchy's p = W'g) and the two-loop oracle stay together to 1e-9 (profiles/r06_fuzz.txt).
// Without `sg` (GDMIX_TEAM_DIRECT_AB=0 builds of the team kernels) the sums are used, as in rounds 3 - 5.

// History layout of the compact-form kernels: tiles of 64 consecutive coefficients, and inside a tile slot-major:
//     (s, y) of coefficient j, slot sl  =  ((double2*)W.ws)[((j >> 6) * m + sl) * 64 + (j & 63)]
// (W.wy is not used; the two arrays are adjacent; 2 m (p rounded up to 64) doubles). A wavefront works on one tile at a time:
// every load / store of a slot is 1 KB contiguous (16 bytes per lane), and the 2 m vectors of a tile are one contiguous 20 KB
// run instead of 2 m streams 8 p bytes apart. (Interleaving per coefficient — 160 contiguous bytes per lane — measured 29 %
// slower than slot-major arrays: 16 of every 64-byte line per instruction, and partial-line stores.)
__device__ __forceinline__ double2* compact_hist(const Work& W, int m, int j) {
  return reinterpret_cast<double2*>(W.ws) + ((size_t)(j >> 6) * m) * 64 + (j & 63);
}
constexpr int COMPACT_HIST_STRIDE = 64;   // compact_hist(W, m, j)[sl * COMPACT_HIST_STRIDE]
inline size_t compact_hist_doubles(int64_t p, int m) { return (size_t)2 * m * (((size_t)p + 63) & ~(size_t)63); }

struct CompactState {   // uniform over the cooperating threads
  LineSearch ls;
  double theta, f, fold, gdold, stp, sbgnrm, gg_k;
  int col, head, nit, nfev, ifun, status, first, iter0;
};

__device__ __forceinline__ void compact_init(CompactState& S) {
  S.theta = 1.0; S.f = 0.0; S.fold = 0.0; S.gdold = 0.0; S.stp = 0.0; S.sbgnrm = 0.0; S.gg_k = 0.0;
  S.col = 0; S.head = 0; S.nit = 0; S.nfev = 0; S.ifun = 0; S.status = -1; S.first = 1; S.iter0 = 1;
}

enum { CA_STOP = 0, CA_STOP_RESTORE = 1, CA_RETRY = 2, CA_DIRECTION = 3 };

struct CompactPlan {    // what the elementwise pass over the p-vectors has to do
  int action;           // CA_STOP: x is the result; CA_STOP_RESTORE: x := t first; CA_RETRY: x := t + stp d;
                        // CA_DIRECTION: store the pair, build d, x := x + stp d
  int store_pair, restore, slot, cnew, col, head, shift;   // shift: the oldest pair was dropped to make room (col == m)
  int phantom;          // restore after a trial the reference would not have evaluated (g'd >= 0): it is not "the previous evaluation"
  double stp, stp_prev, gamma;
};

struct CompactMats {    // the m x m part (LDS; every workgroup keeps a replica)
  double SY[TEAM_MCAP * TEAM_MCAP];   // s_i'y_k, chronological, i <= k used
  double YY[TEAM_MCAP * TEAM_MCAP];   // y_i'y_k
  double ap[TEAM_MCAP], bp[TEAM_MCAP];   // S'g_k, Y'g_k at the last accepted iterate
  double u[TEAM_MCAP], q[TEAM_MCAP];
  double sc[4];
};

// Called by every thread of a workgroup with identical arguments. Contains one __syncthreads() on the
// CA_DIRECTION path.
__device__ __forceinline__ void compact_advance(CompactState& S, const double* acc /* [TEAM_K], registers or LDS */, double f_new,
                                                const SolveParams& o, CompactMats& L, CompactPlan& plan, bool counted = true,
                                                const double* sg = nullptr /* [2 MCAP]: S_i'g, Y_i'g taken directly (chronological), or none */) {
  const int m = o.m;
  counted = counted || S.first;
  S.nfev += counted ? 1 : 0;
  const double gd = acc[1], gg = acc[2], rr = acc[3], yg = acc[4];
  bool restore = false, store_pair = false, shift = false, descent_lost = false;
  double dr = 0.0;
  const double stp_prev = S.stp;
  plan.store_pair = 0; plan.restore = 0; plan.slot = 0; plan.cnew = 0; plan.shift = 0; plan.stp_prev = stp_prev; plan.phantom = 0;
  if (S.first) {
    S.first = 0;
    S.f = f_new;
    S.sbgnrm = acc[TEAM_K - 1];
    if (S.sbgnrm <= o.pgtol) { S.status = 0; plan.action = CA_STOP; return; }
  } else {
    S.f = f_new;
    double stp = S.stp;
    if (S.ifun == 1) {   // first trial of this search: the slope at its start, exactly
      const double g0 = acc[TEAM_RD];
      S.ls.ginit = g0; S.ls.gtest = LS_FTOL * g0; S.ls.gx = g0; S.ls.gy = g0;
      S.gdold = g0;
      if (g0 >= 0.0) {   // not a descent direction after all (lnsrlb info = -4; it would not have evaluated this trial)
        S.nfev -= counted ? 1 : 0;
        descent_lost = true;
      }
    }
    const int task = descent_lost ? LS_FG : dcsrch_step(S.ls, f_new, gd, stp);
    S.stp = stp;
    if (descent_lost) {
      restore = true;
    } else if (task == LS_FG) {
      ++S.ifun;
      if (S.ifun - 1 < o.maxls) { plan.action = CA_RETRY; plan.stp = stp; return; }
      restore = true;   // iback >= maxls: back to the last iterate, forget the history
    } else {
      ++S.nit;
      S.iter0 = 0;
      S.sbgnrm = acc[TEAM_K - 1];
      if (S.nit >= o.max_iter) { S.status = 2; plan.action = CA_STOP; return; }
      if (S.nfev > o.maxfun) { S.status = 3; plan.action = CA_STOP; return; }
      if (S.sbgnrm <= o.pgtol) { S.status = 0; plan.action = CA_STOP; return; }
      {
        const double ddum = fmax(fabs(S.fold), fmax(fabs(S.f), 1.0));
        if (S.fold - S.f <= o.ftol * ddum) { S.status = 1; plan.action = CA_STOP; return; }
      }
      double ddum;
      if (stp == 1.0) { dr = gd - S.gdold; ddum = -S.gdold; }
      else { dr = (gd - S.gdold) * stp; ddum = -S.gdold * stp; }
      store_pair = dr > EPSMCH * ddum;
    }
  }
  // ---- new search direction
  int slot = 0;          // history slot of the pair being stored
  double gg_cur = gg;    // g'g of the gradient the direction is built from
  if (restore) {
    if (S.col == 0) { S.f = S.fold; S.status = 4; plan.action = CA_STOP_RESTORE; return; }
    S.col = 0; S.head = 0; S.theta = 1.0;
    S.f = S.fold;
    gg_cur = S.gg_k;
  }
  int col = S.col, head = S.head;
  double theta = S.theta;
  if (store_pair) {
    if (col < m) { slot = head + col; if (slot >= m) slot -= m; ++col; }
    else { slot = head; ++head; if (head >= m) head = 0; shift = true; }
    theta = rr / dr;
  }
  const int cnew = col - 1;   // chronological index of the stored pair
  // The small dense part, by the first wavefront of every workgroup (each workgroup keeps its own replica in
  // LDS): lane i owns row i of the m x m matrices; the triangular solves broadcast one unknown per step.
  if (threadIdx.x < WAVE) {
    const int i = (int)threadIdx.x;
    constexpr int MM = TEAM_MCAP * TEAM_MCAP;
    if (store_pair && shift) {   // drop the oldest pair: (r, c) <- (r + 1, c + 1)
      double sy[2], yy[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int idx = i + h * WAVE;
        const int r = idx / TEAM_MCAP, c = idx - r * TEAM_MCAP;
        const bool ok = idx < MM && r + 1 < m && c + 1 < m;
        sy[h] = ok ? L.SY[(r + 1) * TEAM_MCAP + c + 1] : 0.0;
        yy[h] = ok ? L.YY[(r + 1) * TEAM_MCAP + c + 1] : 0.0;
      }
      const double pa = (i + 1 < m) ? L.ap[i + 1] : 0.0, pb = (i + 1 < m) ? L.bp[i + 1] : 0.0;
      wave_lds_fence();
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int idx = i + h * WAVE;
        const int r = idx / TEAM_MCAP, c = idx - r * TEAM_MCAP;
        if (idx < MM && r + 1 < m && c + 1 < m) { L.SY[idx] = sy[h]; L.YY[idx] = yy[h]; }
      }
      if (i + 1 < m) { L.ap[i] = pa; L.bp[i] = pb; }
      wave_lds_fence();
    }
    // s_i'y, y_i'y of row i in chronological order after the shift (y = g - g_k); s_i'g, y_i'g likewise where the caller has them
    double sy_i = 0.0, yy_i = 0.0, sg_i = 0.0, yg_i = 0.0;
#pragma unroll
    for (int k = 0; k < TEAM_MCAP; ++k) {
      if (i == k) {
        if (shift) {
          if (k + 1 < TEAM_MCAP) {
            sy_i = acc[5 + (k + 1 < TEAM_MCAP ? k + 1 : k)]; yy_i = acc[5 + TEAM_MCAP + (k + 1 < TEAM_MCAP ? k + 1 : k)];
            if (sg) { sg_i = sg[k + 1 < TEAM_MCAP ? k + 1 : k]; yg_i = sg[TEAM_MCAP + (k + 1 < TEAM_MCAP ? k + 1 : k)]; }
          }
        } else {
          sy_i = acc[5 + k]; yy_i = acc[5 + TEAM_MCAP + k];
          if (sg) { sg_i = sg[k]; yg_i = sg[TEAM_MCAP + k]; }
        }
      }
    }
    // S'g, Y'g at the new gradient: taken directly, or the stored products at g_k plus the products with y; the new pair in closed form
    const int old_rows = store_pair ? cnew : col;
    double ai = 0.0, bi = 0.0;
    if (i < old_rows && !restore) { ai = sg ? sg_i : L.ap[i] + sy_i; bi = sg ? yg_i : L.bp[i] + yy_i; }
    if (store_pair) {
      if (i < cnew) {
        L.SY[i * TEAM_MCAP + cnew] = sy_i;
        L.YY[i * TEAM_MCAP + cnew] = yy_i;
        L.YY[cnew * TEAM_MCAP + i] = yy_i;
      } else if (i == cnew) {
        L.SY[cnew * TEAM_MCAP + cnew] = dr;
        L.YY[cnew * TEAM_MCAP + cnew] = rr;
        ai = stp_prev * gd;   // s'g,  s = stp d
        bi = yg;              // y'g
      }
      wave_lds_fence();
    }
    const bool row = i < col;
    const int ir = row ? i : 0;
    double Rrow[TEAM_MCAP], Rcol[TEAM_MCAP], Yrow[TEAM_MCAP];
#pragma unroll
    for (int k = 0; k < TEAM_MCAP; ++k) {
      const bool ok = row && k < col;
      Rrow[k] = ok ? L.SY[ir * TEAM_MCAP + k] : 0.0;
      Rcol[k] = ok ? L.SY[k * TEAM_MCAP + ir] : 0.0;
      Yrow[k] = ok ? L.YY[ir * TEAM_MCAP + k] : 0.0;
    }
    double diag = 1.0;
#pragma unroll
    for (int k = 0; k < TEAM_MCAP; ++k)
      if (i == k && row) diag = Rrow[k];
    const double rdiag = 1.0 / diag;
    const double gamma = 1.0 / theta;
    // q = R^-1 a, last unknown first
    double qk[TEAM_MCAP], uk[TEAM_MCAP];
    double sv = row ? ai : 0.0, myq = 0.0, myu = 0.0;
#pragma unroll
    for (int k = TEAM_MCAP - 1; k >= 0; --k) {
      qk[k] = 0.0;
      if (k < col) {
        const double cand = sv * rdiag;
        qk[k] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(cand), k),
                                 __builtin_amdgcn_readlane(__double2loint(cand), k));
        if (i == k) myq = qk[k];
        if (i < k) sv -= Rrow[k] * qk[k];
      }
    }
    // u = R^-T ((D + gamma Y'Y) q - gamma b), first unknown first
    double tv = diag * myq - gamma * bi;
#pragma unroll
    for (int k = 0; k < TEAM_MCAP; ++k) tv += gamma * Yrow[k] * qk[k];
    if (!row) tv = 0.0;
#pragma unroll
    for (int k = 0; k < TEAM_MCAP; ++k) {
      uk[k] = 0.0;
      if (k < col) {
        const double cand = tv * rdiag;
        uk[k] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(cand), k),
                                 __builtin_amdgcn_readlane(__double2loint(cand), k));
        if (i == k) myu = uk[k];
        if (i > k) tv -= Rcol[k] * uk[k];
      }
    }
    const double term = row ? (gamma * bi * myq - ai * myu) : 0.0;
    const double gdn0 = wave_sum(term) - gg_cur * (col > 0 ? gamma : 1.0);
    if (row) { L.ap[i] = ai; L.bp[i] = bi; L.q[i] = myq; L.u[i] = myu; }
    if (i == 0) L.sc[0] = gdn0;
  }
  __syncthreads();
  const double gdn = L.sc[0];   // algebraic -g'Hg: a stand-in until r'd arrives with the first trial (see TEAM_RD above), which
                                // also takes the "not a descent direction" decision (lnsrlb info = -4)
  S.col = col; S.head = head; S.theta = theta;
  S.gg_k = gg_cur;
  S.gdold = gdn;
  S.fold = S.f;
  S.stp = S.iter0 ? fmin(1.0 / sqrt(gg_cur), LS_STPMAX) : 1.0;
  dcsrch_start(S.ls, S.f, gdn, S.stp);
  S.ifun = 1;
  plan.action = CA_DIRECTION;
  plan.store_pair = store_pair ? 1 : 0;
  plan.restore = restore ? 1 : 0;
  plan.phantom = descent_lost ? 1 : 0;
  plan.slot = slot;
  plan.cnew = cnew;
  plan.shift = (store_pair && shift) ? 1 : 0;
  plan.col = col;
  plan.head = head;
  plan.stp = S.stp;
  plan.gamma = 1.0 / theta;
}

// Dispatch on the number of stored pairs (uniform over the cooperating threads): BODY sees a constexpr int HC = the count.
// The passes over the history request every pair of a coefficient before the first is used (one memory latency per pass
// instead of one per pair); with the count a compile-time constant that stays true while only the pairs that exist are
// requested — the mean history length of a ~10-iteration solve is 4.5 of the 10 slots.
#define GDMIX_HIST_CASE(N_, BODY_) case N_: { constexpr int HC = N_; BODY_; } break;
#define GDMIX_HIST_DISPATCH(COUNT_, BODY_)                                                                        \
  switch (COUNT_) {                                                                                               \
    GDMIX_HIST_CASE(0, BODY_) GDMIX_HIST_CASE(1, BODY_) GDMIX_HIST_CASE(2, BODY_) GDMIX_HIST_CASE(3, BODY_)       \
    GDMIX_HIST_CASE(4, BODY_) GDMIX_HIST_CASE(5, BODY_) GDMIX_HIST_CASE(6, BODY_) GDMIX_HIST_CASE(7, BODY_)       \
    GDMIX_HIST_CASE(8, BODY_) GDMIX_HIST_CASE(9, BODY_)                                                           \
    default: { constexpr int HC = TEAM_MCAP; BODY_; } break;                                                      \
  }
static_assert(TEAM_MCAP == 10, "GDMIX_HIST_DISPATCH lists the counts 0..10");

// The elementwise part of a step for coefficient j (CA_RETRY / CA_DIRECTION). Vectors as in Work; u, q from L.
// HC >= plan.col: the number of history slots requested (GDMIX_HIST_DISPATCH(plan.col, ...) makes it equal).
// moved (may be NULL): set to 1 if the new trial's coefficient j differs from the previously evaluated point's (after a
// restore: from the restored iterate's, or if the abandoned search's last trial had moved it). A store, not a return value:
// carrying the flag out of the count-dispatched loops in a register made the team kernel spill 36 VGPRs.
template <int HC>
__device__ __forceinline__ void compact_update_n(const CompactPlan& plan, const CompactMats& L, const Work& W, int p, int m, int j,
                                                 unsigned* moved = nullptr) {
  if (plan.action == CA_RETRY) {
    const double xn = plan.stp * W.d[j] + W.t[j];
    if (moved && xn != W.x[j]) *moved = 1u;
    W.x[j] = xn;
    return;
  }
  const double gj = plan.restore ? W.r[j] : W.g[j];
  const double xj = plan.restore ? W.t[j] : W.x[j];
  const bool failed_off = plan.restore && !plan.phantom && W.x[j] != W.t[j];
  double sn = 0.0, yn = 0.0;
  if (plan.store_pair) {
    sn = plan.stp_prev * W.d[j];   // exact for stp == 1
    yn = W.g[j] - W.r[j];
    compact_hist(W, m, j)[plan.slot * COMPACT_HIST_STRIDE] = make_double2(sn, yn);
  }
  double dj = -gj;
  if (HC > 0 && plan.col > 0) {
    // every pair is requested before the first is used (see team_eval)
    double2 h[HC > 0 ? HC : 1];
#pragma unroll
    for (int i = 0; i < HC; ++i) {
      int sl = plan.head + i;
      if (sl >= m) sl -= m;
      if (i >= m) sl = 0;
      h[i] = compact_hist(W, m, j)[sl * COMPACT_HIST_STRIDE];
    }
    double su = 0.0, yq = 0.0;
#pragma unroll
    for (int i = 0; i < HC; ++i) {
      if (i < plan.col) {
        const bool fresh = plan.store_pair && i == plan.cnew;
        const double si = fresh ? sn : h[i].x;
        const double yi = fresh ? yn : h[i].y;
        su += L.u[i] * si;
        yq += L.q[i] * yi;
      }
    }
    dj = plan.gamma * (yq - gj) - su;
  }
  const double z = xj + dj;   // mainlb re-derives d from the subspace point
  dj = z - xj;
  W.d[j] = dj;
  W.t[j] = xj;
  W.r[j] = gj;
  const double xn = plan.stp * dj + xj;
  W.x[j] = xn;
  if (moved && (xn != xj || failed_off)) *moved = 1u;
}

__device__ __forceinline__ void compact_update(const CompactPlan& plan, const CompactMats& L, const Work& W, int p, int m, int j) {
  compact_update_n<TEAM_MCAP>(plan, L, W, p, m, j);
}

}  // namespace gdmix
