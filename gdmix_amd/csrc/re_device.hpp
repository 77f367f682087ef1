// re_device.hpp — device-side building blocks of the gfx950 random-effect solver.
//
// Wave64 only (CDNA4). Cross-lane reductions use DPP row shifts + row broadcasts and land in SGPRs
// through v_readlane, so every reduction result is wave-uniform and bit-identical in all lanes: the
// whole L-BFGS control flow (line search, stop tests) is then uniform by construction.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gdmix {

constexpr int WAVE = 64;

// ---- DPP helpers -------------------------------------------------------------------------------
// dpp_ctrl encodings (GFX9): row_shr:n = 0x110+n, row_bcast:15 = 0x142, row_bcast:31 = 0x143.
// Value of the DPP source lane, 0.0 where the source lane does not exist (bound_ctrl). All rows are
// enabled: the reductions below only guarantee their result in lane 63 (what readlane63 reads), so the
// row broadcasts need no row mask and the destination needs no zero pre-load (ROW_MASK is kept as
// documentation of which rows matter).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_get0(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  int lo2 = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);
  int hi2 = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi2, lo2);
}

__device__ __forceinline__ double readlane63(double v) {
  int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
  return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double readlane0(double v) {
  int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
  int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
  return __hiloint2double(hi, lo);
}

// Sum over the 64 lanes of a wave; all lanes must be active. Fixed association order.
__device__ __forceinline__ double wave_sum(double v) {
  v += dpp_get0<0x111, 0xf>(v);   // row_shr:1
  v += dpp_get0<0x112, 0xf>(v);   // row_shr:2
  v += dpp_get0<0x114, 0xf>(v);   // row_shr:4
  v += dpp_get0<0x118, 0xf>(v);   // row_shr:8  -> lane 15 of each row holds the row sum
  v += dpp_get0<0x142, 0xa>(v);   // row_bcast:15 into rows 1,3
  v += dpp_get0<0x143, 0xc>(v);   // row_bcast:31 into rows 2,3 -> lane 63 holds the total
  return readlane63(v);
}

// Max over the wave of NON-NEGATIVE values (0 is the identity used for invalid DPP sources).
__device__ __forceinline__ double wave_max_nonneg(double v) {
  v = fmax(v, dpp_get0<0x111, 0xf>(v));
  v = fmax(v, dpp_get0<0x112, 0xf>(v));
  v = fmax(v, dpp_get0<0x114, 0xf>(v));
  v = fmax(v, dpp_get0<0x118, 0xf>(v));
  v = fmax(v, dpp_get0<0x142, 0xa>(v));
  v = fmax(v, dpp_get0<0x143, 0xc>(v));
  return readlane63(v);
}

// Max over the wave of non-negative ints, as a uniform value (all lanes must be active).
__device__ __forceinline__ int wave_max_nonneg_i32(int v) {
  v = max(v, __builtin_amdgcn_mov_dpp(v, 0x111, 0xf, 0xf, true));
  v = max(v, __builtin_amdgcn_mov_dpp(v, 0x112, 0xf, 0xf, true));
  v = max(v, __builtin_amdgcn_mov_dpp(v, 0x114, 0xf, 0xf, true));
  v = max(v, __builtin_amdgcn_mov_dpp(v, 0x118, 0xf, 0xf, true));
  v = max(v, __builtin_amdgcn_mov_dpp(v, 0x142, 0xf, 0xf, true));
  v = max(v, __builtin_amdgcn_mov_dpp(v, 0x143, 0xf, 0xf, true));
  return __builtin_amdgcn_readlane(v, 63);
}

// Compiler-level ordering point between LDS accesses of different lanes of ONE wave. The LDS queue
// of a wave is processed in order, so no hardware wait is needed; the fences only stop the compiler
// from moving memory operations across this point.
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Same, for data that may live in global memory (HBM scratch): additionally waits for the wave's own
// outstanding stores so that another lane's later load observes them (same CU, same L1).
__device__ __forceinline__ void wave_mem_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// Loads / stores of the vectors the workgroups of a team exchange (x, the per-sample residuals, partial sums). SC1 = true:
// relaxed agent-scope accesses (global_load / global_store ... sc1): the store is written through to the memory side, the
// load bypasses this CU's L1, so the hand-off needs no cache write-back / invalidate, only a drained store queue
// (s_waitcnt vmcnt(0)) ahead of the barrier arrival (MI355X_MICROARCH.md, "valid forms": sc1 stores and loads on both sides).
template <bool SC1>
__device__ __forceinline__ double ld_x(const double* p) {
  if (SC1)
    return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<unsigned long long*>(const_cast<double*>(p)),
                                                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  return *p;
}
template <bool SC1>
__device__ __forceinline__ void st_x(double* p, double v) {
  if (SC1)
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
  else
    *p = v;
}

// ---- thread groups ------------------------------------------------------------------------------
// One wavefront cooperating on one entity.
struct WaveGroup {
  int tid;
  static constexpr int NT = WAVE;
  __device__ __forceinline__ double sum(double v) const { return wave_sum(v); }
  __device__ __forceinline__ double max_nonneg(double v) const { return wave_max_nonneg(v); }
  __device__ __forceinline__ void sync() const { wave_lds_fence(); }
};

// One workgroup of NW wavefronts cooperating on one (large) entity.
template <int NW>
struct BlockGroup {
  int tid;
  double* red;   // LDS, 2*NW doubles (double-buffered so one barrier per reduction suffices)
  int phase;
  static constexpr int NT = WAVE * NW;
  __device__ __forceinline__ double combine(double v, bool is_max) {
    double* buf = red + phase * NW;
    phase ^= 1;
    if ((tid & (WAVE - 1)) == 0) buf[tid >> 6] = v;
    __syncthreads();
    double s = buf[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) s = is_max ? fmax(s, buf[w]) : s + buf[w];
    return s;
  }
  __device__ __forceinline__ double sum(double v) { return combine(wave_sum(v), false); }
  __device__ __forceinline__ double max_nonneg(double v) { return combine(wave_max_nonneg(v), true); }
  __device__ __forceinline__ void sync() const { __syncthreads(); }
};

// ---- per-sample logistic terms --------------------------------------------------------------------------------
// ce  = max(z,0) - z*y + log(1 + exp(-|z|))            (binary_logistic_regression.py:103)
// sig = expit(z) = 1/(1 + exp(-z))                      (:45-51)
// One exp, one log over [1,2] and two Newton-refined reciprocals, written out instead of calling the
// device libm (ocml exp + log + IEEE division: 158 VALU instructions per sample; this: ~75), same
// semantics as the reference's formula: u = fl(1 + e) is formed first, then log(u) and 1/u.
// Accuracy (checked against long double on 4e6 points, tools/softplus_check.c): exp <= 0.98 ulp,
// log(u) <= 1.5 ulp, reciprocal <= 0.5 ulp.
__device__ __forceinline__ double rcp_nr(double u) {
  double y = __builtin_amdgcn_rcp(u);
  double e = __builtin_fma(-u, y, 1.0);
  y = __builtin_fma(y, e, y);
  e = __builtin_fma(-u, y, 1.0);
  y = __builtin_fma(y, e, y);
  return y;
}

// Horner steps p = fma(p, x, C) with the coefficient in a SCALAR register pair (GDMIX_SGPR_POLY=1; round 6, measured, off). Written plainly,
// the compiler turns each step into v_fmac_f64 with the coefficient moved into the destination first: two v_mov_b32 per step, 38 vector
// instructions per sample's logistic terms in kernels whose vector issue is 67 % busy (profiles/r06_inst_mix.txt). With the switch the
// coefficient goes into vcc by two s_mov_b32 and v_fma_f64 takes it as its one constant-bus operand: same operations on the same numbers, same
// bits (346 parity tests) — and the same time (C2 8.97 vs 8.95 ms, MovieLens per user 8.42 vs 8.45, same box: profiles/r06_c2_ab.txt, session F).
// The moves sat in the shadow of the chain's own latency: a Horner step waits for the step before it either way.
#ifndef GDMIX_SGPR_POLY
#define GDMIX_SGPR_POLY 0
#endif
#define GDMIX_HSTEP(K_) "s_mov_b32 vcc_lo, %[l" #K_ "]\n\ts_mov_b32 vcc_hi, %[h" #K_ "]\n\tv_fma_f64 %[p], %[p], %[x], vcc\n\t"
#define GDMIX_HCONST(K_, C_) [l##K_] "i"((unsigned)(__builtin_bit_cast(unsigned long long, (double)(C_)) & 0xffffffffull)), \
                             [h##K_] "i"((unsigned)(__builtin_bit_cast(unsigned long long, (double)(C_)) >> 32))

__device__ __forceinline__ double exp_neg(double a) {   // exp(-a) for a >= 0
  a = fmin(a, 800.0);
  const double kf = __builtin_rint(a * 1.4426950408889634074);
  double r = __builtin_fma(kf, 6.93147180369123816490e-01, -a);
  r = __builtin_fma(kf, 1.90821492927058770002e-10, r);   // r = k ln2 - a, |r| <= ln2/2
  double p = 1.0 / 6227020800.0;
#if GDMIX_SGPR_POLY
  asm(GDMIX_HSTEP(0) GDMIX_HSTEP(1) GDMIX_HSTEP(2) GDMIX_HSTEP(3) GDMIX_HSTEP(4) GDMIX_HSTEP(5) GDMIX_HSTEP(6) GDMIX_HSTEP(7) GDMIX_HSTEP(8) GDMIX_HSTEP(9)
      : [p] "+v"(p)
      : [x] "v"(r), GDMIX_HCONST(0, 1.0 / 479001600.0), GDMIX_HCONST(1, 1.0 / 39916800.0), GDMIX_HCONST(2, 1.0 / 3628800.0), GDMIX_HCONST(3, 1.0 / 362880.0), GDMIX_HCONST(4, 1.0 / 40320.0), GDMIX_HCONST(5, 1.0 / 5040.0), GDMIX_HCONST(6, 1.0 / 720.0), GDMIX_HCONST(7, 1.0 / 120.0), GDMIX_HCONST(8, 1.0 / 24.0), GDMIX_HCONST(9, 1.0 / 6.0)
      : "vcc");
  p = __builtin_fma(p, r, 0.5);
#else
  p = __builtin_fma(p, r, 1.0 / 479001600.0);
  p = __builtin_fma(p, r, 1.0 / 39916800.0);
  p = __builtin_fma(p, r, 1.0 / 3628800.0);
  p = __builtin_fma(p, r, 1.0 / 362880.0);
  p = __builtin_fma(p, r, 1.0 / 40320.0);
  p = __builtin_fma(p, r, 1.0 / 5040.0);
  p = __builtin_fma(p, r, 1.0 / 720.0);
  p = __builtin_fma(p, r, 1.0 / 120.0);
  p = __builtin_fma(p, r, 1.0 / 24.0);
  p = __builtin_fma(p, r, 1.0 / 6.0);
  p = __builtin_fma(p, r, 0.5);
#endif
  p = __builtin_fma(p, r * r, r) + 1.0;
  return __builtin_amdgcn_ldexp(p, -(int)kf);
}

__device__ __forceinline__ double log_1_2(double u) {   // log(u) for u in [1, 2]
  const bool k = u > 1.4142135623730951;
  const double m = k ? 0.5 * u : u;
  const double num = m - 1.0, den = m + 1.0;
  const double y = rcp_nr(den);
  const double s = num * y;
  const double slo = __builtin_fma(-s, den, num) * y;    // quotient residual, kept as a low part
  const double s2 = s * s;
  double p = 2.0 / 21.0;
#if GDMIX_SGPR_POLY
  asm(GDMIX_HSTEP(0) GDMIX_HSTEP(1) GDMIX_HSTEP(2) GDMIX_HSTEP(3) GDMIX_HSTEP(4) GDMIX_HSTEP(5) GDMIX_HSTEP(6) GDMIX_HSTEP(7) GDMIX_HSTEP(8)
      : [p] "+v"(p)
      : [x] "v"(s2), GDMIX_HCONST(0, 2.0 / 19.0), GDMIX_HCONST(1, 2.0 / 17.0), GDMIX_HCONST(2, 2.0 / 15.0), GDMIX_HCONST(3, 2.0 / 13.0), GDMIX_HCONST(4, 2.0 / 11.0), GDMIX_HCONST(5, 2.0 / 9.0), GDMIX_HCONST(6, 2.0 / 7.0), GDMIX_HCONST(7, 2.0 / 5.0), GDMIX_HCONST(8, 2.0 / 3.0)
      : "vcc");
#else
  p = __builtin_fma(p, s2, 2.0 / 19.0);
  p = __builtin_fma(p, s2, 2.0 / 17.0);
  p = __builtin_fma(p, s2, 2.0 / 15.0);
  p = __builtin_fma(p, s2, 2.0 / 13.0);
  p = __builtin_fma(p, s2, 2.0 / 11.0);
  p = __builtin_fma(p, s2, 2.0 / 9.0);
  p = __builtin_fma(p, s2, 2.0 / 7.0);
  p = __builtin_fma(p, s2, 2.0 / 5.0);
  p = __builtin_fma(p, s2, 2.0 / 3.0);
#endif
  const double lo = __builtin_fma(s * s2, p, (k ? 1.90821492927058770002e-10 : 0.0) + 2.0 * slo);
  const double res = 2.0 * s + lo;
  return k ? res + 6.93147180369123816490e-01 : res;
}

// returns w * ce, writes r = w * (sigma(z) - y)
__device__ __forceinline__ double logistic_terms(double z, double yi, double wi, double& ri) {
#ifdef GDMIX_LIBM_MATH
  const double e = exp(-fabs(z));
  const double ce = fmax(z, 0.0) - z * yi + log(1.0 + e);
  const double sig = (z >= 0.0) ? 1.0 / (1.0 + e) : e / (1.0 + e);
#else
  const double e = exp_neg(fabs(z));
  const double u = 1.0 + e;
  const double ce = fmax(z, 0.0) - z * yi + log_1_2(u);
  const double ru = rcp_nr(u);
  const double sig = (z >= 0.0) ? ru : e * ru;   // z < 0: e/(1+e), no overflow
#endif
  ri = wi * (sig - yi);
  return wi * ce;
}

// logistic variance weight rho (1 - rho) of _compute_variance (binary_logistic_regression.py:167-168)
__device__ __forceinline__ double sigmoid_full(double z) { return 1.0 / (1.0 + exp(-z)); }

// ---- More'-Thuente line search (MINPACK-2 dcsrch/dcstep as L-BFGS-B 3.0's lnsrlb calls it) --------
// All state is uniform across the cooperating threads.
struct LineSearch {
  double ginit, gtest, gx, gy, finit, fx, fy, stx, sty, stmin, stmax, width, width1;
  int brackt, stage;
};

constexpr double LS_FTOL = 1.0e-3, LS_GTOL = 0.9, LS_XTOL = 0.1, LS_STPMIN = 0.0, LS_STPMAX = 1.0e10;
enum { LS_FG = 0, LS_CONV = 1, LS_WARN = 2 };

__device__ __forceinline__ void dcsrch_start(LineSearch& S, double f, double g, double stp) {
  S.brackt = 0; S.stage = 1;
  S.finit = f; S.ginit = g; S.gtest = LS_FTOL * g;
  S.width = LS_STPMAX - LS_STPMIN; S.width1 = S.width / 0.5;
  S.stx = 0.0; S.fx = f; S.gx = g;
  S.sty = 0.0; S.fy = f; S.gy = g;
  S.stmin = 0.0; S.stmax = stp + 4.0 * stp;
}

// One dcstep: safeguarded cubic/quadratic step and interval update. Everything is passed and returned by
// value (no references, no conditional stores through pointers: those made the compiler materialise the
// interval in scratch memory).
struct StepIO {
  double stx, fx, dx, sty, fy, dy, stp;
  int brackt;
};

__device__ __forceinline__ StepIO dcstep(StepIO v, double fp, double dp, double stpmin, double stpmax) {
  const double stx = v.stx, fx = v.fx, dx = v.dx, sty = v.sty, fy = v.fy, dy = v.dy, stp = v.stp;
  int brackt = v.brackt;
  double gamma, p, q, r, s, stpc, stpf, stpq, theta;
  const double sgnd = dp * (dx / fabs(dx));
  if (fp > fx) {
    theta = 3.0 * (fx - fp) / (stp - stx) + dx + dp;
    s = fmax(fabs(theta), fmax(fabs(dx), fabs(dp)));
    gamma = s * sqrt((theta / s) * (theta / s) - (dx / s) * (dp / s));
    if (stp < stx) gamma = -gamma;
    p = (gamma - dx) + theta;
    q = ((gamma - dx) + gamma) + dp;
    r = p / q;
    stpc = stx + r * (stp - stx);
    stpq = stx + ((dx / ((fx - fp) / (stp - stx) + dx)) / 2.0) * (stp - stx);
    if (fabs(stpc - stx) < fabs(stpq - stx)) stpf = stpc;
    else stpf = stpc + (stpq - stpc) / 2.0;
    brackt = 1;
  } else if (sgnd < 0.0) {
    theta = 3.0 * (fx - fp) / (stp - stx) + dx + dp;
    s = fmax(fabs(theta), fmax(fabs(dx), fabs(dp)));
    gamma = s * sqrt((theta / s) * (theta / s) - (dx / s) * (dp / s));
    if (stp > stx) gamma = -gamma;
    p = (gamma - dp) + theta;
    q = ((gamma - dp) + gamma) + dx;
    r = p / q;
    stpc = stp + r * (stx - stp);
    stpq = stp + (dp / (dp - dx)) * (stx - stp);
    if (fabs(stpc - stp) > fabs(stpq - stp)) stpf = stpc;
    else stpf = stpq;
    brackt = 1;
  } else if (fabs(dp) < fabs(dx)) {
    theta = 3.0 * (fx - fp) / (stp - stx) + dx + dp;
    s = fmax(fabs(theta), fmax(fabs(dx), fabs(dp)));
    gamma = s * sqrt(fmax(0.0, (theta / s) * (theta / s) - (dx / s) * (dp / s)));
    if (stp > stx) gamma = -gamma;
    p = (gamma - dp) + theta;
    q = (gamma + (dx - dp)) + gamma;
    r = p / q;
    if (r < 0.0 && gamma != 0.0) stpc = stp + r * (stx - stp);
    else if (stp > stx) stpc = stpmax;
    else stpc = stpmin;
    stpq = stp + (dp / (dp - dx)) * (stx - stp);
    if (brackt) {
      if (fabs(stpc - stp) < fabs(stpq - stp)) stpf = stpc;
      else stpf = stpq;
      if (stp > stx) stpf = fmin(stp + 0.66 * (sty - stp), stpf);
      else stpf = fmax(stp + 0.66 * (sty - stp), stpf);
    } else {
      if (fabs(stpc - stp) > fabs(stpq - stp)) stpf = stpc;
      else stpf = stpq;
      stpf = fmin(stpmax, stpf);
      stpf = fmax(stpmin, stpf);
    }
  } else {
    if (brackt) {
      theta = 3.0 * (fp - fy) / (sty - stp) + dy + dp;
      s = fmax(fabs(theta), fmax(fabs(dy), fabs(dp)));
      gamma = s * sqrt((theta / s) * (theta / s) - (dy / s) * (dp / s));
      if (stp > sty) gamma = -gamma;
      p = (gamma - dp) + theta;
      q = ((gamma - dp) + gamma) + dy;
      r = p / q;
      stpc = stp + r * (sty - stp);
      stpf = stpc;
    } else if (stp > stx) stpf = stpmax;
    else stpf = stpmin;
  }
  // interval update, written as selects
  const bool worse = fp > fx;
  const bool flip = !worse && (sgnd < 0.0);
  StepIO o;
  o.sty = worse ? stp : (flip ? stx : sty);
  o.fy = worse ? fp : (flip ? fx : fy);
  o.dy = worse ? dp : (flip ? dx : dy);
  o.stx = worse ? stx : stp;
  o.fx = worse ? fx : fp;
  o.dx = worse ? dx : dp;
  o.stp = stpf;
  o.brackt = brackt;
  return o;
}

// dcsrch's convergence test (the strong Wolfe conditions), which decides a search whatever its warnings say (it is the last
// assignment of `task` in dcsrch): callers that hold only these three numbers of the state can ask it first.
__device__ __forceinline__ bool dcsrch_converged(double finit, double gtest, double ginit, double f, double g, double stp) {
  const double ftest = finit + stp * gtest;
  return f <= ftest && fabs(g) <= LS_GTOL * (-ginit);
}

__device__ __forceinline__ int dcsrch_step(LineSearch& S, double f, double g, double& stp_io) {
  double stp = stp_io;
  int task = LS_FG;
  const double ftest = S.finit + stp * S.gtest;
  if (S.stage == 1 && f <= ftest && g >= 0.0) S.stage = 2;
  if (S.brackt && (stp <= S.stmin || stp >= S.stmax)) task = LS_WARN;
  if (S.brackt && S.stmax - S.stmin <= LS_XTOL * S.stmax) task = LS_WARN;
  if (stp == LS_STPMAX && f <= ftest && g <= S.gtest) task = LS_WARN;
  if (stp == LS_STPMIN && (f > ftest || g >= S.gtest)) task = LS_WARN;
  if (dcsrch_converged(S.finit, S.gtest, S.ginit, f, g, stp)) task = LS_CONV;
  if (task != LS_FG) return task;
  // dcsrch calls dcstep either on the function values or, in stage 1 while f > ftest, on the modified
  // function psi(a) = phi(a) - a*gtest. Both forms are one call here: with gt = 0 the subtractions and
  // the add-backs below are exact identities, so the unmodified branch is reproduced bit for bit.
  const double gt = (S.stage == 1 && f <= S.fx && f > ftest) ? S.gtest : 0.0;
  {
    StepIO v;
    v.stx = S.stx; v.sty = S.sty; v.stp = stp; v.brackt = S.brackt;
    v.fx = S.fx - S.stx * gt; v.fy = S.fy - S.sty * gt;
    v.dx = S.gx - gt; v.dy = S.gy - gt;
    const StepIO w = dcstep(v, f - stp * gt, g - gt, S.stmin, S.stmax);
    S.stx = w.stx; S.sty = w.sty; S.brackt = w.brackt; stp = w.stp;
    S.fx = w.fx + w.stx * gt;
    S.fy = w.fy + w.sty * gt;
    S.gx = w.dx + gt;
    S.gy = w.dy + gt;
  }
  if (S.brackt) {
    if (fabs(S.sty - S.stx) >= 0.66 * S.width1) stp = S.stx + 0.5 * (S.sty - S.stx);
    S.width1 = S.width;
    S.width = fabs(S.sty - S.stx);
  }
  if (S.brackt) {
    S.stmin = fmin(S.stx, S.sty);
    S.stmax = fmax(S.stx, S.sty);
  } else {
    S.stmin = stp + 1.1 * (stp - S.stx);
    S.stmax = stp + 4.0 * (stp - S.stx);
  }
  stp = fmax(stp, LS_STPMIN);
  stp = fmin(stp, LS_STPMAX);
  if ((S.brackt && (stp <= S.stmin || stp >= S.stmax)) ||
      (S.brackt && S.stmax - S.stmin <= LS_XTOL * S.stmax))
    stp = S.stx;
  stp_io = stp;
  return LS_FG;
}

}  // namespace gdmix
