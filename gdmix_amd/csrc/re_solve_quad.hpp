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
// Register diet (the (s, y) history alone is 40*EPL VGPRs): x, g, d and the history live in VGPRs;
// x_old, g_old and the per-entity scalars that are only needed between evaluations are parked in the
// row's LDS block. LDS offsets are the same for all four rows (capacity-based layout) so addressing
// needs one base VGPR. Row/column extents of the CSR/CSC copies are cached as packed (start | len << 16)
// registers, and the gathers are issued four at a time before the ordered FMA chain, so that an
// evaluation costs ~3 dependent LDS round trips instead of ~2 per non-zero.
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

// The same exchange through the LDS crossbar (ds_swizzle_b32, bit mode: lane' = lane ^ X inside 32 lanes; no memory is touched):
// 0 vector instructions for the move instead of two v_mov_b32_dpp, at the latency of the LDS queue. quad_perm [1,0,3,2] = ^1,
// [2,3,0,1] = ^2, row_half_mirror = ^7, row_mirror = ^15, the other row of a pair = ^16: the partner lanes of the DPP butterflies
// below, so a sum built this way has the same bits. GDMIX_QUAD_SWZ: 1 = the reductions of several values at once (loss / residual
// sum / |x|^2; g'd / y'y / max|g|; d'd / g'd: their moves overlap), 2 = the one-value reductions of the two-loop recursion as
// well (a dependent chain). A/B: profiles/r06_c2_ab.txt.
#ifndef GDMIX_QUAD_SWZ
#define GDMIX_QUAD_SWZ 0
#endif
template <int X>
__device__ __forceinline__ double swz_xor(double v) {
  constexpr int pat = 0x1f | (X << 10);
  const int lo = __builtin_amdgcn_ds_swizzle(__double2loint(v), pat);
  const int hi = __builtin_amdgcn_ds_swizzle(__double2hiint(v), pat);
  return __hiloint2double(hi, lo);
}

// dpp_ctrl: quad_perm:[1,0,3,2] = 0xB1, quad_perm:[2,3,0,1] = 0x4E, row_half_mirror = 0x141, row_mirror = 0x140
// Butterfly: every lane of the row ends with the same bits (each stage adds the same two partial sums).
__device__ __forceinline__ double row_sum(double v) {
#if GDMIX_QUAD_SWZ & 2
  v += swz_xor<1>(v);
  v += swz_xor<2>(v);
  v += swz_xor<7>(v);
  v += swz_xor<15>(v);
  return v;
#endif
  v += dpp_row<0xB1>(v);
  v += dpp_row<0x4E>(v);
  v += dpp_row<0x141>(v);
  v += dpp_row<0x140>(v);
  return v;
}

__device__ __forceinline__ void row_sum2(double& a, double& b) {
#if GDMIX_QUAD_SWZ & 1
#define GDMIX_SSTEP2(X)                \
  {                                    \
    const double ta = swz_xor<X>(a);   \
    const double tb = swz_xor<X>(b);   \
    a += ta;                           \
    b += tb;                           \
  }
  GDMIX_SSTEP2(1) GDMIX_SSTEP2(2) GDMIX_SSTEP2(7) GDMIX_SSTEP2(15)
#undef GDMIX_SSTEP2
  return;
#endif
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

__device__ __forceinline__ double max_nn(double a, double b) { return (a > b) ? a : b; }

// ---- 32-lane groups (two DPP rows per entity): one more butterfly stage across the row pair ----------
// v_permlane16_swap (gfx950) swaps the odd rows of its first operand with the even rows of the second;
// fed with two copies of v it leaves {r0, r0, r2, r2} and {r1, r1, r3, r3}, whose sum is the pair total
// in all 32 lanes of each pair (same operand order in both rows -> bit-identical).
__device__ __forceinline__ void rowpair_split(double v, double& even, double& odd) {
  const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
  const auto l = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  const auto h = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  even = __hiloint2double((int)h[0], (int)l[0]);
  odd = __hiloint2double((int)h[1], (int)l[1]);
}

// v_permlane32_swap swaps the upper half of its first operand with the lower half of the second: fed with two
// copies of v it leaves {lo, lo} and {hi, hi}; their sum is the wave total in all 64 lanes.
__device__ __forceinline__ void half_split(double v, double& lower, double& upper) {
  const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
  const auto l = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  const auto h = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  lower = __hiloint2double((int)h[0], (int)l[0]);
  upper = __hiloint2double((int)h[1], (int)l[1]);
}

// Scratch of the cross-wave stage of groups wider than a wavefront (G = 64*NW): NW partials per value, double
// buffered so that one barrier per reduction suffices (a wave cannot be more than one reduction ahead).
struct XWave {
  double* buf;   // LDS [2][3][NW]
  int phase;
};

template <int G>
__device__ __forceinline__ double xwave_combine(XWave& X, double v, int slot, bool is_max) {
  constexpr int NW = G / WAVE;
  double* b = X.buf + (X.phase * 3 + slot) * NW;
  if ((threadIdx.x & (WAVE - 1)) == 0) b[threadIdx.x >> 6] = v;
  __syncthreads();
  double r = b[0];
#pragma unroll
  for (int w = 1; w < NW; ++w) r = is_max ? max_nn(r, b[w]) : r + b[w];
  return r;
}

// Sum over the G lanes of an entity; every lane of the group receives the same bits.
template <int G>
__device__ __forceinline__ double grp_sum(double v, XWave& X) {
  v = row_sum(v);
  if (G >= 32 && (GDMIX_QUAD_SWZ & 2)) v += swz_xor<16>(v);
  else if (G >= 32) { double a, b; rowpair_split(v, a, b); v = a + b; }
  if (G >= 64) { double a, b; half_split(v, a, b); v = a + b; }
  if (G > 64) { v = xwave_combine<G>(X, v, 0, false); X.phase ^= 1; }
  return v;
}

template <int G>
__device__ __forceinline__ void grp_sum2(double& a, double& b, XWave& X) {
  row_sum2(a, b);
  if (G >= 32 && (GDMIX_QUAD_SWZ & 1)) {
    const double ta = swz_xor<16>(a), tb = swz_xor<16>(b);
    a += ta;
    b += tb;
  } else if (G >= 32) {
    double a0, a1, b0, b1;
    rowpair_split(a, a0, a1);
    rowpair_split(b, b0, b1);
    a = a0 + a1;
    b = b0 + b1;
  }
  if (G >= 64) {
    double a0, a1, b0, b1;
    half_split(a, a0, a1);
    half_split(b, b0, b1);
    a = a0 + a1;
    b = b0 + b1;
  }
  if (G > 64) {
    constexpr int NW = G / WAVE;
    double* pa = X.buf + (X.phase * 3 + 0) * NW;
    double* pb = X.buf + (X.phase * 3 + 1) * NW;
    if ((threadIdx.x & (WAVE - 1)) == 0) { pa[threadIdx.x >> 6] = a; pb[threadIdx.x >> 6] = b; }
    __syncthreads();
    a = pa[0]; b = pb[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) { a += pa[w]; b += pb[w]; }
    X.phase ^= 1;
  }
}

__device__ __forceinline__ void row_sum3(double& a, double& b, double& c) {
#if GDMIX_QUAD_SWZ & 1
#define GDMIX_SSTEP3(X)                \
  {                                    \
    const double ta = swz_xor<X>(a);   \
    const double tb = swz_xor<X>(b);   \
    const double tc = swz_xor<X>(c);   \
    a += ta;                           \
    b += tb;                           \
    c += tc;                           \
  }
  GDMIX_SSTEP3(1) GDMIX_SSTEP3(2) GDMIX_SSTEP3(7) GDMIX_SSTEP3(15)
#undef GDMIX_SSTEP3
  return;
#endif
#define GDMIX_RSTEP3S(CTRL)             \
  {                                     \
    const double ta = dpp_row<CTRL>(a); \
    const double tb = dpp_row<CTRL>(b); \
    const double tc = dpp_row<CTRL>(c); \
    a += ta;                            \
    b += tb;                            \
    c += tc;                            \
  }
  GDMIX_RSTEP3S(0xB1) GDMIX_RSTEP3S(0x4E) GDMIX_RSTEP3S(0x141) GDMIX_RSTEP3S(0x140)
#undef GDMIX_RSTEP3S
}

// three sums over the G lanes of an entity in one pass
template <int G>
__device__ __forceinline__ void grp_sum3(double& a, double& b, double& c, XWave& X) {
  row_sum3(a, b, c);
  if (G >= 32 && (GDMIX_QUAD_SWZ & 1)) {
    const double ta = swz_xor<16>(a), tb = swz_xor<16>(b), tc = swz_xor<16>(c);
    a += ta;
    b += tb;
    c += tc;
  } else if (G >= 32) {
    double a0, a1, b0, b1, c0, c1;
    rowpair_split(a, a0, a1);
    rowpair_split(b, b0, b1);
    rowpair_split(c, c0, c1);
    a = a0 + a1;
    b = b0 + b1;
    c = c0 + c1;
  }
  if (G >= 64) {
    double a0, a1, b0, b1, c0, c1;
    half_split(a, a0, a1);
    half_split(b, b0, b1);
    half_split(c, c0, c1);
    a = a0 + a1;
    b = b0 + b1;
    c = c0 + c1;
  }
  if (G > 64) {
    constexpr int NW = G / WAVE;
    double* pa = X.buf + (X.phase * 3 + 0) * NW;
    double* pb = X.buf + (X.phase * 3 + 1) * NW;
    double* pc = X.buf + (X.phase * 3 + 2) * NW;
    if ((threadIdx.x & (WAVE - 1)) == 0) { pa[threadIdx.x >> 6] = a; pb[threadIdx.x >> 6] = b; pc[threadIdx.x >> 6] = c; }
    __syncthreads();
    a = pa[0]; b = pb[0]; c = pc[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) { a += pa[w]; b += pb[w]; c += pc[w]; }
    X.phase ^= 1;
  }
}

__device__ __forceinline__ void row_sum2_max(double& a, double& b, double& c) {
#if GDMIX_QUAD_SWZ & 1
#define GDMIX_SSTEP3M(X)               \
  {                                    \
    const double ta = swz_xor<X>(a);   \
    const double tb = swz_xor<X>(b);   \
    const double tc = swz_xor<X>(c);   \
    a += ta;                           \
    b += tb;                           \
    c = max_nn(c, tc);                 \
  }
  GDMIX_SSTEP3M(1) GDMIX_SSTEP3M(2) GDMIX_SSTEP3M(7) GDMIX_SSTEP3M(15)
#undef GDMIX_SSTEP3M
  return;
#endif
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

template <int G>
__device__ __forceinline__ void grp_sum2_max(double& a, double& b, double& c, XWave& X) {
  row_sum2_max(a, b, c);
  if (G >= 32 && (GDMIX_QUAD_SWZ & 1)) {
    const double ta = swz_xor<16>(a), tb = swz_xor<16>(b), tc = swz_xor<16>(c);
    a += ta;
    b += tb;
    c = max_nn(c, tc);
  } else if (G >= 32) {
    double a0, a1, b0, b1, c0, c1;
    rowpair_split(a, a0, a1);
    rowpair_split(b, b0, b1);
    rowpair_split(c, c0, c1);
    a = a0 + a1;
    b = b0 + b1;
    c = max_nn(c0, c1);
  }
  if (G >= 64) {
    double a0, a1, b0, b1, c0, c1;
    half_split(a, a0, a1);
    half_split(b, b0, b1);
    half_split(c, c0, c1);
    a = a0 + a1;
    b = b0 + b1;
    c = max_nn(c0, c1);
  }
  if (G > 64) {
    constexpr int NW = G / WAVE;
    double* pa = X.buf + (X.phase * 3 + 0) * NW;
    double* pb = X.buf + (X.phase * 3 + 1) * NW;
    double* pc = X.buf + (X.phase * 3 + 2) * NW;
    if ((threadIdx.x & (WAVE - 1)) == 0) { pa[threadIdx.x >> 6] = a; pb[threadIdx.x >> 6] = b; pc[threadIdx.x >> 6] = c; }
    __syncthreads();
    a = pa[0]; b = pb[0]; c = pc[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) { a += pa[w]; b += pb[w]; c = max_nn(c, pc[w]); }
    X.phase ^= 1;
  }
}

// ordering point between LDS writes and reads of different lanes of one entity
template <int G>
__device__ __forceinline__ void grp_fence() {
  if (G > 64) __syncthreads();
  else wave_lds_fence();
}

// ---- capacity-based LDS layout of one row (entity) ---------------------------------------------------
// Offsets depend only on (PCAP = 16*EPL, NCAP, ZCAP), i.e. they are wave-uniform.
struct QuadLayout {
  int xs, xo, go, rs, csr, csc, row_ptr, col_ptr, y, o, w, bytes;
};

// (The history split by age of round 3 — the oldest pairs of an EPL = 4 kernel in LDS instead of spilled VGPRs — lost 4 - 38 % on
// <16,4> and was deleted in round 4: profiles/r03_c2_history_by_age.txt.)
constexpr int QUAD_HDR_BYTES = 8 * (2 * M_REG + 16 + 8);   // rho, alpha, LineSearch slot, 8 scalars (one per wavefront of the group)

// GDMIX_QUAD_ROW_DW: where the next row of a wavefront starts, in dwords modulo the 64 banks (rows = entities per wavefront > 1
// only; < 0 = the packed size). A 32-lane ds_read_b64 group holds two rows of a G = 16 wavefront: lane-aligned reads (x_old, g_old,
// the published point) of the two rows meet in the banks unless the rows start 32 banks apart — but the CSR pairs of rows of four
// non-zeros (32 bytes a lane) then meet four deep. Sweep and counters: profiles/r06_c2_ab.txt.
#ifndef GDMIX_QUAD_ROW_DW
#define GDMIX_QUAD_ROW_DW (-1)
#endif
constexpr int LDS_BYTES_PER_CU = 160 * 1024;

__host__ __device__ inline QuadLayout quad_layout(int pcap, int ncap, int zcap, int waves = 1, int rows = 1) {
  QuadLayout q;
  // groups wider than a wavefront keep one private copy of the uniform solver state per wavefront (all
  // copies hold the same values; sharing one would race between a fast wave's write and a slow wave's read)
  // plus the cross-wave reduction scratch
  int off = QUAD_HDR_BYTES * waves + (waves > 1 ? 8 * 2 * 3 * waves : 0);
  q.xs = off; off += 8 * pcap;
  q.xo = off; off += 8 * pcap;
  q.go = off; off += 8 * pcap;
  q.rs = off; off += 8 * ncap;
  q.csr = off; off += 8 * zcap;
  q.csc = off; off += 8 * zcap;
  q.row_ptr = off; off += 4 * (ncap + 1);
  q.col_ptr = off; off += 4 * (pcap + 1);
  q.y = off; off += 4 * ncap;
  q.o = off; off += 4 * ncap;
  q.w = off; off += 4 * ncap;
  q.bytes = (off + 15) & ~15;
  if (GDMIX_QUAD_ROW_DW >= 0 && rows > 1) {
    const int padded = q.bytes + 4 * ((GDMIX_QUAD_ROW_DW - q.bytes / 4) & 63);
    // never at the price of a resident wavefront
    if (LDS_BYTES_PER_CU / (rows * padded) >= LDS_BYTES_PER_CU / (rows * q.bytes) || LDS_BYTES_PER_CU / (rows * padded) >= 8) q.bytes = padded;
  }
  return q;
}

// Pointers into one row's LDS block (derived from one per-lane base + uniform offsets).
struct QuadLds {
  unsigned char* base;   // the entity's LDS block
  unsigned char* hdr;    // this wavefront's private header inside it
  QuadLayout q;
  bool has_w;
  __device__ __forceinline__ double* rho() const { return reinterpret_cast<double*>(hdr); }
  __device__ __forceinline__ double* alpha() const { return reinterpret_cast<double*>(hdr) + M_REG; }
  __device__ __forceinline__ LineSearch* ls() const { return reinterpret_cast<LineSearch*>(hdr + 16 * M_REG); }
  __device__ __forceinline__ double* scal() const { return reinterpret_cast<double*>(hdr + 16 * M_REG + 128); }
  __device__ __forceinline__ double* xs() const { return reinterpret_cast<double*>(base + q.xs); }
  __device__ __forceinline__ double* xo() const { return reinterpret_cast<double*>(base + q.xo); }
  __device__ __forceinline__ double* go() const { return reinterpret_cast<double*>(base + q.go); }
  __device__ __forceinline__ double* rs() const { return reinterpret_cast<double*>(base + q.rs); }
  __device__ __forceinline__ int2* csr() const { return reinterpret_cast<int2*>(base + q.csr); }
  __device__ __forceinline__ int2* csc() const { return reinterpret_cast<int2*>(base + q.csc); }
  __device__ __forceinline__ int32_t* row_ptr() const { return reinterpret_cast<int32_t*>(base + q.row_ptr); }
  __device__ __forceinline__ int32_t* col_ptr() const { return reinterpret_cast<int32_t*>(base + q.col_ptr); }
  __device__ __forceinline__ float* y() const { return reinterpret_cast<float*>(base + q.y); }
  __device__ __forceinline__ float* o() const { return reinterpret_cast<float*>(base + q.o); }
  __device__ __forceinline__ float* w() const { return reinterpret_cast<float*>(base + q.w); }
};
static_assert(sizeof(LineSearch) <= 128, "LineSearch must fit its LDS slot");

enum { SC_FOLD = 0, SC_GDOLD = 1, SC_THETA = 2, SC_MOVED = 3 };

// ---- per-entity uniform state in the LANES of a register pair (GDMIX_QUAD_LANE_STATE, round 6) ----------------------------------
// rho[a], alpha[a] and the three scalars an iteration carries (f_old, g'd_old, theta) are uniform per entity. Until round 6 they
// lived in the row's LDS header: one ds_read per use inside the dependent chain of the two-loop recursion, 9 + 9 LDS accesses to
// shift rho at a push. Here lane a of every 16-lane row keeps rho[a] (alpha[a]) of the row's entity in ONE register pair; a use is
// a v_mov_b64_dpp row_newbcast:a (gfx90a+: the one DPP control 64-bit moves take), a store two v_cndmask under a constant lane
// mask, the shift one row_shl:1. Groups wider than a row keep identical copies in every row (all their reductions are bit-equal
// in all lanes). Lanes: 0..9 rho | alpha, 10 f_old, 11 g'd_old, 12 theta (of the first pair only).
// Measured (profiles/r06_c2_ab.txt): LDS instructions of <32,3> - 36 %, vector instructions + 4.8 %, the kernel alone 4.50 -> 4.55 ms,
// <16,4> (four more registers to spill) 2.41 -> 2.61 ms, the C2 step 9.04 -> 9.22 ms: these kernels are bound by vector issue, not by
// LDS traffic. Off; kept for the record and for the next architecture.
#ifndef GDMIX_QUAD_LANE_STATE
#define GDMIX_QUAD_LANE_STATE 0
#endif
#ifndef GDMIX_QUAD_ZERO_STEPS
#define GDMIX_QUAD_ZERO_STEPS 0      // (described where the steps are: quad_solve)
#endif
enum { LN_FOLD = 10, LN_GDOLD = 11, LN_THETA = 12 };
static_assert(M_REG <= 10, "lane slots 10..12 hold the scalars");

template <int LANE>
__device__ __forceinline__ double row_get(double v) {
  // mov_dpp, not update_dpp: no `old` operand the compiler would have to materialise in the destination first
  return __longlong_as_double(__builtin_amdgcn_mov_dpp(__double_as_longlong(v), 0x150 + LANE, 0xf, 0xf, true));
}
template <int LANE>
__device__ __forceinline__ double row_put(double reg, double v) {
  return __builtin_amdgcn_inverse_ballot_w64(0x0001000100010001ull << LANE) ? v : reg;
}

// Is `v` true in any lane of the entity's group? G <= 64: from the wavefront's ballot (all lanes of a group are active
// together). Wider groups: this wavefront's part only (quad_eval collects the wavefronts' parts behind its first fence).
template <int G>
__device__ __forceinline__ bool grp_any(bool v) {
  const unsigned long long m = __ballot(v);
  if (G >= WAVE) return m != 0ull;
  const int sh = (int)threadIdx.x & (WAVE - 1) & ~(G - 1);
  return ((m >> sh) & ((1ull << G) - 1ull)) != 0ull;
}

#ifndef QUAD_FUSED_COLS
#define QUAD_FUSED_COLS 1
#endif
#ifndef QUAD_FUSED_COLS4
#define QUAD_FUSED_COLS4 1
#endif
// sum_k val[k] * vec[idx[k]] over `len` packed {idx, float bits} pairs starting at `pairs`, added to acc
// in index order. Gathers are issued four at a time; the FMA chain keeps the sequential order (masked
// tail entries multiply by an exact 0).
__device__ __forceinline__ double gather_dot(const int2* pairs, int len, const double* vec, double acc) {
  for (int c = 0; c < len; c += 4) {
    const int r = len - c;   // >= 1
    const int2 p0 = pairs[c];
    const int2 p1 = pairs[c + (r > 1 ? 1 : 0)];
    const int2 p2 = pairs[c + (r > 2 ? 2 : 0)];
    const int2 p3 = pairs[c + (r > 3 ? 3 : 0)];
    const double v0 = vec[p0.x], v1 = vec[p1.x], v2 = vec[p2.x], v3 = vec[p3.x];
    acc += (double)__int_as_float(p0.y) * v0;
    acc += (r > 1 ? (double)__int_as_float(p1.y) : 0.0) * v1;
    acc += (r > 2 ? (double)__int_as_float(p2.y) : 0.0) * v2;
    acc += (r > 3 ? (double)__int_as_float(p3.y) : 0.0) * v3;
  }
  return acc;
}

// The same for short runs (the columns of an entity mostly hold one or two entries): two at a time.
__device__ __forceinline__ double gather_dot2(const int2* pairs, int len, const double* vec, double acc) {
  for (int c = 0; c < len; c += 2) {
    const int r = len - c;   // >= 1
    const int2 p0 = pairs[c];
    const int2 p1 = pairs[c + (r > 1 ? 1 : 0)];
    const double v0 = vec[p0.x], v1 = vec[p1.x];
    acc += (double)__int_as_float(p0.y) * v0;
    acc += (r > 1 ? (double)__int_as_float(p1.y) : 0.0) * v1;
  }
  return acc;
}

// The EPL columns of a lane in ONE loop, two entries of each per trip: every trip issues all its pair loads, then all its
// gathers (2 EPL in flight) — EPL loops one after the other were 2 EPL dependent LDS round trips per trip of each and a loop
// branch per column, with two wavefronts per SIMD to hide them. A column's products are added in its own order as before;
// a column that has ended adds exact zeros (gathered from the residual's slot 0, always finite).
template <int EPL>
__device__ __forceinline__ void gather_dot2_cols(const int2* csc, const unsigned (&colc)[EPL], const double* vec, double (&acc)[EPL]) {
  int len[EPL], mx = 0;
  const int2* pairs[EPL];
#pragma unroll
  for (int s = 0; s < EPL; ++s) {
    len[s] = (int)(colc[s] >> 16);
    pairs[s] = csc + (colc[s] & 0xffffu);
    mx = max(mx, len[s]);
  }
  for (int c = 0; c < mx; c += 2) {
    int2 p0[EPL], p1[EPL];
#pragma unroll
    for (int s = 0; s < EPL; ++s) {
      const int r = len[s] - c;
      p0[s] = pairs[s][r > 0 ? c : 0];
      p1[s] = pairs[s][r > 1 ? c + 1 : 0];
      if (r <= 0) p0[s] = make_int2(0, 0);
      if (r <= 1) p1[s] = make_int2(0, 0);
    }
    double v0[EPL], v1[EPL];
#pragma unroll
    for (int s = 0; s < EPL; ++s) { v0[s] = vec[p0[s].x]; v1[s] = vec[p1[s].x]; }
#pragma unroll
    for (int s = 0; s < EPL; ++s) {
      acc[s] += (double)__int_as_float(p0[s].y) * v0[s];
      acc[s] += (double)__int_as_float(p1[s].y) * v1[s];
    }
  }
}

// The same with ONE entry of each column per trip (EPL = 4: two per trip spill).
template <int EPL>
__device__ __forceinline__ void gather_dot1_cols(const int2* csc, const unsigned (&colc)[EPL], const double* vec, double (&acc)[EPL]) {
  int len[EPL], mx = 0;
  const int2* pairs[EPL];
#pragma unroll
  for (int s = 0; s < EPL; ++s) {
    len[s] = (int)(colc[s] >> 16);
    pairs[s] = csc + (colc[s] & 0xffffu);
    mx = max(mx, len[s]);
  }
  for (int c = 0; c < mx; ++c) {
    int2 p0[EPL];
#pragma unroll
    for (int s = 0; s < EPL; ++s) {
      p0[s] = pairs[s][c < len[s] ? c : 0];
      if (c >= len[s]) p0[s] = make_int2(0, 0);
    }
    double v0[EPL];
#pragma unroll
    for (int s = 0; s < EPL; ++s) v0[s] = vec[p0[s].x];
#pragma unroll
    for (int s = 0; s < EPL; ++s) acc[s] += (double)__int_as_float(p0[s].y) * v0[s];
  }
}

// f and g at xt. rowc: packed (start | len << 16) of the lane's first sample; colc[s]: same for the
// lane's coefficient slots (len = 0 for the intercept / unused slots).
template <int G, int EPL, bool LONG_COLS = true>
__device__ __forceinline__ double quad_eval(const QuadLds& L, const SolveParams& o, int gl, int n, int p, int ic,
                                            unsigned rowc, const unsigned (&colc)[EPL], const double (&xt)[EPL],
                                            double (&g)[EPL], XWave& X, bool first, bool& counted) {
  double* const xs = L.xs();
  double* const rs = L.rs();
  // xs still holds the point of the previous evaluation: scipy's ScalarFunction serves an equal point from its cache without
  // counting it (nfev is its funcalls; a step too small to move x, _differentiable_functions.py)
  bool mv = first;
#pragma unroll
  for (int s = 0; s < EPL; ++s) {
    const int j = gl + G * s;
    if (j < p) {
      mv = mv || (xs[j] != xt[s]);
      xs[j] = xt[s];
    }
  }
  counted = grp_any<G>(mv);
  if (G > WAVE) L.scal()[SC_MOVED] = counted ? 1.0 : 0.0;   // this wavefront's part; the others' behind the fence
  grp_fence<G>();
  if (G > WAVE) {
    constexpr int NWG = G / WAVE;
    bool any = false;
#pragma unroll
    for (int w = 0; w < NWG; ++w) any = any || reinterpret_cast<const double*>(L.base + w * QUAD_HDR_BYTES + 16 * M_REG + 128)[SC_MOVED] != 0.0;
    counted = any;
  }
  double part = 0.0, rpart = 0.0;
  const double x0 = ic ? xs[0] : 0.0;
  if (gl < n) {
    const double z = gather_dot(L.csr() + (rowc & 0xffffu), (int)(rowc >> 16), xs + ic, x0) + (double)L.o()[gl];
    double ri;
    part = logistic_terms(z, (double)L.y()[gl], L.has_w ? (double)L.w()[gl] : 1.0, ri);
    rs[gl] = ri;
    rpart = ri;
  }
  for (int i = gl + G; i < n; i += G) {   // entities with more samples than lanes
    const int k0 = L.row_ptr()[i], k1 = L.row_ptr()[i + 1];
    const double z = gather_dot(L.csr() + k0, k1 - k0, xs + ic, x0) + (double)L.o()[i];
    double ri;
    part += logistic_terms(z, (double)L.y()[i], L.has_w ? (double)L.w()[i] : 1.0, ri);
    rs[i] = ri;
    rpart += ri;
  }
  const int first_reg = (ic && !o.regularize_bias) ? 1 : 0;
  double sq = 0.0;
#pragma unroll
  for (int s = 0; s < EPL; ++s) {
    const int j = gl + G * s;
    if (j >= first_reg && j < p) sq += xt[s] * xt[s];
  }
  // cost.sum() and the regulariser are summed separately and added once, as the reference writes it
  // (binary_logistic_regression.py:105-108): spreading (l2/2) x_j^2 over the lanes' loss partials rounds differently, which
  // decides line searches on entities whose loss is ~1e13 (tests/golden exit_extreme_02)
  grp_sum3<G>(part, rpart, sq, X);
  part += 0.5 * o.l2 * sq;
  grp_fence<G>();
  const double inv_n = 1.0 / (double)n;
  if ((EPL <= 3 && QUAD_FUSED_COLS) || (EPL == 4 && QUAD_FUSED_COLS4 && LONG_COLS)) {
    double acc[EPL];
    unsigned cc[EPL];
#pragma unroll
    for (int s = 0; s < EPL; ++s) {
      const int j = gl + G * s;
      acc[s] = (ic && j == 0) ? rpart : 0.0;
      cc[s] = (j < p) ? colc[s] : 0u;
    }
    if (EPL <= 3) gather_dot2_cols<EPL>(L.csc(), cc, rs, acc);
    else gather_dot1_cols<EPL>(L.csc(), cc, rs, acc);
#pragma unroll
    for (int s = 0; s < EPL; ++s) {
      const int j = gl + G * s;
      const double reg = (j < first_reg) ? 0.0 : o.l2 * xt[s];
      g[s] = (j < p) ? inv_n * (acc[s] + reg) : 0.0;
    }
    return inv_n * part;
  }
#pragma unroll
  for (int s = 0; s < EPL; ++s) {
    const int j = gl + G * s;
    double gj = 0.0;
    if (j < p) {
      double acc = (ic && j == 0) ? rpart : 0.0;
      // two entries at a time where registers allow (A/B on C2: -3 % for EPL = 3; EPL = 4 spills more with it and loses 6 %)
      acc = (EPL <= 3) ? gather_dot2(L.csc() + (colc[s] & 0xffffu), (int)(colc[s] >> 16), rs, acc)
                       : gather_dot(L.csc() + (colc[s] & 0xffffu), (int)(colc[s] >> 16), rs, acc);
      const double reg = (j < first_reg) ? 0.0 : o.l2 * xt[s];
      gj = inv_n * (acc + reg);
    }
    g[s] = gj;
  }
  return inv_n * part;
}

// ---- the history without its shift (GDMIX_QUAD_APPEND, round 6) ------------------------------------------------------------------
// A push shifts the whole register history down by one (54 v_mov_b64 at EPL = 3, + 9 + 9 LDS accesses for rho): the pair-push item of
// VERDICT r5. Round 5 tried "pair a in slot a" with the slot chosen per ROW and lost to the ten compare / exec-save / branch skips
// that needs. Here the slot is chosen per WAVEFRONT: while slots are free, every trip on which any row of the wavefront stores a pair
// takes the next slot w for all rows (one scalar jump into ten static bodies); a row that stores nothing on that trip simply does not
// use slot w (its mask `um` lacks the bit: a hole). Slot order is still time order for every row, so both loops of the recursion run
// over the slots any row uses, in the same static order as before, with a row's `use` read from its mask. Once all ten slots are taken
// a row frees, when it stores a pair, its oldest slot that holds nothing it uses (a hole, or a pair beyond its last m) — else its
// oldest pair — by shifting the slots above it down (v_cndmask instead of v_mov: per-row start). Same pairs in the same order: same
// bits. A/B: profiles/r06_c2_ab.txt.
#ifndef GDMIX_QUAD_APPEND
#define GDMIX_QUAD_APPEND 0
#endif
static_assert(!(GDMIX_QUAD_APPEND && (GDMIX_QUAD_LANE_STATE || GDMIX_QUAD_ZERO_STEPS)), "the append form of the history keeps rho in LDS and the masked steps");

// OR of a per-entity value over the wavefront's entities (every lane of a group holds the group's value)
template <int G>
__device__ __forceinline__ unsigned wave_or_groups(unsigned v) {
  unsigned r = (unsigned)__builtin_amdgcn_readlane((int)v, 0);
  if (G < 64) r |= (unsigned)__builtin_amdgcn_readlane((int)v, 32);
  if (G < 32) r |= (unsigned)__builtin_amdgcn_readlane((int)v, 16) | (unsigned)__builtin_amdgcn_readlane((int)v, 48);
  return r;
}

template <int EPL>
struct QuadState {
  double x[EPL], g[EPL], d[EPL];
};

// The solve of the (up to) four entities of a wave. `valid` marks rows that own an entity.
// LONG_COLS: the class holds entities with more samples than the group has lanes (columns of several entries are the rule): the
// EPL = 4 kernels then gather their four columns in one loop too, an entry of each per trip (C5-shaped classes - 1 to - 3 %); where
// columns mostly hold one entry (C2's <16,4>: n <= 16) the four short loops are as fast and spill less (+ 0.5 % with the fused loop)
template <int G, int EPL, bool LONG_COLS = true>
__device__ __forceinline__ void quad_solve(const QuadLds& L, const SolveParams& o, int gl, int n, int p, int ic,
                                           bool valid, unsigned rowc, const unsigned (&colc)[EPL], QuadState<EPL>& V,
                                           XWave& X, SolveStats& out) {
  // pair a of M_REG, newest last: a register shift register (every index a compile-time constant)
  constexpr int KR = M_REG;
  double S[KR][EPL], Y[KR][EPL];
#pragma unroll
  for (int a = 0; a < KR; ++a) {
#pragma unroll
    for (int s = 0; s < EPL; ++s) { S[a][s] = 0.0; Y[a][s] = 0.0; }
  }
  double* const rho = L.rho();      // per-entity uniform state: every lane of the row stores the same value
  double* const alpha = L.alpha();
  double* const scal = L.scal();
  double* const xo = L.xo();
  double* const go = L.go();
  const int m = o.m;
  int cnt = 0;
#if GDMIX_QUAD_APPEND
  int w = 0;            // slots taken so far, the same for every row of the wavefront (M_REG: full)
  unsigned um = 0u;     // this row's pairs in use: bit a = slot a
#endif
  int nit = 0, nfev = 0, ifun = 0;
  int status = valid ? -1 : 0;
  bool iter0 = true, first = true;
  // nfev is scipy's funcalls: an evaluation at the point of the previous evaluation is not counted (quad_eval)
  double f = 0.0, gd = 0.0, rr = 0.0, stp = 0.0, sbgnrm = 0.0;
#if GDMIX_QUAD_LANE_STATE
  double rho_v = row_put<LN_THETA>(0.0, 1.0), alpha_v = 0.0;
#endif
  if (valid) {
#if !GDMIX_QUAD_LANE_STATE
    scal[SC_FOLD] = 0.0; scal[SC_GDOLD] = 0.0; scal[SC_THETA] = 1.0;
#endif
#pragma unroll
    for (int s = 0; s < EPL; ++s) {
      const int j = gl + G * s;
      if (j < p) { xo[j] = 0.0; go[j] = 0.0; }
    }
  }
  while (__any(status < 0)) {
    bool need_dir = false, restart = false;
#if GDMIX_QUAD_APPEND
    bool do_push = false;
#endif
    if (status < 0) {
      // ---- f, g at the trial point; g'd, y'y and max|g| in one reduction pass ------------------------
      bool counted;
      f = quad_eval<G, EPL, LONG_COLS>(L, o, gl, n, p, ic, rowc, colc, V.x, V.g, X, first, counted);
      nfev += counted ? 1 : 0;
      {
        double a = 0.0, b = 0.0, c = 0.0;
#pragma unroll
        for (int s = 0; s < EPL; ++s) {
          const int j = gl + G * s;
          const double gold = (j < p) ? go[j] : 0.0;
          a += V.g[s] * V.d[s];
          const double yj = V.g[s] - gold;
          b += yj * yj;
          c = max_nn(c, fabs(V.g[s]));
        }
        grp_sum2_max<G>(a, b, c, X);
        gd = a; rr = b; sbgnrm = c;
      }
      if (first) {
        first = false;
        if (sbgnrm <= o.pgtol) status = 0;
        else need_dir = true;
      } else {
        // the first trial is accepted 95 % of the time: three numbers of the search's state decide that (dcsrch_converged); the rest
        // of the state is read — and, when the search goes on, written back — only when a row of the wave needs it (an accepted or
        // abandoned search is started afresh by dcsrch_start: nothing reads its state again). LDS traffic is what a trip of these
        // kernels waits for: the 15 stores alone were 2 % of the C2 step.
        int task = LS_CONV;
        {
#if GDMIX_QUAD_LANE_STATE
          // finit, ginit of the search are f_old, g'd_old of the iteration; gtest = ftol * ginit (dcsrch_start): same bits
          const double ginit = row_get<LN_GDOLD>(rho_v);
          const bool conv = dcsrch_converged(row_get<LN_FOLD>(rho_v), LS_FTOL * ginit, ginit, f, gd, stp);
#else
          const LineSearch* const lsp = L.ls();
          const bool conv = dcsrch_converged(lsp->finit, lsp->gtest, lsp->ginit, f, gd, stp);
#endif
          if (__any(!conv)) {
            if (!conv) {
              LineSearch LS = *L.ls();
              task = dcsrch_step(LS, f, gd, stp);
              if (task == LS_FG) *L.ls() = LS;
            }
          }
        }
        if (task == LS_FG) {
          ++ifun;
          if (ifun - 1 < o.maxls) {
#pragma unroll
            for (int s = 0; s < EPL; ++s) {
              const int j = gl + G * s;
              if (j < p) V.x[s] = stp * V.d[s] + xo[j];   // stp == 1: exactly xo + d
            }
          } else {
            restart = true;   // iback >= maxls
            need_dir = true;
          }
        } else {
          // ---- NEW_X: scipy's python loop first (nit / maxiter / maxfun), then mainlb's own tests ---
          ++nit;
          iter0 = false;
#if GDMIX_QUAD_LANE_STATE
          const double fold = row_get<LN_FOLD>(rho_v);
#else
          const double fold = scal[SC_FOLD];
#endif
          const double dmx = fmax(fabs(fold), fmax(fabs(f), 1.0));
          if (nit >= o.max_iter) status = 2;
          else if (nfev > o.maxfun) status = 3;
          else if (sbgnrm <= o.pgtol) status = 0;
          else if (fold - f <= o.ftol * dmx) status = 1;
          else {
            need_dir = true;
#if GDMIX_QUAD_LANE_STATE
            const double gdold = row_get<LN_GDOLD>(rho_v);
#else
            const double gdold = scal[SC_GDOLD];
#endif
            double dr, ddum;
            if (stp == 1.0) { dr = gd - gdold; ddum = -gdold; }
            else { dr = (gd - gdold) * stp; ddum = -gdold * stp; }
#if GDMIX_QUAD_APPEND
            do_push = dr > EPSMCH * ddum;      // stored behind this region, where the slot is uniform (dr is formed again there: no register carries it)
#else
            if (dr > EPSMCH * ddum) {
              // push (s, y): the history shifts down by one, newest at KR-1
#if GDMIX_QUAD_LANE_STATE
              {
                const double sh = dpp_row<0x101>(rho_v);   // row_shl:1: lane a <- lane a + 1
                rho_v = __builtin_amdgcn_inverse_ballot_w64(0x01ff01ff01ff01ffull) ? sh : rho_v;
              }
#else
#pragma unroll
              for (int a = 0; a < M_REG - 1; ++a) rho[a] = rho[a + 1];
#endif
#pragma unroll
              for (int a = 0; a < KR - 1; ++a) {
#pragma unroll
                for (int s = 0; s < EPL; ++s) { S[a][s] = S[a + 1][s]; Y[a][s] = Y[a + 1][s]; }
              }
#pragma unroll
              for (int s = 0; s < EPL; ++s) {
                const int j = gl + G * s;
                S[KR - 1][s] = stp * V.d[s];   // exact for stp == 1
                Y[KR - 1][s] = V.g[s] - ((j < p) ? go[j] : 0.0);
              }
#if GDMIX_QUAD_LANE_STATE
              rho_v = row_put<M_REG - 1>(rho_v, 1.0 / dr);
              rho_v = row_put<LN_THETA>(rho_v, rr / dr);
#else
              rho[M_REG - 1] = 1.0 / dr;
              scal[SC_THETA] = rr / dr;
#endif
              if (cnt < m) ++cnt;
            }
#endif
          }
        }
      }
    }
#if GDMIX_QUAD_APPEND
    // ---- the pairs of this trip: one slot for the whole wavefront while slots are free, then a shift per row ----------------------
    if (__any(do_push)) {
      double p_rho = 0.0;
      if (do_push) {
        const double gdold = scal[SC_GDOLD];
        const double dr = (stp == 1.0) ? gd - gdold : (gd - gdold) * stp;      // as in the region above: same bits
        p_rho = 1.0 / dr;
        scal[SC_THETA] = rr / dr;
      }
      if (w < M_REG) {
#define QUAD_APPEND_AT(a)                                                             \
        case a:                                                                        \
          if (do_push) {                                                               \
            _Pragma("unroll") for (int s = 0; s < EPL; ++s) {                          \
              const int j = gl + G * s;                                                \
              S[a][s] = stp * V.d[s];                                                  \
              Y[a][s] = V.g[s] - ((j < p) ? go[j] : 0.0);                              \
            }                                                                          \
          }                                                                            \
          break;
        switch (w) {
          QUAD_APPEND_AT(0) QUAD_APPEND_AT(1) QUAD_APPEND_AT(2) QUAD_APPEND_AT(3) QUAD_APPEND_AT(4)
          QUAD_APPEND_AT(5) QUAD_APPEND_AT(6) QUAD_APPEND_AT(7) QUAD_APPEND_AT(8) QUAD_APPEND_AT(9)
          default: break;
        }
#undef QUAD_APPEND_AT
        if (do_push) { rho[w] = p_rho; um |= 1u << w; }
        ++w;
      } else if (do_push) {
        const unsigned freeb = ~um & ((1u << M_REG) - 1u);
        const int k = freeb ? (__ffs((int)freeb) - 1) : 0;      // the slot this row gives up: everything above it moves down by one
#pragma unroll
        for (int a = 0; a < KR - 1; ++a) {
          const bool mv = a >= k;
#pragma unroll
          for (int s = 0; s < EPL; ++s) { S[a][s] = mv ? S[a + 1][s] : S[a][s]; Y[a][s] = mv ? Y[a + 1][s] : Y[a][s]; }
        }
#pragma unroll
        for (int s = 0; s < EPL; ++s) {
          const int j = gl + G * s;
          S[KR - 1][s] = stp * V.d[s];
          Y[KR - 1][s] = V.g[s] - ((j < p) ? go[j] : 0.0);
        }
        for (int a = k; a < M_REG - 1; ++a) rho[a] = rho[a + 1];
        rho[M_REG - 1] = p_rho;
        um = (um & ((1u << k) - 1u)) | ((um >> (k + 1)) << k) | (1u << (M_REG - 1));
      }
      if (do_push) {
        if (__popc(um) > m) um &= um - 1u;      // more than the last m pairs: the oldest is not used any more
        cnt = __popc(um);
      }
    }
#endif
    // ---- new search direction for the rows that need one (again after a line-search restart) ---------
    while (__any(need_dir)) {
      if (need_dir && restart) {
#pragma unroll
        for (int s = 0; s < EPL; ++s) {
          const int j = gl + G * s;
          if (j < p) { V.x[s] = xo[j]; V.g[s] = go[j]; }
        }
#if GDMIX_QUAD_LANE_STATE
        f = row_get<LN_FOLD>(rho_v);
#else
        f = scal[SC_FOLD];
#endif
        restart = false;
        if (cnt == 0) { status = 4; need_dir = false; }
        else {
          cnt = 0;
#if GDMIX_QUAD_APPEND
          um = 0u;
#endif
#if GDMIX_QUAD_LANE_STATE
          rho_v = row_put<LN_THETA>(rho_v, 1.0);
#else
          scal[SC_THETA] = 1.0;
#endif
        }
      }
      if (need_dir) {
#pragma unroll
        for (int s = 0; s < EPL; ++s) V.d[s] = -V.g[s];
      }
      // two-loop recursion over each row's last cnt pairs (indices M_REG-cnt .. M_REG-1, newest last). The pairs any row of the wave
      // uses are M_REG-cmax .. M_REG-1 with cmax uniform: the first loop leaves at its lower end, the second enters at it (a switch
      // that falls through) — until round 5 every one of the 2 M_REG steps was skipped on its own (`if (__any(use))`: a compare, an
      // exec save and a taken branch each, ~11 of 20 skipped at C2's mean history of 4.5 pairs).
#if GDMIX_QUAD_APPEND
      // the slots any row of the wavefront uses in this direction: [lo, hi) (uniform); a row's own from its mask
      const unsigned um_all = wave_or_groups<G>(need_dir ? um : 0u);
      const int lo = um_all ? (__ffs((int)um_all) - 1) : M_REG, hi = um_all ? (32 - __clz((int)um_all)) : 0;
#define QUAD_USE(a) (need_dir && ((um >> (a)) & 1u))
#define QUAD_FIRST_EXIT(a) if ((a) < lo) goto first_loop_done;
#define QUAD_SECOND_EXIT(a) if ((a) >= hi) goto second_loop_done;
#else
      const int cmax = wave_max_nonneg_i32(need_dir ? cnt : 0);
      const int a0 = M_REG - cmax;
#define QUAD_USE(a) (need_dir && ((a) >= M_REG - cnt))
#define QUAD_FIRST_EXIT(a) if ((a) < a0) goto first_loop_done;
#define QUAD_SECOND_EXIT(a)
#endif
#if GDMIX_QUAD_LANE_STATE
#define QUAD_RHO(a) row_get<a>(rho_v)
#define QUAD_ALPHA(a) row_get<a>(alpha_v)
#define QUAD_SET_ALPHA(a, v) alpha_v = row_put<a>(alpha_v, v)
#else
#define QUAD_RHO(a) rho[a]
#define QUAD_ALPHA(a) alpha[a]
#define QUAD_SET_ALPHA(a, v) alpha[a] = (v)
#endif
      // GDMIX_QUAD_ZERO_STEPS (round 6): a row that does not use pair a takes the step with a zero multiplier instead of sitting it
      // out under an exec mask — d - 0 * y is d (the stored pairs are finite), and the wave saves the exec save / branch / restore
      // and the copies of d the compiler placed behind every masked region (three v_mov_b64 a step at EPL = 3).
      // Measured (profiles/r06_c2_ab.txt): <32,3> alone 4.50 -> 4.52 ms, <16,4> 2.41 -> 2.77 ms (the compiler keeps more of d live: 70 -> 84
      // spilled registers), step 9.04 -> 9.43 ms. Off.
#ifndef GDMIX_QUAD_ZERO_STEPS
#define GDMIX_QUAD_ZERO_STEPS 0
#endif
#if GDMIX_QUAD_ZERO_STEPS
#define QUAD_FIRST_LOOP_STEP(a)                                                   \
      {                                                                           \
        if ((a) < a0) goto first_loop_done;                                       \
        const bool use = need_dir && ((a) >= M_REG - cnt);                        \
        double t = 0.0;                                                           \
        _Pragma("unroll") for (int s = 0; s < EPL; ++s) t += S[a][s] * V.d[s];    \
        const double nal = use ? -(QUAD_RHO(a) * grp_sum<G>(t, X)) : 0.0;         \
        QUAD_SET_ALPHA(a, -nal);                                                  \
        _Pragma("unroll") for (int s = 0; s < EPL; ++s) V.d[s] = fma(nal, Y[a][s], V.d[s]); \
      }
#define QUAD_SECOND_LOOP_STEP(a)                                                  \
      {                                                                           \
        const bool use = need_dir && ((a) >= M_REG - cnt);                        \
        double t = 0.0;                                                           \
        _Pragma("unroll") for (int s = 0; s < EPL; ++s) t += Y[a][s] * V.d[s];    \
        const double c = use ? QUAD_ALPHA(a) - QUAD_RHO(a) * grp_sum<G>(t, X) : 0.0; \
        _Pragma("unroll") for (int s = 0; s < EPL; ++s) V.d[s] += c * S[a][s];    \
      }
#else
#define QUAD_FIRST_LOOP_STEP(a)                                                   \
      {                                                                           \
        QUAD_FIRST_EXIT(a)                                                        \
        const bool use = QUAD_USE(a);                                             \
        if (use) {                                                                \
          double t = 0.0;                                                         \
          _Pragma("unroll") for (int s = 0; s < EPL; ++s) t += S[a][s] * V.d[s];  \
          const double al = QUAD_RHO(a) * grp_sum<G>(t, X);                       \
          QUAD_SET_ALPHA(a, al);                                                  \
          _Pragma("unroll") for (int s = 0; s < EPL; ++s) V.d[s] -= al * Y[a][s]; \
        }                                                                         \
      }
#endif
      static_assert(M_REG == 10, "the steps below are written out for ten pairs");
#if GDMIX_QUAD_APPEND
      switch (hi) {      // newest slot in use first
        case 10: QUAD_FIRST_LOOP_STEP(9) [[fallthrough]];
        case 9: QUAD_FIRST_LOOP_STEP(8) [[fallthrough]];
        case 8: QUAD_FIRST_LOOP_STEP(7) [[fallthrough]];
        case 7: QUAD_FIRST_LOOP_STEP(6) [[fallthrough]];
        case 6: QUAD_FIRST_LOOP_STEP(5) [[fallthrough]];
        case 5: QUAD_FIRST_LOOP_STEP(4) [[fallthrough]];
        case 4: QUAD_FIRST_LOOP_STEP(3) [[fallthrough]];
        case 3: QUAD_FIRST_LOOP_STEP(2) [[fallthrough]];
        case 2: QUAD_FIRST_LOOP_STEP(1) [[fallthrough]];
        case 1: QUAD_FIRST_LOOP_STEP(0) [[fallthrough]];
        default: break;
      }
#else
      QUAD_FIRST_LOOP_STEP(9) QUAD_FIRST_LOOP_STEP(8) QUAD_FIRST_LOOP_STEP(7) QUAD_FIRST_LOOP_STEP(6) QUAD_FIRST_LOOP_STEP(5)
      QUAD_FIRST_LOOP_STEP(4) QUAD_FIRST_LOOP_STEP(3) QUAD_FIRST_LOOP_STEP(2) QUAD_FIRST_LOOP_STEP(1) QUAD_FIRST_LOOP_STEP(0)
#endif
#undef QUAD_FIRST_LOOP_STEP
    first_loop_done:;
      if (need_dir && cnt > 0) {
#if GDMIX_QUAD_LANE_STATE
        const double h0 = 1.0 / row_get<LN_THETA>(rho_v);
#else
        const double h0 = 1.0 / scal[SC_THETA];
#endif
#pragma unroll
        for (int s = 0; s < EPL; ++s) V.d[s] *= h0;
      }
#if !GDMIX_QUAD_ZERO_STEPS
#define QUAD_SECOND_LOOP_STEP(a)                                                  \
      {                                                                           \
        QUAD_SECOND_EXIT(a)                                                       \
        const bool use = QUAD_USE(a);                                             \
        if (use) {                                                                \
          double t = 0.0;                                                         \
          _Pragma("unroll") for (int s = 0; s < EPL; ++s) t += Y[a][s] * V.d[s];  \
          const double c = QUAD_ALPHA(a) - QUAD_RHO(a) * grp_sum<G>(t, X);        \
          _Pragma("unroll") for (int s = 0; s < EPL; ++s) V.d[s] += c * S[a][s];  \
        }                                                                         \
      }
#endif
#if GDMIX_QUAD_APPEND
      switch (lo) {
#else
      switch (a0) {
#endif
        case 0: QUAD_SECOND_LOOP_STEP(0) [[fallthrough]];
        case 1: QUAD_SECOND_LOOP_STEP(1) [[fallthrough]];
        case 2: QUAD_SECOND_LOOP_STEP(2) [[fallthrough]];
        case 3: QUAD_SECOND_LOOP_STEP(3) [[fallthrough]];
        case 4: QUAD_SECOND_LOOP_STEP(4) [[fallthrough]];
        case 5: QUAD_SECOND_LOOP_STEP(5) [[fallthrough]];
        case 6: QUAD_SECOND_LOOP_STEP(6) [[fallthrough]];
        case 7: QUAD_SECOND_LOOP_STEP(7) [[fallthrough]];
        case 8: QUAD_SECOND_LOOP_STEP(8) [[fallthrough]];
        case 9: QUAD_SECOND_LOOP_STEP(9) [[fallthrough]];
        default: break;
      }
#if GDMIX_QUAD_APPEND
    second_loop_done:;
#endif
#undef QUAD_USE
#undef QUAD_FIRST_EXIT
#undef QUAD_SECOND_EXIT
#undef QUAD_SECOND_LOOP_STEP
#undef QUAD_RHO
#undef QUAD_ALPHA
#undef QUAD_SET_ALPHA
      if (need_dir) {
        // z = x + d ; d = z - x (mainlb re-derives d from the subspace point); save x, g
        double dd = 0.0, gdp = 0.0;
#pragma unroll
        for (int s = 0; s < EPL; ++s) {
          const int j = gl + G * s;
          const double xj = V.x[s];
          const double z = xj + V.d[s];
          const double dj = z - xj;
          V.d[s] = dj;
          if (j < p) { xo[j] = xj; go[j] = V.g[s]; }
          dd += dj * dj;
          gdp += V.g[s] * dj;
        }
        grp_sum2<G>(dd, gdp, X);
        gd = gdp;
#if GDMIX_QUAD_LANE_STATE
        rho_v = row_put<LN_GDOLD>(rho_v, gd);
        rho_v = row_put<LN_FOLD>(rho_v, f);
#else
        scal[SC_GDOLD] = gd;
        scal[SC_FOLD] = f;
#endif
        if (gd >= 0.0) {
          restart = true;   // lnsrlb info = -4: stay in this loop
        } else {
          stp = iter0 ? fmin(1.0 / sqrt(dd), LS_STPMAX) : 1.0;
          LineSearch LS;
          dcsrch_start(LS, f, gd, stp);
          *L.ls() = LS;
          ifun = 1;
#pragma unroll
          for (int s = 0; s < EPL; ++s) V.x[s] = stp * V.d[s] + V.x[s];   // V.x still holds x_old here
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
    grp_sum2_max<G>(d0, d1, mx, X);
    sbgnrm = mx;
  }
  out.f = f;
  out.gnorm = sbgnrm;
  out.nit = nit;
  out.nfev = nfev;
  out.status = status;
}

}  // namespace gdmix
