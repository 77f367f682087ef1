// fe_solve.hip — fixed-effect trainer (include/gdmix_fe.h): one worker's shard resident in HBM, one L-BFGS
// evaluation = two streaming passes over it + a handful of small kernels; the L-BFGS driver is the same resumable
// compact-form step the team kernels use (re_lbfgs_compact.hpp), here with its state in HBM between kernels so
// that the caller can all-reduce [gradient, value] across workers in between
// (fixed_effect_lr_lbfgs_model.py:309-392: _train_model_fn; :394-430 _compute_loss_and_gradients; :635-643).
//
// The two passes (fe_scatter_kernel) are written for HBM: no gather may cost an L2 request per entry, and no sum may
// depend on the launch. Each pass has its own copy of the non-zeros, grouped by block of FE_B consecutive *outputs* (rows
// for X theta, columns for X'r) and, inside a block, in the order of the *gathered* vector (columns for X theta, rows for
// X'r): one wavefront owns a block's FE_B accumulators in LDS, streams the block's entries (8 B each, every byte once:
// fp32 value + one word holding the accumulator index and the gathered element's index relative to the unit's first;
// three arrays and 10 B when a unit spans more than 2^21 gathered elements), reads the gathered vector almost
// sequentially (a 128-byte line serves all the entries that fall into it, ~20 here, instead of one) and adds the
// products into LDS with ds_add_f64. A block with many entries is cut into chunks of equal entry count, one wavefront
// each, whose partial sums a second small kernel adds in chunk order. One wavefront per accumulator set and in-order
// LDS atomics: a row's terms are added one by one in column order, a column's in row order, whatever the launch.
#include <cstdlib>
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "re_internal.hpp"
#include "re_lbfgs_compact.hpp"
#include "../../include/gdmix_fe.h"

#include <algorithm>
#include <new>
#include <vector>

namespace gdmix {

#ifndef GDMIX_FE_B
#define GDMIX_FE_B 2048
#endif
#ifndef GDMIX_FE_UNROLL
#define GDMIX_FE_UNROLL 8
#endif
#ifndef GDMIX_FE_PACK
#define GDMIX_FE_PACK 1      // 0: always the three-array form (tests of that path)
#endif
constexpr int FE_B = GDMIX_FE_B;        // accumulators (rows / columns) per block: 16 KiB of LDS per wavefront
constexpr int FE_U = GDMIX_FE_UNROLL;   // entries per lane in flight
constexpr int FE_THREADS = 256;
constexpr int FE_XCDS = 8;           // accelerator dies of an MI355X, each with its own L2; workgroup i runs on die i % 8
constexpr int FE_WAVES = FE_THREADS / WAVE;
constexpr int FE_DOT_BLOCKS = 512;
constexpr int FE_FIN_BLOCKS = 64;   // workgroups (= lanes of the final wavefront) that add up the per-unit partial sums
static_assert(FE_U % 2 == 0 && FE_B <= 65536 && FE_B % FE_THREADS == 0, "16-bit accumulator index");

// one pass's copy of the non-zeros
constexpr int FE_LOC_BITS = 11;
static_assert((1 << FE_LOC_BITS) == GDMIX_FE_B || !GDMIX_FE_PACK, "the packed word holds the accumulator index in its low bits");
struct FeCopy {
  const uint2* ent;        // packed [z]: x = (key - kbase[unit]) << FE_LOC_BITS | loc, y = bits of the value; else NULL and:
  const int32_t* key;      // [z] element of the gathered vector: local column (row pass) / row (column pass)
  const float* val;        // [z]
  const uint16_t* loc;     // [z] accumulator within the block
  const int32_t* kbase;    // [nunit] key of the unit's first entry (the smallest: a unit's keys ascend)
  const int32_t* ustart;   // [nunit+1] unit -> first entry; units tile the copy
  const int32_t* ublock;   // [nunit]
  const int32_t* ufirst;   // [nblock+1] block -> first unit
  const int32_t* order;    // [nlaunch] workgroup -> unit or -1: units that gather the same stretch of the vector on one XCD (fe_build_copy)
  double* part;            // [nunit][FE_B] partial sums of the blocks that have several units (column pass: of all)
  int nunit, nblock, nlaunch;
  // the 6-byte form (round 5; fe_cpack_kernel): a unit's entries in trips of 512, per trip 2 KB of values + 1 KB of 16-bit words
  // {5-bit key delta to the entry before, 11-bit accumulator}, keys rebuilt by a wavefront scan
  const unsigned char* cdata;
  const int64_t* cbase;    // [nunit] byte offset of the unit in cdata; -1: the unit is read in the 8-byte / three-array form (or NULL)
  const int32_t* ctrip;    // [nunit] trips
  int64_t stream_bytes;    // (host bookkeeping) bytes of entries one pass reads: 6-byte-form units incl. fillers and padding + 8 / 10 B per entry of the others
};

// Frequent features (real feature frequencies are Zipf-like: one feature can hold a tenth of the non-zeros): in the column pass
// their adds pile up on one LDS address (of the 64 lanes of an instruction, those that hold an entry of the same column
// serialise), 0.34 ms instead of 0.25 on the Zipf shard of tools/fe_bench.py. A dense pass per frequent column straight off the
// column-major arrays was tried and lost (0.52 ms): every such column then gathers the residuals on its own, 64 columns =
// 64 sweeps over them instead of one. So the scatter form stays and the frequent columns (at least FE_HOT_MIN entries, the
// FE_HOT_MAX most frequent of them) get FE_HOT_REP accumulators each: in the column pass's copy an entry of frequent column h in
// row r goes to the virtual column vbase + h * FE_HOT_REP + r % FE_HOT_REP. The virtual columns form one more block at the end
// (entries by ascending row like every block: one sweep over the residuals for all of them), neighbouring rows land on different
// accumulators, and fe_hot_finish_block (the first workgroups of fe_finish_kernel) adds a column's FE_HOT_REP sums in replica order. No atomics across workgroups, fixed shape.
constexpr int FE_HOT_MAX = 64;
constexpr int FE_HOT_REP = 32;
static_assert(FE_HOT_MAX * FE_HOT_REP <= FE_B && FE_B % FE_HOT_REP == 0, "the virtual columns are one block");
struct FeHot {
  int n, vbase;              // vbase: first virtual column (a multiple of FE_B, >= d)
  const int32_t* col;        // [n] local column of frequent column h
};

struct FeSync;
struct FeDev {
  int n, d, ic, P, m;
  int64_t z, D;
  FeCopy rc, cc;            // row pass, column pass
  FeHot hot;                // frequent columns: left out of cc
  const int32_t* multi;     // [nmulti] row blocks cut into several units
  int nmulti, nred;         // nred = rc.nunit + nmulti * (workgroups of fe_rows_fix_kernel per block) entries of loss_part / rsum_part
  const float *y, *o, *w;   // w may be NULL
  const int32_t* umap;      // [d] local -> global feature id
  double* xl;               // [d] x of the features present in this shard
  double* rs;               // [n] per-sample residual
  double* fg;               // [P + 1] global data gradient (intercept last), then the data value
  double *loss_part, *rsum_part, *loss_lo_part;   // [nred] each: value (hi), residual sum, value (lo)
  double* acc_part;         // [FE_DOT_BLOCKS][COMPACT_KD]
  double* fin_part;         // [FE_FIN_BLOCKS][3]: value hi, residual sum, value lo
  unsigned* fin_count;      // workgroups of fe_finish_kernel that have delivered their range sums
  int32_t* inv;             // [P] global coefficient -> local column of this shard, -1: absent (intercept: -1)
  struct FeSync* sync;      // ticket + generation stamp of fe_tail_kernel
  CompactState* state;
  CompactPlan* plan;
  CompactMats* mats;
  Work W;                   // global coefficient space, P each; ws / wy m*P
};

__global__ void fe_prepare_kernel(FeDev F) {
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < F.d; j += gridDim.x * blockDim.x) F.xl[j] = F.W.x[F.umap[j]];
}

// ---- the data term's VALUE, added up error-free (round 6) ----------------------------------------------------------------------------
// The value of a shard is a sum of millions of losses: f ~ 5e5 has an ulp of 5.8e-11, and a line search near the optimum asks for
// decreases of that size (tools/fuzz_fe.py case 6100134: five features, 726 k samples, expected decrease of the fifth iteration
// ~ 1.4 ulp). Added over a tree of fp64 sums the value carries a few ulps of noise, every trial of the search fails the
// sufficient-decrease test and the fit ends ABNORMAL where scipy on an accurately summed objective (and the oracle, which adds in
// long double) takes the step and stops on the projected gradient. So the value is added up as (hi, lo) pairs combined with TwoSum
// at EVERY level — a lane's rows, the unit's wavefront, the ranges of fe_finish_kernel, its wavefronts, the final 64 — and rounded
// once at the end: f is the correctly rounded sum of the fp64 loss terms (first version: from the per-unit sums up only; case
// 6500120, 294 k samples, still searched two evaluations longer than the oracle on the per-unit sums' noise). Six more flops per row in
// passes whose vector units idle 90 % of the time; the gradient is not touched.
__device__ __forceinline__ void dd_add(double& hi, double& lo, double x) {      // (hi, lo) += x, Knuth's TwoSum: no error term lost
  const double s = hi + x, bb = s - hi;
  lo += (hi - (s - bb)) + (x - bb);
  hi = s;
}
__device__ __forceinline__ void dd_add2(double& hi, double& lo, double xh, double xl) { dd_add(hi, lo, xh); lo += xl; }
// (hi, lo) of all 64 lanes -> lane 63 holds the total pair (the shift pattern of wave_sum; an invalid source reads 0)
__device__ __forceinline__ void wave_sum_dd(double& hi, double& lo) {
#define GDMIX_DD_STEP(CTRL, MASK)                                                     \
  {                                                                                   \
    const double oh = dpp_get0<CTRL, MASK>(hi), ol = dpp_get0<CTRL, MASK>(lo);        \
    dd_add2(hi, lo, oh, ol);                                                          \
  }
  GDMIX_DD_STEP(0x111, 0xf) GDMIX_DD_STEP(0x112, 0xf) GDMIX_DD_STEP(0x114, 0xf) GDMIX_DD_STEP(0x118, 0xf) GDMIX_DD_STEP(0x142, 0xa) GDMIX_DD_STEP(0x143, 0xc)
#undef GDMIX_DD_STEP
  hi = readlane63(hi);
  lo = readlane63(lo);
}

// what becomes of a finished row sum
// HESS: the pass computes the diagonal of X~' D X~ instead of the gradient (fixed_effect_lr_lbfgs_model.py:271-296): the row pass
// leaves d_i = w_i rho_i (1 - rho_i), rho = sigmoid(logit), the column pass sums val^2 d_i
template <bool HESS>
__device__ __forceinline__ void fe_emit_row(const FeDev& F, const SolveParams& o, int s, double sum, double xb, double& loss, double& loss_lo, double& rsum) {
  const double zi = sum + xb + (double)F.o[s];
  const double yi = (double)F.y[s];
  const double wi = F.w ? (double)F.w[s] : 1.0;
  double ri;
  if (HESS) {
    const double rho = sigmoid_full(zi);
    ri = wi * rho * (1.0 - rho);
  } else if (o.linear) {
    const double e = zi - yi;
    dd_add(loss, loss_lo, wi * e * e);
    ri = 2.0 * wi * e;
  } else {
    dd_add(loss, loss_lo, logistic_terms(zi, yi, wi, ri));
  }
  F.rs[s] = ri;
  rsum += ri;
}

// acc[loc] += term, in LDS, nothing returned
__device__ __forceinline__ void lds_add(double* acc, int loc, double term) {
  __hip_atomic_fetch_add(acc + loc, term, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// inclusive prefix sums over the 64 lanes of two 16-bit counters packed in one word (no carry between the halves: sums < 65 536)
__device__ __forceinline__ unsigned wave_iscan_pair(unsigned v) {
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);   // row_shr:1 (lanes without a source keep 0)
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);   // row_shr:2
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);   // row_shr:4
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);   // row_shr:8: prefix within each row of 16
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1, 3
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2, 3
  return v;
}

#ifndef GDMIX_FE_COMPRESS_DEFAULT
#define GDMIX_FE_COMPRESS_DEFAULT 2     // the column pass: 0.268 -> 0.235 ms; the row pass gets slower in this form (0.246 -> 0.257): measured, docs/rounds/r05.md
#endif
constexpr int FE_COMPRESS_DEFAULT = GDMIX_FE_COMPRESS_DEFAULT;
constexpr int FE_CTRIP = WAVE * 8;            // entries per trip of the 6-byte form
constexpr int FE_CTRIP_BYTES = FE_CTRIP * 6;  // [2][64] x 16 B of values, then [64] x 16 B of index words
constexpr int FE_CDELTA_MAX = 31;
static_assert(FE_LOC_BITS == 11, "index word = delta << 11 | accumulator");

// One unit in the 6-byte form. Trip t: lane l holds entries q * 64 + l, q = 0..7 — every gather instruction covers 64 consecutive
// entries (~100 consecutive elements of the gathered vector), and the adds of one accumulator happen in exact key order.
template <bool ROWS, bool HESS>
__device__ __forceinline__ void fe_scatter_compressed(double* acc, const unsigned char* __restrict__ base, int trips, const double* __restrict__ vec, int lane) {
  int key0 = 0;     // key of the entry before this trip's first, relative to the unit's first (uniform)
  for (int t = 0; t < trips; ++t) {
    const unsigned char* tp = base + (size_t)t * FE_CTRIP_BYTES;
    uint4 v0, v1, ix;
    __builtin_memcpy(&v0, tp + lane * 16, 16);
    __builtin_memcpy(&v1, tp + 1024 + lane * 16, 16);
    __builtin_memcpy(&ix, tp + 2048 + lane * 16, 16);
    const unsigned w[4] = {ix.x, ix.y, ix.z, ix.w};
    const float vq[8] = {__uint_as_float(v0.x), __uint_as_float(v0.y), __uint_as_float(v0.z), __uint_as_float(v0.w),
                         __uint_as_float(v1.x), __uint_as_float(v1.y), __uint_as_float(v1.z), __uint_as_float(v1.w)};
    int kq[8], lq[8];
    double xq[8];
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const unsigned pre = wave_iscan_pair((w[h] >> FE_LOC_BITS) & 0x001f001fu);      // rows 2h (low half) and 2h + 1 (high half)
      const unsigned tot = (unsigned)__builtin_amdgcn_readlane((int)pre, 63);
      kq[2 * h] = key0 + (int)(pre & 0xffffu);
      key0 += (int)(tot & 0xffffu);
      kq[2 * h + 1] = key0 + (int)(pre >> 16);
      key0 += (int)(tot >> 16);
      lq[2 * h] = (int)(w[h] & ((1u << FE_LOC_BITS) - 1));
      lq[2 * h + 1] = (int)((w[h] >> 16) & ((1u << FE_LOC_BITS) - 1));
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) xq[q] = vec[kq[q]];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const double v = (double)vq[q];
      // a filler (value 0: it only carries a key delta too large for five bits) must not turn a non-finite x into a NaN sum
      lds_add(acc, lq[q], vq[q] == 0.0f ? 0.0 : ((HESS && !ROWS) ? v * v * xq[q] : v * xq[q]));
    }
  }
}

template <bool ROWS, bool HESS, bool PACKED>
__global__ __launch_bounds__(WAVE) void fe_scatter_kernel(FeDev F, SolveParams o) {
  // exactly 16 KiB: ten wavefronts per CU (160 KiB of LDS). Round 5: a spare accumulator for the lanes beyond the end of the last
  // trip made it 16 400 B = nine; those lanes now add 0.0 to accumulator 0 (x + 0.0 == x bit for bit; an accumulator that is still
  // -0.0 cannot occur: they start at +0.0). Measured: nine or ten makes no difference (0.556 ms either way, three runs each):
  // the passes are bound by the CU's vector-memory path, not by occupancy (docs/rounds/r05.md)
  __shared__ double acc[FE_B];
  const int lane = threadIdx.x;
  const FeCopy& C = ROWS ? F.rc : F.cc;
  const int u = C.order[blockIdx.x];
  if (u < 0) return;
  if (!HESS && F.state->status >= 0) return;   // the driver has stopped: evaluations enqueued ahead of the status are no-ops
  const int k0 = C.ustart[u], k1 = C.ustart[u + 1], b = C.ublock[u];
  const int kb = PACKED ? C.kbase[u] : 0;
  const bool whole = ROWS && (C.ufirst[b + 1] - C.ufirst[b] == 1);
  const double xb = (ROWS && F.ic) ? F.W.x[F.D] : 0.0;
  const uint2* __restrict__ ent = C.ent;
  const int32_t* __restrict__ key = C.key;
  const float* __restrict__ val = C.val;
  const uint16_t* __restrict__ loc = C.loc;
  const double* __restrict__ vec = (ROWS ? F.xl : F.rs) + kb;
#pragma unroll
  for (int i = 0; i < FE_B / WAVE; i += 2) *reinterpret_cast<double2*>(acc + (i * WAVE + 2 * lane)) = make_double2(0.0, 0.0);
  const int64_t cb = C.cbase ? C.cbase[u] : -1;      // uniform
  if (cb >= 0) {
    fe_scatter_compressed<ROWS, HESS>(acc, C.cdata + cb, C.ctrip[u], (ROWS ? F.xl : F.rs) + C.kbase[u], lane);   // (keys are relative to the unit's first in this form, always)
  } else {
  // full trips without a guard in sight (a load under a branch is waited for inside the branch); the last, partial trip
  // reads clamped addresses and sends what is beyond the end to a spare accumulator.
  // (Round 5, measured and removed: the entry stream two trips ahead of the gathers — prefetch issued BEHIND the gathers so that
  // the in-order return of loads does not make the gathers wait for it, two register sets, clean ISA: row pass 0.232 -> 0.262 ms,
  // column pass 0.247 -> 0.255. More bytes in flight per wavefront make this stream slower, not faster: docs/rounds/r05.md.)
  int base = k0;
  for (; base + WAVE * FE_U <= k1; base += WAVE * FE_U) {
    int kq[FE_U], lq[FE_U];
    float vq[FE_U];
    double xq[FE_U];
#pragma unroll
    for (int q = 0; q < FE_U; ++q) {
      const int k = base + q * WAVE + lane;
      if (PACKED) {
        // two consecutive entries per lane and 16-byte load; the adds keep entry order within a pair, pairs in lane order
        if ((q & 1) == 0) {
          uint4 e2;
          __builtin_memcpy(&e2, ent + (base + (q >> 1) * 2 * WAVE + 2 * lane), 16);
          kq[q] = (int)(e2.x >> FE_LOC_BITS); lq[q] = (int)(e2.x & ((1u << FE_LOC_BITS) - 1)); vq[q] = __uint_as_float(e2.y);
          kq[q + 1] = (int)(e2.z >> FE_LOC_BITS); lq[q + 1] = (int)(e2.z & ((1u << FE_LOC_BITS) - 1)); vq[q + 1] = __uint_as_float(e2.w);
        }
      } else {
        kq[q] = key[k];
        vq[q] = val[k];
        lq[q] = (int)loc[k];
      }
    }
#pragma unroll
    for (int q = 0; q < FE_U; ++q) xq[q] = vec[kq[q]];
#pragma unroll
    for (int q = 0; q < FE_U; ++q) {
      const double v = (double)vq[q];
      lds_add(acc, lq[q], (HESS && !ROWS) ? v * v * xq[q] : v * xq[q]);
    }
  }
  if (base < k1) {
    int kq[FE_U], lq[FE_U];
    float vq[FE_U];
    double xq[FE_U];
    unsigned live = 0u;
#pragma unroll
    for (int q = 0; q < FE_U; ++q) {
      const int k = base + q * WAVE + lane;
      const int kc = k < k1 ? k : k1 - 1;
      int l;
      if (PACKED) {
        const uint2 e = ent[kc];
        kq[q] = (int)(e.x >> FE_LOC_BITS);
        l = (int)(e.x & ((1u << FE_LOC_BITS) - 1));
        vq[q] = __uint_as_float(e.y);
      } else {
        kq[q] = key[kc];
        vq[q] = val[kc];
        l = (int)loc[kc];
      }
      lq[q] = k < k1 ? l : 0;
      live |= (k < k1 ? 1u : 0u) << q;
    }
#pragma unroll
    for (int q = 0; q < FE_U; ++q) xq[q] = vec[kq[q]];
#pragma unroll
    for (int q = 0; q < FE_U; ++q) {
      const double v = (double)vq[q];
      const double term = (HESS && !ROWS) ? v * v * xq[q] : v * xq[q];
      lds_add(acc, lq[q], ((live >> q) & 1u) ? term : 0.0);
    }
  }
  }   // (the 8-byte / three-array forms)
  __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the adds have landed
  __builtin_amdgcn_wave_barrier();
  if (whole) {
    double loss = 0.0, loss_lo = 0.0, rsum = 0.0;
    const int r0 = b * FE_B;
    const int nr = (F.n - r0 < FE_B) ? F.n - r0 : FE_B;
    for (int i = lane; i < nr; i += WAVE) fe_emit_row<HESS>(F, o, r0 + i, acc[i], xb, loss, loss_lo, rsum);
    wave_sum_dd(loss, loss_lo);
    rsum = wave_sum(rsum);
    if (lane == 0) { F.loss_part[u] = loss; F.loss_lo_part[u] = loss_lo; F.rsum_part[u] = rsum; }
  } else {
    double* __restrict__ out = C.part + (size_t)u * FE_B;
#pragma unroll
    for (int i = 0; i < FE_B / WAVE; i += 2)
      *reinterpret_cast<double2*>(out + (i * WAVE + 2 * lane)) = *reinterpret_cast<const double2*>(acc + (i * WAVE + 2 * lane));
    if (ROWS && lane == 0) { F.loss_part[u] = 0.0; F.loss_lo_part[u] = 0.0; F.rsum_part[u] = 0.0; }
  }
}

// Partial sums of one output over the units [u0, u1) of its block. A frequent feature's block can have a thousand units: FE_STRANDS
// threads take every FE_STRANDS-th unit each (several loads in flight), then thread 0 of the output adds the strands in order.
// Fixed shape; with a single unit the result is that unit's value exactly.
constexpr int FE_STRANDS = 16;
constexpr int FE_RED_OUT = FE_THREADS / FE_STRANDS;   // outputs per workgroup; consecutive, so their loads share a line
static_assert(FE_B % FE_RED_OUT == 0, "a workgroup's outputs lie in one block");

__device__ __forceinline__ double fe_strand_sum(const double* __restrict__ part, int u0, int u1, int i, int strand, double (*lds)[FE_RED_OUT],
                                                int out) {
  double t = 0.0;
  int u = u0 + strand;
  for (; u + 3 * FE_STRANDS < u1; u += 4 * FE_STRANDS) {
    const double a0 = part[(size_t)u * FE_B + i], a1 = part[(size_t)(u + FE_STRANDS) * FE_B + i];
    const double a2 = part[(size_t)(u + 2 * FE_STRANDS) * FE_B + i], a3 = part[(size_t)(u + 3 * FE_STRANDS) * FE_B + i];
    t += a0; t += a1; t += a2; t += a3;
  }
  for (; u < u1; u += FE_STRANDS) t += part[(size_t)u * FE_B + i];
  lds[strand][out] = t;
  __syncthreads();
  double g = 0.0;
  if (strand == 0) {
    g = lds[0][out];
#pragma unroll
    for (int k = 1; k < FE_STRANDS; ++k) g += lds[k][out];
  }
  return g;
}

// rows of the blocks that were cut into several units
constexpr int FE_FIX_PER_BLOCK = FE_B / FE_RED_OUT;   // workgroups of fe_rows_fix_kernel per block
template <bool HESS = false>
__global__ __launch_bounds__(FE_THREADS) void fe_rows_fix_kernel(FeDev F, SolveParams o) {
  __shared__ double lds[FE_STRANDS][FE_RED_OUT];
  if (!HESS && F.state->status >= 0) return;
  const int tid = threadIdx.x, out = tid % FE_RED_OUT, strand = tid / FE_RED_OUT;
  const int b = F.multi[blockIdx.x / FE_FIX_PER_BLOCK];
  const int i = (blockIdx.x % FE_FIX_PER_BLOCK) * FE_RED_OUT + out;
  const int row = b * FE_B + i;
  const double xb = F.ic ? F.W.x[F.D] : 0.0;
  const double t = fe_strand_sum(F.rc.part, F.rc.ufirst[b], F.rc.ufirst[b + 1], i, strand, lds, out);
  double loss = 0.0, loss_lo = 0.0, rsum = 0.0;
  if (strand == 0 && row < F.n) fe_emit_row<HESS>(F, o, row, t, xb, loss, loss_lo, rsum);
  if (tid < WAVE) {   // the outputs' threads are the first FE_RED_OUT lanes of wavefront 0
    wave_sum_dd(loss, loss_lo);
    rsum = wave_sum(rsum);
    if (tid == 0) {
      F.loss_part[F.rc.nunit + blockIdx.x] = loss;
      F.loss_lo_part[F.rc.nunit + blockIdx.x] = loss_lo;
      F.rsum_part[F.rc.nunit + blockIdx.x] = rsum;
    }
  }
}

// ---- frequent columns: FE_HOT_REP accumulators each (FeHot above) -------------------------------------------------------------
// next to fe_finish_kernel's own workgroups (which find 0 for a frequent column — the copy holds no entry under its own number — and
// write nothing then: the buffer is clear before an evaluation). One workgroup per
// frequent column: replica r's sum over the virtual block's units by 8 strands, strands in order, then the replicas in order.
// (Round 6: the workgroups of this step are the FIRST F.hot.n workgroups of fe_finish_kernel's grid, not a launch of their own — one
// launch boundary (~6 us on this device) and 10 us of a nearly empty device less per evaluation of a shard with frequent columns.)
__device__ __forceinline__ void fe_hot_finish_block(const FeDev& F, int h) {
  constexpr int STR = FE_THREADS / FE_HOT_REP;
  __shared__ double lds[STR][FE_HOT_REP];
  __shared__ double rep[FE_HOT_REP];
  const FeHot& H = F.hot;
  const int tid = threadIdx.x, r = tid % FE_HOT_REP, strand = tid / FE_HOT_REP;
  const int b = H.vbase / FE_B, i = h * FE_HOT_REP + r;
  const int u0 = F.cc.ufirst[b], u1 = F.cc.ufirst[b + 1];
  const double* __restrict__ part = F.cc.part;
  double t = 0.0;
  int u = u0 + strand;
  for (; u + 3 * STR < u1; u += 4 * STR) {
    const double a0 = part[(size_t)u * FE_B + i], a1 = part[(size_t)(u + STR) * FE_B + i];
    const double a2 = part[(size_t)(u + 2 * STR) * FE_B + i], a3 = part[(size_t)(u + 3 * STR) * FE_B + i];
    t += a0; t += a1; t += a2; t += a3;
  }
  for (; u < u1; u += STR) t += part[(size_t)u * FE_B + i];
  lds[strand][r] = t;
  __syncthreads();
  if (strand == 0) {
    double g = lds[0][r];
#pragma unroll
    for (int k = 1; k < STR; ++k) g += lds[k][r];
    rep[r] = g;
  }
  __syncthreads();
  if (tid == 0) {
    double g = rep[0];
#pragma unroll
    for (int k = 1; k < FE_HOT_REP; ++k) g += rep[k];
    F.fg[F.umap[H.col[h]]] = g;
  }
}

// local gradient (the column blocks' partial sums) into the global coefficient space; the first FE_FIN_BLOCKS workgroups also add
// up a contiguous range of the per-unit value / residual sums each
template <bool HESS = false>
__global__ __launch_bounds__(FE_THREADS) void fe_finish_kernel(FeDev F) {
  __shared__ double lds[FE_STRANDS][FE_RED_OUT];
  __shared__ double red[3][FE_WAVES];
  if (!HESS && F.state->status >= 0) return;
  // the frequent columns' workgroups come FIRST in the grid: each is a long chain (a thousand units' sums by eight strands) and must
  // start with the launch, not behind it
  if ((int)blockIdx.x < F.hot.n) { fe_hot_finish_block(F, (int)blockIdx.x); return; }
  const int bid = (int)blockIdx.x - F.hot.n;
  const int tid = threadIdx.x, out = tid % FE_RED_OUT, strand = tid / FE_RED_OUT;
  const int j0 = bid * FE_RED_OUT;
  if (j0 < F.d) {   // workgroup-uniform
    const int jj = j0 + out < F.d ? j0 + out : F.d - 1;
    const int b = jj / FE_B, i = jj % FE_B;
    const double g = fe_strand_sum(F.cc.part, F.cc.ufirst[b], F.cc.ufirst[b + 1], i, strand, lds, out);
    // (an exact 0 is not written: the buffer is clear before an evaluation — gdmix_fe_eval / the step's dots — and a frequent column's
    // own slot, which holds no entry here, belongs to the workgroup that adds up its replicas in this same launch)
    if (strand == 0 && j0 + out < F.d && g != 0.0) F.fg[F.umap[jj]] = g;
  }
  if (bid >= FE_FIN_BLOCKS) return;
  const int lane = tid & (WAVE - 1), wv = tid >> 6;
  const int chunk = (F.nred + FE_FIN_BLOCKS - 1) / FE_FIN_BLOCKS;
  const int b0 = bid * chunk;
  const int b1 = (b0 + chunk < F.nred) ? b0 + chunk : F.nred;
  double a = 0.0, al = 0.0, r = 0.0;
  for (int b = b0 + tid; b < b1; b += FE_THREADS) {
    dd_add2(a, al, F.loss_part[b], F.loss_lo_part[b]);
    r += F.rsum_part[b];
  }
  wave_sum_dd(a, al);
  r = wave_sum(r);
  if (lane == 0) { red[0][wv] = a; red[1][wv] = r; red[2][wv] = al; }
  __syncthreads();
  __shared__ int last;
  if (tid == 0) {
    double sa = red[0][0], sl = red[2][0], sr = red[1][0];
#pragma unroll
    for (int w = 1; w < FE_WAVES; ++w) { dd_add2(sa, sl, red[0][w], red[2][w]); sr += red[1][w]; }
    st_x<true>(F.fin_part + 3 * bid, sa);
    st_x<true>(F.fin_part + 3 * bid + 1, sr);
    st_x<true>(F.fin_part + 3 * bid + 2, sl);
    __threadfence();
    last = atomicAdd(F.fin_count, 1u) == FE_FIN_BLOCKS - 1;
  }
  __syncthreads();
  // the workgroup that finishes last adds the FE_FIN_BLOCKS range sums (in lane order, whichever workgroup it is) -> data value and
  // intercept gradient (HESS: fg[D] = sum_i d_i, the intercept's entry; fg[P] unused)
  if (last && tid < WAVE) {
    __threadfence();
    double a = tid < FE_FIN_BLOCKS ? ld_x<true>(F.fin_part + 3 * tid) : 0.0, al = tid < FE_FIN_BLOCKS ? ld_x<true>(F.fin_part + 3 * tid + 2) : 0.0;
    wave_sum_dd(a, al);
    const double r = wave_sum(tid < FE_FIN_BLOCKS ? ld_x<true>(F.fin_part + 3 * tid + 1) : 0.0);
    if (tid == 0) {
      if (F.ic) F.fg[F.D] = r;
      F.fg[F.P] = a + al;      // rounded once
      *F.fin_count = 0u;
    }
  }
}

// the column pass's source columns with the frequent ones replaced by their virtual columns; one thread per row
__global__ void fe_hot_remap_kernel(const int32_t* __restrict__ ptr, int n, const int32_t* __restrict__ col, const int32_t* __restrict__ hotmap,
                                    int vbase, int32_t* __restrict__ col2) {
  for (int row = blockIdx.x * blockDim.x + threadIdx.x; row < n; row += gridDim.x * blockDim.x) {
    const int k1 = ptr[row + 1];
    for (int k = ptr[row]; k < k1; ++k) {
      const int c = col[k];
      const int h = hotmap[c];
      // the replica by the entry's POSITION, not by its row (round 4): a wavefront's 128 entries in flight are consecutive in row
      // order and span only ~4 rows of a 32-non-zero shard, so row % 32 sent all entries of all frequent columns of an instruction to
      // four replica slots = four LDS bank pairs (replicas of different columns 256 B apart share banks): SQ_LDS_BANK_CONFLICT + 62 %
      // against a uniform shard (profiles/r04_fe_counters.txt). The position spreads them over all 32; still a fixed assignment, so a
      // replica's terms are added in row order and two fits give the same bits.
      col2[k] = h >= 0 ? vbase + h * FE_HOT_REP + (k % FE_HOT_REP) : c;
    }
  }
}

// g = reduced data gradient + regulariser, and every product the driver needs (re_lbfgs_compact.hpp acc[] layout): the share of
// "virtual block" vb of nvb — coefficients (vb * 256 + tid) + k * nvb * 256 — into acc_part[vb]. The partition is a function of P
// alone (nvb = min(ceil(P / 256), FE_DOT_BLOCKS)), not of the launch: fe_dots_kernel runs one workgroup per virtual block,
// fe_tail_kernel deals them over fewer resident workgroups, and both give the same bits.
template <bool SC1 = false>
__device__ __forceinline__ void fe_dots_block(const FeDev& F, const SolveParams& o, int vb, int nvb, double (*red)[COMPACT_KD]) {
  const int tid = threadIdx.x, lane = tid & (WAVE - 1), wv = tid >> 6;
  const int col = F.state->col, head = F.state->head, m = o.m, P = F.P;
  double acc[COMPACT_KD];   // ... and S'g, Y'g taken directly behind the TEAM_K products (re_lbfgs_compact.hpp: why)
#pragma unroll
  for (int k = 0; k < COMPACT_KD; ++k) acc[k] = 0.0;
  for (int j = vb * FE_THREADS + tid; j < P; j += nvb * FE_THREADS) {
    const bool reg = (j < F.D) || o.regularize_bias;   // the intercept is coefficient D
    const double xj = F.W.x[j];
    const double gj = F.fg[j] + (reg ? o.l2 * xj : 0.0);
    F.fg[j] = 0.0;   // consumed: the next evaluation starts from a clear buffer (gdmix_fe_eval)
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
      const double hx = i < col ? h[i].x : 0.0, hy = i < col ? h[i].y : 0.0;
      acc[5 + i] += hx * yj;
      acc[5 + TEAM_MCAP + i] += hy * yj;
      acc[TEAM_K + i] += hx * gj;
      acc[TEAM_K + TEAM_MCAP + i] += hy * gj;
    }
  }
  static_assert(COMPACT_KD <= WAVE, "one lane per value");
  double mine = 0.0;
#pragma unroll
  for (int k = 0; k < COMPACT_KD; ++k) {
    const double t = (k == TEAM_K - 1) ? wave_max_nonneg(acc[k]) : wave_sum(acc[k]);
    if (lane == k) mine = t;
  }
  if (lane < COMPACT_KD) red[wv][lane] = mine;
  __syncthreads();
  if (tid < COMPACT_KD) {
    double s = red[0][tid];
#pragma unroll
    for (int w = 1; w < FE_WAVES; ++w) s = (tid == TEAM_K - 1) ? fmax(s, red[w][tid]) : s + red[w][tid];
    st_x<SC1>(F.acc_part + (size_t)vb * COMPACT_KD + tid, s);   // SC1 (fe_tail_kernel): written through, read by another workgroup of the same launch
  }
}

__global__ __launch_bounds__(FE_THREADS) void fe_dots_kernel(FeDev F, SolveParams o) {
  __shared__ double red[FE_WAVES][COMPACT_KD];
  if (F.state->status >= 0) return;     // a step enqueued behind the stop (gdmix_fe_step_async) is a no-op
  fe_dots_block(F, o, blockIdx.x, gridDim.x, red);
}

// one workgroup: totals of the products (the virtual blocks' shares, in a fixed order), then the driver's decision
template <bool SC1 = false>
__device__ __forceinline__ void fe_step_body(const FeDev& F, const SolveParams& o, int dot_blocks, int32_t* status_out, double* tot /* LDS [COMPACT_KD] */,
                                             CompactMats& mats /* LDS */, double (*part4)[WAVE] /* LDS [4][64] */) {
  const int tid = threadIdx.x;
  {
    // value v of block b by thread (b % 4) * 64 + v, four partial totals per value, combined in order
    static_assert(FE_THREADS == 4 * WAVE && COMPACT_KD <= WAVE, "four groups of one lane per value");
    const int v = tid & (WAVE - 1), g = tid >> 6;
    if (v < COMPACT_KD) {
      double s = 0.0;
      int b = g;
      for (; b + 60 < dot_blocks; b += 64) {   // sixteen loads in flight (four groups take the blocks where eight did: the chain is twice as long)
        double t[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) t[q] = ld_x<SC1>(F.acc_part + (size_t)(b + 4 * q) * COMPACT_KD + v);
#pragma unroll
        for (int q = 0; q < 16; ++q) s = (v == TEAM_K - 1) ? fmax(s, t[q]) : s + t[q];
      }
      for (; b + 28 < dot_blocks; b += 32) {   // eight
        double t[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) t[q] = ld_x<SC1>(F.acc_part + (size_t)(b + 4 * q) * COMPACT_KD + v);
#pragma unroll
        for (int q = 0; q < 8; ++q) s = (v == TEAM_K - 1) ? fmax(s, t[q]) : s + t[q];
      }
      for (; b < dot_blocks; b += 4) {
        const double t = ld_x<SC1>(F.acc_part + (size_t)b * COMPACT_KD + v);
        s = (v == TEAM_K - 1) ? fmax(s, t) : s + t;
      }
      part4[g][v] = s;
    }
    __syncthreads();
    if (tid < COMPACT_KD) {
      double s = part4[0][tid];
#pragma unroll
      for (int k = 1; k < 4; ++k) s = (tid == TEAM_K - 1) ? fmax(s, part4[k][tid]) : s + part4[k][tid];
      tot[tid] = s;
    }
  }
  {
    const double* src = reinterpret_cast<const double*>(F.mats);
    double* dst = reinterpret_cast<double*>(&mats);
    for (int k = tid; k < (int)(sizeof(CompactMats) / sizeof(double)); k += FE_THREADS) dst[k] = src[k];
  }
  __syncthreads();
  double acc[COMPACT_KD];
#pragma unroll
  for (int k = 0; k < COMPACT_KD; ++k) acc[k] = tot[k];
  CompactState S = *F.state;
  CompactPlan plan;
  plan.action = CA_STOP; plan.col = S.col; plan.head = S.head; plan.stp = S.stp; plan.gamma = 1.0;
  const double f_new = F.fg[F.P] + 0.5 * o.l2 * acc[0];
  // The data term is a sum of non-negative losses: - infinity is the mark a worker whose step was ABORTED leaves in its value slot
  // (fe_tail_kernel's watchdog), and the all-reduce has carried it to every worker — all of them stop in the same evaluation, none
  // is left alone in a collective (fixed_effect.run_stepping_loop raises on every worker).
  if (F.fg[F.P] == -__builtin_inf()) S.status = GDMIX_RE_ST_ABORTED_PEER;
  else compact_advance(S, acc, f_new, o, mats, plan, true, acc + TEAM_K);
  __syncthreads();
  {
    double* dst = reinterpret_cast<double*>(F.mats);
    const double* src = reinterpret_cast<const double*>(&mats);
    for (int k = tid; k < (int)(sizeof(CompactMats) / sizeof(double)); k += FE_THREADS) st_x<SC1>(dst + k, src[k]);
  }
  if (tid == 0) {
    *F.state = S;        // (read by later launches only)
    if (SC1) {           // the plan is read by the other workgroups of this launch: word by word, written through
      static_assert(sizeof(CompactPlan) % 4 == 0, "plan copied as 32-bit words");
      const unsigned* src = reinterpret_cast<const unsigned*>(&plan);
      unsigned* dst = reinterpret_cast<unsigned*>(F.plan);
      for (int k = 0; k < (int)(sizeof(CompactPlan) / 4); ++k) __hip_atomic_store(dst + k, src[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      *F.plan = plan;
    }
    *status_out = (plan.action == CA_STOP || plan.action == CA_STOP_RESTORE) ? S.status : -1;
  }
}

__global__ __launch_bounds__(FE_THREADS) void fe_step_kernel(FeDev F, SolveParams o, int dot_blocks, int32_t* status_out) {
  __shared__ double tot[COMPACT_KD];
  __shared__ CompactMats mats;
  __shared__ double part4[4][WAVE];
  if (F.state->status >= 0) return;     // (status_out keeps the status of the stop; fe_update_kernel repeats the stop's plan: idempotent)
  fe_step_body(F, o, dot_blocks, status_out, tot, mats, part4);
}

// the elementwise part of a step for coefficient j; the shard's copy of x in local order follows it (the row pass gathers from xl)
__device__ __forceinline__ void fe_update_one(const FeDev& F, const CompactPlan& plan, const CompactMats& mats, int m, int j) {
  if (plan.action == CA_STOP_RESTORE) F.W.x[j] = F.W.t[j];
  else compact_update(plan, mats, F.W, F.P, m, j);
  const int jl = F.inv[j];
  if (jl >= 0) F.xl[jl] = F.W.x[j];
}

// ---- the whole step in ONE launch (round 5) ----------------------------------------------------------------------------------------
// dots -> decision -> update were three launches (9.4 + 8.4 + 5.9 us and two boundaries) plus fe_prepare_kernel ahead of the next
// evaluation. Here: at most one workgroup per CU (all resident: the wait below cannot starve anyone), each takes its virtual blocks'
// share of the products; the workgroup that arrives last adds the shares up in the fixed order and takes the driver's decision;
// the others wait for its generation stamp, then every workgroup updates the coefficients of its own virtual blocks — the ones
// whose gradient it wrote itself, so the only data crossing workgroups are the shares (to the last arriver) and state / plan / the
// m x m matrices (back). Hand-off per MI355X_MICROARCH.md: plain stores, one lane's agent-scope release + drained store queue,
// relaxed agent atomics for ticket and stamp, one lane's agent-scope acquire, workgroup barrier, plain loads.
// A stopped problem (status >= 0 from an earlier launch) makes this and every pass kernel return at once: the host may enqueue
// evaluations ahead of the status it has read (gdmix_fe_step_async).
struct FeSync { unsigned arrive, gen, aborted; };      // aborted: sticky, set by a waiter whose watchdog fired (never cleared: the problem is dead)

__global__ __launch_bounds__(FE_THREADS) void fe_tail_kernel(FeDev F, SolveParams o, int dot_blocks, int32_t* status_out, unsigned seq) {
  __shared__ double red[FE_WAVES][COMPACT_KD];
  __shared__ double tot[COMPACT_KD];
  __shared__ CompactMats mats;
  __shared__ double part4[4][WAVE];
  __shared__ CompactPlan plan_s;
  __shared__ int last, timed_out;
  if (F.state->status >= 0) return;     // written by an earlier launch: uniform over the grid
  if (threadIdx.x == 0) timed_out = 0;
  const int tid = threadIdx.x;
  for (int vb = blockIdx.x; vb < dot_blocks; vb += gridDim.x) {
    fe_dots_block<true>(F, o, vb, dot_blocks, red);
    __syncthreads();                     // red is reused
  }
  // What crosses workgroups inside this launch — the shares, then plan and matrices — is written with write-through (sc1) stores
  // and read with sc1 loads, so the hand-off needs a drained store queue and the counters only: no L2 write-back, no L1 invalidate
  // (MI355X_MICROARCH.md, "valid forms": sc1 stores and loads on both sides). The first version used plain accesses with an
  // agent-scope release / acquire pair on either side of both hand-offs: the kernel took 36 - 45 us for 24 us of work
  // (profiles/r05_fe_counters.txt).
  if (tid == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    last = __hip_atomic_fetch_add(&F.sync->arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
  }
  __syncthreads();                       // (the shares are stored by threads of wavefront 0, whose queue thread 0 has just drained)
  if (last) {
    fe_step_body<true>(F, o, dot_blocks, status_out, tot, mats, part4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every wavefront: its part of the matrices is at the memory side
    __syncthreads();
    if (tid == 0) {
      // a waiter of this launch gave up before this workgroup arrived: its ABORTED must survive the state / status this
      // workgroup has just written — the coefficients of the workgroups that left are not updated, the fit is invalid
      if (__hip_atomic_load(&F.sync->aborted, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
        __hip_atomic_store(reinterpret_cast<int*>(&F.state->status), GDMIX_RE_ST_ABORTED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(status_out, (int32_t)GDMIX_RE_ST_ABORTED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __hip_atomic_store(&F.sync->arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&F.sync->gen, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (tid == 0) {
    if (!last) {
      // bounded: a workgroup of this launch that never arrives (it cannot happen while at most one workgroup per CU is launched on an
      // otherwise progressing device) must not hang the GPU — after ~4 s the problem is marked ABORTED, which also stops every later launch
      unsigned spins = 0;
      uint64_t t0 = 0;
      while (__hip_atomic_load(&F.sync->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != seq) {
        __builtin_amdgcn_s_sleep(2);
        if ((++spins & 1023u) == 0) {
          const uint64_t now = wall_clock64();
          if (t0 == 0) t0 = now;
          else if (now - t0 > 400000000ull) {   // 4 s at 100 MHz
            __hip_atomic_store(&F.sync->aborted, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // first: the last arriver reads it
            F.fg[F.P] = -__builtin_inf();      // what this worker contributes to the next all-reduce: every worker stops (fe_step_body)
            __hip_atomic_store(reinterpret_cast<int*>(&F.state->status), GDMIX_RE_ST_ABORTED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(status_out, (int32_t)GDMIX_RE_ST_ABORTED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            timed_out = 1;
            break;
          }
        }
      }
    }
    const unsigned* src = reinterpret_cast<const unsigned*>(F.plan);
    unsigned* dst = reinterpret_cast<unsigned*>(&plan_s);
    for (int k = 0; k < (int)(sizeof(CompactPlan) / 4); ++k) dst[k] = __hip_atomic_load(src + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (timed_out) return;
  if (!last) {
    const double* src = reinterpret_cast<const double*>(F.mats);
    double* dst = reinterpret_cast<double*>(&mats);
    for (int k = tid; k < (int)(sizeof(CompactMats) / sizeof(double)); k += FE_THREADS) dst[k] = ld_x<true>(src + k);
    __syncthreads();
  }
  const CompactPlan plan = plan_s;
  if (plan.action == CA_STOP) return;
  for (int vb = blockIdx.x; vb < dot_blocks; vb += gridDim.x)
    for (int j = vb * FE_THREADS + tid; j < F.P; j += dot_blocks * FE_THREADS) fe_update_one(F, plan, mats, o.m, j);
}

__global__ void fe_update_kernel(FeDev F, int m) {
  const CompactPlan plan = *F.plan;
  if (plan.action == CA_STOP) return;
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < F.P; j += gridDim.x * blockDim.x) fe_update_one(F, plan, *F.mats, m, j);
}

__global__ void fe_init_kernel(FeDev F, const double* __restrict__ theta0) {
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < F.P; j += gridDim.x * blockDim.x) {
    F.W.x[j] = theta0 ? theta0[j] : 0.0;
    F.W.d[j] = 0.0;
    F.W.r[j] = 0.0;
    F.W.g[j] = 0.0;
    F.W.t[j] = 0.0;
  }
  // the shard's local copy of the start point (afterwards the update keeps it current: fe_update_one)
  for (int jl = blockIdx.x * blockDim.x + threadIdx.x; jl < F.d; jl += gridDim.x * blockDim.x) {
    const int64_t j = F.umap[jl];      // (widened: an index into the global coefficient space)
    F.xl[jl] = theta0 ? theta0[j] : 0.0;
    F.inv[j] = jl;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    CompactState S;
    compact_init(S);
    *F.state = S;
  }
}

// ---- the passes' copies of the non-zeros ---------------------------------------------------------------------------------
// From the packed shard's CSR (for the column pass) and CSC (for the row pass) arrays: segment of every entry (flag + scan),
// stable sort by block of the entry's index (rocPRIM radix sort on the block number alone, so the source order — the order of
// the gathered vector — survives inside a block), units = the blocks' runs cut every `chunk` entries.
__global__ void fe_flag_kernel(const int32_t* __restrict__ ptr, int nseg, int64_t z, int32_t* __restrict__ flag) {
  for (int s = blockIdx.x * blockDim.x + threadIdx.x + 1; s < nseg; s += gridDim.x * blockDim.x) {
    const int p = ptr[s];
    if (p < z) atomicAdd(&flag[p], 1);   // empty segments pile up on the next entry
  }
}

struct FeEnt { int32_t seg, idx; float val; };   // an entry on its way through the sort

// Sort key: the block, refined by the window of 2^FE_SPAN_BITS gathered elements the entry's key lies in. The entries of a block
// already come by ascending key, so the windows do not change the sorted order; they only add cut points, so that no unit's keys
// span more than the packed word can hold (a block with few entries is one unit over the whole vector otherwise, and one such
// unit would send the whole copy to the three-array form: the Zipf shard of tools/fe_bench.py, 0.34 ms instead of 0.26).
constexpr int FE_SPAN_BITS = 32 - FE_LOC_BITS;
__global__ void fe_ent_kernel(const int32_t* __restrict__ seg, const int32_t* __restrict__ idx, const float* __restrict__ val, int64_t z,
                              int nwin, int wbits, uint32_t* __restrict__ skey, FeEnt* __restrict__ ent) {
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < z; k += (int64_t)gridDim.x * blockDim.x) {
    const int i = idx[k], sg = seg[k];
    skey[k] = (uint32_t)(i / FE_B) * (uint32_t)nwin + (uint32_t)(sg >> wbits);
    ent[k] = FeEnt{sg, i, val[k]};
  }
}

// widest unit: key of its last entry - key of its first (keys ascend inside a unit)
__global__ void fe_span_kernel(const FeEnt* __restrict__ ent, const int32_t* __restrict__ ustart, int nunit, int32_t* __restrict__ kbase,
                               int32_t* __restrict__ max_span) {
  for (int u = blockIdx.x * blockDim.x + threadIdx.x; u < nunit; u += gridDim.x * blockDim.x) {
    const int k0 = ustart[u], k1 = ustart[u + 1];
    const int first = k1 > k0 ? ent[k0].seg : 0;
    kbase[u] = first;
    if (k1 > k0) atomicMax(max_span, ent[k1 - 1].seg - first);
  }
}

// the copy in the form the pass reads; one workgroup per unit
template <bool PACKED>
__global__ __launch_bounds__(256) void fe_pack_kernel(const FeEnt* __restrict__ ent, const int32_t* __restrict__ ustart,
                                                      const int32_t* __restrict__ kbase, uint2* __restrict__ out, int32_t* __restrict__ ckey,
                                                      float* __restrict__ cval, uint16_t* __restrict__ cloc) {
  const int u = blockIdx.x;
  const int k0 = ustart[u], k1 = ustart[u + 1], kb = kbase[u];
  for (int k = k0 + threadIdx.x; k < k1; k += 256) {
    const FeEnt e = ent[k];
    const int l = e.idx % FE_B;
    if (PACKED) {
      out[k] = make_uint2(((uint32_t)(e.seg - kb) << FE_LOC_BITS) | (uint32_t)l, __float_as_uint(e.val));
    } else {
      ckey[k] = e.seg;
      cval[k] = e.val;
      cloc[k] = (uint16_t)l;
    }
  }
}

// ---- the 6-byte form of a unit's entries --------------------------------------------------------------------------------------
// fillers an entry needs in front of it so that every key delta fits FE_CDELTA_MAX: a gap g > 31 takes (g - 1) / 31 fillers of
// delta 31 (value 0, accumulator 0) and leaves a delta in [1, 31] for the entry itself
__device__ __forceinline__ int fe_fillers(int gap) { return gap > FE_CDELTA_MAX ? (gap - 1) / FE_CDELTA_MAX : 0; }

// entries of unit u in the 6-byte form, fillers included (one workgroup per unit)
__global__ __launch_bounds__(256) void fe_ccount_kernel(const FeEnt* __restrict__ ent, const int32_t* __restrict__ ustart, const int32_t* __restrict__ kbase,
                                                        int32_t* __restrict__ cnt) {
  __shared__ int red[256 / WAVE];
  const int u = blockIdx.x, k0 = ustart[u], k1 = ustart[u + 1], kb = kbase[u];
  int f = 0;
  for (int k = k0 + threadIdx.x; k < k1; k += 256) f += fe_fillers(ent[k].seg - (k > k0 ? ent[k - 1].seg : kb));
  for (int sh = 32; sh > 0; sh >>= 1) f += __shfl_down(f, sh);
  if ((threadIdx.x & (WAVE - 1)) == 0) red[threadIdx.x >> 6] = f;
  __syncthreads();
  if (threadIdx.x == 0) cnt[u] = (k1 - k0) + red[0] + red[1] + red[2] + red[3];
}

__device__ __forceinline__ void fe_cput(unsigned char* base, int pos, float val, unsigned word) {
  const int t = pos / FE_CTRIP, w = pos % FE_CTRIP, q = w / WAVE, l = w % WAVE;
  unsigned char* tp = base + (size_t)t * FE_CTRIP_BYTES;
  *reinterpret_cast<float*>(tp + (q >> 2) * 1024 + l * 16 + (q & 3) * 4) = val;
  *reinterpret_cast<uint16_t*>(tp + 2048 + l * 16 + q * 2) = (uint16_t)word;
}

// the units that take the form (cbase[u] >= 0), written into a zeroed buffer: what stays zero is padding (delta 0, value 0)
__global__ __launch_bounds__(256) void fe_cpack_kernel(const FeEnt* __restrict__ ent, const int32_t* __restrict__ ustart, const int32_t* __restrict__ kbase,
                                                       const int64_t* __restrict__ cbase, unsigned char* __restrict__ cdata) {
  __shared__ int wsum[256 / WAVE];
  __shared__ int carry;
  const int u = blockIdx.x;
  if (cbase[u] < 0) return;
  const int k0 = ustart[u], k1 = ustart[u + 1], kb = kbase[u];
  unsigned char* base = cdata + cbase[u];
  const int tid = threadIdx.x, lane = tid & (WAVE - 1), wv = tid >> 6;
  if (tid == 0) carry = 0;
  __syncthreads();
  for (int c0 = k0; c0 < k1; c0 += 256) {
    const int k = c0 + tid;
    int gap = 0, f = 0;
    FeEnt e{0, 0, 0.0f};
    if (k < k1) {
      e = ent[k];
      gap = e.seg - (k > k0 ? ent[k - 1].seg : kb);
      f = fe_fillers(gap);
    }
    // exclusive prefix of f over the 256 entries of this chunk
    int incl = f;
    for (int sh = 1; sh < WAVE; sh <<= 1) {
      const int up = __shfl_up(incl, sh);
      if (lane >= sh) incl += up;
    }
    if (lane == WAVE - 1) wsum[wv] = incl;
    __syncthreads();
    int before = carry;
    for (int w2 = 0; w2 < wv; ++w2) before += wsum[w2];
    const int total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    if (k < k1) {
      const int first = (k - k0) + before + (incl - f);      // position of this entry's first filler (or of the entry)
      for (int j = 0; j < f; ++j) fe_cput(base, first + j, 0.0f, (unsigned)FE_CDELTA_MAX << FE_LOC_BITS);
      fe_cput(base, first + f, e.val, ((unsigned)(gap - FE_CDELTA_MAX * f) << FE_LOC_BITS) | (unsigned)(e.idx % FE_B));
    }
    __syncthreads();
    if (tid == 0) carry += total;
    __syncthreads();
  }
}

// bp[b] = first sorted entry of a block >= b
__global__ void fe_block_kernel(const uint32_t* __restrict__ sorted, int64_t z, int nblock, int32_t* __restrict__ bp) {
  for (int b = blockIdx.x * blockDim.x + threadIdx.x; b <= nblock; b += gridDim.x * blockDim.x) {
    int64_t lo = 0, hi = z;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if (sorted[mid] < (uint32_t)b) lo = mid + 1; else hi = mid;
    }
    bp[b] = (int32_t)lo;
  }
}

// Entries per unit of a (block, window) with len entries. A unit is one wavefront; where a block has fewer than four entries
// per 128-byte line of the gathered vector, nearly every gather is a line of its own out of the far cache and the unit crawls
// at a few microseconds per trip of 512 entries: 62 500 entries = 0.3 ms, the length of the whole pass (the rare features' blocks
// of a Zipf shard). Such blocks get units an eighth as long (sparse_chunk; 0 = never).
__device__ __forceinline__ int fe_block_chunk(int len, int chunk, int sparse_chunk, int extent) {
  return (sparse_chunk > 0 && (int64_t)len * 4 < (int64_t)extent) ? sparse_chunk : chunk;
}

// units of (block, window) b; the first window of a block keeps one even when empty: the block's outputs are still due
__global__ void fe_chunks_kernel(const int32_t* __restrict__ bp, int nblock, int nwin, int chunk, int sparse_chunk, int extent,
                                 int32_t* __restrict__ nch) {
  for (int b = blockIdx.x * blockDim.x + threadIdx.x; b <= nblock; b += gridDim.x * blockDim.x) {
    const int len = b < nblock ? bp[b + 1] - bp[b] : 0;
    const int c = fe_block_chunk(len, chunk, sparse_chunk, extent);
    nch[b] = b < nblock ? (len == 0 ? (b % nwin == 0 ? 1 : 0) : (len + c - 1) / c) : 0;
  }
}

// ufirst: first unit per (block, window); out: the units' first entries and blocks, and first unit per block
__global__ void fe_units_kernel(const int32_t* __restrict__ bp, const int32_t* __restrict__ ufirst, int nblock, int nwin, int chunk,
                                int sparse_chunk, int extent, int64_t z, int32_t* __restrict__ ustart, int32_t* __restrict__ ublock,
                                int32_t* __restrict__ block_first) {
  for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < nblock; b += gridDim.x * blockDim.x) {
    const int u0 = ufirst[b], u1 = ufirst[b + 1];
    const int c = fe_block_chunk(bp[b + 1] - bp[b], chunk, sparse_chunk, extent);
    for (int u = u0; u < u1; ++u) {
      ustart[u] = bp[b] + (u - u0) * c;
      ublock[u] = b / nwin;
    }
    if (b % nwin == 0) block_first[b / nwin] = u0;
    if (b == nblock - 1) { ustart[u1] = (int32_t)z; block_first[nblock / nwin] = u1; }
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
  void* pool;            // one device allocation carved into the vectors and partial sums
  void* copies[2];       // the row pass's and the column pass's copy of the non-zeros, with their unit tables
  void* ccopies[2];      // ... and their units in the 6-byte form
  int compress;          // bit 0: row pass, bit 1: column pass may use the 6-byte form (GDMIX_FE_COMPRESS; default: FE_COMPRESS_DEFAULT)
  void* hot_mem;         // the frequent columns' tables
  int32_t* status_dev;
  hipEvent_t ev[3];
  bool timed;
  bool dirty;            // the reduce buffer holds a result no step has consumed (and cleared) yet
  bool fused_tail;       // the step is one launch (fe_tail_kernel); GDMIX_FE_FUSED_TAIL=0: dots / step / update as three (A/B)
  unsigned gen;          // launches of fe_tail_kernel so far (its generation stamp)
  int64_t evals;         // gdmix_fe_eval calls so far
  int64_t seq;           // steps enqueued so far; step k's status lands in status_ring[k % FE_RING] behind ring_ev[k % FE_RING]
  int32_t* status_ring;  // page-locked
  hipEvent_t ring_ev[8];
  std::vector<int32_t> uf_c;
};
constexpr int FE_RING = 8;

#define HIP_TRY(expr)                                                                   \
  do {                                                                                  \
    hipError_t _rc = (expr);                                                            \
    if (_rc != hipSuccess) {                                                            \
      set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_rc), __FILE__, __LINE__); \
      return GDMIX_RE_EHIP;                                                             \
    }                                                                                   \
  } while (0)

static size_t up256(size_t x) { return (x + 255) & ~(size_t)255; }

// Entries per unit. Blocks stay whole (a row block then finishes its rows itself) when that still gives the device enough
// units; otherwise every block is cut so that there are about eight wavefronts per CU.
static int fe_chunk_len(int64_t z, int nblock, int num_cus) {
  const int64_t target = (z + (int64_t)num_cus * 8 - 1) / ((int64_t)num_cus * 8);
  const int64_t avg = (z + nblock - 1) / nblock;
  int64_t c = (avg <= 2 * target) ? 2 * avg : target;
  if (c < 8192) c = 8192;
  if (const char* e = getenv("GDMIX_FE_CHUNK")) {   // test hook: small shards through the several-units-per-block code
    const long v = atol(e);
    if (v >= 64) c = v;
  }
  if (c > (1 << 28)) c = 1 << 28;
  return (int)c;
}

// Build one pass's copy from segment-major source arrays (ptr [nseg+1], idx / val [z]); `len` = extent of idx (outputs of the
// pass). Device memory of the result in *owned; the block -> unit table is also returned on the host (ufirst).
static int fe_build_copy(hipStream_t s, int num_cus, const int32_t* ptr, int nseg, const int32_t* idx, const float* val, int64_t z,
                         int len, bool cut_sparse, bool compress, FeCopy* out, void** owned, void** owned_c, std::vector<int32_t>* ufirst_host) {
  *owned = nullptr;
  *owned_c = nullptr;
  const int nblock = len > 0 ? (len + FE_B - 1) / FE_B : 1;
  const int chunk = fe_chunk_len(z, nblock, num_cus);
  const size_t zz = (size_t)(z > 0 ? z : 1);
  int wbits = FE_SPAN_BITS;
  if (const char* e = getenv("GDMIX_FE_WINDOW_BITS")) {   // test hook: several windows on a small shard (narrower is always valid)
    const int v = atoi(e);
    if (v >= 1 && v < FE_SPAN_BITS) wbits = v;
  }
  const int nwin = ((nseg > 0 ? nseg - 1 : 0) >> wbits) + 1;   // windows of the gathered vector (fe_ent_kernel)
  if ((int64_t)nblock * nwin > 0x7fffff00ll) { set_error("shard too large for the pass tables"); return GDMIX_RE_ERANGE; }
  const int nbw = nblock * nwin;
  const int extent = nseg < (1 << wbits) ? (nseg > 0 ? nseg : 1) : (1 << wbits);   // gathered elements per window
  const int sparse_chunk = cut_sparse ? (chunk / 8 > 4096 ? chunk / 8 : (chunk < 4096 ? chunk : 4096)) : 0;
  unsigned bits = 1;
  while (bits < 32 && (1u << bits) < (unsigned)nbw) ++bits;
  size_t sort_tmp = 0, scan_tmp = 0, scan2_tmp = 0;
  hipError_t rc = rocprim::radix_sort_pairs(nullptr, sort_tmp, (uint32_t*)nullptr, (uint32_t*)nullptr, (FeEnt*)nullptr, (FeEnt*)nullptr,
                                            zz, 0u, bits, s);
  if (rc == hipSuccess) rc = rocprim::inclusive_scan(nullptr, scan_tmp, (int32_t*)nullptr, (int32_t*)nullptr, zz, rocprim::plus<int32_t>(), s);
  if (rc == hipSuccess) rc = rocprim::exclusive_scan(nullptr, scan2_tmp, (int32_t*)nullptr, (int32_t*)nullptr, 0, (size_t)nbw + 1, rocprim::plus<int32_t>(), s);
  if (rc != hipSuccess) { set_error("rocPRIM sizing failed: %s", hipGetErrorString(rc)); return GDMIX_RE_EHIP; }
  size_t lib = sort_tmp > scan_tmp ? sort_tmp : scan_tmp;
  if (scan2_tmp > lib) lib = scan2_tmp;
  // upper bound of the unit count: a block of len entries has at most len / chunk + 1 units
  const size_t max_units = (size_t)nbw + (size_t)(z / (sparse_chunk > 0 ? sparse_chunk : chunk)) + 1;
  size_t woff = 0;
  auto wtake = [&](size_t bytes) { size_t r = woff; woff = up256(woff + bytes); return r; };
  const size_t w_a = wtake(zz * 4), w_seg = wtake(zz * 4), w_key = wtake(zz * 4), w_ent = wtake(zz * sizeof(FeEnt)), w_ent2 = wtake(zz * sizeof(FeEnt));
  const size_t w_bp = wtake(((size_t)nbw + 1) * 4), w_nch = wtake(((size_t)nbw + 1) * 4), w_uf = wtake(((size_t)nbw + 1) * 4);
  const size_t w_span = wtake(64), w_lib = wtake(lib);
  void* tmp = nullptr;
  rc = hipMalloc(&tmp, woff);
  if (rc != hipSuccess) { set_error("hipMalloc(%zu) failed: %s", woff, hipGetErrorString(rc)); return GDMIX_RE_ENOMEM; }
  char* wb = static_cast<char*>(tmp);
  int32_t* flag = reinterpret_cast<int32_t*>(wb + w_a);      // later: the sorted block numbers
  int32_t* seg = reinterpret_cast<int32_t*>(wb + w_seg);
  uint32_t* skey = reinterpret_cast<uint32_t*>(wb + w_key);
  uint32_t* skey2 = reinterpret_cast<uint32_t*>(wb + w_a);
  FeEnt* ent = reinterpret_cast<FeEnt*>(wb + w_ent);
  FeEnt* ent2 = reinterpret_cast<FeEnt*>(wb + w_ent2);
  int32_t* bp = reinterpret_cast<int32_t*>(wb + w_bp);
  int32_t* nch = reinterpret_cast<int32_t*>(wb + w_nch);
  int32_t* ufw = reinterpret_cast<int32_t*>(wb + w_uf);     // first unit per (block, window)
  int32_t* span = reinterpret_cast<int32_t*>(wb + w_span);
  // the unit tables; the entries follow once their form is known
  size_t toff = 0;
  auto ttake = [&](size_t bytes) { size_t r = toff; toff = up256(toff + bytes); return r; };
  const size_t c_uf = ttake(((size_t)nblock + 1) * 4), c_us = ttake((max_units + 1) * 4), c_ub = ttake(max_units * 4), c_kb = ttake(max_units * 4), c_ord = ttake((max_units + FE_XCDS) * 4);
  const size_t c_ent = ttake(zz * 10 + 512);   // 8 B per entry packed, 4 + 4 + 2 otherwise
  void* mem = nullptr;
  rc = hipMalloc(&mem, toff);
  if (rc != hipSuccess) { set_error("hipMalloc(%zu) failed: %s", toff, hipGetErrorString(rc)); (void)hipFree(tmp); return GDMIX_RE_ENOMEM; }
  char* cb = static_cast<char*>(mem);
  int32_t* ufirst = reinterpret_cast<int32_t*>(cb + c_uf);
  int32_t* ustart = reinterpret_cast<int32_t*>(cb + c_us);
  int32_t* ublock = reinterpret_cast<int32_t*>(cb + c_ub);
  int32_t* kbase = reinterpret_cast<int32_t*>(cb + c_kb);
  const int ge = num_cus * 16;
  int gb = (nbw + 1 + 255) / 256;
  if (gb > 4096) gb = 4096;
  (void)hipMemsetAsync(span, 0, 64, s);
  if (z > 0) {
    (void)hipMemsetAsync(flag, 0, zz * 4, s);
    int gs = (nseg + 255) / 256;
    if (gs > 4096) gs = 4096;
    if (gs < 1) gs = 1;
    hipLaunchKernelGGL(fe_flag_kernel, dim3(gs), dim3(256), 0, s, ptr, nseg, z, flag);
    size_t lt = scan_tmp;
    rc = rocprim::inclusive_scan(wb + w_lib, lt, flag, seg, (size_t)z, rocprim::plus<int32_t>(), s);
    hipLaunchKernelGGL(fe_ent_kernel, dim3(ge), dim3(256), 0, s, seg, idx, val, z, nwin, wbits, skey, ent);
    lt = sort_tmp;
    if (rc == hipSuccess) rc = rocprim::radix_sort_pairs(wb + w_lib, lt, skey, skey2, ent, ent2, (size_t)z, 0u, bits, s);
  }
  hipLaunchKernelGGL(fe_block_kernel, dim3(gb), dim3(256), 0, s, skey2, z, nbw, bp);
  hipLaunchKernelGGL(fe_chunks_kernel, dim3(gb), dim3(256), 0, s, bp, nbw, nwin, chunk, sparse_chunk, extent, nch);
  size_t lt = scan2_tmp;
  if (rc == hipSuccess) rc = rocprim::exclusive_scan(wb + w_lib, lt, nch, ufw, 0, (size_t)nbw + 1, rocprim::plus<int32_t>(), s);
  hipLaunchKernelGGL(fe_units_kernel, dim3(gb), dim3(256), 0, s, bp, ufw, nbw, nwin, chunk, sparse_chunk, extent, z, ustart, ublock, ufirst);
  ufirst_host->resize((size_t)nblock + 1);
  if (rc == hipSuccess) rc = hipMemcpyAsync(ufirst_host->data(), ufirst, ((size_t)nblock + 1) * 4, hipMemcpyDeviceToHost, s);
  if (rc == hipSuccess) rc = hipStreamSynchronize(s);
  const int nunit = rc == hipSuccess ? (*ufirst_host)[(size_t)nblock] : 0;
  int32_t max_span = 0;
  if (rc == hipSuccess) {
    int gu = (nunit + 255) / 256;
    if (gu > 4096) gu = 4096;
    hipLaunchKernelGGL(fe_span_kernel, dim3(gu), dim3(256), 0, s, ent2, ustart, nunit, kbase, span);
    rc = hipMemcpyAsync(&max_span, span, 4, hipMemcpyDeviceToHost, s);
    if (rc == hipSuccess) rc = hipStreamSynchronize(s);
  }
  // Launch order. Workgroups go to the XCDs round robin and every XCD has its own L2: units that gather the same stretch of the
  // vector should meet in one L2 rather than pull it over the fabric eight times (column pass on 4 M samples: 256 MB of
  // residuals on top of 1 GB of entries). Units sorted by first gathered element, the sorted list cut into one run per XCD,
  // run x dealt to the workgroups x, x + XCDS, ...
  std::vector<int32_t> order;
  if (rc == hipSuccess) {
    std::vector<int32_t> kb((size_t)nunit), by((size_t)nunit);
    if (nunit) rc = hipMemcpy(kb.data(), kbase, (size_t)nunit * 4, hipMemcpyDeviceToHost);
    for (int u = 0; u < nunit; ++u) by[(size_t)u] = u;
    std::stable_sort(by.begin(), by.end(), [&](int32_t a, int32_t b) { return kb[(size_t)a] < kb[(size_t)b]; });
    const int per = (nunit + FE_XCDS - 1) / FE_XCDS;
    order.assign((size_t)per * FE_XCDS, -1);
    for (int x = 0; x < FE_XCDS; ++x)
      for (int j = 0; j < per && x * per + j < nunit; ++j) order[(size_t)j * FE_XCDS + x] = by[(size_t)x * per + j];
    if (rc == hipSuccess && !order.empty()) rc = hipMemcpy(cb + c_ord, order.data(), order.size() * 4, hipMemcpyHostToDevice);
  }
  bool packed = GDMIX_FE_PACK && max_span < (1 << (32 - FE_LOC_BITS));
  if (const char* e = getenv("GDMIX_FE_PACK")) packed = packed && atoi(e) != 0;   // test hook: the three-array form on a small shard
  uint2* pent = reinterpret_cast<uint2*>(cb + c_ent);
  int32_t* ckey = reinterpret_cast<int32_t*>(cb + c_ent);
  float* cval = reinterpret_cast<float*>(cb + c_ent + up256(zz * 4));
  uint16_t* cloc = reinterpret_cast<uint16_t*>(cb + c_ent + 2 * up256(zz * 4));
  if (rc == hipSuccess && nunit > 0 && z > 0) {
    if (packed) hipLaunchKernelGGL((fe_pack_kernel<true>), dim3(nunit), dim3(256), 0, s, ent2, ustart, kbase, pent, ckey, cval, cloc);
    else hipLaunchKernelGGL((fe_pack_kernel<false>), dim3(nunit), dim3(256), 0, s, ent2, ustart, kbase, pent, ckey, cval, cloc);
    rc = hipStreamSynchronize(s);
  }
  if (rc == hipSuccess) rc = hipGetLastError();
  out->stream_bytes = (int64_t)z * (packed ? 8 : 10);
  // ---- the 6-byte form for the units it shortens (round 5) ----
  void* cmem = nullptr;
  out->cdata = nullptr; out->cbase = nullptr; out->ctrip = nullptr;
  if (rc == hipSuccess && compress && nunit > 0 && z > 0) {
    int32_t* cnt_dev = nullptr;
    void* cnt_mem = nullptr;
    rc = hipMalloc(&cnt_mem, (size_t)nunit * 4);
    std::vector<int32_t> cnt((size_t)nunit), us((size_t)nunit + 1);
    if (rc == hipSuccess) {
      cnt_dev = static_cast<int32_t*>(cnt_mem);
      hipLaunchKernelGGL(fe_ccount_kernel, dim3(nunit), dim3(256), 0, s, ent2, ustart, kbase, cnt_dev);
      rc = hipMemcpyAsync(cnt.data(), cnt_dev, (size_t)nunit * 4, hipMemcpyDeviceToHost, s);
      if (rc == hipSuccess) rc = hipMemcpyAsync(us.data(), ustart, ((size_t)nunit + 1) * 4, hipMemcpyDeviceToHost, s);
      if (rc == hipSuccess) rc = hipStreamSynchronize(s);
    }
    if (cnt_mem) (void)hipFree(cnt_mem);
    std::vector<int64_t> cb((size_t)nunit, -1);
    std::vector<int32_t> ct((size_t)nunit, 0);
    size_t total = 0;
    int taken = 0;
    for (int u = 0; rc == hipSuccess && u < nunit; ++u) {
      const int64_t nu = (int64_t)us[(size_t)u + 1] - us[(size_t)u];
      const int64_t trips = ((int64_t)cnt[(size_t)u] + FE_CTRIP - 1) / FE_CTRIP;
      // the form pays when it is shorter than 8 bytes per entry with room to spare (fillers of sparse blocks, padding of short units)
      if (nu > 0 && trips * FE_CTRIP_BYTES * 10 <= nu * 8 * 9) { cb[(size_t)u] = (int64_t)total; ct[(size_t)u] = (int32_t)trips; total += (size_t)trips * FE_CTRIP_BYTES; ++taken; }
    }
    if (rc == hipSuccess && taken > 0) {
      const size_t o_cb = 0, o_ct = up256((size_t)nunit * 8), o_data = o_ct + up256((size_t)nunit * 4);
      rc = hipMalloc(&cmem, o_data + total + 256);
      if (rc == hipSuccess) {
        char* cm = static_cast<char*>(cmem);
        rc = hipMemsetAsync(cm + o_data, 0, total, s);
        if (rc == hipSuccess) rc = hipMemcpyAsync(cm + o_cb, cb.data(), (size_t)nunit * 8, hipMemcpyHostToDevice, s);
        if (rc == hipSuccess) rc = hipMemcpyAsync(cm + o_ct, ct.data(), (size_t)nunit * 4, hipMemcpyHostToDevice, s);
        if (rc == hipSuccess) {
          hipLaunchKernelGGL(fe_cpack_kernel, dim3(nunit), dim3(256), 0, s, ent2, ustart, kbase, reinterpret_cast<const int64_t*>(cm + o_cb),
                             reinterpret_cast<unsigned char*>(cm + o_data));
          rc = hipStreamSynchronize(s);      // (cb / ct go out of scope; tmp is freed below)
        }
        if (rc == hipSuccess) {
          int64_t plain = 0;
          for (int u = 0; u < nunit; ++u) if (cb[(size_t)u] < 0) plain += (int64_t)us[(size_t)u + 1] - us[(size_t)u];
          out->stream_bytes = (int64_t)total + plain * (packed ? 8 : 10);
          out->cdata = reinterpret_cast<const unsigned char*>(cm + o_data);
          out->cbase = reinterpret_cast<const int64_t*>(cm + o_cb);
          out->ctrip = reinterpret_cast<const int32_t*>(cm + o_ct);
        }
      }
    }
  }
  *owned_c = cmem;
  (void)hipFree(tmp);
  if (rc != hipSuccess) { set_error("building a pass's copy failed: %s", hipGetErrorString(rc)); (void)hipFree(mem); if (cmem) (void)hipFree(cmem); *owned_c = nullptr; return GDMIX_RE_EHIP; }
  out->ent = packed ? pent : nullptr;
  out->key = ckey; out->val = cval; out->loc = cloc; out->kbase = kbase; out->ustart = ustart; out->ublock = ublock; out->ufirst = ufirst;
  out->part = nullptr;
  out->nblock = nblock;
  out->nunit = nunit;
  out->order = reinterpret_cast<const int32_t*>(cb + c_ord);
  out->nlaunch = (int)order.size();
  *owned = mem;
  return GDMIX_RE_OK;
}

static void fe_free(gdmix_fe_problem* p) {
  for (auto& e : p->ev) if (e) (void)hipEventDestroy(e);
  for (auto& e : p->ring_ev) if (e) (void)hipEventDestroy(e);
  if (p->status_ring) (void)hipHostFree(p->status_ring);
  if (p->pool) (void)hipFree(p->pool);
  for (auto& c : p->copies) if (c) (void)hipFree(c);
  for (auto& c : p->ccopies) if (c) (void)hipFree(c);
  if (p->hot_mem) (void)hipFree(p->hot_mem);
  delete p;
}

// The column pass's copy, frequent columns under their virtual numbers (FeHot above). Entry counts per column come from the
// packed shard's column pointers (one read-back at creation).
static int fe_split_hot(gdmix_fe_problem* p, const gdmix_re_packed* b, hipStream_t s) {
  FeDev& F = p->F;
  gdmix_ctx_impl* ci = &p->ctx->impl;
  F.hot = FeHot{0, 0, nullptr};
  long hot_min = 1 << 16;
  if (const char* e = getenv("GDMIX_FE_HOT_MIN")) hot_min = atol(e);   // test hook (0 = no frequent columns)
  std::vector<int32_t> cp((size_t)F.d + 1), hot_cols;
  if (hot_min > 0 && F.d > 0 && F.z > 0) {
    HIP_TRY(hipMemcpyAsync(cp.data(), b->col_ptr, ((size_t)F.d + 1) * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    std::vector<std::pair<int32_t, int32_t>> cand;   // (-count, column): most frequent first, ties by column
    for (int c = 0; c < F.d; ++c)
      if (cp[(size_t)c + 1] - cp[(size_t)c] >= hot_min) cand.emplace_back(-(cp[(size_t)c + 1] - cp[(size_t)c]), c);
    std::sort(cand.begin(), cand.end());
    if (cand.size() > (size_t)FE_HOT_MAX) cand.resize(FE_HOT_MAX);
    for (auto& q : cand) hot_cols.push_back(q.second);
    std::sort(hot_cols.begin(), hot_cols.end());
  }
  if (hot_cols.empty())
    return fe_build_copy(s, ci->num_cus, b->row_ptr, F.n, b->csr_col, b->csr_val, F.z, F.d, true, (p->compress & 2) != 0, &F.cc, &p->copies[1],
                         &p->ccopies[1], &p->uf_c);
  const int nh = (int)hot_cols.size();
  const int vbase = (F.d + FE_B - 1) / FE_B * FE_B;
  std::vector<int32_t> hotmap((size_t)F.d, -1);
  for (int h = 0; h < nh; ++h) hotmap[(size_t)hot_cols[(size_t)h]] = h;
  const size_t zz = (size_t)F.z;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t r = off; off = up256(off + bytes); return r; };
  const size_t o_map = take((size_t)F.d * 4), o_col2 = take(zz * 4);
  void* mem = nullptr;
  hipError_t rc = hipMalloc(&mem, off);
  if (rc != hipSuccess) { set_error("hipMalloc(%zu) failed: %s", off, hipGetErrorString(rc)); return GDMIX_RE_ENOMEM; }
  void* small = nullptr;
  rc = hipMalloc(&small, (size_t)nh * 4);
  if (rc != hipSuccess) { (void)hipFree(mem); set_error("hipMalloc failed: %s", hipGetErrorString(rc)); return GDMIX_RE_ENOMEM; }
  p->hot_mem = small;                                  // freed with the problem
  char* m = static_cast<char*>(mem);
  int32_t* col2 = reinterpret_cast<int32_t*>(m + o_col2);
  rc = hipMemcpyAsync(m + o_map, hotmap.data(), (size_t)F.d * 4, hipMemcpyHostToDevice, s);
  if (rc == hipSuccess) rc = hipMemcpyAsync(small, hot_cols.data(), (size_t)nh * 4, hipMemcpyHostToDevice, s);
  if (rc == hipSuccess) {
    int g = (F.n + 255) / 256;
    if (g > ci->num_cus * 32) g = ci->num_cus * 32;
    hipLaunchKernelGGL(fe_hot_remap_kernel, dim3(g), dim3(256), 0, s, b->row_ptr, F.n, b->csr_col, reinterpret_cast<const int32_t*>(m + o_map), vbase, col2);
    rc = hipStreamSynchronize(s);                      // (also: the host vectors above are done with)
  }
  if (rc != hipSuccess) { (void)hipFree(mem); set_error("frequent-column tables: %s", hipGetErrorString(rc)); return GDMIX_RE_EHIP; }
  const int rc2 = fe_build_copy(s, ci->num_cus, b->row_ptr, F.n, col2, b->csr_val, F.z, vbase + nh * FE_HOT_REP, true, (p->compress & 2) != 0, &F.cc,
                                &p->copies[1], &p->ccopies[1], &p->uf_c);
  (void)hipFree(mem);
  if (rc2 != GDMIX_RE_OK) return rc2;
  F.hot.n = nh;
  F.hot.vbase = vbase;
  F.hot.col = static_cast<const int32_t*>(small);
  return GDMIX_RE_OK;
}

template <bool HESS>
static int fe_passes(gdmix_fe_problem* p, const FeDev& F, hipStream_t s, bool timed) {
  // features absent from this shard must read 0 in the reduce buffer; fe_dots_kernel leaves it cleared behind a step
  if (p->dirty) HIP_TRY(hipMemsetAsync(F.fg, 0, ((size_t)F.P + 1) * 8, s));
  p->dirty = true;
  int gd = (F.d + 255) / 256;
  if (gd > 2048) gd = 2048;
  if (gd < 1) gd = 1;
  // xl (x in the shard's local order) is kept current by the step's update (fe_update_one); the Hessian passes may run at
  // another point: they gather it themselves and put the solver's back afterwards
  if (HESS) hipLaunchKernelGGL(fe_prepare_kernel, dim3(gd), dim3(256), 0, s, F);
  if (timed) HIP_TRY(hipEventRecord(p->ev[0], s));
  if (F.rc.ent) hipLaunchKernelGGL((fe_scatter_kernel<true, HESS, true>), dim3(F.rc.nlaunch), dim3(WAVE), 0, s, F, p->o);
  else hipLaunchKernelGGL((fe_scatter_kernel<true, HESS, false>), dim3(F.rc.nlaunch), dim3(WAVE), 0, s, F, p->o);
  if (F.nmulti) hipLaunchKernelGGL((fe_rows_fix_kernel<HESS>), dim3(F.nmulti * FE_FIX_PER_BLOCK), dim3(FE_THREADS), 0, s, F, p->o);
  if (timed) HIP_TRY(hipEventRecord(p->ev[1], s));
  if (F.cc.ent) hipLaunchKernelGGL((fe_scatter_kernel<false, HESS, true>), dim3(F.cc.nlaunch), dim3(WAVE), 0, s, F, p->o);
  else hipLaunchKernelGGL((fe_scatter_kernel<false, HESS, false>), dim3(F.cc.nlaunch), dim3(WAVE), 0, s, F, p->o);
  if (timed) HIP_TRY(hipEventRecord(p->ev[2], s));
  int gf = (F.d + FE_RED_OUT - 1) / FE_RED_OUT;
  hipLaunchKernelGGL(fe_finish_kernel<HESS>, dim3((gf < FE_FIN_BLOCKS ? FE_FIN_BLOCKS : gf) + F.hot.n), dim3(FE_THREADS), 0, s, F);
  if (HESS) hipLaunchKernelGGL(fe_prepare_kernel, dim3(gd), dim3(256), 0, s, p->F);
  HIP_TRY(hipGetLastError());
  return GDMIX_RE_OK;
}

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
  HIP_TRY(join_unique(ci, s));   // the shard's unique_global is read here
  gdmix_fe_problem* p = new (std::nothrow) gdmix_fe_problem();
  if (!p) { set_error("out of host memory"); return GDMIX_RE_ENOMEM; }
  p->ctx = ctx;
  p->pool = nullptr;
  p->copies[0] = p->copies[1] = nullptr;
  p->ccopies[0] = p->ccopies[1] = nullptr;
  p->compress = FE_COMPRESS_DEFAULT;
  if (const char* e = getenv("GDMIX_FE_COMPRESS")) p->compress = atoi(e) & 3;
  p->hot_mem = nullptr;
  p->timed = false;
  p->dirty = false;      // the pool is zeroed at creation
  for (auto& e : p->ev) e = nullptr;
  for (auto& e : p->ring_ev) e = nullptr;
  p->status_ring = nullptr;
  p->gen = 0u;
  p->seq = 0;
  p->evals = 0;
  {
    // the one-launch step has workgroups waiting for the last arriver: next to ANOTHER process's persistent grid neither might get
    // all its workgroups placed (re_internal.hpp: why this kernel is not behind the inter-process lock). A device that another
    // process is present on gets the three-launch step; GDMIX_FE_FUSED_TAIL=0 / 1 decides whatever the device looks like.
    const char* e = getenv("GDMIX_FE_FUSED_TAIL");
    p->fused_tail = e ? e[0] != '0' : !device_has_another_process(ci->device);
  }
  FeDev& F = p->F;
  const int ic = opts->has_intercept ? 1 : 0;
  F.n = (int)b->N; F.z = b->Z; F.d = (int)b->D; F.ic = ic; F.D = num_features; F.P = (int)num_features + ic; F.m = opts->m;
  F.y = b->y; F.o = b->offset; F.w = b->weight; F.umap = b->unique_global;
  SolveParams& o = p->o;
  o.l2 = opts->l2; o.ftol = opts->ftol; o.pgtol = opts->pgtol; o.threshold = 0.0; o.regularize_bias = opts->regularize_bias;
  o.has_intercept = ic; o.m = opts->m; o.max_iter = opts->max_iter; o.maxfun = opts->maxfun; o.maxls = opts->maxls;
  o.variance_mode = 0; o.sum_loss = 1; o.linear = opts->linear ? 1 : 0;
  // row pass: outputs = rows, gathered = x by local column: from the column-major arrays. Column pass: the other way round.
  std::vector<int32_t> uf_r, uf_c;
  int rc2 = fe_build_copy(s, ci->num_cus, b->col_ptr, F.d, b->csc_row, b->csc_val, F.z, F.n, false, (p->compress & 1) != 0, &F.rc, &p->copies[0],
                          &p->ccopies[0], &uf_r);
  if (rc2 == GDMIX_RE_OK) rc2 = fe_split_hot(p, b, s);    // the frequent columns, and the column pass's copy of the others
  if (rc2 != GDMIX_RE_OK) { fe_free(p); return rc2; }
  uf_c = p->uf_c;
  std::vector<int32_t> multi;
  for (int rb = 0; rb < F.rc.nblock; ++rb) if (uf_r[(size_t)rb + 1] - uf_r[(size_t)rb] > 1) multi.push_back(rb);
  F.nmulti = (int)multi.size();
  F.nred = F.rc.nunit + F.nmulti * FE_FIX_PER_BLOCK;
  const size_t P = (size_t)F.P;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t r = off; off = up256(off + bytes); return r; };
  const size_t o_xl = take((size_t)(F.d + 1) * 8), o_rs = take((size_t)(F.n + 1) * 8), o_fg = take((P + 1) * 8);
  const size_t o_pr = take((size_t)F.rc.nunit * FE_B * 8), o_pc = take((size_t)F.cc.nunit * FE_B * 8);
  const size_t o_multi = take((multi.size() + 1) * 4), o_red = take((size_t)F.nred * 3 * 8 + 16);
  const size_t o_acc = take((size_t)FE_DOT_BLOCKS * COMPACT_KD * 8), o_fin = take((size_t)FE_FIN_BLOCKS * 3 * 8 + 64);
  const size_t o_state = take(sizeof(CompactState)), o_plan = take(sizeof(CompactPlan)), o_mats = take(sizeof(CompactMats));
  const size_t o_vec = take(((size_t)5 * P + compact_hist_doubles((int64_t)P, opts->m)) * 8 + 16), o_status = take(64);
  const size_t o_inv = take(P * 4), o_sync = take(sizeof(FeSync));
  hipError_t rc = hipMalloc(&p->pool, off);
  if (rc != hipSuccess) { set_error("hipMalloc(%zu) failed: %s", off, hipGetErrorString(rc)); fe_free(p); return GDMIX_RE_ENOMEM; }
  char* base = static_cast<char*>(p->pool);
  rc = hipMemsetAsync(base, 0, off, s);
  if (rc == hipSuccess) rc = hipMemsetAsync(base + o_inv, 0xff, P * 4, s);   // -1: the coefficient is not a column of this shard
  if (rc == hipSuccess) rc = hipHostMalloc(reinterpret_cast<void**>(&p->status_ring), FE_RING * sizeof(int32_t), hipHostMallocDefault);
  for (auto& e : p->ring_ev) if (rc == hipSuccess) rc = hipEventCreateWithFlags(&e, hipEventDisableTiming);
  if (rc == hipSuccess && !multi.empty()) {
    rc = hipMemcpyAsync(base + o_multi, multi.data(), multi.size() * 4, hipMemcpyHostToDevice, s);
    if (rc == hipSuccess) rc = hipStreamSynchronize(s);   // `multi` goes out of scope
  }
  if (rc != hipSuccess) { set_error("initialising the problem failed: %s", hipGetErrorString(rc)); fe_free(p); return GDMIX_RE_EHIP; }
  F.xl = reinterpret_cast<double*>(base + o_xl); F.rs = reinterpret_cast<double*>(base + o_rs);
  F.fg = reinterpret_cast<double*>(base + o_fg);
  F.rc.part = reinterpret_cast<double*>(base + o_pr); F.cc.part = reinterpret_cast<double*>(base + o_pc);
  F.multi = reinterpret_cast<const int32_t*>(base + o_multi);
  F.loss_part = reinterpret_cast<double*>(base + o_red); F.rsum_part = F.loss_part + F.nred; F.loss_lo_part = F.rsum_part + F.nred;
  F.acc_part = reinterpret_cast<double*>(base + o_acc);
  F.fin_part = reinterpret_cast<double*>(base + o_fin);
  F.fin_count = reinterpret_cast<unsigned*>(base + o_fin + (size_t)FE_FIN_BLOCKS * 3 * 8);
  F.state = reinterpret_cast<CompactState*>(base + o_state);
  F.plan = reinterpret_cast<CompactPlan*>(base + o_plan);
  F.mats = reinterpret_cast<CompactMats*>(base + o_mats);
  double* v = reinterpret_cast<double*>(base + o_vec);
  F.W.x = v; F.W.g = v + P; F.W.d = v + 2 * P; F.W.t = v + 3 * P; F.W.r = v + 4 * P;
  F.W.ws = v + 5 * P + ((5 * P) & 1);   // 16-byte aligned: the interleaved history (re_lbfgs_compact.hpp) is read with 16-byte loads
  F.W.wy = F.W.ws + (size_t)opts->m * P;
  F.W.rs = F.rs; F.W.alpha = nullptr; F.W.rho = nullptr; F.W.part = nullptr;
  p->status_dev = reinterpret_cast<int32_t*>(base + o_status);
  F.inv = reinterpret_cast<int32_t*>(base + o_inv);
  F.sync = reinterpret_cast<FeSync*>(base + o_sync);
  int gp = (int)((P + 255) / 256);
  if (gp > 1024) gp = 1024;
  hipLaunchKernelGGL(fe_init_kernel, dim3(gp), dim3(256), 0, s, F, theta0);
  rc = hipGetLastError();
  if (rc != hipSuccess) { set_error("launch failed: %s", hipGetErrorString(rc)); fe_free(p); return GDMIX_RE_EHIP; }
  *out = p;
  return GDMIX_RE_OK;
}

GDMIX_API void gdmix_fe_destroy(gdmix_fe_problem* p) {
  if (p) fe_free(p);
}

GDMIX_API double* gdmix_fe_reduce_buffer(gdmix_fe_problem* p, int64_t* count) {
  if (!p) return nullptr;
  if (count) *count = (int64_t)p->F.P + 1;
  return p->F.fg;
}

GDMIX_API int gdmix_fe_eval(gdmix_fe_problem* p, void* stream) {
  if (!p) { set_error("problem is NULL"); return GDMIX_RE_EINVAL; }
  if (!p->ev[0]) for (auto& e : p->ev) HIP_TRY(hipEventCreate(&e));
  // the first two evaluations of a problem are timed (gdmix_fe_last_eval_ms reports the second): with the status read a few steps
  // late the LAST evaluations of a solve are no-ops
  const bool timed = p->evals < 2;
  const int rc = fe_passes<false>(p, p->F, static_cast<hipStream_t>(stream), timed);
  if (rc == GDMIX_RE_OK) { if (timed) p->timed = true; ++p->evals; }
  return rc;
}

GDMIX_API int gdmix_fe_hessian_diag(gdmix_fe_problem* p, const double* theta, void* stream) {
  if (!p) { set_error("problem is NULL"); return GDMIX_RE_EINVAL; }
  FeDev F = p->F;
  if (theta) F.W.x = const_cast<double*>(theta);   // the passes only read x
  p->dirty = true;   // (a step enqueued behind the stop leaves the buffer as it was)
  return fe_passes<true>(p, F, static_cast<hipStream_t>(stream), false);
}

GDMIX_API size_t gdmix_fe_hessian_dense_scratch_bytes(const gdmix_re_packed* shard) {
  return shard ? hessian_dense_scratch_doubles(shard->N) * 8 : 0;
}

GDMIX_API int gdmix_fe_hessian_dense(gdmix_re_ctx* ctx, const gdmix_re_packed* b, int has_intercept, const double* theta_local, double* H,
                                     int64_t ld, void* scratch, size_t scratch_bytes, void* stream) {
  if (!ctx || !b || !theta_local || !H || !scratch) { set_error("NULL argument"); return GDMIX_RE_EINVAL; }
  if (b->E != 1) { set_error("gdmix_fe_hessian_dense takes a one-entity batch (a worker's shard)"); return GDMIX_RE_EINVAL; }
  const int ic = has_intercept ? 1 : 0;
  if (b->D + ic > VAR_FULL_BIG_MAX_P) { set_error("dense Hessian of %lld coefficients: at most %lld", (long long)(b->D + ic), (long long)VAR_FULL_BIG_MAX_P); return GDMIX_RE_ERANGE; }
  if (ld < b->D + ic || ld % 64) { set_error("ld must be d + has_intercept rounded up to a multiple of 64"); return GDMIX_RE_EINVAL; }
  if (scratch_bytes < hessian_dense_scratch_doubles(b->N) * 8) { set_error("scratch too small"); return GDMIX_RE_ENOMEM; }
  HIP_TRY(hipSetDevice(ctx->impl.device));
  BatchDev B;
  B.ent_row_ptr = b->ent_row_ptr; B.ent_nnz_ptr = b->ent_nnz_ptr; B.ent_feat_ptr = b->ent_feat_ptr; B.row_ptr = b->row_ptr; B.csr_col = b->csr_col;
  B.csr_val = b->csr_val; B.col_ptr = b->col_ptr; B.csc_row = b->csc_row; B.csc_val = b->csc_val; B.y = b->y; B.offset = b->offset; B.weight = b->weight;
  B.order = b->order;
  HIP_TRY(launch_hessian_dense(&ctx->impl, B, b->N, b->D, ic, theta_local, H, ld, static_cast<double*>(scratch), static_cast<hipStream_t>(stream)));
  return GDMIX_RE_OK;
}

GDMIX_API int gdmix_fe_variance_of_hessian(gdmix_re_ctx* ctx, double* H, int64_t p, int64_t ld, double l2, int64_t unregularised_index,
                                           double* work, double* variance, void* stream) {
  if (!ctx || !H || !work || !variance) { set_error("NULL argument"); return GDMIX_RE_EINVAL; }
  if (p < 1 || p > VAR_FULL_BIG_MAX_P || ld < p || ld % 64) { set_error("bad matrix dimensions (p = %lld, ld = %lld)", (long long)p, (long long)ld); return GDMIX_RE_EINVAL; }
  HIP_TRY(hipSetDevice(ctx->impl.device));
  HIP_TRY(launch_variance_of_hessian(&ctx->impl, H, work, p, ld, l2, unregularised_index, variance, static_cast<hipStream_t>(stream)));
  return GDMIX_RE_OK;
}

static int fe_enqueue_step(gdmix_fe_problem* p, hipStream_t s) {
  const FeDev& F = p->F;
  int gp = (F.P + 255) / 256;
  const int dot_blocks = gp < FE_DOT_BLOCKS ? gp : FE_DOT_BLOCKS;
  if (p->fused_tail) {
    int g = p->ctx->impl.num_cus;          // at most one workgroup per CU: all resident, the wait inside cannot starve one
    if (g > dot_blocks) g = dot_blocks;
    if (g < 1) g = 1;
    ++p->gen;
    if (p->gen == 0u) ++p->gen;
    hipLaunchKernelGGL(fe_tail_kernel, dim3(g), dim3(FE_THREADS), 0, s, F, p->o, dot_blocks, p->status_dev, p->gen);
  } else {
    hipLaunchKernelGGL(fe_dots_kernel, dim3(dot_blocks), dim3(FE_THREADS), 0, s, F, p->o);
    hipLaunchKernelGGL(fe_step_kernel, dim3(1), dim3(FE_THREADS), 0, s, F, p->o, dot_blocks, p->status_dev);
    if (gp > 1024) gp = 1024;
    hipLaunchKernelGGL(fe_update_kernel, dim3(gp), dim3(256), 0, s, F, p->o.m);
  }
  p->dirty = false;
  HIP_TRY(hipGetLastError());
  const int slot = (int)(p->seq % FE_RING);
  HIP_TRY(hipMemcpyAsync(p->status_ring + slot, p->status_dev, sizeof(int32_t), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipEventRecord(p->ring_ev[slot], s));
  ++p->seq;
  return GDMIX_RE_OK;
}

GDMIX_API int gdmix_fe_step(gdmix_fe_problem* p, void* stream, int32_t* status) {
  if (!p || !status) { set_error("NULL argument"); return GDMIX_RE_EINVAL; }
  const int rc = fe_enqueue_step(p, static_cast<hipStream_t>(stream));
  if (rc != GDMIX_RE_OK) return rc;
  return gdmix_fe_step_status(p, p->seq - 1, status);
}

GDMIX_API int gdmix_fe_step_async(gdmix_fe_problem* p, void* stream, int64_t* seq) {
  if (!p) { set_error("problem is NULL"); return GDMIX_RE_EINVAL; }
  const int rc = fe_enqueue_step(p, static_cast<hipStream_t>(stream));
  if (rc == GDMIX_RE_OK && seq) *seq = p->seq - 1;
  return rc;
}

GDMIX_API int gdmix_fe_step_status(gdmix_fe_problem* p, int64_t seq, int32_t* status) {
  if (!p || !status) { set_error("NULL argument"); return GDMIX_RE_EINVAL; }
  if (seq < 0 || seq >= p->seq || seq + FE_RING <= p->seq) {
    set_error("step %lld: only the last %d of the %lld enqueued steps can be asked for", (long long)seq, FE_RING, (long long)p->seq);
    return GDMIX_RE_EINVAL;
  }
  const int slot = (int)(seq % FE_RING);
  HIP_TRY(hipEventSynchronize(p->ring_ev[slot]));
  *status = p->status_ring[slot];
  return GDMIX_RE_OK;
}

GDMIX_API int gdmix_fe_solve(gdmix_fe_problem* p, void* stream, int32_t lookahead, int64_t max_evals, int32_t* status, int64_t* evals) {
  if (!p || !status) { set_error("NULL argument"); return GDMIX_RE_EINVAL; }
  if (lookahead < 0 || lookahead >= FE_RING) { set_error("0 <= lookahead < %d", FE_RING); return GDMIX_RE_EINVAL; }
  int32_t st = -1;
  int64_t k = 0;
  for (; k < max_evals + lookahead; ++k) {
    int rc = gdmix_fe_eval(p, stream);
    if (rc == GDMIX_RE_OK) rc = fe_enqueue_step(p, static_cast<hipStream_t>(stream));
    if (rc != GDMIX_RE_OK) return rc;
    if (k >= lookahead) {
      rc = gdmix_fe_step_status(p, p->seq - 1 - lookahead, &st);
      if (rc != GDMIX_RE_OK) return rc;
      if (st >= 0) break;
    }
  }
  // the steps behind the one that stopped were no-ops and report the same status; wait for them so that nothing of this solve is
  // still in flight when the caller reads the result
  if (st >= 0) {
    const int rc = gdmix_fe_step_status(p, p->seq - 1, &st);
    if (rc != GDMIX_RE_OK) return rc;
  }
  *status = st;
  if (evals) *evals = k + 1;
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

GDMIX_API int gdmix_fe_stream_bytes(gdmix_fe_problem* p, int64_t* rows_pass, int64_t* cols_pass) {
  if (!p) { set_error("problem is NULL"); return GDMIX_RE_EINVAL; }
  if (rows_pass) *rows_pass = p->F.rc.stream_bytes;
  if (cols_pass) *cols_pass = p->F.cc.stream_bytes;
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
