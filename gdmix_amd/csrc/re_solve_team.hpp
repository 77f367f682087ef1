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
#include "re_lbfgs_compact.hpp"

namespace gdmix {

// -DGDMIX_TEAM_PROFILE: thread 0 of a team's first workgroup accumulates wall-clock ticks (100 MHz) per phase
// and prints them per entity. Exploration only.
#ifdef GDMIX_TEAM_PROFILE
#define TEAM_PROF_DECL unsigned long long prof_t[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long prof_last = wall_clock64();
#define TEAM_PROF(i) do { const unsigned long long _n = wall_clock64(); prof_t[i] += _n - prof_last; prof_last = _n; } while (0)
#else
#define TEAM_PROF_DECL
#define TEAM_PROF(i) do { } while (0)
#endif

// GDMIX_TEAM_DIRECT_AB=1 (round 6, the default): the team kernels take S'g and Y'g directly too (re_lbfgs_compact.hpp: why) — 20 more
// accumulators in registers and a reduction of 47 values instead of 27. With two tiles per trip at every history length that
// spilled 152 registers and cost 5 % on a Zipf partition; with one tile per trip from GDMIX_TEAM_DIRECT_ONE_TILE_FROM + 1 pairs on
// it spills fewer than before (17) and costs 1.7 - 1.9 % (C5 share 700 -> 714 ms, profiles/r06_fuzz.txt). =0: the sums of round 3.
#ifndef GDMIX_TEAM_DIRECT_AB
#define GDMIX_TEAM_DIRECT_AB 1
#endif
#ifndef GDMIX_TEAM_DIRECT_ONE_TILE_FROM
#define GDMIX_TEAM_DIRECT_ONE_TILE_FROM 5
#endif
constexpr int TEAM_KX = GDMIX_TEAM_DIRECT_AB ? COMPACT_KD : TEAM_K;   // the team's accumulators: [0, TEAM_K - 1) as acc[], then S'g, Y'g (direct only), max|g| LAST
constexpr int TEAM_VEC = GDMIX_TEAM_DIRECT_AB ? 48 : 32;   // doubles per workgroup slot of the device-wide exchange buffer
constexpr int TEAM_MAX_BLOCKS = 256;           // workgroups per team
constexpr int TEAM_MAX_TEAMS = 256;
constexpr int TEAM_SHORT_COL = 16;            // tiles whose columns are all this short: one lane per column
constexpr int TEAM_CHUNK = 512;               // CSC entries a wavefront stages through LDS at a time
constexpr int TEAM_LONGC = 2048;              // columns at least this long are split over the team in 64 slices
constexpr int TEAM_LONG_CAP = 256;            // such columns per entity (more: none is split)

// Exchange buffer of a multi-workgroup team (HBM).
struct TeamSync {
  unsigned count;    // barrier arrivals, monotonic inside a launch (zeroed by a memset node before it)
  unsigned abort;    // a workgroup gave up waiting (never expected; keeps a lost workgroup from hanging the GPU)
  unsigned next;     // team 0's copy: next entity of the class to hand out
  int cur;           // entity this team is working on
  unsigned moved[2]; // by parity of the update: some coefficient of the trial differs from the previously evaluated point
  unsigned pad[10];
  double vec[2][TEAM_VEC][TEAM_MAX_BLOCKS];   // [phase][value][workgroup]
  unsigned xcc[TEAM_MAX_BLOCKS];              // HW_REG_XCC_ID of every workgroup of the team, published once per launch (1 + id)
};

// LDS of one workgroup of a team.
template <int NW>
struct TeamLds {
  double red[2][NW][TEAM_VEC];
  double out[2][TEAM_VEC];
  CompactMats mats;
  double buf[NW][TEAM_CHUNK];     // per-wavefront staging of val * r products
  int llist[TEAM_LONG_CAP];       // long columns of the entity, ascending
  int lraw[TEAM_LONG_CAP];
  int n_long, n_long_raw;
  unsigned moved[2];              // one-workgroup teams: the flag of TeamSync::moved
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
  bool one_xcd;           // every workgroup of the team reported the same XCD (measured at the start of the launch, team_placement)

  __device__ __forceinline__ void block_sync() const { __syncthreads(); }

  // Device-wide barrier with release/acquire of everything written before it: the XCDs' L2s are not
  // coherent with each other and a CU's L1 is not refreshed by other CUs' stores, so one lane per workgroup
  // writes the XCD's dirty L2 lines back before arriving and invalidates this CU's L1 after the wait.
  __device__ __forceinline__ void device_barrier() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    ++epoch;
    if (threadIdx.x == 0) {
      // The release writes the XCD's dirty L2 lines back so that another XCD can see them (1.7 - 6.5 us, MI355X_MICROARCH.md).
      // A team whose workgroups all sit on ONE XCD shares one L2: its stores are there once acknowledged (the vmcnt(0) above;
      // the vector L1 writes through), and the acquire below drops this CU's stale L1 lines — no write-back needed. Which case
      // applies is measured, not assumed (team_placement): a different placement changes the speed, never the result.
      if (!one_xcd) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
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
  // Once per launch (teams are fixed for the launch): every workgroup publishes the XCD it runs on, one full barrier, then
  // everybody reads the team's list. HIP promises nothing about placement (block b is observed on XCD b % 8, so teams of a
  // launch with a multiple of 8 teams sit on one XCD each): the fast barrier is used only where the measurement says so.
  __device__ __forceinline__ void team_placement() {
    one_xcd = false;
    if (nblocks <= 1) return;
    const unsigned mine = 1u + ((unsigned)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15u);   // HW_REG_XCC_ID, bits [3:0]
    if (threadIdx.x == 0) __hip_atomic_store(&gs->xcc[bid], mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    device_barrier();
    bool same = true;
    for (unsigned b = threadIdx.x; b < nblocks; b += blockDim.x)
      same = same && __hip_atomic_load(&gs->xcc[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == mine;
    one_xcd = __syncthreads_and(same ? 1 : 0) != 0;
  }
  // "The trial point just formed differs from the previously evaluated point": set by any wavefront that moved a coefficient
  // during update number u (slot u & 1), read by everybody after the evaluation of that trial, cleared for its next use at the
  // start of that evaluation (behind the barrier that follows the update; the slot's previous value was read an evaluation ago).
  __device__ __forceinline__ unsigned* moved_slot(unsigned u) const { return nblocks > 1 ? &gs->moved[u & 1u] : &L->moved[u & 1u]; }
  __device__ __forceinline__ void moved_set(unsigned u, bool mine) {
    if (__ballot(mine) != 0ull && lane == 0) {
      if (nblocks > 1) __hip_atomic_store(&gs->moved[u & 1u], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else L->moved[u & 1u] = 1u;
    }
  }
  __device__ __forceinline__ bool moved_get(unsigned u) const {
    return (nblocks > 1 ? __hip_atomic_load(&gs->moved[u & 1u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : L->moved[u & 1u]) != 0u;
  }
  __device__ __forceinline__ void moved_clear(unsigned u) {
    if (threadIdx.x == 0) {
      if (nblocks > 1) { if (bid == 0) __hip_atomic_store(&gs->moved[u & 1u], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
      else L->moved[u & 1u] = 0u;
    }
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
    double mine = 0.0;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const double t = (k == K - 1) ? wave_max_nonneg(v[k]) : wave_sum(v[k]);
      if (lane == k) mine = t;
    }
    const double* tot = reduce_placed<K>(mine);
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = tot[k];
  }

  // The same with the wavefront's totals already in place: lane k of every wavefront holds that wavefront's total of value
  // k. Returns the team totals (LDS, valid until the reduction after the next one).
  template <int K>
  __device__ __forceinline__ const double* reduce_placed(double mine) {
    static_assert(K <= TEAM_VEC, "reduction wider than the exchange slot");
    const int ph = phase;
    phase ^= 1;
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
      if (nblocks > 1) {
        gs->vec[ph][threadIdx.x][bid] = s;
      } else {
        L->out[ph][threadIdx.x] = s;
      }
    }
    if (nblocks > 1) {
      device_barrier();
      // wavefront w totals values w, w + NW, ...: every partial it needs is requested before the first is used
      constexpr int KPW = (K + NW - 1) / NW;
      constexpr int BPL = TEAM_MAX_BLOCKS / WAVE;
      double t[KPW][BPL];
#pragma unroll
      for (int i = 0; i < KPW; ++i) {
        const int k = w + i * NW;
#pragma unroll
        for (int j = 0; j < BPL; ++j) {
          const unsigned b = lane + j * WAVE;
          t[i][j] = (k < K && b < nblocks) ? gs->vec[ph][k][b] : 0.0;
        }
      }
#pragma unroll
      for (int i = 0; i < KPW; ++i) {
        const int k = w + i * NW;
        double s = t[i][0];
#pragma unroll
        for (int j = 1; j < BPL; ++j) s = (k == K - 1) ? fmax(s, t[i][j]) : s + t[i][j];
        s = (k == K - 1) ? wave_max_nonneg(s) : wave_sum(s);
        if (k < K && lane == 0) L->out[ph][k] = s;
      }
    }
    __syncthreads();
    return L->out[ph];
  }
};

__device__ __forceinline__ int readlane_i(int v, int l) { return __builtin_amdgcn_readlane(v, l); }

// acc + sum_{k in [k0,k1)} val[k] * vec[idx[k]], added in index order. These entities stream from HBM/L2, where
// a dependent (index -> gather) chain costs two memory latencies: eight entries are loaded, then gathered, at a
// time, so a typical row (or short column) costs two latencies instead of two per entry.
template <bool SC1 = false>
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
    for (int q = 0; q < 8; ++q) xv[q] = (k + q < k1) ? ld_x<SC1>(vec + c[q]) : 0.0;
#pragma unroll
    for (int q = 0; q < 8; ++q)
      if (k + q < k1) acc += (double)v[q] * xv[q];
  }
  return acc;
}

// The two passes over the entity's matrix at W.x: logits, per-sample loss and residuals (CSR copy), then X'r by tiles of 64
// coefficients per wavefront (CSC copy). Leaves g in W.g — every coefficient written by the thread that owns it (tile =
// wavefront index + k * wavefronts of the team, lane = coefficient inside the tile) — and returns the loss sum.
// SC1: x, the residuals and the partial sums of split columns are exchanged through sc1 accesses (ld_x / st_x above).
template <int NW, bool SC1>
__device__ __forceinline__ double team_fg(Team<NW>& tm, const EntityView& P, const SolveParams& o, const Work& W
#ifdef GDMIX_TEAM_PROFILE
                                          , unsigned long long (&prof_t)[8], unsigned long long& prof_last
#endif
                                          ) {
  const int n = P.n, p = P.p, ic = P.ic;
  const double* __restrict__ x = W.x;
  // ---- rows: logits, per-sample loss and residual
  double pr[3] = {0.0, 0.0, 0.0};
  const double x0 = ic ? ld_x<SC1>(x) : 0.0;
  for (int i = tm.tid; i < n; i += tm.NT) {
    const double a = gather_dot8<SC1>(P.csr_val, P.csr_col, x + ic, P.row_ptr[i], P.row_ptr[i + 1], x0);
    const double z = a + (double)P.o[i];
    const double yi = (double)P.y[i];
    const double wi = P.w ? (double)P.w[i] : 1.0;
    double ri;
    if (o.linear) {   // squared loss, fixed_effect_lr_lbfgs_model.py:356-358
      const double e = z - yi;
      pr[0] += wi * e * e;
      ri = 2.0 * wi * e;
    } else {
      pr[0] += logistic_terms(z, yi, wi, ri);
    }
    st_x<SC1>(W.rs + i, ri);
    pr[1] += ri;
  }
  TEAM_PROF(0);
  tm.reduce(pr);   // also makes rs[] visible to the whole team
  TEAM_PROF(1);
  const double loss = pr[0], rsum = pr[1];
  // ---- columns: X'r by tiles of 64 coefficients per wavefront, lane c of the tile ends up owning column c
  const int first_reg = (ic && !o.regularize_bias) ? 1 : 0;
  const double inv_n = o.sum_loss ? 1.0 : 1.0 / (double)n;
  TeamLds<NW>& L = *tm.L;
  const int n_long = L.n_long;
  if (n_long > 0) {
    // long columns first: 64 slices each, slices strided over the wavefronts of the team, partial sums to HBM
    for (int item = tm.wid; item < n_long * WAVE; item += tm.nwaves) {
      const int c = L.llist[item >> 6], sl = item & (WAVE - 1);
      const int b0 = P.col_ptr[c], e0 = P.col_ptr[c + 1];
      const int ssz = (e0 - b0 + WAVE - 1) >> 6;
      const int k0 = b0 + sl * ssz;
      const int k1 = (k0 + ssz < e0) ? k0 + ssz : e0;
      double s = 0.0;
      for (int k = k0 + tm.lane; k < k1; k += 8 * WAVE) {
        float v[8];
        int r[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int kk = k + q * WAVE;
          const bool ok = kk < k1;
          v[q] = ok ? P.csc_val[kk] : 0.0f;
          r[q] = ok ? P.csc_row[kk] : 0;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q)
          if (k + q * WAVE < k1) s += (double)v[q] * ld_x<SC1>(W.rs + r[q]);
      }
      s = wave_sum(s);
      if (tm.lane == 0) st_x<SC1>(W.part + item, s);
    }
    tm.sync();
  }
  double* const buf = L.buf[threadIdx.x >> 6];
  for (int tile = tm.wid; tile * WAVE < p; tile += tm.nwaves) {
    const int j = tile * WAVE + tm.lane;
    const bool valid = j < p;
    const bool feat = valid && !(ic && j == 0);
    int cb = 0, ce = 0;
    if (feat) { cb = P.col_ptr[j - ic]; ce = P.col_ptr[j - ic + 1]; }
    // split columns: their sum comes from the partials
    int li = -1;
    if (n_long > 0 && ce - cb >= TEAM_LONGC) {
      int lo = 0, hi = n_long - 1;
      const int c = j - ic;
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (L.llist[mid] < c) lo = mid + 1; else hi = mid;
      }
      li = (L.llist[lo] == c) ? lo : -1;
    }
    const int cbx = (li >= 0) ? ce : cb;   // empty range in the flat pass
    const int maxlen = (int)wave_max_nonneg((double)(ce - cbx));
    double mine = 0.0;
    if (maxlen <= TEAM_SHORT_COL) {
      mine = gather_dot8<SC1>(P.csc_val, P.csc_row, W.rs, cbx, ce, 0.0);
    } else {
      // the tile's columns are one contiguous run of the CSC arrays: the wavefront streams it in chunks (coalesced
      // loads, eight per lane in flight), parks the products in LDS, and every lane adds up its own column's part
      const int first = (ic && tile == 0) ? 1 : 0;
      const int fb = readlane_i(cb, first);
      const int fe = (int)wave_max_nonneg((double)ce);
      for (int s0 = fb; s0 < fe; s0 += TEAM_CHUNK) {
        const int s1 = (s0 + TEAM_CHUNK < fe) ? s0 + TEAM_CHUNK : fe;
        const int lo = cbx > s0 ? cbx : s0;
        const int hi = ce < s1 ? ce : s1;
        const unsigned long long owners = __ballot(lo < hi);
        if (owners == 0) continue;   // the chunk lies inside a split column
        double pr8[8];
        {
          float v[8];
          int r[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const int kk = s0 + tm.lane + q * WAVE;
            const bool ok = kk < s1;
            v[q] = ok ? P.csc_val[kk] : 0.0f;
            r[q] = ok ? P.csc_row[kk] : 0;
          }
#pragma unroll
          for (int q = 0; q < 8; ++q) pr8[q] = (s0 + tm.lane + q * WAVE < s1) ? (double)v[q] * ld_x<SC1>(W.rs + r[q]) : 0.0;
        }
        if (__popcll(owners) == 1) {
          // one column owns the whole chunk (other entries, if any, belong to split columns and are masked out)
          const int own = __ffsll((long long)owners) - 1;
          const int olo = readlane_i(lo, own), ohi = readlane_i(hi, own);
          double t = 0.0;
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const int kk = s0 + tm.lane + q * WAVE;
            if (kk >= olo && kk < ohi) t += pr8[q];
          }
          t = wave_sum(t);
          if (tm.lane == own) mine += t;
        } else {
#pragma unroll
          for (int q = 0; q < 8; ++q) buf[tm.lane + q * WAVE] = pr8[q];
          wave_lds_fence();
          for (int k = lo; k < hi; ++k) mine += buf[k - s0];
          wave_lds_fence();
        }
      }
    }
    // split columns of this tile: 64 partials each, eight columns in flight
    unsigned long long lm = __ballot(li >= 0);
    while (lm) {
      int cl[8];
      double t[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        cl[q] = -1;
        t[q] = 0.0;
        if (lm) {
          cl[q] = __ffsll((long long)lm) - 1;
          lm &= lm - 1;
          t[q] = ld_x<SC1>(W.part + (size_t)readlane_i(li, cl[q]) * WAVE + tm.lane);
        }
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        if (cl[q] >= 0) {
          const double tt = wave_sum(t[q]);
          if (tm.lane == cl[q]) mine = tt;
        }
      }
    }
    if (valid) {
      const double a = (ic && j == 0) ? rsum : mine;
      W.g[j] = inv_n * (a + ((j < first_reg) ? 0.0 : o.l2 * ld_x<SC1>(x + j)));
    }
  }
  TEAM_PROF(2);
  return loss;
}

// The products of team_eval over this thread's coefficients, with HC (>= col) history slots requested per coefficient.
// Two tiles per trip (one for the long histories, below): all 2 x (4 + HC) loads of the trip are in flight before the first product (the
// passes over the history are bound by how many bytes a CU keeps in flight, not by arithmetic). A wavefront takes its tiles in the
// same order either way: the sums do not depend on the tiles per trip.
template <int NW, int HC>
__device__ __forceinline__ void team_products(const Team<NW>& tm, const Work& W, int p, int m, int col, int head, int first_reg,
                                              double (&acc)[TEAM_KX]) {
  const double* __restrict__ x = W.x;
  // (with the direct products the long histories go one tile per trip: their 2 x HC pairs and 4 HC + 7 accumulators do not fit the registers)
  constexpr int T = (GDMIX_TEAM_DIRECT_AB && HC > GDMIX_TEAM_DIRECT_ONE_TILE_FROM) ? 1 : 2;
  for (int tile = tm.wid; tile * WAVE < p; tile += T * tm.nwaves) {
    int jv[T];
    bool ok[T];
#pragma unroll
    for (int u = 0; u < T; ++u) { jv[u] = (tile + u * tm.nwaves) * WAVE + tm.lane; ok[u] = jv[u] < p; }
    double xj[T], gj[T], dj[T], rj[T];
    double2 h[T][HC > 0 ? HC : 1];
#pragma unroll
    for (int u = 0; u < T; ++u) {
      const int j = ok[u] ? jv[u] : tm.lane;   // tile 0 always exists: a safe address for the lanes past the end
      xj[u] = x[j]; gj[u] = W.g[j]; dj[u] = W.d[j]; rj[u] = W.r[j];
#pragma unroll
      for (int i = 0; i < HC; ++i) {
        int sl = head + i;
        if (sl >= m) sl -= m;
        if (i >= m) sl = 0;
        h[u][i] = compact_hist(W, m, j)[sl * COMPACT_HIST_STRIDE];
      }
    }
#pragma unroll
    for (int u = 0; u < T; ++u) {
      if (ok[u]) {
        if (jv[u] >= first_reg) acc[0] += xj[u] * xj[u];
        acc[1] += gj[u] * dj[u];
        acc[2] += gj[u] * gj[u];
        const double yj = gj[u] - rj[u];
        acc[3] += yj * yj;
        acc[4] += yj * gj[u];
        acc[TEAM_RD] += rj[u] * dj[u];
        acc[TEAM_KX - 1] = fmax(acc[TEAM_KX - 1], fabs(gj[u]));
#pragma unroll
        for (int i = 0; i < HC; ++i) {
          const double hx = i < col ? h[u][i].x : 0.0, hy = i < col ? h[u][i].y : 0.0;
          acc[5 + i] += hx * yj;
          acc[5 + TEAM_MCAP + i] += hy * yj;
          if (GDMIX_TEAM_DIRECT_AB) {
            acc[TEAM_K - 1 + i] += hx * gj[u];
            acc[TEAM_K - 1 + TEAM_MCAP + i] += hy * gj[u];
          }
        }
      }
    }
  }
}

// f, g and every dot product the driver needs, at W.x, all vectors in HBM. acc[] layout: 0 sum x_j^2 over regularised j,
// 1 g'd, 2 g'g, 3 y'y, 4 y'g, 5.. S_i'y, 5+MCAP.. Y_i'y (y = g - r; chronological i < col), TEAM_RD r'd, then (direct products)
// S_i'g, Y_i'g, and max|g_j| last (Team::reduce takes the maximum of its last value).
template <int NW>
__device__ __forceinline__ double team_eval(Team<NW>& tm, const EntityView& P, const SolveParams& o, const Work& W,
                                            int col, int head, double (&acc)[TEAM_KX]
#ifdef GDMIX_TEAM_PROFILE
                                            , unsigned long long (&prof_t)[8], unsigned long long& prof_last
#endif
                                            ) {
  const int n = P.n, p = P.p, ic = P.ic, m = o.m;
  const int first_reg = (ic && !o.regularize_bias) ? 1 : 0;
  const double inv_n = o.sum_loss ? 1.0 : 1.0 / (double)n;
#ifdef GDMIX_TEAM_PROFILE
  const double loss = team_fg<NW, false>(tm, P, o, W, prof_t, prof_last);
#else
  const double loss = team_fg<NW, false>(tm, P, o, W);
#endif
  // Products with the gradient in a second sweep over the same coefficients (the thread that stored g[j] reads it
  // back: no synchronisation), so that the accumulators and the tile staging above are not live at the same time.
#pragma unroll
  for (int k = 0; k < TEAM_KX; ++k) acc[k] = 0.0;
  GDMIX_HIST_DISPATCH(col, (team_products<NW, HC>(tm, W, p, m, col, head, first_reg, acc)))
  TEAM_PROF(2);
  tm.reduce(acc);
  TEAM_PROF(3);
  return inv_n * (loss + 0.5 * o.l2 * acc[0]);
}

// The whole fmin_l_bfgs_b run for one entity by a team. Requires 1 <= o.m <= TEAM_MCAP. W.x holds theta0 on
// entry (visible to the team), theta on exit.
// Which columns get split (TEAM_LONGC): every workgroup scans all column lengths and ends with the same ascending
// list in its LDS. More than TEAM_LONG_CAP of them: none is split (they are then streamed by their tile's wavefront).
template <int NW>
__device__ __forceinline__ void team_long_setup(Team<NW>& tm, const EntityView& P) {
  TeamLds<NW>& L = *tm.L;
  if (threadIdx.x == 0) { L.n_long = 0; L.n_long_raw = 0; }
  __syncthreads();
  if (P.col_ptr[P.d] < TEAM_LONGC) return;   // uniform: fewer non-zeros than one long column
  for (int c = threadIdx.x; c < P.d; c += blockDim.x) {
    if (P.col_ptr[c + 1] - P.col_ptr[c] >= TEAM_LONGC) {
      const int pos = atomicAdd(&L.n_long_raw, 1);
      if (pos < TEAM_LONG_CAP) L.lraw[pos] = c;
    }
  }
  __syncthreads();
  const int cnt = L.n_long_raw;
  if (cnt > TEAM_LONG_CAP) return;
  if ((int)threadIdx.x < cnt) {
    const int mine = L.lraw[threadIdx.x];
    int rank = 0;
    for (int k = 0; k < cnt; ++k) rank += (L.lraw[k] < mine) ? 1 : 0;
    L.llist[rank] = mine;
  }
  if (threadIdx.x == 0) L.n_long = cnt;
  __syncthreads();
}

template <int NW>
__device__ __forceinline__ void team_solve(Team<NW>& tm, const EntityView& P, const SolveParams& o, const Work& W, SolveStats& out) {
  const int p = P.p, m = o.m;
  TeamLds<NW>& L = *tm.L;
  team_long_setup(tm, P);
  CompactState S;
  compact_init(S);
  CompactPlan plan;
  double acc[TEAM_KX];
  TEAM_PROF_DECL
  for (int j = tm.tid; j < p; j += tm.NT) { W.d[j] = 0.0; W.r[j] = 0.0; }
  tm.sync();
  unsigned upd = 0;   // updates so far = trials formed
  tm.moved_clear(0u); tm.moved_clear(1u);
  tm.sync();
  for (;;) {
    tm.moved_clear(upd + 1u);   // the slot the next update uses
#ifdef GDMIX_TEAM_PROFILE
    const double f_new = team_eval(tm, P, o, W, S.col, S.head, acc, prof_t, prof_last);
#else
    const double f_new = team_eval(tm, P, o, W, S.col, S.head, acc);
#endif
    if (tm.aborted()) { ++S.nfev; S.status = GDMIX_RE_ST_ABORTED; break; }
#if GDMIX_TEAM_DIRECT_AB
    {
      double a[TEAM_K];
#pragma unroll
      for (int k = 0; k < TEAM_K - 1; ++k) a[k] = acc[k];
      a[TEAM_K - 1] = acc[TEAM_KX - 1];
      compact_advance(S, a, f_new, o, L.mats, plan, tm.moved_get(upd), acc + (TEAM_K - 1));
    }
#else
    compact_advance(S, acc, f_new, o, L.mats, plan, tm.moved_get(upd));
#endif
    TEAM_PROF(4);
    if (plan.action == CA_STOP) break;
    if (plan.action == CA_STOP_RESTORE) {
      for (int j = tm.tid; j < p; j += tm.NT) W.x[j] = W.t[j];
      tm.sync();
      break;
    }
    ++upd;
    {
      unsigned* const mv = tm.moved_slot(upd);   // plain stores of 1 (a team's workgroups: made visible by the barrier's release)
      GDMIX_HIST_DISPATCH((plan.action == CA_DIRECTION ? plan.col : 0),
                          for (int j = tm.tid; j < p; j += tm.NT) compact_update_n<HC>(plan, L.mats, W, p, m, j, mv))
    }
    TEAM_PROF(5);
    tm.sync();
    TEAM_PROF(6);
  }
#ifdef GDMIX_TEAM_PROFILE
  if (tm.tid == 0 && blockIdx.x % 64 == 0)   // a few lines only: device printf perturbs the other workgroups
    printf("team n=%d p=%d nfev=%d us/eval: rows %.1f red1 %.1f cols %.1f red2 %.1f solve %.1f upd %.1f sync %.1f\n", P.n, p, S.nfev,
           prof_t[0] * 0.01 / S.nfev, prof_t[1] * 0.01 / S.nfev, prof_t[2] * 0.01 / S.nfev, prof_t[3] * 0.01 / S.nfev,
           prof_t[4] * 0.01 / S.nfev, prof_t[5] * 0.01 / S.nfev, prof_t[6] * 0.01 / S.nfev);
#endif
  out.f = S.f;
  out.gnorm = S.sbgnrm;
  out.nit = S.nit;
  out.nfev = S.nfev;
  out.status = S.status;
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
