// re_pack.hip — device-side replacement of the per-entity slicing in prepare_jobs
// (gdmix-trainer/src/gdmix/models/custom/scipy/job_consumers.py:209-258): for every entity,
// np.unique(cols, return_inverse=True) (:243), the locally indexed COO/CSR block (:247) and — new
// here — a column-sorted copy so that X'r needs no atomics. One wavefront per entity.
//
// Sort key = (global column << 32) | position-in-entity, so equal columns stay in row-major order
// (the order scipy's COO mat-vec accumulates them in) and the sort needs no stability.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

#include "re_internal.hpp"

namespace gdmix {

constexpr int PACK_WAVES = 4;          // wavefronts (entities) per workgroup
#ifndef GDMIX_PACK_LDS_KEYS
#define GDMIX_PACK_LDS_KEYS 256
#endif
constexpr int PACK_LDS_KEYS = GDMIX_PACK_LDS_KEYS;     // per-wave LDS sort capacity; larger entities sort in HBM scratch

struct PackStats {      // device-side, read back once per pack
  unsigned long long D;
  int max_p, max_n, max_nnz, err;
  int n_big, n_mid;       // entities deferred to the device-wide sort / the larger-LDS wavefront pack kernel
  unsigned long long big_nnz;   // non-zeros of the n_big entities
  int n_mid2, pad;        // entities deferred from the 512-key to the 1024-key kernel
};

// ent_nnz_ptr, and the route of every entity through the pack: by max(samples, non-zeros) the 256-key wavefront tier (nothing to
// do here: that tier walks all entities and skips the others), the 512-key list, the 1 024-key list, or the list of the device-wide
// sort. The lists are complete when this kernel ends, so the tiers and the device-wide path run next to each other (round 4; before,
// each tier appended what it could not hold to the next one's list: three launches and a read-back in a row, ~0.45 ms of a 17 k-entity
// MovieLens share's 0.83 ms pack). One atomic per wavefront and list (ballot + count); the order inside a list is arbitrary, as before,
// and matters to nothing: every entity's output lies at offsets of its own.
constexpr int64_t PACK_FAN_MAX_ENTITIES = 1 << 18;   // batches up to this size run the pack's stages on several streams (pack_impl)
constexpr int PACK_CAP1 = PACK_LDS_KEYS, PACK_CAP2 = 512, PACK_CAP3 = 1024;   // the tiers' capacities (pack_entity_kernel<CAP, ...>)
__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const int w = __shfl_xor(v, o); v = w > v ? w : v; }
  return v;
}
constexpr int ENTNNZ_THREADS = 1024;
__global__ __launch_bounds__(ENTNNZ_THREADS) void pack_entnnz_kernel(const int64_t* __restrict__ ent_row_ptr, const int64_t* __restrict__ row_nnz_ptr,
                                                          int64_t E, int64_t* __restrict__ ent_nnz_ptr, int32_t* __restrict__ mid_list,
                                                          int32_t* __restrict__ mid2_list, int32_t* __restrict__ big_list,
                                                          PackStats* __restrict__ stats) {
  // One atomic per WORKGROUP, list and trip (round 5; per wavefront before: a MovieLens population puts most of its entities on the
  // lists, 2 164 wavefronts x 3 same-address atomics were 0.15 ms for 138 k users against 0.04 ms for C2's million, which lists none).
  constexpr int NW = ENTNNZ_THREADS / WAVE;
  __shared__ int wcnt[3][NW];      // entities of wavefront w for list t
  __shared__ int wbase[3];         // the workgroup's first slot in list t
  __shared__ unsigned long long s_nnz;
  __shared__ int s_mn, s_mz;
  const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x >> 6;
  if (threadIdx.x == 0) { s_nnz = 0ull; s_mn = 0; s_mz = 0; }
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t rounds = (E + 1 + stride - 1) / stride;     // every thread of a workgroup makes the same number of trips (barriers below)
  for (int64_t r = 0; r < rounds; ++r) {
    const int64_t e = r * stride + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int tier = 0;
    int n = 0, z = 0;
    if (e <= E) {
      const int64_t r0 = ent_row_ptr[e];
      const int64_t z0 = row_nnz_ptr[r0];
      ent_nnz_ptr[e] = z0;
      if (e < E) {
        const int64_t r1 = ent_row_ptr[e + 1];
        const int64_t n64 = r1 - r0, z64 = row_nnz_ptr[r1] - z0;
        if (n64 <= 0x7ffffff0ll && z64 <= 0x7ffffff0ll) {   // (beyond: the first tier reports the entity, GDMIX_RE_ERANGE)
          n = (int)n64; z = (int)z64;
          const int mx = n > z ? n : z;
          tier = mx <= PACK_CAP1 ? 0 : (mx <= PACK_CAP2 ? 1 : (mx <= PACK_CAP3 ? 2 : 3));
        }
      }
    }
    unsigned long long mask[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      mask[t] = __ballot(tier == t + 1);
      if (lane == 0) wcnt[t][wv] = __popcll(mask[t]);
    }
    __syncthreads();
    if (threadIdx.x < 3) {
      const int t = threadIdx.x;
      int total = 0;
      for (int w = 0; w < NW; ++w) total += wcnt[t][w];
      int* const counter = t == 0 ? &stats->n_mid : (t == 1 ? &stats->n_mid2 : &stats->n_big);
      wbase[t] = total ? atomicAdd(counter, total) : 0;
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      if (tier == t + 1) {
        int base = wbase[t];
        for (int w = 0; w < wv; ++w) base += wcnt[t][w];
        int32_t* const list = t == 0 ? mid_list : (t == 1 ? mid2_list : big_list);
        list[base + __popcll(mask[t] & ((1ull << lane) - 1ull))] = (int32_t)e;
      }
    }
    if (mask[2] != 0ull) {   // the device-wide sort's entities: their total, and their part of the batch's maxima
      const unsigned long long zz = (unsigned long long)wave_sum_u64(tier == 3 ? (unsigned long long)z : 0ull);
      const int mn = wave_max_i32(tier == 3 ? n : 0), mz = wave_max_i32(tier == 3 ? z : 0);
      if (lane == 0) { atomicAdd(&s_nnz, zz); atomicMax(&s_mn, mn); atomicMax(&s_mz, mz); }
    }
    __syncthreads();   // wcnt / wbase are rewritten by the next trip
  }
  if (threadIdx.x == 0 && (s_nnz != 0ull || s_mn != 0 || s_mz != 0)) {
    if (s_nnz != 0ull) atomicAdd(&stats->big_nnz, s_nnz);
    if (s_mn > __atomic_load_n(&stats->max_n, __ATOMIC_RELAXED)) atomicMax(&stats->max_n, s_mn);
    if (s_mz > __atomic_load_n(&stats->max_nnz, __ATOMIC_RELAXED)) atomicMax(&stats->max_nnz, s_mz);
  }
}

// ---- per-entity pack -----------------------------------------------------------------------------------------
// Sort key = (global column << 32) | position-in-entity: equal columns stay in row-major order (the order
// scipy's COO mat-vec accumulates them in) and the sort needs no stability.
//
// Small entities (nnz <= PACK_LDS_KEYS) are packed entirely out of LDS in one pass: keys + values staged in
// LDS, RANK SORT for nnz <= 128 (every lane counts the keys below its own against LDS broadcast reads:
// 2 VALU per comparison instead of the ~40 per compare-exchange step of a bitonic network), head-flag ballot
// scan, then all outputs. col_ptr and the per-entity unique list are addressed by the entity's NON-ZERO
// offset (z0 + e, z0), which is known up front, so nothing here waits for the prefix sum over d_e; the compact
// unique_global[D] the host reads is produced by a small gather afterwards.
constexpr int PACK_RANK_MAX = 128;

template <class KeyPtr>
__device__ __forceinline__ void wave_rank_sort(KeyPtr keys, KeyPtr sorted, int n, int lane) {
  // n <= 128: lane owns keys lane and lane + 64
  const unsigned long long k0 = (lane < n) ? keys[lane] : ~0ull;
  const unsigned long long k1 = (lane + WAVE < n) ? keys[lane + WAVE] : ~0ull;
  int r0 = 0, r1 = 0;
  for (int j = 0; j < n; ++j) {
    const unsigned long long kj = keys[j];   // same address in every lane: LDS broadcast
    r0 += (kj < k0) ? 1 : 0;
    r1 += (kj < k1) ? 1 : 0;
  }
  wave_lds_fence();
  if (lane < n) sorted[r0] = k0;             // keys are distinct (position is part of the key)
  if (lane + WAVE < n) sorted[r1] = k1;
  wave_lds_fence();
}

// All comparators ascending, so keys at index >= n act as +inf padding and are simply skipped. Keys in LDS (the passes are fenced
// for LDS only: a fence that also drained the wavefront's global accesses waited, 55 times per 1 024 keys, for the stores of the
// entity before and the loads of the one after).
template <class KeyPtr>
__device__ __forceinline__ void wave_bitonic_sort_lds(KeyPtr a, int n, int lane) {
  if (n < 2) return;
  int np2 = 1;
  while (np2 < n) np2 <<= 1;
  const int half = np2 >> 1;
  for (int k = 2; k <= np2; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = lane; t < half; t += WAVE) {
        int i, partner;
        if (j == (k >> 1)) {
          const int blk = t / j, off = t - blk * j;
          i = blk * k + off;
          partner = blk * k + (k - 1 - off);
        } else {
          const int blk = t / j, off = t - blk * j;
          i = blk * 2 * j + off;
          partner = i + j;
        }
        if (partner < n) {
          const unsigned long long x = a[i], y = a[partner];
          if (x > y) { a[i] = y; a[partner] = x; }
        }
      }
      wave_lds_fence();
    }
  }
}

// Outputs of one entity from its sorted keys (any address space) and a value source indexed by position.
template <class KeyPtr, class ValPtr>
__device__ __forceinline__ int emit_entity(KeyPtr sk, ValPtr vals, const int32_t* __restrict__ rp, int n, int nnz,
                                           int lane, int32_t* __restrict__ csr_col, int32_t* __restrict__ col_ptr,
                                           int32_t* __restrict__ csc_row, float* __restrict__ csc_val,
                                           int32_t* __restrict__ uniq_sparse) {
  int carry = 0;
  for (int base = 0; base < nnz; base += WAVE) {
    const int k = base + lane;
    bool head = false;
    unsigned long long key = 0;
    if (k < nnz) {
      key = sk[k];
      unsigned long long prev = ~key;
      if (k > 0) prev = sk[k - 1];
      head = (prev >> 32) != (key >> 32);
    }
    const unsigned long long mask = __ballot(head);
    const unsigned long long below = (lane == 63) ? ~0ull : ((2ull << lane) - 1ull);
    const int lid = carry + __popcll(mask & below) - 1;
    if (k < nnz) {
      const int pos = (int)(key & 0xffffffffull);
      if (head) {
        uniq_sparse[lid] = (int32_t)(key >> 32);
        col_ptr[lid] = k;
      }
      csr_col[pos] = lid;
      csc_val[k] = vals[pos];
      int lo = 0, hi = n - 1;   // sample of non-zero `pos`: last i with rp[i] <= pos
      while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (rp[mid] <= pos) lo = mid; else hi = mid - 1;
      }
      csc_row[k] = lo;
    }
    carry += __popcll(mask);
  }
  if (lane == 0) col_ptr[carry] = nnz;
  return carry;
}

// Inclusive prefix sum over the 64 lanes by DPP: four shifts inside the rows of 16, then the last lane of row 0 / 2 into rows 1 / 3 and the
// last lane of the lower half into the upper half (row_bcast:15, row_bcast:31) — six VALU instructions where six __shfl_up steps are
// six LDS permutes with their address arithmetic.
__device__ __forceinline__ unsigned wave_incl_scan_u32(unsigned v) {
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);   // row_shr:1
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);   // row_shr:2
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);   // row_shr:4
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);   // row_shr:8
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1, 3
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2, 3
  return v;
}

// Inclusive prefix maximum over the 64 lanes, same DPP ladder.
__device__ __forceinline__ unsigned wave_incl_max_u32(unsigned v) {
  unsigned w;
  w = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false); v = w > v ? w : v;   // row_shr:1
  w = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false); v = w > v ? w : v;   // row_shr:2
  w = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false); v = w > v ? w : v;   // row_shr:4
  w = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false); v = w > v ? w : v;   // row_shr:8
  w = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false); v = w > v ? w : v;   // row_bcast:15 -> rows 1, 3
  w = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false); v = w > v ? w : v;   // row_bcast:31 -> rows 2, 3
  return v;
}

// An entity of more than PACK_RANK_MAX non-zeros all of whose columns are below 64 (MovieLens bags: 20 and 24 features) needs no
// sort: lane c counts column c, a scan over the lanes gives every column's first CSC position and local id, and the entries are
// placed 64 at a time in position order (lanes of equal column rank themselves with a ballot per distinct column of the tile;
// in-order LDS, one wavefront, no atomics in the placement). keys: the staged (column << 32 | position) keys, unsorted; base: 64
// words of LDS. Same outputs as the sort + emit_entity, without the 55 passes of the bitonic network over 1 024 keys.
constexpr int PACK_COUNT_COLS = 64;
template <class ValPtr>
__device__ __forceinline__ int count_entity(unsigned long long* keys, unsigned* base, ValPtr vals, const int32_t* __restrict__ rp, int n,
                                            int nnz, int lane, int32_t* __restrict__ csr_col, int32_t* __restrict__ col_ptr,
                                            int32_t* __restrict__ csc_row, float* __restrict__ csc_val,
                                            int32_t* __restrict__ uniq_sparse) {
  // The low word of a key (the position, which is the key's index here) makes room for the row marks: row i > 0 marks its first position
  // with i, a prefix maximum per tile + a carry gives every position its sample (as bitmap_entity does; a bisection of the row pointers per
  // non-zero was a quarter of this path's instructions — MovieLens entities have hundreds of samples: ten steps each).
  unsigned* const kw = reinterpret_cast<unsigned*>(keys);
  base[lane] = 0u;
  for (int k = lane; k < nnz; k += WAVE) kw[2 * k] = 0u;
  wave_lds_fence();
  for (int k = lane; k < nnz; k += WAVE) atomicAdd(&base[kw[2 * k + 1]], 1u);   // integer counts: any order
  for (int i = lane; i < n; i += WAVE) {
    const int s = rp[i];
    if (i > 0 && s < nnz) atomicMax(&kw[2 * s], (unsigned)i);
  }
  wave_lds_fence();
  const unsigned tot = base[lane];
  unsigned xn = tot;
#pragma unroll
  for (int d = 1; d < WAVE; d <<= 1) {
    const unsigned yn = __shfl_up(xn, d);
    if (lane >= d) xn += yn;
  }
  const unsigned start = xn - tot;
  const unsigned long long present = __ballot(tot != 0u);
  const int d = __popcll(present);
  if (tot) {
    const int lid = __popcll(present & ((1ull << lane) - 1ull));
    uniq_sparse[lid] = lane;
    col_ptr[lid] = (int32_t)start;
  }
  if (lane == 0) col_ptr[d] = nnz;
  wave_lds_fence();
  base[lane] = start;
  wave_lds_fence();
  unsigned row_carry = 0;
  for (int t = 0; t < nnz; t += WAVE) {
    const int k = t + lane;
    const bool valid = k < nnz;
    const unsigned c = valid ? kw[2 * k + 1] : 0u;
    unsigned row = wave_incl_max_u32(valid ? kw[2 * k] : 0u);
    row = row > row_carry ? row : row_carry;
    row_carry = (unsigned)__builtin_amdgcn_readlane((int)row, WAVE - 1);
    unsigned rank = 0, cnt = 0;
    unsigned long long todo = __ballot(valid);
    while (todo) {
      const int leader = __ffsll((long long)todo) - 1;
      const unsigned cl = (unsigned)__shfl((int)c, leader);
      const unsigned long long m = __ballot(valid && c == cl);
      if (valid && c == cl) {
        rank = (unsigned)__popcll(m & ((1ull << lane) - 1ull));
        cnt = (unsigned)__popcll(m);
      }
      todo &= ~m;
    }
    if (valid) {
      const unsigned kpos = base[c] + rank;
      csr_col[k] = __popcll(present & ((1ull << c) - 1ull));
      csc_val[kpos] = vals[k];
      csc_row[kpos] = (int32_t)row;
    }
    wave_lds_fence();
    if (valid && rank == 0) base[c] += cnt;
    wave_lds_fence();
  }
  return d;
}

// An entity of at most PACK_RANK_MAX non-zeros all of whose columns are below PACK_BITMAP_COLS (C2's bag: 1 024 features; the
// MovieLens bags) needs no sort either: the set of its columns is a 2 048-bit map, one 32-bit word per lane; a column's local id is
// the number of set bits below it (a prefix over the words' popcounts + a masked popcount), col_ptr is a scan over the per-id counts,
// and an entry's CSC slot is its column's start plus its rank among the entries of the same column in position order — 0 for all
// but the few columns that occur twice (ballot ranking inside a 64-entry tile, as count_entity does, for those columns only).
// Same outputs as wave_rank_sort + emit_entity, bit for bit (tests/test_gpu_parity.py::_check_pack on every fixture), without the
// rank sort's nnz broadcast-compare trips: ~470 -> ~250 wavefront instructions for a 64-entry entity (pack_entity_kernel<256> on C2:
// 1.01 ms, three quarters of it VALU issue — profiles/r04_final_c2_1m.txt). w: 5 x 64 words of LDS.
constexpr int PACK_BITMAP_COLS = 2048;
constexpr int PACK_BITMAP_WORDS = 5 * WAVE;
template <class ValPtr>
__device__ __forceinline__ int bitmap_entity(const unsigned long long* keys, unsigned* w, unsigned* rowid, ValPtr vals, const int32_t* __restrict__ rp, int n,
                                             int nnz, int lane, int32_t* __restrict__ csr_col, int32_t* __restrict__ col_ptr,
                                             int32_t* __restrict__ csc_row, float* __restrict__ csc_val,
                                             int32_t* __restrict__ uniq_sparse) {
  unsigned* const bm = w;               // [64] the columns present
  unsigned* const pre = w + WAVE;       // [64] set bits in the words below
  unsigned* const cnt = w + 2 * WAVE;   // [128] entries per local id
  unsigned* const start = w + 4 * WAVE; // [64] CSC start of ids 0 .. 63 (ids 64 .. 127: in the bitmap's words once it is done with)
  bm[lane] = 0u;
  cnt[lane] = 0u;
  cnt[lane + WAVE] = 0u;
  rowid[lane] = 0u;
  rowid[lane + WAVE] = 0u;
  wave_lds_fence();
  // the sample of every non-zero = the last row that starts at or before it: row i > 0 marks its first position with i (the largest
  // wins where empty rows share a start), a prefix maximum over the positions carries the marks forward (round 5; a binary search
  // per non-zero over the row pointers was ~120 of this path's ~275 vector instructions per entity)
  for (int i = lane; i < n; i += WAVE) {
    const int s = rp[i];
    if (i > 0 && s < nnz) atomicMax(&rowid[s], (unsigned)i);
  }
  const bool v0 = lane < nnz, v1 = lane + WAVE < nnz;
  const unsigned c0 = v0 ? (unsigned)(keys[lane] >> 32) : 0u, c1 = v1 ? (unsigned)(keys[lane + WAVE] >> 32) : 0u;
  if (v0) atomicOr(&bm[c0 >> 5], 1u << (c0 & 31u));
  if (v1) atomicOr(&bm[c1 >> 5], 1u << (c1 & 31u));
  wave_lds_fence();
  const unsigned word = bm[lane];
  const unsigned pc = (unsigned)__popc(word);
  const unsigned incl = wave_incl_scan_u32(pc);
  const int d = __builtin_amdgcn_readlane((int)incl, WAVE - 1);
  pre[lane] = incl - pc;
  wave_lds_fence();
  unsigned l0 = 0, l1 = 0;
  if (v0) {
    l0 = pre[c0 >> 5] + (unsigned)__popc(bm[c0 >> 5] & ((1u << (c0 & 31u)) - 1u));
    csr_col[lane] = (int32_t)l0;
    atomicAdd(&cnt[l0], 1u);
  }
  if (v1) {
    l1 = pre[c1 >> 5] + (unsigned)__popc(bm[c1 >> 5] & ((1u << (c1 & 31u)) - 1u));
    csr_col[lane + WAVE] = (int32_t)l1;
    atomicAdd(&cnt[l1], 1u);
  }
  // the columns of this lane's word, ascending
  {
    unsigned rest = word;
    int out = (int)(incl - pc);
    while (__any(rest != 0u)) {
      if (rest) {
        const int b = __ffs((int)rest) - 1;
        uniq_sparse[out++] = lane * 32 + b;
        rest &= rest - 1u;
      }
    }
  }
  wave_lds_fence();
  // col_ptr: exclusive scan of the counts of ids 0 .. d - 1 (two per lane)
  const unsigned x0 = cnt[lane];
  const unsigned s0 = wave_incl_scan_u32(x0);
  const unsigned b0 = s0 - x0;
  start[lane] = b0;
  if (lane < d) col_ptr[lane] = (int32_t)b0;
  if (lane == 0) col_ptr[d] = nnz;
  unsigned* const start_hi = bm;   // the bitmap is done with: its 64 words hold the starts of ids 64 .. 127
  if (d > WAVE) {                  // (uniform) ids of the second half exist
    const unsigned x1 = cnt[lane + WAVE];
    const unsigned b1 = (unsigned)__builtin_amdgcn_readlane((int)s0, WAVE - 1) + wave_incl_scan_u32(x1) - x1;
    if (lane + WAVE < d) col_ptr[lane + WAVE] = (int32_t)b1;
    wave_lds_fence();
    start_hi[lane] = b1;
  }
  wave_lds_fence();
  const int tiles = nnz > WAVE ? 2 : 1;
  const unsigned row0 = wave_incl_max_u32(rowid[lane]);
  unsigned row1 = 0;
  if (tiles > 1) {
    const unsigned m = wave_incl_max_u32(rowid[lane + WAVE]), top = (unsigned)__builtin_amdgcn_readlane((int)row0, WAVE - 1);
    row1 = m > top ? m : top;
  }
  for (int t = 0; t < tiles; ++t) {
    const int k = t * WAVE + lane;
    const bool valid = t ? v1 : v0;
    const unsigned lid = t ? l1 : l0;
    const bool dup = valid && cnt[lid] > 1u;
    unsigned rank = 0, tilecnt = 1;
    unsigned long long todo = __ballot(dup);
    while (todo) {
      const int leader = __ffsll((long long)todo) - 1;
      const unsigned ll = (unsigned)__shfl((int)lid, leader);
      const unsigned long long m = __ballot(valid && lid == ll);
      if (valid && lid == ll) {
        rank = (unsigned)__popcll(m & ((1ull << lane) - 1ull));
        tilecnt = (unsigned)__popcll(m);
      }
      todo &= ~m;
    }
    unsigned* const st = lid < (unsigned)WAVE ? &start[lid] : &start_hi[lid - WAVE];
    if (valid) {
      const unsigned kpos = *st + rank;
      csc_val[kpos] = vals[k];
      csc_row[kpos] = (int32_t)(t ? row1 : row0);
    }
    if (t + 1 < tiles) {   // a second tile follows: the columns that occur more than once move their start on
      wave_lds_fence();
      if (dup && rank == 0) *st += tilecnt;
      wave_lds_fence();
    }
  }
  return d;
}

// CAP = LDS staging capacity (non-zeros and samples) of a wavefront, NWAVES = wavefronts per workgroup.
// in_list == nullptr: all E entities, those above CAP skipped (pack_entnnz_kernel put them on a later stage's list);
// otherwise the *in_count entities of in_list, which fit by construction.
//
// The walk is software-pipelined (round 5; before, a wavefront's trip was three memory latencies in a row — the entity's four pointers, its
// columns / values / row pointers, and a wait for its own stores before the LDS staging was reused — and the kernel spent 71 % of its
// wavefront-cycles in s_waitcnt at 2.2 TB/s, profiles/r05_final_c2_1m.txt):
//   * the pointers of a wavefront's next 64 trips are loaded at once, one trip per lane, and handed out with readlane (scalars);
//   * the first PACK_PF tiles of the NEXT trip's columns and values and its first 64 row pointers are requested right after the
//     current trip's are staged into LDS — into the registers just consumed, so nothing is copied and nothing waits — and arrive
//     while the current entity is ranked and written;
//   * the end of a trip fences LDS only (the staging buffers); stores are never waited for.
constexpr int PACK_PF = 4;       // 64-entry tiles of an entity requested one trip ahead (256 non-zeros: all of the first tier)
struct PackAhead {
  int64_t c[PACK_PF];
  float v[PACK_PF];
  int32_t rp;      // low word of the row pointer (the difference to the entity's first fits 31 bits; a register pair half of which
                   // is dead gets reused while the load is in flight, and the wavefront then waits for it)
};
__device__ __forceinline__ int64_t readlane_i64(int64_t x, int src) {
  const int lo = __builtin_amdgcn_readlane((int)(uint32_t)(uint64_t)x, src);
  const int hi = __builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)x >> 32), src);
  return (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint64_t)(uint32_t)lo);
}
__device__ __forceinline__ void pack_request(PackAhead& A, const int64_t* __restrict__ row_nnz_ptr, const int64_t* __restrict__ col_global,
                                             const float* __restrict__ val, int64_t r0, int64_t z0, int n, int nnz, int lane) {
#pragma unroll
  for (int q = 0; q < PACK_PF; ++q) {
    const int k = q * WAVE + lane;
    if (k < nnz) {
      A.c[q] = col_global[z0 + k];
      A.v[q] = val[z0 + k];
    }
  }
  if (lane <= n) A.rp = reinterpret_cast<const int32_t*>(row_nnz_ptr + r0)[2 * lane];
}

template <int CAP, int NWAVES>
__global__ __launch_bounds__(WAVE* NWAVES) void pack_entity_kernel(
    const int32_t* __restrict__ in_list, const int* __restrict__ in_count,
    const int64_t* __restrict__ ent_row_ptr, const int64_t* __restrict__ row_nnz_ptr,
    const int64_t* __restrict__ ent_nnz_ptr, const int64_t* __restrict__ col_global, const float* __restrict__ val,
    int64_t E, int ic, int32_t* __restrict__ row_ptr, unsigned long long* __restrict__ sort_key,
    int32_t* __restrict__ csr_col, int32_t* __restrict__ col_ptr, int32_t* __restrict__ csc_row,
    float* __restrict__ csc_val, int32_t* __restrict__ uniq_sparse, int32_t* __restrict__ d_cnt,
    PackStats* __restrict__ stats, int pack_bitmap_on) {
  __shared__ unsigned long long lds_keys[NWAVES][CAP];
  __shared__ unsigned long long lds_sorted[NWAVES][PACK_RANK_MAX];
  __shared__ unsigned lds_bm[NWAVES][PACK_BITMAP_WORDS];
  __shared__ float lds_val[NWAVES][CAP];
  __shared__ int32_t lds_rp[NWAVES][CAP + 1];
  __shared__ int blk_max[3];
  const int lane = threadIdx.x & (WAVE - 1);
  const int wv = threadIdx.x >> 6;
  if (threadIdx.x < 3) blk_max[threadIdx.x] = 0;
  __syncthreads();
  int mx_p = 0, mx_n = 0, mx_z = 0;
  // grid-stride over entities: per-entity same-address atomics would serialise in L2 (~10 ns each), so maxima
  // are carried in registers and published once per workgroup.
  const int64_t total = in_list ? (int64_t)*in_count : E;
  const int64_t stride = (int64_t)gridDim.x * NWAVES;
  enum { SKIP = 0, SMALL = 1, RANGE = 2 };
  for (int64_t it0 = (int64_t)blockIdx.x * NWAVES + wv; it0 < total; it0 += stride * WAVE) {
    // this lane's trip of the next 64
    const int64_t my_it = it0 + (int64_t)lane * stride;
    int64_t my_e = 0, my_r0 = 0, my_z0 = 0;
    int my_n = 0, my_nnz = 0, my_state = SKIP;
    if (my_it < total) {
      my_e = in_list ? (int64_t)in_list[my_it] : my_it;
      my_r0 = ent_row_ptr[my_e];
      my_z0 = ent_nnz_ptr[my_e];
      const int64_t n64 = ent_row_ptr[my_e + 1] - my_r0, nnz64 = ent_nnz_ptr[my_e + 1] - my_z0;
      if (n64 > 0x7ffffff0ll || nnz64 > 0x7ffffff0ll) {
        my_state = RANGE;
      } else {
        my_n = (int)n64; my_nnz = (int)nnz64;
        // not small: a later stage's (which also writes its row pointers: one wavefront walking the rows of a 4 M-sample entity here
        // cost 27 ms; the last stage does it one thread per row)
        my_state = (my_nnz <= CAP && my_n <= CAP) ? SMALL : SKIP;
      }
    }
    const int64_t left = (total - it0 + stride - 1) / stride;
    const int cnt = __builtin_amdgcn_readfirstlane((int)(left < WAVE ? left : WAVE));
    PackAhead A;
    if (__builtin_amdgcn_readlane(my_state, 0) == SMALL)
      pack_request(A, row_nnz_ptr, col_global, val, readlane_i64(my_r0, 0), readlane_i64(my_z0, 0), __builtin_amdgcn_readlane(my_n, 0),
                   __builtin_amdgcn_readlane(my_nnz, 0), lane);
    for (int t = 0; t < cnt; ++t) {
      const int state = __builtin_amdgcn_readlane(my_state, t);
      const int64_t e = readlane_i64(my_e, t), r0 = readlane_i64(my_r0, t), z0 = readlane_i64(my_z0, t);
      const int n = __builtin_amdgcn_readlane(my_n, t), nnz = __builtin_amdgcn_readlane(my_nnz, t);
      // all the columns OR-ed: the three questions asked of them are thresholds at powers of two (any column outside [0, 2^31)?
      // any of PACK_COUNT_COLS or above? any of PACK_BITMAP_COLS or above?), which the OR answers for every one of them at once
      uint64_t orc = 0;
      bool bad = false;
      int32_t* const rp_out = row_ptr + r0 + e;
      if (state == SMALL) {   // stage what was requested a trip ago
#pragma unroll
        for (int q = 0; q < PACK_PF; ++q) {
          const int k = q * WAVE + lane;
          if (k < nnz) {
            const int64_t c = A.c[q];
            orc |= (uint64_t)c;
            lds_keys[wv][k] = ((unsigned long long)(uint32_t)c << 32) | (unsigned)k;
            lds_val[wv][k] = A.v[q];
          }
        }
        if (lane <= n) {
          const int32_t v = (int32_t)((uint32_t)A.rp - (uint32_t)(uint64_t)z0);
          rp_out[lane] = v;
          lds_rp[wv][lane] = v;
        }
      }
      if (t + 1 < cnt && __builtin_amdgcn_readlane(my_state, t + 1) == SMALL)
        pack_request(A, row_nnz_ptr, col_global, val, readlane_i64(my_r0, t + 1), readlane_i64(my_z0, t + 1),
                     __builtin_amdgcn_readlane(my_n, t + 1), __builtin_amdgcn_readlane(my_nnz, t + 1), lane);
      if (state != SMALL) {
        if (state == RANGE && lane == 0) { atomicExch(&stats->err, GDMIX_RE_ERANGE); d_cnt[e] = 0; }
        continue;
      }
      if (CAP > PACK_PF * WAVE) {   // the rest of a larger tier's entity, not requested ahead
        for (int k = PACK_PF * WAVE + lane; k < nnz; k += WAVE) {
          const int64_t c = col_global[z0 + k];
          orc |= (uint64_t)c;
          lds_keys[wv][k] = ((unsigned long long)(uint32_t)c << 32) | (unsigned)k;
          lds_val[wv][k] = val[z0 + k];
        }
      }
      for (int i = WAVE + lane; i <= n; i += WAVE) {
        const int32_t v = (int32_t)(row_nnz_ptr[r0 + i] - z0);
        rp_out[i] = v;
        lds_rp[wv][i] = v;
      }
      int d;
      {
        // explicit LDS instantiation (ds_* accesses; a generic pointer selecting between LDS and HBM compiles
        // to flat_* accesses whose base+offset folding faults at the LDS aperture edge)
        wave_lds_fence();
        static_assert((PACK_COUNT_COLS & (PACK_COUNT_COLS - 1)) == 0 && (PACK_BITMAP_COLS & (PACK_BITMAP_COLS - 1)) == 0, "thresholds of the OR");
        const bool wide = orc >= (uint64_t)PACK_COUNT_COLS, wide2 = orc >= (uint64_t)PACK_BITMAP_COLS;
        bad = orc > 0x7fffffffull;
        if (nnz <= PACK_RANK_MAX && pack_bitmap_on && __ballot(wide2) == 0ull) {
          d = bitmap_entity(lds_keys[wv], lds_bm[wv], reinterpret_cast<unsigned*>(lds_sorted[wv]), lds_val[wv], lds_rp[wv], n, nnz, lane, csr_col + z0, col_ptr + z0 + e,
                            csc_row + z0, csc_val + z0, uniq_sparse + z0);
        } else if (nnz <= PACK_RANK_MAX) {
          wave_rank_sort(lds_keys[wv], lds_sorted[wv], nnz, lane);
          d = emit_entity(lds_sorted[wv], lds_val[wv], lds_rp[wv], n, nnz, lane, csr_col + z0, col_ptr + z0 + e,
                          csc_row + z0, csc_val + z0, uniq_sparse + z0);
        } else if (__ballot(wide) == 0ull) {
          d = count_entity(lds_keys[wv], reinterpret_cast<unsigned*>(lds_sorted[wv]), lds_val[wv], lds_rp[wv], n, nnz, lane, csr_col + z0,
                           col_ptr + z0 + e, csc_row + z0, csc_val + z0, uniq_sparse + z0);
        } else {
          wave_bitonic_sort_lds(lds_keys[wv], nnz, lane);
          d = emit_entity(lds_keys[wv], lds_val[wv], lds_rp[wv], n, nnz, lane, csr_col + z0, col_ptr + z0 + e,
                          csc_row + z0, csc_val + z0, uniq_sparse + z0);
        }
      }
      if (__ballot(bad) && lane == 0) atomicExch(&stats->err, GDMIX_RE_ERANGE);
      if (lane == 0) d_cnt[e] = d;
      mx_p = max(mx_p, d + ic); mx_n = max(mx_n, n); mx_z = max(mx_z, nnz);
      wave_lds_fence();   // the LDS staging buffers are reused by the next entity of this wave (its stores are not waited for)
    }
  }
  if (lane == 0) {
    atomicMax(&blk_max[0], mx_p); atomicMax(&blk_max[1], mx_n); atomicMax(&blk_max[2], mx_z);
  }
  __syncthreads();
  // same-address atomics of thousands of workgroups ending together serialise in L2 (an empty later tier's 1 024 workgroups: 37 us of
  // nothing else); the maxima only grow, so a workgroup whose own are not above what is there already has nothing to publish
  if (threadIdx.x < 3) {
    int* const dst = threadIdx.x == 0 ? &stats->max_p : (threadIdx.x == 1 ? &stats->max_n : &stats->max_nnz);
    const int mine = blk_max[threadIdx.x];
    if (mine > __atomic_load_n(dst, __ATOMIC_RELAXED)) atomicMax(dst, mine);
  }
}

// unique_global[ent_feat_ptr[e] + l] = uniq_sparse[ent_nnz_ptr[e] + l]: the compact local -> global map.
// A wavefront takes 64 consecutive entities: their pointers in one coalesced load (one entity per lane), then entity by entity
// with the pointers broadcast from the lane that holds them, COMPACT_DEPTH entities' loads in flight before the first store (one
// entity per trip with its pointers loaded inside the trip was three dependent memory latencies per entity: 0.31 ms on C2; four in
// flight 0.21 ms: C2's entities have ~60 columns, 0.24 GB read + 0.48 GB written = 3.4 TB/s, what a copy of that shape reaches (round 6:
// the output is int32 like the input — 0.24 GB written; the array was int64 for the host's convenience only);
// eight in flight 0.31 ms, one entity per lane no better: profiles/r05_pack_ab.txt).
#ifndef GDMIX_COMPACT_DEPTH
#define GDMIX_COMPACT_DEPTH 4
#endif
constexpr int COMPACT_DEPTH = GDMIX_COMPACT_DEPTH;
__global__ __launch_bounds__(256) void pack_compact_unique_kernel(const int64_t* __restrict__ ent_nnz_ptr,
                                                                   const int64_t* __restrict__ ent_feat_ptr, int64_t E,
                                                                   const int32_t* __restrict__ uniq_sparse,
                                                                   int32_t* __restrict__ unique_global) {
  const int lane = threadIdx.x & (WAVE - 1);
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t e0 = (((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6) * WAVE; e0 < E; e0 += nwaves * WAVE) {
    const int64_t e = e0 + lane < E ? e0 + lane : E - 1;
    const int64_t my_z0 = ent_nnz_ptr[e], my_f0 = ent_feat_ptr[e];
    const int my_d = e0 + lane < E ? (int)(ent_feat_ptr[e + 1] - my_f0) : 0;
    const int cnt = (int)(E - e0 < WAVE ? E - e0 : WAVE);
    for (int k = 0; k < cnt; k += COMPACT_DEPTH) {
      int64_t z0[COMPACT_DEPTH], f0[COMPACT_DEPTH];
      int d[COMPACT_DEPTH], v[COMPACT_DEPTH];
#pragma unroll
      for (int q = 0; q < COMPACT_DEPTH; ++q) {
        const int src = k + q < WAVE ? k + q : WAVE - 1;
        z0[q] = __shfl(my_z0, src);
        f0[q] = __shfl(my_f0, src);
        d[q] = k + q < cnt ? __shfl(my_d, src) : 0;
        v[q] = lane < d[q] ? uniq_sparse[z0[q] + lane] : 0;
      }
#pragma unroll
      for (int q = 0; q < COMPACT_DEPTH; ++q) {
        if (lane < d[q]) unique_global[f0[q] + lane] = v[q];
        for (int l = lane + WAVE; l < d[q]; l += WAVE) unique_global[f0[q] + l] = uniq_sparse[z0[q] + l];   // p > 64: rare here
      }
    }
  }
}

// ---- exclusive scan int32 -> int64 over E+1 outputs (three small kernels) --------------------------
constexpr int SCAN_CHUNK = 2048;   // elements per workgroup (256 threads x 8)

__global__ __launch_bounds__(256) void scan_reduce_kernel(const int32_t* __restrict__ in, int64_t count,
                                                          long long* __restrict__ block_sums) {
  __shared__ long long part[4];
  const int64_t base = (int64_t)blockIdx.x * SCAN_CHUNK;
  long long s = 0;
  for (int k = threadIdx.x; k < SCAN_CHUNK; k += 256) {
    const int64_t i = base + k;
    if (i < count) s += in[i];
  }
  double sd = wave_sum((double)s);   // exact: per-wave partials stay far below 2^53
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = (long long)sd;
  __syncthreads();
  if (threadIdx.x == 0) block_sums[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

// exclusive scan of the per-chunk sums by one workgroup: every thread takes a contiguous run of chunks
__global__ __launch_bounds__(1024) void scan_blocksums_kernel(long long* __restrict__ block_sums, int nb, PackStats* __restrict__ stats) {
  __shared__ long long tsum[1024];
  const int tid = threadIdx.x;
  const int per = (nb + 1023) / 1024;
  const int b0 = tid * per, b1 = (b0 + per < nb) ? b0 + per : nb;
  long long mine = 0;
  for (int b = b0; b < b1; ++b) mine += block_sums[b];
  tsum[tid] = mine;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {   // Hillis-Steele inclusive scan
    const long long v = (tid >= off) ? tsum[tid - off] : 0;
    __syncthreads();
    tsum[tid] += v;
    __syncthreads();
  }
  long long run = tsum[tid] - mine;
  for (int b = b0; b < b1; ++b) { const long long v = block_sums[b]; block_sums[b] = run; run += v; }
  if (tid == 1023) stats->D = (unsigned long long)tsum[1023];
}

__global__ __launch_bounds__(256) void scan_apply_kernel(const int32_t* __restrict__ in, int64_t count,
                                                         const long long* __restrict__ block_sums,
                                                         int64_t* __restrict__ out /* [count+1] */) {
  __shared__ long long tsum[256];
  const int64_t base = (int64_t)blockIdx.x * SCAN_CHUNK + (int64_t)threadIdx.x * 8;
  long long v[8], s = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) { v[k] = (base + k < count) ? in[base + k] : 0; s += v[k]; }
  tsum[threadIdx.x] = s;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {   // Hillis-Steele inclusive scan of the thread sums
    long long add = (threadIdx.x >= off) ? tsum[threadIdx.x - off] : 0;
    __syncthreads();
    tsum[threadIdx.x] += add;
    __syncthreads();
  }
  long long run = block_sums[blockIdx.x] + tsum[threadIdx.x] - s;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    if (base + k < count) out[base + k] = run;
    run += v[k];
    if (base + k == count - 1) out[count] = run;
  }
}

// ---- host side ----------------------------------------------------------------------------------------
static size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

struct PackLayout {
  size_t ent_nnz_ptr, ent_feat_ptr, row_ptr, csr_col, col_ptr, csc_row, csc_val, unique_global, order, cls_tmp,
      d_cnt, class_count, block_sums, stats, sort_key, uniq_sparse, big_list, mid_list, mid2_list, total;
};

static PackLayout pack_layout(int64_t E, int64_t N, int64_t Z) {
  PackLayout L;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes); return o; };
  L.ent_nnz_ptr = take((size_t)(E + 1) * 8);
  L.ent_feat_ptr = take((size_t)(E + 1) * 8);
  L.row_ptr = take((size_t)(N + E + 1) * 4);
  L.csr_col = take((size_t)(Z + 1) * 4);
  L.col_ptr = take((size_t)(Z + E + 1) * 4);
  L.csc_row = take((size_t)(Z + 1) * 4);
  L.csc_val = take((size_t)(Z + 1) * 4);
  L.unique_global = take((size_t)(Z + 1) * 4);      // int32 since ABI 12 (global feature indices are below 2^31: INTEGRATION.md)
  L.order = take((size_t)(E + 1) * 4);
  L.cls_tmp = take((size_t)(E + 1) * 4);
  L.d_cnt = take((size_t)(E + 1) * 4);
  L.class_count = take(6 * GDMIX_RE_NUM_CLASSES * 4);
  L.block_sums = take((size_t)((E + 1) / SCAN_CHUNK + 2) * 8);
  L.stats = take(sizeof(PackStats));
  L.sort_key = take((size_t)(Z + 1) * 8);
  L.uniq_sparse = take((size_t)(Z + 1) * 4);
  L.big_list = take((size_t)(E + 1) * 4);
  L.mid_list = take((size_t)(E + 1) * 4);
  L.mid2_list = take((size_t)(E + 1) * 4);
  L.total = off;
  return L;
}

size_t pack_workspace_bytes(int64_t E, int64_t N, int64_t Z) { return pack_layout(E, N, Z).total; }

#define HIP_TRY(expr)                                                                   \
  do {                                                                                  \
    hipError_t _rc = (expr);                                                            \
    if (_rc != hipSuccess) {                                                            \
      set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_rc), __FILE__, __LINE__); \
      return GDMIX_RE_EHIP;                                                             \
    }                                                                                   \
  } while (0)

static bool debug_sync() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("GDMIX_RE_DEBUG_SYNC"); v = (e && e[0] == '1') ? 1 : 0; }
  return v == 1;
}
#define DBG_STAGE(name)                                                        \
  do {                                                                         \
    if (debug_sync()) {                                                        \
      hipError_t _e = hipStreamSynchronize(s);                                 \
      fprintf(stderr, "[gdmix_re] %s: %s\n", name, hipGetErrorString(_e));     \
      fflush(stderr);                                                          \
    }                                                                          \
  } while (0)

// GDMIX_PACK_BITMAP=0: entities of at most 128 non-zeros through the rank sort as before (A/B, and the tests that compare the paths)
static int pack_bitmap_on() {
  const char* e = getenv("GDMIX_PACK_BITMAP");
  return (e && atoi(e) == 0) ? 0 : 1;
}

int pack_impl(gdmix_ctx_impl* ctx, const gdmix_re_raw_batch* raw, int has_intercept, void* ws, size_t ws_bytes,
              gdmix_re_packed* out, hipStream_t s) {
  const int64_t E = raw->E, N = raw->N, Z = raw->Z;
  if (E < 0 || N < 0 || Z < 0) { set_error("negative batch dimension"); return GDMIX_RE_EINVAL; }
  if (E > 0x7ffffff0ll) { set_error("more than 2^31 entities in one batch"); return GDMIX_RE_ERANGE; }
  const PackLayout L = pack_layout(E, N, Z);
  if (ws_bytes < L.total) {
    set_error("pack workspace too small: %zu < %zu", ws_bytes, L.total);
    return GDMIX_RE_ENOMEM;
  }
  char* base = static_cast<char*>(ws);
  const int ic = has_intercept ? 1 : 0;
  out->E = E; out->N = N; out->Z = Z; out->D = 0;
  out->ent_row_ptr = raw->ent_row_ptr;
  out->ent_nnz_ptr = reinterpret_cast<int64_t*>(base + L.ent_nnz_ptr);
  out->ent_feat_ptr = reinterpret_cast<int64_t*>(base + L.ent_feat_ptr);
  out->row_ptr = reinterpret_cast<int32_t*>(base + L.row_ptr);
  out->csr_col = reinterpret_cast<int32_t*>(base + L.csr_col);
  out->csr_val = const_cast<float*>(raw->val);   // CSR order is the raw order: no copy
  out->col_ptr = reinterpret_cast<int32_t*>(base + L.col_ptr);
  out->csc_row = reinterpret_cast<int32_t*>(base + L.csc_row);
  out->csc_val = reinterpret_cast<float*>(base + L.csc_val);
  out->unique_global = reinterpret_cast<int32_t*>(base + L.unique_global);
  out->y = raw->y; out->offset = raw->offset; out->weight = raw->weight;
  out->order = reinterpret_cast<int32_t*>(base + L.order);
  out->class_count = reinterpret_cast<int32_t*>(base + L.class_count);
  out->cls_tmp = reinterpret_cast<int32_t*>(base + L.cls_tmp);
  out->scratch = base + L.sort_key;
  out->scratch_bytes = (size_t)(Z + 1) * 8;
  out->max_p = ic; out->max_n = 0; out->max_nnz = 0;
  if (E == 0) return GDMIX_RE_OK;

  int32_t* d_cnt = reinterpret_cast<int32_t*>(base + L.d_cnt);
  long long* block_sums = reinterpret_cast<long long*>(base + L.block_sums);
  PackStats* stats = reinterpret_cast<PackStats*>(base + L.stats);
  unsigned long long* sort_key = reinterpret_cast<unsigned long long*>(base + L.sort_key);

  HIP_TRY(hipMemsetAsync(stats, 0, sizeof(PackStats), s));
  int32_t* uniq_sparse = reinterpret_cast<int32_t*>(base + L.uniq_sparse);
  int32_t* big_list = reinterpret_cast<int32_t*>(base + L.big_list);
  int32_t* mid_list = reinterpret_cast<int32_t*>(base + L.mid_list);
  int32_t* mid2_list = reinterpret_cast<int32_t*>(base + L.mid2_list);
  {
    int grid = (int)((E + 1 + ENTNNZ_THREADS - 1) / ENTNNZ_THREADS);
    if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(pack_entnnz_kernel, dim3(grid), dim3(ENTNNZ_THREADS), 0, s, raw->ent_row_ptr, raw->row_nnz_ptr, E,
                       out->ent_nnz_ptr, mid_list, mid2_list, big_list, stats);
  }
  DBG_STAGE("pack_entnnz_kernel");
  // Four stages by entity size — wavefront + 256-key LDS staging (rank sort), + 512 keys, + 1 024 keys, and the device-wide sort
  // for what is larger. Their entity lists are complete (pack_entnnz_kernel) and their outputs disjoint, so they run next to each
  // other: the first on the caller's stream; the counts of the others are read back on side stream 0 meanwhile (their number and
  // size are only known on the device), and each stage that has entities gets a stream — side 1, side 2, and side 0 for the
  // device-wide sort. A batch of small entities only (C2) touches the caller's stream and side 0, like its solve: a context that
  // woke all its streams for every partition cost three concurrent contexts a third of their throughput (hand-over leg 77 -> 50 M
  // entities/s: more active streams than the device has hardware queues). With fewer side streams (GDMIX_RE_SIDE_STREAM) the
  // stages share what there is; with none they run one after another on the caller's stream, all launched blind as before round 4.
  SideJoin side_join{ctx, s};
  PackStats* hs = reinterpret_cast<PackStats*>(ctx->host_pinned);
  // ... and only for a batch small enough for the fixed costs to matter (a MovieLens share: 0.83 -> 0.62 ms): in a large one the
  // stages fill the device one after another anyway, and a pack that forks and joins streams costs contexts working side by side
  // (three of them on 1 M-entity partitions: 77 -> 52 M entities/s even with one side stream — cross-stream waits block hardware
  // queues the contexts share).
  const int ns = E <= PACK_FAN_MAX_ENTITIES ? ctx->n_side : 0;
  hipStream_t const sb = ns >= 1 ? ctx->side[0] : s;
  hipStream_t const s2 = ns >= 2 ? ctx->side[1] : (ns == 1 ? ctx->side[0] : s);
  hipStream_t const s3 = ns >= 3 ? ctx->side[2] : s2;
  SmallFetch counts;
  if (ns >= 1) {
    // the counts go to the host from the caller's stream, BEFORE the first tier is launched on it: queued on a side stream next to
    // that tier they arrived 100 us later (a one-wavefront kernel — and the runtime's copy kernel before it — sat for that long
    // behind the tier's 1 536 workgroups: tools/timeline_session.sh), and the other tiers and the large entities wait for them
    HIP_TRY(post_small(ctx, 3, stats, sizeof(PackStats), hs, s, &counts));
    HIP_TRY(hipEventRecord(ctx->side_fork, s));
    HIP_TRY(side_join.use(0));
  }
  // one round of workgroups, all resident (every workgroup walks the same number of entities: 16 per CU were 2.7 rounds of the 6 that
  // fit, the last one two-thirds full)
  static int pack_occupancy = 0;
  if (pack_occupancy == 0) {
    int occ = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, pack_entity_kernel<PACK_LDS_KEYS, PACK_WAVES>, WAVE * PACK_WAVES, 0) != hipSuccess || occ < 1) occ = 4;
    pack_occupancy = occ;
  }
  int eblocks = (int)((E + PACK_WAVES - 1) / PACK_WAVES);
  if (eblocks > ctx->num_cus * pack_occupancy) eblocks = ctx->num_cus * pack_occupancy;
  hipLaunchKernelGGL((pack_entity_kernel<PACK_LDS_KEYS, PACK_WAVES>), dim3(eblocks), dim3(WAVE * PACK_WAVES), 0, s,
                     (const int32_t*)nullptr, (const int*)nullptr, raw->ent_row_ptr,
                     raw->row_nnz_ptr, out->ent_nnz_ptr, raw->col_global, raw->val, E, ic, out->row_ptr, sort_key,
                     out->csr_col, out->col_ptr, out->csc_row, out->csc_val, uniq_sparse, d_cnt, stats, pack_bitmap_on());
  DBG_STAGE("pack_entity_kernel<256>");
  if (ns >= 1) HIP_TRY(wait_small(ctx, counts));   // the counts (the first stage is running)
  // (a C5-shaped entity has 256 +- 50 non-zeros: half of them overflow the first tier; the 512-key tier runs four workgroups of
  // four wavefronts per CU where the 1024-key one runs four of two, and sorts half as many keys)
  if (ns == 0 || hs->n_mid > 0) {
    if (ns >= 2) HIP_TRY(side_join.use(1));
    hipLaunchKernelGGL((pack_entity_kernel<PACK_CAP2, 4>), dim3(ctx->num_cus * 4), dim3(WAVE * 4), 0, s2,
                       (const int32_t*)mid_list, (const int*)&stats->n_mid, raw->ent_row_ptr,
                       raw->row_nnz_ptr, out->ent_nnz_ptr, raw->col_global, raw->val, E, ic, out->row_ptr, sort_key,
                       out->csr_col, out->col_ptr, out->csc_row, out->csc_val, uniq_sparse, d_cnt, stats, pack_bitmap_on());
  }
  DBG_STAGE("pack_entity_kernel<512>");
  if (ns == 0 || hs->n_mid2 > 0) {
    if (ns >= 2) HIP_TRY(side_join.use(ns >= 3 ? 2 : 1));
    hipLaunchKernelGGL((pack_entity_kernel<PACK_CAP3, 2>), dim3(ctx->num_cus * 4), dim3(WAVE * 2), 0, s3,
                       (const int32_t*)mid2_list, (const int*)&stats->n_mid2, raw->ent_row_ptr,
                       raw->row_nnz_ptr, out->ent_nnz_ptr, raw->col_global, raw->val, E, ic, out->row_ptr, sort_key,
                       out->csr_col, out->col_ptr, out->csc_row, out->csc_val, uniq_sparse, d_cnt, stats, pack_bitmap_on());
  }
  DBG_STAGE("pack_entity_kernel<1024>");
  {
    if (ns == 0) HIP_TRY(fetch_small(ctx, 1, stats, sizeof(PackStats), hs, s));
    if (hs->n_big > 0) {
      BigPackArgs A{raw->ent_row_ptr, out->ent_nnz_ptr, raw->row_nnz_ptr, raw->col_global, raw->val, ic, out->row_ptr, out->csr_col,
                    out->col_ptr, out->csc_row, out->csc_val, uniq_sparse, d_cnt, big_list, hs->n_big,
                    (int64_t)hs->big_nnz, &stats->max_p, &stats->err};
      // (its row table on the caller's stream, behind the first tier, next to its column passes on sb)
      const int rc = pack_big_entities(ctx, A, sb, ns >= 1 ? s : sb);
      if (rc != GDMIX_RE_OK) return rc;
    }
  }
  side_join.join();
  DBG_STAGE("pack_big_entities");
  const int nb = (int)((E + SCAN_CHUNK - 1) / SCAN_CHUNK);
  hipLaunchKernelGGL(scan_reduce_kernel, dim3(nb), dim3(256), 0, s, d_cnt, E, block_sums);
  hipLaunchKernelGGL(scan_blocksums_kernel, dim3(1), dim3(1024), 0, s, block_sums, nb, stats);
  hipLaunchKernelGGL(scan_apply_kernel, dim3(nb), dim3(256), 0, s, d_cnt, E, block_sums, out->ent_feat_ptr);
  DBG_STAGE("scan kernels");
  // The compaction is the one kernel of a pack that no solve kernel reads the output of (unique_global is for the model table and the
  // fixed-effect shard): with gdmix_re_set_defer_unique it runs on the last side stream, behind the scan, NEXT to the solve the caller
  // queues after this function returns — a copy-shaped kernel beside kernels bound by their arithmetic — and the statistics come back
  // without waiting for it. Whoever reads unique_global waits for `unique_ev` (join_unique).
  const bool defer = ctx->defer_unique && ctx->n_side > 0;
  hipStream_t const sc = defer ? ctx->side[ctx->n_side - 1] : s;
  SmallFetch tail_fetch;
  if (defer) {
    HIP_TRY(post_small(ctx, 1, stats, sizeof(PackStats), hs, s, &tail_fetch));
    HIP_TRY(hipEventRecord(ctx->side_fork, s));   // (this call's side streams have been joined above)
    HIP_TRY(hipStreamWaitEvent(sc, ctx->side_fork, 0));
  }
  {
    // deferred: a grid that is resident at once with room to spare — workgroups of a grid that is still being placed hold up the
    // placement of every other queue's kernels (the solve's first launches would wait for this kernel's last round instead of running
    // next to it: measured, the step did not move with the 16-per-CU grid)
    static int defer_wgs = 0;
    if (defer_wgs == 0) { const char* e = getenv("GDMIX_RE_DEFER_WGS"); defer_wgs = (e && atoi(e) > 0) ? atoi(e) : 4; }
    const int per_cu = defer ? defer_wgs : 16;
    int grid = (int)((E + 3) / 4);
    if (grid > ctx->num_cus * per_cu) grid = ctx->num_cus * per_cu;
    hipLaunchKernelGGL(pack_compact_unique_kernel, dim3(grid), dim3(256), 0, sc, out->ent_nnz_ptr, out->ent_feat_ptr, E,
                       uniq_sparse, out->unique_global);
  }
  DBG_STAGE("pack_compact_unique_kernel");
  HIP_TRY(hipGetLastError());
  if (defer) {
    HIP_TRY(hipEventRecord(ctx->unique_ev, sc));
    ctx->unique_pending = true;
    HIP_TRY(wait_small(ctx, tail_fetch));
  } else {
    HIP_TRY(fetch_small(ctx, 1, stats, sizeof(PackStats), hs, s));
  }
  if (hs->err) {
    set_error("pack: an entity exceeds a per-entity int32 limit or a feature index is outside [0, 2^31)");
    return hs->err;
  }
  out->D = (int64_t)hs->D;
  out->max_p = hs->max_p; out->max_n = hs->max_n; out->max_nnz = hs->max_nnz;
  return GDMIX_RE_OK;
}

// ---- B4: partition ids of decimal int64 entity ids ----------------------------------------------------
// id.toString for a Long is its decimal rendering with a leading '-' when negative; hashCode runs over
// the UTF-16 code units (ASCII here); Math.abs(Int.MinValue) stays Int.MinValue; % keeps the sign.
__global__ void partition_ids_kernel(const int64_t* __restrict__ ids, int64_t count, int32_t num_partitions,
                                     int32_t* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t v = ids[i];
    unsigned long long mag = v < 0 ? (unsigned long long)(-(v + 1)) + 1ull : (unsigned long long)v;
    char digits[20];
    int nd = 0;
    do { digits[nd++] = (char)('0' + (int)(mag % 10ull)); mag /= 10ull; } while (mag);
    uint32_t h = 0;
    if (v < 0) h = 31u * h + (uint32_t)'-';
    for (int k = nd - 1; k >= 0; --k) h = 31u * h + (uint32_t)digits[k];
    const int32_t hs = (int32_t)h;
    const int32_t a = (hs == INT32_MIN) ? hs : (hs < 0 ? -hs : hs);
    out[i] = (int32_t)((int64_t)a % (int64_t)num_partitions);
  }
}

hipError_t launch_partition_ids(const int64_t* ids, int64_t count, int32_t num_partitions, int32_t* out,
                                hipStream_t s) {
  if (count <= 0) return hipSuccess;
  int grid = (int)((count + 255) / 256);
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(partition_ids_kernel, dim3(grid), dim3(256), 0, s, ids, count, num_partitions, out);
  return hipGetLastError();
}

}  // namespace gdmix
