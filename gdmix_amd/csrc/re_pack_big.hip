// re_pack_big.hip — pack of the entities too large for the wavefront pack kernels (more than 1024
// non-zeros or samples: the head of a Zipf-distributed partition, per-movie models, ...). Same outputs as
// pack_entity_kernel (re_pack.hip): local feature ids in order of the global index, CSR column ids, CSC
// copy with every column's entries in row-major order (the order scipy's COO mat-vec accumulates them
// in; job_consumers.py:209-258 slices, binary_logistic_regression.py:84-131 multiplies).
//
// All such entities are sorted together, device-wide: key = (rank in the big list << cbits) | global column with cbits = the
// bits of the largest column index among them (a radix pass per 8 key bits: MovieLens bags need 3 passes instead of the 6 of
// a 32-bit column field, a 65 536-feature space 4), value = position inside the entity. The input is in position order and the radix sort is stable, so
// equal columns stay in row-major order without the position being part of the key. The sort itself is the
// library's (rocPRIM radix_sort_pairs); everything around it is here.
//
// When the largest column index among them is below 2^BIG_COUNT_CBITS (MovieLens bags: 20 and 24 features; C3's per-movie entities
// are 48 M non-zeros of such columns) no sort is needed at all: a stable counting sort per entity over chunks of BIG_CH entries —
// histogram per chunk (big_hist_kernel), per entity the columns present, their local ids and every (column, chunk) bucket's first
// CSC position (big_cols_kernel), then one wavefront per chunk places its entries in position order (big_scatter_kernel). Same
// bytes out as the sort path (tests/test_gpu_parity.py::_check_pack on every fixture, both paths forced).
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "re_internal.hpp"

namespace gdmix {

#define HIP_TRY(expr)                                                                   \
  do {                                                                                  \
    hipError_t _rc = (expr);                                                            \
    if (_rc != hipSuccess) {                                                            \
      set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_rc), __FILE__, __LINE__); \
      return GDMIX_RE_EHIP;                                                             \
    }                                                                                   \
  } while (0)

constexpr int BIG_CH = 1024;               // entries per chunk: the unit of work of every per-entry pass here (4 096 until the end of round 4: a share of a
                                           // strongly scaled job has ~1 200 chunks of that size, one wavefront each for 120 us of big_scatter_kernel; 1 024: per-user share
                                           // pack 0.58 -> 0.51 ms, whole populations unchanged; 512 is worse again: tools/r04_bigch.sh)
constexpr unsigned BIG_COUNT_CBITS = 11;   // counting path for column indices below 2^11 (its tables are C counters per chunk)

// max statistics of a batch: thousands of entities report to one address, and the maximum only grows — read it first, and only the
// few that raise it pay for an atomic (same-address atomics serialise in L2, ~10 ns each)
__device__ __forceinline__ void raise_max(int* addr, int v) {
  if (v > __atomic_load_n(addr, __ATOMIC_RELAXED)) atomicMax(addr, v);
}

// sizes[b] = non-zeros of big entity b, nch[b] = its chunks (at least one, so that an entity without non-zeros still gets its
// pointers written); nch[n_big] = 0 closes the scan
__global__ void big_sizes_kernel(const int64_t* __restrict__ ent_nnz_ptr, const int32_t* __restrict__ big_list, int n_big,
                                 int64_t* __restrict__ sizes, int64_t* __restrict__ nch) {
  for (int b = blockIdx.x * blockDim.x + threadIdx.x; b <= n_big; b += gridDim.x * blockDim.x) {
    if (b == n_big) { nch[b] = 0; continue; }
    const int64_t e = big_list[b];
    const int64_t z = ent_nnz_ptr[e + 1] - ent_nnz_ptr[e];
    sizes[b] = z;
    nch[b] = z > 0 ? (z + BIG_CH - 1) / BIG_CH : 1;
  }
}

// The head of the device-wide pack in ONE launch, for up to BIG_HEAD_MAX big entities (round 5): their sizes, chunk counts and row counts
// and the three exclusive scans over them (offs, chunk0 with its closing total, row_offs), and the zero the column maximum starts
// from. It was seven launches of a few microseconds each (two per-entity kernels, three library scans of two launches, a memset):
// 0.1 ms of launch latency in front of a share's 0.45 ms pack (tools/timeline_session.sh). One workgroup, tiles of 1 024 entities.
constexpr int BIG_HEAD_MAX = 1 << 16;
__global__ __launch_bounds__(1024) void big_head_kernel(const int64_t* __restrict__ ent_nnz_ptr, const int64_t* __restrict__ ent_row_ptr,
                                                        const int32_t* __restrict__ big_list, int n_big, int64_t* __restrict__ sizes,
                                                        int64_t* __restrict__ offs, int64_t* __restrict__ nch, int64_t* __restrict__ chunk0,
                                                        int64_t* __restrict__ rows, int64_t* __restrict__ row_offs, unsigned* __restrict__ max_col) {
  __shared__ long long wsum[3][16];
  __shared__ long long carry[3];
  const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x >> 6;
  if (threadIdx.x == 0) { carry[0] = carry[1] = carry[2] = 0; *max_col = 0u; }
  __syncthreads();
  for (int b0 = 0; b0 <= n_big; b0 += 1024) {
    const int b = b0 + (int)threadIdx.x;
    long long v[3] = {0, 0, 0};
    if (b < n_big) {
      const int64_t e = big_list[b];
      const int64_t z = ent_nnz_ptr[e + 1] - ent_nnz_ptr[e];
      v[0] = z;
      v[1] = z > 0 ? (z + BIG_CH - 1) / BIG_CH : 1;
      v[2] = ent_row_ptr[e + 1] - ent_row_ptr[e] + 1;
      sizes[b] = v[0]; nch[b] = v[1]; rows[b] = v[2];
    } else if (b == n_big) {
      nch[b] = 0;
    }
    long long inc[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      long long x = v[k];
#pragma unroll
      for (int d = 1; d < WAVE; d <<= 1) {
        const long long y = __shfl_up(x, d);
        if (lane >= d) x += y;
      }
      inc[k] = x;
      if (lane == WAVE - 1) wsum[k][wv] = x;
    }
    __syncthreads();
    long long excl[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      long long base = carry[k];
      for (int w = 0; w < wv; ++w) base += wsum[k][w];
      excl[k] = base + inc[k] - v[k];
    }
    if (b < n_big) { offs[b] = excl[0]; chunk0[b] = excl[1]; row_offs[b] = excl[2]; }
    else if (b == n_big) chunk0[b] = excl[1];
    __syncthreads();
    if (threadIdx.x == 1023) {
#pragma unroll
      for (int k = 0; k < 3; ++k) carry[k] = excl[k] + v[k];
    }
    __syncthreads();
  }
}

// largest b with offs[b] <= i
__device__ __forceinline__ int big_find(const int64_t* __restrict__ offs, int n_big, int64_t i) {
  int lo = 0, hi = n_big - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (offs[mid] <= i) lo = mid; else hi = mid - 1;
  }
  return lo;
}

// Chunk q of the big entities' non-zeros: entity b (one bisection per workgroup instead of one per entry, which is what the
// per-entry passes used to spend their time on), first position inside the entity, length. chunk0 [n_big + 1] = exclusive scan of nch.
struct BigChunk { int b; int64_t e, z0, start; int len; };
// chunk_ent[q] = the big entity chunk q belongs to (big_chunk_ent_kernel, once per pack): every per-chunk pass used to find it by a
// bisection of chunk0 — 13 dependent loads in front of the chunk's first useful one, 8 of the ~11 us a workgroup spent per chunk
// (big_maxcol_kernel 144 us for 32 k chunks of a MovieLens population, tools/timeline_session.sh)
__device__ __forceinline__ BigChunk big_chunk(const int64_t* __restrict__ chunk0, const int32_t* __restrict__ chunk_ent, const int64_t* __restrict__ ent_nnz_ptr,
                                              const int32_t* __restrict__ big_list, int n_big, int64_t q) {
  BigChunk c;
  c.b = chunk_ent[q];
  c.e = big_list[c.b];
  c.z0 = ent_nnz_ptr[c.e];
  const int64_t z = ent_nnz_ptr[c.e + 1] - c.z0;
  c.start = (q - chunk0[c.b]) * BIG_CH;
  const int64_t left = z - c.start;
  c.len = (int)(left < BIG_CH ? (left > 0 ? left : 0) : BIG_CH);
  return c;
}

// chunk_ent: one wavefront per big entity writes its index over its chunks
__global__ __launch_bounds__(256) void big_chunk_ent_kernel(const int64_t* __restrict__ chunk0, int n_big, int32_t* __restrict__ chunk_ent) {
  const int lane = threadIdx.x & (WAVE - 1);
  const int64_t nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t b = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; b < n_big; b += nw) {
    const int64_t q1 = chunk0[b + 1];
    for (int64_t q = chunk0[b] + lane; q < q1; q += WAVE) chunk_ent[q] = (int32_t)b;
  }
}

// largest global column index among the big entities (and the range check of all of them)
__global__ __launch_bounds__(256) void big_maxcol_kernel(const int64_t* __restrict__ ent_nnz_ptr, const int64_t* __restrict__ col_global,
                                                         const int32_t* __restrict__ big_list, const int64_t* __restrict__ chunk0, const int32_t* __restrict__ chunk_ent, int n_big,
                                                         unsigned* __restrict__ max_col, int* __restrict__ err) {
  bool bad = false;
  unsigned mx = 0;
  const int64_t nq = chunk0[n_big];
  for (int64_t q = blockIdx.x; q < nq; q += gridDim.x) {
    const BigChunk k = big_chunk(chunk0, chunk_ent, ent_nnz_ptr, big_list, n_big, q);
    const int64_t* __restrict__ col = col_global + k.z0 + k.start;
    for (int i = threadIdx.x; i < k.len; i += 256) {
      const int64_t c = col[i];
      bad |= (c < 0 || c > 0x7fffffffll);
      mx = (!bad && (unsigned)c > mx) ? (unsigned)c : mx;
    }
  }
  __shared__ unsigned smx;
  if (threadIdx.x == 0) smx = 0;
  __syncthreads();
  atomicMax(&smx, mx);
  __syncthreads();
  // (one same-address atomic per workgroup was most of this kernel: 16 384 of them, 0.21 ms on a MovieLens population whose largest
  // column is 19; the maximum only grows, so a workgroup that cannot raise it has nothing to say)
  if (threadIdx.x == 0 && smx > __atomic_load_n(max_col, __ATOMIC_RELAXED)) atomicMax(max_col, smx);
  if (bad) atomicExch(err, GDMIX_RE_ERANGE);
}

__global__ __launch_bounds__(256) void big_fill_kernel(const int64_t* __restrict__ ent_nnz_ptr,
                                                       const int64_t* __restrict__ col_global,
                                                       const int32_t* __restrict__ big_list, const int64_t* __restrict__ chunk0, const int32_t* __restrict__ chunk_ent,
                                                       const int64_t* __restrict__ offs, int n_big, unsigned cbits,
                                                       unsigned long long* __restrict__ keys, uint32_t* __restrict__ vals) {
  const int64_t nq = chunk0[n_big];
  for (int64_t q = blockIdx.x; q < nq; q += gridDim.x) {
    const BigChunk k = big_chunk(chunk0, chunk_ent, ent_nnz_ptr, big_list, n_big, q);
    const int64_t* __restrict__ col = col_global + k.z0 + k.start;
    const int64_t o = offs[k.b] + k.start;
    for (int i = threadIdx.x; i < k.len; i += 256) {
      keys[o + i] = ((unsigned long long)(uint32_t)k.b << cbits) | ((unsigned long long)col[i] & ((1ull << cbits) - 1ull));
      vals[o + i] = (uint32_t)(k.start + i);
    }
  }
}

// ---- counting path ------------------------------------------------------------------------------------------------------
// hist[q][c] = entries of column c in chunk q (C = 2^cbits counters per chunk)
__global__ __launch_bounds__(256) void big_hist_kernel(const int64_t* __restrict__ ent_nnz_ptr, const int64_t* __restrict__ col_global,
                                                       const int32_t* __restrict__ big_list, const int64_t* __restrict__ chunk0, const int32_t* __restrict__ chunk_ent, int n_big,
                                                       int C, uint32_t* __restrict__ hist) {
  __shared__ unsigned h[1 << BIG_COUNT_CBITS];
  const int64_t nq = chunk0[n_big];
  for (int64_t q = blockIdx.x; q < nq; q += gridDim.x) {
    const BigChunk k = big_chunk(chunk0, chunk_ent, ent_nnz_ptr, big_list, n_big, q);
    for (int c = threadIdx.x; c < C; c += 256) h[c] = 0u;
    __syncthreads();
    const int64_t* __restrict__ col = col_global + k.z0 + k.start;
    for (int i = threadIdx.x; i < k.len; i += 256) atomicAdd(&h[(unsigned)col[i] & (unsigned)(C - 1)], 1u);
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) hist[q * C + c] = h[c];
    __syncthreads();
  }
}

// One workgroup per big entity: column totals over its chunks -> which columns are present, their local ids (rank among the
// present ones: the order of the global index), the column pointers, and in place of every chunk's count the CSC position of
// the chunk's first entry of that column. bits / lidbase [n_big][W], W = max(C / 32, 1): presence bitmap and the local id of a
// word's first column, from which the scatter pass derives the local id of any column.
__global__ __launch_bounds__(256) void big_cols_kernel(BigPackArgs a, const int64_t* __restrict__ chunk0, int C, uint32_t* __restrict__ hist,
                                                       uint32_t* __restrict__ bits, uint32_t* __restrict__ lidbase) {
  constexpr int CMAX = 1 << BIG_COUNT_CBITS, PER = CMAX / 256;
  __shared__ unsigned tot[CMAX], lid[CMAX], start[CMAX];
  __shared__ unsigned part_n[256], part_p[256];
  const int tid = threadIdx.x;
  for (int b = blockIdx.x; b < a.n_big; b += gridDim.x) {
    const int64_t e = a.big_list[b], z0 = a.ent_nnz_ptr[e];
    const int nnz = (int)(a.ent_nnz_ptr[e + 1] - z0);
    const int64_t q0 = chunk0[b], q1 = chunk0[b + 1];
    for (int c = tid; c < CMAX; c += 256) {
      unsigned t = 0;
      if (c < C)
        for (int64_t q = q0; q < q1; ++q) t += hist[q * C + c];
      tot[c] = t;
    }
    __syncthreads();
    // exclusive scans over the columns of the totals (-> first CSC position) and of the presence flags (-> local id): PER
    // consecutive columns per thread, the 256 thread sums by thread 0 .. 255 in order
    unsigned sn = 0, sp = 0;
#pragma unroll
    for (int i = 0; i < PER; ++i) { const unsigned t = tot[tid * PER + i]; sn += t; sp += t ? 1u : 0u; }
    part_n[tid] = sn; part_p[tid] = sp;
    __syncthreads();
    if (tid < WAVE) {   // 256 partial sums: four per lane, then a wavefront scan
      unsigned n4[4], p4[4], rn = 0, rp = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) { n4[i] = rn; p4[i] = rp; rn += part_n[tid * 4 + i]; rp += part_p[tid * 4 + i]; }
      unsigned xn = rn, xp = rp;
#pragma unroll
      for (int d = 1; d < WAVE; d <<= 1) {
        const unsigned yn = __shfl_up(xn, d), yp = __shfl_up(xp, d);
        if (tid >= d) { xn += yn; xp += yp; }
      }
      xn -= rn; xp -= rp;   // exclusive
#pragma unroll
      for (int i = 0; i < 4; ++i) { part_n[tid * 4 + i] = xn + n4[i]; part_p[tid * 4 + i] = xp + p4[i]; }
    }
    __syncthreads();
    unsigned rn = part_n[tid], rp = part_p[tid];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int c = tid * PER + i;
      const unsigned t = tot[c];
      start[c] = rn; lid[c] = rp;
      rn += t; rp += t ? 1u : 0u;
    }
    __syncthreads();
    const int d = (int)(lid[CMAX - 1] + (tot[CMAX - 1] ? 1u : 0u));
    int32_t* const cp = a.col_ptr + z0 + e;
    for (int c = tid; c < C; c += 256) {
      if (tot[c]) { a.uniq_sparse[z0 + lid[c]] = c; cp[lid[c]] = (int32_t)start[c]; }
      unsigned run = start[c];
      for (int64_t q = q0; q < q1; ++q) { const unsigned t = hist[q * C + c]; hist[q * C + c] = run; run += t; }
    }
    const int W = C >= 32 ? C / 32 : 1;
    for (int w = tid; w < W; w += 256) {
      unsigned m = 0;
      for (int i = 0; i < 32 && w * 32 + i < C; ++i) m |= (tot[w * 32 + i] ? 1u : 0u) << i;
      bits[(size_t)b * W + w] = m;
      lidbase[(size_t)b * W + w] = lid[w * 32];
    }
    if (tid == 0) {
      cp[d] = nnz;
      a.d_cnt[e] = d;
      raise_max(a.max_p, d + a.ic);
    }
    __syncthreads();
  }
}

// The same for C <= 64 columns (MovieLens bags: 32): one wavefront per entity, lane = column, four entities per workgroup — with
// tens of thousands of entities of a few chunks each a whole workgroup per entity is mostly launch and barriers.
__global__ __launch_bounds__(256) void big_cols_small_kernel(BigPackArgs a, const int64_t* __restrict__ chunk0, int C, uint32_t* __restrict__ hist,
                                                             uint32_t* __restrict__ bits, uint32_t* __restrict__ lidbase) {
  const int lane = threadIdx.x & (WAVE - 1);
  const int nw = (int)(gridDim.x * (blockDim.x >> 6));
  for (int b = (int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)); b < a.n_big; b += nw) {
    const int64_t e = a.big_list[b], z0 = a.ent_nnz_ptr[e];
    const int nnz = (int)(a.ent_nnz_ptr[e + 1] - z0);
    const int64_t q0 = chunk0[b], q1 = chunk0[b + 1];
    const bool col = lane < C;
    unsigned tot = 0;
    if (col)
      for (int64_t q = q0; q < q1; ++q) tot += hist[q * C + lane];
    // exclusive scans over the lanes: first CSC position and local id of the lane's column
    unsigned xn = tot, xp = tot ? 1u : 0u;
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
      const unsigned yn = __shfl_up(xn, d), yp = __shfl_up(xp, d);
      if (lane >= d) { xn += yn; xp += yp; }
    }
    const unsigned start = xn - tot, lid = xp - (tot ? 1u : 0u);
    const int d = (int)__shfl((int)xp, WAVE - 1);
    int32_t* const cp = a.col_ptr + z0 + e;
    if (tot) { a.uniq_sparse[z0 + lid] = lane; cp[lid] = (int32_t)start; }
    if (col) {
      unsigned run = start;
      for (int64_t q = q0; q < q1; ++q) { const unsigned t = hist[q * C + lane]; hist[q * C + lane] = run; run += t; }
    }
    const unsigned long long present = __ballot(tot != 0u);
    const int W = C >= 32 ? C / 32 : 1;
    if (lane < W) {
      bits[(size_t)b * W + lane] = (unsigned)(present >> (32 * lane));
      lidbase[(size_t)b * W + lane] = (unsigned)__popcll(present & ((1ull << (32 * lane)) - 1ull));
    }
    if (lane == 0) {
      cp[d] = nnz;
      a.d_cnt[e] = d;
      raise_max(a.max_p, d + a.ic);
    }
  }
}

// One wavefront per chunk: its entries, 64 at a time in position order, go to the CSC position base[column] + (entries of the
// same column before it in the tile); the first lane of every column of a tile then advances the base. In-order LDS, one
// wavefront: no atomics, and the CSC copy is in row-major order within a column, as the stable sort leaves it.
__global__ __launch_bounds__(WAVE) void big_scatter_kernel(BigPackArgs a, const int64_t* __restrict__ chunk0, const int32_t* __restrict__ chunk_ent, const int64_t* __restrict__ offs,
                                                           int C, const uint32_t* __restrict__ hist, const uint32_t* __restrict__ bits,
                                                           const uint32_t* __restrict__ lidbase, const uint32_t* __restrict__ row_of) {
  constexpr int CMAX = 1 << BIG_COUNT_CBITS;
  __shared__ unsigned base[CMAX], bm[CMAX / 32], lb[CMAX / 32];
  const int lane = threadIdx.x;
  const int W = C >= 32 ? C / 32 : 1;
  const int64_t nq = chunk0[a.n_big];
  for (int64_t q = blockIdx.x; q < nq; q += gridDim.x) {
    const BigChunk k = big_chunk(chunk0, chunk_ent, a.ent_nnz_ptr, a.big_list, a.n_big, q);
    for (int c = lane; c < C; c += WAVE) base[c] = hist[q * C + c];
    for (int w = lane; w < W; w += WAVE) { bm[w] = bits[(size_t)k.b * W + w]; lb[w] = lidbase[(size_t)k.b * W + w]; }
    wave_lds_fence();
    const int64_t* __restrict__ col = a.col_global + k.z0 + k.start;
    const float* __restrict__ val = a.val + k.z0 + k.start;
    const uint32_t* __restrict__ rof = row_of + offs[k.b] + k.start;
    for (int t = 0; t < k.len; t += WAVE) {
      const int i = t + lane;
      const bool valid = i < k.len;
      const unsigned c = valid ? ((unsigned)col[i] & (unsigned)(C - 1)) : 0u;
      const float v = valid ? val[i] : 0.0f;
      const uint32_t r = valid ? rof[i] : 0u;
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
        const unsigned lid = lb[c >> 5] + (unsigned)__popc(bm[c >> 5] & ((1u << (c & 31u)) - 1u));
        a.csr_col[k.z0 + k.start + i] = (int32_t)lid;
        a.csc_val[k.z0 + kpos] = v;
        a.csc_row[k.z0 + kpos] = (int32_t)r;
      }
      wave_lds_fence();
      if (valid && rank == 0) base[c] += cnt;
      wave_lds_fence();
    }
  }
}

// rows[b] = samples of big entity b + 1 (its row-pointer entries)
__global__ void big_rowcount_kernel(const int64_t* __restrict__ ent_row_ptr, const int32_t* __restrict__ big_list, int n_big,
                                    int64_t* __restrict__ rows) {
  for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < n_big; b += gridDim.x * blockDim.x) {
    const int64_t e = big_list[b];
    rows[b] = ent_row_ptr[e + 1] - ent_row_ptr[e] + 1;
  }
}

// One thread per row-pointer entry of the big entities: the entity-relative row pointers (which the wavefront pack kernels
// write for their own entities) and, for the emit pass, the row of every raw position.
__global__ __launch_bounds__(256) void big_rows_kernel(BigPackArgs a, const int64_t* __restrict__ offs,
                                                       const int64_t* __restrict__ row_offs, const int64_t* __restrict__ rows,
                                                       uint32_t* __restrict__ row_of) {
  const int nb = a.n_big;
  const int64_t total = row_offs[nb - 1] + rows[nb - 1];
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < total; r += (int64_t)gridDim.x * blockDim.x) {
    const int b = big_find(row_offs, nb, r);
    const int64_t i = r - row_offs[b];
    const int64_t e = a.big_list[b];
    const int64_t r0 = a.ent_row_ptr[e], z0 = a.ent_nnz_ptr[e];
    const int64_t n = a.ent_row_ptr[e + 1] - r0;
    const int64_t k0 = a.row_nnz_ptr[r0 + i] - z0;
    a.row_ptr[r0 + e + i] = (int32_t)k0;
    if (i == 0 && a.ent_nnz_ptr[e + 1] == z0) {   // samples but not a single non-zero: no entry reaches the passes that write these
      a.col_ptr[z0 + e] = 0;
      a.d_cnt[e] = 0;
      raise_max(a.max_p, a.ic);
    }
    if (i < n) {
      const int64_t k1 = a.row_nnz_ptr[r0 + i + 1] - z0;
      uint32_t* const dst = row_of + offs[b];
      for (int64_t k = k0; k < k1; ++k) dst[k] = (uint32_t)i;
    }
  }
}

// head[i] = 1 where a new (entity, column) run starts in the sorted keys
__global__ __launch_bounds__(256) void big_heads_kernel(const unsigned long long* __restrict__ keys, int64_t total,
                                                        int32_t* __restrict__ head) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
    head[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1 : 0;
}

__global__ __launch_bounds__(256) void big_emit_kernel(BigPackArgs a, const int64_t* __restrict__ offs, int64_t total, unsigned cbits,
                                                       const unsigned long long* __restrict__ keys,
                                                       const uint32_t* __restrict__ vals, const int32_t* __restrict__ head,
                                                       const int64_t* __restrict__ scan, const uint32_t* __restrict__ row_of) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const unsigned long long key = keys[i];
    const int b = (int)(key >> cbits);
    const int64_t e = a.big_list[b];
    const int64_t z0 = a.ent_nnz_ptr[e];
    const int nnz = (int)(a.ent_nnz_ptr[e + 1] - z0);
    const int64_t o = offs[b];
    const int k = (int)(i - o);             // position in the entity's CSC order
    const int pos = (int)vals[i];           // position in the entity's CSR (raw) order
    const int h = head[i];
    const int lid = (int)(scan[i] + h - 1 - scan[o]);   // runs started before or at i, minus one; the entity's first run is 0
    int32_t* const cp = a.col_ptr + z0 + e;
    if (h) { a.uniq_sparse[z0 + lid] = (int32_t)(key & ((1ull << cbits) - 1ull)); cp[lid] = k; }
    a.csr_col[z0 + pos] = lid;
    a.csc_val[z0 + k] = a.val[z0 + pos];
    a.csc_row[z0 + k] = (int32_t)row_of[o + pos];   // filled by big_rows_kernel (a bisection of the row pointers per entry before)
    if (k == nnz - 1) {
      const int d = lid + 1;
      cp[d] = nnz;
      a.d_cnt[e] = d;
      raise_max(a.max_p, d + a.ic);
    }
  }
}

static size_t up256(size_t x) { return (x + 255) & ~(size_t)255; }

int pack_big_entities(gdmix_ctx_impl* ctx, const BigPackArgs& a, hipStream_t s, hipStream_t aux) {
  const int64_t T = a.big_nnz;
  const int nb = a.n_big;
  if (nb <= 0) return GDMIX_RE_OK;
  unsigned ebits = 1;
  while ((1ll << ebits) < nb) ++ebits;
  const size_t TT = (size_t)(T > 0 ? T : 1);
  const int64_t max_chunks = T / BIG_CH + nb;          // every entity: floor(z / CH) + 1 chunks at most

  size_t sort_tmp = 0, scan_tmp = 0, scan2_tmp = 0;
  HIP_TRY((rocprim::radix_sort_pairs(nullptr, sort_tmp, (unsigned long long*)nullptr, (unsigned long long*)nullptr,
                                     (uint32_t*)nullptr, (uint32_t*)nullptr, TT, 0u, 32 + ebits, s)));
  HIP_TRY((rocprim::exclusive_scan(nullptr, scan_tmp, (int32_t*)nullptr, (int64_t*)nullptr, (int64_t)0, TT,
                                   rocprim::plus<int64_t>(), s)));
  HIP_TRY((rocprim::exclusive_scan(nullptr, scan2_tmp, (int64_t*)nullptr, (int64_t*)nullptr, (int64_t)0, (size_t)nb + 1,
                                   rocprim::plus<int64_t>(), s)));
  size_t lib_tmp = sort_tmp > scan_tmp ? sort_tmp : scan_tmp;
  if (scan2_tmp > lib_tmp) lib_tmp = scan2_tmp;

  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off = up256(off + bytes); return o; };
  // sort path: keys / values in and out, run heads and their scan. The counting path's tables (chunk histograms of at most
  // 2^BIG_COUNT_CBITS counters, presence bitmaps) live in the same stretch; it is sized for the larger of the two.
  const size_t sort_bytes = up256(TT * 8) * 2 + up256(TT * 4) * 3 + up256(TT * 8);
  const size_t W_MAX = (size_t)1 << (BIG_COUNT_CBITS - 5);
  const size_t count_bytes = up256((size_t)max_chunks * ((size_t)4 << BIG_COUNT_CBITS)) + 2 * up256((size_t)nb * W_MAX * 4) + up256(TT * 4);
  const size_t o_work = take(sort_bytes > count_bytes ? sort_bytes : count_bytes);
  const size_t o_sizes = take((size_t)(nb + 2) * 8), o_offs = take((size_t)(nb + 2) * 8);
  const size_t o_rows = take((size_t)(nb + 2) * 8), o_row_offs = take((size_t)(nb + 2) * 8);
  const size_t o_nch = take((size_t)(nb + 2) * 8), o_chunk0 = take((size_t)(nb + 2) * 8);
  const size_t o_lib = take(lib_tmp);
  const size_t o_chunk_ent = take((size_t)(max_chunks + 1) * 4);
  const size_t o_row_of = take(TT * 4);        // the row of every raw position: built before the path is known, on another stream
  if (ctx->big_tmp_bytes < off) {
    HIP_TRY(hipStreamSynchronize(s));
    if (ctx->big_tmp) { HIP_TRY(hipFree(ctx->big_tmp)); ctx->big_tmp = nullptr; ctx->big_tmp_bytes = 0; }
    const size_t want = off + off / 4;   // grow-only, with headroom for the next partition
    hipError_t rc = hipMalloc(&ctx->big_tmp, want);
    if (rc != hipSuccess) {
      set_error("pack: %d entities with %lld non-zeros need a %zu-byte device temporary: %s", nb, (long long)T, want,
                hipGetErrorString(rc));
      return GDMIX_RE_ENOMEM;
    }
    ctx->big_tmp_bytes = want;
  }
  char* base = static_cast<char*>(ctx->big_tmp);
  int64_t* sizes = reinterpret_cast<int64_t*>(base + o_sizes);
  int64_t* offs = reinterpret_cast<int64_t*>(base + o_offs);
  int64_t* rows = reinterpret_cast<int64_t*>(base + o_rows);
  int64_t* row_offs = reinterpret_cast<int64_t*>(base + o_row_offs);
  int64_t* nch = reinterpret_cast<int64_t*>(base + o_nch);
  int64_t* chunk0 = reinterpret_cast<int64_t*>(base + o_chunk0);
  void* lib = base + o_lib;

  unsigned* max_col_dev = reinterpret_cast<unsigned*>(sizes + nb + 1);   // the spare entry of `sizes`
  int grid = (nb + 1 + 255) / 256;
  size_t tmp = lib_tmp;
  const bool one_launch_head = nb <= BIG_HEAD_MAX;
  if (one_launch_head) {
    hipLaunchKernelGGL(big_head_kernel, dim3(1), dim3(1024), 0, s, a.ent_nnz_ptr, a.ent_row_ptr, a.big_list, nb, sizes, offs, nch, chunk0, rows, row_offs,
                       max_col_dev);
  } else {
    hipLaunchKernelGGL(big_sizes_kernel, dim3(grid), dim3(256), 0, s, a.ent_nnz_ptr, a.big_list, nb, sizes, nch);
    HIP_TRY((rocprim::exclusive_scan(lib, tmp, sizes, offs, (int64_t)0, (size_t)nb, rocprim::plus<int64_t>(), s)));
    tmp = lib_tmp;
    HIP_TRY((rocprim::exclusive_scan(lib, tmp, nch, chunk0, (int64_t)0, (size_t)nb + 1, rocprim::plus<int64_t>(), s)));
    hipLaunchKernelGGL(big_rowcount_kernel, dim3((nb + 255) / 256), dim3(256), 0, s, a.ent_row_ptr, a.big_list, nb, rows);
    tmp = lib_tmp;
    HIP_TRY((rocprim::exclusive_scan(lib, tmp, rows, row_offs, (int64_t)0, (size_t)nb, rocprim::plus<int64_t>(), s)));
    HIP_TRY(hipMemsetAsync(max_col_dev, 0, 4, s));
  }
  int32_t* const chunk_ent = reinterpret_cast<int32_t*>(base + o_chunk_ent);
  {
    const int g = (nb + 3) / 4 < ctx->num_cus * 8 ? (nb + 3) / 4 : ctx->num_cus * 8;
    hipLaunchKernelGGL(big_chunk_ent_kernel, dim3(g), dim3(256), 0, s, chunk0, nb, chunk_ent);
  }
  // row pointers of these entities and the row of every raw position (both paths). Nothing here depends on the columns: the table is
  // built on `aux` while this stream finds the largest column, waits for the host's decision and counts the columns (round 5: it was
  // 0.2 ms in the middle of the 0.8 ms chain a MovieLens pack waits for, tools/timeline_session.sh)
  const bool two_streams = aux != s && ctx->aux_ev[0] && ctx->aux_ev[1];
  // (the table lives behind everything either path carves out of the work area: its place does not depend on the path)
  uint32_t* const row_of = reinterpret_cast<uint32_t*>(base + o_row_of);
  if (two_streams) {
    HIP_TRY(hipEventRecord(ctx->aux_ev[0], s));
    HIP_TRY(hipStreamWaitEvent(aux, ctx->aux_ev[0], 0));
  }
  hipLaunchKernelGGL(big_rows_kernel, dim3(ctx->num_cus * 32), dim3(256), 0, two_streams ? aux : s, a, offs, row_offs, rows, row_of);
  if (two_streams) HIP_TRY(hipEventRecord(ctx->aux_ev[1], aux));
  // (`aux` is the caller's own stream when there are two: a way out of this function in between leaves nothing running that the caller's
  // stream does not order; `s` waits for the table right in front of the pass that reads it)
  auto rows_done = [&]() -> hipError_t { return two_streams ? hipStreamWaitEvent(s, ctx->aux_ev[1], 0) : hipSuccess; };
  // workgroups of the per-entry passes: one per chunk up to a few per CU's worth, strided beyond
  const int cgrid = (int)(max_chunks > (int64_t)ctx->num_cus * 64 ? (int64_t)ctx->num_cus * 64 : (max_chunks > 0 ? max_chunks : 1));
  // the bits of the column field: of the largest column index among these entities (one small read-back)
  hipLaunchKernelGGL(big_maxcol_kernel, dim3(cgrid), dim3(256), 0, s, a.ent_nnz_ptr, a.col_global, a.big_list, chunk0, chunk_ent, nb, max_col_dev, a.err);
  unsigned max_col = 0;
  HIP_TRY(fetch_small(ctx, 2, max_col_dev, 4, &max_col, s));
  unsigned cbits = 1;
  while (cbits < 32 && (max_col >> cbits) != 0u) ++cbits;
  bool counting = cbits <= BIG_COUNT_CBITS;
  if (const char* e = getenv("GDMIX_PACK_BIG_COUNT")) counting = counting && atoi(e) != 0;   // test hook: 0 = always the sort path
  if (counting) {
    const int C = 1 << cbits;
    const size_t W = C >= 32 ? (size_t)C / 32 : 1;
    size_t o = o_work;
    uint32_t* hist = reinterpret_cast<uint32_t*>(base + o); o += up256((size_t)max_chunks * (size_t)C * 4);
    uint32_t* bits = reinterpret_cast<uint32_t*>(base + o); o += up256((size_t)nb * W * 4);
    uint32_t* lidbase = reinterpret_cast<uint32_t*>(base + o); o += up256((size_t)nb * W * 4);
    hipLaunchKernelGGL(big_hist_kernel, dim3(cgrid), dim3(256), 0, s, a.ent_nnz_ptr, a.col_global, a.big_list, chunk0, chunk_ent, nb, C, hist);
    if (C <= WAVE) hipLaunchKernelGGL(big_cols_small_kernel, dim3((nb + 3) / 4 < ctx->num_cus * 32 ? (nb + 3) / 4 : ctx->num_cus * 32), dim3(256), 0, s, a, chunk0, C, hist, bits, lidbase);
    else hipLaunchKernelGGL(big_cols_kernel, dim3(nb < ctx->num_cus * 16 ? nb : ctx->num_cus * 16), dim3(256), 0, s, a, chunk0, C, hist, bits, lidbase);
    const int sgrid = (int)(max_chunks > (int64_t)ctx->num_cus * 256 ? (int64_t)ctx->num_cus * 256 : (max_chunks > 0 ? max_chunks : 1));
    HIP_TRY(rows_done());
    hipLaunchKernelGGL(big_scatter_kernel, dim3(sgrid), dim3(WAVE), 0, s, a, chunk0, chunk_ent, offs, C, hist, bits, lidbase, row_of);
    HIP_TRY(hipGetLastError());
    return GDMIX_RE_OK;
  }
  if (T <= 0) { HIP_TRY(rows_done()); return GDMIX_RE_OK; }
  size_t o = o_work;
  unsigned long long* keys_a = reinterpret_cast<unsigned long long*>(base + o); o += up256(TT * 8);
  unsigned long long* keys_b = reinterpret_cast<unsigned long long*>(base + o); o += up256(TT * 8);
  uint32_t* vals_a = reinterpret_cast<uint32_t*>(base + o); o += up256(TT * 4);
  uint32_t* vals_b = reinterpret_cast<uint32_t*>(base + o); o += up256(TT * 4);
  int32_t* head = reinterpret_cast<int32_t*>(base + o); o += up256(TT * 4);
  int64_t* scan = reinterpret_cast<int64_t*>(base + o);
  int64_t g64 = (T + 255) / 256;
  grid = (int)(g64 > ctx->num_cus * 32 ? ctx->num_cus * 32 : g64);
  const unsigned end_bit = cbits + ebits;
  hipLaunchKernelGGL(big_fill_kernel, dim3(cgrid), dim3(256), 0, s, a.ent_nnz_ptr, a.col_global, a.big_list, chunk0, chunk_ent, offs, nb, cbits,
                     keys_a, vals_a);
  tmp = lib_tmp;
  HIP_TRY((rocprim::radix_sort_pairs(lib, tmp, keys_a, keys_b, vals_a, vals_b, (size_t)T, 0u, end_bit, s)));
  hipLaunchKernelGGL(big_heads_kernel, dim3(grid), dim3(256), 0, s, keys_b, T, head);
  tmp = lib_tmp;
  HIP_TRY((rocprim::exclusive_scan(lib, tmp, head, scan, (int64_t)0, (size_t)T, rocprim::plus<int64_t>(), s)));
  HIP_TRY(rows_done());
  hipLaunchKernelGGL(big_emit_kernel, dim3(grid), dim3(256), 0, s, a, offs, T, cbits, keys_b, vals_b, head, scan, row_of);
  HIP_TRY(hipGetLastError());
  return GDMIX_RE_OK;
}

}  // namespace gdmix
