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

__global__ void big_sizes_kernel(const int64_t* __restrict__ ent_nnz_ptr, const int32_t* __restrict__ big_list, int n_big,
                                 int64_t* __restrict__ sizes) {
  for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < n_big; b += gridDim.x * blockDim.x) {
    const int64_t e = big_list[b];
    sizes[b] = ent_nnz_ptr[e + 1] - ent_nnz_ptr[e];
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

// largest global column index among the big entities (and the range check of all of them)
__global__ __launch_bounds__(256) void big_maxcol_kernel(const int64_t* __restrict__ ent_nnz_ptr, const int64_t* __restrict__ col_global,
                                                         const int32_t* __restrict__ big_list, const int64_t* __restrict__ offs, int n_big,
                                                         int64_t total, unsigned* __restrict__ max_col, int* __restrict__ err) {
  bool bad = false;
  unsigned mx = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int b = big_find(offs, n_big, i);
    const int64_t c = col_global[ent_nnz_ptr[big_list[b]] + (i - offs[b])];
    bad |= (c < 0 || c > 0x7fffffffll);
    mx = (!bad && (unsigned)c > mx) ? (unsigned)c : mx;
  }
  __shared__ unsigned smx;
  if (threadIdx.x == 0) smx = 0;
  __syncthreads();
  atomicMax(&smx, mx);
  __syncthreads();
  if (threadIdx.x == 0 && smx) atomicMax(max_col, smx);
  if (bad) atomicExch(err, GDMIX_RE_ERANGE);
}

__global__ __launch_bounds__(256) void big_fill_kernel(const int64_t* __restrict__ ent_nnz_ptr,
                                                       const int64_t* __restrict__ col_global,
                                                       const int32_t* __restrict__ big_list,
                                                       const int64_t* __restrict__ offs, int n_big, int64_t total, unsigned cbits,
                                                       unsigned long long* __restrict__ keys, uint32_t* __restrict__ vals) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int b = big_find(offs, n_big, i);
    const int64_t pos = i - offs[b];
    const int64_t c = col_global[ent_nnz_ptr[big_list[b]] + pos];
    keys[i] = ((unsigned long long)(uint32_t)b << cbits) | ((unsigned long long)c & ((1ull << cbits) - 1ull));
    vals[i] = (uint32_t)pos;
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
      atomicMax(a.max_p, d + a.ic);
    }
  }
}

static size_t up256(size_t x) { return (x + 255) & ~(size_t)255; }

int pack_big_entities(gdmix_ctx_impl* ctx, const BigPackArgs& a, hipStream_t s) {
  const int64_t T = a.big_nnz;
  const int nb = a.n_big;
  if (nb <= 0 || T <= 0) return GDMIX_RE_OK;
  unsigned ebits = 1;
  while ((1ll << ebits) < nb) ++ebits;

  size_t sort_tmp = 0, scan_tmp = 0, scan2_tmp = 0;
  HIP_TRY((rocprim::radix_sort_pairs(nullptr, sort_tmp, (unsigned long long*)nullptr, (unsigned long long*)nullptr,
                                     (uint32_t*)nullptr, (uint32_t*)nullptr, (size_t)T, 0u, 32 + ebits, s)));
  HIP_TRY((rocprim::exclusive_scan(nullptr, scan_tmp, (int32_t*)nullptr, (int64_t*)nullptr, (int64_t)0, (size_t)T,
                                   rocprim::plus<int64_t>(), s)));
  HIP_TRY((rocprim::exclusive_scan(nullptr, scan2_tmp, (int64_t*)nullptr, (int64_t*)nullptr, (int64_t)0, (size_t)nb,
                                   rocprim::plus<int64_t>(), s)));
  size_t lib_tmp = sort_tmp > scan_tmp ? sort_tmp : scan_tmp;
  if (scan2_tmp > lib_tmp) lib_tmp = scan2_tmp;

  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off = up256(off + bytes); return o; };
  const size_t o_keys_a = take((size_t)T * 8), o_keys_b = take((size_t)T * 8);
  const size_t o_vals_a = take((size_t)T * 4), o_vals_b = take((size_t)T * 4);
  const size_t o_head = take((size_t)T * 4), o_scan = take((size_t)T * 8);
  const size_t o_sizes = take((size_t)(nb + 1) * 8), o_offs = take((size_t)(nb + 1) * 8);
  const size_t o_rows = take((size_t)(nb + 1) * 8), o_row_offs = take((size_t)(nb + 1) * 8);
  const size_t o_lib = take(lib_tmp);
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
  unsigned long long* keys_a = reinterpret_cast<unsigned long long*>(base + o_keys_a);
  unsigned long long* keys_b = reinterpret_cast<unsigned long long*>(base + o_keys_b);
  uint32_t* vals_a = reinterpret_cast<uint32_t*>(base + o_vals_a);
  uint32_t* vals_b = reinterpret_cast<uint32_t*>(base + o_vals_b);
  int32_t* head = reinterpret_cast<int32_t*>(base + o_head);
  int64_t* scan = reinterpret_cast<int64_t*>(base + o_scan);
  int64_t* sizes = reinterpret_cast<int64_t*>(base + o_sizes);
  int64_t* offs = reinterpret_cast<int64_t*>(base + o_offs);
  int64_t* rows = reinterpret_cast<int64_t*>(base + o_rows);
  int64_t* row_offs = reinterpret_cast<int64_t*>(base + o_row_offs);
  void* lib = base + o_lib;

  int grid = (nb + 255) / 256;
  hipLaunchKernelGGL(big_sizes_kernel, dim3(grid), dim3(256), 0, s, a.ent_nnz_ptr, a.big_list, nb, sizes);
  size_t tmp = lib_tmp;
  HIP_TRY((rocprim::exclusive_scan(lib, tmp, sizes, offs, (int64_t)0, (size_t)nb, rocprim::plus<int64_t>(), s)));
  int64_t g64 = (T + 255) / 256;
  grid = (int)(g64 > ctx->num_cus * 32 ? ctx->num_cus * 32 : g64);
  // the bits of the column field: of the largest column index among these entities (one small read-back)
  unsigned* max_col_dev = reinterpret_cast<unsigned*>(sizes + nb);   // the spare entry of `sizes`
  HIP_TRY(hipMemsetAsync(max_col_dev, 0, 4, s));
  hipLaunchKernelGGL(big_maxcol_kernel, dim3(grid), dim3(256), 0, s, a.ent_nnz_ptr, a.col_global, a.big_list, offs, nb, T, max_col_dev, a.err);
  unsigned max_col = 0;
  HIP_TRY(hipMemcpyAsync(&max_col, max_col_dev, 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  unsigned cbits = 1;
  while (cbits < 32 && (max_col >> cbits) != 0u) ++cbits;
  const unsigned end_bit = cbits + ebits;
  hipLaunchKernelGGL(big_fill_kernel, dim3(grid), dim3(256), 0, s, a.ent_nnz_ptr, a.col_global, a.big_list, offs, nb, T, cbits,
                     keys_a, vals_a);
  tmp = lib_tmp;
  HIP_TRY((rocprim::radix_sort_pairs(lib, tmp, keys_a, keys_b, vals_a, vals_b, (size_t)T, 0u, end_bit, s)));
  hipLaunchKernelGGL(big_heads_kernel, dim3(grid), dim3(256), 0, s, keys_b, T, head);
  tmp = lib_tmp;
  HIP_TRY((rocprim::exclusive_scan(lib, tmp, head, scan, (int64_t)0, (size_t)T, rocprim::plus<int64_t>(), s)));
  // row pointers of these entities and the row of every raw position; the sort's input values are free again
  uint32_t* row_of = vals_a;
  hipLaunchKernelGGL(big_rowcount_kernel, dim3((nb + 255) / 256), dim3(256), 0, s, a.ent_row_ptr, a.big_list, nb, rows);
  tmp = lib_tmp;
  HIP_TRY((rocprim::exclusive_scan(lib, tmp, rows, row_offs, (int64_t)0, (size_t)nb, rocprim::plus<int64_t>(), s)));
  hipLaunchKernelGGL(big_rows_kernel, dim3(ctx->num_cus * 32), dim3(256), 0, s, a, offs, row_offs, rows, row_of);
  hipLaunchKernelGGL(big_emit_kernel, dim3(grid), dim3(256), 0, s, a, offs, T, cbits, keys_b, vals_b, head, scan, row_of);
  HIP_TRY(hipGetLastError());
  return GDMIX_RE_OK;
}

}  // namespace gdmix
