// re_wire.hip — the 32-bit hand-over form of an entity-grouped batch and its expansion on the device.
//
// What prepare_jobs (job_consumers.py:161-296) slices per entity reaches this library as the int64 pointer arrays of
// gdmix_re_raw_batch. Over PCIe that form is two thirds indices: int64 feature ids (the library itself requires them below
// 2^31), an int64 pointer per sample, float labels that are 0 or 1. The wire form carries counts instead of pointers,
// int32 feature ids and byte labels; gdmix_re_widen rebuilds the raw arrays in HBM (two prefix sums, two widening
// copies) in front of gdmix_re_pack. C2 (1 M entities, 16 M samples, 64 M non-zeros): 1.03 GB -> 0.62 GB on the wire.
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

constexpr int WIRE_CHUNK = 4096;   // counts per workgroup (256 threads x 16)

template <typename T>
__device__ __forceinline__ long long wire_count(const void* p, int64_t i) {
  return (long long)static_cast<const T*>(p)[i];
}

__device__ __forceinline__ long long wire_load(const void* p, int width, int64_t i) {
  return width == 1 ? wire_count<uint8_t>(p, i) : (width == 2 ? wire_count<uint16_t>(p, i) : wire_count<uint32_t>(p, i));
}

__global__ __launch_bounds__(256) void wire_reduce_kernel(const void* __restrict__ in, int width, int64_t count,
                                                          long long* __restrict__ block_sums) {
  __shared__ long long part[4];
  const int64_t base = (int64_t)blockIdx.x * WIRE_CHUNK;
  long long s = 0;
  for (int k = threadIdx.x; k < WIRE_CHUNK; k += 256) {
    const int64_t i = base + k;
    if (i < count) s += wire_load(in, width, i);
  }
  const double sd = wave_sum((double)s);   // exact: a chunk's total stays far below 2^53
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = (long long)sd;
  __syncthreads();
  if (threadIdx.x == 0) block_sums[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

// exclusive scan of the chunk sums by one workgroup; the grand total goes to total_out
__global__ __launch_bounds__(1024) void wire_blocksums_kernel(long long* __restrict__ block_sums, int nb,
                                                              long long* __restrict__ total_out) {
  __shared__ long long tsum[1024];
  const int tid = threadIdx.x;
  const int per = (nb + 1023) / 1024;
  const int b0 = tid * per, b1 = (b0 + per < nb) ? b0 + per : nb;
  long long mine = 0;
  for (int b = b0; b < b1; ++b) mine += block_sums[b];
  tsum[tid] = mine;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    const long long v = (tid >= off) ? tsum[tid - off] : 0;
    __syncthreads();
    tsum[tid] += v;
    __syncthreads();
  }
  long long run = tsum[tid] - mine;
  for (int b = b0; b < b1; ++b) { const long long v = block_sums[b]; block_sums[b] = run; run += v; }
  if (tid == 1023) *total_out = tsum[1023];
}

__global__ __launch_bounds__(256) void wire_apply_kernel(const void* __restrict__ in, int width, int64_t count,
                                                         const long long* __restrict__ block_sums,
                                                         int64_t* __restrict__ out /* [count + 1] */) {
  __shared__ long long tsum[256];
  const int64_t base = (int64_t)blockIdx.x * WIRE_CHUNK + (int64_t)threadIdx.x * 16;
  long long v[16], s = 0;
#pragma unroll
  for (int k = 0; k < 16; ++k) { v[k] = (base + k < count) ? wire_load(in, width, base + k) : 0; s += v[k]; }
  tsum[threadIdx.x] = s;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {
    const long long add = (threadIdx.x >= off) ? tsum[threadIdx.x - off] : 0;
    __syncthreads();
    tsum[threadIdx.x] += add;
    __syncthreads();
  }
  long long run = block_sums[blockIdx.x] + tsum[threadIdx.x] - s;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    if (base + k < count) out[base + k] = run;
    run += v[k];
    if (base + k == count - 1) out[count] = run;
  }
}

// col int32 -> int64, four per thread
__global__ __launch_bounds__(256) void wire_cols_kernel(const int32_t* __restrict__ in, int64_t count, int64_t* __restrict__ out) {
  const int64_t i4 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i4 + 4 <= count) {
    const int4 v = *reinterpret_cast<const int4*>(in + i4);
    longlong2 a, b;
    a.x = v.x; a.y = v.y; b.x = v.z; b.y = v.w;
    *reinterpret_cast<longlong2*>(out + i4) = a;
    *reinterpret_cast<longlong2*>(out + i4 + 2) = b;
  } else {
    for (int64_t i = i4; i < count; ++i) out[i] = in[i];
  }
}

// col uint16 -> int64, eight per thread
__global__ __launch_bounds__(256) void wire_cols16_kernel(const uint16_t* __restrict__ in, int64_t count, int64_t* __restrict__ out) {
  const int64_t i8 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 8;
  if (i8 + 8 <= count) {
    const uint4 v = *reinterpret_cast<const uint4*>(in + i8);
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      longlong2 a;
      a.x = (long long)(w[k] & 0xffffu);
      a.y = (long long)(w[k] >> 16);
      *reinterpret_cast<longlong2*>(out + i8 + 2 * k) = a;
    }
  } else {
    for (int64_t i = i8; i < count; ++i) out[i] = in[i];
  }
}

__global__ __launch_bounds__(256) void wire_labels_kernel(const uint8_t* __restrict__ in, int64_t count, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) out[i] = (float)in[i];
}

struct WireLayout { size_t ent_row_ptr, row_nnz_ptr, col_global, y, block_sums, totals, total; };

static WireLayout wire_layout(int64_t E, int64_t N, int64_t Z, int y_width) {
  WireLayout L;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
  L.ent_row_ptr = take((size_t)(E + 1) * 8);
  L.row_nnz_ptr = take((size_t)(N + 1) * 8);
  L.col_global = take((size_t)(Z + 8) * 8);
  L.y = take(y_width == 1 ? (size_t)(N + 1) * 4 : 0);
  const int64_t most = E > N ? E : N;
  L.block_sums = take((size_t)(most / WIRE_CHUNK + 2) * 8);
  L.totals = take(64);
  L.total = off;
  return L;
}

}  // namespace gdmix

using namespace gdmix;

extern "C" {

GDMIX_API size_t gdmix_re_widen_workspace_bytes(int64_t E, int64_t N, int64_t Z) {
  if (E < 0 || N < 0 || Z < 0) return 0;
  return wire_layout(E, N, Z, 1).total;
}

GDMIX_API int gdmix_re_widen(gdmix_re_ctx* ctx, const gdmix_re_wire_batch* w, void* workspace, size_t workspace_bytes,
                             gdmix_re_raw_batch* out, void* stream) {
  if (!ctx || !w || !out || (!workspace && workspace_bytes)) { set_error("NULL argument"); return GDMIX_RE_EINVAL; }
  const int64_t E = w->E, N = w->N, Z = w->Z;
  if (E < 0 || N < 0 || Z < 0) { set_error("negative batch dimension"); return GDMIX_RE_EINVAL; }
  if (w->row_nnz_width != 1 && w->row_nnz_width != 2 && w->row_nnz_width != 4) { set_error("row_nnz_width must be 1, 2 or 4"); return GDMIX_RE_EINVAL; }
  if (w->col_width != 2 && w->col_width != 4) { set_error("col_width must be 2 (uint16) or 4 (int32)"); return GDMIX_RE_EINVAL; }
  if (w->y_width != 1 && w->y_width != 4) { set_error("y_width must be 1 (uint8 labels) or 4 (float)"); return GDMIX_RE_EINVAL; }
  if ((E > 0 && !w->ent_n) || (N > 0 && (!w->row_nnz || !w->y || !w->offset)) || (Z > 0 && (!w->col_global || !w->val))) {
    set_error("wire batch has NULL arrays");
    return GDMIX_RE_EINVAL;
  }
  const WireLayout L = wire_layout(E, N, Z, w->y_width);
  if (workspace_bytes < L.total) { set_error("widen workspace too small: %zu < %zu", workspace_bytes, L.total); return GDMIX_RE_ENOMEM; }
  HIP_TRY(hipSetDevice(ctx->impl.device));
  hipStream_t s = static_cast<hipStream_t>(stream);
  char* base = static_cast<char*>(workspace);
  int64_t* ent_row_ptr = reinterpret_cast<int64_t*>(base + L.ent_row_ptr);
  int64_t* row_nnz_ptr = reinterpret_cast<int64_t*>(base + L.row_nnz_ptr);
  int64_t* col_global = reinterpret_cast<int64_t*>(base + L.col_global);
  long long* block_sums = reinterpret_cast<long long*>(base + L.block_sums);
  long long* totals = reinterpret_cast<long long*>(base + L.totals);
  out->E = E; out->N = N; out->Z = Z;
  out->ent_row_ptr = ent_row_ptr; out->row_nnz_ptr = row_nnz_ptr; out->col_global = col_global;
  out->val = w->val; out->offset = w->offset; out->weight = w->weight;
  out->y = w->y_width == 4 ? static_cast<const float*>(w->y) : reinterpret_cast<const float*>(base + L.y);
  HIP_TRY(hipMemsetAsync(totals, 0, 64, s));
  HIP_TRY(hipMemsetAsync(ent_row_ptr, 0, 8, s));   // E == 0 / N == 0: the single pointer entry
  HIP_TRY(hipMemsetAsync(row_nnz_ptr, 0, 8, s));
  if (E > 0) {
    const int nb = (int)((E + WIRE_CHUNK - 1) / WIRE_CHUNK);
    hipLaunchKernelGGL(wire_reduce_kernel, dim3(nb), dim3(256), 0, s, (const void*)w->ent_n, 4, E, block_sums);
    hipLaunchKernelGGL(wire_blocksums_kernel, dim3(1), dim3(1024), 0, s, block_sums, nb, totals);
    hipLaunchKernelGGL(wire_apply_kernel, dim3(nb), dim3(256), 0, s, (const void*)w->ent_n, 4, E, block_sums, ent_row_ptr);
  }
  if (N > 0) {
    const int nb = (int)((N + WIRE_CHUNK - 1) / WIRE_CHUNK);
    hipLaunchKernelGGL(wire_reduce_kernel, dim3(nb), dim3(256), 0, s, w->row_nnz, w->row_nnz_width, N, block_sums);
    hipLaunchKernelGGL(wire_blocksums_kernel, dim3(1), dim3(1024), 0, s, block_sums, nb, totals + 1);
    hipLaunchKernelGGL(wire_apply_kernel, dim3(nb), dim3(256), 0, s, w->row_nnz, w->row_nnz_width, N, block_sums, row_nnz_ptr);
    if (w->y_width == 1)
      hipLaunchKernelGGL(wire_labels_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, static_cast<const uint8_t*>(w->y), N,
                         reinterpret_cast<float*>(base + L.y));
  }
  if (Z > 0 && w->col_width == 4)
    hipLaunchKernelGGL(wire_cols_kernel, dim3((unsigned)((Z + 1023) / 1024)), dim3(256), 0, s, static_cast<const int32_t*>(w->col_global), Z, col_global);
  if (Z > 0 && w->col_width == 2)
    hipLaunchKernelGGL(wire_cols16_kernel, dim3((unsigned)((Z + 2047) / 2048)), dim3(256), 0, s, static_cast<const uint16_t*>(w->col_global), Z, col_global);
  HIP_TRY(hipGetLastError());
  // Same contract as the pointer arrays of a raw batch: the caller vouches that the counts add up to N and Z (the Python
  // binding checks it on the host, where the counts were made).
  return GDMIX_RE_OK;
}

}  // extern "C"
