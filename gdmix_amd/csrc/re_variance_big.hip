// re_variance_big.hip — FULL coefficient variances of entities too large for one wavefront (p > VAR_FULL_MAX_P):
// diag((X~' D X~ + (l2 + 1e-12) I - l2 e0 e0')^-1), binary_logistic_regression.py:181-187 (the reference densifies the Hessian
// and calls np.linalg.inv whatever p is). One entity at a time, the whole device on it:
//   vf_rows_kernel      d_i = w_i rho_i (1 - rho_i)
//   vf_build_kernel     H = X~' D X~ column by column: workgroup a spreads d .* column a over the samples and takes its
//                       products with the columns b >= a (ordered sums: deterministic), H symmetric, leading dimension padded to tiles
//   vf_potrf / vf_trsm / vf_syrk   right-looking Cholesky H = L L' on 64 x 64 tiles
//   vf_inverse_kernel   M = L^-1 one block column per workgroup (forward substitution over the row blocks), and
//                       diag(H^-1)_j = sum_i M_ij^2 on the way
// H is SPD (l2 > 0 or the 1e-12 ridge), so Cholesky replaces the reference's LU; both agree to rounding.
#include <vector>

#include "re_internal.hpp"
#include "re_device.hpp"

namespace gdmix {

constexpr int VF_T = 64;              // tile edge
constexpr int VF_LD = VF_T + 1;       // LDS row stride (bank conflicts)
constexpr int VF_THREADS = 256;

struct VfEntity {   // one large entity, offsets into the packed batch
  int64_t e, r0, z0, c0;
  int n, d, p, ld;  // ld = p rounded up to tiles
};

__global__ void vf_list_kernel(BatchDev B, int64_t E, int ic, int min_p, VfEntity* __restrict__ list, int32_t* __restrict__ count, int cap) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t f0 = B.ent_feat_ptr[e];
    const int d = (int)(B.ent_feat_ptr[e + 1] - f0);
    if (d + ic <= min_p) continue;
    const int slot = atomicAdd(count, 1);
    if (slot >= cap) continue;
    VfEntity v;
    v.e = e; v.r0 = B.ent_row_ptr[e]; v.z0 = B.ent_nnz_ptr[e]; v.c0 = f0 + e * ic;
    v.n = (int)(B.ent_row_ptr[e + 1] - v.r0); v.d = d; v.p = d + ic; v.ld = (v.p + VF_T - 1) / VF_T * VF_T;
    list[slot] = v;
  }
}

__global__ void vf_rows_kernel(BatchDev B, VfEntity V, int ic, const double* __restrict__ theta, double* __restrict__ dvec) {
  const double* th = theta + V.c0;
  const int32_t* rp = B.row_ptr + V.r0 + V.e;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < V.n; i += gridDim.x * blockDim.x) {
    double acc = ic ? th[0] : 0.0;
    for (int k = rp[i]; k < rp[i + 1]; ++k) acc += (double)B.csr_val[V.z0 + k] * th[ic + B.csr_col[V.z0 + k]];
    const double rho = sigmoid_full(acc + (double)B.offset[V.r0 + i]);
    dvec[i] = rho * (1.0 - rho) * (B.weight ? (double)B.weight[V.r0 + i] : 1.0);
  }
}

// H = identity on the padding, zero elsewhere
__global__ void vf_clear_kernel(double* __restrict__ H, int p, int ld) {
  const size_t total = (size_t)ld * ld;
  for (size_t a = (size_t)blockIdx.x * blockDim.x + threadIdx.x; a < total; a += (size_t)gridDim.x * blockDim.x) {
    const int r = (int)(a / ld), c = (int)(a % ld);
    H[a] = (r == c && r >= p) ? 1.0 : 0.0;
  }
}

// Column a of X~ times d, spread over the samples (w, all zero on entry and on exit), then its products with the columns b >= a.
// Repeated (row, column) pairs are summed first, as toarray() does.
__global__ __launch_bounds__(VF_THREADS) void vf_build_kernel(BatchDev B, VfEntity V, SolveParams o, const double* __restrict__ dvec,
                                                              double* __restrict__ wslots, double* __restrict__ H, int add_diag) {
  const int tid = threadIdx.x, lane = tid & (WAVE - 1), wv = tid >> 6;
  const int ic = o.has_intercept ? 1 : 0;
  double* w = wslots + (size_t)blockIdx.x * V.n;
  const int32_t* cp = B.col_ptr + V.z0 + V.e;
  const int32_t* crow = B.csc_row + V.z0;
  const float* cval = B.csc_val + V.z0;
  for (int a = blockIdx.x; a < V.p; a += gridDim.x) {
    const bool a_icpt = ic && a == 0;
    const int ka0 = a_icpt ? 0 : cp[a - ic], ka1 = a_icpt ? 0 : cp[a - ic + 1];
    if (a_icpt) {
      for (int i = tid; i < V.n; i += VF_THREADS) w[i] = dvec[i];
    } else {
      for (int k = ka0 + tid; k < ka1; k += VF_THREADS) {
        const int r = crow[k];
        if (k > ka0 && crow[k - 1] == r) continue;   // the first entry of a run of equal rows adds the run up
        double s = (double)cval[k];
        for (int k2 = k + 1; k2 < ka1 && crow[k2] == r; ++k2) s += (double)cval[k2];
        w[r] = s * dvec[r];
      }
    }
    __threadfence_block();
    __syncthreads();
    for (int b = a + wv; b < V.p; b += VF_THREADS / WAVE) {
      double s = 0.0;
      if (ic && b == 0) {
        for (int i = lane; i < V.n; i += WAVE) s += w[i];
      } else {
        const int k0 = cp[b - ic], k1 = cp[b - ic + 1];
        for (int k = k0 + lane; k < k1; k += WAVE) s += (double)cval[k] * w[crow[k]];
      }
      s = wave_sum(s);
      if (lane == 0) {
        if (b == a) {
          double add = o.l2 + 1.0e-12;
          if (a_icpt && !o.regularize_bias) add -= o.l2;
          H[(size_t)a * V.ld + a] = add_diag ? s + add : s;   // (the curvature part alone: gdmix_fe_hessian_dense)
        } else {
          H[(size_t)a * V.ld + b] = s;
          H[(size_t)b * V.ld + a] = s;
        }
      }
    }
    __syncthreads();
    if (a_icpt) {
      for (int i = tid; i < V.n; i += VF_THREADS) w[i] = 0.0;
    } else {
      for (int k = ka0 + tid; k < ka1; k += VF_THREADS) w[crow[k]] = 0.0;
    }
    __threadfence_block();
    __syncthreads();
  }
}

// ---- Cholesky on tiles -----------------------------------------------------------------------------------------------
__device__ __forceinline__ void tile_load(double (*dst)[VF_LD], const double* __restrict__ src, int ld) {
  for (int a = threadIdx.x; a < VF_T * VF_T; a += VF_THREADS) dst[a / VF_T][a % VF_T] = src[(size_t)(a / VF_T) * ld + (a % VF_T)];
}
__device__ __forceinline__ void tile_store(double* __restrict__ dst, int ld, const double (*src)[VF_LD]) {
  for (int a = threadIdx.x; a < VF_T * VF_T; a += VF_THREADS) dst[(size_t)(a / VF_T) * ld + (a % VF_T)] = src[a / VF_T][a % VF_T];
}

// diagonal tile k: A = L L' in place (lower; the upper part is zeroed)
__global__ __launch_bounds__(VF_THREADS) void vf_potrf_kernel(double* __restrict__ H, int ld, int k) {
  __shared__ double A[VF_T][VF_LD];
  double* tile = H + ((size_t)k * VF_T) * ld + (size_t)k * VF_T;
  tile_load(A, tile, ld);
  __syncthreads();
  const int tid = threadIdx.x;
  for (int j = 0; j < VF_T; ++j) {
    if (tid == 0) A[j][j] = sqrt(A[j][j]);
    __syncthreads();
    if (tid > j && tid < VF_T) A[tid][j] /= A[j][j];
    __syncthreads();
    // trailing update of the lower triangle: element (r, c), j < c <= r
    for (int a = tid; a < VF_T * VF_T; a += VF_THREADS) {
      const int r = a / VF_T, c = a % VF_T;
      if (c > j && r >= c) A[r][c] -= A[r][j] * A[c][j];
    }
    __syncthreads();
  }
  for (int a = tid; a < VF_T * VF_T; a += VF_THREADS) if (a % VF_T > a / VF_T) A[a / VF_T][a % VF_T] = 0.0;
  __syncthreads();
  tile_store(tile, ld, A);
}

// panel below the diagonal tile: A_ik <- A_ik L_kk^-T; one workgroup per row tile, one thread per row
__global__ __launch_bounds__(VF_T) void vf_trsm_kernel(double* __restrict__ H, int ld, int k) {
  __shared__ double L[VF_T][VF_LD], X[VF_T][VF_LD];
  const int i = k + 1 + blockIdx.x, r = threadIdx.x;
  const double* lk = H + ((size_t)k * VF_T) * ld + (size_t)k * VF_T;
  double* aik = H + ((size_t)i * VF_T) * ld + (size_t)k * VF_T;
  for (int a = r; a < VF_T * VF_T; a += VF_T) {
    L[a / VF_T][a % VF_T] = lk[(size_t)(a / VF_T) * ld + (a % VF_T)];
    X[a / VF_T][a % VF_T] = aik[(size_t)(a / VF_T) * ld + (a % VF_T)];
  }
  __syncthreads();
  for (int c = 0; c < VF_T; ++c) {
    double s = X[r][c];
    for (int j = 0; j < c; ++j) s -= X[r][j] * L[c][j];
    X[r][c] = s / L[c][c];
  }
  __syncthreads();
  for (int a = r; a < VF_T * VF_T; a += VF_T) aik[(size_t)(a / VF_T) * ld + (a % VF_T)] = X[a / VF_T][a % VF_T];
}

// C (64 x 64, 4 x 4 per thread) -= As * Bs with Bs[t][c]; As[r][t]
__device__ __forceinline__ void tile_mma(double (&C)[4][4], const double (*As)[VF_LD], const double (*Bs)[VF_LD], int tr, int tc) {
  for (int t = 0; t < VF_T; ++t) {
    double a[4], b[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { a[q] = As[tr * 4 + q][t]; b[q] = Bs[t][tc * 4 + q]; }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int u = 0; u < 4; ++u) C[q][u] -= a[q] * b[u];
  }
}

// trailing update: A_ij -= L_ik L_jk' for k < j <= i
__global__ __launch_bounds__(VF_THREADS) void vf_syrk_kernel(double* __restrict__ H, int ld, int k) {
  const int i = k + 1 + blockIdx.y, j = k + 1 + blockIdx.x;
  if (j > i) return;
  __shared__ double As[VF_T][VF_LD], Bs[VF_T][VF_LD];
  const int tid = threadIdx.x, tr = tid / 16, tc = tid % 16;
  const double* ai = H + ((size_t)i * VF_T) * ld + (size_t)k * VF_T;
  const double* aj = H + ((size_t)j * VF_T) * ld + (size_t)k * VF_T;
  for (int a = tid; a < VF_T * VF_T; a += VF_THREADS) {
    As[a / VF_T][a % VF_T] = ai[(size_t)(a / VF_T) * ld + (a % VF_T)];
    Bs[a % VF_T][a / VF_T] = aj[(size_t)(a / VF_T) * ld + (a % VF_T)];   // transposed: Bs[t][c] = L_jk[c][t]
  }
  __syncthreads();
  double* cij = H + ((size_t)i * VF_T) * ld + (size_t)j * VF_T;
  double C[4][4];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int u = 0; u < 4; ++u) C[q][u] = cij[(size_t)(tr * 4 + q) * ld + tc * 4 + u];
  tile_mma(C, As, Bs, tr, tc);
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int u = 0; u < 4; ++u) cij[(size_t)(tr * 4 + q) * ld + tc * 4 + u] = C[q][u];
}

// Block column c of M = L^-1: M_ic = L_ii^-1 (delta_ic I - sum_{k=c}^{i-1} L_ik M_kc), i = c .. T-1; the tiles M_kc this
// workgroup wrote are read back through L2 (ld_x<true>); variance_j = sum_i (M_ij)^2
__global__ __launch_bounds__(VF_THREADS) void vf_inverse_kernel(const double* __restrict__ H, double* __restrict__ M, int ld, int p,
                                                                double* __restrict__ variance) {
  __shared__ double As[VF_T][VF_LD], Bs[VF_T][VF_LD], S[VF_T][VF_LD];
  __shared__ double ss[VF_T];
  const int T = ld / VF_T, c = blockIdx.x;
  const int tid = threadIdx.x, tr = tid / 16, tc = tid % 16;
  if (tid < VF_T) ss[tid] = 0.0;
  for (int i = c; i < T; ++i) {
    double C[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int u = 0; u < 4; ++u) C[q][u] = (i == c && tr * 4 + q == tc * 4 + u) ? 1.0 : 0.0;
    for (int k = c; k < i; ++k) {
      const double* lik = H + ((size_t)i * VF_T) * ld + (size_t)k * VF_T;
      const double* mkc = M + ((size_t)k * VF_T) * ld + (size_t)c * VF_T;
      __syncthreads();
      for (int a = tid; a < VF_T * VF_T; a += VF_THREADS) {
        As[a / VF_T][a % VF_T] = lik[(size_t)(a / VF_T) * ld + (a % VF_T)];
        Bs[a / VF_T][a % VF_T] = ld_x<true>(mkc + (size_t)(a / VF_T) * ld + (a % VF_T));
      }
      __syncthreads();
      tile_mma(C, As, Bs, tr, tc);
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int u = 0; u < 4; ++u) S[tr * 4 + q][tc * 4 + u] = C[q][u];
    tile_load(As, H + ((size_t)i * VF_T) * ld + (size_t)i * VF_T, ld);   // L_ii
    __syncthreads();
    // forward substitution, one thread per column of the tile
    if (tid < VF_T) {
      double col2 = 0.0;
      for (int r = 0; r < VF_T; ++r) {
        double s = S[r][tid];
        for (int j = 0; j < r; ++j) s -= As[r][j] * S[j][tid];
        s /= As[r][r];
        S[r][tid] = s;
        col2 += s * s;
      }
      ss[tid] += col2;
    }
    __syncthreads();
    double* mic = M + ((size_t)i * VF_T) * ld + (size_t)c * VF_T;
    for (int a = tid; a < VF_T * VF_T; a += VF_THREADS) st_x<true>(mic + (size_t)(a / VF_T) * ld + (a % VF_T), S[a / VF_T][a % VF_T]);
    __threadfence();
  }
  __syncthreads();
  // the padding rows / columns are identity: a padded column's sum is 1 and is not written; a real column never meets them
  if (tid < VF_T && c * VF_T + tid < p) variance[c * VF_T + tid] = ss[tid];
}

size_t var_full_big_doubles(int64_t max_p, int64_t max_n) {
  const size_t ld = (size_t)(max_p + VF_T - 1) / VF_T * VF_T;
  return 2 * ld * ld + (size_t)max_n * (1 + VAR_BIG_BUILD_GROUPS) + 64;
}

// H (ld x ld, SPD, identity on the padding) -> diag(H^-1)[0, p) into `out`; H is overwritten by its Cholesky factor, M is ld x ld scratch
static hipError_t vf_factor_and_invert(double* H, double* M, int ld, int p, double* out, hipStream_t s) {
  const int T = ld / VF_T;
  for (int k = 0; k < T; ++k) {
    hipLaunchKernelGGL(vf_potrf_kernel, dim3(1), dim3(VF_THREADS), 0, s, H, ld, k);
    if (k + 1 < T) {
      hipLaunchKernelGGL(vf_trsm_kernel, dim3(T - k - 1), dim3(VF_T), 0, s, H, ld, k);
      hipLaunchKernelGGL(vf_syrk_kernel, dim3(T - k - 1, T - k - 1), dim3(VF_THREADS), 0, s, H, ld, k);
    }
  }
  hipLaunchKernelGGL(vf_inverse_kernel, dim3(T), dim3(VF_THREADS), 0, s, H, M, ld, p, out);
  return hipGetLastError();
}

hipError_t launch_variance_full_big(gdmix_ctx_impl* ci, const BatchDev& B, int64_t E, const SolveParams& o, const double* theta,
                                    double* variance, double* scratch, int64_t max_p, int64_t max_n, hipStream_t s) {
  // the large entities: counted first (cap 0: the list kernel only counts), then listed — however many there are
  const int ic = o.has_intercept ? 1 : 0;
  int32_t* count = nullptr;
  hipError_t rc = hipMalloc(&count, 64);
  if (rc != hipSuccess) return rc;
  int g = (int)((E + 255) / 256);
  if (g > 2048) g = 2048;
  int32_t n_big = 0;
  rc = hipMemsetAsync(count, 0, 4, s);
  if (rc == hipSuccess) hipLaunchKernelGGL(vf_list_kernel, dim3(g), dim3(256), 0, s, B, E, ic, (int)VAR_FULL_MAX_P, static_cast<VfEntity*>(nullptr), count, 0);
  if (rc == hipSuccess) rc = hipMemcpyAsync(&n_big, count, 4, hipMemcpyDeviceToHost, s);
  if (rc == hipSuccess) rc = hipStreamSynchronize(s);
  if (rc != hipSuccess || n_big == 0) { (void)hipFree(count); return rc; }
  VfEntity* list = nullptr;
  rc = hipMalloc(&list, sizeof(VfEntity) * (size_t)n_big);
  if (rc != hipSuccess) { (void)hipFree(count); return rc; }
  std::vector<VfEntity> host((size_t)n_big);
  rc = hipMemsetAsync(count, 0, 4, s);
  if (rc == hipSuccess) hipLaunchKernelGGL(vf_list_kernel, dim3(g), dim3(256), 0, s, B, E, ic, (int)VAR_FULL_MAX_P, list, count, n_big);
  if (rc == hipSuccess) rc = hipMemcpyAsync(host.data(), list, sizeof(VfEntity) * (size_t)n_big, hipMemcpyDeviceToHost, s);
  if (rc == hipSuccess) rc = hipStreamSynchronize(s);
  (void)hipFree(list);
  (void)hipFree(count);
  if (rc != hipSuccess) return rc;
  const size_t ldmax = (size_t)(max_p + VF_T - 1) / VF_T * VF_T;
  double* H = scratch;
  double* M = H + ldmax * ldmax;
  double* dvec = M + ldmax * ldmax;
  double* wslots = dvec + max_n;
  for (int q = 0; q < n_big; ++q) {
    const VfEntity& V = host[(size_t)q];
    hipLaunchKernelGGL(vf_rows_kernel, dim3((V.n + 255) / 256), dim3(256), 0, s, B, V, ic, theta, dvec);
    hipLaunchKernelGGL(vf_clear_kernel, dim3(ci->num_cus * 8), dim3(256), 0, s, H, V.p, V.ld);
    rc = hipMemsetAsync(wslots, 0, (size_t)V.n * VAR_BIG_BUILD_GROUPS * 8, s);
    if (rc != hipSuccess) return rc;
    hipLaunchKernelGGL(vf_build_kernel, dim3(VAR_BIG_BUILD_GROUPS), dim3(VF_THREADS), 0, s, B, V, o, dvec, wslots, H, 1);
    rc = vf_factor_and_invert(H, M, V.ld, V.p, variance + V.c0, s);
    if (rc != hipSuccess) return rc;
  }
  return hipSuccess;
}

// ---- two-stage form for several fixed-effect workers (include/gdmix_fe.h: gdmix_fe_hessian_dense / gdmix_fe_variance_of_hessian) ----
// Stage 1, per worker: the curvature part X~' D X~ of its shard (entity 0 of a one-entity packed batch) as a dense matrix in the
// shard's local index space, no regulariser. Stage 2, after the caller has scattered the matrices into the common index space and
// all-reduced them: regulariser on the diagonal, tiled Cholesky, diag of the inverse (fixed_effect_lr_lbfgs_model.py:296-305, 457-463).
size_t hessian_dense_scratch_doubles(int64_t n) { return (size_t)n * (1 + VAR_BIG_BUILD_GROUPS) + 64; }

hipError_t launch_hessian_dense(gdmix_ctx_impl* ci, const BatchDev& B, int64_t n, int64_t d, int ic, const double* theta, double* H, int64_t ld,
                                double* scratch, hipStream_t s) {
  VfEntity V;
  V.e = 0; V.r0 = 0; V.z0 = 0; V.c0 = 0;
  V.n = (int)n; V.d = (int)d; V.p = (int)d + ic; V.ld = (int)ld;
  if (V.ld < V.p || V.ld % VF_T) return hipErrorInvalidValue;
  SolveParams o{};
  o.has_intercept = ic;
  double* dvec = scratch;
  double* wslots = dvec + n;
  hipLaunchKernelGGL(vf_rows_kernel, dim3((V.n + 255) / 256), dim3(256), 0, s, B, V, ic, theta, dvec);
  hipLaunchKernelGGL(vf_clear_kernel, dim3(ci->num_cus * 8), dim3(256), 0, s, H, V.ld, V.ld);   // (p = ld: all zero, no identity on the padding)
  hipError_t rc = hipMemsetAsync(wslots, 0, (size_t)V.n * VAR_BIG_BUILD_GROUPS * 8, s);
  if (rc != hipSuccess) return rc;
  hipLaunchKernelGGL(vf_build_kernel, dim3(VAR_BIG_BUILD_GROUPS), dim3(VF_THREADS), 0, s, B, V, o, dvec, wslots, H, 0);
  return hipGetLastError();
}

// diagonal of a summed curvature matrix: + l2 + 1e-12 (without l2 at `unreg`), identity on the padding rows / columns
__global__ void vf_regularise_kernel(double* __restrict__ H, int p, int ld, double l2, int unreg) {
  const size_t total = (size_t)ld * ld;
  for (size_t a = (size_t)blockIdx.x * blockDim.x + threadIdx.x; a < total; a += (size_t)gridDim.x * blockDim.x) {
    const int r = (int)(a / ld), c = (int)(a % ld);
    if (r >= p || c >= p) H[a] = (r == c) ? 1.0 : 0.0;
    else if (r == c) H[a] += (r == unreg ? 0.0 : l2) + 1.0e-12;
  }
}

hipError_t launch_variance_of_hessian(gdmix_ctx_impl* ci, double* H, double* M, int64_t p, int64_t ld, double l2, int64_t unreg, double* variance,
                                      hipStream_t s) {
  if (ld < p || ld % VF_T || p < 1) return hipErrorInvalidValue;
  hipLaunchKernelGGL(vf_regularise_kernel, dim3(ci->num_cus * 8), dim3(256), 0, s, H, (int)p, (int)ld, l2, (int)unreg);
  return vf_factor_and_invert(H, M, (int)ld, (int)p, variance, s);
}

}  // namespace gdmix
