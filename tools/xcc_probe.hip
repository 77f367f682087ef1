// Probe: what HW_REG_XCC_ID returns per workgroup (MI355X: 8 XCDs), against blockIdx % 8. Exploration tool, not part of the library.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void probe(int* out) {
  if (threadIdx.x == 0) out[blockIdx.x] = (int)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);   // HW_REG_XCC_ID, bits [3:0]
}
int main() {
  const int n = 1024;
  int* d; hipMalloc(&d, n * sizeof(int));
  hipLaunchKernelGGL(probe, dim3(n), dim3(512), 0, 0, d);
  int h[n]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int hist[16] = {0}, agree = 0;
  for (int i = 0; i < n; ++i) { hist[h[i] & 15]++; agree += (h[i] == i % 8); }
  printf("xcc histogram:"); for (int i = 0; i < 16; ++i) printf(" %d", hist[i]); printf("\nblocks with xcc == blockIdx %% 8: %d of %d\nfirst 16:", agree, n);
  for (int i = 0; i < 16; ++i) printf(" %d", h[i]); printf("\n");
  return 0;
}
