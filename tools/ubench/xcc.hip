#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(int* out) {
  unsigned x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  if (threadIdx.x == 0) out[blockIdx.x] = (int)(x & 0xf);
}
int main() {
  for (int threads : {256, 512})
    for (int grid : {80, 160, 320, 1000}) {
      int* d; hipMalloc(&d, grid * 4);
      hipLaunchKernelGGL(k, dim3(grid), dim3(threads), 65536, 0, d);
      std::vector<int> h(grid); hipMemcpy(h.data(), d, grid * 4, hipMemcpyDeviceToHost);
      int ok = 0; for (int i = 0; i < grid; ++i) ok += (h[i] == i % 8);
      printf("threads %d grid %d: blocks with xcc == bid%%8: %d/%d   first 24:", threads, grid, ok, grid);
      for (int i = 0; i < 24 && i < grid; ++i) printf(" %d", h[i]);
      printf("\n"); hipFree(d);
    }
}
