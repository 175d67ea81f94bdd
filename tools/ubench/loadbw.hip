// Micro-benchmark: per-CU L2->CU load bandwidth on gfx950 for (a) global_load_dwordx4 into VGPRs and (b) global_load_lds
// (LDS-DMA), GEMM-like access (each wave instruction = 8 rows x 128 B), data L2-resident.  One workgroup per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned short bf16_t;
__device__ __forceinline__ void glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
template <int MODE>   // 0: VGPR loads, 1: LDS-DMA
__global__ void k(const char* buf, size_t bytes_per_wg, int iters, int row_stride, unsigned* sink, int shared) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const char* base = buf + (size_t)(shared ? (blockIdx.x % 8) : blockIdx.x) * bytes_per_wg;   // shared: one region per XCD (L2-resident)
  const int rows = bytes_per_wg / row_stride;          // rows of row_stride bytes; we read 128 B of each row per k-step
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (int it = 0; it < iters; ++it) {
    // one "k-step": every wave reads pieces (8 rows x 128 B) round-robin over the rows, column block it % (row_stride/128)
    const int cb = (it * 128) % row_stride;
    for (int p = wave; p < rows / 8; p += nw) {
      const char* src = base + (size_t)(p * 8 + (lane >> 3)) * row_stride + cb + (lane & 7) * 16;
      if (MODE == 0) {
        uint4 v = *(const uint4*)src;
        acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
      } else {
        glds16(src, smem + ((p % 128) * 1024));
      }
    }
    if (MODE == 1) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
  }
  if (MODE == 1) { __syncthreads(); acc.x = *(unsigned*)(smem + lane * 4); }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}
int main() {
  hipFuncSetAttribute((const void*)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072); hipFuncSetAttribute((const void*)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  const int NCU = 256;
  const int row_stride = 2560;                    // bytes (K = 1280 bf16)
  for (int rows : {128, 256, 512, 1024}) {
    size_t per_wg = (size_t)rows * row_stride;
    char* buf; unsigned* sink;
    hipMalloc(&buf, per_wg * NCU); hipMemset(buf, 1, per_wg * NCU); hipMalloc(&sink, 4);
    for (int shared = 1; shared < 2; ++shared)
    for (int mode = 0; mode < 2; ++mode)
      for (int threads : {256, 512, 1024}) {
        int iters = 400;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        auto launch = [&]() {
          if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(NCU), dim3(threads), 131072, 0, buf, per_wg, iters, row_stride, sink, shared);
          else hipLaunchKernelGGL(k<1>, dim3(NCU), dim3(threads), 131072, 0, buf, per_wg, iters, row_stride, sink, shared);
        };
        launch(); hipDeviceSynchronize();
        hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double bytes = (double)NCU * iters * rows * 128;
        printf("%s rows/WG %4d (%.0f KB/k-step/CU)  %s  %4d thr:  %.2f TB/s total, %.1f GB/s per CU, %.1f B/clk/CU @2.1GHz\n", shared ? "L2-shared" : "private  ", rows, rows * 128 / 1024.0,
               mode ? "LDS-DMA " : "VGPR    ", threads, bytes / ms / 1e9, bytes / ms / 1e6 / NCU, bytes / ms / 1e6 / NCU / 2.1);
      }
    hipFree(buf); hipFree(sink);
  }
  return 0;
}
