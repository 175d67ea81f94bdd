// What does a dependent kernel boundary cost inside a replayed hipGraph on this box?  (the bound of a persistent per-block launch, measured before building one.)
// A chain of N dependent kernels of 256 workgroups x 256 threads is captured once and replayed; every kernel spins for `body` microseconds (s_memrealtime, 100 MHz) in every
// workgroup - body 0 = empty kernels.  Reported: microseconds per node, and what is left after subtracting the body = the price of the boundary (dispatch of the next grid,
// drain of the previous one, the release / acquire at the kernel boundary).  Second table: the same chain with the bodies fused into ONE launch of N phases separated by a
// grid-wide barrier (atomic counter, agent-scope release / acquire; 256 co-resident workgroups) - what "phase in launch" costs against "launch per phase".
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ __forceinline__ void spin_us(int us) {
  if (us <= 0) return;
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < (long long)us * 100) __builtin_amdgcn_s_sleep(2);
}
__global__ void node(int us, unsigned* sink) { spin_us(us); if (sink && threadIdx.x == 1024) sink[0] = 1; }
__global__ void fused(int us, int phases, unsigned* bar) {
  for (int ph = 0; ph < phases; ++ph) {
    spin_us(us);
    __syncthreads();
    if (threadIdx.x == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      atomicAdd(bar, 1u);
      const unsigned target = (unsigned)(ph + 1) * gridDim.x;
      while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
  }
}
int main() {
  hipStream_t s; hipStreamCreate(&s);
  unsigned* bar; hipMalloc(&bar, 4);
  const int N = 390;     // ~ the dependent launches of the thirty 1280-wide transformer blocks' forward at 13 per block
  for (int body : {0, 2, 5, 10, 20}) {
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(node, dim3(256), dim3(256), 0, s, body, (unsigned*)nullptr);
    hipStreamEndCapture(s, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) {
      hipEventRecord(e0, s); hipGraphLaunch(ge, s); hipEventRecord(e1, s); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); if (r && ms < best) best = ms;
    }
    float fbest = 1e9f;
    for (int r = 0; r < 5; ++r) {
      hipMemsetAsync(bar, 0, 4, s);
      hipEventRecord(e0, s); hipLaunchKernelGGL(fused, dim3(256), dim3(256), 0, s, body, N, bar); hipEventRecord(e1, s); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); if (r && ms < fbest) fbest = ms;
    }
    printf("body %2d us: %d dependent launches in a graph: %.2f us per node (boundary %.2f us)   |   one launch, %d phases with a grid barrier: %.2f us per phase (boundary %.2f us)\n",
           body, N, best * 1e3f / N, best * 1e3f / N - body, N, fbest * 1e3f / N, fbest * 1e3f / N - body);
  }
  return 0;
}
