// Calibrates clock64() (s_memtime) against wall_clock64() (s_memrealtime, 100 MHz) and against a dependent chain of VALU / MFMA instructions.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__global__ void k(long long* out, float* sink) {
  long long c0 = clock64(), w0 = wall_clock64();
  float x = threadIdx.x;
#pragma unroll 1
  for (int i = 0; i < 20000; ++i) asm volatile("v_add_f32 %0, %0, %0" : "+v"(x));
  long long c1 = clock64(), w1 = wall_clock64();
  f32x16 acc = {0};
  bf16x8 a = {1, 1, 1, 1, 1, 1, 1, 1};
#pragma unroll 1
  for (int i = 0; i < 2000; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, a, acc, 0, 0, 0);
  long long c2 = clock64(), w2 = wall_clock64();
  float y = threadIdx.x;
#pragma unroll 1
  for (int i = 0; i < 5000; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(y));
  long long c3 = clock64();
  if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; out[2] = c2 - c1; out[3] = w2 - w1; out[4] = c3 - c2; }
  sink[threadIdx.x] = x + acc[0] + y;
}
int main() {
  long long* d; float* s; hipMalloc(&d, 64); hipMalloc(&s, 1024);
  for (int r = 0; r < 3; ++r) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, s);
    long long h[5]; hipMemcpy(h, d, 40, hipMemcpyDeviceToHost);
    printf("20000 dependent v_add_f32: clock64 %lld ticks, wall(100MHz) %lld -> %.1f ns; per add %.2f ticks, %.2f ns | clock64 rate %.3f ticks/ns\n", h[0], h[1], h[1] * 10.0, h[0] / 20000.0, h[1] * 10.0 / 20000, h[0] / (h[1] * 10.0));
    printf("2000 dependent mfma 32x32x16 bf16: clock64 %lld, wall %.1f ns: per mfma %.2f ticks %.2f ns\n", h[2], h[3] * 10.0, h[2] / 2000.0, h[3] * 10.0 / 2000);
    printf("5000 dependent v_exp_f32: %.2f ticks each\n", h[4] / 5000.0);
  }
}
