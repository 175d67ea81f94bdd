// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of the LoRA/TI training step.
// wave = 64 lanes everywhere; no other architecture is targeted.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short bf16_t;  // raw storage type used in signatures

#define SDLT_OK 0
#define SDLT_ERR_SHAPE (-1)
#define SDLT_ERR_ALIGN (-2)
#define SDLT_ERR_LAUNCH (-3)
#define SDLT_ERR_UNSUPPORTED (-4)

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// fp32 -> bf16, round-to-nearest-even, NaN stays a quiet NaN: the gfx950 hardware conversion (one v_cvt_pk_bf16_f32 per
// pair).  The software version (add 0x7fff + lsb, NaN branch) cost ~6 VALU instructions and an exec-mask branch per
// element and dominated the VALU time of the attention kernels' P / dS packing.
typedef __bf16 bf16x2_hw __attribute__((ext_vector_type(2)));
typedef float f32x2_hw __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
  f32x2_hw f = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, bf16x2_hw));
}
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(pack2bf(f, 0.f) & 0xffffu); }
__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }
// d/dx silu(x) = s + x*s*(1-s), s = sigmoid(x)
__device__ __forceinline__ float dsilu_f(float x) {
  float s = 1.0f / (1.0f + __expf(-x));
  return s * (1.0f + x * (1.0f - s));
}
// Exact-erf GELU (diffusers GEGLU, OpenCLIP-bigG's MLP) without the library erff (~40 VALU instructions, which made the GEGLU kernels
// VALU-bound: 1024 x 5120 gates x (gelu + gelu') = 8 of the backward kernel's 15 us): Phi(x) = 1 - 0.5 P(t) e^(-x^2/2) for x >= 0 and
// 0.5 P(t) e^(-x^2/2) for x < 0 with Abramowitz-Stegun 7.1.26 (t = 1 / (1 + p |x| / sqrt 2), |error of erf| <= 1.5e-7 - four orders below a
// bf16 ulp) - ONE exponential, shared with the density term of the derivative, one reciprocal and a degree-5 Horner chain.
__device__ __forceinline__ void gelu_parts(float x, float& phi, float& e) {
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.23164190f * fabsf(x));          // 0.3275911 / sqrt(2)
  e = __expf(-0.5f * x * x);
  const float q = 0.5f * e * t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  phi = x >= 0.f ? 1.0f - q : q;
}
__device__ __forceinline__ float gelu_f(float x) {
  float phi, e;
  gelu_parts(x, phi, e);
  return x * phi;
}
__device__ __forceinline__ float dgelu_f(float x) {
  float phi, e;
  gelu_parts(x, phi, e);
  return phi + x * 0.39894228040143268f * e;
}

// Row statistics of a folded LayerNorm from an MFMA operand fragment (8 bf16 of one row per lane): s1 += sum x, s2 += sum x^2 on the VALU's packed
// bf16 dot product (v_dot2c_f32_bf16: exact products, fp32 accumulation) - eight instructions that issue beside the MFMAs instead of two more MFMAs
__device__ __forceinline__ void ln_frag_stats(const bf16x8 x, float& s1, float& s2) {
  const bf16x2_hw one = {(__bf16)1.0f, (__bf16)1.0f};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const bf16x2_hw v = {x[2 * i], x[2 * i + 1]};
    s1 = __builtin_amdgcn_fdot2_f32_bf16(v, one, s1, false);
    s2 = __builtin_amdgcn_fdot2_f32_bf16(v, v, s2, false);
  }
}
// ... and their sum over the four 8-element chunks of a 32-wide K fragment row: lanes frow + 16 fk, fk = 0..3 (every lane gets the total)
__device__ __forceinline__ float ln_sum_fk(float v) {
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// async global -> LDS copy of 16 B per lane; LDS destination = lds_base (wave-uniform) + lane*16
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_base, 16, 0, 0);
}

// Zero-fill as an ordinary kernel.  hipMemsetAsync captured into a hipGraph (ROCm 7.0 runtime inside the PyTorch wheel)
// was observed to lose its ordering against the neighbouring kernel nodes on replay (GroupNorm statistics accumulated
// onto stale sums from the 2nd replay on), so every in-step zeroing goes through a kernel node instead.
static __global__ void sdlt_zero_kernel(uint32_t* p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 0u;
}
static inline void sdlt_zero_async(void* p, size_t bytes, hipStream_t s) {
  size_t n = bytes / 4;
  int blocks = (int)((n + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(sdlt_zero_kernel, dim3(blocks), dim3(256), 0, s, (uint32_t*)p, n);
}

void sdlt_set_error(const char* fmt, ...);
int sdlt_raise_smem(const void* fn, int bytes);   // capi.cpp: dynamic-LDS limit of a kernel, once per (kernel, device), thread-safe; 0 / -1
#define SDLT_FAIL(code, ...)      \
  do {                            \
    sdlt_set_error(__VA_ARGS__);  \
    return (code);                \
  } while (0)
#define SDLT_CHECK_LAUNCH()                                                      \
  do {                                                                           \
    hipError_t e_ = hipGetLastError();                                           \
    if (e_ != hipSuccess) SDLT_FAIL(SDLT_ERR_LAUNCH, "%s: %s", __func__, hipGetErrorString(e_)); \
  } while (0)

// a / b for 0 <= a < 2^22, 0 < b: reciprocal multiply + one correction step (~8 instructions; the ISA has no integer divide and the
// compiler's expansion is ~40 dependent instructions - the tiled GEMM's prologue did five of them before its first load was issued)
__device__ __forceinline__ int div_small(int a, int b) {
  int q = (int)((float)a * __builtin_amdgcn_rcpf((float)b));
  const int r = a - q * b;
  q += (r >= b ? 1 : 0) - (r < 0 ? 1 : 0);
  return q;
}
__device__ __forceinline__ int div_small_u(int a, int b) { return __builtin_amdgcn_readfirstlane(div_small(a, b)); }   // wave-uniform operands
