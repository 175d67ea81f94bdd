// Prodigy (prodigyopt 1.0, the reference's optional optimizer: trainer/optimizer.py:24-34 for the LoRA tensors,
// :135-145 for the token rows) as a device-resident step over a flat fp32 arena.
//
// The published algorithm makes two passes over the parameters with a global scalar update between them:
//   pass 1  exp_avg / exp_avg_sq / s moving averages (scaled by d), numerator += (d/d0) dlr <g, p0 - p>, denom = sum |s|
//   scalar  d_hat = d_coef numerator / denom ; d grows to at most d * growth_rate, never above the running max of d_hat
//   pass 2  p -= dlr (weight_decay p  +  exp_avg / (sqrt(exp_avg_sq) + d eps))      (decoupled decay)
// The reference implementation synchronises with the host twice per parameter tensor (.item()); here the two sums are
// fp64 device accumulators and the scalar update is a one-thread kernel, so the whole step stays inside the hipGraph.
// Both passes are HBM streams: pass 1 reads p,g,p0,m,v,s and writes m,v,s (9 words per element), pass 2 reads p,m,v and
// writes p (4 words).
#include "common.h"
#include "../../include/sdlt_kernels.h"

namespace {

enum { ST_D = 0, ST_D0, ST_DMAX, ST_NUM, ST_DENOM, ST_DHAT, ST_K, ST_DLR, ST_ACTIVE };
enum { HY_LR = 0, HY_B1, HY_B2, HY_B3, HY_EPS, HY_WD, HY_DCOEF, HY_GROWTH, HY_L1C, HY_GS, HY_BIASCORR, HY_SAFEGUARD, HY_DECOUPLE };

__global__ void prodigy_begin_kernel(const float* hyper, float* state, double* acc) {
  const float lr = hyper[HY_LR], b1 = hyper[HY_B1], b2 = hyper[HY_B2];
  const float k = state[ST_K];
  float bc = 1.f;
  if (hyper[HY_BIASCORR] != 0.f) bc = sqrtf(1.f - powf(b2, k + 1.f)) / (1.f - powf(b1, k + 1.f));
  state[ST_DLR] = state[ST_D] * lr * bc;
  acc[0] = 0.0;
  acc[1] = 0.0;
}

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__global__ void prodigy_accum_kernel(const float* p, const float* g, const float* p0, float* m, float* v, float* s, int64_t n,
                                     const float* hyper, const float* state, double* acc, float* l1_partial) {
  const float lr = hyper[HY_LR], b1 = hyper[HY_B1], b2 = hyper[HY_B2], b3 = hyper[HY_B3], wd = hyper[HY_WD],
              l1c = hyper[HY_L1C], gs = hyper[HY_GS];
  const bool coupled = hyper[HY_DECOUPLE] == 0.f && wd != 0.f;
  const float d = state[ST_D], d0 = state[ST_D0], dlr = state[ST_DLR];
  const float sa = hyper[HY_SAFEGUARD] != 0.f ? (d / d0) * d : (d / d0) * dlr;
  float dot = 0.f, den = 0.f, l1 = 0.f;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float pi = p[i];
    l1 += fabsf(pi);
    if (lr > 0.f) {
      float gi = g[i] * gs + l1c * (pi > 0.f ? 1.f : (pi < 0.f ? -1.f : 0.f));
      if (coupled) gi += wd * pi;
      dot += gi * (p0[i] - pi);
      m[i] = b1 * m[i] + d * (1.f - b1) * gi;
      v[i] = b2 * v[i] + d * d * (1.f - b2) * gi * gi;
      const float si = b3 * s[i] + sa * gi;
      s[i] = si;
      den += fabsf(si);
    }
  }
  // per-thread partial sums cover n / (grid * 256) elements in fp32; everything above that is accumulated in fp64
  // (one set of atomics per WORKGROUP: per wave they were up to 3 x 16384 serialised updates of three addresses - see adamw_kernel)
  __shared__ double accw[4][2];
  __shared__ float l1w[4];
  double dd = wave_sum_d((double)dot), de = wave_sum_d((double)den);
  l1 = wave_sum(l1);
  if ((threadIdx.x & 63) == 0) {
    accw[threadIdx.x >> 6][0] = dd;
    accw[threadIdx.x >> 6][1] = de;
    l1w[threadIdx.x >> 6] = l1;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0.0, b = 0.0;
    float t = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) { a += accw[w][0]; b += accw[w][1]; t += l1w[w]; }
    atomicAdd(&acc[0], a);
    atomicAdd(&acc[1], b);
    if (l1_partial) atomicAdd(l1_partial, t);
  }
}

__global__ void prodigy_scalar_kernel(const float* hyper, float* state, const double* acc) {
  const float lr = hyper[HY_LR], b3 = hyper[HY_B3];
  float d = state[ST_D], d_max = state[ST_DMAX];
  const float d0 = state[ST_D0], dlr = state[ST_DLR];
  const double den = acc[1];
  if (den == 0.0) {  // no progress possible (lr == 0 or all-zero gradients): the reference returns before touching any state
    state[ST_ACTIVE] = 0.f;
    return;
  }
  const double num = (double)state[ST_NUM] * b3 + (double)(d / d0) * dlr * acc[0];
  float d_hat = d;
  if (lr > 0.f) {
    d_hat = (float)(hyper[HY_DCOEF] * num / den);
    if (d == d0) d = fmaxf(d, d_hat);
    d_max = fmaxf(d_max, d_hat);
    d = fminf(d_max, d * hyper[HY_GROWTH]);
  }
  state[ST_NUM] = (float)num;
  state[ST_DENOM] = (float)den;
  state[ST_D] = d;
  state[ST_DMAX] = d_max;
  state[ST_DHAT] = d_hat;
  state[ST_K] += 1.f;
  state[ST_ACTIVE] = 1.f;
}

__global__ void prodigy_apply_kernel(float* p, const float* m, const float* v, int64_t n, const float* hyper, const float* state) {
  if (state[ST_ACTIVE] == 0.f) return;
  const float dlr = state[ST_DLR], de = state[ST_D] * hyper[HY_EPS];
  const float decay = hyper[HY_DECOUPLE] != 0.f ? 1.f - hyper[HY_WD] * dlr : 1.f;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    p[i] = p[i] * decay - dlr * m[i] / (sqrtf(v[i]) + de);
}

inline int grid_for(int64_t work_items, int block = 256) {
  int64_t g = (work_items + block - 1) / block;
  return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

}  // namespace

extern "C" int sdlt_prodigy_step(float* p, const float* g, const float* p0, float* m, float* v, float* s, int64_t n,
                                 const float* hyper, float* state, double* acc, float* l1_sum, void* stream) {
  if (n <= 0 || !p || !g || !p0 || !m || !v || !s || !hyper || !state || !acc)
    SDLT_FAIL(SDLT_ERR_SHAPE, "sdlt_prodigy_step: n=%lld or a null buffer", (long long)n);
  if ((uintptr_t)acc % 8) SDLT_FAIL(SDLT_ERR_ALIGN, "sdlt_prodigy_step: acc must be 8-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  if (l1_sum) sdlt_zero_async(l1_sum, sizeof(float), st);
  hipLaunchKernelGGL(prodigy_begin_kernel, dim3(1), dim3(1), 0, st, hyper, state, acc);
  hipLaunchKernelGGL(prodigy_accum_kernel, dim3(grid_for(n)), dim3(256), 0, st, p, g, p0, m, v, s, n, hyper, state, acc, l1_sum);
  hipLaunchKernelGGL(prodigy_scalar_kernel, dim3(1), dim3(1), 0, st, hyper, state, acc);
  hipLaunchKernelGGL(prodigy_apply_kernel, dim3(grid_for(n)), dim3(256), 0, st, p, m, v, n, hyper, state);
  SDLT_CHECK_LAUNCH();
  return SDLT_OK;
}
