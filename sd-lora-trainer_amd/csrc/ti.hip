// Textual-inversion specific small kernels (gfx950): CLIP token+position embedding gather, the gradient of the
// trainable token rows, and the token-std regulariser.  All are launch/latency bound (n_tokens = 3 rows, 77 tokens).
#include "common.h"
#include "../../include/sdlt_kernels.h"

namespace {

// one workgroup per output row; 8 bf16 per lane
__global__ void embed_gather_kernel(const bf16_t* table, int64_t ldt, const int64_t* ids, const bf16_t* pos, int64_t ldp, int T, int Tp,
                                    int D, bf16_t* out, int64_t ldo) {
  const int row = blockIdx.x;            // b*Tp + t
  const int b = row / Tp, t = row - b * Tp;
  for (int c = threadIdx.x * 8; c < D; c += blockDim.x * 8) {
    uint4 o = make_uint4(0, 0, 0, 0);
    if (t < T) {
      const int64_t id = ids[(int64_t)b * T + t];
      uint4 a = *(const uint4*)(table + id * ldt + c);
      uint4 p = *(const uint4*)(pos + (int64_t)t * ldp + c);
      const uint32_t* ap = (const uint32_t*)&a;
      const uint32_t* pp = (const uint32_t*)&p;
      uint32_t* op = (uint32_t*)&o;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        op[j] = pack2bf(bf2f(ap[j] & 0xffff) + bf2f(pp[j] & 0xffff), bf2f(ap[j] >> 16) + bf2f(pp[j] >> 16));
    }
    *(uint4*)(out + (int64_t)row * ldo + c) = o;
  }
}

// grid (n_train); each workgroup scans the B*T token ids (deterministic order -> reproducible sums)
__global__ void embed_grad_kernel(const bf16_t* dx, int64_t lddx, const int64_t* ids, const int64_t* train_ids, int B, int T, int Tp, int D,
                                  float* grad, int accumulate) {
  const int j = blockIdx.x;
  const int64_t tid_ = train_ids[j];
  for (int c = threadIdx.x; c < D; c += blockDim.x) {
    float acc = accumulate ? grad[(int64_t)j * D + c] : 0.f;
    for (int b = 0; b < B; ++b)
      for (int t = 0; t < T; ++t)
        if (ids[(int64_t)b * T + t] == tid_) acc += bf2f(dx[((int64_t)b * Tp + t) * lddx + c]);
    grad[(int64_t)j * D + c] = acc;
  }
}

// one workgroup (256 threads) per row: mean, unbiased std, loss term and gradient
__global__ __launch_bounds__(256) void ti_std_reg_kernel(const float* rows, int n, int D, float tmean, float tvar, float w, float* grad,
                                                         float* loss_out) {
  __shared__ float sh[8];
  const int j = blockIdx.x;
  const float* r = rows + (int64_t)j * D;
  float s = 0.f;
  for (int c = threadIdx.x; c < D; c += 256) s += r[c];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  const float mean = (sh[0] + sh[1] + sh[2] + sh[3]) / (float)D;
  __syncthreads();
  float q = 0.f;
  for (int c = threadIdx.x; c < D; c += 256) { float d = r[c] - mean; q += d * d; }
  q = wave_sum(q);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = q;
  __syncthreads();
  const float var = (sh[0] + sh[1] + sh[2] + sh[3]) / (float)(D - 1);
  const float sd = sqrtf(var);
  // loss_j = w/n * (tmean - sd)^2 / tvar ;  d sd / d r_c = (r_c - mean) / ((D-1) * sd)
  const float coef = w / (float)n * 2.f * (sd - tmean) / tvar / ((float)(D - 1) * fmaxf(sd, 1e-20f));
  for (int c = threadIdx.x; c < D; c += 256) grad[(int64_t)j * D + c] += coef * (r[c] - mean);
  if (threadIdx.x == 0) atomicAdd(loss_out, w / (float)n * (tmean - sd) * (tmean - sd) / tvar);
}

}  // namespace

extern "C" int sdlt_embed_gather(const void* table, int64_t ld_table, const int64_t* ids, const void* pos, int64_t ld_pos, int32_t B,
                                 int32_t T, int32_t Tp, int32_t D, void* out, int64_t ldo, void* stream) {
  if (B <= 0 || T <= 0 || Tp < T || D <= 0 || (D % 8) || (ld_table % 8) || (ld_pos % 8) || (ldo % 8))
    SDLT_FAIL(SDLT_ERR_SHAPE, "sdlt_embed_gather: B=%d T=%d Tp=%d D=%d", B, T, Tp, D);
  hipLaunchKernelGGL(embed_gather_kernel, dim3(B * Tp), dim3(128), 0, (hipStream_t)stream, (const bf16_t*)table, ld_table, ids, (const bf16_t*)pos,
                     ld_pos, T, Tp, D, (bf16_t*)out, ldo);
  SDLT_CHECK_LAUNCH();
  return SDLT_OK;
}

extern "C" int sdlt_embed_grad(const void* dx, int64_t lddx, const int64_t* ids, const int64_t* train_ids, int32_t n_train, int32_t B, int32_t T,
                               int32_t Tp, int32_t D, float* grad, int32_t accumulate, void* stream) {
  if (n_train <= 0 || B <= 0 || T <= 0 || Tp < T || D <= 0) SDLT_FAIL(SDLT_ERR_SHAPE, "sdlt_embed_grad: n=%d B=%d T=%d D=%d", n_train, B, T, D);
  hipLaunchKernelGGL(embed_grad_kernel, dim3(n_train), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dx, lddx, ids, train_ids, B, T, Tp, D, grad,
                     accumulate);
  SDLT_CHECK_LAUNCH();
  return SDLT_OK;
}

extern "C" int sdlt_ti_std_reg(const float* rows, int32_t n, int32_t D, float target_mean, float target_var, float weight, float* grad,
                               float* loss_out, void* stream) {
  if (n <= 0 || D <= 1) SDLT_FAIL(SDLT_ERR_SHAPE, "sdlt_ti_std_reg: n=%d D=%d", n, D);
  hipLaunchKernelGGL(ti_std_reg_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, rows, n, D, target_mean, target_var, weight, grad, loss_out);
  SDLT_CHECK_LAUNCH();
  return SDLT_OK;
}
