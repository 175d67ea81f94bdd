// Support kernels of the full-UNet fine-tune (main.py:144-149: `unet.requires_grad_(True)`, every weight trained): the
// weight gradient dW[N,K] = sum_m dY[m,N] X[m,K] contracts over the TOKEN axis, which both operands store as their slow
// axis.  Both are transposed once into token-contiguous panels ([N,Mp] and [K,Mp], Mp = tokens rounded up to 64, zero
// filled) and the product runs on the MFMA GEMM of gemm.hip (dW = dY^T (X^T)^T, K loop over tokens, fp32 output straight
// into the gradient arena).  For the 3x3 convolutions the X panel is the transposed im2col: row (tap*Cin + ci), with
// out-of-image taps, stride 2 and the nearest-2x upsampled input resolved here.
//
// HBM-bound: 2 B read + 2 B written per element (x9 written for a 3x3 conv), 64x64 tiles through LDS so that reads and
// writes are both 128-B rows.  Also: d gamma / d beta of GroupNorm(+SiLU) and LayerNorm (column reductions over tokens).
#include "common.h"
#include "../../include/sdlt_kernels.h"

namespace {

struct GatherGeom {
  int conv;                                   // 0: plain rows, 1: 3x3 im2col
  int B, H, W, Hout, Wout, stride, ups;
};

// grid (Mp/64, ceil(C/64), taps); block 256.  out[(tap*C + c) * ldo + m] = x[src(m, tap) * ldx + c]
__global__ __launch_bounds__(256) void transpose_gather_kernel(const bf16_t* x, int64_t ldx, int M, int C, bf16_t* out,
                                                                int64_t ldo, GatherGeom g) {
  __shared__ bf16_t tile[64][64 + 2];
  const int m0 = blockIdx.x * 64, c0 = blockIdx.y * 64, tap = blockIdx.z;
  const int t = threadIdx.x;
  {
    const int r = t >> 2, cc = (t & 3) * 16;
    const int m = m0 + r;
    int64_t src = -1;
    if (m < M) {
      if (!g.conv) src = m;
      else {
        const int hw = g.Hout * g.Wout;
        const int b = m / hw, rem = m - b * hw, oy = rem / g.Wout, ox = rem - oy * g.Wout;
        const int iy = oy * g.stride + tap / 3 - 1, ix = ox * g.stride + tap % 3 - 1;
        if (iy >= 0 && iy < g.H * g.ups && ix >= 0 && ix < g.W * g.ups)
          src = ((int64_t)b * g.H + iy / g.ups) * g.W + ix / g.ups;
      }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int c = c0 + cc + h * 8;
      uint4 u = make_uint4(0, 0, 0, 0);
      if (src >= 0 && c + 8 <= C) u = *(const uint4*)(x + src * ldx + c);
      else if (src >= 0 && c < C) {           // ragged channel tail (C % 8 != 0 never happens for 16-B rows; C % 64 may)
        bf16_t tmp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int j = 0; j < 8 && c + j < C; ++j) tmp[j] = x[src * ldx + c + j];
        u = *(const uint4*)tmp;
      }
      const bf16_t* e = (const bf16_t*)&u;
#pragma unroll
      for (int j = 0; j < 8; ++j) tile[r][cc + h * 8 + j] = e[j];
    }
  }
  __syncthreads();
  {
    const int c = t >> 2, mm = (t & 3) * 16;
    if (c0 + c < C) {
      bf16_t v[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] = tile[mm + j][c];
      bf16_t* dst = out + ((int64_t)tap * C + c0 + c) * ldo + m0 + mm;
      *(uint4*)dst = *(const uint4*)&v[0];
      *(uint4*)(dst + 8) = *(const uint4*)&v[8];
    }
  }
}

// ------------------------------------------------------------------ d gamma / d beta
// grid (C/64, row chunks); block 256 = 64 channels x 4 row lanes.  MODE 0 LayerNorm (stats [M,2] = mean, rstd),
// MODE 1 GroupNorm(32) (stats [B,32,2] = sum, sumsq; optional SiLU on the output: dy is the gradient w.r.t. silu(y)).
struct CatIn2 {
  const bf16_t* x1; int64_t ld1; int C1;
  const bf16_t* x2; int64_t ld2;
  __device__ __forceinline__ float at(int64_t row, int c) const {
    return bf2f(c < C1 ? x1[row * ld1 + c] : x2[row * ld2 + (c - C1)]);
  }
};

template <int MODE>
__global__ __launch_bounds__(256) void norm_affine_grad_kernel(CatIn2 in, const bf16_t* dy, int64_t lddy, int64_t M, int HW, int C,
                                                                const float* stats, const float* gamma, const float* beta,
                                                                float eps, int silu, float* dgamma, float* dbeta) {
  __shared__ float red[2][4][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const int64_t per = (M + gridDim.y - 1) / gridDim.y;
  const int64_t r0 = blockIdx.y * per, r1 = (r0 + per < M) ? r0 + per : M;
  float dg = 0.f, db = 0.f;
  if (c < C) {
    const float ga = MODE == 1 ? gamma[c] : 1.f, be = MODE == 1 ? beta[c] : 0.f;
    const int grp = c / (C / 32);
    const float inv_n = 1.f / ((float)HW * (float)(C / 32));
    int64_t cur_b = -1;
    float mean = 0.f, rstd = 0.f;
    for (int64_t r = r0 + rl; r < r1; r += 4) {
      if (MODE == 0) {
        mean = stats[r * 2];
        rstd = stats[r * 2 + 1];
      } else {
        const int64_t b = r / HW;
        if (b != cur_b) {
          cur_b = b;
          mean = stats[(b * 32 + grp) * 2] * inv_n;
          const float var = fmaxf(stats[(b * 32 + grp) * 2 + 1] * inv_n - mean * mean, 0.f);
          rstd = rsqrtf(var + eps);
        }
      }
      const float xh = (in.at(r, c) - mean) * rstd;
      float d = bf2f(dy[r * lddy + c]);
      if (MODE == 1 && silu) d *= dsilu_f(xh * ga + be);
      dg += d * xh;
      db += d;
    }
  }
  red[0][rl][cl] = dg;
  red[1][rl][cl] = db;
  __syncthreads();
  if (rl == 0 && c < C) {
    atomicAdd(&dgamma[c], red[0][0][cl] + red[0][1][cl] + red[0][2][cl] + red[0][3][cl]);
    atomicAdd(&dbeta[c], red[1][0][cl] + red[1][1][cl] + red[1][2][cl] + red[1][3][cl]);
  }
}

int row_chunks(int64_t M, int cblocks) {
  int64_t want = 1024 / cblocks;                // ~4 workgroups per CU in total
  int64_t maxc = (M + 63) / 64;
  if (want < 1) want = 1;
  if (want > maxc) want = maxc;
  return (int)want;
}

}  // namespace

extern "C" int sdlt_wgrad_transpose(const void* x, int64_t ldx, int32_t M, int32_t C, void* out, int64_t ldo, int32_t Mp,
                                    void* stream) {
  if (M <= 0 || C <= 0 || Mp < M || (Mp % 64) || (ldx % 8) || (ldo % 8) || ldo < Mp)
    SDLT_FAIL(SDLT_ERR_SHAPE, "sdlt_wgrad_transpose: M=%d C=%d Mp=%d ldx=%lld ldo=%lld", M, C, Mp, (long long)ldx, (long long)ldo);
  GatherGeom g{};
  hipLaunchKernelGGL(transpose_gather_kernel, dim3(Mp / 64, (C + 63) / 64, 1), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x, ldx, M, C, (bf16_t*)out, ldo, g);
  SDLT_CHECK_LAUNCH();
  return SDLT_OK;
}

extern "C" int sdlt_wgrad_im2col_t(const void* x, int64_t ldx, int32_t B, int32_t H, int32_t W, int32_t C, int32_t stride,
                                   int32_t ups, void* out, int64_t ldo, int32_t Mp, void* stream) {
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || (stride != 1 && stride != 2) || (ups != 1 && ups != 2) || (stride == 2 && ups == 2))
    SDLT_FAIL(SDLT_ERR_SHAPE, "sdlt_wgrad_im2col_t: B=%d H=%d W=%d C=%d stride=%d ups=%d", B, H, W, C, stride, ups);
  GatherGeom g{1, B, H, W, H * ups / stride, W * ups / stride, stride, ups};
  const int64_t M = (int64_t)B * g.Hout * g.Wout;
  if (Mp < M || (Mp % 64) || (ldx % 8) || (ldo % 8) || ldo < Mp)
    SDLT_FAIL(SDLT_ERR_SHAPE, "sdlt_wgrad_im2col_t: M=%lld Mp=%d ldx=%lld ldo=%lld", (long long)M, Mp, (long long)ldx, (long long)ldo);
  hipLaunchKernelGGL(transpose_gather_kernel, dim3(Mp / 64, (C + 63) / 64, 9), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x, ldx, (int)M, C, (bf16_t*)out, ldo, g);
  SDLT_CHECK_LAUNCH();
  return SDLT_OK;
}

extern "C" int sdlt_layernorm_affine_grad(const void* x, int64_t ldx, const void* dy, int64_t lddy, int32_t M, int32_t C,
                                          const float* stats, float* dgamma, float* dbeta, void* stream) {
  if (M <= 0 || C <= 0 || !stats || !dgamma || !dbeta) SDLT_FAIL(SDLT_ERR_SHAPE, "sdlt_layernorm_affine_grad: M=%d C=%d", M, C);
  hipStream_t s = (hipStream_t)stream;
  sdlt_zero_async(dgamma, sizeof(float) * C, s);
  sdlt_zero_async(dbeta, sizeof(float) * C, s);
  CatIn2 in{(const bf16_t*)x, ldx, C, nullptr, 0};
  const int cb = (C + 63) / 64;
  hipLaunchKernelGGL(norm_affine_grad_kernel<0>, dim3(cb, row_chunks(M, cb)), dim3(256), 0, s, in, (const bf16_t*)dy, lddy,
                     (int64_t)M, 1, C, stats, nullptr, nullptr, 0.f, 0, dgamma, dbeta);
  SDLT_CHECK_LAUNCH();
  return SDLT_OK;
}

extern "C" int sdlt_groupnorm_affine_grad(const sdlt_groupnorm_params* pp, float* dgamma, float* dbeta, void* stream) {
  const sdlt_groupnorm_params& p = *pp;
  if (p.B <= 0 || p.HW <= 0 || p.C <= 0 || (p.C % 32) || !p.stats || !p.dy || !dgamma || !dbeta)
    SDLT_FAIL(SDLT_ERR_SHAPE, "sdlt_groupnorm_affine_grad: B=%d HW=%d C=%d", p.B, p.HW, p.C);
  hipStream_t s = (hipStream_t)stream;
  sdlt_zero_async(dgamma, sizeof(float) * p.C, s);
  sdlt_zero_async(dbeta, sizeof(float) * p.C, s);
  CatIn2 in{(const bf16_t*)p.x1, p.ldx1, p.x2 ? p.C1 : p.C, (const bf16_t*)p.x2, p.ldx2};
  const int cb = (p.C + 63) / 64;
  const int64_t M = (int64_t)p.B * p.HW;
  hipLaunchKernelGGL(norm_affine_grad_kernel<1>, dim3(cb, row_chunks(M, cb)), dim3(256), 0, s, in, (const bf16_t*)p.dy, p.lddy,
                     M, p.HW, p.C, p.stats, p.gamma, p.beta, p.eps, p.silu, dgamma, dbeta);
  SDLT_CHECK_LAUNCH();
  return SDLT_OK;
}
