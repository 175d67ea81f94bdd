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
// A thread owns the row PAIR (2i, 2i+1) of the tile and 8 channels: two 16-B loads, and each channel's two values leave
// as one 32-bit word (m, m+1 adjacent in the transposed tile), so the LDS sees 8 word writes and 4 b64 reads per thread.
__device__ __forceinline__ int64_t gather_src(int m, int M, int tap, const GatherGeom& g) {
  if (m >= M) return -1;
  if (!g.conv) return m;
  const int hw = g.Hout * g.Wout;
  const int b = m / hw, rem = m - b * hw, oy = rem / g.Wout, ox = rem - oy * g.Wout;
  const int iy = oy * g.stride + tap / 3 - 1, ix = ox * g.stride + tap % 3 - 1;
  if (iy < 0 || iy >= g.H * g.ups || ix < 0 || ix >= g.W * g.ups) return -1;
  return ((int64_t)b * g.H + iy / g.ups) * g.W + ix / g.ups;
}

__device__ __forceinline__ uint4 load_row8(const bf16_t* x, int64_t ldx, int64_t src, int c, int C) {
  uint4 u = make_uint4(0, 0, 0, 0);
  if (src < 0 || c >= C) return u;
  if (c + 8 <= C) return *(const uint4*)(x + src * ldx + c);
  bf16_t tmp[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // ragged channel tail (conv_in: 4 channels)
  for (int j = 0; j < 8 && c + j < C; ++j) tmp[j] = x[src * ldx + c + j];
  return *(const uint4*)tmp;
}

// items != nullptr: batched launch - blockIdx.z = problem * taps + tap, every problem with its own x / out / colsum pointers
// and the shared geometry (the weight gradients of all layers of one shape are issued together at the end of the backward)
__global__ __launch_bounds__(256) void transpose_gather_kernel(const bf16_t* x, int64_t ldx, int M, int C, bf16_t* out,
                                                                int64_t ldo, GatherGeom g, float* colsum,
                                                                const sdlt_wgrad_tr_item* items, int taps) {
  __shared__ uint32_t tileT[64][34];          // [channel][row pair], 8-B aligned rows
  int tap = blockIdx.z;
  if (items) {
    const sdlt_wgrad_tr_item it = items[blockIdx.z / taps];
    tap = blockIdx.z % taps;
    x = (const bf16_t*)it.x; out = (bf16_t*)it.out; colsum = it.colsum;
  }
  const int m0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int t = threadIdx.x;
  {
    const int i = t >> 3, cc = (t & 7) * 8;
    const int m = m0 + 2 * i;
    const uint4 a = load_row8(x, ldx, gather_src(m, M, tap, g), c0 + cc, C);
    const uint4 b = load_row8(x, ldx, gather_src(m + 1, M, tap, g), c0 + cc, C);
    const uint32_t aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      tileT[cc + 2 * j][i] = (aw[j] & 0xffffu) | (bw[j] << 16);
      tileT[cc + 2 * j + 1][i] = (aw[j] >> 16) | (bw[j] & 0xffff0000u);
    }
  }
  __syncthreads();
  {
    const int c = t >> 2, w0 = (t & 3) * 8;     // 8 words = 16 tokens
    if (c0 + c < C) {
      const uint2* src = (const uint2*)&tileT[c][w0];
      const uint2 p0 = src[0], p1 = src[1], p2 = src[2], p3 = src[3];
      bf16_t* dst = out + ((int64_t)tap * C + c0 + c) * ldo + m0 + w0 * 2;
      *(uint4*)dst = make_uint4(p0.x, p0.y, p1.x, p1.y);
      *(uint4*)(dst + 8) = make_uint4(p2.x, p2.y, p3.x, p3.y);
      if (colsum) {
        // bias gradient for free: the 16 tokens this thread just read, summed; the 4 threads of a channel combine, one atomic each tile
        const uint32_t w[8] = {p0.x, p0.y, p1.x, p1.y, p2.x, p2.y, p3.x, p3.y};
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) sum += bf2f(w[j] & 0xffffu) + bf2f(w[j] >> 16);
        sum += __shfl_xor(sum, 1, 64);
        sum += __shfl_xor(sum, 2, 64);
        if ((t & 3) == 0) atomicAdd(&colsum[c0 + c], sum);
      }
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

// items != nullptr: batched launch, blockIdx.z = problem (all norm layers of one shape at the end of the backward)
template <int MODE>
__global__ __launch_bounds__(256) void norm_affine_grad_kernel(CatIn2 in, const bf16_t* dy, int64_t lddy, int64_t M, int HW, int C,
                                                                const float* stats, const float* gamma, const float* beta,
                                                                float eps, int silu, float* dgamma, float* dbeta,
                                                                const sdlt_affine_grad_item* items) {
  __shared__ float red[2][4][64];
  if (items) {
    const sdlt_affine_grad_item it = items[blockIdx.z];
    in.x1 = (const bf16_t*)it.x1; in.x2 = (const bf16_t*)it.x2; dy = (const bf16_t*)it.dy; stats = it.stats;
    gamma = it.gamma; beta = it.beta; dgamma = it.dgamma; dbeta = it.dbeta;
  }
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const int64_t per = (M + gridDim.y - 1) / gridDim.y;
  const int64_t r0 = blockIdx.y * per, r1 = (r0 + per < M) ? r0 + per : M;
  float dg = 0.f, db = 0.f;
  if (c < C) {
    const float ga = MODE == 1 ? gamma[c] : 1.f, be = MODE == 1 ? beta[c] : 0.f;
    const int grp = c / (C / 32);
    const float inv_n = 1.f / ((float)HW * (float)(C / 32));
    // eight rows per turn, every load of the turn requested before the arithmetic (the row loop was one dependent x / dy / statistics round trip per row: 1.3-2.7 TB/s);
    // rows past the chunk re-read its first row and are masked
    constexpr int U = 8;
    for (int64_t r = r0 + rl; r < r1; r += 4 * U) {
      float xs[U], ds[U];
      float2 st[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t rr = r + 4 * u < r1 ? r + 4 * u : r;
        xs[u] = in.at(rr, c);
        ds[u] = bf2f(dy[rr * lddy + c]);
        st[u] = MODE == 0 ? *(const float2*)(stats + rr * 2) : *(const float2*)(stats + (div_small((int)rr, HW) * 32 + grp) * 2);      // (div_small is exact below 2^22 rows: both GroupNorm launchers check B * HW)
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        float mean, rstd;
        if (MODE == 0) { mean = st[u].x; rstd = st[u].y; }
        else {
          mean = st[u].x * inv_n;
          rstd = rsqrtf(fmaxf(st[u].y * inv_n - mean * mean, 0.f) + eps);
        }
        const float xh = (xs[u] - mean) * rstd;
        float d = ds[u];
        if (MODE == 1 && silu) d *= dsilu_f(xh * ga + be);
        if (r + 4 * u >= r1) d = 0.f;
        dg += d * xh;
        db += d;
      }
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

// gamma and beta gradients are adjacent in the trainer's arena: one zero launch covers both
void zero_pair(float* dgamma, float* dbeta, int C, hipStream_t s) {
  if (dbeta == dgamma + C) sdlt_zero_async(dgamma, sizeof(float) * 2 * C, s);
  else {
    sdlt_zero_async(dgamma, sizeof(float) * C, s);
    sdlt_zero_async(dbeta, sizeof(float) * C, s);
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
                                    float* colsum_acc, void* stream) {
  if (M <= 0 || C <= 0 || Mp < M || (Mp % 64) || (ldx % 8) || (ldo % 8) || ldo < Mp)
    SDLT_FAIL(SDLT_ERR_SHAPE, "sdlt_wgrad_transpose: M=%d C=%d Mp=%d ldx=%lld ldo=%lld", M, C, Mp, (long long)ldx, (long long)ldo);
  GatherGeom g{};
  hipLaunchKernelGGL(transpose_gather_kernel, dim3(Mp / 64, (C + 63) / 64, 1), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x, ldx, M, C, (bf16_t*)out, ldo, g, colsum_acc, (const sdlt_wgrad_tr_item*)nullptr, 1);
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
                     (const bf16_t*)x, ldx, (int)M, C, (bf16_t*)out, ldo, g, (float*)nullptr, (const sdlt_wgrad_tr_item*)nullptr, 9);
  SDLT_CHECK_LAUNCH();
  return SDLT_OK;
}

extern "C" int sdlt_wgrad_transpose_batch(const sdlt_wgrad_tr_item* items_dev, int32_t n, int64_t ldx, int32_t M, int32_t C, int64_t ldo,
                                          int32_t Mp, void* stream) {
  if (!items_dev || n <= 0 || n > 65535 || M <= 0 || C <= 0 || Mp < M || (Mp % 64) || (ldx % 8) || (ldo % 8) || ldo < Mp)
    SDLT_FAIL(SDLT_ERR_SHAPE, "sdlt_wgrad_transpose_batch: n=%d M=%d C=%d Mp=%d ldx=%lld ldo=%lld", n, M, C, Mp, (long long)ldx, (long long)ldo);
  GatherGeom g{};
  hipLaunchKernelGGL(transpose_gather_kernel, dim3(Mp / 64, (C + 63) / 64, n), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)nullptr, ldx, M, C, (bf16_t*)nullptr, ldo, g, (float*)nullptr, items_dev, 1);
  SDLT_CHECK_LAUNCH();
  return SDLT_OK;
}

extern "C" int sdlt_wgrad_im2col_t_batch(const sdlt_wgrad_tr_item* items_dev, int32_t n, int64_t ldx, int32_t B, int32_t H, int32_t W,
                                         int32_t C, int32_t stride, int32_t ups, int64_t ldo, int32_t Mp, void* stream) {
  if (!items_dev || n <= 0 || 9 * n > 65535 || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (stride != 1 && stride != 2) || (ups != 1 && ups != 2) ||
      (stride == 2 && ups == 2))
    SDLT_FAIL(SDLT_ERR_SHAPE, "sdlt_wgrad_im2col_t_batch: n=%d B=%d H=%d W=%d C=%d stride=%d ups=%d", n, B, H, W, C, stride, ups);
  GatherGeom g{1, B, H, W, H * ups / stride, W * ups / stride, stride, ups};
  const int64_t M = (int64_t)B * g.Hout * g.Wout;
  if (Mp < M || (Mp % 64) || (ldx % 8) || (ldo % 8) || ldo < Mp)
    SDLT_FAIL(SDLT_ERR_SHAPE, "sdlt_wgrad_im2col_t_batch: M=%lld Mp=%d ldx=%lld ldo=%lld", (long long)M, Mp, (long long)ldx, (long long)ldo);
  hipLaunchKernelGGL(transpose_gather_kernel, dim3(Mp / 64, (C + 63) / 64, 9 * n), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)nullptr, ldx, (int)M, C, (bf16_t*)nullptr, ldo, g, (float*)nullptr, items_dev, 9);
  SDLT_CHECK_LAUNCH();
  return SDLT_OK;
}

extern "C" int sdlt_layernorm_affine_grad(const void* x, int64_t ldx, const void* dy, int64_t lddy, int32_t M, int32_t C,
                                          const float* stats, float* dgamma, float* dbeta, int32_t accumulate, void* stream) {
  if (M <= 0 || C <= 0 || !stats || !dgamma || !dbeta) SDLT_FAIL(SDLT_ERR_SHAPE, "sdlt_layernorm_affine_grad: M=%d C=%d", M, C);
  hipStream_t s = (hipStream_t)stream;
  if (!accumulate) zero_pair(dgamma, dbeta, C, s);
  CatIn2 in{(const bf16_t*)x, ldx, C, nullptr, 0};
  const int cb = (C + 63) / 64;
  hipLaunchKernelGGL(norm_affine_grad_kernel<0>, dim3(cb, row_chunks(M, cb)), dim3(256), 0, s, in, (const bf16_t*)dy, lddy,
                     (int64_t)M, 1, C, stats, nullptr, nullptr, 0.f, 0, dgamma, dbeta, (const sdlt_affine_grad_item*)nullptr);
  SDLT_CHECK_LAUNCH();
  return SDLT_OK;
}

extern "C" int sdlt_groupnorm_affine_grad(const sdlt_groupnorm_params* pp, float* dgamma, float* dbeta, int32_t accumulate, void* stream) {
  const sdlt_groupnorm_params& p = *pp;
  if (p.B <= 0 || p.HW <= 0 || p.C <= 0 || (p.C % 32) || !p.stats || !p.dy || !dgamma || !dbeta)
    SDLT_FAIL(SDLT_ERR_SHAPE, "sdlt_groupnorm_affine_grad: B=%d HW=%d C=%d", p.B, p.HW, p.C);
  if ((int64_t)p.B * p.HW >= (1 << 22)) SDLT_FAIL(SDLT_ERR_SHAPE, "sdlt_groupnorm_affine_grad: B * HW = %lld rows >= 2^22 (row -> image division)", (long long)p.B * p.HW);
  hipStream_t s = (hipStream_t)stream;
  if (!accumulate) zero_pair(dgamma, dbeta, p.C, s);
  CatIn2 in{(const bf16_t*)p.x1, p.ldx1, p.x2 ? p.C1 : p.C, (const bf16_t*)p.x2, p.ldx2};
  const int cb = (p.C + 63) / 64;
  const int64_t M = (int64_t)p.B * p.HW;
  hipLaunchKernelGGL(norm_affine_grad_kernel<1>, dim3(cb, row_chunks(M, cb)), dim3(256), 0, s, in, (const bf16_t*)p.dy, p.lddy,
                     M, p.HW, p.C, p.stats, p.gamma, p.beta, p.eps, p.silu, dgamma, dbeta, (const sdlt_affine_grad_item*)nullptr);
  SDLT_CHECK_LAUNCH();
  return SDLT_OK;
}

extern "C" int sdlt_affine_grad_batch(const sdlt_affine_grad_item* items_dev, int32_t n, int32_t groupnorm, int64_t ldx1, int32_t C1,
                                      int64_t ldx2, int64_t lddy, int32_t B, int32_t HW, int32_t C, float eps, int32_t silu, void* stream) {
  if (!items_dev || n <= 0 || n > 65535 || B <= 0 || HW <= 0 || C <= 0 || (groupnorm && (C % 32)))
    SDLT_FAIL(SDLT_ERR_SHAPE, "sdlt_affine_grad_batch: n=%d B=%d HW=%d C=%d", n, B, HW, C);
  if (groupnorm && (int64_t)B * HW >= (1 << 22)) SDLT_FAIL(SDLT_ERR_SHAPE, "sdlt_affine_grad_batch: B * HW = %lld rows >= 2^22 (row -> image division)", (long long)B * HW);
  CatIn2 in{nullptr, ldx1, C1 > 0 ? C1 : C, nullptr, ldx2};
  const int cb = (C + 63) / 64;
  const int64_t M = (int64_t)B * HW;
  int rc = row_chunks(M, cb * n);
  if (rc < 1) rc = 1;
  if (groupnorm)
    hipLaunchKernelGGL(norm_affine_grad_kernel<1>, dim3(cb, rc, n), dim3(256), 0, (hipStream_t)stream, in, (const bf16_t*)nullptr, lddy, M, HW, C,
                       (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, eps, silu, (float*)nullptr, (float*)nullptr, items_dev);
  else
    hipLaunchKernelGGL(norm_affine_grad_kernel<0>, dim3(cb, rc, n), dim3(256), 0, (hipStream_t)stream, in, (const bf16_t*)nullptr, lddy, M, 1, C,
                       (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, 0.f, 0, (float*)nullptr, (float*)nullptr, items_dev);
  SDLT_CHECK_LAUNCH();
  return SDLT_OK;
}
