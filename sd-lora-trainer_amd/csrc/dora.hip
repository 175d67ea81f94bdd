// DoRA (weight-decomposed LoRA; peft LoraConfig(use_dora=True), reference trainer/optimizer.py:86-95, config.py:153-157).
// Every adapted layer computes  y = (m / ||W + s B A||_row) * (x W^T + s x A^T B^T) + bias,  the row norm taken as a constant of
// the step (detached) and the magnitude m [N] trained next to A and B.  Three batched launches per step serve ALL adapted layers
// (577 in SDXL), each driven by a device descriptor table built once:
//   dora_refresh_kernel : after the optimizer - row norms of the effective weight on the matrix cores, WITHOUT forming B A:
//                           ||W_n + s B_n A||^2 = ||W_n||^2 + 2 s B_n . (W A^T)_n + s^2 B_n (A A^T) B_n^T
//                         -> scale = m / norm (the forward GEMM's col_scale) and the scaled backward LoRA-down operand (B scale)^T
//   dora_scale_wt_kernel: the dX operand W^T (conv: the flipped tap-major copy) with its K axis pre-multiplied by scale, so that
//                         the dX GEMMs run unchanged on dY (dX = (dY * scale) W + ...)
//   dora_mag_grad_*     : d m[n] = sum_rows dY[row, n] * (y[row, n] - bias[n]) / m[n]  (fixed-order two-stage column reduction) and
//                         the row scaling of the LoRA-up gradient (d B = scale * dY^T (s T))
// HBM-bound: one pass over the adapted weights (refresh), one read + one write of their transposes, one pass over dY and y.
#include "common.h"
#include "../../include/sdlt_kernels.h"

namespace {

template <int NR>    // NR = padded rank / 16
__global__ __launch_bounds__(256) void dora_refresh_kernel(const sdlt_dora_desc* descs, const int32_t* block_desc, const int32_t* block_first, int init) {
  const sdlt_dora_desc d = descs[block_desc[blockIdx.x]];
  constexpr int RP = NR * 16;
  __shared__ float S_sh[4][RP][RP + 1];
  __shared__ float B_sh[4][16][RP + 1];
  __shared__ float red[4][2][16];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, frow = lane & 15, fk = lane >> 4;
  const int n_base = ((blockIdx.x - block_first[block_desc[blockIdx.x]]) * 4 + wave) * 16;
  if (n_base >= d.N) return;                        // (no block-wide barrier below: every exchange is within one wave)
  const int nrow = min(n_base + frow, d.N - 1);
  const bf16_t* wrow = (const bf16_t*)d.W + (size_t)nrow * d.ldw + fk * 8;
  const bf16_t* arow = (const bf16_t*)d.A + (size_t)frow * d.lda + fk * 8;
  f32x4 G[NR], S[NR][NR];
#pragma unroll
  for (int j = 0; j < NR; ++j) {
    G[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NR; ++i) S[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  float w2 = 0.f;
  for (int k0 = 0; k0 < d.K; k0 += 32) {
    const bf16x8 wf = *(const bf16x8*)(wrow + k0);
    bf16x8 af[NR];
#pragma unroll
    for (int j = 0; j < NR; ++j) af[j] = *(const bf16x8*)(arow + (size_t)j * 16 * d.lda + k0);
#pragma unroll
    for (int j = 0; j < NR; ++j) G[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, af[j], G[j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < NR; ++i)
#pragma unroll
      for (int j = 0; j < NR; ++j) S[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], af[j], S[i][j], 0, 0, 0);
#pragma unroll
    for (int e = 0; e < 8; ++e) { const float x = (float)wf[e]; w2 += x * x; }
  }
  // accumulator layout: X[r] of lane = C[row = fk*4 + r][col = frow]
#pragma unroll
  for (int i = 0; i < NR; ++i)
#pragma unroll
    for (int j = 0; j < NR; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) S_sh[wave][i * 16 + fk * 4 + r][j * 16 + frow] = S[i][j][r];
  for (int e = lane; e < 16 * RP; e += 64) {
    const int rr = e / RP, c = e - rr * RP;
    const int n = min(n_base + rr, d.N - 1);
    B_sh[wave][rr][c] = bf2f(((const bf16_t*)d.B)[(size_t)n * d.ldb + c]);
  }
  w2 += __shfl_xor(w2, 16, 64);
  w2 += __shfl_xor(w2, 32, 64);                     // row frow, all K
  if (fk == 0) red[wave][0][frow] = w2;
  __builtin_amdgcn_s_waitcnt(0);
  __builtin_amdgcn_wave_barrier();
  float part[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < NR; ++j) {
    const int r2 = j * 16 + frow;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int nl = fk * 4 + r;
      float t = 0.f;
      for (int r1 = 0; r1 < RP; ++r1) t += B_sh[wave][nl][r1] * S_sh[wave][r1][r2];
      part[r] += B_sh[wave][nl][r2] * (2.f * d.s * G[j][r] + d.s * d.s * t);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) part[r] += __shfl_xor(part[r], o, 64);
    if (frow == 0) red[wave][1][fk * 4 + r] = part[r];
  }
  __builtin_amdgcn_s_waitcnt(0);
  __builtin_amdgcn_wave_barrier();
  float sc = 0.f;
  const int n = n_base + frow;
  if (fk == 0 && n < d.N) {
    const float nrm = sqrtf(fmaxf(red[wave][0][frow] + red[wave][1][frow], 1e-30f));
    if (init) d.mag[n] = nrm;
    sc = d.mag[n] / nrm;
    d.scale[n] = sc;
  }
  sc = __shfl(sc, frow, 64);                        // lanes of every fk group: the scale of row n_base + frow
  if (d.Bt && n < d.N) {
    for (int r = fk; r < RP; r += 4) {
      const float b = r < d.rank ? d.B32[(size_t)n * d.ldb32 + r] : 0.f;
      ((bf16_t*)d.Bt)[(size_t)r * d.ldbt + n] = f2bf(b * sc);
    }
  }
}

__global__ __launch_bounds__(256) void dora_scale_wt_kernel(const sdlt_dora_wt_desc* descs, const int32_t* block_desc, const int32_t* block_first) {
  const sdlt_dora_wt_desc d = descs[block_desc[blockIdx.x]];
  const int t = blockIdx.x - block_first[block_desc[blockIdx.x]];
  const int tiles_c = (d.cols + 255) >> 8;
  const int r0 = (t / tiles_c) * 32, c = (t % tiles_c) * 256 + (threadIdx.x & 31) * 8;
  if (c >= d.cols) return;
  float sc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int j = (c + e) % d.period;
    sc[e] = (c + e < d.cols && j < d.nvalid) ? d.scale[j] : 0.f;
  }
  const bool vec = c + 8 <= d.cols;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + (threadIdx.x >> 5) + i * 8;
    if (r >= d.rows) break;
    const bf16_t* src = (const bf16_t*)d.src + (size_t)r * d.ld + c;
    bf16_t* dst = (bf16_t*)d.dst + (size_t)r * d.ld + c;
    if (vec) {
      const uint4 v = *(const uint4*)src;
      uint4 o;
      o.x = pack2bf(bf2f(v.x & 0xffff) * sc[0], bf2f(v.x >> 16) * sc[1]);
      o.y = pack2bf(bf2f(v.y & 0xffff) * sc[2], bf2f(v.y >> 16) * sc[3]);
      o.z = pack2bf(bf2f(v.z & 0xffff) * sc[4], bf2f(v.z >> 16) * sc[5]);
      o.w = pack2bf(bf2f(v.w & 0xffff) * sc[6], bf2f(v.w >> 16) * sc[7]);
      *(uint4*)dst = o;
    } else {
      for (int e = 0; e < 8 && c + e < d.cols; ++e) dst[e] = f2bf(bf2f(src[e]) * sc[e]);
    }
  }
}

// stage 1: block = (layer, 64-column chunk, row split) -> part[split][0][n] = sum dY*y, part[split][1][n] = sum dY over its rows
__global__ __launch_bounds__(256) void dora_mag_grad_partial_kernel(const sdlt_dora_grad_desc* descs, const int32_t* block_desc, const int32_t* block_first,
                                                                    float* ws) {
  const sdlt_dora_grad_desc d = descs[block_desc[blockIdx.x]];
  const int t = blockIdx.x - block_first[block_desc[blockIdx.x]];
  const int nch = (d.N + 63) >> 6;
  const int chunk = t % nch, split = t / nch;
  const int rows_per = (((d.M + d.splits - 1) / d.splits) + 31) & ~31;
  const int rbeg = split * rows_per, rend = min(d.M, rbeg + rows_per);
  const int cv = threadIdx.x & 7, rl = threadIdx.x >> 3;       // 8 x 16 B = 64 columns per row, 32 rows per pass
  const int c = chunk * 64 + cv * 8;
  float sy[8], sd[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) sy[e] = sd[e] = 0.f;
  if (c < d.N) {
    for (int r = rbeg + rl; r < rend; r += 32) {
      const uint4 g = *(const uint4*)((const bf16_t*)d.dY + (size_t)r * d.lddy + c);
      const uint4 y = *(const uint4*)((const bf16_t*)d.Y + (size_t)r * d.ldy + c);
      const uint32_t gw[4] = {g.x, g.y, g.z, g.w}, yw[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float g0 = bf2f(gw[q] & 0xffff), g1 = bf2f(gw[q] >> 16);
        sy[2 * q] += g0 * bf2f(yw[q] & 0xffff); sy[2 * q + 1] += g1 * bf2f(yw[q] >> 16);
        sd[2 * q] += g0; sd[2 * q + 1] += g1;
      }
    }
  }
  __shared__ float sh[2][32][65];
#pragma unroll
  for (int e = 0; e < 8; ++e) { sh[0][rl][cv * 8 + e] = sy[e]; sh[1][rl][cv * 8 + e] = sd[e]; }
  __syncthreads();
  if (threadIdx.x < 128) {
    const int which = threadIdx.x >> 6, col = threadIdx.x & 63;
    float a = 0.f;
#pragma unroll 8
    for (int r = 0; r < 32; ++r) a += sh[which][r][col];          // fixed order
    const int n = chunk * 64 + col;
    if (n < d.N) ws[d.ws_off + ((size_t)split * 2 + which) * d.N + n] = a;
  }
}

// stage 2: one thread per column: splits summed in order; d m = (sum dY*y - bias * sum dY) / m; d B row *= scale
__global__ __launch_bounds__(256) void dora_mag_grad_finish_kernel(const sdlt_dora_grad_desc* descs, const int32_t* block_desc, const int32_t* block_first,
                                                                   const float* ws) {
  const sdlt_dora_grad_desc d = descs[block_desc[blockIdx.x]];
  const int n = (blockIdx.x - block_first[block_desc[blockIdx.x]]) * 256 + threadIdx.x;
  if (n >= d.N) return;
  float sy = 0.f, sd = 0.f;
  for (int s = 0; s < d.splits; ++s) {
    sy += ws[d.ws_off + ((size_t)s * 2 + 0) * d.N + n];
    sd += ws[d.ws_off + ((size_t)s * 2 + 1) * d.N + n];
  }
  const float b = d.bias ? d.bias[n] : 0.f;
  // y - bias = scale * z, scale = m / norm:  d m = sum dY * z / norm = sum dY * (y - bias) / m
  const float gm = d.grad_scale * (sy - b * sd) / d.mag[n];
  if (d.accumulate) {      // second pass through the same adapters: its own launch behind the first pass's (which also scaled the summed dB rows)
    d.gmag[n] += gm;
    return;
  }
  d.gmag[n] = gm;
  const float sc = d.scale[n];
  float* gb = d.gB + (size_t)n * d.rank;
  for (int r = 0; r < d.rank; ++r) gb[r] *= sc;
}

}  // namespace

extern "C" int sdlt_dora_refresh(const sdlt_dora_desc* descs_dev, const int32_t* block_desc_dev, const int32_t* block_first_dev, int32_t n_blocks,
                                 int32_t Rp, int32_t init, void* stream) {
  if (n_blocks <= 0) SDLT_FAIL(SDLT_ERR_SHAPE, "sdlt_dora_refresh: n_blocks=%d", n_blocks);
  hipStream_t s = (hipStream_t)stream;
  switch (Rp) {
    case 16: hipLaunchKernelGGL(dora_refresh_kernel<1>, dim3(n_blocks), dim3(256), 0, s, descs_dev, block_desc_dev, block_first_dev, init); break;
    case 32: hipLaunchKernelGGL(dora_refresh_kernel<2>, dim3(n_blocks), dim3(256), 0, s, descs_dev, block_desc_dev, block_first_dev, init); break;
    case 64: hipLaunchKernelGGL(dora_refresh_kernel<4>, dim3(n_blocks), dim3(256), 0, s, descs_dev, block_desc_dev, block_first_dev, init); break;
    default: SDLT_FAIL(SDLT_ERR_UNSUPPORTED, "sdlt_dora_refresh: padded rank %d (16/32/64)", Rp);
  }
  SDLT_CHECK_LAUNCH();
  return SDLT_OK;
}

extern "C" int sdlt_dora_scale_wt(const sdlt_dora_wt_desc* descs_dev, const int32_t* block_desc_dev, const int32_t* block_first_dev, int32_t n_blocks,
                                  void* stream) {
  if (n_blocks <= 0) SDLT_FAIL(SDLT_ERR_SHAPE, "sdlt_dora_scale_wt: n_blocks=%d", n_blocks);
  hipLaunchKernelGGL(dora_scale_wt_kernel, dim3(n_blocks), dim3(256), 0, (hipStream_t)stream, descs_dev, block_desc_dev, block_first_dev);
  SDLT_CHECK_LAUNCH();
  return SDLT_OK;
}

extern "C" int sdlt_dora_mag_grad(const sdlt_dora_grad_desc* descs_dev, const int32_t* block_desc_dev, const int32_t* block_first_dev, int32_t n_blocks,
                                  const int32_t* fin_block_desc_dev, const int32_t* fin_block_first_dev, int32_t n_fin_blocks, float* ws, void* stream) {
  if (n_blocks <= 0 || n_fin_blocks <= 0 || !ws) SDLT_FAIL(SDLT_ERR_SHAPE, "sdlt_dora_mag_grad: n_blocks=%d n_fin_blocks=%d ws=%p", n_blocks, n_fin_blocks, (void*)ws);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(dora_mag_grad_partial_kernel, dim3(n_blocks), dim3(256), 0, s, descs_dev, block_desc_dev, block_first_dev, ws);
  hipLaunchKernelGGL(dora_mag_grad_finish_kernel, dim3(n_fin_blocks), dim3(256), 0, s, descs_dev, fin_block_desc_dev, fin_block_first_dev, (const float*)ws);
  SDLT_CHECK_LAUNCH();
  return SDLT_OK;
}
