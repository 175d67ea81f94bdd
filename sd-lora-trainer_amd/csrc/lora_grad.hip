// All LoRA weight gradients of a backward pass in one grouped launch (gfx950).
//
//   out[c][r] / out[r][c] (+)= sum_m P[m][c] * Q[m][r]          P wide bf16 [M,Cw], Q skinny bf16 [M,Rp]
//
// The contraction runs over tokens (both operands would need transposed MFMA fragments) and the total
// work is ~1% of the step's FLOPs, so this is a VALU kernel: a thread owns two adjacent columns c and
// all Rp ranks (2*Rp fp32 accumulators); the wave-uniform Q row goes through the scalar cache (SGPR
// operands of v_fmac), P is read with one coalesced 4-byte load per lane and row.  The 4 waves of a
// workgroup interleave rows and are reduced through LDS.  Keeping every dY / s*T / s*U alive until the
// end of the backward pass (a few GB of the 288 GB) is what lets the 1154 SDXL problems share one launch.
#include "common.h"
#include "../../include/sdlt_kernels.h"

namespace {

template <int RP>
__global__ __launch_bounds__(256) void lora_grad_kernel(const sdlt_lora_grad_desc* descs, const int32_t* block_desc) {
  __shared__ float red[128 * (RP + 1)];
  const sdlt_lora_grad_desc d = descs[block_desc[blockIdx.x]];
  const int cb = blockIdx.x - d.first_block;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int cl = lane * 2;                 // local column pair
  const int c = cb * 128 + cl;
  const bool cok = c < d.Cw;
  float a0[RP], a1[RP];
#pragma unroll
  for (int j = 0; j < RP; ++j) a0[j] = a1[j] = 0.f;

  const bf16_t* P = (const bf16_t*)d.P;
  const bf16_t* Q = (const bf16_t*)d.Q;
  int tap_dy = 0, tap_dx = 0, ci = 0;
  if (d.conv) {
    int tap = c / d.Cin;
    ci = c - tap * d.Cin;
    tap_dy = tap / 3 - 1;
    tap_dx = tap % 3 - 1;
  }
  const int hw = d.conv ? d.Hout * d.Wout : 1;
  for (int m = wave; m < d.M; m += 4) {
    uint32_t pv = 0;
    if (cok) {
      if (!d.conv) {
        pv = *(const uint32_t*)(P + (int64_t)m * d.ldp + c);
      } else {
        int b = m / hw, rem = m - b * hw;
        int ho = rem / d.Wout, wo = rem - ho * d.Wout;
        int hi = ho * d.stride + tap_dy, wi = wo * d.stride + tap_dx;
        if (hi >= 0 && wi >= 0 && hi < d.Hin && wi < d.Win)
          pv = *(const uint32_t*)(P + ((int64_t)(b * d.Hin + hi) * d.Win + wi) * d.ldp + ci);
      }
    }
    const float p0 = bf2f(pv & 0xffff), p1 = bf2f(pv >> 16);
    const uint32_t* qrow = (const uint32_t*)(Q + (int64_t)m * d.ldq);  // wave-uniform address -> s_load
#pragma unroll
    for (int j = 0; j < RP / 2; ++j) {
      uint32_t qq = qrow[j];
      float q0 = bf2f(qq & 0xffff), q1 = bf2f(qq >> 16);
      a0[2 * j] += p0 * q0; a0[2 * j + 1] += p0 * q1;
      a1[2 * j] += p1 * q0; a1[2 * j + 1] += p1 * q1;
    }
  }
  // reduce the 4 waves through LDS: red[col][RP+1]
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int j = 0; j < RP; ++j) {
        if (w == 0) {
          red[cl * (RP + 1) + j] = a0[j];
          red[(cl + 1) * (RP + 1) + j] = a1[j];
        } else {
          red[cl * (RP + 1) + j] += a0[j];
          red[(cl + 1) * (RP + 1) + j] += a1[j];
        }
      }
    }
    __syncthreads();
  }
  const int R = d.R;
  if (d.rank_major) {
    for (int e = threadIdx.x; e < 128 * R; e += 256) {
      int r = e >> 7, cc = e & 127;
      int col = cb * 128 + cc;
      if (col < d.Cw) {
        float v = red[cc * (RP + 1) + r];
        float* o = d.out + (int64_t)r * d.Cw + col;
        *o = d.accumulate ? *o + v : v;
      }
    }
  } else {
    for (int e = threadIdx.x; e < 128 * R; e += 256) {
      int cc = e / R, r = e - cc * R;
      int col = cb * 128 + cc;
      if (col < d.Cw) {
        float v = red[cc * (RP + 1) + r];
        float* o = d.out + (int64_t)col * R + r;
        *o = d.accumulate ? *o + v : v;
      }
    }
  }
}

}  // namespace

extern "C" int sdlt_lora_grad_grouped(const sdlt_lora_grad_desc* descs_dev, const int32_t* block_desc_dev,
                                      int32_t n_blocks, int32_t Rp, void* stream) {
  if (n_blocks <= 0) SDLT_FAIL(SDLT_ERR_SHAPE, "sdlt_lora_grad_grouped: n_blocks=%d", n_blocks);
  hipStream_t s = (hipStream_t)stream;
  switch (Rp) {
    case 16: hipLaunchKernelGGL(lora_grad_kernel<16>, dim3(n_blocks), dim3(256), 0, s, descs_dev, block_desc_dev); break;
    case 32: hipLaunchKernelGGL(lora_grad_kernel<32>, dim3(n_blocks), dim3(256), 0, s, descs_dev, block_desc_dev); break;
    case 64: hipLaunchKernelGGL(lora_grad_kernel<64>, dim3(n_blocks), dim3(256), 0, s, descs_dev, block_desc_dev); break;
    default: SDLT_FAIL(SDLT_ERR_UNSUPPORTED, "sdlt_lora_grad_grouped: padded rank %d (16/32/64)", Rp);
  }
  SDLT_CHECK_LAUNCH();
  return SDLT_OK;
}
