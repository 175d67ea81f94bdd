// All LoRA weight gradients of a backward pass in one grouped launch (gfx950).
//
//   out[c][r] / out[r][c] (+)= sum_m P[m][c] * Q[m][r]          P wide bf16 [M,Cw], Q skinny bf16 [M,Rp]
//
// The contraction runs over tokens, i.e. over the ROW index of both row-major operands, so neither is in MFMA
// fragment order (a lane needs 8 consecutive m of one column).  The MFMA kernel stages [32 tokens] x [64 columns] of P
// and [32] x [Rp] of Q row-major in LDS with LDS-DMA and reads the fragments back with the gfx950 transposing LDS read
// (ds_read_b64_tr_b16: a 16-lane group turns a [4 rows][16 cols] block into "4 consecutive rows of my column").
// One workgroup = one problem x 64 columns; its 4 waves take every 4th 32-token chunk, each with a private
// double-buffered LDS ring (no barriers in the loop), and are reduced through LDS at the end.  The stream is HBM-bound:
// every P element is read exactly once (~4 GB per SDXL step).  Shapes the MFMA path cannot take (column blocks that
// straddle a conv tap, i.e. Cin % 64 != 0, or operands that are not 16-byte aligned) fall to the VALU kernel below.
// Keeping every dY / s*T / s*U alive until the end of the backward pass (a few GB of the 288 GB) is what lets the
// 1154 SDXL problems share one launch.
#include "common.h"
#include "../../include/sdlt_kernels.h"

namespace {

constexpr int BC = 64;     // columns of P per workgroup
constexpr int BT = 32;     // tokens per chunk (= the K of one 16x16x32 MFMA)

typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));

// ds_read_b64_tr_b16: lane (g = l>>4, i = l&15) passes the address of 4 contiguous bf16 = row (i>>2), columns
// 4*(i&3)..+3 of its group's [4][16] block; it receives column i of that block (4 consecutive rows).
__device__ __forceinline__ bf16x4_t lds_read_tr(uint32_t addr) {
  bf16x4_t v;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr));
  return v;
}

template <int RP>
__global__ __launch_bounds__(256) void lora_grad_mfma_kernel(const sdlt_lora_grad_desc* descs, const int32_t* block_desc) {
  constexpr int NR = RP / 16;                       // rank fragments
  constexpr int PT = BT * BC * 2;                   // 4 KB: P chunk, row-major [32][64]
  constexpr int QT = BT * RP * 2;                   // Q chunk, row-major [32][RP]
  constexpr int STG = PT + QT;
  constexpr int QI = QT / 1024;                     // DMA instructions for the Q chunk (RP/16)
  constexpr int LPS = 4 + QI;
  constexpr int RING = 2 * STG;                     // per wave
  constexpr int RED = 4 * BC * RP * 4;              // fp32 partials of the 4 waves
  constexpr int SMEM = 4 * RING > RED ? 4 * RING : RED;
  __shared__ __attribute__((aligned(16))) char smem[SMEM];

  const sdlt_lora_grad_desc d = descs[block_desc[blockIdx.x]];
  const int cb = blockIdx.x - d.first_block;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c0 = cb * BC;
  const bf16_t* P = (const bf16_t*)d.P;
  const bf16_t* Q = (const bf16_t*)d.Q;
  const bf16_t* Z = (const bf16_t*)d.zero;

  // conv: this block's 64 columns lie inside one tap (Cin % 64 == 0)
  int tap_dy = 0, tap_dx = 0, ci0 = c0;
  if (d.conv) {
    int tap = c0 / d.Cin;
    ci0 = c0 - tap * d.Cin;
    tap_dy = tap / 3 - 1;
    tap_dx = tap % 3 - 1;
  }
  const int hw = d.conv ? d.Hout * d.Wout : 1;

  // DMA lane geometry.  P piece j (8 rows x 128 B): row 8j + lane/8, 16-byte chunk lane%8.
  const int prow = lane >> 3, pchk = lane & 7;
  const bool pcol_ok = c0 + pchk * 8 < d.Cw;        // Cw is a multiple of 8 (checked on the host side)
  // Q piece j (1024 B = 1024/(2 RP) rows): row j*(512/RP) + lane/(RP/8), chunk lane%(RP/8)
  constexpr int QCH = RP / 8;
  const int qrow = lane / QCH, qchk = lane % QCH;

  char* ring = smem + wave * RING;
  auto stage = [&](int chunk, int buf) {
    char* base = ring + buf * STG;
    const int m0 = chunk * BT;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = m0 + 8 * j + prow;
      const bf16_t* src = Z + pchk * 8;
      if (m < d.M && pcol_ok) {
        if (!d.conv) {
          src = P + (int64_t)m * d.ldp + c0 + pchk * 8;
        } else {
          int b = m / hw, rem = m - b * hw;
          int ho = rem / d.Wout, wo = rem - ho * d.Wout;
          int hi = ho * d.stride + tap_dy, wi = wo * d.stride + tap_dx;
          if (hi >= 0 && wi >= 0 && hi < d.Hin && wi < d.Win)
            src = P + ((int64_t)(b * d.Hin + hi) * d.Win + wi) * d.ldp + ci0 + pchk * 8;
        }
      }
      glds16(src, base + j * 1024);
    }
#pragma unroll
    for (int j = 0; j < QI; ++j) {
      const int m = m0 + j * (512 / RP) + qrow;
      const bf16_t* src = m < d.M ? Q + (int64_t)m * d.ldq + qchk * 8 : Z + qchk * 8;
      glds16(src, base + PT + j * 1024);
    }
  };

  f32x4 acc[4][NR];
#pragma unroll
  for (int f = 0; f < 4; ++f)
#pragma unroll
    for (int r = 0; r < NR; ++r) acc[f][r] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // transposing-read lane geometry: group g covers tokens 8g..8g+7 of the chunk as two [4][16] blocks (h = 0/1)
  const int g = lane >> 4, i = lane & 15;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)ring;
  const uint32_t pa = lds0 + (8 * g + (i >> 2)) * (BC * 2) + (i & 3) * 8;         // + h*4 rows, + f*32 B
  const uint32_t qa = lds0 + PT + (8 * g + (i >> 2)) * (RP * 2) + (i & 3) * 8;    // + h*4 rows, + r*32 B

  const int nchunk = (d.M + BT - 1) / BT;
  int chunk = wave, buf = 0;
  if (chunk < nchunk) stage(chunk, 0);
  for (; chunk < nchunk; chunk += 4, buf ^= 1) {
    if (chunk + 4 < nchunk) {
      stage(chunk + 4, buf ^ 1);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPS) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const uint32_t so = buf * STG;
    bf16x8 af[4], bfr[NR];
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      bf16x4_t lo = lds_read_tr(pa + so + f * 32), hi = lds_read_tr(pa + so + f * 32 + 4 * BC * 2);
      af[f] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    }
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      bf16x4_t lo = lds_read_tr(qa + so + r * 32), hi = lds_read_tr(qa + so + r * 32 + 4 * RP * 2);
      bfr[r] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int f = 0; f < 4; ++f) asm volatile("" : "+v"(af[f]));
#pragma unroll
    for (int r = 0; r < NR; ++r) asm volatile("" : "+v"(bfr[r]));
    // D[c][r] += sum_t P[t][c] Q[t][r]: A = P^T fragment (row = column c of P), B = Q^T fragment (col = rank r)
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int r = 0; r < NR; ++r) acc[f][r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[f], bfr[r], acc[f][r], 0, 0, 0);
    // the buffer read here is re-filled by the NEXT iteration's stage(): its DMA must not pass these LDS reads - they have
    // completed (lgkmcnt(0) above) before any later instruction issues
  }

  // ---- reduce the 4 waves: red[wave][c][r] fp32 (the rings are dead: every wave is past its last LDS read after the barrier)
  __syncthreads();
  float* red = (float*)smem;
#pragma unroll
  for (int f = 0; f < 4; ++f)
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int c = f * 16 + g * 4 + t, rr = r * 16 + i;      // C/D layout: col = lane&15 (rank), row = (lane>>4)*4 + t (column of P)
        red[(wave * BC + c) * RP + rr] = acc[f][r][t];
      }
  __syncthreads();
  const int R = d.R;
  for (int e = threadIdx.x; e < BC * R; e += 256) {
    int cc, r;
    if (d.rank_major) { r = e / BC; cc = e - r * BC; } else { cc = e / R; r = e - cc * R; }
    const int col = c0 + cc;
    if (col < d.Cw) {
      float v = red[cc * RP + r] + red[(BC + cc) * RP + r] + red[(2 * BC + cc) * RP + r] + red[(3 * BC + cc) * RP + r];
      float* o = d.rank_major ? d.out + (int64_t)r * d.Cw + col : d.out + (int64_t)col * R + r;
      *o = d.accumulate ? *o + v : v;
    }
  }
}

// VALU fallback (odd shapes): a thread owns one column and all Rp ranks (Rp fp32 accumulators); the wave-uniform Q row
// goes through the scalar cache (SGPR operands of v_fmac).  The 4 waves interleave rows and are reduced through LDS.
template <int RP>
__global__ __launch_bounds__(256) void lora_grad_valu_kernel(const sdlt_lora_grad_desc* descs, const int32_t* block_desc) {
  __shared__ float red[BC * (RP + 1)];
  const sdlt_lora_grad_desc d = descs[block_desc[blockIdx.x]];
  const int cb = blockIdx.x - d.first_block;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c = cb * BC + lane;
  const bool cok = c < d.Cw;
  float a0[RP];
#pragma unroll
  for (int j = 0; j < RP; ++j) a0[j] = 0.f;
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
    bf16_t pv = 0;
    if (cok) {
      if (!d.conv) {
        pv = P[(int64_t)m * d.ldp + c];
      } else {
        int b = m / hw, rem = m - b * hw;
        int ho = rem / d.Wout, wo = rem - ho * d.Wout;
        int hi = ho * d.stride + tap_dy, wi = wo * d.stride + tap_dx;
        if (hi >= 0 && wi >= 0 && hi < d.Hin && wi < d.Win) pv = P[((int64_t)(b * d.Hin + hi) * d.Win + wi) * d.ldp + ci];
      }
    }
    const float p0 = bf2f(pv);
    const uint32_t* qrow = (const uint32_t*)(Q + (int64_t)m * d.ldq);  // wave-uniform address -> s_load
#pragma unroll
    for (int j = 0; j < RP / 2; ++j) {
      uint32_t qq = qrow[j];
      a0[2 * j] += p0 * bf2f(qq & 0xffff);
      a0[2 * j + 1] += p0 * bf2f(qq >> 16);
    }
  }
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int j = 0; j < RP; ++j) {
        if (w == 0) red[lane * (RP + 1) + j] = a0[j];
        else red[lane * (RP + 1) + j] += a0[j];
      }
    }
    __syncthreads();
  }
  const int R = d.R;
  for (int e = threadIdx.x; e < BC * R; e += 256) {
    int cc, r;
    if (d.rank_major) { r = e / BC; cc = e - r * BC; } else { cc = e / R; r = e - cc * R; }
    const int col = cb * BC + cc;
    if (col < d.Cw) {
      float v = red[cc * (RP + 1) + r];
      float* o = d.rank_major ? d.out + (int64_t)r * d.Cw + col : d.out + (int64_t)col * R + r;
      *o = d.accumulate ? *o + v : v;
    }
  }
}

template <int RP>
void launch(const sdlt_lora_grad_desc* descs, const int32_t* block_desc, int n_blocks, bool mfma, hipStream_t s) {
  if (mfma) hipLaunchKernelGGL(lora_grad_mfma_kernel<RP>, dim3(n_blocks), dim3(256), 0, s, descs, block_desc);
  else hipLaunchKernelGGL(lora_grad_valu_kernel<RP>, dim3(n_blocks), dim3(256), 0, s, descs, block_desc);
}

}  // namespace

extern "C" int32_t sdlt_lora_grad_block_cols(void) { return BC; }

extern "C" int sdlt_lora_grad_grouped(const sdlt_lora_grad_desc* descs_dev, const int32_t* block_desc_dev,
                                      int32_t n_blocks, int32_t Rp, int32_t mfma, void* stream) {
  if (n_blocks <= 0) SDLT_FAIL(SDLT_ERR_SHAPE, "sdlt_lora_grad_grouped: n_blocks=%d", n_blocks);
  hipStream_t s = (hipStream_t)stream;
  switch (Rp) {
    case 16: launch<16>(descs_dev, block_desc_dev, n_blocks, mfma != 0, s); break;
    case 32: launch<32>(descs_dev, block_desc_dev, n_blocks, mfma != 0, s); break;
    case 64: launch<64>(descs_dev, block_desc_dev, n_blocks, mfma != 0, s); break;
    default: SDLT_FAIL(SDLT_ERR_UNSUPPORTED, "sdlt_lora_grad_grouped: padded rank %d (16/32/64)", Rp);
  }
  SDLT_CHECK_LAUNCH();
  return SDLT_OK;
}
