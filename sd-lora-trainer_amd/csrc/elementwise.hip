// HBM-bound element-wise / small kernels of the training step (gfx950): GEGLU, SiLU, timestep
// sinusoids, DDPM add-noise (+ NCHW->NHWC), masked-MSE loss fwd+bwd, fused AdamW(+L1), LoRA bf16
// shadow refresh, 2x2 sum (dX of nearest upsampling), column sums.
#include "common.h"
#include "../../include/sdlt_kernels.h"

namespace {

__device__ __forceinline__ void load8(const bf16_t* p, float* v) {
  uint4 u = *(const uint4*)p;
  v[0] = bf2f(u.x & 0xffff); v[1] = bf2f(u.x >> 16);
  v[2] = bf2f(u.y & 0xffff); v[3] = bf2f(u.y >> 16);
  v[4] = bf2f(u.z & 0xffff); v[5] = bf2f(u.z >> 16);
  v[6] = bf2f(u.w & 0xffff); v[7] = bf2f(u.w >> 16);
}
__device__ __forceinline__ void store8(bf16_t* p, const float* v) {
  uint4 u;
  u.x = pack2bf(v[0], v[1]); u.y = pack2bf(v[2], v[3]);
  u.z = pack2bf(v[4], v[5]); u.w = pack2bf(v[6], v[7]);
  *(uint4*)p = u;
}

// ------------------------------------------------------------------ GEGLU (diffusers GEGLU: h * gelu(g))
__global__ void geglu_fwd_kernel(const bf16_t* in, int64_t ldin, int M, int Ch, bf16_t* out, int64_t ldout) {
  const int nch = Ch >> 3;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < (int64_t)M * nch; i += (int64_t)gridDim.x * blockDim.x) {
    int m = i / nch, c = (i - (int64_t)m * nch) * 8;
    float h[8], g[8];
    load8(in + m * ldin + c, h);
    load8(in + m * ldin + Ch + c, g);
#pragma unroll
    for (int j = 0; j < 8; ++j) h[j] *= gelu_f(g[j]);
    store8(out + m * ldout + c, h);
  }
}
__global__ void geglu_bwd_kernel(const bf16_t* in, int64_t ldin, const bf16_t* dout, int64_t lddout, int M, int Ch,
                                 bf16_t* din, int64_t lddin) {
  const int nch = Ch >> 3;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < (int64_t)M * nch; i += (int64_t)gridDim.x * blockDim.x) {
    int m = i / nch, c = (i - (int64_t)m * nch) * 8;
    float h[8], g[8], d[8], dh[8], dg[8];
    load8(in + m * ldin + c, h);
    load8(in + m * ldin + Ch + c, g);
    load8(dout + m * lddout + c, d);
#pragma unroll
    for (int j = 0; j < 8; ++j) { dh[j] = d[j] * gelu_f(g[j]); dg[j] = d[j] * h[j] * dgelu_f(g[j]); }
    store8(din + m * lddin + c, dh);
    store8(din + m * lddin + Ch + c, dg);
  }
}

// ------------------------------------------------------------------ generic unary / binary maps on contiguous bf16
// op 0: y = silu(x)            op 1: y = dy * silu'(x)    op 2: y = x + dy (add)
// op 3: y = gelu(x)            op 4: y = dy * gelu'(x)
// op 5: y = x*sigmoid(1.702x) (CLIP quick_gelu)           op 6: y = dy * quick_gelu'(x)
__global__ void map_kernel(int op, const bf16_t* x, const bf16_t* dy, bf16_t* y, int64_t n8) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
    float a[8], b[8];
    load8(x + i * 8, a);
    if (dy) load8(dy + i * 8, b);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float v = a[j];
      switch (op) {
        case 0: v = silu_f(v); break;
        case 1: v = b[j] * dsilu_f(v); break;
        case 2: v = v + b[j]; break;
        case 3: v = gelu_f(v); break;
        case 4: v = b[j] * dgelu_f(v); break;
        case 5: v = v / (1.f + __expf(-1.702f * v)); break;
        case 6: { float s = 1.f / (1.f + __expf(-1.702f * v)); v = b[j] * (s + 1.702f * v * s * (1.f - s)); } break;
      }
      a[j] = v;
    }
    store8(y + i * 8, a);
  }
}

// ------------------------------------------------------------------ timestep sinusoid (diffusers Timesteps, flip_sin_to_cos, shift 0)
// out[row, 0:half] = cos(t*f_i), out[row, half:dim] = sin(t*f_i), f_i = exp(-ln(10000) * i / half)
__global__ void timestep_embed_kernel(const float* t, int rows, int dim, bf16_t* out, int64_t ldo) {
  int half = dim >> 1;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < rows * half; i += gridDim.x * blockDim.x) {
    int r = i / half, k = i - r * half;
    float f = expf(-9.210340371976184f * (float)k / (float)half);
    float a = t[r] * f;
    out[r * ldo + k] = f2bf(cosf(a));
    out[r * ldo + half + k] = f2bf(sinf(a));
  }
}

// ------------------------------------------------------------------ DDPM add_noise, NCHW fp32 -> NHWC bf16 padded to Cpad channels
// noisy = sqrt(abar_t) x0 + sqrt(1-abar_t) eps     (main.py:326 -> diffusers DDPMScheduler.add_noise)
__global__ void add_noise_kernel(const float* x0, const float* noise, const int64_t* ts, const float* acp, int B, int C,
                                 int HW, int Cpad, bf16_t* out, float* noisy_nchw) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B * HW * Cpad; i += gridDim.x * blockDim.x) {
    int c = i % Cpad, p = (i / Cpad) % HW, b = i / (Cpad * HW);
    float v = 0.f;
    if (c < C) {
      float a = acp[ts[b]];
      int64_t src = ((int64_t)b * C + c) * HW + p;
      v = sqrtf(a) * x0[src] + sqrtf(1.f - a) * noise[src];
      if (noisy_nchw) noisy_nchw[src] = v;
    }
    out[i] = f2bf(v);
  }
}

// ------------------------------------------------------------------ masked MSE (+min-SNR weights) fwd + bwd
// reference: trainer/loss.py:127-170 (compute_diffusion_loss) + :83-106 (compute_snr).
// pred is the conv_out result, NHWC fp32 [B*HW, ldp] (channels 0..C-1); noise/noisy/mask NCHW fp32.
// pass 1 (grid (B, S)): partial[b][s] = {sum e, sum mask} over the s-th slice of the C*HW elements of sample b (plain stores into
// sums[2*B + (b*S + s)*2 ...]); pass 2 adds the S partials of every sample in a fixed order in its prologue (the launch boundary
// publishes them) - one workgroup per sample walked 65,536 strided elements alone before: 136 us of a 52 ms step at batch 1.
__global__ __launch_bounds__(256) void mse_reduce_kernel(const float* pred, int64_t ldp, const float* noise, const float* noisy,
                                                          const float* mask, const int64_t* ts, const float* acp, int C, int HW,
                                                          int vpred, float* sums) {
  __shared__ float sh[8];
  const int b = blockIdx.x, S = gridDim.y, B = gridDim.x;
  float a = acp[ts[b]], sa = sqrtf(a), ss = sqrtf(1.f - a);
  float se = 0.f, sm = 0.f;
  for (int i = blockIdx.y * blockDim.x + threadIdx.x; i < C * HW; i += blockDim.x * S) {
    int c = i / HW, p = i - c * HW;
    int64_t src = ((int64_t)b * C + c) * HW + p;
    float tgt = vpred ? sa * noise[src] - ss * noisy[src] : noise[src];
    float d = pred[((int64_t)b * HW + p) * ldp + c] - tgt;
    float mk = mask[src];
    se += d * d * mk; sm += mk;
  }
  se = wave_sum(se); sm = wave_sum(sm);
  int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { sh[wave * 2] = se; sh[wave * 2 + 1] = sm; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float e = 0.f, m = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) { e += sh[w * 2]; m += sh[w * 2 + 1]; }
    float* part = sums + 2 * B + ((size_t)b * S + blockIdx.y) * 2;
    part[0] = e;
    part[1] = m;
  }
}
// per-sample coefficient coef_b such that loss = mean_b(coef_b * mean_chw(e)_b)
__device__ __forceinline__ float mse_coef(int b, int B, const float* sums, const int64_t* ts, const float* acp, float gamma, int vpred) {
  if (gamma > 0.f) {
    float wsum = 0.f, wb = 0.f;
    for (int i = 0; i < B; ++i) {
      float a = acp[ts[i]];
      float al = sqrtf(a), sg = sqrtf(1.f - a);
      float snr = (al / sg) * (al / sg);
      float w = fminf(snr, gamma) / snr + (vpred ? 1.f : 0.f);
      wsum += w;
      if (i == b) wb = w;
    }
    return wb / (wsum / (float)B);
  }
  float msum = 0.f;
  for (int i = 0; i < B; ++i) msum += sums[i * 2 + 1];
  return 1.f / (sums[b * 2 + 1] / (msum / (float)B));
}
// pass 2: loss scalar (block 0, thread 0) and dpred as bf16 NHWC padded to Cpad (input of conv_out's dX GEMM)
__global__ void mse_grad_kernel(const float* pred, int64_t ldp, const float* noise, const float* noisy, const float* mask,
                                const int64_t* ts, const float* acp, int B, int C, int HW, int Cpad, float gamma, int vpred,
                                float loss_scale, float* sums_io, int S, float* loss_out, bf16_t* dpred) {
  // prologue: sums[b] = {mean_chw e, mean_chw mask} from the S partials of pass 1, same order in every block (block 0 stores them)
  extern __shared__ float sums[];
  for (int t = threadIdx.x; t < 2 * B; t += blockDim.x) {
    const float* part = sums_io + 2 * B + ((size_t)(t >> 1) * S) * 2 + (t & 1);
    float acc = 0.f;
    for (int sp = 0; sp < S; ++sp) acc += part[sp * 2];
    acc /= (float)(C * HW);
    sums[t] = acc;
    if (blockIdx.x == 0) sums_io[t] = acc;
  }
  __syncthreads();
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    float l = 0.f;
    for (int b = 0; b < B; ++b) l += mse_coef(b, B, sums, ts, acp, gamma, vpred) * sums[b * 2];
    loss_out[0] = l / (float)B;
  }
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B * HW * Cpad; i += gridDim.x * blockDim.x) {
    int c = i % Cpad, p = (i / Cpad) % HW, b = i / (Cpad * HW);
    float g = 0.f;
    if (c < C) {
      float a = acp[ts[b]], sa = sqrtf(a), ss = sqrtf(1.f - a);
      int64_t src = ((int64_t)b * C + c) * HW + p;
      float tgt = vpred ? sa * noise[src] - ss * noisy[src] : noise[src];
      float d = pred[((int64_t)b * HW + p) * ldp + c] - tgt;
      float coef = mse_coef(b, B, sums, ts, acp, gamma, vpred);
      g = loss_scale * coef * 2.f * d * mask[src] / ((float)B * (float)(C * HW));
    }
    dpred[i] = f2bf(g);
  }
}

// ------------------------------------------------------------------ fused AdamW (+ L1 subgradient), fp32 master params
// reference: torch.optim.AdamW built at trainer/optimizer.py:18 / :113-150, stepped at optimizer.py:270-275,
// L1 penalty main.py:353-356 (grad of l1w * sum|p| / N is l1w*sign(p)/N, folded in here).
// hyper (device, fp32): [0] lr  [1] beta1  [2] beta2  [3] eps  [4] weight_decay  [5] 1-beta1^t  [6] 1-beta2^t
//                       [7] l1 coefficient (= l1_penalty * loss_scale / N_total)  [8] grad scale
__global__ void adamw_kernel(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper, float* l1_partial, int vec) {
  const float lr = hyper[0], b1 = hyper[1], b2 = hyper[2], eps = hyper[3], wd = hyper[4], bc1 = hyper[5], bc2 = hyper[6],
              l1c = hyper[7], gs = hyper[8];
  const float rs2 = rsqrtf(bc2), step = lr / bc1;
  float l1 = 0.f;
  auto upd = [&](float& pi, const float g_, float& mi, float& vi) {      // torch.optim.AdamW (decoupled decay) + the L1 subgradient of main.py:353-356
    l1 += fabsf(pi);
    const float gi = g_ * gs + l1c * (pi > 0.f ? 1.f : (pi < 0.f ? -1.f : 0.f));
    mi = b1 * mi + (1.f - b1) * gi;
    vi = b2 * vi + (1.f - b2) * gi * gi;
    pi = pi * (1.f - lr * wd) - step * mi / (sqrtf(vi) * rs2 + eps);
  };
  // 16 bytes per lane and stream (seven streams: the kernel is pure HBM traffic); the arenas are 16-byte aligned (checked by the launcher), the
  // last n % 4 elements go one by one
  const int64_t n4 = vec ? n >> 2 : 0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    f32x4 pv = ((const f32x4*)p)[i], mv = ((const f32x4*)m)[i], vv = ((const f32x4*)v)[i];
    const f32x4 gv = ((const f32x4*)g)[i];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float pi = pv[k], mi = mv[k], vi = vv[k];
      upd(pi, gv[k], mi, vi);
      pv[k] = pi; mv[k] = mi; vv[k] = vi;
    }
    ((f32x4*)p)[i] = pv; ((f32x4*)m)[i] = mv; ((f32x4*)v)[i] = vv;
  }
  for (int64_t i = (n4 << 2) + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float pi = p[i], mi = m[i], vi = v[i];
    upd(pi, g[i], mi, vi);
    p[i] = pi; m[i] = mi; v[i] = vi;
  }
  if (l1_partial) {      // one atomic per workgroup (16384 wave-level atomics onto one address took longer than the kernel's HBM traffic: 235 us for 713 MB)
    __shared__ float l1w[4];
    l1 = wave_sum(l1);
    if ((threadIdx.x & 63) == 0) l1w[threadIdx.x >> 6] = l1;
    __syncthreads();
    if (threadIdx.x == 0) {
      float t = 0.f;
      for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += l1w[w];
      atomicAdd(l1_partial, t);
    }
  }
}

// ------------------------------------------------------------------ bf16 compute copies (both orientations) of an fp32 master arena
// one descriptor per parameter tensor [rows, cols] (row stride src_ld in the arena):
//   dst  [rows, ld ]  <- same orientation        dstT [cols, ldT] <- transposed
// One workgroup per 64x64 tile: the fp32 rows are read as 256-B lines, converted once, and both copies leave as
// contiguous runs along their own fast axis (the transposed one through LDS).  Used for the LoRA adapters (a few MB) and
// for every UNet weight of the full fine-tune (2.57 G parameters: 10 GB read, 2 x 5 GB written per step).
// ADAMW: the same tiles also carry the optimizer step (p, g, m, v read, p, m, v written) before the conversion - one pass over
// the 2.57 G parameters of the full fine-tune instead of AdamW's 7 words + the refresh's re-read of p.
template <bool ADAMW>
__global__ __launch_bounds__(256) void shadow_kernel(const sdlt_shadow_desc* descs, const int32_t* block_desc, const int32_t* block_first,
                                                     float* arena, const float* g, float* m, float* v, const float* hyper) {
  __shared__ bf16_t tile[64][72];
  const sdlt_shadow_desc d = descs[block_desc[blockIdx.x]];
  const int t = blockIdx.x - block_first[block_desc[blockIdx.x]];
  const int tiles_c = (d.cols + 63) >> 6;
  const int r0 = (t / tiles_c) << 6, c0 = (t % tiles_c) << 6;
  const int thr = threadIdx.x;
  {
    const int c = c0 + (thr & 63);
#pragma unroll 4
    for (int k = 0; k < 16; ++k) {
      const int r = r0 + k * 4 + (thr >> 6);
      float val = 0.f;
      if (r < d.rows && c < d.cols) {
        const int64_t i = d.offset + (int64_t)r * d.src_ld + c;
        val = arena[i];
        if (ADAMW) {          // torch.optim.AdamW, exactly as adamw_kernel (no L1 term: main.py:353 applies it to LoRA tensors only)
          const float lr = hyper[0], b1 = hyper[1], b2 = hyper[2], eps = hyper[3], wd = hyper[4], bc1 = hyper[5], bc2 = hyper[6], gs = hyper[8];
          const float gi = g[i] * gs;
          const float mi = b1 * m[i] + (1.f - b1) * gi;
          const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
          val = val * (1.f - lr * wd) - (lr / bc1) * mi / (sqrtf(vi) * rsqrtf(bc2) + eps);
          arena[i] = val; m[i] = mi; v[i] = vi;
        }
      }
      tile[k * 4 + (thr >> 6)][thr & 63] = f2bf(val);
    }
  }
  __syncthreads();
  const int q = thr >> 2, ch = (thr & 3) * 16;
  if (d.dst) {
    const int r = r0 + q;
    if (r < d.rows) {
      bf16_t* p = (bf16_t*)d.dst + (int64_t)r * d.ld + c0 + ch;
      if (c0 + ch + 16 <= d.cols && (((uintptr_t)p) & 15) == 0) {
        *(uint4*)p = *(const uint4*)&tile[q][ch];
        *(uint4*)(p + 8) = *(const uint4*)&tile[q][ch + 8];
      } else {
        for (int j = 0; j < 16 && c0 + ch + j < d.cols; ++j) p[j] = tile[q][ch + j];
      }
    }
  }
  if (d.dstT) {
    const int c = c0 + q;
    if (c < d.cols) {
      bf16_t v[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] = tile[ch + j][q];
      bf16_t* p = (bf16_t*)d.dstT + (int64_t)c * d.ldT + r0 + ch;
      if (r0 + ch + 16 <= d.rows && (((uintptr_t)p) & 15) == 0) {
        *(uint4*)p = *(const uint4*)&v[0];
        *(uint4*)(p + 8) = *(const uint4*)&v[8];
      } else {
        for (int j = 0; j < 16 && r0 + ch + j < d.rows; ++j) p[j] = v[j];
      }
    }
  }
}

// ------------------------------------------------------------------ AdamW8bit: block-quantised moments, fused into the refresh tiles
// bitsandbytes 0.43.1 `AdamW8bit` (trainer/optimizer.py:19-21; the optimizer full_finetuning_example.json names), restated from its published blockwise
// 8-bit Adam (kOptimizerStatic8bit2StateBlockwise): both moments live as one byte per element - an index into a 256-entry code book ("dynamic" map, signed
// for m, unsigned for v) - times one fp32 absmax per block of 2048 elements.  Per step and block: dequantise, m = b1 m + (1 - b1) g, v = b2 v + (1 - b2) g^2,
// new absmax = max |m|, max v over the block, p += -lr sqrt(bc2) / bc1 * m / (sqrt(v) + sqrt(bc2) eps), then p *= 1 - lr wd, requantise m / absmax and
// v / absmax to the nearest code (m keeps its sign: a code of the other sign moves one step towards it).
// Here a block is one half of a 64 x 64 refresh tile (32 rows x 64 columns = 2048 elements of the weight's [N, K] view) instead of 2048 consecutive elements of
// the flattened tensor: the tile is the unit this pass already owns (fp32 master in, W and W^T bf16 out), so the whole optimizer step stays ONE pass
// - p, g 4 B + m, v 1 B in, p 4 B + m, v 1 B + two bf16 copies out = 20 B per parameter instead of 32 B with fp32 moments.
// m8 / v8: tile-major, uint8 [n_blocks][64][64] (tile b of the descriptor table, element (r, c) of the tile at 4096 b + 64 r + c; positions outside the tensor unused).
// tables (fp32, device): q1[256] | mid1[256] | q2[256] | mid2[256], mid[k] = (q[k] + q[k + 1]) / 2, mid[255] = +inf.  absmax: [n_blocks][4] = {m rows 0-31, m rows 32-63,
// v rows 0-31, v rows 32-63}.  The nearest code is found from the code book's structure (decade i holds 2^i (signed) / 2^(i+1) (unsigned) equally spaced values in
// 10^(i-6) [0.1, 1]: q8_guess) and corrected by at most one step against the two neighbouring midpoints (q8_fix): two LDS reads instead of bnb's eight-step search.
template <bool SIGNED>
__device__ __forceinline__ int q8_guess(float x) {
  const float a = fabsf(x);
  const float L = __builtin_amdgcn_logf(fmaxf(a, 1e-7f)) * 0.30102999566f + 7.f;          // log10(a) + 7 in [0, 7]
  int i = (int)L;
  i = i > 6 ? 6 : i;
  const int n = SIGNED ? (1 << i) : (2 << i);
  // cell j = floor((a 10^(6-i) - 0.1) n / 0.9) = floor(a K - n / 9), K = 10^(6-i) 2^i / 0.9 = exp2(log2(1e6 / 0.9) - i log2(5))   (unsigned book: twice the cells)
  const float K = __builtin_amdgcn_exp2f((SIGNED ? 20.083571662f : 21.083571662f) - (float)i * 2.321928095f);
  int j = (int)(a * K - ldexpf(SIGNED ? (1.f / 9.f) : (2.f / 9.f), i));
  j = j < 0 ? 0 : (j > n - 1 ? n - 1 : j);
  const int pos = n - 1 + j;
  return SIGNED ? (x >= 0.f ? 128 + pos : 126 - pos) : pos;               // 0 <= c <= 254: the nearest code or one of its neighbours
}
// ... corrected against the midpoints on either side of the guess (up = mid[c], dn = mid[max(c - 1, 0)]); plain integer arithmetic on the comparisons, so that the
// compiler keeps the two LDS reads unconditional and batched instead of turning a select into a divergent branch per element
__device__ __forceinline__ int q8_fix(int c, float x, float up, float dn) { return c + (int)(x > up) - (int)((c > 0) & (x <= dn)); }

// FULL: the tile lies inside the tensor (no row / column tests, no clamped addresses)
template <bool FULL>
__device__ __forceinline__ void adamw8_tile(const sdlt_shadow_desc& d, int r0, int c0, float* arena, const float* g, uint8_t* m8, uint8_t* v8, float* absmax_blk,
                                            const float* hyper, const float* tb, float (*red)[4], bf16_t (*tile)[72]) {
  const int thr = threadIdx.x, wave = thr >> 6, lane = thr & 63;
  const float4 am = *(const float4*)absmax_blk;
  const float lr = hyper[0], b1 = hyper[1], b2 = hyper[2], eps = hyper[3], wd = hyper[4], bc1 = hyper[5], bc2 = hyper[6], gs = hyper[8];
  const float c2 = sqrtf(bc2), step_size = -lr * c2 / bc1, eps2 = c2 * eps, decay = wd > 0.f ? 1.f - lr * wd : 1.f;
  const int c = c0 + lane;
  // element (row r0 + 4 k + wave, column c) = tile origin (uniform: a scalar base address) + a 32-bit per-lane offset
  const int64_t org = d.offset + (int64_t)r0 * d.src_ld + c0;
  float* pa = arena + org;
  const float* pg = g + org;
  // codes: tile-major (4096 per tile, row r of the tile at 64 r): a wave's requests are whole 128-byte lines whatever the tensor's row stride
  uint8_t* pm = m8 + (int64_t)blockIdx.x * 4096;
  uint8_t* pq = v8 + (int64_t)blockIdx.x * 4096;
  const uint32_t lo = (uint32_t)wave * (uint32_t)d.src_ld + lane, rstep = 4u * (uint32_t)d.src_ld;
  const uint32_t clo = (uint32_t)wave * 64u + lane;          // (element (4 k + wave, lane) of the tile)
  // all 64 requests of the lane first (unconditional; ragged tiles read element 0 of the tensor instead), then the arithmetic
  float pv[16], gv[16];
  uint8_t cm[16], cv[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const bool ok = FULL || (r0 + k * 4 + wave < d.rows && c < d.cols);
    const uint32_t i = ok ? lo + k * rstep : 0u;
    pv[k] = pa[i]; gv[k] = pg[i]; cm[k] = pm[clo + k * 256]; cv[k] = pq[clo + k * 256];
  }
  __syncthreads();                                                  // the tables are in LDS
  float mn[16], vn[16], mx[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int h = k >> 3;
    const bool ok = FULL || (r0 + k * 4 + wave < d.rows && c < d.cols);
    const float gi = gv[k] * gs;
    float mi = b1 * (tb[cm[k]] * (h ? am.y : am.x)) + (1.f - b1) * gi;
    float vi = b2 * (tb[512 + cv[k]] * (h ? am.w : am.z)) + (1.f - b2) * gi * gi;
    if (!ok) mi = vi = 0.f;
    mn[k] = mi; vn[k] = vi;
    mx[h] = fmaxf(mx[h], fabsf(mi));
    mx[2 + h] = fmaxf(mx[2 + h], vi);
    // (hardware square root and reciprocal, 1 ulp each - bnb's own kernel divides with __fdividef)
    const float val = (pv[k] + step_size * (mi * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(vi) + eps2))) * decay;
    if (ok) pa[lo + k * rstep] = val;
    tile[k * 4 + wave][lane] = f2bf(ok ? val : 0.f);
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) mx[u] = wave_max(mx[u]);
  if (lane == 0) *(float4*)&red[wave][0] = (float4){mx[0], mx[1], mx[2], mx[3]};
  __syncthreads();                                                  // (also: the bf16 tile is complete)
  float inv[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    mx[u] = fmaxf(fmaxf(red[0][u], red[1][u]), fmaxf(red[2][u], red[3][u]));
    inv[u] = mx[u] > 0.f ? __builtin_amdgcn_rcpf(mx[u]) : 0.f;
  }
  if (thr == 0) *(float4*)absmax_blk = (float4){mx[0], mx[1], mx[2], mx[3]};
#pragma unroll
  for (int h = 0; h < 2; ++h) {                                     // one block of 2048 (8 elements per lane) at a time: guesses, then ALL midpoint reads, then the codes
    float x1[8], x2[8], u1[8], d1[8], u2[8], d2[8];
    int g1[8], g2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      x1[e] = mn[h * 8 + e] * inv[h];
      x2[e] = vn[h * 8 + e] * inv[2 + h];
      g1[e] = q8_guess<true>(x1[e]);
      g2[e] = q8_guess<false>(x2[e]);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      u1[e] = tb[256 + g1[e]]; d1[e] = tb[256 + (g1[e] > 0 ? g1[e] - 1 : 0)];
      u2[e] = tb[768 + g2[e]]; d2[e] = tb[768 + (g2[e] > 0 ? g2[e] - 1 : 0)];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = h * 8 + e;
      int q1 = q8_fix(g1[e], x1[e], u1[e], d1[e]);
      q1 += ((q1 < 127) != (mn[k] < 0.f)) ? (mn[k] > 0.f ? 1 : -1) : 0;   // m keeps its sign (codes below 127 are the negative ones; the code of 0 counts as positive)
      const int q2 = q8_fix(g2[e], x2[e], u2[e], d2[e]);
      if (FULL || (r0 + k * 4 + wave < d.rows && c < d.cols)) { pm[clo + k * 256] = (uint8_t)q1; pq[clo + k * 256] = (uint8_t)q2; }
    }
  }
}

// The same tile with 16-byte accesses (interior tiles of tensors whose rows are 16-byte aligned - every tile of the SDXL UNet but the ragged edges): a lane owns four consecutive
// columns of four rows (row = 16 k + 4 wave + lane / 16), i.e. one float4 of p and g and one dword of codes per row instead of four scalar requests each - 28 memory
// instructions per lane instead of 112, and a quarter of the address arithmetic.
__device__ __forceinline__ void adamw8_tile_vec(const sdlt_shadow_desc& d, int r0, int c0, float* arena, const float* g, uint8_t* m8, uint8_t* v8, float* absmax_blk,
                                                const float* hyper, const float* tb, float (*red)[4], bf16_t (*tile)[72]) {
  const int thr = threadIdx.x, wave = thr >> 6, lane = thr & 63, cq = lane & 15, rq = lane >> 4;
  const float4 am = *(const float4*)absmax_blk;
  const float lr = hyper[0], b1 = hyper[1], b2 = hyper[2], eps = hyper[3], wd = hyper[4], bc1 = hyper[5], bc2 = hyper[6], gs = hyper[8];
  const float c2 = sqrtf(bc2), step_size = -lr * c2 / bc1, eps2 = c2 * eps, decay = wd > 0.f ? 1.f - lr * wd : 1.f;
  const int64_t org = d.offset + (int64_t)r0 * d.src_ld + c0;
  float* pa = arena + org;
  const float* pg = g + org;
  uint8_t* pm = m8 + (int64_t)blockIdx.x * 4096;      // tile-major codes: row r of the tile at 64 r (a wave's four rows are 256 contiguous bytes)
  uint8_t* pq = v8 + (int64_t)blockIdx.x * 4096;
  const uint32_t lo = (uint32_t)(wave * 4 + rq) * (uint32_t)d.src_ld + 4u * cq, rstep = 16u * (uint32_t)d.src_ld;
  const uint32_t clo = (uint32_t)(wave * 4 + rq) * 64u + 4u * cq;
  float4 pv[4], gv[4];
  uint32_t cm[4], cv[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint32_t i = lo + k * rstep;
    pv[k] = *(const float4*)(pa + i); gv[k] = *(const float4*)(pg + i);
    cm[k] = *(const uint32_t*)(pm + clo + k * 1024); cv[k] = *(const uint32_t*)(pq + clo + k * 1024);
  }
  __syncthreads();                                                  // the tables are in LDS
  float mn[16], vn[16], mx[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int h = k >> 1;
    float val[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float gi = gv[k][e] * gs;
      const float mi = b1 * (tb[(cm[k] >> (8 * e)) & 255u] * (h ? am.y : am.x)) + (1.f - b1) * gi;
      const float vi = b2 * (tb[512 + ((cv[k] >> (8 * e)) & 255u)] * (h ? am.w : am.z)) + (1.f - b2) * gi * gi;
      mn[k * 4 + e] = mi; vn[k * 4 + e] = vi;
      mx[h] = fmaxf(mx[h], fabsf(mi));
      mx[2 + h] = fmaxf(mx[2 + h], vi);
      val[e] = (pv[k][e] + step_size * (mi * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(vi) + eps2))) * decay;
    }
    *(float4*)(pa + lo + k * rstep) = (float4){val[0], val[1], val[2], val[3]};
    *(uint2*)&tile[k * 16 + wave * 4 + rq][4 * cq] = (uint2){pack2bf(val[0], val[1]), pack2bf(val[2], val[3])};
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) mx[u] = wave_max(mx[u]);
  if (lane == 0) *(float4*)&red[wave][0] = (float4){mx[0], mx[1], mx[2], mx[3]};
  __syncthreads();                                                  // (also: the bf16 tile is complete)
  float inv[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    mx[u] = fmaxf(fmaxf(red[0][u], red[1][u]), fmaxf(red[2][u], red[3][u]));
    inv[u] = mx[u] > 0.f ? __builtin_amdgcn_rcpf(mx[u]) : 0.f;
  }
  if (thr == 0) *(float4*)absmax_blk = (float4){mx[0], mx[1], mx[2], mx[3]};
#pragma unroll
  for (int k = 0; k < 4; ++k) {                                     // one row (four codes of each moment = one dword each) at a time: guesses, then all eight midpoint reads, then the codes
    const int h = k >> 1;
    float x1[4], x2[4], u1[4], d1[4], u2[4], d2[4];
    int g1[4], g2[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      x1[e] = mn[k * 4 + e] * inv[h];
      x2[e] = vn[k * 4 + e] * inv[2 + h];
      g1[e] = q8_guess<true>(x1[e]);
      g2[e] = q8_guess<false>(x2[e]);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      u1[e] = tb[256 + g1[e]]; d1[e] = tb[256 + (g1[e] > 0 ? g1[e] - 1 : 0)];
      u2[e] = tb[768 + g2[e]]; d2[e] = tb[768 + (g2[e] > 0 ? g2[e] - 1 : 0)];
    }
    uint32_t w1 = 0, w2 = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      int q1 = q8_fix(g1[e], x1[e], u1[e], d1[e]);
      q1 += ((q1 < 127) != (mn[k * 4 + e] < 0.f)) ? (mn[k * 4 + e] > 0.f ? 1 : -1) : 0;
      w1 |= (uint32_t)q1 << (8 * e);
      w2 |= (uint32_t)q8_fix(g2[e], x2[e], u2[e], d2[e]) << (8 * e);
    }
    *(uint32_t*)(pm + clo + k * 1024) = w1; *(uint32_t*)(pq + clo + k * 1024) = w2;
  }
}

// The same update on a FLAT range (the owned slices of the sharded optimizer, data-parallel full fine-tune): one workgroup per block of 2048 CONSECUTIVE elements - bitsandbytes' own
// partition, counted from the start of the range -, eight elements per lane (two float4 of p and g, two dwords of codes), no operand refresh (the masters are all-gathered first).
// absmax: fp32 [ceil(n / 2048)][2] = {m, v}.  n % 4 == 0; the last block may be short.
__global__ __launch_bounds__(256) void adamw8_flat_kernel(float* p, const float* g, uint8_t* m8, uint8_t* v8, float* absmax, int64_t n, const float* tables, const float* hyper) {
  __shared__ float tb[1024];
  __shared__ float red[4][2];
  const int thr = threadIdx.x, wave = thr >> 6, lane = thr & 63;
  ((uint4*)tb)[thr] = ((const uint4*)tables)[thr];
  const int64_t e0 = (int64_t)blockIdx.x * 2048 + thr * 8;
  const float2 am = *(const float2*)(absmax + 2 * (int64_t)blockIdx.x);
  const float lr = hyper[0], b1 = hyper[1], b2 = hyper[2], eps = hyper[3], wd = hyper[4], bc1 = hyper[5], bc2 = hyper[6], gs = hyper[8];
  const float c2 = sqrtf(bc2), step_size = -lr * c2 / bc1, eps2 = c2 * eps, decay = wd > 0.f ? 1.f - lr * wd : 1.f;
  float4 pv[2], gv[2];
  uint32_t cm[2], cv[2];
  bool ok[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {      // (unconditional requests; a quad past the end reads the range's first elements and is dropped)
    ok[k] = e0 + 4 * k < n;
    const int64_t i = ok[k] ? e0 + 4 * k : 0;
    pv[k] = *(const float4*)(p + i); gv[k] = *(const float4*)(g + i);
    cm[k] = *(const uint32_t*)(m8 + i); cv[k] = *(const uint32_t*)(v8 + i);
  }
  __syncthreads();
  float mn[8], vn[8], mxm = 0.f, mxv = 0.f;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    float val[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float gi = gv[k][e] * gs;
      float mi = b1 * (tb[(cm[k] >> (8 * e)) & 255u] * am.x) + (1.f - b1) * gi;
      float vi = b2 * (tb[512 + ((cv[k] >> (8 * e)) & 255u)] * am.y) + (1.f - b2) * gi * gi;
      if (!ok[k]) mi = vi = 0.f;
      mn[k * 4 + e] = mi; vn[k * 4 + e] = vi;
      mxm = fmaxf(mxm, fabsf(mi));
      mxv = fmaxf(mxv, vi);
      val[e] = (pv[k][e] + step_size * (mi * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(vi) + eps2))) * decay;
    }
    if (ok[k]) *(float4*)(p + e0 + 4 * k) = (float4){val[0], val[1], val[2], val[3]};
  }
  mxm = wave_max(mxm);
  mxv = wave_max(mxv);
  if (lane == 0) { red[wave][0] = mxm; red[wave][1] = mxv; }
  __syncthreads();
  mxm = fmaxf(fmaxf(red[0][0], red[1][0]), fmaxf(red[2][0], red[3][0]));
  mxv = fmaxf(fmaxf(red[0][1], red[1][1]), fmaxf(red[2][1], red[3][1]));
  if (thr == 0) *(float2*)(absmax + 2 * (int64_t)blockIdx.x) = make_float2(mxm, mxv);
  const float im = mxm > 0.f ? __builtin_amdgcn_rcpf(mxm) : 0.f, iv = mxv > 0.f ? __builtin_amdgcn_rcpf(mxv) : 0.f;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    float x1[4], x2[4], u1[4], d1[4], u2[4], d2[4];
    int g1[4], g2[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      x1[e] = mn[k * 4 + e] * im;
      x2[e] = vn[k * 4 + e] * iv;
      g1[e] = q8_guess<true>(x1[e]);
      g2[e] = q8_guess<false>(x2[e]);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      u1[e] = tb[256 + g1[e]]; d1[e] = tb[256 + (g1[e] > 0 ? g1[e] - 1 : 0)];
      u2[e] = tb[768 + g2[e]]; d2[e] = tb[768 + (g2[e] > 0 ? g2[e] - 1 : 0)];
    }
    uint32_t w1 = 0, w2 = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      int q1 = q8_fix(g1[e], x1[e], u1[e], d1[e]);
      q1 += ((q1 < 127) != (mn[k * 4 + e] < 0.f)) ? (mn[k * 4 + e] > 0.f ? 1 : -1) : 0;
      w1 |= (uint32_t)q1 << (8 * e);
      w2 |= (uint32_t)q8_fix(g2[e], x2[e], u2[e], d2[e]) << (8 * e);
    }
    if (ok[k]) { *(uint32_t*)(m8 + e0 + 4 * k) = w1; *(uint32_t*)(v8 + e0 + 4 * k) = w2; }
  }
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void shadow_adamw8_kernel(const sdlt_shadow_desc* descs, const int32_t* block_desc, const int32_t* block_first, float* arena, const float* g,
                                                            uint8_t* m8, uint8_t* v8, float* absmax, const float* tables, const float* hyper) {
  __shared__ bf16_t tile[64][72];
  __shared__ float tb[1024];
  __shared__ float red[4][4];
  const sdlt_shadow_desc d = descs[block_desc[blockIdx.x]];
  const int t = blockIdx.x - block_first[block_desc[blockIdx.x]];
  const int tiles_c = (d.cols + 63) >> 6;
  const int r0 = (t / tiles_c) << 6, c0 = (t % tiles_c) << 6;
  const int thr = threadIdx.x;
  ((uint4*)tb)[thr] = ((const uint4*)tables)[thr];
  const bool full = r0 + 64 <= d.rows && c0 + 64 <= d.cols;
  if (full && ((d.offset | d.src_ld) & 3) == 0 && ((((uintptr_t)arena | (uintptr_t)g) & 15) | (((uintptr_t)m8 | (uintptr_t)v8) & 3)) == 0)
    adamw8_tile_vec(d, r0, c0, arena, g, m8, v8, absmax + 4 * (int64_t)blockIdx.x, hyper, tb, red, tile);
  else if (full) adamw8_tile<true>(d, r0, c0, arena, g, m8, v8, absmax + 4 * (int64_t)blockIdx.x, hyper, tb, red, tile);
  else adamw8_tile<false>(d, r0, c0, arena, g, m8, v8, absmax + 4 * (int64_t)blockIdx.x, hyper, tb, red, tile);
  const int q = thr >> 2, ch = (thr & 3) * 16;
  if (d.dst) {
    const int r = r0 + q;
    if (r < d.rows) {
      bf16_t* p = (bf16_t*)d.dst + (int64_t)r * d.ld + c0 + ch;
      if (c0 + ch + 16 <= d.cols && (((uintptr_t)p) & 15) == 0) {
        *(uint4*)p = *(const uint4*)&tile[q][ch];
        *(uint4*)(p + 8) = *(const uint4*)&tile[q][ch + 8];
      } else {
        for (int j = 0; j < 16 && c0 + ch + j < d.cols; ++j) p[j] = tile[q][ch + j];
      }
    }
  }
  if (d.dstT) {
    const int cc = c0 + q;
    if (cc < d.cols) {
      bf16_t v[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] = tile[ch + j][q];
      bf16_t* p = (bf16_t*)d.dstT + (int64_t)cc * d.ldT + r0 + ch;
      if (r0 + ch + 16 <= d.rows && (((uintptr_t)p) & 15) == 0) {
        *(uint4*)p = *(const uint4*)&v[0];
        *(uint4*)(p + 8) = *(const uint4*)&v[8];
      } else {
        for (int j = 0; j < 16 && r0 + ch + j < d.rows; ++j) p[j] = v[j];
      }
    }
  }
}

// ------------------------------------------------------------------ out = a + b on strided [M,C] views
__global__ void add2d_kernel(const bf16_t* a, int64_t lda, const bf16_t* b, int64_t ldb, bf16_t* out, int64_t ldo, int M, int C) {
  const int nch = C >> 3;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < (int64_t)M * nch; i += (int64_t)gridDim.x * blockDim.x) {
    int m = i / nch, c = (i - (int64_t)m * nch) * 8;
    float x[8], y[8];
    load8(a + m * lda + c, x);
    load8(b + m * ldb + c, y);
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] += y[j];
    store8(out + m * ldo + c, x);
  }
}

// ------------------------------------------------------------------ dX of nearest-2x upsampling: sum of each 2x2 block
__global__ void sum2x2_kernel(const bf16_t* in, int B, int H, int W, int C, bf16_t* out) {
  const int nch = C >> 3;
  const int64_t total = (int64_t)B * H * W * nch;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int c = (i % nch) * 8;
    int64_t pix = i / nch;
    int w = pix % W, h = (pix / W) % H, b = pix / ((int64_t)W * H);
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        float v[8];
        load8(in + (((int64_t)b * 2 * H + 2 * h + dy) * 2 * W + 2 * w + dx) * C + c, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += v[j];
      }
    store8(out + pix * C + c, acc);
  }
}

// ------------------------------------------------------------------ column sums per batch: out[b, c] = sum_r x[b*R + r, c]
// grid (C/64, rowsplit, B); same 8 rows x 64 channels wave footprint as the norms.  Every row split leaves its partial sums in its
// own row of an fp32 scratch ws[split][B*C] (plain stores: no atomics, no zero fill); the second kernel adds the rows in a fixed
// order (bitwise reproducible) and writes the fp32 and / or bf16 result.
__global__ __launch_bounds__(256) void colsum_kernel(const bf16_t* x, int64_t ldx, int R, int C, float* ws) {
  __shared__ float sh[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.z, c0 = blockIdx.x * 64 + (lane & 7) * 8;
  float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int r = blockIdx.y * 32 + wave * 8 + (lane >> 3); r < R; r += gridDim.y * 32) {
    float v[8];
    load8(x + ((int64_t)b * R + r) * ldx + c0, v);
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] += v[j];
  }
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int o = 8; o < 64; o <<= 1) s[j] += __shfl_xor(s[j], o, 64);
  if ((lane >> 3) == 0)
#pragma unroll
    for (int j = 0; j < 8; ++j) sh[wave][(lane & 7) * 8 + j] = s[j];
  __syncthreads();
  if (threadIdx.x < 64)
    ws[((int64_t)blockIdx.y * gridDim.z + b) * C + blockIdx.x * 64 + threadIdx.x] = (sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x]);
}
__global__ void colsum_finish_kernel(const float* ws, int nsplit, float* out32, bf16_t* out16, int n) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    float a = 0.f;
    for (int sp = 0; sp < nsplit; ++sp) a += ws[(int64_t)sp * n + i];
    if (out32) out32[i] = a;
    if (out16) out16[i] = f2bf(a);
  }
}

inline int grid_for(int64_t work_items, int block = 256) {
  int64_t g = (work_items + block - 1) / block;
  return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

}  // namespace

extern "C" int sdlt_geglu_fwd(const void* in, int64_t ldin, int32_t M, int32_t Ch, void* out, int64_t ldout, void* stream) {
  if (M <= 0 || Ch <= 0 || (Ch % 8) || (ldin % 8) || (ldout % 8)) SDLT_FAIL(SDLT_ERR_SHAPE, "sdlt_geglu_fwd: M=%d Ch=%d", M, Ch);
  hipLaunchKernelGGL(geglu_fwd_kernel, dim3(grid_for((int64_t)M * Ch / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)in, ldin, M, Ch, (bf16_t*)out, ldout);
  SDLT_CHECK_LAUNCH();
  return SDLT_OK;
}
extern "C" int sdlt_geglu_bwd(const void* in, int64_t ldin, const void* dout, int64_t lddout, int32_t M, int32_t Ch, void* din,
                              int64_t lddin, void* stream) {
  if (M <= 0 || Ch <= 0 || (Ch % 8) || (ldin % 8) || (lddout % 8) || (lddin % 8)) SDLT_FAIL(SDLT_ERR_SHAPE, "sdlt_geglu_bwd: M=%d Ch=%d", M, Ch);
  hipLaunchKernelGGL(geglu_bwd_kernel, dim3(grid_for((int64_t)M * Ch / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)in, ldin, (const bf16_t*)dout, lddout, M, Ch, (bf16_t*)din, lddin);
  SDLT_CHECK_LAUNCH();
  return SDLT_OK;
}
extern "C" int sdlt_map_bf16(int32_t op, const void* x, const void* dy, void* y, int64_t n, void* stream) {
  if (n <= 0 || (n % 8) || op < 0 || op > 6) SDLT_FAIL(SDLT_ERR_SHAPE, "sdlt_map_bf16: op=%d n=%lld (n %% 8 == 0)", op, (long long)n);
  if ((op == 1 || op == 2 || op == 4 || op == 6) && !dy) SDLT_FAIL(SDLT_ERR_SHAPE, "sdlt_map_bf16: op %d needs dy", op);
  hipLaunchKernelGGL(map_kernel, dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)stream, op, (const bf16_t*)x, (const bf16_t*)dy, (bf16_t*)y, n / 8);
  SDLT_CHECK_LAUNCH();
  return SDLT_OK;
}
extern "C" int sdlt_timestep_embedding(const float* t, int32_t rows, int32_t dim, void* out, int64_t ldo, void* stream) {
  if (rows <= 0 || dim <= 0 || (dim & 1)) SDLT_FAIL(SDLT_ERR_SHAPE, "sdlt_timestep_embedding: rows=%d dim=%d", rows, dim);
  hipLaunchKernelGGL(timestep_embed_kernel, dim3(grid_for((int64_t)rows * dim / 2)), dim3(256), 0, (hipStream_t)stream, t, rows, dim, (bf16_t*)out, ldo);
  SDLT_CHECK_LAUNCH();
  return SDLT_OK;
}
extern "C" int sdlt_add_noise_nhwc(const float* x0, const float* noise, const int64_t* timesteps, const float* alphas_cumprod,
                                   int32_t B, int32_t C, int32_t HW, int32_t Cpad, void* out_nhwc, float* noisy_nchw, void* stream) {
  if (B <= 0 || C <= 0 || HW <= 0 || Cpad < C) SDLT_FAIL(SDLT_ERR_SHAPE, "sdlt_add_noise_nhwc: B=%d C=%d HW=%d Cpad=%d", B, C, HW, Cpad);
  hipLaunchKernelGGL(add_noise_kernel, dim3(grid_for((int64_t)B * HW * Cpad)), dim3(256), 0, (hipStream_t)stream, x0, noise, timesteps, alphas_cumprod, B, C, HW, Cpad, (bf16_t*)out_nhwc, noisy_nchw);
  SDLT_CHECK_LAUNCH();
  return SDLT_OK;
}
extern "C" int sdlt_masked_mse_fwd_bwd(const float* pred, int64_t ldp, const float* noise, const float* noisy, const float* mask,
                                       const int64_t* timesteps, const float* alphas_cumprod, int32_t B, int32_t C, int32_t HW,
                                       int32_t Cpad, float snr_gamma, int32_t v_prediction, float loss_scale, float* sums,
                                       int32_t sums_floats, float* loss_out, void* dpred, void* stream) {
  if (B <= 0 || B > 1024 || C <= 0 || HW <= 0 || Cpad < C) SDLT_FAIL(SDLT_ERR_SHAPE, "sdlt_masked_mse_fwd_bwd: B=%d C=%d HW=%d Cpad=%d", B, C, HW, Cpad);
  if (v_prediction && !noisy) SDLT_FAIL(SDLT_ERR_SHAPE, "sdlt_masked_mse_fwd_bwd: v-prediction needs the noisy latent");
  hipStream_t s = (hipStream_t)stream;
  int S = (C * HW + 1023) / 1024;
  if (S > 64) S = 64;
  if (sums_floats < 2 * B * (1 + S)) SDLT_FAIL(SDLT_ERR_SHAPE, "sdlt_masked_mse_fwd_bwd: sums needs %d floats (2*B finals + 2*B*%d partials)", 2 * B * (1 + S), S);
  hipLaunchKernelGGL(mse_reduce_kernel, dim3(B, S), dim3(256), 0, s, pred, ldp, noise, noisy, mask, timesteps, alphas_cumprod, C, HW, v_prediction, sums);
  hipLaunchKernelGGL(mse_grad_kernel, dim3(grid_for((int64_t)B * HW * Cpad)), dim3(256), sizeof(float) * 2 * B, s, pred, ldp, noise, noisy, mask, timesteps, alphas_cumprod, B, C, HW, Cpad, snr_gamma, v_prediction, loss_scale, sums, S, loss_out, (bf16_t*)dpred);
  SDLT_CHECK_LAUNCH();
  return SDLT_OK;
}
extern "C" int sdlt_adamw_fused(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper, float* l1_sum,
                                void* stream) {
  if (n <= 0) SDLT_FAIL(SDLT_ERR_SHAPE, "sdlt_adamw_fused: n=%lld", (long long)n);
  hipStream_t s = (hipStream_t)stream;
  if (l1_sum) sdlt_zero_async(l1_sum, sizeof(float), s);
  const int vec = ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0) ? 1 : 0;
  hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(vec ? (n + 3) / 4 : n)), dim3(256), 0, s, p, g, m, v, n, hyper, l1_sum, vec);
  SDLT_CHECK_LAUNCH();
  return SDLT_OK;
}
extern "C" int sdlt_lora_shadow_refresh(const sdlt_shadow_desc* descs_dev, const int32_t* block_desc_dev, const int32_t* block_first_dev,
                                        int32_t n_blocks, const float* arena, void* stream) {
  if (n_blocks <= 0) SDLT_FAIL(SDLT_ERR_SHAPE, "sdlt_lora_shadow_refresh: n_blocks=%d", n_blocks);
  hipLaunchKernelGGL(shadow_kernel<false>, dim3(n_blocks), dim3(256), 0, (hipStream_t)stream, descs_dev, block_desc_dev, block_first_dev,
                     const_cast<float*>(arena), (const float*)nullptr, (float*)nullptr, (float*)nullptr, (const float*)nullptr);
  SDLT_CHECK_LAUNCH();
  return SDLT_OK;
}
extern "C" int sdlt_adamw_shadow_refresh(const sdlt_shadow_desc* descs_dev, const int32_t* block_desc_dev, const int32_t* block_first_dev,
                                         int32_t n_blocks, float* p, const float* g, float* m, float* v, const float* hyper, void* stream) {
  if (n_blocks <= 0 || !p || !g || !m || !v || !hyper) SDLT_FAIL(SDLT_ERR_SHAPE, "sdlt_adamw_shadow_refresh: n_blocks=%d or a null buffer", n_blocks);
  hipLaunchKernelGGL(shadow_kernel<true>, dim3(n_blocks), dim3(256), 0, (hipStream_t)stream, descs_dev, block_desc_dev, block_first_dev, p, g, m, v, hyper);
  SDLT_CHECK_LAUNCH();
  return SDLT_OK;
}
extern "C" int sdlt_adamw8_shadow_refresh(const sdlt_shadow_desc* descs_dev, const int32_t* block_desc_dev, const int32_t* block_first_dev, int32_t n_blocks, float* p,
                                          const float* g, uint8_t* m8, uint8_t* v8, float* absmax, const float* tables, const float* hyper, void* stream) {
  if (n_blocks <= 0 || !p || !g || !m8 || !v8 || !absmax || !tables || !hyper) SDLT_FAIL(SDLT_ERR_SHAPE, "sdlt_adamw8_shadow_refresh: n_blocks=%d or a null buffer", n_blocks);
  if ((((uintptr_t)absmax | (uintptr_t)tables) & 15) != 0) SDLT_FAIL(SDLT_ERR_ALIGN, "sdlt_adamw8_shadow_refresh: absmax / tables must be 16-byte aligned");
  hipLaunchKernelGGL(shadow_adamw8_kernel, dim3(n_blocks), dim3(256), 0, (hipStream_t)stream, descs_dev, block_desc_dev, block_first_dev, p, g, m8, v8, absmax, tables, hyper);
  SDLT_CHECK_LAUNCH();
  return SDLT_OK;
}
extern "C" int sdlt_adamw8_flat(float* p, const float* g, uint8_t* m8, uint8_t* v8, float* absmax, int64_t n, const float* tables, const float* hyper, void* stream) {
  if (n <= 0 || (n & 3) || !p || !g || !m8 || !v8 || !absmax || !tables || !hyper) SDLT_FAIL(SDLT_ERR_SHAPE, "sdlt_adamw8_flat: n=%lld (n %% 4 == 0) or a null buffer", (long long)n);
  if ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)tables) & 15) || (((uintptr_t)m8 | (uintptr_t)v8) & 3) || ((uintptr_t)absmax & 7))
    SDLT_FAIL(SDLT_ERR_ALIGN, "sdlt_adamw8_flat: p / g / tables 16-byte, codes 4-byte, absmax 8-byte aligned");
  hipLaunchKernelGGL(adamw8_flat_kernel, dim3((unsigned)((n + 2047) / 2048)), dim3(256), 0, (hipStream_t)stream, p, g, m8, v8, absmax, n, tables, hyper);
  SDLT_CHECK_LAUNCH();
  return SDLT_OK;
}
extern "C" int sdlt_add2d(const void* a, int64_t lda, const void* b, int64_t ldb, void* out, int64_t ldo, int32_t M, int32_t C, void* stream) {
  if (M <= 0 || C <= 0 || (C % 8) || (lda % 8) || (ldb % 8) || (ldo % 8)) SDLT_FAIL(SDLT_ERR_SHAPE, "sdlt_add2d: M=%d C=%d", M, C);
  hipLaunchKernelGGL(add2d_kernel, dim3(grid_for((int64_t)M * C / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)a, lda, (const bf16_t*)b, ldb, (bf16_t*)out, ldo, M, C);
  SDLT_CHECK_LAUNCH();
  return SDLT_OK;
}
extern "C" int sdlt_sum2x2(const void* in, int32_t B, int32_t H, int32_t W, int32_t C, void* out, void* stream) {
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C % 8)) SDLT_FAIL(SDLT_ERR_SHAPE, "sdlt_sum2x2: B=%d H=%d W=%d C=%d", B, H, W, C);
  hipLaunchKernelGGL(sum2x2_kernel, dim3(grid_for((int64_t)B * H * W * C / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)in, B, H, W, C, (bf16_t*)out);
  SDLT_CHECK_LAUNCH();
  return SDLT_OK;
}
extern "C" int sdlt_colsum(const void* x, int64_t ldx, int32_t B, int32_t R, int32_t C, float* ws, int64_t ws_floats, float* out, void* out_bf16, void* stream) {
  if (B <= 0 || R <= 0 || C <= 0 || (C % 64) || (ldx % 8) || (!out && !out_bf16)) SDLT_FAIL(SDLT_ERR_SHAPE, "sdlt_colsum: B=%d R=%d C=%d", B, R, C);
  hipStream_t s = (hipStream_t)stream;
  int rs = (R + 255) / 256;
  if (rs > 32) rs = 32;
  if (!ws || ws_floats < (int64_t)rs * B * C) SDLT_FAIL(SDLT_ERR_SHAPE, "sdlt_colsum: scratch too small (%lld floats needed)", (long long)rs * B * C);
  hipLaunchKernelGGL(colsum_kernel, dim3(C / 64, rs, B), dim3(256), 0, s, (const bf16_t*)x, ldx, R, C, ws);
  hipLaunchKernelGGL(colsum_finish_kernel, dim3((B * C + 255) / 256), dim3(256), 0, s, (const float*)ws, rs, out, (bf16_t*)out_bf16, B * C);
  SDLT_CHECK_LAUNCH();
  return SDLT_OK;
}
