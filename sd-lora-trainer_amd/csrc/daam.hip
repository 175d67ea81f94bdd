// Token-attention (DAAM) loss of the textual-inversion path and its gradient w.r.t. the hooked cross-attention score maps, in four launches.
//
// reference: trainer/ti_cross_attn_loss.py:239-268 (process_and_stack_attention_scores: per-layer [B,N,77] maps -> [B,h,w,77], the larger
// ones bicubic-resized to the smallest, stacked) + trainer/loss.py:10-80 (compute_token_attention_loss), weighted by
// token_attention_loss_w (main.py:342-345).  Every term depends on the stack only through its MEAN over layers A [B,h0,w0,77], and the
// bicubic resize is linear and separable (1-D operators Wh [h0,h], Ww [w0,w] taken from torch's own F.interpolate on identity images), so
// the input is ONE fp32 sum of raw score maps per resolution (S_g, accumulated by the score GEMMs) and the output ONE gradient per
// resolution, shared by its layers.  The loss (sd-lora-trainer_amd/daam.py holds the same arithmetic as ~90 torch launches - the oracle
// of this file's tests and the CPU path):
//   r0 = 5 mean_b sum_t tok_w[b,t] relu(mean_pix A[b,.,.,t])^2 / cnt[b]
//   heat[b,j] = A[b,.,.,pos_j]  (pos_j: position of the j-th trained token in caption b; captions without all of them are skipped)
//   r1 = sum relu(heat M)^2 / den,  r2 = 2 sum relu(heat (1 - M) + 10)^2 / den,  den = max(n_ti,1) n_tok h0 w0,  M = nearest-resized mask
//   r3 = sum_b has[b] var_j(mean_pix heat[b,j]) / max(n_ti,1),   loss = [n_ti > 0] (r0 + r1 + r2 + r3)
// d(weight loss)/dA is a per-(b,t) constant over the pixels (r0) plus dense maps at the <= n_tok trained positions (r1..r3); pulled back
// through the resize that is coef[b,t] ch[h] cw[w] (ch, cw: column sums of Wh, Ww) plus Wh^T dheat Ww at those positions.
//   ta_colsum_kernel : pixel sums of S_g weighted by ch, cw, per 64-row chunk (fixed-order partials)          -> mean_pix A
//   ta_heat_kernel   : one workgroup per (trained token, batch element): its heat map (two small separable products on LDS operands), the
//                      map's pixel mean and r1 / r2 sums
//   ta_grad_kernel   : same grid: d / d heat and its pull-back to every resolution; block (0, b) the coefficient row and loss terms, (0, 0) r0
//   ta_write_kernel  : dS_g = coef ch cw + scatter(dheat_g) as bf16 [B N, 128] and transposed [B 128, N] (the operands of the score-gradient
//                      GEMMs), 16-byte stores both ways; block 0 adds the loss terms up
#include "common.h"
#include "../../include/sdlt_kernels.h"
#define N_OF(G_) ((G_).h * (G_).w)

namespace {

constexpr int TT = 77, TP = 128, MAXB = 16, MAXTOK = 8;

__device__ __forceinline__ float block_sum(float v, float* red) {     // 256 threads; red: 4 floats of LDS
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// partial[b][chunk][t] = sum over the chunk's 64 rows of ch[h] cw[w] S_g[b, row, t]   (chunks numbered over all groups)
__global__ __launch_bounds__(256) void ta_colsum_kernel(const sdlt_ta_params p, int nchunks) {
  const int b = blockIdx.y, c = blockIdx.x;
  int g = 0, c0 = 0;
  for (; g < p.ngroups; ++g) {
    const int n = (p.g[g].h * p.g[g].w + 63) / 64;
    if (c < c0 + n) break;
    c0 += n;
  }
  const sdlt_ta_group& G = p.g[g];
  const int N = G.h * G.w, r0 = (c - c0) * 64;
  const int t = threadIdx.x & 127, half = threadIdx.x >> 7;
  float acc = 0.f;
  for (int r = r0 + half; r < min(r0 + 64, N); r += 2) {
    const int h = r / G.w, w = r - h * G.w;
    const float wt = (G.ch ? G.ch[h] : 1.f) * (G.cw ? G.cw[w] : 1.f);
    acc += wt * G.S[((int64_t)b * N + r) * TP + t];
  }
  __shared__ float sh[128];
  if (half) sh[t] = acc;
  __syncthreads();
  if (!half) p.ws[((int64_t)b * nchunks + c) * TP + t] = acc + sh[t];
}

// small dense products on LDS operands, 256 threads: out[m][n] (+)= sum_k A[m*lda + k*sa] * Bm[k*ldb + n*sb]
__device__ __forceinline__ void lds_mm(float* out, const float* A, int lda, int sa, const float* Bm, int ldb, int sb, int M, int N, int K, bool acc) {
  for (int i = threadIdx.x; i < M * N; i += 256) {
    const int m = i / N, n = i - m * N;
    float s = acc ? out[i] : 0.f;
    const float* a = A + m * lda;
    const float* bp = Bm + n * sb;
#pragma unroll 8
    for (int k = 0; k < K; ++k) s += a[k * sa] * bp[k * ldb];
    out[i] = s;
  }
}

// ws layout behind the colsum partials: coef [B][TP] | terms [B][8] (element 0's slot 7: r0) | pos [B][MAXTOK] | tokmean [B][MAXTOK] + r1/r2 sums [B][MAXTOK][2] | heat [B][ntok][P0]
struct TaWs {
  float *coef, *terms, *pos, *tokmean, *heat;
};
__device__ __forceinline__ TaWs ta_ws(const sdlt_ta_params& p, int nchunks) {
  TaWs w;
  w.coef = p.ws + (int64_t)p.B * nchunks * TP;
  w.terms = w.coef + (int64_t)p.B * TP;
  w.pos = w.terms + (int64_t)p.B * 8;
  w.tokmean = w.pos + (int64_t)p.B * MAXTOK;
  w.heat = w.tokmean + (int64_t)p.B * MAXTOK * 3;
  return w;
}

// grid (n_tok, B): heat[b,j] = A[b,.,.,pos_j] = (1/L) sum_g Wh_g S_g[b,.,.,pos_j] Ww_g^T, its pixel mean and the r1 / r2 sums of (b, j)
__global__ __launch_bounds__(256) void ta_heat_kernel(const sdlt_ta_params p, int nchunks) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int j = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, ntok = p.n_tok;
  const int h0 = p.g[0].h, w0 = p.g[0].w, P0 = h0 * w0;
  float* heat = (float*)smem;             // [P0]
  float* inm = heat + P0;                 // [max_px]
  float* tmp = inm + p.max_px;            // [max_tmp]
  float* wh = tmp + p.max_tmp;            // [h0 * hmax]
  float* ww = wh + p.max_w;               // [w0 * wmax]
  float* red = ww + p.max_w;              // [8]
  const TaWs W = ta_ws(p, nchunks);
  __shared__ int ps_sh;
  if (tid == 0) ps_sh = TP;
  __syncthreads();
  if (tid < TT && p.has_ti[b] > 0.5f && p.ti_onehot[((int64_t)b * ntok + j) * TT + tid] > 0.5f) atomicMin(&ps_sh, tid);
  __syncthreads();
  const int ps = ps_sh < TP ? ps_sh : -1;      // first position of the j-th trained token in caption b
  if (tid == 0) W.pos[b * MAXTOK + j] = (float)ps;
  for (int i = tid; i < P0; i += 256) heat[i] = ps >= 0 ? p.g[0].S[((int64_t)b * P0 + i) * TP + ps] : 0.f;
  for (int g = 1; g < p.ngroups && ps >= 0; ++g) {
    const sdlt_ta_group& G = p.g[g];
    const int N = G.h * G.w;
    __syncthreads();
    for (int i = tid; i < N; i += 256) inm[i] = G.S[((int64_t)b * N + i) * TP + ps];
    for (int i = tid; i < h0 * G.h; i += 256) wh[i] = G.Wh[i];
    for (int i = tid; i < w0 * G.w; i += 256) { const int q = i / G.w, w = i - q * G.w; ww[w * w0 + q] = G.Ww[i]; }   // transposed: lanes run over q
    __syncthreads();
    lds_mm(tmp, wh, G.h, 1, inm, G.w, 1, h0, G.w, G.h, false);            // tmp[pp][w] = sum_h Wh[pp,h] in[h][w]
    __syncthreads();
    lds_mm(heat, tmp, G.w, 1, ww, w0, 1, h0, w0, G.w, true);               // heat[pp][q] += sum_w tmp[pp][w] Ww[q,w]
  }
  __syncthreads();
  const float invL = 1.f / (float)p.n_layers;
  float s1 = 0.f, s2 = 0.f, sm = 0.f;
  for (int i = tid; i < P0; i += 256) {
    const int pp = i / w0, q = i - pp * w0;
    const float M = p.mask[((int64_t)b * 4 * p.mH + (int64_t)pp * p.mH / h0) * p.mW + (int64_t)q * p.mW / w0];
    const float hv = heat[i] * invL;
    W.heat[((int64_t)b * ntok + j) * P0 + i] = hv;
    const float a = fmaxf(hv * M, 0.f), c = fmaxf(hv * (1.f - M) + 10.f, 0.f);
    s1 += a * a;
    s2 += c * c;
    sm += hv;
  }
  s1 = block_sum(s1, red);
  s2 = block_sum(s2, red);
  sm = block_sum(sm, red);
  if (tid == 0) {
    W.tokmean[b * MAXTOK + j] = sm / (float)P0;
    W.tokmean[(int64_t)p.B * MAXTOK + (b * MAXTOK + j) * 2] = s1;       // (the r1 / r2 sums ride behind the means)
    W.tokmean[(int64_t)p.B * MAXTOK + (b * MAXTOK + j) * 2 + 1] = s2;
  }
}

// grid (n_tok, B): d(weight loss)/d heat[b,j] and its pull-back to the larger maps; block (0, b) also the coefficient row and the loss
// terms of element b, block (0, 0) r0
__global__ __launch_bounds__(256) void ta_grad_kernel(const sdlt_ta_params p, int nchunks) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int j = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, B = p.B, ntok = p.n_tok;
  const int h0 = p.g[0].h, w0 = p.g[0].w, P0 = h0 * w0;
  float* dh = (float*)smem;               // [P0]
  float* tmp = dh + P0;                   // [max_tmp]
  float* wh = tmp + p.max_tmp;
  float* ww = wh + p.max_w;
  float* red = ww + p.max_w;
  float* ma = red + 8;                    // [B][TP] (block (0, .) only)
  const TaWs W = ta_ws(p, nchunks);
  const float invL = 1.f / (float)p.n_layers;
  float nti = 0.f;
  for (int bb = 0; bb < B; ++bb) nti += p.has_ti[bb];
  const float gate = nti > 0.f ? 1.f : 0.f, nti1 = fmaxf(nti, 1.f), den = nti1 * ntok * P0;
  const float has = p.has_ti[b], k = p.weight * gate * invL;
  float tm = 0.f, var = 0.f, r1 = 0.f, r2 = 0.f;
  for (int jj = 0; jj < ntok; ++jj) {
    tm += W.tokmean[b * MAXTOK + jj];
    r1 += W.tokmean[(int64_t)B * MAXTOK + (b * MAXTOK + jj) * 2];
    r2 += W.tokmean[(int64_t)B * MAXTOK + (b * MAXTOK + jj) * 2 + 1];
  }
  tm /= (float)ntok;
  if (ntok > 1) {
    for (int jj = 0; jj < ntok; ++jj) { const float d = W.tokmean[b * MAXTOK + jj] - tm; var += d * d; }
    var /= (float)(ntok - 1);
  }
  if (j == 0) {
    // mean over the pixels of A for the rows this block needs (all of them for r0 in block (0, 0)), from the colsum partials in chunk order
    const int nb = b == 0 ? B : 1;
    for (int i = tid; i < nb * TT; i += 256) {
      const int bb = b == 0 ? i / TT : b, t = i - (i / TT) * TT;
      float s = 0.f;
      for (int c = 0; c < nchunks; ++c) s += p.ws[((int64_t)bb * nchunks + c) * TP + t];
      ma[bb * TP + t] = s * invL / (float)P0;
    }
    __syncthreads();
    if (b == 0) {
      float s = 0.f;
      for (int i = tid; i < B * TT; i += 256) {
        const int bb = i / TT, t = i - bb * TT;
        const float r = fmaxf(ma[bb * TP + t], 0.f);
        s += p.tok_w[bb * TT + t] * r * r / p.tok_cnt[bb];
      }
      s = block_sum(s, red);
      if (tid == 0) W.terms[7] = 5.f * s / (float)B;
    }
    if (tid < TP) {
      float c = 0.f;
      if (tid < TT) c = k * 5.f / (float)B * p.tok_w[b * TT + tid] / p.tok_cnt[b] * 2.f * fmaxf(ma[b * TP + tid], 0.f) / (float)P0;
      W.coef[(int64_t)b * TP + tid] = c;
    }
    if (tid == 0) {
      W.terms[b * 8 + 0] = has * r1 / den;
      W.terms[b * 8 + 1] = 2.f * has * r2 / den;
      W.terms[b * 8 + 2] = has * var / nti1;
    }
  }
  const float dv = ntok > 1 ? 2.f * (W.tokmean[b * MAXTOK + j] - tm) / (float)(ntok - 1) / (float)P0 / nti1 : 0.f;
  for (int i = tid; i < P0; i += 256) {
    const int pp = i / w0, q = i - pp * w0;
    const float M = p.mask[((int64_t)b * 4 * p.mH + (int64_t)pp * p.mH / h0) * p.mW + (int64_t)q * p.mW / w0];
    const float hv = W.heat[((int64_t)b * ntok + j) * P0 + i];
    const float a = fmaxf(hv * M, 0.f), c = fmaxf(hv * (1.f - M) + 10.f, 0.f);
    const float d = k * has * (2.f * a * M / den + 4.f * c * (1.f - M) / den + dv);
    dh[i] = d;
    p.g[0].dheat[((int64_t)b * ntok + j) * P0 + i] = d;
  }
  for (int g = 1; g < p.ngroups; ++g) {     // dheat_g = Wh^T dheat Ww
    const sdlt_ta_group& G = p.g[g];
    const int N = G.h * G.w;
    __syncthreads();
    for (int i = tid; i < h0 * G.h; i += 256) wh[i] = G.Wh[i];
    for (int i = tid; i < w0 * G.w; i += 256) ww[i] = G.Ww[i];
    __syncthreads();
    lds_mm(tmp, wh, 1, G.h, dh, w0, 1, G.h, w0, h0, false);               // tmp[h][q] = sum_pp Wh[pp,h] dheat[pp][q]
    __syncthreads();
    for (int i = tid; i < N; i += 256) {                                   // dheat_g[h][w] = sum_q tmp[h][q] Ww[q,w]
      const int h = i / G.w, w = i - h * G.w;
      float s = 0.f;
#pragma unroll 8
      for (int q = 0; q < w0; ++q) s += tmp[h * w0 + q] * ww[q * G.w + w];
      G.dheat[((int64_t)b * ntok + j) * N + i] = s;
    }
  }
}

// dS_g[b, row, t] = coef[b,t] ch[h] cw[w] + sum_j [t == pos_j] dheat_g[b,j,row]  as bf16, row-major and transposed (16-byte stores both ways)
__global__ __launch_bounds__(256) void ta_write_kernel(const sdlt_ta_params p, int nchunks) {
  const int b = blockIdx.y, c = blockIdx.x, tid = threadIdx.x, B = p.B, ntok = p.n_tok;
  int g = 0, c0 = 0;
  for (; g < p.ngroups; ++g) {
    const int n = (p.g[g].h * p.g[g].w + 63) / 64;
    if (c < c0 + n) break;
    c0 += n;
  }
  const sdlt_ta_group& G = p.g[g];
  const int N = G.h * G.w, r0 = (c - c0) * 64;
  const TaWs W = ta_ws(p, nchunks);
  constexpr int LD = TP + 8;
  __shared__ __attribute__((aligned(16))) bf16_t tile[64 * LD];
  __shared__ float coef[TP], rw[64];
  __shared__ int pos[MAXTOK];
  if (tid < TP) coef[tid] = W.coef[(int64_t)b * TP + tid];
  if (tid < MAXTOK) pos[tid] = tid < ntok ? (int)W.pos[b * MAXTOK + tid] : -1;
  if (tid < 64) {
    const int r = r0 + tid, h = r < N ? r / G.w : 0, w = r < N ? r - h * G.w : 0;
    rw[tid] = r < N ? (G.ch ? G.ch[h] : 1.f) * (G.cw ? G.cw[w] : 1.f) : 0.f;
  }
  if (c == 0 && b == 0 && tid == 0) {         // the loss value, terms added in a fixed order
    float nti = 0.f, s = W.terms[7];
    for (int bb = 0; bb < B; ++bb) nti += p.has_ti[bb];
    for (int bb = 0; bb < B; ++bb) s += W.terms[bb * 8 + 0] + W.terms[bb * 8 + 1] + W.terms[bb * 8 + 2];
    p.loss[0] = nti > 0.f ? s : 0.f;
  }
  __syncthreads();
  const int t = tid & 127, half = tid >> 7;
  int jhit = -1;
  for (int j = 0; j < ntok; ++j)
    if (pos[j] == t) jhit = j;               // (distinct trained tokens sit at distinct positions)
  for (int rr = half; rr < 64; rr += 2) {
    float v = t < TT ? coef[t] * rw[rr] : 0.f;
    if (jhit >= 0 && r0 + rr < N) v += G.dheat[((int64_t)b * ntok + jhit) * N + r0 + rr];
    tile[rr * LD + t] = f2bf(v);
  }
  __syncthreads();
  for (int i = tid; i < 64 * 16; i += 256) {              // row-major: 16 chunks of 8 columns per row
    const int rr = i >> 4, ch = i & 15;
    if (r0 + rr < N) *(uint4*)((bf16_t*)G.dS + ((int64_t)b * N + r0 + rr) * TP + ch * 8) = *(const uint4*)(tile + rr * LD + ch * 8);
  }
  for (int i = tid; i < TP * 8; i += 256) {               // transposed: 8 chunks of 8 rows per column t
    const int tt = i >> 3, seg = (i & 7) * 8;
    if (r0 + seg + 8 <= N) {
      uint32_t w4[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) w4[e] = (uint32_t)tile[(seg + 2 * e) * LD + tt] | ((uint32_t)tile[(seg + 2 * e + 1) * LD + tt] << 16);
      *(uint4*)((bf16_t*)G.dSt + ((int64_t)b * TP + tt) * N + r0 + seg) = make_uint4(w4[0], w4[1], w4[2], w4[3]);
    } else {
      for (int e = 0; e < 8; ++e)
        if (r0 + seg + e < N) ((bf16_t*)G.dSt)[((int64_t)b * TP + tt) * N + r0 + seg + e] = tile[(seg + e) * LD + tt];
    }
  }
}

}  // namespace

extern "C" int64_t sdlt_token_attention_ws_floats(const sdlt_ta_params* p) {
  int nch = 0;
  for (int g = 0; g < p->ngroups; ++g) nch += (p->g[g].h * p->g[g].w + 63) / 64;
  return (int64_t)p->B * nch * TP + (int64_t)p->B * TP + (int64_t)p->B * 8 + (int64_t)p->B * MAXTOK * 4 + (int64_t)p->B * p->n_tok * p->g[0].h * p->g[0].w + 64;
}

extern "C" int sdlt_token_attention_loss(const sdlt_ta_params* pp, void* stream) {
  sdlt_ta_params p = *pp;
  if (p.ngroups < 1 || p.ngroups > 4 || p.B < 1 || p.B > MAXB || p.n_tok < 1 || p.n_tok > MAXTOK || p.n_layers < 1)
    SDLT_FAIL(SDLT_ERR_SHAPE, "sdlt_token_attention_loss: ngroups=%d B=%d n_tok=%d n_layers=%d", p.ngroups, p.B, p.n_tok, p.n_layers);
  int nch = 0, maxpx = 0, maxtmp = 0, maxw = 0;
  const int h0 = p.g[0].h, w0 = p.g[0].w;
  for (int g = 0; g < p.ngroups; ++g) {
    const sdlt_ta_group& G = p.g[g];
    if (G.h < 1 || G.w < 1 || !G.S || !G.dS || !G.dSt || !G.dheat || (g > 0 && (!G.Wh || !G.Ww || !G.ch || !G.cw || G.h * G.w < h0 * w0)))
      SDLT_FAIL(SDLT_ERR_SHAPE, "sdlt_token_attention_loss: group %d (%d x %d)", g, G.h, G.w);
    nch += (G.h * G.w + 63) / 64;
    if (g > 0) {
      maxpx = G.h * G.w > maxpx ? G.h * G.w : maxpx;
      const int t1 = h0 * G.w, t2 = G.h * w0;
      maxtmp = (t1 > t2 ? t1 : t2) > maxtmp ? (t1 > t2 ? t1 : t2) : maxtmp;
      const int m1 = h0 * G.h, m2 = w0 * G.w;
      maxw = (m1 > m2 ? m1 : m2) > maxw ? (m1 > m2 ? m1 : m2) : maxw;
      if ((N_OF(G) % 8) != 0) SDLT_FAIL(SDLT_ERR_SHAPE, "sdlt_token_attention_loss: group %d has %d pixels (multiple of 8 needed)", g, G.h * G.w);
    }
  }
  p.max_px = maxpx ? maxpx : 4;
  p.max_tmp = maxtmp ? maxtmp : 4;
  p.max_w = maxw ? maxw : 4;
  if ((h0 * w0) % 8) SDLT_FAIL(SDLT_ERR_SHAPE, "sdlt_token_attention_loss: %d x %d pixels (multiple of 8 needed)", h0, w0);
  const size_t smem = sizeof(float) * ((size_t)h0 * w0 + p.max_px + p.max_tmp + 2 * (size_t)p.max_w + 8 + (size_t)MAXB * TP);
  if (smem > 150 * 1024) SDLT_FAIL(SDLT_ERR_UNSUPPORTED, "sdlt_token_attention_loss: maps of %d x %d / %d px do not fit the LDS", h0, w0, maxpx);
  if (!p.ws || p.ws_floats < sdlt_token_attention_ws_floats(&p) || !p.loss || !p.mask || !p.tok_w || !p.tok_cnt || !p.ti_onehot || !p.has_ti)
    SDLT_FAIL(SDLT_ERR_SHAPE, "sdlt_token_attention_loss: workspace / operands");
  hipStream_t s = (hipStream_t)stream;
  if (sdlt_raise_smem((const void*)ta_heat_kernel, 150 * 1024) || sdlt_raise_smem((const void*)ta_grad_kernel, 150 * 1024))
    SDLT_FAIL(SDLT_ERR_LAUNCH, "sdlt_token_attention_loss: cannot raise the dynamic LDS limit");
  hipLaunchKernelGGL(ta_colsum_kernel, dim3(nch, p.B), dim3(256), 0, s, p, nch);
  hipLaunchKernelGGL(ta_heat_kernel, dim3(p.n_tok, p.B), dim3(256), smem, s, p, nch);
  hipLaunchKernelGGL(ta_grad_kernel, dim3(p.n_tok, p.B), dim3(256), smem, s, p, nch);
  hipLaunchKernelGGL(ta_write_kernel, dim3(nch, p.B), dim3(256), 0, s, p, nch);
  SDLT_CHECK_LAUNCH();
  return SDLT_OK;
}
