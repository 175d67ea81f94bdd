// Flash-style multi-head attention forward / backward on MFMA (gfx950), used for both the UNet
// self-attention (attn1) and the 77-token cross-attention (attn2) of every BasicTransformerBlock,
// and for the CLIP text encoders (causal flag).
//
// Replaces F.scaled_dot_product_attention inside diffusers' AttnProcessor2_0 and the reference's own
// DAAMLossAttnProcessor2_0.__call__ (trainer/ti_cross_attn_loss.py:197-199) together with its autograd
// backward.  (The head-summed raw-score side output of ti_cross_attn_loss.py:201-212 is one dense GEMM,
// sum_h Q_h K_h^T = Q K^T, and goes through sdlt_gemm_bf16.)
//
// Layout contract: Q/K/V/O/dO/dQ/dK/dV are token-major [B*N, C] with head h in columns [h*d,(h+1)*d).
// Every LDS tile is the natural [64 tokens][d] image filled with plain 16-byte copies.  Operands that an MFMA must
// contract over the TOKEN axis (V in P.V, K in dS.K, Q and dO in the dK/dV products) are read from that same image
// with the gfx950 transposing LDS read (ds_read_b64_tr_b16 x2 = one 16x16x32 fragment), so no transposed copy of any
// tensor exists in HBM and each tile is fetched once.  (The Kt/Vt/Qt/dOt fields of sdlt_attn_params are ignored.)
//
// MFMA chaining without cross-lane traffic: all products are computed "swapped" (D[i][j] with j the
// row that owns the softmax statistics), and the tile rows fed as the MFMA A operand are PERMUTED
// (row i of a 16-row fragment <-> token (i/4)*8 + half*4 + i%4 of a 32-token block) so that the two
// 16x16 accumulators of a 32-token block are, per lane, exactly the 8 consecutive tokens the next
// 16x16x32 MFMA wants as its B operand.
#include <cstdlib>
#include "common.h"
#include "../../include/sdlt_kernels.h"
#include "attn32.h"

namespace {

// -DSDLT_ATTN_TRACE (tools/attn_trace.py): thread 0 of workgroup 0 stamps clock64() into p.D at the phase boundaries of a kernel
#ifdef SDLT_ATTN_TRACE
#define TR() do { if (tr) trb[trn++] = clock64(); } while (0)
#else
#define TR() do {} while (0)
#endif

constexpr float LOG2E = 1.4426950408889634f;

__device__ __forceinline__ bf16x8 pack8(const float* a, const float* b) {
  union { uint4 u; bf16x8 v; } c;
  c.u.x = pack2bf(a[0], a[1]); c.u.y = pack2bf(a[2], a[3]);
  c.u.z = pack2bf(b[0], b[1]); c.u.w = pack2bf(b[2], b[3]);
  return c.v;
}
__device__ __forceinline__ bf16x8 ld_frag_global(const bf16_t* p, bool ok) {
  union { uint4 u; bf16x8 v; } c;
  c.u = ok ? *(const uint4*)p : make_uint4(0, 0, 0, 0);
  return c.v;
}
__device__ __forceinline__ int prow(int i, int half) { return ((i >> 2) << 3) + half * 4 + (i & 3); }

// MFMA fragment contracted over the ROW axis of a row-major bf16 LDS tile: lane (g = lane>>4, i = lane&15) receives rows
// r0 + 8g + {0..7} of column c0 + i.  `p` = tile + (r0 + 8g + (i>>2)) * stride + (c0 + 4*(i&3)) * 2: each 16-lane group
// hands the hardware a [4 rows][16 cols] block (lane i: row i>>2, columns 4*(i&3)..+3) and gets it back transposed.
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x8 lds_tr_frag(const char* p, int stride) {
  typedef __attribute__((address_space(3))) bf16x4* lds_ptr;
  bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_ptr)p);
  bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_ptr)(p + 4 * stride));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// Workgroups are dealt to the 8 XCDs round-robin by linear id and every XCD has its own L2.  Renumber them so that XCD k runs a
// CONTIGUOUS range of the (tile fastest, then head, then batch) order: the tiles of one head - which all stream the same K / V (forward,
// dQ) or Q / dO (dK / dV) rows - then sit on one or two XCDs instead of all eight, and those rows are fetched into one L2, not eight.
struct WgId { int x, y, z; };
__device__ int g_attn_xcd = 1;          // SDLT_ATTN_XCD=0 (A/B): plain blockIdx order
__device__ __forceinline__ WgId xcd_wg() {
  if (!g_attn_xcd) { WgId w; w.x = blockIdx.x; w.y = blockIdx.y; w.z = blockIdx.z; return w; }
  const int gx = gridDim.x, gy = gridDim.y, n = gx * gy * gridDim.z;
  const int L = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
  const int k = L & 7, slot = L >> 3, q = n >> 3, r = n & 7;
  const int logical = (k < r ? k * (q + 1) : r * (q + 1) + (k - r) * q) + slot;
  WgId w;
  const int t = div_small_u(logical, gx);
  w.x = logical - t * gx;
  w.z = gridDim.z == 1 ? 0 : div_small_u(t, gy);
  w.y = t - w.z * gy;
  return w;
}

// LDS tile geometry.  d = 64 (every SDXL head; DP = 64): 128-byte rows with the 16-byte chunk index XOR-ed by row bits 1 and 3,
//   chunk' = chunk ^ (((row >> 1) & 1) * 2 + ((row >> 3) & 1) * 4).
// With the +16-byte padded rows used before (and still for the other head widths) the two 16-lane halves of a transposing read
// (rows r..r+3 and r+8..r+11, 32 B each) overlap in 8 of their 16 bank groups for EVERY pitch that keeps rows 16-byte aligned -
// 38-41 % of the attention kernels' LDS cycles were bank-conflict cycles (PMC, round 1).  Under the XOR the 8 rows of such a half
// land in 8 different 32-byte bank groups, the 16 lanes of a ds_read_b128 lane group (rows {0-3, 24-27} chunk c / rows {8-11,
// 16-19} chunk c+1 of the permuted fragments, or 16 consecutive rows) in 16 different 16-byte groups, and a row's eight 16-byte
// stores stay a permutation of one 128-byte line.
template <int DP>
__device__ __host__ constexpr int tile_stride() { return DP == 64 ? 128 : DP * 2 + 16; }
template <int DP>
__device__ __forceinline__ int row_sw(int row) { return DP == 64 ? ((((row >> 1) & 1) << 1) | (((row >> 3) & 1) << 2)) : 0; }

// Tiles are staged global -> registers -> LDS in two halves so the HBM/L2 latency of tile t+1 hides under the MFMAs
// of tile t (registers are loaded before the compute phase and written to the other LDS buffer after it).
// natural tile: [64 rows][DP] (row stride NSTR bytes) <- src[(row0+row)*ld + col0 + c]; rows >= nrows or c >= d read as 0
template <int DP>
struct TileRegs { uint4 r[DP / 32]; };

template <int DP>
__device__ __forceinline__ void gload_nat(TileRegs<DP>& t, const bf16_t* src, int64_t ld, int64_t row0, int nrows, int col0, int d, int tid = threadIdx.x) {
  constexpr int CH = DP / 8;
#pragma unroll
  for (int i = 0; i < DP / 32; ++i) {
    int c = tid + 256 * i;
    int row = c / CH, ch = c - row * CH;
    t.r[i] = (row < nrows && ch * 8 < d) ? *(const uint4*)(src + (row0 + row) * ld + col0 + ch * 8) : make_uint4(0, 0, 0, 0);
  }
}
template <int DP>
__device__ __forceinline__ void sstore_nat(const TileRegs<DP>& t, char* dst, int tid = threadIdx.x) {
  constexpr int NSTR = tile_stride<DP>(), CH = DP / 8;
#pragma unroll
  for (int i = 0; i < DP / 32; ++i) {
    int c = tid + 256 * i;
    int row = c / CH, ch = c - row * CH;
    *(uint4*)(dst + row * NSTR + ((ch ^ row_sw<DP>(row)) << 4)) = t.r[i];
  }
}
// =============================================================================== forward
// KS = 2: the 64-key tiles of a query tile are dealt out to TWO groups of 4 waves (even / odd tiles), each with its own online softmax
// state and K / V buffers; the groups' (m, l, O) meet through LDS at the end.  A 1024-token layer is 16 dependent tile steps per
// workgroup and 1.25 workgroups per CU: with one wave per SIMD the softmax VALU work and the MFMAs of a step run one after the other -
// two waves per SIMD on half the chain each overlap them (and halve the chain).
template <int DP, int KS, int DV = DP>
__device__ __forceinline__ void attn_fwd_body(const sdlt_attn_params& p, char* smem, const WgId wg) {
  constexpr int NSTR = tile_stride<DP>();
  const int b = wg.z, h = wg.y, q0 = wg.x * 64;
  const int lane = threadIdx.x & 63, wave = (threadIdx.x >> 6) & 3, g = lane >> 4, i = lane & 15;
  const int grp = KS == 1 ? 0 : __builtin_amdgcn_readfirstlane(threadIdx.x >> 8), tid = threadIdx.x & 255;
  const int d = p.d, hc = h * d;
  const int q = q0 + wave * 16 + i;
  bf16x8 qf[DP / 32];
#pragma unroll
  for (int kk = 0; kk < DP / 32; ++kk) {
    int col = kk * 32 + g * 8;
    qf[kk] = ld_frag_global((const bf16_t*)p.Q + ((int64_t)b * p.Nqp + q) * p.ldq + hc + col, q < p.Nq && col < d);
  }
  float m = -1e30f, lsum = 0.f;
  f32x4 o[DV / 16];
#pragma unroll
  for (int df = 0; df < DV / 16; ++df) o[df] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const float sl2 = p.scale * LOG2E;
  const int kend = p.causal ? min(p.Nk, q0 + 64) : p.Nk;

  constexpr int FBUF = 2 * 64 * NSTR;   // one K tile + one V tile
  char* gsm = smem + grp * (2 * FBUF);  // this group's two buffers
  const int ntile = (kend + 63) >> 6, niter = (ntile + KS - 1) / KS;
  TileRegs<DP> kr, vr;
  if (grp < ntile) {
    const int kf0 = grp * 64;
    gload_nat<DP>(kr, (const bf16_t*)p.K, p.ldk, (int64_t)b * p.Nkp + kf0, min(64, p.Nkp - kf0), hc, d, tid);
    gload_nat<DP>(vr, (const bf16_t*)p.V, p.ldv, (int64_t)b * p.Nkp + kf0, min(64, p.Nkp - kf0), hc, d, tid);
    sstore_nat<DP>(kr, gsm, tid);
    sstore_nat<DP>(vr, gsm + 64 * NSTR, tid);
  }
  const int troff = (8 * g + (i >> 2)) * NSTR + (i & 3) * 8;   // transposing-read lane offset inside a natural tile
  const int tsw = row_sw<DP>(8 * g + (i >> 2)) >> 1;   // this lane's XOR on the 32-byte column block of a transposing read
  __syncthreads();
#ifdef SDLT_ATTN_TRACE
  const bool tr = blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0 && p.D;
  long long* trb = (long long*)p.D;
  int trn = 0;
  TR();
#endif
  for (int it = 0; it < niter; ++it) {
    TR();
    const int k0 = (it * KS + grp) * 64;
    if (KS > 1 && k0 >= kend) { __syncthreads(); continue; }     // (the shorter group idles through the last turn; wave-uniform)
    const char* Ks = gsm + (it & 1) * FBUF;
    const char* Vs = Ks + 64 * NSTR;
    const int kn = k0 + 64 * KS;          // this group's next tile
    const bool more = kn < kend;
    if (more) {   // next tile's loads fly while this tile is computed
      const int nk = min(64, p.Nkp - kn);
      gload_nat<DP>(kr, (const bf16_t*)p.K, p.ldk, (int64_t)b * p.Nkp + kn, nk, hc, d, tid);
      gload_nat<DP>(vr, (const bf16_t*)p.V, p.ldv, (int64_t)b * p.Nkp + kn, nk, hc, d, tid);
    }
    f32x4 s[4];
#pragma unroll
    for (int kf = 0; kf < 4; ++kf) {
      s[kf] = (f32x4){0.f, 0.f, 0.f, 0.f};
      const int krow = (kf >> 1) * 32 + prow(i, kf & 1);
#pragma unroll
      for (int kk = 0; kk < DP / 32; ++kk) {
        bf16x8 kfr = *(const bf16x8*)(Ks + krow * NSTR + (((kk * 4 + g) ^ row_sw<DP>(krow)) << 4));
        s[kf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfr, qf[kk], s[kf], 0, 0, 0);
      }
    }
    // scores stay raw; the softmax scale rides in the exponent's FMA.  Masking only runs on tiles that need it (wave-uniform).
    const bool full = k0 + 64 <= p.Nk && !(p.causal && k0 + 63 > q0 + wave * 16);
    if (!full) {
#pragma unroll
      for (int kf = 0; kf < 4; ++kf)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          int key = k0 + (kf >> 1) * 32 + g * 8 + (kf & 1) * 4 + r;
          if (key >= p.Nk || (p.causal && key > q)) s[kf][r] = -1e30f;
        }
    }
    float tmax = -1e30f;
#pragma unroll
    for (int kf = 0; kf < 4; ++kf)
#pragma unroll
      for (int r = 0; r < 4; ++r) tmax = fmaxf(tmax, s[kf][r]);
    tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float mn = fmaxf(m, tmax * sl2);
    const float alpha = __builtin_amdgcn_exp2f(m - mn);
    m = mn;
    float rs = 0.f;
#pragma unroll
    for (int kf = 0; kf < 4; ++kf)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kf][r], sl2, -mn));
        s[kf][r] = pv;
        rs += pv;
      }
    lsum = lsum * alpha + rs;
    TR();
#pragma unroll
    for (int df = 0; df < DV / 16; ++df) o[df] *= alpha;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      float a4[4] = {s[2 * kb][0], s[2 * kb][1], s[2 * kb][2], s[2 * kb][3]};
      float b4[4] = {s[2 * kb + 1][0], s[2 * kb + 1][1], s[2 * kb + 1][2], s[2 * kb + 1][3]};
      bf16x8 pf = pack8(a4, b4);
#pragma unroll
      for (int df = 0; df < DV / 16; ++df) {
        bf16x8 vf = lds_tr_frag(Vs + troff + kb * 32 * NSTR + ((df ^ tsw) << 5), NSTR);   // V[key block kb][columns df*16..]^T
        o[df] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, o[df], 0, 0, 0);
      }
    }
    TR();
    if (more) {
      char* nb = gsm + ((it + 1) & 1) * FBUF;
      sstore_nat<DP>(kr, nb, tid);
      sstore_nat<DP>(vr, nb + 64 * NSTR, tid);
    }
    TR();
    __syncthreads();
  }
  TR();
  lsum += __shfl_xor(lsum, 16, 64);
  lsum += __shfl_xor(lsum, 32, 64);
  if constexpr (KS > 1) {
    // merge the groups' softmax states: (m, l, O) of group 1 -> LDS -> group 0 (every tile buffer is dead after the last barrier)
    float* mg = (float*)smem + (wave * 64 + lane) * (4 + DP / 4);       // [m, l, -, -, O...]: 16-byte aligned rows
    if (grp == 1) {
      mg[0] = m;
      mg[1] = lsum;
#pragma unroll
      for (int df = 0; df < DV / 16; ++df) *(f32x4*)(mg + 4 + df * 4) = o[df];
    }
    __syncthreads();
    if (grp == 1) return;
    const float m1 = mg[0], l1 = mg[1];
    const float mn = fmaxf(m, m1), a0 = __builtin_amdgcn_exp2f(m - mn), a1 = __builtin_amdgcn_exp2f(m1 - mn);
    m = mn;
    lsum = lsum * a0 + l1 * a1;
#pragma unroll
    for (int df = 0; df < DV / 16; ++df) {
      const f32x4 o1 = *(const f32x4*)(mg + 4 + df * 4);
      o[df] = o[df] * a0 + o1 * a1;
    }
  }
  if (q < p.Nqp) {   // pad rows get finite values too, so no consumer ever reads uninitialised memory
    const float inv = 1.f / lsum;
    if (g == 0 && p.L && q < p.Nq) p.L[((int64_t)b * p.H + h) * p.Nq + q] = (m + log2f(lsum)) / LOG2E;
#pragma unroll
    for (int df = 0; df < DV / 16; ++df) {
      int col = df * 16 + g * 4;
      if (col < d) {
        uint2 w;
        w.x = pack2bf(o[df][0] * inv, o[df][1] * inv);
        w.y = pack2bf(o[df][2] * inv, o[df][3] * inv);
        *(uint2*)((bf16_t*)p.O + ((int64_t)b * p.Nqp + q) * p.ldo + hc + col) = w;
      }
    }
  }
}

// DV: width of the OUTPUT column blocks (O, dQ, dK, dV accumulators and the products that feed them).  Heads of 40 columns (SD1.5's 320-wide blocks: 4096 tokens x 8 heads)
// sit in the 64-column tiles of DP = 64 - the contraction over d keeps its two 32-wide steps - but need only three of the four 16-column output blocks: DV = 48 drops a quarter
// of the P V / dS K / P^T dO / dS^T Q MFMAs and of the accumulator registers.
template <int DP, int KS = 1, int DV = DP>
__global__ __launch_bounds__(256 * KS) __attribute__((amdgpu_waves_per_eu(DP == 64 ? 4 : 1, DP == 64 ? 4 : 8))) void attn_fwd_kernel(const sdlt_attn_params p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  attn_fwd_body<DP, KS, DV>(p, smem, xcd_wg());
}
// Two independent attention problems of the same head width in ONE launch: the heads of p1 follow the heads of p0 along grid y (the text
// encoders' layer i of CLIP-L and of OpenCLIP-bigG; see strip_pair_kernel).
template <int DP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(DP == 64 ? 4 : 1, DP == 64 ? 4 : 8))) void attn_fwd_pair_kernel(const sdlt_attn_params p0, const sdlt_attn_params p1) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  WgId wg = xcd_wg();
  if (wg.y < p0.H) {
    if (wg.x * 64 < p0.Nq && wg.z < p0.B) attn_fwd_body<DP, 1>(p0, smem, wg);
  } else {
    wg.y -= p0.H;
    if (wg.x * 64 < p1.Nq && wg.z < p1.B) attn_fwd_body<DP, 1>(p1, smem, wg);
  }
}

// =============================================================================== backward dQ (per 64-query tile)
template <int DP, bool WRITE_D, int DV = DP>
__device__ __forceinline__ void attn_bwd_dq_body(const sdlt_attn_params& p, char* smem, const int bx, const WgId wg) {
  constexpr int NSTR = tile_stride<DP>();
  const int b = wg.z, h = wg.y, q0 = bx * 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, i = lane & 15;
  const int d = p.d, hc = h * d;
  const int q = q0 + wave * 16 + i;
  const bool qok = q < p.Nq;
  bf16x8 qf[DP / 32], gf[DP / 32];
#pragma unroll
  for (int kk = 0; kk < DP / 32; ++kk) {
    int col = kk * 32 + g * 8;
    qf[kk] = ld_frag_global((const bf16_t*)p.Q + ((int64_t)b * p.Nqp + q) * p.ldq + hc + col, qok && col < d);
    gf[kk] = ld_frag_global((const bf16_t*)p.dO + ((int64_t)b * p.Nqp + q) * p.lddo + hc + col, qok && col < d);
  }
  const float Lq = qok ? p.L[((int64_t)b * p.H + h) * p.Nq + q] * LOG2E : 0.f;
  // D = rowsum(dO * O) of this wave's 16 query rows, from the dO fragments it holds anyway (no separate prep launch); stored
  // for the dK/dV pass that follows on the same stream
  float Dq = 0.f;
#pragma unroll
  for (int kk = 0; kk < DP / 32; ++kk) {
    int col = kk * 32 + g * 8;
    bf16x8 of = ld_frag_global((const bf16_t*)p.O + ((int64_t)b * p.Nqp + q) * p.ldo + hc + col, qok && col < d);
#pragma unroll
    for (int j = 0; j < 8; ++j) Dq += (float)gf[kk][j] * (float)of[j];
  }
  Dq += __shfl_xor(Dq, 16, 64);
  Dq += __shfl_xor(Dq, 32, 64);
  if (WRITE_D && g == 0 && qok) p.D[((int64_t)b * p.H + h) * p.Nq + q] = Dq;   // (merged launch: written by attn_prep_kernel instead)
  f32x4 dq[DV / 16];
#pragma unroll
  for (int df = 0; df < DV / 16; ++df) dq[df] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const float sl2 = p.scale * LOG2E;
  const int kend = p.causal ? min(p.Nk, q0 + 64) : p.Nk;

  constexpr int QBUF = 2 * 64 * NSTR;   // K, V natural
  TileRegs<DP> kr, vr;
  const int troff = (8 * g + (i >> 2)) * NSTR + (i & 3) * 8;
  const int tsw = row_sw<DP>(8 * g + (i >> 2)) >> 1;   // this lane's XOR on the 32-byte column block of a transposing read
  {
    const int nk = min(64, p.Nkp);
    gload_nat<DP>(kr, (const bf16_t*)p.K, p.ldk, (int64_t)b * p.Nkp, nk, hc, d);
    gload_nat<DP>(vr, (const bf16_t*)p.V, p.ldv, (int64_t)b * p.Nkp, nk, hc, d);
    sstore_nat<DP>(kr, smem);
    sstore_nat<DP>(vr, smem + 64 * NSTR);
  }
  __syncthreads();
  int it = 0;
  for (int k0 = 0; k0 < kend; k0 += 64, ++it) {
    const char* Ks = smem + (it & 1) * QBUF;
    const char* Vs = Ks + 64 * NSTR;
    const bool more = k0 + 64 < kend;
    if (more) {
      const int nk = min(64, p.Nkp - (k0 + 64));
      gload_nat<DP>(kr, (const bf16_t*)p.K, p.ldk, (int64_t)b * p.Nkp + k0 + 64, nk, hc, d);
      gload_nat<DP>(vr, (const bf16_t*)p.V, p.ldv, (int64_t)b * p.Nkp + k0 + 64, nk, hc, d);
    }
    f32x4 s[4], dp[4];
#pragma unroll
    for (int kf = 0; kf < 4; ++kf) {
      s[kf] = (f32x4){0.f, 0.f, 0.f, 0.f};
      dp[kf] = (f32x4){0.f, 0.f, 0.f, 0.f};
      const int krow = (kf >> 1) * 32 + prow(i, kf & 1);
#pragma unroll
      for (int kk = 0; kk < DP / 32; ++kk) {
        bf16x8 kfr = *(const bf16x8*)(Ks + krow * NSTR + (((kk * 4 + g) ^ row_sw<DP>(krow)) << 4));
        s[kf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfr, qf[kk], s[kf], 0, 0, 0);
        bf16x8 vfr = *(const bf16x8*)(Vs + krow * NSTR + (((kk * 4 + g) ^ row_sw<DP>(krow)) << 4));
        dp[kf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vfr, gf[kk], dp[kf], 0, 0, 0);
      }
    }
    // wave-uniform: every (query row of this wave, key of this tile) pair is live -> no per-element masking
    const bool full = k0 + 64 <= p.Nk && q0 + wave * 16 + 16 <= p.Nq && !(p.causal && k0 + 63 > q0 + wave * 16);
    if (full) {
#pragma unroll
      for (int kf = 0; kf < 4; ++kf)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kf][r], sl2, -Lq));
          s[kf][r] = pv * (dp[kf][r] - Dq);            // dS / scale (the softmax scale multiplies the finished dQ rows once instead of every score)
        }
    } else {
#pragma unroll
      for (int kf = 0; kf < 4; ++kf)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          int key = k0 + (kf >> 1) * 32 + g * 8 + (kf & 1) * 4 + r;
          bool ok = qok && key < p.Nk && !(p.causal && key > q);
          float pv = ok ? __builtin_amdgcn_exp2f(__builtin_fmaf(s[kf][r], sl2, -Lq)) : 0.f;
          s[kf][r] = pv * (dp[kf][r] - Dq);            // dS / scale (the softmax scale multiplies the finished dQ rows once instead of every score)
        }
    }
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      float a4[4] = {s[2 * kb][0], s[2 * kb][1], s[2 * kb][2], s[2 * kb][3]};
      float b4[4] = {s[2 * kb + 1][0], s[2 * kb + 1][1], s[2 * kb + 1][2], s[2 * kb + 1][3]};
      bf16x8 dsf = pack8(a4, b4);
#pragma unroll
      for (int df = 0; df < DV / 16; ++df) {
        bf16x8 ktf = lds_tr_frag(Ks + troff + kb * 32 * NSTR + ((df ^ tsw) << 5), NSTR);
        dq[df] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ktf, dsf, dq[df], 0, 0, 0);
      }
    }
    if (more) {
      char* nb = smem + ((it + 1) & 1) * QBUF;
      sstore_nat<DP>(kr, nb);
      sstore_nat<DP>(vr, nb + 64 * NSTR);
    }
    __syncthreads();
  }
  if (q < p.Nqp) {   // pad rows: dq == 0
#pragma unroll
    for (int df = 0; df < DV / 16; ++df) {
      int col = df * 16 + g * 4;
      if (col < d) {
        uint2 w;
        w.x = pack2bf(dq[df][0] * p.scale, dq[df][1] * p.scale);
        w.y = pack2bf(dq[df][2] * p.scale, dq[df][3] * p.scale);
        *(uint2*)((bf16_t*)p.dQ + ((int64_t)b * p.Nqp + q) * p.lddq + hc + col) = w;
      }
    }
  }
}

// =============================================================================== backward dK,dV (per 64-key tile, optional query split)
template <int DP, int DV = DP>
__device__ __forceinline__ void attn_bwd_dkdv_body(const sdlt_attn_params& p, char* smem, const int bx, const WgId wg) {
  constexpr int NSTR = tile_stride<DP>();
  const int b = wg.z, h = wg.y;
  const int ktile = bx / p.qsplit, split = bx - ktile * p.qsplit;
  const int k0 = ktile * 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, i = lane & 15;
  const int d = p.d, hc = h * d;
  const int key = k0 + wave * 16 + i;
  const bool kok = key < p.Nk;
  bf16x8 kf[DP / 32], vf[DP / 32];
#pragma unroll
  for (int kk = 0; kk < DP / 32; ++kk) {
    int col = kk * 32 + g * 8;
    kf[kk] = ld_frag_global((const bf16_t*)p.K + ((int64_t)b * p.Nkp + key) * p.ldk + hc + col, kok && col < d);
    vf[kk] = ld_frag_global((const bf16_t*)p.V + ((int64_t)b * p.Nkp + key) * p.ldv + hc + col, kok && col < d);
  }
  f32x4 dk[DV / 16], dv[DV / 16];
#pragma unroll
  for (int df = 0; df < DV / 16; ++df) {
    dk[df] = (f32x4){0.f, 0.f, 0.f, 0.f};
    dv[df] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  const float sl2 = p.scale * LOG2E;
  const int nqt = (p.Nq + 63) / 64;
  const int per = (nqt + p.qsplit - 1) / p.qsplit;
  int qt_lo = split * per, qt_hi = min(nqt, qt_lo + per);
  if (p.causal) qt_lo = max(qt_lo, ktile);  // queries before this key tile never see it

  constexpr int KBUF = 2 * 64 * NSTR + 512;   // Q, dO natural + L, D
  constexpr bool DB = 2 * KBUF <= 160 * 1024;                 // double-buffered unless it would not fit the LDS
  TileRegs<DP> qr, gr;
  const int troff = (8 * g + (i >> 2)) * NSTR + (i & 3) * 8;
  const int tsw = row_sw<DP>(8 * g + (i >> 2)) >> 1;   // this lane's XOR on the 32-byte column block of a transposing read
  float lreg = 0.f, dreg = 0.f;
  auto gload_all = [&](int qt) {
    const int q0 = qt * 64;
    const int nq = min(64, p.Nqp - q0);
    gload_nat<DP>(qr, (const bf16_t*)p.Q, p.ldq, (int64_t)b * p.Nqp + q0, nq, hc, d);
    gload_nat<DP>(gr, (const bf16_t*)p.dO, p.lddo, (int64_t)b * p.Nqp + q0, nq, hc, d);
    if (threadIdx.x < 64) {
      int qq = q0 + threadIdx.x;
      lreg = qq < p.Nq ? p.L[((int64_t)b * p.H + h) * p.Nq + qq] * LOG2E : 0.f;
      dreg = qq < p.Nq ? p.D[((int64_t)b * p.H + h) * p.Nq + qq] : 0.f;
    }
  };
  auto sstore_all = [&](char* base) {
    sstore_nat<DP>(qr, base);
    sstore_nat<DP>(gr, base + 64 * NSTR);
    if (threadIdx.x < 64) {
      float* ls = (float*)(base + 2 * 64 * NSTR);
      ls[threadIdx.x] = lreg;
      ls[64 + threadIdx.x] = dreg;
    }
  };
  if (qt_lo < qt_hi) {
    gload_all(qt_lo);
    sstore_all(smem);
  }
  __syncthreads();
  for (int qt = qt_lo; qt < qt_hi; ++qt) {
    const int q0 = qt * 64;
    const char* Qs = smem + (DB ? ((qt - qt_lo) & 1) * KBUF : 0);
    const char* Gs = Qs + 64 * NSTR;
    const float* Ls = (const float*)(Gs + 64 * NSTR);
    const float* Ds = Ls + 64;
    const bool more = qt + 1 < qt_hi;
    if (more) gload_all(qt + 1);
    f32x4 s[4], dp[4];
#pragma unroll
    for (int qf = 0; qf < 4; ++qf) {
      s[qf] = (f32x4){0.f, 0.f, 0.f, 0.f};
      dp[qf] = (f32x4){0.f, 0.f, 0.f, 0.f};
      const int qrow = (qf >> 1) * 32 + prow(i, qf & 1);
#pragma unroll
      for (int kk = 0; kk < DP / 32; ++kk) {
        bf16x8 qfr = *(const bf16x8*)(Qs + qrow * NSTR + (((kk * 4 + g) ^ row_sw<DP>(qrow)) << 4));
        s[qf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qfr, kf[kk], s[qf], 0, 0, 0);
        bf16x8 gfr = *(const bf16x8*)(Gs + qrow * NSTR + (((kk * 4 + g) ^ row_sw<DP>(qrow)) << 4));
        dp[qf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gfr, vf[kk], dp[qf], 0, 0, 0);
      }
    }
    // s[qf][r] = S[q = q0 + (qf>>1)*32 + g*8 + (qf&1)*4 + r][key]
    const bool full = q0 + 64 <= p.Nq && k0 + wave * 16 + 16 <= p.Nk && !(p.causal && k0 + wave * 16 + 15 > q0);
    if (full) {
#pragma unroll
      for (int qf = 0; qf < 4; ++qf) {
        const f32x4 l4 = *(const f32x4*)(Ls + (qf >> 1) * 32 + g * 8 + (qf & 1) * 4);
        const f32x4 d4 = *(const f32x4*)(Ds + (qf >> 1) * 32 + g * 8 + (qf & 1) * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(s[qf][r], sl2, -l4[r]));
          s[qf][r] = pv;
          dp[qf][r] = pv * (dp[qf][r] - d4[r]);          // dS / scale (applied to the finished dK rows)
        }
      }
    } else {
#pragma unroll
      for (int qf = 0; qf < 4; ++qf)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          int ql = (qf >> 1) * 32 + g * 8 + (qf & 1) * 4 + r;
          int qq = q0 + ql;
          bool ok = kok && qq < p.Nq && !(p.causal && key > qq);
          float pv = ok ? __builtin_amdgcn_exp2f(__builtin_fmaf(s[qf][r], sl2, -Ls[ql])) : 0.f;
          s[qf][r] = pv;
          dp[qf][r] = pv * (dp[qf][r] - Ds[ql]);
        }
    }
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      float a4[4] = {s[2 * qb][0], s[2 * qb][1], s[2 * qb][2], s[2 * qb][3]};
      float b4[4] = {s[2 * qb + 1][0], s[2 * qb + 1][1], s[2 * qb + 1][2], s[2 * qb + 1][3]};
      bf16x8 pf = pack8(a4, b4);
      float c4[4] = {dp[2 * qb][0], dp[2 * qb][1], dp[2 * qb][2], dp[2 * qb][3]};
      float e4[4] = {dp[2 * qb + 1][0], dp[2 * qb + 1][1], dp[2 * qb + 1][2], dp[2 * qb + 1][3]};
      bf16x8 dsf = pack8(c4, e4);
#pragma unroll
      for (int df = 0; df < DV / 16; ++df) {
        bf16x8 gtf = lds_tr_frag(Gs + troff + qb * 32 * NSTR + ((df ^ tsw) << 5), NSTR);
        dv[df] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gtf, pf, dv[df], 0, 0, 0);
        bf16x8 qtf = lds_tr_frag(Qs + troff + qb * 32 * NSTR + ((df ^ tsw) << 5), NSTR);
        dk[df] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qtf, dsf, dk[df], 0, 0, 0);
      }
    }
    if (more) {
      if (!DB) __syncthreads();   // single LDS buffer (DP = 160): everyone must be done reading it
      sstore_all(smem + (DB ? ((qt - qt_lo + 1) & 1) * KBUF : 0));
    }
    __syncthreads();
  }
  if (key < p.Nkp) {   // pad rows [Nk, Nkp) receive zeros (their accumulators are zero: p == 0 there)
    const int64_t row = (int64_t)b * p.Nkp + key;
#pragma unroll
    for (int df = 0; df < DV / 16; ++df) {
      int col = df * 16 + g * 4;
      if (col < d) {
        if (p.qsplit > 1) {
          // partial dK / dV of this query range -> slab[split] (plain stores; attn_splitsum_kernel adds the slabs in order and converts).  Until round 5 the splits
          // met in ONE fp32 buffer through float atomics: the only order-dependent sum left on the LoRA / TI path (SD1.5's 160-wide cross-attention heads, which the
          // single-pass cross kernel does not take) - tools/determinism_probe.py sd15 found the step differing run to run through it.
          const int64_t srow = ((int64_t)split * p.B + b) * p.Nkp + key;
          *(f32x4*)(p.dK32 + srow * p.ld32 + hc + col) = (f32x4){dk[df][0] * p.scale, dk[df][1] * p.scale, dk[df][2] * p.scale, dk[df][3] * p.scale};
          *(f32x4*)(p.dV32 + srow * p.ld32 + hc + col) = dv[df];
        } else {
          uint2 w;
          w.x = pack2bf(dk[df][0] * p.scale, dk[df][1] * p.scale); w.y = pack2bf(dk[df][2] * p.scale, dk[df][3] * p.scale);
          *(uint2*)((bf16_t*)p.dK + row * p.lddk + hc + col) = w;
          w.x = pack2bf(dv[df][0], dv[df][1]); w.y = pack2bf(dv[df][2], dv[df][3]);
          *(uint2*)((bf16_t*)p.dV + row * p.lddv + hc + col) = w;
        }
      }
    }
  }
}

// =============================================================================== backward launch forms
template <int DP, int DV = DP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(DP == 64 ? 3 : 1, DP == 64 ? 3 : 8))) void attn_bwd_dq_kernel(const sdlt_attn_params p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const WgId wg = xcd_wg();
  attn_bwd_dq_body<DP, true, DV>(p, smem, wg.x, wg);
}
template <int DP, int DV = DP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(DP == 64 ? 3 : 1, DP == 64 ? 3 : 8))) void attn_bwd_dkdv_kernel(const sdlt_attn_params p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const WgId wg = xcd_wg();
  attn_bwd_dkdv_body<DP, DV>(p, smem, wg.x, wg);
}
// Self-attention: the dQ tiles and the dK/dV tiles of one layer in ONE launch (blockIdx.x < #query tiles: dQ role).  At
// 1024 tokens x 20 heads either pass alone is 320 workgroups of 16 dependent steps - latency-bound, the chip half empty;
// together they overlap.  The dK/dV role needs D of every query row, so D comes from attn_prep_kernel here.
template <int DP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(DP == 64 ? 3 : 1, DP == 64 ? 3 : 8))) void attn_bwd_both_kernel(const sdlt_attn_params p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int ndq = (p.Nq + 63) / 64;
  const WgId wg = xcd_wg();
  if (wg.x < ndq) attn_bwd_dq_body<DP, false>(p, smem, wg.x, wg);
  else attn_bwd_dkdv_body<DP>(p, smem, wg.x - ndq, wg);
}
// D[b,h,q] = sum_d dO*O (8 lanes per (row, head))
__device__ __forceinline__ void attn_prep_body(const bf16_t* O, int64_t ldo, const bf16_t* dO, int64_t lddo, int B, int H, int Nq, int Nqp, int d, float* D, const int bx, const int gx) {
  const int total = B * Nqp * H;                 // (< 2^22: the launcher checks; 64-bit divisions were ~300 of the kernel's ~450 instructions)
  const int sub = threadIdx.x & 7;
  for (int idx = (bx * (int)blockDim.x + (int)threadIdx.x) >> 3; idx < total; idx += (gx * (int)blockDim.x) >> 3) {
    const int row = div_small(idx, H);   // b*Nqp + q
    const int h = idx - row * H;
    const int b = div_small(row, Nqp), q = row - b * Nqp;
    if (q >= Nq) continue;   // uniform within the 8-lane group
    float acc = 0.f;
    for (int c = sub * 8; c < d; c += 64) {
      uint4 a = *(const uint4*)(O + (int64_t)row * ldo + h * d + c);
      uint4 g = *(const uint4*)(dO + (int64_t)row * lddo + h * d + c);
      const uint32_t* ap = (const uint32_t*)&a;
      const uint32_t* gp = (const uint32_t*)&g;
#pragma unroll
      for (int j = 0; j < 4; ++j) acc += bf2f(ap[j] & 0xffff) * bf2f(gp[j] & 0xffff) + bf2f(ap[j] >> 16) * bf2f(gp[j] >> 16);
    }
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    acc += __shfl_xor(acc, 4, 64);
    if (sub == 0) D[((int64_t)b * H + h) * Nq + q] = acc;
  }
}
__global__ void attn_prep_kernel(const bf16_t* O, int64_t ldo, const bf16_t* dO, int64_t lddo, int B, int H, int Nq, int Nqp, int d, float* D) {
  attn_prep_body(O, ldo, dO, lddo, B, H, Nq, Nqp, d, D, blockIdx.x, gridDim.x);
}
__global__ void attn_prep_pair_kernel(const sdlt_attn_params p0, const sdlt_attn_params p1) {
  const sdlt_attn_params& p = blockIdx.y == 0 ? p0 : p1;
  attn_prep_body((const bf16_t*)p.O, p.ldo, (const bf16_t*)p.dO, p.lddo, p.B, p.H, p.Nq, p.Nqp, p.d, p.D, blockIdx.x, gridDim.x);
}
template <int DP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(DP == 64 ? 3 : 1, DP == 64 ? 3 : 8))) void attn_bwd_both_pair_kernel(const sdlt_attn_params p0, const sdlt_attn_params p1) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  WgId wg = xcd_wg();
  if (wg.y < p0.H) {
    const int ndq = (p0.Nq + 63) / 64, nkv = (p0.Nk + 63) / 64;
    if (wg.x >= ndq + nkv || wg.z >= p0.B) return;
    if (wg.x < ndq) attn_bwd_dq_body<DP, false>(p0, smem, wg.x, wg);
    else attn_bwd_dkdv_body<DP>(p0, smem, wg.x - ndq, wg);
  } else {
    wg.y -= p0.H;
    const int ndq = (p1.Nq + 63) / 64, nkv = (p1.Nk + 63) / 64;
    if (wg.x >= ndq + nkv || wg.z >= p1.B) return;
    if (wg.x < ndq) attn_bwd_dq_body<DP, false>(p1, smem, wg.x, wg);
    else attn_bwd_dkdv_body<DP>(p1, smem, wg.x - ndq, wg);
  }
}

// =============================================================================== backward, cross-attention: dQ, dK, dV (and D) in ONE pass
// Nk <= 128 keys (the 77 text tokens): one workgroup owns ALL keys of one (batch, head) and a range of 64-query tiles.
// Per tile it runs both orientations of the score product: q-major (this wave's 16 query rows x 128 keys -> dQ rows,
// written once, plus D = rowsum(dO * O), which replaces the separate prep kernel) and key-major (its 2 x 16 keys x
// the 64 queries -> dK / dV accumulated in registers over the whole query range, then fp32 atomics).  Replaces the
// prep + dQ + query-split dK/dV launches of the generic path for the UNet's cross-attention.
// FIVE (64 < Nk <= 80, the 77 text tokens): the keys are FIVE 16-key blocks.  Wave w owns block w against all 64 queries of a tile as
// before; the fifth block (keys 64..79) is shared - wave w takes it against one 16-query set of the tile and the four partial dK / dV
// blocks meet through LDS at the end (the two-blocks-per-wave layout spent half of every wave's key-major work on keys 80..127, which
// do not exist, and gave wave 0 twice the live work of the others); the q-major pass skips the dead 32-key blocks of its second half.
// ROLE 0: both orientations in one workgroup (as described above).  ROLE 1 / 2: the q-major pass (dQ) and the key-major pass (dK / dV slabs) as
// SEPARATE workgroups of one launch (attn_bwd_cross_roles_kernel).  One workgroup doing both needs 346 registers - one wave per SIMD, every
// LDS / MFMA latency of its ~11 k clocks per 64-query tile exposed (tools/attn_trace.py); split, each role fits 256 registers, two workgroups
// share a CU and a tile costs each of them one orientation.  The key-major role computes D = rowsum(dO * O) itself (it stages the O tile too).
template <int DP, bool FIVE, int ROLE>
__device__ __forceinline__ void attn_bwd_cross_body(const sdlt_attn_params& p, char* smem, const int split, const int nsplit) {
  constexpr int NSTR = tile_stride<DP>();
  constexpr bool QM = ROLE != 2, KM = ROLE != 1;        // q-major / key-major pass present
  constexpr int QBUF = (ROLE == 2 ? 3 : 2) * 64 * NSTR; // Q, dO (+ O for ROLE 2) natural
  char* Ks = smem;                                      // [128][NSTR], resident (q-major only)
  char* Vs = Ks + 128 * NSTR;
  char* ring = QM ? Vs + 128 * NSTR : smem;             // 2 x QBUF
  float* LD = (float*)(ring + 2 * QBUF);                // L*log2e [64], D [64] of the current tile
  const int b = blockIdx.z, h = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), g = lane >> 4, i = lane & 15;
  const int d = p.d, hc = h * d;
  const float sl2 = p.scale * LOG2E;
  const int troff = (8 * g + (i >> 2)) * NSTR + (i & 3) * 8;
  const int tsw = row_sw<DP>(8 * g + (i >> 2)) >> 1;   // this lane's XOR on the 32-byte column block of a transposing read

#ifdef SDLT_ATTN_TRACE
  const bool tr = (ROLE == 2 ? (int)blockIdx.x == (int)gridDim.x - p.qsplit : blockIdx.x == 0) && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0 && p.D;
  long long* trb = (long long*)p.D + (ROLE == 2 ? 64 : 0);      // (the key-major role's stamps sit 64 slots further)
  int trn = 0;
  TR();
#endif
  TileRegs<DP> qr, gr, orr;
  if constexpr (QM) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {                       // resident K / V tiles (rows >= Nkp read as zero)
      const int nk = min(64, max(0, p.Nkp - t * 64));
      gload_nat<DP>(qr, (const bf16_t*)p.K, p.ldk, (int64_t)b * p.Nkp + t * 64, nk, hc, d);
      gload_nat<DP>(gr, (const bf16_t*)p.V, p.ldv, (int64_t)b * p.Nkp + t * 64, nk, hc, d);
      sstore_nat<DP>(qr, Ks + t * 64 * NSTR);
      sstore_nat<DP>(gr, Vs + t * 64 * NSTR);
    }
  }
  // key-major operands: this wave's keys w*64 + wave*16 + i
  int key[2];
  bool kok[2];
  bf16x8 kf[2][DP / 32], vf[2][DP / 32];
  f32x4 dk[2][DP / 16], dv[2][DP / 16];
#pragma unroll
  for (int w = 0; w < 2; ++w) {
    key[w] = (FIVE && w == 1) ? 64 + i : w * 64 + wave * 16 + i;
    kok[w] = key[w] < p.Nk;
#pragma unroll
    for (int kk = 0; kk < DP / 32; ++kk) {
      int col = kk * 32 + g * 8;
      kf[w][kk] = ld_frag_global((const bf16_t*)p.K + ((int64_t)b * p.Nkp + key[w]) * p.ldk + hc + col, KM && kok[w] && col < d);
      vf[w][kk] = ld_frag_global((const bf16_t*)p.V + ((int64_t)b * p.Nkp + key[w]) * p.ldv + hc + col, KM && kok[w] && col < d);
    }
#pragma unroll
    for (int df = 0; df < DP / 16; ++df) {
      dk[w][df] = (f32x4){0.f, 0.f, 0.f, 0.f};
      dv[w][df] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  }
  const int nqt = (p.Nq + 63) / 64;
  const int per = (nqt + nsplit - 1) / nsplit;          // (nsplit: workgroups of this role along the queries; the key-major role's is p.qsplit = the slab count)
  const int qt_lo = split * per, qt_hi = min(nqt, qt_lo + per);
  float lreg2 = 0.f;                                   // ROLE 2: L of the tile whose loads are in flight (threads 0..63)
  auto gload_all = [&](int qt) {
    const int q0 = qt * 64;
    const int nq = min(64, p.Nqp - q0);
    gload_nat<DP>(qr, (const bf16_t*)p.Q, p.ldq, (int64_t)b * p.Nqp + q0, nq, hc, d);
    gload_nat<DP>(gr, (const bf16_t*)p.dO, p.lddo, (int64_t)b * p.Nqp + q0, nq, hc, d);
    if constexpr (ROLE == 2) {
      gload_nat<DP>(orr, (const bf16_t*)p.O, p.ldo, (int64_t)b * p.Nqp + q0, nq, hc, d);
      if (threadIdx.x < 64) {
        const int qq = q0 + threadIdx.x;
        lreg2 = qq < p.Nq ? p.L[((int64_t)b * p.H + h) * p.Nq + qq] * LOG2E : 0.f;
      }
    }
  };
  auto sstore_all = [&](char* base) {
    sstore_nat<DP>(qr, base);
    sstore_nat<DP>(gr, base + 64 * NSTR);
    if constexpr (ROLE == 2) sstore_nat<DP>(orr, base + 2 * 64 * NSTR);
  };
  // O fragments and L of this lane's q-major row, fetched one tile ahead like the Q / dO tiles (as plain loads at their point of use they
  // put two exposed global round trips at the head of every tile)
  bf16x8 ofn[DP / 32];
  float Lqn = 0.f;
  auto gload_ol = [&](int qt) {
    if constexpr (!QM) return;
    const int q = qt * 64 + wave * 16 + i;
    const bool qok = q < p.Nq;
#pragma unroll
    for (int kk = 0; kk < DP / 32; ++kk) {
      const int col = kk * 32 + g * 8;
      ofn[kk] = ld_frag_global((const bf16_t*)p.O + ((int64_t)b * p.Nqp + q) * p.ldo + hc + col, qok && col < d);
    }
    Lqn = qok ? p.L[((int64_t)b * p.H + h) * p.Nq + q] * LOG2E : 0.f;
  };
  if (qt_lo < qt_hi) {
    gload_all(qt_lo);
    gload_ol(qt_lo);
    sstore_all(ring);
  }
  __syncthreads();
  TR();
  for (int qt = qt_lo; qt < qt_hi; ++qt) {
    const int q0 = qt * 64;
    const char* Qs = ring + ((qt - qt_lo) & 1) * QBUF;
    const char* Gs = Qs + 64 * NSTR;
    const bool more = qt + 1 < qt_hi;
    const float lcur = lreg2;                          // (ROLE 2) L of THIS tile, loaded a tile ahead
    bf16x8 of[DP / 32];
#pragma unroll
    for (int kk = 0; kk < DP / 32; ++kk) of[kk] = ofn[kk];
    const float Lq = Lqn;
    if (more) {
      gload_all(qt + 1);
      gload_ol(qt + 1);
    }

    if constexpr (ROLE == 2) {
      // D = rowsum(dO * O) and L of the tile's 64 rows: four lanes per row, 16 columns each, straight from the staged tiles
      const char* Os = Gs + 64 * NSTR;
      const int row = threadIdx.x >> 2, part = threadIdx.x & 3;
      float acc = 0.f;
#pragma unroll
      for (int c = 0; c < DP / 32; ++c) {
        const int ch = part * (DP / 32) + c;                                  // 16-byte chunk of the row
        const bf16x8 gv = *(const bf16x8*)(Gs + row * NSTR + ((ch ^ row_sw<DP>(row)) << 4));
        const bf16x8 ov = *(const bf16x8*)(Os + row * NSTR + ((ch ^ row_sw<DP>(row)) << 4));
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += (float)gv[j] * (float)ov[j];
      }
      acc += __shfl_xor(acc, 1, 64);
      acc += __shfl_xor(acc, 2, 64);
      if (part == 0) LD[64 + row] = acc;
      if (threadIdx.x < 64) LD[threadIdx.x] = lcur;
    }
    // ---------------- q-major: this wave's rows q0 + wave*16 + i against all 128 keys -> D, dQ
    if constexpr (QM) {
      const int ql = wave * 16 + i, q = q0 + ql;
      const bool qok = q < p.Nq;
      bf16x8 qf[DP / 32], gf[DP / 32];
      float Dq = 0.f;
#pragma unroll
      for (int kk = 0; kk < DP / 32; ++kk) {
        qf[kk] = *(const bf16x8*)(Qs + ql * NSTR + (((kk * 4 + g) ^ row_sw<DP>(ql)) << 4));
        gf[kk] = *(const bf16x8*)(Gs + ql * NSTR + (((kk * 4 + g) ^ row_sw<DP>(ql)) << 4));
#pragma unroll
        for (int j = 0; j < 8; ++j) Dq += (float)gf[kk][j] * (float)of[kk][j];
      }
      Dq += __shfl_xor(Dq, 16, 64);
      Dq += __shfl_xor(Dq, 32, 64);
      if (ROLE == 0 && g == 0) {
        LD[ql] = Lq;
        LD[64 + ql] = Dq;
      }
      f32x4 dq[DP / 16];
#pragma unroll
      for (int df = 0; df < DP / 16; ++df) dq[df] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        if (half * 64 >= p.Nk) break;                   // uniform: no live key in the second half
        f32x4 s2[4], dp2[4];
#pragma unroll
        for (int f = 0; f < 4; ++f) {
          s2[f] = (f32x4){0.f, 0.f, 0.f, 0.f};
          dp2[f] = (f32x4){0.f, 0.f, 0.f, 0.f};
          if (FIVE && half == 1 && f >= 2) continue;          // keys 96..127
          const int krow = half * 64 + (f >> 1) * 32 + prow(i, f & 1);
#pragma unroll
          for (int kk = 0; kk < DP / 32; ++kk) {
            bf16x8 kfr = *(const bf16x8*)(Ks + krow * NSTR + (((kk * 4 + g) ^ row_sw<DP>(krow)) << 4));
            s2[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfr, qf[kk], s2[f], 0, 0, 0);
            bf16x8 vfr = *(const bf16x8*)(Vs + krow * NSTR + (((kk * 4 + g) ^ row_sw<DP>(krow)) << 4));
            dp2[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vfr, gf[kk], dp2[f], 0, 0, 0);
          }
        }
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (FIVE && half == 1 && f >= 2) continue;
            const int kx = half * 64 + (f >> 1) * 32 + g * 8 + (f & 1) * 4 + r;
            const bool ok = qok && kx < p.Nk;
            const float pv = ok ? __builtin_amdgcn_exp2f(__builtin_fmaf(s2[f][r], sl2, -Lq)) : 0.f;
            s2[f][r] = pv * (dp2[f][r] - Dq) * p.scale;  // dS
          }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
          if (FIVE && half == 1 && kb == 1) continue;
          float a4[4] = {s2[2 * kb][0], s2[2 * kb][1], s2[2 * kb][2], s2[2 * kb][3]};
          float b4[4] = {s2[2 * kb + 1][0], s2[2 * kb + 1][1], s2[2 * kb + 1][2], s2[2 * kb + 1][3]};
          bf16x8 dsf = pack8(a4, b4);
#pragma unroll
          for (int df = 0; df < DP / 16; ++df) {
            bf16x8 ktf = lds_tr_frag(Ks + troff + (half * 64 + kb * 32) * NSTR + ((df ^ tsw) << 5), NSTR);
            dq[df] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ktf, dsf, dq[df], 0, 0, 0);
          }
        }
      }
      if (q < p.Nqp) {   // pad rows: dq == 0
        // accumulate_dq: the buffer already holds the score side output's dQ (batched GEMM before the backward pass).  All DP / 16 old words are requested at once
        // (unconditional per block, column clamped) - read where they are added, each was a round trip of its own behind the last MFMA (round 5, ISA scan)
        uint2 oldq[DP / 16];
        if (p.accumulate_dq) {
          const bf16_t* qrow = (const bf16_t*)p.dQ + ((int64_t)b * p.Nqp + q) * p.lddq + hc;
#pragma unroll
          for (int df = 0; df < DP / 16; ++df) { const int col = df * 16 + g * 4; oldq[df] = *(const uint2*)(qrow + (col < d ? col : 0)); }
        }
#pragma unroll
        for (int df = 0; df < DP / 16; ++df) {
          int col = df * 16 + g * 4;
          if (col < d) {
            uint2* dst = (uint2*)((bf16_t*)p.dQ + ((int64_t)b * p.Nqp + q) * p.lddq + hc + col);
            if (p.accumulate_dq) {
              const uint2 old = oldq[df];
              dq[df][0] += bf2f(old.x & 0xffff); dq[df][1] += bf2f(old.x >> 16);
              dq[df][2] += bf2f(old.y & 0xffff); dq[df][3] += bf2f(old.y >> 16);
            }
            uint2 wv;
            wv.x = pack2bf(dq[df][0], dq[df][1]);
            wv.y = pack2bf(dq[df][2], dq[df][3]);
            *dst = wv;
          }
        }
      }
    }
    TR();
    __syncthreads();   // L, D of all 64 rows visible
    TR();

    // ---------------- key-major: this wave's 2 x 16 keys against the tile's 64 queries -> dK, dV
    if constexpr (KM) {
      const float* Ls = LD;
      const float* Ds = LD + 64;
      constexpr int NWK = FIVE ? 1 : 2;      // 64-query key blocks of this wave
      f32x4 s[NWK][4], dp[NWK][4];
      f32x4 s5 = (f32x4){0.f, 0.f, 0.f, 0.f}, dp5 = (f32x4){0.f, 0.f, 0.f, 0.f};     // FIVE: block 4 x this wave's 16-query set (f == wave)
#pragma unroll
      for (int f = 0; f < 4; ++f) {
#pragma unroll
        for (int w = 0; w < NWK; ++w) {
          s[w][f] = (f32x4){0.f, 0.f, 0.f, 0.f};
          dp[w][f] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        const int qrow = (f >> 1) * 32 + prow(i, f & 1);
#pragma unroll
        for (int kk = 0; kk < DP / 32; ++kk) {
          bf16x8 qfr = *(const bf16x8*)(Qs + qrow * NSTR + (((kk * 4 + g) ^ row_sw<DP>(qrow)) << 4));
          bf16x8 gfr = *(const bf16x8*)(Gs + qrow * NSTR + (((kk * 4 + g) ^ row_sw<DP>(qrow)) << 4));
#pragma unroll
          for (int w = 0; w < NWK; ++w) {
            s[w][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qfr, kf[w][kk], s[w][f], 0, 0, 0);
            dp[w][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gfr, vf[w][kk], dp[w][f], 0, 0, 0);
          }
          if (FIVE && f == wave) {
            s5 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qfr, kf[1][kk], s5, 0, 0, 0);
            dp5 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gfr, vf[1][kk], dp5, 0, 0, 0);
          }
        }
      }
      if constexpr (FIVE) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ql2 = (wave >> 1) * 32 + g * 8 + (wave & 1) * 4 + r;
          const bool ok = kok[1] && q0 + ql2 < p.Nq;
          const float pv = ok ? __builtin_amdgcn_exp2f(__builtin_fmaf(s5[r], sl2, -Ls[ql2])) : 0.f;
          s5[r] = pv;
          dp5[r] = pv * (dp5[r] - Ds[ql2]) * p.scale;
        }
      }
#pragma unroll
      for (int w = 0; w < NWK; ++w)
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int ql2 = (f >> 1) * 32 + g * 8 + (f & 1) * 4 + r;
            const bool ok = kok[w] && q0 + ql2 < p.Nq;
            const float pv = ok ? __builtin_amdgcn_exp2f(__builtin_fmaf(s[w][f][r], sl2, -Ls[ql2])) : 0.f;
            s[w][f][r] = pv;
            dp[w][f][r] = pv * (dp[w][f][r] - Ds[ql2]) * p.scale;
          }
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) {
        bf16x8 pf[NWK], dsf[NWK];
#pragma unroll
        for (int w = 0; w < NWK; ++w) {
          float a4[4] = {s[w][2 * qb][0], s[w][2 * qb][1], s[w][2 * qb][2], s[w][2 * qb][3]};
          float b4[4] = {s[w][2 * qb + 1][0], s[w][2 * qb + 1][1], s[w][2 * qb + 1][2], s[w][2 * qb + 1][3]};
          pf[w] = pack8(a4, b4);
          float c4[4] = {dp[w][2 * qb][0], dp[w][2 * qb][1], dp[w][2 * qb][2], dp[w][2 * qb][3]};
          float e4[4] = {dp[w][2 * qb + 1][0], dp[w][2 * qb + 1][1], dp[w][2 * qb + 1][2], dp[w][2 * qb + 1][3]};
          dsf[w] = pack8(c4, e4);
        }
#pragma unroll
        for (int df = 0; df < DP / 16; ++df) {
          bf16x8 gtf = lds_tr_frag(Gs + troff + qb * 32 * NSTR + ((df ^ tsw) << 5), NSTR);
          bf16x8 qtf = lds_tr_frag(Qs + troff + qb * 32 * NSTR + ((df ^ tsw) << 5), NSTR);
#pragma unroll
          for (int w = 0; w < NWK; ++w) {
            dv[w][df] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gtf, pf[w], dv[w][df], 0, 0, 0);
            dk[w][df] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qtf, dsf[w], dk[w][df], 0, 0, 0);
          }
          if (FIVE && qb == (wave >> 1)) {      // this wave's 16 queries sit in the (wave & 1) half of the 32-query contraction; the other half is zero
            const float z4[4] = {0.f, 0.f, 0.f, 0.f};
            const float p4[4] = {s5[0], s5[1], s5[2], s5[3]}, e4[4] = {dp5[0], dp5[1], dp5[2], dp5[3]};
            const bf16x8 pf5 = (wave & 1) ? pack8(z4, p4) : pack8(p4, z4), dsf5 = (wave & 1) ? pack8(z4, e4) : pack8(e4, z4);
            dv[1][df] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gtf, pf5, dv[1][df], 0, 0, 0);
            dk[1][df] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qtf, dsf5, dk[1][df], 0, 0, 0);
          }
        }
      }
    }
    TR();
    if (more) sstore_all(ring + ((qt - qt_lo + 1) & 1) * QBUF);
    __syncthreads();
    TR();
  }
  // partial dK / dV of this query range -> slab[split][b*Nkp + key][C] (plain 16-byte stores; attn_splitsum_kernel adds the
  // slabs and converts).  Float atomics on the 77 x C block shared by every split were the whole cost of the old path.
  if constexpr (!KM) return;
  if constexpr (FIVE) {
    // the four waves' partial dK / dV of key block 4 -> LDS (the tile ring is dead) -> wave w adds and keeps column block df == w
    f32x4* red = (f32x4*)ring;             // [wave][dk | dv][df][lane]
#pragma unroll
    for (int df = 0; df < DP / 16; ++df) {
      red[((wave * 2 + 0) * (DP / 16) + df) * 64 + lane] = dk[1][df];
      red[((wave * 2 + 1) * (DP / 16) + df) * 64 + lane] = dv[1][df];
    }
    __syncthreads();
#pragma unroll
    for (int df = 0; df < DP / 16; ++df) {
      f32x4 a = (f32x4){0.f, 0.f, 0.f, 0.f}, c = a;
      if (df == wave || (DP / 16 > 4 && df == wave + 4)) {
#pragma unroll
        for (int w2 = 0; w2 < 4; ++w2) {
          a += red[((w2 * 2 + 0) * (DP / 16) + df) * 64 + lane];
          c += red[((w2 * 2 + 1) * (DP / 16) + df) * 64 + lane];
        }
      }
      dk[1][df] = a;
      dv[1][df] = c;
    }
  }
#pragma unroll
  for (int w = 0; w < 2; ++w)
    if (kok[w]) {
      const int64_t row = ((int64_t)split * p.B + b) * p.Nkp + key[w];
#pragma unroll
      for (int df = 0; df < DP / 16; ++df) {
        int col = df * 16 + g * 4;
        if (FIVE && w == 1 && df != wave && !(DP / 16 > 4 && df == wave + 4)) continue;     // (another wave holds this column block's sum)
        if (col < d) {
          *(f32x4*)(p.dK32 + row * p.ld32 + hc + col) = dk[w][df];
          *(f32x4*)(p.dV32 + row * p.ld32 + hc + col) = dv[w][df];
        }
      }
    }
  TR();
}

template <int DP, bool FIVE>
__global__ __launch_bounds__(256) void attn_bwd_cross_kernel(const sdlt_attn_params p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  attn_bwd_cross_body<DP, FIVE, 0>(p, smem, blockIdx.x, p.qsplit);
}
// grid.x = nq + qsplit: the first nq workgroups of a (head, batch element) are the q-major role, the other qsplit the key-major role (one
// dK / dV slab each).  A key-major tile costs ~1.6x a q-major one and that role also writes the slabs, so it gets the finer split.
template <int DP, bool FIVE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(DP == 64 ? 2 : 1, DP == 64 ? 2 : 8))) void attn_bwd_cross_roles_kernel(const sdlt_attn_params p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int nq = (int)gridDim.x - p.qsplit;
  if ((int)blockIdx.x < nq) attn_bwd_cross_body<DP, FIVE, 1>(p, smem, blockIdx.x, nq);
  else attn_bwd_cross_body<DP, FIVE, 2>(p, smem, blockIdx.x - nq, p.qsplit);
}

// out[b*Nkp + key][c] = bf16(sum over splits of slab[split][b*Nkp + key][c]); pad keys [Nk, Nkp) get zeros
__global__ void attn_splitsum_kernel(const float* s0, const float* s1, int nsplit, int64_t ld32, bf16_t* out0, int64_t ldo0, bf16_t* out1, int64_t ldo1,
                                     int B, int Nk, int Nkp, int C, int acc0) {
  const int nch = C >> 2;
  const int64_t per = (int64_t)B * Nkp * nch, sstride = (int64_t)B * Nkp * ld32;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < 2 * per; t += (int64_t)gridDim.x * blockDim.x) {
    const bool second = t >= per;
    const int64_t u = second ? t - per : t;
    const int r = u / nch, c = (u - (int64_t)r * nch) * 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r % Nkp < Nk) {
      const float* src = (second ? s1 : s0) + r * ld32 + c;
      for (int sp = 0; sp < nsplit; ++sp) {
        float4 v = *(const float4*)(src + sp * sstride);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
    }
    uint2* dst = (uint2*)((second ? out1 + r * ldo1 : out0 + r * ldo0) + c);
    if (acc0 && !second) {   // dK += : the buffer holds the score side output's dK
      const uint2 old = *dst;
      acc.x += bf2f(old.x & 0xffff); acc.y += bf2f(old.x >> 16); acc.z += bf2f(old.y & 0xffff); acc.w += bf2f(old.y >> 16);
    }
    uint2 w;
    w.x = pack2bf(acc.x, acc.y); w.y = pack2bf(acc.z, acc.w);
    *dst = w;
  }
}

// the same for a table of layers in one launch (sdlt_attn_splitsum_batch): every layer owns d.nblocks consecutive blocks
__global__ void attn_splitsum_batch_kernel(const sdlt_splitsum_desc* descs, const int32_t* block_desc, const int32_t* block_first) {
  const sdlt_splitsum_desc d = descs[block_desc[blockIdx.x]];
  const int blk = blockIdx.x - block_first[block_desc[blockIdx.x]];
  const int nch = d.C >> 2;
  const int64_t per = (int64_t)d.B * d.Nkp * nch, sstride = (int64_t)d.B * d.Nkp * d.ld32;
  for (int64_t t = blk * (int64_t)blockDim.x + threadIdx.x; t < 2 * per; t += (int64_t)d.nblocks * blockDim.x) {
    const bool second = t >= per;
    const int64_t u = second ? t - per : t;
    const int r = u / nch, c = (u - (int64_t)r * nch) * 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r % d.Nkp < d.Nk) {
      const float* src = (second ? d.s1 : d.s0) + r * d.ld32 + c;
      for (int sp = 0; sp < d.nsplit; ++sp) {
        float4 v = *(const float4*)(src + sp * sstride);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
    }
    uint2* dst = (uint2*)((second ? (bf16_t*)d.out1 + r * d.ldo1 : (bf16_t*)d.out0 + r * d.ldo0) + c);
    if (d.acc0 && !second) {
      const uint2 old = *dst;
      acc.x += bf2f(old.x & 0xffff); acc.y += bf2f(old.x >> 16); acc.z += bf2f(old.y & 0xffff); acc.w += bf2f(old.y >> 16);
    }
    uint2 w;
    w.x = pack2bf(acc.x, acc.y); w.y = pack2bf(acc.z, acc.w);
    *dst = w;
  }
}

int attn_dp(int d) {
  if (d <= 0 || (d % 8)) return -1;
  if (d <= 64) return 64;
  if (d <= 96) return 96;
  if (d <= 128) return 128;
  if (d <= 160) return 160;
  return -1;
}

int attn_check(const sdlt_attn_params& p, const char* fn) {
  if (p.B <= 0 || p.H <= 0 || p.Nq <= 0 || p.Nk <= 0 || p.Nkp < p.Nk || attn_dp(p.d) < 0)
    SDLT_FAIL(SDLT_ERR_SHAPE, "%s: B=%d H=%d Nq=%d Nk=%d Nkp=%d d=%d (d %% 8 == 0, d <= 160)", fn, p.B, p.H, p.Nq, p.Nk, p.Nkp, p.d);
  if (p.Nqp < p.Nq || (p.Nkp % 8) || (p.Nqp % 8)) SDLT_FAIL(SDLT_ERR_ALIGN, "%s: Nqp/Nkp (padded rows per batch) must be multiples of 8 and >= Nq/Nk", fn);
  if ((p.ldq % 8) || (p.ldk % 8) || (p.ldv % 8)) SDLT_FAIL(SDLT_ERR_ALIGN, "%s: ld %% 8", fn);
  return SDLT_OK;
}

template <typename F>
int set_smem(F f, int bytes) { return sdlt_raise_smem((const void*)f, bytes); }   // once per (kernel, device), thread-safe (capi.cpp)

#define ATTN_DISPATCH(DPV, KERNEL, GRID, SMEM)                                                  \
  switch (DPV) {                                                                                \
    case 64: set_smem(KERNEL<64>, SMEM(64)); hipLaunchKernelGGL(KERNEL<64>, GRID, dim3(256), SMEM(64), s, p); break;     \
    case 96: set_smem(KERNEL<96>, SMEM(96)); hipLaunchKernelGGL(KERNEL<96>, GRID, dim3(256), SMEM(96), s, p); break;     \
    case 128: set_smem(KERNEL<128>, SMEM(128)); hipLaunchKernelGGL(KERNEL<128>, GRID, dim3(256), SMEM(128), s, p); break; \
    default: set_smem(KERNEL<160>, SMEM(160)); hipLaunchKernelGGL(KERNEL<160>, GRID, dim3(256), SMEM(160), s, p); break; \
  }

}  // namespace

static bool attn32_on() {
  static const bool on = !(getenv("SDLT_ATTN_R32") && atoi(getenv("SDLT_ATTN_R32")) == 0);
  return on;
}
static void attn_env_once() {
  static bool done = false;
  if (done) return;
  done = true;
  if (getenv("SDLT_ATTN_XCD") && atoi(getenv("SDLT_ATTN_XCD")) == 0) {
    const int zero = 0;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_attn_xcd), &zero, sizeof(int));
  }
}

extern "C" int sdlt_attn_fwd(const sdlt_attn_params* pp, void* stream) {
  attn_env_once();
  const sdlt_attn_params& p = *pp;
  hipStream_t s = (hipStream_t)stream;
  int rc = attn_check(p, "sdlt_attn_fwd");
  if (rc) return rc;
  if (p.ldo % 4) SDLT_FAIL(SDLT_ERR_ALIGN, "sdlt_attn_fwd: ldo %% 4");
  const int dp = attn_dp(p.d);
  dim3 grid((p.Nq + 63) / 64, p.H, p.B);
  // head width 64, whole 64-row tiles, no mask: 32 rows per wave on the 32x32x16 MFMA (attn32.hip).  SDLT_ATTN_R32=0 keeps the 16-row kernels (A/B).
  // Two wave groups per workgroup split the key tiles wherever there are >= 2 of them (tools/attn32_probe.sh, 1 / 2 / 4 groups: 1024 tokens x 20 heads
  // 23.8 / 16.7 / 20.6 us, 4096 x 10 90.5 / 80.0 / 84.3, 4 x 256 x 20 8.9 / 7.5 / 11.5, 4 x 1024 x 10 26.2 / 24.5 / 29.7)
  if (attn32_on() && sdlt_attn32_ok(p) && ((uintptr_t)p.O % 16) == 0 && (p.ldo % 8) == 0) {      // (16-byte epilogue stores)
    static const int ks_env = getenv("SDLT_ATTN32_KS_FWD") ? atoi(getenv("SDLT_ATTN32_KS_FWD")) : 0;
    return sdlt_attn32_fwd(p, ks_env > 0 ? ks_env : (p.Nk >= 128 ? 2 : 1), s);      // (zeroes D in its epilogue)
  }
  if (p.D) sdlt_zero_async(p.D, sizeof(float) * (size_t)p.B * p.H * p.Nq, s);
#define NSTRH(D_) ((D_) == 64 ? 128 : (D_) * 2 + 16)
#define SMEM_FWD(D_) (2 * (2 * 64 * NSTRH(D_)))
  // the key split (two wave groups per workgroup) pays on latency-bound grids only - less than two workgroups per CU and a chain of >= 4
  // key tiles (tools/attn_probe.py: 1024 tokens x 20 heads 25.6 -> 22.9 us; 4096 x 10, 640 workgroups: 102 -> 114 us, so not there);
  // SDLT_ATTN_KS=1 switches it off (A/B)
  static const int ks_env = getenv("SDLT_ATTN_KS") ? atoi(getenv("SDLT_ATTN_KS")) : 2;
  if (dp == 64 && ks_env == 2 && p.Nk >= 256 && (int64_t)grid.x * grid.y * grid.z <= 512) {
    set_smem(attn_fwd_kernel<64, 2>, 2 * SMEM_FWD(64));
    hipLaunchKernelGGL((attn_fwd_kernel<64, 2>), grid, dim3(512), 2 * SMEM_FWD(64), s, p);
    SDLT_CHECK_LAUNCH();
    return SDLT_OK;
  }
  static const bool dv48 = !(getenv("SDLT_ATTN_DV48") && atoi(getenv("SDLT_ATTN_DV48")) == 0);      // (A/B switch: 0 = four output blocks for heads of <= 48 columns too)
  if (dv48 && dp == 64 && p.d <= 48) {
    set_smem((attn_fwd_kernel<64, 1, 48>), SMEM_FWD(64));
    hipLaunchKernelGGL((attn_fwd_kernel<64, 1, 48>), grid, dim3(256), SMEM_FWD(64), s, p);
    SDLT_CHECK_LAUNCH();
    return SDLT_OK;
  }
  ATTN_DISPATCH(dp, attn_fwd_kernel, grid, SMEM_FWD)
  SDLT_CHECK_LAUNCH();
  return SDLT_OK;
}

extern "C" int sdlt_attn_bwd(const sdlt_attn_params* pp, void* stream) {
  attn_env_once();
  const sdlt_attn_params& p = *pp;
  hipStream_t s = (hipStream_t)stream;
  int rc = attn_check(p, "sdlt_attn_bwd");
  if (rc) return rc;
  if (!p.L || !p.D || !p.O || !p.dO || !p.dQ)
    SDLT_FAIL(SDLT_ERR_SHAPE, "sdlt_attn_bwd: missing operand (L, D, O, dO, dQ are required)");
  if ((p.lddo % 8) || (p.ldo % 8) || (p.lddq % 4))
    SDLT_FAIL(SDLT_ERR_ALIGN, "sdlt_attn_bwd: ld alignment");
  if (p.qsplit < 1) SDLT_FAIL(SDLT_ERR_SHAPE, "sdlt_attn_bwd: qsplit=%d", p.qsplit);
  if (!p.dK || !p.dV || (p.lddk % 4) || (p.lddv % 4) || (p.qsplit > 1 && (!p.dK32 || !p.dV32 || (p.ld32 % 4))))
    SDLT_FAIL(SDLT_ERR_SHAPE, "sdlt_attn_bwd: dK/dV outputs for qsplit=%d", p.qsplit);
  const int dp = attn_dp(p.d);
  const int C = p.H * p.d, krows = p.B * p.Nkp;
  if (p.qsplit > 1 && !p.causal && p.Nk <= 128 && p.Nkp <= 128 && dp <= 96) {
    // cross-attention: prep + dQ + dK/dV in one kernel (see attn_bwd_cross_kernel); qsplit = workgroups along the queries,
    // dK32 / dV32 = [qsplit][B*Nkp][ld32] partial slabs (any contents)
    dim3 gx(p.qsplit, p.H, p.B);
#define SMEM_X(D_) (2 * 128 * NSTRH(D_) + 2 * (2 * 64 * NSTRH(D_)) + 512)
    static const bool five_env = !(getenv("SDLT_XATTN_FIVE") && atoi(getenv("SDLT_XATTN_FIVE")) == 0);
    const bool five = five_env && p.Nk > 64 && p.Nk <= 80;
    // SDLT_XATTN_ROLES=0: both orientations in one workgroup (the single-role kernel)
    static const bool roles_env = !(getenv("SDLT_XATTN_ROLES") && atoi(getenv("SDLT_XATTN_ROLES")) == 0);
    static const int qdiv_env = getenv("SDLT_XATTN_QDIV") ? atoi(getenv("SDLT_XATTN_QDIV")) : 1;      // q-major workgroups = qsplit / this
    const int nq_roles = p.qsplit / (qdiv_env > 0 ? qdiv_env : 1) > 0 ? p.qsplit / (qdiv_env > 0 ? qdiv_env : 1) : 1;
#define SMEM_XR(D_) (3 * 2 * 64 * NSTRH(D_) > 2 * 128 * NSTRH(D_) + 2 * 2 * 64 * NSTRH(D_) ? 3 * 2 * 64 * NSTRH(D_) + 512 : 2 * 128 * NSTRH(D_) + 2 * 2 * 64 * NSTRH(D_) + 512)
#define XLAUNCH(D_, F_) do { \
      if (roles_env && (D_) == 64) {       /* (head width 96: the split roles spill) */  set_smem(attn_bwd_cross_roles_kernel<D_, F_>, SMEM_XR(D_)); hipLaunchKernelGGL((attn_bwd_cross_roles_kernel<D_, F_>), dim3(nq_roles + p.qsplit, p.H, p.B), dim3(256), SMEM_XR(D_), s, p); } \
      else { set_smem(attn_bwd_cross_kernel<D_, F_>, SMEM_X(D_)); hipLaunchKernelGGL((attn_bwd_cross_kernel<D_, F_>), gx, dim3(256), SMEM_X(D_), s, p); } } while (0)
    if (dp == 64) { if (five) XLAUNCH(64, true); else XLAUNCH(64, false); }
    else { if (five) XLAUNCH(96, true); else XLAUNCH(96, false); }
    if (!p.defer_splitsum) {
      int blocks = (int)(((int64_t)krows * C / 2 + 255) / 256);
      if (blocks > 4096) blocks = 4096;
      hipLaunchKernelGGL(attn_splitsum_kernel, dim3(blocks), dim3(256), 0, s, p.dK32, p.dV32, p.qsplit, p.ld32, (bf16_t*)p.dK, p.lddk,
                         (bf16_t*)p.dV, p.lddv, p.B, p.Nk, p.Nkp, C, p.accumulate_dk);
    }
    SDLT_CHECK_LAUNCH();
    return SDLT_OK;
  }
  if (p.qsplit == 1 && !p.accumulate_dq && !p.accumulate_dk &&
      (int64_t)((p.Nq + 63) / 64 + (p.Nk + 63) / 64) * p.H * p.B <= 2048) {   // (bigger grids are throughput-bound: two launches are 5 % faster there)
    // self-attention (UNet, text encoders): D, then dQ and dK/dV tiles in one launch (see attn_bwd_both_kernel)
    int64_t groups = (int64_t)p.B * p.Nqp * p.H;
    if (groups >= (1 << 22)) SDLT_FAIL(SDLT_ERR_UNSUPPORTED, "sdlt_attn_bwd: B * Nq * H = %lld rows (the D pre-pass indexes < 2^22)", (long long)groups);
    int blocks = (int)((groups * 8 + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    if (!p.d_ready)      // (d_ready: sdlt_wsk_gemm_rowdot left D while it produced dO)
      hipLaunchKernelGGL(attn_prep_kernel, dim3(blocks), dim3(256), 0, s, (const bf16_t*)p.O, p.ldo, (const bf16_t*)p.dO, p.lddo, p.B, p.H, p.Nq, p.Nqp, p.d, p.D);
    if (attn32_on() && sdlt_attn32_ok(p) && ((uintptr_t)p.L % 16) == 0 && ((uintptr_t)p.D % 16) == 0 &&
        (((uintptr_t)p.dQ | (uintptr_t)p.dK | (uintptr_t)p.dV) % 16) == 0 && ((p.lddq | p.lddk | p.lddv) % 8) == 0) {      // (16-byte epilogue stores)
      static const int ks_env = getenv("SDLT_ATTN32_KS_BWD") ? atoi(getenv("SDLT_ATTN32_KS_BWD")) : 0;
      // (1 / 2 / 4 groups: 1024 x 20 44.0 / 38.1 / 46.4 us, 4096 x 10 219.9 / 203.3 / 201.5, 4 x 1024 x 10 72.0 / 60.1 / 75.1)
      return sdlt_attn32_bwd_both(p, ks_env > 0 ? ks_env : (p.Nk >= 128 && p.Nq >= 128 ? 2 : 1), s);
    }
    dim3 gb((p.Nq + 63) / 64 + (p.Nk + 63) / 64, p.H, p.B);
#define SMEM_BOTH(D_) (2 * (2 * 64 * NSTRH(D_) + 512))
    ATTN_DISPATCH(dp, attn_bwd_both_kernel, gb, SMEM_BOTH)
    SDLT_CHECK_LAUNCH();
    return SDLT_OK;
  }
  if (p.accumulate_dq || p.accumulate_dk) SDLT_FAIL(SDLT_ERR_UNSUPPORTED, "sdlt_attn_bwd: accumulate_dq/dk exist for the single-pass cross-attention kernel only");
  dim3 gq((p.Nq + 63) / 64, p.H, p.B);
#define SMEM_DQ(D_) (2 * (2 * 64 * NSTRH(D_)))
  static const bool dv48 = !(getenv("SDLT_ATTN_DV48") && atoi(getenv("SDLT_ATTN_DV48")) == 0);
  const bool v48 = dv48 && dp == 64 && p.d <= 48;
  if (v48) { set_smem((attn_bwd_dq_kernel<64, 48>), SMEM_DQ(64)); hipLaunchKernelGGL((attn_bwd_dq_kernel<64, 48>), gq, dim3(256), SMEM_DQ(64), s, p); }
  else { ATTN_DISPATCH(dp, attn_bwd_dq_kernel, gq, SMEM_DQ) }
  if (p.qsplit > 1 && (((uintptr_t)p.dK32 | (uintptr_t)p.dV32) & 15 || (p.ld32 & 3) || (C & 3)))
    SDLT_FAIL(SDLT_ERR_ALIGN, "sdlt_attn_bwd: the dK / dV slabs of a query-split backward need 16-byte rows (ld32 %% 4, C %% 4)");
  dim3 gk(((p.Nk + 63) / 64) * p.qsplit, p.H, p.B);
#define SMEM_DKV(D_) (2 * (2 * 64 * NSTRH(D_) + 512))
  if (v48) { set_smem((attn_bwd_dkdv_kernel<64, 48>), SMEM_DKV(64)); hipLaunchKernelGGL((attn_bwd_dkdv_kernel<64, 48>), gk, dim3(256), SMEM_DKV(64), s, p); }
  else { ATTN_DISPATCH(dp, attn_bwd_dkdv_kernel, gk, SMEM_DKV) }
  if (p.qsplit > 1) {      // [qsplit][B * Nkp][ld32] partial slabs -> bf16 dK / dV, summed in slab order (bitwise reproducible)
    int blocks = (int)(((int64_t)krows * C / 2 + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(attn_splitsum_kernel, dim3(blocks), dim3(256), 0, s, p.dK32, p.dV32, p.qsplit, p.ld32, (bf16_t*)p.dK, p.lddk, (bf16_t*)p.dV, p.lddv, p.B, p.Nk, p.Nkp, C, 0);
  }
  SDLT_CHECK_LAUNCH();
  return SDLT_OK;
}

extern "C" int sdlt_attn_splitsum_batch(const sdlt_splitsum_desc* descs_dev, const int32_t* block_desc_dev, const int32_t* block_first_dev, int32_t n_blocks,
                                        void* stream) {
  if (n_blocks <= 0) SDLT_FAIL(SDLT_ERR_SHAPE, "sdlt_attn_splitsum_batch: n_blocks=%d", n_blocks);
  hipLaunchKernelGGL(attn_splitsum_batch_kernel, dim3(n_blocks), dim3(256), 0, (hipStream_t)stream, descs_dev, block_desc_dev, block_first_dev);
  SDLT_CHECK_LAUNCH();
  return SDLT_OK;
}

// reference: none (launch structure only).  Two causal / plain self-attention problems of head width <= 64 in one launch each way; the
// caller falls back to two single calls when sdlt_attn_pair_ok says no.
static bool attn_pair_shape_ok(const sdlt_attn_params& a, const sdlt_attn_params& b) {
  return attn_dp(a.d) == 64 && attn_dp(b.d) == 64 && a.Nk < 256 && b.Nk < 256 && a.qsplit == 1 && b.qsplit == 1 && !a.accumulate_dq && !b.accumulate_dq &&
         !a.accumulate_dk && !b.accumulate_dk && (a.Nq + 63) / 64 + (a.Nk + 63) / 64 <= 64 && (b.Nq + 63) / 64 + (b.Nk + 63) / 64 <= 64;
}
extern "C" int sdlt_attn_pair_ok(const sdlt_attn_params* a, const sdlt_attn_params* b) { return attn_pair_shape_ok(*a, *b) ? 1 : 0; }

extern "C" int sdlt_attn_fwd_pair(const sdlt_attn_params* pa, const sdlt_attn_params* pb, void* stream) {
  attn_env_once();
  const sdlt_attn_params &a = *pa, &b = *pb;
  int rc = attn_check(a, "sdlt_attn_fwd_pair");
  if (rc) return rc;
  rc = attn_check(b, "sdlt_attn_fwd_pair");
  if (rc) return rc;
  if ((a.ldo % 4) || (b.ldo % 4)) SDLT_FAIL(SDLT_ERR_ALIGN, "sdlt_attn_fwd_pair: ldo %% 4");
  if (!attn_pair_shape_ok(a, b)) SDLT_FAIL(SDLT_ERR_UNSUPPORTED, "sdlt_attn_fwd_pair: shapes (d <= 64, Nk < 256 both)");
  const int ta = (a.Nq + 63) / 64, tb = (b.Nq + 63) / 64;
  dim3 grid(ta > tb ? ta : tb, a.H + b.H, a.B > b.B ? a.B : b.B);
  set_smem(attn_fwd_pair_kernel<64>, SMEM_FWD(64));
  hipLaunchKernelGGL(attn_fwd_pair_kernel<64>, grid, dim3(256), SMEM_FWD(64), (hipStream_t)stream, a, b);
  SDLT_CHECK_LAUNCH();
  return SDLT_OK;
}

extern "C" int sdlt_attn_bwd_pair(const sdlt_attn_params* pa, const sdlt_attn_params* pb, void* stream) {
  attn_env_once();
  const sdlt_attn_params &a = *pa, &b = *pb;
  hipStream_t s = (hipStream_t)stream;
  for (const sdlt_attn_params* q : {pa, pb}) {
    const sdlt_attn_params& p = *q;
    int rc = attn_check(p, "sdlt_attn_bwd_pair");
    if (rc) return rc;
    if (!p.L || !p.D || !p.O || !p.dO || !p.dQ || !p.dK || !p.dV) SDLT_FAIL(SDLT_ERR_SHAPE, "sdlt_attn_bwd_pair: missing operand");
    if ((p.lddo % 8) || (p.ldo % 8) || (p.lddq % 4) || (p.lddk % 4) || (p.lddv % 4)) SDLT_FAIL(SDLT_ERR_ALIGN, "sdlt_attn_bwd_pair: ld alignment");
  }
  if (!attn_pair_shape_ok(a, b)) SDLT_FAIL(SDLT_ERR_UNSUPPORTED, "sdlt_attn_bwd_pair: shapes");
  const int64_t ga = (int64_t)a.B * a.Nqp * a.H, gb = (int64_t)b.B * b.Nqp * b.H;
  int blocks = (int)(((ga > gb ? ga : gb) * 8 + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(attn_prep_pair_kernel, dim3(blocks, 2), dim3(256), 0, s, a, b);
  const int xa = (a.Nq + 63) / 64 + (a.Nk + 63) / 64, xb = (b.Nq + 63) / 64 + (b.Nk + 63) / 64;
  dim3 grid(xa > xb ? xa : xb, a.H + b.H, a.B > b.B ? a.B : b.B);
  set_smem(attn_bwd_both_pair_kernel<64>, SMEM_BOTH(64));
  hipLaunchKernelGGL(attn_bwd_both_pair_kernel<64>, grid, dim3(256), SMEM_BOTH(64), s, a, b);
  SDLT_CHECK_LAUNCH();
  return SDLT_OK;
}
