// Self-attention forward / backward for head width 64 with 32 query (key) rows per wave on the 32x32x16 MFMA (gfx950).
//
// Replaces, for the UNet's attn1 layers (F.scaled_dot_product_attention inside diffusers' AttnProcessor2_0 reached from main.py:329-336, and its
// autograd backward), the 16-rows-per-wave kernels of attn.hip on the shapes where they were issue-bound (round 3: 6.8 % of the MFMA peak; a
// 64-key step cost a wave ~960 SIMD cycles for 256 cycles of MFMA - the rest softmax VALU, cross-lane steps and LDS fragment reads).  What the
// wider layout changes per score: every wave reads a K / V (Q / dO) tile once per 32 rows instead of once per 16 (half the LDS fragment bytes),
// the row statistics of a query live in ONE lane pair (one v_permlane32_swap per reduction instead of two ds_bpermute), and max / sum run as
// v_max3 / adds over 32 in-lane scores.
//
// Layout trick (same idea as attn.hip's `prow`, for the 32x32 shape): all products are computed swapped, D[i][j] with j = the row that owns the
// statistics (query in forward / dQ, key in dK / dV).  The 32x32 accumulator gives lane (hl = lane >> 5, j = lane & 31) the rows
// rho = 8*(r >> 2) + 4*hl + (r & 3), r = 0..15; the next MFMA wants, as its B operand, k = 8*hl + e (e = 0..7) of a 16-row chunk.  Feeding the
// A operand's row rho with tile row rmap(rho) = rho with bits 2 and 3 swapped makes registers 8c..8c+7 of the accumulator exactly chunk c's
// B fragment: no cross-lane traffic between the two products.
//
// LDS tiles are [64 rows][128 bytes] images filled by LDS-DMA (global_load_lds, 16 B per lane, no staging registers) with the 16-byte chunk index
// XOR-ed on the SOURCE side by sw32(row) = (row bit 1) << 2 | (row bits 3:2): conflict-free for the permuted ds_read_b128 row fragments (every
// 16-lane service group of the instruction touches 16 distinct 16-byte bank groups) and for the transposing ds_read_b64_tr_b16 fragments (a
// 32-lane group reads 4 rows x 64 bytes; rows r and r + 2 land in different halves of the 128-byte line).
//
// Shapes: d == 64, Nq % 64 == 0, Nk % 64 == 0, no causal mask (sdlt_attn32_ok); everything else stays on attn.hip.
// KS: the key tiles (forward, dQ role) / query tiles (dK / dV role) of a 64-row workgroup are dealt to KS groups of two waves with their own LDS
// rings; the groups' partial results meet through LDS in group order (bitwise reproducible).
#include "common.h"
#include "../../include/sdlt_kernels.h"
#include "attn32.h"

namespace {

constexpr float LOG2E = 1.4426950408889634f;
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int TILE = 64 * 128;

__device__ __forceinline__ int sw32(int row) { return (((row >> 1) & 1) << 2) | ((row >> 2) & 3); }
__device__ __forceinline__ int rmap(int rho) { return (rho & 0x13) | ((rho & 4) << 1) | ((rho & 8) >> 1); }

struct WgId { int x, y, z; };
// XCD k (workgroups are dealt round-robin to the 8 XCDs) runs a contiguous range of the (tile, head, batch) order: the workgroups of one head
// share its K / V (Q / dO) rows through one L2 (see attn.hip)
__device__ __forceinline__ WgId xcd_wg() {
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

// one [64 rows][64 columns] bf16 tile global -> LDS by two waves (4 DMA pieces of 8 rows each per wave)
__device__ __forceinline__ void tile_dma(const bf16_t* src, int64_t ld, char* dst, int w2, int lane) {
#pragma unroll
  for (int pc = 0; pc < 4; ++pc) {
    const int piece = w2 * 4 + pc;
    const int r = piece * 8 + (lane >> 3);
    glds16(src + (int64_t)r * ld + (((lane & 7) ^ sw32(r)) << 3), dst + piece * 1024);
  }
}

// Every LDS fragment read of the main loops is inline asm with hand-placed s_waitcnt lgkmcnt: (1) hipcc puts an s_waitcnt vmcnt(0) in front of a
// transposing-read BUILTIN that follows an LDS-DMA (it cannot prove that the read does not alias the tile in flight), which would serialise the
// next tile's DMA with this tile's products; (2) it issues each compiler-visible ds_read right before its MFMA (read - wait - MFMA chains).  The
// asm reads are issued a phase ahead and waited for once; `pin` makes their consumers depend on the wait (volatile asms keep their order).
template <int OFF>
__device__ __forceinline__ bf16x8 lds_b128(uint32_t a) {
  bf16x8 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(OFF));
  return v;
}
template <int OFF>
__device__ __forceinline__ bf16x4 lds_tr(uint32_t a) {
  bf16x4 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(OFF));
  return v;
}
template <int N>
__device__ __forceinline__ void wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void pin(bf16x8& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void pin(bf16x4& v) { asm volatile("" : "+v"(v)); }
// the four row fragments (A operand, contraction chunks kk = 0..3) of 32-row block BLK of a tile
template <int BLK>
__device__ __forceinline__ void rows_issue(bf16x8 (&f)[4], uint32_t tile, const int (&row)[4]) {
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) f[kk] = lds_b128<BLK * 4096>(tile + row[kk]);
}
__device__ __forceinline__ void rows_pin(bf16x8 (&f)[4]) {
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) pin(f[kk]);
}
// the transposed fragments (A operand: 32 columns of block db x 16 rows of chunk CH) of a tile, as 4 x 8-byte pieces: [db][lo | hi]
struct TrFrag { bf16x4 v[4]; };
template <int CH>
__device__ __forceinline__ void tr_issue(TrFrag& f, uint32_t tile, const int (&t1)[2], const int (&t2)[2]) {
  f.v[0] = lds_tr<CH * 2048>(tile + t1[0]);
  f.v[1] = lds_tr<CH * 2048>(tile + t2[0]);
  f.v[2] = lds_tr<CH * 2048>(tile + t1[1]);
  f.v[3] = lds_tr<CH * 2048>(tile + t2[1]);
}
__device__ __forceinline__ void tr_pin(TrFrag& f) {
#pragma unroll
  for (int k = 0; k < 4; ++k) pin(f.v[k]);
}
__device__ __forceinline__ bf16x8 tr_get(const TrFrag& f, int db) { return __builtin_shufflevector(f.v[2 * db], f.v[2 * db + 1], 0, 1, 2, 3, 4, 5, 6, 7); }
__device__ __forceinline__ uint32_t lds_addr(const char* p) { return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)p; }
__device__ __forceinline__ bf16x8 ld8(const bf16_t* p) {
  union { uint4 u; bf16x8 v; } c;
  c.u = *(const uint4*)p;
  return c.v;
}
__device__ __forceinline__ bf16x8 pack8v(const f32x16& v, int c2) {
  union { uint4 u; bf16x8 b; } c;
  c.u.x = pack2bf(v[8 * c2 + 0], v[8 * c2 + 1]); c.u.y = pack2bf(v[8 * c2 + 2], v[8 * c2 + 3]);
  c.u.z = pack2bf(v[8 * c2 + 4], v[8 * c2 + 5]); c.u.w = pack2bf(v[8 * c2 + 6], v[8 * c2 + 7]);
  return c.b;
}
// v_permlane32_swap with both operands = v leaves {v.lo, v.lo} and {v.hi, v.hi}: the combination over the two half-waves, in every lane
__device__ __forceinline__ float half_max(float v) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float half_sum(float v) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// Epilogue stores as 16-byte pieces: lane (hl, c) holds, for every 8-column group j of an output block, the four columns 8 j + 4 hl .. + 3 of row c - an 8-byte
// store per group, 32 rows x 16 bytes per wave instruction, and the store tail of a launch is store-ISSUE-bound (MI355X_MICROARCH.md: 16 x dwordx2 per lane ~ 9.3 k
// cycles, halved by 8 x dwordx4).  One v_permlane32_swap per dword hands the lane pair (c, c + 32) each other's halves: the lower lane ends up with the even group's
// eight columns, the upper lane with the odd group's - half as many stores, twice as wide.  ev / od: the packed groups 2 jp / 2 jp + 1 of this lane.
// -DSDLT_A32_STORE16=0 builds the 8-byte form (A/B).
#ifndef SDLT_A32_STORE16
#define SDLT_A32_STORE16 1
#endif
// p = the row's pointer at column 8 hl of the output block's 16-column pair jp (elements)
__device__ __forceinline__ void store_pair(bf16_t* p, int hl, uint2 ev, uint2 od) {
#if SDLT_A32_STORE16
  auto x = __builtin_amdgcn_permlane32_swap(ev.x, od.x, false, false);       // -> {ev.lo | od.lo}, {ev.hi | od.hi} over (lower | upper) half-wave
  auto y = __builtin_amdgcn_permlane32_swap(ev.y, od.y, false, false);
  *(uint4*)p = make_uint4(x[0], y[0], x[1], y[1]);
#else
  *(uint2*)(p - 4 * hl) = ev;
  *(uint2*)(p - 4 * hl + 8) = od;
#endif
}

__device__ __forceinline__ void wait_dma_barrier() {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}

// per-lane LDS offsets of the two fragment kinds inside a 64-row tile
struct FragOff {
  int row[4];      // row fragment (A operand, 32 permuted rows x 16 columns): + blk * 4096; chunk kk
  int t1[2], t2[2];  // transposed fragment (A operand, 32 columns x 16 rows): + (blk * 32 + c2 * 16) * 128; column block db
};
__device__ __forceinline__ FragOff frag_offsets(int lane) {
  FragOff f;
  const int hl = lane >> 5, c = lane & 31, rr = rmap(c);
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) f.row[kk] = rr * 128 + (((2 * kk + hl) ^ sw32(rr)) << 4);
  const int g = lane >> 4, i = lane & 15;
  const int R = 8 * (g >> 1) + (i >> 2), swl = (((i >> 3) & 1) << 2) | ((g >> 1) << 1), chl = 2 * (g & 1) + ((i & 3) >> 1);
#pragma unroll
  for (int db = 0; db < 2; ++db) {
    f.t1[db] = R * 128 + (((4 * db + chl) ^ swl) << 4) + (i & 1) * 8;
    f.t2[db] = (R + 4) * 128 + (((4 * db + chl) ^ swl ^ 1) << 4) + (i & 1) * 8;
  }
  return f;
}

#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)

// SDLT_A32_DMA_SPLIT (default 1): the next tile's LDS-DMA pieces are issued in two halves BEHIND MFMA groups of the current tile instead of in one
// burst at the loop top.  After the per-iteration barrier all waves of a CU issue their 8 pieces at once and each `global_load_lds` blocks its wave
// while the CU's 64 B/clk texture path takes the 1 KB (tools/attn32_trace.py: 570-1000 of the 2800-4800 cycles of an iteration); behind MFMAs that
// wait overlaps the matrix pipe.  A/B in one session (tools/attn32_ab.sh): 1024 x 20 forward 17.0 -> 16.4 us, backward 37.0 -> 36.2; 4096 x 10
// 80 -> 78, 200 -> 196.  -DSDLT_A32_DMA_SPLIT=0 builds the burst form.
#ifndef SDLT_A32_DMA_SPLIT
#define SDLT_A32_DMA_SPLIT 1
#endif

// -DSDLT_ATTN32_TRACE (tools/attn32_trace.py): lane 0 of wave 0 of the first workgroup of a role stamps clock64() at the phase boundaries of its
// loop into `trace_buf` (forward: p.D, backward: p.dK32 - both unused by these kernels otherwise; dQ role slots 0.., dK / dV role 512..)
#ifdef SDLT_ATTN32_TRACE
#define TR32_DECL(BUF_, ON_) long long* tr_ = (long long*)(BUF_); int tn_ = 0; const bool tron_ = (ON_) && tr_ != nullptr
#define TR32() do { if (tron_ && tn_ < 500) { __builtin_amdgcn_sched_barrier(0); tr_[tn_++] = clock64(); __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define TR32_DECL(BUF_, ON_)
#define TR32() do { } while (0)
#endif

// =============================================================================== forward
template <int KS>
__global__ __launch_bounds__(128 * KS) void attn32_fwd_kernel(const sdlt_attn_params p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const WgId wg = xcd_wg();
  const int b = wg.z, h = wg.y, q0 = wg.x * 64;
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int grp = wv >> 1, w2 = wv & 1, hl = lane >> 5, c = lane & 31;
  const int hc = h * 64;
  const int q = q0 + w2 * 32 + c;
  const bf16_t* Kb = (const bf16_t*)p.K + (int64_t)b * p.Nkp * p.ldk + hc;
  const bf16_t* Vb = (const bf16_t*)p.V + (int64_t)b * p.Nkp * p.ldv + hc;
  char* ring = smem + grp * (4 * TILE);
  const int ntile = p.Nk >> 6, niter = (ntile + KS - 1) / KS;
  if (grp < ntile) {
    tile_dma(Kb + (int64_t)grp * 64 * p.ldk, p.ldk, ring, w2, lane);
    tile_dma(Vb + (int64_t)grp * 64 * p.ldv, p.ldv, ring + TILE, w2, lane);
  }
  bf16x8 qf[4];
  {
    const bf16_t* qp = (const bf16_t*)p.Q + ((int64_t)b * p.Nqp + q) * p.ldq + hc + 8 * hl;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) qf[kk] = ld8(qp + 16 * kk);
  }
  const FragOff fo = frag_offsets(lane);
  float m = -1e30f, lsum = 0.f;
  f32x16 o[2];
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
  const float sl2 = p.scale * LOG2E;
  TR32_DECL(p.D, wg.x == 0 && wg.y == 0 && wg.z == 0 && threadIdx.x == 0);
  TR32();
  wait_dma_barrier();
  for (int it = 0; it < niter; ++it) {
    TR32();
    const int t = it * KS + grp;
    const char* Ks = ring + (it & 1) * (2 * TILE);
    const char* Vs = Ks + TILE;
    char* nb = ring + ((it + 1) & 1) * (2 * TILE);
    const bool more = t + KS < ntile;
#if SDLT_A32_DMA_SPLIT == 0
    if (more) {
      tile_dma(Kb + (int64_t)(t + KS) * 64 * p.ldk, p.ldk, nb, w2, lane);
      tile_dma(Vb + (int64_t)(t + KS) * 64 * p.ldv, p.ldv, nb + TILE, w2, lane);
    }
#endif
    TR32();
    if (t < ntile) {
      const uint32_t ka = lds_addr(Ks), va = lds_addr(Vs);
      bf16x8 k0f[4], k1f[4];
      TrFrag vt[4];
      rows_issue<0>(k0f, ka, fo.row);
      rows_issue<1>(k1f, ka, fo.row);
      tr_issue<0>(vt[0], va, fo.t1, fo.t2);
      tr_issue<1>(vt[1], va, fo.t1, fo.t2);
      f32x16 s[2];
#pragma unroll
      for (int blk = 0; blk < 2; ++blk)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[blk][r] = 0.f;
      wait_lgkm<12>();
      rows_pin(k0f);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) s[0] = MFMA32(k0f[kk], qf[kk], s[0]);
#if SDLT_A32_DMA_SPLIT       // the next tile's DMA pieces go out under the MFMAs just issued (their issue blocks on the CU's texture path)
      if (more) tile_dma(Kb + (int64_t)(t + KS) * 64 * p.ldk, p.ldk, nb, w2, lane);
#endif
      tr_issue<2>(vt[2], va, fo.t1, fo.t2);
      tr_issue<3>(vt[3], va, fo.t1, fo.t2);
      wait_lgkm<15>();          // 24 reads issued: the oldest 9 (both row-fragment sets) have landed
      rows_pin(k1f);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) s[1] = MFMA32(k1f[kk], qf[kk], s[1]);
#if SDLT_A32_DMA_SPLIT
      if (more) tile_dma(Vb + (int64_t)(t + KS) * 64 * p.ldv, p.ldv, nb + TILE, w2, lane);
#endif
      __builtin_amdgcn_sched_barrier(0);
      TR32();
      float tmax = s[0][0];
#pragma unroll
      for (int blk = 0; blk < 2; ++blk)
#pragma unroll
        for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, s[blk][r]);
      tmax = half_max(tmax);
      const float mn = fmaxf(m, tmax * sl2);
      const float alpha = __builtin_amdgcn_exp2f(m - mn);
      m = mn;
      // (v_pk_fma_f32 / v_pk_add_f32 over score pairs and a wave-uniform skip of the rescale were measured: 25 % fewer VALU instructions, the
      // same time to 0.3 us on every shape - packed f32 issues at half rate next to MFMAs, MI355X_MICROARCH.md)
      float rs = 0.f;
#pragma unroll
      for (int blk = 0; blk < 2; ++blk)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(s[blk][r], sl2, -mn));
          s[blk][r] = pv;
          rs += pv;
        }
      lsum = lsum * alpha + rs;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
      __builtin_amdgcn_sched_barrier(0);
      TR32();
      wait_lgkm<0>();
#pragma unroll
      for (int ch = 0; ch < 4; ++ch) tr_pin(vt[ch]);
#pragma unroll
      for (int ch = 0; ch < 4; ++ch) {
        const bf16x8 pf = pack8v(s[ch >> 1], ch & 1);
#pragma unroll
        for (int db = 0; db < 2; ++db) o[db] = MFMA32(tr_get(vt[ch], db), pf, o[db]);
      }
    }
    TR32();
    wait_dma_barrier();
  }
  TR32();
  lsum = half_sum(lsum);
  if constexpr (KS > 1) {
    // (m, l, O) of groups 1.. -> LDS -> group 0, in group order (the rings are dead after the last barrier)
    float* mg = (float*)smem + (((grp - 1) * 2 + w2) * 64 + lane) * 36;
    if (grp > 0) {
      mg[0] = m;
      mg[1] = lsum;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int j = 0; j < 4; ++j) *(f32x4*)(mg + 4 + db * 16 + j * 4) = (f32x4){o[db][4 * j], o[db][4 * j + 1], o[db][4 * j + 2], o[db][4 * j + 3]};
    }
    __syncthreads();
    if (grp > 0) return;
#pragma unroll
    for (int g2 = 1; g2 < KS; ++g2) {
      const float* sg = (const float*)smem + (((g2 - 1) * 2 + w2) * 64 + lane) * 36;
      const float m1 = sg[0], l1 = sg[1];
      const float mn = fmaxf(m, m1), a0 = __builtin_amdgcn_exp2f(m - mn), a1 = __builtin_amdgcn_exp2f(m1 - mn);
      m = mn;
      lsum = lsum * a0 + l1 * a1;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f32x4 o1 = *(const f32x4*)(sg + 4 + db * 16 + j * 4);
#pragma unroll
          for (int r = 0; r < 4; ++r) o[db][4 * j + r] = o[db][4 * j + r] * a0 + o1[r] * a1;
        }
    }
  }
  const float inv = 1.f / lsum;
  if (hl == 0 && p.L) p.L[((int64_t)b * p.H + h) * p.Nq + q] = (m + log2f(lsum)) / LOG2E;
#ifndef SDLT_ATTN32_TRACE
  if (hl == 0 && p.D) p.D[((int64_t)b * p.H + h) * p.Nq + q] = 0.f;      // the slots sdlt_wsk_gemm_rowdot accumulates rowsum(dO o O) into during the backward pass
#endif
  bf16_t* op = (bf16_t*)p.O + ((int64_t)b * p.Nqp + q) * p.ldo + hc + 8 * hl;
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int jp = 0; jp < 2; ++jp) {
      uint2 w[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int j = 2 * jp + e;
        w[e].x = pack2bf(o[db][4 * j] * inv, o[db][4 * j + 1] * inv);
        w[e].y = pack2bf(o[db][4 * j + 2] * inv, o[db][4 * j + 3] * inv);
      }
      store_pair(op + 32 * db + 16 * jp, hl, w[0], w[1]);
    }
}

// =============================================================================== backward, dQ role: 64 query rows, key tiles dealt to KS groups
template <int KS>
__device__ __forceinline__ void attn32_dq_body(const sdlt_attn_params& p, char* smem, const int bx, const WgId wg) {
  const int b = wg.z, h = wg.y, q0 = bx * 64;
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int grp = wv >> 1, w2 = wv & 1, hl = lane >> 5, c = lane & 31;
  const int hc = h * 64;
  const int q = q0 + w2 * 32 + c;
  const bf16_t* Kb = (const bf16_t*)p.K + (int64_t)b * p.Nkp * p.ldk + hc;
  const bf16_t* Vb = (const bf16_t*)p.V + (int64_t)b * p.Nkp * p.ldv + hc;
  char* ring = smem + grp * (4 * TILE);
  const int ntile = p.Nk >> 6, niter = (ntile + KS - 1) / KS;
  if (grp < ntile) {
    tile_dma(Kb + (int64_t)grp * 64 * p.ldk, p.ldk, ring, w2, lane);
    tile_dma(Vb + (int64_t)grp * 64 * p.ldv, p.ldv, ring + TILE, w2, lane);
  }
  bf16x8 qf[4], gf[4];
  {
    const bf16_t* qp = (const bf16_t*)p.Q + ((int64_t)b * p.Nqp + q) * p.ldq + hc + 8 * hl;
    const bf16_t* gp = (const bf16_t*)p.dO + ((int64_t)b * p.Nqp + q) * p.lddo + hc + 8 * hl;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      qf[kk] = ld8(qp + 16 * kk);
      gf[kk] = ld8(gp + 16 * kk);
    }
  }
  const float Lq = p.L[((int64_t)b * p.H + h) * p.Nq + q] * LOG2E;
  const float Dq = p.D[((int64_t)b * p.H + h) * p.Nq + q];
  const FragOff fo = frag_offsets(lane);
  f32x16 dq[2];
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[db][r] = 0.f;
  const float sl2 = p.scale * LOG2E;
  TR32_DECL(p.dK32, bx == 0 && wg.y == 0 && wg.z == 0 && threadIdx.x == 0);
  TR32();
  wait_dma_barrier();
  for (int it = 0; it < niter; ++it) {
    TR32();
    const int t = it * KS + grp;
    const char* Ks = ring + (it & 1) * (2 * TILE);
    const char* Vs = Ks + TILE;
    char* nb = ring + ((it + 1) & 1) * (2 * TILE);
    const bool more = t + KS < ntile;
#if SDLT_A32_DMA_SPLIT == 0
    if (more) {
      tile_dma(Kb + (int64_t)(t + KS) * 64 * p.ldk, p.ldk, nb, w2, lane);
      tile_dma(Vb + (int64_t)(t + KS) * 64 * p.ldv, p.ldv, nb + TILE, w2, lane);
    }
#endif
    TR32();
    if (t < ntile) {
      const uint32_t ka = lds_addr(Ks), va = lds_addr(Vs);
      bf16x8 kr[2][4], vr[2][4];
      TrFrag kt[2][2];
      rows_issue<0>(kr[0], ka, fo.row);
      rows_issue<0>(vr[0], va, fo.row);
      tr_issue<0>(kt[0][0], ka, fo.t1, fo.t2);
      tr_issue<1>(kt[0][1], ka, fo.t1, fo.t2);
#pragma unroll
      for (int blk = 0; blk < 2; ++blk) {
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
        wait_lgkm<8>();           // the row fragments of this block (the 8 transposed pieces issued after them may still fly)
        rows_pin(kr[blk]);
        rows_pin(vr[blk]);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          s = MFMA32(kr[blk][kk], qf[kk], s);
          dp = MFMA32(vr[blk][kk], gf[kk], dp);
        }
        if (blk == 0) {           // block 1's row fragments fly under block 0's softmax arithmetic
          rows_issue<1>(kr[1], ka, fo.row);
          rows_issue<1>(vr[1], va, fo.row);
        }
#if SDLT_A32_DMA_SPLIT
        if (more) {
          if (blk == 0) tile_dma(Kb + (int64_t)(t + KS) * 64 * p.ldk, p.ldk, nb, w2, lane);
          else tile_dma(Vb + (int64_t)(t + KS) * 64 * p.ldv, p.ldv, nb + TILE, w2, lane);
        }
#endif
        __builtin_amdgcn_sched_barrier(0);
        TR32();
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], sl2, -Lq));
          s[r] = pv * (dp[r] - Dq);      // dS / scale (the softmax scale multiplies the finished dQ rows once)
        }
        __builtin_amdgcn_sched_barrier(0);
        TR32();
        if (blk == 0) wait_lgkm<8>(); else wait_lgkm<0>();
        tr_pin(kt[blk][0]);
        tr_pin(kt[blk][1]);
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) {
          const bf16x8 dsf = pack8v(s, c2);
#pragma unroll
          for (int db = 0; db < 2; ++db) dq[db] = MFMA32(tr_get(kt[blk][c2], db), dsf, dq[db]);
        }
        if (blk == 0) {
          tr_issue<2>(kt[1][0], ka, fo.t1, fo.t2);
          tr_issue<3>(kt[1][1], ka, fo.t1, fo.t2);
        }
        TR32();
      }
    }
    wait_dma_barrier();
  }
  TR32();
  if constexpr (KS > 1) {
    float* mg = (float*)smem + (((grp - 1) * 2 + w2) * 64 + lane) * 36;
    if (grp > 0) {
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int j = 0; j < 4; ++j) *(f32x4*)(mg + db * 16 + j * 4) = (f32x4){dq[db][4 * j], dq[db][4 * j + 1], dq[db][4 * j + 2], dq[db][4 * j + 3]};
    }
    __syncthreads();
    if (grp > 0) return;
#pragma unroll
    for (int g2 = 1; g2 < KS; ++g2) {
      const float* sg = (const float*)smem + (((g2 - 1) * 2 + w2) * 64 + lane) * 36;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f32x4 o1 = *(const f32x4*)(sg + db * 16 + j * 4);
#pragma unroll
          for (int r = 0; r < 4; ++r) dq[db][4 * j + r] += o1[r];
        }
    }
  }
  bf16_t* op = (bf16_t*)p.dQ + ((int64_t)b * p.Nqp + q) * p.lddq + hc + 8 * hl;
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int jp = 0; jp < 2; ++jp) {
      uint2 w[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int j = 2 * jp + e;
        w[e].x = pack2bf(dq[db][4 * j] * p.scale, dq[db][4 * j + 1] * p.scale);
        w[e].y = pack2bf(dq[db][4 * j + 2] * p.scale, dq[db][4 * j + 3] * p.scale);
      }
      store_pair(op + 32 * db + 16 * jp, hl, w[0], w[1]);
    }
}

// =============================================================================== backward, dK / dV role: 64 keys, query tiles dealt to KS groups
template <int KS>
__device__ __forceinline__ void attn32_dkdv_body(const sdlt_attn_params& p, char* smem, const int bx, const WgId wg) {
  constexpr int RING = 2 * (2 * TILE + 512);      // two buffers of [Q tile | dO tile | L[64] | D[64]]
  const int b = wg.z, h = wg.y, k0 = bx * 64;
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int grp = wv >> 1, w2 = wv & 1, hl = lane >> 5, c = lane & 31;
  const int hc = h * 64;
  const int key = k0 + w2 * 32 + c;
  const bf16_t* Qb = (const bf16_t*)p.Q + (int64_t)b * p.Nqp * p.ldq + hc;
  const bf16_t* Gb = (const bf16_t*)p.dO + (int64_t)b * p.Nqp * p.lddo + hc;
  const float* Lb = p.L + ((int64_t)b * p.H + h) * p.Nq;
  const float* Db = p.D + ((int64_t)b * p.H + h) * p.Nq;
  char* ring = smem + grp * RING;
  const int ntile = p.Nq >> 6, niter = (ntile + KS - 1) / KS;
  auto stage_q = [&](int t, char* buf) { tile_dma(Qb + (int64_t)t * 64 * p.ldq, p.ldq, buf, w2, lane); };
  auto stage_g = [&](int t, char* buf) {
    tile_dma(Gb + (int64_t)t * 64 * p.lddo, p.lddo, buf + TILE, w2, lane);
    if (lane < 16) glds16((w2 ? Db : Lb) + t * 64 + lane * 4, buf + 2 * TILE + w2 * 256);
  };
  auto stage = [&](int t, char* buf) { stage_q(t, buf); stage_g(t, buf); };
  if (grp < ntile) stage(grp, ring);
  bf16x8 kf[4], vf[4];
  {
    const bf16_t* kp = (const bf16_t*)p.K + ((int64_t)b * p.Nkp + key) * p.ldk + hc + 8 * hl;
    const bf16_t* vp = (const bf16_t*)p.V + ((int64_t)b * p.Nkp + key) * p.ldv + hc + 8 * hl;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      kf[kk] = ld8(kp + 16 * kk);
      vf[kk] = ld8(vp + 16 * kk);
    }
  }
  const FragOff fo = frag_offsets(lane);
  f32x16 dk[2], dv[2];
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[db][r] = 0.f; dv[db][r] = 0.f; }
  const float sl2 = p.scale * LOG2E;
  TR32_DECL(p.dK32 ? p.dK32 + 1024 : nullptr, bx == 0 && wg.y == 0 && wg.z == 0 && threadIdx.x == 0);
  TR32();
  wait_dma_barrier();
  for (int it = 0; it < niter; ++it) {
    TR32();
    const int t = it * KS + grp;
    const char* Qs = ring + (it & 1) * (2 * TILE + 512);
    const char* Gs = Qs + TILE;
    const float* Ls = (const float*)(Gs + TILE);
    const float* Ds = Ls + 64;
    char* nb = ring + ((it + 1) & 1) * (2 * TILE + 512);
    const bool more = t + KS < ntile;
#if SDLT_A32_DMA_SPLIT == 0
    if (more) stage(t + KS, nb);
#endif
    TR32();
    if (t < ntile) {
      const uint32_t qa = lds_addr(Qs), ga = lds_addr(Gs);
      bf16x8 qr[2][4], gr[2][4];
      TrFrag gt[2][2], qt[2][2];
      rows_issue<0>(qr[0], qa, fo.row);
      rows_issue<0>(gr[0], ga, fo.row);
      tr_issue<0>(gt[0][0], ga, fo.t1, fo.t2);
      tr_issue<0>(qt[0][0], qa, fo.t1, fo.t2);
#pragma unroll
      for (int blk = 0; blk < 2; ++blk) {
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
        wait_lgkm<8>();           // this block's row fragments (chunk 0's transposed pieces, issued after them, may still fly)
        rows_pin(qr[blk]);
        rows_pin(gr[blk]);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          s = MFMA32(qr[blk][kk], kf[kk], s);
          dp = MFMA32(gr[blk][kk], vf[kk], dp);
        }
        if (blk == 0) {
          tr_issue<1>(gt[0][1], ga, fo.t1, fo.t2);
          tr_issue<1>(qt[0][1], qa, fo.t1, fo.t2);
        } else {
          tr_issue<3>(gt[1][1], ga, fo.t1, fo.t2);
          tr_issue<3>(qt[1][1], qa, fo.t1, fo.t2);
        }
#if SDLT_A32_DMA_SPLIT
        if (more) {
          if (blk == 0) stage_q(t + KS, nb);
          else stage_g(t + KS, nb);
        }
#endif
        __builtin_amdgcn_sched_barrier(0);
        TR32();
        // s[r] = S[query blk*32 + 16*(r>>3) + 8*hl + (r&7)][key]
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) {
          const int qb = blk * 32 + c2 * 16 + 8 * hl;
          const f32x4 l0 = *(const f32x4*)(Ls + qb), l1 = *(const f32x4*)(Ls + qb + 4);
          const f32x4 d0 = *(const f32x4*)(Ds + qb), d1 = *(const f32x4*)(Ds + qb + 4);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float lq = (e < 4 ? l0[e & 3] : l1[e & 3]) * LOG2E, dd = e < 4 ? d0[e & 3] : d1[e & 3];
            const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(s[8 * c2 + e], sl2, -lq));
            s[8 * c2 + e] = pv;
            dp[8 * c2 + e] = pv * (dp[8 * c2 + e] - dd);     // dS / scale
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        TR32();
        wait_lgkm<8>();           // chunk 0's pieces (chunk 1's are the newest 8)
        tr_pin(gt[blk][0]);
        tr_pin(qt[blk][0]);
        {
          const bf16x8 pf = pack8v(s, 0), dsf = pack8v(dp, 0);
#pragma unroll
          for (int db = 0; db < 2; ++db) {
            dv[db] = MFMA32(tr_get(gt[blk][0], db), pf, dv[db]);
            dk[db] = MFMA32(tr_get(qt[blk][0], db), dsf, dk[db]);
          }
        }
        if (blk == 0) {           // block 1's row fragments and first transposed pieces fly under the rest of block 0
          rows_issue<1>(qr[1], qa, fo.row);
          rows_issue<1>(gr[1], ga, fo.row);
          tr_issue<2>(gt[1][0], ga, fo.t1, fo.t2);
          tr_issue<2>(qt[1][0], qa, fo.t1, fo.t2);
          wait_lgkm<15>();        // (24 in flight at most: the oldest 9 - chunk 1's pieces are the oldest 8)
        } else {
          wait_lgkm<0>();
        }
        tr_pin(gt[blk][1]);
        tr_pin(qt[blk][1]);
        {
          const bf16x8 pf = pack8v(s, 1), dsf = pack8v(dp, 1);
#pragma unroll
          for (int db = 0; db < 2; ++db) {
            dv[db] = MFMA32(tr_get(gt[blk][1], db), pf, dv[db]);
            dk[db] = MFMA32(tr_get(qt[blk][1], db), dsf, dk[db]);
          }
        }
        TR32();
      }
    }
    wait_dma_barrier();
  }
  TR32();
  if constexpr (KS > 1) {
    float* mg = (float*)smem + (((grp - 1) * 2 + w2) * 64 + lane) * 68;
    if (grp > 0) {
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          *(f32x4*)(mg + db * 16 + j * 4) = (f32x4){dk[db][4 * j], dk[db][4 * j + 1], dk[db][4 * j + 2], dk[db][4 * j + 3]};
          *(f32x4*)(mg + 32 + db * 16 + j * 4) = (f32x4){dv[db][4 * j], dv[db][4 * j + 1], dv[db][4 * j + 2], dv[db][4 * j + 3]};
        }
    }
    __syncthreads();
    if (grp > 0) return;
#pragma unroll
    for (int g2 = 1; g2 < KS; ++g2) {
      const float* sg = (const float*)smem + (((g2 - 1) * 2 + w2) * 64 + lane) * 68;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f32x4 a = *(const f32x4*)(sg + db * 16 + j * 4), e = *(const f32x4*)(sg + 32 + db * 16 + j * 4);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            dk[db][4 * j + r] += a[r];
            dv[db][4 * j + r] += e[r];
          }
        }
    }
  }
  const int64_t row = (int64_t)b * p.Nkp + key;
  bf16_t* kp2 = (bf16_t*)p.dK + row * p.lddk + hc + 8 * hl;
  bf16_t* vp2 = (bf16_t*)p.dV + row * p.lddv + hc + 8 * hl;
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int jp = 0; jp < 2; ++jp) {
      uint2 wk[2], wv[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int j = 2 * jp + e;
        wk[e].x = pack2bf(dk[db][4 * j] * p.scale, dk[db][4 * j + 1] * p.scale);
        wk[e].y = pack2bf(dk[db][4 * j + 2] * p.scale, dk[db][4 * j + 3] * p.scale);
        wv[e].x = pack2bf(dv[db][4 * j], dv[db][4 * j + 1]);
        wv[e].y = pack2bf(dv[db][4 * j + 2], dv[db][4 * j + 3]);
      }
      store_pair(kp2 + 32 * db + 16 * jp, hl, wk[0], wk[1]);
      store_pair(vp2 + 32 * db + 16 * jp, hl, wv[0], wv[1]);
    }
}

// the dQ tiles and the dK / dV tiles of one layer in ONE launch (blockIdx.x < Nq / 64: dQ role); D = rowsum(dO * O) comes from attn_prep_kernel
template <int KS>
__global__ __launch_bounds__(128 * KS) void attn32_bwd_both_kernel(const sdlt_attn_params p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int ndq = p.Nq >> 6;
  const WgId wg = xcd_wg();
  if (wg.x < ndq) attn32_dq_body<KS>(p, smem, wg.x, wg);
  else attn32_dkdv_body<KS>(p, smem, wg.x - ndq, wg);
}

template <typename F>
int set_smem32(F f, int bytes) { return sdlt_raise_smem((const void*)f, bytes); }

}  // namespace

bool sdlt_attn32_ok(const sdlt_attn_params& p) {
  return p.d == 64 && !p.causal && p.Nq >= 64 && p.Nk >= 64 && (p.Nq % 64) == 0 && (p.Nk % 64) == 0 && p.Nqp >= p.Nq && p.Nkp >= p.Nk;
}

int sdlt_attn32_fwd(const sdlt_attn_params& p, int ks, hipStream_t s) {
  dim3 grid(p.Nq / 64, p.H, p.B);
#define FWD32(KS_)                                                                                           \
  do {                                                                                                       \
    const int sm = (KS_) * 4 * TILE;                                                                         \
    if (set_smem32(attn32_fwd_kernel<KS_>, sm)) SDLT_FAIL(SDLT_ERR_LAUNCH, "sdlt_attn32_fwd: LDS attribute"); \
    hipLaunchKernelGGL((attn32_fwd_kernel<KS_>), grid, dim3(128 * (KS_)), sm, s, p);                          \
  } while (0)
  if (ks >= 4) FWD32(4);
  else if (ks >= 2) FWD32(2);
  else FWD32(1);
  SDLT_CHECK_LAUNCH();
  return SDLT_OK;
}

int sdlt_attn32_bwd_both(const sdlt_attn_params& p, int ks, hipStream_t s) {
  dim3 grid(p.Nq / 64 + p.Nk / 64, p.H, p.B);
#define BWD32(KS_)                                                                                                 \
  do {                                                                                                             \
    const int sm = (KS_) * 2 * (2 * TILE + 512);                                                                   \
    if (set_smem32(attn32_bwd_both_kernel<KS_>, sm)) SDLT_FAIL(SDLT_ERR_LAUNCH, "sdlt_attn32_bwd: LDS attribute"); \
    hipLaunchKernelGGL((attn32_bwd_both_kernel<KS_>), grid, dim3(128 * (KS_)), sm, s, p);                           \
  } while (0)
  if (ks >= 4) BWD32(4);
  else if (ks >= 2) BWD32(2);
  else BWD32(1);
  SDLT_CHECK_LAUNCH();
  return SDLT_OK;
}
