// bf16 MFMA GEMM family for the UNet training step (gfx950).
//
//   C[M,N] = alpha * ( sum_seg  X_seg[M,K_seg] . W_seg[N,K_seg]^T  +  s * (X.Adown^T) . Bup^T )
//            + bias[n] + rowbias[m / rows_per_batch, n] + R[m,n]
//
// * every operand is K-contiguous ("NT" form).  Frozen weights are kept in HBM in BOTH orientations
//   (W [N,K] for forward, W^T [K,N] for dX) - 288 GB makes that free - so forward and backward are
//   the same kernel.
// * X rows can be gathered on the fly: MODE 1 turns the kernel into an implicit-GEMM 3x3 convolution
//   over an NHWC activation (K = 9*Cin, k = tap*Cin + ci), incl. stride 2, nearest-2x upsampled input,
//   tap flip (dX of a stride-1 conv) and the transposed stride-2 form (dX of a downsampling conv).
//   Out-of-image taps read a zero page.
// * rank-r LoRA is fused: the LoRA-down product T = X.Adown^T is accumulated by the same K loop on
//   16 extra MFMA columns, scaled, rounded to bf16, and applied with one 16x16x16 MFMA per tile pair
//   (the 16x16 accumulator layout IS the 16x16x16 B-operand layout, so no cross-lane movement).
//   Replaces peft's three launches per adapted layer (reference: trainer/optimizer.py:84-95).
// * tile BMxBNx64, 4 waves (2x2), mfma_f32_16x16x32_bf16, operands swapped (W is the MFMA "A" operand)
//   so each lane ends with 4 consecutive n of one row m -> 8-byte epilogue accesses.
// * global->LDS via global_load_lds (16 B/lane, no VGPR round trip), LDS image XOR-swizzled through
//   the per-lane SOURCE address (chunk ^= row&7) -> conflict-free ds_read_b128; double-buffered,
//   one barrier per K step, next tile's DMA in flight under the MFMAs.
#include <type_traits>
#include "common.h"
#include "../../include/sdlt_kernels.h"

namespace {

constexpr int BK = 64;         // bf16 elements per K step (128 B rows in LDS)
constexpr int ROW_BYTES = 128;

struct ConvGeom {
  int Hin, Win, Cin, Hout, Wout, stride, ups, flip, tr;
};

// MI x NI 16x16 fragments per wave; waves are laid out WM (m) x WN (n), WM = 2 by default: WN = 2 -> 256 threads, WN = 4 -> 512
// threads (two waves per SIMD: the second half of the waves computes first and issues its DMA afterwards, so one wave's DMA
// issue stalls overlap the other's MFMAs on every SIMD).  WM = 4, WN = 2, NI = 5 are the 160-column tiles (256x160, 128x160):
// every SDXL / SD1.5 width is a multiple of 320, so they cut 1024 x 10240, 1024 x 5120, 4096 x 2560, 16384 x 320 ... into exactly
// 256 workgroups of wave tiles wide enough (64x80 / 32x80) for the MFMAs not to starve on LDS reads.
template <int N_>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory"); }

// EPI: the fused epilogue (sdlt_gemm_params.epi_op) as a TEMPLATE parameter - with a run-time switch the GEGLU / activation code sat
// in every instantiation and the whole GEMM family ran ~6 % slower (code size; the step alternates between ~60 kernels).
// -DSDLT_GEMM_TRACE (tools/gemm_trace.py): thread 0 of workgroup 0 stamps clock64() at the phase boundaries; sdlt_gemm_trace_read copies them out
#ifdef SDLT_GEMM_TRACE
__device__ long long g_gemm_tr[16];
#define GTR(i_) do { if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) g_gemm_tr[i_] = clock64(); } while (0)
#else
#define GTR(i_) do {} while (0)
#endif

// LN: the LayerNorm in front of the product folded in (sdlt_gemm_params.ln_c1): raw rows in, W = W o gamma, row statistics from the X fragments of
// the same K walk (ln_frag_stats: packed bf16 dot products on the VALU, beside the MFMAs), C = rstd (acc - mean c1) + c2 applied to the
// accumulators before the LoRA-up.
template <int MI, int NI, int WN, int MODE, int R16, int NSTAGE, int KG = 0, int BT = 0, int WM = 2, int EPI = 0, int LN = 0>
__global__ __launch_bounds__(64 * WM * WN) void gemm_kernel(const sdlt_gemm_params p) {
  static_assert(!LN || (MODE == 0 && KG == 0 && BT == 0 && R16 <= 1), "folded LayerNorm: plain products, rank pad 16 at most");
  GTR(0);
  // batched launch: blockIdx.y picks the problem; its operand pointers replace the launch-wide ones (wave-uniform scalar loads)
  // (BT is a template switch so that ordinary launches do not pay the extra kernarg loads and selects in their prologue)
  const sdlt_gemm_batch_item* bi = BT ? p.batch + blockIdx.y : nullptr;
  const void* pX = (bi && bi->X) ? bi->X : p.X;
  const void* pW = (bi && bi->W) ? bi->W : p.W;
  const void* pAdown = (bi && bi->Adown) ? bi->Adown : p.Adown;
  const void* pBup = (bi && bi->Bup) ? bi->Bup : p.Bup;
  void* pTout = (bi && bi->T_out) ? bi->T_out : p.T_out;
  void* pC = (bi && bi->C) ? bi->C : p.C;
  void* pCt = (bi && bi->Ct) ? bi->Ct : p.Ct;
  const float* pBias = (bi && bi->bias) ? bi->bias : p.bias;
  const float* pCs = R16 ? ((bi && bi->col_scale) ? bi->col_scale : p.col_scale) : nullptr;     // DoRA: adapter launches only
  constexpr int NW = WM * WN, NTHR = NW * 64;
  constexpr int BM = WM * MI * 16, BN = WN * NI * 16;
  static_assert(WM == 2 || WM == 4, "wave rows");
  constexpr int XT = BM * ROW_BYTES, WT = BN * ROW_BYTES, AT = (R16 ? R16 * 16 : 0) * ROW_BYTES;
  constexpr int STAGE = XT + WT + AT;
  constexpr int S = NSTAGE;                        // LDS ring depth: S-1 K-steps of DMA in flight under the MFMAs
  // KG > 0 (K-grouped adapters, the dX of stacked projections): K is G <= KG groups of lora_group_k columns, each with its own
  // rank-16 LoRA-down result; the accumulator is flushed to its 16 columns of Tsh at every group boundary.
  constexpr int TW = (KG ? KG * R16 : R16) * 16;   // T columns held in Tsh
  constexpr int TROW = R16 ? (TW + 4) : 4;         // bf16 elements per Tsh row (+4 pad)
  static_assert(KG == 0 || R16 >= 1, "K-grouped adapters need an adapter");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* tsh = smem + (S == 1 ? 2 : S) * STAGE;
  float* lnsh = (float*)(tsh + (R16 ? BM * TROW * 2 : 0));     // LN: (mean, rstd) of the tile's rows

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform by construction: keep it in an SGPR
  const int wm = wave & (WM - 1), wn = wave / WM;
  const int frow = lane & 15, fk = lane >> 4;
  // XCD-aware remap: consecutive tile ids (sharing an X panel) land on the same XCD/L2.
  const int nbm = (p.M + BM - 1) / BM, nbn = (p.N + BN - 1) / BN;
  int bid = blockIdx.x;
  {
    const int nwg = nbm * nbn * (p.splitk > 1 ? p.splitk : 1), q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int splitk = p.splitk > 1 ? p.splitk : 1;
  // (integer divisions are ~40 scalar instructions each on this ISA: the common no-split case skips them)
  const int tile_id = splitk == 1 ? bid : div_small_u(bid, splitk), split = splitk == 1 ? 0 : bid - tile_id * splitk;   // splits of one tile are neighbours -> same XCD
  // Grouped rasterisation inside each XCD's contiguous chunk of tiles: walk GROUP_M row-tiles before moving to the next
  // column-tile, so the ~32 workgroups an XCD runs concurrently form a compact super-tile (they share X panels AND W
  // panels in that XCD's 4 MB L2 instead of streaming every W panel once per row of tiles).
  int bm, bn;
  {
    // GROUP_M ~ sqrt(tiles one XCD holds at a time * BN / BM): the footprint of a g_m x g_n super-tile is g_m X panels +
    // g_n W panels.  A fixed 8 made every XCD stream ALL of X when M is 8 row-tiles (1024 x 1280 x 5120 split 3: 97 MB of L2
    // misses for 23.6 MB of operands).
    const int per_xcd = (nbm * nbn + 7) >> 3;
    int live = splitk == 1 ? 32 : div_small_u(32, splitk);                          // tiles of the ~32 workgroups an XCD runs concurrently
    live = live < 1 ? 1 : live;
    live = per_xcd < live ? per_xcd : live;
    // (implicit-GEMM convolution: the nine taps of a row tile re-read the same activation rows, so an X panel costs the L2 a ninth of what its K
    //  length says while a W panel costs all of it - the super-tile that minimises fetched bytes is nine times taller)
#ifndef SDLT_CONV_GROUP_FACTOR
#define SDLT_CONV_GROUP_FACTOR 9
#endif
    int GROUP_M = (int)(sqrtf((float)live * BN / BM * (MODE == 1 ? SDLT_CONV_GROUP_FACTOR : 1)) + 0.5f);
    GROUP_M = GROUP_M < 1 ? 1 : (GROUP_M > nbm ? nbm : GROUP_M);
    const int per_group = GROUP_M * nbn;
    const int grp = div_small_u(tile_id, per_group), first_m = grp * GROUP_M;
    const int gsz = nbm - first_m < GROUP_M ? nbm - first_m : GROUP_M;
    const int in_grp = tile_id - grp * per_group;
    bn = div_small_u(in_grp, gsz);
    bm = first_m + in_grp - bn * gsz;
  }
  const int m0 = bm * BM, n0 = bn * BN;

  GTR(12);
  // ---------------- per-lane staging geometry (fixed rows, fixed swizzled chunk) ----------------
  const int srow = lane >> 3;                       // row within an 8-row DMA piece
  const int schunk = (lane & 7) ^ (srow & 7);       // source chunk so that LDS holds chunk^(row&7)
  // DMA pieces (8 rows x 128 B): X pieces split evenly over the waves; W pieces WF each plus one more for the first W_REM waves
  // (160 columns = 20 pieces over 8 waves), which then run with their own DMA count (see the counted waits)
  constexpr int XI = BM / 8 / NW, WF = BN / 8 / NW, W_REM = (BN / 8) % NW, WI = WF + (W_REM ? 1 : 0);
  static_assert(XI >= 1 && WI >= 1 && XI * 8 * NW == BM, "tile does not split evenly over the waves");
  const bool w_extra = W_REM && wave < W_REM;
  const bf16_t* xptr[XI];
  int xb[XI], xh[XI], xw[XI];                       // conv: decoded output pixel (b<0 => row invalid)
#pragma unroll
  for (int i = 0; i < XI; ++i) {
    int m = m0 + (wave + NW * i) * 8 + srow;
    if (MODE == 0) {
      int mc = m < p.M ? m : p.M - 1;
      xptr[i] = (const bf16_t*)pX + (size_t)mc * p.ldx + schunk * 8;
      xb[i] = xh[i] = xw[i] = 0;
    } else {
      if (m < p.M) {
        int hw = p.Hout * p.Wout;
        xb[i] = div_small(m, hw);
        int rem = m - xb[i] * hw;
        xh[i] = div_small(rem, p.Wout);
        xw[i] = rem - xh[i] * p.Wout;
      } else {
        xb[i] = -1; xh[i] = xw[i] = 0;
      }
      xptr[i] = nullptr;
    }
  }
  const bf16_t* wptr[WI];
#pragma unroll
  for (int i = 0; i < WI; ++i) {
    int n = n0 + (wave + NW * i) * 8 + srow;     // (i == WF: only used by the first W_REM waves)
    int nc = n < p.N ? n : p.N - 1;
    wptr[i] = (const bf16_t*)pW + (size_t)nc * p.ldw + schunk * 8;
  }
  const bf16_t* x2ptr[XI];
  const bf16_t* w2ptr[WI];
#pragma unroll
  for (int i = 0; i < XI; ++i) x2ptr[i] = xptr[i];
#pragma unroll
  for (int i = 0; i < WI; ++i) w2ptr[i] = wptr[i];
  if (p.K2 > 0) {
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      int m = m0 + (wave + NW * i) * 8 + srow;
      int mc = m < p.M ? m : p.M - 1;
      x2ptr[i] = (const bf16_t*)p.X2 + (size_t)mc * p.ldx2 + schunk * 8;
    }
#pragma unroll
    for (int i = 0; i < WI; ++i) {
      int n = n0 + (wave + NW * i) * 8 + srow;
      int nc = n < p.N ? n : p.N - 1;
      w2ptr[i] = (const bf16_t*)p.W2 + (size_t)nc * p.ldw2 + schunk * 8;
    }
  }
  // LoRA-down tile: R16*16 rows -> R16*2 pieces, taken by waves 0..(R16*2-1) round robin
  const bf16_t* aptr = nullptr;
  if (R16) {
    // piece index handled by this wave in round j: wave + 4*j  (< R16*2)
    aptr = (const bf16_t*)pAdown + schunk * 8;
  }
  // grouped adapters (fused projections): this tile's column group selects the Adown rows and the T_out columns
  const int lgrp = (R16 && p.lora_group_n > 0) ? div_small_u(n0, p.lora_group_n) : 0;
  const bool t_writer = R16 && (p.lora_group_n > 0 ? n0 == lgrp * p.lora_group_n : bn == 0);
  if (R16) aptr += (size_t)lgrp * (R16 * 16) * p.ld_adown;

  const int nk1 = p.K / BK, nk2 = p.K2 / BK, nk = nk1 + nk2;
  // LoRA-down tile: 2*R16 pieces per stage.  Fewer pieces than waves (rank pad 16 in an 8-wave tile: 2 pieces): only the first
  // 2*R16 waves move one each and run with their own DMA count - the counted vmcnt below is an immediate, so the two kinds
  // of waves take different (wave-uniform) branches.  (Every wave re-loading a piece, as before, added 8 KB to the 24 KB of
  // a 64x128 stage on the texture-address path that bounds the K loop.)
  constexpr bool A_FEW = R16 && 2 * R16 < NW;
  constexpr int AI = R16 ? (A_FEW ? 1 : (2 * R16 + NW - 1) / NW) : 0;   // LoRA-down DMA instructions per (loading) wave and stage
  constexpr int LPS = XI + WI + AI;                         // most DMA instructions any wave issues per stage (S == 1 register path)
  constexpr int LPSB = XI + WF + (A_FEW ? 0 : AI);          // DMA instructions per stage every wave issues ...
  const bool a_loader = !A_FEW || wave < 2 * R16;
  const int lps_extra = (w_extra ? 1 : 0) + ((A_FEW && a_loader) ? 1 : 0);   // ... plus this wave's own extras (wave-uniform)
  // "at most MULT stages of this wave's DMA still in flight"
  auto wait_stages = [&](auto mult) {
    constexpr int MULT = decltype(mult)::value;
    if (lps_extra == 0) wait_vmcnt<MULT * LPSB>();
    else if (lps_extra == 1) wait_vmcnt<MULT * (LPSB + 1)>();
    else wait_vmcnt<MULT * (LPSB + 2)>();
  };

  // this workgroup's share of the K steps (split-K: contiguous ranges of the combined segment-1 + segment-2 steps)
  const int kbeg = splitk == 1 ? 0 : div_small_u(nk * split, splitk), kend = splitk == 1 ? nk : div_small_u(nk * (split + 1), splitk);   // 32-bit: nk < 2^15

  // implicit-GEMM conv: tap / channel offset of the NEXT stage to be issued (stages are issued strictly in K order from kbeg)
  int cv_tap = 0, cv_ci0 = 0;
  if (MODE == 1) {
    cv_tap = kbeg == 0 ? 0 : div_small_u(kbeg * BK, p.Cin);
    cv_ci0 = kbeg * BK - cv_tap * p.Cin;
  }
  // Enumerates the 8-row x 128-byte pieces this wave moves for K-step kt: f(j, src, lds_off) with j in [0, LPS).
  // NOTE: the segment-1 / segment-2 operand pointers are picked per element with value selects.  Handing the two pointer
  // ARRAYS to a common tail (if/else around the loops) made hipcc keep them in scratch and index them at run time: a
  // scratch_load + s_waitcnt vmcnt(0) in front of the W pieces of EVERY stage, i.e. the whole DMA ring drained once per
  // K-step in every kernel without LoRA.
  auto for_each_piece = [&](int kt, auto&& f) {
    constexpr bool TWOSEG = MODE == 0 && R16 == 0;     // only plain GEMMs may carry a second K segment (host-checked)
    const bool seg1 = !TWOSEG || kt < nk1;
    const int k0 = (seg1 ? kt : kt - nk1) * BK;
    if (MODE == 1) {
        // stages are issued in K order, so the (tap, channel offset) of the step is tracked incrementally (no division)
        const int tap = cv_tap, ci0 = cv_ci0;
        cv_ci0 += BK;
        if (cv_ci0 >= p.Cin) { cv_ci0 = 0; ++cv_tap; }
        const int dy = tap / 3, dx = tap - dy * 3;
        const int oy = p.flip ? 1 - dy : dy - 1, ox = p.flip ? 1 - dx : dx - 1;
#pragma unroll
        for (int i = 0; i < XI; ++i) {
          const bf16_t* src = (const bf16_t*)p.zero + schunk * 8;
          if (xb[i] >= 0) {
            int hi, wi;
            bool ok;
            if (p.tr) {
              int ny = xh[i] + oy, nx = xw[i] + ox;
              ok = ny >= 0 && nx >= 0 && !(ny & 1) && !(nx & 1);
              hi = ny >> 1; wi = nx >> 1;
              ok = ok && hi < p.Hin && wi < p.Win;
            } else {
              hi = xh[i] * p.stride + oy; wi = xw[i] * p.stride + ox;
              ok = hi >= 0 && wi >= 0 && hi < p.Hin * p.ups && wi < p.Win * p.ups;
              if (p.ups == 2) { hi >>= 1; wi >>= 1; }
            }
            if (ok) src = (const bf16_t*)pX + ((size_t)(xb[i] * p.Hin + hi) * p.Win + wi) * p.ldx + ci0 + schunk * 8;
          }
          f(i, src, (wave + NW * i) * 1024);
        }
    } else {
#pragma unroll
      for (int i = 0; i < XI; ++i) {
        const bf16_t* src = (!TWOSEG || seg1) ? xptr[i] : x2ptr[i];
        f(i, src + k0, (wave + NW * i) * 1024);
      }
    }
#pragma unroll
    for (int i = 0; i < WF; ++i) {
      const bf16_t* src = (!TWOSEG || seg1) ? wptr[i] : w2ptr[i];
      f(XI + i, src + k0, XT + (wave + NW * i) * 1024);
    }
    if constexpr (W_REM > 0) {
      if (w_extra) {
        const bf16_t* src = (!TWOSEG || seg1) ? wptr[WF] : w2ptr[WF];
        f(XI + WF, src + k0, XT + (wave + NW * WF) * 1024);
      }
    }
    if (R16) {
      // (LoRA excludes a second segment.)  Every wave moves the SAME number of pieces per stage (the counted vmcnt of the
      // DMA path relies on it); when there are fewer pieces than waves some waves re-load a piece - identical bytes to
      // identical LDS addresses.
      if constexpr (A_FEW) {
        if (a_loader) f(XI + WI, aptr + (size_t)(wave * 8 + srow) * p.ld_adown + k0, XT + WT + wave * 1024);
      } else {
#pragma unroll
        for (int j = 0; j < AI; ++j) {
          int piece = (wave + NW * j) % (R16 * 2);
          f(XI + WI + j, aptr + (size_t)(piece * 8 + srow) * p.ld_adown + k0, XT + WT + piece * 1024);
        }
      }
    }
  };
  // LDS-DMA path: global -> LDS directly (16 B per lane, destination = piece base + lane*16)
  auto stage = [&](int kt, int buf) {
#ifndef SDLT_LAB_NO_DMA
    char* base = smem + buf * STAGE;
    for_each_piece(kt, [&](int, const bf16_t* src, int off) { glds16(src, base + off); });
#endif
  };

  GTR(13);
  // the first K step's DMA goes out as soon as its addresses exist (clock stamps, tools/gemm_trace.py: the ~600 instructions of index
  // math, staging geometry and epilogue prefetch in front of it were 1.5 us of every launch with no load in flight)
  if (S > 1 && kbeg < kend) stage(kbeg, 0);
  // ---------------- epilogue operands, fetched BEFORE the K loop ----------------
  // LoRA-up fragments, bias and (staged epilogue) the residual tile used to be loaded where they are consumed: three dependent
  // global-load latencies after the last MFMA of every launch (~1 us each; the K loop of a 1024 x 1280 x 1280 projection is
  // ~7 us).  They are issued here instead - right behind the first DMA stage and older than every other one, so the counted vmcnt
  // waits of the ring still hold (loads retire in order: "stage kt landed" only ever waits for MORE than it names) - and are long
  // complete when the loop ends.
  constexpr int NUPMAX = KG ? KG * R16 : (R16 ? R16 : 1);
  s16x4 bupf[NUPMAX][NI];
  if (R16) {
    const int nup_ = KG ? div_small_u(p.K, p.lora_group_k) * R16 : R16;
#pragma unroll
    for (int j = 0; j < NUPMAX; ++j)
#pragma unroll
      for (int a = 0; a < NI; ++a) {
        const int n = n0 + wn * NI * 16 + a * 16 + frow;
        const int nc = n < p.N ? n : p.N - 1;
        // (UNCONDITIONAL loads from clamped addresses, here and below - round 5: a load under a condition, merged with a zero default, made hipcc copy the
        // loaded registers at the join and put an s_waitcnt vmcnt(0) behind it: three to five serial global round trips - the first DMA stage included - in the
        // prologue of every launch, seen in the ISA.  What an absent operand would have been is decided where it is USED.)
        bupf[j][a] = *(const s16x4*)((const bf16_t*)pBup + (size_t)nc * p.ld_bup + (j < nup_ ? j : 0) * 16 + fk * 4);
      }
  }
  // bias / c1: four columns per lane.  N % 4 == 0 (every layer of the UNet and the text encoders): one 16-byte load per block, columns beyond N read chunk 0 and
  // are never stored; any other N (wave-uniform branch): four clamped scalar loads
  f32x4 biasf[NI];
  {
    const float* bsrc = pBias ? pBias : (const float*)pW;         // (no bias: any readable N floats; the epilogue skips the addition)
    if ((p.N & 3) == 0) {
#pragma unroll
      for (int a = 0; a < NI; ++a) {
        const int n = n0 + wn * NI * 16 + a * 16 + fk * 4;
        biasf[a] = *(const f32x4*)(bsrc + (n < p.N ? n : 0));
      }
    } else {
#pragma unroll
      for (int a = 0; a < NI; ++a) {
        const int n = n0 + wn * NI * 16 + a * 16 + fk * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) biasf[a][r] = bsrc[n + r < p.N ? n + r : p.N - 1];
      }
    }
  }
  f32x4 c1f[LN ? NI : 1], lnca = (f32x4){0.f, 0.f, 0.f, 0.f}, lnab = lnca;
  if constexpr (LN) {          // (host-checked: N % 4 == 0 with a folded LayerNorm)
#pragma unroll
    for (int a = 0; a < NI; ++a) {
      const int n = n0 + wn * NI * 16 + a * 16 + fk * 4;
      c1f[a] = *(const f32x4*)(p.ln_c1 + (n < p.N ? n : 0));
    }
    if constexpr (R16 != 0) {      // adapter constants of this lane's four rank rows (rank = fk * 4 + i) of this tile's adapter group
      lnca = *(const f32x4*)(p.ln_adapter + lgrp * 32 + fk * 4);
      lnab = *(const f32x4*)(p.ln_adapter + lgrp * 32 + 16 + fk * 4);
    }
  }
  // LN == 2: the producer of the rows left (sum x, sum x^2) per row and column tile (sdlt_wsk_gemm_parts): thread t < BM fetches row t's partials
  // here (old loads: the counted waits of the ring only ever wait longer for them) and adds them after the K loop
  f32x4 lnp[LN == 2 ? 8 : 1];
  if constexpr (LN == 2) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int tr = tid < BM ? tid : 0;
      const int m = m0 + tr < p.M ? m0 + tr : p.M - 1;
      lnp[i] = *(const f32x4*)((const float*)p.ln_parts + ((size_t)m * p.ln_nparts + (2 * i < p.ln_nparts ? 2 * i : 0)) * 2);    // (absent pairs: pair 0 again, skipped below)
    }
  }
  // staged (bf16, full-row-segment) epilogue: geometry and the residual tile in the store loop's own layout
  constexpr int CST = BN + 4;                                    // fp32 elements per staged row (+4: bank spread)
  constexpr int REGION = (S == 1 ? 2 : S) * STAGE;
  constexpr int CROWS = BM * CST * 4 <= REGION ? BM : (BM / 2 * CST * 4 <= REGION ? BM / 2 : BM / 4);
  static_assert(CROWS * CST * 4 <= REGION && CROWS % 16 == 0, "staged epilogue does not fit the staging LDS");
  constexpr int NCH = BN / 8;                                    // 16-byte output chunks per row
  constexpr int QPP = CROWS * NCH / NTHR;                        // store-loop iterations per thread and pass
  constexpr int RIT = (BM / CROWS) * QPP;
  constexpr bool RPRE = RIT <= 4 && QPP * NTHR == CROWS * NCH;   // residual prefetch: at most 4 x 16 B per lane
  const bool vec_ok = ((p.ldc & 3) == 0) && (p.R == nullptr || (p.ldr & 3) == 0) &&
                      (p.rowbias == nullptr || (p.ld_rowbias & 3) == 0);
  const bool staged = EPI != 0 ||      // (the fused GEGLU epilogues live in the staged store loop; the host checked their alignment)
                      ((long)p.M * p.N >= (1l << 20) && !p.out_fp32 && pCt == nullptr && vec_ok && (p.ldc & 7) == 0 && (p.N & 7) == 0 &&
                       (p.R == nullptr || (p.ldr & 7) == 0) && (((uintptr_t)pC | (uintptr_t)p.R) & 15) == 0);
  uint4 rpre[RPRE ? RIT : 1];
  const bool rpre_on = RPRE && staged && p.R != nullptr && splitk == 1;   // (split-K: only the last arriver of a tile would use it)
  if (rpre_on) {
#pragma unroll
    for (int t = 0; t < RIT; ++t) {
      const int it = tid + (t % QPP) * NTHR, row = it / NCH, ch = it - row * NCH;
      const int m = m0 + (t / QPP) * CROWS + row, n = n0 + ch * 8;
      rpre[t] = *(const uint4*)((const bf16_t*)p.R + (size_t)(m < p.M ? m : p.M - 1) * p.ldr + (n < p.N ? n : 0));      // (rows / chunks outside the matrix are never stored)
    }
  }

  GTR(14);
  // ---------------- accumulators ----------------
  f32x4 acc[NI][MI];
#pragma unroll
  for (int a = 0; a < NI; ++a)
#pragma unroll
    for (int b = 0; b < MI; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  constexpr int TMI = MI >= WN ? MI / WN : 1;  // T fragments per wave (the WN waves of a wave-row split its MI fragments)
  const bool t_active = wn * TMI < MI;
  f32x4 tacc[R16 ? R16 : 1][TMI];
#pragma unroll
  for (int a = 0; a < (R16 ? R16 : 1); ++a)
#pragma unroll
    for (int b = 0; b < TMI; ++b) tacc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // LN: sum x / sum x^2 of this lane's 8-column chunks.  With an adapter the wave that owns a row block's LoRA-down fragments (xt2) owns its
  // statistics too and shares them through LDS; without one every wave keeps the statistics of its own MI row blocks (no LDS, no barrier)
  constexpr int LNB = LN ? (R16 ? TMI : MI) : 1;
  float ls1[LNB], ls2[LNB];
#pragma unroll
  for (int b = 0; b < LNB; ++b) ls1[b] = ls2[b] = 0.f;

  // byte offset of this lane's fragment chunk inside a 16-row block, for kk = 0/1
  const int foff0 = frow * ROW_BYTES + (((0 * 4 + fk) ^ (frow & 7)) << 4);
  const int foff1 = frow * ROW_BYTES + (((1 * 4 + fk) ^ (frow & 7)) << 4);

  auto compute = [&](const char* base, int kt) {
#ifdef SDLT_LAB_NO_COMPUTE
    return;
#endif
    const char* xs = base + (wm * MI * 16) * ROW_BYTES;
    const char* ws = base + XT + (wn * NI * 16) * ROW_BYTES;
    const char* as = base + XT + WT;
    const bool lora_step = R16 && kt < nk1;
    // every LDS read of the K-step first (one LDS round trip per step instead of four serial ones), then the MFMAs
    bf16x8 xf2[2][MI], wf2[2][NI], af2[2][R16 ? R16 : 1], xt2[2][TMI];
    const int tb = t_active ? wn * TMI : 0;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int fo = kk ? foff1 : foff0;
#pragma unroll
      for (int b = 0; b < MI; ++b) xf2[kk][b] = *(const bf16x8*)(xs + b * 16 * ROW_BYTES + fo);
#pragma unroll
      for (int a = 0; a < NI; ++a) wf2[kk][a] = *(const bf16x8*)(ws + a * 16 * ROW_BYTES + fo);
      if (R16) {
#pragma unroll
        for (int j = 0; j < R16; ++j) af2[kk][j] = *(const bf16x8*)(as + j * 16 * ROW_BYTES + fo);
      }
      if (R16) {
#pragma unroll
        for (int b = 0; b < TMI; ++b) xt2[kk][b] = *(const bf16x8*)(xs + (tb + b) * 16 * ROW_BYTES + fo);
      }
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
      for (int a = 0; a < NI; ++a)
#pragma unroll
        for (int b = 0; b < MI; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf2[kk][a], xf2[kk][b], acc[a][b], 0, 0, 0);
      if (R16) {
        if (lora_step && t_active) {
#pragma unroll
          for (int j = 0; j < R16; ++j)
#pragma unroll
            for (int b = 0; b < TMI; ++b)
              tacc[j][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af2[kk][j], xt2[kk][b], tacc[j][b], 0, 0, 0);
        }
      }
#ifndef SDLT_LAB_LN_NOSTATS
      if constexpr (LN == 1 && R16 != 0) {
        if (t_active) {
#pragma unroll
          for (int b = 0; b < TMI; ++b) ln_frag_stats(xt2[kk][b], ls1[b], ls2[b]);
        }
      }
      if constexpr (LN == 1 && R16 == 0) {
#pragma unroll
        for (int b = 0; b < MI; ++b) ln_frag_stats(xf2[kk][b], ls1[b], ls2[b]);
      }
#endif
    }
  };

  if (S == 1) {
    // ---- register-staged pipeline: global -> VGPR two K-steps ahead, VGPR -> LDS one step ahead, 2 LDS buffers ----
    uint4 ra[LPS], rb[LPS];
    const int loff = lane * 16;
    auto gload = [&](int kt, uint4 (&r)[LPS]) {
      for_each_piece(kt, [&](int j, const bf16_t* src, int) { r[j] = *(const uint4*)src; });
    };
    auto sstore = [&](int kt, const uint4 (&r)[LPS], char* base) {
      for_each_piece(kt, [&](int j, const bf16_t*, int off) { *(uint4*)(base + off + loff) = r[j]; });
    };
    char* buf0 = smem;
    char* buf1 = smem + STAGE;
    if (kbeg < kend) gload(kbeg, ra);
    if (kbeg + 1 < kend) gload(kbeg + 1, rb);
    if (kbeg < kend) sstore(kbeg, ra, buf0);
    __syncthreads();
    for (int kt = kbeg; kt < kend; kt += 2) {
      if (kt + 2 < kend) gload(kt + 2, ra);
      compute(buf0, kt);
      if (kt + 1 < kend) sstore(kt + 1, rb, buf1);
      __syncthreads();
      if (kt + 1 >= kend) break;
      if (kt + 3 < kend) gload(kt + 3, rb);
      compute(buf1, kt + 1);
      if (kt + 2 < kend) sstore(kt + 2, ra, buf0);
      __syncthreads();
    }
  } else {
  // prologue: S-1 stages in flight
  GTR(1);
  // (Round 5, measured and dropped: touching this tile's weight rows behind the ring's reach at launch - the trick that takes a quarter off the text encoders' long strips -
  // makes the whole step 1.0 ms SLOWER here (SD1.5 +0.35, full fine-tune +4.8): like the wave-split-K products these launches are bandwidth-bound, not latency-bound.)
#pragma unroll
  for (int t = 1; t < S - 1; ++t)          // (stage kbeg went out in front of the epilogue prefetch)
    if (kbeg + t < kend) stage(kbeg + t, t);
  GTR(2);
  const bool early = NW <= 4 || wave < NW / 2;
  auto kg_flush = [&](int kt) {
    if constexpr (KG > 0) {
      const int gsteps = p.lora_group_k / BK;
      const int grp = div_small_u(kt, gsteps);
      if (splitk == 1 && kt + 1 == (grp + 1) * gsteps) {   // last K-step of adapter group g (under split-K a split IS a group, see the
                                                     // reduction): park s*T_g in its Tsh columns and restart the accumulator
        if (t_active) {
#pragma unroll
          for (int j = 0; j < R16; ++j)
#pragma unroll
          for (int b = 0; b < TMI; ++b) {
            int ml = wm * MI * 16 + (wn * TMI + b) * 16 + frow;
            uint2 v;
            v.x = pack2bf(tacc[j][b][0] * p.lora_scale, tacc[j][b][1] * p.lora_scale);
            v.y = pack2bf(tacc[j][b][2] * p.lora_scale, tacc[j][b][3] * p.lora_scale);
            *(uint2*)(tsh + ((size_t)ml * TROW + (grp * R16 + j) * 16 + fk * 4) * 2) = v;
            tacc[j][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
          }
        }
      }
    }
  };
  // Steady state (a stage is still to be issued): no data-dependent control flow in the loop - the wave has issued stages up
  // to kt+S-2, so "stage kt landed" is always vmcnt((S-2)*LPS); ring slots advance by increment-and-wrap.  The scalar
  // bookkeeping of the general form (min/sub/compare per wait, signed modulo per slot) was ~70 SALU instructions and six
  // branches per K-step.
  int kt = kbeg;
  int rd = 0, wr = S - 1;                           // ring slot being consumed / being filled
  for (; kt + S - 1 < kend; ++kt) {
    wait_stages(std::integral_constant<int, S - 2>{});
    __builtin_amdgcn_s_barrier();   // stage kt visible to all waves; everyone is done reading stage kt-1's buffer
    asm volatile("" ::: "memory");
    if (kt - kbeg < 4) GTR(3 + kt - kbeg);
    if (early) stage(kt + S - 1, wr);
    compute(smem + rd * STAGE, kt);
    if (!early) stage(kt + S - 1, wr);
    kg_flush(kt);
    rd = rd + 1 == S ? 0 : rd + 1;
    wr = wr + 1 == S ? 0 : wr + 1;
  }
  // Drain: the last (up to) S-1 stages are in flight, nothing left to issue.
  for (; kt < kend; ++kt) {
    const int ahead = kend - 1 - kt;                // stages issued after stage kt
    if (S >= 4 && ahead >= 2) wait_stages(std::integral_constant<int, 2>{});
    else if (S >= 3 && ahead >= 1) wait_stages(std::integral_constant<int, 1>{});
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    compute(smem + rd * STAGE, kt);
    kg_flush(kt);
    rd = rd + 1 == S ? 0 : rd + 1;
  }
  }

  GTR(8);
  // ---------------- folded LayerNorm: (mean, rstd) of the row blocks this wave owns ----------------
  float ln_mean[LNB], ln_rstd[LNB];
  if constexpr (LN == 2) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tid < BM) {
      // partial t = (sum, centred sum of squares) of K / ln_nparts columns: total M2 = sum_t M2_t + n_t (mean_t - mean)^2 (no large cancellation)
      float s1 = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (2 * i < p.ln_nparts) s1 += lnp[i][0] + lnp[i][2];            // (ln_nparts even)
      const float inv = 1.f / (float)p.K, nt = (float)p.K / (float)p.ln_nparts, invt = 1.f / nt;
      const float mean = s1 * inv;
      float m2 = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (2 * i < p.ln_nparts) {
          const float d0 = lnp[i][0] * invt - mean, d1 = lnp[i][2] * invt - mean;
          m2 += lnp[i][1] + lnp[i][3] + nt * (d0 * d0 + d1 * d1);
        }
      }
      const float rstd = rsqrtf(m2 * inv + p.ln_eps);
      *(float2*)(lnsh + 2 * tid) = make_float2(mean, rstd);
      if (p.ln_stats && bn == 0 && m0 + tid < p.M) *(float2*)(p.ln_stats + (size_t)(m0 + tid) * 2) = make_float2(mean, rstd);
    }
    __syncthreads();
    if constexpr (R16 != 0) {
      if (t_active) {
#pragma unroll
        for (int b = 0; b < TMI; ++b) {
          const float2 st = *(const float2*)(lnsh + 2 * (wm * MI * 16 + (wn * TMI + b) * 16 + frow));
          ln_mean[b] = st.x; ln_rstd[b] = st.y;
        }
      }
    } else {
#pragma unroll
      for (int b = 0; b < MI; ++b) {
        const float2 st = *(const float2*)(lnsh + 2 * (wm * MI * 16 + b * 16 + frow));
        ln_mean[b] = st.x; ln_rstd[b] = st.y;
      }
    }
  }
  if constexpr (LN == 1) {
    if (R16 == 0 || t_active) {
#pragma unroll
      for (int b = 0; b < LNB; ++b) {
        const float inv = 1.f / (float)p.K;
        const float mean = ln_sum_fk(ls1[b]) * inv, var = ln_sum_fk(ls2[b]) * inv - mean * mean;
        const float rstd = rsqrtf((var > 0.f ? var : 0.f) + p.ln_eps);
        ln_mean[b] = mean; ln_rstd[b] = rstd;
        if (fk == 0 && (R16 != 0 || wn == 0)) {
          const int ml = wm * MI * 16 + ((R16 ? wn * TMI : 0) + b) * 16 + frow;
          if constexpr (R16 != 0) *(float2*)(lnsh + 2 * ml) = make_float2(mean, rstd);
          if (p.ln_stats && bn == 0 && m0 + ml < p.M) *(float2*)(p.ln_stats + (size_t)(m0 + ml) * 2) = make_float2(mean, rstd);
        }
      }
    }
  }
  // ---------------- split-K: publish the partial tile, last arriver reduces (agent-scope release/acquire) ----------------
  if (splitk > 1) {
    constexpr int NACC = NI * MI, NT = R16 ? R16 * TMI : 0;
    constexpr size_t SLAB = (size_t)(NACC + NT) * NTHR;    // float4 per (tile, split)
    f32x4* slab = (f32x4*)p.ws_slab + ((size_t)tile_id * splitk + split) * SLAB;
#pragma unroll
    for (int a = 0; a < NI; ++a)
#pragma unroll
      for (int b = 0; b < MI; ++b) slab[(a * MI + b) * NTHR + tid] = acc[a][b];
    if (R16) {
#pragma unroll
      for (int j = 0; j < R16; ++j)
#pragma unroll
        for (int b = 0; b < TMI; ++b) slab[(NACC + j * TMI + b) * NTHR + tid] = tacc[j][b];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                       // also: every wave is done with the staging LDS, smem[0] can carry the ticket
    int* cnt = p.ws_cnt + tile_id;
    if (tid == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      *(volatile int*)smem = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const int ticket = *(volatile int*)smem;
    if (ticket != splitk - 1) return;      // not the last arriver of this tile
    if (tid == 0) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm for the next launch / graph replay
    }
    __syncthreads();
    // fixed summation order 0..splitk-1 (own partial re-read from its slab) -> bitwise reproducible results
#pragma unroll
    for (int a = 0; a < NI; ++a)
#pragma unroll
      for (int b = 0; b < MI; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (R16) {
#pragma unroll
      for (int j = 0; j < R16; ++j)
#pragma unroll
        for (int b = 0; b < TMI; ++b) tacc[j][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    for (int sp = 0; sp < splitk; ++sp) {
      const f32x4* o = (const f32x4*)p.ws_slab + ((size_t)tile_id * splitk + sp) * SLAB;
#pragma unroll
      for (int a = 0; a < NI; ++a)
#pragma unroll
        for (int b = 0; b < MI; ++b) acc[a][b] += o[(a * MI + b) * NTHR + tid];
      if (R16) {
        if constexpr (KG > 0) {
          // K-grouped adapters under split-K: split sp covered exactly adapter group sp, its T goes to its own Tsh columns
          if (t_active) {
#pragma unroll
            for (int j = 0; j < R16; ++j)
#pragma unroll
            for (int b = 0; b < TMI; ++b) {
              const f32x4 t = o[(NACC + j * TMI + b) * NTHR + tid];
              int ml = wm * MI * 16 + (wn * TMI + b) * 16 + frow;
              uint2 v;
              v.x = pack2bf(t[0] * p.lora_scale, t[1] * p.lora_scale);
              v.y = pack2bf(t[2] * p.lora_scale, t[3] * p.lora_scale);
              *(uint2*)(tsh + ((size_t)ml * TROW + (sp * R16 + j) * 16 + fk * 4) * 2) = v;
            }
          }
        } else {
#pragma unroll
        for (int j = 0; j < R16; ++j)
#pragma unroll
          for (int b = 0; b < TMI; ++b) tacc[j][b] += o[(NACC + j * TMI + b) * NTHR + tid];
        }
      }
    }
    __syncthreads();
  }

  GTR(9);
  // ---------------- fused LoRA-up ----------------
  if (R16) {
    // tacc[j][b][r] = T[m = wm*MI*16 + (wn*TMI+b)*16 + (lane&15)][rank = j*16 + (lane>>4)*4 + r]
    if constexpr (KG == 0) {
#pragma unroll
    for (int j = 0; j < R16; ++j)
#pragma unroll
      for (int b = 0; b < TMI; ++b) {
        if (!t_active) continue;
        int ml = wm * MI * 16 + (wn * TMI + b) * 16 + frow;
        if constexpr (LN != 0) {      // T = rstd (x (A o gamma)^T - mean cA) + A beta
#pragma unroll
          for (int i = 0; i < 4; ++i) tacc[j][b][i] = ln_rstd[b] * (tacc[j][b][i] - ln_mean[b] * lnca[i]) + lnab[i];
        }
        uint2 v;
        v.x = pack2bf(tacc[j][b][0] * p.lora_scale, tacc[j][b][1] * p.lora_scale);
        v.y = pack2bf(tacc[j][b][2] * p.lora_scale, tacc[j][b][3] * p.lora_scale);
        *(uint2*)(tsh + ((size_t)ml * TROW + j * 16 + fk * 4) * 2) = v;
      }
    }
    const int nup = KG ? div_small_u(p.K, p.lora_group_k) * R16 : R16;   // 16-column blocks of T (K-grouped: R16 per adapter)
    __syncthreads();
    if constexpr (LN != 0) {
#pragma unroll
      for (int b = 0; b < MI; ++b) {
        const float2 st = *(const float2*)(lnsh + 2 * (wm * MI * 16 + b * 16 + frow));
#pragma unroll
        for (int a = 0; a < NI; ++a) acc[a][b] = (acc[a][b] - c1f[a] * st.x) * st.y;
      }
    }
    if (pTout != nullptr && t_writer) {
      // [BM rows][R] bf16 -> global, 8 B per lane
      const int CH = nup * 4;  // 8-byte chunks per row
      for (int c = tid; c < BM * CH; c += NTHR) {
        int ml = c / CH, cc = c - ml * CH;
        int m = m0 + ml;
        if (m < p.M) *(uint2*)((bf16_t*)pTout + (size_t)m * p.ld_t + lgrp * (R16 * 16) + cc * 4) = *(const uint2*)(tsh + ((size_t)ml * TROW + cc * 4) * 2);
      }
    }
#pragma unroll
    for (int j = 0; j < (KG ? KG * R16 : R16); ++j) {
      if (j >= nup) break;
      s16x4 tf[MI];
#pragma unroll
      for (int b = 0; b < MI; ++b) {
        int ml = wm * MI * 16 + b * 16 + frow;
        tf[b] = *(const s16x4*)(tsh + ((size_t)ml * TROW + j * 16 + fk * 4) * 2);
      }
#pragma unroll
      for (int a = 0; a < NI; ++a) {
#pragma unroll
        for (int b = 0; b < MI; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(bupf[j][a], tf[b], acc[a][b], 0, 0, 0);
      }
    }
  }

  if constexpr (LN != 0 && R16 == 0) {
#pragma unroll
    for (int b = 0; b < MI; ++b) {
#pragma unroll
      for (int a = 0; a < NI; ++a) acc[a][b] = (acc[a][b] - c1f[a] * ln_mean[b]) * ln_rstd[b];
    }
  }
  GTR(10);
  // ---------------- epilogue ----------------
#ifdef SDLT_LAB_NO_EPILOGUE
  if (acc[0][0][0] == 12345.f) ((float*)pC)[0] = 1.f;
  return;
#endif
  // acc[a][b][r] = C[m = m0 + wm*MI*16 + b*16 + (lane&15)][n = n0 + wn*NI*16 + a*16 + (lane>>4)*4 + r]
#ifndef SDLT_LAB_DIRECT_EPILOGUE
  // ---- staged store (the common bf16 case): the fragment layout gives every lane 4 consecutive columns of one row, i.e.
  // a wave store would touch 16 rows x 32 B.  Instead the fp32 tile (alpha, bias, row bias applied) goes through the
  // now idle staging LDS and leaves as full 256-byte row segments, 16 B per lane; the residual is read the same way and
  // added before the single bf16 rounding.
  // (not for the M = 128 text-encoder GEMMs: a handful of tiles, where the two extra barriers cost more than the wider stores save)
  if (staged) {
    float* csh = (float*)smem;
    __syncthreads();                                               // every wave is past its last read of the staging LDS
#pragma unroll
    for (int pass = 0; pass < BM / CROWS; ++pass) {
#pragma unroll
      for (int b = 0; b < MI; ++b) {
        const int ml = wm * MI * 16 + b * 16 + frow;
        if ((wm * MI * 16 + b * 16) / CROWS != pass) continue;     // compile-time after unrolling except for wm
        const int m = m0 + ml;
        const int brow = (p.rowbias && m < p.M) ? m / p.rows_per_batch : 0;
#pragma unroll
        for (int a = 0; a < NI; ++a) {
          const int nl = wn * NI * 16 + a * 16 + fk * 4;
          const int n = n0 + nl;
          f32x4 v = acc[a][b] * p.alpha;
          if constexpr (R16 != 0) {
            if (pCs && n < p.N) v *= *(const f32x4*)(pCs + n);    // (staged: N % 8 == 0)
          }
          if (pBias) v += biasf[a];
          if (n < p.N) {
            if (p.rowbias) {
              uint2 rb = *(const uint2*)((const bf16_t*)p.rowbias + (size_t)brow * p.ld_rowbias + n);
              v[0] += bf2f(rb.x & 0xffff); v[1] += bf2f(rb.x >> 16); v[2] += bf2f(rb.y & 0xffff); v[3] += bf2f(rb.y >> 16);
            }
          }
          *(f32x4*)(csh + (ml - pass * CROWS) * CST + nl) = v;
        }
      }
      __syncthreads();
#pragma unroll
      for (int q = 0; q < (RPRE ? QPP : (CROWS * NCH + NTHR - 1) / NTHR); ++q) {
        const int it = tid + q * NTHR;
        if (!RPRE && it >= CROWS * NCH) break;
        const int row = it / NCH, ch = it - row * NCH;
        const int m = m0 + pass * CROWS + row, n = n0 + ch * 8;
        if (m >= p.M || n >= p.N) continue;
        const float* src = csh + row * CST + ch * 8;
        f32x4 lo = *(const f32x4*)src, hi = *(const f32x4*)(src + 4);
        if (p.R) {
          uint4 rv;
          bool have = false;
          if constexpr (RPRE) {
            if (rpre_on) { rv = rpre[pass * QPP + q]; have = true; }
          }
          if (!have) rv = *(const uint4*)((const bf16_t*)p.R + (size_t)m * p.ldr + n);
          lo[0] += bf2f(rv.x & 0xffff); lo[1] += bf2f(rv.x >> 16); lo[2] += bf2f(rv.y & 0xffff); lo[3] += bf2f(rv.y >> 16);
          hi[0] += bf2f(rv.z & 0xffff); hi[1] += bf2f(rv.z >> 16); hi[2] += bf2f(rv.w & 0xffff); hi[3] += bf2f(rv.w >> 16);
        }
        uint4 o;
        if constexpr (EPI == 2) {
          // dX of ff.net.2 fused with GEGLU's backward: lo/hi = dG[m, n..n+7] (hidden index n); hidden / gate of the forward from F1
          const size_t fo = (size_t)m * p.ld_epi_in + (n >> 4) * 32 + (n & 15);
          const uint4 hv = *(const uint4*)((const bf16_t*)p.epi_in + fo), gv = *(const uint4*)((const bf16_t*)p.epi_in + fo + 16);
          const uint32_t* hp = (const uint32_t*)&hv;
          const uint32_t* gp = (const uint32_t*)&gv;
          float dgl[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]}, dh[8], dgt[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float hh = bf2f((hp[j >> 1] >> ((j & 1) * 16)) & 0xffff), gg = bf2f((gp[j >> 1] >> ((j & 1) * 16)) & 0xffff);
            dh[j] = dgl[j] * gelu_f(gg);
            dgt[j] = dgl[j] * hh * dgelu_f(gg);
          }
          bf16_t* dst = (bf16_t*)p.epi_out + (size_t)m * p.ld_epi_out + (n >> 4) * 32 + (n & 15);
          o.x = pack2bf(dh[0], dh[1]); o.y = pack2bf(dh[2], dh[3]); o.z = pack2bf(dh[4], dh[5]); o.w = pack2bf(dh[6], dh[7]);
          *(uint4*)dst = o;
          o.x = pack2bf(dgt[0], dgt[1]); o.y = pack2bf(dgt[2], dgt[3]); o.z = pack2bf(dgt[4], dgt[5]); o.w = pack2bf(dgt[6], dgt[7]);
          *(uint4*)(dst + 16) = o;
          continue;
        }
        if constexpr (EPI == 4) {      // dX of the MLP's second projection times act'(pre-activation of the first)
          const uint4 pv = *(const uint4*)((const bf16_t*)p.epi_in + (size_t)m * p.ld_epi_in + n);
          const uint32_t* pp = (const uint32_t*)&pv;
          float v8[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float x = bf2f((pp[j >> 1] >> ((j & 1) * 16)) & 0xffff);
            float d;
            if (p.epi_act == 1) { const float sg = 1.f / (1.f + __expf(-1.702f * x)); d = sg + 1.702f * x * sg * (1.f - sg); }
            else d = dgelu_f(x);
            v8[j] *= d;
          }
          lo = (f32x4){v8[0], v8[1], v8[2], v8[3]};
          hi = (f32x4){v8[4], v8[5], v8[6], v8[7]};
        }
        o.x = pack2bf(lo[0], lo[1]); o.y = pack2bf(lo[2], lo[3]); o.z = pack2bf(hi[0], hi[1]); o.w = pack2bf(hi[2], hi[3]);
        *(uint4*)((bf16_t*)pC + (size_t)m * p.ldc + n) = o;
        if constexpr (EPI == 3) {      // activation side output (from the fp32 values, before the bf16 rounding of C)
          float v8[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
          for (int j = 0; j < 8; ++j) v8[j] = p.epi_act == 1 ? v8[j] / (1.f + __expf(-1.702f * v8[j])) : gelu_f(v8[j]);
          uint4 q;
          q.x = pack2bf(v8[0], v8[1]); q.y = pack2bf(v8[2], v8[3]); q.z = pack2bf(v8[4], v8[5]); q.w = pack2bf(v8[6], v8[7]);
          *(uint4*)((bf16_t*)p.epi_out + (size_t)m * p.ld_epi_out + n) = q;
        }
        if (EPI == 1 && ((n >> 4) & 1) == 0) {
          // ff.net.0.proj fused with GEGLU: this chunk holds 8 hidden columns, their gates sit 16 columns further in the same staged row
          const f32x4 glo = *(const f32x4*)(src + 16), ghi = *(const f32x4*)(src + 20);
          uint4 q;
          q.x = pack2bf(lo[0] * gelu_f(glo[0]), lo[1] * gelu_f(glo[1])); q.y = pack2bf(lo[2] * gelu_f(glo[2]), lo[3] * gelu_f(glo[3]));
          q.z = pack2bf(hi[0] * gelu_f(ghi[0]), hi[1] * gelu_f(ghi[1])); q.w = pack2bf(hi[2] * gelu_f(ghi[2]), hi[3] * gelu_f(ghi[3]));
          *(uint4*)((bf16_t*)p.epi_out + (size_t)m * p.ld_epi_out + (n >> 5) * 16 + (n & 15)) = q;
        }
      }
      if (pass + 1 < BM / CROWS) __syncthreads();
    }
    GTR(11);
    return;
  }
#endif
#pragma unroll
  for (int b = 0; b < MI; ++b) {
    const int m = m0 + wm * MI * 16 + b * 16 + frow;
    if (m >= p.M) continue;
    const int brow = p.rowbias ? m / p.rows_per_batch : 0;
#pragma unroll
    for (int a = 0; a < NI; ++a) {
      const int n = n0 + wn * NI * 16 + a * 16 + fk * 4;
      if (n >= p.N) continue;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = acc[a][b][r] * p.alpha;
      if constexpr (R16 != 0) {
        if (pCs) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (n + r < p.N) v[r] *= pCs[n + r];
        }
      }
      if (vec_ok && n + 3 < p.N) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += pBias ? biasf[a][r] : 0.f;
        if (p.rowbias) {
          uint2 rb = *(const uint2*)((const bf16_t*)p.rowbias + (size_t)brow * p.ld_rowbias + n);
          v[0] += bf2f(rb.x & 0xffff); v[1] += bf2f(rb.x >> 16); v[2] += bf2f(rb.y & 0xffff); v[3] += bf2f(rb.y >> 16);
        }
        if (p.R) {
          uint2 rv = *(const uint2*)((const bf16_t*)p.R + (size_t)m * p.ldr + n);
          v[0] += bf2f(rv.x & 0xffff); v[1] += bf2f(rv.x >> 16); v[2] += bf2f(rv.y & 0xffff); v[3] += bf2f(rv.y >> 16);
        }
        if (pCt) {
#pragma unroll
          for (int r = 0; r < 4; ++r) ((bf16_t*)pCt)[(size_t)(n + r) * p.ldct + m] = f2bf(v[r]);
        }
        if (p.out_fp32) {
          if (p.accumulate) {
            float4 old = *(const float4*)((const float*)pC + (size_t)m * p.ldc + n);
            v[0] += old.x; v[1] += old.y; v[2] += old.z; v[3] += old.w;
          }
          *(float4*)((float*)pC + (size_t)m * p.ldc + n) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
          uint2 o;
          o.x = pack2bf(v[0], v[1]); o.y = pack2bf(v[2], v[3]);
          *(uint2*)((bf16_t*)pC + (size_t)m * p.ldc + n) = o;
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (n + r >= p.N) break;
          float x = v[r] + (pBias ? biasf[a][r] : 0.f);
          if (p.rowbias) x += bf2f(((const bf16_t*)p.rowbias)[(size_t)brow * p.ld_rowbias + n + r]);
          if (p.R) x += bf2f(((const bf16_t*)p.R)[(size_t)m * p.ldr + n + r]);
          if (pCt) ((bf16_t*)pCt)[(size_t)(n + r) * p.ldct + m] = f2bf(x);
          if (p.out_fp32) ((float*)pC)[(size_t)m * p.ldc + n + r] = p.accumulate ? ((float*)pC)[(size_t)m * p.ldc + n + r] + x : x;
          else ((bf16_t*)pC)[(size_t)m * p.ldc + n + r] = f2bf(x);
        }
      }
    }
  }
  GTR(11);
}

template <int MI, int NI, int WN, int MODE, int R16, int NSREQ, int KG = 0, int BT = 0, int WM = 2, int EPI = 0, int LN = 0>
int launch(const sdlt_gemm_params& p, hipStream_t stream) {
  constexpr int NTHR = 64 * WM * WN;
  constexpr int BM = WM * MI * 16, BN = WN * NI * 16;
  constexpr int STAGE = (BM + BN + R16 * 16) * ROW_BYTES;
  constexpr int TSH = (R16 ? BM * ((KG ? KG * R16 : R16) * 16 + 4) * 2 : 0) + (LN ? BM * 8 : 0);
  // LDS ring depth: NSREQ == 2 keeps the footprint small (several workgroups per CU overlap each other);
  // otherwise as deep as 160 KB allows, up to 4.
  constexpr int NS = NSREQ <= 2 ? NSREQ : ((4 * STAGE + TSH <= 160 * 1024) ? 4 : ((3 * STAGE + TSH <= 160 * 1024) ? 3 : 2));
  constexpr int NBUF = NS == 1 ? 2 : NS;
  if constexpr (NBUF * STAGE + TSH > 160 * 1024) {
    SDLT_FAIL(SDLT_ERR_UNSUPPORTED, "sdlt_gemm_bf16: tile %dx%d with LoRA rank pad %d does not fit the 160 KB LDS", BM, BN, R16 * 16);
  } else {
  const int smem = NBUF * STAGE + TSH;
  if (sdlt_raise_smem((const void*)gemm_kernel<MI, NI, WN, MODE, R16, NS, KG, BT, WM, EPI, LN>, smem))
    SDLT_FAIL(SDLT_ERR_LAUNCH, "sdlt_gemm_bf16: cannot raise the dynamic LDS limit to %d bytes", smem);
  const int nbm = (p.M + BM - 1) / BM, nbn = (p.N + BN - 1) / BN;
  const int splitk = p.splitk > 1 ? p.splitk : 1;
  if (splitk > 1) {
    constexpr int TMI = MI >= WN ? MI / WN : 1;
    const size_t slab_bytes = (size_t)(NI * MI + (R16 ? R16 * TMI : 0)) * NTHR * 16;
    if (!p.ws_slab || !p.ws_cnt || (size_t)nbm * nbn * splitk * slab_bytes > (size_t)p.ws_slab_bytes || nbm * nbn > p.ws_cnt_len)
      SDLT_FAIL(SDLT_ERR_SHAPE, "sdlt_gemm_bf16: split-K workspace too small (%d tiles x %d splits x %zu B)", nbm * nbn, splitk, slab_bytes);
  }
  hipLaunchKernelGGL((gemm_kernel<MI, NI, WN, MODE, R16, NS, KG, BT, WM, EPI, LN>), dim3(nbm * nbn * splitk, BT ? p.n_batch : 1), dim3(NTHR), smem, stream, p);
  }
  return SDLT_OK;
}

// tile ids: 1 = 128x128 (8 waves), 2 = 64x128 (8 waves), 3 = 64x64 (4 waves), 4 = 256x128 (8 waves), 5 = 128x128 (4 waves),
//           6 = 256x256 (8 waves), 7 = 256x160 (8 waves, 4x2), 8 = 128x160 (8 waves, 4x2)
// (round 6, measured and dropped - tools/conv_tile_probe.py, profiles/r06_conv_tile_probe.txt: 128 x 80 tiles of 4 waves (4 x 1) that cut a 4096 x 640 convolution into exactly
//  256 workgroups WITHOUT a K split - no fp32 slabs, no last-arriver seam - run 47-76 % SLOWER than the 128 x 160 tiles with their 2-3 splits (64 x 64 x 640 -> 640: 88.0 vs
//  59.9 us); 128 x 160 unsplit: +27-48 %.  The slabs' traffic is not what bounds these launches, the 8-wave K loop's DMA / MFMA staggering is what makes them fast.)
void tile_dims(int tile, int& bm, int& bn) {
  switch (tile) {
    case 7: bm = 256; bn = 160; break;
    case 8: bm = 128; bn = 160; break;
    case 1: case 5: bm = 128; bn = 128; break;
    case 2: bm = 64; bn = 128; break;
    case 4: bm = 256; bn = 128; break;
    case 6: bm = 256; bn = 256; break;
    default: bm = 64; bn = 64; break;
  }
}

template <int MODE, int R16>
int dispatch_tile(const sdlt_gemm_params& pin, hipStream_t s) {
  sdlt_gemm_params p = pin;
  if (p.lora_group_k > 0) {
    // K-grouped adapters: either one workgroup walks the whole K range (T flushed per group) or exactly one split per group
    const int G = p.K / p.lora_group_k;
    const long t128 = (long)((p.M + 127) / 128) * ((p.N + 127) / 128);
    p.splitk = (pin.splitk == G || (pin.splitk == 0 && G > 1 && t128 * G <= 320 && p.ws_slab && p.ws_cnt)) ? G : 1;
  }
  if (p.batch) p.splitk = 1;            // batched launch: the problems fill the chip, the split-K scratch is per launch
  if (p.ln_c1) p.splitk = 1;            // folded LayerNorm: a row's statistics come out of one K walk
  const int ktot = p.K + p.K2;
  if (p.tile == 0) {
    // Shape heuristics from the tools/gemm_probe.py sweep on MI355X ("GEMM tile selection"):
    //  * plenty of 128x128 tiles: 8-wave 128x128; with >= 2 tiles per CU use the shallow ring so 2 workgroups share a CU
    //  * 160..511 tiles: short K -> 64x64 tiles, 2+ workgroups per CU; long K -> 128x128 with the deep ring
    //  * fewer: 64x128 (8 waves) when K is short, otherwise 128x128 + split-K
    const long t128 = (long)((p.M + 127) / 128) * ((p.N + 127) / 128);
    const int nk = ktot / BK;
    const bool ws = p.ws_slab && p.ws_cnt;
    // 160-column tiles (tools/tile160_probe.py on MI355X): every UNet width is a multiple of 320, so 256x160 / 128x160 tiles cut
    // the wide feed-forward products and the top-resolution convolutions into exactly 256 (or 512, 1024) workgroups with no
    // padded columns, and their 64x80 / 32x80 wave tiles read fewer LDS bytes per MFMA than the 32x32 ones of the 64x128 tile.
    //   1024 x 10240 x 1280: 48.5 -> 34.0 us (256x160);  1024 x 5120 x 1280: 24.4 -> 19.8 (128x160);  4096 x 2560 x 640: 25.7 -> 19.4;
    //   16384 x 320 conv K 2880: 56.1 -> 42.6, K 8640: 156.6 -> 118.6;  4096 x 640 conv: 63.0 -> 56.2 with 2 splits.
    // Not for the attention projections (1024 x 1280 x 1280: 64 tiles) nor the long-K / narrow-N products (split-K 128x128 is as good).
    if (R16 <= 1 && !p.batch && !p.lora_group_k && p.N % 160 == 0 && p.M > 128 && !p.throughput_hint) {
      const long t7 = (long)((p.M + 255) / 256) * (p.N / 160), t8 = (long)((p.M + 127) / 128) * (p.N / 160);
      if (MODE == 0 && R16 == 0 && ktot <= 2560) {
        if (t7 >= 256) p.tile = 7;
        else if (t8 >= 224) p.tile = 8;
      } else if (MODE == 0 && R16 == 1 && !p.lora_group_n && ktot <= 640 && t8 >= 224 && t8 <= 320) {
        p.tile = 8;
      } else if (MODE == 1 && (t8 == 256 || t8 == 128 || t8 == 64)) {
        const int sk = (int)(256 / t8);
        if (sk == 1 || (ws && !p.splitk && nk / sk >= 16)) { p.tile = 8; if (sk > 1) p.splitk = sk; }
      }
      if (p.tile && !p.splitk) p.splitk = 1;
    }
    if (p.tile) {}
    else if (p.M <= 64) p.tile = p.N > 64 ? 2 : 3;
    else if (p.N <= 64) p.tile = 3;
    else if (p.M <= 128) {
      // text encoders / text-conditioning projections (M = 128): a few 64x64 tiles, latency-bound; split K only while the
      // grid is far from one workgroup per CU (120+ tiles: none; 60+: 3; fewer: 4)
      p.tile = 3;
      if (!p.splitk) {
        const long tiles = (long)((p.M + 63) / 64) * ((p.N + 63) / 64);
        int sk = !ws ? 1 : (tiles >= 100 ? 1 : (tiles >= 60 ? (nk >= 24 ? 3 : 1) : 4));   // (60+ tiles and a short K, CLIP-L fc1 / qkv: 10.8 -> 6.9 us unsplit)
        while (sk > 1 && nk / sk < 4) --sk;
        p.splitk = sk;
      }
    }
    else if (p.throughput_hint && !p.batch && !p.lora_group_k && (t128 >= 320 || (MODE == 1 && t128 >= 160))) {
      // several jobs share the device (whole-step A/B with two concurrent SDXL jobs: 85.0 -> 82.8 ms per pair): the other job
      // fills the CUs a launch leaves idle, so the 256x128 tile (0.75 operand-path cycles per MFMA cycle instead of 1.0) wins
      // although it halves the workgroup count.  With ONE job the same rule loses 0.3 ms.
      p.tile = 4;
    }
    else if (!R16 && MODE == 0 && t128 >= 320 && t128 <= 512 && p.N >= 1024) p.tile = 4;    // 4096x1920x640: 17.6 vs 19.7 us
    else if (t128 >= 320) { p.tile = ktot <= 640 ? 1 : 2; if (!p.stages) p.stages = 2; }     // >= 2 workgroups per CU, shallow ring
    else if (t128 >= 160) {
      // measured in the whole step (not in the hot-loop probe, where the 64x64 tile wins): 160..319 tiles of a plain / LoRA GEMM
      // run best as 128x128 tiles with the deep ring (55.2 -> 54.5 ms per SDXL step); the short-K convs keep 64x64
      // (the other classes were re-checked the same way, whole-step A/B per class: every alternative tile / ring / split was slower)
      if (MODE == 0 && ktot <= 6144) { p.tile = 1; if (!p.stages) p.stages = 4; }
      else if (ktot <= 2560 || (MODE == 0 && ktot <= 6144)) { p.tile = 3; if (!p.stages) p.stages = 2; }
      else {
        // 4096 x 640 with a long K (3x3 convs of the 64x64 blocks): 160 tiles x 3 splits, two workgroups per CU
        p.tile = 1;
        if (!p.stages) p.stages = 2;
        if (!p.splitk && ws && nk / 3 >= 8) p.splitk = 3;
      }
    } else if (ktot <= 2560 && t128 > 32) { p.tile = 2; if (!p.splitk) p.splitk = 1; }
    else if (ktot <= 2560) p.tile = 2;
    else if (t128 <= 24 && !p.batch && !p.lora_group_k && !p.out_fp32) {     // (fp32 outputs: the weight-gradient products keep their rule)
      // a handful of 128x128 tiles and a long K (SD1.5's 8x8-pixel level at batch 4): 64x128 tiles, split until ~one workgroup per CU
      // while every split keeps >= 20 K-steps.  tools/gemm_probe4.py: conv 256 x 1280 x 11520 41.7 -> 28.2 us (with adapter 48.5 -> 33.8),
      // x 23040 61.2 -> 41.2, ff2 256 x 1280 x 5120 31.2 -> 18.0 (the 128x128 tile split 13 ways moved 17 MB of partial slabs).
      p.tile = 2;
      if (!p.splitk) {
        const long t2 = (long)((p.M + 63) / 64) * ((p.N + 127) / 128);
        int sk = (int)((256 + t2 / 2) / t2);
        if (sk > nk / 20) sk = nk / 20;
        p.splitk = (ws && sk > 1) ? sk : 1;
      }
    }
    else p.tile = 1;
  }
  int bm, bn;
  tile_dims(p.tile, bm, bn);
  if (R16 && p.lora_group_n > 0 && (p.lora_group_n % bn) && pin.tile == 0 && p.lora_group_n % 64 == 0) {
    p.tile = 3;   // narrow groups (head width 64): the 64x64 tile never straddles two adapters
    tile_dims(p.tile, bm, bn);
  }
  if (R16 && p.lora_group_n > 0 && (p.lora_group_n % bn))
    SDLT_FAIL(SDLT_ERR_SHAPE, "sdlt_gemm_bf16: lora_group_n=%d is not a multiple of the %d-column tile", p.lora_group_n, bn);
  if (p.splitk == 0) {
    // Split K until ~one workgroup per CU exists, keeping >= 8 K-steps per split (the fenced hand-off costs a few us).
    const long tiles = (long)((p.M + bm - 1) / bm) * ((p.N + bn - 1) / bn);
    const int nk = ktot / BK;
    int sk = 1;
    if (p.ws_slab && p.ws_cnt && tiles <= 96) {
      sk = (int)((256 + tiles / 2) / tiles);
      const int min_steps = tiles <= 32 ? 3 : 8;    // a handful of tiles (text encoders, M = 128): latency-bound, split harder
      if (sk > nk / min_steps) sk = nk / min_steps;
      if (sk > 16) sk = 16;
      if (sk < 1) sk = 1;
    }
    p.splitk = sk;
  }
#define SDLT_TILE_CASES(NSV)                                         \
  switch (p.tile) {                                                  \
    case 1: return launch<4, 2, 4, MODE, R16, NSV>(p, s);            \
    case 2: return launch<2, 2, 4, MODE, R16, NSV>(p, s);            \
    case 3: return launch<2, 2, 2, MODE, R16, NSV>(p, s);            \
    case 4: return launch<8, 2, 4, MODE, R16, NSV>(p, s);            \
    case 5: return launch<4, 4, 2, MODE, R16, NSV>(p, s);            \
    case 6: return launch<8, 4, 4, MODE, R16, NSV>(p, s);            \
    case 7: if constexpr (R16 <= 1) return launch<4, 5, 2, MODE, R16, NSV, 0, 0, 4>(p, s); else break; \
    case 8: if constexpr (R16 <= 1) return launch<2, 5, 2, MODE, R16, NSV, 0, 0, 4>(p, s); else break; \
  }
  // (stages == 1, the register-staged loader, is kept in the kernel source but not instantiated: hipcc places its
  //  staging registers in scratch - measured 3-6x slower than the LDS-DMA ring on every SDXL shape.)
  if (p.ln_c1) {   // folded LayerNorm: rank pad 16 on tiles 1, 2, 3, 8 (either ring depth), ff.net.0.proj + GEGLU on tiles 1, 2, 3, 7, 8 (deep ring)
    // ln_parts (the producer left row partials: no statistics in the K walk) exists for the shapes of the 1280-wide blocks; anything else
    // computes the statistics itself
    const bool parts = p.ln_parts && p.ln_nparts >= 2 && p.ln_nparts <= 16 && !(p.ln_nparts & 1) && !(((uintptr_t)p.ln_parts) & 15);
    if constexpr (MODE == 0 && R16 == 1) {
      if (p.tile != 1 && p.tile != 2 && p.tile != 3 && p.tile != 8) p.tile = (p.lora_group_n > 0 && (p.lora_group_n % 128)) ? 3 : 1;
      if (parts && p.stages != 2) {
        switch (p.tile) {
          case 1: return launch<4, 2, 4, 0, 1, 4, 0, 0, 2, 0, 2>(p, s);
          case 2: return launch<2, 2, 4, 0, 1, 4, 0, 0, 2, 0, 2>(p, s);
        }
      }
      if (p.stages == 2) {
        switch (p.tile) {
          case 1: return launch<4, 2, 4, 0, 1, 2, 0, 0, 2, 0, 1>(p, s);
          case 2: return launch<2, 2, 4, 0, 1, 2, 0, 0, 2, 0, 1>(p, s);
          case 3: return launch<2, 2, 2, 0, 1, 2, 0, 0, 2, 0, 1>(p, s);
          case 8: return launch<2, 5, 2, 0, 1, 2, 0, 0, 4, 0, 1>(p, s);
        }
      } else {
        switch (p.tile) {
          case 1: return launch<4, 2, 4, 0, 1, 4, 0, 0, 2, 0, 1>(p, s);
          case 2: return launch<2, 2, 4, 0, 1, 4, 0, 0, 2, 0, 1>(p, s);
          case 3: return launch<2, 2, 2, 0, 1, 4, 0, 0, 2, 0, 1>(p, s);
          case 8: return launch<2, 5, 2, 0, 1, 4, 0, 0, 4, 0, 1>(p, s);
        }
      }
    }
    if constexpr (MODE == 0 && R16 == 0) {
      if (p.epi_op == 1) {
        if (p.tile == 4 || p.tile == 5 || p.tile == 6) p.tile = 1;
        if (parts && p.tile == 7) return launch<4, 5, 2, 0, 0, 4, 0, 0, 4, 1, 2>(p, s);
        switch (p.tile) {
          case 1: return launch<4, 2, 4, 0, 0, 4, 0, 0, 2, 1, 1>(p, s);
          case 2: return launch<2, 2, 4, 0, 0, 4, 0, 0, 2, 1, 1>(p, s);
          case 3: return launch<2, 2, 2, 0, 0, 4, 0, 0, 2, 1, 1>(p, s);
          case 7: return launch<4, 5, 2, 0, 0, 4, 0, 0, 4, 1, 1>(p, s);
          case 8: return launch<2, 5, 2, 0, 0, 4, 0, 0, 4, 1, 1>(p, s);
        }
      }
    }
    SDLT_FAIL(SDLT_ERR_UNSUPPORTED, "sdlt_gemm_bf16: folded LayerNorm exists for rank-16 adapter products and for ff.net.0.proj + GEGLU (lora_R %d, epi_op %d, tile %d)", p.lora_R, p.epi_op, p.tile);
  }
  if (p.lora_group_k > 0 && pin.tile == 0) {
    // only the deep-ring variants of tiles 1..3 exist for K-grouped adapters; with the long K = G*C loop the 128x128 tile
    // wins as soon as it yields >= 64 workgroups (4096x640x1920: 31 us vs 61 us for 64x64)
    const long t128k = (long)((p.M + 127) / 128) * ((p.N + 127) / 128);
    if (t128k >= 64 || p.tile < 1 || p.tile > 3) p.tile = 1;
  }
  if (p.batch) {   // batched launches exist for what uses them: plain GEMM mode, no LoRA or rank pad 16, tiles 1..3, deep ring
    if constexpr (MODE == 0 && R16 == 0) {
      if (p.n_batch < 1 || p.n_batch > 65535) SDLT_FAIL(SDLT_ERR_SHAPE, "sdlt_gemm_bf16: n_batch=%d", p.n_batch);
      switch (p.tile) {
        case 1: return launch<4, 2, 4, 0, 0, 4, 0, 1>(p, s);
        case 2: return launch<2, 2, 4, 0, 0, 4, 0, 1>(p, s);
        case 3: return launch<2, 2, 2, 0, 0, 4, 0, 1>(p, s);
      }
    }
    if constexpr (MODE == 0 && R16 >= 1) {
      if (p.n_batch < 1 || p.n_batch > 65535) SDLT_FAIL(SDLT_ERR_SHAPE, "sdlt_gemm_bf16: n_batch=%d", p.n_batch);
      if (p.lora_group_k > 0) {
        switch (p.tile) {
          case 1: return launch<4, 2, 4, 0, R16, 4, 4, 1>(p, s);
          case 2: return launch<2, 2, 4, 0, R16, 4, 4, 1>(p, s);
          case 3: return launch<2, 2, 2, 0, R16, 4, 4, 1>(p, s);
        }
      } else {
        switch (p.tile) {
          case 1: return launch<4, 2, 4, 0, R16, 4, 0, 1>(p, s);
          case 2: return launch<2, 2, 4, 0, R16, 4, 0, 1>(p, s);
          case 3: return launch<2, 2, 2, 0, R16, 4, 0, 1>(p, s);
        }
      }
    }
    SDLT_FAIL(SDLT_ERR_UNSUPPORTED, "sdlt_gemm_bf16: batched launch needs mode 0 and tile 1..3 (tile %d)", p.tile);
  }
  if (p.lora_group_k > 0) {   // K-grouped adapters: plain GEMM mode, tiles 1..3, deep ring (as deep as the G x rank T tile leaves room for)
    if constexpr (MODE == 0 && R16 >= 1) {
      switch (p.tile) {
        case 1: return launch<4, 2, 4, 0, R16, 4, 4>(p, s);
        case 2: return launch<2, 2, 4, 0, R16, 4, 4>(p, s);
        case 3: return launch<2, 2, 2, 0, R16, 4, 4>(p, s);
      }
    }
    SDLT_FAIL(SDLT_ERR_UNSUPPORTED, "sdlt_gemm_bf16: lora_group_k needs mode 0 and tile 1..3 (tile %d)", p.tile);
  }
  if (p.epi_op) {   // fused GEGLU / activation epilogues: plain GEMM mode without LoRA, tiles 1, 2, 3, 7, 8 with the deep ring
    if constexpr (MODE == 0 && R16 == 0) {
      if (p.tile == 4 || p.tile == 5 || p.tile == 6) p.tile = 1;
#define SDLT_EPI_CASES(E)                                                      \
      switch (p.tile) {                                                        \
        case 1: return launch<4, 2, 4, 0, 0, 4, 0, 0, 2, E>(p, s);             \
        case 2: return launch<2, 2, 4, 0, 0, 4, 0, 0, 2, E>(p, s);             \
        case 3: return launch<2, 2, 2, 0, 0, 4, 0, 0, 2, E>(p, s);             \
        case 7: return launch<4, 5, 2, 0, 0, 4, 0, 0, 4, E>(p, s);             \
        case 8: return launch<2, 5, 2, 0, 0, 4, 0, 0, 4, E>(p, s);             \
      }
      switch (p.epi_op) {
        case 1: SDLT_EPI_CASES(1) break;
        case 2: SDLT_EPI_CASES(2) break;
        case 3: SDLT_EPI_CASES(3) break;
        case 4: SDLT_EPI_CASES(4) break;
      }
#undef SDLT_EPI_CASES
    }
    SDLT_FAIL(SDLT_ERR_UNSUPPORTED, "sdlt_gemm_bf16: fused epilogue %d needs mode 0 without LoRA (tile %d)", p.epi_op, p.tile);
  }
  if (p.stages == 2) { SDLT_TILE_CASES(2) } else { SDLT_TILE_CASES(4) }
#undef SDLT_TILE_CASES
  SDLT_FAIL(SDLT_ERR_UNSUPPORTED, "sdlt_gemm_bf16: tile id %d", p.tile);
}

}  // namespace

#ifdef SDLT_GEMM_TRACE
extern "C" int sdlt_gemm_trace_read(long long* out16) { return (int)hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_gemm_tr), sizeof(long long) * 16); }
#endif

extern "C" int sdlt_gemm_bf16(const sdlt_gemm_params* pp, void* stream) {
  const sdlt_gemm_params& p = *pp;
  hipStream_t s = (hipStream_t)stream;
  if (p.M <= 0 || p.N <= 0) SDLT_FAIL(SDLT_ERR_SHAPE, "sdlt_gemm_bf16: M=%d N=%d", p.M, p.N);
  if (p.K <= 0 || (p.K % BK) || (p.K2 % BK) || p.K2 < 0)
    SDLT_FAIL(SDLT_ERR_SHAPE, "sdlt_gemm_bf16: K=%d K2=%d must be positive multiples of 64", p.K, p.K2);
  if ((p.ldx % 8) || (p.ldw % 8) || (p.K2 && ((p.ldx2 % 8) || (p.ldw2 % 8))))
    SDLT_FAIL(SDLT_ERR_ALIGN, "sdlt_gemm_bf16: operand rows must be 16-byte aligned (ld %% 8)");
  if (((uintptr_t)p.X | (uintptr_t)p.W | (uintptr_t)p.X2 | (uintptr_t)p.W2 | (uintptr_t)p.Adown) & 15)
    SDLT_FAIL(SDLT_ERR_ALIGN, "sdlt_gemm_bf16: operand base pointers must be 16-byte aligned");
  if (p.ln_c1 && (p.N & 3)) SDLT_FAIL(SDLT_ERR_SHAPE, "sdlt_gemm_bf16: N=%d with a folded LayerNorm must be a multiple of 4", p.N);
  if (p.mode == 1) {
    if (!p.zero) SDLT_FAIL(SDLT_ERR_SHAPE, "sdlt_gemm_bf16: conv mode needs a zero page");
    if (p.Cin % BK || p.K != 9 * p.Cin) SDLT_FAIL(SDLT_ERR_SHAPE, "sdlt_gemm_bf16: conv needs Cin%%64==0 and K==9*Cin (Cin=%d K=%d)", p.Cin, p.K);
    if (p.K2) SDLT_FAIL(SDLT_ERR_UNSUPPORTED, "sdlt_gemm_bf16: conv + second segment");
    if (p.M % (p.Hout * p.Wout)) SDLT_FAIL(SDLT_ERR_SHAPE, "sdlt_gemm_bf16: conv M %% (Hout*Wout)");
  } else if (p.mode != 0) {
    SDLT_FAIL(SDLT_ERR_UNSUPPORTED, "sdlt_gemm_bf16: mode %d", p.mode);
  }
  if (p.epi_op) {
    if (p.epi_op < 1 || p.epi_op > 4) SDLT_FAIL(SDLT_ERR_UNSUPPORTED, "sdlt_gemm_bf16: epi_op %d", p.epi_op);
    if (p.mode != 0 || p.lora_R) SDLT_FAIL(SDLT_ERR_UNSUPPORTED, "sdlt_gemm_bf16: fused epilogues exist for plain GEMMs without LoRA");
    const int nmul = p.epi_op == 1 ? 32 : (p.epi_op == 2 ? 16 : 8);
    if (p.out_fp32 || p.Ct || p.batch || p.rowbias || (p.N % nmul) || (p.epi_op != 4 && (!p.epi_out || (p.ld_epi_out & 7) || ((uintptr_t)p.epi_out & 15))))
      SDLT_FAIL(SDLT_ERR_UNSUPPORTED, "sdlt_gemm_bf16: fused epilogue %d needs a bf16 output, no Ct / batch / row bias, N %% %d == 0, 16-byte aligned rows", p.epi_op, nmul);
    if ((p.epi_op == 3 || p.epi_op == 4) && (!p.C || (p.ldc & 7) || ((uintptr_t)p.C & 15) || (p.R && ((p.ldr & 7) || ((uintptr_t)p.R & 15)))))
      SDLT_FAIL(SDLT_ERR_ALIGN, "sdlt_gemm_bf16: fused activation epilogue: C / R rows must be 16-byte aligned");
    if (p.epi_op == 4 && (!p.epi_in || (p.ld_epi_in & 7) || ((uintptr_t)p.epi_in & 15)))
      SDLT_FAIL(SDLT_ERR_ALIGN, "sdlt_gemm_bf16: fused activation backward: pre-activation rows must be 16-byte aligned");
    if (p.epi_op == 1 && (!p.C || (p.ldc & 7) || ((uintptr_t)p.C & 15) || (p.R && ((p.ldr & 7) || ((uintptr_t)p.R & 15)))))
      SDLT_FAIL(SDLT_ERR_ALIGN, "sdlt_gemm_bf16: fused GEGLU forward: C / R rows must be 16-byte aligned");
    if (p.epi_op == 2 && (!p.epi_in || (p.ld_epi_in & 7) || ((uintptr_t)p.epi_in & 15) || p.R))
      SDLT_FAIL(SDLT_ERR_ALIGN, "sdlt_gemm_bf16: fused GEGLU backward: F1 rows must be 16-byte aligned, no residual");
  }
  if (p.ln_c1) {
    if (p.mode != 0 || p.K2 || p.batch || p.alpha != 1.f || p.col_scale || p.lora_group_k > 0 || p.out_fp32 || (p.lora_R && p.lora_R != 16) ||
        (p.lora_R && !p.ln_adapter) || (p.epi_op != 0 && p.epi_op != 1) || (p.splitk > 1))
      SDLT_FAIL(SDLT_ERR_UNSUPPORTED, "sdlt_gemm_bf16: folded LayerNorm needs mode 0, one K segment, alpha 1, bf16 out, lora_R 0 / 16 (+ ln_adapter), epi_op 0 / 1, no batch / DoRA / split-K");
    if ((((uintptr_t)p.ln_c1) & 15) || (((uintptr_t)p.ln_adapter) & 15) || (((uintptr_t)p.ln_stats) & 7))
      SDLT_FAIL(SDLT_ERR_ALIGN, "sdlt_gemm_bf16: ln_c1 / ln_adapter 16-byte, ln_stats 8-byte aligned");
  }
  if (p.accumulate && !p.out_fp32) SDLT_FAIL(SDLT_ERR_UNSUPPORTED, "sdlt_gemm_bf16: accumulate needs an fp32 output (use R for bf16)");
  int r16 = 0;
  if (p.col_scale && !p.lora_R) SDLT_FAIL(SDLT_ERR_UNSUPPORTED, "sdlt_gemm_bf16: col_scale (DoRA) is an option of adapter launches (lora_R > 0)");
  if (p.col_scale && (((uintptr_t)p.col_scale) & 15)) SDLT_FAIL(SDLT_ERR_ALIGN, "sdlt_gemm_bf16: col_scale must be 16-byte aligned");
  if (p.lora_R) {
    if (p.lora_R != 16 && p.lora_R != 32 && p.lora_R != 64) SDLT_FAIL(SDLT_ERR_UNSUPPORTED, "sdlt_gemm_bf16: padded LoRA rank %d (16/32/64)", p.lora_R);
    if (p.K2) SDLT_FAIL(SDLT_ERR_UNSUPPORTED, "sdlt_gemm_bf16: LoRA + second segment");
    if ((p.ld_adown % 8) || (p.ld_bup % 4) || (p.T_out && (p.ld_t % 4))) SDLT_FAIL(SDLT_ERR_ALIGN, "sdlt_gemm_bf16: LoRA operand alignment");
    r16 = p.lora_R / 16;
    if (p.lora_group_k > 0) {
      if (p.lora_group_n > 0 || p.mode != 0 || (p.lora_group_k % BK) || (p.K % p.lora_group_k) || p.K / p.lora_group_k > 4)
        SDLT_FAIL(SDLT_ERR_UNSUPPORTED, "sdlt_gemm_bf16: lora_group_k=%d needs mode 0, K = G*group_k with G <= 4, group_k %% 64 == 0", p.lora_group_k);
    }
  }
  int rc;
#define DISPATCH(MODE_)                                                 \
  switch (r16) {                                                        \
    case 0: rc = dispatch_tile<MODE_, 0>(p, s); break;                  \
    case 1: rc = dispatch_tile<MODE_, 1>(p, s); break;                  \
    case 2: rc = dispatch_tile<MODE_, 2>(p, s); break;                  \
    default: rc = dispatch_tile<MODE_, 4>(p, s); break;                 \
  }
  if (p.mode == 0) { DISPATCH(0) } else { DISPATCH(1) }
#undef DISPATCH
  if (rc != SDLT_OK) return rc;
  SDLT_CHECK_LAUNCH();
  return SDLT_OK;
}
