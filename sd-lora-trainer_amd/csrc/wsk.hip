// Wave-split-K GEMM for the short, wide products of the batch-1 UNet (gfx950):  C[M,N] = X[M,K] . W[N,K]^T (+ bias, + residual),
// M = 1024 ... 4096 tokens, N = 640 / 1280.
//
// The tiled kernel (gemm.hip) needs >= 128 x 128 tiles for its block-synchronous LDS pipeline to run well, which cuts a 1024 x 1280
// output into 80 workgroups: 80 of 256 CUs stream 655 KB of operands each and the launch takes 14.5 us for 3.4 GFLOP.  What bounds such a
// product is the bytes each CU pulls through its load path, (BM + BN) K 2 per tile - so the tile count has to match the CU count:
//   * 64 x 80 tiles: exactly 256 workgroups for 1024 x 1280, 369 KB per CU at K = 1280;
//   * a workgroup is 4 waves that SPLIT K (interleaved 64-column steps) - every wave owns the whole 64 x 80 tile (20 accumulators), stages
//     its own operands through a private LDS ring (global_load_lds, full 128-byte rows, XOR-swizzled source side) and never meets a block
//     barrier in the K loop: 36 KB of loads in flight per CU from the first instruction, counted vmcnt waits;
//   * the 4 partial tiles are added through LDS once, in wave order (bitwise reproducible), then bias / residual / bf16 store.
#include <cstdlib>
#include "common.h"
#include "../../include/sdlt_kernels.h"

namespace {

constexpr int NW = 4;
constexpr int ROWB = 128;

template <int N_>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory"); }

struct wsk_params {
  const bf16_t* X; int64_t ldx;
  const bf16_t* W; int64_t ldw;
  const float* bias;
  const bf16_t* R; int64_t ldr;
  bf16_t* Y; int64_t ldy;
  int M, N, K, map2d;
  // rank-16 adapter (LORA kernels): Y += bf16(s * X Adown^T) . Bup^T;  Adown [16, K], Bup [N, 16] (bf16 shadows, rank padded with zero rows /
  // columns); T_out [M, 16] receives bf16(s * X Adown^T) for the adapter-gradient launch (or NULL)
  const bf16_t* Adown; int64_t ld_adown;
  const bf16_t* Bup; int64_t ld_bup;
  bf16_t* T_out; int64_t ld_t;
  float lora_scale;
  // K-grouped adapters (group_k > 0; the input gradient of stacked projections): K is G = K / group_k groups of group_k columns, group g has
  // its own rank-16 adapter: T_g = X[:, group g] . Adown[:, group g]^T, Y += sum_g bf16(s T_g) . Bup[:, 16 g .. 16 g + 15]^T; T_out [M, 16 G]
  int group_k;
  int stagger;
  // LayerNorm folded into the product (LN kernels): X holds the RAW rows, W = W o gamma (bf16), `bias` = c2 = W beta + bias and
  //   Y = rstd (X W^T - mean c1[n]) + c2[n] (+ adapter, + residual);  (mean, rstd) of a row come out of the same K walk (ln_frag_stats on the X fragments: packed bf16 dot
  //   products on the VALU beside the MFMAs) and are written to ln_stats [M, 2] by column tile 0.
  //   With an adapter: Adown = A o gamma and T = s (rstd (X Adown^T - mean cA) + abeta), ln_adapter = cA[16] | abeta[16].
  const float* ln_c1;
  float* ln_stats;
  const float* ln_adapter;
  float ln_eps;
  // row partials for the NEXT LayerNorm (any kernel): ln_parts [M, N / 80] float2 = (sum y, sum (y - tile mean)^2) of the ROUNDED output row over this tile's 80
  // columns - the consumer GEMM of a folded LayerNorm (sdlt_gemm_params.ln_parts) adds the N / 80 partials of a row instead of walking it
  float2* ln_parts;
  // W in fragment-major order (kernels with WP; sdlt_wsk_pack_weight's output, frozen weights only): [N / 80][K / 64][2][5][64 lanes][8 bf16] - lane (r, g)
  // of fragment (kk, j) holds W[n0 + 16 j + r][k0 + 32 kk + 8 g .. + 7], i.e. the MFMA A operand as it sits in registers.  A wave's K step is
  // ten contiguous 1 KB wave loads (16 B per lane, whole 128-byte lines) that never touch LDS.
  const bf16_t* Wp;
  // implicit 3 x 3 convolution (CONV kernels; stride 1, padding 1): X is the NHWC activation [B Hc Wc, Cin], K = 9 Cin with k = tap Cin + ci (tap = 3 dy + dx), row m of the
  // product reads pixel (y + dy - 1, x + dx - 1) of its image (flip: (y + 1 - dy, x + 1 - dx), the input gradient's flipped taps) or the zero page outside it;
  // rowbias bf16 [B, N] (or NULL) is added per image (the resnets' time-embedding projection), rows_per_batch = Hc Wc
  int Hc, Wc, Cin, flip;
  const bf16_t* zero;
  const bf16_t* rowbias; int64_t ld_rowbias;
  // row dots for the attention backward that reads Y next (Y = dO of a self-attention with 64-wide heads, R = the forward's O instead of a residual: it is NOT added):
  // dotD[(b H + h) dot_nq + q] += sum over head h's columns of this tile of rounded(Y[m, n]) O[m, n], H = N / 64, m = b dot_nq + q.  A head's 64 columns lie in at
  // most TWO 80-column tiles and every tile adds ONE value per (row, head): onto a zeroed slot the result does not depend on the order (x + y == y + x), so the float
  // atomics stay bitwise reproducible.  Replaces the D pre-pass of the attention backward (sdlt_attn_params.d_ready).
  float* dotD; int dot_nq;
  // DoRA (LORA kernels): per-column factor m / ||W + s B A|| applied to the product + adapter before the bias (sdlt_gemm_params.col_scale's contract), fp32 [N] or NULL
  const float* col_scale;
  // ... and the layer's own output before the residual (DoRA's magnitude gradient needs it: d m = sum_rows dY (y0 - bias) / m): Y0 [M, N] = rounded(col_scale (X W^T + adapter) + bias),
  // written next to Y = Y0's fp32 value + R - one launch instead of the product + an add2d launch.  NULL: not written.
  bf16_t* Y0; int64_t ldy0;
  // next-weight prefetch (round 6, lab): pf_ptr = the PACKED weight of the wave-split-K product that follows this launch in the step (sdlt_wsk_pack_weight's layout, N_next / 80 column-tile
  // blocks of pf_tile_bytes each); behind its K walk every workgroup touches - one lane per 128-byte line - its share of the first pf_head_bytes of the pf_ntn / 4 column tiles
  // the SAME XCD will walk in that launch (the 2 x 4 XCD map: XCD x owns column quarter x & 3), so that the next launch's first K steps hit in this XCD's L2 instead of paying
  // the fabric's first-touch burst (the launches are bound by it: DESIGN 4.14).  NULL: off.
  const char* pf_ptr; int pf_tile_bytes, pf_head_bytes, pf_ntn;
};

// -DSDLT_WSK_TRACE (tools/wsk_trace.py): thread 0 of workgroup 0 stamps clock64() at the phase boundaries; sdlt_wsk_trace_read copies them out
#ifdef SDLT_WSK_TRACE
__device__ long long g_wsk_tr[16];
#define WTR(i_) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_wsk_tr[i_] = clock64(); } while (0)
#else
#define WTR(i_) do {} while (0)
#endif

// MBK x 16 rows, JN x 16 columns per workgroup; R ring slots per wave
// KG: 0 no adapter, 1 one rank-16 adapter, 2..3 that many K groups with an adapter each
// WP: the weight operand comes from the fragment-major copy straight into a register ring of R stages (wsk_params::Wp); the LDS rings hold the
//     activation (and LoRA-down) rows only.  In wsk_kernel no wave shares an operand with another wave, so LDS is only a layout converter for W -
//     and the K walk is bound by the rate four waves can issue LDS-DMA pieces at: taking the frozen operand's 56 % of the bytes out of
//     that path, and its layout conversion to pack time, is what this variant is for.
// CONV: implicit-GEMM 3 x 3 convolution (wsk_params::Hc ...): the activation pieces are gathered per tap (per-lane addresses, zero page for the halo), everything else is
//     the same walk.  Replaces the tiled kernel's split-K (fp32 slabs through HBM: 23.6 MB written for a 2.6 MB output) for the 32 x 32 level of the UNet.
// RG (KG == 1 only): the adapter's padded rank as RG groups of 16 rows - rank pad 32 (the hyper-parameter sweep's rank 24, create_hyperparam_sweep.py:76; instantiated for
//     RG = 2): RG x 16 LoRA-down rows ride the K walk, T is RG tiles of 16 ranks, the LoRA-up RG MFMAs of 16x16x16 per output block.  Packed weights only (the row-major ring has no room).
template <int MBK, int JN, int R, int KG = 0, bool LN = false, bool WP = false, bool CONV = false, int RG = 1>
__global__ __launch_bounds__(64 * NW) void wsk_kernel(const wsk_params p) {
  constexpr bool LORA = KG > 0;
  static_assert(!LN || KG <= 1, "the folded LayerNorm belongs to forward products (no K-grouped adapters)");
  static_assert(!CONV || (!LN && KG <= 1), "convolutions: plain or with one rank-16 adapter");
  static_assert(RG == 1 || (KG == 1 && WP && !CONV), "rank pads above 16: one adapter, packed weights, no convolution");
  constexpr int XR = 16 * MBK, WR = 16 * JN, AOFF = XR + (WP ? 0 : WR), SROWS = AOFF + (LORA ? 16 * RG : 0), SLOT = SROWS * ROWB,
                PIECES = SROWS / 8 + (WP ? 2 * JN : 0);      // vector-memory operations of one K step of one wave (what the counted waits count)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  WTR(0);
  // every kernel argument is fetched by the FIRST scalar-load batch (hipcc otherwise fetches a field where it is first used: three more scalar round trips,
  // each behind an s_waitcnt lgkmcnt(0), in the prologue and in front of the epilogue)
  asm volatile("" ::"s"(p.X), "s"(p.ldx), "s"(p.W), "s"(p.ldw), "s"(p.bias), "s"(p.R), "s"(p.ldr), "s"(p.Y), "s"(p.ldy), "s"(p.M), "s"(p.N), "s"(p.K), "s"(p.map2d));
  asm volatile("" ::"s"(p.Adown), "s"(p.ld_adown), "s"(p.Bup), "s"(p.ld_bup), "s"(p.T_out), "s"(p.ld_t), "s"(p.lora_scale), "s"(p.group_k), "s"(p.stagger));
  asm volatile("" ::"s"(p.ln_c1), "s"(p.ln_stats), "s"(p.ln_adapter), "s"(p.ln_eps), "s"(p.ln_parts), "s"(p.Wp), "s"(p.dotD), "s"(p.dot_nq));
  if constexpr (CONV) asm volatile("" ::"s"(p.Hc), "s"(p.Wc), "s"(p.Cin), "s"(p.flip), "s"(p.zero), "s"(p.rowbias), "s"(p.ld_rowbias));
  if constexpr (LORA) asm volatile("" ::"s"(p.col_scale), "s"(p.Y0), "s"(p.ldy0));
  asm volatile("" ::"s"(p.pf_ptr), "s"(p.pf_tile_bytes), "s"(p.pf_head_bytes), "s"(p.pf_ntn));
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int ntn = p.N / WR, ntm = p.M / XR;
  int tm, tn;
  {
    // XCD-aware: the workgroups of one XCD (blockIdx % 8) take neighbouring column tiles over all row tiles, so every weight panel is
    // fetched into one L2 only and the activation rows are what the L2s share
    // 8 private L2s: an XCD that owns a (rows / 2) x (columns / 4) block of tiles fetches half of X and a quarter of W - 2.1x the operand
    // bytes over the fabric in total, against 4x when every XCD reads all of X (measured: the products are fabric-bound at ~5 TB/s)
    const int b = blockIdx.x, xcd = b & 7, idx = b >> 3;
    if (p.map2d && (ntm & 1) == 0 && (ntn & 3) == 0) {
      const int pr = ntm >> 1, pc = ntn >> 2;           // tiles per XCD: pr x pc
      const int dq = div_small_u(idx, pc);
      tm = (xcd >> 2) * pr + dq;
      tn = (xcd & 3) * pc + idx - dq * pc;
    } else {
      const int per = ntn >> 3;                          // column tiles per XCD (ntn % 8 == 0)
      tm = div_small_u(idx, per);
      tn = xcd * per + idx - tm * per;
    }
  }
  if (tm >= ntm || tn >= ntn) return;
  WTR(13);
  const int m0 = tm * XR, n0 = tn * WR;
  // wave w walks the 64-column steps w, w + 4, ... of K: K / 256 each, or one more for the first waves when K / 64 is not a multiple of 4 (convolutions: Cin = 1920)
  const int nsteps = ((p.K >> 6) - wave + NW - 1) >> 2;

  const int srow = lane >> 3, schunk = (lane & 7) ^ srow;
#ifdef SDLT_WSK_LAB      // bound experiments (tools/wsk_lab.sh; results are garbage): bit 1 of `stagger` = every workgroup reads row tile 0 of X, bit 2 = column tile 0 of W
  const int m0x = (p.stagger & 2) ? 0 : m0, tnw = (p.stagger & 4) ? 0 : tn;     // (operands then stay L2-resident: no fabric traffic for them), bit 3 = no MFMAs
#else
  const int m0x = m0, tnw = tn;
#endif
  const bf16_t* xsrc = p.X + (int64_t)(m0x + srow) * p.ldx + schunk * 8;
  const bf16_t* wsrc = p.W + (int64_t)(tnw * WR + srow) * p.ldw + schunk * 8;
  const bf16_t* asrc = LORA ? p.Adown + (int64_t)srow * p.ld_adown + schunk * 8 : nullptr;
  // CONV: the pixel of each of this lane's XR / 8 piece rows - linear index (= row of X) and (y, x) inside its image
  [[maybe_unused]] int pm[XR / 8], pyx[XR / 8];
  [[maybe_unused]] const bf16_t* zsrc = nullptr;
  if constexpr (CONV) {
    const int hw = p.Hc * p.Wc;
#pragma unroll
    for (int q = 0; q < XR / 8; ++q) {
      const int m = m0x + q * 8 + srow;
      const int rem = m - div_small(m, hw) * hw, y = div_small(rem, p.Wc);
      pm[q] = m;
      pyx[q] = (y << 16) | (rem - y * p.Wc);
    }
    zsrc = p.zero + schunk * 8;
  }
  const int64_t x8 = 8 * p.ldx, w8 = 8 * p.ldw, a8 = 8 * p.ld_adown;
  char* ring = smem + wave * (R * SLOT);
  // column tile tn starts its K walk `rot` steps in (the 16 tiles of a row block do not ask the L2 for the same X lines at the same time).  The
  // fp32 K sums are therefore ordered per tile: Y is a fixed function of the operands per tile (bitwise reproducible), but the adapter's
  // T = s X Adown^T, which every tile recomputes for itself, can differ between tiles in the last bit of its fp32 sum, i.e. rarely by one bf16 ulp
  // after rounding; T_out (the copy the adapter-gradient launch reads) is column tile 0's.  That is inside the
  // rounding of T itself (2^-9 relative) - the parity tests compare Y and the adapter gradients with that tolerance.
  // (Round 5, measured and dropped: a rotation that also spreads the 8 workgroups of an XCD that walk the SAME weight panel over the K range - so that each is the
  // first to touch only 1 / 8 of it - makes the XCD's working set the whole of X / 2 + W / 4 at once instead of a window that slides along K: K = 10240 went from
  // 44 to 74 us.  The lockstep of the workgroups is what lets a 4 MB L2 serve a 17 MB operand set.)
  const int rotn = p.K >> 8;              // (the shortest walk among the waves)
  const int rot = tn - div_small_u(tn, rotn) * rotn;
  auto issue = [&](int i, int slot) -> int {
    int ii = i + rot;
    ii = ii >= nsteps ? ii - nsteps : ii;
    const int k0 = (wave + NW * ii) * 64;
    char* dst = ring + slot * SLOT;
    if constexpr (CONV) {
      const int tap = div_small_u(k0, p.Cin), ci0 = k0 - tap * p.Cin;        // (wave-uniform; a 64-column step lies inside one tap: Cin % 64 == 0)
      const int dy = (tap * 11) >> 5, dx = tap - 3 * dy;                       // tap / 3 for tap < 9
      const int oy = p.flip ? 1 - dy : dy - 1, ox = p.flip ? 1 - dx : dx - 1;
      const int shift = oy * p.Wc + ox;
#pragma unroll
      for (int q = 0; q < XR / 8; ++q) {
        const int yy = (pyx[q] >> 16) + oy, xx = (pyx[q] & 0xffff) + ox;
        const bool ok = (unsigned)yy < (unsigned)p.Hc && (unsigned)xx < (unsigned)p.Wc;
        const bf16_t* src = ok ? p.X + (int64_t)(pm[q] + shift) * p.ldx + ci0 + schunk * 8 : zsrc;
        glds16(src, dst + q * 1024);
      }
    } else {
#pragma unroll
      for (int q = 0; q < XR / 8; ++q) glds16(xsrc + q * x8 + k0, dst + q * 1024);
    }
    if constexpr (!WP) {
#pragma unroll
      for (int q = 0; q < WR / 8; ++q) glds16(wsrc + q * w8 + k0, dst + XR * ROWB + q * 1024);
    }
    if constexpr (LORA) {
#pragma unroll
      for (int q = 0; q < 2 * RG; ++q) glds16(asrc + q * a8 + k0, dst + AOFF * ROWB + q * 1024);
    }
    return k0;
  };
  // WP: the 2 x JN weight fragments of K step i of this wave -> registers.  Hand-issued loads (saddr form: wave-uniform base, 32-bit lane offset): the
  // compiler's own wait-count pass cannot see that a refill and the step that consumes it are R steps apart (it merges the paths with and without a
  // refill and falls back to vmcnt(0) in front of the MFMAs - seen in the ISA), so these loads are invisible to it and the counted waits of the K walk
  // below cover them: queue order per step = activation pieces, then the weight loads, so "step i's weight loads have landed" implies its pieces have.
  // (The fragments are only read by the MFMAs behind that step's wait; nothing else may touch wr[][][] in between - checked in the ISA: no copies.)
  const uint32_t wvo = (uint32_t)lane * 16u;
  const char* wpbase = WP ? (const char*)p.Wp + ((int64_t)tnw * (p.K >> 6)) * (2 * JN * 1024) : nullptr;
  auto issue_w = [&](int i, bf16x8 (&dst)[2][JN]) {
    int ii = i + rot;
    ii = ii >= nsteps ? ii - nsteps : ii;
    const char* src = wpbase + (int64_t)(wave + NW * ii) * (2 * JN * 1024);
    const uint64_t sb = (uint64_t)src;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)sb), hi = __builtin_amdgcn_readfirstlane((uint32_t)(sb >> 32));
    const uint64_t sbase = ((uint64_t)hi << 32) | lo;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int j = 0; j < JN; ++j) {
        constexpr int dummy = 0; (void)dummy;
        const int f = kk * JN + j;              // fragment f at byte f * 1024: immediate offsets reach 4095, so every fourth fragment moves the base
        const uint64_t b4 = sbase + (uint64_t)(f >> 2) * 4096u;
        switch (f & 3) {
          case 0: asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(dst[kk][j]) : "v"(wvo), "s"(b4) : "memory"); break;
          case 1: asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024" : "=&v"(dst[kk][j]) : "v"(wvo), "s"(b4) : "memory"); break;
          case 2: asm volatile("global_load_dwordx4 %0, %1, %2 offset:2048" : "=&v"(dst[kk][j]) : "v"(wvo), "s"(b4) : "memory"); break;
          default: asm volatile("global_load_dwordx4 %0, %1, %2 offset:3072" : "=&v"(dst[kk][j]) : "v"(wvo), "s"(b4) : "memory"); break;
        }
      }
  };
  const int foff0 = r * ROWB + (((0 * 4 + g) ^ (r & 7)) << 4), foff1 = r * ROWB + (((1 * 4 + g) ^ (r & 7)) << 4);

  constexpr int TG = KG > 1 ? KG : (KG == 1 ? RG : 1);      // T tiles of 16 ranks: the K groups' adapters, or the rank groups of one adapter
  f32x4 acc[JN][MBK], tacc[TG][MBK];
#pragma unroll
  for (int mb = 0; mb < MBK; ++mb) {
#pragma unroll
    for (int tg = 0; tg < TG; ++tg) tacc[tg][mb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < JN; ++j) acc[j][mb] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  float ls1[LN ? MBK : 1], ls2[LN ? MBK : 1];      // LN: sum x / sum x^2 of row (mb, r) over this lane's 8-column chunks of this wave's K share
#pragma unroll
  for (int mb = 0; mb < (LN ? MBK : 1); ++mb) ls1[mb] = ls2[mb] = 0.f;
  // ---- the first K step's loads go out before anything else is computed (round 5; clock stamps: the prefill used to be issued 2.7 - 3.4 us into a launch that lasts 8)
  static_assert(MBK == NW, "the epilogue gives row block w of the tile to wave w");
  [[maybe_unused]] bf16x8 wr[WP ? R : 1][2][JN];      // WP: register ring of the weight fragments
  int kq[R];               // slot s holds the step whose first column is kq[s]
#pragma unroll
  for (int s = 0; s < R; ++s) kq[s] = 0;
  kq[0] = issue(0, 0);
  if constexpr (WP) issue_w(0, wr[0]);
  // ---- epilogue operands of the units this wave finishes - row block `wave` of the tile, all JN column blocks (unit q = column block q) -, requested now: loads at their point
  // of use sat exposed behind the last barrier (+1.1 us per launch with a residual; the LoRA-up fragments even behind an s_waitcnt vmcnt(0) of their own).  Queue order:
  // step 0, these, steps 1 .. R - 1 - so the counted waits of the K walk cover them from step 0 on and never wait FOR them alone.
  // Unconditional loads through selected pointers: with `if (p.R)` / `if (p.bias)` around them hipcc merged each loaded vector with its zero default through register copies and
  // put an s_waitcnt vmcnt(0) behind EVERY unit's pair - five serial round trips in front of the ring prefill (found in the ISA in round 5).  A missing residual / bias reads
  // in-bounds rows of Y / X instead and the epilogue skips the addition; adapter groups beyond the ones in use read group 0's columns and are skipped.
  const bf16_t* rsrc = (p.R ? p.R : p.Y) + (int64_t)(m0 + wave * 16 + r) * (p.R ? p.ldr : p.ldy) + n0 + 4 * g;
  const float* bsrc = (p.bias ? p.bias : (const float*)p.X) + n0 + 4 * g;       // (N floats <= 64 rows of X: K >= 256 columns)
  uint2 rpre[JN];
  f32x4 bpre[JN], c1pre[LN ? JN : 1];
  // RG = 4 (rank pad 64): the LoRA-up fragments and the column factors (60 registers) would not fit beside the K walk's 512 - they are requested behind it, in front of the
  // reduction's first barrier, which covers most of their round trip
  constexpr bool LATE = RG >= 4;
  uint2 bupf[JN][TG];
  [[maybe_unused]] f32x4 cspre[LORA ? JN : 1];
  [[maybe_unused]] const float* cssrc = nullptr;
  if constexpr (LORA) cssrc = (p.col_scale ? p.col_scale : (const float*)p.X) + n0 + 4 * g;
#pragma unroll
  for (int q = 0; q < JN; ++q) {
    rpre[q] = *(const uint2*)(rsrc + 16 * q);
    bpre[q] = *(const f32x4*)(bsrc + 16 * q);
    if constexpr (LORA && !LATE) cspre[q] = *(const f32x4*)(cssrc + 16 * q);
    if constexpr (LN) c1pre[q] = *(const f32x4*)(p.ln_c1 + n0 + 16 * q + 4 * g);
  }
  [[maybe_unused]] uint2 rbpre[JN];
  if constexpr (CONV) {      // the image's row of the per-image bias (none: the residual's address again, skipped by the epilogue)
    const int mrow = m0 + wave * 16 + r;
    const bf16_t* rb = p.rowbias ? p.rowbias + (int64_t)div_small(mrow, p.Hc * p.Wc) * p.ld_rowbias + n0 + 4 * g : rsrc;
#pragma unroll
    for (int q = 0; q < JN; ++q) rbpre[q] = *(const uint2*)(rb + 16 * q);
  }
  auto load_bup = [&]() {
    const int ngrp0 = KG > 1 ? p.K / p.group_k : TG;
    const bf16_t* bu = p.Bup + (int64_t)(n0 + r) * p.ld_bup + 4 * g;
#pragma unroll
    for (int q = 0; q < JN; ++q)
#pragma unroll
      for (int tg = 0; tg < TG; ++tg) bupf[q][tg] = *(const uint2*)(bu + (int64_t)(16 * q) * p.ld_bup + (tg < ngrp0 ? 16 * tg : 0));
  };
  if constexpr (LORA && !LATE) load_bup();
  WTR(14);
  [[maybe_unused]] f32x4 lnca[TG], lnab[TG];      // adapter constants of this lane's four rank rows 4g .. 4g+3 of every rank group (cA[16] | abeta[16] per group)
  if constexpr (LN && LORA) {
#pragma unroll
    for (int tg = 0; tg < TG; ++tg) { lnca[tg] = *(const f32x4*)(p.ln_adapter + 32 * tg + 4 * g); lnab[tg] = *(const f32x4*)(p.ln_adapter + 32 * tg + 16 + 4 * g); }
  }
  static_assert(WP || R == 2 || KG <= 1, "the K-grouped bookkeeping below tracks a 2-slot ring");

  if constexpr (WP) {
    // ---- K walk, weight fragments in registers: R stages in flight per wave; queue order per step: X (+ LoRA-down) pieces, then the weight loads
    // (Round 5, measured and dropped: touching the steps behind the ring's reach at the top of the kernel, one lane per 128-byte line, so that their HBM round
    // trips start with the launch - 5-11 % SLOWER on every shape: the extra requests queue in front of the prefill's own first-touch loads.)
#pragma unroll
    for (int s = 1; s < R; ++s)
      if (s < nsteps) { kq[s] = issue(s, s); issue_w(s, wr[s]); }
    WTR(1);
    const bool refill_first = !(p.stagger & 1) || (wave & 1) == 0;
    for (int i0 = 0; i0 < nsteps; i0 += R) {
#pragma unroll
      for (int s = 0; s < R; ++s) {
        const int i = i0 + s;
        if (i < nsteps) {            // (wave-uniform)
          const int kcur = kq[s];
          const int after = nsteps - 1 - i;
          if (after >= R - 1) wait_vmcnt<PIECES * (R - 1)>();
          else if (R > 2 && after == R - 2) wait_vmcnt<PIECES * (R > 2 ? R - 2 : 0)>();
          else if (R > 3 && after == R - 3) wait_vmcnt<PIECES * (R > 3 ? R - 3 : 0)>();
          else wait_vmcnt<0>();
          if (i < 6) WTR(2 + i);
          const char* base = ring + s * SLOT;
          bf16x8 xf[2][MBK], af[RG][2];
#pragma unroll
          for (int kk = 0; kk < 2; ++kk) {
            const int fo = kk ? foff1 : foff0;
#pragma unroll
            for (int mb = 0; mb < MBK; ++mb) xf[kk][mb] = *(const bf16x8*)(base + mb * 16 * ROWB + fo);
            if constexpr (LORA) {
#pragma unroll
              for (int rg = 0; rg < RG; ++rg) af[rg][kk] = *(const bf16x8*)(base + (AOFF + 16 * rg) * ROWB + fo);
            }
          }
          if (refill_first && i + R < nsteps) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            kq[s] = issue(i + R, s);
          }
#ifdef SDLT_WSK_LAB
          if (p.stagger & 8) {       // no MFMAs: the fragments are still consumed (one VALU op each)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
              for (int j = 0; j < JN; ++j) acc[j][0][0] += (float)wr[s][kk][j][0];
#pragma unroll
              for (int mb = 0; mb < MBK; ++mb) acc[0][mb][1] += (float)xf[kk][mb][0];
            }
          } else
#endif
#pragma unroll
          for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int j = 0; j < JN; ++j)
#pragma unroll
              for (int mb = 0; mb < MBK; ++mb) acc[j][mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wr[s][kk][j], xf[kk][mb], acc[j][mb], 0, 0, 0);
          if constexpr (KG > 1) {
            const int tgrp = kcur / p.group_k;
#pragma unroll
            for (int tg = 0; tg < TG; ++tg)
              if (tg == tgrp) {
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                  for (int mb = 0; mb < MBK; ++mb) tacc[tg][mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0][kk], xf[kk][mb], tacc[tg][mb], 0, 0, 0);
              }
          } else if constexpr (LORA) {      // one adapter: every rank group rides every step
#pragma unroll
            for (int rg = 0; rg < RG; ++rg)
#pragma unroll
              for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int mb = 0; mb < MBK; ++mb) tacc[rg][mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[rg][kk], xf[kk][mb], tacc[rg][mb], 0, 0, 0);
          }
          if constexpr (LN) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
              for (int mb = 0; mb < MBK; ++mb) ln_frag_stats(xf[kk][mb], ls1[mb], ls2[mb]);
          }
          if (i + R < nsteps) {
            if (!refill_first) kq[s] = issue(i + R, s);
            issue_w(i + R, wr[s]);
          }
        }
      }
    }
  } else {
#pragma unroll
  for (int s = 1; s < R; ++s)
    if (s < nsteps) kq[s] = issue(s, s);
  int slot = 0;
  WTR(1);
  for (int i = 0; i < nsteps; ++i) {
    const int kcur = slot == 0 ? kq[0] : kq[R - 1];
    const int after = nsteps - 1 - i;
    if (after >= R - 1) wait_vmcnt<PIECES * (R - 1)>();
    else if (R > 2 && after == 1) wait_vmcnt<PIECES>();
    else wait_vmcnt<0>();
    if (i < 6) WTR(2 + i);
    const char* base = ring + slot * SLOT;
    bf16x8 xf[2][MBK], wf[2][JN], af[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int fo = kk ? foff1 : foff0;
#pragma unroll
      for (int mb = 0; mb < MBK; ++mb) xf[kk][mb] = *(const bf16x8*)(base + mb * 16 * ROWB + fo);
#pragma unroll
      for (int j = 0; j < JN; ++j) wf[kk][j] = *(const bf16x8*)(base + (XR + 16 * j) * ROWB + fo);
      if constexpr (LORA) af[kk] = *(const bf16x8*)(base + (XR + WR) * ROWB + fo);
    }
    // (spreading the refill's DMA instructions between the MFMAs was measured and LOSES 10-15 %: the ring is one step deep, every cycle
    // a piece is issued later is a cycle less of its latency hidden)
    // The CU's texture path takes 1 KB of DMA per 16 cycles, shared by the 4 waves: when all of them refill at once each sits ~1100 cycles in
    // DMA issue and then all run their 680 cycles of MFMAs with the path idle (PMC: 49 % of the wave cycles are issue stalls).  Odd waves
    // therefore refill AFTER their MFMAs: the two halves of the workgroup alternate between the two resources.
    const bool refill_first = !(p.stagger & 1) || (wave & 1) == 0;
    if (refill_first && i + R < nsteps) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the slot is refilled: its fragments must be in registers first
      const int kn = issue(i + R, slot);
      if (slot == 0) kq[0] = kn; else kq[R - 1] = kn;
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int j = 0; j < JN; ++j)
#pragma unroll
        for (int mb = 0; mb < MBK; ++mb) acc[j][mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[kk][j], xf[kk][mb], acc[j][mb], 0, 0, 0);
    if constexpr (LORA) {      // the LoRA-down product rides the same K walk: 16 more MFMA rows, D[r][m] = sum_k Adown[r,k] x[m,k]
      const int tgrp = KG > 1 ? kcur / p.group_k : 0;          // (wave-uniform; a 64-column step never straddles a group: group_k % 64 == 0)
#pragma unroll
      for (int tg = 0; tg < TG; ++tg)
        if (tg == tgrp) {
#pragma unroll
          for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int mb = 0; mb < MBK; ++mb) tacc[tg][mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[kk], xf[kk][mb], tacc[tg][mb], 0, 0, 0);
        }
    }
    if constexpr (LN) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int mb = 0; mb < MBK; ++mb) ln_frag_stats(xf[kk][mb], ls1[mb], ls2[mb]);
    }
    if (!refill_first && i + R < nsteps) {
      const int kn = issue(i + R, slot);
      if (slot == 0) kq[0] = kn; else kq[R - 1] = kn;
    }
    slot = slot + 1 == R ? 0 : slot + 1;
  }

  }   // !WP

  // ---- the 4 partial tiles meet in LDS; unit (row block mb, column block j); wave w finishes row block w (all JN column blocks): its adapter tile T_w never
  // leaves the wave's registers and the row partials of ln_parts are finished inside the wave - two barriers and two LDS round trips less than the
  // round-3 assignment (units w, w + 4, ... per wave)
  constexpr int UNITS = MBK * JN, TUN = LORA ? TG * MBK : 0, SMU = UNITS + TUN, UALL = UNITS + TUN + (LN ? 1 : 0);
  WTR(8);
  // next-weight prefetch: the touches go out behind this wave's last operand load (nothing of the K walk waits for them) and are waited for at the very end of the kernel, beside
  // the output stores; ONE destination register kept alive until then ("+v": hipcc must not hand it to anything else while loads into it are in flight)
  uint32_t pf_tmp = 0;
  if (p.pf_ptr) {
    const int xcdp = blockIdx.x & 7, idxp = blockIdx.x >> 3, nxp = gridDim.x >> 3;
    const int pcp = p.pf_ntn >> 2, lpt = p.pf_head_bytes >> 7, totalp = pcp * lpt;
    for (int l = idxp * (64 * NW) + (int)threadIdx.x; l < totalp; l += nxp * (64 * NW)) {
      const int t = div_small(l, lpt), o = l - t * lpt;
      const char* a = p.pf_ptr + (int64_t)((xcdp & 3) * pcp + t) * p.pf_tile_bytes + (int64_t)o * 128;
      asm volatile("global_load_dword %0, %1, off" : "+v"(pf_tmp) : "v"(a) : "memory");
    }
  }
  if constexpr (LORA && LATE) {
    load_bup();
#pragma unroll
    for (int q = 0; q < JN; ++q) cspre[q] = *(const f32x4*)(cssrc + 16 * q);
  }
  __syncthreads();
  WTR(9);
  f32x4* red = (f32x4*)smem;                                  // [NW][UALL][64 lanes]
#pragma unroll
  for (int j = 0; j < JN; ++j)
#pragma unroll
    for (int mb = 0; mb < MBK; ++mb) red[(wave * UALL + mb * JN + j) * 64 + lane] = acc[j][mb];
  if constexpr (LORA) {
#pragma unroll
    for (int tg = 0; tg < TG; ++tg)
#pragma unroll
      for (int mb = 0; mb < MBK; ++mb) red[(wave * UALL + UNITS + tg * MBK + mb) * 64 + lane] = tacc[tg][mb];
  }
  if constexpr (LN) {
#pragma unroll
    for (int mb = 0; mb < MBK; ++mb) {
      const float a = ln_sum_fk(ls1[mb]), b = ln_sum_fk(ls2[mb]);       // over the lane's three partners of the same row
      if (g == 0) ((float2*)(red + (wave * UALL + SMU) * 64))[mb * 16 + r] = make_float2(a, b);   // one unit: [MBK * 16 rows] (sum x, sum x^2)
    }
  }
  const int ngrp = KG > 1 ? p.K / p.group_k : TG;           // groups in use (<= TG; the unused accumulators stay zero)
  __syncthreads();
  WTR(12);
  const int mb = wave;                                       // this wave's row block; lane (r, g): row m, columns 16 j + 4 g .. + 3 of unit j
  const int m = m0 + mb * 16 + r;
  // LN: (mean, rstd) of row (mb, r) from the four waves' partial sums, added in wave order
  float mean = 0.f, rstd = 1.f;
  if constexpr (LN) {
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const float2 ps = ((const float2*)(red + (w * UALL + SMU) * 64))[mb * 16 + r];
      s1 += ps.x; s2 += ps.y;
    }
    const float inv = 1.f / (float)p.K;
    mean = s1 * inv;
    const float var = s2 * inv - mean * mean;
    rstd = rsqrtf((var > 0.f ? var : 0.f) + p.ln_eps);
    if (p.ln_stats && tn == 0 && g == 0) *(float2*)(p.ln_stats + (int64_t)m * 2) = make_float2(mean, rstd);
  }
  // T_g = s * X_g Adown_g^T of this row block, summed over the waves and rounded to bf16: the accumulator layout D[rank][m] (lane: row m, ranks 4g..4g+3) IS the
  // B-operand layout of the 16x16x16 MFMA, so the LoRA-up below takes it as it is
  [[maybe_unused]] uint2 tb[TG];
  if constexpr (LORA) {
#pragma unroll
    for (int tg = 0; tg < TG; ++tg) {
      tb[tg] = make_uint2(0u, 0u);
      if (tg < ngrp) {
        f32x4 t = red[(UNITS + tg * MBK + mb) * 64 + lane];
#pragma unroll
        for (int w = 1; w < NW; ++w) t += red[(w * UALL + UNITS + tg * MBK + mb) * 64 + lane];
        if constexpr (LN) {
#pragma unroll
          for (int i = 0; i < 4; ++i) t[i] = rstd * (t[i] - mean * lnca[tg][i]) + lnab[tg][i];
        }
        tb[tg] = make_uint2(pack2bf(t[0] * p.lora_scale, t[1] * p.lora_scale), pack2bf(t[2] * p.lora_scale, t[3] * p.lora_scale));
        if (p.T_out && tn == 0) *(uint2*)(p.T_out + (int64_t)m * p.ld_t + 16 * tg + 4 * g) = tb[tg];
      }
    }
  }
  WTR(10);
  float ps4[JN], pm2[JN];     // ln_parts: (sum, centred sum of squares) of this lane's four columns of every unit
  [[maybe_unused]] float dq4[JN];      // dotD: rounded output . O over this lane's four columns of every unit
#pragma unroll
  for (int q = 0; q < JN; ++q) {
    f32x4 v = red[(mb * JN + q) * 64 + lane];
#pragma unroll
    for (int w = 1; w < NW; ++w) v += red[(w * UALL + mb * JN + q) * 64 + lane];
    if constexpr (LN) v = (v - c1pre[q] * mean) * rstd;
    if constexpr (LORA) {
#pragma unroll
      for (int tg = 0; tg < TG; ++tg) {
        if (tg >= ngrp) break;
        v = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, bupf[q][tg]), __builtin_bit_cast(s16x4, tb[tg]), v, 0, 0, 0);
      }
    }
    if constexpr (LORA) {
      if (p.col_scale) v *= cspre[q];
    }
    if (p.bias) v += bpre[q];
    if constexpr (CONV) {
      if (p.rowbias) {
        const uint2 rv = rbpre[q];
        v[0] += bf2f(rv.x & 0xffff); v[1] += bf2f(rv.x >> 16); v[2] += bf2f(rv.y & 0xffff); v[3] += bf2f(rv.y >> 16);
      }
    }
    if constexpr (LORA) {
      if (p.Y0) *(uint2*)(p.Y0 + (int64_t)m * p.ldy0 + n0 + 16 * q + 4 * g) = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
    }
    if (p.R && !p.dotD) {
      const uint2 rv = rpre[q];
      v[0] += bf2f(rv.x & 0xffff); v[1] += bf2f(rv.x >> 16); v[2] += bf2f(rv.y & 0xffff); v[3] += bf2f(rv.y >> 16);
    }
    const uint2 ov = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
    *(uint2*)(p.Y + (int64_t)m * p.ldy + n0 + 16 * q + 4 * g) = ov;
    if constexpr (!LN && !CONV) {
      const uint2 rv = rpre[q];
      dq4[q] = (bf2f(ov.x & 0xffff) * bf2f(rv.x & 0xffff) + bf2f(ov.x >> 16) * bf2f(rv.x >> 16)) + (bf2f(ov.y & 0xffff) * bf2f(rv.y & 0xffff) + bf2f(ov.y >> 16) * bf2f(rv.y >> 16));
    }
    // (sum, CENTRED sum of squares) of the four ROUNDED columns: the slots of a row are merged below around the tile's row mean (the update of Chan et al.), so
    // the variance never comes out of a difference of two large numbers (rows whose mean dwarfs their spread)
    const float y0 = bf2f(ov.x & 0xffff), y1 = bf2f(ov.x >> 16), y2 = bf2f(ov.y & 0xffff), y3 = bf2f(ov.y >> 16);
    const float s4 = y0 + y1 + y2 + y3, m4 = 0.25f * s4;
    ps4[q] = s4;
    pm2[q] = (y0 - m4) * (y0 - m4) + (y1 - m4) * (y1 - m4) + (y2 - m4) * (y2 - m4) + (y3 - m4) * (y3 - m4);
  }
  if (p.ln_parts) {      // (wave-uniform) the row's 4 JN slots sit in the lanes r + 16 g of this wave: M2 = sum_slots M2_s + 4 (mean_s - mean)^2 around the tile's row mean
    float sx = 0.f;
#pragma unroll
    for (int q = 0; q < JN; ++q) sx += ps4[q];
    sx = ln_sum_fk(sx);
    const float mt = sx * (1.f / (16 * JN));
    float m2 = 0.f;
#pragma unroll
    for (int q = 0; q < JN; ++q) { const float d = ps4[q] * 0.25f - mt; m2 += pm2[q] + 4.f * d * d; }
    m2 = ln_sum_fk(m2);
    if (g == 0) p.ln_parts[(int64_t)m * ntn + tn] = make_float2(sx, m2);
  }
  if constexpr (!LN && !CONV) {
    if (p.dotD) {      // (wave-uniform) units 0 .. nA - 1 of this tile belong to head hA = n0 / 64, the rest to head hA + 1; the row's four lanes are summed like the LayerNorm partials
      const int hA = n0 >> 6, nA = ((hA + 1) * 64 - n0) >> 4;
      float sA = 0.f, sB = 0.f;
#pragma unroll
      for (int q = 0; q < JN; ++q) { sA += q < nA ? dq4[q] : 0.f; sB += q < nA ? 0.f : dq4[q]; }
      sA = ln_sum_fk(sA);
      sB = ln_sum_fk(sB);
      if (g == 0) {
        const int b = div_small(m, p.dot_nq), qrow = m - b * p.dot_nq, H = p.N >> 6;
        float* dst = p.dotD + ((int64_t)b * H + hA) * p.dot_nq + qrow;
        atomicAdd(dst, sA);
        if (nA < JN && hA + 1 < H) atomicAdd(dst + p.dot_nq, sB);
      }
    }
  }
  WTR(11);
  if (p.pf_ptr) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("" ::"v"(pf_tmp));
  }
}

template <int MBK, int JN, int R, int KG, bool LN = false, bool WP = false, bool CONV = false, int RG = 1>
int launch_wsk(const wsk_params& p, hipStream_t s) {
  constexpr bool LORA = KG > 0;
  constexpr int TGN = KG > 1 ? KG : (KG == 1 ? RG : 0);
  constexpr int SLOT = (16 * MBK + (WP ? 0 : 16 * JN) + (LORA ? 16 * RG : 0)) * ROWB;
  constexpr int RED = NW * (MBK * JN + TGN * MBK + (LN ? 1 : 0)) * 1024;          // the four partial tiles (+ adapter tiles, + row sums) of the reduction
  constexpr int smem = NW * R * SLOT > RED ? NW * R * SLOT : RED;
  static_assert(smem <= 160 * 1024, "LDS budget");
  if (sdlt_raise_smem((const void*)wsk_kernel<MBK, JN, R, KG, LN, WP, CONV, RG>, smem)) SDLT_FAIL(SDLT_ERR_LAUNCH, "sdlt_wsk_gemm: cannot raise the dynamic LDS limit to %d bytes", smem);
  const int tiles = (p.M / (16 * MBK)) * (p.N / (16 * JN));
  hipLaunchKernelGGL((wsk_kernel<MBK, JN, R, KG, LN, WP, CONV, RG>), dim3(tiles), dim3(64 * NW), smem, s, p);
  SDLT_CHECK_LAUNCH();
  return SDLT_OK;
}

// W [N, K] (row-major, leading dimension ldw) -> the fragment-major copy of wsk_params::Wp; one 16-byte chunk per thread
__global__ void wsk_pack_kernel(const bf16_t* __restrict__ W, int64_t ldw, int N, int K, bf16_t* __restrict__ Wp) {
  const int64_t nchunk = (int64_t)N * K / 8;
  for (int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; c < nchunk; c += (int64_t)gridDim.x * blockDim.x) {
    const int lane = (int)(c & 63);
    int64_t f = c >> 6;                       // fragment index: ((tn * (K / 64) + ks) * 2 + kk) * 5 + j
    const int j = (int)(f % 5); f /= 5;
    const int kk = (int)(f & 1); f >>= 1;
    const int nks = K >> 6;
    const int ks = (int)(f % nks), tn = (int)(f / nks);
    const int r = lane & 15, g = lane >> 4;
    *(uint4*)(Wp + c * 8) = *(const uint4*)(W + (int64_t)(tn * 80 + j * 16 + r) * ldw + ks * 64 + kk * 32 + g * 8);
  }
}

}  // namespace

extern "C" int sdlt_wsk_pack_weight(const void* W, int64_t ldw, int32_t N, int32_t K, void* Wp, void* stream) {
  if (N <= 0 || K <= 0 || (N % 80) || (K % 64)) SDLT_FAIL(SDLT_ERR_SHAPE, "sdlt_wsk_pack_weight: N=%d K=%d (N %% 80, K %% 64 == 0)", N, K);
  if (!W || !Wp || (ldw % 8) || ((uintptr_t)W & 15) || ((uintptr_t)Wp & 15)) SDLT_FAIL(SDLT_ERR_ALIGN, "sdlt_wsk_pack_weight: 16-byte aligned operands, ldw %% 8 == 0");
  const int64_t nchunk = (int64_t)N * K / 8;
  const int blocks = (int)((nchunk + 255) / 256 > 4096 ? 4096 : (nchunk + 255) / 256);
  hipLaunchKernelGGL(wsk_pack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)W, ldw, N, K, (bf16_t*)Wp);
  SDLT_CHECK_LAUNCH();
  return SDLT_OK;
}

#ifdef SDLT_WSK_TRACE
extern "C" int sdlt_wsk_trace_read(long long* out16) { return (int)hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_wsk_tr), sizeof(long long) * 16); }
#endif

static int wsk_gemm_impl(const void* X, int64_t ldx, const void* W, int64_t ldw, int32_t M, int32_t N, int32_t K, const float* bias,
                         const void* R, int64_t ldr, void* Y, int64_t ldy, const void* Adown, int64_t ld_adown, const void* Bup, int64_t ld_bup,
                         float lora_scale, void* T_out, int64_t ld_t, int32_t lora_group_k, const float* ln_c1, float* ln_stats, float ln_eps,
                         const float* ln_adapter, void* ln_parts, void* stream, float* dotD = nullptr, int32_t dot_nq = 0, int32_t lora_rp = 16,
                         const float* col_scale = nullptr, void* Y0 = nullptr, int64_t ldy0 = 0,
                         const void* pf_w = nullptr, int32_t pf_n = 0, int32_t pf_k = 0, int32_t pf_steps = 0) {
  if (M <= 0 || N <= 0 || K <= 0 || (M % 64) || (N % 80) || ((N / 80) % 8) || (K % 256))
    SDLT_FAIL(SDLT_ERR_SHAPE, "sdlt_wsk_gemm: M=%d N=%d K=%d (M %% 64, N %% 640, K %% 256 == 0)", M, N, K);
  const bool packed = ldw == 0;          // W is sdlt_wsk_pack_weight's output
  if (!X || !W || !Y || (ldx % 8) || (ldw % 8) || (ldy % 4) || ((uintptr_t)X & 15) || ((uintptr_t)W & 15) || ((uintptr_t)Y & 7) || (R && ((ldr % 4) || ((uintptr_t)R & 7))) ||
      (bias && ((uintptr_t)bias & 15)))
    SDLT_FAIL(SDLT_ERR_ALIGN, "sdlt_wsk_gemm: operand alignment");
  if (Adown && (!Bup || (ld_adown % 8) || ((uintptr_t)Adown & 15) || (ld_bup % 4) || ((uintptr_t)Bup & 7) || (T_out && ((ld_t % 4) || ((uintptr_t)T_out & 7)))))
    SDLT_FAIL(SDLT_ERR_ALIGN, "sdlt_wsk_gemm: adapter operands (Adown [16, K] 16-byte rows, Bup [N, 16] / T_out [M, 16] 8-byte rows)");
  if (lora_rp <= 0) lora_rp = 16;
  // (rank pad 64 - four groups - was built and dropped in round 6: with 64 T accumulators beside the 80 of the tile and the weight ring the kernel needs all 512 registers, hipcc
  // starts moving values between VGPRs and AGPRs around the hand-issued weight loads - whose destinations it believes to be ready at once - and the launch faults; the tiled
  // kernel keeps those products)
  if (Adown && lora_rp != 16 && (lora_rp != 32 || lora_group_k > 0 || ldw != 0))
    SDLT_FAIL(SDLT_ERR_UNSUPPORTED, "sdlt_wsk_gemm_p: lora_rp=%d (16; 32 with ONE adapter and a packed weight, ldw == 0)", lora_rp);
  if (Adown && lora_rp != 16 && ln_c1) SDLT_FAIL(SDLT_ERR_UNSUPPORTED, "sdlt_wsk_gemm_p: the folded LayerNorm exists for rank pad 16 only");
  if (col_scale && (!Adown || ((uintptr_t)col_scale & 15) || ln_c1)) SDLT_FAIL(SDLT_ERR_UNSUPPORTED, "sdlt_wsk_gemm_p: col_scale (DoRA) needs an adapter, 16-byte alignment and no folded LayerNorm");
  if (Y0 && (!Adown || (ldy0 % 4) || ((uintptr_t)Y0 & 7))) SDLT_FAIL(SDLT_ERR_UNSUPPORTED, "sdlt_wsk_gemm_p: Y0 (the output before the residual) exists for adapter launches; 8-byte aligned rows");
  if (ln_c1 && (((uintptr_t)ln_c1 & 15) || ((uintptr_t)ln_stats & 7) || lora_group_k > 0 || (Adown && (!ln_adapter || ((uintptr_t)ln_adapter & 15)))))
    SDLT_FAIL(SDLT_ERR_ALIGN, "sdlt_wsk_gemm_ln: c1 [N] / ln_adapter [32] 16-byte aligned, stats [M, 2] 8-byte aligned, no K-grouped adapters");
  static const int stagger_env = getenv("SDLT_WSK_STAGGER") ? atoi(getenv("SDLT_WSK_STAGGER")) : 1;   // (read once: A/B switch)
  wsk_params p{(const bf16_t*)X, ldx, (const bf16_t*)W, ldw, bias, (const bf16_t*)R, ldr, (bf16_t*)Y, ldy, M, N, K, 1,
               (const bf16_t*)Adown, ld_adown, (const bf16_t*)Bup, ld_bup, (bf16_t*)T_out, ld_t, lora_scale, lora_group_k, stagger_env,
               ln_c1, ln_stats, ln_adapter, ln_eps, (float2*)ln_parts, packed ? (const bf16_t*)W : nullptr, 0, 0, 0, 0, nullptr, nullptr, 0, dotD, dot_nq, col_scale, (bf16_t*)Y0, ldy0, nullptr, 0, 0, 0};
  if (pf_w && pf_steps > 0) {      // (a hint: anything that does not fit the scheme is ignored, never an error)
    if (pf_n > 0 && pf_k > 0 && (pf_n % 320) == 0 && (pf_k % 64) == 0 && !((uintptr_t)pf_w & 127)) {
      const int steps = pf_steps < (pf_k >> 6) ? pf_steps : (pf_k >> 6);
      p.pf_ptr = (const char*)pf_w; p.pf_tile_bytes = (pf_k >> 6) * 10240; p.pf_head_bytes = steps * 10240; p.pf_ntn = pf_n / 80;
    }
  }
  if (((uintptr_t)ln_parts) & 7) SDLT_FAIL(SDLT_ERR_ALIGN, "sdlt_wsk_gemm: ln_parts must be 8-byte aligned");
  if (dotD && (!R || ln_c1 || (N % 64) || dot_nq <= 0 || (M % dot_nq) || ((uintptr_t)dotD & 3)))
    SDLT_FAIL(SDLT_ERR_SHAPE, "sdlt_wsk_gemm_rowdot: O is required, N %% 64 == 0 (64-wide heads), M = B * Nq, no folded LayerNorm");
  hipStream_t s = (hipStream_t)stream;
  if (packed) {          // register ring of 3 stages (probed 2 / 3 / 4 in round 5: 3 wins on every shape; 4 runs out of registers with an adapter)
#define WSK_WP(KG_, LN_) launch_wsk<4, 5, 3, KG_, LN_, true>(p, s)
    if (Adown && lora_rp == 32) return launch_wsk<4, 5, 3, 1, false, true, false, 2>(p, s);
    if (ln_c1) return Adown ? WSK_WP(1, true) : WSK_WP(0, true);
    if (!Adown) return WSK_WP(0, false);
    if (lora_group_k <= 0) return WSK_WP(1, false);
    const int G = K / lora_group_k;
    if ((lora_group_k % 64) || G * lora_group_k != K || G < 2 || G > 3) SDLT_FAIL(SDLT_ERR_UNSUPPORTED, "sdlt_wsk_gemm: lora_group_k=%d with K=%d (2 or 3 groups of a multiple of 64 columns)", lora_group_k, K);
    return WSK_WP(3, false);
#undef WSK_WP
  }
  if (ln_c1) return Adown ? launch_wsk<4, 5, 2, 1, true>(p, s) : launch_wsk<4, 5, 2, 0, true>(p, s);
  if (!Adown) return launch_wsk<4, 5, 2, 0>(p, s);
  if (lora_group_k <= 0) return launch_wsk<4, 5, 2, 1>(p, s);
  const int G = K / lora_group_k;
  if ((lora_group_k % 64) || G * lora_group_k != K || G < 2 || G > 3) SDLT_FAIL(SDLT_ERR_UNSUPPORTED, "sdlt_wsk_gemm: lora_group_k=%d with K=%d (2 or 3 groups of a multiple of 64 columns)", lora_group_k, K);
  return launch_wsk<4, 5, 2, 3>(p, s);
}

extern "C" int sdlt_wsk_gemm(const void* X, int64_t ldx, const void* W, int64_t ldw, int32_t M, int32_t N, int32_t K, const float* bias,
                             const void* R, int64_t ldr, void* Y, int64_t ldy, const void* Adown, int64_t ld_adown, const void* Bup, int64_t ld_bup,
                             float lora_scale, void* T_out, int64_t ld_t, int32_t lora_group_k, void* stream) {
  return wsk_gemm_impl(X, ldx, W, ldw, M, N, K, bias, R, ldr, Y, ldy, Adown, ld_adown, Bup, ld_bup, lora_scale, T_out, ld_t, lora_group_k,
                       nullptr, nullptr, 0.f, nullptr, nullptr, stream);
}

extern "C" int sdlt_wsk_gemm_rowdot(const void* X, int64_t ldx, const void* W, int64_t ldw, int32_t M, int32_t N, int32_t K, const float* bias,
                                    const void* O, int64_t ldo, void* Y, int64_t ldy, const void* Adown, int64_t ld_adown, const void* Bup, int64_t ld_bup,
                                    float lora_scale, void* T_out, int64_t ld_t, int32_t lora_group_k, float* D, int32_t Nq, void* stream) {
  if (!D) SDLT_FAIL(SDLT_ERR_SHAPE, "sdlt_wsk_gemm_rowdot: D == NULL");
  return wsk_gemm_impl(X, ldx, W, ldw, M, N, K, bias, O, ldo, Y, ldy, Adown, ld_adown, Bup, ld_bup, lora_scale, T_out, ld_t, lora_group_k,
                       nullptr, nullptr, 0.f, nullptr, nullptr, stream, D, Nq);
}

extern "C" int sdlt_wsk_gemm_parts(const void* X, int64_t ldx, const void* W, int64_t ldw, int32_t M, int32_t N, int32_t K, const float* bias,
                                   const void* R, int64_t ldr, void* Y, int64_t ldy, const void* Adown, int64_t ld_adown, const void* Bup, int64_t ld_bup,
                                   float lora_scale, void* T_out, int64_t ld_t, int32_t lora_group_k, void* ln_parts, void* stream) {
  return wsk_gemm_impl(X, ldx, W, ldw, M, N, K, bias, R, ldr, Y, ldy, Adown, ld_adown, Bup, ld_bup, lora_scale, T_out, ld_t, lora_group_k,
                       nullptr, nullptr, 0.f, nullptr, ln_parts, stream);
}

extern "C" int sdlt_wsk_gemm_ln(const void* X, int64_t ldx, const void* W, int64_t ldw, int32_t M, int32_t N, int32_t K, const float* c2,
                                const void* R, int64_t ldr, void* Y, int64_t ldy, const void* Adown, int64_t ld_adown, const void* Bup, int64_t ld_bup,
                                float lora_scale, void* T_out, int64_t ld_t, const float* ln_c1, float* ln_stats, float ln_eps,
                                const float* ln_adapter, void* stream) {
  if (!ln_c1) SDLT_FAIL(SDLT_ERR_SHAPE, "sdlt_wsk_gemm_ln: c1 is required");
  return wsk_gemm_impl(X, ldx, W, ldw, M, N, K, c2, R, ldr, Y, ldy, Adown, ld_adown, Bup, ld_bup, lora_scale, T_out, ld_t, 0,
                       ln_c1, ln_stats, ln_eps, ln_adapter, nullptr, stream);
}

extern "C" int sdlt_wsk_gemm_p(const sdlt_wsk_gemm_params* q, void* stream) {
  if (!q) SDLT_FAIL(SDLT_ERR_SHAPE, "sdlt_wsk_gemm_p: params == NULL");
  if (q->dotD && (q->ln_c1 || q->ln_parts)) SDLT_FAIL(SDLT_ERR_UNSUPPORTED, "sdlt_wsk_gemm_p: the row-dot side output excludes the folded LayerNorm / row partials");
  return wsk_gemm_impl(q->X, q->ldx, q->W, q->ldw, q->M, q->N, q->K, q->bias, q->R, q->ldr, q->Y, q->ldy, q->Adown, q->ld_adown, q->Bup, q->ld_bup, q->lora_scale,
                       q->T_out, q->ld_t, q->lora_group_k, q->ln_c1, q->ln_stats, q->ln_eps, q->ln_adapter, q->ln_parts, stream, q->dotD, q->dot_nq, q->lora_rp, q->col_scale, q->Y0, q->ldy0, q->pf_next_w, q->pf_next_n, q->pf_next_k, q->pf_steps);
}

extern "C" int sdlt_wsk_conv(const void* X, int64_t ldx, const void* W, int64_t ldw, int32_t B, int32_t H, int32_t Wd, int32_t Cin, int32_t N, int32_t flip,
                             const float* bias, const void* rowbias, int64_t ld_rowbias, const void* R, int64_t ldr, void* Y, int64_t ldy,
                             const void* Adown, int64_t ld_adown, const void* Bup, int64_t ld_bup, float lora_scale, void* T_out, int64_t ld_t,
                             const void* zero, void* stream) {
  const int64_t M64 = (int64_t)B * H * Wd;
  const int32_t M = (int32_t)M64, K = 9 * Cin;
  if (B <= 0 || H <= 0 || Wd <= 0 || Cin <= 0 || N <= 0 || M64 >= (1ll << 22) || (M % 64) || (N % 80) || ((N / 80) % 8) || (Cin % 64) || K >= (1 << 22) || H >= 32768 || Wd >= 32768)
    SDLT_FAIL(SDLT_ERR_SHAPE, "sdlt_wsk_conv: B=%d H=%d W=%d Cin=%d N=%d (B H W %% 64, N %% 640, Cin %% 64 == 0)", B, H, Wd, Cin, N);
  const bool packed = ldw == 0;
  if (!X || !W || !Y || !zero || (ldx % 8) || (ldw % 8) || (ldy % 4) || ((uintptr_t)X & 15) || ((uintptr_t)W & 15) || ((uintptr_t)Y & 7) || ((uintptr_t)zero & 15) ||
      (R && ((ldr % 4) || ((uintptr_t)R & 7))) || (bias && ((uintptr_t)bias & 15)) || (rowbias && ((ld_rowbias % 4) || ((uintptr_t)rowbias & 7))))
    SDLT_FAIL(SDLT_ERR_ALIGN, "sdlt_wsk_conv: operand alignment (zero: a >= 128-byte zero page)");
  if (Adown && (!Bup || (ld_adown % 8) || ((uintptr_t)Adown & 15) || (ld_bup % 4) || ((uintptr_t)Bup & 7) || (T_out && ((ld_t % 4) || ((uintptr_t)T_out & 7)))))
    SDLT_FAIL(SDLT_ERR_ALIGN, "sdlt_wsk_conv: adapter operands (Adown [16, 9 Cin] 16-byte rows, Bup [N, 16] / T_out [M, 16] 8-byte rows)");
  static const int stagger_env = getenv("SDLT_WSK_STAGGER") ? atoi(getenv("SDLT_WSK_STAGGER")) : 1;
  wsk_params p{(const bf16_t*)X, ldx, (const bf16_t*)W, ldw, bias, (const bf16_t*)R, ldr, (bf16_t*)Y, ldy, M, N, K, 1,
               (const bf16_t*)Adown, ld_adown, (const bf16_t*)Bup, ld_bup, (bf16_t*)T_out, ld_t, lora_scale, 0, stagger_env,
               nullptr, nullptr, nullptr, 0.f, nullptr, packed ? (const bf16_t*)W : nullptr, H, Wd, Cin, flip ? 1 : 0, (const bf16_t*)zero, (const bf16_t*)rowbias, ld_rowbias};
  hipStream_t s = (hipStream_t)stream;
  if (packed) return Adown ? launch_wsk<4, 5, 3, 1, false, true, true>(p, s) : launch_wsk<4, 5, 3, 0, false, true, true>(p, s);
  return Adown ? launch_wsk<4, 5, 2, 1, false, false, true>(p, s) : launch_wsk<4, 5, 2, 0, false, false, true>(p, s);
}
