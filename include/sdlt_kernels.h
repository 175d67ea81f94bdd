/* C-ABI of the MI355X (gfx950) LoRA / textual-inversion training-step kernels.
 *
 * The reference (edenartlab/sd-lora-trainer) is pure Python and has no FFI; the seams this library
 * sits behind are the duck-typed third-party interfaces listed in SURVEY.md 8b.  Each entry point
 * below names the reference call site whose arithmetic it replaces (paths under /root/reference).
 * INTEGRATION.md shows the ctypes binding a maintainer of the reference would add.
 *
 * Conventions: every pointer is a DEVICE pointer owned by the caller (PyTorch-ROCm on the host side:
 * tensor.data_ptr()); kernels are enqueued on `stream` (a hipStream_t passed as void*), never allocate,
 * never synchronise; return 0 on success or a negative SDLT_ERR_* code, with a human-readable message
 * available from sdlt_last_error().  bf16 tensors are passed as raw 16-bit storage.  Activations are
 * NHWC / token-major: a [B,C,H,W] feature map is the row-major matrix [B*H*W, C].
 */
#ifndef SDLT_KERNELS_H
#define SDLT_KERNELS_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

const char* sdlt_last_error(void);
int sdlt_abi_version(void);
int sdlt_struct_size(int which); /* 0 gemm, 1 lora_grad_desc, 2 attn, 3 groupnorm, 4 shadow_desc, 5 gemm_batch_item, 6 dora_desc, 7 dora_wt_desc, 8 dora_grad_desc, 9 splitsum_desc */

/* ------------------------------------------------------------------------------------------------
 * sdlt_gemm_bf16 : C = alpha*(X.W^T [+ X2.W2^T] [+ s*(X.Adown^T).Bup^T]) + bias + rowbias + R
 * Replaces: every nn.Linear / nn.Conv2d (3x3, 1x1) of the UNet forward and their dX backward
 * (main.py:329-336 -> diffusers UNet2DConditionModel), with the peft LoRA adapter of
 * trainer/optimizer.py:84-95 fused in (y = base(x) + (alpha/r) * B(A(x))).
 * mode 0: X is a plain row-major [M,K] matrix (ldx).
 * mode 1: implicit 3x3 convolution: X is an NHWC activation [B,Hin,Win,(ldx)], K = 9*Cin with
 *         k = tap*Cin + ci, M = B*Hout*Wout.  stride in {1,2}; ups=2 reads a virtually nearest-2x
 *         upsampled input; flip=1 mirrors the taps (dX of a stride-1 conv); tr=1 is the transposed
 *         stride-2 form (dX of a stride-2 conv).  `zero` = >=128 B of zeros (out-of-image taps).
 * K, K2 multiples of 64; lora_R (padded rank) in {0,16,32,64}; Adown [lora_R,K], Bup [N,lora_R].
 * T_out (optional) receives s*X.Adown^T as bf16 [M,lora_R] (needed by the LoRA weight gradients).
 * tile: 0 = auto, 1 = 128x128 (8 waves), 2 = 64x128 (8 waves), 3 = 64x64 (4 waves), 4 = 256x128 (8 waves),
 *       5 = 128x128 (4 waves), 6 = 256x256 (8 waves)   (rows x cols of C per workgroup).
 * splitk: small-M problems leave most of the 256 CUs idle; the K loop is then split over several workgroups per
 *         tile whose fp32 partials meet in ws_slab; the last arriver (ws_cnt ticket, agent-scope release/acquire)
 *         reduces them and runs the epilogue - no extra launch.
 */
/* One problem of a batched launch (sdlt_gemm_params.batch): same M, N, K, leading dimensions, LoRA rank and epilogue
 * options as the launch, its own operands.  NULL members fall back to the launch-wide pointer. */
typedef struct sdlt_gemm_batch_item {
  const void* X; const void* W;
  const void* Adown; const void* Bup; void* T_out;
  void* C; void* Ct;
  const float* bias;
  const float* col_scale;
} sdlt_gemm_batch_item;

typedef struct sdlt_gemm_params {
  const void* X; int64_t ldx;
  const void* W; int64_t ldw;
  int32_t M, N, K;
  const void* X2; int64_t ldx2;
  const void* W2; int64_t ldw2;
  int32_t K2;
  int32_t mode;
  int32_t Hin, Win, Cin, Hout, Wout, stride, ups, flip, tr;
  const void* zero;
  const void* Adown; int64_t ld_adown;
  const void* Bup; int64_t ld_bup;
  void* T_out; int64_t ld_t;
  int32_t lora_R;
  float lora_scale;
  float alpha;
  const float* bias;                 /* fp32 [N] or NULL */
  const void* rowbias; int64_t ld_rowbias; int32_t rows_per_batch;
  const void* R; int64_t ldr;
  void* C; int64_t ldc;
  int32_t out_fp32;
  int32_t tile;
  void* Ct; int64_t ldct;     /* optional transposed bf16 copy: Ct[n*ldct + m] */
  int32_t splitk;             /* 0 = auto, 1 = off, n = split the K loop over n workgroups per tile */
  int32_t ws_cnt_len;         /* ints in ws_cnt */
  void* ws_slab; int64_t ws_slab_bytes;   /* split-K scratch: fp32 partial tiles (caller-owned, any contents) */
  int32_t stages;             /* LDS ring depth: 0 = auto (deepest that fits), 2 = double buffer (2 workgroups per CU) */
  int32_t accumulate;         /* fp32 output only: C += result (DAAM score sums over layers) */
  int32_t* ws_cnt;            /* split-K arrival counters, zero-initialised ONCE by the caller; kernels re-arm them */
  int32_t lora_group_n;       /* > 0: N is a concatenation of projections, `lora_group_n` columns each, every one with its
                                 own adapter: group g = n / lora_group_n uses Adown rows [g*lora_R, (g+1)*lora_R) and writes
                                 T_out columns [g*lora_R, (g+1)*lora_R); Bup stays [N, lora_R].  Fused to_q|to_k|to_v and
                                 the batched cross-attention to_k|to_v of all blocks.  Must be a multiple of the N tile. */
  int32_t lora_group_k;       /* > 0: K is a concatenation of G <= 4 groups of `lora_group_k` columns (the stacked dY of fused
                                 projections), each with its own adapter of padded rank lora_R: Adown stays [lora_R, K] (the groups' B^T
                                 side by side), T_out is [M, G*lora_R] and Bup [N, G*lora_R].  Mode 0; split-K only as one split per group. */
  const sdlt_gemm_batch_item* batch;   /* device array of n_batch problems sharing this launch (the to_k|to_v projections of
                                          every cross-attention layer read the same text conditioning: one launch for all of
                                          them, and one for all their input gradients); no split-K */
  int32_t n_batch;
  int32_t throughput_hint;            /* 1: several independent jobs share the device (train_concurrent): the other jobs fill idle CUs,
                                         so the tile heuristics trade workgroup count for per-tile efficiency (256x128 tiles
                                         for the >= 320-tile classes and the mid-size convs).  0: one job - fill the chip. */
  /* GEGLU of the transformer feed-forward (diffusers GEGLU: ff.net.0.proj -> hidden * gelu(gate)) fused into the GEMMs on either
     side of it.  Both use the INTERLEAVED-16 layout of the [M, 2H] projection output F1: hidden column j lives at
     (j / 16) * 32 + j % 16, its gate 16 columns further (the rows of ff.net.0.proj's weight are permuted accordingly at load), so
     that a hidden value and its gate sit in the same output tile.
       epi_op = 1 (forward, this GEMM IS ff.net.0.proj, N = 2H): C = F1 as usual, and epi_out [M, H] bf16 = hidden * gelu(gate).
       epi_op = 2 (backward, this GEMM is the dX of ff.net.2, N = H): with dG the product that would have gone to C,
                  epi_out [M, 2H] bf16 (interleaved) = d F1 = [dG * gelu(gate) | dG * hidden * gelu'(gate)], hidden / gate read
                  from epi_in = F1 [M, 2H]; C is not written (may be NULL).
       epi_op = 3 (activation side output, the CLIP MLP's fc1): C = pre-activation as usual, epi_out [M, N] bf16 = act(C).
       epi_op = 4 (activation backward, the dX of fc2): C = product * act'(epi_in), epi_in [M, N] bf16 = the forward pre-activation.
       epi_act: 0 = gelu (erf), 1 = quick_gelu x*sigmoid(1.702x)  (ops 3, 4).
     bf16 outputs only, no Ct, N % 32 == 0 (op 1) / N % 16 == 0 (op 2) / N % 8 == 0 (ops 3, 4). */
  int32_t epi_op; int32_t epi_act;
  void* epi_out; int64_t ld_epi_out;
  const void* epi_in; int64_t ld_epi_in;
  /* DoRA (peft LoraConfig(use_dora=True), trainer/optimizer.py:86-95): fp32 [N] or NULL; the product INCLUDING the adapter term is
     multiplied per output column before bias / row bias / residual:  C = alpha * col_scale[n] * (X W^T + s X A^T B^T) + bias ...
     (col_scale = magnitude / || W + s B A ||_row, kept up to date by sdlt_dora_refresh).  LoRA launches (lora_R > 0) only. */
  const float* col_scale;
  /* LayerNorm folded into the product (ln_c1 != NULL): the LayerNorm in front of attn1's to_q|to_k|to_v, attn2.to_q and ff.net.0.proj of a
     BasicTransformerBlock (norm1 / norm2 / norm3, reached from main.py:329-336) never runs as a launch of its own.  X holds the RAW rows,
     W = W o gamma (bf16), bias = c2 = W beta + bias, ln_c1 fp32 [N] = rowsum(W o gamma) taken from the rounded operand, and
       C = rstd[m] (X W^T - mean[m] c1[n]) + c2[n]  (+ adapter, row bias, residual; alpha must be 1);
     (mean, rstd) of every row come out of the same K walk (two more MFMAs per X fragment: ones . x -> row sums, x . x^T -> its diagonal =
     row sums of squares) and are written to ln_stats fp32 [M, 2] (or NULL) in sdlt_layernorm_fwd's layout, for sdlt_layernorm_bwd[_y].
     With an adapter (lora_R == 16 only): Adown = A o gamma and T = s (rstd (X Adown^T - mean cA) + abeta); ln_adapter fp32 [G][32] =
     cA[16] | abeta[16] per adapter group (sdlt_ln_fold_adapters keeps Adown and the constants up to date).  Mode 0, no second K segment,
     no split-K, no batch, epi_op 0 or 1. */
  const float* ln_c1;
  float* ln_stats;
  const float* ln_adapter;
  float ln_eps;
  /* ln_parts (optional, with ln_nparts even, 2..16): fp32 [M, ln_nparts, 2] = (sum x, sum (x - tile mean)^2) of every row per column tile
     (K / ln_nparts columns each) of the launch that PRODUCED X (sdlt_wsk_gemm_parts) - the statistics are then merged from a row's partials
     (M2 = sum_t M2_t + n_t (mean_t - mean)^2: as stable as a two-pass variance) and the K walk carries no extra work.  Used where a
     kernel variant for it exists (the shapes of the 1280-wide blocks); otherwise ignored and the statistics are computed from the K walk. */
  int32_t ln_nparts;
  const void* ln_parts;
} sdlt_gemm_params;
int sdlt_gemm_bf16(const sdlt_gemm_params* p, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Optional transposed copy of C written by the sdlt_gemm_bf16 epilogue: set via sdlt_gemm_params.Ct
 * (bf16 [N, ldct], Ct[n][m] = C[m][n]).  The attention kernels contract over tokens and want their
 * operands token-contiguous; producing the transposed copy in the projection GEMM removes every
 * in-kernel transpose.
 */

/* ------------------------------------------------------------------------------------------------
 * sdlt_lora_grad_grouped : all LoRA weight gradients of one backward pass in ONE launch.
 *   problem i:  out[c][r] or out[r][c] (+)= sum_m P[m][c] * Q[m][r]        (fp32 out, bf16 in)
 *   dBup  = (dY)^T . (s*T)   : P = dY [M,N],  Q = s*T [M,Rp]  (T_out of the forward GEMM), out [N,R]
 *   dAdown= (s*U)^T . X      : P = X  [M,K],  Q = s*U [M,Rp]  (T_out of the dX GEMM),     out [R,K]
 * conv=1 gathers P rows like sdlt_gemm_bf16 mode 1 (forward geometry), for the 3x3 LoRA-down conv.
 * Replaces autograd's per-adapter dA/dB matmuls behind loss.backward() (main.py:363) for the
 * adapters created at trainer/optimizer.py:86-95.  One workgroup = one problem x sdlt_lora_grad_block_cols()
 * columns (64): block_desc[b] = problem of workgroup b, desc.first_block = its first workgroup.
 * mfma=1 selects the MFMA kernel (needs: ldp, ldq, Cw multiples of 8; 16-byte aligned P, Q; conv: Cin % 64 == 0;
 * desc.zero = a >= 512-byte zero page), mfma=0 the VALU kernel that takes any shape.
 */
typedef struct sdlt_lora_grad_desc {
  const void* P; int64_t ldp;
  const void* Q; int64_t ldq;
  float* out;
  int32_t M, Cw, R, Rp;          /* R = real rank (columns of Q used), Rp = padded rank in {16,32,64} */
  int32_t rank_major;            /* 1: out[r*Cw + c]   0: out[c*R + r] */
  int32_t accumulate;            /* 1: out += result (gradient accumulation) */
  int32_t conv, Hin, Win, Cin, Hout, Wout, stride;
  const void* zero;
  int32_t first_block;           /* id of this problem's first workgroup in the launch */
  int32_t pad_;
} sdlt_lora_grad_desc;
int sdlt_lora_grad_grouped(const sdlt_lora_grad_desc* descs_dev, const int32_t* block_desc_dev,
                           int32_t n_blocks, int32_t Rp, int32_t mfma, void* stream);
int32_t sdlt_lora_grad_block_cols(void);

/* ------------------------------------------------------------------------------------------------
 * sdlt_attn_fwd / sdlt_attn_bwd : multi-head softmax attention, flash style (no score matrix in HBM).
 * Replaces F.scaled_dot_product_attention in the reference's DAAMLossAttnProcessor2_0.__call__
 * (trainer/ti_cross_attn_loss.py:197-199), diffusers' AttnProcessor2_0 for attn1, and their backward.
 * Q,dO,O,dQ: [B*Nqp, ld] token-major, head h in columns [h*d,(h+1)*d); K,V,dK,dV: [B*Nkp, ld].
 * Nq/Nk = valid rows per batch, Nqp/Nkp = allocated rows per batch (multiples of 8; pad rows are
 * masked).  Kt/Vt/Qt/dOt are the transposed copies [H*d, B*N?p].  L = log-sum-exp [B,H,Nq] fp32
 * (written by fwd, read by bwd), D = rowsum(dO*O) [B,H,Nq] fp32 scratch (written by bwd).
 * qsplit>1 splits the dK/dV reduction over query ranges (cross-attention: few keys, many queries),
 * accumulating in dK32/dV32 fp32 [B*Nkp, ld32] scratch before the bf16 store.  causal: CLIP.
 * Cross-attention (qsplit > 1, !causal, Nkp <= 128, d <= 96) takes a single-pass kernel (D, dQ, dK, dV together):
 * there qsplit = workgroups along the queries and dK32/dV32 must hold qsplit slabs, [qsplit][B*Nkp][ld32] each.
 */
typedef struct sdlt_attn_params {
  const void* Q; int64_t ldq;
  const void* K; int64_t ldk;
  const void* V; int64_t ldv;
  const void* Kt; int64_t ldkt;
  const void* Vt; int64_t ldvt;
  const void* Qt; int64_t ldqt;
  const void* dOt; int64_t lddot;
  void* O; int64_t ldo;
  float* L;
  const void* dO; int64_t lddo;
  float* D;
  void* dQ; int64_t lddq;
  void* dK; int64_t lddk;
  void* dV; int64_t lddv;
  float* dK32; float* dV32; int64_t ld32;
  int32_t B, H, Nq, Nk, Nqp, Nkp, d;
  float scale;
  int32_t qsplit;
  int32_t causal;
  int32_t accumulate_dq;   /* single-pass cross-attention backward only: dQ += result, dK += result (the buffers already hold */
  int32_t accumulate_dk;   /* the gradient of the score side output, written by one batched GEMM for all hooked layers)    */
  int32_t defer_splitsum;  /* single-pass cross-attention backward only: leave the per-split partial dK / dV slabs in dK32 / dV32
                              (they must then be layer-owned, not scratch); sdlt_attn_splitsum_batch sums the slabs of ALL layers in
                              one launch before their consumer (the batched to_k|to_v input-gradient GEMM) */
  int32_t d_ready;         /* backward, self-attention: D already holds rowsum(dO o O) - sdlt_wsk_gemm_rowdot accumulated it while it produced dO - so the D
                              pre-pass launch is skipped.  Its slots must start from zero: sdlt_attn_fwd zeroes D when D != NULL (in the forward kernel's
                              epilogue on the 32-row path, with a memset node otherwise). */
} sdlt_attn_params;
int sdlt_attn_fwd(const sdlt_attn_params* p, void* stream);
int sdlt_attn_bwd(const sdlt_attn_params* p, void* stream);
/* out0 / out1 [B*Nkp, C] bf16 = sum over nsplit slabs of s0 / s1 (fp32 [nsplit][B*Nkp][ld32]) in slab order (rows with key >= Nk: zeros;
 * acc0: out0 += instead of =), for a table of layers: block_desc[b] = descriptor of block b, block_first[d] = first block of descriptor d,
 * a layer owns ceil(B*Nkp*C/2 / 256) blocks (capped at 64; the kernel strides). */
typedef struct sdlt_splitsum_desc {
  const float* s0; const float* s1; int64_t ld32;
  void* out0; int64_t ldo0; void* out1; int64_t ldo1;
  int32_t nsplit, B, Nk, Nkp, C, acc0, nblocks, pad_;
} sdlt_splitsum_desc;
int sdlt_attn_splitsum_batch(const sdlt_splitsum_desc* descs_dev, const int32_t* block_desc_dev, const int32_t* block_first_dev, int32_t n_blocks, void* stream);

/* ------------------------------------------------------------------------------------------------
 * sdlt_groupnorm_fwd / _bwd : GroupNorm(32 groups) [+ SiLU] over an NHWC activation, and its dX.
 * Replaces ResnetBlock2D.norm1/norm2 (+nonlinearity), Transformer2DModel.norm and conv_norm_out of
 * the diffusers UNet reached from main.py:329-336.  The input may be the channel concatenation of two
 * tensors (x1: C1 channels, x2: C-C1 channels) - the up-block skip concat is never materialised.
 * stats [B,32,2] fp32 (sum, sumsq) is written by fwd and consumed by bwd; bstats is bwd scratch.
 * bwd: dx = dres + dGN(dy) where dy is the gradient w.r.t. the (optionally SiLU'd) output.
 */
typedef struct sdlt_groupnorm_params {
  const void* x1; int64_t ldx1; int32_t C1;
  const void* x2; int64_t ldx2;
  int32_t B, HW, C;
  const float* gamma; const float* beta; float eps;
  int32_t silu;
  void* y; int64_t ldy;
  float* stats;
  const void* dy; int64_t lddy;
  const void* dres; int64_t lddres;
  void* dx; int64_t lddx;
  float* bstats;
  /* statistics are reduced in a FIXED order (bitwise reproducible; no float atomics, zero fills or fences): the statistics
     kernel leaves 64 partial sums per block in `ws`, the kernel that consumes them (apply / dX) adds the rows in its
     prologue.  ws is scratch, only live inside the call: ws_floats >= sdlt_groupnorm_ws_floats(); cnt / cnt_len: reserved. */
  float* ws; int64_t ws_floats;
  int32_t* cnt; int32_t cnt_len;
  int32_t pad_;
  /* bwd only, optional side output: the column sums over each image's pixels of dx (as stored, bf16-rounded) - the gradient of a per-image bias added in front of
     this norm (ResnetBlock2D: time_emb_proj(silu(emb)) in front of norm2; was sdlt_colsum over the stored gradient).  Every block leaves its partial sums in
     colsum_ws[(split * B + b) * C + c], split < nsplit = sdlt_groupnorm_ws_floats(B, HW, C) / (B * C): fp32 [nsplit][B][C], persistent until
     sdlt_colsum_finish_batch has added the splits (one launch for all norms of a step). */
  float* colsum_ws;
} sdlt_groupnorm_params;
int sdlt_groupnorm_ws_floats(int32_t B, int32_t HW, int32_t C);
/* out[i] = sum_{split < nsplit} ws[split * n + i] (fixed order: bitwise reproducible) for n_desc reductions in one launch; out32 (fp32) and / or out16 (bf16), max_n = the largest n. */
typedef struct sdlt_colsum_finish_desc {
  const float* ws; float* out32; void* out16;
  int32_t nsplit, n;
} sdlt_colsum_finish_desc;
int sdlt_colsum_finish_batch(const sdlt_colsum_finish_desc* descs_dev, int32_t n_desc, int32_t max_n, void* stream);
int sdlt_groupnorm_fwd(const sdlt_groupnorm_params* p, void* stream);
int sdlt_groupnorm_bwd(const sdlt_groupnorm_params* p, void* stream);

/* LayerNorm over the last dim of [M,C] and its dX (dx = dres + dLN(dy)); stats [M,2] = (mean, rstd).
 * Replaces BasicTransformerBlock.norm1/2/3 and the CLIP LayerNorms. */
int sdlt_layernorm_fwd(const void* x, int64_t ldx, int32_t M, int32_t C, const float* gamma, const float* beta,
                       float eps, void* y, int64_t ldy, float* stats, void* stream);
int sdlt_layernorm_bwd(const void* x, int64_t ldx, const void* dy, int64_t lddy, int32_t M, int32_t C,
                       const float* gamma, const float* stats, const void* dres, int64_t lddres, void* dx,
                       int64_t lddx, void* stream);

/* sdlt_layernorm_bwd that also writes the normalised rows y = xhat gamma + beta (bf16 [M, C]): the backward of a LayerNorm whose forward was
 * folded into its consumer GEMM (sdlt_gemm_params.ln_c1 / sdlt_wsk_gemm_ln) - the adapter-gradient launch (sdlt_lora_grad) reads y as its P rows. */
int sdlt_layernorm_bwd_y(const void* x, int64_t ldx, const void* dy, int64_t lddy, int32_t M, int32_t C, const float* gamma, const float* beta,
                         const float* stats, const void* dres, int64_t lddres, void* dx, int64_t lddx, void* y, int64_t ldy, void* stream);

/* Adapter operands behind a folded LayerNorm, refreshed after every optimizer step (one launch for all adapters; one wave per rank row):
 *   Ag [16, K] bf16 = A32 o gamma (rows >= rank zero), consts[r] = sum_k float(Ag[r,k]), consts[16 + r] = sum_k A32[r,k] beta[k].
 * K % 4 == 0, 16-byte aligned rows. */
typedef struct sdlt_ln_fold_desc {
  const float* A32; int64_t lda;     /* fp32 master [rank, K] */
  const float* gamma; const float* beta;
  void* Ag; int64_t ldag;
  float* consts;                     /* fp32 [32] */
  int32_t rank, K;
} sdlt_ln_fold_desc;
int sdlt_ln_fold_adapters(const sdlt_ln_fold_desc* descs, int32_t n, void* stream);

/* sdlt_layernorm_bwd with dy given as nslab fp32 slabs [nslab][M][lddy32] that are added in slab order (the partial outputs of a K-split
 * sdlt_strip_gemm: the split's reduction rides in this kernel's prologue). */
int sdlt_layernorm_bwd_slabs(const void* x, int64_t ldx, const float* dy32, int64_t lddy32, int32_t nslab, int32_t M, int32_t C,
                             const float* gamma, const float* stats, const void* dres, int64_t lddres, void* dx, int64_t lddx, void* stream);

/* ------------------------------------------------------------------------------------------------ full fine-tune
 * Weight gradients of the full-UNet fine-tune (main.py:144-149, train_configs/full_finetuning_example.json): every
 * nn.Linear / nn.Conv2d weight, bias and norm affine parameter of the UNet receives a gradient.
 * dW[N,K] = dY^T X contracts over tokens; both operands are first transposed into token-contiguous bf16 panels and the
 * product is an ordinary sdlt_gemm_bf16 call (X := dY^T [N,Mp], W := X^T [K,Mp], fp32 output [N,K]):
 *   sdlt_wgrad_transpose : out[c*ldo + m] = x[m*ldx + c]           (m < M; columns M..Mp-1 zero; Mp % 64 == 0);
 *                          colsum_acc (optional, fp32 [C]) += column sums of x - the bias gradient when x = dY, for free
 *   sdlt_wgrad_im2col_t  : transposed im2col of an NHWC activation for a 3x3 / pad-1 conv:
 *                          out[(tap*C + c)*ldo + m] = x[b, (oy*stride+ky-1)/ups, (ox*stride+kx-1)/ups, c] or 0 outside,
 *                          m = (b, oy, ox) over the conv OUTPUT grid (H*ups/stride x W*ups/stride), tap = ky*3 + kx.
 * d gamma / d beta of the norms (fp32 atomics; zeroed first unless `accumulate`, in which case the caller has zeroed them -
 * the trainer keeps all vector gradients in one region and clears it with a single launch per step):
 *   sdlt_layernorm_affine_grad : dgamma[c] = sum_m dy*xhat, dbeta[c] = sum_m dy     (stats of sdlt_layernorm_fwd)
 *   sdlt_groupnorm_affine_grad : same for GroupNorm(32) (+SiLU: dy is the gradient w.r.t. the activated output); reads
 *                                x1/x2, dy, stats, gamma, beta, eps, silu, B, HW, C of the params struct. */
int sdlt_wgrad_transpose(const void* x, int64_t ldx, int32_t M, int32_t C, void* out, int64_t ldo, int32_t Mp, float* colsum_acc,
                         void* stream);
int sdlt_wgrad_im2col_t(const void* x, int64_t ldx, int32_t B, int32_t H, int32_t W, int32_t C, int32_t stride, int32_t ups,
                        void* out, int64_t ldo, int32_t Mp, void* stream);
/* All norm layers of one shape in one launch (accumulating: the caller has zeroed dgamma / dbeta).  groupnorm = 0: LayerNorm
 * over [B*HW, C] rows (x2 unused); 1: GroupNorm(32)(+SiLU), x = concat(x1 [.., C1], x2 [.., C - C1]) or x1 alone (C1 = 0). */
typedef struct sdlt_affine_grad_item {
  const void* x1; const void* x2; const void* dy; const float* stats; const float* gamma; const float* beta; float* dgamma; float* dbeta;
} sdlt_affine_grad_item;
int sdlt_affine_grad_batch(const sdlt_affine_grad_item* items_dev, int32_t n, int32_t groupnorm, int64_t ldx1, int32_t C1, int64_t ldx2,
                           int64_t lddy, int32_t B, int32_t HW, int32_t C, float eps, int32_t silu, void* stream);

/* Batched forms: n problems of identical geometry, each with its own pointers (device array of items).  The trainer defers
 * the weight gradients of a backward pass to its end and issues all layers of one shape together: one panel launch per
 * operand and one batched sdlt_gemm_bf16 (sdlt_gemm_params.batch) instead of three launches per layer. */
typedef struct sdlt_wgrad_tr_item { const void* x; void* out; float* colsum; } sdlt_wgrad_tr_item;
int sdlt_wgrad_transpose_batch(const sdlt_wgrad_tr_item* items_dev, int32_t n, int64_t ldx, int32_t M, int32_t C, int64_t ldo, int32_t Mp,
                               void* stream);
int sdlt_wgrad_im2col_t_batch(const sdlt_wgrad_tr_item* items_dev, int32_t n, int64_t ldx, int32_t B, int32_t H, int32_t W, int32_t C,
                              int32_t stride, int32_t ups, int64_t ldo, int32_t Mp, void* stream);
int sdlt_layernorm_affine_grad(const void* x, int64_t ldx, const void* dy, int64_t lddy, int32_t M, int32_t C,
                               const float* stats, float* dgamma, float* dbeta, int32_t accumulate, void* stream);
int sdlt_groupnorm_affine_grad(const sdlt_groupnorm_params* p, float* dgamma, float* dbeta, int32_t accumulate, void* stream);

/* GEGLU (diffusers FeedForward, activation_fn="geglu"): in [M, 2*Ch] = (h | g), out = h * gelu(g). */
int sdlt_geglu_fwd(const void* in, int64_t ldin, int32_t M, int32_t Ch, void* out, int64_t ldout, void* stream);
int sdlt_geglu_bwd(const void* in, int64_t ldin, const void* dout, int64_t lddout, int32_t M, int32_t Ch, void* din,
                   int64_t lddin, void* stream);

/* Contiguous bf16 maps: op 0 silu, 1 dy*silu'(x), 2 x+dy, 3 gelu, 4 dy*gelu'(x), 5 quick_gelu, 6 dy*quick_gelu'(x). */
int sdlt_map_bf16(int32_t op, const void* x, const void* dy, void* y, int64_t n, void* stream);

/* diffusers Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0): out[r] = [cos(t_r f) | sin(t_r f)] (bf16). */
int sdlt_timestep_embedding(const float* t, int32_t rows, int32_t dim, void* out, int64_t ldo, void* stream);

/* DDPMScheduler.add_noise (main.py:326) fused with NCHW fp32 -> NHWC bf16 (channels padded with zeros to Cpad). */
int sdlt_add_noise_nhwc(const float* x0, const float* noise, const int64_t* timesteps, const float* alphas_cumprod,
                        int32_t B, int32_t C, int32_t HW, int32_t Cpad, void* out_nhwc, float* noisy_nchw, void* stream);

/* compute_diffusion_loss + compute_snr (trainer/loss.py:127-170, 83-106) forward AND d(loss)/d(pred).
 * pred: NHWC fp32 [B*HW, ldp]; noise/noisy/mask: NCHW fp32; dpred: NHWC bf16 [B*HW, Cpad]; sums: scratch of sums_floats >=
 * 2*B*(1 + min(64, ceil(C*HW/1024))) floats - [B,2] per-sample {mean e, mean mask} followed by the per-slice partial sums the
 * reduction combines in a fixed order. */
int sdlt_masked_mse_fwd_bwd(const float* pred, int64_t ldp, const float* noise, const float* noisy, const float* mask,
                            const int64_t* timesteps, const float* alphas_cumprod, int32_t B, int32_t C, int32_t HW,
                            int32_t Cpad, float snr_gamma, int32_t v_prediction, float loss_scale, float* sums,
                            int32_t sums_floats, float* loss_out, void* dpred, void* stream);

/* torch.optim.AdamW step (trainer/optimizer.py:18,113-150; stepped optimizer.py:270-275) over a flat fp32 arena,
 * with the L1 penalty of main.py:353-356 folded in as a subgradient.  hyper (device fp32[9]):
 * lr, beta1, beta2, eps, weight_decay, 1-beta1^t, 1-beta2^t, l1 coefficient, grad scale.  l1_sum (optional) <- sum|p|. */
int sdlt_adamw_fused(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper, float* l1_sum, void* stream);

/* One Prodigy step over a flat fp32 arena (third-party prodigyopt==1.0, selected by `unet_optimizer_type` / `ti_optimizer`
 * = "prodigy": trainer/optimizer.py:24-34 and :135-145; effective-lr read-out trainer/optimizer.py:206-234), with the L1
 * penalty of main.py:353-356 folded in as a subgradient like sdlt_adamw_fused.  All state stays on the device:
 *   p0 = parameters at the first step, m/v = exp_avg / exp_avg_sq, s = the Prodigy direction estimate   (fp32 [n] each)
 *   hyper (fp32[13]): lr, beta1, beta2, beta3, eps, weight_decay, d_coef, growth_rate, l1 coefficient, grad scale,
 *                     use_bias_correction, safeguard_warmup, decouple (flags as 0/1)
 *   state (fp32[9]):  d, d0, d_max, d_numerator, d_denom, d_hat, k, dlr of this step, 1 if the step was applied
 *   acc   (fp64[2]):  scratch for the two global sums (<g, p0 - p> and sum|s|)
 * A step with lr == 0 or all-zero gradients changes nothing (the reference returns before updating any state). */
int sdlt_prodigy_step(float* p, const float* g, const float* p0, float* m, float* v, float* s, int64_t n,
                      const float* hyper, float* state, double* acc, float* l1_sum, void* stream);

/* bf16 compute copies of an fp32 master arena (LoRA adapters; every UNet weight in the full fine-tune), in both
 * orientations (dst [rows, ld], dstT [cols, ldT]).  One workgroup per 64x64 tile of a tensor: block_desc[b] = descriptor of
 * block b, block_first[d] = first block of descriptor d, a tensor owns ceil(rows/64)*ceil(cols/64) consecutive blocks. */
typedef struct sdlt_shadow_desc {
  int64_t offset;      /* element offset of the tensor in the fp32 arena */
  int64_t src_ld;      /* row stride of the tensor inside the arena (elements) */
  int32_t rows, cols;
  void* dst; int64_t ld;
  void* dstT; int64_t ldT;
} sdlt_shadow_desc;
int sdlt_lora_shadow_refresh(const sdlt_shadow_desc* descs_dev, const int32_t* block_desc_dev, const int32_t* block_first_dev,
                             int32_t n_blocks, const float* arena, void* stream);
/* The same tiles with the AdamW step fused in (full fine-tune: one pass over all matrix parameters): p, m, v are updated in
 * place from g with the hyper row of sdlt_adamw_fused (its L1 coefficient is ignored), then converted as above.  The
 * descriptors must tile every element exactly once (tensors not covered keep their old value: step them with
 * sdlt_adamw_fused). */
int sdlt_adamw_shadow_refresh(const sdlt_shadow_desc* descs_dev, const int32_t* block_desc_dev, const int32_t* block_first_dev,
                              int32_t n_blocks, float* p, const float* g, float* m, float* v, const float* hyper, void* stream);

/* bitsandbytes 0.43.1 `AdamW8bit` (trainer/optimizer.py:19-21; unet_optimizer_type of train_configs/full_finetuning_example.json), restated from its published
 * blockwise 8-bit Adam [3P-unverified: bitsandbytes is absent from /root/reference and from this image]: the same tiles as sdlt_adamw_shadow_refresh with
 * the moments held as one byte per element (m8, v8: indices into the signed / unsigned "dynamic" code books; TILE-MAJOR, uint8 [n_blocks][64][64]: element (r, c) of tile b
 * of the descriptor table at 4096 b + 64 r + c, positions outside the tensor unused - whole 128-byte lines per wave whatever the tensor's row stride) times one fp32
 * absmax per block of 2048 elements.  A block is one half of a 64 x 64 tile (rows 0-31 | rows 32-63) - absmax: fp32 [n_blocks][4] = {m lo, m hi, v lo, v hi},
 * 16-byte aligned, zero before the first step (m8 / v8 zero too).  tables: fp32 [1024], 16-byte aligned = q1[256] | mid1[256] | q2[256] | mid2[256] with
 * q the sorted code book and mid[k] = (q[k] + q[k + 1]) / 2, mid[255] = +inf.  hyper as sdlt_adamw_fused.  Update order as bnb: p += step, then p *= 1 - lr wd. */
int sdlt_adamw8_shadow_refresh(const sdlt_shadow_desc* descs_dev, const int32_t* block_desc_dev, const int32_t* block_first_dev, int32_t n_blocks, float* p,
                               const float* g, uint8_t* m8, uint8_t* v8, float* absmax, const float* tables, const float* hyper, void* stream);

/* The AdamW8bit update of sdlt_adamw8_shadow_refresh on a flat range of n elements (n % 4 == 0): blocks of 2048 CONSECUTIVE elements counted from p - bitsandbytes' own
 * partition - with absmax fp32 [ceil(n / 2048)][2] = {m, v} (8-byte aligned, zero before the first step), no operand refresh.  The owned slices of the sharded optimizer
 * (data-parallel full fine-tune, ZeRO-1): their masters are all-gathered before the refresh. */
int sdlt_adamw8_flat(float* p, const float* g, uint8_t* m8, uint8_t* v8, float* absmax, int64_t n, const float* tables, const float* hyper, void* stream);

/* ------------------------------------------------------------------------------------------------ DoRA
 * Weight-decomposed adapters: peft `LoraConfig(use_dora=True)` as the reference requests it (trainer/optimizer.py:86-95; L1 penalty
 * and weight decay are switched off with it, config.py:153-157).  [3P-unverified: peft 0.10.0 LoraLayer._apply_dora / Conv2d]
 *     y = (m / ||W + s B A||_row) * (x W^T + s x A^T B^T) + bias          m [N] trained, the norm detached
 * The forward GEMM applies the factor through sdlt_gemm_params.col_scale; these three batched launches (device descriptor tables,
 * block_desc[b] = descriptor of block b, block_first[d] = first block of descriptor d) keep everything else in step:
 *   sdlt_dora_refresh   per adapted layer, ceil(N/64) blocks: scale[n] = mag[n] / ||W_n + s B_n A||  (init != 0: mag := the norm first,
 *                       peft's dora_init) and Bt = (B32 * scale)^T as bf16 [Rp, N] (the LoRA-down operand of the dX GEMM), K % 32 == 0
 *   sdlt_dora_scale_wt  per dX weight, ceil(rows/32)*ceil(cols/256) blocks: dst[r, c] = src[r, c] * scale[c % period]
 *                       (columns with c % period >= nvalid become 0)
 *   sdlt_dora_mag_grad  stage 1, per layer ceil(N/64)*splits blocks into ws (fp32 [splits, 2, N] at ws_off per layer); stage 2, per
 *                       layer ceil(N/256) blocks: gmag[n] = sum_rows dY * (Y - bias) / mag  and  gB[n, :] *= scale[n]   (Y = the
 *                       layer's output BEFORE any residual; dY, Y bf16 with N % 8 == 0 and 16-byte aligned rows) */
typedef struct sdlt_dora_desc {
  const void* W; int64_t ldw;        /* bf16 [N, K] forward operand (3x3 conv: tap-major [Cout, 9*Cin]) */
  const void* A; int64_t lda;        /* bf16 [Rp, K] LoRA-down compute copy (rows >= rank zero) */
  const void* B; int64_t ldb;        /* bf16 [N, Rp] LoRA-up compute copy (columns >= rank zero) */
  float* mag;                        /* fp32 [N] trained magnitude */
  float* scale;                      /* fp32 [N] out */
  void* Bt; int64_t ldbt;            /* bf16 [Rp, N] out, or NULL */
  const float* B32; int64_t ldb32;   /* fp32 master of B [N, rank] */
  int32_t N, K, Rp, rank;
  float s;                           /* lora_alpha / r */
  int32_t pad_;
} sdlt_dora_desc;
typedef struct sdlt_dora_wt_desc {
  const void* src; void* dst; int64_t ld;    /* bf16 [rows, cols], same leading dimension */
  const float* scale;
  int32_t rows, cols, period, nvalid;
} sdlt_dora_wt_desc;
typedef struct sdlt_dora_grad_desc {
  const void* dY; int64_t lddy;
  const void* Y; int64_t ldy;
  const float* bias;                 /* fp32 [N] or NULL */
  const float* mag; const float* scale;
  float* gmag;                       /* fp32 [N] out */
  float* gB;                         /* fp32 [N, rank], scaled in place */
  int64_t ws_off;
  int32_t M, N, rank, splits;
  float grad_scale;
  int32_t accumulate;                /* 0: gmag = the layer's gradient and gB rows *= scale; 1: gmag += it and gB is left alone - a SECOND pass through
                                        the same adapters (the tok_cond_reg_w captions), run in its own launch after the first pass's */
} sdlt_dora_grad_desc;
int sdlt_dora_refresh(const sdlt_dora_desc* descs_dev, const int32_t* block_desc_dev, const int32_t* block_first_dev, int32_t n_blocks,
                      int32_t Rp, int32_t init, void* stream);
int sdlt_dora_scale_wt(const sdlt_dora_wt_desc* descs_dev, const int32_t* block_desc_dev, const int32_t* block_first_dev, int32_t n_blocks,
                       void* stream);
int sdlt_dora_mag_grad(const sdlt_dora_grad_desc* descs_dev, const int32_t* block_desc_dev, const int32_t* block_first_dev, int32_t n_blocks,
                       const int32_t* fin_block_desc_dev, const int32_t* fin_block_first_dev, int32_t n_fin_blocks, float* ws, void* stream);

/* out[M,C] = a + b on strided 2-D bf16 views (gradient fan-in of the UNet skip connections). */
int sdlt_add2d(const void* a, int64_t lda, const void* b, int64_t ldb, void* out, int64_t ldo, int32_t M, int32_t C, void* stream);

/* ------------------------------------------------------------------------------------------------ text-encoder / TI
 * CLIP token + position embedding: out[b*Tp + t] = table[ids[b*T + t]] + pos[t] (t < T), rows T..Tp-1 zero.
 * Replaces CLIPTextEmbeddings.forward inside pipe.encode_prompt (trainer/inference.py:132-139).  bf16 tables. */
int sdlt_embed_gather(const void* table, int64_t ld_table, const int64_t* ids, const void* pos, int64_t ld_pos,
                      int32_t B, int32_t T, int32_t Tp, int32_t D, void* out, int64_t ldo, void* stream);
/* Gradient of the trainable token rows only: grad[j] (+)= sum over (b,t) with ids[b,t] == train_ids[j] of dx[b*Tp+t].
 * Equivalent to the reference's full-table gradient followed by `grad[:-n_tokens] *= 0` (main.py:368-371). */
int sdlt_embed_grad(const void* dx, int64_t lddx, const int64_t* ids, const int64_t* train_ids, int32_t n_train,
                    int32_t B, int32_t T, int32_t Tp, int32_t D, float* grad, int32_t accumulate, void* stream);
/* DistributionLoss.compute_std_loss (trainer/loss.py:291-297) on the trainable rows [n,D] (fp32) and its gradient:
 * loss_out += w * mean_j (target_mean - std(row_j))^2 / target_var ; grad += d/d rows.  std is the unbiased std. */
int sdlt_ti_std_reg(const float* rows, int32_t n, int32_t D, float target_mean, float target_var, float weight,
                    float* grad, float* loss_out, void* stream);

/* Row-strip GEMM of the text encoders:  Y[b*Tp + t, :] = X[b*Tp + t, :] . W^T  for the t < T (<= 80) valid rows of every batch element,
 * W [N, K] K-contiguous (forward: the weight, input gradient: its transposed copy), K % 256 == 0, N % 16 == 0.  A workgroup owns 16 (32 for
 * N >= 4096) output columns over the full K, its 8 waves split K; operands go from global memory straight into MFMA fragments (no LDS
 * staging) - the shape of a weight-streaming product with 77 rows.  Rows t >= T are neither read as results nor written.
 * Replaces every nn.Linear of CLIPTextModel / CLIPTextModelWithProjection inside pipe.encode_prompt (trainer/inference.py:131-177, called
 * with autograd from main.py:306-308) and its input gradient in loss.backward() (main.py:363), together with what surrounds it:
 *   ln = 1   : CLIPEncoderLayer.layer_norm1 / layer_norm2 in front of the product.  X holds the RAW rows, W holds W o gamma (bf16),
 *              c1[n] = sum_k (W o gamma)[n,k], c2[n] = sum_k beta[k] W[n,k] + bias[n];  Y = rstd (acc - mean c1) + c2 with the row
 *              statistics taken from the same fragments; stats [B*Tp, 2] = (mean, rstd) is written for sdlt_layernorm_bwd (optional).
 *   bias, R  : + bias[n] (ln = 0) and + residual R[row, n]
 *   Y2, act  : second output act(Y) (from the fp32 values): act 1 = quick_gelu (CLIP-L), 2 = gelu (OpenCLIP bigG)  - mlp.fc1
 *   Z, act   : Y = (acc + bias + R) * act'(Z[row, n])                                                              - input gradient of mlp.fc2 */
typedef struct sdlt_strip_params {
  const void* X; int64_t ldx;
  const void* W; int64_t ldw;
  const float* bias;
  const void* R; int64_t ldr;
  void* Y; int64_t ldy;
  void* Y2; int64_t ldy2;
  const void* Z; int64_t ldz;
  const float* c1; const float* c2;
  float* stats;
  int32_t B, T, Tp, N, K, act, ln;
  float eps;
  /* splitk = S > 1 (K % (256 S) == 0): S workgroups share a strip and walk K / S columns each; their fp32 tiles meet in `ws` and the last
     arriver adds them in split order and runs the epilogue (bitwise reproducible).  ws: scratch, only live inside the call, >= B * strips *
     S * 7680 bytes (strips = N / 16, or N / 32 with 15360); cnt: one zero-initialised int per (batch element, strip), re-armed by the kernel. */
  int32_t splitk, cnt_len;
  void* ws; int64_t ws_bytes;
  int32_t* cnt;
  /* P != NULL: partial output instead of Y - split s writes its fp32 tile to P[s][row][n] (P: [S][B*Tp][ldp] fp32; rows t >= T untouched) and
     the consumer adds the S slabs (sdlt_layernorm_bwd_slabs): the K split without an in-kernel seam.  No ln / bias / R / Y2 / Z. */
  float* P; int64_t ldp;
} sdlt_strip_params;
int sdlt_strip_gemm(const sdlt_strip_params* p, void* stream);

/* Paired launches (reference: none - launch structure only).  SDXL conditions on TWO text encoders (trainer/inference.py:131-177 through
 * diffusers' encode_prompt): layer i of CLIP-L and layer i of OpenCLIP-bigG are independent chains of launch-bound 77-row kernels, and a
 * launch costs more than CLIP-L's share of it.  Each *_pair entry runs problem a and problem b of the same kernel in ONE launch, with the
 * results of the two single calls (bit for bit: the per-workgroup arithmetic is the same code).
 *   sdlt_strip_gemm_pair: both with or both without a LayerNorm fold; no in-kernel K split (partial outputs are fine).
 *   sdlt_attn_fwd_pair / sdlt_attn_bwd_pair: self-attention (qsplit 1) with head width <= 64 and < 256 keys each (sdlt_attn_pair_ok).
 *   sdlt_layernorm_bwd_slabs_pair: two sdlt_layernorm_bwd_slabs problems. */
int sdlt_strip_gemm_pair(const sdlt_strip_params* a, const sdlt_strip_params* b, void* stream);
int sdlt_attn_pair_ok(const sdlt_attn_params* a, const sdlt_attn_params* b);
int sdlt_attn_fwd_pair(const sdlt_attn_params* a, const sdlt_attn_params* b, void* stream);
int sdlt_attn_bwd_pair(const sdlt_attn_params* a, const sdlt_attn_params* b, void* stream);
typedef struct sdlt_ln_slabs_params {
  const void* x; int64_t ldx;
  const float* dy32; int64_t lddy32;
  const float* gamma; const float* stats;
  const void* dres; int64_t lddres;
  void* dx; int64_t lddx;
  int32_t nslab, M, C, pad_;
} sdlt_ln_slabs_params;
int sdlt_layernorm_bwd_slabs_pair(const sdlt_ln_slabs_params* a, const sdlt_ln_slabs_params* b, void* stream);

/* Wave-split-K GEMM for the long-K, 1280-wide products of the batch-1 UNet:  Y[M,N] = X[M,K] . W[N,K]^T + bias[n] + R[m,n]  (bf16 in / out,
 * fp32 accumulation), M % 64 == 0, N % 640 == 0, K % 256 == 0.  64 x 80 tiles (exactly 256 workgroups for 1024 x 1280), the 4 waves of a
 * workgroup split K and stage their own operands through private LDS rings - no block barrier in the K loop; the 4 partial tiles are added in
 * wave order (bitwise reproducible).  Replaces FeedForward.net[2] (nn.Linear(4 C, C)) of the 1280-wide BasicTransformerBlocks (+ the block's
 * residual) reached from main.py:329-336, where the 128 x 128-tile kernel leaves two thirds of the CUs idle.
 * Adown != NULL: a rank-16 adapter rides along (peft LoRA on to_q / to_out.0 ..., trainer/optimizer.py:84-95) -
 *   Y += bf16(lora_scale * X Adown^T) . Bup^T,  Adown [16, K], Bup [N, 16] (bf16 shadows, rank padded with zeros); the LoRA-down product uses 16
 *   more MFMA rows of the same K walk, the LoRA-up is one 16x16x16 MFMA per output block in the epilogue; T_out [M, 16] (optional) receives
 *   bf16(lora_scale * X Adown^T), the operand of the adapter-gradient launch - the contract of sdlt_gemm_bf16's lora_R = 16 path.
 * lora_group_k > 0: K is 2 or 3 groups of lora_group_k columns with one adapter each (the input gradient of the stacked to_q|to_k|to_v):
 *   T_g = X[:, group g] . Adown[:, group g]^T, Y += sum_g bf16(s T_g) . Bup[:, 16 g ..]^T, Bup [N, 16 G], T_out [M, 16 G] - sdlt_gemm_bf16's
 *   lora_group_k contract. */
int sdlt_wsk_gemm(const void* X, int64_t ldx, const void* W, int64_t ldw, int32_t M, int32_t N, int32_t K, const float* bias,
                  const void* R, int64_t ldr, void* Y, int64_t ldy, const void* Adown, int64_t ld_adown, const void* Bup, int64_t ld_bup,
                  float lora_scale, void* T_out, int64_t ld_t, int32_t lora_group_k, void* stream);

/* sdlt_wsk_gemm whose output is the gradient dO of a self-attention with 64-wide heads (the dX of attn1.to_out.0; diffusers Attention under main.py:329-336):
 * next to Y it accumulates D[(b H + h) Nq + q] += sum over head h's columns of rounded(Y[m, n]) O[m, n] (H = N / 64, m = b Nq + q, O bf16 [M, N] = the forward's
 * attention output, NOT added to Y) - the row term of the softmax backward, which sdlt_attn_bwd otherwise computes in a pre-pass launch (sdlt_attn_params.d_ready).
 * D must be zero on entry (sdlt_attn_fwd with D set).  Float atomics, still bitwise reproducible: a slot receives at most two values (a head's 64 columns lie in
 * at most two 80-column tiles) and x + y == y + x.  N % 64 == 0, M % Nq == 0. */
int sdlt_wsk_gemm_rowdot(const void* X, int64_t ldx, const void* W, int64_t ldw, int32_t M, int32_t N, int32_t K, const float* bias,
                         const void* O, int64_t ldo, void* Y, int64_t ldy, const void* Adown, int64_t ld_adown, const void* Bup, int64_t ld_bup,
                         float lora_scale, void* T_out, int64_t ld_t, int32_t lora_group_k, float* D, int32_t Nq, void* stream);

/* FROZEN weights (the UNet under LoRA / textual inversion, main.py:329-336 with trainer/optimizer.py:84-95: only adapters train) can be handed to the three
 * sdlt_wsk_gemm* entry points in fragment-major order: Wp = [N / 80][K / 64][2][5][64 lanes][8 bf16], lane (r = lane % 16, g = lane / 16) of fragment
 * (kk, j) of block (tn, ks) holding W[80 tn + 16 j + r][64 ks + 32 kk + 8 g .. + 7] - the MFMA operand as it sits in registers.  Pass the packed
 * copy as W with ldw = 0: a wave's K step is then ten contiguous 1 KB loads straight into a register ring, the LDS rings carry the activation (and
 * LoRA-down) rows only, and the K walk no longer runs at the rate four waves can issue LDS-DMA pieces at.  Same arithmetic, same
 * summation order: results are bit-identical to the row-major call.  sdlt_wsk_pack_weight writes the copy (N % 80 == 0, K % 64 == 0, N K bf16). */
int sdlt_wsk_pack_weight(const void* W, int64_t ldw, int32_t N, int32_t K, void* Wp, void* stream);

/* 3 x 3 convolution (stride 1, padding 1) of the 32 x 32 level of the UNet on the wave-split-K kernel - the ResnetBlock2D conv1 / conv2 of the 1280-wide
 * stages reached from main.py:329-336 and their input gradients - as an implicit GEMM:  Y[B H W, N] = im2col(X) . W^T + bias[n] + rowbias[b, n] + R[m, n],
 * X the NHWC activation [B H W, Cin] (bf16, row stride ldx), W [N, 9 Cin] with k = tap Cin + ci, tap = 3 dy + dx (row-major, or sdlt_wsk_pack_weight's copy
 * with ldw = 0 for frozen weights), flip = 1: taps mirrored (the input gradient: W = the [Cin, 9 Cout] backward operand).  B H W % 64 == 0, N % 640 == 0,
 * Cin % 64 == 0; zero: a zero page of >= 128 bytes (the halo).  rowbias bf16 [B, N] or NULL: the per-image bias (time-embedding projection).  Adown [16, 9 Cin] /
 * Bup [N, 16] / T_out [M, 16]: the rank-16 adapter of sdlt_wsk_gemm (peft LoRA on conv2: a 3 x 3 LoRA-down convolution and a 1 x 1 up-projection,
 * trainer/optimizer.py:84-95).  One workgroup owns a whole 64 x 80 output tile and its four waves split K: no split-K partials travel through HBM (the tiled
 * kernel wrote 23.6 MB of fp32 slabs for a 2.6 MB output here), and the nine taps of a pixel block re-read the same rows from L2. */
int sdlt_wsk_conv(const void* X, int64_t ldx, const void* W, int64_t ldw, int32_t B, int32_t H, int32_t Wd, int32_t Cin, int32_t N, int32_t flip,
                  const float* bias, const void* rowbias, int64_t ld_rowbias, const void* R, int64_t ldr, void* Y, int64_t ldy,
                  const void* Adown, int64_t ld_adown, const void* Bup, int64_t ld_bup, float lora_scale, void* T_out, int64_t ld_t,
                  const void* zero, void* stream);

/* sdlt_wsk_gemm with the LayerNorm in front of the projection folded in (sdlt_gemm_params.ln_c1's contract; attn2.to_q of the 1280-wide blocks):
 * X raw rows, W = W o gamma, c2 = W beta + bias, Y = rstd (X W^T - mean c1) + c2 (+ adapter with Adown = A o gamma, ln_adapter = cA | abeta, + R);
 * ln_stats [M, 2] (or NULL) receives (mean, rstd). */
int sdlt_wsk_gemm_ln(const void* X, int64_t ldx, const void* W, int64_t ldw, int32_t M, int32_t N, int32_t K, const float* c2,
                     const void* R, int64_t ldr, void* Y, int64_t ldy, const void* Adown, int64_t ld_adown, const void* Bup, int64_t ld_bup,
                     float lora_scale, void* T_out, int64_t ld_t, const float* ln_c1, float* ln_stats, float ln_eps,
                     const float* ln_adapter, void* stream);

/* The wave-split-K product with every option as one parameter block - what sdlt_wsk_gemm / _rowdot / _ln / _parts take as arguments, plus the two that only
 * exist here:
 *   lora_rp   padded adapter rank: 16 (0 = 16), or 32 - the sweep's rank 24 (scripts/create_hyperparam_sweep.py:76) - with ONE adapter and a packed
 *             weight (ldw == 0), no folded LayerNorm: Adown [lora_rp, K], Bup [N, lora_rp], T_out [M, lora_rp];
 *   col_scale fp32 [N] or NULL: DoRA's column factor m / ||W + s B A|| (use_dora, trainer/optimizer.py:86-95) applied to product + adapter before the bias -
 *             sdlt_gemm_params.col_scale's contract; adapter launches only, not with a folded LayerNorm;
 *   Y0        bf16 [M, N] or NULL: the layer's own output BEFORE the residual, rounded(col_scale (X W^T + adapter) + bias), written next to Y (= that value in fp32 + R):
 *             DoRA's magnitude gradient reads it (sdlt_dora_mag_grad), and the product + residual add stay one launch.
 * R: residual [M, N], or (dotD != NULL) the attention output O of sdlt_wsk_gemm_rowdot, which is not added. */
typedef struct sdlt_wsk_gemm_params {
  const void* X; int64_t ldx;
  const void* W; int64_t ldw;
  const float* bias;
  const void* R; int64_t ldr;
  void* Y; int64_t ldy;
  const void* Adown; int64_t ld_adown;
  const void* Bup; int64_t ld_bup;
  void* T_out; int64_t ld_t;
  const float* col_scale;
  void* Y0; int64_t ldy0;
  const float* ln_c1; float* ln_stats; const float* ln_adapter;
  void* ln_parts;
  float* dotD;
  const void* pf_next_w;             /* hint (may be NULL): the PACKED weight [pf_next_n, pf_next_k] of the wave-split-K product that follows in the step - this launch */
  int32_t M, N, K;                   /* touches the first pf_steps K steps of it into the L2s that will read them (never an error if it does not fit the scheme) */
  int32_t lora_group_k, lora_rp, dot_nq;
  float lora_scale, ln_eps;
  int32_t pf_next_n, pf_next_k, pf_steps, pad_;
} sdlt_wsk_gemm_params;
int sdlt_wsk_gemm_p(const sdlt_wsk_gemm_params* p, void* stream);

/* sdlt_wsk_gemm that also leaves, for the LayerNorm that reads its output next, ln_parts fp32 [M, N / 80, 2] = (sum y, sum (y - tile mean)^2) of every ROUNDED
 * output row over each 80-column tile (to_out.0 + residual -> norm2 / norm3, ff.net.2 + residual -> the next block's norm1): the consumer
 * (sdlt_gemm_params.ln_parts) adds N / 80 partials per row instead of reducing the row in its K walk. */
int sdlt_wsk_gemm_parts(const void* X, int64_t ldx, const void* W, int64_t ldw, int32_t M, int32_t N, int32_t K, const float* bias,
                        const void* R, int64_t ldr, void* Y, int64_t ldy, const void* Adown, int64_t ld_adown, const void* Bup, int64_t ld_bup,
                        float lora_scale, void* T_out, int64_t ld_t, int32_t lora_group_k, void* ln_parts, void* stream);

/* Token-attention (DAAM) loss and its gradient w.r.t. the hooked cross-attention score maps: trainer/ti_cross_attn_loss.py:239-268
 * (process_and_stack_attention_scores) + trainer/loss.py:10-80 (compute_token_attention_loss), main.py:342-345.  The loss depends on the
 * stacked maps only through their mean over layers, so the inputs are ONE fp32 sum of raw score maps per resolution (groups sorted by size,
 * g[0] the smallest; S [B*h*w, 128], 77 valid columns) and the outputs one gradient per resolution, d(weight * loss)/dS as bf16 [B*h*w, 128]
 * (dS) and transposed [B*128, h*w] (dSt) - the operands of the score-gradient GEMMs.  Wh [h0, h] / Ww [w0, w]: the 1-D operators of the
 * bicubic resize to the smallest map (F.interpolate, align_corners=False), ch / cw their column sums; NULL for g[0].  mask [B,4,mH,mW] fp32
 * (channel 0, nearest-resized to h0 x w0), tok_w [B,77] / tok_cnt [B] / ti_onehot [B,n_tok,77] / has_ti [B]: the per-caption constants
 * (which columns are caption tokens, their count, where the trained tokens sit, whether all of them are present).  loss[0] = the un-weighted
 * loss.  dheat [B, n_tok, h*w] fp32 and ws (>= sdlt_token_attention_ws_floats floats) are scratch.  n_layers = stacked layers in total. */
typedef struct sdlt_ta_group {
  const float* S; const float* Wh; const float* Ww; const float* ch; const float* cw;
  void* dS; void* dSt; float* dheat;
  int32_t h, w;
} sdlt_ta_group;
typedef struct sdlt_ta_params {
  sdlt_ta_group g[4];
  const float* mask; const float* tok_w; const float* tok_cnt; const float* ti_onehot; const float* has_ti;
  float* ws; int64_t ws_floats;
  float* loss;
  int32_t ngroups, B, n_tok, n_layers, mH, mW;
  float weight;
  int32_t max_px, max_tmp, max_w, pad_;      /* filled in by the entry point */
} sdlt_ta_params;
int64_t sdlt_token_attention_ws_floats(const sdlt_ta_params* p);
int sdlt_token_attention_loss(const sdlt_ta_params* p, void* stream);

/* dX of nearest-2x upsampling: out[b,h,w,:] = sum of the 2x2 block of in [B,2H,2W,C]. */
int sdlt_sum2x2(const void* in, int32_t B, int32_t H, int32_t W, int32_t C, void* out, void* stream);
/* out[b,c] = sum_r x[b*R + r, c] as fp32 [B,C] and / or bf16 [B,C] (either may be NULL) - gradient of the per-batch time-embedding
 * bias of a ResnetBlock2D (h + time_emb_proj(silu(temb))[:, :, None, None]).  Row splits meet in the scratch `ws`
 * (>= min(32, ceil(R/256)) * B * C floats, any contents) and are added in a fixed order: bitwise reproducible. */
int sdlt_colsum(const void* x, int64_t ldx, int32_t B, int32_t R, int32_t C, float* ws, int64_t ws_floats, float* out, void* out_bf16, void* stream);

#ifdef __cplusplus
}
#endif
#endif
