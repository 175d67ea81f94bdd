// Row-strip GEMM for the text encoders (gfx950):  Y[B*Tp, N] = X[B*Tp, K] . W[N, K]^T  with 77 valid rows per batch element.
//
// The CLIP towers run on ONE 77-token sequence per image (trainer/inference.py:131-177 via main.py:306-308): every Linear is an
// 80-row product whose cost is streaming its weights once - 39 MB per OpenCLIP-bigG layer, 1.4 GB per encoder pass and direction.
// The tiled GEMM (gemm.hip) cuts such a product into 64x64 tiles + split-K: 10-40 workgroups, 9-18 us per launch, 30-150 TFLOP/s.
// Here the roles are turned round:
//   * a workgroup owns a STRIP of 16*J output columns over the full K; its 4 waves split K in interleaved 64-column steps
//     (wave w takes steps w, w+4, ...), so the 4 waves read 512 contiguous bytes of every row;
//   * every wave stages its own operands - the 80 activation rows and its 16 J weight rows of a step - through a PRIVATE LDS ring with
//     global_load_lds in full 128-byte rows (source-side XOR swizzle, conflict-free ds_read_b128 fragments): no block barrier in the K
//     loop, counted vmcnt waits, 24 KB of loads in flight per wave.  (A first version loaded both MFMA operands straight from global
//     memory in fragment layout - 16 rows x 64 B per instruction - and ran at 30 GB/s per CU: the texture path serves one row per quad
//     and cycle; full-line staging is 2-3x faster, tools/strip_probe.py.)
//   * the 4 partial 80 x 16J tiles meet in LDS once, at the end (fixed order: bitwise reproducible);
//   * N / (16 J) workgroups (48 ... 320 for the CLIP widths).  Long-K products can be cut into K slices whose fp32 tiles the CONSUMER adds in
//     its prologue (P != NULL -> sdlt_layernorm_bwd_slabs): a reduction at the launch boundary is free, while the in-kernel last-arriver
//     seam (splitk > 1: write-through slab stores, drain, ticket, acquire, slab reads) costs 4-5 us - more than the shorter K walk saves
//     at every CLIP shape, so it stays an option.
// Fused around it:
//   * LayerNorm in front (ln = 1): the product runs on the RAW rows with pre-scaled weights W' = W o gamma; the row statistics
//     come from the same fragments through two more MFMAs per fragment (ones . x -> row sums, x . x^T -> its diagonal = row sums
//     of squares), and the epilogue applies  y = rstd (acc - mean c1[n]) + c2[n],  c1 = rowsum(W'), c2 = W beta + bias.
//     No normalised copy of the activations exists; (mean, rstd) are written for the LayerNorm backward.
//   * bias, residual, the MLP activation as a second output (act = 1 quick_gelu, 2 gelu) or its derivative as a factor (Z given).
// Measured (tools/strip_probe.py, weights rotating through HBM): 1280 x 1280 5.3 us (tiled 9.9), 3840 x 1280 + LayerNorm 8.0 (13.7),
// 5120 x 1280 + LayerNorm + gelu 10.8 (15.7), 1280 x 5120 13.0 (15.1), 768 x 768 3.7 (7.9).
#include <cstdlib>
#include "common.h"
#include "../../include/sdlt_kernels.h"

namespace {

constexpr int NW = 4;            // waves per workgroup = K split (interleaved 64-column steps: 4 waves read 512 contiguous bytes of a row)
constexpr int MB = 5;            // 16-row blocks per batch element: 80 rows hold the 77 tokens
constexpr int XROWS = 16 * MB;
constexpr int ROWB = 128;        // bytes of one staged row (64 bf16)

// 16-byte WRITE-THROUGH store (sc1): the bytes leave this XCD's L2, so publishing them to a workgroup on another XCD needs no release fence
// (buffer_wbl2 walks the whole L2: 1.7 - 6.5 us per workgroup) - only a drained vmcnt before the ticket (MI355X_MICROARCH.md, hand-off rows)
__device__ __forceinline__ void store16_sc1(void* ptr, f32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(ptr), "v"(v) : "memory");
}

template <int N_>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory"); }

__device__ __forceinline__ float act_f(int act, float x) { return act == 1 ? x / (1.f + __expf(-1.702f * x)) : gelu_f(x); }
__device__ __forceinline__ float dact_f(int act, float x) {
  if (act == 1) { const float sg = 1.f / (1.f + __expf(-1.702f * x)); return sg + 1.702f * x * sg * (1.f - sg); }
  return dgelu_f(x);
}

// J: 16-column blocks per workgroup; LN: LayerNorm folded in front; R: ring slots per wave (R - 1 K steps of DMA in flight)
// (bx, by, gx: this workgroup's place in ITS problem's grid - blockIdx / gridDim of a single launch, a share of the grid of a paired one)
template <int J, bool LN, int R>
__device__ __forceinline__ void strip_body(const sdlt_strip_params& p, char* smem, const int bx, const int by, const int gx) {
  constexpr int SROWS = XROWS + 16 * J, SLOT = SROWS * ROWB;          // a ring slot: 80 activation rows + 16 J weight rows of one K step
  constexpr int PIECES = SROWS / 8;                                   // DMA instructions per step (8 rows x 128 B each)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 15, g = lane >> 4;
  // split-K (p.splitk = S > 1): S neighbouring workgroups share a strip, each walks K / S columns; the last one to arrive adds the S
  // partial tiles in split order and runs the epilogue (fixed order: bitwise reproducible)
  const int S = p.splitk > 1 ? p.splitk : 1;
  const int strip = S == 1 ? bx : div_small_u(bx, S), split = S == 1 ? 0 : bx - strip * S;
  const int n0 = strip * (16 * J);
  const int64_t row0 = (int64_t)by * p.Tp;
  // 64-column steps of THIS wave: the K / 256 steps are dealt out to the splits as evenly as they go (the first `rem` splits take one more)
  const int tsteps = p.K >> 8, sbase = tsteps / S, srem = tsteps - sbase * S;
  const int nsteps = sbase + (split < srem ? 1 : 0);
  const int kbase = (split * sbase + (split < srem ? split : srem)) * (NW * 64);      // first column of this workgroup's K range

  // ---- epilogue operands of the (row block, column block) units this wave finishes: requested first (older than every DMA)
  constexpr int UNITS = MB * J, UPW = (UNITS + NW - 1) / NW;
  f32x4 e_bias[UPW], e_c1[UPW];
  uint2 e_res[UPW], e_z[UPW];
#pragma unroll
  for (int q = 0; q < UPW; ++q) {
    const int u = wave + q * NW;
    e_bias[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
    e_c1[q] = e_bias[q];
    e_res[q] = make_uint2(0u, 0u);
    e_z[q] = e_res[q];
    if (u < UNITS) {
      const int j = u / MB, mb = u - j * MB, n = n0 + 16 * j + 4 * g, t = mb * 16 + r;
      if (LN) { e_c1[q] = *(const f32x4*)(p.c1 + n); e_bias[q] = *(const f32x4*)(p.c2 + n); }
      else if (p.bias) e_bias[q] = *(const f32x4*)(p.bias + n);
      if (t < p.T) {
        if (p.R) e_res[q] = *(const uint2*)((const bf16_t*)p.R + (row0 + t) * p.ldr + n);
        if (p.Z) e_z[q] = *(const uint2*)((const bf16_t*)p.Z + (row0 + t) * p.ldz + n);
      }
    }
  }

  // ---- staging geometry: a DMA piece is 8 rows x 128 B, lane -> (row lane/8, 16-byte position lane%8); the LDS image holds source
  // chunk c of row `row` at position c ^ (row & 7) (applied to the SOURCE address: the DMA's LDS side is lane-linear), which makes the
  // fragment reads below (16 rows x one chunk per 16 lanes) conflict-free
  const int srow = lane >> 3, schunk = (lane & 7) ^ srow;
  const bf16_t* xsrc = (const bf16_t*)p.X + (row0 + srow) * p.ldx + schunk * 8;
  const bf16_t* wsrc = (const bf16_t*)p.W + (int64_t)(n0 + srow) * p.ldw + schunk * 8;
  const int64_t x8 = 8 * p.ldx, w8 = 8 * p.ldw;
  char* ring = smem + wave * (R * SLOT);
  // every workgroup reads the same 80 activation rows: each starts its K walk at a different step (a fixed function of its index, so
  // the summation order of a given output is the same in every run), otherwise all of them hit the same L2 channels at the same time
  const int rot = strip - div_small_u(strip, nsteps) * nsteps;
  auto issue = [&](int i, int slot) {
    int ii = i + rot;
    ii = ii >= nsteps ? ii - nsteps : ii;
    const int k0 = kbase + (wave + NW * ii) * 64;
    char* dst = ring + slot * SLOT;
#pragma unroll
    for (int q = 0; q < XROWS / 8; ++q) glds16(xsrc + q * x8 + k0, dst + q * 1024);
#pragma unroll
    for (int q = 0; q < 2 * J; ++q) glds16(wsrc + q * w8 + k0, dst + XROWS * ROWB + q * 1024);
  };
  const int foff0 = r * ROWB + (((0 * 4 + g) ^ (r & 7)) << 4), foff1 = r * ROWB + (((1 * 4 + g) ^ (r & 7)) << 4);

  f32x4 acc[J][MB], gs[MB], sm[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    gs[mb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    sm[mb] = gs[mb];
#pragma unroll
    for (int j = 0; j < J; ++j) acc[j][mb] = gs[mb];
  }
  bf16x8 ones;
#pragma unroll
  for (int i = 0; i < 8; ++i) ones[i] = (__bf16)1.0f;

  // Long K walks (more than two ring turns: K >= 1792 per split): the weight rows of the steps BEHIND the ring's reach are touched now, one lane per 128-byte line.  A strip
  // launch is a handful of workgroups that stream a weight matrix from HBM exactly once; with R - 1 steps in flight every ring turn paid its own first-touch round trip.
  // Round 5, tools/strip_probe.py (weights rotating through 600 MB): 1280 x 5120 18.0 -> 13.8 us, 1280 x 3840 14.7 -> 11.8, 768 x 3072 12.1 -> 9.9; the five-step walks
  // (K = 1280) are neutral to 0.4 us worse and keep the plain prefill; whole step -0.25 ms.  (The same idea LOST on the wave-split-K kernel, whose 256 workgroups are
  // bandwidth-bound - DESIGN 4.14; here 16-320 workgroups are latency-bound.)  Issued before the prefill: oldest loads in flight, covered by every counted wait below; the
  // dummy registers stay reserved until behind the K walk.
  uint32_t tdum[5] = {0u, 0u, 0u, 0u, 0u};
  if (nsteps > 2 * R) {
    const int nline = (nsteps - R) * 16 * J;          // (step, weight row) pairs to touch (at most 320: the tail of a very long walk is left to the ring)
#pragma unroll
    for (int t = 0; t < 5; ++t) {
      const int idx = lane + 64 * t;
      if (idx < nline) {
        const int st = idx / (16 * J), row = idx - st * (16 * J);
        int ii = R + st + rot;
        ii = ii >= nsteps ? ii - nsteps : ii;
        const char* a = (const char*)((const bf16_t*)p.W + (int64_t)(n0 + row) * p.ldw + kbase + (wave + NW * ii) * 64);
        asm volatile("global_load_dword %0, %1, off" : "=&v"(tdum[t]) : "v"(a) : "memory");
      }
    }
  }
#pragma unroll
  for (int s = 0; s < R; ++s)
    if (s < nsteps) issue(s, s);
  int slot = 0;
  for (int i = 0; i < nsteps; ++i) {
    // step i has landed once at most min(R - 1, steps issued after it) steps of DMA are still in flight (in-order completion)
    const int after = nsteps - 1 - i;
    if (after >= R - 1) wait_vmcnt<PIECES * (R - 1)>();
    else if (R > 2 && after == 1) wait_vmcnt<PIECES>();
    else wait_vmcnt<0>();
    const char* base = ring + slot * SLOT;
    bf16x8 xf[2][MB], wf[2][J];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int fo = kk ? foff1 : foff0;
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) xf[kk][mb] = *(const bf16x8*)(base + mb * 16 * ROWB + fo);
#pragma unroll
      for (int j = 0; j < J; ++j) wf[kk][j] = *(const bf16x8*)(base + (XROWS + 16 * j) * ROWB + fo);
    }
    if (i + R < nsteps) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the slot is refilled: its fragments must be in registers first
      issue(i + R, slot);
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
        for (int j = 0; j < J; ++j) acc[j][mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[kk][j], xf[kk][mb], acc[j][mb], 0, 0, 0);
        if constexpr (LN) {
          sm[mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, xf[kk][mb], sm[mb], 0, 0, 0);          // D[.][m] = sum_k x[m,k]
          gs[mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[kk][mb], xf[kk][mb], gs[mb], 0, 0, 0);  // D[m'][m] = <x[m'], x[m]>
        }
      }
    slot = slot + 1 == R ? 0 : slot + 1;
  }

  asm volatile("" ::"v"(tdum[0]), "v"(tdum[1]), "v"(tdum[2]), "v"(tdum[3]), "v"(tdum[4]));
  // ---- the NW partial tiles (and row statistics) meet in LDS (the rings are dead: every DMA has been waited for and read)
  __syncthreads();
  f32x4* red = (f32x4*)smem;                                        // [NW][UNITS][64 lanes]
  float* st = (float*)(smem + (size_t)NW * UNITS * 64 * 16);        // [NW][MB][16 rows][sum, sumsq]   (LN only)
#pragma unroll
  for (int j = 0; j < J; ++j)
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) red[(wave * UNITS + j * MB + mb) * 64 + lane] = acc[j][mb];
  if constexpr (LN) {
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      float* d = st + ((wave * MB + mb) * 16 + r) * 2;
      if (g == 0) d[0] = sm[mb][0];
      if ((r >> 2) == g) {                       // the Gram block's diagonal element of row r lives in lane 16*(r/4) + r, component r%4
        const int c = r & 3;
        d[1] = c == 0 ? gs[mb][0] : (c == 1 ? gs[mb][1] : (c == 2 ? gs[mb][2] : gs[mb][3]));
      }
    }
  }
  __syncthreads();
  // this workgroup's tile (and row statistics), one (row block, column block) unit per wave and round
  f32x4 part[UPW];
  float2 pst[UPW];
#pragma unroll
  for (int q = 0; q < UPW; ++q) {
    const int u = wave + q * NW;
    part[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
    pst[q] = make_float2(0.f, 0.f);
    if (u < UNITS) {
      const int mb = u % MB;
      f32x4 v = red[u * 64 + lane];
#pragma unroll
      for (int w = 1; w < NW; ++w) v += red[(w * UNITS + u) * 64 + lane];
      part[q] = v;
      if constexpr (LN) {
#pragma unroll
        for (int w = 0; w < NW; ++w) {
          const float2 pr = *(const float2*)(st + ((w * MB + mb) * 16 + r) * 2);
          pst[q].x += pr.x;
          pst[q].y += pr.y;
        }
      }
    }
  }
  if (p.P) {
    // partial output (S >= 1): this workgroup's fp32 tile goes to slab `split` of P ([S][B*Tp][N] fp32) as it is - no bias, residual or
    // activation - and whoever consumes P adds the slabs in its prologue (sdlt_layernorm_bwd_slabs): a reduction at the launch boundary
    // costs nothing, the in-kernel seam below 4 - 5 us
#pragma unroll
    for (int q = 0; q < UPW; ++q) {
      const int u = wave + q * NW;
      if (u >= UNITS) break;
      const int j = u / MB, mb = u - j * MB, n = n0 + 16 * j + 4 * g, t = mb * 16 + r;
      if (t < p.T) *(f32x4*)(p.P + ((size_t)split * p.B * p.Tp + row0 + t) * p.ldp + n) = part[q];
    }
    return;
  }
  if (S > 1) {
    // slabs: [batch][strip][split][unit][lane] f32x4 (+ float2 statistics behind them); hand-off as MI355X_MICROARCH.md prescribes:
    // write-through (sc1) stores -> every wave drains them -> barrier -> ticket; the last arriver: ONE agent-scope acquire, plain loads
    constexpr int UB = UNITS * 64 * (LN ? 24 : 16);
    const int nstrips = gx / S;
    char* slab0 = (char*)p.ws + ((size_t)(by * nstrips + strip) * S) * UB;
    char* mine = slab0 + (size_t)split * UB;
#pragma unroll
    for (int q = 0; q < UPW; ++q) {
      const int u = wave + q * NW;
      if (u < UNITS) {
        store16_sc1(mine + (u * 64 + lane) * 16, part[q]);
        if constexpr (LN)
          __hip_atomic_store((unsigned long long*)(mine + UNITS * 64 * 16 + (u * 64 + lane) * 8), __builtin_bit_cast(unsigned long long, pst[q]),
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int* flag = (int*)smem;                 // (the reduction scratch is dead: `part` holds what this workgroup needs of it)
    if (threadIdx.x == 0) {
      int* cnt = p.cnt + by * nstrips + strip;
      const int ticket = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (ticket == S - 1) {
        __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // re-armed for the next launch
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
      *flag = ticket;
    }
    __syncthreads();
    if (*flag != S - 1) return;
#pragma unroll
    for (int q = 0; q < UPW; ++q) {
      const int u = wave + q * NW;
      if (u < UNITS) {
        f32x4 v = *(const f32x4*)(slab0 + (u * 64 + lane) * 16);
        float2 sv = make_float2(0.f, 0.f);
        if constexpr (LN) sv = *(const float2*)(slab0 + UNITS * 64 * 16 + (u * 64 + lane) * 8);
        for (int sp = 1; sp < S; ++sp) {
          const char* o = slab0 + (size_t)sp * UB;
          v += *(const f32x4*)(o + (u * 64 + lane) * 16);
          if constexpr (LN) {
            const float2 t2 = *(const float2*)(o + UNITS * 64 * 16 + (u * 64 + lane) * 8);
            sv.x += t2.x;
            sv.y += t2.y;
          }
        }
        part[q] = v;
        pst[q] = sv;
      }
    }
  }
#pragma unroll
  for (int q = 0; q < UPW; ++q) {
    const int u = wave + q * NW;
    if (u >= UNITS) break;
    const int j = u / MB, mb = u - j * MB, n = n0 + 16 * j + 4 * g, t = mb * 16 + r;
    f32x4 v = part[q];
    if constexpr (LN) {
      const float s1 = pst[q].x, s2 = pst[q].y;
      const float inv = 1.f / (float)p.K, mean = s1 * inv;
      const float var = fmaxf(s2 * inv - mean * mean, 0.f), rstd = rsqrtf(var + p.eps);
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = rstd * (v[i] - mean * e_c1[q][i]) + e_bias[q][i];
      if (p.stats && strip == 0 && j == 0 && g == 0 && t < p.T) *(float2*)(p.stats + (row0 + t) * 2) = make_float2(mean, rstd);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] += e_bias[q][i];
    }
    if (t >= p.T) continue;
    if (p.R) {
      v[0] += bf2f(e_res[q].x & 0xffff); v[1] += bf2f(e_res[q].x >> 16); v[2] += bf2f(e_res[q].y & 0xffff); v[3] += bf2f(e_res[q].y >> 16);
    }
    if (p.Z) {
      v[0] *= dact_f(p.act, bf2f(e_z[q].x & 0xffff)); v[1] *= dact_f(p.act, bf2f(e_z[q].x >> 16));
      v[2] *= dact_f(p.act, bf2f(e_z[q].y & 0xffff)); v[3] *= dact_f(p.act, bf2f(e_z[q].y >> 16));
    }
    *(uint2*)((bf16_t*)p.Y + (row0 + t) * p.ldy + n) = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
    if (p.Y2) {       // activation side output, from the fp32 values (as the tiled GEMM's epilogue 3 does)
      *(uint2*)((bf16_t*)p.Y2 + (row0 + t) * p.ldy2 + n) =
          make_uint2(pack2bf(act_f(p.act, v[0]), act_f(p.act, v[1])), pack2bf(act_f(p.act, v[2]), act_f(p.act, v[3])));
    }
  }
}

template <int J, bool LN, int R>
__global__ __launch_bounds__(64 * NW) void strip_kernel(const sdlt_strip_params p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  strip_body<J, LN, R>(p, smem, blockIdx.x, blockIdx.y, gridDim.x);
}
// Two independent problems of the same kernel variant in ONE launch (blockIdx.z picks the problem): layer i of CLIP-L rides along layer i of
// OpenCLIP-bigG - each is a chain of launch-bound 77-row products, neither fills the chip, and a launch costs more than CLIP-L's share of it.
template <int J, bool LN, int R>
__global__ __launch_bounds__(64 * NW) void strip_pair_kernel(const sdlt_strip_params p0, const sdlt_strip_params p1) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if (blockIdx.z == 0) {
    const int gx = (p0.N / (16 * J)) * (p0.splitk > 1 ? p0.splitk : 1);
    if ((int)blockIdx.x < gx && (int)blockIdx.y < p0.B) strip_body<J, LN, R>(p0, smem, blockIdx.x, blockIdx.y, gx);
  } else {
    const int gx = (p1.N / (16 * J)) * (p1.splitk > 1 ? p1.splitk : 1);
    if ((int)blockIdx.x < gx && (int)blockIdx.y < p1.B) strip_body<J, LN, R>(p1, smem, blockIdx.x, blockIdx.y, gx);
  }
}

template <int J, bool LN, int R>
int launch_strip(const sdlt_strip_params& p, hipStream_t s) {
  constexpr int SLOT = (XROWS + 16 * J) * ROWB;
  const int smem = NW * R * SLOT;          // >= the reduction scratch NW * MB * J * 64 * 16 + statistics
  static_assert(NW * R * SLOT >= NW * MB * J * 64 * 16 + NW * MB * 16 * 2 * 4 && NW * R * SLOT <= 160 * 1024, "LDS budget");
  if (sdlt_raise_smem((const void*)strip_kernel<J, LN, R>, smem)) SDLT_FAIL(SDLT_ERR_LAUNCH, "sdlt_strip_gemm: cannot raise the dynamic LDS limit to %d bytes", smem);
  const int S = p.splitk > 1 ? p.splitk : 1;
  if (S > 1 && !p.P) {
    const size_t need = (size_t)p.B * (p.N / (16 * J)) * S * (MB * J * 64 * (LN ? 24 : 16));
    if (!p.ws || !p.cnt || need > (size_t)p.ws_bytes || p.B * (p.N / (16 * J)) > p.cnt_len)
      SDLT_FAIL(SDLT_ERR_SHAPE, "sdlt_strip_gemm: split-K %d needs %zu workspace bytes and %d counters (have %lld, %d)", S, need, p.B * (p.N / (16 * J)),
                (long long)p.ws_bytes, p.cnt_len);
  }
  hipLaunchKernelGGL((strip_kernel<J, LN, R>), dim3((p.N / (16 * J)) * S, p.B), dim3(64 * NW), smem, s, p);
  SDLT_CHECK_LAUNCH();
  return SDLT_OK;
}

template <int J, bool LN, int R>
int launch_strip_pair(const sdlt_strip_params& a, const sdlt_strip_params& b, hipStream_t s) {
  constexpr int SLOT = (XROWS + 16 * J) * ROWB;
  const int smem = NW * R * SLOT;
  if (sdlt_raise_smem((const void*)strip_pair_kernel<J, LN, R>, smem)) SDLT_FAIL(SDLT_ERR_LAUNCH, "sdlt_strip_gemm_pair: cannot raise the dynamic LDS limit to %d bytes", smem);
  const int ga = (a.N / (16 * J)) * (a.splitk > 1 ? a.splitk : 1), gb = (b.N / (16 * J)) * (b.splitk > 1 ? b.splitk : 1);
  hipLaunchKernelGGL((strip_pair_kernel<J, LN, R>), dim3(ga > gb ? ga : gb, a.B > b.B ? a.B : b.B, 2), dim3(64 * NW), smem, s, a, b);
  SDLT_CHECK_LAUNCH();
  return SDLT_OK;
}

// 32-column strips for the q|k|v- and mlp-wide products (N >= 2304; SDLT_STRIP_WIDE_MIN for A/B): 16-column ones would exceed one wave of
// workgroups at 5120, and the two text encoders' launches pair only when both take the same width - q|k|v 2304 / 32 + 3840 / 32 = 192
// workgroups, fc1 3072 / 32 + 5120 / 32 = 256.  Whole step: 2304 -> 44.9 ms, 3072 -> 45.0, 4096 (the rule before pairing) -> 45.1; SD1.5
// 24.83 / 24.85 / 24.99.  The strip width fixes the order of the fp32 K summation, so single and paired launches agree bit for bit only
// because both use this one rule.
bool strip_wide(int N) {
  static const int wmin = getenv("SDLT_STRIP_WIDE_MIN") ? atoi(getenv("SDLT_STRIP_WIDE_MIN")) : 2304;
  return N >= wmin && (N % 32) == 0;
}

int strip_check(const sdlt_strip_params& p) {
  const int S_ = p.splitk > 1 ? p.splitk : 1;
  if (p.B <= 0 || p.T <= 0 || p.T > 16 * MB || p.Tp < 16 * MB || p.N <= 0 || (p.N % 16) || p.K <= 0 || (p.K % 256) || S_ > 16 || S_ > (p.K >> 8))
    SDLT_FAIL(SDLT_ERR_SHAPE, "sdlt_strip_gemm: B=%d T=%d Tp=%d N=%d K=%d (T <= 80 <= Tp, N %% 16 == 0, K %% 256 == 0)", p.B, p.T, p.Tp, p.N, p.K);
  if (p.P && (p.ln || p.bias || p.R || p.Z || p.Y2 || (p.ldp % 4) || ((uintptr_t)p.P & 15)))
    SDLT_FAIL(SDLT_ERR_UNSUPPORTED, "sdlt_strip_gemm: a partial output (P) takes no LayerNorm / bias / residual / activation and needs 16-byte rows");
  if (!p.X || !p.W || (!p.Y && !p.P) || (p.ldx % 8) || (p.ldw % 8) || (p.ldy % 4) || ((uintptr_t)p.X & 15) || ((uintptr_t)p.W & 15) || ((uintptr_t)p.Y & 7) ||
      (p.R && ((p.ldr % 4) || ((uintptr_t)p.R & 7))) || (p.Z && ((p.ldz % 4) || ((uintptr_t)p.Z & 7))) || (p.Y2 && ((p.ldy2 % 4) || ((uintptr_t)p.Y2 & 7))))
    SDLT_FAIL(SDLT_ERR_ALIGN, "sdlt_strip_gemm: operand alignment (X / W rows 16 B, outputs 8 B)");
  if ((p.Y2 || p.Z) && p.act != 1 && p.act != 2) SDLT_FAIL(SDLT_ERR_UNSUPPORTED, "sdlt_strip_gemm: act %d", p.act);
  if (p.ln && (!p.c1 || !p.c2 || ((uintptr_t)p.c1 & 15) || ((uintptr_t)p.c2 & 15))) SDLT_FAIL(SDLT_ERR_UNSUPPORTED, "sdlt_strip_gemm: ln needs c1 / c2");
  if (!p.ln && p.bias && ((uintptr_t)p.bias & 15)) SDLT_FAIL(SDLT_ERR_ALIGN, "sdlt_strip_gemm: bias alignment");
  return SDLT_OK;
}

}  // namespace

extern "C" int sdlt_strip_gemm(const sdlt_strip_params* pp, void* stream) {
  const sdlt_strip_params& p = *pp;
  const int rc = strip_check(p);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  const bool wide = strip_wide(p.N);
  if (p.ln) return wide ? launch_strip<2, true, 2>(p, s) : launch_strip<1, true, 3>(p, s);
  return wide ? launch_strip<2, false, 2>(p, s) : launch_strip<1, false, 3>(p, s);
}

// reference: none (launch structure only) - see strip_pair_kernel.  Both problems must ask for the same kernel variant (LayerNorm fold or
// not, strip width) and neither may use the in-kernel K split
// (its workspace is one per stream).
extern "C" int sdlt_strip_gemm_pair(const sdlt_strip_params* pa, const sdlt_strip_params* pb, void* stream) {
  const sdlt_strip_params &a = *pa, &b = *pb;
  int rc = strip_check(a);
  if (rc) return rc;
  rc = strip_check(b);
  if (rc) return rc;
  if ((a.ln != 0) != (b.ln != 0)) SDLT_FAIL(SDLT_ERR_UNSUPPORTED, "sdlt_strip_gemm_pair: one problem folds a LayerNorm, the other does not");
  if ((a.splitk > 1 && !a.P) || (b.splitk > 1 && !b.P)) SDLT_FAIL(SDLT_ERR_UNSUPPORTED, "sdlt_strip_gemm_pair: in-kernel K split");
  hipStream_t s = (hipStream_t)stream;
  if (strip_wide(a.N) != strip_wide(b.N)) SDLT_FAIL(SDLT_ERR_UNSUPPORTED, "sdlt_strip_gemm_pair: N = %d and N = %d want different strip widths", a.N, b.N);
  const bool wide = strip_wide(a.N);
  if (a.ln) return wide ? launch_strip_pair<2, true, 2>(a, b, s) : launch_strip_pair<1, true, 3>(a, b, s);
  return wide ? launch_strip_pair<2, false, 2>(a, b, s) : launch_strip_pair<1, false, 3>(a, b, s);
}
