// Local Spatial Encoding + attentive pooling (RandLA-Net "LocalFeatureAggregation") for gfx950.
//
// Replaces LocalFeatureAggregation.propagate/message
// (/root/reference/myria3d/models/modules/pyg_randla_net.py:112-152): PyG gathers x_j / pos_i / pos_j into
// [E,.] tensors, runs mlp_encoder (Linear 10->d + BatchNorm + LeakyReLU) and mlp_attention (Linear ch->ch,
// no bias) over every edge, a scatter-softmax over each centre's K neighbours, and a scatter-add.
//
// Here (forward, m3d_lfa_fwd): ONE kernel, nothing of size [E,.] ever reaches HBM.  A 256-thread workgroup
// owns TC centres = TC*KP edge rows:
//   phase 1  neighbour ids -> LDS; x_j rows gathered with 16-byte loads into an LDS tile F[rows][ch];
//            relative position encoding r = [p_i, p_j, p_j-p_i, |p_j-p_i|] and the (BatchNorm-folded) encoder
//            evaluated per edge on the VALU with wave-uniform weights -> F[:, d:ch]
//   phase 2  attention logits A = F * W_att^T on v_mfma_f32_16x16x4_f32 (exact fp32): each 16-row MFMA tile is
//            one centre's 16 neighbours; W_att comes pre-packed in fragment order (one coalesced 16-byte load
//            per lane per 4 k-steps) and every B fragment is reused for the 4 M-tiles a wave owns
//   phase 3  softmax over the neighbours directly in the MFMA C layout (4 registers + 2 cross-lane steps per
//            channel) and the weighted sum with F re-read from LDS; 64-byte row segments written to HBM.
// Train-mode BatchNorm of the encoder is folded as well: the encoder is affine in r, so the batch mean and
// variance of W r + b over all edges follow from the first and second moments of r (m3d_lfa_moments, 65
// numbers per level, fp64), see m3d_lfa_enc_finalize.
//
// Backward: the fused kernel of lfa_bwd.hip (nothing of size [E,.] in HBM either).  The per-edge entry points further down
// (m3d_lfa_edge_features / _edge_softmax_* / _edge_features_bwd: round 1's unfused backward, which materialises F and the
// logits) serve neighbour counts above 32, which the fused kernels are not instantiated for, and are the step-by-step
// cross-check the parity tests run against the fused kernels.  The encoder's parameter gradients (through its train-mode BatchNorm) are reconstructed analytically from
// 11*d accumulated numbers in m3d_lfa_enc_bwd_finalize.
#include "m3d_common.h"
#include "lfa_common.h"
#include "../../include/m3d_hip.h"

// 1: double-buffer the attention-weight fragments of the MFMA loop (untuned knob for a same-box A/B; default off)
#ifndef LFA_B_PREFETCH
#define LFA_B_PREFETCH 0
#endif

// CH is a template parameter: with a runtime channel count the index arithmetic of the gather / encoder loops
// (f / D4, f % D4, ...) compiled to integer divisions and made the ch <= 32 kernels VALU-issue-bound
// (rocprofv3 SQ_INSTS_VALU: 650 VALU instructions per wave of 64 edges at ch = 16).
// BF: the attention GEMM runs on bf16 matrix cores (operands rounded to bf16 when the fragments are built from the fp32
// LDS tile / pre-packed as bf16, fp32 accumulate): BASELINE config 2's "bf16" arithmetic (torch autocast semantics);
// everything else — gathers, encoder, softmax, the pooled sum, all storage — stays fp32.  CH >= 32 only.
// IOH: x and out hold bf16 (M3D_IO_BF16 in the entry point's flags); the tile, the products and the softmax stay fp32
template <int CH, int KP, bool BF = false, bool IOH = false>
__global__ __launch_bounds__(256) void lfa_fwd_kernel(LfaArgs a) {
  constexpr int CHP = CH < 16 ? 16 : CH;
  constexpr int D = CH / 2;
  constexpr int ROWS = LfaCfg<CHP>::ROWS;
  constexpr int TC = ROWS / KP;          // centres per workgroup
  constexpr int KT = KP / 16;            // MFMA M-tiles per centre
  constexpr int STR = CHP + 2;           // LDS row stride (floats): bank = 2*row + k for fragment reads
  constexpr int MT = ROWS / 16, NT = CHP / 16;
  constexpr int WN = NT < 4 ? NT : 4, WM = 4 / WN;
  constexpr int NTW = NT / WN, MTW = MT / WM;
  constexpr int S4 = CHP / 16;           // groups of 4 k-steps
  static_assert(MTW % KT == 0, "centre tiles must stay inside one wave");
  __shared__ float F[ROWS * STR];
  __shared__ int nbr[ROWS];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  const int K = a.K;
  const int64_t c0 = xcd_major(blockIdx.x, gridDim.x) * TC;  // (XCD-aware: m3d_common.h)

  // ---- phase 1a: neighbour ids
  for (int e = tid; e < ROWS; e += 256) {
    int ci = e / KP, k = e % KP;
    int64_t i = c0 + ci;
    int j = -1;
    if (i < a.n && k < K) j = a.idx[i * K + k];
    nbr[e] = j;
  }
  __syncthreads();
  // ---- phase 1b: gather x_j into F[:, 0:D]; zero the padding columns (only when CH < CHP)
  {
    constexpr int D4 = D >> 2;
#pragma unroll
    for (int f = tid; f < ROWS * D4; f += 256) {
      const int e = f / D4, c4 = f % D4;
      const int j = nbr[e];
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (j >= 0) v = io_load4<IOH>(a.x, (size_t)((int64_t)j * D + c4 * 4));
      float* d = &F[e * STR + c4 * 4];
      *(float2*)d = make_float2(v.x, v.y);
      *(float2*)(d + 2) = make_float2(v.z, v.w);
    }
    if (CH < CHP) {
      constexpr int P = CHP - CH;
      for (int f = tid; f < ROWS * P; f += 256) F[(f / P) * STR + CH + (f % P)] = 0.f;
    }
  }
  // ---- phase 1c: relative position encoding + folded encoder -> F[:, D:2D]
  {
    constexpr int NG = 256 / ROWS;
    const int e = tid % ROWS;
    const int grp = __builtin_amdgcn_readfirstlane(tid / ROWS);
    constexpr int DG = D / NG;
    const int j = nbr[e];
    const int64_t i = c0 + e / KP;
    float r[10];
#pragma unroll
    for (int q = 0; q < 10; ++q) r[q] = 0.f;
    if (j >= 0) rel_pos(a.pos4[i], a.pos4[j], r);
    for (int c = grp * DG; c < (grp + 1) * DG; ++c) {
      const float* w = a.wf + c * 10;
      float v = a.bf[c];
#pragma unroll
      for (int q = 0; q < 10; ++q) v += w[q] * r[q];
      F[e * STR + D + c] = j >= 0 ? lrelu(v, a.slope) : 0.f;
    }
  }
  __syncthreads();

  // ---- phase 2: A = F * W_att^T on MFMA
  const int wn = wid % WN, wm = wid / WN;
  f32x4 acc[MTW][NTW];
#pragma unroll
  for (int m = 0; m < MTW; ++m)
#pragma unroll
    for (int t = 0; t < NTW; ++t) acc[m][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if constexpr (BF) {
    static_assert(CHP % 32 == 0, "bf16 tiles are 32 deep");
    constexpr int KS = CHP / 32;
    const uint4* wpb = (const uint4*)a.wp;  // [NT][KS][64 lanes] x 8 bf16: W[16 nt + (l & 15)][32 ks + 8 (l >> 4) + i]
    const float* fa = &F[((wm * MTW) * 16 + lr) * STR + lg * 8];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      Bf16Frag b[NTW];
#pragma unroll
      for (int t = 0; t < NTW; ++t) b[t].q = wpb[((size_t)(wn * NTW + t) * KS + ks) * 64 + lane];
#pragma unroll
      for (int m = 0; m < MTW; ++m) {
        const bf16x8 av = lds_row_to_bf16(fa + m * 16 * STR + ks * 32);
#pragma unroll
        for (int t = 0; t < NTW; ++t) acc[m][t] = mfma_bf16(av, b[t].v, acc[m][t]);
      }
    }
  } else {
    const float* fa = &F[((wm * MTW) * 16 + lr) * STR + lg];
  #if LFA_B_PREFETCH
    // B fragments double-buffered: the loads of k-step group s4+1 are in flight during the MFMAs of group s4
    float4 bn[NTW];
  #pragma unroll
    for (int t = 0; t < NTW; ++t) bn[t] = a.wp[((size_t)(wn * NTW + t) * S4) * 64 + lane];
  #endif
  #pragma unroll 1
    for (int s4 = 0; s4 < S4; ++s4) {
      float4 b[NTW];
  #if LFA_B_PREFETCH
  #pragma unroll
      for (int t = 0; t < NTW; ++t) b[t] = bn[t];
      {
        const int sn = s4 + 1 < S4 ? s4 + 1 : s4;  // last trip: a harmless re-load
  #pragma unroll
        for (int t = 0; t < NTW; ++t) bn[t] = a.wp[((size_t)(wn * NTW + t) * S4 + sn) * 64 + lane];
      }
  #else
  #pragma unroll
      for (int t = 0; t < NTW; ++t) b[t] = a.wp[((size_t)(wn * NTW + t) * S4 + s4) * 64 + lane];
  #endif
  #pragma unroll
      for (int i = 0; i < 4; ++i) {
        float av[MTW];
  #pragma unroll
        for (int m = 0; m < MTW; ++m) av[m] = fa[m * 16 * STR + (s4 * 4 + i) * 4];
  #pragma unroll
        for (int t = 0; t < NTW; ++t) {
          const float bv = i == 0 ? b[t].x : (i == 1 ? b[t].y : (i == 2 ? b[t].z : b[t].w));
  #pragma unroll
          for (int m = 0; m < MTW; ++m) acc[m][t] = mfma16(av[m], bv, acc[m][t]);
        }
      }
    }
  }

  // ---- phase 3: softmax over each centre's neighbours + weighted sum, in the MFMA C layout
  const float pinf = opaque_pinf();
#pragma unroll
  for (int cc = 0; cc < MTW / KT; ++cc) {
    const int mt0 = wm * MTW + cc * KT;
    const int64_t i = c0 + mt0 / KT;
    bool vr[KT][4];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) vr[kt][r] = nbr[(mt0 + kt) * 16 + lg * 4 + r] >= 0;
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
      const int col = (wn * NTW + t) * 16 + lr;
      float mx = -__builtin_inff();
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (vr[kt][r]) mx = max_f(mx, acc[cc * KT + kt][t][r], pinf);
      mx = xgroup_max(mx, pinf);
      float num = 0.f, den = 0.f;
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (vr[kt][r]) {
            float p = __expf(acc[cc * KT + kt][t][r] - mx);
            float f = F[((mt0 + kt) * 16 + lg * 4 + r) * STR + col];
            num += p * f;
            den += p;
          }
      num = xgroup_sum(num);
      den = xgroup_sum(den);
      // (v_rcp_f32: 1 ulp, against ~10 instructions of IEEE division per output; the parity bar is 2e-5)
      if (lg == 0 && i < a.n && col < CH) io_store1<IOH>(a.out, (size_t)(i * CH + col), num * __builtin_amdgcn_rcpf(den + 1e-16f));
    }
  }
}

// ---- the fast path (round 5): every neighbourhood is complete (K == KP and every id >= 0: the caller's promise,
// flags bit 0) and every byte offset fits 32 bits.  Same arithmetic order per output as lfa_fwd_kernel except where noted;
// built from the ISA of the general kernel (tools/isa_count.py: 484 / 523 VALU instructions per wavefront of 64 edges at
// ch = 8 / 16, a kernel that is VALU-issue bound):
//   * no validity masks (the general kernel pays four s_and_saveexec branches, a v_cndmask chain and zero fills per
//     output register for the -1 padding of clouds with fewer than K points);
//   * one dependent load chain per thread — id -> (x_j chunk, p_j) with p_i beside it — instead of ids through LDS, a
//     barrier, then the gathers: the thread that owns edge row e loads ITS row's chunk of x_j, so the id stays in a
//     register and the nbr[] table and its barrier are gone;
//   * 32-bit byte offsets on scalar bases (global_load ... v_off, s[base]) instead of 64-bit pointer arithmetic;
//   * softmax: exp2(fma(a, log2 e, -max * log2 e)) — one fma + v_exp_f32 per edge instead of sub, mul, exp —, one-instruction
//     maxima (v_med3_f32 with +inf: max_f), raw v_sqrt_f32 for the edge length, LeakyReLU as max(v, slope * v)
//     (0 <= slope <= 1: checked by the host);
//   * ch = 8 (PACK2): TWO centres per 16-row MFMA tile.  The general kernel pads ch = 8 to a 16 x 16 x 16 product, i.e.
//     3/4 of the tile's flops and half of the softmax lanes are padding.  Here packed row p * KP + k holds neighbour k of
//     centre 2p in columns 0-7 and of centre 2p + 1 in columns 8-15, the B operand is diag(W^T, W^T) (assembled from the
//     standard fragments: lane (n, g) of the lower block reads what lane (n - 8, g) holds), and output column c of a tile
//     is channel c % 8 of centre 2p + c / 8: half the MFMAs, half the softmax passes, every lane busy, no zero fill.
//   * IN-LANE NEIGHBOURHOODS (second half of round 5, from the SQ counters: the kernels issue VALU instructions for 100 % of
//     their SIMDs' issue slots, a third of them the softmax's cross-lane maxima / sums).  The MFMA C layout gives lane
//     (lr, lg) rows 4 lg + r of a 16-row tile.  A wave owns 4 row tiles; edge (unit u of the wave, neighbour k) is stored in
//     LDS row 64 w + 16 (k / 4) + 4 u + (k % 4) instead of 64 w + 16 u + k — i.e. tile k / 4 holds neighbours 4 (k / 4) .. + 3
//     of all four units of the wave.  Lane (lr, lg) then holds ALL 16 neighbours of unit lg for column lr in its 4 tiles x 4
//     registers: maximum, exponentials and both sums are plain register arithmetic, no permlane swaps, and all four lane
//     groups store an output row (before: only group 0).  K = 32: a unit is half a neighbourhood, ONE row swap joins the
//     halves.  Same MFMA instructions and weight fragments as before — only the row permutation of the LDS tile changed.
// rows of the LDS tile per workgroup in this kernel: 64 per wave-row (WM) — ch = 8 takes 512 edges (256 packed rows)
template <int CHP> struct LfaFullCfg { static constexpr int ROWS = LfaCfg<CHP>::ROWS; };
template <> struct LfaFullCfg<16> { static constexpr int ROWS = 256; };
template <int CH> struct LfaFullRows { static constexpr int ROWS = CH == 8 ? 512 : LfaFullCfg<(CH < 16 ? 16 : CH)>::ROWS; };

// BF: 0 = f32-input MFMA, 1 = bf16 operands (one product), 2 = split-bf16 (hi + lo operands, three products: m3d_common.h)
template <int CH, int KP, int BF, bool IOH = false>
__global__ __launch_bounds__(256) void lfa_fwd_full_kernel(LfaArgs a) {
  constexpr bool PACK2 = CH == 8;
  constexpr int CHP = CH < 16 ? 16 : CH;
  constexpr int D = CH / 2;
  constexpr int ROWS = LfaFullRows<CH>::ROWS;  // edge rows per workgroup
  constexpr int TC = ROWS / KP;                // centres per workgroup
  constexpr int STR = CHP + 2;
  constexpr int PROWS = PACK2 ? ROWS / 2 : ROWS;  // rows of the LDS tile
  constexpr int MT = PROWS / 16, NT = CHP / 16;
  constexpr int WN = NT < 4 ? NT : 4, WM = 4 / WN;
  constexpr int NTW = NT / WN, MTW = MT / WM;
  constexpr int S4 = CHP / 16;
  constexpr int EPT = ROWS >= 256 ? ROWS / 256 : 1;  // edges per thread
  constexpr int NG = ROWS >= 256 ? 1 : 256 / ROWS;   // threads per edge row
  constexpr int DG = D / NG;                         // x_j floats and encoder channels per thread
  constexpr int G4 = DG / 4;
  static_assert(MTW == 4, "a wave owns four row tiles: the 16 neighbours of a unit sit in one lane");
  static_assert(DG % 4 == 0 && G4 >= 1, "a thread gathers whole float4s");
  static_assert(KP == 16 || KP == 32, "whole MFMA tiles per neighbourhood");
  __shared__ float F[PROWS * STR];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  const unsigned n = (unsigned)a.n;
  const unsigned c0 = (unsigned)xcd_major(blockIdx.x, gridDim.x) * TC;
  const int g = __builtin_amdgcn_readfirstlane(NG > 1 ? tid / ROWS : 0);
  const int wn = wid % WN, wm = wid / WN;
  const unsigned elast = n * KP - 1;

  // ---- phase 1: one load chain per thread and edge: id -> (x_j chunk, p_j), p_i beside it
  unsigned jj[EPT];
  float4 pi[EPT], pj[EPT], xg[EPT][G4];
#pragma unroll
  for (int q = 0; q < EPT; ++q) {
    const int e = (NG > 1 ? tid % ROWS : tid) + q * 256;
    unsigned eo = c0 * KP + e;
    eo = eo < elast ? eo : elast;  // rows past the last centre repeat its last edge (computed, never stored)
    jj[q] = (unsigned)a.idx[eo];
    unsigned ci = c0 + e / KP;
    ci = ci < n ? ci : n - 1;
    pi[q] = *(const float4*)((const char*)a.pos4 + ci * 16u);
  }
#pragma unroll
  for (int q = 0; q < EPT; ++q) {
    pj[q] = *(const float4*)((const char*)a.pos4 + jj[q] * 16u);
#pragma unroll
    for (int u = 0; u < G4; ++u)
      xg[q][u] = io_load4_b<IOH>(a.x, jj[q] * (unsigned)(D * 4) + (unsigned)((g * DG + u * 4) * 4));
  }
  // B fragments of the first k-step group: independent of everything above, in flight during phase 1
  float4 b0[NTW];
  if constexpr (BF == 0) {
    if constexpr (PACK2) {
      const float4 t = a.wp[lr < 8 ? lane : lane - 8];
      b0[0] = lr < 8 ? make_float4(t.x, t.y, 0.f, 0.f) : make_float4(0.f, 0.f, t.x, t.y);
    } else {
#pragma unroll
      for (int t = 0; t < NTW; ++t) b0[t] = a.wp[((size_t)(wn * NTW + t) * S4) * 64 + lane];
    }
  }
#pragma unroll
  for (int q = 0; q < EPT; ++q) {
    const int e = (NG > 1 ? tid % ROWS : tid) + q * 256;
    // LDS row of edge (local centre cl, neighbour k): unit = centre (pair of centres: PACK2; half a neighbourhood: KP = 32)
    const int cl = e / KP, k = e % KP;
    const int unit = ((PACK2 ? cl >> 1 : cl) * (KP / 16)) + (k >> 4), kk = k & 15;
    const int prow = (unit >> 2) * 64 + (kk >> 2) * 16 + (unit & 3) * 4 + (kk & 3);
    float* frow = &F[prow * STR + (PACK2 ? (cl & 1) * 8 : 0)];
#pragma unroll
    for (int u = 0; u < G4; ++u) {
      float* d = frow + g * DG + u * 4;
      *(float2*)d = make_float2(xg[q][u].x, xg[q][u].y);
      *(float2*)(d + 2) = make_float2(xg[q][u].z, xg[q][u].w);
    }
    float r[10];
    rel_pos_fast(pi[q], pj[q], r);
#pragma unroll
    for (int cc = 0; cc < DG; ++cc) {
      const int c = g * DG + cc;
      const float* w = a.wf + c * 10;
      float v = a.bf[c];
#pragma unroll
      for (int qq = 0; qq < 10; ++qq) v += w[qq] * r[qq];
      frow[D + c] = fmaxf(v, v * a.slope);
    }
  }
  __syncthreads();

  // ---- phase 2: A = F * W_att^T on MFMA
  f32x4 acc[MTW][NTW];
#pragma unroll
  for (int m = 0; m < MTW; ++m)
#pragma unroll
    for (int t = 0; t < NTW; ++t) acc[m][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if constexpr (BF != 0) {
    static_assert(BF == 0 || CHP % 32 == 0, "bf16 tiles are 32 deep");
    constexpr int KS = CHP / 32;
    const uint4* wpb = (const uint4*)a.wp;
    const uint4* wpl = wpb + (size_t)CH * CH / 8;  // (BF == 2) the lo fragments behind the hi ones
    const float* fa = &F[((wm * MTW) * 16 + lr) * STR + lg * 8];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      Bf16Frag b[NTW], bl[NTW];
#pragma unroll
      for (int t = 0; t < NTW; ++t) {
        b[t].q = wpb[((size_t)(wn * NTW + t) * KS + ks) * 64 + lane];
        if constexpr (BF == 2) bl[t].q = wpl[((size_t)(wn * NTW + t) * KS + ks) * 64 + lane];
      }
#pragma unroll
      for (int m = 0; m < MTW; ++m) {
        if constexpr (BF == 2) {
          const Bf16Split av = lds_row_to_bf16_split(fa + m * 16 * STR + ks * 32);
#pragma unroll
          for (int t = 0; t < NTW; ++t) {
            acc[m][t] = mfma_bf16(av.lo, b[t].v, acc[m][t]);
            acc[m][t] = mfma_bf16(av.hi, bl[t].v, acc[m][t]);
            acc[m][t] = mfma_bf16(av.hi, b[t].v, acc[m][t]);
          }
        } else {
          const bf16x8 av = lds_row_to_bf16(fa + m * 16 * STR + ks * 32);
#pragma unroll
          for (int t = 0; t < NTW; ++t) acc[m][t] = mfma_bf16(av, b[t].v, acc[m][t]);
        }
      }
    }
  } else {
    const float* fa = &F[((wm * MTW) * 16 + lr) * STR + lg];
#pragma unroll 1
    for (int s4 = 0; s4 < S4; ++s4) {
      float4 b[NTW];
#pragma unroll
      for (int t = 0; t < NTW; ++t) b[t] = b0[t];
      if (s4 + 1 < S4) {
#pragma unroll
        for (int t = 0; t < NTW; ++t) b0[t] = a.wp[((size_t)(wn * NTW + t) * S4 + s4 + 1) * 64 + lane];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float av[MTW];
#pragma unroll
        for (int m = 0; m < MTW; ++m) av[m] = fa[m * 16 * STR + (s4 * 4 + i) * 4];
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
          const float bv = i == 0 ? b[t].x : (i == 1 ? b[t].y : (i == 2 ? b[t].z : b[t].w));
#pragma unroll
          for (int m = 0; m < MTW; ++m) acc[m][t] = mfma16(av[m], bv, acc[m][t]);
        }
      }
    }
  }

  // ---- phase 3: softmax over the neighbours + weighted sum: lane (lr, lg) holds the 16 neighbours of unit 4 wm + lg (tile
  // m, register r = neighbour 4 m + r) for column 16 (wn NTW + t) + lr
  const float pinf = fast_pinf();
  constexpr float LOG2E = 1.4426950408889634f;
  const int unit = wm * 4 + lg;  // unit of this lane inside the workgroup
  const float* fcol = &F[(wm * 64 + lg * 4) * STR];
#pragma unroll
  for (int t = 0; t < NTW; ++t) {
    const int col = (wn * NTW + t) * 16 + lr;
    float mx = acc[0][t][0];
#pragma unroll
    for (int m = 0; m < MTW; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (m + r > 0) mx = fast_max(mx, acc[m][t][r], pinf);
    if constexpr (KP == 32) {  // the other half of the neighbourhood sits in lane group lg ^ 1
      float p, q;
      xgroup_pair16(mx, p, q);
      mx = fast_max(p, q, pinf);
    }
    const float ml = mx * LOG2E;
    float num = 0.f, den = 0.f;
#pragma unroll
    for (int m = 0; m < MTW; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(acc[m][t][r], LOG2E, -ml));
        const float f = fcol[(m * 16 + r) * STR + col];
        num = __builtin_fmaf(p, f, num);
        den += p;
      }
    if constexpr (KP == 32) {
      float p, q;
      xgroup_pair16(num, p, q); num = p + q;
      xgroup_pair16(den, p, q); den = p + q;
    }
    // output row of this lane.  PACK2: unit = centre pair, its tile's 16 columns are 2 x 8 channels = 16 consecutive floats of out
    unsigned orow, ocol;
    bool ok;
    if constexpr (PACK2) {
      const unsigned pair = (unsigned)(KP == 32 ? unit >> 1 : unit);
      orow = c0 + 2u * pair;
      ok = orow + (unsigned)(lr >> 3) < n;
      ocol = (unsigned)lr;
    } else {
      orow = c0 + (unsigned)(KP == 32 ? unit >> 1 : unit);
      ok = orow < n && col < CH;
      ocol = (unsigned)col;
    }
    if (KP == 32 && (lg & 1)) ok = false;  // (both halves hold the result: the even group stores it)
    if (ok) io_store1_b<IOH>(a.out, orow * (unsigned)(CH * 4) + ocol * 4u, num * __builtin_amdgcn_rcpf(den + 1e-16f));
  }
}

// fast path: the caller promises complete neighbourhoods (flags bit 0), K fills its MFMA tiles exactly, every byte offset
// fits 32 bits and LeakyReLU can be written max(v, slope v)
static inline bool lfa_full_ok(const LfaArgs& a, int flags) {
  const int64_t lim = (int64_t)1 << 31;
  return (flags & 1) && (a.K == 16 || a.K == 32) && a.n * a.K < lim && a.n * (int64_t)a.CH * 4 < lim && a.n * 16 < lim &&
         a.slope >= 0.f && a.slope <= 1.f;
}

template <int CH, bool BF, bool IOH>
static int launch_lfa_fwd_io(const LfaArgs& a, hipStream_t st, int flags) {
  constexpr int ROWS = LfaCfg<(CH < 16 ? 16 : CH)>::ROWS;
  if (lfa_full_ok(a, flags)) {
    constexpr int FROWS = LfaFullRows<CH>::ROWS;
    const dim3 g16((unsigned)m3d_cdiv(a.n, FROWS / 16)), g32((unsigned)m3d_cdiv(a.n, FROWS / 32));
    if constexpr (BF) {
      if (flags & 2) {  // split-bf16 operands: att_w_packed holds the hi fragments, then the lo fragments
        if (a.K == 16) hipLaunchKernelGGL((lfa_fwd_full_kernel<CH, 16, 2, IOH>), g16, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((lfa_fwd_full_kernel<CH, 32, 2, IOH>), g32, dim3(256), 0, st, a);
        return hipGetLastError() == hipSuccess ? M3D_OK : M3D_ERR_LAUNCH;
      }
    }
    if (a.K == 16) hipLaunchKernelGGL((lfa_fwd_full_kernel<CH, 16, BF ? 1 : 0, IOH>), g16, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((lfa_fwd_full_kernel<CH, 32, BF ? 1 : 0, IOH>), g32, dim3(256), 0, st, a);
    if (hipGetLastError() != hipSuccess) return M3D_ERR_LAUNCH;
    return M3D_OK;
  }
  if (BF && (flags & 2)) return M3D_ERR_UNSUPPORTED;  // the split-bf16 product exists in the complete-neighbourhood kernels only
  if (a.K <= 16) {
    hipLaunchKernelGGL((lfa_fwd_kernel<CH, 16, BF, IOH>), dim3((unsigned)m3d_cdiv(a.n, ROWS / 16)), dim3(256), 0, st, a);
  } else {
    hipLaunchKernelGGL((lfa_fwd_kernel<CH, 32, BF, IOH>), dim3((unsigned)m3d_cdiv(a.n, ROWS / 32)), dim3(256), 0, st, a);
  }
  if (hipGetLastError() != hipSuccess) return M3D_ERR_LAUNCH;
  return M3D_OK;
}
// flags: bit 0 = M3D_LFA_FULL, bit 1 = split-bf16 operands, M3D_IO_BF16 = x and out hold bf16
template <int CH, bool BF = false>
static int launch_lfa_fwd(const LfaArgs& a, hipStream_t st, int flags) {
  if (flags & M3D_IO_BF16) return launch_lfa_fwd_io<CH, BF, true>(a, st, flags);
  return launch_lfa_fwd_io<CH, BF, false>(a, st, flags);
}

extern "C" int m3d_lfa_fwd(const float* x, const float* pos4, const int32_t* idx, int64_t n, int32_t K, int32_t CH,
                           const float* enc_w_folded, const float* enc_b_folded, const float* att_w_packed,
                           float slope, float* out, int32_t flags, void* stream) {
  if (n < 0 || K < 1 || CH < 8) return M3D_ERR_INVALID;
  if (n == 0) return M3D_OK;
  if (!x || !pos4 || !idx || !enc_w_folded || !enc_b_folded || !att_w_packed || !out) return M3D_ERR_INVALID;
  if (K > 32) return M3D_ERR_UNSUPPORTED;
  if (CH != 8 && CH != 16 && CH != 32 && CH != 64 && CH != 128 && CH != 256) return M3D_ERR_UNSUPPORTED;
  if ((((uintptr_t)x) & 15) || (((uintptr_t)pos4) & 15) || (((uintptr_t)att_w_packed) & 15)) return M3D_ERR_INVALID;
  LfaArgs a;
  a.x = x; a.pos4 = (const float4*)pos4; a.idx = idx; a.wf = enc_w_folded; a.bf = enc_b_folded;
  a.wp = (const float4*)att_w_packed; a.out = out; a.n = n; a.K = K; a.CH = CH; a.D = CH / 2; a.slope = slope;
  hipStream_t st = (hipStream_t)stream;
  switch (CH) {
    case 8: return launch_lfa_fwd<8>(a, st, flags);
    case 16: return launch_lfa_fwd<16>(a, st, flags);
    case 32: return launch_lfa_fwd<32>(a, st, flags);
    case 64: return launch_lfa_fwd<64>(a, st, flags);
    case 128: return launch_lfa_fwd<128>(a, st, flags);
    default: return launch_lfa_fwd<256>(a, st, flags);
  }
}

extern "C" int m3d_lfa_fwd_bf16(const float* x, const float* pos4, const int32_t* idx, int64_t n, int32_t K, int32_t CH,
                                const float* enc_w_folded, const float* enc_b_folded, const void* att_w_packed_bf16,
                                float slope, float* out, int32_t flags, void* stream) {
  if (n < 0 || K < 1 || CH < 8) return M3D_ERR_INVALID;
  if (n == 0) return M3D_OK;
  if (!x || !pos4 || !idx || !enc_w_folded || !enc_b_folded || !att_w_packed_bf16 || !out) return M3D_ERR_INVALID;
  if (K > 32) return M3D_ERR_UNSUPPORTED;
  if (CH != 32 && CH != 64 && CH != 128 && CH != 256) return M3D_ERR_UNSUPPORTED;
  if ((((uintptr_t)x) & 15) || (((uintptr_t)pos4) & 15) || (((uintptr_t)att_w_packed_bf16) & 15)) return M3D_ERR_INVALID;
  LfaArgs a;
  a.x = x; a.pos4 = (const float4*)pos4; a.idx = idx; a.wf = enc_w_folded; a.bf = enc_b_folded;
  a.wp = (const float4*)att_w_packed_bf16; a.out = out; a.n = n; a.K = K; a.CH = CH; a.D = CH / 2; a.slope = slope;
  hipStream_t st = (hipStream_t)stream;
  switch (CH) {
    case 32: return launch_lfa_fwd<32, true>(a, st, flags);
    case 64: return launch_lfa_fwd<64, true>(a, st, flags);
    case 128: return launch_lfa_fwd<128, true>(a, st, flags);
    default: return launch_lfa_fwd<256, true>(a, st, flags);
  }
}

// W_att [CH,CH] fp32 -> bf16 operand fragments of v_mfma_f32_16x16x32_bf16 (CH a multiple of 32):
//   packed  [NT][KS][64][8]: element i of lane l = W[16 nt + (l & 15)][32 ks + 8 (l >> 4) + i]     (B of F * W^T)
//   packed_t, same shape:     element i of lane l = W[32 ks + 8 (l >> 4) + i][16 nt + (l & 15)]     (B of dA * W)
__global__ __launch_bounds__(256) void lfa_pack_att_bf16_kernel(const float* __restrict__ w, int CH,
                                                                unsigned short* __restrict__ packed,
                                                                unsigned short* __restrict__ packed_t) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= CH * CH) return;
  const int KS = CH / 32;
  const int i = t & 7, lane = (t >> 3) & 63, ks = (t >> 9) % KS, nt = (t >> 9) / KS;
  const int r = 16 * nt + (lane & 15), c = 32 * ks + 8 * (lane >> 4) + i;
  packed[t] = (unsigned short)(pack_bf16(w[r * CH + c], 0.f) & 0xffffu);
  if (packed_t) packed_t[t] = (unsigned short)(pack_bf16(w[c * CH + r], 0.f) & 0xffffu);
}

extern "C" int m3d_lfa_pack_att_bf16(const float* w, int32_t CH, void* packed, void* packed_t, void* stream) {
  if (CH < 32 || !w || !packed) return M3D_ERR_INVALID;
  if (CH % 32) return M3D_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(lfa_pack_att_bf16_kernel, dim3((CH * CH + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, CH,
                     (unsigned short*)packed, (unsigned short*)packed_t);
  M3D_CHECK_LAUNCH();
  return M3D_OK;
}

// ------------------------------------------------------------------------------------------
// W_att [CH,CH] -> MFMA B-fragment order (layout: include/m3d_hip.h), for W and for W^T in one launch
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lfa_pack_att_kernel(const float* __restrict__ w, int CH, int CHP,
                                                           float* __restrict__ packed, float* __restrict__ packed_t) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= CHP * CHP) return;
  const int S4 = CHP / 16;
  const int i = t & 3, lane = (t >> 2) & 63, s4 = (t >> 8) % S4, nt = (t >> 8) / S4;
  const int row = 16 * nt + (lane & 15), col = 4 * (4 * s4 + i) + (lane >> 4);
  const bool ok = row < CH && col < CH;
  packed[t] = ok ? w[row * CH + col] : 0.f;
  if (packed_t) packed_t[t] = ok ? w[col * CH + row] : 0.f;
}

extern "C" int m3d_lfa_pack_att(const float* w, int32_t CH, float* packed, float* packed_t, void* stream) {
  if (CH < 1 || !w || !packed) return M3D_ERR_INVALID;
  const int CHP = CH < 16 ? 16 : CH;
  if (CHP % 16) return M3D_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(lfa_pack_att_kernel, dim3((CHP * CHP + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, CH, CHP,
                     packed, packed_t);
  M3D_CHECK_LAUNCH();
  return M3D_OK;
}

// ------------------------------------------------------------------------------------------
// moments of r over all valid edges: mom[0:10] = sum r, mom[10:65] = sum r_p r_q (p<=q, row-major upper)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void lfa_moments_body(const float4* __restrict__ pos4, const int32_t* __restrict__ idx,
                                                 int64_t n, int K, double* __restrict__ mom, int64_t blk, int64_t nblk) {
  __shared__ double red[4][65];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  double acc[65];
#pragma unroll
  for (int q = 0; q < 65; ++q) acc[q] = 0.0;
  for (int64_t i = blk * 256 + tid; i < n; i += nblk * 256) {
    const float4 pi = pos4[i];
    // 8 neighbours per trip: ids first, then all 8 positions in flight (unconditional loads from a clamped id; a
    // branch around the load serialises one memory round trip per neighbour)
    for (int k0 = 0; k0 < K; k0 += 8) {
      int jj[8];
      float4 pj[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) jj[u] = k0 + u < K ? idx[i * K + k0 + u] : -1;
#pragma unroll
      for (int u = 0; u < 8; ++u) pj[u] = pos4[jj[u] < 0 ? 0 : jj[u]];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        float r[10];
        rel_pos(pi, pj[u], r);
        const float m = jj[u] < 0 ? 0.f : 1.f;  // missing neighbour: contributes nothing
#pragma unroll
        for (int p = 0; p < 10; ++p) r[p] *= m;
        int o = 10;
#pragma unroll
        for (int p = 0; p < 10; ++p) {
          acc[p] += (double)r[p];
#pragma unroll
          for (int q = p; q < 10; ++q) acc[o++] += (double)r[p] * (double)r[q];
        }
      }
    }
  }
#pragma unroll
  for (int q = 0; q < 65; ++q) {
    double v = wave_sum_d(acc[q]);
    if (lane == 0) red[wid][q] = v;
  }
  __syncthreads();
  if (tid < 65) atomicAdd(&mom[tid], red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid]);
}

__global__ __launch_bounds__(256) void lfa_moments_kernel(const float4* __restrict__ pos4,
                                                          const int32_t* __restrict__ idx, int64_t n, int K,
                                                          double* __restrict__ mom) {
  lfa_moments_body(pos4, idx, n, K, mom, blockIdx.x, gridDim.x);
}

// the encoder moments of every resolution level in one launch (m3d_lfa_moments_batch): the deep levels are a few
// thousand points each — four latency-bound launches of 20-40 us one after the other otherwise
#define MOM_BATCH_MAX 8
struct MomBatch {
  const float4* pos4[MOM_BATCH_MAX];
  const int32_t* idx[MOM_BATCH_MAX];
  int64_t n[MOM_BATCH_MAX];
  unsigned wg_start[MOM_BATCH_MAX + 1];
  int njobs;
};
__global__ __launch_bounds__(256) void lfa_moments_batch_kernel(MomBatch a, int K, double* __restrict__ mom, int64_t ldm) {
  int j = 0;
#pragma unroll
  for (int i = 1; i < MOM_BATCH_MAX; ++i) j += (i < a.njobs && blockIdx.x >= a.wg_start[i]) ? 1 : 0;
  lfa_moments_body(a.pos4[j], a.idx[j], a.n[j], K, mom + (size_t)j * ldm, (int64_t)(blockIdx.x - a.wg_start[j]),
                   (int64_t)(a.wg_start[j + 1] - a.wg_start[j]));
}

// Zero fill by a kernel, not by hipMemsetAsync: captured into a hipGraph, the 520-byte memset of the 65 moments (not a
// multiple of 16 bytes) left the LAST double uninitialised on replay (ROCm 7.0; found by the dual-graph parity test of
// round 3: the encoder's running variance turned inf after the third step) — every other memset in this library is a
// multiple of 16 bytes, the eager path was never affected.
__global__ void zero_f64_kernel(double* __restrict__ p, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0.0;
}

extern "C" int m3d_lfa_moments_batch(int32_t njobs, const float* const* pos4, const int32_t* const* idx,
                                     const int64_t* n, int32_t K, double* mom, int64_t mom_stride, void* stream) {
  if (njobs < 0 || njobs > MOM_BATCH_MAX) return M3D_ERR_UNSUPPORTED;
  if (njobs == 0) return M3D_OK;
  if (!pos4 || !idx || !n || !mom || K < 1 || mom_stride < 65) return M3D_ERR_INVALID;
  hipStream_t st = (hipStream_t)stream;
  {
    const int tot = (int)(njobs * mom_stride);
    hipLaunchKernelGGL(zero_f64_kernel, dim3((tot + 255) / 256), dim3(256), 0, st, mom, tot);
  }
  MomBatch a;
  a.njobs = njobs;
  unsigned total = 0;
  for (int j = 0; j < MOM_BATCH_MAX; ++j) {
    a.wg_start[j] = total;
    if (j < njobs) {
      if (n[j] < 0 || (n[j] > 0 && (!pos4[j] || !idx[j]))) return M3D_ERR_INVALID;
      a.pos4[j] = (const float4*)pos4[j]; a.idx[j] = idx[j]; a.n[j] = n[j];
      int64_t gx = m3d_cdiv(n[j], 256);
      total += (unsigned)(gx > 512 ? 512 : gx);
    } else {
      a.pos4[j] = nullptr; a.idx[j] = nullptr; a.n[j] = 0;
    }
  }
  a.wg_start[MOM_BATCH_MAX] = total;
  if (total == 0) return M3D_OK;
  hipLaunchKernelGGL(lfa_moments_batch_kernel, dim3(total), dim3(256), 0, st, a, K, mom, mom_stride);
  M3D_CHECK_LAUNCH();
  return M3D_OK;
}

extern "C" int m3d_lfa_moments(const float* pos4, const int32_t* idx, int64_t n, int32_t K, double* mom65,
                               void* stream) {
  if (n < 0 || K < 1) return M3D_ERR_INVALID;
  if (!mom65) return M3D_ERR_INVALID;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(zero_f64_kernel, dim3(1), dim3(128), 0, st, mom65, 65);
  if (n == 0) return M3D_OK;
  if (!pos4 || !idx) return M3D_ERR_INVALID;
  int64_t gx = m3d_cdiv(n, 256);
  if (gx > 512) gx = 512;
  hipLaunchKernelGGL(lfa_moments_kernel, dim3((unsigned)gx), dim3(256), 0, st, (const float4*)pos4, idx, n, K, mom65);
  M3D_CHECK_LAUNCH();
  return M3D_OK;
}

__device__ __forceinline__ double mom2(const double* mom, int p, int q) {
  if (p > q) { int t = p; p = q; q = t; }
  // offset of (p,q), p<=q, in the row-major upper triangle of a 10x10 matrix
  return mom[10 + p * 10 - (p * (p - 1)) / 2 + (q - p)];
}

// Train-mode fold of mlp_encoder's BatchNorm: z = W r + b is affine in r, so over the E edges
//   mean_c = w_c . m + b_c,  var_c = w_c^T (M2/E - m m^T) w_c      (m = S1/E)
// -> folded weights  wf = scale_c * w_c,  bf = scale_c*(b_c - mean_c) + beta_c,  scale_c = gamma_c/sqrt(var_c+eps)
// In eval mode (mom == nullptr) the running statistics are used instead.
// one wave per encoder channel (the serial 10x10 fp64 loop of a one-thread-per-channel kernel cost ~10 us)
__global__ __launch_bounds__(64) void lfa_enc_finalize_kernel(const double* __restrict__ mom, double E,
                                                              const float* __restrict__ w, const float* __restrict__ b,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float eps, float momentum,
                                                              float* running_mean, float* running_var, float* wf,
                                                              float* bf, float* mean_out, float* invstd_out, int D) {
  const int c = blockIdx.x, lane = threadIdx.x;
  if (c >= D) return;
  double mean, var;
  if (mom) {
    // lane l: terms (p, q) = l, l + 64 of  w^T (M2/E - m m^T) w;  lanes < 10 also the mean term w_p m_p
    double pv = 0.0, pm = 0.0;
    for (int t = lane; t < 100; t += 64) {
      const int p = t / 10, q = t % 10;
      pv += (double)w[c * 10 + p] * (double)w[c * 10 + q] * (mom2(mom, p, q) / E - (mom[p] / E) * (mom[q] / E));
    }
    if (lane < 10) pm = (double)w[c * 10 + lane] * (mom[lane] / E);
    var = wave_sum_d(pv);
    mean = wave_sum_d(pm) + (double)b[c];
    if (var < 0.0) var = 0.0;
  } else {
    mean = (double)running_mean[c];
    var = (double)running_var[c];
  }
  const double invstd = 1.0 / sqrt(var + (double)eps);
  const double sc = (double)gamma[c] * invstd;
  if (lane < 10) wf[c * 10 + lane] = (float)(sc * (double)w[c * 10 + lane]);
  if (lane != 0) return;
  if (mom) {
    if (running_mean) running_mean[c] = (float)((1.0 - momentum) * (double)running_mean[c] + momentum * mean);
    if (running_var) {
      double unb = E > 1.0 ? var * E / (E - 1.0) : var;
      running_var[c] = (float)((1.0 - momentum) * (double)running_var[c] + momentum * unb);
    }
  }
  bf[c] = (float)(sc * ((double)b[c] - mean) + (double)beta[c]);
  if (mean_out) mean_out[c] = (float)mean;
  if (invstd_out) invstd_out[c] = (float)invstd;
}

extern "C" int m3d_lfa_enc_finalize(const double* mom65, int64_t num_edges, const float* w, const float* b,
                                    const float* gamma, const float* beta, float eps, float momentum,
                                    float* running_mean, float* running_var, float* w_folded, float* b_folded,
                                    float* mean_out, float* invstd_out, int32_t D, void* stream) {
  if (D < 0) return M3D_ERR_INVALID;
  if (D == 0) return M3D_OK;
  if (!w || !b || !gamma || !beta || !w_folded || !b_folded) return M3D_ERR_INVALID;
  if (mom65 && num_edges < 1) return M3D_ERR_INVALID;
  if (!mom65 && (!running_mean || !running_var)) return M3D_ERR_INVALID;
  hipLaunchKernelGGL(lfa_enc_finalize_kernel, dim3(D), dim3(64), 0, (hipStream_t)stream, mom65,
                     (double)num_edges, w, b, gamma, beta, eps, momentum, running_mean, running_var, w_folded,
                     b_folded, mean_out, invstd_out, D);
  M3D_CHECK_LAUNCH();
  return M3D_OK;
}

// Everything one LocalFeatureAggregation needs before its fused kernels, in ONE launch (the training step is a chain
// of ~400 graph nodes and every tiny node costs 5-9 us of it): the encoder fold of m3d_lfa_enc_finalize (blocks
// [0, D)) and the attention-weight fragments of m3d_lfa_pack_att — fp32 W / W^T, or with bf16 != 0 the bf16 operand
// fragments of m3d_lfa_pack_att_bf16 — in the remaining blocks.
__device__ __forceinline__ void lfa_prepare_body(const double* __restrict__ mom, double E,
                                                 const float* __restrict__ w, const float* __restrict__ b,
                                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                                 float eps, float momentum, float* running_mean,
                                                 float* running_var, float* wf, float* bf, float* mean_out,
                                                 float* invstd_out, int D, const float* __restrict__ w_att, int CH,
                                                 int CHP, void* packed, void* packed_t, int bf16, int blk) {
  const int lane = threadIdx.x;
  if (blk >= D) {
    const int t = (blk - D) * 64 + lane;
    if (bf16) {
      if (t >= CH * CH) return;
      const int KS = CH / 32;
      const int i = t & 7, ln = (t >> 3) & 63, ks = (t >> 9) % KS, nt = (t >> 9) / KS;
      const int r = 16 * nt + (ln & 15), c = 32 * ks + 8 * (ln >> 4) + i;
      const float v = w_att[r * CH + c], vt = w_att[c * CH + r];
      const unsigned h = pack_bf16(v, 0.f) & 0xffffu, ht = pack_bf16(vt, 0.f) & 0xffffu;
      ((unsigned short*)packed)[t] = (unsigned short)h;
      if (packed_t) ((unsigned short*)packed_t)[t] = (unsigned short)ht;
      if (bf16 == 2) {  // split-bf16: the lo fragments (bf16 of the rounding remainder) behind the CH * CH hi values
        ((unsigned short*)packed)[CH * CH + t] = (unsigned short)(pack_bf16(v - __uint_as_float(h << 16), 0.f) & 0xffffu);
        if (packed_t)
          ((unsigned short*)packed_t)[CH * CH + t] = (unsigned short)(pack_bf16(vt - __uint_as_float(ht << 16), 0.f) & 0xffffu);
      }
    } else {
      if (t >= CHP * CHP) return;
      const int S4 = CHP / 16;
      const int i = t & 3, ln = (t >> 2) & 63, s4 = (t >> 8) % S4, nt = (t >> 8) / S4;
      const int row = 16 * nt + (ln & 15), col = 4 * (4 * s4 + i) + (ln >> 4);
      const bool ok = row < CH && col < CH;
      ((float*)packed)[t] = ok ? w_att[row * CH + col] : 0.f;
      if (packed_t) ((float*)packed_t)[t] = ok ? w_att[col * CH + row] : 0.f;
    }
    return;
  }
  const int c = blk;
  double mean, var;
  if (mom) {
    double pv = 0.0, pm = 0.0;
    for (int t = lane; t < 100; t += 64) {
      const int p = t / 10, q = t % 10;
      pv += (double)w[c * 10 + p] * (double)w[c * 10 + q] * (mom2(mom, p, q) / E - (mom[p] / E) * (mom[q] / E));
    }
    if (lane < 10) pm = (double)w[c * 10 + lane] * (mom[lane] / E);
    var = wave_sum_d(pv);
    mean = wave_sum_d(pm) + (double)b[c];
    if (var < 0.0) var = 0.0;
  } else {
    mean = (double)running_mean[c];
    var = (double)running_var[c];
  }
  const double invstd = 1.0 / sqrt(var + (double)eps);
  const double sc = (double)gamma[c] * invstd;
  if (lane < 10) wf[c * 10 + lane] = (float)(sc * (double)w[c * 10 + lane]);
  if (lane != 0) return;
  if (mom) {
    if (running_mean) running_mean[c] = (float)((1.0 - momentum) * (double)running_mean[c] + momentum * mean);
    if (running_var) {
      double unb = E > 1.0 ? var * E / (E - 1.0) : var;
      running_var[c] = (float)((1.0 - momentum) * (double)running_var[c] + momentum * unb);
    }
  }
  bf[c] = (float)(sc * ((double)b[c] - mean) + (double)beta[c]);
  if (mean_out) mean_out[c] = (float)mean;
  if (invstd_out) invstd_out[c] = (float)invstd;
}

__global__ __launch_bounds__(64) void lfa_prepare_kernel(const double* __restrict__ mom, double E,
                                                         const float* __restrict__ w, const float* __restrict__ b,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         float eps, float momentum, float* running_mean,
                                                         float* running_var, float* wf, float* bf, float* mean_out,
                                                         float* invstd_out, int D, const float* __restrict__ w_att, int CH,
                                                         int CHP, void* packed, void* packed_t, int bf16) {
  lfa_prepare_body(mom, E, w, b, gamma, beta, eps, momentum, running_mean, running_var, wf, bf, mean_out, invstd_out, D,
                   w_att, CH, CHP, packed, packed_t, bf16, (int)blockIdx.x);
}

// m3d_lfa_prepare of several LFA layers in ONE launch (m3d_lfa_prepare_batch): with the position-only tables of a step
// prefetched, the encoder moments of all levels exist when the forward pass starts, and the eight 5-us launches that sat in
// the chain in front of every lfa_fwd become one at its head
#define LFA_PREP_BATCH_MAX 8
struct LfaPrepBatch {
  const double* mom[LFA_PREP_BATCH_MAX]; double E[LFA_PREP_BATCH_MAX];
  const float* w[LFA_PREP_BATCH_MAX]; const float* b[LFA_PREP_BATCH_MAX];
  const float* gamma[LFA_PREP_BATCH_MAX]; const float* beta[LFA_PREP_BATCH_MAX];
  float* running_mean[LFA_PREP_BATCH_MAX]; float* running_var[LFA_PREP_BATCH_MAX];
  float* wf[LFA_PREP_BATCH_MAX]; float* bf[LFA_PREP_BATCH_MAX]; float* mean[LFA_PREP_BATCH_MAX]; float* invstd[LFA_PREP_BATCH_MAX];
  const float* w_att[LFA_PREP_BATCH_MAX]; void* packed[LFA_PREP_BATCH_MAX]; void* packed_t[LFA_PREP_BATCH_MAX];
  int D[LFA_PREP_BATCH_MAX], CH[LFA_PREP_BATCH_MAX], bf16[LFA_PREP_BATCH_MAX];
  unsigned wg_start[LFA_PREP_BATCH_MAX + 1];
  int njobs;
  float eps, momentum;
};
__global__ __launch_bounds__(64) void lfa_prepare_batch_kernel(LfaPrepBatch a) {
  int j = 0;
#pragma unroll
  for (int i = 1; i < LFA_PREP_BATCH_MAX; ++i) j += (i < a.njobs && blockIdx.x >= a.wg_start[i]) ? 1 : 0;
  const int CH = a.CH[j], CHP = CH < 16 ? 16 : CH;
  lfa_prepare_body(a.mom[j], a.E[j], a.w[j], a.b[j], a.gamma[j], a.beta[j], a.eps, a.momentum, a.running_mean[j],
                   a.running_var[j], a.wf[j], a.bf[j], a.mean[j], a.invstd[j], a.D[j], a.w_att[j], CH, CHP, a.packed[j],
                   a.packed_t[j], a.bf16[j], (int)(blockIdx.x - a.wg_start[j]));
}

extern "C" int m3d_lfa_prepare_batch(int32_t njobs, const double* const* mom65, const int64_t* num_edges,
                                     const float* const* w, const float* const* b, const float* const* gamma,
                                     const float* const* beta, float eps, float momentum, float* const* running_mean,
                                     float* const* running_var, float* const* w_folded, float* const* b_folded,
                                     float* const* mean_out, float* const* invstd_out, const int32_t* D,
                                     const float* const* w_att, const int32_t* CH, void* const* packed,
                                     void* const* packed_t, const int32_t* bf16, void* stream) {
  if (njobs < 0 || njobs > LFA_PREP_BATCH_MAX) return M3D_ERR_UNSUPPORTED;
  if (njobs == 0) return M3D_OK;
  if (!mom65 || !num_edges || !w || !b || !gamma || !beta || !running_mean || !running_var || !w_folded || !b_folded ||
      !mean_out || !invstd_out || !D || !w_att || !CH || !packed || !packed_t || !bf16)
    return M3D_ERR_INVALID;
  LfaPrepBatch a;
  a.njobs = njobs; a.eps = eps; a.momentum = momentum;
  unsigned total = 0;
  for (int j = 0; j < LFA_PREP_BATCH_MAX; ++j) {
    a.wg_start[j] = total;
    if (j >= njobs) {
      a.mom[j] = nullptr; a.E[j] = 1.0; a.w[j] = a.b[j] = a.gamma[j] = a.beta[j] = a.w_att[j] = nullptr;
      a.running_mean[j] = a.running_var[j] = a.wf[j] = a.bf[j] = a.mean[j] = a.invstd[j] = nullptr;
      a.packed[j] = a.packed_t[j] = nullptr; a.D[j] = 1; a.CH[j] = 16; a.bf16[j] = 0;
      continue;
    }
    if (D[j] < 1 || CH[j] < 1 || !mom65[j] || num_edges[j] < 1 || !w[j] || !b[j] || !gamma[j] || !beta[j] || !w_folded[j] ||
        !b_folded[j] || !w_att[j] || !packed[j])
      return M3D_ERR_INVALID;
    const int CHP = CH[j] < 16 ? 16 : CH[j];
    if (CHP % 16) return M3D_ERR_UNSUPPORTED;
    if (bf16[j] && (CH[j] % 32)) return M3D_ERR_UNSUPPORTED;
    a.mom[j] = mom65[j]; a.E[j] = (double)num_edges[j]; a.w[j] = w[j]; a.b[j] = b[j]; a.gamma[j] = gamma[j]; a.beta[j] = beta[j];
    a.running_mean[j] = running_mean[j]; a.running_var[j] = running_var[j]; a.wf[j] = w_folded[j]; a.bf[j] = b_folded[j];
    a.mean[j] = mean_out[j]; a.invstd[j] = invstd_out[j]; a.w_att[j] = w_att[j]; a.packed[j] = packed[j];
    a.packed_t[j] = packed_t[j]; a.D[j] = D[j]; a.CH[j] = CH[j]; a.bf16[j] = bf16[j];
    const int elems = bf16[j] ? CH[j] * CH[j] : CHP * CHP;
    total += (unsigned)(D[j] + (elems + 63) / 64);
  }
  a.wg_start[LFA_PREP_BATCH_MAX] = total;
  for (int j = njobs; j < LFA_PREP_BATCH_MAX; ++j) a.wg_start[j] = total;
  hipLaunchKernelGGL(lfa_prepare_batch_kernel, dim3(total), dim3(64), 0, (hipStream_t)stream, a);
  M3D_CHECK_LAUNCH();
  return M3D_OK;
}

extern "C" int m3d_lfa_prepare(const double* mom65, int64_t num_edges, const float* w, const float* b,
                               const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                               float* running_var, float* w_folded, float* b_folded, float* mean_out,
                               float* invstd_out, int32_t D, const float* w_att, int32_t CH, void* packed,
                               void* packed_t, int32_t bf16, void* stream) {
  if (D < 1 || CH < 1) return M3D_ERR_INVALID;
  if (!w || !b || !gamma || !beta || !w_folded || !b_folded || !w_att || !packed) return M3D_ERR_INVALID;
  if (mom65 && num_edges < 1) return M3D_ERR_INVALID;
  if (!mom65 && (!running_mean || !running_var)) return M3D_ERR_INVALID;
  const int CHP = CH < 16 ? 16 : CH;
  if (CHP % 16) return M3D_ERR_UNSUPPORTED;
  if (bf16 && (CH % 32)) return M3D_ERR_UNSUPPORTED;
  const int elems = bf16 ? CH * CH : CHP * CHP;
  hipLaunchKernelGGL(lfa_prepare_kernel, dim3(D + (elems + 63) / 64), dim3(64), 0, (hipStream_t)stream, mom65,
                     (double)num_edges, w, b, gamma, beta, eps, momentum, running_mean, running_var, w_folded, b_folded,
                     mean_out, invstd_out, D, w_att, CH, CHP, packed, packed_t, bf16);
  M3D_CHECK_LAUNCH();
  return M3D_OK;
}

// ------------------------------------------------------------------------------------------
// unfused pieces (used by the backward pass; also a fallback / cross-check of the fused forward)
// ------------------------------------------------------------------------------------------
// F[e, 0:D] = x[j_e], F[e, D:CH] = LeakyReLU(wf r_e + bf);  rows of missing neighbours are zero.  e = i*K + k
__global__ __launch_bounds__(256) void edge_features_kernel(LfaArgs a, float* __restrict__ Fout) {
  const int C4 = a.CH >> 2, D4 = a.D >> 2;
  const int64_t total = a.n * a.K * C4;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
    const int64_t e = t / C4;
    const int c4 = (int)(t % C4);
    const int j = a.idx[e];
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (j >= 0) {
      if (c4 < D4) {
        v = *(const float4*)(a.x + (int64_t)j * a.D + c4 * 4);
      } else {
        float r[10];
        rel_pos(a.pos4[e / a.K], a.pos4[j], r);
        float o[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int c = (c4 - D4) * 4 + u;
          const float* w = a.wf + c * 10;
          float s = a.bf[c];
#pragma unroll
          for (int q = 0; q < 10; ++q) s += w[q] * r[q];
          o[u] = lrelu(s, a.slope);
        }
        v = make_float4(o[0], o[1], o[2], o[3]);
      }
    }
    ((float4*)Fout)[t] = v;
  }
}

extern "C" int m3d_lfa_edge_features(const float* x, const float* pos4, const int32_t* idx, int64_t n, int32_t K,
                                     int32_t CH, const float* enc_w_folded, const float* enc_b_folded, float slope,
                                     float* F, void* stream) {
  if (n < 0 || K < 1 || CH < 8 || (CH % 8)) return M3D_ERR_INVALID;
  if (n == 0) return M3D_OK;
  if (!x || !pos4 || !idx || !enc_w_folded || !enc_b_folded || !F) return M3D_ERR_INVALID;
  LfaArgs a;
  a.x = x; a.pos4 = (const float4*)pos4; a.idx = idx; a.wf = enc_w_folded; a.bf = enc_b_folded; a.wp = nullptr;
  a.out = nullptr; a.n = n; a.K = K; a.CH = CH; a.D = CH / 2; a.slope = slope;
  int64_t gx = m3d_cdiv(n * K * (CH / 4), 256);
  if (gx > 16384) gx = 16384;
  hipLaunchKernelGGL(edge_features_kernel, dim3((unsigned)gx), dim3(256), 0, (hipStream_t)stream, a, F);
  M3D_CHECK_LAUNCH();
  return M3D_OK;
}

// thread = (centre i, channel c).  mode 0: out[i,c] = sum_k softmax_k(A)[k,c] * F[k,c]
// mode 1 (backward): additionally, given dout[i,c]:
//    dA[e,c] = s * dout * (F - out)   (written over A),   dF[e,c] = dout * s
__global__ __launch_bounds__(256) void edge_softmax_kernel(float* __restrict__ A, const float* __restrict__ Fm,
                                                           const int32_t* __restrict__ idx, int64_t n, int K, int CH,
                                                           const float* __restrict__ dout, float* __restrict__ out,
                                                           float* __restrict__ dF, int mode) {
  const int64_t total = n * CH;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
    const int64_t i = t / CH;
    const int c = (int)(t % CH);
    const int32_t* nb = idx + i * K;
    const int64_t base = i * K * (int64_t)CH + c;
    float mx = -__builtin_inff();
    for (int k = 0; k < K; ++k)
      if (nb[k] >= 0) mx = fmaxf(mx, A[base + (int64_t)k * CH]);
    float num = 0.f, den = 0.f;
    for (int k = 0; k < K; ++k)
      if (nb[k] >= 0) {
        float p = __expf(A[base + (int64_t)k * CH] - mx);
        num += p * Fm[base + (int64_t)k * CH];
        den += p;
      }
    const float inv = 1.f / (den + 1e-16f);
    const float o = num * inv;
    if (mode == 0) {
      out[t] = o;
    } else {
      const float g = dout[t];
      for (int k = 0; k < K; ++k) {
        const int64_t p_ = base + (int64_t)k * CH;
        float da = 0.f, df = 0.f;
        if (nb[k] >= 0) {
          float s = __expf(A[p_] - mx) * inv;
          da = s * g * (Fm[p_] - o);
          df = g * s;
        }
        A[p_] = da;
        dF[p_] = df;
      }
    }
  }
}

extern "C" int m3d_lfa_edge_softmax_fwd(const float* A, const float* F, const int32_t* idx, int64_t n, int32_t K,
                                        int32_t CH, float* out, void* stream) {
  if (n < 0 || K < 1 || CH < 1) return M3D_ERR_INVALID;
  if (n == 0) return M3D_OK;
  if (!A || !F || !idx || !out) return M3D_ERR_INVALID;
  int64_t gx = m3d_cdiv(n * CH, 256);
  if (gx > 16384) gx = 16384;
  hipLaunchKernelGGL(edge_softmax_kernel, dim3((unsigned)gx), dim3(256), 0, (hipStream_t)stream, (float*)A, F, idx, n,
                     K, CH, nullptr, out, nullptr, 0);
  M3D_CHECK_LAUNCH();
  return M3D_OK;
}

extern "C" int m3d_lfa_edge_softmax_bwd(float* A_inout_dA, const float* F, const int32_t* idx, int64_t n, int32_t K,
                                        int32_t CH, const float* dout, float* dF, void* stream) {
  if (n < 0 || K < 1 || CH < 1) return M3D_ERR_INVALID;
  if (n == 0) return M3D_OK;
  if (!A_inout_dA || !F || !idx || !dout || !dF) return M3D_ERR_INVALID;
  int64_t gx = m3d_cdiv(n * CH, 256);
  if (gx > 16384) gx = 16384;
  hipLaunchKernelGGL(edge_softmax_kernel, dim3((unsigned)gx), dim3(256), 0, (hipStream_t)stream, A_inout_dA, F, idx, n,
                     K, CH, dout, nullptr, dF, 1);
  M3D_CHECK_LAUNCH();
  return M3D_OK;
}

// Backward of edge_features: given dF[E, CH]
//   dx[j_e, 0:D] += dF[e, 0:D]                                          (atomic scatter)
//   dy[e, c] = dF[e, D+c] * lrelu'(wf_c r_e + bf_c);  G[c, 0:10] += dy * r_e,  G[c, 10] += dy   (fp64)
#define EFB 256
__global__ __launch_bounds__(256) void edge_features_bwd_kernel(LfaArgs a, const float* __restrict__ dF,
                                                                float* __restrict__ dx, double* __restrict__ G) {
  __shared__ __attribute__((aligned(16))) float rt[EFB][12];
  __shared__ int nb[EFB];
  __shared__ double gred[256 * 11];
  const int tid = threadIdx.x;
  const int D = a.D, CH = a.CH, D4 = D >> 2;
  const int64_t E = a.n * a.K;
  // G part: thread owns channel c = tid % Dg of group grp; channels beyond 256 are looped (D <= 128 in this net)
  const int Dg = D < 256 ? D : 256;
  const int ng = 256 / Dg;
  const int c = tid % Dg, grp = tid / Dg;
  const bool gthread = grp < ng;
  float w[10], bias = 0.f;
#pragma unroll
  for (int q = 0; q < 10; ++q) w[q] = gthread ? a.wf[c * 10 + q] : 0.f;
  if (gthread) bias = a.bf[c];
  double g[11];
#pragma unroll
  for (int q = 0; q < 11; ++q) g[q] = 0.0;

  for (int64_t e0 = (int64_t)blockIdx.x * EFB; e0 < E; e0 += (int64_t)gridDim.x * EFB) {
    {
      const int64_t e = e0 + tid;
      int j = -1;
      if (e < E) j = a.idx[e];
      float r[10];
#pragma unroll
      for (int q = 0; q < 10; ++q) r[q] = 0.f;
      if (j >= 0) rel_pos(a.pos4[e / a.K], a.pos4[j], r);
#pragma unroll
      for (int q = 0; q < 10; ++q) rt[tid][q] = r[q];
      rt[tid][10] = j >= 0 ? 1.f : 0.f;
      rt[tid][11] = 0.f;
      nb[tid] = j;
    }
    __syncthreads();
    const int cnt = (int)((E - e0) < EFB ? (E - e0) : EFB);
    // scatter dx
    for (int f = tid; f < cnt * D4; f += 256) {
      const int el = f / D4, c4 = f % D4;
      const int j = nb[el];
      if (j >= 0) {
        float4 v = *(const float4*)(dF + (e0 + el) * (int64_t)CH + c4 * 4);
        float* d = dx + (int64_t)j * D + c4 * 4;
        atomicAdd(d + 0, v.x); atomicAdd(d + 1, v.y); atomicAdd(d + 2, v.z); atomicAdd(d + 3, v.w);
      }
    }
    // accumulate G
    if (gthread) {
      float gp[11];
#pragma unroll
      for (int q = 0; q < 11; ++q) gp[q] = 0.f;
      for (int el = grp; el < cnt; el += ng) {
        const float4 r0 = *(const float4*)&rt[el][0], r1 = *(const float4*)&rt[el][4], r2 = *(const float4*)&rt[el][8];
        const float rr[11] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x, r2.y, r2.z};
        float pre = bias;
#pragma unroll
        for (int q = 0; q < 10; ++q) pre += w[q] * rr[q];
        float dy = dF[(e0 + el) * (int64_t)CH + D + c] * (pre > 0.f ? 1.f : a.slope);
#pragma unroll
        for (int q = 0; q < 11; ++q) gp[q] += dy * rr[q];
      }
#pragma unroll
      for (int q = 0; q < 11; ++q) g[q] += (double)gp[q];
    }
    __syncthreads();
  }
#pragma unroll
  for (int q = 0; q < 11; ++q) gred[q * 256 + tid] = g[q];
  __syncthreads();
  if (tid < Dg) {
    for (int q = 0; q < 11; ++q) {
      double v = 0.0;
      for (int u = 0; u < ng; ++u) v += gred[q * 256 + u * Dg + tid];
      atomicAdd(&G[tid * 11 + q], v);
    }
  }
}

extern "C" int m3d_lfa_edge_features_bwd(const float* dF, const float* pos4, const int32_t* idx, int64_t n, int32_t K,
                                         int32_t CH, const float* enc_w_folded, const float* enc_b_folded,
                                         float slope, float* dx, double* G, void* stream) {
  if (n < 0 || K < 1 || CH < 8 || (CH % 8)) return M3D_ERR_INVALID;
  if (CH / 2 > 256) return M3D_ERR_UNSUPPORTED;
  if (!G) return M3D_ERR_INVALID;
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(G, 0, sizeof(double) * 11 * (size_t)(CH / 2), st) != hipSuccess) return M3D_ERR_LAUNCH;
  if (n == 0) return M3D_OK;
  if (!dF || !pos4 || !idx || !enc_w_folded || !enc_b_folded || !dx) return M3D_ERR_INVALID;
  LfaArgs a;
  a.x = nullptr; a.pos4 = (const float4*)pos4; a.idx = idx; a.wf = enc_w_folded; a.bf = enc_b_folded; a.wp = nullptr;
  a.out = nullptr; a.n = n; a.K = K; a.CH = CH; a.D = CH / 2; a.slope = slope;
  int64_t gx = m3d_cdiv(n * K, EFB);
  if (gx > 1024) gx = 1024;
  hipLaunchKernelGGL(edge_features_bwd_kernel, dim3((unsigned)gx), dim3(256), 0, st, a, dF, dx, G);
  M3D_CHECK_LAUNCH();
  return M3D_OK;
}

// Parameter gradients of mlp_encoder (Linear 10->D + train-mode BatchNorm) from G (see lfa.hip header):
//   dbeta = g0, dgamma = invstd*(w.G + (b-mean) g0)
//   dW[c,q] = scale*( G[c,q] - (g0/E) S1[q] - (dgamma/E) * invstd*( sum_p w_p M2[p,q] + (b-mean) S1[q] ) )
//   db = 0 (BatchNorm removes the mean)
__device__ __forceinline__ void lfa_enc_bwd_finalize_body(const double* __restrict__ G, const double* __restrict__ mom,
                                                          double E, const float* __restrict__ w,
                                                          const float* __restrict__ b, const float* __restrict__ gamma,
                                                          const float* __restrict__ mean,
                                                          const float* __restrict__ invstd, float* dw, float* db,
                                                          float* dgamma, float* dbeta, int D, int acc, int c) {
  // one wave per channel; lane q < 10 owns dW[c, q]
  const int lane = threadIdx.x;
  if (c >= D) return;
  const double* g = G + c * 11;
  const double g0 = g[10];
  const double is = (double)invstd[c], mu = (double)mean[c], bb = (double)b[c];
  const double wg = wave_sum_d(lane < 10 ? (double)w[c * 10 + lane] * g[lane] : 0.0);
  const double dgam = is * (wg + (bb - mu) * g0);
  const double sc = (double)gamma[c] * is;
  if (lane < 10) {
    const int q = lane;
    double wm = 0.0;
#pragma unroll
    for (int p = 0; p < 10; ++p) wm += (double)w[c * 10 + p] * mom2(mom, p, q);
    const double zr = is * (wm + (bb - mu) * mom[q]);
    const float dwv = (float)(sc * (g[q] - (g0 / E) * mom[q] - (dgam / E) * zr));
    dw[c * 10 + q] = acc ? dw[c * 10 + q] + dwv : dwv;
  }
  if (lane != 0) return;
  if (!acc) db[c] = 0.f;  // BatchNorm removes the mean: d/d(bias) is exactly 0
  dgamma[c] = acc ? dgamma[c] + (float)dgam : (float)dgam;
  dbeta[c] = acc ? dbeta[c] + (float)g0 : (float)g0;
}

__global__ __launch_bounds__(64) void lfa_enc_bwd_finalize_kernel(const double* __restrict__ G,
                                                                  const double* __restrict__ mom, double E,
                                                                  const float* __restrict__ w, const float* __restrict__ b,
                                                                  const float* __restrict__ gamma,
                                                                  const float* __restrict__ mean,
                                                                  const float* __restrict__ invstd, float* dw, float* db,
                                                                  float* dgamma, float* dbeta, int D, int acc) {
  lfa_enc_bwd_finalize_body(G, mom, E, w, b, gamma, mean, invstd, dw, db, dgamma, dbeta, D, acc, blockIdx.x);
}

// the encoder parameter gradients of several LFA layers in one launch (leaves of the backward pass, see
// lfa_bwd_reduce_batch_kernel): wave w of the flattened grid is channel w - start[j] of job j
#define LFA_FIN_BATCH_MAX 16
struct LfaFinBatch {
  const double* G[LFA_FIN_BATCH_MAX]; const double* mom[LFA_FIN_BATCH_MAX]; double E[LFA_FIN_BATCH_MAX];
  const float* w[LFA_FIN_BATCH_MAX]; const float* b[LFA_FIN_BATCH_MAX]; const float* gamma[LFA_FIN_BATCH_MAX];
  const float* mean[LFA_FIN_BATCH_MAX]; const float* invstd[LFA_FIN_BATCH_MAX];
  float* dw[LFA_FIN_BATCH_MAX]; float* db[LFA_FIN_BATCH_MAX]; float* dgamma[LFA_FIN_BATCH_MAX]; float* dbeta[LFA_FIN_BATCH_MAX];
  int D[LFA_FIN_BATCH_MAX]; unsigned start[LFA_FIN_BATCH_MAX + 1];
  int njobs, acc;
};
__global__ __launch_bounds__(64) void lfa_enc_bwd_finalize_batch_kernel(LfaFinBatch a) {
  int j = 0;
#pragma unroll
  for (int i = 1; i < LFA_FIN_BATCH_MAX; ++i) j += (i < a.njobs && blockIdx.x >= a.start[i]) ? 1 : 0;
  lfa_enc_bwd_finalize_body(a.G[j], a.mom[j], a.E[j], a.w[j], a.b[j], a.gamma[j], a.mean[j], a.invstd[j], a.dw[j], a.db[j],
                            a.dgamma[j], a.dbeta[j], a.D[j], a.acc, (int)(blockIdx.x - a.start[j]));
}

extern "C" int m3d_lfa_enc_bwd_finalize_batch(int32_t njobs, const double* const* G, const double* const* mom65,
                                              const int64_t* num_edges, const float* const* w, const float* const* b,
                                              const float* const* gamma, const float* const* mean,
                                              const float* const* invstd, float* const* dw, float* const* db,
                                              float* const* dgamma, float* const* dbeta, const int32_t* D,
                                              int32_t accumulate, void* stream) {
  if (njobs < 0) return M3D_ERR_INVALID;
  if (njobs == 0) return M3D_OK;
  if (!G || !mom65 || !num_edges || !w || !b || !gamma || !mean || !invstd || !dw || !db || !dgamma || !dbeta || !D)
    return M3D_ERR_INVALID;
  for (int j0 = 0; j0 < njobs; j0 += LFA_FIN_BATCH_MAX) {
    LfaFinBatch a;
    const int m = njobs - j0 < LFA_FIN_BATCH_MAX ? njobs - j0 : LFA_FIN_BATCH_MAX;
    a.njobs = m; a.acc = accumulate;
    unsigned total = 0;
    for (int i = 0; i < LFA_FIN_BATCH_MAX; ++i) {
      a.start[i] = total;
      const int j = j0 + (i < m ? i : 0);
      if (i < m && (D[j] < 0 || num_edges[j] < 1 || !G[j] || !mom65[j] || !w[j] || !b[j] || !gamma[j] || !mean[j] ||
                    !invstd[j] || !dw[j] || !db[j] || !dgamma[j] || !dbeta[j]))
        return M3D_ERR_INVALID;
      a.G[i] = G[j]; a.mom[i] = mom65[j]; a.E[i] = (double)num_edges[j]; a.w[i] = w[j]; a.b[i] = b[j]; a.gamma[i] = gamma[j];
      a.mean[i] = mean[j]; a.invstd[i] = invstd[j]; a.dw[i] = dw[j]; a.db[i] = db[j]; a.dgamma[i] = dgamma[j];
      a.dbeta[i] = dbeta[j]; a.D[i] = i < m ? D[j] : 0;
      if (i < m) total += (unsigned)D[j];
    }
    a.start[LFA_FIN_BATCH_MAX] = total;
    if (total) hipLaunchKernelGGL(lfa_enc_bwd_finalize_batch_kernel, dim3(total), dim3(64), 0, (hipStream_t)stream, a);
  }
  M3D_CHECK_LAUNCH();
  return M3D_OK;
}

extern "C" int m3d_lfa_enc_bwd_finalize(const double* G, const double* mom65, int64_t num_edges, const float* w,
                                        const float* b, const float* gamma, const float* mean, const float* invstd,
                                        float* dw, float* db, float* dgamma, float* dbeta, int32_t D,
                                        int32_t accumulate, void* stream) {
  if (D < 0) return M3D_ERR_INVALID;
  if (D == 0) return M3D_OK;
  if (!G || !mom65 || !w || !b || !gamma || !mean || !invstd || !dw || !db || !dgamma || !dbeta || num_edges < 1)
    return M3D_ERR_INVALID;
  hipLaunchKernelGGL(lfa_enc_bwd_finalize_kernel, dim3(D), dim3(64), 0, (hipStream_t)stream, G, mom65,
                     (double)num_edges, w, b, gamma, mean, invstd, dw, db, dgamma, dbeta, D, accumulate);
  M3D_CHECK_LAUNCH();
  return M3D_OK;
}
