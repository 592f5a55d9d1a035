// Fused backward of Local Spatial Encoding + attentive pooling (gfx950).
//
// Autograd transpose of LocalFeatureAggregation.propagate/message
// (/root/reference/myria3d/models/modules/pyg_randla_net.py:112-152), which upstream runs as ~30 separate
// gather / scatter / GEMM / softmax backward kernels over [E,.] tensors kept alive from the forward pass.
//
// Here nothing of size [E,.] exists in HBM in either direction: the workgroup recomputes its F tile and attention
// logits exactly like lfa_fwd_kernel, then (all GEMMs on v_mfma_f32_16x16x4_f32, tiles in LDS)
//   3'  softmax over each centre's neighbours in the MFMA C layout,  s = softmax(A),  out = sum s*F,
//         dA = s * dout * (F - out)  -> LDS tile DA;   accumulator <- dout * s   (direct path of d(s*F)/dF)
//   4   dF  = dout*s + DA * W_att            (A operand: DA tile, B: W_att^T pre-packed in fragment order)
//   5   dW_att += DA^T * F                   (per-workgroup partial kept in registers across its whole loop)
//   6   dx[j_e, :] += dF[e, :D]              (fp32 atomics straight from the accumulator: 64-byte row segments)
//       dy = dF[e, D:] * LeakyReLU'(lse)     -> LDS
//   7   G += dy^T * [r | 1]                  (11 numbers per encoder channel: everything the encoder's Linear +
//                                             train-mode BatchNorm backward needs, see m3d_lfa_enc_bwd_finalize)
// Workgroup partials of dW_att and G are written with plain stores and summed by a second tiny kernel
// (deterministic; same-address atomics cost ~15 ns each on MI355X and would dominate).
#include <stdlib.h>
#include "m3d_common.h"
#include "lfa_common.h"
#include "../../include/m3d_hip.h"

struct LfaBwdArgs {
  const float* x; const float4* pos4; const int32_t* idx;
  const float* wf; const float* bf;
  const float4* wp;   // packed W_att    (GEMM-1 B fragments)
  const float4* wpt;  // packed W_att^T  (GEMM-2 B fragments)
  const float* dout;  // [n, CH]
  float* dx;          // [n, D], atomically accumulated
  float* dxe;         // (flags bit 5) [n * K, D]: the x-part of dF per EDGE, plainly stored; the caller sums it per neighbour
  const int32_t* eslot;  // (with dxe; may be null) row of dxe that edge e is stored in — its position in the reverse list of the
                         // point it names (m3d_knn_reverse), so that a point's contributions are CONTIGUOUS rows; null: row e
  float* dw_part;     // [parts][CHP*CHP]
  float* g_part;      // [parts][GP*16],  GP = max(16, D)
  int64_t n;
  int K, CH, D;
  float slope;
};

// phase-ablation builds only (tools/build_variant.sh bwd_dbgN lfa_bwd.hip -DLFA_BWD_DBG=N; profiles/r02d_*): bit 0 skips
// the dx atomics, bits 1..5 stop a group after phase 1 / 2 / 3 / 5 / 6.  0 in the product: the branches fold away.
#ifndef LFA_BWD_DBG
#define LFA_BWD_DBG 0
#endif
// A/B builds only: 1 = ignore the FULL promise (flags bit 3) and launch the general kernels
// A/B builds only: 0 = the complete-neighbourhood kernels keep the cross-lane softmax (no in-lane row permutation)
#ifndef LFA_BWD_INL
#define LFA_BWD_INL 1
#endif
#ifndef LFA_BWD_INL_MINCH
#define LFA_BWD_INL_MINCH 128
#endif
#ifndef LFA_BWD_DBG_NOFULL
#define LFA_BWD_DBG_NOFULL 0
#endif
// tile geometry of the backward kernel per padded channel count: edge rows per workgroup iteration, waves per
// workgroup, cap on resident (persistent) workgroups.  Overridable at compile time for tuning sweeps.
// 1: double-buffer the weight fragments of GEMM-1 / GEMM-2 in the non-pipelined kernel (ch >= 128): untuned A/B knob
#ifndef BWD_B_PREFETCH
#define BWD_B_PREFETCH 0
#endif
#ifndef BWD_PIPE_8
#define BWD_PIPE_8 1
#endif
#ifndef BWD_PIPE_16
#define BWD_PIPE_16 1
#endif
#ifndef BWD_PIPE_32
#define BWD_PIPE_32 1
#endif
#ifndef BWD_PIPE_64
#define BWD_PIPE_64 1
#endif
#ifndef BWD_ROWS_16
#define BWD_ROWS_16 128  // with MINW 4: four 4-wave workgroups per CU instead of two (ch=16: 334 -> 244 us, ch=8: 290 -> 240)
#endif
#ifndef BWD_ROWS_32
#define BWD_ROWS_32 64   // with MINW 4: four workgroups per CU (154 -> 121 us)
#endif
#ifndef BWD_ROWS_64
#define BWD_ROWS_64 64   // 4-wave workgroups, four per CU (286 -> 270 us at level 2)
#endif
#ifndef BWD_ROWS_128
#define BWD_ROWS_128 64
#endif
#ifndef BWD_ROWS_256
#define BWD_ROWS_256 64
#endif
#ifndef BWD_NW_16
#define BWD_NW_16 4
#endif
#ifndef BWD_NW_32
#define BWD_NW_32 4
#endif
#ifndef BWD_NW_64
#define BWD_NW_64 4
#endif
#ifndef BWD_NW_128
#define BWD_NW_128 8   // with MINW 4: two 8-wave workgroups per CU (277 -> 255 us at level 3)
#endif
#ifndef BWD_NW_256
#define BWD_NW_256 8
#endif
#ifndef BWD_CAP_16
#define BWD_CAP_16 1024
#endif
#ifndef BWD_CAP_32
#define BWD_CAP_32 1024
#endif
#ifndef BWD_CAP_64
#define BWD_CAP_64 1024
#endif
#ifndef BWD_CAP_128
#define BWD_CAP_128 512
#endif
#ifndef BWD_CAP_256
#define BWD_CAP_256 256
#endif
// MINW: wavefronts per SIMD the register allocator must leave room for (2nd argument of __launch_bounds__ in HIP)
#ifndef BWD_MINW_16
#define BWD_MINW_16 4
#endif
#ifndef BWD_MINW_32
#define BWD_MINW_32 4
#endif
#ifndef BWD_MINW_64
#define BWD_MINW_64 4  // 128 VGPRs: 385 -> 293 us at level 2 with the former 8-wave workgroups
#endif
#ifndef BWD_MINW_128
#define BWD_MINW_128 4
#endif
#ifndef BWD_MINW_256
#define BWD_MINW_256 1
#endif
// wave priority by phase (s_setprio; the workgroups that share a CU sit in different phases, and fp32 MFMA runs at the
// vector rate on the SIMD the VALU phases need): a mask of the phases that run at priority LFA_BWD_PRIO_LVL, the others
// at 0.  bit 0 = gather / encoder (phase 1), 1 = logits GEMM (2), 2 = softmax / dA (3'), 3 = dF and dW GEMMs (4, 5),
// 4 = scatter / dy / encoder sums (6, 7).
#ifndef LFA_BWD_SETPRIO
#define LFA_BWD_SETPRIO 17  // gather / encoder and scatter phases first: -2.5 ... -5 % per launch at levels 1-2, -0.03 ms per step (profiles/r04h_*, r04i_*)
#endif
#ifndef LFA_BWD_PRIO_LVL
#define LFA_BWD_PRIO_LVL 1
#endif
#define BWD_PRIO(bit) do { if (LFA_BWD_SETPRIO) __builtin_amdgcn_s_setprio((LFA_BWD_SETPRIO & (bit)) ? LFA_BWD_PRIO_LVL : 0); } while (0)
template <int CHP> struct BwdCfg {};
template <> struct BwdCfg<16> { static constexpr int ROWS = BWD_ROWS_16, NW = BWD_NW_16, CAP = BWD_CAP_16, MINW = BWD_MINW_16; };
template <> struct BwdCfg<32> { static constexpr int ROWS = BWD_ROWS_32, NW = BWD_NW_32, CAP = BWD_CAP_32, MINW = BWD_MINW_32; };
template <> struct BwdCfg<64> { static constexpr int ROWS = BWD_ROWS_64, NW = BWD_NW_64, CAP = BWD_CAP_64, MINW = BWD_MINW_64; };
template <> struct BwdCfg<128> { static constexpr int ROWS = BWD_ROWS_128, NW = BWD_NW_128, CAP = BWD_CAP_128, MINW = BWD_MINW_128; };
template <> struct BwdCfg<256> { static constexpr int ROWS = BWD_ROWS_256, NW = BWD_NW_256, CAP = BWD_CAP_256, MINW = BWD_MINW_256; };

// PIPE (ch <= 64): software-pipelined load schedule — neighbour ids, x_j rows, positions and dout of the NEXT group
// are loaded into registers while the current group's GEMMs run, and all B fragments of a GEMM are fetched up front.
// Identical arithmetic either way; the launcher picks per channel count (BWD_PIPE_*, M3D_LFA_BWD_PIPE).
// BF (ch >= 64): the three attention GEMMs (recomputed logits, dF, dW_att) take bf16 operands on
// v_mfma_f32_16x16x32_bf16 with fp32 accumulation — fragments are rounded when they are built from the fp32 LDS tiles
// (a.wp / a.wpt then hold the bf16 fragments of m3d_lfa_pack_att_bf16); everything else is unchanged fp32.
// FULL (round 5; flags bit 3 of m3d_lfa_bwd): the caller promises complete neighbourhoods (K == KP, every id >= 0) and the
// host has checked that every byte offset fits 32 bits — the validity masks of every phase (zero fills of masked rows, the
// s_and_saveexec branches around each output register, the v_cndmask chains of the maxima), the 64-bit pointer arithmetic
// and the three-instruction maxima / IEEE square root / sub-mul-exp of the general kernel are compiled out: read off the
// ISA (tools/isa_count.py), the kernels are VALU-issue bound beside their MFMAs.  Centres past n (last group) are
// computed from clamped rows with dout = 0: every contribution they make is an exact zero.
// X3 (with BF and FULL): split-bf16 operands — every bf16 product becomes hi*hi + hi*lo + lo*hi (m3d_common.h); a.wp / a.wpt hold
// the hi fragments followed by the lo fragments (m3d_lfa_prepare(bf16 = 2)).
// IOH (flags M3D_IO_BF16): x, dout (and the small kernel's edge rows) hold bf16; the atomically accumulated dx stays fp32
template <int CH, int KP, bool PIPE, bool BF = false, bool FULL = false, bool X3 = false, bool IOH = false>
__global__ __launch_bounds__(BwdCfg<(CH < 16 ? 16 : CH)>::NW * 64, BwdCfg<(CH < 16 ? 16 : CH)>::MINW) void lfa_bwd_kernel(LfaBwdArgs a) {
  constexpr int CHP = CH < 16 ? 16 : CH;
  constexpr int D = CH / 2;  // compile-time channel counts: no integer divisions in the index arithmetic
  constexpr int NW = BwdCfg<CHP>::NW, NTHR = NW * 64;
  constexpr int ROWS = BwdCfg<CHP>::ROWS;
  constexpr int TC = ROWS / KP, KT = KP / 16;
  constexpr int STR = CHP + 2, RSTR = 18;
  constexpr int MT = ROWS / 16, NT = CHP / 16;
  // tile ownership: columns first (a wave owns one or two column tiles and every centre tile of the group).  Round 4 tried rows
  // first — few centre tiles x many column tiles per wave, one LDS read of an A fragment feeding 4-8 MFMAs instead of 1-2, on
  // the theory that the GEMM phases are bound by the LDS pipe — and square dW_att blocks: bit-identical and SLOWER at every
  // deep level (ch = 64 / 128 / 256: 265 -> 292, 247 -> 278, 287 -> 325 us; the forward kernel likewise, 79 -> 89 and 89 -> 107
  // us; profiles/r04f_lfa_tile_ownership_ab.log): the wider B-fragment sets cost more registers and loads than the LDS reads
  // they save.  Removed.
  constexpr int WN = NT < NW ? NT : NW, WM = NW / WN;
  constexpr int NTW = NT / WN, MTW = MT / WM;
  constexpr int S4 = CHP / 16;
  static_assert(MTW % KT == 0, "centre tiles must stay inside one wave");
  // GEMM-3 (dW_att) tile ownership
  constexpr int T3 = NT * NT;
  constexpr int KSPL3 = T3 >= NW ? 1 : NW / T3;
  constexpr int TPW3 = T3 >= NW ? T3 / NW : 1;
  constexpr int KTW3 = TPW3 < NT ? TPW3 : NT;
  constexpr int CTW3 = TPW3 / KTW3;
  // GEMM-4 (G) tile ownership: GT tiles of 16 encoder channels
  constexpr int DP = CHP / 2 < 16 ? 16 : CHP / 2;  // padded encoder width
  constexpr int GT = DP / 16;
  constexpr int KSPL4 = GT >= NW ? 1 : NW / GT;
  static_assert(GT <= NW, "one G tile per wave at most");
  constexpr int D4 = D >> 2;
  constexpr int GPT = (ROWS * D4 + NTHR - 1) / NTHR;  // prefetched x_j segments (float4) per thread
  constexpr int NCW = MTW / KT;                       // centres per wave in the softmax phase
  // IN-LANE NEIGHBOURHOODS (round 5, as in lfa_fwd_full_kernel): a wave owns four row tiles and LDS row
  // 64 w + 16 m + 4 u + r holds neighbour 4 m + r of the wave's centre u, so that lane (lr, lg) of the MFMA C layout has all 16
  // logits of centre lg for its column in its 4 tiles x 4 registers — the softmax backward needs no cross-lane maxima / sums.
  // Everything indexed by LDS row (ids, F, RT, DA, the dx scatter) stays as it is; only "which edge is row rho" changes.
  // (ch = 64 keeps the cross-lane softmax: same-box A/B, profiles/r05n_*: isolated launches 233 vs 237 us in favour of the
  // in-lane layout, but INSIDE training steps 234 vs 223 us against it, three runs out of three; ch = 128 / 256: 227 vs 232 and
  // 270 vs 280 us)
  constexpr bool INL = LFA_BWD_INL && FULL && KP == 16 && MTW == 4 && ROWS == 64 * WM && CH >= LFA_BWD_INL_MINCH;
  auto row_edge = [](int rho) -> int {    // natural edge number (centre * KP + neighbour) of LDS row rho
    return INL ? (((rho >> 6) * 4 + ((rho >> 2) & 3)) * 16 + ((rho >> 4) & 3) * 4 + (rho & 3)) : rho;
  };
  auto row_centre = [](int rho) -> int {  // group-local centre of LDS row rho
    return INL ? ((rho >> 6) * 4 + ((rho >> 2) & 3)) : rho / KP;
  };
  static_assert(!PIPE || ROWS <= NTHR, "one neighbour id per thread");
  static_assert(!PIPE || S4 * NTW <= 8, "B fragments of one GEMM are held in registers");

  __shared__ float F[ROWS * STR];
  __shared__ float DA[ROWS * STR];
  __shared__ float RT[ROWS * RSTR];
  __shared__ int nbr2[PIPE ? 2 : 1][ROWS];  // neighbour ids of the current (and, pipelined, of the next) group

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  const int K = a.K;
  const int wn = wid % WN, wm = wid / WN;

  // persistent accumulators
  f32x4 acc3[CTW3][KTW3];
#pragma unroll
  for (int c = 0; c < CTW3; ++c)
#pragma unroll
    for (int k = 0; k < KTW3; ++k) acc3[c][k] = (f32x4){0.f, 0.f, 0.f, 0.f};
  f32x4 accg = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int t0 = (wid / KSPL3) * TPW3;
  const int ct0 = t0 / NT, kt0 = t0 % NT, ks3 = wid % KSPL3;
  const int gt = wid / KSPL4, ks4 = wid % KSPL4;

  const int64_t ngroups = (a.n + TC - 1) / TC;
  const int64_t nlast = a.n - 1;
  // this workgroup's groups: g0, g0 + gs, ... < gend — the contiguous eighth of its XCD (m3d_common.h: the dx atomics of a
  // tile then all come from ONE XCD and stay in its L2; round 3's order, group b, b + gridDim.x, ..., sent every XCD to every
  // tile and every atomic to the memory side: 97 / 113 MB written for 3.3 / 6.6 MB of dx at level 1)
  int64_t g0, gs, gend;
  xcd_range(blockIdx.x, gridDim.x, ngroups, g0, gs, gend);
  // ---- software pipeline over the persistent loop: everything a group needs from HBM (neighbour ids, x_j rows,
  // positions, dout) is loaded one group ahead into registers, so the loads fly during the MFMA phases of the
  // previous group.  All prefetch loads are unconditional (clamped addresses); validity is applied on use.
  float4 xg[GPT], ppi, ppj;
  float dgp[NCW][NTW], dgc[NCW][NTW];
  int jn = -1;
  const unsigned n32 = (unsigned)a.n, elast = n32 * KP - 1u;
  auto load_idx = [&](int64_t g) -> int {
    if constexpr (FULL) {
      // (groups past the end re-read the last edge: an unconditional load, its row never used)
      unsigned eo = (unsigned)g * ROWS + (unsigned)row_edge(tid < ROWS ? tid : 0);
      eo = eo < elast ? eo : elast;
      return a.idx[eo];
    }
    int j = -1;
    if (tid < ROWS && g < gend) {
      const int ci = tid / KP, k = tid % KP;
      const int64_t i = g * TC + ci;
      if (i < a.n && k < K) j = a.idx[i * K + k];
    }
    return j;
  };
  auto prefetch = [&](int64_t g, const int* nb) {
    const int64_t c0 = g * TC;
    if constexpr (FULL) {
      const unsigned c32 = (unsigned)c0;
#pragma unroll
      for (int u = 0; u < GPT; ++u) {
        const int f = tid + u * NTHR;
        const int e = (f / D4) % ROWS, c4 = f % D4;
        xg[u] = io_load4_b<IOH>(a.x, (unsigned)nb[e] * (unsigned)(D * 4) + (unsigned)(c4 * 16));
      }
      {
        const int e = tid % ROWS;
        unsigned i = c32 + (unsigned)row_centre(e);
        i = i < n32 ? i : n32 - 1u;
        ppi = *(const float4*)((const char*)a.pos4 + i * 16u);
        ppj = *(const float4*)((const char*)a.pos4 + (unsigned)nb[e] * 16u);
      }
#pragma unroll
      for (int cc = 0; cc < (INL ? 1 : NCW); ++cc) {
        // (INL: ONE centre per lane — centre 4 wm + lg of the group — instead of the wave's NCW centres in every lane)
        unsigned i = c32 + (unsigned)(INL ? wm * 4 + lg : (wm * MTW + cc * KT) / KT);
        i = i < n32 ? i : n32 - 1u;
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
          const int col = (wn * NTW + t) * 16 + lr;
          dgp[cc][t] = io_load1_b<IOH>(a.dout, i * (unsigned)(CH * 4) + (unsigned)((col < CH ? col : CH - 1) * 4));
        }
      }
      return;
    }
#pragma unroll
    for (int u = 0; u < GPT; ++u) {
      const int f = tid + u * NTHR;
      const int e = (f / D4) % ROWS, c4 = f % D4;  // (f >= ROWS*D4 only in a padded last trip: harmless extra load)
      const int j = nb[e];
      xg[u] = io_load4<IOH>(a.x, (size_t)((int64_t)(j < 0 ? 0 : j) * D + c4 * 4));
    }
    {
      const int e = tid % ROWS;
      const int j = nb[e];
      const int64_t i = c0 + e / KP;
      ppi = a.pos4[i < a.n ? i : nlast];
      ppj = a.pos4[j < 0 ? 0 : j];
    }
#pragma unroll
    for (int cc = 0; cc < NCW; ++cc) {
      const int64_t i = c0 + (wm * MTW + cc * KT) / KT;
#pragma unroll
      for (int t = 0; t < NTW; ++t) {
        const int col = (wn * NTW + t) * 16 + lr;
        dgp[cc][t] = io_load1<IOH>(a.dout, (size_t)((i < a.n ? i : nlast) * CH + (col < CH ? col : CH - 1)));
      }
    }
  };
  int cur = 0;
  if (PIPE && g0 < gend) {
    const int j0 = load_idx(g0);
    if (tid < ROWS) nbr2[0][tid] = j0;
    __syncthreads();
    prefetch(g0, nbr2[0]);
    jn = load_idx(g0 + gs);
  }
  for (int64_t grp = g0; grp < gend; grp += gs, cur ^= (PIPE ? 1 : 0)) {
    const int64_t c0 = grp * TC;
    int* nbr = nbr2[cur];
    BWD_PRIO(1);
    if constexpr (PIPE) {
      // ---- phase 1 (from the prefetched registers): x_j -> F[:, 0:D]
      {
#pragma unroll
        for (int u = 0; u < GPT; ++u) {
          const int f = tid + u * NTHR;
          if (f < ROWS * D4) {
            const int e = f / D4, c4 = f % D4;
            float4 v = xg[u];
            if constexpr (!FULL) {
              if (nbr[e] < 0) v = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            float* d = &F[e * STR + c4 * 4];
            *(float2*)d = make_float2(v.x, v.y);
            *(float2*)(d + 2) = make_float2(v.z, v.w);
          }
        }
        if (CH < CHP) {
          constexpr int P = CHP - CH;
          for (int f = tid; f < ROWS * P; f += NTHR) F[(f / P) * STR + CH + (f % P)] = 0.f;
        }
      }
      // ---- phase 1c: r, folded encoder -> F[:, D:2D];  [r | 1 | 0...] -> RT
      {
        constexpr int NG = NTHR / ROWS;
        const int e = tid % ROWS;
        const int grp_c = __builtin_amdgcn_readfirstlane(tid / ROWS);
        constexpr int DG = D / NG;
        const int j = FULL ? 0 : nbr[e];
        float r[10];
        if constexpr (FULL) rel_pos_fast(ppi, ppj, r);
        else rel_pos(ppi, ppj, r);
        if (j < 0) {
#pragma unroll
          for (int q = 0; q < 10; ++q) r[q] = 0.f;
        }
        if (grp_c == 0) {
          float* rt = &RT[e * RSTR];
#pragma unroll
          for (int q = 0; q < 10; ++q) rt[q] = r[q];
          rt[10] = j >= 0 ? 1.f : 0.f;
#pragma unroll
          for (int q = 11; q < 16; ++q) rt[q] = 0.f;
        }
        for (int c = grp_c * DG; c < (grp_c + 1) * DG; ++c) {
          const float* w = a.wf + c * 10;
          float v = a.bf[c];
#pragma unroll
          for (int q = 0; q < 10; ++q) v += w[q] * r[q];
          if constexpr (FULL) F[e * STR + D + c] = fmaxf(v, v * a.slope);  // (0 <= slope <= 1: checked by the host)
          else F[e * STR + D + c] = j >= 0 ? lrelu(v, a.slope) : 0.f;
        }
      }
#pragma unroll
      for (int cc = 0; cc < NCW; ++cc)
#pragma unroll
        for (int t = 0; t < NTW; ++t) dgc[cc][t] = dgp[cc][t];
      if (tid < ROWS) nbr2[cur ^ 1][tid] = jn;  // ids of the next group (all -1 past the end)
      __syncthreads();
    } else {
      // ---- phase 1a: neighbour ids
      for (int e = tid; e < ROWS; e += NTHR) {
        if constexpr (FULL) {
          unsigned eo = (unsigned)c0 * KP + (unsigned)row_edge(e);
          nbr[e] = a.idx[eo < elast ? eo : elast];
        } else {
          int ci = e / KP, k = e % KP;
          int64_t i = c0 + ci;
          int j = -1;
          if (i < a.n && k < K) j = a.idx[i * K + k];
          nbr[e] = j;
        }
      }
      __syncthreads();
      // ---- phase 1b: gather x_j
      {
        constexpr int D4 = D >> 2;
        for (int f = tid; f < ROWS * D4; f += NTHR) {
          int e = f / D4, c4 = f % D4;
          int j = nbr[e];
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          if constexpr (FULL) v = io_load4_b<IOH>(a.x, (unsigned)j * (unsigned)(D * 4) + (unsigned)(c4 * 16));
          else if (j >= 0) v = io_load4<IOH>(a.x, (size_t)((int64_t)j * D + c4 * 4));
          float* d = &F[e * STR + c4 * 4];
          *(float2*)d = make_float2(v.x, v.y);
          *(float2*)(d + 2) = make_float2(v.z, v.w);
        }
        if (CH < CHP) {
          constexpr int P = CHP - CH;
          for (int f = tid; f < ROWS * P; f += NTHR) F[(f / P) * STR + CH + (f % P)] = 0.f;
        }
      }
      // ---- phase 1c: r, folded encoder -> F[:, D:2D];  [r | 1 | 0...] -> RT
      {
        constexpr int NG = NTHR / ROWS;
        const int e = tid % ROWS;
        const int grp_c = __builtin_amdgcn_readfirstlane(tid / ROWS);
        constexpr int DG = D / NG;
        const int j = nbr[e];
        const int64_t i = c0 + e / KP;
        float r[10];
#pragma unroll
        for (int q = 0; q < 10; ++q) r[q] = 0.f;
        if constexpr (FULL) {
          unsigned i32 = (unsigned)c0 + (unsigned)row_centre(e);
          i32 = i32 < n32 ? i32 : n32 - 1u;
          rel_pos_fast(*(const float4*)((const char*)a.pos4 + i32 * 16u), *(const float4*)((const char*)a.pos4 + (unsigned)j * 16u), r);
        } else if (j >= 0) rel_pos(a.pos4[i], a.pos4[j], r);
        if (grp_c == 0) {
          float* rt = &RT[e * RSTR];
#pragma unroll
          for (int q = 0; q < 10; ++q) rt[q] = r[q];
          rt[10] = j >= 0 ? 1.f : 0.f;
#pragma unroll
          for (int q = 11; q < 16; ++q) rt[q] = 0.f;
        }
        for (int c = grp_c * DG; c < (grp_c + 1) * DG; ++c) {
          const float* w = a.wf + c * 10;
          float v = a.bf[c];
#pragma unroll
          for (int q = 0; q < 10; ++q) v += w[q] * r[q];
          if constexpr (FULL) F[e * STR + D + c] = fmaxf(v, v * a.slope);
          else F[e * STR + D + c] = j >= 0 ? lrelu(v, a.slope) : 0.f;
        }
      }
      __syncthreads();
    }

    if (LFA_BWD_DBG & 2) continue;   // timing experiment: phase 1 only
    // ---- phase 2: A = F * W_att^T
    BWD_PRIO(2);
    f32x4 acc[MTW][NTW];
#pragma unroll
    for (int m = 0; m < MTW; ++m)
#pragma unroll
      for (int t = 0; t < NTW; ++t) acc[m][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if constexpr (BF) {
      constexpr int KS = CHP / 32;
      const uint4* wpb = (const uint4*)a.wp;
      const uint4* wpl = wpb + (size_t)CH * CH / 8;
      const float* fa = &F[((wm * MTW) * 16 + lr) * STR + lg * 8];
#pragma unroll(KS <= 2 ? KS : 1)
      for (int ks = 0; ks < KS; ++ks) {
        Bf16Frag b[NTW], bl[NTW];
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
          b[t].q = wpb[((size_t)(wn * NTW + t) * KS + ks) * 64 + lane];
          if constexpr (X3) bl[t].q = wpl[((size_t)(wn * NTW + t) * KS + ks) * 64 + lane];
        }
#pragma unroll
        for (int m = 0; m < MTW; ++m) {
          if constexpr (X3) {
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
    } else     if constexpr (PIPE) {
      {
        const float* fa = &F[((wm * MTW) * 16 + lr) * STR + lg];
        float4 b[S4][NTW];  // every B fragment of this wave's column tiles: one latency exposure, not one per k-step
#pragma unroll
        for (int s4 = 0; s4 < S4; ++s4)
#pragma unroll
          for (int t = 0; t < NTW; ++t) b[s4][t] = a.wp[((size_t)(wn * NTW + t) * S4 + s4) * 64 + lane];
#pragma unroll
        for (int s4 = 0; s4 < S4; ++s4) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float av[MTW];
#pragma unroll
            for (int m = 0; m < MTW; ++m) av[m] = fa[m * 16 * STR + (s4 * 4 + i) * 4];
#pragma unroll
            for (int t = 0; t < NTW; ++t) {
              const float bv = i == 0 ? b[s4][t].x : (i == 1 ? b[s4][t].y : (i == 2 ? b[s4][t].z : b[s4][t].w));
#pragma unroll
              for (int m = 0; m < MTW; ++m) acc[m][t] = mfma16(av[m], bv, acc[m][t]);
            }
          }
        }
      }
    } else {
      {
        const float* fa = &F[((wm * MTW) * 16 + lr) * STR + lg];
#if BWD_B_PREFETCH
        float4 bn[NTW];  // B fragments double-buffered: group s4+1 is in flight during the MFMAs of group s4
#pragma unroll
        for (int t = 0; t < NTW; ++t) bn[t] = a.wp[((size_t)(wn * NTW + t) * S4) * 64 + lane];
#endif
#pragma unroll 1
        for (int s4 = 0; s4 < S4; ++s4) {
          float4 b[NTW];
#if BWD_B_PREFETCH
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
    }
    // B fragments of GEMM-2 (W_att^T): issued now, they arrive during the softmax phase
    float4 b4[(PIPE && !BF) ? S4 : 1][NTW];
    if constexpr (PIPE && !BF) {
#pragma unroll
      for (int s4 = 0; s4 < S4; ++s4)
#pragma unroll
        for (int t = 0; t < NTW; ++t) b4[s4][t] = a.wpt[((size_t)(wn * NTW + t) * S4 + s4) * 64 + lane];
    }

    BWD_PRIO(4);
    if (LFA_BWD_DBG & 4) continue;   // timing experiment: phases 1-2
    // ---- phase 3': softmax, dA -> LDS, acc <- dout * s
    const float pinf = FULL ? fast_pinf() : opaque_pinf();
    if constexpr (INL) {
      // lane (lr, lg): centre 4 wm + lg of the group, column 16 (wn NTW + t) + lr, neighbour 4 m + r in acc[m][t][r]
      const int64_t ic = c0 + wm * 4 + lg;
      const int rbase = (wm * 64 + lg * 4) * STR;
#pragma unroll
      for (int t = 0; t < NTW; ++t) {
        const int col = (wn * NTW + t) * 16 + lr;
        float mx = acc[0][t][0];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (m + r > 0) mx = fast_max(mx, acc[m][t][r], pinf);
        const float ml = mx * 1.4426950408889634f;
        float num = 0.f, den = 0.f;
        float fv[4][4];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(acc[m][t][r], 1.4426950408889634f, -ml));
            const float f = F[rbase + (m * 16 + r) * STR + col];
            num = __builtin_fmaf(p, f, num);
            den += p;
            acc[m][t][r] = p;
            fv[m][r] = f;
          }
        const float inv = __builtin_amdgcn_rcpf(den + 1e-16f);
        const float o = num * inv;
        float g = 0.f;
        if (ic < a.n && col < CH) {
          if constexpr (PIPE) g = dgc[0][t];
          else g = io_load1_b<IOH>(a.dout, (unsigned)ic * (unsigned)(CH * 4) + (unsigned)(col * 4));
        }
        const float gi = g * inv;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float gs = acc[m][t][r] * gi;  // dout * softmax weight
            DA[rbase + (m * 16 + r) * STR + col] = gs * (fv[m][r] - o);
            acc[m][t][r] = gs;
          }
      }
    } else
#pragma unroll
    for (int cc = 0; cc < MTW / KT; ++cc) {
      const int mt0 = wm * MTW + cc * KT;
      const int64_t i = c0 + mt0 / KT;
      bool vr[KT][4];
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) vr[kt][r] = FULL ? true : nbr[(mt0 + kt) * 16 + lg * 4 + r] >= 0;
#pragma unroll
      for (int t = 0; t < NTW; ++t) {
        const int col = (wn * NTW + t) * 16 + lr;
        float mx = -__builtin_inff();
        if constexpr (FULL) {
          mx = acc[cc * KT][t][0];
#pragma unroll
          for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (kt + r > 0) mx = fast_max(mx, acc[cc * KT + kt][t][r], pinf);
          float p_, q_;
          xgroup_pair16(mx, p_, q_); mx = fast_max(p_, q_, pinf);
          xgroup_pair32(mx, p_, q_); mx = fast_max(p_, q_, pinf);
        } else {
#pragma unroll
          for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (vr[kt][r]) mx = max_f(mx, acc[cc * KT + kt][t][r], pinf);
          mx = xgroup_max(mx, pinf);
        }
        float num = 0.f, den = 0.f;
        float fv[KT][4];
        const float ml = mx * 1.4426950408889634f;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float p = 0.f, f = 0.f;
            if constexpr (FULL) {
              p = __builtin_amdgcn_exp2f(__builtin_fmaf(acc[cc * KT + kt][t][r], 1.4426950408889634f, -ml));
              f = F[((mt0 + kt) * 16 + lg * 4 + r) * STR + col];
              num = __builtin_fmaf(p, f, num);
            } else {
              if (vr[kt][r]) {
                p = __expf(acc[cc * KT + kt][t][r] - mx);
                f = F[((mt0 + kt) * 16 + lg * 4 + r) * STR + col];
              }
              num += p * f;
            }
            den += p;
            acc[cc * KT + kt][t][r] = p;
            fv[kt][r] = f;
          }
        num = xgroup_sum(num);
        den = xgroup_sum(den);
        const float inv = __builtin_amdgcn_rcpf(den + 1e-16f);  // (1 ulp; IEEE division costs ~10 instructions)
        const float o = num * inv;
        float g = 0.f;
        if (i < a.n && col < CH) {
          if constexpr (PIPE) g = dgc[cc][t];
          else if constexpr (FULL) g = io_load1_b<IOH>(a.dout, (unsigned)i * (unsigned)(CH * 4) + (unsigned)(col * 4));
          else g = io_load1<IOH>(a.dout, (size_t)(i * CH + col));
        }
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float s = acc[cc * KT + kt][t][r] * inv;
            DA[((mt0 + kt) * 16 + lg * 4 + r) * STR + col] = s * g * (fv[kt][r] - o);
            acc[cc * KT + kt][t][r] = g * s;
          }
      }
    }
    __syncthreads();

    if (LFA_BWD_DBG & 8) continue;   // timing experiment: phases 1-3
    // ---- phase 4: dF = dout*s + DA * W_att
    BWD_PRIO(8);
    if constexpr (BF) {
      constexpr int KS = CHP / 32;
      const uint4* wptb = (const uint4*)a.wpt;
      const uint4* wptl = wptb + (size_t)CH * CH / 8;
      const float* da = &DA[((wm * MTW) * 16 + lr) * STR + lg * 8];
#pragma unroll(KS <= 2 ? KS : 1)
      for (int ks = 0; ks < KS; ++ks) {
        Bf16Frag b[NTW], bl[NTW];
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
          b[t].q = wptb[((size_t)(wn * NTW + t) * KS + ks) * 64 + lane];
          if constexpr (X3) bl[t].q = wptl[((size_t)(wn * NTW + t) * KS + ks) * 64 + lane];
        }
#pragma unroll
        for (int m = 0; m < MTW; ++m) {
          if constexpr (X3) {
            const Bf16Split av = lds_row_to_bf16_split(da + m * 16 * STR + ks * 32);
#pragma unroll
            for (int t = 0; t < NTW; ++t) {
              acc[m][t] = mfma_bf16(av.lo, b[t].v, acc[m][t]);
              acc[m][t] = mfma_bf16(av.hi, bl[t].v, acc[m][t]);
              acc[m][t] = mfma_bf16(av.hi, b[t].v, acc[m][t]);
            }
          } else {
            const bf16x8 av = lds_row_to_bf16(da + m * 16 * STR + ks * 32);
#pragma unroll
            for (int t = 0; t < NTW; ++t) acc[m][t] = mfma_bf16(av, b[t].v, acc[m][t]);
          }
        }
        if constexpr (PIPE) {
          if (ks == 0 && grp + gs < gend) {  // next group's loads (see the fp32 branch below)
            prefetch(grp + gs, nbr2[cur ^ 1]);
            jn = load_idx(grp + 2 * gs);
          }
        }
      }
    } else if constexpr (PIPE) {
      {
        // next group's loads go out here: the last in-iteration global load has been consumed before the first MFMA
        // below, so waiting for it (vmcnt is in order) no longer drags these along
        const float* da = &DA[((wm * MTW) * 16 + lr) * STR + lg];
        bool first = true;
#pragma unroll
        for (int s4 = 0; s4 < S4; ++s4) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float av[MTW];
#pragma unroll
            for (int m = 0; m < MTW; ++m) av[m] = da[m * 16 * STR + (s4 * 4 + i) * 4];
#pragma unroll
            for (int t = 0; t < NTW; ++t) {
              const float bv = i == 0 ? b4[s4][t].x : (i == 1 ? b4[s4][t].y : (i == 2 ? b4[s4][t].z : b4[s4][t].w));
#pragma unroll
              for (int m = 0; m < MTW; ++m) acc[m][t] = mfma16(av[m], bv, acc[m][t]);
            }
          }
          if (first) {
            first = false;
            if (grp + gs < gend) {
              prefetch(grp + gs, nbr2[cur ^ 1]);
              jn = load_idx(grp + 2 * gs);
            }
          }
        }
      }
    } else {
      {
        const float* da = &DA[((wm * MTW) * 16 + lr) * STR + lg];
#if BWD_B_PREFETCH
        float4 bn[NTW];  // B fragments double-buffered: group s4+1 is in flight during the MFMAs of group s4
#pragma unroll
        for (int t = 0; t < NTW; ++t) bn[t] = a.wpt[((size_t)(wn * NTW + t) * S4) * 64 + lane];
#endif
#pragma unroll 1
        for (int s4 = 0; s4 < S4; ++s4) {
          float4 b[NTW];
#if BWD_B_PREFETCH
#pragma unroll
          for (int t = 0; t < NTW; ++t) b[t] = bn[t];
          {
            const int sn = s4 + 1 < S4 ? s4 + 1 : s4;  // last trip: a harmless re-load
#pragma unroll
            for (int t = 0; t < NTW; ++t) bn[t] = a.wpt[((size_t)(wn * NTW + t) * S4 + sn) * 64 + lane];
          }
#else
#pragma unroll
          for (int t = 0; t < NTW; ++t) b[t] = a.wpt[((size_t)(wn * NTW + t) * S4 + s4) * 64 + lane];
#endif
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float av[MTW];
#pragma unroll
            for (int m = 0; m < MTW; ++m) av[m] = da[m * 16 * STR + (s4 * 4 + i) * 4];
#pragma unroll
            for (int t = 0; t < NTW; ++t) {
              const float bv = i == 0 ? b[t].x : (i == 1 ? b[t].y : (i == 2 ? b[t].z : b[t].w));
#pragma unroll
              for (int m = 0; m < MTW; ++m) acc[m][t] = mfma16(av[m], bv, acc[m][t]);
            }
          }
        }
      }
    }
    // ---- phase 5: dW_att[c, k] += sum_e DA[e, c] * F[e, k]
    if constexpr (BF) {
      static_assert(!BF || KSPL3 == 1, "bf16 variant: one wave owns whole (c, k) tiles");
#pragma unroll
      for (int sr = 0; sr < ROWS / 32; ++sr) {
        const int eo = (32 * sr + 8 * lg) * STR + lr;
        constexpr int KB = KTW3 < 4 ? KTW3 : 4;  // F fragments in flight (a [16, 8] bf16 fragment is 4 VGPRs)
        if constexpr (X3) {
          constexpr int KB3 = KTW3 < 2 ? KTW3 : 2;
          Bf16Split av[CTW3];
#pragma unroll
          for (int c = 0; c < CTW3; ++c) av[c] = lds_col_to_bf16_split(&DA[eo + (ct0 + c) * 16], STR);
#pragma unroll
          for (int k0 = 0; k0 < KTW3; k0 += KB3) {
            Bf16Split bv[KB3];
#pragma unroll
            for (int k = 0; k < KB3; ++k) bv[k] = lds_col_to_bf16_split(&F[eo + (kt0 + k0 + k) * 16], STR);
#pragma unroll
            for (int c = 0; c < CTW3; ++c)
#pragma unroll
              for (int k = 0; k < KB3; ++k) {
                acc3[c][k0 + k] = mfma_bf16(av[c].lo, bv[k].hi, acc3[c][k0 + k]);
                acc3[c][k0 + k] = mfma_bf16(av[c].hi, bv[k].lo, acc3[c][k0 + k]);
                acc3[c][k0 + k] = mfma_bf16(av[c].hi, bv[k].hi, acc3[c][k0 + k]);
              }
          }
        } else {
        bf16x8 av[CTW3];
#pragma unroll
        for (int c = 0; c < CTW3; ++c) av[c] = lds_col_to_bf16(&DA[eo + (ct0 + c) * 16], STR);
#pragma unroll
        for (int k0 = 0; k0 < KTW3; k0 += KB) {
          bf16x8 bv[KB];
#pragma unroll
          for (int k = 0; k < KB; ++k) bv[k] = lds_col_to_bf16(&F[eo + (kt0 + k0 + k) * 16], STR);
#pragma unroll
          for (int c = 0; c < CTW3; ++c)
#pragma unroll
            for (int k = 0; k < KB; ++k) acc3[c][k0 + k] = mfma_bf16(av[c], bv[k], acc3[c][k0 + k]);
        }
        }
      }
    } else {
#pragma unroll 2
      for (int s = ks3; s < ROWS / 4; s += KSPL3) {
        const int eo = (4 * s + lg) * STR + lr;
        float av[CTW3], bv[KTW3];
#pragma unroll
        for (int c = 0; c < CTW3; ++c) av[c] = DA[eo + (ct0 + c) * 16];
#pragma unroll
        for (int k = 0; k < KTW3; ++k) bv[k] = F[eo + (kt0 + k) * 16];
#pragma unroll
        for (int c = 0; c < CTW3; ++c)
#pragma unroll
          for (int k = 0; k < KTW3; ++k) acc3[c][k] = mfma16(av[c], bv[k], acc3[c][k]);
      }
    }
    BWD_PRIO(16);
    __syncthreads();
    if (LFA_BWD_DBG & 16) continue;  // timing experiment: phases 1-5
    // ---- phase 6: scatter dx; dy -> DA[:, D:2D]
#pragma unroll
    for (int m = 0; m < MTW; ++m)
#pragma unroll
      for (int t = 0; t < NTW; ++t) {
        const int col = (wn * NTW + t) * 16 + lr;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = (wm * MTW + m) * 16 + lg * 4 + r;
          const float v = acc[m][t][r];
          if (col < D) {
            if (CHP == 16) {
              DA[row * STR + col] = v;  // D < 16: only D of 16 lanes hold dx columns -> repacked below
            } else {
              const int j = nbr[row];
              if constexpr (FULL) {
                if (!(LFA_BWD_DBG & 1)) atomicAdd((float*)((char*)a.dx + ((unsigned)j * (unsigned)(D * 4) + (unsigned)(col * 4))), v);
              } else if (j >= 0 && !(LFA_BWD_DBG & 1)) atomicAdd(a.dx + (int64_t)j * D + col, v);
            }
          } else if (col < CH) {
            const float lse = F[row * STR + col];
            DA[row * STR + col] = v * (lse > 0.f ? 1.f : a.slope);
          }
        }
      }
    __syncthreads();
    if (CHP == 16) {
      // dx scatter with every lane busy: lane -> (edge row, column) over the ROWS x D block staged in DA, so one
      // wave-level atomic covers 64 / D whole rows instead of D of 16 lanes of a 16-column MFMA tile
      for (int f = tid; f < ROWS * D; f += NTHR) {
        const int row = f / D, col = f % D;
        const int j = nbr[row];
        if constexpr (FULL) {
          if (!(LFA_BWD_DBG & 1)) atomicAdd((float*)((char*)a.dx + ((unsigned)j * (unsigned)(D * 4) + (unsigned)(col * 4))), DA[row * STR + col]);
        } else if (j >= 0 && !(LFA_BWD_DBG & 1)) atomicAdd(a.dx + (int64_t)j * D + col, DA[row * STR + col]);
      }
    }
    if (LFA_BWD_DBG & 32) continue;  // timing experiment: phases 1-6
    // ---- phase 7: G[c', q] += sum_e dy[e, c'] * [r|1][e, q]
    if (wid < GT * KSPL4) {
      const bool crow = gt * 16 + lr < D;
#pragma unroll 4
      for (int s = ks4; s < ROWS / 4; s += KSPL4) {
        const int e = 4 * s + lg;
        const float av = crow ? DA[e * STR + D + gt * 16 + lr] : 0.f;
        const float bv = RT[e * RSTR + lr];
        accg = mfma16(av, bv, accg);
      }
    }
    __syncthreads();
  }

  // ---- write this workgroup's partials
  {
    float* dst = a.dw_part + ((size_t)blockIdx.x * KSPL3 + ks3) * (CHP * CHP);
#pragma unroll
    for (int c = 0; c < CTW3; ++c)
#pragma unroll
      for (int k = 0; k < KTW3; ++k)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          dst[((ct0 + c) * 16 + lg * 4 + r) * CHP + (kt0 + k) * 16 + lr] = acc3[c][k][r];
    if (wid < GT * KSPL4) {
      float* gd = a.g_part + ((size_t)blockIdx.x * KSPL4 + ks4) * (DP * 16);
#pragma unroll
      for (int r = 0; r < 4; ++r) gd[(gt * 16 + lg * 4 + r) * 16 + lr] = accg[r];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// ch <= 16, K = 16, complete neighbourhoods: WAVE-AUTONOMOUS backward (round 5).
// lfa_bwd_kernel at ch = 8 / 16 is four waves around one 128-edge tile with five workgroup barriers per tile and 18 MFMAs
// per wave between them: the SQ counters show the level-1 launches parked on memory / barriers for half of their wave-cycles
// at 30-35 % VALU issue (profiles/r05mid_pmc_sq*.csv) — 158 / 215 us for the work the forward kernels do in 21 / 35.
// Here a WAVE owns 64 LDS rows (64 edges; 128 at ch = 8, two centres per packed row as in lfa_fwd_full_kernel) from the gather
// to the atomics, in the in-lane neighbourhood layout (row 16 (k / 4) + 4 u + k % 4 = neighbour k of the wave's unit u), so
//   * there is no workgroup barrier in the loop: LDS traffic of one wave is processed in order, the phases are separated by
//     compiler fences only, and the four waves of a workgroup drift apart;
//   * the thread that owns an edge loads id -> (x_j, p_j) one trip ahead into registers (as PIPE does);
//   * the softmax backward is register arithmetic (lane (lr, lg) holds the 16 logits of unit lg for column lr);
//   * the encoder sums G = dy^T [r | 1] leave the matrix pipe (16 MFMAs of a tile that is 1/3 - 1/6 useful, plus an LDS tile of
//     r): the centre position and the 1 are constant per unit, so their columns come from the in-lane neighbour sums of dy in
//     the C layout (4 accumulators per lane); the thread that owns an edge accumulates dy * (p_j - p_i) and dy * |p_j - p_i|
//     (4 D accumulators per lane), and the p_j columns are the sum of the two (no cancellation: small added to large).
//     One cross-lane reduction per wave at the end of the kernel.
// Same partial-sum layout and workspace as lfa_bwd_kernel (four partials per workgroup); workgroups past `nwork` only write
// zero partials (the resident count is bounded by registers / LDS, the workspace by bwd_plan).
#ifndef LFA_BWD_SMALL
#define LFA_BWD_SMALL 1
#endif
#ifndef BWD_SMALL_CAP_8
#define BWD_SMALL_CAP_8 768
#endif
#ifndef BWD_SMALL_CAP_16
#define BWD_SMALL_CAP_16 768
#endif
#ifndef BWD_SMALL_MINW
#define BWD_SMALL_MINW 3
#endif
__device__ __forceinline__ float row16_sum_f(float v) {  // every lane of a 16-lane row ends with the row's sum (DPP)
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, false));   // quad_perm [2,3,0,1]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, false));  // row_half_mirror
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, false));  // row_mirror
  return v;
}
// LDS written by some lanes of a wave, read by others of the SAME wave: the LDS pipe keeps a wave's instructions in order, so
// only the compiler has to be kept from moving accesses across this point
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// EDGE (flags bit 5 of m3d_lfa_bwd): no atomics at all — the x-part of dF is stored per edge ([n K, D], plain 16 / 32-byte row
// stores) and the caller sums the rows of every point's REVERSE neighbour list (m3d_gather_sum_rows over the CSR inverse of
// the neighbour table, built with the position-only work).  Measured (profiles/r05p_*): with the atomics this kernel takes
// 193 / 199 us at level 1, without them 83 / 113 — the L2 needs ~30 ps per 16 / 32-byte row atomic whatever the kernel does.
template <int CH, bool EDGE, bool IOH = false>
__global__ __launch_bounds__(256, BWD_SMALL_MINW) void lfa_bwd_small_kernel(LfaBwdArgs a, int nwork) {
  constexpr bool PACK2 = CH == 8;
  constexpr int D = CH / 2, D4 = D / 4, STR = 18;
  constexpr int EPT = PACK2 ? 2 : 1;  // edges per lane and trip
  constexpr int EW = 64 * EPT;        // edges per wave and trip
  constexpr int UC = EW / 16;         // centres per wave and trip
  static_assert(CH == 8 || CH == 16, "one 16-column MFMA tile");
  __shared__ __attribute__((aligned(16))) float Fs[4 * 64 * STR];
  __shared__ __attribute__((aligned(16))) float DAs[4 * 64 * STR];
  __shared__ __attribute__((aligned(16))) int NBs[4 * EW];
  __shared__ float4 PCs[4 * UC];  // positions of the trip's centres
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, lr = lane & 15, lg = lane >> 4;
  float* const F = Fs + wid * 64 * STR;
  float* const DA = DAs + wid * 64 * STR;
  int* const NB = NBs + wid * EW;
  float4* const PC = PCs + wid * UC;
  float* const dwp = a.dw_part + ((size_t)blockIdx.x * 4 + wid) * 256;
  float* const gp = a.g_part + ((size_t)blockIdx.x * 4 + wid) * 256;
  if ((int)blockIdx.x >= nwork) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { dwp[lane + 64 * i] = 0.f; gp[lane + 64 * i] = 0.f; }
    return;
  }
  const unsigned n32 = (unsigned)a.n, elast = n32 * 16u - 1u;
  const int64_t ngroups = (a.n + 4 * UC - 1) / (4 * UC);
  int64_t g0, gs, gend;
  xcd_range(blockIdx.x, nwork, ngroups, g0, gs, gend);

  // where this lane's edges live: edge el = lane + 64 q of the trip = neighbour el % 16 of local centre el / 16
  int prow[EPT], hoff[EPT];
#pragma unroll
  for (int q = 0; q < EPT; ++q) {
    const int el = lane + 64 * q, cl = el >> 4, k = el & 15;
    const int unit = PACK2 ? cl >> 1 : cl;
    prow[q] = (k >> 2) * 16 + unit * 4 + (k & 3);
    hoff[q] = PACK2 ? (cl & 1) * 8 : 0;
  }
  // the two weight fragments are loop invariants (ch = 8: diag(W, W), see lfa_fwd_full_kernel)
  float4 bw, bwt;
  if constexpr (PACK2) {
    const float4 t = a.wp[lr < 8 ? lane : lane - 8], u = a.wpt[lr < 8 ? lane : lane - 8];
    bw = lr < 8 ? make_float4(t.x, t.y, 0.f, 0.f) : make_float4(0.f, 0.f, t.x, t.y);
    bwt = lr < 8 ? make_float4(u.x, u.y, 0.f, 0.f) : make_float4(0.f, 0.f, u.x, u.y);
  } else {
    bw = a.wp[lane];
    bwt = a.wpt[lane];
  }
  // folded encoder weights: read through the CONSTANT address space (nothing writes them during the kernel), so the loads inside
  // the loop stay scalar (s_load into SGPRs) although dx atomics precede them — as plain global loads they become 44 / 88
  // uniform vector loads per trip held in VGPRs
  typedef const float __attribute__((address_space(4))) * cptr_t;
  const cptr_t wfc = (cptr_t)(unsigned long long)a.wf, bfc = (cptr_t)(unsigned long long)a.bf;
  // column roles in the C layout: lane (lr, lg) holds column lr of the 16 rows of unit lg
  const int cc = PACK2 ? (lr & 7) : lr;  // channel inside its centre's CH columns
  const bool is_dx = cc < D;             // x-part column (scattered to dx) or encoder column (dy)

  f32x4 acc3 = (f32x4){0.f, 0.f, 0.f, 0.f};
  float accd[D][4];  // this lane's edges: sum dy[c] * (dx, dy, dz, length)
#pragma unroll
  for (int c = 0; c < D; ++c)
#pragma unroll
    for (int t = 0; t < 4; ++t) accd[c][t] = 0.f;
  float accp[4] = {0.f, 0.f, 0.f, 0.f};  // this lane's (unit, encoder column): sum dy * (p_i, 1)

  unsigned jc[EPT], jn[EPT];  // ids of the trip whose rows are in flight / of the trip after it (two trips ahead of the compute)
  unsigned sc[EPT], sn[EPT];  // (EDGE with a slot table) the rows of dxe these edges are stored in, fetched with the ids
  const bool slotted = EDGE && a.eslot != nullptr;
  float4 pi[EPT], pj[EPT], xg[EPT][D4];
  float dgn = 0.f;
  auto ld_ids = [&](int64_t g) {
    const unsigned c0w = ((unsigned)g * 4u + (unsigned)wid) * UC;
#pragma unroll
    for (int q = 0; q < EPT; ++q) {
      unsigned eo = c0w * 16u + (unsigned)(lane + 64 * q);
      eo = eo < elast ? eo : elast;
      jn[q] = (unsigned)a.idx[eo];
      if constexpr (EDGE) sn[q] = slotted ? (unsigned)a.eslot[eo] : eo;
    }
  };
  auto ld_rows = [&](int64_t g) {  // everything of trip g that hangs on its ids (jc), plus dout
    const unsigned c0w = ((unsigned)g * 4u + (unsigned)wid) * UC;
#pragma unroll
    for (int q = 0; q < EPT; ++q) {
      unsigned ci = c0w + (unsigned)((lane + 64 * q) >> 4);
      ci = ci < n32 ? ci : n32 - 1u;
      pi[q] = *(const float4*)((const char*)a.pos4 + ci * 16u);
      pj[q] = *(const float4*)((const char*)a.pos4 + jc[q] * 16u);
#pragma unroll
      for (int u = 0; u < D4; ++u) xg[q][u] = io_load4_b<IOH>(a.x, jc[q] * (unsigned)(D * 4) + (unsigned)(u * 16));
    }
    unsigned cu = PACK2 ? c0w + 2u * (unsigned)lg + (unsigned)(lr >> 3) : c0w + (unsigned)lg;
    const bool ok = cu < n32;  // (centres past n: clamped rows with dout = 0 — every contribution an exact zero)
    cu = ok ? cu : n32 - 1u;
    const float v = io_load1_b<IOH>(a.dout, cu * (unsigned)(CH * 4) + (unsigned)(cc * 4));
    dgn = ok ? v : 0.f;
  };
  if (g0 < gend) {
    ld_ids(g0);
#pragma unroll
    for (int q = 0; q < EPT; ++q) { jc[q] = jn[q]; sc[q] = sn[q]; }
    ld_rows(g0);
    ld_ids(g0 + gs);
  }
  for (int64_t grp = g0; grp < gend; grp += gs) {
    // ---- phase 1: x_j, folded encoder -> F; ids -> NB
    float rd[EPT][4];
    const float dg = dgn;
    // ch = 16: 88 weights + the kernel's other scalars exceed the SGPR file once the loads are hoisted out of the loop (78
    // v_readlane / v_writelane spills in the ISA): an offset the optimiser cannot see through keeps the s_loads inside the trip
    int wz = 0;
    if constexpr (CH == 16) asm volatile("s_mov_b32 %0, 0" : "=s"(wz));
#pragma unroll
    for (int q = 0; q < EPT; ++q) {
      float r[10];
      rel_pos_fast(pi[q], pj[q], r);
      float* frow = F + prow[q] * STR + hoff[q];
#pragma unroll
      for (int u = 0; u < D4; ++u) {
        *(float2*)(frow + u * 4) = make_float2(xg[q][u].x, xg[q][u].y);
        *(float2*)(frow + u * 4 + 2) = make_float2(xg[q][u].z, xg[q][u].w);
      }
#pragma unroll
      for (int c = 0; c < D; ++c) {
        float v = bfc[wz + c];
#pragma unroll
        for (int t = 0; t < 10; ++t) v += wfc[wz + c * 10 + t] * r[t];
        frow[D + c] = fmaxf(v, v * a.slope);
      }
      NB[(PACK2 ? (hoff[q] >> 3) * 64 : 0) + prow[q]] = EDGE ? (int)sc[q] : (int)jc[q];  // (phase 6 needs one or the other)
      rd[q][0] = r[6]; rd[q][1] = r[7]; rd[q][2] = r[8]; rd[q][3] = r[9];
      if ((lane & 15) == 0) PC[(lane + 64 * q) >> 4] = pi[q];
    }
    // next trip's loads fly during the phases below (past the end of this workgroup's range: loaded, never used)
#pragma unroll
    for (int q = 0; q < EPT; ++q) { jc[q] = jn[q]; sc[q] = sn[q]; }
    ld_rows(grp + gs);
    ld_ids(grp + 2 * gs);
    wave_lds_fence();

    // ---- phase 2: A = F W_att^T
    f32x4 acc[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
    {
      const float* fa = F + lr * STR + lg;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float bv = i == 0 ? bw.x : (i == 1 ? bw.y : (i == 2 ? bw.z : bw.w));
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[m] = mfma16(fa[m * 16 * STR + i * 4], bv, acc[m]);
      }
    }
    // ---- phase 3': softmax over the unit's 16 neighbours, dA -> DA, acc <- dout * s
    {
      const float pinf = fast_pinf();
      const float* fcol = F + lg * 4 * STR + lr;
      float* dcol = DA + lg * 4 * STR + lr;
      float mx = acc[0][0];
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (m + r > 0) mx = fast_max(mx, acc[m][r], pinf);
      const float ml = mx * 1.4426950408889634f;
      float num = 0.f, den = 0.f;
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(acc[m][r], 1.4426950408889634f, -ml));
          const float f = fcol[(m * 16 + r) * STR];
          num = __builtin_fmaf(p, f, num);
          den += p;
          acc[m][r] = p;
        }
      const float inv = __builtin_amdgcn_rcpf(den + 1e-16f);
      const float o = num * inv, gi = dg * inv;
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float gsw = acc[m][r] * gi;  // dout * softmax weight
          dcol[(m * 16 + r) * STR] = gsw * (fcol[(m * 16 + r) * STR] - o);  // (F re-read: 16 registers less across the phase)
          acc[m][r] = gsw;
        }
    }
    wave_lds_fence();
    // ---- phase 4: dF = dout * s + DA W_att
    {
      const float* da = DA + lr * STR + lg;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float bv = i == 0 ? bwt.x : (i == 1 ? bwt.y : (i == 2 ? bwt.z : bwt.w));
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[m] = mfma16(da[m * 16 * STR + i * 4], bv, acc[m]);
      }
    }
    // ---- phase 5: dW_att += DA^T F over the wave's 64 rows (ch = 8: the 16 x 16 product of the PACKED tiles, whose two
    // diagonal 8 x 8 blocks are added at the end of the kernel)
    {
      const float* da = DA + lg * STR + lr;
      const float* fb = F + lg * STR + lr;
#pragma unroll
      for (int s = 0; s < 16; ++s) acc3 = mfma16(da[4 * s * STR], fb[4 * s * STR], acc3);
    }
    // ---- phase 6: x-part columns -> dx atomics; encoder columns -> dy = dF * LeakyReLU'(lse) -> DA, neighbour sums of dy
    {
      const int* nb = NB + (PACK2 ? (lr >> 3) * 64 : 0) + lg * 4;
      if (is_dx && EDGE) {
        const unsigned ci = ((unsigned)grp * 4u + (unsigned)wid) * UC + (unsigned)(PACK2 ? 2 * lg + (lr >> 3) : lg);
        if (ci < n32) {  // (edges of centres past n are clamped copies of the last edge: their rows must not be written)
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            const int4 s4 = *(const int4*)(nb + m * 16);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const unsigned row = (unsigned)(r == 0 ? s4.x : (r == 1 ? s4.y : (r == 2 ? s4.z : s4.w)));
              io_store1_b<IOH>(a.dxe, row * (unsigned)(D * 4) + (unsigned)(cc * 4), acc[m][r]);
            }
          }
        }
      } else if (is_dx) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const int4 j4 = *(const int4*)(nb + m * 16);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const unsigned j = (unsigned)(r == 0 ? j4.x : (r == 1 ? j4.y : (r == 2 ? j4.z : j4.w)));
            if (!(LFA_BWD_DBG & 1)) atomicAdd((float*)((char*)a.dx + (j * (unsigned)(D * 4) + (unsigned)(cc * 4))), acc[m][r]);
          }
        }
      } else {
        const float* fcol = F + lg * 4 * STR + lr;
        float* dcol = DA + lg * 4 * STR + lr;
        float sdy = 0.f;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float lse = fcol[(m * 16 + r) * STR];
            const float dy = acc[m][r] * (lse > 0.f ? 1.f : a.slope);
            dcol[(m * 16 + r) * STR] = dy;
            sdy += dy;
          }
        const float4 pu = PC[PACK2 ? 2 * lg + (lr >> 3) : lg];
        accp[0] = __builtin_fmaf(sdy, pu.x, accp[0]);
        accp[1] = __builtin_fmaf(sdy, pu.y, accp[1]);
        accp[2] = __builtin_fmaf(sdy, pu.z, accp[2]);
        accp[3] += sdy;
      }
    }
    wave_lds_fence();
    // ---- phase 7: this lane's edges: dy * (p_j - p_i, |p_j - p_i|)
#pragma unroll
    for (int q = 0; q < EPT; ++q) {
      const float* dyr = DA + prow[q] * STR + hoff[q] + D;
#pragma unroll
      for (int c2 = 0; c2 < D; c2 += 2) {
        const float2 dy = *(const float2*)(dyr + c2);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          accd[c2][t] = __builtin_fmaf(dy.x, rd[q][t], accd[c2][t]);
          accd[c2 + 1][t] = __builtin_fmaf(dy.y, rd[q][t], accd[c2 + 1][t]);
        }
      }
    }
    wave_lds_fence();
  }

  // ---- this wave's partials
  {
    if constexpr (PACK2) {
#pragma unroll
      for (int r = 0; r < 4; ++r) F[(lg * 4 + r) * STR + lr] = acc3[r];
      wave_lds_fence();
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = lg * 4 + r;
        dwp[c * 16 + lr] = (c < 8 && lr < 8) ? F[c * STR + lr] + F[(c + 8) * STR + lr + 8] : 0.f;
      }
      wave_lds_fence();
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) dwp[(lg * 4 + r) * 16 + lr] = acc3[r];
    }
    // G[c][0..2] = sum dy p_i, [3..5] = sum dy p_j = [6..8] + [0..2], [6..8] = sum dy (p_j - p_i), [9] = sum dy |p_j - p_i|, [10] = sum dy
    float* PL = F;            // [D][4]
    float* DL = F + D * 4;    // [D][4]
    float ps[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) ps[t] = xgroup_sum(accp[t]);  // over the four units of the wave
    if constexpr (PACK2) {
      // columns 4 + c and 12 + c hold the two centres of a pair: lane lr and lane lr ^ 8
#pragma unroll
      for (int t = 0; t < 4; ++t) ps[t] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(ps[t]), 0x128, 0xF, 0xF, false));  // row_ror:8
    }
    if (lg == 0 && !is_dx && lr < 8 + (PACK2 ? 0 : 8)) {
#pragma unroll
      for (int t = 0; t < 4; ++t) PL[(cc - D) * 4 + t] = ps[t];
    }
#pragma unroll
    for (int c = 0; c < D; ++c)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float v = xgroup_sum(row16_sum_f(accd[c][t]));
        if (lane == 0) DL[c * 4 + t] = v;
      }
    wave_lds_fence();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = lane + 64 * i, c = e >> 4, t = e & 15;
      float v = 0.f;
      if (c < D && t < 11) {
        if (t < 3) v = PL[c * 4 + t];
        else if (t < 6) v = PL[c * 4 + t - 3] + DL[c * 4 + t - 3];
        else if (t < 10) v = DL[c * 4 + t - 6];
        else v = PL[c * 4 + 3];
      }
      gp[e] = v;
    }
  }
}

// (round 4 prepared eight partials in flight in both loops of this reduce — -DLFA_RED_WIDE=1; its A/B in round 5 moved nothing,
// profiles/r05a_step_lfa_full_ab.log; removed)
// sums the workgroup partials: dw_att[CH, CH] (fp32) and G[D, 11] (fp64).  blockIdx.x owns 256 consecutive
// elements, blockIdx.y a contiguous chunk of the partials; chunks are combined with one atomic per element per
// chunk into the (pre-zeroed) outputs, so the pass runs at HBM speed instead of one block walking every partial.
__device__ __forceinline__ void lfa_bwd_reduce_body(const float* __restrict__ dw_part, int parts3, int CHP, int CH,
                                                    float* __restrict__ dw_att, const float* __restrict__ g_part,
                                                    int parts4, int DP, int D, double* __restrict__ G, unsigned bx,
                                                    unsigned by, unsigned nyy) {
  const int t = bx * 256 + threadIdx.x;
  const int nw = CHP * CHP, nq = nw >> 2;  // (CHP is a multiple of 16: a thread sums FOUR consecutive elements, 16-byte loads)
  const int ny = (int)nyy, y = (int)by;
  if (t < nq) {
    const int per = (parts3 + ny - 1) / ny;
    const int p0 = y * per, p1 = min(parts3, p0 + per);
    // fp64 like G below (round 6): a chunk holds up to a few hundred partials of cancelling terms (dA sums to zero over every
    // neighbourhood) — the loads bind this pass, the wider adds are free
    const float4* __restrict__ src = (const float4*)dw_part + t;
    double s[4] = {0.0, 0.0, 0.0, 0.0}, r[4] = {0.0, 0.0, 0.0, 0.0};
    int p = p0;
    for (; p + 7 < p1; p += 8) {  // eight 16-byte loads in flight per thread
      float4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = src[(size_t)(p + j) * nq];
#pragma unroll
      for (int j = 0; j < 8; j += 2) {
        s[0] += (double)v[j].x; s[1] += (double)v[j].y; s[2] += (double)v[j].z; s[3] += (double)v[j].w;
        r[0] += (double)v[j + 1].x; r[1] += (double)v[j + 1].y; r[2] += (double)v[j + 1].z; r[3] += (double)v[j + 1].w;
      }
    }
    for (; p < p1; ++p) {
      const float4 a = src[(size_t)p * nq];
      s[0] += (double)a.x; s[1] += (double)a.y; s[2] += (double)a.z; s[3] += (double)a.w;
    }
    const int e = 4 * t, c = e / CHP, k = e % CHP;
    if (c < CH && p1 > p0) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (k + j < CH) atomicAdd(&dw_att[c * CH + k + j], (float)(s[j] + r[j]));
    }
  } else {
    const int u = t - nq;
    if (u < DP * 16) {
      const int per = (parts4 + ny - 1) / ny;
      const int p0 = y * per, p1 = min(parts4, p0 + per);
      // (eight partials in flight: one load at a time this loop was a chain of round trips as long as the chunk — the pass
      // took the same ~70 us whatever its dW half did)
      double s = 0.0;
      const size_t gs = (size_t)DP * 16;
      int p = p0;
      for (; p + 7 < p1; p += 8) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = g_part[(size_t)(p + j) * gs + u];
#pragma unroll
        for (int j = 0; j < 8; ++j) s += (double)v[j];
      }
      for (; p < p1; ++p) s += (double)g_part[(size_t)p * gs + u];
      const int c = u / 16, q = u % 16;
      if (c < D && q < 11 && p1 > p0) atomicAdd(&G[c * 11 + q], s);
    }
  }
}

__global__ __launch_bounds__(256) void lfa_bwd_reduce_kernel(const float* __restrict__ dw_part, int parts3, int CHP,
                                                             int CH, float* __restrict__ dw_att,
                                                             const float* __restrict__ g_part, int parts4, int DP,
                                                             int D, double* __restrict__ G) {
  lfa_bwd_reduce_body(dw_part, parts3, CHP, CH, dw_att, g_part, parts4, DP, D, G, blockIdx.x, blockIdx.y, gridDim.y);
}

// the partial sums of several LFA layers in one launch (m3d_lfa_bwd_reduce_batch): dW_att and the encoder sums G are
// PARAMETER-gradient material — nothing in the backward chain reads them — so every layer's reduce can wait for the
// end of the backward pass (8 launches of 10-20 us in the dependent chain otherwise)
#ifndef LFA_RED_WGS
#define LFA_RED_WGS 256  // workgroups per layer of the partial-sum reduce: every chunk of partials costs one float atomic per element
#endif
#define LFA_RED_BATCH_MAX 16
struct LfaRedBatch {
  const float* dw_part[LFA_RED_BATCH_MAX]; const float* g_part[LFA_RED_BATCH_MAX];
  float* dw_att[LFA_RED_BATCH_MAX]; double* G[LFA_RED_BATCH_MAX];
  int parts3[LFA_RED_BATCH_MAX], parts4[LFA_RED_BATCH_MAX], chp[LFA_RED_BATCH_MAX], ch[LFA_RED_BATCH_MAX], dp[LFA_RED_BATCH_MAX];
  unsigned gx[LFA_RED_BATCH_MAX], gy[LFA_RED_BATCH_MAX], wg_start[LFA_RED_BATCH_MAX + 1];
  int njobs;
};
__global__ __launch_bounds__(256) void lfa_bwd_reduce_batch_kernel(LfaRedBatch b) {
  int j = 0;
#pragma unroll
  for (int i = 1; i < LFA_RED_BATCH_MAX; ++i) j += (i < b.njobs && blockIdx.x >= b.wg_start[i]) ? 1 : 0;
  const unsigned w = blockIdx.x - b.wg_start[j];
  lfa_bwd_reduce_body(b.dw_part[j], b.parts3[j], b.chp[j], b.ch[j], b.dw_att[j], b.g_part[j], b.parts4[j], b.dp[j],
                      b.ch[j] / 2, b.G[j], w % b.gx[j], w / b.gx[j], b.gy[j]);
}

struct BwdPlan { int chp, rows, grid, kspl3, kspl4, dp; };

static inline BwdPlan bwd_plan(int64_t n, int K, int CH) {
  BwdPlan p;
  p.chp = CH < 16 ? 16 : CH;
  p.rows = p.chp == 16 ? BwdCfg<16>::ROWS : (p.chp == 32 ? BwdCfg<32>::ROWS : (p.chp == 64 ? BwdCfg<64>::ROWS : (p.chp == 128 ? BwdCfg<128>::ROWS : BwdCfg<256>::ROWS)));
  const int kp = K <= 16 ? 16 : 32;
  const int tc = p.rows / kp;
  const int nw = p.chp == 16 ? BwdCfg<16>::NW : (p.chp == 32 ? BwdCfg<32>::NW : (p.chp == 64 ? BwdCfg<64>::NW : (p.chp == 128 ? BwdCfg<128>::NW : BwdCfg<256>::NW)));
  const int nt = p.chp / 16;
  p.kspl3 = nt * nt >= nw ? 1 : nw / (nt * nt);
  p.dp = p.chp / 2 < 16 ? 16 : p.chp / 2;
  const int gt = p.dp / 16;
  p.kspl4 = gt >= nw ? 1 : nw / gt;
  const int64_t ngroups = m3d_cdiv(n, tc);
  // resident workgroups: bounded by LDS (2 tiles of rows*(chp+2) floats)
  const int cap = p.chp == 16 ? BwdCfg<16>::CAP : (p.chp == 32 ? BwdCfg<32>::CAP : (p.chp == 64 ? BwdCfg<64>::CAP : (p.chp == 128 ? BwdCfg<128>::CAP : BwdCfg<256>::CAP)));
  p.grid = (int)(ngroups < cap ? (ngroups < 1 ? 1 : ngroups) : cap);
  return p;
}

extern "C" size_t m3d_lfa_bwd_workspace_bytes(int64_t n, int32_t K, int32_t CH) {
  if (n < 0 || K < 1 || CH < 8) return 0;
  BwdPlan p = bwd_plan(n, K, CH);
  return ((size_t)p.grid * p.kspl3 * p.chp * p.chp + (size_t)p.grid * p.kspl4 * p.dp * 16) * sizeof(float) + 256;
}

template <int CH, bool IOH>
static int launch_lfa_bwd_io(const LfaBwdArgs& a, const BwdPlan& p, hipStream_t st, bool bf16, bool full, bool x3) {
  constexpr int NTHR = BwdCfg<(CH < 16 ? 16 : CH)>::NW * 64;
  // software-pipelined variant: per channel count where it measured faster (profiles/r01p_*; BWD_PIPE_* at compile time)
  constexpr bool pipe = CH <= 64 && !(LFA_BWD_DBG & ~1) &&  // (bit 0, no dx atomics, keeps the pipelined kernel: no `continue` in it)
                        (CH == 8 ? BWD_PIPE_8 : (CH == 16 ? BWD_PIPE_16 : (CH == 32 ? BWD_PIPE_32 : BWD_PIPE_64)));
  if constexpr (CH >= 32) {
    if (bf16) {  // bf16 matrix-core operands for the three attention GEMMs
      constexpr bool P = pipe;
      if (full && x3) {
        if (a.K == 16) hipLaunchKernelGGL((lfa_bwd_kernel<CH, 16, P, true, true, true, IOH>), dim3(p.grid), dim3(NTHR), 0, st, a);
        else hipLaunchKernelGGL((lfa_bwd_kernel<CH, 32, P, true, true, true, IOH>), dim3(p.grid), dim3(NTHR), 0, st, a);
      } else if (x3) {
        return M3D_ERR_UNSUPPORTED;  // (the split-bf16 product exists in the complete-neighbourhood kernels only)
      } else if (full) {
        if (a.K == 16) hipLaunchKernelGGL((lfa_bwd_kernel<CH, 16, P, true, true, false, IOH>), dim3(p.grid), dim3(NTHR), 0, st, a);
        else hipLaunchKernelGGL((lfa_bwd_kernel<CH, 32, P, true, true, false, IOH>), dim3(p.grid), dim3(NTHR), 0, st, a);
      } else if (a.K <= 16) hipLaunchKernelGGL((lfa_bwd_kernel<CH, 16, P, true, false, false, IOH>), dim3(p.grid), dim3(NTHR), 0, st, a);
      else hipLaunchKernelGGL((lfa_bwd_kernel<CH, 32, P, true, false, false, IOH>), dim3(p.grid), dim3(NTHR), 0, st, a);
      return hipGetLastError() == hipSuccess ? M3D_OK : M3D_ERR_LAUNCH;
    }
  }
  if (bf16) return M3D_ERR_UNSUPPORTED;
  if constexpr (CH <= 16) {
    if (full && a.K == 16 && LFA_BWD_SMALL) {
      constexpr int UC4 = CH == 8 ? 32 : 16;  // centres per workgroup trip
      const int64_t ng = (a.n + UC4 - 1) / UC4;
      int nwork = p.grid < (CH == 8 ? BWD_SMALL_CAP_8 : BWD_SMALL_CAP_16) ? p.grid : (CH == 8 ? BWD_SMALL_CAP_8 : BWD_SMALL_CAP_16);
      if (ng < nwork) nwork = (int)ng;
      if (a.dxe) hipLaunchKernelGGL((lfa_bwd_small_kernel<CH, true, IOH>), dim3(p.grid), dim3(256), 0, st, a, nwork);
      else hipLaunchKernelGGL((lfa_bwd_small_kernel<CH, false, IOH>), dim3(p.grid), dim3(256), 0, st, a, nwork);
      return hipGetLastError() == hipSuccess ? M3D_OK : M3D_ERR_LAUNCH;
    }
  }
  if (a.dxe) return M3D_ERR_UNSUPPORTED;  // (edge rows exist in the wave-autonomous kernels only: m3d_lfa_bwd_edge_rows_ok)
  if (full) {
    if (a.K == 16) hipLaunchKernelGGL((lfa_bwd_kernel<CH, 16, pipe, false, true, false, IOH>), dim3(p.grid), dim3(NTHR), 0, st, a);
    else hipLaunchKernelGGL((lfa_bwd_kernel<CH, 32, pipe, false, true, false, IOH>), dim3(p.grid), dim3(NTHR), 0, st, a);
  } else if (a.K <= 16) hipLaunchKernelGGL((lfa_bwd_kernel<CH, 16, pipe, false, false, false, IOH>), dim3(p.grid), dim3(NTHR), 0, st, a);
  else hipLaunchKernelGGL((lfa_bwd_kernel<CH, 32, pipe, false, false, false, IOH>), dim3(p.grid), dim3(NTHR), 0, st, a);
  return hipGetLastError() == hipSuccess ? M3D_OK : M3D_ERR_LAUNCH;
}

template <int CH>
static int launch_lfa_bwd(const LfaBwdArgs& a, const BwdPlan& p, hipStream_t st, bool bf16, bool full, bool x3, bool io16) {
  if (io16) return launch_lfa_bwd_io<CH, true>(a, p, st, bf16, full, x3);
  return launch_lfa_bwd_io<CH, false>(a, p, st, bf16, full, x3);
}

// 1: m3d_lfa_bwd(flags | 8 | 32) is honoured for this layer shape (the wave-autonomous kernels: ch <= 16, K = 16)
extern "C" int m3d_lfa_bwd_edge_rows_ok(int64_t n, int32_t K, int32_t CH, float slope) {
  const int64_t lim = (int64_t)1 << 31;
  return LFA_BWD_SMALL && !LFA_BWD_DBG_NOFULL && (CH == 8 || CH == 16) && K == 16 && n > 0 && n * K * (int64_t)(CH / 2) * 4 < lim &&
         n * (int64_t)CH * 4 < lim && slope >= 0.f && slope <= 1.f;
}

static int lfa_bwd_impl(const float* x, const float* pos4, const int32_t* idx, int64_t n, int32_t K, int32_t CH,
                        const float* enc_w_folded, const float* enc_b_folded, const void* att_w_packed,
                        const void* att_wt_packed, float slope, const float* dout, float* dx, float* dw_att,
                        int32_t flags, double* G, void* ws, void* stream, bool bf16, const int32_t* edge_slot = nullptr) {
  if (n < 0 || K < 1 || CH < 8) return M3D_ERR_INVALID;
  if (K > 32) return M3D_ERR_UNSUPPORTED;
  if (CH != 8 && CH != 16 && CH != 32 && CH != 64 && CH != 128 && CH != 256) return M3D_ERR_UNSUPPORTED;
  if (bf16 && CH < 32) return M3D_ERR_UNSUPPORTED;
  if (!dw_att || !G || !ws) return M3D_ERR_INVALID;
  if (n > 0 && (!x || !pos4 || !idx || !enc_w_folded || !enc_b_folded || !att_w_packed || !att_wt_packed || !dout || !dx))
    return M3D_ERR_INVALID;
  const int accumulate_dw = flags & 1, g_is_zero = flags & 2;
  hipStream_t st = (hipStream_t)stream;
  BwdPlan p = bwd_plan(n, K, CH);
  LfaBwdArgs a;
  a.x = x; a.pos4 = (const float4*)pos4; a.idx = idx; a.wf = enc_w_folded; a.bf = enc_b_folded;
  a.wp = (const float4*)att_w_packed; a.wpt = (const float4*)att_wt_packed; a.dout = dout; a.dx = dx;
  a.dxe = nullptr; a.eslot = nullptr;
  a.dw_part = (float*)ws;
  a.g_part = a.dw_part + (size_t)p.grid * p.kspl3 * p.chp * p.chp;
  a.n = n; a.K = K; a.CH = CH; a.D = CH / 2; a.slope = slope;
  int rc;
  // flags bit 3 (M3D_LFA_BWD_FULL): complete neighbourhoods promised; taken when K fills its MFMA tiles, every byte offset
  // fits 32 bits and LeakyReLU can be written max(v, slope v)
  const int64_t lim = (int64_t)1 << 31;
  const bool full = (flags & 8) && (K == 16 || K == 32) && n > 0 && n * K < lim && n * (int64_t)CH * 4 < lim && n * 16 < lim &&
                    slope >= 0.f && slope <= 1.f && !LFA_BWD_DBG_NOFULL;
  const bool x3 = bf16 && (flags & 16);  // flags bit 4: split-bf16 operands (att_w*_packed hold hi then lo fragments)
  const bool io16 = (flags & M3D_IO_BF16) != 0;  // x, dout and the edge rows hold bf16 (dx, accumulated with atomics: fp32)
  if (flags & 32) {  // flags bit 5: `dx` is the [n K, D] edge-row buffer (stored, not accumulated)
    if (!full || bf16 || !m3d_lfa_bwd_edge_rows_ok(n, K, CH, slope)) return M3D_ERR_UNSUPPORTED;
    a.dxe = dx;
    a.dx = nullptr;
    a.eslot = edge_slot;
  } else if (edge_slot) return M3D_ERR_INVALID;
  switch (CH) {
    case 8: rc = launch_lfa_bwd<8>(a, p, st, bf16, full, x3, io16); break;
    case 16: rc = launch_lfa_bwd<16>(a, p, st, bf16, full, x3, io16); break;
    case 32: rc = launch_lfa_bwd<32>(a, p, st, bf16, full, x3, io16); break;
    case 64: rc = launch_lfa_bwd<64>(a, p, st, bf16, full, x3, io16); break;
    case 128: rc = launch_lfa_bwd<128>(a, p, st, bf16, full, x3, io16); break;
    default: rc = launch_lfa_bwd<256>(a, p, st, bf16, full, x3, io16); break;
  }
  if (rc != M3D_OK) return rc;
  if (flags & 4) return M3D_OK;  // the caller sums the partials later (m3d_lfa_bwd_reduce_batch): ws must live until then
  const int total = p.chp * p.chp / 4 + p.dp * 16;  // (a thread sums four dW elements or one G element)
  const int gx = (total + 255) / 256;
  const int parts = p.grid * p.kspl3;
  int gy = LFA_RED_WGS / gx;     // blocks in flight per layer; >= 8 partials per chunk
  if (gy > parts / 8) gy = parts / 8;
  if (gy < 1) gy = 1;
  if (!accumulate_dw && hipMemsetAsync(dw_att, 0, sizeof(float) * (size_t)CH * CH, st) != hipSuccess)
    return M3D_ERR_LAUNCH;
  if (!g_is_zero && hipMemsetAsync(G, 0, sizeof(double) * 11 * (size_t)(CH / 2), st) != hipSuccess) return M3D_ERR_LAUNCH;
  hipLaunchKernelGGL(lfa_bwd_reduce_kernel, dim3(gx, gy), dim3(256), 0, st, a.dw_part, parts, p.chp, CH, dw_att,
                     a.g_part, p.grid * p.kspl4, p.dp, CH / 2, G);
  M3D_CHECK_LAUNCH();
  return M3D_OK;
}

// the deferred reduces of m3d_lfa_bwd(..., flags | 4) calls: job j = the layer launched with (n[j], K[j], CH[j], ws[j]);
// its partial sums are ADDED into dw_att[j] (a gradient sink) and G[j] (pre-zeroed [11 * CH / 2] doubles)
extern "C" int m3d_lfa_bwd_reduce_batch(int32_t njobs, const int64_t* n, const int32_t* K, const int32_t* CH,
                                        void* const* ws, float* const* dw_att, double* const* G, void* stream) {
  if (njobs < 0) return M3D_ERR_INVALID;
  if (njobs == 0) return M3D_OK;
  if (!n || !K || !CH || !ws || !dw_att || !G) return M3D_ERR_INVALID;
  hipStream_t st = (hipStream_t)stream;
  for (int j0 = 0; j0 < njobs; j0 += LFA_RED_BATCH_MAX) {
    LfaRedBatch b;
    const int m = njobs - j0 < LFA_RED_BATCH_MAX ? njobs - j0 : LFA_RED_BATCH_MAX;
    b.njobs = m;
    unsigned total_wg = 0;
    for (int i = 0; i < LFA_RED_BATCH_MAX; ++i) {
      b.wg_start[i] = total_wg;
      if (i >= m) {
        b.dw_part[i] = nullptr; b.g_part[i] = nullptr; b.dw_att[i] = nullptr; b.G[i] = nullptr;
        b.parts3[i] = b.parts4[i] = 0; b.chp[i] = b.dp[i] = 16; b.ch[i] = 8; b.gx[i] = b.gy[i] = 1;
        continue;
      }
      const int j = j0 + i;
      if (n[j] < 0 || K[j] < 1 || K[j] > 32 || !ws[j] || !dw_att[j] || !G[j]) return M3D_ERR_INVALID;
      if (CH[j] != 8 && CH[j] != 16 && CH[j] != 32 && CH[j] != 64 && CH[j] != 128 && CH[j] != 256) return M3D_ERR_UNSUPPORTED;
      const BwdPlan p = bwd_plan(n[j], K[j], CH[j]);
      b.dw_part[i] = (const float*)ws[j];
      b.g_part[i] = b.dw_part[i] + (size_t)p.grid * p.kspl3 * p.chp * p.chp;
      b.dw_att[i] = dw_att[j]; b.G[i] = G[j];
      b.parts3[i] = p.grid * p.kspl3; b.parts4[i] = p.grid * p.kspl4; b.chp[i] = p.chp; b.ch[i] = CH[j]; b.dp[i] = p.dp;
      const int total = p.chp * p.chp / 4 + p.dp * 16;
      const int gx = (total + 255) / 256;
      int gy = LFA_RED_WGS / gx;
      if (gy > b.parts3[i] / 8) gy = b.parts3[i] / 8;
      if (gy < 1) gy = 1;
      b.gx[i] = (unsigned)gx; b.gy[i] = (unsigned)gy;
      total_wg += (unsigned)(gx * gy);
    }
    b.wg_start[LFA_RED_BATCH_MAX] = total_wg;
    if (total_wg) hipLaunchKernelGGL(lfa_bwd_reduce_batch_kernel, dim3(total_wg), dim3(256), 0, st, b);
  }
  M3D_CHECK_LAUNCH();
  return M3D_OK;
}

// flags: bit 0 = add into dw_att (gradient sink) instead of overwriting it, bit 1 = G is already zero (skips a memset),
// bit 2 = do not sum the partials (m3d_lfa_bwd_reduce_batch does, later; dw_att / G are not touched)
extern "C" int m3d_lfa_bwd(const float* x, const float* pos4, const int32_t* idx, int64_t n, int32_t K, int32_t CH,
                           const float* enc_w_folded, const float* enc_b_folded, const float* att_w_packed,
                           const float* att_wt_packed, float slope, const float* dout, float* dx, float* dw_att,
                           int32_t flags, double* G, void* ws, void* stream) {
  return lfa_bwd_impl(x, pos4, idx, n, K, CH, enc_w_folded, enc_b_folded, att_w_packed, att_wt_packed, slope, dout, dx,
                      dw_att, flags, G, ws, stream, false);
}

// m3d_lfa_bwd(flags | 8 | 32) with the edge rows stored in REVERSE-LIST order: row edge_slot[i * K + k] of dx_edges (the slot
// table of m3d_knn_reverse) holds the gradient edge (i, k) sends to x[idx[i][k]] — the rows of a point are then contiguous and
// m3d_gather_sum_rows(inv = NULL) streams them (the edge-order gather fetches 2-4 x the rows' bytes: 64-byte fetches of 16 /
// 32-byte rows scattered over the table, profiles/r05zfin_pmc_fetch_size.csv)
extern "C" int m3d_lfa_bwd_edge_rows(const float* x, const float* pos4, const int32_t* idx, int64_t n, int32_t K, int32_t CH,
                                     const float* enc_w_folded, const float* enc_b_folded, const float* att_w_packed,
                                     const float* att_wt_packed, float slope, const float* dout, float* dx_edges,
                                     const int32_t* edge_slot, float* dw_att, int32_t flags, double* G, void* ws, void* stream) {
  return lfa_bwd_impl(x, pos4, idx, n, K, CH, enc_w_folded, enc_b_folded, att_w_packed, att_wt_packed, slope, dout, dx_edges,
                      dw_att, flags | 8 | 32, G, ws, stream, false, edge_slot);
}

// bf16 matrix-core variant (CH in {64, 128, 256}; att_w*_packed: m3d_lfa_pack_att_bf16 / m3d_lfa_prepare(bf16 = 1))
extern "C" int m3d_lfa_bwd_bf16(const float* x, const float* pos4, const int32_t* idx, int64_t n, int32_t K, int32_t CH,
                                const float* enc_w_folded, const float* enc_b_folded, const void* att_w_packed_bf16,
                                const void* att_wt_packed_bf16, float slope, const float* dout, float* dx, float* dw_att,
                                int32_t flags, double* G, void* ws, void* stream) {
  return lfa_bwd_impl(x, pos4, idx, n, K, CH, enc_w_folded, enc_b_folded, att_w_packed_bf16, att_wt_packed_bf16, slope,
                      dout, dx, dw_att, flags, G, ws, stream, true);
}
