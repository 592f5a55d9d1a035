// LDS-free fp32 MFMA GEMMs for the SharedMLP layers (gfx950): fragments are loaded straight from global memory
// in the v_mfma_f32_16x16x4_f32 operand layout, so there is no staging, no barrier and no exposed load latency
// between K chunks.
//
// Replaces the Linear of PyG's MLP / torch.nn.Linear (/root/reference/myria3d/models/modules/pyg_randla_net.py:
// 42,53,97-109) forward, dgrad and wgrad; same semantics as the LDS-tiled kernel in gemm.hip, which stays as the
// fallback for shapes these kernels do not cover.
//
// The network's GEMMs are tall and skinny (M = 204 800 ... 800 rows, K, N <= 768), i.e. HBM- or latency-bound, so
// the layouts are chosen for 16-byte global accesses rather than for MFMA reuse:
//   * the product is computed TRANSPOSED, D[n][m] = sum_k W[n][k] X[m][k]: MFMA A operand = weight tile,
//     B operand = 16 rows of X.  A lane then owns 4 CONSECUTIVE output columns of one row (C/D layout: row
//     (lane>>4)*4+r, col lane&15) -> one float4 store per 16x16 tile instead of four scattered dwords;
//   * within a 16-wide K chunk, MFMA step i of the chunk uses k = 16q + 4*(lane>>4) + i for BOTH operands (a
//     permutation of the reduction order), so each lane's four k values are one float4 of its row: one 16-byte
//     load per lane per chunk for X and for W.
//   * every load in a hot loop is an UNCONDITIONAL buffer load: out-of-range rows / columns / k use an out-of-bounds
//     offset, for which the hardware returns 0.  (Guarding loads with branches or selects makes the compiler sink
//     them into per-load basic blocks, each behind its own s_waitcnt vmcnt(0) — measured 3-5x slower on these
//     latency-bound shapes.)
// Kernels:
//   gemm_rowstream_kernel<NT,KQ,MODE,..>   K <= 64, per-wave column slice <= 64: the weight slice lives in registers
//                                 and the wave streams 16-row tiles (204 800-row layers: pure HBM streaming)
//   gemm_kloop_kernel<MTW,NTW,MODE,..>  any K: operands re-read from L1/L2 every chunk (deep layers: few rows, long
//                                 K); a wave owns MTW x NTW tiles so that a loaded fragment feeds several MFMAs
//   wgrad2_kernel<TN,TK>          dW[n][k] = sum_m dZ[m][n] X[m][k]: the reduction runs over the rows; every wave
//                                 owns one row split and stores its partial to a workspace that wgrad_reduce_kernel
//                                 sums (no LDS atomics, no same-address global atomics)
// Train-mode BatchNorm column statistics (sum, sum of squares of the raw output) are accumulated in fp64 per lane,
// then across lanes / waves (LDS); every workgroup stores its partial row and m3d_bn_finalize sums the rows.
#include <stdlib.h>
#include <vector>
#include "gemm_common.h"

// wavefronts per SIMD the register allocator must leave room for (2nd __launch_bounds__ argument in HIP).  hipcc sizes
// registers to the launch bounds only; the LFA backward gained 20-30 % from such caps.  1 = no cap:
// untuned knobs for same-box A/B runs (tools/build_variant.sh NAME gemm_direct.hip -DGEMM_RS_MINW=4 ...).
#ifndef GEMM_RS_MINW
#define GEMM_RS_MINW 1
#endif
#ifndef GEMM_KL_MINW
#define GEMM_KL_MINW 1
#endif
#ifndef WGRAD_MINW
#define WGRAD_MINW 1
#endif
// launch-geometry constants (round 3's M3D_* environment knobs, now compile time: the library reads no environment)
#ifndef GEMM_KL_MINWAVES
#define GEMM_KL_MINWAVES 1536
#endif
#ifndef GEMM_KSPLIT
#define GEMM_KSPLIT 1
#endif
#ifndef GEMM_KSPLIT_MINK
#define GEMM_KSPLIT_MINK 128
#endif
#ifndef GEMM_RS_CAP
#define GEMM_RS_CAP 768
#endif
#ifndef GEMM_RS_TILES
#define GEMM_RS_TILES 128
#endif
#ifndef GEMM_DISABLE
#define GEMM_DISABLE 0
#endif
#ifndef WGRAD_WAVES
#define WGRAD_WAVES 0
#endif
#ifndef WGRAD_VEC
#define WGRAD_VEC 1
#endif
#ifndef WGRAD_XCD
#define WGRAD_XCD 1  // batched weight gradients: the output tiles of one row slice on ONE XCD (see wgrad2_batch_kernel)
#endif
#ifndef WGRAD_BATCH_WAVES_BIG
#define WGRAD_BATCH_WAVES_BIG 4096
#endif
#ifndef WGRAD_BATCH_WAVES_SMALL
#define WGRAD_BATCH_WAVES_SMALL 8192
#endif
#include "../../include/m3d_hip.h"

// (GEMM_DBG: timing ablations of the statistics epilogue, never in the product build — bit 0: no flush at all (the
// accumulation is dead code then), bit 1: no atomics / stores at its end, bit 2: no per-element accumulation)
#ifndef GEMM_DBG
#define GEMM_DBG 0
#endif

__device__ __forceinline__ float f4(const float4& v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w)); }

// ------------------------------------------------------------------------------------------
// branch-free operand fragments: buffer loads with hardware range checking.  Every operand gets a raw buffer
// descriptor of 2 GiB; a lane whose row / column / k is out of range uses the byte offset OOB (= the descriptor
// size), for which the hardware returns 0 without touching memory.  No select, no branch, nothing the compiler can
// sink behind a condition: the loads of a tile issue back to back.
// ------------------------------------------------------------------------------------------
#define M3D_BUF_BYTES 0x80000000u
#define OOB 0x80000000u
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t mk_rsrc(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, M3D_BUF_BYTES, 0x00020000);
}
__device__ __forceinline__ float4 ld4(rsrc_t r, unsigned off) {
  f32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
  return make_float4(v[0], v[1], v[2], v[3]);
}
// (the b32 builtin returns the raw 32 bits as an integer: reinterpret, do not convert)
__device__ __forceinline__ float ld1(rsrc_t r, unsigned off) {
  return __uint_as_float((unsigned)__builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
}
// ---- bf16 activation storage (round 6, M3D_IO_BF16).  Offsets stay what the fp32 code computes — BYTE offsets of fp32
// elements, OOB = 2 GiB — and a bf16 operand is read at HALF the offset through a descriptor of 1 GiB: OOB >> 1 is then
// exactly its size, so the hardware range check keeps working without a select.  H = false: the fp32 loads above.
typedef int i32x2_t __attribute__((ext_vector_type(2)));
template <bool H>
__device__ __forceinline__ rsrc_t mk_rsrc_h(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, H ? (M3D_BUF_BYTES >> 1) : M3D_BUF_BYTES, 0x00020000);
}
template <bool H>
__device__ __forceinline__ float4 ld4h(rsrc_t r, unsigned off) {
  if constexpr (H) {
    const i32x2_t v = __builtin_amdgcn_raw_buffer_load_b64(r, off >> 1, 0, 0);
    return bf16x4_to_f32(make_uint2((unsigned)v[0], (unsigned)v[1]));
  } else {
    return ld4(r, off);
  }
}
template <bool H>
__device__ __forceinline__ float ld1h(rsrc_t r, unsigned off) {
  if constexpr (H) return bf16_to_f32((unsigned short)__builtin_amdgcn_raw_buffer_load_b16(r, off >> 1, 0, 0));
  else return ld1(r, off);
}
// activation layouts of a GEMM launch (template parameter IO = the M3D_IO_* bits >> 12): bit 0 = bf16 storage of every
// activation matrix (A0, A1, C, C1, the prologues' z / dz / y), bit 1 = ... except A0 (fp32: an incoming gradient that was
// accumulated with float atomics), bit 2 = ... except C / C1 (fp32: the logits)
#define IO_A0H(IO) ((((IO) & 1) != 0) && (((IO) & 2) == 0))
#define IO_SH(IO) (((IO) & 1) != 0)
#define IO_CH(IO) ((((IO) & 1) != 0) && (((IO) & 4) == 0))

struct ARow {
  unsigned o0;  // byte offset of the (gathered) row in A0, OOB when the row does not exist
  unsigned o1;  // byte offset of the row in A1, OOB without a second operand
};

__device__ __forceinline__ ARow a_row(const GemmArgs& g, int64_t m) {
  const bool ok = m < g.M;
  const int64_t mc = ok ? m : 0;
  int64_t rr = mc;
  if (g.a0_rows) rr = (int64_t)g.a0_rows[mc];
  ARow r;
  r.o0 = (ok && rr >= 0) ? (unsigned)(rr * g.lda0 * 4) : OOB;
  r.o1 = (ok && g.k1 > 0) ? (unsigned)(mc * g.lda1 * 4) : OOB;
  return r;
}

// elements k .. k+3 of the (concatenated) row, zeros past K / for missing rows.
// VEC: k0, k1 multiples of 4, 16-byte aligned rows -> the four elements are one float4 of A0 or of A1 (CAT: a second
// operand exists; the lane's offset is OOB in the operand that does not hold k).  !VEC: single operand, dword loads.
template <bool VEC, bool CAT, bool AH = false, bool SH = false>
__device__ __forceinline__ float4 a_frag(const GemmArgs& g, rsrc_t ra0, rsrc_t ra1, const ARow& r, int k, int K) {
  if (VEC) {
    const bool in0 = k < g.k0;
    const unsigned f0 = (in0 && r.o0 != OOB) ? r.o0 + 4u * (unsigned)k : OOB;
    float4 v = ld4h<AH>(ra0, f0);
    if (CAT) {
      const unsigned f1 = (!in0 && k < K && r.o1 != OOB) ? r.o1 + 4u * (unsigned)(k - g.k0) : OOB;
      const float4 u = ld4h<SH>(ra1, f1);
      v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
    }
    return v;
  } else {
    float t[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) t[j] = ld1h<AH>(ra0, (k + j < K && r.o0 != OOB) ? r.o0 + 4u * (unsigned)(k + j) : OOB);
    return make_float4(t[0], t[1], t[2], t[3]);
  }
}

// W[n][k .. k+3]   (BCM: element (n, k) at b[k*ldb + n]); zeros for n >= N or k >= K
template <bool VEC, bool BCM>
__device__ __forceinline__ float4 w_frag(const GemmArgs& g, rsrc_t rb, int n, int k, int K) {
  const bool nok = n < g.N;
  if (BCM) {
    float t[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
      t[j] = ld1(rb, (nok && k + j < K) ? 4u * (unsigned)((k + j) * (int)g.ldb + n) : OOB);
    return make_float4(t[0], t[1], t[2], t[3]);
  }
  const unsigned base = 4u * (unsigned)(n * (int)g.ldb + k);
  if (VEC) return ld4(rb, (nok && k < K) ? base : OOB);
  float t[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) t[j] = ld1(rb, (nok && k + j < K) ? base + 4u * j : OOB);
  return make_float4(t[0], t[1], t[2], t[3]);
}

// ------------------------------------------------------------------------------------------
// BatchNorm-backward A-prologue (GemmArgs::pro_*): per-column constants in LDS, [6][KP] =
// scale, shift, mean, invstd, s1/M, s2/M (zeros past K: padded k contribute dz = 0).
// ------------------------------------------------------------------------------------------
template <int KP>
__device__ __forceinline__ void pro_setup(const GemmArgs& g, float (&cf)[6][KP]) {
  const int K = g.k0;
  const double invM = 1.0 / (double)g.M;
  const int kmax = ((K + 31) & ~31) < KP ? ((K + 31) & ~31) : KP;
  const bool writer = blockIdx.x == 0 && blockIdx.y == 0;
  for (int k = threadIdx.x; k < kmax; k += 256) {
    float sc = 0.f, sh = 0.f, mu = 0.f, is = 0.f, m1 = 0.f, m2 = 0.f;
    if (k < K) {
      // every load of this column up front: its constants, the writer's old sink values, then all slot rows at once
      sc = g.pro_scale[k]; sh = g.pro_shift[k]; mu = g.pro_mean[k]; is = g.pro_invstd[k];
      const bool hb = writer && g.pro_acc && g.pro_dbeta, hg = writer && g.pro_acc && g.pro_dgamma;
      const float rb = (hb ? g.pro_dbeta : g.pro_scale)[k], rg = (hg ? g.pro_dgamma : g.pro_scale)[k];  // (branch-free)
      const float ob = hb ? rb : 0.f, og = hg ? rg : 0.f;
      double t[2];
      slot_sums<2>(g.pro_sums + k, g.pro_slots, 3 * (size_t)K, (size_t)K, t);
      m1 = (float)(t[0] * invM); m2 = (float)(t[1] * invM);
      if (writer) {  // dbeta = s1, dgamma = s2 (what bn_bwd_apply_kernel's block 0 writes in the unfused path)
        if (g.pro_dbeta) g.pro_dbeta[k] = ob + (float)t[0];
        if (g.pro_dgamma) g.pro_dgamma[k] = og + (float)t[1];
      }
    }
    cf[0][k] = sc; cf[1][k] = sh; cf[2][k] = mu; cf[3][k] = is; cf[4][k] = m1; cf[5][k] = m2;
  }
  __syncthreads();
}
__device__ __forceinline__ uint32_t pro_key(const GemmArgs& g) { return g.pro_drop.thr16 ? drop_key(g.pro_drop) : 0u; }

// dz[m][k .. k+3] from dy and z (same arithmetic as bn_bwd_apply_kernel); `store`: also write it to pro_dz
template <int KP, bool AH = false, bool SH = false>
__device__ __forceinline__ float4 a_frag_pro(const GemmArgs& g, rsrc_t ra0, rsrc_t rz, const ARow& r, int k,
                                             const float (&cf)[6][KP], bool store) {
  const unsigned f0 = (k < g.k0 && r.o0 != OOB) ? r.o0 + 4u * (unsigned)k : OOB;
  float4 gy = ld4h<AH>(ra0, f0);
  const float4 zv = ld4h<SH>(rz, f0);
  if (g.pro_drop.thr16) {  // (f0 / 16 = number of this float4 in the row-major [M, k0] output: lda0 == k0 is checked by the host)
    const int n4 = g.k0 >> 2;
    const int64_t i4 = f0 != OOB ? (int64_t)(f0 >> 4) : 0;  // (rows past the end: gy is 0 anyway, but rows[] must not be read there)
    const float4 m = drop_mul4(g.pro_dkey, drop_index(g.pro_drop, i4 / n4, (int)(i4 % n4), n4), g.pro_drop.thr16, g.pro_drop.scale);
    gy.x *= m.x; gy.y *= m.y; gy.z *= m.z; gy.w *= m.w;
  }
  const float4 sc = *(const float4*)&cf[0][k], mu = *(const float4*)&cf[2][k], is = *(const float4*)&cf[3][k];
  const float4 m1 = *(const float4*)&cf[4][k], m2 = *(const float4*)&cf[5][k];
  if (g.pro_act) {
    const float4 sh = *(const float4*)&cf[1][k];
    gy.x *= (zv.x * sc.x + sh.x) > 0.f ? 1.f : g.pro_slope; gy.y *= (zv.y * sc.y + sh.y) > 0.f ? 1.f : g.pro_slope;
    gy.z *= (zv.z * sc.z + sh.z) > 0.f ? 1.f : g.pro_slope; gy.w *= (zv.w * sc.w + sh.w) > 0.f ? 1.f : g.pro_slope;
  }
  float4 o;
  o.x = sc.x * (gy.x - m1.x - (zv.x - mu.x) * is.x * m2.x);
  o.y = sc.y * (gy.y - m1.y - (zv.y - mu.y) * is.y * m2.y);
  o.z = sc.z * (gy.z - m1.z - (zv.z - mu.z) * is.z * m2.z);
  o.w = sc.w * (gy.w - m1.w - (zv.w - mu.w) * is.w * m2.w);
  if (store && f0 != OOB) io_store4<SH>(g.pro_dz, f0 >> 2, o);
  return o;
}

// ------------------------------------------------------------------------------------------
// forward A-prologue (GemmArgs::fpro_*): scale / shift of the layer in front from its slot statistics -> LDS [2][64]
// (zeros past k0), same arithmetic as bn_stats_col (bn.hip): fp64 mean / biased variance / invstd, running statistics with
// the unbiased variance
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void fpro_setup(const GemmArgs& g, float (&cf)[2][64]) {
  const int K = g.k0;
  const bool writer = blockIdx.x == 0 && blockIdx.y == 0;
  for (int k = threadIdx.x; k < 64; k += 256) {
    float sc = 0.f, sh = 0.f;
    if (k < K) {
      // (absent operands read the slot table instead — a read-only address — and are dropped: a load under its own branch
      // would be waited for there; never through fpro_scale, which workgroup (0, 0) is writing: ADVICE r5)
      const float* dummy = (const float*)g.fpro_slots;
      const float gam0 = (g.fpro_gamma ? g.fpro_gamma : dummy)[k], bet0 = (g.fpro_beta ? g.fpro_beta : dummy)[k];
      const float rm = (writer && g.fpro_rmean ? g.fpro_rmean : dummy)[k];
      const float rv = (writer && g.fpro_rvar ? g.fpro_rvar : dummy)[k];
      const float gam = g.fpro_gamma ? gam0 : 1.f, bet = g.fpro_beta ? bet0 : 0.f;
      double sq[2];
      slot_sums<2>(g.fpro_slots + k, g.fpro_nslots, 2 * (size_t)K, (size_t)K, sq);
      const double mean = sq[0] / g.fpro_count;
      double var = sq[1] / g.fpro_count - mean * mean;
      if (var < 0.0) var = 0.0;
      const double invstd = 1.0 / sqrt(var + (double)g.fpro_eps);
      const double scd = (double)gam * invstd;
      sc = (float)scd;
      sh = (float)((double)bet - mean * scd);
      if (writer) {
        g.fpro_scale[k] = sc; g.fpro_shift[k] = sh;
        if (g.fpro_mean) g.fpro_mean[k] = (float)mean;
        if (g.fpro_invstd) g.fpro_invstd[k] = (float)invstd;
        if (g.fpro_rmean) g.fpro_rmean[k] = (float)((1.0 - g.fpro_momentum) * (double)rm + g.fpro_momentum * mean);
        if (g.fpro_rvar) {
          const double unbiased = g.fpro_count > 1.0 ? var * g.fpro_count / (g.fpro_count - 1.0) : var;
          g.fpro_rvar[k] = (float)((1.0 - g.fpro_momentum) * (double)rv + g.fpro_momentum * unbiased);
        }
      }
    }
    cf[0][k] = sc; cf[1][k] = sh;
  }
  __syncthreads();
}
// y[m][k .. k+3] = lrelu(z * scale + shift) of the (never gathered) row; `store`: also write it to fpro_y (same layout as z)
template <bool SH = false>
__device__ __forceinline__ float4 a_frag_fpro(const GemmArgs& g, rsrc_t ra0, const ARow& r, int k, const float (&cf)[2][64],
                                              bool store) {
  const unsigned f0 = (k < g.k0 && r.o0 != OOB) ? r.o0 + 4u * (unsigned)k : OOB;
  const float4 z = ld4h<SH>(ra0, f0);
  const float4 sc = *(const float4*)&cf[0][k], sh = *(const float4*)&cf[1][k];
  float4 y = make_float4(z.x * sc.x + sh.x, z.y * sc.y + sh.y, z.z * sc.z + sh.z, z.w * sc.w + sh.w);
  if (g.fpro_act) { y.x = lrelu(y.x, g.fpro_slope); y.y = lrelu(y.y, g.fpro_slope); y.z = lrelu(y.z, g.fpro_slope); y.w = lrelu(y.w, g.fpro_slope); }
  const bool live = f0 != OOB;  // (rows / columns that do not exist must stay 0: lrelu(shift) is not)
  y.x = live ? y.x : 0.f; y.y = live ? y.y : 0.f; y.z = live ? y.z : 0.f; y.w = live ? y.w : 0.f;
  if (store && live) io_store4<SH>(g.fpro_y, f0 >> 2, y);
  return y;
}

// accumulate, pre-loaded (GemmArgs::acc_pre): the lane's four old output values of row m, columns n0..n0+3 (the C/D
// fragment layout of the transposed product) as the initial accumulator; zeros where the tile has no output
template <bool CHh = false>
__device__ __forceinline__ f32x4 c_prev(const GemmArgs& g, rsrc_t rc, rsrc_t rc1, int64_t m, int n0) {
  const bool ok = m < g.M && n0 < g.N;
  auto ldc4 = [](rsrc_t r, unsigned off) -> f32x4 {
    if constexpr (CHh) { const float4 t = ld4h<true>(r, off); return (f32x4){t.x, t.y, t.z, t.w}; }
    else return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
  };
  if (g.c_split > 0) {
    const bool lo = n0 < g.c_split;
    const f32x4 u = ldc4(rc, (ok && lo) ? (unsigned)((m * g.ldc + n0) * 4) : OOB);
    const f32x4 v = ldc4(rc1, (ok && !lo) ? (unsigned)((m * g.ldc1 + (n0 - g.c_split)) * 4) : OOB);
    return u + v;
  }
  return ldc4(rc, ok ? (unsigned)((m * g.ldc + n0) * 4) : OOB);
}

// Epilogue modes (template parameter MODE): the network never needs statistics and an affine map in one launch
//   0 PLAIN   + bias                      (dgrad, plain Linear)
//   1 STATS   + bias, column sum/sumsq    (train-mode SharedMLP: BatchNorm statistics of the raw output)
//   2 AFFINE  + bias, *scale + shift, act (eval-mode SharedMLP: folded BatchNorm + LeakyReLU)
template <int MODE>
struct Epi {
  float4 bia, sc, sh;
};

template <int MODE>
__device__ __forceinline__ Epi<MODE> epi_load(const GemmArgs& g, int n0) {
  // buffer loads: a null bias / scale / shift pointer simply reads as OOB (= 0) — no branches, no serialised waits
  const rsrc_t rbias = mk_rsrc(g.bias), rsc = mk_rsrc(g.scale), rsh = mk_rsrc(g.shift);
  Epi<MODE> e;
  float b[4], s[4], h[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int n = n0 + r;
    const bool ok = n < g.N;
    b[r] = ld1(rbias, (ok && g.bias) ? 4u * (unsigned)n : OOB);
    s[r] = 1.f; h[r] = 0.f;
    if (MODE == 2) {
      const float sv = ld1(rsc, (ok && g.scale) ? 4u * (unsigned)n : OOB);
      s[r] = (ok && g.scale) ? sv : 1.f;
      h[r] = ld1(rsh, (ok && g.shift) ? 4u * (unsigned)n : OOB);
    }
  }
  e.bia = make_float4(b[0], b[1], b[2], b[3]);
  e.sc = make_float4(s[0], s[1], s[2], s[3]);
  e.sh = make_float4(h[0], h[1], h[2], h[3]);
  return e;
}

// finish one 16x16 tile owned by this lane: columns n0..n0+3 of row m
template <int MODE, bool CHh = false>
__device__ __forceinline__ void epi_store(const GemmArgs& g, const Epi<MODE>& e, f32x4 acc, int64_t m, int n0, bool cvec,
                                          double (&ssum)[4], double (&ssq)[4]) {
  const bool rowok = m < g.M;
  float y[4];
  // affine epilogue with a RESIDUAL (MODE 2 + accumulate, round 6): C = act(scale * (A B^T + bias) + shift + C_old) — the tail of
  // a DilatedResidualBlock in eval mode, LeakyReLU(BN(mlp2(x)) + BN(shortcut(x))) (pyg_randla_net.py:186-187), as the epilogue of
  // the mlp2 GEMM over the buffer the shortcut GEMM (affine epilogue, no activation) has just written: no bn_apply launch
  float res[4] = {0.f, 0.f, 0.f, 0.f};
  if (MODE == 2 && g.accumulate && rowok) {
    const size_t re = (size_t)(m * g.ldc + n0);
    if (cvec) {
      if (n0 < g.N) { const float4 p = io_load4<CHh>(g.c, re); res[0] = p.x; res[1] = p.y; res[2] = p.z; res[3] = p.w; }
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) res[r] = n0 + r < g.N ? io_load1<CHh>(g.c, re + r) : 0.f;
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float z = acc[r] + f4(e.bia, r);
    if (MODE == 1 && !(GEMM_DBG & 4)) {
      // fp64: columns with |mean| >> std (tiny batches at the deepest level) lose the variance otherwise
      const double zd = (rowok && n0 + r < g.N) ? (double)z : 0.0;
      ssum[r] += zd;
      ssq[r] += zd * zd;
    }
    float v = z;
    if (MODE == 2) {
      v = z * f4(e.sc, r) + f4(e.sh, r) + res[r];
      if (g.act) v = lrelu(v, g.slope);
    }
    y[r] = v;
  }
  if (!rowok) return;
  // accumulate (plain epilogue only): C += A B^T — every element has exactly one writer (no split-K here), so a plain
  // read-modify-write.  Used by the backward pass to add an input gradient into the buffer another consumer of the same
  // tensor has already written (no separate elementwise add, no zero fill)
  const bool acc_c = MODE == 0 && g.accumulate && !g.acc_pre;
  // (CHh: C / C1 hold bf16 — element offsets, io_load / io_store: m3d_common.h)
  if (MODE == 0 && g.c_split > 0) {  // (c_split, N multiples of 4: a lane's four columns fall on one side)
    void* dst = nullptr;
    size_t eo = 0;
    if (n0 < g.c_split) { dst = g.c; eo = (size_t)(m * g.ldc + n0); }
    else if (n0 < g.N) { dst = g.c1; eo = (size_t)(m * g.ldc1 + (n0 - g.c_split)); }
    if (dst) {
      float4 o = make_float4(y[0], y[1], y[2], y[3]);
      if (acc_c) { const float4 p = io_load4<CHh>(dst, eo); o.x += p.x; o.y += p.y; o.z += p.z; o.w += p.w; }
      io_store4<CHh>(dst, eo, o);
    }
    return;
  }
  const size_t ce = (size_t)(m * g.ldc + n0);
  if (cvec) {  // N % 4 == 0, ldc % 4 == 0, 16-byte aligned C
    if (n0 < g.N) {
      float4 o = make_float4(y[0], y[1], y[2], y[3]);
      if (acc_c) { const float4 p = io_load4<CHh>(g.c, ce); o.x += p.x; o.y += p.y; o.z += p.z; o.w += p.w; }
      io_store4<CHh>(g.c, ce, o);
    }
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (n0 + r < g.N) io_store1<CHh>(g.c, ce + r, acc_c ? io_load1<CHh>(g.c, ce + r) + y[r] : y[r]);
  }
}

// per-lane fp64 column partials -> over the 16 row lanes -> LDS over the workgroup's waves -> this workgroup's
// partial row of stat_part.  Lane (lr, lg) holds columns nb + 16t + 4lg + r.
// Cross-lane part: row16_sum_d (m3d_common.h) — exact fp64 sums over the 16 row lanes with DPP moves.  (__shfl_xor on a
// double is two ds_bpermute_b32: 256 LDS-pipe round trips per wave in this flush, at the END of every workgroup, every
// workgroup of a CU at once — the statistics epilogue cost 156 us of the 522 us the forward GEMMs of a step take, almost all
// of it here: profiles/r03y_gemm_stats_epilogue.log.  With DPP moves: 440 us.  Accumulating in fp32 around a shared
// per-column shift and reducing in fp32 was 420 us but gives up what the fp64 partials have: they are EXACT sums of fp32
// values, so the fp64 atomics that combine workgroups commute and the train-mode forward is bit-reproducible.)
template <int NT, int NWAVES>
__device__ __forceinline__ void stats_flush(const GemmArgs& g, int nb, double (&ssum)[NT][4], double (&ssq)[NT][4],
                                            const unsigned bx) {
  if (GEMM_DBG & 1) return;
  __shared__ double sred[NWAVES][2][16 * NT];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, lr = lane & 15, lg = lane >> 4;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const double a = row16_sum_d(ssum[t][r]), q = row16_sum_d(ssq[t][r]);
      if (lr == 0) { sred[wid][0][t * 16 + lg * 4 + r] = a; sred[wid][1][t * 16 + lg * 4 + r] = q; }
    }
  __syncthreads();
  if (threadIdx.x < 2 * 16 * NT) {
    const int which = threadIdx.x / (16 * NT), col = threadIdx.x % (16 * NT);
    const int n = nb + col;
    if (n < g.N) {
      double v = 0.0;
#pragma unroll
      for (int w = 0; w < NWAVES; ++w) v += sred[w][which][col];
      if (GEMM_DBG & 2) return;
      if (g.stat_slots > 0) {
        // slot mode: a few fp64 atomics per address (workgroups / slots); the table was zeroed by the caller
        atomicAdd(&g.stat_part[((size_t)(bx % g.stat_slots) * 2 + which) * g.N + n], v);
      } else {
        // plain store of this workgroup's partial (summed by m3d_bn_finalize): no same-address atomics, no
        // zero-fill, bitwise reproducible
        g.stat_part[((size_t)bx * 2 + which) * g.N + n] = v;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// K <= 16*KQ <= 64, column slice 16*NT <= 64: weights in registers, rows streamed.
// grid: (row workgroups, column slices); 4 waves per workgroup take interleaved 16-row tiles.
// ------------------------------------------------------------------------------------------
template <int NT, int KQ, int MODE, bool VEC, bool CAT, bool BCM, bool PRO = false, bool FPRO = false, int IO = 0>
__global__ __launch_bounds__(256, GEMM_RS_MINW) void gemm_rowstream_kernel(GemmArgs g, int cvec) {
  constexpr bool AH = IO_A0H(IO), SH = IO_SH(IO), CHh = IO_CH(IO);
  if constexpr (PRO) g.pro_dkey = pro_key(g);
  __shared__ float fcf[FPRO ? 2 : 1][FPRO ? 64 : 4];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, lr = lane & 15, lg = lane >> 4;
  const int K = g.k0 + g.k1;
  const int nb = blockIdx.y * 16 * NT;
  const rsrc_t ra0 = mk_rsrc_h<AH>(g.a0), ra1 = mk_rsrc_h<SH>(g.a1), rb = mk_rsrc(g.b), rc = mk_rsrc_h<CHh>(g.c), rc1 = mk_rsrc_h<CHh>(g.c1);
  __shared__ float cf[PRO ? 6 : 1][PRO ? 64 : 4];
  const rsrc_t rz = mk_rsrc_h<SH>(PRO ? g.pro_z : nullptr);
  float4 w[NT][KQ];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int q = 0; q < KQ; ++q) w[t][q] = w_frag<VEC, BCM>(g, rb, nb + 16 * t + lr, 16 * q + 4 * lg, K);
  Epi<MODE> e[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) e[t] = epi_load<MODE>(g, nb + 16 * t + 4 * lg);
  // (after the weight / bias loads were issued: the column pass is a dependent chain of its own — slot loads, fp64, LDS,
  // barrier — and the small layers are nothing but latency)
  if constexpr (PRO) pro_setup<64>(g, (float (&)[6][64])cf);
  if constexpr (FPRO) fpro_setup(g, (float (&)[2][64])fcf);
  double ssum[NT][4], ssq[NT][4];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) { ssum[t][r] = 0.0; ssq[t][r] = 0.0; }

  const int64_t ntiles = (g.M + 15) >> 4;
  const int64_t stride = (int64_t)gridDim.x * 4;
  for (int64_t tile = (int64_t)blockIdx.x * 4 + wid; tile < ntiles; tile += stride) {
    const int64_t m = tile * 16 + lr;
    const ARow row = a_row(g, m);
    float4 a[KQ];
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
      if constexpr (PRO) a[q] = a_frag_pro<64, AH, SH>(g, ra0, rz, row, 16 * q + 4 * lg, (const float (&)[6][64])cf, blockIdx.y == 0);
      else if constexpr (FPRO) a[q] = a_frag_fpro<SH>(g, ra0, row, 16 * q + 4 * lg, (const float (&)[2][64])fcf, blockIdx.y == 0 && g.fpro_y);
      else a[q] = a_frag<VEC, CAT, AH, SH>(g, ra0, ra1, row, 16 * q + 4 * lg, K);
    }
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (MODE == 0 && g.acc_pre) acc[t] = c_prev<CHh>(g, rc, rc1, m, nb + 16 * t + 4 * lg);
    }
#pragma unroll
    for (int q = 0; q < KQ; ++q)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = mfma16(f4(w[t][q], i), f4(a[q], i), acc[t]);
#pragma unroll
    for (int t = 0; t < NT; ++t) epi_store<MODE, CHh>(g, e[t], acc[t], m, nb + 16 * t + 4 * lg, cvec, ssum[t], ssq[t]);
  }
  if (MODE == 1) stats_flush<NT, 4>(g, nb, ssum, ssq, blockIdx.x);
}

// ------------------------------------------------------------------------------------------
// any K: weights streamed from L1/L2 chunk by chunk.  grid: (row workgroups, column slices of 16*NTW)
// ------------------------------------------------------------------------------------------
// BF: both operands are rounded to bf16 as their fragments are assembled (8 consecutive k per lane = two 16-byte loads)
// and the product runs on v_mfma_f32_16x16x32_bf16 with fp32 accumulation (K a multiple of 32): the deep SharedMLP
// layers in the net's "bf16" matmul precision.  Epilogue, statistics and storage stay fp32.
#define PRO_KMAX 1024  // widest BatchNorm the prologue keeps in LDS (this network: 512)
// (bx, gx): this workgroup's index among the gx row workgroups of ITS problem — the launch's own blockIdx.x / gridDim.x, or
// a sub-range of them when one launch carries two problems (gemm_kloop_pair_kernel)
template <int MTW, int NTW, int MODE, bool VEC, bool CAT, bool BCM, bool BF, bool PRO, int IO = 0>
__device__ __forceinline__ void gemm_kloop_body(const GemmArgs& g, int cvec, const unsigned bx, const unsigned gx) {
  constexpr bool AH = IO_A0H(IO), SH = IO_SH(IO), CHh = IO_CH(IO);
  // a wave owns MTW x NTW tiles of 16x16: every A / W fragment it loads feeds NTW / MTW MFMAs (these shapes are
  // L2-bandwidth bound on operand re-reads when MTW = NTW = 1)
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, lr = lane & 15, lg = lane >> 4;
  const int K = g.k0 + g.k1;
  const int KQ = (K + 15) >> 4;
  const int nb = blockIdx.y * 16 * NTW;
  const rsrc_t ra0 = mk_rsrc_h<AH>(g.a0), ra1 = mk_rsrc_h<SH>(g.a1), rb = mk_rsrc(g.b), rc = mk_rsrc_h<CHh>(g.c), rc1 = mk_rsrc_h<CHh>(g.c1);
  __shared__ float cf[PRO ? 6 : 1][PRO ? PRO_KMAX : 4];
  const rsrc_t rz = mk_rsrc_h<SH>(PRO ? g.pro_z : nullptr);
  const bool pst = blockIdx.y == 0;  // column slice 0 stores dz
  auto afr = [&](const ARow& r, int k) -> float4 {
    if constexpr (PRO) return a_frag_pro<PRO_KMAX, AH, SH>(g, ra0, rz, r, k, (const float (&)[6][PRO_KMAX])cf, pst);
    else return a_frag<VEC, CAT, AH, SH>(g, ra0, ra1, r, k, K);
  };
  Epi<MODE> e[NTW];
#pragma unroll
  for (int t = 0; t < NTW; ++t) e[t] = epi_load<MODE>(g, nb + 16 * t + 4 * lg);
  if constexpr (PRO) pro_setup<PRO_KMAX>(g, (float (&)[6][PRO_KMAX])cf);
  double ssum[NTW][4], ssq[NTW][4];
#pragma unroll
  for (int t = 0; t < NTW; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) { ssum[t][r] = 0.0; ssq[t][r] = 0.0; }

  const int64_t ngroups = (g.M + 16 * MTW - 1) / (16 * MTW);
  const bool ks = g.ksplit != 0;  // (uniform) split K over the four waves: same group, a quarter of the chunks each
  __shared__ float kred[3][MTW * NTW * 4 * 64];
  const int64_t stride = ks ? (int64_t)gx : (int64_t)gx * 4;
  for (int64_t grp = ks ? (int64_t)bx : (int64_t)bx * 4 + wid; grp < ngroups; grp += stride) {
    ARow row[MTW];
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt) row[mt] = a_row(g, (grp * MTW + mt) * 16 + lr);
    f32x4 acc[MTW][NTW];
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
      for (int t = 0; t < NTW; ++t) {
        acc[mt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // (split K: wave 0 owns the epilogue, so only its accumulators start from the old tile)
        if (MODE == 0 && g.acc_pre && (!ks || wid == 0))
          acc[mt][t] = c_prev<CHh>(g, rc, rc1, (grp * MTW + mt) * 16 + lr, nb + 16 * t + 4 * lg);
      }
    if constexpr (BF) {
      const int nq = K >> 5, per = (nq + 3) >> 2;
      const int q0 = ks ? wid * per : 0, q1 = ks ? (q0 + per < nq ? q0 + per : nq) : nq;
#pragma unroll 2
      for (int q = q0; q < q1; ++q) {
        const int k = 32 * q + 8 * lg;
        Bf16Frag a[MTW], w[NTW];
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt) {
          const float4 lo = afr(row[mt], k), hi = afr(row[mt], k + 4);
          a[mt].u[0] = pack_bf16(lo.x, lo.y); a[mt].u[1] = pack_bf16(lo.z, lo.w);
          a[mt].u[2] = pack_bf16(hi.x, hi.y); a[mt].u[3] = pack_bf16(hi.z, hi.w);
        }
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
          const float4 lo = w_frag<VEC, BCM>(g, rb, nb + 16 * t + lr, k, K), hi = w_frag<VEC, BCM>(g, rb, nb + 16 * t + lr, k + 4, K);
          w[t].u[0] = pack_bf16(lo.x, lo.y); w[t].u[1] = pack_bf16(lo.z, lo.w);
          w[t].u[2] = pack_bf16(hi.x, hi.y); w[t].u[3] = pack_bf16(hi.z, hi.w);
        }
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
          for (int t = 0; t < NTW; ++t) acc[mt][t] = mfma_bf16(w[t].v, a[mt].v, acc[mt][t]);
      }
    } else {
    const int per = (KQ + 3) >> 2;
    const int q0 = ks ? wid * per : 0, q1 = ks ? (q0 + per < KQ ? q0 + per : KQ) : KQ;
#pragma unroll 4
    for (int q = q0; q < q1; ++q) {
      const int k = 16 * q + 4 * lg;
      float4 a[MTW], w[NTW];
#pragma unroll
      for (int mt = 0; mt < MTW; ++mt) a[mt] = afr(row[mt], k);
#pragma unroll
      for (int t = 0; t < NTW; ++t) w[t] = w_frag<VEC, BCM>(g, rb, nb + 16 * t + lr, k, K);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
          for (int t = 0; t < NTW; ++t) acc[mt][t] = mfma16(f4(w[t], i), f4(a[mt], i), acc[mt][t]);
    }
    }
    if (ks) {  // the partial products of waves 1..3 meet wave 0's in LDS (same lane layout)
      if (wid > 0) {
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
          for (int t = 0; t < NTW; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) kred[wid - 1][((mt * NTW + t) * 4 + r) * 64 + lane] = acc[mt][t][r];
      }
      __syncthreads();
      if (wid == 0) {
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
          for (int t = 0; t < NTW; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int o = ((mt * NTW + t) * 4 + r) * 64 + lane;
              acc[mt][t][r] += (kred[0][o] + kred[1][o]) + kred[2][o];
            }
      }
    }
    if (!ks || wid == 0) {
#pragma unroll
      for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
        for (int t = 0; t < NTW; ++t)
          epi_store<MODE, CHh>(g, e[t], acc[mt][t], (grp * MTW + mt) * 16 + lr, nb + 16 * t + 4 * lg, cvec, ssum[t], ssq[t]);
    }
    if (ks) __syncthreads();  // kred is reused by the next group
  }
  if (MODE == 1) stats_flush<NTW, 4>(g, nb, ssum, ssq, bx);
}

template <int MTW, int NTW, int MODE, bool VEC, bool CAT, bool BCM, bool BF = false, bool PRO = false, int IO = 0>
__global__ __launch_bounds__(256, GEMM_KL_MINW) void gemm_kloop_kernel(GemmArgs g, int cvec) {
  if constexpr (PRO) g.pro_dkey = pro_key(g);
  gemm_kloop_body<MTW, NTW, MODE, VEC, CAT, BCM, BF, PRO, IO>(g, cvec, blockIdx.x, gridDim.x);
}

// TWO independent products with the same output shape and tile plan in one launch: row workgroups [0, wgs0) work on g0,
// the rest on g1 (same column slices).  The mlp2 and shortcut Linears of a DilatedResidualBlock (pyg_randla_net.py:172-188)
// — forward, and their input gradients — on the deep levels: each alone is 400-800 workgroups of a few microseconds.
template <int MTW, int NTW, int MODE, bool BCM, bool BF, int IO = 0>
__global__ __launch_bounds__(256, GEMM_KL_MINW) void gemm_kloop_pair_kernel(GemmArgs g0, GemmArgs g1, int cvec0, int cvec1,
                                                                            unsigned wgs0) {
  if (blockIdx.x < wgs0) gemm_kloop_body<MTW, NTW, MODE, true, false, BCM, BF, false, IO>(g0, cvec0, blockIdx.x, wgs0);
  else gemm_kloop_body<MTW, NTW, MODE, true, false, BCM, BF, false, IO>(g1, cvec1, blockIdx.x - wgs0, gridDim.x - wgs0);
}

// ------------------------------------------------------------------------------------------
// dispatch of the forward / dgrad kernels
// ------------------------------------------------------------------------------------------
// variants: 0 vec, 1 vec + concatenated A (forward of the FP modules), 2 vec + column-major W (dgrad), 3 scalar loads,
// 4 scalar loads + column-major W
// IO (template): the activation layouts of the launch (M3D_IO_* >> 12, see IO_A0H / IO_SH / IO_CH): 0 = fp32, 1 = bf16
// storage, 3 = bf16 storage with an fp32 A0 (the BatchNorm-backward prologue only: an incoming gradient accumulated with
// float atomics), 5 = bf16 storage with an fp32 C (vector-load forward, plain epilogue only: the logits).  false: this
// combination is not instantiated.
template <int NT, int KQ, int MODE, int IO>
static bool launch_rs(const GemmArgs& g, int variant, dim3 grid, hipStream_t st, int cvec) {
  if constexpr (IO == 3) {
    if (variant != 2 || !g.pro_z || MODE != 0) return false;
    hipLaunchKernelGGL((gemm_rowstream_kernel<NT, KQ, 0, true, false, true, true, false, 3>), grid, dim3(256), 0, st, g, cvec);
    return true;
  } else if constexpr (IO == 5) {
    if (variant != 0 || MODE != 0 || g.fpro) return false;
    hipLaunchKernelGGL((gemm_rowstream_kernel<NT, KQ, 0, true, false, false, false, false, 5>), grid, dim3(256), 0, st, g, cvec);
    return true;
  } else {
    switch (variant) {
      case 0:
        if (MODE == 1 && g.fpro) hipLaunchKernelGGL((gemm_rowstream_kernel<NT, KQ, 1, true, false, false, false, true, IO>), grid, dim3(256), 0, st, g, cvec);
        else hipLaunchKernelGGL((gemm_rowstream_kernel<NT, KQ, MODE, true, false, false, false, false, IO>), grid, dim3(256), 0, st, g, cvec);
        break;
      case 1: hipLaunchKernelGGL((gemm_rowstream_kernel<NT, KQ, MODE, true, true, false, false, false, IO>), grid, dim3(256), 0, st, g, cvec); break;
      case 2:
        if (g.pro_z) hipLaunchKernelGGL((gemm_rowstream_kernel<NT, KQ, 0, true, false, true, true, false, IO>), grid, dim3(256), 0, st, g, cvec);
        else hipLaunchKernelGGL((gemm_rowstream_kernel<NT, KQ, 0, true, false, true, false, false, IO>), grid, dim3(256), 0, st, g, cvec);
        break;
      case 3: hipLaunchKernelGGL((gemm_rowstream_kernel<NT, KQ, MODE, false, false, false, false, false, IO>), grid, dim3(256), 0, st, g, cvec); break;
      default: hipLaunchKernelGGL((gemm_rowstream_kernel<NT, KQ, 0, false, false, true, false, false, IO>), grid, dim3(256), 0, st, g, cvec); break;
    }
    return true;
  }
}

template <int NT, int KQ, int IO>
static bool launch_rs_mode(const GemmArgs& g, int mode, int variant, dim3 grid, hipStream_t st, int cvec) {
  if (mode == 1) return launch_rs<NT, KQ, 1, IO>(g, variant, grid, st, cvec);
  if (mode == 2) return launch_rs<NT, KQ, 2, IO>(g, variant, grid, st, cvec);
  return launch_rs<NT, KQ, 0, IO>(g, variant, grid, st, cvec);
}

template <int NT, int KQ>
static bool launch_rs_io(const GemmArgs& g, int mode, int variant, dim3 grid, hipStream_t st, int cvec) {
  switch (g.io) {
    case 0: return launch_rs_mode<NT, KQ, 0>(g, mode, variant, grid, st, cvec);
    case 1: return launch_rs_mode<NT, KQ, 1>(g, mode, variant, grid, st, cvec);
    case 3: return mode == 0 && launch_rs<NT, KQ, 0, 3>(g, variant, grid, st, cvec);
    case 5: return mode == 0 && launch_rs<NT, KQ, 0, 5>(g, variant, grid, st, cvec);
    default: return false;
  }
}

template <int NT>
static bool launch_rowstream(const GemmArgs& g, int mode, int KQ, int variant, dim3 grid, hipStream_t st, int cvec) {
  if (KQ == 1) return launch_rs_io<NT, 1>(g, mode, variant, grid, st, cvec);
  if (KQ == 2) return launch_rs_io<NT, 2>(g, mode, variant, grid, st, cvec);
  return launch_rs_io<NT, 4>(g, mode, variant, grid, st, cvec);
}

template <int MTW, int NTW, int MODE, int IO>
static bool launch_kl(const GemmArgs& g, int variant, dim3 grid, hipStream_t st, int cvec) {
  const bool bf = g.bf16 && variant <= 2 && ((g.k0 + g.k1) & 31) == 0;  // bf16 matrix cores (vector-load variants only)
  if constexpr (IO == 3) {
    if (variant != 2 || !g.pro_z || MODE != 0) return false;
    if (bf) hipLaunchKernelGGL((gemm_kloop_kernel<MTW, NTW, 0, true, false, true, true, true, 3>), grid, dim3(256), 0, st, g, cvec);
    else hipLaunchKernelGGL((gemm_kloop_kernel<MTW, NTW, 0, true, false, true, false, true, 3>), grid, dim3(256), 0, st, g, cvec);
    return true;
  } else {
    if (IO != 0 && variant > 2) return false;  // (the scalar-load k-loop variants exist for fp32 storage only: odd K > 64)
    if (bf) {
      switch (variant) {
        case 0: hipLaunchKernelGGL((gemm_kloop_kernel<MTW, NTW, MODE, true, false, false, true, false, IO>), grid, dim3(256), 0, st, g, cvec); break;
        case 1: hipLaunchKernelGGL((gemm_kloop_kernel<MTW, NTW, MODE, true, true, false, true, false, IO>), grid, dim3(256), 0, st, g, cvec); break;
        default:
          if (g.pro_z) hipLaunchKernelGGL((gemm_kloop_kernel<MTW, NTW, 0, true, false, true, true, true, IO>), grid, dim3(256), 0, st, g, cvec);
          else hipLaunchKernelGGL((gemm_kloop_kernel<MTW, NTW, 0, true, false, true, true, false, IO>), grid, dim3(256), 0, st, g, cvec);
          break;
      }
      return true;
    }
    switch (variant) {
      case 0: hipLaunchKernelGGL((gemm_kloop_kernel<MTW, NTW, MODE, true, false, false, false, false, IO>), grid, dim3(256), 0, st, g, cvec); break;
      case 1: hipLaunchKernelGGL((gemm_kloop_kernel<MTW, NTW, MODE, true, true, false, false, false, IO>), grid, dim3(256), 0, st, g, cvec); break;
      case 2:
        if (g.pro_z) hipLaunchKernelGGL((gemm_kloop_kernel<MTW, NTW, 0, true, false, true, false, true, IO>), grid, dim3(256), 0, st, g, cvec);
        else hipLaunchKernelGGL((gemm_kloop_kernel<MTW, NTW, 0, true, false, true, false, false, IO>), grid, dim3(256), 0, st, g, cvec);
        break;
      case 3:
        if constexpr (IO == 0) hipLaunchKernelGGL((gemm_kloop_kernel<MTW, NTW, MODE, false, false, false>), grid, dim3(256), 0, st, g, cvec);
        break;
      default:
        if constexpr (IO == 0) hipLaunchKernelGGL((gemm_kloop_kernel<MTW, NTW, 0, false, false, true>), grid, dim3(256), 0, st, g, cvec);
        break;
    }
    return true;
  }
}

template <int MTW, int NTW>
static bool launch_kloop(const GemmArgs& g, int mode, int variant, dim3 grid, hipStream_t st, int cvec) {
  if (g.io == 3) return mode == 0 && launch_kl<MTW, NTW, 0, 3>(g, variant, grid, st, cvec);
  if (g.io == 1) {
    if (mode == 1) return launch_kl<MTW, NTW, 1, 1>(g, variant, grid, st, cvec);
    if (mode == 2) return launch_kl<MTW, NTW, 2, 1>(g, variant, grid, st, cvec);
    return launch_kl<MTW, NTW, 0, 1>(g, variant, grid, st, cvec);
  }
  if (g.io != 0) return false;
  if (mode == 1) return launch_kl<MTW, NTW, 1, 0>(g, variant, grid, st, cvec);
  if (mode == 2) return launch_kl<MTW, NTW, 2, 0>(g, variant, grid, st, cvec);
  return launch_kl<MTW, NTW, 0, 0>(g, variant, grid, st, cvec);
}

static inline bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// launch geometry of the forward / dgrad kernels (shared with m3d_gemm_direct_stat_parts)
struct RowPlan { int rowstream, NT, MT, KQ, ksplit; int64_t wgs, slices; };
static RowPlan plan_rows(int64_t M, int N, int K, int mode) {
  RowPlan p;
  const int64_t ntiles = m3d_cdiv(M, 16);
  const int ncol16 = (int)m3d_cdiv(N, 16);
  p.rowstream = K <= 64;
  p.ksplit = 0;
  p.KQ = K <= 16 ? 1 : (K <= 32 ? 2 : 4);
  if (p.rowstream) {
    int NT = ncol16 >= 4 ? 4 : (ncol16 >= 2 ? 2 : 1);
    while (NT > 1 && NT * p.KQ > 8) NT >>= 1;          // <= 8 weight float4 (32 VGPRs) per lane
    if (mode != 0 && NT == 4 && p.KQ >= 2) NT = 2;      // statistics / affine registers on top: stay <= 128 VGPRs
    p.NT = NT;
  } else {
    // K > 64: the largest per-wave tile (MT x NT blocks of 16x16) that still leaves >= ~768 waves for 1024 SIMDs
    // (4 x 4 tiles were measured slower: too few waves to hide the fragment-load latency; 1 536 instead of 768 waves:
    // 4.85 -> 4.80 ms per step)
    static const int cand[4][2] = {{2, 4}, {2, 2}, {1, 2}, {1, 1}};
    const int64_t kl_min_waves = GEMM_KL_MINWAVES;
    // split K over the four waves of a workgroup (GemmArgs::ksplit) when K is long: a (row group, column slice) pair
    // then counts as four waves, so the big wave tiles stay affordable on the deep levels.  GEMM_KSPLIT=0: never
    const bool can_split = GEMM_KSPLIT != 0 && K >= GEMM_KSPLIT_MINK;
    p.MT = 1; p.NT = 1; p.ksplit = 0;
    for (int c = 0; c < 4; ++c) {
      const int64_t waves = m3d_cdiv(ntiles, cand[c][0]) * m3d_cdiv(ncol16, cand[c][1]);
      if (waves >= kl_min_waves || c == 3) { p.MT = cand[c][0]; p.NT = cand[c][1]; break; }
      if (can_split && waves * 4 >= kl_min_waves) { p.MT = cand[c][0]; p.NT = cand[c][1]; p.ksplit = 1; break; }
    }
    p.slices = m3d_cdiv(ncol16, p.NT);
    const int64_t ngroups = m3d_cdiv(ntiles, p.MT);
    int64_t wgs = p.ksplit ? ngroups : m3d_cdiv(ngroups, 4);
    int64_t cap = 2048 / p.slices;  // statistics partial rows = wgs: keep them bounded
    if (cap < 1) cap = 1;
    if (wgs > cap) wgs = cap;
    p.wgs = wgs < 1 ? 1 : wgs;
    return p;
  }
  p.MT = 1;
  p.slices = m3d_cdiv(ncol16, p.NT);
  // ~4096 waves to fill 1024 SIMDs, at most 16 tiles per wave
  const int rs_cap = GEMM_RS_CAP, rs_tiles = GEMM_RS_TILES;
  int64_t wgs = m3d_cdiv(ntiles, 4);
  int64_t cap = rs_cap / p.slices;
  if (cap < m3d_cdiv(ntiles, rs_tiles)) cap = m3d_cdiv(ntiles, rs_tiles);
  if (cap < 1) cap = 1;
  if (wgs > cap) wgs = cap;
  if (wgs < 1) wgs = 1;
  p.wgs = wgs;
  return p;
}

int m3d_gemm_direct_stat_parts(int64_t M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0) return 1;
  return (int)plan_rows(M, N, K, 1).wgs;
}

static int gemm_direct_try_impl(const GemmArgs& g, hipStream_t st);
// bf16 activation storage (g.io != 0) exists in these kernels only: "not covered" (1: the caller's LDS-tiled fp32 fallback)
// becomes M3D_ERR_UNSUPPORTED
int m3d_gemm_direct_try(const GemmArgs& g, hipStream_t st) {
  if (g.io != 0 && g.io != 1 && g.io != 3 && g.io != 5) return M3D_ERR_INVALID;
  const int rc = gemm_direct_try_impl(g, st);
  return (rc == 1 && g.io != 0) ? M3D_ERR_UNSUPPORTED : rc;
}
static int gemm_direct_try_impl(const GemmArgs& g, hipStream_t st) {
  const int K = g.k0 + g.k1;
  // debugging aid (compile time): GEMM_DISABLE bit mask (2 rowstream, 4 kloop, 8 statistics mode) -> LDS-tiled fallback
  const int disable = GEMM_DISABLE;
  if (g.a_cm || g.splitk > 1) return g.pro_z ? M3D_ERR_UNSUPPORTED : 1;  // column-major A / split-K: the LDS-tiled kernel
  // accumulate: the plain epilogue (C += A B^T), or — round 6 — the affine one as a residual (C = act(affine(A B^T) + C))
  if (g.accumulate && g.stat_part) return g.pro_z ? M3D_ERR_UNSUPPORTED : 1;
  if (g.accumulate && (g.scale || g.shift || g.act) && (g.c_split > 0 || g.pro_z)) return M3D_ERR_UNSUPPORTED;
  if ((disable & 2) && K <= 64) return 1;
  if ((disable & 4) && K > 64) return 1;
  if ((disable & 8) && g.stat_part) return 1;
  const bool affine = g.scale || g.shift || g.act;
  if (g.stat_part && affine) return 1;
  const int mode = g.stat_part ? 1 : (affine ? 2 : 0);
  // vector loads: every fragment is one aligned float4
  const bool vec = ((g.lda0 & 3) == 0) && ((g.k0 & 3) == 0) && al16(g.a0) &&
                   (g.k1 == 0 || (((g.lda1 & 3) == 0) && ((g.k1 & 3) == 0) && al16(g.a1))) &&
                   (g.b_cm || (((g.ldb & 3) == 0) && al16(g.b)));
  const int cvec = ((g.ldc & 3) == 0) && al16(g.c) && ((g.N & 3) == 0);
  if (!vec && g.k1 != 0) return 1;       // the scalar variant reads a single A operand
  if (g.b_cm && (g.k1 != 0 || mode != 0)) return 1;  // column-major W is the dgrad pattern only
  // 32-bit byte offsets inside 2 GiB buffer descriptors (a gathered A0 is assumed no larger than A1 / C rows)
  const int64_t lim = (int64_t)M3D_BUF_BYTES - 64;
  if (g.M * g.lda0 * 4 > lim || (g.k1 > 0 && g.M * g.lda1 * 4 > lim) || (int64_t)(g.b_cm ? K : g.N) * g.ldb * 4 > lim)
    return 1;
  const int variant = vec ? (g.b_cm ? 2 : (g.k1 > 0 ? 1 : 0)) : (g.b_cm ? 4 : 3);
  // the BatchNorm-backward prologue exists for the vector-load dgrad kernels only (m3d_bn_dgrad_f32 checks the rest)
  if (g.pro_z && (variant != 2 || K > PRO_KMAX || g.a0_rows || g.lda0 != K)) return M3D_ERR_UNSUPPORTED;
  if (g.c_split > 0 && (mode != 0 || (g.c_split & 3) || (g.N & 3) || g.c_split >= g.N || !g.c1 || (g.ldc1 & 3) ||
                        (g.ldc & 3) || !al16(g.c1) || !al16(g.c)))
    return M3D_ERR_UNSUPPORTED;
  const RowPlan rp = plan_rows(g.M, g.N, K, mode);
  if (rp.slices > 65535) return 1;
  // the forward BatchNorm prologue: row-stream kernel, vector loads of ONE un-gathered operand, statistics epilogue
  if (g.fpro && (!rp.rowstream || variant != 0 || mode != 1 || g.a0_rows || g.lda0 != g.k0 || K > 64 || !g.fpro_slots ||
                 !g.fpro_scale || !g.fpro_shift || (g.fpro_y && !al16(g.fpro_y))))
    return M3D_ERR_UNSUPPORTED;
  dim3 grid((unsigned)rp.wgs, (unsigned)rp.slices);
  GemmArgs gk = g;
  gk.ksplit = rp.ksplit;
  gk.acc_pre = g.accumulate && mode == 0 && cvec && g.M * g.ldc * 4 <= lim &&
               (g.c_split == 0 || g.M * g.ldc1 * 4 <= lim);
  bool ok;
  if (rp.rowstream) {
    if (rp.NT == 4) ok = launch_rowstream<4>(gk, mode, rp.KQ, variant, grid, st, cvec);
    else if (rp.NT == 2) ok = launch_rowstream<2>(gk, mode, rp.KQ, variant, grid, st, cvec);
    else ok = launch_rowstream<1>(gk, mode, rp.KQ, variant, grid, st, cvec);
  } else {
    if (rp.MT == 2 && rp.NT == 4) ok = launch_kloop<2, 4>(gk, mode, variant, grid, st, cvec);
    else if (rp.MT == 2) ok = launch_kloop<2, 2>(gk, mode, variant, grid, st, cvec);
    else if (rp.NT == 2) ok = launch_kloop<1, 2>(gk, mode, variant, grid, st, cvec);
    else ok = launch_kloop<1, 1>(gk, mode, variant, grid, st, cvec);
  }
  if (!ok) return M3D_ERR_UNSUPPORTED;  // (an activation layout this shape's kernels are not instantiated for)
  return hipGetLastError() == hipSuccess ? M3D_OK : M3D_ERR_LAUNCH;
}

// ---- two products in one launch (gemm_kloop_pair_kernel) ----
template <int MTW, int NTW, int IO>
static void launch_kl_pair_io(const GemmArgs& g0, const GemmArgs& g1, int mode, bool bcm, bool bf, dim3 grid, hipStream_t st,
                              int cvec0, int cvec1, unsigned wgs0) {
#define M3D_PAIR(MODE_, BCM_, BF_) \
  hipLaunchKernelGGL((gemm_kloop_pair_kernel<MTW, NTW, MODE_, BCM_, BF_, IO>), grid, dim3(256), 0, st, g0, g1, cvec0, cvec1, wgs0)
  if (mode == 1) { if (bf) M3D_PAIR(1, false, true); else M3D_PAIR(1, false, false); }
  else if (bcm) { if (bf) M3D_PAIR(0, true, true); else M3D_PAIR(0, true, false); }
  else { if (bf) M3D_PAIR(0, false, true); else M3D_PAIR(0, false, false); }
#undef M3D_PAIR
}
template <int MTW, int NTW>
static void launch_kl_pair(const GemmArgs& g0, const GemmArgs& g1, int mode, bool bcm, bool bf, dim3 grid, hipStream_t st,
                           int cvec0, int cvec1, unsigned wgs0) {
  if (g0.io == 1) launch_kl_pair_io<MTW, NTW, 1>(g0, g1, mode, bcm, bf, grid, st, cvec0, cvec1, wgs0);
  else launch_kl_pair_io<MTW, NTW, 0>(g0, g1, mode, bcm, bf, grid, st, cvec0, cvec1, wgs0);
}

// M3D_OK when both products went out as one launch; 1 when the pair does not fit (the caller launches them one by one):
// K > 64 on both sides, plain row-major operands with 16-byte rows, the same output shape, epilogue (plain, or slot-mode
// statistics) and tile plan.
int m3d_gemm_direct_pair_try(const GemmArgs& a, const GemmArgs& b, hipStream_t st) {
  const GemmArgs* gs[2] = {&a, &b};
  if (a.M != b.M || a.N != b.N || a.b_cm != b.b_cm || a.bf16 != b.bf16 || a.io != b.io || (a.io != 0 && a.io != 1)) return 1;
  if ((a.stat_part != nullptr) != (b.stat_part != nullptr)) return 1;
  const int mode = a.stat_part ? 1 : 0;
  int cvec[2];
  RowPlan rp[2];
  GemmArgs gk[2];
  const int64_t lim = (int64_t)M3D_BUF_BYTES - 64;
  for (int i = 0; i < 2; ++i) {
    const GemmArgs& g = *gs[i];
    const int K = g.k0;
    if (g.k1 != 0 || g.a0_rows || g.a_cm || g.splitk > 1 || g.pro_z || g.c_split > 0 || K <= 64) return 1;
    if (g.scale || g.shift || g.act) return 1;
    if (g.accumulate && g.stat_part) return 1;
    if (g.b_cm && mode != 0) return 1;
    if (g.stat_part && g.stat_slots <= 0) return 1;  // (per-workgroup partial rows are sized for the single launch)
    const bool vec = ((g.lda0 & 3) == 0) && ((K & 3) == 0) && al16(g.a0) && (g.b_cm || (((g.ldb & 3) == 0) && al16(g.b)));
    if (!vec) return 1;
    if (g.bf16 && (K & 31)) return 1;
    if (g.M * g.lda0 * 4 > lim || (int64_t)(g.b_cm ? K : g.N) * g.ldb * 4 > lim) return 1;
    cvec[i] = ((g.ldc & 3) == 0) && al16(g.c) && ((g.N & 3) == 0);
    rp[i] = plan_rows(g.M, g.N, K, mode);
    if (rp[i].rowstream) return 1;
    gk[i] = g;
    gk[i].ksplit = rp[i].ksplit;
    gk[i].acc_pre = g.accumulate && mode == 0 && cvec[i] && g.M * g.ldc * 4 <= lim;
  }
  if (rp[0].MT != rp[1].MT || rp[0].NT != rp[1].NT || rp[0].slices != rp[1].slices || rp[0].slices > 65535) return 1;
  if ((GEMM_DISABLE) & 4) return 1;
  dim3 grid((unsigned)(rp[0].wgs + rp[1].wgs), (unsigned)rp[0].slices);
  const bool bcm = a.b_cm != 0, bf = a.bf16 != 0;
  const unsigned w0 = (unsigned)rp[0].wgs;
  if (rp[0].MT == 2 && rp[0].NT == 4) launch_kl_pair<2, 4>(gk[0], gk[1], mode, bcm, bf, grid, st, cvec[0], cvec[1], w0);
  else if (rp[0].MT == 2) launch_kl_pair<2, 2>(gk[0], gk[1], mode, bcm, bf, grid, st, cvec[0], cvec[1], w0);
  else if (rp[0].NT == 2) launch_kl_pair<1, 2>(gk[0], gk[1], mode, bcm, bf, grid, st, cvec[0], cvec[1], w0);
  else launch_kl_pair<1, 1>(gk[0], gk[1], mode, bcm, bf, grid, st, cvec[0], cvec[1], w0);
  return hipGetLastError() == hipSuccess ? M3D_OK : M3D_ERR_LAUNCH;
}

// ------------------------------------------------------------------------------------------
// m3d_linear_wgrad_f32: dW[N, k0+k1] (+)= dZ[M,N]^T [X0[rows] | X1]
// MFMA A operand = dZ^T (i = n, kk = row), B operand = X (kk = row, j = k): per 4-row step a lane loads one dword
// of dZ and one of X per tile (16 consecutive lanes = 64 contiguous bytes of a row).  The concatenation / row
// gather of the FP modules is read in place.  Row splits do not meet in same-address atomics: split s stores its
// [N, K] partial into the workspace and wgrad_reduce_kernel adds the partials.
// ------------------------------------------------------------------------------------------
struct WgradArgs {
  const float* dz; int64_t lddz;
  const float* x0; int64_t ldx0; const int32_t* rows; int k0;
  const float* x1; int64_t ldx1; int k1;
  int64_t M; int N;
  float* dw; int64_t lddw; int accumulate;
  float* ws;  // [S][N][K] partials (S > 1)
  int S; int64_t steps_per_split;
  int bf16;  // != 0 (with vec): operands rounded to bf16, v_mfma_f32_16x16x32_bf16, 32 rows per step (the net's "bf16" mode)
  int io;   // != 0: dz, x0 and x1 hold bf16 (M3D_IO_BF16; leading dimensions in elements); dw / ws stay fp32
  int vec;  // != 0: permuted tile columns (tile a of a TN-tile group holds n = nb + TN*i + a, likewise k): a lane's TN / TK
            // operand values are consecutive floats -> ONE 4*TN / 4*TK-byte load instead of TN / TK dword loads, and one
            // vector store per accumulator row (the level-1 layers stream at dword granularity otherwise: ~2 TB/s)
};

template <int V>
__device__ __forceinline__ void ldv(rsrc_t r, unsigned off, float (&out)[V]) {
  if constexpr (V == 1) {
    out[0] = ld1(r, off);
  } else if constexpr (V == 2) {
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    const f32x2_t v = __builtin_bit_cast(f32x2_t, __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0));
    out[0] = v[0]; out[1] = v[1];
  } else {
    const float4 v = ld4(r, off);
    out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
  }
}

// V consecutive elements at the fp32-byte offset `off` (H: stored as bf16, read at off / 2 through a 1 GiB descriptor)
template <int V, bool H>
__device__ __forceinline__ void ldvh(rsrc_t r, unsigned off, float (&out)[V]) {
  if constexpr (!H) {
    ldv<V>(r, off, out);
  } else if constexpr (V == 1) {
    out[0] = ld1h<true>(r, off);
  } else if constexpr (V == 2) {
    const unsigned v = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(r, off >> 1, 0, 0);
    out[0] = __uint_as_float(v << 16); out[1] = __uint_as_float(v & 0xffff0000u);
  } else {
    const float4 v = ld4h<true>(r, off);
    out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
  }
}

template <int TN, int TK>
struct WgradFrag {
  float a[TN], b[TK];
};

// operands of the 4-row step `s` (rows 4s .. 4s+3, this lane: row 4s + lg); buffer loads, no branches
template <int TN, int TK, bool H = false>
// `rr`: the row of x0 that row 4s + lg reads (the caller's prefetched rows[] entry, or the row itself)
__device__ __forceinline__ WgradFrag<TN, TK> wgrad_load(const WgradArgs& g, rsrc_t rz, rsrc_t rx0, rsrc_t rx1, int64_t s,
                                                        bool live, int64_t rr, int nb, int kb, int lr, int lg, int K) {
  WgradFrag<TN, TK> f;
  const int64_t m = 4 * s + lg;
  const bool ok = live && m < g.M;
  const int64_t mc = ok ? m : 0;
  const unsigned oz = ok ? (unsigned)(mc * g.lddz * 4) : OOB;
  const unsigned o0 = (ok && rr >= 0) ? (unsigned)(rr * g.ldx0 * 4) : OOB;
  const unsigned o1 = (ok && g.k1 > 0) ? (unsigned)(mc * g.ldx1 * 4) : OOB;
  if (g.vec) {
    const int n0 = nb + TN * lr;
    ldvh<TN, H>(rz, (oz != OOB && n0 < g.N) ? oz + 4u * (unsigned)n0 : OOB, f.a);
    const int kk = kb + TK * lr;
    const bool in0 = kk < g.k0;
    float t0[TK], t1[TK];
    ldvh<TK, H>(rx0, (in0 && o0 != OOB) ? o0 + 4u * (unsigned)kk : OOB, t0);
    ldvh<TK, H>(rx1, (!in0 && kk < K && o1 != OOB) ? o1 + 4u * (unsigned)(kk - g.k0) : OOB, t1);
#pragma unroll
    for (int b = 0; b < TK; ++b) f.b[b] = t0[b] + t1[b];
    return f;
  }
#pragma unroll
  for (int a = 0; a < TN; ++a) {
    const int n = nb + 16 * a + lr;
    f.a[a] = ld1h<H>(rz, (oz != OOB && n < g.N) ? oz + 4u * (unsigned)n : OOB);
  }
#pragma unroll
  for (int b = 0; b < TK; ++b) {
    const int k = kb + 16 * b + lr;
    const bool in0 = k < g.k0;
    // both operands are always read: the one that does not hold column k is OOB (returns 0, no memory access)
    f.b[b] = ld1h<H>(rx0, (in0 && o0 != OOB) ? o0 + 4u * (unsigned)k : OOB) +
             ld1h<H>(rx1, (!in0 && k < K && o1 != OOB) ? o1 + 4u * (unsigned)(k - g.k0) : OOB);
  }
  return f;
}

// The 4-row steps s0 .. s1 of one wave on permuted columns (WgradArgs::vec), DEPTH steps per trip.  A trip is: the row
// numbers of the NEXT trip (x0[rows]: the FP modules' gather, fc0's cell order), every operand load of this trip, ONE wait,
// the DEPTH x TN x TK MFMAs.  Nothing is computed from a loaded value in front of the last load and there is no branch in
// the body.  Round 6: the earlier body added the two halves of a concatenated X row (x0 | x1) inside the load helper of each
// step and branched on `vec` there — the compiler could not move the next step's loads across either, a trip was DEPTH + 1
// dependent round trips to memory (5.4 us for 32 MFMAs on the 16-tile waves: 23 % of the fp32 matrix peak with the matrix
// pipe idle three quarters of the time, and the same time for 1, 2 or 4 steps per trip: profiles/r04u_*).
// ONEX: the tile's 16 TK columns lie in ONE of the two X operands (k0 a multiple of the tile width, or no x1): one load.
template <int TN, int TK, bool H, bool ONEX, int DEPTH>
__device__ __forceinline__ void wgrad_trips_vec(const WgradArgs& g, int64_t s0, int64_t s1, int nb, int kb, int lr, int lg,
                                                f32x4 (&acc)[TN][TK]) {
  const int K = g.k0 + g.k1;
  const bool has_rows = g.rows != nullptr;
  const rsrc_t rrows = mk_rsrc(g.rows);
  const rsrc_t rz = mk_rsrc_h<H>(g.dz);
  const int n0 = nb + TN * lr, kk = kb + TK * lr;
  const unsigned acol = n0 < g.N ? 4u * (unsigned)n0 : OOB;
  const bool in0 = kb < g.k0;  // (ONEX: uniform over the tile)
  const rsrc_t rxa = mk_rsrc_h<H>(ONEX ? (in0 ? g.x0 : g.x1) : g.x0), rxb = mk_rsrc_h<H>(g.x1);
  const int64_t ldxa = ONEX ? (in0 ? g.ldx0 : g.ldx1) : g.ldx0;
  const unsigned xcol = ONEX ? (kk < K ? 4u * (unsigned)(in0 ? kk : kk - g.k0) : OOB)
                             : (kk < g.k0 ? 4u * (unsigned)kk : OOB);
  const unsigned xcol1 = (!ONEX && kk >= g.k0 && kk < K) ? 4u * (unsigned)(kk - g.k0) : OOB;
  const bool mapped = has_rows && (!ONEX || in0);  // x0 is read through rows[]
  auto row_req = [&](int64_t st) -> int32_t {  // rows[4 st + lg] (0 where there is none: the lane then reads no operand either)
    const int64_t m = 4 * st + lg;
    return __builtin_amdgcn_raw_buffer_load_b32(rrows, (mapped && st < s1 && m < g.M) ? (unsigned)(4 * m) : OOB, 0, 0);
  };
  int32_t rnext[DEPTH];
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) rnext[d] = row_req(s0 + d);
  for (int64_t s = s0; s < s1; s += DEPTH) {
    float a[DEPTH][TN], b0[DEPTH][TK], b1[DEPTH][ONEX ? 1 : TK];
    int32_t rcur[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) { rcur[d] = rnext[d]; rnext[d] = row_req(s + DEPTH + d); }
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const int64_t m = 4 * (s + d) + lg;
      const bool ok = s + d < s1 && m < g.M;
      const int64_t mc = ok ? m : 0;
      const int64_t rr = mapped ? (int64_t)rcur[d] : mc;
      const unsigned oz = ok ? (unsigned)(mc * g.lddz * 4) : OOB;
      const unsigned ox = (ok && rr >= 0) ? (unsigned)(rr * ldxa * 4) : OOB;
      ldvh<TN, H>(rz, (oz != OOB && acol != OOB) ? oz + acol : OOB, a[d]);
      ldvh<TK, H>(rxa, (ox != OOB && xcol != OOB) ? ox + xcol : OOB, b0[d]);
      if constexpr (!ONEX) {
        const unsigned o1 = (ok && g.k1 > 0) ? (unsigned)(mc * g.ldx1 * 4) : OOB;
        ldvh<TK, H>(rxb, (o1 != OOB && xcol1 != OOB) ? o1 + xcol1 : OOB, b1[d]);
      }
    }
    // (the scheduler otherwise sinks the last loads between the MFMA groups to save registers and waits for them there)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TK; ++j) {
          float bv = b0[d][j];
          if constexpr (!ONEX) bv += b1[d][j];
          acc[i][j] = mfma16(a[d][i], bv, acc[i][j]);
        }
  }
}

// The same on the bf16 matrix cores (the net's "bf16" mode): a trip is a 32-row step; lane (lr, lg) supplies rows 8 lg .. 8 lg + 7
// of the step for its TN / TK (permuted) columns: 8 vector loads per operand, all in flight behind the row numbers of the NEXT
// trip, then TN x TK MFMAs of 32 rows each.  (Rounds 3-5 read rows[] inside the step and waited for it in front of every one of
// the 8 row loads: the FP modules' jobs — 29 % of the class's flops — were 8 dependent round trips per 16 MFMAs.)
template <int TN, int TK, bool H, bool ONEX>
__device__ __forceinline__ void wgrad_trips_bf16(const WgradArgs& g, int64_t s0, int64_t s1, int nb, int kb, int lr, int lg,
                                                 f32x4 (&acc)[TN][TK]) {
  const int K = g.k0 + g.k1;
  const bool has_rows = g.rows != nullptr;
  const rsrc_t rrows = mk_rsrc(g.rows);
  const rsrc_t rz = mk_rsrc_h<H>(g.dz);
  const int n0 = nb + TN * lr, kk = kb + TK * lr;
  const unsigned acol = n0 < g.N ? 4u * (unsigned)n0 : OOB;
  const bool in0 = kb < g.k0;
  const rsrc_t rxa = mk_rsrc_h<H>(ONEX ? (in0 ? g.x0 : g.x1) : g.x0), rxb = mk_rsrc_h<H>(g.x1);
  const int64_t ldxa = ONEX ? (in0 ? g.ldx0 : g.ldx1) : g.ldx0;
  const unsigned xcol = ONEX ? (kk < K ? 4u * (unsigned)(in0 ? kk : kk - g.k0) : OOB)
                             : (kk < g.k0 ? 4u * (unsigned)kk : OOB);
  const unsigned xcol1 = (!ONEX && kk >= g.k0 && kk < K) ? 4u * (unsigned)(kk - g.k0) : OOB;
  const bool mapped = has_rows && (!ONEX || in0);
  auto row_req = [&](int64_t s, int j) -> int32_t {
    const int64_t m = 4 * s + 8 * lg + j;
    return __builtin_amdgcn_raw_buffer_load_b32(rrows, (mapped && m < 4 * s1 && m < g.M) ? (unsigned)(4 * m) : OOB, 0, 0);
  };
  int32_t rnext[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) rnext[j] = row_req(s0, j);
  for (int64_t s = s0; s < s1; s += 8) {
    float av[8][TN], b0[8][TK], b1[8][ONEX ? 1 : TK];
    int32_t rcur[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { rcur[j] = rnext[j]; rnext[j] = row_req(s + 8, j); }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int64_t m = 4 * s + 8 * lg + j;
      const bool ok = m < 4 * s1 && m < g.M;
      const int64_t mc = ok ? m : 0;
      const int64_t rr = mapped ? (int64_t)rcur[j] : mc;
      const unsigned oz = ok ? (unsigned)(mc * g.lddz * 4) : OOB;
      const unsigned ox = (ok && rr >= 0) ? (unsigned)(rr * ldxa * 4) : OOB;
      ldvh<TN, H>(rz, (oz != OOB && acol != OOB) ? oz + acol : OOB, av[j]);
      ldvh<TK, H>(rxa, (ox != OOB && xcol != OOB) ? ox + xcol : OOB, b0[j]);
      if constexpr (!ONEX) {
        const unsigned o1 = (ok && g.k1 > 0) ? (unsigned)(mc * g.ldx1 * 4) : OOB;
        ldvh<TK, H>(rxb, (o1 != OOB && xcol1 != OOB) ? o1 + xcol1 : OOB, b1[j]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    Bf16Frag fa[TN], fb[TK];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[a].u[i] = pack_bf16(av[2 * i][a], av[2 * i + 1][a]);
#pragma unroll
    for (int b = 0; b < TK; ++b)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float lo = b0[2 * i][b], hi = b0[2 * i + 1][b];
        if constexpr (!ONEX) { lo += b1[2 * i][b]; hi += b1[2 * i + 1][b]; }
        fb[b].u[i] = pack_bf16(lo, hi);
      }
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
      for (int b = 0; b < TK; ++b) acc[a][b] = mfma_bf16(fa[a].v, fb[b].v, acc[a][b]);
  }
}

// Every WAVE owns a contiguous row range; the four waves of a workgroup then add their [16*TN, 16*TK] accumulators
// through LDS (plain stores + one barrier; LDS float atomics run at ~1 lane per 3 clocks on gfx950) and the workgroup
// stores ONE partial to ws, so the chip can be filled with waves (HBM streaming needs bytes in flight: with 2 048 waves
// = 2 per SIMD and 8 loads each the level-1 layers sat at 2 TB/s) without multiplying the partial traffic.
// WGRAD_DEPTH 4-row steps are in flight per trip.
#ifndef WGRAD_DEPTH
#define WGRAD_DEPTH 4
#endif
#ifndef WGRAD_WGR16
#define WGRAD_WGR16 1  // the 16-tile waves (64 x 64 outputs: the deep layers) also meet in LDS (48 KB per workgroup): a quarter of
#endif                 // the partial-sum traffic, which was most of those layers' time (62 partials of 512 KB for a 3 200-row layer)
#ifndef WGRAD_DEPTH_BIG
#define WGRAD_DEPTH_BIG 4  // 4-row steps in flight per trip of the waves with 8 or 16 accumulator tiles
#endif
template <int TN, int TK, bool BF = false, bool H = false>
__device__ __forceinline__ void wgrad2_body(const WgradArgs& g, const unsigned bx, const unsigned by, const unsigned bz) {
  // (WGRAD_WGR16 = 0: the 16-tile waves of the deep, few-row layers keep one partial per WAVE, rounds 3-5)
  constexpr bool WGR = TN * TK < 16 || WGRAD_WGR16;
  constexpr int DEPTH = TN * TK < 8 ? WGRAD_DEPTH : WGRAD_DEPTH_BIG;
  __shared__ float red[WGR ? 3 : 1][WGR ? TN * TK * 256 : 1];  // accumulators of waves 1..3 (wave 0 keeps its own)
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, lr = lane & 15, lg = lane >> 4;
  const int K = g.k0 + g.k1;
  const int nb = by * 16 * TN, kb = bz * 16 * TK;
  const int64_t steps_total = (g.M + 3) >> 2;
  const int64_t split = (int64_t)bx * 4 + wid;           // wave-level row split
  const int64_t s0 = split * g.steps_per_split;
  const int64_t s1 = s0 + g.steps_per_split < steps_total ? s0 + g.steps_per_split : steps_total;
  f32x4 acc[TN][TK];
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TK; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const rsrc_t rz = mk_rsrc_h<H>(g.dz), rx0 = mk_rsrc_h<H>(g.x0), rx1 = mk_rsrc_h<H>(g.x1);

  if constexpr (BF && TN * TK >= 8) {
    if (g.bf16 && g.vec) {
      if (g.k1 == 0 || g.k0 % (16 * TK) == 0) wgrad_trips_bf16<TN, TK, H, true>(g, s0, s1, nb, kb, lr, lg, acc);
      else wgrad_trips_bf16<TN, TK, H, false>(g, s0, s1, nb, kb, lr, lg, acc);
      goto reduce_and_store;
    }
  }
  if (g.vec) {
    // (uniform) one X load per step where the tile does not straddle the seam of a concatenated input
    if (g.k1 == 0 || g.k0 % (16 * TK) == 0) wgrad_trips_vec<TN, TK, H, true, DEPTH>(g, s0, s1, nb, kb, lr, lg, acc);
    else wgrad_trips_vec<TN, TK, H, false, (TN * TK < 8 ? DEPTH : 2)>(g, s0, s1, nb, kb, lr, lg, acc);  // (3 loads per step)
  } else {
  // x0[rows] (fc0 on the cell-sorted order): the row numbers of a trip are requested one trip AHEAD, behind the operand
  // loads of the trip before — read inside wgrad_load they were a dependent round trip in front of every 4-row step, and the
  // wait for them drained the operand loads of the step before (the waves of that job ran one step at a time)
  // (kept as the raw 32 bits until the next trip uses them: a sign extension right behind the load would wait for it)
  auto x0_row = [&](int64_t st) -> int32_t {
    const int64_t m = 4 * st + lg;
    const int64_t mc = (st < s1 && m < g.M) ? m : 0;
    return g.rows ? g.rows[mc] : (int32_t)mc;
  };
  int32_t rnext[DEPTH];
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) rnext[d] = x0_row(s0 + d);
  for (int64_t s = s0; s < s1; s += DEPTH) {
    WgradFrag<TN, TK> f[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
      f[d] = wgrad_load<TN, TK, H>(g, rz, rx0, rx1, s + d, s + d < s1, rnext[d], nb, kb, lr, lg, K);
    if (g.rows) {
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) rnext[d] = x0_row(s + DEPTH + d);
    } else {
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        const int64_t m = 4 * (s + DEPTH + d) + lg;
        rnext[d] = (s + DEPTH + d < s1 && m < g.M) ? (int32_t)m : 0;
      }
    }
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
      for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TK; ++b) acc[a][b] = mfma16(f[d].a[a], f[d].b[b], acc[a][b]);
  }
  }
reduce_and_store:
  int64_t part = split;
  if constexpr (WGR) {
    // ---- the workgroup's four accumulators meet in LDS (same lane layout: element (a, b, r) of lane l <-> l)
    if (wid > 0) {
#pragma unroll
      for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TK; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r) red[wid - 1][((a * TK + b) * 4 + r) * 64 + lane] = acc[a][b][r];
    }
    __syncthreads();
    if (wid > 0) return;
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
      for (int b = 0; b < TK; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int o = ((a * TK + b) * 4 + r) * 64 + lane;
          acc[a][b][r] += (red[0][o] + red[1][o]) + red[2][o];
        }
    part = bx;
  } else {
    if (s0 >= steps_total) return;  // (no rows: this wave has no partial slot either)
  }
  if (g.vec) {
    // permuted columns: accumulator (a, b, r) of lane (lr, lg) is dW[nb + TN*(4lg + r) + a][kb + TK*lr + b]: the TK values
    // of a row are consecutive -> one 4*TK-byte store (16 lanes = 64*TK contiguous bytes)
    const int kk = kb + TK * lr;
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = nb + TN * (4 * lg + r) + a;
        if (n < g.N && kk < K) {
          float* cp = g.S > 1 ? g.ws + ((size_t)part * g.N + n) * K + kk : g.dw + (int64_t)n * g.lddw + kk;
          const bool add = g.S <= 1 && g.accumulate;
          if constexpr (TK == 4) {
            float4 v = make_float4(acc[a][0][r], acc[a][1][r], acc[a][2][r], acc[a][3][r]);
            if (add) { const float4 o = *(float4*)cp; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
            *(float4*)cp = v;
          } else if constexpr (TK == 2) {
            float2 v = make_float2(acc[a][0][r], acc[a][1][r]);
            if (add) { const float2 o = *(float2*)cp; v.x += o.x; v.y += o.y; }
            *(float2*)cp = v;
          } else {
            *cp = add ? *cp + acc[a][0][r] : acc[a][0][r];
          }
        }
      }
    return;
  }
  // D layout: row n = 16a + 4lg + r, col k = 16b + lr  (16 lanes = 64 contiguous bytes per store)
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TK; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = nb + 16 * a + 4 * lg + r, k = kb + 16 * b + lr;
        if (n < g.N && k < K) {
          if (g.S > 1) {
            g.ws[((size_t)part * g.N + n) * K + k] = acc[a][b][r];
          } else {
            float* cp = g.dw + (int64_t)n * g.lddw + k;
            *cp = g.accumulate ? *cp + acc[a][b][r] : acc[a][b][r];
          }
        }
      }
}

template <int TN, int TK, bool H = false>
__global__ __launch_bounds__(256, WGRAD_MINW) void wgrad2_kernel(WgradArgs g) {
  wgrad2_body<TN, TK, false, H>(g, blockIdx.x, blockIdx.y, blockIdx.z);
}

// Several weight-gradient GEMMs of one tile class in ONE launch (m3d_linear_wgrad_batch).  The 29 Linear layers of the
// network need their dW only when the optimizer runs, so the whole backward pass can hand them over at its end: launched
// layer by layer they are 58 launches whose tails (a few waves per SIMD on the deep levels, partial-sum reduces of a few
// microseconds) never overlap, and — knock-out timing, tools/scratch/knockout.sh — they cost the training step 1.0 ms of
// wall time although they run on a side stream.  Workgroup w of the flattened grid belongs to job j with
// wg_start[j] <= w < wg_start[j + 1]; inside the job it is (x, y, z) of that job's own grid.
#define WGRAD_BATCH_MAX 16
struct WgradBatch {
  WgradArgs g[WGRAD_BATCH_MAX];
  unsigned wg_start[WGRAD_BATCH_MAX + 1];
  unsigned gx[WGRAD_BATCH_MAX], gy[WGRAD_BATCH_MAX], gz[WGRAD_BATCH_MAX];
  int njobs;
};
// Which workgroup of a job works on what (WGRAD_XCD, round 6).  Workgroups go to the 8 XCDs round-robin by their flat index and
// every XCD has an L2 of its own.  The gy x gz output tiles of ONE row slice read the same rows of dZ and X (dZ gz times, X gy
// times over): with the slice as the fastest index (rounds 3-5) those tiles landed on all eight XCDs and every one of them
// fetched its operands from the Infinity Cache — the 16-tile class moved ~540 MB for ~100 MB of operands and ran at that
// traffic's pace whatever the depth of its pipeline or the number of its waves.  Now XCD c (= flat index % 8; every job
// starts at a multiple of 8) takes the c-th eighth of the job in (tile fastest, slice slowest) order: the tiles of a slice
// run on one XCD at the same time and share its L2.
template <int TN, int TK, bool BF = false, bool H = false>
__global__ __launch_bounds__(256, WGRAD_MINW) void wgrad2_batch_kernel(WgradBatch b) {
  int j = 0;
#pragma unroll
  for (int i = 1; i < WGRAD_BATCH_MAX; ++i) j += (i < b.njobs && blockIdx.x >= b.wg_start[i]) ? 1 : 0;
  const unsigned w = blockIdx.x - b.wg_start[j];
  const unsigned gx = b.gx[j], gy = b.gy[j];
#if WGRAD_XCD
  const unsigned tiles = gy * b.gz[j], per = (b.wg_start[j + 1] - b.wg_start[j]) >> 3;
  const unsigned l = (w & 7u) * per + (w >> 3);
  if (l >= gx * tiles) return;  // (padding of the job to a multiple of 8 workgroups)
  const unsigned t = l % tiles;
  wgrad2_body<TN, TK, BF, H>(b.g[j], l / tiles, t % gy, t / gy);
#else
  wgrad2_body<TN, TK, BF, H>(b.g[j], w % gx, (w / gx) % gy, w / (gx * gy));
#endif
}

// dw[n][k] (+)= sum_s ws[s][n][k]
__device__ __forceinline__ void wgrad_reduce_body(const float* __restrict__ ws, int S, int N, int K,
                                                  float* __restrict__ dw, int64_t lddw, int accumulate, unsigned bx,
                                                  unsigned by, unsigned ny) {
  const int e = bx * 256 + threadIdx.x;
  const int E = N * K;
  if (e >= E) return;
  const int per = (S + ny - 1) / ny;
  const int p0 = by * per, p1 = min(S, p0 + per);
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int p = p0;
  for (; p + 3 < p1; p += 4) {
    s0 += ws[(size_t)p * E + e];
    s1 += ws[(size_t)(p + 1) * E + e];
    s2 += ws[(size_t)(p + 2) * E + e];
    s3 += ws[(size_t)(p + 3) * E + e];
  }
  for (; p < p1; ++p) s0 += ws[(size_t)p * E + e];
  const float v = (s0 + s1) + (s2 + s3);
  float* cp = dw + (int64_t)(e / K) * lddw + (e % K);
  if (ny == 1) *cp = accumulate ? *cp + v : v;
  else if (p1 > p0) atomicAdd(cp, v);
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws, int S, int N, int K,
                                                           float* __restrict__ dw, int64_t lddw, int accumulate) {
  wgrad_reduce_body(ws, S, N, K, dw, lddw, accumulate, blockIdx.x, blockIdx.y, gridDim.y);
}

#define WREDUCE_BATCH_MAX 32
struct WreduceBatch {
  const float* ws[WREDUCE_BATCH_MAX]; float* dw[WREDUCE_BATCH_MAX]; int64_t lddw[WREDUCE_BATCH_MAX];
  int S[WREDUCE_BATCH_MAX], N[WREDUCE_BATCH_MAX], K[WREDUCE_BATCH_MAX];
  unsigned gx[WREDUCE_BATCH_MAX], gy[WREDUCE_BATCH_MAX], wg_start[WREDUCE_BATCH_MAX + 1];
  int njobs, accumulate;
};
__global__ __launch_bounds__(256) void wgrad_reduce_batch_kernel(WreduceBatch b) {
  int j = 0;
#pragma unroll
  for (int i = 1; i < WREDUCE_BATCH_MAX; ++i) j += (i < b.njobs && blockIdx.x >= b.wg_start[i]) ? 1 : 0;
  const unsigned w = blockIdx.x - b.wg_start[j];
  wgrad_reduce_body(b.ws[j], b.S[j], b.N[j], b.K[j], b.dw[j], b.lddw[j], b.accumulate, w % b.gx[j], w / b.gx[j], b.gy[j]);
}

__global__ __launch_bounds__(256) void zero_rows_kernel(float* __restrict__ p, int64_t ld, int N, int K) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e < N * K) p[(int64_t)(e / K) * ld + (e % K)] = 0.f;
}

struct WgradPlan { int TN, TK; int64_t by, bz, S, spw, wgs; };  // S: partials in ws, spw: 4-row steps per WAVE, wgs: grid.x
static WgradPlan wgrad_plan(int64_t M, int N, int K, int64_t target_waves = 0) {
  WgradPlan p;
  const int tn = (int)m3d_cdiv(N, 16), tk = (int)m3d_cdiv(K, 16);
  p.TN = tn >= 4 ? 4 : (tn >= 2 ? 2 : 1);
  p.TK = tk >= 4 ? 4 : (tk >= 2 ? 2 : 1);
  p.by = m3d_cdiv(N, 16 * p.TN);
  p.bz = m3d_cdiv(K, 16 * p.TK);
  const bool wgr = p.TN * p.TK < 16 || WGRAD_WGR16;  // workgroup-level partials (see wgrad2_kernel)
  const int64_t steps_total = m3d_cdiv(M > 0 ? M : 1, 4);
  const int target_env = WGRAD_WAVES;
  // target_waves > 0: this job's share of a batched launch (m3d_linear_wgrad_batch), instead of the whole chip
  const int64_t target = target_waves > 0 ? target_waves : (target_env > 0 ? target_env : (p.TN * p.TK < 16 ? 8192 : 2048));
  int64_t waves = m3d_cdiv(target, p.by * p.bz);  // streaming tiles: ~8 waves per SIMD over the chip
  if (waves > steps_total / 8) waves = steps_total / 8;  // >= 8 steps (32 rows) per wave
  if (waves < 1) waves = 1;
  if (wgr) {
    p.wgs = m3d_cdiv(waves, 4);
    p.spw = m3d_cdiv(steps_total, p.wgs * 4);
    p.wgs = m3d_cdiv(steps_total, p.spw * 4);
    p.S = p.wgs;
  } else {
    p.spw = m3d_cdiv(steps_total, waves);
    p.S = m3d_cdiv(steps_total, p.spw);  // one partial per wave that owns rows
    p.wgs = m3d_cdiv(p.S, 4);
  }
  return p;
}

extern "C" size_t m3d_linear_wgrad_workspace_bytes(int64_t M, int32_t N, int32_t K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  const WgradPlan p = wgrad_plan(M, N, K);
  return p.S > 1 ? (size_t)p.S * N * K * sizeof(float) : 0;
}

// permuted-column vector loads / stores (WgradArgs::vec): every row offset must stay a multiple of the vector width
static int wgrad_vec_ok(const WgradArgs& g, int TN, int TK) {
  const bool off = WGRAD_VEC == 0;
  if (off) return 0;
  const int K = g.k0 + g.k1;
  auto al = [](const void* p) { return (((uintptr_t)p) & 15) == 0; };
  if (g.N % TN || K % TK || (g.k1 > 0 && g.k0 % TK)) return 0;
  if (g.lddz % TN || g.ldx0 % TK || (g.k1 > 0 && g.ldx1 % TK) || g.lddw % TK) return 0;
  if (!al(g.dz) || !al(g.x0) || (g.k1 > 0 && !al(g.x1)) || !al(g.dw) || (g.ws && !al(g.ws))) return 0;
  return 1;
}

template <int TN>
static void launch_wgrad2(const WgradArgs& g, int TK, dim3 grid, hipStream_t st) {
  if (g.io) {
    switch (TK) {
      case 1: hipLaunchKernelGGL((wgrad2_kernel<TN, 1, true>), grid, dim3(256), 0, st, g); break;
      case 2: hipLaunchKernelGGL((wgrad2_kernel<TN, 2, true>), grid, dim3(256), 0, st, g); break;
      default: hipLaunchKernelGGL((wgrad2_kernel<TN, 4, true>), grid, dim3(256), 0, st, g); break;
    }
    return;
  }
  switch (TK) {
    case 1: hipLaunchKernelGGL((wgrad2_kernel<TN, 1>), grid, dim3(256), 0, st, g); break;
    case 2: hipLaunchKernelGGL((wgrad2_kernel<TN, 2>), grid, dim3(256), 0, st, g); break;
    default: hipLaunchKernelGGL((wgrad2_kernel<TN, 4>), grid, dim3(256), 0, st, g); break;
  }
}

extern "C" int m3d_linear_wgrad_f32(const float* dz, int64_t lddz, const float* x0, int64_t ldx0,
                                    const int32_t* x0_rows, int32_t k0, const float* x1, int64_t ldx1, int32_t k1,
                                    int64_t M, int32_t N, float* dw, int64_t lddw, int32_t accumulate, void* ws,
                                    void* stream) {
  if (M < 0 || N < 0 || k0 < 0 || k1 < 0) return M3D_ERR_INVALID;
  const int K = k0 + k1;
  if (N == 0 || K == 0) return M3D_OK;
  if (!dw) return M3D_ERR_INVALID;
  hipStream_t st = (hipStream_t)stream;
  if (M == 0) {
    if (!(accumulate & 1)) hipLaunchKernelGGL(zero_rows_kernel, dim3((N * K + 255) / 256), dim3(256), 0, st, dw, lddw, N, K);
    M3D_CHECK_LAUNCH();
    return M3D_OK;
  }
  if (!dz || k0 < 1 || !x0 || (k1 > 0 && !x1)) return M3D_ERR_INVALID;
  const WgradPlan p = wgrad_plan(M, N, K);
  if (p.S > 1 && !ws) return M3D_ERR_INVALID;
  if (p.by > 65535 || p.bz > 65535) return M3D_ERR_UNSUPPORTED;
  {  // 32-bit byte offsets inside 2 GiB buffer descriptors
    const int64_t lim = (int64_t)M3D_BUF_BYTES - 64;
    if (M * lddz * 4 > lim || M * ldx0 * 4 > lim || (k1 > 0 && M * ldx1 * 4 > lim)) return M3D_ERR_UNSUPPORTED;
  }
  WgradArgs g;
  g.dz = dz; g.lddz = lddz; g.x0 = x0; g.ldx0 = ldx0; g.rows = x0_rows; g.k0 = k0; g.x1 = x1; g.ldx1 = ldx1; g.k1 = k1;
  g.M = M; g.N = N; g.dw = dw; g.lddw = lddw; g.accumulate = accumulate; g.ws = (float*)ws; g.S = (int)p.S;
  g.steps_per_split = p.spw;
  g.io = (accumulate & M3D_IO_BF16) ? 1 : 0;  // accumulate: bit 0 = add into dw, M3D_IO_BF16 = dz / x0 / x1 hold bf16
  g.accumulate = accumulate & 1;
  accumulate &= 1;
  g.vec = wgrad_vec_ok(g, p.TN, p.TK);
  g.bf16 = 0;
  dim3 grid((unsigned)p.wgs, (unsigned)p.by, (unsigned)p.bz);
  if (p.TN == 4) launch_wgrad2<4>(g, p.TK, grid, st);
  else if (p.TN == 2) launch_wgrad2<2>(g, p.TK, grid, st);
  else launch_wgrad2<1>(g, p.TK, grid, st);
  if (p.S > 1) {
    const int E = N * K;
    const int gx = (E + 255) / 256;
    int gy = (int)(p.S / 16);  // >= 16 partials per chunk
    if (gy > 2048 / gx) gy = 2048 / gx;
    if (gy < 1) gy = 1;
    if (gy > 1 && !accumulate) hipLaunchKernelGGL(zero_rows_kernel, dim3(gx), dim3(256), 0, st, dw, lddw, N, K);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(gx, gy), dim3(256), 0, st, (const float*)ws, (int)p.S, N, K, dw, lddw,
                       accumulate);
  }
  M3D_CHECK_LAUNCH();
  return M3D_OK;
}

template <int TN, int TK>
static void launch_wgrad2_batch(const WgradBatch& b, unsigned total, hipStream_t st) {
  const bool h = b.g[0].io != 0;  // (one activation layout per call: m3d_linear_wgrad_batch)
  if constexpr (TN * TK >= 8) {  // bf16 matrix-core variant (its register budget would cost the fp32 kernel a wave per SIMD)
    if (b.g[0].bf16) {
      if (h) hipLaunchKernelGGL((wgrad2_batch_kernel<TN, TK, true, true>), dim3(total), dim3(256), 0, st, b);
      else hipLaunchKernelGGL((wgrad2_batch_kernel<TN, TK, true>), dim3(total), dim3(256), 0, st, b);
      return;
    }
  }
  if (h) hipLaunchKernelGGL((wgrad2_batch_kernel<TN, TK, false, true>), dim3(total), dim3(256), 0, st, b);
  else hipLaunchKernelGGL((wgrad2_batch_kernel<TN, TK>), dim3(total), dim3(256), 0, st, b);
}

// dW of several layers at once (see WgradBatch).  Host arrays of length njobs; `accumulate` != 0 (gradient sinks) is
// required: the partial-sum reduce of a job may add into dw from several workgroups.  ws[j]: the job's own
// m3d_linear_wgrad_workspace_bytes(M, N, K) scratch (NULL when that is 0).
extern "C" int m3d_linear_wgrad_batch(int32_t njobs, const float* const* dz, const int64_t* lddz, const float* const* x0,
                                      const int64_t* ldx0, const int32_t* const* x0_rows, const int32_t* k0,
                                      const float* const* x1, const int64_t* ldx1, const int32_t* k1, const int64_t* M,
                                      const int32_t* N, float* const* dw, const int64_t* lddw, int32_t accumulate,
                                      void* const* ws, void* stream) {
  if (njobs < 0) return M3D_ERR_INVALID;
  if (njobs == 0) return M3D_OK;
  // bit 0: add into dw (required); bit 8: bf16 matrix cores; M3D_IO_BF16: dz / x0 / x1 of EVERY job hold bf16
  if (!(accumulate & 1)) return M3D_ERR_UNSUPPORTED;
  if (!dz || !lddz || !x0 || !ldx0 || !x0_rows || !k0 || !x1 || !ldx1 || !k1 || !M || !N || !dw || !lddw || !ws)
    return M3D_ERR_INVALID;
  hipStream_t st = (hipStream_t)stream;
  static const int variants[7][2] = {{4, 4}, {4, 2}, {2, 4}, {2, 2}, {1, 2}, {2, 1}, {1, 1}};
  WreduceBatch rb;
  rb.njobs = 0; rb.accumulate = 1;
  unsigned rtotal = 0;
  auto flush_reduce = [&]() {
    if (rb.njobs == 0) return;
    rb.wg_start[rb.njobs] = rtotal;
    for (int i = rb.njobs + 1; i <= WREDUCE_BATCH_MAX; ++i) rb.wg_start[i] = rtotal;
    hipLaunchKernelGGL(wgrad_reduce_batch_kernel, dim3(rtotal), dim3(256), 0, st, rb);
    rb.njobs = 0; rtotal = 0;
  };
  // every job gets a share of its tile class's wave budget in proportion to its flops: planned one by one (as if each
  // had the chip to itself) the 14 deep-layer jobs of a batch would split their rows 14 x finer than needed and pay
  // for it in partial-sum traffic
  const int budget_big = WGRAD_BATCH_WAVES_BIG, budget_small = WGRAD_BATCH_WAVES_SMALL;
  double class_flops[7] = {0, 0, 0, 0, 0, 0, 0};
  std::vector<int> cls_of(njobs, -1);
  for (int j = 0; j < njobs; ++j) {
    const int K = k0[j] + k1[j];
    if (M[j] <= 0 || N[j] <= 0 || K <= 0) continue;
    const WgradPlan p = wgrad_plan(M[j], N[j], K);
    for (int c = 0; c < 7; ++c) if (variants[c][0] == p.TN && variants[c][1] == p.TK) cls_of[j] = c;
    if (cls_of[j] >= 0) class_flops[cls_of[j]] += (double)M[j] * N[j] * K;
  }
  auto plan_of = [&](int j) {
    const int K = k0[j] + k1[j];
    const int c = cls_of[j];
    const bool big = variants[c][0] * variants[c][1] >= 16;
    int64_t share = (int64_t)((big ? budget_big : budget_small) * ((double)M[j] * N[j] * K / class_flops[c]));
    if (share < 64) share = 64;
    const int64_t cap = big ? 2048 : 8192;  // never finer than the single-job plan: ws[j] is sized for that one
    if (share > cap) share = cap;
    return wgrad_plan(M[j], N[j], K, share);
  };
  // pass 1: the GEMMs, one launch per tile class (and per 16 jobs); their reduces are queued and launched afterwards
  for (int v = 0; v < 7; ++v) {
    WgradBatch b;
    b.njobs = 0;
    unsigned total = 0;
    auto flush = [&]() {
      if (b.njobs == 0) return;
      b.wg_start[b.njobs] = total;
      for (int i = b.njobs + 1; i <= WGRAD_BATCH_MAX; ++i) b.wg_start[i] = total;
      for (int i = b.njobs; i < WGRAD_BATCH_MAX; ++i) { b.g[i] = b.g[0]; b.gx[i] = 1; b.gy[i] = 1; b.gz[i] = 1; }
      const int TN = variants[v][0], TK = variants[v][1];
      if (TN == 4 && TK == 4) launch_wgrad2_batch<4, 4>(b, total, st);
      else if (TN == 4 && TK == 2) launch_wgrad2_batch<4, 2>(b, total, st);
      else if (TN == 2 && TK == 4) launch_wgrad2_batch<2, 4>(b, total, st);
      else if (TN == 2 && TK == 2) launch_wgrad2_batch<2, 2>(b, total, st);
      else if (TN == 1 && TK == 2) launch_wgrad2_batch<1, 2>(b, total, st);
      else if (TN == 2 && TK == 1) launch_wgrad2_batch<2, 1>(b, total, st);
      else launch_wgrad2_batch<1, 1>(b, total, st);
      b.njobs = 0; total = 0;
    };
    for (int j = 0; j < njobs; ++j) {
      const int K = k0[j] + k1[j];
      if (M[j] <= 0 || N[j] <= 0 || K <= 0) continue;  // (accumulate: nothing to add)
      const int cls = cls_of[j];
      if (cls < 0) {  // (4,1) / (1,4): rare shapes, launched on their own
        if (v == 0) {
          const int rc = m3d_linear_wgrad_f32(dz[j], lddz[j], x0[j], ldx0[j], x0_rows[j], k0[j], x1[j], ldx1[j], k1[j], M[j],
                                              N[j], dw[j], lddw[j], 1 | (accumulate & M3D_IO_BF16), ws[j], stream);
          if (rc != M3D_OK) return rc;
        }
        continue;
      }
      if (cls != v) continue;
      const WgradPlan p = plan_of(j);
      if (!dz[j] || k0[j] < 1 || !x0[j] || (k1[j] > 0 && !x1[j]) || !dw[j] || (p.S > 1 && !ws[j])) return M3D_ERR_INVALID;
      const int64_t lim = (int64_t)M3D_BUF_BYTES - 64;
      if (M[j] * lddz[j] * 4 > lim || M[j] * ldx0[j] * 4 > lim || (k1[j] > 0 && M[j] * ldx1[j] * 4 > lim))
        return M3D_ERR_UNSUPPORTED;
      WgradArgs& g = b.g[b.njobs];
      g.dz = dz[j]; g.lddz = lddz[j]; g.x0 = x0[j]; g.ldx0 = ldx0[j]; g.rows = x0_rows[j]; g.k0 = k0[j]; g.x1 = x1[j];
      g.ldx1 = ldx1[j]; g.k1 = k1[j]; g.M = M[j]; g.N = N[j]; g.dw = dw[j]; g.lddw = lddw[j]; g.accumulate = 1;
      g.ws = (float*)ws[j]; g.S = (int)p.S; g.steps_per_split = p.spw;
      g.io = (accumulate & M3D_IO_BF16) ? 1 : 0;
      g.vec = wgrad_vec_ok(g, p.TN, p.TK);
      g.bf16 = ((accumulate >> 8) & 1) && g.vec && p.TN * p.TK >= 8;  // the matrix-bound (deep) layers only
      b.wg_start[b.njobs] = total;
      b.gx[b.njobs] = (unsigned)p.wgs; b.gy[b.njobs] = (unsigned)p.by; b.gz[b.njobs] = (unsigned)p.bz;
      total += (unsigned)(p.wgs * p.by * p.bz);
      if (WGRAD_XCD) total = (total + 7u) & ~7u;  // every job starts on XCD 0 (see wgrad2_batch_kernel)
      if (++b.njobs == WGRAD_BATCH_MAX) flush();
    }
    flush();
  }
  // pass 2: the partial-sum reduces (stream order puts them behind every GEMM launch)
  for (int j = 0; j < njobs; ++j) {
    const int K = k0[j] + k1[j];
    if (M[j] <= 0 || N[j] <= 0 || K <= 0) continue;
    if (cls_of[j] < 0) continue;
    const WgradPlan p = plan_of(j);
    if (p.S <= 1) continue;
    const int E = N[j] * K;
    const int gx = (E + 255) / 256;
    int gy = (int)(p.S / 16);
    if (gy > 2048 / gx) gy = 2048 / gx;
    if (gy < 1) gy = 1;
    const int i = rb.njobs;
    rb.ws[i] = (const float*)ws[j]; rb.dw[i] = dw[j]; rb.lddw[i] = lddw[j]; rb.S[i] = (int)p.S; rb.N[i] = N[j]; rb.K[i] = K;
    rb.gx[i] = (unsigned)gx; rb.gy[i] = (unsigned)gy; rb.wg_start[i] = rtotal;
    rtotal += (unsigned)(gx * gy);
    if (++rb.njobs == WREDUCE_BATCH_MAX) flush_reduce();
  }
  flush_reduce();
  M3D_CHECK_LAUNCH();
  return M3D_OK;
}
