// Train-mode BatchNorm1d pieces around the SharedMLP GEMMs (gfx950).
//
// Restates torch.nn.BatchNorm1d(momentum=0.01, eps=1e-6) as wrapped by PyG's BatchNorm inside SharedMLP
// (/root/reference/myria3d/models/modules/pyg_randla_net.py:92-109): batch statistics over ALL rows of the
// batch, biased variance for normalisation, unbiased for running_var.  The GEMM epilogue (gemm.hip) has
// already accumulated per-column sum / sum-of-squares of the raw Linear output in fp64; here:
//   m3d_bn_finalize   -> mean, invstd, folded (scale, shift), running-stat update
//   m3d_bn_apply      -> y = act(z*scale + shift [+ z2*scale2 + shift2])     (the [+...] is the block's
//                        residual: LeakyReLU(mlp2(x) + shortcut(x)), pyg_randla_net.py:186-187)
//   m3d_bn_stats_apply -> the two above in ONE launch from slot-mode statistics (what the training step uses), with the
//                        classifier's dropout on the way out
//   m3d_bn_bwd        -> column sums (bn_bwd_reduce_kernel) and, unless the input-gradient GEMM's prologue does it
//                        (m3d_bn_dgrad_f32), the gradient w.r.t. the raw Linear output(s) (bn_bwd_apply_kernel); gamma, beta.
// All HBM-bound elementwise / column-reduction kernels: float4 accesses, fp64 column accumulators, straight-line row
// loops with 4-8 rows of every operand in flight (no load behind a branch inside a loop: DESIGN.md section 5).
#include <stdlib.h>
#include "m3d_common.h"
#include "../../include/m3d_hip.h"

// one wave per column: lanes sum the per-workgroup partial rows written by the GEMM ([parts][2][N], fp64)
__global__ __launch_bounds__(64) void bn_finalize_kernel(const double* __restrict__ part, int parts, double count,
                                                         const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float eps, float momentum,
                                                         float* running_mean, float* running_var, float* scale,
                                                         float* shift, float* mean_out, float* invstd_out, int N) {
  const int n = blockIdx.x, lane = threadIdx.x;
  // 4 independent accumulator pairs: 8 loads in flight per lane (a plain loop waits for every load in turn)
  double s4[4] = {0.0, 0.0, 0.0, 0.0}, q4[4] = {0.0, 0.0, 0.0, 0.0};
  int p = lane;
  for (; p + 192 < parts; p += 256) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      s4[u] += part[((size_t)(p + 64 * u) * 2 + 0) * N + n];
      q4[u] += part[((size_t)(p + 64 * u) * 2 + 1) * N + n];
    }
  }
  for (; p < parts; p += 64) {
    s4[0] += part[((size_t)p * 2 + 0) * N + n];
    q4[0] += part[((size_t)p * 2 + 1) * N + n];
  }
  double s = wave_sum_d((s4[0] + s4[1]) + (s4[2] + s4[3]));
  double q = wave_sum_d((q4[0] + q4[1]) + (q4[2] + q4[3]));
  if (lane != 0) return;
  double mean = s / count;
  double var = q / count - mean * mean;
  if (var < 0.0) var = 0.0;
  double invstd = 1.0 / sqrt(var + (double)eps);
  float g = gamma ? gamma[n] : 1.f, b = beta ? beta[n] : 0.f;
  double sc = (double)g * invstd;
  scale[n] = (float)sc;
  shift[n] = (float)((double)b - mean * sc);
  if (mean_out) mean_out[n] = (float)mean;
  if (invstd_out) invstd_out[n] = (float)invstd;
  if (running_mean) running_mean[n] = (float)((1.0 - momentum) * (double)running_mean[n] + momentum * mean);
  if (running_var) {
    double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
    running_var[n] = (float)((1.0 - momentum) * (double)running_var[n] + momentum * unbiased);
  }
}

extern "C" int m3d_bn_finalize(const double* stat_part, int32_t parts, int64_t count, const float* gamma,
                               const float* beta, float eps, float momentum, float* running_mean,
                               float* running_var, float* scale, float* shift, float* mean_out, float* invstd_out,
                               int32_t N, void* stream) {
  if (!stat_part || parts < 1 || !scale || !shift || N < 0 || count < 1) return M3D_ERR_INVALID;
  if (N == 0) return M3D_OK;
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(N), dim3(64), 0, (hipStream_t)stream, stat_part, parts, (double)count,
                     gamma, beta, eps, momentum, running_mean, running_var, scale, shift, mean_out, invstd_out, N);
  M3D_CHECK_LAUNCH();
  return M3D_OK;
}

// eval-mode fold: scale = gamma / sqrt(running_var + eps), shift = beta - running_mean * scale
__global__ void bn_fold_eval_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                    const float* __restrict__ rm, const float* __restrict__ rv, float eps,
                                    float* scale, float* shift, int N) {
  int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  double invstd = 1.0 / sqrt((double)rv[n] + (double)eps);
  double sc = (double)gamma[n] * invstd;
  scale[n] = (float)sc;
  shift[n] = (float)((double)beta[n] - (double)rm[n] * sc);
}

extern "C" int m3d_bn_fold_eval(const float* gamma, const float* beta, const float* running_mean,
                                const float* running_var, float eps, float* scale, float* shift, int32_t N,
                                void* stream) {
  if (!gamma || !beta || !running_mean || !running_var || !scale || !shift || N < 0) return M3D_ERR_INVALID;
  if (N == 0) return M3D_OK;
  hipLaunchKernelGGL(bn_fold_eval_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, gamma, beta,
                     running_mean, running_var, eps, scale, shift, N);
  M3D_CHECK_LAUNCH();
  return M3D_OK;
}

// ------------------------------------------------------------------------------------------
// y = act(z*scale + shift [+ z2*scale2 + shift2]);  N % 4 == 0, rows contiguous (ld == N)
// ------------------------------------------------------------------------------------------
// H: z, z2 and y hold bf16 (M3D_IO_BF16)
template <bool H>
__global__ __launch_bounds__(256) void bn_apply_kernel(const void* __restrict__ z, const float* __restrict__ scale,
                                                       const float* __restrict__ shift,
                                                       const void* __restrict__ z2,
                                                       const float* __restrict__ scale2,
                                                       const float* __restrict__ shift2, int act, float slope,
                                                       void* __restrict__ y, int64_t total4, int N4) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (int64_t)gridDim.x * 256) {
    int c = (int)(i % N4) * 4;
    float4 v = io_load4<H>(z, 4 * (size_t)i);
    float4 sc = *(const float4*)(scale + c), sh = *(const float4*)(shift + c);
    float4 u = make_float4(v.x * sc.x + sh.x, v.y * sc.y + sh.y, v.z * sc.z + sh.z, v.w * sc.w + sh.w);
    if (z2) {
      float4 w = io_load4<H>(z2, 4 * (size_t)i);
      float4 s2 = *(const float4*)(scale2 + c), h2 = *(const float4*)(shift2 + c);
      u.x += w.x * s2.x + h2.x; u.y += w.y * s2.y + h2.y; u.z += w.z * s2.z + h2.z; u.w += w.w * s2.w + h2.w;
    }
    if (act) { u.x = lrelu(u.x, slope); u.y = lrelu(u.y, slope); u.z = lrelu(u.z, slope); u.w = lrelu(u.w, slope); }
    io_store4<H>(y, 4 * (size_t)i, u);
  }
}

extern "C" int m3d_bn_apply(const float* z, const float* scale, const float* shift, const float* z2,
                            const float* scale2, const float* shift2, int32_t act, float slope, float* y, int64_t M,
                            int32_t N, void* stream) {
  if (M < 0 || N < 0) return M3D_ERR_INVALID;
  if (M == 0 || N == 0) return M3D_OK;
  if (!z || !scale || !shift || !y) return M3D_ERR_INVALID;
  if (z2 && (!scale2 || !shift2)) return M3D_ERR_INVALID;
  if (N % 4) return M3D_ERR_UNSUPPORTED;
  int64_t total4 = M * (N / 4);
  int64_t gx = m3d_cdiv(total4, 256 * 4);
  if (gx > 4096) gx = 4096;
  if (gx < 1) gx = 1;
  // act: bit 0 = LeakyReLU, M3D_IO_BF16 = z, z2 and y hold bf16
  if (act & M3D_IO_BF16)
    hipLaunchKernelGGL(bn_apply_kernel<true>, dim3((unsigned)gx), dim3(256), 0, (hipStream_t)stream, (const void*)z, scale,
                       shift, (const void*)z2, scale2, shift2, act & 1, slope, (void*)y, total4, N / 4);
  else
    hipLaunchKernelGGL(bn_apply_kernel<false>, dim3((unsigned)gx), dim3(256), 0, (hipStream_t)stream, (const void*)z, scale,
                       shift, (const void*)z2, scale2, shift2, act & 1, slope, (void*)y, total4, N / 4);
  M3D_CHECK_LAUNCH();
  return M3D_OK;
}

// ------------------------------------------------------------------------------------------
// m3d_bn_stats_apply: m3d_bn_finalize + m3d_bn_apply in ONE launch (slot-mode statistics of m3d_gemm_f32).
// The training step is a ~400-node chain in which a tiny kernel costs >= 5 us: instead of a finalize kernel per
// BatchNorm, every thread of the apply kernel sums the few slots of ITS four columns (fp64) and derives scale / shift
// itself; block 0 also stores mean / invstd / scale / shift for the backward pass and updates the running statistics.
// N a power of two (all of this network's widths) so that a thread keeps its columns over the grid-stride loop.
// ------------------------------------------------------------------------------------------
struct BnStatsArgs {
  const double* slots; const float* gamma; const float* beta; float* running_mean; float* running_var;
  float* scale; float* shift; float* mean; float* invstd;
};
struct BnStatsApplyArgs {
  BnStatsArgs b1, b2;  // b2.slots == nullptr: no residual branch
  int nslots; double count; float eps, momentum;
  const void* z; const void* z2; int act; float slope; void* y; int64_t total4; int N;  // (fp32, or bf16: template flag H)
  DropArgs drop;  // thr16 != 0: y = dropout(activation) (m3d_common.h)
};

#ifndef BN_BWD_RPT
#define BN_BWD_RPT 0  // rows per thread of the backward column sums; 0: by layer size (see bn_bwd_plan)
#endif
#define BN_MAXN 1024  // widest BatchNorm the fused kernels keep in LDS (this network: 512)

// scale / shift of column n from the slot sums (fp64); `writer`: also store the backward inputs and update the
// running statistics (one block does it)
__device__ __forceinline__ void bn_stats_col(const BnStatsApplyArgs& a, const BnStatsArgs& b, int n, bool writer,
                                             float& sc, float& sh) {
  // everything this column reads is requested up front (slot rows, affine parameters, the writer's running statistics)
  // (branch-free: an absent operand reads scale[n] instead and drops it — a load under its own branch is waited for there)
  const float gam0 = (b.gamma ? b.gamma : b.scale)[n], bet0 = (b.beta ? b.beta : b.scale)[n];
  const float rm = (writer && b.running_mean ? b.running_mean : b.scale)[n];
  const float rv = (writer && b.running_var ? b.running_var : b.scale)[n];
  const float gam = b.gamma ? gam0 : 1.f, bet = b.beta ? bet0 : 0.f;
  double sq[2];
  slot_sums<2>(b.slots + n, a.nslots, 2 * (size_t)a.N, (size_t)a.N, sq);
  const double mean = sq[0] / a.count;
  double var = sq[1] / a.count - mean * mean;
  if (var < 0.0) var = 0.0;
  const double invstd = 1.0 / sqrt(var + (double)a.eps);
  const double scd = (double)gam * invstd;
  sc = (float)scd;
  sh = (float)((double)bet - mean * scd);
  if (writer) {
    b.scale[n] = sc; b.shift[n] = sh;
    if (b.mean) b.mean[n] = (float)mean;
    if (b.invstd) b.invstd[n] = (float)invstd;
    if (b.running_mean) b.running_mean[n] = (float)((1.0 - a.momentum) * (double)rm + a.momentum * mean);
    if (b.running_var) {
      const double unbiased = a.count > 1.0 ? var * a.count / (a.count - 1.0) : var;
      b.running_var[n] = (float)((1.0 - a.momentum) * (double)rv + a.momentum * unbiased);
    }
  }
}

template <bool B2, bool H = false>
__global__ __launch_bounds__(256) void bn_stats_apply_kernel(BnStatsApplyArgs a) {
  // every block derives the N column constants ONCE (one thread per column, all slot rows at once, fp64) into LDS
  __shared__ float s_sc[BN_MAXN], s_sh[BN_MAXN], s_sc2[BN_MAXN], s_sh2[BN_MAXN];
  const bool writer = blockIdx.x == 0;
  // four float4 per thread and trip, all requested before the first is used (B2 is a template parameter: straight-line
  // code); the first four BEFORE the column pass — the pass is a dependent chain of its own (slot loads -> fp64 divide /
  // sqrt -> LDS) and the small layers are nothing but latency
  constexpr int U = 4;
  const int64_t stride = (int64_t)gridDim.x * 256;
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  float4 v[U], w[U];
  auto load = [&](int64_t at) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t j = at + u * stride < a.total4 ? at + u * stride : a.total4 - 1;  // (clamped: branch-free loads)
      v[u] = io_load4<H>(a.z, 4 * (size_t)j);
      if constexpr (B2) w[u] = io_load4<H>(a.z2, 4 * (size_t)j);
    }
  };
  load(i);
  for (int n = threadIdx.x; n < a.N; n += 256) {
    bn_stats_col(a, a.b1, n, writer, s_sc[n], s_sh[n]);
    if constexpr (B2) bn_stats_col(a, a.b2, n, writer, s_sc2[n], s_sh2[n]);
  }
  __syncthreads();
  const int N4 = a.N / 4;
  if (i >= a.total4) return;
  const int c = (int)(i % N4) * 4;  // fixed for this thread: the grid stride is a multiple of N4 (checked by the host)
  const float4 sc = *(const float4*)&s_sc[c], sh = *(const float4*)&s_sh[c];
  float4 s2 = make_float4(0, 0, 0, 0), h2 = make_float4(0, 0, 0, 0);
  if constexpr (B2) { s2 = *(const float4*)&s_sc2[c]; h2 = *(const float4*)&s_sh2[c]; }
  const uint32_t dkey = a.drop.thr16 ? drop_key(a.drop) : 0u;
  if (a.drop.thr16 && a.drop.snap && blockIdx.x == 0 && threadIdx.x == 0) *a.drop.snap = a.drop.counter[0];  // (M3DDropout::snapshot)
  while (true) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t j = i + u * stride;
      if (j >= a.total4) break;
      float4 o = make_float4(v[u].x * sc.x + sh.x, v[u].y * sc.y + sh.y, v[u].z * sc.z + sh.z, v[u].w * sc.w + sh.w);
      if constexpr (B2) {
        o.x += w[u].x * s2.x + h2.x; o.y += w[u].y * s2.y + h2.y; o.z += w[u].z * s2.z + h2.z; o.w += w[u].w * s2.w + h2.w;
      }
      if (a.act) { o.x = lrelu(o.x, a.slope); o.y = lrelu(o.y, a.slope); o.z = lrelu(o.z, a.slope); o.w = lrelu(o.w, a.slope); }
      if (a.drop.thr16) {
        const float4 m = drop_mul4(dkey, drop_index(a.drop, j / N4, (int)(j % N4), N4), a.drop.thr16, a.drop.scale);
        o.x *= m.x; o.y *= m.y; o.z *= m.z; o.w *= m.w;
      }
      io_store4<H>(a.y, 4 * (size_t)j, o);
    }
    i += U * stride;
    if (i >= a.total4) break;
    load(i);
  }
}

extern "C" int m3d_bn_stats_apply(const double* slots, int32_t nslots, int64_t count, const float* gamma,
                                  const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                                  float* scale, float* shift, float* mean_out, float* invstd_out, const float* z,
                                  const double* slots2, const float* gamma2, const float* beta2, float* running_mean2,
                                  float* running_var2, float* scale2, float* shift2, float* mean_out2,
                                  float* invstd_out2, const float* z2, int32_t act, float slope, float* y, int64_t M,
                                  int32_t N, const M3DDropout* drop, void* stream) {
  if (M < 0 || N < 0 || nslots < 1 || count < 1) return M3D_ERR_INVALID;
  if (M == 0 || N == 0) return M3D_OK;
  if (!slots || !scale || !shift || !z || !y) return M3D_ERR_INVALID;
  if (slots2 && (!scale2 || !shift2 || !z2)) return M3D_ERR_INVALID;
  if ((N % 4) || (N & (N - 1)) || N > BN_MAXN) return M3D_ERR_UNSUPPORTED;  // power-of-two widths (a thread keeps its columns)
  BnStatsApplyArgs a;
  a.b1 = {slots, gamma, beta, running_mean, running_var, scale, shift, mean_out, invstd_out};
  a.b2 = {slots2, gamma2, beta2, running_mean2, running_var2, scale2, shift2, mean_out2, invstd_out2};
  a.nslots = nslots; a.count = (double)count; a.eps = eps; a.momentum = momentum;
  a.z = z; a.z2 = z2; a.act = act & 1; a.slope = slope; a.y = y;  // act: bit 0 = LeakyReLU, M3D_IO_BF16 = z, z2, y hold bf16
  a.total4 = M * (N / 4); a.N = N;
  a.drop = drop_args(drop);
  int64_t gx = m3d_cdiv(a.total4, 256 * 8);  // ~8 float4 per thread: the per-block column pass is amortised
  if (gx > 1024) gx = 1024;
  if (gx < 1) gx = 1;
  // stride = gx * 256 must be a multiple of N4 (a power of two <= 256 divides 256)
  const bool h = (act & M3D_IO_BF16) != 0;
  if (slots2 && h) hipLaunchKernelGGL((bn_stats_apply_kernel<true, true>), dim3((unsigned)gx), dim3(256), 0, (hipStream_t)stream, a);
  else if (slots2) hipLaunchKernelGGL((bn_stats_apply_kernel<true, false>), dim3((unsigned)gx), dim3(256), 0, (hipStream_t)stream, a);
  else if (h) hipLaunchKernelGGL((bn_stats_apply_kernel<false, true>), dim3((unsigned)gx), dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL((bn_stats_apply_kernel<false, false>), dim3((unsigned)gx), dim3(256), 0, (hipStream_t)stream, a);
  M3D_CHECK_LAUNCH();
  return M3D_OK;
}

// ------------------------------------------------------------------------------------------
// backward, pass 1: sums[0][n] = sum_m dact, sums[1][n] = sum_m dact*zhat, sums[2][n] = sum_m dact*zhat2
//   u = z*scale+shift (+ z2*scale2+shift2); dact = dy * (act ? (u>0 ? 1 : slope) : 1)
//   zhat = (z - mean)*invstd
// ------------------------------------------------------------------------------------------
struct BnBwdArgs {
  const float* dy; const float* z; const float* scale; const float* shift; const float* mean; const float* invstd;
  const float* z2; const float* scale2; const float* shift2; const float* mean2; const float* invstd2;
  int act; float slope; int64_t M; int N;
  double* sums;  // [3][N]
  float* dz; float* dz2; float* dgamma; float* dbeta; float* dgamma2; float* dbeta2;
  int acc_pg;  // != 0: add into dgamma/dbeta (gradient sinks) instead of overwriting
  int nslots;  // > 0: sums is a pre-zeroed [nslots][3][N] table (reduce adds, apply sums; no finalize kernel)
  DropArgs drop; uint32_t dkey;  // thr16 != 0: the layer's output went through dropout: dy is masked first (dkey: set in-kernel)
};

// grid (row blocks, column passes): block (x, y) owns rows [x*rows_per_block, ...) and 256/rpp float4 column groups.
// Per-thread fp32 partials over 8-16 rows, fp64 across threads (LDS), one partial row per block (slot atomics, or
// part[x][3][N] summed by bn_bwd_finalize_kernel).
// The row loop is straight-line code (Z2 / DROP are template parameters, the column constants are loaded in front of it,
// rows past the end re-read the first row and count as 0): the loads of eight rows — dy, z (, z2) — are in flight together.
// With the operands' null checks and the activation's scale / shift loads inside the loop every row was two to three
// DEPENDENT round trips (z, dy -> wait -> scale, shift -> wait -> ...), 16-32 of them per thread: the 9-27 us of the launches.
// IO (M3D_IO_* >> 12): bit 0 = z, z2 (and the apply kernel's dz, dz2) hold bf16; dy too unless bit 1 is set (fp32 dy)
template <bool Z2, bool DROP, int IO = 0>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(BnBwdArgs a, int rows_per_block, double* __restrict__ part) {
  constexpr bool SH = (IO & 1) != 0, DH = (IO & 1) != 0 && (IO & 2) == 0;
  __shared__ double red[256 * 12];
  uint32_t dkey = 0u;
  if constexpr (DROP) dkey = drop_key(a.drop);
  const int tid = threadIdx.x;
  const int N4 = a.N / 4;
  const int CG = N4 < 256 ? N4 : 256;  // column groups per block
  const int rpp = 256 / CG;            // rows per pass
  const int cg = tid % CG, rl = tid / CG;
  const int c4 = blockIdx.y * CG + cg;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < a.M ? r0 + rows_per_block : a.M;
  float s[12];
#pragma unroll
  for (int j = 0; j < 12; ++j) s[j] = 0.f;
  if (rl < rpp && c4 < N4) {
    const int c = c4 * 4;
    const float4 zero4 = make_float4(0, 0, 0, 0);
    const float4 mu = *(const float4*)(a.mean + c), is = *(const float4*)(a.invstd + c);
    const float4 sc = *(const float4*)(a.scale + c), sh = *(const float4*)(a.shift + c);
    float4 mu2 = zero4, is2 = zero4, sc2 = zero4, sh2 = zero4;
    if constexpr (Z2) {
      mu2 = *(const float4*)(a.mean2 + c); is2 = *(const float4*)(a.invstd2 + c);
      sc2 = *(const float4*)(a.scale2 + c); sh2 = *(const float4*)(a.shift2 + c);
    }
    constexpr int U = 8;
    for (int64_t r = r0 + rl; r < r1; r += (int64_t)U * rpp) {
      float4 zv[U], gv[U], z2v[U];
      int64_t di[U];
      bool ok[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t rr = r + (int64_t)u * rpp;
        ok[u] = rr < r1;
        const int64_t rc = ok[u] ? rr : r;
        const int64_t i = rc * N4 + c4;
        zv[u] = io_load4<SH>(a.z, 4 * (size_t)i);
        gv[u] = io_load4<DH>(a.dy, 4 * (size_t)i);
        z2v[u] = zero4;
        if constexpr (Z2) z2v[u] = io_load4<SH>(a.z2, 4 * (size_t)i);
        di[u] = 0;
        if constexpr (DROP) di[u] = drop_index(a.drop, rc, c4, N4);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        float4 g = gv[u];
        const float4 zz = zv[u], z2 = z2v[u];
        if constexpr (DROP) {
          const float4 m = drop_mul4(dkey, di[u], a.drop.thr16, a.drop.scale);
          g.x *= m.x; g.y *= m.y; g.z *= m.z; g.w *= m.w;
        }
        if (a.act) {
          float4 v = make_float4(zz.x * sc.x + sh.x, zz.y * sc.y + sh.y, zz.z * sc.z + sh.z, zz.w * sc.w + sh.w);
          if constexpr (Z2) {
            v.x += z2.x * sc2.x + sh2.x; v.y += z2.y * sc2.y + sh2.y; v.z += z2.z * sc2.z + sh2.z; v.w += z2.w * sc2.w + sh2.w;
          }
          g.x *= v.x > 0.f ? 1.f : a.slope; g.y *= v.y > 0.f ? 1.f : a.slope;
          g.z *= v.z > 0.f ? 1.f : a.slope; g.w *= v.w > 0.f ? 1.f : a.slope;
        }
        if (!ok[u]) g = zero4;
        s[0] += g.x; s[1] += g.y; s[2] += g.z; s[3] += g.w;
        s[4] += g.x * ((zz.x - mu.x) * is.x); s[5] += g.y * ((zz.y - mu.y) * is.y);
        s[6] += g.z * ((zz.z - mu.z) * is.z); s[7] += g.w * ((zz.w - mu.w) * is.w);
        if constexpr (Z2) {
          s[8] += g.x * ((z2.x - mu2.x) * is2.x); s[9] += g.y * ((z2.y - mu2.y) * is2.y);
          s[10] += g.z * ((z2.z - mu2.z) * is2.z); s[11] += g.w * ((z2.w - mu2.w) * is2.w);
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 12; ++j) red[j * 256 + tid] = (double)s[j];
  __syncthreads();
  // sum over the row lanes (rl) as a tree with every thread at work: narrow layers have 2-4 column groups and 64-128 row
  // lanes, and the column groups' first threads used to walk all of them one after the other (1 536 dependent LDS reads:
  // 34 us for a 204 800 x 8 layer whose data streams in 3 us)
  {
    int top = 1;
    while (top < rpp) top <<= 1;
    for (int stride = top >> 1; stride >= 1; stride >>= 1) {
      if (rl < stride && rl + stride < rpp) {
#pragma unroll
        for (int j = 0; j < 12; ++j)
          if (j < 8 || a.z2) red[j * 256 + tid] += red[j * 256 + tid + stride * CG];
      }
      __syncthreads();
    }
  }
  if (rl == 0 && c4 < N4) {
    double* dst = a.nslots > 0 ? a.sums + (size_t)(blockIdx.x % a.nslots) * 3 * a.N : part + (size_t)blockIdx.x * 3 * a.N;
#pragma unroll
    for (int j = 0; j < 12; ++j) {
      const double v = (j < 8 || a.z2) ? red[j * 256 + cg] : 0.0;
      double* d = &dst[(size_t)(j / 4) * a.N + c4 * 4 + (j & 3)];
      if (a.nslots > 0) {
        if (j < 8 || a.z2) atomicAdd(d, v);
      } else {
        *d = v;
      }
    }
  }
}

// one wave per (which, column): sums[which][n] = sum over the partial rows; also the parameter gradients
// dbeta = s1, dgamma = s2 (dgamma2 = s3)
__global__ __launch_bounds__(64) void bn_bwd_finalize_kernel(BnBwdArgs a, const double* __restrict__ part, int parts) {
  const int n = blockIdx.x, which = blockIdx.y, lane = threadIdx.x;
  double v4[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  int p = lane;
  for (; p + 448 < parts; p += 512) {
#pragma unroll
    for (int u = 0; u < 8; ++u) v4[u] += part[((size_t)(p + 64 * u) * 3 + which) * a.N + n];
  }
  for (; p < parts; p += 64) v4[0] += part[((size_t)p * 3 + which) * a.N + n];
  double v = wave_sum_d(((v4[0] + v4[1]) + (v4[2] + v4[3])) + ((v4[4] + v4[5]) + (v4[6] + v4[7])));
  if (lane != 0) return;
  a.sums[(size_t)which * a.N + n] = v;
  if (which == 0) {
    if (a.dbeta) a.dbeta[n] = (a.acc_pg ? a.dbeta[n] : 0.f) + (float)v;
    if (a.z2 && a.dbeta2) a.dbeta2[n] = (a.acc_pg ? a.dbeta2[n] : 0.f) + (float)v;
  } else if (which == 1) {
    if (a.dgamma) a.dgamma[n] = (a.acc_pg ? a.dgamma[n] : 0.f) + (float)v;
  } else if (a.z2 && a.dgamma2) {
    a.dgamma2[n] = (a.acc_pg ? a.dgamma2[n] : 0.f) + (float)v;
  }
}

// pass 2: dz = scale*(dact - s1/M - zhat*s2/M);  dz2 likewise with its own zhat2/s2';  dgamma = s2, dbeta = s1
struct BnCol {  // per-column constants of 4 consecutive columns
  float4 sc, sh, mu, is, m1, m2;
};

__device__ __forceinline__ float4 bn_dz(const BnCol& k, float4 g, float4 zv) {
  float4 o;
  o.x = k.sc.x * (g.x - k.m1.x - (zv.x - k.mu.x) * k.is.x * k.m2.x);
  o.y = k.sc.y * (g.y - k.m1.y - (zv.y - k.mu.y) * k.is.y * k.m2.y);
  o.z = k.sc.z * (g.z - k.m1.z - (zv.z - k.mu.z) * k.is.z * k.m2.z);
  o.w = k.sc.w * (g.w - k.m1.w - (zv.w - k.mu.w) * k.is.w * k.m2.w);
  return o;
}

template <bool Z2, int IO = 0>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(BnBwdArgs a) {
  constexpr bool SH = (IO & 1) != 0, DH = (IO & 1) != 0 && (IO & 2) == 0;
  if (a.drop.thr16) a.dkey = drop_key(a.drop);
  const int N4 = a.N / 4;
  const int64_t total4 = a.M * N4;
  const double invM = 1.0 / (double)a.M;
  const int64_t stride = (int64_t)gridDim.x * 256;
  // the grid stride is a multiple of N4 for every power-of-two width (all of this network's): a thread then stays on
  // the same 4 columns and their constants (incl. the fp64 column sums) are loaded once, not per element
  const bool fixed = (stride % N4) == 0;
  BnCol k1, k2;
  // slot mode: every block sums the slot rows of all N columns ONCE into LDS (one thread per column); block 0 also
  // writes the parameter gradients the finalize kernel used to write (dbeta = s1, dgamma = s2, dgamma2 = s3)
  __shared__ float s_m1[BN_MAXN], s_m2[BN_MAXN], s_m3[BN_MAXN];
  if (a.nslots > 0) {
    for (int n = threadIdx.x; n < a.N; n += 256) {
      // (the gradient sinks' old values are requested before the slot rows, not after them)
      const bool old = blockIdx.x == 0 && a.acc_pg;  // branch-free: what is absent reads scale[n] and counts as 0
      const bool h0 = old && a.dbeta, h1 = old && a.z2 && a.dbeta2, h2 = old && a.dgamma, h3 = old && a.z2 && a.dgamma2;
      const float r0 = (h0 ? a.dbeta : a.scale)[n], r1 = (h1 ? a.dbeta2 : a.scale)[n];
      const float r2 = (h2 ? a.dgamma : a.scale)[n], r3 = (h3 ? a.dgamma2 : a.scale)[n];
      const float ob = h0 ? r0 : 0.f, ob2 = h1 ? r1 : 0.f, og = h2 ? r2 : 0.f, og2 = h3 ? r3 : 0.f;
      double s1, s2, s3 = 0.0;
      if (a.z2) {
        double t[3];
        slot_sums<3>(a.sums + n, a.nslots, 3 * (size_t)a.N, (size_t)a.N, t);
        s1 = t[0]; s2 = t[1]; s3 = t[2];
      } else {
        double t[2];
        slot_sums<2>(a.sums + n, a.nslots, 3 * (size_t)a.N, (size_t)a.N, t);
        s1 = t[0]; s2 = t[1];
      }
      s_m1[n] = (float)(s1 * invM); s_m2[n] = (float)(s2 * invM); s_m3[n] = (float)(s3 * invM);
      if (blockIdx.x == 0) {
        if (a.dbeta) a.dbeta[n] = ob + (float)s1;
        if (a.z2 && a.dbeta2) a.dbeta2[n] = ob2 + (float)s1;
        if (a.dgamma) a.dgamma[n] = og + (float)s2;
        if (a.z2 && a.dgamma2) a.dgamma2[n] = og2 + (float)s3;
      }
    }
    __syncthreads();
  }
  auto load_cols = [&](int c) {
    k1.sc = *(const float4*)(a.scale + c); k1.sh = *(const float4*)(a.shift + c);
    k1.mu = *(const float4*)(a.mean + c); k1.is = *(const float4*)(a.invstd + c);
    if (a.nslots > 0) {
      k1.m1 = *(const float4*)&s_m1[c];
      k1.m2 = *(const float4*)&s_m2[c];
    } else {
      k1.m1 = make_float4((float)(a.sums[c] * invM), (float)(a.sums[c + 1] * invM), (float)(a.sums[c + 2] * invM),
                          (float)(a.sums[c + 3] * invM));
      k1.m2 = make_float4((float)(a.sums[a.N + c] * invM), (float)(a.sums[a.N + c + 1] * invM),
                          (float)(a.sums[a.N + c + 2] * invM), (float)(a.sums[a.N + c + 3] * invM));
    }
    if (a.z2) {
      const size_t o = 2 * (size_t)a.N + c;
      k2.sc = *(const float4*)(a.scale2 + c); k2.sh = *(const float4*)(a.shift2 + c);
      k2.mu = *(const float4*)(a.mean2 + c); k2.is = *(const float4*)(a.invstd2 + c);
      k2.m1 = k1.m1;
      if (a.nslots > 0)
        k2.m2 = *(const float4*)&s_m3[c];
      else
        k2.m2 = make_float4((float)(a.sums[o] * invM), (float)(a.sums[o + 1] * invM), (float)(a.sums[o + 2] * invM),
                            (float)(a.sums[o + 3] * invM));
    }
  };
  // four float4 per thread and trip: dy, z (, z2) of all four requested before the first is used (Z2 is a template
  // parameter: straight-line code; elements past the end re-read the last one and are not stored)
  constexpr int U = 4;
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (fixed && i < total4) load_cols((int)(i % N4) * 4);
  for (; i < total4; i += U * stride) {
    float4 zq[U], gq[U], z2q[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t j = i + u * stride < total4 ? i + u * stride : total4 - 1;
      zq[u] = io_load4<SH>(a.z, 4 * (size_t)j);
      gq[u] = io_load4<DH>(a.dy, 4 * (size_t)j);
      z2q[u] = make_float4(0, 0, 0, 0);
      if constexpr (Z2) z2q[u] = io_load4<SH>(a.z2, 4 * (size_t)j);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t j = i + u * stride;
      if (j >= total4) break;
      if (!fixed) load_cols((int)(j % N4) * 4);
      const float4 zv = zq[u], z2v = z2q[u];
      float4 g = gq[u];
      if (a.drop.thr16) {
        const float4 m = drop_mul4(a.dkey, drop_index(a.drop, j / N4, (int)(j % N4), N4), a.drop.thr16, a.drop.scale);
        g.x *= m.x; g.y *= m.y; g.z *= m.z; g.w *= m.w;
      }
      if (a.act) {
        float4 v = make_float4(zv.x * k1.sc.x + k1.sh.x, zv.y * k1.sc.y + k1.sh.y, zv.z * k1.sc.z + k1.sh.z,
                               zv.w * k1.sc.w + k1.sh.w);
        if constexpr (Z2) {
          v.x += z2v.x * k2.sc.x + k2.sh.x; v.y += z2v.y * k2.sc.y + k2.sh.y;
          v.z += z2v.z * k2.sc.z + k2.sh.z; v.w += z2v.w * k2.sc.w + k2.sh.w;
        }
        g.x *= v.x > 0.f ? 1.f : a.slope; g.y *= v.y > 0.f ? 1.f : a.slope;
        g.z *= v.z > 0.f ? 1.f : a.slope; g.w *= v.w > 0.f ? 1.f : a.slope;
      }
      io_store4<SH>(a.dz, 4 * (size_t)j, bn_dz(k1, g, zv));
      if constexpr (Z2) io_store4<SH>(a.dz2, 4 * (size_t)j, bn_dz(k2, g, z2v));
    }
  }
}

struct BnBwdPlan { int64_t blocks, rows_per_block, passes; };
static BnBwdPlan bn_bwd_plan(int64_t M, int N) {
  BnBwdPlan p;
  const int N4 = N / 4;
  const int CG = N4 < 256 ? N4 : 256;
  const int rpp = 256 / CG;
  p.passes = m3d_cdiv(N4, CG);
  // rows per thread: every block ends in an LDS tree + atomics that cost as much as streaming ~8 rows per thread
  // (BN_BWD_RPT: compile-time A/B knob, tools/opbench.py bnbwd)
  // round 3: 16 for the big layers (>= 1.5 M float4 per operand: 20 -> 16, 36 -> 29, 41 -> 30 us inside the step), 8 below —
  // back-to-back microbenchmarks like 16 everywhere, but inside the dependent chain of the step the small layers lose
  // 3-4 us each to the smaller grid (profiles/r03end_step_timeline.csv vs the run before)
  // (tried: fewer rows per thread on the small layers for >= 512 blocks — 372 -> 439 us inside the step: more block tails
  // and slot atomics cost more than the extra parallelism gives)
  const int rpt = BN_BWD_RPT > 0 ? BN_BWD_RPT : (M * N4 >= 1500000 ? 16 : 8);
  int64_t blocks = m3d_cdiv(M, (int64_t)rpp * rpt);
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  p.rows_per_block = m3d_align(m3d_cdiv(M, blocks), rpp);
  p.blocks = m3d_cdiv(M, p.rows_per_block);
  return p;
}

extern "C" size_t m3d_bn_bwd_workspace_bytes(int64_t M, int32_t N) {
  if (M <= 0 || N <= 0 || (N % 4)) return 0;
  return sizeof(double) * 3 * (size_t)N * (size_t)(1 + bn_bwd_plan(M, N).blocks);
}

extern "C" int m3d_bn_bwd(const float* dy, const float* z, const float* scale, const float* shift, const float* mean,
                          const float* invstd, const float* z2, const float* scale2, const float* shift2,
                          const float* mean2, const float* invstd2, int32_t act, float slope, int64_t M, int32_t N,
                          double* sums_ws, float* dz, float* dz2, float* dgamma, float* dbeta, float* dgamma2,
                          float* dbeta2, int32_t accumulate_param_grads, const M3DDropout* drop, void* stream) {
  if (M < 0 || N < 0) return M3D_ERR_INVALID;
  if (M == 0 || N == 0) return M3D_OK;
  // bit 1 of accumulate_param_grads: pass 1 only (slot mode; pass 2 is the A-prologue of m3d_bn_dgrad_f32)
  const bool reduce_only = (accumulate_param_grads & 2) != 0;
  if (reduce_only && (z2 || !((accumulate_param_grads >> 8) & 0xff))) return M3D_ERR_INVALID;
  if (!dy || !z || !scale || !shift || !mean || !invstd || !sums_ws || (!dz && !reduce_only)) return M3D_ERR_INVALID;
  if (z2 && (!scale2 || !shift2 || !mean2 || !invstd2 || !dz2)) return M3D_ERR_INVALID;
  if (N % 4) return M3D_ERR_UNSUPPORTED;
  BnBwdArgs a;
  a.dy = dy; a.z = z; a.scale = scale; a.shift = shift; a.mean = mean; a.invstd = invstd;
  a.z2 = z2; a.scale2 = scale2; a.shift2 = shift2; a.mean2 = mean2; a.invstd2 = invstd2;
  a.act = act; a.slope = slope; a.M = M; a.N = N; a.sums = sums_ws;
  a.dz = dz; a.dz2 = dz2; a.dgamma = dgamma; a.dbeta = dbeta; a.dgamma2 = dgamma2; a.dbeta2 = dbeta2;
  // accumulate_param_grads: bit 0 = add into dgamma / dbeta; bit 1 = pass 1 only; bits 8.. = slot count of a PRE-ZEROED [slots][3][N] sums_ws
  // (slot mode: two launches instead of three; N must be a power of two)
  a.acc_pg = accumulate_param_grads & 1;
  a.nslots = (accumulate_param_grads >> 8) & 0xff;
  a.drop = drop_args(drop); a.dkey = 0u;
  if (a.nslots > 0 && ((N & (N - 1)) || N > BN_MAXN)) return M3D_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const BnBwdPlan pl = bn_bwd_plan(M, N);
  double* part = sums_ws + 3 * (size_t)N;  // sums_ws = [3][N] totals, then [blocks][3][N] partial rows
  const int N4 = N / 4;
  // bits 16..17 of accumulate_param_grads: (M3D_IO_BF16 | M3D_IO_A32) >> 12 — dy, z, z2, dz, dz2 hold bf16 (dy fp32 with A32)
  const int io = (accumulate_param_grads >> 16) & 3;
  if (io == 2) return M3D_ERR_INVALID;
  {
    const dim3 rgrid((unsigned)pl.blocks, (unsigned)pl.passes);
    const int rpb = (int)pl.rows_per_block;
#define M3D_BN_RED(IO_) \
    do { \
      if (z2 && a.drop.thr16) hipLaunchKernelGGL((bn_bwd_reduce_kernel<true, true, IO_>), rgrid, dim3(256), 0, st, a, rpb, part); \
      else if (z2) hipLaunchKernelGGL((bn_bwd_reduce_kernel<true, false, IO_>), rgrid, dim3(256), 0, st, a, rpb, part); \
      else if (a.drop.thr16) hipLaunchKernelGGL((bn_bwd_reduce_kernel<false, true, IO_>), rgrid, dim3(256), 0, st, a, rpb, part); \
      else hipLaunchKernelGGL((bn_bwd_reduce_kernel<false, false, IO_>), rgrid, dim3(256), 0, st, a, rpb, part); \
    } while (0)
    if (io == 1) M3D_BN_RED(1); else if (io == 3) M3D_BN_RED(3); else M3D_BN_RED(0);
#undef M3D_BN_RED
  }
  if (a.nslots <= 0)
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((unsigned)N, z2 ? 3 : 2), dim3(64), 0, st, a, (const double*)part,
                       (int)pl.blocks);
  if (reduce_only) {
    M3D_CHECK_LAUNCH();
    return M3D_OK;
  }
  int64_t total4 = M * N4;
  int64_t gy = m3d_cdiv(total4, 256 * 4);
  if (gy > 4096) gy = 4096;
  if (gy < 1) gy = 1;
  if (a.nslots > 0) {  // every block pays one pass over the slot table: fewer, fatter blocks
    gy = m3d_cdiv(total4, 256 * 8);
    if (gy > 1024) gy = 1024;
    if (gy < 1) gy = 1;
  }
#define M3D_BN_APP(IO_) \
  do { \
    if (z2) hipLaunchKernelGGL((bn_bwd_apply_kernel<true, IO_>), dim3((unsigned)gy), dim3(256), 0, st, a); \
    else hipLaunchKernelGGL((bn_bwd_apply_kernel<false, IO_>), dim3((unsigned)gy), dim3(256), 0, st, a); \
  } while (0)
  if (io == 1) M3D_BN_APP(1); else if (io == 3) M3D_BN_APP(3); else M3D_BN_APP(0);
#undef M3D_BN_APP
  M3D_CHECK_LAUNCH();
  return M3D_OK;
}
