// fp32 MFMA GEMM for the SharedMLP 1x1 "convolutions" (gfx950).
//
// Replaces the Linear (+BatchNorm +LeakyReLU) stack that PyG's MLP runs for SharedMLP
// (/root/reference/myria3d/models/modules/pyg_randla_net.py:97-109) and torch.nn.Linear
// (pyg_randla_net.py:42,53), forward and both backward GEMMs.
//
//   C[M,N] (+)= A[M,K] * B[N,K]^T          (K = k0 + k1: A may be the concatenation of two column blocks,
//                                            the first optionally row-gathered: FPModule's cat([x[nn], skip]))
// v_mfma_f32_16x16x4_f32 (exact fp32, bitwise an fmaf chain): 256-thread workgroup = 4 waves, 64 x (16*NT)
// output tile, wave w owns rows 16w..16w+15 and NT accumulators; K is walked in chunks of 32 through LDS
// tiles padded to a 34-float row stride (bank = 2*row + k: conflict-free fragment reads).
// Either operand may be given "column-major" (element (r,k) at p[k*ld + r]) so that the same kernel serves
// forward (A row-major, W row-major), dgrad (dZ row-major, W column-major) and wgrad (dZ^T, X^T: both
// column-major, split over the reduction dimension with atomic accumulation).
// Epilogue: + bias, per-column affine (folded eval BatchNorm), LeakyReLU, and — for train-mode BatchNorm —
// per-column sum / sum-of-squares of the raw output accumulated in fp64.
#include "m3d_common.h"
#include "../../include/m3d_hip.h"

#define BM 64
#define BK 32
#define LDS_STRIDE (BK + 2)

#include <stdlib.h>
#include "gemm_common.h"
#ifndef GEMM_LEGACY
#define GEMM_LEGACY 0  // 1: cross-check builds route every product through the LDS-tiled kernel below
#endif

template <int NT>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs g) {
  __shared__ float As[BM * LDS_STRIDE];
  __shared__ float Bs[16 * NT * LDS_STRIDE];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  const int BN = 16 * NT;
  const int n0 = blockIdx.y * BN;
  const int K = g.k0 + g.k1;
  int64_t kbeg = 0, kend = K;
  if (g.splitk > 1) {
    kbeg = (int64_t)blockIdx.z * g.kchunk;
    kend = kbeg + g.kchunk < (int64_t)K ? kbeg + g.kchunk : (int64_t)K;
  }
  const bool a_vec = !g.a_cm && ((g.lda0 & 3) == 0) && ((g.k0 & 3) == 0) && ((((uintptr_t)g.a0) & 15) == 0) &&
                     (g.k1 == 0 || (((g.lda1 & 3) == 0) && ((((uintptr_t)g.a1) & 15) == 0)));
  const bool b_vec = !g.b_cm && ((g.ldb & 3) == 0) && ((((uintptr_t)g.b) & 15) == 0);

  double ssum[NT], ssq[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) { ssum[t] = 0.0; ssq[t] = 0.0; }

  const int64_t mtiles = (g.M + BM - 1) / BM;
  for (int64_t mt = blockIdx.x; mt < mtiles; mt += gridDim.x) {
    const int64_t m0 = mt * BM;
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int64_t kk = kbeg; kk < kend; kk += BK) {
      // ---------------- stage A tile: As[r][k], r < 64, k < 32
      if (!g.a_cm) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          int f = tid + i * 256;
          int r = f >> 3, c4 = (f & 7) * 4;
          int64_t row = m0 + r;
          int64_t kg = kk + c4;
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          if (row < g.M) {
            if (a_vec && kg + 3 < kend) {
              if (kg < g.k0) {
                int64_t rr = g.a0_rows ? (int64_t)g.a0_rows[row] : row;
                v = *(const float4*)(g.a0 + rr * g.lda0 + kg);
              } else {
                v = *(const float4*)(g.a1 + row * g.lda1 + (kg - g.k0));
              }
            } else {
              float tmp[4];
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                int64_t kj = kg + j;
                float e = 0.f;
                if (kj < kend) {
                  if (kj < g.k0) {
                    int64_t rr = g.a0_rows ? (int64_t)g.a0_rows[row] : row;
                    e = g.a0[rr * g.lda0 + kj];
                  } else {
                    e = g.a1[row * g.lda1 + (kj - g.k0)];
                  }
                }
                tmp[j] = e;
              }
              v = make_float4(tmp[0], tmp[1], tmp[2], tmp[3]);
            }
          }
          float* d = &As[r * LDS_STRIDE + c4];
          *(float2*)d = make_float2(v.x, v.y);
          *(float2*)(d + 2) = make_float2(v.z, v.w);
        }
      } else {
        // column-major A: element (row, k) at a0[k*lda0 + row]; coalesce along rows
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          int f = tid + i * 256;
          int k = f >> 4, r4 = (f & 15) * 4;
          int64_t kg = kk + k;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            int64_t row = m0 + r4 + j;
            float e = 0.f;
            if (kg < kend && row < g.M) e = g.a0[kg * g.lda0 + row];
            As[(r4 + j) * LDS_STRIDE + k] = e;
          }
        }
      }
      // ---------------- stage B tile: Bs[n][k], n < BN, k < 32
      if (!g.b_cm) {
        for (int f = tid; f < BN * 8; f += 256) {
          int r = f >> 3, c4 = (f & 7) * 4;
          int n = n0 + r;
          int64_t kg = kk + c4;
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          if (n < g.N) {
            if (b_vec && kg + 3 < kend) {
              v = *(const float4*)(g.b + (int64_t)n * g.ldb + kg);
            } else {
              float tmp[4];
#pragma unroll
              for (int j = 0; j < 4; ++j) tmp[j] = (kg + j < kend) ? g.b[(int64_t)n * g.ldb + kg + j] : 0.f;
              v = make_float4(tmp[0], tmp[1], tmp[2], tmp[3]);
            }
          }
          float* d = &Bs[r * LDS_STRIDE + c4];
          *(float2*)d = make_float2(v.x, v.y);
          *(float2*)(d + 2) = make_float2(v.z, v.w);
        }
      } else {
        // column-major B: element (n, k) at b[k*ldb + n]; coalesce along n
        for (int f = tid; f < BN * BK; f += 256) {
          int k = f / BN, r = f % BN;
          int n = n0 + r;
          int64_t kg = kk + k;
          float e = 0.f;
          if (kg < kend && n < g.N) e = g.b[kg * g.ldb + n];
          Bs[r * LDS_STRIDE + k] = e;
        }
      }
      __syncthreads();
      const float* ap = &As[(wid * 16 + lr) * LDS_STRIDE + lg];
      const float* bp = &Bs[lr * LDS_STRIDE + lg];
#pragma unroll
      for (int s = 0; s < BK / 4; ++s) {
        float a = ap[4 * s];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = mfma16(a, bp[t * 16 * LDS_STRIDE + 4 * s], acc[t]);
      }
      __syncthreads();
    }
    // ---------------- epilogue. C layout: col = lr, row = lg*4 + reg
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int n = n0 + t * 16 + lr;
      const bool ncol = n < g.N;
      const float bia = (ncol && g.bias && blockIdx.z == 0) ? g.bias[n] : 0.f;
      const float sc = (ncol && g.scale) ? g.scale[n] : 1.f;
      const float sh = (ncol && g.shift) ? g.shift[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t row = m0 + wid * 16 + lg * 4 + r;
        if (ncol && row < g.M) {
          float z = acc[t][r] + bia;
          if (g.stat_sum) { ssum[t] += (double)z; ssq[t] += (double)z * (double)z; }
          float y = z * sc + sh;
          if (g.act) y = lrelu(y, g.slope);
          float* cp = g.c + row * g.ldc + n;
          if (g.accumulate) atomicAdd(cp, y); else *cp = y;
        }
      }
    }
  }
  if (g.stat_sum) {
    __shared__ double sred[4][2][16 * NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      double a = xgroup_sum_d(ssum[t]), q = xgroup_sum_d(ssq[t]);
      if (lg == 0) { sred[wid][0][t * 16 + lr] = a; sred[wid][1][t * 16 + lr] = q; }
    }
    __syncthreads();
    if (tid < 2 * BN) {
      int which = tid / BN, col = tid % BN;
      int n = n0 + col;
      if (n < g.N) {
        double v = sred[0][which][col] + sred[1][which][col] + sred[2][which][col] + sred[3][which][col];
        atomicAdd(which ? &g.stat_sumsq[n] : &g.stat_sum[n], v);
      }
    }
  }
}

extern "C" int m3d_gemm_stat_parts(int64_t M, int32_t N, int32_t K) { return m3d_gemm_direct_stat_parts(M, N, K); }

extern "C" int m3d_gemm_f32(const float* a0, int64_t lda0, int32_t a_colmajor, const int32_t* a0_rows, int32_t k0,
                            const float* a1, int64_t lda1, int32_t k1, const float* b, int64_t ldb,
                            int32_t b_colmajor, int64_t M, int32_t N, const float* bias, const float* scale,
                            const float* shift, int32_t act, float slope, double* stat_part, int32_t stat_parts,
                            float* c, int64_t ldc, int32_t accumulate, int32_t splitk, void* stream) {
  if (M < 0 || N < 0 || k0 < 0 || k1 < 0) return M3D_ERR_INVALID;
  if (M == 0 || N == 0) return M3D_OK;
  if (!c || !b || (k0 > 0 && !a0) || (k1 > 0 && !a1)) return M3D_ERR_INVALID;
  if (a_colmajor && (k1 > 0 || a0_rows)) return M3D_ERR_UNSUPPORTED;
  // stat_parts < 0: slot mode — stat_part is a PRE-ZEROED [-stat_parts][2][N] table the workgroups add to
  const int stat_slots = (stat_part && stat_parts < 0) ? -stat_parts : 0;
  if (stat_part && !stat_slots && stat_parts != m3d_gemm_stat_parts(M, N, k0 + k1)) return M3D_ERR_INVALID;
  if (splitk < 1) splitk = 1;
  if (splitk > 1 && (!accumulate || stat_part || scale || shift || act)) return M3D_ERR_INVALID;
  if (k0 + k1 == 0) return M3D_ERR_INVALID;
  GemmArgs g{};
  g.a0 = a0; g.lda0 = lda0; g.a0_rows = a0_rows; g.k0 = k0; g.a1 = a1; g.lda1 = lda1; g.k1 = k1;
  g.a_cm = a_colmajor; g.b = b; g.ldb = ldb; g.b_cm = b_colmajor; g.M = M; g.N = N;
  g.bias = bias; g.scale = scale; g.shift = shift; g.act = act; g.slope = slope;
  g.stat_part = stat_part; g.stat_sum = stat_part; g.stat_sumsq = stat_part ? stat_part + N : nullptr; g.c = c; g.ldc = ldc; g.accumulate = accumulate;
  g.stat_slots = stat_slots;
  g.bf16 = (act >> 8) & 1;  // act: bit 0 = LeakyReLU, bit 8 = bf16 matrix-core operands (deep layers, K % 32 == 0)
  g.io = (act >> 12) & 7;   // M3D_IO_BF16 / M3D_IO_A32 / M3D_IO_C32: activation storage (fragment-direct kernels only)
  g.act = act & 1;
  const int64_t K = (int64_t)k0 + k1;
  int64_t kchunk = m3d_align(m3d_cdiv(K, splitk), BK);
  splitk = (int)m3d_cdiv(K, kchunk);
  g.splitk = splitk; g.kchunk = kchunk;
  {
    // fragment-direct kernels (gemm_direct.hip) cover the network's shapes; this LDS-tiled kernel is the fallback
    // (a -DGEMM_LEGACY=1 build routes everything through it: cross-check builds only)
    if (!GEMM_LEGACY) {
      const int rc = m3d_gemm_direct_try(g, (hipStream_t)stream);
      if (rc != 1) return rc;
    }
  }
  if (accumulate && (scale || shift || g.act)) return M3D_ERR_UNSUPPORTED;  // (the residual epilogue: fragment-direct kernels only)
  // fallback: atomically accumulated statistics in partial row 0, the other rows stay zero
  if (stat_part && !stat_slots &&
      hipMemsetAsync(stat_part, 0, sizeof(double) * 2 * (size_t)N * stat_parts, (hipStream_t)stream) != hipSuccess)
    return M3D_ERR_LAUNCH;  // (slot mode: the table is already zero; everything lands in slot 0)
  const int NT = N <= 16 ? 1 : (N <= 32 ? 2 : 4);
  const int64_t mtiles = m3d_cdiv(M, BM);
  const int64_t ntiles = m3d_cdiv(N, 16 * NT);
  // persistent over M tiles when column statistics are accumulated (fewer fp64 atomics per column)
  int64_t gx = mtiles;
  const int64_t cap = stat_part ? 1024 : 65535;
  if (gx > cap) gx = cap;
  if (ntiles > 65535 || splitk > 65535) return M3D_ERR_UNSUPPORTED;
  dim3 grid((unsigned)gx, (unsigned)ntiles, (unsigned)splitk), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (NT == 1) hipLaunchKernelGGL(gemm_kernel<1>, grid, block, 0, st, g);
  else if (NT == 2) hipLaunchKernelGGL(gemm_kernel<2>, grid, block, 0, st, g);
  else hipLaunchKernelGGL(gemm_kernel<4>, grid, block, 0, st, g);
  M3D_CHECK_LAUNCH();
  return M3D_OK;
}

// SharedMLP layer whose INPUT is the raw output z of the SharedMLP layer in front (its train-mode BatchNorm + LeakyReLU not
// applied yet): C[M, N] = lrelu(z * scale + shift) B^T + bias with slot-mode statistics of C, scale / shift derived in the
// launch from the front layer's slot statistics (pyg_randla_net.py:97-109: torch runs BatchNorm1d, LeakyReLU and the next
// Linear as three kernels and three passes).  K <= 64 (the row-stream kernel: levels 1-2), 16-byte rows, un-gathered A.
extern "C" int m3d_gemm_bn_on_load_f32(const M3DBnOnLoad* pro, const float* z, int32_t k0, const float* b, int64_t ldb,
                                       int64_t M, int32_t N, const float* bias, double* stat_part, int32_t stat_slots,
                                       float* c, int64_t ldc, void* stream) {
  if (!pro || M < 0 || N < 0 || k0 < 1 || stat_slots < 1) return M3D_ERR_INVALID;
  if (M == 0 || N == 0) return M3D_OK;
  if (!z || !b || !c || !stat_part || !pro->slots || !pro->scale || !pro->shift || pro->nslots < 1 || pro->count < 1)
    return M3D_ERR_INVALID;
  if (k0 > 64 || (k0 & 3)) return M3D_ERR_UNSUPPORTED;
  GemmArgs g{};
  g.a0 = z; g.lda0 = k0; g.k0 = k0; g.b = b; g.ldb = ldb; g.M = M; g.N = N; g.bias = bias;
  g.stat_part = stat_part; g.stat_sum = stat_part; g.stat_sumsq = stat_part + N; g.stat_slots = stat_slots;
  g.c = c; g.ldc = ldc; g.splitk = 1; g.kchunk = m3d_align((int64_t)k0, BK);
  g.fpro = 1; g.fpro_slots = pro->slots; g.fpro_nslots = pro->nslots; g.fpro_count = (double)pro->count;
  g.fpro_gamma = pro->gamma; g.fpro_beta = pro->beta; g.fpro_eps = pro->eps; g.fpro_momentum = pro->momentum;
  g.fpro_rmean = pro->running_mean; g.fpro_rvar = pro->running_var; g.fpro_scale = pro->scale; g.fpro_shift = pro->shift;
  g.fpro_mean = pro->mean; g.fpro_invstd = pro->invstd; g.fpro_act = pro->act & 1; g.fpro_slope = pro->slope;
  g.io = (pro->act & M3D_IO_BF16) ? 1 : 0;  // (pro->act: bit 0 = LeakyReLU, M3D_IO_BF16 = z, y and c hold bf16)
  g.fpro_y = pro->y;
  const int rc = m3d_gemm_direct_try(g, (hipStream_t)stream);
  return rc == 1 ? M3D_ERR_UNSUPPORTED : rc;  // (never the LDS-tiled fallback: it knows no prologue)
}

// Two independent products C_i[M, N] (+)= A_i[M, K_i] B_i^T with one output shape as ONE launch when the fragment-direct
// k-loop kernel covers both (K_i > 64, 16-byte aligned rows, same tile plan); otherwise two ordinary launches.  The mlp2 /
// shortcut Linears of a DilatedResidualBlock (/root/reference/myria3d/models/modules/pyg_randla_net.py:172-188) and their
// input gradients on the deep levels, where each product alone leaves most of the chip idle.
// flags: bit 0 = column-major B (the dgrad pattern: element (n, k) of B_i at b_i[k * ldb_i + n]), bit 8 = bf16 operands.
extern "C" int m3d_gemm_pair_f32(const float* const* a, const int64_t* lda, const int32_t* k, const float* const* b,
                                 const int64_t* ldb, int64_t M, int32_t N, const float* const* bias,
                                 double* const* stat_part, int32_t stat_parts, float* const* c, const int64_t* ldc,
                                 const int32_t* accumulate, int32_t flags, void* stream) {
  if (!a || !lda || !k || !b || !ldb || !c || !ldc || !accumulate) return M3D_ERR_INVALID;
  if (M < 0 || N < 0) return M3D_ERR_INVALID;
  if (M == 0 || N == 0) return M3D_OK;
  const int b_cm = flags & 1, bf16 = (flags >> 8) & 1, io = (flags & M3D_IO_BF16) ? 1 : 0;
  GemmArgs g[2];
  for (int i = 0; i < 2; ++i) {
    if (!a[i] || !b[i] || !c[i] || k[i] < 1) return M3D_ERR_INVALID;
    double* sp = stat_part ? stat_part[i] : nullptr;
    if (sp && stat_parts >= 0) return M3D_ERR_INVALID;  // statistics: slot mode only (a pre-zeroed [-stat_parts][2][N] table)
    if (sp && (accumulate[i] || b_cm)) return M3D_ERR_INVALID;
    GemmArgs z{};
    z.a0 = a[i]; z.lda0 = lda[i]; z.k0 = k[i]; z.b = b[i]; z.ldb = ldb[i]; z.b_cm = b_cm; z.M = M; z.N = N;
    z.bias = bias ? bias[i] : nullptr; z.slope = 0.f;
    z.stat_part = sp; z.stat_sum = sp; z.stat_sumsq = sp ? sp + N : nullptr; z.stat_slots = sp ? -stat_parts : 0;
    z.c = c[i]; z.ldc = ldc[i]; z.accumulate = accumulate[i]; z.bf16 = bf16; z.splitk = 1; z.kchunk = m3d_align(k[i], BK);
    z.io = io;
    g[i] = z;
  }
  int rc = 1;
  if (!GEMM_LEGACY) rc = m3d_gemm_direct_pair_try(g[0], g[1], (hipStream_t)stream);
  if (rc != 1) return rc;
  for (int i = 0; i < 2; ++i) {
    rc = m3d_gemm_f32(a[i], lda[i], 0, nullptr, k[i], nullptr, 0, 0, b[i], ldb[i], b_cm, M, N, bias ? bias[i] : nullptr,
                      nullptr, nullptr, (bf16 << 8) | (flags & M3D_IO_BF16), 0.f, stat_part ? stat_part[i] : nullptr, stat_parts, c[i], ldc[i],
                      accumulate[i], 1, stream);
    if (rc != M3D_OK) return rc;
  }
  return M3D_OK;
}

// BatchNorm backward (pass 2) fused into the dgrad GEMM of the Linear in front of it:
//   dz = scale * (dy * act' - s1/M - zhat * s2/M)      (never read back by this launch; stored for the wgrad GEMM)
//   dx[M, Kin] = dz[M, N] W[N, Kin]
// `sums` is the slot table m3d_bn_bwd leaves behind in reduce-only mode.  Replaces bn_bwd_apply_kernel + the plain dgrad
// launch for torch's BatchNorm1d / Linear backward (/root/reference/myria3d/models/modules/pyg_randla_net.py:97-109).
extern "C" int m3d_bn_dgrad_f32(const float* dy, const float* z, const float* scale, const float* shift,
                                const float* mean, const float* invstd, int32_t act, float slope, const double* sums,
                                int32_t nslots, int64_t M, int32_t N, const float* w, int64_t ldw, int32_t Kin,
                                float* dx, int64_t lddx, float* dz, float* dgamma, float* dbeta, int32_t flags,
                                int32_t dx_split, float* dx1, int64_t lddx1, const M3DDropout* drop, void* stream) {
  if (M < 0 || N < 0 || Kin < 0 || nslots < 1 || dx_split < 0) return M3D_ERR_INVALID;
  if (M == 0 || N == 0 || Kin == 0) return M3D_OK;
  if (!dy || !z || !scale || !shift || !mean || !invstd || !sums || !w || !dx || !dz) return M3D_ERR_INVALID;
  if ((N % 4) || ((((uintptr_t)dy) | ((uintptr_t)z) | ((uintptr_t)dz)) & 15)) return M3D_ERR_UNSUPPORTED;
  GemmArgs g{};
  g.a0 = dy; g.lda0 = N; g.k0 = N; g.b = w; g.ldb = ldw; g.b_cm = 1; g.M = M; g.N = Kin;
  g.c = dx; g.ldc = lddx; g.splitk = 1; g.kchunk = m3d_align((int64_t)N, BK);
  g.bf16 = (flags >> 8) & 1;  // flags: bit 0 = add into dgamma / dbeta, bit 8 = bf16 matrix-core operands,
  g.accumulate = (flags >> 9) & 1;  // bit 9 = add the input gradient into dx (and dx1) instead of storing it
  g.io = (flags >> 12) & 3;         // M3D_IO_BF16: dy, z, dz, dx, dx1 hold bf16; with M3D_IO_A32 dy is fp32
  g.pro_z = z; g.pro_scale = scale; g.pro_shift = shift; g.pro_mean = mean; g.pro_invstd = invstd;
  g.pro_sums = sums; g.pro_slots = nslots; g.pro_act = act & 1; g.pro_slope = slope;
  g.pro_dz = dz; g.pro_dgamma = dgamma; g.pro_dbeta = dbeta; g.pro_acc = flags & 1;
  g.pro_drop = drop_args(drop); g.pro_dkey = 0u;
  g.c_split = dx_split; g.c1 = dx1; g.ldc1 = lddx1;
  const int rc = m3d_gemm_direct_try(g, (hipStream_t)stream);
  return rc == 1 ? M3D_ERR_UNSUPPORTED : rc;
}

// per-column sum of a row-major [M, N] matrix, accumulated (atomically) into out[N]:
// the bias gradient of a Linear that is not followed by BatchNorm (fc0, fc_classif).
template <bool H>  // H: x holds bf16 (m3d_colsum_bf16)
__global__ __launch_bounds__(256) void colsum_kernel(const void* __restrict__ xv, int64_t ld, int64_t M, int N,
                                                     float* __restrict__ out) {
  auto X = [&](int64_t e) -> float { return io_load1<H>(xv, (size_t)e); };
  // thread = (column c = tid % cols, row lane rl); grid-stride over rows; one atomic per column per workgroup
  // (same-address atomics serialise at ~15 ns each on MI355X: never one per thread)
  __shared__ float red[256];
  const int tid = threadIdx.x;
  const int cols = N < 256 ? N : 256;
  const int rpp = 256 / cols;
  const int c = tid % cols, rl = tid / cols;
  for (int cb = 0; cb < N; cb += cols) {
    const int n = cb + c;
    float acc = 0.f;
    if (rl < rpp && n < N) {
      // eight rows in flight per thread (one load at a time made the 204 800-row bias gradients a chain of ~50 round trips)
      const int64_t step = (int64_t)gridDim.x * rpp;
      int64_t r = (int64_t)blockIdx.x * rpp + rl;
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
      for (; r + 7 * step < M; r += 8 * step) {
        const float v0 = X(r * ld + n), v1 = X((r + step) * ld + n), v2 = X((r + 2 * step) * ld + n),
                    v3 = X((r + 3 * step) * ld + n), v4 = X((r + 4 * step) * ld + n), v5 = X((r + 5 * step) * ld + n),
                    v6 = X((r + 6 * step) * ld + n), v7 = X((r + 7 * step) * ld + n);
        a0 += v0 + v4; a1 += v1 + v5; a2 += v2 + v6; a3 += v3 + v7;
      }
      for (; r < M; r += step) a0 += X(r * ld + n);
      acc = (a0 + a1) + (a2 + a3);
    }
    __syncthreads();
    red[tid] = acc;
    __syncthreads();
    if (rl == 0 && n < N) {
      float v = 0.f;
      for (int q = 0; q < rpp; ++q) v += red[q * cols + c];
      atomicAdd(&out[n], v);
    }
  }
}

// the same for rows of float4 (N and ld multiples of 4, N / 4 a power of two <= 256, the matrix inside one 2 GiB buffer
// descriptor): 1 024 threads per workgroup = (column group, row lane), eight rows of 16 B in flight per thread, buffer loads
// whose offset past the end reads as 0 (no tail loop).  (The dword kernel keeps 2 MB in flight over the chip: 33 us for fc0's
// 204 800 x 32 bias gradient, whose bytes stream in 4; at most 256 workgroups, because each ends in one same-address atomic
// per column, ~15 ns apiece — hence the 16 waves per workgroup.)
template <bool H>  // H: x holds bf16 — 8-byte loads at half the offset through a 1 GiB descriptor (oob >> 1 = its size)
__global__ __launch_bounds__(1024) void colsum4_kernel(const void* __restrict__ x, int64_t ld4, int64_t M, int N4,
                                                       float* __restrict__ out) {
  __shared__ float4 red[1024];
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, H ? 0x40000000u : 0x80000000u, 0x00020000);
  const unsigned oob = 0x80000000u;
  const int tid = threadIdx.x;
  const int rpp = 1024 / N4;
  const int c = tid % N4, rl = tid / N4;
  const int64_t step = (int64_t)gridDim.x * rpp;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int64_t r = (int64_t)blockIdx.x * rpp + rl; r < M; r += 8 * step) {
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int64_t rr = r + u * step;
      const unsigned off = rr < M ? (unsigned)((rr * ld4 + c) * 16) : oob;
      if constexpr (H) {
        typedef int i32x2_t __attribute__((ext_vector_type(2)));
        const i32x2_t w = __builtin_amdgcn_raw_buffer_load_b64(rs, off >> 1, 0, 0);
        const float4 t = bf16x4_to_f32(make_uint2((unsigned)w[0], (unsigned)w[1]));
        v[u] = (f32x4){t.x, t.y, t.z, t.w};
      } else {
        v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) { acc.x += v[u][0]; acc.y += v[u][1]; acc.z += v[u][2]; acc.w += v[u][3]; }
  }
  red[tid] = acc;
  __syncthreads();
  for (int h = 512; h >= N4; h >>= 1) {  // tree over the row lanes (1024 / N4 of them, a power of two)
    if (tid < h) {
      const float4 o = red[tid + h];
      red[tid].x += o.x; red[tid].y += o.y; red[tid].z += o.z; red[tid].w += o.w;
    }
    __syncthreads();
  }
  if (tid < N4) {
    const float4 v = red[tid];
    atomicAdd(&out[4 * tid + 0], v.x); atomicAdd(&out[4 * tid + 1], v.y);
    atomicAdd(&out[4 * tid + 2], v.z); atomicAdd(&out[4 * tid + 3], v.w);
  }
}

template <bool H>
static int colsum_impl(const void* x, int64_t ld, int64_t M, int32_t N, float* out, void* stream) {
  if (M < 0 || N < 0) return M3D_ERR_INVALID;
  if (M == 0 || N == 0) return M3D_OK;
  if (!x || !out) return M3D_ERR_INVALID;
  const int n4 = N / 4;
  if (N % 4 == 0 && ld % 4 == 0 && (((uintptr_t)x) & 15) == 0 && (n4 & (n4 - 1)) == 0 && n4 <= 256 &&
      M * ld * 4 <= (int64_t)0x80000000u - 64) {
    const int rpp4 = 1024 / n4;
    int64_t g4 = m3d_cdiv(M, (int64_t)rpp4 * 8);
    if (g4 > 256) g4 = 256;
    hipLaunchKernelGGL(colsum4_kernel<H>, dim3((unsigned)g4), dim3(1024), 0, (hipStream_t)stream, x, ld / 4, M, n4, out);
    M3D_CHECK_LAUNCH();
    return M3D_OK;
  }
  int cols = N < 256 ? N : 256;
  int rpp = 256 / cols;
  int64_t gx = m3d_cdiv(M, (int64_t)rpp * 16);
  if (gx > 256) gx = 256;  // (every workgroup ends in one same-address atomic per column: ~15 ns apiece, serialised)
  if (gx < 1) gx = 1;
  hipLaunchKernelGGL(colsum_kernel<H>, dim3((unsigned)gx), dim3(256), 0, (hipStream_t)stream, x, ld, M, N, out);
  M3D_CHECK_LAUNCH();
  return M3D_OK;
}
extern "C" int m3d_colsum_f32(const float* x, int64_t ld, int64_t M, int32_t N, float* out, void* stream) {
  return colsum_impl<false>(x, ld, M, N, out, stream);
}
// the same for a bf16 matrix (M3D_IO_BF16 storage: the bias gradient of fc0 / fc_classif from a bf16 incoming gradient)
extern "C" int m3d_colsum_bf16(const void* x, int64_t ld, int64_t M, int32_t N, float* out, void* stream) {
  return colsum_impl<true>(x, ld, M, N, out, stream);
}
