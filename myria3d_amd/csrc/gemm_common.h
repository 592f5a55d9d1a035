// Argument block shared by the SharedMLP GEMM kernels (gemm.hip: LDS-tiled fallback; gemm_direct.hip: the
// LDS-free fragment-direct kernels).  See m3d_gemm_f32 in include/m3d_hip.h for the meaning of each field.
#pragma once
#include "m3d_common.h"

struct GemmArgs {
  const float* a0; int64_t lda0; const int32_t* a0_rows; int k0;
  const float* a1; int64_t lda1; int k1;
  int a_cm;
  const float* b; int64_t ldb; int b_cm;
  int64_t M; int N;
  const float* bias; const float* scale; const float* shift; int act; float slope;
  double* stat_sum; double* stat_sumsq;  // LDS-tiled fallback: atomically accumulated [N] vectors (row 0 of stat_part)
  double* stat_part;                     // direct kernels: [gridDim.x][2][N] per-workgroup partials, plainly stored
  int stat_slots;                        // > 0: stat_part is a pre-zeroed [stat_slots][2][N] table; workgroup w ADDS its
                                         // partial to slot w % stat_slots (fp64 atomics) -> the consumer sums a handful
                                         // of slots itself and no finalize kernel is needed
  float* c; int64_t ldc; int accumulate;
  int acc_pre;  // (set by m3d_gemm_direct_try) accumulate: the old C tile is loaded INTO the MFMA accumulators at the top of a
                // tile — next to the operand loads, one memory round trip per tile — instead of read-modify-written in the epilogue
  int splitk; int64_t kchunk;  // reduction elements per split (multiple of BK)
  int bf16;                    // != 0: operands rounded to bf16 on load, v_mfma_f32_16x16x32_bf16 (K % 32 == 0, K > 64)
  int ksplit;                  // k-loop kernels: != 0: the four waves of a workgroup share ONE row group and a quarter of K each
                               // (accumulators meet in LDS): the deep layers have < 1 wave per SIMD and a K loop of 16-48
                               // dependent chunks otherwise
  // BatchNorm-backward A-prologue (m3d_bn_dgrad_f32; direct kernels, dgrad pattern only).  a0 = dy and pro_z = z share
  // the [M, k0] layout; the A operand the MFMAs see is
  //   dz = scale * (dy * act'(z*scale + shift) - s1/M - (z - mean) * invstd * s2/M)
  // with s1, s2 summed from the pre-reduced slot table pro_sums[pro_slots][3][k0] (m3d_bn_bwd, reduce-only mode).
  // Column slice 0 also stores dz (the weight-gradient GEMM reads it) and workgroup (0, 0) the parameter gradients.
  const float* pro_z; const float* pro_scale; const float* pro_shift; const float* pro_mean; const float* pro_invstd;
  const double* pro_sums; int pro_slots; int pro_act; float pro_slope;
  float* pro_dz; float* pro_dgamma; float* pro_dbeta; int pro_acc;
  DropArgs pro_drop; uint32_t pro_dkey;  // the layer's output went through dropout (thr16 != 0): dy is masked on load
  // split output (input gradient of a layer whose input was cat([x0[rows], x1]), FPModule): columns [0, c_split) go to
  // c[m][.] (ldc), columns [c_split, N) to c1[m][. - c_split] (ldc1) — two contiguous matrices instead of one that has to
  // be sliced and copied.  c_split = 0: plain output.  Direct kernels, plain epilogue only.
  int c_split; float* c1; int64_t ldc1;
  // forward A-prologue (m3d_gemm_bn_on_load_f32, round 5; row-stream kernels, statistics epilogue): A0 holds the RAW output z
  // [M, k0] of the SharedMLP layer in front, whose train-mode BatchNorm + LeakyReLU has not been applied: this launch
  // derives that layer's scale / shift from its slot statistics (every workgroup, as m3d_bn_stats_apply's do; workgroup (0, 0)
  // also stores scale / shift / mean / invstd for the backward pass and updates the running statistics), its MFMAs see
  // y = lrelu(z * scale + shift), and column slice 0 stores y (the weight-gradient GEMM of THIS layer reads it) — the
  // m3d_bn_stats_apply launch between the two GEMMs and one pass over the activation are gone.
  // activation layouts (round 6): the M3D_IO_* bits of the entry point's flags >> 12.  0: every matrix fp32.  Bit 0: A0, A1,
  // C, C1, pro_z, pro_dz, fpro_y hold bf16 (declared float* here: the kernels reinterpret; leading dimensions stay in
  // ELEMENTS); bit 1: ... but A0 is fp32; bit 2: ... but C / C1 are fp32.  Weights, bias, statistics: always fp32 / fp64.
  int io;
  int fpro;
  const double* fpro_slots; int fpro_nslots; double fpro_count;
  const float* fpro_gamma; const float* fpro_beta; float fpro_eps, fpro_momentum;
  float* fpro_rmean; float* fpro_rvar; float* fpro_scale; float* fpro_shift; float* fpro_mean; float* fpro_invstd;
  int fpro_act; float fpro_slope; float* fpro_y;
};

// gemm_direct.hip: returns M3D_OK when it handled the problem, 1 when the shape is not covered (caller falls back)
int m3d_gemm_direct_try(const GemmArgs& g, hipStream_t st);
// two products of one output shape as ONE launch (deep levels: the mlp2 / shortcut pair of a block); 1: launch them singly
int m3d_gemm_direct_pair_try(const GemmArgs& a, const GemmArgs& b, hipStream_t st);
// number of statistics partial rows the direct kernels write for a forward GEMM of this shape (>= 1)
int m3d_gemm_direct_stat_parts(int64_t M, int N, int K);
