// Shared pieces of the LocalFeatureAggregation kernels (forward: lfa.hip, fused backward: lfa_bwd.hip).
#pragma once
#include "m3d_common.h"

struct LfaArgs {
  const float* x;      // [n, D]
  const float4* pos4;  // [n]
  const int32_t* idx;  // [n, K], -1 padded
  const float* wf;     // [D, 10] folded encoder weight
  const float* bf;     // [D]     folded encoder bias
  const float4* wp;    // packed attention weight, see m3d_hip.h
  float* out;          // [n, CH]
  int64_t n;
  int K, CH, D;
  float slope;
};

__device__ __forceinline__ void rel_pos(float4 pi, float4 pj, float (&r)[10]) {
  float dx = pj.x - pi.x, dy = pj.y - pi.y, dz = pj.z - pi.z;
  r[0] = pi.x; r[1] = pi.y; r[2] = pi.z;
  r[3] = pj.x; r[4] = pj.y; r[5] = pj.z;
  r[6] = dx; r[7] = dy; r[8] = dz;
  r[9] = sqrtf(dx * dx + dy * dy + dz * dz);
}

// edge rows per forward workgroup (256 threads) per padded channel count; overridable for tuning sweeps
#ifndef FWD_ROWS_16
#define FWD_ROWS_16 256
#endif
#ifndef FWD_ROWS_32
#define FWD_ROWS_32 128
#endif
#ifndef FWD_ROWS_64
#define FWD_ROWS_64 64
#endif
#ifndef FWD_ROWS_128
#define FWD_ROWS_128 64
#endif
#ifndef FWD_ROWS_256
#define FWD_ROWS_256 64
#endif
template <int CHP> struct LfaCfg {};  // ROWS >= 64: a wavefront must not span two encoder channel groups
template <> struct LfaCfg<16> { static constexpr int ROWS = FWD_ROWS_16; };
template <> struct LfaCfg<32> { static constexpr int ROWS = FWD_ROWS_32; };
template <> struct LfaCfg<64> { static constexpr int ROWS = FWD_ROWS_64; };
template <> struct LfaCfg<128> { static constexpr int ROWS = FWD_ROWS_128; };
template <> struct LfaCfg<256> { static constexpr int ROWS = FWD_ROWS_256; };

// ---- bf16 matrix-core operands (v_mfma_f32_16x16x32_bf16: 8 bf16 per lane per operand, fp32 accumulate).
// A[m][k]: lane l holds row m = l & 15, k = 8 (l >> 4) .. + 7;  B[k][n]: lane l holds k = 8 (l >> 4) .. + 7, column
// n = l & 15;  C/D as the fp32 tile: column l & 15, rows 4 (l >> 4) + reg.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {  // round-to-nearest-even, v_cvt_pk_bf16_f32
  f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
union Bf16Frag {
  bf16x8 v;
  unsigned u[4];
  uint4 q;
};
// 8 consecutive floats (8-byte aligned) of an LDS row -> one operand fragment
__device__ __forceinline__ bf16x8 lds_row_to_bf16(const float* p) {
  const float2 a = *(const float2*)p, b = *(const float2*)(p + 2), c = *(const float2*)(p + 4), d = *(const float2*)(p + 6);
  Bf16Frag f;
  f.u[0] = pack_bf16(a.x, a.y); f.u[1] = pack_bf16(b.x, b.y); f.u[2] = pack_bf16(c.x, c.y); f.u[3] = pack_bf16(d.x, d.y);
  return f.v;
}
// 8 floats of one LDS COLUMN (rows e0 .. e0 + 7, stride `str` floats) -> one operand fragment
__device__ __forceinline__ bf16x8 lds_col_to_bf16(const float* p, int str) {
  Bf16Frag f;
#pragma unroll
  for (int i = 0; i < 4; ++i) f.u[i] = pack_bf16(p[(2 * i) * str], p[(2 * i + 1) * str]);
  return f.v;
}
__device__ __forceinline__ f32x4 mfma_bf16(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
