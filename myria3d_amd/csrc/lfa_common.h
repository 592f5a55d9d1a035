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

#ifndef LFA_FAST_SQRT
#define LFA_FAST_SQRT 0
#endif
__device__ __forceinline__ void rel_pos(float4 pi, float4 pj, float (&r)[10]) {
  float dx = pj.x - pi.x, dy = pj.y - pi.y, dz = pj.z - pi.z;
  r[0] = pi.x; r[1] = pi.y; r[2] = pi.z;
  r[3] = pj.x; r[4] = pj.y; r[5] = pj.z;
  r[6] = dx; r[7] = dy; r[8] = dz;
  // LFA_FAST_SQRT=1: v_sqrt_f32 (1 ulp) instead of the ~15-instruction correctly rounded sequence, once per edge in every
  // LFA kernel (prepared at the end of round 4 with M3D_FAST_MAX; off until parity and time have been looked at on the GPU)
#if LFA_FAST_SQRT
  r[9] = __builtin_amdgcn_sqrtf(dx * dx + dy * dy + dz * dz);
#else
  r[9] = sqrtf(dx * dx + dy * dy + dz * dz);
#endif
}

// the same with the raw v_sqrt_f32 (1 ulp) for the edge length: the round-5 fast kernels
__device__ __forceinline__ void rel_pos_fast(float4 pi, float4 pj, float (&r)[10]) {
  const float dx = pj.x - pi.x, dy = pj.y - pi.y, dz = pj.z - pi.z;
  r[0] = pi.x; r[1] = pi.y; r[2] = pi.z;
  r[3] = pj.x; r[4] = pj.y; r[5] = pj.z;
  r[6] = dx; r[7] = dy; r[8] = dz;
  r[9] = __builtin_amdgcn_sqrtf(dx * dx + dy * dy + dz * dz);
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
