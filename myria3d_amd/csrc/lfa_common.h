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

template <int CHP> struct LfaCfg {};
template <> struct LfaCfg<16> { static constexpr int ROWS = 256; };
template <> struct LfaCfg<32> { static constexpr int ROWS = 128; };
template <> struct LfaCfg<64> { static constexpr int ROWS = 64; };
template <> struct LfaCfg<128> { static constexpr int ROWS = 64; };
template <> struct LfaCfg<256> { static constexpr int ROWS = 64; };

