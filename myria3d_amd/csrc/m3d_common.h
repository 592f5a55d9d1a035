// Shared device/host helpers for the myria3d_amd HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define M3D_OK 0
#define M3D_ERR_INVALID (-1)      // bad argument (null pointer, negative size, unsupported k ...)
#define M3D_ERR_UNSUPPORTED (-2)  // shape outside what the kernels are instantiated for
#define M3D_ERR_LAUNCH (-3)       // hipGetLastError() after a launch was not hipSuccess

#define M3D_WAVE 64

#define M3D_CHECK_LAUNCH()                                   \
  do {                                                       \
    if (hipGetLastError() != hipSuccess) return M3D_ERR_LAUNCH; \
  } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));

static inline int64_t m3d_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int64_t m3d_align(int64_t a, int64_t b) { return m3d_cdiv(a, b) * b; }

// exact-fp32 MFMA: D = A(16x4) * B(4x16) + C, lane l holds A[l&15][l>>4], B[l>>4][l&15],
// C/D: col = l&15, row = (l>>4)*4 + reg   (cdna_hip_programming.md §3)
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ float lrelu(float v, float slope) { return v > 0.f ? v : v * slope; }

// reduce over the four 16-lane groups of a wave (lanes l, l^16, l^32, l^48): all lanes get the result
__device__ __forceinline__ float xgroup_sum(float v) {
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}
__device__ __forceinline__ float xgroup_max(float v) {
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  v = fmaxf(v, __shfl_xor(v, 32, 64));
  return v;
}
__device__ __forceinline__ double xgroup_sum_d(double v) {
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
