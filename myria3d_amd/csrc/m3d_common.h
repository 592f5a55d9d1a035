// Shared device/host helpers for the myria3d_amd HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/m3d_hip.h"

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

// XCD-aware work order.  The dispatcher deals consecutive workgroups round-robin over the 8 XCDs (observed; speed only, never
// correctness), each XCD has an L2 of its own, and the points of a batch are stored tile after tile: workgroup b of nblk takes
// work item xcd_major(b, nblk), so that every XCD walks ONE contiguous eighth of the items (two whole tiles at BASELINE
// config 2).  Its L2 then holds just those tiles' rows — and every atomic on a row comes from one XCD, i.e. is resolved in
// that L2 instead of bouncing the line between the chiplets.
#ifndef M3D_XCD_ORDER
#define M3D_XCD_ORDER 1
#endif

// ---- counter-based dropout mask (torch.nn.Dropout of mlp_classif, pyg_randla_net.py:49-52; see m3d_dropout in rows.hip) ----
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
// thr16 == 0: no dropout.  keep(element) <=> its 16 random bits >= thr16; kept values are multiplied by scale = 1 / P(keep).
// rows (nullable): row r of the tensor is row rows[r] of the CALLER's tensor — the net works on a cell-sorted order whose
// ties (points of one grid cell) fall differently from run to run, and the mask must not: it is a function of the seed, the
// step and the caller's element, like the reference's seeded dropout.
struct DropArgs { const int64_t* counter; uint64_t seed; uint32_t thr16; float scale; const int32_t* rows; int64_t* snap; };
// number of float4 `q` of row `r` (rows of n4 float4) in the caller's tensor
__device__ __forceinline__ int64_t drop_index(const DropArgs& d, int64_t r, int q, int n4) {
  return (d.rows ? (int64_t)d.rows[r] : r) * n4 + q;
}
__device__ __forceinline__ uint32_t drop_key(const DropArgs& d) {
  const uint64_t c = (uint64_t)d.counter[0];
  return mix32((uint32_t)d.seed ^ mix32((uint32_t)(d.seed >> 32) + 0x85ebca6bu * ((uint32_t)c + 1u)) ^
               mix32((uint32_t)(c >> 32) * 0xc2b2ae35u + 0x27d4eb2fu));
}
// multipliers (0 or scale) of the four elements of float4 number i of the row-major tensor
__device__ __forceinline__ float4 drop_mul4(uint32_t key, int64_t i, uint32_t thr16, float scale) {
  const uint32_t lo = (uint32_t)i, hi = (uint32_t)(i >> 32);
  const uint32_t r0 = mix32(key ^ mix32(2u * lo + 0x9e3779b9u * hi)), r1 = mix32(key ^ mix32(2u * lo + 1u + 0x9e3779b9u * hi));
  return make_float4((r0 & 0xffffu) >= thr16 ? scale : 0.f, (r0 >> 16) >= thr16 ? scale : 0.f,
                     (r1 & 0xffffu) >= thr16 ? scale : 0.f, (r1 >> 16) >= thr16 ? scale : 0.f);
}
// host side: the C ABI's M3DDropout (nullable) -> kernel arguments
static inline DropArgs drop_args(const M3DDropout* d) {
  DropArgs a{nullptr, 0, 0u, 1.f, nullptr, nullptr};
  if (d && d->counter && d->p > 0.f) {
    uint32_t thr = (uint32_t)(d->p * 65536.f + 0.5f);
    if (thr > 65535u) thr = 65535u;
    a.counter = d->counter; a.seed = d->seed; a.thr16 = thr; a.scale = 1.f / (1.f - (float)thr / 65536.f);
    a.rows = d->rows; a.snap = d->snapshot;
  }
  return a;
}
__device__ __forceinline__ int64_t xcd_major(int64_t b, int64_t nblk) {
  if (!M3D_XCD_ORDER) return b;
  const int64_t q8 = nblk >> 3, r8 = nblk & 7, xcd = b & 7, i8 = b >> 3;
  return xcd * q8 + (xcd < r8 ? xcd : r8) + i8;
}
// persistent kernels: workgroup b of nwg walks the items [g0, gend) with stride gs — the contiguous eighth of its XCD,
// shared with the other workgroups dealt to that XCD
__device__ __forceinline__ void xcd_range(int64_t b, int64_t nwg, int64_t nitems, int64_t& g0, int64_t& gs, int64_t& gend) {
  if (!M3D_XCD_ORDER || nwg < 8) { g0 = b; gs = nwg; gend = nitems; return; }
  const int64_t xcd = b & 7;
  const int64_t lo = nitems * xcd / 8, hi = nitems * (xcd + 1) / 8;
  gs = (nwg + 7 - xcd) >> 3;  // workgroups with b % 8 == xcd
  g0 = lo + (b >> 3);
  gend = hi;
}

// reduce over the four 16-lane groups of a wave (lanes l, l^16, l^32, l^48): all lanes get the result.
// gfx950 row swaps instead of __shfl_xor (which lowers to ds_bpermute_b32: an LDS-pipe round trip per step — the
// softmax of the LFA kernels does six of them per centre): v_permlane16_swap exchanges the odd 16-lane rows of its first
// operand with the even rows of its second, v_permlane32_swap the upper half of the first with the lower half of the
// second; fed the same value twice, the two results hold both members of every pair (row r, row r ^ 1) resp.
// (half h, half h ^ 1) in every lane.
typedef unsigned m3d_u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void xgroup_pair16(float v, float& a, float& b) {
  const m3d_u2 t = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  a = __uint_as_float(t[0]); b = __uint_as_float(t[1]);
}
__device__ __forceinline__ void xgroup_pair32(float v, float& a, float& b) {
  const m3d_u2 t = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  a = __uint_as_float(t[0]); b = __uint_as_float(t[1]);
}
__device__ __forceinline__ float xgroup_sum(float v) {
  float a, b;
  xgroup_pair16(v, a, b); v = a + b;
  xgroup_pair32(v, a, b); v = a + b;
  return v;
}
// fmaxf() is three v_max_f32 whenever the compiler cannot prove that its operands are quiet (values out of a permlane
// swap, an AGPR, LDS): both are canonicalised first.  M3D_FAST_MAX=1: v_med3_f32(a, b, +inf) — the median of the three IS
// the maximum of the two, one instruction, the same value for every pair without a NaN.  The +inf comes from opaque_pinf(),
// once per kernel: the optimiser folds med3(a, b, +inf) with a visible constant back into maxnum and its canonicalisations.
// Prepared at the end of round 4 from the ISA of lfa_fwd_kernel<8,16> (36 of its 56 v_max_f32 are canonicalisations: 500 ->
// 474 VALU instructions); off until it has had its A/B run on the GPU.
#ifndef M3D_FAST_MAX
#define M3D_FAST_MAX 0
#endif
__device__ __forceinline__ float opaque_pinf() {
#if M3D_FAST_MAX
  unsigned u;
  asm("s_mov_b32 %0, 0x7f800000" : "=s"(u));
  return __uint_as_float(u);
#else
  return __builtin_inff();
#endif
}
__device__ __forceinline__ float max_f(float a, float b, float pinf) {
#if M3D_FAST_MAX
  return __builtin_amdgcn_fmed3f(a, b, pinf);
#else
  return fmaxf(a, b);
#endif
}
// the one-instruction maximum, unconditionally (round-5 fast kernels): fast_pinf() once per kernel, then
// fast_max(a, b, pinf) = v_med3_f32(a, b, +inf) = max(a, b) for every pair without a NaN
__device__ __forceinline__ float fast_pinf() {
  unsigned u;
  asm("s_mov_b32 %0, 0x7f800000" : "=s"(u));
  return __uint_as_float(u);
}
__device__ __forceinline__ float fast_max(float a, float b, float pinf) { return __builtin_amdgcn_fmed3f(a, b, pinf); }
__device__ __forceinline__ float xgroup_max(float v, float pinf) {
  float a, b;
  xgroup_pair16(v, a, b); v = max_f(a, b, pinf);
  xgroup_pair32(v, a, b); v = max_f(a, b, pinf);
  return v;
}
// fp64 sums across lanes without the LDS pipe: __shfl_xor on a double is two ds_bpermute_b32 per step; the two 32-bit halves
// travel as DPP moves inside a row of 16 lanes (quad xor 1, quad xor 2, half mirror, mirror: every lane ends with the row
// sum) and as row swaps across rows.  The GEMM statistics flush went from 156 to 74 us per step with it
// (profiles/r03y_gemm_stats_epilogue.log).
template <int CTRL>
__device__ __forceinline__ double dpp_d(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row16_sum_d(double v) {
  v += dpp_d<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_d<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_d<0x141>(v);  // row_half_mirror
  v += dpp_d<0x140>(v);  // row_mirror
  return v;
}
__device__ __forceinline__ double xgroup_sum_d(double v) {
  float alo, blo, ahi, bhi;
  xgroup_pair16(__int_as_float(__double2loint(v)), alo, blo);
  xgroup_pair16(__int_as_float(__double2hiint(v)), ahi, bhi);
  v = __hiloint2double(__float_as_int(ahi), __float_as_int(alo)) + __hiloint2double(__float_as_int(bhi), __float_as_int(blo));
  xgroup_pair32(__int_as_float(__double2loint(v)), alo, blo);
  xgroup_pair32(__int_as_float(__double2hiint(v)), ahi, bhi);
  return __hiloint2double(__float_as_int(ahi), __float_as_int(alo)) + __hiloint2double(__float_as_int(bhi), __float_as_int(blo));
}
__device__ __forceinline__ double wave_sum_d(double v) { return xgroup_sum_d(row16_sum_d(v)); }

// acc[s] = sum over the slot rows p < nslots of base[p * row_stride + s * series_stride], in row order, with the loads of
// eight rows of all S series in flight.  (A rolled `for p: acc += row[p]` waits for every row: 4-8 dependent L2 round
// trips in front of every workgroup of a consumer — the ~8 us floor of the small layers' BatchNorm launches.)
template <int S>
__device__ __forceinline__ void slot_sums(const double* __restrict__ base, int nslots, size_t row_stride, size_t series_stride,
                                          double (&acc)[S]) {
#pragma unroll
  for (int s = 0; s < S; ++s) acc[s] = 0.0;
  for (int p0 = 0; p0 < nslots; p0 += 8) {
    double v[8][S];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int q = p0 + u < nslots ? p0 + u : nslots - 1;  // clamped: an unconditional load, its value dropped below
#pragma unroll
      for (int s = 0; s < S; ++s) v[u][s] = base[(size_t)q * row_stride + (size_t)s * series_stride];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int s = 0; s < S; ++s) acc[s] += p0 + u < nslots ? v[u][s] : 0.0;
  }
}
__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

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
// ---- bf16 ACTIVATION STORAGE (round 6; M3D_IO_* bits of the entry points' flags, include/m3d_hip.h).  A feature matrix
// [rows, channels] may live in HBM as bf16 (2-byte elements, round-to-nearest-even on store, exact widening on load); every
// kernel computes in fp32 registers as before.  The helpers take the element type as a template flag (H = bf16) and ELEMENT
// offsets, so one kernel body serves both layouts; H = false compiles to the fp32 code it replaces.
__device__ __forceinline__ float4 bf16x4_to_f32(uint2 v) {
  return make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16),
                     __uint_as_float(v.y & 0xffff0000u));
}
__device__ __forceinline__ uint2 f32x4_to_bf16(float4 v) { return make_uint2(pack_bf16(v.x, v.y), pack_bf16(v.z, v.w)); }
__device__ __forceinline__ unsigned short f32_to_bf16(float v) { return (unsigned short)(pack_bf16(v, 0.f) & 0xffffu); }
__device__ __forceinline__ float bf16_to_f32(unsigned short v) { return __uint_as_float((unsigned)v << 16); }
// four consecutive elements starting at element `e` (a multiple of 4; base 16-byte resp. 8-byte aligned)
template <bool H>
__device__ __forceinline__ float4 io_load4(const void* base, size_t e) {
  if constexpr (H) return bf16x4_to_f32(*(const uint2*)((const unsigned short*)base + e));
  else return *(const float4*)((const float*)base + e);
}
template <bool H>
__device__ __forceinline__ void io_store4(void* base, size_t e, float4 v) {
  if constexpr (H) *(uint2*)((unsigned short*)base + e) = f32x4_to_bf16(v);
  else *(float4*)((float*)base + e) = v;
}
template <bool H>
__device__ __forceinline__ float io_load1(const void* base, size_t e) {
  if constexpr (H) return bf16_to_f32(((const unsigned short*)base)[e]);
  else return ((const float*)base)[e];
}
template <bool H>
__device__ __forceinline__ void io_store1(void* base, size_t e, float v) {
  if constexpr (H) ((unsigned short*)base)[e] = f32_to_bf16(v);
  else ((float*)base)[e] = v;
}
// the same through BYTE offsets computed for fp32 elements (the LFA kernels' 32-bit offset arithmetic): a bf16 element sits
// at half the offset
template <bool H>
__device__ __forceinline__ float4 io_load4_b(const void* base, unsigned byte_off_f32) {
  if constexpr (H) return bf16x4_to_f32(*(const uint2*)((const char*)base + (byte_off_f32 >> 1)));
  else return *(const float4*)((const char*)base + byte_off_f32);
}
template <bool H>
__device__ __forceinline__ float io_load1_b(const void* base, unsigned byte_off_f32) {
  if constexpr (H) return bf16_to_f32(*(const unsigned short*)((const char*)base + (byte_off_f32 >> 1)));
  else return *(const float*)((const char*)base + byte_off_f32);
}
template <bool H>
__device__ __forceinline__ void io_store1_b(void* base, unsigned byte_off_f32, float v) {
  if constexpr (H) *(unsigned short*)((char*)base + (byte_off_f32 >> 1)) = f32_to_bf16(v);
  else *(float*)((char*)base + byte_off_f32) = v;
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
// ---- split-bf16 ("bf16x3") operands: x = hi + lo with hi = bf16(x), lo = bf16(x - hi): 16 mantissa bits in two bf16 values.
// a * b ~= a_hi b_hi + a_hi b_lo + a_lo b_hi on the bf16 matrix cores (fp32 accumulate): the fp32 contract's attention GEMMs at
// ~2^-16 relative operand error instead of bf16's 2^-8, off the vector pipe the f32-input MFMAs share with the VALU phases.
struct Bf16Split { bf16x8 hi, lo; };
__device__ __forceinline__ void split_pair(float a, float b, unsigned& hi, unsigned& lo) {
  hi = pack_bf16(a, b);
  const float ah = __uint_as_float(hi << 16), bh = __uint_as_float(hi & 0xffff0000u);
  lo = pack_bf16(a - ah, b - bh);
}
__device__ __forceinline__ Bf16Split lds_row_to_bf16_split(const float* p) {
  const float2 a = *(const float2*)p, b = *(const float2*)(p + 2), c = *(const float2*)(p + 4), d = *(const float2*)(p + 6);
  Bf16Frag h, l;
  split_pair(a.x, a.y, h.u[0], l.u[0]); split_pair(b.x, b.y, h.u[1], l.u[1]);
  split_pair(c.x, c.y, h.u[2], l.u[2]); split_pair(d.x, d.y, h.u[3], l.u[3]);
  return Bf16Split{h.v, l.v};
}
__device__ __forceinline__ Bf16Split lds_col_to_bf16_split(const float* p, int str) {
  Bf16Frag h, l;
#pragma unroll
  for (int i = 0; i < 4; ++i) split_pair(p[(2 * i) * str], p[(2 * i + 1) * str], h.u[i], l.u[i]);
  return Bf16Split{h.v, l.v};
}
__device__ __forceinline__ f32x4 mfma_bf16(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
