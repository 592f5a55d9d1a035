// Row gather / scatter-add, xyz padding and device-side random decimation (gfx950).
//
// Replaces, for /root/reference/myria3d/models/modules/pyg_randla_net.py:
//   * `tensor[idx_decim]` in decimate() (:234-238) and the x[nn] gather inside knn_interpolate(k=1) (:250)
//       -> m3d_gather_rows; its autograd transpose -> m3d_scatter_add_rows
//   * decimation_indices() (:192-231): a Python loop of per-cloud torch.randperm calls with host syncs
//       -> m3d_decimation_indices: one launch; slot r of cloud b gets ptr[b] + P_b(r) where P_b is a keyed
//          pseudo-random permutation of [0, n_b) (cycle-walking balanced Feistel network), i.e. the head of a
//          random permutation, exactly the reference's sampling scheme without materialising the permutation.
#include "m3d_common.h"
#include "../../include/m3d_hip.h"

// H: src and out hold bf16 (m3d_gather_rows_bf16)
template <bool H>
__global__ __launch_bounds__(256) void gather_rows_kernel(const void* __restrict__ src, int64_t ld,
                                                          const int32_t* __restrict__ idx, void* __restrict__ out,
                                                          int64_t m, int C) {
  if ((C & 3) == 0 && (ld & 3) == 0) {
    const int C4 = C >> 2;
    const int64_t total = m * C4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
      int64_t r = i / C4;
      int c = (int)(i % C4);
      int64_t s = idx ? (int64_t)idx[r] : r;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (s >= 0) v = io_load4<H>(src, (size_t)(s * ld + c * 4));
      io_store4<H>(out, 4 * (size_t)i, v);
    }
  } else {
    const int64_t total = m * C;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
      int64_t r = i / C;
      int c = (int)(i % C);
      int64_t s = idx ? (int64_t)idx[r] : r;
      io_store1<H>(out, (size_t)i, s >= 0 ? io_load1<H>(src, (size_t)(s * ld + c)) : 0.f);
    }
  }
}

extern "C" int m3d_gather_rows(const float* src, int64_t ld, const int32_t* idx, float* out, int64_t m, int32_t C,
                               void* stream) {
  if (m < 0 || C < 0) return M3D_ERR_INVALID;
  if (m == 0 || C == 0) return M3D_OK;
  if (!src || !out) return M3D_ERR_INVALID;
  if ((((uintptr_t)src) & 15) || (((uintptr_t)out) & 15)) return M3D_ERR_INVALID;
  int64_t gx = m3d_cdiv(m * (int64_t)((C + 3) / 4), 256);
  if (gx > 8192) gx = 8192;
  hipLaunchKernelGGL(gather_rows_kernel<false>, dim3((unsigned)gx), dim3(256), 0, (hipStream_t)stream, (const void*)src, ld, idx,
                     (void*)out, m, C);
  M3D_CHECK_LAUNCH();
  return M3D_OK;
}

// the same for bf16 rows (M3D_IO_BF16 storage of the feature matrices: decimate() and the output un-permutation of a net
// whose activations live in bf16); ld in elements, 16-byte aligned bases
extern "C" int m3d_gather_rows_bf16(const void* src, int64_t ld, const int32_t* idx, void* out, int64_t m, int32_t C,
                                    void* stream) {
  if (m < 0 || C < 0) return M3D_ERR_INVALID;
  if (m == 0 || C == 0) return M3D_OK;
  if (!src || !out) return M3D_ERR_INVALID;
  if ((((uintptr_t)src) & 15) || (((uintptr_t)out) & 15)) return M3D_ERR_INVALID;
  int64_t gx = m3d_cdiv(m * (int64_t)((C + 3) / 4), 256);
  if (gx > 8192) gx = 8192;
  hipLaunchKernelGGL(gather_rows_kernel<true>, dim3((unsigned)gx), dim3(256), 0, (hipStream_t)stream, src, ld, idx, out, m, C);
  M3D_CHECK_LAUNCH();
  return M3D_OK;
}

// H: src holds bf16; the atomically accumulated `out` is fp32 either way
template <bool H>
__global__ __launch_bounds__(256) void scatter_add_rows_kernel(const void* __restrict__ src,
                                                               const int32_t* __restrict__ idx,
                                                               float* __restrict__ out, int64_t ldo, int64_t m,
                                                               int C) {
  // XCD-aware (m3d_common.h): the rows of a tile — sources and targets — stay with one XCD, its atomics in that L2
  int64_t b0, bs, bend;
  xcd_range(blockIdx.x, gridDim.x, (m * C + 255) / 256, b0, bs, bend);
  const int64_t total = m * C;
  for (int64_t blk = b0; blk < bend; blk += bs) {
    const int64_t i = blk * 256 + threadIdx.x;
    if (i >= total) break;
    int64_t r = i / C;
    int c = (int)(i % C);
    int64_t d = (int64_t)idx[r];
    if (d >= 0) atomicAdd(out + d * ldo + c, io_load1<H>(src, (size_t)i));
  }
}

// the same for DISTINCT targets (the transpose of a subset selection: decimate(), pyg_randla_net.py:234-238): no two
// rows meet, so a plain 16-byte read-modify-write does what 4 float atomics did
// (H: src AND out hold bf16 — a plain read-modify-write of 8 bytes)
template <bool H>
__global__ __launch_bounds__(256) void scatter_add_distinct_rows_kernel(const void* __restrict__ src,
                                                                        const int32_t* __restrict__ idx,
                                                                        void* __restrict__ out, int64_t ldo, int64_t m,
                                                                        int C4) {
  const int64_t total = m * C4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / C4;
    const int q = (int)(i % C4);
    const int64_t d = (int64_t)idx[r];
    if (d < 0) continue;
    const size_t de = (size_t)(d * ldo + 4 * q);
    const float4 v = io_load4<H>(src, 4 * (size_t)i);
    float4 o = io_load4<H>(out, de);
    o.x += v.x; o.y += v.y; o.z += v.z; o.w += v.w;
    io_store4<H>(out, de, o);
  }
}

extern "C" int m3d_scatter_add_rows(const float* src, const int32_t* idx, float* out, int64_t ldo, int64_t m,
                                    int32_t C, int32_t flags, void* stream) {
  if (m < 0 || C < 0) return M3D_ERR_INVALID;
  if (m == 0 || C == 0) return M3D_OK;
  if (!src || !out || !idx) return M3D_ERR_INVALID;
  // flags: bit 0 = distinct targets; M3D_IO_BF16 = src holds bf16 — and so does out with distinct targets; the ATOMIC
  // form (bit 0 clear) accumulates into an fp32 `out` whatever src is
  const bool h = (flags & M3D_IO_BF16) != 0;
  if (flags & 1) {
    if ((C & 3) || (ldo & 3) || ((((uintptr_t)src) | ((uintptr_t)out)) & 15)) {
      if (h) return M3D_ERR_UNSUPPORTED;  // (the fp32 form falls back to atomics; a bf16 target has none)
    } else {
      int64_t g4 = m3d_cdiv(m * (int64_t)(C / 4), 256);
      if (g4 > 8192) g4 = 8192;
      if (h) hipLaunchKernelGGL(scatter_add_distinct_rows_kernel<true>, dim3((unsigned)g4), dim3(256), 0, (hipStream_t)stream,
                                (const void*)src, idx, (void*)out, ldo, m, C / 4);
      else hipLaunchKernelGGL(scatter_add_distinct_rows_kernel<false>, dim3((unsigned)g4), dim3(256), 0, (hipStream_t)stream,
                              (const void*)src, idx, (void*)out, ldo, m, C / 4);
      M3D_CHECK_LAUNCH();
      return M3D_OK;
    }
  }
  int64_t gx = m3d_cdiv(m * (int64_t)C, 256 * 2);
  if (gx > 8192) gx = 8192;
  if (gx < 1) gx = 1;
  if (h) hipLaunchKernelGGL(scatter_add_rows_kernel<true>, dim3((unsigned)gx), dim3(256), 0, (hipStream_t)stream, (const void*)src,
                            idx, out, ldo, m, C);
  else hipLaunchKernelGGL(scatter_add_rows_kernel<false>, dim3((unsigned)gx), dim3(256), 0, (hipStream_t)stream, (const void*)src, idx, out,
                     ldo, m, C);
  M3D_CHECK_LAUNCH();
  return M3D_OK;
}

// ------------------------------------------------------------------------------------------
// The transpose of a many-to-one row gather WITHOUT atomics.  knn_interpolate(k = 1) (pyg_randla_net.py:250) reads
// x_coarse[nn[f]] for every fine point f; its backward pass adds the fine rows back into their coarse row: ~4 fine rows per
// coarse row, 6.5 M fp32 atomics for the [204 800, 32] gradient of level 1 (32 us; 81 us over the four FP modules).  The
// map is position-only: its inverse (CSR lists, built on the geometry stream with the 1-NN tables) turns the scatter into
// a gather-and-sum at streaming speed.
// ------------------------------------------------------------------------------------------
#define CSR_BATCH_MAX 8
struct CsrBatch {
  const int32_t* idx[CSR_BATCH_MAX]; int32_t* cnt[CSR_BATCH_MAX]; int32_t* ptr[CSR_BATCH_MAX]; int32_t* inv[CSR_BATCH_MAX];
  int64_t n[CSR_BATCH_MAX], m[CSR_BATCH_MAX], start[CSR_BATCH_MAX + 1];  // start: first flat fine-row number of job j
  int njobs;
};
__device__ __forceinline__ int csr_job(const CsrBatch& b, int64_t t) {
  int j = 0;
#pragma unroll
  for (int i = 1; i < CSR_BATCH_MAX; ++i) j += (i < b.njobs && t >= b.start[i]) ? 1 : 0;
  return j;
}
// pass 1: cnt[c] = number of fine rows mapped to c
__global__ __launch_bounds__(256) void csr_count_kernel(CsrBatch b) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= b.start[b.njobs]) return;
  const int j = csr_job(b, t);
  const int32_t c = b.idx[j][t - b.start[j]];
  if (c >= 0 && c < b.m[j]) atomicAdd(&b.cnt[j][c], 1);
}
// pass 2 (one workgroup per job): ptr = exclusive prefix sums of cnt, ptr[m] = total
__global__ __launch_bounds__(1024) void csr_scan_kernel(CsrBatch b) {
  __shared__ int wsum[16];
  __shared__ int carry;
  const int j = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int64_t m = b.m[j];
  const int32_t* cnt = b.cnt[j];
  int32_t* ptr = b.ptr[j];
  if (tid == 0) carry = 0;
  __syncthreads();
  for (int64_t base = 0; base < m; base += 4096) {  // four consecutive targets per thread and trip
    const int64_t c0 = base + 4 * (int64_t)tid;
    int v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = c0 + i < m ? cnt[c0 + i] : 0;
    const int tot = (v[0] + v[1]) + (v[2] + v[3]);
    int x = tot;  // inclusive scan of the thread totals inside the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int y = __shfl_up(x, o, 64);
      if (lane >= o) x += y;
    }
    if (lane == 63) wsum[wid] = x;
    __syncthreads();
    int off = carry;
    for (int w = 0; w < wid; ++w) off += wsum[w];
    int run = off + x - tot;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (c0 + i < m) ptr[c0 + i] = run;
      run += v[i];
    }
    __syncthreads();
    if (tid == 1023) carry = off + x;
    __syncthreads();
  }
  if (tid == 0) ptr[m] = carry;
}
// pass 3: inv[ptr[c] + k] = f for the k-th fine row of c (k counts cnt[c] down: cnt is zero again afterwards)
__global__ __launch_bounds__(256) void csr_fill_kernel(CsrBatch b) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= b.start[b.njobs]) return;
  const int j = csr_job(b, t);
  const int64_t f = t - b.start[j];
  const int32_t c = b.idx[j][f];
  if (c >= 0 && c < b.m[j]) {
    const int k = atomicSub(&b.cnt[j][c], 1) - 1;
    b.inv[j][b.ptr[j][c] + k] = (int32_t)f;
  }
}

extern "C" int m3d_csr_invert_batch(int32_t njobs, const int32_t* const* idx, const int64_t* n, const int64_t* m,
                                    int32_t* const* cnt, int32_t* const* ptr, int32_t* const* inv, void* stream) {
  if (njobs < 0 || njobs > CSR_BATCH_MAX) return M3D_ERR_INVALID;
  if (njobs == 0) return M3D_OK;
  if (!idx || !n || !m || !cnt || !ptr || !inv) return M3D_ERR_INVALID;
  CsrBatch b;
  b.njobs = njobs;
  int64_t tot = 0;
  for (int j = 0; j < CSR_BATCH_MAX; ++j) {
    b.start[j] = tot;
    if (j >= njobs) { b.idx[j] = nullptr; b.cnt[j] = b.ptr[j] = b.inv[j] = nullptr; b.n[j] = b.m[j] = 0; continue; }
    if (n[j] < 0 || m[j] < 0 || n[j] > 0x7fffffff || m[j] >= 0x7fffffff || !ptr[j]) return M3D_ERR_INVALID;
    if ((n[j] > 0 && (!idx[j] || !inv[j])) || (m[j] > 0 && !cnt[j])) return M3D_ERR_INVALID;
    b.idx[j] = idx[j]; b.cnt[j] = cnt[j]; b.ptr[j] = ptr[j]; b.inv[j] = inv[j]; b.n[j] = n[j]; b.m[j] = m[j];
    tot += n[j];
  }
  b.start[CSR_BATCH_MAX] = tot;
  for (int j = njobs; j <= CSR_BATCH_MAX; ++j) b.start[j] = tot;
  hipStream_t st = (hipStream_t)stream;
  const unsigned gx = (unsigned)m3d_cdiv(tot > 0 ? tot : 1, 256);
  if (tot > 0) hipLaunchKernelGGL(csr_count_kernel, dim3(gx), dim3(256), 0, st, b);
  hipLaunchKernelGGL(csr_scan_kernel, dim3((unsigned)njobs), dim3(1024), 0, st, b);
  if (tot > 0) hipLaunchKernelGGL(csr_fill_kernel, dim3(gx), dim3(256), 0, st, b);
  M3D_CHECK_LAUNCH();
  return M3D_OK;
}

// out[c][:] (+)= sum of src[f][:] over f in inv[ptr[c] .. ptr[c + 1])   (C % 4 == 0, 16-byte aligned rows)
// (round 4 prepared a variant with eight contributors' ids, then rows, per trip — -DROWS_GATHER_BATCH=1; its A/B in round 5,
// profiles/r05a_step_lfa_full_ab.log, moved nothing: 4.130 vs 4.118 / 4.135 ms per step; removed)
// H: src and out hold bf16 (fp32 sums in registers)
template <bool H>
__global__ __launch_bounds__(256) void gather_sum_rows_kernel(const void* __restrict__ src, int64_t lds,
                                                              const int32_t* __restrict__ ptr,
                                                              const int32_t* __restrict__ inv, void* __restrict__ out,
                                                              int64_t ldo, int64_t m, int C4, int accumulate) {
  const int64_t total = m * C4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t c = i / C4;
    const int q = (int)(i % C4);
    const int p0 = ptr[c], p1 = ptr[c + 1];
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
    int p = p0;
    for (; p + 1 < p1; p += 2) {
      const float4 a = io_load4<H>(src, (size_t)((int64_t)(inv ? inv[p] : p) * lds + 4 * q));
      const float4 b = io_load4<H>(src, (size_t)((int64_t)(inv ? inv[p + 1] : p + 1) * lds + 4 * q));
      s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
      s1.x += b.x; s1.y += b.y; s1.z += b.z; s1.w += b.w;
    }
    if (p < p1) {
      const float4 a = io_load4<H>(src, (size_t)((int64_t)(inv ? inv[p] : p) * lds + 4 * q));
      s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
    }
    const size_t de = (size_t)(c * ldo + 4 * q);
    float4 o = make_float4(s0.x + s1.x, s0.y + s1.y, s0.z + s1.z, s0.w + s1.w);
    if (accumulate) { const float4 old = io_load4<H>(out, de); o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w; }
    io_store4<H>(out, de, o);
  }
}

#ifndef GATHER_LPL
#define GATHER_LPL 4  // lanes per list of the long-list gather (4 or 8)
#endif
template <int LPL, bool H>
__global__ void gather_sum_rows4_kernel(const void* __restrict__ src, int64_t lds, const int32_t* __restrict__ ptr,
                                        const int32_t* __restrict__ inv, void* __restrict__ out, int64_t ldo, int64_t m, int C4,
                                        int accumulate);
extern "C" int m3d_gather_sum_rows(const float* src, int64_t lds, const int32_t* ptr, const int32_t* inv, float* out,
                                   int64_t ldo, int64_t m, int32_t C, int32_t accumulate, void* stream) {
  if (m < 0 || C < 0) return M3D_ERR_INVALID;
  if (m == 0 || C == 0) return M3D_OK;
  if (!src || !ptr || !out) return M3D_ERR_INVALID;  // (inv == NULL: list c is the rows ptr[c] .. ptr[c + 1] of src themselves)
  if ((C & 3) || (lds & 3) || (ldo & 3) || ((((uintptr_t)src) | ((uintptr_t)out)) & 15)) return M3D_ERR_UNSUPPORTED;
  // accumulate: bit 0 = add into out, bit 1 = long lists, M3D_IO_BF16 = src and out hold bf16
  const bool h = (accumulate & M3D_IO_BF16) != 0;
  const void* sv = src;
  void* ov = out;
  if (accumulate & 2) {  // long lists: four lanes per (target, chunk)
    int64_t gx4 = m3d_cdiv(m * (int64_t)(C / 4), 256 / GATHER_LPL);
    if (gx4 > 65536) gx4 = 65536;
    if (h) hipLaunchKernelGGL((gather_sum_rows4_kernel<GATHER_LPL, true>), dim3((unsigned)gx4), dim3(256), 0, (hipStream_t)stream, sv, lds, ptr, inv,
                              ov, ldo, m, C / 4, accumulate & 1);
    else hipLaunchKernelGGL((gather_sum_rows4_kernel<GATHER_LPL, false>), dim3((unsigned)gx4), dim3(256), 0, (hipStream_t)stream, sv, lds, ptr, inv,
                       ov, ldo, m, C / 4, accumulate & 1);
    M3D_CHECK_LAUNCH();
    return M3D_OK;
  }
  int64_t gx = m3d_cdiv(m * (int64_t)(C / 4), 256);
  if (gx > 8192) gx = 8192;
  if (h) hipLaunchKernelGGL(gather_sum_rows_kernel<true>, dim3((unsigned)gx), dim3(256), 0, (hipStream_t)stream, sv, lds, ptr, inv,
                            ov, ldo, m, C / 4, accumulate & 1);
  else hipLaunchKernelGGL(gather_sum_rows_kernel<false>, dim3((unsigned)gx), dim3(256), 0, (hipStream_t)stream, sv, lds, ptr, inv,
                     ov, ldo, m, C / 4, accumulate & 1);
  M3D_CHECK_LAUNCH();
  return M3D_OK;
}

// fp32 -> bf16 (round-to-nearest-even) of n contiguous values: the network INPUT features of a net whose activations live in
// bf16 (M3D_IO_BF16) — one pass over [sum N, F] per batch, every later kernel then reads 2-byte elements
__global__ __launch_bounds__(256) void convert_bf16_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) dst[i] = f32_to_bf16(src[i]);
}
extern "C" int m3d_convert_f32_bf16(const float* src, void* dst, int64_t n, void* stream) {
  if (n < 0) return M3D_ERR_INVALID;
  if (n == 0) return M3D_OK;
  if (!src || !dst) return M3D_ERR_INVALID;
  int64_t gx = m3d_cdiv(n, 256 * 4);
  if (gx > 4096) gx = 4096;
  hipLaunchKernelGGL(convert_bf16_kernel, dim3((unsigned)gx), dim3(256), 0, (hipStream_t)stream, src, (unsigned short*)dst, n);
  M3D_CHECK_LAUNCH();
  return M3D_OK;
}

__global__ __launch_bounds__(256) void zero_i32_kernel(int32_t* __restrict__ p, int64_t n) {
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i + 4 <= n) *(int4*)(p + i) = make_int4(0, 0, 0, 0);
  else for (int64_t k = i; k < n; ++k) p[k] = 0;
}
// ---- long lists (flags bit 1 of m3d_gather_sum_rows; the reverse neighbour lists of a K-NN table: ~K rows per point): FOUR
// lanes share a (target, float4 chunk) — lane l sums contributors p0 + l, p0 + l + 4, ... — and meet through two quad DPP adds:
// the dependent id -> row load chain of a target is a quarter as long (16 contributors: 64 -> ~20 us for 204 800 targets).
__device__ __forceinline__ float quad_sum(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false));  // quad_perm [1,0,3,2]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, false));  // quad_perm [2,3,0,1]
  return v;
}
template <int LPL, bool H>
__global__ __launch_bounds__(256) void gather_sum_rows4_kernel(const void* __restrict__ src, int64_t lds,
                                                               const int32_t* __restrict__ ptr,
                                                               const int32_t* __restrict__ inv, void* __restrict__ out,
                                                               int64_t ldo, int64_t m, int C4, int accumulate) {
  const int64_t total = m * C4;
  const int l = threadIdx.x & (LPL - 1);
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) / LPL; i < total; i += (int64_t)gridDim.x * (256 / LPL)) {  // (uniform per lane group)
    const int64_t c = i / C4;
    const int q = (int)(i % C4);
    const int p0 = ptr[c], p1 = ptr[c + 1];
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
    int p = p0 + l;
    for (; p + LPL < p1; p += 2 * LPL) {
      const float4 a = io_load4<H>(src, (size_t)((int64_t)(inv ? inv[p] : p) * lds + 4 * q));
      const float4 b = io_load4<H>(src, (size_t)((int64_t)(inv ? inv[p + LPL] : p + LPL) * lds + 4 * q));
      s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
      s1.x += b.x; s1.y += b.y; s1.z += b.z; s1.w += b.w;
    }
    if (p < p1) {
      const float4 a = io_load4<H>(src, (size_t)((int64_t)(inv ? inv[p] : p) * lds + 4 * q));
      s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
    }
    float4 o = make_float4(quad_sum(s0.x + s1.x), quad_sum(s0.y + s1.y), quad_sum(s0.z + s1.z), quad_sum(s0.w + s1.w));
    if constexpr (LPL == 8) {  // lane 0 of the group: + the quad four lanes up (row_ror:12: lane i reads lane i + 4 of its 16-lane row)
      o.x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(o.x), 0x12C, 0xF, 0xF, false));
      o.y += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(o.y), 0x12C, 0xF, 0xF, false));
      o.z += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(o.z), 0x12C, 0xF, 0xF, false));
      o.w += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(o.w), 0x12C, 0xF, 0xF, false));
    }
    if (l == 0) {
      const size_t de = (size_t)(c * ldo + 4 * q);
      if (accumulate) { const float4 old = io_load4<H>(out, de); o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w; }
      io_store4<H>(out, de, o);
    }
  }
}

// ---- reverse neighbour lists of a K-NN table (round 5: the LFA backward kernels of the 8 / 16-channel layers store their input
// gradient per edge; point j's list = the edges (i, k) with idx[i][k] == j).  The generic m3d_csr_invert_batch would do it in
// count (atomics) + one-workgroup scan + fill (atomics with return): 57 + 168 + 126 us for 204 800 x 16 (profiles/r05q_*).
// Here: ONE pass of atomics that also keeps each edge's rank in its list, a three-launch scan over 4096-element blocks, and a
// fill without atomics.
#define REV_BLK 4096
__global__ __launch_bounds__(256) void rev_rank_kernel(const int32_t* __restrict__ idx, int64_t ne, int32_t n, int32_t* __restrict__ cnt,
                                                       int32_t* __restrict__ rank) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= ne) return;
  const int32_t c = idx[e];
  rank[e] = (c >= 0 && c < n) ? atomicAdd(&cnt[c], 1) : -1;
}
__global__ __launch_bounds__(256) void rev_block_sums_kernel(const int32_t* __restrict__ cnt, int64_t n, int32_t* __restrict__ bsum) {
  __shared__ int wsum[4];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int64_t base = (int64_t)blockIdx.x * REV_BLK + (int64_t)tid * 16;
  int s = 0;
  for (int i = 0; i < 16; ++i) s += base + i < n ? cnt[base + i] : 0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (lane == 0) wsum[wid] = s;
  __syncthreads();
  if (tid == 0) bsum[blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}
__global__ __launch_bounds__(1024) void rev_scan_blocks_kernel(int32_t* __restrict__ bsum, int nb) {  // exclusive, in place
  __shared__ int wsum[16];
  __shared__ int carry;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  if (tid == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < nb; base += 1024) {
    const int v = base + tid < nb ? bsum[base + tid] : 0;
    int x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int y = __shfl_up(x, o, 64);
      if (lane >= o) x += y;
    }
    if (lane == 63) wsum[wid] = x;
    __syncthreads();
    int off = carry;
    for (int w = 0; w < wid; ++w) off += wsum[w];
    if (base + tid < nb) bsum[base + tid] = off + x - v;
    __syncthreads();
    if (tid == 1023) carry = off + x;
    __syncthreads();
  }
}
__global__ __launch_bounds__(256) void rev_ptr_kernel(const int32_t* __restrict__ cnt, int64_t n, const int32_t* __restrict__ boff,
                                                      int32_t* __restrict__ ptr) {
  __shared__ int wsum[4];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int64_t base = (int64_t)blockIdx.x * REV_BLK + (int64_t)tid * 16;
  int v[16], s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) { v[i] = base + i < n ? cnt[base + i] : 0; s += v[i]; }
  int incl = s;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(incl, o, 64);
    if (lane >= o) incl += t;
  }
  if (lane == 63) wsum[wid] = incl;
  __syncthreads();
  int run = boff[blockIdx.x] + incl - s;
  for (int i = 0; i < wid; ++i) run += wsum[i];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    if (base + i < n) ptr[base + i] = run;
    run += v[i];
    if (base + i == n - 1) ptr[n] = run;
  }
}
__global__ __launch_bounds__(256) void rev_fill_kernel(const int32_t* __restrict__ idx, int64_t ne, const int32_t* __restrict__ rank,
                                                       const int32_t* __restrict__ ptr, int32_t* __restrict__ inv,
                                                       int32_t* __restrict__ slot) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= ne) return;
  const int r = rank[e];
  const int s = r >= 0 ? ptr[idx[e]] + r : -1;
  if (inv && r >= 0) inv[s] = (int32_t)e;
  if (slot) slot[e] = s;  // (-1: the edge is in no list)
}
static inline size_t rev_al(size_t b) { return (b + 255) & ~(size_t)255; }
extern "C" size_t m3d_knn_reverse_workspace_bytes(int64_t n, int32_t K) {
  if (n < 0 || K < 1) return 0;
  return rev_al((size_t)n * K * 4) + rev_al((size_t)n * 4) + rev_al((size_t)(m3d_cdiv(n, REV_BLK) + 1) * 4) + 256;
}
extern "C" int m3d_knn_reverse(const int32_t* idx, int64_t n, int32_t K, int32_t* ptr, int32_t* inv, int32_t* slot, void* ws,
                               void* stream) {
  if (n < 0 || K < 1 || n * (int64_t)K > 0x7fffffff) return M3D_ERR_INVALID;
  if (!ptr) return M3D_ERR_INVALID;
  hipStream_t st = (hipStream_t)stream;
  if (n == 0) return hipMemsetAsync(ptr, 0, 4, st) == hipSuccess ? M3D_OK : M3D_ERR_LAUNCH;
  if (!idx || (!inv && !slot) || !ws) return M3D_ERR_INVALID;  // (inv or slot may be NULL: whoever stores rows in list order
                                                               // needs the slots only, and the scattered writes of inv go away)
  const int64_t ne = n * K;
  int32_t* rank = (int32_t*)ws;
  int32_t* cnt = (int32_t*)((char*)ws + rev_al((size_t)ne * 4));
  int32_t* bsum = (int32_t*)((char*)cnt + rev_al((size_t)n * 4));
  const int nb = (int)m3d_cdiv(n, REV_BLK);
  hipLaunchKernelGGL(zero_i32_kernel, dim3((unsigned)m3d_cdiv(n, 1024)), dim3(256), 0, st, cnt, n);
  hipLaunchKernelGGL(rev_rank_kernel, dim3((unsigned)m3d_cdiv(ne, 256)), dim3(256), 0, st, idx, ne, (int32_t)n, cnt, rank);
  hipLaunchKernelGGL(rev_block_sums_kernel, dim3((unsigned)nb), dim3(256), 0, st, (const int32_t*)cnt, n, bsum);
  hipLaunchKernelGGL(rev_scan_blocks_kernel, dim3(1), dim3(1024), 0, st, bsum, nb);
  hipLaunchKernelGGL(rev_ptr_kernel, dim3((unsigned)nb), dim3(256), 0, st, (const int32_t*)cnt, n, (const int32_t*)bsum, ptr);
  hipLaunchKernelGGL(rev_fill_kernel, dim3((unsigned)m3d_cdiv(ne, 256)), dim3(256), 0, st, idx, ne, (const int32_t*)rank,
                     (const int32_t*)ptr, inv, slot);
  M3D_CHECK_LAUNCH();
  return M3D_OK;
}

// [n,3] (row stride `stride` floats) -> [n,4] with w = 0: 16-byte rows for single-load neighbour gathers
__global__ __launch_bounds__(256) void pad_pos_kernel(const float* __restrict__ pos, int stride, float4* __restrict__ out,
                                                      int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float* p = pos + i * stride;
    out[i] = make_float4(p[0], p[1], p[2], 0.f);
  }
}

extern "C" int m3d_pad_pos(const float* pos, int32_t stride, float* out4, int64_t n, void* stream) {
  if (n < 0 || stride < 3) return M3D_ERR_INVALID;
  if (n == 0) return M3D_OK;
  if (!pos || !out4 || (((uintptr_t)out4) & 15)) return M3D_ERR_INVALID;
  int64_t gx = m3d_cdiv(n, 256);
  if (gx > 4096) gx = 4096;
  hipLaunchKernelGGL(pad_pos_kernel, dim3((unsigned)gx), dim3(256), 0, (hipStream_t)stream, pos, stride, (float4*)out4,
                     n);
  M3D_CHECK_LAUNCH();
  return M3D_OK;
}

// ------------------------------------------------------------------------------------------
// decimation indices
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t feistel_perm(uint32_t r, uint32_t n, uint32_t key) {
  if (n <= 1) return 0;
  int bits = 32 - __clz(n - 1);  // ceil(log2 n), n >= 2
  int half = (bits + 1) >> 1;
  uint32_t mask = (1u << half) - 1u;
  uint32_t x = r;
  do {
    uint32_t L = x >> half, R = x & mask;
#pragma unroll
    for (int round = 0; round < 6; ++round) {
      uint32_t f = mix32(R ^ (key + 0x9e3779b9u * (uint32_t)(round + 1))) & mask;
      uint32_t t = L ^ f;
      L = R;
      R = t;
    }
    x = (L << half) | R;
  } while (x >= n);
  return x;
}

// ------------------------------------------------------------------------------------------
// torch.nn.Dropout(p) in train mode (the second layer of mlp_classif: MLP(..., dropout=[0.0, 0.5]),
// /root/reference/myria3d/models/modules/pyg_randla_net.py:49-52): y = x * keep / (1 - p), keep ~ Bernoulli(1 - p).
// Counter-based: keep(i) is a hash of (seed, step counter, i), so the backward pass recomputes the mask with the same
// launch (dx = dy * keep / (1 - p)) instead of storing it, and a replayed hipGraph draws a new mask every step because the
// step counter lives on the device (torch's own dropout under graph capture costs two extra fill launches per replay
// for its Philox state).  16 random bits per element: P(keep) is 1 - p rounded to 2^-16.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dropout_kernel(const float4* __restrict__ x, float4* __restrict__ y, int64_t n4,
                                                      DropArgs d) {
  const uint32_t key = drop_key(d);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const float4 v = x[i], m = drop_mul4(key, i, d.thr16, d.scale);
    y[i] = make_float4(v.x * m.x, v.y * m.y, v.z * m.z, v.w * m.w);
  }
}

extern "C" int m3d_dropout(const float* x, float* y, int64_t n, float p, const int64_t* counter, uint64_t seed,
                           void* stream) {
  if (n < 0 || (n & 3) || !(p >= 0.f && p < 1.f)) return M3D_ERR_INVALID;
  if (n == 0) return M3D_OK;
  if (!x || !y || !counter || ((((uintptr_t)x) | ((uintptr_t)y)) & 15)) return M3D_ERR_INVALID;
  const M3DDropout md{counter, seed, p, nullptr, nullptr};
  DropArgs d = drop_args(&md);
  if (d.thr16 == 0) { d.counter = counter; d.seed = seed; }  // p rounds to 0: the identity (every 16-bit draw is >= 0)
  int64_t gx = m3d_cdiv(n / 4, 256 * 4);
  if (gx > 4096) gx = 4096;
  if (gx < 1) gx = 1;
  hipLaunchKernelGGL(dropout_kernel, dim3((unsigned)gx), dim3(256), 0, (hipStream_t)stream, (const float4*)x, (float4*)y,
                     n / 4, d);
  M3D_CHECK_LAUNCH();
  return M3D_OK;
}

__global__ __launch_bounds__(256) void decimation_kernel(const int64_t* __restrict__ ptr,
                                                         const int64_t* __restrict__ ptr_out, int B,
                                                         const uint64_t* __restrict__ seed, uint32_t level,
                                                         int32_t* __restrict__ idx_out, int64_t m) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= m) return;
  int lo = 0, hi = B;
  while (hi - lo > 1) {
    int mid = (lo + hi) >> 1;
    if (ptr_out[mid] <= t) lo = mid; else hi = mid;
  }
  const int b = lo;
  uint32_t r = (uint32_t)(t - ptr_out[b]);
  const uint32_t n = (uint32_t)(ptr[b + 1] - ptr[b]);
  const uint64_t s = seed[0];
  uint32_t key = mix32((uint32_t)s ^ mix32((uint32_t)(s >> 32) + 0x85ebca6bu * (level + 1u)) ^ mix32((uint32_t)b * 0xc2b2ae35u + 1u));
  if (r >= n && n > 0) {
    // more slots than points (MinimumNumNodes, transforms.py:66-84): concatenated independent permutations
    const uint32_t rep = r / n;
    r -= rep * n;
    key = mix32(key ^ (rep * 0x27d4eb2fu));
  }
  idx_out[t] = (int32_t)(ptr[b] + (int64_t)feistel_perm(r, n, key));
}

// decimate() of one level in ONE launch (pyg_randla_net.py:234-238 with the indices of :192-231): slot t of the next level
// draws its reference row d_ref (or takes it from d_ref_in), maps it to this level's cell-sorted slot d_int = inv[d_ref]
// and fetches that slot's position record.  Same draws as m3d_decimation_indices; replaces that launch + two gathers.
__global__ __launch_bounds__(256) void decimate_level_kernel(const int64_t* __restrict__ ptr,
                                                             const int64_t* __restrict__ ptr_out, int B,
                                                             const uint64_t* __restrict__ seed, uint32_t level,
                                                             const int32_t* __restrict__ d_ref_in,
                                                             const int32_t* __restrict__ inv,
                                                             const float4* __restrict__ pos4,
                                                             int32_t* __restrict__ d_ref_out, int32_t* __restrict__ d_int_out,
                                                             float4* __restrict__ pos_out, int64_t m) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= m) return;
  int32_t d_ref;
  if (d_ref_in) {
    d_ref = d_ref_in[t];
  } else {
    int lo = 0, hi = B;
    while (hi - lo > 1) {
      int mid = (lo + hi) >> 1;
      if (ptr_out[mid] <= t) lo = mid; else hi = mid;
    }
    const int b = lo;
    uint32_t r = (uint32_t)(t - ptr_out[b]);
    const uint32_t n = (uint32_t)(ptr[b + 1] - ptr[b]);
    const uint64_t s = seed[0];
    uint32_t key = mix32((uint32_t)s ^ mix32((uint32_t)(s >> 32) + 0x85ebca6bu * (level + 1u)) ^ mix32((uint32_t)b * 0xc2b2ae35u + 1u));
    if (r >= n && n > 0) {
      const uint32_t rep = r / n;
      r -= rep * n;
      key = mix32(key ^ (rep * 0x27d4eb2fu));
    }
    d_ref = (int32_t)(ptr[b] + (int64_t)feistel_perm(r, n, key));
    d_ref_out[t] = d_ref;
  }
  const int32_t d_int = inv[d_ref];
  d_int_out[t] = d_int;
  pos_out[t] = pos4[d_int];
}

extern "C" int m3d_decimate_level(const int64_t* ptr, const int64_t* ptr_out, int32_t num_clouds, const uint64_t* seed,
                                  uint32_t level, const int32_t* d_ref_in, const int32_t* inv, const float* pos4,
                                  int32_t* d_ref_out, int32_t* d_int_out, float* pos4_out, int64_t m, void* stream) {
  if (num_clouds < 0 || m < 0) return M3D_ERR_INVALID;
  if (m == 0 || num_clouds == 0) return M3D_OK;
  if (!inv || !pos4 || !d_int_out || !pos4_out) return M3D_ERR_INVALID;
  if (!d_ref_in && (!ptr || !ptr_out || !seed || !d_ref_out)) return M3D_ERR_INVALID;
  if ((((uintptr_t)pos4) | ((uintptr_t)pos4_out)) & 15) return M3D_ERR_INVALID;
  hipLaunchKernelGGL(decimate_level_kernel, dim3((unsigned)m3d_cdiv(m, 256)), dim3(256), 0, (hipStream_t)stream, ptr,
                     ptr_out, num_clouds, seed, level, d_ref_in, inv, (const float4*)pos4, d_ref_out, d_int_out,
                     (float4*)pos4_out, m);
  M3D_CHECK_LAUNCH();
  return M3D_OK;
}

extern "C" int m3d_decimation_indices(const int64_t* ptr, const int64_t* ptr_out, int32_t num_clouds,
                                      const uint64_t* seed, uint32_t level, int32_t* idx_out, int64_t m,
                                      void* stream) {
  if (num_clouds < 0 || m < 0) return M3D_ERR_INVALID;
  if (m == 0 || num_clouds == 0) return M3D_OK;
  if (!ptr || !ptr_out || !seed || !idx_out) return M3D_ERR_INVALID;
  hipLaunchKernelGGL(decimation_kernel, dim3((unsigned)m3d_cdiv(m, 256)), dim3(256), 0, (hipStream_t)stream, ptr,
                     ptr_out, num_clouds, seed, level, idx_out, m);
  M3D_CHECK_LAUNCH();
  return M3D_OK;
}

// ------------------------------------------------------------------------------------------
// many small device-to-device copies in ONE launch.  A hipGraph replays its nodes at ~5-10 us apiece when they are
// tiny (memcpy nodes ~9 us: profiles/r02f_step_timeline.csv), so the ~25 buffer copies that move a prefetched geometry
// into its persistent slot cost more submission time than the kernels they follow.  dst / src are 16-byte aligned.
// ------------------------------------------------------------------------------------------
#define M3D_COPY_MANY_MAX 48
struct CopyManyArgs {
  void* dst[M3D_COPY_MANY_MAX];
  const void* src[M3D_COPY_MANY_MAX];
  int64_t bytes[M3D_COPY_MANY_MAX];
};
__global__ __launch_bounds__(256) void copy_many_kernel(CopyManyArgs a) {
  const int e = blockIdx.y;
  const int64_t nb = a.bytes[e];
  const int64_t n16 = nb >> 4;
  const uint4* s = (const uint4*)a.src[e];
  uint4* d = (uint4*)a.dst[e];
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) d[i] = s[i];
  if (blockIdx.x == 0) {
    const int64_t tail = nb & 15;
    if ((int64_t)threadIdx.x < tail) ((char*)a.dst[e])[(n16 << 4) + threadIdx.x] = ((const char*)a.src[e])[(n16 << 4) + threadIdx.x];
  }
}

extern "C" int m3d_copy_many(void* const* dst, const void* const* src, const int64_t* bytes, int32_t count, void* stream) {
  if (count < 0 || count > M3D_COPY_MANY_MAX) return M3D_ERR_UNSUPPORTED;
  if (count == 0) return M3D_OK;
  if (!dst || !src || !bytes) return M3D_ERR_INVALID;
  CopyManyArgs a;
  int64_t mx = 0;
  for (int i = 0; i < count; ++i) {
    if (bytes[i] < 0 || (bytes[i] > 0 && (!dst[i] || !src[i]))) return M3D_ERR_INVALID;
    if ((((uintptr_t)dst[i]) | ((uintptr_t)src[i])) & 15) return M3D_ERR_INVALID;
    a.dst[i] = dst[i]; a.src[i] = src[i]; a.bytes[i] = bytes[i];
    if (bytes[i] > mx) mx = bytes[i];
  }
  int64_t gx = m3d_cdiv(mx, 256 * 16 * 4);  // ~4 x 16 bytes per thread for the largest buffer
  if (gx > 1024) gx = 1024;
  if (gx < 1) gx = 1;
  hipLaunchKernelGGL(copy_many_kernel, dim3((unsigned)gx, (unsigned)count), dim3(256), 0, (hipStream_t)stream, a);
  M3D_CHECK_LAUNCH();
  return M3D_OK;
}
