// Voxel-grid subsampling of Lidar tiles on the device (gfx950).
//
// Replaces torch_geometric.transforms.GridSampling(0.25) as configured by the reference's data preparation
// (/root/reference/configs/datamodule/transforms/preparations/points_budget.yaml:14-17, 50-53, 78-81), which runs on
// the CPU in the dataloader workers, one tile at a time:
//   c = voxel_grid(pos, size)            cluster id = sum_d trunc((pos_d - min_d) / size) * stride_d   (x fastest)
//   c, perm = consecutive_cluster(c)     ids renumbered in ascending order of the occupied voxels
//   pos, x  -> scatter(.., c, reduce="mean");   y -> one_hot -> scatter sum -> argmax  (majority, first maximum)
// Here all tiles of a batch go through ONE pass (every tile keeps its own bounding box, i.e. exactly the per-tile
// semantics): 64-bit keys (tile << 40 | cluster id), a stable LSD radix sort of (key, point) pairs, head flags +
// prefix sum = consecutive ids, then one lane per voxel reduces its points IN ORIGINAL ORDER (the sort is stable), so
// the fp32 sums are accumulated in the same order as a sequential scatter_add and the means are bit-comparable.
// Nothing is atomic on floats; the result is deterministic.
#include "m3d_common.h"
#include "../../include/m3d_hip.h"

typedef unsigned long long u64;

#define RS_TILE 2048   // keys per radix-sort workgroup (one wavefront)
#define SC_TILE 1024   // elements per head-flag scan workgroup
#define VOX_CLUSTER_BITS 40

struct VoxWs {
  float* start;      // [B][4]   per-tile minimum (x, y, z, -)
  int64_t* stride;   // [B][4]   1, n0, n0*n1, n0*n1*n2
  u64* keys[2];      // [n] ping-pong
  int32_t* vals[2];  // [n] ping-pong (original point index)
  int32_t* hist;     // [256 * nblk]
  int32_t* vid;      // [n]     voxel id of the i-th sorted point
  int32_t* bsum;     // [nsc + 1]
  int32_t* seg;      // [n + 1] first sorted position of voxel v
  int32_t* misc;     // [4]     0: number of voxels, 1: error flag
  int32_t* scan;     // block sums of the multi-workgroup prefix sums (exscan_launch)
};

static inline size_t al256(size_t b) { return (b + 255) & ~(size_t)255; }

static VoxWs vox_carve(void* ws, int64_t n, int B) {
  VoxWs w;
  char* p = (char*)ws;
  const int64_t nblk = m3d_cdiv(n, RS_TILE), nsc = m3d_cdiv(n, SC_TILE);
  w.start = (float*)p; p += al256((size_t)B * 4 * sizeof(float));
  w.stride = (int64_t*)p; p += al256((size_t)B * 4 * sizeof(int64_t));
  for (int i = 0; i < 2; ++i) { w.keys[i] = (u64*)p; p += al256((size_t)n * sizeof(u64)); }
  for (int i = 0; i < 2; ++i) { w.vals[i] = (int32_t*)p; p += al256((size_t)n * sizeof(int32_t)); }
  w.hist = (int32_t*)p; p += al256((size_t)256 * nblk * sizeof(int32_t));
  w.vid = (int32_t*)p; p += al256((size_t)n * sizeof(int32_t));
  w.bsum = (int32_t*)p; p += al256((size_t)(nsc + 1) * sizeof(int32_t));
  w.seg = (int32_t*)p; p += al256((size_t)(n + 1) * sizeof(int32_t));
  w.misc = (int32_t*)p; p += 256;
  w.scan = (int32_t*)p;
  return w;
}

extern "C" size_t m3d_grid_sampling_workspace_bytes(int64_t n, int32_t num_clouds) {
  if (n < 0 || num_clouds < 0) return 0;
  const int64_t nblk = m3d_cdiv(n, RS_TILE), nsc = m3d_cdiv(n, SC_TILE);
  return al256((size_t)num_clouds * 16) + al256((size_t)num_clouds * 32) + 2 * al256((size_t)n * 8) +
         2 * al256((size_t)n * 4) + al256((size_t)256 * nblk * 4) + al256((size_t)n * 4) +
         al256((size_t)(nsc + 1) * 4) + al256((size_t)(n + 1) * 4) + 256 + 256 +
         al256((size_t)((256 * nblk > nsc ? 256 * nblk : nsc) / 4096 + 2) * 4);
}

// ------------------------------------------------------------------------------------------
// per-tile bounding box -> voxel grid (one 1024-thread workgroup per tile)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void vox_bounds_kernel(const float* __restrict__ pos, int pstride,
                                                          const int64_t* __restrict__ ptr, float size, VoxWs w) {
  __shared__ float red[6][16];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int64_t s0 = ptr[b], n = ptr[b + 1] - s0;
  float mn[3] = {3.4e38f, 3.4e38f, 3.4e38f}, mx[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
  for (int64_t i = tid; i < n; i += 1024) {
    const float* p = pos + (s0 + i) * pstride;
#pragma unroll
    for (int d = 0; d < 3; ++d) { mn[d] = fminf(mn[d], p[d]); mx[d] = fmaxf(mx[d], p[d]); }
  }
#pragma unroll
  for (int d = 0; d < 3; ++d) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      mn[d] = fminf(mn[d], __shfl_xor(mn[d], o, 64));
      mx[d] = fmaxf(mx[d], __shfl_xor(mx[d], o, 64));
    }
    if (lane == 0) { red[d][wid] = mn[d]; red[3 + d][wid] = mx[d]; }
  }
  __syncthreads();
  if (tid == 0) {
    int64_t nv[3];
    for (int d = 0; d < 3; ++d) {
      float lo = red[d][0], hi = red[3 + d][0];
      for (int i = 1; i < 16; ++i) { lo = fminf(lo, red[d][i]); hi = fmaxf(hi, red[3 + d][i]); }
      if (n <= 0) { lo = 0.f; hi = 0.f; }
      w.start[b * 4 + d] = lo;
      nv[d] = (int64_t)((hi - lo) / size) + 1;  // torch_cluster grid: ((end - start) / size).long() + 1
    }
    w.start[b * 4 + 3] = 0.f;
    w.stride[b * 4 + 0] = 1;
    w.stride[b * 4 + 1] = nv[0];
    w.stride[b * 4 + 2] = nv[0] * nv[1];
    w.stride[b * 4 + 3] = nv[0] * nv[1] * nv[2];
    // cluster ids must fit the key layout (40 bits): flag instead of wrapping
    const double tot = (double)nv[0] * (double)nv[1] * (double)nv[2];
    if (tot >= (double)(1ull << VOX_CLUSTER_BITS)) w.misc[1] = 1;
  }
}

__global__ __launch_bounds__(256) void vox_keys_kernel(const float* __restrict__ pos, int pstride,
                                                       const int64_t* __restrict__ ptr, int B, int64_t n, float size,
                                                       VoxWs w) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  int lo = 0, hi = B;
  while (hi - lo > 1) {
    int mid = (lo + hi) >> 1;
    if (ptr[mid] <= i) lo = mid; else hi = mid;
  }
  const int b = lo;
  const float* p = pos + i * pstride;
  u64 c = 0;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const float rel = p[d] - w.start[b * 4 + d];
    const int64_t q = (int64_t)(rel / size);  // (pos - start).true_divide(size).long(): truncation, rel >= 0
    c += (u64)q * (u64)w.stride[b * 4 + d];
  }
  w.keys[0][i] = ((u64)b << VOX_CLUSTER_BITS) | (c & ((1ull << VOX_CLUSTER_BITS) - 1ull));
  w.vals[0][i] = (int32_t)i;
}

// ------------------------------------------------------------------------------------------
// stable LSD radix sort, 8 bits per pass, one wavefront per RS_TILE keys
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void rs_hist_kernel(const u64* __restrict__ keys, int64_t n, int shift,
                                                     int32_t* __restrict__ gh, int nblk) {
  __shared__ int h[256];
  const int tid = threadIdx.x, blk = blockIdx.x;
  for (int d = tid; d < 256; d += 64) h[d] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blk * RS_TILE;
  for (int i = tid; i < RS_TILE && base + i < n; i += 64) atomicAdd(&h[(int)((keys[base + i] >> shift) & 255ull)], 1);
  __syncthreads();
  for (int d = tid; d < 256; d += 64) gh[(size_t)d * nblk + blk] = h[d];
}

// exclusive prefix sum of a[0..len) in place by ONE workgroup; total (optional) receives the sum
__global__ __launch_bounds__(1024) void exscan_kernel(int32_t* __restrict__ a, int64_t len, int32_t* __restrict__ total) {
  __shared__ int wsum[16];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int64_t chunk = (len + 1023) / 1024;
  const int64_t b = tid * chunk, e = b + chunk < len ? b + chunk : len;
  int s = 0;
  for (int64_t i = b; i < e; ++i) s += a[i];
  int incl = s;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int t = __shfl_up(incl, o, 64);
    if (lane >= o) incl += t;
  }
  if (lane == 63) wsum[wid] = incl;
  __syncthreads();
  int woff = 0;
  for (int i = 0; i < wid; ++i) woff += wsum[i];
  int run = woff + incl - s;
  for (int64_t i = b; i < e; ++i) {
    int v = a[i];
    a[i] = run;
    run += v;
  }
  if (total && tid == 1023) *total = woff + incl;
}

// ---- the same over many workgroups (round 5).  exscan_kernel walks its array with ONE workgroup whose threads each sum a
// contiguous chunk (stride-`chunk` accesses across the lanes): 0.37 ms for the 156 k radix histogram entries of a 1.25 M-point
// batch, seven times per GridSampling call, and 3.9 ms for the 1.95 M (sample, chunk) counters of m3d_tile_select on a
// 10 M-point cloud — 24.7 of the 65 ms of the predict chain (profiles/r05g_predict_trace_summary.log).  Three phases over
// SCAN_BLK-element blocks with coalesced 16-byte accesses: block sums, their scan (one workgroup: a few hundred entries), then
// the per-block scan with its offset.  `scratch`: cdiv(len, SCAN_BLK) + 1 ints.
#define SCAN_BLK 4096  // elements per 256-thread workgroup: 16 per thread
__global__ __launch_bounds__(256) void scan_reduce_kernel(const int32_t* __restrict__ a, int64_t len, int32_t* __restrict__ bsum) {
  __shared__ int wsum[4];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int64_t base = (int64_t)blockIdx.x * SCAN_BLK + (int64_t)tid * 16;
  int s = 0;
  if (base + 16 <= len) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int4 v = *(const int4*)(a + base + 4 * q);
      s += (v.x + v.y) + (v.z + v.w);
    }
  } else {
    for (int64_t i = base; i < len && i < base + 16; ++i) s += a[i];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (lane == 0) wsum[wid] = s;
  __syncthreads();
  if (tid == 0) bsum[blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}
__global__ __launch_bounds__(256) void scan_apply_kernel(int32_t* __restrict__ a, int64_t len, const int32_t* __restrict__ boff) {
  __shared__ int wsum[4];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int64_t base = (int64_t)blockIdx.x * SCAN_BLK + (int64_t)tid * 16;
  int v[16];
  const bool whole = base + 16 <= len;
  if (whole) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int4 t = *(const int4*)(a + base + 4 * q);
      v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
    }
  } else {
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = base + i < len ? a[base + i] : 0;
  }
  int s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += v[i];
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
  int o[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) { o[i] = run; run += v[i]; }
  if (whole) {
#pragma unroll
    for (int q = 0; q < 4; ++q) *(int4*)(a + base + 4 * q) = make_int4(o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]);
  } else {
#pragma unroll
    for (int i = 0; i < 16; ++i)
      if (base + i < len) a[base + i] = o[i];
  }
}
static inline size_t scan_scratch_bytes(int64_t len) { return al256((size_t)(m3d_cdiv(len > 0 ? len : 1, SCAN_BLK) + 1) * 4); }
// exclusive prefix sum of a[0..len) in place (a 16-byte aligned); *total (optional) receives the sum
static void exscan_launch(int32_t* a, int64_t len, int32_t* total, int32_t* scratch, hipStream_t st) {
  if (len <= 2 * SCAN_BLK || !scratch) {
    hipLaunchKernelGGL(exscan_kernel, dim3(1), dim3(1024), 0, st, a, len, total);
    return;
  }
  const int64_t nb = m3d_cdiv(len, SCAN_BLK);
  hipLaunchKernelGGL(scan_reduce_kernel, dim3((unsigned)nb), dim3(256), 0, st, (const int32_t*)a, len, scratch);
  hipLaunchKernelGGL(exscan_kernel, dim3(1), dim3(1024), 0, st, scratch, nb, total);
  hipLaunchKernelGGL(scan_apply_kernel, dim3((unsigned)nb), dim3(256), 0, st, a, len, (const int32_t*)scratch);
}

__global__ __launch_bounds__(64) void rs_scatter_kernel(const u64* __restrict__ kin, const int32_t* __restrict__ vin,
                                                        u64* __restrict__ kout, int32_t* __restrict__ vout, int64_t n,
                                                        int shift, const int32_t* __restrict__ gh, int nblk) {
  __shared__ int base[256];
  const int lane = threadIdx.x, blk = blockIdx.x;
  for (int d = lane; d < 256; d += 64) base[d] = gh[(size_t)d * nblk + blk];
  __syncthreads();
  const u64 lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));  // lanes below this one
  for (int i0 = 0; i0 < RS_TILE; i0 += 64) {
    const int64_t idx = (int64_t)blk * RS_TILE + i0 + lane;
    const bool valid = idx < n;
    const u64 key = valid ? kin[idx] : 0ull;
    const int val = valid ? vin[idx] : 0;
    const int d = (int)((key >> shift) & 255ull);
    u64 peers = __ballot(valid);
#pragma unroll
    for (int bit = 0; bit < 8; ++bit) {
      const bool one = (d >> bit) & 1;
      const u64 m = __ballot(valid && one);
      peers &= one ? m : ~m;
    }
    const int rank = __popcll(peers & lt), cnt = __popcll(peers);
    const int b = valid ? base[d] : 0;
    __syncthreads();  // every lane has read its base before the group leaders advance it
    if (valid) {
      kout[b + rank] = key;
      vout[b + rank] = val;
      if (rank == 0) base[d] = b + cnt;
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------
// head flags -> consecutive voxel ids, segment starts, new ptr
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ bool vox_head(const u64* __restrict__ k, int64_t i) { return i == 0 || k[i] != k[i - 1]; }

__global__ __launch_bounds__(256) void vox_flagsum_kernel(const u64* __restrict__ keys, int64_t n,
                                                          int32_t* __restrict__ bsum) {
  __shared__ int ws4[4];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int64_t base = (int64_t)blockIdx.x * SC_TILE;
  int s = 0;
  for (int j = 0; j < SC_TILE / 256; ++j) {
    const int64_t i = base + tid + j * 256;
    if (i < n && vox_head(keys, i)) ++s;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (lane == 0) ws4[wid] = s;
  __syncthreads();
  if (tid == 0) bsum[blockIdx.x] = ws4[0] + ws4[1] + ws4[2] + ws4[3];
}

__global__ __launch_bounds__(256) void vox_assign_kernel(const u64* __restrict__ keys, int64_t n,
                                                         const int32_t* __restrict__ bsum, int32_t* __restrict__ vid,
                                                         int32_t* __restrict__ seg, int32_t* __restrict__ misc) {
  __shared__ int ws4[4];
  __shared__ int carry;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int64_t base = (int64_t)blockIdx.x * SC_TILE;
  if (tid == 0) carry = bsum[blockIdx.x];  // heads before this workgroup's elements
  __syncthreads();
  for (int j = 0; j < SC_TILE / 256; ++j) {  // elements in order: sub-tile j holds base + j*256 .. + 255
    const int64_t i = base + j * 256 + tid;
    const int f = (i < n && vox_head(keys, i)) ? 1 : 0;
    int incl = f;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      int t = __shfl_up(incl, o, 64);
      if (lane >= o) incl += t;
    }
    if (lane == 63) ws4[wid] = incl;
    __syncthreads();
    int woff = 0;
    for (int q = 0; q < wid; ++q) woff += ws4[q];
    const int c0 = carry;
    const int v = c0 + woff + incl - 1;  // id of the voxel element i belongs to
    if (i < n) {
      vid[i] = v;
      if (f) seg[v] = (int32_t)i;
      if (i == n - 1) { seg[v + 1] = (int32_t)n; misc[0] = v + 1; }
    }
    __syncthreads();
    if (tid == 255) carry = c0 + woff + incl;
    __syncthreads();
  }
}

__global__ void vox_ptr_kernel(const int64_t* __restrict__ ptr, int B, int64_t n, const int32_t* __restrict__ vid,
                               const int32_t* __restrict__ misc, int64_t* __restrict__ out_ptr) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b > B) return;
  const int64_t p = ptr[b];
  out_ptr[b] = p < n ? (int64_t)vid[p] : (int64_t)(n > 0 ? misc[0] : 0);
}

// ------------------------------------------------------------------------------------------
// one lane per voxel: means of pos / x in original point order, majority label
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vox_reduce_kernel(const float* __restrict__ pos, int pstride,
                                                         const float* __restrict__ x, int64_t ldx, int F,
                                                         const int64_t* __restrict__ y,
                                                         const int32_t* __restrict__ vals,
                                                         const int32_t* __restrict__ seg,
                                                         const int32_t* __restrict__ misc, float* __restrict__ out_pos,
                                                         float* __restrict__ out_x, int64_t* __restrict__ out_y) {
  const int M = misc[0];
  for (int v = blockIdx.x * 256 + threadIdx.x; v < M; v += gridDim.x * 256) {
    const int s = seg[v], e = seg[v + 1];
    const float cnt = (float)(e - s);
    float ps0 = 0.f, ps1 = 0.f, ps2 = 0.f;
    for (int i = s; i < e; ++i) {
      const float* p = pos + (int64_t)vals[i] * pstride;
      ps0 += p[0]; ps1 += p[1]; ps2 += p[2];
    }
    out_pos[(int64_t)v * 3 + 0] = ps0 / cnt;
    out_pos[(int64_t)v * 3 + 1] = ps1 / cnt;
    out_pos[(int64_t)v * 3 + 2] = ps2 / cnt;
    if (x) {
      for (int c0 = 0; c0 < F; c0 += 16) {
        float acc[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) acc[c] = 0.f;
        for (int i = s; i < e; ++i) {
          const float* r = x + (int64_t)vals[i] * ldx + c0;
#pragma unroll
          for (int c = 0; c < 16; ++c)
            if (c0 + c < F) acc[c] += r[c];
        }
#pragma unroll
        for (int c = 0; c < 16; ++c)
          if (c0 + c < F) out_x[(int64_t)v * F + c0 + c] = acc[c] / cnt;
      }
    }
    if (y) {
      // argmax over the class histogram, first maximum = smallest label among the most frequent ones
      int64_t best = 0;
      int bestc = 0;
      if (e - s <= 64) {
        for (int i = s; i < e; ++i) {
          const int64_t yi = y[vals[i]];
          int c = 0;
          for (int j = s; j < e; ++j) c += (y[vals[j]] == yi);
          if (c > bestc || (c == bestc && yi < best)) { bestc = c; best = yi; }
        }
      } else {
        int64_t ymax = 0;
        for (int i = s; i < e; ++i) { const int64_t yi = y[vals[i]]; ymax = yi > ymax ? yi : ymax; }
        for (int64_t cls = 0; cls <= ymax; ++cls) {
          int c = 0;
          for (int j = s; j < e; ++j) c += (y[vals[j]] == cls);
          if (c > bestc) { bestc = c; best = cls; }
        }
      }
      out_y[v] = best;
    }
  }
}

extern "C" int m3d_grid_sampling(const float* pos, int32_t pos_stride, const float* x, int64_t ldx, int32_t F,
                                 const int64_t* y, const int64_t* ptr, int32_t num_clouds, int64_t n, float size,
                                 void* ws, float* out_pos, float* out_x, int64_t* out_y, int64_t* out_ptr,
                                 void* stream) {
  if (n < 0 || num_clouds < 0 || F < 0 || !(size > 0.f)) return M3D_ERR_INVALID;
  if (!ptr || !ws || !out_ptr) return M3D_ERR_INVALID;
  if (n >= (int64_t)1 << 31 || num_clouds >= (1 << 16)) return M3D_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  VoxWs w = vox_carve(ws, n, num_clouds);
  if (n == 0 || num_clouds == 0) {
    if (hipMemsetAsync(out_ptr, 0, (size_t)(num_clouds + 1) * sizeof(int64_t), st) != hipSuccess) return M3D_ERR_LAUNCH;
    return M3D_OK;
  }
  if (!pos || pos_stride < 3 || !out_pos || (x && (!out_x || ldx < F)) || (y && !out_y)) return M3D_ERR_INVALID;
  if (hipMemsetAsync(w.misc, 0, 16, st) != hipSuccess) return M3D_ERR_LAUNCH;
  hipLaunchKernelGGL(vox_bounds_kernel, dim3(num_clouds), dim3(1024), 0, st, pos, pos_stride, ptr, size, w);
  hipLaunchKernelGGL(vox_keys_kernel, dim3((unsigned)m3d_cdiv(n, 256)), dim3(256), 0, st, pos, pos_stride, ptr,
                     num_clouds, n, size, w);
  // sort on the bits that can differ: 40 cluster bits + the tile bits
  int tile_bits = 0;
  while ((1 << tile_bits) < num_clouds) ++tile_bits;
  const int passes = (VOX_CLUSTER_BITS + tile_bits + 7) / 8;
  const int nblk = (int)m3d_cdiv(n, RS_TILE);
  int cur = 0;
  for (int p = 0; p < passes; ++p) {
    hipLaunchKernelGGL(rs_hist_kernel, dim3(nblk), dim3(64), 0, st, w.keys[cur], n, p * 8, w.hist, nblk);
    exscan_launch(w.hist, (int64_t)256 * nblk, nullptr, w.scan, st);
    hipLaunchKernelGGL(rs_scatter_kernel, dim3(nblk), dim3(64), 0, st, w.keys[cur], w.vals[cur], w.keys[cur ^ 1],
                       w.vals[cur ^ 1], n, p * 8, w.hist, nblk);
    cur ^= 1;
  }
  const int nsc = (int)m3d_cdiv(n, SC_TILE);
  hipLaunchKernelGGL(vox_flagsum_kernel, dim3(nsc), dim3(256), 0, st, w.keys[cur], n, w.bsum);
  exscan_launch(w.bsum, (int64_t)nsc, nullptr, w.scan, st);
  hipLaunchKernelGGL(vox_assign_kernel, dim3(nsc), dim3(256), 0, st, w.keys[cur], n, w.bsum, w.vid, w.seg, w.misc);
  hipLaunchKernelGGL(vox_ptr_kernel, dim3((unsigned)m3d_cdiv(num_clouds + 1, 256)), dim3(256), 0, st, ptr, num_clouds,
                     n, w.vid, w.misc, out_ptr);
  int64_t gx = m3d_cdiv(n, 256);
  if (gx > 4096) gx = 4096;
  hipLaunchKernelGGL(vox_reduce_kernel, dim3((unsigned)gx), dim3(256), 0, st, pos, pos_stride, x, ldx, F, y,
                     w.vals[cur], w.seg, w.misc, out_pos, x ? out_x : nullptr, y ? out_y : nullptr);
  M3D_CHECK_LAUNCH();
  return M3D_OK;
}

// 0 = ok, 1 = a tile's voxel grid has >= 2^40 cells (size too small for its extent): results invalid
extern "C" int m3d_grid_sampling_status(const void* ws, int64_t n, int32_t num_clouds, int32_t* status_dev_out,
                                        void* stream) {
  if (!ws || !status_dev_out) return M3D_ERR_INVALID;
  VoxWs w = vox_carve((void*)ws, n, num_clouds);
  if (hipMemcpyAsync(status_dev_out, w.misc + 1, sizeof(int32_t), hipMemcpyDeviceToDevice, (hipStream_t)stream) !=
      hipSuccess)
    return M3D_ERR_LAUNCH;
  return M3D_OK;
}

// ------------------------------------------------------------------------------------------
// Per-tile normalisations of the reference's data preparation, fused into two launches for a whole batch:
//   torch_geometric.transforms.Center            pos -= mean(pos)                       (points_budget.yaml:29-30)
//   NullifyLowestZ                               z   -= min(z)                          (transforms.py:141-146)
//   NormalizePos                                 pos *= 1 / (subtile_width / 2)         (transforms.py:149-162)
//   StandardizeRGBAndIntensity                   Intensity: v = log(v + 1); both channels: s = std + 1e-6,
//                                                clamp((v - mean) / s, -3 s, 3 s)       (transforms.py:115-138)
// stats (fp64 [B][8]): sum x, sum y, sum z, min z, then per channel sum v, sum v^2 — one workgroup per tile.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void tile_stats_kernel(const float* __restrict__ pos, int pstride,
                                                          const float* __restrict__ x, int64_t ldx, int ci, int cr,
                                                          const int64_t* __restrict__ ptr, double* __restrict__ stats) {
  __shared__ double red[8][16];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int64_t s0 = ptr[b], n = ptr[b + 1] - s0;
  double a[8] = {0., 0., 0., 3.4e38, 0., 0., 0., 0.};
  for (int64_t i = tid; i < n; i += 1024) {
    const float* p = pos + (s0 + i) * pstride;
    a[0] += p[0]; a[1] += p[1]; a[2] += p[2];
    a[3] = fmin(a[3], (double)p[2]);
    if (x) {
      const float* r = x + (s0 + i) * ldx;
      if (ci >= 0) { const double v = (double)logf(r[ci] + 1.f); a[4] += v; a[5] += v * v; }
      if (cr >= 0) { const double v = (double)r[cr]; a[6] += v; a[7] += v * v; }
    }
  }
#pragma unroll
  for (int q = 0; q < 8; ++q) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const double t = __shfl_xor(a[q], o, 64);
      a[q] = q == 3 ? fmin(a[q], t) : a[q] + t;
    }
    if (lane == 0) red[q][wid] = a[q];
  }
  __syncthreads();
  if (tid < 8) {
    double v = red[tid][0];
    for (int i = 1; i < 16; ++i) v = tid == 3 ? fmin(v, red[tid][i]) : v + red[tid][i];
    stats[b * 8 + tid] = v;
  }
}

__global__ __launch_bounds__(256) void tile_apply_kernel(float* __restrict__ pos, int pstride, float* __restrict__ x,
                                                         int64_t ldx, int ci, int cr,
                                                         const int64_t* __restrict__ ptr, int B, int64_t n, int center,
                                                         int nullify_z, float pos_scale, float clamp_sigma,
                                                         const double* __restrict__ stats) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  int lo = 0, hi = B;
  while (hi - lo > 1) {
    int mid = (lo + hi) >> 1;
    if (ptr[mid] <= i) lo = mid; else hi = mid;
  }
  const int b = lo;
  const double cnt = (double)(ptr[b + 1] - ptr[b]);
  const double* st = stats + b * 8;
  float* p = pos + i * pstride;
  float px = p[0], py = p[1], pz = p[2];
  float zmin = (float)st[3];
  if (center) {
    const float mx = (float)(st[0] / cnt), my = (float)(st[1] / cnt), mz = (float)(st[2] / cnt);
    px -= mx; py -= my; pz -= mz;
    zmin -= mz;  // min(z - mz) == min(z) - mz in fp32 (rounding is monotone)
  }
  if (nullify_z) pz -= zmin;
  p[0] = px * pos_scale; p[1] = py * pos_scale; p[2] = pz * pos_scale;
  if (x) {
    float* r = x + i * ldx;
#pragma unroll
    for (int which = 0; which < 2; ++which) {
      const int c = which == 0 ? ci : cr;
      if (c < 0) continue;
      const double sum = st[4 + 2 * which], sq = st[5 + 2 * which];
      const double mean = sum / cnt;
      // torch.std (unbiased): sqrt(sum (v - mean)^2 / (n - 1)); NaN for n == 1 -> the reference then uses std = 1
      double var = cnt > 1. ? (sq - sum * mean) / (cnt - 1.) : __builtin_nan("");
      if (var < 0.) var = 0.;
      float sd = (float)sqrt(var) + 1e-6f;
      if (sd != sd) sd = 1.f;
      const float v = which == 0 ? logf(r[c] + 1.f) : r[c];
      const float z = (v - (float)mean) / sd;
      const float lim = clamp_sigma * sd;
      r[c] = fminf(fmaxf(z, -lim), lim);
    }
  }
}

extern "C" int m3d_tile_normalize(float* pos, int32_t pos_stride, float* x, int64_t ldx, int32_t intensity_col,
                                  int32_t rgb_col, const int64_t* ptr, int32_t num_clouds, int64_t n, int32_t center,
                                  int32_t nullify_z, float pos_scale, float clamp_sigma, double* stats_ws,
                                  void* stream) {
  if (n < 0 || num_clouds < 0) return M3D_ERR_INVALID;
  if (n == 0 || num_clouds == 0) return M3D_OK;
  if (!pos || pos_stride < 3 || !ptr || !stats_ws) return M3D_ERR_INVALID;
  if (!x && (intensity_col >= 0 || rgb_col >= 0)) return M3D_ERR_INVALID;
  if (x && (intensity_col >= ldx || rgb_col >= ldx)) return M3D_ERR_INVALID;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(tile_stats_kernel, dim3(num_clouds), dim3(1024), 0, st, pos, pos_stride, x, ldx, intensity_col,
                     rgb_col, ptr, stats_ws);
  hipLaunchKernelGGL(tile_apply_kernel, dim3((unsigned)m3d_cdiv(n, 256)), dim3(256), 0, st, pos, pos_stride, x, ldx,
                     intensity_col, rgb_col, ptr, num_clouds, n, center, nullify_z, pos_scale, clamp_sigma, stats_ws);
  M3D_CHECK_LAUNCH();
  return M3D_OK;
}

// ------------------------------------------------------------------------------------------
// Tiling of a whole cloud into square receptive fields
//
// Replaces the selection loop of split_cloud_into_samples()
// (/root/reference/myria3d/pctl/dataset/utils.py:126-158): a cKDTree over the xy coordinates shifted by their minimum
// and, per centre of the mosaic (get_mosaic_of_centers, utils.py:29-39), query_ball_point(centre, r=subtile_width // 2,
// p=inf) = every point whose Chebyshev distance to the centre is <= r (closed ball) — one CPU tree query per 50 m
// sample.  Here all samples of a cloud are produced in three launches: each point finds the few mosaic cells it can
// belong to arithmetically (the centres form a regular lattice) and tests them with the reference's own arithmetic
// (float32 shift by the minimum, then float64 |d| <= r); a wavefront walks its chunk of points IN ORDER and ranks the
// members of each sample with ballots, so every sample's index list comes out ascending and deterministic (the tree
// returns them in traversal order: the SET is the contract).  Layout: CSR (sample_ptr, idx).
// ------------------------------------------------------------------------------------------
#define TS_CHUNK 2048     // points per wavefront (one workgroup)
#define TS_MAX_SAMPLES 8192

struct TileSelArgs {
  const float* pos; int pstride; int64_t n;
  const double* centers;  // [nc] mosaic coordinates along one axis (the same values are used for x and y)
  int nc; double r; double start, step;
  float xmin, ymin;       // unused when minxy != nullptr
  const float* minxy;     // device [2]
  int32_t* hist;          // [nc*nc][nwg]: pass 0 writes counts, the host scans it in place, pass 1 reads offsets
  int nwg;
  int32_t* out_idx;
  int* err;                // device int: set when a point belongs to more than TS_MAX_MEMB samples
};

// partial xy minima: one float2 per workgroup, combined by tile_sel_min_final
__global__ __launch_bounds__(256) void tile_sel_min_kernel(const float* __restrict__ pos, int pstride, int64_t n,
                                                           float* __restrict__ part) {
  __shared__ float red[2][4];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  float mx = 3.4e38f, my = 3.4e38f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + tid; i < n; i += (int64_t)gridDim.x * 256) {
    mx = fminf(mx, pos[i * pstride]);
    my = fminf(my, pos[i * pstride + 1]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { mx = fminf(mx, __shfl_xor(mx, o, 64)); my = fminf(my, __shfl_xor(my, o, 64)); }
  if (lane == 0) { red[0][wid] = mx; red[1][wid] = my; }
  __syncthreads();
  if (tid == 0) {
    part[2 * blockIdx.x] = fminf(fminf(red[0][0], red[0][1]), fminf(red[0][2], red[0][3]));
    part[2 * blockIdx.x + 1] = fminf(fminf(red[1][0], red[1][1]), fminf(red[1][2], red[1][3]));
  }
}
__global__ __launch_bounds__(64) void tile_sel_min_final(const float* __restrict__ part, int nparts, float* __restrict__ out) {
  float mx = 3.4e38f, my = 3.4e38f;
  for (int i = threadIdx.x; i < nparts; i += 64) { mx = fminf(mx, part[2 * i]); my = fminf(my, part[2 * i + 1]); }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { mx = fminf(mx, __shfl_xor(mx, o, 64)); my = fminf(my, __shfl_xor(my, o, 64)); }
  if (threadIdx.x == 0) { out[0] = mx; out[1] = my; }
}

// candidate lattice range of one coordinate: every centre index i with |v - c[i]| <= r lies in [lo, hi]
__device__ __forceinline__ void ts_range(double v, const TileSelArgs& a, int& lo, int& hi) {
  lo = (int)floor((v - a.r - a.start) / a.step) - 1;
  hi = (int)ceil((v + a.r - a.start) / a.step) + 1;
  lo = lo < 0 ? 0 : lo;
  hi = hi > a.nc - 1 ? a.nc - 1 : hi;
}

#define TS_MAX_MEMB 64  // samples one point may belong to (overlapping mosaics: (floor(2r / step) + 1)^2)

template <bool WRITE>
__global__ __launch_bounds__(64) void tile_select_kernel(TileSelArgs a) {
  __shared__ int memb[TS_MAX_MEMB][64];  // per lane: the samples of its point, ascending
  // dynamic LDS sized by the number of samples (400 for a 1 km tile of 50 m samples: 5 KB, seven workgroups per CU beside memb;
  // sized for TS_MAX_SAMPLES it would be 96 KB and one wavefront per CU): mask[S] (write pass: lanes of the current 64 points
  // per sample), then cnt[S]
  extern __shared__ unsigned long long ts_dyn[];
  const int lane = threadIdx.x, wg = blockIdx.x;
  const int S = a.nc * a.nc;
  unsigned long long* mask = ts_dyn;
  int* cnt = (int*)(ts_dyn + (WRITE ? S : 0));
  for (int s = lane; s < S; s += 64) {
    cnt[s] = WRITE ? a.hist[(size_t)s * a.nwg + wg] : 0;
    if (WRITE) mask[s] = 0ull;
  }
  __syncthreads();
  const float xmin = a.minxy[0], ymin = a.minxy[1];
  const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  const int64_t base = (int64_t)wg * TS_CHUNK;
  bool overflow = false;
  for (int i0 = 0; i0 < TS_CHUNK; i0 += 64) {
    const int64_t p = base + i0 + lane;
    int nm = 0;
    if (p < a.n) {
      // the reference's arithmetic: float32 subtraction of the minimum, then float64 distances (cKDTree data is f64)
      const double xs = (double)(a.pos[p * a.pstride] - xmin);
      const double ys = (double)(a.pos[p * a.pstride + 1] - ymin);
      int ilo, ihi, jlo, jhi;
      ts_range(xs, a, ilo, ihi);
      ts_range(ys, a, jlo, jhi);
      for (int i = ilo; i <= ihi; ++i) {  // x-major = ascending sample number
        if (!(fabs(xs - a.centers[i]) <= a.r)) continue;
        for (int j = jlo; j <= jhi; ++j) {
          if (!(fabs(ys - a.centers[j]) <= a.r)) continue;
          if (nm < TS_MAX_MEMB) memb[nm++][lane] = i * a.nc + j;
          else overflow = true;
        }
      }
    }
    // Round 5: rank by per-sample lane masks instead of walking the distinct samples of the 64 points one after the other
    // (6 cross-lane steps, a ballot and 2 barriers per DISTINCT sample: a cloud whose points arrive in no spatial order — the
    // bench's synthetic one — has ~60 of them per 64 points, 13.8 ms for 10 M points; a scan-ordered LAS has a few).
    //   count pass: an LDS counter per membership, nothing else;
    //   write pass: every lane ORs its bit into mask[s] for EACH of its samples, then its slot in sample s is
    //   cnt[s] + (set bits below its own): ascending by point index inside every sample, whatever route a point took to it (a
    //   lane is in a sample at most once); the lowest lane of a mask advances the counter and clears the mask.
    if (!WRITE) {
      for (int k = 0; k < nm; ++k) atomicAdd(&cnt[memb[k][lane]], 1);
    } else {
      for (int k = 0; k < nm; ++k) atomicOr(&mask[memb[k][lane]], 1ull << lane);
      __syncthreads();
      for (int k = 0; k < nm; ++k) {
        const int s0 = memb[k][lane];
        a.out_idx[cnt[s0] + __popcll(mask[s0] & lt)] = (int32_t)p;
      }
      __syncthreads();
      for (int k = 0; k < nm; ++k) {
        const int s0 = memb[k][lane];
        const unsigned long long m = mask[s0];
        if ((m & lt) == 0ull) { cnt[s0] += __popcll(m); mask[s0] = 0ull; }  // (one leader per sample: no race)
      }
      __syncthreads();
    }
  }
  if (!WRITE) {
    __syncthreads();
    for (int s = lane; s < S; s += 64) a.hist[(size_t)s * a.nwg + wg] = cnt[s];
    if (__ballot(overflow) != 0 && lane == 0) atomicExch(a.err, 1);
  }
}

__global__ __launch_bounds__(256) void tile_sel_ptr_kernel(const int32_t* __restrict__ scanned, int S, int nwg,
                                                           int64_t* __restrict__ sample_ptr, const int* __restrict__ err) {
  const int s = blockIdx.x * 256 + threadIdx.x;
  if (s > S) return;
  int64_t v = (int64_t)scanned[(size_t)s * nwg];  // s == S: the grand total stored behind the table
  if (s == S && *err) v = -1;  // a point belongs to more than TS_MAX_MEMB samples: the lists are incomplete
  sample_ptr[s] = v;
}

extern "C" size_t m3d_tile_select_workspace_bytes(int64_t n, int32_t centers_per_axis) {
  if (n < 0 || centers_per_axis < 1) return 0;
  const int64_t nwg = m3d_cdiv(n > 0 ? n : 1, TS_CHUNK);
  const int64_t S = (int64_t)centers_per_axis * centers_per_axis;
  return al256((size_t)(S * nwg + 1) * 4) + al256(2 * 1024 * 4) + 512 + al256((size_t)((S * nwg) / 4096 + 2) * 4);
}

// pass 0 (count_only != 0): fills the histogram, scans it, writes sample_ptr[S + 1] (int64) — the caller reads
// sample_ptr[S] to size idx_out; pass 1: writes idx_out (int32 point indices, ascending inside each sample).
extern "C" int m3d_tile_select(const float* pos, int32_t pos_stride, int64_t n, const double* centers_dev,
                               int32_t centers_per_axis, double radius, double start, double step, void* ws,
                               int32_t count_only, int64_t* sample_ptr, int32_t* idx_out, void* stream) {
  if (n < 0 || pos_stride < 2 || centers_per_axis < 1 || !ws || !sample_ptr || !centers_dev) return M3D_ERR_INVALID;
  if (!(step > 0.0) || !(radius >= 0.0)) return M3D_ERR_INVALID;
  const int64_t S = (int64_t)centers_per_axis * centers_per_axis;
  if (S > TS_MAX_SAMPLES) return M3D_ERR_UNSUPPORTED;
  if (n > 0 && !pos) return M3D_ERR_INVALID;
  if (!count_only && !idx_out && n > 0) return M3D_ERR_INVALID;
  if (n >= (1ll << 31)) return M3D_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const int64_t nwg = m3d_cdiv(n > 0 ? n : 1, TS_CHUNK);
  if (S * nwg + 1 >= (1ll << 31)) return M3D_ERR_UNSUPPORTED;
  {
    // the write pass keeps a 12-byte record per sample in DYNAMIC LDS on top of the kernels' static 16 KB: beyond
    // the default 64 KB per workgroup the launch has to ask for it (gfx950: 160 KB per CU), and what the device cannot give
    // is reported as "unsupported", not as a failed launch
    const size_t dyn = (size_t)S * 12 + 16;
    int dev = 0, lim = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&lim, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess)
      return M3D_ERR_LAUNCH;
    if (dyn + 16384 > (size_t)lim) return M3D_ERR_UNSUPPORTED;
    if (dyn + 16384 > 65536 &&
        (hipFuncSetAttribute((const void*)tile_select_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn) != hipSuccess ||
         hipFuncSetAttribute((const void*)tile_select_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn) != hipSuccess))
      return M3D_ERR_UNSUPPORTED;
  }
  char* p = (char*)ws;
  int32_t* hist = (int32_t*)p; p += al256((size_t)(S * nwg + 1) * 4);
  float* part = (float*)p;  // [1024][2] partial minima, then [2] final at part + 2048
  TileSelArgs a;
  a.pos = pos; a.pstride = pos_stride; a.n = n; a.centers = centers_dev; a.nc = centers_per_axis; a.r = radius;
  a.start = start; a.step = step; a.xmin = a.ymin = 0.f; a.minxy = part + 2048 - 2; a.hist = hist; a.nwg = (int)nwg;
  a.out_idx = idx_out;
  a.err = (int*)(part + 2048);
  if (count_only) {
    if (hipMemsetAsync(a.err, 0, sizeof(int), st) != hipSuccess) return M3D_ERR_LAUNCH;
    int nparts = (int)(n / 4096 + 1);
    if (nparts > 1023) nparts = 1023;
    hipLaunchKernelGGL(tile_sel_min_kernel, dim3(nparts), dim3(256), 0, st, pos, pos_stride, n, part);
    hipLaunchKernelGGL(tile_sel_min_final, dim3(1), dim3(64), 0, st, (const float*)part, nparts, part + 2048 - 2);
    hipLaunchKernelGGL((tile_select_kernel<false>), dim3((unsigned)nwg), dim3(64), (size_t)S * 4 + 16, st, a);
    // exclusive scan in sample-major order: hist[s][wg] -> first output slot of (sample s, chunk wg); total at the end
    exscan_launch(hist, S * nwg, hist + S * nwg, (int32_t*)(p + al256(2 * 1024 * 4) + 512), st);
    hipLaunchKernelGGL(tile_sel_ptr_kernel, dim3((unsigned)m3d_cdiv(S + 1, 256)), dim3(256), 0, st, (const int32_t*)hist,
                       (int)S, (int)nwg, sample_ptr, (const int*)a.err);
  } else {
    hipLaunchKernelGGL((tile_select_kernel<true>), dim3((unsigned)nwg), dim3(64), (size_t)S * 12 + 16, st, a);
  }
  M3D_CHECK_LAUNCH();
  return M3D_OK;
}
