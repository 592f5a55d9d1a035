// Batched exact k-nearest-neighbour search for gfx950.
//
// Replaces torch_cluster.knn as reached from knn_graph(pos, K, batch, loop=True)
// (/root/reference/myria3d/models/modules/pyg_randla_net.py:180) and from knn_interpolate
// (pyg_randla_net.py:250, myria3d/models/model.py:90).
//
// Design (MI355X-first, not the upstream 1-thread-per-query brute force): every cloud gets a uniform
// xy grid of vertical columns (cell size chosen for ~M3D_KNN_CELL_TARGET points per column) built by ONE
// workgroup with an LDS histogram + scan + scatter; sources are stored cell-sorted as float4
// (x, y, z, original index) so that a wavefront of consecutive (cell-sorted) queries walks the same few
// cache lines.  Each lane keeps its top-k as sorted 64-bit keys (fp32 bits of d2 << 32 | index): one
// u64 compare gives the total order (d2, index), so results are deterministic and bit-comparable with the
// CPU oracle.  Rings of cells are visited until the k-th distance is inside the explored block (exact).
// Distances are computed as (dx*dx + dy*dy) + dz*dz with contraction disabled.
#include "m3d_common.h"
#include "../../include/m3d_hip.h"
#include <stdlib.h>

#define GMAX 64
#define CELLS_MAX (GMAX * GMAX)
#define GP_STRIDE 8  // per-cloud grid record, 8 x 4 bytes
#define M3D_KNN_CELL_TARGET 7.0f
#ifndef KNN_UNROLL
#define KNN_UNROLL 4
#endif
#ifndef KNN_PP_OFF_CLAMPED
#define KNN_PP_OFF_CLAMPED 0
#endif
#ifndef KNN_PP
#define KNN_PP 1  // direct-insertion kernel, lists of 8+ keys: ping-pong register sets in scan_range
#endif
#ifndef KNN_PIPE
#define KNN_PIPE 0  // direct-insertion kernel: the same pipelined walk (levels 2-4: 120 / 77 / 49 vs 108 / 68 / 45 us: slower; off)
#endif
#ifndef KNN_MIN_BLOCKS
#define KNN_MIN_BLOCKS 1
#endif
#ifndef M3D_KNN_DEFAULT_F64
#define M3D_KNN_DEFAULT_F64 1
#endif
// deferred-insertion query kernel (k > 4): depth of the per-lane LDS candidate queue, register cap (waves per SIMD)
#ifndef KNNQ_DEPTH
#define KNNQ_DEPTH 16
#endif
#ifndef KNNQ_MINW
#define KNNQ_MINW 4
#endif
#ifndef KNNQ_UNROLL
#define KNNQ_UNROLL 4
#endif
#ifndef KNNQ_DRAIN
#define KNNQ_DRAIN 2
#endif
#ifndef KNNQ_PP
#define KNNQ_PP 1  // deferred-insertion kernel: ping-pong register sets in the candidate loop (0: round 2's loop): 192 -> 183 us
#endif
#ifndef KNNQ_WAVES
#define KNNQ_WAVES 1  // independent wavefronts per workgroup of the deferred-insertion kernel (A/B knob)
#endif
#ifndef KNNQ_PIPE
// 1: software-pipelined walk over a ring's runs (bounds of run r+1 and its first records in flight while run r is
// scanned).  Bit-identical and measured SLOWER on the same box (level 1: 202-205 vs 188-191 us; profiles/r03_knn_pipe_ab.log):
// the exposed round trips between runs are not what bounds the kernel.  Kept as an A/B knob.
#define KNNQ_PIPE 0
#endif
static_assert((KNNQ_DEPTH & (KNNQ_DEPTH - 1)) == 0, "queue depth must be a power of two");

struct KnnWs {
  float* gridp;     // [B][8]: xmin, ymin, inv_h, h, eps, (int)Gx, (int)Gy, (int)n
  int* cell_start;  // [B][CELLS_MAX + 1]
  float4* sorted;   // [n_src]  (x, y, z, bits of the original row) in cell-sorted order
  int* perm;        // [n_src]  perm[slot] = original row
  int* inv;         // [n_src]  inv[original row] = slot
};

static inline size_t ws_gridp_bytes(int B) { return (size_t)m3d_align((int64_t)B * GP_STRIDE * 4, 256); }
static inline size_t ws_cells_bytes(int B) { return (size_t)m3d_align((int64_t)B * (CELLS_MAX + 1) * 4, 256); }

static inline size_t ws_sorted_bytes(int64_t n) { return (size_t)m3d_align(n * (int64_t)sizeof(float4), 256); }
static inline size_t ws_perm_bytes(int64_t n) { return (size_t)m3d_align(n * (int64_t)sizeof(int), 256); }

static inline KnnWs ws_carve(void* ws, int B, int64_t n) {
  KnnWs w;
  char* p = (char*)ws;
  w.gridp = (float*)p;
  p += ws_gridp_bytes(B);
  w.cell_start = (int*)p;
  p += ws_cells_bytes(B);
  w.sorted = (float4*)p;
  p += ws_sorted_bytes(n);
  w.perm = (int*)p;
  p += ws_perm_bytes(n);
  w.inv = (int*)p;
  return w;
}

extern "C" size_t m3d_knn_workspace_bytes(int64_t n_src, int32_t num_clouds) {
  if (n_src < 0 || num_clouds < 0) return 0;
  return ws_gridp_bytes(num_clouds) + ws_cells_bytes(num_clouds) + ws_sorted_bytes(n_src) + 2 * ws_perm_bytes(n_src) + 256;
}

// byte offsets of the cell-sorted arrays inside a built workspace: which = 0 sorted float4 [n] (x, y, z, row bits),
// 1 perm int32 [n] (slot -> original row), 2 inv int32 [n] (original row -> slot)
extern "C" size_t m3d_knn_workspace_offset(int64_t n_src, int32_t num_clouds, int32_t which) {
  size_t o = ws_gridp_bytes(num_clouds) + ws_cells_bytes(num_clouds);
  if (which >= 1) o += ws_sorted_bytes(n_src);
  if (which >= 2) o += ws_perm_bytes(n_src);
  return o;
}

// ------------------------------------------------------------------------------------------
// grid build: one 1024-thread workgroup per cloud
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void knn_build_kernel(const float* __restrict__ pos, int pstride,
                                                         const int64_t* __restrict__ ptr, KnnWs w,
                                                         float cell_target, const int32_t* __restrict__ map_in,
                                                         int32_t* __restrict__ map_out) {
  __shared__ float red[4][16];
  __shared__ int cnt[CELLS_MAX];
  __shared__ int wsum[16];
  __shared__ float gp[8];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int64_t s0 = ptr[b];
  const int n = (int)(ptr[b + 1] - s0);
  float* gpo = w.gridp + (size_t)b * GP_STRIDE;
  int* cso = w.cell_start + (size_t)b * (CELLS_MAX + 1);
  if (n <= 0) {
    if (tid == 0) {
      gpo[0] = 0.f; gpo[1] = 0.f; gpo[2] = 1.f; gpo[3] = 1.f; gpo[4] = 0.f;
      ((int*)gpo)[5] = 1; ((int*)gpo)[6] = 1; ((int*)gpo)[7] = 0;
      cso[0] = 0; cso[1] = 0;
    }
    return;
  }
  // ---- bounding box in xy
  float xmin = 3.4e38f, xmax = -3.4e38f, ymin = 3.4e38f, ymax = -3.4e38f;
  for (int i = tid; i < n; i += 1024) {
    const float* p = pos + (s0 + i) * pstride;
    float x = p[0], y = p[1];
    xmin = fminf(xmin, x); xmax = fmaxf(xmax, x);
    ymin = fminf(ymin, y); ymax = fmaxf(ymax, y);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    xmin = fminf(xmin, __shfl_xor(xmin, o, 64)); xmax = fmaxf(xmax, __shfl_xor(xmax, o, 64));
    ymin = fminf(ymin, __shfl_xor(ymin, o, 64)); ymax = fmaxf(ymax, __shfl_xor(ymax, o, 64));
  }
  if (lane == 0) { red[0][wid] = xmin; red[1][wid] = xmax; red[2][wid] = ymin; red[3][wid] = ymax; }
  for (int c = tid; c < CELLS_MAX; c += 1024) cnt[c] = 0;
  __syncthreads();
  if (tid == 0) {
    for (int i = 1; i < 16; ++i) {
      xmin = fminf(xmin, red[0][i]); xmax = fmaxf(xmax, red[1][i]);
      ymin = fminf(ymin, red[2][i]); ymax = fmaxf(ymax, red[3][i]);
    }
    float wx = xmax - xmin, wy = ymax - ymin;
    float wmax = fmaxf(wx, wy);
    float h;
    if (!(wmax > 0.f)) {
      h = 1.f;
    } else {
      float area = fmaxf(wx, wmax * 1e-3f) * fmaxf(wy, wmax * 1e-3f);
      h = sqrtf(area * cell_target / (float)n);
      h = fmaxf(h, wmax / (float)GMAX * 1.0001f);
    }
    int Gx = min(GMAX, (int)(wx / h) + 1), Gy = min(GMAX, (int)(wy / h) + 1);
    float amax = fmaxf(fmaxf(fabsf(xmin), fabsf(xmax)), fmaxf(fabsf(ymin), fabsf(ymax)));
    gp[0] = xmin; gp[1] = ymin; gp[2] = 1.f / h; gp[3] = h;
    gp[4] = 2e-4f * h + 16.f * 1.1920929e-7f * amax;  // slack for cell-assignment rounding
    ((int*)gp)[5] = Gx; ((int*)gp)[6] = Gy; ((int*)gp)[7] = n;
    for (int i = 0; i < 8; ++i) gpo[i] = gp[i];
  }
  __syncthreads();
  const float gx0 = gp[0], gy0 = gp[1], inv_h = gp[2];
  const int Gx = ((int*)gp)[5], Gy = ((int*)gp)[6];
  const int ncell = Gx * Gy;
  // ---- histogram
  for (int i = tid; i < n; i += 1024) {
    const float* p = pos + (s0 + i) * pstride;
    int cx = min(Gx - 1, max(0, (int)((p[0] - gx0) * inv_h)));
    int cy = min(Gy - 1, max(0, (int)((p[1] - gy0) * inv_h)));
    atomicAdd(&cnt[cy * Gx + cx], 1);
  }
  __syncthreads();
  // ---- exclusive scan over <= 4096 cells: 4 cells per thread
  int c0 = tid * 4;
  int v0 = c0 + 0 < ncell ? cnt[c0 + 0] : 0, v1 = c0 + 1 < ncell ? cnt[c0 + 1] : 0;
  int v2 = c0 + 2 < ncell ? cnt[c0 + 2] : 0, v3 = c0 + 3 < ncell ? cnt[c0 + 3] : 0;
  int tsum = v0 + v1 + v2 + v3;
  int incl = tsum;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int t = __shfl_up(incl, o, 64);
    if (lane >= o) incl += t;
  }
  if (lane == 63) wsum[wid] = incl;
  __syncthreads();
  int woff = 0;
  for (int i = 0; i < wid; ++i) woff += wsum[i];
  int ex = woff + incl - tsum;
  __syncthreads();
  if (c0 + 0 < ncell) { cnt[c0 + 0] = ex; cso[c0 + 0] = ex; }
  if (c0 + 1 < ncell) { cnt[c0 + 1] = ex + v0; cso[c0 + 1] = ex + v0; }
  if (c0 + 2 < ncell) { cnt[c0 + 2] = ex + v0 + v1; cso[c0 + 2] = ex + v0 + v1; }
  if (c0 + 3 < ncell) { cnt[c0 + 3] = ex + v0 + v1 + v2; cso[c0 + 3] = ex + v0 + v1 + v2; }
  if (tid == 0) cso[ncell] = n;
  __syncthreads();
  // ---- scatter into cell-sorted order (order inside a cell is arbitrary; results do not depend on it)
  for (int i = tid; i < n; i += 1024) {
    const float* p = pos + (s0 + i) * pstride;
    float x = p[0], y = p[1], z = p[2];
    int cx = min(Gx - 1, max(0, (int)((x - gx0) * inv_h)));
    int cy = min(Gy - 1, max(0, (int)((y - gy0) * inv_h)));
    int slot = atomicAdd(&cnt[cy * Gx + cx], 1);
    w.sorted[s0 + slot] = make_float4(x, y, z, __int_as_float((int)(s0 + i)));
    w.perm[s0 + slot] = (int)(s0 + i);
    w.inv[s0 + i] = (int)(s0 + slot);
    if (map_out) map_out[s0 + slot] = map_in[s0 + i];  // a per-row payload carried into cell-sorted order (m3d_knn_build_map)
  }
}

// ------------------------------------------------------------------------------------------
// query
// ------------------------------------------------------------------------------------------
typedef unsigned long long u64;

// The running top-k of a lane is a sorted register array of 64-bit keys (d2 bits, original row): two key policies
// with the SAME total order, selected at launch (M3D_KNN_KEYS=u64|f64).
//
// KeyU64: the key is an integer; insertion = compare + select per slot (~6 32-bit VALU instructions per slot).
struct KeyU64 {
  typedef u64 T;
  static constexpr bool IS_F64 = false;
  static __device__ __forceinline__ T make(float d2, int row) {
    return ((u64)__float_as_uint(d2) << 32) | (u64)(unsigned)row;
  }
  static __device__ __forceinline__ T empty() { return ~0ull; }
  static __device__ __forceinline__ bool is_empty(T k) { return k == ~0ull; }
  static __device__ __forceinline__ unsigned d2bits(T k) { return (unsigned)(k >> 32); }
  static __device__ __forceinline__ unsigned hi32(T k) { return (unsigned)(k >> 32); }
  static __device__ __forceinline__ unsigned d2bits_of_hi(unsigned hw) { return hw; }
  static __device__ __forceinline__ int row(T k) { return (int)(unsigned)(k & 0xffffffffull); }
  template <int KMAX>
  static __device__ __forceinline__ void insert(T (&best)[KMAX], T key) {
    if (key < best[KMAX - 1]) {
#pragma unroll
      for (int j = KMAX - 1; j > 0; --j) {
        u64 prev = best[j - 1];
        best[j] = key < prev ? prev : (key < best[j] ? key : best[j]);
      }
      best[0] = key < best[0] ? key : best[0];
    }
  }
};

// KeyF64: the same 64 bits read as an IEEE double.  For sign bit 0 the order of doubles IS the order of their bit
// patterns, so a sorted insertion is a chain of v_min_f64 / v_max_f64 — 2 full-rate VALU instructions per slot.
// The high word is biased by one double-exponent step (0x00100000): every finite / inf / canonical-NaN fp32 d2 (bit
// patterns up to 0x7FDFFFFF; tests/test_host.py pins the mapping) then maps to a
// NORMAL finite double (no denormal or NaN operand ever reaches min/max, so the bits pass through unchanged), and
// +inf (0x7FF00000'00000000) is the "empty slot" sentinel, above every key.
struct KeyF64 {
  typedef double T;
  static constexpr bool IS_F64 = true;
  static constexpr unsigned BIAS = 0x00100000u;
  static __device__ __forceinline__ T make(float d2, int row) {
    return __hiloint2double((int)(__float_as_uint(d2) + BIAS), row);
  }
  static __device__ __forceinline__ T empty() { return __hiloint2double(0x7FF00000, 0); }
  static __device__ __forceinline__ bool is_empty(T k) { return (unsigned)__double2hiint(k) == 0x7FF00000u; }
  static __device__ __forceinline__ unsigned d2bits(T k) { return (unsigned)__double2hiint(k) - BIAS; }
  static __device__ __forceinline__ unsigned hi32(T k) { return (unsigned)__double2hiint(k); }
  static __device__ __forceinline__ unsigned d2bits_of_hi(unsigned hw) { return hw - BIAS; }
  static __device__ __forceinline__ int row(T k) { return __double2loint(k); }
  template <int KMAX>
  static __device__ __forceinline__ void insert(T (&best)[KMAX], T key) {
    if (key < best[KMAX - 1]) {
#pragma unroll
      for (int j = 0; j < KMAX; ++j) {
        // raw instructions: fmin()/fmax() would add a canonicalising v_max_f64 per operand in IEEE mode
        T hi;
        asm("v_max_f64 %0, %1, %2" : "=&v"(hi) : "v"(best[j]), "v"(key));
        asm("v_min_f64 %0, %0, %1" : "+v"(best[j]) : "v"(key));  // in place: no register copies at the join
        key = hi;
      }
    }
  }
};

__device__ __forceinline__ float dist2_exact(float qx, float qy, float qz, float4 s) {
#pragma clang fp contract(off)
  float dx = s.x - qx, dy = s.y - qy, dz = s.z - qz;
  float a = dx * dx;
  float b = dy * dy;
  float c = dz * dz;
  float ab = a + b;
  return ab + c;
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 dist2_exact_pk(f32x2 qx, f32x2 qy, f32x2 qz, f32x2 x, f32x2 y, f32x2 z) {
#pragma clang fp contract(off)
  f32x2 dx = x - qx, dy = y - qy, dz = z - qz;
  f32x2 a = dx * dx;
  f32x2 b = dy * dy;
  f32x2 c = dz * dz;
  f32x2 ab = a + b;
  return ab + c;
}

template <int KMAX, class KP>
__device__ __forceinline__ void scan_range(typename KP::T (&best)[KMAX], const float4* __restrict__ sorted, int p0,
                                           int p1, float qx, float qy, float qz) {
  if (p1 <= p0) return;
  const int last = p1 - 1;
  auto examine = [&](const float4 (&s)[KNN_UNROLL], int p) {
#pragma unroll
    for (int u = 0; u < KNN_UNROLL; ++u) {
      float d2 = dist2_exact(qx, qy, qz, s[u]);
      typename KP::T key = KP::make(d2, __float_as_int(s[u].w));
      if (p + u > last) key = KP::empty();
      KP::template insert<KMAX>(best, key);
    }
  };
#if KNN_PP_OFF_CLAMPED
  // round 2's loop: clamped addresses, load then use
  for (int p = p0; p < p1; p += KNN_UNROLL) {
    float4 s[KNN_UNROLL];
#pragma unroll
    for (int u = 0; u < KNN_UNROLL; ++u) s[u] = sorted[min(p + u, last)];
    examine(s, p);
  }
#else
  // Unclamped addresses: one base per trip + immediate offsets (reads run up to 3*KNN_UNROLL-1 records past p1: inside the
  // 256-byte-padded workspace, masked in examine()) — per-index clamps cost a v_min and an address computation per load in
  // kernels whose bound is instruction issue.  Lists of 8+ keys (the insertion chain is long enough to cover a load):
  // two register sets used in turn, the next trip's loads in flight while this trip's candidates are inserted
  // (levels 2-4: 110 / 70 / 46 -> 99 / 65 / 44 us); the 1-NN and 4-NN queries just load and use (49 -> 46 us)
  if constexpr (KNN_PP && KMAX >= 8) {
    float4 ra[KNN_UNROLL], rb[KNN_UNROLL];
#pragma unroll
    for (int u = 0; u < KNN_UNROLL; ++u) ra[u] = sorted[p0 + u];
    for (int p = p0; p < p1; p += 2 * KNN_UNROLL) {
#pragma unroll
      for (int u = 0; u < KNN_UNROLL; ++u) rb[u] = sorted[p + KNN_UNROLL + u];
      examine(ra, p);
#pragma unroll
      for (int u = 0; u < KNN_UNROLL; ++u) ra[u] = sorted[p + 2 * KNN_UNROLL + u];
      examine(rb, p + KNN_UNROLL);
    }
  } else {
    for (int p = p0; p < p1; p += KNN_UNROLL) {
      float4 s[KNN_UNROLL];
#pragma unroll
      for (int u = 0; u < KNN_UNROLL; ++u) s[u] = sorted[p + u];
      examine(s, p);
    }
  }
#endif
}

// qmode 0: queries are pos_qry rows (row index = output row)
// qmode 1: queries are the float4 records of qsorted (output row = record.w) — cell-sorted, wave-coherent
template <int KMAX, class KP>
__device__ __forceinline__ void knn_query_direct_body(const KnnWs& w, const int64_t* __restrict__ ptr_src, int B,
                                                      const float* __restrict__ pos_qry, int qstride,
                                                      const float4* __restrict__ qsorted,
                                                      const int64_t* __restrict__ ptr_qry, int64_t n_qry, int k,
                                                      int* __restrict__ idx_out, float* __restrict__ d2_out,
                                                      int sorted_io, int64_t t) {
  if (t >= n_qry) return;
  // cloud of this query: largest b with ptr_qry[b] <= t
  int lo = 0, hi = B;
  while (hi - lo > 1) {
    int mid = (lo + hi) >> 1;
    if (ptr_qry[mid] <= t) lo = mid; else hi = mid;
  }
  const int b = lo;
  float qx, qy, qz;
  int64_t orow;
  if (qsorted) {
    float4 q = qsorted[t];
    qx = q.x; qy = q.y; qz = q.z; orow = sorted_io ? t : (int64_t)__float_as_int(q.w);
  } else {
    const float* p = pos_qry + t * qstride;
    qx = p[0]; qy = p[1]; qz = p[2]; orow = t;
  }
  const float* gp = w.gridp + (size_t)b * GP_STRIDE;
  const float gx0 = gp[0], gy0 = gp[1], inv_h = gp[2], h = gp[3], eps = gp[4];
  const int Gx = ((const int*)gp)[5], Gy = ((const int*)gp)[6], n = ((const int*)gp)[7];
  const int* cs = w.cell_start + (size_t)b * (CELLS_MAX + 1);
  const float4* sorted = w.sorted + ptr_src[b];

  typename KP::T best[KMAX];
#pragma unroll
  for (int j = 0; j < KMAX; ++j) best[j] = KP::empty();

  if (n > 0) {
    const int cx = min(Gx - 1, max(0, (int)((qx - gx0) * inv_h)));
    const int cy = min(Gy - 1, max(0, (int)((qy - gy0) * inv_h)));
    for (int R = 0;; ++R) {
#if KNN_PIPE
      // software-pipelined walk over the ring's 4R runs (see knn_query_queue_body): the bounds of run r+1 and its first
      // records are in flight while run r is scanned — the deep-level and 1-NN launches are a few wavefronts per CU whose
      // whole life is a chain of dependent round trips (cell bounds -> records -> next cell bounds ...)
      const int nr = R == 0 ? 1 : 4 * R;
      auto run_bounds = [&](int r, int& a, int& b2) {
        int yy, xa, xb;
        if (r < 2) {
          yy = r == 0 ? cy - R : cy + R;
          xa = max(cx - R, 0); xb = min(cx + R, Gx - 1);
        } else {
          const int m = r - 2;
          yy = cy - R + 1 + (m >> 1);
          xa = xb = (m & 1) ? cx + R : cx - R;
        }
        const bool ok = yy >= 0 && yy < Gy && xa >= 0 && xb < Gx;
        const int ia = ok ? yy * Gx + xa : 0, ib = ok ? yy * Gx + xb + 1 : 0;
        a = cs[ia]; b2 = cs[ib];
      };
      const int nlast = n - 1;
      int p0, p1;
      run_bounds(0, p0, p1);
      float4 nx[KNN_UNROLL];
#pragma unroll
      for (int u = 0; u < KNN_UNROLL; ++u) nx[u] = sorted[min(p0 + u, nlast)];
      for (int r = 0; r < nr; ++r) {
        int q0 = 0, q1 = 0;
        if (r + 1 < nr) run_bounds(r + 1, q0, q1);
        for (int p = p0; p < p1; p += KNN_UNROLL) {
          float4 sv[KNN_UNROLL];
#pragma unroll
          for (int u = 0; u < KNN_UNROLL; ++u) sv[u] = nx[u];
          const int pn = p + KNN_UNROLL < p1 ? p + KNN_UNROLL : q0;
#pragma unroll
          for (int u = 0; u < KNN_UNROLL; ++u) nx[u] = sorted[min(pn + u, nlast)];
#pragma unroll
          for (int u = 0; u < KNN_UNROLL; ++u) asm volatile("" : "+v"(nx[u].w));
#pragma unroll
          for (int u = 0; u < KNN_UNROLL; ++u) {
            float d2 = dist2_exact(qx, qy, qz, sv[u]);
            typename KP::T key = KP::make(d2, __float_as_int(sv[u].w));
            if (p + u >= p1) key = KP::empty();
            KP::template insert<KMAX>(best, key);
          }
        }
        if (p1 <= p0) {
#pragma unroll
          for (int u = 0; u < KNN_UNROLL; ++u) nx[u] = sorted[min(q0 + u, nlast)];
        }
        p0 = q0; p1 = q1;
      }
#else
      for (int dy = -R; dy <= R; ++dy) {
        int yy = cy + dy;
        if (yy < 0 || yy >= Gy) continue;
        if (dy == -R || dy == R) {
          int x0 = max(cx - R, 0), x1 = min(cx + R, Gx - 1);
          scan_range<KMAX, KP>(best, sorted, cs[yy * Gx + x0], cs[yy * Gx + x1 + 1], qx, qy, qz);
        } else {
          if (cx - R >= 0) scan_range<KMAX, KP>(best, sorted, cs[yy * Gx + cx - R], cs[yy * Gx + cx - R + 1], qx, qy, qz);
          if (cx + R < Gx) scan_range<KMAX, KP>(best, sorted, cs[yy * Gx + cx + R], cs[yy * Gx + cx + R + 1], qx, qy, qz);
        }
      }
#endif
      const bool covers = (cx - R <= 0) && (cx + R >= Gx - 1) && (cy - R <= 0) && (cy + R >= Gy - 1);
      if (covers) break;
      float bound = 3.4e38f;
      if (cx - R > 0) bound = fminf(bound, qx - (gx0 + (float)(cx - R) * h));
      if (cx + R < Gx - 1) bound = fminf(bound, (gx0 + (float)(cx + R + 1) * h) - qx);
      if (cy - R > 0) bound = fminf(bound, qy - (gy0 + (float)(cy - R) * h));
      if (cy + R < Gy - 1) bound = fminf(bound, (gy0 + (float)(cy + R + 1) * h) - qy);
      bound = fmaxf(bound - eps, 0.f);
      typename KP::T kb = best[KMAX - 1];
#pragma unroll
      for (int j = 0; j < KMAX - 1; ++j)
        if (j == k - 1) kb = best[j];
      // an unfilled slot reads as NaN (u64 keys) or as a huge value (f64 keys): either way the search continues
      float kth = KP::is_empty(kb) ? __builtin_nanf("") : __uint_as_float(KP::d2bits(kb));
      if (kth <= bound * bound) break;
    }
  }
  int* io = idx_out + orow * k;
#pragma unroll
  for (int j = 0; j < KMAX; ++j) {
    if (j < k) {
      bool ok = !KP::is_empty(best[j]);
      int id = ok ? KP::row(best[j]) : -1;
      if (sorted_io && ok) id = w.inv[id];  // neighbours selected by (d2, original row); reported as cell-sorted slots
      io[j] = id;
      if (d2_out) d2_out[orow * k + j] = ok ? __uint_as_float(KP::d2bits(best[j])) : __builtin_inff();
    }
  }
}

template <int KMAX, class KP>
__global__ __launch_bounds__(256, KNN_MIN_BLOCKS) void knn_query_kernel(KnnWs w, const int64_t* __restrict__ ptr_src, int B,
                                                        const float* __restrict__ pos_qry, int qstride,
                                                        const float4* __restrict__ qsorted,
                                                        const int64_t* __restrict__ ptr_qry, int64_t n_qry, int k,
                                                        int* __restrict__ idx_out, float* __restrict__ d2_out,
                                                        int sorted_io) {
  knn_query_direct_body<KMAX, KP>(w, ptr_src, B, pos_qry, qstride, qsorted, ptr_qry, n_qry, k, idx_out, d2_out, sorted_io,
                                  (int64_t)blockIdx.x * 256 + threadIdx.x);
}

// Several independent query problems in ONE launch (m3d_knn_query_batch): the K-NN tables of the four resolution levels
// (or the decoder's four 1-NN tables) are separate launches otherwise, and the deep ones are a few wavefronts per CU
// of pure latency (105 / 77 / 45 us for 51 200 / 12 800 / 3 200 queries) that run one after the other while the
// level-1 launch ends on its slowest wavefronts with most SIMDs idle.  Workgroup b belongs to job j with
// wg_start[j] <= b < wg_start[j + 1]; cell-sorted queries only (qry_ws), no distances.
#define KNN_BATCH_MAX 8
struct KnnBatch {
  KnnWs w[KNN_BATCH_MAX];
  const int64_t* ptr_src[KNN_BATCH_MAX];
  const float4* qsorted[KNN_BATCH_MAX];
  const int64_t* ptr_qry[KNN_BATCH_MAX];
  int64_t n_qry[KNN_BATCH_MAX];
  int* idx_out[KNN_BATCH_MAX];
  unsigned wg_start[KNN_BATCH_MAX + 1];
  int njobs;
};
__device__ __forceinline__ int knn_batch_job(const KnnBatch& a) {
  int j = 0;
#pragma unroll
  for (int i = 1; i < KNN_BATCH_MAX; ++i) j += (i < a.njobs && blockIdx.x >= a.wg_start[i]) ? 1 : 0;
  return j;
}
template <int KMAX, class KP>
__global__ __launch_bounds__(256, KNN_MIN_BLOCKS) void knn_query_batch_kernel(KnnBatch a, int B, int k, int sorted_io) {
  const int j = knn_batch_job(a);
  knn_query_direct_body<KMAX, KP>(a.w[j], a.ptr_src[j], B, nullptr, 0, a.qsorted[j], a.ptr_qry[j], a.n_qry[j], k,
                                  a.idx_out[j], nullptr, sorted_io,
                                  (int64_t)(blockIdx.x - a.wg_start[j]) * 256 + threadIdx.x);
}

// ------------------------------------------------------------------------------------------
// query, deferred insertion (the default for k > 4)
//
// Same search (per-lane ring walk over the xy grid, same keys, same total order => bit-identical results), different
// inner loop.  The direct kernel above runs its 2*KMAX-instruction sorted insertion whenever ANY of the 64 lanes
// improves its list — measured ~200 times per wavefront at K = 16 although a lane improves only ~55 times — and
// every candidate slot pays key construction, a 64-bit compare and a divergent branch.  Here a candidate costs a
// distance, one fp32 compare against the lane's current k-th distance and, if it passes, an 8-byte append to a
// per-lane queue in LDS ([slot][lane]: conflict-free).  Queues are drained into the sorted register list together —
// when a queue is about to fill and at the end of every ring, where the termination test needs the exact k-th
// distance — so the insertion chain runs (max queue length over the lanes) times per drain instead of once per
// improving slot of any lane, and the threshold tightens after every drain.  One wavefront per workgroup: no
// barriers, the hardware balances 3 200 independent wavefronts over the CUs.
// ------------------------------------------------------------------------------------------
template <int KMAX, class KP, int QD>
__device__ __forceinline__ void knn_query_queue_body(
    const KnnWs& w, const int64_t* __restrict__ ptr_src, int B, const float* __restrict__ pos_qry, int qstride,
    const float4* __restrict__ qsorted, const int64_t* __restrict__ ptr_qry, int64_t n_qry, int k,
    int* __restrict__ idx_out, float* __restrict__ d2_out, int flags, int64_t wg_in, int64_t nblk_in) {
  const int sorted_io = flags & 1;  // bit 1: idx_out / d2_out are 16-byte aligned (vector stores allowed)
  typedef typename KP::T KT;
  __shared__ KT queue[QD][64 * KNNQ_WAVES];
  const int lane = threadIdx.x;  // (column of the queue; KNNQ_WAVES independent wavefronts per workgroup, no barriers)
  // XCD-aware order: the dispatcher deals consecutive workgroups round-robin over the 8 XCDs (observed; affects speed
  // only), so workgroup b takes the queries of chunk (b % 8): every XCD then walks one contiguous eighth of the
  // (cell-sorted) queries — two whole tiles at BASELINE config 2 — and its private L2 holds just those tiles' records
  // instead of all of them (memory waits were ~50 % of the wave cycles with the plain order, profiles/r02c_*)
  int64_t wg = wg_in * KNNQ_WAVES + (threadIdx.x >> 6);
  {
    const int64_t nblk = nblk_in * KNNQ_WAVES, q8 = nblk >> 3, r8 = nblk & 7;
    const int64_t xcd = wg & 7, i8 = wg >> 3;
    wg = xcd * q8 + (xcd < r8 ? xcd : r8) + i8;
  }
  const int64_t t = wg * 64 + (lane & 63);
  if (t >= n_qry) return;  // (the drains below are per-lane loops: lanes that leave early are simply inactive)
  int lo = 0, hi = B;
  while (hi - lo > 1) {
    int mid = (lo + hi) >> 1;
    if (ptr_qry[mid] <= t) lo = mid; else hi = mid;
  }
  const int b = lo;
  float qx, qy, qz;
  int64_t orow;
  if (qsorted) {
    float4 q = qsorted[t];
    qx = q.x; qy = q.y; qz = q.z; orow = sorted_io ? t : (int64_t)__float_as_int(q.w);
  } else {
    const float* p = pos_qry + t * qstride;
    qx = p[0]; qy = p[1]; qz = p[2]; orow = t;
  }
  const float* gp = w.gridp + (size_t)b * GP_STRIDE;
  const float gx0 = gp[0], gy0 = gp[1], inv_h = gp[2], h = gp[3], eps = gp[4];
  const int Gx = ((const int*)gp)[5], Gy = ((const int*)gp)[6], n = ((const int*)gp)[7];
  const int* cs = w.cell_start + (size_t)b * (CELLS_MAX + 1);
  const float4* sorted = w.sorted + ptr_src[b];

  KT best[KMAX];
#pragma unroll
  for (int j = 0; j < KMAX; ++j) best[j] = KP::empty();
  int cnt = 0;                       // entries in this lane's queue
  float kth = __builtin_inff();      // this lane's current k-th squared distance (+inf while the list is not full)

  auto chain = [&](KT key) {
#pragma unroll
    for (int j = 0; j < KMAX; ++j) {
      if constexpr (KP::IS_F64) {
        KT hi2;
        asm("v_max_f64 %0, %1, %2" : "=&v"(hi2) : "v"(best[j]), "v"(key));
        asm("v_min_f64 %0, %0, %1" : "+v"(best[j]) : "v"(key));
        key = hi2;
      } else {
        const KT cur = best[j];
        const bool lt = key < cur;
        best[j] = lt ? key : cur;
        key = lt ? cur : key;
      }
    }
  };
  auto drain = [&]() {
    // divergent trip count: the wavefront runs max(cnt) insertion chains, all lanes in step.  KNNQ_DRAIN keys per
    // trip: their chains are independent up to a one-slot skew, so the scheduler can interleave them (a lone chain
    // is KMAX dependent v_max_f64 long)
    for (int i = 0; i < cnt; i += KNNQ_DRAIN) {
      KT key[KNNQ_DRAIN];
#pragma unroll
      for (int u = 0; u < KNNQ_DRAIN; ++u) key[u] = queue[(i + u) & (QD - 1)][lane];
#pragma unroll
      for (int u = 0; u < KNNQ_DRAIN; ++u) chain(u == 0 || i + u < cnt ? key[u] : KP::empty());
    }
    cnt = 0;
    // k-th key = the largest of the first k (the list is ascending; empty slots sort above every key).  Written as
    // a max over the high words so that a run-time k costs KMAX selects, not a register array spilled to scratch
    // for dynamic indexing
    unsigned hw = KP::hi32(best[KMAX - 1]);
    if (k < KMAX) {
      hw = 0u;
#pragma unroll
      for (int j = 0; j < KMAX; ++j) {
        const unsigned v = j < k ? KP::hi32(best[j]) : 0u;
        hw = v > hw ? v : hw;
      }
    }
    kth = hw == KP::hi32(KP::empty()) ? __builtin_inff() : __uint_as_float(KP::d2bits_of_hi(hw));
  };

  if (n > 0) {
    const int cx = min(Gx - 1, max(0, (int)((qx - gx0) * inv_h)));
    const int cy = min(Gy - 1, max(0, (int)((qy - gy0) * inv_h)));
    for (int R = 0;; ++R) {
#if KNNQ_PIPE
      // The cells a ring adds are 4R runs of the sorted array (row cy-R, row cy+R, then the two side cells of every row in
      // between).  Walked one by one, a run costs two dependent round trips before its first candidate can be examined
      // (the cell bounds, then the records): 26 exposed latencies for rings 0-2, a large part of a wavefront's life
      // (profiles/r03_knn_staged.log: 175 candidates per lane take 112 us).  Here the bounds of run r+1 are loaded while run
      // r is scanned, and the last batch of run r already fetches the first records of run r+1: one exposure per ring.
      // (Runs outside the grid read cs[0] twice: an empty run.  Candidate order does not matter: the list is a set.)
      const int nr = R == 0 ? 1 : 4 * R;
      auto run_bounds = [&](int r, int& a, int& b) {
        int yy, xa, xb;
        if (r < 2) {
          yy = r == 0 ? cy - R : cy + R;
          xa = max(cx - R, 0); xb = min(cx + R, Gx - 1);
        } else {
          const int m = r - 2;
          yy = cy - R + 1 + (m >> 1);
          xa = xb = (m & 1) ? cx + R : cx - R;
        }
        const bool ok = yy >= 0 && yy < Gy && xa >= 0 && xb < Gx;
        const int ia = ok ? yy * Gx + xa : 0, ib = ok ? yy * Gx + xb + 1 : 0;
        a = cs[ia]; b = cs[ib];
      };
      const int nlast = n - 1;
      int p0, p1;
      run_bounds(0, p0, p1);
      float4 nx[KNNQ_UNROLL];
#pragma unroll
      for (int u = 0; u < KNNQ_UNROLL; ++u) nx[u] = sorted[min(p0 + u, nlast)];
      for (int r = 0; r < nr; ++r) {
        int q0 = 0, q1 = 0;
        if (r + 1 < nr) run_bounds(r + 1, q0, q1);  // in flight while this run is scanned
        for (int p = p0; p < p1; p += KNNQ_UNROLL) {
          float4 s[KNNQ_UNROLL];
#pragma unroll
          for (int u = 0; u < KNNQ_UNROLL; ++u) s[u] = nx[u];
          const int pn = p + KNNQ_UNROLL < p1 ? p + KNNQ_UNROLL : q0;  // last batch of the run: the next run's first records
#pragma unroll
          for (int u = 0; u < KNNQ_UNROLL; ++u) nx[u] = sorted[min(pn + u, nlast)];
#pragma unroll
          for (int u = 0; u < KNNQ_UNROLL; ++u) asm volatile("" : "+v"(nx[u].w));  // whole 16-byte loads, issued together
          if (__builtin_amdgcn_ballot_w64(cnt > QD - KNNQ_UNROLL) != 0) drain();
#pragma unroll
          for (int u = 0; u < KNNQ_UNROLL; ++u) {
            const float d2 = dist2_exact(qx, qy, qz, s[u]);
            // !(d2 > kth): ties with the current k-th distance go through the exact (d2, row) order in the drain
            if (p + u < p1 && !(d2 > kth)) {
              queue[cnt][lane] = KP::make(d2, __float_as_int(s[u].w));
              ++cnt;
            }
          }
        }
        if (p1 <= p0) {  // an empty run fetched nothing for its successor
#pragma unroll
          for (int u = 0; u < KNNQ_UNROLL; ++u) nx[u] = sorted[min(q0 + u, nlast)];
        }
        p0 = q0; p1 = q1;
      }
#else
      for (int dy = -R; dy <= R; ++dy) {
        const int yy = cy + dy;
        if (yy < 0 || yy >= Gy) continue;
        const bool edge = (dy == -R || dy == R);
        for (int sg = 0; sg < (edge ? 1 : 2); ++sg) {
          int xa, xb;
          if (edge) {
            xa = max(cx - R, 0); xb = min(cx + R, Gx - 1);
          } else {
            xa = xb = (sg == 0 ? cx - R : cx + R);
            if (xa < 0 || xa >= Gx) continue;
          }
          const int p0 = cs[yy * Gx + xa], p1 = cs[yy * Gx + xb + 1];
          // (reads run up to 2*KNNQ_UNROLL-1 records past p1: still inside the workspace — the sorted array is
          // followed by the perm / inv arrays — and masked out below).  The loads of batch i+1 are issued before batch
          // i is consumed: two batches of 16-byte loads in flight per lane
#if KNNQ_PP
          // two register sets used in turn: the loads of the next KNNQ_UNROLL records are in flight while the current ones
          // are examined, with no register copies and no artificial use of the loaded values (round 2's loop pinned the
          // prefetched registers with an empty asm statement — which made the compiler wait for them, vmcnt(0), right
          // after issuing them: no overlap at all, and 16 v_mov per trip to rotate the registers; ISA of round 3)
          auto examine = [&](const float4 (&s)[KNNQ_UNROLL], int p) {
            if (__builtin_amdgcn_ballot_w64(cnt > QD - KNNQ_UNROLL) != 0) drain();
#pragma unroll
            for (int u = 0; u < KNNQ_UNROLL; ++u) {
              const float d2 = dist2_exact(qx, qy, qz, s[u]);
              if (p + u < p1 && !(d2 > kth)) {
                queue[cnt][lane] = KP::make(d2, __float_as_int(s[u].w));
                ++cnt;
              }
            }
          };
          if (p1 > p0) {
            // (unclamped addresses: ONE base per trip + immediate offsets.  Reads run up to 3*KNNQ_UNROLL-1 records past p1:
            // still inside the workspace — every array of it is padded to 256 bytes and the sorted array is followed by
            // the perm / inv arrays — and masked out in examine().  Clamping every index cost 16 VALU instructions per
            // trip: +29 % instructions in a kernel whose bound is instruction issue, profiles/r03m_*)
            float4 ra[KNNQ_UNROLL], rb[KNNQ_UNROLL];
#pragma unroll
            for (int u = 0; u < KNNQ_UNROLL; ++u) ra[u] = sorted[p0 + u];
            for (int p = p0; p < p1; p += 2 * KNNQ_UNROLL) {
#pragma unroll
              for (int u = 0; u < KNNQ_UNROLL; ++u) rb[u] = sorted[p + KNNQ_UNROLL + u];
              examine(ra, p);
#pragma unroll
              for (int u = 0; u < KNNQ_UNROLL; ++u) ra[u] = sorted[p + 2 * KNNQ_UNROLL + u];
              examine(rb, p + KNNQ_UNROLL);
            }
          }
#else
          float4 nx[KNNQ_UNROLL];
#pragma unroll
          for (int u = 0; u < KNNQ_UNROLL; ++u) nx[u] = sorted[p0 + u];
          for (int p = p0; p < p1; p += KNNQ_UNROLL) {
            float4 s[KNNQ_UNROLL];
#pragma unroll
            for (int u = 0; u < KNNQ_UNROLL; ++u) s[u] = nx[u];
#pragma unroll
            for (int u = 0; u < KNNQ_UNROLL; ++u) nx[u] = sorted[p + KNNQ_UNROLL + u];
#pragma unroll
            for (int u = 0; u < KNNQ_UNROLL; ++u) asm volatile("" : "+v"(nx[u].w));  // whole 16-byte loads, issued together
            if (__builtin_amdgcn_ballot_w64(cnt > QD - KNNQ_UNROLL) != 0) drain();
#pragma unroll
            for (int u = 0; u < KNNQ_UNROLL; ++u) {
              const float d2 = dist2_exact(qx, qy, qz, s[u]);
              // !(d2 > kth): ties with the current k-th distance go through the exact (d2, row) order in the drain
              if (p + u < p1 && !(d2 > kth)) {
                queue[cnt][lane] = KP::make(d2, __float_as_int(s[u].w));
                ++cnt;
              }
            }
          }
#endif
        }
      }
#endif
      drain();
      const bool covers = (cx - R <= 0) && (cx + R >= Gx - 1) && (cy - R <= 0) && (cy + R >= Gy - 1);
      if (covers) break;
      float bound = 3.4e38f;
      if (cx - R > 0) bound = fminf(bound, qx - (gx0 + (float)(cx - R) * h));
      if (cx + R < Gx - 1) bound = fminf(bound, (gx0 + (float)(cx + R + 1) * h) - qx);
      if (cy - R > 0) bound = fminf(bound, qy - (gy0 + (float)(cy - R) * h));
      if (cy + R < Gy - 1) bound = fminf(bound, (gy0 + (float)(cy + R + 1) * h) - qy);
      bound = fmaxf(bound - eps, 0.f);
      if (kth <= bound * bound) break;  // (kth = +inf while fewer than k neighbours are known: keeps searching)
    }
  }
  // ---- results: all slot translations (w.inv) in flight together, rows stored 16 bytes at a time when k == KMAX
  int ids[KMAX];
#pragma unroll
  for (int j = 0; j < KMAX; ++j) ids[j] = KP::is_empty(best[j]) ? -1 : KP::row(best[j]);
  if (sorted_io) {  // neighbours selected by (d2, original row); reported as cell-sorted slots
    int tr[KMAX];
#pragma unroll
    for (int j = 0; j < KMAX; ++j) tr[j] = w.inv[ids[j] < 0 ? 0 : ids[j]];
#pragma unroll
    for (int j = 0; j < KMAX; ++j) ids[j] = ids[j] < 0 ? -1 : tr[j];
  }
  int* io = idx_out + orow * k;
  if (k == KMAX && KMAX % 4 == 0 && (flags & 2)) {
#pragma unroll
    for (int j = 0; j < KMAX; j += 4) *(int4*)(io + j) = make_int4(ids[j], ids[j + 1], ids[j + 2], ids[j + 3]);
    if (d2_out) {
      float* dq = d2_out + orow * k;
#pragma unroll
      for (int j = 0; j < KMAX; j += 4) {
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
          v[u] = KP::is_empty(best[j + u]) ? __builtin_inff() : __uint_as_float(KP::d2bits(best[j + u]));
        *(float4*)(dq + j) = make_float4(v[0], v[1], v[2], v[3]);
      }
    }
  } else {
#pragma unroll
    for (int j = 0; j < KMAX; ++j) {
      if (j < k) {
        io[j] = ids[j];
        if (d2_out)
          d2_out[orow * k + j] = KP::is_empty(best[j]) ? __builtin_inff() : __uint_as_float(KP::d2bits(best[j]));
      }
    }
  }
}

template <int KMAX, class KP, int QD>
__global__ __launch_bounds__(64 * KNNQ_WAVES, KNNQ_MINW) void knn_query_queue_kernel(
    KnnWs w, const int64_t* __restrict__ ptr_src, int B, const float* __restrict__ pos_qry, int qstride,
    const float4* __restrict__ qsorted, const int64_t* __restrict__ ptr_qry, int64_t n_qry, int k,
    int* __restrict__ idx_out, float* __restrict__ d2_out, int flags) {
  knn_query_queue_body<KMAX, KP, QD>(w, ptr_src, B, pos_qry, qstride, qsorted, ptr_qry, n_qry, k, idx_out, d2_out, flags,
                                     blockIdx.x, gridDim.x);
}

template <int KMAX, class KP, int QD>
__global__ __launch_bounds__(64 * KNNQ_WAVES, KNNQ_MINW) void knn_query_queue_batch_kernel(KnnBatch a, int B, int k, int flags) {
  const int j = knn_batch_job(a);
  knn_query_queue_body<KMAX, KP, QD>(a.w[j], a.ptr_src[j], B, nullptr, 0, a.qsorted[j], a.ptr_qry[j], a.n_qry[j], k,
                                     a.idx_out[j], nullptr, flags, (int64_t)(blockIdx.x - a.wg_start[j]),
                                     (int64_t)(a.wg_start[j + 1] - a.wg_start[j]));
}

// ------------------------------------------------------------------------------------------
// query, LDS WINDOW (round 3): the deferred-insertion search with the candidates read from LDS.
//
// The 64 queries of a wavefront are consecutive in cell-sorted order: a run of ~9 cells of one grid row (two runs when
// the wavefront wraps into the next row).  Everything their first rings touch lies in a small window of the grid: the
// rows cy-RW .. cy+RW and the columns of the run widened by RW on both sides — 7 x ~15 cells, ~750 records, 12 KB.  The
// wavefront copies that window into LDS ONCE with coalesced loads (each window row is one contiguous run of the sorted
// array) together with the window's cell boundaries, and every lane then does ITS OWN ring walk — the same rings, the
// same candidates, the same keys, the same termination test => bit-identical tables — but its candidate fetches and
// cell-boundary look-ups are LDS reads (~100 cycles) instead of per-lane global gathers behind ~850 cycles of memory
// wait per batch of four (profiles/r02c_*: half of a wavefront's life).  Rings that leave the window (R > RW: 2.7 % of
// the queries at RW = 3) fall back to global loads for those runs.  Lanes of a wavefront that sit in different grid rows
// or clouds are processed as successive segments, each with its own window.
// ------------------------------------------------------------------------------------------
#ifndef M3D_KNN_LDS_DEFAULT
#define M3D_KNN_LDS_DEFAULT 0
#endif
#ifndef KNNL_CAP
#define KNNL_CAP 1024   // records of a window (16 KB of LDS); a window that does not fit shrinks its RW
#endif
#ifndef KNNL_RW
#define KNNL_RW 3
#endif
#ifndef KNNL_MINW
#define KNNL_MINW 2
#endif
#define KNNL_MAXR (2 * KNNL_RW + 1)
#define KNNL_MAXC 32
template <int KMAX, class KP, int QD>
__global__ __launch_bounds__(64, KNNL_MINW) void knn_query_lds_kernel(
    KnnWs w, const int64_t* __restrict__ ptr_src, int B, const float4* __restrict__ qsorted,
    const int64_t* __restrict__ ptr_qry, int64_t n_qry, int k, int* __restrict__ idx_out, float* __restrict__ d2_out,
    int flags) {
  const int sorted_io = flags & 1;
  typedef typename KP::T KT;
  __shared__ KT queue[QD][64];
  __shared__ float4 rec[KNNL_CAP];
  __shared__ int lcs[KNNL_MAXR][KNNL_MAXC + 1];
  const int lane = threadIdx.x;
  int64_t wg = blockIdx.x;
  {  // XCD-aware order (see knn_query_queue_body)
    const int64_t nblk = gridDim.x, q8 = nblk >> 3, r8 = nblk & 7;
    const int64_t xcd = wg & 7, i8 = wg >> 3;
    wg = xcd * q8 + (xcd < r8 ? xcd : r8) + i8;
  }
  const int64_t t = wg * 64 + lane;
  const bool valid = t < n_qry;
  const int64_t tc = valid ? t : n_qry - 1;
  int b;
  {
    int lo = 0, hi = B;
    while (hi - lo > 1) {
      int mid = (lo + hi) >> 1;
      if (ptr_qry[mid] <= tc) lo = mid; else hi = mid;
    }
    b = lo;
  }
  const float4 q = qsorted[tc];
  const float qx = q.x, qy = q.y, qz = q.z;
  int cx, cy;
  {
    const float* gpl = w.gridp + (size_t)b * GP_STRIDE;
    const int Gxl = ((const int*)gpl)[5], Gyl = ((const int*)gpl)[6];
    cx = min(Gxl - 1, max(0, (int)((qx - gpl[0]) * gpl[2])));
    cy = min(Gyl - 1, max(0, (int)((qy - gpl[1]) * gpl[2])));
  }
  KT best[KMAX];
#pragma unroll
  for (int j = 0; j < KMAX; ++j) best[j] = KP::empty();
  int cnt = 0;
  float kth = __builtin_inff();

  auto chain = [&](KT key) {
#pragma unroll
    for (int j = 0; j < KMAX; ++j) {
      if constexpr (KP::IS_F64) {
        KT hi2;
        asm("v_max_f64 %0, %1, %2" : "=&v"(hi2) : "v"(best[j]), "v"(key));
        asm("v_min_f64 %0, %0, %1" : "+v"(best[j]) : "v"(key));
        key = hi2;
      } else {
        const KT cur = best[j];
        const bool lt = key < cur;
        best[j] = lt ? key : cur;
        key = lt ? cur : key;
      }
    }
  };
  auto drain = [&]() {
    for (int i = 0; i < cnt; i += KNNQ_DRAIN) {
      KT key[KNNQ_DRAIN];
#pragma unroll
      for (int u = 0; u < KNNQ_DRAIN; ++u) key[u] = queue[(i + u) & (QD - 1)][lane];
#pragma unroll
      for (int u = 0; u < KNNQ_DRAIN; ++u) chain(u == 0 || i + u < cnt ? key[u] : KP::empty());
    }
    cnt = 0;
    unsigned hw = KP::hi32(best[KMAX - 1]);
    if (k < KMAX) {
      hw = 0u;
#pragma unroll
      for (int j = 0; j < KMAX; ++j) {
        const unsigned v = j < k ? KP::hi32(best[j]) : 0u;
        hw = v > hw ? v : hw;
      }
    }
    kth = hw == KP::hi32(KP::empty()) ? __builtin_inff() : __uint_as_float(KP::d2bits_of_hi(hw));
  };

  unsigned long long todo = __builtin_amdgcn_ballot_w64(valid);
  while (todo != 0ull) {
    // ---- next segment: the pending lanes that share the leader's cloud and grid row
    const int lead = __builtin_ctzll(todo);
    const int b0 = __builtin_amdgcn_readlane(b, lead), cy0 = __builtin_amdgcn_readlane(cy, lead);
    const bool mine = valid && b == b0 && cy == cy0 && ((todo >> lane) & 1ull);
    todo &= ~__builtin_amdgcn_ballot_w64(mine);
    const float* gp = w.gridp + (size_t)b0 * GP_STRIDE;
    const float gx0 = gp[0], gy0 = gp[1], h = gp[3], eps = gp[4];
    const int Gx = ((const int*)gp)[5], Gy = ((const int*)gp)[6], n = ((const int*)gp)[7];
    const int* cs = w.cell_start + (size_t)b0 * (CELLS_MAX + 1);
    const float4* sorted = w.sorted + ptr_src[b0];
    if (n <= 0) continue;
    int cxmin = mine ? cx : 0x7fffffff, cxmax = mine ? cx : -1;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      cxmin = min(cxmin, __shfl_xor(cxmin, o, 64));
      cxmax = max(cxmax, __shfl_xor(cxmax, o, 64));
    }
    cxmin = __builtin_amdgcn_readfirstlane(cxmin);
    cxmax = __builtin_amdgcn_readfirstlane(cxmax);
    // ---- the window: the largest RW <= KNNL_RW whose columns and records fit (wave-uniform)
    int wx0 = 0, wx1 = -1, wy0 = 0, wy1 = -1, nrows = 0;
    int ra = 0, rb = 0;  // lane r < nrows: bounds of window row r in the sorted array
    for (int rw = KNNL_RW; rw >= 0; --rw) {
      const int x0 = max(cxmin - rw, 0), x1 = min(cxmax + rw, Gx - 1);
      const int y0 = max(cy0 - rw, 0), y1 = min(cy0 + rw, Gy - 1);
      const int nr = y1 - y0 + 1;
      if (x1 - x0 + 1 > KNNL_MAXC) continue;
      int a = 0, bb = 0;
      if (lane < nr) { a = cs[(y0 + lane) * Gx + x0]; bb = cs[(y0 + lane) * Gx + x1 + 1]; }
      int tot = bb - a;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o, 64);
      if (__builtin_amdgcn_readfirstlane(tot) <= KNNL_CAP) {
        wx0 = x0; wx1 = x1; wy0 = y0; wy1 = y1; nrows = nr; ra = a; rb = bb;
        break;
      }
    }
    // ---- stage the window rows (coalesced: lane i takes record i of the run) and the local cell boundaries
    {
      int off = 0;
      for (int r = 0; r < nrows; ++r) {
        const int a = __builtin_amdgcn_readlane(ra, r), len = __builtin_amdgcn_readlane(rb, r) - a;
        for (int i = lane; i < len; i += 64) rec[off + i] = sorted[a + i];
        const int ncol = wx1 - wx0 + 1;
        if (lane <= ncol) lcs[r][lane] = off + (cs[(wy0 + r) * Gx + wx0 + lane] - a);
        off += len;
      }
    }
    __syncthreads();  // (one wavefront per workgroup: orders the LDS writes before the reads below)
    if (mine) {
      for (int R = 0;; ++R) {
        for (int dy = -R; dy <= R; ++dy) {
          const int yy = cy + dy;
          if (yy < 0 || yy >= Gy) continue;
          const bool edge = (dy == -R || dy == R);
          for (int sg = 0; sg < (edge ? 1 : 2); ++sg) {
            int xa, xb;
            if (edge) {
              xa = max(cx - R, 0); xb = min(cx + R, Gx - 1);
            } else {
              xa = xb = (sg == 0 ? cx - R : cx + R);
              if (xa < 0 || xa >= Gx) continue;
            }
            const bool inw = yy >= wy0 && yy <= wy1 && xa >= wx0 && xb <= wx1;
            if (inw) {
              const int p0 = lcs[yy - wy0][xa - wx0], p1 = lcs[yy - wy0][xb + 1 - wx0];
              for (int p = p0; p < p1; p += KNNQ_UNROLL) {
                float4 s[KNNQ_UNROLL];
#pragma unroll
                for (int u = 0; u < KNNQ_UNROLL; ++u) s[u] = rec[min(p + u, p1 - 1)];
                if (__builtin_amdgcn_ballot_w64(cnt > QD - KNNQ_UNROLL) != 0) drain();
#pragma unroll
                for (int u = 0; u < KNNQ_UNROLL; ++u) {
                  const float d2 = dist2_exact(qx, qy, qz, s[u]);
                  if (p + u < p1 && !(d2 > kth)) {
                    queue[cnt][lane] = KP::make(d2, __float_as_int(s[u].w));
                    ++cnt;
                  }
                }
              }
            } else {
              const int p0 = cs[yy * Gx + xa], p1 = cs[yy * Gx + xb + 1];
              for (int p = p0; p < p1; p += KNNQ_UNROLL) {
                float4 s[KNNQ_UNROLL];
#pragma unroll
                for (int u = 0; u < KNNQ_UNROLL; ++u) s[u] = sorted[min(p + u, p1 - 1)];
                if (__builtin_amdgcn_ballot_w64(cnt > QD - KNNQ_UNROLL) != 0) drain();
#pragma unroll
                for (int u = 0; u < KNNQ_UNROLL; ++u) {
                  const float d2 = dist2_exact(qx, qy, qz, s[u]);
                  if (p + u < p1 && !(d2 > kth)) {
                    queue[cnt][lane] = KP::make(d2, __float_as_int(s[u].w));
                    ++cnt;
                  }
                }
              }
            }
          }
        }
        drain();
        const bool covers = (cx - R <= 0) && (cx + R >= Gx - 1) && (cy - R <= 0) && (cy + R >= Gy - 1);
        if (covers) break;
        float bound = 3.4e38f;
        if (cx - R > 0) bound = fminf(bound, qx - (gx0 + (float)(cx - R) * h));
        if (cx + R < Gx - 1) bound = fminf(bound, (gx0 + (float)(cx + R + 1) * h) - qx);
        if (cy - R > 0) bound = fminf(bound, qy - (gy0 + (float)(cy - R) * h));
        if (cy + R < Gy - 1) bound = fminf(bound, (gy0 + (float)(cy + R + 1) * h) - qy);
        bound = fmaxf(bound - eps, 0.f);
        if (kth <= bound * bound) break;
      }
    }
    __syncthreads();  // the next segment restages the window
  }
  if (!valid) return;
  const int64_t orow = sorted_io ? t : (int64_t)__float_as_int(q.w);
  int ids[KMAX];
#pragma unroll
  for (int j = 0; j < KMAX; ++j) ids[j] = KP::is_empty(best[j]) ? -1 : KP::row(best[j]);
  if (sorted_io) {
    int tr[KMAX];
#pragma unroll
    for (int j = 0; j < KMAX; ++j) tr[j] = w.inv[ids[j] < 0 ? 0 : ids[j]];
#pragma unroll
    for (int j = 0; j < KMAX; ++j) ids[j] = ids[j] < 0 ? -1 : tr[j];
  }
  int* io = idx_out + orow * k;
  if (k == KMAX && KMAX % 4 == 0 && (flags & 2)) {
#pragma unroll
    for (int j = 0; j < KMAX; j += 4) *(int4*)(io + j) = make_int4(ids[j], ids[j + 1], ids[j + 2], ids[j + 3]);
    if (d2_out) {
      float* dq = d2_out + orow * k;
#pragma unroll
      for (int j = 0; j < KMAX; j += 4) {
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
          v[u] = KP::is_empty(best[j + u]) ? __builtin_inff() : __uint_as_float(KP::d2bits(best[j + u]));
        *(float4*)(dq + j) = make_float4(v[0], v[1], v[2], v[3]);
      }
    }
  } else {
#pragma unroll
    for (int j = 0; j < KMAX; ++j) {
      if (j < k) {
        io[j] = ids[j];
        if (d2_out)
          d2_out[orow * k + j] = KP::is_empty(best[j]) ? __builtin_inff() : __uint_as_float(KP::d2bits(best[j]));
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// query, STAGED (round 3; opt-in, M3D_KNN_STAGED=1): the deferred-insertion search above, cut into launches by ring
// radius, with the unfinished queries compacted between the launches.
//
// Why it was built: a lane's ring walk ends when ITS k-th distance is inside the explored block, a wavefront ends with its
// slowest lane — the mean lane scans ~200 candidates, the slowest of a wavefront ~570 (rings 0..4).  What the measurement
// said (profiles/r03_knn_staged.log): there is no long tail to cut — 24.6 / 49.7 / 23.0 / 2.6 % of the queries of a
// Lidar-HD-shaped tile close at ring 1 / 2 / 3 / 4 — and a group stage costs what the idle lanes it removes cost, so the
// single launch stays the default.  The design:
//   stage 0   every query, one lane each, rings 0 .. r0 (uniform work); a query whose search is still open leaves its
//             state — the sorted key list, its position, cloud and query index — in a pool in HBM, appended at a slot
//             from ONE atomic per wavefront (ballot + prefix);
//   stage s   reads the pool of stage s-1 DENSELY (no idle lanes) and gives every open query a GROUP of G lanes that
//             split each ring's candidates (lane g takes records p0+g, p0+g+G, ...: adjacent lanes read adjacent
//             records); candidates that pass the group's shared k-th distance go to the lane's LDS queue and the group
//             OWNER merges the G queues into the key list at the end of the ring — no partial lists, no merge network;
//             the late rings, which are long (8 R columns) and concern few queries, are scanned G-wide instead of serially;
//   last      the same with the largest group, looping until every query is closed.
// Same keys, same total order, same termination test per query => tables bit-identical to the kernels above.
// ------------------------------------------------------------------------------------------
#define KNNS_MAX_STAGES 4
struct KnnPoolBuf {
  float4* q;    // [cap] (x, y, z, bits of the query's index t)
  int* cloud;   // [cap]
  void* best;   // KT [KMAX][cap]
};
struct KnnStageArgs {
  KnnPoolBuf in, out;
  const unsigned* cnt_in;  // open queries left by the previous stage
  unsigned* cnt_out;
  int64_t cap;
  int r_first, r_last;  // rings of this stage (r_last < 0: until closed)
};

template <int KMAX, class KP, int QD, int G, bool FIRST, bool LAST>
__global__ __launch_bounds__(64, (KMAX <= 16 ? KNNQ_MINW : 2)) void knn_stage_kernel(
    KnnWs w, const int64_t* __restrict__ ptr_src, int B, const float4* __restrict__ qsorted,
    const int64_t* __restrict__ ptr_qry, int64_t n_qry, int k, int* __restrict__ idx_out, float* __restrict__ d2_out,
    int flags, KnnStageArgs sa) {
  static_assert(!FIRST || G == 1, "stage 0 walks one query per lane");
  static_assert((G & (G - 1)) == 0 && G <= 64, "group size: a power of two");
  typedef typename KP::T KT;
  constexpr int QPW = 64 / G;  // queries per wavefront
  const int sorted_io = flags & 1;
  __shared__ KT queue[QD][64];
  __shared__ int qcnt[64];
  const int lane = threadIdx.x;
  const int g = lane & (G - 1), own = lane & ~(G - 1);
  const bool owner = g == 0;
  const int64_t count = FIRST ? n_qry : (int64_t)*sa.cnt_in;
  const int64_t nwork = (count + QPW - 1) / QPW;
  for (int64_t wg0 = blockIdx.x; wg0 < nwork; wg0 += gridDim.x) {
    int64_t wg = wg0;
    if constexpr (FIRST) {  // XCD-aware order (see knn_query_queue_body)
      const int64_t nblk = nwork, q8 = nblk >> 3, r8 = nblk & 7;
      const int64_t xcd = wg & 7, i8 = wg >> 3;
      wg = xcd * q8 + (xcd < r8 ? xcd : r8) + i8;
    }
    const int64_t slot = wg * QPW + (lane / G);
    const bool valid = slot < count;
    const int64_t sl = valid ? slot : count - 1;
    // ---- the query and its list
    float qx, qy, qz;
    int t, b;
    KT best[KMAX];
    if constexpr (FIRST) {
      t = (int)sl;
      int lo = 0, hi = B;
      while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (ptr_qry[mid] <= (int64_t)t) lo = mid; else hi = mid;
      }
      b = lo;
      const float4 q = qsorted[t];
      qx = q.x; qy = q.y; qz = q.z;
#pragma unroll
      for (int j = 0; j < KMAX; ++j) best[j] = KP::empty();
    } else {
      const float4 q = sa.in.q[sl];
      qx = q.x; qy = q.y; qz = q.z; t = __float_as_int(q.w);
      b = sa.in.cloud[sl];
      const KT* bi = (const KT*)sa.in.best + sl;
#pragma unroll
      for (int j = 0; j < KMAX; ++j) best[j] = owner ? bi[(size_t)j * sa.cap] : KP::empty();
    }
    const float* gp = w.gridp + (size_t)b * GP_STRIDE;
    const float gx0 = gp[0], gy0 = gp[1], inv_h = gp[2], h = gp[3], eps = gp[4];
    const int Gx = ((const int*)gp)[5], Gy = ((const int*)gp)[6], n = ((const int*)gp)[7];
    const int* cs = w.cell_start + (size_t)b * (CELLS_MAX + 1);
    const float4* sorted = w.sorted + ptr_src[b];
    int cnt = 0;
    float kth = __builtin_inff();

    auto chain = [&](KT key) {
#pragma unroll
      for (int j = 0; j < KMAX; ++j) {
        if constexpr (KP::IS_F64) {
          KT hi2;
          asm("v_max_f64 %0, %1, %2" : "=&v"(hi2) : "v"(best[j]), "v"(key));
          asm("v_min_f64 %0, %0, %1" : "+v"(best[j]) : "v"(key));
          key = hi2;
        } else {
          const KT cur = best[j];
          const bool lt = key < cur;
          best[j] = lt ? key : cur;
          key = lt ? cur : key;
        }
      }
    };
    auto kth_of_list = [&]() {  // (the owner's list; G > 1: shared with the group below)
      unsigned hw = KP::hi32(best[KMAX - 1]);
      if (k < KMAX) {
        hw = 0u;
#pragma unroll
        for (int j = 0; j < KMAX; ++j) {
          const unsigned v = j < k ? KP::hi32(best[j]) : 0u;
          hw = v > hw ? v : hw;
        }
      }
      float v = hw == KP::hi32(KP::empty()) ? __builtin_inff() : __uint_as_float(KP::d2bits_of_hi(hw));
      if constexpr (G > 1) v = __shfl(v, own, 64);
      kth = v;
    };
    auto drain = [&]() {
      if constexpr (G == 1) {
        for (int i = 0; i < cnt; i += KNNQ_DRAIN) {
          KT key[KNNQ_DRAIN];
#pragma unroll
          for (int u = 0; u < KNNQ_DRAIN; ++u) key[u] = queue[(i + u) & (QD - 1)][lane];
#pragma unroll
          for (int u = 0; u < KNNQ_DRAIN; ++u) chain(u == 0 || i + u < cnt ? key[u] : KP::empty());
        }
      } else {
        // the owner merges the queues of its group (same wavefront: program order is the only ordering needed; the
        // wave barrier keeps the compiler from moving the LDS reads above the writes of the other lanes)
        qcnt[lane] = cnt;
        __builtin_amdgcn_wave_barrier();
        if (owner) {
          for (int gg = 0; gg < G; ++gg) {
            const int c = qcnt[own + gg];
            for (int i = 0; i < c; ++i) chain(queue[i][own + gg]);
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
      cnt = 0;
      kth_of_list();
    };
    if constexpr (!FIRST) kth_of_list();

    bool open = valid && n > 0;  // the search of this lane's query is not closed yet
    if (open) {
      const int cx = min(Gx - 1, max(0, (int)((qx - gx0) * inv_h)));
      const int cy = min(Gy - 1, max(0, (int)((qy - gy0) * inv_h)));
      for (int R = sa.r_first; LAST || R <= sa.r_last; ++R) {
        for (int dy = -R; dy <= R; ++dy) {
          const int yy = cy + dy;
          if (yy < 0 || yy >= Gy) continue;
          const bool edge = (dy == -R || dy == R);
          for (int sg = 0; sg < (edge ? 1 : 2); ++sg) {
            int xa, xb;
            if (edge) {
              xa = max(cx - R, 0); xb = min(cx + R, Gx - 1);
            } else {
              xa = xb = (sg == 0 ? cx - R : cx + R);
              if (xa < 0 || xa >= Gx) continue;
            }
            const int p0 = cs[yy * Gx + xa], p1 = cs[yy * Gx + xb + 1];
            if (p1 <= p0) continue;
            const int last = p1 - 1;
            // lane g of the group takes records p0 + g, p0 + g + G, ...; KNNQ_UNROLL of them per trip, the loads of trip
            // i + 1 issued before trip i is consumed (addresses clamped to the run: masked out below)
            // (the trip count depends on p0 / p1 only: every lane of a group runs the same trips, so the group's lanes
            // are always together when its owner merges their queues)
            float4 nx[KNNQ_UNROLL];
#pragma unroll
            for (int u = 0; u < KNNQ_UNROLL; ++u) nx[u] = sorted[min(p0 + g + u * G, last)];
            for (int pb = p0; pb < p1; pb += KNNQ_UNROLL * G) {
              const int p = pb + g;
              float4 s[KNNQ_UNROLL];
#pragma unroll
              for (int u = 0; u < KNNQ_UNROLL; ++u) s[u] = nx[u];
#pragma unroll
              for (int u = 0; u < KNNQ_UNROLL; ++u) nx[u] = sorted[min(p + (KNNQ_UNROLL + u) * G, last)];
#pragma unroll
              for (int u = 0; u < KNNQ_UNROLL; ++u) asm volatile("" : "+v"(nx[u].w));  // whole 16-byte loads, issued together
              if (__builtin_amdgcn_ballot_w64(cnt > QD - KNNQ_UNROLL) != 0) drain();
#pragma unroll
              for (int u = 0; u < KNNQ_UNROLL; ++u) {
                const float d2 = dist2_exact(qx, qy, qz, s[u]);
                // !(d2 > kth): ties with the current k-th distance go through the exact (d2, row) order in the drain
                if (p + u * G < p1 && !(d2 > kth)) {
                  queue[cnt][lane] = KP::make(d2, __float_as_int(s[u].w));
                  ++cnt;
                }
              }
            }
          }
        }
        drain();
        const bool covers = (cx - R <= 0) && (cx + R >= Gx - 1) && (cy - R <= 0) && (cy + R >= Gy - 1);
        if (covers) { open = false; break; }
        float bound = 3.4e38f;
        if (cx - R > 0) bound = fminf(bound, qx - (gx0 + (float)(cx - R) * h));
        if (cx + R < Gx - 1) bound = fminf(bound, (gx0 + (float)(cx + R + 1) * h) - qx);
        if (cy - R > 0) bound = fminf(bound, qy - (gy0 + (float)(cy - R) * h));
        if (cy + R < Gy - 1) bound = fminf(bound, (gy0 + (float)(cy + R + 1) * h) - qy);
        bound = fmaxf(bound - eps, 0.f);
        if (kth <= bound * bound) { open = false; break; }  // (kth = +inf while fewer than k neighbours are known)
      }
    }
    // ---- open queries go to the next stage's pool: one atomic per wavefront
    if constexpr (!LAST) {
      const bool spill = open && owner;
      const unsigned long long m = __builtin_amdgcn_ballot_w64(spill);
      if (m != 0ull) {
        unsigned base = 0;
        if (lane == 0) base = atomicAdd(sa.cnt_out, (unsigned)__builtin_popcountll(m));
        base = __builtin_amdgcn_readfirstlane(base);
        if (spill) {
          const size_t pos = base + (unsigned)__builtin_popcountll(m & ((1ull << lane) - 1ull));
          sa.out.q[pos] = make_float4(qx, qy, qz, __int_as_float(t));
          sa.out.cloud[pos] = b;
          KT* bo = (KT*)sa.out.best + pos;
#pragma unroll
          for (int j = 0; j < KMAX; ++j) bo[(size_t)j * sa.cap] = best[j];
        }
      }
    }
    // ---- closed queries: results (all slot translations in flight together, rows stored 16 bytes at a time)
    if (valid && owner && !open) {
      const int64_t orow = sorted_io ? (int64_t)t : (int64_t)__float_as_int(qsorted[t].w);
      int ids[KMAX];
#pragma unroll
      for (int j = 0; j < KMAX; ++j) ids[j] = KP::is_empty(best[j]) ? -1 : KP::row(best[j]);
      if (sorted_io) {  // neighbours selected by (d2, original row); reported as cell-sorted slots
        int tr[KMAX];
#pragma unroll
        for (int j = 0; j < KMAX; ++j) tr[j] = w.inv[ids[j] < 0 ? 0 : ids[j]];
#pragma unroll
        for (int j = 0; j < KMAX; ++j) ids[j] = ids[j] < 0 ? -1 : tr[j];
      }
      int* io = idx_out + orow * k;
      if (k == KMAX && KMAX % 4 == 0 && (flags & 2)) {
#pragma unroll
        for (int j = 0; j < KMAX; j += 4) *(int4*)(io + j) = make_int4(ids[j], ids[j + 1], ids[j + 2], ids[j + 3]);
        if (d2_out) {
          float* dq = d2_out + orow * k;
#pragma unroll
          for (int j = 0; j < KMAX; j += 4) {
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
              v[u] = KP::is_empty(best[j + u]) ? __builtin_inff() : __uint_as_float(KP::d2bits(best[j + u]));
            *(float4*)(dq + j) = make_float4(v[0], v[1], v[2], v[3]);
          }
        }
      } else {
#pragma unroll
        for (int j = 0; j < KMAX; ++j) {
          if (j < k) {
            io[j] = ids[j];
            if (d2_out)
              d2_out[orow * k + j] = KP::is_empty(best[j]) ? __builtin_inff() : __uint_as_float(KP::d2bits(best[j]));
          }
        }
      }
    }
  }
}

// m3d_knn_build that also carries one int32 per source row into cell-sorted order: map_out[slot] = map_in[row] (the
// decimation map of a level composed with the next level's order: one launch less per level)
extern "C" int m3d_knn_build_map(const float* pos_src, int32_t pos_stride, const int64_t* ptr_src, int32_t num_clouds,
                                 int64_t n_src, void* ws, const int32_t* map_in, int32_t* map_out, void* stream) {
  if ((map_in == nullptr) != (map_out == nullptr)) return M3D_ERR_INVALID;
  if (!ptr_src || !ws || num_clouds < 0 || n_src < 0 || pos_stride < 3) return M3D_ERR_INVALID;
  if (num_clouds == 0) return M3D_OK;
  if (!pos_src && n_src > 0) return M3D_ERR_INVALID;
  KnnWs w = ws_carve(ws, num_clouds, n_src);
  // points per grid column the cell size aims at (tuning knob; any positive value gives the same exact result)
  static const float cell_target = [] {
    const char* e = getenv("M3D_KNN_CELL_TARGET");
    float v = e ? (float)atof(e) : M3D_KNN_CELL_TARGET;
    return v > 0.f ? v : M3D_KNN_CELL_TARGET;
  }();
  hipLaunchKernelGGL(knn_build_kernel, dim3(num_clouds), dim3(1024), 0, (hipStream_t)stream, pos_src, pos_stride,
                     ptr_src, w, cell_target, map_in, map_out);
  M3D_CHECK_LAUNCH();
  return M3D_OK;
}

extern "C" int m3d_knn_build(const float* pos_src, int32_t pos_stride, const int64_t* ptr_src, int32_t num_clouds,
                             int64_t n_src, void* ws, void* stream) {
  return m3d_knn_build_map(pos_src, pos_stride, ptr_src, num_clouds, n_src, ws, nullptr, nullptr, stream);
}

// key policy of the top-k registers: M3D_KNN_KEYS=u64 (integer compare/select chain) | f64 (v_min/v_max_f64 chain)
static bool knn_f64_keys() {
  const char* e = getenv("M3D_KNN_KEYS");
  if (!e) return M3D_KNN_DEFAULT_F64 != 0;
  return e[0] == 'f';
}

extern "C" int m3d_knn_query(const void* ws, const int64_t* ptr_src, int64_t n_src, int32_t num_clouds,
                             const float* pos_qry, int32_t qry_stride, const void* qry_ws, const int64_t* ptr_qry,
                             int64_t n_qry, int32_t k, int32_t sorted_io, int32_t* idx_out, float* d2_out,
                             void* stream) {
  if (!ws || !ptr_src || !ptr_qry || !idx_out || num_clouds < 0 || n_qry < 0) return M3D_ERR_INVALID;
  if (k < 1 || k > 64) return M3D_ERR_UNSUPPORTED;  // upstream CUDA kNN asserts k <= 100
  if (!pos_qry && !qry_ws && n_qry > 0) return M3D_ERR_INVALID;
  if (pos_qry && qry_stride < 3) return M3D_ERR_INVALID;
  if (sorted_io && !qry_ws) return M3D_ERR_INVALID;  // sorted rows are defined by the query workspace
  if (n_qry == 0 || num_clouds == 0) return M3D_OK;
  KnnWs w = ws_carve((void*)ws, num_clouds, n_src);
  const float4* qs = qry_ws ? ws_carve((void*)qry_ws, num_clouds, n_qry).sorted : nullptr;
  dim3 grid((unsigned)m3d_cdiv(n_qry, 256)), block(256);
  hipStream_t st = (hipStream_t)stream;
  static const bool f64_keys = knn_f64_keys();
  // M3D_KNN_QUEUE=0: the direct-insertion kernel for every k (cross-check of the deferred-insertion kernel);
  const int qflags = (sorted_io ? 1 : 0) | (((((uintptr_t)idx_out) | ((uintptr_t)d2_out)) & 15) == 0 ? 2 : 0);
  // M3D_KNN_QUEUE=1: always, unset: where it measured faster — large query sets (>= 1M (query, neighbour) pairs: level
  // 1 of BASELINE config 2 189 vs 224 us; the K = 32 tiles), while the small deep-level launches (a few waves per CU,
  // latency-bound) stay on the direct kernel (profiles/r02d_knn_ab.log)
  static const int queue_env = getenv("M3D_KNN_QUEUE") ? atoi(getenv("M3D_KNN_QUEUE")) : -1;
  const bool use_queue = queue_env < 0 ? n_qry * (int64_t)k >= (1 << 20) : queue_env != 0;
#define LAUNCH_KP(KM, KP)                                                                                      \
  hipLaunchKernelGGL((knn_query_kernel<KM, KP>), grid, block, 0, st, w, ptr_src, num_clouds, pos_qry, qry_stride, \
                     qs, ptr_qry, n_qry, k, idx_out, d2_out, sorted_io)
#define LAUNCH_Q(KM, KP)                                                                                          \
  hipLaunchKernelGGL((knn_query_queue_kernel<KM, KP, KNNQ_DEPTH>), dim3((unsigned)m3d_cdiv(n_qry, 64 * KNNQ_WAVES)),   \
                     dim3(64 * KNNQ_WAVES), 0, st, w, ptr_src, num_clouds, pos_qry, qry_stride, qs, ptr_qry, n_qry, k,  \
                     idx_out, d2_out, qflags)
  // M3D_KNN_LDS=1 (read at every call): the LDS-window kernel for cell-sorted queries with k > 4 (f64 keys)
  const char* lds_s = getenv("M3D_KNN_LDS");
  const bool use_lds = qs && f64_keys && (lds_s ? atoi(lds_s) != 0 : M3D_KNN_LDS_DEFAULT != 0);
#define LAUNCH_L(KM)                                                                                              \
  hipLaunchKernelGGL((knn_query_lds_kernel<KM, KeyF64, KNNQ_DEPTH>), dim3((unsigned)m3d_cdiv(n_qry, 64)), dim3(64), 0, st, \
                     w, ptr_src, num_clouds, qs, ptr_qry, n_qry, k, idx_out, d2_out, qflags)
#define LAUNCH(KM)                       \
  do {                                   \
    if (f64_keys) LAUNCH_KP(KM, KeyF64); \
    else LAUNCH_KP(KM, KeyU64);          \
  } while (0)
#define LAUNCHQ(KM)                      \
  do {                                   \
    if (use_lds) LAUNCH_L(KM);           \
    else if (!use_queue) LAUNCH(KM);     \
    else if (f64_keys) LAUNCH_Q(KM, KeyF64); \
    else LAUNCH_Q(KM, KeyU64);           \
  } while (0)
  if (k == 1) LAUNCH(1);
  else if (k <= 4) LAUNCH(4);
  else if (k <= 8) LAUNCHQ(8);
  else if (k <= 16) LAUNCHQ(16);
  else if (k <= 32) LAUNCHQ(32);
  else LAUNCHQ(64);
#undef LAUNCH
#undef LAUNCHQ
#undef LAUNCH_Q
#undef LAUNCH_L
#undef LAUNCH_KP
  M3D_CHECK_LAUNCH();
  return M3D_OK;
}

// ---- staged query (see knn_stage_kernel) -------------------------------------------------------------------
// schedule: rings of stage 0, then up to KNNS_MAX_STAGES - 1 group stages "G:last_ring" (the final one "G": until
// closed).  M3D_KNN_STAGES overrides it for A/B runs, e.g. "2,4:3,8:4,16" (the default) or "1,2:2,4:3,16".
struct KnnSchedule {
  int nstage;
  int g[KNNS_MAX_STAGES];
  int r_last[KNNS_MAX_STAGES];
};
static KnnSchedule knn_schedule() {
  KnnSchedule sc;
  const char* e = getenv("M3D_KNN_STAGES");
  const char* txt = e && e[0] ? e : "2,4:3,8:4,16";
  sc.nstage = 0;
  const char* c = txt;
  while (*c && sc.nstage < KNNS_MAX_STAGES) {
    char* end;
    long a = strtol(c, &end, 10);
    int s = sc.nstage;
    if (s == 0) { sc.g[0] = 1; sc.r_last[0] = (int)a; }
    else {
      sc.g[s] = (int)a; sc.r_last[s] = -1;
      if (*end == ':') { sc.r_last[s] = (int)strtol(end + 1, &end, 10); }
    }
    ++sc.nstage;
    c = end;
    if (*c == ',') ++c; else break;
  }
  bool ok = sc.nstage >= 1;
  for (int s = 1; s < sc.nstage && ok; ++s) {
    const int g = sc.g[s];
    ok = (g == 2 || g == 4 || g == 8 || g == 16) && (sc.r_last[s] < 0 || sc.r_last[s] > sc.r_last[s - 1]);
    if (sc.r_last[s] < 0 && s != sc.nstage - 1) ok = false;
  }
  if (!ok || sc.r_last[0] < 0) {  // malformed: the default
    sc.nstage = 4;
    sc.g[0] = 1; sc.r_last[0] = 2; sc.g[1] = 4; sc.r_last[1] = 3; sc.g[2] = 8; sc.r_last[2] = 4; sc.g[3] = 16; sc.r_last[3] = -1;
  }
  sc.r_last[sc.nstage - 1] = -1;  // the final stage always runs until every query is closed
  return sc;
}

static inline size_t knns_buf_bytes(int64_t cap, int kmax) {
  return (size_t)m3d_align(cap * 16, 256) + (size_t)m3d_align(cap * 4, 256) + (size_t)m3d_align(cap * 8 * (int64_t)kmax, 256);
}
static inline int knns_kmax(int k) { return k <= 16 ? 16 : 32; }

extern "C" size_t m3d_knn_staged_workspace_bytes(int64_t n_qry, int32_t k) {
  if (n_qry <= 0 || k < 5 || k > 32) return 0;
  return 256 + 2 * knns_buf_bytes(n_qry, knns_kmax(k));
}

// 1 when the host side should route a cell-sorted query with 4 < k <= 32 through m3d_knn_query_staged: only with
// M3D_KNN_STAGED=1 (f64 keys)
extern "C" int m3d_knn_staged_supported(int64_t n_qry, int32_t k) {
  const char* e = getenv("M3D_KNN_STAGED");  // (read at every call: tests and A/B runs flip it inside one process)
  const int env = e ? atoi(e) : -1;
  static const bool f64_keys = knn_f64_keys();
  if (env == 0 || !f64_keys || k < 5 || k > 32 || n_qry >= (1ll << 31)) return 0;
  // opt-in: measured SLOWER than the single launch on the Lidar-HD-shaped tiles (level 1 of BASELINE config 2: 203-240 us
  // against 191 us; profiles/r03_knn_staged.log with the ring histogram that shows why: 97 % of the queries close by ring
  // 3, so there is no long tail to cut, and a group stage costs what the idle lanes it removes cost)
  return env > 0 ? 1 : 0;
}

__global__ void knns_zero_kernel(unsigned* __restrict__ cnt) { cnt[threadIdx.x] = 0u; }

template <int KMAX, int G, bool FIRST>
static void knns_launch(bool last, unsigned grid, hipStream_t st, const KnnWs& w, const int64_t* ptr_src, int B,
                        const float4* qs, const int64_t* ptr_qry, int64_t n_qry, int k, int* idx_out, float* d2_out,
                        int flags, const KnnStageArgs& sa) {
  if (last)
    hipLaunchKernelGGL((knn_stage_kernel<KMAX, KeyF64, KNNQ_DEPTH, G, FIRST, true>), dim3(grid), dim3(64), 0, st, w,
                       ptr_src, B, qs, ptr_qry, n_qry, k, idx_out, d2_out, flags, sa);
  else
    hipLaunchKernelGGL((knn_stage_kernel<KMAX, KeyF64, KNNQ_DEPTH, G, FIRST, false>), dim3(grid), dim3(64), 0, st, w,
                       ptr_src, B, qs, ptr_qry, n_qry, k, idx_out, d2_out, flags, sa);
}

template <int KMAX>
static int knns_run(const KnnWs& w, const int64_t* ptr_src, int B, const float4* qs, const int64_t* ptr_qry,
                    int64_t n_qry, int k, int* idx_out, float* d2_out, int flags, void* scratch, hipStream_t st) {
  const KnnSchedule sc = knn_schedule();
  char* p = (char*)scratch;
  unsigned* cnt = (unsigned*)p;
  p += 256;
  KnnPoolBuf buf[2];
  for (int i = 0; i < 2; ++i) {
    buf[i].q = (float4*)p; p += m3d_align(n_qry * 16, 256);
    buf[i].cloud = (int*)p; p += m3d_align(n_qry * 4, 256);
    buf[i].best = (void*)p; p += m3d_align(n_qry * 8 * (int64_t)KMAX, 256);
  }
  // (a kernel, not hipMemsetAsync: memset nodes of a captured hipGraph proved unreliable on this stack — csrc/lfa.hip,
  // zero_f64_kernel — and a counter that is not reset sends the next replay's pool writes out of bounds)
  if (sc.nstage > 1) hipLaunchKernelGGL(knns_zero_kernel, dim3(1), dim3(64), 0, st, cnt);
  // later stages: grid-stride over the pool, whose size only the device knows — enough workgroups to fill the chip at
  // the worst case, cheap to launch when the pool turns out small
  static const int stage_grid = getenv("M3D_KNN_STAGE_GRID") ? atoi(getenv("M3D_KNN_STAGE_GRID")) : 4096;
  for (int s = 0; s < sc.nstage; ++s) {
    KnnStageArgs sa;
    sa.in = buf[(s + 1) & 1];
    sa.out = buf[s & 1];
    sa.cnt_in = s > 0 ? cnt + (s - 1) : nullptr;
    sa.cnt_out = cnt + s;
    sa.cap = n_qry;
    sa.r_first = s > 0 ? sc.r_last[s - 1] + 1 : 0;
    sa.r_last = sc.r_last[s];
    const bool last = s == sc.nstage - 1;
    if (s == 0) {
      knns_launch<KMAX, 1, true>(last, (unsigned)m3d_cdiv(n_qry, 64), st, w, ptr_src, B, qs, ptr_qry, n_qry, k, idx_out,
                                 d2_out, flags, sa);
    } else {
      const int64_t worst = m3d_cdiv(n_qry, 64 / sc.g[s]);
      const unsigned grid = (unsigned)(worst < stage_grid ? worst : stage_grid);
      switch (sc.g[s]) {
        case 2: knns_launch<KMAX, 2, false>(last, grid, st, w, ptr_src, B, qs, ptr_qry, n_qry, k, idx_out, d2_out, flags, sa); break;
        case 4: knns_launch<KMAX, 4, false>(last, grid, st, w, ptr_src, B, qs, ptr_qry, n_qry, k, idx_out, d2_out, flags, sa); break;
        case 8: knns_launch<KMAX, 8, false>(last, grid, st, w, ptr_src, B, qs, ptr_qry, n_qry, k, idx_out, d2_out, flags, sa); break;
        default: knns_launch<KMAX, 16, false>(last, grid, st, w, ptr_src, B, qs, ptr_qry, n_qry, k, idx_out, d2_out, flags, sa); break;
      }
    }
    if (hipGetLastError() != hipSuccess) return M3D_ERR_LAUNCH;
  }
  return M3D_OK;
}

// m3d_knn_query for cell-sorted queries (qry_ws) through the staged kernels; ``scratch``: m3d_knn_staged_workspace_bytes
// bytes of device memory (the pools of open queries).  Bit-identical tables.
extern "C" int m3d_knn_query_staged(const void* ws, const int64_t* ptr_src, int64_t n_src, int32_t num_clouds,
                                    const void* qry_ws, const int64_t* ptr_qry, int64_t n_qry, int32_t k,
                                    int32_t sorted_io, int32_t* idx_out, float* d2_out, void* scratch, void* stream) {
  if (!ws || !ptr_src || !ptr_qry || !idx_out || !qry_ws || num_clouds < 0 || n_qry < 0) return M3D_ERR_INVALID;
  if (k < 5 || k > 32 || n_qry >= (1ll << 31)) return M3D_ERR_UNSUPPORTED;
  if (n_qry == 0 || num_clouds == 0) return M3D_OK;
  if (!scratch) return M3D_ERR_INVALID;
  KnnWs w = ws_carve((void*)ws, num_clouds, n_src);
  const float4* qs = ws_carve((void*)qry_ws, num_clouds, n_qry).sorted;
  const int qflags = (sorted_io ? 1 : 0) | (((((uintptr_t)idx_out) | ((uintptr_t)d2_out)) & 15) == 0 ? 2 : 0);
  hipStream_t st = (hipStream_t)stream;
  if (k <= 16) return knns_run<16>(w, ptr_src, num_clouds, qs, ptr_qry, n_qry, k, idx_out, d2_out, qflags, scratch, st);
  return knns_run<32>(w, ptr_src, num_clouds, qs, ptr_qry, n_qry, k, idx_out, d2_out, qflags, scratch, st);
}

// the four K-NN tables of a forward pass (or its four 1-NN tables) in one launch: see KnnBatch
extern "C" int m3d_knn_query_batch(int32_t njobs, const void* const* ws, const int64_t* const* ptr_src,
                                   const int64_t* n_src, const void* const* qry_ws, const int64_t* const* ptr_qry,
                                   const int64_t* n_qry, int32_t num_clouds, int32_t k, int32_t sorted_io,
                                   int32_t* const* idx_out, void* stream) {
  if (njobs < 0 || njobs > KNN_BATCH_MAX) return M3D_ERR_UNSUPPORTED;
  if (njobs == 0 || num_clouds == 0) return M3D_OK;
  if (!ws || !ptr_src || !n_src || !qry_ws || !ptr_qry || !n_qry || !idx_out || num_clouds < 0) return M3D_ERR_INVALID;
  if (k < 1 || k > 64) return M3D_ERR_UNSUPPORTED;
  static const bool f64_keys = knn_f64_keys();
  // same kernel choice for every job: deferred insertion when the largest job is big enough (see m3d_knn_query)
  int64_t nmax = 0;
  bool al = true;
  for (int j = 0; j < njobs; ++j) {
    if (n_qry[j] < 0 || n_src[j] < 0) return M3D_ERR_INVALID;
    if (n_qry[j] > 0 && (!ws[j] || !qry_ws[j] || !ptr_src[j] || !ptr_qry[j] || !idx_out[j])) return M3D_ERR_INVALID;
    nmax = n_qry[j] > nmax ? n_qry[j] : nmax;
    al = al && ((((uintptr_t)idx_out[j]) & 15) == 0);
  }
  static const int queue_env = getenv("M3D_KNN_QUEUE") ? atoi(getenv("M3D_KNN_QUEUE")) : -1;
  const bool use_queue = k > 4 && (queue_env < 0 ? nmax * (int64_t)k >= (1 << 20) : queue_env != 0);
  const int per_wg = use_queue ? 64 * KNNQ_WAVES : 256;
  KnnBatch a;
  a.njobs = njobs;
  unsigned total = 0;
  for (int j = 0; j < KNN_BATCH_MAX; ++j) {
    a.wg_start[j] = total;
    if (j < njobs) {
      a.w[j] = ws_carve((void*)ws[j], num_clouds, n_src[j]);
      a.ptr_src[j] = ptr_src[j];
      a.qsorted[j] = ws_carve((void*)qry_ws[j], num_clouds, n_qry[j]).sorted;
      a.ptr_qry[j] = ptr_qry[j];
      a.n_qry[j] = n_qry[j];
      a.idx_out[j] = idx_out[j];
      total += (unsigned)m3d_cdiv(n_qry[j], per_wg);
    } else {
      a.w[j] = a.w[0]; a.ptr_src[j] = nullptr; a.qsorted[j] = nullptr; a.ptr_qry[j] = nullptr; a.n_qry[j] = 0;
      a.idx_out[j] = nullptr;
    }
  }
  a.wg_start[KNN_BATCH_MAX] = total;
  for (int j = njobs; j < KNN_BATCH_MAX; ++j) a.wg_start[j] = total;
  if (total == 0) return M3D_OK;
  hipStream_t st = (hipStream_t)stream;
  const int qflags = (sorted_io ? 1 : 0) | (al ? 2 : 0);
#define LAUNCH_BD(KM, KP) \
  hipLaunchKernelGGL((knn_query_batch_kernel<KM, KP>), dim3(total), dim3(256), 0, st, a, num_clouds, k, sorted_io)
#define LAUNCH_BQ(KM, KP) \
  hipLaunchKernelGGL((knn_query_queue_batch_kernel<KM, KP, KNNQ_DEPTH>), dim3(total), dim3(64 * KNNQ_WAVES), 0, st, a, num_clouds, k, qflags)
#define LAUNCH_B(KM)                                            \
  do {                                                          \
    if (use_queue) { if (f64_keys) LAUNCH_BQ(KM, KeyF64); else LAUNCH_BQ(KM, KeyU64); } \
    else { if (f64_keys) LAUNCH_BD(KM, KeyF64); else LAUNCH_BD(KM, KeyU64); }           \
  } while (0)
  if (k == 1) LAUNCH_B(1);
  else if (k <= 4) LAUNCH_B(4);
  else if (k <= 8) LAUNCH_B(8);
  else if (k <= 16) LAUNCH_B(16);
  else if (k <= 32) LAUNCH_B(32);
  else LAUNCH_B(64);
#undef LAUNCH_B
#undef LAUNCH_BQ
#undef LAUNCH_BD
  M3D_CHECK_LAUNCH();
  return M3D_OK;
}
