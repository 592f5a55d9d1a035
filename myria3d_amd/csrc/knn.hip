// Batched exact k-nearest-neighbour search for gfx950.
//
// Replaces torch_cluster.knn as reached from knn_graph(pos, K, batch, loop=True)
// (/root/reference/myria3d/models/modules/pyg_randla_net.py:180) and from knn_interpolate
// (pyg_randla_net.py:250, myria3d/models/model.py:90).
//
// Design (MI355X-first, not the upstream 1-thread-per-query brute force): every cloud gets a uniform xy grid of vertical
// columns (cell size chosen for ~M3D_KNN_CELL_TARGET points per column) built by ONE workgroup with an LDS histogram +
// scan + scatter; sources are stored cell-sorted as float4 (x, y, z, original index) so that a wavefront of consecutive
// (cell-sorted) queries walks the same few cache lines.  Each lane keeps its top-k as sorted 64-bit keys (fp32 bits of
// d2 << 32 | index) held as IEEE doubles: one v_min_f64 / v_max_f64 pair is a compare-exchange in the total order
// (d2, index), so results are deterministic and bit-comparable with the CPU oracle.  Rings of cells are visited until the
// k-th distance is inside the explored block (exact).  Distances are (dx*dx + dy*dy) + dz*dz with contraction disabled.
//
// Round 4: ONE search kernel per launch size instead of the five variants of rounds 2-3 (cooperative, staged, LDS-window
// and pipelined walks were all bit-identical and within +-10 % of this one: DESIGN.md section 5; they live in the git
// history).  What the SQ counters of round 3 said — the level-1 query is VALU-issue bound — is attacked where the
// instructions are:
//   * circular rings: a ring's runs are trimmed to the disc of the lane's current k-th distance (rows cut to the chord,
//     side cells outside the disc skipped): -34 % candidates per query, -25 % lock-step candidate slots per wavefront
//     on the Lidar-HD-shaped tiles (tools/sim/knn_trim_sim.py);
//   (a sorting-network drain of the candidate queues — 60 + 16 + 32 compare-exchanges for up to 16 queued keys instead of a
//   16-deep insertion chain per key — was built, bit-identical, and measured 6-7 % SLOWER: profiles/r04b_knn_ab.log; removed.)
// No run-time knobs: tuning constants are compile-time macros (tools/build_variant.sh NAME knn.hip -D...).
#include "m3d_common.h"
#include "../../include/m3d_hip.h"

#define GMAX 64
#define CELLS_MAX (GMAX * GMAX)
#define GP_STRIDE 8  // per-cloud grid record, 8 x 4 bytes
#ifndef M3D_KNN_CELL_TARGET
#define M3D_KNN_CELL_TARGET 7.0f  // points per grid column the cell size aims at (any positive value: same exact result)
#endif
#ifndef KNN_UNROLL
#define KNN_UNROLL 4
#endif
#ifndef KNN_MIN_BLOCKS
#define KNN_MIN_BLOCKS 1
#endif
// deferred-insertion query kernel (k > 4, large query sets): depth of the per-lane LDS candidate queue, register cap
// (waves per SIMD), candidates per batch
#define KNNQ_DEPTH 16
#ifndef KNNQ_MINW
#define KNNQ_MINW 3  // 168 VGPRs: spill-free with the trimming arithmetic (a 128-register cap spills inside the candidate loop: 193 vs 138 us)
#endif
#ifndef KNNQ_UNROLL
#define KNNQ_UNROLL 4
#endif
#ifndef KNNQ_TRIM
#define KNNQ_TRIM 1  // circular rings (0: square rings, the round-3 walk; same results)
#endif
// query sets with at least this many (query, neighbour) pairs take the deferred-insertion kernel (level 1 of BASELINE
// config 2, the K = 32 tiles); the small deep-level launches (a few wavefronts per CU: latency-bound) the direct one
#ifndef KNNQ_MIN_PAIRS
#define KNNQ_MIN_PAIRS (1 << 20)
#endif

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
// The running top-k of a lane is a sorted register array of 64-bit keys (d2 bits, original row) read as IEEE doubles.
// For sign bit 0 the order of doubles IS the order of their bit patterns, so a compare-exchange is v_min_f64 + v_max_f64.
// The high word is biased by one double-exponent step (0x00100000): every finite / inf / canonical-NaN fp32 d2 (bit
// patterns up to 0x7FDFFFFF; tests/test_host.py pins the mapping) then maps to a NORMAL finite double (no denormal or
// NaN operand ever reaches min / max, so the bits pass through unchanged), and +inf (0x7FF00000'00000000) is the
// "empty slot" sentinel, above every key.
struct KeyF64 {
  typedef double T;
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
  // (raw instructions: fmin() / fmax() would add a canonicalising v_max_f64 per operand in IEEE mode)
  // sorted insertion: the key sinks through the list, every slot keeps the smaller of (slot, carried key)
  template <int KMAX>
  static __device__ __forceinline__ void chain(T (&best)[KMAX], T key) {
#pragma unroll
    for (int j = 0; j < KMAX; ++j) {
      T hi;
      asm("v_max_f64 %0, %1, %2" : "=&v"(hi) : "v"(best[j]), "v"(key));
      asm("v_min_f64 %0, %0, %1" : "+v"(best[j]) : "v"(key));  // in place: no register copies at the join
      key = hi;
    }
  }
  template <int KMAX>
  static __device__ __forceinline__ void insert(T (&best)[KMAX], T key) {
    if (key < best[KMAX - 1]) chain<KMAX>(best, key);
  }
};
typedef KeyF64 KP;
typedef KeyF64::T KT;

__device__ __forceinline__ float dist2_exact(float qx, float qy, float qz, float4 s) {
#pragma clang fp contract(off)
  float dx = s.x - qx, dy = s.y - qy, dz = s.z - qz;
  float a = dx * dx;
  float b = dy * dy;
  float c = dz * dz;
  float ab = a + b;
  return ab + c;
}

// direct insertion (small k, small launches): candidates p0 .. p1-1 of the cell-sorted array.  Unclamped addresses: one
// base per trip + immediate offsets (reads run up to 3*KNN_UNROLL-1 records past p1: inside the 256-byte-padded workspace,
// masked in examine()) — per-index clamps cost a v_min and an address computation per load in kernels whose bound is
// instruction issue.  Lists of 8+ keys (the insertion chain is long enough to cover a load): two register sets used in
// turn, the next trip's loads in flight while this trip's candidates are inserted; the 1-NN and 4-NN queries just load and
// use.  `lower`: keys <= lower are not admitted (second pass of a k > 64 query; 0.0 = no bound, every key is > 0).
template <int KMAX>
__device__ __forceinline__ void scan_range(KT (&best)[KMAX], const float4* __restrict__ sorted, int p0, int p1, float qx,
                                           float qy, float qz, KT lower) {
  if (p1 <= p0) return;
  const int last = p1 - 1;
  auto examine = [&](const float4 (&s)[KNN_UNROLL], int p) {
#pragma unroll
    for (int u = 0; u < KNN_UNROLL; ++u) {
      float d2 = dist2_exact(qx, qy, qz, s[u]);
      KT key = KP::make(d2, __float_as_int(s[u].w));
      if (p + u > last || !(key > lower)) key = KP::empty();
      KP::insert<KMAX>(best, key);
    }
  };
  if constexpr (KMAX >= 8) {
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
}

// what a query lane needs to know about its cloud's grid
struct GridView {
  float gx0, gy0, inv_h, h, eps;
  int Gx, Gy, n;
  const int* cs;
  const float4* sorted;
};
__device__ __forceinline__ int cloud_of(const int64_t* __restrict__ ptr_qry, int B, int64_t t) {
  int lo = 0, hi = B;  // largest b with ptr_qry[b] <= t
  while (hi - lo > 1) {
    int mid = (lo + hi) >> 1;
    if (ptr_qry[mid] <= t) lo = mid; else hi = mid;
  }
  return lo;
}
__device__ __forceinline__ GridView grid_of(const KnnWs& w, const int64_t* __restrict__ ptr_src, int b) {
  const float* gp = w.gridp + (size_t)b * GP_STRIDE;
  GridView g;
  g.gx0 = gp[0]; g.gy0 = gp[1]; g.inv_h = gp[2]; g.h = gp[3]; g.eps = gp[4];
  g.Gx = ((const int*)gp)[5]; g.Gy = ((const int*)gp)[6]; g.n = ((const int*)gp)[7];
  g.cs = w.cell_start + (size_t)b * (CELLS_MAX + 1);
  g.sorted = w.sorted + ptr_src[b];
  return g;
}
// squared distance below which the block of cells (cx +- R, cy +- R) is known to hold every neighbour; < 0: the block
// covers the grid (search complete)
__device__ __forceinline__ float ring_bound2(const GridView& g, int cx, int cy, int R, float qx, float qy) {
  const bool covers = (cx - R <= 0) && (cx + R >= g.Gx - 1) && (cy - R <= 0) && (cy + R >= g.Gy - 1);
  if (covers) return -1.f;
  float bound = 3.4e38f;
  if (cx - R > 0) bound = fminf(bound, qx - (g.gx0 + (float)(cx - R) * g.h));
  if (cx + R < g.Gx - 1) bound = fminf(bound, (g.gx0 + (float)(cx + R + 1) * g.h) - qx);
  if (cy - R > 0) bound = fminf(bound, qy - (g.gy0 + (float)(cy - R) * g.h));
  if (cy + R < g.Gy - 1) bound = fminf(bound, (g.gy0 + (float)(cy + R + 1) * g.h) - qy);
  bound = fmaxf(bound - g.eps, 0.f);
  return bound * bound;
}

// qmode 0: queries are pos_qry rows (row index = output row)
// qmode 1: queries are the float4 records of qsorted (output row = record.w, or the slot itself with sorted_io)
// ostride / ooff: row stride and first column of this pass in idx_out / d2_out; lower_col >= 0: second pass of a k > 64
// query — only keys above the key of the neighbour stored in column lower_col (the last one of the first pass) are
// admitted, which makes the pass the next-best neighbours in the same total order
template <int KMAX>
__device__ __forceinline__ void knn_query_direct_body(const KnnWs& w, const int64_t* __restrict__ ptr_src, int B,
                                                      const float* __restrict__ pos_qry, int qstride,
                                                      const float4* __restrict__ qsorted,
                                                      const int64_t* __restrict__ ptr_qry, int64_t n_qry, int k,
                                                      int* __restrict__ idx_out, float* __restrict__ d2_out,
                                                      int sorted_io, int ostride, int ooff, int lower_col, int64_t t) {
  if (t >= n_qry) return;
  const int b = cloud_of(ptr_qry, B, t);
  float qx, qy, qz;
  int64_t orow;
  if (qsorted) {
    float4 q = qsorted[t];
    qx = q.x; qy = q.y; qz = q.z; orow = sorted_io ? t : (int64_t)__float_as_int(q.w);
  } else {
    const float* p = pos_qry + t * qstride;
    qx = p[0]; qy = p[1]; qz = p[2]; orow = t;
  }
  const GridView g = grid_of(w, ptr_src, b);
  int* io = idx_out + orow * ostride + ooff;
  float* dq = d2_out ? d2_out + orow * ostride + ooff : nullptr;

  KT best[KMAX];
#pragma unroll
  for (int j = 0; j < KMAX; ++j) best[j] = KP::empty();
  KT lower = 0.0;
  bool open = g.n > 0;
  if (lower_col >= 0) {
    const int id = idx_out[orow * ostride + lower_col];
    if (id < 0) {
      open = false;  // the first pass already ran out of points
    } else {
      const float4 s = w.sorted[sorted_io ? id : w.inv[id]];
      lower = KP::make(dist2_exact(qx, qy, qz, s), __float_as_int(s.w));
    }
  }
  if (open) {
    const int cx = min(g.Gx - 1, max(0, (int)((qx - g.gx0) * g.inv_h)));
    const int cy = min(g.Gy - 1, max(0, (int)((qy - g.gy0) * g.inv_h)));
    for (int R = 0;; ++R) {
      for (int dy = -R; dy <= R; ++dy) {
        int yy = cy + dy;
        if (yy < 0 || yy >= g.Gy) continue;
        if (dy == -R || dy == R) {
          int x0 = max(cx - R, 0), x1 = min(cx + R, g.Gx - 1);
          scan_range<KMAX>(best, g.sorted, g.cs[yy * g.Gx + x0], g.cs[yy * g.Gx + x1 + 1], qx, qy, qz, lower);
        } else {
          if (cx - R >= 0) scan_range<KMAX>(best, g.sorted, g.cs[yy * g.Gx + cx - R], g.cs[yy * g.Gx + cx - R + 1], qx, qy, qz, lower);
          if (cx + R < g.Gx) scan_range<KMAX>(best, g.sorted, g.cs[yy * g.Gx + cx + R], g.cs[yy * g.Gx + cx + R + 1], qx, qy, qz, lower);
        }
      }
      const float b2 = ring_bound2(g, cx, cy, R, qx, qy);
      if (b2 < 0.f) break;
      KT kb = best[KMAX - 1];
#pragma unroll
      for (int j = 0; j < KMAX - 1; ++j)
        if (j == k - 1) kb = best[j];
      // an unfilled slot reads as a huge value: the search continues
      const float kth = KP::is_empty(kb) ? __builtin_inff() : __uint_as_float(KP::d2bits(kb));
      if (kth <= b2) break;
    }
  }
  // A cloud with at least ooff + k source points fills every slot — unless a position is not finite (NaN distances are never
  // inserted).  Consumers that were promised complete neighbourhoods (M3D_LFA_FULL: the plan's host-side edge count) read the
  // table unmasked, so such a slot names the cloud's first row (a valid row in either numbering) instead of -1: the outputs
  // are NaN like the reference's, not an out-of-bounds access.  The distance stays +inf.
  const int fallback = g.n >= ooff + k ? (int)ptr_src[b] : -1;
#pragma unroll
  for (int j = 0; j < KMAX; ++j) {
    if (j < k) {
      bool ok = !KP::is_empty(best[j]);
      int id = ok ? KP::row(best[j]) : fallback;
      if (sorted_io && ok) id = w.inv[id];  // neighbours selected by (d2, original row); reported as cell-sorted slots
      io[j] = id;
      if (dq) dq[j] = ok ? __uint_as_float(KP::d2bits(best[j])) : __builtin_inff();
    }
  }
}

template <int KMAX>
__global__ __launch_bounds__(256, KNN_MIN_BLOCKS) void knn_query_kernel(KnnWs w, const int64_t* __restrict__ ptr_src, int B,
                                                        const float* __restrict__ pos_qry, int qstride,
                                                        const float4* __restrict__ qsorted,
                                                        const int64_t* __restrict__ ptr_qry, int64_t n_qry, int k,
                                                        int* __restrict__ idx_out, float* __restrict__ d2_out,
                                                        int sorted_io, int ostride, int ooff, int lower_col) {
  // (a BACKGROUND launch — flags bits 8-15 of m3d_knn_query — has fewer workgroups than blocks of 256 queries: each walks several)
  const int64_t nblk = (n_qry + 255) >> 8;
  for (int64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x)
    knn_query_direct_body<KMAX>(w, ptr_src, B, pos_qry, qstride, qsorted, ptr_qry, n_qry, k, idx_out, d2_out, sorted_io,
                                ostride, ooff, lower_col, blk * 256 + threadIdx.x);
}

// Several independent query problems in ONE launch (m3d_knn_query_batch): the K-NN tables of the four resolution levels
// (or the decoder's four 1-NN tables) are separate launches otherwise, and the deep ones are a few wavefronts per CU
// of pure latency that run one after the other while the level-1 launch ends on its slowest wavefronts with most SIMDs
// idle.  Workgroup b belongs to job j with wg_start[j] <= b < wg_start[j + 1]; cell-sorted queries only (qry_ws), no
// distances.
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
__device__ __forceinline__ int knn_batch_job(const KnnBatch& a, unsigned vb) {
  int j = 0;
#pragma unroll
  for (int i = 1; i < KNN_BATCH_MAX; ++i) j += (i < a.njobs && vb >= a.wg_start[i]) ? 1 : 0;
  return j;
}
template <int KMAX>
__global__ __launch_bounds__(256, KNN_MIN_BLOCKS) void knn_query_batch_kernel(KnnBatch a, int B, int k, int sorted_io) {
  // (virtual workgroups: a BACKGROUND launch — flags bits 8-15 of m3d_knn_query_batch — has fewer real ones)
  const unsigned total = a.wg_start[KNN_BATCH_MAX];
  for (unsigned vb = blockIdx.x; vb < total; vb += gridDim.x) {
    const int j = knn_batch_job(a, vb);
    knn_query_direct_body<KMAX>(a.w[j], a.ptr_src[j], B, nullptr, 0, a.qsorted[j], a.ptr_qry[j], a.n_qry[j], k,
                                a.idx_out[j], nullptr, sorted_io, k, 0, -1, (int64_t)(vb - a.wg_start[j]) * 256 + threadIdx.x);
  }
}

// ------------------------------------------------------------------------------------------
// query, deferred insertion (k > 4, large query sets)
//
// Same search (per-lane ring walk over the xy grid, same keys, same total order => bit-identical results), different
// inner loop.  The direct kernel above runs its 2*KMAX-instruction sorted insertion whenever ANY of the 64 lanes
// improves its list — measured ~200 times per wavefront at K = 16 although a lane improves only ~55 times — and
// every candidate slot pays key construction, a 64-bit compare and a divergent branch.  Here a candidate costs a
// distance, one fp32 compare against the lane's current k-th distance and, if it passes, an 8-byte append to a
// per-lane queue in LDS ([slot][lane]: conflict-free).  Queues are drained into the sorted register list together —
// when a queue is about to fill and at the end of every ring, where the termination test needs the exact k-th
// distance — and the threshold tightens after every drain.  One wavefront per workgroup: no barriers, the hardware
// balances 3 200 independent wavefronts over the CUs.
//
// Round 4 (the kernel is VALU-issue bound: r03m SQ counters):
//   * circular rings (KNNQ_TRIM).  kth is EXACT at the start of a ring (the drain at the end of the previous one), and a
//     candidate is admitted only if d2 <= kth: a row of ring cells whose distance in y already exceeds sqrt(kth) is
//     skipped, the others are cut to the chord |dx| <= sqrt(kth - gap_y^2), a side cell is skipped when its nearest corner
//     lies outside the disc.  All comparisons are made conservative (kth inflated by 2^-16, gaps shrunk by the grid's
//     rounding slack eps), so that no point a square ring would have admitted is lost: bit-identical tables.
// ------------------------------------------------------------------------------------------
template <int KMAX>
__device__ __forceinline__ void knn_query_queue_body(
    const KnnWs& w, const int64_t* __restrict__ ptr_src, int B, const float* __restrict__ pos_qry, int qstride,
    const float4* __restrict__ qsorted, const int64_t* __restrict__ ptr_qry, int64_t n_qry, int k,
    int* __restrict__ idx_out, float* __restrict__ d2_out, int flags, int64_t wg_in, int64_t nblk) {
  constexpr int QD = KNNQ_DEPTH;
  const int sorted_io = flags & 1;  // bit 1: idx_out / d2_out are 16-byte aligned (vector stores allowed)
  __shared__ KT queue[QD][64];
  const int lane = threadIdx.x;
  // XCD-aware order: the dispatcher deals consecutive workgroups round-robin over the 8 XCDs (observed; affects speed
  // only), so workgroup b takes the queries of chunk (b % 8): every XCD then walks one contiguous eighth of the
  // (cell-sorted) queries — two whole tiles at BASELINE config 2 — and its private L2 holds just those tiles' records
  int64_t wg = wg_in;
  {
    const int64_t q8 = nblk >> 3, r8 = nblk & 7;
    const int64_t xcd = wg & 7, i8 = wg >> 3;
    wg = xcd * q8 + (xcd < r8 ? xcd : r8) + i8;
  }
  const int64_t t = wg * 64 + lane;
  if (t >= n_qry) return;  // (the drains below are per-lane loops: lanes that leave early are simply inactive)
  const int b = cloud_of(ptr_qry, B, t);
  float qx, qy, qz;
  int64_t orow;
  if (qsorted) {
    float4 q = qsorted[t];
    qx = q.x; qy = q.y; qz = q.z; orow = sorted_io ? t : (int64_t)__float_as_int(q.w);
  } else {
    const float* p = pos_qry + t * qstride;
    qx = p[0]; qy = p[1]; qz = p[2]; orow = t;
  }
  const GridView g = grid_of(w, ptr_src, b);

  KT best[KMAX];
#pragma unroll
  for (int j = 0; j < KMAX; ++j) best[j] = KP::empty();
  int cnt = 0;                       // entries in this lane's queue
  float kth = __builtin_inff();      // this lane's current k-th squared distance (+inf while the list is not full)

  auto drain = [&]() {
    // divergent trip count: the wavefront runs max(cnt) insertion chains, all lanes in step, two keys per trip (their chains
    // are independent up to a one-slot skew, so the scheduler can interleave them)
    for (int i = 0; i < cnt; i += 2) {
      KT key[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) key[u] = queue[(i + u) & (QD - 1)][lane];
#pragma unroll
      for (int u = 0; u < 2; ++u) KP::chain<KMAX>(best, u == 0 || i + u < cnt ? key[u] : KP::empty());
    }
    cnt = 0;
    // k-th key = the largest of the first k (the list is ascending; empty slots sort above every key).  Written as a max
    // over the high words so that a run-time k costs KMAX selects, not a register array spilled to scratch
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
  // candidates p0 .. p1-1.  Two register sets used in turn: the loads of the next KNNQ_UNROLL records are in flight while
  // the current ones are examined (unclamped addresses, ONE base per trip + immediate offsets; reads run up to
  // 3*KNNQ_UNROLL-1 records past p1: still inside the workspace — every array of it is padded to 256 bytes and the
  // sorted array is followed by the perm / inv arrays — and masked out in examine())
  auto scan = [&](int p0, int p1) {
    auto examine = [&](const float4 (&s)[KNNQ_UNROLL], int p) {
#pragma unroll
      for (int u = 0; u < KNNQ_UNROLL; ++u) {
        const float d2 = dist2_exact(qx, qy, qz, s[u]);
        // !(d2 > kth): ties with the current k-th distance go through the exact (d2, row) order in the drain
        if (p + u < p1 && !(d2 > kth)) {
          queue[cnt][lane] = KP::make(d2, __float_as_int(s[u].w));
          ++cnt;
        }
      }
    };
    if (p1 > p0) {
      float4 ra[KNNQ_UNROLL], rb[KNNQ_UNROLL];
#pragma unroll
      for (int u = 0; u < KNNQ_UNROLL; ++u) ra[u] = g.sorted[p0 + u];
      for (int p = p0; p < p1; p += 2 * KNNQ_UNROLL) {
        // (the drain check sits BEFORE the other register set is requested: one set of candidates is live across a drain)
        if (__builtin_amdgcn_ballot_w64(cnt > QD - KNNQ_UNROLL) != 0) drain();
#pragma unroll
        for (int u = 0; u < KNNQ_UNROLL; ++u) rb[u] = g.sorted[p + KNNQ_UNROLL + u];
        examine(ra, p);
        if (__builtin_amdgcn_ballot_w64(cnt > QD - KNNQ_UNROLL) != 0) drain();
#pragma unroll
        for (int u = 0; u < KNNQ_UNROLL; ++u) ra[u] = g.sorted[p + 2 * KNNQ_UNROLL + u];
        examine(rb, p + KNNQ_UNROLL);
      }
    }
  };

  if (g.n > 0) {
    const int cx = min(g.Gx - 1, max(0, (int)((qx - g.gx0) * g.inv_h)));
    const int cy = min(g.Gy - 1, max(0, (int)((qy - g.gy0) * g.inv_h)));
    for (int R = 0;; ++R) {
      // the disc of this ring: every admitted candidate has d2 <= kth (exact here); inflated so that fp32 rounding of a
      // candidate's d2 cannot make a trimmed point admissible
      const float k2 = KNNQ_TRIM ? kth * 1.0000153f : __builtin_inff();
      for (int dy = -R; dy <= R; ++dy) {
        const int yy = cy + dy;
        if (yy < 0 || yy >= g.Gy) continue;
        float rem = k2;
        if (KNNQ_TRIM && dy != 0) {
          const float edge = dy > 0 ? (g.gy0 + (float)yy * g.h) - qy : qy - (g.gy0 + (float)(yy + 1) * g.h);
          const float gap = fmaxf(edge - g.eps, 0.f);
          rem = k2 - gap * gap;
          if (rem < 0.f) continue;
        }
        const bool edge_row = (dy == -R || dy == R);
        for (int sg = 0; sg < (edge_row ? 1 : 2); ++sg) {  // (ONE scan site: the candidate loop is inlined once)
          int xa, xb;
          if (edge_row) {
            xa = max(cx - R, 0); xb = min(cx + R, g.Gx - 1);
            if (KNNQ_TRIM && rem < 3.0e38f) {
              // chord |dx| <= sqrt(rem); cell indices with the build kernel's own expression (monotonic in x)
              const float xr = __builtin_sqrtf(rem) * 1.000001f + g.eps;
              const float fa = fminf(fmaxf((qx - xr - g.gx0) * g.inv_h, 0.f), 65535.f);
              const float fb = fminf(fmaxf((qx + xr - g.gx0) * g.inv_h, 0.f), 65535.f);
              xa = max(xa, (int)fa);
              xb = min(xb, (int)fb);
            }
          } else {
            xa = xb = sg == 0 ? cx - R : cx + R;
            if (xa < 0 || xa >= g.Gx) continue;
            if (KNNQ_TRIM) {
              const float ex = sg == 0 ? qx - (g.gx0 + (float)(xa + 1) * g.h) : (g.gx0 + (float)xa * g.h) - qx;
              const float gx = fmaxf(ex - g.eps, 0.f);
              if (gx * gx > rem) continue;
            }
          }
          if (xa <= xb) scan(g.cs[yy * g.Gx + xa], g.cs[yy * g.Gx + xb + 1]);
        }
      }
      drain();
      const float b2 = ring_bound2(g, cx, cy, R, qx, qy);
      if (b2 < 0.f || kth <= b2) break;  // (kth = +inf while fewer than k neighbours are known: keeps searching)
    }
  }
  // ---- results: all slot translations (w.inv) in flight together, rows stored 16 bytes at a time when k == KMAX
  int ids[KMAX];
#pragma unroll
  for (int j = 0; j < KMAX; ++j) ids[j] = KP::is_empty(best[j]) ? -1 : KP::row(best[j]);
  // (see knn_query_direct_body: an unfilled slot of a cloud with >= k points — non-finite positions — names the cloud's
  // first row, never -1: the mask-free consumers must not read out of bounds)
  const int fallback = g.n >= k ? (int)ptr_src[b] : -1;
  if (sorted_io) {  // neighbours selected by (d2, original row); reported as cell-sorted slots
    int tr[KMAX];
#pragma unroll
    for (int j = 0; j < KMAX; ++j) tr[j] = w.inv[ids[j] < 0 ? 0 : ids[j]];
#pragma unroll
    for (int j = 0; j < KMAX; ++j) ids[j] = ids[j] < 0 ? fallback : tr[j];
  } else {
#pragma unroll
    for (int j = 0; j < KMAX; ++j) ids[j] = ids[j] < 0 ? fallback : ids[j];
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

template <int KMAX>
__global__ __launch_bounds__(64, KNNQ_MINW) void knn_query_queue_kernel(
    KnnWs w, const int64_t* __restrict__ ptr_src, int B, const float* __restrict__ pos_qry, int qstride,
    const float4* __restrict__ qsorted, const int64_t* __restrict__ ptr_qry, int64_t n_qry, int k,
    int* __restrict__ idx_out, float* __restrict__ d2_out, int flags, int64_t nblk) {
  // nblk query groups of 64 on gridDim.x <= nblk one-wave workgroups: with fewer workgroups than groups (flags bits 8-15 of
  // m3d_knn_query: a BACKGROUND launch that shares the chip with another stream's work) every wave walks several groups
  for (int64_t wg = blockIdx.x; wg < nblk; wg += gridDim.x)
    knn_query_queue_body<KMAX>(w, ptr_src, B, pos_qry, qstride, qsorted, ptr_qry, n_qry, k, idx_out, d2_out, flags, wg, nblk);
}

template <int KMAX>
__global__ __launch_bounds__(64, KNNQ_MINW) void knn_query_queue_batch_kernel(KnnBatch a, int B, int k, int flags) {
  const unsigned total = a.wg_start[KNN_BATCH_MAX];
  for (unsigned vb = blockIdx.x; vb < total; vb += gridDim.x) {
    const int j = knn_batch_job(a, vb);
    knn_query_queue_body<KMAX>(a.w[j], a.ptr_src[j], B, nullptr, 0, a.qsorted[j], a.ptr_qry[j], a.n_qry[j], k,
                               a.idx_out[j], nullptr, flags, (int64_t)(vb - a.wg_start[j]),
                               (int64_t)(a.wg_start[j + 1] - a.wg_start[j]));
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
  hipLaunchKernelGGL(knn_build_kernel, dim3(num_clouds), dim3(1024), 0, (hipStream_t)stream, pos_src, pos_stride,
                     ptr_src, w, M3D_KNN_CELL_TARGET, map_in, map_out);
  M3D_CHECK_LAUNCH();
  return M3D_OK;
}

extern "C" int m3d_knn_build(const float* pos_src, int32_t pos_stride, const int64_t* ptr_src, int32_t num_clouds,
                             int64_t n_src, void* ws, void* stream) {
  return m3d_knn_build_map(pos_src, pos_stride, ptr_src, num_clouds, n_src, ws, nullptr, nullptr, stream);
}

// k <= 64: one launch.  64 < k <= 100 (the upstream CUDA kNN asserts k <= 100): two passes of the direct kernel — the 64
// nearest, then the k - 64 next ones in the same total order (keys above the 64th's) — into columns 0..63 / 64..k-1
extern "C" int m3d_knn_query(const void* ws, const int64_t* ptr_src, int64_t n_src, int32_t num_clouds,
                             const float* pos_qry, int32_t qry_stride, const void* qry_ws, const int64_t* ptr_qry,
                             int64_t n_qry, int32_t k, int32_t flags, int32_t* idx_out, float* d2_out,
                             void* stream) {
  // flags: bit 0 = sorted_io; bits 1-2 = kernel choice: 0 by size, 1 deferred insertion (4 < k <= 64), 2 direct insertion
  const int sorted_io = flags & 1, choice = (flags >> 1) & 3;
  if (!ws || !ptr_src || !ptr_qry || !idx_out || num_clouds < 0 || n_qry < 0) return M3D_ERR_INVALID;
  if (k < 1 || choice == 3) return M3D_ERR_INVALID;
  if (k > 100) return M3D_ERR_UNSUPPORTED;
  if (!pos_qry && !qry_ws && n_qry > 0) return M3D_ERR_INVALID;
  if (pos_qry && qry_stride < 3) return M3D_ERR_INVALID;
  if (sorted_io && !qry_ws) return M3D_ERR_INVALID;  // sorted rows are defined by the query workspace
  if (n_qry == 0 || num_clouds == 0) return M3D_OK;
  KnnWs w = ws_carve((void*)ws, num_clouds, n_src);
  const float4* qs = qry_ws ? ws_carve((void*)qry_ws, num_clouds, n_qry).sorted : nullptr;
  // flags bits 8-15: BACKGROUND launch — at most that many x 64 wavefronts (0: as many as the queries fill).  The tables are
  // the same; a launch that shares the chip with another stream's work (HipRandLANet.prefetch_geometry beside a training
  // step) then occupies one wave slot per SIMD instead of flooding every CU for its whole duration (round 6: the level-1
  // self-kNN capped at 1 024 waves costs the step 0.05 ms less, profiles/r06i_*)
  const int64_t wcap = (int64_t)((flags >> 8) & 0xff) * 64;
  const int64_t dblk = m3d_cdiv(n_qry, 256);
  dim3 grid((unsigned)((wcap > 0 && wcap / 4 < dblk) ? (wcap / 4 > 0 ? wcap / 4 : 1) : dblk)), block(256);
  hipStream_t st = (hipStream_t)stream;
  const int qflags = (sorted_io ? 1 : 0) | (((((uintptr_t)idx_out) | ((uintptr_t)d2_out)) & 15) == 0 ? 2 : 0);
  const bool use_queue = choice == 0 ? n_qry * (int64_t)k >= KNNQ_MIN_PAIRS : choice == 1;
  const int64_t qblk = m3d_cdiv(n_qry, 64);
  const int64_t qgrid = (wcap > 0 && wcap < qblk) ? wcap : qblk;
#define LAUNCH(KM, KK, OOFF, LOWER)                                                                              \
  hipLaunchKernelGGL((knn_query_kernel<KM>), grid, block, 0, st, w, ptr_src, num_clouds, pos_qry, qry_stride, qs, \
                     ptr_qry, n_qry, KK, idx_out, d2_out, sorted_io, k, OOFF, LOWER)
#define LAUNCHQ(KM)                                                                                               \
  do {                                                                                                            \
    if (use_queue)                                                                                                \
      hipLaunchKernelGGL((knn_query_queue_kernel<KM>), dim3((unsigned)qgrid), dim3(64), 0, st, w,                 \
                         ptr_src, num_clouds, pos_qry, qry_stride, qs, ptr_qry, n_qry, k, idx_out, d2_out, qflags, qblk); \
    else LAUNCH(KM, k, 0, -1);                                                                                    \
  } while (0)
  if (k == 1) LAUNCH(1, k, 0, -1);
  else if (k <= 4) LAUNCH(4, k, 0, -1);
  else if (k <= 8) LAUNCHQ(8);
  else if (k <= 16) LAUNCHQ(16);
  else if (k <= 32) LAUNCHQ(32);
  else if (k <= 64) LAUNCHQ(64);
  else {
    LAUNCH(64, 64, 0, -1);
    LAUNCH(64, k - 64, 64, 63);
  }
#undef LAUNCH
#undef LAUNCHQ
  M3D_CHECK_LAUNCH();
  return M3D_OK;
}

// the four K-NN tables of a forward pass (or its four 1-NN tables) in one launch: see KnnBatch
extern "C" int m3d_knn_query_batch(int32_t njobs, const void* const* ws, const int64_t* const* ptr_src,
                                   const int64_t* n_src, const void* const* qry_ws, const int64_t* const* ptr_qry,
                                   const int64_t* n_qry, int32_t num_clouds, int32_t k, int32_t flags,
                                   int32_t* const* idx_out, void* stream) {
  const int sorted_io = flags & 1, choice = (flags >> 1) & 3;  // as m3d_knn_query
  if (choice == 3) return M3D_ERR_INVALID;
  if (njobs < 0 || njobs > KNN_BATCH_MAX) return M3D_ERR_UNSUPPORTED;
  if (njobs == 0 || num_clouds == 0) return M3D_OK;
  if (!ws || !ptr_src || !n_src || !qry_ws || !ptr_qry || !n_qry || !idx_out || num_clouds < 0) return M3D_ERR_INVALID;
  if (k < 1) return M3D_ERR_INVALID;
  if (k > 64) return M3D_ERR_UNSUPPORTED;
  // same kernel choice for every job: deferred insertion when the largest job is big enough (see m3d_knn_query)
  int64_t nmax = 0;
  bool al = true;
  for (int j = 0; j < njobs; ++j) {
    if (n_qry[j] < 0 || n_src[j] < 0) return M3D_ERR_INVALID;
    if (n_qry[j] > 0 && (!ws[j] || !qry_ws[j] || !ptr_src[j] || !ptr_qry[j] || !idx_out[j])) return M3D_ERR_INVALID;
    nmax = n_qry[j] > nmax ? n_qry[j] : nmax;
    al = al && ((((uintptr_t)idx_out[j]) & 15) == 0);
  }
  const bool use_queue = k > 4 && (choice == 0 ? nmax * (int64_t)k >= KNNQ_MIN_PAIRS : choice == 1);
  const int per_wg = use_queue ? 64 : 256;
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
  // flags bits 8-15: BACKGROUND launch, at most that many x 64 wavefronts (as m3d_knn_query)
  const unsigned wcap = (unsigned)((flags >> 8) & 0xff) * 64u;
  const unsigned cap_wg = use_queue ? wcap : (wcap / 4 > 0 ? wcap / 4 : (wcap ? 1u : 0u));
  const unsigned grid_b = (cap_wg > 0 && cap_wg < total) ? cap_wg : total;
#define LAUNCH_B(KM)                                                                                                    \
  do {                                                                                                                  \
    if (use_queue) hipLaunchKernelGGL((knn_query_queue_batch_kernel<KM>), dim3(grid_b), dim3(64), 0, st, a, num_clouds, k, qflags); \
    else hipLaunchKernelGGL((knn_query_batch_kernel<KM>), dim3(grid_b), dim3(256), 0, st, a, num_clouds, k, sorted_io);  \
  } while (0)
  if (k == 1) LAUNCH_B(1);
  else if (k <= 4) LAUNCH_B(4);
  else if (k <= 8) LAUNCH_B(8);
  else if (k <= 16) LAUNCH_B(16);
  else if (k <= 32) LAUNCH_B(32);
  else LAUNCH_B(64);
#undef LAUNCH_B
  M3D_CHECK_LAUNCH();
  return M3D_OK;
}
