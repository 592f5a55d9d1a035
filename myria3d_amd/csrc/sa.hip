// PointNet++ set-abstraction variant (BASELINE.json configs[4]: "Dense tiles 40 000 pts K=32 + PointNet++ set-abstraction
// variant"; north_star: "random/FPS subsampling") for gfx950.
//
// There is NO reference implementation of this variant: /root/reference/myria3d/models/model.py:12 holds
// MODEL_ZOO = [PyGRandLANet] only and the repository has no farthest-point sampling anywhere.  The kernels restate the
// published operators the variant is made of (Qi et al., PointNet++, 2017, as packaged by PyG):
//   * m3d_fps            torch_cluster.fps(random_start=False): iterative farthest-point sampling inside each cloud
//   * m3d_sa_group       the gathers of PointNetConv.message: [x_j, pos_j - pos_i] per edge (compact edge list:
//                        a cloud with fewer than K points simply has fewer edges, like PyG's edge_index)
//   * m3d_seg_max(+bwd)  aggr="max" over the edges of a centre (torch_scatter.scatter_max: the gradient goes to the
//                        arg-max edge, first edge on ties)
//   * m3d_sa_group_bwd   transpose of the x_j gather (scatter-add)
// oracle/pointnet2_oracle.py is the checker (parity unpinned: nothing in the reference to pin it to).
#include <limits.h>
#include "m3d_common.h"
#include "../../include/m3d_hip.h"

// ------------------------------------------------------------------------------------------
// farthest-point sampling: one 1024-thread workgroup per cloud; thread t owns points t, t + 1024, ...; their running
// minimum distance to the selected set lives in registers for the whole loop.  Positions: registers when the cloud has
// <= 16 points per thread (CACHE), otherwise re-read every iteration (a tile is 640 KB at most: L2-resident).
// An iteration = distance update + arg-max: wave reduction by DPP / row swaps, 16 wave candidates (value, index, coordinates)
// through LDS (double-buffered: ONE barrier per iteration), every thread reduces the 16 candidates itself (broadcast
// reads) and takes the winner's coordinates from LDS: no global memory access on the loop-carried path.
// Arithmetic: d2 = (dx*dx + dy*dy) + dz*dz with separately rounded products (no FMA contraction), the order of the
// oracle; arg-max ties -> the smaller point index.  Bit-exact index lists follow.
// ------------------------------------------------------------------------------------------
#define FPS_THREADS 1024

__device__ __forceinline__ float fps_d2(const float4& p, const float4& q) {
#pragma clang fp contract(off)
  const float dx = p.x - q.x, dy = p.y - q.y, dz = p.z - q.z;
  const float a = dx * dx;
  const float b = dy * dy;
  const float c = dz * dz;
  const float ab = a + b;
  return ab + c;
}

// one butterfly step of the arg-max reduction over a total order (larger distance, then smaller index): DPP lane
// permutations inside a row of 16 (VALU moves; __shfl_xor would be an LDS-pipe round trip per step)
template <int CTRL>
__device__ __forceinline__ void fps_step_dpp(float& best, int& besti) {
  const float ob = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(best), CTRL, 0xF, 0xF, false));
  const int oi = __builtin_amdgcn_update_dpp(0, besti, CTRL, 0xF, 0xF, false);
  if (ob > best || (ob == best && oi < besti)) { best = ob; besti = oi; }
}
template <int CTRL>
__device__ __forceinline__ void fps_step_dpp3(float& best, int& besti, int& tag) {
  const float ob = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(best), CTRL, 0xF, 0xF, false));
  const int oi = __builtin_amdgcn_update_dpp(0, besti, CTRL, 0xF, 0xF, false);
  const int ot = __builtin_amdgcn_update_dpp(0, tag, CTRL, 0xF, 0xF, false);
  if (ob > best || (ob == best && oi < besti)) { best = ob; besti = oi; tag = ot; }
}
__device__ __forceinline__ void fps_pick(float a, int ia, float b, int ib, float& best, int& besti) {
  const bool tb = b > a || (b == a && ib < ia);
  best = tb ? b : a;
  besti = tb ? ib : ia;
}

template <int PPT, bool CACHE>
__global__ __launch_bounds__(FPS_THREADS) void fps_kernel(const float4* __restrict__ pos4, const int64_t* __restrict__ ptr_src,
                                                         const int64_t* __restrict__ ptr_out,
                                                         const int32_t* __restrict__ start, int32_t* __restrict__ idx_out) {
  const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wid = t >> 6;
  const int64_t s0 = ptr_src[b], o0 = ptr_out[b];
  const int n = (int)(ptr_src[b + 1] - s0), m = (int)(ptr_out[b + 1] - o0);
  if (m <= 0 || n <= 0) return;
  constexpr int NW = FPS_THREADS / 64;
  __shared__ float wbest[2][NW];
  __shared__ int wbesti[2][NW];
  __shared__ float4 wq[2][NW];  // coordinates of every wave's candidate: the next iteration needs no global load
  const float4* p = pos4 + s0;
  float mind[PPT];
  float4 pc[CACHE ? PPT : 1];
#pragma unroll
  for (int j = 0; j < PPT; ++j) {
    const int i = t + FPS_THREADS * j;
    mind[j] = i < n ? __builtin_inff() : -1.f;  // (a valid point's distance is >= 0: padding never wins)
    if (CACHE) pc[j] = i < n ? p[i] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  int cur = start ? start[b] : 0;
  cur = cur < 0 ? 0 : (cur >= n ? n - 1 : cur);
  float4 q = p[cur];
  const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, n * 16, 0x00020000);
  int32_t mine = 0;  // selection s is kept by thread s % 1024 and written in coalesced blocks of 1024: a store per
                     // iteration would put its write acknowledgement (the barrier waits for vmcnt) on the loop-carried path
  for (int s = 0; s < m; ++s) {
    if (t == (s & (FPS_THREADS - 1))) mine = (int32_t)(s0 + cur);
    if ((s & (FPS_THREADS - 1)) == FPS_THREADS - 1 || s == m - 1) {
      const int base = s & ~(FPS_THREADS - 1);
      if (base + t <= s) idx_out[o0 + base + t] = mine;
    }
    if (s == m - 1) break;
    float best = -2.f;
    int besti = INT_MAX;
    float bx = 0.f, by = 0.f, bz = 0.f;
    constexpr int CH = PPT < 8 ? PPT : 8;  // re-read positions: 8 loads in flight at a time (all 64 at once spill)
#pragma unroll
    for (int j0 = 0; j0 < PPT; j0 += CH) {
      float4 pj[CH];
#pragma unroll
      for (int u = 0; u < CH; ++u) {
        const int i = t + FPS_THREADS * (j0 + u);
        if (CACHE) {
          pj[u] = pc[j0 + u];
        } else {  // 32-bit byte offsets into the cloud's buffer descriptor (num_records = n * 16): the hardware range check
                  // covers voffset only — NOT soffset — so the whole row offset sits there; rows >= n read as 0
          const f32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rp, (unsigned)(t + FPS_THREADS * (j0 + u)) * 16u, 0, 0);
          pj[u] = make_float4(v[0], v[1], v[2], v[3]);
        }
      }
#pragma unroll
      for (int u = 0; u < CH; ++u) {
        const int j = j0 + u, i = t + FPS_THREADS * j;
        const float d = fps_d2(pj[u], q);
        if (i < n) mind[j] = fminf(mind[j], d);
        if (mind[j] > best) {  // (ascending i inside a thread: '>' keeps the first)
          best = mind[j]; besti = i; bx = pj[u].x; by = pj[u].y; bz = pj[u].z;
        }
      }
      if (!CACHE) __builtin_amdgcn_sched_barrier(0);
    }
    const int own = besti;
    fps_step_dpp<0xB1>(best, besti);   // quad_perm [1,0,3,2]
    fps_step_dpp<0x4E>(best, besti);   // quad_perm [2,3,0,1]
    fps_step_dpp<0x141>(best, besti);  // row_half_mirror: quads (0,1), (2,3) — every quad is uniform by now
    fps_step_dpp<0x140>(best, besti);  // row_mirror: halves of the row
    {
      float a, c, ia, ic;
      xgroup_pair16(best, a, c); xgroup_pair16(__int_as_float(besti), ia, ic);
      fps_pick(a, __float_as_int(ia), c, __float_as_int(ic), best, besti);
      xgroup_pair32(best, a, c); xgroup_pair32(__int_as_float(besti), ia, ic);
      fps_pick(a, __float_as_int(ia), c, __float_as_int(ic), best, besti);
    }
    const int par = s & 1;
    if (own == besti) { wbest[par][wid] = best; wbesti[par][wid] = besti; wq[par][wid] = make_float4(bx, by, bz, 0.f); }
    __syncthreads();
    // second stage in every wave at once: lane l takes candidate l % 16, four DPP steps inside the row of 16 (a serial scan
    // of the 16 candidates compiles to 32 dependent LDS round trips: 2 us per iteration)
    static_assert(NW == 16, "one candidate per lane of a DPP row");
    int bw = lane & 15;
    best = wbest[par][bw]; besti = wbesti[par][bw];
    fps_step_dpp3<0xB1>(best, besti, bw);
    fps_step_dpp3<0x4E>(best, besti, bw);
    fps_step_dpp3<0x141>(best, besti, bw);
    fps_step_dpp3<0x140>(best, besti, bw);
    cur = besti;
    q = wq[par][bw];
  }
}

// ------------------------------------------------------------------------------------------
// farthest-point sampling with EXACT bucket skipping (round 4; the "block order for exact sub-set skipping" of DESIGN r3).
// The serial chain of m iterations stays, but an iteration no longer touches every point: the points of a cloud are taken in
// the cell-sorted order of its kNN grid (csrc/knn.hip: float4 records x, y, z, original row — spatially compact runs) and cut
// into buckets of 64 consecutive records.  A bucket keeps its bounding box, the largest running minimum distance of its
// points (bmax) and that point.  Selecting q can lower a point's minimum only if d2(q, p) < mind[p]; for a whole bucket
//     d2(q, p) >= d2(q, box) for every p inside      (the same fp32 expression on both sides: subtraction, squares and sums
//                                                     are monotonic, so the inequality holds for the ROUNDED values too)
// so a bucket with d2(q, box) >= bmax is skipped without loading a point — after the first few hundred selections that is
// all but the handful of buckets around q.  Bucket b is owned by lane b / 16 of wave b % 16 (neighbouring buckets — affected
// together — go to different waves); the owner wave updates the 64 points (positions: one coalesced 1 KB load from L2;
// running minima: LDS, 40 000 x 4 bytes = all of it), reduces (value, original row) with the total order of the oracle
// (larger distance, then smaller ORIGINAL row) and leaves the result in the owner lane's registers.  The arg-max over the
// bucket maxima is the two-stage reduction of fps_kernel.  Same arithmetic, same order => bit-identical index lists.
// Clouds of 16 385 ... 40 000 points (below, fps_kernel keeps every position in registers and is as fast; above, the minima
// no longer fit in LDS).  16 x 40 000 -> 10 000 -> 2 500 -> 625 points: 71.5 -> 36.6 ms for the three launches (r04d), ...
// (Also built in round 4 and removed: G workgroups per cloud that exchange one candidate per iteration through L2 — 70 ms, no
// better than one workgroup: an agent-scope round trip per iteration costs what the 40 points per thread cost;
// profiles/r04c_bench.json.)
// ------------------------------------------------------------------------------------------
#define FPSB_MAXN 40000
#define FPSB_BS 64
template <int CTRL>
__device__ __forceinline__ void fpsb_step(float& best, int& besti, int& tag) {  // (value desc, original row asc); tag rides along
  const float ob = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(best), CTRL, 0xF, 0xF, false));
  const int oi = __builtin_amdgcn_update_dpp(0, besti, CTRL, 0xF, 0xF, false);
  const int ot = __builtin_amdgcn_update_dpp(0, tag, CTRL, 0xF, 0xF, false);
  if (ob > best || (ob == best && oi < besti)) { best = ob; besti = oi; tag = ot; }
}
__device__ __forceinline__ void fpsb_pick3(float a, int ia, int ta, float b, int ib, int tb, float& best, int& besti, int& tag) {
  const bool t = b > a || (b == a && ib < ia);
  best = t ? b : a; besti = t ? ib : ia; tag = t ? tb : ta;
}
__device__ __forceinline__ void fpsb_wave_argmax(float& best, int& besti, int& tag) {
  fpsb_step<0xB1>(best, besti, tag);
  fpsb_step<0x4E>(best, besti, tag);
  fpsb_step<0x141>(best, besti, tag);
  fpsb_step<0x140>(best, besti, tag);
  // across the four rows of 16 lanes: row swaps (VALU), not __shfl_xor (an LDS-pipe round trip per value and step)
  float a, c, ia, ic, ta, tc;
  xgroup_pair16(best, a, c); xgroup_pair16(__int_as_float(besti), ia, ic); xgroup_pair16(__int_as_float(tag), ta, tc);
  fpsb_pick3(a, __float_as_int(ia), __float_as_int(ta), c, __float_as_int(ic), __float_as_int(tc), best, besti, tag);
  xgroup_pair32(best, a, c); xgroup_pair32(__int_as_float(besti), ia, ic); xgroup_pair32(__int_as_float(tag), ta, tc);
  fpsb_pick3(a, __float_as_int(ia), __float_as_int(ta), c, __float_as_int(ic), __float_as_int(tc), best, besti, tag);
}

__device__ __forceinline__ float readlane_f(float v, int l) {  // (the builtin moves 32 raw bits: reinterpret, do not convert)
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}
// one affected bucket: lane-wise minimum update of its 64 records, the bucket's new (max, original row, coordinates) into the
// owner lane's registers
__device__ __forceinline__ void fpsb_update(const float4 v, int i, int l, int n, int64_t s0, int lane, const float4 q, float* mind,
                                            float& bmax, int& barg, float& bx, float& by, float& bz) {
  const float d = fps_d2(v, q);
  float nm = mind[i];
  if (i < n) { nm = fminf(nm, d); mind[i] = nm; }
  float best = nm;
  int besti = i < n ? (int)(__float_as_int(v.w) - (int)s0) : INT_MAX;
  int tag = lane;
  fpsb_wave_argmax(best, besti, tag);
  const int tl = __builtin_amdgcn_readfirstlane(tag);
  const float wx = readlane_f(v.x, tl), wy = readlane_f(v.y, tl), wz = readlane_f(v.z, tl);
  if (lane == l) { bmax = best; barg = besti; bx = wx; by = wy; bz = wz; }
}

__global__ __launch_bounds__(FPS_THREADS) void fps_bucket_kernel(const float4* __restrict__ sorted4, const int32_t* __restrict__ inv,
                                                                const int64_t* __restrict__ ptr_src,
                                                                const int64_t* __restrict__ ptr_out,
                                                                const int32_t* __restrict__ start, int32_t* __restrict__ idx_out) {
  __shared__ float mind[FPSB_MAXN];
  constexpr int NW = FPS_THREADS / 64;
  __shared__ float wbest[2][NW];
  __shared__ int wbesti[2][NW];
  __shared__ float4 wq[2][NW];
  const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wid = t >> 6;
  const int64_t s0 = ptr_src[b], o0 = ptr_out[b];
  const int n = (int)(ptr_src[b + 1] - s0), m = (int)(ptr_out[b + 1] - o0);
  if (m <= 0 || n <= 0) return;
  const float4* p = sorted4 + s0;  // cell-sorted records of this cloud; .w = original GLOBAL row
  const int nb = (n + FPSB_BS - 1) / FPSB_BS;
  for (int i = t; i < nb * FPSB_BS; i += FPS_THREADS) mind[i] = i < n ? __builtin_inff() : -1.f;
  // my bucket: lane l of wave w owns bucket l * 16 + w
  const int mb = lane * NW + wid;
  const bool has = mb < nb;
  float lox = 0.f, loy = 0.f, loz = 0.f, hix = 0.f, hiy = 0.f, hiz = 0.f;
  float bmax = has ? __builtin_inff() : -2.f;  // +inf: every bucket is updated by the first selection
  int barg = INT_MAX;                          // original cloud-relative row of the bucket's farthest point
  float bx = 0.f, by = 0.f, bz = 0.f;          // ... and its coordinates
  // bounding boxes: the owner wave scans its buckets (64 records = one coalesced load each)
  for (int l = 0; l < 64; ++l) {
    const int bb = l * NW + wid;
    if (bb >= nb) break;  // (wave-uniform)
    const int i = bb * FPSB_BS + lane;
    const float4 v = p[i < n ? i : n - 1];
    float ax = v.x, ay = v.y, az = v.z, cx = v.x, cy = v.y, cz = v.z;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      ax = fminf(ax, __shfl_xor(ax, o, 64)); ay = fminf(ay, __shfl_xor(ay, o, 64)); az = fminf(az, __shfl_xor(az, o, 64));
      cx = fmaxf(cx, __shfl_xor(cx, o, 64)); cy = fmaxf(cy, __shfl_xor(cy, o, 64)); cz = fmaxf(cz, __shfl_xor(cz, o, 64));
    }
    if (lane == l) { lox = ax; loy = ay; loz = az; hix = cx; hiy = cy; hiz = cz; }
  }
  int cur = start ? start[b] : 0;
  cur = cur < 0 ? 0 : (cur >= n ? n - 1 : cur);
  float4 q = sorted4[inv[s0 + cur]];  // (inv: original global row -> global cell-sorted slot)
  __syncthreads();
  int32_t mine = 0;
  float cbest = -2.f, cqx = 0.f, cqy = 0.f, cqz = 0.f;  // this wave's candidate (wave-uniform), kept across iterations
  int cbesti = INT_MAX;
  for (int s = 0; s < m; ++s) {
    if (t == (s & (FPS_THREADS - 1))) mine = (int32_t)(s0 + cur);
    if ((s & (FPS_THREADS - 1)) == FPS_THREADS - 1 || s == m - 1) {
      const int base = s & ~(FPS_THREADS - 1);
      if (base + t <= s) idx_out[o0 + base + t] = mine;
    }
    if (s == m - 1) break;
    // ---- which of this wave's buckets can the new point reach?
    bool aff = false;
    if (has) {
      const float4 c = make_float4(fminf(fmaxf(q.x, lox), hix), fminf(fmaxf(q.y, loy), hiy), fminf(fmaxf(q.z, loz), hiz), 0.f);
      aff = fps_d2(c, q) < bmax;  // c = the box's nearest point to q: |c - q| <= |p - q| per axis for every p in the box
    }
    unsigned long long mask = __builtin_amdgcn_ballot_w64(aff);
    const bool touched = mask != 0;
    while (mask) {
      // two buckets per trip: both position loads are in flight before the first one is reduced
      const int l = __builtin_ctzll(mask);
      mask &= mask - 1;
      const int l2 = mask ? __builtin_ctzll(mask) : l;
      const int i2 = (l2 * NW + wid) * FPSB_BS + lane;
      const int bb = l * NW + wid;
      const int i = bb * FPSB_BS + lane;
      float4 v = p[i < n ? i : n - 1];
      const float4 v2 = p[i2 < n ? i2 : n - 1];
      fpsb_update(v, i, l, n, s0, lane, q, mind, bmax, barg, bx, by, bz);
      if (mask) {
        mask &= mask - 1;
        fpsb_update(v2, i2, l2, n, s0, lane, q, mind, bmax, barg, bx, by, bz);
      }
    }
    // ---- arg-max over the bucket maxima: this wave's candidate (recomputed only when one of its buckets changed — the
    // bucket of the point just selected always does), then the 16 wave candidates
    if (touched || s == 0) {
      float best = bmax;
      int besti = barg, tag = lane;
      fpsb_wave_argmax(best, besti, tag);
      const int tl = __builtin_amdgcn_readfirstlane(tag);
      cbest = best; cbesti = besti;
      cqx = readlane_f(bx, tl); cqy = readlane_f(by, tl); cqz = readlane_f(bz, tl);
    }
    const int par = s & 1;
    if (lane == 0) { wbest[par][wid] = cbest; wbesti[par][wid] = cbesti; wq[par][wid] = make_float4(cqx, cqy, cqz, 0.f); }
    __syncthreads();
    int bw = lane & 15;
    float best = wbest[par][bw];
    int besti = wbesti[par][bw];
    fps_step_dpp3<0xB1>(best, besti, bw);
    fps_step_dpp3<0x4E>(best, besti, bw);
    fps_step_dpp3<0x141>(best, besti, bw);
    fps_step_dpp3<0x140>(best, besti, bw);
    cur = besti;
    q = wq[par][bw];
  }
}

// sorted_ws: the cloud set's BUILT kNN workspace (m3d_knn_build over the same points / ptr_src) or NULL.  With it, clouds of
// 4 097 ... 40 000 points are sampled by fps_bucket_kernel (exact bucket skipping); index lists are identical either way
extern "C" int m3d_fps_sorted(const void* sorted_ws, int64_t n_src, const int64_t* ptr_src, const int64_t* ptr_out,
                              int32_t num_clouds, int64_t max_points, const int32_t* start, int32_t* idx_out, void* stream) {
  if (num_clouds < 0 || max_points < 0 || n_src < 0) return M3D_ERR_INVALID;
  if (num_clouds == 0 || max_points == 0) return M3D_OK;
  if (!sorted_ws || !ptr_src || !ptr_out || !idx_out) return M3D_ERR_INVALID;
  if (max_points > FPSB_MAXN) return M3D_ERR_UNSUPPORTED;
  const char* base = (const char*)sorted_ws;
  const float4* sorted4 = (const float4*)(base + m3d_knn_workspace_offset(n_src, num_clouds, 0));
  const int32_t* inv = (const int32_t*)(base + m3d_knn_workspace_offset(n_src, num_clouds, 2));
  hipLaunchKernelGGL(fps_bucket_kernel, dim3((unsigned)num_clouds), dim3(FPS_THREADS), 0, (hipStream_t)stream, sorted4, inv,
                     ptr_src, ptr_out, start, idx_out);
  M3D_CHECK_LAUNCH();
  return M3D_OK;
}

extern "C" int m3d_fps(const float* pos4, const int64_t* ptr_src, const int64_t* ptr_out, int32_t num_clouds,
                       int64_t max_points, const int32_t* start, int32_t* idx_out, void* stream) {
  if (num_clouds < 0 || max_points < 0) return M3D_ERR_INVALID;
  if (num_clouds == 0 || max_points == 0) return M3D_OK;
  if (!pos4 || !ptr_src || !ptr_out || !idx_out || (((uintptr_t)pos4) & 15)) return M3D_ERR_INVALID;
  if (max_points > (int64_t)FPS_THREADS * 64) return M3D_ERR_UNSUPPORTED;  // 65 536 points per cloud
  hipStream_t st = (hipStream_t)stream;
  const float4* p = (const float4*)pos4;
  const dim3 grid((unsigned)num_clouds), block(FPS_THREADS);
  if (max_points <= FPS_THREADS * 4) hipLaunchKernelGGL((fps_kernel<4, true>), grid, block, 0, st, p, ptr_src, ptr_out, start, idx_out);
  else if (max_points <= FPS_THREADS * 16) hipLaunchKernelGGL((fps_kernel<16, true>), grid, block, 0, st, p, ptr_src, ptr_out, start, idx_out);
  else if (max_points <= FPS_THREADS * 40) hipLaunchKernelGGL((fps_kernel<40, false>), grid, block, 0, st, p, ptr_src, ptr_out, start, idx_out);  // the 40 000-point node budget (points_budget.yaml:24-27)
  else hipLaunchKernelGGL((fps_kernel<64, false>), grid, block, 0, st, p, ptr_src, ptr_out, start, idx_out);
  M3D_CHECK_LAUNCH();
  return M3D_OK;
}

// ------------------------------------------------------------------------------------------
// grouping: edge e = seg[i] + k (k < seg[i+1] - seg[i] <= K) of centre i gets the row
//   out[e] = [ x[j][0..C), pos_j - pos_i (3), zeros up to ldo ],  j = nbr[i][k]
// plus esrc[e] = j and ectr[e] = i.  One thread per output float: coalesced row writes.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sa_group_kernel(const float* __restrict__ x, int64_t ldx, int C,
                                                       const float4* __restrict__ pos_src, const float4* __restrict__ pos_ctr,
                                                       const int32_t* __restrict__ nbr, const int64_t* __restrict__ seg,
                                                       int64_t m, int K, float* __restrict__ out, int ldo,
                                                       int32_t* __restrict__ esrc, int32_t* __restrict__ ectr) {
  const int64_t total = m * K * (int64_t)ldo;
  for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < total; g += (int64_t)gridDim.x * 256) {
    const int c = (int)(g % ldo);
    const int64_t ik = g / ldo;
    const int k = (int)(ik % K);
    const int64_t i = ik / K;
    const int64_t e0 = seg[i];
    if (k >= (int)(seg[i + 1] - e0)) continue;
    const int j = nbr[ik];
    const int64_t e = e0 + k;
    float v = 0.f;
    if (c < C) {
      v = x[(int64_t)j * ldx + c];
    } else if (c < C + 3) {
      const float4 a = pos_src[j], o = pos_ctr[i];
      v = c == C ? a.x - o.x : (c == C + 1 ? a.y - o.y : a.z - o.z);
    }
    out[e * ldo + c] = v;
    if (c == 0) { esrc[e] = j; ectr[e] = (int32_t)i; }
  }
}

extern "C" int m3d_sa_group(const float* x, int64_t ldx, int32_t C, const float* pos4_src, const float* pos4_ctr,
                            const int32_t* nbr, const int64_t* seg, int64_t m, int32_t K, float* out, int64_t ldo,
                            int32_t* esrc, int32_t* ectr, void* stream) {
  if (m < 0 || K <= 0 || C < 0 || ldo < C + 3 || ldo > INT_MAX) return M3D_ERR_INVALID;
  if (m == 0) return M3D_OK;
  if ((C > 0 && !x) || !pos4_src || !pos4_ctr || !nbr || !seg || !out || !esrc || !ectr) return M3D_ERR_INVALID;
  if ((((uintptr_t)pos4_src) & 15) || (((uintptr_t)pos4_ctr) & 15)) return M3D_ERR_INVALID;
  int64_t gx = m3d_cdiv(m * K * ldo, 256 * 4);
  if (gx > 16384) gx = 16384;
  if (gx < 1) gx = 1;
  hipLaunchKernelGGL(sa_group_kernel, dim3((unsigned)gx), dim3(256), 0, (hipStream_t)stream, x, ldx, C,
                     (const float4*)pos4_src, (const float4*)pos4_ctr, nbr, seg, m, K, out, (int)ldo, esrc, ectr);
  M3D_CHECK_LAUNCH();
  return M3D_OK;
}

// dx[esrc[e]][c] += dE[e][c], c < C  (dx zeroed / pre-loaded by the caller)
__global__ __launch_bounds__(256) void sa_group_bwd_kernel(const float* __restrict__ de, int64_t ld, const int32_t* __restrict__ esrc,
                                                           int64_t E, int C, float* __restrict__ dx, int64_t lddx) {
  const int64_t total = E * C;
  for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < total; g += (int64_t)gridDim.x * 256) {
    const int c = (int)(g % C);
    const int64_t e = g / C;
    atomicAdd(dx + (int64_t)esrc[e] * lddx + c, de[e * ld + c]);
  }
}

extern "C" int m3d_sa_group_bwd(const float* de, int64_t ld, const int32_t* esrc, int64_t E, int32_t C, float* dx,
                                int64_t lddx, void* stream) {
  if (E < 0 || C < 0 || ld < C || lddx < C) return M3D_ERR_INVALID;
  if (E == 0 || C == 0) return M3D_OK;
  if (!de || !esrc || !dx) return M3D_ERR_INVALID;
  int64_t gx = m3d_cdiv(E * C, 256 * 2);
  if (gx > 16384) gx = 16384;
  hipLaunchKernelGGL(sa_group_bwd_kernel, dim3((unsigned)gx), dim3(256), 0, (hipStream_t)stream, de, ld, esrc, E, C, dx, lddx);
  M3D_CHECK_LAUNCH();
  return M3D_OK;
}

// ------------------------------------------------------------------------------------------
// max over the edges of a centre, per channel; arg[i][c] = k of the first maximal edge (seg[i] + k).
// A centre without edges (cannot happen: the centre is its own neighbour) would get 0 / -1.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void seg_max_kernel(const float* __restrict__ y, int64_t ldy, const int64_t* __restrict__ seg,
                                                      int64_t m, int C, float* __restrict__ out, int32_t* __restrict__ arg) {
  const int64_t total = m * C;
  for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < total; g += (int64_t)gridDim.x * 256) {
    const int c = (int)(g % C);
    const int64_t i = g / C;
    const int64_t e0 = seg[i];
    const int len = (int)(seg[i + 1] - e0);
    float best = 0.f;
    int bk = -1;
    for (int k = 0; k < len; ++k) {
      const float v = y[(e0 + k) * ldy + c];
      if (bk < 0 || v > best) { best = v; bk = k; }
    }
    out[g] = best;
    arg[g] = bk;
  }
}

extern "C" int m3d_seg_max(const float* y, int64_t ldy, const int64_t* seg, int64_t m, int32_t C, float* out,
                           int32_t* arg, void* stream) {
  if (m < 0 || C < 0 || ldy < C) return M3D_ERR_INVALID;
  if (m == 0 || C == 0) return M3D_OK;
  if (!y || !seg || !out || !arg) return M3D_ERR_INVALID;
  int64_t gx = m3d_cdiv(m * C, 256);
  if (gx > 16384) gx = 16384;
  hipLaunchKernelGGL(seg_max_kernel, dim3((unsigned)gx), dim3(256), 0, (hipStream_t)stream, y, ldy, seg, m, C, out, arg);
  M3D_CHECK_LAUNCH();
  return M3D_OK;
}

// dy[e][c] = dout[i][c] if e is the arg-max edge of (i = ectr[e], c), else 0: every element written, no zero fill
__global__ __launch_bounds__(256) void seg_max_bwd_kernel(const float* __restrict__ dout, const int32_t* __restrict__ arg,
                                                          const int64_t* __restrict__ seg, const int32_t* __restrict__ ectr,
                                                          int64_t E, int C, float* __restrict__ dy) {
  const int64_t total = E * C;
  for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < total; g += (int64_t)gridDim.x * 256) {
    const int c = (int)(g % C);
    const int64_t e = g / C;
    const int64_t i = ectr[e];
    const int k = (int)(e - seg[i]);
    dy[g] = arg[i * C + c] == k ? dout[i * C + c] : 0.f;
  }
}

extern "C" int m3d_seg_max_bwd(const float* dout, const int32_t* arg, const int64_t* seg, const int32_t* ectr, int64_t E,
                               int32_t C, float* dy, void* stream) {
  if (E < 0 || C < 0) return M3D_ERR_INVALID;
  if (E == 0 || C == 0) return M3D_OK;
  if (!dout || !arg || !seg || !ectr || !dy) return M3D_ERR_INVALID;
  int64_t gx = m3d_cdiv(E * C, 256 * 2);
  if (gx > 16384) gx = 16384;
  hipLaunchKernelGGL(seg_max_bwd_kernel, dim3((unsigned)gx), dim3(256), 0, (hipStream_t)stream, dout, arg, seg, ectr, E, C, dy);
  M3D_CHECK_LAUNCH();
  return M3D_OK;
}
