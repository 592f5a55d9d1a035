// Inverse-squared-distance interpolation of per-point logits/features (gfx950).
//
// Replaces torch_geometric.nn.knn_interpolate's arithmetic
// (/root/reference/myria3d/models/model.py:90-98 with k = interpolation_k = 10 — run on the CPU by the reference —
// and /root/reference/myria3d/models/modules/pyg_randla_net.py:250 with k = 1) once m3d_knn_query has produced the
// dense neighbour table:   y[q, c] = sum_k w_k x[idx[q,k], c] / sum_k w_k,   w_k = 1 / max(d2[q,k], 1e-16).
#include "m3d_common.h"
#include "../../include/m3d_hip.h"

__global__ __launch_bounds__(256) void idw_kernel(const float* __restrict__ x, int64_t ldx,
                                                  const int32_t* __restrict__ idx, const float* __restrict__ d2,
                                                  int64_t nq, int k, int C, float* __restrict__ y) {
  const int64_t total = nq * C;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
    const int64_t q = t / C;
    const int c = (int)(t % C);
    float num = 0.f, den = 0.f;
    for (int j = 0; j < k; ++j) {
      const int s = idx[q * k + j];
      if (s >= 0) {
        const float w = 1.f / fmaxf(d2[q * k + j], 1e-16f);
        num += x[(int64_t)s * ldx + c] * w;
        den += w;
      }
    }
    y[t] = num / den;
  }
}

extern "C" int m3d_idw_interpolate_fwd(const float* x, int64_t ldx, const int32_t* idx, const float* d2, int64_t n_qry,
                                       int32_t k, int32_t C, float* y, void* stream) {
  if (n_qry < 0 || k < 1 || C < 0) return M3D_ERR_INVALID;
  if (n_qry == 0 || C == 0) return M3D_OK;
  if (!x || !idx || !d2 || !y) return M3D_ERR_INVALID;
  int64_t gx = m3d_cdiv(n_qry * C, 256);
  if (gx > 16384) gx = 16384;
  hipLaunchKernelGGL(idw_kernel, dim3((unsigned)gx), dim3(256), 0, (hipStream_t)stream, x, ldx, idx, d2, n_qry, k, C, y);
  M3D_CHECK_LAUNCH();
  return M3D_OK;
}
