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

// ------------------------------------------------------------------------------------------
// Per-point class probabilities, predicted class and Shannon entropy from the merged logits: the arithmetic of
// Interpolator.reduce_predictions_and_save (/root/reference/myria3d/models/interpolation.py:142-169) after the
// scatter_sum merge (interpolation.py:98-121):  row = reduced_logits[idx[i]];  probas = Softmax(dim=1)(row);
// preds = argmax(row) (first maximum);  entropy = Categorical(probs=probas).entropy(), i.e. with
// p' = probas / sum(probas):  -sum_c p'_c * log(clamp(p'_c, eps, 1 - eps)),  eps = FLT_EPSILON.
// One lane per point; C is small (6-7 classes): the row stays in registers for C <= 16.
// ------------------------------------------------------------------------------------------
template <int CMAX>
__global__ __launch_bounds__(256) void predict_reduce_kernel(const float* __restrict__ logits, int64_t ld,
                                                             const int32_t* __restrict__ idx, int64_t m, int C,
                                                             float* __restrict__ probas, int64_t ldp,
                                                             int32_t* __restrict__ preds,
                                                             float* __restrict__ entropy) {
  const float EPS = 1.1920929e-7f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < m; i += (int64_t)gridDim.x * 256) {
    const int64_t r = idx ? (int64_t)idx[i] : i;
    const float* row = logits + r * ld;
    float v[CMAX];
    float mx = -__builtin_inff();
    int am = 0;
#pragma unroll
    for (int c = 0; c < CMAX; ++c) {
      v[c] = c < C ? row[c] : -__builtin_inff();
      if (v[c] > mx) { mx = v[c]; am = c; }
    }
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < CMAX; ++c) {
      v[c] = c < C ? expf(v[c] - mx) : 0.f;
      sum += v[c];
    }
    float psum = 0.f;
#pragma unroll
    for (int c = 0; c < CMAX; ++c) {
      v[c] = v[c] / sum;
      psum += v[c];
    }
    if (probas) {
#pragma unroll
      for (int c = 0; c < CMAX; ++c)
        if (c < C) probas[i * ldp + c] = v[c];
    }
    if (preds) preds[i] = am;
    if (entropy) {
      float h = 0.f;
#pragma unroll
      for (int c = 0; c < CMAX; ++c) {
        if (c < C) {
          const float pn = v[c] / psum;
          h += pn * logf(fminf(fmaxf(pn, EPS), 1.f - EPS));
        }
      }
      entropy[i] = -h;
    }
  }
}

extern "C" int m3d_predict_reduce(const float* logits, int64_t ld, const int32_t* idx, int64_t m, int32_t C,
                                  float* probas, int64_t ldp, int32_t* preds, float* entropy, void* stream) {
  if (m < 0 || C < 1) return M3D_ERR_INVALID;
  if (C > 64) return M3D_ERR_UNSUPPORTED;
  if (m == 0) return M3D_OK;
  if (!logits || ld < C || (probas && ldp < C)) return M3D_ERR_INVALID;
  int64_t gx = m3d_cdiv(m, 256);
  if (gx > 16384) gx = 16384;
  dim3 grid((unsigned)gx), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (C <= 8)
    hipLaunchKernelGGL(predict_reduce_kernel<8>, grid, block, 0, st, logits, ld, idx, m, C, probas, ldp, preds, entropy);
  else if (C <= 16)
    hipLaunchKernelGGL(predict_reduce_kernel<16>, grid, block, 0, st, logits, ld, idx, m, C, probas, ldp, preds, entropy);
  else
    hipLaunchKernelGGL(predict_reduce_kernel<64>, grid, block, 0, st, logits, ld, idx, m, C, probas, ldp, preds, entropy);
  M3D_CHECK_LAUNCH();
  return M3D_OK;
}
