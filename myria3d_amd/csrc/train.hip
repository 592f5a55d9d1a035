// Loss and optimizer kernels of the training step (gfx950).
//
// The reference gets these from torch through Lightning:
//   * torch.nn.CrossEntropyLoss(ignore_index=65) applied to the net's logits
//     (/root/reference/myria3d/models/model.py:118, configs/model/criterion/CrossEntropyLoss.yaml:1-3)
//   * torch.optim.Adam(lr=3.93e-3) (/root/reference/configs/model/optimizer/Adam.yaml:1-4,
//     configs/model/pyg_randla_net_model.yaml:4)
// torch's generic kernels cost ~0.3 ms (nll_loss fwd+bwd on [204 800, 6]) and ~250 tiny launches (capturable
// Adam over 141 parameter tensors) per step; here the loss is two streaming kernels and the optimizer is ONE
// launch over the flat parameter / gradient buffers (the same flat buffer RCCL all-reduces).
#include "m3d_common.h"
#include "../../include/m3d_hip.h"

#define CE_MAXC 64

// lse[i] = logsumexp(row i); acc[0] = sum of the per-row losses, acc[1] = number of rows with target != ignore_index,
// loss[0] = acc[0] / acc[1].  Workgroup b stores its two partial sums to acc[4 + 2b ..] and takes an arrival ticket
// (acc[2], first 4 bytes); the workgroup that arrives last adds the partials and writes the results — no finalize launch
// (5 us apiece in a replayed graph) and ONE same-address atomic per workgroup instead of three (~15 ns each, serialised).
#define CE_MAX_BLOCKS 256
__global__ __launch_bounds__(256) void ce_fwd_kernel(const float* __restrict__ logits, int64_t ld,
                                                     const int64_t* __restrict__ target, int64_t n, int C,
                                                     int64_t ignore_index, float* __restrict__ lse,
                                                     double* __restrict__ acc, float* __restrict__ loss) {
  __shared__ double red[2][4];
  __shared__ bool last;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  double ls = 0.0, cnt = 0.0;
  const int64_t gstride = (int64_t)gridDim.x * 256;
  // C <= 8 (this network: 6 / 7 classes): four rows per thread, their logits and targets all requested before the first is
  // used, the target's logit picked from registers.  (The rolled loop below re-reads a row three times and reads row[t]
  // behind target[i]: ~3 dependent round trips per row, 3-4 rows per thread, 17 us for 204 800 rows of 24 B.)
  int64_t i = (int64_t)blockIdx.x * 256 + tid;
  if (C <= 8) {
    for (; i < n; i += 4 * gstride) {
      float v[4][8];
      int64_t tg[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t r = i + u * gstride < n ? i + u * gstride : n - 1;  // (clamped: branch-free loads)
        tg[u] = target[r];
#pragma unroll
        for (int c = 0; c < 8; ++c) v[u][c] = logits[r * ld + (c < C ? c : C - 1)];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (i + u * gstride >= n) break;
        float mx = -__builtin_inff();
#pragma unroll
        for (int c = 0; c < 8; ++c) mx = c < C ? fmaxf(mx, v[u][c]) : mx;
        float s = 0.f, pick = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          s += c < C ? expf(v[u][c] - mx) : 0.f;
          pick = tg[u] == c ? v[u][c] : pick;
        }
        const float l = mx + logf(s);
        lse[i + u * gstride] = l;
        if (tg[u] != ignore_index) {
          if (tg[u] >= 0 && tg[u] < C) {
            ls += (double)(l - pick);
            cnt += 1.0;
          } else {
            ls += (double)__builtin_nanf("");  // (see below)
          }
        }
      }
    }
  }
  for (; i < n; i += gstride) {
    const float* row = logits + i * ld;
    float mx = -__builtin_inff();
    for (int c = 0; c < C; ++c) mx = fmaxf(mx, row[c]);
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += expf(row[c] - mx);
    const float l = mx + logf(s);
    lse[i] = l;
    const int64_t t = target[i];
    if (t != ignore_index) {
      if (t >= 0 && t < C) {
        ls += (double)(l - row[t]);
        cnt += 1.0;
      } else {
        // a class code outside [0, C) that is not ignore_index (e.g. an unmapped LAS code): torch's CrossEntropyLoss
        // raises / device-asserts; here the loss is poisoned with NaN so the mistake cannot pass silently
        ls += (double)__builtin_nanf("");
      }
    }
  }
  ls = wave_sum_d(ls);
  cnt = wave_sum_d(cnt);
  if (lane == 0) { red[0][wid] = ls; red[1][wid] = cnt; }
  __syncthreads();
  if (tid == 0) {
    double* part = acc + 4 + 2 * (size_t)blockIdx.x;
    __hip_atomic_store(&part[0], red[0][0] + red[0][1] + red[0][2] + red[0][3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&part[1], red[1][0] + red[1][1] + red[1][2] + red[1][3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence();
    last = atomicAdd((unsigned*)&acc[2], 1u) == gridDim.x - 1;
  }
  __syncthreads();
  if (!last) return;
  // every other workgroup's partials precede its ticket
  double s = 0.0, c = 0.0;
  if (tid < (int)gridDim.x) {
    s = __hip_atomic_load(&acc[4 + 2 * tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    c = __hip_atomic_load(&acc[5 + 2 * tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  s = wave_sum_d(s);
  c = wave_sum_d(c);
  __syncthreads();
  if (lane == 0) { red[0][wid] = s; red[1][wid] = c; }
  __syncthreads();
  if (tid == 0) {
    const double st = red[0][0] + red[0][1] + red[0][2] + red[0][3], ct = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    acc[0] = st;
    acc[1] = ct;
    loss[0] = (float)(st / ct);  // mean over the non-ignored rows (0/0 = NaN when every row is ignored, like torch)
    *(unsigned*)&acc[2] = 0u;
  }
}

__global__ void ce_finalize_kernel(double* __restrict__ acc, float* __restrict__ loss) {
  acc[0] = 0.0; acc[1] = 0.0;
  loss[0] = (float)(acc[0] / acc[1]);  // (n == 0: no forward workgroup exists; NaN like torch)
}

// dlogits[i, c] = gout * (softmax(i)[c] - [c == target_i]) / count   (0 for ignored rows)
template <bool H>
__global__ __launch_bounds__(256) void ce_bwd_kernel(const float* __restrict__ logits, int64_t ld,
                                                     const int64_t* __restrict__ target, int64_t n, int C,
                                                     int64_t ignore_index, const float* __restrict__ lse,
                                                     const double* __restrict__ acc, const float* __restrict__ gout,
                                                     void* __restrict__ dlogits) {
  const float g = gout[0] / (float)acc[1];
  const int64_t total = n * C;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
    const int64_t i = t / C;
    const int c = (int)(t % C);
    const int64_t y = target[i];
    float d = 0.f;
    if (y != ignore_index && y >= 0 && y < C) d = g * (expf(logits[i * ld + c] - lse[i]) - (c == (int)y ? 1.f : 0.f));
    io_store1<H>(dlogits, (size_t)t, d);
  }
}

extern "C" int m3d_ce_loss_fwd(const float* logits, int64_t ld, const int64_t* target, int64_t n, int32_t C,
                               int64_t ignore_index, float* lse, double* acc4, float* loss, int32_t flags, void* stream) {
  if (n < 0 || C < 1) return M3D_ERR_INVALID;
  if (!acc4 || !loss) return M3D_ERR_INVALID;
  if (n > 0 && (!logits || !target || !lse)) return M3D_ERR_INVALID;
  hipStream_t st = (hipStream_t)stream;
  // flags bit 0: the ticket word acc[2] is already zero (a slice of the caller's pre-zeroed arena): no memset node
  if (!(flags & 1) && hipMemsetAsync(acc4, 0, 4 * sizeof(double), st) != hipSuccess) return M3D_ERR_LAUNCH;
  if (n > 0) {
    int64_t gx = m3d_cdiv(n, C <= 8 ? 1024 : 256);  // (four rows per thread in the register path)
    if (gx > CE_MAX_BLOCKS) gx = CE_MAX_BLOCKS;  // (one partial pair per workgroup, summed by the last one's 256 threads)
    hipLaunchKernelGGL(ce_fwd_kernel, dim3((unsigned)gx), dim3(256), 0, st, logits, ld, target, n, C, ignore_index, lse,
                       acc4, loss);
  } else {
    hipLaunchKernelGGL(ce_finalize_kernel, dim3(1), dim3(1), 0, st, acc4, loss);
  }
  M3D_CHECK_LAUNCH();
  return M3D_OK;
}

template <bool H>
static int ce_loss_bwd_impl(const float* logits, int64_t ld, const int64_t* target, int64_t n, int32_t C,
                            int64_t ignore_index, const float* lse, const double* acc2, const float* gout,
                            void* dlogits, void* stream) {
  if (n < 0 || C < 1) return M3D_ERR_INVALID;
  if (n == 0) return M3D_OK;
  if (!logits || !target || !lse || !acc2 || !gout || !dlogits) return M3D_ERR_INVALID;
  int64_t gx = m3d_cdiv(n * C, 256);
  if (gx > 8192) gx = 8192;
  hipLaunchKernelGGL(ce_bwd_kernel<H>, dim3((unsigned)gx), dim3(256), 0, (hipStream_t)stream, logits, ld, target, n, C,
                     ignore_index, lse, acc2, gout, dlogits);
  M3D_CHECK_LAUNCH();
  return M3D_OK;
}
extern "C" int m3d_ce_loss_bwd(const float* logits, int64_t ld, const int64_t* target, int64_t n, int32_t C,
                               int64_t ignore_index, const float* lse, const double* acc2, const float* gout,
                               float* dlogits, void* stream) {
  return ce_loss_bwd_impl<false>(logits, ld, target, n, C, ignore_index, lse, acc2, gout, dlogits, stream);
}
// ------------------------------------------------------------------------------------------
// Adam over flat buffers.  state[0] = step count (float, incremented here by a 1-thread launch so that a
// replayed hipGraph keeps counting), matching torch.optim.Adam(amsgrad=False, maximize=False):
//   g' = g + wd*p;  m = b1 m + (1-b1) g';  v = b2 v + (1-b2) g'^2
//   p -= lr / (1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
// grad_scale multiplies g on the way in (1/world_size after a SUM all-reduce); zero_grad != 0 clears g.
// ------------------------------------------------------------------------------------------
__global__ void adam_tick_kernel(float* __restrict__ state) { state[0] += 1.f; }
#define ADAM_SUBTICKETS 64

// state[0] = steps done so far (fp32), state[1 .. 1 + ADAM_TICKETS] = arrival tickets (uint32, zero between launches):
// every workgroup reads state[0] when it starts and takes a ticket when it is done; the last one stores the new count —
// the counter bump without a 1-thread launch in front of the update (a replayed graph pays ~5 us per node).  Two levels of
// tickets (workgroup b -> counter 2 + b % 64, the last arriver of each counter -> the master counter 1): ~1 100 arrivals at ONE
// address are ~15 ns apiece, one after the other (+8 us on a 10-us kernel when they all sit on the master).
__global__ __launch_bounds__(256) void adam_kernel(float4* __restrict__ p, float4* __restrict__ g,
                                                   float4* __restrict__ m, float4* __restrict__ v,
                                                   float* __restrict__ state, const float* __restrict__ lr_dev,
                                                   float lr, float b1, float b2, float eps, float wd, float gscale,
                                                   int zero_grad, int64_t n4) {
  const float t = __hip_atomic_load(&state[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1.f;
  const float lrv = lr_dev ? lr_dev[0] : lr;
  const float bc1 = 1.f - powf(b1, t), bc2 = 1.f - powf(b2, t);
  const float step = lrv / bc1, rs = 1.f / sqrtf(bc2);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    float4 pv = p[i], gv = g[i], mv = m[i], vv = v[i];
    float* pp = (float*)&pv; float* gp = (float*)&gv; float* mp = (float*)&mv; float* vp = (float*)&vv;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float gg = gp[j] * gscale + wd * pp[j];
      mp[j] = mp[j] + (1.f - b1) * (gg - mp[j]);
      vp[j] = b2 * vp[j] + (1.f - b2) * gg * gg;
      pp[j] -= step * mp[j] / (sqrtf(vp[j]) * rs + eps);
    }
    p[i] = pv; m[i] = mv; v[i] = vv;
    if (zero_grad) g[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __syncthreads();  // every thread of this workgroup has used its copy of the count
  if (threadIdx.x == 0) {
    unsigned* tk = (unsigned*)&state[1];
    const unsigned sub = blockIdx.x % ADAM_SUBTICKETS;
    const unsigned members = gridDim.x / ADAM_SUBTICKETS + (sub < gridDim.x % ADAM_SUBTICKETS ? 1u : 0u);
    if (atomicAdd(&tk[1 + sub], 1u) == members - 1) {
      tk[1 + sub] = 0u;
      const unsigned groups = gridDim.x < ADAM_SUBTICKETS ? gridDim.x : ADAM_SUBTICKETS;
      if (atomicAdd(&tk[0], 1u) == groups - 1) {
        __hip_atomic_store(&state[0], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        tk[0] = 0u;
      }
    }
  }
}

extern "C" int m3d_adam_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq, float* state,
                             const float* lr_dev, float lr, float beta1, float beta2, float eps, float weight_decay,
                             float grad_scale, int32_t zero_grad, int64_t n, void* stream) {
  if (n < 0 || (n & 3)) return M3D_ERR_INVALID;  // flat buffers are padded to a multiple of 4 floats
  if (!state) return M3D_ERR_INVALID;
  if (n > 0 && (!params || !grads || !exp_avg || !exp_avg_sq)) return M3D_ERR_INVALID;
  if ((((uintptr_t)params) | ((uintptr_t)grads) | ((uintptr_t)exp_avg) | ((uintptr_t)exp_avg_sq)) & 15)
    return M3D_ERR_INVALID;
  hipStream_t st = (hipStream_t)stream;
  if (n == 0) hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(1), 0, st, state);
  if (n > 0) {
    int64_t gx = m3d_cdiv(n / 4, 256);
    if (gx > 2048) gx = 2048;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)gx), dim3(256), 0, st, (float4*)params, (float4*)grads,
                       (float4*)exp_avg, (float4*)exp_avg_sq, state, lr_dev, lr, beta1, beta2, eps, weight_decay,
                       grad_scale, zero_grad, n / 4);
  }
  M3D_CHECK_LAUNCH();
  return M3D_OK;
}

// ------------------------------------------------------------------------------------------
// start of a training step: zero-fill of the step's accumulation arena (one buffer: the dx targets of the LFA backward
// atomics, the row scatter-add outputs, the statistics slots) and the "+ 1" of the BatchNorm step counters
// (torch.nn.BatchNorm1d.num_batches_tracked, one int64 per layer) in ONE launch.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void zero_bump_kernel(float4* __restrict__ buf, int64_t n16, int64_t* __restrict__ counters,
                                                        int ncounters) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256)
    buf[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (blockIdx.x == 0)
    for (int c = threadIdx.x; c < ncounters; c += 256) counters[c] += 1;
}

extern "C" int m3d_zero_bump(void* buf, int64_t nbytes, int64_t* counters, int32_t ncounters, void* stream) {
  if (nbytes < 0 || (nbytes & 15) || ncounters < 0) return M3D_ERR_INVALID;
  if ((nbytes > 0 && (!buf || (((uintptr_t)buf) & 15))) || (ncounters > 0 && !counters)) return M3D_ERR_INVALID;
  if (nbytes == 0 && ncounters == 0) return M3D_OK;
  int64_t gx = m3d_cdiv(nbytes / 16, 256 * 4);
  if (gx > 4096) gx = 4096;
  if (gx < 1) gx = 1;
  hipLaunchKernelGGL(zero_bump_kernel, dim3((unsigned)gx), dim3(256), 0, (hipStream_t)stream, (float4*)buf, nbytes / 16,
                     counters, ncounters);
  M3D_CHECK_LAUNCH();
  return M3D_OK;
}
