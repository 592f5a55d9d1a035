/* m3d_hip.h — C ABI of libm3d_hip.so: the MI355X (gfx950) kernels behind the Myria3D RandLA-Net hot path.
 *
 * The reference (IGNF/myria3d) is pure Python and has no FFI of its own: everything native on this path is
 * reached through the third-party wheels torch_cluster / torch_scatter / torch_geometric.
 * Each entry point below therefore cites the reference call site whose arithmetic it replaces
 * (paths relative to /root/reference).  The Python side (myria3d_amd/_lib.py) binds these with ctypes and
 * passes `tensor.data_ptr()` + `torch.cuda.current_stream().cuda_stream`; see INTEGRATION.md for the stub a
 * Myria3D maintainer would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless stated otherwise; the caller (torch) owns every buffer; no entry
 *     point allocates user-visible memory (scratch comes in through `ws` arguments sized by *_workspace_bytes)
 *   - everything is enqueued on `stream` (a hipStream_t passed as void*); nothing synchronises the device
 *   - return value: 0 = ok, -1 = invalid argument, -2 = unsupported shape, -3 = launch failure; never throws
 *   - re-entrant, no global mutable state
 *   - feature matrices are row-major fp32; neighbour tables are dense int32 [n, k] with -1 padding where a
 *     cloud has fewer than k points (the reference's edge lists simply have fewer edges there)
 *   - `ptr` arrays are the PyG Batch.ptr vectors: int64 [num_clouds + 1], on the device
 */
#ifndef M3D_HIP_H
#define M3D_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* torch.nn.Dropout(p) fused into a SharedMLP layer's BatchNorm kernels (mlp_classif, pyg_randla_net.py:49-52): the mask
 * of m3d_dropout — a hash of (seed, counter[0], element) — applied to the layer's output in the forward pass and to the
 * incoming gradient in the backward pass.  A null pointer, a null counter or p == 0 mean "no dropout".  rows (nullable,
 * device int32 [M]): row r of the layer's tensors is row rows[r] of the caller's tensor (the net's cell-sorted order) — the
 * mask is then a function of the CALLER's element, independent of that order. */
/* snapshot (nullable, device int64): FORWARD launches copy *counter there; the backward launches of the same layer are then
 * handed counter = snapshot, so that a second train-mode forward between a forward and its backward (loss = f(a) + f(b), a
 * no_grad recalibration pass), which advances the live counter, cannot change the mask the backward pass rebuilds. */
typedef struct { const int64_t* counter; uint64_t seed; float p; const int32_t* rows; int64_t* snapshot; } M3DDropout;

/* ---- bf16 ACTIVATION STORAGE (round 6; BASELINE config 2 "bf16"; the reference's switch is Lightning's `precision`,
 * configs/experiment/RandLaNet_base_run_FR-2x3GPUs.yaml:12).  Entry points whose `flags` / `act` / `accumulate` argument
 * documents "M3D_IO_*" accept these bits:
 *   M3D_IO_BF16  every ACTIVATION matrix of the call — the [rows, channels] feature tensors and their gradients: a0, a1, c, c1,
 *                z, y, dy, dz, dx, x, out, src, ... — holds bf16 (2-byte elements, round-to-nearest-even on store) although the
 *                prototype says `float*`; leading dimensions stay in ELEMENTS, pointers 16-byte aligned as before.  Weights,
 *                biases, BatchNorm vectors, statistics, positions, index tables and parameter gradients stay fp32 / fp64; all
 *                arithmetic and accumulation stays fp32 (fp64 statistics).
 *   M3D_IO_A32   (with M3D_IO_BF16) ... except the FIRST input matrix of the call (a0 / dy / dout), which is fp32: an incoming
 *                gradient that was accumulated with float atomics (the LFA layers' dx at ch >= 32)
 *   M3D_IO_C32   (with M3D_IO_BF16) ... except the OUTPUT matrix, which is fp32 (the logits)
 * Which argument carries the bits (everything else about the call is unchanged):
 *   m3d_gemm_f32 `act` (all three bits; bf16 storage exists in the fragment-direct kernels: -2 for shapes only the LDS-tiled
 *   fallback covers), m3d_gemm_pair_f32 `flags`, m3d_gemm_bn_on_load_f32 `pro->act`, m3d_bn_dgrad_f32 `flags` (BF16, A32),
 *   m3d_linear_wgrad_f32 / m3d_linear_wgrad_batch `accumulate` (BF16: dz, x0, x1 of every job), m3d_bn_stats_apply / m3d_bn_apply
 *   `act`, m3d_bn_bwd `accumulate_param_grads` bits 16-17 (= (BF16 | A32) >> 12 << 16: its bits 8-15 hold the slot count),
 *   m3d_lfa_fwd / m3d_lfa_fwd_bf16 / m3d_lfa_bwd / m3d_lfa_bwd_bf16 / m3d_lfa_bwd_edge_rows `flags` (x, out, dout and the edge
 *   rows; the atomically accumulated dx stays fp32), m3d_scatter_add_rows `flags` (src; out too with distinct targets),
 *   m3d_gather_sum_rows `accumulate` (src and out); own entry points: m3d_gather_rows_bf16, m3d_colsum_bf16,
 *   m3d_convert_f32_bf16. */
#define M3D_IO_BF16 0x1000
#define M3D_IO_A32 0x2000
#define M3D_IO_C32 0x4000

#define M3D_ABI_VERSION 18
#define M3D_ADAM_STATE_WORDS 66
#define M3D_CE_ACC_DOUBLES 516
int m3d_abi_version(void);

/* count <= 48 device-to-device copies (dst[i] <- src[i], bytes[i] bytes; 16-byte aligned pointers) in ONE launch;
 * dst / src / bytes are HOST arrays read at call time (the pointers travel as kernel arguments). */
int m3d_copy_many(void* const* dst, const void* const* src, const int64_t* bytes, int32_t count, void* stream);

/* *id_out <- id of the hipGraph capture `stream` is part of, 0 when it is not capturing (host pointer). */
int m3d_stream_capture_id(void* stream, uint64_t* id_out);

/* ---- k nearest neighbours -----------------------------------------------------------------------------
 * torch_cluster.knn as called by knn_graph(pos, K, batch, loop=True)
 * (myria3d/models/modules/pyg_randla_net.py:180), knn_interpolate(k=1) (pyg_randla_net.py:250) and
 * knn_interpolate(k=10) (myria3d/models/model.py:90-98).  Exact; squared-L2 within each cloud; self included;
 * ascending by (d2, source index).  k <= 100, the bound the upstream CUDA kernel asserts (k > 64: two passes of the
 * search, the 64 nearest and then the next k - 64 in the same total order).  No run-time knobs: the library reads no
 * environment variable. */
size_t m3d_knn_workspace_bytes(int64_t n_src, int32_t num_clouds);
/* byte offset, inside a built workspace, of the cell-sorted arrays: which = 0: float4 [n_src] (x, y, z, bits of the
 * original row); 1: perm int32 [n_src] (sorted slot -> original row); 2: inv int32 [n_src] (original row -> slot).
 * Cell-sorted order is the spatially coherent order the net works in internally. */
size_t m3d_knn_workspace_offset(int64_t n_src, int32_t num_clouds, int32_t which);
/* builds the per-cloud search grid over the SOURCE points into ws */
int m3d_knn_build(const float* pos_src, int32_t pos_stride /* floats per row, >= 3 */, const int64_t* ptr_src,
                  int32_t num_clouds, int64_t n_src, void* ws, void* stream);
/* m3d_knn_build that also carries one int32 per source row into cell-sorted order: map_out[slot] = map_in[row]
 * (both NULL: plain build).  Composes a level's decimation map with the next level's order in the same launch. */
int m3d_knn_build_map(const float* pos_src, int32_t pos_stride, const int64_t* ptr_src, int32_t num_clouds,
                      int64_t n_src, void* ws, const int32_t* map_in, int32_t* map_out, void* stream);
/* queries: either pos_qry rows (row q -> idx_out[q]) or, if qry_ws != NULL, the points of another built
 * workspace in its cell-sorted order (wave-coherent; each record carries its original row).  Self-kNN =
 * qry_ws == ws.  d2_out may be NULL.  flags bit 0 (sorted_io; needs qry_ws): output row = the query's cell-sorted slot
 * and neighbour ids = cell-sorted slots of the source workspace (selection and tie-breaking stay on original rows).
 * flags bits 1-2: which of the two bit-identical search kernels runs — 0: chosen by size (deferred insertion from
 * 2^20 (query, neighbour) pairs), 1: deferred insertion (used for 4 < k <= 64), 2: direct insertion; for parity tests
 * and A/B timing (the library has no other switch).
 * flags bits 8-15 (round 6): BACKGROUND launch — at most that many x 64 wavefronts (0: as many as the queries fill); every
 * wavefront then walks several groups of queries.  Same tables; for a launch that shares the chip with another stream. */
int m3d_knn_query(const void* ws, const int64_t* ptr_src, int64_t n_src, int32_t num_clouds, const float* pos_qry,
                  int32_t qry_stride, const void* qry_ws, const int64_t* ptr_qry, int64_t n_qry, int32_t k,
                  int32_t flags, int32_t* idx_out /* [n_qry, k] */, float* d2_out /* [n_qry, k] or NULL */,
                  void* stream);
/* Up to 8 independent queries in ONE launch: job j searches the grid ws[j] (n_src[j] sources) for the cell-sorted queries
 * of qry_ws[j] (n_qry[j]; may be ws[j] itself) and writes idx_out[j][n_qry[j], k].  Same k (<= 64), num_clouds and
 * flags (as m3d_knn_query) for every job, no distances, bit-identical to njobs calls of m3d_knn_query.  The pointer arrays are HOST arrays (read
 * during the call).  Used for the four K-NN tables / the four decoder 1-NN tables of a forward pass
 * (pyg_randla_net.py:180, :250), whose deep-level launches are otherwise latency-bound one after the other. */
int m3d_knn_query_batch(int32_t njobs, const void* const* ws, const int64_t* const* ptr_src, const int64_t* n_src,
                        const void* const* qry_ws, const int64_t* const* ptr_qry, const int64_t* n_qry,
                        int32_t num_clouds, int32_t k, int32_t flags, int32_t* const* idx_out, void* stream);

/* ---- SharedMLP GEMM -------------------------------------------------------------------------------------
 * Linear of PyG MLP / torch.nn.Linear (pyg_randla_net.py:42,53,97-109), forward, dgrad and wgrad:
 *   C[M,N] (+)= [A0[rows] | A1][M, k0+k1] * B[N, k0+k1]^T  (+ bias) -> *scale + shift -> LeakyReLU
 * a_colmajor / b_colmajor: element (r,k) of that operand lives at p[k*ld + r].  a0_rows: optional int32 row
 * gather on A0 (FPModule's x[nn], pyg_randla_net.py:250-251).  stat_part: optional fp64
 * [stat_parts][2][N] buffer receiving per-workgroup partial column sums / sums of squares of the raw (pre
 * scale/shift) output for train-mode BatchNorm (fully overwritten, no zero-fill needed; summed by
 * m3d_bn_finalize); stat_parts must equal m3d_gemm_stat_parts(M, N, k0 + k1).  stat_parts = -S (S > 0): SLOT MODE —
 * stat_part is a PRE-ZEROED [S][2][N] table, workgroup w adds its partials to slot w % S (consumer: m3d_bn_stats_apply).
 * act: bit 0 = LeakyReLU(slope) on the output; bit 8 = bf16 matrix-core operands (both operands rounded to bf16 as
 * the fragments are built, v_mfma_f32_16x16x32_bf16, fp32 accumulate / epilogue / storage) — honoured by the K > 64
 * kernels when K % 32 == 0, ignored elsewhere (the K <= 64 layers are HBM-bound).
 * accumulate != 0: add into C (atomically when splitk > 1, which splits the K dimension).  With the affine epilogue
 * (scale / shift / act) accumulate is a RESIDUAL (round 6): C = act(scale * (A B^T + bias) + shift + C_old) — the tail of a
 * DilatedResidualBlock in eval mode (pyg_randla_net.py:186-187) over the buffer the shortcut's GEMM has written. */
int m3d_gemm_stat_parts(int64_t M, int32_t N, int32_t K);
int m3d_gemm_f32(const float* a0, int64_t lda0, int32_t a_colmajor, const int32_t* a0_rows, int32_t k0,
                 const float* a1, int64_t lda1, int32_t k1, const float* b, int64_t ldb, int32_t b_colmajor,
                 int64_t M, int32_t N, const float* bias, const float* scale, const float* shift, int32_t act,
                 float slope, double* stat_part, int32_t stat_parts, float* c, int64_t ldc, int32_t accumulate,
                 int32_t splitk, void* stream);
/* The SharedMLP layer BEHIND another one, fused with that layer's train-mode BatchNorm + LeakyReLU (pyg_randla_net.py:97-109):
 *   y = lrelu((z - mean) * invstd * gamma + beta)   (z [M, k0]: the raw Linear output of the layer in front; statistics from ITS
 *                                                     slot table `slots` [nslots][2][k0], as m3d_bn_stats_apply derives them)
 *   C[M, N] = y B[N, k0]^T + bias,  column sums / sums of squares of C into the pre-zeroed slot table stat_part [stat_slots][2][N]
 * One launch instead of m3d_bn_stats_apply + m3d_gemm_f32, one pass over the activation less.  The launch also writes what the
 * backward pass of the layer in front needs (scale / shift / mean / invstd [k0]), updates its running statistics, and stores
 * y (nullable) for this layer's weight-gradient GEMM.  k0 <= 64 and k0 % 4 == 0 (the row-stream kernel: the 204 800 / 51 200-row
 * layers); M3D_ERR_UNSUPPORTED otherwise (callers then use the two launches). */
typedef struct {
  const double* slots; int32_t nslots; int64_t count;
  const float* gamma; const float* beta; float eps; float momentum;
  float* running_mean; float* running_var;               /* nullable: updated in place */
  float* scale; float* shift; float* mean; float* invstd; /* [k0] outputs */
  int32_t act; float slope;
  float* y;                                               /* [M, k0] output (nullable) */
} M3DBnOnLoad;
int m3d_gemm_bn_on_load_f32(const M3DBnOnLoad* pro, const float* z, int32_t k0, const float* b, int64_t ldb, int64_t M,
                            int32_t N, const float* bias, double* stat_part, int32_t stat_slots, float* c, int64_t ldc,
                            void* stream);
/* two products with ONE output shape, C_i[M, N] (+)= A_i[M, k_i] B_i^T (+ bias_i), i = 0, 1, as one launch when the
 * fragment-direct k-loop kernel takes both (k_i > 64, 16-byte aligned rows, same tile plan), as two m3d_gemm_f32 launches
 * otherwise: the mlp2 / shortcut Linears of a DilatedResidualBlock (pyg_randla_net.py:172-188) and their input
 * gradients on the deep levels.  Every array argument has two entries.  stat_part (nullable; entries nullable together):
 * slot-mode statistics tables as in m3d_gemm_f32 (stat_parts < 0).  flags: bit 0 = column-major B_i (the dgrad pattern),
 * bit 8 = bf16 matrix-core operands. */
int m3d_gemm_pair_f32(const float* const* a, const int64_t* lda, const int32_t* k, const float* const* b,
                      const int64_t* ldb, int64_t M, int32_t N, const float* const* bias, double* const* stat_part,
                      int32_t stat_parts, float* const* c, const int64_t* ldc, const int32_t* accumulate, int32_t flags,
                      void* stream);
/* Linear weight gradient  dW[N, k0+k1] (+)= dZ[M,N]^T [X0[x0_rows] | X1]   (autograd transpose of the Linear in
 * SharedMLP / FPModule, pyg_randla_net.py:97-109,249-252).  The rows are split over workgroups; the splits meet in
 * the workspace (m3d_linear_wgrad_workspace_bytes(M, N, k0+k1) bytes, may be 0), not in same-address atomics.
 * accumulate != 0: add into dW (a gradient sink) instead of overwriting it. */
size_t m3d_linear_wgrad_workspace_bytes(int64_t M, int32_t N, int32_t K);
int m3d_linear_wgrad_f32(const float* dz, int64_t lddz, const float* x0, int64_t ldx0, const int32_t* x0_rows,
                         int32_t k0, const float* x1, int64_t ldx1, int32_t k1, int64_t M, int32_t N, float* dw,
                         int64_t lddw, int32_t accumulate, void* ws, void* stream);
/* The weight gradients of several layers in a handful of launches (one per tile class + one for the partial sums): job j
 * is m3d_linear_wgrad_f32(dz[j], ..., dw[j], lddw[j], 1, ws[j]).  accumulate: bit 0 must be set (gradient sinks: the flat
 * gradient buffer); bit 8 = the matrix-bound jobs (16 x 16-tile groups of 8 or more per wave) take bf16 operands on
 * v_mfma_f32_16x16x32_bf16 with fp32 accumulation; host arrays of length njobs.  The backward pass of the network hands all 29 Linear layers
 * over at its end (torch autograd would run each Linear.backward's weight product where it stands). */
int m3d_linear_wgrad_batch(int32_t njobs, const float* const* dz, const int64_t* lddz, const float* const* x0,
                           const int64_t* ldx0, const int32_t* const* x0_rows, const int32_t* k0, const float* const* x1,
                           const int64_t* ldx1, const int32_t* k1, const int64_t* M, const int32_t* N, float* const* dw,
                           const int64_t* lddw, int32_t accumulate, void* const* ws, void* stream);
/* out[N] += column sums of x[M,N] (bias gradient of a Linear without BatchNorm: fc0, fc_classif) */
int m3d_colsum_f32(const float* x, int64_t ld, int64_t M, int32_t N, float* out, void* stream);
/* the same for a bf16 matrix (M3D_IO_BF16 storage): the bias gradient of fc0 / fc_classif from a bf16 incoming gradient */
int m3d_colsum_bf16(const void* x, int64_t ld, int64_t M, int32_t N, float* out, void* stream);

/* ---- BatchNorm1d(momentum=0.01, eps=1e-6) of SharedMLP (pyg_randla_net.py:92-109) ------------------------ */
int m3d_bn_finalize(const double* stat_part /* [parts][2][N] from m3d_gemm_f32 */, int32_t parts, int64_t count,
                    const float* gamma, const float* beta,
                    float eps, float momentum, float* running_mean /* updated in place, may be NULL */,
                    float* running_var, float* scale, float* shift, float* mean_out, float* invstd_out, int32_t N,
                    void* stream);
int m3d_bn_fold_eval(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                     float eps, float* scale, float* shift, int32_t N, void* stream);
/* y = act(z*scale+shift [+ z2*scale2+shift2]); the second term is the residual of DilatedResidualBlock
 * (pyg_randla_net.py:186-187).  N % 4 == 0, contiguous rows. */
int m3d_bn_apply(const float* z, const float* scale, const float* shift, const float* z2, const float* scale2,
                 const float* shift2, int32_t act, float slope, float* y, int64_t M, int32_t N, void* stream);

/* m3d_bn_finalize + m3d_bn_apply in ONE launch, fed by SLOT-MODE statistics: m3d_gemm_f32 called with
 * stat_parts = -nslots adds every workgroup's column partials into slot (workgroup % nslots) of a PRE-ZEROED fp64 table
 * stat_part [nslots][2][N] (fp64 atomics), and each thread here sums the slots of its own four columns.  Writes scale /
 * shift / mean / invstd (inputs of m3d_bn_bwd) and updates the running statistics exactly like m3d_bn_finalize.  The
 * second argument block (slots2 ... z2, all optional) is the residual branch of m3d_bn_apply.  N: a power of two >= 4.
 * (Summation order over workgroups is not fixed in this mode: the fp64 sums may differ in their last bits from run
 * to run; the classic partial-row mode of m3d_gemm_f32 / m3d_bn_finalize is bitwise reproducible.) */
int m3d_bn_stats_apply(const double* slots, int32_t nslots, int64_t count, const float* gamma, const float* beta,
                       float eps, float momentum, float* running_mean, float* running_var, float* scale, float* shift,
                       float* mean_out, float* invstd_out, const float* z, const double* slots2, const float* gamma2,
                       const float* beta2, float* running_mean2, float* running_var2, float* scale2, float* shift2,
                       float* mean_out2, float* invstd_out2, const float* z2, int32_t act, float slope, float* y,
                       int64_t M, int32_t N, const M3DDropout* drop, void* stream);
/* backward of m3d_bn_apply in train mode: dz (and dz2), dgamma/dbeta (and dgamma2/dbeta2).
 * sums_ws: m3d_bn_bwd_workspace_bytes(M, N) bytes of scratch (per-block partial column sums, no atomics).
 * accumulate_param_grads: bit 0 = dgamma/dbeta are gradient sinks (e.g. slices of the flat gradient buffer) and are added
 * to instead of overwritten; bits 8..15 = nslots > 0 selects SLOT MODE: sums_ws is then a PRE-ZEROED fp64 table
 * [nslots][3][N] the reduce pass adds to (fp64 atomics) and the apply pass sums per column — two launches instead of
 * three (no finalize kernel); N must be a power of two.  Bit 1 (slot mode, single BatchNorm only) = pass 1 ONLY: the
 * slot table is filled and nothing else is written (dz may be NULL); pass 2 is then the prologue of m3d_bn_dgrad_f32. */
size_t m3d_bn_bwd_workspace_bytes(int64_t M, int32_t N);
int m3d_bn_bwd(const float* dy, const float* z, const float* scale, const float* shift, const float* mean,
               const float* invstd, const float* z2, const float* scale2, const float* shift2, const float* mean2,
               const float* invstd2, int32_t act, float slope, int64_t M, int32_t N, double* sums_ws, float* dz,
               float* dz2, float* dgamma, float* dbeta, float* dgamma2, float* dbeta2,
               int32_t accumulate_param_grads, const M3DDropout* drop, void* stream);
/* Pass 2 of the BatchNorm backward fused into the input-gradient GEMM of the Linear in front of it (SharedMLP layer,
 * pyg_randla_net.py:97-109: torch autograd runs BatchNorm1d.backward, then Linear.backward):
 *   dz = scale * (dy * act'(z*scale+shift) - s1/M - (z-mean)*invstd * s2/M)   computed as the GEMM's A fragments are loaded
 *   dx[M, Kin] = dz[M, N] w[N, Kin]     dz is also stored (the weight-gradient GEMM reads it), dbeta = s1, dgamma = s2
 * sums = the [nslots][3][N] table left by m3d_bn_bwd(..., accumulate_param_grads = 2 | nslots << 8).
 * flags: bit 0 = dgamma / dbeta are gradient sinks (added to); bit 8 = bf16 matrix-core operands (N % 32 == 0, N > 64);
 * bit 9 = the input gradient is ADDED to dx (and dx1) — another consumer of the same tensor wrote its gradient there.
 * N % 4 == 0, N <= 1024, contiguous dy / z / dz; M3D_ERR_UNSUPPORTED otherwise (callers then use m3d_bn_bwd + m3d_gemm_f32).
 * dx_split > 0 (the layer's input was a concatenation, FPModule: pyg_randla_net.py:249-252): columns [0, dx_split) of the
 * input gradient are stored to dx[m][.] (lddx), columns [dx_split, Kin) to dx1[m][. - dx_split] (lddx1); multiples of 4. */
int m3d_bn_dgrad_f32(const float* dy, const float* z, const float* scale, const float* shift, const float* mean,
                     const float* invstd, int32_t act, float slope, const double* sums, int32_t nslots, int64_t M,
                     int32_t N, const float* w, int64_t ldw, int32_t Kin, float* dx, int64_t lddx, float* dz,
                     float* dgamma, float* dbeta, int32_t flags, int32_t dx_split, float* dx1, int64_t lddx1,
                     const M3DDropout* drop, void* stream);

/* ---- rows: decimation / upsampling gathers (pyg_randla_net.py:192-238, :250) ---------------------------- */
int m3d_gather_rows(const float* src, int64_t ld, const int32_t* idx /* NULL = identity */, float* out, int64_t m,
                    int32_t C, void* stream);
/* the same for bf16 rows (M3D_IO_BF16 storage: decimate() and the un-permutation of bf16 feature matrices); ld in elements */
int m3d_gather_rows_bf16(const void* src, int64_t ld, const int32_t* idx, void* out, int64_t m, int32_t C, void* stream);
/* dst[i] = bf16(src[i]) (round-to-nearest-even) for n contiguous fp32 values: the network INPUT features of a net with
 * M3D_IO_BF16 activation storage (x[sum N, F] once per batch; the reference's input is fp32, pyg_randla_net.py:55-58) */
int m3d_convert_f32_bf16(const float* src, void* dst, int64_t n, void* stream);
/* out[idx[i]][0..C) += src[i][0..C) (the transpose of m3d_gather_rows; negative ids are skipped).  flags bit 0: the ids are
 * DISTINCT (the transpose of a subset selection, decimate(): pyg_randla_net.py:234-238): plain 16-byte read-modify-writes
 * instead of float atomics (C % 4 == 0; otherwise the atomic kernel runs). */
int m3d_scatter_add_rows(const float* src, const int32_t* idx, float* out, int64_t ldo, int64_t m, int32_t C,
                         int32_t flags, void* stream);

/* CSR inverse of many-to-one row maps idx_j[n_j] -> [0, m_j) (the 1-NN tables of knn_interpolate(k = 1),
 * pyg_randla_net.py:250): rows f with idx_j[f] == c are inv_j[ptr_j[c] .. ptr_j[c + 1]) (in no particular order; negative
 * or out-of-range entries are left out).  njobs <= 8 maps in three launches; the job arrays are HOST arrays.  cnt_j: int32
 * [m_j] workspace that must be ZERO on entry and is zero again afterwards; ptr_j: [m_j + 1]; inv_j: [n_j]. */
int m3d_csr_invert_batch(int32_t njobs, const int32_t* const* idx, const int64_t* n, const int64_t* m,
                         int32_t* const* cnt, int32_t* const* ptr, int32_t* const* inv, void* stream);
/* out[c][0..C) (+)= sum over f in inv[ptr[c] .. ptr[c + 1]) of src[f][0..C): the transpose of m3d_gather_rows(idx) through
 * the CSR inverse of idx — no atomics, no zero fill (accumulate bit 0: added to what out holds; bit 1: the lists are long
 * (~16 rows, m3d_knn_reverse): four lanes share a list).  inv == NULL: list c is rows ptr[c] .. ptr[c + 1] of src themselves
 * (m3d_lfa_bwd_edge_rows stores them that way).  C % 4 == 0. */
int m3d_gather_sum_rows(const float* src, int64_t lds, const int32_t* ptr, const int32_t* inv, float* out, int64_t ldo,
                        int64_t m, int32_t C, int32_t accumulate, void* stream);
/* Reverse neighbour lists of a K-NN table idx [n, K] (entries outside [0, n) are left out): edge e = i * K + k with
 * idx[i][k] == j is one of inv[ptr[j] .. ptr[j + 1]) (in no particular order); ptr: [n + 1], inv: [n * K].  With
 * m3d_gather_sum_rows(accumulate | 2) this is the scatter-free backward of every gather x[idx] (m3d_lfa_bwd flags bit 5). */
size_t m3d_knn_reverse_workspace_bytes(int64_t n, int32_t K);
int m3d_knn_reverse(const int32_t* idx, int64_t n, int32_t K, int32_t* ptr, int32_t* inv /* NULL: not wanted */,
                    int32_t* slot /* NULL or [n * K]: slot[e] = position of edge e in inv (-1: in no list) */, void* ws, void* stream);
int m3d_pad_pos(const float* pos, int32_t stride, float* out4 /* [n,4] */, int64_t n, void* stream);
/* decimation_indices(): slot r of cloud b <- ptr[b] + P_b(r), P_b a keyed pseudo-random permutation of
 * [0, n_b); ptr_out is the decimated ptr (computed by the caller: max(1, n_b // factor) per cloud);
 * seed: device uint64[1].  A cloud may ask for MORE slots than it has points (MinimumNumNodes,
 * myria3d/pctl/transforms/transforms.py:66-84): slots n_b.. continue with further independent permutations. */
int m3d_decimation_indices(const int64_t* ptr, const int64_t* ptr_out, int32_t num_clouds, const uint64_t* seed,
                           uint32_t level, int32_t* idx_out, int64_t m, void* stream);
/* decimate() of one level (pyg_randla_net.py:234-238) in one launch: slot t of the next level draws d_ref_out[t] exactly
 * as m3d_decimation_indices does (or takes d_ref_in[t] when given: then ptr / ptr_out / seed / d_ref_out may be NULL),
 * d_int_out[t] = inv[d_ref] (this level's cell-sorted slot) and pos4_out[t] = pos4[d_int] (16-byte records). */
int m3d_decimate_level(const int64_t* ptr, const int64_t* ptr_out, int32_t num_clouds, const uint64_t* seed,
                       uint32_t level, const int32_t* d_ref_in, const int32_t* inv, const float* pos4,
                       int32_t* d_ref_out, int32_t* d_int_out, float* pos4_out, int64_t m, void* stream);

/* ---- Local spatial encoding + attentive pooling (pyg_randla_net.py:112-152) ----------------------------- */
/* first/second moments of the 10-vector r over all valid edges: mom65 = [sum r (10) | upper triangle of
 * sum r r^T (55)], fp64, zeroed inside */
int m3d_lfa_moments(const float* pos4, const int32_t* idx, int64_t n, int32_t K, double* mom65, void* stream);
/* m3d_lfa_moments for up to 8 levels in one launch: job j writes mom[j * mom_stride .. + 65) (mom_stride >= 65 doubles;
 * the whole [njobs][mom_stride] block is zeroed by the call); host pointer arrays. */
int m3d_lfa_moments_batch(int32_t njobs, const float* const* pos4, const int32_t* const* idx, const int64_t* n, int32_t K,
                          double* mom, int64_t mom_stride, void* stream);
/* folds mlp_encoder's BatchNorm into its Linear.  mom65 != NULL: train mode (batch statistics over
 * num_edges edges, running stats updated); mom65 == NULL: eval mode (running stats). */
int m3d_lfa_enc_finalize(const double* mom65, int64_t num_edges, const float* w /* [D,10] */, const float* b,
                         const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                         float* running_var, float* w_folded, float* b_folded, float* mean_out, float* invstd_out,
                         int32_t D, void* stream);

/* m3d_lfa_enc_finalize + m3d_lfa_pack_att (bf16 = 0: fp32 fragments of W_att in `packed`, of W_att^T in `packed_t`,
 * each max(CH,16)^2 floats) or + m3d_lfa_pack_att_bf16 (bf16 != 0: CH^2 bf16 each, CH % 32 == 0) in ONE launch. */
int m3d_lfa_prepare(const double* mom65, int64_t num_edges, const float* w, const float* b, const float* gamma,
                    const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                    float* w_folded, float* b_folded, float* mean_out, float* invstd_out, int32_t D,
                    const float* w_att, int32_t CH, void* packed, void* packed_t, int32_t bf16, void* stream);
/* m3d_lfa_prepare (train mode: mom65 given) of up to 8 LFA layers in ONE launch; HOST arrays of length njobs.  eps / momentum
 * are shared (every encoder BatchNorm of the net has the same, pyg_randla_net.py:94). */
int m3d_lfa_prepare_batch(int32_t njobs, const double* const* mom65, const int64_t* num_edges, const float* const* w,
                          const float* const* b, const float* const* gamma, const float* const* beta, float eps,
                          float momentum, float* const* running_mean, float* const* running_var, float* const* w_folded,
                          float* const* b_folded, float* const* mean_out, float* const* invstd_out, const int32_t* D,
                          const float* const* w_att, const int32_t* CH, void* const* packed, void* const* packed_t,
                          const int32_t* bf16, void* stream);


/* bf16 matrix-core variant of m3d_lfa_fwd (CH in {32, 64, 128, 256}): the attention GEMM takes bf16 operands
 * (v_mfma_f32_16x16x32_bf16, fp32 accumulate; BASELINE config 2's "bf16"), everything else stays fp32.
 * att_w_packed_bf16: m3d_lfa_pack_att_bf16 / m3d_lfa_prepare(bf16 = 1) output `packed`. */
int m3d_lfa_fwd_bf16(const float* x, const float* pos4, const int32_t* idx, int64_t n, int32_t K, int32_t CH,
                     const float* enc_w_folded, const float* enc_b_folded, const void* att_w_packed_bf16, float slope,
                     float* out, int32_t flags /* as m3d_lfa_fwd */, void* stream);
/* W_att [CH, CH] fp32 -> bf16 operand fragments [CH/16][CH/32][64 lanes][8]:
 * packed: element i of lane l = W[16 nt + (l & 15)][32 ks + 8 (l >> 4) + i]; packed_t (optional): the same of W^T. */
int m3d_lfa_pack_att_bf16(const float* w, int32_t CH, void* packed, void* packed_t, void* stream);
/* fused forward: out[n, CH] = sum_k softmax_k(W_att f_k) * f_k,  f_k = [x[j_k] | LeakyReLU(wf r_k + bf)].
 * CH in {8,16,32,64,128,256}, K <= 32.  att_w_packed: W_att ([CH,CH] row-major, zero-padded to
 * CHP = max(CH,16)) re-laid as [CHP/16][CHP/16][64 lanes][4]:
 *   packed[nt][s4][lane][i] = W[16*nt + (lane & 15)][4*(4*s4 + i) + (lane >> 4)]. */
/* packs W_att (and, if packed_t != NULL, W_att^T) into that order: max(CH,16)^2 floats each */
int m3d_lfa_pack_att(const float* w /* [CH, CH] row-major */, int32_t CH, float* packed, float* packed_t, void* stream);
/* flags bit 0 (M3D_LFA_FULL): the caller PROMISES that every entry of idx is a valid row (no -1 padding: every cloud of the
 * level has at least K points).  With K in {16, 32} the launch then takes the mask-free kernel (round 5; two centres per MFMA
 * tile at CH = 8); without the promise, or for other K, the general kernel.  Same results up to fp32 rounding of the
 * softmax exponent and the edge length (1 ulp). */
#define M3D_LFA_FULL 1
int m3d_lfa_fwd(const float* x /* [n, CH/2] */, const float* pos4, const int32_t* idx, int64_t n, int32_t K,
                int32_t CH, const float* enc_w_folded, const float* enc_b_folded, const float* att_w_packed,
                float slope, float* out, int32_t flags, void* stream);
/* fused backward of m3d_lfa_fwd (recomputes the forward tile-by-tile; nothing of size [E,.] in HBM):
 *   dx[n, CH/2]  += d/dx           (atomically accumulated: zero it first)
 *   dw_att[CH,CH] = d/dW_att,  G[CH/2][11] (fp64) = sum_e dy_e [r_e | 1]  (input of m3d_lfa_enc_bwd_finalize)
 * att_wt_packed: W_att^T packed like att_w_packed.  ws: m3d_lfa_bwd_workspace_bytes(n, K, CH) bytes of scratch. */
size_t m3d_lfa_bwd_workspace_bytes(int64_t n, int32_t K, int32_t CH);
int m3d_lfa_bwd(const float* x, const float* pos4, const int32_t* idx, int64_t n, int32_t K, int32_t CH,
                const float* enc_w_folded, const float* enc_b_folded, const float* att_w_packed,
                const float* att_wt_packed, float slope, const float* dout, float* dx, float* dw_att,
                int32_t flags /* bit 0: add into dw_att instead of overwriting; bit 1: G is already zero; bit 2: leave the
                                 workgroup partials in ws (m3d_lfa_bwd_reduce_batch sums them later); bit 3: complete
                                 neighbourhoods promised (as M3D_LFA_FULL of m3d_lfa_fwd: the mask-free kernel); bit 5
                                 (with bit 3, where m3d_lfa_bwd_edge_rows_ok says 1): `dx` is an [n * K, D] EDGE-row buffer
                                 — row i * K + k = the gradient that edge (centre i, neighbour k) sends to x[idx[i][k]] —
                                 plainly stored: no atomics; the caller sums the rows of every point's reverse neighbour
                                 list (m3d_csr_invert_batch of idx + m3d_gather_sum_rows) */, double* G,
                void* ws, void* stream);
/* m3d_lfa_bwd(flags | 8 | 32) with the edge rows in REVERSE-LIST order: edge (i, k) is stored in row edge_slot[i * K + k] of
 * dx_edges (the slot table of m3d_knn_reverse): a point's contributions are contiguous rows, summed by
 * m3d_gather_sum_rows(ptr, inv = NULL).  edge_slot == NULL: row i * K + k. */
int m3d_lfa_bwd_edge_rows(const float* x, const float* pos4, const int32_t* idx, int64_t n, int32_t K, int32_t CH,
                          const float* enc_w_folded, const float* enc_b_folded, const float* att_w_packed,
                          const float* att_wt_packed, float slope, const float* dout, float* dx_edges,
                          const int32_t* edge_slot, float* dw_att, int32_t flags, double* G, void* ws, void* stream);
/* 1: flags bit 5 of m3d_lfa_bwd is honoured for this layer (CH in {8, 16}, K = 16, 32-bit offsets); 0: M3D_ERR_UNSUPPORTED */
int m3d_lfa_bwd_edge_rows_ok(int64_t n, int32_t K, int32_t CH, float slope);
/* bf16 matrix-core variant (CH in {64, 128, 256}): the recomputed attention logits, dF and dW_att GEMMs take bf16
 * operands (fp32 accumulate); att_w*_packed_bf16 from m3d_lfa_pack_att_bf16 / m3d_lfa_prepare(bf16 = 1). */
/* flags bit 4 (with bit 3): SPLIT-bf16 operands ("bf16x3": every operand x = hi + lo, hi = bf16(x), lo = bf16(x - hi); the
 * products run as hi*hi + hi*lo + lo*hi on the bf16 matrix cores with fp32 accumulation: ~2^-16 relative operand error
 * instead of 2^-8).  att_w*_packed_bf16 then hold the CH*CH hi values followed by the CH*CH lo values
 * (m3d_lfa_prepare(bf16 = 2)).  The same for m3d_lfa_fwd_bf16: flags bit 1. */
int m3d_lfa_bwd_bf16(const float* x, const float* pos4, const int32_t* idx, int64_t n, int32_t K, int32_t CH,
                     const float* enc_w_folded, const float* enc_b_folded, const void* att_w_packed_bf16,
                     const void* att_wt_packed_bf16, float slope, const float* dout, float* dx, float* dw_att,
                     int32_t flags, double* G, void* ws, void* stream);
/* unfused pieces (fallback for K > 32, cross-check of the fused kernels): F[n*K, CH] edge features */
int m3d_lfa_edge_features(const float* x, const float* pos4, const int32_t* idx, int64_t n, int32_t K, int32_t CH,
                          const float* enc_w_folded, const float* enc_b_folded, float slope, float* F, void* stream);
int m3d_lfa_edge_softmax_fwd(const float* A, const float* F, const int32_t* idx, int64_t n, int32_t K, int32_t CH,
                             float* out, void* stream);
/* A (attention logits) is overwritten with dA; dF <- dout * softmax */
int m3d_lfa_edge_softmax_bwd(float* A_inout_dA, const float* F, const int32_t* idx, int64_t n, int32_t K, int32_t CH,
                             const float* dout, float* dF, void* stream);
/* dx[n, CH/2] += scatter of dF[:, :CH/2];  G[CH/2][11] (fp64, zeroed inside) = sum_e dy_e [r_e | 1] */
int m3d_lfa_edge_features_bwd(const float* dF, const float* pos4, const int32_t* idx, int64_t n, int32_t K,
                              int32_t CH, const float* enc_w_folded, const float* enc_b_folded, float slope,
                              float* dx, double* G, void* stream);
int m3d_lfa_enc_bwd_finalize(const double* G, const double* mom65, int64_t num_edges, const float* w, const float* b,
                             const float* gamma, const float* mean, const float* invstd, float* dw, float* db,
                             float* dgamma, float* dbeta, int32_t D, int32_t accumulate, void* stream);
/* m3d_lfa_enc_bwd_finalize for up to any number of layers in one launch per 16 (host arrays of length njobs). */
int m3d_lfa_enc_bwd_finalize_batch(int32_t njobs, const double* const* G, const double* const* mom65,
                                   const int64_t* num_edges, const float* const* w, const float* const* b,
                                   const float* const* gamma, const float* const* mean, const float* const* invstd,
                                   float* const* dw, float* const* db, float* const* dgamma, float* const* dbeta,
                                   const int32_t* D, int32_t accumulate, void* stream);
/* The partial-sum reduces of m3d_lfa_bwd / m3d_lfa_bwd_bf16 calls made with flags bit 2 set, in one launch per 16 layers:
 * job j is the layer launched with (n[j], K[j], CH[j], ws[j]); its sums are ADDED into dw_att[j] (gradient sink) and G[j]
 * (pre-zeroed).  dW_att and G feed parameter gradients only, so the whole backward pass can hand them over at its end. */
int m3d_lfa_bwd_reduce_batch(int32_t njobs, const int64_t* n, const int32_t* K, const int32_t* CH, void* const* ws,
                             float* const* dw_att, double* const* G, void* stream);

/* ---- knn_interpolate arithmetic (model.py:90-98; pyg_randla_net.py:250) ---------------------------------- */
int m3d_idw_interpolate_fwd(const float* x, int64_t ldx, const int32_t* idx, const float* d2, int64_t n_qry,
                            int32_t k, int32_t C, float* y, void* stream);

/* ---- merged predictions (Interpolator.reduce_predictions_and_save, myria3d/models/interpolation.py:142-169) ----
 * for i < m:  row = logits[idx ? idx[i] : i];  probas[i, :C] = softmax(row);  preds[i] = argmax(row) (first maximum);
 * entropy[i] = torch.distributions.Categorical(probs=probas[i]).entropy().  probas / preds / entropy may be NULL.
 * C <= 64. */
int m3d_predict_reduce(const float* logits, int64_t ld, const int32_t* idx, int64_t m, int32_t C, float* probas,
                       int64_t ldp, int32_t* preds, float* entropy, void* stream);

/* ---- data preparation (next row after the net): torch_geometric.transforms.GridSampling(size) as configured in
 * configs/datamodule/transforms/preparations/points_budget.yaml:14-17 — voxel_grid + consecutive_cluster + per-voxel
 * mean of pos / x and majority vote of y — for all tiles of a batch at once, every tile with its own bounding box
 * (= the reference's per-tile application).  Outputs have capacity n rows; the number of voxels of tile b is
 * out_ptr[b+1] - out_ptr[b] (device int64 [num_clouds + 1]; the caller reads out_ptr[num_clouds] to slice).
 * x / y (and out_x / out_y) may be NULL.  Rows come out in ascending voxel id per tile, like consecutive_cluster. */
size_t m3d_grid_sampling_workspace_bytes(int64_t n, int32_t num_clouds);
int m3d_grid_sampling(const float* pos, int32_t pos_stride, const float* x, int64_t ldx, int32_t F, const int64_t* y,
                      const int64_t* ptr, int32_t num_clouds, int64_t n, float size, void* ws, float* out_pos,
                      float* out_x, int64_t* out_y, int64_t* out_ptr, void* stream);
/* status_dev_out: device int32 [1] <- 0 ok | 1 some tile's voxel grid has >= 2^40 cells (results invalid) */
int m3d_grid_sampling_status(const void* ws, int64_t n, int32_t num_clouds, int32_t* status_dev_out, void* stream);

/* Tiling of a whole cloud into square samples: the selection of split_cloud_into_samples()
 * (myria3d/pctl/dataset/utils.py:126-158: cKDTree.query_ball_point(centre, r = subtile_width // 2, p = inf) per centre
 * of get_mosaic_of_centers(), utils.py:29-39) for ALL centres at once.  centers_dev: fp64 [centers_per_axis] lattice
 * coordinates (identical for x and y); sample s = ix * centers_per_axis + iy (the reference's x-major order); a point
 * belongs to sample s iff |x - xmin - c[ix]| <= radius and |y - ymin - c[iy]| <= radius, evaluated like the reference
 * (float32 shift by the per-cloud minimum, float64 compare).  Two calls: count_only = 1 fills sample_ptr (int64
 * [S + 1], CSR offsets; sample_ptr[S] = total memberships, or -1 when some point lies in more than 64 samples —
 * an overlap beyond what the kernel lists), count_only = 0 writes idx_out (int32 [total], point
 * indices, ascending inside each sample; the reference returns them in tree order — the set is the contract).
 * start / step: first lattice coordinate and lattice pitch (candidate pre-selection only). */
size_t m3d_tile_select_workspace_bytes(int64_t n, int32_t centers_per_axis);
int m3d_tile_select(const float* pos, int32_t pos_stride, int64_t n, const double* centers_dev,
                    int32_t centers_per_axis, double radius, double start, double step, void* ws, int32_t count_only,
                    int64_t* sample_ptr, int32_t* idx_out, void* stream);

/* Per-tile normalisations, in place, for a whole batch (two launches): Center (pos -= mean), NullifyLowestZ
 * (z -= min z; myria3d/pctl/transforms/transforms.py:141-146), NormalizePos (pos *= pos_scale; :149-162) and
 * StandardizeRGBAndIntensity (:115-138) on columns intensity_col (log(v + 1) first) and rgb_col of x (-1 = skip):
 * s = std_unbiased + 1e-6 (1 if NaN), v <- clamp((v - mean) / s, -clamp_sigma * s, clamp_sigma * s).
 * stats_ws: fp64 [num_clouds][8] scratch. */
int m3d_tile_normalize(float* pos, int32_t pos_stride, float* x, int64_t ldx, int32_t intensity_col, int32_t rgb_col,
                       const int64_t* ptr, int32_t num_clouds, int64_t n, int32_t center, int32_t nullify_z,
                       float pos_scale, float clamp_sigma, double* stats_ws, void* stream);

/* ---- PointNet++ set-abstraction variant (BASELINE.json configs[4]; north_star "random/FPS subsampling") ----------
 * No reference implementation exists (myria3d/models/model.py:12: MODEL_ZOO = [PyGRandLANet]; no FPS anywhere in the
 * repository): the entry points restate the published operators the variant is built from (PointNet++ as packaged by
 * PyG: torch_cluster.fps, PointNetConv's gathers and aggr="max"); oracle/pointnet2_oracle.py is the checker.
 *
 * m3d_fps: farthest-point sampling inside each cloud (torch_cluster.fps semantics, random_start=False unless `start` is
 * given): cloud b keeps ptr_out[b+1] - ptr_out[b] points; slot 0 is point start[b] (cloud-relative; NULL: point 0), slot
 * s + 1 the point with the largest distance to slots 0..s (d2 = (dx*dx + dy*dy) + dz*dz in fp32 without FMA contraction,
 * ties -> smaller index).  pos4: [n, 4] rows (x, y, z, -), 16-byte aligned.  idx_out: global rows, selection order.
 * max_points: the largest cloud (host value; <= 65 536). */
int m3d_fps(const float* pos4, const int64_t* ptr_src, const int64_t* ptr_out, int32_t num_clouds, int64_t max_points,
            const int32_t* start, int32_t* idx_out, void* stream);
/* The same sampling with EXACT bucket skipping (csrc/sa.hip: fps_bucket_kernel): sorted_ws = the BUILT kNN workspace of the
 * same points and ptr_src (m3d_knn_build; n_src points in all) — its cell-sorted records are cut into buckets of 64 whose
 * bounding boxes let an iteration skip every bucket the new point cannot reach (2x faster at 40 000 points per cloud; no
 * gain below ~16 000, where m3d_fps keeps every position in registers).  max_points <= 40 000 (the running minima of
 * a cloud live in LDS).  idx_out: original global rows, selection order; identical to m3d_fps. */
int m3d_fps_sorted(const void* sorted_ws, int64_t n_src, const int64_t* ptr_src, const int64_t* ptr_out, int32_t num_clouds,
                   int64_t max_points, const int32_t* start, int32_t* idx_out, void* stream);
/* Edge rows of a set-abstraction level over a COMPACT edge list: centre i owns edges seg[i] .. seg[i+1] - 1 (at most K;
 * fewer when its cloud has fewer than K points, as in PyG's edge_index), edge seg[i] + k has source j = nbr[i][k]:
 *   out[e][0..C) = x[j],  out[e][C..C+3) = pos_src[j] - pos_ctr[i],  out[e][C+3..ldo) = 0;  esrc[e] = j, ectr[e] = i. */
int m3d_sa_group(const float* x, int64_t ldx, int32_t C, const float* pos4_src, const float* pos4_ctr,
                 const int32_t* nbr /* [m, K] */, const int64_t* seg /* [m + 1] */, int64_t m, int32_t K, float* out,
                 int64_t ldo, int32_t* esrc, int32_t* ectr, void* stream);
/* transpose of the x_j gather: dx[esrc[e]][c] += de[e][c], c < C (dx zeroed or pre-loaded by the caller) */
int m3d_sa_group_bwd(const float* de, int64_t ld, const int32_t* esrc, int64_t E, int32_t C, float* dx, int64_t lddx,
                     void* stream);
/* aggr="max": out[i][c] = max over the edges of centre i of y[e][c]; arg[i][c] = k of the first maximal edge */
int m3d_seg_max(const float* y, int64_t ldy, const int64_t* seg, int64_t m, int32_t C, float* out, int32_t* arg,
                void* stream);
/* dy[e][c] = dout[ectr[e]][c] when e is that centre's arg-max edge for channel c, else 0 (dy [E, C] contiguous) */
int m3d_seg_max_bwd(const float* dout, const int32_t* arg, const int64_t* seg, const int32_t* ectr, int64_t E, int32_t C,
                    float* dy, void* stream);

/* ---- training step: loss and optimizer -------------------------------------------------------------------
 * torch.nn.CrossEntropyLoss(ignore_index=65, reduction="mean") on the logits (myria3d/models/model.py:118,
 * configs/model/criterion/CrossEntropyLoss.yaml:1-3).  lse: [n] scratch kept for the backward; acc4: fp64
 * [M3D_CE_ACC_DOUBLES] ([0] sum of row losses, [1] number of non-ignored rows — both kept for the backward —, [2]
 * arrival ticket of the workgroups, [4 ...] their partial sums; the first four are zeroed inside unless flags bit 0 says
 * the caller passes zeros); loss: fp32 [1], written by the last workgroup to arrive. */
int m3d_ce_loss_fwd(const float* logits, int64_t ld, const int64_t* target, int64_t n, int32_t C,
                    int64_t ignore_index, float* lse, double* acc4, float* loss, int32_t flags, void* stream);
/* dlogits[n, C] (contiguous) = gout[0] * d loss / d logits */
int m3d_ce_loss_bwd(const float* logits, int64_t ld, const int64_t* target, int64_t n, int32_t C,
                    int64_t ignore_index, const float* lse, const double* acc4, const float* gout, float* dlogits,
                    void* stream);
/* torch.optim.Adam (configs/model/optimizer/Adam.yaml:1-4; lr from configs/model/pyg_randla_net_model.yaml:4) in ONE
 * launch over flat, 16-byte aligned fp32 buffers of n (multiple of 4) elements.  state: [M3D_ADAM_STATE_WORDS] x 4
 * bytes on the device: [0] fp32 step counter (incremented inside, so a replayed hipGraph keeps counting), the rest
 * uint32 arrival tickets of the update's workgroups (zero before the first call; left zero).  lr_dev: optional device fp32 [1]
 * overriding `lr` (schedulers under graph replay).  grad_scale multiplies the gradient on the way in (1/world
 * after a SUM all-reduce); zero_grad != 0 clears the gradient buffer after it has been consumed. */
int m3d_adam_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq, float* state, const float* lr_dev,
                  float lr, float beta1, float beta2, float eps, float weight_decay, float grad_scale,
                  int32_t zero_grad, int64_t n, void* stream);
/* torch.nn.Dropout(p), train mode (mlp_classif: pyg_randla_net.py:49-52): y[i] = x[i] * keep(i) / (1 - p'), keep(i) a hash
 * of (seed, counter[0], i) with P(keep) = 1 - p' = 1 - p rounded to 2^-16; n a multiple of 4, 16-byte aligned buffers
 * (y may be x).  The same call with dy in place of x is the backward pass (the mask is recomputed, not stored);
 * counter: a device int64 the caller advances once per training forward (graph-replay safe). */
int m3d_dropout(const float* x, float* y, int64_t n, float p, const int64_t* counter, uint64_t seed, void* stream);
/* start of a training step: zero-fill of `nbytes` (multiple of 16, 16-byte aligned: the step's accumulation arena) and
 * counters[0 .. ncounters) += 1 (the BatchNorm layers' num_batches_tracked, int64) in one launch */
int m3d_zero_bump(void* buf, int64_t nbytes, int64_t* counters, int32_t ncounters, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* M3D_HIP_H */
