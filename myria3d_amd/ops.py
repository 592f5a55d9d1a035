"""Thin Python wrappers (and autograd glue) over the C ABI of ``libm3d_hip.so``.

Every function here launches hand-written HIP kernels on the current torch stream; torch is used only to own
device memory, to provide the stream and for autograd bookkeeping.  Reference call sites are cited per op
(paths relative to ``/root/reference``).
"""
from __future__ import annotations

import os
from typing import List, Optional, Tuple

import torch
from torch import Tensor

from ._lib import call, lib

LRELU_SLOPE = 0.2  # myria3d/models/modules/pyg_randla_net.py:92
BN_MOMENTUM = 0.01  # pyg_randla_net.py:94
BN_EPS = 1e-6  # pyg_randla_net.py:94


def _p(t: Optional[Tensor]):
    return None if t is None else t.data_ptr()


# raw handle of the current HIP stream: torch.cuda.current_stream().cuda_stream builds two Python objects per call, and
# every kernel launch asks (~280 per training step, host-bound when launched eagerly)
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def _st():
    if _raw_stream is not None:
        return _raw_stream(_cur_device())
    return torch.cuda.current_stream().cuda_stream


def capture_id(stream) -> int:
    """Id of the hipGraph capture ``stream`` is part of, 0 when it is not capturing."""
    import ctypes

    out = ctypes.c_uint64(0)
    call("m3d_stream_capture_id", stream.cuda_stream, ctypes.byref(out))
    return int(out.value)


class ZeroArena:
    """ONE zero-filled device buffer per training step, cut into the accumulation targets of the backward pass (the dx
    of every LFA layer, the outputs of the row scatter-adds, the fp64 encoder sums): a replayed hipGraph pays 5-9 us per
    tiny fill / memset node and the step had ~45 of them.  The size is learned from the previous step (the first step
    falls back to individual ``torch.zeros``)."""

    def __init__(self):
        self.buf: Optional[Tensor] = None
        self.off = 0
        self.grew = 0
        self.need = 0
        self.active = False

    def begin(self, device, bump: Optional[Tensor] = None) -> None:
        """``bump``: int64 counters incremented by the same launch (the BatchNorm layers' ``num_batches_tracked``)."""
        self._learn()
        self.active = True
        if torch.device(device).type != "cuda":  # (bookkeeping exercised on the host by tests/test_host.py; no kernels there)
            self.buf = torch.zeros(self.need, dtype=torch.uint8, device=device) if self.need else None
            if bump is not None:
                bump += 1
        elif self.need or bump is not None:
            self.buf = torch.empty(self.need, dtype=torch.uint8, device=device) if self.need else None
            call("m3d_zero_bump", _p(self.buf), self.need, _p(bump), 0 if bump is None else bump.numel(), _st())
        else:
            self.buf = None

    def stop(self) -> None:
        self._learn()  # (an eval pass between two training steps must not forget what the last step needed)
        self.active = False
        self.buf = None

    def _learn(self) -> None:
        if self.active:
            self.need = max(self.need, self.off + self.grew)
        self.off = self.grew = 0

    def zeros(self, shape, dtype, device) -> Tensor:
        numel = 1
        for d in shape:
            numel *= int(d)
        nbytes = numel * torch.empty((), dtype=dtype).element_size()
        span = (nbytes + 255) // 256 * 256
        if self.active and self.buf is not None and self.buf.device == device and self.off + span <= self.buf.numel():
            v = self.buf[self.off:self.off + nbytes].view(dtype).view(*shape)
            self.off += span
            return v
        if self.active:
            self.grew += span
        return torch.zeros(shape, dtype=dtype, device=device)


arena = ZeroArena()


def copy_many(dsts, srcs) -> None:
    """``dst.copy_(src)`` for up to 48 contiguous same-size device buffers in ONE launch (``m3d_copy_many``)."""
    import ctypes

    n = len(dsts)
    assert n == len(srcs)
    for i0 in range(0, n, 48):
        d, s_ = dsts[i0:i0 + 48], srcs[i0:i0 + 48]
        m = len(d)
        for a, b in zip(d, s_):
            assert a.is_contiguous() and b.is_contiguous() and a.numel() * a.element_size() == b.numel() * b.element_size()
        dp = (ctypes.c_void_p * m)(*[t.data_ptr() for t in d])
        sp = (ctypes.c_void_p * m)(*[t.data_ptr() for t in s_])
        nb = (ctypes.c_int64 * m)(*[t.numel() * t.element_size() for t in d])
        call("m3d_copy_many", dp, sp, nb, m, _st())


def _chk(t: Tensor, dtype=torch.float32):
    assert t.is_cuda and t.dtype == dtype and t.is_contiguous(), (t.device, t.dtype, t.is_contiguous())
    return t


# --------------------------------------------------------------------------------------------------
# bf16 ACTIVATION STORAGE (round 6; BASELINE config 2 "bf16", the reference's switch: Lightning's ``precision``,
# configs/experiment/RandLaNet_base_run_FR-2x3GPUs.yaml:12).  Feature matrices ``[rows, channels]`` and their gradients may be
# torch.bfloat16 tensors; the wrappers below read the layout off the tensors' dtypes and pass the M3D_IO_* bits of
# include/m3d_hip.h.  Parameters, statistics, positions, logits and parameter gradients stay fp32.
# --------------------------------------------------------------------------------------------------
IO_BF16, IO_A32, IO_C32 = 0x1000, 0x2000, 0x4000  # M3D_IO_BF16 / M3D_IO_A32 / M3D_IO_C32
BF16 = torch.bfloat16


def _h(t: Optional[Tensor]) -> bool:
    return t is not None and t.dtype == BF16


def _chka(t: Tensor):
    """An activation matrix: contiguous fp32 or bf16 on the device."""
    assert t.is_cuda and t.dtype in (torch.float32, BF16) and t.is_contiguous(), (t.device, t.dtype, t.is_contiguous())
    return t


def to_bf16(x: Tensor) -> Tensor:
    """fp32 -> bf16 copy through ``m3d_convert_f32_bf16`` (the net's input features; an fp32 gradient handed to a bf16 layer)."""
    x = _chk(x.contiguous())
    out = torch.empty(x.shape, dtype=BF16, device=x.device)
    call("m3d_convert_f32_bf16", _p(x), _p(out), x.numel(), _st())
    return out


# --------------------------------------------------------------------------------------------------
# kNN  (torch_cluster.knn via knn_graph / knn_interpolate: pyg_randla_net.py:180,250; model.py:90)
# --------------------------------------------------------------------------------------------------
_KNN_KERNEL = {"auto": 0, "queue": 1, "direct": 2}


class KnnIndex:
    """Per-cloud search grid over a set of source points (device workspace owned by a torch tensor)."""

    def __init__(self, pos: Tensor, ptr: Tensor, carry: Optional[Tensor] = None):
        """``carry``: an int32 value per source row; ``self.carried[slot]`` is then the value of the row that landed in
        cell-sorted slot ``slot`` (same launch)."""
        assert pos.is_cuda and pos.dtype == torch.float32 and pos.dim() == 2 and pos.stride(1) == 1
        assert ptr.is_cuda and ptr.dtype == torch.int64 and ptr.is_contiguous()
        self.n = pos.shape[0]
        self.num_clouds = ptr.numel() - 1
        self.ptr = ptr
        nbytes = lib().m3d_knn_workspace_bytes(self.n, self.num_clouds)
        self.ws = torch.empty(nbytes, dtype=torch.uint8, device=pos.device)
        self.carried = None
        if carry is not None:
            assert carry.dtype == torch.int32 and carry.is_contiguous() and carry.numel() == self.n
            self.carried = torch.empty_like(carry)
        call("m3d_knn_build_map", _p(pos), pos.stride(0), _p(ptr), self.num_clouds, self.n, _p(self.ws), _p(carry),
             _p(self.carried), _st())

    def _view(self, which: int, dtype, cols: int) -> Tensor:
        off = lib().m3d_knn_workspace_offset(self.n, self.num_clouds, which)
        nbytes = self.n * cols * 4
        return self.ws[off:off + nbytes].view(dtype).view(self.n, cols) if cols > 1 else \
            self.ws[off:off + nbytes].view(dtype)

    @property
    def sorted_pos4(self) -> Tensor:
        """float4 ``[n, 4]`` (x, y, z, bits of the original row) in cell-sorted order (a view of the workspace)."""
        return self._view(0, torch.float32, 4)

    @property
    def perm(self) -> Tensor:
        """int32 ``[n]``: cell-sorted slot -> original row."""
        return self._view(1, torch.int32, 1)

    @property
    def inv(self) -> Tensor:
        """int32 ``[n]``: original row -> cell-sorted slot."""
        return self._view(2, torch.int32, 1)

    def query(self, k: int, qry: Optional["KnnIndex"] = None, pos_qry: Optional[Tensor] = None,
              ptr_qry: Optional[Tensor] = None, want_d2: bool = False,
              sorted_io: bool = False, kernel: str = "auto", background: int = 0) -> Tuple[Tensor, Optional[Tensor]]:
        """``qry`` (another built index, possibly ``self``) gives wave-coherent cell-sorted queries;
        otherwise ``pos_qry``/``ptr_qry`` rows are queried in order.  Returns int32 ``[nq, k]`` (+ fp32 d2).
        ``sorted_io``: rows and neighbour ids are cell-sorted slots (of ``qry`` / of ``self``).  ``kernel``: "auto" (by
        size), "queue" (deferred insertion) or "direct" — bit-identical tables; parity tests and A/B timing.
        ``background`` (0 ... 255): a launch that runs BESIDE another stream's work takes at most that many x 64 wavefronts
        (flags bits 8-15 of ``m3d_knn_query``; 0: as many as the queries fill) — same tables."""
        if qry is not None:
            nq, ptr_q, qws, pq, qs = qry.n, qry.ptr, qry.ws, None, 0
            assert qry.num_clouds == self.num_clouds
        else:
            assert pos_qry is not None and ptr_qry is not None and pos_qry.stride(1) == 1 and not sorted_io
            nq, ptr_q, qws, pq, qs = pos_qry.shape[0], ptr_qry, None, pos_qry, pos_qry.stride(0)
        idx = torch.empty((nq, k), dtype=torch.int32, device=self.ws.device)
        d2 = torch.empty((nq, k), dtype=torch.float32, device=self.ws.device) if want_d2 else None
        call("m3d_knn_query", _p(self.ws), _p(self.ptr), self.n, self.num_clouds, _p(pq), qs, _p(qws), _p(ptr_q), nq, k,
             int(sorted_io) | (_KNN_KERNEL[kernel] << 1) | ((int(background) & 0xff) << 8), _p(idx), _p(d2), _st())
        return idx, d2


# --------------------------------------------------------------------------------------------------
# GEMM / BatchNorm primitives
# --------------------------------------------------------------------------------------------------
def gemm(a0: Tensor, b: Tensor, M: int, N: int, k0: int, *, lda0: Optional[int] = None, a_cm: bool = False,
         rows: Optional[Tensor] = None, a1: Optional[Tensor] = None, k1: int = 0, lda1: Optional[int] = None,
         b_cm: bool = False, ldb: Optional[int] = None, bias: Optional[Tensor] = None,
         scale: Optional[Tensor] = None, shift: Optional[Tensor] = None, act: bool = False,
         stats: Optional[Tensor] = None, out: Optional[Tensor] = None, ldc: Optional[int] = None,
         accumulate: bool = False, splitk: int = 1, stat_slots: bool = False, bf16: bool = False,
         out_dtype=None) -> Tensor:
    """C[M,N] (+)= [A0[rows] | A1] B^T, see ``m3d_gemm_f32`` in include/m3d_hip.h.  bf16 operands (``a0`` / ``a1``) give a bf16
    product unless ``out`` / ``out_dtype`` say fp32 (the logits); an fp32 ``a0`` next to bf16 storage is M3D_IO_A32."""
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype or a0.dtype, device=a0.device)
    io = 0
    if _h(a0) or _h(a1) or _h(out):
        io = IO_BF16 | (0 if _h(a0) else IO_A32) | (0 if _h(out) else IO_C32)
        assert a1 is None or _h(a1)
    lda0 = lda0 if lda0 is not None else (a0.stride(0) if not a_cm else a0.stride(0))
    lda1 = lda1 if lda1 is not None else (a1.stride(0) if a1 is not None else 0)
    ldb = ldb if ldb is not None else b.stride(0)
    ldc = ldc if ldc is not None else out.stride(0)
    call("m3d_gemm_f32", _p(a0), lda0, int(a_cm), _p(rows), k0, _p(a1), lda1, k1, _p(b), ldb, int(b_cm), M, N,
         _p(bias), _p(scale), _p(shift), int(act) | (256 if bf16 else 0) | io, LRELU_SLOPE, _p(stats),
         0 if stats is None else (-stats.shape[0] if stat_slots else stats.shape[0]),
         _p(out), ldc, int(accumulate), splitk, _st())
    return out


def gemm_pair(a, b, M: int, N: int, *, bias=None, stats=None, out=None, accumulate=(False, False), b_cm: bool = False,
              bf16: bool = False):
    """Two products with one output shape, ``C_i[M,N] (+)= A_i B_i^T`` (``b_cm``: ``A_i B_i``, the input-gradient
    pattern), as ONE launch where the k-loop kernel takes both (``m3d_gemm_pair_f32``; two launches otherwise).
    ``stats``: a pair of slot-mode tables (``stat_slots``).  Returns the two outputs."""
    import ctypes

    dev = a[0].device
    assert a[0].dtype == a[1].dtype
    outs = [o if o is not None else torch.empty((M, N), dtype=a[0].dtype, device=dev) for o in (out or (None, None))]
    assert all(o.dtype == a[0].dtype for o in outs)
    vp = lambda ts: (ctypes.c_void_p * 2)(*[_p(t) for t in ts])
    i64 = lambda vs: (ctypes.c_int64 * 2)(*vs)
    i32 = lambda vs: (ctypes.c_int32 * 2)(*vs)
    call("m3d_gemm_pair_f32", vp(a), i64([t.stride(0) for t in a]), i32([t.shape[1] for t in a]), vp(b),
         i64([t.stride(0) for t in b]), M, N, vp(bias) if bias is not None else None,
         vp(stats) if stats is not None else None, -stats[0].shape[0] if stats is not None else 0, vp(outs),
         i64([t.stride(0) for t in outs]), i32([int(x) for x in accumulate]),
         int(b_cm) | (256 if bf16 else 0) | (IO_BF16 if _h(a[0]) else 0), _st())
    return outs


def stat_buffer(M: int, N: int, K: int, device) -> Tensor:
    """fp64 ``[parts, 2, N]`` buffer for the train-mode BatchNorm statistics of a ``[M,K] x [N,K]^T`` GEMM: every
    row-workgroup of the GEMM stores its partial column sums / sums of squares (no zero-fill, no atomics) and
    ``bn_finalize`` adds the rows."""
    return torch.empty((lib().m3d_gemm_stat_parts(M, N, K), 2, N), dtype=torch.float64, device=device)


def _splitk_for(red: int, m: int, n: int) -> int:
    tiles = ((m + 63) // 64) * ((n + 63) // 64)
    return max(1, min(1024 // max(tiles, 1), red // 256))


def linear_dgrad(dz: Tensor, w: Tensor, bf16: bool = False, acc: Optional[Tensor] = None) -> Tensor:
    """dX[M,K] = dZ[M,N] W[N,K]   (W stored [out,in] like torch.nn.Linear); ``acc``: added to this buffer instead."""
    M, N = dz.shape
    K = w.shape[1]
    return gemm(dz, w, M, K, N, b_cm=True, ldb=w.stride(0), bf16=bf16, out=acc, accumulate=acc is not None)


class GradSideStream:
    """Weight-gradient kernels are leaves of the backward pass (nothing downstream reads dW before the optimizer), so
    with gradient sinks they can run on a side stream next to the dgrad / BatchNorm / LFA chain.  The stream rejoins the
    main stream at the END of every backward pass (an autograd-engine callback queued by the first kernel sent here), so
    whatever reads ``p.grad`` after ``backward()`` — gradient clipping, logging, an all-reduce, any optimizer — is
    ordered behind the weight-gradient kernels.  Operand tensors are kept alive until that join."""

    def __init__(self, device):
        self.stream = torch.cuda.Stream(device=device)
        self.keep: list = []
        self.jobs: list = []  # deferred weight-gradient GEMMs (``defer``): launched together by ``join``
        self.lfa_jobs: list = []  # deferred LFA partial-sum reduces + encoder parameter gradients (``defer_lfa``)
        self._join_queued = False

    def begin_step(self) -> None:
        """Start of a training forward.  A backward pass that raised (out of memory, an anomaly check) never ran the
        engine callback: its queued jobs — operands of a step that no longer exists — must not be flushed into the
        next step's gradients."""
        if self._join_queued:
            self.jobs.clear(), self.lfa_jobs.clear(), self.keep.clear()
            self._join_queued = False

    def _queue_join(self):
        if not self._join_queued:
            try:  # inside a backward pass: join when the engine has run its last node
                torch.autograd.Variable._execution_engine.queue_callback(self.join)
                self._join_queued = True
            except RuntimeError:  # called outside a backward pass (direct use of the op): the caller joins
                pass

    def run(self, fn, *tensors):
        main = torch.cuda.current_stream()
        self.stream.wait_stream(main)  # operands were produced on the main stream
        with torch.cuda.stream(self.stream):
            fn()
        self.keep.append(tensors)      # their memory must not be recycled by the main stream before join()
        self._queue_join()

    def defer(self, job: tuple) -> None:
        """Queue one weight-gradient GEMM ``(dz, x0, k0, rows, x1, k1, out)`` for the batched launch at the end of the
        backward pass (``m3d_linear_wgrad_batch``); the tuple keeps the operands alive until then."""
        self.jobs.append(job)
        self._queue_join()

    def defer_lfa(self, job: tuple) -> None:
        """Queue the partial-sum reduce and the encoder parameter gradients of one LFA backward:
        ``(n, K, ch, ws, dw_att, G, mom, num_edges, enc_w, enc_b, enc_gamma, mean, invstd, dw, db, dgamma, dbeta)``."""
        self.lfa_jobs.append(job)
        self._queue_join()

    def _flush_lfa(self) -> None:
        jobs, self.lfa_jobs = self.lfa_jobs, []
        if not jobs:
            return
        import ctypes

        m = len(jobs)
        vp = lambda k: (ctypes.c_void_p * m)(*[j[k].data_ptr() for j in jobs])
        call("m3d_lfa_bwd_reduce_batch", m, (ctypes.c_int64 * m)(*[j[0] for j in jobs]),
             (ctypes.c_int32 * m)(*[j[1] for j in jobs]), (ctypes.c_int32 * m)(*[j[2] for j in jobs]), vp(3), vp(4), vp(5),
             _st())
        call("m3d_lfa_enc_bwd_finalize_batch", m, vp(5), vp(6), (ctypes.c_int64 * m)(*[j[7] for j in jobs]), vp(8), vp(9),
             vp(10), vp(11), vp(12), vp(13), vp(14), vp(15), vp(16), (ctypes.c_int32 * m)(*[j[2] // 2 for j in jobs]), 1,
             _st())

    def flush(self) -> None:
        """Launch the queued weight-gradient GEMMs (current stream): a handful of launches for all layers."""
        self._flush_lfa()
        jobs, self.jobs = self.jobs, []
        if not jobs:
            return
        # the bf16 switch of m3d_linear_wgrad_batch applies to a whole call: layers that asked for bf16 matrix-core
        # operands (K > 64 in the net's "bf16" mode) and layers that did not (fc0 / fc_classif, the narrow SharedMLPs) go
        # in separate calls, so that no layer documented as fp32 is rounded
        # (likewise the activation layout: a call's jobs all read fp32, or all read bf16, dz / x0 / x1 — M3D_IO_BF16)
        for io in (0, IO_BF16):
            sel = [j for j in jobs if (IO_BF16 if _h(j[0]) else 0) == io]
            lo = [j for j in sel if not (len(j) > 7 and j[7])]
            hi = [j for j in sel if len(j) > 7 and j[7]]
            for part, flag in ((lo, 0), (hi, 256)):
                if part:
                    self._launch_wgrad_batch(part, flag | io)

    @staticmethod
    def _launch_wgrad_batch(jobs, bf16_flag: int) -> None:
        import ctypes

        m = len(jobs)
        h = lib()
        need = [h.m3d_linear_wgrad_workspace_bytes(j[0].shape[0], j[0].shape[1], j[2] + j[5]) for j in jobs]
        offs, tot = [], 0
        for nb in need:
            offs.append(tot)
            tot += (nb + 255) // 256 * 256
        ws = torch.empty(max(tot, 1), dtype=torch.uint8, device=jobs[0][0].device)
        base = ws.data_ptr()
        vp = lambda vals: (ctypes.c_void_p * m)(*vals)
        i64 = lambda vals: (ctypes.c_int64 * m)(*vals)
        i32 = lambda vals: (ctypes.c_int32 * m)(*vals)
        call("m3d_linear_wgrad_batch", m,
             vp([j[0].data_ptr() for j in jobs]), i64([j[0].stride(0) for j in jobs]),
             vp([j[1].data_ptr() for j in jobs]), i64([j[1].stride(0) for j in jobs]),
             vp([_p(j[3]) for j in jobs]), i32([j[2] for j in jobs]),
             vp([_p(j[4]) for j in jobs]), i64([j[4].stride(0) if j[4] is not None else 0 for j in jobs]),
             i32([j[5] for j in jobs]), i64([j[0].shape[0] for j in jobs]), i32([j[0].shape[1] for j in jobs]),
             vp([j[6].data_ptr() for j in jobs]), i64([j[6].stride(0) for j in jobs]),
             1 | bf16_flag,
             vp([base + o if nb else None for o, nb in zip(offs, need)]), _st())

    def join(self):
        self.flush()
        torch.cuda.current_stream().wait_stream(self.stream)
        self.keep.clear()
        self._join_queued = False


# the side stream of the net whose forward pass is being recorded; every autograd Function below copies it into its own
# ctx at forward time, so a backward pass always uses the stream of the net (and optimizer) it belongs to
_grad_side: Optional[GradSideStream] = None


def linear_wgrad(dz: Tensor, x0: Tensor, k0: int, rows: Optional[Tensor] = None, x1: Optional[Tensor] = None,
                 k1: int = 0, out: Optional[Tensor] = None, side: Optional[GradSideStream] = None,
                 bf16: bool = False) -> Optional[Tensor]:
    """dW[N, k0+k1] = dZ^T [X0[rows] | X1] (``m3d_linear_wgrad_f32``): the reduction over the M rows is split across
    waves whose partials meet in a workspace.  ``out``: a gradient sink (contiguous ``[N, k0+k1]``, e.g. a slice of the
    flat gradient buffer) that is added to; nothing is returned then (and the kernels may run on the gradient side
    stream ``side``)."""
    M, N = dz.shape
    K = k0 + k1
    sink = out is not None
    assert x0.dtype == dz.dtype and (x1 is None or x1.dtype == dz.dtype), (dz.dtype, x0.dtype)
    dw = out if sink else torch.empty((N, K), dtype=torch.float32, device=dz.device)
    if sink and side is not None and DEFER_WGRAD:
        side.defer((dz, x0, k0, rows, x1, k1, dw, bool(bf16)))  # launched with all the others at the end of the backward pass
        return None
    nbytes = lib().m3d_linear_wgrad_workspace_bytes(M, N, K)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dz.device) if nbytes else None

    def launch():
        call("m3d_linear_wgrad_f32", _p(dz), dz.stride(0), _p(x0), x0.stride(0), _p(rows), k0, _p(x1),
             x1.stride(0) if x1 is not None else 0, k1, M, N, _p(dw), dw.stride(0), int(sink) | (IO_BF16 if _h(dz) else 0),
             _p(ws), _st())

    if sink and side is not None:
        side.run(launch, dz, x0, x1, rows, ws)
    else:
        launch()
    return None if sink else dw


def colsum(x: Tensor, out: Optional[Tensor] = None) -> Optional[Tensor]:
    """Column sums; with ``out`` (a gradient sink) they are added to it and nothing is returned."""
    sink = out is not None
    if not sink:
        out = torch.zeros(x.shape[1], dtype=torch.float32, device=x.device)
    call("m3d_colsum_bf16" if _h(x) else "m3d_colsum_f32", _p(x), x.stride(0), x.shape[0], x.shape[1], _p(out), _st())
    return None if sink else out


def gather_rows(src: Tensor, idx: Optional[Tensor]) -> Tensor:
    """``src[idx]`` for a row-major fp32 matrix (decimate(): pyg_randla_net.py:234-238)."""
    m = idx.numel() if idx is not None else src.shape[0]
    out = torch.empty((m, src.shape[1]), dtype=src.dtype, device=src.device)
    call("m3d_gather_rows_bf16" if _h(src) else "m3d_gather_rows", _p(src), src.stride(0), _p(idx), _p(out), m, src.shape[1], _st())
    return out


def gather_i32(src: Tensor, idx: Tensor) -> Tensor:
    """``src[idx]`` for int32 vectors (bit copies through the row-gather kernel): composes index maps on the device."""
    assert src.dtype == torch.int32 and idx.dtype == torch.int32 and src.is_contiguous()
    out = torch.empty(idx.numel(), dtype=torch.int32, device=src.device)
    call("m3d_gather_rows", _p(src), 1, _p(idx), _p(out), idx.numel(), 1, _st())
    return out


class GradSlot:
    """Gradient meeting point of a tensor with several consumers inside the net (a block's input feeds mlp1, the shortcut
    and — decimated levels — the FP module's skip; block 1's output feeds the decimation gather and fp1's skip).
    Autograd would hand every consumer's input gradient to an AccumulateGrad add: one elementwise launch (and one
    [rows, channels] round trip through HBM) per extra consumer.  Instead the consumers' backward passes, which run in a
    fixed order (decoder first, then the block's tail, then its head), share a buffer: the first DEPOSITS its gradient
    and tells autograd "None", the following ones ADD theirs in their GEMM / scatter epilogue, the last one (``final``)
    returns the buffer as the tensor's gradient."""

    __slots__ = ("buf",)

    def __init__(self):
        self.buf: Optional[Tensor] = None

    def take(self) -> Optional[Tensor]:
        b, self.buf = self.buf, None
        return b


def scatter_add_rows(src: Tensor, idx: Tensor, n_out: int, out: Optional[Tensor] = None, distinct: bool = False) -> Tensor:
    """``out[idx[i]] += src[i]`` (``out``: an existing ``[n_out, C]`` buffer to add into; default: zeros).  ``distinct``: the
    caller guarantees that no id occurs twice (plain read-modify-writes instead of atomics)."""
    # bf16 rows: distinct targets are a plain read-modify-write of a bf16 buffer; the ATOMIC form accumulates in fp32 (the
    # caller gets an fp32 gradient: injected decimation indices, tests)
    if _h(src) and not distinct:
        assert out is None or out.dtype == torch.float32
    if out is None:
        out = arena.zeros((n_out, src.shape[1]), src.dtype if distinct else torch.float32, src.device)
    call("m3d_scatter_add_rows", _p(_chka(src)), _p(idx), _p(out), out.stride(0), src.shape[0], src.shape[1],
         int(distinct) | (IO_BF16 if _h(src) else 0), _st())
    return out


def csr_invert_batch(idxs, ms):
    """CSR inverses of the many-to-one int32 row maps ``idxs[j]: [n_j] -> [0, ms[j])`` (``m3d_csr_invert_batch``, at most 8 per
    call): a list of ``(ptr [m_j + 1], inv [n_j])`` int32 pairs for ``gather_sum_rows``."""
    import ctypes

    k = len(idxs)
    assert 0 < k <= 8 and k == len(ms)
    dev = idxs[0].device
    idxs = [_chk(t.reshape(-1), torch.int32) for t in idxs]
    cnt = torch.zeros(sum(int(m) for m in ms) + 1, dtype=torch.int32, device=dev)  # (zero again when the call has run)
    cnts, off = [], 0
    for m in ms:
        cnts.append(cnt[off:off + int(m)])
        off += int(m)
    # one allocation for the eight index vectors (host time matters when the step is launched eagerly)
    sizes = [int(m) + 1 for m in ms] + [t.numel() for t in idxs]
    buf = torch.empty(sum((n + 3) // 4 * 4 for n in sizes), dtype=torch.int32, device=dev)
    views, off = [], 0
    for n in sizes:
        views.append(buf[off:off + n])
        off += (n + 3) // 4 * 4
    ptrs, invs = views[:k], views[k:]
    vp = lambda ts: (ctypes.c_void_p * k)(*[t.data_ptr() for t in ts])
    call("m3d_csr_invert_batch", k, vp(idxs), (ctypes.c_int64 * k)(*[t.numel() for t in idxs]),
         (ctypes.c_int64 * k)(*[int(m) for m in ms]), vp(cnts), vp(ptrs), vp(invs), _st())
    return list(zip(ptrs, invs))


def gather_sum_rows(src: Tensor, ptr: Tensor, inv: Optional[Tensor], n_out: int, out: Optional[Tensor] = None,
                    long_lists: bool = False) -> Tensor:
    """``out[c] (+)= sum of src[f] over the rows f the CSR inverse ``(ptr, inv)`` lists for c``: what
    ``scatter_add_rows(src, idx, n_out)`` computes, without atomics and without a zero fill.  ``long_lists``: ~16 rows per
    list (``knn_reverse``): four lanes share a list.  ``inv=None``: list c is rows ``ptr[c] .. ptr[c + 1]`` of ``src``."""
    acc = out is not None
    if out is None:
        out = torch.empty((n_out, src.shape[1]), dtype=src.dtype, device=src.device)
    assert out.dtype == src.dtype
    call("m3d_gather_sum_rows", _p(_chka(src)), src.stride(0), _p(ptr), _p(inv) if inv is not None else None, _p(out),
         out.stride(0), n_out, src.shape[1], int(acc) | (2 if long_lists else 0) | (IO_BF16 if _h(src) else 0), _st())
    return out


def knn_reverse(idx: Tensor, with_inv: bool = True):
    """Reverse neighbour lists of a K-NN table (``m3d_knn_reverse``): ``(ptr [n + 1], inv [n K], slot [n K])`` — edge
    ``e = i * K + k`` is entry ``slot[e]`` of ``inv``, inside the list ``inv[ptr[j] : ptr[j + 1]]`` of the point ``j = idx[i][k]``
    it names (``slot[e] = -1``: padding, in no list).  ``with_inv=False``: ``inv`` is None (rows stored in list order need the
    slots only; the builder then skips its scattered writes)."""
    n, K = idx.shape
    idx = _chk(idx, torch.int32)
    o1 = (n + 1 + 3) // 4 * 4
    buf = torch.empty(o1 + (2 if with_inv else 1) * n * K, dtype=torch.int32, device=idx.device)
    ptr, slot = buf[:n + 1], buf[o1:o1 + n * K]
    inv = buf[o1 + n * K:] if with_inv else None
    ws = torch.empty(lib().m3d_knn_reverse_workspace_bytes(n, K), dtype=torch.uint8, device=idx.device)
    call("m3d_knn_reverse", _p(idx), n, K, _p(ptr), _p(inv) if with_inv else None, _p(slot), _p(ws), _st())
    return ptr, inv, slot


def pad_pos(pos: Tensor) -> Tensor:
    out = torch.empty((pos.shape[0], 4), dtype=torch.float32, device=pos.device)
    call("m3d_pad_pos", _p(pos), pos.stride(0), _p(out), pos.shape[0], _st())
    return out


def decimation_indices(ptr: Tensor, ptr_out: Tensor, m: int, seed: Tensor, level: int) -> Tensor:
    """decimation_indices() (pyg_randla_net.py:192-231) in one launch; ``seed``: device int64[1]."""
    idx = torch.empty(m, dtype=torch.int32, device=ptr.device)
    call("m3d_decimation_indices", _p(ptr), _p(ptr_out), ptr.numel() - 1, _p(seed), level, _p(idx), m, _st())
    return idx


def decimate_level(ptr: Tensor, ptr_out: Tensor, m: int, seed: Tensor, level: int, index: "KnnIndex",
                   d_ref: Optional[Tensor] = None) -> Tuple[Tensor, Tensor, Tensor]:
    """``decimate()`` of one level in one launch (``m3d_decimate_level``): returns ``(d_ref, d_int, pos4_next)`` = the
    surviving reference rows (drawn like ``decimation_indices`` unless given), their cell-sorted slots in ``index`` and
    their position records."""
    dev = ptr.device
    given = d_ref is not None
    if not given:
        d_ref = torch.empty(m, dtype=torch.int32, device=dev)
    d_int = torch.empty(m, dtype=torch.int32, device=dev)
    pos_next = torch.empty((m, 4), dtype=torch.float32, device=dev)
    call("m3d_decimate_level", _p(ptr), _p(ptr_out), ptr.numel() - 1, _p(seed), level, _p(d_ref) if given else None,
         _p(index.inv), _p(index.sorted_pos4), None if given else _p(d_ref), _p(d_int), _p(pos_next), m, _st())
    return d_ref, d_int, pos_next


def bn_fold_eval(bn: torch.nn.BatchNorm1d) -> Tuple[Tensor, Tensor]:
    n = bn.num_features
    scale = torch.empty(n, dtype=torch.float32, device=bn.weight.device)
    shift = torch.empty_like(scale)
    call("m3d_bn_fold_eval", _p(bn.weight), _p(bn.bias), _p(bn.running_mean), _p(bn.running_var), float(bn.eps),
         _p(scale), _p(shift), n, _st())
    return scale, shift


def bn_finalize(stats: Tensor, count: int, bn: torch.nn.BatchNorm1d):
    """mean/invstd/(scale, shift) from the fp64 partial column sums (``stat_buffer``); updates the running statistics
    in place."""
    if count < 2:
        raise ValueError(f"Expected more than 1 value per channel when training, got input size [{count}, {bn.num_features}]")
    n = bn.num_features
    dev = stats.device
    scale, shift, mean, invstd = (torch.empty(n, dtype=torch.float32, device=dev) for _ in range(4))
    call("m3d_bn_finalize", _p(stats), stats.shape[0], count, _p(bn.weight), _p(bn.bias),
         float(bn.eps), float(bn.momentum), _p(bn.running_mean), _p(bn.running_var), _p(scale), _p(shift), _p(mean),
         _p(invstd), n, _st())
    if not getattr(bn, "_m3d_flat_counter", False):  # flattened nets bump all counters with one add per step
        bn.num_batches_tracked += 1
    return scale, shift, mean, invstd


# weight gradients of all layers in a handful of launches at the END of the backward pass (GradSideStream.defer / flush)
# instead of one GEMM + one reduce per layer on the side stream; M3D_DEFER_WGRAD=0: the per-layer launches (A/B)
DEFER_WGRAD = os.environ.get("M3D_DEFER_WGRAD", "1") != "0"
FUSE_BN_DGRAD = os.environ.get("M3D_FUSE_BN_DGRAD", "1") != "0"  # A/B switch for bn_dgrad (see its docstring)
FUSE_MIN_ROWS = 1600
PAIR_GEMMS = os.environ.get("M3D_PAIR_GEMMS", "1") != "0"  # mlp2 / shortcut Linears of a block as one launch (A/B switch)
BN_SLOTS = 16  # slot-mode statistics: workgroups add their column partials into (at most) this many fp64 rows
# slot rows by layer size: (rows threshold, slots) pairs, first match wins; M3D_BN_SLOTS="100000:8,25000:4,0:2"
_SLOT_TABLE = tuple((int(a), int(b)) for a, b in
                    (t.split(":") for t in os.environ.get("M3D_BN_SLOTS", "100000:8,0:4").split(",")))


_BWD_SLOT_TABLE = tuple((int(a), int(b)) for a, b in
                        (t.split(":") for t in os.environ.get("M3D_BN_BWD_SLOTS", "100000:8,0:4").split(",")))


def bn_bwd_slots(M: int) -> int:
    """Slot rows of the backward column sums (filled by up to 2 048 reduce workgroups, read by the dgrad prologue)."""
    for rows, slots in _BWD_SLOT_TABLE:
        if M >= rows:
            return min(slots, BN_SLOTS)
    return 1


def bn_slots(M: int) -> int:
    """Slot rows for a layer with ``M`` rows.  Every consumer workgroup sums all slot rows of its columns (the apply
    kernels, the dgrad prologue), so fewer rows are cheaper to read; the deep levels (a few thousand rows, a few dozen
    producer workgroups) hardly contend on the atomics anyway.  4.96 -> 4.82 ms per step against 16 rows everywhere."""
    for rows, slots in _SLOT_TABLE:
        if M >= rows:
            return min(slots, BN_SLOTS)
    return 1


def _pow2(n: int) -> bool:
    return n >= 4 and (n & (n - 1)) == 0


def stat_slots(N: int, device, M: int = 1 << 30) -> Tensor:
    """Pre-zeroed fp64 ``[BN_SLOTS, 2, N]`` table for the slot-mode statistics of a GEMM (``gemm(..., stats=table,
    stat_slots=True)``), cut from the step's zero arena."""
    return arena.zeros((bn_slots(M), 2, N), torch.float64, device)


def _drop_ref(drop):
    """``drop = (p, counter, seed[, rows])`` or None -> a by-reference ``M3DDropout`` argument (None: a null pointer).
    ``rows``: int32 map from the tensors' rows to the caller's rows (the mask is a function of the caller's element)."""
    if drop is None:
        return None
    import ctypes
    from ._lib import M3DDropout

    p, counter, seed = drop[:3]
    rows = drop[3] if len(drop) > 3 else None
    snap = drop[4] if len(drop) > 4 else None  # forward launches: the counter's value is copied there (see SharedLayerTrainFn)
    return ctypes.byref(M3DDropout(counter.data_ptr(), int(seed), float(p), _p(rows), _p(snap)))


def bn_stats_apply(stats: Tensor, count: int, bn: torch.nn.BatchNorm1d, z: Tensor, act: bool, stats2=None, bn2=None,
                   z2=None, drop=None, out=None):
    """``bn_finalize`` + ``bn_apply`` in one launch (``m3d_bn_stats_apply``) from slot-mode statistics.  Returns
    ``(y, (scale, shift, mean, invstd)[, (scale2, shift2, mean2, invstd2)])``."""
    if count < 2:
        raise ValueError(f"Expected more than 1 value per channel when training, got input size [{count}, {bn.num_features}]")
    n = bn.num_features
    dev = z.device
    # (one allocation for the 4 (+4) per-column vectors: widths are multiples of 4, so every row stays 16-byte aligned)
    if out is not None:  # (y, (scale, shift, mean, invstd)) allocated by the caller: PendingBN.materialize
        y, p1 = out
        p2 = (None,) * 4
    else:
        pv = torch.empty((8 if bn2 is not None else 4, n), dtype=torch.float32, device=dev).unbind(0)
        p1 = pv[:4]
        p2 = pv[4:] if bn2 is not None else (None,) * 4
        y = torch.empty_like(z)
    assert y.dtype == z.dtype and (z2 is None or z2.dtype == z.dtype)
    call("m3d_bn_stats_apply", _p(stats), stats.shape[0], count, _p(bn.weight), _p(bn.bias), float(bn.eps),
         float(bn.momentum), _p(bn.running_mean), _p(bn.running_var), _p(p1[0]), _p(p1[1]), _p(p1[2]), _p(p1[3]),
         _p(_chka(z)), _p(stats2), _p(bn2.weight if bn2 is not None else None),
         _p(bn2.bias if bn2 is not None else None), _p(bn2.running_mean if bn2 is not None else None),
         _p(bn2.running_var if bn2 is not None else None), _p(p2[0]), _p(p2[1]), _p(p2[2]), _p(p2[3]), _p(z2),
         int(act) | (IO_BF16 if _h(z) else 0), LRELU_SLOPE, _p(y), z.shape[0], z.shape[1], _drop_ref(drop), _st())
    for b in (bn, bn2):
        if b is not None and not getattr(b, "_m3d_flat_counter", False):  # flattened nets bump all counters at once
            b.num_batches_tracked += 1
    return (y, p1, p2) if bn2 is not None else (y, p1)


# --------------------------------------------------------------------------------------------------
# BatchNorm apply-on-load (round 5): a SharedMLP layer whose only consumer is the next SharedMLP layer's GEMM (levels 1-2:
# the row-stream kernel) does not launch m3d_bn_stats_apply; the consumer's GEMM derives scale / shift from the slot
# statistics, applies BatchNorm + LeakyReLU to its A fragments as they are loaded and stores the activation on the way
# (m3d_gemm_bn_on_load_f32).  The producer hands over ``y`` — allocated, NOT yet written — tagged with a PendingBN.
# --------------------------------------------------------------------------------------------------
BN_ON_LOAD = os.environ.get("M3D_BN_ON_LOAD", "1") != "0"  # A/B switch
# eval mode: the residual tail of a block as the affine + residual epilogue of mlp2's GEMM instead of a bn_apply launch (A/B switch)
EVAL_RESIDUAL_EPILOGUE = os.environ.get("M3D_EVAL_RESIDUAL", "1") != "0"


class PendingBN:
    """The not-yet-applied BatchNorm (+ LeakyReLU) of a SharedMLP layer: raw output ``z``, slot statistics, the module, and
    the buffers the apply would have written (``y`` and the four per-column vectors the backward pass reads)."""

    __slots__ = ("z", "stats", "count", "bn", "act", "y", "vecs", "done")

    def __init__(self, z, stats, count, bn, act, y, vecs):
        self.z, self.stats, self.count, self.bn, self.act, self.y, self.vecs, self.done = z, stats, count, bn, act, y, vecs, False

    def materialize(self) -> Tensor:
        """The unfused launch after all (a consumer the fused GEMM does not cover)."""
        if not self.done:
            bn_stats_apply(self.stats, self.count, self.bn, self.z, self.act, out=(self.y, self.vecs))
            self.done = True
        return self.y


_pending: List[PendingBN] = []  # created during the current forward and not consumed yet (checked at its end)
_last_pending: Optional[PendingBN] = None


def take_pending(t: Optional[Tensor]) -> Optional[PendingBN]:
    """The PendingBN whose (unwritten) activation buffer ``t`` is, or None."""
    if t is None or not _pending:
        return None
    for p in _pending:
        if p.y is t or (p.y.data_ptr() == t.data_ptr() and p.y.shape == t.shape):
            return p
    return None


def _settle(p: PendingBN) -> None:
    p.materialize()
    if p in _pending:
        _pending.remove(p)


def settle_pending() -> None:
    """End of a forward: an activation nobody consumed through a fused GEMM is written by the plain launch."""
    while _pending:
        _pending.pop().materialize()


def drop_pending() -> None:
    """Start of a forward: entries a forward that RAISED left behind (out of memory, a one-row BatchNorm) are forgotten, not
    materialised — that would push an extra momentum update of the old batch's statistics into the running buffers, on
    whatever net and stream happen to be current."""
    global _last_pending
    _pending.clear()
    _last_pending = None


def gemm_bn_on_load(pend: PendingBN, w: Tensor, M: int, N: int, bias: Optional[Tensor], stats: Tensor) -> Optional[Tensor]:
    """``C = lrelu(BN(pend.z)) W^T + bias`` with slot-mode statistics into ``stats``; also writes ``pend.y`` and
    ``pend.vecs`` and updates the running statistics.  None when the shape is not covered (nothing was launched)."""
    import ctypes
    from ._lib import M3DBnOnLoad

    bn = pend.bn
    k0 = pend.z.shape[1]
    if pend.count < 2:
        raise ValueError(f"Expected more than 1 value per channel when training, got input size [{pend.count}, {k0}]")
    if k0 > 64 or k0 % 4 or not pend.z.is_contiguous() or w.shape[1] != k0:
        return None
    out = torch.empty((M, N), dtype=pend.z.dtype, device=w.device)
    sc, sh, mu, isd = pend.vecs
    pro = M3DBnOnLoad(_p(pend.stats), pend.stats.shape[0], pend.count, _p(bn.weight), _p(bn.bias), float(bn.eps),
                      float(bn.momentum), _p(bn.running_mean), _p(bn.running_var), _p(sc), _p(sh), _p(mu), _p(isd),
                      int(pend.act) | (IO_BF16 if _h(pend.z) else 0), LRELU_SLOPE, _p(pend.y))
    rc = lib().m3d_gemm_bn_on_load_f32(ctypes.byref(pro), _p(pend.z), k0, _p(w), w.stride(0), M, N, _p(bias), _p(stats),
                                       stats.shape[0], _p(out), out.stride(0), _st())
    if rc == -2:
        return None
    if rc != 0:
        raise RuntimeError(f"m3d_gemm_bn_on_load_f32 failed with status {rc}")
    if not getattr(bn, "_m3d_flat_counter", False):
        bn.num_batches_tracked += 1
    pend.done = True
    if pend in _pending:
        _pending.remove(pend)
    return out


def bn_apply(z: Tensor, scale: Tensor, shift: Tensor, act: bool, z2: Optional[Tensor] = None,
             scale2: Optional[Tensor] = None, shift2: Optional[Tensor] = None) -> Tensor:
    y = torch.empty_like(z)
    assert z2 is None or z2.dtype == z.dtype
    call("m3d_bn_apply", _p(_chka(z)), _p(scale), _p(shift), _p(z2), _p(scale2), _p(shift2),
         int(act) | (IO_BF16 if _h(z) else 0), LRELU_SLOPE, _p(y), z.shape[0], z.shape[1], _st())
    return y


def bn_bwd(dy, z, scale, shift, mean, invstd, act, z2=None, scale2=None, shift2=None, mean2=None, invstd2=None,
           sinks=None, drop=None):
    """``sinks = (dgamma, dbeta[, dgamma2, dbeta2])``: gradient sinks that are added to (None is returned for them).
    Power-of-two widths take the slot mode of ``m3d_bn_bwd`` (two launches, pre-zeroed sums from the zero arena)."""
    M, N = z.shape
    dev = z.device
    slots = bn_bwd_slots(M) if _pow2(N) else 0
    if slots:
        sums = arena.zeros((slots, 3, N), torch.float64, dev)
    else:
        sums = torch.empty(lib().m3d_bn_bwd_workspace_bytes(M, N) // 8, dtype=torch.float64, device=dev)
    io = _bn_io(dy, z)
    assert z2 is None or z2.dtype == z.dtype
    dz = torch.empty_like(z)
    dz2 = dgamma2 = dbeta2 = None
    if sinks is not None:
        dgamma, dbeta = sinks[0], sinks[1]
        if z2 is not None:
            dgamma2, dbeta2 = sinks[2], sinks[3]
    else:
        dgamma, dbeta = torch.empty(N, device=dev), torch.empty(N, device=dev)
        if z2 is not None:
            dgamma2, dbeta2 = torch.empty(N, device=dev), torch.empty(N, device=dev)
    if z2 is not None:
        dz2 = torch.empty_like(z2)
    call("m3d_bn_bwd", _p(_chka(dy)), _p(z), _p(scale), _p(shift), _p(mean), _p(invstd), _p(z2), _p(scale2),
         _p(shift2), _p(mean2), _p(invstd2), int(act), LRELU_SLOPE, M, N, _p(sums), _p(dz), _p(dz2), _p(dgamma),
         _p(dbeta), _p(dgamma2), _p(dbeta2), int(sinks is not None) | (slots << 8) | (io << 16), _drop_ref(drop), _st())
    if sinks is not None:
        return dz, None, None, dz2, None, None
    return dz, dgamma, dbeta, dz2, dgamma2, dbeta2


def _bn_io(dy: Tensor, z: Tensor) -> int:
    """(M3D_IO_* >> 12) of a BatchNorm-backward launch: 0 = fp32, 1 = dy and z (dz, dx) bf16, 3 = bf16 storage with an fp32
    dy (an incoming gradient that was accumulated with float atomics)."""
    if not _h(z):
        assert not _h(dy), "a bf16 gradient for an fp32 layer"
        return 0
    return 1 if _h(dy) else 3


def bn_dgrad_ok(N: int) -> bool:
    """Shapes ``m3d_bn_dgrad_f32`` takes: power-of-two BatchNorm widths (slot-mode sums) up to 1024."""
    return _pow2(N) and N <= 1024


def bn_dgrad(dy, z, scale, shift, mean, invstd, act, w, sinks=None, bf16=False, split: int = 0, acc: Optional[Tensor] = None,
             drop=None):
    """BatchNorm backward + input gradient of the Linear in front of it in TWO launches: the column sums
    (``m3d_bn_bwd`` pass 1, slot mode), then ``m3d_bn_dgrad_f32``, whose A fragments are dz computed on the fly.
    Returns ``(dx, dz, dgamma, dbeta)`` (the last two None with sinks).  ``split = k0 > 0``: ``dx`` is the pair
    ``(dx[:, :k0], dx[:, k0:])`` as two contiguous matrices (the layer's input was a concatenation).  ``acc`` (no
    split): an existing ``[M, Kin]`` buffer the input gradient is ADDED to (and which is returned as ``dx``)."""
    M, N = z.shape
    dev = z.device
    dy = _chka(dy)
    io = _bn_io(dy, z)
    iof = io << 12  # (IO_BF16 [| IO_A32])
    ns = bn_bwd_slots(M)
    sums = arena.zeros((ns, 3, N), torch.float64, dev)
    call("m3d_bn_bwd", _p(dy), _p(z), _p(scale), _p(shift), _p(mean), _p(invstd), None, None, None, None, None,
         int(act), LRELU_SLOPE, M, N, _p(sums), None, None, None, None, None, None, 2 | (ns << 8) | (io << 16),
         _drop_ref(drop), _st())
    if sinks is not None:
        dgamma, dbeta = sinks
    else:
        dgamma, dbeta = torch.empty(N, device=dev), torch.empty(N, device=dev)
    Kin = w.shape[1]
    dz = torch.empty_like(z)
    if split:
        dx = (torch.empty((M, split), dtype=z.dtype, device=dev),
              torch.empty((M, Kin - split), dtype=z.dtype, device=dev))
        call("m3d_bn_dgrad_f32", _p(dy), _p(z), _p(scale), _p(shift), _p(mean), _p(invstd), int(act), LRELU_SLOPE,
             _p(sums), ns, M, N, _p(w), w.stride(0), Kin, _p(dx[0]), split, _p(dz), _p(dgamma), _p(dbeta),
             int(sinks is not None) | (256 if bf16 else 0) | iof, split, _p(dx[1]), Kin - split, _drop_ref(drop), _st())
    else:
        dx = acc if acc is not None else torch.empty((M, Kin), dtype=z.dtype, device=dev)
        assert dx.shape == (M, Kin) and dx.is_contiguous() and dx.dtype == z.dtype
        call("m3d_bn_dgrad_f32", _p(dy), _p(z), _p(scale), _p(shift), _p(mean), _p(invstd), int(act), LRELU_SLOPE,
             _p(sums), ns, M, N, _p(w), w.stride(0), Kin, _p(dx), Kin, _p(dz), _p(dgamma), _p(dbeta),
             int(sinks is not None) | (256 if bf16 else 0) | (512 if acc is not None else 0) | iof, 0, None, 0,
             _drop_ref(drop), _st())
    if sinks is not None:
        return dx, dz, None, None
    return dx, dz, dgamma, dbeta


# --------------------------------------------------------------------------------------------------
# autograd: Linear (fc0, fc_classif: pyg_randla_net.py:42,53)
# --------------------------------------------------------------------------------------------------
class LinearFn(torch.autograd.Function):
    """``sinks = (grad_w, grad_b)`` or None: see ``HipRandLANet.flatten_parameters`` (parameter gradients are
    accumulated straight into the flat gradient buffer and autograd gets None for them)."""

    @staticmethod
    def forward(ctx, x, w, b, sinks=None, rows=None, out_dtype=None):
        # rows: the layer reads x[rows] (a row gather fused into the GEMM's A operand — fc0 on the cell-sorted order of
        # level 1 — and into the weight gradient's); only for inputs that need no gradient themselves
        # out_dtype: fp32 logits from bf16 activations (fc_classif of a net with bf16 activation storage)
        x = x.contiguous()
        assert rows is None or not x.requires_grad
        ctx.save_for_backward(x, w, rows)
        ctx.sinks = sinks
        ctx.side = _grad_side if sinks is not None else None
        return gemm(x, w, x.shape[0] if rows is None else rows.numel(), w.shape[0], w.shape[1], rows=rows, bias=b,
                    out_dtype=out_dtype)

    @staticmethod
    def backward(ctx, dy):
        x, w, rows = ctx.saved_tensors
        sk = ctx.sinks
        dy = dy.contiguous()
        if _h(x) and not _h(dy):
            dy = to_bf16(dy)  # (fp32 logits of a bf16 net: autograd hands over their gradient in fp32)
        dx = linear_dgrad(dy, w) if ctx.needs_input_grad[0] else None
        dw = linear_wgrad(dy, x, x.shape[1], rows=rows, out=sk[0] if sk else None, side=ctx.side)
        return dx, dw, colsum(dy, out=sk[1] if sk else None), None, None, None


# --------------------------------------------------------------------------------------------------
# autograd: one SharedMLP layer in train mode = Linear -> BatchNorm(batch stats) -> [LeakyReLU]
# (pyg_randla_net.py:97-109).  The input may be cat([x0[rows], x1]) (FPModule, pyg_randla_net.py:249-252).
# --------------------------------------------------------------------------------------------------
class SharedLayerTrainFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x0, x1, w, b, gamma, beta, bn, act, rows, sinks=None, bf16=False, x0_slot=None, x1_slot=None,
                drop=None, rows_inv=None, defer_apply=False, y_slot=None):
        # sinks = (grad_w, grad_b, grad_gamma, grad_beta) or None;  bf16: matrix-core precision of the K > 64 GEMMs
        # x0_slot: GradSlot of x0, this layer being its LAST consumer in backward order (adds its input gradient to what
        # the others deposited and returns the sum); x1_slot: GradSlot of x1, this layer being the FIRST (deposits)
        # drop = (p, counter, seed): the layer's output goes through Dropout(p) (mlp_classif, pyg_randla_net.py:49-52) inside the
        # BatchNorm kernels — masked on the way out here, the incoming gradient masked on load in the backward pass
        # rows_inv = (ptr, inv): CSR inverse of ``rows`` (csr_invert_batch) — the backward pass then sums the gathered rows'
        # gradients per source row instead of scattering them with atomics
        # y_slot (bf16 activation storage): the layer's ONLY consumer — an LFA layer whose backward kernel accumulates its
        # input gradient with fp32 atomics — deposits that fp32 buffer there and hands autograd an uninitialised bf16
        # placeholder (the engine would otherwise cast every such gradient to bf16 with a kernel of its own)
        ctx.y_slot = y_slot
        ctx.slots = (x0_slot, x1_slot)
        ctx.rows_inv = rows_inv if (rows is not None and x0.shape[1] % 4 == 0) else None
        if drop is not None:
            # the backward pass rebuilds the mask from the counter value THIS forward saw, not from the live counter: another
            # train-mode forward may run (and advance it) before this one's backward
            snap = torch.empty(1, dtype=torch.int64, device=w.device)
            rows_d = drop[3] if len(drop) > 3 else None
            ctx.drop = (drop[0], snap, drop[2], rows_d)
            drop = (drop[0], drop[1], drop[2], rows_d, snap)
        else:
            ctx.drop = None
        ctx.sinks = sinks
        ctx.bf16 = bool(bf16)
        ctx.side = _grad_side if sinks is not None else None
        M = x1.shape[0] if x1 is not None else (rows.numel() if rows is not None else x0.shape[0])
        N = w.shape[0]
        k0 = x0.shape[1]
        k1 = x1.shape[1] if x1 is not None else 0
        # x0 may be the (unwritten) activation buffer of the layer in front (PendingBN): this GEMM then applies that layer's
        # BatchNorm + LeakyReLU as it loads its A fragments and stores the activation (defer_apply of the layer in front)
        pend = take_pending(x0) if (x1 is None and rows is None) else None
        if pend is None and x0 is not None and take_pending(x0) is not None:
            take_pending(x0).materialize()  # (a gathered / concatenated read: the plain launch first)
            _pending.remove(take_pending(x0))
        global _last_pending
        if _pow2(N):  # slot-mode statistics: GEMM + ONE fused finalize/apply launch
            stats = stat_slots(N, w.device, M)
            z = gemm_bn_on_load(pend, w, M, N, b, stats) if pend is not None else None
            if z is None:
                if pend is not None:
                    pend.materialize()
                    _pending.remove(pend)
                z = gemm(x0, w, M, N, k0, rows=rows, a1=x1, k1=k1, bias=b, stats=stats, stat_slots=True, bf16=bf16)
            if defer_apply and drop is None and BN_ON_LOAD:
                # the consumer's GEMM applies this layer's BatchNorm (the caller knows it is a SharedMLP layer on the same rows)
                pv = torch.empty((4, N), dtype=torch.float32, device=w.device).unbind(0)
                y = torch.empty_like(z)
                scale, shift, mean, invstd = pv
                if M < 2:
                    raise ValueError(f"Expected more than 1 value per channel when training, got input size [{M}, {N}]")
                _last_pending = PendingBN(z, stats, M, bn, act, y, pv)
                _pending.append(_last_pending)
            else:
                y, (scale, shift, mean, invstd) = bn_stats_apply(stats, M, bn, z, act, drop=drop)
        else:
            if drop is not None:
                raise ValueError("fused dropout needs a power-of-two layer width")
            if pend is not None:
                pend.materialize()
                _pending.remove(pend)
            stats = stat_buffer(M, N, k0 + k1, w.device)
            z = gemm(x0, w, M, N, k0, rows=rows, a1=x1, k1=k1, bias=b, stats=stats, bf16=bf16)
            scale, shift, mean, invstd = bn_finalize(stats, M, bn)
            y = bn_apply(z, scale, shift, act)
        ctx.save_for_backward(x0, x1, w, z, scale, shift, mean, invstd, rows)
        ctx.act = act
        return y

    @staticmethod
    def backward(ctx, dy):
        x0, x1, w, z, scale, shift, mean, invstd, rows = ctx.saved_tensors
        sk = ctx.sinks
        if ctx.y_slot is not None and ctx.y_slot.buf is not None:
            dy = ctx.y_slot.take()  # (the consumer's real gradient: fp32; `dy` itself is a placeholder)
        k0 = x0.shape[1]
        k1 = x1.shape[1] if x1 is not None else 0
        dx0 = dx1 = None
        want_dx = ctx.needs_input_grad[0] or (x1 is not None and ctx.needs_input_grad[1])
        dxc = None
        x0_slot, x1_slot = ctx.slots
        acc0 = x0_slot.take() if (x0_slot is not None and ctx.needs_input_grad[0]) else None
        # (a few hundred rows x wide layers — mlp_summit: 800 x 512 -> 512 — the prologue, repeated in every column slice,
        # costs more than the apply launch it replaces: 35 vs 23 us, profiles/r04h_gemm_*.log)
        fused = want_dx and FUSE_BN_DGRAD and bn_dgrad_ok(z.shape[1]) and (z.shape[0] >= FUSE_MIN_ROWS or k1 > 0)
        if fused and z.shape[0] * max(z.shape[1], w.shape[1]) * 4 >= (1 << 31) - 64:
            fused = False  # beyond the fused kernel's 2 GiB buffer descriptors: the two-pass path handles it
        if fused and k1 and k0 % 4 == 0 and k1 % 4 == 0:
            # concatenated input: the two column blocks of the input gradient leave the GEMM as two contiguous matrices
            (s0, s1), dz, dgamma, dbeta = bn_dgrad(dy.contiguous(), z, scale, shift, mean, invstd, ctx.act, w,
                                                   sinks=(sk[2], sk[3]) if sk else None, bf16=ctx.bf16, split=k0,
                                                   drop=ctx.drop)
            if ctx.needs_input_grad[0]:
                if rows is not None and ctx.rows_inv is not None:
                    dx0 = gather_sum_rows(s0, ctx.rows_inv[0], ctx.rows_inv[1], x0.shape[0])
                else:
                    dx0 = scatter_add_rows(s0, rows, x0.shape[0]) if rows is not None else s0
            if ctx.needs_input_grad[1]:
                dx1 = s1
            want_dx = False
        elif fused:
            direct = acc0 is not None and k1 == 0 and rows is None and acc0.shape == (z.shape[0], k0)
            dxc, dz, dgamma, dbeta = bn_dgrad(dy.contiguous(), z, scale, shift, mean, invstd, ctx.act, w,
                                              sinks=(sk[2], sk[3]) if sk else None, bf16=ctx.bf16,
                                              acc=acc0 if direct else None, drop=ctx.drop)
            if direct:
                acc0 = None  # already inside dxc
        else:
            dz, dgamma, dbeta, _, _, _ = bn_bwd(dy.contiguous(), z, scale, shift, mean, invstd, ctx.act,
                                                sinks=(sk[2], sk[3]) if sk else None, drop=ctx.drop)
        if want_dx:
            if dxc is None:
                dxc = linear_dgrad(dz, w, ctx.bf16)
            if ctx.needs_input_grad[0]:
                dx0 = dxc[:, :k0]
                if rows is not None and ctx.rows_inv is not None:
                    dx0 = gather_sum_rows(dx0.contiguous(), ctx.rows_inv[0], ctx.rows_inv[1], x0.shape[0])
                elif rows is not None:
                    dx0 = scatter_add_rows(dx0.contiguous(), rows, x0.shape[0])
                elif k1:
                    dx0 = dx0.contiguous()
            if x1 is not None and ctx.needs_input_grad[1]:
                dx1 = dxc[:, k0:].contiguous()
        if acc0 is not None and dx0 is not None:  # (shapes the fused accumulate does not take: one torch add)
            dx0 = acc0.add_(dx0)
        if x1_slot is not None and dx1 is not None:
            x1_slot.buf = dx1 if dx1.is_contiguous() else dx1.contiguous()  # deposited: the tensor's last consumer returns it
            dx1 = None
        dw = linear_wgrad(dz, x0, k0, rows, x1, k1, out=sk[0] if sk else None, side=ctx.side, bf16=ctx.bf16)
        db = None if sk else torch.zeros_like(dbeta)  # BatchNorm removes the mean: d/d(bias) is exactly 0
        return dx0, dx1, dw, db, dgamma, dbeta, None, None, None, None, None, None, None, None, None, None, None


# LeakyReLU(BN(mlp2(x2)) + BN(shortcut(xs)))   (DilatedResidualBlock tail, pyg_randla_net.py:186-187)
class ResidualTailTrainFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x2, w2, b2, g2, be2, bn2, xs, ws, bs, gs, bes, bns, sinks2=None, sinkss=None, bf16=False,
                xs_slot=None):
        # xs_slot: GradSlot of the block input xs (its gradient is deposited / added there; mlp1, the last consumer in
        # backward order, returns the sum)
        ctx.xs_slot = xs_slot
        ctx.sinks = (sinks2, sinkss) if sinks2 is not None else None
        ctx.bf16 = bool(bf16)
        ctx.side = _grad_side if sinks2 is not None else None
        M, N = x2.shape[0], w2.shape[0]
        if _pow2(N):
            st2, sts = stat_slots(N, w2.device, M), stat_slots(N, w2.device, M)
            if PAIR_GEMMS and min(x2.shape[1], xs.shape[1]) > 64 and st2.shape == sts.shape:
                # deep levels: both Linears in one launch (each alone is a few hundred workgroups)
                if take_pending(x2) is not None:
                    _settle(take_pending(x2))
                z2, zs = gemm_pair((x2.contiguous(), xs.contiguous()), (w2, ws), M, N, bias=(b2, bs), stats=(st2, sts), bf16=bf16)
            else:
                pend = take_pending(x2)
                z2 = gemm_bn_on_load(pend, w2, M, N, b2, st2) if pend is not None else None
                if z2 is None:
                    if pend is not None:
                        pend.materialize()
                        _pending.remove(pend)
                    z2 = gemm(x2, w2, M, N, x2.shape[1], bias=b2, stats=st2, stat_slots=True, bf16=bf16)
                zs = gemm(xs, ws, M, N, xs.shape[1], bias=bs, stats=sts, stat_slots=True, bf16=bf16)
            y, (sc2, sh2, mu2, is2), (scs, shs, mus, iss) = bn_stats_apply(st2, M, bn2, z2, True, sts, bns, zs)
        else:
            if take_pending(x2) is not None:
                _settle(take_pending(x2))
            st2 = stat_buffer(M, N, x2.shape[1], w2.device)
            sts = stat_buffer(M, N, xs.shape[1], w2.device)
            z2 = gemm(x2, w2, M, N, x2.shape[1], bias=b2, stats=st2, bf16=bf16)
            zs = gemm(xs, ws, M, N, xs.shape[1], bias=bs, stats=sts, bf16=bf16)
            sc2, sh2, mu2, is2 = bn_finalize(st2, M, bn2)
            scs, shs, mus, iss = bn_finalize(sts, M, bns)
            y = bn_apply(z2, sc2, sh2, True, zs, scs, shs)
        ctx.save_for_backward(x2, w2, z2, sc2, sh2, mu2, is2, xs, ws, zs, scs, shs, mus, iss)
        return y

    @staticmethod
    def backward(ctx, dy):
        x2, w2, z2, sc2, sh2, mu2, is2, xs, ws, zs, scs, shs, mus, iss = ctx.saved_tensors
        sk = ctx.sinks
        dz2, dg2, db2, dzs, dgs, dbs = bn_bwd(dy.contiguous(), z2, sc2, sh2, mu2, is2, True, zs, scs, shs, mus, iss,
                                              sinks=(sk[0][2], sk[0][3], sk[1][2], sk[1][3]) if sk else None)
        slot = ctx.xs_slot
        use_slot = slot is not None and ctx.needs_input_grad[6]
        prev = slot.buf if use_slot else None
        ok = prev is not None and prev.shape == (dzs.shape[0], ws.shape[1]) and prev.is_contiguous()
        if PAIR_GEMMS and w2.shape[1] == ws.shape[1] and dz2.shape[1] > 64:
            # dX_i[M, K] = dZ_i[M, N] W_i: the same output shape on both sides (deep levels: K_in of mlp2 = K_in of the shortcut)
            dx2, dxs = gemm_pair((dz2, dzs), (w2, ws), dz2.shape[0], w2.shape[1], out=(None, prev if ok else None),
                                 accumulate=(False, ok), b_cm=True, bf16=ctx.bf16)
        else:
            dx2 = linear_dgrad(dz2, w2, ctx.bf16)
            dxs = linear_dgrad(dzs, ws, ctx.bf16, acc=prev if ok else None)
        if use_slot:
            slot.buf = dxs
            if prev is not None and not ok:
                slot.buf.add_(prev)
            dxs = None
        dw2 = linear_wgrad(dz2, x2, x2.shape[1], out=sk[0][0] if sk else None, side=ctx.side, bf16=ctx.bf16)
        dws = linear_wgrad(dzs, xs, xs.shape[1], out=sk[1][0] if sk else None, side=ctx.side, bf16=ctx.bf16)
        z0_2 = None if sk else torch.zeros_like(db2)
        z0_s = None if sk else torch.zeros_like(dbs)
        return (dx2, dw2, z0_2, dg2, db2, None, dxs, dws, z0_s, dgs, dbs, None, None, None, None, None)


# --------------------------------------------------------------------------------------------------
# autograd in EVAL mode (BatchNorm = an affine map with the running statistics; the reference's eval forward is
# differentiable like any torch module: saliency maps, fine-tuning with frozen statistics).  Rarely used and not tuned:
# the GEMMs / LFA kernels are the HIP ones, the BatchNorm-backward arithmetic is a few torch elementwise ops.
# --------------------------------------------------------------------------------------------------
def _eval_bn_terms(bn):
    scale, shift = bn_fold_eval(bn)
    return scale, shift, bn.running_mean, torch.rsqrt(bn.running_var + bn.eps)


def _eval_bn_backward(g, z, scale, mean, invstd):
    """``g`` = gradient w.r.t. the BatchNorm OUTPUT (activation derivative already applied).  Returns dz, dgamma, dbeta."""
    dgamma = (g * ((z - mean) * invstd)).sum(0)
    return (g * scale).contiguous(), dgamma, g.sum(0)


class SharedLayerEvalFn(torch.autograd.Function):
    """One SharedMLP layer with eval-mode BatchNorm, differentiable (cf. SharedLayerTrainFn)."""

    @staticmethod
    def forward(ctx, x0, x1, w, b, gamma, beta, bn, act, rows):
        scale, shift, mean, invstd = _eval_bn_terms(bn)
        M = x1.shape[0] if x1 is not None else (rows.numel() if rows is not None else x0.shape[0])
        k0 = x0.shape[1]
        k1 = x1.shape[1] if x1 is not None else 0
        x0 = x0.contiguous()
        z = gemm(x0, w, M, w.shape[0], k0, rows=rows, a1=x1, k1=k1, bias=b)
        y = bn_apply(z, scale, shift, act)
        ctx.save_for_backward(x0, x1, w, z, scale, shift, mean, invstd, rows)
        ctx.act = act
        return y

    @staticmethod
    def backward(ctx, dy):
        x0, x1, w, z, scale, shift, mean, invstd, rows = ctx.saved_tensors
        k0 = x0.shape[1]
        k1 = x1.shape[1] if x1 is not None else 0
        g = dy
        if ctx.act:
            g = dy * torch.where(z * scale + shift > 0, 1.0, LRELU_SLOPE)
        dz, dgamma, dbeta = _eval_bn_backward(g, z, scale, mean, invstd)
        dxc = linear_dgrad(dz, w)
        dx0 = dxc[:, :k0].contiguous()
        if rows is not None:
            dx0 = scatter_add_rows(dx0, rows, x0.shape[0])
        dx1 = dxc[:, k0:].contiguous() if x1 is not None else None
        dw = linear_wgrad(dz, x0, k0, rows, x1, k1)
        return dx0, dx1, dw, dz.sum(0), dgamma, dbeta, None, None, None


class ResidualTailEvalFn(torch.autograd.Function):
    """LeakyReLU(BN(mlp2(x2)) + BN(shortcut(xs))) with eval-mode BatchNorms, differentiable."""

    @staticmethod
    def forward(ctx, x2, w2, b2, g2, be2, bn2, xs, ws, bs, gs, bes, bns):
        sc2, sh2, mu2, is2 = _eval_bn_terms(bn2)
        scs, shs, mus, iss = _eval_bn_terms(bns)
        x2, xs = x2.contiguous(), xs.contiguous()
        M, N = x2.shape[0], w2.shape[0]
        z2 = gemm(x2, w2, M, N, x2.shape[1], bias=b2)
        zs = gemm(xs, ws, M, N, xs.shape[1], bias=bs)
        y = bn_apply(z2, sc2, sh2, True, zs, scs, shs)
        ctx.save_for_backward(x2, w2, z2, sc2, mu2, is2, xs, ws, zs, scs, mus, iss, y)
        return y

    @staticmethod
    def backward(ctx, dy):
        x2, w2, z2, sc2, mu2, is2, xs, ws, zs, scs, mus, iss, y = ctx.saved_tensors
        g = dy * torch.where(y > 0, 1.0, LRELU_SLOPE)
        dz2, dg2, db2 = _eval_bn_backward(g, z2, sc2, mu2, is2)
        dzs, dgs, dbs = _eval_bn_backward(g, zs, scs, mus, iss)
        return (linear_dgrad(dz2, w2), linear_wgrad(dz2, x2, x2.shape[1]), dz2.sum(0), dg2, db2, None,
                linear_dgrad(dzs, ws), linear_wgrad(dzs, xs, xs.shape[1]), dzs.sum(0), dgs, dbs, None)


class LFAEvalFn(torch.autograd.Function):
    """aggregate() of LocalFeatureAggregation with the encoder's BatchNorm in eval mode, differentiable."""

    @staticmethod
    def forward(ctx, x, pos4, idx, enc_w, enc_b, enc_gamma, enc_beta, enc_lin, enc_bn, w_att):
        x = x.contiguous()
        K = idx.shape[1]
        if K <= 32:
            wf, bf, _, _, wp, wpt = lfa_prepare(enc_lin, enc_bn, None, 0, w_att, False, True)
        else:
            wf, bf, _, _ = lfa_enc_fold(enc_lin, enc_bn, None, 0)
            wp = wpt = None
        out = lfa_forward(x, pos4, idx, wf, bf, w_att, wp)
        ctx.packed = (wp, wpt)
        ctx.save_for_backward(x, pos4, idx, wf, bf, enc_w, enc_b, enc_gamma, w_att, enc_bn.running_mean,
                              torch.rsqrt(enc_bn.running_var + enc_bn.eps))
        return out

    @staticmethod
    def backward(ctx, dout):
        x, pos4, idx, wf, bf, enc_w, enc_b, enc_gamma, w_att, rmean, invstd = ctx.saved_tensors
        n, K = idx.shape
        ch = w_att.shape[0]
        D = ch // 2
        dev = x.device
        dout = dout.contiguous()
        dx = torch.zeros((n, D), dtype=torch.float32, device=dev)
        G = torch.zeros(11 * D, dtype=torch.float64, device=dev)
        if K <= 32:
            dw_att = torch.empty((ch, ch), dtype=torch.float32, device=dev)
            ws = torch.empty(lib().m3d_lfa_bwd_workspace_bytes(n, K, ch), dtype=torch.uint8, device=dev)
            wp, wpt = ctx.packed
            call("m3d_lfa_bwd", _p(x), _p(pos4), _p(idx), n, K, ch, _p(wf), _p(bf), _p(wp), _p(wpt), LRELU_SLOPE, _p(dout),
                 _p(dx), _p(dw_att), 2, _p(G), _p(ws), _st())
        else:
            dw_att = _lfa_backward_unfused(x, pos4, idx, wf, bf, w_att, dout, dx, G)
        # G[c] = sum over the edges of dy[e, c] * [r_e | 1], dy = gradient at the encoder BatchNorm's output
        Gm = G.view(D, 11).to(torch.float32)
        sc = enc_gamma * invstd
        dw = sc[:, None] * Gm[:, :10]
        db = sc * Gm[:, 10]
        dgamma = invstd * ((enc_w * Gm[:, :10]).sum(1) + (enc_b - rmean) * Gm[:, 10])
        dbeta = Gm[:, 10].clone()
        return dx, None, None, dw, db, dgamma, dbeta, None, None, dw_att


class GatherRowsFn(torch.autograd.Function):
    """``x[idx]``.  ``inverse``: ``idx`` is a permutation and ``inverse`` its inverse map — the backward pass is then the
    gather ``dy[inverse]`` (no atomics, no zero fill) instead of a scatter-add."""

    @staticmethod
    def forward(ctx, x, idx, inverse=None, slot=None, distinct=False):
        # slot: GradSlot of x, this gather being its LAST consumer in backward order: the rows are scatter-added into
        # the gradient another consumer deposited (no zero fill, no elementwise add)
        # distinct: no row is gathered twice (a subset selection): the backward pass needs no atomics
        ctx.save_for_backward(idx, inverse)
        ctx.n = x.shape[0]
        ctx.slot = slot
        ctx.distinct = bool(distinct)
        return gather_rows(x.contiguous(), idx)

    @staticmethod
    def backward(ctx, dy):
        idx, inverse = ctx.saved_tensors
        if inverse is not None:
            return gather_rows(dy.contiguous(), inverse), None, None, None, None
        prev = ctx.slot.take() if ctx.slot is not None else None
        if _h(dy) and not ctx.distinct:
            # bf16 rows that may repeat (injected decimation indices: tests): float atomics into an fp32 buffer; autograd casts
            res = scatter_add_rows(dy.contiguous(), idx, ctx.n)
            return (res if prev is None else res.add_(prev)), None, None, None, None
        if prev is not None and not (prev.shape == (ctx.n, dy.shape[1]) and prev.is_contiguous() and prev.dtype == dy.dtype):
            return scatter_add_rows(dy.contiguous(), idx, ctx.n, distinct=ctx.distinct).add_(prev), None, None, None, None
        return scatter_add_rows(dy.contiguous(), idx, ctx.n, out=prev, distinct=ctx.distinct), None, None, None, None


# --------------------------------------------------------------------------------------------------
# Local spatial encoding + attentive pooling (pyg_randla_net.py:112-152)
# --------------------------------------------------------------------------------------------------
def pack_attention_weight(w: Tensor) -> Tensor:
    """[CH,CH] row-major -> MFMA B-fragment order expected by ``m3d_lfa_fwd`` (layout: include/m3d_hip.h)."""
    ch = w.shape[0]
    chp = max(ch, 16)
    if chp != ch:
        wpad = torch.zeros((chp, chp), dtype=w.dtype, device=w.device)
        wpad[:ch, :ch] = w
        w = wpad
    nt = chp // 16
    # W[16*nt + r][4*(4*s4 + i) + g]  ->  [nt][s4][g][r][i]   (lane = 16*g + r)
    return w.reshape(nt, 16, nt, 4, 4).permute(0, 2, 4, 1, 3).contiguous()


def pack_attention_weights(w: Tensor, with_transpose: bool) -> Tuple[Tensor, Optional[Tensor]]:
    """``pack_attention_weight(w)`` (and of ``w.t()``) in ONE kernel launch (``m3d_lfa_pack_att``)."""
    chp = max(w.shape[0], 16)
    wp = torch.empty(chp * chp, dtype=torch.float32, device=w.device)
    wpt = torch.empty_like(wp) if with_transpose else None
    call("m3d_lfa_pack_att", _p(_chk(w)), w.shape[0], _p(wp), _p(wpt), _st())
    return wp, wpt


def lfa_moments_batch(pos4s, idxs) -> List[Tensor]:
    """``lfa_moments`` of several levels in one launch (``m3d_lfa_moments_batch``); returns ``[65]`` views of one buffer."""
    import ctypes

    m = len(pos4s)
    mom = torch.empty((m, 72), dtype=torch.float64, device=pos4s[0].device)  # 576-byte rows: 16-byte aligned views
    pp = (ctypes.c_void_p * m)(*[t.data_ptr() for t in pos4s])
    ip = (ctypes.c_void_p * m)(*[_chk(t, torch.int32).data_ptr() for t in idxs])
    nn_ = (ctypes.c_int64 * m)(*[t.shape[0] for t in idxs])
    call("m3d_lfa_moments_batch", m, pp, ip, nn_, idxs[0].shape[1], _p(mom), mom.stride(0), _st())
    return [mom[i, :65] for i in range(m)]


def knn_query_batch(pairs, k: int, sorted_io: bool = True, kernel: str = "auto", background: int = 0) -> List[Tensor]:
    """``src.query(k, qry=qry, sorted_io=...)`` for up to 8 ``(src, qry)`` pairs of built ``KnnIndex`` objects in ONE
    launch (``m3d_knn_query_batch``); bit-identical tables."""
    import ctypes

    m = len(pairs)
    dev = pairs[0][0].ws.device
    outs = [torch.empty((q.n, k), dtype=torch.int32, device=dev) for _, q in pairs]
    for s_, q in pairs:
        assert s_.num_clouds == q.num_clouds == pairs[0][0].num_clouds
    vp = lambda ts: (ctypes.c_void_p * m)(*[t.data_ptr() for t in ts])
    call("m3d_knn_query_batch", m, vp([s_.ws for s_, _ in pairs]), vp([s_.ptr for s_, _ in pairs]),
         (ctypes.c_int64 * m)(*[s_.n for s_, _ in pairs]), vp([q.ws for _, q in pairs]), vp([q.ptr for _, q in pairs]),
         (ctypes.c_int64 * m)(*[q.n for _, q in pairs]), pairs[0][0].num_clouds, k,
         int(sorted_io) | (_KNN_KERNEL[kernel] << 1) | ((int(background) & 0xff) << 8), vp(outs), _st())
    return outs


def lfa_moments(pos4: Tensor, idx: Tensor) -> Tensor:
    mom = torch.empty(65, dtype=torch.float64, device=pos4.device)
    call("m3d_lfa_moments", _p(pos4), _p(idx), idx.shape[0], idx.shape[1], _p(mom), _st())
    return mom


def lfa_enc_fold(enc_lin, enc_bn, mom: Optional[Tensor], num_edges: int):
    """Fold mlp_encoder's BatchNorm into its Linear (train: batch moments + running-stat update; eval: running)."""
    D = enc_lin.weight.shape[0]
    dev = enc_lin.weight.device
    wf = torch.empty((D, 10), dtype=torch.float32, device=dev)
    bf, mean, invstd = (torch.empty(D, dtype=torch.float32, device=dev) for _ in range(3))
    if mom is not None and num_edges < 2:
        raise ValueError(f"Expected more than 1 value per channel when training, got input size [{num_edges}, {D}]")
    call("m3d_lfa_enc_finalize", _p(mom), num_edges, _p(enc_lin.weight), _p(enc_lin.bias), _p(enc_bn.weight),
         _p(enc_bn.bias), float(enc_bn.eps), float(enc_bn.momentum), _p(enc_bn.running_mean),
         _p(enc_bn.running_var), _p(wf), _p(bf), _p(mean), _p(invstd), D, _st())
    if mom is not None and not getattr(enc_bn, "_m3d_flat_counter", False):
        enc_bn.num_batches_tracked += 1
    return wf, bf, mean, invstd


def lfa_bf16_ok(ch: int, K: int) -> bool:
    """The bf16 matrix-core variants are used where the layer is matrix-bound: ch >= 64 (and K <= 32).  (The kernels exist from
    ch = 32 — tests/test_gpu_ops.py::test_lfa_bf16_matrix_core_variant — but at ch = 32 they measured nothing, 3.075 vs 3.083
    ms per step, and cost block 1's attention-weight gradient 0.09 -> 0.19 relative L2 against the fp32 kernels: round 6.)"""
    return ch >= 64 and K <= 32


def lfa_bf16_kernels_ok(ch: int, K: int) -> bool:
    """Shapes the bf16 matrix-core LFA kernels are instantiated for (``LFATrainFn`` honours a caller's request there)."""
    return ch >= 32 and K <= 32


def lfa_prepare(enc_lin, enc_bn, mom: Optional[Tensor], num_edges: int, w_att: Tensor, bf16: bool, want_t: bool):
    """``lfa_enc_fold`` + ``pack_attention_weights`` (fp32 fragments, or the bf16 operand fragments) in ONE launch
    (``m3d_lfa_prepare``).  Returns ``(wf, bf, mean, invstd, wp, wpt)``."""
    D = enc_lin.weight.shape[0]
    ch = w_att.shape[0]
    dev = enc_lin.weight.device
    if mom is not None and num_edges < 2:
        raise ValueError(f"Expected more than 1 value per channel when training, got input size [{num_edges}, {D}]")
    wf = torch.empty((D, 10), dtype=torch.float32, device=dev)
    bf, mean, invstd = (torch.empty(D, dtype=torch.float32, device=dev) for _ in range(3))
    if bf16:  # (2 = split-bf16: the hi fragments, then the lo fragments)
        wp = torch.empty(ch * ch * (2 if int(bf16) == 2 else 1), dtype=torch.int16, device=dev)
    else:
        wp = torch.empty(max(ch, 16) ** 2, dtype=torch.float32, device=dev)
    wpt = torch.empty_like(wp) if want_t else None
    call("m3d_lfa_prepare", _p(mom), num_edges, _p(enc_lin.weight), _p(enc_lin.bias), _p(enc_bn.weight), _p(enc_bn.bias),
         float(enc_bn.eps), float(enc_bn.momentum), _p(enc_bn.running_mean), _p(enc_bn.running_var), _p(wf), _p(bf),
         _p(mean), _p(invstd), D, _p(_chk(w_att)), ch, _p(wp), _p(wpt), int(bf16), _st())
    if mom is not None and not getattr(enc_bn, "_m3d_flat_counter", False):
        enc_bn.num_batches_tracked += 1
    return wf, bf, mean, invstd, wp, wpt


def lfa_prepare_batch(jobs) -> list:
    """``lfa_prepare(enc_lin, enc_bn, mom, num_edges, w_att, bf16, True)`` (train mode) for several LFA layers in ONE launch
    (``m3d_lfa_prepare_batch``).  ``jobs``: ``(enc_lin, enc_bn, mom, num_edges, w_att, bf16)`` tuples; returns the
    ``(wf, bf, mean, invstd, wp, wpt)`` tuple of each."""
    import ctypes

    m = len(jobs)
    outs = []
    for enc_lin, enc_bn, mom, num_edges, w_att, bf16 in jobs:
        D, ch, dev = enc_lin.weight.shape[0], w_att.shape[0], enc_lin.weight.device
        if num_edges < 2:
            raise ValueError(f"Expected more than 1 value per channel when training, got input size [{num_edges}, {D}]")
        wf = torch.empty((D, 10), dtype=torch.float32, device=dev)
        vec = torch.empty((3, D), dtype=torch.float32, device=dev)
        wp = torch.empty(ch * ch * (2 if int(bf16) == 2 else 1), dtype=torch.int16, device=dev) if bf16 else \
            torch.empty(max(ch, 16) ** 2, dtype=torch.float32, device=dev)
        outs.append((wf, vec[0], vec[1], vec[2], wp, torch.empty_like(wp)))
    vp = lambda vals: (ctypes.c_void_p * m)(*vals)
    bn0 = jobs[0][1]
    call("m3d_lfa_prepare_batch", m, vp([j[2].data_ptr() for j in jobs]), (ctypes.c_int64 * m)(*[j[3] for j in jobs]),
         vp([j[0].weight.data_ptr() for j in jobs]), vp([j[0].bias.data_ptr() for j in jobs]),
         vp([j[1].weight.data_ptr() for j in jobs]), vp([j[1].bias.data_ptr() for j in jobs]), float(bn0.eps),
         float(bn0.momentum), vp([j[1].running_mean.data_ptr() for j in jobs]),
         vp([j[1].running_var.data_ptr() for j in jobs]), vp([o[0].data_ptr() for o in outs]),
         vp([o[1].data_ptr() for o in outs]), vp([o[2].data_ptr() for o in outs]), vp([o[3].data_ptr() for o in outs]),
         (ctypes.c_int32 * m)(*[j[0].weight.shape[0] for j in jobs]), vp([_chk(j[4]).data_ptr() for j in jobs]),
         (ctypes.c_int32 * m)(*[j[4].shape[0] for j in jobs]), vp([o[4].data_ptr() for o in outs]),
         vp([o[5].data_ptr() for o in outs]), (ctypes.c_int32 * m)(*[int(j[5]) for j in jobs]), _st())
    for enc_lin, enc_bn, *_ in jobs:
        assert float(enc_bn.eps) == float(bn0.eps) and float(enc_bn.momentum) == float(bn0.momentum)
        if not getattr(enc_bn, "_m3d_flat_counter", False):
            enc_bn.num_batches_tracked += 1
    return outs


USE_LFA_EDGE_ROWS = os.environ.get("M3D_LFA_EDGE_ROWS", "1") != "0"  # A/B switch: 0 = dx by float atomics everywhere
USE_LFA_EDGE_SLOTS = os.environ.get("M3D_LFA_EDGE_SLOTS", "1") != "0"  # A/B switch: 0 = edge rows in edge order (gather through inv)
LFA_BWD_TIMER = None  # bench.py sets {"key": (n, ch) | "keys": {(n, ch), ...}, "events": []}: LFATrainFn.backward then brackets those layers' launches with HIP events (per layer in "by_key")
LFA_FULL = 1  # M3D_LFA_FULL (include/m3d_hip.h): every entry of the neighbour table is a valid row
USE_LFA_FULL = os.environ.get("M3D_LFA_FULL", "1") != "0"  # A/B switch: 0 = the general (masked) kernels everywhere


def lfa_forward(x: Tensor, pos4: Tensor, idx: Tensor, wf: Tensor, bf: Tensor, w_att: Tensor,
                wp: Optional[Tensor] = None, bf16: bool = False, full: bool = False) -> Tensor:
    """``bf16``: ``wp`` holds the bf16 operand fragments and the attention GEMM runs on bf16 matrix cores.
    ``full``: the caller knows that ``idx`` has no -1 padding (every cloud of the level has at least K points:
    ``plan.num_edges[level] == n * K``) — the launch takes the mask-free kernel (``M3D_LFA_FULL``)."""
    n, K = idx.shape
    fl = LFA_FULL if (full and USE_LFA_FULL) else 0
    ch = w_att.shape[0]
    if K > 32:  # the fused kernels tile one centre's neighbours onto <= 2 MFMA row tiles
        if _h(x):
            raise ValueError("bf16 activation storage needs the fused LFA kernels: num_neighbors <= 32")
        return lfa_forward_unfused(x, pos4, idx, wf, bf, w_att)
    out = torch.empty((n, ch), dtype=x.dtype, device=x.device)
    fl |= IO_BF16 if _h(x) else 0  # (x and out hold bf16)
    if bf16:
        assert wp is not None and wp.dtype == torch.int16
        call("m3d_lfa_fwd_bf16", _p(_chka(x)), _p(pos4), _p(idx), n, K, ch, _p(wf), _p(bf), _p(wp), LRELU_SLOPE, _p(out),
             fl | (2 if int(bf16) == 2 else 0), _st())
        return out
    if wp is None:
        wp, _ = pack_attention_weights(w_att, False)  # named local: stays alive until the launch is enqueued
    call("m3d_lfa_fwd", _p(_chka(x)), _p(pos4), _p(idx), n, K, ch, _p(wf), _p(bf), _p(wp), LRELU_SLOPE, _p(out), fl, _st())
    return out


def lfa_forward_unfused(x: Tensor, pos4: Tensor, idx: Tensor, wf: Tensor, bf: Tensor, w_att: Tensor) -> Tensor:
    """Same result through the materialising kernels (cross-check of the fused kernel; any K)."""
    n, K = idx.shape
    ch = w_att.shape[0]
    F = torch.empty((n * K, ch), dtype=torch.float32, device=x.device)
    call("m3d_lfa_edge_features", _p(_chk(x)), _p(pos4), _p(idx), n, K, ch, _p(wf), _p(bf), LRELU_SLOPE, _p(F), _st())
    A = gemm(F, w_att, n * K, ch, ch)
    out = torch.empty((n, ch), dtype=torch.float32, device=x.device)
    call("m3d_lfa_edge_softmax_fwd", _p(A), _p(F), _p(idx), n, K, ch, _p(out), _st())
    return out


class LFATrainFn(torch.autograd.Function):
    """aggregate() of LocalFeatureAggregation in train mode (encoder BatchNorm on batch statistics)."""

    force_unfused_backward = False  # tests flip this to cross-check the fused backward kernel

    @staticmethod
    def forward(ctx, x, pos4, idx, mom, num_edges, enc_w, enc_b, enc_gamma, enc_beta, enc_lin, enc_bn, w_att,
                sinks=None, bf16=False, prepared=None, rev=None, x_slot=None):
        # sinks = (grad_enc_w, grad_enc_b, grad_enc_gamma, grad_enc_beta, grad_w_att) or None
        # prepared = this layer's (wf, bf, mean, invstd, wp, wpt) from lfa_prepare_batch (one launch for all layers)
        # rev = (ptr, inv, slot): reverse neighbour lists of idx (knn_reverse) — where the kernel can store its input gradient
        #       per edge (m3d_lfa_bwd_edge_rows_ok) it writes edge e to row slot[e], i.e. every point's contributions as
        #       CONTIGUOUS rows, and the backward pass sums them per point (gather_sum_rows(inv=None)) instead of 16 / 32-byte
        #       float atomics (round 5: ~30 ps each at the L2, half of the level-1 launches)
        # x_slot (bf16 activation storage): GradSlot shared with the layer that produced x (its y_slot) — an fp32 input
        # gradient (float atomics) travels through it instead of through autograd, which would cast it to bf16
        ctx.x_slot = x_slot
        ctx.rev = rev
        ctx.sinks = sinks
        ctx.side = _grad_side if sinks is not None else None
        x = x.contiguous()
        K = idx.shape[1]
        bf16 = int(bf16) if lfa_bf16_kernels_ok(w_att.shape[0], K) else 0  # 0 fp32, 1 bf16 operands, 2 split-bf16 (three products)
        if bf16 == 2 and not (num_edges == idx.shape[0] * K and USE_LFA_FULL and K in (16, 32)):
            assert prepared is None, "split-bf16 needs complete neighbourhoods: the caller packs fp32 weights otherwise"
            bf16 = 0  # (the split product exists in the complete-neighbourhood kernels only)
        if prepared is not None:
            wf, bf, mean, invstd, wp, wpt = prepared
        elif K <= 32:  # encoder fold + both weight packings: one launch
            wf, bf, mean, invstd, wp, wpt = lfa_prepare(enc_lin, enc_bn, mom, num_edges, w_att, bf16, True)
        else:
            wf, bf, mean, invstd = lfa_enc_fold(enc_lin, enc_bn, mom, num_edges)
            wp = wpt = None
        # complete neighbourhoods (every cloud has >= K points: what the plan's edge count says) take the mask-free kernels
        ctx.full = bool(num_edges == idx.shape[0] * K) and USE_LFA_FULL
        out = lfa_forward(x, pos4, idx, wf, bf, w_att, wp, bf16=bf16, full=ctx.full)
        ctx.packed = (wp, wpt)
        ctx.bf16 = bf16
        ctx.save_for_backward(x, pos4, idx, mom, wf, bf, mean, invstd, enc_w, enc_b, enc_gamma, w_att)
        ctx.num_edges = num_edges
        return out

    @staticmethod
    def backward(ctx, dout):
        x, pos4, idx, mom, wf, bf, mean, invstd, enc_w, enc_b, enc_gamma, w_att = ctx.saved_tensors
        n, K = idx.shape
        ch = w_att.shape[0]
        D = ch // 2
        dev = x.device
        sk = ctx.sinks
        dout = dout.contiguous()
        io = IO_BF16 if _h(x) else 0  # x, dout and the edge rows hold bf16; the atomically accumulated dx stays fp32
        assert dout.dtype == x.dtype, (dout.dtype, x.dtype)
        edge_rows = bool(ctx.rev is not None and ctx.full and not ctx.bf16 and USE_LFA_EDGE_ROWS and K <= 32 and
                         not LFATrainFn.force_unfused_backward and lib().m3d_lfa_bwd_edge_rows_ok(n, K, ch, LRELU_SLOPE))
        dx = torch.empty((n * K, D), dtype=x.dtype, device=dev) if edge_rows else arena.zeros((n, D), torch.float32, dev)
        if K <= 32 and not LFATrainFn.force_unfused_backward:
            G = arena.zeros((11 * D,), torch.float64, dev)  # (pre-zeroed: flag bit 1 below skips the memset)
            dw_att = sk[4] if sk else torch.empty((ch, ch), dtype=torch.float32, device=dev)
            ws = torch.empty(lib().m3d_lfa_bwd_workspace_bytes(n, K, ch), dtype=torch.uint8, device=dev)
            wp, wpt = ctx.packed  # packed in the forward pass (one launch for both orientations)
            defer = sk is not None and ctx.side is not None and DEFER_WGRAD
            tm = LFA_BWD_TIMER  # (bench.py: HIP events around ONE layer's launch inside real training steps)
            ev = None
            if tm is not None and ((n, ch) == tm.get("key") or (n, ch) in tm.get("keys", ())):
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
            fl = (1 if sk is not None else 0) | 2 | (4 if defer else 0) | (8 if ctx.full else 0) | (16 if ctx.bf16 == 2 else 0) | io
            if edge_rows:
                slot = ctx.rev[2] if (len(ctx.rev) > 2 and (USE_LFA_EDGE_SLOTS or ctx.rev[1] is None)) else None
                call("m3d_lfa_bwd_edge_rows", _p(x), _p(pos4), _p(idx), n, K, ch, _p(wf), _p(bf), _p(wp), _p(wpt), LRELU_SLOPE,
                     _p(dout), _p(dx), _p(slot) if slot is not None else None, _p(dw_att), fl, _p(G), _p(ws), _st())
                dx = gather_sum_rows(dx, ctx.rev[0], None if slot is not None else ctx.rev[1], n, long_lists=True)
            else:
                call("m3d_lfa_bwd_bf16" if ctx.bf16 else "m3d_lfa_bwd", _p(x), _p(pos4), _p(idx), n, K, ch, _p(wf), _p(bf),
                     _p(wp), _p(wpt), LRELU_SLOPE, _p(dout), _p(dx), _p(dw_att), fl, _p(G), _p(ws), _st())
            if ev is not None:
                ev[1].record()
                tm["events"].append(ev)
                tm.setdefault("by_key", {}).setdefault((n, ch), []).append(ev)
                tm["flags"] = (1 if sk is not None else 0) | 2 | (4 if defer else 0) | (8 if ctx.full else 0)
            if defer:
                # dW_att, G and the encoder parameter gradients are leaves: summed / finished with every other LFA
                # layer's at the end of the backward pass (GradSideStream.flush) instead of two launches in the chain
                ctx.side.defer_lfa((n, K, ch, ws, dw_att, G, mom, ctx.num_edges, enc_w, enc_b, enc_gamma, mean, invstd,
                                    sk[0], sk[1], sk[2], sk[3]))
                return (LFATrainFn._dx_out(ctx, x, dx),) + (None,) * 16
        else:
            if io:
                raise ValueError("bf16 activation storage needs the fused LFA backward kernel (num_neighbors <= 32)")
            G = torch.empty(11 * D, dtype=torch.float64, device=dev)
            dw_att = _lfa_backward_unfused(x, pos4, idx, wf, bf, w_att, dout, dx, G)
            if sk:
                sk[4].add_(dw_att)
        if sk:
            dw, db, dgamma, dbeta = sk[0], sk[1], sk[2], sk[3]
        else:
            dw = torch.empty((D, 10), dtype=torch.float32, device=dev)
            db, dgamma, dbeta = (torch.empty(D, dtype=torch.float32, device=dev) for _ in range(3))
        call("m3d_lfa_enc_bwd_finalize", _p(G), _p(mom), ctx.num_edges, _p(enc_w), _p(enc_b), _p(enc_gamma), _p(mean),
             _p(invstd), _p(dw), _p(db), _p(dgamma), _p(dbeta), D, int(sk is not None), _st())
        dx = LFATrainFn._dx_out(ctx, x, dx)
        if sk:
            return (dx,) + (None,) * 16
        return dx, None, None, None, None, dw, db, dgamma, dbeta, None, None, dw_att, None, None, None, None, None

    @staticmethod
    def _dx_out(ctx, x, dx):
        """What autograd gets as the gradient of ``x``.  bf16 storage with an fp32 ``dx`` (float atomics): the buffer goes to the
        producer through ``x_slot`` and autograd sees an uninitialised bf16 placeholder (no cast kernel); without a slot the
        engine casts it (correct, one elementwise launch)."""
        if _h(x) and not _h(dx) and ctx.x_slot is not None:
            ctx.x_slot.buf = dx
            return torch.empty(x.shape, dtype=x.dtype, device=x.device)
        return dx


def _lfa_backward_unfused(x, pos4, idx, wf, bf, w_att, dout, dx, G):
    """Backward through materialised [E, ch] tensors (any K; cross-check of ``m3d_lfa_bwd``)."""
    n, K = idx.shape
    ch = w_att.shape[0]
    dev = x.device
    E = n * K
    F = torch.empty((E, ch), dtype=torch.float32, device=dev)
    call("m3d_lfa_edge_features", _p(x), _p(pos4), _p(idx), n, K, ch, _p(wf), _p(bf), LRELU_SLOPE, _p(F), _st())
    A = gemm(F, w_att, E, ch, ch)
    dF = torch.empty((E, ch), dtype=torch.float32, device=dev)
    call("m3d_lfa_edge_softmax_bwd", _p(A), _p(F), _p(idx), n, K, ch, _p(dout), _p(dF), _st())
    dA = A
    dw_att = torch.zeros((ch, ch), dtype=torch.float32, device=dev)  # dW_att[c,k] = sum_e dA[e,c] F[e,k]
    gemm(dA, F, ch, ch, E, lda0=ch, a_cm=True, b_cm=True, ldb=ch, out=dw_att, accumulate=True,
         splitk=_splitk_for(E, ch, ch))
    gemm(dA, w_att, E, ch, ch, b_cm=True, ldb=w_att.stride(0), out=dF, accumulate=True)  # dF += dA W_att
    call("m3d_lfa_edge_features_bwd", _p(dF), _p(pos4), _p(idx), n, K, ch, _p(wf), _p(bf), LRELU_SLOPE, _p(dx), _p(G),
         _st())
    return dw_att


# --------------------------------------------------------------------------------------------------
# knn_interpolate (model.py:90-98, pyg_randla_net.py:250)
# --------------------------------------------------------------------------------------------------
def idw_interpolate(x: Tensor, idx: Tensor, d2: Tensor) -> Tensor:
    nq, k = idx.shape
    y = torch.empty((nq, x.shape[1]), dtype=torch.float32, device=x.device)
    call("m3d_idw_interpolate_fwd", _p(_chk(x)), x.stride(0), _p(idx), _p(d2), nq, k, x.shape[1], _p(y), _st())
    return y


# --------------------------------------------------------------------------------------------------
# training step: loss (model.py:118) and optimizer (configs/model/optimizer/Adam.yaml)
# --------------------------------------------------------------------------------------------------
class CrossEntropyFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, ignore_index):
        logits = _chk(logits.contiguous())
        n, C = logits.shape
        dev = logits.device
        lse = torch.empty(n, dtype=torch.float32, device=dev)
        pre_zeroed = arena.active and arena.buf is not None
        acc = arena.zeros((516,), torch.float64, dev) if pre_zeroed else torch.empty(516, dtype=torch.float64, device=dev)  # M3D_CE_ACC_DOUBLES
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        call("m3d_ce_loss_fwd", _p(logits), logits.stride(0), _p(target), n, C, ignore_index, _p(lse), _p(acc), _p(loss),
             1 if pre_zeroed else 0, _st())
        ctx.save_for_backward(logits, target, lse, acc)
        ctx.ignore_index = ignore_index
        return loss.view(())

    @staticmethod
    def backward(ctx, gout):
        logits, target, lse, acc = ctx.saved_tensors
        n, C = logits.shape
        d = torch.empty_like(logits)
        gout = gout.reshape(1).to(torch.float32).contiguous()
        call("m3d_ce_loss_bwd", _p(logits), logits.stride(0), _p(target), n, C, ctx.ignore_index, _p(lse), _p(acc),
             _p(gout), _p(d), _st())
        return d, None, None


class DropoutFn(torch.autograd.Function):
    """``torch.nn.functional.dropout(x, p, training=True)`` with a counter-based mask (``m3d_dropout``): ``counter`` is a
    device int64 the net advances once per training forward, so a replayed hipGraph draws a fresh mask every step and the
    backward pass recomputes the forward's mask instead of reading a stored one."""

    @staticmethod
    def forward(ctx, x, p, counter, seed):
        x = _chk(x.contiguous())
        y = torch.empty_like(x)
        call("m3d_dropout", _p(x), _p(y), x.numel(), float(p), _p(counter), int(seed), _st())
        # (the counter's value at THIS forward: a later train-mode forward advances the live one before our backward runs)
        ctx.args = (float(p), counter.clone(), int(seed))
        return y

    @staticmethod
    def backward(ctx, dy):
        p, counter, seed = ctx.args
        dy = _chk(dy.contiguous())
        dx = torch.empty_like(dy)
        call("m3d_dropout", _p(dy), _p(dx), dy.numel(), p, _p(counter), seed, _st())
        return dx, None, None, None


def cross_entropy(logits: Tensor, target: Tensor, ignore_index: int = -100) -> Tensor:
    """``torch.nn.functional.cross_entropy(logits, target, ignore_index=..., reduction="mean")`` for ``[n, C]``
    fp32 logits and int64 class targets (the reference's criterion, model.py:118)."""
    assert target.dtype == torch.int64 and target.is_cuda and target.is_contiguous()
    return CrossEntropyFn.apply(logits, target, int(ignore_index))


# --------------------------------------------------------------------------------------------------
# PointNet++ set-abstraction variant (BASELINE.json configs[4]): farthest-point sampling, grouping, max aggregation.
# No reference implementation exists (myria3d/models/model.py:12); csrc/sa.hip restates the published operators.
# --------------------------------------------------------------------------------------------------
def fps(pos4: Tensor, ptr: Tensor, ptr_out: Tensor, m: int, max_points: int, start: Optional[Tensor] = None,
        index: Optional["KnnIndex"] = None) -> Tensor:
    """Farthest-point sampling inside each cloud (``torch_cluster.fps`` semantics): int32 ``[m]`` global rows in selection
    order; cloud ``b`` keeps ``ptr_out[b+1] - ptr_out[b]`` points starting from ``start[b]`` (cloud-relative; default 0).
    ``index``: the built ``KnnIndex`` of the same points and ``ptr`` — clouds of 16 385 ... 40 000 points are then sampled with
    exact bucket skipping over its cell-sorted records (``m3d_fps_sorted``; same index lists, 2x faster at 40 000 points)."""
    assert pos4.shape[1] == 4 and pos4.is_contiguous()
    idx = torch.empty(m, dtype=torch.int32, device=pos4.device)
    if start is not None:
        assert start.dtype == torch.int32 and start.is_contiguous() and start.numel() == ptr.numel() - 1
    B = ptr.numel() - 1
    if index is not None and FPS_BUCKET_MIN < max_points <= 40000:
        assert index.n == pos4.shape[0] and index.num_clouds == B
        call("m3d_fps_sorted", _p(index.ws), index.n, _p(ptr), _p(ptr_out), B, int(max_points), _p(start), _p(idx), _st())
        return idx
    call("m3d_fps", _p(_chk(pos4)), _p(ptr), _p(ptr_out), B, int(max_points), _p(start), _p(idx), _st())
    return idx


FPS_BUCKET_MIN = 16384  # largest cloud the register-resident sampler handles (m3d_fps): the bucket-skipping one takes over above


class SAGroupFn(torch.autograd.Function):
    """Edge rows ``[x_j | pos_j - pos_i | 0-pad]`` of a set-abstraction level over the compact edge list ``seg``
    (PointNetConv.message's gathers).  Returns ``(rows [E, ldo], esrc int32 [E], ectr int32 [E])``."""

    @staticmethod
    def forward(ctx, x, pos4_src, pos4_ctr, nbr, seg, num_edges, ldo):
        x = _chk(x.contiguous())
        m, K = nbr.shape
        C = x.shape[1]
        dev = x.device
        out = torch.empty((num_edges, ldo), dtype=torch.float32, device=dev)
        esrc = torch.empty(num_edges, dtype=torch.int32, device=dev)
        ectr = torch.empty(num_edges, dtype=torch.int32, device=dev)
        call("m3d_sa_group", _p(x), x.stride(0), C, _p(_chk(pos4_src)), _p(_chk(pos4_ctr)), _p(_chk(nbr, torch.int32)),
             _p(seg), m, K, _p(out), ldo, _p(esrc), _p(ectr), _st())
        ctx.save_for_backward(esrc)
        ctx.shape = (x.shape[0], C)
        ctx.mark_non_differentiable(esrc, ectr)
        return out, esrc, ectr

    @staticmethod
    def backward(ctx, dout, _a, _b):
        if not ctx.needs_input_grad[0]:  # set-abstraction level 1 reads the raw input features: nobody wants this gradient
            return None, None, None, None, None, None, None  # (E = m K rows of atomics were thrown away every step)
        (esrc,) = ctx.saved_tensors
        n, C = ctx.shape
        dout = dout.contiguous()
        dx = torch.zeros((n, C), dtype=torch.float32, device=dout.device)
        call("m3d_sa_group_bwd", _p(dout), dout.stride(0), _p(esrc), esrc.numel(), C, _p(dx), C, _st())
        return dx, None, None, None, None, None, None


class SegMaxFn(torch.autograd.Function):
    """``aggr="max"`` over the edges of each centre (torch_scatter.scatter_max: gradient to the arg-max edge)."""

    @staticmethod
    def forward(ctx, y, seg, ectr, m):
        y = _chk(y.contiguous())
        E, C = y.shape
        out = torch.empty((m, C), dtype=torch.float32, device=y.device)
        arg = torch.empty((m, C), dtype=torch.int32, device=y.device)
        call("m3d_seg_max", _p(y), y.stride(0), _p(seg), m, C, _p(out), _p(arg), _st())
        ctx.save_for_backward(arg, seg, ectr)
        ctx.E = E
        return out

    @staticmethod
    def backward(ctx, dout):
        arg, seg, ectr = ctx.saved_tensors
        dout = _chk(dout.contiguous())
        C = dout.shape[1]
        dy = torch.empty((ctx.E, C), dtype=torch.float32, device=dout.device)
        call("m3d_seg_max_bwd", _p(dout), _p(arg), _p(seg), _p(ectr), ctx.E, C, _p(dy), _st())
        return dy, None, None, None
