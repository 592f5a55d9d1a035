"""``HipRandLANet`` — MI355X-native drop-in for ``myria3d.models.modules.pyg_randla_net.PyGRandLANet``.

Same constructor keywords, same ``forward(x, pos, batch, ptr)``, same ``state_dict`` keys/shapes as the reference
(``/root/reference/myria3d/models/modules/pyg_randla_net.py:22-88``; key names per PyG's ``MLP``/``BatchNorm``
wrappers), so ``Model.load_from_checkpoint`` and the Hydra config surface keep working.  All
per-point arithmetic runs in the hand-written HIP kernels of ``libm3d_hip.so``; the ``torch.nn`` modules below
only *hold* parameters and buffers.
"""
from __future__ import annotations

import warnings
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F
from torch import Tensor, nn

from . import ops


# ----------------------------------------------------------------------------------------------
# parameter containers mirroring PyG's module tree (keys: lins.i.{weight,bias}, norms.i.module.*)
# ----------------------------------------------------------------------------------------------
class _BatchNormHolder(nn.Module):
    def __init__(self, channels: int):
        super().__init__()
        self.module = nn.BatchNorm1d(channels, eps=ops.BN_EPS, momentum=ops.BN_MOMENTUM)


class SharedMLPParams(nn.Module):
    """Parameters of ``SharedMLP(channels, ...)`` (pyg_randla_net.py:97-109)."""

    def __init__(self, channels: Sequence[int], act: bool = True, norm: bool = True, bias: bool = True,
                 dropout: Optional[Sequence[float]] = None):
        super().__init__()
        nl = len(channels) - 1
        self.act, self.has_norm = act, norm
        self.dropout = list(dropout) if dropout is not None else [0.0] * nl
        self.lins = nn.ModuleList([nn.Linear(channels[i], channels[i + 1], bias=bias) for i in range(nl)])
        self.norms = nn.ModuleList([_BatchNormHolder(channels[i + 1]) if norm else nn.Identity() for i in range(nl)])


class LFAParams(nn.Module):
    def __init__(self, channels: int):
        super().__init__()
        self.mlp_encoder = SharedMLPParams([10, channels // 2])
        self.mlp_attention = SharedMLPParams([channels, channels], act=False, norm=False, bias=False)
        self.mlp_post_attention = SharedMLPParams([channels, channels])


class BlockParams(nn.Module):
    def __init__(self, d_in: int, d_out: int):
        super().__init__()
        self.mlp1 = SharedMLPParams([d_in, d_out // 8])
        self.shortcut = SharedMLPParams([d_in, d_out], act=False)
        self.mlp2 = SharedMLPParams([d_out // 2, d_out], act=False)
        self.lfa1 = LFAParams(d_out // 4)
        self.lfa2 = LFAParams(d_out // 2)


class FPParams(nn.Module):
    def __init__(self, mlp: SharedMLPParams):
        super().__init__()
        self.nn = mlp


# ----------------------------------------------------------------------------------------------
@dataclass
class LevelPlan:
    """Host-side description of a batch at every resolution level (sizes are known from ``ptr``)."""

    sizes: List[List[int]]  # per level, per cloud
    ptrs: List[Tensor]  # device int64 [B+1] per level
    totals: List[int]
    num_edges: List[int] = field(default_factory=list)  # valid kNN edges per encoder level
    staging: Optional[Tensor] = None  # pinned host copy of ``ptrs`` (source of the asynchronous upload)
    ready: Optional[object] = None  # event behind that upload, None once it is known to have completed


def make_plan(ptr_host: Sequence[int], decimation: int, num_neighbors: int, device, levels: int = 4) -> LevelPlan:
    sizes = [[int(ptr_host[i + 1]) - int(ptr_host[i]) for i in range(len(ptr_host) - 1)]]
    for _ in range(levels):
        sizes.append([max(1, n // decimation) for n in sizes[-1]])  # pyg_randla_net.py:215-217
    flat, totals = [], []
    for s in sizes:
        p = 0
        flat.append(0)
        for n in s:
            p += n
            flat.append(p)
        totals.append(p)
    # the per-level ``ptr`` vectors are ONE device tensor, uploaded by ONE copy.  On a GPU the copy leaves from pinned
    # memory without waiting for the stream (``torch.tensor(list, device=cuda)`` is a blocking copy behind everything
    # enqueued so far: five of them per layout, and a loop that builds its plans from host-side tile sizes —
    # INTEGRATION.md section 3 — would stop at each one instead of running ahead of the device)
    dev = torch.device(device)
    if dev.type == "cuda":
        staging = torch.empty(len(flat), dtype=torch.int64, pin_memory=True)
        staging.copy_(torch.tensor(flat, dtype=torch.int64))
        allp = staging.to(dev, non_blocking=True)
    else:
        staging = None
        allp = torch.tensor(flat, dtype=torch.int64, device=dev)
    nb = len(sizes[0]) + 1
    ptrs = [allp[i * nb:(i + 1) * nb] for i in range(len(sizes))]
    num_edges = [sum(n * min(num_neighbors, n) for n in s) for s in sizes[:levels]]
    plan = LevelPlan(sizes, ptrs, totals, num_edges)
    plan.staging = staging  # (kept until the plan goes: the copy may still be in flight)
    if staging is not None:
        with torch.cuda.device(dev):  # (the copy went to ``dev``'s current stream, whichever device is current here)
            plan.ready = torch.cuda.Event()
            plan.ready.record()  # consumers on OTHER streams wait for it (plan_ready): the blocking copy used to cover them
    return plan


def _check_plan(plan: LevelPlan, pos: Tensor, ptr: Tensor) -> None:
    """A caller-supplied plan must describe THIS batch: the mask-free kernels are chosen from its host-side edge
    counts and the per-level launches from its totals, so a plan of another layout would read and write out of bounds where
    the plan-less path gave masked results.  Host-side arithmetic only — the point of handing over a plan is that nothing
    is read back from the device; the tile SIZES behind equal totals are the caller's word."""
    if plan.totals[0] != pos.shape[0] or len(plan.sizes[0]) != ptr.numel() - 1:
        raise ValueError(f"forward(plan=...): the plan describes {len(plan.sizes[0])} clouds / {plan.totals[0]} points, the "
                         f"batch has {ptr.numel() - 1} clouds / {pos.shape[0]} points")


def plan_ready(plan: LevelPlan) -> None:
    """Order the current stream behind the upload of ``plan.ptrs`` (a no-op once the copy has completed, and inside a
    stream capture — plans are built before a capture begins)."""
    ev = plan.ready
    if ev is None:
        return
    main = torch.cuda.current_stream()
    if ops.capture_id(main) != 0:
        return  # (no event query inside a capture; ``torch.cuda.graph`` synchronises the device before it begins)
    if ev.query():
        plan.ready = None
        return
    main.wait_event(ev)


class HipRandLANet(nn.Module):
    """RandLA-Net for batched variable-size point clouds on one MI355X.

    Args (identical to the reference, pyg_randla_net.py:23-30):
        num_features, num_classes, decimation=4, num_neighbors=16, return_logits=False
    """

    def __init__(self, num_features: int, num_classes: int, decimation: int = 4, num_neighbors: int = 16,
                 return_logits: bool = False):
        super().__init__()
        self.decimation = decimation
        self.num_neighbors = num_neighbors
        self.return_logits = return_logits
        d_bottleneck = max(32, num_classes, num_features)
        self.fc0 = nn.Linear(num_features, d_bottleneck)
        self.block1 = BlockParams(d_bottleneck, 32)
        self.block2 = BlockParams(32, 128)
        self.block3 = BlockParams(128, 256)
        self.block4 = BlockParams(256, 512)
        self.mlp_summit = SharedMLPParams([512, 512])
        self.fp4 = FPParams(SharedMLPParams([512 + 256, 256]))
        self.fp3 = FPParams(SharedMLPParams([256 + 128, 128]))
        self.fp2 = FPParams(SharedMLPParams([128 + 32, 32]))
        self.fp1 = FPParams(SharedMLPParams([32 + 32, d_bottleneck]))
        self.mlp_classif = SharedMLPParams([d_bottleneck, 64, 32], dropout=[0.0, 0.5])
        self.fc_classif = nn.Linear(32, num_classes)
        # device-side RNG state for decimation (bumped every forward; hipGraph-replay safe).  Seeded lazily, on the
        # first forward, from torch's global seed (``torch.manual_seed`` / ``seed_everything``) mixed with the rank —
        # the reference draws its ``torch.randperm`` from the global generator (pyg_randla_net.py:221), so there too
        # the seed controls decimation and every DDP rank draws its own permutations
        self.register_buffer("_decim_seed", torch.tensor([0x5DEECE66D], dtype=torch.int64), persistent=False)
        self._decim_seeded = False
        self._plans: Dict[tuple, LevelPlan] = {}
        self._grad_eval = False  # eval-mode forward that records an autograd graph (set per call)
        self._use_sinks = False
        self._streams: Dict = {}
        self._plan_ident = None  # up to four (identity of a ptr tensor read lately, weak reference to it, its plan)
        # eval-mode derived tensors (folded BatchNorm scale/shift, folded encoder, packed attention weights) depend on
        # parameters / running statistics only: cached across forwards, dropped whenever those may have changed
        self._eval_cache: Dict = {}
        # "fp32": the reference's arithmetic.  "bf16": the matrix-bound layers (LFA attention GEMMs at ch >= 64, SharedMLP
        # GEMMs with more than 64 input channels, forward and input gradient) take bf16
        # operands on the matrix cores with fp32 accumulation (BASELINE config 2); storage, positions, kNN, softmax and
        # the statistics stay fp32.  Also switched on by torch.autocast(dtype=bfloat16) around the call (Lightning's
        # ``trainer.precision: bf16-mixed``), like any autocast-aware module.
        self.matmul_precision = "fp32"
        # torch.bfloat16: every feature matrix of the pass — fc0's output to the classifier's last hidden layer — and its
        # gradient live in HBM as bf16 (round 6; BASELINE config 2 "bf16", the storage half of Lightning's ``precision: bf16``,
        # configs/experiment/RandLaNet_base_run_FR-2x3GPUs.yaml:12): the HBM-bound SharedMLP / BatchNorm / weight-gradient chain
        # of levels 1-2 moves half the bytes.  Parameters, statistics (fp64 sums of the fp32 accumulators), positions, kNN,
        # softmax, the logits and every parameter gradient stay fp32; arithmetic is fp32 registers either way.  Usually paired
        # with ``matmul_precision = "bf16"`` (operands of the matrix-bound layers rounded to bf16 for the matrix cores).
        self.activation_dtype = torch.float32
        self.overlap_geometry = True  # run the position-only work (kNN, decimation) on a side stream
        # K-NN / 1-NN queries of a PREFETCH (the next batch's tables beside the current step: prefetch_geometry, graph A of a
        # GraphedStep) take at most this many x 64 wavefronts (0: no cap).  The level-1 query is 3 200 wavefronts of pure VALU
        # work that otherwise flood every CU for 140-200 us and make the step's latency-bound chain queue up behind them; at
        # one wave per SIMD it takes ~2 x as long — hidden, the prefetch has the whole step — and costs the step 0.05 ms less
        # (3.962 -> 3.915 ms, same box: profiles/r06i_knn_background_cap_ab.log).  Tables are bit-identical.
        self.background_knn_cap = int(__import__("os").environ.get("M3D_KNN_BG_CAP", "16"))
        # (the prefetch beside a graphed EVAL forward: the chain has less slack there — 24 x 64 wavefronts: 0.989 -> 0.978 ms,
        # bf16 0.759 -> 0.742; 16: slower, 1.02; 0 = uncapped.  profiles/r06v_*)
        self.background_knn_cap_eval = int(__import__("os").environ.get("M3D_KNN_BG_CAP_EVAL", "24"))
        # the input gradients of a tensor with several consumers meet in one buffer (ops.GradSlot) instead of autograd's
        # accumulation adds; False: plain autograd (cross-check)
        self.share_input_gradients = __import__("os").environ.get("M3D_GRAD_SLOTS", "1") != "0"
        self.batch_lfa_prepare = __import__("os").environ.get("M3D_LFA_PREP_BATCH", "1") != "0"  # (A/B switch)
        # the K-NN tables / encoder moments / decoder 1-NN tables of the four levels as one launch each (see
        # _geometry_stages); M3D_GEO_BATCH=0: level by level (A/B and cross-check)
        self.batch_geometry = __import__("os").environ.get("M3D_GEO_BATCH", "1") != "0"
        self.batch_geometry_eval_capture = __import__("os").environ.get("M3D_GEO_BATCH_EVAL", "1") != "0"
        self.grad_side: Optional[ops.GradSideStream] = None  # weight-gradient side stream (owned by FusedAdam)
        self._flat: Optional[tuple] = None  # (flat_params, flat_grads) once flatten_parameters() has run
        # geometry of the NEXT forward (see prefetch_geometry()): two persistent slots used in turn
        # persistent buffer sets, keyed (owner, train flag, 0 / 1): a GraphedStep owns its pair (captured graphs hold
        # the addresses), the train- and eval-mode layouts differ (encoder moments), plain callers share owner None
        self._look_slots: Dict[tuple, "_GeoSlot"] = {}
        self._look_queue: List[tuple] = []  # slots holding prefetched geometry nobody has consumed yet, oldest first
        self._look_turn = 1
        self.interleave_paced = bool(int(__import__("os").environ.get("M3D_INTERLEAVE_PACED", "0")))
        self._look_job = None  # an interleaved prefetch in progress: (stage generator, geometry, slot, key, pos, stream)
        self._fwd_start = None  # event: start of the most recent forward (prefetch_geometry(after="forward_start"))
        self._fwd_count = 0  # forwards started so far (which forward consumed a slot: _GeoSlot.consumer_fwd)
        self._bf16 = False
        # a parent module's load_state_dict() reaches this module through _load_from_state_dict only
        self.register_load_state_dict_post_hook(lambda module, incompatible: module._eval_cache.clear())

    def set_decimation_seed(self, seed: int) -> None:
        """Fix the device-side decimation RNG state (otherwise derived from ``torch.initial_seed()`` and the rank on the
        first forward)."""
        seed = (int(seed) * 0x9E3779B97F4A7C15 + 0x5DEECE66D) & ((1 << 63) - 1)
        with torch.no_grad():
            self._decim_seed.fill_(seed)
        self._decim_seeded = True

    def _dropout_seed(self) -> int:
        s = getattr(self, "_drop_seed", None)
        if s is None:
            import torch.distributed as dist

            rank = dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0
            s = self._drop_seed = ((torch.initial_seed() & 0xFFFFFFFFFFFF) * 1024 + rank) * 0x9E3779B97F4A7C15 % (1 << 64)
        return s

    def _seed_decimation(self) -> None:
        if self._decim_seeded:
            return
        if torch.cuda.is_current_stream_capturing():
            return  # a fill recorded into a graph would reset the state at every replay: keep the current value
        import torch.distributed as dist

        rank = dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0
        self.set_decimation_seed((torch.initial_seed() & 0xFFFFFFFFFFFF) * 1024 + rank)

    # ------------------------------------------------------------------------------------------
    # flat parameter / gradient buffers (opt-in): every parameter becomes a view of ONE fp32 buffer and every
    # .grad a view of a second one.  The backward kernels then accumulate parameter gradients straight into that
    # buffer ("gradient sinks"; autograd sees None for them), RCCL all-reduces it as a single 4.45 MB bucket and
    # m3d_adam_step updates the whole model in one launch.  state_dict() / load_state_dict() are unaffected.
    def flatten_parameters(self) -> "HipRandLANet":
        params = list(self.parameters())
        dev = params[0].device
        if any(p.dtype != torch.float32 or p.device != dev for p in params):
            raise RuntimeError("flatten_parameters(): all parameters must be fp32 on one device")
        # (host tensors are accepted: the bucket layout / broadcast / all-reduce logic is device-agnostic and is
        # exercised by the world-size-2 gloo test; the kernels that USE the bucket exist on the HIP device only)
        sizes = [(p.numel() + 3) // 4 * 4 for p in params]  # every slice stays 16-byte aligned
        flat_p = torch.zeros(sum(sizes), dtype=torch.float32, device=dev)
        flat_g = torch.zeros_like(flat_p)
        off = 0
        with torch.no_grad():
            for p, sz in zip(params, sizes):
                n = p.numel()
                flat_p[off:off + n].copy_(p.data.reshape(-1))
                p.data = flat_p[off:off + n].view(p.shape)
                p.grad = flat_g[off:off + n].view(p.shape)
                off += sz
        # BatchNorm step counters: views of one int64 vector, bumped by a single add per training forward
        bns = [m for m in self.modules() if isinstance(m, nn.BatchNorm1d)]
        # (one extra slot at the end: the step counter of the classifier's dropout mask, ops.DropoutFn)
        self._nbt_flat = torch.stack([m.num_batches_tracked.to(dev) for m in bns]
                                     + [torch.zeros((), dtype=torch.int64, device=dev)]).contiguous()
        for i, m in enumerate(bns):
            m.num_batches_tracked = self._nbt_flat[i]  # registered buffer: same state_dict key, now a view
            m._m3d_flat_counter = True
        self._flat = (flat_p, flat_g)
        self._flat_ends = (params[0], params[-1])
        # (owner module, name, parameter, offset): _check_flat runs every training forward — walking the module tree
        # (self.parameters()) cost ~0.4 ms of host time per step there
        owners = {id(p): (m, n) for m in self.modules() for n, p in m._parameters.items() if p is not None}
        self._flat_index, off = [], 0
        for p, sz in zip(params, sizes):
            self._flat_index.append((*owners[id(p)], p, off))
            off += sz
        return self

    @property
    def flat_parameters(self) -> Optional[Tensor]:
        return self._flat[0] if self._flat is not None else None

    @property
    def flat_grads(self) -> Optional[Tensor]:
        return self._flat[1] if self._flat is not None else None

    def _check_flat(self) -> bool:
        """True when the gradient sinks can be used for this step; repairs detached .grad views
        (``zero_grad(set_to_none=True)``) and re-flattens after ``.to()`` / ``load_state_dict(assign=True)``."""
        if self._flat is None:
            return False
        flat_p, flat_g = self._flat
        ok, lost_grad = True, False
        bp, bg = flat_p.data_ptr(), flat_g.data_ptr()
        for mod, name, p, off in self._flat_index:
            if mod._parameters.get(name) is not p or p.data_ptr() != bp + 4 * off:
                ok = False  # moved (.to()) or replaced (load_state_dict(assign=True))
                break
            g = p.grad
            if g is None or g.data_ptr() != bg + 4 * off:
                lost_grad = True
        if not ok:
            self.flatten_parameters()
            return True
        if lost_grad:
            flat_g.zero_()
            off = 0
            for p in self.parameters():
                n = p.numel()
                p.grad = flat_g[off:off + n].view(p.shape)
                off += (n + 3) // 4 * 4
        return True

    def _flat_intact(self) -> bool:
        """O(1) variant of ``_check_flat`` for the optimizer's hot path: first and last parameter still sit in the
        flat buffers (a ``.to()`` or ``zero_grad(set_to_none=True)`` moves / detaches all of them)."""
        if self._flat is None:
            return False
        flat_p, flat_g = self._flat
        first, last = self._flat_ends
        return (first.data_ptr() == flat_p.data_ptr() and first.grad is not None
                and first.grad.data_ptr() == flat_g.data_ptr() and last.grad is not None
                and last.data_ptr() + 4 * last.numel() <= flat_p.data_ptr() + 4 * flat_p.numel()
                and last.data_ptr() >= flat_p.data_ptr())

    @staticmethod
    def _sinks(*params):
        return tuple(p.grad for p in params)

    def invalidate_eval_cache(self) -> None:
        """Call after changing parameters or running statistics behind the module's back (raw-pointer writes)."""
        self._eval_cache.clear()

    def _cached(self, key, fn, deps=()):
        """``deps``: the parameters / buffers the cached value was derived from.  Their autograd version counters are
        part of the entry, so in-place updates through torch (``load_state_dict`` of a parent module, an optimizer
        step, EMA / SWA swaps with ``copy_``) are noticed; writes through ``.data`` or raw pointers are not —
        ``invalidate_eval_cache()`` is for those (``FusedAdam.step`` calls it)."""
        stamp = tuple((t.data_ptr(), t._version) for t in deps)
        hit = self._eval_cache.get(key)
        if hit is None or hit[0] != stamp:
            hit = self._eval_cache[key] = (stamp, fn())
        return hit[1]

    @staticmethod
    def _bn_deps(bn):
        return (bn.weight, bn.bias, bn.running_mean, bn.running_var)

    def train(self, mode: bool = True):
        if mode or self.training:  # entering or leaving a training phase: weights / running statistics change
            self._eval_cache.clear()
        if bool(mode) != self.training:
            # geometry prefetched for the other mode (train-mode sets carry the encoder moments) is never consumed
            self._finish_interleaved()
            self._look_queue.clear()
        return super().train(mode)

    def _apply(self, fn, recurse=True):
        self._eval_cache.clear()
        return super()._apply(fn, recurse)

    def load_state_dict(self, *args, **kwargs):
        self._eval_cache.clear()
        return super().load_state_dict(*args, **kwargs)

    # ------------------------------------------------------------------------------------------
    def plan_for(self, ptr: Tensor) -> LevelPlan:
        """One device->host read of ``ptr`` per forward (the reference syncs 2*B times per level,
        pyg_randla_net.py:219-229); cached by tile sizes."""
        # the SAME tensor asked twice (prefetch_geometry for the next batch, then its forward) is read from the device once:
        # a second .tolist() is a second host sync (0.45 ms of host time per step on the variable-layout path)
        # (the last FOUR tensors are remembered: with a prefetch for the next batch in front of every forward, two alternate)
        ident = (ptr.data_ptr(), ptr._version, ptr.numel(), ptr.device)
        for ent in self._plan_ident or ():
            if ent[0] == ident and ent[1]() is ptr:
                return ent[2]
        key = tuple(ptr.tolist())
        plan = self._plans.get(key)
        if plan is None:
            if len(self._plans) > 64:
                self._plans.clear()
            plan = make_plan(key, self.decimation, self.num_neighbors, ptr.device)
            self._plans[key] = plan
        import weakref

        try:
            self._plan_ident = [(ident, weakref.ref(ptr), plan)] + list(self._plan_ident or ())[:3]
        except TypeError:
            pass
        return plan

    def plan_from_host_sizes(self, ptr_host: Sequence[int]) -> LevelPlan:
        """The level plan of a batch from its CSR offsets held on the HOST (a loader has them before the transfer;
        ``grid_sampling(..., return_host_ptr=True)`` returns them): ``forward(..., plan=...)`` and ``prefetch_geometry`` then
        read nothing back from the device.  Cached by tile sizes like ``plan_for``."""
        key = tuple(int(v) for v in ptr_host)
        plan = self._plans.get(key)
        if plan is None:
            if len(self._plans) > 64:
                self._plans.clear()
            plan = make_plan(key, self.decimation, self.num_neighbors, next(self.parameters()).device)
            self._plans[key] = plan
        return plan

    def invalidate_plan_cache(self) -> None:
        """Forget every remembered ``ptr`` tensor / level plan.  ``plan_for`` recognises a ``ptr`` tensor it has seen by
        identity + autograd version counter; a write that bypasses the counter (a raw kernel, a hipGraph replay into a static
        ``ptr`` buffer) is invisible to it — call this after such a write, or hand ``forward`` an explicit ``plan``."""
        self._plan_ident = []
        self._plans.clear()

    # ------------------------------------------------------------------------------------------
    def _shared_layer(self, mlp: SharedMLPParams, li: int, x0: Tensor, x1: Optional[Tensor] = None,
                      rows: Optional[Tensor] = None, train: bool = False, x0_slot=None, x1_slot=None, drop=None,
                      rows_inv=None, defer: bool = False, y_slot=None) -> Tensor:
        """``defer`` (train mode): the layer's ONLY consumer is the next SharedMLP layer on the same rows — its BatchNorm +
        LeakyReLU are applied by that layer's GEMM as it loads its input (``ops.PendingBN``, round 5), no launch here."""
        lin, bn = mlp.lins[li], mlp.norms[li].module
        if train:
            sk = self._sinks(lin.weight, lin.bias, bn.weight, bn.bias) if self._use_sinks else None
            return ops.SharedLayerTrainFn.apply(x0, x1, lin.weight, lin.bias, bn.weight, bn.bias, bn, mlp.act, rows,
                                                sk, self._bf16, x0_slot, x1_slot, drop, rows_inv, defer, y_slot)
        if self._grad_eval:
            return ops.SharedLayerEvalFn.apply(x0, x1, lin.weight, lin.bias, bn.weight, bn.bias, bn, mlp.act, rows)
        scale, shift = self._cached(("bn", id(bn)), lambda: ops.bn_fold_eval(bn), self._bn_deps(bn))
        M = x1.shape[0] if x1 is not None else (rows.numel() if rows is not None else x0.shape[0])
        return ops.gemm(x0, lin.weight, M, lin.weight.shape[0], x0.shape[1], rows=rows, a1=x1,
                        k1=0 if x1 is None else x1.shape[1], bias=lin.bias, scale=scale, shift=shift, act=mlp.act,
                        bf16=self._bf16)

    def _lfa_mode(self, ch: int, K: int, full: bool) -> int:
        """Matrix-core mode of an LFA layer's attention GEMMs: 0 = f32-input MFMA, 1 = bf16 operands, 2 = split-bf16."""
        if not ops.lfa_bf16_ok(ch, K):
            return 0
        if self._bf16:
            return 1
        return 2 if (ch >= 64 and getattr(self, "_bf16x3", False) and full and ops.USE_LFA_FULL and K in (16, 32)) else 0

    def _lfa(self, p: LFAParams, x: Tensor, pos4: Tensor, idx: Tensor, mom: Optional[Tensor], num_edges: int,
             train: bool, prepared=None, defer_post: bool = False, rev=None, x_slot=None, post_slot=None) -> Tensor:
        """``x_slot`` / ``post_slot`` (bf16 activation storage, train): GradSlots through which an fp32 input gradient of this
        LFA layer reaches the layer that produced ``x``, resp. through which the NEXT LFA layer's reaches this one's
        ``mlp_post_attention`` (``ops.LFATrainFn._dx_out``)."""
        enc_lin, enc_bn = p.mlp_encoder.lins[0], p.mlp_encoder.norms[0].module
        w_att = p.mlp_attention.lins[0].weight
        bf16 = self._lfa_mode(w_att.shape[0], idx.shape[1], num_edges == idx.shape[0] * idx.shape[1])
        if train:
            sk = self._sinks(enc_lin.weight, enc_lin.bias, enc_bn.weight, enc_bn.bias, w_att) if self._use_sinks \
                else None
            agg = ops.LFATrainFn.apply(x, pos4, idx, mom, num_edges, enc_lin.weight, enc_lin.bias, enc_bn.weight,
                                       enc_bn.bias, enc_lin, enc_bn, w_att, sk, bf16, prepared, rev, x_slot)
        elif self._grad_eval:
            agg = ops.LFAEvalFn.apply(x, pos4, idx, enc_lin.weight, enc_lin.bias, enc_bn.weight, enc_bn.bias, enc_lin, enc_bn,
                                      w_att)
        else:
            if idx.shape[1] <= 32:
                wf, bf, wp = self._cached(("lfa", id(p), bf16),
                                          lambda: (lambda r: (r[0], r[1], r[4]))(
                                              ops.lfa_prepare(enc_lin, enc_bn, None, 0, w_att, bf16, False)),
                                          (enc_lin.weight, enc_lin.bias, w_att) + self._bn_deps(enc_bn))
            else:
                wf, bf, wp = self._cached(("lfa", id(p), False), lambda: ops.lfa_enc_fold(enc_lin, enc_bn, None, 0)[:2]
                                          + (None,), (enc_lin.weight, enc_lin.bias, w_att) + self._bn_deps(enc_bn))
            agg = ops.lfa_forward(x, pos4, idx, wf, bf, w_att, wp, bf16=bf16,
                                  full=bool(num_edges == idx.shape[0] * idx.shape[1]))
        return self._shared_layer(p.mlp_post_attention, 0, agg, train=train, defer=defer_post, y_slot=post_slot)

    def _block(self, blk: BlockParams, x: Tensor, pos4: Tensor, index: ops.KnnIndex, idx: Tensor,
               mom: Optional[Tensor], num_edges: int, train: bool, rec: Optional[dict], name: str,
               wait_graph=None, x_slot=None, prepared=(None, None), rev=None) -> Tensor:
        # idx: knn_graph(loop=True), pyg_randla_net.py:180 — rows and neighbour ids are cell-sorted slots of this level
        # x_slot (train): the block input has several consumers (mlp1, the shortcut, and on the decimated levels the FP
        # module's skip): their input gradients meet in one buffer, mlp1 — last in backward order — returns the sum
        # (bf16 storage: the fp32 input gradients of the two LFA layers travel through side slots, see _lfa)
        s1 = ops.GradSlot() if (train and ops._h(x) and torch.is_grad_enabled()) else None
        s2 = ops.GradSlot() if s1 is not None else None
        h = self._shared_layer(blk.mlp1, 0, x, train=train, x0_slot=x_slot, y_slot=s1)  # does not need the graph: runs while kNN finishes
        if wait_graph is not None:
            wait_graph()
        if rec is not None:
            rec[name + ".knn_idx"] = _knn_to_reference_order(idx, index)
            rec[name + ".mlp1"] = h[index.inv.long()]
        h = self._lfa(blk.lfa1, h, pos4, idx, mom, num_edges, train, prepared[0], rev=rev, x_slot=s1, post_slot=s2)
        if rec is not None:
            rec[name + ".lfa1"] = h[index.inv.long()]
        # lfa2's SharedMLP feeds mlp2 only: on levels 1-2 (<= 64 channels: the row-stream GEMM) mlp2's GEMM applies its
        # BatchNorm on load instead of a launch of its own
        defer2 = bool(train and rec is None and ops.BN_ON_LOAD and blk.mlp2.lins[0].weight.shape[1] <= 64
                      and blk.mlp2.lins[0].weight.shape[1] % 4 == 0 and torch.is_grad_enabled())
        h = self._lfa(blk.lfa2, h, pos4, idx, mom, num_edges, train, prepared[1], defer_post=defer2, rev=rev, x_slot=s2)
        l2, n2 = blk.mlp2.lins[0], blk.mlp2.norms[0].module
        ls, ns = blk.shortcut.lins[0], blk.shortcut.norms[0].module
        if train:
            sk2 = self._sinks(l2.weight, l2.bias, n2.weight, n2.bias) if self._use_sinks else None
            sks = self._sinks(ls.weight, ls.bias, ns.weight, ns.bias) if self._use_sinks else None
            out = ops.ResidualTailTrainFn.apply(h, l2.weight, l2.bias, n2.weight, n2.bias, n2, x, ls.weight, ls.bias,
                                                ns.weight, ns.bias, ns, sk2, sks, self._bf16, x_slot)
        elif self._grad_eval:
            out = ops.ResidualTailEvalFn.apply(h, l2.weight, l2.bias, n2.weight, n2.bias, n2, x, ls.weight, ls.bias,
                                               ns.weight, ns.bias, ns)
        else:
            sc2, sh2 = self._cached(("bn", id(n2)), lambda: ops.bn_fold_eval(n2), self._bn_deps(n2))
            scs, shs = self._cached(("bn", id(ns)), lambda: ops.bn_fold_eval(ns), self._bn_deps(ns))
            if ops.EVAL_RESIDUAL_EPILOGUE:
                # LeakyReLU(BN(mlp2(h)) + BN(shortcut(x))) as two launches (round 6): the shortcut's GEMM with its folded
                # BatchNorm as the epilogue, then mlp2's GEMM whose epilogue adds that buffer before the activation
                out = ops.gemm(x, ls.weight, x.shape[0], ls.weight.shape[0], x.shape[1], bias=ls.bias, scale=scs, shift=shs,
                               bf16=self._bf16)
                ops.gemm(h, l2.weight, h.shape[0], l2.weight.shape[0], h.shape[1], bias=l2.bias, scale=sc2, shift=sh2, act=True,
                         out=out, accumulate=True, bf16=self._bf16)
            else:
                z2 = ops.gemm(h, l2.weight, h.shape[0], l2.weight.shape[0], h.shape[1], bias=l2.bias, bf16=self._bf16)
                zs = ops.gemm(x, ls.weight, x.shape[0], ls.weight.shape[0], x.shape[1], bias=ls.bias, bf16=self._bf16)
                out = ops.bn_apply(z2, sc2, sh2, True, zs, scs, shs)
        if rec is not None:
            rec[name + ".out"] = out[index.inv.long()]
        return out

    # ------------------------------------------------------------------------------------------
    def forward(self, x: Optional[Tensor], pos: Tensor, batch: Optional[Tensor], ptr: Tensor,
                decimation_idx: Optional[List[Tensor]] = None, dropout_mask: Optional[Tensor] = None,
                plan: Optional[LevelPlan] = None, record: Optional[dict] = None) -> Tensor:
        """``forward(x, pos, batch, ptr) -> [sum N, num_classes]`` (pyg_randla_net.py:55-88).

        ``batch`` is accepted for signature parity and unused (``ptr`` carries the same information).
        Testing/benchmark extras: injected ``decimation_idx`` (one index tensor per level), injected dropout
        keep-``dropout_mask``, a pre-computed ``plan`` (skips the ptr read-back; needed under hipGraph capture),
        ``record`` of intermediates."""
        if self.decimation < 1:
            raise ValueError(
                "Argument `decimation_factor` should be higher than (or equal to) 1 for downsampling. "
                f"(Current value: {self.decimation})"
            )
        if not pos.is_cuda:
            raise RuntimeError("HipRandLANet runs on an MI355X (cuda/HIP device) only; there is no CPU fallback")
        x = x if x is not None else pos
        x = x.to(torch.float32).contiguous()
        pos = pos.to(torch.float32).contiguous()
        train = self.training
        # eval mode with autograd enabled and something to differentiate: the reference's eval forward records a graph like
        # any torch module (BatchNorm on running statistics, no dropout) — so does this one (ops.*EvalFn)
        self._grad_eval = (not train) and torch.is_grad_enabled() and \
            (x.requires_grad or any(p.requires_grad for p in self.parameters()))
        ctx = torch.enable_grad() if ((train or self._grad_eval) and torch.is_grad_enabled()) else torch.no_grad()
        with ctx:
            return self._forward(x, pos, ptr.to(torch.int64).contiguous(), decimation_idx, dropout_mask, plan,
                                 record, train)

    def _geometry(self, pos: Tensor, plan: LevelPlan, decimation_idx, train: bool, wait_main: bool = True) -> "_Geometry":
        """Everything that depends on positions only — kNN grids and tables of the 4 levels, encoder moments, random
        decimation, the decoder's 1-NN tables — enqueued on a SIDE stream: these kernels are latency / VALU-bound
        and overlap with the MFMA- and HBM-bound feature kernels of the main stream (under hipGraph capture the two
        streams become parallel branches of the graph).  ``wait(stage)`` makes the main stream wait for a stage."""
        main = torch.cuda.current_stream()
        side = self._side_stream(pos.device) if self.overlap_geometry else main
        g = _Geometry(main, side)
        if side is not main and wait_main:
            side.wait_stream(main)
        # (in place, inside the forward: whether a backward pass can follow is known)
        for _ in self._geometry_stages(g, pos, plan, decimation_idx, train, differentiable=torch.is_grad_enabled()):
            pass
        return g

    def _reverse_neighbours(self, plan: LevelPlan, lvl: int, idx: Tensor, train: bool, differentiable: bool = True):
        """(ptr, inv, slot): reverse neighbour lists of a level's K-NN table — point j's list holds the edges (i, k) with idx[i][k] == j — for the
        levels whose LFA backward kernels store their input gradient per EDGE instead of adding it with float atomics
        (``m3d_lfa_bwd`` flags bit 5: the 8 / 16-channel layers of block 1 with complete neighbourhoods; round 5: the atomics
        were half of those launches).  Position-only, enqueued behind the table on the geometry stream (off the step's
        critical path; the backward pass is ordered behind it through the stages the forward pass waits for).  None: the
        level's layers scatter with atomics."""
        K = self.num_neighbors
        n = plan.totals[lvl]
        # (only a pass that will be differentiated needs them: a train-mode forward under no_grad — a BatchNorm recalibration, a
        # metrics pass — skips the six launches)
        # (a prefetch cannot know: it builds them for every train-mode batch)
        if not (train and differentiable and ops.USE_LFA_FULL and ops.USE_LFA_EDGE_ROWS and plan.num_edges[lvl] == n * K):
            return None
        blk = (self.block1, self.block2, self.block3, self.block4)[lvl]
        chs = [lfa.mlp_attention.lins[0].weight.shape[0] for lfa in (blk.lfa1, blk.lfa2)]
        if not any(ops.lib().m3d_lfa_bwd_edge_rows_ok(n, K, ch, ops.LRELU_SLOPE) for ch in chs):
            return None
        return ops.knn_reverse(idx, with_inv=not ops.USE_LFA_EDGE_SLOTS)  # (rows in list order: the slot table is all it takes)

    def _geometry_stages(self, g: "_Geometry", pos: Tensor, plan: LevelPlan, decimation_idx, train: bool,
                         differentiable: bool = True, background: bool = False):
        """The position-only work as a generator: each ``next()`` enqueues one stage on ``g.side`` (10 stages: grid of
        level 1; then per level its kNN table + encoder moments, and its decimation + the next level's grid; last the
        decoder's four 1-NN tables).  ``_geometry`` runs them back to back; an interleaved prefetch lets the forward pass
        enqueue one stage between its own blocks (see ``prefetch_geometry(interleave=True)``)."""
        K = self.num_neighbors
        side = g.side
        bg = max(0, min(255, int(self.background_knn_cap if train else self.background_knn_cap_eval))) if background else 0
        # batched launches pay off where launches are the cost: eagerly (7.5 -> 6.8 ms per training step).  Inside a
        # captured graph the per-level launches are free for the host and run one after the other without competing with
        # the feature kernels, which measured 0.04 ms better (profiles/r02x_geo_batch.log) — same tables either way
        # (eval: the forward is SHORTER than the position-only chain and waits for it — there the four K-NN queries as one launch
        # take the time of the slowest instead of the sum, captured or not: M3D_GEO_BATCH_EVAL, round 6)
        # (training: batched under capture measured 3.96-3.98 vs 3.905 ms per step even with the background cap, r06s)
        batched = self.batch_geometry and (not torch.cuda.is_current_stream_capturing() or
                                           (not train and self.batch_geometry_eval_capture))
        with torch.cuda.stream(side):
            g.index.append(ops.KnnIndex(ops.pad_pos(pos), plan.ptrs[0]))
            g.pos4.append(g.index[0].sorted_pos4)
            g.mark(0)
        yield
        for lvl in range(4):
            if not batched:
                with torch.cuda.stream(side):
                    ix = g.index[lvl]
                    idx, _ = ix.query(K, qry=ix, sorted_io=True, background=bg)
                    g.knn.append(idx)
                    g.mom.append(ops.lfa_moments(g.pos4[lvl], idx) if train else None)
                    g.mark(1 + 2 * lvl)
                    g.knn_inv.append(self._reverse_neighbours(plan, lvl, idx, train, differentiable))
                yield
            with torch.cuda.stream(side):
                ix = g.index[lvl]
                # decimate(): pyg_randla_net.py:234-238.  d_int: sorted slots of this level that survive, listed in
                # the reference order of the next level; d_ref: the same as reference rows of this level
                d_ref = None
                if decimation_idx is not None:
                    d_ref = decimation_idx[lvl].to(device=pos.device, dtype=torch.int32).contiguous()
                    assert d_ref.numel() == plan.totals[lvl + 1]
                # drawn in REFERENCE rows (like the reference's randperm, pyg_randla_net.py:221): which points survive
                # depends on the seed only, not on the (arbitrary) order of points inside a grid cell.  One launch draws,
                # maps to sorted slots and fetches the survivors' positions; the next level's grid build carries the slot
                # map into its own order (src: sorted slot of level lvl+1 -> sorted slot of level lvl)
                d_ref, d_int, pos_next = ops.decimate_level(plan.ptrs[lvl], plan.ptrs[lvl + 1], plan.totals[lvl + 1],
                                                            self._decim_seed, lvl, ix, d_ref)
                g.dec_ref.append(d_ref)
                nxt = ops.KnnIndex(pos_next, plan.ptrs[lvl + 1], carry=d_int)
                g.src.append(nxt.carried)
                g.index.append(nxt)
                g.pos4.append(nxt.sorted_pos4)
                g.mark(2 + 2 * lvl)
            yield
        if batched:
            # the grids of all five levels exist (a chain of short kernels: nothing above waits for a table): the four
            # K-NN tables are ONE launch, the four moment sets one, the four decoder 1-NN tables one.  Launched level by
            # level the deep ones are latency-bound (105 / 77 / 45 us for 51 200 / 12 800 / 3 200 queries) and the
            # level-1 launch ends on its slowest wavefronts with most SIMDs idle (profiles/r02u_step_timeline.csv)
            with torch.cuda.stream(side):
                g.knn.extend(ops.knn_query_batch([(g.index[l], g.index[l]) for l in range(4)], K, background=bg))
                g.mom.extend(ops.lfa_moments_batch(g.pos4[:4], g.knn) if train else [None] * 4)
                for lvl in range(4):
                    g.mark(1 + 2 * lvl, new=(lvl == 0))
                g.knn_inv.extend(self._reverse_neighbours(plan, lvl, g.knn[lvl], train, differentiable) for lvl in range(4))
            yield
        with torch.cuda.stream(side):
            if batched:
                g.nn.extend(ops.knn_query_batch([(g.index[l + 1], g.index[l]) for l in range(4)], 1, background=bg))
            else:
                for lvl in range(4):  # FPModule(k=1): pyg_randla_net.py:250
                    g.nn.append(g.index[lvl + 1].query(1, qry=g.index[lvl], sorted_io=True, background=bg)[0])
            g.mark(9)
            # CSR inverses of the four 1-NN tables (train): the backward pass of the decoder's x[nn] gathers sums rows per
            # coarse point instead of scattering them with atomics (ops.gather_sum_rows).  A stage of its own: the forward
            # pass needs the tables (stage 9) a millisecond before the backward pass needs the inverses
            g.nn_inv.extend(ops.csr_invert_batch([t.view(-1) for t in g.nn], [plan.totals[l + 1] for l in range(4)])
                            if train else [None] * 4)
            g.mark(10)
        yield

    # ------------------------------------------------------------------------------------------
    # geometry lookahead.  The position-only work of a forward pass (kNN grids and tables, encoder moments, random
    # decimation, decoder 1-NN tables: ~0.9 ms of kernels at BASELINE config 2, ~0.25 ms of them in front of the
    # first feature kernel that needs a table) depends on nothing the previous step produces — in training the next
    # batch's positions are in the dataloader's prefetch queue — so it can be enqueued one step ahead:
    #     net.prefetch_geometry(pos_next, ptr_next)            # side stream, returns at once
    #     out = net(x, pos, None, ptr)                         # consumes the tables prefetched for THIS batch earlier
    #     loss(out).backward(); optimizer.step()               # ... all of it runs beside the side stream's kernels
    # Results live in two persistent buffer sets ("slots") used in turn: the slot written during step i is read by step
    # i + 1 (forward AND backward) and rewritten during step i + 2.  hipGraph: capture two steps (one per slot) and
    # replay them in turn; every captured step ends with join_geometry().
    # ------------------------------------------------------------------------------------------
    def prefetch_geometry(self, pos: Tensor, ptr: Tensor, plan: Optional[LevelPlan] = None,
                          train: Optional[bool] = None, after: str = "now", interleave: bool = False,
                          slot: Optional[int] = None, owner=None, after_event=None) -> None:
        """Enqueue the position-only work for a batch the network will see later (at most two may be outstanding:
        the one the next ``forward`` consumes and the one after it).  The tables are matched to the forward that
        consumes them by the IDENTITY of ``pos`` (the same tensor object, unmodified since: its autograd version
        counter is part of the match) — a different batch that happens to land at the same address, or a static
        buffer refilled in place after the prefetch, gets fresh tables instead of stale ones.  ``slot`` (0 / 1): which
        of the two persistent buffer sets to write (default: the one not written last).

        ``after="now"``: ordered behind everything enqueued on the current stream so far (always safe).
        ``after="forward_start"``: ordered behind the START of the most recent forward only — for callers whose ``pos``
        was resident before that forward began (a dataloader prefetch queue).  Honoured only when a forward has
        started SINCE the one that consumed the buffer set rewritten here (that forward's start is then behind the
        consumer's backward pass in stream order — one training step in flight at a time); otherwise the call falls
        back to ``"now"``: the consumer's forward and backward kernels still read the buffers.
        ``after_event``: an event the position-only work must also wait for (``pos`` written on a stream of the caller's).
        ``interleave=True``: only the first stage is enqueued now; the NEXT ``forward`` call (which consumes an older
        prefetch, or works in place) enqueues the remaining stages one by one between its own blocks.  A captured
        hipGraph submits its nodes in capture order at several microseconds apiece: a hundred position-only nodes in
        front of the feature kernels keep the chip nearly idle for the first millisecond of every replay, behind
        them they start when the forward pass is over and fight the backward pass for the machine; interleaved, both
        chains progress from the first microsecond."""
        if not pos.is_cuda:
            raise RuntimeError("HipRandLANet runs on an MI355X (cuda/HIP device) only; there is no CPU fallback")
        self._finish_interleaved()  # (a request nobody advanced: complete it before its slot partner is rewritten)
        pos = pos.to(torch.float32).contiguous()
        ptr = ptr.to(torch.int64).contiguous()
        if plan is None:
            plan = self.plan_for(ptr)
        else:
            _check_plan(plan, pos, ptr)
        plan_ready(plan)
        train = self.training if train is None else train
        key = (tuple(pos.shape), id(plan), bool(train))
        self._look_turn = (self._look_turn ^ 1) if slot is None else int(slot) & 1
        turn = (owner, bool(train), self._look_turn)
        side = self._side_stream(pos.device)
        main = torch.cuda.current_stream()
        # what comes first: the producer of ``pos`` and the step that last read the slot rewritten here
        old = self._look_slots.get(turn)
        if (after == "forward_start" and self._fwd_start is not None
                and (old is None or old.consumer_fwd < self._fwd_count)
                and ops.capture_id(main) == 0):
            side.wait_event(self._fwd_start)
            if plan.ready is not None:
                side.wait_event(plan.ready)  # (this branch does not wait for the main stream, which carries the plan's upload)
        else:
            side.wait_stream(main)
        if after_event is not None:
            side.wait_event(after_event)
        with torch.cuda.stream(side):
            self._seed_decimation()
            self._decim_seed += 0x9E3779B97F4A7C15 - (1 << 64)  # (side stream: ordered with the kernels that read it)
        geo = _Geometry(main, side)
        stages = self._geometry_stages(geo, pos, plan, None, train, background=True)
        self._look_job = (stages, geo, turn, key, pos, main)
        if interleave:
            next(stages)
        else:
            self._finish_interleaved()

    def _advance_interleaved(self) -> None:
        """One more stage of the interleaved prefetch (called by the forward pass between its blocks)."""
        if self._look_job is not None:
            if self.interleave_paced:
                # pace the side stream by the forward pass: stage k starts no earlier than the feature block in front of
                # it.  A dependency the data does not need — it makes a hipGraph executor, which submits a branch until
                # it meets a node whose predecessor is not submitted yet, alternate between the two chains
                self._look_job[1].side.wait_stream(torch.cuda.current_stream())
            try:
                next(self._look_job[0])
            except StopIteration:
                self._finish_interleaved()

    def _finish_interleaved(self) -> None:
        job = self._look_job
        if job is None:
            return
        self._look_job = None
        stages, geo, turn, key, pos, main = job
        for _ in stages:
            pass
        slot = self._look_slots.get(turn)
        with torch.cuda.stream(geo.side):
            fresh = geo.tensors()
            if slot is None or slot.key != key:
                # (a new layout: eagerly the fresh tables BECOME the buffer set — a predict chain changes layout with every
                # batch; under capture they live in the graph's pool and are copied out)
                adopt = ops.capture_id(geo.side) == 0
                slot = self._look_slots[turn] = _GeoSlot(key, list(fresh) if adopt else [t.clone() for t in fresh], geo, main)
            else:
                ops.copy_many(slot.bufs, fresh)  # one launch (a replayed graph pays ~9 us per memcpy node)
            slot.ready = torch.cuda.Event()
            slot.ready.record(geo.side)
            slot.ready_capture = ops.capture_id(geo.side)
        slot.pos, slot.version = pos, pos._version  # (the reference also keeps the address from being recycled)
        if turn in self._look_queue:
            self._look_queue.remove(turn)  # a prefetch nobody consumed: its slot has just been rewritten
        self._look_queue.append(turn)

    def join_geometry(self) -> None:
        """Make the current stream wait for every outstanding ``prefetch_geometry`` (end of a captured step: all
        streams of a hipGraph capture must be joined before the capture ends)."""
        self._finish_interleaved()
        main = torch.cuda.current_stream()
        cap = ops.capture_id(main)
        for turn in self._look_queue:
            slot = self._look_slots[turn]
            if slot.ready_capture == cap:
                main.wait_event(slot.ready)
            elif cap == 0:  # recorded inside a finished capture: not a waitable event in eager mode
                main.wait_stream(self._side_stream(main.device))

    def _consume_lookahead(self, pos: Tensor, plan: LevelPlan, train: bool) -> Optional["_Geometry"]:
        want = (tuple(pos.shape), id(plan), bool(train))
        while self._look_queue:
            turn = self._look_queue.pop(0)  # oldest first
            slot = self._look_slots[turn]
            if slot.key != want or slot.pos is not pos or slot.version != pos._version:
                continue  # prefetched for another batch / mode, or ``pos`` was written since: dropped (the in-place
                # path computes fresh tables if nothing fits)
            slot.consumer_fwd = self._fwd_count
            main = torch.cuda.current_stream()
            cap = ops.capture_id(main)
            if slot.ready_capture == cap:
                main.wait_event(slot.ready)  # same capture (becomes a graph edge) or plain eager execution
            elif cap == 0:
                main.wait_stream(self._side_stream(pos.device))  # written by a replayed graph: order behind the side stream
            # else: capturing, and the tables were written before this graph starts (eager warm-up, or the previously
            # replayed graph, whose side branch joined at its end): graphs replay in stream order, nothing to wait for
            return slot.geo
        return None

    def _side_stream(self, device) -> "torch.cuda.Stream":
        st = self._streams.get(device)
        if st is None:
            st = self._streams[device] = torch.cuda.Stream(device=device)
        return st

    def _forward(self, x, pos, ptr, decimation_idx, dropout_mask, plan, record, train):
        """Internally every level lives in the CELL-SORTED order of its kNN grid (spatially coherent: a centre's
        neighbours sit a few cache lines away instead of anywhere in the tile).  ``perm[l]`` maps a level's sorted
        slot to its reference row (level 0: the caller's row; level l+1: position in the decimation index list),
        ``inv[l]`` is the inverse.  Inputs are permuted once, logits are un-permuted once; decimation composes with
        the next level's permutation into a single row gather."""
        if plan is None:
            plan = self.plan_for(ptr)
        else:
            _check_plan(plan, pos, ptr)
        plan_ready(plan)
        ops.drop_pending()  # (a forward that raised may have left unapplied BatchNorms behind: never materialised later)
        self._use_sinks = bool(train and torch.is_grad_enabled() and self._check_flat())
        if self.matmul_precision not in ("fp32", "bf16", "bf16x3"):
            raise ValueError(f"matmul_precision must be 'fp32', 'bf16' or 'bf16x3', got {self.matmul_precision!r}")
        # "bf16x3": the attention GEMMs of the LFA layers with >= 64 channels as split-bf16 products (hi + lo operands, three
        # matrix-core products, fp32 accumulate: ~fp32 accuracy off the vector pipe); everything else stays fp32
        self._bf16x3 = self.matmul_precision == "bf16x3"
        self._bf16 = self.matmul_precision == "bf16" or (
            torch.is_autocast_enabled() and torch.get_autocast_gpu_dtype() == torch.bfloat16)
        bump = self._nbt_flat if (train and self._flat is not None) else None  # BatchNorm step counters (one int64 vector)
        if train and torch.is_grad_enabled():
            # one launch: zero fill of every accumulation target of the coming backward pass + the counters' "+ 1"
            ops.arena.begin(pos.device, bump=bump)
        else:
            ops.arena.stop()
            if bump is not None:
                bump += 1
        if train:
            self._eval_cache.clear()  # this pass updates the running statistics (and an optimizer step follows)
        # weight gradients on a side stream: only with gradient sinks and an optimizer that joins the stream
        ops._grad_side = self.grad_side if self._use_sinks else None
        blocks = (self.block1, self.block2, self.block3, self.block4)
        self._fwd_count += 1
        if self._look_slots or self._look_queue:  # lookahead in use: mark where this forward starts
            self._fwd_start = torch.cuda.Event()
            self._fwd_start.record(torch.cuda.current_stream())
        if train and self._use_sinks and self.grad_side is not None:
            self.grad_side.begin_step()
        geo = self._consume_lookahead(pos, plan, train) if decimation_idx is None else None
        if geo is None:
            if decimation_idx is None:
                self._seed_decimation()
                self._decim_seed += 0x9E3779B97F4A7C15 - (1 << 64)  # device-side bump, hipGraph-replay safe
            geo = self._geometry(pos, plan, decimation_idx, train)
        index, pos4, dec_ref = geo.index, geo.pos4, geo.dec_ref
        feats: List[Tensor] = []
        hin: List[Optional[Tensor]] = [None]  # decimated input of block l (= skip tensor of the FP module above it)
        geo.wait(0)
        diff = train or self._grad_eval  # the pass records an autograd graph
        # bf16 activation storage (not for the differentiable eval pass, whose BatchNorm backward is a few torch ops)
        act16 = self.activation_dtype == torch.bfloat16 and not self._grad_eval
        if self.activation_dtype not in (torch.float32, torch.bfloat16):
            raise ValueError(f"activation_dtype must be torch.float32 or torch.bfloat16, got {self.activation_dtype!r}")
        if act16 and self.num_neighbors > 32:
            raise ValueError("bf16 activation storage needs the fused LFA kernels: num_neighbors <= 32")
        # fc0 on the cell-sorted order of level 1: the input permutation is a row gather inside the GEMM's A operand (and inside
        # the weight gradient's); an input that needs a gradient itself takes the differentiable gather in front
        if x.requires_grad:
            x = ops.GatherRowsFn.apply(x, index[0].perm, index[0].inv)
            in_rows = None
            if act16:
                x = x.to(torch.bfloat16)  # (differentiable; a rare path: saliency of the input features)
        else:
            in_rows = index[0].perm
            if act16:
                x = ops.to_bf16(x)  # the input features once, [sum N, F]: every later kernel reads 2-byte elements
        h = ops.LinearFn.apply(x, self.fc0.weight, self.fc0.bias,
                               self._sinks(self.fc0.weight, self.fc0.bias) if self._use_sinks else None, in_rows) if diff else \
            ops.gemm(x, self.fc0.weight, pos.shape[0], self.fc0.weight.shape[0], x.shape[1], rows=in_rows, bias=self.fc0.bias)
        # gradient meeting points (train): in_slots[l] = input of block l (hin[l]), out_slot = output of block 1
        use_slots = train and torch.is_grad_enabled() and self.share_input_gradients
        in_slots = [ops.GradSlot() if use_slots else None for _ in range(4)]
        out_slot = ops.GradSlot() if use_slots else None
        # with prefetched tables the encoder moments of every level exist already: fold the eight encoder BatchNorms and pack
        # the eight attention weights in ONE launch at the head of the chain (in place of eight 5-us launches inside it)
        prepared = [(None, None)] * 4
        if train and geo.side is geo.main and self.num_neighbors <= 32 and self.batch_lfa_prepare:
            jobs = []
            for lvl, blk in enumerate(blocks):
                for lfa in (blk.lfa1, blk.lfa2):
                    w_att = lfa.mlp_attention.lins[0].weight
                    jobs.append((lfa.mlp_encoder.lins[0], lfa.mlp_encoder.norms[0].module, geo.mom[lvl], plan.num_edges[lvl],
                                 w_att, self._lfa_mode(w_att.shape[0], self.num_neighbors,
                                                       plan.num_edges[lvl] == plan.totals[lvl] * self.num_neighbors)))
            outs = ops.lfa_prepare_batch(jobs)
            prepared = [(outs[2 * l], outs[2 * l + 1]) for l in range(4)]
        for lvl, blk in enumerate(blocks):
            h = self._block(blk, h, pos4[lvl], index[lvl], geo.knn[lvl], geo.mom[lvl], plan.num_edges[lvl], train,
                            record, f"block{lvl + 1}",
                            wait_graph=lambda s=1 + 2 * lvl: geo.wait(s),  # kNN table (+ encoder moments) of this level
                            x_slot=in_slots[lvl], prepared=prepared[lvl],
                            rev=geo.knn_inv[lvl] if (train and lvl < len(geo.knn_inv)) else None)
            feats.append(h)
            self._advance_interleaved()
            self._advance_interleaved()  # (two position-only stages per level: table + moments, decimation + next grid)
            geo.wait(2 + 2 * lvl)  # decimation map into the next level
            # (drawn indices are the head of a permutation of each cloud: distinct whenever a cloud keeps no more points
            # than it has — the backward scatter then needs no atomics; injected indices may repeat rows)
            distinct = decimation_idx is None and all(b <= a for a, b in zip(plan.sizes[lvl], plan.sizes[lvl + 1]))
            h = ops.GatherRowsFn.apply(h, geo.src[lvl], None, out_slot if lvl == 0 else None, distinct) if diff \
                else ops.gather_rows(h, geo.src[lvl])
            hin.append(h)
        self.last_decimation_idx = dec_ref
        self._advance_interleaved()  # (the last stage; the slot copy follows it)
        self._advance_interleaved()
        geo.wait(9)  # decoder 1-NN tables
        h = self._shared_layer(self.mlp_summit, 0, h, train=train)
        if record is not None:
            record["summit"] = h[index[4].inv.long()]
        # decoder: FPModule(k=1) x4 (pyg_randla_net.py:76-79, 241-253)
        for fp, lvl in ((self.fp4, 3), (self.fp3, 2), (self.fp2, 1), (self.fp1, 0)):
            nn_idx = geo.nn[lvl]  # 1-NN of every level-`lvl` point among level lvl+1 (both in sorted slots)
            skip = feats[0] if lvl == 0 else hin[lvl]  # b1_out, resp. the decimated output of block lvl
            # knn_interpolate(k=1) == x[nn] (weights cancel); fused as a row gather into the GEMM's A operand
            # (the skip tensor's other consumers run later in the backward pass: this layer deposits its gradient)
            # (fp1 -> mlp_classif[0] -> mlp_classif[1]: a chain of SharedMLP layers on the level-1 rows — each next GEMM applies
            # the BatchNorm of the layer in front on load)
            chain = bool(train and record is None and ops.BN_ON_LOAD and torch.is_grad_enabled())
            h = self._shared_layer(fp.nn, 0, h, x1=skip, rows=nn_idx.view(-1), train=train,
                                   x1_slot=out_slot if lvl == 0 else in_slots[lvl],
                                   rows_inv=geo.nn_inv[lvl] if (train and geo.nn_inv) else None,
                                   defer=chain and lvl == 0)
            if record is not None:
                record[f"fp{lvl + 1}"] = h[index[lvl].inv.long()]
        h = self._shared_layer(self.mlp_classif, 0, h, train=train,
                               defer=bool(train and record is None and ops.BN_ON_LOAD and torch.is_grad_enabled()))
        if train and 10 in geo.events:
            geo.wait(10)  # the CSR inverses the decoder's backward pass reads
        p = self.mlp_classif.dropout[1]
        # Dropout(p) behind the layer (pyg_randla_net.py:49-52).  Flattened nets: a counter-based mask on the device step counter
        # the step prologue advances (seeded like the decimation: torch's global seed and the rank), applied INSIDE the
        # layer's BatchNorm kernels, forward and backward (no launches of its own)
        fused_drop = (train and p > 0.0 and dropout_mask is None and self._flat is not None and torch.is_grad_enabled()
                      and ops._pow2(self.mlp_classif.lins[1].weight.shape[0]))
        h = self._shared_layer(self.mlp_classif, 1, h, train=train,
                               drop=(p, self._nbt_flat[-1:], self._dropout_seed(), index[0].perm) if fused_drop else None)
        if train and p > 0.0 and not fused_drop:
            if dropout_mask is not None:  # given in the caller's row order
                mask = ops.gather_rows(dropout_mask.to(h.dtype).contiguous(), index[0].perm)
                h = h * (mask / (1.0 - p))
            elif self._flat is not None and h.numel() % 4 == 0 and h.dtype == torch.float32:
                # counter-based mask on the device step counter the step prologue advances (seeded like the decimation:
                # torch's global seed and the rank)
                h = ops.DropoutFn.apply(h, p, self._nbt_flat[-1:], self._dropout_seed())
            else:
                h = F.dropout(h, p=p, training=True)
        if diff:
            logits = ops.LinearFn.apply(h, self.fc_classif.weight, self.fc_classif.bias,
                                        self._sinks(self.fc_classif.weight, self.fc_classif.bias)
                                        if self._use_sinks else None, None, torch.float32)  # (fp32 logits either way)
            logits = ops.GatherRowsFn.apply(logits, index[0].inv, index[0].perm)  # back to the caller's row order
        else:
            logits = ops.gemm(h, self.fc_classif.weight, h.shape[0], self.fc_classif.weight.shape[0], h.shape[1],
                              bias=self.fc_classif.bias, out_dtype=torch.float32)
            logits = ops.gather_rows(logits, index[0].inv)
        ops.settle_pending()  # (nothing is left pending on the paths above: a guard)
        if self.return_logits:
            return logits
        return logits.log_softmax(dim=-1)


class _Geometry:
    """Position-only intermediates of one forward pass (see ``HipRandLANet._geometry``)."""

    def __init__(self, main, side):
        self.main, self.side = main, side
        self.index: List[ops.KnnIndex] = []
        self.pos4: List[Tensor] = []
        self.knn: List[Tensor] = []
        self.mom: List[Optional[Tensor]] = []
        self.src: List[Tensor] = []
        self.dec_ref: List[Tensor] = []
        self.nn: List[Tensor] = []
        self.nn_inv: List[Optional[tuple]] = []  # train: CSR inverse (ptr, inv) of every 1-NN table
        self.knn_inv: List[Optional[tuple]] = []  # train: CSR inverse of a level's K-NN table where its LFA layers use it
        self.events: Dict[int, object] = {}
        self._last_event = None

    def mark(self, stage: int, new: bool = True) -> None:
        """Stage ``stage`` is complete at this point of the side stream (``new=False``: at the same point as the stage
        marked just before)."""
        if self.side is not self.main:
            if new or self._last_event is None:
                self._last_event = torch.cuda.Event()
                self._last_event.record(self.side)
            self.events[stage] = self._last_event

    def wait(self, stage: int) -> None:
        if self.side is not self.main:
            self.main.wait_event(self.events[stage])

    def tensors(self) -> List[Tensor]:
        """Every device buffer of this geometry, in a fixed order (``pos4`` / ``perm`` / ``inv`` are views of the
        index workspaces and come along with them)."""
        return [ix.ws for ix in self.index] + self.knn + [m for m in self.mom if m is not None] + self.src + \
            self.dec_ref + self.nn + [t for pair in self.nn_inv if pair is not None for t in pair] + \
            [t for pair in self.knn_inv if pair is not None for t in pair if t is not None]

    def rebound(self, buffers: List[Tensor], main) -> "_Geometry":
        """A geometry with the same structure whose buffers are ``buffers`` (same order as ``tensors()``); complete
        by construction, so ``wait`` is a no-op."""
        g = _Geometry(main, main)
        it = iter(buffers)
        for ix in self.index:
            cp = ops.KnnIndex.__new__(ops.KnnIndex)
            cp.n, cp.num_clouds, cp.ptr, cp.ws = ix.n, ix.num_clouds, ix.ptr, next(it)
            g.index.append(cp)
        g.pos4 = [ix.sorted_pos4 for ix in g.index]
        g.knn = [next(it) for _ in self.knn]
        g.mom = [next(it) if m is not None else None for m in self.mom]
        g.src = [next(it) for _ in self.src]
        g.dec_ref = [next(it) for _ in self.dec_ref]
        g.nn = [next(it) for _ in self.nn]
        g.nn_inv = [(next(it), next(it)) if pair is not None else None for pair in self.nn_inv]
        g.knn_inv = [tuple(next(it) if t is not None else None for t in pair) if pair is not None else None for pair in self.knn_inv]
        return g


class _GeoSlot:
    """One persistent buffer set of ``HipRandLANet.prefetch_geometry`` plus a ``_Geometry`` that points at it."""

    def __init__(self, key, bufs: List[Tensor], template: _Geometry, main):
        self.key = key
        self.bufs = bufs
        self.geo = template.rebound(bufs, main)
        self.ready = None   # event (side stream): the slot holds the prefetched geometry
        self.ready_capture = 0  # id of the hipGraph capture the event was recorded in (0: eager)
        self.pos: Optional[Tensor] = None  # the tensor the tables were computed from, and its version counter then
        self.version = -1
        self.consumer_fwd = 0  # index (HipRandLANet._fwd_count) of the forward that last consumed this buffer set


def _knn_to_reference_order(idx: Tensor, index: "ops.KnnIndex") -> Tensor:
    """Test/record helper: a sorted-slot kNN table as reference rows x reference neighbour ids."""
    perm = index.perm.long()
    ref = torch.where(idx >= 0, perm[idx.clamp(min=0).long()], torch.full_like(idx, -1).long())
    return ref[index.inv.long()].to(torch.int32)
