"""``HipPointNet2`` — the PointNet++ set-abstraction variant of BASELINE.json ``configs[4]`` ("Dense tiles 40 000 pts K=32 +
PointNet++ set-abstraction variant (stresses LDS/HBM kNN gather)") behind the same constructor / ``forward(x, pos, batch,
ptr)`` surface as ``PyGRandLANet`` (``/root/reference/myria3d/models/modules/pyg_randla_net.py:23-30,55``), so that it is
selectable through the same ``MODEL_ZOO`` / Hydra keys (``/root/reference/myria3d/models/model.py:12-29``).

THERE IS NO REFERENCE IMPLEMENTATION of this variant (``MODEL_ZOO = [PyGRandLANet]``; the repository has no farthest-point
sampling).  The structure is PointNet++ (Qi et al. 2017) as PyG packages it, written with the reference's own building
blocks:

* encoder: three set-abstraction levels.  Each keeps ``max(1, n // decimation)`` points per cloud (the count rule of
  ``decimation_indices``, pyg_randla_net.py:215-217) chosen by farthest-point sampling (``subsampling="fps"``,
  torch_cluster.fps semantics) or at random (``"random"``, the reference's scheme), groups the ``num_neighbors`` nearest
  points of the level around every kept point (kNN grouping instead of a radius ball: the config's "kNN gather"), runs the
  reference's ``SharedMLP`` (Linear -> BatchNorm(0.01, 1e-6) -> LeakyReLU(0.2), pyg_randla_net.py:92-109) over the edge
  rows ``[x_j, pos_j - pos_i]`` (PointNetConv.message) and takes the per-channel max over each group (``aggr="max"``);
* decoder: the reference's ``FPModule`` (1-NN upsample + skip concatenation + SharedMLP, pyg_randla_net.py:241-253);
* head: the reference's ``mlp_classif`` / ``fc_classif`` / log-softmax (pyg_randla_net.py:52-53,81-88).

Every op runs through the HIP kernels of ``csrc/`` (``sa.hip``: FPS, grouping, max; ``knn.hip``; ``gemm*.hip`` + ``bn.hip``);
``oracle/pointnet2_oracle.py`` restates the same net in plain torch and is the parity checker (parity unpinned).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F
from torch import Tensor, nn

from . import ops
from .randla import FPParams, SharedMLPParams

SA_WIDTHS = ((32, 32, 64), (64, 64, 128), (128, 128, 256))  # SharedMLP widths of the three set-abstraction levels


class SAParams(nn.Module):
    def __init__(self, mlp: SharedMLPParams):
        super().__init__()
        self.nn = mlp


@dataclass
class SAPlan:
    """Host-side description of a batch at every level (sizes are known from ``ptr``)."""

    sizes: List[List[int]]  # per level, per cloud
    ptrs: List[Tensor]  # device int64 [B+1] per level
    totals: List[int]
    segs: List[Tensor]  # device int64 [m_{l+1} + 1]: edge offsets of the centres of set-abstraction level l
    num_edges: List[int]
    max_points: List[int]


def make_sa_plan(ptr_host: Sequence[int], decimation: int, num_neighbors: int, device, levels: int = 3) -> SAPlan:
    if decimation < 1:
        raise ValueError("decimation factor must be >= 1")  # (the reference's own check, pyg_randla_net.py:208-212)
    sizes = [[int(ptr_host[i + 1]) - int(ptr_host[i]) for i in range(len(ptr_host) - 1)]]
    for _ in range(levels):
        sizes.append([max(1, n // decimation) if n > 0 else 0 for n in sizes[-1]])
    ptrs, totals = [], []
    for s in sizes:
        p = torch.zeros(len(s) + 1, dtype=torch.int64)
        p[1:] = torch.tensor(s, dtype=torch.int64).cumsum(0) if s else 0
        ptrs.append(p.to(device))
        totals.append(int(p[-1]))
    segs, num_edges = [], []
    for l in range(levels):
        keff = torch.tensor([min(num_neighbors, n) for n in sizes[l]], dtype=torch.int64)
        per_centre = torch.repeat_interleave(keff, torch.tensor(sizes[l + 1], dtype=torch.int64))
        seg = torch.zeros(per_centre.numel() + 1, dtype=torch.int64)
        seg[1:] = per_centre.cumsum(0)
        segs.append(seg.to(device))
        num_edges.append(int(seg[-1]))
    return SAPlan(sizes, ptrs, totals, segs, num_edges, [max(s) if s else 0 for s in sizes])


def _pad_cols(t: Tensor, to: int) -> Tensor:
    return t if t.shape[1] == to else F.pad(t, (0, to - t.shape[1]))


class HipPointNet2(nn.Module):
    """PointNet++ (set abstraction + feature propagation) for batched variable-size point clouds on one MI355X.

    Args (the reference net's, pyg_randla_net.py:23-30, plus the sampler):
        num_features, num_classes, decimation=4, num_neighbors=32, return_logits=False,
        subsampling="fps" | "random", random_start=False (FPS starts from a random point of each cloud when training)
    """

    def __init__(self, num_features: int, num_classes: int, decimation: int = 4, num_neighbors: int = 32,
                 return_logits: bool = False, subsampling: str = "fps", random_start: bool = False):
        super().__init__()
        if subsampling not in ("fps", "random"):
            raise ValueError(f"subsampling must be 'fps' or 'random', got {subsampling!r}")
        if not 1 <= num_neighbors <= 64:
            raise ValueError("num_neighbors must be in [1, 64]")
        self.num_features, self.num_classes = num_features, num_classes
        self.decimation, self.num_neighbors, self.return_logits = decimation, num_neighbors, return_logits
        self.subsampling, self.random_start = subsampling, random_start
        self.matmul_precision = "fp32"
        c = num_features
        sas = []
        for widths in SA_WIDTHS:
            sas.append(SAParams(SharedMLPParams([c + 3, *widths])))
            c = widths[-1]
        self.sa1, self.sa2, self.sa3 = sas
        self.fp3 = FPParams(SharedMLPParams([256 + 128, 128]))
        self.fp2 = FPParams(SharedMLPParams([128 + 64, 64]))
        self.fp1 = FPParams(SharedMLPParams([64 + num_features, 64]))
        self.mlp_classif = SharedMLPParams([64, 64, 32], dropout=[0.0, 0.5])
        self.fc_classif = nn.Linear(32, num_classes)
        self._plans: Dict[tuple, SAPlan] = {}
        self.last_sample_idx: List[Tensor] = []
        # position-only work of the NEXT batch (prefetch_geometry): (pos tensor, its version, ptr key, train flag, tables, event)
        self._look = None
        self._side = self._side_q = None
        # how many batches' position-only work may be in flight (prefetch_geometry): every one on a stream pair of its own.  The
        # farthest-point sampler is ONE serial chain per cloud — 33.5 ms for 16 x 40 000 points on 16 of the 256 CUs — so a
        # single chain in flight bounds the step at its latency; with three, a chain completes every ~11 ms and the step is
        # bound by the feature kernels (round 6).  A dataloader's prefetch queue holds the future batches' positions.
        self.prefetch_depth = 3
        self._sides: list = []
        self._look_n = 0

    # ------------------------------------------------------------------------------------------
    def plan_for(self, ptr: Tensor) -> SAPlan:
        key = tuple(ptr.tolist())
        plan = self._plans.get(key)
        if plan is None:
            if len(self._plans) > 64:
                self._plans.clear()
            plan = make_sa_plan(key, self.decimation, self.num_neighbors, ptr.device)
            self._plans[key] = plan
        return plan

    def _layer(self, mlp: SharedMLPParams, li: int, x0: Tensor, x1: Optional[Tensor], rows: Optional[Tensor], w: Tensor,
               train: bool, grad_eval: bool) -> Tensor:
        lin, bn = mlp.lins[li], mlp.norms[li].module
        if train:
            return ops.SharedLayerTrainFn.apply(x0, x1, w, lin.bias, bn.weight, bn.bias, bn, mlp.act, rows, None,
                                                self._bf16, None, None)
        if grad_eval:
            return ops.SharedLayerEvalFn.apply(x0, x1, w, lin.bias, bn.weight, bn.bias, bn, mlp.act, rows)
        scale, shift = ops.bn_fold_eval(bn)
        M = x1.shape[0] if x1 is not None else (rows.numel() if rows is not None else x0.shape[0])
        return ops.gemm(x0, w, M, w.shape[0], x0.shape[1], rows=rows, a1=x1, k1=0 if x1 is None else x1.shape[1],
                        bias=lin.bias, scale=scale, shift=shift, act=mlp.act, bf16=self._bf16)

    def _sample(self, lvl: int, pos4: Tensor, plan: SAPlan, train: bool, index=None) -> Tensor:
        m = plan.totals[lvl + 1]
        dev = pos4.device
        if self.subsampling == "fps":
            start = None
            if train and self.random_start:
                sizes = (plan.ptrs[lvl][1:] - plan.ptrs[lvl][:-1]).to(torch.float32)
                start = (torch.rand(sizes.numel(), device=dev) * sizes).to(torch.int32)
            return ops.fps(pos4, plan.ptrs[lvl], plan.ptrs[lvl + 1], m, plan.max_points[lvl], start, index=index)
        seed = torch.randint(-(1 << 62), 1 << 62, (1,), dtype=torch.int64, device=dev)
        return ops.decimation_indices(plan.ptrs[lvl], plan.ptrs[lvl + 1], m, seed, lvl)

    # ------------------------------------------------------------------------------------------
    def _geometry(self, pos: Tensor, plan: SAPlan, train: bool, sample_idx: Optional[List[Tensor]] = None,
                  query_stream: Optional["torch.cuda.Stream"] = None) -> dict:
        """Everything that depends on positions only, for all levels: padded positions, kNN grids, the sampled centres
        (farthest-point or random), the grouping tables and the decoder's 1-NN tables.  Farthest-point sampling is a serial
        chain on ONE compute unit per cloud (33.5 ms for 16 x 40 000 points on 16 of the 256 CUs): ``prefetch_geometry`` runs
        this one step ahead on a side stream, under the previous step's forward / backward (round 5; the RandLA net's kNN
        tables travel the same way)."""
        K = self.num_neighbors
        pos4 = [ops.pad_pos(pos.to(torch.float32).contiguous())]
        index = [ops.KnnIndex(pos4[0], plan.ptrs[0])]
        sels, nbrs = [], []
        cur = torch.cuda.current_stream()

        def on_query_stream(fn):
            # ``query_stream`` (prefetch only): the grouping / 1-NN queries leave the sampler's stream — the chain sampler ->
            # centres -> next level's sampler is the critical path (one CU per cloud), the queries fill the other 240 CUs beside it
            if query_stream is None:
                return fn()
            query_stream.wait_stream(cur)
            with torch.cuda.stream(query_stream):
                out = fn()
            return out

        for lvl in range(3):
            m = plan.totals[lvl + 1]
            if sample_idx is not None:
                sel = sample_idx[lvl].to(device=pos.device, dtype=torch.int32).contiguous()
                if sel.numel() != m:
                    raise ValueError(f"level {lvl}: {sel.numel()} sample indices given, {m} expected")
            else:
                sel = self._sample(lvl, pos4[lvl], plan, train, index[lvl])
            sels.append(sel)
            ctr = ops.gather_rows(pos4[lvl], sel)
            nbrs.append(on_query_stream(lambda: index[lvl].query(K, pos_qry=ctr, ptr_qry=plan.ptrs[lvl + 1])[0]))
            pos4.append(ctr)
            index.append(ops.KnnIndex(ctr, plan.ptrs[lvl + 1]))
        nn = on_query_stream(lambda: {lvl: index[lvl + 1].query(1, pos_qry=pos4[lvl], ptr_qry=plan.ptrs[lvl])[0]
                                      for lvl in (2, 1, 0)})
        if query_stream is not None:
            cur.wait_stream(query_stream)
        return {"pos4": pos4, "index": index, "sel": sels, "nbr": nbrs, "nn": nn}

    def prefetch_geometry(self, pos: Tensor, ptr: Tensor, wait_main: bool = True) -> None:
        """Enqueue the position-only work of the batch ``(pos, ptr)`` on a side stream now; a later ``forward`` on the SAME
        ``pos`` tensor (unchanged since: identity + version counter) and tile layout picks the tables up instead of
        computing them (first in, first out: up to ``prefetch_depth`` batches may be waiting, each on its own pair of streams).
        Call it for the NEXT batch(es) in front of the current step's ``forward``: the sampler then runs under whole steps.  ``wait_main=False``: ``pos`` is known to be
        complete (a resident / already transferred batch) — the side stream then does not wait for what the calling stream
        still has queued (the previous step's backward and optimizer).  Sampling is deterministic (farthest-point from point
        0 of every cloud; with ``random_start`` / ``subsampling="random"`` the draw happens here instead of in the forward)."""
        if not pos.is_cuda:
            raise RuntimeError("HipPointNet2 runs on an MI355X only (no CPU fallback by design)")
        plan = self.plan_for(ptr)
        main = torch.cuda.current_stream()
        depth = max(1, int(self.prefetch_depth))
        while len(self._sides) < depth:
            self._sides.append((torch.cuda.Stream(device=pos.device), torch.cuda.Stream(device=pos.device)))
        side, side_q = self._sides[self._look_n % depth]
        self._look_n += 1
        self._side, self._side_q = side, side_q
        if wait_main:
            side.wait_stream(main)  # (pos may just have been written on the calling stream)
        train = self.training
        with torch.cuda.stream(side), torch.no_grad():
            geo = self._geometry(pos, plan, train, query_stream=side_q)
            ev = torch.cuda.Event()
            ev.record(side)
        self._look = ((self._look or []) + [(pos, pos._version, tuple(plan.totals), train, geo, ev)])[-depth:]

    def _take_lookahead(self, pos: Tensor, plan: SAPlan, train: bool) -> Optional[dict]:
        queue = self._look or []
        hit = None
        for i, ent in enumerate(queue):  # oldest first
            if ent[0] is pos and ent[1] == pos._version and ent[2] == tuple(plan.totals) and ent[3] == train:
                hit = i
                break
        if hit is None:
            # prefetched for another batch / mode, or ``pos`` was written since: a stale entry for THIS tensor is dropped, entries
            # for other tensors (the batch after this one) stay
            self._look = [e for e in queue if e[0] is not pos] or None
            return None
        lpos, ver, totals, ltrain, geo, ev = queue[hit]
        self._look = (queue[:hit] + queue[hit + 1:]) or None
        main = torch.cuda.current_stream()
        main.wait_event(ev)
        for t in geo["pos4"] + geo["sel"] + geo["nbr"] + list(geo["nn"].values()):  # allocated on the side stream, read on this one
            t.record_stream(main)
        for ix in geo["index"]:
            ix.ws.record_stream(main)
        return geo

    def forward(self, x: Optional[Tensor], pos: Tensor, batch: Optional[Tensor], ptr: Tensor,
                sample_idx: Optional[List[Tensor]] = None, dropout_mask: Optional[Tensor] = None,
                record: Optional[Dict[str, Tensor]] = None) -> Tensor:
        """``sample_idx`` (per level: rows of that level to keep, cloud by cloud), ``dropout_mask`` and ``record`` exist
        for the parity tests, like ``HipRandLANet``'s."""
        if not pos.is_cuda:
            raise RuntimeError("HipPointNet2 runs on an MI355X only (no CPU fallback by design)")
        x = pos if x is None else x
        if x.shape[1] != self.num_features:
            raise ValueError(f"expected {self.num_features} features per point, got {x.shape[1]}")
        if self.matmul_precision not in ("fp32", "bf16"):
            raise ValueError(f"matmul_precision must be 'fp32' or 'bf16', got {self.matmul_precision!r}")
        self._bf16 = self.matmul_precision == "bf16"
        train = self.training
        grad_eval = (not train) and torch.is_grad_enabled() and (
            x.requires_grad or any(p.requires_grad for p in self.parameters()))
        diff = train or grad_eval
        plan = self.plan_for(ptr)
        ops.arena.stop()  # (the zero arena and the gradient side stream belong to HipRandLANet's flattened step)
        ops._grad_side = None
        x = x.to(torch.float32).contiguous()
        geo = self._take_lookahead(pos, plan, train) if sample_idx is None else None
        if geo is None:
            geo = self._geometry(pos, plan, train, sample_idx)
        pos4, index = geo["pos4"], geo["index"]
        feats: List[Tensor] = [x]
        h = x
        self.last_sample_idx = list(geo["sel"])
        for lvl, sa in enumerate((self.sa1, self.sa2, self.sa3)):
            m = plan.totals[lvl + 1]
            ctr, nbr = pos4[lvl + 1], geo["nbr"][lvl]
            C = h.shape[1]
            ldo = (C + 3 + 3) // 4 * 4  # 16-byte rows: the GEMM streams float4 fragments
            e, esrc, ectr = ops.SAGroupFn.apply(h, pos4[lvl], ctr, nbr, plan.segs[lvl], plan.num_edges[lvl], ldo)
            for li in range(len(sa.nn.lins)):
                w = sa.nn.lins[li].weight
                e = self._layer(sa.nn, li, e, None, None, _pad_cols(w, ldo) if li == 0 else w, train, grad_eval)
            h = ops.SegMaxFn.apply(e, plan.segs[lvl], ectr, m)
            if record is not None:
                record[f"sa{lvl + 1}"] = h
            feats.append(h)
        for fp, lvl in ((self.fp3, 2), (self.fp2, 1), (self.fp1, 0)):
            nn_idx = geo["nn"][lvl]
            skip = feats[lvl]
            kpad = (skip.shape[1] + 3) // 4 * 4
            w = fp.nn.lins[0].weight
            # knn_interpolate(k=1) == x[nn] (the weights cancel): a row gather fused into the GEMM's A operand
            h = self._layer(fp.nn, 0, h, _pad_cols(skip, kpad), nn_idx.view(-1),
                            _pad_cols(w, w.shape[1] + kpad - skip.shape[1]), train, grad_eval)
            if record is not None:
                record[f"fp{lvl + 1}"] = h
        for li in range(2):
            h = self._layer(self.mlp_classif, li, h, None, None, self.mlp_classif.lins[li].weight, train, grad_eval)
        p = self.mlp_classif.dropout[1]
        if train and p > 0.0:
            if dropout_mask is not None:
                h = h * (dropout_mask.to(h.dtype) / (1.0 - p))
            else:
                h = F.dropout(h, p=p, training=True)
        if diff:
            logits = ops.LinearFn.apply(h, self.fc_classif.weight, self.fc_classif.bias, None)
        else:
            logits = ops.gemm(h, self.fc_classif.weight, h.shape[0], self.fc_classif.weight.shape[0], h.shape[1],
                              bias=self.fc_classif.bias)
        if self.return_logits:
            return logits
        return logits.log_softmax(dim=-1)
