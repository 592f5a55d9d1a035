"""CPU oracle for the RandLA-Net hot path — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Op-for-op, unfused, pure-torch (fp32, CPU) restatement of

* ``myria3d/models/modules/pyg_randla_net.py:22-253`` (net, SharedMLP, LocalFeatureAggregation,
  DilatedResidualBlock, decimation, FPModule), and
* the third-party semantics those lines call into (PyG 2.4 ``MLP`` / ``BatchNorm`` / ``knn_graph`` /
  ``MessagePassing.propagate(aggr="add")`` / ``utils.softmax`` / ``knn_interpolate``; torch_cluster
  ``knn``; torch_scatter ``scatter_sum``) — restated from their published behaviour because the wheels
  are absent from this image.

Parity: pinned to the reference's own file through ``tests/_pyg_stub`` (round 3), third-party semantics restated; see ``oracle/__init__.py``.

The parameter tree reproduces the reference's ``state_dict`` keys (``block1.lfa1.mlp_encoder.lins.0.weight``,
``...norms.0.module.running_mean`` ...) so a Myria3D checkpoint loads unchanged.

Two kNN back-ends:
  * ``"exact"``  — brute force in fp32 with the explicit op order ``(dx*dx + dy*dy) + dz*dz`` and a total
    order ``(d2, index)``; this is what the HIP kernel is compared against bit-for-bit.
  * ``"kdtree"`` — ``scipy.spatial.cKDTree`` per cloud (mirrors torch_cluster's nanoflann CPU path); used
    for the timed CPU baseline and as an independent cross-check of ``"exact"``.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F
from torch import Tensor, nn

LRELU_SLOPE = 0.2  # pyg_randla_net.py:92
BN_MOMENTUM = 0.01  # pyg_randla_net.py:94
BN_EPS = 1e-6  # pyg_randla_net.py:94


# --------------------------------------------------------------------------------------
# third-party semantics, restated
# --------------------------------------------------------------------------------------
def knn_exact(
    pos_src: Tensor, ptr_src: Sequence[int], pos_qry: Tensor, ptr_qry: Sequence[int], k: int
) -> Tuple[Tensor, Tensor]:
    """torch_cluster.knn semantics: per cloud, the ``min(k, n_src)`` nearest sources
    of every query by squared L2, ascending; ties broken by source index (our convention — upstream leaves
    tie order unspecified).  Returns dense ``idx[int64, Nq, k]`` (global source indices, ``-1`` padding) and
    ``d2[float32, Nq, k]`` (``+inf`` padding).  Distance arithmetic is fp32 ``(dx*dx + dy*dy) + dz*dz``.
    """
    nq = pos_qry.shape[0]
    idx = torch.full((nq, k), -1, dtype=torch.int64)
    d2o = torch.full((nq, k), float("inf"), dtype=torch.float32)
    ps = pos_src.detach().to(torch.float32)
    pq = pos_qry.detach().to(torch.float32)
    for b in range(len(ptr_src) - 1):
        s0, s1 = int(ptr_src[b]), int(ptr_src[b + 1])
        q0, q1 = int(ptr_qry[b]), int(ptr_qry[b + 1])
        ns = s1 - s0
        if ns == 0 or q1 == q0:
            continue
        keff = min(k, ns)
        S = ps[s0:s1]
        for c0 in range(q0, q1, 1024):
            c1 = min(q1, c0 + 1024)
            Q = pq[c0:c1]
            dx = S[None, :, 0] - Q[:, None, 0]
            dy = S[None, :, 1] - Q[:, None, 1]
            dz = S[None, :, 2] - Q[:, None, 2]
            d2 = (dx * dx + dy * dy) + dz * dz  # [q, ns]
            if ns <= 4096:
                dsort, isort = torch.sort(d2, dim=1, stable=True)
                dk, ik = dsort[:, :keff], isort[:, :keff]
            else:
                m = min(ns, keff + 48)
                dc, ic = torch.topk(d2, m, dim=1, largest=False, sorted=True)
                # lexicographic (d2, idx): sort by idx first (stable), then by d2 (stable)
                o1 = torch.argsort(ic, dim=1, stable=True)
                dc, ic = torch.gather(dc, 1, o1), torch.gather(ic, 1, o1)
                o2 = torch.argsort(dc, dim=1, stable=True)
                dc, ic = torch.gather(dc, 1, o2), torch.gather(ic, 1, o2)
                dk, ik = dc[:, :keff].clone(), ic[:, :keff].clone()
                if m < ns:
                    # rows whose k-th distance ties with the candidate horizon need the full sort
                    bad = (dc[:, keff - 1] >= dc[:, m - 1]).nonzero().flatten()
                    if bad.numel():
                        ds2, is2 = torch.sort(d2[bad], dim=1, stable=True)
                        dk[bad], ik[bad] = ds2[:, :keff], is2[:, :keff]
            idx[c0:c1, :keff] = ik + s0
            d2o[c0:c1, :keff] = dk
    return idx, d2o


def knn_kdtree(
    pos_src: Tensor, ptr_src: Sequence[int], pos_qry: Tensor, ptr_qry: Sequence[int], k: int
) -> Tuple[Tensor, Tensor]:
    """Same contract as :func:`knn_exact` through ``scipy.spatial.cKDTree`` (fp64 tree; leaf size 10 as
    torch_cluster's nanoflann adaptor).  Tie order is whatever the tree returns."""
    from scipy.spatial import cKDTree

    nq = pos_qry.shape[0]
    idx = torch.full((nq, k), -1, dtype=torch.int64)
    d2o = torch.full((nq, k), float("inf"), dtype=torch.float32)
    ps = pos_src.detach().numpy()
    pq = pos_qry.detach().numpy()
    for b in range(len(ptr_src) - 1):
        s0, s1 = int(ptr_src[b]), int(ptr_src[b + 1])
        q0, q1 = int(ptr_qry[b]), int(ptr_qry[b + 1])
        ns = s1 - s0
        if ns == 0 or q1 == q0:
            continue
        keff = min(k, ns)
        tree = cKDTree(ps[s0:s1], leafsize=10)
        d, i = tree.query(pq[q0:q1], k=keff, workers=-1)
        if keff == 1:
            d, i = d[:, None], i[:, None]
        idx[q0:q1, :keff] = torch.from_numpy(i.astype(np.int64)) + s0
        d2o[q0:q1, :keff] = torch.from_numpy((d * d).astype(np.float32))
    return idx, d2o


def knn_cdist(
    pos_src: Tensor, ptr_src: Sequence[int], pos_qry: Tensor, ptr_qry: Sequence[int], k: int
) -> Tuple[Tensor, Tensor]:
    """Same contract through stock torch ops on whatever device the positions live on (``torch.cdist`` + ``topk`` per
    cloud): the kNN of the "restated path through stock PyTorch-ROCm ops" baseline (BASELINE.md section 3,
    ``bench.py`` ``torch_rocm_baseline``).  Not bit-exact with :func:`knn_exact` (cdist's arithmetic)."""
    nq, dev = pos_qry.shape[0], pos_qry.device
    idx = torch.full((nq, k), -1, dtype=torch.int64, device=dev)
    d2o = torch.full((nq, k), float("inf"), dtype=torch.float32, device=dev)
    for b in range(len(ptr_src) - 1):
        s0, s1 = int(ptr_src[b]), int(ptr_src[b + 1])
        q0, q1 = int(ptr_qry[b]), int(ptr_qry[b + 1])
        if s1 == s0 or q1 == q0:
            continue
        keff = min(k, s1 - s0)
        d = torch.cdist(pos_qry[q0:q1].detach(), pos_src[s0:s1].detach())
        dk, ik = torch.topk(d, keff, dim=1, largest=False, sorted=True)
        idx[q0:q1, :keff] = ik + s0
        d2o[q0:q1, :keff] = dk * dk
    return idx, d2o


_KNN = {"exact": knn_exact, "kdtree": knn_kdtree, "cdist": knn_cdist}


def dense_to_edge_index(idx: Tensor) -> Tensor:
    """``knn_graph(..., loop=True, flow='source_to_target')`` layout:
    row 0 = neighbour j (source), row 1 = centre i (target); edges grouped by centre, ascending distance."""
    n, k = idx.shape
    centre = torch.arange(n, dtype=torch.int64, device=idx.device)[:, None].expand(n, k)
    keep = idx >= 0
    return torch.stack([idx[keep], centre[keep]], dim=0)


def scatter_sum(src: Tensor, index: Tensor, dim_size: int) -> Tensor:
    """torch_scatter.scatter_sum(src, index, dim=0, dim_size=...) (Appendix A.6)."""
    out = torch.zeros((dim_size,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    return out.index_add(0, index, src)


def segment_softmax(src: Tensor, index: Tensor, num_nodes: int) -> Tensor:
    """torch_geometric.utils.softmax(src, index) (Appendix A.4): per-channel softmax over the rows that
    share ``index``; max subtracted on a detached copy; ``+1e-16`` in the denominator."""
    c = src.shape[1]
    mx = torch.full((num_nodes, c), float("-inf"), dtype=src.dtype, device=src.device)
    mx = mx.scatter_reduce(0, index[:, None].expand(-1, c), src.detach(), reduce="amax", include_self=True)
    e = torch.exp(src - mx[index])
    s = scatter_sum(e, index, num_nodes) + 1e-16
    return e / s[index]


def knn_interpolate(
    x: Tensor,
    pos_x: Tensor,
    pos_y: Tensor,
    ptr_x: Sequence[int],
    ptr_y: Sequence[int],
    k: int,
    knn: str = "exact",
    nn_idx: Optional[Tensor] = None,
) -> Tensor:
    """torch_geometric.nn.knn_interpolate (Appendix A.5): inverse-squared-distance interpolation of ``x``
    (living on ``pos_x``) onto ``pos_y``; weights carry no gradient; gradient flows to ``x`` only."""
    with torch.no_grad():
        if nn_idx is None:
            nn_idx, _ = _KNN[knn](pos_x, ptr_x, pos_y, ptr_y, k)
        ny = pos_y.shape[0]
        keep = nn_idx >= 0
        y_idx = torch.arange(ny, dtype=torch.int64, device=nn_idx.device)[:, None].expand_as(nn_idx)[keep]
        x_idx = nn_idx[keep]
        diff = pos_x[x_idx] - pos_y[y_idx]
        d2 = (diff * diff).sum(dim=-1, keepdim=True)
        w = 1.0 / torch.clamp(d2, min=1e-16)
    num = scatter_sum(x[x_idx] * w, y_idx, ny)
    den = scatter_sum(w, y_idx, ny)
    return num / den


def interpolator_reduce(logits_list: Sequence[Tensor], idx_list: Sequence[Tensor], nb_points: int):
    """Arithmetic of ``Interpolator.reduce_predicted_logits`` + ``reduce_predictions_and_save``
    (myria3d/models/interpolation.py:98-121, 142-164) without the LAS I/O: concatenate the stored logits, sum the
    predictions that land on the same point of the full cloud, read them back per stored prediction, then
    ``Softmax(dim=1)``, ``argmax(dim=1)`` and ``Categorical(probs=probas).entropy()``.
    Returns ``(reduced_logits[idx], probas, preds, entropy, idx)``."""
    logits = torch.cat([l.cpu() for l in logits_list])
    idx = torch.cat([torch.as_tensor(i).reshape(-1).to(torch.int64) for i in idx_list])
    reduced = torch.zeros((nb_points, logits.size(1)))
    reduced = reduced + scatter_sum(logits, idx, nb_points)          # scatter_sum(..., out=reduced, dim=0)
    rows = reduced[idx]
    probas = torch.nn.Softmax(dim=1)(rows)
    preds = torch.argmax(rows, dim=1)
    entropy = torch.distributions.Categorical(probs=probas).entropy()
    return rows, probas, preds, entropy, idx


# --------------------------------------------------------------------------------------
# PyG MLP / SharedMLP restated (pyg_randla_net.py:97-109; Appendix A.1)
# --------------------------------------------------------------------------------------
class _PygBatchNorm(nn.Module):
    """PyG ``BatchNorm`` wraps ``torch.nn.BatchNorm1d`` as ``.module`` (→ key ``norms.i.module.*``)."""

    def __init__(self, channels: int):
        super().__init__()
        self.module = nn.BatchNorm1d(channels, eps=BN_EPS, momentum=BN_MOMENTUM)

    def forward(self, x: Tensor) -> Tensor:
        return self.module(x)


class SharedMLP(nn.Module):
    """``SharedMLP(channel_list, dropout=0., act='LeakyReLU'|None, norm='batch_norm'|None, bias=True)`` with
    ``plain_last=False``: every layer is Linear → norm → act → dropout."""

    def __init__(self, channels: Sequence[int], dropout=0.0, act: bool = True, norm: bool = True, bias: bool = True):
        super().__init__()
        nl = len(channels) - 1
        self.dropout = list(dropout) if isinstance(dropout, (list, tuple)) else [float(dropout)] * nl
        assert len(self.dropout) == nl
        self.act = act
        self.lins = nn.ModuleList([nn.Linear(channels[i], channels[i + 1], bias=bias) for i in range(nl)])
        self.norms = nn.ModuleList([_PygBatchNorm(channels[i + 1]) if norm else nn.Identity() for i in range(nl)])

    def forward(self, x: Tensor, dropout_masks: Optional[List[Optional[Tensor]]] = None) -> Tensor:
        for i, (lin, norm) in enumerate(zip(self.lins, self.norms)):
            x = norm(lin(x))
            if self.act:
                x = F.leaky_relu(x, LRELU_SLOPE)
            p = self.dropout[i]
            if p > 0.0 and self.training:
                if dropout_masks is not None and dropout_masks[i] is not None:
                    x = x * dropout_masks[i] / (1.0 - p)  # injected keep-mask (parity runs)
                else:
                    x = F.dropout(x, p=p, training=True)
        return x


# --------------------------------------------------------------------------------------
# the net (pyg_randla_net.py:22-253)
# --------------------------------------------------------------------------------------
class LocalFeatureAggregation(nn.Module):
    """pyg_randla_net.py:112-152 with ``propagate`` (gather j/i, message, scatter-add over i) written out."""

    def __init__(self, channels: int):
        super().__init__()
        self.mlp_encoder = SharedMLP([10, channels // 2])
        self.mlp_attention = SharedMLP([channels, channels], bias=False, act=False, norm=False)
        self.mlp_post_attention = SharedMLP([channels, channels])

    def aggregate(self, edge_index: Tensor, x: Tensor, pos: Tensor) -> Tensor:
        j, i = edge_index[0], edge_index[1]
        x_j, pos_i, pos_j = x[j], pos[i], pos[j]
        pos_diff = pos_j - pos_i
        distance = torch.sqrt((pos_diff * pos_diff).sum(1, keepdim=True))
        relative_infos = torch.cat([pos_i, pos_j, pos_diff, distance], dim=1)  # [E,10]
        local_spatial_encoding = self.mlp_encoder(relative_infos)
        local_features = torch.cat([x_j, local_spatial_encoding], dim=1)
        att_features = self.mlp_attention(local_features)
        att_scores = segment_softmax(att_features, i, x.shape[0])
        return scatter_sum(att_scores * local_features, i, x.shape[0])

    def forward(self, edge_index: Tensor, x: Tensor, pos: Tensor, rec: Optional[dict] = None, name: str = "") -> Tensor:
        agg = self.aggregate(edge_index, x, pos)
        if rec is not None:
            rec[name + "_agg"] = agg
        return self.mlp_post_attention(agg)


class DilatedResidualBlock(nn.Module):
    """pyg_randla_net.py:155-189."""

    def __init__(self, num_neighbors: int, d_in: int, d_out: int):
        super().__init__()
        self.num_neighbors = num_neighbors
        self.mlp1 = SharedMLP([d_in, d_out // 8])
        self.shortcut = SharedMLP([d_in, d_out], act=False)
        self.mlp2 = SharedMLP([d_out // 2, d_out], act=False)
        self.lfa1 = LocalFeatureAggregation(d_out // 4)
        self.lfa2 = LocalFeatureAggregation(d_out // 2)

    def forward(self, x: Tensor, pos: Tensor, ptr: Sequence[int], knn: str, rec: Optional[dict], name: str):
        idx, d2 = _KNN[knn](pos, ptr, pos, ptr, self.num_neighbors)
        edge_index = dense_to_edge_index(idx)
        shortcut_of_x = self.shortcut(x)
        x = self.mlp1(x)
        if rec is not None:
            rec[name + ".knn_idx"], rec[name + ".knn_d2"] = idx, d2
            rec[name + ".mlp1"] = x
        x = self.lfa1(edge_index, x, pos, rec, name + ".lfa1")
        if rec is not None:
            rec[name + ".lfa1"] = x
        x = self.lfa2(edge_index, x, pos, rec, name + ".lfa2")
        x = self.mlp2(x)
        x = F.leaky_relu(x + shortcut_of_x, LRELU_SLOPE)
        if rec is not None:
            rec[name + ".out"] = x
        return x


def decimation_indices(ptr: Sequence[int], factor: int, generator: Optional[torch.Generator] = None, device=None):
    """pyg_randla_net.py:192-231: per cloud keep ``max(1, n // factor)`` points, the head of a random
    permutation; returns (global indices, new ptr)."""
    if factor < 1:
        raise ValueError(
            "Argument `decimation_factor` should be higher than (or equal to) 1 for downsampling. "
            f"(Current value: {factor})"
        )
    idx, new_ptr = [], [0]
    for b in range(len(ptr) - 1):
        n = int(ptr[b + 1]) - int(ptr[b])
        m = max(1, n // factor)
        idx.append(int(ptr[b]) + torch.randperm(n, generator=generator, device=device)[:m])
        new_ptr.append(new_ptr[-1] + m)
    return torch.cat(idx), new_ptr


class RandLANetOracle(nn.Module):
    """``PyGRandLANet(num_features, num_classes, decimation=4, num_neighbors=16, return_logits=False)``
    (pyg_randla_net.py:22-88) on CPU.  Extras for testing only: ``knn`` back-end, injected decimation
    indices / dropout keep-mask, and an optional record of intermediates."""

    def __init__(self, num_features: int, num_classes: int, decimation: int = 4, num_neighbors: int = 16,
                 return_logits: bool = False, knn: str = "exact"):
        super().__init__()
        self.decimation, self.return_logits, self.knn = decimation, return_logits, knn
        self.num_neighbors = num_neighbors
        d_bottleneck = max(32, num_classes, num_features)
        self.fc0 = nn.Linear(num_features, d_bottleneck)
        self.block1 = DilatedResidualBlock(num_neighbors, d_bottleneck, 32)
        self.block2 = DilatedResidualBlock(num_neighbors, 32, 128)
        self.block3 = DilatedResidualBlock(num_neighbors, 128, 256)
        self.block4 = DilatedResidualBlock(num_neighbors, 256, 512)
        self.mlp_summit = SharedMLP([512, 512])
        self.fp4 = _FP(SharedMLP([512 + 256, 256]))
        self.fp3 = _FP(SharedMLP([256 + 128, 128]))
        self.fp2 = _FP(SharedMLP([128 + 32, 32]))
        self.fp1 = _FP(SharedMLP([32 + 32, d_bottleneck]))
        self.mlp_classif = SharedMLP([d_bottleneck, 64, 32], dropout=[0.0, 0.5])
        self.fc_classif = nn.Linear(32, num_classes)

    def forward(self, x: Optional[Tensor], pos: Tensor, batch: Optional[Tensor], ptr: Tensor,
                decimation_idx: Optional[List[Tensor]] = None, dropout_mask: Optional[Tensor] = None,
                record: Optional[Dict[str, Tensor]] = None) -> Tensor:
        x = x if x is not None else pos
        ptrs = [[int(v) for v in ptr]]
        xs, poss = [], [pos]
        h = self.fc0(x)
        used_idx = []
        for lvl, block in enumerate((self.block1, self.block2, self.block3, self.block4)):
            h = block(h, poss[lvl], ptrs[lvl], self.knn, record, f"block{lvl + 1}")
            xs.append(h)
            if decimation_idx is not None:
                idx = decimation_idx[lvl].to(torch.int64)
                new_ptr = [0]
                for b in range(len(ptrs[lvl]) - 1):
                    new_ptr.append(new_ptr[-1] + max(1, (ptrs[lvl][b + 1] - ptrs[lvl][b]) // self.decimation))
                assert idx.numel() == new_ptr[-1]
            else:
                idx, new_ptr = decimation_indices(ptrs[lvl], self.decimation, device=pos.device)
            used_idx.append(idx)
            h = h[idx]
            poss.append(poss[lvl][idx])
            ptrs.append(new_ptr)
        self.last_decimation_idx = used_idx
        h = self.mlp_summit(h)
        if record is not None:
            record["summit"] = h
        # decoder (pyg_randla_net.py:76-79): fp4 onto level-4 points ... fp1 onto level-1 points
        skips = [xs[0], xs[0][used_idx[0]], xs[1][used_idx[1]], xs[2][used_idx[2]]]
        # skip features are the *decimated* block outputs for fp4..fp2, and the undecimated block1 output for fp1
        for fp, lvl in ((self.fp4, 3), (self.fp3, 2), (self.fp2, 1), (self.fp1, 0)):
            h = fp(h, poss[lvl + 1], ptrs[lvl + 1], skips[lvl], poss[lvl], ptrs[lvl], self.knn)
            if record is not None:
                record[f"fp{lvl + 1}"] = h
        h = self.mlp_classif(h, dropout_masks=[None, dropout_mask])
        logits = self.fc_classif(h)
        if self.return_logits:
            return logits
        return logits.log_softmax(dim=-1)


class _FP(nn.Module):
    """FPModule(k=1, nn) (pyg_randla_net.py:241-253): 1-NN interpolate, concat the skip, SharedMLP."""

    def __init__(self, mlp: SharedMLP):
        super().__init__()
        self.k = 1
        self.nn = mlp

    def forward(self, x, pos, ptr, x_skip, pos_skip, ptr_skip, knn):
        x = knn_interpolate(x, pos, pos_skip, ptr, ptr_skip, k=self.k, knn=knn)
        return self.nn(torch.cat([x, x_skip], dim=1))


# synthetic inputs live with the product's bench helpers; re-exported here for the tests
from myria3d_amd.synthetic import synthetic_batch, synthetic_tile  # noqa: E402,F401


def fixed_decimation_indices(ptr: Sequence[int], factor: int, levels: int = 4, seed: int = 0) -> List[Tensor]:
    """Deterministic per-level decimation indices (seeded ``randperm``) for parity runs."""
    g = torch.Generator().manual_seed(seed)
    out, p = [], [int(v) for v in ptr]
    for _ in range(levels):
        idx, p = decimation_indices(p, factor, generator=g)
        out.append(idx)
    return out
