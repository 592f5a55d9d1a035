"""CPU oracle for the PointNet++ set-abstraction variant — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

PARITY UNPINNED: the reference has no implementation of this variant to pin the oracle to
(``/root/reference/myria3d/models/model.py:12``: ``MODEL_ZOO = [PyGRandLANet]``; no farthest-point sampling anywhere in
the repository), and PyG / torch_cluster are absent from the image.  The functions restate the published behaviour of the
third-party operators the variant is made of, op for op and unfused:

* ``fps_exact``         torch_cluster.fps(random_start=False): per cloud, start from one point, then repeatedly take the
                        point farthest from everything selected so far (fp32 ``(dx*dx + dy*dy) + dz*dz``, first index on ties)
* ``PointNet2Oracle``   PyG's PointNet++ segmentation structure (SAModule = sample -> group -> PointNetConv(local_nn,
                        aggr="max") on ``[x_j, pos_j - pos_i]``; FPModule = knn_interpolate + skip concat + nn) written with
                        the reference's own blocks: ``SharedMLP`` (pyg_randla_net.py:97-109), ``FPModule(k=1)``
                        (pyg_randla_net.py:241-253), the classification head (pyg_randla_net.py:52-53,81-88) and the
                        count rule of ``decimation_indices`` (pyg_randla_net.py:215-217).  Grouping is by the K nearest
                        points of the level (``knn_exact``), max aggregation gives its gradient to the first arg-max edge
                        (torch_scatter.scatter_max).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s baseline legs may import this module.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
from torch import Tensor, nn

from .randla_oracle import SharedMLP, _FP, decimation_indices, knn_exact

SA_WIDTHS = ((32, 32, 64), (64, 64, 128), (128, 128, 256))


def fps_exact(pos: Tensor, ptr: Sequence[int], ptr_out: Sequence[int], start: Optional[Sequence[int]] = None) -> Tensor:
    """int64 ``[ptr_out[-1]]`` global rows, cloud by cloud, in selection order."""
    p = pos.detach().cpu().numpy().astype(np.float32)
    out: List[int] = []
    for b in range(len(ptr) - 1):
        s0, s1 = int(ptr[b]), int(ptr[b + 1])
        m = int(ptr_out[b + 1]) - int(ptr_out[b])
        n = s1 - s0
        if m <= 0 or n <= 0:
            continue
        pts = p[s0:s1]
        mind = np.full(n, np.inf, dtype=np.float32)
        cur = int(start[b]) if start is not None else 0
        cur = min(max(cur, 0), n - 1)
        for s in range(m):
            out.append(s0 + cur)
            if s == m - 1:
                break
            d = pts - pts[cur]
            d2 = d[:, 0] * d[:, 0]
            d2 = d2 + d[:, 1] * d[:, 1]
            d2 = d2 + d[:, 2] * d[:, 2]
            mind = np.minimum(mind, d2)
            cur = int(np.argmax(mind))  # first index on ties
    return torch.tensor(out, dtype=torch.int64)


def level_sizes(ptr: Sequence[int], decimation: int) -> List[int]:
    new = [0]
    for b in range(len(ptr) - 1):
        n = int(ptr[b + 1]) - int(ptr[b])
        new.append(new[-1] + (max(1, n // decimation) if n > 0 else 0))
    return new


class _SA(nn.Module):
    def __init__(self, mlp: SharedMLP):
        super().__init__()
        self.nn = mlp

    def forward(self, x: Tensor, pos: Tensor, ptr: Sequence[int], idx: Tensor, ptr_c: Sequence[int], k: int) -> Tensor:
        pos_c = pos[idx]
        nbr, _ = knn_exact(pos, ptr, pos_c, ptr_c, k)  # [m, k] global rows of this level, -1 padding
        m = nbr.shape[0]
        valid = nbr >= 0
        i = torch.arange(m)[:, None].expand(m, k)[valid]
        j = nbr[valid]
        msg = torch.cat([x[j], (pos[j] - pos_c[i]).to(x.dtype)], dim=1)  # PointNetConv.message
        h = self.nn(msg)
        dense = h.new_full((m, k, h.shape[1]), float("-inf"))
        dense[valid] = h  # valid entries are the leading ones of every row: edge order = (centre, rank)
        val, arg = dense.max(dim=1)  # (gradient to one arg-max edge per (centre, channel))
        return val


class PointNet2Oracle(nn.Module):
    def __init__(self, num_features: int, num_classes: int, decimation: int = 4, num_neighbors: int = 32,
                 return_logits: bool = False, subsampling: str = "fps"):
        super().__init__()
        self.decimation, self.num_neighbors, self.return_logits = decimation, num_neighbors, return_logits
        self.subsampling = subsampling
        c = num_features
        sas = []
        for widths in SA_WIDTHS:
            sas.append(_SA(SharedMLP([c + 3, *widths])))
            c = widths[-1]
        self.sa1, self.sa2, self.sa3 = sas
        self.fp3 = _FP(SharedMLP([256 + 128, 128]))
        self.fp2 = _FP(SharedMLP([128 + 64, 64]))
        self.fp1 = _FP(SharedMLP([64 + num_features, 64]))
        self.mlp_classif = SharedMLP([64, 64, 32], dropout=[0.0, 0.5])
        self.fc_classif = nn.Linear(32, num_classes)

    def forward(self, x: Optional[Tensor], pos: Tensor, batch: Optional[Tensor], ptr: Tensor,
                sample_idx: Optional[List[Tensor]] = None, dropout_mask: Optional[Tensor] = None,
                record: Optional[Dict[str, Tensor]] = None) -> Tensor:
        x = pos if x is None else x
        ptrs = [[int(v) for v in ptr]]
        poss, feats = [pos], [x]
        h = x
        used = []
        for lvl, sa in enumerate((self.sa1, self.sa2, self.sa3)):
            new_ptr = level_sizes(ptrs[lvl], self.decimation)
            if sample_idx is not None:
                idx = sample_idx[lvl].to(torch.int64)
            elif self.subsampling == "fps":
                idx = fps_exact(poss[lvl], ptrs[lvl], new_ptr)
            else:
                idx, _ = decimation_indices(ptrs[lvl], self.decimation)
            assert idx.numel() == new_ptr[-1]
            used.append(idx)
            h = sa(h, poss[lvl], ptrs[lvl], idx, new_ptr, self.num_neighbors)
            if record is not None:
                record[f"sa{lvl + 1}"] = h
            feats.append(h)
            poss.append(poss[lvl][idx])
            ptrs.append(new_ptr)
        self.last_sample_idx = used
        for fp, lvl in ((self.fp3, 2), (self.fp2, 1), (self.fp1, 0)):
            h = fp(h, poss[lvl + 1], ptrs[lvl + 1], feats[lvl], poss[lvl], ptrs[lvl], "exact")
            if record is not None:
                record[f"fp{lvl + 1}"] = h
        h = self.mlp_classif(h, dropout_masks=[None, dropout_mask])
        logits = self.fc_classif(h)
        if self.return_logits:
            return logits
        return logits.log_softmax(dim=-1)
