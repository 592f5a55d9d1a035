"""Synthetic Lidar-HD-shaped inputs for tests and ``bench.py``: pure torch, no HIP.

There is no network for datasets, and the reference's toy LAS blob is missing (``.MISSING_LARGE_BLOBS``), so the
benchmark tiles are generated: 50 m x 50 m, ground + vegetation + buildings, normalised the way the reference's
``NormalizePos`` does (``/root/reference/myria3d/pctl/transforms/transforms.py:141-162``).
"""
from __future__ import annotations

import math
from typing import Sequence

import numpy as np
import torch


def synthetic_tile(n: int, tile_id: int, num_features: int = 9, num_classes: int = 6):
    """One synthetic 50 m x 50 m Lidar-HD-shaped tile, normalised like the reference's ``NormalizePos``
    (``myria3d/pctl/transforms/transforms.py:141-162``): xy centred, z shifted to min 0, everything / 25."""
    g = torch.Generator().manual_seed(12345 + tile_id)
    xy = torch.rand(n, 2, generator=g) * 50.0 - 25.0
    z0 = 2.0 * torch.sin(2 * math.pi * xy[:, 0] / 50.0) + 1.5 * torch.cos(2 * math.pi * xy[:, 1] / 37.0)
    u = torch.rand(n, generator=g)
    cls = torch.zeros(n, dtype=torch.int64)
    z = z0 + 0.05 * torch.randn(n, generator=g)  # ground
    veg = (u >= 0.45) & (u < 0.80)
    z = torch.where(veg, z0 + 15.0 * torch.rand(n, generator=g), z)
    cls[veg] = 1
    bld = (u >= 0.80) & (u < 0.95)
    nb = 3
    c = torch.rand(nb, 2, generator=g) * 30.0 - 15.0
    half = (8.0 + 7.0 * torch.rand(nb, 2, generator=g)) / 2
    hb = 3.0 + 9.0 * torch.rand(nb, generator=g)
    which = torch.randint(0, nb, (n,), generator=g)
    bxy = c[which] + (torch.rand(n, 2, generator=g) * 2 - 1) * half[which]
    xy = torch.where(bld[:, None], bxy, xy)
    z = torch.where(bld, z0 + hb[which] + 0.1 * torch.randn(n, generator=g), z)
    cls[bld] = 2
    oth = u >= 0.95
    z = torch.where(oth, z0 + 3.0 * torch.rand(n, generator=g), z)
    cls[oth] = torch.randint(3, max(4, num_classes), (int(oth.sum()),), generator=g)
    pos = torch.cat([xy, z[:, None]], dim=1)
    pos = pos - pos.mean(dim=0, keepdim=True)
    pos[:, 2] = pos[:, 2] - pos[:, 2].min()
    pos = (pos / 25.0).to(torch.float32)
    x = torch.rand(n, num_features, generator=g)
    x[:, 0] = torch.randn(n, generator=g).clamp(-3, 3)
    if num_features > 7:
        x[:, 7] = torch.randn(n, generator=g).clamp(-3, 3)
    if num_features > 2:
        x[:, 1] = torch.randint(1, 6, (n,), generator=g) / 7.0
        x[:, 2] = torch.randint(1, 6, (n,), generator=g) / 7.0
    if num_features > 8:
        x[:, 8] = torch.rand(n, generator=g) * 2 - 1
    y = cls.clamp(max=num_classes - 1)
    return x.to(torch.float32).contiguous(), pos.contiguous(), y


def synthetic_batch(sizes: Sequence[int], first_tile_id: int = 0, num_features: int = 9, num_classes: int = 6):
    xs, ps, ys, bs = [], [], [], []
    for b, n in enumerate(sizes):
        x, p, y = synthetic_tile(n, first_tile_id + b, num_features, num_classes)
        xs.append(x), ps.append(p), ys.append(y), bs.append(torch.full((n,), b, dtype=torch.int64))
    ptr = torch.tensor([0] + list(np.cumsum(sizes)), dtype=torch.int64)
    return torch.cat(xs), torch.cat(ps), torch.cat(bs), ptr, torch.cat(ys)
