"""CPU restatement of the reference's per-tile data preparation (TEST INFRASTRUCTURE ONLY — see oracle/__init__.py).

Scope: what `configs/datamodule/transforms/preparations/points_budget.yaml` and
`configs/datamodule/transforms/normalizations/default.yaml` apply to every tile between the HDF5 read and the
collater, on the CPU, in the dataloader workers:

  GridSampling(0.25)            torch_geometric.transforms.GridSampling  (points_budget.yaml:14-17)
  MinimumNumNodes(300)          myria3d/pctl/transforms/transforms.py:66-87
  MaximumNumNodes(40000)        myria3d/pctl/transforms/transforms.py:48-63
  Center                        torch_geometric.transforms.Center        (points_budget.yaml:29-30)
  NullifyLowestZ                myria3d/pctl/transforms/transforms.py:141-146
  NormalizePos                  myria3d/pctl/transforms/transforms.py:149-162
  StandardizeRGBAndIntensity    myria3d/pctl/transforms/transforms.py:115-138

PARITY UNPINNED for GridSampling / Center: they live in torch_geometric 2.4 / torch_cluster (not installed here, no
fixtures in the reference); restated from their published behaviour:
  voxel_grid -> torch_cluster.grid_cluster: start = pos.min(0), end = pos.max(0),
      num_voxels = ((end - start) / size).long() + 1, strides = [1, n0, n0*n1],
      cluster = sum_d ((pos_d - start_d) / size).long() * stride_d
  consecutive_cluster: unique(cluster, sorted) -> new ids in ascending order of the old ones
  per key: "y" -> one_hot -> scatter sum -> argmax (first maximum); other node-level tensors -> scatter mean
      (sum / count.clamp(min=1)).
The myria3d transforms are restated from the reference's own source (cited above).
"""
from __future__ import annotations

import math
from typing import Optional, Sequence, Tuple

import numpy as np
import torch
from torch import Tensor


def grid_sampling(pos: Tensor, x: Optional[Tensor], y: Optional[Tensor], size: float):
    """One tile through GridSampling(size): returns (pos', x', y', cluster) with rows in ascending voxel id."""
    n = pos.shape[0]
    if n == 0:
        return pos, x, y, torch.zeros(0, dtype=torch.int64)
    size_t = torch.tensor([size] * 3, dtype=pos.dtype)
    start, end = pos.min(0).values, pos.max(0).values
    num_voxels = ((end - start) / size_t).to(torch.int64) + 1
    strides = torch.cat([torch.ones(1, dtype=torch.int64), num_voxels.cumprod(0)])[:3]
    cluster = (((pos - start[None]) / size_t[None]).to(torch.int64) * strides[None]).sum(1)
    uniq, inv = torch.unique(cluster, sorted=True, return_inverse=True)
    m = uniq.numel()
    count = torch.zeros(m, dtype=pos.dtype).index_add_(0, inv, torch.ones(n, dtype=pos.dtype)).clamp(min=1)

    def mean(t: Tensor) -> Tensor:
        return torch.zeros((m, t.shape[1]), dtype=t.dtype).index_add_(0, inv, t) / count[:, None]

    pos_o = mean(pos)
    x_o = mean(x) if x is not None else None
    y_o = None
    if y is not None:
        onehot = torch.nn.functional.one_hot(y)
        y_o = torch.zeros((m, onehot.shape[1]), dtype=onehot.dtype).index_add_(0, inv, onehot).argmax(dim=-1)
    return pos_o, x_o, y_o, inv


def budget_counts(num_nodes: int, minimum: int, maximum: int) -> int:
    """Nodes a tile has after MinimumNumNodes(minimum) then MaximumNumNodes(maximum)."""
    n = num_nodes
    if 0 < n < minimum:
        n = minimum
    if n > maximum:
        n = maximum
    return n


def minimum_num_nodes_choice(num_nodes: int, num: int, generator: Optional[torch.Generator] = None) -> Tensor:
    """transforms.py:66-84: ceil(num / n) independent permutations, concatenated, cut at num (identity if n >= num)."""
    if num_nodes >= num:
        return torch.arange(num_nodes)
    reps = math.ceil(num / num_nodes)
    return torch.cat([torch.randperm(num_nodes, generator=generator) for _ in range(reps)])[:num]


def maximum_num_nodes_choice(num_nodes: int, num: int, generator: Optional[torch.Generator] = None) -> Tensor:
    """transforms.py:48-63: head of one random permutation (identity if n <= num)."""
    if num_nodes <= num:
        return torch.arange(num_nodes)
    return torch.randperm(num_nodes, generator=generator)[:num]


def center(pos: Tensor) -> Tensor:
    """torch_geometric.transforms.Center: subtract the mean position (all three axes)."""
    return pos - pos.mean(dim=-2, keepdim=True)


def nullify_lowest_z(pos: Tensor) -> Tensor:
    out = pos.clone()
    out[:, 2] = out[:, 2] - out[:, 2].min()
    return out


def normalize_pos(pos: Tensor, subtile_width: float = 50) -> Tensor:
    return pos * (1 / (subtile_width / 2))


def standardize_channel(channel: Tensor, clamp_sigma: int = 3) -> Tensor:
    """transforms.py:128-138 — note the clamp bound is clamp_sigma * std of the ORIGINAL channel, applied to the
    standardised values (kept as the reference does it)."""
    mean = channel.mean()
    std = channel.std() + 10 ** -6
    if torch.isnan(std):
        std = 1.0
    lim = clamp_sigma * std
    return torch.clamp((channel - mean) / std, min=-lim, max=lim)


def standardize_rgb_and_intensity(x: Tensor, intensity_col: int, rgb_col: int) -> Tensor:
    out = x.clone()
    out[:, intensity_col] = standardize_channel(torch.log(out[:, intensity_col] + 1))
    out[:, rgb_col] = standardize_channel(out[:, rgb_col])
    return out


def prepare_tiles(pos: Tensor, x: Tensor, y: Optional[Tensor], ptr: Sequence[int], size: float = 0.25,
                  subtile_width: float = 50, intensity_col: int = 0, rgb_col: int = 7
                  ) -> Tuple[Tensor, Tensor, Optional[Tensor], list]:
    """GridSampling -> Center -> NullifyLowestZ -> NormalizePos -> StandardizeRGBAndIntensity per tile (the
    deterministic part of the chain; the node-budget transforms are random and checked by their properties)."""
    P, X, Y, out_ptr = [], [], [], [0]
    for b in range(len(ptr) - 1):
        s, e = int(ptr[b]), int(ptr[b + 1])
        p, xx, yy, _ = grid_sampling(pos[s:e], x[s:e], None if y is None else y[s:e], size)
        if p.shape[0]:
            p = normalize_pos(nullify_lowest_z(center(p)), subtile_width)
            xx = standardize_rgb_and_intensity(xx, intensity_col, rgb_col)
        P.append(p), X.append(xx), Y.append(yy)
        out_ptr.append(out_ptr[-1] + p.shape[0])
    return torch.cat(P), torch.cat(X), (None if y is None else torch.cat(Y)), out_ptr


# --------------------------------------------------------------------------------------
# tiling of a whole cloud into square samples (myria3d/pctl/dataset/utils.py:29-39, 126-158)
# --------------------------------------------------------------------------------------
def get_mosaic_of_centers(tile_width, subtile_width, subtile_overlap=0):
    """utils.py:29-39."""
    if subtile_overlap < 0:
        raise ValueError("datamodule.subtile_overlap must be positive.")
    xy_range = np.arange(subtile_width / 2, tile_width + (subtile_width / 2) - subtile_overlap,
                         step=subtile_width - subtile_overlap)
    return [np.array([x, y]) for x in xy_range for y in xy_range]


def split_cloud_into_samples(pos, tile_width, subtile_width, subtile_overlap=0):
    """utils.py:139-158 without the pdal read: the SAME third-party call as the reference (scipy's cKDTree is
    installed here, so this part of the oracle is pinned to the reference's own arithmetic).  ``pos``: float32
    ``[N, 3]`` numpy array.  Yields ``(sample number in the mosaic, sample_idx)`` for the non-empty samples; the order
    inside ``sample_idx`` is the tree's traversal order, as in the reference."""
    from scipy.spatial import cKDTree

    pos = np.asarray(pos, dtype=np.float32)
    kd_tree = cKDTree(pos[:, :2] - pos[:, :2].min(axis=0))
    for s, center in enumerate(get_mosaic_of_centers(tile_width, subtile_width, subtile_overlap=subtile_overlap)):
        radius = subtile_width // 2  # Square receptive field.
        sample_idx = np.array(kd_tree.query_ball_point(center, r=radius, p=np.inf))
        if not len(sample_idx):
            continue
        yield s, sample_idx
