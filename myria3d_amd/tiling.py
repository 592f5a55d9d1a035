"""Tiling of a whole Lidar cloud into square samples on the device.

Mirror of the selection part of ``/root/reference/myria3d/pctl/dataset/utils.py``:

* ``get_mosaic_of_centers(tile_width, subtile_width, subtile_overlap=0)`` (utils.py:29-39) — same values (numpy
  ``arange``), same order (x-major), same ``ValueError`` for a negative overlap;
* ``split_cloud_into_samples(...)`` (utils.py:126-158) — the reference reads the LAS with pdal, builds a ``cKDTree`` on
  ``pos[:, :2] - pos[:, :2].min(0)`` and runs one ``query_ball_point(center, r=subtile_width // 2, p=inf)`` per centre
  on the CPU, skipping empty samples.  Here the caller hands over the xyz tensor it has read (LAS / HDF5 I/O stays with
  pdal / h5py: storage, out of scope) and ALL samples come out of ``m3d_tile_select`` (``csrc/voxel.hip``) as one CSR
  pair; the generator yields ``sample_idx`` per non-empty sample in the reference's order.  Inside a sample the indices
  are ascending (the tree returns them in traversal order; the set is the contract).
"""
from __future__ import annotations

from numbers import Number
from typing import Iterator, List, Tuple

import numpy as np
import torch
from torch import Tensor

from ._lib import call, lib


def get_mosaic_of_centers(tile_width: Number, subtile_width: Number, subtile_overlap: Number = 0) -> List[np.ndarray]:
    """utils.py:29-39, verbatim semantics."""
    if subtile_overlap < 0:
        raise ValueError("datamodule.subtile_overlap must be positive.")
    xy_range = np.arange(subtile_width / 2, tile_width + (subtile_width / 2) - subtile_overlap,
                         step=subtile_width - subtile_overlap)
    return [np.array([x, y]) for x in xy_range for y in xy_range]


def tile_select(pos: Tensor, tile_width: Number, subtile_width: Number,
                subtile_overlap: Number = 0) -> Tuple[Tensor, Tensor, np.ndarray]:
    """All samples of one cloud: ``(sample_ptr int64 [S + 1], idx int32 [total], centers float64 [S, 2])``; sample
    ``s`` holds points ``idx[sample_ptr[s]:sample_ptr[s + 1]]`` (ascending), ``centers[s]`` is its mosaic centre."""
    if not pos.is_cuda:
        raise RuntimeError("myria3d_amd.tiling runs on the HIP device only (no CPU fallback)")
    if subtile_overlap < 0:
        raise ValueError("datamodule.subtile_overlap must be positive.")
    pos = pos.to(torch.float32)
    if pos.stride(1) != 1:
        pos = pos.contiguous()
    xy_range = np.arange(subtile_width / 2, tile_width + (subtile_width / 2) - subtile_overlap,
                         step=subtile_width - subtile_overlap).astype(np.float64)
    nc = int(xy_range.shape[0])
    S = nc * nc
    centers = np.stack(np.meshgrid(xy_range, xy_range, indexing="ij"), axis=-1).reshape(S, 2)
    dev = pos.device
    sample_ptr = torch.zeros(S + 1, dtype=torch.int64, device=dev)
    if nc == 0 or pos.shape[0] == 0:
        return sample_ptr, torch.empty(0, dtype=torch.int32, device=dev), centers
    radius = float(subtile_width // 2)  # "Square receptive field" (utils.py:150): floor division, like the reference
    step = float(subtile_width - subtile_overlap)
    cdev = torch.from_numpy(xy_range).to(dev)
    n = pos.shape[0]
    ws = torch.empty(lib().m3d_tile_select_workspace_bytes(n, nc), dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    args = (pos.data_ptr(), pos.stride(0), n, cdev.data_ptr(), nc, radius, float(xy_range[0]), step, ws.data_ptr())
    call("m3d_tile_select", *args, 1, sample_ptr.data_ptr(), None, st)
    total = int(sample_ptr[-1].item())  # one host read per cloud: sizes the index list
    if total < 0:
        raise NotImplementedError("subtile_overlap too large: a point belongs to more than 64 samples")
    idx = torch.empty(total, dtype=torch.int32, device=dev)
    if total:
        call("m3d_tile_select", *args, 0, sample_ptr.data_ptr(), idx.data_ptr(), st)
    return sample_ptr, idx, centers


def split_cloud_into_samples(pos: Tensor, tile_width: Number, subtile_width: Number,
                             subtile_overlap: Number = 0) -> Iterator[Tensor]:
    """Generator over the non-empty samples in the reference's order (utils.py:147-158): yields ``sample_idx`` (int64
    device tensor); the caller gathers its own point attributes with it (the reference yields
    ``(sample_idx, points[sample_idx])`` for a pdal structured array)."""
    sample_ptr, idx, _ = tile_select(pos, tile_width, subtile_width, subtile_overlap)
    bounds = sample_ptr.tolist()
    idx = idx.to(torch.int64)
    for s in range(len(bounds) - 1):
        if bounds[s + 1] == bounds[s]:
            continue  # no points in this receptive field
        yield idx[bounds[s]:bounds[s + 1]]
