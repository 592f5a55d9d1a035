"""Function-level drop-ins for the second boundary of the hot path:

* ``torch_geometric.nn.knn_interpolate(x, pos_x, pos_y, batch_x, batch_y, k, num_workers)`` as called by
  ``Model.forward`` at test/predict time (``/root/reference/myria3d/models/model.py:88-98``) — the reference moves
  the logits to the CPU for it; here kNN + inverse-squared-distance weighting stay on the MI355X.
* ``torch_scatter.scatter_sum(src, index, out=..., dim=0)`` as called by ``Interpolator.reduce_predicted_logits``
  (``/root/reference/myria3d/models/interpolation.py:116``).
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import Tensor

from . import ops


def _ptr_from_batch(batch: Optional[Tensor], n: int, device) -> Tensor:
    if batch is None:
        return torch.tensor([0, n], dtype=torch.int64, device=device)
    num = int(batch.max().item()) + 1 if batch.numel() else 0
    counts = torch.bincount(batch.to(device), minlength=num)
    return torch.cat([counts.new_zeros(1), counts.cumsum(0)]).to(torch.int64)


def knn_interpolate(x: Tensor, pos_x: Tensor, pos_y: Tensor, batch_x: Optional[Tensor] = None,
                    batch_y: Optional[Tensor] = None, k: int = 3, num_workers: int = 1) -> Tensor:
    """Same signature and semantics as PyG's ``knn_interpolate`` (``batch_*`` must be sorted, as PyG requires).
    ``num_workers`` is accepted and ignored (it only steers torch_cluster's CPU path)."""
    if not x.is_cuda:
        raise RuntimeError("myria3d_amd.knn_interpolate runs on the HIP device only (no CPU fallback)")
    dev = x.device
    with torch.no_grad():
        pos_x = pos_x.to(dev, torch.float32).contiguous()
        pos_y = pos_y.to(dev, torch.float32).contiguous()
        ptr_x = _ptr_from_batch(batch_x, pos_x.shape[0], dev)
        ptr_y = _ptr_from_batch(batch_y, pos_y.shape[0], dev)
        if ptr_y.numel() < ptr_x.numel():  # trailing clouds without queries
            ptr_y = torch.cat([ptr_y, ptr_y[-1:].expand(ptr_x.numel() - ptr_y.numel())])
        elif ptr_x.numel() < ptr_y.numel():
            ptr_x = torch.cat([ptr_x, ptr_x[-1:].expand(ptr_y.numel() - ptr_x.numel())])
        index = ops.KnnIndex(pos_x, ptr_x.contiguous())
        # the queries get a grid of their own: cell-sorted query order keeps every wavefront inside a few grid cells
        # (rows of the result stay in the caller's order)
        qindex = ops.KnnIndex(pos_y, ptr_y.contiguous())
        idx, d2 = index.query(k, qry=qindex, want_d2=True)
        return ops.idw_interpolate(x.to(torch.float32).contiguous(), idx, d2)


def scatter_sum(src: Tensor, index: Tensor, dim: int = 0, out: Optional[Tensor] = None,
                dim_size: Optional[int] = None) -> Tensor:
    """``out[index[i]] += src[i]`` along dim 0 for 2-D fp32 ``src`` (torch_scatter.scatter_sum, dim=0)."""
    if dim != 0 or src.dim() != 2:
        raise NotImplementedError("scatter_sum drop-in covers the reference's use: 2-D src, dim=0")
    if not src.is_cuda:
        raise RuntimeError("myria3d_amd.scatter_sum runs on the HIP device only (no CPU fallback)")
    src = src.to(torch.float32).contiguous()
    idx = index.to(src.device, torch.int32).contiguous()
    if out is not None and out.is_cuda and out.dtype == torch.float32 and out.stride(1) == 1 \
            and out.shape[1] == src.shape[1]:
        # accumulate straight into the caller's buffer (the reference's `out=` use, interpolation.py:116)
        ops.call("m3d_scatter_add_rows", src.data_ptr(), idx.data_ptr(), out.data_ptr(), out.stride(0), src.shape[0],
                 src.shape[1], torch.cuda.current_stream().cuda_stream)
        return out
    n = dim_size if dim_size is not None else (out.shape[0] if out is not None else int(index.max().item()) + 1)
    res = ops.scatter_add_rows(src, idx, n)
    if out is not None:
        out += res.to(out.dtype)
        return out
    return res
