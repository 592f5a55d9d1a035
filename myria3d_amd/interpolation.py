"""Function-level drop-ins for the second boundary of the hot path:

* ``torch_geometric.nn.knn_interpolate(x, pos_x, pos_y, batch_x, batch_y, k, num_workers)`` as called by
  ``Model.forward`` at test/predict time (``/root/reference/myria3d/models/model.py:88-98``) — the reference moves
  the logits to the CPU for it; here kNN + inverse-squared-distance weighting stay on the MI355X.
* ``torch_scatter.scatter_sum(src, index, out=..., dim=0)`` as called by ``Interpolator.reduce_predicted_logits``
  (``/root/reference/myria3d/models/interpolation.py:116``).
* ``DeviceInterpolator``: the arithmetic of ``Interpolator`` (``interpolation.py:94-169``: store, scatter-sum merge,
  softmax / argmax / entropy) kept on the device; the LAS reading / writing around it (pdal) stays with the caller.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple, Union

import torch
from torch import Tensor

from . import ops


def _ptr_from_batch(batch: Optional[Tensor], n: int, device) -> Tensor:
    if batch is None:
        return torch.tensor([0, n], dtype=torch.int64, device=device)
    num = int(batch.max().item()) + 1 if batch.numel() else 0
    counts = torch.bincount(batch.to(device), minlength=num)
    return torch.cat([counts.new_zeros(1), counts.cumsum(0)]).to(torch.int64)


def knn_interpolation_table(pos_x: Tensor, pos_y: Tensor, ptr_x: Tensor, ptr_y: Tensor, k: int,
                            background: int = 0) -> Tuple[Tensor, Tensor]:
    """The position-only half of ``knn_interpolate``: ``(idx [Ny, k] int32, d2 [Ny, k] fp32)``, the ``k`` nearest points of
    ``pos_x`` for every point of ``pos_y`` inside the same cloud (-1 / padding where a cloud holds fewer) and the squared
    distances.  ``predict_cloud`` computes it for the NEXT batch beside the current batch's forward (``background``: the
    launch cap of ``KnnIndex.query``)."""
    dev = pos_x.device
    with torch.no_grad():
        pos_x = pos_x.to(dev, torch.float32).contiguous()
        pos_y = pos_y.to(dev, torch.float32).contiguous()
        ptr_x, ptr_y = ptr_x.to(dev, torch.int64), ptr_y.to(dev, torch.int64)
        if ptr_y.numel() < ptr_x.numel():  # trailing clouds without queries
            ptr_y = torch.cat([ptr_y, ptr_y[-1:].expand(ptr_x.numel() - ptr_y.numel())])
        elif ptr_x.numel() < ptr_y.numel():
            ptr_x = torch.cat([ptr_x, ptr_x[-1:].expand(ptr_y.numel() - ptr_x.numel())])
        index = ops.KnnIndex(pos_x, ptr_x.contiguous())
        # the queries get a grid of their own: cell-sorted query order keeps every wavefront inside a few grid cells
        # (rows of the result stay in the caller's order)
        qindex = ops.KnnIndex(pos_y, ptr_y.contiguous())
        return index.query(k, qry=qindex, want_d2=True, background=background)


def knn_interpolate(x: Tensor, pos_x: Tensor, pos_y: Tensor, batch_x: Optional[Tensor] = None,
                    batch_y: Optional[Tensor] = None, k: int = 3, num_workers: int = 1,
                    ptr_x: Optional[Tensor] = None, ptr_y: Optional[Tensor] = None,
                    table: Optional[Tuple[Tensor, Tensor]] = None) -> Tensor:
    """Same signature and semantics as PyG's ``knn_interpolate`` (``batch_*`` must be sorted, as PyG requires).
    ``num_workers`` is accepted and ignored (it only steers torch_cluster's CPU path).  Extensions: ``ptr_x`` / ``ptr_y``
    (device int64 ``[B + 1]`` CSR offsets, what ``Batch.ptr`` holds) instead of the batch vectors — converting a batch vector
    costs a ``bincount`` and a device read-back per call; ``table``: the result of ``knn_interpolation_table`` for these
    positions, computed earlier (the positions are then not looked at)."""
    if not x.is_cuda:
        raise RuntimeError("myria3d_amd.knn_interpolate runs on the HIP device only (no CPU fallback)")
    dev = x.device
    with torch.no_grad():
        if table is None:
            pos_x = pos_x.to(dev, torch.float32)
            pos_y = pos_y.to(dev, torch.float32)
            ptr_x = ptr_x if ptr_x is not None else _ptr_from_batch(batch_x, pos_x.shape[0], dev)
            ptr_y = ptr_y if ptr_y is not None else _ptr_from_batch(batch_y, pos_y.shape[0], dev)
            table = knn_interpolation_table(pos_x, pos_y, ptr_x, ptr_y, k)
        idx, d2 = table
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
                 src.shape[1], 0, torch.cuda.current_stream().cuda_stream)
        return out
    n = dim_size if dim_size is not None else (out.shape[0] if out is not None else int(index.max().item()) + 1)
    res = ops.scatter_add_rows(src, idx, n)
    if out is not None:
        out += res.to(out.dtype)
        return out
    return res


def predict_reduce(logits: Tensor, index: Optional[Tensor] = None, want_probas: bool = True, want_preds: bool = True,
                   want_entropy: bool = True) -> Tuple[Optional[Tensor], Optional[Tensor], Optional[Tensor]]:
    """``(Softmax(dim=1)(rows), argmax(rows, dim=1), Categorical(probs=probas).entropy())`` of
    ``rows = logits[index]`` (``index=None``: all rows) in one HIP launch (``interpolation.py:145-164``)."""
    if not logits.is_cuda:
        raise RuntimeError("myria3d_amd.predict_reduce runs on the HIP device only (no CPU fallback)")
    if logits.dim() != 2:
        raise ValueError("logits must be [points, classes]")
    logits = logits.to(torch.float32)
    if logits.stride(1) != 1:
        logits = logits.contiguous()
    dev = logits.device
    idx = None if index is None else index.to(dev, torch.int32).contiguous()
    m, C = (logits.shape[0] if idx is None else idx.shape[0]), logits.shape[1]
    probas = torch.empty((m, C), dtype=torch.float32, device=dev) if want_probas else None
    preds = torch.empty((m,), dtype=torch.int32, device=dev) if want_preds else None
    entropy = torch.empty((m,), dtype=torch.float32, device=dev) if want_entropy else None
    ops.call("m3d_predict_reduce", logits.data_ptr(), logits.stride(0), None if idx is None else idx.data_ptr(), m, C,
             None if probas is None else probas.data_ptr(), C, None if preds is None else preds.data_ptr(),
             None if entropy is None else entropy.data_ptr(), torch.cuda.current_stream().cuda_stream)
    return probas, (None if preds is None else preds.to(torch.int64)), entropy


class DeviceInterpolator:
    """Device-resident mirror of the arithmetic of ``myria3d.models.interpolation.Interpolator``
    (``interpolation.py:94-169``): same method names and meaning for ``store_predictions`` and
    ``reduce_predicted_logits``; ``reduce_predictions`` returns what ``reduce_predictions_and_save`` writes into
    the LAS dimensions (probabilities, predicted class, entropy) instead of writing a file."""

    def __init__(self, reverse_mapper: Optional[Dict[int, int]] = None):
        self.reverse_mapper = reverse_mapper  # class index -> LAS classification code (interpolation.py:52-56)
        self.logits: List[Tensor] = []
        self.idx_in_full_cloud_list: List[Tensor] = []

    def store_predictions(self, logits: Tensor, idx_in_original_cloud: Union[Tensor, Sequence]) -> None:
        """Keep the (already interpolated) logits of one batch and where their points sit in the full cloud."""
        if not logits.is_cuda:
            raise RuntimeError("DeviceInterpolator keeps predictions on the HIP device (no CPU fallback)")
        self.logits.append(logits)
        if isinstance(idx_in_original_cloud, Tensor):
            self.idx_in_full_cloud_list.append(idx_in_original_cloud.to(logits.device))
        else:  # the reference collates a list of numpy arrays, one per tile (interpolation.py:96)
            self.idx_in_full_cloud_list += [torch.as_tensor(a).to(logits.device) for a in idx_in_original_cloud]

    @torch.no_grad()
    def reduce_predicted_logits(self, nb_points: int) -> Tuple[Tensor, Tensor]:
        """Sum the logits of points predicted more than once (overlapping tiles) and return them per stored
        prediction, in stored order, with the index vector (``interpolation.py:98-121``)."""
        logits = torch.cat(self.logits)
        idx = torch.cat([i.reshape(-1) for i in self.idx_in_full_cloud_list])
        self.logits, self.idx_in_full_cloud_list = [], []
        reduced = torch.zeros((nb_points, logits.shape[1]), dtype=torch.float32, device=logits.device)
        scatter_sum(logits, idx, out=reduced, dim=0)
        return ops.gather_rows(reduced, idx.to(torch.int32)), idx

    @torch.no_grad()
    def reduce_predictions(self, nb_points: int) -> Dict[str, Tensor]:
        """``probas`` [M, C], ``preds`` [M] (mapped through ``reverse_mapper`` if given), ``entropy`` [M] and
        ``idx_in_full_cloud`` [M] for the M stored predictions (``interpolation.py:142-164``)."""
        logits = torch.cat(self.logits)
        idx = torch.cat([i.reshape(-1) for i in self.idx_in_full_cloud_list])
        self.logits, self.idx_in_full_cloud_list = [], []
        reduced = torch.zeros((nb_points, logits.shape[1]), dtype=torch.float32, device=logits.device)
        scatter_sum(logits, idx, out=reduced, dim=0)
        probas, preds, entropy = predict_reduce(reduced, idx)
        if self.reverse_mapper is not None:
            lut = torch.zeros(max(self.reverse_mapper) + 1, dtype=torch.int64, device=preds.device)
            for k, v in self.reverse_mapper.items():
                lut[k] = v
            preds = lut[preds]
        return {"probas": probas, "preds": preds, "entropy": entropy, "idx_in_full_cloud": idx}
