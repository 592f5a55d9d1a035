"""``Model.forward`` of the reference (``/root/reference/myria3d/models/model.py:67-103``) around ``HipRandLANet``,
without Lightning: the same branch structure on a PyG-``Batch``-shaped object, with the evaluation-time interpolation
kept on the device.

The reference's LightningModule does, per batch,

    logits = self.model(batch.x, batch.pos, batch.batch, batch.ptr)                                (model.py:79)
    if self.training or "copies" not in batch:  return batch.y, logits                             (model.py:80-84)
    logits = knn_interpolate(logits.cpu(), batch.copies["pos_sampled_copy"].cpu(), batch.copies["pos_copy"].cpu(),
                             batch_x=batch.batch.cpu(), batch_y=<enumeration of idx_in_original_cloud>.cpu(),
                             k=interpolation_k, num_workers=num_workers)                            (model.py:88-98)
    targets = batch.copies.get("transformed_y_copy")                                               (model.py:99-102)
    return targets, logits

``forward_like_model`` is that function for any object with those attributes (a real ``torch_geometric.data.Batch``
when PyG is installed, ``SimpleBatch`` below otherwise); a Myria3D maintainer gets the same effect by swapping the
``knn_interpolate`` import of model.py and dropping the ``.cpu()`` calls.
"""
from __future__ import annotations

from typing import Any, Dict, List, Optional, Sequence, Tuple

import torch
from torch import Tensor

from .interpolation import knn_interpolate


def get_batch_tensor_by_enumeration(pos_x: Sequence) -> Tensor:
    """``Model._get_batch_tensor_by_enumeration`` (model.py:194-198): ``[0,0,...,1,1,...,B-1]`` from a list of
    per-sample arrays (``batch.idx_in_original_cloud`` is a Python list of numpy arrays after collation)."""
    return torch.cat([torch.full((len(sample_pos),), i) for i, sample_pos in enumerate(pos_x)])


class SimpleBatch:
    """The attributes of a collated PyG ``Batch`` that ``Model.forward`` touches (``x, pos, batch, ptr, y``, the
    list-valued ``idx_in_original_cloud`` and the dict-valued ``copies``), with PyG's ``"key" in batch`` protocol."""

    def __init__(self, **kwargs: Any):
        for k, v in kwargs.items():
            setattr(self, k, v)

    def __contains__(self, key: str) -> bool:
        return getattr(self, key, None) is not None

    def to(self, device) -> "SimpleBatch":
        def mv(v):
            if isinstance(v, Tensor):
                return v.to(device)
            if isinstance(v, dict):
                return {k: mv(x) for k, x in v.items()}
            return v

        return SimpleBatch(**{k: mv(v) for k, v in self.__dict__.items()})


def collate_tiles(tiles: Sequence[Dict[str, Any]]) -> SimpleBatch:
    """Concatenates per-tile dicts the way PyG's ``Collater`` does for Myria3D's ``Data`` objects
    (``/root/reference/myria3d/pctl/dataloader/dataloader.py:19-32``): node tensors along dim 0 with ``batch`` / ``ptr``,
    ``copies`` key-wise, ``idx_in_original_cloud`` as a list."""
    sizes = [t["pos"].shape[0] for t in tiles]
    ptr = torch.tensor([0] + list(torch.tensor(sizes).cumsum(0)), dtype=torch.int64)
    out: Dict[str, Any] = {
        "pos": torch.cat([t["pos"] for t in tiles]), "ptr": ptr,
        "batch": torch.repeat_interleave(torch.arange(len(tiles)), torch.tensor(sizes)),
        "x": torch.cat([t["x"] for t in tiles]) if tiles[0].get("x") is not None else None,
        "y": torch.cat([t["y"] for t in tiles]) if tiles[0].get("y") is not None else None,
    }
    if tiles[0].get("idx_in_original_cloud") is not None:
        out["idx_in_original_cloud"] = [t["idx_in_original_cloud"] for t in tiles]
    if tiles[0].get("copies") is not None:
        keys = tiles[0]["copies"].keys()
        out["copies"] = {k: torch.cat([t["copies"][k] for t in tiles]) for k in keys}
    return SimpleBatch(**out)


def forward_like_model(net: torch.nn.Module, batch: Any, interpolation_k: int = 10, num_workers: int = 4,
                       training: Optional[bool] = None) -> Tuple[Optional[Tensor], Tensor]:
    """``(targets, logits) = Model.forward(batch)`` with ``net`` in the place of ``self.model``.  In evaluation with
    ``copies`` in the batch the logits come back on the FULL tile (one row per original point), interpolated on the
    device from the sub-sampled points; otherwise ``(batch.y, logits)`` on the sub-sampled points."""
    training = net.training if training is None else training
    logits = net(batch.x, batch.pos, batch.batch, batch.ptr)
    if training or "copies" not in batch:
        return getattr(batch, "y", None), logits
    copies = batch.copies
    dev = logits.device
    batch_y = get_batch_tensor_by_enumeration(batch.idx_in_original_cloud).to(dev)
    logits = knn_interpolate(logits, copies["pos_sampled_copy"].to(dev), copies["pos_copy"].to(dev),
                             batch_x=batch.batch.to(dev), batch_y=batch_y, k=interpolation_k, num_workers=num_workers)
    targets = None  # no targets in inference mode
    if "transformed_y_copy" in copies:
        targets = copies["transformed_y_copy"].to(dev)
    return targets, logits
