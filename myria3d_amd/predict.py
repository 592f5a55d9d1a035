"""The reference's inference pipeline (``/root/reference/myria3d/predict.py:49-66``) from a cloud in HBM to per-point
predictions, every stage on the MI355X (BASELINE.json ``configs[2]``: "Inference over a synthetic 1 km2 LAS ... tiled 50 m,
predict.py path").

What the reference chains for one LAS file, and where each stage lives here:

==========================================================================  ==========================================
``split_cloud_into_samples`` (``pctl/dataset/utils.py:126-158``)             ``tiling.tile_select`` (``m3d_tile_select``)
``CopyFullPos``, ``GridSampling(0.25)``, ``Minimum/MaximumNumNodes``,        ``transforms.grid_sampling`` / ``node_budget``
``CopySampledPos``, ``Center`` (``points_budget.yaml`` predict list)         / ``normalize_tiles`` on a whole batch of
``NullifyLowestZ``, ``NormalizePos``, ``StandardizeRGBAndIntensity``         samples
``Model.predict_step`` -> ``Model.forward`` (``models/model.py:67-103``):    ``HipRandLANet`` + ``knn_interpolate``
net, then ``knn_interpolate(k=10)`` onto the sample's original points        (``m3d_knn_*``, ``m3d_idw_interpolate_fwd``)
``Interpolator.store_predictions`` / ``reduce_predicted_logits`` /           ``DeviceInterpolator`` (``m3d_scatter_add_rows``,
softmax, argmax, entropy (``models/interpolation.py:94-169``)                ``m3d_predict_reduce``)
==========================================================================  ==========================================

Reading and writing the LAS (pdal) and building the feature matrix from its dimensions stay with the caller: storage, out
of scope.  MULTI-GPU RULE: ranks take whole clouds (``run.py:78-80`` globs the LAS files of a directory), so the
per-cloud ``scatter_sum`` merge never crosses ranks; ``predict_cloud(..., rank, world_size)`` may instead shard the SAMPLES
of one cloud, in which case the ``[N, C]`` logit accumulators are summed over the ranks (one all-reduce) before the softmax.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
from torch import Tensor

from . import ops
from .interpolation import DeviceInterpolator, knn_interpolate, predict_reduce, scatter_sum
from .tiling import tile_select
from .transforms import grid_sampling, node_budget, normalize_tiles


@torch.no_grad()
def predict_cloud(net: torch.nn.Module, pos: Tensor, x: Tensor, *, tile_width: float = 1000, subtile_width: float = 50,
                  subtile_overlap: float = 0, batch_size: int = 50, grid_size: float = 0.25, min_nodes: int = 300,
                  max_nodes: int = 40000, interpolation_k: int = 10, intensity_col: int = 0, rgb_col: int = 7,
                  seed: int = 0, rank: int = 0, world_size: int = 1, process_group=None,
                  decimation_idx_fn=None) -> Dict[str, Tensor]:
    """``pos [N, 3]`` (raw coordinates, as read from the LAS), ``x [N, F]`` (the raw feature matrix) on the device.
    Returns ``probas [M, C]``, ``preds [M]``, ``entropy [M]`` and ``idx_in_full_cloud [M]`` for the ``M`` stored predictions
    (every point of every non-empty sample, in sample order: ``interpolation.py:142-164``), plus ``logits_full [N, C]`` (the
    merged accumulator).  ``batch_size`` samples per forward (``configs/experiment/predict.yaml:21-23``: 50).
    ``decimation_idx_fn(ptr_host_list) -> per-level index lists``: parity runs inject the oracle's random decimation draw
    (the net draws its own otherwise, as the reference's ``torch.randperm`` does)."""
    if not pos.is_cuda:
        raise RuntimeError("myria3d_amd.predict_cloud runs on the HIP device only (no CPU fallback)")
    dev = pos.device
    net.eval()
    pos = pos.to(torch.float32).contiguous()
    x = x.to(dev, torch.float32).contiguous()
    n_full = pos.shape[0]
    sample_ptr, idx, _ = tile_select(pos, tile_width, subtile_width, subtile_overlap)
    bounds = sample_ptr.tolist()
    samples = [s for s in range(len(bounds) - 1) if bounds[s + 1] > bounds[s]]  # empty samples are skipped (utils.py:153)
    samples = samples[rank::world_size] if world_size > 1 else samples
    itp = DeviceInterpolator()
    for b0 in range(0, len(samples), batch_size):
        chunk = samples[b0:b0 + batch_size]
        rows = torch.cat([idx[bounds[s]:bounds[s + 1]] for s in chunk])           # idx_in_original_cloud of the batch
        sizes = torch.tensor([bounds[s + 1] - bounds[s] for s in chunk], dtype=torch.int64)
        ptr_full = torch.cat([sizes.new_zeros(1), sizes.cumsum(0)]).to(dev)
        pos_copy = ops.gather_rows(pos, rows)                                      # CopyFullPos
        x_raw = ops.gather_rows(x, rows)
        p, xx, _, ptr = grid_sampling(pos_copy, x_raw, None, ptr_full, grid_size)
        p, xx, _, ptr, _ = node_budget(p, xx, None, ptr, minimum=min_nodes, maximum=max_nodes, seed=seed + b0)
        pos_sampled_copy = p                                                       # CopySampledPos
        pn, xn = normalize_tiles(p, xx, ptr, center=True, nullify_z=True, subtile_width=subtile_width,
                                 intensity_col=intensity_col, rgb_col=rgb_col)
        if decimation_idx_fn is not None:
            logits = net(xn, pn, None, ptr, decimation_idx=decimation_idx_fn(ptr.tolist()))
        else:
            logits = net(xn, pn, None, ptr)
        cnt = (ptr[1:] - ptr[:-1])
        batch_x = torch.repeat_interleave(torch.arange(len(chunk), device=dev), cnt)
        batch_y = torch.repeat_interleave(torch.arange(len(chunk), device=dev), sizes.to(dev))
        full = knn_interpolate(logits, pos_sampled_copy, pos_copy, batch_x=batch_x, batch_y=batch_y, k=interpolation_k)
        itp.store_predictions(full, rows)
    if world_size > 1:
        # samples of ONE cloud sharded over ranks: the per-point accumulators meet in one all-reduce (interpolation.py:99-121
        # sums overlapping predictions; the sum is over all samples, whoever computed them)
        import torch.distributed as dist

        logits = torch.cat(itp.logits) if itp.logits else torch.zeros((0, net.num_classes), device=dev)
        rows_all = torch.cat([i.reshape(-1) for i in itp.idx_in_full_cloud_list]) if itp.logits else torch.zeros(0, dtype=torch.int32, device=dev)
        acc = torch.zeros((n_full, logits.shape[1]), dtype=torch.float32, device=dev)
        hit = torch.zeros((n_full, 1), dtype=torch.float32, device=dev)
        if logits.shape[0]:
            scatter_sum(logits, rows_all, out=acc, dim=0)
            scatter_sum(torch.ones((rows_all.numel(), 1), device=dev), rows_all, out=hit, dim=0)
        dist.all_reduce(acc, group=process_group)
        dist.all_reduce(hit, group=process_group)
        covered = torch.nonzero(hit[:, 0] > 0).reshape(-1)
        probas, preds, entropy = predict_reduce(acc, covered)
        return {"probas": probas, "preds": preds, "entropy": entropy, "idx_in_full_cloud": covered, "logits_full": acc}
    out = itp_reduce(itp, n_full)
    return out


def itp_reduce(itp: DeviceInterpolator, n_full: int) -> Dict[str, Tensor]:
    """``DeviceInterpolator.reduce_predictions`` that also hands back the merged ``[N, C]`` accumulator."""
    logits = torch.cat(itp.logits)
    idx = torch.cat([i.reshape(-1) for i in itp.idx_in_full_cloud_list])
    itp.logits, itp.idx_in_full_cloud_list = [], []
    acc = torch.zeros((n_full, logits.shape[1]), dtype=torch.float32, device=logits.device)
    scatter_sum(logits, idx, out=acc, dim=0)
    probas, preds, entropy = predict_reduce(acc, idx)
    return {"probas": probas, "preds": preds, "entropy": entropy, "idx_in_full_cloud": idx, "logits_full": acc}
