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

import os
from typing import Dict, Optional

import torch
from torch import Tensor

from . import ops
from .interpolation import DeviceInterpolator, knn_interpolate, knn_interpolation_table, predict_reduce, scatter_sum
from .tiling import tile_select
from .transforms import grid_sampling, node_budget, node_budget_offsets, normalize_tiles


_SIDE: Dict[torch.device, "torch.cuda.Stream"] = {}


def _side_stream(dev) -> "torch.cuda.Stream":
    """One preparation stream per device for the life of the process."""
    dev = torch.device(dev)
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    st = _SIDE.get(dev)
    if st is None:
        st = _SIDE[dev] = torch.cuda.Stream(device=dev)
    return st


@torch.no_grad()
def predict_cloud(net: torch.nn.Module, pos: Tensor, x: Tensor, *, tile_width: float = 1000, subtile_width: float = 50,
                  subtile_overlap: float = 0, batch_size: int = 50, grid_size: float = 0.25, min_nodes: int = 300,
                  max_nodes: int = 40000, interpolation_k: int = 10, intensity_col: int = 0, rgb_col: int = 7,
                  seed: int = 0, rank: int = 0, world_size: int = 1, process_group=None,
                  decimation_idx_fn=None, lookahead: Optional[bool] = None) -> Dict[str, Tensor]:
    """``pos [N, 3]`` (raw coordinates, as read from the LAS), ``x [N, F]`` (the raw feature matrix) on the device.
    Returns ``probas [M, C]``, ``preds [M]``, ``entropy [M]`` and ``idx_in_full_cloud [M]`` for the ``M`` stored predictions
    (every point of every non-empty sample, in sample order: ``interpolation.py:142-164``), plus ``logits_full [N, C]`` (the
    merged accumulator).  ``batch_size`` samples per forward (``configs/experiment/predict.yaml:21-23``: 50).
    ``decimation_idx_fn(ptr_host_list) -> per-level index lists``: parity runs inject the oracle's random decimation draw
    (the net draws its own otherwise, as the reference's ``torch.randperm`` does).
    ``lookahead`` (round 6): everything that depends on POSITIONS only — the interpolation's k-NN table and the net's own
    (``HipRandLANet.prefetch_geometry``: grids, K-NN tables, decimation draw, decoder 1-NN tables) — is computed for batch b + 1
    on side streams while the main stream runs batch b's feature kernels; the same kernels and bit-identical results either way.
    OFF by default (``None``: the environment's ``M3D_PREDICT_LOOKAHEAD``, "0"): measured on the 10 M-point cloud of
    ``bench.py`` it gains nothing at 50 samples per batch (48.8 vs 48.9 ms: the chip is full of kernels either way, the
    chain is bound by their sum), 5 % at 25 (53.7 -> 50.9 ms), and it COSTS 6-8 ms per cloud for the first clouds of a
    process (``bench.py``'s protocol — one warm-up call, two timed ones — measures 56.8 instead of 49.2 ms): the tables
    cross streams, their blocks are not reusable until the consumer's events complete, and the allocator's pool grows by
    1.4 GiB of ``hipMalloc`` calls before it is steady (``profiles/r06y_*``, ``r06zg_*``, ``r06zk_*``)."""
    if not pos.is_cuda:
        raise RuntimeError("myria3d_amd.predict_cloud runs on the HIP device only (no CPU fallback)")
    if lookahead is None:
        lookahead = os.environ.get("M3D_PREDICT_LOOKAHEAD", "0") != "0"
    dev = pos.device
    net.eval()
    pos = pos.to(torch.float32).contiguous()
    x = x.to(dev, torch.float32).contiguous()
    n_full = pos.shape[0]
    num_classes = getattr(net, "num_classes", None)
    sample_ptr, idx, _ = tile_select(pos, tile_width, subtile_width, subtile_overlap)
    bounds = sample_ptr.tolist()
    samples = [s for s in range(len(bounds) - 1) if bounds[s + 1] > bounds[s]]  # empty samples are skipped (utils.py:153)
    samples = samples[rank::world_size] if world_size > 1 else samples
    batches = [samples[b0:b0 + batch_size] for b0 in range(0, len(samples), batch_size)]
    # CSR offsets of every batch's original points, built on the host from the one read of sample_ptr and uploaded ONCE
    # (round 4: one blocking torch.tensor(...).to(dev) per batch)
    offs, flat = [], []
    for chunk in batches:
        offs.append(len(flat))
        acc = 0
        flat.append(0)
        for sidx in chunk:
            acc += bounds[sidx + 1] - bounds[sidx]
            flat.append(acc)
    ptr_all = torch.tensor(flat if flat else [0], dtype=torch.int64).pin_memory().to(dev, non_blocking=True)
    make_plan = getattr(net, "plan_from_host_sizes", None)  # HipRandLANet / HipPointNet2: level plan from host-side tile sizes

    main = torch.cuda.current_stream()
    side = _side_stream(dev)
    side.wait_stream(main)
    bg = int(getattr(net, "background_knn_cap_eval", 0)) if lookahead else 0  # launch cap of work that runs beside the forward
    look_net = lookahead and decimation_idx_fn is None and hasattr(net, "background_knn_cap_eval") \
        and hasattr(net, "prefetch_geometry")

    def prepare(b):
        """Everything in front of the net for batch ``b`` — row lists, CopyFullPos, GridSampling, node budget, normalisations —
        on the SIDE stream: its one device read-back (the voxel counts size the outputs) waits for the side stream only, while
        the main stream runs the previous batch's forward and interpolation (round 4 ran the stages one after the other, each
        behind a device synchronisation: 87 ms per 10 M points for 32 ms of net + interpolation)."""
        chunk = batches[b]
        if world_size == 1:  # consecutive non-empty samples of the CSR list: one slice (the empty ones in between hold no rows)
            rows = idx[bounds[chunk[0]]:bounds[chunk[-1] + 1]]
        else:           # samples sharded over ranks: gather the pieces
            rows = torch.cat([idx[bounds[s]:bounds[s + 1]] for s in chunk])
        ptr_full = ptr_all[offs[b]:offs[b] + len(chunk) + 1]
        pos_copy = ops.gather_rows(pos, rows)                                      # CopyFullPos
        x_raw = ops.gather_rows(x, rows)
        p, xx, _, ptr, ptr_host = grid_sampling(pos_copy, x_raw, None, ptr_full, grid_size, return_host_ptr=True)
        p, xx, _, ptr, kept = node_budget(p, xx, None, ptr, minimum=min_nodes, maximum=max_nodes, seed=seed + b * batch_size,
                                          ptr_host=ptr_host)
        ptr_host = node_budget_offsets(ptr_host, min_nodes, max_nodes)  # node_budget's own rule (one source of truth)
        pn, xn = normalize_tiles(p, xx, ptr, center=True, nullify_z=True, subtile_width=subtile_width,
                                 intensity_col=intensity_col, rgb_col=rgb_col)
        plan = make_plan(ptr_host) if (make_plan is not None and decimation_idx_fn is None) else None
        ev = torch.cuda.Event()
        ev.record(side)  # (what the net reads is complete here)
        table = None
        if lookahead:
            table = knn_interpolation_table(p, pos_copy, ptr, ptr_full, interpolation_k, background=bg)
        ev2 = torch.cuda.Event()
        ev2.record(side)
        out = {"rows": rows, "ptr_full": ptr_full, "pos_copy": pos_copy, "p": p, "pn": pn, "xn": xn, "ptr": ptr,
               "ptr_host": ptr_host, "plan": plan, "ready": ev, "table": table, "table_ready": ev2}
        for t in (rows, ptr_full, pos_copy, p, pn, xn, ptr) + (table if table is not None else ()):
            t.record_stream(main)  # allocated on the side stream, consumed on the main one
        return out

    def prefetch(nb, first):
        """The net's own position-only work for the batch ``nb`` on ITS side stream, ordered behind the preparation (and, past
        the first batch, behind the START of the forward just enqueued: the buffer set it rewrites was last read by the
        forward before that one)."""
        if look_net and nb["plan"] is not None:
            net.prefetch_geometry(nb["pn"], nb["ptr"], plan=nb["plan"], train=False,
                                  after="now" if first else "forward_start", after_event=nb["ready"])

    acc = None
    nxt = None
    kept_rows = []  # (sharded samples only: which rows this rank predicted)
    if batches:
        with torch.cuda.stream(side):
            nxt = prepare(0)
        prefetch(nxt, True)
    for b in range(len(batches)):
        cur, nxt = nxt, None
        main.wait_event(cur["ready"])
        if decimation_idx_fn is not None:
            logits = net(cur["xn"], cur["pn"], None, cur["ptr"], decimation_idx=decimation_idx_fn(cur["ptr_host"]))
        elif cur["plan"] is not None:
            logits = net(cur["xn"], cur["pn"], None, cur["ptr"], plan=cur["plan"])
        else:
            logits = net(cur["xn"], cur["pn"], None, cur["ptr"])
        if b + 1 < len(batches) and lookahead:
            # the next batch's preparation and position-only work, enqueued on the side streams once this batch's FORWARD is
            # queued: the host then blocks on the side stream's read-back while the main stream has ~4 ms of kernels in front
            # of it, and the net's tables for batch b + 1 are built beside them
            with torch.cuda.stream(side):
                nxt = prepare(b + 1)
            prefetch(nxt, False)
        if cur["table"] is not None:
            main.wait_event(cur["table_ready"])
        full = knn_interpolate(logits, cur["p"], cur["pos_copy"], ptr_x=cur["ptr"], ptr_y=cur["ptr_full"], k=interpolation_k,
                               table=cur["table"])
        if acc is None:
            acc = torch.zeros((n_full, full.shape[1]), dtype=torch.float32, device=dev)
        # Interpolator.store_predictions + reduce_predicted_logits (interpolation.py:94-121) in one go: the logits of a batch are
        # added into the per-point accumulator as they arrive (points on sample borders are predicted twice and summed)
        scatter_sum(full, cur["rows"], out=acc, dim=0)
        if world_size > 1:
            kept_rows.append(cur["rows"])
        if b + 1 < len(batches) and not lookahead:
            # the next batch's preparation, enqueued on the side stream once ALL of this batch's main-stream work is queued: the
            # host then blocks on the side stream's read-back while the main stream has ~6 ms of kernels in front of it
            with torch.cuda.stream(side):
                nxt = prepare(b + 1)
    main.wait_stream(side)
    if acc is None:  # no non-empty sample (an empty cloud)
        C = int(num_classes) if num_classes else 0
        acc = torch.zeros((n_full, C), dtype=torch.float32, device=dev)
    if world_size > 1:
        # samples of ONE cloud sharded over ranks: the per-point accumulators meet in one all-reduce (interpolation.py:99-121
        # sums overlapping predictions; the sum is over all samples, whoever computed them)
        import torch.distributed as dist

        hit = torch.zeros((n_full, 1), dtype=torch.float32, device=dev)
        if kept_rows:
            rows_all = torch.cat(kept_rows)
            scatter_sum(torch.ones((rows_all.numel(), 1), device=dev), rows_all, out=hit, dim=0)
        dist.all_reduce(acc, group=process_group)
        dist.all_reduce(hit, group=process_group)
        covered = torch.nonzero(hit[:, 0] > 0).reshape(-1)
        probas, preds, entropy = predict_reduce(acc, covered)
        return {"probas": probas, "preds": preds, "entropy": entropy, "idx_in_full_cloud": covered, "logits_full": acc}
    # the stored predictions in sample order = the CSR index list itself (every non-empty sample, in order)
    rows_all = idx
    if rows_all.numel() == 0:
        C = acc.shape[1]
        return {"probas": torch.zeros((0, C), device=dev), "preds": torch.zeros(0, dtype=torch.int64, device=dev),
                "entropy": torch.zeros(0, device=dev), "idx_in_full_cloud": rows_all.to(torch.int64), "logits_full": acc}
    probas, preds, entropy = predict_reduce(acc, rows_all)
    return {"probas": probas, "preds": preds, "entropy": entropy, "idx_in_full_cloud": rows_all, "logits_full": acc}


def itp_reduce(itp: DeviceInterpolator, n_full: int, num_classes: int = 0, device=None) -> Dict[str, Tensor]:
    """``DeviceInterpolator.reduce_predictions`` that also hands back the merged ``[N, C]`` accumulator.  Nothing stored (a
    cloud without a non-empty sample): empty outputs and a zero accumulator."""
    if not itp.logits:
        dev = device if device is not None else torch.device("cuda")
        return {"probas": torch.zeros((0, num_classes), device=dev), "preds": torch.zeros(0, dtype=torch.int64, device=dev),
                "entropy": torch.zeros(0, device=dev), "idx_in_full_cloud": torch.zeros(0, dtype=torch.int32, device=dev),
                "logits_full": torch.zeros((n_full, num_classes), device=dev)}
    logits = torch.cat(itp.logits)
    idx = torch.cat([i.reshape(-1) for i in itp.idx_in_full_cloud_list])
    itp.logits, itp.idx_in_full_cloud_list = [], []
    acc = torch.zeros((n_full, logits.shape[1]), dtype=torch.float32, device=logits.device)
    scatter_sum(logits, idx, out=acc, dim=0)
    probas, preds, entropy = predict_reduce(acc, idx)
    return {"probas": probas, "preds": preds, "entropy": entropy, "idx_in_full_cloud": idx, "logits_full": acc}
