"""Data-parallel helpers: tiles shard over ranks, the only collective is the gradient all-reduce.

Under Lightning the reference gets this from ``strategy: ddp_find_unused_parameters_false``
(``/root/reference/configs/experiment/RandLaNet_base_run_FR-MultiGPU.yaml:9-13``) and ``HipRandLANet`` works under
``torch.nn.parallel.DistributedDataParallel`` unchanged (its gradients come from ordinary autograd nodes).  For the
stand-alone bench / training loops the same thing comes in its MI355X-friendly form from ``FusedAdam(all_reduce=True)``
(``train.py``): ONE flat fp32 bucket (1 113 686 parameters = 4.45 MB) all-reduced over RCCL/xGMI per step — a single
collective sized for the 7 point-to-point links instead of DDP's default 25 MB bucketing logic.  This module holds
the two pieces around it: which tiles a rank owns, and the start-up broadcast.  BatchNorm statistics stay per rank,
as in the reference (no SyncBatchNorm).
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_tiles(num_tiles: int, rank: int, world_size: int) -> range:
    """Contiguous block of tile ids owned by ``rank`` (independent units; no data-path exchange)."""
    per = (num_tiles + world_size - 1) // world_size
    return range(min(num_tiles, rank * per), min(num_tiles, (rank + 1) * per))


def broadcast_module_state(module: torch.nn.Module, src: int = 0) -> None:
    """Make every rank start from rank ``src``'s parameters and buffers (what DDP does at construction)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    flat = getattr(module, "flat_parameters", None)
    tensors = [flat] if flat is not None else [p.data for p in module.parameters()]  # one bucket when flattened
    for t in tensors + [b.data for b in module.buffers()]:
        dist.broadcast(t, src=src)
