"""Training-step pieces that sit right behind the net: the reference's criterion and optimizer, MI355X-native.

* ``cross_entropy`` — ``torch.nn.CrossEntropyLoss(ignore_index=65)`` on the logits
  (``/root/reference/myria3d/models/model.py:118``, ``configs/model/criterion/CrossEntropyLoss.yaml:1-3``).
* ``FusedAdam`` — ``torch.optim.Adam`` (``/root/reference/configs/model/optimizer/Adam.yaml:1-4``) as ONE kernel
  launch over the flat parameter / gradient buffers of a ``HipRandLANet`` that has run ``flatten_parameters()``;
  stock torch runs ~250 small launches for the same update (141 parameter tensors).
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.distributed as dist

from . import ops
from ._lib import call
from .ops import cross_entropy  # noqa: F401  (re-export)


class FusedAdam(torch.optim.Optimizer):
    """Adam over ``net.flat_parameters`` / ``net.flat_grads`` (same update rule and defaults as ``torch.optim.Adam``,
    ``amsgrad=False``).  ``step()`` also clears the gradient buffer (``zero_grad`` is then free), and under
    ``torch.distributed`` all-reduces the flat gradient first when ``all_reduce=True`` — one RCCL collective on
    the 4.45 MB bucket, no packing copies."""

    def __init__(self, net, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0,
                 all_reduce: bool = False, overlap_wgrad: bool = True):
        if net.flat_parameters is None:
            net.flatten_parameters()
        self.net = net
        super().__init__(list(net.parameters()), dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        flat = net.flat_parameters
        self.exp_avg = torch.zeros_like(flat)
        self.exp_avg_sq = torch.zeros_like(flat)
        self.step_count = torch.zeros(1, dtype=torch.float32, device=flat.device)
        self.lr_dev: Optional[torch.Tensor] = None  # optional device-side learning rate (graph-replay safe)
        self.all_reduce = all_reduce
        if overlap_wgrad:  # weight-gradient GEMMs run beside the rest of the backward pass; step() joins them
            net.grad_side = ops.GradSideStream(flat.device)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        net = self.net
        if net.grad_side is not None:
            net.grad_side.join()  # the weight-gradient side stream has finished writing the flat gradient buffer
        if not net._flat_intact():
            net._check_flat()
        flat_p, flat_g = net.flat_parameters, net.flat_grads
        if flat_p.data_ptr() != getattr(self, "_bound_ptr", flat_p.data_ptr()):
            raise RuntimeError("FusedAdam: the net was re-flattened after the optimizer was created")
        self._bound_ptr = flat_p.data_ptr()
        scale = 1.0
        if self.all_reduce and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(flat_g, op=dist.ReduceOp.SUM)
            scale = 1.0 / dist.get_world_size()
        g = self.param_groups[0]
        call("m3d_adam_step", flat_p.data_ptr(), flat_g.data_ptr(), self.exp_avg.data_ptr(),
             self.exp_avg_sq.data_ptr(), self.step_count.data_ptr(),
             None if self.lr_dev is None else self.lr_dev.data_ptr(), float(g["lr"]), float(g["betas"][0]),
             float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]), scale, 1, flat_p.numel(),
             torch.cuda.current_stream().cuda_stream)
        return loss

    def zero_grad(self, set_to_none: bool = False):  # gradients are views of the flat buffer: never detach them
        self.net.flat_grads.zero_()
