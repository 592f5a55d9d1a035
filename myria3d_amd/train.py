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
    the 4.45 MB bucket, no packing copies.

    ``state_dict()`` / ``load_state_dict()`` speak ``torch.optim.Adam``'s format (per-parameter ``step`` /
    ``exp_avg`` / ``exp_avg_sq``), so Lightning checkpoints (``ckpt_path`` resume, ``/root/reference/myria3d/train.py:145``)
    keep the Adam moments and the bias-correction step, and a run started with the reference's ``torch.optim.Adam``
    can be resumed here (and the other way round).  One parameter group, every parameter trainable: the single
    launch updates the whole flat buffer."""

    def __init__(self, net, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0,
                 all_reduce: bool = False, overlap_wgrad: bool = True, force_collective: bool = False):
        if net.flat_parameters is None:
            net.flatten_parameters()
        self.net = net
        params = list(net.parameters())
        if any(not p.requires_grad for p in params):
            raise ValueError("FusedAdam updates the whole flat parameter buffer: frozen parameters "
                             "(requires_grad=False) are not supported — use torch.optim.Adam for partial fine-tuning")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        flat = net.flat_parameters
        self.exp_avg = torch.zeros_like(flat)
        self.exp_avg_sq = torch.zeros_like(flat)
        # device-side state of m3d_adam_step: [0] the step counter (fp32), then the arrival tickets of the update's workgroups
        self._adam_state = torch.zeros(66, dtype=torch.float32, device=flat.device)  # (M3D_ADAM_STATE_WORDS)
        self.step_count = self._adam_state[:1]
        self.lr_dev: Optional[torch.Tensor] = None  # optional device-side learning rate (graph-replay safe)
        self.all_reduce = all_reduce
        # run the gradient all-reduce even on a 1-rank process group (exercises RCCL and its interplay with hipGraph
        # capture on a one-GPU box: tests, ``bench.py --force-collective``)
        self.force_collective = force_collective
        if overlap_wgrad and flat.is_cuda:  # weight-gradient GEMMs run beside the rest of the backward pass
            net.grad_side = ops.GradSideStream(flat.device)

    def add_param_group(self, param_group):
        if len(self.param_groups) >= 1:
            raise ValueError("FusedAdam supports exactly one parameter group (one launch over the flat buffer)")
        super().add_param_group(param_group)

    # ------------------------------------------------------------------------------------------
    def _slices(self):
        off = 0
        for p in self.net.parameters():
            n = p.numel()
            yield p, off, n
            off += (n + 3) // 4 * 4

    def state_dict(self):
        """``torch.optim.Adam``'s layout: ``state[i] = {step, exp_avg, exp_avg_sq}`` per parameter index."""
        step = self.step_count.detach().reshape(()).clone().cpu()
        state = {}
        for i, (p, off, n) in enumerate(self._slices()):
            state[i] = {"step": step.clone(),
                        "exp_avg": self.exp_avg[off:off + n].view(p.shape).clone(),
                        "exp_avg_sq": self.exp_avg_sq[off:off + n].view(p.shape).clone()}
        groups = [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups]
        groups[0]["params"] = list(range(len(state)))
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, state_dict):
        groups = state_dict["param_groups"]
        if len(groups) != 1:
            raise ValueError("FusedAdam supports exactly one parameter group")
        slices = list(self._slices())
        if len(groups[0]["params"]) != len(slices):
            raise ValueError("loaded state dict has a different number of parameters")
        if groups[0].get("amsgrad", False) or groups[0].get("maximize", False):
            raise ValueError("FusedAdam implements amsgrad=False, maximize=False only")
        state = state_dict["state"]
        steps = set()
        with torch.no_grad():
            self.exp_avg.zero_(), self.exp_avg_sq.zero_()
            for key, (p, off, n) in zip(groups[0]["params"], slices):
                st = state.get(key, state.get(str(key)))
                if st is None:  # torch.optim.Adam has no entry for a parameter that was never stepped
                    continue
                if tuple(st["exp_avg"].shape) != tuple(p.shape):
                    raise ValueError(f"optimizer state of parameter {key} has shape {tuple(st['exp_avg'].shape)}, "
                                     f"expected {tuple(p.shape)}")
                self.exp_avg[off:off + n].copy_(st["exp_avg"].reshape(-1))
                self.exp_avg_sq[off:off + n].copy_(st["exp_avg_sq"].reshape(-1))
                steps.add(float(st["step"]))
            if len(steps) > 1:
                raise ValueError(f"FusedAdam keeps ONE step counter; the loaded state has {sorted(steps)}")
            self.step_count.fill_(steps.pop() if steps else 0.0)
        for k, v in groups[0].items():
            if k != "params" and k in self.param_groups[0]:
                self.param_groups[0][k] = v

    # ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def reduce_gradients(self) -> float:
        """Joins the weight-gradient side stream and, with ``all_reduce`` under ``torch.distributed``, sums the flat
        gradient bucket over the ranks with ONE collective.  Returns the scale (1 / world size) the update applies."""
        net = self.net
        if net.grad_side is not None:
            net.grad_side.join()  # the weight-gradient side stream has finished writing the flat gradient buffer
        if not net._flat_intact():
            net._check_flat()
        if self.uses_collective():
            dist.all_reduce(net.flat_grads, op=dist.ReduceOp.SUM)
            return 1.0 / dist.get_world_size()
        return 1.0

    def uses_collective(self) -> bool:
        """True when ``step()`` exchanges gradients: ``all_reduce`` under an initialised process group with more than
        one rank (or ``force_collective``)."""
        return bool(self.all_reduce and dist.is_available() and dist.is_initialized()
                    and (dist.get_world_size() > 1 or self.force_collective))

    @torch.no_grad()
    def step(self, closure=None, grad_scale: float = 1.0):
        """``grad_scale``: factor applied to the (all-reduced) gradient inside the update — ``1 / k`` after ``k`` micro-batches
        whose backward passes accumulated into the flat gradient buffer (the reference's production run uses
        ``accumulate_grad_batches: 3``, ``configs/experiment/RandLaNet_base_run_FR.yaml:18``; Lightning divides each
        micro-batch loss by ``k`` instead: the same mean)."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        net = self.net
        scale = self.reduce_gradients() * float(grad_scale)
        flat_p, flat_g = net.flat_parameters, net.flat_grads
        if flat_p.data_ptr() != getattr(self, "_bound_ptr", flat_p.data_ptr()):
            raise RuntimeError("FusedAdam: the net was re-flattened after the optimizer was created")
        self._bound_ptr = flat_p.data_ptr()
        g = self.param_groups[0]
        call("m3d_adam_step", flat_p.data_ptr(), flat_g.data_ptr(), self.exp_avg.data_ptr(),
             self.exp_avg_sq.data_ptr(), self._adam_state.data_ptr(),
             None if self.lr_dev is None else self.lr_dev.data_ptr(), float(g["lr"]), float(g["betas"][0]),
             float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]), scale, 1, flat_p.numel(),
             torch.cuda.current_stream().cuda_stream)
        net.invalidate_eval_cache()  # the weights changed behind the module's back (raw-pointer update)
        return loss

    def zero_grad(self, set_to_none: bool = False):  # gradients are views of the flat buffer: never detach them
        self.net.flat_grads.zero_()
