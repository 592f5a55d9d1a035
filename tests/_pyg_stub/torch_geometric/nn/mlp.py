"""``torch_geometric.nn.MLP`` with its ``Linear`` / ``BatchNorm`` wrappers (pyg 2.4 ``nn/models/mlp.py``,
``nn/dense/linear.py``, ``nn/norm/batch_norm.py``)."""
import math

import torch
import torch.nn.functional as F
from torch import nn


class Linear(nn.Module):
    """PyG ``Linear``: ``F.linear(x, weight, bias)``, weight ``[out, in]``; default initialisation
    kaiming-uniform(a=sqrt(5)) for the weight and U(+-1/sqrt(in)) for the bias (= ``torch.nn.Linear``'s)."""

    def __init__(self, in_channels, out_channels, bias=True):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            bound = 1.0 / math.sqrt(self.in_channels) if self.in_channels > 0 else 0.0
            nn.init.uniform_(self.bias, -bound, bound)

    def forward(self, x):
        return F.linear(x, self.weight, self.bias)


class BatchNorm(nn.Module):
    """PyG ``BatchNorm``: a ``torch.nn.BatchNorm1d`` held as ``.module`` (state_dict keys ``norms.i.module.*``)."""

    def __init__(self, in_channels, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True,
                 allow_single_element=False):
        super().__init__()
        self.module = nn.BatchNorm1d(in_channels, eps, momentum, affine, track_running_stats)
        self.in_channels = in_channels

    def forward(self, x):
        return self.module(x)


def _activation(act, act_kwargs):
    if act is None:
        return None
    if not isinstance(act, str):
        return act
    table = {name.lower(): getattr(nn, name) for name in dir(nn) if isinstance(getattr(nn, name), type)}
    return table[act.lower()](**(act_kwargs or {}))


class MLP(nn.Module):
    """``MLP(channel_list, dropout=0., act="relu", act_first=False, act_kwargs=None, norm="batch_norm",
    norm_kwargs=None, plain_last=True, bias=True)``.  Per hidden layer: ``lin -> norm -> act -> dropout``; with
    ``plain_last=False`` the last layer gets norm / act / dropout as well (one norm per Linear)."""

    def __init__(self, channel_list=None, *, dropout=0.0, act="relu", act_first=False, act_kwargs=None,
                 norm="batch_norm", norm_kwargs=None, plain_last=True, bias=True, **kwargs):
        super().__init__()
        assert isinstance(channel_list, (tuple, list)) and len(channel_list) >= 2
        self.channel_list = list(channel_list)
        nl = len(channel_list) - 1
        self.act = _activation(act, act_kwargs)
        self.act_first = act_first
        self.plain_last = plain_last
        if isinstance(dropout, float):
            dropout = [dropout] * nl
            if plain_last:
                dropout[-1] = 0.0
        if len(dropout) != nl:
            raise ValueError(f"Number of dropout values provided ({len(dropout)}) does not match the number of layers ({nl})")
        self.dropout = dropout
        if isinstance(bias, bool):
            bias = [bias] * nl
        self.lins = nn.ModuleList([Linear(i, o, bias=b) for i, o, b in zip(channel_list[:-1], channel_list[1:], bias)])
        self.norms = nn.ModuleList()
        for width in (channel_list[1:-1] if plain_last else channel_list[1:]):
            if norm is None:
                self.norms.append(nn.Identity())
            elif norm in ("batch_norm", "BatchNorm", "batchnorm"):
                self.norms.append(BatchNorm(width, **(norm_kwargs or {})))
            else:
                raise NotImplementedError(f"stub: norm={norm!r}")

    def forward(self, x):
        for i, (lin, norm) in enumerate(zip(self.lins, self.norms)):
            x = lin(x)
            if self.act is not None and self.act_first:
                x = self.act(x)
            x = norm(x)
            if self.act is not None and not self.act_first:
                x = self.act(x)
            x = F.dropout(x, p=self.dropout[i], training=self.training)
        if self.plain_last:
            x = self.lins[-1](x)
            x = F.dropout(x, p=self.dropout[-1], training=self.training)
        return x
