"""Stub of ``torch_scatter.scatter`` (torch-scatter 2.1 ``scatter.py``)."""
import torch

IS_M3D_STUB = True


def _expand(index, src, dim):
    if dim < 0:
        dim += src.dim()
    if index.dim() == 1:
        shape = [1] * src.dim()
        shape[dim] = -1
        index = index.view(shape)
    return index.expand_as(src), dim


def scatter(src, index, dim=-1, out=None, dim_size=None, reduce="sum"):
    """``out[index[i]] (op)= src[i]`` along ``dim`` (1-D ``index`` broadcast over the other dimensions).  ``sum`` /
    ``add``, ``mean`` (count clamped to 1), ``max`` (empty groups = 0, like torch_scatter)."""
    idx, dim = _expand(index, src, dim)
    if out is None:
        size = list(src.shape)
        size[dim] = dim_size if dim_size is not None else (int(index.max()) + 1 if index.numel() else 0)
        out = src.new_zeros(size)
        fresh = True
    else:
        fresh = False
    if reduce in ("sum", "add"):
        return out.scatter_add_(dim, idx, src)
    if reduce == "mean":
        tot = out.scatter_add_(dim, idx, src)
        cnt = torch.zeros_like(tot).scatter_add_(dim, idx, torch.ones_like(src)).clamp_(min=1)
        return tot / cnt
    if reduce == "max":
        if not fresh:
            raise NotImplementedError("stub: scatter(max) with out=")
        return out.scatter_reduce(dim, idx, src, reduce="amax", include_self=False)
    raise ValueError(reduce)


def scatter_sum(src, index, dim=-1, out=None, dim_size=None):
    return scatter(src, index, dim, out, dim_size, "sum")
