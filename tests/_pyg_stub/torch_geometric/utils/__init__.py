"""``torch_geometric.utils.softmax`` (pyg 2.4 ``utils/softmax.py``)."""
import torch

from torch_scatter import scatter


def softmax(src, index=None, ptr=None, num_nodes=None, dim=0):
    """Softmax over the entries that share an ``index`` value, per channel:
    ``m = scatter_max(src.detach())``, ``e = exp(src - m[index])``, ``s = scatter_sum(e) + 1e-16``, ``e / s[index]``."""
    if index is None or ptr is not None:
        raise NotImplementedError("stub: index-based softmax only (what pyg_randla_net.py:150 calls)")
    n = int(index.max()) + 1 if num_nodes is None else num_nodes
    src_max = scatter(src.detach(), index, dim, dim_size=n, reduce="max")
    out = (src - src_max.index_select(dim, index)).exp()
    out_sum = scatter(out, index, dim, dim_size=n, reduce="sum") + 1e-16
    return out / out_sum.index_select(dim, index)
