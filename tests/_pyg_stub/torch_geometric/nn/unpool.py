"""``torch_geometric.nn.unpool.knn_interpolate`` (pyg 2.4 ``nn/unpool/knn_interpolate.py``)."""
import torch

from torch_scatter import scatter

from .pool import knn


def knn_interpolate(x, pos_x, pos_y, batch_x=None, batch_y=None, k=3, num_workers=1):
    """Inverse-squared-distance interpolation of ``x`` (given at ``pos_x``) onto ``pos_y`` from the ``k`` nearest
    ``pos_x`` of the same cloud; weights ``1 / clamp(d^2, min=1e-16)`` are computed under ``no_grad``."""
    with torch.no_grad():
        assign_index = knn(pos_x, pos_y, k, batch_x=batch_x, batch_y=batch_y, num_workers=num_workers)
        y_idx, x_idx = assign_index[0], assign_index[1]
        diff = pos_x[x_idx] - pos_y[y_idx]
        squared_distance = (diff * diff).sum(dim=-1, keepdim=True)
        weights = 1.0 / torch.clamp(squared_distance, min=1e-16)
    y = scatter(x[x_idx] * weights, y_idx, 0, dim_size=pos_y.size(0), reduce="sum")
    y = y / scatter(weights, y_idx, 0, dim_size=pos_y.size(0), reduce="sum")
    return y
