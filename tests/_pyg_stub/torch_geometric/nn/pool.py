"""``knn`` / ``knn_graph`` of ``torch_cluster`` as re-exported by ``torch_geometric.nn.pool`` (torch-cluster 1.6
``knn.py``).  Exact brute force per cloud on squared-L2 distances computed as
``sum((a - b)^2)`` in the input dtype, ascending, ties by lower index (the real CPU path is a nanoflann KD-tree and the
CUDA path a brute force: both exact, tie order unspecified)."""
import torch


def knn(x, y, k, batch_x=None, batch_y=None, cosine=False, num_workers=1):
    """For every row of ``y`` its (at most) ``k`` nearest rows of ``x`` in the same cloud.  Returns ``[2, E]``:
    row 0 = index into ``y``, row 1 = index into ``x``; grouped by ``y`` row, ascending distance."""
    assert not cosine
    if batch_x is None:
        batch_x = x.new_zeros(x.size(0), dtype=torch.long)
    if batch_y is None:
        batch_y = y.new_zeros(y.size(0), dtype=torch.long)
    if x.numel() == 0 or y.numel() == 0:
        return torch.empty(2, 0, dtype=torch.long, device=x.device)
    rows, cols = [], []
    nb = int(max(batch_x.max(), batch_y.max())) + 1
    for b in range(nb):
        ix = (batch_x == b).nonzero().view(-1)
        iy = (batch_y == b).nonzero().view(-1)
        if ix.numel() == 0 or iy.numel() == 0:
            continue
        kk = min(k, ix.numel())
        for s in range(0, iy.numel(), 2048):
            yy = iy[s:s + 2048]
            diff = y[yy][:, None, :] - x[ix][None, :, :]
            d2 = (diff * diff).sum(-1)
            order = torch.sort(d2, dim=1, stable=True).indices[:, :kk]
            rows.append(yy[:, None].expand(-1, kk).reshape(-1))
            cols.append(ix[order].reshape(-1))
    return torch.stack([torch.cat(rows), torch.cat(cols)], dim=0)


def knn_graph(x, k, batch=None, loop=False, flow="source_to_target", cosine=False, num_workers=1, batch_size=None):
    """``knn(x, x, k if loop else k + 1, batch, batch)``; with ``source_to_target`` row 0 of the result is the
    NEIGHBOUR (source j) and row 1 the query (target i); ``loop=False`` drops the self matches."""
    assert flow in ("source_to_target", "target_to_source")
    edge_index = knn(x, x, k if loop else k + 1, batch, batch, cosine, num_workers)
    if flow == "source_to_target":
        row, col = edge_index[1], edge_index[0]
    else:
        row, col = edge_index[0], edge_index[1]
    if not loop:
        mask = row != col
        row, col = row[mask], col[mask]
    return torch.stack([row, col], dim=0)
