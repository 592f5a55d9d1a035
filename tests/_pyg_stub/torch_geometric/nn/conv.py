"""``torch_geometric.nn.conv.MessagePassing`` reduced to what ``LocalFeatureAggregation`` uses
(pyg 2.4 ``nn/conv/message_passing.py``): ``propagate(edge_index, **kwargs)`` with
``flow="source_to_target"`` collects the arguments of ``message`` by name — ``<name>_j = kwargs[name][edge_index[0]]``,
``<name>_i = kwargs[name][edge_index[1]]``, ``index = edge_index[1]`` — then aggregates the messages over ``index``
(``aggr="add"``: scatter-add, ``dim_size`` = number of nodes) and passes the result through ``update``."""
import inspect

import torch

from torch_scatter import scatter


class MessagePassing(torch.nn.Module):
    def __init__(self, aggr="add", flow="source_to_target", node_dim=-2):
        super().__init__()
        if aggr not in ("add", "sum", "mean", "max"):
            raise NotImplementedError(f"stub: aggr={aggr!r}")
        self.aggr = "sum" if aggr == "add" else aggr
        assert flow in ("source_to_target", "target_to_source")
        self.flow = flow
        self.node_dim = node_dim

    def propagate(self, edge_index, size=None, **kwargs):
        j, i = (0, 1) if self.flow == "source_to_target" else (1, 0)
        num_nodes = None
        for v in kwargs.values():
            if torch.is_tensor(v):
                num_nodes = v.size(self.node_dim)
                break
        if size is not None:
            num_nodes = size[i] if not isinstance(size, int) else size
        args = {}
        for name in inspect.signature(self.message).parameters:
            if name.endswith("_j"):
                args[name] = kwargs[name[:-2]].index_select(self.node_dim, edge_index[j])
            elif name.endswith("_i"):
                args[name] = kwargs[name[:-2]].index_select(self.node_dim, edge_index[i])
            elif name == "index":
                args[name] = edge_index[i]
            elif name == "edge_index":
                args[name] = edge_index
            elif name in ("ptr", "size_i", "size_j", "dim_size"):
                args[name] = None if name == "ptr" else num_nodes
            else:
                args[name] = kwargs[name]
        out = self.message(**args)
        out = scatter(out, edge_index[i], dim=self.node_dim, dim_size=num_nodes, reduce=self.aggr)
        return self.update(out)

    def message(self, x_j):
        return x_j

    def update(self, inputs):
        return inputs
