"""``torch_geometric.nn`` symbols the reference's net imports (``pyg_randla_net.py:12-15``)."""
from .mlp import MLP, BatchNorm, Linear  # noqa: F401
from .conv import MessagePassing  # noqa: F401
from .pool import knn, knn_graph  # noqa: F401
from .unpool import knn_interpolate  # noqa: F401
