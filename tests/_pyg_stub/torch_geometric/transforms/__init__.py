"""Demo-only import of the reference (``pyg_randla_net.py:7``, used in its ShapeNet ``main()`` only)."""


def __getattr__(name):
    raise AttributeError(f"torch_geometric.transforms.{name}: not part of the RandLA-Net path (stub)")
