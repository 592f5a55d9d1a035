"""Demo-only import of the reference (``pyg_randla_net.py:10``)."""


class ShapeNet:  # placeholder
    def __init__(self, *a, **k):
        raise NotImplementedError("ShapeNet is not part of the RandLA-Net path (stub)")
