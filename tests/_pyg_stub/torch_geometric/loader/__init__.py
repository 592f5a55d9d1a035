"""Demo-only import of the reference (``pyg_randla_net.py:11``)."""


class DataLoader:  # placeholder
    def __init__(self, *a, **k):
        raise NotImplementedError("DataLoader is not part of the RandLA-Net path (stub)")
