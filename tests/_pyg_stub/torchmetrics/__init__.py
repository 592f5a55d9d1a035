"""Demo-only import of the reference (``pyg_randla_net.py:18``)."""
