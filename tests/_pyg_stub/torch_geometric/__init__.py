"""Stub of ``torch_geometric`` (pyg 2.4; reference ``environment.yml:19``) — see tests/_pyg_stub/README.md."""
__version__ = "2.4.0+m3d.stub"
IS_M3D_STUB = True
