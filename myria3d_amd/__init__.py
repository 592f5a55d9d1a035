"""myria3d_amd — MI355X (gfx950) native RandLA-Net hot path for IGNF/myria3d.

Only what the path needs: ``csrc/`` (HIP kernels + the C ABI of ``include/m3d_hip.h``), the ctypes binding, and
the host-side mirror of the reference's operator interface (``HipRandLANet``, ``knn_interpolate``,
``scatter_sum``, ``DeviceInterpolator``, ``register_in_model_zoo``) plus the device-side data preparation that feeds
it (``transforms``: GridSampling, node budget, normalisations).
"""
from .randla import HipRandLANet, make_plan  # noqa: F401
from .pointnet2 import HipPointNet2  # noqa: F401
from .interpolation import DeviceInterpolator, knn_interpolate, predict_reduce, scatter_sum  # noqa: F401
from .registration import register_in_model_zoo  # noqa: F401
from .model_forward import SimpleBatch, collate_tiles, forward_like_model  # noqa: F401
from .train import FusedAdam, cross_entropy  # noqa: F401
from .graphed import GraphedStep  # noqa: F401
from . import tiling, transforms  # noqa: F401
from .predict import predict_cloud  # noqa: F401

__all__ = ["predict_cloud", "HipRandLANet", "HipPointNet2", "make_plan", "knn_interpolate", "scatter_sum", "predict_reduce", "DeviceInterpolator",
           "register_in_model_zoo", "forward_like_model", "collate_tiles", "SimpleBatch", "FusedAdam", "cross_entropy", "GraphedStep", "transforms", "tiling"]
