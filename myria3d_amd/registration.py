"""Registration shim: make ``HipRandLANet`` (and the ``HipPointNet2`` variant) selectable through Myria3D's own class factory.

``myria3d.models.model.get_neural_net_class`` returns the first class of ``MODEL_ZOO`` whose ``__name__``
*contains* ``neural_net_class_name`` (``/root/reference/myria3d/models/model.py:12-29``).  ``"HipRandLANet"`` is not
a substring of ``"PyGRandLANet"`` (nor vice versa), so both stay individually addressable.
"""
from __future__ import annotations

from .pointnet2 import HipPointNet2
from .randla import HipRandLANet


def register_in_model_zoo() -> bool:
    """Append ``HipRandLANet`` and ``HipPointNet2`` to ``myria3d.models.model.MODEL_ZOO`` if Myria3D is importable.
    Idempotent."""
    try:
        from myria3d.models import model as m3d_model  # type: ignore
    except Exception:
        return False
    for cls in (HipRandLANet, HipPointNet2):
        if cls not in m3d_model.MODEL_ZOO:
            m3d_model.MODEL_ZOO.append(cls)
    return True
