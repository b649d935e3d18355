"""MI355X-native batched control-limited iLQR: a drop-in for the solve path of
ZhengYi0310/ilqr (iLQR.update_u, RecedingHorizonController, Dynamics/Cost, boxQP).

    from ilqr_amd import iLQR, RecedingHorizonController
    from ilqr_amd.cartpole import CartPoleDynamics
    from ilqr_amd.ilqr_cost import QRCost
"""
__version__ = "0.1.0"

from .ilqr_core import iLQR, RecedingHorizonController  # noqa: E402,F401
from .ilqr_dynamics import Dynamics  # noqa: E402,F401
from .ilqr_cost import Cost, PathQRCost, QRCost  # noqa: E402,F401

__all__ = ["iLQR", "RecedingHorizonController", "Dynamics", "Cost", "QRCost", "PathQRCost"]
