"""Import-name compatibility with the reference package: ``from ilqr import iLQR``,
``from ilqr.cartpole import CartPoleDynamics``, ``from ilqr.ilqr_cost import QRCost`` ...
resolve to the MI355X implementation in ``ilqr_amd``."""
import importlib
import sys

from ilqr_amd import iLQR, RecedingHorizonController  # noqa: F401

for _name in ("ilqr_core", "ilqr_boxQP", "ilqr_cost", "ilqr_dynamics", "cartpole", "cartpole_gym"):
    sys.modules[__name__ + "." + _name] = importlib.import_module("ilqr_amd." + _name)
