"""rubiksnet_amd.shiftlib -- same public surface as rubiksnet/shiftlib/__init__.py:1-2."""
from .rubiks2d.layer import RubiksShift2D
from .rubiks3d.layer import RubiksShift3D, RubiksShiftBase

__all__ = ["RubiksShift2D", "RubiksShift3D", "RubiksShiftBase"]
