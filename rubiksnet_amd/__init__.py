"""rubiksnet_amd -- MI355X-native RubiksShift operators and RubiksNet video backbone.

Public surface mirrors the reference package `rubiksnet`: `shiftlib` (RubiksShift2D,
RubiksShift3D, RubiksShiftBase + functionals), `AttentionShift`, `RubiksNetBackbone`,
`RubiksNet`.  All device work goes through librubiks_hip.so (include/rubiks_hip.h);
importing the package does not load it, calling an operator does.
"""
from . import shiftlib  # noqa: F401
from .attention_shift import AttentionShift
from .backbone import RubiksNetBackbone
from .models import RubiksNet
from .shiftlib import RubiksShift2D, RubiksShift3D, RubiksShiftBase

__all__ = ["RubiksNet", "RubiksNetBackbone", "AttentionShift", "RubiksShift2D", "RubiksShift3D",
           "RubiksShiftBase", "shiftlib"]
__version__ = "0.1.0"
