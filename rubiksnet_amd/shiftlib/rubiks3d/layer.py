"""nn.Module interface of RubiksShift3D (API of rubiksnet/shiftlib/rubiks3d/layer.py:21-154).

Public surface kept for drop-in use: `RubiksShiftBase`, `RubiksShift3D`, `init_shift_uniform`,
`init_shift1d_nfold`, `create_3d_from_2d`; attribute names (`shift`, `num_channels`, `stride`, `padding`,
`normalize_grad`, `normalize_t_factor`, `quantize`) and the single state-dict key `shift` are the
reference's, so its checkpoints load unchanged.  The temporal-row initialisers are organised as a table of
small functions keyed by the `init_mode` prefix.
"""
import math

import torch
import torch.nn as nn

from rubiksnet_amd.utils import make_tuple
from ..rubiks2d.layer import RubiksShift2D
from .primitive import rubiks_shift_3d

__all__ = [
    "RubiksShift3D",
    "RubiksShiftBase",
    "init_shift_uniform",
    "init_shift1d_nfold",
    "create_3d_from_2d",
]

_CONFIG_FIELDS = ("stride", "padding", "normalize_grad", "normalize_t_factor", "quantize")


def init_shift_uniform(shift):
    """Every learnable shift starts as U(-1, 1) (reference layer.py:21-22)."""
    with torch.no_grad():
        shift.uniform_(-1.0, 1.0)
    return shift


def _fold_slices(channels, nfold):
    """(channels looking one step back, one step ahead, staying put) of a TSM-style 1/nfold split."""
    g = channels // nfold
    return slice(0, g), slice(g, 2 * g), slice(2 * g, channels)


def init_shift1d_nfold(shift, nfold=8, noise=1e-3):
    """TSM-like init of a 1-row shift table (reference layer.py:25-40): +1 / -1 folds, the remaining channels get
    U(-noise, noise) so they start next to -- not on -- the integer-shift branch of d(shift)."""
    rows, channels = shift.shape
    assert rows == 1, "only works with rubiks1d"
    back, ahead, rest = _fold_slices(channels, nfold)
    with torch.no_grad():
        shift[0, back] = 1.0
        shift[0, ahead] = -1.0
        shift[0, rest] = torch.empty(channels - rest.start).uniform_(-noise, noise)
    return shift


class RubiksShiftBase(nn.Module):
    """Holds the learnable table `shift` [dim, C // shift_groups] and applies `shift_function` with the stored
    configuration (reference layer.py:43-83)."""

    def __init__(self, num_channels, stride=1, padding=0, normalize_grad=True, normalize_t_factor=1.0,
                 shift_groups=1, quantize=False, *, dim, shift_function):
        super().__init__()
        assert num_channels % shift_groups == 0, "Does not satisfy num_channels % shift_groups == 0"
        self.num_channels = num_channels
        config = dict(stride=stride, padding=padding, normalize_grad=normalize_grad,
                      normalize_t_factor=normalize_t_factor, quantize=quantize)
        for field in _CONFIG_FIELDS:
            setattr(self, field, config[field])
        self.shift = nn.Parameter(init_shift_uniform(torch.empty(dim, num_channels // shift_groups)))
        self.shift_function = shift_function

    def forward(self, x):
        return self.shift_function(x, self.shift, **{field: getattr(self, field) for field in _CONFIG_FIELDS})

    def extra_repr(self):
        return f"shift_channels={self.num_channels}"


class RubiksShift3D(RubiksShiftBase):
    """Learnable fractional (T, H, W) shift of an [N, T, C, H, W] tensor (reference layer.py:86-108)."""

    def __init__(self, num_channels, stride=(1, 1, 1), padding=(0, 0, 0), normalize_grad=True,
                 normalize_t_factor=1.0, quantize=False, shift_groups=1):
        super().__init__(num_channels, stride, padding, normalize_grad, normalize_t_factor, shift_groups, quantize,
                         dim=3, shift_function=rubiks_shift_3d)


# ---- temporal-row initialisers of create_3d_from_2d: name prefix -> fn(table [3,C], numeric suffix) ------------
def _t_folds_exact(table, _):
    back, ahead, rest = _fold_slices(table.shape[1], 8)
    for sl, value in ((back, 1.0), (ahead, -1.0), (rest, 0.0)):
        table[0, sl] = value


def _t_folds_gaussian(table, suffix):
    std = float(suffix) or 1e-2                      # "tsm-g0" means the default spread
    back, ahead, rest = _fold_slices(table.shape[1], 8)
    for sl, centre in ((back, 1.0), (ahead, -1.0), (rest, 0.0)):
        n = len(range(*sl.indices(table.shape[1])))
        table[0, sl] = centre + std * torch.randn(n)


def _t_scaled_uniform(table, suffix):
    magnitude = float(suffix)
    assert magnitude > 0 and math.isfinite(magnitude), f"uniform random magnitude must > 0: {magnitude}"
    table[0] *= magnitude                            # the row already holds U(-1, 1)


def _t_unset(table, _):
    table.fill_(float("nan"))                        # to be overwritten by a checkpoint


# longest prefix first ("tsm-g" before "tsm")
_T_ROW_INITS = (("tsm-g", _t_folds_gaussian, True), ("tsm", _t_folds_exact, False), ("uni", _t_scaled_uniform, True),
                ("none", _t_unset, False))


def create_3d_from_2d(module_2d, init_mode="tsm", normalize_t_factor=1.0, quantize=False):
    """3-D layer with the (H, W) rows of a RubiksShift2D and a temporal row chosen by `init_mode`
    (reference layer.py:111-154): "tsm" exact +1 / -1 / 0 folds of C // 8 channels, "tsm-g<std>" the same with
    Gaussian noise, "uni<mag>" U(-mag, mag), "none" NaN (expects a state dict)."""
    assert isinstance(module_2d, RubiksShift2D)
    lifted = RubiksShift3D(module_2d.num_channels, stride=(1, *make_tuple(module_2d.stride, 2)),
                           padding=(0, *make_tuple(module_2d.padding, 2)), normalize_grad=True,
                           normalize_t_factor=normalize_t_factor, quantize=quantize)
    mode = init_mode.lower() if init_mode.lower() == "none" else init_mode
    for prefix, fill, takes_number in _T_ROW_INITS:
        if mode.startswith(prefix) and (takes_number or mode == prefix):
            with torch.no_grad():
                lifted.shift[1:] = module_2d.shift
                fill(lifted.shift, mode[len(prefix):])
            return lifted
    raise NotImplementedError(f"unknown init mode {init_mode}")
