"""nn.Module interface of RubiksShift3D (counterpart of rubiksnet/shiftlib/rubiks3d/layer.py:21-154).

Attribute names (`shift`, `num_channels`, `stride`, `padding`, `normalize_grad`,
`normalize_t_factor`, `quantize`) and the state-dict key `shift` match the reference so its
checkpoints load unchanged.
"""
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


def init_shift_uniform(shift):
    """U(-1, 1), the default init of every learnable shift (layer.py:21-22)."""
    nn.init.uniform_(shift, -1, 1)


def init_shift1d_nfold(shift, nfold=8, noise=1e-3):
    """TSM-like 1D init (layer.py:25-40): 1/nfold of the channels look one step back, the next
    1/nfold one step ahead, the rest get +-noise so they stay off the integer-shift branch."""
    dim, channels = shift.size()
    assert dim == 1, "only works with rubiks1d"
    with torch.no_grad():
        group = channels // nfold
        shift[:, :group] = 1
        shift[:, group:2 * group] = -1
        shift[:, 2 * group:].uniform_(-noise, noise)
    return shift


class RubiksShiftBase(nn.Module):
    """Owns the learnable `shift` [dim, C // shift_groups] and forwards to the functional (layer.py:43-83)."""

    def __init__(self, num_channels, stride=1, padding=0, normalize_grad=True, normalize_t_factor=1.0,
                 shift_groups=1, quantize=False, *, dim, shift_function):
        super().__init__()
        self.num_channels = num_channels
        self.stride = stride
        self.padding = padding
        self.normalize_grad = normalize_grad
        self.normalize_t_factor = normalize_t_factor
        self.quantize = quantize
        assert num_channels % shift_groups == 0, "Does not satisfy num_channels % shift_groups == 0"
        self.shift = nn.Parameter(torch.zeros(dim, num_channels // shift_groups))
        init_shift_uniform(self.shift)
        self.shift_function = shift_function

    def forward(self, x):
        return self.shift_function(
            x,
            self.shift,
            stride=self.stride,
            padding=self.padding,
            normalize_grad=self.normalize_grad,
            normalize_t_factor=self.normalize_t_factor,
            quantize=self.quantize,
        )

    def extra_repr(self):
        return "shift_channels={}".format(self.num_channels)


class RubiksShift3D(RubiksShiftBase):
    """Learnable fractional (T, H, W) shift of an [N, T, C, H, W] tensor (layer.py:86-108)."""

    def __init__(self, num_channels, stride=(1, 1, 1), padding=(0, 0, 0), normalize_grad=True,
                 normalize_t_factor=1.0, quantize=False, shift_groups=1):
        super().__init__(num_channels, stride, padding, normalize_grad, normalize_t_factor, shift_groups,
                         quantize=quantize, dim=3, shift_function=rubiks_shift_3d)


def create_3d_from_2d(module_2d, init_mode="tsm", normalize_t_factor=1.0, quantize=False):
    """Lift a RubiksShift2D into 3D keeping its (H, W) shifts; `init_mode` picks the temporal row
    (layer.py:111-154): 'tsm' (exact +1/-1/0 folds), 'tsm-g<std>' (noisy folds), 'uni<mag>'
    (scaled uniform), 'none' (NaN-filled, must be loaded later)."""
    assert isinstance(module_2d, RubiksShift2D)
    module_3d = RubiksShift3D(
        module_2d.num_channels,
        stride=(1, *make_tuple(module_2d.stride, 2)),
        padding=(0, *make_tuple(module_2d.padding, 2)),
        normalize_grad=True,
        normalize_t_factor=normalize_t_factor,
        quantize=quantize,
    )
    with torch.no_grad():
        D, C = module_3d.shift.size()
        assert D == 3, "INTERNAL ERROR"
        module_3d.shift[1:, :] = module_2d.shift
        t_row = module_3d.shift[0, :]
        fold = C // 8
        if init_mode.startswith("tsm-g"):
            stddev = float(init_mode[5:])
            if stddev == 0:
                stddev = 1e-2
            t_row[:fold] = 1.0 + torch.randn((fold,)) * stddev
            t_row[fold:2 * fold] = -1.0 + torch.randn((fold,)) * stddev
            t_row[2 * fold:] = torch.randn((C - 2 * fold,)) * stddev
        elif init_mode == "tsm":
            t_row[:fold].fill_(1)
            t_row[fold:2 * fold].fill_(-1)
            t_row[2 * fold:].fill_(0)
        elif init_mode.startswith("uni"):
            magnitude = float(init_mode[3:])
            assert magnitude > 0, f"uniform random magnitude must > 0: {magnitude}"
            t_row *= magnitude
        elif init_mode.lower() == "none":
            module_3d.shift.fill_(float("nan"))
        else:
            raise NotImplementedError(f"unknown init mode {init_mode}")
    return module_3d
