"""nn.Module interface of RubiksShift2D (counterpart of rubiksnet/shiftlib/rubiks2d/layer.py:6-52)."""
import torch
import torch.nn as nn

from .primitive import rubiks2d

__all__ = ["RubiksShift2D", "init_shift_group"]


def init_shift_group(shift, kernel_size):
    """Integer shifts enumerating a K x K neighbourhood, repeated over channel groups -- the
    zero-FLOP "group shift" init (layer.py:6-15).  Puts every channel on the integer-shift
    branch of d(shift) (rubiks2d_kernels.cu:189-253)."""
    K = kernel_size
    C = shift.size(1)
    half = kernel_size // 2
    offsets = torch.arange(-half, half + 1, dtype=shift.dtype)
    groups = C // K ** 2
    shift[0, :] = offsets.repeat(K * groups)
    shift[1, :] = offsets.repeat_interleave(K).repeat(groups)


class RubiksShift2D(nn.Module):
    """Learnable fractional (H, W) shift of an [N, C, H, W] tensor; `shift` is [2, C]."""

    def __init__(self, num_channels, stride=1, padding=0, normalize_grad=True, quantize=False,
                 init_shift="uniform"):
        super().__init__()
        self.num_channels = num_channels
        self.stride = stride
        self.padding = padding
        self.normalize_grad = normalize_grad
        self.quantize = quantize
        self.shift = nn.Parameter(torch.zeros(2, num_channels))
        with torch.no_grad():
            if init_shift == "uniform":
                nn.init.uniform_(self.shift, -1, 1)
            elif init_shift.startswith("group"):
                group_kernel = int(init_shift[5:])
                assert group_kernel > 1
                init_shift_group(self.shift, group_kernel)
            else:
                raise NotImplementedError(f"unrecognized init shift {init_shift}")

    def forward(self, x):
        return rubiks2d(x, self.shift, stride=self.stride, padding=self.padding,
                        normalize_grad=self.normalize_grad, enable_shift_grad=True, quantize=self.quantize)

    def extra_repr(self):
        return "shift_channels={}".format(self.num_channels)
