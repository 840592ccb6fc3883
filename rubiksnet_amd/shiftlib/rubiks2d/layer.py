"""nn.Module interface of RubiksShift2D (counterpart of rubiksnet/shiftlib/rubiks2d/layer.py:6-52:
same constructor signature, attribute names and state-dict key `shift`)."""
import torch
import torch.nn as nn

from .primitive import rubiks2d

__all__ = ["RubiksShift2D", "init_shift_group"]


def init_shift_group(shift, kernel_size):
    """Fill `shift` [2, C] in place with the integer offsets of a K x K window, one window position
    per channel, cycling over the channels (the zero-FLOP "group shift" init of layer.py:6-15).
    Channel c gets `c mod K - K//2` in row 0 (H) and `(c div K) mod K - K//2` in row 1 (W), the
    reference's order.  Every channel then sits on the integer-shift
    branch of d(shift) (rubiks2d_kernels.cu:189-253)."""
    K = int(kernel_size)
    C = shift.size(1)
    covered = (C // (K * K)) * K * K
    if covered != C:
        # the reference's `repeat` raises on the size mismatch as well
        raise RuntimeError("init_shift_group: %d channels are not a multiple of %d x %d" % (C, K, K))
    c = torch.arange(C)
    shift[0] = (c % K - K // 2).to(shift.dtype)
    shift[1] = ((c // K) % K - K // 2).to(shift.dtype)


def _fill_shift(shift, how):
    """`uniform`: U(-1, 1) like the 3-D layer; `group<K>`: init_shift_group with a K x K window."""
    if how == "uniform":
        return nn.init.uniform_(shift, -1, 1)
    if how.startswith("group"):
        window = int(how[len("group"):])
        assert window > 1
        return init_shift_group(shift, window)
    raise NotImplementedError(f"unrecognized init shift {how}")


class RubiksShift2D(nn.Module):
    """Learnable fractional (H, W) shift of an [N, C, H, W] tensor; `shift` is [2, C]."""

    def __init__(self, num_channels, stride=1, padding=0, normalize_grad=True, quantize=False,
                 init_shift="uniform"):
        super().__init__()
        for name, value in (("num_channels", num_channels), ("stride", stride), ("padding", padding),
                            ("normalize_grad", normalize_grad), ("quantize", quantize)):
            setattr(self, name, value)
        self.shift = nn.Parameter(torch.empty(2, num_channels))
        with torch.no_grad():
            _fill_shift(self.shift, init_shift)

    def forward(self, x):
        # the module always asks for d(shift) (layer.py:47)
        return rubiks2d(x, self.shift, self.stride, self.padding, self.normalize_grad, True, self.quantize)

    def extra_repr(self):
        return f"shift_channels={self.num_channels}"
