"""nn.Module interface of RubiksShift2D (counterpart of rubiksnet/shiftlib/rubiks2d/layer.py:6-52:
same constructor signature, attribute names and state-dict key `shift`)."""
import torch
import torch.nn as nn

from .primitive import rubiks2d

__all__ = ["RubiksShift2D", "init_shift_group"]


def init_shift_group(shift, kernel_size):
    """Fill `shift` [2, C] in place with the integer offsets of a K x K window, one window position per channel, cycling
    over the channels (the zero-FLOP "group shift" init of layer.py:6-15).  The offsets are r = -(K//2) .. K//2 -- 2 (K//2) + 1
    of them, i.e. K for an odd window and K + 1 for an even one, exactly as the reference builds them: row 0 (H) cycles
    through r, row 1 (W) holds each entry of r K times, both tiled K * (C // K^2) (resp. C // K^2) times.  When the
    tiled length is not C the reference's assignment raises (every even K except the accidental C = K (K + 1) (C // K^2)),
    and so does this.  Every channel then sits on the integer-shift branch of d(shift) (rubiks2d_kernels.cu:189-253)."""
    K = int(kernel_size)
    C = shift.size(1)
    n_off = 2 * (K // 2) + 1
    length = n_off * K * (C // (K * K))
    if length != C:
        raise RuntimeError("init_shift_group: a %d x %d window tiles to %d offsets, the layer has %d channels" % (K, K, length, C))
    i = torch.arange(C)
    shift[0] = (i % n_off - K // 2).to(shift.dtype)
    shift[1] = ((i // K) % n_off - K // 2).to(shift.dtype)


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
