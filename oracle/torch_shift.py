"""The CPU oracle as a torch.autograd.Function -- RubiksShift3D on host tensors.

TEST INFRASTRUCTURE -- NOT PRODUCT CODE (the product has no CPU path: rubiksnet_amd raises on a CPU tensor).  Used by
tests/test_dist_gloo.py (the data-parallel harness on CPU ranks) and by bench.py's cpu_baseline leg (the model-level CPU
column: RubiksNet-Tiny forward + backward on the host cores with this function plugged into every RubiksShift3D layer as
its `shift_function`).  Same argument list as rubiksnet/shiftlib/rubiks3d/primitive.py:193-215 (`rubiks_shift_3d`)."""
import torch

from . import oracle as orc


class OracleShift3D(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, shift, stride, padding, normalize_grad, normalize_t_factor, quantize):
        ctx.save_for_backward(x, shift)
        if normalize_t_factor == "auto":                     # primitive.py:206-210 (T / H of the input)
            normalize_t_factor = float(x.shape[1]) / float(x.shape[3])
        ctx.cfg = (stride, padding, normalize_grad, normalize_t_factor, quantize)
        return torch.from_numpy(orc.rk3d_forward(x.detach().contiguous().numpy(), shift.detach().contiguous().numpy(), stride, padding,
                                                 quantize))

    @staticmethod
    def backward(ctx, gy):
        x, shift = ctx.saved_tensors
        stride, padding, ng, tf, q = ctx.cfg
        gx, gs = orc.rk3d_backward(gy.contiguous().numpy(), x.detach().contiguous().numpy(), shift.detach().contiguous().numpy(), stride,
                                   padding, ng, tf, q)
        return torch.from_numpy(gx), torch.from_numpy(gs), None, None, None, None, None


def oracle_shift(x, shift, stride=1, padding=0, normalize_grad=True, normalize_t_factor=1.0, quantize=False):
    return OracleShift3D.apply(x, shift, stride, padding, normalize_grad, normalize_t_factor, quantize)


def plug_into(model):
    """Every RubiksShift3D layer of `model` evaluates through the oracle from now on (host tensors only).  Returns the count."""
    n = 0
    for m in model.modules():
        if type(m).__name__ == "RubiksShift3D" and hasattr(m, "shift_function"):
            m.shift_function = oracle_shift
            n += 1
    return n
