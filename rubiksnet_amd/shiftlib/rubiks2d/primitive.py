"""RubiksShift2D functional layer on top of the HIP library.

Counterpart of rubiksnet/shiftlib/rubiks2d/primitive.py:9-205, same public names/signatures:

    rubiks2d_forward(x, shift, stride=1, padding=0, quantize=False, output=None)
    rubiks2d_backward(upstream_grad, x, shift, stride, padding, normalize_grad=True,
                      enable_shift_grad=True, quantize=False, x_grad_output=None,
                      shift_grad_output=None)
    rubiks2d(x, shift, stride=1, padding=0, normalize_grad=True, enable_shift_grad=True,
             quantize=False)                                        # autograd-aware

x is [N, C, H, W], shift is [2, C] with rows (H, W); float32 / float64 / float16 (+ bfloat16).
"""
import torch

from rubiksnet_amd import _native, rubiksnet_cuda
from rubiksnet_amd.utils import allocate_output, make_tuple

__all__ = ["rubiks2d", "rubiks2d_forward", "rubiks2d_backward"]

_DIM = 2


def compute_output_shape(x, stride, padding, shift_dim=_DIM):
    """[N, C, H, W] -> [N, C, Ho, Wo] with the library's own length rule (`rk_out_len`,
    cuda_src/rubiks.cpp:18): `(L + 2*pad - 1) // stride + 1`, not the convolution formula."""
    assert shift_dim == _DIM, "only the 2-D shift is defined here (rubiks2d/primitive.py:15-27)"
    out_len = _native.lib().rk_out_len
    moved = [int(out_len(int(length), s, p)) for length, s, p in
             zip(x.shape[2:], make_tuple(stride, _DIM), make_tuple(padding, _DIM))]
    return (int(x.shape[0]), int(x.shape[1]), *moved)


def _mixed(x, shift):
    """16-bit activations with an fp32 shift table: handled natively (the shift stays fp32 in the kernels)."""
    return x.dtype in (torch.float16, torch.bfloat16) and shift.dtype == torch.float32


def rubiks2d_forward(x, shift, stride=1, padding=0, quantize=False, output=None):
    """Pure forward primitive (rubiks2d/primitive.py:44-67)."""
    strides = make_tuple(stride, repeats=_DIM)
    paddings = make_tuple(padding, repeats=_DIM)
    assert x.is_cuda, "shift only works on CUDA tensors"
    assert x.dtype == shift.dtype or _mixed(x, shift), "x and shift must have the same dtype"
    out_shape = compute_output_shape(x, strides, paddings, shift_dim=_DIM)
    # quantize leaves out-of-range outputs untouched (rubiks2d_kernels.cu:116-121) -> needs zeros
    output = allocate_output(output, x, out_shape, zero=bool(quantize))
    ret = rubiksnet_cuda.rubiks2d_forward(
        input=x.contiguous(), shift=shift.contiguous(), strides=strides, paddings=paddings,
        quantize=quantize, output=output)
    assert ret == 0, "HIP kernel return code {} != 0, error".format(ret)
    return output


def rubiks2d_backward(upstream_grad, x, shift, stride, padding, normalize_grad=True, enable_shift_grad=True,
                      quantize=False, x_grad_output=None, shift_grad_output=None):
    """Pure backward primitive (rubiks2d/primitive.py:76-121): returns (x_grad, shift_grad)."""
    strides = make_tuple(stride, repeats=_DIM)
    paddings = make_tuple(padding, repeats=_DIM)
    assert x.is_cuda and upstream_grad.is_cuda and shift.is_cuda, "shift only works on CUDA tensors"
    x_grad = allocate_output(x_grad_output, x, x.size(), zero=bool(quantize))
    # untouched when enable_shift_grad is False (rubiks.cpp:126) -> must read as zeros
    shift_grad = allocate_output(shift_grad_output, shift, shift.size(), zero=not enable_shift_grad)
    ret = rubiksnet_cuda.rubiks2d_backward(
        upstream_grad=upstream_grad.contiguous(), input=x.contiguous(), shift=shift.contiguous(),
        strides=strides, paddings=paddings, normalize_grad=normalize_grad,
        enable_shift_grad=enable_shift_grad, quantize=quantize, input_grad=x_grad, shift_grad=shift_grad)
    assert ret == 0, "HIP return code {} != 0, error".format(ret)
    return x_grad, shift_grad


class VFS2DFunc(torch.autograd.Function):
    """autograd wiring (rubiks2d/primitive.py:133-174; the reference names the class VFS2DFunc)."""

    @staticmethod
    def forward(ctx, x, shift, stride, padding, normalize_grad, enable_shift_grad, quantize):
        assert isinstance(normalize_grad, bool)
        assert isinstance(enable_shift_grad, bool)
        ctx.stride = stride
        ctx.padding = padding
        ctx.normalize_grad = normalize_grad
        ctx.enable_shift_grad = enable_shift_grad
        ctx.quantize = quantize
        ctx.save_for_backward(x, shift)
        return rubiks2d_forward(x, shift, stride, padding, quantize)

    @staticmethod
    def backward(ctx, grad_output):
        x, shift = ctx.saved_tensors
        x_grad = shift_grad = None
        if any(ctx.needs_input_grad):
            _x_grad, _shift_grad = rubiks2d_backward(
                grad_output, x, shift, stride=ctx.stride, padding=ctx.padding,
                normalize_grad=ctx.normalize_grad,
                enable_shift_grad=ctx.enable_shift_grad and ctx.needs_input_grad[1],
                quantize=ctx.quantize)
            if ctx.needs_input_grad[0]:
                x_grad = _x_grad
            if ctx.needs_input_grad[1]:
                shift_grad = _shift_grad
        return x_grad, shift_grad, None, None, None, None, None


def rubiks2d(x, shift, stride=1, padding=0, normalize_grad=True, enable_shift_grad=True, quantize=False):
    """User-facing functional (rubiks2d/primitive.py:177-196)."""
    assert len(x.size()) == 4, "x must be [N, C, H, W]"
    # autocast: 16-bit activations next to the fp32 parameter.  The parameter is NOT rounded to the activations' type
    # (2^-8 relative on the interpolation weights in bf16, and it can flip the 0.5 side of `quantize`): the kernels
    # take the fp32 table as it is and return d(shift) in fp32 (rk2d_*_sf32).
    return VFS2DFunc.apply(x, shift, stride, padding, normalize_grad, enable_shift_grad, quantize)
