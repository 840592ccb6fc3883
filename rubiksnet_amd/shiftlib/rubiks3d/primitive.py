"""RubiksShift3D functional layer on top of the HIP library.

Counterpart of rubiksnet/shiftlib/rubiks3d/primitive.py:13-236 with the same public names
and signatures:

    rubiks_shift_3d_forward(x, shift, stride, padding, quantize=False, output=None)
    rubiks_shift_3d_backward(upstream_grad, x, shift, stride, padding, normalize_grad,
                             normalize_t_factor=1.0, quantize=False,
                             x_grad_output=None, shift_grad_output=None)
    rubiks_shift_3d(x, shift, stride=1, padding=0, normalize_grad=True,
                    normalize_t_factor=1.0, quantize=False)          # autograd-aware

x is [N, T, C, H, W] (NOT NCTHW), shift is [3, C] with rows (T, H, W).
"""
import torch

from rubiksnet_amd import rubiksnet_cuda
from rubiksnet_amd.utils import allocate_output

__all__ = [
    "rubiks_shift_3d_forward",
    "rubiks_shift_3d_backward",
    "rubiks_shift_3d",
]

_DIM = 3


def _make_tuple(elem, repeats):
    if isinstance(elem, int):
        return [elem] * repeats
    assert len(elem) == repeats
    return [int(x) for x in elem]


def _get_output_dim(orig, stride, padding):
    # cuda_src/rubiks.cpp:166; integer form of rubiks3d/primitive.py:25-26 (float divide + int())
    return (orig + 2 * padding - 1) // stride + 1


def compute_output_shape(x, stride, padding, shift_dim=_DIM):
    """Output size of the shift (rubiks3d/primitive.py:29-48); 1D/2D/3D select which dims move."""
    batch, T_in, C_in, H_in, W_in = x.size()
    T_out, H_out, W_out = T_in, H_in, W_in
    strides = _make_tuple(stride, shift_dim)
    paddings = _make_tuple(padding, shift_dim)
    if shift_dim == 1:
        T_out = _get_output_dim(T_in, strides[0], paddings[0])
    elif shift_dim == 2:
        H_out = _get_output_dim(H_in, strides[0], paddings[0])
        W_out = _get_output_dim(W_in, strides[1], paddings[1])
    elif shift_dim == 3:
        T_out = _get_output_dim(T_in, strides[0], paddings[0])
        H_out = _get_output_dim(H_in, strides[1], paddings[1])
        W_out = _get_output_dim(W_in, strides[2], paddings[2])
    else:
        raise NotImplementedError("only 1D, 2D, 3D shifts supported")
    return batch, int(T_out), C_in, int(H_out), int(W_out)


def _pick(x, f32, f64):
    if x.dtype == torch.float32:
        return f32
    if x.dtype == torch.float64:
        return f64
    raise ValueError("rubiks_shift_{}d only supports float32 and float64 (double) dtypes.".format(_DIM))


def rubiks_shift_3d_forward(x, shift, stride, padding, quantize=False, output=None):
    """Pure forward primitive, no autograd (rubiks3d/primitive.py:54-80)."""
    strides = _make_tuple(stride, _DIM)
    paddings = _make_tuple(padding, _DIM)
    assert x.is_cuda, "rubiks shift only works on CUDA tensors"
    assert x.size(2) == shift.size(1), "x tensor channel dim[2] must match shift channel dim[1]"
    assert x.dtype == shift.dtype, "x and shift must have the same dtype"
    func = _pick(x, rubiksnet_cuda.rubiks_shift_3d_forward_float, rubiksnet_cuda.rubiks_shift_3d_forward_double)
    out_shape = compute_output_shape(x, strides, paddings, shift_dim=_DIM)
    # the 3D kernels write every output element, so a fresh buffer need not be zeroed
    output = allocate_output(output, x, out_shape, zero=False)
    ret = func(x.contiguous(), shift.contiguous(), strides, paddings, quantize, output)
    assert ret == 0, "HIP kernel return code {} != 0, error".format(ret)
    return output


def rubiks_shift_3d_backward(
    upstream_grad,
    x,
    shift,
    stride,
    padding,
    normalize_grad,
    normalize_t_factor=1.0,
    quantize=False,
    x_grad_output=None,
    shift_grad_output=None,
    need_x_grad=True,
    need_shift_grad=True,
):
    """Pure backward primitive (rubiks3d/primitive.py:90-140): returns (x_grad, shift_grad).

    `need_x_grad` / `need_shift_grad` are additions: the reference always computes both;
    a skipped half comes back as None.
    """
    strides = _make_tuple(stride, _DIM)
    paddings = _make_tuple(padding, _DIM)
    assert x.is_cuda and upstream_grad.is_cuda, "rubiks shift only works on CUDA tensors"
    func = _pick(x, rubiksnet_cuda.rubiks_shift_3d_backward_float, rubiksnet_cuda.rubiks_shift_3d_backward_double)
    x_grad = allocate_output(x_grad_output, x, x.size(), zero=False) if need_x_grad else None
    shift_grad = allocate_output(shift_grad_output, shift, shift.size(), zero=False) if need_shift_grad else None
    # the reference hands a non-contiguous upstream grad to raw-pointer kernels unchecked
    # (SURVEY 3.2); make it dense instead of misreading it
    ret = func(
        x.contiguous(),
        shift.contiguous(),
        upstream_grad.contiguous(),
        strides,
        paddings,
        x_grad,
        shift_grad,
        normalize_grad,
        normalize_t_factor,
        quantize,
    )
    assert ret == 0, "HIP return code {} != 0, error".format(ret)
    return x_grad, shift_grad


class RubiksShift3DFunc(torch.autograd.Function):
    """autograd wiring, 7 inputs -> grads for the first two (rubiks3d/primitive.py:146-188)."""

    @staticmethod
    def forward(ctx, x, shift, stride, padding, normalize_grad, normalize_t_factor, quantize):
        assert isinstance(normalize_grad, bool)
        ctx.stride = stride
        ctx.padding = padding
        ctx.normalize_grad = normalize_grad
        ctx.normalize_t_factor = normalize_t_factor
        ctx.quantize = quantize
        ctx.save_for_backward(x, shift)
        return rubiks_shift_3d_forward(x, shift, stride, padding, quantize=quantize)

    @staticmethod
    def backward(ctx, grad_output):
        x, shift = ctx.saved_tensors
        x_grad = shift_grad = None
        if any(ctx.needs_input_grad):
            x_grad, shift_grad = rubiks_shift_3d_backward(
                grad_output,
                x,
                shift,
                stride=ctx.stride,
                padding=ctx.padding,
                normalize_grad=ctx.normalize_grad,
                normalize_t_factor=ctx.normalize_t_factor,
                quantize=ctx.quantize,
                need_x_grad=ctx.needs_input_grad[0],
                need_shift_grad=ctx.needs_input_grad[1],
            )
        return x_grad, shift_grad, None, None, None, None, None


def rubiks_shift_3d(x, shift, stride=1, padding=0, normalize_grad=True, normalize_t_factor=1.0, quantize=False):
    """User-facing functional (rubiks3d/primitive.py:193-215)."""
    assert len(x.size()) == 5, "x must be [N, T, C, H, W]"
    _, T, C, H, _ = x.size()
    assert C == shift.size(1), "group shift is deprecated. Now C dim must match."
    if normalize_t_factor == "auto":
        normalize_t_factor = T / H
    else:
        assert isinstance(normalize_t_factor, (int, float))
    if x.dtype in (torch.float16, torch.bfloat16):
        # autocast hands half activations to an fp32 parameter.  The reference's 3D op exists in fp32/fp64 only
        # (primitive.py:66-75): run it in fp32 and hand back the caller's dtype (the explicit-dtype primitives
        # below still raise for half inputs, as the reference does).
        y = RubiksShift3DFunc.apply(x.float(), shift.float(), stride, padding, normalize_grad, normalize_t_factor,
                                    quantize)
        return y.to(x.dtype)
    return RubiksShift3DFunc.apply(x, shift, stride, padding, normalize_grad, normalize_t_factor, quantize)
