"""1x1 convolutions of the backbone through librubiks_hip's NCHW MFMA GEMM (rk_pw_gemm_f32), SURVEY 8(f) f1.

`conv1x1(conv, x)` evaluates an ordinary bias-free `nn.Conv2d(kernel_size=1, stride=1)` module -- the module,
its weight, its state-dict key stay what they are.  The HIP path takes the layers it wins on (fp32, large
planes: the memory-bound 112x112 / 56x56 stages, where MIOpen's NHWC implicit GEMM pays for two layout
transposes); everything else -- other dtypes, small planes, CPU tensors -- goes to `conv(x)` (MIOpen / stock).
`RK_PW=0` disables the HIP path, `RK_PW=all` forces it wherever the kernel's constraints allow.
Forward and d(input) are the HIP GEMM; d(weight) is still aten's convolution_backward.
"""
import os

import torch

from . import _native

__all__ = ["conv1x1", "pointwise_mode"]


def pointwise_mode():
    return os.environ.get("RK_PW", "auto")


def _gemm(a, x, out, Fr, K, M, P, a_is_mk):
    dev = x.device
    with torch.cuda.device(dev):
        rc = _native.lib().rk_pw_gemm_f32(a.data_ptr(), x.data_ptr(), out.data_ptr(), Fr, K, M, P, int(a_is_mk),
                                          torch.cuda.current_stream(dev).cuda_stream)
    _native.check(rc, "rk_pw_gemm_f32")
    return out


class _Conv1x1Func(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight):
        Fr, Cin, H, W = x.shape
        Cout = weight.shape[0]
        y = torch.empty(Fr, Cout, H, W, dtype=x.dtype, device=x.device)
        _gemm(weight, x, y, Fr, Cin, Cout, H * W, True)
        ctx.save_for_backward(x, weight)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = dy.contiguous()
        dx = dw = None
        if ctx.needs_input_grad[0]:
            Fr, Cin, H, W = x.shape
            dx = torch.empty_like(x)
            _gemm(weight, dy, dx, Fr, weight.shape[0], Cin, H * W, False)     # W read as [K=Cout][M=Cin]
        if ctx.needs_input_grad[1]:
            dw = torch.ops.aten.convolution_backward(
                dy, x, weight, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [False, True, False])[1]
        return dx, dw


def _eligible(conv, x):
    mode = pointwise_mode()
    if mode == "0" or not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4):
        return False
    if not (isinstance(conv, torch.nn.Conv2d) and conv.kernel_size == (1, 1) and conv.stride == (1, 1)
            and conv.padding == (0, 0) and conv.groups == 1 and conv.bias is None
            and conv.weight.dtype == torch.float32):
        return False
    P = x.shape[2] * x.shape[3]
    K, M = conv.in_channels, conv.out_channels
    if P % 4 or K % 2 or M % 2 or x.numel() == 0:      # kernel constraints (M even: it is K of the d(input) GEMM)
        return False
    if mode == "all":
        return True
    return P >= 3136 and K <= 128 and M <= 128           # measured win region (tools/pointwise_probe.py)


def conv1x1(conv, x):
    """`conv(x)` for a 1x1 nn.Conv2d module."""
    if not _eligible(conv, x):
        return conv(x)
    return _Conv1x1Func.apply(x.contiguous(), conv.weight)
