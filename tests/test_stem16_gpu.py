"""The stem under bf16 autocast (rk_stem16.hip: fp32 clip, bf16 activation): forward and d(weight) against the same convolution
evaluated by PyTorch in fp64 on the bf16-rounded operands."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


@pytest.mark.parametrize("Fr,C,H,W", [(16, 72, 224, 224), (3, 54, 32, 64), (5, 108, 16, 32), (2, 40, 48, 96), (2, 128, 32, 32)])
def test_stem_bf16_forward_and_weight_gradient(Fr, C, H, W):
    from rubiksnet_amd import pointwise

    g = torch.Generator(device="cpu").manual_seed(Fr + C + H)
    conv = torch.nn.Conv2d(3, C, 3, stride=2, padding=1, bias=False).to(DEV)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(C, 3, 3, 3, generator=g) * 0.2)
    x = torch.randn(Fr, 3, H, W, generator=g).to(DEV)
    gy = torch.randn(Fr, C, H // 2, W // 2, generator=g).to(DEV).bfloat16()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        assert pointwise._stem16_ok(conv, x)
        y = pointwise.stem_conv(conv, x)
    assert y.dtype == torch.bfloat16 and "Stem16" in type(y.grad_fn).__name__
    y.backward(gy)
    torch.cuda.synchronize()
    xd = x.bfloat16().double()
    wd = conv.weight.detach().bfloat16().double().requires_grad_(True)
    y_ref = F.conv2d(xd, wd, stride=2, padding=1)
    y_ref.backward(gy.double())
    a, b = y.detach().double().cpu(), y_ref.detach().cpu()
    tol = 2.0 ** -8 * b.abs() + 1e-3 * float(b.abs().max())                       # one rounding to bf16
    assert bool(((a - b).abs() <= tol).all()), "y: max err %.3e" % float((a - b).abs().max())
    dw, dw_ref = conv.weight.grad.double().cpu(), wd.grad.cpu()
    np.testing.assert_allclose(dw.numpy(), dw_ref.numpy(), rtol=0, atol=2e-5 * max(1.0, float(dw_ref.abs().max())))


def test_other_configurations_stay_on_aten():
    from rubiksnet_amd import pointwise

    conv = torch.nn.Conv2d(3, 72, 3, stride=2, padding=1, bias=False).to(DEV)
    x = torch.randn(2, 3, 30, 40, device=DEV)                    # W % 32 != 0
    with torch.autocast("cuda", dtype=torch.bfloat16):
        assert not pointwise._stem16_ok(conv, x)
        y = pointwise.stem_conv(conv, x)
    assert y.shape == (2, 72, 15, 20) and y.dtype == torch.bfloat16
    x = torch.randn(2, 3, 32, 64, device=DEV)
    assert not pointwise._stem16_ok(conv, x)                         # no autocast: the fp32 stem
