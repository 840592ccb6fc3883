"""1x1 convolutions of bf16 activations on planes without a 16-byte unit in a row (rk_pw16_odd.hip: 7x7, P = 49; also 5x5, 3x7,
9x7): forward (+ residual), d(input), d(weight) through `pointwise.conv1x1` under autocast against the same product evaluated
by PyTorch in fp64 on the bf16-rounded operands -- results within one bf16 rounding of the exact value, d(weight) (fp32) to
fp32 summation error -- and the stride-2 projecting shortcut (14x14 -> 7x7)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _ref(x, w, res, gy, stride):
    xd = x.double()[:, :, ::stride, ::stride].requires_grad_(False)
    xd = xd.clone().requires_grad_(True)
    wd = w.bfloat16().double().clone().requires_grad_(True)           # the kernels round the weight to bf16, as autocast does
    y = torch.einsum("mk,fkhw->fmhw", wd, xd)
    if res is not None:
        y = y + res.double()
    y.backward(gy.double())
    return y.detach(), xd.grad, wd.grad


@pytest.mark.parametrize("Fr,K,M,H,W,stride,res", [
    (256, 576, 576, 7, 7, 1, False), (256, 576, 576, 7, 7, 1, True), (256, 288, 576, 14, 14, 2, False),
    (5, 64, 96, 7, 7, 1, True), (3, 32, 64, 5, 5, 1, False), (7, 96, 320, 3, 7, 1, True), (2, 1152, 1152, 7, 7, 1, False),
    (9, 160, 64, 9, 7, 1, False), (4, 64, 96, 10, 14, 2, False), (3, 704, 32, 7, 7, 1, True)])
def test_conv1x1_bf16_on_odd_planes(Fr, K, M, H, W, stride, res):
    from rubiksnet_amd import pointwise

    g = torch.Generator(device="cpu").manual_seed(Fr + K + M + H)
    conv = torch.nn.Conv2d(K, M, 1, stride=stride, bias=False).to(DEV)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(M, K, 1, 1, generator=g) * (1.0 / K ** 0.5))
    x = torch.randn(Fr, K, H, W, generator=g).to(DEV).bfloat16().requires_grad_(True)
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    r = torch.randn(Fr, M, Ho, Wo, generator=g).to(DEV).bfloat16().requires_grad_(True) if res else None
    gy = torch.randn(Fr, M, Ho, Wo, generator=g).to(DEV).bfloat16()
    assert pointwise._eligible_odd16(conv, x, stride)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = pointwise.conv1x1(conv, x, residual=r)
    assert y.dtype == torch.bfloat16 and y.grad_fn is not None and "Odd16" in type(y.grad_fn).__name__
    y.backward(gy)
    torch.cuda.synchronize()
    y_ref, dx_ref, dw_ref = _ref(x.detach(), conv.weight.detach().view(M, K), r.detach() if res else None, gy, stride)

    def close_bf16(a, b, what):                                   # one rounding to bf16 of a value computed with fp32 accumulation
        a, b = a.double().cpu(), b.cpu()
        tol = 2.0 ** -8 * b.abs() + 1e-3 * float(b.abs().max())
        assert bool(((a - b).abs() <= tol).all()), "%s: max err %.3e" % (what, float((a - b).abs().max()))

    close_bf16(y.detach(), y_ref, "y")
    dx = x.grad[:, :, ::stride, ::stride]
    close_bf16(dx, dx_ref, "dx")
    if stride == 2:                                               # the odd pixels receive exact zeros
        mask = torch.ones_like(x.grad, dtype=torch.bool)
        mask[:, :, ::2, ::2] = False
        assert float(x.grad[mask].abs().max()) == 0.0
    if res:
        assert torch.equal(r.grad, gy)
    dw = conv.weight.grad.view(M, K).double().cpu()
    np.testing.assert_allclose(dw.numpy(), dw_ref.cpu().numpy(), rtol=0, atol=2e-5 * max(1.0, float(dw_ref.abs().max())))


def test_unsupported_shapes_stay_on_aten():
    from rubiksnet_amd import _native, pointwise

    L = _native.lib()
    assert L.rk_pw_odd16_supported(256, 576, 576, 49) == 1
    assert L.rk_pw_odd16_supported(256, 100, 576, 49) == 0          # K % 32
    assert L.rk_pw_odd16_supported(256, 576, 100, 49) == 0          # M % 8
    assert L.rk_pw_odd16_supported(256, 576, 576, 81) == 0          # P > 64
    conv = torch.nn.Conv2d(100, 64, 1, bias=False).to(DEV)
    x = torch.randn(2, 100, 7, 7, device=DEV).bfloat16()
    assert not pointwise._eligible_odd16(conv, x, 1)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = pointwise.conv1x1(conv, x)
    assert y.shape == (2, 64, 7, 7)


@pytest.mark.parametrize("Fr,K,M,H,W", [(8, 72, 144, 56, 56), (4, 64, 96, 28, 28), (6, 288, 576, 14, 14), (3, 32, 64, 10, 6), (2, 16, 32, 16, 24)])
def test_forked_shortcut_equals_separate_consumers(Fr, K, M, H, W):
    """pointwise.fork_shortcut: the activation of a downsampling block as ONE autograd node for its two consumers (main path,
    stride-2 projecting shortcut): same outputs, and d(activation) = the two gradients added -- bit for bit what autograd's own
    accumulation of the separate nodes' gradients gives (one fp32 add, one rounding)."""
    from rubiksnet_amd import pointwise

    g = torch.Generator(device="cpu").manual_seed(Fr + K + H)
    conv = torch.nn.Conv2d(K, M, 1, stride=2, bias=False).to(DEV)
    x0 = torch.randn(Fr, K, H, W, generator=g).to(DEV).bfloat16()
    gm = torch.randn(Fr, K, H, W, generator=g).to(DEV).bfloat16()
    gs = torch.randn(Fr, M, H // 2, W // 2, generator=g).to(DEV).bfloat16()
    res = []
    for forked in (True, False):
        conv.weight.grad = None
        x = x0.clone().requires_grad_(True)
        a = x * 1.0                                              # a non-leaf, as relu(bn1(.)) is
        with torch.autocast("cuda", dtype=torch.bfloat16):
            if forked:
                main, y = pointwise.fork_shortcut(conv, a)
                assert main is not a and "ForkS2" in type(main.grad_fn).__name__
            else:
                main, y = a, pointwise.conv1x1(conv, a)
        torch.autograd.backward([main, y], [gm, gs])
        res.append((y.detach().clone(), x.grad.clone(), conv.weight.grad.clone()))
    torch.cuda.synchronize()
    (yf, dxf, dwf), (yu, dxu, dwu) = res
    assert torch.equal(yf, yu) and torch.equal(dwf, dwu)
    assert torch.equal(dxf, dxu)
    # scatter alone (no gradient from the main path)
    x = x0.clone().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        main, y = pointwise.fork_shortcut(conv, x * 1.0)
    y.backward(gs)
    mask = torch.ones_like(x0, dtype=torch.bool)
    mask[:, :, ::2, ::2] = False
    assert float(x.grad[mask].abs().max()) == 0.0 and float(x.grad.abs().max()) > 0
