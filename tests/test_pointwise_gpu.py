"""GPU parity: the NCHW MFMA 1x1-convolution GEMM (rk_pw_gemm_f32, pointwise.conv1x1) against
torch.nn.functional.conv2d evaluated in fp64 on the CPU -- the op the reference's Conv1x1 layers run
(rubiksnet/backbone.py:44-45)."""
import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F


def _reload_switches():
    from rubiksnet_amd import config
    config.reload()

pytestmark = pytest.mark.gpu

CASES = [   # frames, Cin, Cout, H, W
    (3, 6, 10, 4, 4),          # tiny: one partial tile
    (5, 54, 54, 8, 8),         # K = 54 = 3 x 18
    (4, 24, 54, 12, 12),       # K = 24 = 2 x 12
    (2, 54, 108, 28, 28),      # M > 64: two waves along M
    (3, 216, 432, 6, 6),       # M > 256: two row tiles; K = 12 x 18
    (7, 64, 40, 10, 14),       # K = 4 x 16, ragged columns
    (2, 50, 22, 6, 6),         # K = 50: padded last chunk
    (3, 72, 144, 8, 8),        # Large's widths: the last 32-row block of a 64-row tile is all padding (skipped)
    (2, 144, 72, 8, 8),
    (3, 288, 288, 12, 12),     # channel counts that are multiples of 96: the 96 x 96-per-wave d(weight) kernel, ragged chunk
    (2, 288, 96, 8, 8),
    (1, 576, 192, 4, 8),       # two 288-row groups of X
]


@pytest.mark.parametrize("case", CASES)
def test_forward_and_gradients(monkeypatch, case):
    from rubiksnet_amd.pointwise import conv1x1

    monkeypatch.setenv("RK_PW", "all")

    _reload_switches()
    Fr, Cin, Cout, H, W = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(Fr, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 1, 1, generator=g) / Cin ** 0.5
    dy = torch.randn(Fr, Cout, H, W, generator=g)
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    yr = F.conv2d(xr, wr)
    yr.backward(dy.double())
    conv = nn.Conv2d(Cin, Cout, 1, bias=False).cuda()
    with torch.no_grad():
        conv.weight.copy_(w)
    xd = x.cuda().requires_grad_(True)
    y = conv1x1(conv, xd)
    assert y.grad_fn is not None and "Conv1x1Func" in type(y.grad_fn).__name__      # the HIP path ran
    y.backward(dy.cuda())
    for got, ref in ((y, yr), (xd.grad, xr.grad), (conv.weight.grad, wr.grad)):
        ref = ref.detach()
        np.testing.assert_allclose(got.detach().cpu().numpy(), ref.numpy(), rtol=0, atol=2e-6 * max(1.0, float(ref.abs().max())) * Cin ** 0.5)


@pytest.mark.parametrize("mode", ["all", "auto"])
def test_fused_residual(monkeypatch, mode):
    """conv3(out) + shortcut with the add in the GEMM epilogue (HIP GEMM) or after aten's convolution (auto mode on a
    small plane): same values and gradients as the unfused pair."""
    from rubiksnet_amd.pointwise import conv1x1

    monkeypatch.setenv("RK_PW", mode)

    _reload_switches()
    torch.manual_seed(1)
    conv = nn.Conv2d(12, 20, 1, bias=False).cuda()
    x = torch.randn(3, 12, 8, 8, device="cuda", requires_grad=True)
    r = torch.randn(3, 20, 8, 8, device="cuda", requires_grad=True)
    dy = torch.randn(3, 20, 8, 8, device="cuda")
    y = conv1x1(conv, x, residual=r)
    y.backward(dy)
    got = (y.detach(), x.grad.clone(), r.grad.clone(), conv.weight.grad.clone())
    x.grad = r.grad = conv.weight.grad = None
    y2 = conv(x) + r
    y2.backward(dy)
    for a, b in zip(got, (y2.detach(), x.grad, r.grad, conv.weight.grad)):
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=0, atol=1e-5 * max(1.0, float(b.abs().max())))


@pytest.mark.parametrize("case", [(5, 54, 54, 8, 8), (2, 54, 108, 28, 28), (3, 216, 432, 6, 6), (3, 72, 144, 8, 8)])
def test_bf16_activations_fp32_weight(monkeypatch, case):
    """bf16 storage (autocast): the fp32 weight is rounded to bf16 once per version (as autocast's cast would), bf16 MFMA
    with fp32 accumulation, outputs rounded once, d(weight) in fp32 -- checked against conv2d in fp64 on the bf16-rounded
    activations."""
    from rubiksnet_amd.pointwise import conv1x1

    monkeypatch.setenv("RK_PW", "all")

    _reload_switches()
    Fr, Cin, Cout, H, W = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(Fr, Cin, H, W, generator=g).bfloat16()
    r = torch.randn(Fr, Cout, H, W, generator=g).bfloat16()
    w = torch.randn(Cout, Cin, 1, 1, generator=g) / Cin ** 0.5
    dy = torch.randn(Fr, Cout, H, W, generator=g).bfloat16()
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    yr = F.conv2d(xr, wr) + r.double()
    yr.backward(dy.double())
    conv = nn.Conv2d(Cin, Cout, 1, bias=False).cuda()
    with torch.no_grad():
        conv.weight.copy_(w)
    xd = x.cuda().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = conv1x1(conv, xd, residual=r.cuda())
    assert y.dtype == torch.bfloat16 and "Conv1x1Func" in type(y.grad_fn).__name__
    y.backward(dy.cuda())
    assert conv.weight.grad.dtype == torch.float32 and xd.grad.dtype == torch.bfloat16
    np.testing.assert_allclose(y.float().detach().cpu().numpy(), yr.detach().numpy(), rtol=1e-2, atol=1e-2 * float(yr.abs().max()))
    np.testing.assert_allclose(xd.grad.float().cpu().numpy(), xr.grad.numpy(), rtol=1e-2, atol=1e-2 * float(xr.grad.abs().max()))
    np.testing.assert_allclose(conv.weight.grad.cpu().numpy(), wr.grad.numpy(), rtol=0, atol=2e-5 * float(wr.grad.abs().max()) * Cin ** 0.5)


_PW16 = [(8, 288, 288, 14, 14), (5, 72, 72, 12, 12), (3, 144, 144, 28, 28), (2, 70, 50, 6, 10), (4, 288, 576, 14, 14),
         (3, 576, 288, 14, 14), (1, 32, 16, 2, 4), (3, 40, 24, 3, 4), (2, 64, 64, 4, 5), (7, 100, 330, 6, 6)]


@pytest.mark.parametrize("case", _PW16)
def test_packed_bf16_gemm_against_fp64(case):
    """rk_pw_pack_bf16 + rk_pw_gemm_packed_bf16 (rk_pw16.hip) through the C ABI: forward operand, d(input) operand (W^T) and
    the residual, against the fp64 product of the bf16-rounded operands; the result may differ from it by the one rounding
    to bf16 (+ the fp32 accumulation).  Shapes: 14x14 planes (P % 8 == 4: a frame's last 16-byte unit overlaps the one
    before), ragged channel counts, more rows than one workgroup takes (576), tiles that end inside a frame."""
    from rubiksnet_amd import _native

    L = _native.lib()
    st = torch.cuda.current_stream().cuda_stream
    Fr, K, M, H, W = case
    P = H * W
    g = torch.Generator().manual_seed(7 * K + M)
    x = torch.randn(Fr, K, P, generator=g).bfloat16()
    r = torch.randn(Fr, M, P, generator=g).bfloat16()
    w = torch.randn(M, K, generator=g) / K ** 0.5
    wq = w.bfloat16().double()
    xd, rd, wd = x.cuda(), r.cuda(), w.cuda()
    fwd = torch.empty(int(L.rk_pw_packed_bytes(M, K)), dtype=torch.uint8, device="cuda")
    bwd = torch.empty(int(L.rk_pw_packed_bytes(K, M)), dtype=torch.uint8, device="cuda")
    _native.check(L.rk_pw_pack_bf16(wd.data_ptr(), M, K, fwd.data_ptr(), bwd.data_ptr(), st), "pack")

    def close(got, ref):
        err = (got.double().cpu() - ref).abs()
        bound = ref.abs() * 2.0 ** -8 + 1e-5 * float(ref.abs().max())           # one bf16 rounding + fp32 accumulation
        assert bool((err <= bound).all()), float((err - bound).max())

    for res in (None, rd):
        y = torch.full((Fr, M, P), float("nan"), dtype=torch.bfloat16, device="cuda")
        _native.check(L.rk_pw_gemm_packed_bf16(fwd.data_ptr(), xd.data_ptr(), res.data_ptr() if res is not None else None,
                                               y.data_ptr(), Fr, K, M, P, st), "gemm")
        ref = torch.einsum("mk,fkp->fmp", wq, x.double())
        close(y, ref + r.double() if res is not None else ref)
    dx = torch.full((Fr, K, P), float("nan"), dtype=torch.bfloat16, device="cuda")       # d(input): W^T applied to [F, M, P]
    _native.check(L.rk_pw_gemm_packed_bf16(bwd.data_ptr(), rd.data_ptr(), None, dx.data_ptr(), Fr, M, K, P, st), "dgrad")
    close(dx, torch.einsum("mk,fmp->fkp", wq, r.double()))
    # in place: R = Y
    y = rd.clone()
    _native.check(L.rk_pw_gemm_packed_bf16(fwd.data_ptr(), xd.data_ptr(), y.data_ptr(), y.data_ptr(), Fr, K, M, P, st), "gemm")
    close(y, torch.einsum("mk,fkp->fmp", wq, x.double()) + r.double())


@pytest.mark.parametrize("case", _PW16 + [(256, 288, 288, 14, 14), (32, 72, 144, 56, 56)])
def test_bf16_wgrad16_against_fp64(case):
    """rk_pw_wgrad16_bf16: d(weight) of bf16 operands, fp32 accumulation, against the fp64 sum (incl. the full-size Large-AQ
    layer: 73 splits x 4 output tiles, and a frame's overlapping last unit counted once)."""
    from rubiksnet_amd import _native

    L = _native.lib()
    st = torch.cuda.current_stream().cuda_stream
    Fr, K, M, H, W = case
    P = H * W
    g = torch.Generator().manual_seed(3 * K + M)
    x = torch.randn(Fr, K, P, generator=g).bfloat16().cuda()
    dy = torch.randn(Fr, M, P, generator=g).bfloat16().cuda()
    nb = int(L.rk_pw_wgrad16_workspace_bytes(Fr, K, M, P))
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    dw = torch.full((M, K), float("nan"), device="cuda")
    _native.check(L.rk_pw_wgrad16_bf16(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), Fr, K, M, P, ws.data_ptr(), nb, st), "wgrad16")
    ref = torch.einsum("fmp,fkp->mk", dy.double(), x.double()).cpu()
    np.testing.assert_allclose(dw.double().cpu().numpy(), ref.numpy(), rtol=0, atol=1e-5 * float(ref.abs().max()))
    # deterministic: same bits on a second launch
    dw2 = torch.empty_like(dw)
    _native.check(L.rk_pw_wgrad16_bf16(dy.data_ptr(), x.data_ptr(), dw2.data_ptr(), Fr, K, M, P, ws.data_ptr(), nb, st), "wgrad16")
    assert torch.equal(dw, dw2)
    assert L.rk_pw_wgrad16_bf16(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), Fr, K, M, P, ws.data_ptr(), nb - 1, st) != 0   # workspace


@pytest.mark.parametrize("case", [(4, 72, 144, 8, 8), (3, 24, 24, 12, 8), (2, 144, 288, 28, 28)])
def test_bf16_strided_shortcut(monkeypatch, case):
    """1x1 / stride-2 convolution on bf16 activations: even pixels gathered, stride-1 bf16 kernels on the quarter-size tensor,
    d(input) scattered into zeros -- against conv2d(stride=2) in fp64."""
    from rubiksnet_amd.pointwise import conv1x1

    monkeypatch.setenv("RK_PW", "all")
    _reload_switches()
    Fr, Cin, Cout, H, W = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(Fr, Cin, H, W, generator=g).bfloat16()
    w = torch.randn(Cout, Cin, 1, 1, generator=g) / Cin ** 0.5
    dy = torch.randn(Fr, Cout, H // 2, W // 2, generator=g).bfloat16()
    xr, wr = x.double().requires_grad_(True), w.bfloat16().double().requires_grad_(True)
    yr = F.conv2d(xr, wr, stride=2)
    yr.backward(dy.double())
    conv = nn.Conv2d(Cin, Cout, 1, stride=2, bias=False).cuda()
    with torch.no_grad():
        conv.weight.copy_(w)
    xd = x.cuda().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = conv1x1(conv, xd)
    assert y.dtype == torch.bfloat16 and "ConvS2Bf16Func" in type(y.grad_fn).__name__
    y.backward(dy.cuda())
    tol = lambda t: 2.0 ** -7 * float(t.abs().max())
    np.testing.assert_allclose(y.float().detach().cpu().numpy(), yr.detach().numpy(), rtol=0, atol=tol(yr))
    np.testing.assert_allclose(xd.grad.float().cpu().numpy(), xr.grad.numpy(), rtol=0, atol=tol(xr.grad))
    assert float(xd.grad[:, :, 1::2, :].abs().max()) == 0.0 and float(xd.grad[:, :, :, 1::2].abs().max()) == 0.0
    np.testing.assert_allclose(conv.weight.grad.cpu().numpy(), wr.grad.numpy(), rtol=0, atol=1e-5 * float(wr.grad.abs().max()))


def test_pw16_randomised_shapes():
    """Seeded sweep over (frames, channels in / out, plane) for the bf16 GEMM (+ residual, d(input)) and d(weight) kernels:
    odd channel counts, planes with P % 8 == 4 and P % 8 == 0, tiles ending inside frames, more rows than one workgroup
    takes.  RK_SWEEP_PW16 sets the number of draws (default 40)."""
    import os
    from rubiksnet_amd import _native

    L = _native.lib()
    st = torch.cuda.current_stream().cuda_stream
    rng = np.random.default_rng(20260929)
    draws = int(os.environ.get("RK_SWEEP_PW16", "40"))
    for it in range(draws):
        Fr = int(rng.integers(1, 24))
        K, M = int(rng.integers(1, 330)), int(rng.integers(1, 330))
        if it % 7 == 0:
            M = int(rng.integers(300, 620))
        H, W = int(rng.integers(1, 15)), 4 * int(rng.integers(1, 8))
        if rng.integers(0, 2):
            H, W = W, H
        P = H * W
        if P < 8 or P % 4:
            continue
        g = torch.Generator().manual_seed(it)
        x = torch.randn(Fr, K, P, generator=g).bfloat16()
        r = torch.randn(Fr, M, P, generator=g).bfloat16()
        w = torch.randn(M, K, generator=g) / K ** 0.5
        wq = w.bfloat16().double()
        xd, rd, wd = x.cuda(), r.cuda(), w.cuda()
        fwd = torch.empty(int(L.rk_pw_packed_bytes(M, K)), dtype=torch.uint8, device="cuda")
        bwd = torch.empty(int(L.rk_pw_packed_bytes(K, M)), dtype=torch.uint8, device="cuda")
        _native.check(L.rk_pw_pack_bf16(wd.data_ptr(), M, K, fwd.data_ptr(), bwd.data_ptr(), st), "pack")
        tag = "draw %d: F %d K %d M %d plane %dx%d" % (it, Fr, K, M, H, W)

        def close(got, ref, what):
            err = (got.double().cpu() - ref).abs()
            bound = ref.abs() * 2.0 ** -8 + 1e-5 * float(ref.abs().max())
            assert bool((err <= bound).all()), "%s %s: %g" % (tag, what, float((err - bound).max()))

        y = torch.full((Fr, M, P), float("nan"), dtype=torch.bfloat16, device="cuda")
        _native.check(L.rk_pw_gemm_packed_bf16(fwd.data_ptr(), xd.data_ptr(), rd.data_ptr(), y.data_ptr(), Fr, K, M, P, st), tag)
        close(y, torch.einsum("mk,fkp->fmp", wq, x.double()) + r.double(), "forward + residual")
        dx = torch.full((Fr, K, P), float("nan"), dtype=torch.bfloat16, device="cuda")
        _native.check(L.rk_pw_gemm_packed_bf16(bwd.data_ptr(), rd.data_ptr(), None, dx.data_ptr(), Fr, M, K, P, st), tag)
        close(dx, torch.einsum("mk,fmp->fkp", wq, r.double()), "d(input)")
        nb = int(L.rk_pw_wgrad16_workspace_bytes(Fr, K, M, P))
        ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
        dw = torch.full((M, K), float("nan"), device="cuda")
        _native.check(L.rk_pw_wgrad16_bf16(rd.data_ptr(), xd.data_ptr(), dw.data_ptr(), Fr, K, M, P, ws.data_ptr(), nb, st), tag)
        ref = torch.einsum("fmp,fkp->mk", r.double(), x.double())
        np.testing.assert_allclose(dw.double().cpu().numpy(), ref.numpy(), rtol=0, atol=1e-5 * float(ref.abs().max()) + 1e-12,
                                   err_msg=tag + " d(weight)")


def test_packed_weight_follows_the_parameter(monkeypatch):
    """The packed copy of a weight is made per forward: in-place edits are seen, also those made through `.data` (which no
    version counter records)."""
    from rubiksnet_amd.pointwise import conv1x1

    monkeypatch.setenv("RK_PW", "all")
    _reload_switches()
    torch.manual_seed(0)
    conv = nn.Conv2d(16, 32, 1, bias=False).cuda()
    x = torch.randn(2, 16, 4, 4, device="cuda").bfloat16()
    y1 = conv1x1(conv, x).detach().float()
    with torch.no_grad():
        conv.weight.mul_(2.0)
    y2 = conv1x1(conv, x).detach().float()
    np.testing.assert_allclose(y2.cpu().numpy(), 2 * y1.cpu().numpy(), rtol=2 ** -7, atol=1e-6)
    conv.weight.data.mul_(2.0)
    y4 = conv1x1(conv, x).detach().float()
    np.testing.assert_allclose(y4.cpu().numpy(), 4 * y1.cpu().numpy(), rtol=2 ** -7, atol=1e-6)


def test_ineligible_layers_take_the_stock_path(monkeypatch):
    from rubiksnet_amd.pointwise import conv1x1

    monkeypatch.setenv("RK_PW", "all")

    _reload_switches()
    x = torch.randn(2, 6, 7, 7, device="cuda", requires_grad=True)          # 49 pixels: P % 4 != 0
    conv = nn.Conv2d(6, 8, 1, bias=False).cuda()
    y = conv1x1(conv, x)
    assert "Conv1x1Func" not in type(y.grad_fn).__name__
    assert torch.equal(y, conv(x))
    s2 = nn.Conv2d(6, 8, 1, stride=2, bias=False).cuda()                    # strided shortcut
    assert torch.equal(conv1x1(s2, x), s2(x))
    monkeypatch.setenv("RK_PW", "0")
    _reload_switches()
    x4 = torch.randn(2, 6, 8, 8, device="cuda", requires_grad=True)
    assert "Conv1x1Func" not in type(conv1x1(conv, x4).grad_fn).__name__


ODD_CASES = [   # frames, Cin, Cout, H, W  (H * W % 4 != 0)
    (5, 8, 12, 7, 7),          # one partial tile, partial last chunk of channels
    (13, 432, 432, 7, 7),      # layer4 of Tiny: 7 row tiles, 3 column tiles, the last one ragged
    (16, 216, 100, 7, 7),
    (6, 20, 36, 9, 7),         # 63 pixels per plane
    (40, 16, 16, 7, 7),        # several 256-column tiles, frames straddling them
]


@pytest.mark.parametrize("with_residual", [False, True])
@pytest.mark.parametrize("case", ODD_CASES)
def test_odd_planes_forward_and_gradients(case, with_residual):
    """rk_pw_gemm_odd_f32 / rk_pw_wgrad_odd_f32 (7x7 planes: H * W % 4 != 0) against F.conv2d in fp64: forward (+ the
    residual in the epilogue), d(input), d(weight)."""
    from rubiksnet_amd.pointwise import conv1x1

    Fr, Cin, Cout, H, W = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(Fr, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 1, 1, generator=g) / Cin ** 0.5
    dy = torch.randn(Fr, Cout, H, W, generator=g)
    res = torch.randn(Fr, Cout, H, W, generator=g) if with_residual else None
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    rr = res.double().requires_grad_(True) if with_residual else None
    yr = F.conv2d(xr, wr) + (rr if with_residual else 0)
    yr.backward(dy.double())
    conv = nn.Conv2d(Cin, Cout, 1, bias=False).cuda()
    with torch.no_grad():
        conv.weight.copy_(w)
    xd = x.cuda().requires_grad_(True)
    rd = res.cuda().requires_grad_(True) if with_residual else None
    y = conv1x1(conv, xd, rd)
    assert "Conv1x1OddFunc" in type(y.grad_fn).__name__
    y.backward(dy.cuda())
    pairs = [(y, yr), (xd.grad, xr.grad), (conv.weight.grad, wr.grad)] + ([(rd.grad, rr.grad)] if with_residual else [])
    for got, ref in pairs:
        ref = ref.detach()
        np.testing.assert_allclose(got.detach().cpu().numpy(), ref.numpy(), rtol=0,
                                   atol=3e-6 * max(1.0, float(ref.abs().max())) * max(Cin, Cout) ** 0.5)


@pytest.mark.parametrize("shape", [(3, 8, 14, 14, 12), (9, 216, 14, 14, 432), (5, 20, 18, 14, 24)])
def test_strided_shortcut_onto_odd_planes(shape):
    """The 14x14 -> 7x7 projecting shortcut (rk_pw_s2_*_odd_f32) against F.conv2d(stride=2) in fp64."""
    from rubiksnet_amd.pointwise import conv1x1

    Fr, Cin, H, W, Cout = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(Fr, Cin, H, W, generator=g)
    conv = nn.Conv2d(Cin, Cout, 1, stride=2, bias=False)
    dy = torch.randn(Fr, Cout, H // 2, W // 2, generator=g)
    xr = x.double().requires_grad_(True)
    wr = conv.weight.detach().double().requires_grad_(True)
    ref = F.conv2d(xr, wr, stride=2)
    ref.backward(dy.double())
    conv = conv.cuda()
    xg = x.cuda().requires_grad_(True)
    y = conv1x1(conv, xg)
    assert "ConvS2OddFunc" in type(y.grad_fn).__name__
    np.testing.assert_allclose(y.detach().cpu().numpy(), ref.detach().numpy(), rtol=0,
                               atol=3e-6 * Cin ** 0.5 * float(ref.abs().max()))
    y.backward(dy.cuda())
    np.testing.assert_allclose(xg.grad.cpu().numpy(), xr.grad.numpy(), rtol=0,
                               atol=3e-6 * Cout ** 0.5 * float(xr.grad.abs().max()))
    assert float(xg.grad[:, :, 1::2, :].abs().max()) == 0.0 and float(xg.grad[:, :, :, 1::2].abs().max()) == 0.0
    np.testing.assert_allclose(conv.weight.grad.cpu().numpy(), wr.grad.numpy(), rtol=0,
                               atol=1e-4 * float(wr.grad.abs().max()))


@pytest.mark.parametrize("shape", [(3, 3, 32, 40, 54), (2, 3, 224, 224, 72), (5, 2, 18, 16, 20), (2, 7, 8, 8, 150)])
def test_stem_conv(shape):
    """The 3x3 / stride-2 / pad-1 first layer on the MFMA GEMM (im2col gathered on the fly) against F.conv2d in fp64;
    d(weight) on the d(weight) kernel with the same gather (rk_stem_wgrad3x3s2_f32)."""
    from rubiksnet_amd.pointwise import stem_conv

    Fr, Cin, H, W, Cout = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(Fr, Cin, H, W, generator=g)
    conv = nn.Conv2d(Cin, Cout, 3, stride=2, padding=1, bias=False)
    dy = torch.randn(Fr, Cout, H // 2, W // 2, generator=g)
    ref = F.conv2d(x.double(), conv.weight.detach().double(), stride=2, padding=1)
    conv = conv.cuda()
    y = stem_conv(conv, x.cuda())
    assert "StemFunc" in type(y.grad_fn).__name__
    np.testing.assert_allclose(y.detach().cpu().numpy(), ref.numpy(), rtol=0, atol=3e-6 * float(ref.abs().max()) * (9 * Cin) ** 0.5)
    y.backward(dy.cuda())
    wr = conv.weight.detach().double().cpu().requires_grad_(True)
    F.conv2d(x.double(), wr, stride=2, padding=1).backward(dy.double())
    np.testing.assert_allclose(conv.weight.grad.cpu().numpy(), wr.grad.numpy(), rtol=0, atol=1e-4 * float(wr.grad.abs().max()))


@pytest.mark.parametrize("shape", [(3, 6, 16, 24, 10), (2, 54, 112, 112, 54), (4, 54, 56, 56, 108), (5, 108, 28, 24, 216),
                                   (2, 144, 28, 32, 288), (5, 108, 28, 28, 216), (3, 10, 12, 20, 6)])   # Wo = 14, 10: groups wrap rows
def test_strided_shortcut_conv(shape):
    """1x1 / stride-2 convolution (projecting shortcut) on the GEMM kernels -- strided gather forward, scattered
    d(input) with every element written, gathered d(weight) -- against F.conv2d in fp64."""
    from rubiksnet_amd.pointwise import conv1x1

    Fr, Cin, H, W, Cout = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(Fr, Cin, H, W, generator=g)
    conv = nn.Conv2d(Cin, Cout, 1, stride=2, bias=False)
    dy = torch.randn(Fr, Cout, H // 2, W // 2, generator=g)
    xr = x.double().requires_grad_(True)
    wr = conv.weight.detach().double().requires_grad_(True)
    ref = F.conv2d(xr, wr, stride=2)
    ref.backward(dy.double())
    conv = conv.cuda()
    xg = x.cuda().requires_grad_(True)
    y = conv1x1(conv, xg)
    assert "ConvS2Func" in type(y.grad_fn).__name__
    tol = 3e-6 * Cin ** 0.5
    np.testing.assert_allclose(y.detach().cpu().numpy(), ref.detach().numpy(), rtol=0, atol=tol * float(ref.abs().max()))
    dx_poison = torch.full_like(xg, float("nan"))          # d(input) must be written everywhere (no memset relied on)
    y.backward(dy.cuda())
    del dx_poison
    np.testing.assert_allclose(xg.grad.cpu().numpy(), xr.grad.numpy(), rtol=0,
                               atol=3e-6 * Cout ** 0.5 * float(xr.grad.abs().max()))
    assert float(xg.grad[:, :, 1::2, :].abs().max()) == 0.0 and float(xg.grad[:, :, :, 1::2].abs().max()) == 0.0
    np.testing.assert_allclose(conv.weight.grad.cpu().numpy(), wr.grad.numpy(), rtol=0,
                               atol=1e-4 * float(wr.grad.abs().max()))

