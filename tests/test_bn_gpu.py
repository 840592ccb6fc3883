"""GPU parity: fused BatchNorm2d(+ReLU) (rk_bn_relu_*) versus torch's own batch_norm + relu on the CPU in fp64 --
the pair the reference's blocks are built from (rubiksnet/backbone.py:50-53, :129-131, :196)."""
import copy

import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F


def _reload_switches():
    from rubiksnet_amd import config
    config.reload()

pytestmark = pytest.mark.gpu

SHAPES = [
    (8, 6, 12, 12),      # P % 4 == 0, one frame group
    (5, 3, 7, 7),        # 7x7 planes: element-wise path
    (20, 54, 28, 28),    # two frame groups, the second one ragged
    (3, 4, 56, 56),      # one frame per group
    (1, 2, 2, 2),
]


def _make(shape, seed, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    C = shape[1]
    x = (torch.randn(shape, generator=g) * 1.7 + torch.randn(1, C, 1, 1, generator=g) * 3).to(dtype)
    dy = torch.randn(shape, generator=g).to(dtype)
    bn = nn.BatchNorm2d(C)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(C, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(C, generator=g) * 0.3)
        bn.running_mean.copy_(torch.randn(C, generator=g))
        bn.running_var.copy_(torch.rand(C, generator=g) + 0.5)
    return x, dy, bn


def _reference(bn, x, dy, relu, training):
    """torch CPU, fp64: F.batch_norm + relu, the ops nn.BatchNorm2d / nn.ReLU run."""
    ref = copy.deepcopy(bn).double()
    ref.train(training)
    xr = x.double().requires_grad_(True)
    y = ref(xr)
    if relu:
        y = F.relu(y)
    y.backward(dy.double())
    return y.detach(), xr.grad, ref.weight.grad, ref.bias.grad, ref


@pytest.mark.parametrize("relu", [True, False])
@pytest.mark.parametrize("shape", SHAPES)
def test_training_forward_backward_and_running_stats(shape, relu):
    from rubiksnet_amd.fused_bn import bn_relu

    x, dy, bn = _make(shape, seed=sum(shape))
    y_ref, dx_ref, dw_ref, db_ref, ref = _reference(bn, x, dy, relu, True)
    dev = copy.deepcopy(bn).cuda().train()
    xd = x.cuda().requires_grad_(True)
    y = bn_relu(dev, xd, relu=relu)
    y.backward(dy.cuda())
    tol = dict(rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(y.detach().cpu().numpy(), y_ref.numpy(), **tol)
    scale = max(1.0, float(dx_ref.abs().max()))
    np.testing.assert_allclose(xd.grad.cpu().numpy(), dx_ref.numpy(), rtol=0, atol=2e-5 * scale)
    np.testing.assert_allclose(dev.weight.grad.cpu().numpy(), dw_ref.numpy(), rtol=1e-4, atol=1e-4 * max(1.0, float(dw_ref.abs().max())))
    np.testing.assert_allclose(dev.bias.grad.cpu().numpy(), db_ref.numpy(), rtol=1e-4, atol=1e-4 * max(1.0, float(db_ref.abs().max())))
    np.testing.assert_allclose(dev.running_mean.cpu().numpy(), ref.running_mean.numpy(), **tol)
    np.testing.assert_allclose(dev.running_var.cpu().numpy(), ref.running_var.numpy(), **tol)
    assert int(dev.num_batches_tracked) == int(ref.num_batches_tracked) == 1


@pytest.mark.parametrize("shape", SHAPES[:3])
def test_eval_forward_uses_running_statistics(shape):
    from rubiksnet_amd.fused_bn import bn_relu

    x, dy, bn = _make(shape, seed=7)
    dev = copy.deepcopy(bn).cuda().eval()
    with torch.no_grad():
        y = bn_relu(dev, x.cuda())
        y_ref = F.relu(bn.double().eval()(x.double()))
    np.testing.assert_allclose(y.cpu().numpy(), y_ref.numpy(), rtol=1e-5, atol=1e-5)
    assert int(dev.num_batches_tracked) == 0


def test_cumulative_average_and_no_running_stats():
    from rubiksnet_amd.fused_bn import bn_relu

    x, dy, _ = _make((6, 4, 8, 8), seed=1)
    for kwargs in ({"momentum": None}, {"track_running_stats": False}):
        ref = nn.BatchNorm2d(4, **kwargs).double()
        dev = nn.BatchNorm2d(4, **kwargs).cuda()
        for _ in range(2):
            y_ref = F.relu(ref(x.double()))
            y = bn_relu(dev, x.cuda())
        np.testing.assert_allclose(y.detach().cpu().numpy(), y_ref.detach().numpy(), rtol=1e-5, atol=1e-5)
        if ref.running_mean is not None:
            np.testing.assert_allclose(dev.running_mean.cpu().numpy(), ref.running_mean.numpy(), rtol=1e-5, atol=1e-6)
            np.testing.assert_allclose(dev.running_var.cpu().numpy(), ref.running_var.numpy(), rtol=1e-5, atol=1e-6)


def test_large_mean_does_not_cancel():
    """|mean| >> std: E[x^2] - mean^2 in fp32 would lose the variance; the shifted sums do not."""
    from rubiksnet_amd.fused_bn import bn_relu

    g = torch.Generator().manual_seed(0)
    x = torch.randn(16, 3, 14, 14, generator=g) * 0.01 + 300.0
    bn = nn.BatchNorm2d(3)
    y_ref = F.relu(copy.deepcopy(bn).double()(x.double()))
    y = bn_relu(bn.cuda(), x.cuda())
    np.testing.assert_allclose(y.detach().cpu().numpy(), y_ref.detach().numpy(), rtol=0, atol=2e-2)   # x itself carries 3e-5 / 0.01 of noise


def test_bf16_storage():
    from rubiksnet_amd.fused_bn import bn_relu

    x, dy, bn = _make((8, 6, 12, 12), seed=3, dtype=torch.bfloat16)
    y_ref, dx_ref, dw_ref, db_ref, _ = _reference(bn, x.float(), dy.float(), True, True)
    dev = copy.deepcopy(bn).cuda().train()
    xd = x.cuda().requires_grad_(True)
    y = bn_relu(dev, xd)
    assert y.dtype == torch.bfloat16
    y.backward(dy.cuda())
    np.testing.assert_allclose(y.float().detach().cpu().numpy(), y_ref.numpy(), rtol=1e-2, atol=1e-2)
    np.testing.assert_allclose(xd.grad.float().cpu().numpy(), dx_ref.numpy(), rtol=2e-2, atol=2e-2)
    np.testing.assert_allclose(dev.weight.grad.cpu().numpy(), dw_ref.numpy(), rtol=1e-3, atol=1e-2)
    np.testing.assert_allclose(dev.bias.grad.cpu().numpy(), db_ref.numpy(), rtol=1e-3, atol=1e-2)


def test_block_matches_stock_batchnorm(monkeypatch):
    """A whole RubiksShift block, fused BN+ReLU against the stock nn.BatchNorm2d + ReLU pair."""
    from rubiksnet_amd import RubiksNet, fused_bn

    torch.manual_seed(0)
    net = RubiksNet("tiny", 11, verbose=False).cuda().train()
    block = net.backbone.layer1[0]
    x = torch.randn(16, block.bn1.num_features, 28, 28, device="cuda")
    outs = []
    for enabled in (True, False):
        monkeypatch.setenv("RK_FUSED_BN", "1" if enabled else "0")
        _reload_switches()
        assert fused_bn.fused_bn_enabled() is enabled
        blk = copy.deepcopy(block)
        xi = x.clone().requires_grad_(True)
        y = blk(xi)
        y.square().mean().backward()
        outs.append((y.detach(), xi.grad, blk.bn2.weight.grad, blk.bn1.running_var.clone()))
    for a, b in zip(*outs):
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=2e-4, atol=2e-5 * max(1.0, float(b.abs().max())))


def test_skip_gradient_is_added_inside_the_bn_backward():
    """bn_relu_skip: (relu(bn(x)), alias of x); the alias' gradient joins d(x) inside the BN backward kernel."""
    from rubiksnet_amd.fused_bn import bn_relu_skip

    x, dy, bn = _make((6, 10, 12, 12), seed=11)
    g = torch.Generator().manual_seed(12)
    dskip = torch.randn(x.shape, generator=g)
    ref = copy.deepcopy(bn).double().train()
    xr = x.double().requires_grad_(True)
    yr = F.relu(ref(xr))
    (yr * dy.double()).sum().backward(retain_graph=True)
    (xr * dskip.double()).sum().backward()
    dev = copy.deepcopy(bn).cuda().train()
    xd = x.cuda().requires_grad_(True)
    y, skip = bn_relu_skip(dev, xd)
    assert "BNReLUTrain" in type(skip.grad_fn).__name__
    ((y * dy.cuda()).sum() + (skip * dskip.cuda()).sum()).backward()
    np.testing.assert_allclose(y.detach().cpu().numpy(), yr.detach().numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(xd.grad.cpu().numpy(), xr.grad.numpy(), rtol=0, atol=2e-5 * float(xr.grad.abs().max()))
    np.testing.assert_allclose(dev.weight.grad.cpu().numpy(), ref.weight.grad.numpy(), rtol=1e-4, atol=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(16, 24, 6, 10, 8), (8, 72, 14, 14, 8), (12, 10, 5, 3, 4), (16, 288, 14, 14, 8)])
def test_bn_relu_folded_into_the_temporal_filter(dtype, shape):
    """bn_relu_tshift_skip (the -aq block's training-mode bn1 + ReLU inside its AttentionShift: rk_bn_stats_finish_*,
    rk_tshift3_bn_*, rk_bn_bwd_finish_tiles_f32, rk_bn_bwd_dx_pre_*) against the unfused pair bn_relu_skip + AttentionShift:
    outputs, every gradient (incl. the identity shortcut's, the taps' and BatchNorm's), running statistics, the counter."""
    from rubiksnet_amd.attention_shift import AttentionShift
    from rubiksnet_amd.fused_bn import bn_relu_skip, bn_relu_tshift_skip

    NT, C, H, W, S = shape
    torch.manual_seed(sum(shape))
    x0 = (torch.randn(NT, C, H, W, device="cuda") * 1.5 + 0.3).to(dtype)
    gy = torch.randn(NT, C, H, W, device="cuda").to(dtype)
    gs = torch.randn(NT, C, H, W, device="cuda").to(dtype)
    res = []
    for fused in (True, False):
        torch.manual_seed(7)
        bn = nn.BatchNorm2d(C).cuda().train()
        shift = AttentionShift(S, C).cuda()
        with torch.no_grad():
            bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(0, 0.2)
        x = x0.clone().requires_grad_(True)
        if fused:
            r = bn_relu_tshift_skip(bn, shift, x)
            assert r is not None
            y, skip = r
        else:
            a, skip = bn_relu_skip(bn, x)
            y = shift(a)
        ((y.float() * gy.float()).sum() + 0.01 * (skip.float() * gs.float()).sum()).backward()
        res.append((y.detach().float(), x.grad.float(), bn.weight.grad.clone(), bn.bias.grad.clone(), shift.weight.grad.clone(),
                    bn.running_mean.clone(), bn.running_var.clone(), int(bn.num_batches_tracked)))
    f, u = res
    assert f[7] == u[7] == 1
    tol = 2e-5 if dtype == torch.float32 else 2.0 ** -6        # bf16: the unfused pair rounds the activation and d(activation) once more
    for name, a, b in zip(("y", "dx", "dgamma", "dbeta", "dtaps", "running_mean", "running_var"), f[:7], u[:7]):
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=0, atol=tol * max(1e-3, float(b.abs().max())), err_msg=name)


@pytest.mark.gpu
def test_bn_relu_temporal_filter_randomised_shapes():
    """Seeded sweep of bn_relu_tshift_skip against fp64 PyTorch (batch-norm, ReLU, 3-tap temporal filter, autograd): every
    vector width of the filter kernels (H*W odd, % 2, % 4, % 8), n_segment 1 .. 8, fp32 and bf16."""
    import os
    from rubiksnet_amd.attention_shift import AttentionShift
    from rubiksnet_amd.fused_bn import bn_relu_tshift_skip

    rng = np.random.default_rng(7)
    for it in range(int(os.environ.get("RK_SWEEP_BNT", "24"))):
        S = int(rng.integers(1, 9))
        NT, C = S * int(rng.integers(1, 5)), int(rng.integers(1, 40))
        H, W = int(rng.integers(1, 9)), int(rng.integers(1, 9))
        dtype = torch.bfloat16 if it % 2 else torch.float32
        torch.manual_seed(it)
        bn = nn.BatchNorm2d(C).cuda().train()
        shift = AttentionShift(S, C).cuda()
        with torch.no_grad():
            bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(0, 0.2)
        x = (torch.randn(NT, C, H, W, device="cuda") * 1.5 + 0.3).to(dtype).requires_grad_(True)
        gy = torch.randn(NT, C, H, W, device="cuda").to(dtype)
        if NT * H * W < 2:
            continue
        r = bn_relu_tshift_skip(bn, shift, x)
        assert r is not None
        (r[0].float() * gy.float()).sum().backward()
        soft = shift.soft_taps().detach().double().cpu()
        x64 = x.detach().double().cpu().requires_grad_(True)
        w, b = bn.weight.detach().double().cpu().requires_grad_(True), bn.bias.detach().double().cpu().requires_grad_(True)
        a = F.relu(F.batch_norm(x64, None, None, w, b, True, 0.0, bn.eps))
        a5 = a.view(NT // S, S, C, H, W)
        pad = torch.zeros_like(a5[:, :1])
        s0, s1, s2 = (soft[:, j].view(1, 1, C, 1, 1) for j in range(3))
        y = (s0 * torch.cat([pad, a5[:, :-1]], 1) + s1 * a5 + s2 * torch.cat([a5[:, 1:], pad], 1)).view(NT, C, H, W)
        y.backward(gy.double().cpu())
        tag = "draw %d: NT %d S %d C %d plane %dx%d %s" % (it, NT, S, C, H, W, dtype)
        bar = 2.0 ** -6 if dtype == torch.bfloat16 else 2e-5
        for name, got, ref in (("y", r[0].detach(), y.detach()), ("dx", x.grad, x64.grad), ("dgamma", bn.weight.grad, w.grad),
                               ("dbeta", bn.bias.grad, b.grad)):
            np.testing.assert_allclose(got.double().cpu().numpy(), ref.numpy(), rtol=0, atol=bar * max(1e-3, float(ref.abs().max())),
                                       err_msg=tag + " " + name)


@pytest.mark.gpu
@pytest.mark.parametrize("skip", [False, True])
@pytest.mark.parametrize("inplace", [False, True])
@pytest.mark.parametrize("shape", [(8, 6, 196), (256, 288, 196), (13, 34, 196), (10, 6, 3136), (3, 5, 784), (7, 3, 12), (5, 2, 8),
                                   (9, 7, 4), (6, 5, 49), (33, 3, 100)])
def test_bf16_dx_pass_is_the_fp32_expression_rounded_once(shape, inplace, skip):
    """rk_bn_bwd_dx_pre_bf16 -- 16-byte cells (k_bn_bwd_dx_pre_flat16: a cell's two halves may sit in two planes, P = 196), the
    4-element sweep and the per-channel kernel (P % 4 != 0) -- against the same fp32 expression evaluated by PyTorch, element for
    element: dx = bf16(a (dz - k1 - xhat k2) [+ skip]), a = gamma invstd.  Bit-exact: one fp32 expression, one rounding."""
    from rubiksnet_amd import _native

    L = _native.lib()
    Fr, C, P = shape
    g = torch.Generator(device="cpu").manual_seed(Fr * 31 + C * 7 + P)
    dz = torch.randn(Fr, C, P, generator=g).cuda().bfloat16()
    x = (torch.randn(Fr, C, P, generator=g) * 1.3 + 0.2).cuda().bfloat16()
    sk = torch.randn(Fr, C, P, generator=g).cuda().bfloat16() if skip else None
    gamma = (torch.rand(C, generator=g) + 0.5).cuda()
    mean, invstd = (torch.randn(C, generator=g) * 0.2).cuda(), (torch.rand(C, generator=g) + 0.5).cuda()
    k12 = (torch.randn(2, C, generator=g) * 0.05).cuda()
    v = lambda t: t.view(1, C, 1)
    xh = (x.float() - v(mean)) * v(invstd)
    # the kernel's order of operations, fp32 without contraction (torch evaluates each op separately)
    ref = v(gamma * invstd) * (dz.float() - v(k12[0]) - xh * v(k12[1]))
    if skip:
        ref = ref + sk.float()
    ref = ref.bfloat16()
    out = dz.clone() if inplace else torch.empty_like(dz)
    src = out if inplace else dz
    _native.check(L.rk_bn_bwd_dx_pre_bf16(src.data_ptr(), x.data_ptr(), gamma.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                                          k12.data_ptr(), sk.data_ptr() if skip else None, out.data_ptr(), Fr, C, P,
                                          torch.cuda.current_stream().cuda_stream), "dx_pre")
    torch.cuda.synchronize()
    # (fma contraction inside the kernel may differ from torch's separate ops by an fp32 ulp before the bf16 rounding: allow one
    # bf16 ulp on the few elements that sit on a rounding boundary)
    a, b = out.float(), ref.float()
    ulp = torch.maximum(b.abs(), torch.tensor(1e-30, device=b.device)) * 2.0 ** -7
    assert bool(((a - b).abs() <= ulp).all())
    assert float((a != b).float().mean()) < 2e-3

    # the element-at-a-time per-channel kernel (what buffers off 8-byte alignment get) writes the same bits
    def off(t):
        buf = torch.empty(t.numel() + 1, dtype=t.dtype, device=t.device)
        buf[1:].copy_(t.reshape(-1))
        return buf[1:].view_as(t)
    dz1, x1 = off(dz), off(x)
    sk1 = off(sk) if skip else None
    out1 = off(torch.zeros_like(dz))
    _native.check(L.rk_bn_bwd_dx_pre_bf16(dz1.data_ptr(), x1.data_ptr(), gamma.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                                          k12.data_ptr(), sk1.data_ptr() if skip else None, out1.data_ptr(), Fr, C, P,
                                          torch.cuda.current_stream().cuda_stream), "dx_pre (unaligned)")
    torch.cuda.synchronize()
    assert torch.equal(out1, out)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("shape", [(16, 24, 28, 28, 8), (8, 72, 56, 56, 8), (16, 40, 14, 14, 8), (8, 12, 6, 10, 4), (16, 6, 112, 112, 8)])
def test_bn_relu_folded_into_the_temporal_filter_of_a_downsampling_block(dtype, shape):
    """bn_relu_tshift_fork (a downsampling -aq block: the activation feeds the AttentionShift AND the stride-2 projecting shortcut;
    rk_bn_relu_gather2_*, rk_tshift3_bn_backward_fork_*) against the unfused chain bn_relu -> (AttentionShift, [:, :, ::2, ::2]):
    both outputs, every gradient, running statistics."""
    from rubiksnet_amd.attention_shift import AttentionShift
    from rubiksnet_amd.fused_bn import bn_relu, bn_relu_tshift_fork

    NT, C, H, W, S = shape
    torch.manual_seed(sum(shape))
    x0 = (torch.randn(NT, C, H, W, device="cuda") * 1.5 + 0.3).to(dtype)
    gy = torch.randn(NT, C, H, W, device="cuda").to(dtype)
    gs = torch.randn(NT, C, H // 2, W // 2, device="cuda").to(dtype)
    res = []
    for fused in (True, False):
        torch.manual_seed(7)
        bn = nn.BatchNorm2d(C).cuda().train()
        shift = AttentionShift(S, C).cuda()
        with torch.no_grad():
            bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(0, 0.2)
        x = x0.clone().requires_grad_(True)
        if fused:
            r = bn_relu_tshift_fork(bn, shift, x)
            assert r is not None
            y, xs = r
        else:
            a = bn_relu(bn, x)
            y, xs = shift(a), a[:, :, ::2, ::2]
        torch.autograd.backward([y, xs], [gy, gs])
        res.append((y.detach().float(), xs.detach().float(), x.grad.float(), bn.weight.grad.clone(), bn.bias.grad.clone(),
                    shift.weight.grad.clone(), bn.running_mean.clone(), bn.running_var.clone(), int(bn.num_batches_tracked)))
    f, u = res
    assert f[8] == u[8] == 1
    tol = 2e-5 if dtype == torch.float32 else 2.0 ** -6        # bf16: the unfused chain rounds the activation and its gradient once more
    for name, a, b in zip(("y", "xs", "dx", "dgamma", "dbeta", "dtaps", "running_mean", "running_var"), f[:8], u[:8]):
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=0, atol=tol * max(1e-3, float(b.abs().max())), err_msg=name)
