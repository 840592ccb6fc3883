"""GPU parity of the fused TRAINING block (rubiksnet_amd/train_block.py) and of the kernels it is
built from, against torch in fp64 on the CPU with the shift evaluated by the CPU oracle -- i.e. against the reference's
own block arithmetic (rubiksnet/backbone.py:123-135: relu(bn1) -> conv2 -> relu(bn2) -> as3 -> conv3 + shortcut), not
against other HIP kernels."""
import copy

import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


class _OracleShift3D(torch.autograd.Function):
    """RubiksShift3D on CPU fp64 through the oracle (forward K1, backward K2-K5)."""

    @staticmethod
    def forward(ctx, x, shift, stride, normalize, t_factor, oracle):
        ctx.save_for_backward(x, shift)
        ctx.cfg = (stride, normalize, t_factor, oracle)
        y = oracle.rk3d_forward(x.detach().numpy(), shift.detach().numpy(), [1, stride, stride], [0, 0, 0])
        return torch.from_numpy(y)

    @staticmethod
    def backward(ctx, gy):
        x, shift = ctx.saved_tensors
        stride, normalize, t_factor, oracle = ctx.cfg
        gx, gs = oracle.rk3d_backward(np.ascontiguousarray(gy.numpy()), x.detach().numpy(), shift.detach().numpy(),
                                      [1, stride, stride], [0, 0, 0], normalize_grad=normalize,
                                      normalize_t_factor=t_factor)
        return torch.from_numpy(gx), torch.from_numpy(gs), None, None, None, None


def _reference_block(block, x, dout, T, oracle):
    """The block in fp64 on the CPU: stock nn modules + the oracle's shift.  Returns out, d(x), {param: grad}, module."""
    ref = copy.deepcopy(block).cpu().double().train()
    layer = ref.as3.rubiks3d
    stride = int(layer.stride[1])
    xr = x.detach().cpu().double().requires_grad_(True)
    pre1 = ref.bn1(xr)
    a1 = F.relu(pre1)
    short = xr if isinstance(ref.shortcut, nn.Identity) else ref.shortcut(a1)
    pre2 = ref.bn2(ref.conv2(a1))
    a2 = F.relu(pre2)
    kink = ((pre1.detach().abs() < 1e-4) | (pre2.detach().abs() < 1e-4).any(dim=1, keepdim=True)) \
        if pre1.shape[2:] == pre2.shape[2:] else None
    Fr, C, H, W = a2.shape
    t = layer.normalize_t_factor
    s = _OracleShift3D.apply(a2.view(Fr // T, T, C, H, W), layer.shift, stride, layer.normalize_grad,
                             T / H if t == "auto" else float(t), oracle)
    s = s.view(Fr, C, s.shape[3], s.shape[4])
    if ref.se is not None:                                          # the Small tier's gate (stock modules, fp64)
        s = s * ref.se.fc(s.mean(dim=(2, 3))).view(Fr, C, 1, 1)
    out = ref.conv3(s) + short
    out.backward(dout.detach().cpu().double())
    grads = {n: p.grad for n, p in ref.named_parameters()}
    # elements whose ReLU sits within 1e-4 of its kink in fp64: there an fp32 evaluation may pick the other side, and d(x)
    # at the pixel (bn2: every input channel through conv2's d(input); bn1: the element) differs by a whole gradient term
    ref._near_kink = kink
    return out.detach(), xr.grad, grads, ref


def _make_block(cin, cout, stride, T, seed, width=None, use_se=False):
    from rubiksnet_amd import RubiksNet
    from rubiksnet_amd.backbone import RubiksShiftBlock

    torch.manual_seed(seed)

    class Parent:
        expansion = 1
        normalize_grad = True
        quantize = False
        init_shift = "uniform"

    Parent.use_se = use_se
    block = RubiksShiftBlock(cin, cout, stride=stride, parent=Parent())
    from rubiksnet_amd.models import _Rubiks3DWrap
    block.as3 = _Rubiks3DWrap(block.as3, n_segment=T)
    with torch.no_grad():
        for m in block.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.weight.uniform_(0.5, 1.5)
                m.bias.normal_(0, 0.3)
                m.running_mean.normal_()
                m.running_var.uniform_(0.5, 1.5)
            elif isinstance(m, nn.Conv2d):
                m.weight.normal_(0, (2.0 / m.in_channels) ** 0.5)
    return block.to(DEV).train()


CASES = [
    # cin, cout, stride, (N, T, H, W)
    (16, 16, 1, (2, 4, 14, 14)),        # identity shortcut, tile kernels
    (12, 24, 1, (2, 4, 12, 16)),        # projecting shortcut, stride 1, LDS-DMA shift kernels
    (16, 32, 2, (2, 4, 28, 32)),        # downsampling block: strided shift + 1x1 / stride-2 shortcut
    (54, 54, 1, (1, 8, 28, 28)),        # Tiny's layer2 shape (one clip)
    (72, 144, 2, (1, 8, 56, 56)),       # Large's first layer2 block (one clip): > 128 output rows
    (24, 48, 2, (2, 4, 28, 28)),        # 28 -> 14: output rows of 14 pixels (4-pixel groups wrap rows in the shortcut)
    (288, 288, 1, (32, 8, 14, 14)),     # Large's layer3 block at the bench's per-GPU batch: 256 frames (the 12-wave GEMM of
                                        # rk_pw3.hip, 512 statistics tiles per channel, d(weight) over 50 176 pixels)
]
SE_CASES = [
    (24, 24, 1, (2, 4, 14, 14), 12),    # Small-tier block: SE gate after the shift (reduction 12), identity shortcut
    (36, 72, 2, (2, 4, 28, 28), 12),    # ... downsampling block with a projecting shortcut
]


@pytest.mark.parametrize("cin,cout,stride,dims,se", [c + (False,) for c in CASES] + SE_CASES)
def test_fused_train_block_matches_fp64_reference(oracle, cin, cout, stride, dims, se):
    from rubiksnet_amd import train_block

    N, T, H, W = dims
    block = _make_block(cin, cout, stride, T, seed=cin + cout + H, use_se=se)
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(N * T, cin, H, W, generator=g) * 1.3 + torch.randn(1, cin, 1, 1, generator=g)).to(DEV)
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    dout = torch.randn(N * T, cout, Ho, Wo, generator=g).to(DEV)
    out_ref, dx_ref, g_ref, ref = _reference_block(block, x, dout, T, oracle)

    xg = x.clone().requires_grad_(True)
    out = train_block.fused_train_block(block, xg)
    assert out is not None, "the block must qualify for the fused training path"
    assert getattr(out, "_rk_stats", None) is not None
    out.backward(dout)
    torch.cuda.synchronize()

    def close(a, b, tol, what):
        b = b.to(torch.float64)
        scale = max(1.0, float(b.abs().max()))
        err = float((a.detach().cpu().double() - b).abs().max())
        assert err <= tol * scale, "%s: max err %.3e (scale %.3g)" % (what, err, scale)

    K = max(cin, block.conv2.out_channels)
    close(out, out_ref, 4e-6 * K ** 0.5, "out")
    big = N * T * H * W > 40000                # 288 x 288 on 256 frames, 14.4 M activations: a handful sit on a ReLU kink to fp32 round-off
    dxg = xg.grad
    if big and ref._near_kink is not None:
        keep = ~ref._near_kink
        assert float((~keep).double().mean()) < 0.03
        dxg = xg.grad.detach().cpu().double() * keep
        dx_ref = dx_ref * keep
    close(dxg, dx_ref, 2e-5 * K ** 0.5, "d(x)")
    for name, p in block.named_parameters():
        assert p.grad is not None, name
        if name.endswith("shift"):
            close(p.grad, g_ref[name], 1e-4, name)                    # unit vectors after K5
        else:
            # (big: every parameter gradient sums over all elements, flipped kink elements included: 1e-3 class)
            close(p.grad, g_ref[name], (5e-3 if big else 3e-5 * K ** 0.5), name)
    # nn.BatchNorm2d's bookkeeping: running statistics (unbiased variance) and num_batches_tracked
    for bn, rbn in ((block.bn1, ref.bn1), (block.bn2, ref.bn2)):
        close(bn.running_mean, rbn.running_mean, 1e-5, "running_mean")
        close(bn.running_var, rbn.running_var, 1e-5, "running_var")
        assert int(bn.num_batches_tracked) == int(rbn.num_batches_tracked) == 1
    # the statistics handed to the next BatchNorm describe `out`
    from rubiksnet_amd import _native
    L = _native.lib()
    bn_next = nn.BatchNorm2d(cout).to(DEV).train()
    fin = train_block._finish(L, bn_next, out._rk_stats, N * T * Ho * Wo, out.device)
    close(fin[0], out_ref.mean(dim=(0, 2, 3)), 1e-5, "mean of out from the epilogue statistics")
    close(1.0 / fin[1] ** 2 - bn_next.eps, out_ref.var(dim=(0, 2, 3), unbiased=False), 2e-5, "variance of out")


def test_whole_network_train_step_equals_the_layer_by_layer_path(monkeypatch):
    """RubiksNet-Tiny, one optimizer step with RK_FUSED_TRAIN=1 vs =0 from the same initial state: same loss, same
    updated parameters and buffers to fp32 round-off (the unfused path is itself checked against torch fp64 in
    test_bn_gpu / test_pointwise_gpu and against the oracle in test_model_configs_gpu)."""
    from rubiksnet_amd import RubiksNet, config, dp

    torch.manual_seed(0)
    net0 = RubiksNet("tiny", 11, num_frames=8, verbose=False).to(DEV)
    clips = torch.randn(2, 8, 3, 224, 224, device=DEV)
    labels = torch.randint(0, 11, (2,), device=DEV)
    results = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("RK_FUSED_TRAIN", flag)
        config.reload()
        net = copy.deepcopy(net0).train()
        opt = torch.optim.SGD(net.parameters(), lr=0.05)
        loss = dp.train_step(net, opt, clips, labels)
        torch.cuda.synchronize()
        results[flag] = (float(loss), {k: v.detach().clone() for k, v in net.state_dict().items()})
    monkeypatch.delenv("RK_FUSED_TRAIN")
    config.reload()
    l1, s1 = results["1"]
    l0, s0 = results["0"]
    assert abs(l1 - l0) <= 2e-5 * max(1.0, abs(l0))
    for k in s0:
        a, b = s1[k].double(), s0[k].double()
        tol = 2e-3 if k.endswith("shift") else 2e-4          # shift gradients are unit vectors: a flipped tiny component
        assert float((a - b).abs().max()) <= tol * max(1.0, float(b.abs().max())), k


@pytest.mark.parametrize("shape,K,M,relu_in", [((8, 196), 16, 24, True), ((6, 3136), 54, 54, True), ((3, 784), 72, 144, False),
                                               ((5, 100), 10, 6, True)])
def test_gemm_statistics_epilogue(shape, K, M, relu_in):
    """rk_pw_gemm_stats_f32 + rk_bn_finish_tiles_f32: Y = W relu(a x + b) (+ R) as F.conv2d in fp64, and mean / invstd /
    running statistics of Y as F.batch_norm computes them -- with a large common offset on Y (|mean| >> std) to show
    that the pivoted tile sums do not cancel."""
    from rubiksnet_amd import _native

    L = _native.lib()
    Fr, P = shape
    g = torch.Generator().manual_seed(K + M)
    x = torch.randn(Fr, K, P, generator=g)
    w = torch.randn(M, K, generator=g) / K ** 0.5
    r = torch.randn(Fr, M, P, generator=g) * 0.5 + 300.0            # residual with a huge mean: |mean| = 300 std
    ka, kb = torch.rand(K, generator=g) + 0.5, torch.randn(K, generator=g) * 0.2
    xin = x.double() * ka.double().view(1, K, 1) + kb.double().view(1, K, 1)
    if relu_in:
        xin = xin.clamp_min(0)
    y_ref = torch.einsum("mk,fkp->fmp", w.double(), xin) + r.double()
    xd, wd, rd, kad, kbd = (t.to(DEV).contiguous() for t in (x, w, r, ka, kb))
    y = torch.empty(Fr, M, P, device=DEV)
    J = int(L.rk_pw_gemm_tiles(wd.data_ptr(), Fr, K, M, P, 1))     # 128- or 64-column tiles: the kernel generation's choice
    stats = torch.zeros(M, J, 4, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    _native.check(L.rk_pw_gemm_stats_f32(wd.data_ptr(), xd.data_ptr(), rd.data_ptr(), y.data_ptr(), Fr, K, M, P, 1,
                                         kad.data_ptr(), kbd.data_ptr(), int(relu_in), stats.data_ptr(), J, st), "gemm_stats")
    assert float((y.cpu().double() - y_ref).abs().max()) <= 4e-6 * K ** 0.5 * float(y_ref.abs().max())
    gamma, beta = torch.rand(M, device=DEV) + 0.5, torch.randn(M, device=DEV)
    rm, rv = torch.zeros(M, device=DEV), torch.ones(M, device=DEV)
    nbt = torch.zeros((), dtype=torch.int64, device=DEV)
    out = torch.empty(8, M, device=DEV)
    _native.check(L.rk_bn_finish_tiles_f32(stats.data_ptr(), J, Fr * P, gamma.data_ptr(), beta.data_ptr(), rm.data_ptr(),
                                           rv.data_ptr(), out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(),
                                           out[3].data_ptr(), out[4].data_ptr(), M, 1e-5, 0.1, nbt.data_ptr(), st), "finish")
    packed = out[4:].reshape(M, 4)
    assert torch.equal(packed, torch.stack([out[2], out[3], out[0], out[1]], dim=1))
    yg = y.cpu().double()                                           # statistics of what the kernel actually stored
    mean, var = yg.mean(dim=(0, 2)), yg.var(dim=(0, 2), unbiased=False)
    assert float((out[0].cpu().double() - mean).abs().max()) <= 1e-6 * 300
    np.testing.assert_allclose(out[1].cpu().double().numpy(), (1.0 / torch.sqrt(var + 1e-5)).numpy(), rtol=2e-5)
    n = Fr * P
    np.testing.assert_allclose(rv.cpu().double().numpy(), (0.9 + 0.1 * var * n / (n - 1)).numpy(), rtol=2e-5)
    np.testing.assert_allclose(rm.cpu().double().numpy(), (0.1 * mean).numpy(), rtol=1e-6)
    assert int(nbt) == 1
    # the stand-alone tile-statistics kernel produces the same partial format
    J2 = int(L.rk_pw_tiles(Fr, P))                                  # (its tiles are 128 columns wide whatever the GEMM's are)
    stats2 = torch.zeros(M, J2, 4, device=DEV)
    _native.check(L.rk_bn_tile_stats_f32(y.data_ptr(), stats2.data_ptr(), Fr, M, P, st), "tile_stats")
    out2 = torch.empty(4, M, device=DEV)
    _native.check(L.rk_bn_finish_tiles_f32(stats2.data_ptr(), J2, Fr * P, gamma.data_ptr(), beta.data_ptr(), None, None,
                                           out2[0].data_ptr(), out2[1].data_ptr(), out2[2].data_ptr(), out2[3].data_ptr(),
                                           None, M, 1e-5, 0.1, None, st), "finish")
    np.testing.assert_allclose(out2[:2].cpu().numpy(), out[:2].cpu().numpy(), rtol=3e-6)


@pytest.mark.parametrize("shape,K,M", [((8, 196), 24, 16), ((4, 3136), 54, 54), ((3, 784), 144, 72)])
def test_dgrad_with_bn_backward_epilogue_and_prologue_wgrad(shape, K, M):
    """rk_pw_gemm_bnbwd_f32 / rk_bn_bwd_finish_tiles_f32 / rk_bn_bwd_dx_pre_f32 / rk_pw_wgrad_pro_f32 = autograd of
    conv2(relu(bn1(x))) in fp64: d(x), d(gamma), d(beta), d(W)."""
    from rubiksnet_amd import _native

    L = _native.lib()
    Fr, P = shape
    C = M                                   # channels of x (the GEMM's M); K = channels of dz
    g = torch.Generator().manual_seed(K * M)
    x = torch.randn(Fr, C, P, generator=g) * 1.5 + torch.randn(1, C, 1, generator=g)
    w = torch.randn(K, C, generator=g) / C ** 0.5          # conv2 weight [Cout=K][Cin=C]
    dz2 = torch.randn(Fr, K, P, generator=g)
    skip = torch.randn(Fr, C, P, generator=g)
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3
    # fp64 reference through autograd
    xr = x.double().requires_grad_(True)
    gr, br, wr = gamma.double().requires_grad_(True), beta.double().requires_grad_(True), w.double().requires_grad_(True)
    a1 = F.relu(F.batch_norm(xr.unsqueeze(-1), None, None, gr, br, True, 0.0, 1e-5)).squeeze(-1)
    z = torch.einsum("kc,fcp->fkp", wr, a1)
    (z * dz2.double()).sum().backward()
    dx_ref = xr.grad + skip.double()
    # the fused kernels
    mean = x.double().mean(dim=(0, 2))
    var = x.double().var(dim=(0, 2), unbiased=False)
    inv = 1.0 / torch.sqrt(var + 1e-5)
    a = (gamma.double() * inv).float()
    b = torch.addcmul(beta, -mean.float(), a)
    d = lambda t: t.float().to(DEV).contiguous()                                         # noqa: E731
    xd, wd, dzd, sd, ad, bd, md, ivd, gd = d(x), d(w), d(dz2), d(skip), d(a), d(b), d(mean), d(inv), d(gamma)
    st = torch.cuda.current_stream().cuda_stream
    J = int(L.rk_pw_gemm_tiles(wd.data_ptr(), Fr, K, C, P, 0))
    bred = torch.zeros(C, J, 2, device=DEV)
    dzm = torch.empty(Fr, C, P, device=DEV)
    pack = torch.stack([ad, bd, md, ivd], dim=1).contiguous()
    _native.check(L.rk_pw_gemm_bnbwd_f32(wd.data_ptr(), dzd.data_ptr(), None, dzm.data_ptr(), Fr, K, C, P, 0, xd.data_ptr(),
                                         pack.data_ptr(), bred.data_ptr(), J, st), "bnbwd")
    k12, dg, db = torch.empty(2, C, device=DEV), torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    _native.check(L.rk_bn_bwd_finish_tiles_f32(bred.data_ptr(), J, Fr * P, k12.data_ptr(), dg.data_ptr(), db.data_ptr(), C, st),
                  "bwd_finish")
    dx = torch.empty_like(xd)
    _native.check(L.rk_bn_bwd_dx_pre_f32(dzm.data_ptr(), xd.data_ptr(), gd.data_ptr(), md.data_ptr(), ivd.data_ptr(),
                                         k12.data_ptr(), sd.data_ptr(), dx.data_ptr(), Fr, C, P, st), "dx_pre")
    dw = torch.empty(K, C, device=DEV)
    nb = int(L.rk_pw_wgrad_workspace_bytes(Fr, C, K, P))
    ws = torch.empty(nb, dtype=torch.uint8, device=DEV)
    _native.check(L.rk_pw_wgrad_pro_f32(dzd.data_ptr(), xd.data_ptr(), dw.data_ptr(), Fr, C, K, P, ad.data_ptr(), bd.data_ptr(),
                                        1, ws.data_ptr(), nb, st), "wgrad_pro")
    torch.cuda.synchronize()

    def close(got, ref, tol, what):
        scale = max(1.0, float(ref.abs().max()))
        err = float((got.cpu().double() - ref).abs().max())
        assert err <= tol * scale, "%s: %.3e (scale %.3g)" % (what, err, scale)

    close(dx, dx_ref, 1e-5 * K ** 0.5, "d(x)")
    close(dg, gr.grad, 2e-5 * K ** 0.5, "d(gamma)")
    close(db, br.grad, 2e-5 * K ** 0.5, "d(beta)")
    close(dw, wr.grad, 2e-5 * K ** 0.5, "d(W)")
