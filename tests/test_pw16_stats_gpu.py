"""The bf16 1x1 GEMM with the statistics epilogue (rk_pw_gemm_packed_stats_bf16, round 5): Y must be bit-identical to the plain
GEMM's, and the tile records -- (pivot, sum(y - pivot), sum((y - pivot)^2), columns) per 64 columns of Y as stored --
must add up to the per-channel mean / biased variance of the stored bf16 tensor (what nn.BatchNorm2d would compute from
it), including planes whose last 16-byte unit repeats four pixels (H * W % 8 == 4: the 14 x 14 layers)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


@pytest.mark.parametrize("Fr,K,M,H", [(16, 72, 72, 56), (24, 144, 144, 28), (40, 288, 288, 14), (8, 54, 108, 28),
                                     (3, 288, 576, 14), (5, 32, 16, 6)])
@pytest.mark.parametrize("res", [False, True])
def test_stats_epilogue(Fr, K, M, H, res):
    from rubiksnet_amd import _native

    L = _native.lib()
    P = H * H
    g = torch.Generator(device=DEV).manual_seed(Fr * 131 + K + M)
    x = torch.randn(Fr, K, P, device=DEV, generator=g).bfloat16()
    r = (torch.randn(Fr, M, P, device=DEV, generator=g) * 3).bfloat16() if res else None
    w = torch.randn(M, K, device=DEV, generator=g) / K ** 0.5 + 0.05
    pf = torch.empty(int(L.rk_pw_packed_bytes(M, K)), dtype=torch.uint8, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    _native.check(L.rk_pw_pack_bf16(w.data_ptr(), M, K, pf.data_ptr(), None, st), "pack")
    y0 = torch.empty(Fr, M, P, device=DEV, dtype=torch.bfloat16)
    y1 = torch.full((Fr, M, P), float("nan"), device=DEV, dtype=torch.bfloat16)
    _native.check(L.rk_pw_gemm_packed_bf16(pf.data_ptr(), x.data_ptr(), r.data_ptr() if res else None, y0.data_ptr(), Fr, K, M, P,
                                           st), "gemm")
    J = int(L.rk_pw16_stat_tiles(Fr, P))
    stats = torch.full((M, J, 4), float("nan"), device=DEV)
    _native.check(L.rk_pw_gemm_packed_stats_bf16(pf.data_ptr(), x.data_ptr(), r.data_ptr() if res else None, y1.data_ptr(), Fr, K,
                                                 M, P, stats.data_ptr(), J, st), "gemm_stats")
    torch.cuda.synchronize()
    assert torch.equal(y0, y1)
    s = stats.double()
    assert bool(torch.isfinite(s).all())
    n = s[:, :, 3]
    assert float(n.sum(1).min()) == Fr * P == float(n.sum(1).max())          # every column counted exactly once
    yd = y1.double()
    tot = (s[:, :, 0] * n + s[:, :, 1]).sum(1)
    ref1 = yd.sum(dim=(0, 2))
    assert float((tot - ref1).abs().max()) <= 1e-5 * float(yd.abs().sum(dim=(0, 2)).max())
    sq = (s[:, :, 2] + 2 * s[:, :, 0] * s[:, :, 1] + n * s[:, :, 0] ** 2).sum(1)      # sum of y^2 per row
    ref2 = (yd * yd).sum(dim=(0, 2))
    assert float((sq - ref2).abs().max()) <= 2e-5 * float(ref2.max())
    # through the finisher: mean / invstd as nn.BatchNorm2d would compute them from the stored tensor
    gamma, beta = torch.ones(M, device=DEV), torch.zeros(M, device=DEV)
    out = torch.empty(8, M, device=DEV)
    _native.check(L.rk_bn_finish_tiles_f32(stats.data_ptr(), J, Fr * P, gamma.data_ptr(), beta.data_ptr(), None, None,
                                           out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), out[3].data_ptr(),
                                           out[4].data_ptr(), M, 1e-5, 0.1, None, st), "finish")
    mean = yd.mean(dim=(0, 2))
    var = yd.var(dim=(0, 2), unbiased=False)
    assert float((out[0].double() - mean).abs().max()) <= 1e-5 * max(1.0, float(mean.abs().max()))
    assert float((out[1].double() - 1 / torch.sqrt(var + 1e-5)).abs().max()) <= 1e-4 * float((1 / torch.sqrt(var + 1e-5)).max())


def test_aq_block_uses_the_epilogue_statistics(monkeypatch):
    """RK_PW16_STATS=1: every bn2 of an -aq network under bf16 autocast finds the tile statistics conv2's GEMM left for it,
    and the step's loss equals the default path's (statistics pass) to bf16 round-off."""
    from rubiksnet_amd import RubiksNet, config, fused_bn

    torch.manual_seed(0)
    net0 = RubiksNet("tiny", 7, num_frames=8, variant="rubiks3d-aq", verbose=False).to(DEV).train()
    clips0 = torch.randn(2, 8, 3, 224, 224, device=DEV)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        ref = net0(clips0).float()
    monkeypatch.setenv("RK_PW16_STATS", "1")
    config.reload()
    torch.manual_seed(0)
    net = RubiksNet("tiny", 7, num_frames=8, variant="rubiks3d-aq", verbose=False).to(DEV).train()
    clips = torch.randn(2, 8, 3, 224, 224, device=DEV)
    seen = []
    orig = fused_bn.take_stats

    def spy(x):
        st = orig(x)
        seen.append(st is not None)
        return st
    fused_bn.take_stats = spy
    try:
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = net(clips)
        out.float().sum().backward()
        torch.cuda.synchronize()
    finally:
        fused_bn.take_stats = orig
        monkeypatch.delenv("RK_PW16_STATS")
        config.reload()
    assert sum(seen) >= 15, "every bn2 fed by a bf16 1x1 GEMM must find its tile statistics (%d of %d did)" % (sum(seen), len(seen))
    assert float((out.float() - ref).abs().max()) <= 0.05 * max(1.0, float(ref.abs().max()))
    assert all(p.grad is None or bool(torch.isfinite(p.grad).all()) for p in net.parameters())
