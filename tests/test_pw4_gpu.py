"""Streaming GEMM of the shallow 1x1 convolutions (rk_pw4.hip: operand in registers, everything else through a per-wave
LDS-DMA record ring) through the C ABI hook rk_pw4_gemm_f32, against fp64 PyTorch: every epilogue (plain, + R, prologue,
statistics tiles, BatchNorm-backward mask + sums), both operand layouts, both instances (54- and 72-channel), ragged channel
counts inside an instance, tiles that straddle frames, a last tile that is partly outside the tensor, fewer tiles than waves
and many tiles per wave.  Reference semantics: rubiksnet/backbone.py:44-45 (Conv1x1), :123-135 (the block around it)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")

# (F, P, K, M): P % 4 == 0
SHAPES = [
    (3, 196, 54, 54),        # 14 x 14 planes: tiles straddle frames, last tile ragged (588 columns = 9.19 tiles)
    (2, 3136, 54, 54),       # 56 x 56: 98 tiles
    (2, 784, 72, 72),        # 28 x 28
    (5, 100, 50, 60),        # ragged channel counts inside the (4, 14) instance; 500 columns
    (3, 400, 66, 70),        # ragged inside the (5, 18) instance
    (40, 3136, 54, 54),      # 1960 tiles: every wave slot busy, ring crosses tile boundaries
    (24, 3136, 72, 72),      # 1176 tiles
    (1, 4, 72, 72),          # one column group
]


def _lib():
    from rubiksnet_amd import _native
    return _native, _native.lib()


def _mk(shape, seed):
    Fr, P, K, M = shape
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(Fr, K, P, generator=g)
    w = torch.randn(M, K, generator=g) / K ** 0.5
    r = torch.randn(Fr, M, P, generator=g)
    return x, w, r, g


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("a_is_mk", [1, 0])
@pytest.mark.parametrize("pro,res", [(0, 0), (1, 0), (0, 1), (1, 1)])
def test_plain_prologue_residual(shape, a_is_mk, pro, res):
    native, L = _lib()
    Fr, P, K, M = shape
    x, w, r, g = _mk(shape, 7 * K + M + pro + 2 * res)
    ka, kb = torch.rand(K, generator=g) + 0.5, torch.randn(K, generator=g) * 0.3
    xin = x.double()
    if pro:
        xin = (xin * ka.double().view(1, K, 1) + kb.double().view(1, K, 1)).clamp_min(0)
    ref = torch.einsum("mk,fkp->fmp", w.double(), xin) + (r.double() if res else 0.0)
    A = (w if a_is_mk else w.t()).contiguous().to(DEV)
    xd, rd, kad, kbd = (t.to(DEV).contiguous() for t in (x, r, ka, kb))
    y = torch.full((Fr, M, P), float("nan"), device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    J = (Fr * P + 63) // 64
    native.check(L.rk_pw4_gemm_f32(A.data_ptr(), xd.data_ptr(), rd.data_ptr() if res else None, y.data_ptr(), Fr, K, M, P, a_is_mk,
                                   kad.data_ptr() if pro else None, kbd.data_ptr() if pro else None, 1, 0, None, None, None, None, J,
                                   st), "rk_pw4_gemm_f32")
    err = float((y.cpu().double() - ref).abs().max())
    assert err <= 4e-6 * K ** 0.5 * float(ref.abs().max()), err


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("pro,res", [(1, 0), (0, 1), (0, 0)])
def test_statistics_tiles(shape, pro, res):
    """EPI 1: Y as above, and the per-(row, 64-column tile) records (pivot, sum(y - pivot), sum((y - pivot)^2), n) finished by
    rk_bn_finish_tiles_f32 = mean / invstd of what the kernel stored; a residual with |mean| = 300 sigma shows the pivot."""
    native, L = _lib()
    Fr, P, K, M = shape
    x, w, r, g = _mk(shape, 11 * K + M + pro)
    r = r * 0.5 + 300.0
    ka, kb = torch.rand(K, generator=g) + 0.5, torch.randn(K, generator=g) * 0.3
    A = w.contiguous().to(DEV)
    xd, rd, kad, kbd = (t.to(DEV).contiguous() for t in (x, r, ka, kb))
    y = torch.empty(Fr, M, P, device=DEV)
    J = (Fr * P + 63) // 64
    stats = torch.full((M, J, 4), float("nan"), device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    native.check(L.rk_pw4_gemm_f32(A.data_ptr(), xd.data_ptr(), rd.data_ptr() if res else None, y.data_ptr(), Fr, K, M, P, 1,
                                   kad.data_ptr() if pro else None, kbd.data_ptr() if pro else None, 1, 1, stats.data_ptr(), None,
                                   None, None, J, st), "rk_pw4_gemm_f32")
    xin = x.double()
    if pro:
        xin = (xin * ka.double().view(1, K, 1) + kb.double().view(1, K, 1)).clamp_min(0)
    ref = torch.einsum("mk,fkp->fmp", w.double(), xin) + (r.double() if res else 0.0)
    assert float((y.cpu().double() - ref).abs().max()) <= 4e-6 * K ** 0.5 * float(ref.abs().max())
    assert float(stats[:, :, 3].sum(1).min()) == Fr * P == float(stats[:, :, 3].sum(1).max())     # every column counted once
    out = torch.empty(4, M, device=DEV)
    gamma, beta = torch.ones(M, device=DEV), torch.zeros(M, device=DEV)
    native.check(L.rk_bn_finish_tiles_f32(stats.data_ptr(), J, Fr * P, gamma.data_ptr(), beta.data_ptr(), None, None, out[0].data_ptr(),
                                          out[1].data_ptr(), out[2].data_ptr(), out[3].data_ptr(), None, M, 1e-5, 0.1, None, st),
                 "finish")
    yg = y.cpu().double()
    mean, var = yg.mean(dim=(0, 2)), yg.var(dim=(0, 2), unbiased=False)
    assert float((out[0].cpu().double() - mean).abs().max()) <= 1e-6 * max(1.0, float(mean.abs().max()))
    np.testing.assert_allclose(out[1].cpu().double().numpy(), (1.0 / torch.sqrt(var + 1e-5)).numpy(), rtol=3e-5)


@pytest.mark.parametrize("shape", SHAPES)
def test_batchnorm_backward_epilogue(shape):
    """EPI 2: dz = (A dY) [a x + b > 0] and the tile sums (sum dz, sum dz xhat), as rk_pw_gemm_bnbwd_f32 defines them."""
    native, L = _lib()
    Fr, P, K, M = shape
    x, w, bx, g = _mk(shape, 13 * K + M)
    pack = torch.stack([torch.rand(M, generator=g) + 0.5, torch.randn(M, generator=g) * 0.3, torch.randn(M, generator=g) * 0.1,
                        torch.rand(M, generator=g) + 0.5], dim=1).contiguous()
    A = w.t().contiguous().to(DEV)                                    # [K][M]: the d(input) layout
    xd, bxd, pk = x.to(DEV), bx.to(DEV), pack.to(DEV)
    dz = torch.empty(Fr, M, P, device=DEV)
    J = (Fr * P + 63) // 64
    bred = torch.full((M, J, 2), float("nan"), device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    native.check(L.rk_pw4_gemm_f32(A.data_ptr(), xd.data_ptr(), None, dz.data_ptr(), Fr, K, M, P, 0, None, None, 0, 2, None,
                                   bxd.data_ptr(), pk.data_ptr(), bred.data_ptr(), J, st), "rk_pw4_gemm_f32")
    da = torch.einsum("mk,fkp->fmp", w.double(), x.double())
    pa, pb, mu, iv = (pack[:, i].view(1, M, 1) for i in range(4))
    pre = pa.double() * bx.double() + pb.double()     # the kernel tests fmaf(a, x, b) > 0: the sign of the exact value
    near = pre.abs() < 1e-12
    ref = da * (pre > 0).double()
    got = dz.cpu().double()
    err = ((got - ref).abs() * (~near)).max()
    assert float(err) <= 4e-6 * K ** 0.5 * float(da.abs().max())
    xhat = (bx.double() - mu.double()) * iv.double()
    s1 = got.sum(dim=(0, 2)); s2 = (got * xhat).sum(dim=(0, 2))       # sums of what the kernel stored
    b = bred.cpu().double().sum(1)
    scale = float(got.abs().sum(dim=(0, 2)).max())
    assert float((b[:, 0] - s1).abs().max()) <= 2e-6 * scale
    assert float((b[:, 1] - s2).abs().max()) <= 2e-5 * scale


def test_dispatch_takes_the_shallow_layers_and_matches():
    """rk_pw_gemm_f32 at a size above the dispatch threshold gives what the hook gives bit for bit (same kernel), and
    rk_pw_gemm_tiles promises 64-column records for the 54-channel [M][K] layer (first generation: 128)."""
    native, L = _lib()
    Fr, P, K, M = 96, 3136, 54, 54                                  # 4704 tiles
    x, w, r, g = _mk((Fr, P, K, M), 5)
    A, xd = w.contiguous().to(DEV), x.to(DEV)
    y1, y2 = torch.empty(Fr, M, P, device=DEV), torch.empty(Fr, M, P, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    J = (Fr * P + 63) // 64
    assert int(L.rk_pw_gemm_tiles(A.data_ptr(), Fr, K, M, P, 1)) == J
    native.check(L.rk_pw_gemm_f32(A.data_ptr(), xd.data_ptr(), None, y1.data_ptr(), Fr, K, M, P, 1, st), "rk_pw_gemm_f32")
    native.check(L.rk_pw4_gemm_f32(A.data_ptr(), xd.data_ptr(), None, y2.data_ptr(), Fr, K, M, P, 1, None, None, 0, 0, None, None, None,
                                   None, J, st), "rk_pw4_gemm_f32")
    assert torch.equal(y1, y2)
    ref = torch.einsum("mk,fkp->fmp", w.double(), x.double())
    assert float((y1.cpu().double() - ref).abs().max()) <= 4e-6 * K ** 0.5 * float(ref.abs().max())


def test_no_instance_is_reported():
    native, L = _lib()
    x = torch.zeros(2, 144, 196, device=DEV); w = torch.zeros(144, 144, device=DEV); y = torch.empty(2, 144, 196, device=DEV)
    rc = L.rk_pw4_gemm_f32(w.data_ptr(), x.data_ptr(), None, y.data_ptr(), 2, 144, 144, 196, 1, None, None, 0, 0, None, None, None, None,
                           7, torch.cuda.current_stream().cuda_stream)
    assert rc != 0


@pytest.mark.parametrize("K,M", [(54, 54), (72, 72)])
@pytest.mark.parametrize("pro,res,relu_out", [(1, 1, 1), (0, 0, 0), (1, 0, 1)])
def test_inference_epilogue_through_the_dispatch(K, M, pro, res, relu_out):
    """rk_pw_gemm_fused_f32 (the folded-BatchNorm inference form: Y = relu?(ma (A relu?(ka x + kb)) + mb) + R) above the
    dispatch threshold, i.e. on the streaming kernel, against fp64."""
    native, L = _lib()
    Fr, P = 96, 3136
    x, w, r, g = _mk((Fr, P, K, M), 3 * K + pro + res)
    ka, kb = torch.rand(K, generator=g) + 0.5, torch.randn(K, generator=g) * 0.3
    ma, mb = torch.rand(M, generator=g) + 0.5, torch.randn(M, generator=g) * 0.3
    xin = x.double()
    if pro:
        xin = (xin * ka.double().view(1, K, 1) + kb.double().view(1, K, 1)).clamp_min(0)
    ref = torch.einsum("mk,fkp->fmp", w.double(), xin) * ma.double().view(1, M, 1) + mb.double().view(1, M, 1)
    if relu_out:
        ref = ref.clamp_min(0)
    if res:
        ref = ref + r.double()
    A, xd, rd, kad, kbd, mad, mbd = (t.to(DEV).contiguous() for t in (w, x, r, ka, kb, ma, mb))
    y = torch.full((Fr, M, P), float("nan"), device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    native.check(L.rk_pw_gemm_fused_f32(A.data_ptr(), xd.data_ptr(), rd.data_ptr() if res else None, y.data_ptr(), Fr, K, M, P, 1,
                                        kad.data_ptr() if pro else None, kbd.data_ptr() if pro else None, 1, mad.data_ptr(),
                                        mbd.data_ptr(), relu_out, st), "rk_pw_gemm_fused_f32")
    err = float((y.cpu().double() - ref).abs().max())
    assert err <= 6e-6 * K ** 0.5 * float(ref.abs().max()), err
