"""The fp32 1x1 convolution entry points are served by four kernel generations (rk_pw.hip, rk_pw2.hip, rk_pw3.hip, rk_pw4.hip)
chosen per call from (F, K, M, P, layout, epilogue, residual).  A caller of the training epilogues allocates its tile records
from rk_pw_gemm_tiles() BEFORE the call, so the promise and the dispatch must agree for every shape and batch size a network
can present -- a disagreement is RK_ERR_BAD_DIMS in the middle of a train step.  This sweeps the layer shapes of the four
tiers (rubiksnet/backbone.py:139-171: widths 54 / 72 x (1, 2, 4, 8) on 112 .. 7 pixel planes) over batch sizes from one clip
to 64, through every epilogue the fused training block uses (rubiksnet_amd/train_block.py), and checks: the call succeeds with
the promised tile count, every column is counted exactly once in the statistics records, the BatchNorm-backward sums equal
the sums of what was stored, and Y equals a fp64 reference on a slice."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")

LAYERS = [  # (K, M, H): conv2 / conv3 / projecting conv2 of the stages that run on planes with H * W % 4 == 0
    (54, 54, 112), (54, 54, 56), (54, 108, 56), (108, 108, 28), (108, 216, 28), (216, 216, 14),
    (72, 72, 112), (72, 72, 56), (72, 144, 56), (144, 144, 28), (144, 288, 28), (288, 288, 14),
]
FRAMES = [8, 32, 128, 256, 512]             # 1, 4, 16, 32, 64 clips of 8 frames


def _ref_slice(w, x, f, extra=None):
    y = torch.einsum("mk,kp->mp", w.double().cpu(), x[f].double().cpu())
    return y if extra is None else y + extra[f].double().cpu()


@pytest.mark.parametrize("K,M,H", LAYERS)
@pytest.mark.parametrize("Fr", FRAMES)
def test_tile_promise_and_epilogues(K, M, H, Fr):
    from rubiksnet_amd import _native

    L = _native.lib()
    P = H * H
    if Fr * max(K, M) * P * 4 > (3 << 30):
        pytest.skip("larger than any per-GPU batch of the bench")
    g = torch.Generator(device=DEV).manual_seed(K * 7 + M + Fr)
    x = torch.randn(Fr, K, P, device=DEV, generator=g)
    r = torch.randn(Fr, M, P, device=DEV, generator=g)
    w = torch.randn(M, K, device=DEV, generator=g) / K ** 0.5
    ka, kb = torch.rand(K, device=DEV, generator=g) + 0.5, torch.randn(K, device=DEV, generator=g) * 0.3
    y = torch.empty(Fr, M, P, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    n = Fr * P

    # forward, [M][K]: prologue + statistics (conv2), residual + statistics (conv3)
    J = int(L.rk_pw_gemm_tiles(w.data_ptr(), Fr, K, M, P, 1))
    assert J > 0
    for pro, res in ((1, 0), (0, 1), (0, 0)):
        stats = torch.full((M, J, 4), float("nan"), device=DEV)
        _native.check(L.rk_pw_gemm_stats_f32(w.data_ptr(), x.data_ptr(), r.data_ptr() if res else None, y.data_ptr(), Fr, K, M, P, 1,
                                             ka.data_ptr() if pro else None, kb.data_ptr() if pro else None, 1, stats.data_ptr(), J,
                                             st), "rk_pw_gemm_stats_f32")
        cnt = stats[:, :, 3].sum(1)
        assert float(cnt.min()) == n == float(cnt.max()), (pro, res)
        tot = (stats[:, :, 0].double() * stats[:, :, 3].double() + stats[:, :, 1].double()).sum(1)      # sum of y per row
        ysum = y.double().sum(dim=(0, 2))
        assert float((tot - ysum).abs().max()) <= 1e-5 * float(y.double().abs().sum(dim=(0, 2)).max())
        f = Fr - 1
        xin = x if not pro else torch.relu(x * ka.view(1, K, 1) + kb.view(1, K, 1))
        ref = _ref_slice(w, xin, f, r if res else None)
        assert float((y[f].double().cpu() - ref).abs().max()) <= 4e-6 * K ** 0.5 * float(ref.abs().max())

    # d(input) of conv2, [K][M] operand (A = W^T of a layer with Cin = M, Cmid = K): BatchNorm-backward epilogue, +- residual
    wt = w.t().contiguous()                                           # [K][M]
    Jb = int(L.rk_pw_gemm_tiles(wt.data_ptr(), Fr, K, M, P, 0))
    bx = torch.randn(Fr, M, P, device=DEV, generator=g)
    pack = torch.stack([torch.rand(M, device=DEV, generator=g) + 0.5, torch.randn(M, device=DEV, generator=g) * 0.3,
                        torch.randn(M, device=DEV, generator=g) * 0.1, torch.rand(M, device=DEV, generator=g) + 0.5], dim=1).contiguous()
    for res in (0, 1):
        bred = torch.full((M, Jb, 2), float("nan"), device=DEV)
        _native.check(L.rk_pw_gemm_bnbwd_f32(wt.data_ptr(), x.data_ptr(), r.data_ptr() if res else None, y.data_ptr(), Fr, K, M, P, 0,
                                             bx.data_ptr(), pack.data_ptr(), bred.data_ptr(), Jb, st), "rk_pw_gemm_bnbwd_f32")
        b = bred.double().sum(1)
        assert bool(torch.isfinite(b).all()), res
        s1 = y.double().sum(dim=(0, 2))
        assert float((b[:, 0] - s1).abs().max()) <= 2e-6 * float(y.double().abs().sum(dim=(0, 2)).max())
        xhat = (bx.double() - pack[:, 2].double().view(1, M, 1)) * pack[:, 3].double().view(1, M, 1)
        s2 = (y.double() * xhat).sum(dim=(0, 2))
        assert float((b[:, 1] - s2).abs().max()) <= 2e-5 * float((y.double() * xhat).abs().sum(dim=(0, 2)).max())
    # plain d(input) of conv3
    _native.check(L.rk_pw_gemm_f32(wt.data_ptr(), x.data_ptr(), None, y.data_ptr(), Fr, K, M, P, 0, st), "rk_pw_gemm_f32")
    f = 0
    ref = _ref_slice(w, x, f)
    assert float((y[f].double().cpu() - ref).abs().max()) <= 4e-6 * K ** 0.5 * float(ref.abs().max())


@pytest.mark.parametrize("env", [{"RK_PW2": "0"}, {"RK_PW4": "0"}, {"RK_PW4": "2"}, {"RK_PW2": "0", "RK_PW4": "2"}],
                         ids=["pw2-off", "pw4-off", "pw4-everywhere", "pw2-off-pw4-everywhere"])
def test_tile_promise_under_the_generation_switches(env):
    """The switches are read once per process, so the sweep above runs again in a subprocess on each non-default setting
    (round-4 advisor finding: with RK_PW2=0 the promise of rk_pw_gemm_tiles() and the dispatch disagreed for the 72 -> 72
    conv3 with a residual: RK_ERR_BAD_DIMS in the middle of a train step)."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-k", "test_tile_promise_and_epilogues and not 512",
           os.path.join(root, "tests", "test_pw_dispatch_gpu.py")]
    r = subprocess.run(cmd, cwd=root, env=dict(os.environ, **env), capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout
