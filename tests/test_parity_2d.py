"""GPU parity: RubiksShift2D through the C ABI versus the CPU oracle."""
import numpy as np
import pytest
import torch

from _util import rand, seed_of, special_shifts, to_dev, to_np

pytestmark = pytest.mark.gpu

SHAPES = [
    (2, 16, 14, 14, 1, 0),
    (3, 5, 28, 28, 2, 0),
    (2, 7, 9, 11, 2, 1),
    (1, 4, 10, 7, (1, 3), (2, 0)),
    (16, 9, 7, 7, 1, 0),
    (2, 3, 56, 56, 1, 0),
    # streaming kernels (stride 1, pad 0, W % 4 == 0): frame groups with a ragged last group, row bands
    (20, 6, 28, 28, 1, 0),
    (9, 4, 12, 16, 1, 0),
    (5, 3, 112, 112, 1, 0),
    (70, 512, 4, 8, 1, 0),
    (37, 6, 14, 14, 1, 0),        # tile kernels: ragged last frame group, 3 channel groups
]


def _fwd(x, shift, s, p, q, output=None):
    from rubiksnet_amd.shiftlib.rubiks2d.primitive import rubiks2d_forward
    return to_np(rubiks2d_forward(to_dev(x), to_dev(shift), s, p, quantize=q, output=output))


def _bwd(gy, x, shift, s, p, q, normalize=True, enable=True):
    from rubiksnet_amd.shiftlib.rubiks2d.primitive import rubiks2d_backward
    gx, gs = rubiks2d_backward(to_dev(gy), to_dev(x), to_dev(shift), s, p, normalize, enable, q)
    return to_np(gx), to_np(gs)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("quantize", [False, True])
@pytest.mark.parametrize("kind", ["generic", "wide", "integer", "half", "oob", "tiny"])
@pytest.mark.parametrize("cfg", SHAPES)
def test_forward_and_input_grad_bit_exact(oracle, cfg, kind, quantize, dtype):
    N, C, H, W, s, p = cfg
    rng = np.random.default_rng(seed_of(cfg, kind))
    x = rand(rng, (N, C, H, W), dtype)
    shift = special_shifts(rng, 2, C, dtype, kind)
    y_ref = oracle.rk2d_forward(x, shift, s, p, quantize)
    np.testing.assert_array_equal(_fwd(x, shift, s, p, quantize), y_ref)
    gy = rand(rng, y_ref.shape, dtype)
    gx_ref, _ = oracle.rk2d_backward(gy, x, shift, s, p, quantize=quantize)
    gx, _ = _bwd(gy, x, shift, s, p, quantize)
    np.testing.assert_array_equal(gx, gx_ref)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("kind", ["generic", "wide", "integer", "tiny"])
@pytest.mark.parametrize("cfg", SHAPES)
def test_shift_grad_matches_fp64_oracle(oracle, cfg, kind, dtype):
    N, C, H, W, s, p = cfg
    rng = np.random.default_rng(seed_of(cfg, kind, "g"))
    x = rand(rng, (N, C, H, W), dtype)
    shift = special_shifts(rng, 2, C, dtype, kind)
    gy = rand(rng, oracle.rk2d_forward(x, shift, s, p).shape, dtype)
    x64, s64, g64 = x.astype(np.float64), shift.astype(np.float64), gy.astype(np.float64)
    _, _, raw_ref = oracle.rk2d_backward(g64, x64, s64, s, p, normalize_grad=False, return_raw=True)
    _, raw = _bwd(gy, x, shift, s, p, False, normalize=False)
    scale = max(1.0, float(np.abs(raw_ref).max()))
    np.testing.assert_allclose(raw, raw_ref, rtol=0, atol=(1e-5 if dtype == np.float32 else 1e-12) * scale)
    _, g_ref = oracle.rk2d_backward(g64, x64, s64, s, p, normalize_grad=True)
    _, g = _bwd(gy, x, shift, s, p, False, normalize=True)
    np.testing.assert_allclose(g, g_ref, rtol=0, atol=2e-5 if dtype == np.float32 else 1e-11)


def test_quantize_leaves_user_buffer_untouched_out_of_range(oracle):
    x = np.arange(16, dtype=np.float32).reshape(1, 1, 4, 4) + 1
    shift = np.array([[2.0], [0.0]], np.float32)
    out = torch.full((1, 1, 4, 4), -7.0, device="cuda:0")
    y = _fwd(x, shift, 1, 0, True, output=out)
    ref = oracle.rk2d_forward(x, shift, 1, 0, True, output=np.full((1, 1, 4, 4), -7.0, np.float32))
    np.testing.assert_array_equal(y, ref)
    assert (y[0, 0, 2:] == -7.0).all()


@pytest.mark.parametrize("shape", [(2, 18, 9, 9), (6, 18, 12, 12)])   # generic kernels / streaming kernels
def test_enable_shift_grad_false_and_module(oracle, shape):
    from rubiksnet_amd.shiftlib import RubiksShift2D

    rng = np.random.default_rng(3)
    x = rand(rng, shape, np.float32)
    gy = rand(rng, x.shape, np.float32)
    shift = special_shifts(rng, 2, 18, np.float32, "integer")
    gx, gs = _bwd(gy, x, shift, 1, 0, False, enable=False)
    assert (gs == 0).all()
    gx_ref, _ = oracle.rk2d_backward(gy, x, shift, 1, 0, enable_shift_grad=False)
    np.testing.assert_array_equal(gx, gx_ref)
    mod = RubiksShift2D(18, init_shift="group3").to("cuda:0")        # integer shifts everywhere
    xt = to_dev(x).requires_grad_(True)
    mod(xt).backward(to_dev(gy))
    sh = to_np(mod.shift)
    _, g_ref = oracle.rk2d_backward(gy.astype(np.float64), x.astype(np.float64), sh.astype(np.float64))
    np.testing.assert_allclose(to_np(mod.shift.grad), g_ref, atol=2e-5)


@pytest.mark.parametrize("tdtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("kind", ["generic", "wide", "integer", "tiny"])
@pytest.mark.parametrize("shape", [(4, 12, 14, 14), (20, 6, 28, 28), (3, 5, 56, 56), (5, 3, 112, 112), (9, 4, 12, 16),
                                   (70, 16, 4, 8), (6, 3, 24, 64), (2, 2, 40, 200), (37, 6, 14, 14)])
def test_half_types_are_the_fp32_oracle_rounded_once(oracle, shape, kind, tdtype):
    """f16 / bf16 storage: the arithmetic is the fp32 operator's, rounded once on store.  So y and d(x)
    must equal the fp32 oracle on the widened inputs, rounded to the storage type -- bit for bit -- on the
    per-element kernels (W % 4 != 0) and on the streaming ones alike."""
    from rubiksnet_amd.shiftlib.rubiks2d.primitive import rubiks2d_backward, rubiks2d_forward

    rng = np.random.default_rng(seed_of(shape, kind, str(tdtype)))
    C = shape[1]
    x = torch.from_numpy(rand(rng, shape, np.float32)).to(tdtype)
    shift = torch.from_numpy(special_shifts(rng, 2, C, np.float32, kind)).to(tdtype)
    gy = torch.from_numpy(rand(rng, shape, np.float32)).to(tdtype)
    xf, sf, gf = x.float().numpy(), shift.float().numpy(), gy.float().numpy()
    y = rubiks2d_forward(x.cuda(), shift.cuda(), 1, 0)
    assert y.dtype == tdtype
    assert torch.equal(y.cpu(), torch.from_numpy(oracle.rk2d_forward(xf, sf, 1, 0)).to(tdtype))
    gx, gs = rubiks2d_backward(gy.cuda(), x.cuda(), shift.cuda(), 1, 0, normalize_grad=False)
    gx_ref, _ = oracle.rk2d_backward(gf, xf, sf, 1, 0)
    assert torch.equal(gx.cpu(), torch.from_numpy(gx_ref).to(tdtype))
    _, gs_ref = oracle.rk2d_backward(gf.astype(np.float64), xf.astype(np.float64), sf.astype(np.float64), 1, 0,
                                     normalize_grad=False)
    eps = 2e-3 if tdtype == torch.float16 else 1.6e-2
    scale = max(1.0, float(np.abs(gs_ref).max()))
    np.testing.assert_allclose(gs.float().cpu().numpy(), gs_ref, rtol=0, atol=eps * scale)
    gx2, gs2 = rubiks2d_backward(gy.cuda(), x.cuda(), shift.cuda(), 1, 0, enable_shift_grad=False)
    assert torch.equal(gx2, gx) and (gs2 == 0).all()


@pytest.mark.parametrize("tdtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("kind", ["generic", "wide", "integer", "half", "tiny"])
@pytest.mark.parametrize("cfg", [(4, 12, 14, 14, 1, 0), (20, 6, 28, 28, 1, 0), (3, 5, 56, 56, 1, 0), (6, 3, 24, 64, 1, 0),
                                 (3, 5, 28, 28, 2, 0), (16, 9, 7, 7, 1, 0), (2, 7, 9, 11, 2, 1), (9, 6, 14, 14, 2, 0)])
def test_half_activations_with_an_fp32_shift_table(oracle, cfg, kind, tdtype):
    """rk2d_*_sf32 (what autocast reaches: 16-bit activations, fp32 nn.Parameter): the shift is NOT rounded to the
    storage type, so y / d(x) are the fp32 oracle with the UNROUNDED shift on the widened inputs, rounded once -- bit
    for bit, `quantize` (whose +-0.5 side a bf16-rounded shift could flip) included -- and d(shift) comes back in fp32
    at fp32 accuracy."""
    from rubiksnet_amd.shiftlib.rubiks2d.primitive import rubiks2d_backward, rubiks2d_forward

    N, C, H, W, s, p = cfg
    rng = np.random.default_rng(seed_of(cfg, kind, str(tdtype), "sf32"))
    x = torch.from_numpy(rand(rng, (N, C, H, W), np.float32)).to(tdtype)
    sf = special_shifts(rng, 2, C, np.float32, kind)
    if kind == "generic":
        sf[0, 0], sf[1, 0] = 0.4990234375, -0.5009765625      # fp32 != their bf16 / f16 roundings (0.5 / -0.5)
    shift = torch.from_numpy(sf)
    xf = x.float().numpy()
    for q in (False, True):
        y = rubiks2d_forward(x.cuda(), shift.cuda(), s, p, quantize=q)
        y_ref = oracle.rk2d_forward(xf, sf, s, p, q)
        assert y.dtype == tdtype and torch.equal(y.cpu(), torch.from_numpy(y_ref).to(tdtype)), "forward quantize=%s" % q
        gy = torch.from_numpy(rand(rng, y_ref.shape, np.float32)).to(tdtype)
        gf = gy.float().numpy()
        gx, gs = rubiks2d_backward(gy.cuda(), x.cuda(), shift.cuda(), s, p, normalize_grad=False, quantize=q)
        gx_ref, _ = oracle.rk2d_backward(gf, xf, sf, s, p, quantize=q)
        assert torch.equal(gx.cpu(), torch.from_numpy(gx_ref).to(tdtype)), "d(x) quantize=%s" % q
        assert gs.dtype == torch.float32
        _, gs_ref = oracle.rk2d_backward(gf.astype(np.float64), xf.astype(np.float64), sf.astype(np.float64), s, p,
                                         normalize_grad=False)
        scale = max(1.0, float(np.abs(gs_ref).max()))
        np.testing.assert_allclose(gs.cpu().numpy(), gs_ref, rtol=0, atol=1e-5 * scale)
    gx2, gs2 = rubiks2d_backward(gy.cuda(), x.cuda(), shift.cuda(), s, p, enable_shift_grad=False, quantize=True)
    assert torch.equal(gx2, gx) and (gs2 == 0).all()


@pytest.mark.parametrize("tdtype", [torch.float16, torch.bfloat16])
def test_half_types_close_to_fp32_oracle(oracle, tdtype):
    """f16 (reference dispatches it, rubiks2d_kernels.cu:422) and bf16: computed in fp32, rounded on
    store -> compare with the fp32 oracle on the rounded inputs at the storage type's precision."""
    from rubiksnet_amd.shiftlib.rubiks2d.primitive import rubiks2d_backward, rubiks2d_forward

    rng = np.random.default_rng(9)
    x = torch.from_numpy(rand(rng, (4, 12, 14, 14), np.float32)).to(tdtype)
    shift = torch.from_numpy(special_shifts(rng, 2, 12, np.float32, "generic")).to(tdtype)
    gy = torch.from_numpy(rand(rng, (4, 12, 14, 14), np.float32)).to(tdtype)
    xf, sf, gf = x.float().numpy(), shift.float().numpy(), gy.float().numpy()
    eps = 2e-3 if tdtype == torch.float16 else 1.6e-2
    y = rubiks2d_forward(x.cuda(), shift.cuda(), 1, 0)
    assert y.dtype == tdtype
    np.testing.assert_allclose(y.float().cpu().numpy(), oracle.rk2d_forward(xf, sf, 1, 0), atol=eps)
    gx, gs = rubiks2d_backward(gy.cuda(), x.cuda(), shift.cuda(), 1, 0)
    gx_ref, gs_ref = oracle.rk2d_backward(gf, xf, sf, 1, 0)
    np.testing.assert_allclose(gx.float().cpu().numpy(), gx_ref, atol=eps)
    np.testing.assert_allclose(gs.float().cpu().numpy(), gs_ref, atol=eps)
    # quantize: a pure gather -> bit-exact against positions computed in the storage type
    yq = rubiks2d_forward(x.cuda(), shift.cuda(), 1, 0, quantize=True).cpu()
    h = torch.arange(14).view(14, 1).to(tdtype)
    for c in range(12):
        ph = (h + shift[0, c])
        pw = (h.view(1, 14) + shift[1, c])
        rnd = lambda v: torch.where(v < 0, (v - 0.5).to(tdtype), (v + 0.5).to(tdtype)).float().trunc().long()  # noqa: E731
        ih, iw = rnd(ph).expand(14, 14), rnd(pw).expand(14, 14)
        ok = (ih >= 0) & (ih < 14) & (iw >= 0) & (iw < 14)
        want = torch.where(ok, x[:, c][:, ih.clamp(0, 13), iw.clamp(0, 13)], torch.zeros((), dtype=tdtype))
        assert torch.equal(yq[:, c], want)
