"""GPU parity: RubiksShift3D through the C ABI (rubiksnet_amd.rubiksnet_cuda -> librubiks_hip.so)
versus the CPU oracle on identical seeded inputs.

Bars: forward and d(x) BIT-EXACT in fp32 and fp64 (both sides evaluate the reference's
expression tree with FP contraction off; quantize is a pure gather); d(shift) within
1e-5 (relative to the gradient's scale) of the oracle evaluated in fp64.
"""
import numpy as np
import pytest
import torch

from _util import rand, seed_of, special_shifts, to_dev, to_np

pytestmark = pytest.mark.gpu

SHAPES = [
    # N, T, C, H, W, stride, padding
    (2, 8, 16, 14, 14, (1, 1, 1), (0, 0, 0)),      # BASELINE configs[0] plumbing shape
    (2, 8, 6, 56, 56, (1, 1, 1), (0, 0, 0)),       # benchmark plane size
    (3, 8, 9, 7, 7, (1, 1, 1), (0, 0, 0)),         # small-plane regime (4 planes per workgroup)
    (1, 4, 5, 28, 28, (1, 2, 2), (0, 0, 0)),       # the networks' down-sampling layers
    (2, 3, 4, 9, 7, (1, 2, 2), (0, 1, 1)),
    (1, 6, 3, 10, 11, (2, 1, 3), (1, 2, 0)),       # stride/pad in T too
    (1, 1, 2, 5, 5, (1, 1, 1), (0, 0, 0)),         # T = 1
    (2, 8, 3, 112, 112, (1, 2, 2), (0, 0, 0)),     # first stage of the nets (stride-2 streaming kernels, 4 bands)
    (2, 3, 4, 56, 56, (1, 2, 2), (0, 0, 0)),       # stride-2 streaming, one band
    (1, 2, 5, 16, 24, (1, 2, 2), (0, 0, 0)),       # stride-2 streaming, small ragged plane (Wo4 = 3)
    (1, 4, 3, 112, 112, (1, 1, 1), (0, 0, 0)),     # 112x112 stride 1: 4 row bands on the streaming path
    (1, 3, 2, 96, 64, (1, 1, 1), (0, 0, 0)),       # 4 bands of 24 rows, 16 float4 columns
    (2, 3, 5, 60, 56, (1, 1, 1), (0, 0, 0)),       # one band, ragged last round
    (2, 4, 24, 7, 7, (1, 1, 1), (0, 0, 0)),        # tile kernels: 7x7, channel groups of 16 + 8
    (3, 5, 6, 14, 14, (1, 1, 1), (0, 0, 0)),       # tile kernels: 14x14, channel groups of 4 + 2, odd T
    (2, 8, 44, 7, 7, (1, 1, 1), (0, 0, 0)),        # slab kernels (rk3d_slab.hpp): 7x7, 3 chunks, planes straddling chunk edges
    (1, 8, 576, 7, 7, (1, 1, 1), (0, 0, 0)),       # layer4 of RubiksNet-Large: 28 chunks, the last one ragged
    (2, 5, 8, 7, 8, (1, 1, 1), (0, 0, 0)),         # slab: one ragged chunk, odd T, H != W
    (2, 3, 8, 13, 15, (1, 1, 1), (0, 0, 0)),       # slab: the widest plane it takes (halo of 4 cells)
    (2, 8, 12, 28, 28, (1, 2, 2), (0, 0, 0)),      # slab, stride (1,2,2) backward: 28 -> 14, planes straddling chunks
    (2, 8, 24, 14, 14, (1, 2, 2), (0, 0, 0)),      # slab, stride (1,2,2) backward: 14 -> 7
    (1, 5, 4, 10, 12, (1, 2, 2), (0, 0, 0)),       # slab, stride (1,2,2) backward: ragged chunk, odd T, H != W
]
KINDS = ["generic", "wide", "integer", "half", "oob"]


def _run_fwd(x, shift, s, p, q):
    from rubiksnet_amd.shiftlib.rubiks3d.primitive import rubiks_shift_3d_forward
    return to_np(rubiks_shift_3d_forward(to_dev(x), to_dev(shift), s, p, quantize=q))


def _run_bwd(gy, x, shift, s, p, q, normalize=True, tf=1.0):
    from rubiksnet_amd.shiftlib.rubiks3d.primitive import rubiks_shift_3d_backward
    gx, gs = rubiks_shift_3d_backward(to_dev(gy), to_dev(x), to_dev(shift), s, p, normalize,
                                      normalize_t_factor=tf, quantize=q)
    return to_np(gx), to_np(gs)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("quantize", [False, True])
@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("cfg", SHAPES)
def test_forward_and_input_grad_bit_exact(oracle, cfg, kind, quantize, dtype):
    N, T, C, H, W, s, p = cfg
    rng = np.random.default_rng(seed_of(cfg, kind))
    x = rand(rng, (N, T, C, H, W), dtype)
    shift = special_shifts(rng, 3, C, dtype, kind)
    y_ref = oracle.rk3d_forward(x, shift, s, p, quantize)
    y = _run_fwd(x, shift, s, p, quantize)
    assert y.shape == y_ref.shape
    np.testing.assert_array_equal(y, y_ref)
    gy = rand(rng, y_ref.shape, dtype)
    gx_ref, _ = oracle.rk3d_backward(gy, x, shift, s, p, quantize=quantize)
    gx, _ = _run_bwd(gy, x, shift, s, p, quantize)
    np.testing.assert_array_equal(gx, gx_ref)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("kind", ["generic", "wide", "integer", "oob"])
@pytest.mark.parametrize("cfg", SHAPES)
def test_shift_grad_matches_fp64_oracle(oracle, cfg, kind, dtype):
    N, T, C, H, W, s, p = cfg
    rng = np.random.default_rng(seed_of(cfg, kind, "g"))
    x = rand(rng, (N, T, C, H, W), dtype)
    shift = special_shifts(rng, 3, C, dtype, kind)
    gy = rand(rng, oracle.rk3d_forward(x, shift, s, p).shape, dtype)
    x64, s64, g64 = x.astype(np.float64), shift.astype(np.float64), gy.astype(np.float64)
    _, _, raw_ref = oracle.rk3d_backward(g64, x64, s64, s, p, normalize_grad=False, return_raw=True)
    _, raw = _run_bwd(gy, x, shift, s, p, False, normalize=False)
    scale = max(1.0, float(np.abs(raw_ref).max()))
    tol = 1e-5 if dtype == np.float32 else 1e-12
    np.testing.assert_allclose(raw, raw_ref, rtol=0, atol=tol * scale)
    # normalised (K5), default t_factor and the two other branches
    for tf in (1.0, 0.25, -1.0):
        _, g_ref = oracle.rk3d_backward(g64, x64, s64, s, p, normalize_grad=True, normalize_t_factor=tf)
        _, g = _run_bwd(gy, x, shift, s, p, False, normalize=True, tf=tf)
        np.testing.assert_allclose(g, g_ref, rtol=0, atol=(2e-5 if dtype == np.float32 else 1e-11))


@pytest.mark.parametrize("cfg", [(2, 4, 6, 56, 56, (1, 1, 1), (0, 0, 0)), (3, 5, 6, 14, 14, (1, 1, 1), (0, 0, 0)),
                                 (2, 3, 5, 60, 56, (1, 1, 1), (0, 0, 0)), (2, 8, 16, 14, 14, (1, 2, 2), (0, 1, 1))])
@pytest.mark.parametrize("kind", ["generic", "wide", "half"])
def test_quantize_backward_both_halves(oracle, cfg, kind):
    """quantize=True backward: d(x) is the nearest-position translation (bit-exact), d(shift) is K2 on the fractional
    shift (the reference's K2 takes no quantize flag).  On the streaming shapes both come from one launch (the QUANT
    walk of rk3d_dma.hpp); d(x) alone is rk3d_translate.hpp, d(shift) alone the shift-only streaming backward."""
    N, T, C, H, W, s, p = cfg
    rng = np.random.default_rng(seed_of(cfg, kind, "q"))
    x = rand(rng, (N, T, C, H, W), np.float32)
    shift = special_shifts(rng, 3, C, np.float32, kind)
    gy = rand(rng, oracle.rk3d_forward(x, shift, s, p, True).shape, np.float32)
    gx_ref, _ = oracle.rk3d_backward(gy, x, shift, s, p, quantize=True)
    _, _, raw_ref = oracle.rk3d_backward(gy.astype(np.float64), x.astype(np.float64), shift.astype(np.float64), s, p,
                                         normalize_grad=False, quantize=True, return_raw=True)
    gx, raw = _run_bwd(gy, x, shift, s, p, True, normalize=False)
    np.testing.assert_array_equal(gx, gx_ref)
    scale = max(1.0, float(np.abs(raw_ref).max()))
    np.testing.assert_allclose(raw, raw_ref, rtol=0, atol=1e-5 * scale)
    from rubiksnet_amd.shiftlib.rubiks3d.primitive import rubiks_shift_3d_backward
    gx_only, none_s = rubiks_shift_3d_backward(to_dev(gy), to_dev(x), to_dev(shift), s, p, False, quantize=True,
                                               need_shift_grad=False)
    assert none_s is None
    np.testing.assert_array_equal(to_np(gx_only), gx_ref)


def test_autograd_function_and_module(oracle):
    """rubiks_shift_3d (Function) and RubiksShift3D (Module) wire the same numbers through autograd."""
    from rubiksnet_amd.shiftlib import RubiksShift3D
    from rubiksnet_amd.shiftlib.rubiks3d.primitive import RubiksShift3DFunc, rubiks_shift_3d

    rng = np.random.default_rng(11)
    x = rand(rng, (2, 8, 16, 14, 14), np.float32)
    gy = rand(rng, x.shape, np.float32)
    mod = RubiksShift3D(16).to("cuda:0")
    shift = to_np(mod.shift)
    xt = to_dev(x).requires_grad_(True)
    y = mod(xt)
    y.backward(to_dev(gy))
    np.testing.assert_array_equal(to_np(y), oracle.rk3d_forward(x, shift))
    gx_ref, gs_ref = oracle.rk3d_backward(gy.astype(np.float64), x.astype(np.float64), shift.astype(np.float64))
    np.testing.assert_allclose(to_np(xt.grad), gx_ref, atol=1e-6)
    np.testing.assert_allclose(to_np(mod.shift.grad), gs_ref, atol=2e-5)
    # Function returns (x_grad, shift_grad, None x 5); "auto" t-factor = T / H
    y2 = rubiks_shift_3d(xt, mod.shift, normalize_t_factor="auto")
    assert y2.grad_fn is not None and RubiksShift3DFunc.__name__ == "RubiksShift3DFunc"
    mod.shift.grad = None
    y2.backward(to_dev(gy))
    _, gs_auto = oracle.rk3d_backward(gy.astype(np.float64), x.astype(np.float64), shift.astype(np.float64),
                                      normalize_t_factor=8 / 14)
    np.testing.assert_allclose(to_np(mod.shift.grad), gs_auto, atol=2e-5)
    # frozen shift: only d(x) is produced
    xt2 = to_dev(x).requires_grad_(True)
    rubiks_shift_3d(xt2, mod.shift.detach()).backward(to_dev(gy))
    np.testing.assert_allclose(to_np(xt2.grad), gx_ref, atol=1e-6)


def test_gradcheck_fp64():
    """BASELINE configs[1]: gradcheck of the op in fp64 (shifts kept away from integers, raw grads)."""
    from rubiksnet_amd.shiftlib.rubiks3d.primitive import rubiks_shift_3d

    torch.manual_seed(0)
    x = (torch.rand(1, 3, 2, 4, 5, dtype=torch.float64, device="cuda:0") * 2 - 1).requires_grad_(True)
    shift = torch.tensor([[0.3, -0.6], [0.45, 1.2], [-0.7, 0.15]], dtype=torch.float64, device="cuda:0",
                         requires_grad=True)
    for stride, pad in [(1, 0), ((1, 2, 2), (0, 1, 1))]:
        fn = lambda a, b: rubiks_shift_3d(a, b, stride, pad, False, 1.0, False)  # noqa: E731
        assert torch.autograd.gradcheck(fn, (x, shift), eps=1e-6, atol=1e-7, nondet_tol=0.0)


def test_two_phase_backward_equals_the_one_call_form():
    """rk3d_backward_partials_f32 + rk3d_backward_finalize_f32 (the phases of rubiks.cpp:324-376) == rk3d_backward_f32,
    bit for bit, on a streaming shape, a tile shape, a strided column shape and a strided streaming shape."""
    import ctypes

    from rubiksnet_amd import _native, rubiksnet_cuda

    L = _native.lib()
    for (N, T, C, H, W), s in (((2, 8, 6, 56, 56), (1, 1, 1)), ((2, 4, 8, 14, 14), (1, 1, 1)), ((1, 4, 5, 28, 28), (1, 2, 2)),
                                ((2, 3, 4, 56, 56), (1, 2, 2))):
        torch.manual_seed(N * C + H)
        x = torch.rand(N, T, C, H, W, device="cuda:0") * 2 - 1
        shift = torch.rand(3, C, device="cuda:0") * 2 - 1
        Ho, Wo = (H - 1) // s[1] + 1, (W - 1) // s[2] + 1
        gy = torch.rand(N, T, C, Ho, Wo, device="cuda:0") * 2 - 1
        gx1, gs1 = torch.empty_like(x), torch.empty_like(shift)
        rubiksnet_cuda.rubiks_shift_3d_backward_float(x, shift, gy, list(s), [0, 0, 0], gx1, gs1, True, 0.5, False)
        nbytes = int(L.rk3d_backward_workspace_bytes(N, T, C, H, W, *s, 0, 0, 0, 4))
        ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda:0")
        gx2, gs2, P = torch.empty_like(x), torch.empty_like(shift), ctypes.c_int(0)
        st = torch.cuda.current_stream().cuda_stream
        _native.check(L.rk3d_backward_partials_f32(x.data_ptr(), shift.data_ptr(), gy.data_ptr(), gx2.data_ptr(), N, T, C, H,
                                                   W, *s, 0, 0, 0, 0, ws.data_ptr(), nbytes, ctypes.byref(P), st), "partials")
        assert P.value > 0
        _native.check(L.rk3d_backward_finalize_f32(ws.data_ptr(), C, P.value, gs2.data_ptr(), 1, 0.5, st), "finalize")
        assert torch.equal(gx1, gx2) and torch.equal(gs1, gs2)


@pytest.mark.parametrize("shape,stride", [((2, 8, 6, 56, 56), 1), ((2, 4, 8, 14, 14), 1), ((2, 3, 4, 56, 56), 2),
                                          ((3, 4, 5, 28, 28), 1)])
def test_backward_ignores_adversarial_workspace_contents(shape, stride):
    """The in-launch row-sum polls an UNINITIALISED workspace for granules stamped with this launch's tag (rk_dma.hpp).
    Round-2 advisor finding: with one 32-bit tag per granule a stale word equal to the tag passes for a finished
    partial.  Pre-fill the workspace with exactly such near misses -- the right tag on the value granule, on the check
    granule, on both but with payloads that do not agree, valid pairs for the neighbouring tags -- and demand the same
    bits as with a zeroed workspace."""
    from rubiksnet_amd import _native

    L = _native.lib()
    N, T, C, H, W = shape
    s = (1, stride, stride)
    torch.manual_seed(sum(shape))
    x = torch.rand(shape, device="cuda:0") * 2 - 1
    shift = torch.rand(3, C, device="cuda:0") * 2 - 1
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    gy = torch.rand(N, T, C, Ho, Wo, device="cuda:0") * 2 - 1
    nbytes = int(L.rk3d_backward_workspace_bytes(N, T, C, H, W, *s, 0, 0, 0, 4))
    st = torch.cuda.current_stream().cuda_stream

    def run(ws):
        gx, gs = torch.empty_like(x), torch.empty_like(shift)
        _native.check(L.rk3d_backward_f32(x.data_ptr(), shift.data_ptr(), gy.data_ptr(), gx.data_ptr(), gs.data_ptr(), N, T,
                                          C, H, W, *s, 0, 0, 0, 1, 1.0, 0, ws.data_ptr(), nbytes, st), "rk3d_backward_f32")
        torch.cuda.synchronize()
        return gx, gs

    gx0, gs0 = run(torch.zeros(nbytes, dtype=torch.uint8, device="cuda:0"))
    assert torch.isfinite(gs0).all()
    rng = np.random.default_rng(1)
    npairs = nbytes // 16
    for mode in range(5):
        tag = int(L.rk_debug_peek_launch_tag())                    # the tag run() below will use
        tag2 = ((tag * 2654435761) & 0xffffffff) ^ 0x9e3779b9
        v = rng.integers(0, 2 ** 32, npairs, dtype=np.uint64)
        w = rng.integers(0, 2 ** 32, npairs, dtype=np.uint64)
        inv = (~v) & np.uint64(0xffffffff)
        pairs = np.empty((npairs, 2), np.uint64)
        if mode == 0:      # the old failure mode: every value granule carries the right tag, the check granule is junk
            pairs[:, 0], pairs[:, 1] = (np.uint64(tag) << np.uint64(32)) | v, w << np.uint64(32) | v
        elif mode == 1:    # both tags right, payloads do not agree
            pairs[:, 0], pairs[:, 1] = (np.uint64(tag) << np.uint64(32)) | v, (np.uint64(tag2) << np.uint64(32)) | w
        elif mode == 2:    # payloads agree, the check granule carries the VALUE tag
            pairs[:, 0], pairs[:, 1] = (np.uint64(tag) << np.uint64(32)) | v, (np.uint64(tag) << np.uint64(32)) | inv
        elif mode == 3:    # complete, consistent pairs -- of the previous launch (never retired, say)
            t, t2 = (tag - 1) & 0xffffffff, (((tag - 1) * 2654435761) & 0xffffffff) ^ 0x9e3779b9
            pairs[:, 0], pairs[:, 1] = (np.uint64(t) << np.uint64(32)) | v, (np.uint64(t2) << np.uint64(32)) | inv
        else:              # ... and of the next one
            t, t2 = (tag + 1) & 0xffffffff, (((tag + 1) * 2654435761) & 0xffffffff) ^ 0x9e3779b9
            pairs[:, 0], pairs[:, 1] = (np.uint64(t) << np.uint64(32)) | v, (np.uint64(t2) << np.uint64(32)) | inv
        ws = torch.zeros(nbytes, dtype=torch.uint8, device="cuda:0")
        ws[:npairs * 16] = torch.from_numpy(pairs.view(np.uint8).reshape(-1)).cuda()
        gx, gs = run(ws)
        assert torch.equal(gx, gx0) and torch.equal(gs, gs0), "mode %d" % mode


def test_errors_are_raised_not_fatal():
    from rubiksnet_amd import rubiksnet_cuda
    from rubiksnet_amd.shiftlib.rubiks3d.primitive import rubiks_shift_3d_forward

    x = torch.zeros(1, 2, 3, 4, 4, device="cuda:0")
    sh = torch.zeros(3, 3, device="cuda:0")
    with pytest.raises(ValueError):
        rubiks_shift_3d_forward(x.half(), sh.half(), 1, 0)
    with pytest.raises(AssertionError):
        rubiks_shift_3d_forward(x.cpu(), sh.cpu(), 1, 0)
    with pytest.raises(RuntimeError):      # non-contiguous input reaches the binding
        rubiksnet_cuda.rubiks_shift_3d_forward_float(x.transpose(3, 4), sh, [1, 1, 1], [0, 0, 0], False, x.clone())
    with pytest.raises(RuntimeError):      # bad stride -> RK_ERR_BAD_STRIDE, not a crash
        rubiksnet_cuda.rubiks_shift_3d_forward_float(x, sh, [0, 1, 1], [0, 0, 0], False, x.clone())


# ----------------------------------------------------------------- BASELINE full size
FULL = (32, 8, 64, 56, 56)


@pytest.fixture(scope="module")
def full():
    g = torch.Generator(device="cuda:0").manual_seed(0)
    x = torch.rand(FULL, device="cuda:0", generator=g) * 2 - 1
    gy = torch.rand(FULL, device="cuda:0", generator=g) * 2 - 1
    shift = torch.rand(3, 64, device="cuda:0", generator=g) * 2 - 1
    return x, gy, shift


def test_full_size_slices_bit_exact_vs_oracle(oracle, full):
    """(32,8,64,56,56): clips are independent, so clip 0 / 17 / 31 of the full-size result must equal
    the oracle run on that clip alone."""
    from rubiksnet_amd.shiftlib.rubiks3d.primitive import rubiks_shift_3d_backward, rubiks_shift_3d_forward

    x, gy, shift = full
    y = rubiks_shift_3d_forward(x, shift, 1, 0)
    gx, _ = rubiks_shift_3d_backward(gy, x, shift, 1, 0, True)
    sh = to_np(shift)
    for n in (0, 17, 31):
        xn, gn = to_np(x[n:n + 1]), to_np(gy[n:n + 1])
        np.testing.assert_array_equal(to_np(y[n:n + 1]), oracle.rk3d_forward(xn, sh))
        gx_ref, _ = oracle.rk3d_backward(gn, xn, sh)
        np.testing.assert_array_equal(to_np(gx[n:n + 1]), gx_ref)


def test_full_size_adjoint_and_linearity(full):
    """Size-independent properties at full size: <fwd(x), gy> == <x, d(x)>, and raw d(shift) is additive
    over a partition of the batch."""
    from rubiksnet_amd.shiftlib.rubiks3d.primitive import rubiks_shift_3d_backward, rubiks_shift_3d_forward

    x, gy, shift = full
    y = rubiks_shift_3d_forward(x, shift, 1, 0)
    gx, raw = rubiks_shift_3d_backward(gy, x, shift, 1, 0, False)
    lhs = torch.dot(y.double().flatten(), gy.double().flatten()).item()
    rhs = torch.dot(x.double().flatten(), gx.double().flatten()).item()
    assert lhs == pytest.approx(rhs, rel=1e-6)
    parts = torch.zeros_like(raw, dtype=torch.float64)
    for lo in range(0, 32, 8):
        _, r = rubiks_shift_3d_backward(gy[lo:lo + 8].contiguous(), x[lo:lo + 8].contiguous(), shift, 1, 0, False)
        parts += r.double()
    np.testing.assert_allclose(to_np(raw), to_np(parts), rtol=0, atol=1e-5 * float(parts.abs().max()))


def test_full_size_shift_grad_vs_oracle_subbatch(oracle, full):
    from rubiksnet_amd.shiftlib.rubiks3d.primitive import rubiks_shift_3d_backward

    x, gy, shift = full
    xs, gs = x[:4].contiguous(), gy[:4].contiguous()
    _, g = rubiks_shift_3d_backward(gs, xs, shift, 1, 0, True)
    _, g_ref = oracle.rk3d_backward(to_np(gs).astype(np.float64), to_np(xs).astype(np.float64),
                                    to_np(shift).astype(np.float64))
    np.testing.assert_allclose(to_np(g), g_ref, rtol=0, atol=1e-5)


def test_full_size_quantize_is_translation(full):
    """quantize=True at full size equals an independent torch translate (pad + slice) per channel."""
    from rubiksnet_amd.shiftlib.rubiks3d.primitive import rubiks_shift_3d_forward

    x, _, shift = full
    y = rubiks_shift_3d_forward(x, shift, 1, 0, quantize=True)
    sh = to_np(shift)
    fl = np.floor(sh.astype(np.float32)).astype(np.int64)
    q = np.where((sh - fl) < 0.5, fl, fl + 1)
    P = 3
    xp = torch.nn.functional.pad(x, (P, P, P, P, 0, 0, P, P))      # pad W, H, (C none), T
    for c in (0, 5, 33, 63):
        qt, qh, qw = (int(v) for v in q[:, c])
        want = xp[:, P + qt:P + qt + 8, c, P + qh:P + qh + 56, P + qw:P + qw + 56]
        assert torch.equal(y[:, :, c], want)


@pytest.mark.parametrize("stride", [1, (1, 2, 2)])
@pytest.mark.parametrize("kind", ["generic", "integer"])
def test_backward_halves_match_fused(oracle, kind, stride):
    """d(x)-only, d(shift)-only and the fused backward agree (different kernels on the streaming paths)."""
    from rubiksnet_amd.shiftlib.rubiks3d.primitive import rubiks_shift_3d_backward

    rng = np.random.default_rng(21)
    x = rand(rng, (3, 8, 10, 28, 28) if stride == 1 else (3, 5, 6, 56, 56), np.float32)
    shift = special_shifts(rng, 3, x.shape[2], np.float32, kind)
    gy = rand(rng, oracle.rk3d_forward(x, shift, stride, 0).shape, np.float32)
    args = (to_dev(gy), to_dev(x), to_dev(shift), stride, 0, False)
    gx_f, gs_f = rubiks_shift_3d_backward(*args)
    gx_only, none_s = rubiks_shift_3d_backward(*args, need_shift_grad=False)
    none_x, gs_only = rubiks_shift_3d_backward(*args, need_x_grad=False)
    assert none_s is None and none_x is None
    np.testing.assert_array_equal(to_np(gx_f), to_np(gx_only))
    _, _, raw_ref = oracle.rk3d_backward(gy.astype(np.float64), x.astype(np.float64), shift.astype(np.float64),
                                         stride, 0, normalize_grad=False, return_raw=True)
    scale = max(1.0, float(np.abs(raw_ref).max()))
    np.testing.assert_allclose(to_np(gs_f), raw_ref, rtol=0, atol=1e-5 * scale)
    np.testing.assert_allclose(to_np(gs_only), raw_ref, rtol=0, atol=1e-5 * scale)


@pytest.mark.parametrize("hw", [(56, 56), (28, 28), (112, 112), (14, 14)])
@pytest.mark.parametrize("wide", [False, True])
def test_tsm_init_integer_temporal_shifts(oracle, hw, wide):
    """create_3d_from_2d(init_mode="tsm") (reference layer.py:137-141): EVERY channel has an exactly-integer temporal
    shift (+1 / -1 / 0 folds, here also +-2, +-3 when `wide`) next to fractional (H, W) shifts.  The streaming
    backward walks such a channel twice with the lowered-index pairing (rk3d_dma.hpp) instead of per element."""
    H, W = hw
    rng = np.random.default_rng(H * 131 + wide)
    C = 16
    x = rand(rng, (2, 5, C, H, W), np.float32)
    gy = rand(rng, x.shape, np.float32)
    shift = rng.uniform(-1.7, 1.7, (3, C)).astype(np.float32)
    folds = np.array([1.0, -1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0] * 2, np.float32)
    shift[0] = folds if not wide else rng.choice([-3.0, -2.0, -1.0, 0.0, 1.0, 2.0, 3.0, 5.0], C)
    np.testing.assert_array_equal(_run_fwd(x, shift, 1, 0, False), oracle.rk3d_forward(x, shift, 1, 0, False))
    gx_ref, _, raw_ref = oracle.rk3d_backward(gy.astype(np.float64), x.astype(np.float64), shift.astype(np.float64),
                                              normalize_grad=False, return_raw=True)
    gx32_ref, _ = oracle.rk3d_backward(gy, x, shift, 1, 0)
    gx, gs = _run_bwd(gy, x, shift, 1, 0, False, normalize=False)
    np.testing.assert_array_equal(gx, gx32_ref)
    scale = max(1.0, float(np.abs(raw_ref).max()))
    np.testing.assert_allclose(gs, raw_ref, rtol=0, atol=1e-5 * scale)
    from rubiksnet_amd.shiftlib.rubiks3d.primitive import rubiks_shift_3d_backward
    _, gs_only = rubiks_shift_3d_backward(to_dev(gy), to_dev(x), to_dev(shift), 1, 0, False, need_x_grad=False)
    np.testing.assert_allclose(to_np(gs_only), raw_ref, rtol=0, atol=1e-5 * scale)
    # quantize=True on the same shifts (one launch for both gradients: the QUANT walk of rk3d_dma.hpp), plus
    # remainders of exactly 0.5 -- the rounding boundary of the nearest tap
    shift[1, ::3] = np.floor(shift[1, ::3]) + 0.5
    shift[2, 1::4] = np.floor(shift[2, 1::4]) + 0.5
    gxq_ref, _ = oracle.rk3d_backward(gy, x, shift, 1, 0, quantize=True)
    _, _, rawq_ref = oracle.rk3d_backward(gy.astype(np.float64), x.astype(np.float64), shift.astype(np.float64),
                                          normalize_grad=False, quantize=True, return_raw=True)
    gxq, gsq = _run_bwd(gy, x, shift, 1, 0, True, normalize=False)
    np.testing.assert_array_equal(gxq, gxq_ref)
    np.testing.assert_allclose(gsq, rawq_ref, rtol=0, atol=1e-5 * max(1.0, float(np.abs(rawq_ref).max())))


def test_empty_batch_and_single_element(oracle):
    """Edge sizes: N = 0 (the reference launches over zero elements and returns an empty tensor) and 1x1x1 planes."""
    from rubiksnet_amd.shiftlib.rubiks3d.primitive import rubiks_shift_3d, rubiks_shift_3d_forward

    shn = np.array([[0.3, -0.2], [0.1, 0.7], [-0.4, 0.2]], np.float32)
    sh = to_dev(shn)
    y = rubiks_shift_3d_forward(torch.zeros(0, 4, 2, 5, 5, device="cuda:0"), sh, 1, 0)
    assert y.shape == (0, 4, 2, 5, 5)
    x = torch.zeros(0, 4, 2, 5, 5, device="cuda:0", requires_grad=True)
    shp = sh.clone().requires_grad_(True)
    rubiks_shift_3d(x, shp).sum().backward()
    assert x.grad.shape == x.shape and torch.equal(shp.grad, torch.zeros_like(shp))
    x1 = np.full((1, 1, 2, 1, 1), 2.0, np.float32)        # T = H = W = 1: at most one tap is in range
    np.testing.assert_array_equal(to_np(rubiks_shift_3d_forward(to_dev(x1), sh, 1, 0)), oracle.rk3d_forward(x1, shn))


@pytest.mark.parametrize("shape", [(2, 8, 6, 56, 56), (2, 4, 8, 14, 14), (2, 3, 4, 56, 56)])
def test_backward_replayed_from_a_graph(shape):
    """The boundary allocates and synchronises nothing, so a call can be captured in a hipGraph.  A replay repeats the
    launch WITH THE SAME granule tag: the in-launch row-sum must not take the previous replay's partials for its own
    (rk_dma.hpp: consumed granules are retired)."""
    from rubiksnet_amd.shiftlib.rubiks3d.primitive import rubiks_shift_3d_backward

    stride = (1, 2, 2) if shape == (2, 3, 4, 56, 56) else 1
    torch.manual_seed(5)
    N, T, C, H, W = shape
    x = torch.rand(shape, device="cuda:0") * 2 - 1
    shift = torch.rand(3, C, device="cuda:0") * 2 - 1
    Ho, Wo = ((H - 1) // 2 + 1, (W - 1) // 2 + 1) if stride != 1 else (H, W)
    gy = torch.rand(N, T, C, Ho, Wo, device="cuda:0") * 2 - 1
    rubiks_shift_3d_backward(gy, x, shift, stride, 0, True)              # warm up outside the capture
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            gx_g, gs_g = rubiks_shift_3d_backward(gy, x, shift, stride, 0, True)
    for it in range(4):
        x.copy_(torch.rand_like(x) * 2 - 1)
        gy.copy_(torch.rand_like(gy) * (it + 1))
        torch.cuda.synchronize()
        graph.replay()
        torch.cuda.synchronize()
        gx_e, gs_e = rubiks_shift_3d_backward(gy, x, shift, stride, 0, True)
        assert torch.equal(gx_g, gx_e) and torch.equal(gs_g, gs_e), it
