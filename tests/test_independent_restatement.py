"""CPU: the C oracle against tests/_independent.py -- a second restatement written from SURVEY 7.1 (vectorised
PyTorch fp64 + autograd) -- on every shape x shift-kind the GPU parity suites use, plus closed-form cases for the
reference's quirks, plus INTEGRATION.md's claim that the reference's own rubiksnet/shiftlib binds to our module.

The oracle stays "parity unpinned by reference execution"; what these tests buy is that the bit-exact GPU claims
no longer rest on ONE reading of the reference."""
import importlib
import os
import sys

import numpy as np
import pytest
import torch

import _independent as ind
from _util import seed_of, special_shifts

SHAPES3 = [
    # the shapes of tests/test_parity_3d.py (planes cropped where only the plane SIZE differs: the loops are per channel)
    (2, 8, 16, 14, 14, (1, 1, 1), (0, 0, 0)),
    (2, 8, 6, 56, 56, (1, 1, 1), (0, 0, 0)),
    (3, 8, 9, 7, 7, (1, 1, 1), (0, 0, 0)),
    (1, 4, 5, 28, 28, (1, 2, 2), (0, 0, 0)),
    (2, 3, 4, 9, 7, (1, 2, 2), (0, 1, 1)),
    (1, 6, 3, 10, 11, (2, 1, 3), (1, 2, 0)),
    (1, 1, 2, 5, 5, (1, 1, 1), (0, 0, 0)),
    (2, 8, 3, 112, 112, (1, 2, 2), (0, 0, 0)),
    (2, 3, 4, 56, 56, (1, 2, 2), (0, 0, 0)),
    (1, 2, 5, 16, 24, (1, 2, 2), (0, 0, 0)),
    (1, 4, 3, 112, 112, (1, 1, 1), (0, 0, 0)),
    (1, 3, 2, 96, 64, (1, 1, 1), (0, 0, 0)),
    (2, 3, 5, 60, 56, (1, 1, 1), (0, 0, 0)),
    (2, 4, 24, 7, 7, (1, 1, 1), (0, 0, 0)),
    (3, 5, 6, 14, 14, (1, 1, 1), (0, 0, 0)),
    (2, 8, 44, 7, 7, (1, 1, 1), (0, 0, 0)),
    (1, 8, 576, 7, 7, (1, 1, 1), (0, 0, 0)),
    (2, 5, 8, 7, 8, (1, 1, 1), (0, 0, 0)),
    (2, 3, 8, 13, 15, (1, 1, 1), (0, 0, 0)),
    (2, 8, 12, 28, 28, (1, 2, 2), (0, 0, 0)),
    (2, 8, 24, 14, 14, (1, 2, 2), (0, 0, 0)),
    (1, 5, 4, 10, 12, (1, 2, 2), (0, 0, 0)),
]
KINDS3 = ["generic", "wide", "integer", "half", "oob"]
SHAPES2 = [
    (2, 16, 14, 14, 1, 0),
    (3, 5, 28, 28, 2, 0),
    (2, 7, 9, 11, 2, 1),
    (1, 4, 10, 7, (1, 3), (2, 0)),
    (16, 9, 7, 7, 1, 0),
    (2, 3, 56, 56, 1, 0),
    (20, 6, 28, 28, 1, 0),
    (9, 4, 12, 16, 1, 0),
    (5, 3, 112, 112, 1, 0),
    (70, 512, 4, 8, 1, 0),
    (37, 6, 14, 14, 1, 0),
]
KINDS2 = ["generic", "wide", "integer", "half", "oob", "tiny"]
TOL2D = float(np.float32(1e-7))        # ZERO_TOL = static_cast<T>(1e-7f), rubiks2d_kernels.cu:189


def test_shape_lists_are_the_gpu_suites():
    """Same (shape, stride, pad) lists and shift kinds as the GPU parity tests."""
    import test_parity_2d
    import test_parity_3d

    assert SHAPES3 == test_parity_3d.SHAPES and set(test_parity_3d.KINDS) <= set(KINDS3)
    assert SHAPES2 == test_parity_2d.SHAPES


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64))


# --------------------------------------------------------------------------------------------------- 3-D
@pytest.mark.parametrize("quantize", [False, True])
@pytest.mark.parametrize("kind", KINDS3)
@pytest.mark.parametrize("cfg", SHAPES3)
def test_3d_oracle_equals_independent_restatement(oracle, cfg, kind, quantize):
    N, T, C, H, W, s, p = cfg
    rng = np.random.default_rng(seed_of(cfg, kind, "indep"))
    x = rng.uniform(-1, 1, (N, T, C, H, W))
    shift = special_shifts(rng, 3, C, np.float64, kind)
    y_o = oracle.rk3d_forward(x, shift, s, p, quantize)
    y_i = ind.shift3d_forward(_t(x), _t(shift), s, p, quantize).numpy()
    assert y_o.shape == y_i.shape
    if quantize:
        np.testing.assert_array_equal(y_o, y_i)                     # pure gather: identical values
    else:
        np.testing.assert_allclose(y_o, y_i, rtol=0, atol=1e-13)
    gy = rng.uniform(-1, 1, y_o.shape)
    gx_o, _, raw_o = oracle.rk3d_backward(gy, x, shift, s, p, normalize_grad=False, quantize=quantize, return_raw=True)
    gx_i = ind.shift3d_input_grad(_t(gy), x.shape, _t(shift), s, p, quantize).numpy()
    if quantize:
        np.testing.assert_array_equal(gx_o, gx_i)
    else:
        np.testing.assert_allclose(gx_o, gx_i, rtol=0, atol=1e-13)
    # d(shift): K2 has no quantize argument (straight-through) -> the same numbers either way
    raw_i = ind.shift3d_shift_grad(_t(gy), _t(x), _t(shift), s, p).numpy()
    scale = max(1.0, float(np.abs(raw_o).max()))
    np.testing.assert_allclose(raw_o, raw_i, rtol=0, atol=1e-11 * scale)
    for tf in (1.0, 0.25, -1.0):
        _, g_o = oracle.rk3d_backward(gy, x, shift, s, p, normalize_grad=True, normalize_t_factor=tf)
        np.testing.assert_allclose(g_o, ind.normalize3d(_t(raw_i), tf).numpy(), rtol=0, atol=1e-9)


@pytest.mark.parametrize("cfg", SHAPES3[:6])
def test_3d_shift_grad_rule_is_the_autograd_derivative_away_from_integers(cfg):
    """Self-consistency of the independent restatement: for non-integer shifts the face-difference rule IS
    d(forward)/d(shift) (weights 1-r, r with the floor held constant)."""
    N, T, C, H, W, s, p = cfg
    rng = np.random.default_rng(seed_of(cfg, "autograd"))
    x = _t(rng.uniform(-1, 1, (N, T, C, H, W)))
    shift = _t(special_shifts(rng, 3, C, np.float64, "wide")).requires_grad_(True)
    y = ind.shift3d_forward(x, shift, s, p)
    gy = _t(rng.uniform(-1, 1, tuple(y.shape)))
    (g_auto,) = torch.autograd.grad(y, shift, gy)
    g_rule = ind.shift3d_shift_grad(gy, x, shift.detach(), s, p)
    np.testing.assert_allclose(g_rule.numpy(), g_auto.numpy(), rtol=0, atol=1e-11 * max(1.0, float(g_auto.abs().max())))


def _shifted(x, dt, dh, dw):
    """X(t+dt, h+dh, w+dw) of [N,T,H,W] with zeros outside -- numpy, for the closed forms."""
    N, T, H, W = x.shape
    out = np.zeros_like(x)
    ts, hs, ws = (np.arange(n) for n in (T, H, W))
    tv, hv, wv = ts + dt, hs + dh, ws + dw
    tm, hm, wm = (tv >= 0) & (tv < T), (hv >= 0) & (hv < H), (wv >= 0) & (wv < W)
    out[np.ix_(np.arange(N), ts[tm], hs[hm], ws[wm])] = x[np.ix_(np.arange(N), tv[tm], hv[hm], wv[wm])]
    return out


@pytest.mark.parametrize("s3", [(0.0, 0.0, 0.0), (1.0, -1.0, 2.0), (-2.0, 1.0, -1.0)])
def test_3d_all_three_shifts_integer_closed_form(oracle, s3):
    """SURVEY 7.1-3 with every remainder exactly 0: each derivative is the UN-halved central difference
    X[i+1] - X[i-1] along its own dim, sampled -- because the lowered small index is what the other dims' faces use,
    with weight 1 - r = 1 -- at the positions lowered by one in the other two dims."""
    rng = np.random.default_rng(5)
    N, T, C, H, W = 2, 6, 1, 7, 8
    x = rng.uniform(-1, 1, (N, T, C, H, W))
    gy = rng.uniform(-1, 1, x.shape)
    shift = np.array(s3, np.float64).reshape(3, 1)
    _, _, raw = oracle.rk3d_backward(gy, x, shift, normalize_grad=False, return_raw=True)
    a, b, c = (int(v) for v in s3)
    xc, g = x[:, :, 0], gy[:, :, 0]
    want = [
        (g * (_shifted(xc, a + 1, b - 1, c - 1) - _shifted(xc, a - 1, b - 1, c - 1))).sum(),
        (g * (_shifted(xc, a - 1, b + 1, c - 1) - _shifted(xc, a - 1, b - 1, c - 1))).sum(),
        (g * (_shifted(xc, a - 1, b - 1, c + 1) - _shifted(xc, a - 1, b - 1, c - 1))).sum(),
    ]
    np.testing.assert_allclose(raw[:, 0], want, rtol=0, atol=1e-11)
    np.testing.assert_allclose(ind.shift3d_shift_grad(_t(gy), _t(x), _t(shift)).numpy()[:, 0], want, rtol=0, atol=1e-11)


def test_3d_one_integer_dim_perturbs_the_other_two(oracle):
    """SURVEY 7.1-3 side effect: s_H exactly integer => g_T and g_W are taken one row LOWER (weight 1 - r_H = 1 on
    row h + fl_H - 1) than the plain trilinear derivative would take them."""
    rng = np.random.default_rng(6)
    N, T, C, H, W = 1, 5, 1, 6, 7
    x = rng.uniform(-1, 1, (N, T, C, H, W))
    gy = rng.uniform(-1, 1, x.shape)
    rT, rW = 0.3, 0.6
    shift = np.array([[rT], [1.0], [rW]])
    _, _, raw = oracle.rk3d_backward(gy, x, shift, normalize_grad=False, return_raw=True)
    xc, g = x[:, :, 0], gy[:, :, 0]
    row = 1 - 1                                                    # fl_H - 1: the lowered row
    lerp_w = lambda dt, dh: (1 - rW) * _shifted(xc, dt, dh, 0) + rW * _shifted(xc, dt, dh, 1)   # noqa: E731
    lerp_t = lambda dh, dw: (1 - rT) * _shifted(xc, 0, dh, dw) + rT * _shifted(xc, 1, dh, dw)   # noqa: E731
    want_T = (g * (lerp_w(1, row) - lerp_w(0, row))).sum()
    want_W = (g * (lerp_t(row, 1) - lerp_t(row, 0))).sum()
    want_H = (g * ((1 - rT) * ((1 - rW) * (_shifted(xc, 0, 2, 0) - _shifted(xc, 0, 0, 0)) + rW * (_shifted(xc, 0, 2, 1) - _shifted(xc, 0, 0, 1)))
                   + rT * ((1 - rW) * (_shifted(xc, 1, 2, 0) - _shifted(xc, 1, 0, 0)) + rW * (_shifted(xc, 1, 2, 1) - _shifted(xc, 1, 0, 1))))).sum()
    np.testing.assert_allclose(raw[:, 0], [want_T, want_H, want_W], rtol=0, atol=1e-11)


@pytest.mark.parametrize("dim", [0, 1, 2])
def test_3d_quantize_half_asymmetry_closed_form(oracle, dim):
    """ties round the shift UP in the forward (q = fl + 1 at r = 0.5) and the backward applies the
    same r' < 0.5 rule to the NEGATED shift, so at +-0.5 the backward is not the adjoint of the forward:
      s = +0.5: y[i] = x[i+1]   but  gx[i] = gy[i]     (-0.5 -> fl' = -1, r' = 0.5 -> q' = 0; the adjoint is gy[i-1])
      s = -0.5: y[i] = x[i]     but  gx[i] = gy[i+1]   (+0.5 -> fl' = 0,  r' = 0.5 -> q' = 1; the adjoint is gy[i])"""
    rng = np.random.default_rng(7)
    x = rng.uniform(-1, 1, (1, 5, 1, 6, 7))
    gy = rng.uniform(-1, 1, x.shape)
    for sv, fwd_off, bwd_off in ((0.5, 1, 0), (-0.5, 0, 1)):
        shift = np.zeros((3, 1))
        shift[dim, 0] = sv
        y = oracle.rk3d_forward(x, shift, quantize=True)
        off = [0, 0, 0]
        off[dim] = fwd_off
        np.testing.assert_array_equal(y[:, :, 0], _shifted(x[:, :, 0], *off))
        gx, _ = oracle.rk3d_backward(gy, x, shift, quantize=True)
        off[dim] = bwd_off
        np.testing.assert_array_equal(gx[:, :, 0], _shifted(gy[:, :, 0], *off))
        np.testing.assert_array_equal(ind.shift3d_forward(_t(x), _t(shift), quantize=True).numpy(), y)
        np.testing.assert_array_equal(ind.shift3d_input_grad(_t(gy), x.shape, _t(shift), quantize=True).numpy(), gx)


# --------------------------------------------------------------------------------------------------- 2-D
@pytest.mark.parametrize("kind", KINDS2)
@pytest.mark.parametrize("cfg", SHAPES2)
def test_2d_oracle_equals_independent_restatement(oracle, cfg, kind):
    N, C, H, W, s, p = cfg
    if C > 64:
        C = 24                                                     # per-channel loops: the plane shape is what matters
    rng = np.random.default_rng(seed_of(cfg, kind, "indep"))
    x = rng.uniform(-1, 1, (N, C, H, W))
    shift = special_shifts(rng, 2, C, np.float64, kind)
    y_o = oracle.rk2d_forward(x, shift, s, p)
    np.testing.assert_allclose(y_o, ind.shift2d_forward(_t(x), _t(shift), s, p).numpy(), rtol=0, atol=1e-13)
    keep = rng.uniform(-1, 1, y_o.shape)                           # quantize leaves out-of-range outputs untouched
    yq_o = oracle.rk2d_forward(x, shift, s, p, True, output=keep.copy())
    yq_i = ind.shift2d_forward(_t(x), _t(shift), s, p, True, output=_t(keep)).numpy()
    np.testing.assert_array_equal(yq_o, yq_i)
    gy = rng.uniform(-1, 1, y_o.shape)
    gx_o, _, raw_o = oracle.rk2d_backward(gy, x, shift, s, p, normalize_grad=False, return_raw=True)
    np.testing.assert_allclose(gx_o, ind.shift2d_input_grad(_t(gy), x.shape, _t(shift), s, p).numpy(), rtol=0, atol=1e-13)
    raw_i = ind.shift2d_shift_grad(_t(gy), _t(x), _t(shift), s, p, tol=TOL2D)
    scale = max(1.0, float(np.abs(raw_o).max()))
    np.testing.assert_allclose(raw_o, raw_i.numpy(), rtol=0, atol=1e-11 * scale)
    _, g_o = oracle.rk2d_backward(gy, x, shift, s, p, normalize_grad=True)
    np.testing.assert_allclose(g_o, ind.normalize2d(raw_i).numpy(), rtol=0, atol=1e-9)


@pytest.mark.parametrize("tiny", [1e-8, -1e-8, 8e-8, -8e-8, 1.2e-7])
def test_2d_tiny_shift_closed_form(oracle, tiny):
    """rubiks2d_kernels.cu:189-253: |remainder| < 1e-7 counts as an integer shift -> remainder := 0 and HALF the
    central difference; -1e-8 has floor -1 and remainder 1 - 1e-8 (NOT tiny): the plain one-sided difference with
    weight r ~ 1 on the upper pixel; 1.2e-7 is just outside the tolerance."""
    rng = np.random.default_rng(8)
    x = rng.uniform(-1, 1, (2, 1, 6, 7))
    gy = rng.uniform(-1, 1, x.shape)
    shift = np.array([[tiny], [0.25]])
    _, _, raw = oracle.rk2d_backward(gy, x, shift, normalize_grad=False, return_raw=True)
    xc, g = x[:, None, 0], gy[:, 0]                                # [N,1,H,W] as [N,T=1,H,W] for _shifted
    sh = lambda dh, dw: _shifted(xc, 0, dh, dw)[:, 0]              # noqa: E731
    rw = 0.25
    fl = int(np.floor(tiny))
    r = tiny - fl
    if -TOL2D < r < TOL2D:
        gh = 0.5 * ((1 - rw) * (sh(fl + 1, 0) - sh(fl - 1, 0)) + rw * (sh(fl + 1, 1) - sh(fl - 1, 1)))
        r = 0.0
    else:
        gh = (1 - rw) * (sh(fl + 1, 0) - sh(fl, 0)) + rw * (sh(fl + 1, 1) - sh(fl, 1))
    gw = (1 - r) * (sh(fl, 1) - sh(fl, 0)) + r * (sh(fl + 1, 1) - sh(fl + 1, 0))
    np.testing.assert_allclose(raw[:, 0], [(g * gh).sum(), (g * gw).sum()], rtol=0, atol=1e-12)


# ------------------------------------------------------------------------- INTEGRATION.md's central claim
REF = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "rubiksnet", "shiftlib")), reason="reference tree not present")
def test_reference_shiftlib_binds_to_our_extension_module(monkeypatch):
    """`sys.modules["rubiksnet_cuda"] = rubiksnet_amd.rubiksnet_cuda` lets the reference's own
    rubiksnet/shiftlib run unchanged.  Here (no GPU): the reference package imports over our module, its
    primitives resolve all six binding names with the positional / keyword arguments they use, and calls get as
    far as the device assertion (reference) or our binding's device check -- never an AttributeError/TypeError."""
    from rubiksnet_amd import rubiksnet_cuda as ours

    for name in list(sys.modules):
        if name == "rubiksnet" or name.startswith("rubiksnet."):
            monkeypatch.delitem(sys.modules, name)
    monkeypatch.setitem(sys.modules, "rubiksnet_cuda", ours)
    monkeypatch.syspath_prepend(REF)
    ref3d = importlib.import_module("rubiksnet.shiftlib.rubiks3d.primitive")
    ref2d = importlib.import_module("rubiksnet.shiftlib.rubiks2d.primitive")
    assert ref3d.rubiksnet_cuda is ours and ref2d.rubiksnet_cuda is ours
    for name in ("rubiks_shift_3d_forward_float", "rubiks_shift_3d_forward_double", "rubiks_shift_3d_backward_float",
                 "rubiks_shift_3d_backward_double", "rubiks2d_forward", "rubiks2d_backward"):
        assert callable(getattr(ours, name))

    x5, s3 = torch.zeros(1, 2, 3, 4, 4), torch.zeros(3, 3)
    with pytest.raises(AssertionError):                             # reference: assert x.is_cuda (primitive.py:61)
        ref3d.rubiks_shift_3d_forward(x5, s3, 1, 0)
    # past the reference's own asserts, straight into the binding with the reference's argument lists
    # (rubiks3d/primitive.py:78, :127-138; rubiks2d/primitive.py:56-63, :103-114): our device check must answer
    with pytest.raises(RuntimeError, match="CUDA"):
        ours.rubiks_shift_3d_forward_float(x5, s3, [1, 1, 1], [0, 0, 0], False, torch.zeros_like(x5))
    with pytest.raises(RuntimeError, match="CUDA"):
        ours.rubiks_shift_3d_backward_float(x5, s3, x5, [1, 1, 1], [0, 0, 0], torch.zeros_like(x5),
                                            torch.zeros_like(s3), True, 1.0, False)
    x4, s2 = torch.zeros(2, 3, 4, 4), torch.zeros(2, 3)
    with pytest.raises(RuntimeError, match="CUDA"):
        ours.rubiks2d_forward(input=x4, shift=s2, strides=[1, 1], paddings=[0, 0], quantize=False,
                              output=torch.zeros_like(x4))
    with pytest.raises(RuntimeError, match="CUDA"):
        ours.rubiks2d_backward(upstream_grad=x4, input=x4, shift=s2, strides=[1, 1], paddings=[0, 0],
                               normalize_grad=True, enable_shift_grad=True, quantize=False,
                               input_grad=torch.zeros_like(x4), shift_grad=torch.zeros_like(s2))
    # the reference's 2-D backward has no is_cuda assert of its own: it reaches our binding through ITS call site
    with pytest.raises((RuntimeError, AssertionError)):
        ref2d.rubiks2d_backward(x4, x4, s2, 1, 0)
    # and its layers construct on top of it
    ref_layers = importlib.import_module("rubiksnet.shiftlib")
    layer = ref_layers.RubiksShift3D(6)
    assert tuple(layer.shift.shape) == (3, 6)
