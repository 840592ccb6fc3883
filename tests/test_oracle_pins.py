"""CPU tests that pin and sanity-check the oracle itself (no GPU, no product code)."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_appendix_a_checksums(oracle):
    """SURVEY.md Appendix A recorded sum(y), sum(scratch), sum(gx) from the reference's own K1/K2/K4
    <float> device code on mt19937(0) inputs at [2,8,16,14,14]; the oracle must reproduce all three."""
    exe = os.path.join(ROOT, "oracle", "appendix_a_check")
    assert os.path.exists(exe)
    out = subprocess.check_output([exe], text=True).strip().splitlines()
    rows = {ln.split()[0]: [float(v) for v in ln.split()[1:]] for ln in out}
    want = [-151.035112, 23.565896, -154.699262]
    assert rows["float"] == pytest.approx(want, abs=5e-7), rows


@pytest.mark.parametrize("name", ["a", "b", "c", "d"])
def test_attention_oracle_matches_reference_fixtures(name, golden_dir):
    """tests/golden/attention_*.npz were produced by importing the reference's AttentionShift."""
    from oracle import attention_oracle as ao

    g = np.load(os.path.join(golden_dir, "attention_%s.npz" % name))
    x, gy, w, S = g["x"], g["gy"], g["weight"], int(g["n_segment"])
    tol = 1e-5 if x.dtype == np.float32 else 1e-12
    y = ao.forward(x, w, S)
    np.testing.assert_allclose(y, g["y"], rtol=tol, atol=tol)
    gx, gw = ao.backward(gy, x, w, S)
    np.testing.assert_allclose(gx, g["gx"], rtol=tol, atol=tol)
    np.testing.assert_allclose(gw, g["gweight"], rtol=20 * tol, atol=20 * tol)


def _rand(rng, shape, dtype):
    return rng.uniform(-1, 1, size=shape).astype(dtype)


CFGS = [
    # N, T, C, H, W, stride, padding
    (2, 4, 5, 6, 7, (1, 1, 1), (0, 0, 0)),
    (1, 5, 3, 9, 8, (1, 2, 2), (0, 0, 0)),
    (2, 3, 4, 7, 7, (1, 2, 2), (0, 1, 1)),
    (1, 6, 2, 5, 6, (2, 1, 3), (1, 2, 0)),
]


@pytest.mark.parametrize("cfg", CFGS)
def test_oracle_3d_forward_backward_are_adjoint(oracle, cfg):
    """d(x) (K3/K4) must be the exact adjoint of the forward (K1) for non-integer shifts:
    <forward(x), gy> == <x, backward_input(gy)>."""
    N, T, C, H, W, s, p = cfg
    rng = np.random.default_rng(1)
    x = _rand(rng, (N, T, C, H, W), np.float64)
    shift = rng.uniform(-2.3, 2.3, size=(3, C))
    y = oracle.rk3d_forward(x, shift, s, p)
    gy = _rand(rng, y.shape, np.float64)
    gx, _ = oracle.rk3d_backward(gy, x, shift, s, p, normalize_grad=False)
    assert np.dot(y.ravel(), gy.ravel()) == pytest.approx(np.dot(x.ravel(), gx.ravel()), rel=1e-12)


@pytest.mark.parametrize("cfg", CFGS)
def test_oracle_3d_shift_grad_matches_finite_difference(oracle, cfg):
    """Away from integer shifts d(shift) is the derivative of <forward(x; shift), gy>."""
    N, T, C, H, W, s, p = cfg
    rng = np.random.default_rng(2)
    x = _rand(rng, (N, T, C, H, W), np.float64)
    shift = rng.uniform(-1.4, 1.4, size=(3, C))
    shift = np.where(np.abs(shift - np.round(shift)) < 0.05, shift + 0.1, shift)
    gy = _rand(rng, oracle.rk3d_forward(x, shift, s, p).shape, np.float64)
    _, _, raw = oracle.rk3d_backward(gy, x, shift, s, p, normalize_grad=False, return_raw=True)
    eps = 1e-6
    for d in range(3):
        for c in range(C):
            sp, sm = shift.copy(), shift.copy()
            sp[d, c] += eps
            sm[d, c] -= eps
            fd = (np.vdot(oracle.rk3d_forward(x, sp, s, p), gy) - np.vdot(oracle.rk3d_forward(x, sm, s, p), gy)) / (2 * eps)
            assert raw[d, c] == pytest.approx(fd, rel=1e-6, abs=1e-7)


def test_oracle_3d_quantize_is_a_pure_gather(oracle):
    rng = np.random.default_rng(3)
    x = _rand(rng, (2, 5, 4, 6, 6), np.float32)
    shift = rng.uniform(-1.9, 1.9, size=(3, 4)).astype(np.float32)
    y = oracle.rk3d_forward(x, shift, 1, 0, quantize=True)
    assert np.isin(y, np.concatenate([x.ravel(), [0.0]])).all()
    # every shift rounds half-up: s = 0.5 -> +1, s = -0.5 -> 0
    sh = np.zeros((3, 4), np.float32)
    sh[0] = [0.5, -0.5, 0.49, -0.51]
    y = oracle.rk3d_forward(x, sh, 1, 0, quantize=True)
    np.testing.assert_array_equal(y[:, :-1, 0], x[:, 1:, 0])
    np.testing.assert_array_equal(y[:, :, 1], x[:, :, 1])
    np.testing.assert_array_equal(y[:, :, 2], x[:, :, 2])
    np.testing.assert_array_equal(y[:, 1:, 3], x[:, :-1, 3])


def test_oracle_3d_integer_shift_is_central_difference(oracle):
    """r == 0 exactly: d/ds = X[i+1] - X[i-1], not halved, and the quantize
    backward at s = .5 uses the tap matching forward floor (the documented asymmetry)."""
    rng = np.random.default_rng(4)
    x = _rand(rng, (1, 6, 1, 5, 5), np.float64)
    gy = _rand(rng, x.shape, np.float64)
    shift = np.array([[0.0], [0.3], [0.6]])      # T exactly integer
    _, _, raw = oracle.rk3d_backward(gy, x, shift, 1, 0, normalize_grad=False, return_raw=True)
    # plane-interpolated field B(t): bilinear with the (H,W) remainders; gT = sum gy * (B(t+1) - B(t-1))
    def B(t):
        if t < 0 or t >= 6:
            return np.zeros((5, 5))
        pl = np.pad(x[0, t, 0], ((0, 1), (0, 1)))
        return 0.7 * (0.4 * pl[:-1, :-1] + 0.6 * pl[:-1, 1:]) + 0.3 * (0.4 * pl[1:, :-1] + 0.6 * pl[1:, 1:])
    want = sum(np.vdot(gy[0, t, 0], B(t + 1) - B(t - 1)) for t in range(6))
    assert raw[0, 0] == pytest.approx(want, rel=1e-12)


@pytest.mark.parametrize("stride,padding", [(1, 0), (2, 0), (2, 1), ((1, 3), (2, 0))])
def test_oracle_2d_adjoint_and_fd(oracle, stride, padding):
    rng = np.random.default_rng(5)
    x = _rand(rng, (2, 3, 7, 8), np.float64)
    shift = rng.uniform(-1.6, 1.6, size=(2, 3))
    shift = np.where(np.abs(shift - np.round(shift)) < 0.05, shift + 0.1, shift)
    y = oracle.rk2d_forward(x, shift, stride, padding)
    gy = _rand(rng, y.shape, np.float64)
    gx, _, raw = oracle.rk2d_backward(gy, x, shift, stride, padding, normalize_grad=False, return_raw=True)
    assert np.vdot(y, gy) == pytest.approx(np.vdot(x, gx), rel=1e-12)
    eps = 1e-6
    for d in range(2):
        for c in range(3):
            sp, sm = shift.copy(), shift.copy()
            sp[d, c] += eps
            sm[d, c] -= eps
            fd = (np.vdot(oracle.rk2d_forward(x, sp, stride, padding), gy)
                  - np.vdot(oracle.rk2d_forward(x, sm, stride, padding), gy)) / (2 * eps)
            assert raw[d, c] == pytest.approx(fd, rel=1e-6, abs=1e-7)


def test_oracle_2d_integer_shift_halved_central_difference(oracle):
    """2D halves the central difference (rubiks2d_kernels.cu:238-252), unlike 3D."""
    rng = np.random.default_rng(6)
    x = _rand(rng, (1, 1, 6, 6), np.float64)
    gy = _rand(rng, x.shape, np.float64)
    shift = np.array([[1.0], [0.25]])
    _, _, raw = oracle.rk2d_backward(gy, x, shift, 1, 0, normalize_grad=False, return_raw=True)
    xp = np.pad(x[0, 0], ((3, 3), (3, 3)))
    def at(h, w):
        return xp[h + 3, w + 3]
    want = 0.0
    for h in range(6):
        for w in range(6):
            h0, w0 = h + 1, w + 0
            want += gy[0, 0, h, w] * 0.5 * (0.75 * (at(h0 + 1, w0) - at(h0 - 1, w0)) + 0.25 * (at(h0 + 1, w0 + 1) - at(h0 - 1, w0 + 1)))
    assert raw[0, 0] == pytest.approx(want, rel=1e-12)


def test_oracle_2d_quantize_leaves_out_of_range_untouched(oracle):
    x = np.arange(16, dtype=np.float32).reshape(1, 1, 4, 4) + 1
    shift = np.array([[2.0], [0.0]], np.float32)
    out = np.full((1, 1, 4, 4), -7.0, np.float32)
    y = oracle.rk2d_forward(x, shift, 1, 0, quantize=True, output=out)
    np.testing.assert_array_equal(y[0, 0, :2], x[0, 0, 2:])
    np.testing.assert_array_equal(y[0, 0, 2:], -7.0)       # rubiks2d_kernels.cu:116-121


def test_oracle_normalize(oracle):
    import ctypes
    L = oracle.lib()
    g = np.array([[3.0, 0.0, 2.0], [4.0, 0.0, 0.0], [0.0, 0.0, 1.0]], np.float64)   # [3, C=3]
    a = g.copy()
    L.oracle_rk3d_normalize_f64(a.ctypes.data_as(ctypes.c_void_p), 3, ctypes.c_double(1.0))
    np.testing.assert_allclose(a[:, 0], [0.6, 0.8, 0.0])
    np.testing.assert_allclose(a[:, 1], [0.0, 0.0, 0.0])            # zero norm: untouched
    b = g.copy()
    L.oracle_rk3d_normalize_f64(b.ctypes.data_as(ctypes.c_void_p), 3, ctypes.c_double(-1.0))
    np.testing.assert_allclose(b[:, 0], [1.0, 0.0, 0.0])            # t_factor < 0: T only
    np.testing.assert_allclose(b[:, 2], [1.0, 0.0, 0.0])
    np.testing.assert_allclose(b[:, 1], g[:, 1])                    # |gT| == 0: untouched
