"""ctypes front-end of the CPU oracle (oracle/librubiks_oracle.so).

TEST INFRASTRUCTURE -- NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this module.  It takes and returns numpy
arrays; shapes/layouts are the reference's ([N,T,C,H,W] + shift [3,C] for 3D,
[N,C,H,W] + shift [2,C] for 2D).  Parity status: see oracle/rubiks_oracle.c.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "librubiks_oracle.so")
_lib = None


def build(force=False):
    """Compile the oracle with gcc (a few seconds)."""
    if force or not os.path.exists(_LIB_PATH) or (
        os.path.getmtime(_LIB_PATH)
        < max(os.path.getmtime(os.path.join(_HERE, f)) for f in ("rubiks_oracle.c", "rubiks_oracle_impl.h"))
    ):
        subprocess.check_call(["make", "-C", _HERE, "-s", "all"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.oracle_num_threads.restype = ctypes.c_int
    return _lib


def use_native():
    """Switch this process to a -march=native build of the same source, compiled on THIS host (bench.py's
    cpu_baseline leg; the portable build is what travels with the repo).  Returns True when it is in use."""
    global _lib
    path = os.path.join(_HERE, "librubiks_oracle_native.so")
    try:
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "native"])
        handle = ctypes.CDLL(path)
    except (OSError, subprocess.CalledProcessError):
        return False
    handle.oracle_num_threads.restype = ctypes.c_int
    _lib = handle
    return True


def out_len(size, stride, pad):
    """cuda_src/rubiks.cpp:166 -- NOT the conv formula."""
    return (size + 2 * pad - 1) // stride + 1


def _sfx(dtype):
    dtype = np.dtype(dtype)
    if dtype == np.float32:
        return "f32", ctypes.c_float
    if dtype == np.float64:
        return "f64", ctypes.c_double
    raise ValueError("oracle supports float32/float64 only, got %s" % dtype)


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def _t3(v):
    return [int(v)] * 3 if isinstance(v, int) else [int(e) for e in v]


def _t2(v):
    return [int(v)] * 2 if isinstance(v, int) else [int(e) for e in v]


def set_threads(n):
    lib().oracle_set_num_threads(int(n))


def num_threads():
    return int(lib().oracle_num_threads())


# ------------------------------------------------------------------ 3D
def rk3d_forward(x, shift, stride=1, padding=0, quantize=False):
    sfx, _ = _sfx(x.dtype)
    x = _c(x, x.dtype)
    shift = _c(shift, x.dtype)
    N, T, C, H, W = x.shape
    assert shift.shape == (3, C)
    s, p = _t3(stride), _t3(padding)
    To, Ho, Wo = out_len(T, s[0], p[0]), out_len(H, s[1], p[1]), out_len(W, s[2], p[2])
    y = np.empty((N, To, C, Ho, Wo), dtype=x.dtype)
    getattr(lib(), "oracle_rk3d_forward_" + sfx)(
        _p(x), _p(shift), _p(y), N, T, C, H, W, To, Ho, Wo, *s, *p, int(bool(quantize)))
    return y


def rk3d_backward(gy, x, shift, stride=1, padding=0, normalize_grad=True,
                  normalize_t_factor=1.0, quantize=False, return_raw=False):
    """Returns (gx, gshift[3,C]); with return_raw also the un-normalised gshift."""
    sfx, cty = _sfx(x.dtype)
    x = _c(x, x.dtype)
    shift = _c(shift, x.dtype)
    gy = _c(gy, x.dtype)
    N, T, C, H, W = x.shape
    s, p = _t3(stride), _t3(padding)
    To, Ho, Wo = out_len(T, s[0], p[0]), out_len(H, s[1], p[1]), out_len(W, s[2], p[2])
    assert gy.shape == (N, To, C, Ho, Wo), (gy.shape, (N, To, C, Ho, Wo))
    gx = np.empty_like(x)
    gshift = np.zeros((3, C), dtype=x.dtype)
    scratch = np.empty((3, C, Ho * Wo), dtype=x.dtype)
    L = lib()
    getattr(L, "oracle_rk3d_backward_" + sfx)(
        _p(x), _p(shift), _p(gy), _p(gx), _p(gshift), _p(scratch),
        N, T, C, H, W, To, Ho, Wo, *s, *p, 0, cty(float(normalize_t_factor)), int(bool(quantize)))
    raw = gshift.copy()
    if normalize_grad:
        getattr(L, "oracle_rk3d_normalize_" + sfx)(_p(gshift), C, cty(float(normalize_t_factor)))
    return (gx, gshift, raw) if return_raw else (gx, gshift)


# ------------------------------------------------------------------ 2D
def rk2d_forward(x, shift, stride=1, padding=0, quantize=False, output=None):
    sfx, _ = _sfx(x.dtype)
    x = _c(x, x.dtype)
    shift = _c(shift, x.dtype)
    N, C, H, W = x.shape
    assert shift.shape == (2, C)
    s, p = _t2(stride), _t2(padding)
    Ho, Wo = out_len(H, s[0], p[0]), out_len(W, s[1], p[1])
    # rubiksnet/utils.py:26 -- outputs start as zeros; the quantize branch relies on it
    y = np.zeros((N, C, Ho, Wo), dtype=x.dtype) if output is None else output
    getattr(lib(), "oracle_rk2d_forward_" + sfx)(
        _p(x), _p(shift), _p(y), N, C, H, W, Ho, Wo, *s, *p, int(bool(quantize)))
    return y


def rk2d_backward(gy, x, shift, stride=1, padding=0, normalize_grad=True,
                  enable_shift_grad=True, quantize=False, return_raw=False):
    sfx, _ = _sfx(x.dtype)
    x = _c(x, x.dtype)
    shift = _c(shift, x.dtype)
    gy = _c(gy, x.dtype)
    N, C, H, W = x.shape
    s, p = _t2(stride), _t2(padding)
    Ho, Wo = out_len(H, s[0], p[0]), out_len(W, s[1], p[1])
    assert gy.shape == (N, C, Ho, Wo)
    gx = np.zeros_like(x)
    gshift = np.zeros((2, C), dtype=x.dtype)
    scratch = np.empty((2, C, Ho, Wo), dtype=x.dtype)
    L = lib()
    getattr(L, "oracle_rk2d_backward_" + sfx)(
        _p(gy), _p(x), _p(shift), _p(gx), _p(gshift), _p(scratch),
        N, C, H, W, Ho, Wo, *s, *p, 0, int(bool(enable_shift_grad)), int(bool(quantize)))
    raw = gshift.copy()
    if normalize_grad and enable_shift_grad:
        getattr(L, "oracle_rk2d_normalize_" + sfx)(_p(gshift), C)
    return (gx, gshift, raw) if return_raw else (gx, gshift)
