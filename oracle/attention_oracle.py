"""numpy oracle for AttentionShift (rubiksnet/attention_shift.py:6-39).

TEST INFRASTRUCTURE -- NOT PRODUCT CODE (same rules as oracle/oracle.py).

Parity status: PINNED -- tests/golden/attention_*.npz hold (x, weight, gy) ->
(y, gx, gweight) produced by importing the reference's pure-PyTorch module on CPU
(tests/golden/gen_attention_golden.py); tests/test_oracle_pins.py checks this
restatement against them.
"""
import numpy as np


def soft_weights(weight, temperature=2.0):
    """attention_shift.py:29-30: softmax((w / (std(w, dim=1, unbiased) + 1e-6)) / T, dim=1)."""
    w = np.asarray(weight)
    std = w.std(axis=1, ddof=1, keepdims=True)
    z = (w / (std + 1e-6)) / temperature
    z = z - z.max(axis=1, keepdims=True)
    e = np.exp(z)
    return e / e.sum(axis=1, keepdims=True)


def taps_forward(x, soft, n_segment):
    """attention_shift.py:32-39: y[n,t] = s0*x[n,t-1] + s1*x[n,t] + s2*x[n,t+1], zero padded in t.

    x: [N*T, C, H, W]; soft: [C, 3] (already soft-maxed)."""
    nt, c, h, w = x.shape
    n = nt // n_segment
    xv = x.reshape(n, n_segment, c, h, w)
    s = soft.astype(x.dtype).reshape(1, 1, c, 3, 1, 1)
    y = s[:, :, :, 1] * xv
    y[:, 1:] += s[:, :, :, 0] * xv[:, :-1]
    y[:, :-1] += s[:, :, :, 2] * xv[:, 1:]
    return y.reshape(nt, c, h, w)


def taps_backward(gy, x, soft, n_segment, compute=np.float64):
    """Adjoint of taps_forward: returns (gx, gsoft[C,3]).  `compute` is the arithmetic type of gx
    (float32 reproduces the kernel's per-element expression (s1*g[t] + s0*g[t+1]) + s2*g[t-1] exactly;
    fp addition is commutative, so this is the kernel's (s0*g[t+1] + s1*g[t]) + s2*g[t-1]);
    the tap sums are always float64."""
    nt, c, h, w = x.shape
    n = nt // n_segment
    xv = x.reshape(n, n_segment, c, h, w).astype(np.float64)
    gv = gy.reshape(n, n_segment, c, h, w).astype(np.float64)
    gc = gy.reshape(n, n_segment, c, h, w).astype(compute)
    s = soft.astype(compute).reshape(1, 1, c, 3, 1, 1)
    gx = s[:, :, :, 1] * gc
    gx[:, :-1] += s[:, :, :, 0] * gc[:, 1:]
    gx[:, 1:] += s[:, :, :, 2] * gc[:, :-1]
    gs = np.zeros((c, 3), dtype=np.float64)
    gs[:, 1] = (gv * xv).sum(axis=(0, 1, 3, 4))
    gs[:, 0] = (gv[:, 1:] * xv[:, :-1]).sum(axis=(0, 1, 3, 4))
    gs[:, 2] = (gv[:, :-1] * xv[:, 1:]).sum(axis=(0, 1, 3, 4))
    return gx.reshape(nt, c, h, w).astype(x.dtype), gs


def weight_grad_from_soft_grad(weight, gsoft, temperature=2.0):
    """Chain rule of soft_weights: d/dweight given d/dsoft (float64)."""
    w = np.asarray(weight, dtype=np.float64)
    k = w.shape[1]
    mu = w.mean(axis=1, keepdims=True)
    std = w.std(axis=1, ddof=1, keepdims=True)
    den = std + 1e-6
    soft = soft_weights(w, temperature)
    dz = soft * (gsoft - (soft * gsoft).sum(axis=1, keepdims=True))
    dstd = (w - mu) / ((k - 1) * std)
    return dz / (den * temperature) - (dz * w).sum(axis=1, keepdims=True) / (den * den * temperature) * dstd


def forward(x, weight, n_segment, temperature=2.0):
    return taps_forward(x, soft_weights(weight, temperature), n_segment)


def backward(gy, x, weight, n_segment, temperature=2.0):
    """Returns (gx, gweight) for the full module."""
    soft = soft_weights(np.asarray(weight, dtype=np.float64), temperature)
    gx, gs = taps_backward(gy, x, soft, n_segment)
    return gx, weight_grad_from_soft_grad(weight, gs, temperature)
