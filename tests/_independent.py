"""A second, independent restatement of the shift operators -- TEST INFRASTRUCTURE.

Written from the behavioural spec in SURVEY.md section 7.1 (items 1-4), NOT from oracle/rubiks_oracle_impl.h:
vectorised PyTorch in fp64 -- zero-extended index_select ("pad + slice"), nested lerps, and autograd for the
adjoints -- where the C oracle is loop nests over (n, t, c, h, w).  tests/test_oracle_pins.py asserts that the two
agree on every shape x shift-kind of the GPU parity suites, so the "HIP == oracle" parity claims do not rest on a
single reading of the reference.  The oracle stays "parity unpinned by reference execution" (the reference is
a CUDA extension); this narrows what an error in it could look like to "both readings wrong in the same way".
"""
import torch

F64 = torch.float64


def out_len(size, stride, pad):
    return (size + 2 * pad - 1) // stride + 1          # (in + 2 pad - 1) // stride + 1


def _t(v, n):
    return [int(v)] * n if isinstance(v, int) else [int(e) for e in v]


def _floor_fp32(s):
    """floor() taken in fp32 whatever the tensor dtype is (floorf even for the double kernels)."""
    return torch.floor(s.detach().to(torch.float32)).to(torch.int64)


def _take(x, dim, idx):
    """x indexed by `idx` along `dim`, zero outside [0, size) -- X(.) of the spec."""
    size = x.shape[dim]
    ok = (idx >= 0) & (idx < size)
    shape = [1] * x.dim()
    shape[dim] = -1
    return x.index_select(dim, idx.clamp(0, size - 1)) * ok.view(shape).to(x.dtype)


def _base(n_out, stride, pad):
    return torch.arange(n_out, dtype=torch.int64) * stride - pad


# ----------------------------------------------------------------------------------------- 3-D operator
def shift3d_forward(x, shift, stride=1, padding=0, quantize=False):
    """SURVEY 7.1-1.  x [N,T,C,H,W], shift [3,C] (rows T,H,W).  Differentiable in x and (non-quantized) shift."""
    sT, sH, sW = _t(stride, 3)
    pT, pH, pW = _t(padding, 3)
    N, T, C, H, W = x.shape
    bT, bH, bW = _base(out_len(T, sT, pT), sT, pT), _base(out_len(H, sH, pH), sH, pH), _base(out_len(W, sW, pW), sW, pW)
    planes = []
    for c in range(C):
        xc = x[:, :, c]                                   # [N, T, H, W]
        s = shift[:, c]
        fl = _floor_fp32(s)
        r = s - fl.to(s.dtype)
        if quantize:                                      # q = fl if r < 0.5 else fl + 1
            q = [int(fl[d]) + (0 if float(r[d]) < 0.5 else 1) for d in range(3)]
            planes.append(_take(_take(_take(xc, 1, bT + q[0]), 2, bH + q[1]), 3, bW + q[2]))
            continue
        f = [int(v) for v in fl]

        def tap(i, j, k):
            return _take(_take(_take(xc, 1, bT + f[0] + i), 2, bH + f[1] + j), 3, bW + f[2] + k)

        def lerp_w(i, j):
            return tap(i, j, 0) * (1 - r[2]) + tap(i, j, 1) * r[2]

        def lerp_h(i):
            return (1 - r[1]) * lerp_w(i, 0) + r[1] * lerp_w(i, 1)

        planes.append((1 - r[0]) * lerp_h(0) + r[0] * lerp_h(1))
    return torch.stack(planes, dim=2)


def shift3d_input_grad(gy, x_shape, shift, stride=1, padding=0, quantize=False):
    """SURVEY 7.1-2.  Non-quantized: the exact adjoint of the forward (through autograd on shift3d_forward).
    Quantized: nearest tap of gy at the NEGATED shift with the same r' < 0.5 rule, taps counted only where
    (p + q') is divisible by the stride (C remainder) and the quotient is in range."""
    if not quantize:
        x = torch.zeros(x_shape, dtype=gy.dtype, requires_grad=True)
        y = shift3d_forward(x, shift.detach(), stride, padding)
        return torch.autograd.grad(y, x, gy)[0]
    st, pd = _t(stride, 3), _t(padding, 3)
    N, T, C, H, W = x_shape
    out = []
    for c in range(C):
        gc = gy[:, :, c]
        sp = -shift[:, c]
        fl = _floor_fp32(sp)
        r = sp - fl.to(sp.dtype)
        g = gc
        for d, size in enumerate((T, H, W)):
            q = int(fl[d]) + (0 if float(r[d]) < 0.5 else 1)
            num = torch.arange(size, dtype=torch.int64) + pd[d] + q
            div = torch.div(num, st[d], rounding_mode="trunc")
            idx = torch.where(num - div * st[d] == 0, div, torch.full_like(div, -1))   # C '%': non-multiples skipped
            g = _take(g, d + 1, idx)
        out.append(g)
    return torch.stack(out, dim=2)


def shift3d_shift_grad(gy, x, shift, stride=1, padding=0):
    """SURVEY 7.1-3, raw (un-normalised) [3,C]: per output element g_d += gy (L_d - S_d), L_d / S_d the bilinear
    interpolation (weights of the OTHER two dims) of the large / small face.  Integer rule: a dim whose remainder is
    exactly 0 takes its small index one lower (fl - 1), and that lowered index is what every face uses."""
    sT, sH, sW = _t(stride, 3)
    pT, pH, pW = _t(padding, 3)
    N, T, C, H, W = x.shape
    bases = (_base(out_len(T, sT, pT), sT, pT), _base(out_len(H, sH, pH), sH, pH), _base(out_len(W, sW, pW), sW, pW))
    g = torch.zeros(3, C, dtype=x.dtype)
    for c in range(C):
        xc = x[:, :, c]
        s = shift[:, c]
        fl = _floor_fp32(s)
        r = s - fl.to(s.dtype)
        lo = [int(fl[d]) - (1 if float(r[d]) == 0.0 else 0) for d in range(3)]
        hi = [int(fl[d]) + 1 for d in range(3)]
        w_lo = [1 - r[d] for d in range(3)]
        w_hi = [r[d] for d in range(3)]

        def corner(sel):                                   # sel[d] in {0: small index, 1: large index}
            v = xc
            for d in range(3):
                v = _take(v, d + 1, bases[d] + (hi[d] if sel[d] else lo[d]))
            return v

        for d in range(3):
            o1, o2 = [e for e in range(3) if e != d]
            diff = 0
            for a in (0, 1):
                for b in (0, 1):
                    sel_hi, sel_lo = [0, 0, 0], [0, 0, 0]
                    sel_hi[d], sel_lo[d] = 1, 0
                    sel_hi[o1] = sel_lo[o1] = a
                    sel_hi[o2] = sel_lo[o2] = b
                    wa = w_hi[o1] if a else w_lo[o1]
                    wb = w_hi[o2] if b else w_lo[o2]
                    diff = diff + wa * wb * (corner(sel_hi) - corner(sel_lo))
            g[d, c] = (gy[:, :, c] * diff).sum()
    return g


def normalize3d(g, t_factor=1.0):
    """SURVEY 7.1-3, last paragraph (K5)."""
    g = g.clone()
    if t_factor < 0:
        gT = g[0]
        mag = gT.abs()
        out = torch.zeros_like(g)
        out[0] = torch.where(mag > 0, gT / mag, gT)
        return out
    g[0] = g[0] * t_factor
    mag = g.norm(dim=0)
    return torch.where(mag > 0, g / mag, g)


# ----------------------------------------------------------------------------------------- 2-D operator
def _round_half_away(v):
    return torch.where(v < 0, torch.trunc(v - 0.5), torch.trunc(v + 0.5)).to(torch.int64)


def shift2d_forward(x, shift, stride=1, padding=0, quantize=False, output=None):
    """SURVEY 7.1-4.  x [N,C,H,W], shift [2,C].  quantize: position = round-half-away(base + shift) of the ABSOLUTE
    position (computed in the tensor dtype); out-of-range outputs keep whatever `output` held (zeros by default)."""
    sH, sW = _t(stride, 2)
    pH, pW = _t(padding, 2)
    N, C, H, W = x.shape
    bH, bW = _base(out_len(H, sH, pH), sH, pH), _base(out_len(W, sW, pW), sW, pW)
    planes = []
    for c in range(C):
        xc = x[:, c]
        s = shift[:, c]
        if quantize:
            ih = _round_half_away(bH.to(x.dtype) + s[0])
            iw = _round_half_away(bW.to(x.dtype) + s[1])
            ok = ((ih >= 0) & (ih < H)).view(-1, 1) & ((iw >= 0) & (iw < W)).view(1, -1)
            got = xc.index_select(1, ih.clamp(0, H - 1)).index_select(2, iw.clamp(0, W - 1))
            keep = torch.zeros_like(got) if output is None else output[:, c]
            planes.append(torch.where(ok, got, keep))
            continue
        fl = torch.floor(s.detach()).to(torch.int64)
        r = s - fl.to(s.dtype)
        f = [int(v) for v in fl]

        def tap(j, k):
            return _take(_take(xc, 1, bH + f[0] + j), 2, bW + f[1] + k)

        planes.append(tap(0, 0) * (1 - r[0]) * (1 - r[1]) + tap(0, 1) * (1 - r[0]) * r[1] + tap(1, 0) * r[0] * (1 - r[1])
                      + tap(1, 1) * r[0] * r[1])
    return torch.stack(planes, dim=1)


def shift2d_input_grad(gy, x_shape, shift, stride=1, padding=0):
    x = torch.zeros(x_shape, dtype=gy.dtype, requires_grad=True)
    return torch.autograd.grad(shift2d_forward(x, shift.detach(), stride, padding), x, gy)[0]


def shift2d_shift_grad(gy, x, shift, stride=1, padding=0, tol=1e-7):
    """one-sided differences as in 3-D, but a dim whose remainder is within `tol` of 0 has that
    remainder set to 0 and takes HALF the central difference X[i+1] - X[i-1] (weights of the other dim unchanged)."""
    sH, sW = _t(stride, 2)
    pH, pW = _t(padding, 2)
    N, C, H, W = x.shape
    bases = (_base(out_len(H, sH, pH), sH, pH), _base(out_len(W, sW, pW), sW, pW))
    g = torch.zeros(2, C, dtype=x.dtype)
    for c in range(C):
        xc = x[:, c]
        s = shift[:, c]
        fl = torch.floor(s).to(torch.int64)
        r = s - fl.to(s.dtype)
        is_int = [bool(-tol < float(r[d]) < tol) for d in range(2)]
        r = [torch.zeros((), dtype=x.dtype) if is_int[d] else r[d] for d in range(2)]
        f = [int(v) for v in fl]

        def px(dh, dw):
            return _take(_take(xc, 1, bases[0] + f[0] + dh), 2, bases[1] + f[1] + dw)

        if is_int[0]:
            gh = 0.5 * ((1 - r[1]) * (px(1, 0) - px(-1, 0)) + r[1] * (px(1, 1) - px(-1, 1)))
        else:
            gh = (1 - r[1]) * (px(1, 0) - px(0, 0)) + r[1] * (px(1, 1) - px(0, 1))
        if is_int[1]:
            gw = 0.5 * ((1 - r[0]) * (px(0, 1) - px(0, -1)) + r[0] * (px(1, 1) - px(1, -1)))
        else:
            gw = (1 - r[0]) * (px(0, 1) - px(0, 0)) + r[0] * (px(1, 1) - px(1, 0))
        g[0, c] = (gy[:, c] * gh).sum()
        g[1, c] = (gy[:, c] * gw).sum()
    return g


def normalize2d(g):
    mag = g.norm(dim=0)
    return torch.where(mag > 0, g / mag, g)
