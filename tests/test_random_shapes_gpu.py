"""Randomised shape sweep of both shift operators against the oracle (test infrastructure: oracle/).

The fixed shape lists of test_parity_3d.py / test_parity_2d.py name the networks' layers; here seeded random
(N, T, C, H, W, stride, shift kind) draws walk the dispatch boundaries of the kernel families instead -- row-band
counts and ragged last rounds of the LDS-DMA kernels, W % 8 for the raw 16-bit planes, 14x14 tiles with odd frame /
channel-group counts, stride-2 band splits, integer temporal shifts -- with the same bars: y and d(x) bit-exact,
d(shift) within 1e-5 * scale of the fp64 oracle (16-bit storage: the fp32 oracle on the widened inputs, rounded once).
"""
import os

import numpy as np
import pytest
import torch

from _util import rand, special_shifts, to_dev, to_np

pytestmark = pytest.mark.gpu

KINDS = ["generic", "wide", "integer", "half"]


def _draw3(rng):
    fam = rng.integers(0, 7)
    if fam == 6:      # small strided planes on the slab kernels' stride-(1,2,2) backward: even H, W, C % 4 == 0 mostly
        H, W = 2 * int(rng.integers(2, 16)), 2 * int(rng.integers(2, 16))
        N, T = int(rng.integers(1, 4)), int(rng.integers(1, 9))
        C = 4 * int(rng.integers(1, 10)) if rng.random() < 0.7 else int(rng.integers(1, 20))
        return N, T, C, H, W, (1, 2, 2), (0, 0, 0)
    if fam == 5:      # small planes on the slab kernels (rk3d_slab.hpp): C * H * W % 4 == 0, W <= 15, T <= 8
        H, W = int(rng.integers(2, 16)), int(rng.integers(2, 16))
        s = (1, 1, 1)
        N, T = int(rng.integers(1, 4)), int(rng.integers(1, 9))
        C = 4 * int(rng.integers(1, 16)) if rng.random() < 0.7 else int(rng.integers(1, 40))
        return N, T, C, H, W, s, (0, 0, 0)
    if fam == 0:      # streaming stride 1: W % 4 == 0, assorted heights (band counts, ragged rounds)
        H, W = int(rng.choice([8, 12, 20, 28, 36, 56, 60, 72])), int(rng.choice([8, 16, 28, 40, 56, 64, 96]))
        s = (1, 1, 1)
    elif fam == 1:    # 14x14 tiles
        H = W = 14
        s = (1, 1, 1)
    elif fam == 2:    # stride (1,2,2) streaming: W % 8 == 0 so that Wo % 4 == 0
        H, W = int(rng.choice([8, 16, 28, 56, 64])), int(rng.choice([8, 16, 24, 56, 112]))
        s = (1, 2, 2)
    elif fam == 3:    # odd planes: column / generic kernels
        H, W = int(rng.integers(3, 19)), int(rng.integers(3, 19))
        s = tuple(int(v) for v in rng.choice([1, 2], 3))
    else:             # anything, with padding
        H, W = int(rng.integers(4, 30)), int(rng.integers(4, 30))
        s = (int(rng.choice([1, 2])), int(rng.choice([1, 2, 3])), int(rng.choice([1, 2])))
    N, T, C = int(rng.integers(1, 4)), int(rng.integers(1, 7)), int(rng.integers(1, 11))
    p = (0, 0, 0) if fam < 4 else tuple(int(v) for v in rng.integers(0, 3, 3))
    return N, T, C, H, W, s, p


@pytest.mark.parametrize("seed", range(int(os.environ.get("RK_SWEEP_3D", "210"))))   # RK_SWEEP_3D=2000: soak run
def test_random_3d(oracle, seed):
    from rubiksnet_amd.shiftlib.rubiks3d.primitive import rubiks_shift_3d_backward, rubiks_shift_3d_forward

    rng = np.random.default_rng(1000 + seed)
    N, T, C, H, W, s, p = _draw3(rng)
    kind = KINDS[seed % len(KINDS)]
    quantize = seed % 7 == 3
    x = rand(rng, (N, T, C, H, W), np.float32)
    shift = special_shifts(rng, 3, C, np.float32, kind)
    if seed % 5 == 0:
        shift[0] = rng.choice([-2.0, -1.0, 0.0, 1.0, 2.0], C)          # "tsm"-style integer temporal shifts
    y_ref = oracle.rk3d_forward(x, shift, s, p, quantize)
    y = to_np(rubiks_shift_3d_forward(to_dev(x), to_dev(shift), s, p, quantize=quantize))
    np.testing.assert_array_equal(y, y_ref, err_msg=f"fwd {(N, T, C, H, W, s, p, kind, quantize)}")
    gy = rand(rng, y_ref.shape, np.float32)
    gx_ref, _ = oracle.rk3d_backward(gy, x, shift, s, p, quantize=quantize)
    _, _, raw_ref = oracle.rk3d_backward(gy.astype(np.float64), x.astype(np.float64), shift.astype(np.float64), s, p,
                                         normalize_grad=False, quantize=quantize, return_raw=True)
    gx, gs = rubiks_shift_3d_backward(to_dev(gy), to_dev(x), to_dev(shift), s, p, False, quantize=quantize)
    np.testing.assert_array_equal(to_np(gx), gx_ref, err_msg=f"gx {(N, T, C, H, W, s, p, kind, quantize)}")
    scale = max(1.0, float(np.abs(raw_ref).max()))
    np.testing.assert_allclose(to_np(gs), raw_ref, rtol=0, atol=1e-5 * scale,
                               err_msg=f"gshift {(N, T, C, H, W, s, p, kind, quantize)}")


def _draw2(rng):
    fam = rng.integers(0, 4)
    if fam == 0:      # streaming, W % 4 == 0 (W % 8 == 0 half of the time: raw 16-bit planes)
        H, W = int(rng.choice([4, 8, 12, 28, 40, 56])), int(rng.choice([4, 8, 12, 16, 28, 56, 72]))
        s, p = 1, 0
    elif fam == 1:    # 14x14 tiles: even channel counts, ragged frame groups
        H = W = 14
        s, p = 1, 0
    elif fam == 2:    # strided
        H, W = int(rng.integers(4, 30)), int(rng.integers(4, 30))
        s, p = 2, 0
    else:
        H, W = int(rng.integers(3, 20)), int(rng.integers(3, 20))
        s, p = (int(rng.choice([1, 2, 3])), int(rng.choice([1, 2]))), (int(rng.integers(0, 3)), int(rng.integers(0, 3)))
    F = int(rng.choice([1, 3, 16, 17, 40]))
    C = int(rng.choice([2, 4, 6, 10])) if fam == 1 and rng.random() < 0.8 else int(rng.integers(1, 9))
    return F, C, H, W, s, p


@pytest.mark.parametrize("tdtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("seed", range(int(os.environ.get("RK_SWEEP_2D", "60"))))   # RK_SWEEP_2D=600: soak run
def test_random_2d(oracle, seed, tdtype):
    from rubiksnet_amd.shiftlib.rubiks2d.primitive import rubiks2d_backward, rubiks2d_forward

    rng = np.random.default_rng(5000 + seed)
    F, C, H, W, s, p = _draw2(rng)
    kind = (KINDS + ["tiny"])[seed % 5]
    x = torch.from_numpy(rand(rng, (F, C, H, W), np.float32)).to(tdtype)
    shift = torch.from_numpy(special_shifts(rng, 2, C, np.float32, kind)).to(tdtype)
    xf, sf = x.float().numpy(), shift.float().numpy()
    y_ref = oracle.rk2d_forward(xf, sf, s, p)
    gy = torch.from_numpy(rand(rng, y_ref.shape, np.float32)).to(tdtype)
    gf = gy.float().numpy()
    tag = f"{(F, C, H, W, s, p, kind, tdtype)}"
    y = rubiks2d_forward(x.cuda(), shift.cuda(), s, p)
    assert torch.equal(y.cpu(), torch.from_numpy(y_ref).to(tdtype)), "fwd " + tag
    gx, gs = rubiks2d_backward(gy.cuda(), x.cuda(), shift.cuda(), s, p, normalize_grad=False)
    gx_ref, _ = oracle.rk2d_backward(gf, xf, sf, s, p)
    assert torch.equal(gx.cpu(), torch.from_numpy(gx_ref).to(tdtype)), "gx " + tag
    _, gs_ref = oracle.rk2d_backward(gf.astype(np.float64), xf.astype(np.float64), sf.astype(np.float64), s, p,
                                     normalize_grad=False)
    eps = {torch.float32: 1e-5, torch.float16: 2e-3, torch.bfloat16: 1.6e-2}[tdtype]
    scale = max(1.0, float(np.abs(gs_ref).max()))
    np.testing.assert_allclose(gs.float().cpu().numpy(), gs_ref, rtol=0, atol=eps * scale, err_msg="gshift " + tag)
