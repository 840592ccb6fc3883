"""The in-launch d(shift) row-sum (rk_dma.hpp "Row-sum of the d(shift) partials INSIDE the backward launch") under the two
conditions nothing else exercises: a device that is busy with somebody else's kernels while the backward runs, and the
finalizers' give-up branch -- a finalizer that never sees its partials must end the launch and poison its outputs with
NaN, never hang and never return a plausible number.  Replaces K2's atomics + the addmv_ row-sum + K5
(rubiks3d_kernels.cu:218-452, rubiks.cpp:344-345, rubiks3d_kernels.cu:932-960)."""
import time

import numpy as np
import pytest
import torch

from _util import DEV

pytestmark = pytest.mark.gpu


def _backward(x, shift, gy):
    from rubiksnet_amd import rubiksnet_cuda as rc
    gx = torch.empty_like(x)
    gs = torch.empty_like(shift)
    rc.rubiks_shift_3d_backward_float(x, shift, gy, [1, 1, 1], [0, 0, 0], gx, gs, True, 1.0, False)
    return gx, gs


@pytest.mark.parametrize("shape", [(8, 8, 64, 56, 56), (16, 8, 288, 14, 14), (16, 8, 576, 7, 7)])
def test_backward_is_bit_stable_while_another_stream_keeps_the_device_busy(shape):
    """Producers hand their partials to finalizer waves of the same launch; a co-running kernel stream (here: GEMMs and
    elementwise sweeps on a second stream, enough to hold every CU for the whole duration) changes when producers and
    finalizers get their slots, never what the finalizers sum -- d(x) and d(shift) bit-equal to the quiet run, every
    time, and finite."""
    torch.manual_seed(3)
    x = torch.rand(shape, device=DEV) * 2 - 1
    gy = torch.rand(shape, device=DEV) * 2 - 1
    shift = torch.rand(3, shape[2], device=DEV) * 2 - 1
    gx0, gs0 = _backward(x, shift, gy)
    torch.cuda.synchronize()
    assert torch.isfinite(gs0).all()

    side = torch.cuda.Stream()
    a = torch.randn(4096, 4096, device=DEV)
    big = torch.randn(64 << 20, device=DEV)
    stop = time.time() + 1.5
    rounds = 0
    while time.time() < stop:
        with torch.cuda.stream(side):
            for _ in range(4):
                a = torch.tanh(a @ a) * 0.5          # MFMA-bound neighbour
                big.mul_(1.0001)                      # HBM-bound neighbour
        for _ in range(8):
            gx, gs = _backward(x, shift, gy)
            assert torch.equal(gs, gs0), "d(shift) changed under contention"
            assert torch.equal(gx, gx0), "d(x) changed under contention"
        rounds += 1
    torch.cuda.synchronize()
    assert rounds >= 1


def test_a_finalizer_without_producers_gives_up_with_nan_instead_of_hanging():
    """rk3d_debug_finalize_only_f32 launches ONLY the finalizer waves of the fused 3-D backward over a workspace nobody
    publishes to, with the poll budget cut from ~2 s to a few thousand polls: the launch must end, and every d(shift)
    value must be NaN (the documented give-up, rk_dma.hpp fin_collect) -- for a workspace of zeros, of stale random bits and
    of granules carrying a PREVIOUS launch's tag."""
    from rubiksnet_amd import _native
    L = _native.lib()
    C, P = 48, 100
    prev = L.rk_debug_set_finalize_spins(4000)
    try:
        assert prev > 1000000, "the default poll budget is ~2 s"
        stream = torch.cuda.current_stream().cuda_stream
        for fill in ("zeros", "random", "old-tag"):
            if fill == "zeros":
                ws = torch.zeros(C * 3 * P * 16, dtype=torch.uint8, device=DEV)
            elif fill == "random":
                ws = torch.randint(0, 256, (C * 3 * P * 16,), dtype=torch.uint8, device=DEV)
            else:
                tag = int(L.rk_debug_peek_launch_tag())
                old = (tag - 1) & 0xffffffff or 0xfffffffe
                tag2 = (old * 2654435761 ^ 0x9e3779b9) & 0xffffffff
                val = np.float32(1.5).view(np.uint32)
                pair = np.array([val, old, ~val & np.uint32(0xffffffff), tag2], dtype=np.uint32)   # a complete, consistent pair of launch `old`
                ws = torch.from_numpy(np.tile(pair, C * 3 * P).view(np.uint8)).to(DEV)
            gs = torch.zeros(3, C, device=DEV)
            t0 = time.time()
            _native.check(L.rk3d_debug_finalize_only_f32(ws.data_ptr(), ws.numel(), C, P, gs.data_ptr(), 1, 1.0, stream), "finalize_only")
            torch.cuda.synchronize()
            assert time.time() - t0 < 5.0, "the give-up path took %.1f s" % (time.time() - t0)
            assert torch.isnan(gs).all(), "fill=%s: a finalizer without partials must write NaN, got %s" % (fill, gs[:, :4])
    finally:
        assert L.rk_debug_set_finalize_spins(0) == 4000          # back to the default
    assert L.rk_debug_set_finalize_spins(0) > 1000000


def test_a_finalizer_accepts_exactly_this_launchs_pairs():
    """The other side of the hand-off through the same hook: a workspace pre-filled with complete pairs carrying THIS
    launch's tag is summed (fp64, index order) and normalised (K5) -- so the NaN of the previous test is the tag check
    speaking, not a broken hook."""
    from rubiksnet_amd import _native
    L = _native.lib()
    C, P = 5, 70
    rng = np.random.default_rng(0)
    vals = rng.uniform(-1, 1, (C, 3, P)).astype(np.float32)
    tag = int(L.rk_debug_peek_launch_tag())
    tag2 = (tag * 2654435761 ^ 0x9e3779b9) & 0xffffffff
    bits = vals.view(np.uint32)
    gran = np.empty((C, 3, P, 4), dtype=np.uint32)
    gran[..., 0] = bits
    gran[..., 1] = tag
    gran[..., 2] = ~bits
    gran[..., 3] = tag2
    ws = torch.from_numpy(gran.reshape(-1).view(np.uint8)).to(DEV)
    gs = torch.zeros(3, C, device=DEV)
    _native.check(L.rk3d_debug_finalize_only_f32(ws.data_ptr(), ws.numel(), C, P, gs.data_ptr(), 1, 1.0,
                                                 torch.cuda.current_stream().cuda_stream), "finalize_only")
    torch.cuda.synchronize()
    s = vals.astype(np.float64).sum(axis=2).astype(np.float32)            # [C, 3]
    want = (s / np.sqrt((s * s).sum(axis=1, keepdims=True, dtype=np.float32))).T
    np.testing.assert_allclose(gs.cpu().numpy(), want, rtol=0, atol=2e-6)
    # consumed pairs are retired (tag 0): a replay of the same launch cannot take them for its own
    assert int(torch.count_nonzero(ws)) == 0
