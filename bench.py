#!/usr/bin/env python
"""bench.py -- RubiksShift3D fwd+bwd effective GB/s vs the MI355X HBM roofline (BASELINE.json
`metric`, configs[1]), plus RubiksNet-Tiny train-step clips/s, next to the CPU oracle.

    python bench.py --gpus N --steps K --warmup W

A "step" is one forward + one backward of the operator over one batch of synthetic input,
x [32, 8, 64, 56, 56] fp32 (layout [N,T,C,H,W]), stride 1, pad 0, normalize_grad on.
Inputs are resident in HBM before the timed region.  For N > 1 there is one rank per GPU (RCCL):
either launched by `python -m torch.distributed.run ... bench.py --gpus N` (RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* in the environment), or -- when WORLD_SIZE is not set -- bench.py re-executes
itself under torch.distributed.run on 127.0.0.1.  Every rank runs the same per-GPU batch: the path
shards along clips with no data-path collective (weak scaling); `value` is the whole-job aggregate.
Rank 0 prints ONE JSON line.  `models` holds the model-level legs of BASELINE.json configs[2..4]
(per-GPU share of the global batch; DDP all-reduce over RCCL for N > 1).

Without a GPU (`--dry-run`, implied when torch.cuda is unavailable) only the launcher, the rendezvous,
the barrier / max-over-ranks timing and the all-reduce probe run, on gloo: the product has no CPU path.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from rubiksnet_amd import dp, rubiksnet_cuda  # noqa: E402

SHAPE = (32, 8, 64, 56, 56)          # N, T, C, H, W
HBM_PEAK_GBS = 8000.0                # MI355X HBM3E spec (MI355X_MICROARCH.md); 6290 measured copy ceiling
COPY_CEILING_GBS = 6290.0


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def op_bench(env, steps, warmup, nsets=4, settle_s=0.4):
    dev = env.device
    N, T, C, H, W = SHAPE
    numel = N * T * C * H * W
    g = torch.Generator(device="cpu").manual_seed(0)
    shift = (torch.rand(3, C, generator=g) * 2 - 1).to(dev)
    sets = []
    for _ in range(nsets):   # rotate buffer sets: one 205 MB tensor fits the 256 MiB Infinity Cache
        x = torch.empty(SHAPE, device=dev).uniform_(-1, 1)
        gy = torch.empty(SHAPE, device=dev).uniform_(-1, 1)
        sets.append((x, gy, torch.empty_like(x), torch.empty_like(x)))
    gshift = torch.empty(3, C, device=dev)
    s1, p0 = [1, 1, 1], [0, 0, 0]
    it = [0]

    def step(record=False):
        # forward on set i, backward on set i+2 of FOUR: no launch reads a tensor its predecessor touched (with three
        # sets and "backward on i+1", step k's backward read the x that step k+1's forward read as its very next launch),
        # and a tensor is re-read only after >= 1.4 GB of other traffic -- it cannot still sit in the 256 MiB Infinity Cache.
        x, _, y, _ = sets[it[0] % nsets]
        xb, gy, _, gx = sets[(it[0] + 2) % nsets]
        it[0] += 1
        rubiksnet_cuda.rubiks_shift_3d_forward_float(x, shift, s1, p0, False, y)
        rubiksnet_cuda.rubiks_shift_3d_backward_float(xb, shift, gy, s1, p0, gx, gshift, True, 1.0, False)

    # The chip enters from a low-power state (sclk 95 MHz idle) and its power management needs a few hundred
    # milliseconds of continuous load to settle: the first ~10 ms run FASTER-then-SLOWER than steady state
    # (profiles/r02_sustained.txt: 183 -> 201 us/step inside the first 60 steps, 179.0 +- 0.3 us/step from 20 ms
    # on, for seconds).  K = 20..50 timed steps are only 4..10 ms, so without this they would measure the
    # transient.  Untimed, before the W warm-up steps.
    t_settle = time.perf_counter()
    while time.perf_counter() - t_settle < settle_s:
        for _ in range(20):
            step()
        torch.cuda.synchronize()
    for _ in range(warmup):
        step()
    # The K timed steps run bare.  Per-kernel launch durations come from two more passes of the same K launches
    # right after the bracket closes -- K forwards, then K backwards (+ finalize), each pass bracketed by ONE pair of
    # HIP events on the launch stream: event records between the kernels of a step cost ~15 us/step of marker
    # packets and idle gaps (tools/launch_gap_probe.py), which also changes the clock regime being measured.
    elapsed = dp.timed_region(env, lambda: step(False), steps)

    def kernel_pass(fn):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            fn()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / steps

    def fwd_only():
        x, _, y, _ = sets[it[0] % nsets]
        it[0] += 1
        rubiksnet_cuda.rubiks_shift_3d_forward_float(x, shift, s1, p0, False, y)

    def bwd_only():
        xb, gy, _, gx = sets[it[0] % nsets]
        it[0] += 1
        rubiksnet_cuda.rubiks_shift_3d_backward_float(xb, shift, gy, s1, p0, gx, gshift, True, 1.0, False)

    # the dominant KERNEL by itself, through the two-phase entry points of the C ABI (rk3d_backward_f32 is exactly
    # rk3d_backward_partials_f32 = the backward kernel, then rk3d_backward_finalize_f32 = row-sum + K5)
    import ctypes

    from rubiksnet_amd import _native
    L = _native.lib()
    ws_bytes = int(L.rk3d_backward_workspace_bytes(N, T, C, H, W, 1, 1, 1, 0, 0, 0, 4))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    nparts = ctypes.c_int(0)
    stream = torch.cuda.current_stream(dev).cuda_stream

    def bwd_kernel_only():
        xb, gy, _, gx = sets[it[0] % nsets]
        it[0] += 1
        rc = L.rk3d_backward_partials_f32(xb.data_ptr(), shift.data_ptr(), gy.data_ptr(), gx.data_ptr(), N, T, C, H, W,
                                          1, 1, 1, 0, 0, 0, 0, ws.data_ptr(), ws_bytes, ctypes.byref(nparts), stream)
        _native.check(rc, "rk3d_backward_partials_f32")

    def finalize_only():
        rc = L.rk3d_backward_finalize_f32(ws.data_ptr(), C, nparts.value, gshift.data_ptr(), 1, 1.0, stream)
        _native.check(rc, "rk3d_backward_finalize_f32")

    fwd_ms = kernel_pass(fwd_only)
    bwd_ms = kernel_pass(bwd_only)
    bwd_kernel_ms = kernel_pass(bwd_kernel_only)
    finalize_ms = kernel_pass(finalize_only)
    return {
        "elapsed_s": elapsed, "numel": numel, "fwd_ms": fwd_ms, "bwd_ms": bwd_ms, "bwd_kernel_ms": bwd_kernel_ms,
        "finalize_ms": finalize_ms,
        "bytes_fwd": 8 * numel, "bytes_bwd": 12 * numel,
    }


# how every secondary leg (rk2d, secondary, tshift, pw_*, bn_bwd_dx) is timed -- stated in the JSON so that nobody compares
# these numbers with a mean over launch-by-launch calls (rounds 1-4 reported that; the headline leg still does, per contract)
SECONDARY_TIMING = ("untimed run-in, then a captured hipGraph of `iters` back-to-back launches of ONE kernel, replayed 4 times "
                    "between one pair of HIP events; the figure is the BEST (minimum) of 3 such measurements, per launch")


def _steady(fn, iters, settle_s):
    """Seconds per launch of `fn` on the GPU: an untimed run-in (clock / power transient), then a captured hipGraph of
    `iters` back-to-back launches replayed between ONE pair of HIP events.  The replay keeps Python and the launch path out
    of the number: issued one call at a time, kernels below ~16 us (the 7x7 and 14x14 shift kernels, the bf16 GEMMs) read
    as the host's launch period instead of their own duration (tools/op3d_graph_time.py vs tools/op3d_time.py).  Every C-ABI
    entry point is capturable (no allocation, no synchronisation inside); `fn` must launch on torch's CURRENT stream."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = None
    with torch.cuda.stream(side):
        t_end = time.perf_counter() + settle_s
        while time.perf_counter() < t_end:
            for _ in range(10):
                fn()
            torch.cuda.synchronize()
        try:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):
                for _ in range(iters):
                    fn()
        except Exception as exc:          # not capturable: fall back to launch-by-launch timing
            log("bench: graph capture failed (%r); timing launch by launch" % (exc,))
            graph = None
            torch.cuda.synchronize()
        best = None
        for _ in range(3):
            if graph is not None:
                graph.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            reps = 4 if graph is not None else 1
            for _ in range(reps):
                if graph is not None:
                    graph.replay()
                else:
                    for _ in range(iters):
                        fn()
            e1.record()
            e1.synchronize()
            t = e0.elapsed_time(e1) / (reps * iters) * 1e-3
            best = t if best is None else min(best, t)
    torch.cuda.current_stream().wait_stream(side)
    return best


def _cur_stream(dev):
    return torch.cuda.current_stream(dev).cuda_stream


def tshift_bench(env, iters=60, settle_s=0.2):
    """The temporal 3-tap filter of AttentionShift (rubiksnet/attention_shift.py:29-39; every block of the -aq networks,
    BASELINE configs[4]) on the first-stage shape of Large-AQ at 32 clips: x [256, 72, 56, 56], bf16 and fp32, through
    the C ABI.  Algorithmic bytes: forward 2 passes (read x, write y), backward 3 (read gy, read x, write gx)."""
    import ctypes  # noqa: F401

    from rubiksnet_amd import _native
    L = _native.lib()
    dev = env.device
    stream = lambda: _cur_stream(dev)    # noqa: E731 -- at call time: the legs are captured on a side stream
    NT, S, C, H = 256, 8, 72, 56
    out = {"x": [NT, C, H, H], "n_segment": S}
    for dtype, name in ((torch.bfloat16, "bf16"), (torch.float32, "f32")):
        sets = [(torch.randn(NT, C, H, H, device=dev).to(dtype), torch.randn(NT, C, H, H, device=dev).to(dtype),
                 torch.empty(NT, C, H, H, device=dev, dtype=dtype)) for _ in range(3)]
        taps = torch.softmax(torch.randn(C, 3, device=dev), 1).contiguous()
        gtaps = torch.empty_like(taps)
        wsb = int(L.rk_tshift3_backward_workspace_bytes(NT, S, C, H * H))
        ws = torch.empty(max(wsb, 1), dtype=torch.uint8, device=dev)
        it = [0]

        def fwd():
            x, _, y = sets[it[0] % 3]
            it[0] += 1
            _native.check(getattr(L, "rk_tshift3_forward_" + name)(x.data_ptr(), taps.data_ptr(), y.data_ptr(), NT, S, C,
                                                                   H * H, stream()), "rk_tshift3_forward")

        def bwd():
            x, g, y = sets[it[0] % 3]
            it[0] += 1
            _native.check(getattr(L, "rk_tshift3_backward_" + name)(g.data_ptr(), x.data_ptr(), taps.data_ptr(), y.data_ptr(),
                                                                    gtaps.data_ptr(), NT, S, C, H * H, ws.data_ptr(), wsb,
                                                                    stream()), "rk_tshift3_backward")

        tf, tb = _steady(fwd, iters, settle_s), _steady(bwd, iters, settle_s)
        nb = NT * C * H * H * sets[0][0].element_size()
        out[name] = {"fwd_us": tf * 1e6, "bwd_us": tb * 1e6, "fwd_GBps": 2 * nb / tf / 1e9, "bwd_GBps": 3 * nb / tb / 1e9,
                     "fwd_plus_bwd_frac_of_hbm_peak": 5 * nb / (tf + tb) / 1e9 / HBM_PEAK_GBS}
        del sets
        torch.cuda.empty_cache()
    return out


def pw16_bench(env, iters=40, settle_s=0.2):
    """The bf16 1x1 convolution kernels of rk_pw16.hip (SURVEY 8(f) f1 under configs[4]'s autocast) on the layer 70 of
    Large-AQ's 102 convolutions have, [256, 288 -> 288, 14, 14], through the C ABI: forward, forward + residual, d(input),
    d(weight).  Algorithmic bytes: one read per operand, one write per result (the weights are L2-resident)."""
    from rubiksnet_amd import _native
    L = _native.lib()
    dev = env.device
    stream = lambda: _cur_stream(dev)    # noqa: E731
    Fr, K, M, P = 256, 288, 288, 196
    sets = [(torch.randn(Fr, K, P, device=dev).bfloat16(), torch.randn(Fr, M, P, device=dev).bfloat16(),
             torch.empty(Fr, M, P, device=dev, dtype=torch.bfloat16), torch.empty(Fr, K, P, device=dev, dtype=torch.bfloat16))
            for _ in range(3)]
    w = torch.randn(M, K, device=dev) / K ** 0.5
    pf = torch.empty(int(L.rk_pw_packed_bytes(M, K)), dtype=torch.uint8, device=dev)
    pb = torch.empty(int(L.rk_pw_packed_bytes(K, M)), dtype=torch.uint8, device=dev)
    _native.check(L.rk_pw_pack_bf16(w.data_ptr(), M, K, pf.data_ptr(), pb.data_ptr(), stream()), "rk_pw_pack_bf16")
    nb = int(L.rk_pw_wgrad16_workspace_bytes(Fr, K, M, P))
    ws = torch.empty(max(nb, 1), dtype=torch.uint8, device=dev)
    dw = torch.empty(M, K, device=dev)
    it = [0]

    def nxt():
        it[0] += 1
        return sets[it[0] % 3]

    def fwd():
        x, g, y, o = nxt()
        _native.check(L.rk_pw_gemm_packed_bf16(pf.data_ptr(), x.data_ptr(), None, y.data_ptr(), Fr, K, M, P, stream()), "gemm")

    def fwd_res():
        x, g, y, o = nxt()
        _native.check(L.rk_pw_gemm_packed_bf16(pf.data_ptr(), x.data_ptr(), g.data_ptr(), y.data_ptr(), Fr, K, M, P, stream()), "gemm")

    def dgrad():
        x, g, y, o = nxt()
        _native.check(L.rk_pw_gemm_packed_bf16(pb.data_ptr(), g.data_ptr(), None, o.data_ptr(), Fr, M, K, P, stream()), "dgrad")

    def wgrad():
        x, g, y, o = nxt()
        _native.check(L.rk_pw_wgrad16_bf16(g.data_ptr(), x.data_ptr(), dw.data_ptr(), Fr, K, M, P, ws.data_ptr(), nb, stream()), "wgrad")

    e = Fr * K * P * 2
    out = {"layer": [Fr, K, M, 14, 14], "dtype": "bf16 activations, fp32 accumulation, fp32 d(weight)"}
    for name, fn, passes in (("fwd", fwd, 2), ("fwd_residual", fwd_res, 3), ("dgrad", dgrad, 2), ("wgrad", wgrad, 2)):
        t = _steady(fn, iters, settle_s)
        out[name] = {"us": t * 1e6, "GBps": passes * e / t / 1e9, "frac_of_hbm_peak": passes * e / t / 1e9 / HBM_PEAK_GBS}
    return out


MFMA_F32_PEAK_TFLOPS = 157.3   # v_mfma_f32_* at 64 FLOP/clk/SIMD x 1024 SIMDs x 2.4 GHz (MI355X_MICROARCH.md)


def pw32_bench(env, iters=30, settle_s=0.2):
    """The fp32 1x1 convolution kernels (rk_pw.hip / rk_pw2.hip behind the public entry points, whichever the dispatch
    picks) on the layer 70 of RubiksNet-Large's 102 convolutions have, [256, 288 -> 288, 14, 14], on the 28x28 stage's
    [256, 144 -> 144, 28, 28], and on the shallow 56x56 layers of Tiny / Large ([256, 54 -> 54], [256, 72 -> 72]: the
    streaming kernel of rk_pw4.hip; as loaded on HBM as on the matrix pipe, so both fractions are reported): forward, forward + residual + the next BatchNorm's tile statistics (conv3 of a fused
    training block), d(input), d(weight).  MFMA-bound at fp32: fraction of the 157.3 TFLOP/s f32 MFMA peak."""
    from rubiksnet_amd import _native
    L = _native.lib()
    dev = env.device
    stream = lambda: _cur_stream(dev)    # noqa: E731
    out = {"peak_TFLOPs": MFMA_F32_PEAK_TFLOPS, "dtype": "f32 in, f32 accumulate (v_mfma_f32_16x16x4_f32 / 32x32x2_f32)"}
    for Fr, K, M, H in ((256, 288, 288, 14), (256, 144, 144, 28), (256, 54, 54, 56), (256, 72, 72, 56)):
        P = H * H
        sets = [(torch.randn(Fr, K, P, device=dev), torch.randn(Fr, M, P, device=dev), torch.empty(Fr, M, P, device=dev),
                 torch.empty(Fr, K, P, device=dev)) for _ in range(2)]
        w = torch.randn(M, K, device=dev) / K ** 0.5
        nb = int(L.rk_pw_wgrad_workspace_bytes(Fr, K, M, P))
        ws = torch.empty(max(nb, 1), dtype=torch.uint8, device=dev)
        dw = torch.empty(M, K, device=dev)
        J = int(L.rk_pw_gemm_tiles(w.data_ptr(), Fr, K, M, P, 1))
        stats = torch.empty(M, J, 4, device=dev)
        it = [0]

        def nxt():
            it[0] += 1
            return sets[it[0] % 2]

        def fwd():
            x, g, y, o = nxt()
            _native.check(L.rk_pw_gemm_f32(w.data_ptr(), x.data_ptr(), None, y.data_ptr(), Fr, K, M, P, 1, stream()), "gemm")

        def fwd_res_stats():
            x, g, y, o = nxt()
            _native.check(L.rk_pw_gemm_stats_f32(w.data_ptr(), x.data_ptr(), g.data_ptr(), y.data_ptr(), Fr, K, M, P, 1, None,
                                                 None, 0, stats.data_ptr(), J, stream()), "gemm_stats")

        def dgrad():
            x, g, y, o = nxt()
            _native.check(L.rk_pw_gemm_f32(w.data_ptr(), g.data_ptr(), None, o.data_ptr(), Fr, M, K, P, 0, stream()), "dgrad")

        def wgrad():
            x, g, y, o = nxt()
            _native.check(L.rk_pw_wgrad_f32(g.data_ptr(), x.data_ptr(), dw.data_ptr(), Fr, K, M, P, ws.data_ptr(), nb, stream()),
                          "wgrad")
        flop = 2.0 * Fr * P * K * M
        leg = {"layer": [Fr, K, M, H, H], "GFLOP": flop / 1e9}
        # the shallow layers (54 / 72 channels on 56 x 56: rk_pw4.hip) load HBM as much as the matrix pipe: both fractions
        e = 4.0 * Fr * P
        hbm = {"fwd": e * (K + M), "fwd_residual_stats": e * (K + 2 * M), "dgrad": e * (K + M), "wgrad": e * (K + M)}
        for name, fn in (("fwd", fwd), ("fwd_residual_stats", fwd_res_stats), ("dgrad", dgrad), ("wgrad", wgrad)):
            t = _steady(fn, iters, settle_s)
            leg[name] = {"us": t * 1e6, "TFLOPs": flop / t / 1e12, "frac_of_mfma_peak": flop / t / 1e12 / MFMA_F32_PEAK_TFLOPS,
                         "frac_of_hbm_peak": hbm[name] / t / 1e9 / HBM_PEAK_GBS}
        out["%dx%d_%dch" % (H, H, K)] = leg
        del sets, ws
    return out


def bn_bench(env, iters=40, settle_s=0.2):
    """BatchNorm's remaining d(x) pass of a fused training block (rk_bn_bwd_dx_pre_f32: dx = gamma invstd (dz - k1 -
    xhat k2) (+ skip)), the top kernel of the fp32 train steps by time.  Algorithmic bytes: dz, x (, skip) read, dx written."""
    from rubiksnet_amd import _native
    L = _native.lib()
    dev = env.device
    stream = lambda: _cur_stream(dev)    # noqa: E731
    out = {}
    for Fr, C, H in ((256, 288, 14), (256, 72, 56)):
        P = H * H
        sets = [(torch.randn(Fr, C, P, device=dev), torch.randn(Fr, C, P, device=dev), torch.randn(Fr, C, P, device=dev),
                 torch.empty(Fr, C, P, device=dev)) for _ in range(3)]
        gamma, mean, inv = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev), torch.rand(C, device=dev) + 0.5
        k12 = torch.randn(2, C, device=dev) * 0.01
        it = [0]

        def nxt():
            it[0] += 1
            return sets[it[0] % 3]

        def plain():
            dz, x, sk, dx = nxt()
            _native.check(L.rk_bn_bwd_dx_pre_f32(dz.data_ptr(), x.data_ptr(), gamma.data_ptr(), mean.data_ptr(), inv.data_ptr(),
                                                 k12.data_ptr(), None, dx.data_ptr(), Fr, C, P, stream()), "dx_pre")

        def with_skip():
            dz, x, sk, dx = nxt()
            _native.check(L.rk_bn_bwd_dx_pre_f32(dz.data_ptr(), x.data_ptr(), gamma.data_ptr(), mean.data_ptr(), inv.data_ptr(),
                                                 k12.data_ptr(), sk.data_ptr(), dx.data_ptr(), Fr, C, P, stream()), "dx_pre")
        e = 4.0 * Fr * C * P
        leg = {"tensor": [Fr, C, H, H]}
        for name, fn, passes in (("dx", plain, 3), ("dx_plus_skip", with_skip, 4)):
            t = _steady(fn, iters, settle_s)
            leg[name] = {"us": t * 1e6, "GBps": passes * e / t / 1e9, "frac_of_hbm_peak": passes * e / t / 1e9 / HBM_PEAK_GBS}
        out["%dx%d_%dch" % (H, H, C)] = leg
        del sets
    return out


def op2d_bench(env, iters=60, settle_s=0.3):
    """the 2-D operator of the -aq networks on the same number of elements
    ([256,64,56,56] = 32 clips x 8 frames), fp32 and bf16.  Same method as the 3-D leg: an untimed run-in (the
    chip's clock / power transient), then `iters` back-to-back launches of one kernel between ONE pair of events."""
    dev = env.device
    out = {}
    for dtype, name in ((torch.float32, "f32"), (torch.bfloat16, "bf16")):
        shape = (SHAPE[0] * SHAPE[1],) + SHAPE[2:]
        sets = []
        for _ in range(3):
            x = torch.empty(shape, device=dev, dtype=dtype).uniform_(-1, 1)
            gy = torch.empty(shape, device=dev, dtype=dtype).uniform_(-1, 1)
            sets.append((x, gy, torch.empty_like(x), torch.empty_like(x)))
        g = torch.Generator(device="cpu").manual_seed(1)
        shift = (torch.rand(2, shape[1], generator=g) * 1.8 - 0.9).to(dev).to(dtype)
        gs = torch.empty_like(shift)
        it = [0]

        def fwd():
            x, _, y, _ = sets[it[0] % 3]
            it[0] += 1
            rubiksnet_cuda.rubiks2d_forward(x, shift, [1, 1], [0, 0], False, y)

        def bwd():
            xb, gy, _, gx = sets[it[0] % 3]
            it[0] += 1
            rubiksnet_cuda.rubiks2d_backward(gy, xb, shift, [1, 1], [0, 0], True, True, False, gx, gs)

        def timed(fn):
            return _steady(fn, iters, settle_s)

        tf, tb = timed(fwd), timed(bwd)
        es, n = x.element_size(), x.numel()
        out[name] = {"fwd_us": tf * 1e6, "bwd_us": tb * 1e6, "fwd_GBps": 2 * es * n / tf / 1e9,
                     "bwd_GBps": 3 * es * n / tb / 1e9,
                     "fwd_plus_bwd_frac_of_hbm_peak": 5 * es * n / (tf + tb) / 1e9 / HBM_PEAK_GBS}
    out["shape"] = [SHAPE[0] * SHAPE[1]] + list(SHAPE[2:])
    out["timing"] = SECONDARY_TIMING
    # SURVEY 8 a12's own example: the 35 layer-3 blocks of Large-AQ at 32 clips per GPU, [256, 288, 14, 14] in bf16
    shape = (256, 288, 14, 14)
    sets = [(torch.empty(shape, device=dev, dtype=torch.bfloat16).uniform_(-1, 1),
             torch.empty(shape, device=dev, dtype=torch.bfloat16).uniform_(-1, 1),
             torch.empty(shape, device=dev, dtype=torch.bfloat16), torch.empty(shape, device=dev, dtype=torch.bfloat16))
            for _ in range(3)]
    g = torch.Generator(device="cpu").manual_seed(3)
    shift = (torch.rand(2, shape[1], generator=g) * 1.8 - 0.9).to(dev)          # fp32 table next to bf16 activations
    gs = torch.empty_like(shift)
    it = [0]

    def fwd14():
        x, _, y, _ = sets[it[0] % 3]
        it[0] += 1
        rubiksnet_cuda.rubiks2d_forward(x, shift, [1, 1], [0, 0], False, y)

    def bwd14():
        xb, gy, _, gx = sets[it[0] % 3]
        it[0] += 1
        rubiksnet_cuda.rubiks2d_backward(gy, xb, shift, [1, 1], [0, 0], True, True, False, gx, gs)

    tf, tb = _steady(fwd14, iters, settle_s), _steady(bwd14, iters, settle_s)
    n = sets[0][0].numel()
    out["bf16_14x14"] = {"x": list(shape), "shift_dtype": "f32", "fwd_us": tf * 1e6, "bwd_us": tb * 1e6,
                         "fwd_GBps": 4 * n / tf / 1e9, "bwd_GBps": 6 * n / tb / 1e9,
                         "fwd_plus_bwd_frac_of_hbm_peak": 10 * n / (tf + tb) / 1e9 / HBM_PEAK_GBS}
    return out


def secondary_points(env, iters=40, settle_s=0.2):
    """SURVEY 8(d) secondary points of the 3-D operator: the two real-model extremes -- the stride-(1,2,2) layer
    [32,8,54,112,112] and the small-plane layer [32,8,216,14,14] -- and the benchmark shape with quantize=True.
    Algorithmic bytes: 4 (numel_in + numel_out) forward, 4 (numel_out + 2 numel_in) backward.  Timed like the
    other operator legs (run-in, then back-to-back launches of one kernel between one pair of events)."""
    dev = env.device
    out = {}
    points = (("stride_1_2_2", (32, 8, 54, 112, 112), [1, 2, 2], False),
              ("planes_14x14", (32, 8, 216, 14, 14), [1, 1, 1], False),             # layer3 of Tiny (5 layers)
              ("planes_14x14_288ch", (32, 8, 288, 14, 14), [1, 1, 1], False),       # layer3 of Large (35 of its 51 layers)
              ("planes_14x14_72ch", (32, 8, 72, 14, 14), [1, 1, 1], False),         # (third point of the launch-cost line below)
              ("planes_7x7", (32, 8, 576, 7, 7), [1, 1, 1], False),                 # layer4 of Large
              ("stride_1_2_2_28to14", (32, 8, 288, 28, 28), [1, 2, 2], False),      # the third downsampling layer of Large
              ("quantize", SHAPE, [1, 1, 1], True))
    for name, shape, stride, quantize in points:
        N, T, C, H, W = shape
        Ho, Wo = (H - 1) // stride[1] + 1, (W - 1) // stride[2] + 1
        oshape = (N, T, C, Ho, Wo)
        g = torch.Generator(device="cpu").manual_seed(2)
        shift = (torch.rand(3, C, generator=g) * 2 - 1).to(dev)
        sets = []
        for _ in range(3):
            sets.append((torch.empty(shape, device=dev).uniform_(-1, 1), torch.empty(oshape, device=dev).uniform_(-1, 1),
                         torch.empty(oshape, device=dev), torch.empty(shape, device=dev)))
        gs = torch.empty(3, C, device=dev)
        it = [0]

        def fwd():
            x, _, y, _ = sets[it[0] % 3]
            it[0] += 1
            rubiksnet_cuda.rubiks_shift_3d_forward_float(x, shift, stride, [0, 0, 0], quantize, y)

        def bwd():
            x, gy, _, gx = sets[it[0] % 3]
            it[0] += 1
            rubiksnet_cuda.rubiks_shift_3d_backward_float(x, shift, gy, stride, [0, 0, 0], gx, gs, True, 1.0, quantize)

        def timed(fn):
            return _steady(fn, iters, settle_s)

        tf, tb = timed(fwd), timed(bwd)
        nin, nout = N * T * C * H * W, N * T * C * Ho * Wo
        bf, bb = 4 * (nin + nout), 4 * (nout + 2 * nin)
        out[name] = {"x": list(shape), "stride": stride, "quantize": quantize, "fwd_us": tf * 1e6, "bwd_us": tb * 1e6,
                     "fwd_GBps": bf / tf / 1e9, "bwd_GBps": bb / tb / 1e9,
                     "fwd_plus_bwd_frac_of_hbm_peak": (bf + bb) / (tf + tb) / 1e9 / HBM_PEAK_GBS}
        del sets
        torch.cuda.empty_cache()
    # What a small-plane launch costs before it streams a byte, and how fast it streams once it does: the least-squares line
    # t(C) = a + b C through the three 14x14 points (tools/fixed_cost_probe.py does the same over six; profiles/
    # r06_small_plane_fixed_cost.txt).  A 14x14 tensor of 32 clips is 43-58 MB: at ~6 TB/s it streams in 7-10 us, so the
    # ~4 us (forward) / ~6 us (backward: + the in-launch row-sum's hand-off) a launch costs up front is what the
    # fraction-of-peak of these legs measures, not the kernels' streaming rate.
    try:
        pts = [out[k] for k in ("planes_14x14_72ch", "planes_14x14", "planes_14x14_288ch")]
        cs = [p["x"][2] for p in pts]
        n = len(cs)
        mc = sum(cs) / n
        fit = {}
        for key, per_c in (("fwd", 8 * 32 * 8 * 196), ("bwd", 12 * 32 * 8 * 196)):
            ts = [p[key + "_us"] for p in pts]
            mt = sum(ts) / n
            b = sum((c - mc) * (t - mt) for c, t in zip(cs, ts)) / sum((c - mc) ** 2 for c in cs)
            fit[key] = {"launch_cost_us": mt - b * mc, "marginal_GBps": per_c / b / 1e3,
                        "marginal_frac_of_hbm_peak": per_c / b / 1e3 / HBM_PEAK_GBS}
        out["planes_14x14_launch_cost_fit"] = fit
    except Exception as exc:      # a diagnostic, never the reason a bench line is lost
        out["planes_14x14_launch_cost_fit"] = {"error": repr(exc)}
    return out


def pmc_traffic(kernel_substr):
    """HBM bytes per launch of the dominant kernel from the newest PMC summary committed under profiles/
    (tools/pmc.sh + tools/make_profile_summary.py: separate rocprofv3 --pmc passes, FETCH_SIZE doubled)."""
    import glob

    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")), reverse=True):
        with open(path) as f:
            data = json.load(f)
        for name, rec in data.items():
            if kernel_substr in name:
                return rec["hbm_bytes"], "%s :: %s" % (os.path.relpath(path, ROOT), name)
    return None, None


def cpu_baseline(budget_s=12.0):
    """The oracle (a port: the reference has no CPU path) on the host cores, OpenMP over planes,
    on a bounded sample of the same workload: n clips of [8,64,56,56], fwd+bwd."""
    import numpy as np

    from oracle import oracle as orc

    orc.build()
    native = orc.use_native()                     # -march=native build for this host's cores
    threads = os.cpu_count() or 1
    orc.set_threads(threads)
    _, T, C, H, W = SHAPE
    rng = np.random.default_rng(0)
    shift = rng.uniform(-1, 1, (3, C)).astype(np.float32)

    def run(n):
        x = rng.uniform(-1, 1, (n, T, C, H, W)).astype(np.float32)
        gy = rng.uniform(-1, 1, (n, T, C, H, W)).astype(np.float32)
        t0 = time.perf_counter()
        orc.rk3d_forward(x, shift)
        orc.rk3d_backward(gy, x, shift)
        return time.perf_counter() - t0

    run(1)                                        # warm-up (thread pool, page faults)
    n = min(32, max(1, threads // 8))             # enough (n, t) planes to occupy every thread
    reps, dt = 0, 0.0
    while dt < budget_s and reps < 200:           # bounded sample: ~budget_s seconds of CPU work
        dt += run(n)
        reps += 1
    n *= reps
    bytes_total = 20 * n * T * C * H * W
    return {
        "value": bytes_total / dt / 1e9, "unit": "GB/s", "cores": threads, "kind": "port",
        "note": "the oracle: an order-preserving CHECKER (scalar loop nests in the reference's expression order, "
                "-ffp-contract=off), not a tuned CPU implementation -- a stated baseline, not a target; "
                "the roofline fraction is what measures the kernels",
        "when": "before the GPU legs of this run, rank 0, N = 1",
        "sample": "%d clips of [8,64,56,56] fp32 fwd+bwd (same shift/stride/pad as the GPU run), oracle C restatement "
                  "(-O3 %s, OpenMP over (n,t) planes forward / (c,row) backward), %d threads, %.2f s"
                  % (n, "-march=native" if native else "-march=x86-64-v3", threads, dt),
        "threads_effective": threads,
        "clips_per_s": n / dt,
    }


def cpu_model_baseline(budget_s=12.0, batch=8):
    """The model-level CPU column (north_star: RubiksNet-Tiny forward + backward "next to the reference run on the host CPU
    cores"): the same RubiksNet-Tiny module on host tensors -- PyTorch's CPU convolution / BatchNorm
    kernels around the oracle's RubiksShift3D plugged in as every layer's `shift_function` (oracle/torch_shift.py; the
    reference itself has no CPU path) -- one train step (forward + CE + backward + Adam) at a reduced batch,
    repeated for ~budget_s seconds."""
    from oracle import oracle as orc
    from oracle import torch_shift
    from rubiksnet_amd import RubiksNet

    orc.build()
    threads = os.cpu_count() or 1
    orc.set_threads(threads)
    torch.manual_seed(0)
    net = RubiksNet("tiny", num_classes=174, num_frames=8, verbose=False)
    layers = torch_shift.plug_into(net)
    opt = dp.make_optimizer(net, lr=1e-3, kind="adam")
    net.train()
    clips = torch.randn(batch, 8, 3, 224, 224)
    labels = torch.randint(0, 174, (batch,))

    def step():
        t0 = time.perf_counter()
        dp.train_step(net, opt, clips, labels)
        return time.perf_counter() - t0

    step()                                        # warm-up (thread pools, allocator, oneDNN primitive caches)
    reps, dt = 0, 0.0
    while dt < budget_s and reps < 50:
        dt += step()
        reps += 1
    return {
        "value": batch * reps / dt, "unit": "clips/s", "cores": threads, "kind": "port",
        "sample": "%d train steps (fwd + CE + bwd + Adam) of RubiksNet-Tiny on %d clips [8,3,224,224] fp32 on the host: "
                  "PyTorch CPU conv / BatchNorm kernels (%d intra-op threads) + the oracle's RubiksShift3D in all %d shift "
                  "layers (OpenMP, %d threads), %.2f s" % (reps, batch, torch.get_num_threads(), layers, threads, dt),
        "per_step_batch": batch, "ms_per_step": 1e3 * dt / reps,
        "note": "reduced batch (the GPU leg runs 32 clips per step); the shift layers run the order-preserving checker",
    }


# (name, tier, variant, autocast dtype, what) -- BASELINE.json configs[2], [3], [4] + the metric's Tiny train step
MODEL_LEGS = {
    "tiny-train": ("tiny", "rubiks3d", None, "train"),
    "tiny-fwd-b64": ("tiny", "rubiks3d", None, "forward"),          # configs[2]
    "small-train": ("small", "rubiks3d", None, "train"),            # the SE tier (fused squeeze / scale)
    "large-train": ("large", "rubiks3d", None, "train"),            # configs[3], per-GPU share of 256 / 8
    "large-aq-bf16-train": ("large", "rubiks3d-aq", torch.bfloat16, "train"),   # configs[4]
}


def run_in(env, step, min_s=1.5, max_s=12.0):
    times, t0 = [], time.perf_counter()
    while True:
        ts = time.perf_counter()
        step()
        if env.device.type == "cuda":
            torch.cuda.synchronize()
        times.append(time.perf_counter() - ts)
        elapsed = time.perf_counter() - t0
        settled = False
        if elapsed >= min_s and len(times) >= 12:
            a, b = sum(times[-6:]) / 6, sum(times[-12:-6]) / 6
            settled = abs(a - b) <= 0.03 * b
        flags = torch.tensor([0.0 if settled else 1.0, 1.0 if elapsed > max_s else 0.0], device=env.device)
        if env.distributed:
            torch.distributed.all_reduce(flags, op=torch.distributed.ReduceOp.MAX)
        if flags[0].item() == 0.0 or flags[1].item() == 1.0:      # everyone settled, or someone ran out of time
            return len(times)


def model_bench(env, leg, per_gpu_batch, steps, warmup):
    from rubiksnet_amd import RubiksNet

    tier, variant, amp, what = MODEL_LEGS[leg]
    dev = env.device
    torch.manual_seed(0)
    net = RubiksNet(tier, num_classes=174, num_frames=8, variant=variant, verbose=False).to(dev)
    if what == "forward":
        per_gpu_batch = 64
        net.eval()
        clips = torch.randn(per_gpu_batch, 8, 3, 224, 224, device=dev)

        def step():
            with torch.no_grad():
                net(clips)
        desc = "full forward (eval, no grad), fp32"
    else:
        model = dp.wrap_ddp(net, env)
        opt = dp.make_optimizer(model, lr=1e-3, kind="adam")
        clips = torch.randn(per_gpu_batch, 8, 3, 224, 224, device=dev)
        labels = torch.randint(0, 174, (per_gpu_batch,), device=dev)
        model.train()

        def step():
            with torch.autocast("cuda", dtype=amp, enabled=amp is not None):
                dp.train_step(model, opt, clips, labels)
        desc = "train step (fwd+bwd+Adam%s), %s" % (", DDP all-reduce" if env.distributed else "",
                                                     "bf16 autocast" if amp is not None else "fp32")
    # untimed run-in: a fresh process needs tens of steps before a train step reaches its sustained time (MIOpen /
    # rocBLAS pick and compile their kernels, the allocator warms up, then the clocks settle: the FIRST model leg of a
    # fresh box read 31-38 ms for Tiny with a fixed 1.5 s run-in, 26-28 ms in a second process).  Run until the step
    # time has settled -- the last 6 steps within 3 % of the 6 before -- for at least 1.5 s and at most ~12 s.  The
    # stop decision is taken collectively, so every rank runs the same number of steps (a DDP step is a collective).
    run_in(env, step)
    for _ in range(warmup):
        step()
    dt = dp.timed_region(env, step, steps)
    # the leg against its bound: every convolution / shift at its roofline, everything elementwise fused away
    from rubiksnet_amd.roofline import model_bound
    bound = model_bound(net, per_gpu_batch, train=(what != "forward"), compute="bf16" if amp is not None else "f32",
                        storage_bytes=2 if amp is not None else 4)
    ms = 1e3 * dt / steps
    bound["frac"] = bound["bound_ms"] / ms
    bound["clips_per_s_at_bound"] = per_gpu_batch * env.world_size / (bound["bound_ms"] * 1e-3)
    return {
        "name": "rubiksnet-%s%s" % (tier, "-aq" if variant.endswith("aq") else ""), "what": desc + ", synthetic clips",
        "per_gpu_batch": per_gpu_batch, "global_batch": per_gpu_batch * env.world_size, "steps": steps,
        "ms_per_step": ms, "clips_per_s": per_gpu_batch * env.world_size * steps / dt,
        "parallelism": "dp%d" % env.world_size, "roofline": bound,
    }


def feeder_bench(env, batch=32, iters=12):
    """the synthetic feeder alone -- pinned uint8 clips -> H2D on a side stream -> the device
    transform (transpose + /255 + normalise) -- and the transform kernel by itself."""
    from rubiksnet_amd.input_pipeline import SyntheticClipLoader, stacked_u8_to_clips

    loader = SyntheticClipLoader(batch=batch, device=env.device)
    it = iter(loader)
    for _ in range(3):
        next(it)
    dt = dp.timed_region(env, lambda: next(it), iters)
    u8 = torch.randint(0, 256, (batch, 224, 224, 24), dtype=torch.uint8, device=env.device)
    out = torch.empty(batch, 24, 224, 224, device=env.device)
    for _ in range(3):
        stacked_u8_to_clips(u8, 8, out=out)
    dk = dp.timed_region(env, lambda: stacked_u8_to_clips(u8, 8, out=out), iters) / iters
    return {"clips_per_s": batch * env.world_size * iters / dt, "per_gpu_batch": batch,
            "what": "pinned uint8 [B,224,224,24] -> H2D (side stream) -> transform, double buffered",
            "transform_us": 1e6 * dk, "transform_GBps": 5 * u8.numel() / dk / 1e9}


def allreduce_probe(env, mbytes=34, iters=10):
    """One gradient-sized all-reduce (RubiksNet-Large: 34 MB fp32) over the job's process group."""
    import torch.distributed as dist

    if not dist.is_initialized():        # N = 1: a one-rank RCCL group, so that the probe (and its code path) exists at every N
        try:
            dp.ensure_process_group(env)
        except Exception as exc:
            return {"error": repr(exc)}
    buf = torch.ones(mbytes * (1 << 20) // 4, dtype=torch.float32, device=env.device)
    for _ in range(2):
        dist.all_reduce(buf)
    dt = dp.timed_region(env, lambda: dist.all_reduce(buf), iters) / iters
    n = env.world_size
    return {"bytes": buf.numel() * 4, "ms": 1e3 * dt, "ranks": dist.get_world_size(), "backend": dist.get_backend(),
            "bus_GBps": 2 * (n - 1) / n * buf.numel() * 4 / dt / 1e9}


def self_launch(args):
    """`python bench.py --gpus N` with no torchrun environment: re-execute under torch.distributed.run,
    one rank per GPU, rendezvous on 127.0.0.1 (the container hostname may not resolve)."""
    import socket
    import subprocess

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC only on this driver (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.call(cmd, env=env)


def dry_model_leg(env, steps):
    """The model legs' control flow on a stub (no GPU): a DDP-wrapped linear model whose ranks are deliberately out of
    step, run_in()'s collective stop decision (every rank must leave after the SAME number of steps: a DDP step is a
    collective), then timed_region()'s barrier + max-over-ranks bracket."""
    import torch.distributed as dist

    torch.manual_seed(0)
    model = dp.wrap_ddp(torch.nn.Linear(16, 4), env)
    opt = torch.optim.SGD(model.parameters(), lr=1e-2)
    x, y = torch.randn(8, 16), torch.randint(0, 4, (8,))

    def step():
        time.sleep(0.002 * (1 + 2 * env.rank))                      # rank r settles at a different step time
        dp.train_step(model, opt, x, y)
    n = run_in(env, step, min_s=0.15, max_s=4.0)
    counts = [n]
    if env.distributed:
        t = torch.tensor([float(n)])
        gathered = [torch.zeros(1) for _ in range(env.world_size)]
        dist.all_gather(gathered, t)
        counts = [int(g.item()) for g in gathered]
    dt = dp.timed_region(env, step, steps)
    slowest = 0.002 * (1 + 2 * (env.world_size - 1))
    return {"run_in_steps": counts, "ms_per_step": 1e3 * dt / steps, "slowest_rank_sleep_ms": 1e3 * slowest}


def dry_run(env, args):
    """No GPU: everything of the N-rank path that is not the product kernel."""
    probe = allreduce_probe(env, mbytes=4, iters=3)
    dt = dp.timed_region(env, lambda: None, args.steps)
    stub = dry_model_leg(env, args.steps)
    if env.is_main:
        emit(json.dumps({
            "metric": "RubiksShift3D fwd+bwd GB/s vs HBM roofline", "value": None, "unit": "GB/s",
            "n_gpus": env.world_size, "steps": args.steps, "warmup": args.warmup, "ms_per_step": None,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "dry_run": True, "backend": env.backend, "rccl_ranks": env.world_size, "allreduce_probe": probe,
            "empty_timed_region_s": dt, "model_leg_stub": stub,
            "config": {"workload": "dry run on %s: launcher + rendezvous + barrier/max timing + all-reduce only "
                                   "(the operator has no CPU path)" % env.backend},
        }))


_REAL_STDOUT = None


def claim_stdout():
    """stdout carries ONE JSON line and nothing else -- but libraries write there too (RCCL prints a version banner to fd 1
    when its communicator comes up).  Keep the real stdout aside and point fd 1 at stderr for the rest of the process."""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)
        sys.stdout = sys.stderr


def emit(line):
    out = _REAL_STDOUT or sys.stdout
    out.write(line + "\n")
    out.flush()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--models", default="tiny-train,tiny-fwd-b64,small-train,large-train,large-aq-bf16-train",
                    help="comma list of model legs (%s) or 'none'" % ", ".join(MODEL_LEGS))
    ap.add_argument("--model-batch", type=int, default=32, help="clips per GPU for the train-step legs")
    ap.add_argument("--model-steps", type=int, default=6)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-legs", action="store_true", help="skip the secondary kernel legs (2-D, strided / small planes, "
                    "temporal, GEMM, BatchNorm): what tools/rocprof_stats.sh traces is then the headline operator alone")
    ap.add_argument("--settle", type=float, default=0.4,
                    help="seconds of untimed continuous load before the warm-up steps (clock / power settle)")
    ap.add_argument("--dry-run", action="store_true", help="no kernels: launcher / rendezvous / timing only (gloo)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    claim_stdout()

    dry = args.dry_run or not torch.cuda.is_available()
    env = dp.init_distributed(prefer_gpu=not dry)
    assert env.world_size == args.gpus, "--gpus %d but WORLD_SIZE=%d" % (args.gpus, env.world_size)
    if dry:
        dry_run(env, args)
        if env.distributed:
            torch.distributed.destroy_process_group()
        return
    assert env.device.type == "cuda", "bench.py needs a GPU (no CPU fallback for the product path)"

    # The CPU column first (the host legs run BEFORE the GPU legs of the same process, on rank 0 of a
    # one-GPU job only -- at N > 1 the other ranks would sit in a barrier while rank 0 computes on the host).
    cpu = cpu_model = None
    if env.is_main and env.world_size == 1 and not args.no_cpu:
        try:
            cpu = cpu_baseline()
        except Exception as exc:
            cpu = {"error": repr(exc)}
        try:
            cpu_model = cpu_model_baseline()
        except Exception as exc:
            cpu_model = {"error": repr(exc)}

    r = op_bench(env, args.steps, args.warmup, settle_s=args.settle)
    t_step = r["elapsed_s"] / args.steps
    bytes_step = r["bytes_fwd"] + r["bytes_bwd"]
    value = env.world_size * bytes_step / t_step / 1e9
    bwd_gbs = r["bytes_bwd"] / (r["bwd_ms"] * 1e-3) / 1e9
    fwd_gbs = r["bytes_fwd"] / (r["fwd_ms"] * 1e-3) / 1e9
    both_gbs = bytes_step / ((r["fwd_ms"] + r["bwd_ms"]) * 1e-3) / 1e9

    models = {}
    if args.models != "none":
        for leg in args.models.split(","):
            try:
                models[leg] = model_bench(env, leg, args.model_batch, args.model_steps, 3)
            except Exception as exc:  # the op number must still be reported
                models[leg] = {"error": repr(exc)}
            torch.cuda.empty_cache()
    if cpu_model is not None and isinstance(models.get("tiny-train"), dict):
        models["tiny-train"]["cpu_baseline"] = cpu_model
    probe = allreduce_probe(env)
    try:
        feeder = feeder_bench(env)
    except Exception as exc:
        feeder = {"error": repr(exc)}

    # Rank-0-only legs run BEFORE the final barrier: every rank then reaches destroy_process_group() together.  (Round 2
    # ran the 2-D / secondary legs on rank 0 after the last barrier while ranks 1..N-1 were already tearing the RCCL
    # communicator down -- untested on RCCL, flagged by the round-2 review.)  The other ranks idle at the barrier.
    traffic, traffic_src = pmc_traffic("backward")
    rk2d = secondary = tshift = pw16 = pw32 = bnleg = None
    if env.is_main and not args.no_legs:
        rk2d = op2d_bench(env)
        secondary = secondary_points(env)
        try:
            tshift = tshift_bench(env)
        except Exception as exc:
            tshift = {"error": repr(exc)}
        try:
            pw16 = pw16_bench(env)
        except Exception as exc:
            pw16 = {"error": repr(exc)}
        try:
            pw32 = pw32_bench(env)
        except Exception as exc:
            pw32 = {"error": repr(exc)}
        try:
            bnleg = bn_bench(env)
        except Exception as exc:
            bnleg = {"error": repr(exc)}
    dp.barrier(env)

    if env.is_main:
        def frac(d, *keys):
            for k in keys:
                d = d.get(k) if isinstance(d, dict) else None
            return round(d, 4) if isinstance(d, float) else None
        # every other kernel family against ITS roofline, flat, so that the driver's record keeps them (fraction of the 8 TB/s
        # HBM peak for the streaming kernels: fwd + bwd bytes / fwd + bwd time; of the f32 MFMA peak for the fp32 GEMMs)
        others = {
            "shift3d_stride_1_2_2_112to56": frac(secondary, "stride_1_2_2", "fwd_plus_bwd_frac_of_hbm_peak"),
            "shift3d_stride_1_2_2_28to14": frac(secondary, "stride_1_2_2_28to14", "fwd_plus_bwd_frac_of_hbm_peak"),
            "shift3d_planes_14x14": frac(secondary, "planes_14x14", "fwd_plus_bwd_frac_of_hbm_peak"),
            "shift3d_planes_14x14_288ch": frac(secondary, "planes_14x14_288ch", "fwd_plus_bwd_frac_of_hbm_peak"),
            "shift3d_planes_14x14_marginal_fwd": frac(secondary, "planes_14x14_launch_cost_fit", "fwd", "marginal_frac_of_hbm_peak"),
            "shift3d_planes_14x14_marginal_bwd": frac(secondary, "planes_14x14_launch_cost_fit", "bwd", "marginal_frac_of_hbm_peak"),
            "shift3d_planes_7x7": frac(secondary, "planes_7x7", "fwd_plus_bwd_frac_of_hbm_peak"),
            "shift3d_quantize": frac(secondary, "quantize", "fwd_plus_bwd_frac_of_hbm_peak"),
            "shift2d_f32_56x56": frac(rk2d, "f32", "fwd_plus_bwd_frac_of_hbm_peak"),
            "shift2d_bf16_56x56": frac(rk2d, "bf16", "fwd_plus_bwd_frac_of_hbm_peak"),
            "shift2d_bf16_14x14": frac(rk2d, "bf16_14x14", "fwd_plus_bwd_frac_of_hbm_peak"),
            "tshift3_bf16": frac(tshift, "bf16", "fwd_plus_bwd_frac_of_hbm_peak"),
            "tshift3_f32": frac(tshift, "f32", "fwd_plus_bwd_frac_of_hbm_peak"),
            "pw_bf16_fwd_hbm": frac(pw16, "fwd", "frac_of_hbm_peak"),
            "pw_bf16_fwd_residual_hbm": frac(pw16, "fwd_residual", "frac_of_hbm_peak"),
            "pw_bf16_wgrad_hbm": frac(pw16, "wgrad", "frac_of_hbm_peak"),
            "pw_f32_288_fwd_mfma": frac(pw32, "14x14_288ch", "fwd", "frac_of_mfma_peak"),
            "pw_f32_288_dgrad_mfma": frac(pw32, "14x14_288ch", "dgrad", "frac_of_mfma_peak"),
            "pw_f32_288_wgrad_mfma": frac(pw32, "14x14_288ch", "wgrad", "frac_of_mfma_peak"),
            "pw_f32_144_fwd_mfma": frac(pw32, "28x28_144ch", "fwd", "frac_of_mfma_peak"),
            "bn_bwd_dx_14x14_hbm": frac(bnleg, "14x14_288ch", "dx", "frac_of_hbm_peak"),
        }
        for leg, rec in models.items():
            others["model_%s_frac_of_bound" % leg] = frac(rec, "roofline", "frac")
        out = {
            "metric": "RubiksShift3D fwd+bwd GB/s vs HBM roofline",
            "value": value, "unit": "GB/s", "n_gpus": env.world_size, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * t_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": "RubiksShift3D fwd+bwd, x [N=32,T=8,C=64,H=56,W=56] fp32 per GPU "
                            "(layout [N,T,C,H,W]), shift U(-1,1) [3,64], stride 1, pad 0, "
                            "normalize_grad, 4 rotating buffer sets (forward on set k, backward on set k+2)",
                "per_gpu_batch": SHAPE[0], "global_batch": SHAPE[0] * env.world_size,
                "parallelism": "dp%d (clips sharded, no data-path collective)" % env.world_size,
                "algorithmic_bytes_per_step": bytes_step,
                "settle_s": args.settle,
            },
            "clips_per_s": env.world_size * SHAPE[0] / t_step,
            "frac_of_hbm_peak": value / env.world_size / HBM_PEAK_GBS,   # per GPU, from the wall-clock value
            "roofline": {
                "kernel": "rk::dma3d::k3d_dma_backward<2,true,1,1,true,false,false> (two 28-row bands per plane): d(x) + d(shift) + row-sum + K5 in ONE launch "
                          "(= the rk3d_backward_f32 call; the dominant kernel)",
                "bound": "hbm",
                "achieved": bwd_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": bwd_gbs / HBM_PEAK_GBS,
                "traffic": traffic, "traffic_source": traffic_src,
                "avg_launch_ms": r["bwd_ms"], "algorithmic_bytes": r["bytes_bwd"],
                "two_phase": {"what": "the same backward through rk3d_backward_partials_f32 + rk3d_backward_finalize_f32 "
                                      "(unfused kernel, then the separate row-sum + K5 kernel)",
                              "partials_avg_launch_ms": r["bwd_kernel_ms"], "finalize_avg_launch_ms": r["finalize_ms"]},
                "kernel_timing": "HIP events on the launch stream: K back-to-back launches of this kernel (rotating "
                                 "buffer sets) between one pair of events, right after the wall-clock bracket; the "
                                 "per-launch period includes the ~2 us dependent-kernel boundary (rocprofv3 duration: profiles/)",
                "forward": {"achieved": fwd_gbs, "frac": fwd_gbs / HBM_PEAK_GBS, "avg_launch_ms": r["fwd_ms"],
                            "algorithmic_bytes": r["bytes_fwd"]},
                "fwd_plus_bwd": {"achieved": both_gbs, "frac": both_gbs / HBM_PEAK_GBS,
                                 "frac_of_copy_ceiling": both_gbs / COPY_CEILING_GBS},
                "others": others,
                "others_timing": SECONDARY_TIMING,
            },
            "cpu_baseline": cpu,
            "cpu_baseline_model": cpu_model,          # the same record as models["tiny-train"]["cpu_baseline"]
            "rk2d": rk2d,
            "tshift": tshift, "pw_bf16": pw16, "pw_f32": pw32, "bn_bwd_dx": bnleg,
            "secondary": secondary,
            "model": models.get("tiny-train"),
            "models": models,
            "input_pipeline": feeder,
            "rccl_ranks": torch.distributed.get_world_size() if env.distributed else 1,
            "allreduce_probe": probe,
        }
        emit(json.dumps(out))
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
