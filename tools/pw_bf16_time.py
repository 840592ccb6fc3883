#!/usr/bin/env python
"""Steady-state times of the bf16 1x1-convolution kernels on the Large-AQ layer shapes (batch 32 x 8 frames):
python tools/pw_bf16_time.py [shape ...]  with shape = F,K,M,H,W.  Prints us and the fraction of 8 TB/s of the
algorithmic bytes (one read per operand, one write per result)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rubiksnet_amd import _native

L = _native.lib()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
SHAPES = [(256, 288, 288, 14, 14), (256, 144, 144, 28, 28), (256, 72, 72, 56, 56), (256, 72, 144, 56, 56),
          (256, 72, 72, 112, 112), (256, 144, 288, 28, 28), (256, 288, 576, 14, 14)]
if len(sys.argv) > 1:
    SHAPES = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]


def timed(fn, sets, reps=30):
    for i in range(10):
        fn(sets[i % len(sets)])
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(reps):
        fn(sets[i % len(sets)])
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


for (Fr, K, M, H, W) in SHAPES:
    P = H * W
    sets = [dict(x=torch.randn(Fr, K, P, device=dev).bfloat16(), g=torch.randn(Fr, M, P, device=dev).bfloat16(),
                 y=torch.empty(Fr, M, P, device=dev, dtype=torch.bfloat16), o=torch.empty(Fr, K, P, device=dev, dtype=torch.bfloat16))
            for _ in range(3)]
    w = torch.randn(M, K, device=dev) / K ** 0.5
    nb = int(L.rk_pw_wgrad_workspace_bytes(Fr, K, M, P))
    ws = torch.empty(max(nb, 1), dtype=torch.uint8, device=dev)
    dw = torch.empty(M, K, device=dev)
    fwd = lambda s: _native.check(L.rk_pw_gemm_bf16(w.data_ptr(), s["x"].data_ptr(), None, s["y"].data_ptr(), Fr, K, M, P, 1, st), "f")
    fwr = lambda s: _native.check(L.rk_pw_gemm_bf16(w.data_ptr(), s["x"].data_ptr(), s["g"].data_ptr(), s["y"].data_ptr(), Fr, K, M, P, 1, st), "fr")
    dgr = lambda s: _native.check(L.rk_pw_gemm_bf16(w.data_ptr(), s["g"].data_ptr(), None, s["o"].data_ptr(), Fr, M, K, P, 0, st), "d")
    wgr = lambda s: _native.check(L.rk_pw_wgrad_bf16(s["g"].data_ptr(), s["x"].data_ptr(), dw.data_ptr(), Fr, K, M, P, ws.data_ptr(), nb, st), "w")
    ex, ey = Fr * K * P * 2, Fr * M * P * 2
    pf = torch.empty(int(L.rk_pw_packed_bytes(M, K)), dtype=torch.uint8, device=dev)
    pb = torch.empty(int(L.rk_pw_packed_bytes(K, M)), dtype=torch.uint8, device=dev)
    _native.check(L.rk_pw_pack_bf16(w.data_ptr(), M, K, pf.data_ptr(), pb.data_ptr(), st), "pack")
    pck = lambda s: _native.check(L.rk_pw_pack_bf16(w.data_ptr(), M, K, pf.data_ptr(), pb.data_ptr(), st), "pack")
    f2 = lambda s: _native.check(L.rk_pw_gemm_packed_bf16(pf.data_ptr(), s["x"].data_ptr(), None, s["y"].data_ptr(), Fr, K, M, P, st), "f2")
    f2r = lambda s: _native.check(L.rk_pw_gemm_packed_bf16(pf.data_ptr(), s["x"].data_ptr(), s["g"].data_ptr(), s["y"].data_ptr(), Fr, K, M, P, st), "f2r")
    d2 = lambda s: _native.check(L.rk_pw_gemm_packed_bf16(pb.data_ptr(), s["g"].data_ptr(), None, s["o"].data_ptr(), Fr, M, K, P, st), "d2")
    nb2 = int(L.rk_pw_wgrad16_workspace_bytes(Fr, K, M, P))
    ws2 = torch.empty(max(nb2, 1), dtype=torch.uint8, device=dev)
    w2 = lambda s: _native.check(L.rk_pw_wgrad16_bf16(s["g"].data_ptr(), s["x"].data_ptr(), dw.data_ptr(), Fr, K, M, P, ws2.data_ptr(), nb2, st), "w2")
    for name, fn, by in (("wgrad/16", w2, ex + ey), ("fwd", fwd, ex + ey), ("fwd+res", fwr, ex + 2 * ey), ("dgrad", dgr, ex + ey), ("wgrad", wgr, ex + ey),
                         ("pack", pck, 0), ("fwd/pk", f2, ex + ey), ("fwd+res/pk", f2r, ex + 2 * ey), ("dgrad/pk", d2, ex + ey)):
        us = timed(fn, sets)
        print(f"{(Fr, K, M, H, W)} {name:10s} {us:8.1f} us  {by / us / 1e6 / 8:6.3f} of 8 TB/s", flush=True)
