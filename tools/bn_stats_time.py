#!/usr/bin/env python
"""rk_bn_stats_finish_bf16 alone (statistics + finisher, one launch): us per call.  python tools/bn_stats_time.py [F,C,H,W ...]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rubiksnet_amd import _native
L = _native.lib(); dev = torch.device("cuda:0"); st = torch.cuda.current_stream().cuda_stream
SHAPES = [(256, 288, 14, 14), (256, 576, 14, 14), (256, 144, 28, 28), (256, 72, 56, 56)]
if len(sys.argv) > 1:
    SHAPES = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
tag = " ".join(f"{k}={v}" for k, v in sorted(os.environ.items()) if k.startswith("RK_BN"))
for (Fr, C, H, W) in SHAPES:
    P = H * W
    xs = [torch.randn(Fr, C, P, device=dev).bfloat16() for _ in range(4)]
    gamma, beta = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    sm, si, ab = torch.empty(C, device=dev), torch.empty(C, device=dev), torch.empty(2, C, device=dev)
    nb = int(L.rk_bn_workspace_bytes(Fr, C, P)); ws = torch.empty(max(nb, 1), dtype=torch.uint8, device=dev)
    def run(i):
        _native.check(L.rk_bn_stats_finish_bf16(xs[i % 4].data_ptr(), gamma.data_ptr(), beta.data_ptr(), rm.data_ptr(), rv.data_ptr(), sm.data_ptr(),
                                                si.data_ptr(), ab.data_ptr(), Fr, C, P, 1e-5, 0.1, None, ws.data_ptr(), nb, st), "s")
    for i in range(10): run(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(50): run(i)
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) * 1e3 / 50
    print(f"[{tag}] {(Fr, C, H, W)} stats+finish {us:7.1f} us  {Fr * C * P * 2 / us / 1e6 / 8:6.3f} of 8 TB/s", flush=True)
