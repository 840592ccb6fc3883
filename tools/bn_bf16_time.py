#!/usr/bin/env python
"""Steady-state times of the bf16 BatchNorm+ReLU kernels on the Large-AQ shapes: python tools/bn_bf16_time.py [F,C,H,W ...]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rubiksnet_amd import _native
L = _native.lib(); dev = torch.device("cuda:0"); st = torch.cuda.current_stream().cuda_stream
SHAPES = [(256, 288, 14, 14), (256, 144, 28, 28), (256, 72, 56, 56), (256, 72, 112, 112)]
if len(sys.argv) > 1:
    SHAPES = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
def timed(fn, n=3, reps=30):
    for i in range(10): fn(i % n)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(reps): fn(i % n)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps
for (Fr, C, H, W) in SHAPES:
    P = H * W
    xs = [torch.randn(Fr, C, P, device=dev).bfloat16() for _ in range(3)]
    gs = [torch.randn(Fr, C, P, device=dev).bfloat16() for _ in range(3)]
    ys = [torch.empty_like(x) for x in xs]
    gamma, beta = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    sm, si = torch.empty(C, device=dev), torch.empty(C, device=dev)
    dg, db = torch.empty(C, device=dev), torch.empty(C, device=dev)
    nb = int(L.rk_bn_workspace_bytes(Fr, C, P)); ws = torch.empty(max(nb, 1), dtype=torch.uint8, device=dev)
    fwd = lambda i: _native.check(L.rk_bn_relu_forward_bf16(xs[i].data_ptr(), gamma.data_ptr(), beta.data_ptr(), rm.data_ptr(), rv.data_ptr(), sm.data_ptr(), si.data_ptr(), ys[i].data_ptr(), Fr, C, P, 1e-5, 0.1, 1, 1, ws.data_ptr(), nb, st), "f")
    bwd = lambda i: _native.check(L.rk_bn_relu_backward_bf16(gs[i].data_ptr(), xs[i].data_ptr(), gamma.data_ptr(), beta.data_ptr(), sm.data_ptr(), si.data_ptr(), None, ys[i].data_ptr(), dg.data_ptr(), db.data_ptr(), Fr, C, P, 1, ws.data_ptr(), nb, st), "b")
    e = Fr * C * P * 2
    for name, fn, by in (("fwd (stats+apply)", fwd, 3 * e), ("bwd (reduce+dx)", bwd, 5 * e)):
        us = timed(fn)
        print(f"{(Fr, C, H, W)} {name:18s} {us:8.1f} us  {by / us / 1e6 / 8:6.3f} of 8 TB/s", flush=True)
