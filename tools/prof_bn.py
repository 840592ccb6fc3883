#!/usr/bin/env python
"""Timing of the fused BatchNorm2d+ReLU operator (train forward, backward, eval forward) through
rubiksnet_amd.fused_bn on the activation shapes of the Tiny net at 32 clips x 8 frames.
Algorithmic bytes: 12 B/elem train forward (two reads + a write), 20 B/elem backward, 8 B/elem eval (fp32)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rubiksnet_amd.fused_bn import bn_relu

dev = torch.device("cuda:0")
torch.manual_seed(0)
def timeit(fn, iters=16):
    ev = []
    for it in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(it); b.record(); ev.append((a, b))
    torch.cuda.synchronize()
    t = sorted(x.elapsed_time(y) for x, y in ev[3:])
    return t[len(t) // 2] * 1e-3

def run(shape, dtype):
    C = shape[1]
    xs = [torch.randn(shape, device=dev, dtype=dtype).requires_grad_(True) for _ in range(3)]
    gy = [torch.randn(shape, device=dev, dtype=dtype) for _ in range(3)]
    bn = torch.nn.BatchNorm2d(C).to(dev)
    n, es = xs[0].numel(), xs[0].element_size()
    ys = [None] * 3
    def fwd(i):
        ys[i % 3] = bn_relu(bn, xs[i % 3])
    def bwd(i):
        ys[i % 3].backward(gy[i % 3], retain_graph=True); xs[i % 3].grad = None
    bn.train()
    tf = timeit(fwd)
    for i in range(3): fwd(i)
    tb = timeit(bwd)
    bn.eval()
    with torch.no_grad():
        te = timeit(lambda i: bn_relu(bn, xs[i % 3]))
    print("bn_relu %-20s %-8s train fwd %7.1f us %5.0f GB/s | bwd %7.1f us %5.0f GB/s | eval %7.1f us %5.0f GB/s" % (
        shape, str(dtype).split(".")[1], tf * 1e6, 3 * es * n / tf / 1e9, tb * 1e6, 5 * es * n / tb / 1e9,
        te * 1e6, 2 * es * n / te / 1e9))

if len(sys.argv) > 1:
    run(tuple(int(v) for v in sys.argv[1:5]), getattr(torch, sys.argv[5]) if len(sys.argv) > 5 else torch.float32)
    sys.exit(0)
for shape in [(256, 54, 56, 56), (256, 108, 28, 28), (256, 216, 14, 14), (256, 432, 7, 7), (256, 24, 112, 112)]:
    for dt in (torch.float32, torch.bfloat16):
        run(shape, dt)
