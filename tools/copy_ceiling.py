#!/usr/bin/env python
"""HBM ceilings on this box for the benchmark tensor size: torch copy (read+write) and 2-in/1-out add."""
import torch
dev = torch.device("cuda:0")
shape = (32, 8, 64, 56, 56)
sets = [(torch.empty(shape, device=dev).uniform_(-1, 1), torch.empty(shape, device=dev).uniform_(-1, 1),
         torch.empty(shape, device=dev)) for _ in range(3)]
numel = sets[0][0].numel()
def timeit(fn, iters=30):
    ev = []
    for it in range(iters):
        a, b, c = sets[it % 3]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(a, b, c); e1.record(); ev.append((e0, e1))
    torch.cuda.synchronize()
    t = sorted(e0.elapsed_time(e1) for e0, e1 in ev[3:])
    return t[len(t) // 2] * 1e-3
t = timeit(lambda a, b, c: c.copy_(a))
print("copy   (8 B/elem):  %.1f us  %.0f GB/s" % (t * 1e6, 8 * numel / t / 1e9))
t = timeit(lambda a, b, c: torch.add(a, b, out=c))
print("add    (12 B/elem): %.1f us  %.0f GB/s" % (t * 1e6, 12 * numel / t / 1e9))
t = timeit(lambda a, b, c: torch.mul(a, 2.0, out=c))
print("scale  (8 B/elem):  %.1f us  %.0f GB/s" % (t * 1e6, 8 * numel / t / 1e9))
a0 = sets[0][0].flatten()
t = timeit(lambda a, b, c: torch.dot(a.view(-1), b.view(-1)))
print("dot    (8 B/elem, read-only 2 streams): %.1f us  %.0f GB/s" % (t * 1e6, 8 * numel / t / 1e9))
t = timeit(lambda a, b, c: a.sum())
print("sum    (4 B/elem, read-only 1 stream):  %.1f us  %.0f GB/s" % (t * 1e6, 4 * numel / t / 1e9))
t = timeit(lambda a, b, c: torch.addcmul(a, a, b, out=c))
print("addcmul(12 B/elem): %.1f us  %.0f GB/s" % (t * 1e6, 12 * numel / t / 1e9))
