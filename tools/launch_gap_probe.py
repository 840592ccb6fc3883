#!/usr/bin/env python
"""How much of bench.py's per-step wall time is launch gaps: same fwd+bwd loop as bench.op_bench timed
(a) with the three events per step bench.py records, (b) bare, (c) captured in one HIP graph."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rubiksnet_amd import rubiksnet_cuda

dev = torch.device("cuda:0")
SHAPE = (32, 8, 64, 56, 56)
shift = (torch.rand(3, 64) * 2 - 1).to(dev)
sets = []
for _ in range(3):
    x = torch.empty(SHAPE, device=dev).uniform_(-1, 1)
    gy = torch.empty(SHAPE, device=dev).uniform_(-1, 1)
    sets.append((x, gy, torch.empty_like(x), torch.empty_like(x)))
gshift = torch.empty(3, 64, device=dev)
s1, p0 = [1, 1, 1], [0, 0, 0]

def step(i, ev=None):
    x, _, y, _ = sets[i % 3]
    xb, gy, _, gx = sets[(i + 1) % 3]
    if ev: ev[0].record()
    rubiksnet_cuda.rubiks_shift_3d_forward_float(x, shift, s1, p0, False, y)
    if ev: ev[1].record()
    rubiksnet_cuda.rubiks_shift_3d_backward_float(xb, shift, gy, s1, p0, gx, gshift, True, 1.0, False)
    if ev: ev[2].record()

def wall(fn, K=50):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    fn(K)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / K * 1e6

for i in range(10): step(i)
def with_events(K):
    for i in range(K): step(i, [torch.cuda.Event(enable_timing=True) for _ in range(3)])
def bare(K):
    for i in range(K): step(i)
print("events: %.1f us/step" % wall(with_events))
print("bare  : %.1f us/step" % wall(bare))
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for i in range(3): step(i)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        for i in range(48): step(i)
def graphed(K):
    g.replay()
for _ in range(2): g.replay()
torch.cuda.synchronize(); t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize()
print("graph : %.1f us/step" % ((time.perf_counter() - t0) / 48 * 1e6))
