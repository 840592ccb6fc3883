#!/usr/bin/env python
"""Backward of the benchmark shape with a tsm-initialised shift table (every temporal shift an exact integer,
layer.py:137-141) against the ordinary U(-1,1) table: us per call, steady state."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rubiksnet_amd import rubiksnet_cuda as rc

shape = (32, 8, 64, 56, 56)
dev = "cuda:0"
sets = [(torch.empty(shape, device=dev).uniform_(-1, 1), torch.empty(shape, device=dev).uniform_(-1, 1), torch.empty(shape, device=dev)) for _ in range(3)]
g = torch.Generator().manual_seed(0)
base = (torch.rand(3, 64, generator=g) * 2 - 1)
tsm = base.clone(); tsm[0, :8] = 1.0; tsm[0, 8:16] = -1.0; tsm[0, 16:] = 0.0
gs = torch.empty(3, 64, device=dev)
for name, table in (("uniform", base), ("tsm", tsm)):
    sh = table.to(dev)
    def bwd(i):
        x, gy, gx = sets[i % 3]
        rc.rubiks_shift_3d_backward_float(x, sh, gy, [1, 1, 1], [0, 0, 0], gx, gs, True, 1.0, False)
    t_end = time.perf_counter() + 0.3
    i = 0
    while time.perf_counter() < t_end:
        for _ in range(10): bwd(i); i += 1
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(40): bwd(i + k)
    e1.record(); e1.synchronize()
    print("%s table: backward %.1f us" % (name, e0.elapsed_time(e1) / 40 * 1e3), flush=True)
