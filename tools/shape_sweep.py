#!/usr/bin/env python
"""RubiksShift3D fwd / bwd through the C ABI on several shapes with the SAME element count as the
benchmark shape: which (n, c) columns are resident at once is what changes.  One process, HIP-event
medians per kernel.  usage: python tools/shape_sweep.py [N,T,C,H,W ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rubiksnet_amd import rubiksnet_cuda  # noqa: E402

DEFAULT = ["32,8,64,56,56", "4,8,512,56,56", "2,8,1024,56,56", "256,1,64,56,56", "128,2,64,56,56", "64,4,64,56,56",
           "16,16,64,56,56", "8,32,64,56,56", "128,8,16,56,56", "32,8,16,112,112", "32,8,256,28,28"]


def run(shape, iters=14, sets=3):
    N, T, C, H, W = shape
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    shift = torch.rand(3, C, device=dev) * 2 - 1
    bufs = [(torch.empty(shape, device=dev).uniform_(-1, 1), torch.empty(shape, device=dev).uniform_(-1, 1),
             torch.empty(shape, device=dev), torch.empty(shape, device=dev)) for _ in range(sets)]
    gs = torch.empty(3, C, device=dev)
    one, zero = [1, 1, 1], [0, 0, 0]
    ev = []
    for it in range(iters):
        x, gy, y, gx = bufs[it % sets]
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        rubiksnet_cuda.rubiks_shift_3d_forward_float(x, shift, one, zero, False, y)
        e[1].record()
        x, gy, y, gx = bufs[(it + 1) % sets]
        rubiksnet_cuda.rubiks_shift_3d_backward_float(x, shift, gy, one, zero, gx, gs, True, 1.0, False)
        e[2].record()
        ev.append(e)
    torch.cuda.synchronize()
    f = sorted(e[0].elapsed_time(e[1]) for e in ev[2:])
    b = sorted(e[1].elapsed_time(e[2]) for e in ev[2:])
    numel = N * T * C * H * W
    fm, bm = f[len(f) // 2], b[len(b) // 2]
    print("%-18s fwd %7.1f us %5.2f TB/s | bwd(+finalize) %7.1f us %5.2f TB/s" % (
        ",".join(map(str, shape)), 1e3 * fm, 8 * numel / fm / 1e9, 1e3 * bm, 12 * numel / bm / 1e9), flush=True)
    del bufs
    torch.cuda.empty_cache()


if __name__ == "__main__":
    for s in (sys.argv[1:] or DEFAULT):
        run(tuple(int(v) for v in s.split(",")))
