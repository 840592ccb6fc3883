#!/usr/bin/env python
"""Minimal driver for profiling: N iterations of RubiksShift3D fwd + bwd on the benchmark shape
(or --shape N,T,C,H,W) through the C ABI.  Prints per-iteration event timings."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rubiksnet_amd import rubiksnet_cuda  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="32,8,64,56,56")
    ap.add_argument("--iters", type=int, default=12)
    ap.add_argument("--sets", type=int, default=3)
    ap.add_argument("--stride", default="1,1,1")
    ap.add_argument("--bwd", default="fused", choices=["fused", "gx", "gs"])
    ap.add_argument("--coupled", action="store_true", help="bwd on the SAME buffer set as the fwd just run (x may still be in the Infinity Cache)")
    args = ap.parse_args()
    shape = tuple(int(v) for v in args.shape.split(","))
    stride = [int(v) for v in args.stride.split(",")]
    N, T, C, H, W = shape
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    shift = (torch.rand(3, C, device=dev) * 2 - 1)
    so = (N, T, C, (H - 1) // stride[1] + 1, (W - 1) // stride[2] + 1)
    sets = [(torch.empty(shape, device=dev).uniform_(-1, 1), torch.empty(so, device=dev).uniform_(-1, 1),
             torch.empty(so, device=dev), torch.empty(shape, device=dev)) for _ in range(args.sets)]
    gs = torch.empty(3, C, device=dev)
    p0 = [0, 0, 0]
    ev = []
    for it in range(args.iters):
        x, gy, y, gx = sets[it % args.sets]
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        rubiksnet_cuda.rubiks_shift_3d_forward_float(x, shift, stride, p0, False, y)
        e[1].record()
        if not args.coupled:   # bwd on a set whose x was last touched two launches (>= 0.8 GB of traffic) ago
            x, gy, y, gx = sets[(it + 1) % args.sets]
        rubiksnet_cuda.rubiks_shift_3d_backward_float(x, shift, gy, stride, p0, None if args.bwd == "gs" else gx,
                                                      None if args.bwd == "gx" else gs, True, 1.0, False)
        e[2].record()
        ev.append(e)
    torch.cuda.synchronize()
    f = sorted(e[0].elapsed_time(e[1]) for e in ev[2:])
    b = sorted(e[1].elapsed_time(e[2]) for e in ev[2:])
    numel = N * T * C * H * W
    print("fwd median %.1f us  bwd median %.1f us  sum %.1f (fwd %.0f GB/s, bwd %.0f GB/s algorithmic)" % (
        1e3 * f[len(f) // 2], 1e3 * b[len(b) // 2], 1e3 * (f[len(f) // 2] + b[len(b) // 2]), 8 * numel / f[len(f) // 2] / 1e6, 12 * numel / b[len(b) // 2] / 1e6))
    sys.stdout.flush()
    os._exit(0) if os.environ.get("RK_FAST_EXIT") else None


if __name__ == "__main__":
    main()
