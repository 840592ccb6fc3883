#!/usr/bin/env python
"""Timing of the 2D shift operator and the temporal 3-tap kernel through the
product's functional layer; algorithmic GB/s = 8 B/elem forward, 12 B/elem backward (fp32; halves for bf16)."""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rubiksnet_amd import rubiksnet_cuda, _native
from rubiksnet_amd.attention_shift import temporal_shift3

dev = torch.device("cuda:0")
torch.manual_seed(0)
N_INTEGER = int(os.environ.get("PROF_INTEGER_CHANNELS", "0"))   # channels given an exactly-integer shift
def timeit(fn, iters=30):
    ev = []
    for it in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(it); e1.record(); ev.append((e0, e1))
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev[3:])
    return t[len(t) // 2] * 1e-3

def run(shape, dtype, stride=1):
    NT, C, H, W = shape
    es = torch.empty((), dtype=dtype).element_size()
    sets = []
    for _ in range(4):
        x = torch.empty(shape, device=dev, dtype=dtype).uniform_(-1, 1)
        Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
        gy = torch.empty((NT, C, Ho, Wo), device=dev, dtype=dtype).uniform_(-1, 1)
        sets.append((x, gy, torch.empty_like(gy), torch.empty_like(x)))
    shift = (torch.rand(2, C, device=dev) * 1.9 - 0.95).to(dtype)
    shift[(shift - shift.round()).abs() < 1e-3] = 0.37
    shift[:, :N_INTEGER] = 1.0
    gs = torch.empty_like(shift)
    nin, nout = sets[0][0].numel(), sets[0][1].numel()
    tf = timeit(lambda i: rubiksnet_cuda.rubiks2d_forward(sets[i % 4][0], shift, [stride] * 2, [0, 0], False, sets[i % 4][2]))
    tb = timeit(lambda i: rubiksnet_cuda.rubiks2d_backward(sets[i % 4][1], sets[i % 4][0], shift, [stride] * 2, [0, 0], True, True, False, sets[i % 4][3], gs))
    print("rk2d %-22s %-8s s%d: fwd %7.1f us %5.0f GB/s | bwd %7.1f us %5.0f GB/s" % (
        shape, str(dtype).split(".")[1], stride, tf * 1e6, es * (nin + nout) / tf / 1e9, tb * 1e6, es * (nout + 2 * nin) / tb / 1e9))
    if stride == 1:
        taps = torch.softmax(torch.rand(C, 3, device=dev), 1)
        xs = [s[0].clone().requires_grad_(True) for s in sets]
        tp = taps.clone().requires_grad_(True)
        tf = timeit(lambda i: temporal_shift3(sets[i % 4][0], taps, 8))
        def fb(i):
            y = temporal_shift3(xs[i % 4], tp, 8); y.backward(sets[i % 4][1])
        tfb = timeit(fb)
        print("tshift3 %-19s %-8s   : fwd %7.1f us %5.0f GB/s | fwd+bwd %7.1f us %5.0f GB/s (8+12 B/elem fp32-equivalent)" % (
            shape, str(dtype).split(".")[1], tf * 1e6, 2 * es * nin / tf / 1e9, tfb * 1e6, 5 * es * nin / tfb / 1e9))

if len(sys.argv) > 1:      # python tools/prof_2d.py N C H W [dtype]
    run(tuple(int(v) for v in sys.argv[1:5]), getattr(torch, sys.argv[5]) if len(sys.argv) > 5 else torch.float32)
    sys.exit(0)
for shape in [(256, 64, 56, 56), (256, 288, 14, 14), (256, 54, 112, 112)]:
    for dt in (torch.float32, torch.bfloat16):
        run(shape, dt)
run((256, 108, 56, 56), torch.float32, stride=2)
