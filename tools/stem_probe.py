"""The 3x3 / stride-2 stem (backbone.py:154) under bf16 autocast on MIOpen: forward and forward + d(weight), us, batch 32 x 8 frames."""
import torch
import torch.nn.functional as F

dev = torch.device("cuda:0")


def timed(fn, n=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for C in (72, 54):
    conv = torch.nn.Conv2d(3, C, 3, stride=2, padding=1, bias=False).to(dev)
    x = torch.randn(256, 3, 224, 224, device=dev)
    gy = torch.randn(256, C, 112, 112, device=dev).bfloat16()

    def fwd():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            return conv(x)

    def both():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = conv(x)
        y.backward(gy)
        conv.weight.grad = None

    print(f"stem 3->{C} [256,3,224,224] bf16 autocast: fwd {timed(fwd):.0f} us, fwd + d(weight) {timed(both):.0f} us "
          f"(output {256 * C * 112 * 112 * 2 / 1e6:.0f} MB = {256 * C * 112 * 112 * 2 / 8e6:.0f} us at 8 TB/s)", flush=True)

# the HIP stem (rk_stem16.hip) through pointwise.stem_conv
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rubiksnet_amd import pointwise
for C in (72, 54):
    conv = torch.nn.Conv2d(3, C, 3, stride=2, padding=1, bias=False).to(dev)
    x = torch.randn(256, 3, 224, 224, device=dev)
    gy = torch.randn(256, C, 112, 112, device=dev).bfloat16()

    def fwd():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            return pointwise.stem_conv(conv, x)

    def both():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = pointwise.stem_conv(conv, x)
        y.backward(gy)
        conv.weight.grad = None

    print(f"HIP stem 3->{C}: fwd {timed(fwd):.0f} us, fwd + d(weight) {timed(both):.0f} us", flush=True)
