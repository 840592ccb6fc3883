"""Generates tests/golden/attention_*.npz by IMPORTING the reference's pure-PyTorch
AttentionShift (rubiksnet/attention_shift.py) on CPU -- dev container only
(/root/reference does not exist on the GPU box; the committed .npz files travel).

    python tests/golden/gen_attention_golden.py

Each file holds inputs (x, weight, gy, n_segment) and the reference's outputs
(y, gx, gweight) in float32 and float64.
"""
import os
import sys

import numpy as np
import torch

REF = os.environ.get("RUBIKS_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    sys.path.insert(0, REF)
    from rubiksnet.attention_shift import AttentionShift  # pure torch, no extension needed

    cases = [
        # name, n, T, C, H, W, dtype
        ("a", 2, 8, 6, 5, 7, torch.float32),
        ("b", 1, 8, 16, 14, 14, torch.float32),
        ("c", 3, 4, 5, 3, 3, torch.float64),
        ("d", 2, 1, 4, 2, 2, torch.float64),   # single segment: both neighbours are padding
    ]
    for name, n, T, C, H, W, dt in cases:
        g = torch.Generator().manual_seed(ord(name) + 7)
        x = (torch.rand(n * T, C, H, W, generator=g, dtype=torch.float64) * 2 - 1).to(dt)
        gy = (torch.rand(n * T, C, H, W, generator=g, dtype=torch.float64) * 2 - 1).to(dt)
        weight = torch.rand(C, 3, generator=g, dtype=torch.float64).to(dt)
        mod = AttentionShift(T)
        mod.weight = torch.nn.Parameter(weight.clone())      # skip the lazy torch.rand init
        mod.T.data = mod.T.data.to(dt)
        xr = x.clone().requires_grad_(True)
        y = mod(xr)
        y.backward(gy)
        np.savez(
            os.path.join(HERE, "attention_%s.npz" % name),
            x=x.numpy(), gy=gy.numpy(), weight=weight.numpy(), n_segment=np.int64(T),
            y=y.detach().numpy(), gx=xr.grad.numpy(), gweight=mod.weight.grad.numpy(),
        )
        print("attention_%s: x%s -> y sum %.6f" % (name, tuple(x.shape), float(y.sum())))


if __name__ == "__main__":
    main()
