#!/usr/bin/env python
"""Sustained back-to-back RubiksShift3D fwd+bwd for a few seconds while sampling rocm-smi (power, sclk/mclk/fclk):
does the chip hold its clocks under a continuous HBM-bound stream?  Prints per-100-step average step time next to
the samples.  usage: python tools/sustained_probe.py [seconds]"""
import os
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rubiksnet_amd import rubiksnet_cuda  # noqa: E402

samples, stop = [], threading.Event()


def poll():
    while not stop.is_set():
        t = time.perf_counter()
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showtemp"], capture_output=True, text=True,
                                 timeout=5).stdout
            keep = [ln.split(":", 1)[1].strip() for ln in out.splitlines()
                    if ":" in ln and any(k in ln for k in ("Power (W)", "sclk", "mclk", "fclk", "junction", "socclk"))]
            samples.append((t, " | ".join(keep)))
        except Exception as e:  # noqa: BLE001
            samples.append((t, repr(e)))
        time.sleep(0.05)


def main():
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
    dev = torch.device("cuda:0")
    shape = (32, 8, 64, 56, 56)
    shift = torch.rand(3, 64, device=dev) * 2 - 1
    sets = [(torch.empty(shape, device=dev).uniform_(-1, 1), torch.empty(shape, device=dev).uniform_(-1, 1),
             torch.empty(shape, device=dev), torch.empty(shape, device=dev)) for _ in range(3)]
    gs = torch.empty(3, 64, device=dev)
    one, zero = [1, 1, 1], [0, 0, 0]
    th = threading.Thread(target=poll, daemon=True)
    th.start()
    time.sleep(0.5)
    torch.cuda.synchronize()
    t_start = time.perf_counter()
    log = []
    i = 0
    while time.perf_counter() - t_start < secs:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        block = 10 if len(log) < 60 else 100
        for _ in range(block):
            x, gy, y, gx = sets[i % 3]
            rubiksnet_cuda.rubiks_shift_3d_forward_float(x, shift, one, zero, False, y)
            x, gy, y, gx = sets[(i + 1) % 3]
            rubiksnet_cuda.rubiks_shift_3d_backward_float(x, shift, gy, one, zero, gx, gs, True, 1.0, False)
            i += 1
        e1.record()
        e1.synchronize()
        log.append((time.perf_counter() - t_start, e0.elapsed_time(e1) * 1e3 / block))   # us per step
    stop.set()
    th.join()
    print("t[s]   us/step (avg of 10 back-to-back steps for the first 600 steps, then of 100)")
    for t, us in log:
        print("%5.2f  %7.1f" % (t, us))
    print("rocm-smi samples (t relative to loop start):")
    for t, s in samples:
        print("%6.2f  %s" % (t - t_start, s))


if __name__ == "__main__":
    main()
