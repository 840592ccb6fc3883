"""python tools/fixed_cost_probe2d.py [dtype=bf16] [H=14] [Cs...]: the 2-D operator's per-launch time on [256, C, H, H] for several
C (hipGraph replay) and the line t = a + b C -- launch cost and marginal streaming rate, as tools/fixed_cost_probe.py does for
the 3-D operator.  The shift table is fp32 (what the -aq networks run under bf16 autocast)."""
import sys

import numpy as np
import torch

from rubiksnet_amd import rubiksnet_cuda

dt = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[sys.argv[1] if len(sys.argv) > 1 else "bf16"]
H = int(sys.argv[2]) if len(sys.argv) > 2 else 14
Cs = [int(v) for v in sys.argv[3:]] or [72, 144, 288, 432, 576]
F, K = 256, 30
dev = torch.device("cuda:0")


def graph_time(fn):
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for i in range(6):
            fn(i)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for i in range(K):
                fn(i)
    best = 1e9
    for _ in range(5):
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            g.replay()
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / (10 * K) * 1e3)
    return best


rows = []
for C in Cs:
    torch.manual_seed(0)
    shift = torch.rand(2, C, device=dev) * 1.9 - 0.95
    sets = [(torch.empty(F, C, H, H, device=dev, dtype=dt).uniform_(-1, 1), torch.empty(F, C, H, H, device=dev, dtype=dt).uniform_(-1, 1),
             torch.zeros(F, C, H, H, device=dev, dtype=dt), torch.zeros(F, C, H, H, device=dev, dtype=dt)) for _ in range(3)]
    gs = torch.empty_like(shift)
    f = lambda i: rubiksnet_cuda.rubiks2d_forward(sets[i % 3][0], shift, [1, 1], [0, 0], False, sets[i % 3][2])
    b = lambda i: rubiksnet_cuda.rubiks2d_backward(sets[i % 3][1], sets[i % 3][0], shift, [1, 1], [0, 0], True, True, False, sets[i % 3][3], gs)
    tf, tb = graph_time(f), graph_time(b)
    rows.append((C, tf, tb))
    print(f"C={C:4d}: fwd {tf:6.2f}  bwd {tb:6.2f} us", flush=True)
    del sets
    torch.cuda.empty_cache()
A = np.array([[1.0, r[0]] for r in rows])
es = torch.empty(0, dtype=dt).element_size()
for name, col, passes in (("fwd", 1, 2), ("bwd", 2, 3)):
    y = np.array([r[col] for r in rows])
    (a, bb), *_ = np.linalg.lstsq(A, y, rcond=None)
    per_c = passes * es * F * H * H
    print(f"{name:4s} t = {a:5.2f} us + {bb * 1e3:6.2f} ns/channel -> marginal {per_c / bb / 1e3:6.0f} GB/s, residuals {np.round(y - A @ [a, bb], 2)}")
