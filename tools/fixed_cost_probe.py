"""python tools/fixed_cost_probe.py H [Cs...]: per-launch time of the 3-D backward (fused one-launch form, and the producers alone
= the two-phase partials entry) and of the forward on [32, 8, C, H, H] for several C, hipGraph replay, and the least-squares
line t = a + b C: `a` is what a launch costs before it streams a byte (graph-node boundary + ramp + finalizer tail), b the
marginal streaming rate."""
import ctypes
import sys

import numpy as np
import torch

from rubiksnet_amd import _native, rubiksnet_cuda

H = int(sys.argv[1]) if len(sys.argv) > 1 else 14
Cs = [int(v) for v in sys.argv[2:]] or [72, 144, 216, 288, 432, 576]
N, T = 32, 8
dev = torch.device("cuda:0")
L = _native.lib()
K = 30


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
    shift = torch.rand(3, C, device=dev) * 2 - 1
    sets = [(torch.randn(N, T, C, H, H, device=dev), torch.randn(N, T, C, H, H, device=dev), torch.empty(N, T, C, H, H, device=dev))
            for _ in range(3)]
    gs = torch.empty(3, C, device=dev)
    nb = int(L.rk3d_backward_workspace_bytes(N, T, C, H, H, 1, 1, 1, 0, 0, 0, 4))
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    pc = ctypes.c_int(0)

    def fwd(i):
        x, gy, o = sets[i % 3]
        rubiksnet_cuda.rubiks_shift_3d_forward_float(x, shift, [1, 1, 1], [0, 0, 0], False, o)

    def bwd(i):
        x, gy, o = sets[i % 3]
        rubiksnet_cuda.rubiks_shift_3d_backward_float(x, shift, gy, [1, 1, 1], [0, 0, 0], o, gs, True, 1.0, False)

    def part(i):
        x, gy, o = sets[i % 3]
        _native.check(L.rk3d_backward_partials_f32(x.data_ptr(), shift.data_ptr(), gy.data_ptr(), o.data_ptr(), N, T, C, H, H, 1, 1, 1,
                                                   0, 0, 0, 0, ws.data_ptr(), nb, ctypes.byref(pc),
                                                   torch.cuda.current_stream().cuda_stream), "partials")

    f, b, p = graph_time(fwd), graph_time(bwd), graph_time(part)
    rows.append((C, f, b, p))
    print(f"C={C:4d}: fwd {f:6.2f}  bwd fused {b:6.2f}  producers alone {p:6.2f} us", flush=True)
    del sets
    torch.cuda.empty_cache()

# the floor of a graph node: a launch that does nothing measurable
tiny = torch.zeros(64, device=dev)
null = graph_time(lambda i: tiny.add_(1.0))
print(f"graph-node floor (64-element add_): {null:.2f} us")
A = np.array([[1.0, r[0]] for r in rows])
for name, col in (("fwd", 1), ("bwd fused", 2), ("producers alone", 3)):
    y = np.array([r[col] for r in rows])
    (a, bb), *_ = np.linalg.lstsq(A, y, rcond=None)
    elems = N * T * H * H
    per_c = (8 if col == 1 else 12) * elems
    print(f"{name:16s} t = {a:5.2f} us + {bb * 1e3:6.2f} ns/channel  -> marginal {per_c / bb / 1e3:6.0f} GB/s, residuals {np.round(y - A @ [a, bb], 2)}")
