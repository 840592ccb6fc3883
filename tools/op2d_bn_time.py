"""python tools/op2d_bn_time.py F C H W [stride] [dtype]: GPU-side us per launch (hipGraph replay, 4 rotating buffer sets) of
RubiksShift2D forward / backward and of their bn2 + ReLU fused forms (rk2d_*_bn_*), the latter with the bn kernels around them
listed separately: what the fused pair replaces is apply + forward, and backward + reduce + dx."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rubiksnet_amd import _native, rubiksnet_cuda

F, C, H, W = (int(v) for v in sys.argv[1:5])
s = int(sys.argv[5]) if len(sys.argv) > 5 else 1
dt = {"bf16": torch.bfloat16, "f32": torch.float32}[sys.argv[6] if len(sys.argv) > 6 else "bf16"]
sfx = "bf16_sf32" if dt == torch.bfloat16 else "f32"
L = _native.lib()
torch.manual_seed(0)
dev = torch.device("cuda:0")
Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
shift = torch.rand(2, C, device=dev) * 1.9 - 0.95
shift[(shift - shift.round()).abs() < 1e-3] = 0.37
NS, K = 4, 24
sets = [(torch.empty(F, C, H, W, device=dev, dtype=dt).uniform_(-1, 1), torch.empty(F, C, Ho, Wo, device=dev, dtype=dt).uniform_(-1, 1),
         torch.empty(F, C, Ho, Wo, device=dev, dtype=dt), torch.empty(F, C, H, W, device=dev, dtype=dt)) for _ in range(NS)]
gs = torch.empty_like(shift)
ab = torch.stack((torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.2))
abmi = torch.stack((ab[0], ab[1], torch.randn(C, device=dev) * 0.1, torch.rand(C, device=dev) + 0.5), dim=1).contiguous()
k12 = torch.empty(2, C, device=dev)
dg, db = torch.empty(C, device=dev), torch.empty(C, device=dev)
nb = int(L.rk2d_backward_bn_workspace_bytes(F, C, H, W, s, s, 0, 0))
ws = torch.empty(max(nb, 1), dtype=torch.uint8, device=dev)
shift_t = shift.to(dt) if dt != torch.bfloat16 else shift


def st():
    return torch.cuda.current_stream().cuda_stream


def f(i):
    rubiksnet_cuda.rubiks2d_forward(sets[i % NS][0], shift_t, [s, s], [0, 0], False, sets[i % NS][2])


def b(i):
    rubiksnet_cuda.rubiks2d_backward(sets[i % NS][1], sets[i % NS][0], shift_t, [s, s], [0, 0], True, True, False, sets[i % NS][3], gs)


def fbn(i):
    x, gy, y, gx = sets[i % NS]
    _native.check(getattr(L, "rk2d_forward_bn_" + sfx)(x.data_ptr(), ab.data_ptr(), shift.data_ptr(), y.data_ptr(), F, C, H, W, s, s,
                                                       0, 0, 0, st()), "fbn")


def bbn(i):
    x, gy, y, gx = sets[i % NS]
    _native.check(getattr(L, "rk2d_backward_bn_" + sfx)(gy.data_ptr(), x.data_ptr(), abmi.data_ptr(), shift.data_ptr(), gx.data_ptr(),
                                                        gs.data_ptr(), k12.data_ptr(), dg.data_ptr(), db.data_ptr(), F, C, H, W,
                                                        s, s, 0, 0, 1, 0, ws.data_ptr(), nb, st()), "bbn")


def graph_time(fn):
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for i in range(2 * NS):
            fn(i)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for i in range(K):
                fn(i)
    best = 1e9
    for _ in range(5):
        for _ in range(2):
            g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(8):
            g.replay()
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / (8 * K) * 1e3)
    return best


es = sets[0][0].element_size()
nx, ny = F * C * H * W * es, F * C * Ho * Wo * es
line = f"[{F},{C},{H},{W}] s{s} {dt}:"
for name, fn, by in (("fwd", f, nx + ny), ("bwd", b, 2 * nx + ny), ("fwd_bn", fbn, nx + ny), ("bwd_bn", bbn, 2 * nx + ny)):
    try:
        us = graph_time(fn)
        line += f"  {name} {us:.1f} us ({by / us / 1e6 / 8:.3f})"
    except Exception as exc:  # a configuration without a fused kernel
        line += f"  {name} n/a ({type(exc).__name__})"
print(line, flush=True)
