"""python tools/tshift_bn_time.py [C,H ...]: GPU-side us per launch (hipGraph replay, 4 rotating buffer sets) of the bf16
temporal 3-tap kernels on Large-AQ's layer shapes at batch 32 (NT = 256, n_segment 8): plain forward / backward and the
bn1 + ReLU fused forms (rk_tshift3_bn_*).  Fractions of 8 TB/s over 2 / 3 tensor passes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rubiksnet_amd import _native

L = _native.lib()
dev = torch.device("cuda:0")
SHAPES = [(72, 112), (72, 56), (144, 28), (288, 14), (576, 7)]
if len(sys.argv) > 1:
    SHAPES = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
NS, K = 4, 24


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


def st():
    return torch.cuda.current_stream().cuda_stream


for C, H in SHAPES:
    NT, S, HW = 256, 8, H * H
    sets = [(torch.randn(NT, C, H, H, device=dev).bfloat16(), torch.randn(NT, C, H, H, device=dev).bfloat16(),
             torch.empty(NT, C, H, H, device=dev, dtype=torch.bfloat16)) for _ in range(NS)]
    taps = torch.softmax(torch.randn(C, 3, device=dev), 1).contiguous()
    gtaps = torch.empty_like(taps)
    ab = torch.stack((torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.2)).contiguous()
    sm, si = torch.randn(C, device=dev) * 0.1, torch.rand(C, device=dev) + 0.5
    bred = torch.empty(C, NT // S, 2, device=dev)
    wsb = int(L.rk_tshift3_backward_workspace_bytes(NT, S, C, HW))
    ws = torch.empty(max(wsb, 1), dtype=torch.uint8, device=dev)

    def fwd(i):
        x, _, y = sets[i % NS]
        _native.check(L.rk_tshift3_forward_bf16(x.data_ptr(), taps.data_ptr(), y.data_ptr(), NT, S, C, HW, st()), "f")

    def bwd(i):
        x, g, y = sets[i % NS]
        _native.check(L.rk_tshift3_backward_bf16(g.data_ptr(), x.data_ptr(), taps.data_ptr(), y.data_ptr(), gtaps.data_ptr(), NT, S, C,
                                                 HW, ws.data_ptr(), wsb, st()), "b")

    def fbn(i):
        x, _, y = sets[i % NS]
        _native.check(L.rk_tshift3_bn_forward_bf16(x.data_ptr(), taps.data_ptr(), ab.data_ptr(), y.data_ptr(), NT, S, C, HW, st()), "fbn")

    def bbn(i):
        x, g, y = sets[i % NS]
        _native.check(L.rk_tshift3_bn_backward_bf16(g.data_ptr(), x.data_ptr(), taps.data_ptr(), ab.data_ptr(), sm.data_ptr(),
                                                    si.data_ptr(), y.data_ptr(), gtaps.data_ptr(), bred.data_ptr(), NT, S, C, HW,
                                                    ws.data_ptr(), wsb, st()), "bbn")

    nb = NT * C * HW * 2
    line = f"[{NT},{C},{H},{H}] bf16:"
    for name, fn, by in (("fwd", fwd, 2 * nb), ("bwd", bwd, 3 * nb), ("fwd_bn", fbn, 2 * nb), ("bwd_bn", bbn, 3 * nb)):
        us = graph_time(fn)
        line += f"  {name} {us:.1f} us ({by / us / 1e6 / 8:.3f})"
    print(line, flush=True)
