"""python tools/bn_dx_pre_time.py [F,C,H,W ...]: GPU-side us per launch (hipGraph replay, 4 rotating buffer sets) of the bf16
BatchNorm kernels of the -aq train step on its shapes: rk_bn_bwd_dx_pre_bf16 (in place, as fused_bn calls it, and out of
place), rk_bn_stats_finish_bf16 (statistics pass) and rk_bn_apply_affine_bf16.  RK_BN_FLAT16=0: the 4-element dx sweep."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rubiksnet_amd import _native

L = _native.lib()
dev = torch.device("cuda:0")
SHAPES = [(256, 288, 14, 14), (256, 576, 14, 14), (256, 144, 28, 28), (256, 72, 56, 56), (256, 72, 112, 112), (256, 576, 7, 7)]
if len(sys.argv) > 1:
    SHAPES = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
K, NS = 24, 4


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


for (Fr, C, H, W) in SHAPES:
    P = H * W
    xs = [torch.randn(Fr, C, P, device=dev).bfloat16() for _ in range(NS)]
    gs = [torch.randn(Fr, C, P, device=dev).bfloat16() for _ in range(NS)]
    os_ = [torch.empty_like(x) for x in xs]
    gamma, beta = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1
    sm, si = torch.randn(C, device=dev) * 0.1, torch.rand(C, device=dev) + 0.5
    k12 = torch.randn(2, C, device=dev) * 0.01
    ab = torch.randn(2, C, device=dev)
    nb = int(L.rk_bn_workspace_bytes(Fr, C, P))
    ws = torch.empty(max(nb, 1), dtype=torch.uint8, device=dev)

    def st():
        return torch.cuda.current_stream().cuda_stream

    def dx_in(i):
        g, x = gs[i % NS], xs[i % NS]
        _native.check(L.rk_bn_bwd_dx_pre_bf16(g.data_ptr(), x.data_ptr(), gamma.data_ptr(), sm.data_ptr(), si.data_ptr(),
                                              k12.data_ptr(), None, g.data_ptr(), Fr, C, P, st()), "dx")

    def dx_out(i):
        g, x, o = gs[i % NS], xs[i % NS], os_[i % NS]
        _native.check(L.rk_bn_bwd_dx_pre_bf16(g.data_ptr(), x.data_ptr(), gamma.data_ptr(), sm.data_ptr(), si.data_ptr(),
                                              k12.data_ptr(), None, o.data_ptr(), Fr, C, P, st()), "dx")

    def stats(i):
        _native.check(L.rk_bn_stats_finish_bf16(xs[i % NS].data_ptr(), gamma.data_ptr(), beta.data_ptr(), None, None, sm.data_ptr(),
                                                si.data_ptr(), ab.data_ptr(), Fr, C, P, 1e-5, 0.1, None, ws.data_ptr(), nb, st()),
                      "stats")

    def apply(i):
        _native.check(L.rk_bn_apply_affine_bf16(xs[i % NS].data_ptr(), ab[0].data_ptr(), ab[1].data_ptr(), os_[i % NS].data_ptr(),
                                                Fr, C, P, 1, st()), "apply")

    e = Fr * C * P * 2
    line = f"{(Fr, C, H, W)}"
    for name, fn, by in (("dx_pre in place", dx_in, 3 * e), ("dx_pre", dx_out, 3 * e), ("stats", stats, e), ("apply", apply, 2 * e)):
        us = graph_time(fn)
        line += f"  {name} {us:.1f} us ({by / us / 1e6 / 8:.3f})"
    print(line, flush=True)
