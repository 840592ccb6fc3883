"""python tools/pw16_odd_time.py: GPU-side us per launch (hipGraph replay) of the bf16 odd-plane 1x1 kernels (rk_pw16_odd.hip) on
Large-AQ's layer4 shapes at batch 32 (256 frames of 7x7): forward, forward + residual, d(weight)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rubiksnet_amd import _native, pointwise

L = _native.lib()
dev = torch.device("cuda:0")
NS, KR = 4, 24


def graph_time(fn):
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for i in range(2 * NS):
            fn(i)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for i in range(KR):
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
        best = min(best, e0.elapsed_time(e1) / (8 * KR) * 1e3)
    return best


def st():
    return torch.cuda.current_stream().cuda_stream


for (Fr, K, M, P) in ((256, 576, 576, 49), (256, 288, 576, 49), (256, 576, 288, 49), (256, 1152, 1152, 49)):
    w = torch.randn(M, K, device=dev) * 0.05
    fwd, _ = pointwise._pack(w)
    xs = [torch.randn(Fr, K, P, device=dev).bfloat16() for _ in range(NS)]
    ys = [torch.randn(Fr, M, P, device=dev).bfloat16() for _ in range(NS)]
    rs = [torch.randn(Fr, M, P, device=dev).bfloat16() for _ in range(NS)]
    dw = torch.empty(M, K, device=dev)
    nb = int(L.rk_pw_wgrad_odd16_workspace_bytes(Fr, K, M, P))
    ws = torch.empty(max(nb, 1), dtype=torch.uint8, device=dev)

    def f(i):
        _native.check(L.rk_pw_gemm_packed_odd_bf16(fwd.data_ptr(), xs[i % NS].data_ptr(), None, ys[i % NS].data_ptr(), Fr, K, M, P, st()), "f")

    def fr(i):
        _native.check(L.rk_pw_gemm_packed_odd_bf16(fwd.data_ptr(), xs[i % NS].data_ptr(), rs[i % NS].data_ptr(), ys[i % NS].data_ptr(), Fr, K,
                                                   M, P, st()), "fr")

    def wg(i):
        _native.check(L.rk_pw_wgrad_odd16_bf16(ys[i % NS].data_ptr(), xs[i % NS].data_ptr(), dw.data_ptr(), Fr, K, M, P, ws.data_ptr(), nb,
                                               st()), "wg")

    by = Fr * (K + M) * P * 2
    print(f"[{Fr},{K}->{M},{P}] fwd {graph_time(f):.1f} us  fwd+res {graph_time(fr):.1f} us  wgrad {graph_time(wg):.1f} us  "
          f"(operands {by / 1e6:.1f} MB = {by / 8e6:.1f} us at 8 TB/s; {2 * Fr * K * M * P / 2.5e9:.1f} us of bf16 MFMA)", flush=True)
