#!/usr/bin/env python
"""d(weight) of the bf16 1x1 convolutions alone (rk_pw_wgrad16_bf16 = split kernel + partial-sum kernel): steady-state us per
call on the Large-AQ layer shapes, checked against torch's fp32 matmul of the same bf16 operands.
python tools/wgrad16_time.py [F,K,M,H,W ...]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rubiksnet_amd import _native

L = _native.lib()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
SHAPES = [(256, 288, 288, 14, 14), (256, 144, 144, 28, 28), (256, 72, 72, 56, 56), (256, 72, 144, 56, 56), (256, 144, 288, 28, 28),
          (256, 288, 576, 14, 14), (256, 72, 72, 112, 112)]
if len(sys.argv) > 1:
    SHAPES = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
tag = " ".join(f"{k}={v}" for k, v in sorted(os.environ.items()) if k.startswith("RK_WG16"))
for (Fr, K, M, H, W) in SHAPES:
    P = H * W
    sets = [(torch.randn(Fr, K, P, device=dev).bfloat16(), torch.randn(Fr, M, P, device=dev).bfloat16()) for _ in range(3)]
    nb = int(L.rk_pw_wgrad16_workspace_bytes(Fr, K, M, P))
    ws = torch.empty(max(nb, 1), dtype=torch.uint8, device=dev)
    dw = torch.empty(M, K, device=dev)
    def run(i):
        x, g = sets[i % 3]
        _native.check(L.rk_pw_wgrad16_bf16(g.data_ptr(), x.data_ptr(), dw.data_ptr(), Fr, K, M, P, ws.data_ptr(), nb, st), "w")
    run(0)
    x, g = sets[0]
    ref = torch.einsum("fmp,fkp->mk", g[:64].double(), x[:64].double()) if Fr * P > 300000 else None
    if ref is None:
        ref = torch.einsum("fmp,fkp->mk", g.double(), x.double())
        err = float((dw.double() - ref).abs().max() / ref.abs().max())
    else:   # large: check the full result against chunks summed in fp64
        ref = sum(torch.einsum("fmp,fkp->mk", g[i:i + 32].double(), x[i:i + 32].double()) for i in range(0, Fr, 32))
        err = float((dw.double() - ref).abs().max() / ref.abs().max())
    for i in range(10): run(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(50): run(i)
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) * 1e3 / 50
    by = Fr * (K + M) * P * 2
    print(f"[{tag}] {(Fr, K, M, H, W)} wgrad16 {us:8.1f} us  {by / us / 1e6 / 8:6.3f} of 8 TB/s  ws {nb >> 20} MB  rel.err {err:.2e}", flush=True)
