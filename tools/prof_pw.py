#!/usr/bin/env python
"""Minimal driver for profiling the 1x1-convolution kernels: python tools/prof_pw.py <which> F K M H W [iters]
which: wgrad | wgrad_pro | gemm | gemm_stats | dgrad_bnbwd | gemm_odd | gemm16 | gemm16res | wgrad16 | bn_dx (K channels)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rubiksnet_amd import _native

which = sys.argv[1]
Fr, K, M, H, W = (int(v) for v in sys.argv[2:7])
iters = int(sys.argv[7]) if len(sys.argv) > 7 else 6
L = _native.lib()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
P = H * W
sets = [dict(x=torch.randn(Fr, K, P, device=dev), g=torch.randn(Fr, M, P, device=dev), y=torch.empty(Fr, M, P, device=dev),
             o=torch.empty(Fr, K, P, device=dev)) for _ in range(3)]
w = torch.randn(M, K, device=dev) / K ** 0.5
ka, kb = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.1
J = int(L.rk_pw_gemm_tiles(w.data_ptr(), Fr, K, M, P, 1))           # tiles of the statistics epilogue: by kernel generation
Jb = int(L.rk_pw_gemm_tiles(w.data_ptr(), Fr, M, K, P, 0))
stats = torch.empty(M, J, 4, device=dev); bred = torch.empty(K, Jb, 2, device=dev)
pack = torch.stack([ka, kb, kb, ka], dim=1).contiguous()
nb = int(L.rk_pw_wgrad_workspace_bytes(Fr, K, M, P)) if P % 4 == 0 else int(L.rk_pw_wgrad_odd_workspace_bytes(Fr, K, M, P))
ws = torch.empty(max(nb, 1), dtype=torch.uint8, device=dev)
dw = torch.empty(M, K, device=dev)
if which in ("gemm16", "gemm16res", "wgrad16"):
    for d_ in sets:
        for k_ in d_: d_[k_] = d_[k_].bfloat16()
    pk = torch.empty(int(L.rk_pw_packed_bytes(M, K)), dtype=torch.uint8, device=dev)
    _native.check(L.rk_pw_pack_bf16(w.data_ptr(), M, K, pk.data_ptr(), None, st), "pack")
    nb16 = int(L.rk_pw_wgrad16_workspace_bytes(Fr, K, M, P)); ws16 = torch.empty(max(nb16, 1), dtype=torch.uint8, device=dev)
if which == "bn_dx":
    gamma, mean, inv = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev), torch.rand(K, device=dev) + 0.5
    k12 = torch.randn(2, K, device=dev) * 0.01
for i in range(iters):
    s = sets[i % 3]
    if which == "bn_dx":
        _native.check(L.rk_bn_bwd_dx_pre_f32(s["o"].data_ptr(), s["x"].data_ptr(), gamma.data_ptr(), mean.data_ptr(), inv.data_ptr(),
                                             k12.data_ptr(), None, s["o"].data_ptr(), Fr, K, P, st), which)
        continue
    if which == "gemm16":
        rc = L.rk_pw_gemm_packed_bf16(pk.data_ptr(), s["x"].data_ptr(), None, s["y"].data_ptr(), Fr, K, M, P, st)
    elif which == "gemm16res":
        rc = L.rk_pw_gemm_packed_bf16(pk.data_ptr(), s["x"].data_ptr(), s["g"].data_ptr(), s["y"].data_ptr(), Fr, K, M, P, st)
    elif which == "wgrad16":
        rc = L.rk_pw_wgrad16_bf16(s["g"].data_ptr(), s["x"].data_ptr(), dw.data_ptr(), Fr, K, M, P, ws16.data_ptr(), nb16, st)
    if which in ("gemm16", "gemm16res", "wgrad16"):
        _native.check(rc, which)
        continue
    if which == "wgrad":
        rc = L.rk_pw_wgrad_f32(s["g"].data_ptr(), s["x"].data_ptr(), dw.data_ptr(), Fr, K, M, P, ws.data_ptr(), nb, st)
    elif which == "wgrad_pro":
        rc = L.rk_pw_wgrad_pro_f32(s["g"].data_ptr(), s["x"].data_ptr(), dw.data_ptr(), Fr, K, M, P, ka.data_ptr(), kb.data_ptr(), 1, ws.data_ptr(), nb, st)
    elif which == "gemm":
        rc = L.rk_pw_gemm_f32(w.data_ptr(), s["x"].data_ptr(), None, s["y"].data_ptr(), Fr, K, M, P, 1, st)
    elif which == "gemm_stats":
        rc = L.rk_pw_gemm_stats_f32(w.data_ptr(), s["x"].data_ptr(), None, s["y"].data_ptr(), Fr, K, M, P, 1, ka.data_ptr(), kb.data_ptr(), 1, stats.data_ptr(), J, st)
    elif which == "dgrad_bnbwd":
        rc = L.rk_pw_gemm_bnbwd_f32(w.data_ptr(), s["g"].data_ptr(), None, s["o"].data_ptr(), Fr, M, K, P, 0, s["x"].data_ptr(), pack.data_ptr(), bred.data_ptr(), Jb, st)
    elif which == "gemm_odd":
        rc = L.rk_pw_gemm_odd_f32(w.data_ptr(), s["x"].data_ptr(), None, s["y"].data_ptr(), Fr, K, M, P, 1, st)
    _native.check(rc, which)
torch.cuda.synchronize()
print("done", which)
