#!/usr/bin/env python
"""Per-kernel timing of the pieces of a training block, fused vs unfused, on the Tiny / Large layer shapes
(back-to-back launches between one pair of events after a run-in; 3 rotating buffer sets)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rubiksnet_amd import _native

L = _native.lib()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream


def timed(fn, iters=30, settle=0.15):
    t_end = time.perf_counter() + settle
    i = 0
    while time.perf_counter() < t_end:
        for _ in range(5):
            fn(i); i += 1
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(iters):
        fn(i + k)
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


shapes = [(256, 54, 54, 112, 112), (256, 54, 54, 56, 56), (256, 108, 108, 28, 28), (256, 216, 216, 14, 14), (256, 72, 72, 56, 56),
          (256, 288, 288, 14, 14)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
for Fr, K, M, H, W in shapes:
    P = H * W
    E = Fr * K * P * 4 / 1e6
    sets = []
    for _ in range(3):
        sets.append(dict(x=torch.randn(Fr, K, P, device=dev), y=torch.empty(Fr, M, P, device=dev), r=torch.randn(Fr, M, P, device=dev),
                         g=torch.randn(Fr, M, P, device=dev), o=torch.empty(Fr, K, P, device=dev)))
    w = torch.randn(M, K, device=dev) / K ** 0.5
    ka, kb = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.1
    J = int(L.rk_pw_tiles(Fr, P))
    stats = torch.empty(M, J, 4, device=dev)
    bred = torch.empty(K, J, 2, device=dev)
    pack = torch.stack([ka, kb, kb, ka], dim=1).contiguous()
    fin = torch.empty(8, M, device=dev)
    gam, bet = torch.ones(M, device=dev), torch.zeros(M, device=dev)
    k12 = torch.zeros(2, K, device=dev); dg = torch.empty(K, device=dev); db = torch.empty(K, device=dev)
    nb = int(L.rk_pw_wgrad_workspace_bytes(Fr, K, M, P)); ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    dw = torch.empty(M, K, device=dev)
    nbn = int(L.rk_bn_workspace_bytes(Fr, K, P)); wsb = torch.empty(max(nbn, 1), dtype=torch.uint8, device=dev)
    sm, si = torch.zeros(K, device=dev), torch.ones(K, device=dev)
    S = lambda i: sets[i % 3]
    res = {}
    res["gemm"] = timed(lambda i: L.rk_pw_gemm_f32(w.data_ptr(), S(i)["x"].data_ptr(), None, S(i)["y"].data_ptr(), Fr, K, M, P, 1, st))
    res["gemm+R"] = timed(lambda i: L.rk_pw_gemm_f32(w.data_ptr(), S(i)["x"].data_ptr(), S(i)["r"].data_ptr(), S(i)["y"].data_ptr(), Fr, K, M, P, 1, st))
    res["gemm_pro"] = timed(lambda i: L.rk_pw_gemm_fused_f32(w.data_ptr(), S(i)["x"].data_ptr(), None, S(i)["y"].data_ptr(), Fr, K, M, P, 1, ka.data_ptr(), kb.data_ptr(), 1, None, None, 0, st))
    res["gemm_stats"] = timed(lambda i: L.rk_pw_gemm_stats_f32(w.data_ptr(), S(i)["x"].data_ptr(), None, S(i)["y"].data_ptr(), Fr, K, M, P, 1, None, None, 0, stats.data_ptr(), J, st))
    res["gemm_pro_stats"] = timed(lambda i: L.rk_pw_gemm_stats_f32(w.data_ptr(), S(i)["x"].data_ptr(), None, S(i)["y"].data_ptr(), Fr, K, M, P, 1, ka.data_ptr(), kb.data_ptr(), 1, stats.data_ptr(), J, st))
    res["gemm_R_stats"] = timed(lambda i: L.rk_pw_gemm_stats_f32(w.data_ptr(), S(i)["x"].data_ptr(), S(i)["r"].data_ptr(), S(i)["y"].data_ptr(), Fr, K, M, P, 1, None, None, 0, stats.data_ptr(), J, st))
    res["dgrad"] = timed(lambda i: L.rk_pw_gemm_f32(w.data_ptr(), S(i)["g"].data_ptr(), None, S(i)["o"].data_ptr(), Fr, M, K, P, 0, st))
    res["dgrad_bnbwd"] = timed(lambda i: L.rk_pw_gemm_bnbwd_f32(w.data_ptr(), S(i)["g"].data_ptr(), None, S(i)["o"].data_ptr(), Fr, M, K, P, 0, S(i)["x"].data_ptr(), pack.data_ptr(), bred.data_ptr(), J, st))
    res["wgrad"] = timed(lambda i: L.rk_pw_wgrad_f32(S(i)["g"].data_ptr(), S(i)["x"].data_ptr(), dw.data_ptr(), Fr, K, M, P, ws.data_ptr(), nb, st))
    res["wgrad_pro"] = timed(lambda i: L.rk_pw_wgrad_pro_f32(S(i)["g"].data_ptr(), S(i)["x"].data_ptr(), dw.data_ptr(), Fr, K, M, P, ka.data_ptr(), kb.data_ptr(), 1, ws.data_ptr(), nb, st))
    res["finish"] = timed(lambda i: L.rk_bn_finish_tiles_f32(stats.data_ptr(), J, Fr * P, gam.data_ptr(), bet.data_ptr(), None, None, fin[0].data_ptr(), fin[1].data_ptr(), fin[2].data_ptr(), fin[3].data_ptr(), fin[4].data_ptr(), M, 1e-5, 0.1, None, st))
    res["bwd_finish"] = timed(lambda i: L.rk_bn_bwd_finish_tiles_f32(bred.data_ptr(), J, Fr * P, k12.data_ptr(), dg.data_ptr(), db.data_ptr(), K, st))
    res["tile_stats"] = timed(lambda i: L.rk_bn_tile_stats_f32(S(i)["x"].data_ptr(), stats.data_ptr(), Fr, K, P, st)) if K == M else 0
    res["apply_affine"] = timed(lambda i: L.rk_bn_apply_affine_f32(S(i)["x"].data_ptr(), ka.data_ptr(), kb.data_ptr(), S(i)["o"].data_ptr(), Fr, K, P, 1, st))
    res["dx_pre"] = timed(lambda i: L.rk_bn_bwd_dx_pre_f32(S(i)["o"].data_ptr(), S(i)["x"].data_ptr(), ka.data_ptr(), sm.data_ptr(), si.data_ptr(), k12.data_ptr(), None, S((i + 1))["o"].data_ptr(), Fr, K, P, st))
    res["bn_bwd(2k)"] = timed(lambda i: L.rk_bn_relu_backward_f32(S(i)["o"].data_ptr(), S(i)["x"].data_ptr(), ka.data_ptr(), kb.data_ptr(), sm.data_ptr(), si.data_ptr(), None, S((i + 1))["o"].data_ptr(), dg.data_ptr(), db.data_ptr(), Fr, K, P, 1, wsb.data_ptr(), nbn, st))
    print("[%d,%d->%d,%dx%d] E=%.0f MB: " % (Fr, K, M, H, W, E) + "  ".join("%s %.0f" % (k, v) for k, v in res.items()), flush=True)
    del sets
    torch.cuda.empty_cache()
