#!/usr/bin/env python
"""Streaming GEMM of the shallow layers (rk_pw4.hip) against the dispatch's kernel (rk_pw_gemm*_f32): results + time.
   python tools/pw4_probe.py [F]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rubiksnet_amd import _native

L = _native.lib()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
Fr = int(sys.argv[1]) if len(sys.argv) > 1 else 256


def timeit(fn, iters=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for (K, M, H) in ((54, 54, 56), (72, 72, 56), (54, 54, 112), (72, 72, 112), (52, 60, 28), (66, 70, 20)):
    P = H * H
    torch.manual_seed(0)
    sets = [dict(x=torch.randn(Fr, K, P, device=dev), r=torch.randn(Fr, M, P, device=dev), y=torch.empty(Fr, M, P, device=dev),
                 y2=torch.empty(Fr, M, P, device=dev), bx=torch.randn(Fr, M, P, device=dev)) for _ in range(2)]
    w = torch.randn(M, K, device=dev) / K ** 0.5
    wt = w.t().contiguous()
    ka, kb = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.3
    pack = torch.stack([torch.rand(M, device=dev) + 0.5, torch.randn(M, device=dev) * 0.3, torch.randn(M, device=dev) * 0.1,
                        torch.rand(M, device=dev) + 0.5], dim=1).contiguous()
    J = (Fr * P + 63) // 64
    Jd = int(L.rk_pw_gemm_tiles(w.data_ptr(), Fr, K, M, P, 1))
    s4, s2 = torch.zeros(M, J, 4, device=dev), torch.zeros(M, Jd, 4, device=dev)
    b4, b2 = torch.zeros(M, J, 2, device=dev), torch.zeros(M, Jd, 2, device=dev)
    it = [0]

    def nxt():
        it[0] += 1
        return sets[it[0] % 2]

    N = None
    cases = [
        ("plain mk", lambda s, y: L.rk_pw4_gemm_f32(w.data_ptr(), s["x"].data_ptr(), N, y.data_ptr(), Fr, K, M, P, 1, N, N, 0, 0, N, N, N, N, J, st),
                     lambda s, y: L.rk_pw_gemm_f32(w.data_ptr(), s["x"].data_ptr(), N, y.data_ptr(), Fr, K, M, P, 1, st), None),
        ("plain km", lambda s, y: L.rk_pw4_gemm_f32(wt.data_ptr(), s["x"].data_ptr(), N, y.data_ptr(), Fr, K, M, P, 0, N, N, 0, 0, N, N, N, N, J, st),
                     lambda s, y: L.rk_pw_gemm_f32(wt.data_ptr(), s["x"].data_ptr(), N, y.data_ptr(), Fr, K, M, P, 0, st), None),
        ("+R", lambda s, y: L.rk_pw4_gemm_f32(w.data_ptr(), s["x"].data_ptr(), s["r"].data_ptr(), y.data_ptr(), Fr, K, M, P, 1, N, N, 0, 0, N, N, N, N, J, st),
               lambda s, y: L.rk_pw_gemm_f32(w.data_ptr(), s["x"].data_ptr(), s["r"].data_ptr(), y.data_ptr(), Fr, K, M, P, 1, st), None),
        ("pro+stats", lambda s, y: L.rk_pw4_gemm_f32(w.data_ptr(), s["x"].data_ptr(), N, y.data_ptr(), Fr, K, M, P, 1, ka.data_ptr(), kb.data_ptr(), 1, 1, s4.data_ptr(), N, N, N, J, st),
                      lambda s, y: L.rk_pw_gemm_stats_f32(w.data_ptr(), s["x"].data_ptr(), N, y.data_ptr(), Fr, K, M, P, 1, ka.data_ptr(), kb.data_ptr(), 1, s2.data_ptr(), Jd, st), "stats"),
        ("R+stats", lambda s, y: L.rk_pw4_gemm_f32(w.data_ptr(), s["x"].data_ptr(), s["r"].data_ptr(), y.data_ptr(), Fr, K, M, P, 1, N, N, 0, 1, s4.data_ptr(), N, N, N, J, st),
                    lambda s, y: L.rk_pw_gemm_stats_f32(w.data_ptr(), s["x"].data_ptr(), s["r"].data_ptr(), y.data_ptr(), Fr, K, M, P, 1, N, N, 0, s2.data_ptr(), Jd, st), "stats"),
        ("km bnbwd", lambda s, y: L.rk_pw4_gemm_f32(wt.data_ptr(), s["x"].data_ptr(), N, y.data_ptr(), Fr, K, M, P, 0, N, N, 0, 2, N, s["bx"].data_ptr(), pack.data_ptr(), b4.data_ptr(), J, st),
                     lambda s, y: L.rk_pw_gemm_bnbwd_f32(wt.data_ptr(), s["x"].data_ptr(), N, y.data_ptr(), Fr, K, M, P, 0, s["bx"].data_ptr(), pack.data_ptr(), b2.data_ptr(), Jd, st), "bred"),
    ]
    if len(sys.argv) > 2 and sys.argv[2] == "one":                      # PMC mode: one case, the streaming kernel only
        if (K, H) != (int(sys.argv[3]), int(sys.argv[4])): continue
        f4 = cases[int(sys.argv[5])][1]
        for _ in range(8): f4(nxt(), nxt()["y"])
        torch.cuda.synchronize(); print("done"); continue
    print("[%d, %d -> %d, %dx%d]  J=%d (dispatch %d)" % (Fr, K, M, H, H, J, Jd))
    for name, f4, f2, extra in cases:
        s = sets[0]
        rc4 = f4(s, s["y"]); rc2 = f2(s, s["y2"])
        torch.cuda.synchronize()
        if rc4 != 0:
            print("   %-10s rc4=%d rc2=%d" % (name, rc4, rc2)); continue
        err = float((s["y"] - s["y2"]).abs().max())
        e2 = ""
        if extra == "stats" and J == Jd:
            # compare finished sums per row: sum over tiles of (n piv + s1)
            t4 = (s4[..., 0] * s4[..., 3] + s4[..., 1]).double().sum(1); t2 = (s2[..., 0] * s2[..., 3] + s2[..., 1]).double().sum(1)
            e2 = " stats-sum err %.2e cnt %.0f/%.0f" % (float((t4 - t2).abs().max()), float(s4[0, :, 3].sum()), float(s2[0, :, 3].sum()))
        if extra == "bred" and J == Jd:
            e2 = " bred err %.2e" % float((b4 - b2).abs().max())
        t4 = timeit(lambda: f4(nxt(), nxt()["y"])); t2 = timeit(lambda: f2(nxt(), nxt()["y2"]))
        print("   %-10s max|dy| %.2e%s   pw4 %.1f us   dispatch %.1f us" % (name, err, e2, t4, t2))
    del sets
