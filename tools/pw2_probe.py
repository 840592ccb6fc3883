"""Probe of the second-generation fp32 1x1 kernels (rk_pw2.hip) against the first generation (rk_pw.hip): parity vs an
fp64 einsum and steady-state time per call.  python tools/pw2_probe.py [gemm|wgrad|all]"""
import sys
import torch
sys.path.insert(0, ".")
from rubiksnet_amd import _native

L = _native.lib()
dev = torch.device("cuda:0")
st = lambda: torch.cuda.current_stream(dev).cuda_stream
PEAK = 157.3e12


def timeit(fn, n=20, warm=4):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def gemm_case(F, K, M, P, mk, res=False, pro=False, cfgs=((0, -1, 0),)):
    torch.manual_seed(0)
    x = torch.randn(F, K, P, device=dev)
    w = torch.randn(M, K, device=dev) / K ** 0.5
    a = w if mk else w.t().contiguous()
    r = torch.randn(F, M, P, device=dev) if res else None
    ka = torch.rand(K, device=dev) + 0.5 if pro else None
    kb = torch.randn(K, device=dev) * 0.1 if pro else None
    xr = x.double()
    if pro:
        xr = torch.relu(ka.double()[None, :, None] * xr + kb.double()[None, :, None])
    ref = torch.einsum("mk,fkp->fmp", w.double(), xr)
    if res:
        ref = ref + r.double()
    scale = ref.abs().max().item()
    y1, y2 = torch.empty(F, M, P, device=dev), torch.empty(F, M, P, device=dev)
    flops = 2.0 * F * P * K * M
    byts = 4.0 * F * P * (K + M * (2 if res else 1))
    out = []
    if not pro:
        f1 = lambda: _native.check(L.rk_pw_gemm_f32(a.data_ptr(), x.data_ptr(), r.data_ptr() if res else None, y1.data_ptr(),
                                                    F, K, M, P, int(mk), st()), "v1")
        t1 = timeit(f1)
        e1 = (y1.double() - ref).abs().max().item() / scale
        out.append(("v1", t1, e1))
    for rb, am, ct in cfgs:
        y2.zero_()
        f2 = lambda: L.rk_pw2_gemm_cfg_f32(a.data_ptr(), x.data_ptr(), r.data_ptr() if res else None, y2.data_ptr(), F, K, M, P,
                                           int(mk), ka.data_ptr() if pro else None, kb.data_ptr() if pro else None, 1, rb, am,
                                           ct, st())
        rc = f2()
        if rc != 0:
            out.append(("v2 rb%d am%d ct%d" % (rb, am, ct), float("nan"), float(rc)))
            continue
        t2 = timeit(f2)
        e2 = (y2.double() - ref).abs().max().item() / scale
        out.append(("v2 rb%d am%d ct%d" % (rb, am, ct), t2, e2))
    print("GEMM [%d,%d->%d,P=%d] mk=%d res=%d pro=%d  (mfma floor %.1f us, hbm floor %.1f us)" %
          (F, K, M, P, mk, res, pro, flops / PEAK * 1e6, byts / 8e12 * 1e6))
    for name, t, e in out:
        print("   %-18s %8.1f us  %5.1f%% mfma  %5.1f%% hbm   relerr %.2e" % (name, t, flops / PEAK * 1e8 / t, byts / 8e12 * 1e8 / t, e))
    sys.stdout.flush()


def wgrad_case(F, K, M, P, pro=False, cfgs=((-1, 3, 0),)):
    torch.manual_seed(0)
    x = torch.randn(F, K, P, device=dev)
    dy = torch.randn(F, M, P, device=dev)
    ka = torch.rand(K, device=dev) + 0.5 if pro else None
    kb = torch.randn(K, device=dev) * 0.1 if pro else None
    xr = x.double()
    if pro:
        xr = torch.relu(ka.double()[None, :, None] * xr + kb.double()[None, :, None])
    ref = torch.einsum("fmp,fkp->mk", dy.double(), xr)
    scale = ref.abs().max().item()
    flops = 2.0 * F * P * K * M
    byts = 4.0 * F * P * (K + M)
    out = []
    dw = torch.empty(M, K, device=dev)
    nb1 = int(L.rk_pw_wgrad_workspace_bytes(F, K, M, P))
    ws1 = torch.empty(max(nb1, 1), dtype=torch.uint8, device=dev)
    if pro:
        f1 = lambda: _native.check(L.rk_pw_wgrad_pro_f32(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), F, K, M, P, ka.data_ptr(),
                                                         kb.data_ptr(), 1, ws1.data_ptr(), nb1, st()), "v1")
    else:
        f1 = lambda: _native.check(L.rk_pw_wgrad_f32(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), F, K, M, P, ws1.data_ptr(), nb1,
                                                     st()), "v1")
    t1 = timeit(f1)
    out.append(("v1", t1, (dw.double() - ref).abs().max().item() / scale))
    nb2 = 1 << 28
    ws2 = torch.empty(nb2, dtype=torch.uint8, device=dev)
    for inst, ns, sp in cfgs:
        dw.zero_()
        f2 = lambda: L.rk_pw2_wgrad_cfg_f32(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), F, K, M, P, ws2.data_ptr(), nb2,
                                            ka.data_ptr() if pro else None, kb.data_ptr() if pro else None, 1, inst, ns, sp, st())
        rc = f2()
        name = "v2 i%d ns%d S%d" % (inst, ns, sp)
        if rc != 0:
            out.append((name, float("nan"), float(rc)))
            continue
        t2 = timeit(f2)
        out.append((name, t2, (dw.double() - ref).abs().max().item() / scale))
    print("WGRAD [%d,%d->%d,P=%d] pro=%d  (mfma floor %.1f us, hbm floor %.1f us)" % (F, K, M, P, pro, flops / PEAK * 1e6, byts / 8e12 * 1e6))
    for name, t, e in out:
        print("   %-18s %8.1f us  %5.1f%% mfma  %5.1f%% hbm   relerr %.2e" % (name, t, flops / PEAK * 1e8 / t, byts / 8e12 * 1e8 / t, e))
    sys.stdout.flush()


what = sys.argv[1] if len(sys.argv) > 1 else "all"
if what == "one":            # one case, e.g.: one gemm 256 288 288 196 1 rb am ct   |   one wgrad 256 288 288 196 inst ns S
    a = [int(v) for v in sys.argv[3:]]
    if sys.argv[2] == "gemm":
        gemm_case(a[0], a[1], a[2], a[3], a[4], cfgs=((a[5], a[6], a[7]),))
    else:
        wgrad_case(a[0], a[1], a[2], a[3], cfgs=((a[4], a[5] if a[5] > 0 else 0, a[6]),))
if what in ("gemm", "all"):
    gemm_case(256, 288, 288, 196, 1, cfgs=((0, -1, 0), (3, -1, 1), (4, -1, 1), (5, -1, 1), (3, 2, 1)))
    gemm_case(256, 288, 288, 196, 0, cfgs=((0, -1, 0), (4, -1, 1), (5, -1, 1)))
    gemm_case(256, 288, 288, 196, 1, res=True)
    gemm_case(256, 288, 288, 196, 1, pro=True)
    gemm_case(256, 144, 144, 784, 1, cfgs=((0, -1, 0), (3, -1, 1), (5, -1, 1)))
    gemm_case(256, 144, 144, 784, 0)
    gemm_case(256, 72, 72, 3136, 1, cfgs=((0, -1, 0), (5, 2, 4), (5, -1, 2)))
    gemm_case(256, 72, 72, 3136, 0)
    gemm_case(256, 72, 144, 3136, 1)
    gemm_case(256, 216, 216, 196, 1, cfgs=((0, -1, 0), (3, -1, 1), (4, -1, 1)))
    gemm_case(256, 216, 216, 196, 0)
    gemm_case(256, 108, 108, 784, 1, cfgs=((0, -1, 0), (4, 2, 2), (3, -1, 2)))
    gemm_case(256, 108, 108, 784, 0)
    gemm_case(256, 54, 54, 3136, 1, cfgs=((0, -1, 0), (4, 2, 2)))
    gemm_case(256, 54, 54, 3136, 0)
    gemm_case(256, 54, 108, 3136, 1)
    gemm_case(256, 24, 54, 12544, 1)
    gemm_case(16, 54, 54, 3136, 1, res=True, pro=True)
    gemm_case(7, 30, 22, 36, 1, res=True, pro=True)
    gemm_case(7, 30, 22, 36, 0, res=True)
if what in ("wgrad", "all"):
    wgrad_case(256, 288, 288, 196, cfgs=((0, 3, 0), (1, 3, 0), (1, 2, 0), (2, 2, 0), (2, 3, 0), (9, 2, 0), (1, 3, 32), (1, 3, 96), (0, 3, 96)))
    wgrad_case(256, 288, 288, 196, pro=True, cfgs=((1, 3, 0),))
    wgrad_case(256, 144, 144, 784, cfgs=((0, 3, 0), (1, 3, 0), (2, 2, 0), (7, 3, 0)))
    wgrad_case(256, 72, 72, 3136, cfgs=((8, 3, 0), (1, 3, 0), (3, 3, 0)))
    wgrad_case(256, 216, 216, 196, cfgs=((7, 3, 0), (6, 3, 0), (1, 3, 0), (2, 2, 0)))
    wgrad_case(256, 108, 108, 784, cfgs=((7, 3, 0), (6, 3, 0), (1, 3, 0)))
    wgrad_case(256, 54, 54, 3136, cfgs=((4, 3, 0), (5, 3, 0), (4, 2, 0)))
    wgrad_case(256, 54, 108, 3136, cfgs=((-1, 3, 0), (4, 3, 0)))
    wgrad_case(7, 30, 22, 36, pro=True, cfgs=((-1, 3, 0), (1, 2, 3), (8, 3, 2)))
    wgrad_case(5, 100, 50, 64, cfgs=((-1, 3, 0), (7, 2, 2), (0, 3, 1)))
