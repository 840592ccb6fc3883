#!/usr/bin/env python
"""1x1-convolution probe: F.conv2d (MIOpen) versus the same contraction as a broadcast matmul on the NCHW
tensor ([F, Cin, P] -> [F, Cout, P], no layout change), forward and backward, fp32 / bf16."""
import os, sys, torch, torch.nn.functional as F_
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rubiksnet_amd import _native
dev = torch.device("cuda:0")
def pw_fwd(x, w, out=None):
    Fr, Cin, H, W = x.shape
    Cout = w.shape[0]
    y = torch.empty(Fr, Cout, H, W, device=x.device, dtype=x.dtype) if out is None else out
    rc = getattr(_native.lib(), "rk_pw_gemm_" + ("f32" if x.dtype == torch.float32 else "bf16"))(w.data_ptr(), x.data_ptr(), None, y.data_ptr(), Fr, Cin, Cout, H * W, 1,
                                      torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc
    return y
def pw_dx(gy, w):
    Fr, Cout, H, W = gy.shape
    Cin = w.shape[1]
    gx = torch.empty(Fr, Cin, H, W, device=gy.device, dtype=gy.dtype)
    rc = getattr(_native.lib(), "rk_pw_gemm_" + ("f32" if gy.dtype == torch.float32 else "bf16"))(w.data_ptr(), gy.data_ptr(), None, gx.data_ptr(), Fr, Cout, Cin, H * W, 0,
                                      torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc
    return gx
def timeit(fn, iters=12):
    for _ in range(3): fn()
    ev = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); ev.append((a, b))
    torch.cuda.synchronize()
    t = sorted(x.elapsed_time(y) for x, y in ev)
    return t[len(t) // 2] * 1e3
def mm_fwd(x, w):
    Fr, Cin, H, W = x.shape
    return torch.matmul(w.view(w.shape[0], Cin), x.view(Fr, Cin, H * W)).view(Fr, w.shape[0], H, W)
shapes = [(256, 54, 54, 56, 56), (256, 54, 108, 56, 56), (256, 108, 108, 28, 28), (256, 216, 216, 14, 14), (256, 432, 432, 7, 7), (256, 24, 54, 112, 112)]
for dt in (torch.float32, torch.bfloat16):
    for (Fr, Cin, Cout, H, W) in shapes:
        x = torch.randn(Fr, Cin, H, W, device=dev, dtype=dt, requires_grad=True)
        w = torch.randn(Cout, Cin, 1, 1, device=dev, dtype=dt, requires_grad=True)
        gy = torch.randn(Fr, Cout, H, W, device=dev, dtype=dt)
        def conv_fb():
            y = F_.conv2d(x, w); y.backward(gy); x.grad = None; w.grad = None
        def mm_fb():
            y = mm_fwd(x, w); y.backward(gy); x.grad = None; w.grad = None
        with torch.no_grad():
            tcf = timeit(lambda: F_.conv2d(x, w)); tmf = 0.0
            if (H * W) % 4 == 0:
                wf = w.float()                     # the HIP kernels take the fp32 weight whatever the activations are
                yr = F_.conv2d(x, w); yk = pw_fwd(x, wf)
                gr = torch.nn.grad.conv2d_input(x.shape, w, gy); gk = pw_dx(gy, wf)
                e1 = float((yk.float() - yr.float()).abs().max() / yr.float().abs().max()); e2 = float((gk.float() - gr.float()).abs().max() / gr.float().abs().max())
                tk = timeit(lambda: pw_fwd(x, wf)); tkx = timeit(lambda: pw_dx(gy, wf))
                from rubiksnet_amd.pointwise import _wgrad
                wr = torch.ops.aten.convolution_backward(gy, x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [False, True, False])[1]
                wk = _wgrad(gy, x.detach(), wf.detach())
                e3 = float((wk.float() - wr.float()).abs().max() / wr.float().abs().max())
                tw = timeit(lambda: _wgrad(gy, x.detach(), wf.detach()))
                twr = timeit(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [False, True, False]))
                tdr = timeit(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [True, False, False]))
                print("   rk_pw fwd %7.1f us (rel err %.1e) | d(input) %7.1f us (rel err %.1e; MIOpen %7.1f) | d(weight) %7.1f us (rel err %.1e; MIOpen %7.1f)" % (tk, e1, tkx, e2, tdr, tw, e3, twr))
        if os.environ.get("PW_ONLY"): continue
        tc = timeit(conv_fb); tm = 0.0
        es = x.element_size()
        ideal = es * (Cin + Cout) * Fr * H * W / 5.0e6   # us at 5 TB/s, one pass over x and y
        print("%-8s [%d,%d->%d,%dx%d] fwd conv %7.1f us  matmul %7.1f us | fwd+bwd conv %7.1f us  matmul %7.1f us | 1 pass @5TB/s %6.1f us" % (
            str(dt).split(".")[1], Fr, Cin, Cout, H, W, tcf, tmf, tc, tm, ideal))
