#!/usr/bin/env python
"""1x1-convolution probe: F.conv2d (MIOpen) versus the same contraction as a broadcast matmul on the NCHW
tensor ([F, Cin, P] -> [F, Cout, P], no layout change), forward and backward, fp32 / bf16."""
import sys, torch, torch.nn.functional as F_
dev = torch.device("cuda:0")
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
            tcf = timeit(lambda: F_.conv2d(x, w)); tmf = timeit(lambda: mm_fwd(x, w))
        tc = timeit(conv_fb); tm = timeit(mm_fb)
        es = x.element_size()
        ideal = es * (Cin + Cout) * Fr * H * W / 5.0e6   # us at 5 TB/s, one pass over x and y
        print("%-8s [%d,%d->%d,%dx%d] fwd conv %7.1f us  matmul %7.1f us | fwd+bwd conv %7.1f us  matmul %7.1f us | 1 pass @5TB/s %6.1f us" % (
            str(dt).split(".")[1], Fr, Cin, Cout, H, W, tcf, tmf, tc, tm, ideal))
