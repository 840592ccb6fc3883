"""7x7 planes in bf16 (P = 49: no 16-byte unit in a row, rk_pw16 cannot take them): F.conv2d (MIOpen: NCHW <-> NHWC transposes +
implicit GEMM) against the same product as a broadcast batched GEMM on the NCHW tensor (rocBLAS / hipBLASLt, no transposes).
python tools/odd_gemm_probe.py"""
import torch
import torch.nn.functional as F

dev = torch.device("cuda:0")


def timed(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for (Fr, K, M, H, s) in ((256, 576, 576, 7, 1), (256, 576, 1152, 7, 1), (256, 1152, 576, 7, 1), (256, 1152, 1152, 7, 1), (256, 576, 1152, 14, 2)):
    x = torch.randn(Fr, K, H, H, device=dev).bfloat16().requires_grad_(True)
    w = (torch.randn(M, K, 1, 1, device=dev) * 0.05).bfloat16().requires_grad_(True)
    Ho = (H - 1) // s + 1
    gy = torch.randn(Fr, M, Ho, Ho, device=dev).bfloat16()
    P = Ho * Ho
    w2 = w.detach().view(M, K)

    def conv_f():
        return F.conv2d(x, w, stride=s)

    def conv_all():
        y = F.conv2d(x, w, stride=s)
        y.backward(gy)
        x.grad = None; w.grad = None

    xs = x.detach()[:, :, ::s, ::s].contiguous() if s > 1 else x.detach()

    def mm_f():
        return torch.matmul(w2, xs.view(Fr, K, P)).view(Fr, M, Ho, Ho)

    def mm_dgrad():
        return torch.matmul(w2.t(), gy.view(Fr, M, P))

    def mm_wgrad():
        a = gy.view(Fr, M, P).permute(1, 0, 2).reshape(M, Fr * P)
        b = xs.view(Fr, K, P).permute(1, 0, 2).reshape(K, Fr * P)
        return torch.matmul(a, b.t())

    def mm_wgrad_bmm():
        return torch.bmm(gy.view(Fr, M, P), xs.view(Fr, K, P).transpose(1, 2)).sum(0)

    y1, y2 = conv_f(), mm_f()
    err = float((y1.float() - y2.float()).abs().max() / y1.float().abs().max())
    print(f"[{Fr},{K}->{M},{H}x{H}] s{s}: conv fwd {timed(conv_f):.1f} us, fwd+bwd {timed(conv_all):.1f} us | matmul fwd {timed(mm_f):.1f}, "
          f"dgrad {timed(mm_dgrad):.1f}, wgrad (permute + gemm) {timed(mm_wgrad):.1f}, wgrad (bmm + sum) {timed(mm_wgrad_bmm):.1f}  rel err {err:.1e}",
          flush=True)
