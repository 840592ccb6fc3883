"""conv3(shift3d(x)) + residual as ONE launch (shift in the GEMM's operand load) against the shift
kernel followed by the GEMM, per layer shape of the nets (inference, fp32), and the Tiny forward at batch 64."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rubiksnet_amd import _native
from rubiksnet_amd.shiftlib.rubiks3d.primitive import rubiks_shift_3d_forward
dev = "cuda:0"
L = _native.lib()
torch.manual_seed(0)
def t(fn, n=60):
    for _ in range(30): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for (NT, K, M, H, W) in [(512, 54, 54, 56, 56), (512, 108, 108, 28, 28), (256, 72, 72, 56, 56), (256, 144, 144, 28, 28)]:
    T = 8
    x = torch.randn(NT, K, H, W, device=dev); r = torch.randn(NT, M, H, W, device=dev)
    wt = torch.randn(M, K, device=dev) * 0.1
    sh = torch.empty(3, K, device=dev).uniform_(-1, 1)
    y1 = torch.empty(NT, M, H, W, device=dev); y2 = torch.empty_like(y1)
    st = torch.cuda.current_stream().cuda_stream
    def unfused():
        xs = rubiks_shift_3d_forward(x.view(NT // T, T, K, H, W), sh, 1, 0).view(NT, K, H, W)
        _native.check(L.rk_pw_gemm_f32(wt.data_ptr(), xs.data_ptr(), r.data_ptr(), y1.data_ptr(), NT, K, M, H * W, 1, st), "g")
    def gemm_only():
        _native.check(L.rk_pw_gemm_f32(wt.data_ptr(), x.data_ptr(), r.data_ptr(), y1.data_ptr(), NT, K, M, H * W, 1, st), "g")
    def fused():
        _native.check(L.rk_pw_gemm_shift3d_f32(wt.data_ptr(), x.data_ptr(), sh.data_ptr(), r.data_ptr(), y2.data_ptr(), NT, T, K, M, H, W, st), "f")
    unfused(); fused(); torch.cuda.synchronize()
    print((NT, K, M, H, W), "equal", torch.equal(y1, y2), "shift+gemm %.1f us | gemm alone %.1f us | fused %.1f us" % (t(unfused), t(gemm_only), t(fused)), flush=True)
