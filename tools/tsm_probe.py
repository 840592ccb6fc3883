"""Backward time of the headline tensor with ordinary shifts vs the "tsm" temporal init (integer T shifts)."""
import sys, time, torch
sys.path.insert(0, ".")
from rubiksnet_amd.shiftlib.rubiks3d.primitive import rubiks_shift_3d_backward
dev = "cuda:0"
torch.manual_seed(0)
for shape in [(32, 8, 64, 56, 56), (32, 8, 128, 28, 28), (32, 8, 256, 14, 14)]:
    x = torch.randn(*shape, device=dev); gy = torch.randn_like(x)
    C = shape[2]
    sh = torch.empty(3, C, device=dev).uniform_(-1, 1)
    tsm = sh.clone(); g = C // 8
    tsm[0, :g] = 1; tsm[0, g:2 * g] = -1; tsm[0, 2 * g:] = 0
    for name, s in (("uniform", sh), ("tsm", tsm)):
        for _ in range(20): rubiks_shift_3d_backward(gy, x, s, 1, 0, True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): rubiks_shift_3d_backward(gy, x, s, 1, 0, True)
        torch.cuda.synchronize()
        print(shape, name, f"{(time.perf_counter() - t0) / 50 * 1e6:.1f} us", flush=True)
