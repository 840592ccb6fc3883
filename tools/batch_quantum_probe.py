"""Does the backward's time per clip depend on how the workgroup count divides into resident rounds?
(32, 8, 64, 56, 56): 2048 workgroups over 768 resident slots = 2.67 rounds."""
import sys, time, torch
sys.path.insert(0, ".")
from rubiksnet_amd.shiftlib.rubiks3d.primitive import rubiks_shift_3d_backward, rubiks_shift_3d_forward
dev = "cuda:0"
torch.manual_seed(0)
C = 64
sh = torch.empty(3, C, device=dev).uniform_(-1, 1)
for N in (12, 24, 30, 32, 36, 48, 60, 64):
    xs = [torch.randn(N, 8, C, 56, 56, device=dev) for _ in range(3)]
    gs = [torch.randn_like(xs[0]) for _ in range(3)]
    for name, fn in (("bwd", lambda i: rubiks_shift_3d_backward(gs[i], xs[i], sh, 1, 0, True)),
                     ("fwd", lambda i: rubiks_shift_3d_forward(xs[i], sh, 1, 0))):
        for k in range(300): fn(k % 3)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for k in range(200): fn(k % 3)
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / 200 * 1e6
        nbytes = xs[0].numel() * 4 * (3 if name == "bwd" else 2)
        print(f"N={N:3d} {name} {us:7.1f} us  {us / N:6.3f} us/clip  {nbytes / us / 1e3:7.1f} GB/s  wgs={N * C}", flush=True)
    del xs, gs
