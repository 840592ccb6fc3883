"""python tools/op3d_time.py N T C H W [sH] [quantize]: steady-state us of RubiksShift3D forward / backward (back-to-back
launches over 3 rotating buffer sets, like bench.py's secondary points)."""
import sys
import time

import torch

from rubiksnet_amd.shiftlib.rubiks3d.primitive import rubiks_shift_3d_backward, rubiks_shift_3d_forward

N, T, C, H, W = (int(v) for v in sys.argv[1:6])
s = int(sys.argv[6]) if len(sys.argv) > 6 else 1
q = len(sys.argv) > 7 and sys.argv[7] == "1"
stride = (1, s, s)
torch.manual_seed(0)
sets = []
for _ in range(3):
    x = torch.randn(N, T, C, H, W, device="cuda")
    shift = torch.rand(3, C, device="cuda") * 2 - 1
    y = rubiks_shift_3d_forward(x, shift, stride, 0, quantize=q)
    sets.append((x, shift, torch.randn_like(y)))


def timed(fn, reps):
    for i in range(20):
        fn(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(reps):
        fn(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


fwd = lambda i: rubiks_shift_3d_forward(sets[i % 3][0], sets[i % 3][1], stride, 0, quantize=q)
bwd = lambda i: rubiks_shift_3d_backward(sets[i % 3][2], sets[i % 3][0], sets[i % 3][1], stride, 0, True, quantize=q)
best = lambda fn: min(timed(fn, 400) for _ in range(5))
f, b = best(fwd), best(bwd)
nb = x.numel() * 4
print(f"{(N, T, C, H, W)} stride {stride} q={q}: fwd {f:.1f} us ({(nb + y.numel() * 4) / f / 1e3:.0f} GB/s)  "
      f"bwd {b:.1f} us ({(2 * nb + y.numel() * 4) / b / 1e3:.0f} GB/s)  sum {f + b:.1f}", flush=True)
