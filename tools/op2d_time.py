"""python tools/op2d_time.py F C H W [stride] [dtype]: steady-state us of RubiksShift2D forward / backward (3 rotating sets)."""
import sys
import time

import torch

from rubiksnet_amd import rubiksnet_cuda

F, C, H, W = (int(v) for v in sys.argv[1:5])
s = int(sys.argv[5]) if len(sys.argv) > 5 else 1
dt = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[sys.argv[6] if len(sys.argv) > 6 else "bf16"]
torch.manual_seed(0)
Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
shift = (torch.rand(2, C, device="cuda") * 1.9 - 0.95).to(dt)
shift[(shift.float() - shift.float().round()).abs() < 1e-3] = 0.37
sets = [(torch.empty(F, C, H, W, device="cuda", dtype=dt).uniform_(-1, 1), torch.empty(F, C, Ho, Wo, device="cuda", dtype=dt).uniform_(-1, 1),
         torch.empty(F, C, Ho, Wo, device="cuda", dtype=dt), torch.empty(F, C, H, W, device="cuda", dtype=dt)) for _ in range(3)]
gs = torch.empty_like(shift)
f = lambda i: rubiksnet_cuda.rubiks2d_forward(sets[i % 3][0], shift, [s, s], [0, 0], False, sets[i % 3][2])
b = lambda i: rubiksnet_cuda.rubiks2d_backward(sets[i % 3][1], sets[i % 3][0], shift, [s, s], [0, 0], True, True, False, sets[i % 3][3], gs)
out = []
for fn in (f, b):
    best = 1e9
    for _ in range(3):
        for i in range(50):
            fn(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(200):
            fn(i)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 200 * 1e6)
    out.append(best)
es = sets[0][0].element_size()
nx, ny = F * C * H * W * es, F * C * Ho * Wo * es
print(f"[{F},{C},{H},{W}] s{s} {dt}: fwd {out[0]:.1f} us ({(nx + ny) / out[0] / 1e3:.0f} GB/s)  bwd {out[1]:.1f} us ({(2 * nx + ny) / out[1] / 1e3:.0f} GB/s)", flush=True)
