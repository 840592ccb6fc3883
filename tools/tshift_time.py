"""python tools/tshift_time.py [dtype]: steady-state us of the temporal 3-tap kernels (AttentionShift's device half) on the
layer shapes of RubiksNet-Large-AQ at batch 32 (NT = 256, n_segment 8), through the C ABI, 3 rotating buffer sets."""
import sys
import time

import torch

from rubiksnet_amd import _native

dt = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[sys.argv[1] if len(sys.argv) > 1 else "bf16"]
sfx = _native.dtype_suffix(dt)
L = _native.lib()
stream = torch.cuda.current_stream().cuda_stream
for C, H in ((72, 56), (144, 28), (288, 14), (576, 7)):
    NT, S, HW = 256, 8, H * H
    sets = [(torch.randn(NT, C, H, H, device="cuda").to(dt), torch.randn(NT, C, H, H, device="cuda").to(dt),
             torch.empty(NT, C, H, H, device="cuda", dtype=dt)) for _ in range(3)]
    taps = torch.softmax(torch.randn(C, 3, device="cuda"), 1).contiguous()
    gtaps = torch.empty_like(taps)
    wsb = int(L.rk_tshift3_backward_workspace_bytes(NT, S, C, HW))
    ws = torch.empty(max(wsb, 1), dtype=torch.uint8, device="cuda")

    def fwd(i):
        x, _, y = sets[i % 3]
        _native.check(getattr(L, "rk_tshift3_forward_" + sfx)(x.data_ptr(), taps.data_ptr(), y.data_ptr(), NT, S, C, HW, stream), "f")

    def bwd(i):
        x, g, y = sets[i % 3]
        _native.check(getattr(L, "rk_tshift3_backward_" + sfx)(g.data_ptr(), x.data_ptr(), taps.data_ptr(), y.data_ptr(),
                                                             gtaps.data_ptr(), NT, S, C, HW, ws.data_ptr(), wsb, stream), "b")

    def timed(fn, reps=300):
        for i in range(20):
            fn(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(reps):
            fn(i)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e6

    f, b = min(timed(fwd) for _ in range(4)), min(timed(bwd) for _ in range(4))
    nb = NT * C * HW * sets[0][0].element_size()
    print(f"[{NT},{C},{H},{H}] {sfx}: fwd {f:.1f} us ({2 * nb / f / 1e3:.0f} GB/s)  bwd {b:.1f} us ({3 * nb / b / 1e3:.0f} GB/s)", flush=True)
