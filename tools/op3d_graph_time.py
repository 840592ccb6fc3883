"""python tools/op3d_graph_time.py N T C H W [sH]: GPU-side us per launch of RubiksShift3D forward / backward, measured by
replaying a captured hipGraph of 30 back-to-back launches over 3 rotating buffer sets (no Python / launch overhead in the
number: tools/op3d_time.py issues one launch per Python call and bottoms out at ~16 us per call on small kernels)."""
import sys

import torch

from rubiksnet_amd import rubiksnet_cuda

N, T, C, H, W = (int(v) for v in sys.argv[1:6])
s = int(sys.argv[6]) if len(sys.argv) > 6 else 1
stride, p0 = [1, s, s], [0, 0, 0]
Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
dev = torch.device("cuda:0")
torch.manual_seed(0)
shift = torch.rand(3, C, device=dev) * 2 - 1
sets = [(torch.randn(N, T, C, H, W, device=dev), torch.randn(N, T, C, Ho, Wo, device=dev),
         torch.empty(N, T, C, Ho, Wo, device=dev), torch.empty(N, T, C, H, W, device=dev)) for _ in range(3)]
gs = torch.empty(3, C, device=dev)
K = 30


def fwd(i):
    x, gy, y, gx = sets[i % 3]
    rubiksnet_cuda.rubiks_shift_3d_forward_float(x, shift, stride, p0, False, y)


def bwd(i):
    x, gy, y, gx = sets[i % 3]
    rubiksnet_cuda.rubiks_shift_3d_backward_float(x, shift, gy, stride, p0, gx, gs, True, 1.0, False)


def graph_time(fn):
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for i in range(6):
            fn(i)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for i in range(K):
                fn(i)
    best = 1e9
    for _ in range(5):
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            g.replay()
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / (10 * K) * 1e3)
    return best


f, b = graph_time(fwd), graph_time(bwd)
nin, nout = N * T * C * H * W * 4, N * T * C * Ho * Wo * 4
print(f"{(N, T, C, H, W)} stride {tuple(stride)}: fwd {f:.1f} us ({(nin + nout) / f / 1e3:.0f} GB/s)  "
      f"bwd {b:.1f} us ({(2 * nin + nout) / b / 1e3:.0f} GB/s)  sum {f + b:.1f}  "
      f"frac {(3 * nin + 2 * nout) / (f + b) / 1e3 / 8000:.3f}", flush=True)
