import sys, time, torch
sys.path.insert(0, ".")
from rubiksnet_amd import rubiksnet_cuda
dev = "cuda:0"
torch.manual_seed(0)
def bench(shape):
    NT, C, H, W = shape
    sets = []
    for _ in range(3):
        x = torch.empty(shape, device=dev, dtype=torch.bfloat16).uniform_(-1, 1)
        gy = torch.empty_like(x).uniform_(-1, 1)
        sets.append((x, gy, torch.empty_like(x), torch.empty_like(x)))
    shift = (torch.rand(2, C, device=dev) * 1.9 - 0.95).to(torch.bfloat16)
    shift[(shift.float() - shift.float().round()).abs() < 1e-3] = 0.37
    gs = torch.empty_like(shift)
    f = lambda i: rubiksnet_cuda.rubiks2d_forward(sets[i % 3][0], shift, [1, 1], [0, 0], False, sets[i % 3][2])
    b = lambda i: rubiksnet_cuda.rubiks2d_backward(sets[i % 3][1], sets[i % 3][0], shift, [1, 1], [0, 0], True, True, False, sets[i % 3][3], gs)
    out = []
    for fn in (f, b):
        for i in range(400): fn(i)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(300): fn(i)
        torch.cuda.synchronize(); out.append((time.perf_counter() - t0) / 300 * 1e6)
    n = x.numel() * 2
    print(f"{shape} fwd {out[0]:6.1f} us {2*n/out[0]/1e3:6.0f} GB/s | bwd {out[1]:6.1f} us {3*n/out[1]/1e3:6.0f} GB/s | frac {(5*n/(out[0]+out[1])/1e3)/8000:.3f}", flush=True)
bench((256, 64, 56, 56))
bench((256, 54, 112, 112))
