"""H2D rate of the feeder's copy: pinned uint8 [32,224,224,24] -> device, current stream vs side stream, one vs two in flight."""
import os, sys, time, torch
dev = torch.device("cuda:0")
h = [torch.randint(0, 256, (32, 224, 224, 24), dtype=torch.uint8).pin_memory() for _ in range(2)]
d = [torch.empty_like(x, device=dev) for x in h]
nb = h[0].numel()
def rate(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return n * nb / (time.perf_counter() - t0) / 1e9
side = torch.cuda.Stream(dev)
def cur(): d[0].copy_(h[0], non_blocking=True)
def on_side():
    with torch.cuda.stream(side): d[0].copy_(h[0], non_blocking=True)
def two():
    with torch.cuda.stream(side): d[0].copy_(h[0], non_blocking=True)
    d[1].copy_(h[1], non_blocking=True)
print("HSA_ENABLE_SDMA=%s" % os.environ.get("HSA_ENABLE_SDMA"))
print("current stream: %.1f GB/s" % rate(cur))
print("side stream:    %.1f GB/s" % rate(on_side))
print("two streams:    %.1f GB/s (sum)" % (2 * rate(two)))
sys.path.insert(0, ".")
from rubiksnet_amd.input_pipeline import SyntheticClipLoader
it = iter(SyntheticClipLoader(batch=32, device=dev))
for _ in range(3): next(it)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): next(it)
torch.cuda.synchronize(); print("loader: %.0f clips/s" % (20 * 32 / (time.perf_counter() - t0)))
