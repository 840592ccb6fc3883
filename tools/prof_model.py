#!/usr/bin/env python
"""RubiksNet train-step driver for profiling: a few steps of fwd+bwd+Adam on synthetic clips."""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rubiksnet_amd import RubiksNet, dp

ap = argparse.ArgumentParser()
ap.add_argument("--tier", default="tiny"); ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--steps", type=int, default=5); ap.add_argument("--variant", default="rubiks3d")
ap.add_argument("--amp", default="none"); ap.add_argument("--channels-last", action="store_true")
ap.add_argument("--eval", action="store_true")
a = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = RubiksNet(a.tier, 174, variant=a.variant, verbose=False).to(dev)
opt = dp.make_optimizer(net, lr=1e-3)
clips = torch.randn(a.batch, 8, 3, 224, 224, device=dev); labels = torch.randint(0, 174, (a.batch,), device=dev)
amp = {"none": None, "bf16": torch.bfloat16, "fp16": torch.float16}[a.amp]
def step():
    if a.eval:
        with torch.no_grad():
            return net(clips)
    with torch.autocast("cuda", dtype=amp, enabled=amp is not None):
        return dp.train_step(net, opt, clips, labels)
if a.eval: net.eval()
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(a.steps): step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.steps
print("%s %s batch %d: %.1f ms/step, %.1f clips/s" % (a.tier, "eval" if a.eval else "train", a.batch, dt * 1e3, a.batch / dt))
if os.environ.get("RK_TORCH_PROF"):
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        for _ in range(3): step()
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=35, max_name_column_width=70))
