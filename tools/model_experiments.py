#!/usr/bin/env python
"""Quick A/B of PyTorch-level settings for the RubiksNet train step (not part of the product)."""
import os, sys, time, contextlib
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rubiksnet_amd import RubiksNet, dp

dev = torch.device("cuda:0")
def run(tag, tier="tiny", variant="rubiks3d", batch=32, amp=None, native_bn=False, steps=6, eval_=False):
    torch.manual_seed(0)
    net = RubiksNet(tier, 174, variant=variant, verbose=False).to(dev)
    if native_bn:
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                orig = m.forward
                def fwd(x, _o=orig):
                    with torch.backends.cudnn.flags(enabled=False):
                        return _o(x)
                m.forward = fwd
    opt = dp.make_optimizer(net, lr=1e-3)
    clips = torch.randn(batch, 8, 3, 224, 224, device=dev); labels = torch.randint(0, 174, (batch,), device=dev)
    def step():
        ctx = torch.autocast("cuda", dtype=amp) if amp is not None else contextlib.nullcontext()
        if eval_:
            with torch.no_grad(), ctx:
                return net(clips)
        with ctx:
            opt.zero_grad(set_to_none=True)
            loss = torch.nn.functional.cross_entropy(net(clips).float(), labels)
        loss.backward(); opt.step(); return loss
    if eval_: net.eval()
    try:
        for _ in range(3): out = step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(steps): out = step()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
        print("%-44s %7.1f ms/step %8.1f clips/s  (finite=%s)" % (tag, dt * 1e3, batch / dt, bool(torch.isfinite(out.float()).all())), flush=True)
    except Exception as e:
        print("%-44s FAILED: %r" % (tag, e), flush=True)

which = sys.argv[1:] or ["base", "nbn", "bf16", "aq", "aqbf16", "large", "eval"]
if "base" in which: run("tiny fp32 train (baseline)")
if "nbn" in which: run("tiny fp32 train, native BN (no MIOpen BN)", native_bn=True)
if "bf16" in which: run("tiny bf16-autocast train", amp=torch.bfloat16)
if "aq" in which: run("tiny-aq fp32 train", variant="rubiks3d-aq")
if "aqbf16" in which: run("tiny-aq bf16-autocast train", variant="rubiks3d-aq", amp=torch.bfloat16)
if "large" in which: run("large fp32 train batch 16", tier="large", batch=16)
if "eval" in which: run("tiny fp32 eval batch 64", batch=64, eval_=True)
if "large32" in which: run("large fp32 train batch 32", tier="large", batch=32)
if "largeaq" in which: run("large-aq bf16-autocast train batch 32", tier="large", variant="rubiks3d-aq", batch=32, amp=torch.bfloat16)
if "largeeval" in which: run("large fp32 eval batch 64", tier="large", batch=64, eval_=True)
