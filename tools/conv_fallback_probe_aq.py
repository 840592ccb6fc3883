#!/usr/bin/env python
"""Which convolutions of a RubiksNet-Large-AQ bf16 train step still reach aten (MIOpen)?"""
import os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rubiksnet_amd import RubiksNet, dp
from torch.utils._python_dispatch import TorchDispatchMode

seen = collections.Counter()
class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if "convolution" in name:
            seen[(name, tuple(tuple(a.shape) for a in args if torch.is_tensor(a)), str(args[0].dtype))] += 1
        return func(*args, **(kwargs or {}))

tier = sys.argv[1] if len(sys.argv) > 1 else "large"
net = RubiksNet(tier, 174, variant="rubiks3d-aq", verbose=False).cuda()
opt = dp.make_optimizer(net, lr=1e-3)
clips = torch.randn(4, 8, 3, 224, 224, device="cuda"); labels = torch.randint(0, 174, (4,), device="cuda")
def step():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        dp.train_step(net, opt, clips, labels)
step()
with Spy():
    step()
for k, v in sorted(seen.items()): print(v, k)
