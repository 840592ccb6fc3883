#!/usr/bin/env python
"""Which convolutions of a RubiksNet train step still reach aten (MIOpen)?  Prints their shapes."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rubiksnet_amd import RubiksNet, dp
from torch.utils._python_dispatch import TorchDispatchMode


class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if "convolution" in name or "addmm" in name or "mm." in name:
            shapes = [tuple(a.shape) for a in args if torch.is_tensor(a)]
            print(name, shapes, flush=True)
        return func(*args, **(kwargs or {}))


tier = sys.argv[1] if len(sys.argv) > 1 else "tiny"
net = RubiksNet(tier, 174, verbose=False).cuda()
opt = dp.make_optimizer(net, lr=1e-3)
clips = torch.randn(4, 8, 3, 224, 224, device="cuda"); labels = torch.randint(0, 174, (4,), device="cuda")
dp.train_step(net, opt, clips, labels)
with Spy():
    dp.train_step(net, opt, clips, labels)
