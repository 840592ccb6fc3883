"""Generates tests/golden/model_structure.json by IMPORTING the reference's Python
model code on CPU -- dev container only.

    python tests/golden/gen_model_structure_golden.py

The reference's `rubiksnet.shiftlib` does `import rubiksnet_cuda` (a CUDA extension
that cannot exist here), so an EMPTY placeholder module of that name is put into
sys.modules purely so that the import statement succeeds.  No kernel is ever
called: only constructors run, and what is recorded is structure that the
reference's pure-Python code determines -- state_dict key names and shapes,
parameter counts (cross-checked against README.md:87-91: 1.9M/3.6M/6.2M/8.5M), and
the per-layer (channels, stride, padding) list of the RubiksShift3D modules.
"""
import json
import os
import sys
import types

import torch

REF = os.environ.get("RUBIKS_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    stub = types.ModuleType("rubiksnet_cuda")
    for fn in ("rubiks2d_forward", "rubiks2d_backward",
               "rubiks_shift_3d_forward_float", "rubiks_shift_3d_forward_double",
               "rubiks_shift_3d_backward_float", "rubiks_shift_3d_backward_double"):
        setattr(stub, fn, None)          # names looked up at import time (primitive.py:222-231)
    sys.modules["rubiksnet_cuda"] = stub
    sys.path.insert(0, REF)
    from rubiksnet.models import RubiksNet
    from rubiksnet.shiftlib import RubiksShift3D

    out = {}
    for tier in ("tiny", "small", "medium", "large"):
        torch.manual_seed(0)
        net = RubiksNet(tier=tier, num_classes=174, num_frames=8, variant="rubiks3d")
        sd = net.state_dict()
        shifts = [
            [m.num_channels, list(m.stride), list(m.padding)]
            for m in net.modules() if isinstance(m, RubiksShift3D)
        ]
        out[tier] = {
            "num_params": sum(p.numel() for p in net.parameters()),
            "state_dict": [[k, list(v.shape)] for k, v in sd.items()],
            "shift3d_layers": shifts,
            "feature_dim": net.feature_dim,
        }
        print(tier, out[tier]["num_params"], len(sd), len(shifts))
    with open(os.path.join(HERE, "model_structure.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))


if __name__ == "__main__":
    main()
