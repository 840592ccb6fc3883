"""Shared helpers for the GPU parity tests (test infrastructure)."""
import numpy as np
import torch

DEV = "cuda:0"


def seed_of(*parts):
    """Deterministic seed from a test id (hash() of str is salted per process)."""
    import zlib
    return zlib.crc32(repr(parts).encode())


def rand(rng, shape, dtype):
    return rng.uniform(-1, 1, size=shape).astype(dtype)


def to_dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def to_np(t):
    return t.detach().cpu().numpy()


def special_shifts(rng, D, C, dtype, kind):
    """Shift tables exercising the rare branches."""
    if kind == "generic":
        s = rng.uniform(-1, 1, (D, C))
    elif kind == "wide":
        s = rng.uniform(-3.2, 3.2, (D, C))
    elif kind == "integer":      # exactly-integer shifts in some dims of some channels
        s = rng.uniform(-1.5, 1.5, (D, C))
        mask = rng.random((D, C)) < 0.5
        s = np.where(mask, np.round(s), s)
        s[:, 0] = 0.0            # the all-zero special case of d(x)
    elif kind == "half":         # +-0.5: the quantize tie
        s = rng.choice([-1.5, -0.5, 0.5, 1.5, 0.25], size=(D, C))
    elif kind == "oob":          # everything lands outside
        s = rng.choice([-40.0, 37.5, 100.25], size=(D, C))
    elif kind == "tiny":         # around the 2-D operator's 1e-7 "is an integer" tolerance
        s = rng.choice([1e-8, -1e-8, 3e-8, -3e-8, 8e-8, -8e-8, 1e-7, 1.2e-7, -1.2e-7, 1.0 + 1.2e-7, 0.3], size=(D, C))
    else:
        raise ValueError(kind)
    return s.astype(dtype)
