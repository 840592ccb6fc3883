"""AttentionShift: per-channel softmax attention over 3 temporal taps.

Counterpart of rubiksnet/attention_shift.py:6-39.  Same module surface (`n_segment`,
`kernel_size`, frozen temperature parameter `T`, lazily created `weight` [C, 3]) and the same
maths; the difference is where the work runs.  The reference transposes the activation,
inflates the [C,3] weights to [C*H*W,1,3] and calls a grouped conv1d (>= 3 full passes).
Here the tiny [C,3] normalise+softmax stays in PyTorch (so autograd owns it) and the
activation goes once through the HIP 3-tap kernel (rk_tshift3_*, include/rubiks_hip.h).
"""
import contextlib
import struct
import threading

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _native, config

__all__ = ["AttentionShift", "temporal_shift3", "presoftened"]


def _run(name, dev, *args):
    with torch.cuda.device(dev):
        rc = getattr(_native.lib(), name)(*args, torch.cuda.current_stream(dev).cuda_stream)
    _native.check(rc, name)


class _TemporalShift3Func(torch.autograd.Function):
    """y[n,t] = s[c,0] x[n,t-1] + s[c,1] x[n,t] + s[c,2] x[n,t+1], zero padded in t."""

    @staticmethod
    def forward(ctx, x, taps, n_segment):
        assert x.is_cuda, "AttentionShift runs on the HIP device only (no CPU fallback)"
        sfx = _native.dtype_suffix(x.dtype)
        if sfx is None:
            raise ValueError("AttentionShift supports float16/bfloat16/float32/float64, got %s" % x.dtype)
        nt, c, h, w = x.shape
        assert nt % n_segment == 0, "batch*time (%d) is not a multiple of n_segment (%d)" % (nt, n_segment)
        x = x.contiguous()
        # the ABI takes taps in the compute type: fp64 for f64 tensors, fp32 otherwise
        taps32 = taps.detach().to(torch.float64 if x.dtype == torch.float64 else torch.float32).contiguous()
        y = torch.empty_like(x)
        _run("rk_tshift3_forward_" + sfx, x.device, x.data_ptr(), taps32.data_ptr(), y.data_ptr(),
             nt, n_segment, c, h * w)
        ctx.save_for_backward(x, taps32)
        ctx.n_segment = n_segment
        ctx.sfx = sfx
        ctx.taps_dtype = taps.dtype
        return y

    @staticmethod
    def backward(ctx, gy):
        x, taps32 = ctx.saved_tensors
        nt, c, h, w = x.shape
        gy = gy.contiguous()
        gx = torch.empty_like(x)
        gtaps = torch.empty_like(taps32)
        L = _native.lib()
        ws_bytes = int(L.rk_tshift3_backward_workspace_bytes(nt, ctx.n_segment, c, h * w))
        ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=x.device)
        _run("rk_tshift3_backward_" + ctx.sfx, x.device, gy.data_ptr(), x.data_ptr(), taps32.data_ptr(),
             gx.data_ptr(), gtaps.data_ptr(), nt, ctx.n_segment, c, h * w, ws.data_ptr(), ws_bytes)
        return gx, gtaps.to(ctx.taps_dtype), None


class _SoftTapsFunc(torch.autograd.Function):
    """softmax((w / (std(w, dim=1) + 1e-6)) / T, dim=1) of a [C, 3] fp32 weight on the device, one launch each way
    (rk_soft_taps_*): the expression of attention_shift.py:29-30.  T stays a device tensor (no host read)."""

    @staticmethod
    def forward(ctx, weight, temperature):
        w = weight.detach().contiguous()
        t = temperature.detach().to(device=w.device, dtype=torch.float32).reshape(1)
        taps = torch.empty_like(w)
        _run("rk_soft_taps_forward_f32", w.device, w.data_ptr(), t.data_ptr(), taps.data_ptr(), w.shape[0])
        ctx.save_for_backward(w, t, taps)
        return taps

    @staticmethod
    def backward(ctx, gtaps):
        w, t, taps = ctx.saved_tensors
        gtaps = gtaps.contiguous().float()
        gw = torch.empty_like(w)
        _run("rk_soft_taps_backward_f32", w.device, w.data_ptr(), t.data_ptr(), taps.data_ptr(), gtaps.data_ptr(),
             gw.data_ptr(), w.shape[0])
        return gw, None


class _SoftTapsManyFunc(torch.autograd.Function):
    """The taps of EVERY AttentionShift layer of a network in one launch (rk_soft_taps_many_*), and their backward in one
    launch once the last layer's d(taps) has arrived (+ one torch.cat of the 51 small gradients): 2 + 2 launches per train
    step instead of 51 + 51 of ~4 us each between dependent kernels.  Outputs are views of one fresh [sum C, 3] buffer; the
    gradients returned are views of one buffer too (AccumulateGrad takes them as they are)."""

    @staticmethod
    def forward(ctx, plan, *weights):
        jobs, offs, total, max_c = plan
        dev = weights[0].device
        taps = torch.empty(total, 3, dtype=torch.float32, device=dev)
        _run("rk_soft_taps_many_forward_f32", dev, jobs.data_ptr(), len(weights), taps.data_ptr(), max_c)
        ctx.plan = plan
        ctx.dev = dev
        ctx.set_materialize_grads(False)                    # a layer that took no part in the graph: None, not zeros
        ctx.save_for_backward(taps)
        return tuple(taps[o:o + int(w.shape[0])] for w, o in zip(weights, offs))

    @staticmethod
    def backward(ctx, *gtaps):
        jobs, offs, total, max_c = ctx.plan
        taps, = ctx.saved_tensors
        parts = []
        for g, o, o1 in zip(gtaps, offs, list(offs[1:]) + [total]):
            parts.append(torch.zeros(o1 - o, 3, dtype=torch.float32, device=ctx.dev) if g is None else g.float())
        gcat = torch.cat(parts, dim=0).contiguous()
        gw = torch.empty_like(gcat)
        _run("rk_soft_taps_many_backward_f32", ctx.dev, jobs.data_ptr(), len(gtaps), taps.data_ptr(), gcat.data_ptr(),
             gw.data_ptr(), max_c)
        return (None,) + tuple(gw[o:o1] if g is not None else None
                               for g, o, o1 in zip(gtaps, offs, list(offs[1:]) + [total]))


_TABLES = threading.local()        # .presoft: {id(layer): taps} inside a `presoftened` block, per THREAD (cf. pointwise._TABLES)
PRESOFT_GROUP = 16                 # layers per batched node


def _presoft_table():
    return getattr(_TABLES, "presoft", None)


@contextlib.contextmanager
def presoftened(module):
    """Inside the block, `AttentionShift.soft_taps()` of every layer of `module` returns its slice of a batched evaluation
    (see _SoftTapsManyFunc); dp.train_step wraps the forward of a step in it.  The layers are batched in groups of
    PRESOFT_GROUP consecutive layers, one autograd node per group, and a group is evaluated when its FIRST layer asks for
    its taps -- NOT one node for the whole network made up front: a node's backward can run once the d(taps) of all its
    layers have arrived, i.e. after the backward of the earliest block of its group, and the engine takes ready nodes in
    order of creation, latest first -- a node made before the forward would be taken after every block's backward however
    early it became ready.  The gradient of every weight in a node is ready only when the node has run, and
    DistributedDataParallel launches a bucket's all-reduce only when every gradient in it is ready (buckets in order): with
    one up-front node every bucket holding a tap weight -- all of them in an -aq model -- would be exchanged after the
    backward instead of under it.  Made where its first layer runs, a group's node sits in the engine's order right behind
    that layer's own backward, so the last stages' buckets go out while the first stages are still computing
    (RubiksNet-Large-AQ: 51 layers, 4 nodes each way; tests/test_attention_gpu.py checks the order).
    The job tables (weight / temperature pointers, offsets) are cached on the module and rebuilt when a parameter moves.
    Outside a block nothing is shared."""
    layers = [m for m in module.modules() if isinstance(m, AttentionShift) and m.weight is not None and m.weight.is_cuda
              and m.weight.dtype == torch.float32 and m.weight.dim() == 2 and m.weight.shape[1] == 3
              and m.weight.is_contiguous() and m.T.is_cuda and m.T.dtype == torch.float32]
    if (not config.switches().presoft or _presoft_table() is not None or getattr(module, "_is_replica", False) or len(layers) < 2
            or len({m.weight.device for m in layers}) != 1):
        yield
        return
    key = tuple((m.weight.data_ptr(), m.T.data_ptr(), int(m.weight.shape[0])) for m in layers)
    plan = getattr(module, "_rk_presoft_plan", None)
    if plan is None or plan[0] != key or plan[1] != PRESOFT_GROUP:
        groups = []
        for g0 in range(0, len(key), PRESOFT_GROUP):
            recs, offs, off, max_c = [], [], 0, 0
            for wp, tp, c in key[g0:g0 + PRESOFT_GROUP]:
                recs.append(struct.pack("<QQqii", wp, tp, off, c, 0))
                offs.append(off)
                off += c
                max_c = max(max_c, c)
            jobs = torch.frombuffer(bytearray(b"".join(recs)), dtype=torch.uint8).to(layers[0].weight.device)
            groups.append((jobs, tuple(offs), off, max_c))
        plan = (key, PRESOFT_GROUP, tuple(groups))
        module._rk_presoft_plan = plan
    table = {}
    for gi, group in enumerate(plan[2]):
        members = layers[gi * PRESOFT_GROUP:(gi + 1) * PRESOFT_GROUP]
        if len(members) == 1:
            continue                                            # (a lone last layer: its own _SoftTapsFunc node)
        pending = {"plan": group, "members": members, "outs": None}
        table.update((id(m), (pending, i)) for i, m in enumerate(members))
    _TABLES.presoft = table
    try:
        yield
    finally:
        _TABLES.presoft = None


def temporal_shift3(x, taps, n_segment):
    """Functional form: x [N*T, C, H, W], taps [C, 3] (already normalised)."""
    return _TemporalShift3Func.apply(x, taps, n_segment)


class AttentionShift(nn.Module):
    def __init__(self, n_segment, num_channels=None):
        super().__init__()
        self.n_segment = n_segment
        self.kernel_size = 3
        self.T = nn.Parameter(torch.tensor(2.0), requires_grad=False)
        # the reference creates `weight` on the first forward (attention_shift.py:24-27), which
        # is why its -aq model constructor needs a dummy CUDA forward (models.py:100-104);
        # passing num_channels creates it eagerly instead.
        self.weight = None
        if num_channels is not None:
            self.weight = nn.Parameter(torch.rand(num_channels, self.kernel_size))

    def soft_taps(self):
        """softmax((w / (std(w) + 1e-6)) / T) over the 3 taps (attention_shift.py:29-30)."""
        w = self.weight
        table = _presoft_table()
        if table is not None:
            hit = table.get(id(self))
            if hit is not None:
                group, i = hit
                if group["outs"] is None:                       # the group's first layer to run makes the group's node
                    group["outs"] = _SoftTapsManyFunc.apply(group["plan"], *[m.weight for m in group["members"]])
                return group["outs"][i]
        if w.is_cuda and w.dtype == torch.float32 and w.dim() == 2 and w.shape[1] == 3:
            return _SoftTapsFunc.apply(w, self.T)
        weight = w / (torch.std(w, dim=1, keepdim=True) + 1e-6)      # host-side tensors: the same expression in PyTorch
        return F.softmax(weight / self.T, dim=1)

    def forward(self, x):
        return self.attention_shift(x)

    def attention_shift(self, x):
        c = x.size(1)
        if self.weight is None:
            self.weight = nn.Parameter(torch.rand(c, self.kernel_size).to(x.device))
        return temporal_shift3(x, self.soft_taps(), self.n_segment)
