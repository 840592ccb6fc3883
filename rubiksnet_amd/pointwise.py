"""1x1 convolutions of the backbone through librubiks_hip's NCHW MFMA GEMM (rk_pw_gemm_f32), SURVEY 8(f) f1.

`conv1x1(conv, x)` evaluates an ordinary bias-free `nn.Conv2d(kernel_size=1, stride=1)` module -- the module,
its weight, its state-dict key stay what they are.  Activations may be fp32 or bf16 (autocast: the fp32 weight is
used as it is and d(weight) comes out in fp32 -- no casts; arithmetic is fp32 MFMA either way).  For GPU tensors
with H*W % 4 == 0 and even channel
counts, d(weight) always runs on the HIP kernels (fp32: 2.6x / 2.3x / 2.0x / 1.25x MIOpen's at 56x56 / 54->108 /
28x28 / 14x14; bf16, on the bf16 MFMA: 2.4x / 3.2x / 3.9x / 1.7x: MIOpen's NHWC implicit GEMM needs two layout
transposes), and forward / d(input) do where they win -- the
memory-bound 112x112 / 56x56 stages; elsewhere they stay on aten (MIOpen).  Other dtypes, 7x7 planes, strided
shortcuts, CPU tensors: `conv(x)`.  `RK_PW=0` disables the HIP path, `RK_PW=all` forces the HIP GEMM wherever
the kernel's constraints allow (config.py).  Forward, d(input) and d(weight) are HIP MFMA kernels.
"""
import contextlib
import struct
import threading

import torch

from . import _native, config

_S2_CMAX = 320              # 1x1 / stride-2 shortcuts (fp32)
_FUSED_EVAL_CMAX = 320      # inference fusion: channel limit of rk_pw_gemm_fused_f32's register tile
_FUSED_EVAL_PMIN = 196      # ... and smallest plane it pays for

__all__ = ["conv1x1", "stem_conv", "pointwise_mode", "fused_eval_block", "all_frozen"]


def pointwise_mode():
    return config.switches().pointwise


_SFX = {torch.float32: "f32", torch.bfloat16: "bf16"}      # storage of the activations; weights are fp32


# Per-THREAD tables of the `prepacked` / `prefolded` blocks (nn.DataParallel runs one forward thread per replica: a table
# shared between them would be torn down by the first thread to leave its block while another is between its `is not None`
# test and its lookup).  .prepacked: {weight.data_ptr(): (weight shape, fwd image, bwd image)}; .prefolded: {id(bn): (a, b)}.
_TABLES = threading.local()


def _table(name):
    return getattr(_TABLES, name, None)


@contextlib.contextmanager
def prepacked(module):
    """Every 1x1 weight of `module` packed for the bf16 GEMMs in ONE launch (rk_pw_pack_many_bf16), valid inside the `with`
    block: dp.train_step wraps forward + backward of a bf16-autocast step in it (the optimizer -- the only writer of the weights
    -- runs after the block), so `_pack` finds its operands instead of launching k_pw16_pack per layer and forward (x 100 in
    Large-AQ: 0.5 ms of the step).  Outside a block nothing is cached: no key would see an edit made through `.data`.
    The images live in one fresh buffer per block; the d(input) operands saved for backward are views of it."""
    if _table("prepacked") is not None or not config.switches().prepack or pointwise_mode() == "0":
        yield
        return
    weights = [m.weight for m in module.modules()
               if isinstance(m, torch.nn.Conv2d) and m.kernel_size == (1, 1) and m.groups == 1 and m.weight.is_cuda
               and m.weight.dtype == torch.float32 and m.weight.is_contiguous()]
    if not weights or len(weights) > 65535 or len({w.device for w in weights}) != 1:
        yield
        return
    L = _native.lib()
    dev = weights[0].device
    key = tuple((w.data_ptr(), w.shape[0], w.shape[1]) for w in weights)
    plan = getattr(module, "_rk_prepack_plan", None)
    if plan is None or plan[0] != key:
        recs, off, max_units, views = [], 0, 0, []
        for w in weights:
            Cout, Cin = int(w.shape[0]), int(w.shape[1])
            bf, bb = int(L.rk_pw_packed_bytes(Cout, Cin)), int(L.rk_pw_packed_bytes(Cin, Cout))
            recs.append(struct.pack("<Qqqiiii", w.data_ptr(), off, off + bf, Cout, Cin, bf // 16, bb // 16))
            views.append((off, bf, bb))
            max_units = max(max_units, (bf + bb) // 16)
            off += bf + bb
        jobs = torch.frombuffer(bytearray(b"".join(recs)), dtype=torch.uint8).to(dev)
        plan = (key, jobs, off, max_units, views)
        module._rk_prepack_plan = plan                      # (rebuilt when a weight moves: .to(), a reloaded parameter)
    _, jobs, total, max_units, views = plan
    with torch.cuda.device(dev):
        buf = torch.empty(total, dtype=torch.uint8, device=dev)
        _native.check(L.rk_pw_pack_many_bf16(jobs.data_ptr(), len(weights), buf.data_ptr(), max_units,
                                             torch.cuda.current_stream(dev).cuda_stream), "rk_pw_pack_many_bf16")
    _TABLES.prepacked = {w.data_ptr(): (tuple(w.shape[:2]), buf[o:o + bf], buf[o + bf:o + bf + bb])
                         for w, (o, bf, bb) in zip(weights, views)}
    try:
        yield
    finally:
        _TABLES.prepacked = None


def _pack(weight):
    """The weight of a 1x1 convolution packed for rk_pw_gemm_packed_bf16 (rk_pw16.hip): both operands -- W for the forward,
    W^T for d(input) -- in ONE small launch.  Redone on every forward (the d(input) operand rides to the backward in the
    autograd context): no cache key sees in-place edits made through `.data` (cf. _bn_affine) -- except inside
    `prepacked(module)`, whose images of this step's weights are used when present."""
    table = _table("prepacked")
    if table is not None:
        hit = table.get(weight.data_ptr())
        if hit is not None and hit[0] == tuple(weight.shape[:2]):
            return hit[1], hit[2]
    L = _native.lib()
    Cout, Cin = weight.shape[0], weight.shape[1]
    dev = weight.device
    with torch.cuda.device(dev):
        fwd = torch.empty(int(L.rk_pw_packed_bytes(Cout, Cin)), dtype=torch.uint8, device=dev)
        bwd = torch.empty(int(L.rk_pw_packed_bytes(Cin, Cout)), dtype=torch.uint8, device=dev)
        _native.check(L.rk_pw_pack_bf16(weight.data_ptr(), Cout, Cin, fwd.data_ptr(), bwd.data_ptr(),
                                        torch.cuda.current_stream(dev).cuda_stream), "rk_pw_pack_bf16")
    return fwd, bwd


_PW16_BYTES = 1 << 31          # rk_pw16.hip addresses an activation tensor with 32-bit byte offsets


def _pw16_fits(Fr, C, P):
    return Fr * C * P * 2 < _PW16_BYTES


def _packed_ok(a, x, K, M, P, a_is_mk):
    # (tensors of 2 GiB and more -- ~150 clips of 8 x 72 x 112 x 112 in bf16 -- keep the first-generation kernel, which
    # indexes with size_t)
    # (max(K, M): the residual [F, M, P] goes through the same 32-bit-offset DMA as X since round 5)
    return (x.dtype == torch.bfloat16 and P >= 8 and a.dim() >= 2 and a.shape[0] == (M if a_is_mk else K)
            and _pw16_fits(x.shape[0], max(K, M), P))


def _gemm(a, x, out, Fr, K, M, P, a_is_mk, residual=None, packed=None):
    dev = x.device
    if _packed_ok(a, x, K, M, P, a_is_mk):
        # bf16 activations: the packed-weight kernel (`packed`: the operand, packed by the caller's forward)
        if packed is None:
            packed = _pack(a)[0 if a_is_mk else 1]
        with torch.cuda.device(dev):
            rc = _native.lib().rk_pw_gemm_packed_bf16(
                packed.data_ptr(), x.data_ptr(), residual.data_ptr() if residual is not None else None,
                out.data_ptr(), Fr, K, M, P, torch.cuda.current_stream(dev).cuda_stream)
        _native.check(rc, "rk_pw_gemm_packed_bf16")
        return out
    fn = getattr(_native.lib(), "rk_pw_gemm_" + _SFX[x.dtype])
    with torch.cuda.device(dev):
        rc = fn(a.data_ptr(), x.data_ptr(), residual.data_ptr() if residual is not None else None, out.data_ptr(),
                Fr, K, M, P, int(a_is_mk), torch.cuda.current_stream(dev).cuda_stream)
    _native.check(rc, "rk_pw_gemm")
    return out


def _as(weight, dtype):
    return weight if weight.dtype == dtype else weight.to(dtype)


def _wgrad(dy, x, weight):
    # the HIP kernels win on every probed shape: fp32 113 vs 295 us, bf16 (bf16 MFMA) 55 vs 133 us at
    # [256,54->54,56x56] -- MIOpen's d(weight) needs two layout transposes
    Fr, Cin, H, W = x.shape
    Cout = weight.shape[0]
    dev = x.device
    L = _native.lib()
    dw = torch.empty_like(weight)                       # fp32, whatever the activations' storage type
    with torch.cuda.device(dev):
        if x.dtype == torch.bfloat16 and H * W >= 8 and _pw16_fits(Fr, max(Cin, Cout), H * W):
            nbytes = int(L.rk_pw_wgrad16_workspace_bytes(Fr, Cin, Cout, H * W))
            ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=dev)
            _native.check(L.rk_pw_wgrad16_bf16(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), Fr, Cin, Cout, H * W, ws.data_ptr(),
                                               nbytes, torch.cuda.current_stream(dev).cuda_stream), "rk_pw_wgrad16_bf16")
            return dw
        nbytes = int(L.rk_pw_wgrad_workspace_bytes(Fr, Cin, Cout, H * W))
        ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=dev)
        rc = getattr(L, "rk_pw_wgrad_" + _SFX[x.dtype])(
            dy.data_ptr(), x.data_ptr(), dw.data_ptr(), Fr, Cin, Cout, H * W, ws.data_ptr(), nbytes,
            torch.cuda.current_stream(dev).cuda_stream)
    _native.check(rc, "rk_pw_wgrad")
    return dw


_ATEN_ARGS = ([1, 1], [0, 0], [1, 1], False, [0, 0], 1)


class _Conv1x1Func(torch.autograd.Function):
    """1x1 convolution; `hip_gemm` selects the HIP GEMM for forward / d(input) (else aten = MIOpen);
    d(weight) is the HIP kernel either way (it wins on every shape: MIOpen's needs two layout transposes)."""

    @staticmethod
    def forward(ctx, x, weight, hip_gemm, residual, hip_dx=None, want_stats=False):
        Fr, Cin, H, W = x.shape
        Cout = weight.shape[0]
        ctx.packed_bwd = None
        stats = None
        if hip_gemm:
            y = torch.empty(Fr, Cout, H, W, dtype=x.dtype, device=x.device)
            fwd = None
            if _packed_ok(weight, x, Cin, Cout, H * W, True):
                fwd, ctx.packed_bwd = _pack(weight)
            if want_stats and fwd is not None and residual is None:
                # (conv2 -> bn2 only: with the residual's 72 registers on top of 144 accumulators the statistics variant of
                # the 288-row kernel spills, and the 144-row one drops from 3 to 2 waves per SIMD: measured slower than the
                # statistics pass it saves)
                # training: the GEMM also leaves the tile statistics of y for the BatchNorm that consumes it (fused_bn.py)
                L = _native.lib()
                J = int(L.rk_pw16_stat_tiles(Fr, H * W))
                stats = torch.empty(Cout, J, 4, dtype=torch.float32, device=x.device)
                with torch.cuda.device(x.device):
                    _native.check(L.rk_pw_gemm_packed_stats_bf16(
                        fwd.data_ptr(), x.data_ptr(), residual.data_ptr() if residual is not None else None, y.data_ptr(), Fr, Cin,
                        Cout, H * W, stats.data_ptr(), J, torch.cuda.current_stream(x.device).cuda_stream),
                        "rk_pw_gemm_packed_stats_bf16")
            else:
                _gemm(weight, x, y, Fr, Cin, Cout, H * W, True, residual, fwd)  # `+ residual` in the GEMM's epilogue
        else:
            y = torch.ops.aten.convolution(x, _as(weight, x.dtype), None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1)
            if residual is not None:
                y.add_(residual)
        ctx.save_for_backward(x, weight)
        ctx.hip_dx = hip_gemm if hip_dx is None else hip_dx
        ctx.has_residual = residual is not None
        if want_stats:
            if stats is None:
                stats = torch.empty(0, device=x.device)
            ctx.mark_non_differentiable(stats)
            ctx.set_materialize_grads(False)                # (no zero-filled `_dstats` per backward)
            return y, stats
        return y

    @staticmethod
    def backward(ctx, dy, _dstats=None):
        x, weight = ctx.saved_tensors
        if dy is None:
            return (None,) * 6
        dy = dy.contiguous()
        if dy.dtype != x.dtype:
            dy = dy.to(x.dtype)
        dx = dw = None
        if ctx.needs_input_grad[0]:
            if ctx.hip_dx:
                Fr, Cin, H, W = x.shape
                dx = torch.empty_like(x)
                _gemm(weight, dy, dx, Fr, weight.shape[0], Cin, H * W, False, None, ctx.packed_bwd)     # W read as [K=Cout][M=Cin]
            else:
                dx = torch.ops.aten.convolution_backward(dy, x, _as(weight, x.dtype), None, *_ATEN_ARGS,
                                                         [True, False, False])[0]
        if ctx.needs_input_grad[1]:
            dw = _wgrad(dy, x, weight)
        return dx, dw, None, (dy if ctx.has_residual and ctx.needs_input_grad[3] else None), None, None


def _eligible(conv, x, has_residual=False):
    """None: stock path; else whether forward / d(input) should use the HIP GEMM too."""
    mode = pointwise_mode()
    if mode == "0" or not (x.is_cuda and x.dtype in _SFX and x.dim() == 4):
        return None
    if x.dtype == torch.float32 and torch.is_autocast_enabled():
        return None                                       # autocast would run this layer in bf16: let it
    if not (isinstance(conv, torch.nn.Conv2d) and conv.kernel_size == (1, 1) and conv.stride == (1, 1)
            and conv.padding == (0, 0) and conv.groups == 1 and conv.bias is None
            and conv.weight.dtype == torch.float32):
        return None
    P = x.shape[2] * x.shape[3]
    K, M = conv.in_channels, conv.out_channels
    if P % 4 or K % 2 or M % 2 or x.numel() == 0:      # kernel constraints (M even: it is K of the d(input) GEMM)
        return None
    # Round 1 restricted "auto" to a per-plane-size channel window measured with isolated, event-bracketed probes
    # (tools/pointwise_probe.py).  Measured on the train steps themselves the HIP GEMM wins wherever it can run
    # (RK_PW=all against that window: Large-AQ bf16 49.3 -> 45.0 ms, Large 74.4 -> 73.5, Small 44.9 -> 44.3,
    # Tiny 25.8 -> 25.6, Tiny forward b64 equal), so "auto" and "all" now mean the same.
    return True


class _ConvS2Func(torch.autograd.Function):
    """1x1 / stride-2 convolution (the projecting shortcut of a downsampling block) on the HIP GEMM kernels:
    forward reads the activation at stride 2, d(input) scatters into the even positions (zeros elsewhere, every
    element written), d(weight) gathers like the forward.  MIOpen runs these as two asm kernels + an implicit-GEMM
    d(weight) between layout transposes (1.3 ms + of a Tiny train step)."""

    @staticmethod
    def forward(ctx, x, weight):
        Fr, Cin, H, W = x.shape
        Cout = weight.shape[0]
        y = torch.empty(Fr, Cout, H // 2, W // 2, dtype=x.dtype, device=x.device)
        dev = x.device
        with torch.cuda.device(dev):
            rc = _native.lib().rk_pw_s2_forward_f32(weight.data_ptr(), x.data_ptr(), y.data_ptr(), Fr, Cin, Cout, H, W,
                                                    torch.cuda.current_stream(dev).cuda_stream)
        _native.check(rc, "rk_pw_s2_forward_f32")
        ctx.save_for_backward(x, weight)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = dy.contiguous()
        Fr, Cin, H, W = x.shape
        Cout = weight.shape[0]
        dev = x.device
        L = _native.lib()
        dx = dw = None
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            if ctx.needs_input_grad[0]:
                dx = torch.empty_like(x)
                _native.check(L.rk_pw_s2_dgrad_f32(weight.data_ptr(), dy.data_ptr(), dx.data_ptr(), Fr, Cin, Cout, H, W,
                                                   stream), "rk_pw_s2_dgrad_f32")
            if ctx.needs_input_grad[1]:
                dw = torch.empty_like(weight)
                nbytes = int(L.rk_pw_wgrad_workspace_bytes(Fr, Cin, Cout, (H // 2) * (W // 2)))
                ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=dev)
                _native.check(L.rk_pw_s2_wgrad_f32(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), Fr, Cin, Cout, H, W,
                                                   ws.data_ptr(), nbytes, stream), "rk_pw_s2_wgrad_f32")
        return dx, dw


def _eligible_s2(conv, x):
    return (pointwise_mode() != "0" and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.numel() > 0
            and not torch.is_autocast_enabled()
            and isinstance(conv, torch.nn.Conv2d) and conv.kernel_size == (1, 1) and conv.stride == (2, 2)
            and conv.padding == (0, 0) and conv.groups == 1 and conv.bias is None and conv.weight.dtype == torch.float32
            and conv.in_channels % 2 == 0 and conv.out_channels % 2 == 0
            and max(conv.in_channels, conv.out_channels) <= _S2_CMAX
            and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0 and ((x.shape[2] // 2) * (x.shape[3] // 2)) % 4 == 0)


class _ConvS2Bf16Func(torch.autograd.Function):
    """The same projecting shortcut for bf16 activations (autocast): the even pixels are gathered once (a strided copy: a
    quarter of the elements), and the three GEMMs run on the stride-1 bf16 kernels of rk_pw16.hip on that quarter-size
    tensor -- which is also what is saved for backward.  d(input) is scattered back into zeros.  MIOpen ran these between
    NCHW <-> NHWC transposes of the FULL-size tensors: 0.5 / 0.8 ms forward / backward at [256, 72, 112, 112]."""

    @staticmethod
    def forward(ctx, x, weight):
        Fr, Cin, H, W = x.shape
        Cout = weight.shape[0]
        xs = x[:, :, ::2, ::2].contiguous()
        y = torch.empty(Fr, Cout, H // 2, W // 2, dtype=x.dtype, device=x.device)
        fwd, ctx.packed_bwd = _pack(weight)
        _gemm(weight, xs, y, Fr, Cin, Cout, (H // 2) * (W // 2), True, None, fwd)
        ctx.save_for_backward(xs, weight)
        ctx.full = (H, W)
        return y

    @staticmethod
    def backward(ctx, dy):
        xs, weight = ctx.saved_tensors
        dy = dy.contiguous()
        if dy.dtype != xs.dtype:
            dy = dy.to(xs.dtype)
        Fr, Cin, Ho, Wo = xs.shape
        H, W = ctx.full
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dxs = torch.empty_like(xs)
            _gemm(weight, dy, dxs, Fr, weight.shape[0], Cin, Ho * Wo, False, None, ctx.packed_bwd)
            dx = torch.zeros(Fr, Cin, H, W, dtype=xs.dtype, device=xs.device)
            dx[:, :, ::2, ::2] = dxs
        if ctx.needs_input_grad[1]:
            dw = _wgrad(dy, xs, weight)
        return dx, dw


def _eligible_s2_bf16(conv, x):
    return (pointwise_mode() != "0" and x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 4 and x.numel() > 0
            and isinstance(conv, torch.nn.Conv2d) and conv.kernel_size == (1, 1) and conv.stride == (2, 2)
            and conv.padding == (0, 0) and conv.groups == 1 and conv.bias is None and conv.weight.dtype == torch.float32
            and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0
            and ((x.shape[2] // 2) * (x.shape[3] // 2)) % 4 == 0 and (x.shape[2] // 2) * (x.shape[3] // 2) >= 8
            and _pw16_fits(x.shape[0], max(conv.in_channels, conv.out_channels), (x.shape[2] // 2) * (x.shape[3] // 2)))


class _Conv1x1OddFunc(torch.autograd.Function):
    """1x1 / stride-1 convolution (+ residual) on planes with H * W % 4 != 0 -- the 7x7 planes of layer4 -- on the
    LDS-staged HIP GEMM (k_pw_gemm_odd) and the scalar-pixel d(weight) kernel.  MIOpen ran these as NHWC implicit GEMMs
    between two layout transposes (igemm + Cijk_ + batched_transpose: 1.8 ms of a 25 ms Tiny train step)."""

    @staticmethod
    def forward(ctx, x, weight, residual):
        Fr, Cin, H, W = x.shape
        Cout = weight.shape[0]
        y = torch.empty(Fr, Cout, H, W, dtype=x.dtype, device=x.device)
        dev = x.device
        with torch.cuda.device(dev):
            rc = _native.lib().rk_pw_gemm_odd_f32(weight.data_ptr(), x.data_ptr(),
                                                  residual.data_ptr() if residual is not None else None, y.data_ptr(),
                                                  Fr, Cin, Cout, H * W, 1, torch.cuda.current_stream(dev).cuda_stream)
        _native.check(rc, "rk_pw_gemm_odd_f32")
        ctx.save_for_backward(x, weight)
        ctx.has_residual = residual is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = dy.contiguous()
        Fr, Cin, H, W = x.shape
        Cout = weight.shape[0]
        dev = x.device
        L = _native.lib()
        dx = dw = None
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            if ctx.needs_input_grad[0]:
                dx = torch.empty_like(x)
                _native.check(L.rk_pw_gemm_odd_f32(weight.data_ptr(), dy.data_ptr(), None, dx.data_ptr(), Fr, Cout, Cin,
                                                   H * W, 0, stream), "rk_pw_gemm_odd_f32")
            if ctx.needs_input_grad[1]:
                dw = torch.empty_like(weight)
                nbytes = int(L.rk_pw_wgrad_odd_workspace_bytes(Fr, Cin, Cout, H * W))
                ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=dev)
                _native.check(L.rk_pw_wgrad_odd_f32(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), Fr, Cin, Cout, H * W,
                                                    ws.data_ptr(), nbytes, stream), "rk_pw_wgrad_odd_f32")
        return dx, dw, (dy if ctx.has_residual and ctx.needs_input_grad[2] else None)


class _ConvS2OddFunc(torch.autograd.Function):
    """The 1x1 / stride-2 projecting shortcut onto planes with Ho * Wo % 4 != 0 (14x14 -> 7x7)."""

    @staticmethod
    def forward(ctx, x, weight):
        Fr, Cin, H, W = x.shape
        Cout = weight.shape[0]
        y = torch.empty(Fr, Cout, H // 2, W // 2, dtype=x.dtype, device=x.device)
        dev = x.device
        with torch.cuda.device(dev):
            rc = _native.lib().rk_pw_s2_forward_odd_f32(weight.data_ptr(), x.data_ptr(), y.data_ptr(), Fr, Cin, Cout, H, W,
                                                        torch.cuda.current_stream(dev).cuda_stream)
        _native.check(rc, "rk_pw_s2_forward_odd_f32")
        ctx.save_for_backward(x, weight)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = dy.contiguous()
        Fr, Cin, H, W = x.shape
        Cout = weight.shape[0]
        dev = x.device
        L = _native.lib()
        dx = dw = None
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            if ctx.needs_input_grad[0]:
                dx = torch.empty_like(x)
                _native.check(L.rk_pw_s2_dgrad_odd_f32(weight.data_ptr(), dy.data_ptr(), dx.data_ptr(), Fr, Cin, Cout, H, W,
                                                       stream), "rk_pw_s2_dgrad_odd_f32")
            if ctx.needs_input_grad[1]:
                dw = torch.empty_like(weight)
                nbytes = int(L.rk_pw_wgrad_odd_workspace_bytes(Fr, Cin, Cout, (H // 2) * (W // 2)))
                ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=dev)
                _native.check(L.rk_pw_s2_wgrad_odd_f32(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), Fr, Cin, Cout, H, W,
                                                       ws.data_ptr(), nbytes, stream), "rk_pw_s2_wgrad_odd_f32")
        return dx, dw


class _Conv1x1Odd16Func(torch.autograd.Function):
    """1x1 convolution (+ residual) of bf16 activations on planes with H * W % 4 != 0 (the 7x7 planes of layer4 under
    autocast; `stride` 2: the 14x14 -> 7x7 projecting shortcut, its even pixels gathered first as in _ConvS2Bf16Func) on
    rk_pw16_odd.hip: a workgroup per frame, the packed weight of rk_pw_pack_bf16.  MIOpen ran these as NHWC implicit GEMMs
    between two layout transposes of each tensor: 58 / 173 us forward / forward + backward at [256, 576 -> 576, 7, 7]."""

    @staticmethod
    def forward(ctx, x, weight, residual, stride):
        if stride == 2:
            H, W = x.shape[2], x.shape[3]
            ctx.full = (H, W)
            x = x[:, :, ::2, ::2].contiguous()
        else:
            ctx.full = None
        Fr, Cin, H, W = x.shape
        Cout = weight.shape[0]
        y = torch.empty(Fr, Cout, H, W, dtype=x.dtype, device=x.device)
        fwd, ctx.packed_bwd = _pack(weight)
        dev = x.device
        with torch.cuda.device(dev):
            rc = _native.lib().rk_pw_gemm_packed_odd_bf16(fwd.data_ptr(), x.data_ptr(),
                                                          residual.data_ptr() if residual is not None else None, y.data_ptr(),
                                                          Fr, Cin, Cout, H * W, torch.cuda.current_stream(dev).cuda_stream)
        _native.check(rc, "rk_pw_gemm_packed_odd_bf16")
        ctx.save_for_backward(x, weight)
        ctx.has_residual = residual is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = dy.contiguous()
        if dy.dtype != x.dtype:
            dy = dy.to(x.dtype)
        Fr, Cin, H, W = x.shape
        Cout = weight.shape[0]
        dev = x.device
        L = _native.lib()
        dx = dw = None
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            if ctx.needs_input_grad[0]:
                dx = torch.empty_like(x)
                _native.check(L.rk_pw_gemm_packed_odd_bf16(ctx.packed_bwd.data_ptr(), dy.data_ptr(), None, dx.data_ptr(), Fr, Cout,
                                                           Cin, H * W, stream), "rk_pw_gemm_packed_odd_bf16")
                if ctx.full is not None:
                    full = torch.zeros(Fr, Cin, ctx.full[0], ctx.full[1], dtype=x.dtype, device=dev)
                    full[:, :, ::2, ::2] = dx
                    dx = full
            if ctx.needs_input_grad[1]:
                dw = torch.empty_like(weight)
                nbytes = int(L.rk_pw_wgrad_odd16_workspace_bytes(Fr, Cin, Cout, H * W))
                ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=dev)
                _native.check(L.rk_pw_wgrad_odd16_bf16(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), Fr, Cin, Cout, H * W,
                                                       ws.data_ptr(), nbytes, stream), "rk_pw_wgrad_odd16_bf16")
        return dx, dw, (dy if ctx.has_residual and ctx.needs_input_grad[2] else None), None


def _eligible_odd16(conv, x, stride):
    if not (pointwise_mode() != "0" and x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 4 and x.numel() > 0
            and x.data_ptr() % 16 == 0
            and isinstance(conv, torch.nn.Conv2d) and conv.kernel_size == (1, 1) and conv.stride == (stride, stride)
            and conv.padding == (0, 0) and conv.groups == 1 and conv.bias is None and conv.weight.dtype == torch.float32
            and conv.weight.is_cuda):
        return False
    H, W = x.shape[2], x.shape[3]
    if stride == 2:
        if H % 2 or W % 2:
            return False
        H, W = H // 2, W // 2
    P = H * W
    L = _native.lib()
    # (both directions: the forward's [K -> M] and d(input)'s [M -> K])
    return (P % 4 != 0 and bool(L.rk_pw_odd16_supported(x.shape[0], conv.in_channels, conv.out_channels, P))
            and bool(L.rk_pw_odd16_supported(x.shape[0], conv.out_channels, conv.in_channels, P)))


class _ForkS2(torch.autograd.Function):
    """x -> (x, x[:, :, ::2, ::2] gathered): the two consumers of a downsampling block's activation -- the main path and the
    stride-2 projecting shortcut (backbone.py:98-104) -- as one node, so that their gradients are joined in ONE pass
    (rk_scatter2x2_add_bf16: main + the small gradient scattered to the even pixels).  As separate nodes the shortcut's backward
    wrote zeros, scattered into them, and autograd added the two full-size tensors: three passes, 0.9 ms of a Large-AQ step."""

    @staticmethod
    def forward(ctx, x):
        ctx.shape = tuple(x.shape)
        return x.view_as(x), x[:, :, ::2, ::2].contiguous()

    @staticmethod
    def backward(ctx, g_main, g_small):
        if g_small is None:
            return g_main
        Fr, C, H, W = ctx.shape
        g_small = g_small.contiguous()
        if g_main is not None:
            g_main = g_main.contiguous()
            if g_main.dtype != g_small.dtype:
                g_main = g_main.to(g_small.dtype)
        out = torch.empty(ctx.shape, dtype=g_small.dtype, device=g_small.device)
        dev = g_small.device
        with torch.cuda.device(dev):
            rc = _native.lib().rk_scatter2x2_add_bf16(g_main.data_ptr() if g_main is not None else None, g_small.data_ptr(),
                                                      out.data_ptr(), Fr * C, H, W, torch.cuda.current_stream(dev).cuda_stream)
        _native.check(rc, "rk_scatter2x2_add_bf16")
        return out


def _gathered_kind(conv, Fr, P, dtype, device_ok=True):
    """Which HIP kernels run the stride-2 projecting shortcut `conv` on its GATHERED operand [Fr, K, P]: "pw16" / "odd16" / None."""
    if not (pointwise_mode() != "0" and dtype == torch.bfloat16
            and isinstance(conv, torch.nn.Conv2d) and conv.kernel_size == (1, 1) and conv.stride == (2, 2)
            and conv.padding == (0, 0) and conv.groups == 1 and conv.bias is None and conv.weight.dtype == torch.float32
            and conv.weight.is_cuda):
        return None
    K, M = conv.in_channels, conv.out_channels
    L = _native.lib()
    if P % 4 == 0 and P >= 8 and K % 2 == 0 and M % 2 == 0 and _pw16_fits(Fr, max(K, M), P):
        return "pw16"
    if P % 4 != 0 and L.rk_pw_odd16_supported(Fr, K, M, P) and L.rk_pw_odd16_supported(Fr, M, K, P):
        return "odd16"
    return None


def conv_on_gathered(conv, xs, kind):
    """The stride-2 shortcut `conv` applied to its already gathered operand xs = x[:, :, ::2, ::2] (kind: _gathered_kind)."""
    if kind == "pw16":
        return _Conv1x1Func.apply(xs, conv.weight, True, None, True)
    return _Conv1x1Odd16Func.apply(xs, conv.weight, None, 1)


def fork_shortcut(conv, x):
    """`(x', conv(x))` for the stride-2 projecting shortcut `conv` of a downsampling block whose activation x also feeds the
    main path: the caller continues with x' (an autograd alias of x).  bf16 activations on the HIP kernels; otherwise
    `(x, conv1x1(conv, x))`."""
    ok = (torch.is_grad_enabled() and x.requires_grad and x.is_cuda and x.dim() == 4 and x.is_contiguous() and x.numel() > 0
          and x.data_ptr() % 16 == 0 and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0)
    if ok:
        kind = _gathered_kind(conv, x.shape[0], (x.shape[2] // 2) * (x.shape[3] // 2), x.dtype)
        if kind is not None:
            x_main, xs = _ForkS2.apply(x)
            return x_main, conv_on_gathered(conv, xs, kind)
    return x, conv1x1(conv, x)


_ODD_PMIN, _ODD_PMAX = 37, 64        # k_pw_gemm_odd: frames per 256-column tile / LDS


def _odd_common(conv, x, stride):
    return (pointwise_mode() != "0" and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.numel() > 0
            and not torch.is_autocast_enabled() and x.data_ptr() % 16 == 0
            and isinstance(conv, torch.nn.Conv2d) and conv.kernel_size == (1, 1) and conv.stride == (stride, stride)
            and conv.padding == (0, 0) and conv.groups == 1 and conv.bias is None and conv.weight.dtype == torch.float32
            and conv.in_channels % 4 == 0 and conv.out_channels % 4 == 0)


def _eligible_odd(conv, x):
    P = x.shape[2] * x.shape[3]
    return P % 4 != 0 and _ODD_PMIN <= P <= _ODD_PMAX and _odd_common(conv, x, 1)


def _eligible_s2_odd(conv, x):
    H, W = x.shape[2], x.shape[3]
    Po = (H // 2) * (W // 2)
    return (H % 2 == 0 and W % 2 == 0 and Po % 4 != 0 and _ODD_PMIN <= Po <= _ODD_PMAX and _odd_common(conv, x, 2))


def conv1x1(conv, x, residual=None):
    """`conv(x)` (`conv(x) + residual` when a residual is given) for a 1x1 nn.Conv2d module -- or an nn.Sequential
    ending in one (the -aq variant prepends its AttentionShift to conv2, models.py:_prepare_backbone)."""
    if isinstance(conv, torch.nn.Sequential) and len(conv) > 0 and isinstance(conv[-1], torch.nn.Conv2d):
        for m in list(conv)[:-1]:
            x = m(x)
        conv = conv[-1]
    if residual is None and _eligible_s2(conv, x):
        return _ConvS2Func.apply(x.contiguous(), conv.weight)
    if residual is None and _eligible_s2_bf16(conv, x):
        return _ConvS2Bf16Func.apply(x, conv.weight)
    if x.dim() == 4 and x.is_contiguous() and x.dtype == torch.bfloat16:
        if residual is None and _eligible_odd16(conv, x, 2):
            return _Conv1x1Odd16Func.apply(x, conv.weight, None, 2)
        if (_eligible_odd16(conv, x, 1) and (residual is None or (residual.is_contiguous() and residual.dtype == x.dtype
                                                                     and residual.data_ptr() % 16 == 0))):
            return _Conv1x1Odd16Func.apply(x, conv.weight, residual, 1)
    if x.dim() == 4 and x.is_contiguous():
        if residual is None and _eligible_s2_odd(conv, x):
            return _ConvS2OddFunc.apply(x, conv.weight)
        if (_eligible_odd(conv, x) and (residual is None or (residual.is_contiguous() and residual.dtype == x.dtype))):
            return _Conv1x1OddFunc.apply(x, conv.weight, residual)
    hip_gemm = _eligible(conv, x, residual is not None)
    if hip_gemm is None or (residual is not None and not (residual.is_contiguous() and residual.dtype == x.dtype)):
        y = conv(x)
        if residual is not None:
            y += residual
        return y
    hip_dx = hip_gemm
    if (hip_gemm and x.dtype == torch.bfloat16 and conv.training and torch.is_grad_enabled() and config.switches().fused_train
            and config.switches().fused_bn and config.switches().pw16_stats):
        from .fused_bn import attach_stats
        y, stats = _Conv1x1Func.apply(x.contiguous(), conv.weight, hip_gemm, residual, hip_dx, True)
        return attach_stats(y, stats) if stats.dim() == 3 else y
    return _Conv1x1Func.apply(x.contiguous(), conv.weight, hip_gemm, residual, hip_dx)


# ---------------------------------------------------------------------------------------------------------------
# Inference: BatchNorm in eval mode is a per-channel affine map with constant coefficients, so the block's two
# BN+ReLU pairs ride on the conv2 GEMM -- relu(bn1(x)) on its operand load, relu(bn2(.)) on its epilogue -- and the
# residual add on the conv3 GEMM: per block two GEMMs and the shift touch memory, nothing else.

@contextlib.contextmanager
def prefolded(module):
    """Inside the block `_bn_affine(bn)` of every eval-mode BatchNorm2d of `module` returns its slice of ONE batched fold
    (rk_bn_fold_many_f32: one launch per forward instead of one per BatchNorm -- 28 x 5 us in a RubiksNet-Tiny forward).
    Entered by RubiksNet.forward in eval mode; recomputed on every entry (an edit made through `.data` between two forwards
    is seen), the job table cached on the module and rebuilt when a parameter or buffer moves.  A replica made by
    nn.DataParallel (a fresh object on every forward: its job table would be rebuilt -- one synchronous host-to-device copy
    -- each time) folds layer by layer instead."""
    bns = [m for m in module.modules()
           if isinstance(m, torch.nn.BatchNorm2d) and not m.training and m.running_mean is not None and m.weight is not None
           and all(t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
                   for t in (m.weight, m.bias, m.running_mean, m.running_var))]
    if (_table("prefolded") is not None or getattr(module, "_is_replica", False) or len(bns) < 2 or len(bns) > 65535
            or len({m.weight.device for m in bns}) != 1 or not config.switches().fused_eval):
        yield
        return
    dev = bns[0].weight.device
    key = tuple((m.weight.data_ptr(), m.bias.data_ptr(), m.running_mean.data_ptr(), m.running_var.data_ptr(),
                 int(m.weight.shape[0]), float(m.eps)) for m in bns)
    plan = getattr(module, "_rk_prefold_plan", None)
    if plan is None or plan[0] != key:
        recs, offs, off, max_c = [], [], 0, 0
        for wp, bp, mp, vp, c, eps in key:
            recs.append(struct.pack("<QQQQqif", wp, bp, mp, vp, off, c, eps))
            offs.append(off)
            off += c
            max_c = max(max_c, c)
        jobs = torch.frombuffer(bytearray(b"".join(recs)), dtype=torch.uint8).to(dev)
        plan = (key, jobs, tuple(offs), off, max_c)
        module._rk_prefold_plan = plan
    _, jobs, offs, total, max_c = plan
    ab = torch.empty(2, total, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _native.check(_native.lib().rk_bn_fold_many_f32(jobs.data_ptr(), len(bns), ab.data_ptr(), total, max_c,
                                                        torch.cuda.current_stream(dev).cuda_stream), "rk_bn_fold_many_f32")
    _TABLES.prefolded = {id(m): (ab[0, o:o + int(m.weight.shape[0])], ab[1, o:o + int(m.weight.shape[0])]) for m, o in zip(bns, offs)}
    try:
        yield
    finally:
        _TABLES.prefolded = None


def _bn_affine(bn):
    """(a, b) with bn(x) = a x + b in eval mode.  Recomputed on every call -- ONE launch on [C] elements
    (rk_bn_fold_f32) -- because no cache key sees in-place edits made through `.data` (EMA updates, checkpoint
    surgery; the reference itself initialises with `fc.weight.data.normal_`)."""
    w, bias, mean, var = bn.weight, bn.bias, bn.running_mean, bn.running_var
    table = _table("prefolded")
    if table is not None:
        hit = table.get(id(bn))
        if hit is not None:
            return hit
    if all(t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() for t in (w, bias, mean, var)):
        ab = torch.empty(2, w.shape[0], dtype=torch.float32, device=w.device)
        a, b = ab[0], ab[1]
        with torch.cuda.device(w.device):
            rc = _native.lib().rk_bn_fold_f32(w.data_ptr(), bias.data_ptr(), mean.data_ptr(), var.data_ptr(), float(bn.eps),
                                              a.data_ptr(), b.data_ptr(), w.shape[0],
                                              torch.cuda.current_stream(w.device).cuda_stream)
        _native.check(rc, "rk_bn_fold_f32")
        return a, b
    with torch.no_grad():
        a = (w.float() * torch.rsqrt(var.float() + bn.eps)).contiguous()
        b = (bias.float() - mean.float() * a).contiguous()
    return a, b


def _gemm_fused(conv, x, pro=None, epi=None, residual=None):
    Fr, Cin, H, W = x.shape
    Cout = conv.out_channels
    y = torch.empty(Fr, Cout, H, W, dtype=x.dtype, device=x.device)
    dev = x.device
    ka, kb = pro if pro is not None else (None, None)
    ma, mb = epi if epi is not None else (None, None)
    ptr = lambda t: t.data_ptr() if t is not None else None      # noqa: E731
    with torch.cuda.device(dev):
        rc = _native.lib().rk_pw_gemm_fused_f32(
            conv.weight.data_ptr(), x.data_ptr(), ptr(residual), y.data_ptr(), Fr, Cin, Cout, H * W, 1,
            ptr(ka), ptr(kb), 1, ptr(ma), ptr(mb), 1, torch.cuda.current_stream(dev).cuda_stream)
    _native.check(rc, "rk_pw_gemm_fused_f32")
    return y


def _plain_1x1(conv, x):
    return (isinstance(conv, torch.nn.Conv2d) and conv.kernel_size == (1, 1) and conv.stride == (1, 1)
            and conv.padding == (0, 0) and conv.groups == 1 and conv.bias is None and conv.weight.dtype == torch.float32
            and conv.in_channels % 2 == 0 and conv.out_channels % 2 == 0
            and max(conv.in_channels, conv.out_channels) <= _FUSED_EVAL_CMAX)


def _eval_bn(bn):
    return (isinstance(bn, torch.nn.BatchNorm2d) and not bn.training and bn.affine and bn.track_running_stats
            and bn.running_mean is not None and bn.weight.dtype == torch.float32)


def _strided_shortcut_ok(conv, x):
    return (isinstance(conv, torch.nn.Conv2d) and conv.kernel_size == (1, 1) and conv.stride == (2, 2)
            and conv.padding == (0, 0) and conv.groups == 1 and conv.bias is None and conv.weight.dtype == torch.float32
            and conv.in_channels % 2 == 0 and conv.out_channels % 2 == 0
            and max(conv.in_channels, conv.out_channels) <= _S2_CMAX and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0
            and ((x.shape[2] // 2) * (x.shape[3] // 2)) % 4 == 0)


def _shift_layers(as3):
    """The shift modules inside a block's `as3` (RubiksShift2D, or the 3-D layer behind the temporal wrapper) -- by type,
    not by "anything with a `stride` attribute"."""
    from .shiftlib import RubiksShift2D, RubiksShiftBase

    return [m for m in as3.modules() if isinstance(m, (RubiksShift2D, RubiksShiftBase))]


def _as3_stride_2(as3):
    """True when the only subsampling inside `as3` is one shift with spatial stride (2, 2) and temporal stride 1."""
    strides = []
    for m in _shift_layers(as3):
        st = getattr(m, "stride", None)
        if st is None:
            continue
        st = (st,) * 2 if isinstance(st, int) else tuple(int(v) for v in st)
        if any(v != 1 for v in st):
            strides.append(st)
    return len(strides) == 1 and strides[0][-2:] == (2, 2) and all(v == 1 for v in strides[0][:-2])


def _gemm_s2_fused(conv, x, pro):
    Fr, Cin, H, W = x.shape
    Cout = conv.out_channels
    y = torch.empty(Fr, Cout, H // 2, W // 2, dtype=x.dtype, device=x.device)
    dev = x.device
    ka, kb = pro
    with torch.cuda.device(dev):
        rc = _native.lib().rk_pw_s2_forward_fused_f32(conv.weight.data_ptr(), x.data_ptr(), y.data_ptr(), Fr, Cin, Cout, H, W,
                                                      ka.data_ptr(), kb.data_ptr(), 1,
                                                      torch.cuda.current_stream(dev).cuda_stream)
    _native.check(rc, "rk_pw_s2_forward_fused_f32")
    return y


def _stride_one(as3):
    """True when no shift inside `as3` (RubiksShift2D, the 3-D wrapper, the attention + 2-D pair) subsamples."""
    for m in _shift_layers(as3):
        st = getattr(m, "stride", None)
        if st is None:
            continue
        st = (st,) if isinstance(st, int) else tuple(st)
        if any(int(v) != 1 for v in st):
            return False
    return True


# "Does anything in this block want a gradient?" -- asked once per MODEL forward (RubiksNetBackbone.forward brackets its
# stage loop with `all_frozen(...)`), not once per block per forward by walking block.parameters() (advisor finding: host
# overhead on the launch-bound inference path).  Outside such a bracket the block's own parameters are walked.
_ALL_FROZEN = threading.local()


class all_frozen:
    def __init__(self, module):
        self.value = not any(p.requires_grad for p in module.parameters())

    def __enter__(self):
        self.prev = getattr(_ALL_FROZEN, "value", None)
        _ALL_FROZEN.value = self.value
        return self

    def __exit__(self, *exc):
        _ALL_FROZEN.value = self.prev
        return False


def _frozen(block):
    hint = getattr(_ALL_FROZEN, "value", None)
    if hint:
        return True                                   # the whole backbone is frozen
    return not any(p.requires_grad for p in block.parameters())


def fused_eval_block(block, x):
    """Inference forward of a RubiksShiftBlock with bn1 / bn2 / the residual add fused into the two 1x1 GEMMs, or None
    when the block does not qualify (training mode, a gradient is actually wanted, bf16, SE layer, strided or wide
    layers, small planes, `RK_FUSED_EVAL=0`) -- the caller then runs the layer-by-layer path.  "A gradient is wanted"
    means grad mode is on AND the input or one of the block's parameters requires grad: a plain `model.eval()`
    forward of a frozen model takes the same path, and gives the same logits, with or without `torch.no_grad()`."""
    sw = config.switches()
    if (not sw.fused_eval or sw.pointwise == "0" or block.training
            or not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4)
            or block.se is not None or x.numel() == 0):
        return None
    if torch.is_grad_enabled() and (x.requires_grad or not _frozen(block)):
        return None
    P = x.shape[2] * x.shape[3]
    identity = isinstance(block.shortcut, torch.nn.Identity)
    strided = not _stride_one(block.as3)
    if (P % 4 or P < _FUSED_EVAL_PMIN
            or not (_plain_1x1(block.conv2, x) and _plain_1x1(block.conv3, x))
            or not (_eval_bn(block.bn1) and _eval_bn(block.bn2))):
        return None
    if strided:
        # a downsampling block: as3 subsamples by (2, 2), so does the projecting shortcut (1x1 / stride 2); both GEMMs
        # that read x still take bn1 + ReLU in their operand load
        if (identity or not _strided_shortcut_ok(block.shortcut, x) or not _as3_stride_2(block.as3)
                or (P // 4) % 4 or P // 4 < _FUSED_EVAL_PMIN):
            return None
    elif not (identity or _plain_1x1(block.shortcut, x)):
        return None
    x = x.contiguous()
    pro = _bn_affine(block.bn1)
    if identity:
        shortcut = x
    elif strided:
        shortcut = _gemm_s2_fused(block.shortcut, x, pro)
    else:
        shortcut = _gemm_fused(block.shortcut, x, pro=pro)
    mid = _gemm_fused(block.conv2, x, pro=pro, epi=_bn_affine(block.bn2))
    mid = block.as3(mid)
    return _gemm_fused(block.conv3, mid.contiguous(), residual=shortcut)


# ---------------------------------------------------------------------------------------------------------------
# The stem: Conv3x3(3, width, stride=2) (backbone.py:154) on the same MFMA GEMM, im2col gathered on the fly.

class _StemFunc(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, with_stats=False):
        Fr, Cin, H, W = x.shape
        Cout = weight.shape[0]
        y = torch.empty(Fr, Cout, H // 2, W // 2, dtype=x.dtype, device=x.device)
        dev = x.device
        L = _native.lib()
        stats = None
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            if with_stats:
                # training: the statistics pass of the first block's bn1 rides on this GEMM's epilogue (train_block.py)
                J = int(L.rk_pw_tiles(Fr, (H // 2) * (W // 2)))
                stats = torch.empty(Cout, J, 4, dtype=torch.float32, device=dev)
                rc = L.rk_stem_conv3x3s2_stats_f32(weight.data_ptr(), x.data_ptr(), y.data_ptr(), Fr, Cin, Cout, H, W,
                                                   stats.data_ptr(), J, stream)
            else:
                rc = L.rk_stem_conv3x3s2_f32(weight.data_ptr(), x.data_ptr(), y.data_ptr(), Fr, Cin, Cout, H, W, stream)
        _native.check(rc, "rk_stem_conv3x3s2_f32")
        ctx.save_for_backward(x, weight)
        if with_stats:
            ctx.mark_non_differentiable(stats)
            ctx.set_materialize_grads(False)
            return y, stats
        return y

    @staticmethod
    def backward(ctx, dy, _dstats=None):
        x, weight = ctx.saved_tensors
        if dy is None:
            return None, None, None
        dy = dy.contiguous()
        gx = gw = None
        if ctx.needs_input_grad[0]:                       # (never in the networks: the stem's input is the clip)
            gx = torch.ops.aten.convolution_backward(dy, x, weight, None, [2, 2], [1, 1], [1, 1], False, [0, 0], 1,
                                                     [True, False, False])[0]
        if ctx.needs_input_grad[1]:
            # d(weight) on the same im2col-on-the-fly gather as the forward: MIOpen's pair of asm kernels takes
            # 1.04 ms per step at [256,3,224,224] -> 54 channels
            Fr, Cin, H, W = x.shape
            Cout = weight.shape[0]
            dev = x.device
            L = _native.lib()
            gw = torch.empty_like(weight)
            with torch.cuda.device(dev):
                nbytes = int(L.rk_pw_wgrad_workspace_bytes(Fr, 9 * Cin, Cout, (H // 2) * (W // 2)))
                ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=dev)
                rc = L.rk_stem_wgrad3x3s2_f32(dy.data_ptr(), x.data_ptr(), gw.data_ptr(), Fr, Cin, Cout, H, W,
                                              ws.data_ptr(), nbytes, torch.cuda.current_stream(dev).cuda_stream)
            _native.check(rc, "rk_stem_wgrad3x3s2_f32")
        return gx, gw, None


class _Stem16Func(torch.autograd.Function):
    """The stem under bf16 autocast (rk_stem16.hip): fp32 clip in, bf16 activation out; d(weight) from the bf16 gradient and the
    clip.  MIOpen: cast + layout transposes + implicit GEMM, 0.89 ms forward + 0.56 ms d(weight) at [256, 3, 224, 224] -> 72."""

    @staticmethod
    def forward(ctx, x, weight):
        Fr, Cin, H, W = x.shape
        Cout = weight.shape[0]
        y = torch.empty(Fr, Cout, H // 2, W // 2, dtype=torch.bfloat16, device=x.device)
        dev = x.device
        with torch.cuda.device(dev):
            rc = _native.lib().rk_stem_conv3x3s2_bf16out(weight.data_ptr(), x.data_ptr(), y.data_ptr(), Fr, Cin, Cout, H, W,
                                                         torch.cuda.current_stream(dev).cuda_stream)
        _native.check(rc, "rk_stem_conv3x3s2_bf16out")
        ctx.save_for_backward(x, weight)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        gx = gw = None
        if ctx.needs_input_grad[0]:                       # (never in the networks: the stem's input is the clip)
            gx = torch.ops.aten.convolution_backward(dy.float(), x, weight, None, [2, 2], [1, 1], [1, 1], False, [0, 0], 1,
                                                     [True, False, False])[0]
        if ctx.needs_input_grad[1]:
            dy = dy.contiguous()
            if dy.dtype != torch.bfloat16:
                dy = dy.to(torch.bfloat16)
            Fr, Cin, H, W = x.shape
            Cout = weight.shape[0]
            dev = x.device
            L = _native.lib()
            gw = torch.empty_like(weight)
            with torch.cuda.device(dev):
                nbytes = int(L.rk_stem_wgrad16_workspace_bytes(Fr, Cin, Cout, H, W))
                ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=dev)
                rc = L.rk_stem_wgrad3x3s2_bf16(dy.data_ptr(), x.data_ptr(), gw.data_ptr(), Fr, Cin, Cout, H, W, ws.data_ptr(),
                                               nbytes, torch.cuda.current_stream(dev).cuda_stream)
            _native.check(rc, "rk_stem_wgrad3x3s2_bf16")
        return gx, gw


def _stem16_ok(conv, x):
    return (pointwise_mode() != "0" and torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16
            and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.numel() > 0 and x.data_ptr() % 16 == 0
            and isinstance(conv, torch.nn.Conv2d) and conv.kernel_size == (3, 3) and conv.stride == (2, 2)
            and conv.padding == (1, 1) and conv.dilation == (1, 1) and conv.groups == 1 and conv.bias is None
            and conv.weight.dtype == torch.float32 and conv.weight.is_cuda and conv.weight.is_contiguous()
            and bool(_native.lib().rk_stem16_supported(x.shape[0], conv.in_channels, conv.out_channels, x.shape[2], x.shape[3])))


def stem_conv(conv, x):
    """`conv(x)` for the backbone's 3x3 / stride-2 / pad-1 first layer (forward and d(weight) on the HIP GEMM kernels)."""
    if _stem16_ok(conv, x):
        return _Stem16Func.apply(x.contiguous(), conv.weight)     # bf16 autocast: fp32 clip in, bf16 activation out
    # under any other autocast the stock layer decides the activation's type: left to it (an fp32 output here would keep
    # the next BatchNorm / shift in fp32 storage)
    ok = (pointwise_mode() != "0" and not torch.is_autocast_enabled()
          and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.numel() > 0
          and isinstance(conv, torch.nn.Conv2d) and conv.kernel_size == (3, 3) and conv.stride == (2, 2)
          and conv.padding == (1, 1) and conv.dilation == (1, 1) and conv.groups == 1 and conv.bias is None
          and conv.weight.dtype == torch.float32 and 9 * conv.in_channels <= 64
          and x.shape[2] % 2 == 0 and x.shape[3] % 8 == 0)
    if not ok:
        return conv(x)
    sw = config.switches()
    if (sw.fused_train and sw.fused_bn and conv.training and torch.is_grad_enabled() and conv.out_channels <= 128
            and ((x.shape[2] // 2) * (x.shape[3] // 2)) % 4 == 0):
        from .train_block import _attach_stats

        y, stats = _StemFunc.apply(x.contiguous(), conv.weight, True)
        return _attach_stats(y, stats)
    return _StemFunc.apply(x.contiguous(), conv.weight)
