"""BatchNorm2d + ReLU as one HIP operator (librubiks_hip: rk_bn_relu_*), SURVEY 8(f) row f3.

Every BatchNorm2d of the backbone is followed by `nn.ReLU(inplace=True)` (rubiksnet/backbone.py:129-131,
:196).  `bn_relu(bn, x)` computes `relu(bn(x))` for an ordinary `nn.BatchNorm2d` module -- its parameters,
buffers, momentum / eps / training flag and state-dict stay exactly what they are -- through the fused kernels
when x is a CUDA fp32 / bf16 NCHW tensor, and through `F.relu(bn(x))` (stock PyTorch) otherwise (CPU tensors
of the gloo tests, other dtypes, eval mode with gradients).  `RK_FUSED_BN=0` forces the stock path.

Training forward: 12 B/elem (stats pass + normalise pass), backward: 20 B/elem, nothing saved but x and the
[C] mean / invstd (stock: BN saves x, ReLU saves its output).
"""
import torch
import torch.nn.functional as F

from . import _native, config

__all__ = ["bn_relu", "bn_relu_skip", "bn_relu_tshift_skip", "bn_relu_shift2d", "fused_bn_enabled"]

_SFX = {torch.float32: "f32", torch.bfloat16: "bf16"}


def fused_bn_enabled():
    return config.switches().fused_bn


def _ws(L, Fr, C, P, dev):
    nbytes = int(L.rk_bn_workspace_bytes(Fr, C, P))
    return torch.empty(max(nbytes, 1), dtype=torch.uint8, device=dev), nbytes


def _ptr(t):
    return t.data_ptr() if t is not None else None


# ---- tile statistics handed over by the GEMM that produced a tensor (rk_pw_gemm_packed_stats_bf16, round 5) ----
def attach_stats(t, stats):
    """`stats` (float4 [C][tiles], one record per 64 columns) describe `t` as it is now."""
    t._rk_stats = stats
    t._rk_stats_count = t.numel() // t.shape[1]
    t._rk_stats_version = t._version          # an in-place edit of the tensor invalidates them
    return t


def take_stats(x):
    st = getattr(x, "_rk_stats", None)
    if st is None:
        return None
    if (st.dim() != 3 or st.shape[0] != x.shape[1] or st.shape[1] < 1 or st.shape[2] != 4 or st.device != x.device
            or st.dtype != torch.float32 or getattr(x, "_rk_stats_count", None) != x.numel() // x.shape[1]
            or getattr(x, "_rk_stats_version", None) != x._version):
        return None
    return st


def _finish_tiles(L, stats, count, weight, bias, running_mean, running_var, momentum, eps, counter_ptr, dev, stream):
    """Tile records -> rows (mean, invstd, a, b) [+ 4 packed rows]; running statistics / num_batches_tracked as
    nn.BatchNorm2d's training forward (the same finisher the fp32 fused block uses, train_block._finish)."""
    C = weight.shape[0]
    out = torch.empty(8, C, dtype=torch.float32, device=dev)
    _native.check(L.rk_bn_finish_tiles_f32(stats.data_ptr(), stats.shape[1], count, weight.data_ptr(), bias.data_ptr(),
                                           _ptr(running_mean), _ptr(running_var), out[0].data_ptr(), out[1].data_ptr(),
                                           out[2].data_ptr(), out[3].data_ptr(), out[4].data_ptr(), C, float(eps),
                                           float(momentum), counter_ptr, stream), "rk_bn_finish_tiles_f32")
    return out


class _BNReLUTrain(torch.autograd.Function):
    """y = relu?(batch_norm(x)) with batch statistics; updates running_mean / running_var in place."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, momentum, eps, relu, with_skip=False, counter_ptr=None,
                stats=None):
        L = _native.lib()
        Fr, C, H, W = x.shape
        P = H * W
        dev = x.device
        y = torch.empty_like(x)
        if stats is not None:
            # the producing GEMM already reduced x tile by tile: finish + ONE normalising pass (no statistics pass over x)
            with torch.cuda.device(dev):
                stream = torch.cuda.current_stream(dev).cuda_stream
                fin = _finish_tiles(L, stats, Fr * P, weight, bias, running_mean, running_var, momentum, eps, counter_ptr, dev,
                                    stream)
                _native.check(getattr(L, "rk_bn_apply_affine_" + _SFX[x.dtype])(
                    x.data_ptr(), fin[2].data_ptr(), fin[3].data_ptr(), y.data_ptr(), Fr, C, P, int(relu), stream),
                    "rk_bn_apply_affine")
            ctx.save_for_backward(x, weight, bias, fin[0], fin[1])
            ctx.relu = relu
            ctx.with_skip = with_skip
            if with_skip:
                return y, x.view_as(x)
            return y
        save_mean = torch.empty(C, dtype=torch.float32, device=dev)
        save_invstd = torch.empty(C, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            ws, nbytes = _ws(L, Fr, C, P, dev)
            # counter_ptr: address of the module's num_batches_tracked (int64 device scalar), incremented in the launch
            rc = getattr(L, "rk_bn_relu_forward_counted_" + _SFX[x.dtype])(
                x.data_ptr(), weight.data_ptr(), bias.data_ptr(), _ptr(running_mean), _ptr(running_var),
                save_mean.data_ptr(), save_invstd.data_ptr(), y.data_ptr(), Fr, C, P, float(eps), float(momentum),
                int(relu), counter_ptr, ws.data_ptr(), nbytes, torch.cuda.current_stream(dev).cuda_stream)
        _native.check(rc, "rk_bn_relu_forward")
        ctx.save_for_backward(x, weight, bias, save_mean, save_invstd)
        ctx.relu = relu
        ctx.with_skip = with_skip
        # (running_mean / running_var are buffers updated in place by the kernel, as F.batch_norm does)
        if with_skip:
            # second output: x itself, for the block's identity shortcut -- its gradient then arrives HERE and is
            # added inside the d(x) kernel instead of by a separate autograd accumulation pass
            return y, x.view_as(x)
        return y

    @staticmethod
    def backward(ctx, dy, dskip=None):
        x, weight, bias, save_mean, save_invstd = ctx.saved_tensors
        L = _native.lib()
        Fr, C, H, W = x.shape
        P = H * W
        dev = x.device
        dy = dy.contiguous()
        if dy.dtype != x.dtype:
            dy = dy.to(x.dtype)
        if dskip is not None:
            dskip = dskip.contiguous()
            if dskip.dtype != x.dtype:
                dskip = dskip.to(x.dtype)
        dx = torch.empty_like(x)
        dgamma = torch.empty(C, dtype=torch.float32, device=dev)
        dbeta = torch.empty(C, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            ws, nbytes = _ws(L, Fr, C, P, dev)
            rc = getattr(L, "rk_bn_relu_backward_" + _SFX[x.dtype])(
                dy.data_ptr(), x.data_ptr(), weight.data_ptr(), bias.data_ptr(), save_mean.data_ptr(),
                save_invstd.data_ptr(), _ptr(dskip), dx.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), Fr, C, P,
                int(ctx.relu), ws.data_ptr(), nbytes, torch.cuda.current_stream(dev).cuda_stream)
        _native.check(rc, "rk_bn_relu_backward")
        return dx, dgamma.to(weight.dtype), dbeta.to(bias.dtype), None, None, None, None, None, None, None, None


class _BNReLUTShiftTrain(torch.autograd.Function):
    """(tshift3(relu(batch_norm(x)), taps), x): the -aq block's training-mode bn1 + ReLU folded into the temporal 3-tap
    filter that consumes it (rk_tshift3_bn_*): the activation is never stored.  Forward = statistics pass + one filter
    pass (instead of statistics, normalise, filter: 5 -> 3 tensor passes); backward = the filter's backward, which also
    emits the ReLU-masked gradient and BatchNorm's two reduction sums, + the d(x) pass (8 -> 6).  The second output is x
    for the block's identity shortcut, whose gradient joins inside the d(x) kernel (cf. _BNReLUTrain with_skip)."""

    @staticmethod
    def forward(ctx, x, weight, bias, taps, running_mean, running_var, momentum, eps, n_segment, counter_ptr, stats=None):
        L = _native.lib()
        Fr, C, H, W = x.shape
        P = H * W
        dev = x.device
        sfx = _SFX[x.dtype]
        taps32 = taps.detach().float().contiguous()
        y = torch.empty_like(x)
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            if stats is not None:          # x came out of a GEMM that reduced it tile by tile: no statistics pass
                fin = _finish_tiles(L, stats, Fr * P, weight, bias, running_mean, running_var, momentum, eps, counter_ptr, dev,
                                    stream)
                save_mean, save_invstd, ab = fin[0], fin[1], fin[2:4]
            else:
                save_mean = torch.empty(C, dtype=torch.float32, device=dev)
                save_invstd = torch.empty(C, dtype=torch.float32, device=dev)
                ab = torch.empty(2, C, dtype=torch.float32, device=dev)
                ws, nbytes = _ws(L, Fr, C, P, dev)
                _native.check(getattr(L, "rk_bn_stats_finish_" + sfx)(
                    x.data_ptr(), weight.data_ptr(), bias.data_ptr(), _ptr(running_mean), _ptr(running_var), save_mean.data_ptr(),
                    save_invstd.data_ptr(), ab.data_ptr(), Fr, C, P, float(eps), float(momentum), counter_ptr, ws.data_ptr(),
                    nbytes, stream), "rk_bn_stats_finish")
            _native.check(getattr(L, "rk_tshift3_bn_forward_" + sfx)(
                x.data_ptr(), taps32.data_ptr(), ab.data_ptr(), y.data_ptr(), Fr, n_segment, C, P, stream),
                "rk_tshift3_bn_forward")
        ctx.save_for_backward(x, weight, bias, taps32, save_mean, save_invstd, ab)
        ctx.n_segment = n_segment
        ctx.taps_dtype = taps.dtype
        return y, x.view_as(x)

    @staticmethod
    def backward(ctx, gy, dskip=None):
        x, weight, bias, taps32, save_mean, save_invstd, ab = ctx.saved_tensors
        L = _native.lib()
        Fr, C, H, W = x.shape
        P = H * W
        S = ctx.n_segment
        dev = x.device
        sfx = _SFX[x.dtype]
        gy = gy.contiguous()
        if gy.dtype != x.dtype:
            gy = gy.to(x.dtype)
        if dskip is not None:
            dskip = dskip.contiguous()
            if dskip.dtype != x.dtype:
                dskip = dskip.to(x.dtype)
        dz = torch.empty_like(x)
        gtaps = torch.empty_like(taps32)
        k12 = torch.empty(2, C, dtype=torch.float32, device=dev)
        dgamma = torch.empty(C, dtype=torch.float32, device=dev)
        dbeta = torch.empty(C, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            # (BatchNorm's two sums ride as granules next to the tap sums and are finished by the same finalizer waves: no
            # rk_bn_bwd_finish_tiles launch)
            nb = int(L.rk_tshift3_bn_backward_fin_workspace_bytes(Fr, S, C, P))
            ws = torch.empty(max(nb, 1), dtype=torch.uint8, device=dev)
            _native.check(getattr(L, "rk_tshift3_bn_backward_fin_" + sfx)(
                gy.data_ptr(), x.data_ptr(), taps32.data_ptr(), ab.data_ptr(), save_mean.data_ptr(), save_invstd.data_ptr(),
                dz.data_ptr(), gtaps.data_ptr(), k12.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), Fr, S, C, P, ws.data_ptr(),
                nb, stream), "rk_tshift3_bn_backward_fin")
            _native.check(getattr(L, "rk_bn_bwd_dx_pre_" + sfx)(
                dz.data_ptr(), x.data_ptr(), weight.data_ptr(), save_mean.data_ptr(), save_invstd.data_ptr(), k12.data_ptr(),
                _ptr(dskip), dz.data_ptr(), Fr, C, P, stream), "rk_bn_bwd_dx_pre")            # in place: dz -> d(x)
        return (dz, dgamma.to(weight.dtype), dbeta.to(bias.dtype), gtaps.to(ctx.taps_dtype), None, None, None, None, None, None,
                None)


def bn_relu_tshift_skip(bn, shift, x):
    """(`shift(relu(bn(x)))`, x) for an -aq block with an identity shortcut in training mode, the activation never stored
    (`shift`: the block's AttentionShift).  None when it does not apply (the caller then takes bn_relu_skip + shift)."""
    sw = config.switches()
    if not (sw.fused_train and _fusable(bn, x) and bn.training and torch.is_grad_enabled() and x.requires_grad):
        return None
    w = getattr(shift, "weight", None)
    S = getattr(shift, "n_segment", 0)
    if (w is None or not w.is_cuda or w.dim() != 2 or w.shape != (x.shape[1], 3) or S <= 0 or x.shape[0] % S
            or bn.num_features != x.shape[1]):
        return None
    x = x.contiguous()
    momentum, counter = _count_batch(bn)
    rm = bn.running_mean if bn.track_running_stats else None
    rv = bn.running_var if bn.track_running_stats else None
    return _BNReLUTShiftTrain.apply(x, bn.weight, bn.bias, shift.soft_taps(), rm, rv, momentum, bn.eps, S, _ptr(counter),
                                    take_stats(x))


class _BNReLUTShiftForkTrain(torch.autograd.Function):
    """(tshift3(relu(batch_norm(x)), taps), relu(batch_norm(x))[:, :, ::2, ::2]): a DOWNSAMPLING -aq block's bn1 + ReLU, whose
    activation feeds the AttentionShift in front of conv2 AND the stride-2 projecting shortcut (backbone.py:98-104, :129).  As
    _BNReLUTShiftTrain, with the shortcut's operand gathered by rk_bn_relu_gather2_* (the full-size activation is never stored)
    and its gradient -- quarter size -- joined to d(activation) INSIDE the filter's backward, before the ReLU mask and
    BatchNorm's sums (rk_tshift3_bn_backward_fork_*).  Unfused this block ran normalise + reduce + d(x) passes, and zeros +
    scatter + add for the shortcut's gradient, all over the full-size tensor."""

    @staticmethod
    def forward(ctx, x, weight, bias, taps, running_mean, running_var, momentum, eps, n_segment, counter_ptr, stats=None):
        L = _native.lib()
        Fr, C, H, W = x.shape
        P = H * W
        dev = x.device
        sfx = _SFX[x.dtype]
        taps32 = taps.detach().float().contiguous()
        y = torch.empty_like(x)
        xs = torch.empty(Fr, C, H // 2, W // 2, dtype=x.dtype, device=dev)
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            if stats is not None:
                fin = _finish_tiles(L, stats, Fr * P, weight, bias, running_mean, running_var, momentum, eps, counter_ptr, dev,
                                    stream)
                save_mean, save_invstd, ab = fin[0], fin[1], fin[2:4]
            else:
                save_mean = torch.empty(C, dtype=torch.float32, device=dev)
                save_invstd = torch.empty(C, dtype=torch.float32, device=dev)
                ab = torch.empty(2, C, dtype=torch.float32, device=dev)
                ws, nbytes = _ws(L, Fr, C, P, dev)
                _native.check(getattr(L, "rk_bn_stats_finish_" + sfx)(
                    x.data_ptr(), weight.data_ptr(), bias.data_ptr(), _ptr(running_mean), _ptr(running_var), save_mean.data_ptr(),
                    save_invstd.data_ptr(), ab.data_ptr(), Fr, C, P, float(eps), float(momentum), counter_ptr, ws.data_ptr(),
                    nbytes, stream), "rk_bn_stats_finish")
            _native.check(getattr(L, "rk_tshift3_bn_forward_" + sfx)(
                x.data_ptr(), taps32.data_ptr(), ab.data_ptr(), y.data_ptr(), Fr, n_segment, C, P, stream),
                "rk_tshift3_bn_forward")
            _native.check(getattr(L, "rk_bn_relu_gather2_" + sfx)(x.data_ptr(), ab.data_ptr(), xs.data_ptr(), Fr, C, H, W, stream),
                          "rk_bn_relu_gather2")
        ctx.save_for_backward(x, weight, bias, taps32, save_mean, save_invstd, ab)
        ctx.n_segment = n_segment
        ctx.taps_dtype = taps.dtype
        return y, xs

    @staticmethod
    def backward(ctx, gy, gxs):
        x, weight, bias, taps32, save_mean, save_invstd, ab = ctx.saved_tensors
        L = _native.lib()
        Fr, C, H, W = x.shape
        P = H * W
        S = ctx.n_segment
        dev = x.device
        sfx = _SFX[x.dtype]
        if gy is None:
            gy = torch.zeros_like(x)
        gy = gy.contiguous()
        if gy.dtype != x.dtype:
            gy = gy.to(x.dtype)
        if gxs is None:
            gxs = torch.zeros(Fr, C, H // 2, W // 2, dtype=x.dtype, device=dev)
        gxs = gxs.contiguous()
        if gxs.dtype != x.dtype:
            gxs = gxs.to(x.dtype)
        dz = torch.empty_like(x)
        gtaps = torch.empty_like(taps32)
        k12 = torch.empty(2, C, dtype=torch.float32, device=dev)
        dgamma = torch.empty(C, dtype=torch.float32, device=dev)
        dbeta = torch.empty(C, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            nb = int(L.rk_tshift3_bn_backward_fin_workspace_bytes(Fr, S, C, P))
            ws = torch.empty(max(nb, 1), dtype=torch.uint8, device=dev)
            _native.check(getattr(L, "rk_tshift3_bn_backward_fork_" + sfx)(
                gy.data_ptr(), x.data_ptr(), taps32.data_ptr(), ab.data_ptr(), save_mean.data_ptr(), save_invstd.data_ptr(),
                gxs.data_ptr(), dz.data_ptr(), gtaps.data_ptr(), k12.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), Fr, S, C, H, W,
                ws.data_ptr(), nb, stream), "rk_tshift3_bn_backward_fork")
            _native.check(getattr(L, "rk_bn_bwd_dx_pre_" + sfx)(
                dz.data_ptr(), x.data_ptr(), weight.data_ptr(), save_mean.data_ptr(), save_invstd.data_ptr(), k12.data_ptr(),
                None, dz.data_ptr(), Fr, C, P, stream), "rk_bn_bwd_dx_pre")                       # in place: dz -> d(x)
        return (dz, dgamma.to(weight.dtype), dbeta.to(bias.dtype), gtaps.to(ctx.taps_dtype), None, None, None, None, None, None,
                None)


def bn_relu_tshift_fork(bn, shift, x):
    """(`shift(relu(bn(x)))`, `relu(bn(x))[:, :, ::2, ::2]`) for a downsampling -aq block in training mode (`shift`: the block's
    AttentionShift; the second element is what the stride-2 projecting shortcut reads).  None when it does not apply."""
    sw = config.switches()
    if not (sw.fused_train and sw.bn_tshift_fork and _fusable(bn, x) and bn.training and torch.is_grad_enabled()):
        return None
    w = getattr(shift, "weight", None)
    S = getattr(shift, "n_segment", 0)
    H, W = x.shape[2], x.shape[3]
    if (w is None or not w.is_cuda or w.dim() != 2 or w.shape != (x.shape[1], 3) or S <= 0 or x.shape[0] % S
            or bn.num_features != x.shape[1] or H % 2 or W % 2 or x.data_ptr() % 16):
        return None
    # the widest pack with W % pack == 0 must hold at least 2 elements (always, W is even) and stay inside the operand's limits
    x = x.contiguous()
    momentum, counter = _count_batch(bn)
    rm = bn.running_mean if bn.track_running_stats else None
    rv = bn.running_var if bn.track_running_stats else None
    return _BNReLUTShiftForkTrain.apply(x, bn.weight, bn.bias, shift.soft_taps(), rm, rv, momentum, bn.eps, S, _ptr(counter),
                                        take_stats(x))


class _BNReLUShift2DTrain(torch.autograd.Function):
    """shift2d(relu(batch_norm(z))): the -aq block's training-mode bn2 + ReLU folded into the RubiksShift2D that consumes
    it (rk2d_*_bn_*): the activation is never stored.  Forward = statistics pass + ONE shift pass (instead of statistics,
    normalise, shift: 5 -> 3 tensor passes); backward = the shift's backward, which recomputes the activation where d(shift)
    needs it and emits the ReLU-masked gradient together with BatchNorm's two reduction sums, + the d(x) pass (8 -> 6)."""

    _SHIFT_SFX = {torch.float32: "f32", torch.bfloat16: "bf16_sf32"}

    @staticmethod
    def forward(ctx, z, weight, bias, shift, running_mean, running_var, momentum, eps, counter_ptr, normalize_grad, stats=None,
                stride=(1, 1), padding=(0, 0)):
        L = _native.lib()
        Fr, C, H, W = z.shape
        P = H * W
        dev = z.device
        sfx = _SFX[z.dtype]
        (sH, sW), (pH, pW) = stride, padding
        # (rubiks.cpp:18's output size, not the convolution formula)
        y = torch.empty((Fr, C, (H + 2 * pH - 1) // sH + 1, (W + 2 * pW - 1) // sW + 1), dtype=z.dtype, device=dev)
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            abmi = None
            if stats is not None:
                fin = _finish_tiles(L, stats, Fr * P, weight, bias, running_mean, running_var, momentum, eps, counter_ptr, dev,
                                    stream)
                save_mean, save_invstd, ab = fin[0], fin[1], fin[2:4]
            else:
                save_mean = torch.empty(C, dtype=torch.float32, device=dev)
                save_invstd = torch.empty(C, dtype=torch.float32, device=dev)
                ab = torch.empty(2, C, dtype=torch.float32, device=dev)
                abmi = torch.empty(C, 4, dtype=torch.float32, device=dev)      # (a, b, mean, invstd): what the backward reads
                ws, nbytes = _ws(L, Fr, C, P, dev)
                _native.check(getattr(L, "rk_bn_stats_finish_abmi_" + sfx)(
                    z.data_ptr(), weight.data_ptr(), bias.data_ptr(), _ptr(running_mean), _ptr(running_var), save_mean.data_ptr(),
                    save_invstd.data_ptr(), ab.data_ptr(), abmi.data_ptr(), Fr, C, P, float(eps), float(momentum), counter_ptr,
                    ws.data_ptr(), nbytes, stream), "rk_bn_stats_finish_abmi")
            if abmi is None:
                abmi = torch.stack((ab[0], ab[1], save_mean, save_invstd), dim=1).contiguous()
            rc = getattr(L, "rk2d_forward_bn_" + _BNReLUShift2DTrain._SHIFT_SFX[z.dtype])(
                z.data_ptr(), ab.data_ptr(), shift.data_ptr(), y.data_ptr(), Fr, C, H, W, sH, sW, pH, pW, 0, stream)
        _native.check(rc, "rk2d_forward_bn")
        ctx.save_for_backward(z, weight, bias, shift, save_mean, save_invstd, ab, abmi)
        ctx.normalize_grad = bool(normalize_grad)
        ctx.geometry = (sH, sW, pH, pW)
        return y

    @staticmethod
    def backward(ctx, gy):
        z, weight, bias, shift, save_mean, save_invstd, ab, abmi = ctx.saved_tensors
        L = _native.lib()
        Fr, C, H, W = z.shape
        P = H * W
        dev = z.device
        sfx = _SFX[z.dtype]
        gy = gy.contiguous()
        if gy.dtype != z.dtype:
            gy = gy.to(z.dtype)
        if gy.data_ptr() % 16:
            # a contiguous VIEW at an odd storage offset: the fused kernels want 16-byte aligned planes, and by now bn2's
            # running statistics are updated -- there is no unfused path to fall back to.  A fresh buffer is aligned.
            gy = gy.clone()
        dz = torch.empty_like(z)
        gshift = torch.empty_like(shift)
        k12 = torch.empty(2, C, dtype=torch.float32, device=dev)
        dgamma = torch.empty(C, dtype=torch.float32, device=dev)
        dbeta = torch.empty(C, dtype=torch.float32, device=dev)
        sH, sW, pH, pW = ctx.geometry
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            nb = int(L.rk2d_backward_bn_workspace_bytes(Fr, C, H, W, sH, sW, pH, pW))
            ws = torch.empty(max(nb, 1), dtype=torch.uint8, device=dev)
            _native.check(getattr(L, "rk2d_backward_bn_" + _BNReLUShift2DTrain._SHIFT_SFX[z.dtype])(
                gy.data_ptr(), z.data_ptr(), abmi.data_ptr(), shift.data_ptr(), dz.data_ptr(), gshift.data_ptr(), k12.data_ptr(),
                dgamma.data_ptr(), dbeta.data_ptr(), Fr, C, H, W, sH, sW, pH, pW, int(ctx.normalize_grad), 0, ws.data_ptr(), nb,
                stream), "rk2d_backward_bn")
            _native.check(getattr(L, "rk_bn_bwd_dx_pre_" + sfx)(
                dz.data_ptr(), z.data_ptr(), weight.data_ptr(), save_mean.data_ptr(), save_invstd.data_ptr(), k12.data_ptr(),
                None, dz.data_ptr(), Fr, C, P, stream), "rk_bn_bwd_dx_pre")                       # in place: dz -> d(z)
        return (dz, dgamma.to(weight.dtype), dbeta.to(bias.dtype), gshift) + (None,) * 9


def _pair(v):
    if isinstance(v, int):
        return (v, v)
    v = tuple(int(k) for k in v)
    return v if len(v) == 2 else None


def bn_relu_shift2d(bn, as3, z):
    """`as3(relu(bn(z)))` for an -aq block in training mode with the activation never stored (`as3`: the block's
    RubiksShift2D).  None when no fused kernel applies (the caller then takes bn_relu + as3)."""
    sw = config.switches()
    if not (sw.fused_train and sw.bn_shift2d and _fusable(bn, z) and bn.training and torch.is_grad_enabled()):
        return None
    shift = getattr(as3, "shift", None)
    if (shift is None or not shift.is_cuda or shift.dtype != torch.float32 or shift.dim() != 2
            or shift.shape != (2, z.shape[1]) or bn.num_features != z.shape[1]):
        return None
    stride, padding = _pair(getattr(as3, "stride", 1)), _pair(getattr(as3, "padding", 0))
    if (getattr(as3, "quantize", False) or stride is None or padding is None or min(stride) < 1 or min(padding) < 0
            or z.data_ptr() % 16 or z.numel() >= 1 << 31):
        return None
    # (asked BEFORE bn2's statistics run: their side effects -- the running statistics -- must happen exactly once)
    if not _native.lib().rk2d_bn_fused_shape(z.shape[0], z.shape[1], z.shape[2], z.shape[3], stride[0], stride[1], padding[0],
                                             padding[1], z.element_size()):
        return None                  # (fp32 planes the LDS-DMA kernels stream keep normalise + shift)
    z = z.contiguous()
    momentum, counter = _count_batch(bn)
    rm = bn.running_mean if bn.track_running_stats else None
    rv = bn.running_var if bn.track_running_stats else None
    return _BNReLUShift2DTrain.apply(z, bn.weight, bn.bias, shift, rm, rv, momentum, bn.eps, _ptr(counter),
                                     bool(getattr(as3, "normalize_grad", True)), take_stats(z), stride, padding)


def _eval_forward(x, weight, bias, running_mean, running_var, eps, relu):
    L = _native.lib()
    Fr, C, H, W = x.shape
    dev = x.device
    y = torch.empty_like(x)
    with torch.cuda.device(dev):
        rc = getattr(L, "rk_bn_relu_forward_" + _SFX[x.dtype])(
            x.data_ptr(), weight.data_ptr(), bias.data_ptr(), running_mean.data_ptr(), running_var.data_ptr(),
            None, None, y.data_ptr(), Fr, C, H * W, float(eps), 0.0, int(relu), 0, None, 0,
            torch.cuda.current_stream(dev).cuda_stream)
    _native.check(rc, "rk_bn_relu_forward")
    return y


def _fusable(bn, x):
    return (
        fused_bn_enabled()
        and isinstance(bn, torch.nn.BatchNorm2d)
        and x.is_cuda and x.dim() == 4 and x.dtype in _SFX and x.numel() > 0
        and bn.affine and bn.weight.dtype == torch.float32
        and (bn.running_mean is None or bn.running_mean.dtype == torch.float32)
    )


def _count_batch(bn):
    """nn.BatchNorm2d.forward's bookkeeping (torch/nn/modules/batchnorm.py): (momentum, counter).  `counter` is the
    num_batches_tracked tensor for the kernel to increment, or None when it was already incremented here (momentum=None:
    the cumulative average needs the new count on the host) or is not tracked."""
    momentum = 0.0 if bn.momentum is None else bn.momentum
    counter = None
    if bn.training and bn.track_running_stats and bn.num_batches_tracked is not None:
        nbt = bn.num_batches_tracked
        if bn.momentum is None:
            nbt.add_(1)
            momentum = 1.0 / float(nbt)
        elif nbt.is_cuda and nbt.dtype == torch.int64 and nbt.numel() == 1:
            counter = nbt
        else:
            nbt.add_(1)
    return momentum, counter


def bn_relu_skip(bn, x):
    """(`relu(bn(x))`, x) for a block whose input also feeds an identity shortcut.  On the fused training path the
    second element is an autograd alias of x whose gradient is added inside the BN d(x) kernel (one elementwise pass
    less per block); elsewhere it is x itself."""
    if (_fusable(bn, x) and (bn.training or (bn.running_mean is None and bn.running_var is None))
            and torch.is_grad_enabled() and x.requires_grad):
        x = x.contiguous()
        momentum, counter = _count_batch(bn)
        rm = bn.running_mean if (bn.training and bn.track_running_stats) else None
        rv = bn.running_var if (bn.training and bn.track_running_stats) else None
        return _BNReLUTrain.apply(x, bn.weight, bn.bias, rm, rv, momentum, bn.eps, True, True, _ptr(counter),
                                  take_stats(x) if bn.training else None)
    return bn_relu(bn, x), x


def bn_relu(bn, x, relu=True):
    """`relu(bn(x))` (or `bn(x)` with relu=False) for an nn.BatchNorm2d module `bn`."""
    if not _fusable(bn, x):
        y = bn(x)
        return F.relu(y, inplace=True) if relu else y
    use_batch_stats = bn.training or (bn.running_mean is None and bn.running_var is None)
    x = x.contiguous()
    if use_batch_stats:
        momentum, counter = _count_batch(bn)
        rm = bn.running_mean if (bn.training and bn.track_running_stats) else None
        rv = bn.running_var if (bn.training and bn.track_running_stats) else None
        return _BNReLUTrain.apply(x, bn.weight, bn.bias, rm, rv, momentum, bn.eps, relu, False, _ptr(counter),
                                  take_stats(x) if bn.training else None)
    if torch.is_grad_enabled() and (x.requires_grad or bn.weight.requires_grad):
        y = bn(x)                                  # frozen-statistics fine-tuning: stock kernels
        return F.relu(y, inplace=True) if relu else y
    return _eval_forward(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, relu)
