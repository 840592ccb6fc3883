"""Training step of a RubiksShiftBlock with its BatchNorms folded into the neighbouring kernels.

The reference block (rubiksnet/backbone.py:123-135)

    a1 = relu(bn1(x));  z = conv2(a1);  a2 = relu(bn2(z));  s = as3(a2);  out = conv3(s) + shortcut(x or a1)

costs, layer by layer, 13 passes over an activation forward and 22 backward.  `fused_train_block(block, x)` evaluates
the same block -- same modules, parameters, buffers, state-dict, same gradients -- as ONE autograd node in which

  * every BatchNorm's statistics pass is the epilogue of the GEMM that produces its input (conv2 -> bn2; conv3 +
    shortcut -> the NEXT block's bn1: the tile partials travel with the output tensor as `out._rk_stats`),
  * relu(bn1(x)) is never stored: it is the operand prologue of conv2, of the projecting shortcut and of their d(weight)
    kernels, all of which read x,
  * bn1's backward reduction rides on conv2's d(input) GEMM (which masks its result with the ReLU on the way out), and
    the identity shortcut's gradient is added inside bn1's d(x) pass,
  * the residual add is conv3's epilogue.

Qualifies: training mode, fp32 CUDA tensors, the 3-D shift variant (with or without the Small tier's SE gate), planes with H*W % 4 == 0 on both
sides of the shift (every block of the networks except the two that touch 7x7 planes).  Anything else: None, and the
caller runs the layer-by-layer path.  `RK_FUSED_TRAIN=0` switches it off.
"""
import logging

import torch

from . import _native, config, rubiksnet_cuda
from .fused_bn import _count_batch

__all__ = ["fused_train_block", "take_stats", "bn_relu_from_stats", "stats_fallbacks"]

_LOG = logging.getLogger("rubiksnet_amd")
_FALLBACKS = [0]


def stats_fallbacks():
    """How many fused blocks of this process had to run their own statistics pass over x because the producer's tile
    statistics did not arrive with the tensor (see _note_fallback)."""
    return _FALLBACKS[0]


def _note_fallback(shape):
    """The tile statistics travel as an attribute of the producing block's output tensor; anything between two blocks
    that makes a new tensor object (a DDP / checkpoint hook, `.contiguous()` on a view, an in-place edit) drops them and
    this block pays one extra read of x.  Correct, but a silent performance cliff -- so it is counted, and said once."""
    _FALLBACKS[0] += 1
    if _FALLBACKS[0] == 1:
        _LOG.warning("rubiksnet_amd: a fused training block received x %s without its producer's BatchNorm tile statistics "
                     "and runs a statistics pass of its own (logged once; train_block.stats_fallbacks() counts them)",
                     tuple(shape))

_CMAX = 320          # register tile of the GEMM kernels (as the inference fusion)


def _stream(dev):
    return torch.cuda.current_stream(dev).cuda_stream


_SIDE = {}        # device index -> the HIP stream the d(weight) kernels of a backward run on


def _side_stream(dev):
    """d(weight) has no consumer inside the block's backward, is MFMA / latency bound (its waves sit in s_waitcnt 60 % of
    the time, PMC) and runs next to pure streaming kernels (BatchNorm d(x), the shift backward): launched on a second
    stream it fills their gaps.  The backward joins the streams before it returns, so autograd sees finished gradients."""
    if not config.switches().wgrad_overlap:
        return None
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    st = _SIDE.get(key)
    if st is None:
        st = _SIDE[key] = torch.cuda.Stream(dev)
    return st


def _ptr(t):
    return t.data_ptr() if t is not None else None


def take_stats(x, channels, count):
    """The tile statistics a producing GEMM attached to `x` (float4 [C][tiles]), if they describe exactly this tensor."""
    st = getattr(x, "_rk_stats", None)
    if st is None:
        return None
    # (tiles of 128 / 64 columns or a workgroup's column halves, by kernel generation: every tile carries its own count)
    if (st.dim() != 3 or st.shape[0] != channels or st.shape[1] < 1 or st.shape[2] != 4 or st.device != x.device
            or getattr(x, "_rk_stats_count", None) != count or getattr(x, "_rk_stats_version", None) != x._version):
        return None
    return st


def _attach_stats(t, stats):
    t._rk_stats = stats
    t._rk_stats_count = t.numel() // t.shape[1]        # elements per channel the tiles add up to
    t._rk_stats_version = t._version       # an in-place edit of the tensor invalidates them
    return t


def _tile_stats(L, x, Fr, C, P):
    J = int(L.rk_pw_tiles(Fr, P))
    st = torch.empty(C, J, 4, dtype=torch.float32, device=x.device)
    _native.check(L.rk_bn_tile_stats_f32(x.data_ptr(), st.data_ptr(), Fr, C, P, _stream(x.device)), "rk_bn_tile_stats_f32")
    return st


def _finish(L, bn, stats, count, dev):
    """Tile partials -> (save_mean, save_invstd, a, b) of `bn`'s training forward; running statistics and
    num_batches_tracked updated exactly as nn.BatchNorm2d does."""
    C = bn.num_features
    out = torch.empty(8, C, dtype=torch.float32, device=dev)         # rows 0-3: mean, invstd, a, b; rows 4-7: [C][4] packed
    momentum, counter = _count_batch(bn)
    tracked = bn.training and bn.track_running_stats
    rm = bn.running_mean if tracked else None
    rv = bn.running_var if tracked else None
    rc = L.rk_bn_finish_tiles_f32(stats.data_ptr(), stats.shape[1], count, bn.weight.data_ptr(), bn.bias.data_ptr(),
                                  _ptr(rm), _ptr(rv), out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(),
                                  out[3].data_ptr(), out[4].data_ptr(), C, float(bn.eps), float(momentum), _ptr(counter),
                                  _stream(dev))
    _native.check(rc, "rk_bn_finish_tiles_f32")
    return out            # rows: mean, invstd, a, b, then (a, b, mean, invstd) per channel packed as [C][4]


def _bn_ok(bn):
    return (isinstance(bn, torch.nn.BatchNorm2d) and bn.training and bn.affine and bn.weight.dtype == torch.float32
            and bn.weight.is_cuda and (bn.running_mean is None or bn.running_mean.dtype == torch.float32))


def _conv_ok(conv, stride):
    return (isinstance(conv, torch.nn.Conv2d) and conv.kernel_size == (1, 1) and conv.stride == (stride, stride)
            and conv.padding == (0, 0) and conv.groups == 1 and conv.bias is None and conv.weight.dtype == torch.float32
            and conv.in_channels % 2 == 0 and conv.out_channels % 2 == 0
            and max(conv.in_channels, conv.out_channels) <= _CMAX)


def _shift_config(as3):
    """(RubiksShift3D module, n_segment, spatial stride) of a block's temporal shift wrapper, or None."""
    from .shiftlib import RubiksShift3D

    layer, T = getattr(as3, "rubiks3d", None), getattr(as3, "n_segment", None)
    if not isinstance(layer, RubiksShift3D) or not isinstance(T, int):
        return None
    three = lambda v: tuple(int(e) for e in ((v,) * 3 if isinstance(v, int) else v))      # noqa: E731
    st, pd = three(layer.stride), three(layer.padding)
    if pd != (0, 0, 0) or st not in ((1, 1, 1), (1, 2, 2)) or layer.shift.dtype != torch.float32:
        return None
    if not isinstance(layer.normalize_grad, bool):
        return None
    return layer, T, st[1]


class _Plan:
    """Everything of a block that is not a tensor argument of the autograd node."""

    def __init__(self, block, x, shift_cfg):
        self.block = block
        self.layer, self.T, self.stride = shift_cfg
        self.identity = isinstance(block.shortcut, torch.nn.Identity)
        Fr, Cin, H, W = x.shape
        self.Fr, self.Cin, self.H, self.W = Fr, Cin, H, W
        self.Cmid = block.conv2.out_channels
        self.Cout = block.conv3.out_channels
        self.Ho, self.Wo = (H - 1) // self.stride + 1, (W - 1) // self.stride + 1
        t = self.layer.normalize_t_factor
        self.t_factor = self.T / H if t == "auto" else float(t)


class _FusedTrainBlock(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, stats_in, g1, b1, w2, g2, b2, shift, w3, wsc, wse1, wse2, plan):
        L = _native.lib()
        blk, dev = plan.block, x.device
        Fr, Cin, H, W, Cmid, Cout, Ho, Wo = plan.Fr, plan.Cin, plan.H, plan.W, plan.Cmid, plan.Cout, plan.Ho, plan.Wo
        P, Po = H * W, Ho * Wo
        s3, pd = [1, plan.stride, plan.stride], [0, 0, 0]
        with torch.cuda.device(dev):
            side = None
            try:
                st = _stream(dev)
                if stats_in is None:
                    _note_fallback(x.shape)
                    stats_in = _tile_stats(L, x, Fr, Cin, P)
                bn1 = _finish(L, blk.bn1, stats_in, Fr * P, dev)                       # mean, invstd, a, b
                # shortcut branch: x itself, or the projection of relu(bn1(x)) (prologue in the operand load).  The projection
                # has no consumer before conv3's epilogue: it runs on the second stream next to conv2 and the shift.
                if plan.identity:
                    short = x
                else:
                    side = _side_stream(dev)
                    cur = torch.cuda.current_stream(dev)
                    sst = st
                    if side is not None:
                        side.wait_stream(cur)
                        sst = side.cuda_stream
                    if plan.stride == 2:
                        short = torch.empty(Fr, Cout, Ho, Wo, dtype=x.dtype, device=dev)
                        _native.check(L.rk_pw_s2_forward_fused_f32(wsc.data_ptr(), x.data_ptr(), short.data_ptr(), Fr, Cin, Cout,
                                                                   H, W, bn1[2].data_ptr(), bn1[3].data_ptr(), 1, sst),
                                      "rk_pw_s2_forward_fused_f32")
                    else:
                        short = torch.empty(Fr, Cout, H, W, dtype=x.dtype, device=dev)
                        _native.check(L.rk_pw_gemm_fused_f32(wsc.data_ptr(), x.data_ptr(), None, short.data_ptr(), Fr, Cin, Cout,
                                                             P, 1, bn1[2].data_ptr(), bn1[3].data_ptr(), 1, None, None, 0, sst),
                                      "rk_pw_gemm_fused_f32")
                # conv2 on relu(bn1(x)), + the statistics of its output for bn2
                z = torch.empty(Fr, Cmid, H, W, dtype=x.dtype, device=dev)
                J = int(L.rk_pw_gemm_tiles(w2.data_ptr(), Fr, Cin, Cmid, P, 1))
                stats2 = torch.empty(Cmid, J, 4, dtype=torch.float32, device=dev)
                _native.check(L.rk_pw_gemm_stats_f32(w2.data_ptr(), x.data_ptr(), None, z.data_ptr(), Fr, Cin, Cmid, P, 1,
                                                     bn1[2].data_ptr(), bn1[3].data_ptr(), 1, stats2.data_ptr(), J, st),
                              "rk_pw_gemm_stats_f32")
                bn2 = _finish(L, blk.bn2, stats2, Fr * P, dev)
                # the shift of relu(bn2(z)), on [N, T, C, H, W] views of the same memory.  Fused (the kernels normalise the
                # landed planes in LDS: the activation is never stored) where a fused kernel exists, else normalise + shift.
                N = Fr // plan.T
                s = torch.empty(Fr, Cmid, Ho, Wo, dtype=x.dtype, device=dev)
                shift_c = shift.detach().contiguous()
                abmi2 = bn2[4:].view(Cmid, 4)
                a2 = None
                rc = rubiksnet_cuda.rubiks_shift_3d_forward_bn_float(z.view(N, plan.T, Cmid, H, W), abmi2, shift_c, s3, pd,
                                                                     bool(plan.layer.quantize), s.view(N, plan.T, Cmid, Ho, Wo))
                if rc == rubiksnet_cuda.UNSUPPORTED:
                    a2 = torch.empty_like(z)
                    _native.check(L.rk_bn_apply_affine_f32(z.data_ptr(), bn2[2].data_ptr(), bn2[3].data_ptr(), a2.data_ptr(),
                                                           Fr, Cmid, P, 1, st), "rk_bn_apply_affine_f32")
                    rubiksnet_cuda.rubiks_shift_3d_forward_float(a2.view(N, plan.T, Cmid, H, W), shift_c, s3, pd,
                                                                 bool(plan.layer.quantize), s.view(N, plan.T, Cmid, Ho, Wo))
                # SE gate (Small tier, backbone.py:56-71, :131-132): squeeze = one pass over s, the two tiny Linear layers in
                # PyTorch (no autograd here: their backward is written out below), scale = one pass
                se_q = se_h = se_g = s_in = None
                if wse1 is not None:
                    se_q = torch.empty(Fr, Cmid, dtype=torch.float32, device=dev)
                    _native.check(L.rk_se_squeeze_f32(s.data_ptr(), se_q.data_ptr(), Fr, Cmid, Po, st), "rk_se_squeeze_f32")
                    Cr = wse1.shape[0]
                    se_h = torch.empty(Fr, Cr, dtype=torch.float32, device=dev)
                    se_g = torch.empty(Fr, Cmid, dtype=torch.float32, device=dev)
                    rc = L.rk_se_mlp_forward_f32(se_q.data_ptr(), wse1.data_ptr(), wse2.data_ptr(), se_h.data_ptr(), se_g.data_ptr(),
                                                 Fr, Cmid, Cr, st)
                    if rc == rubiksnet_cuda.UNSUPPORTED:          # (wider than the kernel's tables: the two Linear layers in PyTorch)
                        se_h = torch.relu(se_q @ wse1.t())
                        se_g = torch.sigmoid(se_h @ wse2.t()).contiguous()
                    else:
                        _native.check(rc, "rk_se_mlp_forward_f32")
                    s_in = torch.empty_like(s)
                    _native.check(L.rk_se_scale_f32(s.data_ptr(), se_g.data_ptr(), s_in.data_ptr(), Fr, Cmid, Po, st),
                                  "rk_se_scale_f32")
                conv3_in = s_in if s_in is not None else s
                # conv3 + shortcut, + the statistics of the block's output for whoever normalises it next
                if side is not None:
                    torch.cuda.current_stream(dev).wait_stream(side)      # the projection is complete
                out = torch.empty(Fr, Cout, Ho, Wo, dtype=x.dtype, device=dev)
                Jo = int(L.rk_pw_gemm_tiles(w3.data_ptr(), Fr, Cmid, Cout, Po, 1))
                stats_out = torch.empty(Cout, Jo, 4, dtype=torch.float32, device=dev)
                _native.check(L.rk_pw_gemm_stats_f32(w3.data_ptr(), conv3_in.data_ptr(), short.data_ptr(), out.data_ptr(), Fr, Cmid,
                                                     Cout, Po, 1, None, None, 0, stats_out.data_ptr(), Jo, st),
                              "rk_pw_gemm_stats_f32")
            finally:
                # an exception between the fork and the join must not let the allocator hand out buffers the side
                # stream still uses: the join always runs
                if side is not None:
                    torch.cuda.current_stream(dev).wait_stream(side)
        ctx.plan = plan
        ctx.has_a2 = a2 is not None
        ctx.has_se = wse1 is not None
        se_saved = (s_in, se_q, se_h, se_g, wse1, wse2) if ctx.has_se else ()
        ctx.save_for_backward(x, z, a2 if a2 is not None else z, s, bn1, bn2, g1, g2, b2, w2, w3,
                              wsc if wsc is not None else w3, shift_c, *se_saved)
        ctx.mark_non_differentiable(stats_out)
        # (or autograd hands backward() a freshly zero-filled tensor of stats_out's shape for `_dstats`: one fill kernel per block
        # and step -- 51 launches of ~6 us in RubiksNet-Large's kernel table)
        ctx.set_materialize_grads(False)
        return out, stats_out

    @staticmethod
    def backward(ctx, dout, _dstats):
        plan = ctx.plan
        x, z, a2, s, bn1, bn2, g1, g2, b2, w2, w3, wsc, shift = ctx.saved_tensors[:13]
        s_in = se_q = se_h = se_g = wse1 = wse2 = None
        if ctx.has_se:
            s_in, se_q, se_h, se_g, wse1, wse2 = ctx.saved_tensors[13:]
        dwse1 = dwse2 = None
        L = _native.lib()
        dev = x.device
        Fr, Cin, H, W, Cmid, Cout, Ho, Wo = plan.Fr, plan.Cin, plan.H, plan.W, plan.Cmid, plan.Cout, plan.Ho, plan.Wo
        P, Po = H * W, Ho * Wo
        s3, pd = [1, plan.stride, plan.stride], [0, 0, 0]
        if dout is None:                                    # (the block's output took no part in the loss)
            return (None,) * 13
        dout = dout.contiguous()
        need = ctx.needs_input_grad
        with torch.cuda.device(dev):
            st = _stream(dev)
            cur = torch.cuda.current_stream(dev)
            side = _side_stream(dev)
            keep = []                                       # buffers the side stream uses: alive until the streams have joined

            def wgrad_ws(K, M, Pn):
                nbytes = int(L.rk_pw_wgrad_workspace_bytes(Fr, K, M, Pn))
                buf = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=dev)
                keep.append(buf)
                return buf, nbytes

            def wg_stream():
                """Stream handle for a d(weight) launch: the side stream (after everything queued so far), or the current one."""
                if side is None:
                    return st
                side.wait_stream(cur)
                return side.cuda_stream

            try:
                # conv3: d(s) = W3^T dout, d(W3) = dout s^T
                ds = torch.empty_like(s)
                _native.check(L.rk_pw_gemm_f32(w3.data_ptr(), dout.data_ptr(), None, ds.data_ptr(), Fr, Cout, Cmid, Po, 0, st),
                              "rk_pw_gemm_f32")
                dw3 = None
                if need[8]:
                    dw3 = torch.empty_like(w3)
                    ws, nb = wgrad_ws(Cmid, Cout, Po)
                    _native.check(L.rk_pw_wgrad_f32(dout.data_ptr(), (s_in if ctx.has_se else s).data_ptr(), dw3.data_ptr(), Fr,
                                                    Cmid, Cout, Po, ws.data_ptr(), nb, wg_stream()), "rk_pw_wgrad_f32")
                if ctx.has_se:
                    # SE backward: ds (so far d(s * gate)) -> d(s) = ds gate + d(squeeze) / HW, d(gate) = sum_hw(ds s) in one
                    # pass over (ds, s); the two Linear layers and their activations by hand (a few [F, C]-sized kernels)
                    # (4 tensor passes: the reduction alone, then d(s) with the squeeze's share added in the same pass --
                    # scale_backward + a broadcast add were 5)
                    dgate = torch.empty_like(se_g)
                    _native.check(L.rk_se_dgate_f32(ds.data_ptr(), s.data_ptr(), dgate.data_ptr(), Fr, Cmid, Po, st),
                                  "rk_se_dgate_f32")
                    Cr = wse1.shape[0]
                    dpre2 = torch.empty_like(se_g)
                    dpre1 = torch.empty_like(se_h)
                    dq = torch.empty_like(se_g)
                    dwse1, dwse2 = torch.empty_like(wse1), torch.empty_like(wse2)
                    rc = L.rk_se_mlp_backward_f32(dgate.data_ptr(), se_g.data_ptr(), se_h.data_ptr(), se_q.data_ptr(), wse1.data_ptr(),
                                                  wse2.data_ptr(), dpre2.data_ptr(), dpre1.data_ptr(), dq.data_ptr(), dwse1.data_ptr(),
                                                  dwse2.data_ptr(), Fr, Cmid, Cr, st)
                    if rc == rubiksnet_cuda.UNSUPPORTED:
                        dpre2 = dgate * se_g * (1.0 - se_g)
                        dwse2 = dpre2.t() @ se_h
                        dpre1 = (dpre2 @ wse2) * (se_h > 0).to(dpre2.dtype)
                        dwse1 = dpre1.t() @ se_q
                        dq = (dpre1 @ wse1).contiguous()
                    else:
                        _native.check(rc, "rk_se_mlp_backward_f32")
                    dsg = torch.empty_like(s)
                    _native.check(L.rk_se_scale_add_f32(ds.data_ptr(), se_g.data_ptr(), dq.data_ptr(), 1.0 / float(Po),
                                                        dsg.data_ptr(), Fr, Cmid, Po, st), "rk_se_scale_add_f32")
                    ds = dsg
                # the shift and bn2: d(shift), and d(z) = bn2 + ReLU backward of d(a2).  Fused: the shift backward reads z,
                # masks its d(x) with the ReLU and reduces bn2's sums in the same launch (dg2, db2, k12); one d(x) pass finishes.
                N = Fr // plan.T
                dshift = torch.empty_like(shift)
                dz = torch.empty_like(z)
                dg2 = torch.empty(Cmid, dtype=torch.float32, device=dev)
                db2 = torch.empty(Cmid, dtype=torch.float32, device=dev)
                rc = rubiksnet_cuda.UNSUPPORTED
                if not ctx.has_a2:
                    k12b = torch.empty(2, Cmid, dtype=torch.float32, device=dev)
                    dzm = dz                                  # masked d(a2) first, finished in place by the d(x) pass
                    rc = rubiksnet_cuda.rubiks_shift_3d_backward_bn_float(
                        z.view(N, plan.T, Cmid, H, W), bn2[4:].view(Cmid, 4), shift, ds.view(N, plan.T, Cmid, Ho, Wo), s3, pd,
                        dzm.view(N, plan.T, Cmid, H, W), dshift, k12b, dg2, db2, plan.layer.normalize_grad, plan.t_factor,
                        bool(plan.layer.quantize))
                    if rc == 0:
                        _native.check(L.rk_bn_bwd_dx_pre_f32(dzm.data_ptr(), z.data_ptr(), g2.data_ptr(), bn2[0].data_ptr(),
                                                             bn2[1].data_ptr(), k12b.data_ptr(), None, dz.data_ptr(), Fr, Cmid,
                                                             P, st), "rk_bn_bwd_dx_pre_f32")
                    else:                                     # (no fused backward for a shape the forward took: recompute a2)
                        a2 = torch.empty_like(z)
                        _native.check(L.rk_bn_apply_affine_f32(z.data_ptr(), bn2[2].data_ptr(), bn2[3].data_ptr(),
                                                               a2.data_ptr(), Fr, Cmid, P, 1, st), "rk_bn_apply_affine_f32")
                if rc != 0:
                    da2 = torch.empty_like(a2)
                    rubiksnet_cuda.rubiks_shift_3d_backward_float(
                        a2.view(N, plan.T, Cmid, H, W), shift, ds.view(N, plan.T, Cmid, Ho, Wo), s3, pd,
                        da2.view(N, plan.T, Cmid, H, W), dshift, plan.layer.normalize_grad, plan.t_factor,
                        bool(plan.layer.quantize))
                    nbn = int(L.rk_bn_workspace_bytes(Fr, Cmid, P))
                    wsb = torch.empty(max(nbn, 1), dtype=torch.uint8, device=dev)
                    _native.check(L.rk_bn_relu_backward_f32(da2.data_ptr(), z.data_ptr(), g2.data_ptr(), b2.data_ptr(),
                                                            bn2[0].data_ptr(), bn2[1].data_ptr(), None, dz.data_ptr(),
                                                            dg2.data_ptr(), db2.data_ptr(), Fr, Cmid, P, 1, wsb.data_ptr(), nbn,
                                                            st), "rk_bn_relu_backward_f32")
                    del da2
                if not need[7]:
                    dshift = None
                # the projecting shortcut's share of d(relu(bn1(x))), handed to conv2's d(input) GEMM as its residual
                res, dwsc = None, None
                if not plan.identity:
                    res = torch.empty_like(x)
                    if plan.stride == 2:
                        _native.check(L.rk_pw_s2_dgrad_f32(wsc.data_ptr(), dout.data_ptr(), res.data_ptr(), Fr, Cin, Cout, H, W,
                                                           st), "rk_pw_s2_dgrad_f32")
                    else:
                        _native.check(L.rk_pw_gemm_f32(wsc.data_ptr(), dout.data_ptr(), None, res.data_ptr(), Fr, Cout, Cin, P,
                                                       0, st), "rk_pw_gemm_f32")
                    if need[9]:
                        dwsc = torch.empty_like(wsc)
                        ws, nb = wgrad_ws(Cin, Cout, Po)
                        if plan.stride == 2:
                            _native.check(L.rk_pw_s2_wgrad_pro_f32(dout.data_ptr(), x.data_ptr(), dwsc.data_ptr(), Fr, Cin, Cout,
                                                                   H, W, bn1[2].data_ptr(), bn1[3].data_ptr(), 1, ws.data_ptr(),
                                                                   nb, wg_stream()), "rk_pw_s2_wgrad_pro_f32")
                        else:
                            _native.check(L.rk_pw_wgrad_pro_f32(dout.data_ptr(), x.data_ptr(), dwsc.data_ptr(), Fr, Cin, Cout, P,
                                                                bn1[2].data_ptr(), bn1[3].data_ptr(), 1, ws.data_ptr(), nb,
                                                                wg_stream()), "rk_pw_wgrad_pro_f32")
                # conv2: d(W2) from (dz, relu(bn1(x)) recomputed); d(input) masked by the ReLU, + bn1's reduction sums
                dw2 = None
                if need[4]:
                    dw2 = torch.empty_like(w2)
                    ws, nb = wgrad_ws(Cin, Cmid, P)
                    _native.check(L.rk_pw_wgrad_pro_f32(dz.data_ptr(), x.data_ptr(), dw2.data_ptr(), Fr, Cin, Cmid, P,
                                                        bn1[2].data_ptr(), bn1[3].data_ptr(), 1, ws.data_ptr(), nb, wg_stream()),
                                  "rk_pw_wgrad_pro_f32")
                J = int(L.rk_pw_gemm_tiles(w2.data_ptr(), Fr, Cmid, Cin, P, 0))
                bred = torch.empty(Cin, J, 2, dtype=torch.float32, device=dev)
                dzm = res if res is not None else torch.empty_like(x)             # (the residual may alias the result)
                _native.check(L.rk_pw_gemm_bnbwd_f32(w2.data_ptr(), dz.data_ptr(), _ptr(res), dzm.data_ptr(), Fr, Cmid, Cin, P,
                                                     0, x.data_ptr(), bn1[4].data_ptr(), bred.data_ptr(), J, st),
                              "rk_pw_gemm_bnbwd_f32")
                k12 = torch.empty(2, Cin, dtype=torch.float32, device=dev)
                dg1 = torch.empty(Cin, dtype=torch.float32, device=dev)
                db1 = torch.empty(Cin, dtype=torch.float32, device=dev)
                _native.check(L.rk_bn_bwd_finish_tiles_f32(bred.data_ptr(), J, Fr * P, k12.data_ptr(), dg1.data_ptr(),
                                                           db1.data_ptr(), Cin, st), "rk_bn_bwd_finish_tiles_f32")
                dx = None
                if need[0]:
                    dx = torch.empty_like(x)
                    skip = dout if plan.identity else None                        # identity shortcut: out = ... + x
                    _native.check(L.rk_bn_bwd_dx_pre_f32(dzm.data_ptr(), x.data_ptr(), g1.data_ptr(), bn1[0].data_ptr(),
                                                         bn1[1].data_ptr(), k12.data_ptr(), _ptr(skip), dx.data_ptr(), Fr, Cin,
                                                         P, st), "rk_bn_bwd_dx_pre_f32")
            finally:
                if side is not None:
                    cur.wait_stream(side)                   # always: gradients complete (and the side stream's buffers
                del keep                                    # released) even when a launch raised in between
        return (dx, None, dg1, db1, dw2, dg2, db2, dshift, dw3, dwsc if not plan.identity else None, dwse1, dwse2, None)


def fused_train_block(block, x):
    """`block(x)` in training mode through the fused node, or None when the block does not qualify."""
    sw = config.switches()
    if not (sw.fused_train and sw.fused_bn and sw.pointwise != "0"):
        return None
    if not (block.training and torch.is_grad_enabled() and not torch.is_autocast_enabled()):
        return None
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.numel() > 0):
        return None
    se1 = se2 = None
    if block.se is not None:                # the stock SELayer (two bias-free Linear layers, ReLU, Sigmoid) only
        fc = getattr(block.se, "fc", None)
        if not (isinstance(fc, torch.nn.Sequential) and len(fc) == 4 and isinstance(fc[0], torch.nn.Linear)
                and isinstance(fc[1], torch.nn.ReLU) and isinstance(fc[2], torch.nn.Linear)
                and isinstance(fc[3], torch.nn.Sigmoid) and fc[0].bias is None and fc[2].bias is None
                and fc[0].weight.dtype == torch.float32 and fc[0].weight.is_cuda):
            return None
        se1, se2 = fc[0].weight, fc[2].weight
    cfg = _shift_config(block.as3)
    if cfg is None:
        return None
    _, T, stride = cfg
    Fr, Cin, H, W = x.shape
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    identity = isinstance(block.shortcut, torch.nn.Identity)
    if (H * W) % 4 or (Ho * Wo) % 4 or Fr % T:
        return None
    if not (_bn_ok(block.bn1) and _bn_ok(block.bn2) and _conv_ok(block.conv2, 1) and _conv_ok(block.conv3, 1)):
        return None
    if block.conv2.in_channels != Cin or block.bn1.num_features != Cin:
        return None
    if identity:
        if stride != 1 or block.conv3.out_channels != Cin:
            return None
    else:
        if not _conv_ok(block.shortcut, stride):
            return None
        if stride == 2 and (H % 2 or W % 2):
            return None
    stats_in = take_stats(x, Cin, Fr * H * W)
    x = x.contiguous()
    plan = _Plan(block, x, cfg)
    out, stats_out = _FusedTrainBlock.apply(
        x, stats_in, block.bn1.weight, block.bn1.bias, block.conv2.weight, block.bn2.weight, block.bn2.bias,
        cfg[0].shift, block.conv3.weight, None if identity else block.shortcut.weight, se1, se2, plan)
    return _attach_stats(out, stats_out)


def bn_relu_from_stats(bn, x, relu=True):
    """`relu(bn(x))` in training mode when `x` carries the tile statistics of its producer (the last block's output in
    front of `bn_last`): the statistics pass is skipped.  None when it does not apply."""
    sw = config.switches()
    if not (sw.fused_train and sw.fused_bn) or not (bn.training and torch.is_grad_enabled()):
        return None
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and _bn_ok(bn)) or torch.is_autocast_enabled():
        return None
    Fr, C, H, W = x.shape
    stats = take_stats(x, C, Fr * H * W)
    if stats is None or bn.num_features != C or not x.is_contiguous():
        return None
    return _BNFromStats.apply(x, stats, bn.weight, bn.bias, bn, relu)


class _BNFromStats(torch.autograd.Function):
    """Training-mode relu?(bn(x)) from given tile statistics; backward = the ordinary fused BN backward."""

    @staticmethod
    def forward(ctx, x, stats, weight, bias, bn, relu):
        L = _native.lib()
        dev = x.device
        Fr, C, H, W = x.shape
        with torch.cuda.device(dev):
            fin = _finish(L, bn, stats, Fr * H * W, dev)
            y = torch.empty_like(x)
            _native.check(L.rk_bn_apply_affine_f32(x.data_ptr(), fin[2].data_ptr(), fin[3].data_ptr(), y.data_ptr(), Fr, C,
                                                   H * W, int(relu), _stream(dev)), "rk_bn_apply_affine_f32")
        ctx.save_for_backward(x, weight, bias, fin)
        ctx.relu = relu
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, bias, fin = ctx.saved_tensors
        L = _native.lib()
        dev = x.device
        Fr, C, H, W = x.shape
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        dg = torch.empty(C, dtype=torch.float32, device=dev)
        db = torch.empty(C, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            nb = int(L.rk_bn_workspace_bytes(Fr, C, H * W))
            ws = torch.empty(max(nb, 1), dtype=torch.uint8, device=dev)
            _native.check(L.rk_bn_relu_backward_f32(dy.data_ptr(), x.data_ptr(), weight.data_ptr(), bias.data_ptr(),
                                                    fin[0].data_ptr(), fin[1].data_ptr(), None, dx.data_ptr(), dg.data_ptr(),
                                                    db.data_ptr(), Fr, C, H * W, int(ctx.relu), ws.data_ptr(), nb,
                                                    _stream(dev)), "rk_bn_relu_backward_f32")
        return dx, None, dg, db, None, None
