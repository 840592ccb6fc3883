"""GPU tests of the BASELINE.json model-level configs, with the shift operators checked against the
oracle on the activations and gradients they actually see inside the network:

  configs[2]  RubiksNet-Tiny (rubiks3d) full forward, batch 64                (2 clips of the batch checked)
  configs[3]  RubiksNet-Large (rubiks3d) train step: forward + backward + Adam (per-GPU share of the DP batch)
  configs[4]  RubiksNet-Large-AQ (rubiks3d-aq) under bf16 autocast, train step

The taps sit on the drop-in boundary itself -- the six callables of rubiksnet_amd.rubiksnet_cuda
(cuda_src/rubiks.cpp:384-396) and the temporal 3-tap autograd Function of AttentionShift -- so what is
compared is exactly what librubiks_hip.so was handed and what it wrote.  Bars: y and d(x) bit-exact
(fp32), or equal to the fp32 oracle on the widened inputs rounded once (bf16 storage); d(shift) / d(taps)
against the oracle evaluated in fp64.  Reference call sites: rubiksnet/models.py:67-110,
scripts/test_installation.py:6-10, scripts/example_finetune.py:85-97.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

# (C, H, stride_hw) of the 9 distinct RubiksShift3D call shapes per tier width
def _expected_shapes(width):
    w = width
    return {(w, 112, 1), (w, 112, 2), (w, 56, 1), (2 * w, 56, 2), (2 * w, 28, 1), (4 * w, 28, 2), (4 * w, 14, 1),
            (8 * w, 14, 2), (8 * w, 7, 1)}


def _np(t):
    return t.detach().float().cpu().numpy() if t.dtype in (torch.bfloat16, torch.float16) else t.detach().cpu().numpy()


class _Taps:
    """Records the first call per distinct (shape, stride, dtype) at the binding level."""

    def __init__(self, monkeypatch, keep_clips=None):
        from rubiksnet_amd import rubiksnet_cuda as rc
        from rubiksnet_amd.attention_shift import _TemporalShift3Func as TS

        self.f3, self.b3, self.f2, self.b2, self.fa, self.ba = {}, {}, {}, {}, {}, {}
        self.calls = dict(f3=0, b3=0, f2=0, b2=0, fa=0, ba=0)
        k = keep_clips
        f3, b3, f2, b2 = (rc.rubiks_shift_3d_forward_float, rc.rubiks_shift_3d_backward_float, rc.rubiks2d_forward,
                          rc.rubiks2d_backward)
        tsf, tsb = TS.forward, TS.backward
        taps = self

        def fwd3(input, shift, strides, paddings, quantize, output):
            ret = f3(input, shift, strides, paddings, quantize, output)
            taps.calls["f3"] += 1
            key = (tuple(input.shape), tuple(strides))
            if key not in taps.f3:
                taps.f3[key] = dict(x=_np(input[:k]), shift=_np(shift), s=list(strides), p=list(paddings), q=bool(quantize),
                                    y=_np(output[:k]))
            return ret

        def bwd3(input, shift, output_grad, strides, paddings, input_grad, shift_grad, normalize_grad,
                 normalize_t_factor, quantize):
            ret = b3(input, shift, output_grad, strides, paddings, input_grad, shift_grad, normalize_grad,
                     normalize_t_factor, quantize)
            taps.calls["b3"] += 1
            key = (tuple(input.shape), tuple(strides))
            if key not in taps.b3:
                taps.b3[key] = dict(x=_np(input), shift=_np(shift), gy=_np(output_grad), s=list(strides), p=list(paddings),
                                    q=bool(quantize), norm=bool(normalize_grad), tf=float(normalize_t_factor),
                                    gx=None if input_grad is None else _np(input_grad),
                                    gs=None if shift_grad is None else _np(shift_grad))
            return ret

        # the training fusion's shift entry points (train_block.py): same records, with x = relu(bn(z)) materialised by
        # the stand-alone normalise kernel -- so the checks below read "normalise, then the oracle's shift"
        fb3, bb3 = rc.rubiks_shift_3d_forward_bn_float, rc.rubiks_shift_3d_backward_bn_float

        def _act(z, abmi):
            from rubiksnet_amd import _native
            N, T, C, H, W = z.shape
            a, b = abmi[:, 0].contiguous(), abmi[:, 1].contiguous()
            out = torch.empty_like(z)
            _native.check(_native.lib().rk_bn_apply_affine_f32(z.data_ptr(), a.data_ptr(), b.data_ptr(), out.data_ptr(), N * T,
                                                               C, H * W, 1, torch.cuda.current_stream().cuda_stream), "apply")
            return out

        def fwd3bn(input, abmi, shift, strides, paddings, quantize, output):
            ret = fb3(input, abmi, shift, strides, paddings, quantize, output)
            if ret != 0:
                return ret                                  # not fused: the plain entry point follows (and is tapped)
            taps.calls["f3"] += 1
            taps.fused_f3 += 1
            key = (tuple(input.shape), tuple(strides))
            if key not in taps.f3:
                taps.f3[key] = dict(x=_np(_act(input, abmi)[:k]), shift=_np(shift), s=list(strides), p=list(paddings),
                                    q=bool(quantize), y=_np(output[:k]))
            return ret

        def bwd3bn(input, abmi, shift, output_grad, strides, paddings, input_grad, shift_grad, k12, dgamma, dbeta,
                   normalize_grad, normalize_t_factor, quantize):
            ret = bb3(input, abmi, shift, output_grad, strides, paddings, input_grad, shift_grad, k12, dgamma, dbeta,
                      normalize_grad, normalize_t_factor, quantize)
            if ret != 0:
                return ret
            taps.calls["b3"] += 1
            key = (tuple(input.shape), tuple(strides))
            if key not in taps.b3:
                a2 = _act(input, abmi)
                taps.b3[key] = dict(x=_np(a2), shift=_np(shift), gy=_np(output_grad), s=list(strides), p=list(paddings),
                                    q=bool(quantize), norm=bool(normalize_grad), tf=float(normalize_t_factor),
                                    gx=_np(input_grad), gs=_np(shift_grad), masked=True, z=_np(input), abmi=_np(abmi),
                                    k12=_np(k12), dgamma=_np(dgamma), dbeta=_np(dbeta))
            return ret

        self.fused_f3 = 0
        monkeypatch.setattr(rc, "rubiks_shift_3d_forward_bn_float", fwd3bn)
        monkeypatch.setattr(rc, "rubiks_shift_3d_backward_bn_float", bwd3bn)

        def fwd2(input, shift, strides, paddings, quantize, output):
            ret = f2(input=input, shift=shift, strides=strides, paddings=paddings, quantize=quantize, output=output)
            taps.calls["f2"] += 1
            key = (tuple(input.shape), tuple(strides), input.dtype)
            if key not in taps.f2:
                taps.f2[key] = dict(x=input.detach().cpu(), shift=shift.detach().cpu(), s=list(strides), p=list(paddings),
                                    q=bool(quantize), y=output.detach().cpu())
            return ret

        def bwd2(upstream_grad, input, shift, strides, paddings, normalize_grad, enable_shift_grad, quantize,
                 input_grad, shift_grad):
            ret = b2(upstream_grad=upstream_grad, input=input, shift=shift, strides=strides, paddings=paddings,
                     normalize_grad=normalize_grad, enable_shift_grad=enable_shift_grad, quantize=quantize,
                     input_grad=input_grad, shift_grad=shift_grad)
            taps.calls["b2"] += 1
            key = (tuple(input.shape), tuple(strides), input.dtype)
            if key not in taps.b2:
                taps.b2[key] = dict(x=input.detach().cpu(), shift=shift.detach().cpu(), gy=upstream_grad.detach().cpu(),
                                    s=list(strides), p=list(paddings), q=bool(quantize), norm=bool(normalize_grad),
                                    enable=bool(enable_shift_grad), gx=input_grad.detach().cpu(),
                                    gs=shift_grad.detach().cpu())
            return ret

        def tfwd(ctx, x, soft, n_segment):
            y = tsf(ctx, x, soft, n_segment)
            taps.calls["fa"] += 1
            key = (tuple(x.shape), x.dtype)
            if key not in taps.fa:
                taps.fa[key] = dict(x=x.detach().cpu(), soft=soft.detach().cpu(), S=int(n_segment), y=y.detach().cpu())
            return y

        def tbwd(ctx, gy):
            out = tsb(ctx, gy)
            taps.calls["ba"] += 1
            x, soft = ctx.saved_tensors
            key = (tuple(x.shape), x.dtype)
            if key not in taps.ba:
                taps.ba[key] = dict(x=x.detach().cpu(), soft=soft.detach().cpu(), S=int(ctx.n_segment), gy=gy.detach().cpu(),
                                    gx=out[0].detach().cpu(), gsoft=out[1].detach().cpu())
            return out

        # -aq blocks with an identity shortcut in training: bn1 + ReLU folded into the temporal filter (fused_bn.py): counted with
        # the plain calls, recorded separately (fa_bn / ba_bn) and checked against fp64 PyTorch below
        from rubiksnet_amd.fused_bn import _BNReLUTShiftTrain as BT
        btf, btb = BT.forward, BT.backward
        self.fa_bn, self.ba_bn = {}, {}

        def btfwd(ctx, x, weight, bias, soft, rm, rv, momentum, eps, n_segment, counter, stats=None):
            out = btf(ctx, x, weight, bias, soft, rm, rv, momentum, eps, n_segment, counter, stats)
            taps.calls["fa"] += 1
            key = (tuple(x.shape), x.dtype)
            if key not in taps.fa_bn:
                taps.fa_bn[key] = dict(x=x.detach().cpu(), w=weight.detach().cpu(), b=bias.detach().cpu(), soft=soft.detach().cpu(),
                                       S=int(n_segment), eps=float(eps), y=out[0].detach().cpu())
            return out

        def btbwd(ctx, gy, dskip=None):
            key = (tuple(gy.shape), ctx.saved_tensors[0].dtype)
            first = key not in taps.ba_bn
            if first:
                sx, sw, sb, ssoft = ctx.saved_tensors[:4]          # (blocks share shapes: the record carries its own inputs)
                rec = dict(gy=gy.detach().cpu(), dskip=None if dskip is None else dskip.detach().cpu(), x=sx.detach().cpu(),
                           w=sw.detach().cpu(), b=sb.detach().cpu(), soft=ssoft.detach().cpu(), S=int(ctx.n_segment))
            out = btb(ctx, gy, dskip)
            taps.calls["ba"] += 1
            if first:
                rec.update(dx=out[0].detach().cpu(), dgamma=out[1].detach().cpu(), dbeta=out[2].detach().cpu(),
                           gsoft=out[3].detach().cpu())
                taps.ba_bn[key] = rec
            return out

        monkeypatch.setattr(BT, "forward", staticmethod(btfwd))
        monkeypatch.setattr(BT, "backward", staticmethod(btbwd))

        # downsampling -aq blocks (bf16): the same with the projecting shortcut's operand gathered and its gradient joined inside the
        # filter's backward (fused_bn._BNReLUTShiftForkTrain): fa_fork / ba_fork
        from rubiksnet_amd.fused_bn import _BNReLUTShiftForkTrain as BF
        bff, bfb = BF.forward, BF.backward
        self.fa_fork, self.ba_fork = {}, {}

        def bffwd(ctx, x, weight, bias, soft, rm, rv, momentum, eps, n_segment, counter, stats=None):
            out = bff(ctx, x, weight, bias, soft, rm, rv, momentum, eps, n_segment, counter, stats)
            taps.calls["fa"] += 1
            key = (tuple(x.shape), x.dtype)
            if key not in taps.fa_fork:
                taps.fa_fork[key] = dict(x=x.detach().cpu(), w=weight.detach().cpu(), b=bias.detach().cpu(), soft=soft.detach().cpu(),
                                         S=int(n_segment), eps=float(eps), y=out[0].detach().cpu(), xs=out[1].detach().cpu())
            return out

        def bfbwd(ctx, gy, gxs):
            key = (tuple(gy.shape), ctx.saved_tensors[0].dtype)
            first = key not in taps.ba_fork
            if first:
                sx, sw, sb, ssoft = ctx.saved_tensors[:4]
                rec = dict(gy=gy.detach().cpu(), gxs=gxs.detach().cpu(), x=sx.detach().cpu(), w=sw.detach().cpu(), b=sb.detach().cpu(),
                           soft=ssoft.detach().cpu(), S=int(ctx.n_segment))
            out = bfb(ctx, gy, gxs)
            taps.calls["ba"] += 1
            if first:
                rec.update(dx=out[0].detach().cpu(), dgamma=out[1].detach().cpu(), dbeta=out[2].detach().cpu(),
                           gsoft=out[3].detach().cpu())
                taps.ba_fork[key] = rec
            return out

        monkeypatch.setattr(BF, "forward", staticmethod(bffwd))
        monkeypatch.setattr(BF, "backward", staticmethod(bfbwd))

        # -aq blocks on 14 x 14 planes in training: bn2 + ReLU folded into the 2-D shift (fused_bn._BNReLUShift2DTrain): counted
        # with the plain 2-D calls, recorded separately (f2_bn / b2_bn) with the activation materialised by the stand-alone
        # normalise kernel -- so the checks read "normalise, then the oracle's shift", as for the 3-D fusion above
        from rubiksnet_amd.fused_bn import _BNReLUShift2DTrain as B2
        b2f, b2b = B2.forward, B2.backward
        self.f2_bn, self.b2_bn = {}, {}

        def _act2(z, ab):
            from rubiksnet_amd import _native
            Fr, C, H, W = z.shape
            out = torch.empty_like(z)
            sfx = "bf16" if z.dtype == torch.bfloat16 else "f32"
            _native.check(getattr(_native.lib(), "rk_bn_apply_affine_" + sfx)(
                z.data_ptr(), ab[0].contiguous().data_ptr(), ab[1].contiguous().data_ptr(), out.data_ptr(), Fr, C, H * W, 1,
                torch.cuda.current_stream().cuda_stream), "apply")
            return out

        def b2fwd(ctx, z, weight, bias, shift, rm, rv, momentum, eps, counter, normalize_grad, stats=None, stride=(1, 1),
                  padding=(0, 0)):
            y = b2f(ctx, z, weight, bias, shift, rm, rv, momentum, eps, counter, normalize_grad, stats, stride, padding)
            taps.calls["f2"] += 1
            key = (tuple(z.shape), tuple(stride) + tuple(padding), z.dtype)
            if key not in taps.f2_bn:
                ab = ctx.to_save[6]
                taps.f2_bn[key] = dict(x=_act2(z, ab).detach().cpu(), shift=shift.detach().cpu(), y=y.detach().cpu())
            return y

        def b2bwd(ctx, gy):
            z, weight, bias, shift, save_mean, save_invstd, ab = ctx.saved_tensors[:7]
            key = (tuple(z.shape), tuple(ctx.geometry), z.dtype)
            first = key not in taps.b2_bn
            rec = None
            if first:
                rec = dict(x=_act2(z, ab).detach().cpu(), z=z.detach().cpu(), shift=shift.detach().cpu(), gy=gy.detach().cpu(),
                           w=weight.detach().cpu(), mean=save_mean.detach().cpu(), invstd=save_invstd.detach().cpu())
            out = b2b(ctx, gy)
            taps.calls["b2"] += 1
            if first:
                rec.update(dz=out[0].detach().cpu(), dgamma=out[1].detach().cpu(), dbeta=out[2].detach().cpu(),
                           gs=out[3].detach().cpu())
                taps.b2_bn[key] = rec
            return out

        monkeypatch.setattr(B2, "forward", staticmethod(b2fwd))
        monkeypatch.setattr(B2, "backward", staticmethod(b2bwd))
        monkeypatch.setattr(rc, "rubiks_shift_3d_forward_float", fwd3)
        monkeypatch.setattr(rc, "rubiks_shift_3d_backward_float", bwd3)
        monkeypatch.setattr(rc, "rubiks2d_forward", fwd2)
        monkeypatch.setattr(rc, "rubiks2d_backward", bwd2)
        monkeypatch.setattr(TS, "forward", staticmethod(tfwd))
        monkeypatch.setattr(TS, "backward", staticmethod(tbwd))


def _check_3d_forward(oracle, rec, what):
    for key, r in rec.items():
        y_ref = oracle.rk3d_forward(r["x"], r["shift"], r["s"], r["p"], r["q"])
        np.testing.assert_array_equal(r["y"], y_ref, err_msg="%s forward %s" % (what, key))


def _check_3d_backward(oracle, rec, what):
    for key, r in rec.items():
        gx_ref, _ = oracle.rk3d_backward(r["gy"], r["x"], r["shift"], r["s"], r["p"], quantize=r["q"])
        if r.get("masked"):
            # fused with bn2's backward: d(x) leaves the kernel masked by the ReLU, and the BatchNorm constants with it
            gx_ref = np.where(r["x"] > 0, gx_ref, np.float32(0))
            mean, inv = r["abmi"][:, 2].astype(np.float64), r["abmi"][:, 3].astype(np.float64)
            zhat = (r["z"].astype(np.float64) - mean[None, None, :, None, None]) * inv[None, None, :, None, None]
            s1 = gx_ref.astype(np.float64).sum(axis=(0, 1, 3, 4))
            s2 = (gx_ref.astype(np.float64) * zhat).sum(axis=(0, 1, 3, 4))
            count = r["z"].size / r["z"].shape[2]
            scale = max(1.0, float(np.abs(s2).max()), float(np.abs(s1).max()))
            np.testing.assert_allclose(r["dbeta"], s1, rtol=0, atol=2e-5 * scale, err_msg="%s d(beta) %s" % (what, key))
            np.testing.assert_allclose(r["dgamma"], s2, rtol=0, atol=2e-5 * scale, err_msg="%s d(gamma) %s" % (what, key))
            np.testing.assert_allclose(r["k12"], np.stack([s1, s2]) / count, rtol=0, atol=2e-5 * scale / count,
                                       err_msg="%s k12 %s" % (what, key))
        np.testing.assert_array_equal(r["gx"], gx_ref, err_msg="%s d(x) %s" % (what, key))
        _, gs_ref = oracle.rk3d_backward(r["gy"].astype(np.float64), r["x"].astype(np.float64),
                                         r["shift"].astype(np.float64), r["s"], r["p"], normalize_grad=r["norm"],
                                         normalize_t_factor=r["tf"], quantize=r["q"])
        assert r["norm"], "the model's shift layers normalise their gradient (K5)"
        # per-channel unit vectors after K5: same absolute bar as tests/test_parity_3d.py
        np.testing.assert_allclose(r["gs"], gs_ref, rtol=0, atol=2e-5, err_msg="%s d(shift) %s" % (what, key))


def _snapshot(net, names):
    return {n: p.detach().clone() for n, p in net.named_parameters() if n in names}


def test_large_train_step_shift_layers_match_oracle(oracle, monkeypatch):
    """configs[3], one replica's share: RubiksNet-Large forward + backward + Adam; every distinct RubiksShift3D
    call shape (9 of them over 51 layers) is compared with the oracle, forward AND backward."""
    from rubiksnet_amd import RubiksNet, dp

    torch.manual_seed(0)
    B = 4
    net = RubiksNet("large", 174, verbose=False).to(DEV)
    opt = dp.make_optimizer(net, lr=1e-3, lr_shift_mult=0.1, kind="adam")
    watched = {"backbone.conv1.weight", "backbone.layer3.17.as3.rubiks3d.shift", "backbone.layer4.2.conv3.weight",
               "new_fc.weight"}
    before = _snapshot(net, watched)
    assert set(before) == watched
    taps = _Taps(monkeypatch)
    clips = torch.randn(B, 8, 3, 224, 224, device=DEV)
    labels = torch.randint(0, 174, (B,), device=DEV)
    loss = dp.train_step(net, opt, clips, labels)
    torch.cuda.synchronize()
    assert torch.isfinite(loss)
    assert taps.calls["f3"] == 51 and taps.calls["b3"] == 51 and taps.calls["f2"] == 0
    assert taps.fused_f3 >= 47, "every layer on 112 / 56 / 28 / 14-wide planes but the 28 -> 14 one takes the BatchNorm-fused shift kernels"
    want = {((B, 8, c, h, h), (1, s, s)) for c, h, s in _expected_shapes(72)}
    assert set(taps.f3) == want and set(taps.b3) == want
    _check_3d_forward(oracle, taps.f3, "large")
    _check_3d_backward(oracle, taps.b3, "large")
    after = _snapshot(net, watched)
    for n in watched:                                   # the optimizer really stepped every kind of parameter
        assert torch.isfinite(after[n]).all() and not torch.equal(after[n], before[n]), n
    shifts = [p for n, p in net.named_parameters() if n.endswith("shift")]
    assert len(shifts) == 51 and all(p.grad is not None and torch.isfinite(p.grad).all() for p in shifts)


def test_tiny_train_step_at_the_bench_batch_matches_oracle(oracle, monkeypatch):
    """The per-GPU batch of the bench (32 clips, what one rank of configs[3] / the metric's Tiny leg runs): RubiksNet-Tiny
    forward + backward + Adam with every distinct RubiksShift3D call -- 9 shapes over 17 layers, at N = 32 the launches take
    the paths the bench times (full workgroup rounds, the slab kernels' finalizers with 64 partials per channel) -- compared
    with the oracle, forward AND backward (d(shift) sums over all 32 clips)."""
    from rubiksnet_amd import RubiksNet, dp

    torch.manual_seed(7)
    B = 32
    net = RubiksNet("tiny", 174, verbose=False).to(DEV)
    opt = dp.make_optimizer(net, lr=1e-3, lr_shift_mult=0.1, kind="adam")
    taps = _Taps(monkeypatch)
    clips = torch.randn(B, 8, 3, 224, 224, device=DEV)
    labels = torch.randint(0, 174, (B,), device=DEV)
    loss = dp.train_step(net, opt, clips, labels)
    torch.cuda.synchronize()
    assert torch.isfinite(loss)
    assert taps.calls["f3"] == 17 and taps.calls["b3"] == 17
    want = {((B, 8, c, h, h), (1, s, s)) for c, h, s in _expected_shapes(54)}
    assert set(taps.f3) == want and set(taps.b3) == want
    _check_3d_forward(oracle, taps.f3, "tiny b32")
    _check_3d_backward(oracle, taps.b3, "tiny b32")


def test_large_train_step_at_the_bench_batch_matches_oracle(oracle, monkeypatch):
    """configs[3] at the per-GPU batch the bench times (32 clips = one rank's share of the global 256): RubiksNet-Large
    forward + backward + Adam, every distinct RubiksShift3D call shape -- 9 over 51 layers, at N = 32 the
    launches take the paths the `large-train` bench leg runs (full rounds of resident workgroups, 14x14 planes of 288
    channels, in-launch finalizers over 32+ partials per channel) -- compared with the oracle, forward AND backward, over the
    whole batch (d(shift) is a sum over all 32 clips, so nothing can be sub-sampled there)."""
    from rubiksnet_amd import RubiksNet, dp

    torch.manual_seed(11)
    B = 32
    net = RubiksNet("large", 174, verbose=False).to(DEV)
    opt = dp.make_optimizer(net, lr=1e-3, lr_shift_mult=0.1, kind="adam")
    taps = _Taps(monkeypatch)
    clips = torch.randn(B, 8, 3, 224, 224, device=DEV)
    labels = torch.randint(0, 174, (B,), device=DEV)
    loss = dp.train_step(net, opt, clips, labels)
    torch.cuda.synchronize()
    assert torch.isfinite(loss)
    del clips
    assert taps.calls["f3"] == 51 and taps.calls["b3"] == 51 and taps.calls["f2"] == 0
    want = {((B, 8, c, h, h), (1, s, s)) for c, h, s in _expected_shapes(72)}
    assert set(taps.f3) == want and set(taps.b3) == want
    _check_3d_forward(oracle, taps.f3, "large b32")
    _check_3d_backward(oracle, taps.b3, "large b32")
    shifts = [p for n, p in net.named_parameters() if n.endswith("shift")]
    assert len(shifts) == 51 and all(p.grad is not None and torch.isfinite(p.grad).all() for p in shifts)


def test_tiny_forward_batch64_shift_layers_match_oracle(oracle, monkeypatch):
    """configs[2]: RubiksNet-Tiny full forward at batch 64 (eval, no grad -> the fused inference blocks);
    the first two clips of every distinct shift call are compared with the oracle (the operator is per clip)."""
    from rubiksnet_amd import RubiksNet

    torch.manual_seed(2)
    net = RubiksNet("tiny", 174, verbose=False).to(DEV).eval()
    taps = _Taps(monkeypatch, keep_clips=2)
    with torch.no_grad():
        out = net(torch.randn(64, 8, 3, 224, 224, device=DEV))
    assert out.shape == (64, 174) and torch.isfinite(out).all()
    assert taps.calls["f3"] == 17
    assert {(k[0][2], k[0][3], k[1][1]) for k in taps.f3} == _expected_shapes(54)
    assert all(k[0][0] == 64 for k in taps.f3)
    _check_3d_forward(oracle, taps.f3, "tiny b64")


def _rounded(a, dtype):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dtype)


@pytest.mark.parametrize("tier,amp", [("large", torch.bfloat16), ("tiny", None)])
def test_aq_train_step_shift_layers_match_oracle(oracle, monkeypatch, tier, amp):
    """configs[4]: the attention-quantized variant (RubiksShift2D + AttentionShift per block), Large under bf16
    autocast (and Tiny in fp32): train step with every distinct 2-D shift / temporal-tap call checked."""
    from oracle import attention_oracle as ao
    from rubiksnet_amd import RubiksNet, dp

    torch.manual_seed(4)
    B = 4
    width = 72 if tier == "large" else 54
    nblocks = 51 if tier == "large" else 17
    net = RubiksNet(tier, 174, variant="rubiks3d-aq", verbose=False).to(DEV)
    opt = dp.make_optimizer(net, lr=1e-3, kind="adam")
    taps = _Taps(monkeypatch)
    clips = torch.randn(B, 8, 3, 224, 224, device=DEV)
    labels = torch.randint(0, 174, (B,), device=DEV)
    with torch.autocast("cuda", dtype=amp, enabled=amp is not None):
        loss = dp.train_step(net, opt, clips, labels)
    torch.cuda.synchronize()
    assert torch.isfinite(loss)
    assert taps.calls == dict(f3=0, b3=0, f2=nblocks, b2=nblocks, fa=nblocks, ba=nblocks)
    st = torch.bfloat16 if amp is not None else torch.float32
    # (every block takes the bn2-folded form -- f2_bn / b2_bn -- except, in fp32, the stride-1 planes the LDS-DMA kernels stream:
    # 112 x 112, 56 x 56, 28 x 28)
    assert ({(k[0][1], k[0][2], k[1][0]) for k in taps.f2} | {(k[0][1], k[0][2], k[1][0]) for k in taps.f2_bn}) == _expected_shapes(width)
    assert all(k[2] == st and k[0][0] == B * 8 for k in taps.f2) and set(taps.b2) == set(taps.f2)
    assert set(taps.b2_bn) == set(taps.f2_bn) and any(k[0][2:] == (14, 14) for k in taps.f2_bn)
    assert any(k[1][0] == 2 for k in taps.f2_bn) and any(k[0][2:] == (7, 7) for k in taps.f2_bn)
    assert all(k[2] == st and k[0][0] == B * 8 for k in taps.f2_bn)
    if amp is not None:
        assert not taps.f2                             # bf16: all 51 blocks fused
    for key, r in taps.f2_bn.items():                  # ---- bn2 + ReLU + RubiksShift2D as one operator: "normalise, then shift", bit for bit
        y_ref = oracle.rk2d_forward(r["x"].float().numpy(), r["shift"].float().numpy(), list(key[1][:2]), list(key[1][2:]), False)
        assert torch.equal(r["y"], _rounded(y_ref, st)), "bn2 + 2-D forward %s" % (key,)
    for key, r in taps.b2_bn.items():
        xf, sf, gf = r["x"].float().numpy(), r["shift"].float().numpy(), r["gy"].float().numpy()
        sp, pp = list(key[1][:2]), list(key[1][2:])
        gx_ref, _ = oracle.rk2d_backward(gf, xf, sf, sp, pp, quantize=False)
        _, gs_ref = oracle.rk2d_backward(gf.astype(np.float64), xf.astype(np.float64), sf.astype(np.float64), sp, pp,
                                         normalize_grad=True)
        np.testing.assert_allclose(r["gs"].float().numpy(), gs_ref, rtol=0, atol=2e-5, err_msg="bn2 + 2-D d(shift) %s" % (key,))
        # bn2's backward from the oracle's d(x): mask, the two sums, d(z) = gamma invstd (dz - k1 - zhat k2)
        dzm = _rounded(gx_ref, st).double() * (r["x"].double() > 0)
        zhat = (r["z"].double() - r["mean"].double().view(1, -1, 1, 1)) * r["invstd"].double().view(1, -1, 1, 1)
        dbeta, dgamma = dzm.sum(dim=(0, 2, 3)), (dzm * zhat).sum(dim=(0, 2, 3))
        cnt = dzm.numel() / dzm.shape[1]
        dz = (r["w"].double() * r["invstd"].double()).view(1, -1, 1, 1) * (dzm - (dbeta / cnt).view(1, -1, 1, 1)
                                                                           - zhat * (dgamma / cnt).view(1, -1, 1, 1))
        tol = 2.0 ** -6 if st == torch.bfloat16 else 1e-5
        for name, got, ref in (("dbeta", r["dbeta"], dbeta), ("dgamma", r["dgamma"], dgamma), ("dz", r["dz"], dz)):
            np.testing.assert_allclose(got.double().numpy(), ref.numpy(), rtol=0,
                                       atol=(5e-3 if name != "dz" and st == torch.bfloat16 else tol) * max(1e-6, float(ref.abs().max())),
                                       err_msg="bn2 + 2-D %s %s" % (name, key))

    for key, r in taps.f2.items():                     # ---- RubiksShift2D forward
        xf, sf = r["x"].float().numpy(), r["shift"].float().numpy()
        y_ref = oracle.rk2d_forward(xf, sf, r["s"], r["p"], r["q"])
        assert torch.equal(r["y"], _rounded(y_ref, st)), "2-D forward %s" % (key,)
    for key, r in taps.b2.items():                     # ---- RubiksShift2D backward
        xf, sf, gf = r["x"].float().numpy(), r["shift"].float().numpy(), r["gy"].float().numpy()
        gx_ref, _ = oracle.rk2d_backward(gf, xf, sf, r["s"], r["p"], quantize=r["q"])
        assert torch.equal(r["gx"], _rounded(gx_ref, st)), "2-D d(x) %s" % (key,)
        assert r["norm"] and r["enable"]
        _, gs_ref = oracle.rk2d_backward(gf.astype(np.float64), xf.astype(np.float64), sf.astype(np.float64), r["s"],
                                         r["p"], normalize_grad=True)
        # unit vectors after K9.  Under autocast the shift table reaches the kernels as the fp32 parameter it is and
        # d(shift) comes back in fp32 (rk2d_*_sf32): the same 2e-5 bar as the fp32 network
        assert r["shift"].dtype == torch.float32 and r["gs"].dtype == torch.float32
        np.testing.assert_allclose(r["gs"].float().numpy(), gs_ref, rtol=0, atol=2e-5,
                                   err_msg="2-D d(shift) %s" % (key,))

    # AttentionShift sits in front of conv2: block inputs (C, H) = (w,112), (w,56), (2w,28), (4w,14), (8w,7); the blocks
    # with an identity shortcut take the bn1-folded form (fa_bn / ba_bn), the 4 projecting ones the plain filter
    assert {(k[0][1], k[0][2]) for k in list(taps.fa) + list(taps.fa_bn) + list(taps.fa_fork)} == {
        (width, 112), (width, 56), (2 * width, 28), (4 * width, 14), (8 * width, 7)}
    assert set(taps.fa) == set(taps.ba) and set(taps.fa_bn) == set(taps.ba_bn) and len(taps.fa_bn) >= 4
    assert set(taps.fa_fork) == set(taps.ba_fork)
    if amp is not None:
        assert not taps.fa and len(taps.fa_fork) == 4          # bf16: the 4 projecting blocks take the forked form
    def bn_taps_ref(r, eps):
        x = r["x"].double().requires_grad_(True)
        w, b, soft = (r[n].double().requires_grad_(True) for n in ("w", "b", "soft"))
        a = F.relu(F.batch_norm(x, None, None, w, b, True, 0.0, eps))
        a.retain_grad()
        NT, C, H, W = a.shape
        a5 = a.view(NT // r["S"], r["S"], C, H, W)
        pad = torch.zeros_like(a5[:, :1])
        s0, s1, s2 = (soft[:, j].view(1, 1, C, 1, 1) for j in range(3))
        y = (s0 * torch.cat([pad, a5[:, :-1]], 1) + s1 * a5 + s2 * torch.cat([a5[:, 1:], pad], 1)).view(NT, C, H, W)
        return x, w, b, soft, a, y

    bar = 2.0 ** -6 if st == torch.bfloat16 else 1e-5
    eps = float(net.backbone.layer0[0].bn1.eps)
    for key, r in taps.fa_bn.items():                  # ---- bn1 + ReLU + temporal taps as one operator, against fp64 PyTorch
        y = bn_taps_ref(r, r["eps"])[5].detach()
        np.testing.assert_allclose(r["y"].double().numpy(), y.numpy(), rtol=0, atol=bar * float(y.abs().max()),
                                   err_msg="bn1 + taps forward %s" % (key,))
    for key, g in taps.ba_bn.items():
        x, w, b, soft, a, y = bn_taps_ref(g, eps)
        y.backward(g["gy"].double())
        dx = x.grad + (g["dskip"].double() if g["dskip"] is not None else 0)
        # d(x) = a (dz - k1 - xhat k2) cancels: its error scales with d(activation), which bf16 storage rounds (in the
        # unfused kernels as well)
        scale = max(float(dx.abs().max()), float(w.abs().max()) * float(a.grad.abs().max()) / float(x.detach().std()))
        # (an element whose pre-activation rounds to the other side of 0 in fp32 than in fp64 takes the other ReLU branch:
        # a handful per million)
        off = (g["dx"].double() - dx).abs() > bar * scale
        assert float(off.double().mean()) < 1e-5, "bn1 + taps d(x) %s: %d elements off" % (key, int(off.sum()))
        for name, ref in (("dgamma", w.grad), ("dbeta", b.grad), ("gsoft", soft.grad)):
            np.testing.assert_allclose(g[name].double().numpy(), ref.numpy(), rtol=0,
                                       atol=(5e-3 if st == torch.bfloat16 else 1e-4) * float(ref.abs().max()),
                                       err_msg="bn1 + taps %s %s" % (name, key))
    for key, r in taps.fa_fork.items():                # ---- the forked form: + the gathered activation ...
        _, _, _, _, a, y = bn_taps_ref(r, r["eps"])
        np.testing.assert_allclose(r["y"].double().numpy(), y.detach().numpy(), rtol=0, atol=bar * float(y.abs().max()),
                                   err_msg="forked bn1 + taps forward %s" % (key,))
        xs = a.detach()[:, :, ::2, ::2]
        np.testing.assert_allclose(r["xs"].double().numpy(), xs.numpy(), rtol=0, atol=bar * float(xs.abs().max()),
                                   err_msg="forked bn1: gathered activation %s" % (key,))
    for key, g in taps.ba_fork.items():                # ---- ... and the shortcut's gradient joined before the ReLU mask
        x, w, b, soft, a, y = bn_taps_ref(g, eps)
        torch.autograd.backward([y, a[:, :, ::2, ::2]], [g["gy"].double(), g["gxs"].double()])
        dx = x.grad
        scale = max(float(dx.abs().max()), float(w.abs().max()) * float(a.grad.abs().max()) / float(x.detach().std()))
        off = (g["dx"].double() - dx).abs() > bar * scale
        assert float(off.double().mean()) < 1e-5, "forked bn1 + taps d(x) %s: %d elements off" % (key, int(off.sum()))
        for name, ref in (("dgamma", w.grad), ("dbeta", b.grad), ("gsoft", soft.grad)):
            np.testing.assert_allclose(g[name].double().numpy(), ref.numpy(), rtol=0,
                                       atol=(5e-3 if st == torch.bfloat16 else 1e-4) * float(ref.abs().max()),
                                       err_msg="forked bn1 + taps %s %s" % (name, key))
    for key, r in taps.fa.items():                     # ---- AttentionShift taps, forward
        assert r["x"].dtype == st and r["soft"].dtype == torch.float32
        y_ref = ao.taps_forward(r["x"].float().numpy(), r["soft"].numpy(), r["S"])
        assert torch.equal(r["y"], _rounded(y_ref, st)), "temporal taps forward %s" % (key,)
    for key, r in taps.ba.items():                     # ---- AttentionShift taps, backward
        xf, gf, soft = r["x"].float().numpy(), r["gy"].float().numpy(), r["soft"].numpy()
        gx_ref, _ = ao.taps_backward(gf, xf, soft, r["S"], compute=np.float32)
        assert torch.equal(r["gx"], _rounded(gx_ref, st)), "temporal taps d(x) %s" % (key,)
        _, gs_ref = ao.taps_backward(gf, xf, soft, r["S"])
        scale = float(np.abs(gs_ref).max())
        np.testing.assert_allclose(r["gsoft"].double().numpy(), gs_ref, rtol=0, atol=1e-3 * scale,
                                   err_msg="temporal taps d(taps) %s" % (key,))
    att = [p for n, p in net.named_parameters() if n.endswith("conv2.0.weight")]
    assert len(att) == nblocks and all(p.grad is not None and torch.isfinite(p.grad).all() for p in att)
