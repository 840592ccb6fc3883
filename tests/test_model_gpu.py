"""GPU tests of the callers of the hot path: RubiksNet end to end on the HIP kernels."""
import numpy as np
import pytest
import torch


def _reload_switches():
    from rubiksnet_amd import config
    config.reload()

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_installation_smoke_large():
    """The reference's scripts/test_installation.py: Large, 42 classes, randn(2,8,3,224,224) -> [2,42]."""
    from rubiksnet_amd import RubiksNet

    torch.manual_seed(0)
    net = RubiksNet(tier="large", num_classes=42, num_frames=8, verbose=False).to(DEV)
    video = torch.randn((2, 8, 3, 224, 224), device=DEV)
    out = net(video)
    assert out.shape == (2, 42) and torch.isfinite(out).all()


def test_tiny_overfits_random_batch():
    """README.md:104-106 of the reference: the loss should fall on random data."""
    from rubiksnet_amd import RubiksNet, dp

    torch.manual_seed(0)
    net = RubiksNet("tiny", 5, verbose=False).to(DEV)
    opt = dp.make_optimizer(net, lr=0.01, lr_shift_mult=0.1, kind="sgd")
    clips = torch.randn(4, 8, 3, 224, 224, device=DEV)
    labels = torch.tensor([0, 1, 2, 3], device=DEV)
    losses = [float(dp.train_step(net, opt, clips, labels)) for _ in range(12)]
    assert losses[-1] < 0.6 * losses[0], losses
    shifts = [p for n, p in net.named_parameters() if n.endswith("shift")]
    assert len(shifts) == 17 and all(p.grad is not None and torch.isfinite(p.grad).all() for p in shifts)
    # every replica-local shift gradient is L2-normalised per channel (K5)
    g = shifts[3].grad
    norms = g.norm(dim=0)
    assert torch.allclose(norms[norms > 0], torch.ones_like(norms[norms > 0]), atol=1e-4)


def test_model_shift_layers_match_oracle(oracle):
    """Every RubiksShift3D inside RubiksNet-Tiny (all 9 distinct shapes incl. stride (1,2,2), 112x112 bands,
    14x14 and 7x7 planes) reproduces the oracle on the activations it actually sees."""
    from rubiksnet_amd import RubiksNet
    from rubiksnet_amd.shiftlib import RubiksShift3D

    torch.manual_seed(1)
    net = RubiksNet("tiny", 7, verbose=False).to(DEV).eval()
    seen = {}

    def hook(mod, inp, out):
        key = (tuple(inp[0].shape), tuple(mod.stride))
        if key not in seen:
            seen[key] = (inp[0].detach().cpu().numpy(), mod.shift.detach().cpu().numpy(), mod.stride, mod.padding,
                         out.detach().cpu().numpy())

    hs = [m.register_forward_hook(hook) for m in net.modules() if isinstance(m, RubiksShift3D)]
    with torch.no_grad():
        net(torch.randn(1, 8, 3, 224, 224, device=DEV))
    for h in hs:
        h.remove()
    assert len(seen) == 9
    for (shape, stride), (x, shift, s, p, y) in seen.items():
        np.testing.assert_array_equal(y, oracle.rk3d_forward(x, shift, s, p), err_msg=str((shape, stride)))


@pytest.mark.parametrize("amp", [None, torch.bfloat16])
def test_aq_variant_trains(amp):
    """BASELINE configs[4]: the attention-quantized variant (2D shift + AttentionShift), fp32 and bf16 autocast."""
    from rubiksnet_amd import RubiksNet, dp

    torch.manual_seed(0)
    net = RubiksNet("tiny", 6, variant="rubiks3d-aq", verbose=False).to(DEV)
    opt = dp.make_optimizer(net, lr=1e-3)
    clips = torch.randn(2, 8, 3, 224, 224, device=DEV)
    labels = torch.tensor([1, 4], device=DEV)
    with torch.autocast("cuda", dtype=amp, enabled=amp is not None):
        loss = dp.train_step(net, opt, clips, labels)
    assert torch.isfinite(loss)
    att = [p for n, p in net.named_parameters() if n.endswith("conv2.0.weight")]
    assert len(att) == 17 and all(p.grad is not None and torch.isfinite(p.grad).all() for p in att)


def test_rubiks3d_variant_under_autocast():
    from rubiksnet_amd import RubiksNet

    torch.manual_seed(0)
    net = RubiksNet("tiny", 6, verbose=False).to(DEV)
    clips = torch.randn(1, 8, 3, 224, 224, device=DEV)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = net(clips)
    out.float().sum().backward()
    assert torch.isfinite(out.float()).all()


def test_fused_paths_match_the_stock_backbone(monkeypatch):
    """RubiksNet-Tiny with the fused BN+ReLU and the HIP 1x1 convolutions against the same weights on the stock
    nn.BatchNorm2d / ReLU / MIOpen convolution path: logits, loss, running statistics and a spread of
    gradients agree to fp32 round-off (train mode), and so do the eval-mode logits."""
    import copy

    from rubiksnet_amd import RubiksNet

    torch.manual_seed(3)
    ref = RubiksNet("tiny", 9, verbose=False).to(DEV).train()
    clips = torch.randn(2, 8, 3, 224, 224, device=DEV)
    labels = torch.tensor([1, 7], device=DEV)
    results = []
    for fast in (True, False):
        monkeypatch.setenv("RK_FUSED_BN", "1" if fast else "0")
        _reload_switches()
        monkeypatch.setenv("RK_PW", "auto" if fast else "0")
        _reload_switches()
        net = copy.deepcopy(ref)
        logits = net(clips)
        loss = torch.nn.functional.cross_entropy(logits, labels)
        loss.backward()
        grads = {n: p.grad.detach().clone() for n, p in net.named_parameters()
                 if n.endswith(("conv1.weight", "layer1.0.conv2.weight", "layer1.1.bn2.weight", "layer3.0.conv3.weight",
                                "layer0.0.bn1.bias", "fc.weight"))}
        rv = net.backbone.layer2[0].bn2.running_var.clone()
        net.eval()
        with torch.no_grad():
            ev = net(clips)
        results.append((logits.detach(), loss.detach(), grads, rv, ev))
    (la, lossa, ga, rva, eva), (lb, lossb, gb, rvb, evb) = results
    np.testing.assert_allclose(la.cpu().numpy(), lb.cpu().numpy(), rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(float(lossa), float(lossb), rtol=1e-4)
    np.testing.assert_allclose(rva.cpu().numpy(), rvb.cpu().numpy(), rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(eva.cpu().numpy(), evb.cpu().numpy(), rtol=1e-3, atol=1e-4)
    assert len(ga) == 6
    for n in ga:
        # gradients of the first layers sit behind ~35 layers of differently-ordered fp32 sums and ReLU / BN
        # decisions (the two paths differ by ~1e-7 per layer going forward): compare in norm, not element by element
        err = float((ga[n] - gb[n]).norm() / gb[n].norm())
        assert err < 1e-2, (n, err)


@pytest.mark.parametrize("variant", ["rubiks3d", "rubiks3d-aq"])
def test_fused_inference_blocks_match_layer_by_layer(monkeypatch, variant):
    """Eval mode: blocks whose BatchNorms / residual add ride on the 1x1 GEMMs (pointwise.fused_eval_block) against
    the layer-by-layer path, on random running statistics so that the BN coefficients are not trivial."""
    from rubiksnet_amd import RubiksNet
    from rubiksnet_amd import pointwise

    torch.manual_seed(5)
    net = RubiksNet("tiny", 9, variant=variant, verbose=False).to(DEV)
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            with torch.no_grad():
                m.running_mean.normal_(0, 0.3); m.running_var.uniform_(0.5, 2.0)
                m.weight.uniform_(0.5, 1.5); m.bias.normal_(0, 0.2)
    net.eval()
    clips = torch.randn(2, 8, 3, 224, 224, device=DEV)
    calls = []
    real = pointwise._gemm_fused
    monkeypatch.setattr(pointwise, "_gemm_fused", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    outs = []
    with torch.no_grad():
        for fused in ("1", "0"):
            monkeypatch.setenv("RK_FUSED_EVAL", fused)
            _reload_switches()
            outs.append(net(clips))
    if variant == "rubiks3d":
        assert len(calls) >= 6                               # the fused path really ran (>= 3 blocks x 2 GEMMs)
    else:
        assert len(calls) == 0                               # -aq blocks wrap conv2 with the attention shift: not fused
    scale = float(outs[1].abs().max())
    np.testing.assert_allclose(outs[0].cpu().numpy(), outs[1].cpu().numpy(), rtol=0, atol=2e-4 * scale)
