"""CPU tests of the host side: reference-compatible surface, structure fixtures, argument handling.
No GPU and no compute calls into the HIP library."""
import inspect
import json
import os

import numpy as np
import pytest
import torch

import rubiksnet_amd
from rubiksnet_amd import AttentionShift, RubiksNet, RubiksShift2D, RubiksShift3D, RubiksShiftBase, utils
from rubiksnet_amd.shiftlib.rubiks2d import primitive as p2
from rubiksnet_amd.shiftlib.rubiks3d import layer as l3
from rubiksnet_amd.shiftlib.rubiks3d import primitive as p3


@pytest.fixture(scope="module")
def structure(golden_dir):
    with open(os.path.join(golden_dir, "model_structure.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("tier", ["tiny", "small", "medium", "large"])
def test_state_dict_matches_reference_structure(structure, tier):
    """Keys, order and shapes of state_dict() equal what the reference's RubiksNet builds
    (tests/golden/model_structure.json, captured by importing the reference Python)."""
    net = RubiksNet(tier, num_classes=174, num_frames=8, verbose=False)
    got = [[k, list(v.shape)] for k, v in net.state_dict().items()]
    assert got == structure[tier]["state_dict"]
    assert sum(p.numel() for p in net.parameters()) == structure[tier]["num_params"]
    assert net.feature_dim == structure[tier]["feature_dim"]
    shifts = [[m.num_channels, list(m.stride), list(m.padding)] for m in net.modules() if isinstance(m, RubiksShift3D)]
    assert shifts == structure[tier]["shift3d_layers"]


def test_param_counts_match_readme():
    """README.md:87-91 of the reference: 1.9M / 3.6M / 6.2M / 8.5M parameters."""
    want = {"tiny": 1.9, "small": 3.6, "medium": 6.2, "large": 8.5}
    for tier, m in want.items():
        n = sum(p.numel() for p in RubiksNet(tier, 174, verbose=False).parameters())
        assert round(n / 1e6, 1) == m


def test_aq_variant_structure():
    """-aq keeps the 2D shift and prepends AttentionShift to conv2 (models.py:71-79); here it builds on CPU."""
    net = RubiksNet("tiny", 10, variant="rubiks3d-aq", verbose=False)
    sd = net.state_dict()
    assert sd["backbone.layer1.0.conv2.0.weight"].shape == (54, 3)
    assert sd["backbone.layer1.0.conv2.0.T"].item() == 2.0
    assert sd["backbone.layer1.0.as3.shift"].shape == (2, 54)
    assert not any("rubiks3d" in k for k in sd)
    assert all(isinstance(m.conv2[0], AttentionShift) for m in net.backbone.layer2)


def test_checkpoint_roundtrip(tmp_path):
    """load_pretrained reads the reference's checkpoint dict (models.py:52-62)."""
    net = RubiksNet("tiny", 7, num_frames=4, verbose=False)
    path = tmp_path / "ck.pth.tar"
    torch.save({"tier": "tiny", "num_classes": 7, "num_frames": 4, "variant": "rubiks3d", "model": net.state_dict()}, path)
    net2 = RubiksNet.load_pretrained(str(path))
    for (k, a), (_, b) in zip(net.state_dict().items(), net2.state_dict().items()):
        assert torch.equal(a, b), k
    net2.replace_new_fc(3)
    assert net2.new_fc.out_features == 3


def test_public_signatures_match_reference():
    sig = inspect.signature
    assert list(sig(p3.rubiks_shift_3d).parameters) == [
        "x", "shift", "stride", "padding", "normalize_grad", "normalize_t_factor", "quantize"]
    assert sig(p3.rubiks_shift_3d).parameters["normalize_t_factor"].default == 1.0
    assert list(sig(p3.rubiks_shift_3d_forward).parameters)[:6] == ["x", "shift", "stride", "padding", "quantize", "output"]
    assert list(sig(p3.rubiks_shift_3d_backward).parameters)[:10] == [
        "upstream_grad", "x", "shift", "stride", "padding", "normalize_grad", "normalize_t_factor", "quantize",
        "x_grad_output", "shift_grad_output"]
    assert list(sig(p2.rubiks2d).parameters) == [
        "x", "shift", "stride", "padding", "normalize_grad", "enable_shift_grad", "quantize"]
    assert list(sig(RubiksShift3D.__init__).parameters)[1:] == [
        "num_channels", "stride", "padding", "normalize_grad", "normalize_t_factor", "quantize", "shift_groups"]
    assert list(sig(RubiksShift2D.__init__).parameters)[1:] == [
        "num_channels", "stride", "padding", "normalize_grad", "quantize", "init_shift"]
    assert p3.RubiksShift3DFunc.__name__ == "RubiksShift3DFunc" and p2.VFS2DFunc.__name__ == "VFS2DFunc"
    assert set(rubiksnet_amd.shiftlib.__all__) == {"RubiksShift2D", "RubiksShift3D", "RubiksShiftBase"}
    from rubiksnet_amd import rubiksnet_cuda
    assert set(rubiksnet_cuda.__all__) == {      # the six names of cuda_src/rubiks.cpp:384-396
        "rubiks2d_forward", "rubiks2d_backward", "rubiks_shift_3d_forward_float", "rubiks_shift_3d_forward_double",
        "rubiks_shift_3d_backward_float", "rubiks_shift_3d_backward_double"}


def test_modules_and_inits():
    m = RubiksShift3D(12, stride=(1, 2, 2))
    assert isinstance(m, RubiksShiftBase) and m.shift.shape == (3, 12)
    assert float(m.shift.abs().max()) <= 1.0 and list(m.state_dict()) == ["shift"]
    assert "shift_channels=12" in repr(m)
    with pytest.raises(AssertionError):
        RubiksShift3D(10, shift_groups=3)
    g = RubiksShift2D(18, init_shift="group3")
    assert sorted(set(g.shift.detach().flatten().tolist())) == [-1.0, 0.0, 1.0]
    with pytest.raises(NotImplementedError):
        RubiksShift2D(4, init_shift="nope")
    s = torch.zeros(1, 16)
    l3.init_shift1d_nfold(s, nfold=8)
    assert s[0, :2].tolist() == [1, 1] and s[0, 2:4].tolist() == [-1, -1] and float(s[0, 4:].abs().max()) <= 1e-3
    m2 = RubiksShift2D(16, stride=2)
    for mode in ("tsm", "tsm-g0.1", "uni0.5", "none"):
        m3 = l3.create_3d_from_2d(m2, init_mode=mode)
        assert m3.stride == (1, 2, 2) and m3.padding == (0, 0, 0)
        if mode == "tsm":
            assert m3.shift[0].tolist() == [1.0] * 2 + [-1.0] * 2 + [0.0] * 12
            assert torch.equal(m3.shift[1:], m2.shift)
        if mode == "none":
            assert torch.isnan(m3.shift).all()


def test_output_shape_formula_is_not_the_conv_formula():
    """out = (in + 2p - 1) // s + 1 (cuda_src/rubiks.cpp:166)."""
    x = torch.zeros(2, 8, 3, 56, 57)
    assert p3.compute_output_shape(x, (1, 2, 2), (0, 0, 0), 3) == (2, 8, 3, 28, 29)
    assert p3.compute_output_shape(x, (2, 1, 3), (1, 2, 0), 3) == (2, 5, 3, 60, 19)
    assert p2.compute_output_shape(torch.zeros(2, 3, 7, 7), 2, 1) == (2, 3, 5, 5)


def test_utils():
    assert utils.make_tuple(3, 2) == [3, 3] and utils.make_tuple((1, 2.0), 2) == [1, 2]
    with pytest.raises(AssertionError):
        utils.make_tuple((1, 2, 3), 2)
    t = torch.ones(2, 3)
    assert utils.allocate_output(None, t, (4, 5)).abs().sum() == 0
    assert utils.allocate_output(None, t, (4, 5), zero=False).shape == (4, 5)
    buf = torch.full((4, 5), 7.0)
    assert utils.allocate_output(buf, t, (4, 5)) is buf
    with pytest.raises(AssertionError):
        utils.allocate_output(buf, t, (5, 4))
    with pytest.raises(AssertionError):
        utils.allocate_output(buf.double(), t, (4, 5))


def test_no_cpu_fallback():
    """CPU tensors are refused, exactly like the reference's `assert x.is_cuda` (primitive.py:61)."""
    x = torch.zeros(1, 2, 3, 4, 4)
    with pytest.raises(AssertionError, match="CUDA"):
        p3.rubiks_shift_3d(x, torch.zeros(3, 3))
    with pytest.raises(AssertionError, match="CUDA"):
        p2.rubiks2d(x[0], torch.zeros(2, 3))
    with pytest.raises(AssertionError):
        AttentionShift(2, num_channels=3)(x[0])
    with pytest.raises(ValueError):
        p3._pick(x.half(), 1, 2)


def test_attention_soft_taps_match_oracle():
    from oracle import attention_oracle as ao

    mod = AttentionShift(8, num_channels=5)
    np.testing.assert_allclose(mod.soft_taps().detach().numpy(), ao.soft_weights(mod.weight.detach().numpy()), rtol=1e-5)
    lazy = AttentionShift(8)
    assert lazy.weight is None and list(lazy.state_dict()) == ["T"]


def test_product_code_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under rubiksnet_amd/ may reference it."""
    root = os.path.dirname(rubiksnet_amd.__file__)
    for dp, _, fs in os.walk(root):
        for f in fs:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".sh")):
                text = open(os.path.join(dp, f), errors="ignore").read()
                assert "import oracle" not in text and "from oracle" not in text and "librubiks_oracle" not in text, f


def test_bn_relu_on_cpu_is_the_stock_pair():
    """fused_bn.bn_relu leaves non-GPU tensors to nn.BatchNorm2d + relu (the gloo tests train on the CPU)."""
    import copy

    import torch
    import torch.nn as nn
    import torch.nn.functional as F

    from rubiksnet_amd.fused_bn import bn_relu

    torch.manual_seed(0)
    bn = nn.BatchNorm2d(5)
    ref = copy.deepcopy(bn)
    x = torch.randn(4, 5, 6, 6)
    assert torch.equal(bn_relu(bn, x), F.relu(ref(x)))
    assert torch.equal(bn.running_var, ref.running_var) and int(bn.num_batches_tracked) == 1
    bn.eval(); ref.eval()
    assert torch.equal(bn_relu(bn, x, relu=False), ref(x))


def test_switches_are_read_once_into_a_frozen_object(monkeypatch):
    """rubiksnet_amd.config: six switches, read at import / on reload(), nothing else consulted per call."""
    import dataclasses

    from rubiksnet_amd import config, fused_bn, pointwise

    sw = config.reload({})
    assert sw == config.Switches(True, "auto", True) and config.switches() is sw
    with pytest.raises(dataclasses.FrozenInstanceError):
        sw.fused_bn = False
    monkeypatch.setenv("RK_PW", "0")
    assert pointwise.pointwise_mode() == "auto"            # not re-read per call
    assert config.reload().pointwise == "0" and pointwise.pointwise_mode() == "0"
    assert config.reload({"RK_FUSED_BN": "0", "RK_FUSED_EVAL": "0", "RK_PW": "all"}) == config.Switches(
        False, "all", False)
    assert fused_bn.fused_bn_enabled() is False
    with pytest.raises(ValueError):
        config.reload({"RK_PW": "sometimes"})
    config.reload({})
    import inspect
    for mod in (fused_bn, pointwise):
        assert "os.environ" not in inspect.getsource(mod)


def test_optim_policy_groups_follow_the_reference_rules():
    """backbone.py:202-235: first matching family wins, only parameter-owning LEAF modules of an unknown type raise."""
    import torch.nn as nn

    net = RubiksNet("tiny", num_classes=5, num_frames=8, verbose=False).backbone
    groups = {g["name"]: g for g in net.get_optim_policy(shift_lr_mult=0.02)}
    assert list(groups) == ["weight", "bias", "bn", "shift"]
    assert [groups[k]["decay_mult"] for k in groups] == [1, 0, 0, 0]
    assert groups["shift"]["lr_mult"] == 0.02 and groups["weight"]["lr_mult"] == 1
    ids = [id(p) for g in groups.values() for p in g["params"]]
    assert len(ids) == len(set(ids)) == len(list(net.parameters()))          # every parameter exactly once
    assert all(p.dim() == 2 and p.shape[0] == 3 for p in groups["shift"]["params"])

    class Container(nn.Module):                                              # unknown type, owns a parameter, NOT a leaf
        def __init__(self):
            super().__init__()
            self.scale = nn.Parameter(torch.ones(1))
            self.inner = nn.Linear(2, 2)

    net.extra = Container()
    names = {g["name"]: len(g["params"]) for g in net.get_optim_policy()}
    assert names["weight"] == len(groups["weight"]["params"]) + 1            # walked through; `scale` is ignored

    class Leaf(nn.Module):
        def __init__(self):
            super().__init__()
            self.scale = nn.Parameter(torch.ones(1))

    net.extra = Leaf()
    with pytest.raises(ValueError, match="New atomic module type"):
        net.get_optim_policy()


def test_model_roofline_bound_is_consistent():
    """rubiksnet_amd/roofline.py: the bound the bench's model legs are reported against (pure host arithmetic)."""
    from rubiksnet_amd.roofline import HBM_PEAK, MFMA_PEAK, model_bound

    tiny = RubiksNet("tiny", 174, verbose=False)
    tr = model_bound(tiny, 32, train=True)
    fw = model_bound(tiny, 32, train=False)
    assert set(tr) >= {"algorithmic_bytes", "mfma_flops", "bound_ms", "hbm_only_ms", "mfma_only_ms", "formula"}
    assert tr["bound_ms"] >= max(tr["hbm_only_ms"], tr["mfma_only_ms"]) - 1e-9          # sum of per-operator maxima
    assert tr["bound_ms"] <= tr["hbm_only_ms"] + tr["mfma_only_ms"] + 1e-9
    assert abs(tr["hbm_only_ms"] - 1e3 * tr["algorithmic_bytes"] / HBM_PEAK) < 1e-9
    assert abs(tr["mfma_only_ms"] - 1e3 * tr["mfma_flops"] / MFMA_PEAK["f32"]) < 1e-9
    assert 2.9 < tr["mfma_flops"] / fw["mfma_flops"] < 3.0        # forward + d(input) + d(weight); the stem has no d(input)
    assert tr["algorithmic_bytes"] > 2 * fw["algorithmic_bytes"]
    assert abs(model_bound(tiny, 64, train=True)["bound_ms"] - 2 * tr["bound_ms"]) < 1e-6       # linear in the batch
    # shift traffic alone, per clip, forward: SURVEY 8(d) quotes 148.3 MB for Tiny
    from rubiksnet_amd.roofline import _shift
    total, h = 0, 112
    for stage in (tiny.backbone.layer0, tiny.backbone.layer1, tiny.backbone.layer2, tiny.backbone.layer3, tiny.backbone.layer4):
        for blk in stage:
            stride = int(blk.shortcut.stride[0]) if hasattr(blk.shortcut, "weight") else 1
            ho = (h - 1) // stride + 1
            total += _shift(blk.conv2.out_channels, h * h, ho * ho, 8, 4, False)[0]
            h = ho
    assert abs(total / 1e6 - 148.3) < 0.1
    aq = model_bound(RubiksNet("large", 174, variant="rubiks3d-aq", verbose=False), 32, True, "bf16", 2)
    assert aq["bound_ms"] == pytest.approx(aq["hbm_only_ms"])     # bf16 MFMA peak: every layer HBM-bound


def test_fused_train_block_declines_what_it_cannot_run():
    """train_block.fused_train_block returns None -- the layer-by-layer path follows -- for CPU tensors, eval mode and the
    -aq variant; no oracle, no CPU arithmetic of its own."""
    from rubiksnet_amd import train_block

    net = RubiksNet("tiny", 5, verbose=False).train()
    blk = net.backbone.layer1[1]
    x = torch.randn(8, 54, 8, 8, requires_grad=True)
    assert train_block.fused_train_block(blk, x) is None                      # CPU tensor
    assert train_block.bn_relu_from_stats(net.backbone.bn_last, torch.randn(8, 432, 7, 7)) is None
    aq = RubiksNet("tiny", 5, variant="rubiks3d-aq", verbose=False).train()
    assert train_block._shift_config(aq.backbone.layer1[1].as3) is None       # 2-D shift + AttentionShift: not this path
    assert train_block._shift_config(blk.as3) is not None
    assert train_block.take_stats(x, 54, x.numel() // 54) is None             # no producer attached statistics


def test_bn_tshift_fusion_declines_off_device():
    """bn_relu_tshift_skip (the -aq block's bn1 + ReLU inside its AttentionShift) applies to CUDA tensors in training only:
    on the host, in eval mode or without gradients it returns None and the block takes bn_relu_skip + the module."""
    import torch
    import torch.nn as nn
    from rubiksnet_amd.attention_shift import AttentionShift
    from rubiksnet_amd.fused_bn import bn_relu_tshift_skip

    bn, shift = nn.BatchNorm2d(6), AttentionShift(4, 6)
    x = torch.randn(8, 6, 3, 3, requires_grad=True)
    assert bn_relu_tshift_skip(bn, shift, x) is None                   # CPU tensor
    bn.eval()
    assert bn_relu_tshift_skip(bn, shift, x) is None                   # eval mode
    with torch.no_grad():
        assert bn_relu_tshift_skip(bn.train(), shift, x) is None       # no gradient wanted
    assert bn_relu_tshift_skip(bn, AttentionShift(4), x) is None       # taps not created yet
