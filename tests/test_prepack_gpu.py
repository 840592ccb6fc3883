"""pointwise.prepacked (rk_pw_pack_many_bf16): every 1x1 weight of a network packed in one launch per train step.  The images
are those of the per-layer rk_pw_pack_bf16, bit for bit; a bf16 train step inside the block is the step outside it; nothing
survives the block (an edit of a weight between steps is seen by the next one)."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _net():
    from rubiksnet_amd import RubiksNet

    torch.manual_seed(11)
    return RubiksNet("tiny", 17, variant="rubiks3d-aq", verbose=False).to(DEV)


def test_images_equal_the_per_layer_pack():
    from rubiksnet_amd import pointwise

    net = _net()
    convs = [m for m in net.modules() if isinstance(m, torch.nn.Conv2d) and m.kernel_size == (1, 1)]
    assert len(convs) > 30
    ref = {c.weight.data_ptr(): pointwise._pack(c.weight) for c in convs}          # outside a block: one launch each
    with pointwise.prepacked(net):
        assert pointwise._table("prepacked") is not None and len(pointwise._table("prepacked")) == len(convs)
        for c in convs:
            f, b = pointwise._pack(c.weight)
            rf, rb = ref[c.weight.data_ptr()]
            assert f.data_ptr() != rf.data_ptr() and torch.equal(f, rf) and torch.equal(b, rb), c
    assert pointwise._table("prepacked") is None
    # the next block sees an edit made through .data
    with torch.no_grad():
        convs[3].weight.data.mul_(2.0)
    with pointwise.prepacked(net):
        f, b = pointwise._pack(convs[3].weight)
    rf, rb = pointwise._pack(convs[3].weight)
    assert torch.equal(f, rf) and torch.equal(b, rb) and not torch.equal(rf, ref[convs[3].weight.data_ptr()][0])


def test_bf16_train_step_is_unchanged(monkeypatch):
    from rubiksnet_amd import config, dp, pointwise

    clips = torch.randn(2, 8, 3, 224, 224, device=DEV)
    labels = torch.randint(0, 17, (2,), device=DEV)
    out = {}
    for on in ("1", "0"):
        monkeypatch.setenv("RK_PREPACK", on)
        config.reload()
        net = _net()
        opt = dp.make_optimizer(net, lr=1e-2, kind="sgd", momentum=0.0)     # (updates proportional to the gradients)
        calls = {"many": 0}
        real = pointwise.prepacked

        def spy(module, _real=real, _calls=calls):
            _calls["many"] += 1
            return _real(module)

        monkeypatch.setattr(pointwise, "prepacked", spy)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            losses = [float(dp.train_step(net, opt, clips, labels).detach())]
        monkeypatch.setattr(pointwise, "prepacked", real)
        torch.cuda.synchronize()
        assert calls["many"] == 1
        out[on] = (losses, copy.deepcopy(net.state_dict()))
    monkeypatch.delenv("RK_PREPACK")
    config.reload()
    assert out["1"][0] == out["0"][0]
    exact = 0
    for k, v in out["1"][1].items():
        w = out["0"][1][k]
        # (the bf16 stem and Tiny's 7x7 layers -- 432 channels: not a multiple of 32 -- are MIOpen's, whose d(weight) uses
        # atomics: not reproducible run to run; everything on the HIP kernels is, bit for bit)
        assert torch.allclose(v.float(), w.float(), rtol=0, atol=1e-4 * max(1.0, float(w.float().abs().max()))), k
        exact += int(torch.equal(v, w))
        if not ("layer4" in k or k.startswith("backbone.conv1") or k.startswith("backbone.bn1")):
            assert torch.equal(v, w), k
    assert exact >= 0.8 * len(out["1"][1])
