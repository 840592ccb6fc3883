"""Multi-GPU readiness on ONE GPU (the reference's only multi-GPU call site is scripts/test_models.py:153):
RubiksNet-Tiny wrapped by dp.wrap_ddp under a world-size-1 `nccl` (= RCCL) process group.  This is where the fused training
blocks' side-stream d(weight) kernels, DDP's autograd hooks and `gradient_as_bucket_view` meet on a device: one
dp.train_step must leave the same gradients and the same post-step weights as the un-wrapped model."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(model, opt, clips, labels, steps):
    from rubiksnet_amd import dp
    losses = []
    for _ in range(steps):
        losses.append(float(dp.train_step(model, opt, clips, labels)))
    torch.cuda.synchronize()
    return losses


@pytest.mark.parametrize("tier,variant", [("tiny", "rubiks3d"), ("tiny", "rubiks3d-aq")])
def test_ddp_world_size_one_matches_plain_model(tier, variant):
    import torch.distributed as dist

    from rubiksnet_amd import RubiksNet, dp

    env = dp.init_distributed(prefer_gpu=True)
    assert env.device.type == "cuda" and env.backend == "nccl"
    created = dp.ensure_process_group(env)
    try:
        assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
        torch.manual_seed(0)
        # (-aq: the AttentionShift tap weights ride in DDP's buckets; their batched softmax nodes -- attention_shift.presoftened --
        #  are made per group of layers where the group's first layer runs, so that a bucket is not held back to the end of backward)
        net = RubiksNet(tier, num_classes=17, num_frames=8, variant=variant, verbose=False).to(env.device)
        ref = copy.deepcopy(net)
        model = dp.wrap_ddp(net, env, force=True)
        assert isinstance(model, torch.nn.parallel.DistributedDataParallel)
        clips = torch.randn(4, 8, 3, 224, 224, device=env.device)
        labels = torch.randint(0, 17, (4,), device=env.device)
        opt_d = dp.make_optimizer(model, lr=1e-3, kind="sgd")
        opt_r = dp.make_optimizer(ref, lr=1e-3, kind="sgd")
        model.train(); ref.train()
        l_d = _run(model, opt_d, clips, labels, 2)
        l_r = _run(ref, opt_r, clips, labels, 2)
        assert l_d == l_r, (l_d, l_r)
        for (name, p), q in zip(net.named_parameters(), ref.parameters()):
            if not p.requires_grad:                      # (the frozen temperature of an AttentionShift layer)
                assert p.grad is None and q.grad is None and torch.equal(p, q), name
                continue
            assert p.grad is not None and q.grad is not None, name
            assert torch.equal(p.grad, q.grad), "gradient of %s differs under DDP" % name
            assert torch.equal(p, q), "weight %s differs after the step under DDP" % name
        # the probe bench.py reports at N = 1: a gradient-sized self-reduce through RCCL
        buf = torch.ones(34 * (1 << 20) // 4, device=env.device)
        dist.all_reduce(buf)
        torch.cuda.synchronize()
        assert float(buf[0]) == 1.0 and float(buf[-1]) == 1.0
    finally:
        if created:
            dist.destroy_process_group()
