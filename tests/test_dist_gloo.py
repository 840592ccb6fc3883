"""world_size-2 `gloo` tests of the data-parallel harness on CPU (the N > 1 path of bench.py).

The shift operator has no CPU path in the product, so -- in THIS TEST ONLY -- the functional is
swapped for an autograd.Function backed by the CPU oracle (test infrastructure).  What is under
test is the harness: sharding, DDP gradient averaging, per-replica shift-grad normalisation,
barrier + max-over-ranks timing."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _oracle_shift(*args, **kwargs):
    from oracle.torch_shift import oracle_shift        # the CPU oracle as an autograd.Function (test infrastructure)
    return oracle_shift(*args, **kwargs)


class _TinyVideoNet(torch.nn.Module):
    """conv -> RubiksShift3D -> conv -> pool -> fc on [N, T, 3, 8, 8] clips."""

    def __init__(self, T=4, C=6, classes=5):
        super().__init__()
        from rubiksnet_amd.shiftlib import RubiksShift3D
        self.T = T
        self.conv = torch.nn.Conv2d(3, C, 1, bias=False)
        self.as3 = RubiksShift3D(C)
        self.as3.shift_function = _oracle_shift      # test-only: the product has no CPU path
        self.fc = torch.nn.Linear(C, classes)

    def forward(self, clips):
        n = clips.shape[0]
        f = self.conv(clips.view(-1, 3, 8, 8))
        f = self.as3(f.view(n, self.T, -1, 8, 8)).reshape(n * self.T, -1, 8, 8)
        return self.fc(f.mean(dim=(2, 3))).view(n, self.T, -1).mean(dim=1)


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(1)
    from rubiksnet_amd import dp
    env = dp.init_distributed(prefer_gpu=False)
    assert env.backend == "gloo" and env.world_size == world and env.rank == rank

    torch.manual_seed(0)                       # same init on every rank
    net = _TinyVideoNet()
    model = dp.wrap_ddp(net, env)
    opt = dp.make_optimizer(model, lr=0.1, lr_shift_mult=0.5, kind="sgd", momentum=0.0)
    assert len(opt.param_groups[0]["params"]) == 1 and opt.param_groups[0]["lr"] == pytest.approx(0.05)

    g = torch.Generator().manual_seed(123)     # one global batch, identical on every rank, then sharded
    clips = torch.randn(6, 4, 3, 8, 8, generator=g)
    labels = torch.randint(0, 5, (6,), generator=g)
    lo, hi = dp.shard_range(6, rank, world)
    loss = dp.train_step(model, opt, clips[lo:hi], labels[lo:hi])
    dt = dp.timed_region(env, lambda: None, 3)
    state = {k: v.detach().clone() for k, v in net.state_dict().items()}
    grads = {k: p.grad.detach().clone() for k, p in net.named_parameters()}
    torch.save({"state": state, "grads": grads, "loss": float(loss), "dt": dt, "range": (lo, hi)},
               os.path.join(out, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_partitions_the_batch():
    from rubiksnet_amd import dp
    for gb in (1, 7, 8, 256):
        for w in (1, 2, 3, 8):
            r = [dp.shard_range(gb, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == gb
            assert all(r[i][1] == r[i + 1][0] for i in range(w - 1))
            assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1


def test_ddp_gloo_world2(tmp_path, oracle):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "rank0.pt")
    r1 = torch.load(tmp_path / "rank1.pt")
    assert r0["range"] == (0, 3) and r1["range"] == (3, 6)
    assert r0["dt"] == r1["dt"] >= 0.0                        # MAX over ranks is shared
    # replicas stay in lock-step: identical averaged grads and identical weights after the step
    for k in r0["grads"]:
        assert torch.allclose(r0["grads"][k], r1["grads"][k], atol=1e-7), k
    for k in r0["state"]:
        assert torch.equal(r0["state"][k], r1["state"][k]), k

    # single-process reference: the DDP gradient is the MEAN over ranks of per-replica grads; the shift
    # grad is normalised inside each replica's backward before the all-reduce
    g = torch.Generator().manual_seed(123)
    clips = torch.randn(6, 4, 3, 8, 8, generator=g)
    labels = torch.randint(0, 5, (6,), generator=g)
    per_rank = []
    for lo, hi in ((0, 3), (3, 6)):
        torch.manual_seed(0)
        net = _TinyVideoNet()
        torch.nn.functional.cross_entropy(net(clips[lo:hi]), labels[lo:hi]).backward()
        per_rank.append({k: p.grad.clone() for k, p in net.named_parameters()})
    for k in per_rank[0]:
        want = 0.5 * (per_rank[0][k] + per_rank[1][k])
        assert torch.allclose(r0["grads"][k], want, atol=1e-6), k
    unit = per_rank[0]["as3.shift"].norm(dim=0)
    assert np.allclose(unit.numpy(), 1.0, atol=1e-5)          # each replica's shift grad is unit-norm per channel


def test_bench_launches_itself_for_two_ranks_dry_run():
    """`python bench.py --gpus 2` with no torchrun environment re-executes itself under torch.distributed.run on
    127.0.0.1; without a GPU it takes the dry-run path (launcher, rendezvous, barrier + max-over-ranks
    timing, all-reduce probe on gloo) and rank 0 prints ONE JSON line."""
    import json
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["CUDA_VISIBLE_DEVICES"] = ""           # force the CPU path even on a GPU box
    env["HIP_VISIBLE_DEVICES"] = ""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--dry-run"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["dry_run"] is True and out["backend"] == "gloo"
    assert out["allreduce_probe"]["ranks"] == 2 and out["allreduce_probe"]["bus_GBps"] > 0
    assert out["steps"] == 2 and out["scaling"] == "weak" and out["metric"].startswith("RubiksShift3D")
    # the model legs' control flow on a stub model: both ranks left run_in() after the same number of (collective) steps
    # although their step times differ, and the timed bracket reports the SLOWEST rank's time
    stub = out["model_leg_stub"]
    assert len(stub["run_in_steps"]) == 2 and stub["run_in_steps"][0] == stub["run_in_steps"][1] >= 12
    assert stub["ms_per_step"] >= stub["slowest_rank_sleep_ms"]
