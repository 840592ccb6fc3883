"""World size 2 with the REAL kernels: two ranks, each its own process, sharing the one GPU of the box, the
gradient exchange over `gloo` (RCCL refuses two ranks on one device; what is under test is not the transport but everything
around it: the fused training blocks and their side-stream d(weight) kernels under DistributedDataParallel's bucket hooks with
a real second rank, clip sharding, per-replica d(shift) normalisation BEFORE the exchange, BatchNorm statistics per replica,
the batched tap softmax of the -aq variant).  One SGD step; rank r trains on its shard of the clips.

Expected result, exactly: DDP divides every bucket by the world size and sums, so the gradient each rank ends up with is
g0/2 + g1/2 of the two single-replica gradients -- computed here by the plain (un-wrapped) model on each shard in the same
processes -- bit for bit (halving is exact, a two-term sum commutes), and the post-step weights are identical on both ranks."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, variant, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import copy

    import torch.distributed as dist

    from rubiksnet_amd import RubiksNet, dp

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)                                         # same initialisation on both ranks
        net = RubiksNet("tiny", num_classes=13, num_frames=8, variant=variant, verbose=False).to(dev)
        g = torch.Generator().manual_seed(321)
        clips = torch.randn(4, 8, 3, 224, 224, generator=g)
        labels = torch.randint(0, 13, (4,), generator=g)
        lo, hi = dp.shard_range(4, rank, world)
        mine, lab = clips[lo:hi].to(dev), labels[lo:hi].to(dev)

        # the single-replica gradient of THIS rank's shard, from the plain model (same weights, same kernels)
        plain = copy.deepcopy(net).train()
        torch.nn.functional.cross_entropy(plain(mine), lab).backward()
        torch.cuda.synchronize()
        single = {k: p.grad.detach().cpu() for k, p in plain.named_parameters() if p.grad is not None}
        del plain

        model = torch.nn.parallel.DistributedDataParallel(net, device_ids=[0], bucket_cap_mb=25, gradient_as_bucket_view=True)
        opt = dp.make_optimizer(model, lr=1e-2, kind="sgd", momentum=0.0)
        model.train()
        loss = dp.train_step(model, opt, mine, lab)
        torch.cuda.synchronize()
        torch.save({"range": (lo, hi), "loss": float(loss),
                    "single": single,
                    "ddp": {k: p.grad.detach().cpu() for k, p in net.named_parameters() if p.grad is not None},
                    "state": {k: v.detach().cpu() for k, v in net.state_dict().items()}},
                   os.path.join(out_dir, "rank%d.pt" % rank))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("variant", ["rubiks3d", "rubiks3d-aq"])
def test_two_ranks_on_one_gpu_average_the_per_replica_gradients(tmp_path, variant):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, variant, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "rank0.pt")
    r1 = torch.load(tmp_path / "rank1.pt")
    assert r0["range"] == (0, 2) and r1["range"] == (2, 4)
    assert set(r0["ddp"]) == set(r1["ddp"]) == set(r0["single"]) == set(r1["single"]) and len(r0["ddp"]) > 60
    for k in r0["ddp"]:
        want = r0["single"][k] / 2 + r1["single"][k] / 2
        assert torch.equal(r0["ddp"][k], want), "rank 0: %s is not the mean of the two replicas' gradients" % k
        assert torch.equal(r1["ddp"][k], want), "rank 1: %s" % k
    # per-replica K5 / K9 before the exchange: every replica's shift gradient is a unit vector per channel, so the averaged one
    # is not (unless the replicas agree) -- the reference's DataParallel semantics, SURVEY 8e
    shifts = [k for k in r0["single"] if k.endswith("shift")]
    assert len(shifts) == 17
    for k in shifts:
        n0 = r0["single"][k].norm(dim=0)
        assert torch.allclose(n0[n0 > 0], torch.ones_like(n0[n0 > 0]), atol=1e-4), k
        assert float(r0["ddp"][k].norm(dim=0).max()) <= 1.0 + 1e-4
    # the replicas stay in lock-step: same weights after the step (BatchNorm's running statistics are per replica and differ)
    for k in r0["state"]:
        if "running_" in k or "num_batches" in k:
            continue
        assert torch.equal(r0["state"][k], r1["state"][k]), k
    assert any(not torch.equal(r0["state"][k], r1["state"][k]) for k in r0["state"] if "running_mean" in k)


def test_the_drivers_two_rank_bench_command_runs_on_one_gpu():
    """The command the driver's SCALE run issues for N = 2 -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node 2
    --master-addr 127.0.0.1 --master-port P bench.py --gpus 2 --steps K --warmup W` -- end to end on the box's one GPU:
    RK_DIST_BACKEND=gloo (dp.init_distributed) lets both ranks share the device, everything else is the real job -- the
    operator leg on both ranks, barrier + max-over-ranks timing, the Tiny train-step leg under DDP with its collective run-in,
    the all-reduce probe, rank-0-only legs ahead of the final barrier, ONE JSON line from rank 0."""
    import json
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["RK_DIST_BACKEND"] = "gloo"
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2",
           "--models", "tiny-train", "--model-steps", "2", "--no-legs"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-3000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["steps"] == 5 and out["warmup"] == 2
    assert out["scaling"] == "weak" and out["config"]["global_batch"] == 64 and out["config"]["per_gpu_batch"] == 32
    assert out["value"] > 0 and out["ms_per_step"] > 0
    # whole-job aggregate: the bytes of BOTH ranks over the slowest rank's time
    assert abs(out["value"] - 2 * out["config"]["algorithmic_bytes_per_step"] / (out["ms_per_step"] * 1e-3) / 1e9) < 1e-6 * out["value"]
    assert out["cpu_baseline"] is None                                # the CPU column is a one-GPU-job leg
    leg = out["models"]["tiny-train"]
    assert "error" not in leg, leg
    assert leg["global_batch"] == 64 and leg["parallelism"] == "dp2" and leg["ms_per_step"] > 0
    assert out["allreduce_probe"]["ranks"] == 2 and out["allreduce_probe"]["backend"] == "gloo"
