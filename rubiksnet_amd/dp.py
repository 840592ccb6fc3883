"""Clip-level data parallelism: one process per GPU, torch.distributed over RCCL/xGMI.

The reference has no distributed code (its only multi-GPU call site is nn.DataParallel in
scripts/test_models.py:153).  The shift operator shards along N with no data-path collective:
forward and d(x) are per-(n, c) plane, d(shift) is reduced -- and L2-normalised -- inside each
replica's backward exactly as under the reference's DataParallel.  The only
exchange is the gradient all-reduce of a training step, done by DDP buckets (backend "nccl"
is RCCL on ROCm; "gloo" on CPU for tests).
"""
import contextlib
import os
import time

import torch
import torch.distributed as dist
import torch.nn as nn

from . import attention_shift, pointwise

__all__ = [
    "DistEnv", "init_distributed", "shard_range", "wrap_ddp", "make_optimizer", "train_step",
    "timed_region", "barrier", "ensure_process_group",
]


class DistEnv:
    def __init__(self, rank, local_rank, world_size, device, backend):
        self.rank, self.local_rank, self.world_size = rank, local_rank, world_size
        self.device, self.backend = device, backend

    @property
    def is_main(self):
        return self.rank == 0

    @property
    def distributed(self):
        return self.world_size > 1


def init_distributed(prefer_gpu=True):
    """Read RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torchrun contract) and join the job."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    use_gpu = prefer_gpu and torch.cuda.is_available()
    if use_gpu:
        torch.cuda.set_device(local_rank % torch.cuda.device_count())
        device = torch.device("cuda", local_rank % torch.cuda.device_count())
    else:
        device = torch.device("cpu")
    backend = "nccl" if use_gpu else "gloo"
    # RK_DIST_BACKEND=gloo: the exchange over gloo although the tensors live on a GPU -- a TEST arrangement: it lets N ranks
    # share one device (RCCL refuses that), so that the whole N > 1 job -- launcher, rendezvous, DDP around the real kernels,
    # the collective run-in, barrier + max-over-ranks timing -- runs on a one-GPU box (tests/test_dist_world2_gpu.py)
    forced = os.environ.get("RK_DIST_BACKEND")
    if forced in ("gloo", "nccl"):
        backend = forced
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if use_gpu and backend == "nccl":
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return DistEnv(rank, local_rank, world, device, backend)


def ensure_process_group(env):
    """A process group for a job of ONE rank (RCCL on a GPU, gloo on CPU): what `bench.py --gpus 1` uses so that its
    all-reduce probe and a DDP-wrapped replica run through the same RCCL code path as an N-rank job -- the bucket hooks,
    `gradient_as_bucket_view` and the side-stream d(weight) kernels then meet on a device without an 8-GPU node.
    Returns True when this call created the group (the caller destroys it)."""
    if dist.is_initialized():
        return False
    import socket
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if "MASTER_PORT" not in os.environ:
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(sock.getsockname()[1])
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC only on this driver (RCCL needs it)
    if env.device.type == "cuda" and env.backend == "nccl":
        dist.init_process_group(env.backend, rank=env.rank, world_size=env.world_size, device_id=env.device)
    else:
        dist.init_process_group(env.backend, rank=env.rank, world_size=env.world_size)
    return True


def shard_range(global_batch, rank, world_size):
    """[lo, hi) of the clips rank `rank` owns; remainder clips go to the lowest ranks."""
    base, rem = divmod(global_batch, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def wrap_ddp(model, env, bucket_cap_mb=25, force=False):
    """DDP replica.  Gradient volume is tiny (Tiny 7.6 MB, Large 34 MB fp32), so a single default
    bucket size keeps the all-reduce count low; xGMI rings are per-link bound (~153 GB/s), i.e.
    ~0.4 ms for Large -- hidden behind backward by bucket overlap.  `force`: wrap at world size 1 too
    (needs a process group: ensure_process_group)."""
    if not env.distributed and not (force and dist.is_initialized()):
        return model
    kwargs = dict(bucket_cap_mb=bucket_cap_mb, gradient_as_bucket_view=True)
    if env.device.type == "cuda":
        return nn.parallel.DistributedDataParallel(model, device_ids=[env.device.index], **kwargs)
    return nn.parallel.DistributedDataParallel(model, **kwargs)


def make_optimizer(model, lr=0.01, lr_shift_mult=0.01, kind="adam", momentum=0.9, weight_decay=0.0):
    """Two groups as in scripts/example_finetune.py:49-64: parameters whose name ends with
    'shift' train at lr * lr_shift_mult."""
    shift_params, regular = [], []
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        (shift_params if name.endswith("shift") else regular).append(p)
    groups = [{"params": shift_params, "lr": lr * lr_shift_mult}, {"params": regular}]
    if kind == "adam":
        # one fused multi-tensor launch per parameter group instead of ~20 foreach kernels per step (RubiksNet-Large: 43
        # multi_tensor_apply launches, 0.6 ms); same update rule.  CPU parameters (the gloo tests) keep the default.
        on_gpu = all(p.is_cuda and p.is_floating_point() for g in groups for p in g["params"])
        if on_gpu and (shift_params or regular):
            try:
                return torch.optim.Adam(groups, lr=lr, weight_decay=weight_decay, fused=True)
            except (RuntimeError, TypeError):
                pass
        return torch.optim.Adam(groups, lr=lr, weight_decay=weight_decay)
    return torch.optim.SGD(groups, lr=lr, momentum=momentum, weight_decay=weight_decay)


def train_step(model, optimizer, clips, labels, criterion=None):
    """zero_grad -> forward -> CE -> backward (+ DDP all-reduce) -> step
    (scripts/example_finetune.py:85-97)."""
    criterion = criterion or nn.functional.cross_entropy
    optimizer.zero_grad(set_to_none=True)
    bf16_step = (clips.is_cuda and torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16)
    # (bf16 autocast: the 1x1 weights packed for the bf16 GEMMs once for the whole step, pointwise.prepacked)
    # (and the [C, 3] tap softmax of every AttentionShift layer evaluated in one launch, attention_shift.presoftened)
    with (pointwise.prepacked(model) if bf16_step else contextlib.nullcontext()):
        with (attention_shift.presoftened(model) if clips.is_cuda else contextlib.nullcontext()):
            out = model(clips)
        loss = criterion(out, labels)
        loss.backward()
    optimizer.step()
    return loss


def barrier(env):
    if env.distributed:
        if env.device.type == "cuda" and env.backend == "nccl":
            dist.barrier(device_ids=[env.device.index])
        else:
            dist.barrier()


def timed_region(env, fn, steps):
    """barrier + synchronize, run fn() `steps` times, synchronize + barrier; returns the MAX over
    ranks of the elapsed seconds."""
    cuda = env.device.type == "cuda"
    barrier(env)
    if cuda:
        torch.cuda.synchronize(env.device)
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    if cuda:
        torch.cuda.synchronize(env.device)
    barrier(env)
    dt = time.perf_counter() - t0
    if env.distributed:
        t = torch.tensor([dt], dtype=torch.float64, device=env.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt
