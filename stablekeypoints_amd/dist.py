"""One process per MI355X; the only exchange step of the path is the SUM all-reduce of the learned
embedding's gradient (T*768 fp32 = 236,544 B at T=77) once per optimizer step -- RCCL over xGMI
(`backend="nccl"` is RCCL on ROCm), `gloo` for the CPU tests.  Replaces the implicit
replicate/scatter/gather of `nn.DataParallel` (optimize_token.py:41-50)."""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def init_from_env(backend: str | None = None):
    """Initialise torch.distributed from RANK/WORLD_SIZE/MASTER_* (torchrun contract). No-op for 1 rank."""
    world, rank, local = env_world()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = os.environ.get("SKP_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)            # one process per GPU; RCCL binds to the current device
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return world, rank, local


def world_size() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank() -> int:
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def allreduce_sum_(t: torch.Tensor) -> torch.Tensor:
    """In-place SUM over ranks (no-op for a single rank)."""
    if world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def broadcast_(t: torch.Tensor, src: int = 0) -> torch.Tensor:
    if world_size() > 1:
        dist.broadcast(t, src=src)
    return t


def barrier():
    if world_size() > 1:
        dist.barrier()


def shard_indices(perm, rank_: int, world: int):
    """Rank r takes elements r, r+world, ... of a global permutation (drop_last semantics)."""
    n = (len(perm) // world) * world
    return perm[rank_:n:world]


class EmbeddingReducer:
    """Gradient exchange + identical optimizer step on every rank.

    Each rank accumulates d(sum of its images' losses)/d(context) locally; `step()` all-reduces the SUM,
    which -- with every per-image loss pre-divided by the GLOBAL batch size -- equals the reference's
    mean over the DataParallel devices and the accumulation steps (optimize.py:405-425).  All ranks then
    apply the same Adam update, so `context` stays bit-identical without any parameter broadcast.
    """

    def __init__(self, context: torch.Tensor, optimizer: torch.optim.Optimizer):
        self.context, self.optimizer = context, optimizer
        broadcast_(context.data, 0)

    def step(self):
        if self.context.grad is None:
            self.context.grad = torch.zeros_like(self.context)
        allreduce_sum_(self.context.grad)
        self.optimizer.step()
        self.optimizer.zero_grad(set_to_none=False)
