"""Data-parallel glue: one process per GPU, batch sharded by cloud, ONE gradient all-reduce per step.

The reference has no distributed code at all; every op of the path is independent per
cloud except the train-mode BatchNorm statistics (kept per replica, like running the reference on each
shard) and the final scalar means.  All 1,404,243 gradients (5.6 MB fp32) live in ONE flat buffer that the
parameters' .grad tensors are views of, so the exchange is a single RCCL all-reduce over xGMI with no
packing copies; with backend "gloo" the same code runs on CPU tensors for the tests.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun contract)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("P2C_ONE_GPU_RANKS"):          # test hook: several ranks share GPU 0 (RCCL refuses that; use the gloo backend)
        local, backend = 0, backend or "gloo"
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


class FlatGradSync:
    """One collective per step: after backward the per-parameter gradients are packed into ONE flat buffer
    (a single concat kernel), all-reduced, averaged, and the parameters' .grad are re-pointed at views of it
    (no unpack copy).  With world_size 1 nothing is packed at all.  Call zero() before backward."""

    def __init__(self, params, world_size=None):
        self.params = [p for p in params if p.requires_grad]
        self.world = world_size if world_size is not None else (dist.get_world_size() if dist.is_initialized() else 1)
        self.flat = None

    def zero(self):
        for p in self.params:
            p.grad = None          # autograd then adopts the kernels' output buffers: no accumulate kernels

    def allreduce(self):
        if self.world <= 1:
            return
        with torch.no_grad():
            self.flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in self.params])
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
            self.flat.div_(self.world)
            o = 0
            for p in self.params:
                p.grad = self.flat[o:o + p.numel()].view_as(p)
                o += p.numel()


def broadcast_module(module, src=0):
    """Same initial parameters and buffers on every rank."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src)


def shard_range(n_items, rank, world):
    """Contiguous shard [lo, hi) of n_items for this rank (clouds are the unit; no data-path collective)."""
    per = (n_items + world - 1) // world
    lo = min(n_items, rank * per)
    return lo, min(n_items, lo + per)
