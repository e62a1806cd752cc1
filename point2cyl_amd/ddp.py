"""Data-parallel glue: one process per GPU, batch sharded by cloud, ONE gradient all-reduce per step.

The reference has no distributed code at all; every op of the path is independent per
cloud except the train-mode BatchNorm statistics (kept per replica, like running the reference on each
shard) and the final scalar means.  All 1,404,243 gradients (5.6 MB fp32) are exchanged as ONE flat buffer:
after backward a single multi-tensor copy packs them into a persistent buffer (the copy is part of the captured
HIP graph when the step is replayed), one RCCL all-reduce over xGMI averages it in place, and the parameters'
.grad are views of that buffer from then on - nothing is allocated or unpacked per step.  The exchange is issued on the step's own
stream right behind the replay (`allreduce`, the default) or from a side stream gated on the replay (`allreduce_async` / `wait`), with the
step's tail - the copies of the next batch's prefetched geometry, a second small graph - under it (measured: the side-stream form costs
0.06 ms of replay / event overhead to hide a 0.04 ms tail).  With backend "gloo"
the same code runs on CPU tensors for the tests.  With world_size 1 nothing is packed at all - unless P2C_FORCE_EXCHANGE=1, which
creates a one-rank process group and sends every step through the real exchange (pack, RCCL all-reduce on the side stream, wait): AVG over
one rank is the identity, so a forced run must reproduce the plain run bit for bit (ddp_selftest.py, the one-GPU proof that librccl loads,
supports ncclAvg on gfx950 and orders correctly between the two graph replays of a split-tail step).

BatchNorm buffers: per-replica running statistics during training (no SyncBN upstream either); `average_buffers`
makes the checkpoint hold the MEAN of the replicas' statistics instead of rank 0's alone.
"""
import os

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC handles between the ranks' processes (read when HIP initialises)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def force_exchange():
    return os.environ.get("P2C_FORCE_EXCHANGE", "0") == "1"


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun contract)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("P2C_ONE_GPU_RANKS"):          # test hook: several ranks share GPU 0 (RCCL refuses that; use the gloo backend)
        local, backend = 0, backend or "gloo"
    if (world > 1 or force_exchange()) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            if local >= torch.cuda.device_count():
                raise RuntimeError("rank %d wants GPU %d but this node exposes %d: one process per GPU (RCCL refuses shared devices)"
                                   % (rank, local, torch.cuda.device_count()))
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def backend_name():
    return dist.get_backend() if dist.is_initialized() else "none"


class FlatGradSync:
    """One collective per step.  zero() before backward (autograd then adopts the kernels' output buffers: no accumulate
    kernels); pack() right after backward (ONE multi-tensor copy into the persistent flat buffer, .grad -> views of it; inside a
    captured step this copy is a graph node and the re-pointing happens once); allreduce() averages the buffer in place."""

    def __init__(self, params, world_size=None):
        self.params = [p for p in params if p.requires_grad]
        self.world = world_size if world_size is not None else (dist.get_world_size() if dist.is_initialized() else 1)
        self.flat = None
        self.views = None
        self._src = None          # keeps the packed-from tensors of a captured step alive
        self.active = self.world > 1 or (force_exchange() and dist.is_initialized())     # is there an exchange at all
        self._avg_ok = True
        self.exchanges = 0        # all-reduces issued (a forced one-rank run must count one per step)
        self._side, self._pending = None, False

    def zero(self):
        for p in self.params:
            p.grad = None

    def _alloc(self):
        p0 = self.params[0]
        self.flat = torch.zeros(sum(p.numel() for p in self.params), dtype=p0.dtype, device=p0.device)
        self.views, o = [], 0
        for p in self.params:
            self.views.append(self.flat[o:o + p.numel()].view_as(p))
            o += p.numel()

    def pack(self):
        if not self.active:
            return
        if self.flat is None:
            self._alloc()
        with torch.no_grad():
            src, dst, missing = [], [], []
            for p, v in zip(self.params, self.views):
                if p.grad is None:
                    missing.append(v)
                elif p.grad.data_ptr() != v.data_ptr():
                    src.append(p.grad)
                    dst.append(v)
            if missing:
                torch._foreach_zero_(missing)
            if src:
                torch._foreach_copy_(dst, src)
            self._src = src
            for p, v in zip(self.params, self.views):
                p.grad = v

    def allreduce(self):
        if not self.active:
            return
        if self.flat is None or any(p.grad is None or p.grad.data_ptr() != v.data_ptr() for p, v in zip(self.params, self.views)):
            self.pack()           # eager callers that did not pack after backward
        self.exchanges += 1
        with torch.no_grad():
            if dist.get_backend() == "nccl" and self._avg_ok:
                try:
                    dist.all_reduce(self.flat, op=dist.ReduceOp.AVG)
                    return
                except (RuntimeError, ValueError):      # a collective library without ncclAvg: sum and scale from here on
                    self._avg_ok = False
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
            if self.world > 1:
                self.flat.div_(self.world)

    def allreduce_async(self):
        """The exchange on a SIDE stream, gated on everything enqueued on the current stream so far (the replay that produced the gradients):
        work the caller enqueues next on the current stream (graph.GraphedForwardBackward.tail) overlaps it.  `wait()` before the optimizer."""
        if not self.active:
            return
        if not self.params[0].is_cuda:
            return self.allreduce()
        if self._side is None:
            self._side = torch.cuda.Stream(self.params[0].device)
        ev = torch.cuda.Event()
        ev.record()
        self._side.wait_event(ev)
        with torch.cuda.stream(self._side):
            self.allreduce()
        self._pending = True

    def wait(self):
        if self._pending:
            torch.cuda.current_stream().wait_stream(self._side)
            self._pending = False


def preflight(modules, device=None):
    """Before anything is captured or timed in a job of several ranks: ONE eager all-reduce on the device (does the collective library work
    on this node at all - xGMI / IPC / device visibility problems surface here, with the backend's own error text, not inside a graph
    capture) and proof that the replicas start from the same parameters (float64 sum of squares, gathered).  Raises RuntimeError with
    the backend's message; returns a dict for the bench / trainer report.  world_size 1: nothing."""
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not force_exchange()):
        return None
    world, rank = dist.get_world_size(), dist.get_rank()
    backend = dist.get_backend()
    if isinstance(modules, torch.nn.Module):
        modules = [modules]
    params = [p for m in modules for p in m.parameters()]
    dev = device if device is not None else params[0].device
    try:
        probe = torch.full((1024,), float(rank + 1), dtype=torch.float32, device=dev)
        dist.all_reduce(probe, op=dist.ReduceOp.SUM)
        if probe.is_cuda:
            torch.cuda.synchronize(dev)
        want = world * (world + 1) / 2.0
        if not bool((probe == want).all()):
            raise RuntimeError("all-reduce(SUM) of rank+1 over %d ranks returned %r, expected %r" % (world, float(probe[0]), want))
        with torch.no_grad():
            ck = torch.zeros(1, dtype=torch.float64, device=dev)
            for p in params:
                ck += (p.detach().double() ** 2).sum()
        got = [torch.zeros_like(ck) for _ in range(world)]
        dist.all_gather(got, ck)
        sums = [float(g) for g in got]
    except Exception as e:
        raise RuntimeError("point2cyl_amd.ddp.preflight: the eager collective check failed on rank %d of %d (backend %s, device %s): %s: %s"
                           % (rank, world, backend, dev, type(e).__name__, e)) from e
    if any(s_ != sums[0] for s_ in sums):
        raise RuntimeError("point2cyl_amd.ddp.preflight: replicas do not start from identical parameters (sum of squares per rank: %r); "
                           "broadcast_module must run before the first step" % (sums,))
    return dict(backend=backend, world_size=world, eager_allreduce_ok=True, param_checksum=sums[0], params_identical=True)


def broadcast_module(module, src=0):
    """Same initial parameters and buffers on every rank."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src)


def average_buffers(module):
    """Mean over the replicas of every floating-point buffer (the BatchNorm running statistics), in place on every rank; integer
    counters (num_batches_tracked) are identical already.  Called before a checkpoint is written."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    bufs = [b for b in module.buffers() if b.dtype.is_floating_point]
    if not bufs:
        return
    with torch.no_grad():
        flat = torch.cat([b.reshape(-1) for b in bufs])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat.div_(dist.get_world_size())
        o = 0
        for b in bufs:
            b.copy_(flat[o:o + b.numel()].view_as(b))
            o += b.numel()


def shard_range(n_items, rank, world):
    """Contiguous shard [lo, hi) of n_items for this rank (clouds are the unit; no data-path collective)."""
    per = (n_items + world - 1) // world
    lo = min(n_items, rank * per)
    return lo, min(n_items, lo + per)
