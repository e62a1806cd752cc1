"""HIP-graph capture of the forward+backward of one training step.

A step launches ~400 kernels, many of them a few microseconds long; issued from Python they leave the GPU idle
~20% of the time.  The shapes are static, so the whole forward + loss + backward is captured once into a HIP
graph (torch.cuda.CUDAGraph drives hipStreamBeginCapture on the stream our C-ABI launches use) and replayed.
What is NOT static is handled explicitly:
  * the FPS start indices are drawn on the CPU generator like the reference (pointnet_util.py:75); they go
    through a pinned host buffer that is refreshed before every replay and copied to the device INSIDE the graph;
  * dropout uses torch's graph-safe Philox offsets;
  * the gradient all-reduce and the optimizer step stay outside the graph (eager), so the multi-GPU path does
    not depend on capturing RCCL collectives.
"""
import torch

from . import backbone as _bb


class _PinnedStarts:
    """FPS start indices: CPU draw -> pinned buffer -> device copy that is part of the captured graph."""

    def __init__(self, device):
        self.device = device
        self.slots = []      # [(N, host_pinned, dev)]
        self.cursor = 0

    def __call__(self, N, B):
        if self.cursor == len(self.slots):
            h = torch.empty(B, dtype=torch.long).pin_memory()
            self.slots.append((N, h, torch.empty(B, dtype=torch.long, device=self.device)))
            h.copy_(_bb.draw_fps_start(N, B))
        N_, h, d = self.slots[self.cursor]
        self.cursor += 1
        d.copy_(h, non_blocking=True)
        return d

    def refresh(self):
        """Draw the next step's indices in the reference's order (SA1 then SA2)."""
        for N, h, _ in self.slots:
            h.copy_(_bb.draw_fps_start(N, h.shape[0]))
        self.cursor = 0


def _flatten(g):
    out = []
    for v in g.values():
        if isinstance(v, dict):
            out += _flatten(v)
        elif isinstance(v, (tuple, list)):
            out += _flatten(dict(enumerate(v)))
        elif v is not None:
            out.append(v)
    return out


class GraphedForwardBackward:
    """fn(geom) must run forward + backward on static input tensors and return a dict of tensors.

    prefetch_xyz: when given (the NEXT batch's clouds, a static tensor the caller refills before each replay), the
    parameter-free geometry of that batch (FPS, ball query, 3-NN: backbone.compute_geometry) is computed on a forked
    stream INSIDE the same graph while the main stream trains on the current batch with the geometry produced by the
    previous replay; the results are copied into the static 'current' buffers after the join.  The 512-step FPS
    loop keeps only B of the 256 CUs busy, so it disappears behind the GEMMs."""

    def __init__(self, model, fn, warmup=2, prefetch_xyz=None):
        dev = next(model.parameters()).device
        self.starts = _PinnedStarts(dev)
        for m in model.modules():
            if isinstance(m, _bb.PointNetSetAbstraction) and not m.group_all and m.fps_start is None:
                m.fps_start = self.starts
        self.fn = fn
        self.prefetch = prefetch_xyz is not None
        main = torch.cuda.current_stream()

        def body():
            if not self.prefetch:
                return fn(None)
            cap = torch.cuda.current_stream()
            side = self._side
            side.wait_stream(cap)                                   # fork
            with torch.cuda.stream(side):
                nxt = model.compute_geometry(prefetch_xyz)
            out = fn(self.cur)
            cap.wait_stream(side)                                   # join
            by_dtype = {}
            for dst, src in zip(_flatten(self.cur), _flatten(nxt)):
                by_dtype.setdefault(dst.dtype, ([], []))[0].append(dst)
                by_dtype[dst.dtype][1].append(src)
            for dsts, srcs in by_dtype.values():                    # one fused launch per dtype instead of ~25 copies
                torch._foreach_copy_(dsts, srcs)
            return out

        self._side = torch.cuda.Stream() if self.prefetch else None
        if self.prefetch:
            with torch.no_grad():
                self.cur = model.compute_geometry(prefetch_xyz)     # geometry for the first replay
            self.starts.cursor = 0
        warm = torch.cuda.Stream()
        warm.wait_stream(main)
        with torch.cuda.stream(warm):
            for _ in range(warmup):
                self.starts.cursor = 0
                body()
        main.wait_stream(warm)
        torch.cuda.synchronize()
        self.starts.cursor = 0
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = body()
        self.starts.refresh()
        # the gradient tensors the graph writes (static addresses).  A data-parallel exchange re-points .grad at views of its flat
        # buffer after every step (ddp.FlatGradSync.allreduce); the next replay still writes HERE, so .grad is pointed back first
        self._grads = [(p, p.grad) for p in model.parameters() if p.grad is not None]

    def __call__(self):
        self.graph.replay()
        if self._grads and self._grads[0][0].grad is not self._grads[0][1]:
            for p, g in self._grads:
                p.grad = g
        self.starts.refresh()      # host work for the NEXT step overlaps this step's GPU time
        return self.out
