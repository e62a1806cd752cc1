"""HIP-graph capture of the forward+backward of one training step.

A step launches ~150 kernels, many of them a few microseconds long; issued from Python they leave the GPU idle
a large part of the time.  The shapes are static, so the whole forward + loss + backward is captured once into a HIP
graph (torch.cuda.CUDAGraph drives hipStreamBeginCapture on the stream our C-ABI launches use) and replayed.
What is NOT static is handled explicitly:
  * the FPS start indices are drawn on the CPU generator like the reference (pointnet_util.py:75), in the reference's
    order (SA1 then SA2, once per forward).  The graph reads them from STATIC DEVICE tensors; each replay is preceded by
    an eager host->device copy out of a ring of pinned staging buffers.  A staging buffer is rewritten only after the
    event recorded behind its last copy has completed, so a CPU running several replays ahead of the GPU can neither
    tear nor overtake the draws an in-flight step is going to read (the copy is stream-ordered before the replay that
    consumes it and after the previous replay that read the same device tensor);
  * dropout uses a device-resident counter that the captured kernels advance themselves;
  * BatchNorm momentum is a scalar kernel argument, i.e. baked into the graph: callers re-capture when the trainer's
    staircase schedule changes it (every 200 k samples);
  * the gradient all-reduce and the optimizer step stay outside the graph (eager), so the multi-GPU path does
    not depend on capturing RCCL collectives.
Capturing needs a few warm-up executions of the step; the BatchNorm running statistics / counters they touch and the
dropout counter are restored afterwards, so building a graph does not change the model state.
"""
import torch

from . import backbone as _bb
from . import ops as _ops

_RING = 4


class _PinnedStarts:
    """FPS start indices: CPU draw -> pinned ring slot -> (eager, stream-ordered) copy into the static device tensor the graph reads."""

    def __init__(self, device):
        self.device = device
        self.slots = []      # [(N, [pinned host buffers], dev)]
        self.events = [None] * _RING
        self.ring = 0
        self.cursor = 0

    def __call__(self, N, B):
        if self.cursor == len(self.slots):
            hs = [torch.empty(B, dtype=torch.long).pin_memory() for _ in range(_RING)]
            d = torch.empty(B, dtype=torch.long, device=self.device)
            hs[0].copy_(_bb.draw_fps_start(N, B))
            d.copy_(hs[0])                               # synchronous: the first user (warm-up / capture) sees the draw
            self.slots.append((N, hs, d))
        d = self.slots[self.cursor][2]
        self.cursor += 1
        return d

    def stage(self):
        """Draw the next step's indices in the reference's order (SA1 then SA2) and enqueue their copies on the current stream."""
        r = self.ring = (self.ring + 1) % _RING
        if self.events[r] is not None:
            self.events[r].synchronize()                 # the copy that last read this staging slot has finished
        for N, hs, d in self.slots:
            hs[r].copy_(_bb.draw_fps_start(N, hs[r].shape[0]))
            d.copy_(hs[r], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.events[r] = ev
        self.cursor = 0

    def set(self, values):
        """Test / replay hook: put the given (B,) index tensors (one per level, SA1 first) into the static device tensors now."""
        for (N, hs, d), v in zip(self.slots, values):
            d.copy_(v.to(torch.long))
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
    loop keeps only B of the 256 CUs busy, so it disappears behind the GEMMs.
    At construction prefetch_xyz must hold the clouds of the FIRST batch to be trained on (its geometry is computed here).

    draw_starts=False leaves the FPS start tensors alone between replays (callers that set them through `starts.set`).

    split_tail=True (data-parallel jobs): what follows the last gradient - the copies of the prefetched geometry into the 'current' buffers -
    is captured as a SECOND graph, replayed by `tail()`.  Between `__call__` and `tail()` the caller records an event and starts the
    gradient exchange on a side stream (ddp.FlatGradSync.allreduce_async): the exchange then overlaps the tail instead of waiting for it.
    (An external event-record node inside one graph would do the same; torch refuses external events on ROCm.)

    inference=True: fn runs without gradients; the prefetched geometry omits the inverse maps of the gathers (only a backward reads them)."""

    def __init__(self, model, fn, warmup=2, prefetch_xyz=None, draw_starts=True, stream=None, split_tail=False, inference=False):
        import gc
        gc.collect()       # autograd graphs of earlier eager steps that only reference cycles keep alive: their AccumulateGrad nodes are
        # bound to the stream they were created on, and autograd would order the capture stream against that stream (work the capture
        # never joins -> hipErrorStreamCaptureUnjoined)
        dev = next(model.parameters()).device
        self.starts = _PinnedStarts(dev)
        self.draw_starts = draw_starts
        self._hooked = []
        for m in model.modules():
            if isinstance(m, _bb.PointNetSetAbstraction) and not m.group_all and m.fps_start is None:
                m.fps_start = self.starts
                self._hooked.append(m)
        self.fn = fn
        self.prefetch = prefetch_xyz is not None
        self.split_tail = bool(split_tail) and self.prefetch
        self.graph_tail = None
        self._nxt = None
        main = torch.cuda.current_stream()
        # state the warm-up executions must not leave behind: BatchNorm running statistics / counters, the dropout counter
        keep = [(b, b.detach().clone()) for b in model.buffers()]
        seed = getattr(model, "_drop_seed", None)
        keep_seed = None if seed is None else seed.detach().clone()

        executed = [0]                       # bodies run for real (warm-up), not captured

        def body():
            _ops.step_done()                 # warm-up / capture passes: the previous pass's gradients are discarded
            if not torch.cuda.is_current_stream_capturing():
                executed[0] += 1
            if not self.prefetch:
                return fn(None)
            cap = torch.cuda.current_stream()
            side = self._side
            side.wait_stream(cap)                                   # fork
            with torch.cuda.stream(side):
                nxt = model.compute_geometry(prefetch_xyz, with_csr=not inference)
            out = fn(self.cur)
            cap.wait_stream(side)                                   # join
            self._nxt = nxt
            if not self.split_tail:
                tail_body()
            return out

        def tail_body():
            # ONE launch for the ~25 tensors (40 MB at B = 32 x 8192) instead of torch's two multi-tensor copies (21 us each: a block per 64 K
            # elements); the descriptors are kernel arguments, so the node survives the re-allocation of `nxt` by every capture
            _ops.copy_flat_batch(_flatten(self.cur), _flatten(self._nxt))

        self._side = torch.cuda.Stream() if self.prefetch else None

        def restore():          # BatchNorm running statistics / counters and the dropout counter as they were before warm-up and capture
            with torch.no_grad():
                for b, v in keep:
                    b.copy_(v)
                if keep_seed is not None and getattr(model, "_drop_seed", None) is not None:
                    model._drop_seed.copy_(keep_seed)
                elif keep_seed is None and getattr(model, "_drop_seed", None) is not None:
                    model._drop_seed.sub_(executed[0] * (0x9E3779B97F4A7C15 % (2 ** 62)))      # created by the first warm-up pass: take the warm-up
                    executed[0] = 0                                                            # advances back (backbone.py:311)

        try:
            if self.prefetch:
                with torch.no_grad():
                    self.cur = model.compute_geometry(prefetch_xyz, with_csr=not inference)     # geometry for the first replay
                self.starts.cursor = 0
            # warm-up passes and the capture run on ONE stream (`stream`, e.g. the stream the caller's whole loop lives on, or a private one):
            # the autograd accumulator nodes created by the warm-up are then bound to the stream that is captured
            cap = stream if stream is not None else torch.cuda.Stream()
            cap.wait_stream(main)
            with torch.cuda.stream(cap):
                for _ in range(warmup):
                    self.starts.cursor = 0
                    body()
                    if self.split_tail:
                        tail_body()
            main.wait_stream(cap)
            torch.cuda.synchronize()
            self.starts.cursor = 0
            self.graph = torch.cuda.CUDAGraph()
            # thread_local: only THIS thread's calls are policed while capturing.  With the default ("global") a HIP call from any other thread -
            # RCCL's watchdog polling its events in a multi-GPU job - invalidates the capture
            with torch.cuda.graph(self.graph, stream=cap, capture_error_mode="thread_local"):
                self.out = body()
            if self.split_tail:
                self.graph_tail = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph_tail, stream=cap, pool=self.graph.pool(), capture_error_mode="thread_local"):
                    tail_body()
        except BaseException:
            # a failed warm-up / capture must not leave 1-2 extra BatchNorm updates and an advanced dropout counter behind: the caller
            # falls back to eager launches of the SAME batch
            try:
                torch.cuda.synchronize()
                restore()
            except Exception:
                pass
            self.release()
            raise
        self.starts.cursor = 0
        restore()
        # the gradient tensors the graph leaves in .grad (static addresses): either the kernels' own output buffers, or - when fn packs
        # them (ddp.FlatGradSync.pack, captured) - views of the exchange's flat buffer.  Code between replays that re-points .grad
        # (an optimizer's zero_grad(set_to_none=True), a test) is undone before the next replay's results are consumed
        self._grads = [(p, p.grad) for p in model.parameters() if p.grad is not None]
        self._first = True
        _ops.step_done()

    def __call__(self, draw=True):
        """draw=False: no fresh FPS start draws for this replay (PipelinedForward's last batch: the geometry it prefetches is discarded, and
        the CPU generator must be left where the serial path leaves it)."""
        if draw and self.draw_starts and (self.prefetch or not self._first):
            # with the prefetch the geometry of replay k+1's batch is computed DURING replay k, from the next pair of draws; without it
            # the first replay consumes the pair drawn at construction
            self.starts.stage()
        self._first = False
        self.graph.replay()
        if self._grads and self._grads[0][0].grad is not self._grads[0][1]:
            for p, g in self._grads:
                p.grad = g
        return self.out

    def tail(self):
        """split_tail: the rest of the step (after the caller has gated its gradient exchange on the main replay).  Must be called once per
        `__call__`, before the next one."""
        if self.graph_tail is not None:
            self.graph_tail.replay()

    def release(self):
        """Detach the FPS-start hook from the model (before building a replacement graph)."""
        for m in self._hooked:
            if m.fps_start is self.starts:
                m.fps_start = None


class PipelinedForward:
    """Inference over a sequence of equal-shaped batches with the geometry hidden: ONE HIP graph per call runs the backbone forward of the
    CURRENT batch on the geometry computed by the previous call, while a forked stream inside the same graph computes the parameter-free
    geometry (FPS -> ball query -> grouped coordinates -> second FPS level -> 3-NN stencils -> inverse maps) of the NEXT batch - what
    point2cyl_amd/train.py does for training, for the evaluation loop of the reference (eval.py:231-268 knows its next batch: it iterates a
    DataLoader).  The serial forward spends 0.75 of its 1.84 ms in that latency-bound chain; pipelined, a batch costs max(forward, geometry).

        pf = PipelinedForward(model, first_pcs)          # geometry of the first batch is computed here
        for k in range(n):
            heads, sizes = pf(pcs[k + 1] if k + 1 < n else None)      # forward of batch k; starts the geometry of batch k + 1
            ... consume heads (a static buffer: overwritten by the next call) ...

    The FPS start indices are drawn on the CPU generator per batch in the reference's order (SA1 then SA2), one batch ahead of their use
    (a call without a next batch draws nothing: the generator ends where the serial path leaves it); the dropout counter advances per forward as in the serial path.  Outputs equal the serial
    `model.forward_heads(pcs)` on the same draws bit for bit in eval mode (tests/test_gpu_parity.py)."""

    def __init__(self, model, first_pcs, stream=None):
        if first_pcs.shape[2] != 3:
            raise ValueError("PipelinedForward: (B, N, 3) clouds only (normal_channel inputs take the serial path)")
        self.model = model
        self.cur = first_pcs.detach().float().contiguous().clone()
        self.nxt = self.cur.clone()
        dev = self.cur.device
        self.sizes = None

        def fn(geom):
            with torch.no_grad(), _ops.step_arena(dev):
                heads, sizes = model.forward_heads(self.cur, geom)
            self.sizes = sizes
            return {"heads": heads}

        self.graph = GraphedForwardBackward(model, fn, prefetch_xyz=self.nxt, stream=stream, inference=True)      # no inverse maps: forward only

    def __call__(self, next_pcs=None):
        """Forward of the current batch -> (heads (B*N, ld) static buffer, head sizes); next_pcs: the batch the NEXT call will return
        (None: there is none - the prefetch runs on stale clouds and is discarded)."""
        if next_pcs is not None:
            if tuple(next_pcs.shape) != tuple(self.nxt.shape):
                raise ValueError("PipelinedForward: batch of shape %s in a pipeline of %s" % (tuple(next_pcs.shape), tuple(self.nxt.shape)))
            self.nxt.copy_(next_pcs)
        out = self.graph(draw=next_pcs is not None)
        _ops.step_done()
        self.cur.copy_(self.nxt)          # stream-ordered behind the replay that read `cur`
        return out["heads"], self.sizes

    def release(self):
        self.graph.release()
