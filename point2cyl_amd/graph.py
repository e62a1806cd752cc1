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

    def __init__(self, model, fn, warmup=2, prefetch_xyz=None, draw_starts=True, stream=None, split_tail=False, inference=False, starts=None):
        import gc
        gc.collect()       # autograd graphs of earlier eager steps that only reference cycles keep alive: their AccumulateGrad nodes are
        # bound to the stream they were created on, and autograd would order the capture stream against that stream (work the capture
        # never joins -> hipErrorStreamCaptureUnjoined)
        dev = next(model.parameters()).device
        self.starts = starts if starts is not None else _PinnedStarts(dev)      # (PipelinedForward injects its per-batch-ordered group draws)
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


class _GroupStarts(_PinnedStarts):
    """FPS start indices for G batches of B clouds computed as ONE geometry call: the draws are made batch by batch in the reference's
    order (SA1 then SA2 of batch 0, then of batch 1, ...: pointnet_util.py:75 once per level and forward) and laid side by side into the
    (G x B,) device tensors the grouped call reads, so every batch sees the draws the serial loop would have given it."""

    def __init__(self, device, levels, B, G):
        super().__init__(device)
        self.B, self.G = B, G
        for N in levels:
            hs = [torch.empty(G * B, dtype=torch.long).pin_memory() for _ in range(_RING)]
            self.slots.append((N, hs, torch.empty(G * B, dtype=torch.long, device=device)))
        self.stage(G)
        torch.cuda.current_stream().synchronize()

    def stage(self, n_batches=None):
        n = self.G if n_batches is None else n_batches
        assert n == self.G
        r = self.ring = (self.ring + 1) % _RING
        if self.events[r] is not None:
            self.events[r].synchronize()
        B = self.B
        for j in range(n):
            for N, hs, d in self.slots:
                hs[r][j * B:(j + 1) * B].copy_(_bb.draw_fps_start(N, B))
        for N, hs, d in self.slots:
            d.copy_(hs[r], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.events[r] = ev
        self.cursor = 0


def _slice_geometry(g, j, G):
    """Batch j's part of a geometry computed for G batches at once (every tensor's leading dimension is a multiple of G)."""
    if isinstance(g, dict):
        return {k: _slice_geometry(v, j, G) for k, v in g.items()}
    if isinstance(g, (tuple, list)):
        return tuple(_slice_geometry(v, j, G) for v in g)
    if g is None:
        return None
    per = g.shape[0] // G
    return g[j * per:(j + 1) * per]


class PipelinedForward:
    """Inference over a sequence of equal-shaped batches with the geometry hidden: ONE HIP graph per call runs the backbone forward of the
    CURRENT group of G batches (batch by batch: BatchNorm statistics and the dropout counter per batch, as the serial loop) on the geometry
    computed by the previous call, while a forked stream inside the same graph computes the parameter-free geometry (FPS -> ball query ->
    grouped coordinates -> second FPS level -> 3-NN stencils; no inverse maps: only a backward reads them) of the NEXT group - what
    point2cyl_amd/train.py does for training, for the evaluation loop of the reference (eval.py:231-268 knows its next batches: it iterates a
    DataLoader).  group = G > 1 computes the geometry of G batches TOGETHER: farthest point sampling is 512 dependent steps on one CU per
    cloud (0.52 ms whether 32 or 128 clouds are sampled), so its latency is shared by G batches - the stage FPS + ball query + grouped MLP
    goes from 0.83 ms per batch (G = 1: the FPS chain is the period) to 0.40 ms (G = 3) and 0.35 ms (G = 4), DESIGN.md 5.0.

        pf = PipelinedForward(model, first_group, group=G)          # first_group: list of G (B, N, 3) tensors; their geometry is computed here
        while groups remain:
            outs = pf(next_group or None)                            # [(heads, sizes)] * G for the CURRENT group; starts the geometry of next_group
            ... consume (the heads are static buffers: overwritten by the next call) ...

    The FPS start indices are drawn on the CPU generator per batch in the reference's order (SA1 then SA2), one group ahead of their use (a
    call without a next group draws nothing: the generator ends where the serial path leaves it); the dropout counter advances per forward as in
    the serial path.  Outputs equal the serial `model.forward_heads(pcs)` on the same draws bit for bit in eval mode (tests/test_gpu_parity.py)."""

    def __init__(self, model, first, stream=None, group=1):
        single = torch.is_tensor(first)
        first = [first] if single else list(first)
        G = int(group)
        if len(first) != G:
            raise ValueError("PipelinedForward: group = %d but %d first batches were given" % (G, len(first)))
        if any(f.shape != first[0].shape for f in first) or first[0].shape[2] != 3:
            raise ValueError("PipelinedForward: G equal-shaped (B, N, 3) batches (normal_channel inputs take the serial path)")
        self.model, self.G, self.single = model, G, single
        B, N, _ = first[0].shape
        self.B = B
        self.cur = torch.cat([f.detach().float() for f in first], 0).contiguous()
        self.nxt = self.cur.clone()
        dev = self.cur.device
        self.sizes = None

        def fn(geom):
            outs = []
            for j in range(G):
                _ops.step_done()
                with torch.no_grad(), _ops.step_arena(dev):
                    heads, sizes = model.forward_heads(self.cur[j * B:(j + 1) * B], _slice_geometry(geom, j, G))
                outs.append(heads)
                self.sizes = sizes
            return {"heads%d" % j: h for j, h in enumerate(outs)}

        starts = _GroupStarts(dev, [N, model.sa1.npoint], B, G) if G > 1 else None
        self.graph = GraphedForwardBackward(model, fn, prefetch_xyz=self.nxt, stream=stream, inference=True, starts=starts)      # no inverse maps: forward only

    def __call__(self, next_pcs=None):
        """Forward of the current group -> [(heads (B*N, ld) static buffer, head sizes)] * G (a single pair when constructed from one tensor);
        next_pcs: the batch(es) the NEXT call will return - a tensor (G = 1), a list of exactly G tensors, or None (there is no further full
        group: the prefetch runs on stale clouds and is discarded; leftover batches take the serial forward after release())."""
        n_next = 0
        if next_pcs is not None:
            nxt = [next_pcs] if torch.is_tensor(next_pcs) else list(next_pcs)
            if len(nxt) != self.G or any(tuple(t.shape) != (self.B,) + tuple(self.nxt.shape[1:]) for t in nxt):
                raise ValueError("PipelinedForward: next group of %d batches / shapes %s in a pipeline of %d x %s (a short last group takes the "
                                 "serial forward: its empty slots would run real forwards - BatchNorm / dropout side effects - on stale clouds)"
                                 % (len(nxt), [tuple(t.shape) for t in nxt], self.G, (self.B,) + tuple(self.nxt.shape[1:])))
            for j, t in enumerate(nxt):
                self.nxt[j * self.B:(j + 1) * self.B].copy_(t)
            n_next = len(nxt)
        if self.G > 1:
            if n_next:
                self.graph.starts.stage(n_next)       # per-batch draws in the serial order, only for the batches that exist
            out = self.graph(draw=False)
        else:
            out = self.graph(draw=n_next > 0)
        _ops.step_done()
        self.cur.copy_(self.nxt)          # stream-ordered behind the replay that read `cur`
        res = [(out["heads%d" % j], self.sizes) for j in range(self.G)]
        return res[0] if self.single else res

    def release(self):
        self.graph.release()
