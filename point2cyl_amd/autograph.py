"""HIP graphs INSIDE `backbone.forward`: what makes an unchanged caller fast.

The reference's trainer calls `model(pcs)` and later `total_loss.backward()` (train_Point2Cyl_without_sketch.py:244, :368).  Launched from
Python the ~150 kernels of the backbone's forward + backward leave the GPU idle most of the time (13.8 ms per step against 4 ms of kernel
time).  point2cyl_amd.step / graph.py solve that for OUR trainer by capturing the whole step; a caller that only knows the nn.Module interface
gets the same effect here: per input shape the module captures TWO graphs - its forward, and the backward of that forward
(torch.autograd.grad from the head outputs to the parameters, captured with a static upstream-gradient buffer) - and `forward` becomes one
autograd node that replays the first in its forward and the second in its backward.  The caller's loss code, optimizer and `.item()`
reads stay whatever they are.

What is not static is handled like graph.py does: FPS start indices are drawn on the CPU generator per call (pointnet_util.py:75) and
travel through pinned staging into static device tensors; the dropout counter and the BatchNorm running statistics are device state the
captured kernels advance themselves; the BatchNorm momentum is a kernel argument, so a change of momentum (train...:357-360) is part of the
cache key and triggers a new capture.  Parameter gradients are REAL autograd outputs: the backward graph packs them into one static flat
buffer, `_Replay.backward` takes ONE copy of it (5.6 MB, one launch) and returns per-parameter views of that fresh copy - so
`loss.backward()` (AccumulateGrad adopts the views without a further copy when `.grad` is None, adds otherwise), `torch.autograd.grad(loss,
params)`, `backward(inputs=...)`, parameter hooks and torch's DistributedDataParallel all see them like any other node's gradients.

Contract (the same as torch.cuda.make_graphed_callables): the returned head tensors are views of a static buffer that the NEXT forward of
the same shape overwrites, and the saved activations live inside the graph: a backward through a forward that is no longer the LATEST
replay of its graph (two forwards of one shape, then backward of the first: gradient accumulation over micro-batches is fine as long as
each micro-batch does forward -> backward in turn) RAISES instead of differentiating the wrong activations; no double backward through the
module; the input needs no gradient.  Anything outside the contract - test hooks (dropout_mask, fps_start), an input that requires grad,
P2C_AUTOGRAPH=0, a capture that fails - takes the eager path."""
import os
import warnings
import weakref

import torch

ENABLED = os.environ.get("P2C_AUTOGRAPH", "1") != "0"
_MAX_CACHED = 4
_STATE = weakref.WeakKeyDictionary()      # model -> dict(graphs, bns, params, failed): kept off the module so that deepcopy / pickling never meet a graph


def _state(model):
    st = _STATE.get(model)
    if st is None:
        slots = [(m._parameters, n) for m in model.modules() for n, p in m._parameters.items() if p is not None]
        slots += [(m._buffers, n) for m in model.modules() for n, b in m._buffers.items() if b is not None]
        st = _STATE[model] = dict(graphs={}, bns=_bn_modules(model), slots=slots, failed=False)
    return st


def _live_ptrs(model):
    """Storage addresses of the module's CURRENT parameters and buffers (the dict slots are read every call, ~20 us: `p.data = ...`,
    `model.cpu(); model.cuda()`, load_state_dict(assign=True), a replaced nn.Parameter or buffer all show up here)."""
    return tuple(d[n].data_ptr() for d, n in _state(model)["slots"])


def reset(model=None):
    """Drop the cached graphs (of one model, or of all): after replacing parameters, or to free the graphs' memory pools."""
    if model is None:
        _STATE.clear()
    else:
        _STATE.pop(model, None)


def _bn_modules(model):
    return [m for m in model.modules() if isinstance(m, torch.nn.modules.batchnorm._BatchNorm)]


class _Captured:
    """Forward graph (+ backward graph when gradients are enabled) of `model.forward_heads` at one input shape."""

    def __init__(self, model, x, want_grad):
        from .graph import _PinnedStarts
        from . import backbone as _bb
        from torch.nn.utils import stateless
        dev = x.device
        named = list(model.named_parameters())
        self.params = [p for _, p in named if p.requires_grad] if want_grad else []
        self.want_grad = want_grad and len(self.params) > 0
        # The capture never touches the caller's autograd leaves: the module runs on ALIASES of its parameters (same storage, fresh leaves).
        # A parameter that has been through an eager backward owns an AccumulateGrad node bound to the stream of that backward (the
        # default stream of an unchanged trainer); routing the captured backward towards it makes autograd order the capturing stream
        # against that stream - work the capture never joins, which this ROCm ends with a segmentation fault in capture_end.
        alias = {n: (p.detach().requires_grad_(True) if (self.want_grad and p.requires_grad) else p.detach()) for n, p in named}
        leaves = [alias[n] for n, p in named if p.requires_grad] if self.want_grad else []
        self.x = torch.empty_like(x).copy_(x)
        self.starts = _PinnedStarts(dev)
        hooked = [m for m in model.modules() if isinstance(m, _bb.PointNetSetAbstraction) and not m.group_all]
        keep = [(b, b.detach().clone()) for b in model.buffers()]
        seed = getattr(model, "_drop_seed", None)
        keep_seed = None if seed is None else seed.detach().clone()
        main = torch.cuda.current_stream()
        cap = torch.cuda.Stream()
        self.fwd = self.bwd = None

        def fwd_body():
            self.starts.cursor = 0
            return model.forward_heads(self.x)

        def restore():
            with torch.no_grad():
                for b, v in keep:
                    b.copy_(v)
                if keep_seed is not None and getattr(model, "_drop_seed", None) is not None:
                    model._drop_seed.copy_(keep_seed)
                elif keep_seed is None and getattr(model, "_drop_seed", None) is not None:
                    model._drop_seed.sub_(2 * (0x9E3779B97F4A7C15 % (2 ** 62)))      # created by the first warm-up pass: take the two warm-up advances back

        for m in hooked:
            m.fps_start = self.starts
        try:
            import gc
            gc.collect()
            cap.wait_stream(main)
            with torch.cuda.stream(cap), torch.set_grad_enabled(self.want_grad), stateless._reparametrize_module(model, alias):
                for _ in range(2):                                         # warm-up: allocator, lazily created state, autograd accumulators
                    heads, sizes = fwd_body()
                    if self.want_grad:
                        torch.autograd.grad((heads,), leaves, (torch.ones_like(heads),), allow_unused=True)
                    del heads
                cap.synchronize()
                pool = torch.cuda.graph_pool_handle()
                self.fwd = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.fwd, pool=pool, stream=cap, capture_error_mode="thread_local"):
                    heads, sizes = fwd_body()
                self.heads, self.sizes = heads.detach(), sizes
                if self.want_grad:
                    self.gout = torch.zeros_like(heads)
                    self.bwd = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(self.bwd, pool=pool, stream=cap, capture_error_mode="thread_local"):
                        grads = torch.autograd.grad((heads,), leaves, (self.gout,), allow_unused=True)
                        # .grad in the parameter's own layout (a (64,3,1,1) weight's gradient arrives as a slice of the kernels' 4-padded
                        # rows): one mismatch sends torch.optim's multi-tensor kernels down their one-launch-per-tensor path (measured:
                        # Adam.step 0.35 -> 0.85 ms)
                        # (packed: element ranges of ONE flat buffer in the parameters' own contiguous layout)
                        have = [(g, p) for g, p in zip(grads, self.params) if g is not None]
                        self.flat = torch.cat([g.reshape(-1) for g, _ in have]) if have else None
                    self.has_grad = [g is not None for g in grads]
                    # how _Replay.backward cuts a copy of `flat` into per-parameter gradients with the fewest Python-level tensor operations
                    # (123 slice + view pairs cost 0.6 ms of host time per step): ONE split_with_sizes, a view only where the parameter is not 1-D
                    self.grad_pos = [i for i, h in enumerate(self.has_grad) if h]
                    self.grad_sizes = [self.params[i].numel() for i in self.grad_pos]
                    self.grad_views = [(j, tuple(self.params[i].shape)) for j, i in enumerate(self.grad_pos) if self.params[i].dim() != 1]
                    del grads, have
            main.wait_stream(cap)
            torch.cuda.synchronize()
        finally:
            for m in hooked:
                if m.fps_start is self.starts:
                    m.fps_start = None
            try:
                torch.cuda.synchronize()
                restore()
            except Exception:
                pass
        self.starts.cursor = 0
        self._first = True
        self.generation = 0          # number of forward replays: a backward checks that it differentiates the latest one

    def replay_forward(self, x):
        if x.data_ptr() != self.x.data_ptr():
            self.x.copy_(x)
        if not self._first:
            self.starts.stage()          # fresh CPU draws for this call (the first replay consumes the pair drawn at construction)
        self._first = False
        self.generation += 1
        self.fwd.replay()
        return self.heads.detach()       # a fresh alias per call: autograd stamps its node on the object the Function returns


class _Replay(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cap, x, *params):
        ctx.cap = cap
        ctx.set_materialize_grads(False)
        out = cap.replay_forward(x)
        ctx.generation = cap.generation
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gout):
        cap = ctx.cap
        n = len(cap.params)
        if gout is None:
            return (None, None) + (None,) * n
        if ctx.generation != cap.generation:
            raise RuntimeError("point2cyl_amd.autograph: backward through a forward pass that is not the latest one of its shape - the activations "
                               "saved inside the HIP graph now belong to forward #%d, this backward belongs to #%d.  Run forward -> backward per "
                               "(micro-)batch, or set P2C_AUTOGRAPH=0 P2C_STAGE_WEIGHTS=0 for the eager path with per-forward weight copies (INTEGRATION.md, 'autograd contract')"
                               % (cap.generation, ctx.generation))
        if gout.data_ptr() != cap.gout.data_ptr():
            cap.gout.copy_(gout)
        cap.bwd.replay()
        if cap.flat is None:
            return (None, None) + (None,) * n
        fresh = cap.flat.clone()            # the static buffer is overwritten by the next replay; autograd gets tensors of its own (ONE launch)
        pieces = list(fresh.split_with_sizes(cap.grad_sizes))
        for j, shape in cap.grad_views:
            pieces[j] = pieces[j].view(shape)
        if len(pieces) == n:
            return (None, None) + tuple(pieces)
        outs = [None] * n
        for j, i in enumerate(cap.grad_pos):
            outs[i] = pieces[j]
        return (None, None) + tuple(outs)


def _key(model, x, want_grad):
    # the LIVE parameters and buffers: tensors that got new storage must not meet a graph captured on the old one
    st = _state(model)
    # (every BatchNorm's own mode: model.train(); model.sa1.eval() is a different launch sequence than the all-train one)
    return (tuple(x.shape), x.dtype, x.device.index, model.training, tuple(b.training for b in st["bns"]), want_grad, tuple(b.momentum for b in st["bns"]),
            _live_ptrs(model), tuple(d[n].requires_grad for d, n in st["slots"]) if want_grad else None)


def applicable(model, x):
    if not ENABLED or not x.is_cuda or x.requires_grad or torch.cuda.is_current_stream_capturing():
        return False
    if model.dropout_mask is not None or model.sa1.fps_start is not None or model.sa2.fps_start is not None:
        return False          # test hooks / an outer graph (graph.py) own these: eager path
    return not _state(model)["failed"]


def forward_heads(model, x):
    """-> (heads, sizes) like backbone.forward_heads, through the cached graphs of x's shape (captured on first use)."""
    st = _state(model)
    want_grad = torch.is_grad_enabled() and any(d[n].requires_grad for d, n in st["slots"])
    cache = st["graphs"]
    key = _key(model, x, want_grad)
    cap = cache.get(key)
    if cap is None:
        try:
            cap = _Captured(model, x, want_grad)
        except Exception as e:          # keep the caller alive: eager launches from here on
            st["failed"] = True
            warnings.warn("point2cyl_amd.autograph: HIP-graph capture of backbone.forward failed (%s: %s); continuing with eager launches"
                          % (type(e).__name__, e))
            return None
        if len(cache) >= _MAX_CACHED:
            cache.pop(next(iter(cache)))
        cache[key] = cap
    if cap.want_grad:
        heads = _Replay.apply(cap, x, *cap.params)
    else:
        heads = cap.replay_forward(x)
    return heads, cap.sizes
