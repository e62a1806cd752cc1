"""Torch-facing wrappers of the HIP kernels (C ABI in include/p2c_hip.h).

Everything here is plumbing: allocate outputs with torch, pass raw device pointers + the current HIP
stream to libp2c_hip.so, and register backward passes with torch.autograd.  No arithmetic of the hot
path happens in torch ops in this file.  All feature tensors are POINT-MAJOR: [rows, channels].
"""
import os

import torch

from . import _lib
from ._lib import PROFILE, call, ptr, stream  # noqa: F401

I32 = torch.int32
PENDING_NBT = []      # num_batches_tracked counters to bump with ONE foreach add per forward (see backbone.forward)
USE_FUSED_BWD = os.environ.get("P2C_FUSED_BWD", "1") != "0"
USE_PRE_LINEAR = os.environ.get("P2C_PRE_LINEAR", "1") != "0"   # first conv of a gather-fed stack on the sparse rows (gather.hip)
USE_FUSED256 = os.environ.get("P2C_FUSED256", "1") != "0"   # 256-wide layers through the fused backward in two passes (A/B switch)
USE_DUAL_BWD = os.environ.get("P2C_DUAL_BWD", "1") != "0"   # dX and dW of the small layers in one launch (gemm_dual_kernel)
USE_FOLD0 = os.environ.get("P2C_FOLD0", "1") != "0"         # first layer with <= 4 input channels never materialised (bn.hip)
USE_CSR_BWD = os.environ.get("P2C_CSR_BWD", "1") != "0"      # gather-formulated backward of the gathers (no atomics)
USE_NARROW_BWD = os.environ.get("P2C_NARROW_BWD", "1") != "0"  # the per-point heads' backward in one pass (csrc/heads.hip)
STRICT_LABELS = os.environ.get("P2C_STRICT_LABELS", "0") == "1"   # validate labels with a device->host sync in every loss call instead of deferred
USE_POOL_ALG = os.environ.get("P2C_POOL_ALG", "1") != "0"    # pooled last layer's backward without its pre-BN output (csrc/bwd_pool.hip)
USE_POOL_EPI = os.environ.get("P2C_POOL_EPI", "1") != "0"    # max over 64 neighbours from extremes emitted by the last layer's GEMM epilogue
USE_INFER_PATHS = True     # eval mode without gradients takes the folded first layer / the pooled last layer without its Y too (a test switches it off)
USE_BN_EVAL_BATCH = os.environ.get("P2C_BN_EVAL_BATCH", "1") != "0"    # every eval-mode BatchNorm affine of a forward from one launch
_EVAL_AFF = [None]        # {running_mean.data_ptr(): (4, C) affine} while a forward that ran its BNEvalStage is in flight, else None


def _f32c(t):
    return t if (t.dtype == torch.float32 and t.is_contiguous()) else t.contiguous().float()


# ------------------------------------------------------------------------------------------ geometry
def fps(xyz, npoint, start):
    """farthest_point_sample (pointnet_util.py:63-84).  xyz (B,N,3) cuda, start (B,) int64 (the CPU
    randint draw of :75) -> idx (B,npoint) int32, new_xyz (B,npoint,3)."""
    _lib.require_device(xyz)
    xyz = _f32c(xyz)
    B, N, _ = xyz.shape
    start = start.to(device=xyz.device, dtype=torch.int64).contiguous()
    idx = torch.empty(B, npoint, dtype=I32, device=xyz.device)
    new_xyz = torch.empty(B, npoint, 3, dtype=torch.float32, device=xyz.device)
    call("p2c_fps_f32", ptr(xyz), B, N, ptr(start), npoint, ptr(idx), ptr(new_xyz), stream())
    return idx, new_xyz


def ball_query(radius, nsample, xyz, new_xyz):
    """query_ball_point (pointnet_util.py:87-107) -> (B,S,nsample) int32."""
    _lib.require_device(xyz, new_xyz)
    xyz, new_xyz = _f32c(xyz), _f32c(new_xyz)
    B, N, _ = xyz.shape
    S = new_xyz.shape[1]
    idx = torch.empty(B, S, nsample, dtype=I32, device=xyz.device)
    r2 = float(torch.tensor(radius ** 2, dtype=torch.float32))
    call("p2c_ball_query_f32", ptr(xyz), ptr(new_xyz), B, N, S, r2, nsample, ptr(idx), stream())
    return idx


def three_nn(xyz1, xyz2, return_dist=False):
    """3 nearest of xyz2 for each xyz1 point + normalised inverse-distance weights (pointnet_util.py:301-307)."""
    _lib.require_device(xyz1, xyz2)
    xyz1, xyz2 = _f32c(xyz1), _f32c(xyz2)
    B, N, _ = xyz1.shape
    S = xyz2.shape[1]
    idx = torch.empty(B, N, 3, dtype=I32, device=xyz1.device)
    w = torch.empty(B, N, 3, dtype=torch.float32, device=xyz1.device)
    d = torch.empty(B, N, 3, dtype=torch.float32, device=xyz1.device) if return_dist else None
    call("p2c_three_nn_f32", ptr(xyz1), ptr(xyz2), B, N, S, ptr(idx), ptr(w), ptr(d), stream())
    return (idx, w, d) if return_dist else (idx, w)


def build_csr(idx, T, w=None, ediv=1):
    """Inverse map of a gather: idx (B, ...) int32 with values in [0,T) -> (offsets (B,T+1), rows (B,E) int32, wsorted (B,E) | None):
    for every target t the source rows (entry // ediv) of the entries that read it, and their weights in the same order."""
    B = idx.shape[0]
    flat = idx.reshape(B, -1)
    E = flat.shape[1]
    offsets = torch.empty(B, T + 1, dtype=I32, device=idx.device)
    rows = torch.empty(B, E, dtype=I32, device=idx.device)
    ws = torch.empty(B, E, dtype=torch.float32, device=idx.device) if w is not None else None
    call("p2c_build_csr_i32", ptr(flat), ptr(w) if w is not None else None, B, E, ediv, T, ptr(offsets), ptr(rows),
         ptr(ws) if ws is not None else None, stream())
    return offsets, rows, ws


class _GroupGather(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz, feats, new_xyz, idx, csr=None):
        B, N, _ = xyz.shape
        S, ns = idx.shape[1], idx.shape[2]
        D = 0 if feats is None else feats.shape[-1]
        ldo = (3 + D + 3) // 4 * 4
        out = torch.empty(B * S * ns, ldo, dtype=torch.float32, device=xyz.device)
        if feats is not None:
            feats = _f32c(feats)
        call("p2c_group_gather_f32", ptr(xyz), ptr(feats), D, ptr(new_xyz), ptr(idx), B, N, S, ns, D, ptr(out), ldo, 1, stream())
        ctx.save_for_backward(idx)
        ctx.csr = csr
        ctx.dims = (B, N, S, ns, D, ldo)
        return out

    @staticmethod
    def backward(ctx, dout):
        (idx,) = ctx.saved_tensors
        B, N, S, ns, D, ldo = ctx.dims
        if D == 0 or not ctx.needs_input_grad[1]:
            return None, None, None, None, None
        if dout.stride(1) != 1:
            dout = dout.contiguous()
        if USE_CSR_BWD and D <= 256:
            offsets, rows, _ = ctx.csr if ctx.csr is not None else build_csr(idx, N)
            dfeats = torch.empty(B, N, D, dtype=torch.float32, device=dout.device)
            call("p2c_csr_gather_f32", ptr(dout), dout.stride(0), 0, ptr(offsets), ptr(rows), None, B, S * ns, S * ns, N, D,
                 ptr(dfeats), D, stream())
        else:
            dfeats = torch.zeros(B, N, D, dtype=torch.float32, device=dout.device)
            call("p2c_group_gather_bwd_f32", ptr(dout), dout.stride(0), ptr(idx), B, N, S, ns, D, ptr(dfeats), D, 1, stream())
        return None, dfeats, None, None, None


def group_gather(xyz, feats, new_xyz, idx, csr=None):
    """rows (b,s,j) = [feats[idx] | xyz[idx]-new_xyz | 0-pad] -> (B*S*ns, ld): the reference's concat (pointnet_util.py:137)
    with the feature block first ("xyz_last"); mlp_stack(..., xyz_last=True) permutes the first conv's input channels to match."""
    return _GroupGather.apply(_f32c(xyz), feats, _f32c(new_xyz), idx, csr)


class _ThreeInterp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feats, idx, w, csr=None):
        B, S, C = feats.shape
        N = idx.shape[1]
        feats = _f32c(feats)
        out = torch.empty(B * N, C, dtype=torch.float32, device=feats.device)
        call("p2c_three_interp_f32", ptr(feats), C, ptr(idx), ptr(w), B, N, S, C, ptr(out), C, stream())
        ctx.save_for_backward(idx, w)
        ctx.csr = csr
        ctx.dims = (B, N, S, C)
        return out

    @staticmethod
    def backward(ctx, dout):
        idx, w = ctx.saved_tensors
        B, N, S, C = ctx.dims
        if dout.stride(1) != 1:
            dout = dout.contiguous()
        if USE_CSR_BWD and C <= 256:
            offsets, rows, ws = ctx.csr if ctx.csr is not None else build_csr(idx, S, w, 3)
            dfeats = torch.empty(B, S, C, dtype=torch.float32, device=dout.device)
            call("p2c_csr_gather_f32", ptr(dout), dout.stride(0), 0, ptr(offsets), ptr(rows), ptr(ws), B, N * 3, N, S, C, ptr(dfeats), C,
                 stream())
        else:
            dfeats = torch.zeros(B, S, C, dtype=torch.float32, device=dout.device)
            call("p2c_three_interp_bwd_f32", ptr(dout), dout.stride(0), ptr(idx), ptr(w), B, N, S, C, ptr(dfeats), C, stream())
        return dfeats, None, None, None


def three_interpolate(feats, idx, w, csr=None):
    """(B,S,C), idx/w (B,N,3) -> (B*N, C)  (pointnet_util.py:308).  csr = build_csr(idx, S) if already available."""
    return _ThreeInterp.apply(feats, idx, w, csr)


# ------------------------------------------------------------------------------------------ MLP stack
_DEFER_NBT = [False]
_NBT_BUMPED = [False]      # this forward's num_batches_tracked were advanced by the weight-staging launch (WeightStage.run(bump=...)): nothing to queue


def flush_nbt():
    """num_batches_tracked += 1 for every BatchNorm touched since the last flush, as one multi-tensor add."""
    if PENDING_NBT:
        torch._foreach_add_(PENDING_NBT, 1)
        del PENDING_NBT[:]


class BNState:
    """Per-layer BatchNorm buffers handed to the stack (updated in place by the finalize kernel)."""

    def __init__(self, running_mean, running_var, num_batches_tracked, momentum, eps):
        self.running_mean, self.running_var, self.nbt = running_mean, running_var, num_batches_tracked
        self.momentum, self.eps = momentum, eps


def _pad4(n):
    return (n + 3) // 4 * 4


def _weight_slots(L, tail):
    """Indices of the weight tensors in the flat params list [W,b,(gamma,beta)]*."""
    out, pi = [], 0
    for i in range(L):
        out.append(pi)
        pi += 2 if (tail == "linear" and i == L - 1) else 4
    return out


STAT_SLOTS = 64


def _slots(C, dev):
    """Zeroed fp64 accumulator rows for one per-channel reduction (see include/p2c_hip.h, P2C_STAT_SLOTS)."""
    return torch.zeros(STAT_SLOTS, 2, C, dtype=torch.float64, device=dev)


class _StepArena:
    """All zero-initialised scratch of one training step (fp64 stat slots, per-XCD dW copies, bias gradients) comes out
    of ONE buffer that is cleared by one fill at the start of the step instead of one fill per stack and direction.
    Only active inside `with ops.step_arena(dev):`; the first step measures the size, later steps reuse the buffer
    (static address: HIP-graph friendly).  Tensors handed out alias the buffer and die at the next step's start, so the
    caller must have consumed the gradients by then - true when .grad already exists and autograd accumulates into it
    (ddp.FlatGradSync) or when the optimizer steps before the next forward."""

    def __init__(self):
        self.buf, self.off, self.need, self.run, self.active = None, 0, 0, 0, False
        self.epoch = 0            # number of arena steps begun: a backward that reads an arena slice checks it is still the step that wrote it
        self.pending = False      # gradients handed out by the last arena step have not been consumed yet (see step_done)

    def take(self, n_f64, dev):
        self.run += n_f64
        if self.active and self.buf is not None and self.buf.device == dev and self.off + n_f64 <= self.buf.numel():
            out = self.buf[self.off:self.off + n_f64]
            self.off += n_f64
            return out
        return torch.zeros(n_f64, dtype=torch.float64, device=dev)


STEP_ARENA = _StepArena()


def step_done():
    """The gradients of the last `with step_arena(...)` step have been consumed (optimizer stepped, or copied away): the next
    arena step may clear the buffer they alias."""
    STEP_ARENA.pending = False


class step_arena:
    """`with ops.step_arena(dev): forward; backward` then consume the gradients and call ops.step_done().  Entering a second
    arena step before that raises: parameter gradients of the first step are views of the buffer the second would clear
    (gradient accumulation over several backward passes must run WITHOUT the arena, or copy .grad away first)."""

    def __init__(self, dev):
        self.dev = torch.device(dev)

    def __enter__(self):
        A = STEP_ARENA
        if A.pending:
            raise RuntimeError("point2cyl_amd.ops.step_arena: the previous arena step's gradients were not marked consumed (ops.step_done()); "
                               "they alias the buffer this step is about to clear")
        if A.buf is None or A.buf.numel() < A.need or A.buf.device != self.dev:
            A.buf = torch.zeros(max(A.need, 1), dtype=torch.float64, device=self.dev)
        else:
            A.buf.zero_()
        A.off, A.run, A.active = 0, 0, True
        A.epoch += 1
        return A

    def __exit__(self, *exc):
        A = STEP_ARENA
        A.need, A.active = max(A.need, A.run), False
        A.pending = exc[0] is None
        return False


class _ZeroArena:
    """One zero-filled allocation carved into the many small accumulators a stack needs (fp64 stat slots, the 8
    per-XCD copies of each dW, bias gradients): one fill kernel instead of dozens (none inside a step_arena)."""

    def __init__(self, n_f64, dev):
        self.buf = STEP_ARENA.take(n_f64, dev)
        self.off = 0

    def _take(self, n64):
        if self.off + n64 > self.buf.numel():          # reservation too small (it is an estimate): a zero tensor of its own
            return STEP_ARENA.take(n64, self.buf.device)
        out = self.buf[self.off:self.off + n64]
        self.off += n64
        return out

    def f64(self, *shape):
        n = 1
        for d in shape:
            n *= d
        return self._take(n).view(*shape)

    def f32(self, *shape):
        n = 1
        for d in shape:
            n *= d
        return self._take((n + 1) // 2).view(torch.float32)[:n].view(*shape)


def _zero_padded(t, *shape):
    """t copied into the top-left corner of a zero tensor of `shape`, the zeros coming out of the step arena (already cleared by the
    step's one fill) - F.pad would launch a fill and a copy for each of the half-dozen odd-sized weights of a step."""
    n = 1
    for d in shape:
        n *= d
    buf = STEP_ARENA.take((n + 1) // 2, t.device).view(torch.float32)[:n].view(*shape)
    buf[tuple(slice(0, d) for d in t.shape)].copy_(t)
    return buf


def _eval_affine(bn, Co, infer):
    """The (4, Co) affine BNEvalStage's launch at the top of this forward produced for `bn`, or None (train mode, gradients wanted, no stage)."""
    if not infer or _EVAL_AFF[0] is None:
        return None
    st = _EVAL_AFF[0].get(bn.running_mean.data_ptr())
    return st if (st is not None and st.shape[1] == Co) else None


class _MLPStack(torch.autograd.Function):
    """A chain of 1x1-conv layers  Y_i = act_{i-1}(Y_{i-1}) W_i^T + b_i  with BatchNorm+ReLU folded into the
    NEXT layer's operand load.  tail: 'maxpool' (max over ns of relu(bn(Y_last))), 'bnrelu' (materialise
    relu(bn(Y_last))), 'linear' (last layer has no BN; `drop_mask` multiplies its input).
    params = [W_0, b_0, gamma_0, beta_0, W_1, ...] (gamma/beta absent for a BN-less last layer).
    The kernels want channel counts that are multiples of 4: X0 carries zero columns up to pad4(in_channels)
    and odd-sized weights are zero-padded here (131 -> 132 inputs, 19 -> 20 head outputs)."""

    @staticmethod
    def forward(ctx, cfg, X0, *params):
        dev = X0.device
        # pre = "linear before the gather" (csrc/gather.hip): X0 holds the SPARSE rows; layer 0 runs on them and its dense pre-BN
        # output (pre["rows"] rows) is produced by the gather itself
        pre = cfg.get("pre")
        rep_v = None
        if pre is not None and pre["kind"] == "repeat":       # params[0] is the per-group vector V (G, D2), a differentiable input
            rep_v, params = params[0], params[1:]
        Ms, ldx0 = X0.shape[0], X0.stride(0)
        M = pre["rows"] if pre is not None else Ms
        K = _pad4(cfg["in_channels"])
        assert X0.shape[1] >= K and ldx0 % 4 == 0, (X0.shape, K)
        training = cfg["training"]
        # INFERENCE: eval-mode BatchNorm and no gradient wanted by anyone - nothing of this call is read again, so the forms that keep an
        # activation out of HBM (folded first layer, pooled last layer without its Y) apply as they do in training, minus the statistics
        infer = (not training) and USE_INFER_PATHS and not cfg.get("wants_grad", True)
        bns = cfg["bns"]
        L = cfg["n_layers"]
        tail = cfg["tail"]
        mask = cfg.get("drop_mask")
        seed = cfg.get("drop_seed")       # device int64 scalar: counter-hash dropout (no mask tensor)
        dscale = cfg.get("drop_scale", 1.0)
        Ys, aff, Ws = [], [], []
        pool = None
        fold_b0 = pre_wx = pre_wb = None
        X, ldx, in_mode, sc, sh = X0, ldx0, 0, None, None
        pi = 0
        arena = None
        if training:
            widths = [_pad4(params[j].shape[0]) for j in _weight_slots(L, tail)]
            arena = _ZeroArena(sum(STAT_SLOTS * 2 * c for c in widths) + 16, dev)
        # Folded first layer (csrc/bn.hip): 3 input channels (+pad), 64 outputs, train mode, no gradient wanted for the input,
        # a BatchNorm'ed middle layer of 64/128 channels next: Y_0 is never written; layer 1 rebuilds it from the input rows.
        fold0 = (USE_FOLD0 and pre is None and (training or infer) and K == 4 and cfg["in_channels"] <= 3 and ldx0 == 4 and M >= 8192 and not X0.requires_grad and mask is None and seed is None
                 and (L >= 3 or (L == 2 and tail == "bnrelu")) and params[0].shape[0] == 64 and params[4].shape[0] in (64, 128)
                 and bns[0] is not None and bns[1] is not None)
        mom = None
        staged = cfg.get("staged")
        stage_gen = None
        if staged:
            for sd_ in staged.values():
                stg = getattr(sd_.get("W2"), "_p2c_stage", None) if sd_ else None
                if stg is not None:
                    stage_gen = (stg, stg.generation)
        mptr_free = mask is None and seed is None
        for i in range(L):
            has_bn = not (tail == "linear" and i == L - 1)
            W, b = params[pi], params[pi + 1]
            pi += 2
            Co_true = W.shape[0]
            Co = _pad4(Co_true)
            st = staged.get(i) if staged else None
            if st is not None:
                # every operand derived from this layer's parameters was written by the step's ONE batched copy (WeightStage): same
                # layouts as the branch below produces, persistent buffers, no launch here
                W2 = st["W2"]
                b = st.get("b", b)
                pre_wb = st.get("pre_wb", pre_wb)
                pre_wx = st.get("pre_wx", pre_wx)
                assert tuple(W2.shape) == (Co, K), (tuple(W2.shape), Co, K)
            else:
                W2 = W.reshape(Co_true, -1)
                if i == 0 and rep_v is not None:
                    # input = [X | V repeated over the rows of a group] (reference column order [points1 | interpolated]): V's product once
                    # per group, as a per-group additive term of the GEMM over X
                    pre_wb = W2[:, K:].contiguous()
                    W2 = W2[:, :K]
                if i == 0 and pre is not None and pre["kind"] == "group":
                    # reference column order [xyz(3) | features]: the coordinate part goes to the gather, the feature part to the GEMM
                    pre_wx = _zero_padded(W2[:, :3], _pad4(Co_true), 4)
                    W2 = W2[:, 3:]
                if i == 0 and cfg.get("xyz_last") and W2.shape[1] > 3:
                    W2 = torch.cat([W2[:, 3:], W2[:, :3]], 1)          # reference order [xyz(3) | feats] -> [feats | xyz(3)]
                if W2.shape[1] != K or Co != Co_true:
                    W2 = _zero_padded(W2, Co, K)
                    if Co != Co_true:
                        b = _zero_padded(b, Co)
                W2 = W2.contiguous()
            if fold0 and i == 0:
                gamma, beta = params[pi], params[pi + 1]
                pi += 2
                bn = bns[0]
                st = torch.empty(4, Co, dtype=torch.float32, device=dev)
                if training:
                    mom = arena.f64(16)
                    call("p2c_input_moments_f32", ptr(X0), ldx0, M, ptr(mom), stream())
                    call("p2c_bn_finalize_affine_f32", ptr(mom), M, ptr(W2), ptr(b), ptr(gamma), ptr(beta), float(bn.eps), float(bn.momentum),
                         ptr(bn.running_mean), ptr(bn.running_var), Co, ptr(st), stream())
                    if bn.nbt is not None and not _NBT_BUMPED[0]:
                        PENDING_NBT.append(bn.nbt)
                elif _eval_affine(bn, Co, infer) is not None:
                    st = _eval_affine(bn, Co, infer)
                else:           # running statistics: the layer's affine without its output
                    call("p2c_bn_finalize_f32", None, Co, M, ptr(b), ptr(gamma), ptr(beta), float(bn.eps), float(bn.momentum), 0,
                         ptr(bn.running_mean), ptr(bn.running_var), ptr(st[0]), ptr(st[1]), ptr(st[2]), ptr(st[3]), stream())
                Ys.append(None)
                Ws.append(W2)
                aff.append(st)
                fold_b0 = b
                sc, sh, in_mode = st[0], st[1], 1
                K = Co
                continue
            # the pooled last layer of SA1: when its backward takes the Y-free route (csrc/bwd_pool.hip) the forward does not store Y either
            no_y = (USE_POOL_EPI and tail == "maxpool" and i == L - 1 and in_mode == 1 and mptr_free and ldx == K
                    and L > 1 and _lib.lib().p2c_linear_fwd_pool_supported(M, Co, K, 1, cfg["ns"])
                    and (infer or (USE_POOL_ALG and training and _lib.lib().p2c_linear_bwd_pool_alg_supported(M, Co, K, cfg["ns"]))))
            Y = torch.empty(0 if no_y else M, Co, dtype=torch.float32, device=dev)
            if fold0 and i == 1:
                partials = arena.f64(STAT_SLOTS, 2, Co) if training else None
                call("p2c_linear_fwd_fold0_f32", ptr(X0), ldx0, ptr(Ws[0]), ptr(fold_b0), ptr(sc), ptr(sh), K, ptr(W2), K, ptr(b), ptr(Y), Co, M, Co,
                     ptr(partials), stream(), flops=2.0 * M * Co * K)
                Ys.append(Y)
                Ws.append(W2)
                gamma, beta = params[pi], params[pi + 1]
                pi += 2
                bn = bns[1]
                st = _eval_affine(bn, Co, infer)
                if st is None:
                    st = torch.empty(4, Co, dtype=torch.float32, device=dev)
                    call("p2c_bn_finalize_f32", ptr(partials), Co, M, ptr(b), ptr(gamma),
                         ptr(beta), float(bn.eps), float(bn.momentum), 1 if training else 0, ptr(bn.running_mean), ptr(bn.running_var),
                         ptr(st[0]), ptr(st[1]), ptr(st[2]), ptr(st[3]), stream())
                if training and bn.nbt is not None and not _NBT_BUMPED[0]:
                    PENDING_NBT.append(bn.nbt)
                aff.append(st)
                sc, sh, in_mode = st[0], st[1], 1
                X, ldx, K = Y, Co, Co
                continue
            mode, mptr, mld = in_mode, None, 0
            if (not has_bn) and in_mode == 1:
                if mask is not None:
                    mode, mptr, mld = 2, ptr(mask), mask.stride(0)
                elif seed is not None:
                    mode, mptr, mld = 3, ptr(seed), 0
            partials = arena.f64(STAT_SLOTS, 2, Co) if (has_bn and training) else None
            if rep_v is not None and i == 0:
                assert has_bn and Co == Co_true
                D2 = pre_wb.shape[1]
                Gb = torch.empty(rep_v.shape[0], Co, dtype=torch.float32, device=dev)
                call("p2c_linear_fwd_f32", ptr(rep_v), rep_v.stride(0), ptr(pre_wb), D2, None, ptr(Gb), Co, rep_v.shape[0], Co, D2, 0, None, None,
                     None, 0, 1.0, None, stream(), flops=2.0 * rep_v.shape[0] * Co * D2)
                call("p2c_linear_fwd_gbias_f32", ptr(X), ldx, ptr(W2), K, ptr(b), ptr(Gb), Co, pre["rpg"], ptr(Y), Co, M, Co, K, 0, None, None,
                     ptr(partials), stream(), flops=2.0 * M * Co * K)
            elif pre is not None and i == 0:
                assert has_bn
                Gs = torch.empty(Ms, Co, dtype=torch.float32, device=dev)
                call("p2c_linear_fwd_f32", ptr(X), ldx, ptr(W2), K, None, ptr(Gs), Co, Ms, Co, K, 0, None, None, None, 0, 1.0, None, stream(),
                     flops=2.0 * Ms * Co * K)
                if pre["kind"] == "interp":
                    call("p2c_three_interp_bias_stats_f32", ptr(Gs), Co, ptr(pre["idx"]), ptr(pre["w"]), pre["B"], pre["N"], pre["S"], Co, ptr(b),
                         ptr(Y), Co, ptr(partials), stream())
                else:
                    call("p2c_group_linear_bias_stats_f32", ptr(Gs), Co, ptr(pre["xyz"]), ptr(pre["new_xyz"]), ptr(pre["idx"]), ptr(pre_wx),
                         ptr(b), pre["B"], pre["N"], pre["S"], pre["ns"], Co, ptr(Y), Co, ptr(partials), stream())
            elif (USE_POOL_EPI and tail == "maxpool" and i == L - 1 and mode == 1 and mptr is None and (training or infer) and ldx == K
                  and _lib.lib().p2c_linear_fwd_pool_supported(M, Co, K, 1, cfg["ns"])):
                # last layer of a set-abstraction stack: the GEMM epilogue also emits the extremes the max over the 64 neighbours needs
                pool = (torch.empty(2 * cfg["G"], Co, dtype=torch.float32, device=dev), torch.empty(2 * cfg["G"], Co, dtype=torch.float32, device=dev),
                        torch.empty(2 * cfg["G"], Co, dtype=I32, device=dev))
                call("p2c_linear_fwd_pool_f32", ptr(X), ldx, ptr(W2), K, ptr(b), None if no_y else ptr(Y), Co, M, Co, K, ptr(sc), ptr(sh), ptr(partials),
                     ptr(pool[0]), ptr(pool[1]), ptr(pool[2]), stream(), flops=2.0 * M * Co * K)
            else:
                call("p2c_linear_fwd_f32", ptr(X), ldx, ptr(W2), K, ptr(b), ptr(Y), Co, M, Co, K, mode, ptr(sc), ptr(sh),
                     mptr, mld, float(dscale), ptr(partials), stream(), flops=2.0 * M * Co * K)
            Ys.append(Y)
            Ws.append(W2)
            if has_bn:
                gamma, beta = params[pi], params[pi + 1]
                pi += 2
                bn = bns[i]
                # (inference: the affine of every eval-mode BatchNorm was produced by ONE launch at the top of the forward, BNEvalStage)
                st = _eval_affine(bn, Co, infer)
                if st is None:
                    st = torch.empty(4, Co, dtype=torch.float32, device=dev)      # scale, shift, mean, invstd
                    call("p2c_bn_finalize_f32", ptr(partials), Co, M, ptr(b), ptr(gamma),
                         ptr(beta), float(bn.eps), float(bn.momentum), 1 if training else 0, ptr(bn.running_mean), ptr(bn.running_var),
                         ptr(st[0]), ptr(st[1]), ptr(st[2]), ptr(st[3]), stream())
                if training and bn.nbt is not None and not _NBT_BUMPED[0]:
                    PENDING_NBT.append(bn.nbt)
                aff.append(st)
                sc, sh, in_mode = st[0], st[1], 1
            else:
                aff.append(None)
                in_mode = 0
            X, ldx, K = Y, Co, Co
        arg = None
        if tail == "maxpool":
            G, ns = cfg["G"], cfg["ns"]
            out = torch.empty(G, K, dtype=torch.float32, device=dev)
            arg = torch.empty(G, K, dtype=I32, device=dev)
            ywin = torch.empty(G, K, dtype=torch.float32, device=dev)
            if pool is not None:
                call("p2c_pool_select_f32", ptr(pool[0]), ptr(pool[1]), ptr(pool[2]), ptr(sc), ptr(sh), G, K, ptr(out), K, ptr(arg), ptr(ywin), stream())
            else:
                call("p2c_maxpool_bnrelu_f32", ptr(Ys[-1]), K, ptr(sc), ptr(sh), G, ns, K, ptr(out), K, ptr(arg), ptr(ywin), stream())
            if cfg.get("aux_out") is not None:
                cfg["aux_out"]["pool_arg"] = arg          # (G, C) int32: the row (0..ns-1) of its group that won each pooled entry
            arg = (arg, ywin)
        elif tail == "bnrelu":
            out = torch.empty(M, K, dtype=torch.float32, device=dev)
            call("p2c_bn_relu_apply_f32", ptr(Ys[-1]), K, ptr(sc), ptr(sh), M, K, ptr(out), K, stream())
        else:
            out = Ys[-1]
        ctx.cfg = cfg
        ctx.stage_gen = stage_gen
        ctx.saved = (X0, Ys, aff, Ws, arg, params)      # params: the layer parameters (without a leading repeat vector)
        ctx.fold = (mom, fold_b0) if fold0 else None
        ctx.pre_wx = pre_wx
        ctx.rep = (rep_v, pre_wb) if rep_v is not None else None
        return out

    @staticmethod
    def backward(ctx, dout):
        cfg = ctx.cfg
        X0, Ys, aff, Ws, arg, params = ctx.saved
        if ctx.stage_gen is not None and ctx.stage_gen[0].generation != ctx.stage_gen[1]:
            raise RuntimeError("point2cyl_amd.ops.mlp_stack: backward after the staged weight operands were rewritten by a later forward "
                               "(ops.WeightStage.run()): the persistent buffers no longer hold the weights this forward used.  Run forward -> "
                               "backward in turn (no second forward of the module in between), or set P2C_STAGE_WEIGHTS=0 (per-layer weight "
                               "copies owned by each forward: any forward / backward interleaving; INTEGRATION.md, 'autograd contract')")
        rep_grad = None
        arg, ywin = arg if isinstance(arg, tuple) else (arg, None)
        # Eval mode (running statistics, e.g. fine-tuning with frozen BatchNorm, train_Point2Cyl.py:354-357): y = scale * x + shift with a
        # FIXED affine, so dY = scale * (dZ masked by the ReLU) - the batch-statistic terms q * Y + p of the train-mode backward vanish
        # (coef rows 3, 4 zeroed after every finalize), dgamma / dbeta are the same sums taken with the running mean / invstd the forward
        # saved, and the conv bias in front of the BatchNorm gets a real gradient: dbias = sum_m dY = gs * dbeta.
        evalm = not cfg["training"]
        dev = X0.device
        pre = cfg.get("pre")
        Ms = X0.shape[0]
        M = pre["rows"] if pre is not None else Ms
        L, tail = cfg["n_layers"], cfg["tail"]
        mask, seed, dscale = cfg.get("drop_mask"), cfg.get("drop_seed"), cfg.get("drop_scale", 1.0)
        dout = _f32c(dout)
        Cl = Ys[-1].shape[1]
        grads = [None] * len(params)
        n64 = 0
        for W2 in Ws:       # per layer: dW + dbias (fp32), 8 per-XCD dW copies where the long narrow layers use them (the fused backward, or the
            # split-k weight kernel from 65536 rows on), two sets of fp64 stat slots (own top-of-stack + layer below).  The step's one fill
            # clears all of it: 8 copies of the 512 x 1024 matrices of the 4096-row levels alone were 40 of its 72 MB
            copies = 9 if (M >= 65536 or (M >= 4096 and W2.shape[0] <= 256 and W2.shape[1] <= 132)) else 1
            n64 += (copies * W2.shape[0] * W2.shape[1] + W2.shape[0]) // 2 + 8 + STAT_SLOTS * 5 * (W2.shape[0] + W2.shape[1])
        arena = _ZeroArena(n64, dev)
        slots, pi = [], 0
        for i in range(L):
            has_bn = not (tail == "linear" and i == L - 1)
            slots.append((pi, has_bn))
            pi += 4 if has_bn else 2

        def standalone_stats(dZ, i):
            """coef/dgamma/dbeta of layer i from a materialised dZ (top of the stack only)."""
            p0, _ = slots[i]
            st, Co = aff[i], Ys[i].shape[1]
            coef = torch.empty(5, Co, dtype=torch.float32, device=dev)
            dgamma = torch.empty(Co, dtype=torch.float32, device=dev)
            dbeta = torch.empty(Co, dtype=torch.float32, device=dev)
            call("p2c_bn_relu_bwd_stats_f32", ptr(dZ), dZ.stride(0), ptr(Ys[i]), Co, ptr(st[0]), ptr(st[1]), ptr(st[2]), ptr(st[3]),
                 ptr(params[p0 + 2]), M, Co, ptr(dgamma), ptr(dbeta), ptr(coef), ptr(arena.f64(STAT_SLOTS, 2, Co)), stream())
            grads[p0 + 2], grads[p0 + 3] = dgamma, dbeta
            if evalm:
                coef[3:].zero_()
            return coef

        pool_ns = 0
        if tail == "maxpool":
            # dZ of the pooled layer is never materialised (grad_mode 2): winners + pooled gradient are enough
            G, pool_ns = cfg["G"], cfg["ns"]
            p0, _ = slots[L - 1]
            dZ, grad_mode = dout, 2
            coef = torch.empty(5, Cl, dtype=torch.float32, device=dev)
            dgamma = torch.empty(Cl, dtype=torch.float32, device=dev)
            dbeta = torch.empty(Cl, dtype=torch.float32, device=dev)
            call("p2c_maxpool_bn_bwd_stats_f32", ptr(dout), Cl, ptr(ywin), ptr(aff[-1]), ptr(params[p0 + 2]), G, pool_ns, Cl,
                 ptr(dgamma), ptr(dbeta), ptr(coef), ptr(arena.f64(STAT_SLOTS, 2, Cl)), stream())
            grads[p0 + 2], grads[p0 + 3] = dgamma, dbeta
            if evalm:
                coef[3:].zero_()
        elif tail == "bnrelu":
            dZ = dout
            grad_mode, coef = 1, standalone_stats(dZ, L - 1)
        else:
            dZ, grad_mode, coef = dout, 0, None
            if dZ.shape[1] != Cl:
                dZ = torch.nn.functional.pad(dZ, (0, Cl - dZ.shape[1]))
        fold = ctx.fold
        for i in range(L - 1, -1, -1):
            p0, has_bn = slots[i]
            Y, W2 = Ys[i], Ws[i]
            Co, Ci = W2.shape
            if fold is not None and i == 0:
                break                         # handled together with layer 1 below
            if ctx.rep is not None and i == 0:
                assert grad_mode == 1
                V, Wb = ctx.rep
                G_, D2 = V.shape[0], Wb.shape[1]
                dGb = torch.empty(G_, Co, dtype=torch.float32, device=dev)
                call("p2c_group_colsum_bn_f32", ptr(dZ), dZ.stride(0), ptr(Y), Co, ptr(coef), G_, pre["rpg"], Co, ptr(dGb), Co, stream())
                dWb = arena.f32(Co, D2)
                call("p2c_linear_bwd_weight_f32", ptr(dGb), Co, None, 0, 0, None, ptr(V), V.stride(0), 0, None, None, None, 0, 1.0, ptr(dWb), D2, 0,
                     None, G_, Co, D2, None, 0, stream(), flops=2.0 * G_ * Co * D2)
                dV = torch.empty(G_, D2, dtype=torch.float32, device=dev)
                call("p2c_linear_bwd_data_f32", ptr(dGb), Co, None, 0, 0, None, ptr(Wb), D2, ptr(dV), D2, G_, Co, D2, None, 0, 1.0, None, 0,
                     None, None, None, 0, stream(), flops=2.0 * G_ * Co * D2)
                dWa = arena.f32(Co, Ci)
                call("p2c_linear_bwd_weight_f32", ptr(dZ), dZ.stride(0), ptr(Y), Co, 1, ptr(coef), ptr(X0), X0.stride(0), 0, None, None, None, 0, 1.0,
                     ptr(dWa), Ci, 0, None, M, Co, Ci, None, 0, stream(), flops=2.0 * M * Co * Ci)
                Wp = params[p0]
                grads[p0] = torch.cat([dWa[:, :Ci], dWb], 1).reshape(Wp.shape)
                grads[p0 + 1] = arena.f32(Co)
                rep_grad = dV
                dX = None
                if ctx.needs_input_grad[1]:
                    dX = torch.empty(M, Ci, dtype=torch.float32, device=dev)
                    call("p2c_linear_bwd_data_f32", ptr(dZ), dZ.stride(0), ptr(Y), Co, 1, ptr(coef), ptr(W2), Ci, ptr(dX), Ci, M, Co, Ci, None, 0,
                         1.0, None, 0, None, None, None, 0, stream(), flops=2.0 * M * Co * Ci)
                dZ = dX
                break
            if pre is not None and i == 0:
                # dY0 -> sparse rows (CSR gather with the ReLU+BN backward rebuilt per element), then two small GEMMs
                assert grad_mode == 1
                offsets, rows_, ws_ = pre["csr"]
                dG = torch.empty(Ms, Co, dtype=torch.float32, device=dev) if pre["kind"] == "interp" else STEP_ARENA.take((Ms * Co + 1) // 2, dev).view(torch.float32)[:Ms * Co].view(Ms, Co)   # zeroed: the grouped path accumulates with atomics
                dwx = None
                if pre["kind"] == "interp":
                    call("p2c_csr_gather_bn_f32", ptr(dZ), dZ.stride(0), ptr(Y), Co, ptr(coef), ptr(offsets), ptr(rows_), ptr(ws_), pre["B"],
                         pre["N"] * 3, pre["N"], pre["S"], Co, ptr(dG), Co, stream())
                else:
                    dwx = arena.f64(STAT_SLOTS, 3, Co)
                    call("p2c_group_linear_bwd_f32", ptr(dZ), dZ.stride(0), ptr(Y), Co, ptr(coef), ptr(offsets), ptr(rows_), ptr(pre["xyz"]),
                         ptr(pre["new_xyz"]), pre["B"], pre["N"], pre["S"], pre["ns"], Co, ptr(dG), Co, ptr(dwx), stream())
                dW = arena.f32(Co, Ci)
                call("p2c_linear_bwd_weight_f32", ptr(dG), Co, None, 0, 0, None, ptr(X0), X0.stride(0), 0, None, None, None, 0, 1.0, ptr(dW), Ci, 0,
                     None, Ms, Co, Ci, None, 0, stream(), flops=2.0 * Ms * Co * Ci)
                Wp = params[p0]
                if dwx is not None:
                    # [coordinate part (slot sums) | feature part] in the parameter's own layout by one launch (was: reduction, transpose+cast, cat)
                    gW = torch.empty(Wp.shape, dtype=torch.float32, device=dev)
                    call("p2c_group_weight_grad_f32", ptr(dwx), Co, ptr(dW), Ci, Wp.shape[0], Wp.numel() // Wp.shape[0] - 3, ptr(gW), stream())
                    grads[p0] = gW
                else:
                    grads[p0] = dW[:Wp.shape[0], :Wp.numel() // Wp.shape[0]].reshape(Wp.shape)
                grads[p0 + 1] = arena.f32(Co)[:Wp.shape[0]]          # bias in front of a train-mode BatchNorm: exactly zero
                dZ = None
                if ctx.needs_input_grad[1]:
                    dZ = torch.empty(Ms, Ci, dtype=torch.float32, device=dev)
                    call("p2c_linear_bwd_data_f32", ptr(dG), Co, None, 0, 0, None, ptr(W2), Ci, ptr(dZ), Ci, Ms, Co, Ci, None, 0, 1.0, None, 0,
                         None, None, None, 0, stream(), flops=2.0 * Ms * Co * Ci)
                break
            if fold is not None and i == 1:
                # fused backward of layer 1 with its X operand rebuilt from the stack input; no dX, 5 sums per column for layer 0
                mom, b0 = fold
                q0, _ = slots[0]
                W0p, st0, C0 = Ws[0], aff[0], Ws[0].shape[0]
                dW8 = arena.f32(8, Co, Ci)
                part5 = arena.f64(STAT_SLOTS, 5, C0)
                call("p2c_linear_bwd_fused_fold0_f32", ptr(dZ), dZ.stride(0), ptr(Y), Co, ptr(coef), ptr(X0), X0.stride(0), ptr(W0p), ptr(b0),
                     ptr(st0), ptr(W2), Ci, ptr(dW8), Ci, Co * Ci, ptr(part5), M, Co, C0, stream(), flops=4.0 * M * Co * Ci,
                     nbytes=4.0 * M * (2 * Co + 4))
                Wp = params[p0]
                W0param = params[q0]
                c0_in = W0param.numel() // W0param.shape[0]
                # ONE launch: dgamma0 / dbeta0 / dW0 (in the parameter's own (C0, 3) layout) and, by extra workgroups, the sum of this layer's
                # per-XCD dW copies (was: a torch reduction, the finalize, and a slicing copy)
                dWsum = torch.empty(Co, Ci, dtype=torch.float32, device=dev)
                dW0 = torch.empty(C0, c0_in, dtype=torch.float32, device=dev)
                dgamma0 = torch.empty(C0, dtype=torch.float32, device=dev)
                dbeta0 = torch.empty(C0, dtype=torch.float32, device=dev)
                call("p2c_fold0_bwd_finalize_sum_f32", ptr(part5), ptr(mom), M, ptr(W0p), ptr(b0), ptr(st0), ptr(params[q0 + 2]), C0, ptr(dgamma0),
                     ptr(dbeta0), ptr(dW0), c0_in, ptr(dW8), Co * Ci, 8, ptr(dWsum), Co * Ci, stream())
                grads[p0] = dWsum[:Wp.shape[0], :Wp.numel() // Wp.shape[0]].reshape(Wp.shape)
                grads[p0 + 1] = arena.f32(Co)[:Wp.shape[0]]
                grads[q0] = dW0.reshape(W0param.shape)
                grads[q0 + 1] = arena.f32(C0)
                grads[q0 + 2], grads[q0 + 3] = dgamma0, dbeta0
                continue
            if i == 0:
                Xin, ldxin, in_mode, sc, sh = X0, X0.stride(0), 0, None, None
            else:
                Xin, ldxin, in_mode, sc, sh = Ys[i - 1], Ys[i - 1].shape[1], 1, aff[i - 1][0], aff[i - 1][1]
            mode, mptr, mld, omld = in_mode, None, 0, 0
            if (not has_bn) and in_mode == 1:
                if mask is not None:
                    mode, mptr, mld, omld = 2, ptr(mask), mask.stride(0), mask.stride(0)
                elif seed is not None:
                    mode, mptr, mld, omld = 3, ptr(seed), 0, -1
            # one zero-fill for all weight/bias gradients of the stack (they are accumulated with atomics)
            dW = arena.f32(Co, Ci)
            # a conv bias in front of a train-mode BatchNorm has an exactly zero gradient (the batch mean absorbs it)
            db = arena.f32(Co)
            Wp = params[p0]
            co_t, ci_t = Wp.shape[0], Wp.numel() // Wp.shape[0]
            def _to_param_layout(g):
                g = g[:co_t, :ci_t]
                if i == 0 and cfg.get("xyz_last") and ci_t > 3:
                    g = torch.cat([g[:, ci_t - 3:], g[:, :ci_t - 3]], 1)     # back to the reference's [xyz | feats] input order
                return g.reshape(Wp.shape)

            grads[p0 + 1] = db[:co_t]
            dW_final = dW                     # replaced by the sum of the per-XCD copies where those are used
            dW_copies = None                  # (the 8 copies, summed by the finalize launch below - or a launch of its own where there is none)
            need_dx = i > 0 or ctx.needs_input_grad[1]
            stats_below = i > 0                       # the layer below has a BatchNorm whose backward sums we produce here
            L_ = _lib.lib()
            fused_kind = L_.p2c_linear_bwd_fused_supported(Co, Ci, mode) if (USE_FUSED_BWD and mode <= 1 and M >= 4096) else 0
            if grad_mode == 2 and (pool_ns < 32 or pool_ns % 16):
                fused_kind = 0           # the pooled variant wants a row tile to span at most two groups
            if fused_kind == 2 and not (i == 0 and cfg.get("xyz_last")):
                fused_kind = 0           # kind 2 yields no dX for the 4 trailing input columns: fine for [feats | xyz | pad] only
            if fused_kind == 3 and (not need_dx or M < 8192 or not USE_FUSED256):
                fused_kind = 0           # the two-pass form of a 256-wide layer accumulates dX across its passes: long layers with a dX only
            narrow = (USE_NARROW_BWD and grad_mode == 0 and need_dx and stats_below and mode in (1, 3)
                      and L_.p2c_linear_bwd_narrow_supported(M, Co, Ci, mode))
            pool_alg = (USE_POOL_ALG and grad_mode == 2 and i == L - 1 and need_dx and stats_below and mode == 1 and ldxin == Ci
                        and ywin is not None and L_.p2c_linear_bwd_pool_alg_supported(M, Co, Ci, pool_ns))
            if Y.shape[0] == 0 and not pool_alg:
                raise RuntimeError("the pooled layer's forward did not store its pre-BatchNorm output (P2C_POOL_ALG route) but its backward "
                                   "was asked to take the generic kernel; do not toggle ops.USE_POOL_ALG between a forward and its backward")
            if pool_alg:
                # SA1's last layer: dY = gs*G + q*Y + p with Y = A W^T + b linear in the staged input, so neither Y nor a dense dY exists
                # (csrc/bwd_pool.hip): dX + the sums of the layer below from one read of X, dW assembled from Gs^T A, A^T A and 1^T A
                dX = torch.empty(M, Ci, dtype=torch.float32, device=dev)
                part = arena.f64(STAT_SLOTS, 2, Ci)
                acc = torch.empty(L_.p2c_linear_bwd_pool_alg_ws_bytes(Co, Ci) // 4 + 4, dtype=torch.float32, device=dev)      # scratch, no zero-fill
                dW_final = torch.empty(Co, Ci, dtype=torch.float32, device=dev)
                call("p2c_linear_bwd_pool_alg_f32", ptr(dZ), dZ.stride(0), ptr(ywin), ptr(arg), ptr(coef), ptr(Xin), ldxin, ptr(sc), ptr(sh),
                     ptr(W2), Ci, ptr(params[p0 + 1]), ptr(dX), Ci, ptr(aff[i - 1]), ptr(part), ptr(acc), ptr(dW_final), Ci, M, Co, Ci, pool_ns,
                     stream(), flops=4.0 * M * Co * Ci, nbytes=4.0 * M * 2 * Ci)
            elif narrow:
                # a few outputs on many rows (the heads): dW, dbias, dX and the sums of the BatchNorm below from ONE read of dZ and the input
                dX = torch.empty(M, Ci, dtype=torch.float32, device=dev)
                part = arena.f64(STAT_SLOTS, 2, Ci)
                dW8 = arena.f32(8, Co, Ci)
                call("p2c_linear_bwd_narrow_f32", ptr(dZ), dZ.stride(0), ptr(Xin), ldxin, ptr(aff[i - 1]), mptr if mode == 3 else None, float(dscale),
                     ptr(W2), Ci, ptr(dX), Ci, ptr(dW8), Ci, Co * Ci, ptr(db), ptr(part), M, Co, Ci, stream(),
                     flops=4.0 * M * Co * Ci, nbytes=4.0 * M * (Co + 2 * Ci))
                dW_copies = dW8
            elif fused_kind:
                dX = torch.empty(M, Ci, dtype=torch.float32, device=dev) if need_dx else None
                part = arena.f64(STAT_SLOTS, 2, Ci) if stats_below else None
                dW8 = arena.f32(8, Co, Ci)     # one copy per XCD, summed below
                call("p2c_linear_bwd_fused_f32", ptr(dZ), dZ.stride(0), ptr(Y), Co, grad_mode, ptr(coef), ptr(arg) if grad_mode == 2 else None,
                     pool_ns, ptr(Xin), ldxin, mode, ptr(sc), ptr(sh), ptr(W2), Ci, ptr(dX), Ci, ptr(dW8), Ci, Co * Ci,
                     ptr(db) if grad_mode == 0 else None, ptr(aff[i - 1]) if stats_below else None, ptr(part), M, Co, Ci, stream(),
                     flops=(4.0 if need_dx else 2.0) * M * Co * Ci,
                     nbytes=4.0 * M * ((1 if grad_mode == 2 else 2) * Co + (2 if need_dx else 1) * Ci))   # dZ (unless pooled), Y, X read once; dX written once
                dW_copies = dW8
            elif (USE_DUAL_BWD and need_dx and M <= 8192 and Co > 64 and Ci > 64 and grad_mode in (1, 2) and mode <= 1 and mptr is None):
                # a few thousand rows: neither backward GEMM fills the chip -> both in one launch, side by side (csrc/gemm.hip)
                dX = torch.empty(M, Ci, dtype=torch.float32, device=dev)
                part = arena.f64(STAT_SLOTS, 2, Ci) if stats_below else None
                call("p2c_linear_bwd_both_f32", ptr(dZ), dZ.stride(0), ptr(Y), Co, grad_mode, ptr(coef), ptr(arg) if grad_mode == 2 else None,
                     pool_ns, ptr(Xin), ldxin, mode, ptr(sc), ptr(sh), ptr(W2), Ci, ptr(dX), Ci, ptr(Ys[i - 1]) if stats_below else None, Ci,
                     ptr(aff[i - 1]) if stats_below else None, ptr(part), ptr(dW), Ci, M, Co, Ci, stream(), flops=4.0 * M * Co * Ci)
            else:
                use_slots = M >= 65536       # many split-k workgroups: spread the atomics over 8 copies of dW
                dWs = arena.f32(8, Co, Ci) if use_slots else dW
                call("p2c_linear_bwd_weight_f32", ptr(dZ), dZ.stride(0), ptr(Y), Co, grad_mode, ptr(coef), ptr(Xin), ldxin, mode, ptr(sc),
                     ptr(sh), mptr, mld, float(dscale), ptr(dWs), Ci, Co * Ci if use_slots else 0,
                     ptr(db) if grad_mode == 0 else None, M, Co, Ci, ptr(arg) if grad_mode == 2 else None, pool_ns, stream(),
                     flops=2.0 * M * Co * Ci)
                if use_slots:
                    dW_copies = dWs
                dX = part = None
                if need_dx:
                    dX = torch.empty(M, Ci, dtype=torch.float32, device=dev)
                    part = arena.f64(STAT_SLOTS, 2, Ci) if stats_below else None
                    call("p2c_linear_bwd_data_f32", ptr(dZ), dZ.stride(0), ptr(Y), Co, grad_mode, ptr(coef), ptr(W2), Ci, ptr(dX), Ci, M, Co, Ci,
                         mptr, omld, float(dscale),
                         ptr(Ys[i - 1]) if stats_below else None, Ci, ptr(aff[i - 1]) if stats_below else None, ptr(part),
                         ptr(arg) if grad_mode == 2 else None, pool_ns, stream(), flops=2.0 * M * Co * Ci)
            if dW_copies is not None:
                dW_final = torch.empty(Co, Ci, dtype=torch.float32, device=dev)
            if need_dx:
                dZ, grad_mode = dX, 1
                if stats_below:
                    q0, _ = slots[i - 1]
                    coef = torch.empty(5, Ci, dtype=torch.float32, device=dev)
                    dgamma = torch.empty(Ci, dtype=torch.float32, device=dev)
                    dbeta = torch.empty(Ci, dtype=torch.float32, device=dev)
                    if dW_copies is not None:      # the copies are summed by extra workgroups of the same launch
                        call("p2c_bn_bwd_finalize_sum_f32", ptr(part), Ci, M, ptr(aff[i - 1]), ptr(params[q0 + 2]), ptr(dgamma), ptr(dbeta),
                             ptr(coef), ptr(dW_copies), Co * Ci, dW_copies.shape[0], ptr(dW_final), Co * Ci, stream())
                        dW_copies = None
                    else:
                        call("p2c_bn_bwd_finalize_f32", ptr(part), Ci, M, ptr(aff[i - 1]), ptr(params[q0 + 2]), ptr(dgamma),
                             ptr(dbeta), ptr(coef), stream())
                    grads[q0 + 2], grads[q0 + 3] = dgamma, dbeta
                    if evalm:
                        coef[3:].zero_()
            if dW_copies is not None:
                call("p2c_sum_copies_f32", ptr(dW_copies), Co * Ci, dW_copies.shape[0], ptr(dW_final), Co * Ci, stream())
            grads[p0] = _to_param_layout(dW_final)       # after the kernels are enqueued (the permuted layout is a copy)
        if evalm:
            for i in range(L):
                p0, has_bn = slots[i]
                if has_bn and grads[p0 + 3] is not None:
                    st, gamma = aff[i], params[p0 + 2]
                    nb = params[p0 + 1].shape[0]
                    grads[p0 + 1] = (gamma.detach().reshape(-1)[:nb] * st[3][:nb] * grads[p0 + 3][:nb]).reshape(params[p0 + 1].shape)
        dX0 = None
        if ctx.needs_input_grad[1]:
            dX0 = dZ
            if dX0.shape[1] > X0.shape[1]:
                dX0 = dX0[:, : X0.shape[1]]
            elif dX0.shape[1] < X0.shape[1]:
                dX0 = torch.nn.functional.pad(dX0, (0, X0.shape[1] - dX0.shape[1]))
        if ctx.rep is not None:
            return (None, dX0, rep_grad) + tuple(grads)
        return (None, dX0) + tuple(grads)


def mlp_stack(X0, in_channels, layers, tail, training, G=None, ns=None, drop_mask=None, drop_scale=1.0, drop_seed=None,
              keep_padding=False, xyz_last=False, pre=None, staged=None, aux_out=None):
    """layers: list of dicts {W, b, gamma, beta, bn: BNState} (gamma/beta/bn None for a BN-less last layer).
    aux_out: optional dict that receives `pool_arg` (the max-pool's winner rows) for inspection."""
    params, bns = [], []
    for ly in layers:
        params += [ly["W"], ly["b"]]
        if ly.get("gamma") is not None:
            params += [ly["gamma"], ly["beta"]]
        bns.append(ly.get("bn"))
    cfg = dict(in_channels=in_channels, n_layers=len(layers), tail=tail, training=training, bns=bns, G=G, ns=ns,
               drop_mask=drop_mask, drop_scale=drop_scale, drop_seed=drop_seed, xyz_last=xyz_last, pre=pre, staged=staged, aux_out=aux_out)
    if pre is not None and pre["kind"] == "repeat":
        params = [pre.pop("V")] + params
    # (decided HERE: inside Function.forward grad mode is off and ctx.needs_input_grad is True for every Parameter even under no_grad)
    cfg["wants_grad"] = torch.is_grad_enabled() and (X0.requires_grad or any(torch.is_tensor(p_) and p_.requires_grad for p_ in params))
    out = _MLPStack.apply(cfg, X0, *params)
    if not _DEFER_NBT[0]:
        flush_nbt()
    co_last = layers[-1]["W"].shape[0]
    if tail == "linear" and out.shape[1] != co_last and not keep_padding:
        out = out[:, :co_last]          # the kernels work on 4-padded channel counts
    return out


# ------------------------------------------------------------------------------------------ weight staging (one launch per step)
def copy_flat_batch(dsts, srcs):
    """dst_i.copy_(src_i) for lists of contiguous device tensors of equal shape and dtype, in ONE launch (p2c_copy_flat_batch: the descriptors
    ride in the kernel arguments, so the call can be captured without a device table).  Tensors whose byte size is not a multiple of 4
    fall back to torch's copy."""
    import ctypes
    fast_d, fast_s = [], []
    for d, s_ in zip(dsts, srcs):
        if d.shape != s_.shape or d.dtype != s_.dtype:
            raise ValueError("copy_flat_batch: %s %s <- %s %s" % (tuple(d.shape), d.dtype, tuple(s_.shape), s_.dtype))
        if d.is_contiguous() and s_.is_contiguous() and (d.numel() * d.element_size()) % 4 == 0 and d.data_ptr() % 4 == 0 and s_.data_ptr() % 4 == 0:
            if d.numel():
                fast_d.append(d); fast_s.append(s_)
        else:
            d.copy_(s_)
    n = len(fast_d)
    if n:
        _lib.require_device(*fast_d, *fast_s)
        sp = (ctypes.c_void_p * n)(*[t.data_ptr() for t in fast_s])
        dp = (ctypes.c_void_p * n)(*[t.data_ptr() for t in fast_d])
        nb = (ctypes.c_longlong * n)(*[t.numel() * t.element_size() for t in fast_d])
        call("p2c_copy_flat_batch", ctypes.cast(sp, ctypes.c_void_p), ctypes.cast(dp, ctypes.c_void_p), ctypes.cast(nb, ctypes.c_void_p), n, stream(),
             nbytes=2.0 * sum(nb))


USE_STAGED_WEIGHTS = os.environ.get("P2C_STAGE_WEIGHTS", "1") != "0"


class WeightStage:
    """Every operand a step derives from the PARAMETERS alone - zero-padded weights (3 -> 4, 259 -> 260 input channels, 19 -> 20 head
    outputs), the grouped layers' [xyz | features] -> [features | xyz] column order, the column blocks of a first-layer weight that
    multiply different inputs - prepared by ONE launch (p2c_copy2d_batch_f32) into persistent buffers instead of a dozen torch copy / cat
    launches spread over the forward pass.  entries: (key, dst_shape, [(src, src col0, ncols, dst row0, dst col0), ...]) where src is a
    CALLABLE that returns the live source tensor (`lambda: conv.weight`): it is resolved on every run(), so a parameter that got new
    storage (`p.data = ...`, `.cpu()` / `.cuda()`, load_state_dict(assign=True)) or was replaced is followed - a tensor captured once
    would keep an alias of the old storage alive and the staged operands would silently go stale.  Padding stays zero from
    the allocation; the device table is rebuilt when an address changed.  `generation` counts the runs (ops._MLPStack's backward
    checks that the staged operands it reads are still the ones its forward used)."""

    def __init__(self, entries, device, counters=()):
        """counters: [(resolver of a device int64 scalar tensor, increment), ...] advanced by the same launch when run(bump=True): the BatchNorm
        num_batches_tracked (+1 each) and the dropout hash seed (+ its stride) - otherwise a multi-tensor add and a scalar add launch per forward."""
        self.entries, self.device = entries, device
        self.counters = list(counters)
        self.ctables = {}         # one device table per live counter set, kept for the stage's lifetime: captured graphs hold their addresses
        self.bufs = {key: torch.zeros(*shape, dtype=torch.float32, device=device) for key, shape, _ in entries}
        for b in self.bufs.values():
            b._p2c_stage = self          # (ops._MLPStack finds the stage of a staged operand through this tag)
        self.table, self.n, self.src_ptrs = None, 0, None
        self.generation = 0

    def _build(self):
        import struct
        rows = []
        for key, shape, parts in self.entries:
            dst = self.bufs[key]
            ldd = dst.shape[-1]
            for get, c0, nc, r0, d0 in parts:
                src = get()
                s2 = src.detach().reshape(src.shape[0], -1) if src.dim() > 1 else src.detach().reshape(1, -1)
                assert s2.is_contiguous() and s2.dtype == torch.float32
                rows.append(struct.pack("<QQiiii", s2.data_ptr() + 4 * c0, dst.data_ptr() + 4 * (r0 * ldd + d0), s2.shape[0], nc, s2.shape[1], ldd))
        raw = b"".join(rows)
        self.table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(self.device)
        self.n = len(rows)
        self.src_ptrs = self._ptrs()

    def _ptrs(self):
        return tuple(get().data_ptr() for _, _, parts in self.entries for get, *_ in parts)

    def _counter_table(self, which):
        import struct
        live = [(get(), inc) for get, inc in which]
        live = [(t, inc) for t, inc in live if t is not None]
        key = tuple((t.data_ptr(), inc) for t, inc in live)
        # The live set depends on the mode (train: every num_batches_tracked + the dropout counter; eval: the dropout counter only).  A HIP
        # graph captured in one mode has the ADDRESS of its table baked into the copy2d_batch_inc node, and graphs of both modes stay
        # cached (autograph keeps one per model.training): a table is therefore never dropped or rewritten once built (a single
        # cached table was freed on the mode switch and the earlier graph's replay wrote through it).
        if key not in self.ctables:
            for t, _ in live:
                assert t.dtype == torch.int64 and t.is_cuda and t.numel() == 1
            raw = b"".join(struct.pack("<Qq", t.data_ptr(), inc) for t, inc in live)
            self.ctables[key] = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(self.device) if live else None
        return self.ctables[key], len(live)

    def run(self, bump=None):
        """bump: None (copies only) or the subset of self.counters to advance in the same launch."""
        if self.table is None or self.src_ptrs != self._ptrs():
            self._build()
        self.generation += 1
        if bump:
            ctab, nc = self._counter_table(bump)
            if nc:
                call("p2c_copy2d_batch_inc_f32", ptr(self.table), self.n, ptr(ctab), nc, stream())
                return True
        call("p2c_copy2d_batch_f32", ptr(self.table), self.n, stream())
        return False

    def __getitem__(self, key):
        return self.bufs[key]


class BNEvalStage:
    """Inference: the (scale, shift, mean, invstd) of every eval-mode BatchNorm of the model, from ONE launch at the top of the forward
    (p2c_bn_eval_affine_batch_f32) into persistent buffers, instead of one p2c_bn_finalize_f32 launch per layer between the GEMMs (17
    graph nodes, 0.125 ms of a 1.17 ms forward).  The kernel reads the live parameters and running statistics on every run - a captured
    forward stays right when an optimizer or load_state_dict changes them in place; modules whose tensors got NEW storage are followed by
    rebuilding the device table (same rule as WeightStage).  Tables are never freed once built: captured graphs hold their addresses."""

    def __init__(self, mods, device):
        self.mods, self.device = list(mods), device
        self.st = [torch.empty(4, m.num_features, dtype=torch.float32, device=device) for m in self.mods]
        self.tables = {}
        self.table, self.key = None, None

    def _ptrs(self):
        return tuple(t.data_ptr() for m in self.mods for t in (m.weight, m.bias, m.running_mean, m.running_var)) + tuple(float(m.eps) for m in self.mods)

    def run(self):
        """-> {running_mean.data_ptr(): affine (4, C)} for ops._MLPStack's eval-mode layers."""
        import struct
        key = self._ptrs()
        if key != self.key:
            if key not in self.tables:
                rows = []
                for m, st in zip(self.mods, self.st):
                    for t in (m.weight, m.bias, m.running_mean, m.running_var):
                        assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.numel() == m.num_features
                    rows.append(struct.pack("<QQQQQif", m.weight.data_ptr(), m.bias.data_ptr(), m.running_mean.data_ptr(),
                                            m.running_var.data_ptr(), st.data_ptr(), m.num_features, float(m.eps)))
                self.tables[key] = torch.frombuffer(bytearray(b"".join(rows)), dtype=torch.uint8).to(self.device)
            self.table, self.key = self.tables[key], key
            self.by_ptr = {m.running_mean.data_ptr(): st for m, st in zip(self.mods, self.st)}
        call("p2c_bn_eval_affine_batch_f32", ptr(self.table), len(self.mods), stream())
        return self.by_ptr


class _HeadParams(torch.autograd.Function):
    """The per-point heads share one GEMM: their weights stacked to (sum o_i, 128).  Forward returns the rows of the staged (zero-padded)
    buffer the batched copy has filled - no launch; backward hands every head its rows of the stacked gradient - views, no launch."""

    @staticmethod
    def forward(ctx, Wbuf, bbuf, n_heads, *wb):
        ctx.sizes = [w.shape[0] for w in wb[:n_heads]]
        ctx.shapes = [tuple(t.shape) for t in wb]
        tot = sum(ctx.sizes)
        return Wbuf[:tot], bbuf[:tot]

    @staticmethod
    def backward(ctx, gW, gb):
        outs_w, outs_b, o = [], [], 0
        n = len(ctx.sizes)
        for i, sz in enumerate(ctx.sizes):
            outs_w.append(None if gW is None else gW[o:o + sz].reshape(ctx.shapes[i]))
            outs_b.append(None if gb is None else gb[o:o + sz].reshape(ctx.shapes[n + i]))
            o += sz
        return (None, None, None) + tuple(outs_w) + tuple(outs_b)


class _SkipInterpCat(torch.autograd.Function):
    """[skip features | 3-NN interpolated features | zero pad] of a feature-propagation level (pointnet_util.py:308-312) as ONE buffer: the
    interpolation kernel writes its column block in place (ldo), the skip features are one strided copy - instead of interpolating into a
    tensor of its own and concatenating (a 25 MB cat at FP2)."""

    @staticmethod
    def forward(ctx, feats1, feats2, idx, w, csr, width):
        B, S, C2 = feats2.shape
        N = idx.shape[1]
        C1 = feats1.shape[1]
        feats2 = _f32c(feats2)
        out = torch.empty(B * N, width, dtype=torch.float32, device=feats2.device)
        if feats1.dtype == torch.float32 and feats1.stride(1) == 1:
            # skip block, interpolated block and the zero pad of every row by ONE kernel
            call("p2c_three_interp_skip_f32", ptr(feats2), C2, ptr(idx), ptr(w), B, N, S, C2, ptr(feats1), feats1.stride(0), C1, ptr(out), width, width,
                 stream())
        else:
            out[:, :C1].copy_(feats1)
            if width > C1 + C2:
                out[:, C1 + C2:].zero_()
            call("p2c_three_interp_f32", ptr(feats2), C2, ptr(idx), ptr(w), B, N, S, C2, out.data_ptr() + 4 * C1, width, stream())
        ctx.save_for_backward(idx, w)
        ctx.csr, ctx.dims = csr, (B, N, S, C1, C2)
        return out

    @staticmethod
    def backward(ctx, dout):
        idx, w = ctx.saved_tensors
        B, N, S, C1, C2 = ctx.dims
        if dout.stride(1) != 1:
            dout = dout.contiguous()
        d1 = dout[:, :C1] if ctx.needs_input_grad[0] else None
        d2 = None
        if ctx.needs_input_grad[1]:
            dpart = dout[:, C1:C1 + C2]
            d2 = torch.empty(B, S, C2, dtype=torch.float32, device=dout.device)
            if USE_CSR_BWD and C2 <= 256:
                offsets, rows, ws = ctx.csr if ctx.csr is not None else build_csr(idx, S, w, 3)
                call("p2c_csr_gather_f32", ptr(dpart), dout.stride(0), 0, ptr(offsets), ptr(rows), ptr(ws), B, N * 3, N, S, C2, ptr(d2), C2, stream())
            else:
                d2.zero_()
                call("p2c_three_interp_bwd_f32", ptr(dpart), dout.stride(0), ptr(idx), ptr(w), B, N, S, C2, ptr(d2), C2, stream())
        return d1, d2, None, None, None, None


def skip_interp_cat(feats1, feats2, idx, w, csr=None):
    """feats1 (B*N, C1) skip features, feats2 (B,S,C2) -> (B*N, pad4(C1 + C2)) = [feats1 | interp(feats2) | 0]."""
    return _SkipInterpCat.apply(feats1, feats2, idx, w, csr, _pad4(feats1.shape[1] + feats2.shape[2]))


# ------------------------------------------------------------------------------------------ fitting
class _ExtrusionAxis(torch.autograd.Function):
    @staticmethod
    def forward(ctx, X, Wb, Wc, bb_gt, inst_gt, normalize, axis64):
        X, Wb, Wc = _f32c(X), _f32c(Wb), _f32c(Wc)
        B, N, K = Wb.shape
        axis = torch.empty(B, K, 3, dtype=torch.float32, device=X.device)
        eig = torch.empty(B, K, 12, dtype=torch.float32, device=X.device)
        if normalize:
            bb_gt = bb_gt.to(torch.int64).contiguous()
            inst_gt = inst_gt.to(torch.int64).contiguous()
        call("p2c_extrusion_axis_f32", ptr(X), ptr(Wb), ptr(Wc), ptr(bb_gt) if normalize else None,
             ptr(inst_gt) if normalize else None, 1 if normalize else 0, B, N, K, ptr(axis), ptr(eig), ptr(axis64), stream())
        ctx.save_for_backward(X, Wb, Wc, axis, eig)
        return axis

    @staticmethod
    def backward(ctx, daxis):
        X, Wb, Wc, axis, eig = ctx.saved_tensors
        B, N, K = Wb.shape
        daxis = _f32c(daxis)
        dX, dWb, dWc = torch.empty_like(X), torch.empty_like(Wb), torch.empty_like(Wc)
        call("p2c_extrusion_axis_bwd_f32", ptr(daxis), ptr(axis), ptr(eig), ptr(X), ptr(Wb), ptr(Wc), B, N, K, ptr(dX), ptr(dWb),
             ptr(dWc), stream())
        return dX, dWb, dWc, None, None, None, None


def extrusion_axis(X, Wb, Wc, bb_gt=None, inst_gt=None, normalize=False, axis64=None):
    """axis64: optional (B,K,3) float64 device tensor that receives the unit eigenvector before its rounding to fp32 (no gradient)."""
    _lib.require_device(X, Wb, Wc, axis64)
    if axis64 is not None and (axis64.dtype != torch.float64 or not axis64.is_contiguous() or tuple(axis64.shape) != (Wb.shape[0], Wb.shape[2], 3)):
        raise ValueError("extrusion_axis: axis64 must be a contiguous float64 (B,K,3) tensor")
    return _ExtrusionAxis.apply(X, Wb, Wc, bb_gt, inst_gt, bool(normalize), axis64)


class _ExtrusionCenters(torch.autograd.Function):
    @staticmethod
    def forward(ctx, W, P):
        W, P = _f32c(W), _f32c(P)
        B, N, K = W.shape
        out = torch.empty(B, K, 3, dtype=torch.float32, device=W.device)
        call("p2c_extrusion_centers_f32", ptr(W), ptr(P), B, N, K, ptr(out), stream())
        ctx.save_for_backward(P)
        ctx.dims = (B, N, K)
        return out

    @staticmethod
    def backward(ctx, dC):
        (P,) = ctx.saved_tensors
        B, N, K = ctx.dims
        dW = torch.empty(B, N, K, dtype=torch.float32, device=P.device)
        call("p2c_extrusion_centers_bwd_f32", ptr(_f32c(dC)), ptr(P), B, N, K, ptr(dW), stream())
        return dW, None


def extrusion_centers(W, P):
    _lib.require_device(W, P)
    return _ExtrusionCenters.apply(W, P)


def segment_centroids(P, label, K):
    """eval.py:409-436 from integer labels (-1 = no segment) -> centroids (B,K,3), found (B,K)."""
    _lib.require_device(P, label)
    P = _f32c(P)
    B, N, _ = P.shape
    label = label.to(torch.int64).contiguous()
    cen = torch.empty(B, K, 3, dtype=torch.float32, device=P.device)
    found = torch.empty(B, K, dtype=torch.float32, device=P.device)
    call("p2c_segment_centroids_f32", ptr(P), ptr(label), B, N, K, ptr(cen), ptr(found), stream())
    return cen, found


def extrusion_extents(P, seg, bb, axes, centers, rand_idx):
    """data_utils.py:1650-1730.  rand_idx (B,K,S) int64 -> extents (K,B,2), found (B,K)."""
    _lib.require_device(P, seg, bb, axes, centers, rand_idx)
    P, axes, centers = _f32c(P), _f32c(axes), _f32c(centers)
    B, N, _ = P.shape
    K = axes.shape[1]
    S = rand_idx.shape[2]
    seg, bb, rand_idx = seg.to(torch.int64).contiguous(), bb.to(torch.int64).contiguous(), rand_idx.to(torch.int64).contiguous()
    ext = torch.empty(K, B, 2, dtype=torch.float32, device=P.device)
    found = torch.empty(B, K, dtype=torch.float32, device=P.device)
    ws = torch.empty(_lib.lib().p2c_extents_ws_bytes(B, K) // 4 + 4, dtype=torch.float32, device=P.device)
    call("p2c_extrusion_extents_f32", ptr(P), ptr(seg), ptr(bb), ptr(axes), ptr(centers), ptr(rand_idx), B, N, K, S, ptr(ext),
         ptr(found), ptr(ws), stream())
    return ext, found


def fit_fused_supported(N, K, S):
    return bool(_lib.lib().p2c_fit_fused_supported(int(N), int(K), int(S)))


def fit_fused(X, Wb, Wc, bb, seg, P, rand_idx, normalize=False, axes64=False, K=None):
    """csrc/fit.hip fit_fused_kernel: axis -> hard centroids -> extents of pre-segmented clouds in one pass (eval.py:397, :409-436,
    data_utils.py:1650-1730).  -> axes (B,K,3), centroids (B,K,3), centroid found (B,K), extents (K,B,2), extent found (B,K)
    [, axes in float64 (B,K,3) with axes64=True].  No gradient.  Wb = Wc = None (K given): memberships implied by the labels, not read."""
    _lib.require_device(X, Wb, Wc, bb, seg, P, rand_idx)
    hard = Wb is None and Wc is None
    if (Wb is None) != (Wc is None):
        raise ValueError("fit_fused: W_barrel and W_base are given together or not at all")
    X, P = _f32c(X.detach()), _f32c(P)
    if hard:
        B, N = seg.shape
        if K is None:
            raise ValueError("fit_fused: K is required when the memberships are implied by the labels")
    else:
        Wb, Wc = _f32c(Wb.detach()), _f32c(Wc.detach())
        B, N, K = Wb.shape
    S = rand_idx.shape[2]
    if not fit_fused_supported(N, K, S):
        raise ValueError("fit_fused: shape N=%d K=%d S=%d is outside the fused kernel (K in {1,2,4,8}, cloud within the LDS); call the three ops" % (N, K, S))
    seg, bb, rand_idx = seg.to(torch.int64).contiguous(), bb.to(torch.int64).contiguous(), rand_idx.to(torch.int64).contiguous()
    dev = P.device
    axes = torch.empty(B, K, 3, dtype=torch.float32, device=dev)
    cen = torch.empty(B, K, 3, dtype=torch.float32, device=dev)
    cfound = torch.empty(B, K, dtype=torch.float32, device=dev)
    ext = torch.empty(K, B, 2, dtype=torch.float32, device=dev)
    found = torch.empty(B, K, dtype=torch.float32, device=dev)
    ws = torch.empty(_lib.lib().p2c_extents_ws_bytes(B, K) // 4 + 4, dtype=torch.float32, device=dev)
    a64 = torch.empty(B, K, 3, dtype=torch.float64, device=dev) if axes64 else None
    call("p2c_fit_fused_f32", ptr(X), ptr(Wb), ptr(Wc), ptr(bb), ptr(seg), 1 if normalize else 0, ptr(P), ptr(rand_idx), B, N, K, S,
         ptr(axes), ptr(cen), ptr(cfound), ptr(ext), ptr(found), ptr(a64), ptr(ws), stream(),
         nbytes=float(B) * N * (12 + 12 + (0 if hard else 2 * K * 4) + 16) + float(B) * K * S * 8)
    if axes64:
        return axes, cen, cfound, ext, found, a64
    return axes, cen, cfound, ext, found


def sketch_projection(P, X, seg, bb, axes, centers, rand_idx, S, all_points=False):
    """data_utils.py:1014-1417 (csrc/fit.hip).  rand_idx (B,K,S) int64 or None with all_points ->
    P_projected (K,B,S,2), X_projected (K,B,S,2), scales (K,B), found (B,K).  No gradient (the reference's projection is built
    from index gathers of detached samples as well: its callers only backpropagate through what consumes the sketch)."""
    _lib.require_device(P, X, axes, centers)
    P, X, axes, centers = _f32c(P.detach()), _f32c(X.detach()), _f32c(axes.detach()), _f32c(centers.detach())
    B, N, _ = P.shape
    K = axes.shape[1]
    if all_points:
        S, seg_, bb_, ri = N, None, None, None
    else:
        _lib.require_device(seg, bb, rand_idx)
        seg_, bb_, ri = seg.to(torch.int64).contiguous(), bb.to(torch.int64).contiguous(), rand_idx.to(torch.int64).contiguous()
        assert tuple(ri.shape) == (B, K, S)
    Pp = torch.empty(K, B, S, 2, dtype=torch.float32, device=P.device)
    Xp = torch.empty(K, B, S, 2, dtype=torch.float32, device=P.device)
    scales = torch.empty(K, B, dtype=torch.float32, device=P.device)
    found = torch.empty(B, K, dtype=torch.float32, device=P.device)
    ws = torch.empty(_lib.lib().p2c_extents_ws_bytes(B, K) // 4 + 4, dtype=torch.float32, device=P.device)
    call("p2c_sketch_projection_f32", ptr(P), ptr(X), ptr(seg_), ptr(bb_), ptr(axes), ptr(centers), ptr(ri), B, N, K, S, int(all_points),
         ptr(Pp), ptr(Xp), ptr(scales), ptr(found), ptr(ws), stream(), nbytes=float(B * K * S * (24 + 16)))
    return Pp, Xp, scales, found


def linear_sum_assignment(cost, solver=1):
    """scipy.optimize.linear_sum_assignment(cost)[1] for a batch: cost (P, nr, nc) fp64 on the device, nr <= nc <= 15 -> (P, nr) int32."""
    _lib.require_device(cost)
    cost = cost.to(torch.float64).contiguous()
    P, nr, nc = cost.shape
    out = torch.empty(P, nr, dtype=torch.int32, device=cost.device)
    call("p2c_linear_sum_assignment_f64", ptr(cost), P, nr, nc, ptr(out), int(solver), stream())
    return out


def check_labels(I_gt, K):
    """Instance labels must lie in [-1, K): losses.py:36-46 indexes eye(n_gt+1) and matching_indices[b, :n_gt] with them and raises
    otherwise; the kernels would silently drop the offending points instead.  One device->host sync - the reference's matching pays B
    of them per call; the graph-replayed training step validates its dataset once when it is loaded instead (train.ResidentDataset)."""
    if I_gt.is_cuda and torch.cuda.is_current_stream_capturing():
        return        # a device->host read cannot be captured; the eager warm-up passes in front of every capture have checked these labels
    hi, lo = int(I_gt.max()), int(I_gt.min())
    if hi >= K or lo < -1:
        raise ValueError("instance labels must be in [-1, %d); got [%d, %d]" % (K, lo, hi))


class _DeferredLabelCheck:
    """Label validation WITHOUT a device->host sync in the step: the range test runs on the device, its verdict travels through a pinned
    host word with an asynchronous copy, and the NEXT call (or flush_label_check()) raises if an earlier batch was out of range.  The
    synchronous check costs the caller of compute_all_losses its whole launch-ahead: the host waits for the forward to finish before it
    can enqueue the first loss kernel (measured on the drop-in step: 7.2 -> see DESIGN.md)."""

    SYNC_FIRST = 3       # the first calls validate synchronously (the reference raises in the offending call: a mis-labelled dataset
    #                      shows up at once); later batches are checked one call late, and at interpreter exit (atexit below)

    def __init__(self):
        self.st = {}
        self.calls = 0

    def __call__(self, I_gt, K):
        if torch.cuda.is_current_stream_capturing():
            return
        self.calls += 1
        if self.calls <= self.SYNC_FIRST:
            check_labels(I_gt, K)
            # ... and run the deferred machinery too while the caller is synchronous anyway: its first use pins a host word (tens of
            # milliseconds of page faults on a fresh process) and loads three torch kernels - in call SYNC_FIRST + 1 that was one step of
            # 18 - 60 ms in the middle of the first epoch (bench.py's drop-in leg)
        dev = I_gt.device
        st = self.st.get(dev)
        if st is None:
            st = self.st[dev] = dict(acc=torch.zeros(1, dtype=torch.int32, device=dev), host=torch.zeros(1, dtype=torch.int32).pin_memory(),
                                     ev=None, K=K)
        self.poll(dev)
        mn, mx = torch.aminmax(I_gt)
        st["acc"] |= ((mx >= K) | (mn < -1)).to(torch.int32)
        st["host"].copy_(st["acc"], non_blocking=True)
        st["ev"] = torch.cuda.Event()
        st["ev"].record()
        st["K"] = K

    def poll(self, dev, wait=False):
        st = self.st.get(dev)
        if st is None or st["ev"] is None:
            return
        if wait:
            st["ev"].synchronize()
        if st["ev"].query() and int(st["host"][0]) != 0:
            st["acc"].zero_()
            st["host"].zero_()
            st["ev"] = None
            raise ValueError("instance labels of an earlier batch were outside [-1, %d) (deferred check: losses.py:36-46 would have raised "
                             "in that call)" % st["K"])


check_labels_deferred = _DeferredLabelCheck()


def flush_label_check(device=None):
    """Wait for the outstanding deferred label checks and raise if one failed (end of an epoch, before a checkpoint)."""
    for dev in list(check_labels_deferred.st):
        if device is None or dev == torch.device(device):
            check_labels_deferred.poll(dev, wait=True)


def _flush_label_check_at_exit():
    """The LAST batch of an unchanged caller (the reference's trainer never calls flush_label_check) is still reported: on stderr, at exit."""
    import sys
    try:
        flush_label_check()
    except ValueError as e:
        sys.stderr.write("point2cyl_amd: %s\n" % e)
    except Exception:
        pass


import atexit
atexit.register(_flush_label_check_at_exit)


def hungarian(W, I_gt, validate=True):
    """losses.py:22-52 on the device -> matching_indices (B,K) int64, mask (B,K) bool.  No gradient.
    validate=False: the caller has range-checked these labels already (the evaluation loop does, on the host copy before the upload)."""
    _lib.require_device(W, I_gt)
    W = _f32c(W.detach())
    B, N, K = W.shape
    I_gt = I_gt.to(torch.int64).contiguous()
    if validate:
        check_labels(I_gt, K)
    match = torch.empty(B, K, dtype=torch.int64, device=W.device)
    mask = torch.empty(B, K, dtype=torch.uint8, device=W.device)
    call("p2c_hungarian_f32", ptr(W), ptr(I_gt), B, N, K, ptr(match), ptr(mask), stream())
    return match, mask.bool()


_HUNG_WS = {}


def _hungarian_ws(B, dev):
    """Partial sums of the split matching kernel (p2c_hip.h); one buffer per (B, device), reused."""
    key = (B, dev.index)
    if key not in _HUNG_WS:
        _HUNG_WS[key] = torch.empty(_lib.lib().p2c_hungarian_ws_bytes(B) // 4 + 4, dtype=torch.float32, device=dev)
    return _HUNG_WS[key]


_EVAL_WS = {}


def eval_metrics_supported(K):
    return bool(_lib.lib().p2c_eval_metrics_supported(int(K)))


def eval_metrics_fused(heads, xoff, woff, pcs, gt_normals, gt_inst, gt_bb, gt_axes, gt_centers, K, normalize=False, pi=None, out=None, details=False):
    """eval.py:270-446 with its default operands as TWO launches (csrc/metrics.hip) from the backbone's raw head output heads (B*N, ld).
    gt_inst int64 in [-1, K) (the caller checks the range), gt_bb float or integer 0/1.  -> out (5, B) float64 = per cloud mIoU, normal
    angle error, base/barrel accuracy, extrusion angle error, centroid difference (written into `out` when given: a captured graph's
    static block); details=True: (out, dict(matching_indices, mask, E64, predicted_centroids, found_centers_mask))."""
    _lib.require_device(heads, pcs)
    dev = heads.device
    B, N, _ = pcs.shape
    M, ld = heads.shape
    assert M == B * N and heads.stride(0) == ld and heads.stride(1) == 1 and heads.dtype == torch.float32
    if pi is None:
        from .losses import TORCH_PI as pi
    key = (B, K, dev.index)
    if key not in _EVAL_WS:
        _EVAL_WS[key] = torch.empty(_lib.lib().p2c_eval_metrics_ws_bytes(B, K) // 8 + 8, dtype=torch.float64, device=dev)
    if out is None:
        out = torch.empty(5, B, dtype=torch.float64, device=dev)
    det = None
    if details:
        det = dict(matching_indices=torch.empty(B, K, dtype=torch.int64, device=dev), mask=torch.empty(B, K, dtype=torch.uint8, device=dev),
                   E64=torch.empty(B, K, 3, dtype=torch.float64, device=dev), predicted_centroids=torch.empty(B, K, 3, dtype=torch.float32, device=dev),
                   found_centers_mask=torch.empty(B, K, dtype=torch.float32, device=dev))
    bbf = gt_bb if gt_bb.dtype == torch.float32 else gt_bb.to(torch.float32)
    gi = gt_inst if gt_inst.dtype == torch.int64 else gt_inst.to(torch.int64)
    d = det or {}
    call("p2c_eval_metrics_f32", ptr(heads), ld, int(xoff), int(woff), ptr(_f32c(pcs)), ptr(_f32c(gt_normals)), ptr(gi.contiguous()), ptr(bbf.contiguous()),
         ptr(_f32c(gt_axes)), ptr(_f32c(gt_centers)), int(bool(normalize)), float(N) * 0.005, float(pi), B, N, K, ptr(out),
         ptr(d.get("matching_indices")), ptr(d.get("mask")), ptr(d.get("E64")), ptr(d.get("predicted_centroids")), ptr(d.get("found_centers_mask")),
         ptr(_EVAL_WS[key]), stream())
    return (out, det) if details else out


class _SegLosses(torch.autograd.Function):
    """total = w_seg*mIoU + w_normal*normal + w_bb*base/barrel on the raw head output (fused forward + gradient)."""

    @staticmethod
    def forward(ctx, heads, normals_gt, I_gt, bb_gt, B, N, K, xoff, woff, w_seg, w_normal, w_bb):
        dev = heads.device
        M, ld = heads.shape
        assert heads.stride(0) == ld and heads.stride(1) == 1
        I_gt, bb_gt = I_gt.to(torch.int64).contiguous(), bb_gt.to(torch.int64).contiguous()
        match = torch.empty(B, K, dtype=torch.int64, device=dev)
        mask = torch.empty(B, K, dtype=torch.uint8, device=dev)
        hd = heads.detach()
        call("p2c_hungarian_logits_f32", ptr(hd), ld, woff, ptr(I_gt), B, N, K, ptr(match), ptr(mask), ptr(_hungarian_ws(B, dev)), stream())
        out = torch.empty(4, dtype=torch.float32, device=dev)
        ws = STEP_ARENA.take(_lib.lib().p2c_seg_losses_ws_bytes(B, K) // 8 + 8, dev)        # zeroed scratch: out of the step arena when one is active
        ngt = _f32c(normals_gt)
        # forward launches only (dheads = NULL): the gradient is produced in backward, already multiplied by the upstream gradient
        # (it used to be kept from the forward and scaled by a 21 MB torch multiply)
        call("p2c_seg_losses_f32", ptr(hd), ld, xoff, woff, ptr(ngt), ptr(I_gt), ptr(bb_gt), ptr(match), ptr(mask), B, N, K,
             float(w_seg), float(w_normal), float(w_bb), ptr(out), None, ptr(ws), stream())
        ctx.save_for_backward(hd, ngt, I_gt, bb_gt, match, mask)
        ctx.epoch = STEP_ARENA.epoch if STEP_ARENA.active else None
        ctx.ws = ws                  # (a slice of the step arena shares the arena's version counter: every in-place write to ANY arena slice would
        #                              trip save_for_backward's check; the slice itself is written by these two entry points only)
        ctx.cfg = (B, N, K, xoff, woff, float(w_seg), float(w_normal), float(w_bb))
        ctx.mark_non_differentiable(match, mask)
        ctx.set_materialize_grads(False)       # no zero tensors (a fill launch each, ~5 us of the stream) for the outputs nobody differentiates
        return out, match, mask

    @staticmethod
    def backward(ctx, gout, gmatch, gmask):
        if gout is None:
            return (None,) * 12
        hd, ngt, I_gt, bb_gt, match, mask = ctx.saved_tensors
        ws = ctx.ws
        if ctx.epoch is not None and ctx.epoch != STEP_ARENA.epoch:
            raise RuntimeError("point2cyl_amd.ops.seg_losses: backward after the step arena it was recorded in has been cleared (a later "
                               "`with ops.step_arena(...)` began): the matched sums it reads are gone.  Run backward inside the same arena step")
        B, N, K, xoff, woff, w_seg, w_normal, w_bb = ctx.cfg
        M, ld = hd.shape
        gout = _f32c(gout)
        dheads = torch.empty(M, ld, dtype=torch.float32, device=hd.device)
        # only d / d total is propagated (the other three scalars are logging values): gout[0] is the kernel's scale
        call("p2c_seg_losses_grad_f32", ptr(hd), ld, xoff, woff, ptr(ngt), ptr(I_gt), ptr(bb_gt), ptr(match), ptr(mask), B, N, K,
             w_seg, w_normal, w_bb, ptr(gout), ptr(dheads), ptr(ws), stream())
        return (dheads,) + (None,) * 11


class _HeadPost(torch.autograd.Function):
    """heads (B*N, ld), matching_indices (B,K) -> unit normals X (B,N,3), matched barrel / base probabilities Wb, Wc (B,N,K)
    (csrc/loss.hip: head_post_kernel and its backward)."""

    @staticmethod
    def forward(ctx, heads, match, B, N, K, xoff, woff):
        _lib.require_device(heads, match)
        heads = heads if heads.is_contiguous() else heads.contiguous()
        match = match.to(torch.int64).contiguous()
        ld = heads.stride(0)
        X = torch.empty(B, N, 3, dtype=torch.float32, device=heads.device)
        Wb = torch.empty(B, N, K, dtype=torch.float32, device=heads.device)
        Wc = torch.empty(B, N, K, dtype=torch.float32, device=heads.device)
        call("p2c_head_post_f32", ptr(heads), ld, xoff, woff, ptr(match), B, N, K, ptr(X), ptr(Wb), ptr(Wc), stream())
        ctx.save_for_backward(heads, match)
        ctx.cfg = (B, N, K, xoff, woff)
        return X, Wb, Wc

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dX, dWb, dWc):
        heads, match = ctx.saved_tensors
        B, N, K, xoff, woff = ctx.cfg
        c = lambda t: None if t is None else t.contiguous()
        dX, dWb, dWc = c(dX), c(dWb), c(dWc)
        dh = torch.empty_like(heads)
        call("p2c_head_post_bwd_f32", ptr(heads), heads.stride(0), xoff, woff, ptr(match), B, N, K, ptr(dX), ptr(dWb), ptr(dWc), ptr(dh),
             dh.stride(0), stream())
        return dh, None, None, None, None, None, None


class _FitTerms(torch.autograd.Function):
    """(w_ext * extrusion-axis loss, w_center * centre loss) from the fitted axes / centres (B,K,3), forward + gradient in ONE launch
    (csrc/loss.hip fit_terms_kernel) instead of ~36 torch launches on 256-element tensors forward and as many backward."""

    @staticmethod
    def forward(ctx, E_AX, gt_axes, centers, gt_centers, mask, w_ext, w_center):
        ref = E_AX if E_AX is not None else centers
        B, K = ref.shape[0], ref.shape[1]
        dev = ref.device
        out2 = torch.empty(2, dtype=torch.float32, device=dev)
        E = _f32c(E_AX.detach()) if E_AX is not None else None
        C = _f32c(centers.detach()) if centers is not None else None
        dE = torch.empty_like(E) if E is not None else None
        dC = torch.empty_like(C) if C is not None else None
        m8 = mask.view(torch.uint8) if mask.dtype == torch.bool else mask.to(torch.uint8)
        call("p2c_fit_terms_f32", ptr(E), ptr(_f32c(gt_axes)) if E is not None else None, ptr(C), ptr(_f32c(gt_centers)) if C is not None else None,
             ptr(m8.contiguous()), B, K, float(w_ext), float(w_center), ptr(out2), ptr(dE), ptr(dC), stream())
        ctx.grads = (dE, dC)
        ctx.set_materialize_grads(False)
        return out2

    @staticmethod
    def backward(ctx, gout):
        dE, dC = ctx.grads
        if gout is None:
            return (None,) * 7
        return (None if dE is None else dE * gout[0], None, None if dC is None else dC * gout[1], None, None, None, None)


def fit_terms(E_AX, gt_axes, centers, gt_centers, mask, w_ext=1.0, w_center=1.0):
    """-> (2,) = [w_ext * mean_b masked-mean_k (1 - |E_AX . gt_axes|), w_center * mean_b masked-mean_k |centers - gt_centers|^2]
    (train_Point2Cyl_without_sketch.py:326-332, :342-353; losses.py:83-88).  E_AX / centers (B,K,3) or None (term off: 0);
    mask (B,K) bool / bytes = k < number of ground-truth instances (hungarian's mask = losses.get_mask_gt)."""
    _lib.require_device(mask)
    return _FitTerms.apply(E_AX, gt_axes, centers, gt_centers, mask, w_ext, w_center)


def head_post(heads, match, B, N, K, xoff=0, woff=3):
    return _HeadPost.apply(heads, match, B, N, K, xoff, woff)


def seg_losses(heads, normals_gt, I_gt, bb_gt, B, N, K, xoff, woff, w_seg=1.0, w_normal=1.0, w_bb=1.0):
    """-> (out[4] = total, normal, miou, bb ; matching_indices (B,K) int64 ; mask (B,K) bool)."""
    out, match, mask = _SegLosses.apply(heads, normals_gt, I_gt, bb_gt, B, N, K, xoff, woff, w_seg, w_normal, w_bb)
    return out, match, mask.view(torch.bool)          # (0 / 1 bytes: a reinterpretation, not a conversion launch)


class _AllLosses(torch.autograd.Function):
    """losses.compute_all_losses (losses.py:317-351, collapse=True) on (W, X): matching + both means forward, the two unweighted gradients
    kept for backward (csrc/loss.hip all_losses_*).  out3 = [w_seg * miou + w_normal * normal, normal, miou]."""

    @staticmethod
    def forward(ctx, W, X, normals_gt, I_gt, w_normal, w_seg):
        dev = W.device
        Wc, Xc = _f32c(W.detach()), _f32c(X.detach())
        B, N, K = Wc.shape
        I_gt = I_gt.to(torch.int64).contiguous()
        (check_labels if STRICT_LABELS else check_labels_deferred)(I_gt, K)
        match = torch.empty(B, K, dtype=torch.int64, device=dev)
        mask = torch.empty(B, K, dtype=torch.uint8, device=dev)
        call("p2c_hungarian_f32", ptr(Wc), ptr(I_gt), B, N, K, ptr(match), ptr(mask), stream())
        out2 = torch.empty(2, dtype=torch.float32, device=dev)
        dW, dX = torch.empty_like(Wc), torch.empty_like(Xc)
        ws = STEP_ARENA.take(_lib.lib().p2c_all_losses_ws_bytes(B, K) // 8 + 8, dev)
        call("p2c_all_losses_f32", ptr(Wc), ptr(Xc), ptr(_f32c(normals_gt)), ptr(I_gt), ptr(match), ptr(mask), B, N, K, ptr(out2), ptr(dW), ptr(dX),
             ptr(ws), stream())
        ctx.save_for_backward(dW, dX)
        ctx.w = (float(w_normal), float(w_seg))
        ctx.mark_non_differentiable(match, mask)
        ctx.set_materialize_grads(False)
        out3 = torch.stack([w_seg * out2[1] + w_normal * out2[0], out2[0], out2[1]])
        return out3, match, mask

    @staticmethod
    def backward(ctx, gout, gmatch, gmask):
        dW, dX = ctx.saved_tensors
        if gout is None:
            return (None,) * 6
        w_normal, w_seg = ctx.w
        return dW * (gout[0] * w_seg + gout[2]), dX * (gout[0] * w_normal + gout[1]), None, None, None, None


def all_losses(W, X, normals_gt, I_gt, w_normal, w_seg):
    """-> (out3 = [total, normal, miou], matching_indices (B,K) int64, mask (B,K) bool); K = 1 ... 8."""
    _lib.require_device(W, X, normals_gt, I_gt)
    out3, match, mask = _AllLosses.apply(W, X, normals_gt, I_gt, w_normal, w_seg)
    return out3, match, mask.bool()
