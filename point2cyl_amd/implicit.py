"""The sketch branch's implicit decoder: drop-in for the reference's IGR/network.py `ImplicitNet`
(:20-92), `gradient` (:8-17) and `add_latent` (:200-206).

The with-sketch trainer differentiates the decoder TWICE: the eikonal / normal losses are functions of d(output)/d(point)
(`gradient`, torch.autograd.grad with create_graph=True), and the optimiser step needs their derivative w.r.t. the weights
(train_Point2Cyl.py:608-648).  Every matrix product of all three passes runs on this package's GEMM kernels (csrc/gemm.hip through
the C ABI; the 512-wide products of the trainer's shapes on csrc/gemm_big.hip, `_big`): the three product shapes

    NT(X, W) = X W^T        NN(A, W) = A W        TN(A, X) = A^T X

are closed under differentiation (d NT = {NN, TN}, d NN = {NT, TN}, d TN = {NT, NN}), so three autograd Functions whose backward
passes call each other give derivatives of any order with nothing but those kernels.  Channel counts are zero-padded to multiples of
4 on the way in (258 -> 260 inputs, 254 -> 256 outputs of the skip layer, 1 -> 4 outputs of the last) and sliced on the way out;
parameters keep the reference's shapes and names (lin0 .. lin8).  The bias is added in the GEMM epilogue; the softplus between the products and the two derivatives of it that
the double backward evaluates are one pass each (csrc/softplus.hip); moving them into the GEMM prologues / epilogues is the next
step for this row.
There is no CPU path."""
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from ._lib import call, ptr, stream


def _c(t):
    """fp32, contiguous, 16-byte aligned - as a DIFFERENTIABLE op applied before a Function sees the tensor: a Function must save
    its own inputs (not a private copy made inside forward), or the second derivative through the saved tensor is cut."""
    if t is None:
        return None
    t = t if t.dtype == torch.float32 else t.float()
    if not t.is_contiguous():
        t = t.contiguous()
    return t if t.data_ptr() % 16 == 0 else t.clone()


def _nt(X, W, b=None):
    return _NT.apply(_c(X), _c(W), _c(b))


def _nn(A, W):
    return _NN.apply(_c(A), _c(W))


def _tn(A, X):
    return _TN.apply(_c(A), _c(X))


USE_BIG = os.environ.get("P2C_GEMM_BIG", "1") != "0"      # A/B switch: the decoder's large products on csrc/gemm_big.hip
INFER_EPILOGUE = True     # ImplicitNet.forward without gradients: softplus in the big-tile product's epilogue, no pre-activations kept (a test switches it off)


def _big(M, N, K, dev):
    """-> workspace for the split image of W if (M, N, K) is a shape of the big-tile kernels, else None."""
    if not USE_BIG or not _lib.lib().p2c_linear_big_supported(M, N, K):
        return None
    return torch.empty(_lib.lib().p2c_linear_big_ws_bytes(N, K), dtype=torch.uint8, device=dev)


def _check(*ts):
    _lib.require_device(*ts)
    for t in ts:
        if t.shape[-1] % 4:
            raise ValueError("implicit GEMMs need channel counts that are multiples of 4 (got %s)" % (tuple(t.shape),))


class _NT(torch.autograd.Function):
    """X [M,K], W [N,K], bias [N] or None -> X W^T + bias [M,N]  (p2c_linear_fwd_f32: the bias is added in the GEMM epilogue)."""

    @staticmethod
    def forward(ctx, X, W, bias):
        _check(X, W)
        M, K = X.shape
        N = W.shape[0]
        Y = torch.empty(M, N, dtype=torch.float32, device=X.device)
        ws = _big(M, N, K, X.device)
        if ws is not None:
            call("p2c_linear_fwd_big_f32", ptr(X), K, ptr(W), K, ptr(bias), ptr(Y), N, M, N, K, ptr(ws), stream(), flops=2.0 * M * N * K)
        else:
            call("p2c_linear_fwd_f32", ptr(X), K, ptr(W), K, ptr(bias), ptr(Y), N, M, N, K, 0, None, None, None, 0, 1.0, None, stream(),
                 flops=2.0 * M * N * K)
        ctx.save_for_backward(X, W)
        return Y

    @staticmethod
    def backward(ctx, dY):
        X, W = ctx.saved_tensors
        return (_nn(dY, W) if ctx.needs_input_grad[0] else None, _tn(dY, X) if ctx.needs_input_grad[1] else None,
                dY.sum(0) if ctx.needs_input_grad[2] else None)


class _NN(torch.autograd.Function):
    """A [M,N], W [N,K] -> A W [M,K]  (p2c_linear_bwd_data_f32, plain gradient mode)."""

    @staticmethod
    def forward(ctx, A, W):
        _check(A, W)
        M, N = A.shape
        K = W.shape[1]
        O = torch.empty(M, K, dtype=torch.float32, device=A.device)
        ws = _big(M, N, K, A.device)
        if ws is not None:
            call("p2c_linear_bwd_data_big_f32", ptr(A), N, ptr(W), K, None, 0, 0.0, 0.0, ptr(O), K, M, N, K, ptr(ws), stream(), flops=2.0 * M * N * K)
        else:
            call("p2c_linear_bwd_data_f32", ptr(A), N, None, 0, 0, None, ptr(W), K, ptr(O), K, M, N, K, None, 0, 1.0, None, 0, None, None, None, 0,
                 stream(), flops=2.0 * M * N * K)
        ctx.save_for_backward(A, W)
        return O

    @staticmethod
    def backward(ctx, dO):
        A, W = ctx.saved_tensors
        return (_nt(dO, W) if ctx.needs_input_grad[0] else None, _tn(A, dO) if ctx.needs_input_grad[1] else None)


class _TN(torch.autograd.Function):
    """A [M,N], X [M,K] -> A^T X [N,K]  (p2c_linear_bwd_weight_f32: split over the M rows, fp32 atomics into a zeroed result)."""

    @staticmethod
    def forward(ctx, A, X):
        _check(A, X)
        M, N = A.shape
        K = X.shape[1]
        G = torch.zeros(N, K, dtype=torch.float32, device=A.device)
        call("p2c_linear_bwd_weight_f32", ptr(A), N, None, 0, 0, None, ptr(X), K, 0, None, None, None, 0, 1.0, ptr(G), K, 0, None, M, N, K, None, 0,
             stream(), flops=2.0 * M * N * K)
        ctx.save_for_backward(A, X)
        return G

    @staticmethod
    def backward(ctx, dG):
        A, X = ctx.saved_tensors
        return (_nt(X, dG) if ctx.needs_input_grad[0] else None, _nn(A, dG) if ctx.needs_input_grad[1] else None)


class _Softplus(torch.autograd.Function):
    """h = softplus(z) (csrc/softplus.hip); backward = u * sigmoid(beta z), itself differentiable (below)."""

    @staticmethod
    def forward(ctx, z, beta, thr):
        _lib.require_device(z)
        h = torch.empty_like(z)
        call("p2c_softplus_fwd_f32", ptr(z), ptr(h), z.numel(), beta, thr, stream(), nbytes=8.0 * z.numel())
        ctx.save_for_backward(z)
        ctx.bt = (beta, thr)
        return h

    @staticmethod
    def backward(ctx, dh):
        (z,) = ctx.saved_tensors
        return _SoftplusBwd.apply(_c(dh), z, *ctx.bt), None, None


class _SoftplusBwd(torch.autograd.Function):
    """o = u * s(z), s = sigmoid(beta z); backward (one pass for both): du = g * s(z), dz = g * u * beta * s (1 - s)."""

    @staticmethod
    def forward(ctx, u, z, beta, thr):
        o = torch.empty_like(z)
        call("p2c_softplus_bwd_f32", ptr(u), ptr(z), ptr(o), z.numel(), beta, thr, stream(), nbytes=12.0 * z.numel())
        ctx.save_for_backward(u, z)
        ctx.bt = (beta, thr)
        return o

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        u, z = ctx.saved_tensors
        g = _c(g)
        du, dz = torch.empty_like(z), torch.empty_like(z)
        call("p2c_softplus_bwd_bwd_f32", ptr(g), ptr(u), ptr(z), ptr(du), ptr(dz), z.numel(), *ctx.bt, stream(), nbytes=20.0 * z.numel())
        return du, dz, None, None


def softplus(z, beta=100.0, threshold=20.0):
    """F.softplus(z, beta, threshold) on the device kernels, twice differentiable (what a training step of the decoder needs)."""
    return _Softplus.apply(_c(z), float(beta), float(threshold))


class _NNSig(torch.autograd.Function):
    """A [M,N], W [N,K], z [M,K] -> (A W) * s(z), s = sigmoid(beta z): the gradient that a layer fed by softplus(z) sends to z, with the
    activation's derivative in the GEMM epilogue (p2c_linear_bwd_data_sig_f32).  Its own backward needs only its output a:
    d/d(A W) = g * s(z) and d/dz = g * a * beta * (1 - s(z)), one pass (p2c_softplus_sig_bwd_f32)."""

    @staticmethod
    def forward(ctx, A, W, z, beta, thr):
        _check(A, W, z)
        M, N = A.shape
        K = W.shape[1]
        a = torch.empty(M, K, dtype=torch.float32, device=A.device)
        ws = _big(M, N, K, A.device)
        if ws is not None:
            call("p2c_linear_bwd_data_big_f32", ptr(A), N, ptr(W), K, ptr(z), K, beta, thr, ptr(a), K, M, N, K, ptr(ws), stream(), flops=2.0 * M * N * K)
        else:
            call("p2c_linear_bwd_data_sig_f32", ptr(A), N, ptr(W), K, ptr(z), K, beta, thr, ptr(a), K, M, N, K, stream(), flops=2.0 * M * N * K)
        ctx.save_for_backward(A, W, z, a)
        ctx.bt = (beta, thr)
        return a

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        A, W, z, a = ctx.saved_tensors
        g = _c(g)
        t, dz = torch.empty_like(z), torch.empty_like(z)
        call("p2c_softplus_sig_bwd_f32", ptr(g), ptr(a), ptr(z), ptr(t), ptr(dz), z.numel(), *ctx.bt, stream(), nbytes=20.0 * z.numel())
        dA = _nt(t, W) if ctx.needs_input_grad[0] else None
        dW = _tn(A, t) if ctx.needs_input_grad[1] else None
        return dA, dW, (dz if ctx.needs_input_grad[2] else None), None, None


class _SpLinear(torch.autograd.Function):
    """z [M,K], W [N,K], bias -> softplus(z) W^T + bias: one decoder layer on the PRE-activations.  Its backward is differentiable
    (the trainer differentiates the decoder w.r.t. its input with the graph kept): dz = NNSig(dY, W, z)."""

    @staticmethod
    def forward(ctx, z, W, bias, beta, thr):
        _check(z, W)
        M, K = z.shape
        N = W.shape[0]
        h = torch.empty_like(z)
        call("p2c_softplus_fwd_f32", ptr(z), ptr(h), z.numel(), beta, thr, stream(), nbytes=8.0 * z.numel())
        Y = torch.empty(M, N, dtype=torch.float32, device=z.device)
        ws = _big(M, N, K, z.device)
        if ws is not None:
            call("p2c_linear_fwd_big_f32", ptr(h), K, ptr(W), K, ptr(bias), ptr(Y), N, M, N, K, ptr(ws), stream(), flops=2.0 * M * N * K)
        else:
            call("p2c_linear_fwd_f32", ptr(h), K, ptr(W), K, ptr(bias), ptr(Y), N, M, N, K, 0, None, None, None, 0, 1.0, None, stream(), flops=2.0 * M * N * K)
        ctx.save_for_backward(z, W)
        ctx.bt = (beta, thr)
        ctx.has_bias = bias is not None
        return Y

    @staticmethod
    def backward(ctx, dY):
        z, W = ctx.saved_tensors
        dY = _c(dY)
        dz = _NNSig.apply(dY, W, z, *ctx.bt) if ctx.needs_input_grad[0] else None
        dW = _tn(dY, softplus(z, *ctx.bt)) if ctx.needs_input_grad[1] else None          # h is recomputed: only a trainable decoder needs it
        db = dY.sum(0) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        return dz, dW, db, None, None


def sp_linear(z, weight, bias, beta=100.0, threshold=20.0):
    """F.linear(F.softplus(z, beta, threshold), weight, bias), twice differentiable; widths that are multiples of 4 only."""
    return _SpLinear.apply(_c(z), _c(weight), _c(bias), float(beta), float(threshold))


def linear(x, weight, bias):
    """F.linear(x, weight, bias) for x [M,K] on the GEMM kernels, differentiable to any order."""
    N, K = weight.shape
    kp, np_ = (-K) % 4, (-N) % 4
    if kp:
        x = F.pad(x, (0, kp))
    w = F.pad(weight, (0, kp, 0, np_)) if (kp or np_) else weight
    b = F.pad(bias, (0, np_)) if (bias is not None and np_) else bias
    y = _nt(x, w, b)
    return y[:, :N] if np_ else y


def gradient(inputs, outputs):
    """IGR/network.py:8-17: d(sum of outputs)/d(inputs) with the graph kept (the result is differentiated again by the training step),
    restricted to the last two input columns - the 2-D sketch point behind the latent code."""
    (g,) = torch.autograd.grad(outputs, inputs, grad_outputs=torch.ones_like(outputs), create_graph=True, retain_graph=True)
    return g[:, -2:]


def add_latent(points, latent_codes, pad=False):
    """IGR/network.py:200-206: points (B', S, d), latent_codes (B', L) -> rows [code of the sketch | point], (B'*S, L + d).
    pad=True: zero columns up to the next multiple of 4 are appended by the same concatenation (what the GEMM kernels want: no padding copy)."""
    nb, ns, d = points.shape
    L = latent_codes.shape[-1]
    codes = latent_codes[:, None, :].expand(nb, ns, L)
    cols = [codes.reshape(nb * ns, -1), points.reshape(nb * ns, d)]
    extra = (-(L + d)) % 4 if pad else 0
    if extra:
        cols.append(torch.zeros(1, extra, dtype=points.dtype, device=points.device).expand(nb * ns, extra))
    return torch.cat(cols, dim=1)


class ImplicitNet(nn.Module):
    """IGR/network.py:20-92: d_in -> dims... -> 1 with softplus(beta) between the layers, the input re-injected (concat, / sqrt 2)
    at the layers in `skip_in`, IGR's geometric initialisation.  Same constructor, same parameter names (lin<i>.weight / .bias)."""

    def __init__(self, d_in, dims, skip_in=(), geometric_init=True, radius_init=1, beta=100):
        super().__init__()
        dims = [d_in] + dims + [1]
        self.num_layers = len(dims)
        self.skip_in = skip_in
        self.beta = beta
        for layer in range(0, self.num_layers - 1):
            out_dim = dims[layer + 1] - d_in if layer + 1 in skip_in else dims[layer + 1]
            lin = nn.Linear(dims[layer], out_dim)
            if geometric_init:                                         # network.py:46-56
                if layer == self.num_layers - 2:
                    torch.nn.init.normal_(lin.weight, mean=np.sqrt(np.pi) / np.sqrt(dims[layer]), std=0.00001)
                    torch.nn.init.constant_(lin.bias, -radius_init)
                else:
                    torch.nn.init.constant_(lin.bias, 0.0)
                    torch.nn.init.normal_(lin.weight, 0.0, np.sqrt(2) / np.sqrt(out_dim))
            setattr(self, "lin" + str(layer), lin)

    def forward(self, input):
        if not input.is_cuda:
            raise RuntimeError("point2cyl_amd.implicit.ImplicitNet runs on the HIP device only (got %s); there is no CPU path" % input.device)
        # x is kept as the PRE-activation of the previous layer where the next layer can take it that way (_SpLinear: softplus and its
        # derivative ride on the GEMMs); the skip layer (needs the activation itself for the concat) and ReLU take the plain route.
        # Widths: the GEMMs want multiples of 4 and the reference's are 258 (input), 254 (the layer in front of the skip), 1 (output).  x keeps
        # its PADDED width from layer to layer - `segs` lists its column segments as (real, padded) - and every layer's weight is laid out
        # to match (zero columns under x's pad columns, zero rows / bias entries for its own): a few small pads of 0.5 MB weights per
        # evaluation instead of a slice + copy (forward), a zero-fill + copy (backward) and again (backward of backward) of 270 - 600 MB
        # activations per odd-width layer.  A pad column of x holds 0 after a product and softplus(0) after the activation; it only ever meets
        # zero weights.
        d_in = input.shape[1]
        inp = F.pad(input, (0, (-d_in) % 4)) if d_in % 4 else input
        x, pending = inp, False                         # pending: x still needs its softplus
        infer = INFER_EPILOGUE and not (torch.is_grad_enabled() and (input.requires_grad or any(p_.requires_grad for p_ in self.parameters())))
        segs = [(d_in, inp.shape[1])]
        N = Np = d_in
        for layer in range(0, self.num_layers - 1):
            lin = getattr(self, "lin" + str(layer))
            N, K = lin.weight.shape
            fused = pending and layer not in self.skip_in
            if pending and not fused:
                x, pending = softplus(x, self.beta), False
            w = lin.weight
            if layer in self.skip_in:
                # IGR/network.py:75-76: x = cat([x, input]) / sqrt(2) in front of the layer.  The division is applied to the layer's WEIGHT
                # instead ((c / s) W^T = c (W / s)^T): a 0.5 MB operand instead of three passes over the 537 MB activation (forward, backward
                # and backward-of-backward each carried a scale kernel).
                x = torch.cat([x, inp], -1)
                segs = segs + [(d_in, inp.shape[1])]
                w = w * (1.0 / np.sqrt(2))
            if sum(r for r, _ in segs) != K:
                raise RuntimeError("ImplicitNet: layer %d expects %d inputs, got %d" % (layer, K, sum(r for r, _ in segs)))
            Np = N + (-N) % 4
            b = lin.bias
            if any(r != p_ for r, p_ in segs):
                pieces, o = [], 0
                for r, p_ in segs:
                    pieces.append(F.pad(w[:, o:o + r], (0, p_ - r)) if p_ != r else w[:, o:o + r])
                    o += r
                w = torch.cat(pieces, 1) if len(pieces) > 1 else pieces[0]
            if Np != N:
                w = F.pad(w, (0, 0, 0, Np - N))
                b = F.pad(b, (0, Np - N)) if b is not None else None
            segs = [(N, Np)]
            if infer:
                # inference: x carries ACTIVATIONS; the layer's softplus rides in the product's epilogue, no pre-activation is kept
                x = _prod_nt(_c(x), _c(w), b, sp=(self.beta, 20.0) if (layer < self.num_layers - 2 and self.beta > 0) else None)
                if layer < self.num_layers - 2 and not self.beta > 0:
                    x = F.relu(x)
                continue
            x = sp_linear(x, w, b, self.beta) if fused else _nt(x, w, b)
            if layer < self.num_layers - 2:
                if self.beta > 0:
                    pending = True
                else:
                    x = F.relu(x)
        return x[:, :N] if Np != N else x


class NormalPerPoint:
    """IGR/sampler.py:18-40: off-surface samples of a batch of sketches (B', S, d): every sketch point jittered by N(0, local_sigma)
    (a per-point sigma tensor may be passed instead), followed by S // 8 points drawn uniformly from [-global_sigma, global_sigma]^d;
    drawn on the input's device with torch's generator, like the reference."""

    def __init__(self, global_sigma, local_sigma=0.01):
        self.global_sigma = global_sigma
        self.local_sigma = local_sigma

    def get_points(self, pc_input, local_sigma=None):
        nb, ns, d = pc_input.shape
        sigma = self.local_sigma if local_sigma is None else local_sigma.unsqueeze(-1)
        near = pc_input + torch.randn_like(pc_input) * sigma
        far = (torch.rand(nb, ns // 8, d, device=pc_input.device) * 2.0 - 1.0) * self.global_sigma
        return torch.cat([near, far], dim=1)


# ------------------------------------------------------------------------------------------ value + input gradient of a FROZEN decoder as one node
def _prod_nt(X, W, bias=None, add=None, sp=None):
    """X [M,K] . W [N,K]^T (+ bias) (+ add [M,N]) -> [M,N]; the big-tile kernel takes the addend in its epilogue, other shapes add afterwards.
    sp = (beta, threshold): softplus of the result (inference: in the big-tile product's epilogue, a separate pass for other shapes)."""
    M, K = X.shape
    N = W.shape[0]
    Y = torch.empty(M, N, dtype=torch.float32, device=X.device)
    ws = _big(M, N, K, X.device)
    if ws is not None and X.stride(0) % 4 == 0:
        if sp is not None:
            call("p2c_linear_fwd_big_sp_f32", ptr(X), X.stride(0), ptr(W), K, ptr(bias), ptr(add), add.stride(0) if add is not None else 0,
                 float(sp[0]), float(sp[1]), ptr(Y), N, M, N, K, ptr(ws), stream(), flops=2.0 * M * N * K)
        elif add is not None:
            call("p2c_linear_fwd_big_add_f32", ptr(X), X.stride(0), ptr(W), K, ptr(bias), ptr(add), add.stride(0), ptr(Y), N, M, N, K, ptr(ws), stream(),
                 flops=2.0 * M * N * K)
        else:
            call("p2c_linear_fwd_big_f32", ptr(X), X.stride(0), ptr(W), K, ptr(bias), ptr(Y), N, M, N, K, ptr(ws), stream(), flops=2.0 * M * N * K)
        return Y
    call("p2c_linear_fwd_f32", ptr(X), X.stride(0), ptr(W), K, ptr(bias), ptr(Y), N, M, N, K, 0, None, None, None, 0, 1.0, None, stream(),
         flops=2.0 * M * N * K)
    if add is not None:
        Y.add_(add)
    if sp is not None:
        call("p2c_softplus_fwd_f32", ptr(Y), ptr(Y), Y.numel(), float(sp[0]), float(sp[1]), stream())
    return Y


def _prod_nn(A, W, z=None, beta=0.0, thr=20.0, add=None):
    """(A [M,N] . W [N,K]) (* sigmoid(beta z) when z is given) (+ add) -> [M,K]."""
    M, N = A.shape
    K = W.shape[1]
    O = torch.empty(M, K, dtype=torch.float32, device=A.device)
    ws = _big(M, N, K, A.device)
    if ws is not None:
        if add is not None:
            call("p2c_linear_bwd_data_big_add_f32", ptr(A), N, ptr(W), K, ptr(z), K if z is not None else 0, beta, thr, ptr(add), add.stride(0), ptr(O), K,
                 M, N, K, ptr(ws), stream(), flops=2.0 * M * N * K)
        else:
            call("p2c_linear_bwd_data_big_f32", ptr(A), N, ptr(W), K, ptr(z), K if z is not None else 0, beta, thr, ptr(O), K, M, N, K, ptr(ws), stream(),
                 flops=2.0 * M * N * K)
        return O
    if z is not None:
        call("p2c_linear_bwd_data_sig_f32", ptr(A), N, ptr(W), K, ptr(z), K, beta, thr, ptr(O), K, M, N, K, stream(), flops=2.0 * M * N * K)
    else:
        call("p2c_linear_bwd_data_f32", ptr(A), N, None, 0, 0, None, ptr(W), K, ptr(O), K, M, N, K, None, 0, 1.0, None, 0, None, None, None, 0, stream(),
             flops=2.0 * M * N * K)
    if add is not None:
        O.add_(add)
    return O


_LAYOUTS = None


def _decoder_layout_cached(net, d_in):
    """_decoder_layout(net, d_in) + the small derived operands (the g-block / point-column slices), rebuilt only when a parameter of the
    (frozen) decoder changed: ~150 pad / slice launches per step otherwise.  Built outside stream capture (the capture's eager warm-up
    passes come first), so a graph never owns the cached tensors."""
    global _LAYOUTS
    import weakref
    if _LAYOUTS is None:
        _LAYOUTS = weakref.WeakKeyDictionary()
    key = (d_in,) + tuple((p.data_ptr(), p._version) for p in net.parameters())
    hit = _LAYOUTS.get(net)
    if hit is not None and hit[0] == key:
        return hit[1]
    if torch.cuda.is_current_stream_capturing():
        return _decoder_layout_full(net, d_in)          # (not cached: tensors created under capture belong to the graph's pool)
    out = _decoder_layout_full(net, d_in)
    _LAYOUTS[net] = (key, out)
    return out


def _decoder_layout_full(net, d_in):
    lay = _decoder_layout(net, d_in)
    g0 = (d_in - 2) // 4 * 4
    extra = dict(g0=g0, w0g=lay[0][0][:, g0:].contiguous(), w_last=lay[-1][0][0].contiguous(),
                 wbg=[None if l[1] is None else l[1][:, g0:].contiguous() for l in lay],
                 cols=[None if (i > 0 and l[1] is None) else ((l[0] if i == 0 else l[1])[:, d_in - 2].contiguous(), (l[0] if i == 0 else l[1])[:, d_in - 1].contiguous())
                       for i, l in enumerate(lay)])
    return lay, extra


def _decoder_layout(net, d_in):
    """The padded operands of `net` for inputs of width d_in: per layer (Wa [Np, Kp_prev], Wb [Np, Dp] | None, bias [Np], N, Np) with
    Dp = pad4(d_in); Wb is the block of a skip layer's weight that multiplies the re-injected input, both blocks already divided by sqrt 2
    (IGR/network.py:75-76).  Pad rows / columns / bias entries are zero."""
    Dp = d_in + (-d_in) % 4
    out, prev_n, prev_np = [], d_in, Dp
    for layer in range(net.num_layers - 1):
        lin = getattr(net, "lin" + str(layer))
        w, b = lin.weight.detach(), lin.bias.detach()
        N = w.shape[0]
        Np = N + (-N) % 4
        skip = layer in net.skip_in
        if w.shape[1] != prev_n + (d_in if skip else 0):
            raise RuntimeError("ImplicitNet: layer %d expects %d inputs, the layout gives %d" % (layer, w.shape[1], prev_n + (d_in if skip else 0)))
        sc = 1.0 / np.sqrt(2) if skip else 1.0
        Wa = torch.zeros(Np, prev_np, dtype=torch.float32, device=w.device)
        Wa[:N, :prev_n] = w[:, :prev_n] * sc
        Wb = None
        if skip:
            Wb = torch.zeros(Np, Dp, dtype=torch.float32, device=w.device)
            Wb[:N, :d_in] = w[:, prev_n:] * sc
        bp = torch.zeros(Np, dtype=torch.float32, device=w.device)
        bp[:N] = b
        out.append((Wa, Wb, bp, N, Np))
        prev_n, prev_np = N, Np
    return out


class _DecoderVG(torch.autograd.Function):
    """(pred, g) = (net(a), d sum(net(a)) / d a) of a FROZEN softplus decoder (IGR/network.py:8-17, :20-92) as ONE autograd node whose
    backward is the hand-written double backward w.r.t. the input `a` (train_Point2Cyl.py:608-648 differentiates the eikonal / SALD terms
    through `gradient(a, pred)`).  Composed from autograd Functions (the route a trainable decoder still takes) every pre-activation
    receives two gradients that autograd sums with a 3-stream pass over 0.5 GB, and the 258 / 254-wide tensors are sliced, padded and
    concatenated around the products; here the four sweeps are written out and those sums ride in the products' epilogues:
      F   z_0 = a W_0^T + b_0, z_l = softplus(z_{l-1}) W_l^T + b_l (a skip layer as two products: [h | a] W^T = h Wa^T + a Wb^T)
      R1  e_L = 1, e_{l-1} = (e_l W_l) * s_{l-1},  s = sigmoid(beta z);  da = e_0 W_0 + (e_skip Wb)          -> g = da[:, -2:] (only those
          columns are formed: the reference's gradient() keeps nothing else, IGR/network.py:17)
      T   (given ga = dL/dg)  E_0 = ga W_0^T, (t_l, q_l) = (E_l s_l, E_l e_l beta (1 - s_l)),  E_{l+1} = t_l W_{l+1}^T (+ ga Wb^T at the skip)
      B   Z_L = dL/dpred,  Z_{l-1} = (Z_l W_l) * s_{l-1} + q_{l-1},  dL/da = Z_0 W_0 + Z_skip Wb
    T is the forward-mode derivative of the network in direction ga; q collects what the first backward contributes to each z."""

    @staticmethod
    def forward(ctx, a, net, d_in):
        lay, ex = _decoder_layout_cached(net, d_in)
        beta, thr = float(net.beta), 20.0
        L = len(lay)
        M = a.shape[0]
        dev = a.device
        Dp = d_in + (-d_in) % 4
        ap = a if a.shape[1] == Dp else F.pad(a, (0, Dp - a.shape[1]))
        ap = _c(ap.detach())
        ctx.padded_in = a.shape[1] == Dp
        # the last layer has ONE output unit (IGR/network.py:38: dims + [1]): its products are a dot product / outer products with one weight
        # row and ride on the activation passes (csrc/softplus.hip, row-structured kernels) instead of [M x 4]-padded GEMMs over 0.5 GB operands
        one_out = L >= 2 and lay[-1][3] == 1 and lay[-1][1] is None
        w_last = ex["w_last"] if one_out else None
        zs, x = [], ap
        for l, (Wa, Wb, b, N, Np) in enumerate(lay):
            if one_out and l == L - 1:
                pred = torch.empty(M, 1, dtype=torch.float32, device=dev)
                call("p2c_softplus_dot_f32", ptr(x), ptr(w_last), ptr(b), ptr(pred), M, x.shape[1], beta, thr, stream(), nbytes=4.0 * x.numel())
                zs.append(None)
                break
            h = x
            if l > 0:
                h = torch.empty_like(x)
                call("p2c_softplus_fwd_f32", ptr(x), ptr(h), x.numel(), beta, thr, stream(), nbytes=8.0 * x.numel())
            z = _prod_nt(h, Wa, b)
            if Wb is not None:
                z = _prod_nt(ap, Wb, None, add=z)
            zs.append(z)
            x = z
        if not one_out:
            pred = zs[-1][:, :lay[-1][3]].contiguous() if lay[-1][4] != lay[-1][3] else zs[-1]
        # R1: e_{L-1} = ones (only the real output columns)
        es = [None] * L
        skip_term = None
        g0, w0g = ex["g0"], ex["w0g"]                               # the 4-aligned column block that holds the two point columns of the input
        if one_out:
            e = torch.empty_like(zs[L - 2])
            call("p2c_softplus_row_bwd_f32", None, 0, ptr(w_last), ptr(zs[L - 2]), None, ptr(e), M, e.shape[1], beta, thr, stream(), nbytes=8.0 * e.numel())
            es[L - 2] = e
            top = L - 2
        else:
            e = torch.zeros(M, lay[-1][4], dtype=torch.float32, device=dev)
            e[:, :lay[-1][3]] = 1.0
            top = L - 1
        for l in range(top, 0, -1):
            Wa, Wb, _, _, _ = lay[l]
            if Wb is not None:
                wbg = ex["wbg"][l]
                skip_term = _prod_nn(e, wbg) if skip_term is None else _prod_nn(e, wbg, add=skip_term)
            e = _prod_nn(e, Wa, zs[l - 1], beta, thr)
            es[l - 1] = e
        da = _prod_nn(e, w0g, add=skip_term)                      # only the g0..g1 columns of da = e_0 W_0 + e_skip Wb are ever used
        g = da[:, d_in - 2 - g0:d_in - g0]
        ctx.lay, ctx.ex, ctx.bt, ctx.d_in, ctx.Dp, ctx.g0, ctx.one_out = lay, ex, (beta, thr), d_in, Dp, g0, one_out
        ctx.save_for_backward(ap, *zs[:L - 1], *es[:L - 1])
        ctx.set_materialize_grads(False)
        return pred, g

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gpred, gg):
        lay, ex, (beta, thr), d_in, Dp, g0, one_out = ctx.lay, ctx.ex, ctx.bt, ctx.d_in, ctx.Dp, ctx.g0, ctx.one_out
        L = len(lay)
        saved = ctx.saved_tensors
        ap, zs, es = saved[0], saved[1:L], saved[L:]
        M = ap.shape[0]
        dev = ap.device
        qs = [None] * (L - 1)
        if gg is not None:
            # T: forward-mode sweep in direction ga = dL/dg, which lives in the input's two point columns only: the products with W_0 (and with
            # a skip layer's Wb) are rank-2 updates with those two weight columns, applied inside the activation pass of the layer
            gg = _c(gg)
            t = None
            for l in range(L - 1):
                Wa, Wb = lay[l][0], lay[l][1]
                Ein = None if l == 0 else _prod_nt(t, Wa)
                t, q = torch.empty_like(zs[l]), torch.empty_like(zs[l])
                if ex["cols"][l] is not None:
                    wa_, wb_ = ex["cols"][l]
                    call("p2c_softplus_sig_bwd_rank2_f32", ptr(gg), gg.stride(0), ptr(wa_), ptr(wb_), ptr(Ein), ptr(es[l]), ptr(zs[l]), ptr(t), ptr(q), M,
                         zs[l].shape[1], beta, thr, stream(), nbytes=(20.0 if Ein is not None else 16.0) * zs[l].numel())
                else:
                    call("p2c_softplus_sig_bwd_f32", ptr(Ein), ptr(es[l]), ptr(zs[l]), ptr(t), ptr(q), zs[l].numel(), beta, thr, stream(),
                         nbytes=20.0 * zs[l].numel())
                qs[l] = q
        # B: reverse sweep of the total gradients of the pre-activations
        Z = None
        top = L - 1
        if one_out:
            top = L - 2
            if gpred is not None:
                gp = _c(gpred)
                Z = torch.empty_like(zs[L - 2])
                call("p2c_softplus_row_bwd_f32", ptr(gp), gp.stride(0), ptr(ex["w_last"]), ptr(zs[L - 2]), ptr(qs[L - 2]), ptr(Z), M,
                     Z.shape[1], beta, thr, stream(), nbytes=(12.0 if qs[L - 2] is not None else 8.0) * Z.numel())
            else:
                Z = qs[L - 2]
        elif gpred is not None:
            Np_last, N_last = lay[-1][4], lay[-1][3]
            Z = F.pad(_c(gpred), (0, Np_last - N_last)) if Np_last != N_last else _c(gpred)
        skip_term = None
        for l in range(top, 0, -1):
            Wa, Wb = lay[l][0], lay[l][1]
            q = qs[l - 1]
            if Z is None:                        # no gradient arrives from above: this pre-activation's total is its q alone
                Z = q
                continue
            if Wb is not None:
                skip_term = _prod_nn(Z, Wb) if skip_term is None else _prod_nn(Z, Wb, add=skip_term)
            Z = _prod_nn(Z, Wa, zs[l - 1], beta, thr, add=q)
        if Z is None:
            return None, None, None
        dA = _prod_nn(Z, lay[0][0], add=skip_term)
        return (dA[:, :d_in] if (Dp != d_in and not ctx.padded_in) else dA), None, None


def decoder_value_and_grad_applicable(net):
    """The one-node route needs a frozen decoder with softplus activations (the with-sketch trainer's: implicit_net.eval(), requires_grad False)."""
    return net.beta > 0 and not any(p.requires_grad for p in net.parameters())


def decoder_value_and_grad(net, a, d_in=None):
    """-> (net(a) [M,1], gradient(a, net(a)) = d sum(net(a)) / d a [:, -2:] [M,2]) with the graph kept through ONE node (see _DecoderVG);
    `a` [M, d_in] on the device - or, with d_in given, [M, pad4(d_in)] with zero pad columns (add_latent(..., pad=True)): no padding copy."""
    if not a.is_cuda:
        raise RuntimeError("point2cyl_amd.implicit.decoder_value_and_grad runs on the HIP device only (got %s); there is no CPU path" % a.device)
    d_in = a.shape[1] if d_in is None else int(d_in)
    if a.shape[1] not in (d_in, d_in + (-d_in) % 4):
        raise ValueError("decoder_value_and_grad: input of width %d for d_in = %d" % (a.shape[1], d_in))
    return _DecoderVG.apply(a, net, d_in)
