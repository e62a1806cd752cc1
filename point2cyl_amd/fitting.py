"""Differentiable extrusion-cylinder fitting (mirror of the reference's data_utils.py hot-path API).

estimate_extrusion_axis / estimate_extrusion_centers / get_extrusion_extents keep the reference's
signatures (data_utils.py:99, :253, :1650); the work is done by fit.hip through point2cyl_amd.ops.
"""
import numpy as np
import torch

from . import ops

TORCH_PI = torch.acos(torch.zeros(1)).item() * 2


def add_noise(batch_xyz, batch_normal, sigma=0.01):
    """data_utils.py:84-96: p + N(0,sigma) * normal with NumPy's RNG, float64 result (host side)."""
    B, N, _ = batch_xyz.shape
    noise = np.random.normal(0.0, sigma, (B, N))
    return batch_xyz + torch.tensor(noise).unsqueeze(-1) * batch_normal


def add_noise_on_device(batch_xyz, batch_normal, sigma=0.01):
    """add_noise for clouds that are on the device already: the SAME draws (np.random.normal on the host generator, data_utils.py:84-96) and the
    same float64 arithmetic (one multiply, one add: IEEE, bit-identical to the host's), but only the (B, N) noise crosses the bus and the
    (B, N, 3) float64 intermediates never exist on the host (1.5 ms of the evaluation loader's 5.3 ms per batch)."""
    B, N, _ = batch_xyz.shape
    noise = torch.from_numpy(np.random.normal(0.0, sigma, (B, N))).to(batch_xyz.device)
    return batch_xyz.double() + noise.unsqueeze(-1) * batch_normal.double()


def estimate_extrusion_axis(X, W_barrel, W_base, gt_bb_labels, gt_extrusion_instances, normalize=False, return_float64=False):
    """data_utils.py:99-177 -> E_AX (B,K,3): eigenvector of the smallest eigenvalue of B^T B - C^T C.
    The sign is arbitrary in the reference (LAPACK); here the largest component is positive.  All
    consumers take abs(dot) (losses.py:130, :149).
    return_float64 (not in the reference): also return the unit vector as the kernel's fp64 eigen-solve left it, (B,K,3) float64,
    detached - what eval.py's axis-angle metric (:398-405) is evaluated on here (an acos next to its clamp: the 3e-8 of an fp32-stored
    unit vector alone moves a 0.1 degree angle by more than the 1e-4 parity bar)."""
    if not return_float64:
        return ops.extrusion_axis(X, W_barrel, W_base, gt_bb_labels, gt_extrusion_instances, normalize)
    B, _, K = W_barrel.shape
    a64 = torch.empty(B, K, 3, dtype=torch.float64, device=X.device)
    E = ops.extrusion_axis(X, W_barrel, W_base, gt_bb_labels, gt_extrusion_instances, normalize, axis64=a64)
    return E, a64


def estimate_extrusion_centers(W, pcs):
    """data_utils.py:253-266 -> (B,K,3) = (1/N) sum_n W[b,n,k] p[b,n]."""
    return ops.extrusion_centers(W, pcs)


def segment_centroids(EA_W, pcs):
    """eval.py:409-436: per-segment mean of the points with EA_W == 1 -> (centroids (B,K,3), found (B,K)); a segment with <= 1 such point
    is "not found" (zeros).  EA_W may be multi-hot: eval.py's --use_gt_bb branch gathers the hard encoding by matching_indices, whose
    unmatched slots all point at column 0 (:382), so a point can carry a 1 in several columns.  The per-segment coordinate sums are the
    weighted-centre kernel on the 0/1 indicator (fit.hip: (1/N) sum_n w p), rescaled by N / count."""
    N = EA_W.shape[1]
    ind = (EA_W == 1).to(torch.float32)
    cnt = ind.sum(dim=1)                                            # (B,K)
    cen = ops.extrusion_centers(ind, pcs) * (float(N) / cnt.clamp(min=1.0)).unsqueeze(-1)
    found = (cnt > 1).to(torch.float32)
    return cen * found.unsqueeze(-1), found


def barrel_counts(seg_label, bb_labels, K):
    """-> [K][B] nested list: the number of barrel points (bb == 0) of every segment of every cloud - what decides which draws the
    reference makes.  On host tensors this costs no device sync (the evaluation loop calls it before the upload)."""
    idx = seg_label.clamp(min=0).to(torch.int64)
    sel = ((seg_label >= 0) & (seg_label < K) & (bb_labels == 0)).to(torch.int64)
    return torch.zeros(seg_label.shape[0], K, dtype=torch.int64, device=seg_label.device).scatter_add_(1, idx.clamp(max=K - 1), sel).t().tolist()


def barrel_counts_tensor(seg_label, bb_labels, K):
    """barrel_counts as a (B, K) int64 tensor on the labels' device, no host round trip (the evaluation loop's loader thread: the counts
    only feed the on-device extent draws there)."""
    idx = seg_label.clamp(min=0, max=K - 1)
    sel = ((seg_label >= 0) & (seg_label < K) & (bb_labels == 0)).to(torch.int64)
    return torch.zeros(seg_label.shape[0], K, dtype=torch.int64, device=seg_label.device).scatter_add_(1, idx, sel)


def get_extrusion_extents(P, seg_label, bb_labels, extrusion_axes, extrusion_centers, num_points_to_sample=1024, rand_idx=None, counts=None):
    """data_utils.py:1650-1730 -> extents (K,B,2), found_centers_mask (B,K).
    The reference samples barrel points with torch.randint on the CPU generator inside a K x B loop (:1696);
    the same draws are made here in the same order (k outer, b inner, only where > 1 barrel point exists),
    or taken from `rand_idx` (B,K,S) when given.  counts (not in the reference): barrel_counts(seg_label, bb_labels, K) when the caller
    has it already - the draws then need no device->host sync."""
    B, K, _ = extrusion_axes.shape
    S = num_points_to_sample
    if rand_idx is None:
        rand_idx = _barrel_draws(seg_label, bb_labels, K, S, device=P.device, counts=counts)
    return ops.extrusion_extents(P, seg_label, bb_labels, extrusion_axes, extrusion_centers, rand_idx.to(P.device))


def fit_cylinders(X, W_barrel, W_base, gt_bb_labels, seg_label, P, num_points_to_sample=1024, rand_idx=None, normalize=False,
                  return_float64=False, validate=True, K=None):
    """The fitting-only chain on pre-segmented clouds (BASELINE configs[3]): estimate_extrusion_axis (eval.py:397) -> hard centroids
    (eval.py:409-436) -> get_extrusion_extents on the fitted axes / centroids (data_utils.py:1650-1730), one pass per cloud where the
    shape allows (ops.fit_fused), the three ops otherwise.  -> axes (B,K,3), centroids (B,K,3), centroid found (B,K), extents (K,B,2),
    extent found (B,K) [, axes in float64 with return_float64, see estimate_extrusion_axis].  Forward only.
    validate: the kernels read the labels' low words; labels outside [-1, K) / {0, 1} raise here instead of aliasing (one device->host
    sync; a caller that loops over the same labels passes validate=False after its first call).
    W_barrel = W_base = None (then K must be given): the memberships are the one-hot encodings IMPLIED by the labels - W_barrel[b,n,k] =
    [seg == k and bb == 0], W_base = [seg == k and bb == 1], what pre-segmented clouds (eval.py --use_gt_segmentation --use_gt_bb, configs[3])
    hand in - and are not read at all: 40 bytes per point instead of 104.  Same arithmetic, another summation order (1e-6)."""
    hard = W_barrel is None and W_base is None
    if hard:
        if K is None:
            raise ValueError("fit_cylinders: pass K when the memberships are implied by the labels (W_barrel = W_base = None)")
        B, N = seg_label.shape
    else:
        B, N, K = W_barrel.shape
    S = num_points_to_sample if rand_idx is None else rand_idx.shape[2]
    if validate:
        ops.check_labels(seg_label, K)
        ops.check_labels(gt_bb_labels, 2)
    if rand_idx is None:
        rand_idx = _barrel_draws(seg_label, gt_bb_labels, K, S, device=P.device)
    rand_idx = rand_idx.to(P.device)
    if ops.fit_fused_supported(N, K, S):
        return ops.fit_fused(X, W_barrel, W_base, gt_bb_labels, seg_label, P, rand_idx, normalize=normalize, axes64=return_float64, K=K)
    if hard:
        onehot = torch.nn.functional.one_hot(seg_label.clamp(min=0), K).float() * (seg_label >= 0).unsqueeze(-1)
        W_barrel, W_base = onehot * (gt_bb_labels == 0).unsqueeze(-1), onehot * (gt_bb_labels == 1).unsqueeze(-1)
    with torch.no_grad():
        axes = estimate_extrusion_axis(X, W_barrel, W_base, gt_bb_labels, seg_label, normalize=normalize, return_float64=return_float64)
        axes, a64 = axes if return_float64 else (axes, None)
        cen, cfound = ops.segment_centroids(P, seg_label, K)
        ext, found = ops.extrusion_extents(P, seg_label, gt_bb_labels, axes, cen, rand_idx)
    return (axes, cen, cfound, ext, found, a64) if return_float64 else (axes, cen, cfound, ext, found)


def barrel_draws_on_device(seg_label, bb_labels, K, S):
    """The sampling draws of data_utils.py:1064 / :1696 made ON THE DEVICE: uniform integers in [0, n_barrel(b,k)) for every (b, k), from
    torch's device generator - no device->host sync (the reference's CPU draws need the K x B barrel counts on the host), so a step that uses
    them can be captured into a HIP graph.  Not the reference's random stream (its CPU generator draws only where a segment has > 1 barrel
    point); the projection ignores the draws of segments it does not find, and tests that need the reference's draws pass rand_idx."""
    barrel = (seg_label.unsqueeze(-1) == torch.arange(K, device=seg_label.device)) & (bb_labels == 0).unsqueeze(-1)
    counts = barrel.sum(dim=1)                                                             # (B,K) on the device
    u = torch.rand(seg_label.shape[0], K, S, device=seg_label.device)
    return torch.minimum((u * counts.unsqueeze(-1)).long(), (counts - 1).clamp_min(0).unsqueeze(-1))


class _DrawRing:
    """Two pinned (B,K,S) int64 host buffers per shape, written by the CPU generator and copied to the device without blocking: a fresh
    4 MB host tensor per batch costs 6 - 20 ms of page faults on the virtualised hosts this runs on, more than the draws themselves."""
    rings = {}

    def __init__(self, shape):
        self.bufs = [torch.zeros(shape, dtype=torch.int64).pin_memory() for _ in range(2)]
        self.rows = [[r.unbind(0) for r in b.unbind(0)] for b in self.bufs]
        self.copied, self.i = [None, None], 0

    @classmethod
    def take(cls, shape):
        import threading
        key = (tuple(shape), threading.get_ident())       # (the evaluation loop draws in its loader thread: one ring per drawing thread)
        ring = cls.rings.get(key)
        if ring is None:
            ring = cls.rings[key] = cls(shape)
        i = ring.i
        ring.i ^= 1
        if ring.copied[i] is not None:
            ring.copied[i].synchronize()                 # the copy that last read this buffer (two batches ago) is done
        return ring, i


def _barrel_draws(seg_label, bb_labels, K, S, device=None, counts=None, out=None, generator=None):
    """The reference's torch.randint draws for its K x B sampling loops (data_utils.py:1064, :1696): k outer, b inner, only
    where the segment has > 1 barrel point in the batch and in the cloud, on the CPU generator.  -> (B,K,S) int64 on the host, or, with
    `device`, on that device (drawn into a recycled pinned buffer, copied on the current stream - into `out`, a (B,K,S) int64 device tensor,
    when given: a captured graph reads its draws from a fixed address).  generator: another CPU generator than the default one (the
    evaluation loop's loader thread draws from a private one, seeded from the default generator, so that the two threads' draws never interleave)."""
    B = seg_label.shape[0]
    if counts is None:
        barrel = (seg_label.unsqueeze(-1) == torch.arange(K, device=seg_label.device)) & (bb_labels == 0).unsqueeze(-1)
        counts = barrel.sum(dim=1).t().tolist()          # [K][B]: ONE sync; the reference syncs K*B times here
    if device is None or torch.device(device).type != "cuda":
        rand_idx = torch.zeros(B, K, S, dtype=torch.int64)
        rows, ring = [r.unbind(0) for r in rand_idx.unbind(0)], None
    else:
        ring, i = _DrawRing.take((B, K, S))
        rand_idx, rows = ring.bufs[i], ring.rows[i]
    for k in range(K):
        ck = counts[k]
        none = sum(ck) <= 1
        for b in range(B):
            if ck[b] > 1 and not none:
                torch.randint(0, ck[b], (S,), out=rows[b][k], generator=generator)      # (the draw itself: same generator, same order, same values)
            elif ring is not None:
                rows[b][k].zero_()
    if ring is None:
        res = rand_idx if device is None else rand_idx.to(device)
        return res if out is None else out.copy_(res)
    out = rand_idx.to(device, non_blocking=True) if out is None else out.copy_(rand_idx, non_blocking=True)
    ring.copied[i] = torch.cuda.Event()
    ring.copied[i].record()
    return out


def sketch_implicit_projection(P, X, seg_label, bb_labels, extrusion_axes, extrusion_centers, num_points_to_sample=1024, rand_idx=None):
    """data_utils.py:1014-1146 -> P_projected (K,B,S,2), X_projected (K,B,S,2), scales (K,B).  The barrel samples are drawn
    like the reference draws them (same generator, same order) unless `rand_idx` (B,K,S) is given."""
    return sketch_implicit_projection2(P, X, seg_label, bb_labels, extrusion_axes, extrusion_centers, num_points_to_sample, rand_idx)[:3]


def sketch_implicit_projection2(P, X, seg_label, bb_labels, extrusion_axes, extrusion_centers, num_points_to_sample=1024, rand_idx=None):
    """data_utils.py:1149-1282: the same + found_centers_mask (B,K)."""
    K, S = extrusion_axes.shape[1], num_points_to_sample
    if rand_idx is None:
        rand_idx = _barrel_draws(seg_label, bb_labels, K, S, device=P.device)
    return ops.sketch_projection(P, X, seg_label, bb_labels, extrusion_axes, extrusion_centers, rand_idx.to(P.device), S)


def sketch_implicit_projection3(P, X, seg_label, bb_labels, extrusion_axes, extrusion_centers, num_points_to_sample=8192):
    """data_utils.py:1284-1417: every point of the cloud for every segment (labels unused there as well: its mask is all ones,
    :1294), no sampling; num_points_to_sample must equal N as in the reference (its buffers are that size)."""
    if num_points_to_sample != P.shape[1]:
        raise ValueError("sketch_implicit_projection3 takes all N points: num_points_to_sample must be N (data_utils.py:1306-1336)")
    return ops.sketch_projection(P, X, None, None, extrusion_axes, extrusion_centers, None, P.shape[1], all_points=True)
