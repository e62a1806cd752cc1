"""Synthetic "Fusion-Gallery-shaped" clouds.

There is no dataset on the box, so benches and tests draw seeded unions of extrusion cylinders and
return them in the tuple layout the reference's dataset yields (dataloader.py:95-96, center=True):
pcs (N,3) f32, normals (N,3) f32, extrusion_labels (N) i64, bb_labels (N) i64 {0 barrel, 1 base},
per-point axes (N,3), per-point distances (N), extrusion_axes (K,3), extrusion_distances (K),
extrusion_centers (K,3); rows >= k are zero.  Host-side numpy only; nothing here is on the GPU path.
"""
import math

import numpy as np
import torch


def _unit(v):
    return v / np.linalg.norm(v, axis=-1, keepdims=True)


def _frame(a):
    t = np.array([1.0, 0.0, 0.0]) if abs(a[0]) < 0.9 else np.array([0.0, 1.0, 0.0])
    u = _unit(np.cross(a, t))
    return u, np.cross(a, u)


def make_shape(rng, num_point=8192, K=8, k=None):
    """One cloud.  rng: np.random.Generator."""
    k = int(rng.integers(1, K + 1)) if k is None else k
    cyl = []
    for _ in range(k):
        a = _unit(rng.normal(size=3))
        c = rng.uniform(-0.4, 0.4, size=3)
        h = rng.uniform(0.1, 0.8)
        if rng.random() < 0.5:
            verts = None
            r = rng.uniform(0.1, 0.5)
            perim, cap = 2 * math.pi * r, math.pi * r * r
        else:
            m = int(rng.integers(3, 9))
            ang = np.sort(rng.uniform(0, 2 * math.pi, size=m))
            ang = np.linspace(0, 2 * math.pi, m, endpoint=False) + 0.5 * (ang - ang.mean()) / m
            r = rng.uniform(0.1, 0.5)
            verts = r * np.stack([np.cos(ang), np.sin(ang)], -1)
            nxt = np.roll(verts, -1, 0)
            perim = np.linalg.norm(nxt - verts, axis=1).sum()
            cap = 0.5 * np.abs(verts[:, 0] * nxt[:, 1] - verts[:, 1] * nxt[:, 0]).sum()
        cyl.append(dict(a=a, c=c, h=h, r=r, verts=verts, area_barrel=perim * h, area_cap=cap))
    areas = np.array([[c["area_barrel"], c["area_cap"], c["area_cap"]] for c in cyl]).reshape(-1)
    counts = rng.multinomial(num_point, areas / areas.sum()).reshape(k, 3)
    # every present segment keeps at least two barrel points and two base points
    for i in range(k):
        for j in (0, 1):
            while counts[i, j] < 2:
                big = np.unravel_index(np.argmax(counts), counts.shape)
                counts[big] -= 1
                counts[i, j] += 1
    P, Nrm, seg, bb = [], [], [], []
    for i, cy in enumerate(cyl):
        u, v = _frame(cy["a"])
        for part in range(3):
            n = int(counts[i, part])
            if n == 0:
                continue
            if part == 0:   # barrel
                t = rng.uniform(-0.5, 0.5, size=n) * cy["h"]
                if cy["verts"] is None:
                    th = rng.uniform(0, 2 * math.pi, size=n)
                    xy = cy["r"] * np.stack([np.cos(th), np.sin(th)], -1)
                    nxy = np.stack([np.cos(th), np.sin(th)], -1)
                else:
                    vt = cy["verts"]
                    nx = np.roll(vt, -1, 0)
                    ln = np.linalg.norm(nx - vt, axis=1)
                    e = rng.choice(len(vt), size=n, p=ln / ln.sum())
                    s = rng.random(n)[:, None]
                    xy = vt[e] * (1 - s) + nx[e] * s
                    d = (nx - vt)[e]
                    nxy = _unit(np.stack([d[:, 1], -d[:, 0]], -1))
                p = cy["c"] + xy[:, :1] * u + xy[:, 1:] * v + t[:, None] * cy["a"]
                nr = nxy[:, :1] * u + nxy[:, 1:] * v
                lab = 0
            else:           # caps
                sign = 1.0 if part == 1 else -1.0
                if cy["verts"] is None:
                    rr = cy["r"] * np.sqrt(rng.random(n))
                    th = rng.uniform(0, 2 * math.pi, size=n)
                    xy = np.stack([rr * np.cos(th), rr * np.sin(th)], -1)
                else:
                    vt = cy["verts"]
                    nx = np.roll(vt, -1, 0)
                    ta = 0.5 * np.abs(vt[:, 0] * nx[:, 1] - vt[:, 1] * nx[:, 0])
                    e = rng.choice(len(vt), size=n, p=ta / ta.sum())
                    r1, r2 = np.sqrt(rng.random(n))[:, None], rng.random(n)[:, None]
                    xy = (1 - r1) * 0 + r1 * (1 - r2) * vt[e] + r1 * r2 * nx[e]
                p = cy["c"] + xy[:, :1] * u + xy[:, 1:] * v + sign * 0.5 * cy["h"] * cy["a"]
                nr = np.broadcast_to(sign * cy["a"], p.shape)
                lab = 1
            P.append(p)
            Nrm.append(nr)
            seg.append(np.full(n, i))
            bb.append(np.full(n, lab))
    P, Nrm = np.concatenate(P), np.concatenate(Nrm)
    seg, bb = np.concatenate(seg), np.concatenate(bb)
    perm = rng.permutation(num_point)
    P, Nrm, seg, bb = P[perm], Nrm[perm], seg[perm], bb[perm]
    # centre + scale to unit max-norm (utils.py:938-950 semantics)
    ctr = P.mean(0)
    P = P - ctr
    scale = np.linalg.norm(P, axis=1).max()
    P = P / scale
    axes = np.zeros((K, 3))
    dist = np.zeros(K)
    cen = np.zeros((K, 3))
    for i, cy in enumerate(cyl):
        axes[i], dist[i], cen[i] = cy["a"], cy["h"] / scale, (cy["c"] - ctr) / scale
    f = np.float32
    return (P.astype(f), Nrm.astype(f), seg.astype(np.int64), bb.astype(np.int64), axes[seg].astype(f),
            dist[seg].astype(f), axes.astype(f), dist.astype(f), cen.astype(f))


def make_batch(batch_size, num_point=8192, K=8, seed=1234, k=None):
    """Stacked torch tensors (CPU) in the same order as one collated reference batch."""
    items = [make_shape(np.random.default_rng(seed + i), num_point, K, k) for i in range(batch_size)]
    return tuple(torch.from_numpy(np.stack([it[j] for it in items])) for j in range(9))


class SyntheticExtrusionDataset(torch.utils.data.Dataset):
    """Stand-in for AutodeskDataset_h5 (dataloader.py:15-127): same 9-tuple per item."""

    def __init__(self, n_shapes, num_point=8192, K=8, seed=1234):
        self.n, self.num_point, self.K, self.seed = n_shapes, num_point, K, seed

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        return make_shape(np.random.default_rng(self.seed + i), self.num_point, self.K)


def make_fitting_inputs(n_clouds, N=8192, K=8, seed=4321, distinct=64, noise_deg=2.0):
    """BASELINE configs[3]'s workload: pre-segmented synthetic cylinders for the fitting-only path of eval.py.
    `distinct` different clouds tiled to n_clouds; X = ground-truth normals turned by N(0, noise_deg) about a random perpendicular axis,
    W_barrel / W_base one-hot from the labels.  -> pcs, X, seg, bb, gt axes (n,K,3), W_barrel, W_base, one-hot (all CPU, fp32 / int64)."""
    pcs, nrm, seg, bb, _, _, axes, _, cen = make_batch(min(distinct, n_clouds), N, K, seed=seed)
    pcs, nrm, axes = pcs.float(), nrm.float(), axes.float()
    reps = (n_clouds + pcs.shape[0] - 1) // pcs.shape[0]
    tile = lambda t: t.repeat((reps,) + (1,) * (t.dim() - 1))[:n_clouds].contiguous()
    pcs, nrm, seg, bb, axes = tile(pcs), tile(nrm), tile(seg), tile(bb), tile(axes)
    g = torch.Generator().manual_seed(seed + 1)
    r = torch.randn(nrm.shape, generator=g)
    perp = r - (r * nrm).sum(-1, keepdim=True) * nrm
    perp = perp / perp.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    ang = torch.randn(nrm.shape[:2], generator=g).unsqueeze(-1) * (noise_deg * math.pi / 180.0)
    X = torch.cos(ang) * nrm + torch.sin(ang) * perp
    onehot = torch.nn.functional.one_hot(seg.clamp_min(0), K).float() * (seg >= 0).unsqueeze(-1)
    Wb = onehot * (bb == 0).unsqueeze(-1)
    Wc = onehot * (bb == 1).unsqueeze(-1)
    return pcs, X.float(), seg, bb, axes, Wb, Wc, onehot


def axis_angle_error_deg64(E, gt_axes, seg, K):
    """eval.py:398-405 in float64: masked mean over the segments that exist of acos_safe(|a . a_gt|) in degrees (losses.py:123, :146-159)."""
    dot = (E.double() * gt_axes.double()).sum(-1).abs().clamp(min=-1.0 + 1e-6, max=1.0 - 1e-6)
    deg = torch.acos(dot) * 180.0 / math.pi
    present = (torch.nn.functional.one_hot(seg.clamp_min(0), K) * (seg >= 0).unsqueeze(-1)).sum(1) > 0
    return float((deg * present).sum() / present.sum())
