"""CPU restatement of Point2Cyl's hot path in plain PyTorch fp32 (TEST INFRASTRUCTURE).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module, and
only as the checker / the timed CPU baseline ("port").  The product (point2cyl_amd/) never does.

It re-states, in our own functional style, the stock-op sequences of the reference
(citations are file:line into the upstream repo).  It is pinned against the reference itself
by tests/golden/*.npz (made by oracle/make_golden.py, which imports /root/reference) -- see
tests/test_oracle_golden.py.

Two geometry back-ends:
  geom="torch": the literal op sequence of the reference (BLAS matmul + full sort), used to time
                the CPU baseline and to validate against the imported reference on the same box.
  geom="c":     oracle/p2c_oracle.c with the reference's rounding order spelled out (fmaf), which is
                host-independent; this is the checker the GPU parity tests use.
"""
import math

import torch
import torch.nn.functional as F

G_ZERO_TOL = 1.0e-6          # global_variables.py:15
TORCH_PI = torch.acos(torch.zeros(1)).item() * 2   # losses.py:17

SA_CFG = (  # pointnet_extrusion.py:21-23
    dict(name="sa1", npoint=512, radius=0.2, nsample=64, mlp=(64, 64, 128)),
    dict(name="sa2", npoint=128, radius=0.4, nsample=64, mlp=(128, 128, 256)),
    dict(name="sa3", npoint=None, radius=None, nsample=None, mlp=(256, 512, 1024)),
)
FP_CFG = (  # pointnet_extrusion.py:25-27
    dict(name="fp3", cin=1280, mlp=(256, 256)),
    dict(name="fp2", cin=384, mlp=(256, 128)),
    dict(name="fp1", cin=128, mlp=(128, 128, 128)),
)


# --------------------------------------------------------------------------- geometry
def square_distance(src, dst):
    """pointnet_util.py:19-40: ((-2*src.dst^T) + |src|^2) + |dst|^2, in that order."""
    d = torch.matmul(src, dst.transpose(1, 2)) * -2
    d = d + (src * src).sum(-1).unsqueeze(2)
    d = d + (dst * dst).sum(-1).unsqueeze(1)
    return d


def gather_rows(points, idx):
    """pointnet_util.py:43-60 index_points: points (B,N,C), idx (B,...) -> (B,...,C)."""
    B = points.shape[0]
    flat = idx.reshape(B, -1)
    out = torch.gather(points, 1, flat.unsqueeze(-1).expand(-1, -1, points.shape[-1]))
    return out.reshape(*idx.shape, points.shape[-1])


def farthest_point_sample(xyz, npoint, start, geom="c"):
    """pointnet_util.py:63-84.  `start` = the CPU randint draw of :75."""
    if geom == "c":
        from . import cref
        return torch.from_numpy(cref.fps(xyz.detach().numpy(), start.numpy(), npoint))
    B, N, _ = xyz.shape
    running = torch.full((B, N), 1e10)
    far = start.clone()
    rows = torch.arange(B)
    picked = []
    for _ in range(npoint):
        picked.append(far)
        c = xyz[rows, far].unsqueeze(1)
        d = ((xyz - c) ** 2).sum(-1)
        running = torch.where(d < running, d, running)
        far = running.max(-1)[1]
    return torch.stack(picked, 1)


def query_ball_point(radius, nsample, xyz, new_xyz, geom="c"):
    """pointnet_util.py:87-107."""
    if geom == "c":
        from . import cref
        return torch.from_numpy(cref.ball_query(radius, nsample, xyz.detach().numpy(), new_xyz.detach().numpy()))
    B, N, _ = xyz.shape
    S = new_xyz.shape[1]
    ids = torch.arange(N).expand(B, S, N).clone()
    ids[square_distance(new_xyz, xyz) > radius ** 2] = N
    ids = ids.sort(-1)[0][:, :, :nsample]
    first = ids[:, :, :1].expand(-1, -1, nsample)
    return torch.where(ids == N, first, ids)


def three_nn(xyz1, xyz2, geom="c"):
    """pointnet_util.py:301-303 -> (dists (B,N,3), idx (B,N,3))."""
    if geom == "c":
        from . import cref
        d, i = cref.three_nn(xyz1.detach().numpy(), xyz2.detach().numpy())
        return torch.from_numpy(d), torch.from_numpy(i)
    d, i = square_distance(xyz1, xyz2).sort(-1)
    return d[:, :, :3], i[:, :, :3]


# --------------------------------------------------------------------------- backbone
def make_state_dict(output_sizes=(3,), normal_channel=False, seed=None):
    """Same registration order and default torch init as pointnet_extrusion.py:9-34 /
    pointnet_util.py:167-179, 271-280, so the same torch.manual_seed gives the same tensors."""
    import torch.nn as nn
    if seed is not None:
        torch.manual_seed(seed)
    add = 3 if normal_channel else 0
    mods = {}

    def stack(prefix, cin, widths, conv, bn):
        for i, co in enumerate(widths):
            mods["%s.mlp_convs.%d" % (prefix, i)] = conv(cin, co, 1)
            mods["%s.mlp_bns.%d" % (prefix, i)] = bn(co)
            cin = co

    stack("sa1", 3 + add, SA_CFG[0]["mlp"], nn.Conv2d, nn.BatchNorm2d)
    stack("sa2", 128 + 3, SA_CFG[1]["mlp"], nn.Conv2d, nn.BatchNorm2d)
    stack("sa3", 256 + 3, SA_CFG[2]["mlp"], nn.Conv2d, nn.BatchNorm2d)
    stack("fp3", 1280, FP_CFG[0]["mlp"], nn.Conv1d, nn.BatchNorm1d)
    stack("fp2", 384, FP_CFG[1]["mlp"], nn.Conv1d, nn.BatchNorm1d)
    stack("fp1", 128 + add, FP_CFG[2]["mlp"], nn.Conv1d, nn.BatchNorm1d)
    mods["fc1"] = nn.Conv1d(128, 128, 1)
    mods["bn1"] = nn.BatchNorm1d(128)
    for i, o in enumerate(output_sizes):
        mods["fc2.%d" % i] = nn.Conv1d(128, o, 1)
    # state_dict order follows attribute registration (mlp_convs list, then mlp_bns list), while the
    # RNG is consumed in creation order (conv, bn, conv, bn, ...) as above.
    def rank(name):
        top = name.split(".")[0]
        order = ["sa1", "sa2", "sa3", "fp3", "fp2", "fp1", "fc1", "bn1", "fc2"]
        return (order.index(top), 0 if "mlp_convs" in name else 1 if "mlp_bns" in name else 2)
    sd = {}
    for name in sorted(mods, key=lambda n: (rank(n), int(n.split(".")[-1]) if "." in n else 0)):
        for k, v in mods[name].state_dict().items():
            sd["%s.%s" % (name, k)] = v
    return sd


def _mlp(sd, prefix, n_layers, x, training, momentum, conv):
    for i in range(n_layers):
        c, b = "%s.mlp_convs.%d" % (prefix, i), "%s.mlp_bns.%d" % (prefix, i)
        x = conv(x, sd[c + ".weight"], sd[c + ".bias"])
        x = F.batch_norm(x, sd[b + ".running_mean"], sd[b + ".running_var"], sd[b + ".weight"], sd[b + ".bias"],
                         training, momentum, 1e-5)
        if training:
            sd[b + ".num_batches_tracked"] += 1
        x = F.relu(x)
    return x


def set_abstraction(sd, cfg, xyz, feats, start, training, momentum, geom, forced_winners=None):
    """pointnet_util.py:181-207 (+ sample_and_group :110-143 / sample_and_group_all :146-163).
    xyz (B,N,3), feats (B,N,D) or None -> new_xyz (B,S,3), new_feats (B,S,C'), aux dict.
    forced_winners (B,S,C') int64 (test instrument, not in the reference): the max over the neighbours (:205) takes THESE rows instead of its
    own arg-max - the device path's stored winners - so that two candidates within fp32 rounding of each other send the gradient to the same
    row in both implementations; aux then carries this oracle's own winners and, per pooled entry, how far below its own maximum the forced
    row's value lies (`pool_gap` >= 0; ~1e-7 relative at a near-tie, large if a winner were really wrong)."""
    B, N, _ = xyz.shape
    aux = {}
    if cfg["npoint"] is None:
        new_xyz = torch.zeros(B, 1, 3)
        grouped = xyz.unsqueeze(1) if feats is None else torch.cat([xyz, feats], -1).unsqueeze(1)
    else:
        fps_idx = farthest_point_sample(xyz, cfg["npoint"], start, geom)
        new_xyz = gather_rows(xyz, fps_idx)
        gidx = query_ball_point(cfg["radius"], cfg["nsample"], xyz, new_xyz, geom)
        rel = gather_rows(xyz, gidx) - new_xyz.unsqueeze(2)
        grouped = rel if feats is None else torch.cat([rel, gather_rows(feats, gidx)], -1)
        aux.update(fps_idx=fps_idx, group_idx=gidx)
    x = grouped.permute(0, 3, 2, 1)                      # (B, C, nsample, S)  :200
    x = _mlp(sd, cfg["name"], len(cfg["mlp"]), x, training, momentum, F.conv2d)
    if forced_winners is not None:
        own_val, own_idx = x.max(2)                                                   # (B,C',S)
        fw = forced_winners.to(torch.int64).permute(0, 2, 1).unsqueeze(2)             # (B,C',1,S)
        pooled = x.gather(2, fw).squeeze(2)
        aux.update(own_winners=own_idx.transpose(1, 2), pool_gap=(own_val - pooled).detach().transpose(1, 2), pool_max=own_val.detach().transpose(1, 2))
        return new_xyz, pooled.transpose(1, 2), aux
    return new_xyz, x.max(2)[0].transpose(1, 2), aux     # (B,S,C')


def feature_propagation(sd, cfg, xyz1, xyz2, feats1, feats2, training, momentum, geom):
    """pointnet_util.py:281-320.  xyz1 (B,N,3) dense, xyz2 (B,S,3) sparse, feats (B,*,C) point-major."""
    B, N, _ = xyz1.shape
    S = xyz2.shape[1]
    aux = {}
    if S == 1:
        interp = feats2.expand(B, N, feats2.shape[-1])
    else:
        d, idx = three_nn(xyz1, xyz2, geom)
        rec = 1.0 / (d + 1e-8)
        w = rec / rec.sum(2, keepdim=True)
        interp = (gather_rows(feats2, idx) * w.unsqueeze(-1)).sum(2)
        aux.update(nn_idx=idx, nn_w=w)
    x = interp if feats1 is None else torch.cat([feats1, interp], -1)
    x = _mlp(sd, cfg["name"], len(cfg["mlp"]), x.transpose(1, 2), training, momentum, F.conv1d)
    return x.transpose(1, 2), aux


def backbone_forward(sd, x, fps_start, dropout_mask=None, training=True, momentum=0.1, geom="c",
                     return_aux=False, forced_winners=None):
    """pointnet_extrusion.py:37-66.  x (B,N,3[+D]); fps_start = [start_sa1 (B,), start_sa2 (B,)]
    (the two CPU randint draws); dropout_mask (B,N,128) of {0,1} or None for
    identity.  The reference's F.dropout(p=.5) is always on (:60): out = feat*mask*2.
    Returns list of (B,N,o_i) [, aux]."""
    xyz = x[:, :, :3]
    feats0 = x[:, :, 3:] if x.shape[2] > 3 else None
    fw = forced_winners or {}          # {"sa1": (B,512,128), "sa2": (B,128,256), "sa3": (B,1,1024)} (see set_abstraction)
    l1_xyz, l1, a1 = set_abstraction(sd, SA_CFG[0], xyz, feats0, fps_start[0], training, momentum, geom, fw.get("sa1"))
    l2_xyz, l2, a2 = set_abstraction(sd, SA_CFG[1], l1_xyz, l1, fps_start[1], training, momentum, geom, fw.get("sa2"))
    l3_xyz, l3, a3 = set_abstraction(sd, SA_CFG[2], l2_xyz, l2, None, training, momentum, geom, fw.get("sa3"))
    l4, _ = feature_propagation(sd, FP_CFG[0], l2_xyz, l3_xyz, l2, l3, training, momentum, geom)
    l5, a5 = feature_propagation(sd, FP_CFG[1], l1_xyz, l2_xyz, l1, l4, training, momentum, geom)
    l6, a6 = feature_propagation(sd, FP_CFG[2], xyz, l1_xyz, feats0, l5, training, momentum, geom)
    f = F.conv1d(l6.transpose(1, 2), sd["fc1.weight"], sd["fc1.bias"])
    f = F.batch_norm(f, sd["bn1.running_mean"], sd["bn1.running_var"], sd["bn1.weight"], sd["bn1.bias"],
                     training, momentum, 1e-5)
    if training:
        sd["bn1.num_batches_tracked"] += 1
    f = F.relu(f)
    pre_dropout = f.transpose(1, 2)
    if dropout_mask is not None:
        f = f * dropout_mask.transpose(1, 2) * 2.0
    outs = []
    i = 0
    while "fc2.%d.weight" % i in sd:
        outs.append(F.conv1d(f, sd["fc2.%d.weight" % i], sd["fc2.%d.bias" % i]).transpose(1, 2))
        i += 1
    if return_aux:
        aux = dict(sa1=a1, sa2=a2, sa3=a3, fp2=a5, fp1=a6, l1_xyz=l1_xyz, l2_xyz=l2_xyz, l1=l1, l2=l2, l3=l3, l4=l4,
                   l5=l5, l6=l6, pre_dropout=pre_dropout)
        return outs, aux
    return outs


# --------------------------------------------------------------------------- losses
def hungarian_matching(W_pred, I_gt):
    """losses.py:22-52 -> matching_indices (B,K) int64, mask (B,K) bool.  Uses scipy like the
    reference (cref.lsa_max is the restated solver, checked equal in tests)."""
    from scipy.optimize import linear_sum_assignment
    B, N, K = W_pred.shape
    match = torch.zeros(B, K, dtype=torch.long)
    mask = torch.zeros(B, K, dtype=torch.bool)
    for b in range(B):
        n_gt = int(I_gt[b].max()) + 1
        onehot = torch.eye(n_gt + 1)[I_gt[b]]           # -1 -> last (background) row  :38
        inter = onehot.t() @ W_pred[b]
        union = onehot.sum(0).unsqueeze(1) + W_pred[b].sum(0).unsqueeze(0) - inter
        iou = (inter / union.clamp(min=1e-10))[:n_gt]
        _, col = linear_sum_assignment(-iou.detach().numpy())
        match[b, :n_gt] = torch.from_numpy(col)
        mask[b, :n_gt] = True
    return match, mask


def hard_W_encoding(W, to_null_mask=False, W_null_threshold=0.005):
    """losses.py:55-68."""
    N, K = W.shape[1], W.shape[2]
    hard = F.one_hot(W.argmax(2), K).float()
    if to_null_mask:
        null = (W.sum(1) < float(N) * W_null_threshold).float()
        hard = hard * (1.0 - null.unsqueeze(1))
    return hard


def get_mask_gt(I_gt, K):
    """losses.py:70-81."""
    return torch.arange(K).unsqueeze(0) < (I_gt.max(1)[0] + 1).unsqueeze(1)


def reduce_mean_masked_instance(loss, mask_gt):
    """losses.py:83-88."""
    s = torch.where(mask_gt, loss, torch.zeros_like(loss)).sum(1)
    n = mask_gt.float().sum(1)
    return torch.where(n > 0, s / n, torch.zeros_like(s))


def reorder(W, match):
    return torch.gather(W, 2, match.unsqueeze(1).expand(-1, W.shape[1], -1))


def compute_miou_loss(W, I_gt, match, div_eps=1e-10):
    """losses.py:90-103 -> (1-IoU) (B,K)."""
    K = match.shape[1]
    Wr = reorder(W, match)
    gt = torch.eye(K + 2)[I_gt][:, :, :K]               # -1 -> zero row
    dot = (gt * Wr).sum(1)
    den = gt.sum(1) + Wr.sum(1) - dot
    return 1.0 - dot / (den + div_eps)


def compute_segmentation_iou(W, I_gt, match, mask):
    """losses.py:106-109."""
    miou = 1 - compute_miou_loss(W, I_gt, match)
    return (mask * miou).sum(1) / mask.sum(1)


def acos_safe(x):
    return torch.acos(x.clamp(-1.0 + 1e-6, 1.0 - 1e-6))   # losses.py:123-124


def compute_normal_loss(n, n_gt, angle_diff=False, collapse=True):
    """losses.py:127-143."""
    a = (n * n_gt).sum(2).abs()
    v = acos_safe(a) if angle_diff else 1.0 - a
    return v.mean(1) if collapse else v


def compute_normal_difference(X, X_gt, in_radians=True, collapse=True):
    """losses.py:146-159."""
    v = acos_safe((X * X_gt).sum(2).abs())
    if not in_radians:
        v = v * 180.0 / TORCH_PI
    return v.mean(1) if collapse else v


def compute_all_losses(W, I_gt, X, X_gt, w_normal, w_miou):
    """losses.py:317-351 with collapse=True -> (total, normal, miou, match, mask)."""
    B, _, K = W.shape
    mask_gt = get_mask_gt(I_gt, K)
    normal = compute_normal_loss(X, X_gt) if w_normal > 0 else torch.zeros(B, K)
    if w_miou > 0:
        match, mask = hungarian_matching(W, I_gt)
        miou = reduce_mean_masked_instance(compute_miou_loss(W, I_gt, match), mask_gt)
    else:
        match, mask, miou = None, None, torch.zeros(B, K)
    t_miou, t_normal = miou.mean(), normal.mean()
    return w_miou * t_miou + w_normal * t_normal, t_normal, t_miou, match, mask


def bb_loss(W, W_raw, match, mask, bb_gt, K):
    """Base/barrel cross-entropy, inline in train_Point2Cyl_without_sketch.py:283-307."""
    B, N, _ = W.shape
    Wr = reorder(W, match)
    Wr = torch.where(mask.float().unsqueeze(1).expand(B, N, K) == 1, Wr, torch.zeros_like(Wr))
    Wr = torch.softmax(Wr, -1)
    Ws, label = torch.sort(Wr, -1)
    barrel = torch.gather(W_raw[:, :, 0::2], 2, label)
    base = torch.gather(W_raw[:, :, 1::2], 2, label)
    logits = torch.stack([barrel, base], -1).reshape(B * N * K, 2)
    tgt = bb_gt.unsqueeze(-1).expand(B, N, K).reshape(-1)
    ce = F.cross_entropy(logits, tgt, reduction="none").view(B, N, K)
    return (ce * Ws).sum(-1).mean(-1).mean()


# --------------------------------------------------------------------------- fitting
def _seg_norms(bb_gt, inst_gt, k):
    sel = (inst_gt == k).float()
    nb = (sel * (bb_gt == 0).float()).sum(-1)
    nc = (sel * (bb_gt == 1).float()).sum(-1)
    return nb.sqrt() + 1.0, nc.sqrt() + 1.0


def estimate_extrusion_axis(X, W_barrel, W_base, bb_gt=None, inst_gt=None, normalize=False, literal=False):
    """data_utils.py:99-177.  Eigenvector of the smallest eigenvalue of B^T B - C^T C where
    B = diag(w_barrel) X, C = diag(w_base) X, read from the upper triangle, ascending order.
    literal=True builds the N x N diag_embed like the reference (memory hungry); otherwise the
    same products are formed as (w*X)^T (w*X) (equal up to fp32 summation order)."""
    Bsz, N, K = W_barrel.shape
    out = torch.zeros(Bsz, K, 3, dtype=X.dtype)       # (a float64 run keeps its float64 eigenvectors)
    for k in range(K):
        wb, wc = W_barrel[:, :, k], W_base[:, :, k]
        if literal:
            Bm = torch.bmm(torch.diag_embed(wb), X)
            Cm = torch.bmm(torch.diag_embed(wc), X)
        else:
            Bm, Cm = wb.unsqueeze(-1) * X, wc.unsqueeze(-1) * X
        if normalize:
            sb, sc = _seg_norms(bb_gt, inst_gt, k)
            Bm, Cm = Bm / sb.view(-1, 1, 1), Cm / sc.view(-1, 1, 1)
        M = Bm.transpose(1, 2) @ Bm - Cm.transpose(1, 2) @ Cm
        _, v = torch.linalg.eigh(M, UPLO="U")
        out[:, k] = v[:, :, 0]
    return out


def estimate_extrusion_centers(W, pcs):
    """data_utils.py:253-266: (1/N) sum_n W[b,n,k] p[b,n,:]  (a mean over N, not a weighted mean)."""
    return (W.transpose(1, 2).unsqueeze(-1) * pcs.unsqueeze(1)).mean(-2)


def hard_centroids(EA_W, pcs):
    """eval.py:409-436: mean of the points with EA_W == 1; <= 1 point => not found (zeros)."""
    B, N, K = EA_W.shape
    cen, found = torch.zeros(B, K, 3), torch.zeros(B, K)
    for b in range(B):
        for k in range(K):
            sel = EA_W[b, :, k] == 1
            if int(sel.sum()) <= 1:
                continue
            cen[b, k] = pcs[b][sel].mean(0)
            found[b, k] = 1.0
    return cen, found


def get_extrusion_extents(P, seg, bb, axes, centers, rand_idx):
    """data_utils.py:1650-1730.  rand_idx[(k,b)] is the torch.randint draw of :1696 (indices into the
    ascending list of barrel points of segment k in cloud b) -> extents (K,B,2), found (B,K).
    Note :1671 skips segment k for the WHOLE batch when <= 1 barrel point exists across the batch."""
    B, K, _ = axes.shape
    found = torch.zeros(B, K)
    ext = torch.zeros(K, B, 2)
    barrel = F.one_hot(seg, K).bool() & (bb == 0).unsqueeze(-1)
    for k in range(K):
        if int(barrel[:, :, k].sum()) <= 1:
            continue
        S = None
        proj = {}
        for b in range(B):
            ids = barrel[b, :, k].nonzero().flatten()
            if ids.numel() <= 1:
                continue
            r = rand_idx[(k, b)]
            S = r.numel()
            proj[b] = P[b][ids[r]]
            found[b, k] = 1.0
        if S is None:
            S = 1
        for b in range(B):
            pts = proj.get(b, torch.zeros(S, 3))
            t = ((pts - centers[b, k]) * axes[b, k]).sum(-1)   # bmm (1x3)(3x1) :1712
            ext[k, b, 0], ext[k, b, 1] = t.min(), t.max()
    return ext, found


# --------------------------------------------------------------------------- sketch branch
def angle_axis_to_rotation_matrix(angle_axis, eps=1e-6):
    """torchgeometry==0.1.2 `angle_axis_to_rotation_matrix` (requirements.txt; call site data_utils.py:1101), which is NOT
    installed here and not vendored in the reference: PARITY UNPINNED at this boundary.  This restates its published
    algorithm: Rodrigues' formula with the axis normalised as aa / (theta + eps) where theta^2 = aa.aa > eps, and the
    first-order form I + [aa]x otherwise.  (N,3) -> (N,3,3) (the 3x3 block the reference keeps of the 4x4 result)."""
    aa = angle_axis
    theta2 = (aa * aa).sum(-1, keepdim=True)
    theta = torch.sqrt(theta2)
    w = aa / (theta + eps)
    wx, wy, wz = w[:, 0:1], w[:, 1:2], w[:, 2:3]
    c, s = torch.cos(theta), torch.sin(theta)
    one = 1.0
    R = torch.cat([c + wx * wx * (one - c), wx * wy * (one - c) - wz * s, wy * s + wx * wz * (one - c),
                   wz * s + wx * wy * (one - c), c + wy * wy * (one - c), -wx * s + wy * wz * (one - c),
                   -wy * s + wx * wz * (one - c), wx * s + wy * wz * (one - c), c + wz * wz * (one - c)], 1).view(-1, 3, 3)
    rx, ry, rz = aa[:, 0:1], aa[:, 1:2], aa[:, 2:3]
    k1 = torch.ones_like(rx)
    T = torch.cat([k1, -rz, ry, rz, k1, -rx, -ry, rx, k1], 1).view(-1, 3, 3)
    return torch.where((theta2 > eps).view(-1, 1, 1), R, T)


def axis_to_z_rotation(ax):
    """data_utils.py:1086-1103: the matrix the reference builds to turn axis `ax` (B,3) onto z.  The rotation vector is
    cross(ax, z) * acos(ax.z) with the cross product NOT normalised (so its length is angle*sin(angle), not angle) - kept."""
    B = ax.shape[0]
    z = torch.tensor([0.0, 0.0, 1.0]).expand(B, 3)
    ang = torch.acos((ax * z).sum(-1))
    R = torch.eye(3).repeat(B, 1, 1)
    for a in range(B):
        if ang[a] > G_ZERO_TOL:                                   # NaN (|ax.z| > 1) compares False: identity
            R[a] = angle_axis_to_rotation_matrix((torch.linalg.cross(ax[a], z[a]) * ang[a]).unsqueeze(0))[0]
    return R


def sketch_implicit_projection(P, X, seg, bb, axes, centers, rand_idx, S, all_points=False):
    """data_utils.py:1014-1146 (and :1149 = the same + the found mask, :1284 = all points of the cloud for every segment,
    no sampling).  rand_idx[(k,b)] is the torch.randint draw of :1064 (indices into the ascending list of barrel points of
    segment k in cloud b).  -> P_projected (K,B,S,2), X_projected (K,B,S,2), scales (K,B), found (B,K).
    Kept quirks: a segment with <= 1 barrel point in the WHOLE batch is skipped (outputs zero, scale 1) :1043; a cloud with
    <= 1 barrel point of the segment keeps zero samples, which still go through the projection and the centring (so its
    rows are -centroid_projected) and get scale 1 :1054, :1131, :1143."""
    B, K, _ = axes.shape
    Pp, Xp = torch.zeros(K, B, S, 2), torch.zeros(K, B, S, 2)
    found, scales = torch.zeros(B, K), torch.ones(K, B)
    member = torch.ones(B, P.shape[1], K, dtype=torch.bool) if all_points else (F.one_hot(seg, K).bool() & (bb == 0).unsqueeze(-1))
    for k in range(K):
        if int(member[:, :, k].sum()) <= 1:
            continue
        pts, nrm = torch.zeros(B, S, 3), torch.zeros(B, S, 3)
        for b in range(B):
            ids = member[b, :, k].nonzero().flatten()
            if ids.numel() <= 1:
                continue
            sel = ids if all_points else ids[rand_idx[(k, b)]]
            pts[b], nrm[b] = P[b][sel], X[b][sel]
            found[b, k] = 1.0
        R = axis_to_z_rotation(axes[:, k])
        q = torch.bmm(pts, R)[:, :, :2]                           # row vector times matrix :1110
        xq = torch.bmm(nrm, R)[:, :, :2]
        q = q - torch.bmm(centers[:, k].unsqueeze(1), R)[:, :, :2]
        scales[k] = (q.abs() ** 2).sum(-1).sqrt().max(-1)[0]
        Pp[k], Xp[k] = q, xq
    scales = torch.where(found.T == 1, scales, torch.ones(()))
    return Pp, Xp, scales, found


PN_ENCODER_LAYERS = (("mlp1.0", "mlp1.1"), ("mlp1.3", "mlp1.4"), ("mlp2.0", "mlp2.1"), ("mlp2.3", "mlp2.4"), ("mlp2.6", "mlp2.7"))


def pointnet_encoder_forward(sd, x, training=True, momentum=0.1):
    """IGR/network.py:132-174: x (B', S, C>=input_channels) -> unit-norm latent codes (B', E).  sd = its state_dict
    (mlp1.{0,3} / mlp2.{0,3,6} Conv1d, the BatchNorm1d after each, fc); running stats are updated in place when training."""
    cin = sd["mlp1.0.weight"].shape[1]
    h = x[:, :, :cin].transpose(2, 1)
    for c, b in PN_ENCODER_LAYERS:
        h = F.conv1d(h, sd[c + ".weight"], sd[c + ".bias"])
        h = F.batch_norm(h, sd[b + ".running_mean"], sd[b + ".running_var"], sd[b + ".weight"], sd[b + ".bias"], training, momentum, 1e-5)
        if training:
            sd[b + ".num_batches_tracked"] += 1
        h = F.relu(h)
    h = h.max(dim=2)[0]                                           # F.max_pool1d over all points :170
    h = F.linear(h, sd["fc.weight"], sd["fc.bias"])
    return F.normalize(h)


def implicit_net_forward(sd, inp, skip_in=(4,), beta=100):
    """IGR/network.py:67-92 with its state_dict (lin0..lin<L-1>): softplus(beta) between the layers, input re-injected at skip_in."""
    L = len([k for k in sd if k.endswith(".weight")])
    x = inp
    for layer in range(L):
        if layer in skip_in:
            x = torch.cat([x, inp], -1) / math.sqrt(2)
        x = F.linear(x, sd["lin%d.weight" % layer], sd["lin%d.bias" % layer])
        if layer < L - 1:
            x = F.softplus(x, beta=beta)
    return x


def implicit_losses(sd, sk_pnts, sk_normals, nonmnfld_pnts, latent, mask_gt, B, K, skip_in=(4,), beta=100):
    """train_Point2Cyl.py:611-648: manifold |f|, eikonal (|grad f| - 1)^2 on the off-surface samples, SALD normal term
    min(|grad f - n|, |grad f + n|) on the surface samples, masked means over the K segments -> (im_loss, mnfld, eikonal, normal)."""
    def add_latent(points, codes):
        b, n, d = points.shape
        return torch.cat([codes.unsqueeze(1).repeat(1, n, 1).reshape(b * n, -1), points.reshape(b * n, d)], 1)

    def grad(inputs, outputs):
        return torch.autograd.grad(outputs, inputs, torch.ones_like(outputs), create_graph=True, retain_graph=True)[0][:, -2:]
    a = add_latent(sk_pnts, latent).requires_grad_()
    n = add_latent(nonmnfld_pnts, latent).requires_grad_()
    fa, fn = implicit_net_forward(sd, a, skip_in, beta), implicit_net_forward(sd, n, skip_in, beta)
    ga, gn = grad(a, fa).reshape(B, K, -1, 2), grad(n, fn).reshape(B, K, -1, 2)
    mn = reduce_mean_masked_instance(fa.reshape(B, K, -1, 1).abs().mean(-1).mean(-1), mask_gt).mean()
    ek = reduce_mean_masked_instance(((gn.norm(2, dim=-1) - 1) ** 2).mean(-1), mask_gt).mean()
    nr = sk_normals.reshape(B, K, -1, 2)
    nl = torch.minimum((ga - nr).norm(2, dim=-1), (ga + nr).norm(2, dim=-1)).mean(-1)
    nl = reduce_mean_masked_instance(nl, mask_gt).mean()
    return mn + 0.1 * ek + 1.0 * nl, mn, ek, nl
