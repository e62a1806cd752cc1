"""Generate tests/golden/g10_sketch.npz by RUNNING the upstream reference's sketch branch (this container only):

    python -m oracle.make_golden_sketch

sketch_implicit_projection / sketch_implicit_projection3 (data_utils.py:1014, :1284) and PointNetEncoder
(IGR/network.py:132-174).  Only DATA is written.  torchgeometry is not installed, so the one function of it that the
projection calls (angle_axis_to_rotation_matrix, data_utils.py:1101) is supplied from oracle/ref_torch.py's restatement of
its published algorithm: the fixture pins everything the reference itself does around that call, not that call
(PARITY UNPINNED at the torchgeometry boundary)."""
import importlib
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import _refload, ref_torch as R  # noqa: E402
from oracle.make_golden import RandintTap, save  # noqa: E402
from point2cyl_amd import synth  # noqa: E402


def main():
    ref = _refload.load()
    du = ref["data_utils"]
    tgm = sys.modules["torchgeometry"]

    def aa2rot(aa):                       # 4x4 like the real function; the reference takes [0, :3, :3]
        out = torch.eye(4).repeat(aa.shape[0], 1, 1)
        out[:, :3, :3] = R.angle_axis_to_rotation_matrix(aa)
        return out
    tgm.angle_axis_to_rotation_matrix = aa2rot
    du.tgm = tgm

    B, N, K, S = 3, 512, 8, 128
    pcs, nrm, seg, bb, _, _, axes, _, cen = synth.make_batch(B, N, K, seed=311)
    pcs, nrm, axes, cen = pcs.float(), nrm.float(), axes.float(), cen.float()
    seg = seg.clone(); bb = bb.clone()
    # exercise the quirks: segment 1 of cloud 0 keeps ONE barrel point (not found in that cloud), an axis equal to +z
    # (identity rotation) and one nearly +z (first-order branch of the rotation)
    ids = ((seg[0] == 1) & (bb[0] == 0)).nonzero().flatten()
    if ids.numel() > 1:
        bb[0, ids[1:]] = 1
    axes[1, 0] = torch.tensor([0.0, 0.0, 1.0])
    axes[2, 0] = F.normalize(torch.tensor([3e-4, -2e-4, 1.0]), dim=0)
    torch.manual_seed(10)
    with RandintTap() as tap:
        Pp, Xp, sc = du.sketch_implicit_projection(pcs, nrm, seg, bb, axes, cen, num_points_to_sample=S)
    _, _, _, found = du.sketch_implicit_projection2(pcs, nrm, seg, bb, axes, cen, num_points_to_sample=S)
    keys, draws, di = [], [], 0
    barrel = F.one_hot(seg, K).bool() & (bb == 0).unsqueeze(-1)
    for k in range(K):
        if int(barrel[:, :, k].sum()) <= 1:
            continue
        for b in range(B):
            if int(barrel[b, :, k].sum()) <= 1:
                continue
            keys.append((k, b)); draws.append(tap.draws[di]); di += 1
    assert di == len(tap.draws), (di, len(tap.draws))
    P3, X3, sc3, found3 = du.sketch_implicit_projection3(pcs, nrm, seg, bb, axes, cen, num_points_to_sample=N)

    # PointNetEncoder forward + backward, train-mode BatchNorm
    saved = list(sys.path)
    sys.path[:0] = [os.path.join(_refload.REF_ROOT, "IGR")]
    try:
        for n in ("general", "network"):
            sys.modules.pop(n, None)
        net = importlib.import_module("network")
    finally:
        sys.path[:] = saved
        for n in ("general", "network"):
            sys.modules.pop(n, None)
    torch.manual_seed(77)
    enc = net.PointNetEncoder(32, 2, with_normals=True).train()
    sd0 = {k: v.detach().clone() for k, v in enc.state_dict().items()}
    g = torch.Generator().manual_seed(5)
    x = torch.randn(6, 96, 4, generator=g).requires_grad_(True)
    z = enc(x)
    tgt = F.normalize(torch.randn(6, 32, generator=g))
    loss = ((z - tgt) ** 2).sum()
    loss.backward()
    sd1 = enc.state_dict()
    arrs = dict(pcs=pcs, normals=nrm, seg=seg, bb=bb, axes=axes, centers=cen, S=S, P_proj=Pp, X_proj=Xp, scales=sc, found=found,
                rand_keys=np.array(keys), rand_idx=torch.stack(draws), P_proj3=P3, X_proj3=X3, scales3=sc3, found3=found3,
                enc_x=x.detach(), enc_tgt=tgt, enc_z=z, enc_loss=loss, enc_gx=x.grad, enc_names=np.array(list(sd0.keys())))
    for k, v in sd0.items():
        arrs["enc_sd:" + k] = v
    for k in ("mlp1.1.running_mean", "mlp2.7.running_var", "mlp2.7.num_batches_tracked"):
        arrs["enc_after:" + k] = sd1[k]
    for n, p in enc.named_parameters():
        arrs["enc_grad:" + n] = p.grad
    save("g10_sketch", **arrs)


if __name__ == "__main__":
    main()
