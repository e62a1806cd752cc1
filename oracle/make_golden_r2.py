"""Round-2 fixtures, made by RUNNING the upstream reference in this container (TEST INFRASTRUCTURE; data only is written).

    python -m oracle.make_golden_r2

G12  g12_train_5steps  five consecutive Adam steps of the reference trainer's step (train_Point2Cyl_without_sketch.py:244-369;
                       the inline base/barrel block :286-307 executed from the reference file) at B=8, N=1024: per-step FPS
                       starts, dropout masks (seed + checksums), the four loss scalars, and parameter checksums after the fifth step.  Pins a
                       WELL-CONDITIONED end-to-end quantity (the loss trajectory) next to G9's single step.
G13  g13_eval_flow     eval.py:270-457 executed from the reference file on seeded "predictions" (B=3, N=1024), for the four
                       --use_gt_* operand choices, --use_gt_normals and --norm_eig: every per-cloud metric, the matching, labels,
                       axes, centroids, extents (with the recorded randint draws) -- and the SAME lines run in float64, which is
                       the yardstick for the axis-angle metric (an acos next to its clamp: two correct fp32 paths differ by
                       more than 1e-4 from each other).
G14  g14_add_noise     data_utils.py:84-96 under a fixed NumPy seed.
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import _refload  # noqa: E402
from oracle.make_golden import DropoutOff, RandintTap, cksum, exec_reference_lines, save  # noqa: E402
from point2cyl_amd import synth  # noqa: E402

K = 8


def g12(ref):
    pe, ls = ref["pointnet_extrusion"], ref["losses"]
    B, N, STEPS = 8, 1024, 5          # B=2 would put 2-row BatchNorms (SA3 / FP3 see one row per cloud) into the chain
    pcs, normals, seg, bb, _, _, axes, _, centers = synth.make_batch(B, N, K, seed=77)
    arrs = dict(pcs=pcs, normals=normals, seg=seg, bb=bb, seed=2468, steps=STEPS, mask_seed=12)
    # Adam's first updates are ~lr*sign(g) and the deep gradients of this network are ill-conditioned in fp32: ANY
    # two fp32 implementations - the reference and oracle/ref_torch.py's restatement of it included, bit-equal at step 0 - drift apart
    # by ~1e-5 at the third step, ~1e-3 at the fourth and ~1e-2 at the fifth (plain SGD drifts the same way).  The tests bound each
    # step accordingly; the early steps are the pin, the late ones a gross-error check.
    for tag, mk in (("adam", lambda ps: torch.optim.Adam(ps, lr=1e-3)),):
        torch.manual_seed(2468)
        model = pe.backbone(output_sizes=[3, 2 * K]).train()
        for name, m in model.named_modules():
            if "bn" in name:
                m.momentum = 0.5
        opt = mk(model.parameters())
        gm = torch.Generator().manual_seed(12)
        torch.manual_seed(555)
        scal, starts1, starts2, masks = [], [], [], []
        for s in range(STEPS):
            dmask = (torch.rand(B, 128, N, generator=gm) < 0.5).float()          # (B,C,N) like the reference tensor
            with RandintTap() as tap, DropoutOff(dmask):
                Xo, W_raw = model(pcs)
            Xo = F.normalize(Xo, p=2, dim=2, eps=1e-12)
            W_2K = torch.softmax(W_raw, dim=2)
            W = W_2K[:, :, ::2] + W_2K[:, :, 1::2]
            total, nl, ml, match, mask = ls.compute_all_losses(pcs, W, seg, Xo, normals, 1.0, 1.0, return_match_indices=True)
            ns = dict(torch=torch, F=F, W=W, matching_indices=match, mask=mask, sampled_pcs=pcs, NUM_POINT=N, K=K,
                      W_barrel_bb=W_raw[:, :, ::2], W_base_bb=W_raw[:, :, 1::2], gt_bb_labels=bb, batch_size=B)
            exec_reference_lines(os.path.join(_refload.REF_ROOT, "train_Point2Cyl_without_sketch.py"), 286, 307, ns)
            bbl = ns["total_bb_loss"]
            total = total + 1.0 * bbl
            opt.zero_grad()
            total.backward()
            opt.step()
            scal.append([total.item(), nl.item(), ml.item(), bbl.item()])
            starts1.append(tap.draws[0])
            starts2.append(tap.draws[1])
            masks.append(cksum(dmask))          # the masks are re-drawn from `mask_seed` by the tests; their checksums guard the draw
            arrs["%s_match_%d" % (tag, s)] = match
        sd = model.state_dict()
        arrs.update({tag + "_losses": np.array(scal), tag + "_start1": torch.stack(starts1), tag + "_start2": torch.stack(starts2),
                     "dropout_mask_ck": np.stack(masks), tag + "_param_ck": np.stack([cksum(p) for _, p in model.named_parameters()])})
        arrs.update({"%s_after:%s" % (tag, n): sd[n] for n in ("sa1.mlp_bns.0.running_mean", "bn1.running_var", "fp1.mlp_bns.2.running_mean")})
    arrs["param_names"] = np.array([n for n, _ in model.named_parameters()])
    # the same five steps with the reference module in float64 (geometry pinned to the fp32 indices, same starts / masks): the
    # yardstick that tells how far the reference's own fp32 trajectory is from the exact one at every step
    pu = ref["pointnet_util"]
    torch.manual_seed(2468)
    m64 = pe.backbone(output_sizes=[3, 2 * K]).double().train()
    for name, m in m64.named_modules():
        if "bn" in name:
            m.momentum = 0.5
    opt = torch.optim.Adam(m64.parameters(), lr=1e-3)
    gm = torch.Generator().manual_seed(12)
    o_fps, o_ball, o_sq, o_randint = pu.farthest_point_sample, pu.query_ball_point, pu.square_distance, torch.randint
    def in_f32(fn):                 # the geometry stays the fp32 one (indices / distances pinned), whatever the default dtype is
        def w(*a):
            torch.set_default_dtype(torch.float32)
            try:
                return fn(*a)
            finally:
                torch.set_default_dtype(torch.float64)
        return w

    pu.farthest_point_sample = in_f32(lambda xyz, n: o_fps(xyz.float(), n))
    pu.query_ball_point = in_f32(lambda r, ns_, xyz, nx: o_ball(r, ns_, xyz.float(), nx.float()))
    pu.square_distance = in_f32(lambda a, b: o_sq(a.float(), b.float()).to(a.dtype))
    scal64 = []
    torch.set_default_dtype(torch.float64)          # the reference's torch.eye / torch.zeros constants follow the run's dtype
    try:
        for s in range(STEPS):
            dmask = (torch.rand(B, 128, N, generator=gm) < 0.5).double()
            forced = iter([arrs["adam_start1"][s], arrs["adam_start2"][s]])
            torch.randint = lambda *a, **k: next(forced).clone()
            with DropoutOff(dmask):
                Xo, W_raw = m64(pcs.double())
            torch.randint = o_randint
            Xo = F.normalize(Xo, p=2, dim=2, eps=1e-12)
            W_2K = torch.softmax(W_raw, dim=2)
            W = W_2K[:, :, ::2] + W_2K[:, :, 1::2]
            total, nl, ml, match, mask = ls.compute_all_losses(pcs.double(), W, seg, Xo, normals.double(), 1.0, 1.0, return_match_indices=True)
            ns = dict(torch=torch, F=F, W=W, matching_indices=match, mask=mask, sampled_pcs=pcs, NUM_POINT=N, K=K,
                      W_barrel_bb=W_raw[:, :, ::2], W_base_bb=W_raw[:, :, 1::2], gt_bb_labels=bb, batch_size=B)
            exec_reference_lines(os.path.join(_refload.REF_ROOT, "train_Point2Cyl_without_sketch.py"), 286, 307, ns)
            bbl = ns["total_bb_loss"]
            total = total + 1.0 * bbl
            opt.zero_grad()
            total.backward()
            opt.step()
            scal64.append([total.item(), nl.item(), ml.item(), bbl.item()])
    finally:
        pu.farthest_point_sample, pu.query_ball_point, pu.square_distance, torch.randint = o_fps, o_ball, o_sq, o_randint
        torch.set_default_dtype(torch.float32)
    arrs["adam_losses64"] = np.array(scal64)
    print("fp32 reference vs its float64 run, max relative loss difference per step:",
          np.abs(arrs["adam_losses"] / arrs["adam_losses64"] - 1).max(1))
    save("g12_train_5steps", **arrs)


def g13(ref):
    ls, du = ref["losses"], ref["data_utils"]
    B, N, S = 3, 1024, 256
    pcs, normals, seg, bb, _, _, axes, _, centers = synth.make_batch(B, N, K, seed=1313)
    g = torch.Generator().manual_seed(13)
    # "predictions": logits biased towards a PERMUTED labelling (so the matching is a real permutation), 12 % of the points flipped to
    # a random class; normals = gt + noise (degrees-level axis errors: the metric is then well-conditioned)
    perm = torch.stack([torch.randperm(K, generator=g) for _ in range(B)])
    plab = torch.gather(perm, 1, seg)
    cls = plab * 2 + bb
    flip = torch.rand(B, N, generator=g) < 0.12
    cls = torch.where(flip, torch.randint(0, 2 * K, (B, N), generator=g), cls)
    W_raw = torch.randn(B, N, 2 * K, generator=g) + 5.0 * F.one_hot(cls, 2 * K)
    X_head = (normals + 0.15 * torch.randn(B, N, 3, generator=g)) * (0.5 + torch.rand(B, N, 1, generator=g))
    arrs = dict(pcs=pcs, normals=normals, seg=seg, bb=bb, axes=axes, centers=centers, W_raw=W_raw, X_head=X_head, S=S)
    combos = (("pred", {}), ("gtn", dict(USE_GT_NORMALS=True)), ("gtseg_gtbb", dict(USE_GT_SEGMENTATION=True, USE_GT_BB=True)),
              ("gtseg", dict(USE_GT_SEGMENTATION=True)), ("gtbb", dict(USE_GT_BB=True)), ("pred_normeig", dict(NORM_EIG=True)))
    path = os.path.join(_refload.REF_ROOT, "eval.py")
    for tag, over in combos:
        for dt, suffix in ((torch.float32, ""), (torch.float64, "64")):
            ns = dict(torch=torch, F=F, K=K, NUM_POINT=N, NUM_SK_POINT=S, PRED_NORMAL=True, PRED_SEG=True, PRED_BB=True, PRED_EXT=True,
                      USE_GT_NORMALS=False, USE_GT_SEGMENTATION=False, USE_GT_BB=False, NORM_EIG=False, device=torch.device("cpu"),
                      cur_batch_size=B, batch_size=B, pcs=pcs.to(dt), gt_normals=normals.to(dt), gt_extrusion_instances=seg,
                      gt_bb_labels=bb.to(dt), gt_extrusion_axes=axes.to(dt), gt_extrusion_centers=centers.to(dt),
                      X=X_head.to(dt), W_raw=W_raw.to(dt))
            for mod in (ls, du):
                ns.update({k: v for k, v in vars(mod).items() if not k.startswith("__")})
            ns.update(over)
            torch.manual_seed(130)
            orig_float = torch.Tensor.float
            if suffix:      # float64 twin: the reference's explicit .float() casts and float32 constants / buffers follow the run's dtype
                torch.set_default_dtype(torch.float64)
                torch.Tensor.float = lambda self, *a, **k: self.double()
            try:
                with RandintTap() as tap:
                    exec_reference_lines(path, 270, 457, ns)
            finally:
                torch.Tensor.float = orig_float
                torch.set_default_dtype(torch.float32)
            out = {k: ns[k] for k in ("mIoU", "normal_difference", "pred_bb_acc", "extrusion_difference", "centroid_difference",
                                      "extrusion_difference_uncollapsed", "centroid_difference_uncollapsed", "E_AX", "predicted_centroids",
                                      "found_centers_mask", "extents", "label", "pred_bb_label", "matching_indices", "mask")}
            if suffix:          # float64 twin: only the real-valued metrics are kept
                for k in ("normal_difference", "extrusion_difference", "centroid_difference", "extrusion_difference_uncollapsed", "E_AX"):
                    arrs["%s:%s64" % (tag, k)] = out[k]
                continue
            for k, v in out.items():
                arrs["%s:%s" % (tag, k)] = v
            if tag == "pred":
                # extents' sampling draws in the reference's order (k outer, b inner, only where > 1 barrel point): data_utils.py:1696
                barrel = F.one_hot(seg, K).bool() & (bb == 0).unsqueeze(-1)
                keys, di = [], 0
                for k in range(K):
                    if int(barrel[:, :, k].sum()) <= 1:
                        continue
                    for b in range(B):
                        if int(barrel[b, :, k].sum()) <= 1:
                            continue
                        keys.append((k, b))
                        di += 1
                assert di == len(tap.draws), (di, len(tap.draws))
                arrs["rand_keys"] = np.array(keys)
                arrs["rand_idx"] = torch.stack(tap.draws)
    save("g13_eval_flow", **arrs)


def g14(ref):
    """add_noise (data_utils.py:84-96): NumPy's global generator, float64 result."""
    du = ref["data_utils"]
    pcs, normals = synth.make_batch(2, 64, K, seed=14)[:2]
    np.random.seed(1414)
    out = du.add_noise(pcs, normals, sigma=0.02)
    save("g14_add_noise", pcs=pcs, normals=normals, np_seed=1414, sigma=0.02, out=out, out_dtype=str(out.dtype))


def main():
    ref = _refload.load()
    torch.set_num_threads(8)
    g12(ref)
    g13(ref)
    g14(ref)


if __name__ == "__main__":
    main()
