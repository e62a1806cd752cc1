"""Round-3 fixtures, made by RUNNING the upstream reference in this container (TEST INFRASTRUCTURE; only data is written).

    python -m oracle.make_golden_r3

G7b  g7b_axis64        data_utils.py:99-177 (`estimate_extrusion_axis`) run in FLOAT64 on G7's inputs (the reference function itself,
                       default dtype switched for the call so that its `torch.zeros` / `torch.tensor([1.0])` constants follow): axes, the
                       masked axis loss and the axis error in degrees for the four (hard | soft) x (norm_eig off | on) variants.
                       The yardstick that lets test_extrusion_axis_golden hold |sin(angle)| < 3e-7 instead of 1 - |dot| < 1e-6.
G9b  g9b_step_grads    the PRE-ADAM parameter gradients of G9's training step (train_Point2Cyl_without_sketch.py:244-368 at B=2, N=1024, the
                       inline base/barrel block :286-307 executed from the reference file) from the reference in fp32 AND in float64 (geometry
                       pinned to the fp32 indices, same FPS starts and dropout mask), every parameter.  Lets test_train_step_golden compare
                       gradients against the float64 yardstick instead of sign-matching Adam's first update.
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
from oracle.make_golden import DropoutOff, RandintTap, exec_reference_lines, save  # noqa: E402

K = 8
GOLDEN = os.path.join(ROOT, "tests", "golden")


def g7b(ref):
    du, ls = ref["data_utils"], ref["losses"]
    g = dict(np.load(os.path.join(GOLDEN, "g7_axis.npz")))
    t = torch.from_numpy
    arrs = {}
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        X, seg, bb, gt = t(g["X"]).double(), t(g["seg"]), t(g["bb"]), t(g["gt_axes"]).double()
        mg = ls.get_mask_gt(seg, K)
        for wtag in ("hard", "soft"):
            wb, wc = t(g["Wb_" + wtag]).double(), t(g["Wc_" + wtag]).double()
            for norm in (False, True):
                xx = X.clone().requires_grad_(True)
                wbb, wcc = wb.clone().requires_grad_(True), wc.clone().requires_grad_(True)
                E = du.estimate_extrusion_axis(xx, wbb, wcc, bb, seg, normalize=norm)
                assert E.dtype == torch.float64
                lo = ls.reduce_mean_masked_instance(ls.compute_normal_loss(E, gt, angle_diff=False, collapse=False), mg).mean()
                lo.backward()
                tag = "%s_%d" % (wtag, int(norm))
                arrs.update({"E64_" + tag: E, "loss64_" + tag: lo, "gX64_" + tag: xx.grad, "gWb64_" + tag: wbb.grad, "gWc64_" + tag: wcc.grad,
                             "deg64_" + tag: ls.compute_normal_difference(E.detach(), gt, in_radians=False, collapse=False)})
    finally:
        torch.set_default_dtype(old)
    save("g7b_axis64", **arrs)


def g9b(ref):
    pu, pe, ls = ref["pointnet_util"], ref["pointnet_extrusion"], ref["losses"]
    g = dict(np.load(os.path.join(GOLDEN, "g9_train_step.npz")))
    t = torch.from_numpy
    pcs, normals, seg, bb = t(g["pcs"]), t(g["normals"]), t(g["seg"]), t(g["bb"])
    dmask = t(np.unpackbits(g["dropout_mask_bcn"])[: 2 * 128 * 1024].reshape(2, 128, 1024).astype(np.float32))
    arrs = {}

    def run(dtype, forced=None):
        old = torch.get_default_dtype()
        torch.set_default_dtype(dtype)
        o_fps, o_ball, o_sq = pu.farthest_point_sample, pu.query_ball_point, pu.square_distance
        def in_f32(fn):            # the geometry stays the fp32 one (indices / distances pinned), whatever the default dtype is
            def w(*a):
                torch.set_default_dtype(torch.float32)
                try:
                    return fn(*a)
                finally:
                    torch.set_default_dtype(dtype)
            return w

        if dtype == torch.float64:                     # (as make_golden_r2.g12's float64 twin)
            pu.farthest_point_sample = in_f32(lambda xyz, n: o_fps(xyz.float(), n))
            pu.query_ball_point = in_f32(lambda r, ns_, xyz, nx: o_ball(r, ns_, xyz.float(), nx.float()))
            pu.square_distance = in_f32(lambda a, b: o_sq(a.float(), b.float()).to(a.dtype))
        o_hm = ls.hungarian_matching
        if forced is not None:     # at random initialisation the soft IoU costs are within 1e-7 of ties: the assignment is pinned to the fp32
            # run's, like the geometry - the float64 twin is the same PROBLEM in exact arithmetic, not a different matching
            ls.hungarian_matching = lambda W, I, with_mask=False: (forced if with_mask else forced[0])
        try:
            torch.manual_seed(int(g["seed"]))
            torch.set_default_dtype(torch.float32)     # the SAME initial parameters: drawn in fp32, then widened
            model = pe.backbone(output_sizes=[3, 2 * K])
            torch.set_default_dtype(dtype)
            model = model.to(dtype).train()
            for name, m in model.named_modules():
                if "bn" in name:
                    m.momentum = 0.5
            torch.manual_seed(100)                     # the FPS start draws of G9
            with RandintTap() as tap, DropoutOff(dmask.to(dtype)):
                Xo, W_raw = model(pcs.to(dtype))
            assert torch.equal(tap.draws[0], t(g["start1"])) and torch.equal(tap.draws[1], t(g["start2"]))
            Xo = F.normalize(Xo, p=2, dim=2, eps=1e-12)
            W_2K = torch.softmax(W_raw, dim=2)
            W = W_2K[:, :, ::2] + W_2K[:, :, 1::2]
            total, nl, ml, match, mask = ls.compute_all_losses(pcs.to(dtype), W, seg, Xo, normals.to(dtype), 1.0, 1.0, return_match_indices=True)
            ns = dict(torch=torch, F=F, W=W, matching_indices=match, mask=mask, sampled_pcs=pcs, NUM_POINT=1024, K=K,
                      W_barrel_bb=W_raw[:, :, ::2], W_base_bb=W_raw[:, :, 1::2], gt_bb_labels=bb, batch_size=2)
            exec_reference_lines(os.path.join(_refload.REF_ROOT, "train_Point2Cyl_without_sketch.py"), 286, 307, ns)
            total = total + 1.0 * ns["total_bb_loss"]
            total.backward()
            return total.item(), (match, mask), {n: p.grad.detach().clone() for n, p in model.named_parameters()}
        finally:
            pu.farthest_point_sample, pu.query_ball_point, pu.square_distance = o_fps, o_ball, o_sq
            ls.hungarian_matching = o_hm
            torch.set_default_dtype(old)

    t32, m32, g32 = run(torch.float32)
    assert abs(t32 - float(g["total"])) <= 1e-6 * abs(t32), (t32, float(g["total"]))     # the same step as G9
    assert np.array_equal(m32[0].numpy(), g["match"])
    t64, m64, g64 = run(torch.float64, forced=m32)
    print('match fp32', m32[0].tolist(), 'fp64', m64[0].tolist(), 'mask', m32[1].tolist())
    assert torch.equal(m32[1], m64[1]) and torch.equal(m32[0][m32[1].bool()], m64[0][m64[1].bool()]), 'the matched (unmasked) pairs must agree'
    arrs.update(total=t32, total64=t64, names=np.array(list(g32.keys())))
    kept = []
    for n in g32:                                      # every 1-D parameter and the weight matrices up to 20 k elements (fixture size)
        if g32[n].dim() > 1 and g32[n].numel() > 20000:
            continue
        kept.append(n)
        arrs["g32:" + n] = g32[n].reshape(g32[n].shape[0], -1) if g32[n].dim() > 1 else g32[n]
        arrs["g64:" + n] = g64[n].reshape(g64[n].shape[0], -1) if g64[n].dim() > 1 else g64[n]
    arrs["kept"] = np.array(kept)
    # the large matrices: their norms and the fp32 run's distance from float64, so that a test can still bound them in norm
    arrs["big_names"] = np.array([n for n in g32 if n not in kept])
    arrs["big_norm64"] = np.array([g64[n].norm().item() for n in g32 if n not in kept])
    arrs["big_relerr32"] = np.array([((g32[n].double() - g64[n]).norm() / g64[n].norm()).item() for n in g32 if n not in kept])
    save("g9b_step_grads", **arrs)


def g15(ref):
    """G15 g15_dataloader: items of the reference's AutodeskDataset_h5 / AutodeskDataset_h5_sketches (dataloader.py:15-296) on a small
    in-memory file of the schema utils.py:1174-1188 / :1251-1268 writes (its load_h5 / load_h5_sk are pointed at the arrays: h5py is not
    installed here), for the flag combinations the scripts use and the full ones, under fixed torch seeds."""
    import importlib
    saved = list(sys.path)
    sys.path[:0] = [_refload.REF_ROOT, os.path.join(_refload.REF_ROOT, "models")]
    try:
        dl = importlib.import_module("dataloader")
    finally:
        sys.path[:] = saved
        for name in ("dataloader", "utils", "data_utils", "global_variables", "losses", "pointnet_util", "models", "models.pointnet_util"):
            sys.modules.pop(name, None)
    rng = np.random.default_rng(15)
    n, P, K, S = 3, 96, 8, 40
    f = dict(point_cloud=rng.normal(size=(n, P, 3)).astype(np.float32), normals=rng.normal(size=(n, P, 3)).astype(np.float32),
             extrusion_labels=rng.integers(0, 5, (n, P)).astype(np.int64), base_barrel_labels=rng.integers(0, 2, (n, P)).astype(np.int64),
             n_instances=np.array([5, 8, 3], dtype=np.int64), extrusion_axes=rng.normal(size=(n, K, 3)).astype(np.float32),
             extrusion_distances=rng.random((n, K)).astype(np.float32), extrusion_operation=rng.integers(0, 3, (n, P)).astype(np.int64),
             extrusion_centers=rng.normal(size=(n, K, 3)).astype(np.float32), extrusion_extents=rng.random((n, K, 2)).astype(np.float32),
             sketches=rng.normal(size=(n, K, S, 4)).astype(np.float32), sketches_norms=rng.random((n, K)).astype(np.float32))

    def load_h5(fn, op=False, center=False, extent=False):
        out = [f[k] for k in ("point_cloud", "normals", "extrusion_labels", "base_barrel_labels", "n_instances", "extrusion_axes", "extrusion_distances")]
        if op:
            out.append(f["extrusion_operation"])
        if center:
            out.append(f["extrusion_centers"])
        if extent:
            out.append(f["extrusion_extents"])
        return tuple(out)

    def load_h5_sk(fn, op=False, center=False, extent=False):
        out = [f[k] for k in ("point_cloud", "normals", "extrusion_labels", "base_barrel_labels", "n_instances", "extrusion_axes", "extrusion_distances")]
        if op:
            out.append(f["extrusion_operation"])
        if center:
            out.append(f["extrusion_centers"])
        out += [f["sketches"], f["sketches_norms"]]
        if extent:
            out.append(f["extrusion_extents"])
        return tuple(out)

    dl.load_h5, dl.load_h5_sk = load_h5, load_h5_sk
    arrs = {"file:" + k: v for k, v in f.items()}
    cases = []
    for ci, (op, center, extent) in enumerate([(False, True, False), (False, False, False), (True, True, True), (True, False, False)]):
        ds = dl.AutodeskDataset_h5("mem", 32, K, op=op, center=center, extent=extent)
        torch.manual_seed(150 + ci)
        item = ds[ci % n]
        cases.append(("h5", op, center, False, extent, 150 + ci, ci % n, len(item)))
        for j, v in enumerate(item):
            arrs["h5_%d:%d" % (ci, j)] = np.asarray(v)
    for ci, (op, center, scale, extent) in enumerate([(False, True, False, False), (False, False, False, False), (True, True, True, True),
                                                      (False, True, True, False), (True, False, False, True)]):
        ds = dl.AutodeskDataset_h5_sketches("mem", 32, 16, K, op=op, center=center, with_scale=scale, extent=extent)
        torch.manual_seed(250 + ci)
        item = ds[(ci + 1) % n]
        cases.append(("sk", op, center, scale, extent, 250 + ci, (ci + 1) % n, len(item)))
        for j, v in enumerate(item):
            arrs["sk_%d:%d" % (ci, j)] = np.asarray(v)
    arrs["cases"] = np.array([[c[0]] + [str(int(x)) for x in c[1:]] for c in cases])
    save("g15_dataloader", **arrs)


def main():
    ref = _refload.load()
    torch.set_num_threads(8)
    if len(sys.argv) > 1:
        for name in sys.argv[1:]:
            globals()[name](ref)
        return
    g7b(ref)
    g9b(ref)
    g15(ref)


if __name__ == "__main__":
    main()
