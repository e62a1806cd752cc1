"""GPU (MI355X): the script-level flows of the reference on the HIP path, against reference-generated fixtures.

  * eval.py:270-457 (G13, oracle/make_golden_r2.py executes those lines from the reference file): every metric of the
    fitting-accuracy report, the four --use_gt_* operand choices, --use_gt_normals, --norm_eig; the axis-angle metric against
    the float64 run of the same reference lines (1e-4 relative on the batch mean);
  * five consecutive Adam steps of train_Point2Cyl_without_sketch.py:244-369 (G12): the loss trajectory at rtol 1e-3;
  * the trainer CLI at BASELINE configs[0] (4 synthetic shapes, B=2, N=1024, 1 epoch) - through the HIP-graph path bench.py times;
  * the eval CLI on a trainer checkpoint; backbone(normal_channel=True).
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ref_torch as R
from tests.conftest import load_golden

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from point2cyl_amd import ops, step, synth
    from point2cyl_amd import eval as p2c_eval
    from point2cyl_amd.backbone import backbone

DEV = "cuda"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
t = torch.from_numpy


def cu(a):
    return (t(a) if isinstance(a, np.ndarray) else a).to(DEV)


COMBOS = {"pred": {}, "gtn": dict(use_gt_normals=True), "gtseg_gtbb": dict(use_gt_segmentation=True, use_gt_bb=True),
          "gtseg": dict(use_gt_segmentation=True), "gtbb": dict(use_gt_bb=True), "pred_normeig": dict(norm_eig=True)}


@pytest.mark.parametrize("tag", list(COMBOS))
def test_eval_flow_golden(tag):
    g = load_golden("g13_eval_flow")
    B, N, K, S = 3, 1024, 8, int(g["S"])
    fl = p2c_eval.EvalFlags(K=K, num_sk_point=S, **COMBOS[tag])
    ridx = torch.zeros(B, K, S, dtype=torch.int64)
    for (k, b), draw in zip(g["rand_keys"], g["rand_idx"]):
        ridx[b, k] = t(draw)
    m = p2c_eval.eval_metrics(cu(g["X_head"]), cu(g["W_raw"]), cu(g["pcs"]), cu(g["normals"]), cu(g["seg"]), cu(g["bb"]).float(),
                              cu(g["axes"]), cu(g["centers"]), fl, extent_rand_idx=ridx)
    r = lambda k: g["%s:%s" % (tag, k)]
    c = lambda k: m[k].detach().cpu().numpy()
    # integer structure: bit-exact
    assert np.array_equal(c("matching_indices"), r("matching_indices")) and np.array_equal(c("mask") > 0, r("mask") > 0)
    assert np.array_equal(c("label"), r("label")) and np.array_equal(c("pred_bb_label"), r("pred_bb_label"))
    assert np.array_equal(c("found_centers_mask") > 0, r("found_centers_mask") > 0)
    # fp32 metrics: 1e-4
    np.testing.assert_allclose(c("mIoU"), r("mIoU"), rtol=1e-4)
    np.testing.assert_allclose(c("pred_bb_acc"), r("pred_bb_acc"), rtol=1e-6)
    np.testing.assert_allclose(c("normal_difference"), r("normal_difference"), rtol=1e-4)
    np.testing.assert_allclose(c("predicted_centroids"), r("predicted_centroids"), rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(c("centroid_difference"), r("centroid_difference"), rtol=1e-4, atol=1e-8)
    np.testing.assert_allclose(c("centroid_difference_uncollapsed"), r("centroid_difference_uncollapsed"), rtol=1e-4, atol=1e-8)
    np.testing.assert_allclose(c("extents"), r("extents"), rtol=1e-4, atol=2e-6)
    # fitted axes: up to the sign every consumer ignores, against the float64 run of the reference lines
    mg = c("mask_gt")
    E64 = g["%s:E_AX64" % tag]
    sin = np.linalg.norm(np.cross(c("E_AX").astype(np.float64), E64), axis=-1)      # |sin(angle)|: sign-free, exact for small angles
    assert (sin[mg] < 3e-7).all(), sin[mg].max()                                      # fp32 storage of a unit vector: ~1e-7
    # the headline eval metric (eval.py:398-405) against the float64 run: 1e-4 relative on the batch mean and per cloud
    ref64 = g["%s:extrusion_difference64" % tag]
    got = c("extrusion_difference").astype(np.float64)
    assert abs(got.mean() - ref64.mean()) <= 1e-4 * ref64.mean(), (got.mean(), ref64.mean())
    np.testing.assert_allclose(got, ref64, rtol=1e-4)
    u64 = g["%s:extrusion_difference_uncollapsed64" % tag]
    np.testing.assert_allclose(c("extrusion_difference_uncollapsed")[mg], u64[mg], rtol=2e-4, atol=1e-6)
    # and the reference's own fp32 value agrees with us no worse than with its float64 self (x2)
    r32 = r("extrusion_difference").astype(np.float64)
    assert np.abs(got - r32).max() <= 2 * np.abs(r32 - ref64).max() + 1e-5


@pytest.mark.parametrize("tag", ["pred", "pred_normeig"])
def test_eval_flow_golden_fused_metrics(tag):
    """G13 through csrc/metrics.hip (ops.eval_metrics_fused): the reference's own eval.py:270-446 outputs for its default operand choice -
    matching, mask, found-centroid mask bit-exact; the five report metrics within 1e-4 (the axis angle against the reference's float64 run,
    as test_eval_flow_golden holds the torch-op mirror to)."""
    g = load_golden("g13_eval_flow")
    B, N, K = 3, 1024, 8
    heads = torch.cat([t(g["X_head"]).float().reshape(B * N, 3), t(g["W_raw"]).float().reshape(B * N, 2 * K), torch.zeros(B * N, 1)], 1).contiguous().to(DEV)
    out, det = ops.eval_metrics_fused(heads, 0, 3, cu(g["pcs"]), cu(g["normals"]), cu(g["seg"]), cu(g["bb"]).float(), cu(g["axes"]), cu(g["centers"]), K,
                                      normalize=(tag == "pred_normeig"), details=True)
    r = lambda k: g["%s:%s" % (tag, k)]
    o = out.cpu().numpy()
    assert np.array_equal(det["matching_indices"].cpu().numpy(), r("matching_indices")) and np.array_equal(det["mask"].cpu().numpy() > 0, r("mask") > 0)
    mg = r("mask") > 0
    assert np.array_equal(det["found_centers_mask"].cpu().numpy()[mg] > 0, (r("found_centers_mask") > 0)[mg])
    np.testing.assert_allclose(o[0], r("mIoU"), rtol=1e-4)
    np.testing.assert_allclose(o[1], r("normal_difference"), rtol=1e-4)
    np.testing.assert_allclose(o[2], r("pred_bb_acc"), rtol=1e-6)
    np.testing.assert_allclose(o[3], g["%s:extrusion_difference64" % tag], rtol=1e-4)
    np.testing.assert_allclose(o[4], r("centroid_difference"), rtol=1e-4, atol=1e-8)
    np.testing.assert_allclose(det["predicted_centroids"].cpu().numpy()[mg], r("predicted_centroids")[mg], rtol=1e-4, atol=2e-6)
    sin = np.linalg.norm(np.cross(det["E64"].cpu().numpy(), g["%s:E_AX64" % tag]), axis=-1)
    assert (sin[mg] < 3e-7).all(), sin[mg].max()


def test_five_adam_steps_golden():
    """G12: the loss trajectory of five consecutive reference steps (B=8, N=1024) with the reference's FPS starts and dropout masks
    injected, against the reference's fp32 trajectory AND its float64 twin.  The trajectory is ill-conditioned (a chain of train-mode
    BatchNorms under Adam's sign-like first updates): the reference's own fp32 run is 1e-3 away from its float64 run at step 0,
    3e-2 after ONE update, 5e-2 after two.  Bar: step 0 (pure forward) within 1e-4 of the reference's fp32 losses, with a bit-exact
    matching; step 1 (after one Adam update of every weight) within 5e-3 - six times closer than the reference's fp32 run is to its
    float64 twin there; later steps within twice that fp32-vs-float64 distance (gross-error check of a chaotic trajectory)."""
    g = load_golden("g12_train_5steps")
    B, N, K = 8, 1024, 8
    torch.manual_seed(int(g["seed"]))
    m = backbone(output_sizes=[3, 2 * K]).to(DEV).train()
    step.update_momentum(m, 0.5)
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    z = torch.zeros(B, K, 3, device=DEV)
    batch = (cu(g["pcs"]), cu(g["normals"]), cu(g["seg"]), cu(g["bb"]), z, z)
    gm = torch.Generator().manual_seed(int(g["mask_seed"]))
    ref_gap = np.maximum.accumulate(np.abs(g["adam_losses"] / g["adam_losses64"] - 1).max(1))
    errs = []
    for s in range(int(g["steps"])):
        m.sa1.fps_start, m.sa2.fps_start = t(g["adam_start1"][s]), t(g["adam_start2"][s])
        dm = (torch.rand(B, 128, N, generator=gm) < 0.5).float()
        ck = dm.double()
        np.testing.assert_allclose([ck.sum().item(), ck.abs().sum().item(), (ck * ck).sum().item()], g["dropout_mask_ck"][s])
        m.dropout_mask = dm.permute(0, 2, 1).contiguous()
        out = step.train_step(m, opt, batch, step.StepFlags(K=K), fused=(s % 2 == 1))       # both loss paths along the way
        got = np.array([out[k].item() for k in ("total", "normal", "miou", "bb")])
        errs.append(float(np.abs(got / g["adam_losses"][s] - 1).max()))
        # steps 0 / 1 are the pin; from step 2 on the trajectory is chaotic (the reference's own fp32 run is 3-5 % off its float64 twin) and the
        # bound is a gross-error check at 3x that gap.  Measured: fp32-MFMA kernels 2.4e-7, 1.6e-3, 1.5e-2, 5.4e-2; bf16x3-split kernels
        # 1.5e-7, 2.4e-3, 3.5e-2, 6.0e-2, 1.2e-1 (the oracle's restatement of the reference's own ops: 0, 2e-7, 4e-5, 2e-3, 1.4e-2)
        bound = (1e-4, 5e-3)[s] if s <= 1 else 3 * ref_gap[s]
        assert errs[-1] <= bound, "step %d: |ours/ref32 - 1| = %s, bound %.2e (|ref32/ref64 - 1| so far %s)" % (s, errs, bound, ref_gap)
        if s == 0:      # (after an update the near-random predictions put several IoU costs within rounding of each other)
            assert np.array_equal(out["match"].cpu().numpy(), g["adam_match_%d" % s]), "step %d matching" % s
    print("G12 per-step max relative loss error vs the reference's fp32 run:", errs, "; fp32 reference vs its float64 run:", ref_gap.tolist())
    # parameters after five steps: sum|.| of every tensor (Adam moves each weight by <= 5e-3 in total)
    for (name, p), ck in zip(m.named_parameters(), g["adam_param_ck"]):
        v = p.detach().double()
        assert abs(v.abs().sum().item() - ck[1]) <= 5e-3 * p.numel() + 1e-4 * ck[1], name


def _run(cmd, timeout=900, env=None):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    e.update(env or {})
    return subprocess.run([sys.executable] + cmd, capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=e)


def test_trainer_cli_config0_then_eval_cli(tmp_path):
    """BASELINE configs[0]: the trainer on 4 synthetic shapes, B=2, N=1024, 1 epoch (2 steps: one eager, one graph capture) + a
    third epoch-run that reaches graph replays; first-step scalars are those of the eager step function on the same data and seed;
    the checkpoint has the reference's layout; the eval CLI reads it back and prints the report."""
    logdir = str(tmp_path / "run")
    rep = str(tmp_path / "rep.json")
    out = _run(["-m", "point2cyl_amd.train", "--pred_seg", "--pred_normal", "--pred_bb", "--synthetic", "4", "--batch_size", "2",
                "--num_point", "1024", "--num_epochs", "3", "--save_every", "1", "--logdir", logdir, "--report", rep])
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("Epoch:")]
    assert len(lines) == 6, out.stdout[-2000:]
    vals = np.array([[float(x.split(":")[1]) for x in l.split("|")[2:]] for l in lines])
    assert np.isfinite(vals).all() and (vals[:, 0] > 0).all()
    assert vals[-2:, 0].mean() < vals[:2, 0].mean() + 0.05, vals[:, 0]                 # six Adam steps do not diverge
    r = json.load(open(rep))
    assert r["graph"] and r["graph_captures"] == 1 and r["steps"] == 6
    # the '> Epoch' summaries are the MEANS of their epoch's per-step lines - also in graph mode, where every replay returns the same
    # static scalars tensor (epochs 2 and 3 are graph replays; ADVICE r2: a list of those aliases averaged to the last step's values)
    summ = [l for l in out.stdout.splitlines() if l.startswith("> Epoch")]
    assert len(summ) == 3, out.stdout[-2000:]
    ep_tot = np.array([float(l.split("total:")[1].split("|")[0]) for l in summ])
    np.testing.assert_allclose(ep_tot, vals[:, 0].reshape(3, 2).mean(1), atol=2e-4)
    np.testing.assert_allclose(r["epoch_means"]["total"], vals[:, 0].reshape(3, 2).mean(1), atol=1e-4)
    assert abs(vals[4, 0] - vals[5, 0]) > 1e-5, "the two steps of the last epoch must differ for this check to mean anything"
    ck = torch.load(os.path.join(logdir, "model.pth"), map_location="cpu")
    assert set(ck.keys()) == {"model"} and len(ck["model"]) == 123
    assert os.path.exists(os.path.join(logdir, "checkpoint_0001.pth"))
    g5 = load_golden("g5_backbone_train")
    assert list(ck["model"].keys()) == [str(k) for k in g5["keys"]]
    assert int(ck["model"]["bn1.num_batches_tracked"]) == 6                            # warm-up / capture passes left no trace
    # first logged step == the eager step function on the same seed / data order (torch.randperm of the trainer's generator)
    torch.manual_seed(0)
    m = backbone(output_sizes=[3, 16]).to(DEV).train()
    torch.manual_seed(0)
    ds = synth.SyntheticExtrusionDataset(4, 1024, 8, seed=1234)
    perm = torch.randperm(4)
    items = [ds[int(i)] for i in perm[:2]]
    b = [torch.from_numpy(np.stack([it[j] for it in items])).to(DEV) for j in (0, 1, 2, 3, 6, 8)]
    o = step.compute_losses_fused(m, *b, step.StepFlags())
    # (FPS starts / dropout differ in their draws' position in the stream only if the trainer drew something else first: it does not)
    np.testing.assert_allclose(vals[0, 1:4], [float(o["normal"]), float(o["miou"]), float(o["bb"])], rtol=2e-3)
    # eval CLI on that checkpoint
    ev = _run(["-m", "point2cyl_amd.eval", "--synthetic", "4", "--batch_size", "2", "--num_point", "1024", "--num_sk_point", "256",
               "--logdir", logdir, "--ckpt", "model.pth", "--dump_dir", str(tmp_path / "dump")])
    assert ev.returncode == 0, ev.stderr[-3000:]
    for key in ("Num evaluated= 4", "Mean mIOU= ", "Mean normal angle error (degrees) = ", "Mean base/barrel accuracy= ",
                "Mean extrusion angle error (degrees) = ", "Mean centroid difference = "):
        assert key in ev.stdout, ev.stdout[-1500:]
    nums = [float(l.split("=")[-1]) for l in ev.stdout.splitlines() if l.startswith("Mean ")]
    assert len(nums) == 5 and np.isfinite(nums).all()


def test_trainer_cli_max_steps_ends_like_the_end_of_the_data(tmp_path):
    """--max_steps in the middle of an epoch: the last step's log line, the epoch summary (mean over the steps taken) and model.pth are
    written, as train_sketch.py does."""
    logdir = str(tmp_path / "run")
    out = _run(["-m", "point2cyl_amd.train", "--pred_seg", "--pred_normal", "--pred_bb", "--synthetic", "8", "--batch_size", "2",
                "--num_point", "1024", "--num_epochs", "2", "--max_steps", "3", "--logdir", logdir])
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("Epoch:")]
    assert len(lines) == 3, out.stdout[-2000:]
    vals = np.array([float(l.split("total loss:")[1].split("|")[0]) for l in lines])
    summ = [l for l in out.stdout.splitlines() if l.startswith("> Epoch")]
    assert len(summ) == 1
    np.testing.assert_allclose(float(summ[0].split("total:")[1].split("|")[0]), vals.mean(), atol=2e-4)
    assert os.path.exists(os.path.join(logdir, "model.pth"))


def test_trainer_cli_other_k_runs_the_torch_loss_path_through_the_graph(tmp_path):
    """--K 4: the fused loss kernels are written for K = 8, so the step composes the losses from torch ops (step.compute_losses) - which
    must still go through the HIP-graph path (the label check's device->host read is skipped while capturing) and train."""
    rep = str(tmp_path / "rep.json")
    out = _run(["-m", "point2cyl_amd.train", "--pred_seg", "--pred_normal", "--pred_bb", "--K", "4", "--synthetic", "8", "--batch_size", "4",
                "--num_point", "2048", "--num_epochs", "3", "--logdir", str(tmp_path / "run"), "--report", rep])
    assert out.returncode == 0, out.stderr[-3000:]
    assert "capture failed" not in out.stderr, out.stderr[-2000:]
    r = json.load(open(rep))
    assert r["graph"] and r["graph_captures"] == 1 and r["steps"] == 6
    tot = r["epoch_means"]["total"]
    assert np.isfinite(tot).all() and tot[-1] < tot[0] + 0.05, tot


def test_trainer_graph_and_eager_paths_agree():
    """The trainer's Runner: after an identical first (eager) step, the second step through the HIP-graph path (capture at this step,
    geometry of the current batch computed at capture, next batch's on the forked stream) and launched from Python give the same six
    loss scalars (5e-4; the gradient equality of the two paths is test_b32_n8192_graph_replay_gradients_equal_eager).  Longer
    trajectories are not comparable: fp32 atomics reorder between ANY two runs and Adam amplifies that ~100x per step."""
    from point2cyl_amd import backbone as bbmod, ddp
    from point2cyl_amd.train import Runner
    B, N, K = 4, 2048, 8
    fl = step.StepFlags(K=K, pred_extrusion=True, pred_center=True)
    data = [[synth.make_batch(B, N, K, seed=100 + 10 * s)[i].to(DEV) for i in (0, 1, 2, 3, 6, 8)] for s in range(4)]
    g = torch.Generator().manual_seed(2)
    fixed = {N: torch.randint(0, N, (B,), generator=g), 512: torch.randint(0, 512, (B,), generator=g)}
    orig = bbmod.draw_fps_start
    bbmod.draw_fps_start = lambda n, b: fixed[n].clone()
    try:
        res = {}
        for mode in ("eager", "graph"):
            torch.manual_seed(3)
            m = backbone(output_sizes=fl.pred_sizes()).to(DEV).train()
            m._drop_seed = torch.tensor([987654321], dtype=torch.int64, device=DEV)
            opt = torch.optim.Adam(m.parameters(), lr=1e-3, fused=True)
            run = Runner(m, opt, ddp.FlatGradSync(m.parameters(), 1), fl, DEV, B, N, K, use_graph=(mode == "graph"))
            sc = []
            for s in range(3):
                run.load(data[s], data[s + 1][0])
                sc.append(run.step(0.1 if s == 0 else 0.5, eager=(s == 0)).cpu().numpy())
            res[mode] = np.stack(sc)
            assert run.captures == (1 if mode == "graph" else 0)
            assert int(m.bn1.num_batches_tracked) == 3                          # warm-up / capture passes left no trace
        np.testing.assert_allclose(res["graph"][0], res["eager"][0], rtol=1e-6, atol=1e-7)        # both eager
        np.testing.assert_allclose(res["graph"][1][[0, 1, 2, 3, 5]], res["eager"][1][[0, 1, 2, 3, 5]], rtol=5e-4, atol=1e-6)
        np.testing.assert_allclose(res["graph"][1][4], res["eager"][1][4], rtol=5e-3)     # axis loss: eigenvectors of near-random predictions
        # third step: gross-error check only.  The trajectory is chaotic by now (fp32 atomics reorder between any two runs, Adam amplifies
        # ~100x per step), and the axis loss (index 4: eigenvectors of near-random predictions) moved by 13 % when a change of the
        # fp64 summation order of the BatchNorm sums was tried - it is compared loosely, the total (index 0) contains it
        idx = [1, 2, 3, 5]
        np.testing.assert_allclose(res["graph"][2][idx], res["eager"][2][idx], rtol=5e-2, atol=1e-4)
        np.testing.assert_allclose(res["graph"][2][[0, 4]], res["eager"][2][[0, 4]], rtol=0.5)
    finally:
        bbmod.draw_fps_start = orig


def test_backbone_normal_channel_vs_oracle():
    """backbone(normal_channel=True): six input channels (SA1 groups [dxyz | normals], FP1 concatenates the raw normals as skip
    features - the branches without the linear-before-gather shortcut) against the oracle, train mode, forward + parameter gradients.
    (B = 4: with two clouds SA3 / FP3 would normalise TWO rows per channel, x_hat = +-1 by the sign of a difference - a chain in which
    any two fp32 implementations scatter around float64 by an order of magnitude per tensor.)"""
    B, N, K = 4, 1024, 8
    pcs, nrm = synth.make_batch(B, N, K, seed=606)[:2]
    x = torch.cat([pcs, nrm], -1)
    sd0 = R.make_state_dict((3, 2 * K), normal_channel=True, seed=77)
    torch.manual_seed(77)
    m = backbone(normal_channel=True, output_sizes=[3, 2 * K])
    assert all(torch.equal(m.state_dict()[k], v) for k, v in sd0.items())
    g = torch.Generator().manual_seed(1)
    s1, s2 = torch.randint(0, N, (B,), generator=g), torch.randint(0, 512, (B,), generator=g)
    mask = (torch.rand(B, N, 128, generator=g) < 0.5).float()
    res = {}
    for dt in (torch.float32, torch.float64):
        sd = {k: (v.clone().to(dt) if v.dtype.is_floating_point else v.clone()) for k, v in sd0.items()}
        leaves = {k: v.requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and "running" not in k}
        outs, aux = R.backbone_forward(sd, x.to(dt), [s1, s2], mask.to(dt), training=True, momentum=0.5, geom="c", return_aux=True)
        ((outs[0] ** 2).mean() + (outs[1] ** 2).mean() * 0.1 + outs[1][..., 0].mean()).backward()
        res[dt] = ([o.detach() for o in outs], {k: v.grad for k, v in leaves.items()}, aux)
    m = m.to(DEV).train()
    step.update_momentum(m, 0.5)
    m.sa1.fps_start, m.sa2.fps_start = s1, s2
    m.dropout_mask = mask
    X, Wr = m(x.to(DEV))
    ((X ** 2).mean() + (Wr ** 2).mean() * 0.1 + Wr[..., 0].mean()).backward()
    o32, g32, aux = res[torch.float32]
    o64, g64, _ = res[torch.float64]
    assert torch.equal(m.sa1.last_aux["group_idx"].cpu().long(), aux["sa1"]["group_idx"])
    for mine, r32, r64 in ((X, o32[0], o64[0]), (Wr, o32[1], o64[1])):
        ref_err = float((r32.double() - r64).abs().max())
        assert float((mine.detach().cpu().double() - r64).abs().max()) <= max(1e-4, 3 * ref_err)
    rel_med = float(np.median([np.linalg.norm(g32[n].double().numpy() - g64[n].numpy()) / np.linalg.norm(g64[n].numpy()) for n, _ in m.named_parameters()
                               if not (n.endswith(".bias") and ("mlp_convs" in n or n == "fc1.bias"))]))
    for name, p in m.named_parameters():
        r32, r64 = g32[name].double().numpy(), g64[name].numpy()
        got = p.grad.cpu().double().numpy().reshape(r64.shape)
        if name.endswith(".bias") and ("mlp_convs" in name or name == "fc1.bias"):
            assert np.abs(got).max() == 0.0
            continue
        # per tensor: 3x the oracle's own fp32 distance from float64 on this tensor (max-abs OR norm: one max-pool winner that resolves
        # differently moves max-abs alone), or the oracle's MEDIAN relative distance over all tensors (small batch: the per-tensor fp32
        # error of any implementation scatters around that level)
        a = np.abs(got - r64).max() / (3 * np.abs(r32 - r64).max() + 1e-6 * np.abs(r64).max())
        b = (np.linalg.norm(got - r64) / np.linalg.norm(r64)) / (max(3 * np.linalg.norm(r32 - r64) / np.linalg.norm(r64), rel_med) + 1e-6)
        assert min(a, b) <= 1.0, (name, a, b)


def test_backbone_eval_mode_backward_vs_oracle():
    """Backward through the whole backbone in EVAL mode against the oracle with training=False (float64): EVERY parameter gradient at 1e-5
    of its norm, the set-abstraction layers below the max-pools included (they were held at 5e-2 because a handful of the
    pooled maxima have two candidates within fp32 rounding of each other and a winner that resolves differently moves one gradient
    entry to another row - documented deviation (vi)).  The oracle is run with `forced_winners` = the rows the
    kernels stored (`last_aux["pool_arg"]`), and the test proves that forcing them is harmless: wherever the kernels' winner differs from
    the float64 oracle's own arg-max, the forced row's value lies within fp32 rounding (2e-6 relative) of the oracle's maximum - a real
    mis-selection would show up as a large gap.  (The stack-level eval backward is pinned by test_mlp_stack_eval_mode_forward_backward.)"""
    B, N, K = 3, 1024, 8
    pcs = synth.make_batch(B, N, K, seed=707)[0]
    torch.manual_seed(78)
    m = backbone(output_sizes=[3, 2 * K])
    with torch.no_grad():                               # running statistics away from (0, 1), scales of both signs
        for name, mod in m.named_modules():
            if isinstance(mod, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
                mod.running_mean.normal_(0, 0.3)
                mod.running_var.uniform_(0.5, 2.0)
                mod.weight.uniform_(0.5, 1.5)
                mod.bias.normal_(0, 0.2)
        m.sa1.mlp_bns[1].weight[5] = -0.9
        m.sa2.mlp_bns[2].weight[7] = -0.8              # a NEGATIVE scale on a pooled layer: its winner is the group's smallest pre-BN value
    sd0 = {k: v.detach().clone() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(2)
    s1, s2 = torch.randint(0, N, (B,), generator=g), torch.randint(0, 512, (B,), generator=g)
    mask = (torch.rand(B, N, 128, generator=g) < 0.5).float()
    m = m.to(DEV).eval()
    m.sa1.fps_start, m.sa2.fps_start = s1, s2
    m.dropout_mask = mask
    X, Wr = m(pcs.to(DEV))
    ((X ** 2).mean() + (Wr ** 2).mean() * 0.1 + Wr[..., 0].mean()).backward()
    winners = {"sa1": m.sa1.last_aux["pool_arg"].cpu().long().view(B, 512, 128), "sa2": m.sa2.last_aux["pool_arg"].cpu().long().view(B, 128, 256),
               "sa3": m.sa3.last_aux["pool_arg"].cpu().long().view(B, 1, 1024)}
    sd = {k: (v.clone().double() if v.dtype.is_floating_point else v.clone()) for k, v in sd0.items()}
    leaves = {k: v.requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and "running" not in k}
    o64, aux = R.backbone_forward(sd, pcs.double(), [s1, s2], mask.double(), training=False, momentum=0.5, geom="c", return_aux=True,
                                  forced_winners=winners)
    ((o64[0] ** 2).mean() + (o64[1] ** 2).mean() * 0.1 + o64[1][..., 0].mean()).backward()
    n_diff = 0
    for lvl in ("sa1", "sa2", "sa3"):
        own, gap, top = aux[lvl]["own_winners"], aux[lvl]["pool_gap"], aux[lvl]["pool_max"]
        differs = own != winners[lvl]
        n_diff += int(differs.sum())
        assert float(gap.min()) >= 0.0
        # a forced row is either the oracle's own winner, or its value is within fp32 rounding of the maximum (a near-tie, or both clamped to 0)
        assert float((gap / top.abs().clamp_min(1.0)).max()) <= 2e-6, (lvl, float(gap.max()))
        # (groups whose rows are all clamped to 0 have 64 equal "winners": the index may differ freely there, the gap is exactly 0)
    for mine, r64 in ((X, o64[0]), (Wr, o64[1])):
        assert float((mine.detach().cpu().double() - r64.detach()).abs().max()) <= 1e-5
    for k in ("sa1.mlp_bns.0.running_mean", "bn1.running_var"):
        assert torch.equal(m.state_dict()[k].cpu(), sd0[k]), "eval mode must not touch the running statistics"
    nz_bias = 0
    worst = 0.0
    for name, p in m.named_parameters():
        r64 = leaves[name].grad.numpy()
        got = p.grad.cpu().double().numpy().reshape(r64.shape)
        if name.endswith(".bias") and ("mlp_convs" in name or name == "fc1.bias"):
            nz_bias += int(np.abs(r64).max() > 0)
        rel = np.linalg.norm(got - r64) / max(np.linalg.norm(r64), 1e-30)
        worst = max(worst, rel)
        assert rel <= 1e-5, (name, rel)
    assert nz_bias >= 15, "the conv biases in front of an eval-mode BatchNorm have real gradients"
    print("eval-mode backward with forced winners: worst relative parameter-gradient error %.2e; %d pooled entries whose winner differs "
          "from the float64 arg-max" % (worst, n_diff))


def test_hungarian_rejects_out_of_range_labels():
    """losses.py:36-38 builds eye(n_gt+1)[I_gt]: the reference raises for labels >= K (n_gt > K makes matching_indices[b, :n_gt]
    overflow); the device path must not return a plausible-looking matching for them."""
    from point2cyl_amd import losses
    W = torch.rand(2, 256, 8, device=DEV)
    I = torch.randint(0, 8, (2, 256), device=DEV)
    losses.hungarian_matching(W, I)
    I[1, 5] = 8
    with pytest.raises((RuntimeError, ValueError)):
        losses.hungarian_matching(W, I)
        torch.cuda.synchronize()
    # compute_all_losses validates WITHOUT a device->host sync (ops.check_labels_deferred): the verdict arrives through pinned memory and
    # the next call - or ops.flush_label_check() - raises
    X = F.normalize(torch.randn(2, 256, 3, device=DEV), dim=-1)
    Wn = torch.softmax(W, -1)
    # (the first SYNC_FIRST calls of a process validate synchronously - the reference raises in the offending call - then deferred)
    ops.check_labels_deferred.calls = 0
    I[1, 5] = 9
    with pytest.raises(ValueError):
        losses.compute_all_losses(None, Wn, I, X, X, 1.0, 1.0)
    ops.check_labels_deferred.calls = ops.check_labels_deferred.SYNC_FIRST
    I[1, 5] = 3
    ops.flush_label_check()
    losses.compute_all_losses(None, Wn, I, X, X, 1.0, 1.0)
    ops.flush_label_check()                                   # in range: nothing pending
    I[1, 5] = 9
    losses.compute_all_losses(None, Wn, I, X, X, 1.0, 1.0)
    with pytest.raises(ValueError):
        ops.flush_label_check()
    I[1, 5] = 3
    losses.compute_all_losses(None, Wn, I, X, X, 1.0, 1.0)    # the flag was cleared with the raise
    ops.flush_label_check()


def test_eval_sketch_fit_losses_vs_oracle():
    """eval.py:459-590 (default branch): projection of the predicted barrels -> PointNetEncoder -> ImplicitNet on the gt-label
    projection along the predicted axes (per-cylinder fitting loss) and on all points (global fitting loss), against the same
    composition of the oracle's pieces (projection with the recorded sampling draws, encoder and decoder restatements).  The rotation
    matrix inside the projection is the one part of this chain whose oracle is not pinned by the reference (torchgeometry absent)."""
    from point2cyl_amd import fitting
    from point2cyl_amd.implicit import ImplicitNet
    from point2cyl_amd.sketch import PointNetEncoder
    g = load_golden("g13_eval_flow")
    B, N, K, S = 3, 1024, 8, 128
    pcs, nrm, seg, bb, axes, cen = (t(g[k]) for k in ("pcs", "normals", "seg", "bb", "axes", "centers"))
    fl = p2c_eval.EvalFlags(K=K, num_sk_point=S)
    m = p2c_eval.eval_metrics(cu(g["X_head"]), cu(g["W_raw"]), cu(pcs), cu(nrm), cu(seg), cu(bb).float(), cu(axes), cu(cen), fl)
    torch.manual_seed(12)
    dec = ImplicitNet(d_in=2 + 64, dims=[128] * 4, skip_in=[2], geometric_init=True, radius_init=1, beta=100).to(DEV).eval()
    enc = PointNetEncoder(64, 2, with_normals=True).to(DEV).eval()
    with torch.no_grad():                        # running statistics away from (0, 1) so that eval-mode BatchNorm is not the identity
        for mod in enc.modules():
            if isinstance(mod, torch.nn.BatchNorm1d):
                mod.running_mean.normal_(0, 0.1)
                mod.running_var.uniform_(0.5, 1.5)
    # the sampling draws of the three projections, made here and handed to both sides
    gen = torch.Generator().manual_seed(4)
    W = m["W"].cpu()
    Wre = torch.gather(W, 2, m["matching_indices"].cpu().unsqueeze(1).expand(B, N, K)) * m["mask"].cpu().unsqueeze(1)
    label = Wre.argmax(-1)
    pbb = m["pred_bb_label"].cpu()

    def draws(seg_, bb_):
        barrel = F.one_hot(seg_, K).bool() & (bb_ == 0).unsqueeze(-1)
        cnt = barrel.sum(1)
        r = torch.zeros(B, K, S, dtype=torch.int64)
        d = {}
        for k in range(K):
            for b in range(B):
                if int(cnt[b, k]) > 1:
                    r[b, k] = torch.randint(0, int(cnt[b, k]), (S,), generator=gen)
                    d[(k, b)] = r[b, k]
        return r, d
    r1, d1 = draws(label, pbb)
    r2, d2 = draws(seg, bb)
    E, C = m["E_AX"], m["predicted_centroids"]
    # device side (the three projections of sketch_fit_losses with the draws injected)
    with torch.no_grad():
        ppc, pn, sc, _ = fitting.sketch_implicit_projection2(cu(pcs), m["X"], cu(label), cu(pbb), E, C, S, rand_idx=r1)
        lat = enc(torch.cat([(ppc / sc.unsqueeze(-1).unsqueeze(-1)).reshape(B * K, S, 2), pn.reshape(B * K, S, 2)], -1))
        p2, _, _, f2 = fitting.sketch_implicit_projection2(cu(pcs), cu(nrm), cu(seg), cu(bb), E, C, S, rand_idx=r2)
        from point2cyl_amd.implicit import add_latent
        sk = dec(add_latent((p2 / sc.unsqueeze(-1).unsqueeze(-1)).reshape(B * K, S, 2), lat)).reshape(K, B, S)
        pm = (m["mask"].T * f2.T).unsqueeze(-1)
        cyl = (sk * pm).abs().permute(1, 0, 2).mean(-1).reshape(B, -1).sum(1) / (cu(seg).max(1)[0] + 1).float()
    # oracle side
    Ec, Cc, Xc = E.cpu(), C.cpu(), m["X"].cpu()
    Pp, Xp, scl, _ = R.sketch_implicit_projection(pcs, Xc, label, pbb, Ec, Cc, d1, S)
    sd_e = {k: v.detach().cpu().clone() for k, v in enc.state_dict().items()}
    lat_r = R.pointnet_encoder_forward(sd_e, torch.cat([(Pp / scl.unsqueeze(-1).unsqueeze(-1)).reshape(B * K, S, 2), Xp.reshape(B * K, S, 2)], -1),
                                       training=False)
    P2, _, _, F2 = R.sketch_implicit_projection(pcs, nrm, seg, bb, Ec, Cc, d2, S)
    sd_d = {k: v.detach().cpu() for k, v in dec.state_dict().items()}
    q = (P2 / scl.unsqueeze(-1).unsqueeze(-1)).reshape(B * K, S, 2)
    inp = torch.cat([lat_r.unsqueeze(1).repeat(1, S, 1).reshape(B * K * S, -1), q.reshape(B * K * S, 2)], 1)
    sk_r = R.implicit_net_forward(sd_d, inp, skip_in=(2,)).reshape(K, B, S)
    pm_r = (m["mask"].cpu().T * F2.T).unsqueeze(-1)
    cyl_r = (sk_r * pm_r).abs().permute(1, 0, 2).mean(-1).reshape(B, -1).sum(1) / (seg.max(1)[0] + 1).float()
    np.testing.assert_allclose(lat.cpu().numpy(), lat_r.numpy(), rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(cyl.cpu().numpy(), cyl_r.numpy(), rtol=2e-4, atol=1e-6)
    # and the packaged function runs end to end (its own random draws): finite, non-negative
    c2, g2 = p2c_eval.sketch_fit_losses(m, cu(pcs), cu(nrm), cu(seg), cu(bb).float(), dec, enc, fl)
    assert c2.shape == (B,) and g2.shape == (B,) and bool(torch.isfinite(c2).all()) and bool(torch.isfinite(g2).all())
    assert float(c2.min()) >= 0 and float(g2.min()) >= 0


def test_with_sketch_trainer_cli(tmp_path):
    """train_Point2Cyl.py counterpart: two steps at a small size with every loss switched on; the log lines, the checkpoint triple
    {"model", "implicit_net", "pn_encoder"} with the reference's key sets, and --is_pc_init from a without-sketch checkpoint."""
    pc = str(tmp_path / "pc")
    out = _run(["-m", "point2cyl_amd.train", "--pred_seg", "--pred_normal", "--pred_bb", "--synthetic", "4", "--batch_size", "2", "--num_point", "1024",
                "--num_epochs", "1", "--save_every", "1", "--logdir", pc, "--quiet"])
    assert out.returncode == 0, out.stderr[-2000:]
    logdir = str(tmp_path / "sk")
    out = _run(["-m", "point2cyl_amd.train_sketch", "--pred_seg", "--pred_normal", "--pred_bb", "--pred_extrusion", "--pred_center", "--is_pc_train",
                "--is_im_train", "--with_im_loss", "--is_pc_init", "--pc_logdir", pc, "--synthetic", "4", "--batch_size", "2", "--num_point", "1024",
                "--num_sk_point", "256", "--num_epochs", "1", "--save_every", "1", "--logdir", logdir, "--im_logdir", str(tmp_path / "none")])
    assert out.returncode == 0, out.stderr[-3000:]
    assert "3D model loaded." in out.stdout and "WARNING" in out.stdout
    im = [l for l in out.stdout.splitlines() if "latent loss" in l]
    pcl = [l for l in out.stdout.splitlines() if "mIOU loss" in l]
    assert len(im) == 2 and len(pcl) == 2
    vals = np.array([float(x.split(":")[1]) for l in im + pcl for x in l.split("|")[2:]])
    assert vals.size == 2 * 5 + 2 * 6 and np.isfinite(vals).all()
    ck = torch.load(os.path.join(logdir, "model.pth"), map_location="cpu")
    assert set(ck.keys()) == {"model", "implicit_net", "pn_encoder"} and len(ck["model"]) == 123
    assert "lin0.weight" in ck["implicit_net"] and "mlp1.0.weight" in ck["pn_encoder"] and "fc.weight" in ck["pn_encoder"]
    assert os.path.exists(os.path.join(logdir, "checkpoint_0001.pth"))


@pytest.mark.parametrize("flags,enc_in", [(["--use_gt_im", "--is_im_train"], 4),
                                          (["--use_gt_im", "--is_im_train", "--use_whole_pc", "--use_extrusion_axis_feat"], 7),
                                          (["--is_pc_train", "--is_im_train", "--use_whole_pc", "--use_extrusion_axis_feat", "--pred_extrusion",
                                            "--is_implicitnet_train"], 7)])
def test_with_sketch_trainer_cli_variants(tmp_path, flags, enc_in):
    """The reference trainer's other encoder inputs through the CLI (train_Point2Cyl.py:268-276, :519-600): --use_gt_im (no backbone pass),
    --use_extrusion_axis_feat (7-channel encoder) and --is_implicitnet_train (parsed and never read by the reference: accepted, no effect);
    two steps each, finite losses, the encoder of the checkpoint has the variant's input width."""
    logdir = str(tmp_path / "sk")
    out = _run(["-m", "point2cyl_amd.train_sketch", "--pred_seg", "--pred_normal", "--pred_bb", "--with_im_loss", "--synthetic", "4", "--batch_size", "2",
                "--num_point", "1024", "--num_sk_point", "256", "--num_epochs", "1", "--save_every", "1", "--logdir", logdir,
                "--im_logdir", str(tmp_path / "none")] + flags)
    assert out.returncode == 0, out.stderr[-3000:]
    im = [l for l in out.stdout.splitlines() if "latent loss" in l]
    assert len(im) == 2
    vals = np.array([float(x.split(":")[1]) for l in im for x in l.split("|")[2:]])
    assert np.isfinite(vals).all() and vals[0] > 0
    ck = torch.load(os.path.join(logdir, "model.pth"), map_location="cpu")
    assert ck["pn_encoder"]["mlp1.0.weight"].shape[1] == enc_in
    # inconsistent combinations are refused with the reference line that makes them meaningless
    bad = _run(["-m", "point2cyl_amd.train_sketch", "--pred_seg", "--pred_normal", "--pred_bb", "--use_gt_im", "--is_pc_train", "--synthetic", "4",
                "--logdir", logdir])
    assert bad.returncode != 0 and "use_gt_im" in (bad.stderr + bad.stdout)


def test_multi_tensor_adam_matches_torch():
    """point2cyl_amd.optim.Adam (one launch over all tensors) against torch.optim.Adam, five steps, odd sizes, two parameter groups
    with their own learning rates, a parameter without gradient."""
    from point2cyl_amd import optim
    g = torch.Generator().manual_seed(0)
    shapes = [(1,), (7, 3), (1023,), (1025,), (70000,), (128, 128, 1), (19, 128)]
    base = [torch.randn(s, generator=g) for s in shapes]
    mine = [b.clone().to(DEV).requires_grad_(True) for b in base]
    ref = [b.clone().to(DEV).requires_grad_(True) for b in base]
    idle_m, idle_r = torch.zeros(5, device=DEV, requires_grad=True), torch.zeros(5, device=DEV, requires_grad=True)
    o1 = optim.Adam([{"params": mine[:4] + [idle_m], "lr": 1e-3}, {"params": mine[4:], "lr": 3e-4}])
    o2 = torch.optim.Adam([{"params": ref[:4] + [idle_r], "lr": 1e-3}, {"params": ref[4:], "lr": 3e-4}])
    for s in range(5):
        grads = [torch.randn(b.shape, generator=g).to(DEV) * (10.0 ** (s - 2)) for b in base]
        for p, q, gr in zip(mine, ref, grads):
            p.grad = gr.clone()
            q.grad = gr.clone()
        if s == 3:
            o1.param_groups[0]["lr"] = o2.param_groups[0]["lr"] = 5e-4
        o1.step()
        o2.step()
    for p, q in zip(mine, ref):
        np.testing.assert_allclose(p.detach().cpu().numpy(), q.detach().cpu().numpy(), rtol=2e-6, atol=1e-7)
    assert float(idle_m.abs().max()) == 0.0


def _fresh_backbone(seed=11):
    torch.manual_seed(seed)
    return backbone(output_sizes=[3, 16]).to(DEV).train()


def test_autograph_module_forward_backward_equals_eager_launches():
    """point2cyl_amd/autograph.py: `model(pcs)` of an unchanged caller replays a per-shape HIP graph of the forward and, in the backward of
    the caller's loss, a graph of the module's backward.  Against the same module launched kernel by kernel (autograph off) on the same RNG
    state - same FPS draws, same dropout counter -, over three consecutive calls: head outputs, parameter gradients, BatchNorm running
    statistics and counters; then the gradient-accumulation path (two backwards without clearing .grad) and the no-grad / eval forward."""
    from point2cyl_amd import autograph
    B, N = 4, 2048
    pcs = synth.make_batch(B, N, 8, seed=77)[0].float().to(DEV)
    tgt = torch.randn(B, N, 19, device=DEV)

    def run(enabled, calls=3):
        old = autograph.ENABLED
        autograph.ENABLED = enabled
        try:
            model = _fresh_backbone()
            torch.manual_seed(5)
            outs, grads = [], []
            for _ in range(calls):
                for p in model.parameters():
                    p.grad = None
                X, W = model(pcs)
                loss = ((torch.cat([X, W], -1) - tgt) ** 2).mean()
                loss.backward()
                outs.append(torch.cat([X, W], -1).detach().clone())
                grads.append({k: p.grad.detach().clone() for k, p in model.named_parameters()})
            bufs = {k: v.detach().clone() for k, v in model.named_buffers()}
            return model, outs, grads, bufs
        finally:
            autograph.ENABLED = old

    m0, o0, g0, b0 = run(False)
    m1, o1, g1, b1 = run(True)
    assert len(autograph._state(m1)["graphs"]) == 1 and not autograph._state(m1)["failed"]
    assert len(autograph._state(m0)["graphs"]) == 0
    for i in range(3):
        assert float((o0[i] - o1[i]).abs().max()) <= 2e-5 * float(o0[i].abs().max()), i
        gmax = max(float(v.norm()) for v in g0[i].values())
        for k in g0[i]:
            d, s = float((g0[i][k] - g1[i][k]).norm()), float(g0[i][k].norm())
            assert d <= 2e-4 * s + 1e-6 * gmax, (i, k, d, s)       # (a BatchNorm shift in front of another train-mode BatchNorm has a zero gradient: rounding noise)
    for k in b0:
        if "num_batches_tracked" in k:
            assert int(b0[k]) == int(b1[k]) == 3, k          # the capture's warm-up passes leave no trace
        else:
            np.testing.assert_allclose(b1[k].cpu().numpy(), b0[k].cpu().numpy(), rtol=2e-5, atol=1e-7, err_msg=k)
    # gradient accumulation: a second backward without clearing .grad adds (the static gradient tensors are not aliased by .grad afterwards)
    torch.manual_seed(9)
    for p in m1.parameters():
        p.grad = None
    X, W = m1(pcs); ((X ** 2).mean() + (W ** 2).mean()).backward()
    first = {k: p.grad.detach().clone() for k, p in m1.named_parameters()}
    X, W = m1(pcs); ((X ** 2).mean() + (W ** 2).mean()).backward()
    second_alone = {}
    acc = {k: p.grad.detach().clone() for k, p in m1.named_parameters()}
    for p in m1.parameters():
        p.grad = None
    # (the second call drew new FPS starts / dropout bits, so compare against accumulated - first, recomputed: it must be a plausible gradient)
    for k in first:
        second_alone[k] = acc[k] - first[k]
        assert torch.isfinite(acc[k]).all()
    w = "fp1.mlp_convs.2.weight"
    assert float(second_alone[w].norm()) > 0.2 * float(first[w].norm()) and float((acc[w] - first[w]).norm()) > 0
    # forward only: no_grad in train mode and eval mode reuse / build their own graphs, outputs close to the eager forward
    autograph.ENABLED = False
    try:
        torch.manual_seed(21); m0.eval()
        with torch.no_grad():
            e = torch.cat(m0(pcs), -1).clone()
    finally:
        autograph.ENABLED = True
    m1.load_state_dict(m0.state_dict()); m1.eval()
    m1._drop_seed.copy_(m0._drop_seed - (0x9E3779B97F4A7C15 % (2 ** 62)))
    torch.manual_seed(21)
    with torch.no_grad():
        g = torch.cat(m1(pcs), -1).clone()
    assert float((e - g).abs().max()) <= 2e-5 * float(e.abs().max())
    # a module that has ALREADY been through eager backward passes on the default stream (its parameters own AccumulateGrad nodes bound to
    # that stream, and the old outputs keep them alive) must still capture: the graphs run on aliases of the parameters
    m0.train()
    torch.manual_seed(33)
    X, W = m0(pcs)
    for p in m0.parameters():
        p.grad = None
    (X.square().mean() + W.square().mean()).backward()
    assert not autograph._state(m0)["failed"] and len(autograph._state(m0)["graphs"]) >= 1
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m0.parameters())
    autograph.reset()


ROUTE_FLAGS = [("USE_FUSED_BWD", "P2C_FUSED_BWD"), ("USE_FUSED256", "P2C_FUSED256"), ("USE_DUAL_BWD", "P2C_DUAL_BWD"), ("USE_NARROW_BWD", "P2C_NARROW_BWD"),
               ("USE_POOL_ALG", "P2C_POOL_ALG"), ("USE_POOL_EPI", "P2C_POOL_EPI"), ("USE_FOLD0", "P2C_FOLD0"), ("USE_PRE_LINEAR", "P2C_PRE_LINEAR"),
               ("USE_CSR_BWD", "P2C_CSR_BWD"), ("USE_STAGED_WEIGHTS", "P2C_STAGE_WEIGHTS")]


@pytest.mark.parametrize("flag,env", ROUTE_FLAGS)
def test_alternate_route_equals_default_route(flag, env):
    """every route switch the host layer still reads from the environment (README table: `env`=0 at load time sets
    ops.`flag` False) has a run behind it.  The switched-off form sends the layers it governs down the GENERIC route of this library (the
    tiled GEMM pair instead of the fused backward, the pooling pass instead of the GEMM epilogue, torch-side weight copies instead of the
    staging launch, ...) - the same routes other shapes take by themselves.  One training-step forward + backward (fused losses) at
    B = 4 x N = 2048 on recorded FPS starts and a recorded dropout mask, default route against the switched route: matching and labels
    identical, the four loss scalars at 1e-5, every parameter gradient at 2e-4 of its norm (+ 1e-6 of the largest: zero gradients)."""
    B, N, K = 4, 2048, 8
    pcs, nrm, seg, bb, _, _, axes, _, cen = synth.make_batch(B, N, K, seed=515)
    batch = tuple(x.to(DEV) for x in (pcs.float(), nrm.float(), seg, bb, axes.float(), cen.float()))
    g = torch.Generator().manual_seed(4)
    s1, s2 = torch.randint(0, N, (B,), generator=g), torch.randint(0, 512, (B,), generator=g)
    mask = (torch.rand(B, N, 128, generator=g) < 0.5).float()
    fl = step.StepFlags(K=K)
    assert getattr(ops, flag) is True, "the default route must be on (is %s set in the environment of the test run?)" % env

    def run():
        torch.manual_seed(21)
        m = backbone(output_sizes=[3, 2 * K]).to(DEV).train()
        m.sa1.fps_start, m.sa2.fps_start, m.dropout_mask = s1, s2, mask
        out = step.compute_losses_fused(m, *batch, fl)
        step.backward(out)
        torch.cuda.synchronize()
        return ([float(out[k]) for k in ("total", "normal", "miou", "bb")], out["match"].cpu(), None,
                {k: p.grad.detach().clone() for k, p in m.named_parameters()},
                {k: v.detach().clone() for k, v in m.named_buffers()})

    ref = run()
    setattr(ops, flag, False)
    try:
        alt = run()
    finally:
        setattr(ops, flag, True)
    np.testing.assert_allclose(alt[0], ref[0], rtol=1e-5)
    assert torch.equal(alt[1], ref[1])
    gmax = max(float(v.norm()) for v in ref[3].values())
    for k in ref[3]:
        d, n_ = float((alt[3][k] - ref[3][k]).norm()), float(ref[3][k].norm())
        # A switch that changes the FORWARD arithmetic of a layer (the folded first layer rebuilds Y0 from moments, the linear-before-gather
        # commutation sums in another order, the pooled epilogue picks winners from tile extremes) moves activations by fp32 rounding, and
        # 17 chained train-mode BatchNorms + ReLU decisions turn that into ~0.6 % on the early layers' gradients - the reference's own fp32
        # run is 0.7 % from its float64 twin there (test_train_step_golden).  Backward-only switches keep the forward bit-identical: 2e-4.
        tol = 2e-2 if flag in ("USE_FOLD0", "USE_PRE_LINEAR", "USE_POOL_EPI") else 2e-4
        assert d <= tol * n_ + 1e-6 * gmax, (flag, k, d, n_)
    fwd_switch = flag in ("USE_FOLD0", "USE_PRE_LINEAR", "USE_POOL_EPI")
    for k in ref[4]:
        if ref[4][k].dtype.is_floating_point:
            np.testing.assert_allclose(alt[4][k].cpu().numpy(), ref[4][k].cpu().numpy(), rtol=2e-4 if fwd_switch else 2e-5,
                                       atol=2e-6 if fwd_switch else 1e-7, err_msg=k)
        else:
            assert torch.equal(alt[4][k], ref[4][k]), k


def test_autograph_train_eval_train_with_cache_flush_keeps_counter_tables_alive():
    """ADVICE r5 (ops.py WeightStage._counter_table): the staging launch advances num_batches_tracked / the dropout counter through a device
    table of (address, increment) pairs whose ADDRESS is baked into the captured graph; the live counter set differs between train and eval
    mode.  train forward+backward (captures graph A) -> eval forward under no_grad (another table) -> torch.cuda.empty_cache() + allocator
    churn -> train again (replays graph A): every counter must advance exactly as a never-graphed module's does, nothing else may move,
    and the stage must still hold one table per mode."""
    from point2cyl_amd import autograph
    B, N = 2, 1024
    pcs = synth.make_batch(B, N, 8, seed=79)[0].float().to(DEV)

    def run(enabled):
        old = autograph.ENABLED
        autograph.ENABLED = enabled
        try:
            m = _fresh_backbone(4)
            torch.manual_seed(3)
            trail = []
            for phase in ("train", "eval", "train", "eval", "train"):
                if phase == "train":
                    m.train()
                    for p in m.parameters():
                        p.grad = None
                    X, W = m(pcs)
                    (X.square().mean() + W.square().mean()).backward()
                else:
                    m.eval()
                    with torch.no_grad():
                        X, W = m(pcs)
                    torch.cuda.synchronize()
                    torch.cuda.empty_cache()
                    junk = [torch.full((257 * (i + 1),), 7, dtype=torch.int64, device=DEV) for i in range(64)]     # reuse whatever was freed
                    del junk
                torch.cuda.synchronize()
                trail.append((phase, {k: int(v) for k, v in m.named_buffers() if "num_batches_tracked" in k}, int(m._drop_seed),
                              bool(torch.isfinite(X).all() and torch.isfinite(W).all())))
            return m, trail
        finally:
            autograph.ENABLED = old

    m0, t0 = run(False)
    m1, t1 = run(True)
    assert not autograph._state(m1)["failed"]
    for (ph, n0, d0, ok0), (_, n1, d1, ok1) in zip(t0, t1):
        assert ok0 and ok1, ph
        assert n0 == n1, (ph, {k: (n0[k], n1[k]) for k in n0 if n0[k] != n1[k]})
    assert set(t1[-1][1].values()) == {3}                    # three train-mode forwards; the eval forwards advance none of them
    # the dropout counter advances once per forward in both modes, by the same stride graphed or not
    assert [t[2] - t1[0][2] for t in t1] == [t[2] - t0[0][2] for t in t0]
    stages = [s_ for s_ in vars(m1).values() if hasattr(s_, "ctables")]
    assert stages and all(len(s_.ctables) >= 2 for s_ in stages), "one counter table per mode must stay alive"
    # partially frozen BatchNorm (backbone.py): model.train(); model.sa1.eval() - sa1's counters and running statistics stand
    # still, as in torch, everyone else's advance; graphed and eager alike
    for m in (m0, m1):
        m.train()
        m.sa1.eval()
        before = {k: v.clone() for k, v in m.named_buffers()}
        old = autograph.ENABLED
        autograph.ENABLED = m is m1
        try:
            X, W = m(pcs)
            (X.square().mean() + W.square().mean()).backward()
        finally:
            autograph.ENABLED = old
        torch.cuda.synchronize()
        for k, v in m.named_buffers():
            if k.startswith("sa1."):
                assert torch.equal(v, before[k]), k
            elif "num_batches_tracked" in k:
                assert int(v) == int(before[k]) + 1, k
            elif "running_mean" in k:
                assert not torch.equal(v, before[k]), k


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_add_noise_on_device_is_the_host_add_noise_bit_for_bit(dtype):
    """fitting.add_noise_on_device (the evaluation loader's --add_noise): the host generator's draws and the reference's float64 multiply-add
    (data_utils.py:84-96) with the (B, N, 3) arithmetic on the device - the same float64 values as fitting.add_noise (G14-pinned) from the
    same NumPy state, hence the same float32 clouds after eval.py:254's rounding."""
    from point2cyl_amd import fitting
    g = torch.Generator().manual_seed(5)
    pcs = torch.randn(3, 4096, 3, generator=g, dtype=torch.float64).to(dtype)
    nrm = torch.nn.functional.normalize(torch.randn(3, 4096, 3, generator=g, dtype=torch.float64), dim=-1).to(dtype)
    np.random.seed(1234)
    want = fitting.add_noise(pcs, nrm, sigma=0.02)
    np.random.seed(1234)
    got = fitting.add_noise_on_device(pcs.to(DEV), nrm.to(DEV), sigma=0.02)
    assert want.dtype == got.dtype == torch.float64
    assert torch.equal(want, got.cpu())
    assert torch.equal(want.float(), got.float().cpu())


def test_batched_eval_batchnorm_affine_equals_the_per_layer_launches_and_follows_in_place_updates():
    """ops.BNEvalStage / p2c_bn_eval_affine_batch_f32: an inference forward computes the affine of all 17 eval-mode BatchNorms in ONE launch
    at its top.  Bit-identical heads to the per-layer p2c_bn_finalize_f32 form (P2C_BN_EVAL_BATCH=0), eager and as a replayed HIP graph,
    over a sequence of forwards between which the BatchNorm parameters and running statistics change IN PLACE (scaled; load_state_dict)
    and once get NEW storage - the captured launch reads the live tensors.  The launch counts say which form ran."""
    from point2cyl_amd import autograph
    from point2cyl_amd._lib import PROFILE
    B, N = 2, 2048
    pcs = synth.make_batch(B, N, 8, seed=91)[0].float().to(DEV)

    def bn_mods(m):
        return [x for x in m.modules() if isinstance(x, torch.nn.modules.batchnorm._BatchNorm)]

    def run(batched, graphed, profile=False):
        old = (ops.USE_BN_EVAL_BATCH, autograph.ENABLED)
        ops.USE_BN_EVAL_BATCH, autograph.ENABLED = batched, graphed
        try:
            m = _fresh_backbone(4)
            g = torch.Generator().manual_seed(11)
            with torch.no_grad():
                for x in bn_mods(m):
                    C = x.num_features
                    x.running_mean.copy_((torch.randn(C, generator=g) * 0.2).to(DEV))
                    x.running_var.copy_((torch.rand(C, generator=g) * 1.5 + 0.25).to(DEV))
                    x.weight.copy_((torch.rand(C, generator=g) + 0.5).to(DEV))
                    x.bias.copy_((torch.randn(C, generator=g) * 0.1).to(DEV))
            m.eval()
            torch.manual_seed(3)
            outs, counts = [], None
            with torch.no_grad():
                for it in range(6):
                    if profile and it == 0:
                        PROFILE.reset(True)
                    X, W = m(pcs)
                    if profile and it == 0:
                        counts = {k: v["launches"] for k, v in PROFILE.summary().items()}
                        PROFILE.reset(False)
                    outs.append((X.clone(), W.clone()))
                    if it == 1:                                   # in place, all of them
                        for x in bn_mods(m):
                            x.running_var.mul_(1.7)
                            x.running_mean.add_(0.05)
                            x.weight.mul_(0.9)
                            x.bias.sub_(0.02)
                    elif it == 2:                                 # load_state_dict copies in place
                        sd = {k: (v * 1.1 if v.dtype.is_floating_point and "running_var" in k else v.clone()) for k, v in m.state_dict().items()}
                        m.load_state_dict(sd)
                    elif it == 3:                                 # NEW storage for one layer's statistics and another's parameters
                        a, b = bn_mods(m)[2], bn_mods(m)[9]
                        a.running_mean = (a.running_mean * 0.5).clone()
                        b.weight = torch.nn.Parameter((b.weight * 1.25).clone())
            torch.cuda.synchronize()
            return m, outs, counts
        finally:
            ops.USE_BN_EVAL_BATCH, autograph.ENABLED = old

    _, ref, c_ref = run(False, False, profile=True)
    _, bat, c_bat = run(True, False, profile=True)
    assert c_ref.get("p2c_bn_finalize_f32", 0) == 17 and "p2c_bn_eval_affine_batch_f32" not in c_ref, c_ref
    assert c_bat.get("p2c_bn_eval_affine_batch_f32", 0) == 1 and "p2c_bn_finalize_f32" not in c_bat, c_bat
    mg, gra, _ = run(True, True)
    assert not autograph._state(mg)["failed"]
    for it, ((X0, W0), (X1, W1), (X2, W2)) in enumerate(zip(ref, bat, gra)):
        assert torch.isfinite(X0).all() and torch.isfinite(W0).all()
        assert torch.equal(X0, X1) and torch.equal(W0, W1), ("eager", it)
        assert torch.equal(X0, X2) and torch.equal(W0, W2), ("graphed", it)
    for it in (1, 2, 3):                                      # (the updates did change the outputs: the comparison above is not vacuous)
        assert not torch.equal(ref[it][0], ref[it + 1][0]), it
    # gradients wanted, or train mode: the per-layer form (its affine is saved for the backward), untouched
    m = _fresh_backbone(4).eval()
    old = autograph.ENABLED
    autograph.ENABLED = False
    try:
        PROFILE.reset(True)
        X, W = m(pcs)
        c = {k: v["launches"] for k, v in PROFILE.summary().items()}
    finally:
        PROFILE.reset(False)
        autograph.ENABLED = old
    assert "p2c_bn_eval_affine_batch_f32" not in c and ops._EVAL_AFF[0] is None, c


def test_autograph_is_an_autograd_citizen_and_follows_moved_parameters():
    """ADVICE r4 (autograph.py:169, ops.py:862).  (a) the parameter gradients of the graphed module are real autograd outputs:
    torch.autograd.grad(loss, params) returns them (and leaves .grad alone), a parameter hook sees them, and after loss.backward() every
    .grad is a tensor of its own (not the graph's static buffer: the next backward does not overwrite it); (b) a backward through a
    forward that is no longer the latest replay of its graph raises instead of differentiating the wrong activations; (c) parameters
    whose STORAGE moved (`p.data = ...`, model.cpu(); model.cuda()) or that were replaced (load_state_dict(assign=True)) are followed:
    the staged weight operands (ops.WeightStage: sa1/sa2/sa3 first layers, fp3, the heads) and the cached graphs are rebuilt, so the
    forward equals the eager forward of the same module on the same draws."""
    from point2cyl_amd import autograph
    B, N = 2, 1024
    pcs = synth.make_batch(B, N, 8, seed=78)[0].float().to(DEV)
    m = _fresh_backbone(3)
    params = [p for p in m.parameters()]
    names = [k for k, _ in m.named_parameters()]
    # (a) autograd.grad == backward, hooks fire, .grad survives the next backward
    torch.manual_seed(1)
    X, W = m(pcs)
    loss = X.square().mean() + W.square().mean()
    seen = {}
    h = m.fc1.weight.register_hook(lambda g: seen.setdefault("fc1", g.detach().clone()))
    got = torch.autograd.grad(loss, params, retain_graph=False, allow_unused=True)
    h.remove()
    assert all(p.grad is None for p in params), "autograd.grad must not touch .grad"
    assert all(g is not None and torch.isfinite(g).all() for g in got)
    assert "fc1" in seen and torch.equal(seen["fc1"], got[names.index("fc1.weight")])
    st = autograph._state(m)
    assert len(st["graphs"]) == 1 and not st["failed"]
    m._drop_seed.sub_(0x9E3779B97F4A7C15 % (2 ** 62))          # same dropout bits and (seeded) FPS draws again
    torch.manual_seed(1)
    X, W = m(pcs)
    (X.square().mean() + W.square().mean()).backward()
    gmax = max(float(g.norm()) for g in got)
    for k, p, g in zip(names, params, got):        # (fp atomics: order noise only; a BatchNorm shift in front of another BatchNorm has a zero gradient)
        assert float((p.grad - g).norm()) <= 1e-4 * float(g.norm()) + 1e-6 * gmax, k
    kept = {k: p.grad for k, p in zip(names, params)}
    snap = {k: g.clone() for k, g in kept.items()}
    for p in params:
        p.grad = None
    X, W = m(pcs)
    (3.0 * X.square().mean()).backward()
    for k in snap:
        assert torch.equal(kept[k], snap[k]), "a gradient handed out earlier was overwritten by the next backward: %s" % k
    # (b) stale backward raises
    for p in params:
        p.grad = None
    X1, W1 = m(pcs)
    l1 = X1.square().mean()
    X2, W2 = m(pcs)
    with pytest.raises(RuntimeError, match="not the latest"):
        l1.backward()
    (X2.square().mean()).backward()                              # the latest one still works
    assert all(torch.isfinite(p.grad).all() for p in params if p.grad is not None)

    # (c) moved / replaced parameters
    def eager_and_graphed(seed):
        for flag in (False, True):
            autograph.ENABLED = flag
            try:
                m._drop_seed.fill_(12345)
                torch.manual_seed(seed)
                with torch.no_grad():
                    yield torch.cat(m(pcs), -1).clone()
            finally:
                autograph.ENABLED = True

    e0, g0 = eager_and_graphed(4)
    assert float((e0 - g0).abs().max()) <= 2e-5 * float(e0.abs().max())
    with torch.no_grad():
        for mod in (m.sa1.mlp_convs[0], m.sa2.mlp_convs[0], m.sa3.mlp_convs[0], m.fp3.mlp_convs[0], m.fc2[1], m.fp1.mlp_convs[1]):
            mod.weight.data = (mod.weight.data * 1.5 + 0.01).clone()          # NEW storage, new values
    e1, g1 = eager_and_graphed(4)
    assert float((e1 - e0).abs().max()) > 1e-3 * float(e0.abs().max()), "the perturbation must change the output"
    assert float((e1 - g1).abs().max()) <= 2e-5 * float(e1.abs().max()), "graph / staged operands kept reading the old weights"
    m.cpu(); m.to(DEV)
    with torch.no_grad():
        m.fc2[0].weight.mul_(0.5)
    e2, g2 = eager_and_graphed(4)
    assert float((e2 - g2).abs().max()) <= 2e-5 * float(e2.abs().max())
    sd = {k: (v * 0.9 if v.dtype.is_floating_point and "running_var" not in k else v).clone() for k, v in m.state_dict().items()}
    m.load_state_dict(sd, assign=True)
    e3, g3 = eager_and_graphed(4)
    assert float((e3 - e2).abs().max()) > 1e-4 * float(e2.abs().max())
    assert float((e3 - g3).abs().max()) <= 2e-5 * float(e3.abs().max())
    autograph.reset()


def test_dropin_trainer_step_equals_native_step_and_trains():
    """point2cyl_amd/dropin/trainer_step.py (train_Point2Cyl_without_sketch.py:244-369 composed on the drop-in import names, torch.optim.Adam,
    six .item()) against point2cyl_amd.step.train_step on the same seeds: the six logged scalars of step 0 at 1e-5, and the loss goes down
    over 25 steps of the drop-in loop (graphs inside backbone.forward: one forward + one backward capture for the whole run)."""
    from point2cyl_amd import autograph
    from point2cyl_amd.dropin.trainer_step import TrainerStep
    B, N, K = 4, 2048, 8
    pcs, nrm, seg, bb, _, _, axes, _, cen = synth.make_batch(B, N, K, seed=3)
    batch = tuple(x.to(DEV) for x in (pcs.float(), nrm.float(), seg, bb, axes.float(), cen.float()))
    for full in (False, True):
        torch.manual_seed(0)
        st = TrainerStep(K=K, batch_size=B, pred_extrusion=full, pred_center=full, device=DEV)
        torch.manual_seed(0)
        model = backbone(output_sizes=[3, 2 * K]).to(DEV).train()
        opt = torch.optim.Adam(model.parameters(), lr=1e-3)
        fl = step.StepFlags(K=K, pred_extrusion=full, pred_center=full)
        torch.manual_seed(42)
        logs = st(*batch)
        autograph.ENABLED = False
        try:
            torch.manual_seed(42)
            out = step.train_step(model, opt, batch, fl)
        finally:
            autograph.ENABLED = True
        ref = [float(out[k]) for k in ("total", "normal", "miou", "bb", "ext", "center")]
        np.testing.assert_allclose(np.array(logs), np.array(ref), rtol=2e-5, atol=1e-7)
        for (k, a), b_ in zip(st.model.state_dict().items(), model.state_dict().values()):
            if a.dtype.is_floating_point and "running" not in k:
                assert float((a - b_).abs().max()) <= 2.5e-3, k          # one Adam step of lr 1e-3: |delta| <= lr, sign flips only where grad ~ 0
        first = logs[0]
        for _ in range(24):
            logs = st(*batch)
        assert np.isfinite(logs).all() and logs[0] < first, (first, logs)
        # two captures for the whole run: the first forward runs with the constructor's BatchNorm momentum 0.1, the trainer sets 0.5 after
        # it (train...:207, :357-360) and the momentum is a kernel argument
        assert len(autograph._state(st.model)["graphs"]) == 2
        autograph.reset(st.model)


@pytest.mark.parametrize("mode", ["eval", "train"])
def test_reference_made_checkpoint_loads_and_evaluates(mode, tmp_path):
    """SURVEY 8(f) rank 4 / eval.py:206-210: `tests/golden/ref_ckpt_3steps.pth` was written by the REFERENCE module (oracle/make_golden_ckpt.py:
    pe.backbone trained for three Adam steps, saved as {"model": state_dict} like train...:405-410).  Loaded here the way eval.py loads it,
    the backbone must reproduce the reference module's forward with that checkpoint (fixture G16: FPS starts and dropout mask recorded) -
    eval mode (running statistics from the file) and train mode; then the evaluation CLI runs on it end to end (--ckpt)."""
    g = load_golden("g16_ref_ckpt_forward")
    ck = torch.load(os.path.join(ROOT, "tests", "golden", "ref_ckpt_3steps.pth"), map_location="cpu")
    assert list(ck) == ["model"] and list(ck["model"].keys()) == [str(k) for k in g["keys"]] and len(ck["model"]) == 123
    m = backbone(output_sizes=[3, 16]).to(DEV)
    missing = m.load_state_dict(ck["model"])                                     # strict: same 123 keys and shapes
    assert not missing.missing_keys and not missing.unexpected_keys
    assert [int(v) for k, v in m.state_dict().items() if k.endswith("num_batches_tracked")] == [int(v) for v in g["nbt"]] and int(g["nbt"][0]) == 3
    m.eval() if mode == "eval" else m.train()
    B, N = g["pcs"].shape[:2]
    dm = np.unpackbits(g["dropout_mask_bcn"])[:B * 128 * N].reshape(B, 128, N)
    m.dropout_mask = torch.from_numpy(dm).permute(0, 2, 1).contiguous()
    m.sa1.fps_start, m.sa2.fps_start = t(g[mode + ":start1"]), t(g[mode + ":start2"])
    with torch.no_grad():
        X, W_raw = m(cu(g["pcs"]))
    # eval mode (statistics from the file): 1e-4.  Train mode normalises 17 layers with the statistics of TWO clouds - the chain the G5
    # fixture shows the reference's own fp32 run to be 1.0e-4 .. 1.6e-4 away from its float64 twin on: 3e-4 (measured here: 1.3e-4 on 4 of
    # 6144 values), labels equal wherever the reference's two largest logits are more than 1e-3 apart
    tol = 1e-4 if mode == "eval" else 3e-4
    np.testing.assert_allclose(X.cpu().numpy(), g[mode + ":X"], rtol=tol, atol=tol)
    np.testing.assert_allclose(W_raw.cpu().numpy(), g[mode + ":W_raw"], rtol=tol, atol=tol)
    ref_w = t(g[mode + ":W_raw"])
    top2 = ref_w.topk(2, dim=-1).values
    clear = (top2[..., 0] - top2[..., 1]) > 1e-3
    assert torch.equal(W_raw.argmax(-1).cpu()[clear], ref_w.argmax(-1)[clear]) and float(clear.float().mean()) > 0.99
    if mode == "train":
        return
    # the evaluation script on the upstream-made file
    import shutil
    shutil.copy(os.path.join(ROOT, "tests", "golden", "ref_ckpt_3steps.pth"), str(tmp_path / "model.pth"))
    # 18 clouds in batches of 4: four full batches through graph.PipelinedForward (one batch ahead, or in groups of two whose geometry is
    # computed together), the last, shorter one through the serial forward; --no_prefetch evaluates the same clouds serially: same report
    reps = []
    for extra in (["--prefetch_group", "1"], ["--no_prefetch"], ["--prefetch_group", "2"]):
        rep = str(tmp_path / ("rep%d.json" % len(reps)))
        out = subprocess.run([sys.executable, "-m", "point2cyl_amd.eval", "--logdir", str(tmp_path), "--ckpt", "model.pth", "--synthetic", "18",
                              "--batch_size", "4", "--num_point", "1024", "--dump_dir", str(tmp_path / "dump"), "--report", rep] + extra, cwd=ROOT,
                             capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        assert "Num evaluated= 18" in out.stdout and "Mean mIOU= " in out.stdout, out.stdout[-1500:]
        reps.append((json.load(open(rep)), [float(l.split("=")[-1]) for l in out.stdout.splitlines() if l.startswith("Mean ")]))
    assert reps[0][0]["batches"] == 5 and reps[0][0]["batches_pipelined"] == 4 and reps[1][0]["batches_pipelined"] == 0
    assert reps[2][0]["batches"] == 5 and reps[2][0]["batches_pipelined"] == 4 and reps[2][0]["prefetch_group"] == 2      # two groups of two, the short batch serial
    np.testing.assert_allclose(reps[2][1], reps[1][1], rtol=0.15, atol=1e-3)
    # The runs are NOT draw-for-draw identical: the pipelined loop draws batch i + 1's FPS starts before batch i's extent samples
    # (data_utils.py:1696 draws those on the same CPU generator), the serial loop after them - another random sampling of the same clouds, as
    # two runs of the reference with different seeds are.  (Bit-equality of the two forwards on the SAME draws is
    # test_pipelined_forward_equals_the_serial_forward's job.)  The report lines agree to the sampling noise of 18 clouds:
    np.testing.assert_allclose(reps[0][1], reps[1][1], rtol=0.15, atol=1e-3)


def test_with_sketch_trainer_two_ranks_on_one_gpu_over_gloo(tmp_path):
    """configs[4] as a data-parallel job: point2cyl_amd.train_sketch with two ranks that share this GPU over gloo, backbone
    + sketch encoder trained, 8 synthetic shapes sharded 4 + 4, B = 2 per rank, 4 steps.  The report's multi_gpu object: identical
    parameters of the trained modules on both ranks after the last step, identical BatchNorm statistics after ddp.average_buffers (what
    rank 0's checkpoint holds), ONE exchange of all trainable parameters per step (backbone 1,404,243 + encoder), the schedules of the
    global batch (world * B samples per step)."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(P2C_ONE_GPU_RANKS="1", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="8")
    rep = str(tmp_path / "report.json")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           "-m", "point2cyl_amd.train_sketch", "--pred_seg", "--pred_normal", "--pred_bb", "--is_pc_train", "--is_im_train", "--with_im_loss",
           "--synthetic", "8", "--batch_size", "2", "--num_point", "1024", "--num_sk_point", "256", "--num_epochs", "2", "--max_steps", "4",
           "--decay_step", "8", "--bn_decay_step", "8", "--logdir", str(tmp_path / "run"), "--im_logdir", str(tmp_path / "none"), "--report", rep]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    r = json.load(open(rep))
    mg = r["multi_gpu"]
    assert r["world"] == 2 and r["steps"] == 4 and mg["backend"] == "gloo" and mg["samples_per_step"] == 4
    assert mg["params_identical"] and mg["param_checksum"][0] == mg["param_checksum"][1] > 0
    assert mg["buffers_identical"]
    assert mg["trainable_parameters"] > 1404243 and mg["allreduce_bytes"] == 4 * mg["trainable_parameters"]
    assert mg["learning_rate"] == [pytest.approx(1e-3 * 0.7)] * 2, mg["learning_rate"]
    assert mg["next_bn_momentum"] == [pytest.approx(0.25)] * 2, mg["next_bn_momentum"]
    ck = torch.load(str(tmp_path / "run" / "model.pth"), map_location="cpu")
    assert set(ck.keys()) == {"model", "implicit_net", "pn_encoder"} and len(ck["model"]) == 123
    assert all(torch.isfinite(v).all() for v in ck["model"].values() if v.dtype.is_floating_point)


def test_trainer_cli_two_ranks_on_one_gpu_over_gloo(tmp_path):
    """point2cyl_amd.train as a data-parallel job: two ranks that share this GPU (P2C_ONE_GPU_RANKS: gloo instead of RCCL; the 8-GPU run uses
    the same code with backend nccl), 16 synthetic shapes sharded 8 + 8, B = 2 per rank, 4 steps through the HIP-graph path.
    After the last step: the replicas hold identical parameters (sum of squares in float64), both followed the schedules of the GLOBAL batch
    (world * B samples per step: with --decay_step 8 / --bn_decay_step 8 the staircases move after every second step - a rank counting its
    own B = 2 would be one stair behind), and the checkpoint rank 0 wrote carries the MEAN of the replicas' BatchNorm statistics."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(P2C_ONE_GPU_RANKS="1", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="8")
    rep = str(tmp_path / "report.json")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           "-m", "point2cyl_amd.train", "--pred_seg", "--pred_normal", "--pred_bb", "--synthetic", "16", "--batch_size", "2", "--num_point", "1024",
           "--num_epochs", "1", "--max_steps", "4", "--decay_step", "8", "--bn_decay_step", "8", "--logdir", str(tmp_path / "run"), "--report", rep, "--quiet"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    r = json.load(open(rep))
    mg = r["multi_gpu"]
    assert r["world"] == 2 and r["steps"] == 4 and r["graph"] and mg["backend"] == "gloo" and mg["samples_per_step"] == 4
    assert mg["params_identical"] and mg["param_checksum"][0] == mg["param_checksum"][1] > 0
    assert mg["buffers_identical"], "the BatchNorm statistics in the checkpoint are the mean over the replicas (ddp.average_buffers)"
    # schedules of the global batch: after 4 steps of 4 samples gstep * B * world / 8 = floor(3 * 4 / 8) = 1 stair for the last lr set
    # (get_learning_rate(gstep = 3)) and floor(3 * 4 / 8) = 1 for the momentum that reaches the next forward
    assert mg["learning_rate"] == [pytest.approx(1e-3 * 0.7)] * 2, mg["learning_rate"]
    assert mg["next_bn_momentum"] == [pytest.approx(0.25)] * 2, mg["next_bn_momentum"]
    assert mg["allreduce_bytes"] == 5616972 == 4 * 1404243
    ck = torch.load(str(tmp_path / "run" / "model.pth"), map_location="cpu")["model"]
    assert len(ck) == 123 and int(ck["bn1.num_batches_tracked"]) == 4
    assert all(torch.isfinite(v).all() for v in ck.values() if v.dtype.is_floating_point)


@pytest.mark.gpu
def test_trainer_and_eval_cli_on_a_dataset_file(tmp_path):
    """The CLIs on a dataset FILE (what a user of the reference has: <data_dir>/<split>.h5, here the same arrays as .npz) instead of
    --synthetic: clouds of 1536 points subsampled to --num_point 1024 by a fresh permutation per item (dataloader.py:69-85).  The trainer
    runs three epochs on it; the evaluation loop - loader thread, per-item permutations from a private generator seeded from torch's -
    is REPRODUCIBLE (two pipelined runs with the same seed print the same report, digit for digit, whatever the two threads' timing),
    --no_prefetch (reference draw order, nothing read ahead) evaluates the same clouds: every cloud counted, metrics close."""
    n, P, K = 18, 1536, 8
    pcs, nrm, seg, bb, _, _, axes, dist_, cen = synth.make_batch(n, P, K, seed=77)
    data = tmp_path / "data"
    data.mkdir()
    arrays = dict(point_cloud=pcs.numpy().astype(np.float32), normals=nrm.numpy().astype(np.float32), extrusion_labels=seg.numpy(),
                  base_barrel_labels=bb.numpy(), n_instances=(seg.max(dim=1)[0] + 1).numpy(), extrusion_axes=axes.numpy().astype(np.float32),
                  extrusion_distances=dist_.numpy().astype(np.float32), extrusion_centers=cen.numpy().astype(np.float32))
    for split in ("train", "test"):
        np.savez(str(data / (split + ".npz")), **arrays)
    logdir = str(tmp_path / "run")
    out = _run(["-m", "point2cyl_amd.train", "--pred_seg", "--pred_normal", "--pred_bb", "--data_dir", str(data), "--batch_size", "4",
                "--num_point", "1024", "--num_epochs", "3", "--logdir", logdir, "--quiet"])
    assert out.returncode == 0, out.stderr[-3000:]
    assert os.path.exists(os.path.join(logdir, "model.pth"))
    reports = []
    for extra in ([], [], ["--no_prefetch"], ["--prefetch_group", "2", "--add_noise"]):
        o = _run(["-m", "point2cyl_amd.eval", "--logdir", logdir, "--ckpt", "model.pth", "--data_dir", str(data), "--data_split", "test",
                  "--batch_size", "4", "--num_point", "1024", "--dump_dir", str(tmp_path / "dump")] + extra)
        assert o.returncode == 0, o.stderr[-3000:]
        assert "Num evaluated= %d" % n in o.stdout, o.stdout[-1500:]
        reports.append([l for l in o.stdout.splitlines() if l.startswith("Mean ")])
        assert len(reports[-1]) == 5
    assert reports[0] == reports[1], (reports[0], reports[1])              # same seed, same report: the loader thread's draws do not race the loop's
    val = lambda rep: np.array([float(l.split("=")[1]) for l in rep])
    a, b = val(reports[0]), val(reports[2])
    assert np.isfinite(a).all() and np.isfinite(b).all() and np.isfinite(val(reports[3])).all()
    np.testing.assert_allclose(a[[0, 2]], b[[0, 2]], atol=0.08)            # mIoU, base / barrel accuracy: other subsamples of the same clouds
    # --with_sketch_fit (eval.py:459-590; no pre-trained decoder on disk: randomly initialised, the path and the two extra report lines are the test)
    o = _run(["-m", "point2cyl_amd.eval", "--logdir", logdir, "--ckpt", "model.pth", "--data_dir", str(data), "--data_split", "test", "--batch_size", "4",
              "--num_point", "1024", "--num_sk_point", "256", "--dump_dir", str(tmp_path / "dump"), "--with_sketch_fit", "--im_logdir", str(tmp_path / "none")])
    assert o.returncode == 0, o.stderr[-3000:]
    extra = [l for l in o.stdout.splitlines() if "fitting loss=" in l]
    assert len(extra) == 2 and all(np.isfinite(float(l.split("=")[1])) for l in extra), o.stdout[-1500:]


@pytest.mark.gpu
def test_with_sketch_trainer_cli_on_a_dataset_file(tmp_path):
    """train_Point2Cyl.py's counterpart on a dataset FILE with ground-truth sketches (schema of utils.py:1251-1268: `sketches` (n,K,S_all,4) =
    [2-D point | 2-D normal], `sketches_norms`): clouds of 1536 points and sketches of 384 subsampled per step to --num_point 1024 /
    --num_sk_point 256 (dataloader.py:69-85, :211-214)."""
    n, P, K, SA = 6, 1536, 8, 384
    pcs, nrm, seg, bb, _, _, axes, dist_, cen = synth.make_batch(n, P, K, seed=91)
    g = np.random.default_rng(5)
    ang = g.uniform(0, 2 * np.pi, (n, K, SA))
    rad = g.uniform(0.2, 0.9, (n, K, 1))
    circ = np.stack([np.cos(ang), np.sin(ang)], -1)
    sketches = np.concatenate([rad[..., None] * circ, circ], -1).astype(np.float32)            # circles: points and outward normals
    data = tmp_path / "data"
    data.mkdir()
    np.savez(str(data / "train.npz"), point_cloud=pcs.numpy().astype(np.float32), normals=nrm.numpy().astype(np.float32),
             extrusion_labels=seg.numpy(), base_barrel_labels=bb.numpy(), n_instances=(seg.max(dim=1)[0] + 1).numpy(),
             extrusion_axes=axes.numpy().astype(np.float32), extrusion_distances=dist_.numpy().astype(np.float32),
             extrusion_centers=cen.numpy().astype(np.float32), sketches=sketches, sketches_norms=rad[..., 0].astype(np.float32))
    logdir = str(tmp_path / "sk")
    out = _run(["-m", "point2cyl_amd.train_sketch", "--pred_seg", "--pred_normal", "--pred_bb", "--pred_extrusion", "--is_pc_train", "--is_im_train",
                "--with_im_loss", "--data_dir", str(data), "--batch_size", "2", "--num_point", "1024", "--num_sk_point", "256", "--num_epochs", "1",
                "--save_every", "1", "--logdir", logdir, "--im_logdir", str(tmp_path / "none")])
    assert out.returncode == 0, out.stderr[-3000:]
    im = [l for l in out.stdout.splitlines() if "latent loss" in l]
    assert len(im) == 3, out.stdout[-2000:]
    vals = np.array([float(x.split(":")[1]) for l in im for x in l.split("|")[2:]])
    assert np.isfinite(vals).all()
    ck = torch.load(os.path.join(logdir, "model.pth"), map_location="cpu")
    assert set(ck.keys()) == {"model", "implicit_net", "pn_encoder"}


@pytest.mark.gpu
def test_eval_cli_two_ranks_on_one_gpu_over_gloo(tmp_path):
    """point2cyl_amd.eval as a job of two ranks (clouds are independent: ddp.shard_range, no data-path collective; ONE exchange at the end,
    the metric sums): 18 clouds sharded 9 + 9, rank 0 prints the report over all 18; per-rank logs exist; the numbers are those of a
    one-process run up to the random draws (FPS starts, extent samples) each rank makes for its own shard."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(P2C_ONE_GPU_RANKS="1", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    shutil = __import__("shutil")
    shutil.copy(os.path.join(ROOT, "tests", "golden", "ref_ckpt_3steps.pth"), str(tmp_path / "model.pth"))
    common = ["-m", "point2cyl_amd.eval", "--logdir", str(tmp_path), "--ckpt", "model.pth", "--synthetic", "18", "--batch_size", "4", "--num_point", "1024"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port)] + \
        common + ["--dump_dir", str(tmp_path / "d2")]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    assert "Num evaluated= 18" in out.stdout, out.stdout[-1500:]
    assert os.path.exists(str(tmp_path / "d2" / "log_evaluate.0.txt")) and os.path.exists(str(tmp_path / "d2" / "log_evaluate.1.txt"))
    one = _run(common + ["--dump_dir", str(tmp_path / "d1")])
    assert one.returncode == 0 and "Num evaluated= 18" in one.stdout
    val = lambda txt: np.array([float(l.split("=")[1]) for l in txt.splitlines() if l.startswith("Mean ")])
    a, b = val(out.stdout), val(one.stdout)
    assert a.shape == b.shape == (5,) and np.isfinite(a).all()
    np.testing.assert_allclose(a[[0, 2]], b[[0, 2]], atol=0.08)


@pytest.mark.gpu
def test_graphed_metrics_equal_the_eager_metrics():
    """eval.GraphedMetrics (the evaluation loop's default for pipelined batches: every metric kernel of a batch as one HIP-graph replay,
    the extent draws made before the replay into a fixed buffer) returns the accumulator block of the eager eval_metrics call, bit for
    bit, for the batch it was captured on and for later batches of other clouds - same generator state, same draws."""
    from point2cyl_amd import fitting
    B, N, K = 4, 2048, 8
    fl = p2c_eval.EvalFlags(K=K, num_sk_point=512)
    torch.manual_seed(3)
    model = backbone(output_sizes=fl.pred_sizes()).to(DEV).eval()
    acc = p2c_eval.Accumulator()

    def make(seed):
        pcs, nrm, inst, bb, _, _, axes, _, cen = synth.make_batch(B, N, K, seed=seed)
        extras = dict(barrel_counts=fitting.barrel_counts(inst.long(), bb.long(), K), labels_validated=True)
        batch = (pcs.float().to(DEV), nrm.float().to(DEV), inst.to(DEV), bb.float().to(DEV), axes.float().to(DEV), cen.float().to(DEV), extras)
        with torch.no_grad():
            heads, sizes = model.forward_heads(batch[0], model.compute_geometry(batch[0], with_csr=False))
        return batch, (heads.clone(), sizes)

    def eager(batch, heads):
        h, sizes = heads
        hv = h.view(B, N, h.shape[-1])
        with torch.no_grad():
            m = p2c_eval.eval_metrics(hv[:, :, 0:sizes[0]], hv[:, :, sizes[0]:sizes[0] + sizes[1]], *batch[:6], fl, **batch[6])
        return acc.block(m).clone()

    b1, h1 = make(11)
    b2, h2 = make(12)
    torch.manual_seed(100); e1 = eager(b1, h1)
    torch.manual_seed(101); e2 = eager(b2, h2)
    torch.manual_seed(100)
    gm = p2c_eval.GraphedMetrics(fl, acc.keys, b1, h1)
    g1 = gm().clone()
    torch.manual_seed(101)
    g2 = gm(b2, h2).clone()
    torch.manual_seed(100)
    g1b = gm(b1, h1).clone()
    assert torch.equal(g1, e1) and torch.equal(g2, e2) and torch.equal(g1b, e1)
    assert not torch.equal(e1, e2) and torch.isfinite(e1).all()


@pytest.mark.gpu
@pytest.mark.parametrize("K,norm_eig", [(8, False), (8, True), (4, False), (2, False)])
def test_fused_metrics_equal_the_eager_metric_chain(K, norm_eig):
    """csrc/metrics.hip (ops.eval_metrics_fused / eval.FusedMetrics: eval.py:270-446 of a batch as two launches) against the torch-op
    mirror eval.eval_metrics on the same head outputs.  The hard labels are formed with torch's own softmax arithmetic, so everything
    that is a ratio of COUNTS - matching, mask, mIoU, base/barrel accuracy, found masks - must be EQUAL (float32-rounded values
    digit for digit); the sums of floats over the points (angle mean, scatter matrices, centroids) run in another order and are held to
    rounding: normal angle 2e-6 relative, fitted axes 1e-5 rad (the angle metric 2e-4 deg), centroids 1e-6.  Untrained heads put most
    points into a few columns: null columns, unmatched segments and not-found centroids all occur (asserted)."""
    B, N = 6, 3000
    fl = p2c_eval.EvalFlags(K=K, num_sk_point=256, norm_eig=norm_eig)
    torch.manual_seed(4)
    model = backbone(output_sizes=fl.pred_sizes()).to(DEV).eval()
    seen_null = seen_nf = False
    for seed, gain in ((21, 1.0), (22, 3.0), (23, 0.2), (24, None)):
        pcs, nrm, inst, bb, _, _, axes, _, cen = synth.make_batch(B, N, K, seed=seed)
        if gain is None:
            inst = (torch.arange(N) % K).repeat(B, 1)          # K instances per cloud: every column gets matched
        if seed == 23:
            inst = inst.clone()
            inst[0, ::7] = -1                       # background points
            inst[1] = torch.where(inst[1] > 0, torch.zeros_like(inst[1]), inst[1])      # one instance only
        batch = (pcs.float().to(DEV), nrm.float().to(DEV), inst.to(DEV), bb.float().to(DEV), axes.float().to(DEV), cen.float().to(DEV))
        with torch.no_grad():
            heads, sizes = model.forward_heads(batch[0], model.compute_geometry(batch[0], with_csr=False))
            heads = heads.clone()
            if gain is not None:
                heads[:, sizes[0]:sizes[0] + sizes[1]] *= gain * 8.0          # sharper / flatter memberships
            else:
                # crafted memberships: predicted label = gt label, except that column K-3 keeps 10 points only (soft mass below 0.005 N: a NULL
                # column, emptied by the encoding) and column K-2 keeps ONE hard point but plenty of soft mass (not null, centroid "not found")
                n = torch.arange(N)
                lab = inst[0].clone()
                lab[(lab == K - 3) & (n >= 10 * K)] = 0
                lone = (inst[0] == K - 2) & (n >= K)
                lab[lone] = min(1, K - 1)
                lg = torch.zeros(N, 2 * K)
                lg[n, 2 * lab + bb[0].long()] = 25.0
                if K >= 4:
                    lg[lone] = -20.0
                    lg[lone, 2 * 1] = 2.0
                    lg[lone, 2 * (K - 2)] = 1.5
                heads[:, sizes[0]:sizes[0] + sizes[1]] = lg.repeat(B, 1).to(DEV)
            hv = heads.view(B, N, heads.shape[-1])
            m = p2c_eval.eval_metrics(hv[:, :, 0:sizes[0]], hv[:, :, sizes[0]:sizes[0] + sizes[1]], *batch, fl,
                                      extent_rand_idx=torch.zeros(B, K, fl.num_sk_point, dtype=torch.int64, device=DEV), labels_validated=True)
            out, det = ops.eval_metrics_fused(heads, 0, sizes[0], *batch, K, normalize=norm_eig, details=True)
        torch.cuda.synchronize()
        mask = m["mask"].bool()
        assert torch.equal(det["mask"].bool(), mask)
        assert torch.equal(det["matching_indices"], m["matching_indices"]), (det["matching_indices"], m["matching_indices"])
        assert torch.equal(out[0].float(), m["mIoU"].float()) or float((out[0].float() - m["mIoU"]).abs().max()) <= 1.2e-7, (out[0], m["mIoU"])
        # (hits / N: the torch chain multiplies by the rounded reciprocal of N on the device, the reference's CPU run and the kernel divide - one
        # ulp apart unless N is a power of two; the COUNT is what must be equal)
        assert torch.equal(torch.round(out[2] * N), torch.round(m["pred_bb_acc"].double() * N))
        assert float((out[2].float() - m["pred_bb_acc"]).abs().max()) <= 1.2e-7
        np.testing.assert_allclose(out[1].cpu().numpy(), m["normal_difference"].double().cpu().numpy(), rtol=2e-6)
        # fitted axes of the matched segments (sign-canonical in both), centroids, found masks
        # (the mirror's float64 axes are not in its dict: compare with the float32 ones)
        sin = torch.linalg.norm(torch.linalg.cross(det["E64"], m["E_AX"].double()), dim=-1)      # |sin(angle)|: sign-free, resolves small angles
        # fp32 storage of the mirror's unit vector: 1e-7; the two chains add the same fp32 products in different orders (1e-7 relative on the
        # scatter sums), and an untrained network's near-isotropic scatter matrices amplify that by 1 / eigen-gap: 2e-5 rad measured at worst.
        # (The parity pin of the axes is test_eval_flow_golden_fused_metrics: 3e-7 against the reference's float64 run.)
        assert float(sin[mask].max()) < 1e-4, sin
        np.testing.assert_allclose(out[3].cpu().numpy(), m["extrusion_difference"].double().cpu().numpy(), rtol=0, atol=5e-3)
        assert torch.equal(det["found_centers_mask"][mask], m["found_centers_mask"][mask])
        np.testing.assert_allclose(det["predicted_centroids"][mask].cpu().numpy(), m["predicted_centroids"][mask].cpu().numpy(), rtol=0, atol=1e-6)
        np.testing.assert_allclose(out[4].cpu().numpy(), m["centroid_difference"].double().cpu().numpy(), rtol=2e-6, atol=1e-7)
        W = m["W"]
        seen_null |= bool((W.sum(1) < N * 0.005).any())
        seen_nf |= bool((m["found_centers_mask"][mask] == 0).any())
        # the loop's wrapper: same block, plus the extents of eval.py:456
        fm = p2c_eval.FusedMetrics(fl, [k for k, _ in p2c_eval.REPORT])
        blk = fm(batch + (dict(extent_rand_idx=torch.zeros(B, K, fl.num_sk_point, dtype=torch.int64, device=DEV)),), (heads, sizes))
        assert torch.equal(blk, out) and fm.extents.shape == (K, B, 2)
    assert K != 8 or (seen_null and seen_nf)
