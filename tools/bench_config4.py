"""BASELINE configs[3] ("configs[4]" in SURVEY numbering): the eval.py fitting-only path on 10 000 pre-segmented synthetic
cylinders = 1250 clouds x K=8 segments x N=8192 points, 1 GPU.

    python tools/bench_config4.py [--clouds 1250] [--cpu_clouds 64] [--steps 20]

Inputs: X = ground-truth normals + Gaussian angular noise (sigma = 2 deg), W_barrel / W_base one-hot from
the labels.  Timed on the GPU, inputs resident: estimate_extrusion_axis (data_utils.py:99-177), per-segment hard centroids
(eval.py:409-436), get_extrusion_extents with S = 2048 samples (data_utils.py:1650-1730; the random draws are made up
front, they are host work in the reference too).  Reported next to it: the oracle's closed-form CPU restatement on a
subset, the HBM roofline (76 B/point algorithmic for the axis fit), and the parity of the eval metric
(axis-angle error in degrees, eval.py:398-405) between the two paths on the same inputs.  One JSON line on stdout."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def make_inputs(n_clouds, N, K, seed, distinct=64):
    from point2cyl_amd import synth
    pcs, nrm, seg, bb, _, _, axes, _, cen = synth.make_batch(min(distinct, n_clouds), N, K, seed=seed)
    pcs, nrm, axes = pcs.float(), nrm.float(), axes.float()
    reps = (n_clouds + pcs.shape[0] - 1) // pcs.shape[0]
    tile = lambda t: t.repeat((reps,) + (1,) * (t.dim() - 1))[:n_clouds].contiguous()
    pcs, nrm, seg, bb, axes = tile(pcs), tile(nrm), tile(seg), tile(bb), tile(axes)
    g = torch.Generator().manual_seed(seed + 1)
    # angular noise: rotate every normal by N(0, 2 deg) about a random axis perpendicular to it
    r = torch.randn(nrm.shape, generator=g)
    perp = r - (r * nrm).sum(-1, keepdim=True) * nrm
    perp = perp / perp.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    ang = torch.randn(nrm.shape[:2], generator=g).unsqueeze(-1) * (2.0 * np.pi / 180.0)
    X = torch.cos(ang) * nrm + torch.sin(ang) * perp
    onehot = torch.nn.functional.one_hot(seg.clamp_min(0), K).float() * (seg >= 0).unsqueeze(-1)
    Wb = onehot * (bb == 0).unsqueeze(-1)
    Wc = onehot * (bb == 1).unsqueeze(-1)
    return pcs, X.float(), seg, bb, axes, Wb, Wc, onehot


def angle_error_deg(E_AX, gt_axes, seg, K):
    """eval.py:398-405: masked mean over the segments that exist of acos(|a . a_gt|) in degrees."""
    dot = (E_AX * gt_axes).sum(-1).abs().clamp(max=1 - 1e-6)
    deg = torch.acos(dot) * 180.0 / np.pi
    present = (torch.nn.functional.one_hot(seg.clamp_min(0), K) * (seg >= 0).unsqueeze(-1)).sum(1) > 0
    return float((deg * present).sum() / present.sum())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clouds", type=int, default=1250)
    ap.add_argument("--num_point", type=int, default=8192)
    ap.add_argument("--K", type=int, default=8)
    ap.add_argument("--samples", type=int, default=2048)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--cpu_clouds", type=int, default=64)
    ap.add_argument("--unfused", action="store_true", help="the three separate kernels (axis, centroids, extents) instead of the one-pass kernel")
    a = ap.parse_args()
    from point2cyl_amd import fitting, ops
    from oracle import ref_torch as R
    dev = torch.device("cuda:0")
    N, K, S = a.num_point, a.K, a.samples
    pcs, X, seg, bb, axes, Wb, Wc, onehot = make_inputs(a.clouds, N, K, 4321)
    g = torch.Generator().manual_seed(7)
    rand_idx = torch.randint(0, 1 << 30, (a.clouds, K, S), generator=g)
    counts = ((seg.unsqueeze(-1) == torch.arange(K)) & (bb == 0).unsqueeze(-1)).sum(1)          # barrel points per segment
    rand_idx = rand_idx % counts.clamp_min(1).unsqueeze(-1)
    d = lambda t: t.to(dev)
    pcs_d, X_d, seg_d, bb_d, Wb_d, Wc_d, oh_d, ri_d = d(pcs), d(X), d(seg), d(bb), d(Wb), d(Wc), d(onehot), d(rand_idx)

    fused = (not a.unfused) and ops.fit_fused_supported(N, K, S)

    def fit():
        if fused:
            E_AX, cen, _, ext, _ = fitting.fit_cylinders(X_d, Wb_d, Wc_d, bb_d, seg_d, pcs_d, rand_idx=ri_d, normalize=False)
            return E_AX, cen, ext
        with torch.no_grad():
            E_AX = fitting.estimate_extrusion_axis(X_d, Wb_d, Wc_d, bb_d, seg_d, normalize=False)
            cen, found = ops.segment_centroids(pcs_d, seg_d, K)
            ext, found2 = fitting.get_extrusion_extents(pcs_d, seg_d, bb_d, E_AX, cen, S, rand_idx=ri_d)
        return E_AX, cen, ext

    for _ in range(3):
        out = fit()
    torch.cuda.synchronize()
    ops.PROFILE.reset(enabled=True)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = fit()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    ops.PROFILE.enabled = False
    prof = ops.PROFILE.summary()
    E_AX, cen, ext = out
    points = a.clouds * N
    axis_ms = prof.get("p2c_extrusion_axis_f32", {}).get("ms", 0.0) / a.steps
    axis_bytes = points * (12 + 2 * K * 4)                     # X + W_barrel + W_base read once (76 B/point at K=8)
    fused_ms = prof.get("p2c_fit_fused_f32", {}).get("ms", 0.0) / a.steps

    # ---- CPU side: the oracle's closed-form restatement on a subset, same inputs
    c = min(a.cpu_clouds, a.clouds)
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    t1 = time.perf_counter()
    E_cpu = R.estimate_extrusion_axis(X[:c], Wb[:c], Wc[:c], None, None, normalize=False, literal=False)
    cen_cpu, found_cpu = R.hard_centroids(onehot[:c], pcs[:c])
    ext_cpu, _ = R.get_extrusion_extents(pcs[:c], seg[:c], bb[:c], E_cpu, cen_cpu, {(k, b): rand_idx[b, k] for k in range(K) for b in range(c)})
    cpu_dt = time.perf_counter() - t1
    # extents parity proper: the oracle on the SAME axes / centres the device path used (with its own axes the figure would mostly
    # show the fp32 eigenvector differences, which the axis fields below report)
    ext_same, _ = R.get_extrusion_extents(pcs[:c], seg[:c], bb[:c], E_AX[:c].cpu(), cen[:c].cpu(), {(k, b): rand_idx[b, k] for k in range(K) for b in range(c)})

    # float64 run of the same restatement: the yardstick for the two float32 paths (the metric is an acos next to its clamp,
    # so two correct fp32 implementations differ from each other by more than 1e-4 relative)
    E_64 = R.estimate_extrusion_axis(X[:c].double(), Wb[:c].double(), Wc[:c].double(), None, None, normalize=False, literal=False)
    ang = lambda A, Bv: torch.acos((A.double() * Bv.double()).sum(-1).abs().clamp(max=1.0)) * 180.0 / np.pi
    err_64 = angle_error_deg(E_64.float(), axes[:c], seg[:c], K)
    err_gpu = angle_error_deg(E_AX[:c].cpu(), axes[:c], seg[:c], K)
    err_cpu = angle_error_deg(E_cpu, axes[:c], seg[:c], K)
    err_all = angle_error_deg(E_AX.cpu(), axes, seg, K)
    present = ((torch.nn.functional.one_hot(seg[:c].clamp_min(0), K) * (seg[:c] >= 0).unsqueeze(-1)).sum(1) > 0)
    axis_absdot = ((E_AX[:c].cpu() * E_cpu).sum(-1).abs())[present]
    # the whole path (axis + centroids + extents): every input read once - normals, points, both membership matrices, both label arrays,
    # the pre-drawn sample indices (they are an input of this boundary: data_utils.py:1690 draws them on the host)
    path_bytes = points * (12 + 12 + 2 * K * 4 + seg.element_size() + bb.element_size()) + a.clouds * K * S * rand_idx.element_size()
    kern_bytes, kern_ms = (path_bytes, fused_ms) if fused else (axis_bytes, axis_ms)
    line = dict(metric="fitting-only cylinders/sec (axis + centroid + extent), 10k pre-segmented cylinders at N=8192",
                value=round(a.clouds * K / dt, 1), unit="cylinders/s", n_gpus=1, steps=a.steps, ms_per_step=round(dt * 1e3, 3),
                points_per_s=round(points / dt, 1), dtype="f32", data="synthetic",
                config=dict(workload="configs[3]: %d clouds x K=%d x N=%d, X = gt normals + 2 deg angular noise, one-hot W, S=%d"
                                     % (a.clouds, K, N, S), kernels="one pass per cloud (fit_fused)" if fused else "axis, centroids, extents"),
                roofline=dict(bound="hbm", kernel="p2c_fit_fused_f32" if fused else "p2c_extrusion_axis_f32",
                              achieved=round(kern_bytes / (kern_ms * 1e-3) / 1e9, 1) if kern_ms else None,
                              peak=8000.0, unit="GB/s", frac=round(kern_bytes / (kern_ms * 1e-3) / 1e9 / 8000.0, 4) if kern_ms else None,
                              algorithmic_bytes_per_launch=kern_bytes, avg_launch_us=round(kern_ms * 1e3, 1), traffic=None,
                              path=dict(algorithmic_bytes=path_bytes, ms=round(dt * 1e3, 3), achieved=round(path_bytes / dt / 1e9, 1),
                                        frac=round(path_bytes / dt / 1e9 / 8000.0, 4))),
                cpu_baseline=dict(value=round(c * K / cpu_dt, 1), unit="cylinders/s", cores=torch.get_num_threads(), kind="port",
                                  sample="oracle closed-form axis + hard centroids + extents on %d clouds, %.1f s" % (c, cpu_dt)),
                parity=dict(axis_angle_error_deg_gpu=round(err_gpu, 5), axis_angle_error_deg_cpu=round(err_cpu, 5),
                            rel_diff=round(abs(err_gpu - err_cpu) / max(err_cpu, 1e-12), 7),
                            rel_diff_vs_f64=round(abs(err_gpu - err_64) / max(err_64, 1e-12), 7),
                            rel_diff_cpu32_vs_f64=round(abs(err_cpu - err_64) / max(err_64, 1e-12), 7), axis_angle_error_deg_all=round(err_all, 5),
                            axis_angle_error_deg_f64=round(err_64, 5),
                            max_axis_angle_deg_gpu_vs_f64=round(float(ang(E_AX[:c].cpu(), E_64)[present].max()), 5),
                            max_axis_angle_deg_cpu32_vs_f64=round(float(ang(E_cpu, E_64)[present].max()), 5),
                            min_abs_dot_gpu_vs_cpu=round(float(axis_absdot.min()), 7),
                            centroid_max_abs_diff=float((cen[:c].cpu() - cen_cpu).abs().max()),
                            extent_max_abs_diff=float((ext[:, :c].cpu() - ext_same).abs().max()),
                            extent_max_abs_diff_own_axes=float((ext[:, :c].cpu() - ext_cpu).abs().max())),
                kernels={k: dict(ms_per_step=round(v["ms"] / a.steps, 3), launches_per_step=v["launches"] / a.steps)
                         for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])})
    print(json.dumps(line))


if __name__ == "__main__":
    main()
