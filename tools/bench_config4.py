"""BASELINE configs[3] ("config 4" in SURVEY numbering): the eval.py fitting-only path on 10 000 pre-segmented synthetic
cylinders = 1250 clouds x K=8 segments x N=8192 points, 1 GPU.

    python tools/bench_config4.py [--clouds 1250] [--steps 20] [--unfused]

Inputs: X = ground-truth normals + Gaussian angular noise (sigma = 2 deg), W_barrel / W_base one-hot from
the labels.  Timed on the GPU, inputs resident: estimate_extrusion_axis (data_utils.py:99-177), per-segment hard centroids
(eval.py:409-436), get_extrusion_extents with S = 2048 samples (data_utils.py:1650-1730; the random draws are made up
front, they are host work in the reference too).  Reported next to it: the oracle's CPU restatement - the closed form on ALL clouds at
os.cpu_count() threads, 32 threads and 1 thread, the literal N x N diag_embed form (data_utils.py:126-163) on a 2-cloud subset -, the HBM
roofline by the path's own bytes and by SURVEY 8(d)'s 76 B/point, and the parity of the eval metric (batch-mean axis-angle error in degrees,
eval.py:398-405) against the oracle run in float64 on all clouds.  One JSON line on stdout."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def cpu_fitting_legs(w, E_dev, cen_dev, ext_dev, E64_dev, literal_clouds=2, thread_counts=None):
    """The oracle side of configs[3] on workload `w` (measure.FittingWorkload): float64 closed form on all clouds (the parity yardstick),
    the fp32 closed form timed at three thread counts, the literal diag_embed form on a small subset.  -> (cpu_baseline dict, parity dict)."""
    from oracle import ref_torch as R
    from point2cyl_amd import synth
    c = w.cpu
    n, K = w.n, w.K
    X, Wb, Wc, pcs, seg, bb, axes, onehot, ridx = (c[k] for k in ("X", "Wb", "Wc", "pcs", "seg", "bb", "axes", "onehot", "ridx"))
    chunk = 125

    def closed(dtype, sl=slice(None)):
        return torch.cat([R.estimate_extrusion_axis(X[sl][i:i + chunk].to(dtype), Wb[sl][i:i + chunk].to(dtype), Wc[sl][i:i + chunk].to(dtype),
                                                    None, None, normalize=False, literal=False) for i in range(0, X[sl].shape[0], chunk)])

    host = os.cpu_count() or 1
    timings = {}
    E_cpu = None
    for thr in sorted(set(thread_counts or (host, min(32, host), 1)) | {1}, reverse=True):
        torch.set_num_threads(thr)
        sl = slice(None) if thr > 1 else slice(0, max(1, n // 10))          # one thread: a tenth of the clouds, scaled
        t0 = time.perf_counter()
        E = closed(torch.float32, sl)
        dt = time.perf_counter() - t0
        timings[thr] = dict(cylinders_per_s=round(E.shape[0] * K / dt, 1), seconds=round(dt, 2), clouds=E.shape[0])
        if E.shape[0] == n:
            E_cpu = E
    best = max(timings, key=lambda t: timings[t]["cylinders_per_s"])
    torch.set_num_threads(min(32, host))
    # literal N x N diag_embed form (268 MB per diag_embed and sample): a small subset, its own rate
    lc = min(literal_clouds, n)
    t0 = time.perf_counter()
    E_lit = R.estimate_extrusion_axis(X[:lc], Wb[:lc], Wc[:lc], None, None, normalize=False, literal=True)
    lit_dt = time.perf_counter() - t0
    # the whole path on the CPU (closed-form axis + hard centroids + extents) on 64 clouds
    cc = min(64, n)
    t0 = time.perf_counter()
    E_c = R.estimate_extrusion_axis(X[:cc], Wb[:cc], Wc[:cc], None, None, normalize=False, literal=False)
    cen_c, _ = R.hard_centroids(onehot[:cc], pcs[:cc])
    ext_c, _ = R.get_extrusion_extents(pcs[:cc], seg[:cc], bb[:cc], E_c, cen_c, {(k, b): ridx[b, k] for k in range(K) for b in range(cc)})
    path_dt = time.perf_counter() - t0
    cpu = dict(value=timings[best]["cylinders_per_s"], unit="cylinders/s", cores=best, kind="port", host_cores=host,
               closed_form_axis_by_threads={str(t): v for t, v in timings.items()},
               literal_diag_embed_axis=dict(cylinders_per_s=round(lc * K / lit_dt, 2), clouds=lc, seconds=round(lit_dt, 2), threads=min(32, host)),
               whole_path_closed_form=dict(cylinders_per_s=round(cc * K / path_dt, 1), clouds=cc, seconds=round(path_dt, 2), threads=min(32, host)),
               sample="oracle closed-form axis fit (data_utils.py:99-177 with (wX)^T(wX) instead of the N x N diag_embed) on all %d clouds at %s "
                      "threads (1 thread: %d clouds); `value` is the fastest; literal diag_embed form on %d clouds; axis + centroids + extents "
                      "on %d clouds" % (n, "/".join(str(t) for t in timings if t > 1), timings[1]["clouds"], lc, cc),
               note="fastest at %d of %d hardware threads (the closed form is K passes of small batched products: bound by memory traffic and "
                    "thread start-up rather than by cores, so the full thread count is not automatically the fastest)" % (best, host))
    # ---- parity (float64 yardstick on ALL clouds)
    E_64 = closed(torch.float64)
    m64 = synth.axis_angle_error_deg64(E_64, axes, seg, K)
    m_gpu64 = synth.axis_angle_error_deg64(E64_dev.cpu(), axes, seg, K)
    m_gpu32 = synth.axis_angle_error_deg64(E_dev.cpu(), axes, seg, K)
    m_cpu32 = synth.axis_angle_error_deg64(E_cpu, axes, seg, K) if E_cpu is not None else None
    m_lit = synth.axis_angle_error_deg64(E_lit, axes[:lc], seg[:lc], K)
    m_lit64 = synth.axis_angle_error_deg64(E_64[:lc], axes[:lc], seg[:lc], K)
    present = (onehot.sum(1) > 0)
    well = present & (Wb.sum(1) > 50) & (Wc.sum(1) > 50)
    sin = torch.linalg.cross(E64_dev.cpu(), E_64).norm(dim=-1)
    ext_same, _ = R.get_extrusion_extents(pcs[:cc], seg[:cc], bb[:cc], E_dev[:cc].cpu(), cen_dev[:cc].cpu(),
                                          {(k, b): ridx[b, k] for k in range(K) for b in range(cc)})
    rel = lambda a, b: round(abs(a - b) / max(b, 1e-300), 9)
    parity = dict(metric="batch-mean axis-angle error in degrees over the segments that exist (eval.py:398-405), evaluated in float64",
                  clouds=n, axis_angle_error_deg_f64_oracle=round(m64, 8), axis_angle_error_deg_gpu_axis64=round(m_gpu64, 8),
                  rel_diff_vs_f64=rel(m_gpu64, m64), tolerance=1e-4,
                  rel_diff_vs_f64_fp32_stored_axes=rel(m_gpu32, m64),
                  rel_diff_cpu32_closed_form_vs_f64=None if m_cpu32 is None else rel(m_cpu32, m64),
                  rel_diff_cpu32_literal_vs_f64_on_subset=rel(m_lit, m_lit64),
                  max_sin_gpu_axis64_vs_f64_well_conditioned=float(sin[well].max()),
                  centroid_max_abs_diff=float((cen_dev[:cc].cpu() - cen_c).abs().max()),
                  extent_max_abs_diff_same_axes=float((ext_dev[:, :cc].cpu() - ext_same).abs().max()))
    return cpu, parity


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clouds", type=int, default=1250)
    ap.add_argument("--num_point", type=int, default=8192)
    ap.add_argument("--K", type=int, default=8)
    ap.add_argument("--samples", type=int, default=2048)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--unfused", action="store_true", help="the three separate kernels (axis, centroids, extents) instead of the one-pass kernel")
    ap.add_argument("--no_cpu", action="store_true", help="skip the CPU baseline and parity legs (kernel A/B runs)")
    a = ap.parse_args()
    from point2cyl_amd import measure
    w = measure.FittingWorkload(a.clouds, a.num_point, a.K, a.samples)
    res, (E, cen, cfound, ext, found, E64) = w.time(a.steps, fused=not a.unfused)
    cpu, parity = (None, None) if a.no_cpu else cpu_fitting_legs(w, E, cen, ext, E64)
    line = dict(metric="fitting-only cylinders/sec (axis + centroid + extent), 10k pre-segmented cylinders at N=8192",
                value=res["cylinders_per_s"], unit="cylinders/s", n_gpus=1, steps=a.steps, ms_per_step=res["ms"],
                points_per_s=res["points_per_s"], dtype="f32 (scatter sums, eigen-solve and the axis output in f64)", data="synthetic",
                config=dict(workload=res["workload"], kernels=res["kernels"]),
                roofline=dict(bound="hbm", kernel=res["kernel"], achieved=round((res["kernel_frac_hbm"] or res["frac_hbm_path_bytes"] or 0) * 8000.0, 1), peak=8000.0,
                              unit="GB/s", frac=res["kernel_frac_hbm"] or res["frac_hbm_path_bytes"], avg_launch_us=res["kernel_us"], traffic=None,
                              path=dict(ms=res["ms"], algorithmic_bytes=res["path_bytes"], frac=res["frac_hbm_path_bytes"],
                                        survey_76B_per_point_bytes=res["survey_bytes_76_per_point"], frac_76B_per_point=res["frac_hbm_76B_per_point"])),
                cpu_baseline=cpu, parity=parity, kernels=res["per_kernel"], route=res.get("route"), vs_general_route=res.get("vs_general_route"),
                soft_membership_route=res.get("soft_membership_route"))
    print(json.dumps(line))


if __name__ == "__main__":
    main()
