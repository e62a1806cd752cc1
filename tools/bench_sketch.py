"""SURVEY 8(f) rank 1 at the shapes of BASELINE configs[4] (the with-sketch trainer: B=16 clouds/GPU, N=8192, K=8,
NUM_SK_POINT=2048; train_Point2Cyl.py:36, :549-558): sketch_implicit_projection on ground-truth labels, then PointNetEncoder
(IGR/network.py:132-174) forward + backward on the B*K = 128 projected sketches.  python tools/bench_sketch.py [--steps 20]
One JSON line: times, the encoder's MFMA roofline fraction (its five 1x1-conv layers are 77.4 GFLOP forward), the projection's
HBM figures, and the oracle's CPU restatement on a subset next to it."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

PEAK_MFMA, PEAK_HBM = 157.3e12, 8.0e12


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--steps", type=int, default=20); ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--cpu_sketches", type=int, default=8); a = ap.parse_args()
    from point2cyl_amd import fitting, ops, synth
    from point2cyl_amd.sketch import PointNetEncoder
    from oracle import ref_torch as R
    dev = torch.device("cuda:0"); B, N, K, S = a.batch, 8192, 8, 2048
    pcs, nrm, seg, bb, _, _, axes, _, cen = synth.make_batch(B, N, K, seed=1234)
    pcs, nrm, axes, cen = pcs.float(), nrm.float(), axes.float(), cen.float()
    torch.manual_seed(0)
    ridx = fitting._barrel_draws(seg, bb, K, S)
    d = [x.to(dev) for x in (pcs, nrm, seg, bb, axes, cen)]
    ridx_d = ridx.to(dev)
    enc = PointNetEncoder(256, 2, with_normals=True).to(dev).train()
    sd_cpu = {k: v.detach().cpu().clone() for k, v in enc.state_dict().items()}

    def project():
        return fitting.sketch_implicit_projection2(*d, S, rand_idx=ridx_d)

    def encode(Pp, Xp, sc):
        with ops.step_arena(dev):
            q = (Pp / sc.unsqueeze(-1).unsqueeze(-1)).reshape(B * K, S, 2)                # train_Point2Cyl.py:592-596
            x = torch.cat((q, Xp.reshape(B * K, S, 2)), -1)
            z = enc(x)
            for p in enc.parameters(): p.grad = None
            z.square().sum().backward()
        return z

    def timed(fn):
        for _ in range(3): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(a.steps): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / a.steps

    t_proj = timed(project)
    Pp, Xp, sc, found = project()
    t_enc = timed(lambda: encode(Pp, Xp, sc))
    ops.PROFILE.reset(enabled=True)
    for _ in range(3): project(); encode(Pp, Xp, sc)
    prof = ops.PROFILE.summary(); ops.PROFILE.enabled = False
    M = B * K * S
    fl_fwd = 2.0 * M * (4 * 64 + 64 * 64 + 64 * 64 + 64 * 128 + 128 * 1024)
    proj_bytes = B * N * 16 + B * K * S * (8 + 24 + 16)          # labels once; per sample its draw, point + normal, two float2 out
    # CPU: the oracle on a few sketches (same inputs)
    c = min(a.cpu_sketches, B * K)
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    x_cpu = torch.cat(((Pp / sc.unsqueeze(-1).unsqueeze(-1)).reshape(B * K, S, 2), Xp.reshape(B * K, S, 2)), -1).cpu()[:c].clone().requires_grad_(True)
    sd = {k: (v.clone().requires_grad_(True) if v.dtype == torch.float32 and "running" not in k else v.clone()) for k, v in sd_cpu.items()}
    t0 = time.perf_counter(); zc = R.pointnet_encoder_forward(sd, x_cpu, training=True); zc.square().sum().backward(); cpu_enc = time.perf_counter() - t0
    t0 = time.perf_counter()
    rk = {(k, b): ridx[b, k] for k in range(K) for b in range(2)}
    R.sketch_implicit_projection(pcs[:2], nrm[:2], seg[:2], bb[:2], axes[:2], cen[:2], rk, S); cpu_proj = time.perf_counter() - t0
    print(json.dumps(dict(
        metric="sketch branch: projection + PointNetEncoder fwd+bwd, sketches/sec (B=%d clouds x K=%d segments x %d samples)" % (B, K, S),
        value=round(B * K / (t_proj + t_enc), 1), unit="sketches/s", dtype="f32", data="synthetic", n_gpus=1, steps=a.steps,
        projection=dict(ms=round(t_proj * 1e3, 3), algorithmic_mb=round(proj_bytes / 1e6, 2), gbs=round(proj_bytes / t_proj / 1e9, 1),
                        frac_hbm=round(proj_bytes / t_proj / PEAK_HBM, 4), found=int(found.sum().item())),
        encoder=dict(ms=round(t_enc * 1e3, 3), rows=M, gflop_fwd_bwd=round(3 * fl_fwd / 1e9, 1), tflops=round(3 * fl_fwd / t_enc / 1e12, 2),
                     frac_mfma=round(3 * fl_fwd / t_enc / PEAK_MFMA, 4)),
        cpu_baseline=dict(kind="port", cores=torch.get_num_threads(), encoder_sketches_per_s=round(c / cpu_enc, 2),
                          projection_clouds_per_s=round(2 / cpu_proj, 2),
                          sample="oracle encoder fwd+bwd on %d sketches (%.1f s), oracle projection on 2 clouds (%.1f s)" % (c, cpu_enc, cpu_proj)),
        kernels={k: dict(us_per_pass=round(v["ms"] / 3 * 1e3, 1), launches=v["launches"] // 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])})))


if __name__ == "__main__":
    main()
