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
        ops.step_done()
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
    # implicit decoder at the trainer's widths: manifold / eikonal / SALD losses and their double backward (train_Point2Cyl.py:608-648)
    from point2cyl_amd.implicit import ImplicitNet, add_latent, gradient
    from point2cyl_amd import losses as LS
    dec = ImplicitNet(d_in=258, dims=[512] * 8, skip_in=[4], geometric_init=True, radius_init=1, beta=100).to(dev)
    gen = torch.Generator().manual_seed(3)
    sk = (Pp / sc.unsqueeze(-1).unsqueeze(-1)).reshape(B * K, S, 2).detach()
    skn = torch.nn.functional.normalize(Xp.reshape(B * K, S, 2) + 1e-6, dim=-1).detach()
    non = torch.cat([sk + 0.01 * torch.randn(B * K, S, 2, generator=gen).to(dev), (torch.rand(B * K, S // 8, 2, generator=gen) * 2 - 1).to(dev)], 1)
    mask_gt = (found.reshape(B, K) > 0)
    lat0 = torch.nn.functional.normalize(torch.randn(B * K, 256, generator=gen)).to(dev)

    def decode():
        lat = lat0.clone().requires_grad_(True)
        a_ = add_latent(sk, lat).requires_grad_(); n_ = add_latent(non, lat).requires_grad_()
        fa, fn = dec(a_), dec(n_)
        ga, gn = gradient(a_, fa).reshape(B, K, -1, 2), gradient(n_, fn).reshape(B, K, -1, 2)
        mn = LS.reduce_mean_masked_instance(fa.reshape(B, K, -1, 1).abs().mean(-1).mean(-1), mask_gt).mean()
        ek = LS.reduce_mean_masked_instance(((gn.norm(2, dim=-1) - 1) ** 2).mean(-1), mask_gt).mean()
        nr = skn.reshape(B, K, -1, 2)
        nl = LS.reduce_mean_masked_instance(torch.minimum((ga - nr).norm(2, dim=-1), (ga + nr).norm(2, dim=-1)).mean(-1), mask_gt).mean()
        for p in dec.parameters(): p.grad = None
        (mn + 0.1 * ek + nl).backward()
    t_dec = timed(decode)
    ops.PROFILE.reset(enabled=True)
    for _ in range(2): decode()
    prof_dec = ops.PROFILE.summary(); ops.PROFILE.enabled = False
    rows_dec = B * K * (S + S + S // 8)
    w_flops = 2.0 * (260 * 512 + 3 * 512 * 512 + 512 * 256 + 3 * 512 * 512 + 512 * 4)        # per row, padded widths as executed
    dec_gemm_flops = sum(v["flops"] for v in prof_dec.values()) / 2
    dec_gemm_ms = sum(v["ms"] for v in prof_dec.values() if v["flops"] > 0) / 2
    dec_act_ms = sum(v["ms"] for v in prof_dec.values() if v["flops"] == 0) / 2
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
        metric="sketch branch: projection + PointNetEncoder fwd+bwd + implicit decoder losses fwd+double-bwd, sketches/sec (B=%d clouds x K=%d segments x %d samples)" % (B, K, S),
        value=round(B * K / (t_proj + t_enc + t_dec), 1), unit="sketches/s", dtype="f32", data="synthetic", n_gpus=1, steps=a.steps,
        projection=dict(ms=round(t_proj * 1e3, 3), algorithmic_mb=round(proj_bytes / 1e6, 2), gbs=round(proj_bytes / t_proj / 1e9, 1),
                        frac_hbm=round(proj_bytes / t_proj / PEAK_HBM, 4), found=int(found.sum().item())),
        encoder=dict(ms=round(t_enc * 1e3, 3), rows=M, gflop_fwd_bwd=round(3 * fl_fwd / 1e9, 1), tflops=round(3 * fl_fwd / t_enc / 1e12, 2),
                     frac_mfma=round(3 * fl_fwd / t_enc / PEAK_MFMA, 4)),
        decoder=dict(ms=round(t_dec * 1e3, 2), rows=rows_dec, gemm_gflop=round(dec_gemm_flops / 1e9, 1), gemm_ms=round(dec_gemm_ms, 2),
                     gemm_tflops=round(dec_gemm_flops / dec_gemm_ms / 1e9, 2), frac_mfma_gemm=round(dec_gemm_flops / dec_gemm_ms / 1e9 / (PEAK_MFMA / 1e12), 4),
                     tflops_whole=round(dec_gemm_flops / t_dec / 1e12, 2), frac_mfma_whole=round(dec_gemm_flops / t_dec / PEAK_MFMA, 4),
                     gemm_launches=int(sum(v["launches"] for v in prof_dec.values() if v["flops"] > 0) / 2), softplus_ms=round(dec_act_ms, 2),
                     note="fwd + d f/d point (create_graph) + losses + double backward; matrix products on csrc/gemm.hip, softplus and its derivatives on csrc/softplus.hip"),
        cpu_baseline=dict(kind="port", cores=torch.get_num_threads(), encoder_sketches_per_s=round(c / cpu_enc, 2),
                          projection_clouds_per_s=round(2 / cpu_proj, 2),
                          sample="oracle encoder fwd+bwd on %d sketches (%.1f s), oracle projection on 2 clouds (%.1f s)" % (c, cpu_enc, cpu_proj)),
        kernels={k: dict(us_per_pass=round(v["ms"] / 3 * 1e3, 1), launches=v["launches"] // 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])})))


if __name__ == "__main__":
    main()
