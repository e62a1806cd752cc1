"""BASELINE configs[4] on ONE GPU: a step of the with-sketch trainer (train_Point2Cyl.py:376-700 with --pred_seg --pred_normal
--pred_bb --is_pc_train --is_im_train --with_im_loss) at B=16 clouds x N=8192 points, K=8, NUM_SK_POINT=2048:

    backbone forward + segmentation / normal / base-barrel losses  (the step bench.py measures, at half its batch)
  + projection of the predicted and the ground-truth segmentation, sketch encoder (trainable) + frozen ground-truth encoder,
    implicit decoder losses with their double backward, latent loss       (point2cyl_amd/step_sketch.py)
  + backward of everything + Adam over backbone and encoder parameters.

    python tools/bench_config5.py [--steps 5] [--train_decoder]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 tools/bench_config5.py     # the 8-GPU form

The implicit decoder is frozen by default like in the reference's optimiser (train_Point2Cyl.py:298-321: only `model` and `pn_encoder`
parameters are optimised); --train_decoder also asks for its weight gradients.  One JSON line; the matrix products of the decoder
dominate (>95 % of the step), so the roofline object prices them against the fp32 MFMA peak."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

PEAK_MFMA = 157.3e12


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--steps", type=int, default=10); ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no_graph", action="store_true", help="launch the step from Python (with the reference's CPU sampling draws) instead of replaying a HIP graph")
    ap.add_argument("--glue", action="store_true", help="after the timing: one eager step under torch.profiler, every aten op that launched device work, by shape (stderr)")
    ap.add_argument("--batch", type=int, default=16); ap.add_argument("--train_decoder", action="store_true"); a = ap.parse_args()
    from point2cyl_amd import ddp, fitting, ops, step, step_sketch, synth
    import torch.distributed as dist
    from point2cyl_amd.backbone import backbone
    from point2cyl_amd.implicit import ImplicitNet, NormalPerPoint
    from point2cyl_amd.sketch import PointNetEncoder
    rank, world, local = ddp.init_from_env()         # one process per GPU, clouds sharded by rank, one flat gradient all-reduce per step
    dev = torch.device("cuda", local); torch.cuda.set_device(dev)
    torch.cuda.set_stream(torch.cuda.Stream(dev))      # everything on one non-default stream (a HIP graph cannot be captured on the default one)
    B, N, K, S = a.batch, 8192, 8, 2048
    fl = step.StepFlags(K=K)
    pcs, nrm, seg, bb, _, _, axes, _, cen = synth.make_batch(B, N, K, seed=1234 + 1000 * rank)
    batch = tuple(x.to(dev) for x in (pcs.float(), nrm.float(), seg, bb, axes.float(), cen.float()))
    torch.manual_seed(0)
    model = backbone(output_sizes=fl.pred_sizes()).to(dev).train()
    enc, enc_gt = PointNetEncoder(256, 2, with_normals=True).to(dev).train(), PointNetEncoder(256, 2, with_normals=True).to(dev).eval()
    dec = ImplicitNet(d_in=258, dims=[512] * 8, skip_in=[4], geometric_init=True, radius_init=1, beta=100).to(dev)
    for p in enc_gt.parameters(): p.requires_grad_(False)
    for p in dec.parameters(): p.requires_grad_(a.train_decoder)
    sampler = NormalPerPoint(1.0, 0.01)
    # synthetic ground-truth sketches (dataloader.py's sampled_sketch, (B,K,S,4) = [point | normal]): the ground-truth projection itself
    torch.manual_seed(5)
    P0, X0, s0, _ = fitting.sketch_implicit_projection2(*batch[:1], batch[1], batch[2], batch[3], batch[4], batch[5], S)
    gt_sk = torch.cat([(P0 / s0.unsqueeze(-1).unsqueeze(-1)), F.normalize(X0 + 1e-6, dim=-1)], -1).permute(1, 0, 2, 3).contiguous()
    for m_ in (model, enc, enc_gt, dec):
        ddp.broadcast_module(m_)
    params = list(model.parameters()) + list(enc.parameters()) + (list(dec.parameters()) if a.train_decoder else [])
    sync = ddp.FlatGradSync(params, world)
    opt = torch.optim.Adam(params, lr=1e-3, fused=True)

    def fwd_bwd(geom=None, device_draws=False):
        ops.step_done()
        with ops.step_arena(dev):
            out = step.compute_losses_fused(model, *batch, fl, geom=geom)
            h = out["heads"].view(B, N, -1)
            with torch.no_grad():
                X = F.normalize(h[:, :, 0:3], p=2, dim=2, eps=1e-12)
                W2K = torch.softmax(h[:, :, 3:3 + 2 * K], dim=2)
                W = W2K[:, :, 0::2] + W2K[:, :, 1::2]
            sk = step_sketch.sketch_branch_losses(batch[0], X, W, W2K, out["match"], out["mask"], batch[1], batch[2], batch[3], batch[4], batch[5], gt_sk,
                                                  enc, enc_gt, dec, sampler, K, S, device_draws=device_draws)
            total = out["total"] + sk["im_loss"]
            sync.zero()
            total.backward()
            sync.pack()
        return dict(total=total.detach(), im_loss=sk["im_loss"].detach())

    graphed, launch = None, "eager"
    if not a.no_graph:
        # the whole forward + losses + double backward as ONE HIP graph (point2cyl_amd/graph.py, as for the without-sketch step): the geometry
        # of the next batch on the forked stream, the projections' sample draws from the device generator (no host sync inside the step)
        from point2cyl_amd.graph import GraphedForwardBackward
        try:
            graphed = GraphedForwardBackward(model, lambda geom=None: fwd_bwd(geom, device_draws=True), prefetch_xyz=batch[0],
                                             stream=torch.cuda.current_stream())
            launch = "hip_graph(fwd + losses + double backward, next batch's geometry on a forked stream) + eager(allreduce, adam)"
        except Exception as e:
            sys.stderr.write("bench_config5: HIP graph capture failed (%s: %s); running eager\n" % (type(e).__name__, e))
            torch.cuda.set_stream(torch.cuda.Stream(dev))
            for m_ in model.modules():
                if hasattr(m_, "fps_start"):
                    m_.fps_start = None
            graphed = None

    def one_step():
        out = graphed() if graphed is not None else fwd_bwd()
        sync.allreduce()
        opt.step()
        ops.step_done()
        return out["total"], out

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup): one_step()
    fence(); t0 = time.perf_counter()
    for _ in range(a.steps): total, sk = one_step()
    fence(); dt = (time.perf_counter() - t0) / a.steps
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    if graphed is not None:
        graphed.starts.cursor = 0
    ops.PROFILE.reset(enabled=rank == 0); fwd_bwd(); sync.allreduce(); opt.step(); ops.step_done()      # one EAGER step for the per-kernel HIP events
    prof = ops.PROFILE.summary(); ops.PROFILE.enabled = False                                         # (every rank: the step has a collective)
    if a.glue and rank == 0:
        import collections
        from torch.profiler import profile, ProfilerActivity
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as tp:
            fwd_bwd(); sync.allreduce(); opt.step(); ops.step_done(); torch.cuda.synchronize()
        rows = collections.defaultdict(lambda: [0, 0.0])
        for e in tp.events():
            if e.device_type.name != "CPU" or not e.name.startswith("aten::") or not e.kernels:
                continue
            rows[(e.name, str(e.input_shapes)[:70])][0] += len(e.kernels)
            rows[(e.name, str(e.input_shapes)[:70])][1] += sum(k.duration for k in e.kernels)
        tot_n = tot_us = 0
        for (name, shp), (n, us) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
            if us >= 50.0:
                sys.stderr.write("%-26s x%-3d %9.1f us  %s\n" % (name, n, us, shp))
            tot_n += n; tot_us += us
        sys.stderr.write("torch-side device launches of one eager with-sketch step: %d, %.2f ms\n" % (tot_n, tot_us / 1e3))
    if world > 1:
        fence()
        dist.destroy_process_group()
    if rank != 0:
        return
    from point2cyl_amd import _lib
    split = bool(_lib.lib().p2c_get_mfma_mode())
    gemm = {k: v for k, v in prof.items() if v["flops"] > 0}
    fl_all, ms_all = sum(v["flops"] for v in gemm.values()), sum(v["ms"] for v in gemm.values())
    print(json.dumps(dict(
        metric="with-sketch training-step points/sec (BxN) at N=8192", value=round(world * B * N / dt, 1), unit="points/s", n_gpus=world, scaling="weak", steps=a.steps, warmup=a.warmup,
        ms_per_step=round(dt * 1e3, 2), higher_is_better=True, dtype="f32", data="synthetic",
        config=dict(workload="configs[4]: B=%d clouds/GPU x N=%d, K=%d, %d points per sketch; backbone + seg/normal/bb losses + projection + sketch encoder "
                             "+ implicit decoder losses (decoder %s) + latent loss; fwd + bwd (double backward through the decoder) + Adam"
                             % (B, N, K, S, "trainable" if a.train_decoder else "frozen, as in the reference's optimiser"),
                    loss=round(float(total), 5), im_loss=round(float(sk["im_loss"]), 5), launch=launch),
        roofline=dict(bound="mfma", kernel="gemm_kernel family (all matrix products of the step)", achieved=round(fl_all / ms_all / 1e9, 2),
                      peak=round(2500.0 / 6, 1) if split else 157.3, unit="TFLOP/s (fp32-equivalent)",
                      frac=round(fl_all / ms_all / 1e9 / (2500.0 / 6 if split else 157.3), 4),
                      frac_of_f32_mfma_peak=round(fl_all / ms_all / 1e9 / 157.3, 4), frac_of_split_ceiling=round(fl_all / ms_all / 1e9 / (2500.0 / 6), 4),
                      mfma="bf16x3-split: six bf16 products per fp32 product, ceiling 2500 / 6 = 417 TFLOP/s" if split else "f32",
                      gflop_per_step=round(fl_all / 1e9, 1), gemm_ms_per_step=round(ms_all, 2),
                      share_of_step=round(ms_all / (dt * 1e3), 3), whole_step_tflops=round(fl_all / dt / 1e12, 2), traffic=None),
        kernels={k: dict(ms_per_step=round(v["ms"], 3), launches=v["launches"]) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])[:12]})))


if __name__ == "__main__":
    main()
