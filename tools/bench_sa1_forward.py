"""The north-star stage target: the "FPS + ball-query + grouped-MLP" forward (SA1: pointnet_util.py:63-143, :166-207) at
B=32, N=8192 on one GPU, train-mode BatchNorm (batch statistics, as the reference's forward always runs in the trainer).

    python tools/bench_sa1_forward.py [--steps 50]

Reports points/s and the roofline fractions of SURVEY 8(d) for this stage (26.2 GFLOP of fp32 MFMA work, 11.7 MB at the
module boundary: the stage is MFMA-bound by two orders of magnitude, and FPS is a chain of 512 dependent steps per cloud), in two
schedules: `serial` (geometry, then the MLP, one stream: the latency of ONE forward) and `pipelined` (the parameter-free geometry
of batch t+1 on a second stream while the MLP of batch t runs: the throughput of a stream of batches, which is how the training
step uses it).  One JSON line."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

PEAK_MFMA, PEAK_HBM = 157.3e12, 8.0e12
FLOPS_PER_SAMPLE = 0.818e9            # SA1 = 2*512*64*(3*64 + 64*64 + 64*128) flop
BYTES_PER_SAMPLE = 366592             # xyz in (98,304) + new_xyz, new_feats out, int32 idx internal


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--steps", type=int, default=50); ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--num_point", type=int, default=8192); a = ap.parse_args()
    from point2cyl_amd import ops, synth
    from point2cyl_amd.backbone import backbone
    dev = torch.device("cuda:0"); B, N = a.batch, a.num_point
    xyz = synth.make_batch(B, N, 8, seed=1234)[0].float().to(dev).contiguous()
    torch.manual_seed(0)
    sa1 = backbone(output_sizes=[3, 16]).to(dev).train().sa1
    start = torch.randint(0, N, (B,)).to(dev)
    sa1.fps_start = start

    def geometry():
        g = sa1.geometry(xyz)
        g["X0"] = ops.group_gather(xyz, None, g["new_xyz"], g["group_idx"], None)
        return g

    def mlp(g):
        ops.step_done()
        with torch.no_grad(), ops.step_arena(dev):
            return sa1.forward_pm(xyz, None, g)

    def timed(fn, steps):
        for _ in range(3): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(steps): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / steps

    t_serial = timed(lambda: mlp(geometry()), a.steps)
    t_geom = timed(geometry, a.steps)
    g0 = geometry()
    t_mlp = timed(lambda: mlp(g0), a.steps)
    side, main_s = torch.cuda.Stream(), torch.cuda.current_stream()
    state = {"g": geometry()}
    def piped():
        side.wait_stream(main_s)
        with torch.cuda.stream(side):
            nxt = geometry()
        mlp(state["g"])
        main_s.wait_stream(side)
        state["g"] = nxt
    t_pipe = timed(piped, a.steps)

    # The same two schedules as HIP-graph replays - how the training step runs the stage.  Launched from Python the ~12 launches of the
    # MLP cost more host time than the GPU needs for them, so the eager figures above are launch-bound (mlp_only in particular).
    def capture(fn):
        cap = torch.cuda.Stream()
        cap.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(cap):
            for _ in range(2): fn()
            cap.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=cap):
                out = fn()
        torch.cuda.current_stream().wait_stream(cap)
        return gr, out

    gr_serial, _ = capture(lambda: mlp(geometry()))
    gr_mlp, _ = capture(lambda: mlp(g0))
    # pipelined: graph A reads geometry set gA and computes set gB next to it on a forked stream; graph B reads gB and refills gA
    side2 = torch.cuda.Stream()
    gA = geometry()
    keys = [k for k, v in gA.items() if torch.is_tensor(v)]

    def stage_with_next(cur, dst=None):
        capst = torch.cuda.current_stream()
        side2.wait_stream(capst)
        with torch.cuda.stream(side2):
            nxt = geometry()
            if dst is not None:
                for k in keys: dst[k].copy_(nxt[k])
        mlp(cur)
        capst.wait_stream(side2)
        return nxt

    gr_a, gB = capture(lambda: stage_with_next(gA))
    gr_b, _ = capture(lambda: stage_with_next(gB, gA))

    def replay_timed(graphs, steps):
        for _ in range(3):
            for g in graphs: g.replay()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(steps): graphs[i % len(graphs)].replay()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / steps

    t_gserial = replay_timed([gr_serial], a.steps)
    t_gmlp = replay_timed([gr_mlp], a.steps)
    t_gpipe = replay_timed([gr_a, gr_b], a.steps)
    ops.PROFILE.reset(enabled=True)
    for _ in range(3): mlp(geometry())
    prof = ops.PROFILE.summary(); ops.PROFILE.enabled = False
    fl, by = FLOPS_PER_SAMPLE * B, BYTES_PER_SAMPLE * B
    floor = max(fl / PEAK_MFMA, by / PEAK_HBM)
    mk = lambda t: dict(ms=round(t * 1e3, 3), points_per_s=round(B * N / t, 1), tflops=round(fl / t / 1e12, 2),
                        frac_of_stage_roofline=round(floor / t, 4), module_boundary_gbs=round(by / t / 1e9, 1))
    print(json.dumps(dict(metric="SA1 forward (FPS + ball query + grouped MLP + max-pool) points/sec, B=%d N=%d, train-mode BN" % (B, N),
                          dtype="f32", data="synthetic", n_gpus=1, steps=a.steps,
                          stage=dict(gflop=round(fl / 1e9, 2), module_boundary_mb=round(by / 1e6, 2), mfma_floor_us=round(fl / PEAK_MFMA * 1e6, 1),
                                     hbm_floor_us=round(by / PEAK_HBM * 1e6, 2), bound="mfma"),
                          graph_serial=mk(t_gserial), graph_pipelined=mk(t_gpipe), graph_mlp_only=mk(t_gmlp),
                          serial=mk(t_serial), pipelined=mk(t_pipe), geometry_only=dict(ms=round(t_geom * 1e3, 3)), mlp_only=mk(t_mlp),
                          note="graph_*: HIP-graph replays (GPU-bound, as in the training step); the other figures are launched from Python and are "
                               "host-bound wherever the kernels are short",
                          kernels={k: dict(us_per_pass=round(v["ms"] / 3 * 1e3, 1), launches=v["launches"] // 3) for k, v in
                                   sorted(prof.items(), key=lambda kv: -kv[1]["ms"])})))


if __name__ == "__main__":
    main()
