"""Stage-level measurements of the hot path on the device, shared by bench.py and tools/.

Everything here times HIP-graph replays or back-to-back launches of the PRODUCT path with the inputs resident in HBM; nothing in this module
touches the oracle (the CPU legs live in bench.py / tools/, which may import it).  Three measurements beside the training step:

  * sa1_stage      the north-star stage "FPS + ball query + grouped MLP forward" (SA1: pointnet_util.py:63-143, :166-207) in train mode;
  * forward_only   the whole backbone forward (pointnet_extrusion.py:37-66), train-mode BatchNorm, geometry included;
  * fitting        BASELINE configs[3]: eval.py's fitting-only path on pre-segmented cylinders (fitting.fit_cylinders).

and the per-stage roofline model of section 8(d) (path_roofline)."""
import time

import torch

from . import fitting, ops, synth

PEAK_F32_MFMA = 157.3e12          # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA = 2500.0e12        # v_mfma_f32_32x32x16_bf16, dense
PEAK_SPLIT = PEAK_BF16_MFMA / 6   # bf16x3 split: six bf16 products per fp32 product
PEAK_HBM = 8.0e12

# forward FLOPs and module-boundary bytes per SAMPLE (N = 8192 points, heads [3, 2K = 16])
STAGES = (("sa1", 0.818e9, 366592), ("sa2", 1.080e9, 400896), ("sa3", 0.185e9, 136716), ("fp3", 0.101e9, 267776),
          ("fp2", 0.134e9, 663040), ("fp1", 0.805e9, 4560896), ("head", 0.308e9, 4816896))
FPS_ISSUE_US_PER_ITER = 8192 * 9 / 64.0 / 2.4e3   # 8192 points x ~9 VALU operations on the ONE CU a cloud occupies = 1150 issue cycles at 2.4 GHz
                                                  # before any reduction: the issue-bound floor of one FPS iteration


def path_roofline(B, ms, work_factor=1.0, N=8192):
    """roofline.achieved = sum over stages of max(bytes / BW, flops / peak) divided by the measured time, per-stage terms
    included.  work_factor 1 = forward; 3 = the training step (forward + two backward products and, to the same approximation, three
    passes over the module-boundary tensors).  Two matrix peaks: the fp32-MFMA instruction and the bf16x3-split ceiling."""
    scale = N / 8192.0
    rows, t32, tsp, thbm = {}, 0.0, 0.0, 0.0
    for name, fl, by in STAGES:
        f, b = fl * B * work_factor * scale, by * B * work_factor * scale
        a, s, h = f / PEAK_F32_MFMA, f / PEAK_SPLIT, b / PEAK_HBM
        rows[name] = dict(gflop=round(f / 1e9, 2), mbytes=round(b / 1e6, 2), mfma_f32_us=round(a * 1e6, 1), mfma_split_us=round(s * 1e6, 1),
                          hbm_us=round(h * 1e6, 1), bound="mfma" if a > h else "hbm")
        t32 += max(a, h)
        tsp += max(s, h)
        thbm += h
    return dict(model="sum over stages of max(bytes / 8 TB/s, flops / peak); module-boundary bytes and forward FLOPs of SURVEY 8(d) x %g" % work_factor,
                floor_ms_f32_mfma=round(t32 * 1e3, 4), floor_ms_split=round(tsp * 1e3, 4), floor_ms_hbm_only=round(thbm * 1e3, 4),
                frac_f32_mfma=round(t32 * 1e3 / ms, 4), frac_split=round(tsp * 1e3 / ms, 4), measured_ms=round(ms, 4), stages=rows)


def capture(fn, warmup=2):
    """fn() on static tensors -> (HIP graph, fn's captured outputs).  Warm-up + capture on a private stream."""
    cap = torch.cuda.Stream()
    cap.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(cap):
        for _ in range(warmup):
            fn()
        cap.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=cap, capture_error_mode="thread_local"):
            out = fn()
    torch.cuda.current_stream().wait_stream(cap)
    return gr, out


def replay_ms(graphs, steps, warmup=3):
    for _ in range(warmup):
        for g in graphs:
            g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        graphs[i % len(graphs)].replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


class _KeepBuffers:
    """BatchNorm running statistics / counters and the dropout counter of `model` are restored on exit: measuring must not train."""

    def __init__(self, model):
        self.model = model

    def __enter__(self):
        self.keep = [(b, b.detach().clone()) for b in self.model.buffers()]
        s = getattr(self.model, "_drop_seed", None)
        self.seed = None if s is None else s.detach().clone()
        return self

    def __exit__(self, *exc):
        torch.cuda.synchronize()
        with torch.no_grad():
            for b, v in self.keep:
                b.copy_(v)
            if self.seed is not None and getattr(self.model, "_drop_seed", None) is not None:
                self.model._drop_seed.copy_(self.seed)
        return False


def sa1_stage(model, xyz, steps=30):
    """The north-star stage at xyz's shape (B, N, 3): FPS(512) -> ball query(0.2, 64) -> grouped gather -> 3 shared-MLP layers with
    train-mode BatchNorm -> max over the 64 neighbours, as HIP-graph replays (the way the training step launches it).
    -> dict(ms serial, ms of the three parts alone, points/s, fractions of the stage's MFMA roofline and of the latency model
    `512 dependent FPS iterations x the measured time per iteration + ball query + the MLP's MFMA floor`)."""
    dev = xyz.device
    B, N, _ = xyz.shape
    sa1 = model.sa1
    hook = sa1.fps_start
    start = torch.randint(0, N, (B,)).to(dev)
    sa1.fps_start = start
    fl, by = STAGES[0][1] * B * N / 8192.0, STAGES[0][2] * B * N / 8192.0

    def geometry():
        g = sa1.geometry(xyz)
        g["X0"] = ops.group_gather(xyz, None, g["new_xyz"], g["group_idx"], None)
        return g

    def mlp(g):
        ops.step_done()
        with torch.no_grad(), ops.step_arena(dev):
            return sa1.forward_pm(xyz, None, g)

    try:
        with _KeepBuffers(model):
            g0 = geometry()
            gr_serial, _ = capture(lambda: mlp(geometry()))
            gr_mlp, _ = capture(lambda: mlp(g0))
            gr_fps, _ = capture(lambda: ops.fps(xyz, sa1.npoint, start))
            gr_bq, _ = capture(lambda: ops.ball_query(sa1.radius, sa1.nsample, xyz, g0["new_xyz"]))
            t_serial, t_mlp = replay_ms([gr_serial], steps), replay_ms([gr_mlp], steps)
            t_fps, t_bq = replay_ms([gr_fps], steps), replay_ms([gr_bq], steps)
            ops.step_done()
            # pipelined: the stage's geometry for batch i + 1 (FPS -> ball query -> grouped coordinates) on a forked stream INSIDE the graph
            # that runs batch i's grouped MLP on the geometry the previous replay produced; hand-over copies after the join
            side = torch.cuda.Stream()
            cur_g = {k: v.clone() for k, v in g0.items()}

            def piped():
                cap = torch.cuda.current_stream()
                side.wait_stream(cap)
                with torch.cuda.stream(side):
                    nxt = geometry()
                out = mlp(cur_g)
                cap.wait_stream(side)
                ks = sorted(cur_g)
                ops.copy_flat_batch([cur_g[k] for k in ks], [nxt[k] for k in ks])
                return out

            gr_pipe, _ = capture(piped)
            t_pipe = replay_ms([gr_pipe], steps)
            ops.step_done()
            # throughput pipeline: the geometry of the next G batches computed TOGETHER (one FPS launch over G x B clouds: G x B workgroups, the same
            # 512 dependent steps) on the forked stream while the grouped MLP of the current G batches runs batch by batch (BatchNorm per batch)
            groups = {}
            for G in (2, 3, 4, 8):
                xg = xyz.repeat(G, 1, 1)
                sa1.fps_start = start.repeat(G)

                def geometry_g():
                    g = sa1.geometry(xg, with_csr=False)
                    g["X0"] = ops.group_gather(xg, None, g["new_xyz"], g["group_idx"], None)
                    return g

                def mlp_g(g, j):
                    ops.step_done()
                    sl = {k: v[j * (v.shape[0] // G):(j + 1) * (v.shape[0] // G)] for k, v in g.items()}
                    with torch.no_grad(), ops.step_arena(dev):
                        return sa1.forward_pm(xg[j * B:(j + 1) * B], None, sl)

                cg = {k: v.clone() for k, v in geometry_g().items()}

                def piped_g():
                    cap = torch.cuda.current_stream()
                    side.wait_stream(cap)
                    with torch.cuda.stream(side):
                        nxt = geometry_g()
                    outs = [mlp_g(cg, j) for j in range(G)]
                    cap.wait_stream(side)
                    ks = sorted(cg)
                    ops.copy_flat_batch([cg[k] for k in ks], [nxt[k] for k in ks])
                    return outs

                gr_g, _ = capture(piped_g)
                tg = replay_ms([gr_g], max(6, steps // G)) / G
                groups[str(G)] = dict(ms_per_batch=round(tg, 4), points_per_s=round(B * N / (tg * 1e-3), 1), frac_of_mfma_roofline=round(fl / PEAK_F32_MFMA * 1e3 / tg, 4))
                ops.step_done()
            sa1.fps_start = start
    finally:
        sa1.fps_start = hook
    it_us = t_fps * 1e3 / sa1.npoint
    mfma_floor_ms = fl / PEAK_F32_MFMA * 1e3
    bq_floor_ms = (B * N * 12 + B * sa1.npoint * (12 + 4 * sa1.nsample)) / PEAK_HBM * 1e3
    lat_model_ms = sa1.npoint * it_us * 1e-3 + bq_floor_ms + mfma_floor_ms
    lat_issue_ms = sa1.npoint * FPS_ISSUE_US_PER_ITER * 1e-3 + bq_floor_ms + mfma_floor_ms
    return dict(workload="SA1 forward: FPS(%d of %d) + ball query(r=%.1f, %d) + grouped MLP 3->64->64->128 + max-pool, B=%d, train-mode BatchNorm"
                         % (sa1.npoint, N, sa1.radius, sa1.nsample, B),
                graph_serial_ms=round(t_serial, 4), points_per_s=round(B * N / (t_serial * 1e-3), 1),
                pipelined_ms=round(t_pipe, 4), points_per_s_pipelined=round(B * N / (t_pipe * 1e-3), 1),
                frac_of_mfma_roofline_pipelined=round(mfma_floor_ms / t_pipe, 4),
                pipelined_frac_of_fps_latency_model=round(sa1.npoint * it_us * 1e-3 / t_pipe, 4),
                pipelined_groups=groups,
                best=(lambda g: dict(mode="pipelined, geometry groups of %s batches" % g, ms_per_batch=groups[g]["ms_per_batch"],
                                     points_per_s=groups[g]["points_per_s"], frac_of_mfma_roofline=groups[g]["frac_of_mfma_roofline"],
                                     note="the stage's throughput figure (north_star: >= 0.40 of its fp32-MFMA roofline); graph_serial_ms / "
                                          "frac_of_mfma_roofline are its single-batch LATENCY, which the 512-step FPS chain bounds"))(
                    max(groups, key=lambda g: groups[g]["frac_of_mfma_roofline"])) if groups else None,
                pipelined_groups_note="steady-state THROUGHPUT per batch of %d clouds when the stage's geometry (FPS -> ball query -> grouped coordinates) of the "
                                      "next G batches is computed as ONE launch chain over G x %d clouds on the forked stream, under the grouped MLPs of the "
                                      "current G batches (BatchNorm statistics per batch): FPS is 512 dependent steps on one CU per cloud - 0.52 ms for 32 clouds "
                                      "and 0.6 - 0.7 ms for 128 - so its latency is shared by the group; graph.PipelinedForward(group=G) runs the whole backbone "
                                      "this way" % (B, B),
                pipelined_note="steady state per batch of: grouped MLP of batch i (on the previous replay's geometry) || FPS + ball query + grouped "
                               "coordinates of batch i + 1 on a forked stream, one graph; bounded below by the FPS chain (512 dependent iterations on "
                               "one CU per cloud = parts_ms.fps); pipelined_frac_of_fps_latency_model = parts_ms.fps / pipelined_ms",
                parts_ms=dict(fps=round(t_fps, 4), ball_query=round(t_bq, 4), grouped_mlp_with_pool=round(t_mlp, 4)),
                gflop=round(fl / 1e9, 2), module_boundary_mbytes=round(by / 1e6, 2),
                mfma_floor_ms=round(mfma_floor_ms, 4), hbm_floor_ms=round(by / PEAK_HBM * 1e3, 5), bound="mfma",
                frac_of_mfma_roofline=round(mfma_floor_ms / t_serial, 4),
                frac_of_split_roofline=round(fl / PEAK_SPLIT * 1e3 / t_serial, 4),
                mlp_frac_of_mfma_roofline=round(mfma_floor_ms / t_mlp, 4),
                fps_latency_model=dict(iterations=sa1.npoint, measured_us_per_iteration=round(it_us, 4),
                                       issue_bound_us_per_iteration=round(FPS_ISSUE_US_PER_ITER, 4),
                                       model_ms=round(lat_model_ms, 4), frac_of_model=round(lat_model_ms / t_serial, 4),
                                       model_ms_issue_bound=round(lat_issue_ms, 4), frac_of_issue_bound_model=round(lat_issue_ms / t_serial, 4),
                                       note="model = 512 dependent iterations x time per iteration (one workgroup = one CU per cloud) + ball-query "
                                            "HBM floor + the MLP's fp32-MFMA floor; `measured` uses the FPS kernel's own time alone on the chip, "
                                            "`issue_bound` the 1150 VALU issue cycles per iteration at 2.4 GHz"))


def forward_only(model, pcs, steps=30):
    """The whole backbone forward in train mode (batch statistics, always-on dropout), geometry included, no autograd: one HIP graph."""
    dev = pcs.device
    B, N, _ = pcs.shape
    hooks = [(m, m.fps_start) for m in (model.sa1, model.sa2)]
    model.sa1.fps_start = torch.randint(0, N, (B,)).to(dev)
    model.sa2.fps_start = torch.randint(0, model.sa1.npoint, (B,)).to(dev)

    def fwd():
        ops.step_done()
        with torch.no_grad(), ops.step_arena(dev):
            return model.forward_heads(pcs)[0]

    def fwd_geom(g):
        ops.step_done()
        with torch.no_grad(), ops.step_arena(dev):
            return model.forward_heads(pcs, g)[0]

    try:
        with _KeepBuffers(model):
            gr, _ = capture(fwd)
            t = replay_ms([gr], steps)
            with torch.no_grad():
                g0 = model.compute_geometry(pcs)
            gr2, _ = capture(lambda: fwd_geom(g0))
            t2 = replay_ms([gr2], steps)
            ops.step_done()
    finally:
        for m, h in hooks:
            m.fps_start = h
    # pipelined (what point2cyl_amd.eval runs): the geometry of batch i + 1 on a forked stream inside the graph of batch i's forward
    t3 = None
    with _KeepBuffers(model):
        from .graph import PipelinedForward
        cur = torch.cuda.current_stream()
        if cur == torch.cuda.default_stream():
            cur = torch.cuda.Stream()
            cur.wait_stream(torch.cuda.default_stream())
        with torch.cuda.stream(cur):
            pf = PipelinedForward(model, pcs, stream=cur)
            try:
                for _ in range(3):
                    pf(pcs)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(steps):
                    pf(pcs)
                torch.cuda.synchronize()
                t3 = (time.perf_counter() - t0) / steps * 1e3
            finally:
                pf.release()
        torch.cuda.current_stream().wait_stream(cur)
        ops.step_done()
    # ... and with the geometry of FOUR batches computed together, one group ahead (the eval CLI's default): FPS's 512 dependent steps per
    # cloud cost the same for 128 clouds as for 32, so the chain's latency is shared by four batches
    t4, G = None, 4
    try:
        with _KeepBuffers(model):
            from .graph import PipelinedForward
            cur = torch.cuda.current_stream()
            grp = [pcs] * G
            pf = PipelinedForward(model, grp, stream=cur, group=G)
            try:
                for _ in range(2):
                    pf(grp)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                n4 = max(4, steps // G)
                for _ in range(n4):
                    pf(grp)
                torch.cuda.synchronize()
                t4 = (time.perf_counter() - t0) / (n4 * G) * 1e3
            finally:
                pf.release()
            ops.step_done()
    except Exception as e:          # (reported, not fatal: the other figures of this leg stand)
        t4 = None
        grp_err = "%s: %s" % (type(e).__name__, e)
    # ... and the INFERENCE forward (eval-mode BatchNorm: what point2cyl_amd.eval runs - running statistics, no statistics passes, the folded /
    # pooled layer forms without their activations in HBM): geometry precomputed, and pipelined in groups of four
    inf = {}
    was_training = model.training
    try:
        model.eval()
        with torch.no_grad():
            g0 = model.compute_geometry(pcs, with_csr=False)
        gr3, _ = capture(lambda: fwd_geom(g0))
        inf["ms_geometry_precomputed"] = round(replay_ms([gr3], steps), 4)
        ops.step_done()
        from .graph import PipelinedForward
        cur = torch.cuda.current_stream()
        grp = [pcs] * 4
        pf = PipelinedForward(model, grp, stream=cur, group=4)
        try:
            for _ in range(2):
                pf(grp)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n4 = max(4, steps // 4)
            for _ in range(n4):
                pf(grp)
            torch.cuda.synchronize()
            inf["ms_pipelined_group4"] = round((time.perf_counter() - t0) / (n4 * 4) * 1e3, 4)
            inf["points_per_s_pipelined_group4"] = round(B * N / (inf["ms_pipelined_group4"] * 1e-3), 1)
        finally:
            pf.release()
        ops.step_done()
    except Exception as e:
        inf["error"] = "%s: %s" % (type(e).__name__, e)
    finally:
        model.train(was_training)
    return dict(workload="backbone forward, B=%d x N=%d, train-mode BatchNorm + dropout, FPS / ball query / 3-NN included, one HIP graph" % (B, N),
                inference_eval_mode=inf,
                ms=round(t, 4), points_per_s=round(B * N / (t * 1e-3), 1),
                ms_geometry_precomputed=round(t2, 4), points_per_s_geometry_precomputed=round(B * N / (t2 * 1e-3), 1),
                ms_pipelined=round(t3, 4), points_per_s_pipelined=round(B * N / (t3 * 1e-3), 1),
                ms_pipelined_group4=None if t4 is None else round(t4, 4), points_per_s_pipelined_group4=None if t4 is None else round(B * N / (t4 * 1e-3), 1),
                pipelined_group4_note=("per batch of %d clouds, geometry of 4 batches per forked-stream launch, one group ahead (graph.PipelinedForward(group=4), "
                                       "the eval CLI's default --prefetch_group 4)" % B) if t4 is not None else grp_err,
                pipelined_note="steady state of graph.PipelinedForward (point2cyl_amd/eval.py): per batch one graph = forward of batch i on the geometry "
                               "the previous replay produced + FPS / ball query / 3-NN / inverse maps of batch i + 1 on a forked stream + the hand-over "
                               "copies; includes the host-side FPS start draws and the copy of the next clouds",
                path_roofline=path_roofline(B, t, 1.0, N), path_roofline_pipelined=path_roofline(B, t3, 1.0, N))


class FittingWorkload:
    """BASELINE configs[3]: n_clouds x K segments x N points of pre-segmented synthetic cylinders (synth.make_fitting_inputs), S pre-drawn
    samples per segment (the reference draws them on the host: data_utils.py:1696).  cpu: the host tensors; dev: their device copies."""

    def __init__(self, n_clouds=1250, N=8192, K=8, S=2048, seed=4321, device="cuda:0"):
        self.n, self.N, self.K, self.S = n_clouds, N, K, S
        pcs, X, seg, bb, axes, Wb, Wc, onehot = synth.make_fitting_inputs(n_clouds, N, K, seed)
        g = torch.Generator().manual_seed(7)
        counts = Wb.sum(1).long()
        ridx = torch.randint(0, 1 << 30, (n_clouds, K, S), generator=g) % counts.clamp_min(1).unsqueeze(-1)
        self.cpu = dict(pcs=pcs, X=X, seg=seg, bb=bb, axes=axes, Wb=Wb, Wc=Wc, onehot=onehot, ridx=ridx)
        self.dev = {k: v.to(device) for k, v in self.cpu.items()}
        self.points = n_clouds * N
        # every input read once: normals, points, both membership matrices, both label arrays, the pre-drawn sample indices
        self.path_bytes = self.points * (12 + 12 + 2 * K * 4 + 8 + 8) + n_clouds * K * S * 8
        self.survey_bytes = self.points * (12 + 2 * K * 4)        # X + W_barrel + W_base = 76 B/point at K = 8

    def fit(self, fused=True, validate=False, hard=False):
        """hard: the memberships implied by the labels (the workload's W ARE the one-hot encodings of (seg, bb)): not read."""
        d = self.dev
        if fused and ops.fit_fused_supported(self.N, self.K, self.S):
            return fitting.fit_cylinders(d["X"], None if hard else d["Wb"], None if hard else d["Wc"], d["bb"], d["seg"], d["pcs"],
                                         rand_idx=d["ridx"], normalize=False, return_float64=True, validate=validate, K=self.K)
        with torch.no_grad():
            E, E64 = fitting.estimate_extrusion_axis(d["X"], d["Wb"], d["Wc"], d["bb"], d["seg"], normalize=False, return_float64=True)
            cen, cfound = ops.segment_centroids(d["pcs"], d["seg"], self.K)
            ext, found = fitting.get_extrusion_extents(d["pcs"], d["seg"], d["bb"], E, cen, self.S, rand_idx=d["ridx"])
        return E, cen, cfound, ext, found, E64

    def time(self, steps=20, fused=True):
        """-> (dict of timings and roofline fractions, the outputs of the last pass)."""
        out = self.fit(fused, validate=True)
        for _ in range(2):
            out = self.fit(fused)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = self.fit(fused)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        ops.PROFILE.reset(enabled=True)          # per-kernel times: HIP events around every launch, in passes of their own
        for _ in range(steps):
            self.fit(fused)
        ops.PROFILE.enabled = False
        prof = ops.PROFILE.summary()
        ops.PROFILE.reset()
        fused = fused and ops.fit_fused_supported(self.N, self.K, self.S)
        kname = "p2c_fit_fused_f32" if fused else "p2c_extrusion_axis_f32"
        kms = prof.get(kname, {}).get("ms", 0.0) / steps
        kbytes = self.path_bytes if fused else self.survey_bytes
        res = dict(workload="configs[3]: %d clouds x K=%d x N=%d = %d cylinders, X = gt normals + 2 deg angular noise, one-hot W, S=%d"
                            % (self.n, self.K, self.N, self.n * self.K, self.S),
                   kernels="one pass per cloud (fit_fused)" if fused else "axis, centroids, extents",
                   ms=round(dt * 1e3, 4), cylinders_per_s=round(self.n * self.K / dt, 1), points_per_s=round(self.points / dt, 1),
                   kernel=kname, kernel_us=round(kms * 1e3, 1),
                   path_bytes=self.path_bytes, frac_hbm_path_bytes=round(self.path_bytes / dt / PEAK_HBM, 4),
                   survey_bytes_76_per_point=self.survey_bytes, frac_hbm_76B_per_point=round(self.survey_bytes / dt / PEAK_HBM, 4),
                   kernel_frac_hbm=round(kbytes / (kms * 1e-3) / PEAK_HBM, 4) if kms else None,
                   per_kernel={k: dict(ms_per_pass=round(v["ms"] / steps, 4), launches_per_pass=v["launches"] / steps)
                               for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])})
        if fused:
            # The same clouds with the memberships IMPLIED by the labels (this workload's W are exactly the one-hot encodings of (seg, bb):
            # "pre-segmented cylinders"; SURVEY 8(d): "16 B of labels if one-hot is implied"): Wb / Wc are not read, 40 B per point.
            # ONE route per object: the top level of the result is the labels-implied route - wall time, the kernel's own
            # HIP-event time and both fractions - and `soft_membership_route` holds the same figures of the route that reads Wb / Wc.
            oh = self.fit(True, hard=True)
            for _ in range(2):
                oh = self.fit(True, hard=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                oh = self.fit(True, hard=True)
            torch.cuda.synchronize()
            dh = (time.perf_counter() - t0) / steps
            ops.PROFILE.reset(enabled=True)
            for _ in range(steps):
                self.fit(True, hard=True)
            ops.PROFILE.enabled = False
            hprof = ops.PROFILE.summary()
            ops.PROFILE.reset()
            hms = hprof.get(kname, {}).get("ms", 0.0) / steps
            hbytes = self.points * (12 + 12 + 8 + 8) + self.n * self.K * self.S * 8
            dot = (oh[5] * out[5]).sum(-1).abs().clamp(max=1.0)
            soft = {k: res[k] for k in ("ms", "cylinders_per_s", "points_per_s", "kernel_us", "path_bytes", "frac_hbm_path_bytes",
                                        "frac_hbm_76B_per_point", "kernel_frac_hbm")}
            soft["route"] = "Wb / Wc (B,N,K) fp32 read: what soft (predicted) memberships need, 104 B/point"
            res = dict(workload=res["workload"], route="labels-implied memberships (pre-segmented clouds): Wb = Wc = NULL, 40 B/point + the pre-drawn samples",
                       kernels="one pass per cloud (p2c_fit_fused_f32) + the batch-level finish", kernel=kname,
                       ms=round(dh * 1e3, 4), cylinders_per_s=round(self.n * self.K / dh, 1), points_per_s=round(self.points / dh, 1),
                       kernel_us=round(hms * 1e3, 1), path_bytes=hbytes, frac_hbm_path_bytes=round(hbytes / dh / PEAK_HBM, 4),
                       kernel_frac_hbm=round(hbytes / (hms * 1e-3) / PEAK_HBM, 4) if hms else None,
                       survey_bytes_76_per_point=self.survey_bytes, frac_hbm_76B_per_point=round(self.survey_bytes / dh / PEAK_HBM, 4),
                       per_kernel={k: dict(ms_per_pass=round(v["ms"] / steps, 4), launches_per_pass=v["launches"] / steps)
                                   for k, v in sorted(hprof.items(), key=lambda kv: -kv[1]["ms"])},
                       vs_general_route=dict(max_axis_angle_deg=float(torch.rad2deg(torch.acos(dot)).max()),
                                             max_centroid_diff=float((oh[1] - out[1]).abs().max()), max_extent_diff=float((oh[3] - out[3]).abs().max()),
                                             found_masks_equal=bool(torch.equal(oh[2], out[2]) and torch.equal(oh[4], out[4]))),
                       soft_membership_route=soft)
            out = oh
        return res, out


def power_cap_probe(dev, M=262144, Co=128, Ci=128):
    """How much of the dominant kernel's time is CLOCK: the fused backward of a 128 x 128 layer at M = 262,144 (the family bench.py's roofline
    object reports) launched (a) in isolation - one launch, a synchronisation and 4 ms of idle time, so that it starts on a cool socket at the
    boost clock - and (b) back to back for ~0.3 s, where the socket power limit sets the shader clock (2.0 GHz at ~1350 W
    against 2.4 GHz).  -> dict(sustained_us, the shader clock / package power rocm-smi reports during (b), after_idle_us for (a))."""
    import re
    import subprocess
    dZ = torch.randn(M, Co, device=dev)
    Y = torch.randn(M, Co, device=dev)
    X = torch.randn(M, Ci, device=dev)
    W = torch.randn(Co, Ci, device=dev) * 0.1
    coef = torch.rand(5, Co, device=dev)
    sc, sh = torch.rand(Ci, device=dev) + 0.5, torch.randn(Ci, device=dev) * 0.1
    dX = torch.empty(M, Ci, device=dev)
    dW8 = torch.zeros(8, Co, Ci, device=dev)
    pstat = torch.rand(4, Ci, device=dev) + 0.5
    parts = torch.zeros(ops.STAT_SLOTS, 2, Ci, device=dev, dtype=torch.float64)

    def run():
        ops.call("p2c_linear_bwd_fused_f32", ops.ptr(dZ), Co, ops.ptr(Y), Co, 1, ops.ptr(coef), None, 0, ops.ptr(X), Ci, 1, ops.ptr(sc), ops.ptr(sh),
                 ops.ptr(W), Ci, ops.ptr(dX), Ci, ops.ptr(dW8), Ci, Co * Ci, None, ops.ptr(pstat), ops.ptr(parts), M, Co, Ci, ops.stream())

    for _ in range(3):
        run()
    torch.cuda.synchronize()
    iso = []
    for _ in range(12):
        time.sleep(0.004)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(); e1.record()
        torch.cuda.synchronize()
        iso.append(e0.elapsed_time(e1) * 1e3)
    iso.sort()
    for _ in range(1500):          # ~0.2 s: the socket reaches its power limit
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 600
    for _ in range(n):
        run()
    e1.record()
    sclk = power = None
    try:                                                   # sampled while the queue above is still draining
        for _ in range(3000):
            run()
        o = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=20).stdout
        p = re.search(r"Package Power \(W\): ([\d.]+)", o)
        c = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", o)
        power, sclk = (float(p.group(1)) if p else None), (int(c.group(1)) if c else None)
    except Exception:
        pass
    torch.cuda.synchronize()
    sus = e0.elapsed_time(e1) * 1e3 / n
    med = iso[len(iso) // 2]
    nbytes = 4.0 * M * (2 * Co + 2 * Ci)
    return dict(kernel="p2c_linear_bwd_fused_f32 (Co = Ci = 128, M = %d, the role-split bf16x3 kernel)" % M,
                sustained_us=round(sus, 1), sustained_hbm_frac=round(nbytes / (sus * 1e-6) / PEAK_HBM, 4), sclk_mhz_sustained=sclk,
                package_power_w_sustained=power, boost_mhz=2400, clock_frac_of_boost=None if not sclk else round(sclk / 2400.0, 3),
                after_idle_us=round(med, 1), after_idle_min_us=round(iso[0], 1),
                note="sustained = 600 launches back to back after 1500 warm-up launches: the socket sits at its power limit and rocm-smi reports the shader "
                     "clock it then holds - clock_frac_of_boost of the 2.4 GHz the part can run at is the share of this kernel's rate that is power, not "
                     "code (its phases are issue-bound: time scales with 1 / clock).  after_idle = one launch after 4 ms of idle time, median of 12: "
                     "NOT a boost-clock figure (the clock has dropped and ramps inside the launch); kept to show that an isolated launch is no faster")
