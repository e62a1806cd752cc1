"""Point2Cyl hot-path bench: training-step points/sec at N=8192 (BASELINE.json configs[1]).

    python bench.py [--gpus N --steps K --warmup W] [--full_losses]

N > 1: either launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` (one rank per GPU, the
driver's form), or - when no WORLD_SIZE is in the environment - bench.py starts those N ranks ITSELF by re-executing under
torch.distributed.run on 127.0.0.1.  Either way the run fails loudly unless exactly N ranks on N distinct GPUs take part.

A step = backbone forward + (seg, normal, base/barrel) losses + backward + Adam on one synthetic batch of
B=32 clouds x 8192 points per GPU, already resident in HBM.  Rank 0 prints ONE compact JSON line (< 4 KB: compact_line) on stdout, last;
the detailed objects (per-kernel table, stage / forward / fitting / drop-in / evaluation legs) go to stderr as "extra" lines before it and to
gpurun_out/bench_extras.json.  Default legs finish in about a minute; --extras all runs every leg.
"""
import argparse
import json
import os
import re
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # (dmabuf IPC: what RCCL's multi-process paths need on this driver; before HIP initialises)

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0    # same guide: v_mfma_f32_32x32x16_bf16, dense (the 5 PF headline is 2:1 sparsity)
PEAK_HBM_GBS = 8000.0


# HBM bytes per launch of a kernel family from the committed rocprofv3 --pmc summary of this same command (FETCH_SIZE and
# WRITE_SIZE need separate passes, so they cannot be read inside the timed run); launch-weighted over the family's kernels.
DEFAULT_EXTRAS = "ab,stages,config3_fitting,eval_loop,rccl_selftest"
ALL_EXTRAS = "path_roofline,power_cap,stages,forward_only,ab,config3_fitting,config3_cpu,dropin,eval_loop,eval_serial,cpu_threads,rccl_selftest"
LINE_LIMIT = 4096            # bytes of the last stdout line (round 5's 20 KB line could not be parsed from the driver's capture)
_PROFILES = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")


def _newest_pmc():
    """profiles/r<NN>*_pmc_step.json of the highest round (then newest file), written by tools/collect_pmc.py (every kernel of the step)."""
    best = None
    try:
        for f in os.listdir(_PROFILES):
            m = re.match(r"r(\d+)([a-z]*)_pmc_step\.json$", f)
            if m:
                key = (int(m.group(1)), m.group(2), os.path.getmtime(os.path.join(_PROFILES, f)))
                if best is None or key > best[0]:
                    best = (key, os.path.join(_PROFILES, f))
    except OSError:
        pass
    return best[1] if best else None


# entry point -> kernel name stems (fp32-MFMA kernel | bf16x3-split kernels: one wave per SIMD, role-split)
_PMC_NAME = {"p2c_linear_bwd_fused_f32": ("bwd_fused_pp_kernel", "bwd_fused3_kernel", "bwd_fused3r_kernel"),
             "p2c_linear_fwd_f32": ("fwd_pp_kernel", "fwd_pp3_kernel")}


def _pmc_traffic(entry):
    """-> dict(per_step, kernel_launches_per_step, by_kernel, source, source_hash, stale) or None.  The counters need their own rocprofv3
    passes, so `traffic` cannot be measured by the run that prints it: the line names the file it was read from and says whether the
    kernels have changed since (stale = the sources hash differently now).  Everything is reported PER STEP and per CALL of the entry
    point (one call = one layer's backward; the 256-wide layer runs as two kernel launches of one call), never per kernel launch."""
    pmc = _newest_pmc()
    if pmc is None:
        return None
    try:
        with open(pmc) as f:
            doc = json.load(f)
        ks = doc["kernels"]
        steps = float(doc.get("steps_in_trace") or 1)
        from point2cyl_amd.build import source_hash
        shash = doc.get("source_hash")
        stale = (shash != source_hash()) if shash else None
        subs = _PMC_NAME.get(entry, ())
        # (the IMODE-2 instantiation "<.., .., .., 2, ...>" belongs to p2c_linear_bwd_fused_fold0_f32, a different entry point)
        sel = {k: v for k, v in ks.items() if any(re.match(r"(void )?%s<" % sub, k) for sub in subs)
               and not re.search(r"bwd_fused_pp_kernel<\d+, \d+, \d+, 2,", k)}
        if not sel:
            return None
        by = {k: dict(launches_per_step=v["launches"] / steps,
                      bytes_per_launch=round(v.get("fetch_bytes_per_launch", 0.0) + v.get("write_bytes_per_launch", 0.0)),
                      mfma_busy=(round(v["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * v["SQ_BUSY_CU_CYCLES"]), 3)
                                 if v.get("SQ_BUSY_CU_CYCLES") and v.get("SQ_VALU_MFMA_BUSY_CYCLES") is not None else None))
              for k, v in sel.items()}
        per_step = sum(d["launches_per_step"] * d["bytes_per_launch"] for d in by.values())
        return dict(per_step=round(per_step), kernel_launches_per_step=sum(d["launches_per_step"] for d in by.values()), by_kernel=by,
                    source="profiles/" + os.path.basename(pmc), source_hash=shash, stale=stale)
    except (OSError, KeyError, ValueError, TypeError):
        return None


def _cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.lower().startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _cpu_all_threads_leg(cb, N, K, threads, limit_s):
    """One CPU training step of the oracle at torch.set_num_threads(os.cpu_count()) in a child process, killed after limit_s seconds.
    -> (points/s or None, seconds or None, note)."""
    import subprocess
    code = ("import sys, json, torch; sys.path.insert(0, %r)\n"
            "from point2cyl_amd import synth\nfrom oracle import ref_step\n"
            "b = synth.make_batch(%d, %d, %d, seed=1234)\n"
            "pps, sec, thr, n = ref_step.time_cpu_baseline(tuple(x.contiguous() for x in b[:4]), threads=%d, steps=1)\n"
            "print(json.dumps([pps, sec, thr]))\n" % (ROOT, cb, N, K, threads))
    t0 = time.perf_counter()
    try:
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=limit_s + 8.0)     # + import time
        pps, sec, thr = json.loads(out.stdout.strip().splitlines()[-1])
        return pps, sec, "one step on %d threads: %.1f s" % (thr, sec)
    except subprocess.TimeoutExpired:
        el = time.perf_counter() - t0
        return None, None, ("one step of B=%d clouds on %d threads did not finish within %.0f s (< %.0f points/s): oversubscription, see DESIGN.md 5"
                            % (cb, threads, el, cb * N / max(el - 5.0, 1.0)))
    except Exception as e:
        return None, None, "all-threads leg failed: %s: %s" % (type(e).__name__, e)


def _self_launch(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks (one per GPU) under torch.distributed.run on this node."""
    import socket
    import subprocess
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < n and not os.environ.get("P2C_ONE_GPU_RANKS"):
        sys.stderr.write("bench: --gpus %d requested but this node exposes %d GPU(s); not reporting a %d-GPU number from fewer devices\n" % (n, have, n))
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch_size", type=int, default=32, help="clouds per GPU")
    ap.add_argument("--num_point", type=int, default=8192)
    ap.add_argument("--K", type=int, default=8)
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--cpu_batch", type=int, default=4)
    ap.add_argument("--full_losses", action="store_true", help="configs[2]: + --pred_extrusion --pred_center")
    ap.add_argument("--torch_losses", action="store_true", help="evaluate the losses with torch ops instead of csrc/loss.hip")
    ap.add_argument("--no_prefetch", action="store_true", help="compute FPS/ball-query/3-NN inline instead of one step ahead on a side stream")
    ap.add_argument("--no_graph", action="store_true", help="launch every kernel from Python instead of replaying a HIP graph")
    ap.add_argument("--sync_exchange", action="store_true", help="(the default since round 6; kept so that older command lines still parse) N > 1: ONE "
                    "graph per step and the all-reduce issued on the step's own stream right behind the replay")
    ap.add_argument("--async_exchange", action="store_true", help="N > 1: the all-reduce on a SIDE stream between the two halves of a split-tail graph, the "
                    "next batch's geometry copies (0.04 ms) under it.  Measured on a one-rank nccl group (tools/xchg_ab.sh): 3.886 ms per step against "
                    "3.819 for the default and 3.79 without any exchange - the second replay and the event hops cost more than the tail can hide")
    ap.add_argument("--extras", type=str, default=DEFAULT_EXTRAS, help="comma-separated subset of the extra legs (%s) to run after the timed "
                    "region, or 'all'; default: %s" % (ALL_EXTRAS, DEFAULT_EXTRAS))
    ap.add_argument("--extras_file", type=str, default=None, help="where the detailed objects go (default gpurun_out/bench_extras.json)")
    ap.add_argument("--no_extras", action="store_true", help="only the training-step line: skip stages / forward_only / config3_fitting / ab / dropin")
    ap.add_argument("--dropin", action="store_true", help="make the DROP-IN step the timed one: the step composed as train_Point2Cyl_without_sketch.py:244-369 "
                    "composes it through the reference's import names (model(pcs), compute_all_losses, inline BB block, torch.optim.Adam, six .item())")
    args = ap.parse_args()

    from point2cyl_amd import ddp, ops, step, synth
    from point2cyl_amd.backbone import backbone
    import torch.distributed as dist

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(_self_launch(args.gpus))
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (the product has no CPU path)"
    _claim_stdout()
    rank, world, local = ddp.init_from_env()
    if world != args.gpus:
        raise SystemExit("bench: --gpus %d but %d rank(s) were launched (WORLD_SIZE); refusing to report a number for the wrong job size"
                         % (args.gpus, world))
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    from point2cyl_amd import hostmem
    hostmem.setup_cli()          # host allocator + torch's CPU pool fitted to the cgroup's CPU quota (the CPU legs set their own thread counts)
    # everything - warm-up, capture, replays, exchange, Adam, the event-timed eager leg - on ONE non-default stream (a HIP graph cannot be
    # captured on the default stream; autograd binds its accumulator nodes to the stream they first ran on)
    with torch.cuda.stream(torch.cuda.Stream(dev)):
        _bench(args, rank, world, local, dev)


def _bench(args, rank, world, local, dev):
    t_bench = time.perf_counter()
    from point2cyl_amd import ddp, ops, optim, step, synth
    from point2cyl_amd.backbone import backbone
    import torch.distributed as dist
    backend = ddp.backend_name()
    devices = [int(local)]
    if world > 1:
        assert dist.get_world_size() == world
        got = [None] * world
        dist.all_gather_object(got, (int(local), torch.cuda.get_device_properties(dev).name))
        devices = [g[0] for g in got]
        if len(set(devices)) != world and not os.environ.get("P2C_ONE_GPU_RANKS"):
            raise SystemExit("bench: %d ranks share GPUs %s; one process per GPU is required" % (world, devices))

    xchg = world > 1 or (ddp.force_exchange() and dist.is_initialized())      # P2C_FORCE_EXCHANGE=1: the real exchange on a one-rank nccl group
    B, N, K = args.batch_size, args.num_point, args.K
    fl = step.StepFlags(K=K, pred_extrusion=args.full_losses, pred_center=args.full_losses)
    pcs, normals, seg, bb, _, _, axes, _, centers = synth.make_batch(B, N, K, seed=1234 + 1000 * rank)
    batch = tuple(x.to(dev) for x in (pcs, normals, seg, bb, axes, centers))
    if args.dropin:
        if world != 1:
            raise SystemExit("bench --dropin: the reference's trainer is single-process; run it with --gpus 1")
        from point2cyl_amd import _lib
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = _dropin_leg(args, batch, fl, dev, B, N, K, native_ms=float("nan"), steps=args.steps)
        res.pop("ratio_to_native_step", None)
        emit(dict(metric="training-step points/sec (BxN) at N=8192", value=res["points_per_s"], unit="points/s", n_gpus=1, steps=args.steps,
                  warmup=3, ms_per_step=res["ms_per_step"], higher_is_better=True, scaling="weak", vs_baseline=None, dtype="f32", data="synthetic",
                  config=dict(workload="configs[%d] through the DROP-IN boundary: B=%d x N=%d, K=%d" % (2 if args.full_losses else 1, B, N, K),
                              mfma="bf16x3-split" if _lib.lib().p2c_get_mfma_mode() else "f32", launch="reference trainer composition on the drop-in import names"),
                  roofline=None, cpu_baseline=None, dropin=res), args.extras_file)
        return

    torch.manual_seed(0)
    model = backbone(output_sizes=fl.pred_sizes()).to(dev).train()
    ddp.broadcast_module(model)
    try:
        pre = ddp.preflight(model, dev)       # N > 1: one EAGER all-reduce + identical-replica proof before anything is captured
    except RuntimeError as e:
        sys.stderr.write("bench: %s\n" % e)
        raise SystemExit(3)
    step.update_momentum(model, step.get_batch_norm_decay(0, B, 200000))
    sync = ddp.FlatGradSync(model.parameters(), world)
    opt = optim.Adam(model.parameters(), lr=1e-3)     # the reference's torch.optim.Adam update rule as ONE launch (point2cyl_amd/optim.py)

    loss_fn = step.compute_losses_fused if (step.fused_loss_applicable(fl) and not args.torch_losses) else step.compute_losses

    def fwd_bwd(geom=None):
        ops.step_done()
        with ops.step_arena(dev):          # every zero-initialised accumulator of the step out of one buffer, one fill
            out = loss_fn(model, *batch, fl, geom=geom)
            sync.zero()
            step.backward(out)
            sync.pack()                    # N > 1: one multi-tensor copy into the exchange's flat buffer (a node of the captured graph)
        return {"total": out["total"].detach()}

    graphed = None
    if not args.no_graph:
        from point2cyl_amd.graph import GraphedForwardBackward
        try:
            graphed = GraphedForwardBackward(model, fwd_bwd, prefetch_xyz=None if args.no_prefetch else batch[0], stream=torch.cuda.current_stream(),
                                             split_tail=xchg and args.async_exchange)
        except Exception as e:      # keep the bench alive: fall back to eager launches
            sys.stderr.write("bench: HIP graph capture failed (%s: %s); running eager\n" % (type(e).__name__, e))
            torch.cuda.set_stream(torch.cuda.Stream(dev))      # a failed capture can leave its stream in capture mode: continue on a fresh one
            for m in model.modules():
                if hasattr(m, "fps_start"):
                    m.fps_start = None
            graphed = None

    ar_events = []          # N > 1: HIP events around the gradient exchange of every timed step (on the step's stream)

    def one_step(timed=False):
        out = graphed() if graphed is not None else fwd_bwd()
        if xchg and timed:         # events on the step's stream: from "gradients ready" to "exchange joined" (the graph's tail runs inside)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        if not args.async_exchange:
            sync.allreduce()            # on the step's stream right behind the one graph (the geometry copies are inside it)
        else:
            sync.allreduce_async()      # N > 1: on a side stream, gated on the replay; N = 1: nothing
            if graphed is not None:
                graphed.tail()          # N > 1: the prefetched geometry's copies, a second graph, under the exchange
            sync.wait()
        if xchg and timed:
            e1.record()
            ar_events.append((e0, e1))
        opt.step()
        ops.step_done()
        return out

    def fence():
        if xchg:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step()
    ops.PROFILE.reset(enabled=graphed is None)      # eager mode: HIP events around every launch of the timed region
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = one_step(timed=True)
    fence()
    dt = time.perf_counter() - t0
    ops.PROFILE.enabled = False
    prof_steps = args.steps
    if graphed is not None and rank == 0:
        # the timed region replayed a HIP graph (no per-launch host code to hang events on): time the SAME kernels
        # with HIP events on the launch stream in a few eager steps right after it
        prof_steps = 3
        ops.PROFILE.reset(enabled=True)
        for _ in range(prof_steps):
            graphed.starts.cursor = 0
            fwd_bwd()                      # (step_done inside: these gradients are not used)
        torch.cuda.synchronize()
        ops.PROFILE.enabled = False
    multi = None
    if xchg:
        # what a scaling line needs to be diagnosable: every rank's own time, the event-timed exchange, and proof that the replicas
        # hold the same parameters after the timed steps (sum of squares in float64, gathered)
        ar_ms = sum(a.elapsed_time(b) for a, b in ar_events) / max(1, len(ar_events))
        with torch.no_grad():
            ck = sum(float((p.detach().double() ** 2).sum()) for p in model.parameters())
        mine = torch.tensor([dt / args.steps * 1e3, ar_ms, ck], dtype=torch.float64, device=dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        allr = torch.stack(allr).cpu()
        multi = dict(rank_ms_per_step=[round(float(v), 4) for v in allr[:, 0]], allreduce_ms=[round(float(v), 4) for v in allr[:, 1]],
                     allreduce_bytes=int(sync.flat.numel() * 4) if sync.flat is not None else 0,
                     param_checksum=[float(v) for v in allr[:, 2]], params_identical=bool((allr[:, 2] == allr[0, 2]).all()),
                     recapture_count=1 if graphed is not None else 0, preflight=pre, exchanges=sync.exchanges, avg_op=bool(sync._avg_ok),
                     exchange="async (side stream under the split tail)" if args.async_exchange else "sync (step's stream, one graph)",
                     allreduce_note="allreduce_ms: from 'gradients ready' to 'exchange joined' on the step's stream; the exchange runs on a side stream and the "
                                    "step's tail (the copies of the next batch's prefetched geometry, a second graph, ~0.04 ms) runs under it")
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    loss = float(out["total"].detach())
    if rank != 0:
        if xchg:
            dist.destroy_process_group()
        return

    ms = dt / args.steps * 1e3
    value = world * B * N * args.steps / dt
    prof = ops.PROFILE.summary()           # per kernel family: launches, total ms (HIP events on the launch stream), flops, bytes
    # the same records split by the call's algorithmic size (bytes, flops): the members of a family at different shapes - the fused backward's
    # 128x128 layers next to the 256-wide two-pass call - each with its own rate (event time per call, this run)
    by_shape = {}
    for name, e0, e1, flops, nbytes in ops.PROFILE.records:
        if nbytes <= 0:
            continue
        r = by_shape.setdefault(name, {}).setdefault((round(nbytes), round(flops)), dict(calls=0, ms=0.0))
        r["calls"] += 1
        r["ms"] += e0.elapsed_time(e1)
    by_shape = {name: [dict(algorithmic_bytes_per_call=k[0], algorithmic_flops_per_call=k[1], calls_per_step=v["calls"] / prof_steps,
                            avg_call_us=round(v["ms"] * 1e3 / v["calls"], 2), hbm_gbs=round(k[0] / (v["ms"] / v["calls"] * 1e-3) / 1e9, 1),
                            hbm_frac=round(k[0] / (v["ms"] / v["calls"] * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
                            tflops_fp32_equivalent=round(k[1] / (v["ms"] / v["calls"] * 1e-3) / 1e12, 2))
                       for k, v in sorted(shapes.items(), key=lambda kv: -kv[1]["ms"])] for name, shapes in by_shape.items()}
    dom = max(prof.items(), key=lambda kv: kv[1]["ms"]) if prof else (None, None)
    roofline = None
    from point2cyl_amd import _lib
    split = bool(_lib.lib().p2c_get_mfma_mode())
    if dom[0] is not None:
        d = dom[1]
        if d["flops"] > 0 and not split:
            ach = d["flops"] / (d["ms"] * 1e-3) / 1e12
            tr = _pmc_traffic(dom[0])
            cps = d["launches"] / prof_steps
            roofline = dict(bound="mfma", kernel=dom[0], achieved=round(ach, 2), peak=PEAK_F32_MFMA_TFLOPS, unit="TFLOP/s",
                            frac=round(ach / PEAK_F32_MFMA_TFLOPS, 4), traffic=None if tr is None else round(tr["per_step"] / cps),
                            traffic_unit="HBM bytes (read + write, PMC) per call of the entry point",
                            traffic_source=None if tr is None else tr["source"], traffic_source_hash=None if tr is None else tr["source_hash"],
                            traffic_stale=None if tr is None else tr["stale"], algorithmic_bytes_per_launch=round(d["bytes"] / max(1, d["launches"])),
                            launches_per_step=d["launches"] / prof_steps, avg_launch_us=round(d["ms"] * 1e3 / d["launches"], 2),
                            share_of_step=round(d["ms"] / prof_steps / ms, 3))
        elif d["flops"] > 0:
            # bf16x3-split: every fp32 product is six bf16 MFMA products, so the matrix pipe's ceiling for this arithmetic is
            # 2500 / 6 = 417 fp32-equivalent TFLOP/s; the kernel then sits nearer its HBM bound than its matrix-pipe bound, and the line
            # reports the tighter of the two as `bound` with the other beside it (and the fraction of the fp32-MFMA peak the
            # round-2 line was priced against, for continuity).
            # UNITS: a "launch" of this object is one CALL of the entry point = one layer's backward (5 per step); the 256-wide
            # layer is one call that runs two kernel launches (6 kernel launches per step).  `achieved` = algorithmic bytes per call /
            # average call duration (HIP events on the launch stream, this run); `traffic` = PMC bytes per call (committed counters:
            # sum over the family's kernels of launches x bytes, / calls); `traffic_ratio` = PMC / algorithmic, both per step.
            calls = d["launches"]
            calls_per_step = calls / prof_steps
            tf = d["flops"] / (d["ms"] * 1e-3) / 1e12
            gbs = d["bytes"] / (d["ms"] * 1e-3) / 1e9
            alg_step = d["bytes"] / prof_steps
            tr = _pmc_traffic(dom[0])
            f_hbm, f_pipe = gbs / PEAK_HBM_GBS, 6.0 * tf / PEAK_BF16_MFMA_TFLOPS
            common = dict(launch_unit="one call of the entry point (a layer's backward); kernel launches are listed in traffic_by_kernel",
                          traffic=None if tr is None else round(tr["per_step"] / calls_per_step), traffic_unit="HBM bytes (read + write, PMC) per call",
                          traffic_per_step=None if tr is None else tr["per_step"], algorithmic_bytes_per_step=round(alg_step),
                          traffic_ratio=None if (tr is None or alg_step <= 0) else round(tr["per_step"] / alg_step, 3),
                          kernel_launches_per_step=None if tr is None else tr["kernel_launches_per_step"],
                          traffic_by_kernel=None if tr is None else tr["by_kernel"],
                          traffic_source=None if tr is None else tr["source"], traffic_source_hash=None if tr is None else tr["source_hash"],
                          traffic_stale=None if tr is None else tr["stale"],
                          algorithmic_bytes_per_launch=round(d["bytes"] / max(1, calls)),
                          algorithmic_flops_per_launch=round(d["flops"] / max(1, calls)),
                          launches_per_step=calls_per_step, avg_launch_us=round(d["ms"] * 1e3 / calls, 2),
                          ms_per_step=round(d["ms"] / prof_steps, 4), share_of_step=round(d["ms"] / prof_steps / ms, 3),
                          hbm=dict(achieved=round(gbs, 1), peak=PEAK_HBM_GBS, unit="GB/s", frac=round(f_hbm, 4)),
                          mfma=dict(achieved_fp32_equivalent=round(tf, 2), issued_bf16=round(6.0 * tf, 1), peak_bf16=PEAK_BF16_MFMA_TFLOPS, unit="TFLOP/s",
                                    frac=round(f_pipe, 4), frac_of_fp32_mfma_peak=round(tf / PEAK_F32_MFMA_TFLOPS, 4)),
                          by_shape=by_shape.get(dom[0]))
            if f_hbm >= f_pipe:
                roofline = dict(bound="hbm", kernel=dom[0], achieved=round(gbs, 1), peak=PEAK_HBM_GBS, unit="GB/s", frac=round(f_hbm, 4), **common)
            else:
                roofline = dict(bound="mfma", kernel=dom[0], achieved=round(6.0 * tf, 1), peak=PEAK_BF16_MFMA_TFLOPS, unit="TFLOP/s", frac=round(f_pipe, 4), **common)
        else:
            ach = d["bytes"] / (d["ms"] * 1e-3) / 1e9
            roofline = dict(bound="hbm", kernel=dom[0], achieved=round(ach, 1), peak=PEAK_HBM_GBS, unit="GB/s",
                            frac=round(ach / PEAK_HBM_GBS, 4), traffic=None,
                            launches_per_step=d["launches"] / prof_steps, avg_launch_us=round(d["ms"] * 1e3 / d["launches"], 2),
                            share_of_step=round(d["ms"] / prof_steps / ms, 3))
    cpu = None
    want = set(ALL_EXTRAS.split(",")) if args.extras == "all" else set(x for x in args.extras.split(",") if x)
    if args.no_extras:
        want = set()
    if world == 1 and not args.no_cpu_baseline:
        from oracle import ref_step
        cb = args.cpu_batch
        sample = tuple(x[:cb].contiguous() for x in (pcs, normals, seg, bb))
        host = os.cpu_count() or 1
        from point2cyl_amd.hostmem import cpu_quota
        quota = cpu_quota()       # what the cgroup lets this process use (16 of the 256 hardware threads on this pool's boxes): the leg's thread count
        pps, sec, thr, nst = ref_step.time_cpu_baseline(sample, threads=min(32, host, quota), budget_s=10.0)
        by_threads = {str(thr): round(pps, 1)}
        best = (pps, thr)
        pps1 = ppsa = all_note = None
        if "cpu_threads" in want:
            # SURVEY 8(d)'s other two thread counts, behind --extras cpu_threads (20 s more): 1 thread on one cloud, and os.cpu_count() threads
            # in a child process with a time limit (one thread per hardware thread under a CPU quota of a sixteenth of them never finishes
            # in seconds: measured once without a limit, 183.6 s for ONE step = 178 points/s, profiles/r04_bench_all_threads_unbounded.json.log)
            one = tuple(x[:1].contiguous() for x in (pcs, normals, seg, bb))
            pps1, sec1, _, nst1 = ref_step.time_cpu_baseline(one, threads=1, budget_s=8.0)
            by_threads["1"] = round(pps1, 1)
            if host > 32:
                ppsa, seca, all_note = _cpu_all_threads_leg(cb, N, K, host, limit_s=12.0)
            else:
                ppsa = pps
            by_threads[str(host)] = None if ppsa is None else round(ppsa, 1)
            torch.set_num_threads(min(32, host, quota))
            best = max(((pps, thr), (ppsa or 0.0, host), (pps1, 1)), key=lambda t: t[0])
        try:
            affinity = len(os.sched_getaffinity(0))
        except (AttributeError, OSError):
            affinity = None
        cpu = dict(value=round(best[0], 1), unit="points/s", cores=best[1], kind="port", cpu_model=_cpu_model(), host_cores=host, sched_affinity=affinity,
                   cpu_quota=quota, by_threads=by_threads, single_thread_value=None if pps1 is None else round(pps1, 1),
                   all_threads_value=None if ppsa is None else round(ppsa, 1), all_threads_note=all_note,
                   sample="%d training steps (fwd+losses+bwd+Adam) of oracle/ref_step.py on B=%d clouds x %d points, %d threads, %.1f s"
                          % (nst, cb, N, thr, sec * nst))
    full = dict(metric="training-step points/sec (BxN) at N=8192", value=round(value, 1), unit="points/s", n_gpus=world,
                steps=args.steps, warmup=args.warmup, ms_per_step=round(ms, 3), higher_is_better=True, scaling="weak",
                vs_baseline=None, dtype="f32", data="synthetic",
                backend=backend, world_size=(dist.get_world_size() if xchg else 1), devices=devices,
                config=dict(workload="configs[%d]: B=%d clouds/GPU x N=%d points, K=%d, %s, random-init backbone, synthetic "
                                     "extrusion-cylinder clouds; step = fwd + losses + bwd + Adam" %
                                     (2 if args.full_losses else 1, B, N, K,
                                      "full loss set" if args.full_losses else "pred_seg+pred_normal+pred_bb"),
                            mfma="bf16x3-split" if split else "f32", batch_per_gpu=B, global_batch=B * world, num_point=N, parallelism="dp%d" % world, loss=round(loss, 5),
                            launch=("hip_graph(fwd+bwd%s)+eager(allreduce,adam)" % ("" if args.no_prefetch else ", next batch's FPS/ball-query/3-NN on a forked stream")) if graphed is not None else "eager"),
                roofline=roofline, cpu_baseline=cpu, multi_gpu=multi,
                kernels={k: dict(ms_per_step=round(v["ms"] / prof_steps, 3), launches_per_step=v["launches"] / prof_steps)
                         for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])})
    if world == 1 and want:
        if graphed is not None:
            graphed.release()
        full.update(_extras(args, want, model, batch, fl, dev, ms, B, N, K, loss_fn, sync, opt))
    full["bench_seconds"] = round(time.perf_counter() - t_bench, 1)
    emit(full, args.extras_file)
    if xchg:
        dist.destroy_process_group()


_LINE_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
_ROOF_KEYS = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_step", "traffic_ratio", "traffic_stale",
              "avg_launch_us", "launches_per_step", "share_of_step")
_CPU_KEYS = ("value", "unit", "cores", "kind", "cpu_model")


def _dig(d, *path):
    for k in path:
        if not isinstance(d, dict) or k not in d:
            return None
        d = d[k]
    return d


_REAL_STDOUT = None      # a private duplicate of the process's stdout once _claim_stdout() has run


def _claim_stdout():
    """From here on file descriptor 1 of this process IS stderr; the record goes out through a private duplicate of the original stdout.
    Why: libraries write to stdout behind Python's back - librccl prints a five-line version banner through C stdio, which is flushed at
    process EXIT, i.e. after the JSON line (seen with a one-rank nccl group; under torchrun every rank's banner lands in the launcher's
    stdout whenever that rank exits).  The contract is ONE JSON line, last, on stdout: nothing but emit() may write there."""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)


def compact_line(full):
    """The record the driver parses: the contract's keys, roofline, cpu_baseline and a handful of scalars; numbers and short names only
    (no prose), < LINE_LIMIT bytes.  Everything else of `full` goes to stderr / the extras file (emit)."""
    cfg = full.get("config") or {}
    line = {k: full.get(k) for k in _LINE_KEYS}
    line["config"] = {k: cfg.get(k) for k in ("workload", "mfma", "batch_per_gpu", "global_batch", "num_point", "parallelism", "loss") if k in cfg}
    if isinstance(line["config"].get("workload"), str):
        line["config"]["workload"] = line["config"]["workload"][:160]
    line["config"]["graph"] = str(cfg.get("launch", "")).startswith("hip_graph")
    roof = full.get("roofline")
    line["roofline"] = None if not roof else {k: roof.get(k) for k in _ROOF_KEYS if k in roof or k in ("bound", "achieved", "peak", "unit", "frac", "traffic")}
    cpu = full.get("cpu_baseline")
    line["cpu_baseline"] = None if not cpu else dict({k: cpu.get(k) for k in _CPU_KEYS}, sample=str(cpu.get("sample", ""))[:140])
    m = full.get("multi_gpu")
    if m:
        line["multi_gpu"] = {k: m.get(k) for k in ("rank_ms_per_step", "allreduce_ms", "allreduce_bytes", "params_identical") if k in m}
        line["multi_gpu"]["exchange"] = str(m.get("exchange", ""))[:5].strip()
        line["backend"] = full.get("backend")
    scal = dict(f32_mfma_ms_per_step=_dig(full, "ab", "f32_mfma_ms_per_step"),
                split_ms_per_step_ab=_dig(full, "ab", "bf16x3_split_ms_per_step"),
                sa1_stage_frac_best=_dig(full, "stages", "sa1_forward", "best", "frac_of_mfma_roofline"),
                sa1_stage_frac_serial=_dig(full, "stages", "sa1_forward", "frac_of_mfma_roofline"),
                fps_ms=_dig(full, "stages", "sa1_forward", "parts_ms", "fps"),
                config3_ms=_dig(full, "config3_fitting", "ms"),
                config3_frac_hbm_path_bytes=_dig(full, "config3_fitting", "frac_hbm_path_bytes"),
                eval_ms_per_batch=_dig(full, "eval_loop", "pipelined", "ms_per_batch_after_first"),
                dropin_ms_per_step=_dig(full, "dropin", "ms_per_step"),
                dropin_max_over_median=_dig(full, "dropin", "max_over_median"),
                rccl_allreduce_us=_dig(full, "rccl_selftest", "allreduce_us"),
                bench_seconds=full.get("bench_seconds"))
    line.update({k: v for k, v in scal.items() if v is not None})
    top = list((full.get("kernels") or {}).items())[:6]
    line["top_kernels_ms"] = {k.replace("p2c_", "").replace("_f32", ""): v.get("ms_per_step") for k, v in top}
    line["extras_file"] = full.get("extras_file")
    while len(json.dumps(line)) >= LINE_LIMIT and line.get("top_kernels_ms"):      # (cannot happen with the keys above; belt and braces)
        line["top_kernels_ms"].popitem()
    return line


def emit(full, extras_file=None):
    """Detailed objects -> stderr (one "extra" line per key) and the extras file; then the compact line, LAST, alone on stdout."""
    path = extras_file or os.path.join(ROOT, "gpurun_out", "bench_extras.json")
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            json.dump(full, f, indent=1)
        full["extras_file"] = os.path.relpath(path, ROOT)
    except OSError:
        full["extras_file"] = None
    basic = set(_LINE_KEYS) | {"config", "extras_file"}
    for k, v in full.items():
        if k not in basic and v is not None:
            sys.stderr.write("extra " + json.dumps({k: v}) + "\n")
    sys.stderr.flush()
    line = compact_line(full)
    text = json.dumps(line)
    assert len(text) < LINE_LIMIT, len(text)
    if _REAL_STDOUT is not None:
        os.write(_REAL_STDOUT, (text + "\n").encode())
    else:
        sys.stdout.write(text + "\n")
        sys.stdout.flush()
    return line


def _extras(args, want, model, batch, fl, dev, ms, B, N, K, loss_fn, sync, opt):
    """The measurements SURVEY 8(d) asks for beside the training-step number, in the same process after the timed region (rank 0, N = 1):
    stages.sa1_forward, forward_only, path_roofline, config3_fitting, ab (fp32-MFMA kernels), dropin (the step through the import names).
    Each leg is guarded: a failure is reported in the line instead of taking the training-step number down with it."""
    from point2cyl_amd import _lib, measure, ops, step
    out = {}
    leg_s = out.setdefault("leg_seconds", {})

    def leg(name, fn):
        if name not in want:
            return
        t0 = time.perf_counter()
        try:
            out[name] = fn()
        except Exception as e:
            out[name] = dict(error="%s: %s" % (type(e).__name__, e))
        leg_s[name] = round(time.perf_counter() - t0, 1)
        try:
            torch.cuda.synchronize()
            ops.step_done()
        except Exception:
            pass

    leg("path_roofline", lambda: measure.path_roofline(B, ms, 3.0, N))
    leg("power_cap", lambda: measure.power_cap_probe(dev))
    leg("stages", lambda: dict(sa1_forward=measure.sa1_stage(model, batch[0], steps=30)))
    leg("forward_only", lambda: measure.forward_only(model, batch[0], steps=30))

    def ab():
        from point2cyl_amd.graph import GraphedForwardBackward
        cur = _lib.lib().p2c_get_mfma_mode()
        res = {}
        for mode, key in ((0, "f32_mfma_ms_per_step"), (1, "bf16x3_split_ms_per_step")):
            _lib.lib().p2c_set_mfma_mode(mode)
            try:
                def fwd_bwd(geom=None):
                    ops.step_done()
                    with ops.step_arena(dev):
                        o = loss_fn(model, *batch, fl, geom=geom)
                        sync.zero()
                        step.backward(o)
                        sync.pack()
                    return {"total": o["total"].detach()}
                g = GraphedForwardBackward(model, fwd_bwd, prefetch_xyz=None if args.no_prefetch else batch[0], stream=torch.cuda.current_stream())
                try:
                    def one():
                        g()
                        opt.step()
                        ops.step_done()
                    for _ in range(3):
                        one()
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(15):
                        one()
                    torch.cuda.synchronize()
                    res[key] = round((time.perf_counter() - t0) / 15 * 1e3, 4)
                finally:
                    g.release()
            finally:
                _lib.lib().p2c_set_mfma_mode(cur)
        res["note"] = "the SAME step, graph re-captured with p2c_set_mfma_mode(0) (v_mfma_f32_32x32x2_f32 kernels) and (1) (bf16x3 split), 15 replays each, alternating on this box"
        return res

    leg("ab", ab)

    def config3():
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_config4
        from point2cyl_amd.hostmem import cpu_quota
        w = measure.FittingWorkload(1250, 8192, 8, 2048, device=dev)
        res, (E, cen, cfound, ext, found, E64) = w.time(20)
        if "config3_cpu" in want:       # the oracle's fitting on the host cores beside it + parity of all 1250 clouds (tests/test_gpu_configs.py holds the same)
            cpu, parity = bench_config4.cpu_fitting_legs(w, E, cen, ext, E64, thread_counts=(min(32, os.cpu_count() or 1, cpu_quota()), 1))
            res.update(cpu_baseline=cpu, parity=parity)
        del w
        return res

    if not args.full_losses:
        leg("config3_fitting", config3)

    def dropin():
        return _dropin_leg(args, batch, fl, dev, B, N, K, ms)

    leg("dropin", dropin)

    def eval_loop():
        # The evaluation script (point2cyl_amd/eval.py = eval.py:231-457, :690-715) on 1024 synthetic clouds in batches of B, randomly
        # initialised backbone, in a child process (it owns a loader thread and process-wide host settings): forward + every metric + report.
        import subprocess, tempfile
        res = {}
        with tempfile.TemporaryDirectory() as d:
            for key, extra in (("pipelined", []), ("serial_reference_order", ["--no_prefetch"]))[:2 if "eval_serial" in want else 1]:
                rep = os.path.join(d, key + ".json")
                cmd = [sys.executable, "-m", "point2cyl_amd.eval", "--random_init", "--synthetic", str(32 * B), "--batch_size", str(B), "--num_point", str(N),
                       "--K", str(K), "--dump_dir", d, "--report", rep] + extra
                env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_PORT")}
                r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=240, env=env)      # (a one-process job of its own)
                if r.returncode != 0 or not os.path.exists(rep):
                    res[key] = dict(error=r.stderr[-400:])
                    continue
                j = json.load(open(rep))
                res[key] = {k: (round(v, 4) if isinstance(v, float) else v) for k, v in j.items()}
        res["what"] = ("python -m point2cyl_amd.eval --random_init --synthetic %d --batch_size %d: per batch the forward (geometry of --prefetch_group batches "
                       "computed together one group ahead), every metric of eval.py:270-457 as one HIP-graph replay, the report's sums; "
                       "serial_reference_order = --no_prefetch (no loader thread, nothing read ahead, every random draw in the reference's order)" % (32 * B, B))
        return res

    leg("eval_loop", eval_loop)

    def rccl_selftest():
        # SURVEY 8(e) on the one GPU this run has: a one-rank nccl group through the real FlatGradSync path (point2cyl_amd/ddp_selftest.py), in a
        # child process (the process group and P2C_FORCE_EXCHANGE are process-wide)
        import subprocess
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        for attempt in (1, 2):           # (a process-group / HIP-runtime thread has aborted the child once in ~20 runs: one more try, and say so)
            r = subprocess.run([sys.executable, "-m", "point2cyl_amd.ddp_selftest", "--steps", "20", "--batch_size", str(B), "--num_point", str(N)],
                               cwd=ROOT, capture_output=True, text=True, timeout=300, env=env)
            try:
                res = json.loads(r.stdout.strip().splitlines()[-1])
                res["attempts"] = attempt
                return res
            except (ValueError, IndexError):
                err = dict(error="rc %d: %s" % (r.returncode, r.stderr[-4000:]), attempts=attempt)
        return err

    leg("rccl_selftest", rccl_selftest)
    return out


def _dropin_leg(args, batch, fl, dev, B, N, K, native_ms, steps=20):
    """The step a user of the BOUNDARY gets: point2cyl_amd/dropin/trainer_step.py composes train_Point2Cyl_without_sketch.py:244-369 through
    the reference's import names with torch.optim.Adam and six .item() reads.  Timed with the HIP graphs inside backbone.forward
    (point2cyl_amd/autograph.py) and, for the A/B, with every kernel launched from Python."""
    from point2cyl_amd import autograph
    from point2cyl_amd.dropin.trainer_step import TrainerStep
    res = {}
    for label, on, n in (("ms_per_step", True, max(steps, 120)), ("eager_ms_per_step", False, max(9, steps // 4))):
        old = autograph.ENABLED
        autograph.ENABLED = on
        try:
            torch.manual_seed(0)
            st = TrainerStep(K=K, batch_size=B, pred_extrusion=fl.pred_extrusion, pred_center=fl.pred_center, device=dev)
            for _ in range(3):
                logs = st(*batch)
            torch.cuda.synchronize()
            per = []
            t0 = time.perf_counter()
            for _ in range(n):
                t1 = time.perf_counter()
                logs = st(*batch)                  # (ends with the trainer's six .item() reads: the step is synchronised by itself)
                per.append(time.perf_counter() - t1)
            torch.cuda.synchronize()
            mean_ms = (time.perf_counter() - t0) / n * 1e3
            slowest = sorted(((round(v * 1e3, 3), i) for i, v in enumerate(per)), reverse=True)[:3]
            per.sort()
            # the step is host-paced (a Python loss composition + six device->host reads): on a shared host single steps stretch by
            # milliseconds (measured: the mean of 20 steps between 5.9 and 8.3 ms on one box within a minute); the MEDIAN step is the
            # reproducible figure, the mean is reported beside it
            res[label] = round(per[len(per) // 2] * 1e3, 4)
            res[label.replace("ms_per_step", "mean_ms_per_step")] = round(mean_ms, 4)
            if on:
                res.update(steps_timed=n, p50_ms=round(per[len(per) // 2] * 1e3, 4), p99_ms=round(per[min(n - 1, int(n * 0.99))] * 1e3, 4),
                           max_ms=round(per[-1] * 1e3, 4), max_over_median=round(per[-1] / per[len(per) // 2], 2),
                           slowest_steps_ms_index=slowest)
                res["loss_after_%d_steps" % (n + 3)] = round(logs[0], 5)
            autograph.reset(st.model)
            del st
        finally:
            autograph.ENABLED = old
    res["points_per_s"] = round(B * N / (res["ms_per_step"] * 1e-3), 1)
    res["ratio_to_native_step"] = round(res["ms_per_step"] / native_ms, 3)
    res["what"] = ("the reference trainer's own step composition (train_Point2Cyl_without_sketch.py:244-369) on the drop-in import names: model(pcs) "
                   "[HIP graphs cached inside backbone.forward / its backward], F.normalize / softmax in torch, losses.compute_all_losses, the inline "
                   "base/barrel block as torch ops, torch.optim.Adam, six .item() reads per step; FPS is on the critical path here (no next-batch prefetch)")
    return res


if __name__ == "__main__":
    main()          # (a normal interpreter exit: rocprofv3 and other tools write their output from exit handlers)
