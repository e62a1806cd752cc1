#!/bin/bash
# All measured artefacts of a round from ONE state of the tree, on the GPU box:   bash tools/round_profiles.sh r02
# Everything lands in gpurun_out/<tag>/ (the only directory gpurun merges back); copy what is to be judged into profiles/.
set -u
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
# 2. kernel trace + stats of the same command (no counters in this pass)
rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o r --output-format csv -- python bench.py --steps 10 --warmup 3 --no_cpu_baseline --no_extras > "$OUT/${TAG}_bench_under_rocprof.json.log" 2> "$OUT/trace.err"
cp "$OUT/trace/r_kernel_stats.csv" "$OUT/${TAG}_rocprofv3_kernel_stats.csv"
python tools/timeline.py "$OUT/trace/r_kernel_trace.csv" 7 > "$OUT/${TAG}_step_timeline.md" 2>> "$OUT/trace.err"
python tools/timeline.py "$OUT/trace/r_kernel_trace.csv" 7 --all > "$OUT/${TAG}_step_timeline_every_launch.md" 2>> "$OUT/trace.err"
python - "$OUT" "$TAG" <<'EOF'
import csv, json, re, sys
out, tag = sys.argv[1:3]
short = lambda n: re.sub(r"\(.*", "", n).replace("void ", "").replace("at::native::", "").replace("(anonymous namespace)::", "")
b = json.loads([l for l in open("%s/%s_bench_under_rocprof.json.log" % (out, tag)) if l.startswith("{")][-1])
steps = b["steps"] + b["warmup"] + 2 + 3
rows = list(csv.DictReader(open("%s/trace/r_kernel_stats.csv" % out)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
with open("%s/%s_kernel_stats.md" % (out, tag), "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats of `python bench.py --steps %d --warmup %d --no_cpu_baseline --no_extras` (configs[1], 1 GPU)\n\n" % (b["steps"], b["warmup"]))
    f.write("bench line of this run: %.1f points/s, %.3f ms/step.  The trace holds %d passes over the step (timed + warm-up + 2 capture warm-ups + the 3 "
            "eager steps bench.py uses for its HIP-event timing); 'us/step' = total / %d.  Kernels of the geometry prefetch (fps, ball_query, three_nn, "
            "build_csr, group_gather_xyz) run on the forked stream, concurrently with the rest.\n\n" % (b["value"], b["ms_per_step"], steps, steps))
    f.write("| kernel | calls | avg us | us/step | % of kernel time |\n|---|---:|---:|---:|---:|\n")
    for r in rows[:50]:
        f.write("| `%s` | %s | %.1f | %.1f | %.2f |\n" % (short(r["Name"])[:100], r["Calls"], float(r["AverageNs"]) / 1e3,
                                                        float(r["TotalDurationNs"]) / 1e3 / steps, float(r["Percentage"])))
    f.write("\ntotal kernel time %.3f ms over %d kernel names\n" % (tot / 1e6, len(rows)))
EOF
rm -rf "$OUT/trace"
if [ "${QUICK:-0}" = "1" ]; then ls -la "$OUT"; exit 0; fi
# 3. counters: HBM traffic and SQ counters of every kernel of the step (one group per pass)
python tools/collect_pmc.py "$TAG" > "$OUT/pmc.log" 2>&1
cp profiles/${TAG}_pmc_step.json profiles/${TAG}_pmc_step.md "$OUT/" 2>/dev/null
rm -rf gpurun_out/${TAG}_pmc_*
# 3b. the bench line exactly as the driver runs it (with the CPU baseline), and configs[2]'s loss set - AFTER the counters, so that
# roofline.traffic is read from the summary collected on this very tree (traffic_stale: false)
python bench.py > "$OUT/${TAG}_bench.json.log" 2> "$OUT/bench.err"
python bench.py --full_losses --no_cpu_baseline --no_extras > "$OUT/${TAG}_bench_config2_full_losses.json.log" 2>> "$OUT/bench.err"
python bench.py --dropin --steps 40 > "$OUT/${TAG}_bench_dropin.json.log" 2>> "$OUT/bench.err"
# 4. the other configs / stages
python tools/bench_config4.py > "$OUT/${TAG}_config4_fitting.json.log" 2> "$OUT/config4.err"
python tools/bench_sa1_forward.py > "$OUT/${TAG}_sa1_forward_stage.json.log" 2> "$OUT/sa1.err"
python tools/bench_config5.py --steps 10 > "$OUT/${TAG}_config5_with_sketch_step.json.log" 2> "$OUT/config5.err"
python tools/bench_config5.py --steps 5 --no_graph > "$OUT/${TAG}_config5_with_sketch_step_eager.json.log" 2>> "$OUT/config5.err"
python tools/bench_pool_alg.py --trace > "$OUT/${TAG}_pool_alg_backward.log" 2> "$OUT/pool.err"
[ -x tools/ubench/grid_barrier.bin ] || /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o tools/ubench/grid_barrier.bin tools/ubench/grid_barrier.hip > /dev/null 2>&1
timeout 100 tools/ubench/grid_barrier.bin > "$OUT/${TAG}_grid_barrier_ubench.log" 2>&1
# 5. the trainers and the evaluation script (throughput through the CLI, convergence log)
python -m point2cyl_amd.train --pred_seg --pred_normal --pred_bb --synthetic 1024 --batch_size 32 --num_epochs 60 --quiet --logdir /tmp/${TAG}_tr \
    --report "$OUT/${TAG}_trainer_report.json" > "$OUT/${TAG}_train_convergence_synthetic.log" 2> "$OUT/train.err"
python -m point2cyl_amd.train --pred_seg --pred_normal --pred_bb --synthetic 256 --batch_size 32 --num_epochs 2 --logdir /tmp/${TAG}_tr2 \
    --report "$OUT/${TAG}_trainer_report_per_step_log.json" > /dev/null 2>> "$OUT/train.err"
python -m point2cyl_amd.eval --synthetic 4096 --batch_size 32 --logdir /tmp/${TAG}_tr --ckpt model.pth --dump_dir /tmp/${TAG}_ev --report "$OUT/${TAG}_eval_report_pipelined.json" > "$OUT/${TAG}_eval_synthetic.log" 2> "$OUT/eval.err"
python -m point2cyl_amd.eval --synthetic 4096 --batch_size 32 --logdir /tmp/${TAG}_tr --ckpt model.pth --dump_dir /tmp/${TAG}_ev --no_prefetch --report "$OUT/${TAG}_eval_report_serial.json" > "$OUT/${TAG}_eval_synthetic_no_prefetch.log" 2>> "$OUT/eval.err"
EV="python -m point2cyl_amd.eval --synthetic 4096 --batch_size 32 --logdir /tmp/${TAG}_tr --ckpt model.pth --dump_dir /tmp/${TAG}_ev"
for f in "--prefetch_group 1" "--prefetch_group 1 --no_graph_metrics" "--prefetch_group 8" "--prefetch_group 4 --add_noise" "--prefetch_group 4 --no_fused_metrics"; do echo "$f: $($EV $f 2>> "$OUT/eval.err" | grep throughput)"; done > "$OUT/${TAG}_eval_variants.log"
python tools/probe/eval_metrics_time.py > "$OUT/${TAG}_eval_metrics_probe.log" 2>> "$OUT/eval.err"
python tools/probe/stage_depth2.py > "$OUT/${TAG}_stage_depth.log" 2>> "$OUT/eval.err"
python tools/probe/forward_modes.py 2>> "$OUT/eval.err" | grep "mode:" > "$OUT/${TAG}_forward_modes.log"
P2C_BN_EVAL_BATCH=0 python tools/probe/forward_modes.py 2>> "$OUT/eval.err" | grep "eval mode:" | sed "s/^/P2C_BN_EVAL_BATCH=0 (one finalize launch per layer): /" >> "$OUT/${TAG}_forward_modes.log"
if [ -d .ab_base ]; then bash tools/ab_commits.sh 3 > "$OUT/${TAG}_ab_vs_round4_tree.log" 2>&1; fi
bash tools/fit_ab.sh > "$OUT/${TAG}_fit_fused_phase_trace.log" 2>&1
python tools/bench_config5.py --steps 3 --glue > /dev/null 2> "$OUT/${TAG}_config5_torch_side_ops.log"
python -m point2cyl_amd.train_sketch --pred_seg --pred_normal --pred_bb --is_pc_train --is_im_train --with_im_loss --synthetic 64 --batch_size 16 \
    --num_epochs 2 --logdir /tmp/${TAG}_sk --im_logdir /tmp/none --report "$OUT/${TAG}_sketch_trainer_report.json" > "$OUT/${TAG}_train_sketch_synthetic.log" 2> "$OUT/sk.err"
tail -c 300 "$OUT"/*.err | tail -40
ls -la "$OUT"
