set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
TAG=r05; OUT=gpurun_out/$TAG; mkdir -p $OUT
python -m point2cyl_amd.train --pred_seg --pred_normal --pred_bb --synthetic 256 --batch_size 32 --num_epochs 2 --quiet --logdir /tmp/${TAG}_tr > /dev/null 2>&1
python -m point2cyl_amd.eval --synthetic 4096 --batch_size 32 --logdir /tmp/${TAG}_tr --ckpt model.pth --dump_dir /tmp/${TAG}_ev --report "$OUT/${TAG}_eval_report_pipelined.json" > "$OUT/${TAG}_eval_synthetic.log" 2> "$OUT/eval.err"
python -m point2cyl_amd.eval --synthetic 4096 --batch_size 32 --logdir /tmp/${TAG}_tr --ckpt model.pth --dump_dir /tmp/${TAG}_ev --no_prefetch --report "$OUT/${TAG}_eval_report_serial.json" > "$OUT/${TAG}_eval_synthetic_no_prefetch.log" 2>> "$OUT/eval.err"
EV="python -m point2cyl_amd.eval --synthetic 4096 --batch_size 32 --logdir /tmp/${TAG}_tr --ckpt model.pth --dump_dir /tmp/${TAG}_ev"
for f in "--prefetch_group 1" "--prefetch_group 1 --no_graph_metrics" "--prefetch_group 8" "--prefetch_group 4 --add_noise"; do echo "$f: $($EV $f 2>> "$OUT/eval.err" | grep throughput)"; done > "$OUT/${TAG}_eval_variants.log"
python tools/probe/forward_modes.py 2>/dev/null | grep "mode:" > "$OUT/${TAG}_forward_modes.log"
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof1 -o ev -- python -m point2cyl_amd.eval --synthetic 2048 --batch_size 32 --logdir /tmp/${TAG}_tr --ckpt model.pth --dump_dir /tmp/${TAG}_ev 2>&1 | grep throughput > "$OUT/${TAG}_eval_cli_under_rocprof.log"
f=$(find /tmp/prof1 -name "*kernel_stats.csv" | head -1); if [ -n "$f" ]; then cp "$f" "$OUT/${TAG}_eval_cli_rocprofv3_kernel_stats.csv"; fi
python bench.py > "$OUT/${TAG}_bench.json.log" 2> "$OUT/bench.err"
grep throughput "$OUT/${TAG}_eval_synthetic.log" "$OUT/${TAG}_eval_synthetic_no_prefetch.log"; cat "$OUT/${TAG}_eval_variants.log" "$OUT/${TAG}_forward_modes.log" "$OUT/${TAG}_eval_cli_under_rocprof.log"
