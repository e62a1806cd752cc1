cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for f in "" "--no_prefetch"; do
rm -rf gpurun_out/gp; mkdir -p gpurun_out/gp
rocprofv3 --kernel-trace -d gpurun_out/gp -o r --output-format csv -- python bench.py --steps 30 --warmup 3 --no_cpu_baseline --no_extras $f > gpurun_out/gp/bench.log 2> gpurun_out/gp/err.log
echo "bench.py $f: $(tail -1 gpurun_out/gp/bench.log | python -c 'import sys,json; print(json.loads(sys.stdin.read())["ms_per_step"])') ms/step under the tracer"
python tools/gap_probe.py gpurun_out/gp/r_kernel_trace.csv
done
rm -rf gpurun_out/gp
