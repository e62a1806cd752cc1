cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/dropin
python -m pytest tests/test_gpu_flows.py -q -x -k "autograph or dropin" 2>&1 | tail -5
rocprofv3 --kernel-trace --stats -d gpurun_out/dropin/trace -o r --output-format csv -- python bench.py --dropin --steps 20 > gpurun_out/dropin/bench.json 2> gpurun_out/dropin/err.log
python - <<'PY'
import csv,re
rows=list(csv.DictReader(open('gpurun_out/dropin/trace/r_kernel_stats.csv')))
short=lambda n: re.sub(r"\(.*","",n).replace("void ","").replace("at::native::","")[:90]
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", tot/1e6)
for r in rows[:60]:
    print("%-92s %6s %9.1f %9.1f" % (short(r["Name"]), r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e3))
PY
cp gpurun_out/dropin/trace/r_kernel_trace.csv gpurun_out/dropin/kernel_trace.csv
rm -rf gpurun_out/dropin/trace
cat gpurun_out/dropin/bench.json | head -c 1500
