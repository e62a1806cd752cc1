#!/bin/bash
# per-kernel average durations of the training step under rocprofv3 (kernel trace only):   [KSTAT_FLAGS=--no_prefetch] bash tools/kstat.sh [name-pattern]
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
PAT=${1:-.}
rm -rf gpurun_out/kstat; mkdir -p gpurun_out/kstat
rocprofv3 --kernel-trace --stats -d gpurun_out/kstat -o r --output-format csv -- python bench.py --steps 20 --warmup 3 --no_cpu_baseline --no_extras ${KSTAT_FLAGS:-} > gpurun_out/kstat/bench.log 2> gpurun_out/kstat/err.log
python - "$PAT" <<'PY'
import csv, re, sys
pat = re.compile(sys.argv[1])
rows = list(csv.DictReader(open("gpurun_out/kstat/r_kernel_stats.csv")))
for r in rows:
    n = re.sub(r"\(.*", "", r["Name"]).replace("void ", "")
    if pat.search(n):
        print("%-90s calls %5s avg %8.1f us" % (n[:90], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
tail -1 gpurun_out/kstat/bench.log | cut -c1-160
rm -rf gpurun_out/kstat/*.csv
