#!/bin/bash
# Same-box A/B of the training step between the tree and a baseline export in .ab_base/ (git archive <commit> | tar -x -C .ab_base; built there):
#   bash tools/ab_commits.sh [rounds] [bench flags...]      -> alternating runs, ms/step of each
cd "$GRAFT_REPO_ROOT"
R=${1:-3}; shift
for r in $(seq $R); do
  for d in .ab_base .; do
    echo -n "$d: "; (cd $d && python bench.py --no_extras --no_cpu_baseline --steps 40 "$@" 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['config']['loss'])")
  done
done
