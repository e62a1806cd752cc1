# A/B of one environment switch on the training step:  bash tools/ab_env.sh VAR [rounds] [value_a value_b]
cd "$GRAFT_REPO_ROOT"
VAR=$1; R=${2:-2}; A=${3:-0}; B=${4:-1}
for r in $(seq $R); do
for v in $A $B; do
  echo -n "$VAR=$v: "; env $VAR=$v python bench.py --no_extras --no_cpu_baseline --steps 40 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d.get('top_kernels_ms', {})
print(d['ms_per_step'], d['config']['loss'], k)"
done; done
