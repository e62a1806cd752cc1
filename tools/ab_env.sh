# A/B of one environment switch on the training step:  bash tools/ab_env.sh VAR [rounds]
cd "$GRAFT_REPO_ROOT"
VAR=$1; R=${2:-2}
for r in $(seq $R); do
for v in 0 1; do
  echo -n "$VAR=$v: "; env $VAR=$v python bench.py --no_extras --no_cpu_baseline --steps 40 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print(d['ms_per_step'], d['config']['loss'], {n: k[n]['ms_per_step'] for n in ('p2c_linear_fwd_pool_f32','p2c_linear_bwd_pool_alg_f32','p2c_csr_gather_bn_f32','p2c_group_linear_bwd_f32','p2c_three_interp_bias_stats_f32','p2c_group_linear_bias_stats_f32') if n in k})"
done; done
