cd "$GRAFT_REPO_ROOT"
for r in 1 2; do
for v in 0 1; do
  echo -n "P2C_POOL_ALG=$v: "; P2C_POOL_ALG=$v python bench.py --no_extras --no_cpu_baseline --steps 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['config']['loss'], {k:v['ms_per_step'] for k,v in d['kernels'].items() if 'pool' in k or 'fused_f32' in k})"
done; done
python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_configs.py tests/test_gpu_parity.py -q -x -k "mlp_stack or backbone or config2 or configs or step or pool" 2>&1 | tail -5
