cd "$GRAFT_REPO_ROOT"
export MASTER_ADDR=127.0.0.1 WORLD_SIZE=1 RANK=0 LOCAL_RANK=0
p=29700
for i in 1 2; do
for m in "async" "sync"; do
  p=$((p+1)); export MASTER_PORT=$p
  extra=""; [ "$m" = "async" ] && extra="--async_exchange"
  echo -n "forced $m: "; P2C_FORCE_EXCHANGE=1 python bench.py --steps 40 --warmup 5 --no_cpu_baseline --no_extras $extra 2>gpurun_out/xa.err | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['multi_gpu'])" || tail -3 gpurun_out/xa.err
done
echo -n "plain: "; env -u MASTER_PORT python bench.py --steps 40 --warmup 5 --no_cpu_baseline --no_extras 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"
done
