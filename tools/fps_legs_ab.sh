cd "$GRAFT_REPO_ROOT"
for v in "P2C_FPS_PPT=8" "P2C_FPS_PPT=16" "P2C_FPS_PPT=32"; do
  echo "== $v"
  env $v python bench.py --no_cpu_baseline --steps 20 --extras stages,eval_loop,dropin,forward_only 2>gpurun_out/x.err | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:d.get(k) for k in ('ms_per_step','sa1_stage_frac_best','sa1_stage_frac_serial','fps_ms','eval_ms_per_batch','dropin_ms_per_step')})"
  grep '^extra {"forward_only' gpurun_out/x.err | python -c "
import sys,json; d=json.loads(sys.stdin.read()[6:])['forward_only']; print({k:d.get(k) for k in ('ms','ms_geometry_precomputed','ms_pipelined','ms_pipelined_group4')})"
done
