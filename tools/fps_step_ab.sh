# same-box A/B of the FPS variants on the training step: bash tools/fps_step_ab.sh [rounds]
cd "$GRAFT_REPO_ROOT"
for r in $(seq ${1:-2}); do
for v in "P2C_FPS_V1=1" "P2C_FPS_PPT=8" "P2C_FPS_PPT=16" "P2C_FPS_PPT=32"; do
  echo -n "$v: "; env $v python bench.py --no_extras --no_cpu_baseline --steps 40 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['top_kernels_ms'].get('fps'))"
done; done
