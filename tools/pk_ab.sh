# Same-box A/B of the library built WITH packed fp32 instructions against the packed-free build that ships (point2cyl_amd/build.py).
# Build the other library here first (no GPU needed):
#   cd point2cyl_amd/csrc && mkdir -p /tmp/pk && for f in *.hip; do hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -c $f -o /tmp/pk/${f%.hip}.o; done
#   hipcc -shared -fPIC --offload-arch=gfx950 -o ../../tools/libp2c_hip_pk.so /tmp/pk/*.o
# then:  gpurun -- 'bash tools/pk_ab.sh'      (P2C_LIB selects the library; results of round 6: profiles/r06_fps_packed_hazard.log)
cd "$GRAFT_REPO_ROOT"
for r in 1 2 3; do
for v in "" tools/libp2c_hip_pk.so; do
  echo -n "lib ${v:-default}: "; P2C_LIB=${v:+$PWD/$v} python bench.py --no_extras --no_cpu_baseline --steps 40 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['config']['loss'])"
done; done
for v in "" tools/libp2c_hip_pk.so; do
  echo -n "fit lib ${v:-default}: "; P2C_LIB=${v:+$PWD/$v} python tools/bench_config4.py --no_cpu 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print(d['ms_per_step'], d['soft_membership_route']['ms'])"
  echo -n "sa1 stage lib ${v:-default}: "; P2C_LIB=${v:+$PWD/$v} python tools/bench_sa1_forward.py 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print(d['graph_serial']['ms'], d['kernels']['p2c_fps_f32']['us_per_pass'])"
done
for v in tools/libp2c_hip_pk.so; do echo "== stress $v"; P2C_LIB=$PWD/$v timeout 200 python tools/stress_prefetch.py 1500 2>&1 | tail -1; done
