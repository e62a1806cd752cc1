cd "$GRAFT_REPO_ROOT"
for i in 1 2; do
echo "== base"; P2C_FIT_TRACE_LIB=tools/libp2c_fit_trace_base.so python tools/fit_trace.py --hard 2>&1 | grep -v amdgpu.ids
echo "== new"; python tools/fit_trace.py --hard 2>&1 | grep -v amdgpu.ids
done
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "fit_fused or fitting_properties" 2>&1 | tail -3
python tools/bench_config4.py 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'], d['soft_membership_route']['ms'])"
