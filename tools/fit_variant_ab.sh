# one workgroup per CU (cloud parked in LDS) against two half-size workgroups per CU (P2C_FIT_VARIANT=1), configs[3]
cd "$GRAFT_REPO_ROOT"
for v in 0 1 0 1; do
P2C_FIT_VARIANT=$v python tools/bench_config4.py --no_cpu 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('variant $v', d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'], d['soft_membership_route']['ms'])"
done
