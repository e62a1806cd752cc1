#!/bin/bash
# A/B of product-library variants on the training step: bash tools/ab/run.sh A B C   (each tools/ab/lib<X>.so, three alternating rounds)
cp point2cyl_amd/libp2c_hip.so /tmp/lib_keep.so
for round in 1 2 3; do
  for v in "$@"; do
    cp tools/ab/lib$v.so point2cyl_amd/libp2c_hip.so
    python bench.py --no_cpu_baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']
print('$v round $round: %.3f ms/step | fused %.3f fold0 %.3f fwd %.3f pool %.3f' % (d['ms_per_step'], k['p2c_linear_bwd_fused_f32']['ms_per_step'], k['p2c_linear_bwd_fused_fold0_f32']['ms_per_step'], k['p2c_linear_fwd_f32']['ms_per_step'], k['p2c_linear_fwd_pool_f32']['ms_per_step']))"
  done
done
cp /tmp/lib_keep.so point2cyl_amd/libp2c_hip.so
