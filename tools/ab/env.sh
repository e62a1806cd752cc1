#!/bin/bash
# A/B of an environment switch on the training step: bash tools/ab/env.sh P2C_GLUE_STREAM 0 1   (three alternating rounds)
var=$1; shift
for round in 1 2 3; do
  for v in "$@"; do
    env $var=$v python bench.py --no_cpu_baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$var=$v round $round: %.3f ms/step  loss %.5f' % (d['ms_per_step'], d['config']['loss']))"
  done
done
