#!/bin/bash
# Where does an evaluation batch's time go?  The eval CLI at B = 32 x 8192 on synthetic clouds: throughput lines for the serial loop and
# the pipelined ones, then a cProfile of the serial loop, once as it runs and once with blocking launches (GPU time then lands in the
# function that launched it).   bash tools/probe/eval_profile.sh [n_clouds=2048]
set -e
OUT=${OUT:-gpurun_out}
N=${1:-2048}
mkdir -p $OUT /tmp/tr /tmp/ev
python -m point2cyl_amd.train --pred_seg --pred_normal --pred_bb --synthetic 64 --batch_size 32 --num_epochs 1 --quiet --logdir /tmp/tr > /dev/null 2>&1
EV="python -m point2cyl_amd.eval --synthetic $N --batch_size 32 --logdir /tmp/tr --ckpt model.pth --dump_dir /tmp/ev"
for flags in "--no_prefetch" "--prefetch_group 1" "--prefetch_group 4"; do
  echo "$flags: $($EV $flags 2>/dev/null | grep throughput)"
done | tee $OUT/eval_throughput.txt
for mode in async blocking; do
  if [ $mode = blocking ]; then export HIP_LAUNCH_BLOCKING=1; fi
  python -m cProfile -o /tmp/ev/prof.$mode -m point2cyl_amd.eval --synthetic 1024 --batch_size 32 --logdir /tmp/tr --ckpt model.pth --dump_dir /tmp/ev --no_prefetch 2>/dev/null | grep throughput
  python - <<PY > $OUT/eval_profile_$mode.txt
import pstats
pstats.Stats("/tmp/ev/prof.$mode").sort_stats("cumulative").print_stats(90)
PY
done
