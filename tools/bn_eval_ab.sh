# A/B of the batched eval-mode BatchNorm affine (P2C_BN_EVAL_BATCH) on the forward and on the evaluation loop:   bash tools/bn_eval_ab.sh
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/bnab
python -m point2cyl_amd.train --pred_seg --pred_normal --pred_bb --synthetic 1024 --batch_size 32 --num_epochs 3 --quiet --logdir /tmp/tr > /dev/null 2>&1
: > gpurun_out/bnab/eval.log
for v in 1 0 1 0 1 0; do
  echo "P2C_BN_EVAL_BATCH=$v $(P2C_BN_EVAL_BATCH=$v python -m point2cyl_amd.eval --synthetic 4096 --batch_size 32 --logdir /tmp/tr --ckpt model.pth --dump_dir /tmp/ev 2>/dev/null | grep throughput)" >> gpurun_out/bnab/eval.log
done
for f in "--prefetch_group 1" "--prefetch_group 2" "--prefetch_group 8" "--no_prefetch" "--prefetch_group 4 --add_noise"; do
  echo "$f: $(python -m point2cyl_amd.eval --synthetic 4096 --batch_size 32 --logdir /tmp/tr --ckpt model.pth --dump_dir /tmp/ev $f 2>/dev/null | grep throughput)" >> gpurun_out/bnab/eval.log
done
cat gpurun_out/bnab/eval.log
