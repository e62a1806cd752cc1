"""Per-launch timeline of ONE eager training step at configs[1] (HIP events on the launch stream around every C-ABI call):
    python tools/layer_times.py [--min_us 20]
prints the calls in launch order with their time, flops and TFLOP/s (median of 3 steps), and the total per entry point."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
ap = argparse.ArgumentParser(); ap.add_argument("--min_us", type=float, default=20.0); a = ap.parse_args()
from point2cyl_amd import ops, step, synth
from point2cyl_amd.backbone import backbone
dev = torch.device("cuda:0"); B, N, K = 32, 8192, 8
fl = step.StepFlags(K=K)
batch = tuple(x.to(dev) for x in [synth.make_batch(B, N, K, seed=1234)[i] for i in (0, 1, 2, 3, 6, 8)])
torch.manual_seed(0)
model = backbone(output_sizes=fl.pred_sizes()).to(dev).train()
def fwd_bwd():
    ops.step_done()          # the previous step's gradients are discarded here
    with ops.step_arena(dev):
        out = step.compute_losses_fused(model, *batch, fl)
        for p in model.parameters(): p.grad = None
        step.backward(out)
for _ in range(2): fwd_bwd()
runs = []
for _ in range(3):
    ops.PROFILE.reset(enabled=True); fwd_bwd(); torch.cuda.synchronize(); ops.PROFILE.enabled = False
    runs.append([(n, e0.elapsed_time(e1) * 1e3, f) for n, e0, e1, f, _ in ops.PROFILE.records])
assert len({len(r) for r in runs}) == 1
tot = {}; allus = 0.0
for i, (n, _, f) in enumerate(runs[0]):
    us = float(np.median([r[i][1] for r in runs])); allus += us
    d = tot.setdefault(n, [0, 0.0, 0.0]); d[0] += 1; d[1] += us; d[2] += f
    if us >= a.min_us: print("%4d %-38s %8.1f us %8.2f GFLOP %6.1f TF/s" % (i, n, us, f / 1e9, f / us / 1e6 if us else 0))
print("---- per entry point (sum %.0f us over %d launches)" % (allus, len(runs[0])))
for n, (c, us, f) in sorted(tot.items(), key=lambda kv: -kv[1][1]): print("%-40s x%3d %8.1f us %8.2f GFLOP %6.1f TF/s" % (n, c, us, f / 1e9, f / us / 1e6 if us else 0))
