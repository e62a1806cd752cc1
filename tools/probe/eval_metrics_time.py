"""Time eval.eval_metrics (everything of an evaluation batch after the forward) on one B = 32 x 8192 batch: wall time per call with the
queue drained each call, and with 20 calls queued back to back.   python tools/probe/eval_metrics_time.py   (under rocprofv3 --kernel-trace
--stats for the per-kernel view)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from point2cyl_amd import eval as ev, hostmem, synth
from point2cyl_amd.backbone import backbone

hostmem.setup_cli()
dev = torch.device("cuda", 0)
B, N, K = 32, 8192, 8
fl = ev.EvalFlags(K=K)
torch.manual_seed(0)
model = backbone(output_sizes=fl.pred_sizes()).to(dev).eval()
pcs, nrm, inst, bb, _, _, axes, _, cen = synth.make_batch(B, N, K, seed=5)
pcs, nrm, axes, cen = [t.to(dev, torch.float) for t in (pcs, nrm, axes, cen)]
inst, bb = inst.to(dev), bb.to(dev).float()
with torch.no_grad():
    X, W = model(pcs)
    for _ in range(3):
        m = ev.eval_metrics(X, W, pcs, nrm, inst, bb, axes, cen, fl)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(20):
        m = ev.eval_metrics(X, W, pcs, nrm, inst, bb, axes, cen, fl)
        torch.cuda.synchronize()
    one = (time.perf_counter() - t) / 20 * 1e3
    from point2cyl_amd import fitting
    acc = ev.Accumulator()
    counts = fitting.barrel_counts(inst.cpu(), bb.long().cpu(), K)      # what the evaluation loop's producer thread hands over
    t = time.perf_counter()
    for _ in range(40):
        m = ev.eval_metrics(X, W, pcs, nrm, inst, bb, axes, cen, fl, barrel_counts=counts, labels_validated=True)
        acc.add(m)
    t_host = (time.perf_counter() - t) / 40 * 1e3
    acc.sums()
    torch.cuda.synchronize()
    many = (time.perf_counter() - t) / 40 * 1e3
print("eval_metrics: %.3f ms a call (drained each call, with its two host syncs); without them (counts and label check from the host copy, "
      "asynchronous accumulator) %.3f ms a call back to back, %.3f ms of it host time to enqueue" % (one, many, t_host))
