"""How much HOST time does one replayed training step take, and where?  (Is the step host-bound?)
    python tools/probe/host_time.py"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from point2cyl_amd import backbone as bb, ddp, optim, step as stepmod, synth
from point2cyl_amd.train import Runner

ap = argparse.ArgumentParser(); a = ap.parse_args()
dev = torch.device("cuda:0")
torch.cuda.set_stream(torch.cuda.Stream(dev))
B, N, K = 32, 8192, 8
fl = stepmod.StepFlags(K=K)
torch.manual_seed(0)
model = bb.backbone(output_sizes=fl.pred_sizes()).to(dev).train()
opt = optim.Adam(model.parameters(), lr=1e-3)
sync = ddp.FlatGradSync(model.parameters(), 1)
pcs, nrm, inst, bbl, _, _, axes, _, cen = synth.make_batch(B, N, K, seed=1234)
cur = tuple(x.to(dev) for x in (pcs.float(), nrm.float(), inst, bbl, axes.float(), cen.float()))
r = Runner(model, opt, sync, fl, dev, B, N, K, stream=torch.cuda.current_stream())
r.load(cur, cur[0])
for _ in range(5):
    r.step(0.5)
torch.cuda.synchronize()
# instrument
t = {"replay": 0.0, "stage": 0.0, "adam": 0.0, "total": 0.0}
g = r.graph
orig_stage = g.starts.stage
def stage():
    t0 = time.perf_counter(); orig_stage(); t["stage"] += time.perf_counter() - t0
g.starts.stage = stage
orig_replay = g.graph.replay
def rep():
    t0 = time.perf_counter(); orig_replay(); t["replay"] += time.perf_counter() - t0
g.graph.replay = rep
orig_step = opt.step
def ostep(*aa, **kk):
    t0 = time.perf_counter(); out = orig_step(*aa, **kk); t["adam"] += time.perf_counter() - t0; return out
opt.step = ostep
n = 100
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    r.step(0.5)
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print("host time per step %.3f ms (graph launch %.3f, FPS-start staging %.3f, Adam %.3f, rest %.3f); wall per step with the final sync %.3f ms"
      % (t_host / n * 1e3, t["replay"] / n * 1e3, t["stage"] / n * 1e3, t["adam"] / n * 1e3,
         (t_host - t["replay"] - t["stage"] - t["adam"]) / n * 1e3, t_all / n * 1e3))
