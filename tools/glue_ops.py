"""Which torch ops (not our kernels) launch device work inside one training step, and from where?  One eager step of bench.py's
configuration under torch.profiler with Python stacks; prints every aten op that ran a device kernel, grouped by call site.
   python tools/glue_ops.py            (on the GPU box)"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from point2cyl_amd import backbone as bb, ops, optim, step as stepmod, synth

dev = torch.device("cuda:0")
B, N, K = 32, 8192, 8
fl = stepmod.StepFlags(K=K)
torch.manual_seed(0)
model = bb.backbone(output_sizes=fl.pred_sizes()).to(dev)
opt = optim.Adam(model.parameters(), lr=1e-3)
pcs, nrm, inst, bbl, _, _, axes, _, cen = synth.make_batch(B, N, K, seed=1234)
pcs, nrm, inst, bbl, axes, cen = (x.to(dev) for x in (pcs, nrm, inst, bbl, axes, cen))
model.train()
stepmod.update_momentum(model, stepmod.get_batch_norm_decay(0, B, 200000))


def one():
    with ops.step_arena(dev):
        res = stepmod.compute_losses_fused(model, pcs, nrm, inst, bbl, axes, cen, fl)
        opt.zero_grad(set_to_none=True)
        stepmod.backward(res)
    opt.step()
    ops.step_done()


for _ in range(3):
    one()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    one()
    torch.cuda.synchronize()
rows = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    if e.device_type.name != "CPU" or not e.name.startswith("aten::"):
        continue
    dt = sum(k.duration for k in e.kernels) if e.kernels else 0
    if not e.kernels:
        continue
    site = next((s for s in e.stack if "point2cyl_amd" in s or "bench.py" in s or "glue_ops" in s), e.stack[0] if e.stack else "?")
    site = site.replace(os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + "/", "")
    rows[(e.name, site, str(e.input_shapes)[:60])][0] += len(e.kernels)
    rows[(e.name, site, str(e.input_shapes)[:60])][1] += dt
tot = 0
for (name, site, shp), (n, dt) in sorted(rows.items(), key=lambda kv: kv[0][1]):
    print("%-22s x%-2d %7.1f us  %-70s %s" % (name, n, dt, site[:70], shp))
    tot += n
print("device launches from torch ops:", tot)
