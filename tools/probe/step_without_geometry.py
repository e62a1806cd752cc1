"""Upper bound for any re-scheduling of the geometry prefetch: the replayed training step with the geometry PRECOMPUTED (no forked stream,
no FPS / ball query / 3-NN / CSR kernels at all) against the normal step on the same box.
    python tools/probe/step_without_geometry.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from point2cyl_amd import backbone as bb, ddp, ops, optim, step as stepmod, synth
from point2cyl_amd.graph import GraphedForwardBackward

dev = torch.device("cuda:0")
torch.cuda.set_stream(torch.cuda.Stream(dev))
B, N, K = 32, 8192, 8
fl = stepmod.StepFlags(K=K)
pcs, nrm, inst, bbl, _, _, axes, _, cen = synth.make_batch(B, N, K, seed=1234)
batch = tuple(x.to(dev) for x in (pcs.float(), nrm.float(), inst, bbl, axes.float(), cen.float()))


def run(mode):
    torch.manual_seed(0)
    model = bb.backbone(output_sizes=fl.pred_sizes()).to(dev).train()
    stepmod.update_momentum(model, 0.5)
    opt = optim.Adam(model.parameters(), lr=1e-3)
    sync = ddp.FlatGradSync(model.parameters(), 1)
    with torch.no_grad():
        geom0 = model.compute_geometry(batch[0])

    def fwd_bwd(geom=None):
        ops.step_done()
        with ops.step_arena(dev):
            out = stepmod.compute_losses_fused(model, *batch, fl, geom=geom0 if mode == "static" else geom)
            sync.zero()
            stepmod.backward(out)
        return {"total": out["total"].detach()}

    g = GraphedForwardBackward(model, fwd_bwd, prefetch_xyz=batch[0] if mode == "prefetch" else None, stream=torch.cuda.current_stream(),
                               draw_starts=mode == "prefetch")
    for _ in range(5):
        g(); opt.step(); ops.step_done()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 40
    for _ in range(n):
        g(); opt.step(); ops.step_done()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    g.release()
    return ms


for r in range(2):
    print("round %d: normal step (geometry of the next batch on the forked stream) %.3f ms | geometry precomputed, no forked stream %.3f ms"
          % (r, run("prefetch"), run("static")), flush=True)
