"""Backbone forward at B = 32 x 8192 with the geometry precomputed, no gradient: train-mode BatchNorm (batch statistics) against eval mode
(running statistics) - as HIP-graph replays and per stack from HIP events.   python tools/probe/forward_modes.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from point2cyl_amd import hostmem, measure, ops, synth
from point2cyl_amd.backbone import backbone

hostmem.setup_cli()
dev = torch.device("cuda", 0)
B, N, K = 32, 8192, 8
torch.manual_seed(0)
model = backbone(output_sizes=[3, 2 * K]).to(dev)
pcs = synth.make_batch(B, N, K, seed=5)[0].to(dev, torch.float)
with torch.no_grad():
    model.train()
    for _ in range(3):
        model(pcs)                      # running statistics that mean something
    geom = model.compute_geometry(pcs, with_csr=False)
    for mode in ("train", "eval"):
        model.train() if mode == "train" else model.eval()

        def fn():
            ops.step_done()
            with ops.step_arena(dev):
                return model.forward_heads(pcs, geom)[0]
        g, _ = measure.capture(fn)
        ms = measure.replay_ms([g], 40)
        ops.PROFILE.reset(enabled=True)
        fn()
        torch.cuda.synchronize()
        prof = ops.PROFILE.summary()
        ops.PROFILE.enabled = False
        top = sorted(prof.items(), key=lambda kv: -kv[1]["ms"])[:8]
        print("%s mode: %.4f ms per forward (graph replay); per entry point (one eager pass): %s"
              % (mode, ms, ", ".join("%s %.3f ms x%d" % (k, v["ms"], v["launches"]) for k, v in top)))
