"""Whole-backbone forward, B = 32 x 8192, steady state per batch for geometry groups of G batches (graph.PipelinedForward(group=G)).
python tools/probe/forward_groups.py [G ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from point2cyl_amd import hostmem, ops, synth
from point2cyl_amd.backbone import backbone
from point2cyl_amd.graph import PipelinedForward

hostmem.setup_cli()
dev = torch.device("cuda", 0)
B, N, K = 32, 8192, 8
torch.manual_seed(0)
model = backbone(output_sizes=[3, 2 * K]).to(dev).train()
pcs = synth.make_batch(B, N, K, seed=5)[0].to(dev, torch.float)
Gs = [int(x) for x in sys.argv[1:]] or [1, 2, 4, 6, 8]
st = torch.cuda.Stream(dev)
st.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(st):
    for G in Gs:
        grp = [pcs] * G
        pf = PipelinedForward(model, grp, stream=st, group=G)
        try:
            for _ in range(3):
                pf(grp)
            torch.cuda.synchronize()
            n = max(6, 48 // G)
            t0 = time.perf_counter()
            for _ in range(n):
                pf(grp)
            torch.cuda.synchronize()
            t = (time.perf_counter() - t0) / (n * G) * 1e3
        finally:
            pf.release()
        ops.step_done()
        print("G = %d: %.4f ms per batch of %d clouds (%.1f M points/s)" % (G, t, B, B * N / t / 1e3))
