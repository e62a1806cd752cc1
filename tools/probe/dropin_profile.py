"""Where a drop-in step's time goes (the reference trainer's own composition on the drop-in import names, bench.py's dropin leg): wall time
of the step's sections with a device synchronisation after each (what the GPU needs), the step as it runs, and a cProfile of 40 steps.
python tools/probe/dropin_profile.py"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from point2cyl_amd import hostmem, synth
from point2cyl_amd.dropin.trainer_step import TrainerStep

hostmem.setup_cli()
dev = torch.device("cuda", 0)
B, N, K = 32, 8192, 8
pcs, normals, seg, bb, _, _, axes, _, centers = synth.make_batch(B, N, K, seed=1234)
batch = tuple(x.to(dev) for x in (pcs, normals, seg, bb, axes, centers))
with torch.cuda.stream(torch.cuda.Stream(dev)):
    torch.manual_seed(0)
    st = TrainerStep(K=K, batch_size=B, device=dev)
    for _ in range(5):
        st(*batch)
    torch.cuda.synchronize()
    per = []
    for _ in range(40):
        t = time.perf_counter()
        st(*batch)
        per.append(time.perf_counter() - t)
    per.sort()
    print("median step %.3f ms, fastest %.3f" % (per[20] * 1e3, per[0] * 1e3))
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(40):
        st(*batch)
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(28)
