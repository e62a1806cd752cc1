"""Do the gradients the graphed backbone hands to autograd reach .grad WITHOUT a copy each (AccumulateGrad adopts a fresh tensor)?  And what
does one drop-in step launch?   python tools/probe/dropin_grads.py"""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.profiler import profile, ProfilerActivity
from point2cyl_amd import synth
from point2cyl_amd.dropin.trainer_step import TrainerStep
dev = torch.device("cuda:0")
B, N, K = 32, 8192, 8
pcs, normals, seg, bb, _, _, axes, _, centers = synth.make_batch(B, N, K, seed=1234)
batch = tuple(x.to(dev) for x in (pcs, normals, seg, bb, axes, centers))
with torch.cuda.stream(torch.cuda.Stream(dev)):
    st = TrainerStep(K=K, batch_size=B, pred_extrusion=False, pred_center=False, device=dev)
    for _ in range(4):
        st(*batch)
    ptrs = collections.Counter(p.grad.untyped_storage().data_ptr() for p in st.model.parameters() if p.grad is not None)
    print("parameters with a gradient: %d, distinct gradient storages: %d (1 = every .grad is a view of the step's one flat copy)" % (sum(ptrs.values()), len(ptrs)))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        st(*batch)
    torch.cuda.synchronize()
    print("%.3f ms / step" % ((time.perf_counter() - t0) / 20 * 1e3))
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        st(*batch)
        torch.cuda.synchronize()
    ev = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
    c = collections.Counter()
    t = collections.Counter()
    for e in ev:
        n = e.name[:70]
        c[n] += 1
        t[n] += e.device_time_total if hasattr(e, "device_time_total") else e.cuda_time_total
    print("device activities of one step: %d" % len(ev))
    for n, v in t.most_common(25):
        print("  %-72s x%-4d %8.1f us" % (n, c[n], v))
    ka = prof.key_averages()
    top = sorted(ka, key=lambda e: -e.self_cpu_time_total)[:18]
    print("host side (self CPU time):")
    for e in top:
        print("  %-60s x%-5d %8.1f us" % (e.key[:60], e.count, e.self_cpu_time_total))
