"""What the evaluation loop's loader thread spends per batch of 32 x 8192 points (collate into pinned buffers, label range check, barrel
counts, upload), each timed alone on the host.   python tools/probe/loader_time.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from point2cyl_amd import eval as ev, fitting, hostmem, synth

hostmem.setup_cli()
dev = torch.device("cuda", 0)
ds = [synth.SyntheticExtrusionDataset(64, 8192, 8, seed=1)[i] for i in range(64)]
col = ev._PinnedCollate()


def t(fn, n=40):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


b = col(ds[:32])
inst, bb = b[2], b[3]
print("collate (6 fields x 32 items -> pinned): %.3f ms" % t(lambda: col(ds[:32])))
print("label range check (min, max):           %.3f ms" % t(lambda: (int(inst.min()), int(inst.max()))))
print("barrel counts:                          %.3f ms" % t(lambda: fitting.barrel_counts(inst.long(), bb.long(), 8)))
up = lambda: [x.to(dev, d, non_blocking=True) for x, d in ((b[0], torch.float), (b[1], torch.float), (b[2], torch.long), (b[3], torch.float), (b[6], torch.float), (b[8], torch.float))]
print("upload (6 tensors, non_blocking) + sync: %.3f ms" % t(lambda: (up(), torch.cuda.current_stream().synchronize())))
