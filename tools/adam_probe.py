"""Why does torch.optim.Adam launch ~150 small kernels on the drop-in step?  Counts device kernels of one optimizer.step() for gradients that
(a) are randn_like(p), (b) come out of our backward.  Prints parameter / gradient strides that differ."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import time
from point2cyl_amd.backbone import backbone
from point2cyl_amd import synth, autograph

dev = "cuda"
torch.manual_seed(0)
m = backbone(output_sizes=[3, 16]).to(dev).train()
opt = torch.optim.Adam(m.parameters(), lr=1e-3)
pcs = synth.make_batch(4, 2048, 8, seed=1)[0].float().to(dev)


def count(tag):
    for _ in range(3):
        opt.step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(20):
        opt.step()
    e1.record(); torch.cuda.synchronize()
    print(tag, "opt.step: %.1f us device, %.1f us host+device" % (e0.elapsed_time(e1) * 50, (time.perf_counter() - t0) * 5e4))


for p in m.parameters():
    p.grad = torch.randn_like(p)
count("randn_like grads")
autograph.ENABLED = False
for p in m.parameters():
    p.grad = None
X, W = m(pcs); (X.square().mean() + W.square().mean()).backward()
bad = [(n, tuple(p.shape), p.stride(), p.grad.stride(), p.grad.is_contiguous()) for n, p in m.named_parameters() if p.grad.stride() != p.stride()]
print("params whose grad strides differ:", len(bad), bad[:6])
print("dtypes", {p.grad.dtype for p in m.parameters()}, "non-contiguous grads", sum(not p.grad.is_contiguous() for p in m.parameters()))
count("eager backward grads")
autograph.ENABLED = True
for p in m.parameters():
    p.grad = None
X, W = m(pcs); (X.square().mean() + W.square().mean()).backward()
bad = [(n, tuple(p.shape), p.stride(), p.grad.stride()) for n, p in m.named_parameters() if p.grad.stride() != p.stride()]
print("autograph: params whose grad strides differ:", len(bad), bad[:6])
count("autograph grads")
