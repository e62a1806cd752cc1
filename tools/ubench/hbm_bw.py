"""HBM streaming rates through plain torch kernels: fill (write only), copy (read + write), sum (read only)."""
import torch
dev = "cuda"
n = 256 * 1024 * 1024            # 1 GiB of fp32
x = torch.empty(n, device=dev); y = torch.empty(n, device=dev)
def t(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
tf = t(lambda: x.zero_()); print("fill  : %.2f TB/s written" % (4 * n / tf / 1e12))
tc = t(lambda: y.copy_(x)); print("copy  : %.2f TB/s read + %.2f TB/s written (%.2f total)" % (4 * n / tc / 1e12, 4 * n / tc / 1e12, 8 * n / tc / 1e12))
ts = t(lambda: x.sum()); print("sum   : %.2f TB/s read" % (4 * n / ts / 1e12))
ta = t(lambda: torch.add(x, y, out=y)); print("add   : 2 reads + 1 write: %.2f TB/s total" % (12 * n / ta / 1e12))
