"""A/B of SA1's last layer's backward at configs[1]'s size (M = 1,048,576 rows, Co = 128, Ci = 64, ns = 64): the generic pooled backward
(p2c_linear_bwd_fused_f32, grad_mode 2: reads Y) against csrc/bwd_pool.hip (p2c_linear_bwd_pool_alg_f32: no Y).  HIP events, back to back."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from point2cyl_amd._lib import call, lib, ptr, stream

DEV = "cuda"
G, Co, Ci, ns = 16384, 128, 64, 64
M = G * ns
torch.manual_seed(0)
X = torch.randn(M, Ci, device=DEV); sc2 = torch.rand(Ci, device=DEV) + 0.5; sh2 = torch.randn(Ci, device=DEV) * 0.3
W = torch.randn(Co, Ci, device=DEV) * 0.1; b = torch.randn(Co, device=DEV) * 0.1
Y = torch.relu(sc2 * X + sh2) @ W.t() + b
arg = torch.randint(0, ns, (G, Co), device=DEV, dtype=torch.int32)
ywin = torch.gather(Y.view(G, ns, Co), 1, arg.long().unsqueeze(1)).squeeze(1).contiguous()
dout = torch.randn(G, Co, device=DEV)
coef = torch.stack([torch.rand(Co, device=DEV) + 0.5, torch.randn(Co, device=DEV) * 0.2, torch.rand(Co, device=DEV) + 0.5,
                    torch.randn(Co, device=DEV) * 0.01, torch.randn(Co, device=DEV) * 0.01]).contiguous()
pstat = torch.stack([sc2, sh2, torch.randn(Ci, device=DEV) * 0.1, torch.rand(Ci, device=DEV) + 0.5]).contiguous()
dX = torch.empty(M, Ci, device=DEV); dW = torch.empty(Co, Ci, device=DEV); dW8 = torch.zeros(8, Co, Ci, device=DEV)
parts = torch.zeros(64, 2, Ci, device=DEV, dtype=torch.float64)
acc = torch.empty(lib().p2c_linear_bwd_pool_alg_ws_bytes(Co, Ci) // 4 + 4, device=DEV)


def alg():
    call("p2c_linear_bwd_pool_alg_f32", ptr(dout), Co, ptr(ywin), ptr(arg), ptr(coef), ptr(X), Ci, ptr(sc2), ptr(sh2), ptr(W), Ci, ptr(b), ptr(dX), Ci,
         ptr(pstat), ptr(parts), ptr(acc), ptr(dW), Ci, M, Co, Ci, ns, stream())


def generic():
    call("p2c_linear_bwd_fused_f32", ptr(dout), Co, ptr(Y), Co, 2, ptr(coef), ptr(arg), ns, ptr(X), Ci, 1, ptr(sc2), ptr(sh2), ptr(W), Ci, ptr(dX), Ci,
         ptr(dW8), Ci, Co * Ci, None, ptr(pstat), ptr(parts), M, Co, Ci, stream())


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for r in range(3):
    tg, ta = timeit(generic), timeit(alg)
    print("round %d: generic (reads Y, 1.07 GB) %.1f us = %.2f TB/s | pool_alg (no Y, 0.54 GB) %.1f us = %.2f TB/s" %
          (r, tg, 4.0 * M * (Co + 2 * Ci) / tg / 1e6, ta, 4.0 * M * 2 * Ci / ta / 1e6))

if "--trace" in sys.argv:
    import ctypes, subprocess, numpy as np
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = os.path.join(root, "tools", "libp2c_pool_trace.so")
    L = ctypes.CDLL(so)
    vp, ci = ctypes.c_void_p, ctypes.c_int
    L.p2c_linear_bwd_pool_alg_f32.argtypes = [vp, ci, vp, vp, vp, vp, ci, vp, vp, vp, ci, vp, vp, ci, vp, vp, vp, vp, ci, ci, ci, ci, ci, vp]
    for _ in range(2):
        assert L.p2c_linear_bwd_pool_alg_f32(ptr(dout), Co, ptr(ywin), ptr(arg), ptr(coef), ptr(X), Ci, ptr(sc2), ptr(sh2), ptr(W), Ci, ptr(b), ptr(dX), Ci,
                                             ptr(pstat), ptr(parts), ptr(acc), ptr(dW), Ci, M, Co, Ci, ns, None) == 0
    torch.cuda.synchronize()
    st = np.zeros(12, dtype=np.uint64)
    assert L.p2c_pool_alg_trace_read(st.ctypes.data_as(vp)) == 0
    names = ["stage", "barrier", "Gs W + A Q MFMAs", "Gram MFMAs", "winner gather", "epilogue", "end barrier", "loop top"]
    tiles = G // 256
    print("workgroup 0, wave 0, shader cycles per 64-row tile (%d tiles): " % tiles + " | ".join("%s %d" % (n, v // tiles) for n, v in zip(names, st[:8])) +
          " | total %d" % (int(st[:8].sum()) // tiles))
    print("prologue %d cycles | loop %d cycles = %.1f us of the 100 MHz counter -> shader clock %.2f GHz | flush %d cycles" %
          (st[8], st[9], st[10] / 100.0, st[9] / (st[10] / 100.0) / 1e3, st[11]))
