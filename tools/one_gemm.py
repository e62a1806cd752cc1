"""Run one GEMM entry point repeatedly (for rocprofv3 --pmc).  usage: one_gemm.py fwd|bwd_data|bwd_weight M K N [iters]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from point2cyl_amd import _lib
from point2cyl_amd._lib import call, ptr, stream
which, M, K, N = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
it = int(sys.argv[5]) if len(sys.argv) > 5 else 5
dev = "cuda"
L = _lib.lib()
X = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * 0.1; b = torch.randn(N, device=dev)
Y = torch.empty(M, N, device=dev); sc = torch.rand(K, device=dev) + 0.5; sh = torch.randn(K, device=dev) * 0.1
tiles = L.p2c_linear_stat_tiles(M); part = torch.zeros(64, 2, N, device=dev, dtype=torch.float64)
dZ = torch.randn(M, N, device=dev); coef = torch.randn(5, N, device=dev); dX = torch.empty(M, K, device=dev)
pstat = torch.rand(4, K, device=dev); partk = torch.zeros(64, 2, K, device=dev, dtype=torch.float64); dW = torch.zeros(N, K, device=dev); dW8 = torch.zeros(8, N, K, device=dev)
for _ in range(it):
    if which == "fwd":
        call("p2c_linear_fwd_f32", ptr(X), K, ptr(W), K, ptr(b), ptr(Y), N, M, N, K, 1, ptr(sc), ptr(sh), None, 0, 1.0, ptr(part), stream())
    elif which == "bwd_data":
        call("p2c_linear_bwd_data_f32", ptr(dZ), N, ptr(Y), N, 1, ptr(coef), ptr(W), K, ptr(dX), K, M, N, K, None, 0, 1.0, ptr(X), K, ptr(pstat),
             ptr(partk), None, 0, stream())
    elif which == "fused":
        parts = torch.zeros(64, 2, K, device=dev, dtype=torch.float64)
        call("p2c_linear_bwd_fused_f32", ptr(dZ), N, ptr(Y), N, 1, ptr(coef), None, 0, ptr(X), K, 1, ptr(sc), ptr(sh), ptr(W), K, ptr(dX), K, ptr(dW), K,
             0, None, ptr(pstat), ptr(parts), M, N, K, stream())
    else:
        call("p2c_linear_bwd_weight_f32", ptr(dZ), N, ptr(Y), N, 1, ptr(coef), ptr(X), K, 1, ptr(sc), ptr(sh), None, 0, 1.0, ptr(dW), K, 0, None, M, N, K,
             None, 0, stream())
torch.cuda.synchronize()
