"""Micro-bench of the GEMM family on the layer shapes of BASELINE configs[1] (B=32, N=8192).
   python tools/gemm_bench.py [fwd|bwd_data|bwd_weight|all]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from point2cyl_amd import _lib
from point2cyl_amd._lib import call, ptr, stream

SHAPES = [  # name, M, K, N
    ("sa1.0", 1048576, 4, 64), ("sa1.1", 1048576, 64, 64), ("sa1.2", 1048576, 64, 128),
    ("sa2.0", 262144, 132, 128), ("sa2.1", 262144, 128, 128), ("sa2.2", 262144, 128, 256),
    ("sa3.2", 4096, 512, 1024), ("fp3.0", 4096, 1280, 256), ("fp2.0", 16384, 384, 256),
    ("fp1.x", 262144, 128, 128), ("heads", 262144, 128, 20),
]
SMALL = [("sa3.0", 4096, 260, 256), ("sa3.1", 4096, 256, 512), ("sa3.2", 4096, 512, 1024), ("fp3.v", 32, 1024, 256), ("fp3.1", 4096, 256, 256),
         ("fp2.0", 16384, 384, 256), ("fp2.1", 16384, 256, 128), ("fp1.p", 16384, 128, 128)]
if os.environ.get("GEMM_BENCH_SMALL"):
    SHAPES = SMALL


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    dev = "cuda"
    L = _lib.lib()
    for name, M, K, N in SHAPES:
        X = torch.randn(M, K, device=dev)
        W = torch.randn(N, K, device=dev) * 0.1
        b = torch.randn(N, device=dev)
        Y = torch.empty(M, N, device=dev)
        sc, sh = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.1
        tiles = L.p2c_linear_stat_tiles(M)
        part = torch.zeros(64, 2, N, device=dev, dtype=torch.float64)
        dZ = torch.randn(M, N, device=dev)
        coef = torch.randn(5, N, device=dev)
        dX = torch.empty(M, K, device=dev)
        pstat = torch.rand(4, K, device=dev)
        partk = torch.zeros(64, 2, K, device=dev, dtype=torch.float64)
        dW = torch.zeros(N, K, device=dev); dW8 = torch.zeros(8, N, K, device=dev)
        fl = 2.0 * M * N * K
        res = []
        if which in ("fwd", "all"):
            t = timeit(lambda: call("p2c_linear_fwd_f32", ptr(X), K, ptr(W), K, ptr(b), ptr(Y), N, M, N, K, 1, ptr(sc), ptr(sh), None, 0, 1.0,
                                    ptr(part), stream()))
            by = 4.0 * M * (K + N)
            res.append("fwd %7.1f us %6.1f TF %5.2f TB/s" % (t * 1e6, fl / t / 1e12, by / t / 1e12))
        if which in ("bwd_data", "all"):
            t = timeit(lambda: call("p2c_linear_bwd_data_f32", ptr(dZ), N, ptr(Y), N, 1, ptr(coef), ptr(W), K, ptr(dX), K, M, N, K, None, 0, 1.0,
                                    ptr(X), K, ptr(pstat), ptr(partk), None, 0, stream()))
            by = 4.0 * M * (2 * N + 2 * K)
            res.append("bwd_data %7.1f us %6.1f TF %5.2f TB/s" % (t * 1e6, fl / t / 1e12, by / t / 1e12))
        if which in ("bwd_weight", "all"):
            t = timeit(lambda: call("p2c_linear_bwd_weight_f32", ptr(dZ), N, ptr(Y), N, 1, ptr(coef), ptr(X), K, 1, ptr(sc), ptr(sh), None, 0, 1.0,
                                    ptr(dW), K, 0, None, M, N, K, None, 0, stream()))
            by = 4.0 * M * (2 * N + K)
            res.append("bwd_w %7.1f us %6.1f TF %5.2f TB/s" % (t * 1e6, fl / t / 1e12, by / t / 1e12))
        if which in ("fused", "all") and L.p2c_linear_bwd_fused_supported(N, K, 1):
            parts = torch.zeros(64, 2, K, device=dev, dtype=torch.float64)
            t = timeit(lambda: call("p2c_linear_bwd_fused_f32", ptr(dZ), N, ptr(Y), N, 1, ptr(coef), None, 0, ptr(X), K, 1, ptr(sc), ptr(sh), ptr(W), K,
                                    ptr(dX), K, ptr(dW8), K, N * K, None, ptr(pstat), ptr(parts), M, N, K, stream()))
            by = 4.0 * M * (2 * N + 2 * K)
            res.append("FUSED %7.1f us %6.1f TF %5.2f TB/s" % (t * 1e6, 2 * fl / t / 1e12, by / t / 1e12))
        print("%-6s M=%8d K=%4d N=%4d | %s" % (name, M, K, N, " | ".join(res)))


if __name__ == "__main__":
    main()
