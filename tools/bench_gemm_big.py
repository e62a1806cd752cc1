"""The decoder's large products: csrc/gemm_big.hip against the generic tiled route of csrc/gemm.hip, same operands (HIP events, one stream).
    python tools/bench_gemm_big.py [--M 262144]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from point2cyl_amd import _lib  # noqa: E402
from point2cyl_amd._lib import call, ptr, stream  # noqa: E402


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--M", type=int, default=262144)
    ap.add_argument("--one", action="store_true", help="the 512 x 512 shape only (profiling runs)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    L = _lib.lib()
    out = []
    for (N, K) in [(512, 512), (512, 260), (256, 512), (512, 256)][:1 if a.one else 4]:
        M = a.M
        X = torch.randn(M, K, device=dev)
        W = torch.randn(N, K, device=dev) / K ** 0.5
        b = torch.randn(N, device=dev)
        dZ = torch.randn(M, N, device=dev)
        Z = torch.randn(M, K, device=dev) * 0.02
        Y, dX = torch.empty(M, N, device=dev), torch.empty(M, K, device=dev)
        ws = torch.empty(L.p2c_linear_big_ws_bytes(N, K), dtype=torch.uint8, device=dev)
        fl = 2.0 * M * N * K
        t = {
            "fwd_generic": timed(lambda: call("p2c_linear_fwd_f32", ptr(X), K, ptr(W), K, ptr(b), ptr(Y), N, M, N, K, 0, None, None, None, 0, 1.0, None, stream())),
            "fwd_big": timed(lambda: call("p2c_linear_fwd_big_f32", ptr(X), K, ptr(W), K, ptr(b), ptr(Y), N, M, N, K, ptr(ws), stream())),
            "bwd_sig_generic": timed(lambda: call("p2c_linear_bwd_data_sig_f32", ptr(dZ), N, ptr(W), K, ptr(Z), K, 100.0, 20.0, ptr(dX), K, M, N, K, stream())),
            "bwd_sig_big": timed(lambda: call("p2c_linear_bwd_data_big_f32", ptr(dZ), N, ptr(W), K, ptr(Z), K, 100.0, 20.0, ptr(dX), K, M, N, K, ptr(ws), stream())),
        }
        out.append({"M": M, "N": N, "K": K, "ms": {k: round(v, 4) for k, v in t.items()},
                    "tflops_fp32_equivalent": {k: round(fl / v / 1e9, 1) for k, v in t.items()}})
        print(json.dumps(out[-1]), flush=True)


if __name__ == "__main__":
    main()
