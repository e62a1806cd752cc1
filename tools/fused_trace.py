"""Phase timeline of the ping-pong fused backward kernel (csrc/bwd_fused.hip built with -DP2C_TRACE into a throw-away
library).  Build here:  python tools/fused_trace.py --build      Run on the GPU box:  python tools/fused_trace.py [M Co Ci]
Stamps (shader clock) of workgroup 0, thread 0 of each half:
  0 MFMA phase start | 1 end of dW MFMAs | 2 end of MFMA phase (before barrier) | 3 after barrier
  4 next tile transformed into LDS (waits for its global loads) | 5 prefetch issued | 6 dX stored + sums | 7 after 2nd barrier"""
import ctypes, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
FWD = "--fwd" in sys.argv             # trace fwd_pp.hip instead of bwd_fused.hip
NODATA = "--nodata" in sys.argv       # MFMA phases only: no global traffic, no LDS staging (isolates the MFMA loops)
LOCK = "--lockstep" in sys.argv       # both halves in the same phase (no one-phase offset)
OLD = "--old" in sys.argv             # a library built by hand from an earlier revision of bwd_fused.hip (A/B runs)
LIB = os.path.join(HERE, "libp2c_trace.so" if FWD else "libp2c_trace_nodata.so" if NODATA else "libp2c_trace_lock.so" if LOCK else "libp2c_trace_old.so" if OLD else "libp2c_trace.so")
if "--build" in sys.argv:
    shim = os.path.join(HERE, "trace_shim.hip")       # fwd_pp + fwd_pp3 + bwd_fused + bwd_fused3 in one library (they share the mode switch)
    for lib, extra in ((os.path.join(HERE, "libp2c_trace.so"), []), (os.path.join(HERE, "libp2c_trace_nodata.so"), ["-DP2C_TRACE_NODATA"]),
                       (os.path.join(HERE, "libp2c_trace_lock.so"), ["-DP2C_LOCKSTEP"])):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-ffp-contract=off", "-Wno-unused-value", "-DP2C_TRACE"] + extra +
                              ["-shared", "-o", lib, shim])
        print(lib)
    sys.exit(0)
sys.path.insert(0, ROOT)
import numpy as np
import torch
args = [a for a in sys.argv[1:] if not a.startswith("--")]
M, Co, Ci = (int(args[0]), int(args[1]), int(args[2])) if len(args) >= 3 else (262144, 128, 128)
L = ctypes.CDLL(LIB)
vp, ci = ctypes.c_void_p, ctypes.c_int
if FWD:
    dev = "cuda"
    X = torch.randn(M, Ci, device=dev); W = torch.randn(Co, Ci, device=dev) * .1; b = torch.randn(Co, device=dev); Y = torch.empty(M, Co, device=dev)
    sc = torch.rand(Ci, device=dev) + .5; sh = torch.randn(Ci, device=dev) * .1; parts = torch.zeros(64, 2, Co, device=dev, dtype=torch.float64)
    L.p2c_trace_fwd.argtypes = [vp, ci, vp, ci, vp, vp, ci, ci, ci, ci, vp, vp, vp, vp]
    st = torch.cuda.current_stream().cuda_stream
    def runf():
        assert L.p2c_trace_fwd(X.data_ptr(), Ci, W.data_ptr(), Ci, b.data_ptr(), Y.data_ptr(), Co, M, Co, Ci, sc.data_ptr(), sh.data_ptr(), parts.data_ptr(), st) == 0
    for _ in range(3):
        runf()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); runf(); e1.record(); torch.cuda.synchronize()
    print("fwd_pp kernel %.1f us" % (e0.elapsed_time(e1) * 1e3))
    buf = np.zeros((2, 12, 8), dtype=np.uint64)
    assert L.p2c_trace_read(buf.ctypes.data_as(vp)) == 0
    t0 = buf[0, 2, 0]
    names = ["mfma0", "mfma", "bar1", "sums+frag", "readback", "stores", "stage", "prefetch+bar2"]
    for it in range(2, 10):
        for h in range(2):
            row = buf[h, it].astype(np.int64) - int(t0)
            d = np.diff(row)
            print("it %d half %d  start %8d | " % (it, h, row[0]) + "  ".join("%s %6d" % (n, v) for n, v in zip(names[1:], d)))
    sys.exit(0)
L.p2c_linear_bwd_fused_f32.argtypes = [vp, ci, vp, ci, ci, vp, vp, ci, vp, ci, ci, vp, vp, vp, ci, vp, ci, vp, ci, ctypes.c_longlong, vp, vp, vp, ci, ci, ci, vp]
dev = "cuda"
dZ = torch.randn(M, Co, device=dev); Y = torch.randn(M, Co, device=dev); X = torch.randn(M, Ci, device=dev)
coef = torch.randn(5, Co, device=dev); sc = torch.rand(Ci, device=dev) + .5; sh = torch.randn(Ci, device=dev) * .1
W = torch.randn(Co, Ci, device=dev) * .1; dX = torch.empty(M, Ci, device=dev); dW8 = torch.zeros(8, Co, Ci, device=dev)
pstat = torch.rand(4, Ci, device=dev); parts = torch.zeros(64, 2, Ci, device=dev, dtype=torch.float64)
st = torch.cuda.current_stream().cuda_stream
def run():
    rc = L.p2c_linear_bwd_fused_f32(dZ.data_ptr(), Co, Y.data_ptr(), Co, 1, coef.data_ptr(), None, 0, X.data_ptr(), Ci, 1, sc.data_ptr(), sh.data_ptr(),
                                    W.data_ptr(), Ci, dX.data_ptr(), Ci, dW8.data_ptr(), Ci, Co * Ci, None, pstat.data_ptr(), parts.data_ptr(), M, Co, Ci, st)
    assert rc == 0, rc
for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    run()
e1.record(); torch.cuda.synchronize()
print("kernel %.1f us (mean of 20 back-to-back launches: sustained, at the socket power cap)" % (e0.elapsed_time(e1) * 1e3 / 20))
# single launches with idle gaps: what the kernel takes when the moving-average power is below the cap, as inside a training step
import time
ts = []
for _ in range(12):
    time.sleep(0.03)
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3)
ts.sort()
print("kernel %.1f us median, %.1f us min (single launches, 30 ms apart)" % (ts[len(ts) // 2], ts[0]))
if not OLD:
    wg = np.zeros((1024, 4), dtype=np.uint64)
    assert L.p2c_trace_read_wg(wg.ctypes.data_as(vp)) == 0
    wg = wg[:min(256, (M + 31) // 32)].astype(np.float64)
    cyc = wg[:, 2] - wg[:, 0]; us = (wg[:, 3] - wg[:, 1]) / 100.0
    span = (wg[:, 3].max() - wg[:, 1].min()) / 100.0
    mid = np.zeros((1024, 8), dtype=np.uint64)
    assert L.p2c_trace_read_wg_mid(mid.ctypes.data_as(vp)) == 0
    mid = mid[:len(wg)].astype(np.float64)
    print("cycles (median over workgroups): prologue %.0f | main loop %.0f | flush %.0f" % (np.median(mid[:, 0] - wg[:, 0]), np.median(mid[:, 1] - mid[:, 0]),
                                                                                          np.median(wg[:, 2] - mid[:, 1])))
    print("prologue stages (median cycles): first loads issued %.0f | W staged %.0f | constants %.0f | first tile transformed + stored, second requested %.0f | barrier %.0f"
          % (np.median(mid[:, 2] - wg[:, 0]), np.median(mid[:, 3] - mid[:, 2]), np.median(mid[:, 4] - mid[:, 3]), np.median(mid[:, 5] - mid[:, 4]),
             np.median(mid[:, 0] - mid[:, 5])))
    print("workgroups: %d | duration us min %.1f median %.1f max %.1f | first start -> last end %.1f us | start skew %.1f us | shader clock %.2f GHz (median)"
          % (len(us), us.min(), np.median(us), us.max(), span, (wg[:, 1].max() - wg[:, 1].min()) / 100.0, np.median(cyc / us) / 1e3))
if not OLD:
    wg = np.zeros((1024, 4), dtype=np.uint64)
    assert L.p2c_trace_read_wg(wg.ctypes.data_as(vp)) == 0
    wg = wg[:min(256, (M + 31) // 32)].astype(np.float64)
    cyc = wg[:, 2] - wg[:, 0]; us = (wg[:, 3] - wg[:, 1]) / 100.0
    span = (wg[:, 3].max() - wg[:, 1].min()) / 100.0
    print("workgroups: %d | duration us min %.1f median %.1f max %.1f | first start -> last end %.1f us | start skew %.1f us | shader clock %.2f GHz (median)"
          % (len(us), us.min(), np.median(us), us.max(), span, (wg[:, 1].max() - wg[:, 1].min()) / 100.0, np.median(cyc / us) / 1e3))
buf = np.zeros((2, 12, 8), dtype=np.uint64)
assert L.p2c_trace_read(buf.ctypes.data_as(vp)) == 0
t0 = buf[0, 2, 0]
names = ["mfma0", "dW_end", "mfma_end", "bar1", "stage", "prefetch", "dx+sums", "bar2"]
if L.p2c_get_mfma_mode():      # bwd_fused3.hip: one wave per SIMD, phases in program order (half 1 does not exist)
    if Ci == 128 and os.environ.get("P2C_BWD3_ROLES", "1") != "0":      # role-split form: row 0 = dY/dX/epilogue waves, row 1 = X/dW waves
        for it in range(2, 10):
            for h in range(2):
                r = buf[h, it].astype(np.int64) - int(t0)
                print("it %d role %s start %8d | staged %6d  barrier1 %6d  prefetch+mfma %6d  %s %6d | barrier2 -> next start %6d"
                      % (it, "AB"[h], r[0], r[1] - r[0], r[2] - r[1], r[3] - r[2], "epilogue" if h == 0 else "-       ", r[7] - r[3],
                         int(buf[h, it + 1, 0]) - int(buf[h, it, 7])))
        sys.exit(0)
    names = ["start", "dY staged", "X staged", "barrier1", "prefetch issued", "dW mfma", "dX mfma", "epilogue(+bar2 -> next start)"]
    for it in range(2, 10):
        row = buf[0, it].astype(np.int64) - int(t0)
        d = np.diff(row)
        print("it %d  start %8d | " % (it, row[0]) + "  ".join("%s %6d" % (n, v) for n, v in zip(names[1:], d)) + "  | to next start %6d" % (int(buf[0, it + 1, 0]) - int(buf[0, it, 7])))
    sys.exit(0)
for it in range(2, 10):
    for h in range(2):
        row = buf[h, it].astype(np.int64) - int(t0)
        d = np.diff(row)
        print("it %d half %d  start %8d | " % (it, h, row[0]) + "  ".join("%s %6d" % (n, v) for n, v in zip(names[1:], d)))
