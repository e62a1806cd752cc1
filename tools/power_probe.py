"""Socket power and shader clock while one kernel runs back to back (is the kernel power-limited?).
   python tools/power_probe.py [fused|fused_nodata|mfma]   -- polls rocm-smi from a thread while the GPU loops for ~6 s."""
import ctypes, os, re, subprocess, sys, threading, time
HERE = os.path.dirname(os.path.abspath(__file__)); ROOT = os.path.dirname(HERE); sys.path.insert(0, ROOT)
import torch
which = sys.argv[1] if len(sys.argv) > 1 else "fused"
samples, stop = [], False


def poll():
    while not stop:
        o = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True).stdout
        p = re.search(r"Package Power \(W\): ([\d.]+)", o); c = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", o)
        samples.append((time.time(), float(p.group(1)) if p else -1, int(c.group(1)) if c else -1))


M, Co, Ci = 262144, 128, 128
dev = "cuda"
if which.startswith("fused"):
    L = ctypes.CDLL(os.path.join(HERE, "libp2c_trace_nodata.so" if which == "fused_nodata" else "libp2c_trace.so"))
    vp = ctypes.c_void_p
    dZ = torch.randn(M, Co, device=dev); Y = torch.randn(M, Co, device=dev); X = torch.randn(M, Ci, device=dev); W = torch.randn(Co, Ci, device=dev) * .1
    coef = torch.rand(5, Co, device=dev); sc = torch.rand(Ci, device=dev) + .5; sh = torch.randn(Ci, device=dev) * .1
    dX = torch.empty(M, Ci, device=dev); dW8 = torch.zeros(8, Co, Ci, device=dev); pstat = torch.rand(4, Ci, device=dev) + .5
    parts = torch.zeros(64, 2, Ci, device=dev, dtype=torch.float64); st = torch.cuda.current_stream().cuda_stream

    def run():
        rc = L.p2c_linear_bwd_fused_f32(vp(dZ.data_ptr()), Co, vp(Y.data_ptr()), Co, 1, vp(coef.data_ptr()), None, 0, vp(X.data_ptr()), Ci, 1, vp(sc.data_ptr()),
                                        vp(sh.data_ptr()), vp(W.data_ptr()), Ci, vp(dX.data_ptr()), Ci, vp(dW8.data_ptr()), Ci, ctypes.c_longlong(Co * Ci), None,
                                        vp(pstat.data_ptr()), vp(parts.data_ptr()), M, Co, Ci, vp(st))
        assert rc == 0, rc
else:
    a = torch.randn(8192, 8192, device=dev); b = torch.randn(8192, 8192, device=dev)

    def run():
        torch.mm(a, b)
for _ in range(5):
    run()
torch.cuda.synchronize()
th = threading.Thread(target=poll); th.start()
t0 = time.time(); n = 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
while time.time() - t0 < 6.0:
    for _ in range(200):
        run()
    n += 200
    torch.cuda.synchronize()
e1.record(); torch.cuda.synchronize()
stop = True; th.join()
print("%s: %d launches, %.1f us per launch" % (which, n, e0.elapsed_time(e1) * 1e3 / n))
for t, p, c in samples:
    print("  t=%.1fs  power %.0f W  sclk %d MHz" % (t - t0, p, c))
