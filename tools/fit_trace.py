"""Phase stamps of fit_fused_kernel (csrc/fit.hip built with -DP2C_FIT_TRACE into a throw-away library).
   Build here: python tools/fit_trace.py --build     Run on the GPU box: python tools/fit_trace.py"""
import ctypes, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__)); ROOT = os.path.dirname(HERE)
LIB = os.environ.get("P2C_FIT_TRACE_LIB") or os.path.join(HERE, "libp2c_fit_trace.so")
if "--build" in sys.argv:
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-ffp-contract=off", "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops", "-DP2C_FIT_TRACE",
                           "-DP2C_FIT_TRACE_WG=%d" % int(os.environ.get("P2C_FIT_TRACE_WG", "0")), "-shared", "-o", LIB,
                           os.path.join(ROOT, "point2cyl_amd", "csrc", "fit.hip")])
    print(LIB); sys.exit(0)
sys.path.insert(0, ROOT)
import numpy as np, torch
from point2cyl_amd.synth import make_fitting_inputs as make_inputs
B, N, K, S = 1250, 8192, 8, 2048
pcs, X, seg, bb, axes, Wb, Wc, onehot = make_inputs(B, N, K, 4321)
g = torch.Generator().manual_seed(7)
counts = ((seg.unsqueeze(-1) == torch.arange(K)) & (bb == 0).unsqueeze(-1)).sum(1)
ri = torch.randint(0, 1 << 30, (B, K, S), generator=g) % counts.clamp_min(1).unsqueeze(-1)
d = lambda t: t.cuda().contiguous()
pcs, X, seg, bb, Wb, Wc, ri = d(pcs), d(X), d(seg), d(bb), d(Wb), d(Wc), d(ri)
L = ctypes.CDLL(LIB)
vp, ci = ctypes.c_void_p, ctypes.c_int
L.p2c_fit_fused_f32.argtypes = [vp] * 5 + [ci] + [vp] * 2 + [ci] * 4 + [vp] * 8
L.p2c_extents_ws_bytes.restype = ctypes.c_size_t
ax = torch.empty(B, K, 3, device="cuda"); ce = torch.empty(B, K, 3, device="cuda"); cf = torch.empty(B, K, device="cuda")
ex = torch.empty(K, B, 2, device="cuda"); ef = torch.empty(B, K, device="cuda"); ws = torch.empty(B * K * 3 + 16, device="cuda")
HARD = "--hard" in sys.argv        # memberships implied by the labels: Wb = Wc = NULL
def run():
    assert L.p2c_fit_fused_f32(X.data_ptr(), None if HARD else Wb.data_ptr(), None if HARD else Wc.data_ptr(), bb.data_ptr(), seg.data_ptr(), 0, pcs.data_ptr(), ri.data_ptr(), B, N, K, S,
                               ax.data_ptr(), ce.data_ptr(), cf.data_ptr(), ex.data_ptr(), ef.data_ptr(), None, ws.data_ptr(), None) == 0
for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record(); torch.cuda.synchronize()
print("fit_fused%s + finish: %.1f us" % (" (labels-implied memberships)" if HARD else "", e0.elapsed_time(e1) * 1e3))
ts = []
for _ in range(30):
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3)
print("  30 launches one at a time: min %.1f us, median %.1f us" % (min(ts), sorted(ts)[15]))
st = np.zeros(40, dtype=np.uint64)
assert L.p2c_fit_trace_read(st.ctypes.data_as(vp)) == 0
d_ = np.diff(st[:6].astype(np.int64))
print("traced workgroup, shader cycles: stream %d | sums (shuffles + 16 waves) %d | eigen + centroid %d | lists %d | projection %d   (2.4 GHz: total %.1f us)"
      % (d_[0], d_[1], d_[2], d_[3], d_[4], d_.sum() / 2400.0))
s64 = st.astype(np.int64)
if s64[8] and s64[9] and s64[10]:
    print("  inside 'sums': wait for the other waves' streams %d | class sums + LDS writes %d | barrier %d | 16-wave fp64 sums + barrier %d"
          % (s64[8] - s64[1], s64[9] - s64[8], s64[10] - s64[9], s64[2] - s64[10]))

if s64[16]:
    print("  end of stream per wave, cycles after the kernel's start: %s" % [int(v - s64[0]) for v in s64[16:32]])

if s64[11] and s64[13]:
    print("  inside 'lists': keys + counts %d | barrier %d | offsets (3 barriers) %d | placement + barrier %d"
          % (s64[11] - s64[3], s64[12] - s64[11], s64[13] - s64[12], s64[4] - s64[13]))
