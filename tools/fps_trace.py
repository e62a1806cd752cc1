"""Phase stamps of fps_pk_kernel (csrc/geom.hip built with -DP2C_FPS_TRACE into a throw-away library).
   Build here: python tools/fps_trace.py --build     Run on the GPU box: python tools/fps_trace.py"""
import ctypes, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__)); ROOT = os.path.dirname(HERE)
LIB = os.path.join(HERE, "libp2c_fps_trace.so")
if "--build" in sys.argv:
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-ffp-contract=off", "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops", "-DP2C_FPS_TRACE", "-shared", "-o", LIB,
                           os.path.join(ROOT, "point2cyl_amd", "csrc", "geom.hip")])
    print(LIB); sys.exit(0)
sys.path.insert(0, ROOT)
import numpy as np, torch
from point2cyl_amd import synth
B, N, S = 32, 8192, 512
xyz = synth.make_batch(B, N, 8, seed=1234)[0].float().cuda().contiguous()
start = torch.randint(0, N, (B,)).cuda()
idx = torch.empty(B, S, dtype=torch.int32, device="cuda"); nx = torch.empty(B, S, 3, device="cuda")
L = ctypes.CDLL(LIB)
vp, ci = ctypes.c_void_p, ctypes.c_int
L.p2c_fps_f32.argtypes = [vp, ci, ci, vp, ci, vp, vp, vp]
def run():
    assert L.p2c_fps_f32(xyz.data_ptr(), B, N, start.data_ptr(), S, idx.data_ptr(), nx.data_ptr(), None) == 0
for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record(); torch.cuda.synchronize()
print("fps (with stamps): %.1f us = %.3f us / iteration" % (e0.elapsed_time(e1) * 1e3, e0.elapsed_time(e1) * 1e3 / S))
st = np.zeros((2, 4, 8), dtype=np.uint64)
assert L.p2c_fps_trace_read(st.ctypes.data_as(vp)) == 0
st = st.astype(np.int64)
names = ["update+best", "wave max", "masks/src/slot/readlanes", "lds write", "barrier", "lds read+winner"]
for w, tag in ((0, "wave 0"), (1, "last wave")):
    for i in range(4):
        d = np.diff(st[w, i, :7])
        nxt = (st[w, i + 1, 0] - st[w, i, 6]) if i < 3 else -1
        print("%s it %d: " % (tag, 100 + i) + " | ".join("%s %d" % (n, v) for n, v in zip(names, d)) + " | total %d | to next top %d" % (st[w, i, 6] - st[w, i, 0], nxt))
print("iteration period (wave 0): %s cycles" % np.diff(st[0, :, 0]))
