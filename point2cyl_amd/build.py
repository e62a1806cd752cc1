"""Build libp2c_hip.so (gfx950 only) in-tree with hipcc.  No GPU is needed to compile.

    python -m point2cyl_amd.build [--force]

The shared object lands next to this file so that it travels with a snapshot of the repo; it is
git-ignored (sources only in history).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(HERE, "libp2c_hip.so")
SOURCES = ["metrics.hip", "geom.hip", "gather.hip", "gemm.hip", "gemm_big.hip", "fwd_pp.hip", "fwd_pp3.hip", "bwd_fused.hip", "bwd_fused3.hip", "bwd_pool.hip", "heads.hip", "bn.hip", "fit.hip", "assign.hip", "loss.hip", "softplus.hip"]
# -ffp-contract=off: geom.hip reproduces the reference's rounding order (explicit fmaf only)
# -target-feature -packed-fp32-ops: NO v_pk_{add,mul,fma}_f32 anywhere (the host pass prints "not a recognized feature", harmless).  Round 6
#   found farthest point sampling picking wrong points in 3 - 54 % of the HIP-graph replays in which it ran on the forked stream UNDER the
#   MFMA kernels (never alone, never in the parity tests): the low half of a v_pk_add_f32 result, last 16 lanes of the wave, read stale by a
#   v_min_i32 two issue slots later - the one wait state the compiler leaves between a packed producer and its consumer is not enough there
#   when the wave has the SIMD's vector issue nearly to itself (16 / 8 / 4 waves per cloud: 0 / 7 / 54 % of replays; tools/stress_prefetch.py,
#   profiles/r06_fps_packed_hazard.log).  Without packed fp32 the same stress shows 0 of 1500 in every shape, and nothing gets slower: a
#   v_fma_f32 issues in 2.5 clocks against 4.2 for the packed form (tools/ubench/valu_rate.hip) - FPS 443 -> 382 us, step -0.01 ms, the
#   fitting pass +1.7 %.
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-ffp-contract=off", "-Wno-unused-value",
         "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]


def source_hash():
    """sha1 over the kernel sources (csrc/*.hip, csrc/*.h, include/p2c_hip.h): stamps profiles so that a number measured on other
    kernels than the ones in the tree is recognisable (bench.py: roofline.traffic_stale)."""
    import hashlib
    h = hashlib.sha1()
    files = sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))) 
    for f in files:
        h.update(f.encode())
        h.update(open(os.path.join(CSRC, f), "rb").read())
    h.update(open(os.path.join(HERE, "..", "include", "p2c_hip.h"), "rb").read())
    h.update(" ".join(FLAGS).encode())           # (code generation options are part of what was measured)
    return h.hexdigest()[:16]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _newer(a, b):
    return not os.path.exists(b) or os.path.getmtime(a) > os.path.getmtime(b)


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    deps = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "fwd_pp.h"), os.path.join(CSRC, "lsa.h"), os.path.join(CSRC, "eigh3.h"), os.path.join(HERE, "..", "include", "p2c_hip.h")]
    objs, dirty = [], False
    procs = []
    stamp = os.path.join(OBJ, "flags.txt")                 # objects built with other options are stale whatever their age
    if not os.path.exists(stamp) or open(stamp).read() != " ".join(FLAGS):
        force = True
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        if not os.path.exists(s):
            continue
        o = os.path.join(OBJ, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _newer(s, o) or any(_newer(d, o) for d in deps):
            cmd = [_hipcc()] + FLAGS + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd))
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
            dirty = True
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError("hipcc failed on %s" % src)
    if dirty or not os.path.exists(LIB):
        cmd = [_hipcc(), "-shared", "-fPIC", "--offload-arch=gfx950", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        with open(stamp, "w") as f:
            f.write(" ".join(FLAGS))
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
