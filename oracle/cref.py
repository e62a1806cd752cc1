"""ctypes front-end of oracle/liboracle.so (TEST INFRASTRUCTURE -- see oracle/p2c_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "p2c_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
    return _LIB


def _f32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


def fps(xyz, start, npoint):
    """xyz (B,N,3) f32, start (B,) int -> (B,npoint) int64.  pointnet_util.py:63-84."""
    xyz = _f32(xyz)
    B, N, _ = xyz.shape
    start = np.ascontiguousarray(np.asarray(start, dtype=np.int64))
    out = np.empty((B, npoint), dtype=np.int32)
    lib().orc_fps(_p(xyz, ctypes.c_float), B, N, _p(start, ctypes.c_int64), int(npoint), _p(out, ctypes.c_int32))
    return out.astype(np.int64)


def square_distance(src, dst):
    """src (B,S,3), dst (B,N,3) -> (B,S,N) f32.  pointnet_util.py:19-40."""
    src, dst = _f32(src), _f32(dst)
    B, S, _ = src.shape
    N = dst.shape[1]
    out = np.empty((B, S, N), dtype=np.float32)
    lib().orc_square_distance(_p(src, ctypes.c_float), _p(dst, ctypes.c_float), B, S, N, _p(out, ctypes.c_float))
    return out


def ball_query(radius, nsample, xyz, new_xyz):
    """pointnet_util.py:87-107 -> (B,S,nsample) int64."""
    xyz, new_xyz = _f32(xyz), _f32(new_xyz)
    B, N, _ = xyz.shape
    S = new_xyz.shape[1]
    out = np.empty((B, S, nsample), dtype=np.int32)
    r2 = ctypes.c_float(np.float32(radius ** 2))
    lib().orc_ball_query(_p(xyz, ctypes.c_float), _p(new_xyz, ctypes.c_float), B, N, S, r2, int(nsample),
                         _p(out, ctypes.c_int32))
    return out.astype(np.int64)


def three_nn(xyz1, xyz2):
    """pointnet_util.py:301-303 -> (dist (B,N,3) f32, idx (B,N,3) int64)."""
    xyz1, xyz2 = _f32(xyz1), _f32(xyz2)
    B, N, _ = xyz1.shape
    S = xyz2.shape[1]
    idx = np.empty((B, N, 3), dtype=np.int32)
    dist = np.empty((B, N, 3), dtype=np.float32)
    lib().orc_three_nn(_p(xyz1, ctypes.c_float), _p(xyz2, ctypes.c_float), B, N, S, _p(idx, ctypes.c_int32),
                       _p(dist, ctypes.c_float))
    return dist, idx.astype(np.int64)


def lsa_max(cost):
    """Row->column assignment maximising sum(cost); cost (nr,nc) with nr<=nc.  losses.py:43."""
    c = np.ascontiguousarray(-np.asarray(cost, dtype=np.float32).astype(np.float64))
    nr, nc = c.shape
    out = np.empty((nr,), dtype=np.int32)
    rc = lib().orc_lsa_min(_p(c, ctypes.c_double), nr, nc, _p(out, ctypes.c_int32))
    if rc != 0:
        raise ValueError("orc_lsa_min failed rc=%d" % rc)
    return out.astype(np.int64)
