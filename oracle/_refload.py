"""Loader for the upstream Point2Cyl reference (TEST INFRASTRUCTURE, this container only).

/root/reference is read-only and does NOT exist on the GPU box.  This module is used
only by oracle/make_golden.py and by CPU tests that are skipped when the reference is
absent.  It never copies reference source: it puts the reference on sys.path, stubs the
third-party modules the reference imports at module-import time but that are not
installed here, and shims the removed ``torch.symeig``.
"""
import importlib
import os
import sys
import types

REF_ROOT = os.environ.get("P2C_REFERENCE", "/root/reference")


def available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "models"))


def _stub(name, **attrs):
    if name in sys.modules:
        return sys.modules[name]
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def load():
    """Return dict of the reference modules on the hot path (imported, not copied)."""
    if not available():
        raise RuntimeError("reference tree not present at %s" % REF_ROOT)
    import torch

    for name in ("h5py", "trimesh", "torchgeometry", "plyfile", "skimage"):
        try:
            importlib.import_module(name)
        except Exception:
            _stub(name)
    try:
        importlib.import_module("skimage.measure")
    except Exception:
        sk = sys.modules["skimage"]
        sk.measure = _stub("skimage.measure")
    try:
        importlib.import_module("chamferdist")
    except Exception:
        class ChamferDistance:  # import-time only (losses.py:14-15); never called on the hot path
            def __call__(self, *a, **k):
                raise RuntimeError("chamferdist is not installed")
        _stub("chamferdist", ChamferDistance=ChamferDistance)

    if not hasattr(torch, "symeig") or _symeig_removed(torch):
        def symeig(A, eigenvectors=True, upper=True):
            return torch.linalg.eigh(A, UPLO="U" if upper else "L")
        torch.symeig = symeig

    # make sure OUR drop-in modules of the same names are not shadowing the reference
    for name in ("pointnet_util", "pointnet_extrusion", "losses", "data_utils", "global_variables",
                 "utils", "models", "models.pointnet_util", "models.pointnet_extrusion"):
        sys.modules.pop(name, None)
    saved = list(sys.path)
    sys.path[:0] = [REF_ROOT, os.path.join(REF_ROOT, "models")]
    try:
        mods = {}
        mods["pointnet_util"] = importlib.import_module("models.pointnet_util")
        mods["pointnet_extrusion"] = importlib.import_module("pointnet_extrusion")
        mods["losses"] = importlib.import_module("losses")
        mods["data_utils"] = importlib.import_module("data_utils")
    finally:
        sys.path[:] = saved
        # leave the reference modules reachable only through the returned dict
        for name in ("pointnet_util", "pointnet_extrusion", "losses", "data_utils", "global_variables",
                     "utils", "models", "models.pointnet_util", "models.pointnet_extrusion"):
            sys.modules.pop(name, None)
    return mods


def _symeig_removed(torch):
    try:
        torch.symeig(torch.eye(2))
        return False
    except Exception:
        return True
