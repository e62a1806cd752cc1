"""Fall-through for the drop-in modules: a drop-in `losses.py` / `data_utils.py` / `global_variables.py` SHADOWS the reference module of the same
name, so every name it does not override has to keep coming from the shadowed module - eval.py --is_visu calls
`visualize_segmentation_pc` / `visualize_segmentation_pc_bb_v2` (eval.py:659-664, data_utils.py:1744, :1874) through `from data_utils import *`,
and star-imports also carry the shadowed module's own imports (np, json, h5py, ...).

reexport(name, file, namespace): find the next `<name>.py` on sys.path after the drop-in's own directory, execute it under a private module
name and copy every public name the drop-in has not defined into `namespace`.  No shadowed module on the path (our own trainers, the
tests) -> nothing to do.  A shadowed module that fails to import (a dependency of its out-of-scope helpers is missing) -> one warning;
the hot-path names still work."""
import importlib.util
import os
import sys
import warnings


def find_shadowed(name, this_file):
    here = os.path.dirname(os.path.abspath(this_file))
    seen_self = False
    for entry in sys.path:
        d = os.path.abspath(entry or os.getcwd())
        if d == here:
            seen_self = True
            continue
        cand = os.path.join(d, name + ".py")
        if os.path.isfile(cand) and os.path.abspath(cand) != os.path.abspath(this_file):
            return cand, seen_self
    return None, seen_self


def reexport(name, this_file, namespace):
    """-> list of the names taken from the shadowed module (empty when there is none)."""
    path, _ = find_shadowed(name, this_file)
    if path is None:
        return []
    private = "_p2c_shadowed_" + name
    mod = sys.modules.get(private)
    if mod is None:
        spec = importlib.util.spec_from_file_location(private, path)
        mod = importlib.util.module_from_spec(spec)
        sys.modules[private] = mod
        try:
            spec.loader.exec_module(mod)
        except Exception as e:      # an out-of-scope dependency of the reference module (trimesh, chamferdist, ...) is missing
            sys.modules.pop(private, None)
            warnings.warn("point2cyl_amd drop-in `%s`: the shadowed module %s could not be imported (%s: %s); only the hot-path names this "
                          "drop-in overrides are available" % (name, path, type(e).__name__, e))
            return []
    public = getattr(mod, "__all__", None) or [k for k in vars(mod) if not k.startswith("_")]
    taken = []
    for k in public:
        if k not in namespace:
            namespace[k] = getattr(mod, k)
            taken.append(k)
    namespace.setdefault("__p2c_shadowed__", path)
    return taken
