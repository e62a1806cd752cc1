"""Real HDF5 fixtures for point2cyl_amd/h5min.py and h5data.py, written by the real library (h5py 3.3 / HDF5 1.10.6 under
/opt/conda/bin/python3.9 of the build image - the interpreter the package itself runs on has no h5py) with the calls the reference's
preprocessing makes (utils.py:1174-1188 `save_h5`-style, :1251-1268 for the sketch files): `h5py.File(fname)` default format,
`create_dataset(name, data=..., compression='gzip', dtype=...)`.

    /opt/conda/bin/python3.9 oracle/make_golden_h5.py        # -> tests/golden/autodesk_schema_small.h5, autodesk_schema_sketches.h5,
                                                             #    h5min_layouts.h5 and their arrays as .npz for the comparison

The arrays are synthetic (seeded numpy); what the fixtures pin is the FILE FORMAT the authors' release uses."""
import os

import h5py
import numpy as np

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
rng = np.random.default_rng(20260929)


def schema_arrays(n, P, K, sketches=None):
    d = dict(point_cloud=rng.standard_normal((n, P, 3)).astype("float32"), normals=rng.standard_normal((n, P, 3)).astype("float32"),
             extrusion_labels=rng.integers(0, K, (n, P)), base_barrel_labels=rng.integers(0, 2, (n, P)), n_instances=rng.integers(1, K + 1, (n,)),
             extrusion_axes=rng.standard_normal((n, K, 3)).astype("float32"), extrusion_distances=rng.random((n, K)).astype("float32"),
             extrusion_operation=rng.integers(0, 3, (n, P)), extrusion_centers=rng.standard_normal((n, K, 3)).astype("float32"),
             extrusion_extents=rng.random((n, K, 2)).astype("float32"))
    if sketches:
        d["sketches"] = rng.standard_normal((n, K, sketches, 4)).astype("float32")
        d["sketches_norms"] = rng.random((n, K)).astype("float32")
    return d


def write_schema(fname, d):
    """The create_dataset calls of utils.py:1174-1188 / :1251-1268, argument for argument."""
    if os.path.exists(fname):
        os.remove(fname)
    fout = h5py.File(fname, "w")
    for name in ("point_cloud", "normals"):
        fout.create_dataset(name, data=d[name], compression="gzip", dtype="float32")
    for name in ("extrusion_labels", "base_barrel_labels", "n_instances"):
        fout.create_dataset(name, data=d[name], compression="gzip", dtype="int")
    for name in ("extrusion_axes", "extrusion_distances"):
        fout.create_dataset(name, data=d[name], compression="gzip", dtype="float32")
    fout.create_dataset("extrusion_operation", data=d["extrusion_operation"], compression="gzip", dtype="int")
    fout.create_dataset("extrusion_centers", data=d["extrusion_centers"], compression="gzip", dtype="float32")
    fout.create_dataset("extrusion_extents", data=d["extrusion_extents"], compression="gzip", dtype="float32")
    if "sketches" in d:
        fout.create_dataset("sketches", data=d["sketches"], compression="gzip", dtype="float32")
        fout.create_dataset("sketches_norms", data=d["sketches_norms"], compression="gzip", dtype="float32")
    fout.close()


def main():
    small = schema_arrays(6, 320, 8)
    write_schema(os.path.join(OUT, "autodesk_schema_small.h5"), small)
    np.savez(os.path.join(OUT, "autodesk_schema_small_arrays.npz"), **small)
    sk = schema_arrays(4, 300, 8, sketches=96)
    write_schema(os.path.join(OUT, "autodesk_schema_sketches.h5"), sk)
    np.savez(os.path.join(OUT, "autodesk_schema_sketches_arrays.npz"), **sk)
    # other layouts the reader claims: contiguous, compact-sized, many chunks (a two-level chunk B-tree), shuffle + gzip + fletcher32,
    # ragged edge chunks, enough datasets for a multi-node group B-tree, float64 / int32 / uint8 element types
    lay = dict(contiguous_f32=rng.standard_normal((5, 7)).astype("float32"), scalarish=np.arange(3, dtype="int64"),
               many_chunks=rng.integers(-5, 5, (150, 16, 3)).astype("int64"), shuffled=rng.standard_normal((33, 17)).astype("float32"),
               ragged=rng.standard_normal((10, 11, 3)), small_u8=rng.integers(0, 255, (9, 4)).astype("uint8"),
               i32=rng.integers(-9, 9, (12,)).astype("int32"))
    for i in range(40):
        lay["filler_%02d" % i] = np.full((2,), i, dtype="float32")
    f = os.path.join(OUT, "h5min_layouts.h5")
    if os.path.exists(f):
        os.remove(f)
    fo = h5py.File(f, "w")
    fo.create_dataset("contiguous_f32", data=lay["contiguous_f32"])
    fo.create_dataset("scalarish", data=lay["scalarish"])
    fo.create_dataset("many_chunks", data=lay["many_chunks"], chunks=(2, 16, 3), compression="gzip")
    fo.create_dataset("shuffled", data=lay["shuffled"], chunks=(8, 8), compression="gzip", shuffle=True, fletcher32=True)
    fo.create_dataset("ragged", data=lay["ragged"], chunks=(4, 4, 2), compression="gzip", compression_opts=9)
    fo.create_dataset("small_u8", data=lay["small_u8"], chunks=(4, 4))
    fo.create_dataset("i32", data=lay["i32"])
    for i in range(40):
        fo.create_dataset("filler_%02d" % i, data=lay["filler_%02d" % i])
    fo.close()
    np.savez(os.path.join(OUT, "h5min_layouts_arrays.npz"), **lay)
    for n in ("autodesk_schema_small.h5", "autodesk_schema_sketches.h5", "h5min_layouts.h5"):
        print(n, os.path.getsize(os.path.join(OUT, n)))


if __name__ == "__main__":
    main()
