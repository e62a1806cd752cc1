"""Readers for the reference's pre-processed datasets (host I/O; not a kernel target).

`AutodeskH5`         mirrors dataloader.py:15-126  (`AutodeskDataset_h5`: train_Point2Cyl_without_sketch.py, eval.py)
`AutodeskH5Sketches` mirrors dataloader.py:129-296 (`AutodeskDataset_h5_sketches`: train_Point2Cyl.py:215, eval.py:152)
on the file schema utils.py:1174-1188 / :1251-1268 writes (float32 `point_cloud` (n,P,3), `normals`, int `extrusion_labels` (n,P),
`base_barrel_labels`, `n_instances`, float32 `extrusion_axes` (n,K,3), `extrusion_distances` (n,K), optional `extrusion_operation`,
`extrusion_centers` (n,K,3), `extrusion_extents`, and for the sketch files `sketches` (n,K,S,4) = [point | normal] and `sketches_norms`).

Same constructor flags, same item tuples in the same order for every flag combination, same RNG draws in the same order
(`torch.randperm(P)` for the cloud, then `torch.randperm(S)` for the sketches), so a seeded reference run and a seeded run here see
the same items.  The arrays come through a 3-line source protocol - anything with `keys()` and `obj[name][:]`:
  * a path ending in .h5 / .hdf5  -> `h5py.File` when h5py is importable, else `point2cyl_amd.h5min.H5File`: a pure-numpy reader of the
                                     format the reference writes (gzip-chunked datasets, utils.py:1174-1188), checked against files made by
                                     the real library (tests/golden/autodesk_schema_*.h5),
  * a path ending in .npz         -> `numpy.load` (the same arrays exported once with numpy),
  * a dict of arrays              -> an in-memory "file" (tests; the fixture made from the imported reference dataloader).
"""
import numpy as np
import torch


def open_arrays(source):
    """-> mapping name -> array-like supporting [:]; caller reads what it needs and drops the handle."""
    if isinstance(source, dict):
        return source
    if hasattr(source, "keys") and hasattr(source, "__getitem__"):
        return source
    path = str(source)
    if path.endswith(".npz"):
        return np.load(path)
    try:
        import h5py
    except ImportError:
        from .h5min import H5File          # the pure-numpy reader of the file format the reference's preprocessing writes (gzip-chunked datasets)
        return H5File(path)
    return h5py.File(path, "r")


def dataset_path(data_dir, split):
    """<data_dir>/<split>.h5 (read through h5py when it is installed, through point2cyl_amd.h5min otherwise), else <data_dir>/<split>.npz
    (the same arrays exported with numpy)."""
    import os
    h5, npz = os.path.join(data_dir, split + ".h5"), os.path.join(data_dir, split + ".npz")
    if os.path.exists(h5):
        return h5
    if os.path.exists(npz):
        return npz
    raise SystemExit("no dataset at %s or %s; use --synthetic N" % (h5, npz))


def _load(source, op, center, extent, sketches):
    """utils.py:1195-1225 (`load_h5`) / :1276-1316 (`load_h5_sk`)."""
    f = open_arrays(source)
    try:
        out = dict(point_cloud=f["point_cloud"][:], normals=f["normals"][:], extrusion_labels=f["extrusion_labels"][:],
                   bb_labels=f["base_barrel_labels"][:], n_instances=f["n_instances"][:] if "n_instances" in f.keys() else None,
                   extrusion_axes=f["extrusion_axes"][:], extrusion_distances=f["extrusion_distances"][:])
        if op:
            out["operations"] = f["extrusion_operation"][:]
        if center:
            out["extrusion_centers"] = f["extrusion_centers"][:]
        if extent:
            out["extrusion_extents"] = f["extrusion_extents"][:]
        if sketches:
            out["sketches"] = f["sketches"][:]
            out["sk_norm_factors"] = f["sketches_norms"][:]
    finally:
        if hasattr(f, "close") and not isinstance(f, dict):
            f.close()
    return out


class _Base(torch.utils.data.Dataset):
    def _common_init(self, d, num_points, max_instances, op, center, extent):
        self.pcs, self.normals = d["point_cloud"], d["normals"]
        self.extrusion_labels, self.bb_labels = d["extrusion_labels"], d["bb_labels"]
        self.extrusion_axes, self.extrusion_distances = d["extrusion_axes"], d["extrusion_distances"]
        self.n_samples = self.pcs.shape[0]
        self.n_instances = d["n_instances"]
        self.npoints, self.K = num_points, max_instances
        self.op, self.center, self.extent = op, center, extent
        self.generator = None      # None: torch's global CPU generator, as the reference; a loader that runs in its own thread sets a private one
        if op:
            self.operations = d["operations"]
        if center:
            self.extrusion_centers = d["extrusion_centers"]
        if extent:
            self.extrusion_extents = d["extrusion_extents"]
        if self.n_instances is not None and int(np.max(self.n_instances)) != self.K:          # dataloader.py:57-59
            print("WARNING. K= " + str(self.K) + ", max_instance in data= " + str(int(np.max(self.n_instances))))

    def __len__(self):
        return self.n_samples

    def _sample_cloud(self, index):
        """dataloader.py:69-85: a fresh permutation per item, the first num_points of it."""
        P = self.pcs.shape[1]
        if self.npoints is None:               # whole item (the device-resident trainer draws its subsample on the GPU)
            sel = torch.arange(P)
        else:
            if P < self.npoints:
                print("ERROR. Sampling more points than point cloud resolution.")       # dataloader.py:72-73 prints and goes on: the item is short
            sel = torch.randperm(P, generator=self.generator)[: self.npoints]
        lab = self.extrusion_labels[index][sel]
        head = (self.pcs[index][sel, :], self.normals[index][sel, :], lab, self.bb_labels[index][sel],
                self.extrusion_axes[index][lab], self.extrusion_distances[index][lab],
                self.extrusion_axes[index][: self.K], self.extrusion_distances[index][: self.K])
        return sel, head


class AutodeskH5(_Base):
    """`AutodeskDataset_h5(filename, num_points, max_instances, op=False, center=False, extent=False)`.  Item = the 8 base fields
    [+ per-point operation] [+ centers (K,3)] [+ extents] in the reference's order (dataloader.py:87-123).
    Defaults as the reference's (center=False: 8 fields); the trainers pass center=True (train_Point2Cyl_without_sketch.py:168): the
    9-tuple synth.SyntheticExtrusionDataset mimics."""

    def __init__(self, source, num_points, max_instances, op=False, center=False, extent=False):
        self._common_init(_load(source, op, center, extent, False), num_points, max_instances, op, center, extent)

    def __getitem__(self, index):
        sel, item = self._sample_cloud(index)
        if self.op:
            item += (self.operations[index][sel],)
        if self.center:
            item += (self.extrusion_centers[index][: self.K],)
        if self.extent:
            item += (self.extrusion_extents[index][: self.K],)
        return item


class AutodeskH5Sketches(_Base):
    """`AutodeskDataset_h5_sketches(filename, num_points, num_sk_points, max_instances, op=False, center=False, with_scale=False,
    extent=False)`.  Item = the 8 base fields [+ operation] [+ centers] + sampled_sketch (K, num_sk_points, 4) [+ norm factors] [+ extents]
    (dataloader.py:215-291).  With op=True the reference indexes the per-point operation labels with the SKETCH permutation
    (`selected_idx` is overwritten at dataloader.py:213 before :227 uses it); kept as it is - a seeded run returns the same item."""

    def __init__(self, source, num_points, num_sk_points, max_instances, op=False, center=False, with_scale=False, extent=False):
        d = _load(source, op, center, extent, True)
        self._common_init(d, num_points, max_instances, op, center, extent)
        self.sketches, self.sk_norm_factors = d["sketches"], d["sk_norm_factors"]
        self.num_sk_points, self.with_scale = num_sk_points, with_scale

    def __getitem__(self, index):
        _, item = self._sample_cloud(index)
        sk_sel = torch.randperm(self.sketches.shape[2], generator=self.generator)[: self.num_sk_points]                  # dataloader.py:211-214
        sampled_sketch = self.sketches[index][:, sk_sel, :]
        if self.op:
            item += (self.operations[index][sk_sel],)
        if self.center:
            item += (self.extrusion_centers[index][: self.K],)
        item += (sampled_sketch,)
        if self.with_scale:
            item += (self.sk_norm_factors[index],)
        if self.extent:
            item += (self.extrusion_extents[index][: self.K],)
        return item
