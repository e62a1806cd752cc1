"""Reader for the reference's H5 dataset (schema written at utils.py:1174-1188, item logic of
AutodeskDataset_h5.__getitem__ with center=True, dataloader.py:69-96).  Used only when h5py and data/<split>.h5
exist; returns the same 9-tuple as synth.SyntheticExtrusionDataset."""
import numpy as np
import torch


class AutodeskH5(torch.utils.data.Dataset):
    def __init__(self, path, num_point, K):
        import h5py
        with h5py.File(path, "r") as f:
            self.pcs = f["point_cloud"][:]
            self.normals = f["normals"][:]
            self.labels = f["extrusion_labels"][:]
            self.bb = f["base_barrel_labels"][:]
            self.axes = f["extrusion_axes"][:]
            self.dist = f["extrusion_distances"][:]
            self.centers = f["extrusion_centers"][:]
        self.num_point, self.K = num_point, K

    def __len__(self):
        return self.pcs.shape[0]

    def __getitem__(self, i):
        if self.num_point is None:             # whole item (the device-resident trainer draws the subsample on the GPU)
            sel = np.arange(self.pcs.shape[1])
        else:
            sel = torch.randperm(self.pcs.shape[1])[: self.num_point].numpy()      # dataloader.py:71-77
        lab = self.labels[i][sel]
        K = self.K
        return (self.pcs[i][sel].astype(np.float32), self.normals[i][sel].astype(np.float32), lab.astype(np.int64),
                self.bb[i][sel].astype(np.int64), self.axes[i][lab].astype(np.float32), self.dist[i][lab].astype(np.float32),
                self.axes[i][:K].astype(np.float32), self.dist[i][:K].astype(np.float32), self.centers[i][:K].astype(np.float32))
