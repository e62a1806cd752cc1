"""The training step AS THE REFERENCE'S TRAINER COMPOSES IT (train_Point2Cyl_without_sketch.py:244-369), written against the drop-in import
names only - `importlib.import_module("pointnet_extrusion").backbone`, `from losses import ...`, `from data_utils import ...`, the inline
base/barrel block as torch ops, `torch.optim.Adam`, six `.item()` reads for the log line - so that `bench.py --dropin` and the tests can time
and check what a user of the boundary gets without the reference's host-I/O dependencies (h5py, tensorboard).  Nothing here is faster or
smarter than the reference's loop; whatever speed it has comes from behind the import names."""
import importlib
import os
import sys

import torch
import torch.nn.functional as F

_HERE = os.path.dirname(os.path.abspath(__file__))


def import_names():
    """-> (backbone class, losses module, data_utils module) through the reference's import statements (train...:14-23, :180)."""
    for p in (os.path.join(_HERE, "models"), _HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    MODEL = importlib.import_module("pointnet_extrusion")
    return MODEL.backbone, importlib.import_module("losses"), importlib.import_module("data_utils")


class TrainerStep:
    """One object per run: model + torch.optim.Adam + the schedules of train...:143-164, :355-366."""

    def __init__(self, K=8, batch_size=32, pred_seg=True, pred_normal=True, pred_bb=True, pred_extrusion=False, pred_center=False, norm_eig=False,
                 multipliers=(1.0, 1.0, 1.0, 1.0, 1.0), lr=1e-3, device="cuda", optimizer=None):
        backbone, self.L, self.D = import_names()
        self.K, self.B = K, batch_size
        self.flags = (pred_seg, pred_normal, pred_bb, pred_extrusion, pred_center, norm_eig)
        self.mult = multipliers
        sizes = [3 if pred_normal else 1, 2 * K if (pred_seg and pred_bb) else (K if pred_seg else 1)]      # train...:183-195
        self.model = backbone(output_sizes=sizes).to(device).train()
        self.optimizer = optimizer(self.model.parameters()) if optimizer else torch.optim.Adam(self.model.parameters(), lr=lr)   # :204
        self.init_lr, self.global_step = lr, 0
        self.old_bn, self.old_lr = None, lr

    def __call__(self, pcs, gt_normals, gt_inst, gt_bb, gt_axes, gt_centers):
        """-> the six floats of the trainer's log line (total, normal, mIoU, bb, extrusion, centre): SIX device->host reads, like :371-375."""
        L, D, K = self.L, self.D, self.K
        PRED_SEG, PRED_NORMAL, PRED_BB, PRED_EXT, PRED_CENTER, NORM_EIG = self.flags
        w_seg, w_normal, w_bb, w_ext, w_cen = self.mult
        model, opt = self.model, self.optimizer
        B, N, _ = pcs.shape
        dev = pcs.device
        X, W_raw = model(pcs)                                                                                   # :244
        X = F.normalize(X, p=2, dim=2, eps=1e-12) if PRED_NORMAL else torch.zeros(B, N, 3, device=dev)          # :246-250
        if PRED_SEG and PRED_BB:                                                                                # :252-271
            W_2K = torch.softmax(W_raw, dim=2)
            W_barrel, W_barrel_bb = W_2K[:, :, ::2], W_raw[:, :, ::2]
            W_base, W_base_bb = W_2K[:, :, 1::2], W_raw[:, :, 1::2]
            W = W_barrel + W_base
        elif PRED_SEG:
            W = torch.softmax(W_raw, dim=2)
        else:
            W = torch.zeros(B, N, K, device=dev)
        total, l_normal, l_miou, match, mask = L.compute_all_losses(pcs, W, gt_inst, X, gt_normals, w_normal, w_seg,
                                                                    return_match_indices=True)               # :280
        if PRED_BB:                                                                                             # :283-307
            idx = match.unsqueeze(1).expand(B, N, K)
            W_re = torch.gather(W, 2, idx)
            mask = mask.float()
            W_re = torch.where(mask.unsqueeze(1).expand(B, N, K) == 1, W_re, torch.zeros_like(W_re))
            W_sorted, label = torch.sort(torch.softmax(W_re, dim=-1), dim=-1)
            pair = torch.cat((torch.gather(W_barrel_bb, 2, label).unsqueeze(-1), torch.gather(W_base_bb, 2, label).unsqueeze(-1)), dim=-1)
            tgt = gt_bb.unsqueeze(-1).repeat(1, 1, K)
            ce = F.cross_entropy(pair.contiguous().view(B * N * K, -1), tgt.view(B * N * K), reduction="none").view(B, N, K)
            l_bb = torch.mean(torch.mean(torch.sum(ce * W_sorted, dim=-1), dim=-1))
        else:
            l_bb = torch.zeros([B], device=dev)
        l_bb = torch.mean(l_bb)
        total = total + w_bb * l_bb                                                                             # :313-314
        mask_gt = L.get_mask_gt(gt_inst, K)
        if PRED_NORMAL and PRED_BB and PRED_EXT:                                                                # :319-338
            idx = match.unsqueeze(1).expand(B, N, K)
            E_AX = D.estimate_extrusion_axis(X, torch.gather(W_barrel, 2, idx), torch.gather(W_base, 2, idx), gt_bb, gt_inst, normalize=NORM_EIG)
            avg_ext = L.reduce_mean_masked_instance(L.compute_normal_loss(E_AX, gt_axes, angle_diff=False, collapse=False), mask_gt)
        else:
            avg_ext = torch.zeros([B, K], device=dev)
        l_ext = torch.mean(avg_ext) * w_ext
        total = total + l_ext
        if PRED_CENTER:                                                                                         # :342-351
            cen = D.estimate_extrusion_centers(torch.gather(W, 2, match.unsqueeze(1).expand(B, N, K)), pcs)
            avg_cen = L.reduce_mean_masked_instance(torch.square(cen - gt_centers).sum(dim=-1), mask_gt)
        else:
            avg_cen = torch.zeros([B], device=dev)
        l_cen = torch.mean(avg_cen) * w_cen
        total = total + l_cen
        opt.zero_grad()                                                                                         # :355
        bn = max(0.5 * (0.5 ** int(self.global_step * self.B // 200000)), 1 - 0.99)                                # :143-151, :357-360
        if bn != self.old_bn:
            for name, m in model.named_modules():
                if "bn" in name:
                    m.momentum = bn
            self.old_bn = bn
        lr = self.init_lr * (0.7 ** int(self.global_step * self.B // 200000))                                   # :159-164, :362-366
        if lr != self.old_lr:
            for g in opt.param_groups:
                g["lr"] = lr
            self.old_lr = lr
        total.backward()                                                                                        # :368
        opt.step()                                                                                              # :369
        self.global_step += 1
        return (total.item(), l_normal.item(), l_miou.item(), l_bb.item(), l_ext.item(), l_cen.item())          # :371-375
