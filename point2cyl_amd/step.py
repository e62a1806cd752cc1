"""One training step of Point2Cyl-without-sketch, as the reference trainer defines it
(train_Point2Cyl_without_sketch.py:244-369): backbone forward, head post-processing, the loss set selected
by the --pred_* flags, backward and the optimizer step.  This is the unit `bench.py` times."""
from dataclasses import dataclass

import numpy as np
import torch
import torch.nn.functional as F

from . import fitting, losses


@dataclass
class StepFlags:
    K: int = 8
    pred_seg: bool = True
    pred_normal: bool = True
    pred_bb: bool = True
    pred_extrusion: bool = False
    pred_center: bool = False
    norm_eig: bool = False
    weight_seg: float = 1.0
    weight_normal: float = 1.0
    weight_bb: float = 1.0
    weight_extrusion: float = 1.0
    weight_center: float = 1.0

    def pred_sizes(self):
        """train…:183-195."""
        return [3 if self.pred_normal else 1,
                2 * self.K if (self.pred_seg and self.pred_bb) else (self.K if self.pred_seg else 1)]


def get_batch_norm_decay(global_step, batch_size, bn_decay_step, staircase=True):
    """train…:143-151."""
    p = global_step * batch_size / bn_decay_step
    if staircase:
        p = int(np.floor(p))
    return max(0.5 * (0.5 ** p), 1 - 0.99)


def update_momentum(module, bn_momentum):
    """train…:153-156: every submodule whose name contains 'bn'."""
    for name, m in module.named_modules():
        if "bn" in name:
            m.momentum = bn_momentum


def get_learning_rate(init_lr, global_step, batch_size, decay_step, decay_rate, staircase=True):
    """train…:159-164."""
    p = global_step * batch_size / decay_step
    if staircase:
        p = int(np.floor(p))
    return init_lr * (decay_rate ** p)


_ZEROS = {}
_SEEDS = {}


def backward(out):
    """`out["total"].backward()` without its three launches: when the total is element 0 of the fused loss vector (compute_losses_fused
    without the axis / centre terms) autograd would fill a ones tensor for the root and a zero 4-vector + an index copy for the select;
    here the loss node receives a CONSTANT [1, 0, 0, 0] as its upstream gradient (the kernel reads element 0)."""
    vec = out.get("_total_vec")
    if vec is None:
        out["total"].backward()
        return
    seed = _SEEDS.get(vec.device)
    if seed is None:
        seed = _SEEDS[vec.device] = torch.tensor([1.0, 0.0, 0.0, 0.0], device=vec.device)
    torch.autograd.backward([vec], [seed])


FUSED_FIT_TERMS = True      # the extrusion-axis / centre terms of the full loss set as one launch (ops.fit_terms); a test switches it off


def compute_losses_fused(model, pcs, gt_normals, gt_inst, gt_bb, gt_axes, gt_centers, fl: StepFlags, geom=None):
    """Same result as compute_losses for --pred_seg --pred_normal --pred_bb (K=8), with the head post-processing, the
    Hungarian matching and the three losses (forward + gradient) in csrc/loss.hip instead of ~60 torch launches.
    The optional extrusion-axis / centre terms are added on top from the same head output."""
    from . import ops
    B, N, _ = pcs.shape
    K = fl.K
    heads, sizes = model.forward_heads(pcs, geom) if geom is not None else model.forward_heads(pcs)
    assert sizes == [3, 2 * K] and fl.pred_seg and fl.pred_normal and fl.pred_bb and 1 <= K <= 8
    out4, match, mask = ops.seg_losses(heads, gt_normals, gt_inst, gt_bb, B, N, K, 0, 3, fl.weight_seg, fl.weight_normal, fl.weight_bb)
    total = out4[0]
    zero = _ZEROS.get(pcs.device)
    if zero is None:
        zero = _ZEROS[pcs.device] = torch.zeros((), device=pcs.device)      # a constant: not re-filled every step
    ext_loss = center_loss = zero
    res = dict(normal=out4[1].detach(), miou=out4[2].detach(), bb=out4[3].detach(), match=match, mask=mask, E_AX=None)
    if fl.pred_extrusion or fl.pred_center:
        # normalised normals, softmax, barrel / base split and the reorder by the matching (train...:247-265, :319-325, :342-344) in
        # one kernel forward and one backward (ops.head_post) instead of ~25 torch launches over (B,N,2K) tensors
        X, Wb_re, Wc_re = ops.head_post(heads, match, B, N, K, 0, 3)
        E_AX = cen = None
        if fl.pred_extrusion:
            E_AX = fitting.estimate_extrusion_axis(X, Wb_re, Wc_re, gt_bb, gt_inst, normalize=fl.norm_eig)
            res["E_AX"] = E_AX                                   # (with its history: train_Point2Cyl.py:528 feeds it to the sketch encoder)
        if fl.pred_center:
            cen = fitting.estimate_extrusion_centers(Wb_re + Wc_re, pcs)
        if FUSED_FIT_TERMS:
            # both terms, forward and gradient, in one launch; the matching's mask IS losses.get_mask_gt(gt_inst, K) (k < instances of the cloud)
            terms = ops.fit_terms(E_AX, gt_axes, cen, gt_centers, mask, fl.weight_extrusion, fl.weight_center)
            ext_loss, center_loss = (terms[0] if fl.pred_extrusion else zero), (terms[1] if fl.pred_center else zero)
            total = total + terms.sum()
        else:
            mask_gt = losses.get_mask_gt(gt_inst, K)
            if fl.pred_extrusion:
                ext = losses.compute_normal_loss(E_AX, gt_axes, angle_diff=False, collapse=False)
                ext_loss = losses.reduce_mean_masked_instance(ext, mask_gt).mean() * fl.weight_extrusion
            if fl.pred_center:
                diff = torch.square(cen - gt_centers).sum(dim=-1)
                center_loss = losses.reduce_mean_masked_instance(diff, mask_gt).mean() * fl.weight_center
            total = total + ext_loss + center_loss
    else:
        res["_total_vec"] = out4              # step.backward(res): the loss node gets a constant seed instead of ones / select-backward launches
    res.update(total=total, ext=ext_loss, center=center_loss, heads=heads)
    return res


def fused_loss_applicable(fl: StepFlags):
    return fl.pred_seg and fl.pred_normal and fl.pred_bb and 1 <= fl.K <= 8


def compute_losses(model, pcs, gt_normals, gt_inst, gt_bb, gt_axes, gt_centers, fl: StepFlags, geom=None):
    """Forward + all loss terms -> dict of scalars (tensors) incl. 'total'."""
    B, N, _ = pcs.shape
    K = fl.K
    dev = pcs.device
    if geom is not None:
        hh, _sz = model.forward_heads(pcs, geom)
        hh = hh.view(B, N, -1)
        X, W_raw = hh[:, :, 0:_sz[0]], hh[:, :, _sz[0]:_sz[0] + _sz[1]]
    else:
        X, W_raw = model(pcs)                                                   # :244
    X_head = X
    if fl.pred_normal:
        X = F.normalize(X, p=2, dim=2, eps=1e-12)                               # :247
    else:
        X = torch.zeros(B, N, 3, device=dev)
    W_barrel = W_base = W_barrel_bb = W_base_bb = None
    if fl.pred_seg and fl.pred_bb:
        W_2K = torch.softmax(W_raw, dim=2)                                      # :254
        W_barrel, W_base = W_2K[:, :, 0::2], W_2K[:, :, 1::2]
        W_barrel_bb, W_base_bb = W_raw[:, :, 0::2], W_raw[:, :, 1::2]
        W = W_barrel + W_base                                                   # :265
    elif fl.pred_seg:
        W = torch.softmax(W_raw, dim=2)
    else:
        W = torch.zeros(B, N, K, device=dev)
    w_n = fl.weight_normal if fl.pred_normal else 0.0
    w_s = fl.weight_seg if fl.pred_seg else 0.0
    total, normal_loss, miou_loss, match, mask = losses.compute_all_losses(
        pcs, W, gt_inst, X, gt_normals, w_n, w_s, return_match_indices=True)   # :280
    if fl.pred_bb:
        bb_loss = losses.compute_bb_loss(W, W_barrel_bb, W_base_bb, match, mask, gt_bb)   # :283-307
    else:
        bb_loss = torch.zeros((), device=dev)
    total = total + (fl.weight_bb if fl.pred_bb else 0.0) * bb_loss
    mask_gt = losses.get_mask_gt(gt_inst, K)
    E_AX = None
    if fl.pred_normal and fl.pred_bb and fl.pred_extrusion:                     # :319-332
        Wb_r, Wc_r = losses._reorder(W_barrel, match), losses._reorder(W_base, match)
        E_AX = fitting.estimate_extrusion_axis(X, Wb_r, Wc_r, gt_bb, gt_inst, normalize=fl.norm_eig)
        ext = losses.compute_normal_loss(E_AX, gt_axes, angle_diff=False, collapse=False)
        ext_loss = losses.reduce_mean_masked_instance(ext, mask_gt).mean() * fl.weight_extrusion
    else:
        ext_loss = torch.zeros((), device=dev)
    total = total + ext_loss
    if fl.pred_center:                                                          # :342-353
        cen = fitting.estimate_extrusion_centers(losses._reorder(W, match), pcs)
        diff = torch.square(cen - gt_centers).sum(dim=-1)
        center_loss = losses.reduce_mean_masked_instance(diff, mask_gt).mean() * fl.weight_center
    else:
        center_loss = torch.zeros((), device=dev)
    total = total + center_loss
    return dict(total=total, normal=normal_loss, miou=miou_loss, bb=bb_loss, ext=ext_loss, center=center_loss,
                match=match, mask=mask, X=X, W=W, W_raw=W_raw, X_head=X_head, E_AX=E_AX)


def train_step(model, optimizer, batch, fl: StepFlags, sync_grads=None, fused=False):
    """batch = (pcs, normals, inst, bb, axes, centers) already on the device.  Returns the loss dict.
    `sync_grads`, if given, is called between backward and optimizer.step (data-parallel all-reduce).
    fused=True evaluates the losses with csrc/loss.hip when the flag set allows it."""
    out = (compute_losses_fused if (fused and fused_loss_applicable(fl)) else compute_losses)(model, *batch, fl)
    optimizer.zero_grad(set_to_none=True)
    backward(out)                                                               # :368
    if sync_grads is not None:
        sync_grads()
    optimizer.step()                                                            # :369
    return out
