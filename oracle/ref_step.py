"""CPU training step of the oracle (TEST INFRASTRUCTURE; the 'port' timed as bench.py's cpu_baseline).

Literal stock-torch op sequence of the reference step (train_Point2Cyl_without_sketch.py:244-369) with
flags --pred_seg --pred_normal --pred_bb: backbone forward (Python FPS loop, full sorts, materialised
activations), Hungarian on the host, losses, backward, Adam."""
import time

import torch
import torch.nn.functional as F

from . import ref_torch as R


class CpuStepper:
    def __init__(self, K=8, seed=0, geom="torch"):
        self.K, self.geom = K, geom
        self.sd = R.make_state_dict((3, 2 * K), seed=seed)
        self.params = [v.requires_grad_(True) for k, v in self.sd.items()
                       if v.dtype.is_floating_point and "running" not in k]
        self.opt = torch.optim.Adam(self.params, lr=1e-3)

    def step(self, pcs, normals, seg, bb):
        B, N, _ = pcs.shape
        starts = [torch.randint(0, N, (B,)), torch.randint(0, 512, (B,))]
        mask = (torch.rand(B, N, 128) < 0.5).float()
        X, W_raw = R.backbone_forward(self.sd, pcs, starts, mask, training=True, momentum=0.5, geom=self.geom)
        X = F.normalize(X, p=2, dim=2, eps=1e-12)
        W2 = torch.softmax(W_raw, 2)
        W = W2[:, :, 0::2] + W2[:, :, 1::2]
        total, nl, ml, match, msk = R.compute_all_losses(W, seg, X, normals, 1.0, 1.0)
        total = total + R.bb_loss(W, W_raw, match, msk, bb, self.K)
        self.opt.zero_grad()
        total.backward()
        self.opt.step()
        return float(total)


def time_cpu_baseline(batch, steps=1, threads=None, budget_s=None):
    """Returns (points_per_second, seconds_per_step, threads, steps_run).  With budget_s the first step sizes the sample:
    further steps run until about budget_s seconds of CPU work are spent (at most 12 steps)."""
    if threads:
        torch.set_num_threads(threads)
    pcs, normals, seg, bb = batch
    st = CpuStepper()
    t0 = time.perf_counter()
    st.step(pcs, normals, seg, bb)
    first = time.perf_counter() - t0
    n = steps
    if budget_s:
        n = max(1, min(12, int(budget_s / max(first, 1e-3))))
    for _ in range(n - 1):
        st.step(pcs, normals, seg, bb)
    dt = (time.perf_counter() - t0) / n
    return pcs.shape[0] * pcs.shape[1] / dt, dt, torch.get_num_threads(), n


def full_loss_step(sd, batch, starts, dropout_mask, K=8, momentum=0.5, pred_extrusion=True, pred_center=True, norm_eig=False,
                   weights=(1.0, 1.0, 1.0, 1.0, 1.0), lr=1e-3, dtype=torch.float32, forced_match=None):
    """ONE step of the reference trainer with the full loss set (train_Point2Cyl_without_sketch.py:244-369 with --pred_seg --pred_normal
    --pred_bb [--pred_extrusion] [--pred_center]) in `dtype`, on a copy of `sd`: backbone forward in train mode on the given FPS
    starts / dropout mask, the five loss terms, backward, one Adam update (lr, torch defaults otherwise).
    batch = (pcs, normals, inst, bb, axes, centers) on the CPU.  weights = (seg, normal, bb, extrusion, center) multipliers.
    Returns dict(losses..., match, mask, heads [X_head, W_raw], grads {name: tensor}, params_after {name: tensor}, buffers {name: tensor}).
    forced_match = (match, mask): use this assignment instead of solving it (a float64 twin of an fp32 run whose soft-IoU costs are
    within rounding of a tie - random initialisation - must stay the same problem).
    The float64 run is the yardstick the GPU tests measure fp32 implementations against (geometry stays pinned to the fp32 indices by
    the oracle's C samplers, which take fp32 coordinates)."""
    old = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        cast = lambda v: v.clone().to(dtype) if v.dtype.is_floating_point else v.clone()
        sd = {k: cast(v.detach()) for k, v in sd.items()}
        names = [k for k, v in sd.items() if v.dtype.is_floating_point and "running" not in k]
        for k in names:
            sd[k].requires_grad_(True)
        pcs, normals, seg, bb, axes, centers = [cast(x) for x in batch]
        w_seg, w_normal, w_bb, w_ext, w_cen = weights
        (X_head, W_raw), aux = R.backbone_forward(sd, pcs, starts, None if dropout_mask is None else dropout_mask.to(dtype), training=True,
                                                  momentum=momentum, geom="c", return_aux=True)
        X = F.normalize(X_head, p=2, dim=2, eps=1e-12)                                   # :247
        W2 = torch.softmax(W_raw, 2)                                                    # :254
        Wb, Wc = W2[:, :, 0::2], W2[:, :, 1::2]
        W = Wb + Wc                                                                     # :265
        if forced_match is None:
            total, nl, ml, match, msk = R.compute_all_losses(W, seg, X, normals, w_normal, w_seg)   # :280
        else:                                                                           # losses.py:317-351 with the assignment given
            match, msk = forced_match
            nl = R.compute_normal_loss(X, normals).mean()
            ml = R.reduce_mean_masked_instance(R.compute_miou_loss(W, seg, match), R.get_mask_gt(seg, K)).mean()
            total = w_seg * ml + w_normal * nl
        bbl = R.bb_loss(W, W_raw, match, msk, bb, K)                                    # :283-307
        total = total + w_bb * bbl
        mask_gt = R.get_mask_gt(seg, K)
        zero = torch.zeros(())
        ext = cen = zero
        if pred_extrusion:                                                              # :319-332
            E = R.estimate_extrusion_axis(X, R.reorder(Wb, match), R.reorder(Wc, match), bb, seg, normalize=norm_eig).to(dtype)
            ext = R.reduce_mean_masked_instance(R.compute_normal_loss(E, axes, angle_diff=False, collapse=False), mask_gt).mean() * w_ext
        total = total + ext
        if pred_center:                                                                 # :342-353
            c = R.estimate_extrusion_centers(R.reorder(W, match), pcs)
            cen = R.reduce_mean_masked_instance(torch.square(c - centers).sum(-1), mask_gt).mean() * w_cen
        total = total + cen
        params = [sd[k] for k in names]
        opt = torch.optim.Adam(params, lr=lr)
        opt.zero_grad()
        total.backward()
        grads = {k: sd[k].grad.detach().clone() for k in names}
        opt.step()
        return dict(total=total.item(), normal=nl.item(), miou=ml.item(), bb=bbl.item(), ext=ext.item(), center=cen.item(),
                    match=match, mask=msk, heads=[X_head.detach(), W_raw.detach()], aux=aux, grads=grads,
                    params_after={k: sd[k].detach() for k in names},
                    buffers={k: v.detach() for k, v in sd.items() if k not in names})
    finally:
        torch.set_default_dtype(old)
