"""The implicit-sketch part of one step of the with-sketch trainer (train_Point2Cyl.py:519-672): predicted labels or - --use_gt_im, :566-600 -
the ground-truth ones; projected sketches or - --use_whole_pc - the whole cloud with the (soft / one-hot) segment membership as a fourth
channel (:522-536; the membership keeps its gradient, so the latent and decoder losses reach the backbone through the encoder's input) and,
--use_extrusion_axis_feat, the segment's extrusion axis as channels 5-7 (:528-531, :577-580); angle or L2 latent loss, --with_im_loss.  Composition only - every piece is a kernel path of
this package: fitting.sketch_implicit_projection (csrc/fit.hip), sketch.PointNetEncoder (MLP-stack kernels), implicit.ImplicitNet and
its double backward (csrc/gemm.hip)."""
import torch

from . import fitting, losses
from .implicit import add_latent, decoder_value_and_grad, decoder_value_and_grad_applicable, gradient

USE_DECODER_NODE = True      # the frozen decoder's value + input gradient as ONE autograd node (implicit._DecoderVG); False: composed Functions


def implicit_losses(implicit_net, sk_pnts, sk_normals, nonmnfld_pnts, latent_codes, mask_gt, batch_size, K):
    """train_Point2Cyl.py:610-648 -> (im_loss, mnfld_loss, grad_loss, normals_loss)."""
    if USE_DECODER_NODE and decoder_value_and_grad_applicable(implicit_net):
        # frozen decoder (the trainer's: :352-365): forward, input gradient and their double backward hand-written as one node - the sums of
        # the two gradients every pre-activation receives ride in the products' epilogues (implicit._DecoderVG).  The surface samples and
        # the off-surface samples of a sketch go through it TOGETHER (one evaluation over S + S' points per sketch instead of two: the
        # decoder is the same, half the launches and weight pre-passes), the rows [code | point | pad] written 4-padded by ONE concatenation.
        nb, S1 = sk_pnts.shape[0], sk_pnts.shape[1]
        both = torch.cat([sk_pnts, nonmnfld_pnts], dim=1)                         # (B', S + S', 2)
        d_in = latent_codes.shape[-1] + both.shape[-1]
        a = add_latent(both, latent_codes, pad=True).requires_grad_()
        pred, g = decoder_value_and_grad(implicit_net, a, d_in=d_in)
        pred, g = pred.view(nb, -1, 1), g.view(nb, -1, 2)
        sk_pred, nonmnfld_pred = pred[:, :S1].reshape(-1, 1), pred[:, S1:].reshape(-1, 1)
        mnfld_grad, nonmnfld_grad = g[:, :S1].reshape(batch_size, K, -1, 2), g[:, S1:].reshape(batch_size, K, -1, 2)
    else:
        a = add_latent(sk_pnts, latent_codes).requires_grad_()
        n = add_latent(nonmnfld_pnts, latent_codes).requires_grad_()
        sk_pred, nonmnfld_pred = implicit_net(a), implicit_net(n)
        mnfld_grad = gradient(a, sk_pred).reshape(batch_size, K, -1, 2)
        nonmnfld_grad = gradient(n, nonmnfld_pred).reshape(batch_size, K, -1, 2)
    sk_normals = sk_normals.reshape(batch_size, K, -1, 2)
    mnfld_loss = losses.reduce_mean_masked_instance(sk_pred.reshape(batch_size, K, -1, 1).abs().mean(dim=-1).mean(dim=-1), mask_gt).mean()
    grad_loss = losses.reduce_mean_masked_instance(((nonmnfld_grad.norm(2, dim=-1) - 1) ** 2).mean(dim=-1), mask_gt).mean()
    normals_loss = torch.minimum((mnfld_grad - sk_normals).norm(2, dim=-1), (mnfld_grad + sk_normals).norm(2, dim=-1)).mean(dim=-1)   # SALD :639-645
    normals_loss = losses.reduce_mean_masked_instance(normals_loss, mask_gt).mean()
    return mnfld_loss + 0.1 * grad_loss + 1.0 * normals_loss, mnfld_loss, grad_loss, normals_loss


def sketch_branch_losses(pcs, X, W, W_2K, matching_indices, mask, gt_normals, gt_extrusion_instances, gt_bb_labels, gt_extrusion_axes,
                         gt_extrusion_centers, gt_sketches, pn_encoder, loaded_pn_encoder, implicit_net, sampler, K, num_sk_point,
                         with_im_loss=True, is_l2=False, rand_idx_pred=None, rand_idx_gt=None, nonmnfld_pnts=None, use_whole_pc=False,
                         W_encoder=None, use_gt_im=False, axis_feat=None, device_draws=False):
    """train_Point2Cyl.py:519-672.  pcs (B,N,3); X (B,N,3) predicted normals; W (B,N,K) and W_2K (B,N,2K) the softmaxed segmentation;
    matching_indices / mask from hungarian_matching; gt_sketches (B,K,S,4) = [point | normal] of the ground-truth profiles.
    use_whole_pc: the encoder (4 input channels) sees [xyz | W_reordered[:, :, k]] of all N points per segment instead of the projected
    sketch; W_encoder = the (B,N,K) segmentation WITH its autograd history (default: W as given).
    axis_feat (B,K,3), with use_whole_pc: the segment's extrusion axis repeated over the points as channels 5-7 of the encoder input
    (7 input channels; the fitted axes E_AX with their history, or the ground-truth axes under use_gt_im).
    use_gt_im: the encoder input is built from the ground-truth labels (X, W, W_2K, matching_indices, mask are not read): the one-hot
    membership for use_whole_pc, else the projection of the ground-truth barrels divided by its OWN scales (:591-593).
    device_draws: the barrel-sample draws of the projections come from the device generator (fitting.barrel_draws_on_device) instead of
    the reference's CPU draws: no device->host sync in the step, which makes it capturable into a HIP graph.
    -> dict(im_loss, latent_loss, mnfld_loss, grad_loss, normals_loss, latent_codes)."""
    B, N, _ = pcs.shape
    S = num_sk_point
    if device_draws and not use_whole_pc:
        with torch.no_grad():
            if rand_idx_gt is None:
                rand_idx_gt = fitting.barrel_draws_on_device(gt_extrusion_instances, gt_bb_labels, K, S)
    mask_gt = losses.get_mask_gt(gt_extrusion_instances, K)
    with torch.no_grad():
        sk_pnts = gt_sketches[:, :, :, :2].reshape(B * K, S, 2)                                                          # :602-604
        sk_normals = gt_sketches[:, :, :, -2:].reshape(B * K, S, 2)
        latent_codes_gt = loaded_pn_encoder(torch.cat((sk_pnts, sk_normals), dim=-1))                                    # :605 (its parameters are frozen)
    if use_whole_pc:                                                                                                     # :522-536, :569-586
        if use_gt_im:
            W_reordered = torch.nn.functional.one_hot(gt_extrusion_instances.reshape(-1), num_classes=K).view(B, N, K).float()  # :572-574
        else:
            Wg = W if W_encoder is None else W_encoder
            W_reordered = torch.gather(Wg, 2, matching_indices.unsqueeze(1).expand(B, N, K))                             # :519
            W_reordered = torch.where(mask.unsqueeze(1).expand(B, N, K) == 1, W_reordered, torch.zeros_like(W_reordered))   # :520
        cols = [pcs.unsqueeze(1).expand(B, K, N, 3), W_reordered.permute(0, 2, 1).unsqueeze(-1)]
        if axis_feat is not None:                                                                                        # :528-531, :577-580
            cols.append(axis_feat.unsqueeze(-2).expand(B, K, N, 3))
        global_pc = torch.cat(cols, dim=-1).reshape(B * K, N, 3 + 1 + (3 if axis_feat is not None else 0))
    elif use_gt_im:                                                                                                      # :588-600
        with torch.no_grad():
            pred_pc, pred_nrm, pred_scales = fitting.sketch_implicit_projection(pcs, gt_normals, gt_extrusion_instances, gt_bb_labels,
                                                                                gt_extrusion_axes, gt_extrusion_centers, S, rand_idx=rand_idx_gt)
            pred_pc = pred_pc / pred_scales.unsqueeze(-1).unsqueeze(-1)
            global_pc = torch.cat((pred_pc.reshape(B * K, S, 2), pred_nrm.reshape(B * K, S, 2)), dim=-1)
    else:
        with torch.no_grad():                                                          # labels: no gradient through the arg-max / sampling
            W_reordered = torch.gather(W, 2, matching_indices.unsqueeze(1).expand(B, N, K))                               # :519
            W_reordered = torch.where(mask.unsqueeze(1).expand(B, N, K) == 1, W_reordered, torch.zeros_like(W_reordered))
            label = torch.argmax(W_reordered, dim=-1)                                                                    # :539
            BB = torch.stack([W_2K[:, :, 0::2].sum(-1), W_2K[:, :, 1::2].sum(-1)], -1)                                   # :542-546
            pred_bb_label = torch.argmax(BB, dim=-1)
            if device_draws and rand_idx_pred is None:
                rand_idx_pred = fitting.barrel_draws_on_device(label, pred_bb_label, K, S)
            pred_pc, pred_nrm, _ = fitting.sketch_implicit_projection(pcs, X, label, pred_bb_label, gt_extrusion_axes, gt_extrusion_centers, S,
                                                                      rand_idx=rand_idx_pred)                            # :548
            _, _, gt_scales = fitting.sketch_implicit_projection(pcs, gt_normals, gt_extrusion_instances, gt_bb_labels, gt_extrusion_axes,
                                                                 gt_extrusion_centers, S, rand_idx=rand_idx_gt)         # :549
            pred_pc = pred_pc / gt_scales.unsqueeze(-1).unsqueeze(-1)                                                    # :551-552
            global_pc = torch.cat((pred_pc.reshape(B * K, S, 2), pred_nrm.reshape(B * K, S, 2)), dim=-1)                 # :554-557 (the reference's own
            #                                                                               (K,B,..) -> (B*K,..) reshape, kept as it is)
    latent_codes = pn_encoder(global_pc)                                                                                 # :559
    zero = torch.zeros((), device=pcs.device)
    if with_im_loss:
        if nonmnfld_pnts is None:
            nonmnfld_pnts = sampler.get_points(sk_pnts)                                                                  # :609
        im_loss, mnfld_loss, grad_loss, normals_loss = implicit_losses(implicit_net, sk_pnts, sk_normals, nonmnfld_pnts, latent_codes, mask_gt, B, K)
    else:
        im_loss, mnfld_loss, grad_loss, normals_loss = zero, zero, zero, zero                                            # :650-654
    lc, lg = latent_codes.reshape(B, K, -1), latent_codes_gt.reshape(B, K, -1)
    if is_l2:
        latent_loss = losses.reduce_mean_masked_instance(torch.square(lc - lg).sum(dim=-1), mask_gt).mean()            # :662-663
    else:
        latent_loss = losses.reduce_mean_masked_instance(1.0 - torch.sum(lc * lg, dim=-1), mask_gt).mean()             # :667-669
    return dict(im_loss=im_loss + latent_loss, latent_loss=latent_loss, mnfld_loss=mnfld_loss, grad_loss=grad_loss, normals_loss=normals_loss,
                latent_codes=latent_codes)
