"""Segmentation / normal / base-barrel losses of Point2Cyl (mirror of the reference's losses.py API).

Signatures follow losses.py (file:line cited per function).  The Hungarian assignment runs on the device
(ops.hungarian: no per-sample .cpu() round trip, losses.py:43); the remaining reductions are thin
torch expressions on device tensors.
"""
import torch
import torch.nn.functional as F

from . import ops

FUSED_ALL_LOSSES = True                              # compute_all_losses through csrc/loss.hip where it applies (tests switch it off for the A/B)
g_zero_tol = 1.0e-6                                  # global_variables.py:15
TORCH_PI = torch.acos(torch.zeros(1)).item() * 2     # losses.py:17


def hungarian_matching(W_pred, I_gt, with_mask=False, validate=True):
    """losses.py:22-52.  W_pred (B,N,K), I_gt (B,N) with -1 = background -> matching_indices (B,K) int64
    [, mask (B,K) bool].  No gradient (by design, :23).  validate (not in the reference): see ops.hungarian."""
    match, mask = ops.hungarian(W_pred, I_gt, validate=validate)
    return (match, mask) if with_mask else match


def hard_W_encoding(W, to_null_mask=False, W_null_threshold=0.005):
    """losses.py:55-68: one-hot of the argmax; optionally zero the columns whose soft mass is < thr*N."""
    n_points, K = W.shape[1], W.shape[2]
    hard = F.one_hot(torch.argmax(W, dim=2), K).to(W.dtype)
    if to_null_mask:
        keep = (W.sum(dim=1) >= float(n_points) * W_null_threshold).to(W.dtype)
        hard = hard * keep.unsqueeze(1)
    return hard


def sequence_mask(lengths, maxlen=None):
    """losses.py:70-76."""
    maxlen = int(lengths.max()) if maxlen is None else maxlen
    return torch.arange(maxlen, device=lengths.device) < lengths.unsqueeze(-1)


def get_mask_gt(I_gt, n_max_instances):
    """losses.py:78-81: k < max(I_gt)+1."""
    return sequence_mask(I_gt.max(dim=1)[0] + 1, maxlen=n_max_instances)


def reduce_mean_masked_instance(loss, mask_gt):
    """losses.py:83-88: mean over the valid instances, 0 when there is none."""
    total = torch.where(mask_gt, loss, torch.zeros_like(loss)).sum(dim=1)
    n = mask_gt.sum(dim=1).to(loss.dtype)
    return torch.where(n > 0, total / n.clamp(min=1), torch.zeros_like(total))


def _reorder(W, matching_indices):
    """torch.gather(W, 2, matching_indices expanded over N) (losses.py:95; train...:479-521): the columns of every cloud permuted by its
    matching.  Done as W @ P with P[b, j, k] = (matching_indices[b, k] == j): the same values bit for bit (each output is one input
    times 1 plus zeros), and the backward is a batched 8 x 8 product instead of a scatter-add with atomics over B*N*K elements."""
    P = torch.nn.functional.one_hot(matching_indices, W.shape[2]).to(W.dtype).transpose(1, 2)
    return torch.bmm(W, P)


def compute_miou_loss(W, I_gt, matching_indices, div_eps=1e-10):
    """losses.py:90-103 -> (1 - IoU (B,K), 1 - dot/N, W_reordered)."""
    n_points = W.shape[1]
    K = matching_indices.shape[1]
    W_re = _reorder(W, matching_indices)
    onehot = (I_gt.unsqueeze(-1) == torch.arange(K, device=W.device)).to(W.dtype)    # -1 -> zero row (:96-98)
    dot = (onehot * W_re).sum(dim=1)
    union = onehot.sum(dim=1) + W_re.sum(dim=1) - dot
    return 1.0 - dot / (union + div_eps), 1 - dot / n_points, W_re


def compute_segmentation_iou(W, I_gt, matching_indices, mask):
    """losses.py:106-109."""
    iou = 1 - compute_miou_loss(W, I_gt, matching_indices)[0]
    return (mask * iou).sum(dim=1) / mask.sum(dim=1)


def acos_safe(x):
    """losses.py:123-124."""
    return torch.acos(torch.clamp(x, min=-1.0 + 1e-6, max=1.0 - 1e-6))


def compute_normal_loss(normal, normal_gt, angle_diff, collapse=True):
    """losses.py:127-143 (unoriented normals)."""
    cos = (normal * normal_gt).sum(dim=2).abs()
    val = acos_safe(cos) if angle_diff else 1.0 - cos
    return val.mean(dim=1) if collapse else val


def compute_normal_difference(X, X_gt, in_radians=True, collapse=True):
    """losses.py:146-159."""
    ang = acos_safe((X * X_gt).sum(dim=2).abs())
    if not in_radians:
        ang = ang * 180.0 / TORCH_PI
    return ang.mean(dim=1) if collapse else ang


def compute_all_losses(P, W, I_gt, X, X_gt, normal_loss_multiplier, miou_loss_multiplier, return_match_indices=False,
                       collapse=True):
    """losses.py:317-351.  On the device with both multipliers > 0, collapse and K <= 8 the matching, both reductions and their
    gradient are three launches of csrc/loss.hip (ops.all_losses) instead of ~35 torch launches forward and as many backward; every
    other call takes the torch expressions below (same values: tests/test_gpu_parity.py)."""
    B, _, K = W.shape
    if (FUSED_ALL_LOSSES and W.is_cuda and collapse and 1 <= K <= 8 and normal_loss_multiplier > 0 and miou_loss_multiplier > 0
            and W.dtype == torch.float32 and X.dtype == torch.float32):
        out3, matching_indices, mask = ops.all_losses(W, X, X_gt, I_gt, normal_loss_multiplier, miou_loss_multiplier)
        if return_match_indices:
            return out3[0], out3[1], out3[2], matching_indices, mask
        return out3[0], out3[1], out3[2]
    mask_gt = get_mask_gt(I_gt, K)
    if normal_loss_multiplier > 0:
        normal_loss = compute_normal_loss(X, X_gt, angle_diff=False)
    else:
        normal_loss = torch.zeros(B, K, device=P.device)
    matching_indices = mask = None
    if miou_loss_multiplier > 0:
        matching_indices, mask = hungarian_matching(W, I_gt, with_mask=True)
        avg_miou = reduce_mean_masked_instance(compute_miou_loss(W, I_gt, matching_indices)[0], mask_gt)
    else:
        avg_miou = torch.zeros(B, K, device=P.device)
    if collapse:
        total_miou, total_normal = avg_miou.mean(), normal_loss.mean()
        total = miou_loss_multiplier * total_miou + normal_loss_multiplier * total_normal
    else:
        total_miou, total_normal = avg_miou, normal_loss
        total = miou_loss_multiplier * total_miou + normal_loss_multiplier * total_normal
    if return_match_indices:
        return total, total_normal, total_miou, matching_indices, mask
    return total, total_normal, total_miou


def compute_bb_loss(W, W_barrel_bb, W_base_bb, matching_indices, mask, gt_bb_labels):
    """The base/barrel cross-entropy that is inline in the reference trainer
    (train_Point2Cyl_without_sketch.py:283-307): reorder W by the matching, zero unmatched columns,
    softmax over K, sort, pick the raw barrel/base logits of the sorted segments and weight a 2-class CE."""
    B, N, K = W.shape
    W_re = _reorder(W, matching_indices) * mask.to(W.dtype).unsqueeze(1)
    W_sorted, label = torch.sort(torch.softmax(W_re, dim=-1), dim=-1)
    barrel = torch.gather(W_barrel_bb, 2, label)
    base = torch.gather(W_base_bb, 2, label)
    # 2-class CE with target bb: logsumexp(barrel, base) - chosen
    lse = torch.logaddexp(barrel, base)
    chosen = torch.where(gt_bb_labels.unsqueeze(-1) == 0, barrel, base)
    ce = lse - chosen
    return (ce * W_sorted).sum(dim=-1).mean(dim=-1).mean()
