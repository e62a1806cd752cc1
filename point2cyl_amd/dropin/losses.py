"""Drop-in for the reference's losses.py (hot-path subset; the sketch/chamfer helpers are dead code upstream)."""
from point2cyl_amd.losses import *  # noqa: F401,F403
from point2cyl_amd.losses import (TORCH_PI, acos_safe, compute_all_losses, compute_bb_loss, compute_miou_loss,  # noqa: F401
                                  compute_normal_difference, compute_normal_loss, compute_segmentation_iou, g_zero_tol,
                                  get_mask_gt, hard_W_encoding, hungarian_matching, reduce_mean_masked_instance, sequence_mask)
