"""Drop-in for the reference's losses.py: the hot-path subset (losses.py:22-159, :317-351) on the HIP kernels; every other name
(the sketch / chamfer helpers, the module's own imports) falls through to the shadowed reference module."""
from point2cyl_amd.losses import (TORCH_PI, acos_safe, compute_all_losses, compute_bb_loss, compute_miou_loss,  # noqa: F401
                                  compute_normal_difference, compute_normal_loss, compute_segmentation_iou, g_zero_tol,
                                  get_mask_gt, hard_W_encoding, hungarian_matching, reduce_mean_masked_instance, sequence_mask)
from point2cyl_amd._shadow import reexport as _reexport

_reexport("losses", __file__, globals())
