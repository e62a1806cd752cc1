"""Drop-in for the reference's models/pointnet_util.py: same public names, MI355X implementation."""
import torch

from point2cyl_amd import ops as _ops
from point2cyl_amd.backbone import (PointNetFeaturePropagation, PointNetSetAbstraction,  # noqa: F401
                                    PointNetSetAbstractionMsg, draw_fps_start)


def farthest_point_sample(xyz, npoint):
    """(B,N,3) -> (B,npoint) int64; start drawn on the CPU generator like pointnet_util.py:75."""
    idx, _ = _ops.fps(xyz, npoint, draw_fps_start(xyz.shape[1], xyz.shape[0]))
    return idx.long()


def query_ball_point(radius, nsample, xyz, new_xyz):
    return _ops.ball_query(radius, nsample, xyz, new_xyz).long()


def index_points(points, idx):
    B = points.shape[0]
    flat = idx.reshape(B, -1)
    out = torch.gather(points, 1, flat.unsqueeze(-1).expand(-1, -1, points.shape[-1]))
    return out.reshape(*idx.shape, points.shape[-1])


def square_distance(src, dst):
    """Kept for API completeness (host-side torch expression; the kernels never materialise this matrix)."""
    d = -2 * torch.matmul(src, dst.permute(0, 2, 1))
    d = d + torch.sum(src ** 2, -1).unsqueeze(-1)
    return d + torch.sum(dst ** 2, -1).unsqueeze(1)
