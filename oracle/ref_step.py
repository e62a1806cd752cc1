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
