"""Evaluation counterpart of the reference's eval.py (fitting-accuracy path, eval.py:232-457 and the report :690-715), on the HIP kernels.

Per batch: backbone forward (eval mode) -> unit normals, 2K-way softmax, barrel / base split (:268-305) -> hard one-hot
segmentation with the null-column rule, Hungarian matching on the device, segmentation IoU (:314-320), reordered labels
(:322-326), normal angle error (:331-333), base/barrel accuracy (:339-342), extrusion axis by the batched signed-smallest
eigenvector fit with the four `--use_gt_*` operand choices (:347-397), axis angle error masked to the existing segments
(:398-405), per-segment hard centroids (:409-445), extents along the fitted axis (:456).  Everything stays on the device; the
per-cloud metric vectors come back in ONE transfer per batch (the reference syncs K*B times in its centroid loop alone).

`eval_metrics` is the batch function on head outputs (what the parity test drives with a reference-generated fixture);
`evaluate_batch` adds the model forward; `main` is the CLI (same flags as eval.py where they apply, plus --synthetic N because
there is no dataset on the box).  The optional sketch-fitting losses (eval.py:459-590: projection -> PointNetEncoder ->
ImplicitNet) run with --with_sketch_fit from randomly initialised / loaded sketch networks.

    python -m point2cyl_amd.eval --synthetic 64 --batch_size 16 [--logdir DIR --ckpt model.pth] [--use_gt_normals ...]
"""
import argparse
import os
import sys
import time
from dataclasses import dataclass

import numpy as np
import torch
import torch.nn.functional as F

from . import fitting, losses, ops
from . import hostmem


@dataclass
class EvalFlags:
    K: int = 8
    pred_seg: bool = True
    pred_normal: bool = True
    pred_bb: bool = True
    pred_extrusion: bool = True
    norm_eig: bool = False
    use_gt_normals: bool = False
    use_gt_segmentation: bool = False
    use_gt_bb: bool = False
    num_sk_point: int = 2048

    def pred_sizes(self):
        """eval.py:167-179."""
        return [3 if self.pred_normal else 1, 2 * self.K if (self.pred_seg and self.pred_bb) else (self.K if self.pred_seg else 1)]


def _reorder_gather(W, matching_indices):
    """torch.gather(W, 2, matching_indices expanded over N) (eval.py:322, :382, :393)."""
    return losses._reorder(W, matching_indices)


def eval_metrics(X_head, W_raw, pcs, gt_normals, gt_inst, gt_bb, gt_axes, gt_centers, fl: EvalFlags, extent_rand_idx=None, barrel_counts=None,
                 labels_validated=False):
    """eval.py:270-457 from the two head outputs.  gt_bb may be float (the reference casts it, :257) or integer.
    barrel_counts / labels_validated (not in the reference): fitting.barrel_counts(gt_inst, gt_bb, K) and "gt_inst is in [-1, K)" when the
    caller established them on the host copy of the labels - the call then has no device->host synchronisation in it.
    Returns a dict: per-cloud `mIoU`, `normal_difference`, `pred_bb_acc`, `extrusion_difference`, `centroid_difference` (B,);
    `extrusion_difference_uncollapsed`, `centroid_difference_uncollapsed` (B,K); `matching_indices`, `mask`, `label`,
    `pred_bb_label`, `E_AX` (B,K,3), `predicted_centroids` (B,K,3), `found_centers_mask` (B,K), `extents` (B,K,2), `mask_gt`."""
    B, N, _ = pcs.shape
    K = fl.K
    dev = pcs.device
    out = {}
    X = F.normalize(X_head, p=2, dim=2, eps=1e-12) if fl.pred_normal else torch.zeros(B, N, 3, device=dev)       # :270-274
    W_barrel = W_base = BB = None
    if fl.pred_seg and fl.pred_bb:
        W_2K = torch.softmax(W_raw, dim=2)                                           # :278
        W_barrel, W_base = W_2K[:, :, 0::2], W_2K[:, :, 1::2]                         # :282-286
        W = W_barrel + W_base                                                         # :289
        BB = torch.stack([W_barrel.sum(-1), W_base.sum(-1)], dim=-1)                  # :297-300
    elif fl.pred_seg:
        W = torch.softmax(W_raw, dim=2)
    else:
        W = torch.zeros(B, N, K, device=dev)
    gt_bb_i = gt_bb.to(torch.long)
    matching_indices = mask = W_reordered = W_ = None
    if fl.pred_seg:
        W_ = losses.hard_W_encoding(W, to_null_mask=True)                             # :316
        matching_indices, mask = losses.hungarian_matching(W_, gt_inst, with_mask=True, validate=not labels_validated)   # :318
        mask = mask.float()
        out["mIoU"] = losses.compute_segmentation_iou(W_, gt_inst, matching_indices, mask)   # :320
        W_re_un = _reorder_gather(W_, matching_indices)
        W_reordered = torch.where(mask.unsqueeze(1) == 1, W_re_un, -torch.ones_like(W_re_un))     # :323-324
        out["label"] = torch.argmax(W_reordered, dim=-1)                              # :326
    else:
        out["mIoU"] = torch.ones(B, device=dev)
    out["normal_difference"] = (losses.compute_normal_difference(X, gt_normals, in_radians=False) if fl.pred_normal
                                else torch.zeros(B, device=dev))                      # :331-336
    if fl.pred_bb:
        pred_bb_label = torch.argmax(BB, dim=-1)                                      # :340
        out["pred_bb_label"] = pred_bb_label
        out["pred_bb_acc"] = (pred_bb_label == gt_bb_i).sum(dim=-1) / float(N)        # :342
    else:
        out["pred_bb_acc"] = torch.zeros(B, device=dev)
    mask_gt = losses.get_mask_gt(gt_inst, K)
    out["mask_gt"] = mask_gt
    if fl.pred_extrusion:
        EA_X = gt_normals if fl.use_gt_normals else X                                 # :348-351
        gt_onehot = F.one_hot(gt_inst.clamp(min=0), K).float() * (gt_inst >= 0).unsqueeze(-1)
        if fl.use_gt_segmentation and fl.use_gt_bb:                                   # :353-360
            EA_W, bbsel = gt_onehot, gt_bb_i
        elif fl.use_gt_segmentation:                                                  # :363-372
            EA_W, bbsel = gt_onehot, torch.argmax(BB, dim=-1)
        elif fl.use_gt_bb:                                                            # :374-383
            EA_W, bbsel = _reorder_gather(W_, matching_indices), gt_bb_i
        else:                                                                         # :386-394
            EA_W, bbsel = W_reordered, None
        if bbsel is not None:
            Wb_re = torch.where((bbsel == 0).unsqueeze(-1), EA_W, torch.zeros_like(EA_W))
            Wc_re = torch.where((bbsel == 1).unsqueeze(-1), EA_W, torch.zeros_like(EA_W))
        else:
            Wb_re, Wc_re = _reorder_gather(W_barrel, matching_indices), _reorder_gather(W_base, matching_indices)
        E_AX, E64 = fitting.estimate_extrusion_axis(EA_X, Wb_re, Wc_re, gt_bb_i, gt_inst, normalize=fl.norm_eig, return_float64=True)   # :397
        # :398, evaluated in float64 on the axes as the kernel's fp64 eigen-solve left them (p2c_extrusion_axis_f32's axis64_out): the angle
        # of two nearly parallel unit vectors is an acos next to its clamp, where an fp32 dot product alone costs up to 1e-3 of the angle and
        # the |E| = 1 +- 3e-8 of an fp32-stored unit vector as much again; the fp64 run of the reference is the yardstick
        ext_diff = losses.compute_normal_difference(E64, gt_axes.double(), in_radians=False, collapse=False).float()
        out["extrusion_difference_uncollapsed"] = torch.where(mask_gt, ext_diff, torch.zeros_like(ext_diff))       # :403
        out["extrusion_difference"] = losses.reduce_mean_masked_instance(ext_diff, mask_gt)                       # :405
        cen, found = fitting.segment_centroids(EA_W, pcs)                                                          # :409-436
        cdiff = torch.square(cen - gt_centers).sum(dim=-1)                                                         # :439
        out["centroid_difference_uncollapsed"] = torch.where(mask_gt, cdiff, torch.zeros_like(cdiff))              # :445
        out["centroid_difference"] = losses.reduce_mean_masked_instance(cdiff, mask_gt)                            # :446
        out.update(E_AX=E_AX, predicted_centroids=cen, found_centers_mask=found)
    else:
        z = torch.zeros(B, device=dev)
        out.update(extrusion_difference=z, centroid_difference=z, extrusion_difference_uncollapsed=torch.zeros(B, K, device=dev),
                   centroid_difference_uncollapsed=torch.zeros(B, K, device=dev))
    extents, _ = fitting.get_extrusion_extents(pcs, gt_inst, gt_bb_i, gt_axes, gt_centers, num_points_to_sample=fl.num_sk_point,
                                               rand_idx=extent_rand_idx, counts=barrel_counts)                     # :456
    out["extents"] = extents.permute(1, 0, 2)                                                                      # :457
    out.update(matching_indices=matching_indices, mask=mask, X=X, W=W)
    return out


@torch.no_grad()
def evaluate_batch(model, pcs, gt_normals, gt_inst, gt_bb, gt_axes, gt_centers, fl: EvalFlags, extent_rand_idx=None, heads=None,
                   barrel_counts=None, labels_validated=False):
    """eval.py:268 + eval_metrics.  heads = (heads (B*N, ld), sizes) from graph.PipelinedForward instead of running the forward here."""
    if heads is not None:
        h, sizes = heads
        B, N = pcs.shape[0], pcs.shape[1]
        h = h.view(B, N, h.shape[-1])
        X_head, W_raw = h[:, :, 0:sizes[0]], h[:, :, sizes[0]:sizes[0] + sizes[1]]
    else:
        X_head, W_raw = model(pcs)
    return eval_metrics(X_head, W_raw, pcs, gt_normals, gt_inst, gt_bb, gt_axes, gt_centers, fl, extent_rand_idx, barrel_counts, labels_validated)


@torch.no_grad()
def sketch_fit_losses(m, pcs, gt_normals, gt_inst, gt_bb, implicit_net, pn_encoder, fl: EvalFlags):
    """eval.py:459-590, default branch (no --use_gt_im / --use_whole_pc): project the predicted barrels (:492-501), encode them,
    evaluate the implicit profile on the gt-label projection along the PREDICTED axis / centroid (:555-575, per-cylinder fitting
    loss) and on all points (:577-590, global fitting loss).  `m` = the dict eval_metrics returned.  -> (pred_fit_cyl_loss (B,),
    pred_fit_glob_loss (B,))."""
    from .implicit import add_latent
    B, N, _ = pcs.shape
    K, S = fl.K, fl.num_sk_point
    gt_bb_i = gt_bb.to(torch.long)
    Wre = _reorder_gather(m["W"], m["matching_indices"]) * m["mask"].unsqueeze(1)                       # :465-466
    label = torch.argmax(Wre, dim=-1)                                                                   # :489
    E_AX, cen = m["E_AX"], m["predicted_centroids"]
    ppc, pn, scales = fitting.sketch_implicit_projection(pcs, m["X"], label, m["pred_bb_label"], E_AX, cen, num_points_to_sample=S)     # :499
    ppc = ppc / scales.unsqueeze(-1).unsqueeze(-1)
    latent = pn_encoder(torch.cat([ppc.reshape(B * K, S, 2), pn.reshape(B * K, S, 2)], dim=-1))          # :502-506
    ppc2, _, _, found2 = fitting.sketch_implicit_projection2(pcs, gt_normals, gt_inst, gt_bb_i, E_AX, cen, num_points_to_sample=S)     # :555
    ppc2 = (ppc2 / scales.unsqueeze(-1).unsqueeze(-1)).reshape(B * K, S, 2)
    sk = implicit_net(add_latent(ppc2, latent)).reshape(K, B, S)                                         # :561-563
    pmask = (m["mask"].T * found2.T).unsqueeze(-1)                                                       # :567-569
    n_inst = (gt_inst.max(dim=1)[0] + 1).float()
    cyl = (sk * pmask).abs().permute(1, 0, 2).mean(-1).reshape(B, -1).sum(1) / n_inst                    # :571-575
    ppc3, _, _, found3 = fitting.sketch_implicit_projection3(pcs, gt_normals, gt_inst, gt_bb_i, E_AX, cen, num_points_to_sample=N)     # :577
    ppc3 = (ppc3 / scales.unsqueeze(-1).unsqueeze(-1)).reshape(B * K, N, 2)
    sk3 = implicit_net(add_latent(ppc3, latent)).reshape(K, B, N)
    pmask3 = (m["mask"].T * found3.T).unsqueeze(-1)
    sk3 = torch.where(pmask3 == 1, sk3.abs().float(), torch.full_like(sk3, 10000.0))                     # :586
    glob = sk3.min(dim=0)[0] * (1.0 - gt_bb.float())                                                     # :587-589
    glob = glob.sum(1) / (float(N) - gt_bb.float().sum(1))                                               # :590
    return cyl, glob


REPORT = (("mIoU", "Mean mIOU= "), ("normal_difference", "Mean normal angle error (degrees) = "), ("pred_bb_acc", "Mean base/barrel accuracy= "),
          ("extrusion_difference", "Mean extrusion angle error (degrees) = "), ("centroid_difference", "Mean centroid difference = "))


class Accumulator:
    """eval.py:638-676, :690-715: per-shape sums in float64 on the host, ONE device->host transfer per batch - asynchronous: the
    (n_metrics, B) block of a batch lands in one of `depth` pinned buffers and is summed when its copy has completed (at the latest
    `depth` batches later, or in means()), so the loop that calls add() never waits for the device."""

    def __init__(self, extra=(), depth=4):
        self.keys = [k for k, _ in REPORT] + list(extra)
        self.n, self.tot = 0, np.zeros(len(self.keys), dtype=np.float64)
        self.depth, self.ring, self.inflight = depth, {}, []

    def _drain(self, keep):
        while len(self.inflight) > keep:
            buf, ev = self.inflight.pop(0)
            ev.synchronize()
            v = buf.numpy()
            self.tot += v.sum(1)
            self.n += v.shape[1]
            self.ring[tuple(buf.shape)].append(buf)

    def block(self, m):
        return torch.stack([m[k].float() for k in self.keys], 0).double()          # (n_metrics, B)

    def add(self, m):
        self.add_block(self.block(m))

    def add_block(self, v):
        if not v.is_cuda:
            self.tot += v.numpy().sum(1)
            self.n += v.shape[1]
            return
        self._drain(self.depth - 1)
        free = self.ring.setdefault(tuple(v.shape), [])
        buf = free.pop() if free else torch.empty(v.shape, dtype=torch.float64).pin_memory()
        buf.copy_(v, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.inflight.append((buf, ev))

    def sums(self):
        self._drain(0)
        return self.tot, self.n

    def means(self):
        tot, n = self.sums()
        return dict(zip(self.keys, tot / max(1, n)))


class _PinnedCollate:
    """collate_fn of the evaluation loader: the fields the loop reads (clouds, normals, labels, axes, centres) stacked into one of two
    sets of recycled pinned buffers, the others left out; -> the reference's 9-tuple with None in the unused places.  The host->device
    copies that follow run at the pinned rate, and nothing is first-touched per batch (hostmem.py)."""
    USED = (0, 1, 2, 3, 6, 8)

    def __init__(self):
        self.slots, self.i = [{}, {}], 0
        self.copied = [None, None]           # event behind the host->device copies that last read a slot (mark_copied)
        self.last = 0

    def mark_copied(self):
        """Called once the copies out of the most recent batch's buffers are enqueued on the current stream: the slot is not written again
        before they have run (the serial loop enqueues ahead of the device; the loader thread synchronises its stream instead)."""
        ev = torch.cuda.Event()
        ev.record()
        self.copied[self.last] = ev
        return ev

    def __call__(self, items):
        if self.copied[self.i] is not None:
            self.copied[self.i].synchronize()
            self.copied[self.i] = None
        slot = self.slots[self.i]
        self.last = self.i
        self.i ^= 1
        n, out = len(items), [None] * len(items[0])
        for j in self.USED:
            first = np.asarray(items[0][j])
            buf = slot.get(j)
            if buf is None or buf.shape[0] < n or buf.shape[1:] != first.shape or buf.numpy().dtype != first.dtype:
                buf = slot[j] = torch.from_numpy(np.empty((n,) + first.shape, first.dtype)).pin_memory()
            np.stack([np.asarray(it[j]) for it in items], out=buf.numpy()[:n])
            out[j] = buf[:n]
        return out


class GraphedMetrics:
    """eval_metrics + the accumulator's (n_metrics, B) block of one batch as ONE HIP-graph replay (the evaluation loop's default for the batches
    of its pipeline: ~120 small launches, 2 ms of host time, become one).  What makes the chain capturable: the barrel counts and the label
    check come from the host copy of the labels, the extent draws (CPU generator, data_utils.py:1696) are made BEFORE the replay into a
    fixed device buffer.  Same kernels on the same values as the eager call; the dict of per-point outputs is not returned (the loop only
    accumulates the report's metrics - --with_sketch_fit, which needs them, takes the eager call)."""

    def __init__(self, fl, keys, batch, heads):
        from . import measure
        h, self.sizes = heads
        self.fl, self.keys = fl, list(keys)
        self.h = torch.empty_like(h)
        self.inp = [torch.empty_like(t) for t in batch[:6]]
        B, N = batch[0].shape[0], batch[0].shape[1]
        self.rand = torch.zeros(B, fl.K, fl.num_sk_point, dtype=torch.int64, device=h.device)
        self.load(batch, heads)

        def body():
            hv = self.h.view(B, N, self.h.shape[-1])
            m = eval_metrics(hv[:, :, 0:self.sizes[0]], hv[:, :, self.sizes[0]:self.sizes[0] + self.sizes[1]], *self.inp, fl,
                             extent_rand_idx=self.rand, labels_validated=True)
            return torch.stack([m[k].float() for k in self.keys], 0).double()

        self.graph, self.out = measure.capture(body)

    def load(self, batch, heads):
        ops.copy_flat_batch(self.inp + [self.h], [t.contiguous() for t in batch[:6]] + [heads[0]])
        fitting._barrel_draws(batch[2], batch[3], self.fl.K, self.fl.num_sk_point, device=self.h.device, counts=batch[6].get("barrel_counts"),
                              out=self.rand)

    def __call__(self, batch=None, heads=None):
        """-> the (n_metrics, B) float64 block of this batch (a static buffer: consume it - Accumulator.add_block - before the next call).
        Without arguments: of the batch the constructor was given (its draws are made once)."""
        if batch is not None:
            self.load(batch, heads)
        self.graph.replay()
        return self.out


def fused_metrics_applicable(fl: EvalFlags, sizes=None):
    """csrc/metrics.hip covers eval.py's DEFAULT operand choice (every head on, no --use_gt_*)."""
    ok = (fl.pred_seg and fl.pred_normal and fl.pred_bb and fl.pred_extrusion and not (fl.use_gt_normals or fl.use_gt_segmentation or fl.use_gt_bb)
          and ops.eval_metrics_supported(fl.K))
    return ok and (sizes is None or list(sizes) == [3, 2 * fl.K])


class FusedMetrics:
    """eval.py:270-457 of one batch as FOUR launches: ops.eval_metrics_fused (two kernels: per-point pass, per-cloud finish with the
    matching and the eigen-solves) + the extents along the ground-truth axes (:456, two kernels) - instead of the ~90 kernel names of the
    torch-op mirror `eval_metrics` (which stays the reference-order path: --no_prefetch, --use_gt_*, --with_sketch_fit).  -> the
    accumulator's (5, B) float64 block.  Counts-based rows (mIoU, base/barrel accuracy) equal the mirror's; the float rows agree to
    rounding (tests/test_gpu_flows.py)."""

    def __init__(self, fl, keys):
        assert list(keys) == [k for k, _ in REPORT], keys
        self.fl, self.out, self.i = fl, None, 0

    def __call__(self, batch, heads):
        h, sizes = heads
        pcs, nrm, inst, bb, axes, cen = batch[:6]
        fl = self.fl
        B = pcs.shape[0]
        if self.out is None or self.out[0].shape[1] != B:
            self.out = [torch.empty(5, B, dtype=torch.float64, device=pcs.device) for _ in range(2)]
        self.i ^= 1
        out = ops.eval_metrics_fused(h, 0, sizes[0], pcs, nrm, inst, bb, axes, cen, fl.K, normalize=fl.norm_eig, out=self.out[self.i])
        ex = batch[6] if len(batch) > 6 else {}
        rand = ex.get("extent_rand_idx")
        if rand is None and ex.get("barrel_counts_dev") is not None:
            # uniform integers in [0, n_barrel(b, k)) from the DEVICE generator: no host draws, no upload (see fitting.barrel_draws_on_device)
            cnt = ex["barrel_counts_dev"].unsqueeze(-1)
            u = torch.rand(B, fl.K, fl.num_sk_point, device=pcs.device)
            rand = torch.minimum((u * cnt).long(), (cnt - 1).clamp_min(0))
        bbi = ex.get("bb_long")
        self.extents, _ = fitting.get_extrusion_extents(pcs, inst, bbi if bbi is not None else bb.to(torch.long), axes, cen,
                                                        num_points_to_sample=fl.num_sk_point, rand_idx=rand, counts=ex.get("barrel_counts"))     # :456
        return out


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument("--model", type=str, default="pointnet_extrusion")
    p.add_argument("--num_point", type=int, default=8192)
    p.add_argument("--num_sk_point", type=int, default=2048)
    p.add_argument("--K", type=int, default=8)
    p.add_argument("--batch_size", type=int, default=4)
    p.add_argument("--logdir", default="./results/", type=str)
    p.add_argument("--ckpt", default="model.pth", type=str)
    p.add_argument("--dump_dir", default="./results/", type=str)
    p.add_argument("--data_dir", type=str, default="data/")
    p.add_argument("--data_split", default="test", type=str)
    for f in ("pred_seg", "pred_normal", "pred_bb", "pred_extrusion"):
        p.add_argument("--" + f, action="store_false")          # eval.py:50-53: on by default, the flag switches the head OFF
    for f in ("norm_eig", "add_noise", "use_gt_normals", "use_gt_segmentation", "use_gt_bb", "with_sketch_fit"):
        p.add_argument("--" + f, action="store_true")
    p.add_argument("--noise_sigma", type=float, default=0.01)
    p.add_argument("--im_logdir", default="./results/IGR_dense/", type=str)
    p.add_argument("--im_ckpt", default="latest.pth", type=str)
    p.add_argument("--synthetic", type=int, default=0, help="evaluate N generated extrusion-cylinder clouds instead of <data_dir>/<split>.h5")
    p.add_argument("--random_init", action="store_true", help="no checkpoint: evaluate a randomly initialised backbone (plumbing / timing runs)")
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--no_prefetch", action="store_true", help="run FPS / ball query / 3-NN of every batch on the critical path instead of one batch "
                   "ahead on a forked stream (graph.PipelinedForward), no loader thread: the ONLY mode that consumes the CPU random generator in "
                   "the reference's order (FPS starts, then extent samples, batch by batch); the default draws the same quantities in another order / "
                   "on the device")
    p.add_argument("--prefetch_group", type=int, default=4, help="batches whose geometry is computed TOGETHER, one group ahead (FPS is 512 dependent "
                   "steps per cloud whether 32 or 128 clouds are sampled: its latency is shared by the group; 1 = one batch ahead)")
    p.add_argument("--report", type=str, default="", help="write a JSON throughput report here")
    p.add_argument("--no_graph_metrics", action="store_true", help="launch the metric kernels of a pipelined batch one by one instead of replaying a HIP graph")
    p.add_argument("--no_fused_metrics", action="store_true", help="pipelined batches: the torch-op metric chain (as one HIP-graph replay) instead of the "
                   "two fused kernels of csrc/metrics.hip")
    return p


def main(argv=None):
    a = build_parser().parse_args(argv)
    hostmem.setup_cli()        # large host blocks stay mapped after free; torch's CPU pool sized to the cgroup's CPU quota (hostmem.py)
    if not torch.cuda.is_available():
        raise SystemExit("point2cyl_amd.eval needs an MI355X (HIP) device; there is no CPU path")
    from . import ddp, synth
    from .backbone import backbone
    rank, world, local = ddp.init_from_env()
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    np.random.seed(0)                                                           # eval.py:128
    torch.manual_seed(a.seed)
    fl = EvalFlags(K=a.K, pred_seg=a.pred_seg, pred_normal=a.pred_normal, pred_bb=a.pred_bb, pred_extrusion=a.pred_extrusion,
                   norm_eig=a.norm_eig, use_gt_normals=a.use_gt_normals, use_gt_segmentation=a.use_gt_segmentation, use_gt_bb=a.use_gt_bb,
                   num_sk_point=a.num_sk_point)
    if a.synthetic > 0:
        gen = synth.SyntheticExtrusionDataset(a.synthetic, a.num_point, a.K, seed=99991)
        ds = [gen[i] for i in range(len(gen))]        # generated up front (2 - 4 ms of host time per cloud): the loop below times the evaluation, not the generator
    else:
        from .h5data import AutodeskH5, dataset_path
        ds = AutodeskH5(dataset_path(a.data_dir, a.data_split), a.num_point, a.K, center=True)
    lo, hi = ddp.shard_range(len(ds), rank, world)                               # clouds are independent: shard, no data-path collective
    # (no pin_memory: the loader pins fresh buffers in the calling thread, 9 ms per tensor here - 80 ms a batch; _PinnedCollate recycles two)
    shuffle = a.data_split != "test"
    collate = _PinnedCollate()          # (--no_prefetch too: recycled pinned buffers change no draw and no value, only the copy rate)
    loader = torch.utils.data.DataLoader(torch.utils.data.Subset(ds, range(lo, hi)), batch_size=a.batch_size, num_workers=0, shuffle=shuffle,
                                         generator=torch.Generator().manual_seed(a.seed) if shuffle else None, collate_fn=collate)
    model = backbone(output_sizes=fl.pred_sizes())
    if not a.random_init:
        sd = torch.load(os.path.join(a.logdir, a.ckpt), map_location="cpu")["model"]            # eval.py:206-207
        model.load_state_dict(sd)
    model.to(dev).eval()
    implicit_net = pn_encoder = None
    if a.with_sketch_fit:
        from .implicit import ImplicitNet
        from .sketch import PointNetEncoder
        implicit_net = ImplicitNet(d_in=2 + 256, dims=[512] * 8, skip_in=[4], geometric_init=True, radius_init=1, beta=100)      # eval.py:190
        pn_encoder = PointNetEncoder(256, 2, with_normals=True)                                                                 # eval.py:193
        f = os.path.join(a.im_logdir, a.im_ckpt)
        if os.path.exists(f):
            ck = torch.load(f, map_location="cpu")
            implicit_net.load_state_dict(ck["model_state_dict"])                                  # eval.py:209-210
            pn_encoder.load_state_dict(ck["encoder_state_dict"])
        implicit_net.to(dev).eval()
        pn_encoder.to(dev).eval()
    acc = Accumulator(("pred_fit_cyl_loss", "pred_fit_glob_loss") if a.with_sketch_fit else ())
    os.makedirs(a.dump_dir, exist_ok=True)
    log = open(os.path.join(a.dump_dir, "log_evaluate.txt" if world == 1 else "log_evaluate.%d.txt" % rank), "w")
    log.write(str(a) + "\n")
    t0 = time.time()

    draw_gen = [None]

    def to_device(b):
        pcs, nrm, inst, bb, axes, cen = b[0], b[1], b[2], b[3], b[6], b[8]
        # on the host copy of the labels, before the upload: the range check losses.py:36-46 makes per cloud, and the barrel counts that
        # decide the extent draws (data_utils.py:1674-1697) - the evaluation of the batch then never waits for the device
        lo, hi = int(inst.min()), int(inst.max())
        if hi >= a.K or lo < -1:
            raise ValueError("instance labels must be in [-1, %d); got [%d, %d]" % (a.K, lo, hi))
        # (the labels go up ONCE, as the int64 the collate pinned; the float copy eval.py:256 makes is a device cast of it - `.to(dev, float)` of
        # the host tensor is a host cast into an unpinned temporary and a synchronous copy)
        inst_d, bb_d = inst.to(dev, torch.long, non_blocking=True), bb.to(dev, torch.long, non_blocking=True)
        if draw_gen[0]:
            # fused pipelined loop: the extent samples of a batch (data_utils.py:1696: K x B torch.randint calls on the CPU generator, 0.8 ms of
            # host time and a 4 MB upload per batch) are drawn ON THE DEVICE (FusedMetrics; another equally valid sampling - the reference's
            # stream is what --no_prefetch keeps) from barrel counts that are taken on the device too: the host scatter-add they replace was
            # 0.3 - 0.6 ms of this thread's 1.3 ms per batch, and the loop runs at this thread's rate once the device side is below it
            extras = dict(labels_validated=True, barrel_counts_dev=fitting.barrel_counts_tensor(inst_d, bb_d, a.K), bb_long=bb_d)
        else:
            extras = dict(barrel_counts=fitting.barrel_counts(inst.long(), bb.long(), a.K), labels_validated=True)
        pcs, nrm, axes, cen = [t.to(dev, torch.float, non_blocking=True) for t in (pcs, nrm, axes, cen)]
        if a.add_noise:           # eval.py:241: the reference's host draws, its float64 multiply-add and float32 rounding (:254) - made on the device
            src = [t if h.dtype == torch.float32 else h.to(dev) for t, h in ((pcs, b[0]), (nrm, b[1]))]       # (float64 datasets: noise before the rounding)
            pcs = fitting.add_noise_on_device(src[0], src[1], sigma=a.noise_sigma).to(torch.float)
        out = pcs, nrm, inst_d, bb_d.to(torch.float), axes, cen, extras     # eval.py:254-257
        if a.no_prefetch:
            collate.mark_copied()
        return out

    # The reference's loop (eval.py:231-268) knows its next batch; the geometry of batch i + 1 (FPS / ball query / 3-NN: 0.75 of the
    # 1.84 ms serial forward at B = 32 x 8192) is computed inside the graph that runs batch i's forward (graph.PipelinedForward).
    # Batches of another shape than the first (the last, shorter one) take the serial `model(pcs)`.  Random streams: the FPS starts of batch
    # i + 1 are drawn (CPU generator, SA1 then SA2, like the reference) BEFORE batch i's extent samples (data_utils.py:1696, same generator),
    # in the serial loop after them: another equally valid random sampling of the same clouds, not the same one (--no_prefetch keeps the
    # reference's order).
    # Host side: the loader's collate, --add_noise and the host->device copies run in a producer thread a few batches ahead (they release the
    # GIL).  Random streams: NumPy's generator is drawn from by that thread only (the noise); torch's CPU generator by this one only (the
    # loader's base seed here below, the FPS starts, the extent samples) - the per-item subsampling permutations of the h5 dataset
    # (dataloader.py:69-85, the same generator in the reference) come from a private generator seeded from it HERE, so that a seeded run is
    # reproducible whatever the two threads' timing.  --no_prefetch: no thread, every draw in the reference's order.
    import collections
    import queue
    import threading
    fused = None
    if fused_metrics_applicable(fl) and not (a.with_sketch_fit or a.no_graph_metrics or a.no_fused_metrics):
        fused = FusedMetrics(fl, acc.keys)
        draw_gen[0] = not a.no_prefetch         # (--no_prefetch: the extent draws stay on the CPU generator, in the reference's order)
    if not a.no_prefetch and hasattr(ds, "generator"):
        ds.generator = torch.Generator().manual_seed(int(torch.empty((), dtype=torch.int64).random_().item()))
    it = iter(loader)
    pending = collections.deque()
    G = max(1, int(a.prefetch_group))
    ready, drained = queue.Queue(maxsize=2 * G + 2), [False]

    stop = threading.Event()          # set when the consuming loop ends - normally or by an exception: the producer then lets go of its buffers

    def hand_over(item):
        while not stop.is_set():
            try:
                ready.put(item, timeout=0.2)
                return True
            except queue.Full:
                pass
        return False

    def produce():
        try:
            torch.cuda.set_device(dev)
            with torch.cuda.stream(torch.cuda.Stream(dev)):
                for b in it:                                     # (the collate waits for the copies that last READ the slot it fills: mark_copied)
                    if stop.is_set():
                        return
                    t = to_device(b)
                    # no host wait for the uploads: the event behind them goes with the batch (the loop's stream waits for it), and this thread
                    # stacks the next batch into the other pinned slot while they run
                    ev = collate.mark_copied()
                    if not hand_over((t, ev)):
                        return
            hand_over(None)
        except BaseException as e:          # surfaces in the consuming loop
            hand_over(e)

    producer = None
    if not a.no_prefetch:
        producer = threading.Thread(target=produce, daemon=True)
        producer.start()

    waited = [0.0]

    def fill(k):
        while len(pending) < k and not drained[0]:
            t_w = time.perf_counter()
            if a.no_prefetch:
                b = next(it, None)
                b = b if b is None else to_device(b)
            else:
                b = ready.get()
            waited[0] += time.perf_counter() - t_w
            if b is None:
                drained[0] = True
            elif isinstance(b, BaseException):
                raise b
            else:
                if not a.no_prefetch:
                    b, ev = b
                    stream.wait_event(ev)         # the loader thread's uploads of this batch
                for t in list(b[:6]) + [v for v in b[6].values() if torch.is_tensor(v)]:
                    t.record_stream(stream)       # allocated on the producer's stream, read on the loop's
                pending.append(b)

    def same(bs):
        return all(tuple(b[0].shape) == tuple(bs[0][0].shape) for b in bs) and bs[0][0].shape[2] == 3

    def evaluate(b, heads=None):
        if fused is not None and b[0].shape[2] == 3:
            # serial batches (--no_prefetch, the last short group): the same two metric kernels behind the plain forward; the CPU
            # generator is consumed as in the reference - FPS starts inside the forward, then the extent draws (FusedMetrics)
            with torch.no_grad():
                if heads is not None:
                    h = heads
                else:            # (what model(pcs) runs: the per-shape HIP graph of the forward where it applies)
                    # - in a serial RUN; the few left-over batches of a pipelined run are not worth a capture (50 - 100 ms each shape)
                    from . import autograph
                    h = autograph.forward_heads(model, b[0]) if (a.no_prefetch and autograph.applicable(model, b[0])) else None
                    h = h if h is not None else model.forward_heads(b[0])
            if list(h[1]) == [3, 2 * fl.K]:
                return acc.add_block(fused(b, h))
            heads = h
        ex = {k: v for k, v in b[6].items() if k in ("barrel_counts", "labels_validated", "extent_rand_idx")}
        m = evaluate_batch(model, *b[:6], fl, heads=heads, **ex)
        if a.with_sketch_fit:
            m["pred_fit_cyl_loss"], m["pred_fit_glob_loss"] = sketch_fit_losses(m, b[0], b[1], b[2], b[3], implicit_net, pn_encoder, fl)
        acc.add(m)

    pipe, pipe_shape, i, n_piped, metrics_graph = None, None, 0, 0, None
    t_first = None
    stream = torch.cuda.Stream(dev)
    stream.wait_stream(torch.cuda.current_stream())
    try:
        with torch.cuda.stream(stream):
            while True:
                group = None
                fill(1 if a.no_prefetch else 2 * G)          # (--no_prefetch: nothing is read ahead, every draw in the reference's order)
                if not pending:
                    break
                head = list(pending)[:G]
                can = not a.no_prefetch and len(head) == G and same(head) and (pipe is None or tuple(head[0][0].shape) == pipe_shape)
                nxt = list(pending)[G:2 * G]
                if not (len(nxt) == G and same(nxt) and can and tuple(nxt[0][0].shape) == tuple(head[0][0].shape)):
                    nxt = None
                if can and (pipe is not None or nxt is not None):          # (a pipeline is only STARTED when there is a next full group to prefetch for)
                    group = [pending.popleft() for _ in range(G)]
                    if pipe is None:
                        from .graph import PipelinedForward
                        pipe, pipe_shape = PipelinedForward(model, [b[0] for b in group], stream=stream, group=G), tuple(group[0][0].shape)
                    outs = pipe([b[0] for b in nxt] if nxt is not None else None)
                    for b, h in zip(group, outs):
                        if fused is not None and list(h[1]) == [3, 2 * fl.K]:
                            acc.add_block(fused(b, h))
                        elif a.with_sketch_fit or a.no_graph_metrics:
                            evaluate(b, heads=h)
                        elif metrics_graph is None or metrics_graph.inp[0].shape != b[0].shape:
                            metrics_graph = GraphedMetrics(fl, acc.keys, b, h)
                            acc.add_block(metrics_graph())
                        else:
                            acc.add_block(metrics_graph(b, h))
                    n_piped += G
                    done = G
                    if nxt is None:
                        pipe.release()        # no further full group: what is left (a short last group, another shape) takes the serial forward
                        pipe = None
                else:
                    if pipe is not None:
                        pipe.release()
                        pipe = None
                    evaluate(pending.popleft())
                    done = 1
                if (i // 20) != ((i + done) // 20) or i == 0:
                    if rank == 0:
                        print("Time elapsed: %s sec for batch %d/%d." % (time.time() - t0, i, len(loader)))
                if t_first is None:
                    torch.cuda.synchronize()
                    t_first, i_first, waited[0] = time.time(), i + done, 0.0
                i += done
    finally:
        stop.set()                      # (a loop that raises must not leave the producer blocked on a full queue)
        while True:
            try:
                ready.get_nowait()
            except queue.Empty:
                break
        if producer is not None:
            producer.join(timeout=10.0)
    if pipe is not None:
        pipe.release()
    torch.cuda.current_stream().wait_stream(stream)
    torch.cuda.synchronize()
    if rank == 0 and t_first is not None and i > i_first:
        dt = (time.time() - t_first) / (i - i_first)
        rep = dict(batches=i, batches_pipelined=n_piped, batch_size=a.batch_size, num_point=a.num_point, prefetch=not a.no_prefetch,
                   prefetch_group=G, graph_metrics=metrics_graph is not None, fused_metrics=fused is not None, ms_per_batch_after_first=dt * 1e3, points_per_s=a.batch_size * a.num_point / dt,
                   ms_per_batch_waiting_for_loader=waited[0] / (i - i_first) * 1e3)
        print("evaluation throughput: %.3f ms/batch (forward + metrics + one host transfer per batch; %.3f of it waiting for the loader thread), "
              "%.1f points/s, %d of %d batches pipelined" % (rep["ms_per_batch_after_first"], rep["ms_per_batch_waiting_for_loader"], rep["points_per_s"], n_piped, i))
        if a.report:
            import json
            with open(a.report, "w") as f:
                json.dump(rep, f)
    tot, n = acc.sums()
    tot, n = torch.tensor(tot), torch.tensor([n], dtype=torch.float64)
    if world > 1:                                      # the only exchange of an evaluation run: the metric sums
        import torch.distributed as dist
        tot, n = tot.to(dev), n.to(dev)
        dist.all_reduce(tot)
        dist.all_reduce(n)
        tot, n = tot.cpu(), n.cpu()
    means = dict(zip(acc.keys, (tot / n.clamp(min=1)).tolist()))
    if rank == 0:
        lines = ["=" * 20, "", "Num evaluated= %d" % int(n.item()), ""]
        for k, txt in REPORT:
            lines += [txt + str(means[k]), ""]
        if a.with_sketch_fit:
            lines += ["Mean per-extrusion cylinder fitting loss= " + str(means["pred_fit_cyl_loss"]), "",
                      "Mean global fitting loss= " + str(means["pred_fit_glob_loss"]), ""]
        for ln in lines:
            print(ln)
            log.write(ln + "\n")
    log.close()
    if world > 1:
        torch.distributed.destroy_process_group()
    return means


if __name__ == "__main__":
    main()
    sys.exit(0)
