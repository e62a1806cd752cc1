"""Trainer counterpart of the reference's train_Point2Cyl_without_sketch.py: same flags, same step semantics
(train…:28-62 flags, :143-164 schedules, :213-391 loop, :395-430 checkpoints), running on the HIP kernels.

Additions: --synthetic N (generate N extrusion-cylinder clouds instead of reading data/<split>.h5; there is no
dataset on the box), and one-process-per-GPU data parallelism when launched with torch.distributed.run
(batch sharded by cloud, one gradient all-reduce per step, per-replica BatchNorm statistics like N independent
reference runs on the shards).

    python -m point2cyl_amd.train --pred_seg --pred_normal --pred_bb --synthetic 64 --batch_size 32 --num_epochs 1
"""
import argparse
import datetime
import os
import sys
from collections import defaultdict

import numpy as np
import torch

from . import ops, ddp, step, synth
from .backbone import backbone


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument("--model", type=str, default="pointnet_extrusion")
    p.add_argument("--num_point", type=int, default=8192)
    p.add_argument("--K", type=int, default=8)
    p.add_argument("--batch_size", type=int, default=4)
    p.add_argument("--logdir", default="Point2Cyl_without_sketch", type=str)
    p.add_argument("--data_dir", type=str, default="data/")
    p.add_argument("--data_split", default="train", type=str)
    p.add_argument("--num_epochs", type=int, default=300)
    p.add_argument("--decay_step", type=int, default=200000)
    p.add_argument("--bn_decay_step", type=int, default=200000)
    p.add_argument("--decay_rate", type=float, default=0.7)
    p.add_argument("--learning_rate", type=float, default=0.001)
    p.add_argument("--momentum", type=float, default=0.9)
    for f in ("pred_seg", "pred_normal", "pred_bb", "pred_extrusion", "pred_center", "norm_eig", "add_noise"):
        p.add_argument("--" + f, action="store_true")
    for f in ("seg", "normal", "bb", "extrusion", "center"):
        p.add_argument("--weight_" + f, type=float, default=1.0)
    p.add_argument("--noise_sigma", type=float, default=0.01)
    p.add_argument("--synthetic", type=int, default=0, help="number of generated shapes (0: read <data_dir>/<split>.h5)")
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--save_every", type=int, default=10)
    return p


def load_dataset(a):
    if a.synthetic > 0:
        return synth.SyntheticExtrusionDataset(a.synthetic, a.num_point, a.K, seed=1234)
    try:
        import h5py  # noqa: F401
    except Exception as e:
        raise SystemExit("reading %s needs h5py (%s); use --synthetic N" % (os.path.join(a.data_dir, a.data_split + ".h5"), e))
    from .h5data import AutodeskH5
    return AutodeskH5(os.path.join(a.data_dir, a.data_split + ".h5"), a.num_point, a.K)


def main(argv=None):
    a = build_parser().parse_args(argv)
    rank, world, local = ddp.init_from_env()
    if not torch.cuda.is_available():
        raise SystemExit("point2cyl_amd.train needs an MI355X (HIP) device; there is no CPU path")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    np.random.seed(0)                                        # train…:135
    torch.manual_seed(a.seed)
    fl = step.StepFlags(K=a.K, pred_seg=a.pred_seg, pred_normal=a.pred_normal, pred_bb=a.pred_bb, pred_extrusion=a.pred_extrusion,
                        pred_center=a.pred_center, norm_eig=a.norm_eig, weight_seg=a.weight_seg, weight_normal=a.weight_normal,
                        weight_bb=a.weight_bb, weight_extrusion=a.weight_extrusion, weight_center=a.weight_center)
    ds = load_dataset(a)
    sampler = torch.utils.data.distributed.DistributedSampler(ds, world, rank, shuffle=True) if world > 1 else None
    loader = torch.utils.data.DataLoader(ds, batch_size=a.batch_size, num_workers=0, pin_memory=True, shuffle=sampler is None, sampler=sampler)
    model = backbone(output_sizes=fl.pred_sizes()).to(dev).train()
    ddp.broadcast_module(model)
    opt = torch.optim.Adam(model.parameters(), lr=a.learning_rate, fused=True)    # train…:204, single multi-tensor kernel
    sync = ddp.FlatGradSync(model.parameters(), world)
    if rank == 0:
        os.makedirs(a.logdir, exist_ok=True)
        log = open(os.path.join(a.logdir, "log.txt"), "w")
        log.write(str(a) + "\n")
    gstep, old_lr, old_bn, best = 0, a.learning_rate, a.momentum, np.inf
    for epoch in range(1, a.num_epochs + 1):
        if sampler is not None:
            sampler.set_epoch(epoch)
        scal = defaultdict(list)
        for i, b in enumerate(loader):
            pcs, nrm, inst, bb, _, _, axes, _, cen = b
            if a.add_noise:
                from .fitting import add_noise
                pcs = add_noise(pcs, nrm, sigma=a.noise_sigma)
            batch = (pcs.to(dev, torch.float), nrm.to(dev, torch.float), inst.to(dev, torch.long), bb.to(dev, torch.long),
                     axes.to(dev, torch.float), cen.to(dev, torch.float))
            bn_m = step.get_batch_norm_decay(gstep, pcs.shape[0], a.bn_decay_step)
            if old_bn != bn_m:
                step.update_momentum(model, bn_m)
                old_bn = bn_m
            lr = step.get_learning_rate(a.learning_rate, gstep, pcs.shape[0], a.decay_step, a.decay_rate)
            if old_lr != lr:
                for g in opt.param_groups:
                    g["lr"] = lr
                old_lr = lr
            with ops.step_arena(dev):
                out = (step.compute_losses_fused if step.fused_loss_applicable(fl) else step.compute_losses)(model, *batch, fl)
                sync.zero()
                out["total"].backward()
            sync.allreduce()
            opt.step()
            gstep += 1
            vals = torch.stack([out[k].detach() for k in ("total", "normal", "miou", "bb", "ext", "center")]).tolist()   # ONE sync
            for k, v in zip(("total_loss", "normal_loss", "mIOU_loss", "bb_loss", "ext_loss", "center"), vals):
                scal[k].append(v)
            if rank == 0:
                msg = ("Epoch: %d/%d | Batch [%04d/%04d] | total loss: %.4f | normal loss: %.4f | mIOU loss: %.4f | bb loss: %.4f | "
                       "ext loss: %.4f | center loss: %.4f" % ((epoch, a.num_epochs, i, len(loader)) + tuple(vals)))
                print(msg)
                log.write(msg + "\n")
                log.flush()
        if rank == 0 and epoch % a.save_every == 0:
            sd = {"model": model.state_dict()}                                  # same checkpoint layout as train…:408
            torch.save(sd, os.path.join(a.logdir, "checkpoint_%04d.pth" % epoch))
            torch.save(sd, os.path.join(a.logdir, "model.pth"))
            mean_total = float(np.mean(scal["total_loss"]))
            if epoch > 20 and mean_total < best:
                best = mean_total
                torch.save(sd, os.path.join(a.logdir, "best_model.pth"))
    if world > 1:
        torch.distributed.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
