"""Trainer counterpart of the reference's train_Point2Cyl.py (the with-sketch trainer: flags :33-90, networks and optimiser
:256-321, checkpoint loading :327-350, loop :369-700, checkpoints :746-775) on the HIP kernels.

Per step (default path: predicted labels; --use_whole_pc, --use_extrusion_axis_feat and --use_gt_im select the reference's other encoder
inputs, train_Point2Cyl.py:268-276, :519-600; --is_implicitnet_train is parsed and never read there (:75, :144) - accepted, no effect):
backbone forward + segmentation / normal / base-barrel (+ axis, + centre) losses exactly as the without-sketch step
(point2cyl_amd/step.py), then the sketch branch (point2cyl_amd/step_sketch.py, train_Point2Cyl.py:519-672): projection of the
predicted and the ground-truth barrels, trainable sketch encoder vs the frozen pre-trained one (latent loss, angle or --is_L2),
and with --with_im_loss the implicit decoder's manifold / eikonal / SALD-normal losses (double backward through the decoder).
`total = (point-cloud losses if --is_pc_train) + im_loss`; Adam over the parameter groups the reference builds
({backbone at --learning_rate} and/or {encoder at StepLearningRateSchedule(0.001, 1000, 0.5)(0)}, the decoder and the loaded
encoder frozen, :298-321, :364-365); BatchNorm-momentum and learning-rate staircases applied after the forward like the reference.

Checkpoints are the reference's triple {"model", "implicit_net", "pn_encoder"} (:348, :760); --is_pc_init loads
<pc_logdir>/<pc_ckpt>["model"] (this package's without-sketch trainer writes that file), --is_im_init / the frozen networks load
<im_logdir>/<im_ckpt>["model_state_dict" | "encoder_state_dict"] when the file exists (there is no pre-trained IGR checkpoint on
the box: without it the frozen networks keep their seeded initialisation and a warning is logged).

Data: --synthetic N generates N clouds plus their ground-truth sketches (dataloader.py's `sampled_sketch`, (K, S, 4) = [2-D point |
2-D normal]: here the ground-truth barrels projected along the ground-truth axes, scaled to unit radius); resident in HBM like the
without-sketch trainer's data.  One process per GPU under torch.distributed.run (clouds sharded, one flat gradient exchange).

    python -m point2cyl_amd.train_sketch --pred_seg --pred_normal --pred_bb --is_pc_train --is_im_train --with_im_loss \
        --synthetic 32 --batch_size 8 --num_epochs 1
"""
import argparse
import json
import os
import sys
import time
from collections import defaultdict

import numpy as np
import torch
import torch.nn.functional as F

from . import ddp, fitting, ops, optim, step, step_sketch, synth
from . import hostmem
from .backbone import backbone
from .implicit import ImplicitNet, NormalPerPoint
from .sketch import PointNetEncoder
from .train import ResidentDataset

PC_SCALARS = ("normal", "miou", "ext", "bb", "center")
IM_SCALARS = ("im_loss", "latent_loss", "mnfld_loss", "grad_loss", "normals_loss")


def step_lr(initial, interval, factor, epoch):
    """IGR/general.py:70-77 StepLearningRateSchedule."""
    return float(np.maximum(initial * (factor ** (epoch // interval)), 5.0e-6))


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument("--model", type=str, default="pointnet_extrusion")
    p.add_argument("--num_point", type=int, default=8192)
    p.add_argument("--num_sk_point", type=int, default=2048)
    p.add_argument("--K", type=int, default=8)
    p.add_argument("--batch_size", type=int, default=4)
    p.add_argument("--logdir", default="Point2Cyl", type=str)
    p.add_argument("--data_dir", type=str, default="data/")
    p.add_argument("--data_split", default="train", type=str)
    p.add_argument("--num_epochs", type=int, default=300)
    p.add_argument("--decay_step", type=int, default=200000)
    p.add_argument("--bn_decay_step", type=int, default=200000)
    p.add_argument("--decay_rate", type=float, default=0.7)
    p.add_argument("--learning_rate", type=float, default=0.001)
    p.add_argument("--momentum", type=float, default=0.9)
    for f in ("pred_seg", "pred_normal", "pred_bb", "pred_extrusion", "pred_center", "norm_eig", "add_noise", "sald", "is_pc_init", "is_im_init",
              "is_pc_train", "is_im_train", "is_implicitnet_train", "is_L2", "with_im_loss", "use_whole_pc", "use_gt_im", "use_extrusion_axis_feat"):
        p.add_argument("--" + f, action="store_true")
    for f in ("seg", "normal", "bb", "extrusion", "center"):
        p.add_argument("--weight_" + f, type=float, default=1.0)
    p.add_argument("--noise_sigma", type=float, default=0.01)
    p.add_argument("--pc_logdir", default="Point2Cyl_without_sketch", type=str)
    p.add_argument("--pc_ckpt", default="model.pth", type=str)
    p.add_argument("--im_logdir", default="./results/IGR_dense/", type=str)
    p.add_argument("--im_ckpt", default="latest.pth", type=str)
    p.add_argument("--synthetic", type=int, default=0, help="number of generated shapes with ground-truth sketches (0: read <data_dir>/<split>.h5)")
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--save_every", type=int, default=10)
    p.add_argument("--max_steps", type=int, default=0)
    p.add_argument("--no_graph", action="store_true", help="launch every kernel of the step from Python (and draw the projections' barrel samples on the "
                   "CPU generator, as the reference does) instead of replaying a HIP graph")
    p.add_argument("--report", type=str, default="")
    return p


def synthetic_sketches(data, K, S, dev, chunk=16):
    """Ground-truth sketches of the resident clouds: the ground-truth barrels projected along the ground-truth axes about the
    ground-truth centres, divided by their scale (what utils.py stores per extrusion as `sketch` after normalisation), with the
    projected normals; (n, K, S, 4).  Absent segments keep zeros."""
    out = []
    for i in range(0, len(data), chunk):
        idx = torch.arange(i, min(i + chunk, len(data)), device=dev)
        pcs, nrm, inst, bb, _, _, axes, _, cen = data.gather(idx)
        P0, X0, s0, found = fitting.sketch_implicit_projection2(pcs, nrm, inst, bb, axes, cen, S)
        sk = torch.cat([P0 / s0.clamp(min=1e-12).unsqueeze(-1).unsqueeze(-1), F.normalize(X0 + 1e-6, dim=-1)], -1).permute(1, 0, 2, 3)
        out.append(sk * found.unsqueeze(-1).unsqueeze(-1))
    return torch.cat(out).contiguous()


def main(argv=None):
    a = build_parser().parse_args(argv)
    hostmem.setup_cli()        # large host blocks stay mapped after free; torch's CPU pool sized to the cgroup's CPU quota (hostmem.py)
    if a.use_extrusion_axis_feat and not a.use_whole_pc:
        raise SystemExit("--use_extrusion_axis_feat only changes the --use_whole_pc encoder input (train_Point2Cyl.py:271-276, :528, :577)")
    if a.use_extrusion_axis_feat and not (a.use_gt_im or a.pred_extrusion):
        raise SystemExit("--use_extrusion_axis_feat feeds the FITTED axes to the encoder: it needs --pred_extrusion (train_Point2Cyl.py:446, :528) or --use_gt_im")
    if a.use_gt_im and a.is_pc_train:
        raise SystemExit("--use_gt_im skips the backbone (train_Point2Cyl.py:405, :566): there is nothing of it to train; drop --is_pc_train")
    if not (a.is_pc_train or a.is_im_train):
        raise SystemExit("nothing to train: pass --is_pc_train and/or --is_im_train (train_Point2Cyl.py:298-321)")
    rank, world, local = ddp.init_from_env()
    if not torch.cuda.is_available():
        raise SystemExit("point2cyl_amd.train_sketch needs an MI355X (HIP) device; there is no CPU path")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    with torch.cuda.stream(torch.cuda.Stream(dev)):
        return _main(a, rank, world, dev)


def _main(a, rank, world, dev):
    K, S, N = a.K, a.num_sk_point, a.num_point
    torch.manual_seed(a.seed)
    fl = step.StepFlags(K=K, pred_seg=a.pred_seg, pred_normal=a.pred_normal, pred_bb=a.pred_bb, pred_extrusion=a.pred_extrusion,
                        pred_center=a.pred_center, norm_eig=a.norm_eig, weight_seg=a.weight_seg, weight_normal=a.weight_normal,
                        weight_bb=a.weight_bb, weight_extrusion=a.weight_extrusion, weight_center=a.weight_center)
    if not a.use_gt_im and not (fl.pred_seg and fl.pred_bb and fl.pred_normal):
        raise SystemExit("the sketch branch needs --pred_seg --pred_normal --pred_bb (labels, base/barrel split and normals feed the projection)")
    model = backbone(output_sizes=fl.pred_sizes()).to(dev)                                                   # train_Point2Cyl.py:256
    implicit_net = ImplicitNet(d_in=2 + 256, dims=[512] * 8, skip_in=[4], geometric_init=True, radius_init=1, beta=100).to(dev)   # :268
    if a.use_whole_pc:                                                                                       # :268-276: [xyz | membership (| axis)] of the whole cloud
        pn_encoder = PointNetEncoder(256, 7 if a.use_extrusion_axis_feat else 4, with_normals=False).to(dev)
    else:
        pn_encoder = PointNetEncoder(256, 2, with_normals=True).to(dev)                                      # :270
    loaded_pn_encoder = PointNetEncoder(256, 2, with_normals=True).to(dev)                                   # :280
    for m_ in (model, implicit_net, pn_encoder, loaded_pn_encoder):
        ddp.broadcast_module(m_)
    ddp.preflight([model, implicit_net, pn_encoder, loaded_pn_encoder], dev)      # world > 1: eager collective + identical replicas (raises otherwise)
    groups = []
    if a.is_pc_train:
        groups.append({"params": list(model.parameters()), "lr": a.learning_rate})                           # :298-311
    if a.is_im_train:
        groups.append({"params": list(pn_encoder.parameters()), "lr": step_lr(0.001, 1000, 0.5, 0)})
    opt = optim.Adam(groups)
    log = None
    if rank == 0:
        os.makedirs(a.logdir, exist_ok=True)
        log = open(os.path.join(a.logdir, "log.txt"), "w")
        log.write(str(a) + "\n")

    def say(msg):
        if rank == 0:
            print(msg)
            log.write(msg + "\n")

    if a.is_pc_init:                                                                                         # :327-330
        model.load_state_dict(torch.load(os.path.join(a.pc_logdir, a.pc_ckpt), map_location="cpu")["model"])
        say("3D model loaded.")
    im_file = os.path.join(a.im_logdir, a.im_ckpt)
    if os.path.exists(im_file):                                                                              # :332-342
        ck = torch.load(im_file, map_location="cpu")
        if a.is_im_init:
            pn_encoder.load_state_dict(ck["encoder_state_dict"])
            say("Implicit model loaded.")
        implicit_net.load_state_dict(ck["model_state_dict"])
        loaded_pn_encoder.load_state_dict(ck["encoder_state_dict"])
        say("Pre-trained fixed implicit model loaded.")
    else:
        say("WARNING: %s not found - the frozen decoder / ground-truth encoder keep their seeded initialisation" % im_file)
    for p in list(implicit_net.parameters()) + list(loaded_pn_encoder.parameters()):
        p.requires_grad_(False)
    if not a.is_pc_train:
        for p in model.parameters():
            p.requires_grad_(False)
    if not a.is_im_train:
        for p in pn_encoder.parameters():
            p.requires_grad_(False)

    def save(name):
        ddp.average_buffers(model)
        ddp.average_buffers(pn_encoder)
        if rank == 0:
            torch.save({"model": model.state_dict(), "implicit_net": implicit_net.state_dict(), "pn_encoder": pn_encoder.state_dict()},
                       os.path.join(a.logdir, name))                                                         # :348, :760
    save("model.pth")                                                                                        # :345-349: the initial combined model
    model.train() if a.is_pc_train else model.eval()                                                         # :352-365
    pn_encoder.train() if a.is_im_train else pn_encoder.eval()
    implicit_net.eval()
    loaded_pn_encoder.eval()
    np.random.seed(0 + rank)
    torch.manual_seed(a.seed + 7919 * rank)
    file_sketches = None
    if a.synthetic > 0:
        ds = synth.SyntheticExtrusionDataset(a.synthetic, N, K, seed=1234)
    else:
        # train_Point2Cyl.py:215: AutodeskDataset_h5_sketches(H5_FILENAME, NUM_POINT, NUM_SK_POINT, K, op=False, center=True, extent=False).
        # The resident loader keeps every cloud whole (the per-step subsample is drawn on the device); the sketches come with it
        from .h5data import AutodeskH5Sketches, dataset_path
        # and are kept WHOLE as well: the reference redraws its num_sk_point subset on every access (dataloader.py:211-214), so the
        # subset is drawn per step on the device below, not frozen at load time
        ds = AutodeskH5Sketches(dataset_path(a.data_dir, a.data_split), None, None, K, op=False, center=True, extent=False)
        if ds.pcs.shape[1] < N:
            raise SystemExit("--num_point %d exceeds the %d points per cloud of the dataset (the reference prints an error and returns short "
                             "items, dataloader.py:72-73; the static-shape trainer refuses)" % (N, ds.pcs.shape[1]))
        if ds.sketches.shape[2] < S:
            raise SystemExit("--num_sk_point %d exceeds the %d points per sketch of the dataset" % (S, ds.sketches.shape[2]))
        file_sketches = True
    lo, hi = ddp.shard_range(len(ds), rank, world)
    if file_sketches:
        items = [ds[i] for i in range(lo, hi)]
        file_sketches = torch.from_numpy(np.stack([np.asarray(it[9]) for it in items])).to(dev, torch.float)        # (n, K, S_all, 4)

        class _Nine(torch.utils.data.Dataset):          # the 9 fields of the sketch-free item (the sketch is item 9)
            def __len__(self):
                return len(items)

            def __getitem__(self, i):
                return items[i][:9]
        data = ResidentDataset(_Nine(), dev, N, subsample=True)
        sketches = file_sketches
    else:
        data = ResidentDataset(torch.utils.data.Subset(ds, range(lo, hi)) if world > 1 else ds, dev, N)
        sketches = synthetic_sketches(data, K, S, dev)
    B = min(a.batch_size, len(data))
    nb = len(data) // B
    if world > 1:
        t = torch.tensor([nb], device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MIN)
        nb = int(t.item())
    if nb == 0:
        raise SystemExit("dataset shard (%d clouds) is smaller than one batch (%d)" % (len(data), B))
    trainable = [p for g in groups for p in g["params"]]
    sync = ddp.FlatGradSync(trainable, world)
    sampler = NormalPerPoint(1.8, 0.01)                                                                       # :262-266
    gstep, best, old_lr = 0, np.inf, a.learning_rate
    mom_fwd = 0.1
    scal = defaultdict(list)
    t0, steps_timed = None, 0

    # ---- one step on STATIC tensors: launched from Python, or - backbone trained, default - replayed as ONE HIP graph (forward, losses, the
    # decoder's double backward, gradient packing) with the next batch's FPS / ball query / 3-NN on a forked stream inside it, as
    # point2cyl_amd.train does (graph.GraphedForwardBackward).  The replayed step draws the projections' barrel samples from the device
    # generator (fitting.barrel_draws_on_device: the reference's CPU draws need the barrel counts on the host); --no_graph keeps the
    # reference's draws.  BatchNorm momentum is a captured constant: the graph is rebuilt when the staircase moves.
    f32, i64 = torch.float32, torch.int64
    st_batch = [torch.zeros(B, N, 3, device=dev), torch.zeros(B, N, 3, device=dev), torch.zeros(B, N, dtype=i64, device=dev),
                torch.zeros(B, N, dtype=i64, device=dev), torch.zeros(B, K, 3, device=dev), torch.zeros(B, K, 3, device=dev),
                torch.zeros(B, K, S, 4, device=dev)]
    st_next_xyz = torch.zeros(B, N, 3, dtype=f32, device=dev)
    use_graph = [not a.no_graph and a.is_pc_train and not a.use_gt_im]
    graph_state = dict(graph=None, momentum=None, captures=0)

    def fwd_bwd(geom=None, device_draws=False):
        pcs, nrm, inst, bb, axes, cen, gt_sk = st_batch
        ops.step_done()
        with ops.step_arena(dev):
            if a.use_gt_im:                                                # :405, :566-600: no backbone pass, ground-truth labels feed the encoder
                zero = torch.zeros((), device=dev)
                out = dict(total=zero, normal=zero, miou=zero, ext=zero, bb=zero, center=zero)
                sk = step_sketch.sketch_branch_losses(pcs, None, None, None, None, None, nrm, inst, bb, axes, cen, gt_sk, pn_encoder,
                                                      loaded_pn_encoder, implicit_net, sampler, K, S, with_im_loss=a.with_im_loss, is_l2=a.is_L2,
                                                      use_whole_pc=a.use_whole_pc, use_gt_im=True,
                                                      axis_feat=axes if a.use_extrusion_axis_feat else None)
            else:
                out = (step.compute_losses_fused if step.fused_loss_applicable(fl) else step.compute_losses)(model, pcs, nrm, inst, bb, axes, cen, fl,
                                                                                                             geom=geom)
                h = out["heads"].view(B, N, -1) if "heads" in out else None
                with torch.no_grad():
                    if h is not None:
                        X = F.normalize(h[:, :, 0:3], p=2, dim=2, eps=1e-12)
                        W2K = torch.softmax(h[:, :, 3:3 + 2 * K], dim=2)
                    else:
                        X, W2K = out["X"].detach(), torch.softmax(out["W_raw"].detach(), dim=2)
                    W = W2K[:, :, 0::2] + W2K[:, :, 1::2]
                W_enc = None
                if a.use_whole_pc and a.is_pc_train:     # the membership channel keeps its history: the sketch losses reach the backbone (:519-536)
                    Wg = torch.softmax(h[:, :, 3:3 + 2 * K], dim=2) if h is not None else torch.softmax(out["W_raw"], dim=2)
                    W_enc = Wg[:, :, 0::2] + Wg[:, :, 1::2]
                ax_feat = None
                if a.use_extrusion_axis_feat:            # :528: the fitted axes, with their history when the backbone trains
                    ax_feat = out["E_AX"] if a.is_pc_train else out["E_AX"].detach()
                sk = step_sketch.sketch_branch_losses(pcs, X, W, W2K, out["match"], out["mask"], nrm, inst, bb, axes, cen, gt_sk, pn_encoder,
                                                      loaded_pn_encoder, implicit_net, sampler, K, S, with_im_loss=a.with_im_loss, is_l2=a.is_L2,
                                                      use_whole_pc=a.use_whole_pc, W_encoder=W_enc, axis_feat=ax_feat, device_draws=device_draws)
            total = (out["total"] + sk["im_loss"]) if a.is_pc_train else sk["im_loss"]                    # :690-693
            sync.zero()
            total.backward()
            sync.pack()
        row = torch.stack([total.detach()] + [sk[k].detach().float().reshape(()) for k in IM_SCALARS] +
                          [out[k].detach().float().reshape(()) for k in PC_SCALARS])
        return dict(row=row)

    def run_step(momentum, eager=False):
        """One optimizer step on the static batch; momentum = the BatchNorm momentum of this step's forward."""
        step.update_momentum(model, momentum)
        if eager or not use_graph[0]:
            out = fwd_bwd()
        else:
            gs = graph_state
            if gs["graph"] is None or gs["momentum"] != momentum:
                from .graph import GraphedForwardBackward
                if gs["graph"] is not None:
                    gs["graph"].release()
                    gs["graph"] = None
                # warm-up passes and the capture run the step three times: the sketch encoder's BatchNorm statistics are put back afterwards
                # (GraphedForwardBackward restores the backbone's itself); its first replay trains on the geometry of what the prefetch
                # buffer holds at construction, which must be the current batch
                keep = [(b_, b_.detach().clone()) for b_ in pn_encoder.buffers()]
                nxt = st_next_xyz.clone()
                st_next_xyz.copy_(st_batch[0])
                try:
                    gs["graph"] = GraphedForwardBackward(model, lambda geom=None: fwd_bwd(geom, device_draws=True), prefetch_xyz=st_next_xyz,
                                                         stream=torch.cuda.current_stream())
                except Exception as e:      # something in this configuration cannot be captured: train on, launched from Python
                    sys.stderr.write("point2cyl_amd.train_sketch: HIP graph capture failed (%s: %s); continuing without the graph\n" % (type(e).__name__, e))
                    fresh = torch.cuda.Stream(dev)                          # the capture stream may be left in capture mode
                    fresh.wait_stream(torch.cuda.current_stream())
                    torch.cuda.set_stream(fresh)
                    for m_ in model.modules():
                        if hasattr(m_, "fps_start"):
                            m_.fps_start = None
                    use_graph[0], gs["graph"] = False, None
                    with torch.no_grad():
                        for b_, v_ in keep:
                            b_.copy_(v_)
                    st_next_xyz.copy_(nxt)
                    return run_step(momentum, eager=True)
                with torch.no_grad():
                    for b_, v_ in keep:
                        b_.copy_(v_)
                st_next_xyz.copy_(nxt)
                gs["momentum"] = momentum
                gs["captures"] += 1
            out = gs["graph"]()
        sync.allreduce()
        opt.step()
        ops.step_done()
        return out["row"]

    def batches():
        for epoch_ in range(1, a.num_epochs + 1):
            perm = torch.randperm(len(data)).to(dev)
            for i_ in range(nb):
                idx = perm[i_ * B:(i_ + 1) * B]
                it_ = data.gather(idx, fields=ResidentDataset.USED)
                pcs, nrm, inst, bb, axes, cen = it_[0], it_[1], it_[2], it_[3], it_[6], it_[8]
                gt_sk = sketches.index_select(0, idx)
                if gt_sk.shape[2] != S:      # dataset sketches: a fresh num_sk_point subset per item and step (dataloader.py:211-214), drawn on the device
                    sel = torch.rand(B, gt_sk.shape[2], device=dev).argsort(dim=1)[:, :S]
                    gt_sk = torch.gather(gt_sk, 2, sel.view(B, 1, S, 1).expand(B, K, S, 4))
                if a.add_noise:
                    pcs = pcs + torch.randn(B, N, 1, device=dev) * a.noise_sigma * nrm
                yield epoch_, i_, [pcs.float(), nrm.float(), inst.long(), bb.long(), axes.float(), cen.float(), gt_sk.float()]

    log_ring = [torch.zeros(1 + len(IM_SCALARS) + len(PC_SCALARS), dtype=torch.float32).pin_memory() for _ in range(2)]
    lagged = None

    def emit(item):
        ep_, i_, hb_, ev_ = item
        ev_.synchronize()
        v = hb_.tolist()
        hist.append(torch.tensor(v))
        say("Epoch: %d/%d | Batch [%04d/%04d] | total loss: %.4f | latent loss: %.4f | manifold loss: %.4f | eikonal loss: %.4f | normal loss: %.4f"
            % (ep_, a.num_epochs, i_, nb, v[1], v[2], v[3], v[4], v[5]))
        if a.is_pc_train:
            say("Epoch: %d/%d | Batch [%04d/%04d] | total loss: %.4f | normal loss: %.4f | mIOU loss: %.4f | ext loss: %.4f | bb loss: %.4f | center loss: %.4f"
                % (ep_, a.num_epochs, i_, nb, v[0], v[6], v[7], v[8], v[9], v[10]))

    stream_it = batches()
    cur = next(stream_it, None)
    hist = []
    while cur is not None:
        nxt = next(stream_it, None)          # one batch ahead: its clouds are the ones whose geometry the replayed step prefetches
        epoch, i, tensors = cur
        ops.copy_flat_batch(st_batch, [t.contiguous() for t in tensors])
        st_next_xyz.copy_(nxt[2][0] if nxt is not None else tensors[0])
        # the staircases count SAMPLES: under data parallelism a step consumes world * B of them (as point2cyl_amd.train does)
        lr = step.get_learning_rate(a.learning_rate, gstep, B * world, a.decay_step, a.decay_rate)     # :703-706: group 0 only
        if old_lr != lr:
            opt.param_groups[0]["lr"] = lr
            old_lr = lr
        row = run_step(mom_fwd, eager=gstep == 0)          # (the first step's momentum differs from every later one: not worth a capture)
        mom_fwd = step.get_batch_norm_decay(gstep, B * world, a.bn_decay_step)                        # :698-701 (reaches the next forward)
        gstep += 1
        # The reference prints every step from ~12 .item() syncs.  Here step k's scalars go to a pinned host buffer asynchronously and its two
        # lines are printed after step k+1 has been ENQUEUED (the host waits for an event that precedes step k+1's work): the log never
        # drains the device queue.  (The copy also un-aliases the row: in graph mode every replay returns the same static tensor.)
        hb = log_ring[gstep % 2]
        hb.copy_(row, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        if lagged is not None:
            emit(lagged)
        lagged = (epoch, i, hb, ev)
        if gstep == 3:                                     # steady state: steps 0 and 1 are the eager one and the capture
            torch.cuda.synchronize()
            t0, steps_timed = time.perf_counter(), 0
        elif gstep > 3:
            steps_timed += 1
        stopping = bool(a.max_steps) and gstep >= a.max_steps
        end_of_epoch = nxt is None or nxt[0] != epoch or stopping
        cur = nxt
        if not end_of_epoch:
            continue
        emit(lagged)                                       # the epoch's last lines in front of its summary
        lagged = None
        ep = torch.stack(hist).mean(0).tolist()
        for k, val in zip(("total_loss",) + tuple("IM_" + s for s in IM_SCALARS) + PC_SCALARS, ep):
            scal[k].append(val)
        hist = []
        last = nxt is None or stopping
        if epoch % a.save_every == 0 or last:
            save("checkpoint_%04d.pth" % epoch)
            save("model.pth")
            if epoch > 20 and ep[0] < best:
                best = ep[0]
                save("best_model.pth")
            say("> Epoch [%04d/%04d] | total_loss: %.4f | IM_total_loss: %.4f | IM_latent_loss: %.4f" % (epoch, a.num_epochs, ep[0], ep[1], ep[2]))
        if last:
            break
    torch.cuda.synchronize()
    multi = None
    if world > 1:
        # as point2cyl_amd.train: identical replicas after the last step (trained modules, float64 sum of squares, gathered), the schedules of
        # the GLOBAL batch, the checkpoint's BatchNorm statistics = the replicas' mean, the size of the one exchange per step
        with torch.no_grad():
            ck = sum(float((p.detach().double() ** 2).sum()) for m_ in (model, pn_encoder) for p in m_.parameters())
            bk = sum(float(b_.detach().double().sum()) for m_ in (model, pn_encoder) for b_ in m_.buffers() if b_.dtype.is_floating_point)
        mine = torch.tensor([ck, bk, old_lr, mom_fwd], dtype=torch.float64, device=dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        torch.distributed.all_gather(allr, mine)
        allr = torch.stack(allr).cpu()
        multi = dict(backend=ddp.backend_name(), param_checksum=[float(v) for v in allr[:, 0]], params_identical=bool((allr[:, 0] == allr[0, 0]).all()),
                     buffer_checksum_after_averaging=[float(v) for v in allr[:, 1]], buffers_identical=bool((allr[:, 1] == allr[0, 1]).all()),
                     learning_rate=[float(v) for v in allr[:, 2]], next_bn_momentum=[float(v) for v in allr[:, 3]], samples_per_step=B * world,
                     allreduce_bytes=int(sync.flat.numel() * 4) if sync.flat is not None else 0, trainable_parameters=sum(p.numel() for p in trainable))
    if t0 is not None and steps_timed > 0 and rank == 0:
        dt = (time.perf_counter() - t0) / steps_timed
        rep = dict(steps=gstep, steady_steps=steps_timed, ms_per_step=dt * 1e3, points_per_s=world * B * N / dt, batch_per_gpu=B, num_point=N, num_sk_point=S,
                   world=world, graph=graph_state["graph"] is not None, graph_captures=graph_state["captures"],
                   epoch_means={k: v for k, v in scal.items()}, multi_gpu=multi)
        print("with-sketch trainer throughput: %.2f ms/step, %.1f points/s" % (rep["ms_per_step"], rep["points_per_s"]))
        if a.report:
            with open(a.report, "w") as f:
                json.dump(rep, f)
    if graph_state["graph"] is not None:
        graph_state["graph"].release()
    if log is not None:
        log.close()
    if world > 1:
        torch.distributed.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
