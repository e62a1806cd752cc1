"""Trainer counterpart of the reference's train_Point2Cyl_without_sketch.py: same flags, same step semantics
(train…:28-62 flags, :143-164 schedules, :213-391 loop, :395-430 checkpoints), running on the HIP kernels — and running the SAME
launch path `bench.py` times: forward + losses + backward replayed as one HIP graph with the next batch's FPS / ball query /
3-NN on a forked stream (point2cyl_amd/graph.py), one flat gradient exchange, fused Adam.

Order of the per-step host actions follows the reference: forward (with the BatchNorm momentum set by the PREVIOUS step; the very
first forward runs with the modules' constructor momentum 0.1, train…:207, :355-360), then the BatchNorm-momentum and
learning-rate staircase updates, backward, optimizer step.  BatchNorm momentum is a scalar kernel argument, so the graph is
captured at step 1 and re-captured when the staircase changes it (every --bn_decay_step samples); step 0 runs eagerly.

Data: the whole dataset is RESIDENT IN HBM (a Fusion-Gallery-sized set of 8k clouds x 8192 points is 3.7 GB of 288 GB): an epoch's
shuffled batches are index gathers on the device into the static tensors the graph reads; nothing crosses PCIe after start-up.
With --synthetic N the N generated clouds are built once on the host; with data/<split>.h5 (needs h5py) the file is read once
and each access draws its num_point-subsample on the device (dataloader.py:71-77 draws it on the host, per item).

Additions over the reference: --synthetic N, --no_graph (launch every kernel from Python), --max_steps, and one-process-per-GPU
data parallelism when launched with torch.distributed.run (batch sharded by cloud, one gradient all-reduce per step, per-replica
BatchNorm statistics like N independent reference runs on the shards, averaged into the checkpoint; parameters initialised from
the same seed and broadcast, data / noise / FPS-start / dropout generators seeded with seed + rank).

    python -m point2cyl_amd.train --pred_seg --pred_normal --pred_bb --synthetic 64 --batch_size 32 --num_epochs 1
"""
import argparse
import json
import os
import sys
import time
from collections import defaultdict

import numpy as np
import torch

from . import ops, ddp, optim, step, synth
from . import hostmem
from .backbone import backbone

SCALARS = ("total", "normal", "miou", "bb", "ext", "center")


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
    p.add_argument("--no_graph", action="store_true", help="launch every kernel from Python instead of replaying a HIP graph")
    p.add_argument("--async_exchange", action="store_true", help="data-parallel jobs: the gradient all-reduce on a SIDE stream between the two halves of a "
                   "split-tail graph (the next batch's geometry copies under it) instead of on the step's own stream right behind ONE graph; "
                   "measured on a one-rank nccl group: +0.09 ms per step against +0.03 (the tail it can hide is 0.04 ms long, the second replay and "
                   "the event hops cost 0.06) - worth it only where the all-reduce is much longer than that")
    p.add_argument("--no_prefetch", action="store_true", help="compute FPS / ball query / 3-NN inline instead of one batch ahead on a forked stream")
    p.add_argument("--max_steps", type=int, default=0, help="stop after this many optimizer steps (0: run all epochs)")
    p.add_argument("--quiet", action="store_true", help="no per-batch log line (each costs a device synchronisation)")
    p.add_argument("--report", type=str, default="", help="write a JSON throughput / loss report here at the end")
    return p


class ResidentDataset:
    """All items of a map-style dataset stacked into device tensors once; batches are index gathers on the device."""

    def __init__(self, ds, dev, num_point, subsample=False):
        items = [ds[i] for i in range(len(ds))]                  # (no DataLoader: its iterator would draw a seed from torch's generator)
        dt = (torch.float, torch.float, torch.long, torch.long, torch.float, torch.float, torch.float, torch.float, torch.float)
        self.t = [torch.from_numpy(np.stack([np.asarray(it[j]) for it in items])).to(dev, d) for j, d in enumerate(dt)]
        self.n, self.num_point, self.subsample = self.t[0].shape[0], num_point, subsample and self.t[0].shape[1] > num_point
        self.dev = dev
        ops.check_labels(self.t[2], self.t[6].shape[1])      # once per dataset: the replayed step itself never syncs to validate

    def __len__(self):
        return self.n

    USED = (0, 1, 2, 3, 6, 8)       # clouds, normals, instance labels, base / barrel labels, axes (K), centres (K): what the trainers read

    def gather(self, idx, fields=None):
        """idx: (B,) long on the device -> the reference's 9-tuple for those clouds; `fields`: only these positions (None elsewhere - the
        per-point axes / distances of positions 4, 5 are a third of the bytes and no trainer reads them)."""
        out = [t.index_select(0, idx) if fields is None or j in fields else None for j, t in enumerate(self.t)]
        if self.subsample:                     # dataloader.py:71-77: a fresh random num_point-subset per access
            B, Nfull = out[0].shape[0], out[0].shape[1]
            sel = torch.rand(B, Nfull, device=self.dev).argsort(dim=1)[:, : self.num_point]
            for j in (0, 1, 2, 3, 4, 5):       # the per-point tensors
                t = out[j]
                if t is not None:
                    out[j] = torch.gather(t, 1, sel.unsqueeze(-1).expand(-1, -1, t.shape[2])) if t.dim() == 3 else torch.gather(t, 1, sel)
        return out


def load_dataset(a, rank=0):
    if a.synthetic > 0:
        return synth.SyntheticExtrusionDataset(a.synthetic, a.num_point, a.K, seed=1234), False
    from .h5data import AutodeskH5, dataset_path
    return AutodeskH5(dataset_path(a.data_dir, a.data_split), None, a.K, center=True), True


class Runner:
    """forward + losses + backward + exchange + Adam on static device tensors, eager or as a HIP-graph replay."""

    def __init__(self, model, opt, sync, fl, dev, B, N, K, use_graph=True, prefetch=True, stream=None, async_exchange=False):
        self.model, self.opt, self.sync, self.fl, self.dev = model, opt, sync, fl, dev
        self.stream = stream          # the stream the caller's loop runs on (graph warm-up and capture use it too); None: a private one
        f32, i64 = torch.float32, torch.int64
        self.batch = (torch.zeros(B, N, 3, device=dev), torch.zeros(B, N, 3, device=dev), torch.zeros(B, N, dtype=i64, device=dev),
                      torch.zeros(B, N, dtype=i64, device=dev), torch.zeros(B, K, 3, device=dev), torch.zeros(B, K, 3, device=dev))
        self.next_xyz = torch.zeros(B, N, 3, dtype=f32, device=dev)
        self.use_graph, self.prefetch = use_graph, prefetch and use_graph
        self.async_exchange = bool(async_exchange)
        self.graph, self.graph_momentum = None, None
        self.loss_fn = step.compute_losses_fused if step.fused_loss_applicable(fl) else step.compute_losses
        self.captures = 0

    def load(self, cur, nxt_xyz):
        """cur = (pcs, normals, inst, bb, axes, centers) device tensors of the batch to train on; nxt_xyz = clouds of the batch after it."""
        ops.copy_flat_batch(list(self.batch), [c.contiguous() for c in cur])      # one launch for the six tensors of a batch
        self.next_xyz.copy_(nxt_xyz if nxt_xyz is not None else cur[0])

    def _fwd_bwd(self, geom=None):
        ops.step_done()
        with ops.step_arena(self.dev):
            out = self.loss_fn(self.model, *self.batch, self.fl, geom=geom)
            self.sync.zero()
            step.backward(out)
            self.sync.pack()
        res = {"scalars": torch.stack([out[k].detach().float().reshape(()) for k in SCALARS])}
        del out
        return res

    def step(self, momentum, eager=False):
        """One optimizer step; `momentum` = the BatchNorm momentum this step's FORWARD uses.  eager=True launches this step from
        Python even in graph mode (the trainer's first step: its momentum differs from every later one, not worth a capture)."""
        step.update_momentum(self.model, momentum)
        if eager or not self.use_graph:
            out = self._fwd_bwd()
        else:
            if self.graph is None or self.graph_momentum != momentum:
                from .graph import GraphedForwardBackward
                if self.graph is not None:
                    self.graph.release()
                # the graph computes the geometry of `next_xyz` for the FOLLOWING replay; its first replay trains on the geometry of
                # what next_xyz holds at construction, which must be the current batch
                nxt = self.next_xyz.clone()
                self.next_xyz.copy_(self.batch[0])
                try:
                    self.graph = GraphedForwardBackward(self.model, self._fwd_bwd, prefetch_xyz=self.next_xyz if self.prefetch else None,
                                                        stream=self.stream, split_tail=self.sync.active and self.async_exchange)
                except Exception as e:      # something in this configuration cannot be captured: train on, launched from Python
                    import sys
                    sys.stderr.write("point2cyl_amd.train: HIP graph capture failed (%s: %s); continuing without the graph\n" % (type(e).__name__, e))
                    self.stream = torch.cuda.Stream(self.dev)                # the capture stream may be left in capture mode: everything
                    self.stream.wait_stream(torch.cuda.current_stream())    # from here on (autograd included) lives on a fresh one
                    torch.cuda.set_stream(self.stream)
                    for m in self.model.modules():
                        if hasattr(m, "fps_start"):
                            m.fps_start = None
                    self.use_graph, self.graph = False, None
                    self.next_xyz.copy_(nxt)
                    return self.step(momentum, eager=True)
                self.next_xyz.copy_(nxt)
                self.graph_momentum = momentum
                self.captures += 1
            out = self.graph()
        if self.async_exchange:
            self.sync.allreduce_async()     # N > 1: the exchange on a side stream, gated on the replay ...
            if not (eager or not self.use_graph) and self.graph is not None:
                self.graph.tail()           # ... with the rest of the step (the prefetched geometry's copies) under it
            self.sync.wait()
        else:
            self.sync.allreduce()           # N > 1: on the step's stream, right behind the one graph of the step (the default: see --async_exchange)
        self.opt.step()
        ops.step_done()
        return out["scalars"]


def main(argv=None):
    a = build_parser().parse_args(argv)
    hostmem.setup_cli()        # large host blocks stay mapped after free; torch's CPU pool sized to the cgroup's CPU quota (hostmem.py)
    rank, world, local = ddp.init_from_env()
    if not torch.cuda.is_available():
        raise SystemExit("point2cyl_amd.train needs an MI355X (HIP) device; there is no CPU path")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    # the whole loop - data gathers, the eager first step, graph warm-up, capture, replays, Adam - lives on ONE non-default stream:
    # a HIP graph cannot be captured on the default stream, and autograd binds its accumulator nodes to the stream they first ran on
    stream = torch.cuda.Stream(dev)
    with torch.cuda.stream(stream):
        return _main(a, rank, world, local, dev, stream)


def _main(a, rank, world, local, dev, stream):
    torch.manual_seed(a.seed)                                # identical parameter init on every rank (broadcast below as well)
    fl = step.StepFlags(K=a.K, pred_seg=a.pred_seg, pred_normal=a.pred_normal, pred_bb=a.pred_bb, pred_extrusion=a.pred_extrusion,
                        pred_center=a.pred_center, norm_eig=a.norm_eig, weight_seg=a.weight_seg, weight_normal=a.weight_normal,
                        weight_bb=a.weight_bb, weight_extrusion=a.weight_extrusion, weight_center=a.weight_center)
    model = backbone(output_sizes=fl.pred_sizes()).to(dev).train()
    ddp.broadcast_module(model)
    ddp.preflight(model, dev)          # world > 1: one eager all-reduce + identical replicas, before the first capture (raises with the backend's text)
    # everything random AFTER the parameters is per replica: data order / subsampling, noise, FPS starts, dropout
    np.random.seed(0 + rank)                                 # train…:135 (rank 0 reproduces the reference's stream)
    torch.manual_seed(a.seed + 7919 * rank)
    ds, subsample = load_dataset(a, rank)
    n_items = len(ds)
    lo, hi = ddp.shard_range(n_items, rank, world)
    if world > 1:
        ds = torch.utils.data.Subset(ds, range(lo, hi))
    data = ResidentDataset(ds, dev, a.num_point, subsample)
    B = min(a.batch_size, len(data))
    opt = optim.Adam(model.parameters(), lr=a.learning_rate)    # train…:204; one launch for all 123 tensors (point2cyl_amd/optim.py)
    sync = ddp.FlatGradSync(model.parameters(), world)
    if a.no_graph:
        from . import autograph
        autograph.ENABLED = False                            # --no_graph means every kernel launched from Python, also inside backbone.forward
    run = Runner(model, opt, sync, fl, dev, B, a.num_point, a.K, use_graph=not a.no_graph, prefetch=not a.no_prefetch, stream=stream, async_exchange=a.async_exchange)
    log = None
    if rank == 0:
        os.makedirs(a.logdir, exist_ok=True)
        log = open(os.path.join(a.logdir, "log.txt"), "w")
        log.write(str(a) + "\n")
    nb = len(data) // B                                      # static shapes: whole batches only (drop_last)
    if world > 1:                                            # every rank must take the same number of steps
        t = torch.tensor([nb], device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MIN)
        nb = int(t.item())
    if nb == 0:
        raise SystemExit("dataset shard (%d clouds) is smaller than one batch (%d)" % (len(data), B))

    def batches():
        """(epoch, i, (pcs, normals, inst, bb, axes, centers)) over all epochs, whole batches, shuffled per epoch."""
        for epoch in range(1, a.num_epochs + 1):
            perm = torch.randperm(len(data)).to(dev)
            for i in range(nb):
                it = data.gather(perm[i * B:(i + 1) * B], fields=ResidentDataset.USED)
                pcs = it[0]
                if a.add_noise:                               # data_utils.py:84-96 on the device: p + N(0, sigma) * normal
                    pcs = pcs + torch.randn(pcs.shape[0], pcs.shape[1], 1, device=dev) * a.noise_sigma * it[1]
                yield epoch, i, (pcs, it[1], it[2], it[3], it[6], it[8])

    gstep, best = 0, np.inf
    mom_fwd = 0.1                                            # the modules' constructor momentum: what the reference's first forward uses
    old_lr = a.learning_rate
    scal = defaultdict(list)
    it = batches()
    cur = next(it)
    t_start = t_steady = None
    steps_steady = 0
    # epoch means: a running device sum.  (Keeping the per-step tensors in a list does not work in graph mode: every replay returns the SAME
    # static output tensor, so the list would hold N aliases of the last step's scalars.)
    ep_sum, ep_n, pending = torch.zeros(len(SCALARS), dtype=torch.float32, device=dev), 0, []
    host_ring = [torch.empty(len(SCALARS), dtype=torch.float32).pin_memory() for _ in range(2)]
    lagged = None

    def emit(item):
        ep_, i_, hb_, ev_ = item
        ev_.synchronize()
        msg = ("Epoch: %d/%d | Batch [%04d/%04d] | total loss: %.4f | normal loss: %.4f | mIOU loss: %.4f | bb loss: %.4f | "
               "ext loss: %.4f | center loss: %.4f" % ((ep_, a.num_epochs, i_, nb) + tuple(hb_.tolist())))
        if rank == 0:
            print(msg)
            log.write(msg + "\n")

    while cur is not None:
        nxt = next(it, None)
        epoch, i, b = cur
        run.load(b, None if nxt is None else nxt[2][0])
        # the staircases count SAMPLES (train…:143-164: global_step * batch_size); under data parallelism a step consumes world * B of them,
        # so an N-GPU run follows the schedule of the single-GPU run with the same global batch
        lr = step.get_learning_rate(a.learning_rate, gstep, B * world, a.decay_step, a.decay_rate)     # train…:361-365: before optimizer.step
        if old_lr != lr:
            for g in opt.param_groups:
                g["lr"] = lr
            old_lr = lr
        sc = run.step(mom_fwd, eager=gstep == 0)
        mom_fwd = step.get_batch_norm_decay(gstep, B * world, a.bn_decay_step)                 # train…:356-359: reaches the NEXT forward
        gstep += 1
        ep_sum += sc                                         # stream-ordered behind the step that wrote `sc`, before the next replay overwrites it
        ep_n += 1
        if gstep == 3:                                       # steady state: graph captured (steps 0 and 1 build it)
            torch.cuda.synchronize()
            t_steady, steps_steady = time.perf_counter(), 0
        elif gstep > 3:
            steps_steady += 1
        if not a.quiet:
            # the reference prints every step from .item() syncs (train...:372-376).  Here step k's six scalars go to a pinned host buffer
            # asynchronously and the line is printed after step k+1 has been ENQUEUED: the host waits for an event that precedes step k+1's
            # work, so the per-step log never drains the GPU queue
            if lagged is not None:
                emit(lagged)
            hb = host_ring[gstep % 2]
            hb.copy_(sc, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            lagged = (epoch, i, hb, ev)
        stopping = bool(a.max_steps) and gstep >= a.max_steps    # --max_steps ends the run like the end of the data: summary + checkpoint
        last_of_epoch = nxt is None or nxt[0] != epoch or stopping
        if last_of_epoch and lagged is not None and (not a.quiet):
            emit(lagged)                                         # keep the per-batch lines in front of their epoch's summary
            lagged = None
        if last_of_epoch:
            pending.append((epoch, ep_sum / ep_n))                   # epoch means stay on the device ...
            ep_sum, ep_n = torch.zeros_like(ep_sum), 0
            saving = epoch % a.save_every == 0 or nxt is None or stopping
            if saving or not a.quiet:                            # ... until something needs them on the host (one sync for all pending epochs)
                for ep_no, ep_t in pending:
                    ep = ep_t.tolist()
                    for k, v in zip(SCALARS, ep):
                        scal[k].append(v)
                    if rank == 0:
                        msg = "> Epoch [%04d/%04d] | " % (ep_no, a.num_epochs) + " | ".join("%s: %.4f" % kv for kv in zip(SCALARS, ep))
                        print(msg)
                        log.write(msg + "\n")
                pending = []
                if rank == 0:
                    log.flush()
            if saving:
                ddp.average_buffers(model)                       # BatchNorm running statistics: mean over the replicas
                if rank == 0:
                    sd = {"model": model.state_dict()}           # same checkpoint layout as train…:408
                    torch.save(sd, os.path.join(a.logdir, "checkpoint_%04d.pth" % epoch))
                    torch.save(sd, os.path.join(a.logdir, "model.pth"))
                    if epoch > 20 and ep[0] < best:
                        best = ep[0]
                        torch.save(sd, os.path.join(a.logdir, "best_model.pth"))
        if stopping:
            break
        cur = nxt
    torch.cuda.synchronize()
    report = None
    multi = None
    if world > 1:
        # what a data-parallel run has to prove: the replicas hold the SAME parameters after the last step (sum of squares in float64,
        # gathered), followed the schedule of the global batch, and the checkpoint's BatchNorm statistics are the replicas' mean
        with torch.no_grad():
            ck = sum(float((p.detach().double() ** 2).sum()) for p in model.parameters())
            bk = sum(float(b_.detach().double().sum()) for b_ in model.buffers() if b_.dtype.is_floating_point)
        mine = torch.tensor([ck, bk, old_lr, mom_fwd], dtype=torch.float64, device=dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        torch.distributed.all_gather(allr, mine)
        allr = torch.stack(allr).cpu()
        multi = dict(backend=ddp.backend_name(), param_checksum=[float(v) for v in allr[:, 0]], params_identical=bool((allr[:, 0] == allr[0, 0]).all()),
                     buffer_checksum_after_averaging=[float(v) for v in allr[:, 1]], buffers_identical=bool((allr[:, 1] == allr[0, 1]).all()),
                     learning_rate=[float(v) for v in allr[:, 2]], next_bn_momentum=[float(v) for v in allr[:, 3]],
                     samples_per_step=B * world, allreduce_bytes=int(sync.flat.numel() * 4) if sync.flat is not None else 0)
    if t_steady is not None and steps_steady > 0:
        dt = time.perf_counter() - t_steady
        report = dict(steps=gstep, steady_steps=steps_steady, ms_per_step=dt / steps_steady * 1e3, points_per_s=world * B * a.num_point * steps_steady / dt,
                      batch_per_gpu=B, num_point=a.num_point, world=world, graph=not a.no_graph, prefetch=run.prefetch, graph_captures=run.captures,
                      per_step_log_sync=not a.quiet, epoch_means={k: v for k, v in scal.items()}, multi_gpu=multi)
        if rank == 0:
            print("trainer throughput: %.3f ms/step, %.1f points/s over %d steady steps (%d graph captures)" %
                  (report["ms_per_step"], report["points_per_s"], steps_steady, run.captures))
            if a.report:
                with open(a.report, "w") as f:
                    json.dump(report, f)
    if log is not None:
        log.close()
    if world > 1:
        torch.distributed.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
