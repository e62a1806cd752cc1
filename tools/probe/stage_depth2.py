"""NS-1 stage (FPS + ball query + grouped MLP forward, B = 32 x 8192) as a THROUGHPUT pipeline: the geometry of the next G batches computed
TOGETHER (one FPS launch over G x 32 clouds: G x 32 workgroups, the same ~0.52 ms of dependent steps) on a forked stream while the grouped
MLP of the current G batches runs batch by batch (BatchNorm statistics per 32 clouds, as the reference).   python tools/probe/stage_depth2.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from point2cyl_amd import measure, ops, synth
from point2cyl_amd.backbone import backbone
dev = torch.device("cuda:0")
B, N = 32, 8192
torch.manual_seed(0)
model = backbone(output_sizes=[3, 16]).to(dev).train()
sa1 = model.sa1
for G in (1, 2, 3, 4, 6, 8):
    xyz = torch.cat([synth.make_batch(B, N, 8, seed=10 + j)[0].float() for j in range(G)]).to(dev)          # (G*B, N, 3)
    start = torch.randint(0, N, (G * B,)).to(dev)
    sa1.fps_start = start

    def geometry():
        g = sa1.geometry(xyz, with_csr=False)
        g["X0"] = ops.group_gather(xyz, None, g["new_xyz"], g["group_idx"], None)
        return g

    def mlp(g, j):
        ops.step_done()
        sl = {k: v[j * B:(j + 1) * B] if k != "X0" else v[j * B * 512 * 64:(j + 1) * B * 512 * 64] for k, v in g.items()}
        with torch.no_grad(), ops.step_arena(dev):
            return sa1.forward_pm(xyz[j * B:(j + 1) * B], None, sl)

    with measure._KeepBuffers(model):
        g0 = geometry()
        cur = {k: v.clone() for k, v in g0.items()}
        side = torch.cuda.Stream()

        def piped():
            cap = torch.cuda.current_stream()
            side.wait_stream(cap)
            with torch.cuda.stream(side):
                nxt = geometry()
            outs = [mlp(cur, j) for j in range(G)]
            cap.wait_stream(side)
            ks = sorted(cur)
            ops.copy_flat_batch([cur[k] for k in ks], [nxt[k] for k in ks])
            return outs

        gr, _ = measure.capture(piped)
        t = measure.replay_ms([gr], 30)
        gr_geo, _ = measure.capture(geometry)
        tg = measure.replay_ms([gr_geo], 30)
        ops.step_done()
    sa1.fps_start = None
    floor = 0.1664
    print("G = %d batches per geometry launch: %.3f ms per group = %.3f ms per batch of 32 clouds (%.1f M points/s), %.3f of the stage's fp32-MFMA roofline; "
          "geometry of the group alone %.3f ms" % (G, t, t / G, B * N / (t / G * 1e-3) / 1e6, floor / (t / G), tg))
