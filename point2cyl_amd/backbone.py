"""PointNet++ backbone of Point2Cyl on the HIP kernels.

Mirror of the reference's module interface (models/pointnet_util.py:166-207, :270-320;
models/pointnet_extrusion.py:8-66): same class names, constructor arguments, forward signatures,
parameter registration order (so the same torch.manual_seed gives the same initial weights) and the same
state_dict keys/shapes (checkpoints interchange).  The torch layer objects only HOLD the parameters and
BatchNorm buffers; the arithmetic runs in libp2c_hip.so through point2cyl_amd.ops, in a point-major
layout (rows = points, channels contiguous) so the reference's permutes disappear.
"""
import torch
import torch.nn as nn

from . import ops


DROP_STRIDE = 0x9E3779B97F4A7C15 % (2 ** 62)      # the dropout hash seed advances by this much per forward


def _layers(convs, bns):
    out = []
    for conv, bn in zip(convs, bns):
        out.append(dict(W=conv.weight, b=conv.bias, gamma=bn.weight, beta=bn.bias,
                        bn=ops.BNState(bn.running_mean, bn.running_var, bn.num_batches_tracked,
                                       0.1 if bn.momentum is None else bn.momentum, bn.eps)))
    return out


def draw_fps_start(N, B):
    """The reference draws the FPS start on the CPU default generator even for GPU runs
    (pointnet_util.py:75: torch.randint(...).to(device)); doing the same keeps FPS bit-identical per seed."""
    return torch.randint(0, N, (B,), dtype=torch.long)


class PointNetSetAbstraction(nn.Module):
    """models/pointnet_util.py:166-207."""

    def __init__(self, npoint, radius, nsample, in_channel, mlp, group_all):
        super().__init__()
        self.npoint, self.radius, self.nsample = npoint, radius, nsample
        self.mlp_convs = nn.ModuleList()
        self.mlp_bns = nn.ModuleList()
        last = in_channel
        for co in mlp:
            self.mlp_convs.append(nn.Conv2d(last, co, 1))
            self.mlp_bns.append(nn.BatchNorm2d(co))
            last = co
        self.group_all = group_all
        self.fps_start = None      # hook: (B,) tensor of FPS start indices, or a callable (N, B) -> device tensor, instead of drawing
        self.last_aux = {}

    def geometry(self, xyz, with_csr=True):
        """Parameter-free part of sample_and_group (pointnet_util.py:122-128): FPS indices, sampled centres, ball-query
        groups.  Depends only on the coordinates, so it can be computed ahead of the step that consumes it.
        with_csr=False (inference): no inverse map of the grouping - only the backward pass reads it."""
        B, N, _ = xyz.shape
        if callable(self.fps_start):
            start = self.fps_start(N, B)
        else:
            start = self.fps_start if self.fps_start is not None else draw_fps_start(N, B)
        fps_idx, new_xyz = ops.fps(xyz, self.npoint, start)
        gidx = ops.ball_query(self.radius, self.nsample, xyz, new_xyz)
        g = dict(fps_idx=fps_idx, new_xyz=new_xyz, group_idx=gidx)
        if with_csr and self.mlp_convs[0].weight.shape[1] > 3:          # grouped FEATURES exist -> their backward wants the inverse map
            g["csr"] = ops.build_csr(gidx, N)
        return g

    def forward_pm(self, xyz, feats, geom=None, staged=None):
        """xyz (B,N,3), feats (B,N,D) or None -> new_xyz (B,S,3), new_feats (B,S,C').
        staged: operands derived from the first layer's weight, prepared by the backbone's WeightStage (keys W2 [, pre_wx]); used only on
        the code path they were laid out for."""
        B, N, _ = xyz.shape
        layers = _layers(self.mlp_convs, self.mlp_bns)
        cin = self.mlp_convs[0].weight.shape[1]
        if self.group_all:
            z = self.__dict__.get("_zero_xyz")
            if z is None or z.shape[0] != B or z.device != xyz.device:
                z = self.__dict__["_zero_xyz"] = torch.zeros(B, 1, 3, device=xyz.device)      # (constant: not re-filled every step)
            new_xyz = z
            cols = ([] if feats is None else [feats.reshape(B * N, -1)]) + [xyz.reshape(B * N, 3)]
            pad = (-cin) % 4
            if pad:
                cols.append(torch.zeros(B * N, pad, device=xyz.device))
            X0 = torch.cat(cols, 1)                       # sample_and_group_all: no centring (:157-160); [feats | xyz | pad]
            G, ns = B, N
        else:
            if geom is None:
                geom = self.geometry(xyz, with_csr=torch.is_grad_enabled())
            fps_idx, new_xyz, gidx = geom["fps_idx"], geom["new_xyz"], geom["group_idx"]
            G, ns = B * self.npoint, self.nsample
            self.last_aux = dict(fps_idx=fps_idx, group_idx=gidx)
            if feats is not None and ops.USE_PRE_LINEAR and feats.shape[-1] % 4 == 0 and self.mlp_convs[0].weight.shape[0] <= 256:
                # the first conv commutes with the grouping gather: it runs on the N points, the gather adds the coordinate part,
                # the bias and the BatchNorm sums (ops.mlp_stack(pre=...), csrc/gather.hip); the grouped input is never built
                csr = geom.get("csr")
                if csr is None and torch.is_grad_enabled():
                    csr = ops.build_csr(gidx, N)          # (no gradient, no inverse map: the forward never reads it)
                pre = dict(kind="group", xyz=xyz.contiguous(), new_xyz=new_xyz.contiguous(), idx=gidx, csr=csr, B=B, N=N, S=self.npoint,
                           ns=ns, rows=G * ns)
                F2 = feats.reshape(B * N, -1)
                out = ops.mlp_stack(F2, F2.shape[1], layers, "maxpool", self.training, G=G, ns=ns, pre=pre,
                                    staged={0: staged} if (staged is not None and "pre_wx" in staged) else None, aux_out=self.last_aux)
                return new_xyz, out.view(B, -1, out.shape[-1])
            X0 = geom["X0"] if (feats is None and "X0" in geom) else ops.group_gather(xyz, feats, new_xyz, gidx, geom.get("csr"))
        use = staged is not None and "pre_wx" not in staged and tuple(staged["W2"].shape) == (self.mlp_convs[0].weight.shape[0], (cin + 3) // 4 * 4)
        if self.group_all:
            self.last_aux = {}
        out = ops.mlp_stack(X0, cin, layers, "maxpool", self.training, G=G, ns=ns, xyz_last=True, staged={0: staged} if use else None,
                            aux_out=self.last_aux)
        return new_xyz, out.view(B, -1, out.shape[-1])

    def forward(self, xyz, points):
        """Reference layout: xyz (B,3,N), points (B,D,N) -> (B,3,S), (B,D',S)."""
        nx, nf = self.forward_pm(xyz.permute(0, 2, 1).contiguous(), None if points is None else points.permute(0, 2, 1).contiguous())
        return nx.permute(0, 2, 1), nf.permute(0, 2, 1)


class PointNetFeaturePropagation(nn.Module):
    """models/pointnet_util.py:270-320."""

    def __init__(self, in_channel, mlp):
        super().__init__()
        self.mlp_convs = nn.ModuleList()
        self.mlp_bns = nn.ModuleList()
        last = in_channel
        for co in mlp:
            self.mlp_convs.append(nn.Conv1d(last, co, 1))
            self.mlp_bns.append(nn.BatchNorm1d(co))
            last = co
        self.last_aux = {}

    def _input_pm(self, xyz1, xyz2, feats1, feats2, nn_=None):
        B, N, _ = xyz1.shape
        S = xyz2.shape[1]
        if S == 1:
            interp = feats2.expand(B, N, feats2.shape[-1]).reshape(B * N, -1)     # :298-299
        else:
            if nn_ is None:
                idx, w = ops.three_nn(xyz1, xyz2)
                nn_ = (idx, w, ops.build_csr(idx, S, w, 3) if torch.is_grad_enabled() else None)      # (the inverse map serves the backward only)
            idx, w, csr = nn_
            self.last_aux = dict(nn_idx=idx, nn_w=w)
            if feats1 is not None:
                # [skip | interpolated | pad] in one buffer: the interpolation writes its column block in place (no 25 MB cat at FP2)
                return ops.skip_interp_cat(feats1.reshape(B * N, -1), feats2, idx, w, csr)
            interp = ops.three_interpolate(feats2, idx, w, csr)
        if feats1 is not None:
            cols = [feats1.reshape(B * N, -1), interp]                            # [skip | interpolated] :312
            pad = (-(cols[0].shape[1] + cols[1].shape[1])) % 4                    # the kernels take 16-byte aligned rows (3 + 128 -> 132)
            if pad:
                cols.append(torch.zeros(B * N, pad, device=interp.device, dtype=interp.dtype))
            return torch.cat(cols, 1)
        return interp

    def forward_pm(self, xyz1, xyz2, feats1, feats2, tail="bnrelu", extra_layers=(), drop_mask=None, drop_scale=1.0, drop_seed=None,
                   keep_padding=False, nn_=None, staged=None):
        """xyz1 (B,N,3) dense, xyz2 (B,S,3) sparse, feats1 (B,N,D1)|None, feats2 (B,S,D2) -> (B,N,C')."""
        B, N, _ = xyz1.shape
        S = xyz2.shape[1]
        layers = _layers(self.mlp_convs, self.mlp_bns) + list(extra_layers)
        if feats1 is None and S > 1 and ops.USE_PRE_LINEAR and feats2.shape[-1] % 4 == 0:
            # no skip features: the first conv commutes with the interpolation, so it runs on the S sparse points and the
            # interpolation produces its dense pre-BN output (ops.mlp_stack(pre=...), csrc/gather.hip)
            if nn_ is None:
                idx, w = ops.three_nn(xyz1, xyz2)
                nn_ = (idx, w, ops.build_csr(idx, S, w, 3) if torch.is_grad_enabled() else None)
            idx, w, csr = nn_
            self.last_aux = dict(nn_idx=idx, nn_w=w)
            pre = dict(kind="interp", idx=idx, w=w, csr=csr, B=B, N=N, S=S, rows=B * N)
            F2 = feats2.reshape(B * S, -1)
            out = ops.mlp_stack(F2, F2.shape[1], layers, tail, self.training, drop_mask=drop_mask, drop_scale=drop_scale, drop_seed=drop_seed,
                                keep_padding=keep_padding, pre=pre, staged=staged if (staged and 0 not in staged) else None)
            return out.view(B, N, -1)
        if (S == 1 and feats1 is not None and ops.USE_PRE_LINEAR and N % 64 == 0 and feats1.shape[-1] % 4 == 0 and feats2.shape[-1] % 4 == 0
                and self.training and self.mlp_convs[0].weight.shape[0] % 4 == 0):
            # one sparse point: its features are REPEATED over the N rows (:298-299), so their product with the weight columns they
            # meet is computed once per cloud and enters the GEMM over the skip features as a per-cloud additive term
            pre = dict(kind="repeat", V=feats2.reshape(B, -1), rpg=N, rows=B * N)
            F1 = feats1.reshape(B * N, -1)
            out = ops.mlp_stack(F1, F1.shape[1], layers, tail, self.training, drop_mask=drop_mask, drop_scale=drop_scale, drop_seed=drop_seed,
                                keep_padding=keep_padding, pre=pre, staged=staged if (staged and 0 in staged and "pre_wb" in staged[0]) else None)
            return out.view(B, N, -1)
        X0 = self._input_pm(xyz1, xyz2, feats1, feats2, nn_)
        out = ops.mlp_stack(X0, self.mlp_convs[0].weight.shape[1], layers, tail, self.training, drop_mask=drop_mask, drop_scale=drop_scale, drop_seed=drop_seed,
                            keep_padding=keep_padding)
        return out.view(B, N, -1)

    def forward(self, xyz1, xyz2, points1, points2):
        """Reference layout: xyz1 (B,3,N), xyz2 (B,3,S), points1 (B,D,N)|None, points2 (B,D,S) -> (B,D',N)."""
        t = lambda a: None if a is None else a.permute(0, 2, 1).contiguous()
        return self.forward_pm(t(xyz1), t(xyz2), t(points1), t(points2)).permute(0, 2, 1)


class PointNetSetAbstractionMsg(nn.Module):
    """models/pointnet_util.py:210-267 is imported by the reference backbone but never instantiated; kept as a
    name so `from models.pointnet_util import PointNetSetAbstractionMsg` succeeds."""

    def __init__(self, *a, **k):
        super().__init__()
        raise NotImplementedError("PointNetSetAbstractionMsg is unused by Point2Cyl's backbone")


class backbone(nn.Module):
    """models/pointnet_extrusion.py:8-66.  forward(x (B,N,3[+D])) -> [ (B,N,o_i) ... ]."""

    def __init__(self, normal_channel=False, output_sizes=[3]):
        super().__init__()
        add = 3 if normal_channel else 0
        self.normal_channel = normal_channel
        self.dim_pos = 3
        self.sa1 = PointNetSetAbstraction(npoint=512, radius=0.2, nsample=64, in_channel=3 + add, mlp=[64, 64, 128], group_all=False)
        self.sa2 = PointNetSetAbstraction(npoint=128, radius=0.4, nsample=64, in_channel=128 + 3, mlp=[128, 128, 256], group_all=False)
        self.sa3 = PointNetSetAbstraction(npoint=None, radius=None, nsample=None, in_channel=256 + 3, mlp=[256, 512, 1024], group_all=True)
        self.fp3 = PointNetFeaturePropagation(in_channel=1024 + 256, mlp=[256, 256])
        self.fp2 = PointNetFeaturePropagation(in_channel=256 + 128, mlp=[256, 128])
        self.fp1 = PointNetFeaturePropagation(in_channel=128 + add, mlp=[128, 128, 128])
        self.fc1 = nn.Conv1d(128, 128, 1)
        self.bn1 = nn.BatchNorm1d(128)
        self.fc2 = nn.ModuleList()
        for o in output_sizes:
            self.fc2.append(nn.Conv1d(128, o, 1))
        self.dropout_p = 0.5
        self.dropout_mask = None   # test hook: (B,N,128) {0,1} mask to use instead of drawing one; "off" disables dropout
        self._drop_seed = None     # device int64 counter feeding the in-kernel dropout hash (advanced every forward)

    def _weight_stage(self, device):
        """The operands this forward derives from parameters alone, as one batched copy (ops.WeightStage).  Built lazily per device."""
        ws = self.__dict__.get("_wstage")
        if ws is not None and ws.device == device and ws.owner_id == id(self):
            return ws
        # sources are resolvers of the LIVE parameter (ops.WeightStage): never a tensor captured at build time
        entries = []
        sa1w, sa2w, sa3w, fp3w = (lambda: self.sa1.mlp_convs[0].weight), (lambda: self.sa2.mlp_convs[0].weight), \
            (lambda: self.sa3.mlp_convs[0].weight), (lambda: self.fp3.mlp_convs[0].weight)
        if not self.normal_channel:
            w = sa1w()                                                                    # (64, 3, 1, 1) -> (64, 4)
            entries.append(("sa1_W", (w.shape[0], 4), [(sa1w, 0, 3, 0, 0)]))
        w = sa2w()                                                                        # (128, 3 + 128): xyz part | feature part
        co, ci = w.shape[0], w.shape[1]
        entries.append(("sa2_wx", (co, 4), [(sa2w, 0, 3, 0, 0)]))
        entries.append(("sa2_W", (co, ci - 3), [(sa2w, 3, ci - 3, 0, 0)]))
        w = sa3w()                                                                        # (256, 3 + 256) -> [features | xyz | pad] (256, 260)
        co, ci = w.shape[0], w.shape[1]
        entries.append(("sa3_W", (co, (ci + 3) // 4 * 4), [(sa3w, 3, ci - 3, 0, 0), (sa3w, 0, 3, 0, ci - 3)]))
        w = fp3w()                                                                        # (256, 256 skip + 1024 repeated): two column blocks
        co, k = w.shape[0], self.sa2.mlp_convs[-1].weight.shape[0]
        entries.append(("fp3_W", (co, k), [(fp3w, 0, k, 0, 0)]))
        entries.append(("fp3_wb", (co, w.shape[1] - k), [(fp3w, k, w.shape[1] - k, 0, 0)]))
        tot = sum(m.weight.shape[0] for m in self.fc2)
        pad = (tot + 3) // 4 * 4
        wparts, bparts, o = [], [], 0
        for i, m in enumerate(self.fc2):
            wparts.append(((lambda i=i: self.fc2[i].weight), 0, m.weight.shape[1], o, 0))
            bparts.append(((lambda i=i: self.fc2[i].bias), 0, m.bias.shape[0], 0, o))
            o += m.weight.shape[0]
        entries.append(("head_W", (pad, self.fc2[0].weight.shape[1]), wparts))
        entries.append(("head_b", (pad,), bparts))
        # device counters the same launch advances: every BatchNorm's num_batches_tracked (+1 per train-mode forward) and the dropout seed
        bn_mods = [m for m in self.modules() if isinstance(m, nn.modules.batchnorm._BatchNorm)]
        nbt = [((lambda m=m: m.num_batches_tracked), 1) for m in bn_mods]
        seed = [((lambda: self._drop_seed), DROP_STRIDE)]
        ws = self.__dict__["_wstage"] = ops.WeightStage(entries, device, counters=nbt + seed)
        ws.nbt_counters, ws.seed_counter, ws.nbt_mods = nbt, seed, bn_mods
        ws.owner_id = id(self)
        return ws

    def __getstate__(self):
        """copy.deepcopy / pickling of the module never carry the weight stage (its resolvers point at THIS object): the copy builds its own."""
        d = self.__dict__.copy()
        d.pop("_wstage", None)
        d.pop("_bnstage", None)
        return d

    def compute_geometry(self, x, with_csr=True):
        """Everything in the forward pass that depends on the point coordinates only (no parameters): both FPS +
        ball-query levels, SA1's grouped relative coordinates and the two 3-NN interpolation stencils.  The result can be
        passed to forward_heads(x, geom=...); computing it for batch k+1 on a side stream hides the latency-bound FPS
        loop behind step k (point2cyl_amd/graph.py).  with_csr=False (inference, graph.PipelinedForward): without the three inverse
        maps of the gathers, which only the backward pass reads."""
        x = x.float()
        xyz = x[:, :, :3].contiguous()
        g1 = self.sa1.geometry(xyz, with_csr)
        if x.shape[2] == 3:
            g1["X0"] = ops.group_gather(xyz, None, g1["new_xyz"], g1["group_idx"])
        g2 = self.sa2.geometry(g1["new_xyz"], with_csr)
        def nn_with_csr(dense, sparse):
            idx, w = ops.three_nn(dense, sparse)
            return idx, w, (ops.build_csr(idx, sparse.shape[1], w, 3) if with_csr else None)

        return dict(sa1=g1, sa2=g2, fp2=nn_with_csr(g1["new_xyz"], g2["new_xyz"]), fp1=nn_with_csr(xyz, g1["new_xyz"]))

    def forward(self, x):
        from . import autograph
        r = autograph.forward_heads(self, x) if autograph.applicable(self, x) else None      # per-shape HIP graphs of this forward and its backward
        heads, sizes = r if r is not None else self.forward_heads(x)
        B, N = x.shape[0], x.shape[1]
        heads = heads.view(B, N, heads.shape[-1])
        outs, o = [], 0
        for s in sizes:
            outs.append(heads[:, :, o:o + s])
            o += s
        return outs

    def forward_heads(self, x, geom=None):
        """-> (heads (B*N, ld) with the outputs of all fc2 heads side by side, [o_0, o_1, ...])."""
        try:
            return self._forward_heads(x, geom)
        finally:
            ops._EVAL_AFF[0] = None           # (the batched eval-mode affines are valid for the forward that launched them only)

    def _bn_eval_stage(self, device):
        st = self.__dict__.get("_bnstage")
        if st is None or st.device != device or st.owner_id != id(self):
            st = self.__dict__["_bnstage"] = ops.BNEvalStage([m for m in self.modules() if isinstance(m, nn.modules.batchnorm._BatchNorm)], device)
            st.owner_id = id(self)
        return st

    def _forward_heads(self, x, geom):
        if not x.is_cuda:
            raise RuntimeError("point2cyl_amd.backbone runs on the HIP device only (got %s); there is no CPU path" % x.device)
        B, N, C = x.shape
        x = x.float()
        ops._DEFER_NBT[0] = True          # one multi-tensor "+= 1" for all 17 num_batches_tracked at the end
        xyz = x[:, :, :3].contiguous()
        feats0 = x[:, :, 3:].contiguous() if C > 3 else None
        gm = geom or {}
        ws = None
        hashed_dropout = self.dropout_mask is None          # (a mask tensor or "off" are test hooks)
        if hashed_dropout:
            # F.dropout(p=0.5) is ALWAYS on in the reference, also in eval (pointnet_extrusion.py:60).  The keep-mask is never stored: the
            # kernels regenerate it from (seed, element index); the seed is a device counter drawn once from torch's generator and
            # advanced per forward (HIP-graph safe).
            if abs(self.dropout_p * 256.0 - round(self.dropout_p * 256.0)) > 1e-6:
                raise ValueError("dropout_p = %r: the in-kernel hashed mask realises drop probabilities in steps of 1/256 only (csrc/common.h); "
                                 "the reference's value is 0.5 (pointnet_extrusion.py:60)" % self.dropout_p)
        # (the seed is created where it always was - AFTER this forward's FPS draws, so the CPU generator is consumed in the same order -
        # and advanced by a torch add in that first forward; from then on the staging launch advances it)
        have_seed = hashed_dropout and self._drop_seed is not None and self._drop_seed.device == x.device
        seed_bumped = False
        if ops.USE_STAGED_WEIGHTS:
            ws = self._weight_stage(x.device)
            # ONE launch: every padded / re-ordered / column-sliced weight operand of this forward, the 17 num_batches_tracked += 1 (train
            # mode) and the dropout seed's advance
            # (only the BatchNorms whose OWN .training is set advance, as in torch: model.train(); model.sa1.eval() freezes sa1's counters too)
            live_nbt = [c for c, m in zip(ws.nbt_counters, ws.nbt_mods) if m.training]
            bumped = ws.run(bump=live_nbt + (ws.seed_counter if have_seed else []))
            ops._NBT_BUMPED[0] = bumped and bool(live_nbt)
            seed_bumped = bumped and have_seed
        if ops.USE_BN_EVAL_BATCH and ops.USE_INFER_PATHS and not self.training and not torch.is_grad_enabled():
            # inference: ONE launch for the affine of every eval-mode BatchNorm (17 launches between the GEMMs otherwise)
            ops._EVAL_AFF[0] = self._bn_eval_stage(x.device).run()
        st = (lambda **kw: {k: ws[v] for k, v in kw.items()}) if ws is not None else (lambda **kw: None)
        l1_xyz, l1 = self.sa1.forward_pm(xyz, feats0, gm.get("sa1"), staged=st(W2="sa1_W") if (ws is not None and feats0 is None) else None)
        l2_xyz, l2 = self.sa2.forward_pm(l1_xyz, l1, gm.get("sa2"), staged=st(W2="sa2_W", pre_wx="sa2_wx"))
        l3_xyz, l3 = self.sa3.forward_pm(l2_xyz, l2, staged=st(W2="sa3_W"))
        l4 = self.fp3.forward_pm(l2_xyz, l3_xyz, l2, l3, staged={0: st(W2="fp3_W", pre_wb="fp3_wb")} if ws is not None else None)
        l5 = self.fp2.forward_pm(l1_xyz, l2_xyz, l1, l4, nn_=gm.get("fp2"))
        # FP1 -> fc1/bn1/relu -> dropout -> fc2 heads as ONE stack: l6 and the head activations stay out of HBM
        seed = None
        if isinstance(self.dropout_mask, str) and self.dropout_mask == "off":
            mask, dscale = None, 1.0
        elif self.dropout_mask is not None:
            mask, dscale = self.dropout_mask.reshape(B * N, 128).to(device=x.device, dtype=torch.uint8).contiguous(), 1.0 / (1.0 - self.dropout_p)
        else:
            if not have_seed:
                self._drop_seed = torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).to(x.device)
            if not seed_bumped:
                self._drop_seed += DROP_STRIDE
            mask, seed, dscale = None, self._drop_seed, 1.0 / (1.0 - self.dropout_p)
        sizes = [m.weight.shape[0] for m in self.fc2]
        head_staged = None
        if ws is not None:
            # the heads' stacked (and zero-padded) weight / bias already sit in the staged buffers; autograd sees their rows as a function
            # of the fc2 parameters (ops._HeadParams: no launch either way)
            Wh, bh = ops._HeadParams.apply(ws["head_W"], ws["head_b"], len(self.fc2), *[m.weight for m in self.fc2], *[m.bias for m in self.fc2])
            head_staged = {len(self.fp1.mlp_convs) + 1: dict(W2=ws["head_W"], b=ws["head_b"])}
        else:
            Wh = torch.cat([m.weight.reshape(m.weight.shape[0], 128) for m in self.fc2], 0)
            bh = torch.cat([m.bias for m in self.fc2], 0)
        extra = [dict(W=self.fc1.weight, b=self.fc1.bias, gamma=self.bn1.weight, beta=self.bn1.bias,
                      bn=ops.BNState(self.bn1.running_mean, self.bn1.running_var, self.bn1.num_batches_tracked,
                                     0.1 if self.bn1.momentum is None else self.bn1.momentum, self.bn1.eps)),
                 dict(W=Wh, b=bh, gamma=None, beta=None, bn=None)]
        heads = self.fp1.forward_pm(xyz, l1_xyz, feats0, l5, tail="linear", extra_layers=extra, drop_mask=mask, drop_scale=dscale,
                                    drop_seed=seed, keep_padding=True, nn_=gm.get("fp1"), staged=head_staged)
        ops._DEFER_NBT[0] = False
        ops._NBT_BUMPED[0] = False
        ops.flush_nbt()
        return heads.reshape(B * N, heads.shape[-1]), sizes
