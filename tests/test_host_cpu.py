"""CPU: host-side logic (no GPU, no kernels): synthetic data, schedules, drop-in import names, data-parallel glue
(world_size 2 over gloo)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from point2cyl_amd import ddp, step, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_synthetic_batch_layout_and_determinism():
    b1 = synth.make_batch(3, 512, 8, seed=7)
    b2 = synth.make_batch(3, 512, 8, seed=7)
    pcs, nrm, seg, bb, pax, pdist, axes, dist, cen = b1
    assert all(torch.equal(x, y) for x, y in zip(b1, b2))
    assert pcs.shape == (3, 512, 3) and pcs.dtype == torch.float32 and seg.dtype == torch.int64 and bb.dtype == torch.int64
    assert axes.shape == (3, 8, 3) and cen.shape == (3, 8, 3) and dist.shape == (3, 8)
    np.testing.assert_allclose(pcs.norm(dim=-1).max(dim=1)[0].numpy(), 1.0, rtol=1e-5)       # unit max-norm (utils.py:938-950)
    np.testing.assert_allclose(nrm.norm(dim=-1).numpy(), 1.0, rtol=1e-4)
    for b in range(3):
        k = int(seg[b].max()) + 1
        assert set(seg[b].tolist()) == set(range(k)), "labels must be gap-free (losses.py:35)"
        assert (axes[b, k:] == 0).all() and (cen[b, k:] == 0).all()
        # barrel normals are perpendicular to the axis, base normals parallel
        d = (nrm[b] * pax[b]).sum(-1).abs()
        assert d[bb[b] == 0].max() < 1e-4 and (d[bb[b] == 1] - 1).abs().max() < 1e-4


def test_schedules_match_reference_formulas():
    assert step.get_batch_norm_decay(0, 32, 200000) == 0.5
    assert step.get_batch_norm_decay(6250, 32, 200000) == 0.25            # 200000 samples seen
    assert step.get_batch_norm_decay(10 ** 7, 32, 200000) == pytest.approx(0.01)
    assert step.get_learning_rate(1e-3, 6250, 32, 200000, 0.7) == pytest.approx(0.7e-3)
    from point2cyl_amd.backbone import backbone
    m = backbone(output_sizes=[3, 16])
    step.update_momentum(m, 0.123)
    hit = [n for n, mod in m.named_modules() if "bn" in n]
    assert len(hit) == 23 and all(getattr(mod, "momentum") == 0.123 for n, mod in m.named_modules() if "bn" in n)
    assert step.StepFlags().pred_sizes() == [3, 16]


def test_state_dict_is_the_reference_checkpoint_abi():
    from point2cyl_amd.backbone import backbone
    sd = backbone(output_sizes=[3, 16]).state_dict()
    assert len(sd) == 123
    assert tuple(sd["sa1.mlp_convs.0.weight"].shape) == (64, 3, 1, 1)
    assert tuple(sd["fp3.mlp_convs.0.weight"].shape) == (256, 1280, 1)
    assert tuple(sd["fc2.1.weight"].shape) == (16, 128, 1)
    assert sum(v.numel() for k, v in sd.items() if "running" not in k and "num_batches" not in k) == 1404243


def test_dropin_import_names():
    code = ("import importlib,sys;"
            "m=importlib.import_module('pointnet_extrusion');"
            "from models.pointnet_util import PointNetSetAbstractionMsg,PointNetSetAbstraction,PointNetFeaturePropagation;"
            "from losses import *;from data_utils import *;from global_variables import *;from network import *;from sampler import *;"
            "assert callable(ImplicitNet) and callable(PointNetEncoder) and callable(gradient) and callable(add_latent) and callable(NormalPerPoint);"
            "assert abs(get_learning_rate_schedules([dict(Type='Step',Initial=1e-3,Interval=10,Factor=0.5)])[0].get_learning_rate(25)-2.5e-4)<1e-12;"
            "assert callable(m.backbone) and callable(compute_all_losses) and callable(estimate_extrusion_axis) and callable(sketch_implicit_projection3) and g_zero_tol==1e-6;"
            "print('ok')")
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "point2cyl_amd", "dropin"),
                                                       os.path.join(ROOT, "point2cyl_amd", "dropin", "models"), ROOT]))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr


def test_shard_range_covers_everything():
    for n, w in ((32, 8), (10, 4), (3, 8), (1250, 8)):
        got = [ddp.shard_range(n, r, w) for r in range(w)]
        assert got[0][0] == 0 and got[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(got, got[1:]))


_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
from point2cyl_amd import ddp
rank, world, _ = ddp.init_from_env(backend="gloo")
torch.manual_seed(100 + rank)
m = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.BatchNorm1d(5), torch.nn.Linear(5, 2))
ddp.broadcast_module(m)
w0 = torch.cat([p.detach().reshape(-1) for p in m.parameters()])
ref = [torch.zeros_like(w0) for _ in range(world)]
dist.all_gather(ref, w0)
assert all(torch.equal(ref[0], r) for r in ref), "broadcast_module must equalise the replicas"
# the pre-flight of a multi-rank job (bench.py, both trainers): one eager all-reduce + identical-replica proof; raises when replicas differ
pre = ddp.preflight(m)
assert pre["eager_allreduce_ok"] and pre["params_identical"] and pre["world_size"] == world and pre["backend"] == "gloo"
with torch.no_grad():
    m[0].weight[0, 0] += float(rank)          # replicas drift apart
try:
    ddp.preflight(m)
    raise SystemExit("preflight accepted replicas that differ")
except RuntimeError as e:
    assert "identical" in str(e)
with torch.no_grad():
    m[0].weight[0, 0] -= float(rank)
sync = ddp.FlatGradSync(m.parameters(), world)
x = torch.randn(4, 6)                      # different shard per rank (seeded by rank)
sync.zero()
m(x).square().mean().backward()
local = torch.cat([p.grad.reshape(-1) for p in m.parameters()]).clone()
sync.allreduce()
avg = torch.cat([p.grad.reshape(-1) for p in m.parameters()])
gathered = [torch.zeros_like(local) for _ in range(world)]
dist.all_gather(gathered, local)
assert torch.allclose(avg, sum(gathered) / world, atol=1e-7)
assert all(p.grad.data_ptr() >= sync.flat.data_ptr() for p in m.parameters()), "grads must be views of the flat buffer"
# second step: the SAME persistent buffer, packed by one multi-tensor copy right after backward
flat_ptr = sync.flat.data_ptr()
sync.zero()
m(x * 2).square().mean().backward()
local2 = torch.cat([p.grad.reshape(-1) for p in m.parameters()]).clone()
sync.pack()
assert sync.flat.data_ptr() == flat_ptr and all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(sync.params, sync.views))
assert torch.equal(sync.flat, local2)
sync.allreduce()
g2 = [torch.zeros_like(local2) for _ in range(world)]
dist.all_gather(g2, local2)
assert torch.allclose(sync.flat, sum(g2) / world, atol=1e-7)
# checkpoint policy: BatchNorm running statistics averaged over the replicas
bn = m[1]
bn.running_mean.fill_(float(rank)); bn.running_var.fill_(1.0 + rank)
ddp.average_buffers(m)
assert torch.allclose(bn.running_mean, torch.full((5,), (world - 1) / 2.0)) and torch.allclose(bn.running_var, torch.full((5,), 1.0 + (world - 1) / 2.0))
lo, hi = ddp.shard_range(10, rank, world)
print("rank", rank, "ok", lo, hi)
dist.destroy_process_group()
"""


def test_grad_sync_two_ranks_gloo(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(_WORKER % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29517")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", "29517", str(script)], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count("ok") == 2


def test_checkpoint_abi_matches_the_reference_module(tmp_path):
    """SURVEY 8(b) checkpoint ABI: the drop-in backbone registers the reference's 123 state_dict keys in the reference's order, with
    the reference's shapes and - same torch seed, same construction order - the reference's initial values (G5 holds the key list and
    per-tensor checksums of the upstream module); `{"model": state_dict}` (train...:408) round-trips through torch.save / load; the
    trainer's update_momentum name matching (train...:153-156) reaches 23 modules."""
    from tests.conftest import load_golden
    from point2cyl_amd.backbone import backbone
    g = load_golden("g5_backbone_train")
    torch.manual_seed(int(g["seed"]))
    m = backbone(output_sizes=[3, 16])
    sd = m.state_dict()
    assert list(sd.keys()) == [str(k) for k in g["keys"]] and len(sd) == 123
    ck = np.stack([[v.double().sum().item(), v.double().abs().sum().item(), (v.double() ** 2).sum().item()] for v in sd.values()])
    np.testing.assert_allclose(ck, g["init_ck"], rtol=2e-5, atol=1e-6)
    assert tuple(sd["sa1.mlp_convs.0.weight"].shape) == (64, 3, 1, 1) and tuple(sd["fp3.mlp_convs.0.weight"].shape) == (256, 1280, 1)
    assert tuple(sd["fc1.weight"].shape) == (128, 128, 1) and tuple(sd["fc2.1.weight"].shape) == (16, 128, 1)
    path = os.path.join(tmp_path, "model.pth")
    torch.save({"model": sd}, path)
    m2 = backbone(output_sizes=[3, 16])
    m2.load_state_dict(torch.load(path)["model"])
    assert all(torch.equal(a, b) for a, b in zip(m2.state_dict().values(), sd.values()))
    step.update_momentum(m2, 0.25)
    hit = [n for n, mod in m2.named_modules() if "bn" in n]
    assert len(hit) == 23 and all(mod.momentum == 0.25 for n, mod in m2.named_modules() if "bn" in n)


def test_reorder_by_matching_equals_gather():
    """losses._reorder (a batched permutation product) == torch.gather(W, 2, matching_indices expanded) bit for bit, also when the
    matching repeats a column (unmatched slots point at column 0), and its gradient equals the gather's scatter-add."""
    from point2cyl_amd import losses
    g = torch.Generator().manual_seed(0)
    W = torch.rand(3, 40, 8, generator=g)
    m = torch.stack([torch.randperm(8, generator=g) for _ in range(3)])
    m[1, 4:] = 0
    idx = m.unsqueeze(1).expand(3, 40, 8)
    a = W.clone().requires_grad_(True)
    b = W.clone().requires_grad_(True)
    wa, wb = losses._reorder(a, m), torch.gather(b, 2, idx)
    assert torch.equal(wa, wb)
    w = torch.rand(3, 40, 8, generator=g)
    (wa * w).sum().backward()
    (wb * w).sum().backward()
    assert torch.allclose(a.grad, b.grad, rtol=0, atol=1e-6)


def test_bench_self_launch_refuses_without_gpus():
    """bench.py --gpus 2 with no launcher and fewer than 2 GPUs: non-zero exit, no JSON line (never a silent 1-GPU number)."""
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("box has >= 2 GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "P2C_ONE_GPU_RANKS")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode != 0
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert "GPU" in out.stderr


def test_add_noise_golden():
    """A19 add_noise (data_utils.py:84-96): same NumPy draws, same float64 result as the reference (fixture G14)."""
    from tests.conftest import load_golden
    from point2cyl_amd.fitting import add_noise
    g = load_golden("g14_add_noise")
    np.random.seed(int(g["np_seed"]))
    out = add_noise(torch.from_numpy(g["pcs"]), torch.from_numpy(g["normals"]), sigma=float(g["sigma"]))
    assert str(out.dtype) == str(g["out_dtype"]) == "torch.float64"
    assert np.array_equal(out.numpy(), g["out"])


def test_dataset_readers_match_the_reference_dataloader(tmp_path):
    """G15 (oracle/make_golden_r3.py): items of the reference's AutodeskDataset_h5 and AutodeskDataset_h5_sketches on a small in-memory
    file, for the flag combinations the scripts use and the full ones, under fixed seeds - h5data's readers must return the same tuples,
    field for field (values, order, shapes), from the same arrays through the dict source, and through an .npz export of them."""
    from tests.conftest import load_golden
    from point2cyl_amd.h5data import AutodeskH5, AutodeskH5Sketches
    g = load_golden("g15_dataloader")
    f = {k[5:]: g[k] for k in g if k.startswith("file:")}
    npz = str(tmp_path / "train.npz")
    np.savez(npz, **f)
    n_h5 = n_sk = 0
    for row in g["cases"]:
        kind, (op, center, scale, extent, seed, index, n_fields) = str(row[0]), [int(x) for x in row[1:]]
        for src in (f, npz):
            if kind == "h5":
                ds = AutodeskH5(src, 32, 8, op=bool(op), center=bool(center), extent=bool(extent))
                tag = "h5_%d" % n_h5
            else:
                ds = AutodeskH5Sketches(src, 32, 16, 8, op=bool(op), center=bool(center), with_scale=bool(scale), extent=bool(extent))
                tag = "sk_%d" % n_sk
            assert len(ds) == 3
            torch.manual_seed(seed)
            item = ds[index]
            assert len(item) == n_fields, (tag, len(item), n_fields)
            for j, v in enumerate(item):
                ref = g["%s:%d" % (tag, j)]
                assert np.asarray(v).shape == ref.shape and np.array_equal(np.asarray(v), ref), (tag, j)
        if kind == "h5":
            n_h5 += 1
        else:
            n_sk += 1
    assert n_h5 == 4 and n_sk == 5
    # whole-item mode (the device-resident trainers subsample on the GPU): no draw, every point
    ds = AutodeskH5Sketches(f, None, 16, 8, center=True)
    it = ds[0]
    assert it[0].shape == (96, 3) and np.array_equal(it[0], f["point_cloud"][0]) and it[9].shape == (8, 16, 4)


def test_h5min_reads_files_written_by_the_real_hdf5_library():
    """point2cyl_amd/h5min.py against HDF5 files made by the real library (oracle/make_golden_h5.py: h5py 3.3 / HDF5 1.10.6, the
    create_dataset calls of utils.py:1174-1188 / :1251-1268 - gzip-chunked float32 / int datasets) and a file of other layouts (contiguous,
    a chunk index with two B-tree levels, shuffle + gzip + fletcher32, ragged edge chunks, a group B-tree with several leaves, float64 /
    int32 / uint8): every dataset name, shape, dtype and value."""
    from point2cyl_amd.h5min import H5File
    G = os.path.join(ROOT, "tests", "golden")
    for stem, n in (("autodesk_schema_small", 10), ("autodesk_schema_sketches", 12), ("h5min_layouts", 47)):
        ref = np.load(os.path.join(G, stem + "_arrays.npz"))
        with H5File(os.path.join(G, stem + ".h5")) as f:
            assert sorted(f.keys()) == sorted(ref.files) and len(ref.files) == n
            for k in ref.files:
                a = f[k][:]
                assert a.shape == ref[k].shape and a.dtype == ref[k].dtype and np.array_equal(a, ref[k]), (stem, k)
            assert "point_cloud" in f or stem == "h5min_layouts"
    with pytest.raises(ValueError):
        H5File(os.path.join(G, "autodesk_schema_small_arrays.npz"))


def test_h5_file_source_through_the_dataset_classes(tmp_path):
    """The `.h5` branch of h5data.open_arrays on REAL files of the reference's schema (h5py when importable, h5min otherwise): the dataset
    classes return, item for item under the same seed, what they return on the same arrays through the dict source; dataset_path finds the
    file; the trainers' center=True 9-tuple and the reference's default 8-tuple (AutodeskDataset_h5(f, n, K): center=False)."""
    import shutil
    from point2cyl_amd.h5data import AutodeskH5, AutodeskH5Sketches, dataset_path
    G = os.path.join(ROOT, "tests", "golden")
    shutil.copy(os.path.join(G, "autodesk_schema_small.h5"), str(tmp_path / "train.h5"))
    path = dataset_path(str(tmp_path), "train")
    assert path.endswith("train.h5")
    arrays = dict(np.load(os.path.join(G, "autodesk_schema_small_arrays.npz")))
    for kw in (dict(), dict(center=True), dict(op=True, center=True, extent=True)):
        a, b = AutodeskH5(path, 64, 8, **kw), AutodeskH5(arrays, 64, 8, **kw)
        assert len(a) == len(b) == 6
        for idx in (0, 5):
            torch.manual_seed(7 + idx)
            ia = a[idx]
            torch.manual_seed(7 + idx)
            ib = b[idx]
            assert len(ia) == len(ib) == 8 + len([k for k in ("op", "center", "extent") if kw.get(k)])
            for u, v in zip(ia, ib):
                assert np.array_equal(np.asarray(u), np.asarray(v))
    sk_arrays = dict(np.load(os.path.join(G, "autodesk_schema_sketches_arrays.npz")))
    a = AutodeskH5Sketches(os.path.join(G, "autodesk_schema_sketches.h5"), 64, 32, 8, center=True, with_scale=True)
    b = AutodeskH5Sketches(sk_arrays, 64, 32, 8, center=True, with_scale=True)
    torch.manual_seed(3)
    ia = a[2]
    torch.manual_seed(3)
    ib = b[2]
    assert len(ia) == 11 and ia[9].shape == (8, 32, 4)
    for u, v in zip(ia, ib):
        assert np.array_equal(np.asarray(u), np.asarray(v))


def test_dropin_falls_through_to_the_shadowed_reference_module(tmp_path):
    """A drop-in `data_utils` / `losses` / `global_variables` shadows the whole reference module: names it does not override (eval.py
    --is_visu needs visualize_segmentation_pc / visualize_segmentation_pc_bb_v2, eval.py:659-664; star-imports also leak the module's own
    imports) must keep coming from the module it shadows, and the launcher must put the drop-ins IN FRONT of the script's directory
    (python puts the script's directory before PYTHONPATH).  A fake reference tree stands in for /root/reference."""
    ref = tmp_path / "fakeref"
    ref.mkdir()
    (ref / "global_variables.py").write_text("g_zero_tol = 123.0\nOTHER_CONSTANT = 7\n")
    (ref / "data_utils.py").write_text(
        "import json\nfrom global_variables import *\n"
        "def estimate_extrusion_axis(*a, **k):\n    return 'reference axis'\n"
        "def visualize_segmentation_pc(model_id, *a):\n    return 'reference visu ' + str(model_id) + ' ' + _helper()\n"
        "def visualize_segmentation_pc_bb_v2(*a):\n    return 'reference visu v2'\n"
        "def _helper():\n    return 'h%d' % OTHER_CONSTANT\n")
    (ref / "losses.py").write_text(
        "import numpy as np\nfrom global_variables import *\n"
        "def hungarian_matching(*a, **k):\n    return 'reference matching'\n"
        "def get_sketch_loss(a, b):\n    return 'reference sketch loss'\n")
    (ref / "broken").mkdir()
    (ref / "script.py").write_text(
        "import sys\nfrom global_variables import *\nfrom data_utils import *\nfrom losses import *\n"
        "import data_utils, losses\n"
        "assert 'point2cyl_amd' in estimate_extrusion_axis.__module__, estimate_extrusion_axis.__module__\n"
        "assert 'point2cyl_amd' in hungarian_matching.__module__\n"
        "assert g_zero_tol == 1e-6 and OTHER_CONSTANT == 7\n"
        "assert visualize_segmentation_pc('m', 1, 2) == 'reference visu m h7'\n"
        "assert visualize_segmentation_pc_bb_v2() == 'reference visu v2' and get_sketch_loss(0, 0) == 'reference sketch loss'\n"
        "assert json.dumps([1]) == '[1]' and np.zeros(1).shape == (1,)      # the shadowed modules' own imports leak through the star-import\n"
        "assert data_utils.__p2c_shadowed__.endswith('fakeref/data_utils.py') and sys.argv[1:] == ['--flag', '3']\n"
        "print('fallthrough ok')\n")
    out = subprocess.run([sys.executable, "-m", "point2cyl_amd.dropin.run", str(ref / "script.py"), "--flag", "3"], cwd=ROOT,
                         capture_output=True, text=True, timeout=180)
    assert out.returncode == 0 and "fallthrough ok" in out.stdout, out.stderr + out.stdout
    # a shadowed module whose out-of-scope dependency is missing: one warning, the hot-path names still resolve
    (ref / "data_utils.py").write_text("import a_module_that_is_not_installed_anywhere\n")
    (ref / "script2.py").write_text("import warnings\nwith warnings.catch_warnings(record=True) as w:\n    warnings.simplefilter('always')\n"
                                    "    from data_utils import *\nassert any('could not be imported' in str(x.message) for x in w)\n"
                                    "assert callable(estimate_extrusion_axis)\nprint('degraded ok')\n")
    out = subprocess.run([sys.executable, "-m", "point2cyl_amd.dropin.run", str(ref / "script2.py")], cwd=ROOT, capture_output=True, text=True, timeout=180)
    assert out.returncode == 0 and "degraded ok" in out.stdout, out.stderr + out.stdout


def test_barrel_counts_from_the_host_labels_give_the_same_draws():
    """The evaluation loop counts the barrel points per (cloud, segment) on the host copy of the labels (fitting.barrel_counts) so that the
    extent draws (data_utils.py:1674-1697) need no device sync: the counts equal the ones the draws derive themselves, and the draws - same
    generator, same order - are the same numbers."""
    from point2cyl_amd import fitting
    g = torch.Generator().manual_seed(3)
    B, N, K, S = 5, 700, 8, 32
    seg = torch.randint(-1, K, (B, N), generator=g)
    bb = torch.randint(0, 2, (B, N), generator=g)
    seg[1][seg[1] == 3] = 0                       # a segment absent from one cloud
    seg[2] = -1                                   # a cloud without any segment
    one = ((seg[3] == 5) & (bb[3] == 0)).nonzero().flatten()
    bb[3, one[1:]] = 1                            # exactly one barrel point: "not found" in the reference, no draw
    barrel = (seg.unsqueeze(-1) == torch.arange(K)) & (bb == 0).unsqueeze(-1)
    counts = fitting.barrel_counts(seg, bb, K)
    assert counts == barrel.sum(dim=1).t().tolist() and counts[5][3] == 1
    torch.manual_seed(11)
    a = fitting._barrel_draws(seg, bb, K, S)
    torch.manual_seed(11)
    b = fitting._barrel_draws(seg, bb, K, S, counts=counts)
    assert torch.equal(a, b) and int(a[3, 5].abs().sum()) == 0 and int(a[2].abs().sum()) == 0


def test_eval_accumulator_on_host_tensors():
    from point2cyl_amd import eval as ev
    acc = ev.Accumulator(extra=("x",))
    keys = acc.keys
    m1 = {k: torch.full((3,), float(i)) for i, k in enumerate(keys)}
    m2 = {k: torch.full((2,), float(10 * i)) for i, k in enumerate(keys)}
    acc.add(m1)
    acc.add(m2)
    means = acc.means()
    for i, k in enumerate(keys):
        assert abs(means[k] - (3 * i + 2 * 10 * i) / 5.0) < 1e-12


def test_cpu_quota_and_thread_fit():
    """hostmem: the CPU count a process may use is the cgroup's quota, not os.cpu_count(); the torch pool follows it."""
    from point2cyl_amd import hostmem
    q = hostmem.cpu_quota()
    assert 1 <= q <= (os.cpu_count() or 1)
    before = torch.get_num_threads()
    try:
        n = hostmem.fit_threads_to_quota(reserve=0)
        assert 1 <= n <= q and torch.get_num_threads() == n
    finally:
        torch.set_num_threads(before)


def test_remaining_env_switches_are_wired():
    """P2C_AUTOGRAPH, P2C_GEMM_BIG, P2C_STRICT_LABELS are read from the environment at import time; each flips the module
    attribute the GPU tests drive (autograph.ENABLED: test_autograph_module_forward_backward_equals_eager_launches; implicit.USE_BIG:
    test_implicit_decoder_big_tiles_vs_oracle_trainer_shapes; ops.STRICT_LABELS: test_strict_labels_raise_at_the_call)."""
    code = ("import json; from point2cyl_amd import autograph, implicit, ops; "
            "print(json.dumps([autograph.ENABLED, implicit.USE_BIG, ops.STRICT_LABELS, ops.USE_BN_EVAL_BATCH]))")
    for env_add, want in (({}, [True, True, False, True]),
                          ({"P2C_AUTOGRAPH": "0", "P2C_GEMM_BIG": "0", "P2C_STRICT_LABELS": "1", "P2C_BN_EVAL_BATCH": "0"}, [False, False, True, False])):
        env = dict(os.environ, **env_add)
        for k in ("P2C_AUTOGRAPH", "P2C_GEMM_BIG", "P2C_STRICT_LABELS", "P2C_BN_EVAL_BATCH"):
            if k not in env_add:
                env.pop(k, None)
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=180, cwd=ROOT)
        assert out.returncode == 0, out.stderr[-1500:]
        assert json.loads(out.stdout.strip().splitlines()[-1]) == want


def test_build_flags_keep_packed_fp32_out():
    """point2cyl_amd/build.py: every kernel is compiled WITHOUT v_pk_{add,mul,fma}_f32 (the consumer of a packed result two issue slots later
    read a stale low half under the MFMA kernels: wrong farthest-point picks in 3 - 54 % of the graph replays, tools/stress_prefetch.py,
    profiles/r06_fps_packed_hazard.log), the option is part of the source hash that stamps the profiles, and objects built with other
    options are rebuilt."""
    from point2cyl_amd import build
    assert build.FLAGS[-4:] == ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
    assert "-ffp-contract=off" in build.FLAGS
    stamp = os.path.join(build.OBJ, "flags.txt")
    if os.path.exists(build.LIB):
        assert os.path.exists(stamp) and open(stamp).read() == " ".join(build.FLAGS), "libp2c_hip.so was built with other options than build.FLAGS"
