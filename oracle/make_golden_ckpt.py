"""A checkpoint MADE BY THE REFERENCE MODULE, for the loader / eval interop test (eval.py:206-210 loads
`torch.load(...)["model"]`).  No upstream model.pth can be obtained (.MISSING_LARGE_BLOBS), so the reference's own backbone is trained
here for three Adam steps on a synthetic batch - enough to move every weight, every BatchNorm running statistic and the counters away
from their initial values - and saved exactly as the trainer saves it (train_Point2Cyl_without_sketch.py:405-410: {"model": state_dict}).
Next to it: the reference module's eval-mode and train-mode forward on a fixed cloud with that checkpoint loaded (FPS starts recorded,
dropout replaced by a recorded mask), which is what `tests/test_gpu_flows.py::test_reference_made_checkpoint_loads_and_evaluates` compares.

    python oracle/make_golden_ckpt.py      # needs /root/reference; -> tests/golden/ref_ckpt_3steps.pth (5.7 MB), g16_ref_ckpt_forward.npz"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import _refload                                   # noqa: E402
from oracle.make_golden import DropoutOff, RandintTap, save   # noqa: E402
from point2cyl_amd import synth                               # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def main():
    mods = _refload.load()
    pe, L = mods["pointnet_extrusion"], mods["losses"]
    torch.manual_seed(2024)
    model = pe.backbone(output_sizes=[3, 16])
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    pcs, nrm, seg, bb = [t for t in synth.make_batch(4, 1024, 8, seed=50)[:4]]
    pcs, nrm = pcs.float(), nrm.float()
    for step in range(3):
        for name, m in model.named_modules():                  # train...:357-360 from step 0 on
            if "bn" in name:
                m.momentum = 0.5
        X, W_raw = model(pcs)
        X = torch.nn.functional.normalize(X, p=2, dim=2, eps=1e-12)
        W2 = torch.softmax(W_raw, dim=2)
        W = W2[:, :, ::2] + W2[:, :, 1::2]
        total, _, _ = L.compute_all_losses(pcs, W, seg, X, nrm, 1.0, 1.0)
        opt.zero_grad()
        total.backward()
        opt.step()
        print("step", step, float(total))
    sd = model.state_dict()
    torch.save({"model": sd}, os.path.join(OUT, "ref_ckpt_3steps.pth"))
    # forward of the REFERENCE module with this checkpoint, on a fresh cloud
    test = synth.make_batch(2, 1024, 8, seed=77)
    x = test[0].float()
    g = torch.Generator().manual_seed(5)
    dmask = (torch.rand(2, 128, 1024, generator=g) < 0.5).float()          # (B,C,N) layout of pointnet_extrusion.py:60's operand
    arrs = dict(pcs=x, normals=test[1].float(), seg=test[2], bb=test[3], axes=test[6].float(), centers=test[8].float(),
                dropout_mask_bcn=np.packbits(dmask.numpy().astype(np.uint8)),
                keys=np.array(list(sd.keys())), nbt=np.array([int(v) for k, v in sd.items() if k.endswith("num_batches_tracked")]))
    for mode in ("eval", "train"):
        m2 = pe.backbone(output_sizes=[3, 16])
        m2.load_state_dict(torch.load(os.path.join(OUT, "ref_ckpt_3steps.pth"))["model"])
        m2.eval() if mode == "eval" else m2.train()
        torch.manual_seed(31)
        with torch.no_grad(), RandintTap() as tap, DropoutOff(dmask):
            X, W_raw = m2(x)
        arrs.update({mode + ":X": X, mode + ":W_raw": W_raw, mode + ":start1": tap.draws[0], mode + ":start2": tap.draws[1]})
    save("g16_ref_ckpt_forward", **arrs)
    print("ckpt bytes", os.path.getsize(os.path.join(OUT, "ref_ckpt_3steps.pth")))


if __name__ == "__main__":
    main()
