"""Generate tests/golden/g11_implicit.npz by RUNNING the upstream reference's implicit decoder (this container only):

    python -m oracle.make_golden_implicit

ImplicitNet / gradient / add_latent (IGR/network.py:8-92, :200-206) and the loss block of the with-sketch trainer
(train_Point2Cyl.py:610-648, executed from the reference file, not copied): values of the three loss terms and the gradients of their
sum w.r.t. every decoder parameter and the latent codes (i.e. through the double backward).  Only DATA is written."""
import importlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import _refload  # noqa: E402
from oracle.make_golden import exec_reference_lines, save  # noqa: E402


def load_network():
    _refload.load()
    saved = list(sys.path)
    sys.path[:0] = [os.path.join(_refload.REF_ROOT, "IGR")]
    try:
        for n in ("general", "network"):
            sys.modules.pop(n, None)
        return importlib.import_module("network")
    finally:
        sys.path[:] = saved
        for n in ("general", "network"):
            sys.modules.pop(n, None)


def main():
    net = load_network()
    ls = _refload.load()["losses"]
    B, K, S, L = 2, 3, 40, 14                  # d_in = 16 -> padded 16; skip layer output 48 - 16 = 32 ... use widths that need padding:
    L = 13                                     # d_in = 2 + 13 = 15 (pad to 16); skip layer output 50 - 15 = 35 (pad to 36)
    torch.manual_seed(21)
    dec = net.ImplicitNet(d_in=2 + L, dims=[50] * 8, skip_in=[4], geometric_init=True, radius_init=1, beta=100)
    sd0 = {k: v.detach().clone() for k, v in dec.state_dict().items()}
    g = torch.Generator().manual_seed(4)
    sk = torch.randn(B * K, S, 2, generator=g) * 0.5
    nrm = torch.nn.functional.normalize(torch.randn(B * K, S, 2, generator=g), dim=-1)
    non = torch.cat([sk + 0.05 * torch.randn(B * K, S, 2, generator=g), torch.rand(B * K, S // 8, 2, generator=g) * 2 - 1], 1)
    lat = torch.nn.functional.normalize(torch.randn(B * K, L, generator=g)).requires_grad_(True)
    mask_gt = torch.tensor([[True, True, False], [True, False, False]])
    ns = dict(torch=torch, add_latent=net.add_latent, gradient=net.gradient, implicit_net=dec, sk_pnts=sk.clone(), nonmnfld_pnts=non.clone(),
              latent_codes=lat, sk_normals=nrm.clone(), batch_size=B, K=K, mask_gt=mask_gt, reduce_mean_masked_instance=ls.reduce_mean_masked_instance)
    exec_reference_lines(os.path.join(_refload.REF_ROOT, "train_Point2Cyl.py"), 610, 648, ns)
    im = ns["im_loss"]
    im.backward()
    arrs = dict(B=B, K=K, sk_pnts=sk, sk_normals=nrm, nonmnfld_pnts=non, latent=lat.detach(), mask_gt=mask_gt, im_loss=im, mnfld_loss=ns["mnfld_loss"],
                grad_loss=ns["grad_loss"], normals_loss=ns["normals_loss"], lat_grad=lat.grad, sk_pred=ns["sk_pred"].detach(),
                mnfld_grad=ns["mnfld_grad"].detach(), names=np.array(list(sd0.keys())))
    for k, v in sd0.items():
        arrs["sd:" + k] = v
    for n, p in dec.named_parameters():
        arrs["grad:" + n] = p.grad
    save("g11_implicit", **arrs)


if __name__ == "__main__":
    main()
