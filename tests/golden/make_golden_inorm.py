#!/usr/bin/env python3
"""tests/golden/nlayerd_in.npz: the REFERENCE's NLayerDiscriminator built with norm_layer = nn.InstanceNorm2d
(model/model.py:598-634; the normalised convolutions then have a bias, :607-610), one train-mode forward / backward and an
eval-mode forward.  Run in the build container only (the reference never travels):

    PYTHONDONTWRITEBYTECODE=1 python -B tests/golden/make_golden_inorm.py

Only tensors are stored (inputs, state_dict, outputs, gradients)."""
import os
import sys

import numpy as np
import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import OUT, REF, _stub_modules, grads_np, npy, sd_np          # noqa: E402


def main():
    sys.dont_write_bytecode = True
    _stub_modules()
    sys.path.insert(0, REF)
    import model.model as legacy               # src/model/model.py
    torch.manual_seed(0)
    ic, ndf, nl = 3, 16, 3
    m = legacy.NLayerDiscriminator(ic, ndf, nl, norm_layer=nn.InstanceNorm2d)
    m.train()
    before = sd_np(m, "sd/")
    x = torch.rand(2, 3, 64, 80, requires_grad=True)
    y = m(x)
    loss = nn.MSELoss()(y, torch.tensor(1.0).expand_as(y))     # GANLoss('lsgan'), train.py:86-87,118-120
    loss.backward()
    m.eval()
    with torch.no_grad():
        y_eval = m(x)
    np.savez(os.path.join(OUT, "nlayerd_in.npz"), cfg=np.array([ic, ndf, nl]), x=npy(x), y=npy(y), loss=npy(loss),
             dx=npy(x.grad), y_eval=npy(y_eval), **before, **grads_np(m))
    print("nlayerd_in.npz:", sorted(before))


if __name__ == "__main__":
    main()
