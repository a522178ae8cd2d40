#!/usr/bin/env python3
"""Golden vectors of the legacy nearest-up-sampling generators (reference src/model/model.py:347-440:
``RDDBNet`` (legacy) and ``RDDBNetB`` = G_A of train.py:172,177), produced by running the REFERENCE classes on CPU.

Run in the build container only:   PYTHONDONTWRITEBYTECODE=1 python -B tests/golden/make_golden_legacy.py

Stores inputs / seeded state_dicts / outputs / gradients as data (no reference source text).  Parameters the
reference's forward does not use have ``.grad is None`` there: they are listed under ``nograd``.
"""
import os
import sys

import numpy as np
import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import OUT, REF, _stub_modules, npy, sd_np          # noqa: E402


def main():
    sys.dont_write_bytecode = True
    _stub_modules()
    sys.path.insert(0, REF)
    import model.model as legacy

    torch.set_num_threads(4)
    cases = (("rddbnetb_x2", "RDDBNetB", (3, 3, 16, 1, 8, "x2"), (2, 3, 12, 10)),
             ("rddbnetb_x4", "RDDBNetB", (1, 1, 16, 2, 8, "x4"), (1, 1, 10, 12)),
             ("legacy_rddbnet_x1", "RDDBNet", (3, 3, 16, 1, 8, "x1"), (1, 3, 9, 14)),
             ("legacy_rddbnet_x2", "RDDBNet", (3, 3, 16, 1, 8, "x2"), (2, 3, 12, 10)),
             ("legacy_rddbnet_x4", "RDDBNet", (1, 3, 16, 1, 8, "x4"), (1, 1, 8, 6)))
    for tag, cls, (ic, oc, nf, nb, gc, mode), shape in cases:
        torch.manual_seed(0)
        m = getattr(legacy, cls)(ic, oc, nf, nb, gc, mode)
        up = {"x1": 1, "x2": 2, "x4": 4}[mode]
        x = torch.rand(*shape, requires_grad=True)
        t = torch.rand(shape[0], oc, shape[2] * up, shape[3] * up)
        y = m(x)
        loss = nn.L1Loss()(y, t)
        loss.backward()
        grads = {"grad/" + k: npy(p.grad) for k, p in m.named_parameters() if p.grad is not None}
        nograd = np.array([k for k, p in m.named_parameters() if p.grad is None])
        np.savez(os.path.join(OUT, f"{tag}.npz"), cfg=np.array([ic, oc, nf, nb, gc, up]), x=npy(x), t=npy(t), y=npy(y),
                 loss=npy(loss), dx=npy(x.grad), nograd=nograd, **sd_np(m), **grads)
        print(tag, tuple(y.shape), float(loss), "params without grad:", len(nograd))


if __name__ == "__main__":
    main()
