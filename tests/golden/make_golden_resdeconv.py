#!/usr/bin/env python3
"""Golden vectors of the ResDeconv colouriser (reference src/model/resdeconv.py:99-195), produced by running the REFERENCE
class on CPU.  Run in the build container only:

    PYTHONDONTWRITEBYTECODE=1 python -B tests/golden/make_golden_resdeconv.py

The network has 15 M parameters (60 MB): instead of the state_dict the file stores the construction seed -- the build's
parameter holders are created in the reference's order, so the same seed reproduces the reference's initial weights
bit-for-bit, which the stored per-parameter fingerprints (sum, sum of squares, first 4 values) verify -- plus inputs,
outputs, the loss, fingerprints of every parameter gradient and a few small gradients in full.
"""
import os
import sys

import numpy as np
import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import OUT, REF, _stub_modules, npy          # noqa: E402

FULL = ("conv1.weight", "bn1.weight", "bn1.bias", "layer2.0.downsample.0.weight", "layer2.0.downsample.1.weight", "layer4.1.bn2.bias",
        "deconv13.weight", "pred.weight", "upRes3.1.bn2.weight")


def fingerprint(t):
    t = t.detach().double().reshape(-1)
    return np.array([float(t.sum()), float((t * t).sum()), *[float(v) for v in t[:4]]])


def main():
    sys.dont_write_bytecode = True
    _stub_modules()
    sys.path.insert(0, REF)
    from model import ResDeconv                       # src/model/__init__.py

    torch.set_num_threads(4)
    for tag, (src, tar), shape, seed in (("resdeconv_gray", (1, 3), (2, 1, 32, 48), 0), ("resdeconv_rgb", (3, 2), (1, 3, 16, 32), 1)):
        torch.manual_seed(seed)
        m = ResDeconv(src, tar)
        m.train()
        x = torch.rand(*shape)
        t = torch.rand(shape[0], tar, shape[2], shape[3])
        y = m(x)
        loss = nn.L1Loss()(y, t)
        loss.backward()
        names = [k for k, _ in m.named_parameters()]
        out = dict(cfg=np.array([src, tar, seed]), x=npy(x), t=npy(t), y=npy(y), loss=npy(loss), names=np.array(names))
        for k, p in m.named_parameters():
            out["wfp/" + k] = fingerprint(p)
            out["gfp/" + k] = fingerprint(p.grad)
            if k in FULL:
                out["grad/" + k] = npy(p.grad)
        np.savez(os.path.join(OUT, f"{tag}.npz"), **out)
        print(tag, tuple(y.shape), float(loss), len(names), "parameters")

    # ---- the constructor's other options (resdeconv.py:107): BN='IN' (InstanceNorm2d, no parameters) and layers=[3,4,6,3]
    for tag, (src, tar), shape, seed, layers, BN in (("resdeconv_in", (1, 2), (1, 1, 64, 64), 3, [2, 2, 2, 2], "IN"),
                                                     ("resdeconv_r34", (1, 3), (1, 1, 32, 48), 3, [3, 4, 6, 3], "GN")):
        from model.resdeconv import BasicBlock
        torch.manual_seed(seed)
        m = ResDeconv(src, tar, BasicBlock, layers, BN)
        m.train()
        x = torch.rand(*shape)
        t = torch.rand(shape[0], tar, shape[2], shape[3])
        y = m(x)
        loss = nn.L1Loss()(y, t)
        loss.backward()
        names = [k for k, _ in m.named_parameters()]
        full = [k for k in names if k in FULL or k in ("layer2.3.bn1.weight", "upRes1.5.bn2.bias", "layer1.2.conv1.weight")]
        out = dict(cfg=np.array([src, tar, seed, *layers, 1 if BN == "IN" else 0]), x=npy(x), t=npy(t), y=npy(y), loss=npy(loss), names=np.array(names))
        for k, p in m.named_parameters():
            out["wfp/" + k] = fingerprint(p)
            out["gfp/" + k] = fingerprint(p.grad)
            if k in full:
                out["grad/" + k] = npy(p.grad)
        np.savez(os.path.join(OUT, f"{tag}.npz"), **out)
        print(tag, tuple(y.shape), float(loss), len(names), "parameters")


if __name__ == "__main__":
    main()
