#!/usr/bin/env python3
"""Generate tests/golden/rddbnet_nb23.npz by running the REFERENCE generator at the depth the benchmark uses.

    PYTHONDONTWRITEBYTECODE=1 python -B tests/golden/make_golden_depth.py          (build container only)

``RDDBNet(3, 3, 4)`` with the reference's defaults (nf=64, nb=23, gc=32: src/model/rddb.py:85-105, the generator of BASELINE
configs[1]) is 16.6 M parameters, too many to store; the weights are the reference's own initialisation under
``torch.manual_seed(0)`` -- srcgan_amd.RDDBNet draws the identical tensors from the same seed
(tests/test_host_logic.py::test_same_seed_gives_reference_initialisation), and the fixture keeps one checksum per parameter to
prove it.  Stored: input, target, the reference's float32 AND float64 (``module.double()``) results for the output, the MSE
loss, the input gradient, and per parameter gradient an L2 norm and a projection on a fixed probe vector; six gradients in
full.  Tensors only, no reference source text.
"""
import os
import sys

import numpy as np
import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import REF, OUT, _stub_modules, npy          # noqa: E402

FULL = ("conv_first.weight", "RRDB_trunk.0.RDB1.conv1.weight", "RRDB_trunk.11.RDB2.conv3.weight", "RRDB_trunk.22.RDB3.conv5.weight",
        "trunk_conv.weight", "conv_last.bias")


def probe(n, k):
    return np.cos(np.arange(n, dtype=np.float64) * 0.37 + k)


def main():
    sys.dont_write_bytecode = True
    _stub_modules()
    sys.path.insert(0, REF)
    from model import RDDBNet                      # src/model/__init__.py:4
    torch.set_num_threads(8)
    torch.manual_seed(0)
    net = RDDBNet(3, 3, 4)
    g = torch.Generator().manual_seed(5)
    x = torch.rand(1, 3, 24, 20, generator=g)
    t = torch.rand(1, 3, 96, 80, generator=g)
    out = {"x": npy(x), "t": npy(t), "seed": np.array(0)}
    names = [k for k, _ in net.named_parameters()]
    out["names"] = np.array(names)
    out["param_sum"] = np.array([float(p.detach().double().sum()) for p in net.parameters()])
    for tag, dt in (("32", torch.float32), ("64", torch.float64)):
        m = net.double() if dt == torch.float64 else net
        for p in m.parameters():
            p.grad = None
        xi = x.detach().clone().to(dt).requires_grad_(True)
        y = m(xi)
        loss = nn.MSELoss()(y, t.to(dt))
        loss.backward()
        out["y" + tag], out["dx" + tag], out["loss" + tag] = npy(y), npy(xi.grad), npy(loss)
        out["gnorm" + tag] = np.array([float(p.grad.double().norm()) for p in m.parameters()])
        out["gproj" + tag] = np.array([float(np.dot(npy(p.grad).astype(np.float64).ravel(), probe(p.numel(), i)))
                                      for i, p in enumerate(m.parameters())])
        for k, p in m.named_parameters():
            if k in FULL:
                out[f"grad{tag}/{k}"] = npy(p.grad)
    np.savez_compressed(os.path.join(OUT, "rddbnet_nb23.npz"), **out)
    print("rddbnet_nb23.npz:", os.path.getsize(os.path.join(OUT, "rddbnet_nb23.npz")), "bytes; loss32", out["loss32"], "loss64", out["loss64"])
    d = lambda a, b: float(np.abs(a - b).max() / np.abs(b).max())
    print("reference f32 vs f64: y", d(out["y32"], out["y64"]), "dx", d(out["dx32"], out["dx64"]),
          "gnorm", float(np.abs(out["gnorm32"] / out["gnorm64"] - 1).max()))


if __name__ == "__main__":
    main()
