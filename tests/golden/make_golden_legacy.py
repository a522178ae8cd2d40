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


def small_sr():
    """ESPCN (the CLI default --SRModel, trainCas.py:169) and SRCNN, reference src/model/{espcn,srcnn}.py."""
    from model import ESPCN, SRCNN
    for tag, cls, args, shape, up in (("espcn_x2", ESPCN, (1, 1, 2), (2, 1, 20, 28), 2), ("espcn_x3", ESPCN, (3, 3, 3), (1, 3, 12, 10), 3),
                                      ("srcnn", SRCNN, (3, 3, 2), (2, 3, 18, 22), 1)):
        torch.manual_seed(0)
        m = cls(*args)
        x = torch.rand(*shape, requires_grad=False)
        t = torch.rand(shape[0], args[1], shape[2] * up, shape[3] * up)
        y = m(x)
        loss = nn.L1Loss()(y, t)
        loss.backward()
        np.savez(os.path.join(OUT, f"{tag}.npz"), cfg=np.array(args), x=npy(x), t=npy(t), y=npy(y), loss=npy(loss), **sd_np(m),
                 **{"grad/" + k: npy(p.grad) for k, p in m.named_parameters()})
        print(tag, tuple(y.shape), float(loss))


def edsr_golden():
    """EDSR (reference src/model/edsr.py:68-110) at reduced depth."""
    from model import EDSR
    for tag, args, shape in (("edsr_x2", (1, 1, 2, 64, 3), (2, 1, 16, 20)), ("edsr_x4", (3, 3, 4, 32, 2), (1, 3, 12, 10))):
        torch.manual_seed(0)
        m = EDSR(*args)
        x = torch.rand(*shape)
        t = torch.rand(shape[0], args[1], shape[2] * args[2], shape[3] * args[2])
        y = m(x)
        loss = nn.L1Loss()(y, t)
        loss.backward()
        np.savez(os.path.join(OUT, f"{tag}.npz"), cfg=np.array(args), x=npy(x), t=npy(t), y=npy(y), loss=npy(loss), **sd_np(m),
                 **{"grad/" + k: npy(p.grad) for k, p in m.named_parameters()})
        print(tag, tuple(y.shape), float(loss))


def srdn_golden():
    """SRDN (reference src/model/srdn.py:56-74) at tiny width."""
    from model import SRDN
    for tag, args, shape in (("srdn_nb1", (3, 3, 2, 16, 1, 8), (2, 3, 14, 18)), ("srdn_nb2", (1, 1, 2, 16, 2, 8), (1, 1, 10, 12))):
        torch.manual_seed(0)
        m = SRDN(*args)
        x = torch.rand(*shape, requires_grad=True)
        t = torch.rand(shape[0], args[1], shape[2], shape[3])
        y = m(x)
        loss = nn.L1Loss()(y, t)
        loss.backward()
        np.savez(os.path.join(OUT, f"{tag}.npz"), cfg=np.array(args), x=npy(x), t=npy(t), y=npy(y), loss=npy(loss), dx=npy(x.grad),
                 nograd=np.array([k for k, p in m.named_parameters() if p.grad is None]), **sd_np(m),
                 **{"grad/" + k: npy(p.grad) for k, p in m.named_parameters() if p.grad is not None})
        print(tag, tuple(y.shape), float(loss))


def cas_default():
    """Two CasSRC.optimize_parameters steps in the reference's DEFAULT configuration (trainCas.py:169-171: --SRModel ESPCN,
    --CModel ResDeconv, --up 2): losses and PSNRs only -- the build reproduces both networks' seeded initial weights."""
    import trainCas

    class Opt:
        device = torch.device("cpu"); lr = 1e-4; batch_size = 1; num_works = 0
        num_epochs = 50; matrix = 0; lr_policy = "cosine"; up = 2
        SRModel = "ESPCN"; CModel = "ResDeconv"
    torch.manual_seed(0)
    cas = trainCas.CasSRC(Opt)
    cas.init_log()
    realA = torch.rand(2, 1, 64, 96)
    realB = torch.rand(2, 3, 64, 96)
    for _ in range(2):
        cas.optimize_parameters(realA, realB)
    np.savez(os.path.join(OUT, "cas_default.npz"), realA=npy(realA), realB=npy(realB), loss_sr=np.array(cas.loss_sr), loss_c=np.array(cas.loss_c),
             psnr_sr=np.array(cas.psnr_sr), psnr_c=np.array(cas.psnr_c), fake_AB=npy(cas.fake_AB), fake_BC=npy(cas.fake_BC))
    print("cas_default", cas.loss_sr, cas.loss_c, cas.psnr_sr, cas.psnr_c)


def cas_variants():
    """One-to-two optimisation steps of the cascade variants (src/trainCasConst.py, trainCasLAB.py, trainCasConstLAB.py) with a
    size-preserving SR network where the script needs one; losses / PSNRs / outputs only (weights are reproduced by seed)."""
    import importlib
    for tag, modname, sr, shape_a in (("cas_const", "trainCasConst", "SRCNN", (1, 1, 48, 64)), ("cas_lab", "trainCasLAB", "ESPCN", (1, 1, 48, 64)),
                                      ("cas_constlab", "trainCasConstLAB", "SRDN", (2, 1, 32, 48))):
        mod = importlib.import_module(modname)

        class Opt:
            device = torch.device("cpu"); lr = 1e-4; batch_size = 1; num_works = 0
            num_epochs = 50; matrix = 0; lr_policy = "cosine"; up = 2
            SRModel = sr; CModel = "ResDeconv"
        torch.manual_seed(0)
        cas = mod.CasSRC(Opt)
        cas.init_log()
        realA = torch.rand(*shape_a)
        realB = torch.rand(shape_a[0], 3, shape_a[2], shape_a[3])
        for _ in range(2):
            cas.optimize_parameters(realA, realB)
        np.savez(os.path.join(OUT, f"{tag}.npz"), realA=npy(realA), realB=npy(realB), loss_sr=np.array(cas.loss_sr), loss_c=np.array(cas.loss_c),
                 psnr_sr=np.array(cas.psnr_sr), psnr_c=np.array(cas.psnr_c), real_BA=npy(cas.real_BA), fake_BC=npy(cas.fake_BC), fake_AC=npy(cas.fake_AC))
        print(tag, cas.loss_sr, cas.loss_c)


def metrics_golden():
    """Reference src/metrics.py (AE, MSE, PSNR, SSIM) on three value ranges that select SSIM's three dynamic ranges."""
    import metrics as ref
    out = {}
    for tag, scale, shift in (("unit", 1.0, 0.0), ("byte", 255.0, 0.0), ("signed", 2.0, -1.0)):
        torch.manual_seed(7)
        t = torch.rand(2, 3, 40, 52) * scale + shift
        p = (t + 0.1 * scale * torch.randn(2, 3, 40, 52)).clamp(shift, shift + scale)
        out[f"{tag}/pred"], out[f"{tag}/true"] = npy(p), npy(t)
        out[f"{tag}/ae"] = npy(ref.AE()(p, t)); out[f"{tag}/mse"] = npy(ref.MSE()(p, t)); out[f"{tag}/psnr"] = npy(ref.PSNR()(p, t))
        s, cs = ref.SSIM()(p, t, full=True)
        out[f"{tag}/ssim"], out[f"{tag}/cs"] = npy(s), npy(cs)
        out[f"{tag}/ssim_per_image"] = npy(ref.SSIM()(p, t, size_average=False))
    np.savez(os.path.join(OUT, "metrics.npz"), **out)
    print("metrics", {k: float(v) for k, v in out.items() if k.endswith(("/ssim", "/psnr"))})


if __name__ == "__main__":
    main()
    small_sr()
    cas_default()
    metrics_golden()
    edsr_golden()
    srdn_golden()
    cas_variants()
