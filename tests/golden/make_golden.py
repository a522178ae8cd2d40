#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE modules on CPU.

Run in the build container only (the reference never travels to the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python -B tests/golden/make_golden.py

It imports ``/root/reference/src`` (model package, model.model, losses,
trainCas) with empty stub modules for the optional third-party imports the hot
path never touches (torchvision, cv2, skimage, visdom), runs small seeded cases
and stores inputs / state_dicts / outputs / gradients as data.  No reference
source text is stored -- only tensors.
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

REF = "/root/reference/src"
OUT = os.path.dirname(os.path.abspath(__file__))


def _stub_modules():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m
    tv = mod("torchvision")
    tv.models = mod("torchvision.models", vgg16=None, vgg19=None)
    tv.transforms = mod("torchvision.transforms")
    mod("cv2")
    sk = mod("skimage")
    sk.io = mod("skimage.io", imsave=None)
    sk.color = mod("skimage.color", lab2rgb=None, rgb2lab=None, rgb2gray=None)
    mod("visdom", Visdom=object)


def npy(t):
    return t.detach().cpu().numpy().copy()


def sd_np(module, prefix="sd/"):
    return {prefix + k: npy(v) for k, v in module.state_dict().items()}


def grads_np(module, prefix="grad/"):
    return {prefix + k: npy(p.grad) for k, p in module.named_parameters()}


def main():
    sys.dont_write_bytecode = True
    _stub_modules()
    sys.path.insert(0, REF)
    from model import RDDBNet                      # src/model/__init__.py:4
    from model.rddb import ResidualDenseBlock_5    # src/model/rddb.py:48
    import model.model as legacy                   # NLayerDiscriminator, model.py:595
    import losses as ref_losses                    # src/losses.py

    torch.set_num_threads(4)

    # ---- G1/G8: one dense block, tiny and full width --------------------------------
    for tag, nf, gc, hw in (("rdb_tiny", 16, 8, (12, 10)), ("rdb_full", 64, 32, (12, 10))):
        torch.manual_seed(0)
        m = ResidualDenseBlock_5(nf, gc)
        x = torch.rand(1, nf, *hw, requires_grad=True)
        t = torch.rand(1, nf, *hw)
        y = m(x)
        loss = nn.L1Loss()(y, t)
        loss.backward()
        np.savez(os.path.join(OUT, f"{tag}.npz"), x=npy(x), t=npy(t), y=npy(y), loss=npy(loss),
                 dx=npy(x.grad), **sd_np(m), **grads_np(m))

    # ---- G2: whole generators (tiny widths) -------------------------------------------
    for tag, (ic, oc, up, nf, nb, gc), shape in (
            ("rddbnet_x2", (3, 3, 2, 16, 1, 8), (2, 3, 16, 12)),
            ("rddbnet_x4", (1, 1, 4, 16, 2, 8), (2, 1, 12, 16)),
            ("rddbnet_x2_w32", (3, 3, 2, 32, 1, 16), (1, 3, 20, 36))):
        torch.manual_seed(0)
        m = RDDBNet(ic, oc, up, nf=nf, nb=nb, gc=gc)
        x = torch.rand(*shape, requires_grad=True)
        t = torch.rand(shape[0], oc, shape[2] * up, shape[3] * up)
        y = m(x)
        loss = nn.L1Loss()(y, t)
        loss.backward()
        np.savez(os.path.join(OUT, f"{tag}.npz"), cfg=np.array([ic, oc, up, nf, nb, gc]), x=npy(x), t=npy(t),
                 y=npy(y), loss=npy(loss), dx=npy(x.grad), **sd_np(m), **grads_np(m))

    # ---- G3: the up-sampler alone (deconv k2 s2 + LeakyReLU) --------------------------
    torch.manual_seed(0)
    m = RDDBNet(3, 3, 4, nf=64, nb=1, gc=8).upscale_layers
    x = torch.rand(1, 64, 5, 7, requires_grad=True)
    y = m(x)
    (y * torch.linspace(-1, 1, y.numel()).view_as(y)).sum().backward()
    np.savez(os.path.join(OUT, "upscale_x4.npz"), x=npy(x), y=npy(y), dx=npy(x.grad), **sd_np(m), **grads_np(m))

    # ---- G4: PatchGAN discriminators, train mode (batch statistics) -------------------
    for tag, (ic, ndf, nl), shape in (("nlayerd_3", (3, 16, 3), (2, 3, 64, 64)),
                                      ("nlayerd_2", (1, 16, 2), (2, 1, 64, 48))):
        torch.manual_seed(0)
        m = legacy.NLayerDiscriminator(ic, ndf, nl)
        m.train()
        before = sd_np(m, "sd/")
        x = torch.rand(*shape, requires_grad=True)
        y = m(x)
        loss = nn.MSELoss()(y, torch.tensor(1.0).expand_as(y))     # GANLoss('lsgan'), train.py:86-87,118-120
        loss.backward()
        after = {k: v for k, v in sd_np(m, "sd_after/").items() if "running" in k or "num_batches" in k}
        m.eval()
        with torch.no_grad():
            y_eval = m(x)
        np.savez(os.path.join(OUT, f"{tag}.npz"), cfg=np.array([ic, ndf, nl]), x=npy(x), y=npy(y), loss=npy(loss),
                 dx=npy(x.grad), y_eval=npy(y_eval), **before, **after, **grads_np(m))

    # ---- G5: losses ---------------------------------------------------------------------
    torch.manual_seed(0)
    a = torch.rand(2, 3, 24, 24, requires_grad=True)
    b = torch.rand(2, 3, 24, 24)
    out = {"a": npy(a), "b": npy(b)}
    for name, crit in (("l1", ref_losses.L1Loss()), ("mse", ref_losses.MSELoss()), ("psnr", ref_losses.PSNRLoss())):
        a.grad = None
        v = crit(a, b)
        v.backward()
        out[name] = npy(v)
        out[name + "_da"] = npy(a.grad)
    for name, lab in (("gan_real", 1.0), ("gan_fake", 0.0)):
        a.grad = None
        v = nn.MSELoss()(a, torch.tensor(lab).expand_as(a))        # GANLoss lsgan path, train.py:98-120
        v.backward()
        out[name] = npy(v)
        out[name + "_da"] = npy(a.grad)
    np.savez(os.path.join(OUT, "losses.npz"), **out)

    # ---- G6: in-step preprocessing (trainCas.py:85-90,104-105; trainCasConst.py:89-92) --
    torch.manual_seed(0)
    img = torch.rand(1, 3, 16, 16)
    gray = 0.2125 * img[:, :1] + 0.7154 * img[:, 1:2] + 0.0721 * img[:, 2:3]
    pre = {"img": npy(img), "gray": npy(gray)}
    for up in (2, 4):
        pre[f"bil_down{up}"] = npy(F.interpolate(gray, scale_factor=1. / up, mode="bilinear"))
        pre[f"near_down{up}"] = npy(F.interpolate(img, scale_factor=1. / up))
        d = F.interpolate(gray, scale_factor=1. / up, mode="bilinear")
        pre[f"bil_downup{up}"] = npy(F.interpolate(d, scale_factor=up, mode="bilinear"))
    np.savez(os.path.join(OUT, "preproc.npz"), **pre)

    # ---- G7: one CasSRC.optimize_parameters (trainCas.py:133-153) ------------------------
    import trainCas
    from model import ResDeconv
    trainCas.RDDBNetTiny = lambda i, o, up: RDDBNet(i, o, up, nf=16, nb=1, gc=8)   # registry alias, tiny width

    class Opt:
        device = torch.device("cpu"); lr = 1e-4; batch_size = 1; num_works = 0
        num_epochs = 50; matrix = 0; lr_policy = "cosine"; up = 2
        SRModel = "RDDBNetTiny"; CModel = "ResDeconv"
    torch.manual_seed(0)
    cas = trainCas.CasSRC(Opt)
    cas.init_log()
    sr0 = {"sr0/" + k: npy(v) for k, v in cas.netG_A2C.state_dict().items()}
    realA = torch.rand(1, 1, 64, 64)
    realB = torch.rand(1, 3, 64, 64)
    cas.update_lr(Opt)
    lr_after = cas.optimizer_G.param_groups[0]["lr"]
    cas.optimize_parameters(realA, realB)
    sr1 = {"sr1/" + k: npy(v) for k, v in cas.netG_A2C.state_dict().items()}
    np.savez(os.path.join(OUT, "cas_step.npz"), realA=npy(realA), realB=npy(realB), lr_after=np.array(lr_after),
             loss_SR=np.array(cas.loss_sr[-1]), loss_C=np.array(cas.loss_c[-1]),
             psnr_SR=np.array(cas.psnr_sr[-1]), psnr_C=np.array(cas.psnr_c[-1]),
             real_BC=npy(cas.real_BC), real_BA=npy(cas.real_BA), fake_BC=npy(cas.fake_BC),
             real_A=npy(cas.real_A), fake_AC=npy(cas.fake_AC), **sr0, **sr1)

    # ---- G9: paired G+D step on the reference modules (config-1 shape, tiny width) --------
    torch.manual_seed(0)
    G = RDDBNet(3, 3, 2, nf=16, nb=1, gc=8)
    D = legacy.NLayerDiscriminator(3, 16, 3)
    g0 = {"g0/" + k: npy(v) for k, v in G.state_dict().items()}
    d0 = {"d0/" + k: npy(v) for k, v in D.state_dict().items()}
    opt_g = torch.optim.Adam(G.parameters(), lr=1e-4, betas=(0.5, 0.999))
    opt_d = torch.optim.Adam(D.parameters(), lr=1e-5, betas=(0.5, 0.999))
    x = torch.rand(2, 3, 32, 32)
    y = torch.rand(2, 3, 64, 64)
    mse = nn.MSELoss()
    l1 = ref_losses.L1Loss()
    logs = {}
    for step in range(2):
        for p in D.parameters():
            p.requires_grad = False
        opt_g.zero_grad()
        fake = G(x)
        pred = D(fake)
        lg_gan = mse(pred, torch.tensor(1.0).expand_as(pred))
        lg_l1 = l1(fake, y)
        lg = lg_gan + 10.0 * lg_l1
        lg.backward()
        opt_g.step()
        for p in D.parameters():
            p.requires_grad = True
        opt_d.zero_grad()
        pr = D(y)
        pf = D(fake.detach())
        ld = 0.5 * (mse(pr, torch.tensor(1.0).expand_as(pr)) + mse(pf, torch.tensor(0.0).expand_as(pf)))
        ld.backward()
        opt_d.step()
        logs[f"loss_G_{step}"] = npy(lg); logs[f"loss_D_{step}"] = npy(ld)
        logs[f"loss_G_GAN_{step}"] = npy(lg_gan); logs[f"loss_L1_{step}"] = npy(lg_l1)
        if step == 0:
            logs["fake_0"] = npy(fake)
    g1 = {"g1/" + k: npy(v) for k, v in G.state_dict().items()}
    d1 = {"d1/" + k: npy(v) for k, v in D.state_dict().items()}
    np.savez(os.path.join(OUT, "paired_step.npz"), x=npy(x), y=npy(y), **logs, **g0, **d0, **g1, **d1)

    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)) // 1024, "KiB")


if __name__ == "__main__":
    main()
