"""Training harnesses with the reference's step surface (zero_grad -> backward -> optimizer.step()).

  PairedSRGAN  -- paired G + D step of BASELINE config 1/2: one generator /
                  discriminator pair of reference src/train.py:262-340 (backward_D_basic, backward_G,
                  optimize_parameters) with the pixel loss L1*lambda.
  SRCycleGAN   -- full cycle of reference src/train.py:145-340 (G_A, G_B, D_A, D_B; GAN + cycle + identity).
  CasSRC       -- cascade of reference src/trainCas.py:18-153 (SR net + colouriser, two L1 losses).

Networks and losses are the native modules of this package; optimisers are ``torch.optim.Adam`` instances with a
fused native ``step()`` (``srcgan_amd.optim.Adam``: same constructor, state and ``state_dict``)
(surface requirement).
"""
from __future__ import annotations

import itertools
import math
import random
from typing import Iterable, List, Optional

import torch
import torch.nn as nn

from . import ops
from .losses import GANLoss, L1Loss, PSNRLoss
from .model import EDSR, ESPCN, SRCNN, SRDN, NLayerDiscriminator, RDDBNet, RDDBNetA, RDDBNetB, ResDeconv
from .optim import Adam

__all__ = ["PairedSRGAN", "StackedSR", "SRCycleGAN", "CycleParams", "ImagePool", "CasSRC", "CasSRCConst", "CasSRCLAB", "CasSRCConstLAB", "CasParams",
           "set_requires_grad"]


def set_requires_grad(nets, requires_grad=False):
    """train.py:215-226 / trainCas.py:63-74."""
    if not isinstance(nets, (list, tuple)):
        nets = [nets]
    for net in nets:
        if net is not None:
            for p in net.parameters():
                p.requires_grad = requires_grad


class _DataParallel:
    """``harness.grad_sync = GradSync()[.attach()]`` makes a harness data parallel (srcgan_amd.dist).  ``_once`` names the networks
    that run SEVERAL times per optimiser step: their backward calls only accumulate and ``_sync(params)`` -- called in front of
    every ``optimizer.step()`` -- exchanges the accumulated gradient ONCE; the others are averaged inside their single backward
    call when the GradSync is attached (phased, overlapped), or here when it is not."""
    _once = ()
    _grad_sync = None

    @property
    def grad_sync(self):
        return self._grad_sync

    @grad_sync.setter
    def grad_sync(self, sync):
        self._grad_sync = sync
        if sync is not None:
            nets = []
            for name in self._once:
                v = getattr(self, name)
                nets += list(v) if isinstance(v, (list, tuple)) else [v]
            sync.once(*nets)

    def _sync(self, params):
        if self._grad_sync is not None:
            self._grad_sync.sync(list(params))


class PairedSRGAN(_DataParallel):
    """G-step: fake=G(x); loss_G = lsgan(D(fake), real) + lambda_l1 * L1(fake, y)   (D frozen, train.py:330)
    D-step: loss_D = 0.5 * (lsgan(D(y), real) + lsgan(D(fake.detach()), fake))      (train.py:262-280)
    Adam(G: 1e-4, D: 1e-5, betas (0.5, 0.999)) as train.py:191-192."""

    def __init__(self, in_ch=3, out_ch=3, up=4, nf=64, nb=23, gc=32, ndf=64, n_layers=3, dtype=None, device="cuda",
                 lambda_l1=10.0, lr_g=1e-4, lr_d=1e-5, beta1=0.5):
        self.device = torch.device(device)
        self.netG = RDDBNet(in_ch, out_ch, up, nf=nf, nb=nb, gc=gc, dtype=dtype).to(self.device)
        self.netD = NLayerDiscriminator(out_ch, ndf, n_layers, dtype=dtype).to(self.device)
        self.criterionGAN = GANLoss("lsgan", device=self.device)
        self.criterionL1 = L1Loss()
        self.lambda_l1 = lambda_l1
        self.optimizer_G = Adam(self.netG.parameters(), lr=lr_g, betas=(beta1, 0.999))      # a torch.optim.Adam with a fused step()
        self.optimizer_D = Adam(self.netD.parameters(), lr=lr_d, betas=(beta1, 0.999))
        # Data parallel (see _DataParallel): attached (what bench.py does) the generator's gradient is averaged INSIDE its one
        # backward call, in phases overlapped with the rest of that backward; the discriminator's real + fake pass accumulate
        # and are exchanged once.

    _once = ("netD",)

    def optimize_parameters(self, x, y):
        # ---- generator (train.py:330-333)
        set_requires_grad(self.netD, False)
        self.optimizer_G.zero_grad()
        self.fake = self.netG(x)
        self.loss_G_GAN = self.criterionGAN(self.netD(self.fake), True)
        self.loss_L1 = self.criterionL1(self.fake, y)
        self.loss_G = self.loss_G_GAN + self.loss_L1 * self.lambda_l1
        self.loss_G.backward()
        self._sync(self.netG.parameters())
        self.optimizer_G.step()
        # ---- discriminator (train.py:335-340)
        set_requires_grad(self.netD, True)
        self.optimizer_D.zero_grad()
        loss_real = self.criterionGAN(self.netD(y), True)
        loss_fake = self.criterionGAN(self.netD(self.fake.detach()), False)
        self.loss_D = (loss_real + loss_fake) * 0.5
        self.loss_D.backward()
        self._sync(self.netD.parameters())
        self.optimizer_D.step()


class StackedSR(_DataParallel):
    """BASELINE.json configs[4] ("Sat2Aerx8 stress"): RDDBNet stages stacked end to end (reference rddb.py:85-114, two instances:
    x4 then x2 by default -- SURVEY.md section 8d restates the configuration), one L1 loss on the final output, one Adam over
    all stages (trainCas.py:38-41 hyper-parameters).  A later stage's input gradient flows into the earlier stage through
    autograd (srcgan_rddbnet_backward's dx).

    ``micro_batch``: the batch is processed in slices of that many images whose gradients accumulate before the one optimiser
    step.  The generator has no cross-sample coupling (no normalisation layers), so this is the same step as the full batch --
    what it buys is memory: the dense-block activations of a 512x512 trunk are 7.2 GB per image in 16-bit storage, 231 GB for the
    configuration's 32 images per GPU; 16-image slices need half of that.
    ``loss_scale``: multiplies the loss before backward and divides the gradients before the step (fp16 storage: gradients below
    2^-24 would vanish; bf16 and fp32 need none).  With a scale the gradients are checked in the same multi-tensor pass that
    unscales them: a step with a non-finite gradient is skipped and the scale halved (``skipped_steps`` counts them), as
    torch.cuda.amp.GradScaler does -- half precision overflows at 65504."""

    def __init__(self, ups=(4, 2), in_ch=3, out_ch=3, nf=64, nb=23, gc=32, dtype=None, device="cuda", lr=1e-4,
                 micro_batch: Optional[int] = None, loss_scale: float = 1.0):
        self.device = torch.device(device)
        self.nets = [RDDBNet(in_ch if i == 0 else out_ch, out_ch, up, nf=nf, nb=nb, gc=gc, dtype=dtype).to(self.device)
                     for i, up in enumerate(ups)]
        self.criterion = L1Loss()
        self.optimizer = Adam(itertools.chain(*[n.parameters() for n in self.nets]), lr=lr)
        self.micro_batch = micro_batch
        self.loss_scale = float(loss_scale)
        self.dynamic_scale = self.loss_scale != 1.0     # fp16: check every step's gradients, back off on overflow, grow back slowly
        self.growth_interval = 200
        self._good_steps = 0
        self.skipped_steps = 0

    _once = ("nets",)         # micro-batches: one exchange of the accumulated gradient per optimiser step

    def parameters(self):
        return itertools.chain(*[n.parameters() for n in self.nets])

    def forward(self, x):
        for net in self.nets:
            x = net(x)
        return x

    def optimize_parameters(self, x, y):
        B = x.shape[0]
        mb = self.micro_batch or B
        self.optimizer.zero_grad()
        total = None
        for i in range(0, B, mb):
            xs, ys = x[i:i + mb], y[i:i + mb]
            out = self.forward(xs)
            loss = self.criterion(out, ys) * (xs.shape[0] / B)
            (loss * self.loss_scale if self.loss_scale != 1.0 else loss).backward()
            total = loss.detach() if total is None else total + loss.detach()
            del out, loss
        self._sync(self.parameters())
        self.loss = total
        if self.dynamic_scale:
            # the check does not depend on the CURRENT scale (it may have backed off to 1): an overflowing step must never reach
            # Adam.  Growth as torch.cuda.amp.GradScaler: x2 after `growth_interval` consecutive finite steps.
            grads = [p.grad for p in self.parameters() if p.grad is not None]
            found_inf = torch.zeros(1, device=grads[0].device)
            inv = torch.full((1,), 1.0 / self.loss_scale, device=grads[0].device)
            torch._amp_foreach_non_finite_check_and_unscale_(grads, found_inf, inv)
            if float(found_inf) != 0.0:          # (one host sync per step, scaled mode only)
                self.skipped_steps += 1
                self._good_steps = 0
                self.loss_scale = max(1.0, self.loss_scale * 0.5)
                return
            self._good_steps += 1
            if self._good_steps >= self.growth_interval:
                self._good_steps = 0
                self.loss_scale = min(self.loss_scale * 2.0, 65536.0)
        self.optimizer.step()


class ImagePool:
    """History of generated images (train.py:20-64): until the pool is full every image is stored and
    returned; afterwards with p=0.5 a stored image is returned and replaced by the new one."""

    def __init__(self, pool_size, rng: Optional[random.Random] = None):
        self.pool_size = pool_size
        self.images: List[torch.Tensor] = []
        self.rng = rng or random

    def query(self, images):
        if self.pool_size == 0:
            return images
        out = []
        for image in images:
            image = image.detach().unsqueeze(0)
            if len(self.images) < self.pool_size:
                self.images.append(image)
                out.append(image)
            elif self.rng.uniform(0, 1) > 0.5:
                k = self.rng.randint(0, self.pool_size - 1)
                out.append(self.images[k].clone())
                self.images[k] = image
            else:
                out.append(image)
        return torch.cat(out, 0)


class CycleParams:
    """train.py:344-361 defaults (net == '1': 3-channel images on both sides)."""

    def __init__(self, device="cuda"):
        self.device = torch.device(device)
        self.lr = 1e-4
        self.beta1 = 0.5
        self.batch_size = 1
        self.num_epochs = 25
        self.pool_size = 4
        self.lambda_identity = 1.0
        self.lambda_A = 10
        self.lambda_B = 10
        self.lr_policy = "cosine"
        self.mode = "x2"
        self.net = "1"
        self.nf, self.nb, self.gc, self.ndf, self.n_layers = 64, 3, 32, 64, 2
        self.dtype = None
        # G_A class: "RDDBNetB" is what reference train.py:172,177 constructs (legacy nearest-up-sampling generator,
        # model/model.py:394); "RDDBNet" (rddb.py, deconv up-sampler) is the generator BASELINE.json's configs name.
        self.G_A = "RDDBNet"


class SRCycleGAN(_DataParallel):
    """Full cycle step of reference src/train.py:145-340.  G_A = RDDBNet (LR->HR), G_B = RDDBNetA (HR->LR,
    build-defined), D_A on HR images, D_B on LR images."""

    def __init__(self, opt: CycleParams):
        self.opt = opt
        up = 2 if opt.mode == "x2" else 4
        self.up = up
        dev = opt.device
        if getattr(opt, "G_A", "RDDBNet") == "RDDBNetB":
            self.netG_A = RDDBNetB(3, 3, opt.nf, nb=opt.nb, gc=opt.gc, mode=opt.mode, dtype=opt.dtype).to(dev)
        else:
            self.netG_A = RDDBNet(3, 3, up, nf=opt.nf, nb=opt.nb, gc=opt.gc, dtype=opt.dtype).to(dev)
        self.netG_B = RDDBNetA(3, 3, up, nf=opt.nf, nb=opt.nb, gc=opt.gc, dtype=opt.dtype).to(dev)
        self.netD_A = NLayerDiscriminator(3, opt.ndf, opt.n_layers, dtype=opt.dtype).to(dev)
        self.netD_B = NLayerDiscriminator(3, opt.ndf, opt.n_layers, dtype=opt.dtype).to(dev)
        self.fake_A_pool = ImagePool(opt.pool_size)
        self.fake_B_pool = ImagePool(opt.pool_size)
        self.criterionGAN = GANLoss("lsgan", device=dev)
        self.criterionCycle = L1Loss()
        self.criterionIdt = L1Loss()
        self.optimizer_G = Adam(itertools.chain(self.netG_A.parameters(), self.netG_B.parameters()),
                                            lr=opt.lr, betas=(opt.beta1, 0.999))
        self.optimizer_D = Adam(itertools.chain(self.netD_A.parameters(), self.netD_B.parameters()),
                                            lr=1e-5, betas=(opt.beta1, 0.999))
        self.optimizers = [self.optimizer_G, self.optimizer_D]

    _once = ("netG_A", "netG_B", "netD_A", "netD_B")     # three passes per generator, two per discriminator and step

    set_requires_grad = staticmethod(set_requires_grad)

    def forward(self, realA, realB):                                     # train.py:228-249 (net == '1')
        self.real_A, self.real_B = realA, realB
        self.fake_B = self.netG_A(self.real_A)
        self.recl_A = self.netG_B(self.fake_B)
        self.fake_A = self.netG_B(self.real_B)
        self.recl_B = self.netG_A(self.fake_A)
        self.real_B_Gray = ops.nearest_resize(self.real_B, 1.0 / self.up)
        self.iden_A = self.netG_A(self.real_B_Gray)
        self.real_A_RGB = ops.nearest_resize(self.real_A, self.up)
        self.iden_B = self.netG_B(self.real_A_RGB)

    def backward_D_basic(self, netD, real, fake):                        # train.py:262-280
        loss_D = (self.criterionGAN(netD(real), True) + self.criterionGAN(netD(fake.detach()), False)) * 0.5
        loss_D.backward()
        return loss_D

    def backward_D_A(self):
        self.loss_D_A = self.backward_D_basic(self.netD_A, self.real_B, self.fake_B_pool.query(self.fake_B))

    def backward_D_B(self):
        self.loss_D_B = self.backward_D_basic(self.netD_B, self.real_A, self.fake_A_pool.query(self.fake_A))

    def backward_G(self):                                                # train.py:292-323
        o = self.opt
        if o.lambda_identity > 0:
            self.loss_iden_A = self.criterionIdt(self.iden_A, self.real_B) * (o.lambda_B / 2 * o.lambda_identity)
            self.loss_iden_B = self.criterionIdt(self.iden_B, self.real_A) * (o.lambda_A / 2 * o.lambda_identity)
        else:
            self.loss_iden_A = self.loss_iden_B = 0
        self.loss_G_A = self.criterionGAN(self.netD_A(self.fake_B), True)
        self.loss_G_B = self.criterionGAN(self.netD_B(self.fake_A), True)
        self.loss_cycle_A = self.criterionCycle(self.recl_A, self.real_A) * (o.lambda_A * 0.5)
        self.loss_cycle_B = self.criterionCycle(self.recl_B, self.real_B) * (o.lambda_B * 0.5)
        self.loss_G = (self.loss_G_A + self.loss_G_B) + self.loss_cycle_A + self.loss_cycle_B + self.loss_iden_A + self.loss_iden_B
        self.loss_G.backward()

    def optimize_parameters(self, realA, realB):                         # train.py:325-340
        self.forward(realA, realB)
        self.set_requires_grad([self.netD_A, self.netD_B], False)
        self.optimizer_G.zero_grad()
        self.backward_G()
        self._sync(itertools.chain(self.netG_A.parameters(), self.netG_B.parameters()))
        self.optimizer_G.step()
        self.set_requires_grad([self.netD_A, self.netD_B], True)
        self.optimizer_D.zero_grad()
        self.backward_D_A()
        self.backward_D_B()
        self._sync(itertools.chain(self.netD_A.parameters(), self.netD_B.parameters()))
        self.optimizer_D.step()


class CasParams:
    """trainCas.py:156-164 defaults + the three argparse flags (:168-177)."""

    def __init__(self, device="cuda", SRModel="RDDBNet", CModel="ResDeconv", up=2):
        self.device = torch.device(device)
        self.lr = 1e-4
        self.batch_size = 1
        self.num_epochs = 50
        self.matrix = 0
        self.lr_policy = "cosine"
        self.up, self.SRModel, self.CModel = up, SRModel, CModel
        self.dtype = None


# name -> constructor(in_ch, out_ch, up) ; the reference resolves these with eval() (trainCas.py:30-31)
MODEL_REGISTRY = {"RDDBNet": RDDBNet, "ESPCN": ESPCN, "SRCNN": SRCNN, "EDSR": EDSR, "SRDN": SRDN, "ResDeconv": ResDeconv}


class CasSRC:
    """Cascade SR + colourisation step (reference src/trainCas.py:18-153).  ``netG_A2C`` (SR on the gray
    image) is the native RDDBNet; ``netG_C2B`` is the native ResDeconv colouriser (the reference's default
    ``--CModel``, trainCas.py:170); both resolve through MODEL_REGISTRY like the reference's ``eval(opt.*Model)``."""

    def __init__(self, opt: CasParams):
        self.opt = opt
        sr = MODEL_REGISTRY[opt.SRModel]
        cm = MODEL_REGISTRY[opt.CModel]
        self.netG_A2C = sr(1, 1, opt.up).to(opt.device)
        co = getattr(self, "_c_out", 3)
        self.netG_C2B = (cm(1, co, 1) if cm is RDDBNet else cm(1, co)).to(opt.device)
        self.criterionSR, self.criterionC, self.criterionPSNR = L1Loss(), L1Loss(), PSNRLoss()
        self.optimizer_G = Adam(self.netG_A2C.parameters(), lr=opt.lr)
        self.optimizer_D = Adam(self.netG_C2B.parameters(), lr=opt.lr)
        self.optimizers = [self.optimizer_G, self.optimizer_D]
        self.init_log()

    set_requires_grad = staticmethod(set_requires_grad)

    def update_lr(self, opt):
        """trainCas.py:45-61 builds a fresh scheduler every epoch and steps it once; the observable effect of
        'cosine' is lr <- lr * (1 + cos(pi / num_epochs)) / 2 per call, 'step' (step_size 50) leaves lr unchanged,
        'plateau' on the constant opt.matrix never triggers within its patience."""
        if opt.lr_policy == "cosine":
            f = (1.0 + math.cos(math.pi / opt.num_epochs)) / 2.0
            for optimizer in self.optimizers:
                for g in optimizer.param_groups:
                    g["lr"] *= f
        elif opt.lr_policy in ("step", "plateau"):
            return None
        else:
            return NotImplementedError("learning rate policy [%s] is not implemented", opt.lr_policy)

    def init_log(self):
        self.loss_sr, self.loss_c, self.psnr_sr, self.psnr_c = [], [], [], []

    def forwardSR(self, realB):                                          # trainCas.py:82-97
        self.real_B = realB
        self.real_BC = ops.rgb_to_gray(realB)
        self.real_BA = ops.bilinear_down(self.real_BC, self.opt.up)
        self.fake_BC = self.netG_A2C(self.real_BA)

    def forwardC(self):                                                  # trainCas.py:99-101
        self.fake_BB = self.netG_C2B(self.real_BC)

    def transfer(self, realA):                                           # trainCas.py:103-112
        self.real_A = ops.bilinear_down(realA, self.opt.up)
        self.netG_A2C.eval()
        self.netG_C2B.eval()
        with torch.no_grad():       # the reference builds and drops a graph here; outputs are identical
            self.fake_AC = self.netG_A2C(self.real_A)
            self.fake_AB = self.netG_C2B(self.fake_AC)

    def backward_D(self):                                                # trainCas.py:114-117
        self.loss_C = self.criterionC(self.fake_BB, self.real_B)
        self.loss_C.backward()
        self.loss_c.append(self.loss_C.detach())

    def backward_G(self):                                                # trainCas.py:119-122
        self.loss_SR = self.criterionSR(self.fake_BC, self.real_BC)
        self.loss_SR.backward()
        self.loss_sr.append(self.loss_SR.detach())

    def validate(self):                                                  # trainCas.py:124-131
        self.psnr_SR = self.criterionPSNR(self.fake_BC.detach(), self.real_BC.detach())
        self.psnr_C = self.criterionPSNR(self.fake_BB.detach(), self.real_B.detach())
        self.psnr_sr.append(self.psnr_SR)
        self.psnr_c.append(self.psnr_C)

    def optimize_parameters(self, realA, realB):                         # trainCas.py:133-153
        self.netG_A2C.train()
        self.netG_C2B.train()
        self.forwardSR(realB)
        self.optimizer_G.zero_grad()
        self.backward_G()
        self.optimizer_G.step()
        self.forwardC()
        self.optimizer_D.zero_grad()
        self.backward_D()
        self.optimizer_D.step()
        self.transfer(realA)
        self.validate()

    def log_means(self):
        """means of the running lists (the reference calls .item() per step, trainCas.py:117-131; here the
        device scalars are only synchronised when the log line is produced)."""
        m = lambda xs: float(torch.stack(list(xs)).mean()) if xs else float("nan")
        return {"loss_SR": m(self.loss_sr), "psnr_SR": m(self.psnr_sr), "loss_C": m(self.loss_c), "psnr_C": m(self.psnr_c)}


class CasSRCConst(CasSRC):
    """reference src/trainCasConst.py: the SR network keeps the resolution (use a size-preserving --SRModel such as SRCNN or
    SRDN); its training input is the gray target blurred by bilinear /up then x up (:89-92); transfer() feeds realA as is (:103-105)."""

    def forwardSR(self, realB):
        self.real_B = realB
        self.real_BC = ops.rgb_to_gray(realB)
        self.real_BA = ops.bilinear_up(ops.bilinear_down(self.real_BC, self.opt.up), self.opt.up)
        self.fake_BC = self.netG_A2C(self.real_BA)

    def transfer(self, realA):
        self.real_A = realA
        self.netG_A2C.eval()
        self.netG_C2B.eval()
        with torch.no_grad():
            self.fake_AC = self.netG_A2C(self.real_A)
            self.fake_AB = self.netG_C2B(self.fake_AC)


class _LabMixin:
    """reference src/trainCasLAB.py / trainCasConstLAB.py: targets arrive as normalised LAB [B,3,H,W]; the SR branch works on L
    (channel 0), the colouriser maps L to the two ab channels (CModel(1, 2), :31; real_B = ab, real_BC = L, :83-84)."""
    _c_out = 2

    def _split(self, realB):
        self.real_B = realB[:, 1:, :, :].contiguous()
        self.real_BC = realB[:, :1, :, :].contiguous()


class CasSRCLAB(_LabMixin, CasSRC):
    def forwardSR(self, realB):
        self._split(realB)
        self.real_BA = ops.bilinear_down(self.real_BC, self.opt.up)
        self.fake_BC = self.netG_A2C(self.real_BA)


class CasSRCConstLAB(_LabMixin, CasSRCConst):
    """BASELINE.json configs[3] ("cascade-const LAB") surface: src/trainCasConstLAB.py."""

    def forwardSR(self, realB):
        self._split(realB)
        self.real_BA = ops.bilinear_up(ops.bilinear_down(self.real_BC, self.opt.up), self.opt.up)
        self.fake_BC = self.netG_A2C(self.real_BA)

