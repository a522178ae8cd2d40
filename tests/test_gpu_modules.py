"""GPU parity of the drop-in modules (one native call per forward / backward) against
(a) the golden vectors captured from the imported reference and (b) the CPU oracle on fresh
seeded inputs at full channel width.  f32 mode: <= 1e-3 relative (north-star gate).
bf16 mode: reported against a looser bound (bf16 storage of 8 significant bits through up to
~50 stacked convs); it is the perf mode, not the parity mode."""
import numpy as np
import pytest
import torch

import oracle
from conftest import load_golden, sub, rel_err, rel_l2

pytestmark = pytest.mark.gpu

F32_TOL = 1e-3


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


def _load(module, sd):
    missing = module.load_state_dict(sd, strict=True)
    return module.cuda()


@pytest.mark.parametrize("tag", ["rddbnet_x2", "rddbnet_x4", "rddbnet_x2_w32"])
def test_rddbnet_golden_f32(tag):
    from srcgan_amd import RDDBNet, L1Loss
    g = load_golden(tag)
    ic, oc, up, nf, nb, gc = [int(v) for v in g["cfg"]]
    net = _load(RDDBNet(ic, oc, up, nf=nf, nb=nb, gc=gc, dtype="fp32"), sub(g, "sd/"))
    x = torch.from_numpy(g["x"]).cuda().requires_grad_(True)
    y = net(x)
    assert rel_err(y.cpu(), g["y"]) < F32_TOL
    loss = L1Loss()(y, torch.from_numpy(g["t"]).cuda())
    assert abs(float(loss) - float(g["loss"])) < 1e-5
    loss.backward()
    assert rel_err(x.grad.cpu(), g["dx"]) < F32_TOL
    grads = sub(g, "grad/")
    for k, p in net.named_parameters():
        assert rel_err(p.grad.cpu(), grads[k]) < F32_TOL, k


def test_rdb_full_width_golden_f32():
    """One full-width dense block (nf=64, gc=32) from the reference, embedded in a generator whose other
    layers are the oracle's: checks the MFMA path at the real (Cin,Cout) pairs against reference numbers."""
    from srcgan_amd import ops
    g = load_golden("rdb_full")
    sd = sub(g, "sd/")
    x = torch.from_numpy(g["x"])
    dense = ops.to_nhwc(torch.cat([x, torch.zeros(1, 128, *x.shape[2:])], 1).cuda(), 192, "fp32")
    out = torch.zeros(1, x.shape[2], x.shape[3], 64, device="cuda")
    for k in range(5):
        cin = 64 + 32 * k
        wp = ops.pack_conv2d_fwd(sd[f"conv{k+1}.weight"].cuda(), "fp32")
        b = sd[f"conv{k+1}.bias"].cuda()
        if k < 4:
            ops.conv_igemm(dense, wp, dense, kh=3, kw=3, Cin=cin, Cout=32, y_coff=cin, pad=(1, 1), bias=b, act=True)
        else:
            ops.conv_igemm(dense, wp, out, kh=3, kw=3, Cin=192, Cout=64, pad=(1, 1), bias=b, alpha=0.2, r1=dense, r1_cend=64, beta1=1.0)
    assert rel_err(ops.to_nchw(out).cpu(), g["y"]) < F32_TOL


@pytest.mark.parametrize("dt,tol", [("fp32", F32_TOL), ("bf16", 5e-2)])
def test_rddbnet_full_width_vs_oracle(dt, tol):
    """nf=64, gc=32, nb=2, x4 on an odd-sized input; forward, input grad and every parameter grad."""
    from srcgan_amd import RDDBNet
    torch.manual_seed(0)
    sd = oracle.rddbnet_state(3, 3, 4, 64, 2, 32, seed=3)
    net = _load(RDDBNet(3, 3, 4, nf=64, nb=2, gc=32, dtype=dt), sd)
    x = torch.rand(2, 3, 19, 35)
    t = torch.rand(2, 3, 76, 140)
    ref_sd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    xr = x.clone().requires_grad_(True)
    yr = oracle.rddbnet_forward(ref_sd, xr, 4)
    oracle.l1_loss(yr, t).backward()
    xg = x.cuda().requires_grad_(True)
    y = net(xg)
    from srcgan_amd import L1Loss
    L1Loss()(y, t.cuda()).backward()
    err = rel_err if dt == "fp32" else rel_l2     # bf16: relative L2 (8-bit mantissa noise per element)
    assert err(y.cpu(), yr) < tol
    assert err(xg.grad.cpu(), xr.grad) < tol * 2
    worst = max(err(p.grad.cpu(), ref_sd[k].grad) for k, p in net.named_parameters())
    assert worst < tol * 2, worst


@pytest.mark.parametrize("tag", ["nlayerd_3", "nlayerd_2"])
def test_nlayerd_golden_f32(tag):
    from srcgan_amd import NLayerDiscriminator, GANLoss
    g = load_golden(tag)
    ic, ndf, nl = [int(v) for v in g["cfg"]]
    net = _load(NLayerDiscriminator(ic, ndf, nl, dtype="fp32"), sub(g, "sd/"))
    net.train()
    x = torch.from_numpy(g["x"]).cuda().requires_grad_(True)
    y = net(x)
    assert rel_err(y.cpu(), g["y"]) < F32_TOL
    loss = GANLoss("lsgan", device="cuda")(y, True)
    assert abs(float(loss) - float(g["loss"])) < 1e-5
    loss.backward()
    assert rel_err(x.grad.cpu(), g["dx"]) < F32_TOL
    grads = sub(g, "grad/")
    for k, p in net.named_parameters():
        assert rel_err(p.grad.cpu(), grads[k]) < F32_TOL, k
    after = sub(g, "sd_after/")
    sd_now = net.state_dict()
    for k, v in after.items():
        assert rel_err(sd_now[k].double().cpu(), v.double()) < 1e-4, k
    net.eval()
    with torch.no_grad():
        assert rel_err(net(x.detach()).cpu(), g["y_eval"]) < F32_TOL


@pytest.mark.parametrize("hw", [(96, 128), (97, 128)])
@pytest.mark.parametrize("dt,tol", [("fp32", F32_TOL), ("bf16", 8e-2)])
def test_nlayerd_full_width_vs_oracle(dt, tol, hw):
    """ndf=64, 3 layers (3->64->128->256->512->1) on a 3x96x128 batch (first layer in its space-to-depth form) and on an odd
    height (plain 4x4 s2 form), incl. a frozen pass (dgrad only).
    bf16 bound is loose on purpose: with a constant lsgan label the incoming gradient is nearly uniform per channel,
    so BatchNorm backward (g - mean g - xhat * mean(g xhat)) cancels most of a bf16-rounded g (f32 mode: 4e-6)."""
    from srcgan_amd import NLayerDiscriminator, GANLoss
    sd = oracle.nlayer_d_state(3, 64, 3, seed=5)
    net = _load(NLayerDiscriminator(3, 64, 3, dtype=dt), sd)
    torch.manual_seed(1)
    x = torch.rand(2, 3, *hw)
    ref_sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone()) for k, v in sd.items()}
    xr = x.clone().requires_grad_(True)
    yr = oracle.nlayer_d_forward(ref_sd, xr, True)
    oracle.gan_loss(yr, False).backward()
    xg = x.cuda().requires_grad_(True)
    y = net(xg)
    GANLoss("lsgan", device="cuda")(y, False).backward()
    err = rel_err if dt == "fp32" else rel_l2     # bf16: relative L2 (8-bit mantissa noise per element)
    assert err(y.cpu(), yr) < tol
    assert err(xg.grad.cpu(), xr.grad) < tol * 2
    worst = max(err(p.grad.cpu(), ref_sd[k].grad) for k, p in net.named_parameters())
    assert worst < tol * 2, worst
    # frozen discriminator (train.py:330): no parameter grads, input grad still flows
    for p in net.parameters():
        p.requires_grad_(False)
        p.grad = None
    xg2 = x.cuda().requires_grad_(True)
    GANLoss("lsgan", device="cuda")(net(xg2), True).backward()
    assert xg2.grad is not None and all(p.grad is None for p in net.parameters())


def test_paired_step_golden_f32():
    """Two paired G+D optimisation steps (BASELINE config-1 structure) from the reference's initial weights:
    losses and post-Adam weights."""
    from srcgan_amd.train import PairedSRGAN
    g = load_golden("paired_step")
    m = PairedSRGAN(3, 3, 2, nf=16, nb=1, gc=8, ndf=16, n_layers=3, dtype="fp32", device="cuda")
    m.netG.load_state_dict(sub(g, "g0/"))
    m.netD.load_state_dict(sub(g, "d0/"))
    x, y = torch.from_numpy(g["x"]).cuda(), torch.from_numpy(g["y"]).cuda()
    for step in range(2):
        m.optimize_parameters(x, y)
        for k, mine in (("loss_G", m.loss_G), ("loss_D", m.loss_D), ("loss_G_GAN", m.loss_G_GAN), ("loss_L1", m.loss_L1)):
            ref = float(g[f"{k}_{step}"])
            assert abs(float(mine) - ref) < 1e-3 * max(1.0, abs(ref)), (k, step)
        if step == 0:
            assert rel_err(m.fake.cpu(), g["fake_0"]) < F32_TOL
    for k, v in sub(g, "g1/").items():
        assert rel_err(m.netG.state_dict()[k].cpu(), v) < F32_TOL, k
    for k, v in sub(g, "d1/").items():
        if v.is_floating_point():
            assert rel_err(m.netD.state_dict()[k].cpu(), v) < F32_TOL, k


def test_rddbneta_vs_oracle_f32():
    """build-defined HR->LR generator (reference RDDBNetA has no source): parity vs the oracle restatement only."""
    from srcgan_amd import RDDBNetA, L1Loss
    sd = oracle.rddbneta_state(3, 3, 2, 32, 1, 16, seed=2)
    net = RDDBNetA(3, 3, 2, nf=32, nb=1, gc=16, dtype="fp32")
    net.load_state_dict(sd)
    net.cuda()
    torch.manual_seed(3)
    x = torch.rand(2, 3, 24, 40)
    t = torch.rand(2, 3, 12, 20)
    ref_sd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    xr = x.clone().requires_grad_(True)
    yr = oracle.rddbneta_forward(ref_sd, xr, 2)
    oracle.l1_loss(yr, t).backward()
    xg = x.cuda().requires_grad_(True)
    y = net(xg)
    L1Loss()(y, t.cuda()).backward()
    assert rel_err(y.cpu(), yr) < F32_TOL
    assert rel_err(xg.grad.cpu(), xr.grad) < F32_TOL
    for k, p in net.named_parameters():
        assert rel_err(p.grad.cpu(), ref_sd[k].grad) < F32_TOL, k


def test_cpu_tensor_fails_loudly():
    from srcgan_amd import RDDBNet
    net = RDDBNet(3, 3, 2, nf=16, nb=1, gc=8)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        net(torch.rand(1, 3, 8, 8))


@pytest.mark.parametrize("tag,kind", [("rddbnetb_x2", "B"), ("rddbnetb_x4", "B"), ("legacy_rddbnet_x1", "L"),
                                      ("legacy_rddbnet_x2", "L"), ("legacy_rddbnet_x4", "L")])
def test_legacy_generators_golden_f32(tag, kind):
    """Legacy nearest-up-sampling generators (reference model/model.py:347-440; RDDBNetB = G_A of train.py:172) against
    reference outputs, input gradients and parameter gradients (shared HRconv / upconv weights accumulate)."""
    from srcgan_amd import RDDBNetB, LegacyRDDBNet, L1Loss
    g = load_golden(tag)
    ic, oc, nf, nb, gc, up = [int(v) for v in g["cfg"]]
    cls = RDDBNetB if kind == "B" else LegacyRDDBNet
    net = _load(cls(ic, oc, nf, nb, gc, f"x{up}", dtype="fp32"), sub(g, "sd/"))
    x = torch.from_numpy(g["x"]).cuda().requires_grad_(True)
    y = net(x)
    assert rel_err(y.cpu(), g["y"]) < F32_TOL
    loss = L1Loss()(y, torch.from_numpy(g["t"]).cuda())
    assert abs(float(loss) - float(g["loss"])) < 1e-5
    loss.backward()
    assert rel_err(x.grad.cpu(), g["dx"]) < F32_TOL
    grads, nograd = sub(g, "grad/"), {str(k) for k in g["nograd"]}
    for k, p in net.named_parameters():
        if k in nograd:
            assert p.grad is None, k          # unused by the reference's forward: .grad stays None there too
        else:
            assert rel_err(p.grad.cpu(), grads[k]) < F32_TOL, k


@pytest.mark.parametrize("dt,tol", [("fp32", F32_TOL), ("bf16", 5e-2)])
def test_rddbnetb_full_width_vs_oracle(dt, tol):
    """RDDBNetB at nf=64, gc=32, x4 on an odd-sized input, f32 and bf16, against the (golden-pinned) oracle restatement."""
    from srcgan_amd import RDDBNetB, L1Loss
    torch.manual_seed(5)
    net = RDDBNetB(3, 3, 64, nb=1, gc=32, mode="x4", dtype=dt).cuda()
    sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    x = torch.rand(1, 3, 13, 21)
    t = torch.rand(1, 3, 52, 84)
    ref_sd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    xr = x.clone().requires_grad_(True)
    yr = oracle.rddbnetb_forward(ref_sd, xr, "x4")
    oracle.l1_loss(yr, t).backward()
    xg = x.cuda().requires_grad_(True)
    y = net(xg)
    L1Loss()(y, t.cuda()).backward()
    err = rel_err if dt == "fp32" else rel_l2
    assert err(y.cpu(), yr) < tol
    assert err(xg.grad.cpu(), xr.grad) < tol * 2
    worst = max(err(p.grad.cpu(), ref_sd[k].grad) for k, p in net.named_parameters())
    assert worst < tol * 2, worst


def _fp(t):
    t = t.detach().double().reshape(-1).cpu()
    return np.array([float(t.sum()), float((t * t).sum()), *[float(v) for v in t[:4]]])


@pytest.mark.parametrize("tag", ["resdeconv_gray", "resdeconv_rgb"])
def test_resdeconv_golden_f32(tag):
    """Native ResDeconv colouriser against the reference's output, loss, full gradients of selected parameters and gradient
    fingerprints (sum, sum of squares) of all 101 parameters; weights = the reference's seeded initialisation."""
    from srcgan_amd import ResDeconv, L1Loss
    g = load_golden(tag)
    src, tar, seed = [int(v) for v in g["cfg"]]
    torch.manual_seed(seed)
    net = ResDeconv(src, tar, dtype="fp32").cuda()
    y = net(torch.from_numpy(g["x"]).cuda())
    assert rel_err(y.cpu(), g["y"]) < F32_TOL
    loss = L1Loss()(y, torch.from_numpy(g["t"]).cuda())
    assert abs(float(loss) - float(g["loss"])) < 1e-5
    loss.backward()
    for k, v in sub(g, "grad/").items():
        assert rel_err(dict(net.named_parameters())[k].grad.cpu(), v) < F32_TOL, k
    for k, p in net.named_parameters():
        ref, mine = g["gfp/" + k], _fp(p.grad)
        assert abs(mine[0] - ref[0]) <= 1e-3 * max(1.0, np.sqrt(ref[1])), k
        assert abs(mine[1] - ref[1]) <= 2e-3 * max(ref[1], 1e-12), k


def test_resdeconv_bf16_vs_oracle():
    """bf16 perf mode of the colouriser (fp32 mode is the parity gate, above).  Forward within 5 % relative L2; the gradients
    of the last decoder stage within 5 %; deeper gradients are compared by direction only: each of the 20 GroupNorm
    backward passes projects the common-mode part of the incoming gradient out, so bf16's 2^-9 rounding of a gradient is
    amplified relative to what survives (measured on this random-initialised case: 0.3 % at pred.weight, 33 % at layer4,
    55 % at layer1 in relative L2 -- cosine similarity >= 0.8 everywhere)."""
    from srcgan_amd import ResDeconv, MSELoss
    torch.manual_seed(3)
    net = ResDeconv(1, 3, dtype="bf16").cuda()
    sd = {k: p.detach().cpu().clone().requires_grad_(True) for k, p in net.named_parameters()}
    x, t = torch.rand(2, 1, 64, 48), torch.rand(2, 3, 64, 48)
    yr = oracle.resdeconv_forward(sd, x)
    oracle.mse_loss(yr, t).backward()
    y = net(x.cuda())
    MSELoss()(y, t.cuda()).backward()
    assert rel_l2(y.cpu(), yr) < 5e-2
    cos = lambda a, b: float((a.double() * b.double()).sum() / (a.double().norm() * b.double().norm()).clamp_min(1e-300))
    for k, p in net.named_parameters():
        if k.startswith(("pred", "deconv13", "upRes3.1")):
            assert rel_l2(p.grad.cpu(), sd[k].grad) < 5e-2, k
        assert cos(p.grad.cpu(), sd[k].grad) > 0.75, (k, cos(p.grad.cpu(), sd[k].grad))


@pytest.mark.parametrize("tag", ["espcn_x2", "espcn_x3", "srcnn", "edsr_x2", "edsr_x4"])
def test_small_sr_models_golden_f32(tag):
    """Native ESPCN (the reference's default --SRModel) and SRCNN against reference outputs, losses and every parameter gradient."""
    from srcgan_amd import ESPCN, SRCNN, EDSR, L1Loss
    g = load_golden(tag)
    cfg = [int(v) for v in g["cfg"]]
    ic, oc, up = cfg[:3]
    net = EDSR(*cfg, dtype="fp32") if tag.startswith("edsr") else (SRCNN if tag == "srcnn" else ESPCN)(ic, oc, up, dtype="fp32")
    net = _load(net, sub(g, "sd/"))
    y = net(torch.from_numpy(g["x"]).cuda())
    assert rel_err(y.cpu(), g["y"]) < F32_TOL
    loss = L1Loss()(y, torch.from_numpy(g["t"]).cuda())
    assert abs(float(loss) - float(g["loss"])) < 1e-5
    loss.backward()
    grads = sub(g, "grad/")
    for k, p in net.named_parameters():
        if float(grads[k].abs().max()) < 1e-7:
            # mathematically zero (a conv bias in front of a one-channel-per-group GroupNorm, edsr_x4): rounding residue in both
            assert float(p.grad.abs().max()) < 1e-6, k
        else:
            assert rel_err(p.grad.cpu(), grads[k]) < F32_TOL, k


def test_espcn_bf16_vs_oracle():
    from srcgan_amd import ESPCN, MSELoss
    torch.manual_seed(6)
    net = ESPCN(1, 1, 2, dtype="bf16").cuda()
    sd = {k: p.detach().cpu().clone().requires_grad_(True) for k, p in net.named_parameters()}
    x, t = torch.rand(2, 1, 40, 56), torch.rand(2, 1, 80, 112)
    yr = oracle.espcn_forward(sd, x, 2)
    oracle.mse_loss(yr, t).backward()
    y = net(x.cuda())
    MSELoss()(y, t.cuda()).backward()
    assert rel_l2(y.cpu(), yr) < 3e-2
    assert max(rel_l2(p.grad.cpu(), sd[k].grad) for k, p in net.named_parameters()) < 6e-2


@pytest.mark.parametrize("tag", ["srdn_nb1", "srdn_nb2"])
def test_srdn_golden_f32(tag):
    """Native SRDN (encoder / decoder RRDB stacks with two skips, srdn.py:56-74) against reference outputs and gradients."""
    from srcgan_amd import SRDN, L1Loss
    g = load_golden(tag)
    cfg = [int(v) for v in g["cfg"]]
    net = _load(SRDN(*cfg, dtype="fp32"), sub(g, "sd/"))
    x = torch.from_numpy(g["x"]).cuda().requires_grad_(True)
    y = net(x)
    assert rel_err(y.cpu(), g["y"]) < F32_TOL
    loss = L1Loss()(y, torch.from_numpy(g["t"]).cuda())
    assert abs(float(loss) - float(g["loss"])) < 1e-5
    loss.backward()
    assert rel_err(x.grad.cpu(), g["dx"]) < F32_TOL
    grads = sub(g, "grad/")
    for k, p in net.named_parameters():
        if k.startswith("trunk_conv"):
            assert p.grad is None, k
        else:
            assert rel_err(p.grad.cpu(), grads[k]) < F32_TOL, k


def test_stacked_generators_x8_vs_oracle():
    """BASELINE configs[4] shape in miniature: a x4 generator feeding a x2 generator (3x16x12 -> 3x128x96); the second network's
    input gradient flows into the first through autograd (srcgan_rddbnet_backward's dx).  fp32 against the oracle."""
    from srcgan_amd import RDDBNet, L1Loss
    s1 = oracle.rddbnet_state(3, 3, 4, 16, 1, 8, seed=11)
    s2 = oracle.rddbnet_state(3, 3, 2, 16, 1, 8, seed=12)
    g1, g2 = _load(RDDBNet(3, 3, 4, nf=16, nb=1, gc=8, dtype="fp32"), s1), _load(RDDBNet(3, 3, 2, nf=16, nb=1, gc=8, dtype="fp32"), s2)
    torch.manual_seed(13)
    x, t = torch.rand(2, 3, 16, 12), torch.rand(2, 3, 128, 96)
    r1 = {k: v.clone().requires_grad_(True) for k, v in s1.items()}
    r2 = {k: v.clone().requires_grad_(True) for k, v in s2.items()}
    yr = oracle.rddbnet_forward(r2, oracle.rddbnet_forward(r1, x, 4), 2)
    oracle.l1_loss(yr, t).backward()
    y = g2(g1(x.cuda()))
    L1Loss()(y, t.cuda()).backward()
    assert rel_err(y.cpu(), yr) < F32_TOL
    for net, ref in ((g1, r1), (g2, r2)):
        for k, p in net.named_parameters():
            assert rel_err(p.grad.cpu(), ref[k].grad) < F32_TOL, k


def test_empty_batch_generator():
    """An empty batch gives an empty output of the right shape and zero parameter gradients (aten::convolution's behaviour on
    the reference side), not a kernel launch."""
    from srcgan_amd import RDDBNet
    net = RDDBNet(3, 3, 4, nf=16, nb=1, gc=8, dtype="fp32").cuda()
    y = net(torch.zeros(0, 3, 8, 12, device="cuda"))
    assert tuple(y.shape) == (0, 3, 32, 48)
    y.sum().backward()
    assert all(p.grad is not None and float(p.grad.abs().sum()) == 0.0 for p in net.parameters())
