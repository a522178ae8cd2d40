"""CPU: the oracle restatement vs the golden vectors produced by the imported
reference (tests/golden/make_golden.py).  This is what pins the oracle."""
import numpy as np
import pytest
import torch

import oracle
from conftest import load_golden, sub, rel_err

TOL = 2e-5   # fp32 accumulation-order noise only; the restatement calls the same ATen ops


def _req(sd):
    return {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone())
            for k, v in sd.items()}


@pytest.mark.parametrize("tag", ["rdb_tiny", "rdb_full"])
def test_rdb(tag):
    g = load_golden(tag)
    sd = _req(sub(g, "sd/"))
    x = torch.from_numpy(g["x"]).requires_grad_(True)
    y = oracle.rdb_forward(sd, "", x)
    loss = oracle.l1_loss(y, torch.from_numpy(g["t"]))
    loss.backward()
    assert rel_err(y, g["y"]) < TOL
    assert abs(float(loss) - float(g["loss"])) < 1e-6
    assert rel_err(x.grad, g["dx"]) < TOL
    for k, v in sub(g, "grad/").items():
        assert rel_err(sd[k].grad, v) < TOL, k


@pytest.mark.parametrize("tag", ["rddbnet_x2", "rddbnet_x4", "rddbnet_x2_w32"])
def test_rddbnet(tag):
    g = load_golden(tag)
    ic, oc, up, nf, nb, gc = [int(v) for v in g["cfg"]]
    sd = _req(sub(g, "sd/"))
    assert list(sd.keys()) == oracle.rddbnet_keys(nb, up)
    x = torch.from_numpy(g["x"]).requires_grad_(True)
    y = oracle.rddbnet_forward(sd, x, up)
    loss = oracle.l1_loss(y, torch.from_numpy(g["t"]))
    loss.backward()
    assert rel_err(y, g["y"]) < TOL
    assert rel_err(x.grad, g["dx"]) < TOL
    for k, v in sub(g, "grad/").items():
        assert rel_err(sd[k].grad, v) < TOL, k
    # state generator has the reference's key order and shapes
    mine = oracle.rddbnet_state(ic, oc, up, nf, nb, gc)
    assert [(k, tuple(v.shape)) for k, v in mine.items()] == [(k, tuple(v.shape)) for k, v in sd.items()]


def test_upscale():
    g = load_golden("upscale_x4")
    w = sub(g, "sd/")
    sd = {f"upscale_layers.{k}": v.clone().requires_grad_(True) for k, v in w.items()}
    x = torch.from_numpy(g["x"]).requires_grad_(True)
    h = x
    for s in range(2):
        h = torch.nn.functional.leaky_relu(
            torch.nn.functional.conv_transpose2d(h, sd[f"upscale_layers.{2*s}.weight"], None, 2, 0), 0.2)
    (h * torch.linspace(-1, 1, h.numel()).view_as(h)).sum().backward()
    assert rel_err(h, g["y"]) < TOL
    assert rel_err(x.grad, g["dx"]) < TOL


@pytest.mark.parametrize("tag", ["nlayerd_3", "nlayerd_2"])
def test_nlayer_d(tag):
    g = load_golden(tag)
    ic, ndf, nl = [int(v) for v in g["cfg"]]
    sd = _req(sub(g, "sd/"))
    assert list(sd.keys()) == oracle.nlayer_d_keys(nl)
    x = torch.from_numpy(g["x"]).requires_grad_(True)
    y = oracle.nlayer_d_forward(sd, x, True)
    loss = oracle.gan_loss(y, True)
    loss.backward()
    assert rel_err(y, g["y"]) < TOL
    assert abs(float(loss) - float(g["loss"])) < 1e-6
    assert rel_err(x.grad, g["dx"]) < 1e-4
    for k, v in sub(g, "grad/").items():
        assert rel_err(sd[k].grad, v) < 1e-4, k
    for k, v in sub(g, "sd_after/").items():
        assert rel_err(sd[k].double(), v.double()) < 1e-5, k
    with torch.no_grad():
        assert rel_err(oracle.nlayer_d_forward(sd, x, False), g["y_eval"]) < TOL
    mine = oracle.nlayer_d_state(ic, ndf, nl)
    assert [(k, tuple(v.shape)) for k, v in mine.items()] == [(k, tuple(v.shape)) for k, v in sd.items()]


def test_nlayer_d_instance_norm():
    """norm_layer = nn.InstanceNorm2d (model/model.py:598-634; fixture: tests/golden/make_golden_inorm.py runs the reference class)."""
    g = load_golden("nlayerd_in")
    ic, ndf, nl = [int(v) for v in g["cfg"]]
    sd = _req(sub(g, "sd/"))
    assert list(sd.keys()) == oracle.nlayer_d_keys(nl, norm="instance")
    x = torch.from_numpy(g["x"]).requires_grad_(True)
    y = oracle.nlayer_d_forward(sd, x, True)
    loss = oracle.gan_loss(y, True)
    loss.backward()
    assert rel_err(y, g["y"]) < TOL
    assert abs(float(loss) - float(g["loss"])) < 1e-6
    assert rel_err(x.grad, g["dx"]) < 1e-4
    for k, v in sub(g, "grad/").items():
        assert rel_err(sd[k].grad, v) < 1e-4, k
    with torch.no_grad():
        assert rel_err(oracle.nlayer_d_forward(sd, x, False), g["y_eval"]) < TOL
    mine = oracle.nlayer_d_state(ic, ndf, nl, norm="instance")
    assert [(k, tuple(v.shape)) for k, v in mine.items()] == [(k, tuple(v.shape)) for k, v in sd.items()]


def test_losses():
    g = load_golden("losses")
    b = torch.from_numpy(g["b"])
    for name, fn in (("l1", oracle.l1_loss), ("mse", oracle.mse_loss), ("psnr", oracle.psnr)):
        a = torch.from_numpy(g["a"]).requires_grad_(True)
        v = fn(a, b)
        v.backward()
        assert abs(float(v) - float(g[name])) < 1e-5 * max(1.0, abs(float(g[name])))
        assert rel_err(a.grad, g[name + "_da"]) < 1e-5
    for name, real in (("gan_real", True), ("gan_fake", False)):
        a = torch.from_numpy(g["a"]).requires_grad_(True)
        v = oracle.gan_loss(a, real)
        v.backward()
        assert abs(float(v) - float(g[name])) < 1e-6
        assert rel_err(a.grad, g[name + "_da"]) < 1e-5


def test_preproc():
    g = load_golden("preproc")
    img = torch.from_numpy(g["img"])
    gray = oracle.rgb_to_gray(img)
    assert rel_err(gray, g["gray"]) < 1e-6
    for up in (2, 4):
        assert rel_err(oracle.bilinear_down(gray, up), g[f"bil_down{up}"]) < 1e-6
        assert rel_err(oracle.nearest_down(img, up), g[f"near_down{up}"]) < 1e-6


def test_cas_step_sr_half():
    """SR half of CasSRC.optimize_parameters (trainCas.py:133-145) reproduced from the
    stored initial state: loss_SR, fake_BC, post-Adam weights, lr after update_lr."""
    g = load_golden("cas_step")
    sd = _req(sub(g, "sr0/"))
    realB = torch.from_numpy(g["realB"])
    bc, ba = oracle.cas_forward_sr_inputs(realB, 2)
    assert rel_err(bc, g["real_BC"]) < 1e-6 and rel_err(ba, g["real_BA"]) < 1e-6
    lr = oracle.cosine_lr_sequence(1e-4, 1, 50)[0]
    assert abs(lr - float(g["lr_after"])) < 1e-12
    opt = torch.optim.Adam(list(sd.values()), lr=lr)
    fake = oracle.rddbnet_forward(sd, ba, 2)
    loss = oracle.l1_loss(fake, bc)
    loss.backward()
    opt.step()
    assert rel_err(fake, g["fake_BC"]) < TOL
    assert abs(float(loss) - float(g["loss_SR"])) < 1e-6
    assert abs(float(oracle.psnr(fake.detach(), bc)) - float(g["psnr_SR"])) < 1e-3
    for k, v in sub(g, "sr1/").items():
        assert rel_err(sd[k], v) < 1e-5, k


def test_paired_step():
    g = load_golden("paired_step")
    st = oracle.PairedStepState(sub(g, "g0/"), sub(g, "d0/"), up=2)
    x, y = torch.from_numpy(g["x"]), torch.from_numpy(g["y"])
    for step in range(2):
        out = oracle.paired_step(st, x, y)
        for k in ("loss_G", "loss_D", "loss_G_GAN", "loss_L1"):
            assert abs(out[k] - float(g[f"{k}_{step}"])) < 2e-5 * max(1.0, abs(float(g[f"{k}_{step}"]))), (k, step)
    for k, v in sub(g, "g1/").items():
        assert rel_err(st.g[k], v) < 1e-4, k
    for k, v in sub(g, "d1/").items():
        if v.is_floating_point():
            assert rel_err(st.d[k], v) < 1e-4, k
        else:
            assert int(st.d[k]) == int(v)


def test_cycle_step_runs():
    """Full-cycle restatement (train.py:228-340); G_B is build-defined so this is a
    self-consistency check only ("parity unpinned" for G_B)."""
    st = oracle.make_cycle_state(up=2, nf=16, nb=1, gc=8, ndf=16)
    torch.manual_seed(1)
    a, b = torch.rand(2, 3, 32, 32), torch.rand(2, 3, 64, 64)
    out1 = oracle.cycle_step(st, a, b)
    out2 = oracle.cycle_step(st, a, b)
    assert all(np.isfinite(v) for v in out1.values())
    assert out2["loss_cycle"] < out1["loss_cycle"] * 1.5


LEGACY = [("rddbnetb_x2", "B"), ("rddbnetb_x4", "B"), ("legacy_rddbnet_x1", "L"), ("legacy_rddbnet_x2", "L"), ("legacy_rddbnet_x4", "L")]


@pytest.mark.parametrize("tag,kind", LEGACY)
def test_legacy_generators(tag, kind):
    """model/model.py:347-440 (legacy RDDBNet, RDDBNetB = G_A of the cycle) against reference outputs and gradients."""
    g = load_golden(tag)
    ic, oc, nf, nb, gc, up = [int(v) for v in g["cfg"]]
    mode = f"x{up}"
    sd = _req(sub(g, "sd/"))
    tail = ("upconv1", "upconv2", "HRconv") if kind == "B" else ("upconv", "HRconv")
    assert list(sd.keys()) == oracle.legacy_keys(nb, tail)
    x = torch.from_numpy(g["x"]).requires_grad_(True)
    y = (oracle.rddbnetb_forward if kind == "B" else oracle.legacy_rddbnet_forward)(sd, x, mode)
    loss = oracle.l1_loss(y, torch.from_numpy(g["t"]))
    loss.backward()
    assert rel_err(y, g["y"]) < TOL
    assert abs(float(loss) - float(g["loss"])) < 1e-5
    assert rel_err(x.grad, g["dx"]) < TOL
    grads = sub(g, "grad/")
    for k, v in grads.items():
        assert rel_err(sd[k].grad, v) < TOL, k
    # parameters the reference's forward never touches get no gradient there, and none here
    for k in g["nograd"]:
        assert sd[str(k)].grad is None, k
    assert set(grads) | {str(k) for k in g["nograd"]} == set(sd)


def _fp(t):
    t = t.detach().double().reshape(-1)
    return np.array([float(t.sum()), float((t * t).sum()), *[float(v) for v in t[:4]]])


def _resdeconv_from_cfg(g, **kw):
    """holder built like the fixture's reference instance: cfg = [src, tar, seed] or [src, tar, seed, l0, l1, l2, l3, BN == 'IN']"""
    import srcgan_amd
    cfg = [int(v) for v in g["cfg"]]
    src, tar, seed = cfg[:3]
    layers, BN = (cfg[3:7], "IN" if cfg[7] else "GN") if len(cfg) > 3 else ([2, 2, 2, 2], "GN")
    torch.manual_seed(seed)
    return srcgan_amd.ResDeconv(src, tar, None, layers, BN, **kw)


@pytest.mark.parametrize("tag", ["resdeconv_gray", "resdeconv_rgb", "resdeconv_in", "resdeconv_r34"])
def test_resdeconv(tag):
    """ResDeconv colouriser (reference src/model/resdeconv.py:99-195).  The 60 MB state_dict is not stored: the build's
    parameter holders reproduce the reference's seeded initial weights (fingerprints checked), then the oracle
    restatement must reproduce the reference's output, loss and every parameter gradient's fingerprint."""
    g = load_golden(tag)
    holder = _resdeconv_from_cfg(g)          # (resdeconv_in: BN='IN', no normalisation parameters; resdeconv_r34: layers=[3,4,6,3])
    names = [str(k) for k in g["names"]]
    assert [k for k, _ in holder.named_parameters()] == names
    for k, p in holder.named_parameters():
        assert np.allclose(_fp(p), g["wfp/" + k], rtol=1e-12, atol=0), k       # bit-identical initialisation
    sd = {k: p.detach().clone().requires_grad_(True) for k, p in holder.named_parameters()}
    y = oracle.resdeconv_forward(sd, torch.from_numpy(g["x"]))
    loss = oracle.l1_loss(y, torch.from_numpy(g["t"]))
    loss.backward()
    assert rel_err(y, g["y"]) < TOL
    assert abs(float(loss) - float(g["loss"])) < 1e-5
    for k in names:
        ref = g["gfp/" + k]
        mine = _fp(sd[k].grad)
        assert abs(mine[0] - ref[0]) <= 2e-4 * max(1.0, np.sqrt(ref[1])), k
        assert abs(mine[1] - ref[1]) <= 1e-3 * max(ref[1], 1e-12), k
    for k, v in sub(g, "grad/").items():
        assert rel_err(sd[k].grad, v) < 1e-4, k


@pytest.mark.parametrize("tag", ["espcn_x2", "espcn_x3", "srcnn", "edsr_x2", "edsr_x4"])
def test_small_sr_models(tag):
    """ESPCN (espcn.py; the CLI default --SRModel) and SRCNN (srcnn.py) restatements against reference outputs / gradients,
    and the seeded holders of the package against the reference's initial weights."""
    import srcgan_amd
    g = load_golden(tag)
    cfg = [int(v) for v in g["cfg"]]
    ic, oc, up = cfg[:3]
    ref = sub(g, "sd/")
    sd = _req(ref)
    x = torch.from_numpy(g["x"])
    y = oracle.srcnn_forward(sd, x) if tag == "srcnn" else oracle.edsr_forward(sd, x) if tag.startswith("edsr") else oracle.espcn_forward(sd, x, up)
    loss = oracle.l1_loss(y, torch.from_numpy(g["t"]))
    loss.backward()
    assert rel_err(y, g["y"]) < TOL
    assert abs(float(loss) - float(g["loss"])) < 1e-6
    for k, v in sub(g, "grad/").items():
        assert rel_err(sd[k].grad, v) < TOL, k
    torch.manual_seed(0)
    net = srcgan_amd.EDSR(*cfg) if tag.startswith("edsr") else (srcgan_amd.SRCNN if tag == "srcnn" else srcgan_amd.ESPCN)(ic, oc, up)
    assert list(net.state_dict().keys()) == list(ref.keys())
    for k, v in ref.items():
        assert torch.equal(net.state_dict()[k], v), k


@pytest.mark.parametrize("tag", ["unit", "byte", "signed"])
def test_metrics(tag):
    """oracle restatements of src/metrics.py against values computed by the reference classes (three SSIM dynamic ranges)."""
    g = load_golden("metrics")
    p, t = torch.from_numpy(g[f"{tag}/pred"]), torch.from_numpy(g[f"{tag}/true"])
    assert rel_err(oracle.metric_ae(p, t), g[f"{tag}/ae"]) < 1e-6
    assert rel_err(oracle.mse_loss(p, t), g[f"{tag}/mse"]) < 1e-6
    assert abs(float(oracle.psnr(p, t)) - float(g[f"{tag}/psnr"])) < 1e-4
    s, cs = oracle.metric_ssim(p, t, full=True)
    assert abs(float(s) - float(g[f"{tag}/ssim"])) < 1e-6 and abs(float(cs) - float(g[f"{tag}/cs"])) < 1e-6
    assert rel_err(oracle.metric_ssim(p, t, size_average=False), g[f"{tag}/ssim_per_image"]) < 1e-6


@pytest.mark.parametrize("tag", ["srdn_nb1", "srdn_nb2"])
def test_srdn(tag):
    """SRDN (srdn.py:56-74) restatement against the reference; seeded holder initialisation; trunk_conv gets no gradient."""
    import srcgan_amd
    g = load_golden(tag)
    cfg = [int(v) for v in g["cfg"]]
    ref = sub(g, "sd/")
    sd = _req(ref)
    x = torch.from_numpy(g["x"]).requires_grad_(True)
    y = oracle.srdn_forward(sd, x)
    oracle.l1_loss(y, torch.from_numpy(g["t"])).backward()
    assert rel_err(y, g["y"]) < TOL and rel_err(x.grad, g["dx"]) < TOL
    for k, v in sub(g, "grad/").items():
        assert rel_err(sd[k].grad, v) < TOL, k
    assert {str(k) for k in g["nograd"]} == {"trunk_conv.weight", "trunk_conv.bias"}
    torch.manual_seed(0)
    net = srcgan_amd.SRDN(*cfg)
    assert list(net.state_dict().keys()) == list(ref.keys())
    for k, v in ref.items():
        assert torch.equal(net.state_dict()[k], v), k


def test_colour_restatement_known_answers():
    """dataset.py:114-159 colour conversions (scikit-image, absent here: PARITY UNPINNED).  The float64 restatement is anchored
    on CIE L*a*b* (D65, 2 degree) known answers for white, black, mid-grey and the sRGB primaries, the luma weights, and the
    8-bit round trip of `_lab2img` (truncating store: within one code value)."""
    import numpy as np
    from oracle import srcgan_oracle as O
    px = np.array([[[255, 255, 255], [0, 0, 0], [255, 0, 0], [0, 255, 0], [0, 0, 255], [128, 128, 128]]], dtype=np.uint8)
    lab = O._rgb2lab_f64(px)[0]
    known = np.array([[100.0, 0.0, 0.0], [0.0, 0.0, 0.0], [53.24, 80.09, 67.20], [87.73, -86.18, 83.18], [32.30, 79.19, -107.86], [53.585, 0.0, 0.0]])
    assert np.abs(lab - known).max() < 0.02
    g = O.arr2gray(px)[0, 0].numpy()
    assert np.allclose(g, [1.0, 0.0, 0.2125, 0.7154, 0.0721, 128 / 255], atol=1e-6)
    n = O.arr2lab(px)
    assert n.shape == (3, 1, 6) and float(n.min()) >= 0.0 and float(n.max()) <= 1.0
    assert torch.equal(O.arr2ab(px), n[1:])
    assert torch.allclose(O.arr2rgb(px)[:, 0, 2], torch.tensor([1.0, 0.0, 0.0]))
    rng = np.random.default_rng(0)
    im = rng.integers(0, 256, (48, 40, 3), dtype=np.uint8)
    back = O.lab2img(O.arr2lab(im).numpy().transpose(1, 2, 0))
    assert np.abs(back.astype(int) - im.astype(int)).max() <= 1


def _depth_probe(n, k):
    return np.cos(np.arange(n, dtype=np.float64) * 0.37 + k)


def depth_case_check(g, y, dx, grads, f32_tol=1e-3):
    """Compare one evaluation of RDDBNet(3,3,4) at nb=23 (y, dx, {name: grad}) with the reference fixture rddbnet_nb23.npz.
    The fixture holds the reference's float32 AND float64 results: through 345 LeakyReLUs the reference's own f32 input
    gradient is 2.9e-3 from its f64 one, so gradients are gated against the EXACT result with the reference's own f32 error as
    the yardstick: err <= max(f32_tol, 3 x |ref32 - ref64|).  Returns the worst observed figures."""
    names = [str(n) for n in g["names"]]
    assert rel_err(y, g["y64"]) < f32_tol
    e_dx, ref_dx = rel_err(dx, g["dx64"]), rel_err(g["dx32"], g["dx64"])
    assert e_dx < max(f32_tol, 3 * ref_dx), (e_dx, ref_dx)
    gn = np.array([float(grads[n].double().norm()) for n in names])
    gp = np.array([float(np.dot(grads[n].double().numpy().ravel(), _depth_probe(grads[n].numel(), i))) for i, n in enumerate(names)])
    e_norm, ref_norm = np.abs(gn / g["gnorm64"] - 1).max(), np.abs(g["gnorm32"] / g["gnorm64"] - 1).max()
    assert e_norm < max(f32_tol, 3 * ref_norm), (e_norm, ref_norm)
    # projections: error relative to the gradient's norm x the probe's norm (the projection itself can be near zero)
    scale = g["gnorm64"] * np.array([np.linalg.norm(_depth_probe(grads[n].numel(), i)) for i, n in enumerate(names)])
    e_proj, ref_proj = (np.abs(gp - g["gproj64"]) / scale).max(), (np.abs(g["gproj32"] - g["gproj64"]) / scale).max()
    assert e_proj < max(f32_tol, 3 * ref_proj), (e_proj, ref_proj)
    worst_full = 0.0
    for k, v in g.items():
        if k.startswith("grad64/"):
            n = k[len("grad64/"):]
            e, ref = rel_err(grads[n], v), rel_err(g["grad32/" + n], v)
            assert e < max(f32_tol, 3 * ref), (n, e, ref)
            worst_full = max(worst_full, e)
    return {"dx": e_dx, "ref_dx": ref_dx, "norm": e_norm, "proj": e_proj, "full": worst_full}


def depth_case_state(g, module_cls):
    """The fixture's network: the reference initialisation under torch.manual_seed(seed), verified by one checksum per parameter."""
    torch.manual_seed(int(g["seed"]))
    net = module_cls(3, 3, 4)
    sums = np.array([float(p.detach().double().sum()) for p in net.parameters()])
    assert [k for k, _ in net.named_parameters()] == [str(n) for n in g["names"]]
    assert np.array_equal(sums, g["param_sum"]), "initialisation differs from the reference's under the same seed"
    return net


def test_rddbnet_nb23_reference_fixture():
    """The oracle at the benchmark's depth (RDDBNet(3,3,4): nf=64, nb=23, gc=32) against the REFERENCE's own run
    (tests/golden/make_golden_depth.py): same-seed initialisation, output, input gradient, all 697 parameter gradients."""
    import srcgan_amd
    g = load_golden("rddbnet_nb23")
    net = depth_case_state(g, srcgan_amd.RDDBNet)
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in net.state_dict().items()}
    x = torch.from_numpy(g["x"]).requires_grad_(True)
    y = oracle.rddbnet_forward(sd, x, 4)
    loss = oracle.mse_loss(y, torch.from_numpy(g["t"]))
    loss.backward()
    assert rel_err(y, g["y32"]) < TOL and abs(float(loss.detach()) - float(g["loss32"])) < 1e-6
    depth_case_check(g, y.detach(), x.grad, {k: v.grad for k, v in sd.items()})
