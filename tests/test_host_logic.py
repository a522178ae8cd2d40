"""CPU: host-side mirror of the reference interface -- names, constructor signatures, state_dict keys / shapes,
initialisation statistics, harness bookkeeping.  Nothing here needs (or may silently fall back from) a GPU."""
import math
import random

import numpy as np
import pytest
import torch

import oracle
import srcgan_amd
from srcgan_amd import train as T
from conftest import load_golden, sub


def test_exports_reference_names():
    for name in ("RDDBNet", "RDDBNetA", "NLayerDiscriminator", "L1Loss", "MSELoss", "PSNRLoss", "GANLoss"):
        assert hasattr(srcgan_amd, name)
    assert repr(srcgan_amd.L1Loss()) == "L1" and repr(srcgan_amd.MSELoss()) == "MSE" and repr(srcgan_amd.PSNRLoss()) == "PSNR"


@pytest.mark.parametrize("tag", ["rddbnet_x2", "rddbnet_x4", "rddbnet_x2_w32"])
def test_rddbnet_state_dict_matches_reference(tag):
    g = load_golden(tag)
    ic, oc, up, nf, nb, gc = [int(v) for v in g["cfg"]]
    ref = sub(g, "sd/")
    net = srcgan_amd.RDDBNet(ic, oc, up, nf=nf, nb=nb, gc=gc)
    sd = net.state_dict()
    assert list(sd.keys()) == list(ref.keys())
    assert [tuple(v.shape) for v in sd.values()] == [tuple(v.shape) for v in ref.values()]
    assert all(v.dtype == torch.float32 for v in sd.values())
    net.load_state_dict(ref, strict=True)          # reference checkpoints load
    assert [n for n, _ in net.named_parameters()] == oracle.rddbnet_keys(nb, up)


@pytest.mark.parametrize("tag", ["nlayerd_3", "nlayerd_2"])
def test_nlayerd_state_dict_matches_reference(tag):
    g = load_golden(tag)
    ic, ndf, nl = [int(v) for v in g["cfg"]]
    ref = sub(g, "sd/")
    net = srcgan_amd.NLayerDiscriminator(ic, ndf, nl)
    sd = net.state_dict()
    assert list(sd.keys()) == list(ref.keys())
    assert [tuple(v.shape) for v in sd.values()] == [tuple(v.shape) for v in ref.values()]
    net.load_state_dict(ref, strict=True)


def test_nlayerd_instance_norm_state_dict_matches_reference():
    """norm_layer = nn.InstanceNorm2d, as the class or as basicModel.get_norm_layer's functools.partial (basicModel.py:24-25): biased
    convolutions, no normalisation parameters or buffers -- the reference's state_dict loads strictly.  Other norm layers are refused."""
    import functools
    g = load_golden("nlayerd_in")
    ic, ndf, nl = [int(v) for v in g["cfg"]]
    ref = sub(g, "sd/")
    for nl_arg in (torch.nn.InstanceNorm2d, functools.partial(torch.nn.InstanceNorm2d, affine=False, track_running_stats=False)):
        net = srcgan_amd.NLayerDiscriminator(ic, ndf, nl, norm_layer=nl_arg)
        sd = net.state_dict()
        assert list(sd.keys()) == list(ref.keys())
        assert [tuple(v.shape) for v in sd.values()] == [tuple(v.shape) for v in ref.values()]
        net.load_state_dict(ref, strict=True)
    for bad in (torch.nn.GroupNorm, functools.partial(torch.nn.InstanceNorm2d, affine=True), functools.partial(torch.nn.BatchNorm2d, affine=False)):
        with pytest.raises(NotImplementedError):
            srcgan_amd.NLayerDiscriminator(ic, ndf, nl, norm_layer=bad)


def test_full_size_parameter_counts():
    # 16 619 968 params / 697 tensors ; 2 765 633 / 13
    g = srcgan_amd.RDDBNet(3, 3, 4, nb=23)
    assert sum(p.numel() for p in g.parameters()) == 16619968 and len(list(g.parameters())) == 697
    d = srcgan_amd.NLayerDiscriminator(3, 64, 3)
    assert sum(p.numel() for p in d.parameters()) == 2765633 and len(list(d.parameters())) == 13


def test_same_seed_gives_reference_initialisation():
    """parameter holders are torch modules created in the reference's order, so the same seed reproduces
    the reference's initial weights (stored in the golden file) exactly."""
    g = load_golden("rddbnet_x2")
    torch.manual_seed(0)
    net = srcgan_amd.RDDBNet(3, 3, 2, nf=16, nb=1, gc=8)
    for k, v in sub(g, "sd/").items():
        assert torch.equal(net.state_dict()[k], v), k


def test_init_statistics():
    torch.manual_seed(1)
    net = srcgan_amd.RDDBNet(3, 3, 4, nf=64, nb=1, gc=32)
    w = net.RRDB_trunk[0].RDB1.conv5.weight          # kaiming normal, fan_out = 64*9
    assert abs(float(w.std()) - math.sqrt(2.0 / (64 * 9))) < 2e-3
    dw = net.upscale_layers[0].weight                 # ConvTranspose2d keeps torch's default: U(+-1/sqrt(64*4))
    assert float(dw.abs().max()) <= 0.0625 + 1e-6 and float(dw.abs().max()) > 0.06


def test_no_cpu_fallback_anywhere():
    x = torch.rand(1, 3, 8, 8)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        srcgan_amd.RDDBNet(3, 3, 2, nf=16, nb=1, gc=8)(x)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        srcgan_amd.NLayerDiscriminator(3, 16, 2)(torch.rand(1, 3, 32, 32))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        srcgan_amd.L1Loss()(x, x)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        srcgan_amd.GANLoss("lsgan")(x, True)


def test_ganloss_modes():
    g = srcgan_amd.GANLoss("lsgan", device="cpu", target_real_label=0.9)
    assert float(g.real_label) == pytest.approx(0.9) and float(g.fake_label) == 0.0
    assert g.get_target_tensor(torch.zeros(2, 1, 3, 3), True).shape == (2, 1, 3, 3)
    with pytest.raises(NotImplementedError):
        srcgan_amd.GANLoss("hinge")
    for mode in ("vanilla", "wgangp"):                      # the other objectives of train.py:88-95 are native too (round 3)
        assert srcgan_amd.GANLoss(mode, device="cpu").gan_mode == mode
    with pytest.raises(NotImplementedError):
        srcgan_amd.GANLoss("DSSIM")
    # the oracle's restatement of the three objectives against the torch modules the reference's class wraps (train.py:86-95)
    x = torch.randn(2, 1, 5, 7) * 4
    assert torch.allclose(oracle.gan_loss(x, True, gan_mode="vanilla"), torch.nn.BCEWithLogitsLoss()(x, torch.tensor(1.0).expand_as(x)))
    assert torch.allclose(oracle.gan_loss(x, False, gan_mode="vanilla"), torch.nn.BCEWithLogitsLoss()(x, torch.tensor(0.0).expand_as(x)))
    assert torch.allclose(oracle.gan_loss(x, True, gan_mode="wgangp"), -x.mean()) and torch.allclose(oracle.gan_loss(x, False, gan_mode="wgangp"), x.mean())
    assert torch.allclose(oracle.gan_loss(x, False, gan_mode="lsgan"), torch.nn.MSELoss()(x, torch.tensor(0.0).expand_as(x)))


def test_unsupported_norm_layer_is_rejected():
    with pytest.raises(NotImplementedError):
        srcgan_amd.NLayerDiscriminator(3, 64, 3, norm_layer=torch.nn.LayerNorm)


def test_image_pool_matches_oracle():
    a, b = random.Random(3), random.Random(3)
    mine, ref = T.ImagePool(4, a), oracle.ImagePoolOracle(4, b)
    torch.manual_seed(0)
    for _ in range(12):
        batch = torch.rand(2, 3, 4, 4)
        assert torch.equal(mine.query(batch), ref.query(batch))
    assert torch.equal(T.ImagePool(0).query(batch), batch)


def test_cas_lr_schedule_matches_reference():
    g = load_golden("cas_step")
    seq = oracle.cosine_lr_sequence(1e-4, 3, 50)
    assert seq[0] == pytest.approx(float(g["lr_after"]), rel=1e-9)     # measured on the reference CasSRC.update_lr

    class Dummy(T.CasSRC):
        def __init__(self):
            p = [torch.nn.Parameter(torch.zeros(1))]
            self.optimizers = [torch.optim.Adam(p, lr=1e-4), torch.optim.Adam(p, lr=1e-4)]
    m, opt = Dummy(), T.CasParams(device="cpu")
    for e in range(3):
        m.update_lr(opt)
        assert m.optimizers[0].param_groups[0]["lr"] == pytest.approx(seq[e], rel=1e-9)
    opt.lr_policy = "bogus"
    assert isinstance(m.update_lr(opt), NotImplementedError)            # the reference returns, not raises (trainCas.py:61)


def test_set_requires_grad():
    d = srcgan_amd.NLayerDiscriminator(3, 16, 2)
    T.set_requires_grad([d, None], False)
    assert not any(p.requires_grad for p in d.parameters())
    T.set_requires_grad(d, True)
    assert all(p.requires_grad for p in d.parameters())


def test_dtype_selection():
    assert srcgan_amd.RDDBNet(3, 3, 2, nf=16, nb=1, gc=8).compute_dtype == "fp32"
    assert srcgan_amd.RDDBNet(3, 3, 2, nf=16, nb=1, gc=8, dtype="bf16").compute_dtype == "bf16"
    with pytest.raises(ValueError):
        srcgan_amd.RDDBNet(3, 3, 2, nf=16, nb=1, gc=8, dtype="fp8")


@pytest.mark.parametrize("tag,kind", [("rddbnetb_x2", "B"), ("rddbnetb_x4", "B"), ("legacy_rddbnet_x2", "L")])
def test_legacy_generators_state_dict_and_seeded_init(tag, kind):
    """RDDBNetB / legacy RDDBNet (reference model/model.py:347-440): reference state_dict keys, shapes and -- modules being
    created and re-initialised in the reference's order -- bit-identical seeded initial weights."""
    g = load_golden(tag)
    ic, oc, nf, nb, gc, up = [int(v) for v in g["cfg"]]
    ref = sub(g, "sd/")
    cls = srcgan_amd.RDDBNetB if kind == "B" else srcgan_amd.LegacyRDDBNet
    torch.manual_seed(0)
    net = cls(ic, oc, nf, nb, gc, f"x{up}")
    sd = net.state_dict()
    assert list(sd.keys()) == list(ref.keys())
    assert [n for n, _ in net.named_parameters()] == oracle.legacy_keys(nb, ("upconv1", "upconv2", "HRconv") if kind == "B" else ("upconv", "HRconv"))
    for k, v in ref.items():
        assert torch.equal(sd[k], v), k
    net.load_state_dict(ref, strict=True)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        net(torch.rand(1, ic, 8, 8))
    with pytest.raises(NotImplementedError):
        srcgan_amd.RDDBNetB(3, 3, 16, 1, 8, "x8")


def test_resdeconv_holder():
    """ResDeconv colouriser: reference parameter count / state_dict layout, no CPU fallback, argument checks."""
    net = srcgan_amd.ResDeconv(1, 3)
    assert sum(p.numel() for p in net.parameters()) == 14982912 and len(list(net.parameters())) == 101
    ks = list(net.state_dict().keys())
    assert ks[:3] == ["conv1.weight", "bn1.weight", "bn1.bias"] and ks[-1] == "pred.weight"
    assert "layer2.0.downsample.1.bias" in ks and "layer1.0.downsample.0.weight" not in ks
    assert tuple(net.deconv10.weight.shape) == (512, 256, 2, 2) and tuple(net.conv1.weight.shape) == (64, 3, 7, 7)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        net(torch.rand(1, 1, 32, 32))
    # the reference's positional signature (resdeconv.py:107): src_ch, tar_ch, block, layers, BN
    srcgan_amd.ResDeconv(1, 3, None, [2, 2, 2, 2], "GN")
    r34 = srcgan_amd.ResDeconv(1, 3, None, [3, 4, 6, 3], "GN")              # ResNet-34 layout: 191 parameter tensors
    assert len(list(r34.parameters())) == 191 and "layer3.5.bn2.bias" in r34.state_dict() and "upRes1.5.conv2.weight" in r34.state_dict()
    inn = srcgan_amd.ResDeconv(1, 3, None, [2, 2, 2, 2], "IN")              # InstanceNorm2d: convolution weights only
    assert len(list(inn.parameters())) == 37 and all("bn" not in k and "downsample.1" not in k for k in inn.state_dict())
    for bad in ((None, [2, 2, 2, 2], "BN"), (torch.nn.Identity, [2, 2, 2, 2], "GN")):
        with pytest.raises(NotImplementedError):
            srcgan_amd.ResDeconv(1, 3, *bad)
    with pytest.raises(ValueError):
        srcgan_amd.ResDeconv(1, 3, None, [2, 2, 2], "GN")
    from srcgan_amd import train as T
    assert T.CasParams().CModel == "ResDeconv" and T.MODEL_REGISTRY["ResDeconv"] is srcgan_amd.ResDeconv


def test_checkpoint_names_round_trip(tmp_path):
    """trainCas.py:221-225 writes '<Model>_A2C_x<up>_<epoch:04d>.pth' / '<CModel>_C2B_...'; testCas.py:41-56 splits the basename on
    '_' to rebuild both networks.  Names, parsing, and a save -> load round trip of two (tiny) networks with reference keys."""
    from srcgan_amd import data as D
    assert D.checkpoint_name("RDDBNet", "A2C", 4, 25) == "RDDBNet_A2C_x4_0025.pth"
    assert D.checkpoint_name("ResDeconv", "C2B", 2, 300) == "ResDeconv_C2B_x2_0300.pth"
    assert D.parse_checkpoint_name("./checkpoints/ESPCN_A2C_x2_0050.pth") == ("ESPCN", "A2C", 2, 50)
    with pytest.raises(ValueError):
        D.parse_checkpoint_name("weights.pth")
    with pytest.raises(ValueError):
        D.checkpoint_name("RDDBNet", "G", 2, 1)
    reg = {"RDDBNetTiny": lambda i, o, up: srcgan_amd.RDDBNet(i, o, up, nf=16, nb=1, gc=8),
           "ColourTiny": lambda i, o: srcgan_amd.RDDBNet(i, o, 1, nf=16, nb=1, gc=8)}

    class _M:
        pass
    m, opt = _M(), _M()
    torch.manual_seed(3)
    m.netG_A2C, m.netG_C2B = reg["RDDBNetTiny"](1, 1, 2), reg["ColourTiny"](1, 3)
    opt.SRModel, opt.CModel, opt.up = "RDDBNetTiny", "ColourTiny", 2
    pa, pb = D.save_checkpoints(m, opt, 25, root=str(tmp_path))
    assert pa.endswith("RDDBNetTiny_A2C_x2_0025.pth") and pb.endswith("ColourTiny_C2B_x2_0025.pth")
    na, nb = D.load_cascade(pa, pb, device="cpu", registry=reg)
    assert not na.training and not nb.training
    for a, b in ((na, m.netG_A2C), (nb, m.netG_C2B)):
        sa, sb = a.state_dict(), b.state_dict()
        assert list(sa) == list(sb) and all(torch.equal(sa[k], sb[k]) for k in sa)
    with pytest.raises(KeyError):
        D.load_cascade(pa.replace("RDDBNetTiny", "Nope"), pb, device="cpu", registry=reg)


def test_input_pipeline_needs_the_device():
    """The colour conversions are product code: no CPU fallback, host tensors are refused."""
    from srcgan_amd import data as D
    with pytest.raises(RuntimeError):
        D.arr2lab(torch.zeros(4, 4, 3, dtype=torch.uint8))


def test_bench_launcher_reports_a_failed_rank():
    """`python bench.py --gpus 2` without WORLD_SIZE is the launcher: with no GPU here the rank processes exit non-zero
    ("needs an MI355X") and the parent must pass that on instead of hanging or printing a JSON line."""
    import os, subprocess, sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    if torch.cuda.is_available():
        pytest.skip("CPU-only check")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    # the launcher counts devices without opening HIP (visibility variables / KFD sysfs): "only 0 GPU(s)" where the topology is
    # readable, otherwise the ranks themselves report the missing device
    assert out.returncode != 0 and ("only 0 GPU" in out.stderr or "MI355X" in out.stderr)
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]
    env["SRCGAN_LOCAL_DEVICE"] = "0"          # skip the device count: the ranks themselves fail
    env["SRCGAN_DIST_BACKEND"] = "gloo"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and not [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert "MI355X" in out.stderr


# ------------------------------------------------------------------------------------------------ folder datasets (dataset.py)
def _make_folder_dataset(tmp_path, n=5, hw=(12, 16), up=1):
    """A miniature '<root>/{train,val,test}.txt + src/ + tar/' tree in the reference's layout (dataset.py:39-45)."""
    from PIL import Image
    rng = np.random.default_rng(3)
    root = tmp_path / "Mini"
    (root / "src").mkdir(parents=True), (root / "tar").mkdir()
    arrays = {}
    names = ["im%02d.png" % i for i in range(n)]
    for nm in names:
        tar = rng.integers(0, 256, (hw[0] * up, hw[1] * up, 3), dtype=np.uint8)
        src = rng.integers(0, 256, (hw[0], hw[1]), dtype=np.uint8)              # a grey file: .convert('RGB') replicates it
        Image.fromarray(src).save(root / "src" / nm), Image.fromarray(tar).save(root / "tar" / nm)
        arrays[nm] = (np.stack([src] * 3, -1), tar)
    (root / "train.txt").write_text("\n".join(names[:3]) + "\n")
    (root / "val.txt").write_text(names[3] + "\n")
    (root / "test.txt").write_text("  " + names[4] + "  \n")                    # lines are stripped (dataset.py:42)
    return names, arrays


def test_folder_dataset_lists_and_decodes(tmp_path):
    from srcgan_amd import data as D
    names, arrays = _make_folder_dataset(tmp_path)
    train, val, test = D.load_dataset("Mini", "G2LAB", dataset_dir=str(tmp_path))
    assert (len(train), len(val), len(test)) == (3, 1, 1) and train.ver == "G2LAB" and (train.src_ch, train.tar_ch) == (1, 3)
    assert isinstance(train, torch.utils.data.Dataset) and isinstance(train, D.Basic)
    for ds, idx, nm in ((train, 2, names[2]), (test, 0, names[4])):
        s = ds[idx]
        assert s["idx"] == idx and s["src"].dtype == torch.uint8
        assert np.array_equal(s["src"].numpy(), arrays[nm][0]) and np.array_equal(s["tar"].numpy(), arrays[nm][1])
    # the transform contract of dataset.py:183-190: PIL images in, arrays out
    flip = lambda smp: {k: np.asarray(v)[:, ::-1] for k, v in smp.items()}
    ds = D.G2RGB("Mini", "train", transform=flip, dataset_dir=str(tmp_path))
    assert np.array_equal(ds[0]["tar"].numpy(), arrays[names[0]][1][:, ::-1])
    with pytest.raises(ValueError):
        D.G2RGB("Mini", "train", transform=lambda smp: {k: np.asarray(v, np.float32) for k, v in smp.items()}, dataset_dir=str(tmp_path))[0]
    with pytest.raises(FileNotFoundError):
        D.G2RGB("Mini", "all", dataset_dir=str(tmp_path))                       # no all.txt in the tree
    with pytest.raises(KeyError):
        D.load_dataset("Mini", "G2XYZ", dataset_dir=str(tmp_path))
    # host-side batching is plain DataLoader collation of the decoded pairs
    batch = next(iter(torch.utils.data.DataLoader(train, 3)))
    assert batch["src"].shape == (3, 12, 16, 3) and batch["src"].dtype == torch.uint8 and batch["idx"].tolist() == [0, 1, 2]
    assert len(D.G2LAB()) == 0                                                   # converter form: no file list
