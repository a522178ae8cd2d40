"""CPU restatement (pure PyTorch, fp32) of the SRCGAN training hot path.

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.  Every network is a
*function of a state_dict* (same key names / shapes as the reference modules)
so the product modules and this oracle can be evaluated on literally the same
weights.  Each function cites the reference lines it restates.

Parity: pinned against vectors produced by the imported reference
(tests/golden/make_golden.py -> tests/golden/*.npz, tests/test_oracle_golden.py).
"""
from __future__ import annotations

import math
import random
from typing import Dict, Iterable, List, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
State = Dict[str, Tensor]

LRELU_SLOPE = 0.2      # rddb.py:60,96 ; model/model.py:612-631
RES_SCALE = 0.2        # "lemda" default, rddb.py:62,78
BN_EPS = 1e-5          # nn.BatchNorm2d default used at model/model.py:598
BN_MOMENTUM = 0.1

__all__ = [
    "LRELU_SLOPE", "RES_SCALE", "rdb_forward", "rrdb_forward", "rddbnet_forward",
    "rddbnet_state", "rddbnet_keys", "nlayer_d_forward", "nlayer_d_state",
    "nlayer_d_keys", "l1_loss", "mse_loss", "psnr", "gan_loss", "rgb_to_gray",
    "bilinear_down", "nearest_down", "cas_forward_sr_inputs", "ImagePoolOracle",
    "paired_step", "PairedStepState", "make_paired_state", "rddbneta_forward",
    "rddbneta_state", "cycle_step", "CycleState", "make_cycle_state", "cosine_lr_sequence",
    "rddbnetb_forward", "legacy_rddbnet_forward", "legacy_keys", "resdeconv_forward", "espcn_forward", "srcnn_forward", "edsr_forward", "srdn_forward", "metric_ae", "metric_ssim",
    "arr2gray", "arr2rgb", "arr2lab", "arr2ab", "lab2img", "storage",
]


# ---------------------------------------------------------------------------
# Storage-dtype emulation (test infrastructure for the bf16 perf mode).
# The native bf16 mode STORES activations and packed conv weights in bf16 (f32 accumulate, f32 biases / norm parameters /
# statistics).  ``with storage(torch.bfloat16):`` makes the forward functions below round at exactly those storage points
# (straight-through for autograd: the backward stays f32), so a test can separate what bf16 storage does to a result --
# present in ANY bf16 implementation, and amplified by ill-conditioned steps such as a BatchNorm backward behind a constant
# lsgan label -- from what the kernels add to it.  Default: no rounding, the functions are the reference's arithmetic.
# ---------------------------------------------------------------------------
_STORAGE = None


class storage:
    def __init__(self, dtype):
        self.dtype = dtype

    def __enter__(self):
        global _STORAGE
        self._old, _STORAGE = _STORAGE, self.dtype
        return self

    def __exit__(self, *a):
        global _STORAGE
        _STORAGE = self._old


def _st(t: Tensor) -> Tensor:
    """value as the native path holds it in HBM (identity unless inside ``storage(dtype)``)."""
    if _STORAGE is None or t is None:
        return t
    return t + (t.to(_STORAGE).to(t.dtype) - t).detach()


# ---------------------------------------------------------------------------
# Generator: RDDBNet  (reference src/model/rddb.py:48-114)
# ---------------------------------------------------------------------------
def _lrelu(t: Tensor) -> Tensor:
    return F.leaky_relu(t, LRELU_SLOPE)


def rdb_forward(sd: State, pre: str, x: Tensor) -> Tensor:
    """ResidualDenseBlock_5.forward, rddb.py:62-68: five 3x3 convs over a growing
    concatenation, LeakyReLU(0.2) after the first four, out = 0.2*x5 + x."""
    feats = [x]
    for k in range(1, 6):
        inp = feats[0] if len(feats) == 1 else torch.cat(feats, 1)
        y = F.conv2d(inp, _st(sd[f"{pre}conv{k}.weight"]), sd[f"{pre}conv{k}.bias"], 1, 1)
        if k < 5:
            feats.append(_st(_lrelu(y)))
    return y * RES_SCALE + x


def rrdb_forward(sd: State, pre: str, x: Tensor) -> Tensor:
    """RRDB.forward, rddb.py:78-82: three dense blocks, out = 0.2*out + x.
    (Storage emulation: the native conv5 epilogue stores a block's output once; RDB3's store already includes the RRDB skip.)"""
    out = x
    for j in (1, 2, 3):
        out = rdb_forward(sd, f"{pre}RDB{j}.", out)
        if j < 3:
            out = _st(out)
    return _st(out * RES_SCALE + x)


def rddbnet_forward(sd: State, x: Tensor, upscale_factor: int) -> Tensor:
    """RDDBNet.forward, rddb.py:107-114.  Number of RRDBs is read from the keys.
    Up-sampler: ConvTranspose2d(k=2,s=2,p=0,no bias)+LeakyReLU per x2 stage
    (rddb.py:9-38,93-97)."""
    nb = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("RRDB_trunk."))
    fea = _st(F.conv2d(_st(x), _st(sd["conv_first.weight"]), sd["conv_first.bias"], 1, 1))
    t = fea
    for i in range(nb):
        t = rrdb_forward(sd, f"RRDB_trunk.{i}.", t)
    t = F.conv2d(t, _st(sd["trunk_conv.weight"]), sd["trunk_conv.bias"], 1, 1)
    fea = _st(fea + t)
    if upscale_factor != 1:
        for s in range(int(math.log2(upscale_factor))):
            fea = _st(_lrelu(F.conv_transpose2d(fea, _st(sd[f"upscale_layers.{2 * s}.weight"]), None, 2, 0)))
    return _st(F.conv2d(fea, _st(sd["conv_last.weight"]), None, 1, 1))


def rddbnetb_forward(sd: State, x: Tensor, mode: str) -> Tensor:
    """Legacy RDDBNetB.forward, model/model.py:417-440 (G_A of train.py:172,177): nearest x2 + upconv1 / upconv2,
    HRconv applied eight times (shared weights), conv_last with bias.  Modes other than 'x2' / 'x4' leave the resolution
    unchanged in the reference (no branch taken)."""
    nb = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("RRDB_trunk."))
    fea = F.conv2d(x, sd["conv_first.weight"], sd["conv_first.bias"], 1, 1)
    t = fea
    for i in range(nb):
        t = rrdb_forward(sd, f"RRDB_trunk.{i}.", t)
    fea = fea + F.conv2d(t, sd["trunk_conv.weight"], sd["trunk_conv.bias"], 1, 1)
    up = lambda z: F.interpolate(z, scale_factor=2, mode="nearest")
    c = lambda name, z: _lrelu(F.conv2d(z, sd[name + ".weight"], sd[name + ".bias"], 1, 1))
    if mode == "x4":
        fea = c("upconv1", up(fea))
        fea = c("upconv2", up(fea))
    elif mode == "x2":
        fea = c("upconv1", up(fea))
        fea = c("upconv1", fea)
    for _ in range(8):
        fea = c("HRconv", fea)
    return F.conv2d(fea, sd["conv_last.weight"], sd["conv_last.bias"], 1, 1)


def legacy_rddbnet_forward(sd: State, x: Tensor, mode: str) -> Tensor:
    """Legacy RDDBNet.forward, model/model.py:381-391: the trunk result is computed and DISCARDED there, so the output
    does not depend on the RRDB / trunk_conv parameters (restated without the dead computation)."""
    fea = F.conv2d(x, sd["conv_first.weight"], sd["conv_first.bias"], 1, 1)
    up = lambda z: F.interpolate(z, scale_factor=2, mode="nearest")
    c = lambda name, z: _lrelu(F.conv2d(z, sd[name + ".weight"], sd[name + ".bias"], 1, 1))
    if mode == "x4":
        fea = c("upconv", up(fea))
        fea = c("upconv", up(fea))
    elif mode == "x2":
        fea = c("upconv", up(fea))
    elif mode == "x1":
        fea = c("upconv", fea)
    fea = c("HRconv", fea)
    return F.conv2d(c("HRconv", fea), sd["conv_last.weight"], sd["conv_last.bias"], 1, 1)


def legacy_keys(nb: int, tail) -> List[str]:
    """state_dict key order of the legacy generators (tail = ('upconv1','upconv2','HRconv') or ('upconv','HRconv'))."""
    keys = ["conv_first.weight", "conv_first.bias"]
    for i in range(nb):
        for j in (1, 2, 3):
            for k in range(1, 6):
                keys += [f"RRDB_trunk.{i}.RDB{j}.conv{k}.weight", f"RRDB_trunk.{i}.RDB{j}.conv{k}.bias"]
    keys += ["trunk_conv.weight", "trunk_conv.bias"]
    for t in tail:
        keys += [t + ".weight", t + ".bias"]
    return keys + ["conv_last.weight", "conv_last.bias"]


# ---------------------------------------------------------------------------
# Colouriser: ResDeconv  (reference src/model/resdeconv.py:56-195, BN='GN', layers=[2,2,2,2])
# ---------------------------------------------------------------------------

def _rd_norm(sd: State, key: str, t: Tensor) -> Tensor:
    """the block's normalisation layer: nn.GroupNorm(32, C) (BN='GN', parameters in the state_dict) or nn.InstanceNorm2d(C)
    (BN='IN': no parameters, resdeconv.py:117-121,150-155)."""
    if key + ".weight" in sd:
        return F.group_norm(t, 32, sd[key + ".weight"], sd[key + ".bias"], 1e-5)
    return F.instance_norm(t, None, None, None, None, True, 0.1, 1e-5)


def _rd_block(sd: State, pre: str, x: Tensor, stride: int) -> Tensor:
    """BasicBlock.forward, resdeconv.py:78-97 (norm, ReLU; 1x1 strided conv + norm shortcut when present)."""
    out = _st(F.relu(_rd_norm(sd, pre + "bn1", _st(F.conv2d(x, _st(sd[pre + "conv1.weight"]), None, stride, 1)))))
    out = _rd_norm(sd, pre + "bn2", _st(F.conv2d(out, _st(sd[pre + "conv2.weight"]), None, 1, 1)))
    idn = x
    if pre + "downsample.0.weight" in sd:
        idn = _st(_rd_norm(sd, pre + "downsample.1", _st(F.conv2d(x, _st(sd[pre + "downsample.0.weight"]), None, stride, 0))))
    return _st(F.relu(out + idn))


def resdeconv_forward(sd: State, x: Tensor) -> Tensor:
    """ResDeconv.forward, resdeconv.py:164-195, for any ``layers`` (the block count of a stage is read off the state_dict keys) and
    BN = 'GN' / 'IN'.  A 1-channel source is replicated to the stem's 3 channels."""
    if x.shape[1] == 1:
        x = torch.cat([x, x, x], dim=1)
    nblk = lambda name: 1 + max(int(k.split(".")[1]) for k in sd if k.startswith(name + "."))
    t = _st(F.relu(_rd_norm(sd, "bn1", _st(F.conv2d(_st(x), _st(sd["conv1.weight"]), None, 2, 3)))))
    for name, stride in (("layer1", 1), ("layer2", 2), ("layer3", 2), ("layer4", 2)):
        for k in range(nblk(name)):
            t = _rd_block(sd, f"{name}.{k}.", t, stride if k == 0 else 1)
    for dc, name in (("deconv10", "upRes1"), ("deconv11", "upRes2"), ("deconv12", "upRes3")):
        t = _st(F.conv_transpose2d(t, _st(sd[dc + ".weight"]), None, 2, 0))
        for k in range(nblk(name)):
            t = _rd_block(sd, f"{name}.{k}.", t, 1)
    t = _st(F.conv_transpose2d(t, _st(sd["deconv13.weight"]), None, 2, 0))
    return _st(F.conv2d(t, _st(sd["pred.weight"]), None, 1, 1))


def espcn_forward(sd: State, x: Tensor, upscale_factor: int) -> Tensor:
    """ESPCN.forward, espcn.py:46-51."""
    c = lambda n, t: F.conv2d(t, sd[n + ".weight"], sd[n + ".bias"], 1, sd[n + ".weight"].shape[-1] // 2)
    t = F.relu(c("conv1", x)); t = F.relu(c("conv2", t)); t = F.relu(c("conv3", t))
    return c("conv5", F.pixel_shuffle(c("conv4", t), upscale_factor))


def srcnn_forward(sd: State, x: Tensor) -> Tensor:
    """SRCNN.forward, srcnn.py:38-42 (ReLU after every convolution, the last included)."""
    c = lambda n, t: F.conv2d(t, sd[n + ".weight"], sd[n + ".bias"], 1, sd[n + ".weight"].shape[-1] // 2)
    return F.relu(c("conv3", F.relu(c("conv2", F.relu(c("conv1", x))))))


# ---------------------------------------------------------------------------
# Evaluation metrics  (reference src/metrics.py:10-144)
# ---------------------------------------------------------------------------

def metric_ae(y_pred: Tensor, y_true: Tensor) -> Tensor:
    """AE.__call__, metrics.py:22-33: per-image mean angular error in degrees."""
    dot = torch.sum(y_pred * y_true, dim=1)
    n1 = torch.sqrt(torch.sum(y_pred * y_pred, dim=1))
    n2 = torch.sqrt(torch.sum(y_true * y_true, dim=1))
    return (180 / math.pi * torch.acos(dot / (n1 * n2 + 1e-6))).mean(1).mean(1)


def metric_ssim(y_pred: Tensor, y_true: Tensor, size_average: bool = True, full: bool = False):
    """SSIM.__call__, metrics.py:85-144 (w_size 11, sigma 1.5, valid depth-wise windows, range from the prediction)."""
    max_val = 255 if torch.max(y_pred) > 128 else 1
    min_val = -1 if torch.min(y_pred) < -0.5 else 0
    L = max_val - min_val
    ch = y_pred.shape[1]
    g = torch.Tensor([math.exp(-(x - 11 // 2) ** 2 / float(2 * 1.5 ** 2)) for x in range(11)])
    g = (g / g.sum()).unsqueeze(1)
    window = g.mm(g.t()).float().unsqueeze(0).unsqueeze(0).expand(ch, 1, 11, 11).contiguous()
    conv = lambda z: F.conv2d(z, window, padding=0, groups=ch)
    mu1, mu2 = conv(y_pred), conv(y_true)
    mu1_sq, mu2_sq, mu12 = mu1.pow(2), mu2.pow(2), mu1 * mu2
    s1, s2, s12 = conv(y_pred * y_pred) - mu1_sq, conv(y_true * y_true) - mu2_sq, conv(y_pred * y_true) - mu12
    C1, C2 = (0.01 * L) ** 2, (0.03 * L) ** 2
    v1, v2 = 2.0 * s12 + C2, s1 + s2 + C2
    cs = torch.mean(v1 / v2)
    m = ((2 * mu12 + C1) * v1) / ((mu1_sq + mu2_sq + C1) * v2)
    ret = m.mean() if size_average else m.mean(1).mean(1).mean(1)
    return (ret, cs) if full else ret


def edsr_forward(sd: State, x: Tensor) -> Tensor:
    """EDSR.forward, edsr.py:101-110; ResnetBlock.forward :44-50 (one GroupNorm module applied twice, LeakyReLU 0.2)."""
    c = lambda n, t: F.conv2d(t, sd[n + ".weight"], sd[n + ".bias"], 1, 1)
    t = c("input_conv", x)
    residual = t
    i = 0
    while f"residual_layers.{i}.conv1.weight" in sd:
        pre = f"residual_layers.{i}."
        gn = lambda z: F.group_norm(z, 32, sd[pre + "gn.weight"], sd[pre + "gn.bias"], 1e-5)
        t = gn(c(pre + "conv2", _lrelu(gn(c(pre + "conv1", t))))) + t
        i += 1
    t = c("mid_conv", t) + residual
    j = 0
    while f"upscale_layers.{j}.weight" in sd:
        t = F.conv_transpose2d(t, sd[f"upscale_layers.{j}.weight"], None, 2, 0)
        j += 1
    return c("output_conv", t)


def srdn_forward(sd: State, x: Tensor) -> Tensor:
    """SRDN.forward, srdn.py:67-74 (trunk_conv is defined and never applied)."""
    fea = F.conv2d(x, sd["conv_first.weight"], sd["conv_first.bias"], 1, 1)
    for stack in ("RRDB_encoder", "RRDB_decoder"):
        t, i = fea, 0
        while f"{stack}.{i}.RDB1.conv1.weight" in sd:
            t = rrdb_forward(sd, f"{stack}.{i}.", t)
            i += 1
        fea = fea + t
    return F.conv2d(fea, sd["conv_last.weight"], None, 1, 1)


def rddbnet_keys(nb: int, up: int) -> List[str]:
    """state_dict key order of the reference RDDBNet."""
    keys = ["conv_first.weight", "conv_first.bias"]
    for i in range(nb):
        for j in (1, 2, 3):
            for k in range(1, 6):
                keys += [f"RRDB_trunk.{i}.RDB{j}.conv{k}.weight", f"RRDB_trunk.{i}.RDB{j}.conv{k}.bias"]
    keys += ["trunk_conv.weight", "trunk_conv.bias"]
    for s in range(int(math.log2(up)) if up > 1 else 0):
        keys.append(f"upscale_layers.{2 * s}.weight")
    keys.append("conv_last.weight")
    return keys


def _kaiming_normal_fan_out(shape, gen) -> Tensor:
    # nn.init.kaiming_normal_(mode='fan_out', nonlinearity='relu'), rddb.py:100-102
    fan_out = shape[0] * shape[2] * shape[3]
    return torch.randn(shape, generator=gen) * math.sqrt(2.0 / fan_out)


def _default_uniform(shape, fan_in, gen) -> Tensor:
    # torch default Conv/ConvTranspose reset_parameters: U(-1/sqrt(fan_in), 1/sqrt(fan_in))
    bound = 1.0 / math.sqrt(fan_in)
    return (torch.rand(shape, generator=gen) * 2 - 1) * bound


def rddbnet_state(in_ch: int, ou_ch: int, up: int, nf: int = 64, nb: int = 3, gc: int = 32,
                  seed: int = 0) -> State:
    """Random state with the reference's *distributions* (not its RNG stream):
    Conv2d weights kaiming-normal(fan_out, relu), biases and ConvTranspose2d
    weights torch-default uniform (rddb.py:100-105; SURVEY section 7 hard parts)."""
    g = torch.Generator().manual_seed(seed)
    sd: State = {}

    def conv(name, co, ci, k, bias=True):
        sd[name + ".weight"] = _kaiming_normal_fan_out((co, ci, k, k), g)
        if bias:
            sd[name + ".bias"] = _default_uniform((co,), ci * k * k, g)

    conv("conv_first", nf, in_ch, 3)
    for i in range(nb):
        for j in (1, 2, 3):
            p = f"RRDB_trunk.{i}.RDB{j}."
            for k in range(1, 5):
                conv(p + f"conv{k}", gc, nf + (k - 1) * gc, 3)
            conv(p + "conv5", nf, nf + 4 * gc, 3)
    conv("trunk_conv", nf, nf, 3)
    for s in range(int(math.log2(up)) if up > 1 else 0):
        # ConvTranspose2d weight [in, out, 2, 2]; torch fan_in for it = out*kh*kw
        sd[f"upscale_layers.{2 * s}.weight"] = _default_uniform((nf, nf, 2, 2), nf * 4, g)
    conv("conv_last", ou_ch, nf, 3, bias=False)
    return {k: sd[k] for k in rddbnet_keys(nb, up)}


# ---------------------------------------------------------------------------
# Discriminator: NLayerDiscriminator (reference src/model/model.py:595-639)
# ---------------------------------------------------------------------------
def _d_plan(n_layers: int) -> List[Tuple[int, int, bool, bool]]:
    """(sequential index of conv, stride, has_bias, followed_by_bn) per conv,
    model/model.py:612-634."""
    plan = [(0, 2, True, False)]
    idx = 2
    for _ in range(1, n_layers):
        plan.append((idx, 2, False, True))
        idx += 3
    plan.append((idx, 1, False, True))
    idx += 3
    plan.append((idx, 1, True, False))
    return plan


def nlayer_d_keys(n_layers: int, norm: str = "batch") -> List[str]:
    keys: List[str] = []
    for (i, _s, has_bias, bn) in _d_plan(n_layers):
        keys.append(f"model.{i}.weight")
        if has_bias or (bn and norm == "instance"):
            keys.append(f"model.{i}.bias")
        if bn and norm == "batch":
            keys += [f"model.{i + 1}.{n}" for n in ("weight", "bias", "running_mean", "running_var", "num_batches_tracked")]
    return keys


def nlayer_d_state(input_nc: int, ndf: int = 64, n_layers: int = 3, seed: int = 0, norm: str = "batch") -> State:
    """torch-default init (the reference never re-initialises D).  norm = "instance": norm_layer = nn.InstanceNorm2d
    (model/model.py:607-610: the normalised convolutions get a bias; InstanceNorm2d itself has no parameters or buffers,
    basicModel.py:24-25)."""
    g = torch.Generator().manual_seed(seed)
    chans = [input_nc, ndf]
    for n in range(1, n_layers):
        chans.append(ndf * min(2 ** n, 8))
    chans.append(ndf * min(2 ** n_layers, 8))
    chans.append(1)
    sd: State = {}
    for li, (i, _s, has_bias, bn) in enumerate(_d_plan(n_layers)):
        ci, co = chans[li], chans[li + 1]
        sd[f"model.{i}.weight"] = _default_uniform((co, ci, 4, 4), ci * 16, g)
        if has_bias or (bn and norm == "instance"):
            sd[f"model.{i}.bias"] = _default_uniform((co,), ci * 16, g)
        if bn and norm == "batch":
            sd[f"model.{i + 1}.weight"] = torch.ones(co)
            sd[f"model.{i + 1}.bias"] = torch.zeros(co)
            sd[f"model.{i + 1}.running_mean"] = torch.zeros(co)
            sd[f"model.{i + 1}.running_var"] = torch.ones(co)
            sd[f"model.{i + 1}.num_batches_tracked"] = torch.tensor(0, dtype=torch.long)
    return sd


def nlayer_d_forward(sd: State, x: Tensor, training: bool = True) -> Tensor:
    """NLayerDiscriminator.forward, model/model.py:637-639.  In training mode the
    BatchNorm layers use batch statistics and update running_mean/var/
    num_batches_tracked *in place* in ``sd`` exactly like nn.BatchNorm2d."""
    conv_ids = sorted({int(k.split(".")[1]) for k in sd if k.endswith(".weight") and sd[k].dim() == 4})
    last = conv_ids[-1]
    strides = {}
    # stride 2 for every conv except the last two (model/model.py:612,620,628,634)
    for n, i in enumerate(conv_ids):
        strides[i] = 2 if n < len(conv_ids) - 2 else 1
    h = _st(x)
    for i in conv_ids:
        h = F.conv2d(h, _st(sd[f"model.{i}.weight"]), sd.get(f"model.{i}.bias"), strides[i], 1)
        if i == last:
            h = _st(h)
            break
        if f"model.{i + 1}.running_mean" in sd:
            j = i + 1
            h = _st(h)            # the pre-normalisation tensor is stored (statistics pass + apply pass)
            if training:
                sd[f"model.{j}.num_batches_tracked"] += 1
            h = F.batch_norm(h, sd[f"model.{j}.running_mean"], sd[f"model.{j}.running_var"],
                             sd[f"model.{j}.weight"], sd[f"model.{j}.bias"], training, BN_MOMENTUM, BN_EPS)
            h = _st(_lrelu(h))
        elif i != conv_ids[0] and f"model.{i + 2}.weight" not in sd and f"model.{i + 3}.weight" in sd:
            # norm_layer = InstanceNorm2d: (conv + bias, InstanceNorm2d, LeakyReLU) occupy three Sequential slots, no parameters in
            # the middle one; instance statistics in train and eval mode (track_running_stats=False)
            h = _st(h)
            h = F.instance_norm(h, None, None, None, None, True, BN_MOMENTUM, BN_EPS)
            h = _st(_lrelu(h))
        else:                 # (no BatchNorm: conv + bias + LeakyReLU is ONE native store)
            h = _st(_lrelu(h))
    return h


# ---------------------------------------------------------------------------
# Losses (reference src/losses.py:95-147, src/train.py:67-128)
# ---------------------------------------------------------------------------
def l1_loss(a: Tensor, b: Tensor) -> Tensor:
    """losses.L1Loss.forward, losses.py:103-105: mean |a-b|."""
    return (a - b).abs().mean()


def mse_loss(a: Tensor, b: Tensor) -> Tensor:
    """losses.MSELoss.forward, losses.py:131-133: mean (a-b)^2."""
    return ((a - b) ** 2).mean()


def psnr(a: Tensor, b: Tensor) -> Tensor:
    """losses.PSNRLoss.forward, losses.py:144-147: 10*log10(1/mse)."""
    return 10.0 * torch.log10(1.0 / mse_loss(a, b))


def gan_loss(pred: Tensor, target_is_real: bool, real_label: float = 1.0, fake_label: float = 0.0, gan_mode: str = "lsgan") -> Tensor:
    """GANLoss(gan_mode).__call__, train.py:84-127: 'lsgan' = MSE against an expanded scalar label (:86-87, the mode train.py:186
    builds); 'vanilla' = BCEWithLogitsLoss against it (:88-89); 'wgangp' = -mean for real, +mean for fake (:121-126)."""
    t = real_label if target_is_real else fake_label
    if gan_mode == "lsgan":
        return ((pred - t) ** 2).mean()
    if gan_mode == "vanilla":
        return F.binary_cross_entropy_with_logits(pred, torch.full_like(pred, t))
    if gan_mode == "wgangp":
        return -pred.mean() if target_is_real else pred.mean()
    raise NotImplementedError("gan mode %s not implemented" % gan_mode)


# ---------------------------------------------------------------------------
# In-step preprocessing (reference src/trainCas.py:82-112, src/train.py:243-258,382)
# ---------------------------------------------------------------------------
def rgb_to_gray(x: Tensor) -> Tensor:
    """trainCas.py:85-87: Y = 0.2125 R + 0.7154 G + 0.0721 B, keeps a channel dim."""
    return 0.2125 * x[:, 0:1] + 0.7154 * x[:, 1:2] + 0.0721 * x[:, 2:3]


def bilinear_down(x: Tensor, up: int) -> Tensor:
    """F.interpolate(scale_factor=1/up, mode='bilinear') (align_corners=False, no
    antialias), trainCas.py:89-90.  For integer ``up`` and sizes divisible by it
    this is the mean of the centre 2x2 of each up x up block (up even)."""
    return F.interpolate(x, scale_factor=1.0 / up, mode="bilinear")


def nearest_down(x: Tensor, up: int) -> Tensor:
    """F.interpolate(scale_factor=1/up) default mode (nearest), train.py:243,382."""
    return F.interpolate(x, scale_factor=1.0 / up, mode="nearest")


def cas_forward_sr_inputs(real_b: Tensor, up: int) -> Tuple[Tensor, Tensor]:
    """CasSRC.forwardSR input preparation, trainCas.py:82-90 -> (real_BC, real_BA)."""
    bc = rgb_to_gray(real_b)
    return bc, bilinear_down(bc, up)


def cosine_lr_sequence(lr0: float, epochs: int, t_max: int) -> List[float]:
    """CasSRC.update_lr with 'cosine' (trainCas.py:45-61): a *fresh*
    CosineAnnealingLR is built every epoch and stepped once, which multiplies the
    current lr by (1+cos(pi/T_max))/2 each epoch."""
    f = (1.0 + math.cos(math.pi / t_max)) / 2.0
    out, lr = [], lr0
    for _ in range(epochs):
        lr *= f
        out.append(lr)
    return out


class ImagePoolOracle:
    """ImagePool.query, train.py:36-64 (pool of previously generated images, p=0.5 swap)."""

    def __init__(self, pool_size: int, rng: Optional[random.Random] = None):
        self.pool_size, self.images, self.rng = pool_size, [], rng or random

    def query(self, images: Tensor) -> Tensor:
        if self.pool_size == 0:
            return images
        out = []
        for im in images:
            im = im.detach().unsqueeze(0)
            if len(self.images) < self.pool_size:
                self.images.append(im)
                out.append(im)
            elif self.rng.uniform(0, 1) > 0.5:
                k = self.rng.randint(0, self.pool_size - 1)
                out.append(self.images[k].clone())
                self.images[k] = im
            else:
                out.append(im)
        return torch.cat(out, 0)


# ---------------------------------------------------------------------------
# HR->LR generator G_B ("RDDBNetA").  NO SOURCE IN THE REFERENCE (train.py:11,173
# import a name defined nowhere) -> build-defined, PARITY UNPINNED vs reference.
# Mirror of RDDBNet with one strided 3x3 s2 conv (+LeakyReLU) per /2 stage placed
# *before* the trunk so the trunk runs at LR.
# ---------------------------------------------------------------------------
def rddbneta_forward(sd: State, x: Tensor, down_factor: int) -> Tensor:
    nb = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("RRDB_trunk."))
    fea = F.conv2d(x, sd["conv_first.weight"], sd["conv_first.bias"], 1, 1)
    for s in range(int(math.log2(down_factor)) if down_factor > 1 else 0):
        fea = _lrelu(F.conv2d(fea, sd[f"down_layers.{2 * s}.weight"], sd[f"down_layers.{2 * s}.bias"], 2, 1))
    t = fea
    for i in range(nb):
        t = rrdb_forward(sd, f"RRDB_trunk.{i}.", t)
    t = F.conv2d(t, sd["trunk_conv.weight"], sd["trunk_conv.bias"], 1, 1)
    fea = fea + t
    return F.conv2d(fea, sd["conv_last.weight"], None, 1, 1)


def rddbneta_state(in_ch: int, ou_ch: int, down: int, nf: int = 64, nb: int = 3, gc: int = 32, seed: int = 0) -> State:
    base = rddbnet_state(in_ch, ou_ch, 1, nf, nb, gc, seed)
    g = torch.Generator().manual_seed(seed + 7919)
    sd: State = {}
    for k, v in base.items():
        sd[k] = v
        if k == "conv_first.bias":
            for s in range(int(math.log2(down)) if down > 1 else 0):
                sd[f"down_layers.{2 * s}.weight"] = _kaiming_normal_fan_out((nf, nf, 3, 3), g)
                sd[f"down_layers.{2 * s}.bias"] = _default_uniform((nf,), nf * 9, g)
    return sd


# ---------------------------------------------------------------------------
# Paired G+D training step (BASELINE config 1/2):
#   G-step: fake=G(x); loss_G = lsgan(D(fake),1) + 10*L1(fake,y); Adam(G, 1e-4, b1 .5)
#   D-step: loss_D = .5*(lsgan(D(y),1)+lsgan(D(fake.detach()),0)); Adam(D, 1e-5, b1 .5)
# restating train.py:262-340 (backward_D_basic / backward_G / optimize_parameters)
# for one generator/discriminator pair.
# ---------------------------------------------------------------------------
class PairedStepState:
    def __init__(self, g_sd: State, d_sd: State, up: int, lambda_l1: float = 10.0,
                 lr_g: float = 1e-4, lr_d: float = 1e-5, beta1: float = 0.5):
        self.up, self.lambda_l1 = up, lambda_l1
        self.g = {k: v.clone().requires_grad_(True) for k, v in g_sd.items()}
        self.d = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone())
                  for k, v in d_sd.items()}
        self.g_params = list(self.g.values())
        self.d_params = [v for v in self.d.values() if v.requires_grad]
        self.opt_g = torch.optim.Adam(self.g_params, lr=lr_g, betas=(beta1, 0.999))
        self.opt_d = torch.optim.Adam(self.d_params, lr=lr_d, betas=(beta1, 0.999))


def make_paired_state(in_ch=3, ou_ch=3, up=2, nf=64, nb=1, gc=32, ndf=64, n_layers=3, seed=0) -> PairedStepState:
    return PairedStepState(rddbnet_state(in_ch, ou_ch, up, nf, nb, gc, seed),
                           nlayer_d_state(ou_ch, ndf, n_layers, seed + 1), up)


def paired_step(st: PairedStepState, x: Tensor, y: Tensor) -> Dict[str, float]:
    # ---- G step (train.py:330-333, D frozen) ----
    for p in st.d_params:
        p.requires_grad_(False)
    st.opt_g.zero_grad()
    fake = rddbnet_forward(st.g, x, st.up)
    loss_gan = gan_loss(nlayer_d_forward(st.d, fake, True), True)
    loss_l1 = l1_loss(fake, y)
    loss_g = loss_gan + st.lambda_l1 * loss_l1
    loss_g.backward()
    st.opt_g.step()
    # ---- D step (train.py:336-340, 262-280) ----
    for p in st.d_params:
        p.requires_grad_(True)
    st.opt_d.zero_grad()
    loss_d = 0.5 * (gan_loss(nlayer_d_forward(st.d, y, True), True)
                    + gan_loss(nlayer_d_forward(st.d, fake.detach(), True), False))
    loss_d.backward()
    st.opt_d.step()
    return {"loss_G": float(loss_g), "loss_G_GAN": float(loss_gan), "loss_L1": float(loss_l1), "loss_D": float(loss_d)}


# ---------------------------------------------------------------------------
# Full cycle step (reference src/train.py:228-340, params :344-361).  G_A = RDDBNet
# (LR->HR), G_B = rddbneta (HR->LR, build-defined), D_A on HR, D_B on LR.
# ---------------------------------------------------------------------------
class CycleState:
    def __init__(self, ga: State, gb: State, da: State, db: State, up: int, pool_size: int = 4,
                 lr: float = 1e-4, lr_d: float = 1e-5, beta1: float = 0.5,
                 lambda_a: float = 10.0, lambda_b: float = 10.0, lambda_idt: float = 1.0, seed: int = 0,
                 ga_kind: str = "rddbnet"):
        self.up = up
        self.ga_kind = ga_kind          # "rddbnet" (rddb.py) or "rddbnetb" (model/model.py:394, what train.py:172 constructs)
        self.lam = (lambda_a, lambda_b, lambda_idt)
        req = lambda sd: {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone())
                          for k, v in sd.items()}
        self.ga, self.gb, self.da, self.db = req(ga), req(gb), req(da), req(db)
        gp = list(self.ga.values()) + list(self.gb.values())
        self.dp = [v for v in list(self.da.values()) + list(self.db.values()) if v.requires_grad]
        self.opt_g = torch.optim.Adam(gp, lr=lr, betas=(beta1, 0.999))        # train.py:191
        self.opt_d = torch.optim.Adam(self.dp, lr=lr_d, betas=(beta1, 0.999))  # train.py:192
        rng = random.Random(seed)
        self.pool_a, self.pool_b = ImagePoolOracle(pool_size, rng), ImagePoolOracle(pool_size, rng)


def make_cycle_state(up=2, nf=64, nb=1, gc=32, ndf=64, n_layers=3, seed=0) -> CycleState:
    return CycleState(rddbnet_state(3, 3, up, nf, nb, gc, seed), rddbneta_state(3, 3, up, nf, nb, gc, seed + 1),
                      nlayer_d_state(3, ndf, n_layers, seed + 2), nlayer_d_state(3, ndf, n_layers, seed + 3), up, seed=seed)


def cycle_step(st: CycleState, real_a: Tensor, real_b: Tensor) -> Dict[str, float]:
    la, lb, lidt = st.lam
    GA = (lambda t: rddbnetb_forward(st.ga, t, f"x{st.up}")) if st.ga_kind == "rddbnetb" else (lambda t: rddbnet_forward(st.ga, t, st.up))
    GB = lambda t: rddbneta_forward(st.gb, t, st.up)
    # forward, train.py:228-249 (opt.net == '1' branch: 3-channel both sides, nearest resampling)
    fake_b = GA(real_a); recl_a = GB(fake_b)
    fake_a = GB(real_b); recl_b = GA(fake_a)
    iden_a = GA(nearest_down(real_b, st.up))
    iden_b = GB(F.interpolate(real_a, scale_factor=st.up))
    # G step, train.py:292-333
    for p in st.dp:
        p.requires_grad_(False)
    st.opt_g.zero_grad()
    l_iden_a = l1_loss(iden_a, real_b) * lb / 2 * lidt
    l_iden_b = l1_loss(iden_b, real_a) * la / 2 * lidt
    l_g_a = gan_loss(nlayer_d_forward(st.da, fake_b, True), True)
    l_g_b = gan_loss(nlayer_d_forward(st.db, fake_a, True), True)
    l_cyc_a = l1_loss(recl_a, real_a) * la * 0.5
    l_cyc_b = l1_loss(recl_b, real_b) * lb * 0.5
    loss_g = (l_g_a + l_g_b) + l_cyc_a + l_cyc_b + l_iden_a + l_iden_b
    loss_g.backward()
    st.opt_g.step()
    # D step, train.py:262-290,336-340
    for p in st.dp:
        p.requires_grad_(True)
    st.opt_d.zero_grad()
    fb = st.pool_b.query(fake_b)
    l_d_a = 0.5 * (gan_loss(nlayer_d_forward(st.da, real_b, True), True) + gan_loss(nlayer_d_forward(st.da, fb.detach(), True), False))
    l_d_a.backward()
    fa = st.pool_a.query(fake_a)
    l_d_b = 0.5 * (gan_loss(nlayer_d_forward(st.db, real_a, True), True) + gan_loss(nlayer_d_forward(st.db, fa.detach(), True), False))
    l_d_b.backward()
    st.opt_d.step()
    return {"loss_G": float(loss_g), "loss_D_A": float(l_d_a), "loss_D_B": float(l_d_b),
            "loss_cycle": float(l_cyc_a + l_cyc_b), "loss_iden": float(l_iden_a + l_iden_b), "loss_G_GAN": float(l_g_a + l_g_b)}


# ---------------------------------------------------------------------------
# Host input pipeline: colour conversions of dataset.Basic (reference src/dataset.py:92-159)
# PARITY UNPINNED: the reference delegates to scikit-image 's skimage.color.{rgb2gray, rgb2lab, lab2rgb} (version not pinned by
# the reference: it ships no requirements file), which is not installed in this image, and the reference holds no fixture
# for them.  What follows restates scikit-image's published algorithm in float64 numpy (as the reference's ndarray code
# runs); tests/test_oracle_golden.py anchors it on CIE known answers (white, black, the sRGB primaries) only.
import numpy as _np

_XYZ_FROM_RGB = _np.array([[0.412453, 0.357580, 0.180423], [0.212671, 0.715160, 0.072169], [0.019334, 0.119193, 0.950227]])
_WHITE_D65_2 = _np.array([0.95047, 1.0, 1.08883])


def _rgb2lab_f64(arr_u8):
    a = _np.asarray(arr_u8).astype(_np.float64) / 255.0                       # img_as_float of an 8-bit image
    lin = _np.where(a > 0.04045, ((a + 0.055) / 1.055) ** 2.4, a / 12.92)     # rgb2xyz
    xyz = lin @ _XYZ_FROM_RGB.T / _WHITE_D65_2                                # xyz2lab, illuminant D65, observer 2
    f = _np.where(xyz > 0.008856, _np.cbrt(xyz), 7.787 * xyz + 16.0 / 116.0)
    L = 116.0 * f[..., 1] - 16.0
    return _np.stack([L, 500.0 * (f[..., 0] - f[..., 1]), 200.0 * (f[..., 1] - f[..., 2])], axis=-1)


def arr2gray(arr_u8) -> Tensor:
    """Basic._arr2gray, dataset.py:114-123: rgb2gray(arr) [H,W,3] uint8 -> [1,H,W] float32."""
    g = (_np.asarray(arr_u8).astype(_np.float64) / 255.0) @ _np.array([0.2125, 0.7154, 0.0721])
    return torch.from_numpy(g[None]).float()


def arr2rgb(arr_u8) -> Tensor:
    """Basic._arr2rgb, dataset.py:125-134."""
    return torch.from_numpy((_np.asarray(arr_u8) / (2 ** 8 - 1)).transpose((2, 0, 1))).float()


def arr2lab(arr_u8) -> Tensor:
    """Basic._arr2lab, dataset.py:148-159."""
    lab = _rgb2lab_f64(arr_u8)
    lab[:, :, :1] = lab[:, :, :1] / 100
    lab[:, :, 1:] = (lab[:, :, 1:] + 128) / 255
    return torch.from_numpy(lab.transpose((2, 0, 1))).float()


def arr2ab(arr_u8) -> Tensor:
    """Basic._arr2ab, dataset.py:136-146."""
    ab = (_rgb2lab_f64(arr_u8)[:, :, 1:] + 128) / 255
    return torch.from_numpy(ab.transpose((2, 0, 1))).float()


def lab2img(lab_hw3) -> "_np.ndarray":
    """Basic._lab2img without the whitespace frame, dataset.py:92-104: normalised LAB [H,W,3] -> uint8 RGB [H,W,3]."""
    lab = _np.array(lab_hw3, dtype=_np.float64)
    lab[:, :, :1] = lab[:, :, :1] * 100
    lab[:, :, 1:] = lab[:, :, 1:] * 255 - 128
    fy = (lab[..., 0] + 16.0) / 116.0                                         # lab2xyz
    fx = lab[..., 1] / 500.0 + fy
    fz = _np.maximum(fy - lab[..., 2] / 200.0, 0.0)
    f = _np.stack([fx, fy, fz], axis=-1)
    xyz = _np.where(f > 0.2068966, f ** 3, (f - 16.0 / 116.0) / 7.787) * _WHITE_D65_2
    rgb = xyz @ _np.linalg.inv(_XYZ_FROM_RGB).T                               # xyz2rgb
    rgb = _np.where(rgb > 0.0031308, 1.055 * _np.power(_np.maximum(rgb, 0.0), 1 / 2.4) - 0.055, rgb * 12.92)
    return (_np.clip(rgb, 0, 1) * 255).astype("uint8")
