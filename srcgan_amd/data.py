"""Host input pipeline and checkpoint wire format.

* Colour conversions of ``dataset.Basic`` (src/dataset.py:114-159) behind ``G2RGB`` / ``G2LAB.__getitem__`` (:179-199, :234-254):
  8-bit RGB -> the float tensors of a training batch, on the device and for a whole batch at once (the reference converts one
  sample at a time on the host with scikit-image).  Same method names as the reference (``_arr2gray`` -> ``arr2gray`` ...).
* ``.pth`` naming and round trip: ``{SRModel}_A2C_x{up}_{epoch:04d}.pth`` / ``{CModel}_C2B_x{up}_{epoch:04d}.pth`` written
  every 25 epochs (src/trainCas.py:221-225) and parsed back by the test script to rebuild the two networks
  (src/testCas.py:41-56).  Files hold ``torch.save(net.state_dict())`` with the reference's key names, so they load into
  either implementation.

There is no CPU fallback: the conversions run in libsrcgan_amd.so (csrc/colour.hip).
"""
from __future__ import annotations

import os
from typing import Dict, Tuple

import torch

from . import _native as N

_MODES = {"gray": (0, 1), "rgb": (1, 3), "lab": (2, 3), "ab": (3, 2)}


def _convert(arr: torch.Tensor, mode: str) -> torch.Tensor:
    """arr: uint8 [H,W,3] or [B,H,W,3] on the device -> float32 [C,H,W] / [B,C,H,W]."""
    N.require_cuda(arr, f"arr2{mode}")
    if arr.dtype != torch.uint8 or arr.shape[-1] != 3 or arr.dim() not in (3, 4):
        raise ValueError(f"arr2{mode}: expected a uint8 [H,W,3] or [B,H,W,3] array, got {arr.dtype} {tuple(arr.shape)}")
    single = arr.dim() == 3
    a = (arr.unsqueeze(0) if single else arr).contiguous()
    B, H, W, _ = a.shape
    m, c = _MODES[mode]
    out = torch.empty(B, c, H, W, dtype=torch.float32, device=a.device)
    N.check(N.lib().srcgan_u8rgb_to_planes(a.data_ptr(), out.data_ptr(), B, H * W, m, N.stream_ptr(a.device)), "srcgan_u8rgb_to_planes")
    return out[0] if single else out


def arr2gray(arr: torch.Tensor) -> torch.Tensor:
    """``Basic._arr2gray`` (dataset.py:114-123): skimage ``rgb2gray`` of an 8-bit image -> tensor(L) in [0,1]."""
    return _convert(arr, "gray")


def arr2rgb(arr: torch.Tensor) -> torch.Tensor:
    """``Basic._arr2rgb`` (dataset.py:125-134): ``arr / 255`` -> tensor(RGB)."""
    return _convert(arr, "rgb")


def arr2lab(arr: torch.Tensor) -> torch.Tensor:
    """``Basic._arr2lab`` (dataset.py:148-159): ``rgb2lab``; L / 100, (a, b) + 128 over 255 -> tensor(LAB) in [0,1]."""
    return _convert(arr, "lab")


def arr2ab(arr: torch.Tensor) -> torch.Tensor:
    """``Basic._arr2ab`` (dataset.py:136-146): the two chroma planes of ``arr2lab``."""
    return _convert(arr, "ab")


def lab2img(lab: torch.Tensor) -> torch.Tensor:
    """``Basic._lab2img`` without the whitespace frame (dataset.py:92-104): normalised LAB [3,H,W] / [B,3,H,W] float32 ->
    uint8 RGB [H,W,3] / [B,H,W,3] (``lab2rgb`` * 255, truncated)."""
    N.require_cuda(lab, "lab2img")
    if lab.dtype != torch.float32 or lab.dim() not in (3, 4) or lab.shape[-3] != 3:
        raise ValueError(f"lab2img: expected float32 [3,H,W] or [B,3,H,W], got {lab.dtype} {tuple(lab.shape)}")
    single = lab.dim() == 3
    a = (lab.unsqueeze(0) if single else lab).contiguous()
    B, _, H, W = a.shape
    out = torch.empty(B, H, W, 3, dtype=torch.uint8, device=a.device)
    N.check(N.lib().srcgan_lab_planes_to_u8rgb(a.data_ptr(), out.data_ptr(), B, H * W, N.stream_ptr(a.device)), "srcgan_lab_planes_to_u8rgb")
    return out[0] if single else out


def ab2img(l: torch.Tensor, ab: torch.Tensor) -> torch.Tensor:
    """``Basic._ab2img`` (dataset.py:106-112): L plane + chroma planes -> uint8 RGB."""
    return lab2img(torch.cat([l, ab], dim=-3))


class G2RGB:
    """Batch form of ``dataset.G2RGB.__getitem__`` (dataset.py:179-199) for arrays already decoded to 8-bit RGB:
    src -> tensor(L), tar -> tensor(RGB).  ``src_ch`` / ``tar_ch`` / ``ver`` as the reference's attributes (:174-176)."""
    src_ch, tar_ch, ver = 1, 3, "G2RGB"

    def __call__(self, src: torch.Tensor, tar: torch.Tensor) -> Dict[str, torch.Tensor]:
        return {"src": arr2gray(src), "tar": arr2rgb(tar)}


class G2LAB:
    """Batch form of ``dataset.G2LAB.__getitem__`` (dataset.py:234-254): src -> tensor(L), tar -> tensor(LAB)."""
    src_ch, tar_ch, ver = 1, 3, "G2LAB"

    def __call__(self, src: torch.Tensor, tar: torch.Tensor) -> Dict[str, torch.Tensor]:
        return {"src": arr2gray(src), "tar": arr2lab(tar)}


# ---------------------------------------------------------------------------------------------------------------- checkpoints
def checkpoint_name(model_name: str, role: str, up: int, epoch: int) -> str:
    """File name of trainCas.py:222-223: role 'A2C' (super-resolution net) or 'C2B' (colouriser)."""
    if role not in ("A2C", "C2B"):
        raise ValueError(f"checkpoint_name: role must be 'A2C' or 'C2B', got {role!r}")
    return "%s_%s_x%d_%04d.pth" % (model_name, role, up, epoch)


def parse_checkpoint_name(path: str) -> Tuple[str, str, int, int]:
    """(model name, role, up, epoch) from a checkpoint path, the way testCas.py:41-42,52 reads it: basename up to '.pth', split
    on '_'; the upscale factor is the single digit after 'x' (``int(checkA[2][1])``)."""
    parts = os.path.basename(path).split(".pth")[0].split("_")
    if len(parts) < 4 or len(parts[2]) < 2 or parts[2][0] != "x":
        raise ValueError(f"parse_checkpoint_name: {path!r} is not '<Model>_<role>_x<up>_<epoch>.pth'")
    return parts[0], parts[1], int(parts[2][1]), int(parts[3])


def save_checkpoints(model, opt, epoch: int, root: str = "./checkpoints") -> Tuple[str, str]:
    """The epoch-end save of trainCas.py:221-225 for a cascade harness (``netG_A2C`` / ``netG_C2B``): two state_dict files."""
    os.makedirs(root, exist_ok=True)
    pa = os.path.join(root, checkpoint_name(opt.SRModel, "A2C", opt.up, epoch))
    pb = os.path.join(root, checkpoint_name(opt.CModel, "C2B", opt.up, epoch))
    torch.save(model.netG_A2C.state_dict(), pa)
    torch.save(model.netG_C2B.state_dict(), pb)
    return pa, pb


def load_cascade(netGA: str, netGB: str, device="cuda", registry=None):
    """testCas.py:52-58: rebuild ``eval(checkA[0])(1, 1, up)`` and ``eval(checkB[0])(1, 3)`` from the file names, load both
    state_dicts (tensors only) and switch to eval mode.  ``registry`` defaults to train.MODEL_REGISTRY."""
    if registry is None:
        from .train import MODEL_REGISTRY as registry
    name_a, _, up, _ = parse_checkpoint_name(netGA)
    name_b = parse_checkpoint_name(netGB)[0]
    for n in (name_a, name_b):
        if n not in registry:
            raise KeyError(f"load_cascade: unknown model {n!r} (known: {sorted(registry)})")
    net_a = registry[name_a](1, 1, up).to(device)
    net_b = registry[name_b](1, 3).to(device)
    net_a.load_state_dict(torch.load(netGA, map_location=device, weights_only=True))
    net_b.load_state_dict(torch.load(netGB, map_location=device, weights_only=True))
    return net_a.eval(), net_b.eval()
