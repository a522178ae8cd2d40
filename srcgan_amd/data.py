"""Host input pipeline and checkpoint wire format.

* Colour conversions of ``dataset.Basic`` (src/dataset.py:114-159) behind ``G2RGB`` / ``G2LAB.__getitem__`` (:179-199, :234-254):
  8-bit RGB -> the float tensors of a training batch, on the device and for a whole batch at once (the reference converts one
  sample at a time on the host with scikit-image).  Same method names as the reference (``_arr2gray`` -> ``arr2gray`` ...).
* The folder datasets themselves (``Basic`` / ``G2RGB`` / ``G2LAB`` / ``load_dataset``, dataset.py:27-285): list file, PIL
  decode and the optional transform on the host (DataLoader workers), conversion per batch on the device (``DeviceLoader``).
* ``.pth`` naming and round trip: ``{SRModel}_A2C_x{up}_{epoch:04d}.pth`` / ``{CModel}_C2B_x{up}_{epoch:04d}.pth`` written
  every 25 epochs (src/trainCas.py:221-225) and parsed back by the test script to rebuild the two networks
  (src/testCas.py:41-56).  Files hold ``torch.save(net.state_dict())`` with the reference's key names, so they load into
  either implementation.

There is no CPU fallback: the conversions run in libsrcgan_amd.so (csrc/colour.hip).
"""
from __future__ import annotations

import os
from typing import Callable, Dict, Iterator, Optional, Tuple

import torch
import torch.utils.data

from . import _native as N

# dataset.py:24-25 resolves '../dataset/' next to its source file.  Here, in this order: the ``dataset_dir`` argument, this
# module attribute when set, $SRCGAN_DATASET_DIR, ./dataset of the working directory.
DATASET_DIR: Optional[str] = None


def _dataset_dir(arg: Optional[str]) -> str:
    return arg or DATASET_DIR or os.environ.get("SRCGAN_DATASET_DIR") or os.path.join(os.getcwd(), "dataset")

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


class Basic(torch.utils.data.Dataset):
    """``dataset.Basic`` (dataset.py:27-47): ``<dataset_dir>/<root>/<split>.txt`` lists one file name per line, the images
    live in ``<root>/src/<name>`` and ``<root>/tar/<name>``.  The division of labour differs from the reference's: a sample is
    the DECODED 8-bit pair (the PIL decode + the optional ``transform`` stay in the DataLoader workers, on the host), the
    colour conversion is done once per batch on the device by :meth:`to_device` / :class:`DeviceLoader`, which yield the
    ``{"src", "tar", "idx"}`` float batch the reference's ``DataLoader`` yields.  ``root=None`` gives a converter without a
    file list (``G2LAB()(src_u8, tar_u8)``).

    ``transform`` follows the reference's contract (:183-190): called with ``{'src': PIL.Image, 'tar': PIL.Image}``, returns
    the same dict holding arrays ([H,W,3] uint8) -- the reference ships no transform of its own."""
    src_ch, tar_ch, ver = 1, 3, "Basic"
    _tar_mode = "rgb"

    def __init__(self, root: Optional[str] = None, split: str = "all", transform: Optional[Callable] = None,
                 dataset_dir: Optional[str] = None):
        self.root, self.split, self.transform = root, split, transform
        self.datalist = []
        if root is None:
            return
        base = os.path.join(_dataset_dir(dataset_dir), root)
        with open(os.path.join(base, "{}.txt".format(split)), "r") as f:        # a missing list raises, as in the reference
            self.datalist = [line.strip() for line in f.readlines()]
        self.srcpath = os.path.join(base, "src", "%s")
        self.tarpath = os.path.join(base, "tar", "%s")

    def __len__(self) -> int:
        return len(self.datalist)

    @staticmethod
    def _decode(path: str):
        from PIL import Image
        return Image.open(path).convert("RGB")

    def __getitem__(self, idx: int) -> Dict[str, object]:
        """Host half of ``G2RGB.__getitem__`` (dataset.py:179-190): decode, transform; uint8 [H,W,3] tensors + the index."""
        import numpy as np
        name = self.datalist[idx]
        sample = {"src": self._decode(self.srcpath % name), "tar": self._decode(self.tarpath % name)}
        if self.transform:
            sample = self.transform(sample)
        out = {}
        for k in ("src", "tar"):
            a = np.array(sample[k])                                    # a copy: PIL's buffer is read-only
            if a.dtype != np.uint8 or a.ndim != 3 or a.shape[-1] != 3:
                raise ValueError(f"{self.ver}[{idx}]: '{k}' must be an 8-bit [H,W,3] array after the transform, got {a.dtype} {a.shape}")
            out[k] = torch.from_numpy(a)
        out["idx"] = idx
        return out

    def __call__(self, src: torch.Tensor, tar: torch.Tensor) -> Dict[str, torch.Tensor]:
        """Device half (dataset.py:191-194 / :246-249): src -> tensor(L), tar -> tensor(RGB | LAB), whole batch at once."""
        return {"src": arr2gray(src), "tar": _convert(tar, self._tar_mode)}

    def to_device(self, batch: Dict[str, object], device="cuda") -> Dict[str, object]:
        """A collated batch of :meth:`__getitem__` samples (uint8 [B,H,W,3] on the host) -> the reference's batch on the device."""
        out = self(batch["src"].to(device, non_blocking=True), batch["tar"].to(device, non_blocking=True))
        out["idx"] = batch["idx"]
        return out

    @staticmethod
    def _whitespace(img: torch.Tensor, width: int = 5) -> torch.Tensor:
        """dataset.py:58-66: a white frame of ``width`` pixels around a uint8 [H,W,C] image."""
        row, col, ch = img.shape
        out = torch.full((row + 2 * width, col + 2 * width, ch), 255, dtype=torch.uint8, device=img.device)
        out[width:row + width, width:col + width] = img
        return out

    def show(self, idx: int, save_dir: Optional[str] = None) -> str:
        """``G2RGB.show`` / ``G2LAB.show`` (dataset.py:201-215, :256-272): source | target side by side, both framed, written
        to ``<save_dir>/<split>-<idx>.png`` (default ``./example/<root><ver>``).  A source of another size is resized to the
        target's with PIL's bilinear filter (the reference calls cv2.resize, which is not installed here: not pixel-pinned)."""
        import numpy as np
        from PIL import Image
        s = self[idx]
        b = self(s["src"].cuda(), s["tar"].cuda())
        src = (b["src"].expand(3, -1, -1).permute(1, 2, 0) * 255).to(torch.uint8)          # _rgb2img of a 1-channel array
        tar = lab2img(b["tar"]) if self._tar_mode == "lab" else (b["tar"].permute(1, 2, 0) * 255).to(torch.uint8)
        src, tar = self._whitespace(src).cpu().numpy(), self._whitespace(tar).cpu().numpy()
        if src.shape[:2] != tar.shape[:2]:
            src = np.asarray(Image.fromarray(src).resize(tar.shape[:2][::-1], Image.BILINEAR))
        save_dir = save_dir or os.path.join("example", "{}{}".format(self.root, self.ver))
        os.makedirs(save_dir, exist_ok=True)
        path = "{}/{}-{}.png".format(save_dir, self.split, idx)
        Image.fromarray(np.concatenate([src, tar], axis=1)).save(path)
        return path


class G2RGB(Basic):
    """``dataset.G2RGB`` (dataset.py:160-215): src -> tensor(L), tar -> tensor(RGB).  ``src_ch`` / ``tar_ch`` / ``ver`` as the
    reference's attributes (:174-176)."""
    src_ch, tar_ch, ver = 1, 3, "G2RGB"
    _tar_mode = "rgb"


class G2LAB(Basic):
    """``dataset.G2LAB`` (dataset.py:217-272): src -> tensor(L), tar -> tensor(LAB)."""
    src_ch, tar_ch, ver = 1, 3, "G2LAB"
    _tar_mode = "lab"


def load_dataset(root: str, ver: str = "G2RGB", mode: str = "training", dataset_dir: Optional[str] = None):
    """``dataset.load_dataset`` (dataset.py:275-285): the train / val / test splits of one dataset version (``mode`` is unused
    there too)."""
    kinds = {"G2RGB": G2RGB, "G2LAB": G2LAB}
    if ver not in kinds:
        raise KeyError(f"load_dataset: unknown version {ver!r} (known: {sorted(kinds)})")
    return tuple(kinds[ver](root=root, split=s, dataset_dir=dataset_dir) for s in ("train", "val", "test"))


class DeviceLoader:
    """``DataLoader(dataset, batch_size, num_workers=..., shuffle=..., drop_last=...)`` of trainCas.py:176-178 / testCas.py:61-62
    whose batches arrive converted on the device: the workers decode into pinned uint8 batches ([B,H,W,3], 3 bytes per pixel
    over PCIe instead of the 16 the reference's float32 src + tar carry), ``dataset.to_device`` converts each batch in one
    launch per tensor.  Iterating yields ``{"src": [B,1,H,W], "tar": [B,3,H,W], "idx": [B]}``."""

    def __init__(self, dataset: Basic, batch_size: int = 1, device="cuda", **kw):
        kw.setdefault("pin_memory", torch.cuda.is_available())
        self.dataset, self.device = dataset, device
        self.loader = torch.utils.data.DataLoader(dataset, batch_size, **kw)

    def __len__(self) -> int:
        return len(self.loader)

    def __iter__(self) -> Iterator[Dict[str, object]]:
        for batch in self.loader:
            yield self.dataset.to_device(batch, self.device)


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
