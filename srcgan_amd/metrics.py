"""Evaluation metrics on device -- drop-in for reference ``src/metrics.py`` (AE, MSE, PSNR, SSIM: same constructors, ``repr`` and
``__call__(y_pred, y_true)`` results) and the test loop of ``src/testCas.py:65-90`` (``evaluate_cascade``).
Inputs are [B,C,H,W] f32 CUDA tensors (what the networks return); results are 0-dim / [B] CUDA tensors, so a test loop only
synchronises where it calls ``.item()`` like the reference does.  No CPU fallback."""
import ctypes as C
from typing import Dict, Iterable, List, Optional

import torch

from . import _native as N
from . import ops
from .losses import MSELoss, PSNRLoss

__all__ = ["AE", "MSE", "PSNR", "SSIM", "evaluate_cascade"]


def _prep(y_pred, y_true, who):
    N.require_cuda(y_pred, who)
    N.require_cuda(y_true, who)
    if y_pred.shape != y_true.shape or y_pred.dim() != 4:
        raise ValueError(f"{who}: expected two [B,C,H,W] tensors of one shape, got {tuple(y_pred.shape)} and {tuple(y_true.shape)}")
    return y_pred.detach().contiguous().float(), y_true.detach().contiguous().float()


class AE(object):
    """average Angular Error in degrees per image (metrics.py:12-33): returns a [B] tensor."""

    def __init__(self, des="average Angular Error"):
        self.des = des

    def __repr__(self):
        return "AE"

    def __call__(self, y_pred, y_true):
        p, t = _prep(y_pred, y_true, "AE")
        lib = N.lib()
        B, Cc, H, W = p.shape
        out = torch.empty(B, dtype=torch.float32, device=p.device)
        scr = torch.empty(lib.srcgan_metric_scratch_floats(B, Cc, H, W), dtype=torch.float32, device=p.device)
        N.check(lib.srcgan_metric_ae(p.data_ptr(), t.data_ptr(), B, Cc, H, W, out.data_ptr(), scr.data_ptr(), N.stream_ptr(p.device)), "srcgan_metric_ae")
        return out


class MSE(object):
    def __init__(self, des="Mean Square Error"):
        self.des = des
        self._f = MSELoss()

    def __repr__(self):
        return "MSE"

    def __call__(self, y_pred, y_true, dim=1):
        p, t = _prep(y_pred, y_true, "MSE")
        return self._f(p, t)


class PSNR(object):
    def __init__(self, des="Peak Signal to Noise Ratio"):
        self.des = des
        self._f = PSNRLoss()

    def __repr__(self):
        return "PSNR"

    def __call__(self, y_pred, y_true, dim=1):
        p, t = _prep(y_pred, y_true, "PSNR")
        return self._f(p, t)


class SSIM(object):
    """structural similarity index (metrics.py:65-144): 11x11 gaussian windows, dynamic range chosen from the prediction."""

    def __init__(self, des="structural similarity index"):
        self.des = des

    def __repr__(self):
        return "SSIM"

    def __call__(self, y_pred, y_true, w_size=11, size_average=True, full=False):
        if w_size != 11:
            raise NotImplementedError("native SSIM implements the reference default w_size=11")
        p, t = _prep(y_pred, y_true, "SSIM")
        lib = N.lib()
        B, Cc, H, W = p.shape
        out = torch.empty(B, 2, dtype=torch.float32, device=p.device)
        scr = torch.empty(lib.srcgan_metric_scratch_floats(B, Cc, H, W), dtype=torch.float32, device=p.device)
        N.check(lib.srcgan_metric_ssim(p.data_ptr(), t.data_ptr(), B, Cc, H, W, out.data_ptr(), scr.data_ptr(), N.stream_ptr(p.device)), "srcgan_metric_ssim")
        ret = out[:, 0].mean() if size_average else out[:, 0]
        if full:
            return ret, out[:, 1].mean()
        return ret


@torch.no_grad()
def evaluate_cascade(netG_A2C: torch.nn.Module, netG_C2B: torch.nn.Module, batches: Iterable[Dict[str, torch.Tensor]], up: int,
                     evaluators: Optional[List] = None, device="cuda"):
    """The inference / scoring loop of reference src/testCas.py:65-90 without its file I/O: for every sample
    ``{'src': realA [B,1,H,W], 'tar': realB [B,3,H,W]}`` gray the target, nearest-downsample both by ``up``, run SR then the
    colouriser (both in eval mode) and score fake_BB against realB.  Returns ({metric name: mean over samples}, last outputs)."""
    evaluators = evaluators or [MSE(), PSNR(), AE(), SSIM()]
    netG_A2C.eval(); netG_C2B.eval()
    performs = [[] for _ in evaluators]
    fake_AB = fake_BB = None
    for sample in batches:
        realA, realB = sample["src"].to(device), sample["tar"].to(device)
        realBC = ops.rgb_to_gray(realB)
        realBA = ops.nearest_resize(realBC, 1.0 / up)
        realAA = ops.nearest_resize(realA, 1.0 / up)
        fake_AB = netG_C2B(netG_A2C(realAA))
        fake_BB = netG_C2B(netG_A2C(realBA))
        for i, ev in enumerate(evaluators):
            performs[i].append(ev(fake_BB.detach(), realB.detach()).item())     # testCas.py:82 (AE: batch of one image)
    return {repr(e): sum(p) / len(p) for e, p in zip(evaluators, performs)}, (fake_AB, fake_BB)
