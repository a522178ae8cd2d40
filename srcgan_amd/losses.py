"""Loss modules of the hot path with the reference's class names and call signatures.

  L1Loss / MSELoss / PSNRLoss  <- reference src/losses.py:95-105,123-133,136-147
  GANLoss                      <- reference src/train.py:67-128 (only 'lsgan' is ever constructed, :186)

Each forward is one native two-stage reduction (wavefront shuffles + fixed-order final sum,
elementwise.hip) returning a 0-dim device tensor; backward is one fused elementwise kernel
(sign(a-b)/N or 2(a-b)/N times the upstream gradient read from device memory -> no host sync).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _native as N

__all__ = ["L1Loss", "MSELoss", "PSNRLoss", "GANLoss"]

_K_L1, _K_MSE, _K_LABEL, _K_BCE, _K_SIGNED = 0, 1, 2, 3, 4       # srcgan_loss_fwd kinds; >= 2: scalar label instead of a target tensor


def _as_f32(t: torch.Tensor, what: str) -> torch.Tensor:
    N.require_cuda(t, what)
    return t.detach().contiguous().float()


class _MeanLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, kind, label, a, b):
        lib = N.lib()
        a32 = _as_f32(a, "loss input")
        b32 = None
        if kind < _K_LABEL:
            if b.shape != a.shape:
                raise ValueError(f"loss: shape mismatch {tuple(a.shape)} vs {tuple(b.shape)}")
            b32 = _as_f32(b, "loss target")
        out = torch.empty((), dtype=torch.float32, device=a.device)
        scratch = torch.empty(lib.srcgan_loss_scratch_floats(), dtype=torch.float32, device=a.device)
        N.check(lib.srcgan_loss_fwd(kind, a32.data_ptr(), None if b32 is None else b32.data_ptr(), float(label),
                                    a32.numel(), out.data_ptr(), scratch.data_ptr(), N.stream_ptr(a.device)), "srcgan_loss_fwd")
        ctx.kind, ctx.label = kind, float(label)
        ctx.save_for_backward(a32, b32 if b32 is not None else a32.new_empty(0))
        return out

    @staticmethod
    def backward(ctx, gout):
        lib = N.lib()
        a32, b32 = ctx.saved_tensors
        has_b = ctx.kind < _K_LABEL
        gout = gout.contiguous().float()
        da = db = None
        st = N.stream_ptr(a32.device)
        if ctx.needs_input_grad[2]:
            da = torch.empty_like(a32)
            N.check(lib.srcgan_loss_bwd(ctx.kind, a32.data_ptr(), b32.data_ptr() if has_b else None, ctx.label, a32.numel(),
                                        gout.data_ptr(), 1.0, da.data_ptr(), st), "srcgan_loss_bwd")
        if has_b and ctx.needs_input_grad[3]:
            db = torch.empty_like(b32)
            N.check(lib.srcgan_loss_bwd(ctx.kind, a32.data_ptr(), b32.data_ptr(), ctx.label, a32.numel(),
                                        gout.data_ptr(), -1.0, db.data_ptr(), st), "srcgan_loss_bwd")
        return None, None, da, db


class L1Loss(nn.Module):
    """mean |output - target| (losses.py:95-105)."""

    def __repr__(self):
        return "L1"

    def forward(self, output, target):
        return _MeanLossFn.apply(_K_L1, 0.0, output, target)


class MSELoss(nn.Module):
    """mean (output - target)^2 (losses.py:123-133)."""

    def __repr__(self):
        return "MSE"

    def forward(self, output, target):
        return _MeanLossFn.apply(_K_MSE, 0.0, output, target)


class PSNRLoss(nn.Module):
    """10*log10(1/mse), peak 1.0 (losses.py:136-147).  Validation metric: no gradient."""

    def __repr__(self):
        return "PSNR"

    def forward(self, output, target):
        lib = N.lib()
        mse = _MeanLossFn.apply(_K_MSE, 0.0, output.detach(), target.detach())
        out = torch.empty_like(mse)
        N.check(lib.srcgan_psnr_from_mse(mse.data_ptr(), out.data_ptr(), N.stream_ptr(mse.device)), "srcgan_psnr_from_mse")
        return out


class GANLoss(nn.Module):
    """GAN objective with the reference's interface (train.py:67-128): 'lsgan' = MSE against the scalar label (the only mode a
    reference script constructs, train.py:186), 'vanilla' = BCE-with-logits against it, 'wgangp' = -mean(prediction) for real,
    +mean for fake.  The label is folded into the reduction kernel as an immediate instead of ``expand_as``.  'DSSIM' (the
    reference wires ``losses.DSSIMLoss`` there) is outside the native path and refused."""

    _KINDS = {"lsgan": _K_LABEL, "vanilla": _K_BCE, "wgangp": _K_SIGNED}

    def __init__(self, gan_mode, device=None, target_real_label=1.0, target_fake_label=0.0):
        super().__init__()
        self.register_buffer("real_label", torch.tensor(target_real_label, device=device))
        self.register_buffer("fake_label", torch.tensor(target_fake_label, device=device))
        self._real, self._fake = float(target_real_label), float(target_fake_label)
        self.gan_mode = gan_mode
        if gan_mode not in self._KINDS:
            if gan_mode == "DSSIM":
                raise NotImplementedError("gan mode DSSIM is outside the native hot path (no reference script selects it)")
            raise NotImplementedError("gan mode %s not implemented" % gan_mode)

    def get_target_tensor(self, prediction, target_is_real):
        return (self.real_label if target_is_real else self.fake_label).expand_as(prediction)

    def forward(self, prediction, target_is_real):
        kind = self._KINDS[self.gan_mode]
        if kind == _K_SIGNED:
            return _MeanLossFn.apply(kind, -1.0 if target_is_real else 1.0, prediction, None)
        return _MeanLossFn.apply(kind, self._real if target_is_real else self._fake, prediction, None)
