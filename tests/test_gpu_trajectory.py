"""Do the perf dtypes train the same network?  The reference is fp32 end to end (trainCas.py:156-164: no AMP); the bench
configurations run bf16 / fp16 storage.  Per-step gradient errors (tests/test_gpu_modules.py) do not answer that by themselves, so
here whole harnesses train for 30 optimiser steps from one seed on STRUCTURED synthetic images (smooth colour fields with edges:
something a super-resolution network can actually learn, unlike torch.rand noise) in fp32, bf16 and fp16, and the loss
trajectories and the final PSNR must agree within the bounds stated at each assert.  The figures are printed (pytest -s)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

STEPS = 30


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


def structured_images(B, C, H, W, seed):
    """[B,C,H,W] in [0,1]: a few low-frequency sinusoids per channel plus axis-aligned rectangles (sharp edges)."""
    g = torch.Generator().manual_seed(seed)
    yy, xx = torch.meshgrid(torch.linspace(0, 1, H), torch.linspace(0, 1, W), indexing="ij")
    img = torch.zeros(B, C, H, W)
    for b in range(B):
        for c in range(C):
            for _ in range(4):
                fx, fy = (torch.rand(2, generator=g) * 3.0 + 0.5).tolist()
                ph, amp = (torch.rand(2, generator=g)).tolist()
                img[b, c] += (0.3 + 0.7 * amp) * torch.sin(2 * math.pi * (fx * xx + fy * yy) + 2 * math.pi * ph)
        for _ in range(3):
            y0, x0 = (torch.rand(2, generator=g) * 0.7).tolist()
            h, w = (torch.rand(2, generator=g) * 0.25 + 0.05).tolist()
            col = torch.rand(C, generator=g) * 2 - 1
            m = ((yy >= y0) & (yy < y0 + h) & (xx >= x0) & (xx < x0 + w)).float()
            img[b] += col.view(C, 1, 1) * m * 2.0
    lo, hi = img.amin(dim=(1, 2, 3), keepdim=True), img.amax(dim=(1, 2, 3), keepdim=True)
    return (img - lo) / (hi - lo)


def _psnr(a, b):
    return float(10.0 * torch.log10(1.0 / torch.mean((a.float() - b.float()) ** 2)))


def _dev(a, b):
    """largest deviation of curve a from curve b, relative to b's largest value"""
    a, b = torch.tensor(a), torch.tensor(b)
    return float((a - b).abs().max() / b.abs().max())


def _paired(dt):
    from srcgan_amd import ops
    from srcgan_amd.train import PairedSRGAN
    torch.manual_seed(0)
    m = PairedSRGAN(3, 3, 4, nf=64, nb=2, gc=32, ndf=32, n_layers=3, dtype=dt, device="cuda")
    y = structured_images(4, 3, 64, 64, seed=7).cuda()
    x = ops.bilinear_down(y, 4)
    curves = {"L1": [], "G_GAN": [], "D": []}
    for _ in range(STEPS):
        m.optimize_parameters(x, y)
        curves["L1"].append(m.loss_L1.detach())
        curves["G_GAN"].append(m.loss_G_GAN.detach())
        curves["D"].append(m.loss_D.detach())
    m.netG.eval()
    with torch.no_grad():
        out = m.netG(x)
    torch.cuda.synchronize()
    return {k: [float(v) for v in vs] for k, vs in curves.items()}, _psnr(out, y)


def test_paired_srgan_trajectories_agree_across_dtypes():
    """PairedSRGAN(nb=2, nf=64, x4) + 3-layer PatchGAN, 30 steps: L1 / GAN / discriminator loss curves and the final PSNR."""
    ref, p32 = _paired("fp32")
    assert ref["L1"][-1] < 0.8 * ref["L1"][0], ("the network must actually learn on this data", ref["L1"][0], ref["L1"][-1])
    # measured (round 3): bf16 L1 3.1 % / GAN 3.4 % / D 2.8 %, PSNR 15.04 -> 14.95 dB; fp16 3.7 / 4.3 / 4.0 %, 14.99 dB: both finish on the
    # fp32 curve (L1 0.1415 vs 0.1411 / 0.1405); the mid-curve deviation is the sensitivity of a 30-step GAN trajectory, not precision
    # (fp16, with 3 more mantissa bits, deviates no less than bf16).  How sensitive the END POINT is, seen on later builds: an fp16
    # epilogue that rounds a few elements per 10^5 differently by one ulp (bf16 bit-identical) ended at L1 0.1507 / 14.73 dB instead of
    # 0.1405 / 14.99 dB; BatchNorm statistics computed in one pass instead of two (a 1e-7 relative change) moved the fp32 run itself
    # from 15.041 to 15.029 dB and the bf16 run to 14.52 dB -- while its curves came CLOSER to fp32 (1.0 / 1.4 / 1.4 %).  The final
    # PSNR of one seed is therefore gated at 1 dB only; the curves carry the comparison.  Bounds on them = 2 x the largest measured.
    for dt, bound_l1, bound_gan, bound_db in (("bf16", 0.06, 0.10, 1.0), ("fp16", 0.07, 0.10, 1.0)):
        cur, p = _paired(dt)
        d = {k: _dev(cur[k], ref[k]) for k in ref}
        print(f"paired {dt}: curve deviation from fp32 (relative to the curve's maximum) L1 {d['L1']:.4f} G_GAN {d['G_GAN']:.4f} D {d['D']:.4f}; "
              f"L1 first/last fp32 {ref['L1'][0]:.4f}/{ref['L1'][-1]:.4f} {dt} {cur['L1'][0]:.4f}/{cur['L1'][-1]:.4f}; final PSNR fp32 {p32:.3f} dB, {dt} {p:.3f} dB")
        assert d["L1"] < bound_l1 and d["G_GAN"] < bound_gan and d["D"] < bound_gan, (dt, d)
        assert abs(p - p32) < bound_db, (dt, p, p32)


def _cas(dt):
    from srcgan_amd import _native as N
    from srcgan_amd.train import CasSRCConstLAB, CasParams
    prev = N.dtype_name(None)
    N.set_default_dtype(dt)
    try:
        torch.manual_seed(0)
        m = CasSRCConstLAB(CasParams("cuda", SRModel="SRDN", CModel="ResDeconv", up=4))
    finally:
        N.set_default_dtype(prev)
    lab = structured_images(4, 3, 64, 64, seed=9).cuda()             # normalised LAB target: L, a, b in [0, 1]
    gray = structured_images(4, 1, 64, 64, seed=10).cuda()           # the "satellite" side
    for _ in range(STEPS):
        m.optimize_parameters(gray, lab)
    torch.cuda.synchronize()
    return ({"SR": [float(v) for v in m.loss_sr], "C": [float(v) for v in m.loss_c]},
            float(m.psnr_sr[-1]), float(m.psnr_c[-1]))


def test_cascade_const_lab_trajectories_agree_across_dtypes():
    """BASELINE configs[3]'s harness (trainCasConstLAB: SRDN on the blurred L channel + ResDeconv L -> ab), 30 steps.  The
    colouriser is the network whose bf16 weight gradients are 30-55 % from fp32 per step (GroupNorm backward cancellation,
    DESIGN.md section 3.3): this is the test that says whether that matters for training."""
    ref, sr32, c32 = _cas("fp32")
    assert ref["C"][-1] < 0.8 * ref["C"][0], ("the colouriser must learn", ref["C"][0], ref["C"][-1])
    # measured (round 3), bf16: colouriser curve 0.5 % from fp32 (loss_C 1.755 -> 0.2161 fp32 / 0.2178 bf16, PSNR 10.97 / 10.89 dB) -- the
    # 30-55 % per-step gradient errors of its GroupNorm layers do not move the trajectory; SR curve 3.3 %.  The SR branch's PSNR is
    # printed, not gated: an untrained SRDN's output is orders of magnitude off (PSNR < 0 dB) after 30 steps at lr 1e-4, and the dB
    # figure of such an output is noise (-9.4 vs -11.1 dB).  Bounds = 2-3 x measured.
    for dt, bound_sr, bound_c, bound_db in (("bf16", 0.08, 0.03, 0.3), ("fp16", 0.08, 0.03, 0.3)):
        cur, sr, c = _cas(dt)
        d = {k: _dev(cur[k], ref[k]) for k in ref}
        print(f"cascade-const LAB {dt}: curve deviation from fp32 SR {d['SR']:.4f} C {d['C']:.4f}; loss_C first/last fp32 {ref['C'][0]:.4f}/{ref['C'][-1]:.4f} "
              f"{dt} {cur['C'][0]:.4f}/{cur['C'][-1]:.4f}; loss_SR last fp32 {ref['SR'][-1]:.4f} {dt} {cur['SR'][-1]:.4f}; "
              f"final PSNR SR fp32 {sr32:.3f} {dt} {sr:.3f} dB, C fp32 {c32:.3f} {dt} {c:.3f} dB")
        assert d["SR"] < bound_sr and d["C"] < bound_c, (dt, d)
        assert abs(c - c32) < bound_db, (dt, c, c32)
