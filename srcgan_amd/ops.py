"""Thin tensor-level wrappers over the op-level C ABI (used by the harness for preprocessing and by
the parity tests to drive single kernels).  Tensors in, tensors out; every call goes to the native
library -- no eager fallback."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch

from . import _native as N

TORCH_DT = {N.F32: torch.float32, N.BF16: torch.bfloat16, N.F16: torch.float16}


def to_nhwc(x: torch.Tensor, cs: Optional[int] = None, dtype=None) -> torch.Tensor:
    """NCHW f32 -> NHWC [B,H,W,cs] in the compute dtype, channels [C,cs) zero."""
    N.require_cuda(x, "to_nhwc")
    dt = N.dtype_id(dtype)
    x = x.contiguous().float()
    B, Cc, H, W = x.shape
    cs = cs or Cc
    out = torch.empty(B, H, W, cs, dtype=TORCH_DT[dt], device=x.device)
    N.check(N.lib().srcgan_nchw_f32_to_nhwc(x.data_ptr(), out.data_ptr(), B, Cc, H, W, cs, dt, N.stream_ptr(x.device)), "to_nhwc")
    return out


def to_nchw(x: torch.Tensor, Cc: Optional[int] = None, coff: int = 0) -> torch.Tensor:
    """NHWC (compute dtype) -> NCHW f32, channels [coff, coff+C)."""
    N.require_cuda(x, "to_nchw")
    dt = N.dtype_id(x.dtype)
    B, H, W, cs = x.shape
    Cc = Cc or cs - coff
    out = torch.empty(B, Cc, H, W, dtype=torch.float32, device=x.device)
    N.check(N.lib().srcgan_nhwc_to_nchw_f32(x.data_ptr(), out.data_ptr(), B, Cc, H, W, cs, coff, dt, N.stream_ptr(x.device)), "to_nchw")
    return out


def make_blocked(x_nhwc: torch.Tensor):
    """NHWC [B,H,W,C] -> blocked [planes,B,H,W,KCE] (KCE = 64 bytes of channels), plus the plane stride in bytes."""
    B, H, W, Cc = x_nhwc.shape
    kce = 64 // x_nhwc.element_size()
    npl = (Cc + kce - 1) // kce
    out = torch.zeros(npl, B, H, W, kce, dtype=x_nhwc.dtype, device=x_nhwc.device)
    for pl in range(npl):
        n = min(kce, Cc - pl * kce)
        out[pl, ..., :n] = x_nhwc[..., pl * kce:pl * kce + n]
    return out, B * H * W * 64


def from_blocked(xb: torch.Tensor, Cc: int) -> torch.Tensor:
    npl, B, H, W, kce = xb.shape
    return xb.permute(1, 2, 3, 0, 4).reshape(B, H, W, npl * kce)[..., :Cc].contiguous()


def pack_weight(w: torch.Tensor, rows: int, kdim: int, tys: int, txs: int, sr: int, sk: int, sty: int, stx: int,
                off: int = 0, dtype=None) -> torch.Tensor:
    N.require_cuda(w, "pack_weight")
    dt = N.dtype_id(dtype)
    lib = N.lib()
    nbytes = lib.srcgan_packed_weight_bytes(rows, kdim, tys * txs, dt)
    out = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
    N.check(lib.srcgan_pack_weight(w.contiguous().data_ptr(), out.data_ptr(), rows, kdim, tys, txs, sr, sk, sty, stx, off, dt,
                                   N.stream_ptr(w.device)), "pack_weight")
    return out


def pack_conv2d_fwd(w: torch.Tensor, dtype=None) -> torch.Tensor:
    co, ci, kh, kw = w.shape
    return pack_weight(w, co, ci, kh, kw, ci * kh * kw, kh * kw, kw, 1, 0, dtype)


def pack_conv2d_dgrad_s1(w: torch.Tensor, dtype=None) -> torch.Tensor:
    co, ci, kh, kw = w.shape
    return pack_weight(w, ci, co, kh, kw, kh * kw, ci * kh * kw, -kw, -1, kh * kw - 1, dtype)


def conv_igemm(x: torch.Tensor, wp: torch.Tensor, y: torch.Tensor, *, kh: int, kw: int, stride: int = 1, Cin: Optional[int] = None,
               x_coff: int = 0, Cout: int, y_coff: int = 0, OH: Optional[int] = None, OW: Optional[int] = None,
               pad: Tuple[int, int] = (0, 0), bias: Optional[torch.Tensor] = None, alpha: float = 1.0, act: bool = False,
               slope: float = 0.2, r1: Optional[torch.Tensor] = None, r1_coff: int = 0, r1_cend: int = 0, beta1: float = 0.0,
               r2: Optional[torch.Tensor] = None, r2_coff: int = 0, r2_cend: int = 0, beta2: float = 0.0,
               mz: Optional[torch.Tensor] = None, mz_coff: int = 0, mz_c0: int = 0, mslope: float = 0.2,
               os: int = 1, oa: int = 0, ob: int = 0, x_plane: int = 0, y_plane: int = 0, r1_plane: int = 0, r2_plane: int = 0,
               mz_plane: int = 0, shape: Optional[Tuple[int, int, int]] = None, sign_out: Optional[torch.Tensor] = None,
               sign_in: Optional[torch.Tensor] = None, npar: int = 0, wpar_stride: int = 0) -> torch.Tensor:
    """x, y, r1, r2, mz: NHWC tensors [B,H,W,cs] of the compute dtype.  Writes into y (returned).
    Blocked-layout tensors ([planes,B,H,W,KCE], see make_blocked) pass *_plane = plane stride in bytes and shape=(B,H,W).
    sign_out / sign_in: int32 [B,OH,OW] LeakyReLU sign masks (bit c = channel c), see srcgan_conv_desc in include/srcgan_amd.h."""
    N.require_cuda(x, "conv_igemm")
    d = N.ConvDesc()
    if shape is not None:
        (B, H, W), xcs = shape, x.shape[-1]
    else:
        B, H, W, xcs = x.shape
    d.x, d.wp, d.y = x.data_ptr(), wp.data_ptr(), y.data_ptr()
    d.bias = bias.data_ptr() if bias is not None else None
    d.dtype = N.dtype_id(x.dtype)
    d.kh, d.kw, d.stride = kh, kw, stride
    d.B, d.H, d.W, d.Cin, d.x_cs, d.x_coff = B, H, W, Cin or (xcs - x_coff), xcs, x_coff
    if OH is None:
        OH = (H + 2 * pad[0] - kh) // stride + 1
        OW = (W + 2 * pad[1] - kw) // stride + 1
    d.OH, d.OW, d.Cout = OH, OW, Cout
    d.YH, d.YW, d.y_cs, d.y_coff = (y.shape[-3], y.shape[-2], y.shape[-1], y_coff)
    d.pad_y, d.pad_x, d.os, d.oa, d.ob = pad[0], pad[1], os, oa, ob
    if r1 is not None:
        d.r1, d.r1_cs, d.r1_coff, d.r1_cend, d.beta1 = r1.data_ptr(), r1.shape[-1], r1_coff, r1_cend, beta1
    if r2 is not None:
        d.r2, d.r2_cs, d.r2_coff, d.r2_cend, d.beta2 = r2.data_ptr(), r2.shape[-1], r2_coff, r2_cend, beta2
    if mz is not None:
        d.mz, d.mz_cs, d.mz_coff, d.mz_c0 = mz.data_ptr(), mz.shape[-1], mz_coff, mz_c0
    if sign_out is not None:
        d.sign_out = sign_out.data_ptr()
    if sign_in is not None:
        d.sign_in = sign_in.data_ptr()
    d.npar, d.wpar_stride = npar, wpar_stride        # npar = 4: all four parities of a 4x4 stride-2 input gradient (see the header)
    d.alpha, d.slope, d.mslope, d.act = alpha, slope, mslope, int(act)
    d.x_plane, d.y_plane, d.r1_plane, d.r2_plane, d.mz_plane = x_plane, y_plane, r1_plane, r2_plane, mz_plane
    N.check(N.lib().srcgan_conv_igemm(C.byref(d), N.stream_ptr(x.device)), "srcgan_conv_igemm")
    return y


def conv_wgrad(dy: torch.Tensor, x: torch.Tensor, grad: torch.Tensor, *, kh: int, kw: int, stride: int = 1, Cout: int, Cin: int,
               dy_coff: int = 0, x_coff: int = 0, pad: Tuple[int, int] = (0, 0), layout: Tuple[int, int, int, int, int],
               alpha: float = 1.0, nsplit: Optional[int] = None, accumulate: bool = False,
               bias_grad: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dy [B,OH,OW,cs], x [B,H,W,cs] NHWC; grad: canonical f32 tensor written through (sr,sk,sty,stx,off)."""
    N.require_cuda(x, "conv_wgrad")
    lib = N.lib()
    d = N.WgradDesc()
    B, OH, OW, dycs = dy.shape
    _, H, W, xcs = x.shape
    ns = nsplit or lib.srcgan_conv_wgrad_nsplit(B, OH, OW, Cout, Cin, stride)
    slab = torch.empty(lib.srcgan_conv_wgrad_slab_bytes(Cout, Cin, kh, kw, ns), dtype=torch.uint8, device=x.device)
    d.dy, d.x, d.slab, d.grad = dy.data_ptr(), x.data_ptr(), slab.data_ptr(), grad.data_ptr()
    d.bias_grad = bias_grad.data_ptr() if bias_grad is not None else None
    d.dtype = N.dtype_id(x.dtype)
    d.kh, d.kw, d.stride = kh, kw, stride
    d.B, d.H, d.W, d.Cin, d.x_cs, d.x_coff = B, H, W, Cin, xcs, x_coff
    d.OH, d.OW, d.Cout, d.dy_cs, d.dy_coff = OH, OW, Cout, dycs, dy_coff
    d.pad_y, d.pad_x, d.nsplit = pad[0], pad[1], ns
    d.sr, d.sk, d.sty, d.stx, d.off = layout
    d.alpha, d.accumulate = alpha, int(accumulate)
    N.check(lib.srcgan_conv_wgrad(C.byref(d), N.stream_ptr(x.device)), "srcgan_conv_wgrad")
    return grad


def wgrad_dense(dy: torch.Tensor, x: torch.Tensor, segs, *, G: Optional[int] = None, Cc: Optional[int] = None,
                dy_plane: int = 0, x_plane: int = 0, shape: Optional[Tuple[int, int, int]] = None) -> None:
    """Dense-block weight gradient.  dy [B,H,W,>=G], x [B,H,W,>=C] NHWC; segs = [(g0, g1, grad|None, bias|None, Cin, alpha)].
    Blocked-layout operands (make_blocked) pass *_plane = plane stride in bytes, shape=(B,H,W) and explicit G / Cc."""
    N.require_cuda(x, "wgrad_dense")
    lib = N.lib()
    d = N.WgradDenseDesc()
    if shape is not None:
        (B, H, W), dycs = shape, dy.shape[-1]
    else:
        B, H, W, dycs = dy.shape
    G = G or dycs
    Cc = Cc or x.shape[-1]
    dt = N.dtype_id(x.dtype)
    slab = torch.empty(lib.srcgan_wgrad_dense_slab_bytes(G, Cc, dt, B, H, W), dtype=torch.uint8, device=x.device)
    d.dy, d.x, d.slab, d.dtype = dy.data_ptr(), x.data_ptr(), slab.data_ptr(), dt
    d.B, d.H, d.W, d.G, d.dy_cs, d.dy_coff, d.C, d.x_cs, d.x_coff = B, H, W, G, dycs, 0, Cc, x.shape[-1], 0
    d.dy_plane, d.x_plane = dy_plane, x_plane
    d.nseg = len(segs)
    for i, (g0, g1, grad, bias, cin, alpha) in enumerate(segs):
        d.seg[i].g0, d.seg[i].g1, d.seg[i].Cin, d.seg[i].alpha = g0, g1, cin, alpha
        d.seg[i].grad = grad.data_ptr() if grad is not None else None
        d.seg[i].bias = bias.data_ptr() if bias is not None else None
    N.check(lib.srcgan_wgrad_dense(C.byref(d), N.stream_ptr(x.device)), "srcgan_wgrad_dense")


def col_sum(a: torch.Tensor, Cc: int, coff: int = 0, scale: float = 1.0) -> torch.Tensor:
    """sum over pixels of NHWC a[..., coff:coff+C] -> f32 [C] (bias gradient)."""
    N.require_cuda(a, "col_sum")
    lib = N.lib()
    npix = a.numel() // a.shape[-1]
    out = torch.empty(Cc, dtype=torch.float32, device=a.device)
    scr = torch.empty(2 * lib.srcgan_col_reduce_blocks(npix) * Cc, dtype=torch.float32, device=a.device)
    N.check(lib.srcgan_col_reduce(0, a.data_ptr(), a.shape[-1], coff, None, 0, 0, None, None, npix, Cc, scale,
                                  out.data_ptr(), None, scr.data_ptr(), N.dtype_id(a.dtype), N.stream_ptr(a.device)), "srcgan_col_reduce")
    return out


# ---- in-step preprocessing (reference trainCas.py:85-90,104-105; train.py:243,382) -------------------
def rgb_to_gray(x: torch.Tensor) -> torch.Tensor:
    """Y = 0.2125 R + 0.7154 G + 0.0721 B on NCHW f32, keeps a channel dim (trainCas.py:85-87)."""
    N.require_cuda(x, "rgb_to_gray")
    x = x.contiguous().float()
    B, Cc, H, W = x.shape
    if Cc != 3:
        raise ValueError("rgb_to_gray expects 3 channels")
    out = torch.empty(B, 1, H, W, dtype=torch.float32, device=x.device)
    N.check(N.lib().srcgan_rgb_to_gray(x.data_ptr(), out.data_ptr(), B, H, W, N.stream_ptr(x.device)), "srcgan_rgb_to_gray")
    return out


def bilinear_down(x: torch.Tensor, up: int) -> torch.Tensor:
    """F.interpolate(x, scale_factor=1/up, mode='bilinear') for even ``up`` dividing H and W
    (== mean of the centre 2x2 of each up x up block; trainCas.py:89-90)."""
    N.require_cuda(x, "bilinear_down")
    x = x.contiguous().float()
    B, Cc, H, W = x.shape
    out = torch.empty(B, Cc, H // up, W // up, dtype=torch.float32, device=x.device)
    N.check(N.lib().srcgan_bilinear_down(x.data_ptr(), out.data_ptr(), B, Cc, H, W, up, N.stream_ptr(x.device)), "srcgan_bilinear_down")
    return out


def bilinear_up(x: torch.Tensor, up: int) -> torch.Tensor:
    """F.interpolate(x, scale_factor=up, mode='bilinear') (align_corners=False) for an integer factor (trainCasConst.py:91-92)."""
    N.require_cuda(x, "bilinear_up")
    x = x.contiguous().float()
    B, Cc, H, W = x.shape
    out = torch.empty(B, Cc, H * up, W * up, dtype=torch.float32, device=x.device)
    N.check(N.lib().srcgan_bilinear_up(x.data_ptr(), out.data_ptr(), B, Cc, H, W, int(up), N.stream_ptr(x.device)), "srcgan_bilinear_up")
    return out


def nearest_resize(x: torch.Tensor, scale_factor: float) -> torch.Tensor:
    """F.interpolate(x, scale_factor=s) with the default 'nearest' mode (train.py:243,248,382)."""
    N.require_cuda(x, "nearest_resize")
    x = x.contiguous().float()
    B, Cc, H, W = x.shape
    OH, OW = int(H * scale_factor), int(W * scale_factor)
    out = torch.empty(B, Cc, OH, OW, dtype=torch.float32, device=x.device)
    N.check(N.lib().srcgan_nearest_resize(x.data_ptr(), out.data_ptr(), B, Cc, H, W, OH, OW, N.stream_ptr(x.device)), "srcgan_nearest_resize")
    return out


def group_norm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, G: int = 32, *, res: Optional[torch.Tensor] = None,
               relu: bool = False, eps: float = 1e-5, slope: float = 0.0):
    """GroupNorm (+ residual + (Leaky)ReLU) on NHWC [B,H,W,C].  Returns (y, stats[B,G,2] = {mean, rstd})."""
    N.require_cuda(x, "group_norm")
    lib = N.lib()
    B, H, W, Cc = x.shape
    y = torch.empty_like(x)
    stats = torch.empty(B, G, 2, dtype=torch.float32, device=x.device)
    scr = torch.empty(lib.srcgan_gn_scratch_floats(B, Cc), dtype=torch.float32, device=x.device)
    N.check(lib.srcgan_gn_forward(x.data_ptr(), Cc, res.data_ptr() if res is not None else None, Cc, y.data_ptr(), Cc,
                                  gamma.data_ptr() if gamma is not None else None, beta.data_ptr() if beta is not None else None, stats.data_ptr(),
                                  B, H * W, Cc, G, eps, int(relu), slope,
                                  N.dtype_id(x.dtype), scr.data_ptr(), N.stream_ptr(x.device)), "srcgan_gn_forward")
    return y, stats


def group_norm_bwd(dy: torch.Tensor, x: torch.Tensor, gamma: torch.Tensor, stats: torch.Tensor, G: int = 32, *,
                   yact: Optional[torch.Tensor] = None, want_dres: bool = False, slope: float = 0.0):
    """Backward of group_norm: returns (dx, dres | None, dgamma, dbeta); yact = forward output when ReLU was applied."""
    N.require_cuda(x, "group_norm_bwd")
    lib = N.lib()
    B, H, W, Cc = x.shape
    dx = torch.empty_like(x)
    dres = torch.empty_like(x) if want_dres else None
    dgamma = torch.empty(Cc, dtype=torch.float32, device=x.device)
    dbeta = torch.empty(Cc, dtype=torch.float32, device=x.device)
    scr = torch.empty(lib.srcgan_gn_scratch_floats(B, Cc), dtype=torch.float32, device=x.device)
    N.check(lib.srcgan_gn_backward(dy.data_ptr(), Cc, yact.data_ptr() if yact is not None else None, Cc, x.data_ptr(), Cc,
                                   gamma.data_ptr() if gamma is not None else None, stats.data_ptr(), dx.data_ptr(), Cc,
                                   dres.data_ptr() if want_dres else None, Cc, 0,
                                   dgamma.data_ptr(), dbeta.data_ptr(), 0, slope, B, H * W, Cc, G, N.dtype_id(x.dtype), scr.data_ptr(),
                                   N.stream_ptr(x.device)), "srcgan_gn_backward")
    return dx, dres, dgamma, dbeta
