"""ctypes binding of libsrcgan_amd.so (the C ABI declared in include/srcgan_amd.h).

The product path has NO fallback: if the shared library is missing or a call fails, a
RuntimeError is raised.  Nothing here imports ``oracle``.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# SRCGAN_AMD_LIB: developer override used by the kernel-variant experiments in scripts/ (A/B builds side by side)
LIB_PATH = os.environ.get("SRCGAN_AMD_LIB") or os.path.join(_HERE, "lib", "libsrcgan_amd.so")

F32, BF16, F16 = 0, 1, 2
_DTYPES = {"fp32": F32, "f32": F32, "float32": F32, torch.float32: F32,
           "bf16": BF16, "bfloat16": BF16, torch.bfloat16: BF16,
           "fp16": F16, "f16": F16, "float16": F16, "half": F16, torch.float16: F16, F32: F32, BF16: BF16, F16: F16}
_default_dtype = _DTYPES[os.environ.get("SRCGAN_AMD_DTYPE", "fp32")]


def dtype_id(d) -> int:
    if d is None:
        return _default_dtype
    try:
        return _DTYPES[d]
    except KeyError:
        raise ValueError(f"unsupported compute dtype {d!r} (use 'fp32', 'bf16' or 'fp16')") from None


def dtype_name(d) -> str:
    return {F32: "fp32", BF16: "bf16", F16: "fp16"}[dtype_id(d)]


def set_default_dtype(d) -> None:
    """Compute dtype of modules built afterwards: 'fp32' (exact f32 MFMA, parity mode,
    the default), 'bf16' (bf16 storage + MFMA, f32 accumulate; perf mode) or 'fp16' (the same on IEEE half)."""
    global _default_dtype
    _default_dtype = _DTYPES[d]


class ConvDesc(C.Structure):
    _fields_ = [("x", C.c_void_p), ("wp", C.c_void_p), ("bias", C.c_void_p), ("y", C.c_void_p),
                ("r1", C.c_void_p), ("r2", C.c_void_p), ("mz", C.c_void_p),
                ("dtype", C.c_int), ("kh", C.c_int), ("kw", C.c_int), ("stride", C.c_int),
                ("B", C.c_int), ("H", C.c_int), ("W", C.c_int), ("Cin", C.c_int), ("x_cs", C.c_int), ("x_coff", C.c_int),
                ("OH", C.c_int), ("OW", C.c_int), ("Cout", C.c_int),
                ("YH", C.c_int), ("YW", C.c_int), ("y_cs", C.c_int), ("y_coff", C.c_int),
                ("pad_y", C.c_int), ("pad_x", C.c_int), ("os", C.c_int), ("oa", C.c_int), ("ob", C.c_int),
                ("r1_cs", C.c_int), ("r1_coff", C.c_int), ("r1_cend", C.c_int),
                ("r2_cs", C.c_int), ("r2_coff", C.c_int), ("r2_cend", C.c_int),
                ("mz_cs", C.c_int), ("mz_coff", C.c_int), ("mz_c0", C.c_int),
                ("alpha", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("slope", C.c_float), ("mslope", C.c_float),
                ("act", C.c_int),
                ("x_plane", C.c_long), ("y_plane", C.c_long), ("r1_plane", C.c_long), ("r2_plane", C.c_long), ("mz_plane", C.c_long), ("rev_batch", C.c_int),
                ("sign_out", C.c_void_p), ("sign_in", C.c_void_p), ("npar", C.c_int), ("wpar_stride", C.c_long)]


class WgradDesc(C.Structure):
    _fields_ = [("dy", C.c_void_p), ("x", C.c_void_p), ("slab", C.c_void_p), ("grad", C.c_void_p), ("bias_grad", C.c_void_p),
                ("dtype", C.c_int), ("kh", C.c_int), ("kw", C.c_int), ("stride", C.c_int),
                ("B", C.c_int), ("H", C.c_int), ("W", C.c_int), ("Cin", C.c_int), ("x_cs", C.c_int), ("x_coff", C.c_int),
                ("OH", C.c_int), ("OW", C.c_int), ("Cout", C.c_int), ("dy_cs", C.c_int), ("dy_coff", C.c_int),
                ("pad_y", C.c_int), ("pad_x", C.c_int), ("nsplit", C.c_int),
                ("sr", C.c_long), ("sk", C.c_long), ("sty", C.c_long), ("stx", C.c_long), ("off", C.c_long),
                ("alpha", C.c_float), ("accumulate", C.c_int)]


class WgradSeg(C.Structure):
    _fields_ = [("g0", C.c_int), ("g1", C.c_int), ("grad", C.c_void_p), ("bias", C.c_void_p), ("Cin", C.c_int), ("alpha", C.c_float)]


class WgradDenseDesc(C.Structure):
    _fields_ = [("dy", C.c_void_p), ("x", C.c_void_p), ("slab", C.c_void_p), ("dtype", C.c_int),
                ("B", C.c_int), ("H", C.c_int), ("W", C.c_int),
                ("G", C.c_int), ("dy_cs", C.c_int), ("dy_coff", C.c_int),
                ("C", C.c_int), ("x_cs", C.c_int), ("x_coff", C.c_int),
                ("nseg", C.c_int), ("seg", WgradSeg * 8), ("accumulate", C.c_int), ("dy_plane", C.c_long), ("x_plane", C.c_long)]


class RddbCfg(C.Structure):
    _fields_ = [("in_ch", C.c_int), ("out_ch", C.c_int), ("up", C.c_int), ("nf", C.c_int), ("nb", C.c_int), ("gc", C.c_int),
                ("B", C.c_int), ("H", C.c_int), ("W", C.c_int), ("dtype", C.c_int), ("down", C.c_int), ("legacy", C.c_int)]


class NetOpts(C.Structure):
    _fields_ = [("wpack", C.c_void_p), ("pack", C.c_int), ("rrdb_lo", C.c_int), ("rrdb_hi", C.c_int), ("guard", C.c_void_p)]


class ResDeconvCfg(C.Structure):
    _fields_ = [("in_ch", C.c_int), ("out_ch", C.c_int), ("B", C.c_int), ("H", C.c_int), ("W", C.c_int), ("dtype", C.c_int),
                ("layers", C.c_int * 4), ("norm", C.c_int)]


class SrNetCfg(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("kind", "in_ch", "out_ch", "up", "base", "B", "H", "W", "dtype", "nres")]


class NLayerDCfg(C.Structure):
    _fields_ = [("in_ch", C.c_int), ("ndf", C.c_int), ("n_layers", C.c_int),
                ("B", C.c_int), ("H", C.c_int), ("W", C.c_int), ("dtype", C.c_int), ("training", C.c_int), ("norm", C.c_int)]


# name -> (restype, argtypes).  Must list every symbol include/srcgan_amd.h declares
# (tests/test_abi.py checks the header against this table and the built library).
_P, _I, _L, _F, _S = C.c_void_p, C.c_int, C.c_long, C.c_float, C.c_size_t
SIGNATURES = {
    "srcgan_version": (_I, []),
    "srcgan_last_error": (C.c_char_p, []),
    "srcgan_dtype_size": (_I, [_I]),
    "srcgan_nchw_f32_to_nhwc": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "srcgan_nhwc_to_nchw_f32": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "srcgan_packed_weight_bytes": (_S, [_I, _I, _I, _I]),
    "srcgan_pack_weight": (_I, [_P, _P, _I, _I, _I, _I, _L, _L, _L, _L, _L, _I, _P]),
    "srcgan_pack_weight_part": (_I, [_P, _P, _I, _I, _I, _I, _L, _L, _L, _L, _L, _I, _I, _F, _I, _P]),
    "srcgan_conv_igemm": (_I, [C.POINTER(ConvDesc), _P]),
    "srcgan_conv_wgrad_slab_bytes": (_S, [_I, _I, _I, _I, _I]),
    "srcgan_conv_wgrad_nsplit": (_I, [_I, _I, _I, _I, _I, _I]),
    "srcgan_conv_wgrad": (_I, [C.POINTER(WgradDesc), _P]),
    "srcgan_wgrad_dense_slab_bytes": (_S, [_I, _I, _I, _I, _I, _I]),
    "srcgan_wgrad_dense": (_I, [C.POINTER(WgradDenseDesc), _P]),
    "srcgan_col_reduce_blocks": (_I, [_L]),
    "srcgan_col_reduce": (_I, [_I, _P, _I, _I, _P, _I, _I, _P, _P, _L, _I, _F, _P, _P, _P, _I, _P]),
    "srcgan_bn_finalize": (_I, [_P, _P, _P, _P, _P, _P, _I, _L, _F, _F, _P]),
    "srcgan_bn_eval_rstd": (_I, [_P, _P, _I, _F, _P]),
    "srcgan_bn_apply_lrelu": (_I, [_P, _P, _P, _P, _P, _P, _L, _I, _I, _F, _I, _P]),
    "srcgan_bn_bwd_apply": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _I, _P]),
    "srcgan_add_inplace": (_I, [_P, _I, _I, _P, _I, _I, _P, _I, _I, _F, _L, _I, _I, _P]),
    "srcgan_add_inplace_planes": (_I, [_P, _I, _I, _L, _P, _I, _I, _L, _P, _I, _I, _L, _F, _L, _I, _I, _P]),
    "srcgan_gn_scratch_floats": (_S, [_I, _I]),
    "srcgan_gn_forward": (_I, [_P, _I, _P, _I, _P, _I, _P, _P, _P, _I, _L, _I, _I, _F, _I, _F, _I, _P, _P]),
    "srcgan_gn_backward": (_I, [_P, _I, _P, _I, _P, _I, _P, _P, _P, _I, _P, _I, _I, _P, _P, _I, _F, _I, _L, _I, _I, _I, _P, _P]),
    "srcgan_upsample2_nhwc": (_I, [_P, _I, _I, _L, _P, _I, _I, _I, _I, _I, _I, _P]),
    "srcgan_sum2x2_nhwc": (_I, [_P, _I, _P, _I, _P, _I, _F, _I, _I, _I, _I, _I, _P]),
    "srcgan_loss_scratch_floats": (_I, []),
    "srcgan_loss_fwd": (_I, [_I, _P, _P, _F, _L, _P, _P, _P]),
    "srcgan_loss_bwd": (_I, [_I, _P, _P, _F, _L, _P, _F, _P, _P]),
    "srcgan_psnr_from_mse": (_I, [_P, _P, _P]),
    "srcgan_rgb_to_gray": (_I, [_P, _P, _I, _I, _I, _P]),
    "srcgan_bilinear_down": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "srcgan_nearest_resize": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "srcgan_bilinear_up": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "srcgan_resdeconv_num_params": (_I, [C.POINTER(ResDeconvCfg)]),
    "srcgan_resdeconv_ws_bytes": (_S, [C.POINTER(ResDeconvCfg)]),
    "srcgan_resdeconv_bwd_scratch_bytes": (_S, [C.POINTER(ResDeconvCfg)]),
    "srcgan_resdeconv_forward": (_I, [C.POINTER(ResDeconvCfg), _P, _P, _P, _P, _P]),
    "srcgan_resdeconv_backward": (_I, [C.POINTER(ResDeconvCfg), _P, _P, _P, _P, _P, _P, _P]),
    "srcgan_srnet_num_params": (_I, [C.POINTER(SrNetCfg)]),
    "srcgan_srnet_ws_bytes": (_S, [C.POINTER(SrNetCfg)]),
    "srcgan_srnet_bwd_scratch_bytes": (_S, [C.POINTER(SrNetCfg)]),
    "srcgan_srnet_forward": (_I, [C.POINTER(SrNetCfg), _P, _P, _P, _P, _P]),
    "srcgan_srnet_backward": (_I, [C.POINTER(SrNetCfg), _P, _P, _P, _P, _P, _P, _P]),
    "srcgan_pixel_shuffle_nhwc": (_I, [_P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "srcgan_mask_inplace": (_I, [_P, _P, _F, _L, _I, _P]),
    "srcgan_metric_scratch_floats": (_I, [_I, _I, _I, _I]),
    "srcgan_metric_ae": (_I, [_P, _P, _I, _I, _I, _I, _P, _P, _P]),
    "srcgan_metric_ssim": (_I, [_P, _P, _I, _I, _I, _I, _P, _P, _P]),
    "srcgan_nchw_f32_to_s2d": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "srcgan_s2d_to_nchw_f32": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "srcgan_s2d_wgrad_unfold": (_I, [_P, _P, _I, _I, _I, _P]),
    "srcgan_u8rgb_to_planes": (_I, [_P, _P, _I, _L, _I, _P]),
    "srcgan_lab_planes_to_u8rgb": (_I, [_P, _P, _I, _L, _P]),
    "srcgan_adam_step": (_I, [_P, _P, _I, C.c_double, C.c_double, C.c_double, C.c_double, _L, _P]),
    "srcgan_params_fingerprint": (_I, [_P, _I, _L, _P, _P]),
    "srcgan_prof_enable": (_I, [_I]),
    "srcgan_prof_collect": (_I, []),
    "srcgan_prof_get": (_I, [_I, C.POINTER(C.c_char_p), C.POINTER(C.c_long), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "srcgan_rddbnet_num_params": (_I, [C.POINTER(RddbCfg)]),
    "srcgan_rddbnet_ws_bytes": (_S, [C.POINTER(RddbCfg)]),
    "srcgan_rddbnet_bwd_scratch_bytes": (_S, [C.POINTER(RddbCfg)]),
    "srcgan_rddbnet_forward": (_I, [C.POINTER(RddbCfg), _P, _P, _P, _P, _P]),
    "srcgan_rddbnet_backward": (_I, [C.POINTER(RddbCfg), _P, _P, _P, _P, _P, _P, _P]),
    "srcgan_rddbnet_wpack_bytes": (_S, [C.POINTER(RddbCfg)]),
    "srcgan_rddbnet_forward_ex": (_I, [C.POINTER(RddbCfg), _P, _P, _P, _P, C.POINTER(NetOpts), _P]),
    "srcgan_rddbnet_backward_ex": (_I, [C.POINTER(RddbCfg), _P, _P, _P, _P, _P, _P, C.POINTER(NetOpts), _P]),
    "srcgan_nlayerd_wpack_bytes": (_S, [C.POINTER(NLayerDCfg)]),
    "srcgan_nlayerd_forward_ex": (_I, [C.POINTER(NLayerDCfg), _P, _P, _P, _P, _P, _P, C.POINTER(NetOpts), _P]),
    "srcgan_nlayerd_backward_ex": (_I, [C.POINTER(NLayerDCfg), _P, _P, _P, _P, _P, _P, C.POINTER(NetOpts), _P]),
    "srcgan_nlayerd_num_params": (_I, [C.POINTER(NLayerDCfg)]),
    "srcgan_nlayerd_out_hw": (_I, [C.POINTER(NLayerDCfg), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "srcgan_nlayerd_ws_bytes": (_S, [C.POINTER(NLayerDCfg)]),
    "srcgan_nlayerd_bwd_scratch_bytes": (_S, [C.POINTER(NLayerDCfg)]),
    "srcgan_nlayerd_forward": (_I, [C.POINTER(NLayerDCfg), _P, _P, _P, _P, _P, _P, _P]),
    "srcgan_nlayerd_backward": (_I, [C.POINTER(NLayerDCfg), _P, _P, _P, _P, _P, _P, _P]),
}

_lib = None


def lib() -> C.CDLL:
    """The loaded native library.  Raises (never falls back) when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"srcgan_amd: native library {LIB_PATH} is missing. Build it with "
                "`python -m srcgan_amd.build` (hipcc --offload-arch=gfx950). There is no CPU fallback.")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)       # AttributeError -> a declared symbol is not exported
            fn.restype, fn.argtypes = res, args
        _lib = handle
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().srcgan_last_error()
        raise RuntimeError(f"srcgan_amd: {what} failed: {msg.decode() if msg else 'unknown error'}")


def stream_ptr(device=None) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def require_cuda(t: torch.Tensor, what: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(
            f"srcgan_amd: {what} got a {t.device} tensor. The native path runs on an MI355X GPU only; "
            "there is no CPU fallback (move the module and its inputs to 'cuda').")


def ptr_array(tensors: Sequence[Optional[torch.Tensor]]):
    arr = (C.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = None if t is None else t.data_ptr()
    return arr


def workspace(nbytes: int, device) -> torch.Tensor:
    if nbytes <= 0:
        raise RuntimeError("srcgan_amd: native planner rejected the configuration: "
                           + (lib().srcgan_last_error() or b"").decode())
    return torch.empty(nbytes, dtype=torch.uint8, device=device)


def prof_enable(on: bool) -> None:
    check(lib().srcgan_prof_enable(int(on)), "srcgan_prof_enable")


def prof_collect():
    """-> list of dicts {cls, count, ms, flops, bytes} aggregated per kernel class since the last collect."""
    l = lib()
    out = []
    for i in range(l.srcgan_prof_collect()):
        cls, cnt, ms, fl, by = C.c_char_p(), C.c_long(), C.c_double(), C.c_double(), C.c_double()
        check(l.srcgan_prof_get(i, C.byref(cls), C.byref(cnt), C.byref(ms), C.byref(fl), C.byref(by)), "srcgan_prof_get")
        out.append({"cls": cls.value.decode(), "count": cnt.value, "ms": ms.value, "flops": fl.value, "bytes": by.value})
    return out
