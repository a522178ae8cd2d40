"""Host-side mirror of the reference model classes on the hot path.

Same class names, constructor signatures, ``state_dict`` keys / shapes / dtypes and NCHW f32
tensor interface as the reference, so ``eval(opt.SRModel)(1, 1, opt.up)``
(trainCas.py:30) and reference ``.pth`` checkpoints keep working -- but ``forward`` / backward run
as one native call each into libsrcgan_amd.so (hand-written gfx950 kernels).  The ``nn.Conv2d`` /
``nn.BatchNorm2d`` children are *parameter holders only* (they give the reference's key names and
its default initialisation); their own ``forward`` is never used and there is no CPU fallback.

  RDDBNet              <- reference src/model/rddb.py:85-114
  NLayerDiscriminator  <- reference src/model/model.py:595-639
  RDDBNetA             <- named by reference src/train.py:11,173 but defined nowhere; build-defined
                          HR->LR mirror (strided 3x3 s2 conv + LeakyReLU per /2 stage, trunk at LR).
"""
from __future__ import annotations

import ctypes as C
import functools
import math
from typing import List, Optional

import torch
import torch.nn as nn

from . import _native as N

__all__ = ["RDDBNet", "RDDBNetA", "RDDBNetB", "LegacyRDDBNet", "ResDeconv", "ESPCN", "SRCNN", "EDSR", "SRDN", "NLayerDiscriminator", "ResidualDenseBlock_5", "RRDB", "deconv",
           "get_deconv_params"]


def get_deconv_params(upscale_factor):
    """(kernel_size, stride, output_padding) of the reference's deconv helper (rddb.py:9-25)."""
    table = {2: (2, 2), 4: (2, 4), 8: (4, 8)}
    if upscale_factor not in table:
        raise ValueError(f"unsupported upscale_factor {upscale_factor}")
    k, s = table[upscale_factor]
    return k, s, s - k


def deconv(in_planes, out_planes, upscale_factor=2):
    """Parameter holder equal to the reference's deconv() (rddb.py:28-38)."""
    k, s, opad = get_deconv_params(upscale_factor)
    return nn.ConvTranspose2d(in_planes, out_planes, kernel_size=k, stride=s, padding=0, bias=False, output_padding=opad)


class _HolderOnly(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover
        raise NotImplementedError(
            f"{type(self).__name__} is a parameter holder of the native srcgan_amd network; call the whole "
            "network (RDDBNet / NLayerDiscriminator), which runs as fused gfx950 kernels.")


class ResidualDenseBlock_5(_HolderOnly):
    """Holder for conv1..conv5 of one dense block (rddb.py:48-60)."""

    def __init__(self, nf=64, gc=32, bias=True):
        super().__init__()
        for k in range(5):
            setattr(self, f"conv{k + 1}", nn.Conv2d(nf + k * gc, gc if k < 4 else nf, 3, 1, 1, bias=bias))


class RRDB(_HolderOnly):
    """Holder for RDB1..RDB3 (rddb.py:71-76)."""

    def __init__(self, nf, gc=32):
        super().__init__()
        for j in (1, 2, 3):
            setattr(self, f"RDB{j}", ResidualDenseBlock_5(nf, gc))


def _kaiming_like_reference(module: nn.Module) -> None:
    # rddb.py:100-105: kaiming-normal(fan_out, relu) on every nn.Conv2d weight; biases and
    # ConvTranspose2d weights keep torch's default init.
    for m in module.modules():
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
        elif isinstance(m, nn.BatchNorm2d):
            nn.init.constant_(m.weight, 1)
            nn.init.constant_(m.bias, 0)


class _PackState:
    """Persistent packed (MFMA-order, compute-dtype) copy of a module's weights: packed once per weight update instead of once per
    native call -- a cycle step runs each generator three times forward and three times backward on the same weights
    (reference train.py:228-260, 331-333).

    Validity is decided ON THE DEVICE, per call and without a host synchronisation: ``srcgan_params_fingerprint`` hashes every bit
    of every parameter (66 MB for the 23-block generator: ~15 us) into the second word of a guard pair, and the pack kernels of
    the native call do nothing when it equals the first word -- the fingerprint the pack was made from (``srcgan_net_opts.pack
    == 2``).  So ANY way of changing the weights is honoured: optimiser steps, ``load_state_dict``, ``p.data.mul_()`` (which does
    not bump autograd's version counter), a raw-pointer write by another library.  The host only forces a pack when the buffer is
    new, the layout changed or the parameters moved (their pointer table is rebuilt then); ``invalidate()`` forces one too."""

    def __init__(self):
        self.buf = None
        self.layout = None
        self.table = None            # device int64 [n, 3]: pointer, elements, first fingerprint block
        self.table_key = None
        self.nblocks = 0
        self.guard = None            # device int64 [4]: forward {packed-from, now}, backward {packed-from, now}
        self.valid = [False, False]  # forward / backward pack made at least once from the current buffer + table

    def invalidate(self):
        """Force the next forward and the next backward to re-pack (never needed for correctness: see the class comment)."""
        self.valid = [False, False]

    def opts(self, lib_bytes_fn, cfg, params, extra, backward, device):
        """-> NetOpts for one native call.  Call ``done(backward)`` after the call succeeded."""
        need = int(lib_bytes_fn(C.byref(cfg)))
        if self.buf is None or self.buf.numel() < need or self.buf.device != device or self.layout != extra:
            self.buf = torch.empty(need, dtype=torch.uint8, device=device)
            self.guard = torch.zeros(4, dtype=torch.int64, device=device)
            self.layout = extra
            self.valid = [False, False]
        key = tuple((p.data_ptr(), p.numel()) for p in params)
        if key != self.table_key:
            rows, blk = [], 0
            for ptr, n in key:
                rows.append((ptr, n, blk))
                blk += (n + 16383) // 16384              # srcgan_params_fingerprint: 16 Ki-element slices
            self.table = torch.tensor(rows, dtype=torch.int64).to(device)
            self.table_key, self.nblocks = key, blk
            self.valid = [False, False]
        slot = 2 if backward else 0
        gptr = self.guard.data_ptr() + 8 * slot
        N.check(N.lib().srcgan_params_fingerprint(self.table.data_ptr(), len(key), self.nblocks, gptr + 8, N.stream_ptr(device)),
                "srcgan_params_fingerprint")
        return N.NetOpts(self.buf.data_ptr(), 2 if self.valid[1 if backward else 0] else 1, 0, 0, gptr)

    def done(self, backward):
        """The native call that packed (or verified) returned without error: what is packed now matches the fingerprint."""
        slot = 2 if backward else 0
        self.guard[slot:slot + 1].copy_(self.guard[slot + 1:slot + 2])
        self.valid[1 if backward else 0] = True


class _GradArena:
    """One flat f32 buffer for all parameter gradients a backward call produces; ``views[i]`` aliases it with the parameter's
    shape.  autograd's AccumulateGrad adopts the views as ``.grad`` (no copy), so the data-parallel all-reduce runs on slices of
    the flat buffer in place -- no flatten / copy-back passes (srcgan_amd.dist.GradSync)."""

    def __init__(self, params, needs):
        sizes = [p.numel() if n else 0 for p, n in zip(params, needs)]
        self.flat = torch.empty(sum(sizes), dtype=torch.float32, device=params[0].device)
        self.views, self.offsets, off = [], [0], 0
        for p, n, sz in zip(params, needs, sizes):
            self.views.append(self.flat[off:off + sz].view_as(p) if n else None)
            off += sz
            self.offsets.append(off)          # offsets[i] .. offsets[i + 1] = parameter i


class _RddbFn(torch.autograd.Function):
    """One native forward / one native backward for the whole generator."""

    @staticmethod
    def forward(ctx, x, cfg_items, *params):
        N.require_cuda(x, "RDDBNet.forward")
        lib = N.lib()
        in_ch, out_ch, up, nf, nb, gc, dtype, down = cfg_items[:8]
        legacy = cfg_items[8] if len(cfg_items) > 8 else 0
        pstate = cfg_items[9] if len(cfg_items) > 9 else None
        if x.dim() != 4 or x.shape[1] != in_ch:
            raise ValueError(f"RDDBNet expects [B,{in_ch},H,W], got {tuple(x.shape)}")
        x = x.detach().contiguous().float()
        B, _, H, W = x.shape
        if B == 0:          # an empty batch yields an empty output and zero gradients, as aten::convolution does
            f0 = (up if down == 0 else 1)
            ctx.empty = True
            ctx.save_for_backward(*params)
            return x.new_zeros((0, out_ch) + ((H * f0, W * f0) if down <= 1 else (H // down, W // down)))
        ctx.empty = False
        cfg = N.RddbCfg(in_ch, out_ch, up, nf, nb, gc, B, H, W, dtype, down, legacy)
        for p in params:
            N.require_cuda(p, "RDDBNet parameter")
        plist = [p.detach().contiguous() for p in params]
        if any(p.dtype != torch.float32 for p in plist):
            raise TypeError("RDDBNet parameters must be float32 (canonical weights stay f32)")
        ws = N.workspace(lib.srcgan_rddbnet_ws_bytes(C.byref(cfg)), x.device)
        f = (up if down == 0 else 1)
        HO, WO = (H * f, W * f) if down <= 1 else (H // down, W // down)
        y = torch.empty(B, out_ch, HO, WO, dtype=torch.float32, device=x.device)
        layout = (in_ch, out_ch, up, nf, nb, gc, dtype, down, legacy)
        opt = pstate.opts(lib.srcgan_rddbnet_wpack_bytes, cfg, plist, layout, False, x.device) if pstate is not None else None
        N.check(lib.srcgan_rddbnet_forward_ex(C.byref(cfg), x.data_ptr(), N.ptr_array(plist), ws.data_ptr(), y.data_ptr(),
                                              C.byref(opt) if opt is not None else None, N.stream_ptr(x.device)), "srcgan_rddbnet_forward")
        if pstate is not None:
            pstate.done(False)
        ctx.cfg, ctx.ws, ctx.n = cfg, ws, len(plist)
        ctx.pstate, ctx.layout = pstate, layout
        ctx.save_for_backward(*plist)
        ctx.phase_hook = _phase_hooks.get("rddb")
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = N.lib()
        params = list(ctx.saved_tensors)
        if ctx.empty:
            return (None, None, *[torch.zeros_like(p) if ctx.needs_input_grad[2 + i] else None for i, p in enumerate(params)])
        cfg = ctx.cfg
        if ctx.ws is None:
            raise RuntimeError("RDDBNet backward called twice (activations were released)")
        dy = dy.contiguous().float()
        need_dx = ctx.needs_input_grad[0]
        arena = _GradArena(params, [ctx.needs_input_grad[2 + i] for i in range(len(params))])
        grads = arena.views
        scratch = N.workspace(lib.srcgan_rddbnet_bwd_scratch_bytes(C.byref(cfg)), dy.device)
        dx = torch.empty(cfg.B, cfg.in_ch, cfg.H, cfg.W, dtype=torch.float32, device=dy.device) if need_dx else None
        pstate = ctx.pstate
        opt = pstate.opts(lib.srcgan_rddbnet_wpack_bytes, cfg, params, ctx.layout, True, dy.device) if pstate is not None else N.NetOpts(None, 0, 0, 0, None)
        gptr, pptr = N.ptr_array(grads), N.ptr_array(params)
        hook = ctx.phase_hook
        nrr = 0 if cfg.legacy == 2 else (2 * cfg.nb if cfg.legacy == 3 else cfg.nb)
        # phases: RRDB ranges, last to first (one phase unless a data-parallel hook asks for more)
        cuts = hook.cuts(cfg, nrr, params) if (hook is not None and nrr > 1) else [0]
        hi = nrr
        for lo in sorted(set(cuts) | {0}, reverse=True):
            if lo >= hi and hi != nrr:
                continue
            opt.rrdb_lo, opt.rrdb_hi = (lo, hi) if nrr > 0 else (0, 0)
            N.check(lib.srcgan_rddbnet_backward_ex(C.byref(cfg), dy.data_ptr(), pptr, ctx.ws.data_ptr(), scratch.data_ptr(), gptr,
                                                   dx.data_ptr() if need_dx else None, C.byref(opt), N.stream_ptr(dy.device)),
                    "srcgan_rddbnet_backward")
            if opt.pack and pstate is not None:
                pstate.done(True)
            opt.pack = 0
            if hook is not None:
                hook.phase_done(arena, params, cfg, lo, hi, nrr)
            hi = lo
        ctx.ws = None
        return (dx, None, *grads)


# Data-parallel hooks (srcgan_amd.dist.GradSync.attach): an object with ``cuts(nrr) -> [rrdb indices]`` (phase boundaries of the
# generator's backward) and ``phase_done(arena, params, cfg, lo, hi, nrr)``, called right after the native call that finalised the
# gradients of RRDBs [lo, hi) (plus the tail when hi == nrr, the head when lo == 0) was queued: the hook launches their all-reduce
# on its side stream while the next phase computes.
_phase_hooks = {}


class RDDBNet(nn.Module):
    """RRDB generator, drop-in for reference ``model.RDDBNet`` (rddb.py:85-114).

    ``forward(x[B,in_ch,H,W] f32 NCHW) -> [B,ou_ch,H*up,W*up]``.  ``dtype``: 'fp32' (default; exact
    f32 MFMA, <=1e-3 of the CPU reference) or 'bf16' (perf mode)."""

    def __init__(self, in_ch, ou_ch, upscale_factor, nf=64, nb=3, gc=32, dtype=None):
        super().__init__()
        self.conv_first = nn.Conv2d(in_ch, nf, 3, 1, 1, bias=True)
        self.RRDB_trunk = nn.Sequential(*[RRDB(nf=nf, gc=gc) for _ in range(nb)])
        self.trunk_conv = nn.Conv2d(nf, nf, 3, 1, 1, bias=True)
        self.upscale_factor = upscale_factor
        ups = []
        for _ in range(int(math.log2(upscale_factor))):
            ups += [deconv(nf, nf, upscale_factor=2), nn.LeakyReLU(negative_slope=0.2, inplace=True)]
        self.upscale_layers = nn.Sequential(*ups)
        self.conv_last = nn.Conv2d(nf, ou_ch, 3, 1, 1, bias=False)
        _kaiming_like_reference(self)
        self._cfg = (in_ch, ou_ch, upscale_factor, nf, nb, gc)
        self.compute_dtype = N.dtype_name(dtype)
        self._pack = _PackState()

    def _down(self):
        return 0

    def invalidate_packed_weights(self):
        """Force a re-pack of the kernels' weight copy at the next forward / backward.  Not needed for correctness -- every native
        call verifies the copy against a device-side fingerprint of the parameters (see _PackState) -- kept as an explicit hook."""
        self._pack.invalidate()

    def forward(self, x):
        cfg = (*self._cfg, N.dtype_id(self.compute_dtype), self._down(), 0, self._pack)
        # parameters in state_dict order == the order the native planner assumes
        return _RddbFn.apply(x, cfg, *self.parameters())

    def extra_repr(self):
        return f"native gfx950, compute_dtype={self.compute_dtype}"


class RDDBNetA(RDDBNet):
    """HR->LR generator G_B of the cycle (reference train.py:173,178 names ``RDDBNetA`` but ships no
    definition -> build-defined, "parity unpinned" vs the reference; pinned against oracle.rddbneta_forward).
    conv_first -> [conv3x3 s2 + bias + LeakyReLU] x log2(down) -> RRDB trunk at LR -> trunk_conv + skip -> conv_last."""

    def __init__(self, in_ch, ou_ch, down_factor, nf=64, nb=3, gc=32, dtype=None):
        super().__init__(in_ch, ou_ch, 1, nf=nf, nb=nb, gc=gc, dtype=dtype)
        self.down_factor = down_factor
        downs = []
        for _ in range(int(math.log2(down_factor))):
            downs += [nn.Conv2d(nf, nf, 3, 2, 1, bias=True), nn.LeakyReLU(negative_slope=0.2, inplace=True)]
        self.down_layers = nn.Sequential(*downs)
        _kaiming_like_reference(self.down_layers)

    def _down(self):
        return max(1, self.down_factor)

    def forward(self, x):
        cfg = (*self._cfg, N.dtype_id(self.compute_dtype), self._down(), 0, self._pack)
        # native order: conv_first, down_layers, trunk, trunk_conv, conv_last
        ps = [self.conv_first.weight, self.conv_first.bias, *self.down_layers.parameters(),
              *self.RRDB_trunk.parameters(), self.trunk_conv.weight, self.trunk_conv.bias, self.conv_last.weight]
        return _RddbFn.apply(x, cfg, *ps)


class _LegacyRRDB(_HolderOnly):
    """Holder for the RRDB of model/model.py:214-226: like rddb.py's, but its constructor already re-initialises its
    own convolutions (kaiming-normal) -- kept so that a seeded construction consumes the RNG exactly like the reference."""

    def __init__(self, nf, gc=32):
        super().__init__()
        for j in (1, 2, 3):
            setattr(self, f"RDB{j}", ResidualDenseBlock_5(nf, gc))
        _kaiming_like_reference(self)


_MODES = {"x1": 1, "x2": 2, "x4": 4}


class RDDBNetB(nn.Module):
    """Legacy nearest-up-sampling generator, drop-in for reference ``model.model.RDDBNetB`` (model/model.py:394-440; G_A of
    train.py:172,177).  ``RDDBNetB(in_nc, out_nc, nf, nb=3, gc=32, mode='x2')``; forward: conv_first -> RRDB trunk ->
    trunk_conv + skip -> [nearest x2 -> upconv -> LeakyReLU] (x4: upconv1 then upconv2; x2: upconv1 twice, the second
    without up-sampling) -> HRconv + LeakyReLU eight times -> conv_last (with bias).  Any other mode leaves the
    resolution unchanged in the reference; only 'x2' / 'x4' are accepted here."""

    _legacy = 1
    _tail = ("upconv1", "upconv2", "HRconv")

    def __init__(self, in_nc, out_nc, nf, nb=3, gc=32, mode="x2", dtype=None):
        super().__init__()
        if mode not in _MODES or (self._legacy == 1 and mode == "x1"):
            raise NotImplementedError(f"{type(self).__name__}: mode {mode!r} is not supported")
        self.conv_first = nn.Conv2d(in_nc, nf, 3, 1, 1, bias=True)
        self.RRDB_trunk = nn.Sequential(*[_LegacyRRDB(nf=nf, gc=gc) for _ in range(nb)])
        self.trunk_conv = nn.Conv2d(nf, nf, 3, 1, 1, bias=True)
        for name in self._tail:
            setattr(self, name, nn.Conv2d(nf, nf, 3, 1, 1, bias=True))
        self.mode = mode
        self.nb = nb
        self.conv_last = nn.Conv2d(nf, out_nc, 3, 1, 1, bias=True)
        self.lrelu = nn.LeakyReLU(negative_slope=0.2, inplace=True)
        _kaiming_like_reference(self)
        self._cfg = (in_nc, out_nc, _MODES[mode], nf, nb, gc)
        self.compute_dtype = N.dtype_name(dtype)

    def forward(self, x):
        cfg = (*self._cfg, N.dtype_id(self.compute_dtype), 0, self._legacy)
        ps = list(self.parameters())                            # state_dict order == native order
        if self.mode == "x2":       # upconv2 is not on the x2 graph (model.py:430-432): its .grad stays None, as in the reference
            skip = {id(self.upconv2.weight), id(self.upconv2.bias)}
            ps = [p.detach() if id(p) in skip else p for p in ps]
        return _RddbFn.apply(x, cfg, *ps)

    def extra_repr(self):
        return f"native gfx950, mode={self.mode}, compute_dtype={self.compute_dtype}"


class LegacyRDDBNet(RDDBNetB):
    """Drop-in for the *legacy* ``model.model.RDDBNet`` (model/model.py:347-391; not the rddb.py class that
    ``from model import *`` exports): ``(in_nc, out_nc, nf, nb, gc=32, mode='x2')``.  Its forward computes the RRDB trunk
    and discards it (model.py:382-383), so the output is conv_first -> [nearest x2 -> upconv -> LeakyReLU] per x2 stage
    (mode 'x1': upconv without up-sampling) -> HRconv + LeakyReLU twice -> conv_last; the trunk parameters exist in the
    state_dict and receive no gradient (``.grad`` stays None, as in the reference)."""

    _legacy = 2
    _tail = ("upconv", "HRconv")

    def __init__(self, in_nc, out_nc, nf, nb, gc=32, mode="x2", dtype=None):
        super().__init__(in_nc, out_nc, nf, nb=nb, gc=gc, mode=mode, dtype=dtype)

    def forward(self, x):
        cfg = (*self._cfg, N.dtype_id(self.compute_dtype), 0, self._legacy)
        # the trunk is not on the graph: pass its parameters detached so autograd leaves their .grad at None
        trunk = {id(p) for p in self.RRDB_trunk.parameters()} | {id(self.trunk_conv.weight), id(self.trunk_conv.bias)}
        ps = [p.detach() if id(p) in trunk else p for p in self.parameters()]
        return _RddbFn.apply(x, cfg, *ps)


# ------------------------------------------------------------------------------------------------ ResDeconv colouriser
class _ResDeconvFn(torch.autograd.Function):
    """One native forward / one native backward for the whole colouriser (resdeconv.py:164-195)."""

    @staticmethod
    def forward(ctx, x, out_ch, dtype, layers, norm, *params):
        N.require_cuda(x, "ResDeconv.forward")
        lib = N.lib()
        if x.dim() != 4 or x.shape[1] != 3:
            raise ValueError(f"ResDeconv's stem expects 3 channels, got {tuple(x.shape)}")
        x = x.detach().contiguous().float()
        B, _, H, W = x.shape
        if H % 16 or W % 16:
            raise ValueError(f"ResDeconv needs H and W to be multiples of 16 (four stride-2 stages), got {H}x{W}")
        cfg = N.ResDeconvCfg(3, out_ch, B, H, W, dtype, (C.c_int * 4)(*layers), norm)
        for p in params:
            N.require_cuda(p, "ResDeconv parameter")
        plist = [p.detach().contiguous() for p in params]
        ws = N.workspace(lib.srcgan_resdeconv_ws_bytes(C.byref(cfg)), x.device)
        y = torch.empty(B, out_ch, H, W, dtype=torch.float32, device=x.device)
        N.check(lib.srcgan_resdeconv_forward(C.byref(cfg), x.data_ptr(), N.ptr_array(plist), ws.data_ptr(), y.data_ptr(),
                                             N.stream_ptr(x.device)), "srcgan_resdeconv_forward")
        ctx.cfg, ctx.ws = cfg, ws
        ctx.save_for_backward(*plist)
        ctx.phase_hook = _phase_hooks.get("resdeconv")       # one rule for every network: the hook in force at FORWARD time
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = N.lib()
        params = list(ctx.saved_tensors)
        cfg = ctx.cfg
        if ctx.ws is None:
            raise RuntimeError("ResDeconv backward called twice (activations were released)")
        need_dx = ctx.needs_input_grad[0]
        dy = dy.contiguous().float()
        dx = torch.empty(cfg.B, 3, cfg.H, cfg.W, dtype=torch.float32, device=dy.device) if need_dx else None
        arena = _GradArena(params, [ctx.needs_input_grad[5 + i] for i in range(len(params))])
        grads = arena.views
        scratch = N.workspace(lib.srcgan_resdeconv_bwd_scratch_bytes(C.byref(cfg)), dy.device)
        N.check(lib.srcgan_resdeconv_backward(C.byref(cfg), dy.data_ptr(), N.ptr_array(params), ctx.ws.data_ptr(), scratch.data_ptr(),
                                              N.ptr_array(grads), dx.data_ptr() if need_dx else None, N.stream_ptr(dy.device)), "srcgan_resdeconv_backward")
        ctx.ws = None
        if ctx.phase_hook is not None:
            ctx.phase_hook.phase_done(arena, params, cfg, 0, 0, 0)
        return (dx, None, None, None, None, *grads)


class _BasicBlockHolder(_HolderOnly):
    """Parameter holder of resdeconv.py:56-76 BasicBlock (attribute order = the reference's state_dict order)."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, BN="GN"):
        super().__init__()
        norm = (lambda c: nn.InstanceNorm2d(c)) if BN == "IN" else (lambda c: nn.GroupNorm(32, c))
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = norm(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = norm(planes)
        self.downsample = downsample
        self.stride = stride


class ResDeconv(nn.Module):
    """Colouriser, drop-in for reference ``model.ResDeconv`` (src/model/resdeconv.py:99-195):
    ``ResDeconv(src_ch=1, tar_ch=3, block=None, layers=[2, 2, 2, 2], BN='GN')`` -- the reference's positional signature
    (resdeconv.py:107).  ``block``: BasicBlock, the only block the reference defines.  ``layers``: BasicBlocks per stage, any
    counts ([2, 2, 2, 2] = ResNet-18 layout, the default; [3, 4, 6, 3] = ResNet-34); the up path uses layers[2], layers[1],
    layers[0] (resdeconv.py:131-137).  ``BN``: 'GN' (GroupNorm(32, C), the default) or 'IN' (nn.InstanceNorm2d(C): no parameters);
    'BN' (BatchNorm2d) is refused.  ``forward(x[B,src_ch,H,W]) -> [B,tar_ch,H,W]`` (H, W multiples of 16).
    A 1-channel source is replicated to 3 channels like the reference (resdeconv.py:166-167)."""

    def __init__(self, src_ch=1, tar_ch=3, block=None, layers=(2, 2, 2, 2), BN="GN", dtype=None):
        super().__init__()
        if block is not None and getattr(block, "__name__", str(block)) not in ("BasicBlock", "_BasicBlockHolder"):
            raise NotImplementedError("native ResDeconv implements block=BasicBlock (the only block resdeconv.py defines)")
        layers = [int(v) for v in layers]
        if len(layers) != 4 or min(layers) < 1:
            raise ValueError("ResDeconv: layers must be four positive block counts")
        if BN not in ("GN", "IN"):
            raise NotImplementedError("native ResDeconv implements BN='GN' (GroupNorm(32, C), the reference's default) and BN='IN' (InstanceNorm2d)")
        self.src_ch = src_ch
        if isinstance(tar_ch, list):
            tar_ch = sum(tar_ch)
        self.tar_ch = tar_ch
        self.layers_cfg, self.BN = tuple(layers), BN
        self.inplanes = 64
        # creation order == the reference's (it fixes which random numbers each default initialisation consumes)
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.InstanceNorm2d(64) if BN == "IN" else nn.GroupNorm(32, 64)
        self.relu = nn.ReLU(inplace=True)
        self.layer1 = self._make_layer(64, layers[0], 1)
        self.layer2 = self._make_layer(128, layers[1], 2)
        self.layer3 = self._make_layer(256, layers[2], 2)
        self.layer4 = self._make_layer(512, layers[3], 2)
        self.deconv10 = nn.ConvTranspose2d(512, 256, 2, 2, 0, bias=False)
        self.inplanes = 256
        self.upRes1 = self._make_layer(256, layers[2], 1)
        self.deconv11 = nn.ConvTranspose2d(256, 128, 2, 2, 0, bias=False)
        self.inplanes = 128
        self.upRes2 = self._make_layer(128, layers[1], 1)
        self.deconv12 = nn.ConvTranspose2d(128, 64, 2, 2, 0, bias=False)
        self.inplanes = 64
        self.upRes3 = self._make_layer(64, layers[0], 1)
        self.deconv13 = nn.ConvTranspose2d(64, 64, 2, 2, 0, bias=False)
        self.pred = nn.Conv2d(64, tar_ch, kernel_size=3, stride=1, padding=1, bias=False)
        _kaiming_like_reference(self)
        self.compute_dtype = N.dtype_name(dtype)

    def _make_layer(self, planes, blocks, stride):
        downsample = None
        norm = nn.InstanceNorm2d(planes) if self.BN == "IN" else nn.GroupNorm(32, planes)
        if stride != 1 or self.inplanes != planes:
            downsample = nn.Sequential(nn.Conv2d(self.inplanes, planes, kernel_size=1, stride=stride, bias=False), norm)
        layers = [_BasicBlockHolder(self.inplanes, planes, stride, downsample, BN=self.BN)]
        self.inplanes = planes
        for _ in range(1, blocks):
            layers.append(_BasicBlockHolder(self.inplanes, planes, BN=self.BN))
        return nn.Sequential(*layers)

    def forward(self, x):
        if self.src_ch == 1:
            x = torch.cat([x, x, x], dim=1)
        return _ResDeconvFn.apply(x, self.tar_ch, N.dtype_id(self.compute_dtype), self.layers_cfg, 1 if self.BN == "IN" else 0, *self.parameters())

    def extra_repr(self):
        return f"native gfx950, compute_dtype={self.compute_dtype}"


# ------------------------------------------------------------------------------------------------ ESPCN / SRCNN
class _SrNetFn(torch.autograd.Function):
    """One native forward / backward for the small --SRModel networks (kind 0 ESPCN, 1 SRCNN)."""

    @staticmethod
    def forward(ctx, x, cfg_items, *params):
        N.require_cuda(x, "ESPCN/SRCNN forward")
        lib = N.lib()
        kind, in_ch, out_ch, up, base, dtype = cfg_items[:6]
        nres = cfg_items[6] if len(cfg_items) > 6 else 0
        if x.dim() != 4 or x.shape[1] != in_ch:
            raise ValueError(f"expected [B,{in_ch},H,W], got {tuple(x.shape)}")
        x = x.detach().contiguous().float()
        B, _, H, W = x.shape
        cfg = N.SrNetCfg(kind, in_ch, out_ch, up, base, B, H, W, dtype, nres)
        for p in params:
            N.require_cuda(p, "parameter")
        plist = [p.detach().contiguous() for p in params]
        ws = N.workspace(lib.srcgan_srnet_ws_bytes(C.byref(cfg)), x.device)
        f = 1 if kind == 1 else up
        y = torch.empty(B, out_ch, H * f, W * f, dtype=torch.float32, device=x.device)
        N.check(lib.srcgan_srnet_forward(C.byref(cfg), x.data_ptr(), N.ptr_array(plist), ws.data_ptr(), y.data_ptr(), N.stream_ptr(x.device)),
                "srcgan_srnet_forward")
        ctx.cfg, ctx.ws = cfg, ws
        ctx.save_for_backward(*plist)
        ctx.phase_hook = _phase_hooks.get("srnet")
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = N.lib()
        params = list(ctx.saved_tensors)
        cfg = ctx.cfg
        if ctx.ws is None:
            raise RuntimeError("backward called twice (activations were released)")
        need_dx = ctx.needs_input_grad[0]
        dy = dy.contiguous().float()
        dx = torch.empty(cfg.B, cfg.in_ch, cfg.H, cfg.W, dtype=torch.float32, device=dy.device) if need_dx else None
        arena = _GradArena(params, [ctx.needs_input_grad[2 + i] for i in range(len(params))])
        grads = arena.views
        scratch = N.workspace(lib.srcgan_srnet_bwd_scratch_bytes(C.byref(cfg)), dy.device)
        N.check(lib.srcgan_srnet_backward(C.byref(cfg), dy.data_ptr(), N.ptr_array(params), ctx.ws.data_ptr(), scratch.data_ptr(),
                                          N.ptr_array(grads), dx.data_ptr() if need_dx else None, N.stream_ptr(dy.device)), "srcgan_srnet_backward")
        ctx.ws = None
        if ctx.phase_hook is not None:
            ctx.phase_hook.phase_done(arena, params, cfg, 0, 0, 0)
        return (dx, None, *grads)


class ESPCN(nn.Module):
    """Drop-in for reference ``model.ESPCN`` (src/model/espcn.py:18-51; the CLI default ``--SRModel``, trainCas.py:169):
    5x5, 3x3, 3x3 convolutions + ReLU, 3x3 to 64 r^2 channels, PixelShuffle(r), 3x3."""

    def __init__(self, in_ch=3, ou_ch=3, upscale_factor=2, base_kernel=64, dtype=None):
        super().__init__()
        kernels = [int(x * base_kernel) for x in [1, 1, 1 / 2]]
        self.relu = nn.ReLU(True)
        self.conv1 = nn.Conv2d(in_ch, kernels[0], kernel_size=5, stride=1, padding=2)
        self.conv2 = nn.Conv2d(kernels[0], kernels[1], kernel_size=3, stride=1, padding=1)
        self.conv3 = nn.Conv2d(kernels[1], kernels[2], kernel_size=3, stride=1, padding=1)
        self.conv4 = nn.Conv2d(kernels[2], base_kernel * upscale_factor ** 2, kernel_size=3, stride=1, padding=1)
        self.pixel_shuffle = nn.PixelShuffle(upscale_factor)
        self.conv5 = nn.Conv2d(base_kernel, ou_ch, kernel_size=3, stride=1, padding=1)
        _kaiming_like_reference(self)
        self._cfg = (0, in_ch, ou_ch, upscale_factor, base_kernel)
        self.compute_dtype = N.dtype_name(dtype)

    def forward(self, x):
        return _SrNetFn.apply(x, (*self._cfg, N.dtype_id(self.compute_dtype)), *self.parameters())

    def extra_repr(self):
        return f"native gfx950, compute_dtype={self.compute_dtype}"


class SRCNN(nn.Module):
    """Drop-in for reference ``model.SRCNN`` (src/model/srcnn.py:17-42): 9x9, 1x1, 5x5 convolutions, each followed by ReLU;
    the output has the input's size (``upscale_factor`` is stored and unused, as in the reference); torch default init."""

    def __init__(self, in_ch=3, ou_ch=3, upscale_factor=2, base_kernel=64, dtype=None):
        super().__init__()
        kernels = [int(x * base_kernel) for x in [1, 1 / 2]]
        self.up = upscale_factor
        self.relu = nn.ReLU(True)
        self.conv1 = nn.Conv2d(in_ch, kernels[0], kernel_size=9, stride=1, padding=4)
        self.conv2 = nn.Conv2d(kernels[0], kernels[1], kernel_size=1, stride=1, padding=0)
        self.conv3 = nn.Conv2d(kernels[1], ou_ch, kernel_size=5, stride=1, padding=2)
        self._cfg = (1, in_ch, ou_ch, upscale_factor, base_kernel)
        self.compute_dtype = N.dtype_name(dtype)

    def forward(self, x):
        return _SrNetFn.apply(x, (*self._cfg, N.dtype_id(self.compute_dtype)), *self.parameters())

    def extra_repr(self):
        return f"native gfx950, compute_dtype={self.compute_dtype}"


class _EdsrBlock(_HolderOnly):
    """Parameter holder of edsr.py:37-50 ResnetBlock: conv1, conv2, ONE GroupNorm(32) applied after each."""

    def __init__(self, num_channel):
        super().__init__()
        self.conv1 = nn.Conv2d(num_channel, num_channel, 3, 1, 1)
        self.conv2 = nn.Conv2d(num_channel, num_channel, 3, 1, 1)
        self.gn = nn.GroupNorm(32, num_channel)
        self.activation = nn.LeakyReLU(negative_slope=0.2, inplace=True)


class EDSR(nn.Module):
    """Drop-in for reference ``model.EDSR`` (src/model/edsr.py:68-110): ``EDSR(in_ch, ou_ch, upscale_factor=2, base_channel=64,
    num_residuals=50)``."""

    def __init__(self, in_ch, ou_ch, upscale_factor=2, base_channel=64, num_residuals=50, dtype=None):
        super().__init__()
        self.input_conv = nn.Conv2d(in_ch, base_channel, kernel_size=3, stride=1, padding=1)
        self.residual_layers = nn.Sequential(*[_EdsrBlock(base_channel) for _ in range(num_residuals)])
        self.mid_conv = nn.Conv2d(base_channel, base_channel, kernel_size=3, stride=1, padding=1)
        self.upscale_layers = nn.Sequential(*[nn.ConvTranspose2d(base_channel, base_channel, 2, 2, 0, bias=False)
                                              for _ in range(int(math.log2(upscale_factor)))])
        self.output_conv = nn.Conv2d(base_channel, ou_ch, kernel_size=3, stride=1, padding=1)
        _kaiming_like_reference(self)
        self._cfg = (2, in_ch, ou_ch, upscale_factor, base_channel)
        self._nres = num_residuals
        self.compute_dtype = N.dtype_name(dtype)

    def forward(self, x):
        return _SrNetFn.apply(x, (*self._cfg, N.dtype_id(self.compute_dtype), self._nres), *self.parameters())

    def extra_repr(self):
        return f"native gfx950, compute_dtype={self.compute_dtype}"


class SRDN(nn.Module):
    """Drop-in for reference ``model.SRDN`` (src/model/srdn.py:56-74): conv_first -> RRDB_encoder (nb RRDBs) -> + skip ->
    RRDB_decoder (nb RRDBs) -> + skip -> conv_last; same resolution in and out (``upscale_factor`` is stored and unused, and
    ``trunk_conv`` exists in the state_dict but is never applied -- it receives no gradient, as in the reference)."""

    def __init__(self, in_ch, ou_ch, upscale_factor, nf=64, nb=3, gc=32, dtype=None):
        super().__init__()
        self.upscale_factor = upscale_factor
        self.conv_first = nn.Conv2d(in_ch, nf, 3, 1, 1, bias=True)
        self.RRDB_encoder = nn.Sequential(*[RRDB(nf=nf, gc=gc) for _ in range(nb)])
        self.trunk_conv = nn.Conv2d(nf, nf, 3, 1, 1, bias=True)
        self.RRDB_decoder = nn.Sequential(*[RRDB(nf=nf, gc=gc) for _ in range(nb)])
        self.conv_last = nn.Conv2d(nf, ou_ch, 3, 1, 1, bias=False)
        _kaiming_like_reference(self)
        self._cfg = (in_ch, ou_ch, 1, nf, nb, gc)
        self.compute_dtype = N.dtype_name(dtype)

    def forward(self, x):
        cfg = (*self._cfg, N.dtype_id(self.compute_dtype), 0, 3)
        skip = {id(self.trunk_conv.weight), id(self.trunk_conv.bias)}
        return _RddbFn.apply(x, cfg, *[p.detach() if id(p) in skip else p for p in self.parameters()])

    def extra_repr(self):
        return f"native gfx950, compute_dtype={self.compute_dtype}"


class _NLayerDFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, cfg_items, running, nbt, *params):
        N.require_cuda(x, "NLayerDiscriminator.forward")
        lib = N.lib()
        in_ch, ndf, n_layers, dtype, training, pstate = cfg_items[:6]
        norm = cfg_items[6] if len(cfg_items) > 6 else 0
        if x.dim() != 4 or x.shape[1] != in_ch:
            raise ValueError(f"NLayerDiscriminator expects [B,{in_ch},H,W], got {tuple(x.shape)}")
        x = x.detach().contiguous().float()
        B, _, H, W = x.shape
        cfg = N.NLayerDCfg(in_ch, ndf, n_layers, B, H, W, dtype, int(training), norm)
        plist = [p.detach().contiguous() for p in params]
        oh, ow = C.c_int(), C.c_int()
        N.check(lib.srcgan_nlayerd_out_hw(C.byref(cfg), C.byref(oh), C.byref(ow)), "srcgan_nlayerd_out_hw")
        ws = N.workspace(lib.srcgan_nlayerd_ws_bytes(C.byref(cfg)), x.device)
        y = torch.empty(B, 1, oh.value, ow.value, dtype=torch.float32, device=x.device)
        layout = (in_ch, ndf, n_layers, dtype, H % 2, W % 2, norm)    # (the first layer's packed form depends on the parity of H, W)
        opt = pstate.opts(lib.srcgan_nlayerd_wpack_bytes, cfg, plist, layout, False, x.device)
        N.check(lib.srcgan_nlayerd_forward_ex(C.byref(cfg), x.data_ptr(), N.ptr_array(plist), N.ptr_array(running),
                                              N.ptr_array(nbt), ws.data_ptr(), y.data_ptr(), C.byref(opt), N.stream_ptr(x.device)),
                "srcgan_nlayerd_forward")
        pstate.done(False)
        ctx.cfg, ctx.ws = cfg, ws
        ctx.pstate, ctx.layout = pstate, layout
        ctx.save_for_backward(*plist)
        ctx.phase_hook = _phase_hooks.get("nlayerd")
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = N.lib()
        params = list(ctx.saved_tensors)
        cfg = ctx.cfg
        if ctx.ws is None:
            raise RuntimeError("NLayerDiscriminator backward called twice (activations were released)")
        dy = dy.contiguous().float()
        need_dx = ctx.needs_input_grad[0]
        needs = [ctx.needs_input_grad[4 + i] for i in range(len(params))]
        arena = _GradArena(params, needs) if any(needs) else None
        grads = arena.views if arena is not None else [None] * len(params)
        scratch = N.workspace(lib.srcgan_nlayerd_bwd_scratch_bytes(C.byref(cfg)), dy.device)
        dx = torch.empty(cfg.B, cfg.in_ch, cfg.H, cfg.W, dtype=torch.float32, device=dy.device) if need_dx else None
        opt = ctx.pstate.opts(lib.srcgan_nlayerd_wpack_bytes, cfg, params, ctx.layout, True, dy.device)
        N.check(lib.srcgan_nlayerd_backward_ex(C.byref(cfg), dy.data_ptr(), N.ptr_array(params), ctx.ws.data_ptr(),
                                               scratch.data_ptr(), N.ptr_array(grads), dx.data_ptr() if need_dx else None,
                                               C.byref(opt), N.stream_ptr(dy.device)), "srcgan_nlayerd_backward")
        ctx.pstate.done(True)
        ctx.ws = None
        if ctx.phase_hook is not None and arena is not None:
            ctx.phase_hook.phase_done(arena, params, cfg, 0, 0, 0)
        return (dx, None, None, None, *grads)


class NLayerDiscriminator(nn.Module):
    """PatchGAN discriminator, drop-in for reference ``model.model.NLayerDiscriminator``
    (model/model.py:595-639): conv4x4 s2 + LeakyReLU | (n-1) x [conv4x4 s2, BatchNorm2d, LeakyReLU] |
    conv4x4 s1, BN, LeakyReLU | conv4x4 s1 -> 1 channel.  BatchNorm uses per-replica batch statistics
    in train mode and updates the running buffers like nn.BatchNorm2d.

    ``norm_layer`` as in the reference: ``nn.BatchNorm2d`` (default) or ``nn.InstanceNorm2d`` -- the class or the
    ``functools.partial`` that ``basicModel.get_norm_layer`` builds (affine=False, track_running_stats=False); with InstanceNorm2d
    the normalised convolutions have a bias (model/model.py:607-610) and the layers normalise every image by its own statistics in
    train and eval mode."""

    def __init__(self, input_nc, ndf=64, n_layers=3, norm_layer=nn.BatchNorm2d, dtype=None):
        super().__init__()
        nkw = {}
        if isinstance(norm_layer, functools.partial):
            nkw, norm_layer = dict(norm_layer.keywords), norm_layer.func
        if norm_layer is nn.InstanceNorm2d:
            if nkw.get("affine", False) or nkw.get("track_running_stats", False):
                raise NotImplementedError("native NLayerDiscriminator: InstanceNorm2d without affine parameters / running statistics "
                                          "(what nn.InstanceNorm2d defaults to and basicModel.get_norm_layer('instance') builds)")
            inorm = True
        elif norm_layer is nn.BatchNorm2d:
            if not nkw.get("affine", True) or not nkw.get("track_running_stats", True):
                raise NotImplementedError("native NLayerDiscriminator: BatchNorm2d with affine parameters and running statistics")
            inorm = False
        else:
            raise NotImplementedError("native NLayerDiscriminator implements norm_layer = nn.BatchNorm2d (the reference's default) or nn.InstanceNorm2d")
        norm = (lambda c: nn.InstanceNorm2d(c)) if inorm else (lambda c: nn.BatchNorm2d(c))
        kw, padw = 4, 1
        seq = [nn.Conv2d(input_nc, ndf, kw, 2, padw), nn.LeakyReLU(0.2, True)]
        mult = 1
        for n in range(1, n_layers):
            prev, mult = mult, min(2 ** n, 8)
            seq += [nn.Conv2d(ndf * prev, ndf * mult, kw, 2, padw, bias=inorm), norm(ndf * mult), nn.LeakyReLU(0.2, True)]
        prev, mult = mult, min(2 ** n_layers, 8)
        seq += [nn.Conv2d(ndf * prev, ndf * mult, kw, 1, padw, bias=inorm), norm(ndf * mult), nn.LeakyReLU(0.2, True)]
        seq += [nn.Conv2d(ndf * mult, 1, kw, 1, padw)]
        self.model = nn.Sequential(*seq)
        self._cfg = (input_nc, ndf, n_layers)
        self._norm = 1 if inorm else 0
        self.compute_dtype = N.dtype_name(dtype)
        self._pack = _PackState()

    def invalidate_packed_weights(self):
        """See RDDBNet.invalidate_packed_weights."""
        self._pack.invalidate()

    def forward(self, input):
        bns = [m for m in self.model if isinstance(m, nn.BatchNorm2d)]
        running = [t for m in bns for t in (m.running_mean, m.running_var)]
        nbt = [m.num_batches_tracked for m in bns]
        cfg = (*self._cfg, N.dtype_id(self.compute_dtype), self.training, self._pack, self._norm)
        return _NLayerDFn.apply(input, cfg, running, nbt, *self.parameters())

    def extra_repr(self):
        return f"native gfx950, compute_dtype={self.compute_dtype}"
