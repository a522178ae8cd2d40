"""Fused multi-tensor Adam behind the ``torch.optim.Adam`` surface.

``Adam`` IS a ``torch.optim.Adam``: same constructor, same ``param_groups`` / ``state`` / ``state_dict()`` layout
(``step`` as a CPU scalar tensor, ``exp_avg``, ``exp_avg_sq`` per parameter), so checkpoints written by either load in
the other.  Only ``step()`` differs: one native launch per parameter group (``srcgan_adam_step``) instead of torch's
foreach kernel sequence -- 697 tensors for the 23-block generator.  Option combinations the kernel does not implement
(weight decay, amsgrad, maximize, capturable, differentiable, non-f32 or CPU parameters) fall back to torch's own step.
``fuse(optimizer)`` converts an existing ``torch.optim.Adam`` instance in place (what the reference harness constructs).
"""
import os
from typing import List

import numpy as np
import torch

from . import _native as N

__all__ = ["Adam", "fuse"]

_CHUNK = 4096
_DISABLED = bool(os.environ.get("SRCGAN_TORCH_ADAM"))      # A/B switch for bench.py


class Adam(torch.optim.Adam):
    def _fusable(self, group) -> bool:
        return not (group.get("weight_decay", 0) or group.get("amsgrad") or group.get("maximize") or group.get("capturable")
                    or group.get("differentiable") or group.get("fused") or isinstance(group["lr"], torch.Tensor))

    @torch.no_grad()
    def step(self, closure=None):
        if _DISABLED or not all(self._fusable(g) for g in self.param_groups) or any(
                p.grad is not None and (not p.is_cuda or p.dtype != torch.float32 or p.grad.dtype != torch.float32 or p.grad.is_sparse
                                        or not p.is_contiguous() or not p.grad.is_contiguous())
                for g in self.param_groups for p in g["params"]):
            return super().step(closure)
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = N.lib()
        cache = self.__dict__.setdefault("_srcgan_tables", {})
        for gi, group in enumerate(self.param_groups):
            ps: List[torch.Tensor] = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            for p in ps:
                st = self.state[p]
                if len(st) == 0:            # torch's lazy state initialisation (optim/adam.py _init_group)
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            step_ts = [self.state[p]["step"] for p in ps]
            torch._foreach_add_(step_ts, 1)
            steps = {int(step_ts[0]), int(step_ts[-1])} if len(ps) < 64 else {int(t) for t in (step_ts[0], step_ts[len(ps) // 2], step_ts[-1])}
            if len(steps) != 1:             # parameters that joined the group at different times
                raise RuntimeError("srcgan_amd.optim.Adam: parameters of one group have different step counts; use torch.optim.Adam")
            ptrs = np.array([(p.data_ptr(), p.grad.data_ptr(), self.state[p]["exp_avg"].data_ptr(), self.state[p]["exp_avg_sq"].data_ptr())
                             for p in ps], dtype=np.uint64)
            shapes = tuple(p.numel() for p in ps)
            ent = cache.get(gi)
            if ent is None or ent["shapes"] != shapes:
                tid = np.concatenate([np.full((n + _CHUNK - 1) // _CHUNK, i, dtype=np.int32) for i, n in enumerate(shapes)])
                off = np.concatenate([np.arange(0, n, _CHUNK, dtype=np.int32) for n in shapes])
                cnt = np.minimum(np.array(shapes, dtype=np.int64)[tid] - off, _CHUNK).astype(np.int32)
                chunks = np.stack([tid, off, cnt, np.zeros_like(tid)], axis=1)
                ent = {"shapes": shapes, "chunks": torch.from_numpy(chunks).to(ps[0].device), "n": int(len(tid)), "ptrs": None,
                       "tensors": torch.empty(len(ps) * 4, dtype=torch.int64, device=ps[0].device),
                       # two pinned staging buffers, used alternately: the upload is asynchronous (a pageable copy would make the
                       # host wait for the whole backward pass queued on the stream).  The host may run several steps ahead of the
                       # device (nothing in a training step synchronises), so each buffer carries the event of its last upload and
                       # is rewritten only after that copy has executed.
                       "pinned": [torch.empty(len(ps) * 4, dtype=torch.int64).pin_memory() for _ in range(2)], "events": [None, None], "flip": 0}
                cache[gi] = ent
            if ent["ptrs"] is None or not np.array_equal(ent["ptrs"], ptrs):      # fresh .grad tensors move between steps
                ent["ptrs"] = ptrs
                f = ent["flip"]
                ent["flip"] ^= 1
                stage = ent["pinned"][f]
                if ent["events"][f] is not None:
                    ent["events"][f].synchronize()          # the copy that last read this staging buffer has run
                stage.numpy()[:] = ptrs.view(np.int64).reshape(-1)
                ent["tensors"].copy_(stage, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(ps[0].device))
                ent["events"][f] = ev
            b1, b2 = group["betas"]
            N.check(lib.srcgan_adam_step(ent["tensors"].data_ptr(), ent["chunks"].data_ptr(), ent["n"], float(group["lr"]), float(b1), float(b2),
                                         float(group["eps"]), steps.pop(), N.stream_ptr(ps[0].device)), "srcgan_adam_step")
            # the native kernel wrote the parameters through raw pointers: tell autograd (version counters), as an in-place torch op
            # would have -- the modules' packed-weight caches are keyed on them (srcgan_amd.model._PackState)
            torch.autograd.graph.increment_version(ps)
        return loss


def fuse(optimizer: torch.optim.Optimizer) -> torch.optim.Optimizer:
    """Give an existing ``torch.optim.Adam`` instance the fused ``step()`` (its state and param_groups are untouched)."""
    if type(optimizer) is torch.optim.Adam:
        optimizer.__class__ = Adam
    return optimizer
