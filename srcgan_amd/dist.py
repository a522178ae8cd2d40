"""Data-parallel replicas: one process per GPU, torch.distributed ('nccl' == RCCL over xGMI on ROCm;
'gloo' for the CPU tests).  The reference has no distributed code at all;
this is the build's addition: parameter broadcast at start (C1) and the gradient mean (C2).

Images are independent units, so the hot path shards with no data-path collective; the only exchange
is the gradient mean.  It happens INSIDE every native backward call (``GradSync.attach``): the call writes
its parameter gradients into one flat f32 arena, the generator's backward runs in phases over RRDB ranges
(``srcgan_net_opts.rrdb_lo/hi``), and when a phase has been queued the arena slice it finalised is
all-reduced in place on a side stream (<= bucket_mb pieces: few, large collectives -- xGMI is
point-to-point, ring steps are per-link bound) while the next phase computes on the main stream.  What
autograd receives from ``backward`` is already the mean over replicas, so the harnesses need no separate
synchronisation step and a network called several times per step (the cycle's generators, train.py:228-260)
needs one reduce per backward call.  A network that runs SEVERAL times per optimiser step (the cycle's generators,
train.py:228-260: three passes each; a discriminator's real + fake pass; StackedSR's micro-batches) is registered with
``GradSync.once(module)``: its backward calls then only accumulate locally, and ``GradSync.sync(params)`` -- what every harness
calls in front of ``optimizer.step()`` -- reduces the ACCUMULATED gradient once, in place in the flat arena the ``.grad``s alias
(one parameter-sized exchange per network and step instead of one per call; the mean is linear, so the result is the same).
BatchNorm statistics stay per replica (no SyncBN), as single-device reference semantics imply.  ``GradSync.allreduce(params)`` is
the plain post-backward form (bucketed, asynchronous, flatten + copy-back) for anything that is not one of this package's networks.
"""
from __future__ import annotations

import os
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist

__all__ = ["init_from_env", "broadcast_module", "GradSync", "shard_range", "dist_info", "visible_gpu_count"]


# SRCGAN_FORCE_DIST=1: create the process group and run the collectives even with one rank (rehearses the RCCL path --
# rendezvous, broadcast, bucketed all-reduce on the side stream -- on a one-GPU box; tests/test_gpu_dist.py)
_FORCE = os.environ.get("SRCGAN_FORCE_DIST") == "1"


def init_from_env(backend: Optional[str] = None):
    """Initialise torch.distributed from RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torchrun contract).
    Returns (rank, local_rank, world).  world == 1 -> no process group is created."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal hooks for a one-GPU box (tests/test_gpu_dist.py): several ranks share device SRCGAN_LOCAL_DEVICE and talk
    # over SRCGAN_DIST_BACKEND=gloo (RCCL refuses two ranks on one device)
    if os.environ.get("SRCGAN_LOCAL_DEVICE") is not None:
        local = int(os.environ["SRCGAN_LOCAL_DEVICE"])
    backend = backend or os.environ.get("SRCGAN_DIST_BACKEND")
    if (world > 1 or _FORCE) and not dist.is_initialized():
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local, world


def visible_gpu_count() -> int:
    """GPUs this process may use, WITHOUT opening the HIP runtime (a launcher that then starts one process per GPU stays GPU-free):
    the visibility lists HIP honours if set, else the nodes the kernel driver (KFD) lists with a non-zero SIMD count; -1 = unknown."""
    for var in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None and v.strip() != "":
            return len([t for t in v.split(",") if t.strip() != ""])
    n, root = 0, "/sys/class/kfd/kfd/topology/nodes"
    try:
        for node in os.listdir(root):
            try:
                with open(os.path.join(root, node, "properties")) as f:
                    props = dict(line.split(None, 1) for line in f if " " in line)
                if int(props.get("simd_count", "0")) > 0:
                    n += 1
            except (OSError, ValueError):
                continue
    except OSError:
        return -1                # no KFD topology readable here: unknown (the ranks themselves will find out)
    return n


def dist_info(device=None) -> dict:
    """What the process group really is -- for the bench line: backend, world size, the number of ranks an all-reduce of ones
    actually saw (proves the collective crossed every rank), and the RCCL/NCCL version when the backend is 'nccl'."""
    if not dist.is_initialized():
        return {"backend": None, "world_size": 1, "ranks_seen": 1, "nccl_version": None}
    backend = dist.get_backend()
    one = torch.ones(1, device=device if (device is not None and backend == "nccl") else "cpu" if backend == "gloo" else device)
    dist.all_reduce(one)
    ver = None
    if backend == "nccl":
        try:
            ver = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:      # pragma: no cover
            ver = "unknown"
    return {"backend": backend, "world_size": dist.get_world_size(), "ranks_seen": int(round(float(one.item()))), "nccl_version": ver}


def shard_range(n_units: int, rank: int, world: int):
    """Contiguous [begin, end) share of n independent units (images) for this rank."""
    base, rem = divmod(n_units, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


@torch.no_grad()
def broadcast_module(module: torch.nn.Module, src: int = 0) -> None:
    """Make every replica identical to rank `src` (parameters and buffers), one flat message per dtype."""
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not _FORCE):
        return
    tensors = [t for t in list(module.parameters()) + list(module.buffers())]
    by_dtype = {}
    for t in tensors:
        by_dtype.setdefault(t.dtype, []).append(t)
    for dt, ts in by_dtype.items():
        flat = torch.cat([t.detach().reshape(-1) for t in ts])
        dist.broadcast(flat, src)
        off = 0
        for t in ts:
            n = t.numel()
            t.copy_(flat[off:off + n].view_as(t))
            off += n


class GradSync:
    """Mean of parameter gradients across replicas.  ``allreduce(params)`` flattens the existing ``.grad``s
    into <= bucket_mb buckets (reverse parameter order: the gradients produced first by backward go first),
    launches every all-reduce asynchronously and writes the averaged values back before returning the
    stream to the optimiser.  Parameters without a gradient (frozen discriminator, train.py:330) are skipped."""

    def __init__(self, bucket_mb: float = 32.0, group=None, phases: int = 4):
        self.bucket_bytes = int(bucket_mb * (1 << 20))
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self._active = self.world > 1 or (_FORCE and dist.is_initialized())
        self._side = torch.cuda.Stream() if torch.cuda.is_available() and self._active else None
        self.phases = max(1, int(phases))
        self.attached = False
        self.stats = {"calls": 0, "phases": 0, "collectives": 0, "bytes": 0}
        self._once_ptrs = set()      # data_ptr of every parameter of a network registered with once()

    # ---- in-backward form -------------------------------------------------------------------------------------------
    def attach(self) -> "GradSync":
        """Average gradients inside every native backward of this package's networks (see the module docstring).  The hook table
        is process-wide: a second GradSync cannot be attached while another one is."""
        from . import model
        other = [h for h in model._phase_hooks.values() if h is not self]
        if other:
            raise RuntimeError("srcgan_amd.dist: another GradSync is attached; detach() it first (the backward hooks are process-wide)")
        for kind in ("rddb", "nlayerd", "resdeconv", "srnet"):
            model._phase_hooks[kind] = self
        self.attached = True
        return self

    def once(self, *modules) -> "GradSync":
        """Networks that run several times per optimiser step: their native backward calls do not reduce; ``sync(params)`` reduces
        the accumulated gradient once.  (Parameters are identified by address: call again after moving a module.)"""
        for m in modules:
            if m is not None:
                self._once_ptrs.update(p.data_ptr() for p in m.parameters())
        return self

    def _deferred(self, params) -> bool:
        return len(params) > 0 and params[0].data_ptr() in self._once_ptrs

    def detach(self) -> None:
        from . import model
        for kind, h in list(model._phase_hooks.items()):
            if h is self:
                del model._phase_hooks[kind]
        self.attached = False

    def cuts(self, cfg, nrr: int, params=()) -> List[int]:
        """Phase boundaries (RRDB indices) of a generator backward: ``phases`` near-equal RRDB ranges.  Only the plain RDDBNet /
        RDDBNetA parameter order (conv_first, [down], RRDBs, trunk_conv, up-sampler, conv_last) is phased."""
        if not self._active or cfg.legacy != 0 or nrr < 2 or self._deferred(params):
            return [0]
        k = min(self.phases, nrr)
        return sorted({(nrr * j) // k for j in range(k)})

    @torch.no_grad()
    def phase_done(self, arena, params, cfg, lo: int, hi: int, nrr: int) -> None:
        """The native call that finalised the gradients of RRDBs [lo, hi) (+ tail if hi == nrr, + head if lo == 0; everything when
        nrr == 0) has been queued on the current stream: reduce that slice of the arena on the side stream, in place."""
        if not self._active or arena is None or arena.flat.numel() == 0 or self._deferred(params):
            return
        n = len(params)
        if nrr > 0 and not (lo == 0 and hi == nrr):
            ndn = 0
            d = int(getattr(cfg, "down", 0))
            while d > 1:
                ndn, d = ndn + 1, d >> 1
            p_rdb0 = 2 + 2 * ndn
            i0 = p_rdb0 + 30 * lo if lo > 0 else 0
            i1 = n if hi == nrr else p_rdb0 + 30 * hi
        else:
            i0, i1 = 0, n
        a, b = arena.offsets[i0], arena.offsets[i1]
        last = (lo == 0)
        flat = arena.flat
        cuda = flat.is_cuda
        main = torch.cuda.current_stream(flat.device) if cuda else None
        if cuda:
            self._side.wait_stream(main)                      # the phase's kernels
        self.stats["phases"] += 1
        with self._ctx(cuda):
            step = max(1, self.bucket_bytes // 4)
            for o in range(a, b, step):
                piece = flat[o:min(o + step, b)]
                dist.all_reduce(piece, op=dist.ReduceOp.SUM, group=self.group)
                piece.mul_(1.0 / self.world)
                self.stats["collectives"] += 1
                self.stats["bytes"] += piece.numel() * 4
        if last:
            self.stats["calls"] += 1
            if cuda:
                main.wait_stream(self._side)                  # autograd / the optimiser see averaged gradients
                flat.record_stream(self._side)

    # ---- once per optimiser step ------------------------------------------------------------------------------------
    @torch.no_grad()
    def sync(self, params: Iterable[torch.nn.Parameter]) -> None:
        """Make the ``.grad``s of ``params`` the mean over replicas, whatever is still missing: nothing for networks whose backward
        already averaged them (attached, one call per step); ONE reduce of the accumulated gradient for the others -- in place in
        the flat arena the gradients alias when they do (model._GradArena: autograd adopts the first call's views and accumulates
        later calls into them), through flattened buckets otherwise."""
        if not self._active:
            return
        todo = [p for p in params if p.grad is not None and not (self.attached and p.data_ptr() not in self._once_ptrs)]
        if not todo:
            return
        groups = {}
        for p in todo:
            g = p.grad
            groups.setdefault(g.untyped_storage().data_ptr() if g.is_contiguous() else -id(g), []).append(p)
        rest, flats = [], []
        for key, ps in groups.items():
            gs = [p.grad for p in ps]
            lo = min(g.storage_offset() for g in gs)
            hi = max(g.storage_offset() + g.numel() for g in gs)
            if key < 0 or len(ps) < 2 or sum(g.numel() for g in gs) != hi - lo or any(g.dtype != gs[0].dtype for g in gs):
                rest += ps                                   # not one gap-free arena: flatten
            else:
                flats.append(torch.empty(0, dtype=gs[0].dtype, device=gs[0].device).set_(gs[0].untyped_storage(), lo, (hi - lo,), (1,)))
        for flat in flats:
            cuda = flat.is_cuda
            main = torch.cuda.current_stream(flat.device) if cuda else None
            if cuda:
                self._side.wait_stream(main)
            with self._ctx(cuda):
                step = max(1, self.bucket_bytes // flat.element_size())
                for o in range(0, flat.numel(), step):
                    piece = flat[o:o + step]
                    dist.all_reduce(piece, op=dist.ReduceOp.SUM, group=self.group)
                    piece.mul_(1.0 / self.world)
                    self.stats["collectives"] += 1
                    self.stats["bytes"] += piece.numel() * piece.element_size()
            if cuda:
                main.wait_stream(self._side)
                flat.record_stream(self._side)
        if rest:
            before = sum(p.grad.numel() * p.grad.element_size() for p in rest)
            self.allreduce(rest)
            self.stats["bytes"] += before
        self.stats["calls"] += 1

    def _buckets(self, grads: List[torch.Tensor]) -> List[List[torch.Tensor]]:
        out, cur, size = [], [], 0
        for g in grads:
            nb = g.numel() * g.element_size()
            if cur and size + nb > self.bucket_bytes:
                out.append(cur)
                cur, size = [], 0
            cur.append(g)
            size += nb
        if cur:
            out.append(cur)
        return out

    def _ctx(self, cuda: bool):
        if cuda:
            return torch.cuda.stream(self._side)
        import contextlib
        return contextlib.nullcontext()

    @torch.no_grad()
    def begin(self, params: Iterable[torch.nn.Parameter]):
        """Launch the bucketed all-reduces of the existing ``.grad``s on the side stream and return a handle for ``end``.
        The caller may keep computing on the main stream (anything that does not touch these gradients) in between: the
        generator's 66 MB reduce rides under the discriminator step (train.PairedSRGAN)."""
        if not self._active:
            return None
        grads = [p.grad for p in reversed(list(params)) if p.grad is not None]
        if not grads:
            return None
        cuda = grads[0].is_cuda
        if cuda:
            self._side.wait_stream(torch.cuda.current_stream())      # the gradients are complete
        pending = []
        with self._ctx(cuda):
            # per bucket: one flatten, one collective, one scale of the flat buffer and ONE multi-tensor copy back (a copy_ and
            # a mul_ per tensor were ~1400 launches = 5.8 ms of a 148 ms step for the 700 generator tensors)
            for bucket in self._buckets(grads):
                flat = torch.cat([g.reshape(-1) for g in bucket])
                work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                pending.append((work, flat, bucket))
        return (cuda, pending)

    @torch.no_grad()
    def end(self, handle) -> None:
        """Wait for the reduces of ``begin``, write the averaged gradients back, hand the stream back to the optimiser."""
        if handle is None:
            return
        cuda, pending = handle
        inv = 1.0 / self.world
        with self._ctx(cuda):
            for work, flat, bucket in pending:
                work.wait()
                flat.mul_(inv)
                views = [v.view_as(g) for v, g in zip(flat.split([g.numel() for g in bucket]), bucket)]
                torch._foreach_copy_(bucket, views)
        if cuda:
            main = torch.cuda.current_stream()
            main.wait_stream(self._side)
            for _, flat, _ in pending:
                flat.record_stream(main)

    @torch.no_grad()
    def allreduce(self, params: Iterable[torch.nn.Parameter]) -> None:
        self.end(self.begin(params))
