"""CPU, 2 processes, gloo: the data-parallel plumbing (parameter broadcast, bucketed gradient mean, unit sharding).
The same code runs over RCCL ('nccl') on the GPUs."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from srcgan_amd import dist as sd
    import srcgan_amd
    r, l, w = sd.init_from_env("gloo")
    assert (r, w) == (rank, world)
    try:
        # 1. broadcast: replicas start from different seeds, end identical to rank 0
        torch.manual_seed(100 + rank)
        net = srcgan_amd.NLayerDiscriminator(3, 16, 2)
        net.model[3].running_mean.fill_(float(rank + 1))
        sd.broadcast_module(net)
        flat = torch.cat([t.detach().float().reshape(-1) for t in list(net.parameters()) + list(net.buffers())])
        gathered = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        assert all(torch.equal(gathered[0], g) for g in gathered)
        # 2. gradient mean over small buckets, frozen parameters skipped
        params = list(net.parameters())
        for i, p in enumerate(params):
            p.grad = torch.full_like(p, float(rank + 1) * (i + 1))
        params[2].grad = None                                   # e.g. a frozen tensor
        sync = sd.GradSync(bucket_mb=0.01)
        assert len(sync._buckets([p.grad for p in params if p.grad is not None])) > 1
        sync.allreduce(params)
        mean = sum(range(1, world + 1)) / world
        for i, p in enumerate(params):
            if i == 2:
                assert p.grad is None
            else:
                assert torch.allclose(p.grad, torch.full_like(p, mean * (i + 1)))
        # 3. weak-scaling shard of independent units covers everything exactly once
        b, e = sd.shard_range(37, rank, world)
        cover = torch.zeros(37)
        cover[b:e] = 1
        dist.all_reduce(cover)
        assert torch.equal(cover, torch.ones(37))
        # 4. the in-backward form: arena slices of generator phases, reduced in place; together they cover the arena exactly once
        import types
        from srcgan_amd.model import _GradArena
        nb = 5
        shapes = [(4, 3), (4,)] + [(2, 2)] * (30 * nb) + [(4, 4), (4,), (3, 3), (1, 3)]
        ps = [torch.nn.Parameter(torch.zeros(s)) for s in shapes]
        arena = _GradArena(ps, [True] * len(ps))
        arena.flat.copy_(torch.arange(arena.flat.numel(), dtype=torch.float32) * (rank + 1))
        cfg = types.SimpleNamespace(legacy=0, down=0)
        gs = sd.GradSync(bucket_mb=0.0001, phases=3)
        cuts = gs.cuts(cfg, nb)
        assert cuts[0] == 0 and len(cuts) == 3, cuts
        hi = nb
        for lo in sorted(cuts, reverse=True):
            gs.phase_done(arena, ps, cfg, lo, hi, nb)
            hi = lo
        want = torch.arange(arena.flat.numel(), dtype=torch.float32) * (sum(range(1, world + 1)) / world)
        assert torch.allclose(arena.flat, want), float((arena.flat - want).abs().max())
        assert gs.stats["calls"] == 1 and gs.stats["phases"] == 3 and gs.stats["bytes"] == arena.flat.numel() * 4, gs.stats
        # 5. a network that runs several times per optimiser step (once()): its backward calls only accumulate, sync() exchanges
        #    the accumulated gradient ONCE, in place in the arena the .grads alias: bytes on the wire == 1 x parameter bytes
        class Net(torch.nn.Module):
            def __init__(self):
                super().__init__()
                self.ps = torch.nn.ParameterList([torch.nn.Parameter(torch.zeros(s)) for s in [(4, 3), (4,), (2, 2), (5,)]])
        net2 = Net()
        ps2 = list(net2.parameters())
        gs2 = sd.GradSync(bucket_mb=0.00002, phases=3).once(net2)          # 5-float pieces: several collectives per arena
        gs2._active = True
        import srcgan_amd.model as smodel
        assert "rddb" not in smodel._phase_hooks
        gs2.attach()
        try:
            sd.GradSync().attach()
            raise AssertionError("a second attach must be refused")
        except RuntimeError:
            pass
        nbytes = sum(p.numel() for p in ps2) * 4
        for call in range(3):                                            # three backward calls of one step
            ar = _GradArena(ps2, [True] * len(ps2))
            ar.flat.copy_(torch.arange(ar.flat.numel(), dtype=torch.float32) + 10.0 * call + rank)
            gs2.phase_done(ar, [p.detach() for p in ps2], types.SimpleNamespace(legacy=0, down=0), 0, 0, 0)      # must not reduce
            for p, v in zip(ps2, ar.views):                              # what AccumulateGrad does: adopt, then add in place
                if p.grad is None:
                    p.grad = v
                else:
                    p.grad += v
        assert gs2.stats["bytes"] == 0 and gs2.stats["collectives"] == 0, gs2.stats
        gs2.sync(ps2)
        tot = sum(torch.arange(nbytes // 4, dtype=torch.float32) + 10.0 * c for c in range(3))
        want2 = tot + 3.0 * (sum(range(world)) / world)
        got2 = torch.cat([p.grad.reshape(-1) for p in ps2])
        assert torch.allclose(got2, want2), float((got2 - want2).abs().max())
        assert gs2.stats["bytes"] == nbytes and gs2.stats["collectives"] > 1 and gs2.stats["calls"] == 1, gs2.stats
        gs2.sync([p for p in ps2 if False])                                # nothing to do
        gs2.detach()
        # ... and the flatten fallback when the gradients are separate allocations
        gs3 = sd.GradSync(bucket_mb=0.0001)
        for i, p in enumerate(ps2):
            p.grad = torch.full_like(p, float(rank + 1) * (i + 1))
        gs3.sync(ps2)
        for i, p in enumerate(ps2):
            assert torch.allclose(p.grad, torch.full_like(p, (sum(range(1, world + 1)) / world) * (i + 1)))
        info = sd.dist_info()
        assert info["backend"] == "gloo" and info["world_size"] == world and info["ranks_seen"] == world, info
        q.put((rank, "ok"))
    except Exception as ex:  # pragma: no cover
        q.put((rank, repr(ex)))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 300)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def test_world_size_one_is_a_noop():
    sys.path.insert(0, ROOT)
    from srcgan_amd import dist as sd
    assert sd.shard_range(10, 0, 1) == (0, 10)
    sync = sd.GradSync()
    p = torch.nn.Parameter(torch.ones(3))
    p.grad = torch.full((3,), 2.0)
    sync.allreduce([p])
    assert torch.equal(p.grad, torch.full((3,), 2.0))
    sync.sync([p])
    assert torch.equal(p.grad, torch.full((3,), 2.0)) and sync.stats["bytes"] == 0
    assert sd.dist_info() == {"backend": None, "world_size": 1, "ranks_seen": 1, "nccl_version": None}


def test_visible_gpu_count_reads_no_runtime(monkeypatch):
    sys.path.insert(0, ROOT)
    from srcgan_amd import dist as sd
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,1,2")
    assert sd.visible_gpu_count() == 3
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "4")
    assert sd.visible_gpu_count() == 1
    monkeypatch.delenv("ROCR_VISIBLE_DEVICES")
    monkeypatch.delenv("CUDA_VISIBLE_DEVICES", raising=False)
    assert sd.visible_gpu_count() >= -1           # sysfs (KFD topology), or -1 where it is not readable: never opens HIP
