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
