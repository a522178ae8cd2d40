"""GPU: the RCCL code path with ONE rank (SRCGAN_FORCE_DIST=1): rendezvous on 127.0.0.1, nccl process group bound to the device,
parameter / buffer broadcast, bucketed asynchronous all-reduce on the side stream, barrier + MAX all-reduce of the bench timing.
The multi-rank arithmetic is covered on CPU (tests/test_dist_gloo.py, gloo, world_size 2); N > 1 GPUs only exist on the driver's
node.  Runs in a child process (a process group must not leak into the test runner)."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _env(port=None):
    port = port or _free_port()
    env = dict(os.environ)
    env.update(SRCGAN_FORCE_DIST="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    return env


def test_bench_one_rank_through_rccl():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "2",
                          "--nb", "1", "--lr-size", "64", "--no-cpu-baseline", "--no-kernel-profile"],
                         env=_env(), cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["scaling"] == "weak"


def test_grad_sync_one_rank_keeps_the_step_identical():
    """With one rank the averaged gradient is the gradient: a step through broadcast + GradSync equals a plain step bit for bit --
    both forms: applied after each backward, and attached (averaging inside the native backward calls, the generator's backward
    split into three RRDB-range phases whose arena slices are reduced in place on the side stream through RCCL)."""
    code = r'''
import torch, sys
sys.path.insert(0, %r)
from srcgan_amd import dist as sdist
from srcgan_amd.train import PairedSRGAN
rank, local, world = sdist.init_from_env()
import torch.distributed as dist
assert dist.is_initialized() and dist.get_backend() == "nccl"
def run(sync):
    torch.manual_seed(0)
    m = PairedSRGAN(3, 3, 2, nf=16, nb=5, gc=8, ndf=16, n_layers=3, dtype="fp32", device="cuda")
    if sync:
        sdist.broadcast_module(m.netG); sdist.broadcast_module(m.netD)
        m.grad_sync = sdist.GradSync(bucket_mb=0.05, phases=3)          # several buckets, three backward phases
        assert m.grad_sync._active
        if sync == "attached":
            m.grad_sync.attach()
    g = torch.Generator().manual_seed(5)
    x, y = torch.rand(2, 3, 32, 32, generator=g).cuda(), torch.rand(2, 3, 64, 64, generator=g).cuda()
    for _ in range(2):
        m.optimize_parameters(x, y)
    torch.cuda.synchronize()
    if sync == "attached":
        st = m.grad_sync.stats
        # per step: one generator backward in 3 phases (averaged inside it) + ONE exchange of the discriminator's accumulated
        # gradient (its real and fake pass only accumulate): every parameter byte crosses the wire exactly once per step
        nbytes = sum(p.numel() * 4 for p in list(m.netG.parameters()) + list(m.netD.parameters()))
        assert st["calls"] == 2 * 2 and st["phases"] == 2 * 3 and st["bytes"] == 2 * nbytes, (st, nbytes)
        m.grad_sync.detach()
    return [p.detach().clone() for p in list(m.netG.parameters()) + list(m.netD.parameters())], float(m.loss_G), float(m.loss_D)
a, b, c = run(False), run(True), run("attached")
assert a[1] == b[1] == c[1] and a[2] == b[2] == c[2], (a[1:], b[1:], c[1:])
assert all(torch.equal(p, q) for p, q in zip(a[0], b[0]))
assert all(torch.equal(p, q) for p, q in zip(a[0], c[0]))          # phased backward + in-place reduce: bit for bit the plain step
dist.barrier(); dist.destroy_process_group()
print("ok")
''' % ROOT
    out = subprocess.run([sys.executable, "-c", code], env=_env(), cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), (out.stdout[-500:], out.stderr[-2000:])


def test_bench_two_ranks_on_one_device_over_gloo():
    """bench.py's whole control flow with world_size 2 (torch.distributed.run, 127.0.0.1): broadcast, the generator reduce under
    the discriminator step, barriers, MAX of the timing, the instrumented extra step on every rank, one JSON line from rank 0.
    Both ranks share the box's one GPU and talk over gloo (rehearsal hooks in srcgan_amd/dist.py); RCCL itself is covered with
    one rank above."""
    env = dict(os.environ)
    env.update(SRCGAN_LOCAL_DEVICE="0", SRCGAN_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "SRCGAN_FORCE_DIST"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--batch", "2", "--nb", "1", "--lr-size", "64"],
                         env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 4 and line["value"] > 0
    assert line["roofline"] is not None and line["cpu_baseline"] is None


@pytest.mark.parametrize("cfg", ["paired", "cycle", "x8"])
def test_bench_gpus2_launches_its_own_ranks(cfg):
    """`python bench.py --gpus 2 [--config ...]` exactly as the driver calls it (no torch.distributed.run, no WORLD_SIZE): the parent
    starts two fresh rank processes before touching a GPU, relays rank 0's single JSON line.  One-device rehearsal over gloo, for
    the default configuration and for the two whose networks run several times per step (cycle: three passes per generator; x8:
    micro-batches) -- the line's `dist` object must show both ranks in the collective and every parameter byte exchanged exactly
    once per step."""
    env = dict(os.environ)
    env.update(SRCGAN_LOCAL_DEVICE="0", SRCGAN_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "SRCGAN_FORCE_DIST"):
        env.pop(k, None)
    extra = {"paired": ["--batch", "2", "--nb", "1", "--lr-size", "64"],
             "cycle": ["--config", "cycle", "--batch", "2", "--nb", "1", "--lr-size", "32"],
             "x8": ["--config", "x8", "--batch", "2", "--nb", "1", "--lr-size", "16", "--micro-batch", "1"]}[cfg]
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", *extra],
                         env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 4 and line["value"] > 0
    assert line["config"]["parallelism"] == "dp2" and line["cpu_baseline"] is None
    d = line["dist"]
    assert d["backend"] == "gloo" and d["world_size"] == 2 and d["ranks_seen"] == 2, d
    assert line["config"]["gradient_exchange"].startswith("gloo all-reduce over 2 rank"), line["config"]
    assert d["grad_sync"]["bytes"] == d["steps_counted"] * d["param_bytes"], d


def test_two_rank_cycle_step_exchanges_each_network_once_and_gives_the_mean():
    """world_size 2 on the one device (gloo), the full cycle harness (three passes per generator, two per discriminator and step):
    with GradSync attached and the harness's networks registered once(), the gradients in front of each optimizer.step() are the
    mean of the two ranks' gradients -- which each rank also computes alone, from both seeded batches -- and the bytes exchanged
    per step are exactly one copy of the parameters."""
    code = r'''
import torch, sys, os, itertools
sys.path.insert(0, %r)
from srcgan_amd import dist as sdist
from srcgan_amd.train import SRCycleGAN, CycleParams
rank, local, world = sdist.init_from_env()
import torch.distributed as dist
assert world == 2 and dist.get_backend() == "gloo"
def batch(r):
    g = torch.Generator().manual_seed(11 + r)
    return torch.rand(2, 3, 16, 16, generator=g).cuda(), torch.rand(2, 3, 32, 32, generator=g).cuda()
opt = CycleParams("cuda"); opt.nf, opt.nb, opt.gc, opt.ndf, opt.n_layers, opt.pool_size, opt.dtype = 16, 2, 8, 16, 2, 0, "fp32"
torch.manual_seed(0)
m = SRCycleGAN(opt)
gen = lambda: list(itertools.chain(m.netG_A.parameters(), m.netG_B.parameters()))
dis = lambda: list(itertools.chain(m.netD_A.parameters(), m.netD_B.parameters()))
def grads(batches, sync):
    for p in gen() + dis(): p.grad = None
    for a, b in batches:
        m.forward(a, b)
        m.set_requires_grad([m.netD_A, m.netD_B], False)
        m.backward_G()
    if sync: m._sync(gen())
    gg = [p.grad.detach().clone() for p in gen()]
    for a, b in batches:
        m.forward(a, b)
        m.set_requires_grad([m.netD_A, m.netD_B], True)
        m.backward_D_A(); m.backward_D_B()
    if sync: m._sync(dis())
    torch.cuda.synchronize()
    return gg + [p.grad.detach().clone() for p in dis()]
ref = [g / 2 for g in grads([batch(0), batch(1)], False)]
gs = sdist.GradSync(bucket_mb=0.05, phases=3).attach()
m.grad_sync = gs
got = grads([batch(rank)], True)
gs.detach()
worst = max(float((a - b).abs().max() / b.abs().max().clamp_min(1e-20)) for a, b in zip(got, ref))
assert worst < 1e-5, worst
nbytes = sum(p.numel() * 4 for p in gen() + dis())
assert gs.stats["bytes"] == nbytes and gs.stats["phases"] == 0 and gs.stats["calls"] == 2, (gs.stats, nbytes)
dist.barrier(); dist.destroy_process_group()
print("ok")
''' % ROOT
    env = dict(os.environ)
    env.update(SRCGAN_LOCAL_DEVICE="0", SRCGAN_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "SRCGAN_FORCE_DIST"):
        env.pop(k, None)
    return _run_two_ranks(code, env)


def test_two_ranks_attached_sync_gives_the_mean_gradient():
    """world_size 2 on the one device (gloo): every rank runs a generator + frozen-discriminator backward on ITS batch with
    GradSync attached (3 phases); the gradients that reach .grad must be the mean of the two ranks' gradients, which each rank
    also computes locally without any collective (both batches are seeded)."""
    code = r'''
import torch, sys, os
sys.path.insert(0, %r)
from srcgan_amd import dist as sdist
from srcgan_amd.train import PairedSRGAN, set_requires_grad
rank, local, world = sdist.init_from_env()
import torch.distributed as dist
assert world == 2 and dist.get_backend() == "gloo"
def batch(r):
    g = torch.Generator().manual_seed(5 + r)
    return torch.rand(2, 3, 32, 32, generator=g).cuda(), torch.rand(2, 3, 64, 64, generator=g).cuda()
def grads(m, xs):
    for p in list(m.netG.parameters()) + list(m.netD.parameters()): p.grad = None
    for x, y in xs:
        # generator step's backward (discriminator frozen) ...
        set_requires_grad(m.netD, False)
        fake = m.netG(x)
        (m.criterionGAN(m.netD(fake), True) + m.criterionL1(fake, y) * 10.0).backward()
        # ... and the discriminator step's (two calls with parameter gradients)
        set_requires_grad(m.netD, True)
        ((m.criterionGAN(m.netD(y), True) + m.criterionGAN(m.netD(fake.detach()), False)) * 0.5).backward()
    torch.cuda.synchronize()
    return [p.grad.detach().clone() for p in list(m.netG.parameters()) + list(m.netD.parameters())]
torch.manual_seed(0)
m = PairedSRGAN(3, 3, 2, nf=16, nb=4, gc=8, ndf=16, n_layers=2, dtype="fp32", device="cuda")
m.netD.eval(); m.netD.train()
ref = [g / 2 for g in grads(m, [batch(0), batch(1)])]          # sum over both ranks' batches / world, no collective
for bn in [mod for mod in m.netD.modules() if isinstance(mod, torch.nn.BatchNorm2d)]:
    bn.reset_running_stats()
sync = sdist.GradSync(bucket_mb=0.05, phases=3).attach()
got = grads(m, [batch(rank)])
sync.detach()
worst = max(float((a - b).abs().max() / b.abs().max().clamp_min(1e-20)) for a, b in zip(got, ref))
assert worst < 1e-5, worst
dist.barrier(); dist.destroy_process_group()
print("ok")
''' % ROOT
    env = dict(os.environ)
    env.update(SRCGAN_LOCAL_DEVICE="0", SRCGAN_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "SRCGAN_FORCE_DIST"):
        env.pop(k, None)
    return _run_two_ranks(code, env)


def _run_two_ranks(code, env):
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:           # (not a *_test.py under the repo: pytest would try to collect it)
        script = os.path.join(tmp, "two_rank_script.py")
        with open(script, "w") as f:
            f.write(code)
        out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                              "--master-port", str(_free_port()), script], env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and out.stdout.count("ok") == 2, (out.stdout[-500:], out.stderr[-3000:])
