#!/usr/bin/env python3
"""Benchmarks of the MI355X-native SRCGAN training hot path.

Default (`--config paired`) = the headline: paired SR-GAN training images/sec at x4, 256->1024 (BASELINE.json configs[1],
"Sat2Aerx4": RDDBNet(3,3,4,nb=23) generator + 3-layer PatchGAN, bf16, batch 16 per MI355X).  One "step" = one optimisation step
on one synthetic batch resident in HBM:
  G-step {G fwd, D(fake) fwd + dgrad, L1*10 + lsgan, G bwd, Adam} + D-step {D(real), D(fake.detach()) fwd/bwd, Adam}
(SURVEY.md section 8d).  The other BASELINE configurations:
  --config cycle          configs[2]  full cycle (G_A, G_B, D_A, D_B; GAN + cycle + identity), 23-block generators, batch 8/GPU
  --config cas-constlab   configs[3]  cascade-const LAB step (trainCasConstLAB surface: SRDN on L + ResDeconv L->ab), batch 8/GPU
  --config x8             configs[4]  two stacked generators 128->512->1024, 16-bit storage, batch 32/GPU in micro-batches

`python bench.py --gpus N --steps K --warmup W`: with N > 1 and no WORLD_SIZE in the environment this process only LAUNCHES -- it
starts N fresh rank processes (one per GPU, RCCL over xGMI, rendezvous on 127.0.0.1) before anything touches a GPU, relays rank
0's JSON line and exits non-zero if a rank fails.  Under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`
(WORLD_SIZE set) it is a rank.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "fp16": 2500.0, "fp32": 157.3}     # dense, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0


# ---- algorithmic MACs per image, SURVEY.md section 8d / BASELINE.md section 3 (P = LR pixels of the trunk)
def mac_g(nb, up, cin, cout, P):
    return P * (576 * cin + 718848 * nb + 36864 + 16384 * sum(4 ** s for s in range(int(math.log2(up)))) + 576 * cout * up * up)


def mac_gb(nb, down, cin, cout, P):
    """build-defined HR->LR generator (RDDBNetA): conv_first at HR, one 3x3 s2 conv per /2 stage, trunk + trunk_conv + conv_last at LR"""
    n = int(math.log2(down))
    return P * (576 * cin * down * down + 36864 * sum(4 ** s for s in range(n)) + 718848 * nb + 36864 + 576 * cout)


def mac_d3(H):
    return 3072 * (H // 2) ** 2 + 131072 * (H // 4) ** 2 + 524288 * (H // 8) ** 2 + 2097152 * (H // 8 - 1) ** 2 + 8192 * (H // 8 - 2) ** 2


def mac_srdn(nb, cin, cout, P):
    """srdn.py:56-74: conv_first, 2 x nb RRDBs, conv_last (trunk_conv is never applied)"""
    return P * (576 * cin + 718848 * 2 * nb + 576 * cout)


def mac_resdeconv(tar, P):
    """resdeconv.py:99-195 forward MACs: 1.7045e10 at 256x256 with 3 output channels (hooks on the reference)"""
    return 1.7045e10 * P / 65536.0 - 576 * (3 - tar) * P


def host_cores():
    """CPU cores this process may really use: scheduler affinity, capped by the cgroup CPU quota (a GPU box hands
    each job a share of a much larger host; torch with one thread per *host* core thrashes)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    n = min(n, max(1, q // int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())))
            break
        except Exception:
            pass
    return max(1, min(n, int(os.environ.get("SRCGAN_BENCH_CPU_THREADS", "16"))))


def _timed(fn, warmup, steps):
    for _ in range(warmup):
        fn()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    return (time.perf_counter() - t0) / steps


def cpu_baseline(config, nb, lr_hw, up, threads):
    """The CPU oracle (restatement of the reference, pinned by golden vectors) timed on this host's cores, as BASELINE.md section 4
    specifies: (1) configs[0] "Sat2Aerx2" EXACTLY -- RDDBNet(3,3,2,nb=1) + 3-layer PatchGAN, batch 2, 3x128x128 -> 3x256x256, fp32,
    3 warm-up + 5 timed paired steps; (2) a BOUNDED sample of the benchmarked configuration -- one image (or crop) through the
    same networks, 1 warm-up + 2 timed steps -- scaled by its share of an image's pixels (every layer is a convolution: work is
    proportional to pixels).  `value` is (2), the figure comparable with the GPU line; `c1` carries (1)."""
    import oracle
    torch.set_num_threads(threads)
    g = torch.Generator().manual_seed(1234)
    print(f"[bench] cpu_baseline: configs[0] (B=2, 128->256, nb=1), 3 warm-up + 5 timed steps, {threads} threads ...", file=sys.stderr, flush=True)
    st1 = oracle.make_paired_state(3, 3, 2, 64, 1, 32, 64, 3, seed=0)
    x1, y1 = torch.rand(2, 3, 128, 128, generator=g), torch.rand(2, 3, 256, 256, generator=g)
    dt1 = _timed(lambda: oracle.paired_step(st1, x1, y1), 3, 5)
    c1 = {"value": 2.0 / dt1, "unit": "images/s", "ms_per_step": 1e3 * dt1, "steps": 5, "warmup": 3,
          "workload": "BASELINE configs[0] Sat2Aerx2: RDDBNet(3,3,2,nb=1) + NLayerDiscriminator(3,64,3), batch 2, 3x128x128->3x256x256, fp32"}
    if config == "paired":
        crop = min(256, lr_hw)
        st = oracle.make_paired_state(3, 3, up, 64, nb, 32, 64, 3, seed=0)
        x, y = torch.rand(1, 3, crop, crop, generator=g), torch.rand(1, 3, crop * up, crop * up, generator=g)
        frac = (crop / lr_hw) ** 2
        what = f"paired G+D step of the same networks (nb={nb}) on one 3x{crop}x{crop}->3x{crop * up}x{crop * up} image"
        fn = lambda: oracle.paired_step(st, x, y)
    elif config == "cycle":
        crop = min(64, lr_hw)
        st = oracle.make_cycle_state(up, 64, nb, 32, 64, 3, seed=0)
        x, y = torch.rand(1, 3, crop, crop, generator=g), torch.rand(1, 3, crop * up, crop * up, generator=g)
        frac = (crop / lr_hw) ** 2
        what = f"full cycle step (nb={nb}) on one 3x{crop}x{crop} / 3x{crop * up}x{crop * up} crop pair"
        fn = lambda: oracle.cycle_step(st, x, y)
    elif config == "cas-constlab":
        crop = min(128, lr_hw * up)
        frac = (crop / (lr_hw * up)) ** 2
        what = f"cascade-const LAB step (SRDN nb={nb} + ResDeconv(1,2)) on one {crop}x{crop} LAB crop"
        fn = _cpu_cas_constlab(oracle, nb, up, crop, g)
    else:
        crop = min(64, lr_hw)
        s1 = {k: v.clone().requires_grad_(True) for k, v in oracle.rddbnet_state(3, 3, 4, 64, nb, 32, seed=0).items()}
        s2 = {k: v.clone().requires_grad_(True) for k, v in oracle.rddbnet_state(3, 3, 2, 64, nb, 32, seed=1).items()}
        opt = torch.optim.Adam(list(s1.values()) + list(s2.values()), lr=1e-4)
        x, y = torch.rand(1, 3, crop, crop, generator=g), torch.rand(1, 3, crop * 8, crop * 8, generator=g)
        frac = (crop / lr_hw) ** 2
        what = f"stacked x4 + x2 generator step (nb={nb}) on one 3x{crop}x{crop}->3x{crop * 8}x{crop * 8} crop"

        def fn():
            opt.zero_grad()
            oracle.l1_loss(oracle.rddbnet_forward(s2, oracle.rddbnet_forward(s1, x, 4), 2), y).backward()
            opt.step()
    print(f"[bench] cpu_baseline: {what}, 1 warm-up + 2 timed steps ...", file=sys.stderr, flush=True)
    dt = _timed(fn, 1, 2)
    return {"value": frac / dt, "unit": "images/s", "cores": threads, "kind": "port",
            "sample": f"{what} = {frac:.4f} of an image's pixels, fp32 torch CPU oracle, 1 warm-up + 2 timed steps, {dt:.2f} s/step; "
                      f"value = {frac:.4f}/{dt:.2f}s", "c1": c1}


def _cpu_cas_constlab(oracle, nb, up, crop, g):
    """one CasSRCConstLAB.optimize_parameters (trainCasConstLAB.py:82-153) on the oracle's functions"""
    import torch.nn.functional as F
    sr = {k: v.clone().requires_grad_(True) for k, v in _srdn_state(oracle, nb).items()}
    from srcgan_amd.model import ResDeconv          # parameter shapes / default initialisation only (CPU module holder)
    torch.manual_seed(0)
    cm = {k: p.detach().clone().requires_grad_(True) for k, p in ResDeconv(1, 2).named_parameters()}
    o1, o2 = torch.optim.Adam(list(sr.values()), lr=1e-4), torch.optim.Adam(list(cm.values()), lr=1e-4)
    lab = torch.rand(1, 3, crop, crop, generator=g)
    gray = torch.rand(1, 1, crop, crop, generator=g)

    def fn():
        L, ab = lab[:, :1], lab[:, 1:]
        blur = F.interpolate(F.interpolate(L, scale_factor=1.0 / up, mode="bilinear", align_corners=False), scale_factor=up, mode="bilinear", align_corners=False)
        o1.zero_grad(); oracle.l1_loss(oracle.srdn_forward(sr, blur), L).backward(); o1.step()
        o2.zero_grad(); oracle.l1_loss(oracle.resdeconv_forward(cm, L), ab).backward(); o2.step()
        with torch.no_grad():
            oracle.resdeconv_forward(cm, oracle.srdn_forward(sr, gray))
    return fn


def _srdn_state(oracle, nb):
    """SRDN(1,1,up,nb) weights with the reference's distributions (srdn.py:56-66: two RRDB stacks, kaiming-normal convolutions)"""
    enc = oracle.rddbnet_state(1, 1, 1, 64, nb, 32, seed=0)
    dec = oracle.rddbnet_state(1, 1, 1, 64, nb, 32, seed=1)
    sd = {}
    for k, v in enc.items():
        sd[k.replace("RRDB_trunk.", "RRDB_encoder.")] = v
    for k, v in dec.items():
        if k.startswith("RRDB_trunk."):
            sd[k.replace("RRDB_trunk.", "RRDB_decoder.")] = v
    return sd


def _ints(text):
    import re
    return [int(t) for t in re.findall(r"(?<![\w.])\d+(?![\w.])", text)]


def pmc_traffic(cls, config="paired"):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (FETCH_SIZE x2 gfx950
    correction + WRITE_SIZE, separate --pmc runs of this same command: scripts/profile_round.sh).  PMC counters
    cannot be read from inside the timed process, so this is the last profiled value or null.
    Matching a profiling class ("wgrad_dense<bf16,MT4,NT2,fast>") to a rocprof kernel name
    ("void wgrad_dense_fast_k<4, 2, 4>(WdP)" or a mangled "_Z..Li4ELi2E..") goes by kernel base name and the leading
    integer template arguments."""
    import glob, re
    # the profile of THIS configuration, latest round (scripts/profile_round.sh <round> [config])
    pat = re.compile(r"r\d+" + ("" if config == "paired" else "_" + re.escape(config)) + r"_hbm_traffic\.json$")
    files = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic.json")) if pat.search(os.path.basename(f)))
    pmc_traffic.source, pmc_traffic.avg_us = None, None
    if not files:
        return None
    try:
        data = json.load(open(files[-1]))["kernels"]
    except Exception:
        return None
    pmc_traffic.source = os.path.relpath(files[-1], ROOT)
    m = re.match(r"(\w+)<(\w+),(.*)>", cls)
    if not m:
        return None
    base = m.group(1) + ("_fast" if ",fast" in cls else "") + "_k"
    want = _ints(re.sub(r"[A-Za-z]+(?=\d)", " ", m.group(3).replace("W8+", "W").replace("x", " ")))
    for name, v in data.items():
        if base not in name:
            continue
        if name.startswith("_Z"):
            have = [int(t) for t in re.findall(r"Li(\d+)E", name)]
        else:
            a = re.search(r"<(.*)>\(", name)
            have = _ints(a.group(1)) if a else []
        if have[:len(want)] == want or (want and not have):
            pmc_traffic.avg_us = v.get("avg_us")          # the kernel's duration in that profile: compare with the live avg_ms (a stale profile shows)
            return v["hbm_bytes_per_launch"]
    return None


def launch_ranks(n):
    """Parent of a multi-GPU run started as plain `python bench.py --gpus N`: N child rank processes, one per device.
    The parent never initialises a GPU (no exec of a GPU-holding process, no fork after HIP init): children are fresh
    interpreters with the torchrun environment contract (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT)."""
    import socket
    import subprocess
    import tempfile
    if os.environ.get("SRCGAN_LOCAL_DEVICE") is None:        # (rehearsal on one device: tests/test_gpu_dist.py)
        from srcgan_amd.dist import visible_gpu_count
        have = visible_gpu_count()                            # visibility variables / the KFD topology in sysfs: the HIP runtime stays closed
        if 0 <= have < n:                                     # (-1: no topology readable here -- the ranks find out)
            raise SystemExit(f"bench.py --gpus {n}: only {have} GPU(s) visible")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs, rc = [], 0
    with tempfile.TemporaryFile(mode="w+") as cap:               # rank 0's stdout (a pipe would need a reader thread)
        try:
            for r in range(n):
                env = dict(os.environ)
                env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                           MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
                procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], env=env, cwd=ROOT,
                                              stdout=cap if r == 0 else subprocess.DEVNULL))
            while any(p.poll() is None for p in procs):
                if any(p.poll() not in (None, 0) for p in procs):    # a failed rank leaves its peers waiting in a collective
                    break
                time.sleep(0.1)
            rc = next((p.returncode for p in procs if p.returncode not in (None, 0)), 0)
        finally:
            for p in procs:                                      # end exactly the processes started here
                if p.poll() is None:
                    p.kill()
                    p.wait()
        cap.seek(0)
        out0 = cap.read()
    lines = [l for l in out0.splitlines() if l.startswith("{")]
    if rc != 0 or len(lines) != 1:
        sys.stdout.write(out0)
        raise SystemExit(rc or 1)
    print(lines[0], flush=True)


# ---- workloads ------------------------------------------------------------------------------------------------------------------
DEFAULTS = {"paired": dict(batch=16, nb=23, lr_size=256, up=4), "cycle": dict(batch=8, nb=23, lr_size=256, up=4),
            "cas-constlab": dict(batch=8, nb=3, lr_size=256, up=4), "x8": dict(batch=32, nb=23, lr_size=128, up=8)}


def build(args, dev, rank):
    """-> (step function, modules to broadcast, harness, per-image MACs, workload text, loss getter)"""
    from srcgan_amd import train as T
    B, h, up, nb, dt = args.batch, args.lr_size, args.up, args.nb, args.dtype
    H = h * up
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    rnd = lambda *s: torch.rand(*s, generator=g).to(dev)       # synthetic, value range of dataset.py:131
    if args.config == "paired":
        m = T.PairedSRGAN(3, 3, up, nf=64, nb=nb, gc=32, ndf=64, n_layers=3, dtype=dt, device=dev)
        x, y = rnd(B, 3, h, h), rnd(B, 3, H, H)
        mac = 3 * mac_g(nb, up, 3, 3, h * h) + 8 * mac_d3(H)
        exact = (B, nb, h, up) == (16, 23, 256, 4)
        text = (f"Sat2Aerx{up} paired G+D step{' (BASELINE configs[1])' if exact else ''}: {nb}-block RRDB generator + 3-layer PatchGAN, "
                f"3x{h}x{h}->3x{H}x{H}, batch {B}/GPU")
        return (lambda: m.optimize_parameters(x, y)), [m.netG, m.netD], m, mac, text, lambda: {"loss_G": float(m.loss_G.detach()), "loss_D": float(m.loss_D.detach())}
    if args.config == "cycle":
        o = T.CycleParams(device=dev)
        o.mode, o.nb, o.n_layers, o.dtype, o.batch_size = f"x{up}", nb, 3, dt, B
        m = T.SRCycleGAN(o)
        a, b = rnd(B, 3, h, h), rnd(B, 3, H, H)
        mac = 9 * (mac_g(nb, up, 3, 3, h * h) + mac_gb(nb, up, 3, 3, h * h)) + 8 * mac_d3(H) + 8 * mac_d3(h)
        exact = (B, nb, h, up) == (8, 23, 256, 4)
        text = (f"Sat2Aerx{up} full cycle step{' (BASELINE configs[2])' if exact else ''}: G_A = RDDBNet(3,3,{up},nb={nb}), G_B = build-defined HR->LR mirror "
                f"(RDDBNetA), D_A / D_B = 3-layer PatchGAN on {H}x{H} / {h}x{h}; GAN + cycle + identity losses, image pools; batch {B}/GPU")
        nets = [m.netG_A, m.netG_B, m.netD_A, m.netD_B]
        return (lambda: m.optimize_parameters(a, b)), nets, m, mac, text, lambda: {"loss_G": float(m.loss_G.detach()), "loss_D": float((m.loss_D_A + m.loss_D_B).detach())}
    if args.config == "cas-constlab":
        o = T.CasParams(device=dev, SRModel="SRDN", CModel="ResDeconv", up=up)
        o.dtype = dt
        from srcgan_amd import _native as N
        N.set_default_dtype(dt)
        m = T.CasSRCConstLAB(o)
        lab, gray = rnd(B, 3, H, H), rnd(B, 1, H, H)
        nbm = len(m.netG_A2C.RRDB_encoder)
        mac = 4 * mac_srdn(nbm, 1, 1, H * H) + 4 * mac_resdeconv(2, H * H)
        exact = (B, h, up) == (8, 256, 4)
        text = (f"Sat2Aerx{up} cascade-const LAB step{' (BASELINE configs[3])' if exact else ''} (trainCasConstLAB.py surface): SRDN(1,1,{up}) [2x{nbm} RRDBs] on the blurred L "
                f"channel + ResDeconv(1,2) L->ab, two L1 losses, two Adam steps, plus the two eval-mode transfer passes; {H}x{H} LAB tiles, batch {B}/GPU")
        nets = [m.netG_A2C, m.netG_C2B]
        return (lambda: m.optimize_parameters(gray, lab)), nets, m, mac, text, lambda: {"loss_SR": float(m.loss_SR.detach()), "loss_C": float(m.loss_C.detach())}
    # x8: 128 -> 512 -> 1024
    m = T.StackedSR(ups=(4, 2), nf=64, nb=nb, gc=32, dtype=dt, device=dev, micro_batch=args.micro_batch or None,
                    loss_scale=args.loss_scale)
    if args.init_scale != 1.0:
        # An UNTRAINED 23-block stack amplifies: with the reference's initialisation (rddb.py:100-105) stage 1 turns [0,1) inputs into
        # values up to ~500 and stage 2 into ~3e5 -- beyond half precision's 65504 (bf16 holds it).  ESRGAN's own recipe scales
        # the initial convolution weights by 0.1; the fp16 run uses that (timing is data-independent as long as values stay finite).
        with torch.no_grad():
            for net in m.nets:
                for name, p in net.named_parameters():
                    if name.endswith("weight") and p.dim() == 4 and "RRDB_trunk" in name:
                        p.mul_(args.init_scale)
    x, y = rnd(B, 3, h, h), rnd(B, 3, 8 * h, 8 * h)
    mac = 3 * (mac_g(nb, 4, 3, 3, h * h) + mac_g(nb, 2, 3, 3, 16 * h * h))
    exact = (B, nb, h) == (32, 23, 128)
    text = (f"Sat2Aerx8 stress{' (BASELINE configs[4])' if exact else ''}: RDDBNet(3,3,4,nb={nb}) {h}->{4 * h} feeding RDDBNet(3,3,2,nb={nb}) {4 * h}->{8 * h}, L1 on the "
            f"{8 * h}x{8 * h} output, one Adam step; batch {B}/GPU in micro-batches of {args.micro_batch or B} (gradient accumulation: the generator has no "
            f"cross-sample coupling, so it is the same step)"
            + (f"; loss scale {args.loss_scale:g} (dynamic), trunk convolution weights initialised at {args.init_scale:g} x the reference's scale "
               f"(an untrained 23-block stack overflows half precision otherwise)" if dt == "fp16" else ""))
    return (lambda: m.optimize_parameters(x, y)), m.nets, m, mac, text, lambda: {"loss": float(m.loss), "skipped_steps": m.skipped_steps}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="paired", choices=list(DEFAULTS))
    ap.add_argument("--batch", type=int, default=None, help="images per GPU")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16", "fp32"])
    ap.add_argument("--nb", type=int, default=None)
    ap.add_argument("--lr-size", type=int, default=None)
    ap.add_argument("--up", type=int, default=None)
    ap.add_argument("--micro-batch", type=int, default=None, help="x8: images per gradient-accumulation slice (default 16)")
    ap.add_argument("--loss-scale", type=float, default=None, help="x8: loss scale (default 1024 for fp16, 1 otherwise)")
    ap.add_argument("--init-scale", type=float, default=None, help="x8: factor on the trunk convolutions' initial weights (default 0.1 for fp16, 1 otherwise)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-profile", action="store_true")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args.gpus)
    for k, v in DEFAULTS[args.config].items():
        if getattr(args, k) is None:
            setattr(args, k, v)
    if args.config == "x8":
        args.up = 8
        if args.micro_batch is None:
            args.micro_batch = min(16, args.batch)
        if args.loss_scale is None:
            args.loss_scale = 1024.0 if args.dtype == "fp16" else 1.0
        if args.init_scale is None:
            args.init_scale = 0.1 if args.dtype == "fp16" else 1.0

    from srcgan_amd import dist as sdist
    from srcgan_amd import _native as N

    rank, local, world = sdist.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the native path has no CPU fallback")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    torch.manual_seed(0)                         # identical init on every rank, then broadcast anyway
    step, nets, harness, per_img_mac, workload, losses = build(args, dev, rank)
    use_dist = world > 1 or dist.is_initialized()          # one rank + SRCGAN_FORCE_DIST=1 rehearses the RCCL path
    dinfo, gsync = None, None
    if use_dist:
        for net in nets:
            sdist.broadcast_module(net)
        # one-call-per-step networks: averaged inside their backward call (generator: 4 overlapped phases); networks that run
        # several times per step accumulate and exchange once in front of optimizer.step() (harness._once)
        gsync = sdist.GradSync().attach()
        harness.grad_sync = gsync
        dinfo = sdist.dist_info(dev)
    B = args.batch

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    torch.cuda.reset_peak_memory_stats(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    loss_vals = losses()
    peak_gb = torch.cuda.max_memory_allocated(dev) / 1e9

    # ---- per-kernel roofline: one extra step with HIP events around every conv launch (launch stream)
    roofline, kernels = None, []
    # EVERY rank runs the extra step (it contains the gradient collectives: a rank-0-only step would leave rank 0's
    # all-reduces without partners); only rank 0 brackets its launches with events.
    if not args.no_kernel_profile:
        if rank == 0:
            N.prof_enable(True)
        step()
        torch.cuda.synchronize()
        if rank == 0:
            N.prof_enable(False)
            kernels = sorted(N.prof_collect(), key=lambda k: -k["ms"])
        if kernels:
            k = kernels[0]
            ach = k["flops"] / (k["ms"] * 1e-3) / 1e12
            peak = MFMA_PEAK_TFLOPS[args.dtype]
            roofline = {"bound": "mfma", "kernel": k["cls"], "launches": k["count"], "avg_ms": k["ms"] / k["count"],
                        "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                        "traffic": pmc_traffic(k["cls"], args.config),
                        "traffic_source": (getattr(pmc_traffic, "source", None) or "none") + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                                          "command, scripts/profile_round.sh; counters cannot be read inside the timed process)",
                        "traffic_profile_avg_ms": (pmc_traffic.avg_us / 1e3) if getattr(pmc_traffic, "avg_us", None) else None,
                        "algorithmic_bytes_per_launch": k["bytes"] / k["count"],
                        "algorithmic_gbytes_per_s": k["bytes"] / (k["ms"] * 1e-3) / 1e9}
    barrier()

    if rank == 0:
        imgs = world * B * args.steps
        metric = {"paired": "paired SR-CycleGAN train images/sec at x4 256->1024",
                  "cycle": "full-cycle SR-CycleGAN train image pairs/sec at x4 256<->1024",
                  "cas-constlab": "cascade-const LAB train images/sec at 1024x1024",
                  "x8": "stacked x8 generator train images/sec 128->1024"}[args.config]
        out = {
            "metric": metric,
            "value": imgs / elapsed, "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": workload, "per_gpu_batch": B, "global_batch": B * world, "parallelism": f"dp{world}",
                       "gradient_exchange": (f"{dinfo['backend']} all-reduce over {dinfo['world_size']} rank(s) of flat f32 gradient arenas, in place: networks that run "
                                             "once per step inside their native backward (generator: 4 RRDB-range phases on a side stream, overlapped); "
                                             "networks that run several times per step once, accumulated, in front of optimizer.step()") if use_dist else "none (1 rank)"},
            "dist": None if not use_dist else {**dinfo, "grad_sync": dict(gsync.stats), "steps_counted": args.warmup + args.steps + (0 if args.no_kernel_profile else 1),
                                                "param_bytes": sum(p.numel() * 4 for net in nets for p in net.parameters())},
            "algorithmic_tflop_per_image": 2 * per_img_mac / 1e12,
            "step_tflops": 2 * per_img_mac * imgs / elapsed / 1e12,
            "peak_memory_gb": round(peak_gb, 2),
            **loss_vals,
            "roofline": roofline,
            "kernels": [{"kernel": k["cls"], "launches": k["count"], "ms": round(k["ms"], 3),
                         "tflops": round(k["flops"] / (k["ms"] * 1e-3) / 1e12, 1)} for k in kernels[:10]],
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.config, args.nb, args.lr_size, 4 if args.config == "x8" else args.up, host_cores())
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
