#!/usr/bin/env python3
"""Headline benchmark: paired SR-GAN training images/sec at x4, 256->1024 (BASELINE.json configs[1],
"Sat2Aerx4": RDDBNet(3,3,4,nb=23) generator + 3-layer PatchGAN, bf16, batch 16 per MI355X).

One "step" = one paired optimisation step on one synthetic batch resident in HBM:
  G-step {G fwd, D(fake) fwd + dgrad, L1*10 + lsgan, G bwd, Adam} + D-step {D(real), D(fake.detach()) fwd/bwd, Adam}
(SURVEY.md section 8d).  `python bench.py --gpus N --steps K --warmup W`: with N > 1 and no WORLD_SIZE in the
environment this process only LAUNCHES -- it starts N fresh rank processes (one per GPU, RCCL over xGMI, rendezvous on
127.0.0.1) before anything touches a GPU, relays rank 0's JSON line and exits non-zero if a rank fails.  Under
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` (WORLD_SIZE set) it is a rank.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3}     # dense, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0

# algorithmic MACs per image, SURVEY.md section 8d / BASELINE.md section 3
def mac_g(nb, up, cin, cout, P):
    import math
    return P * (576 * cin + 718848 * nb + 36864 + 16384 * sum(4 ** s for s in range(int(math.log2(up)))) + 576 * cout * up * up)


def mac_d3(H):
    return 3072 * (H // 2) ** 2 + 131072 * (H // 4) ** 2 + 524288 * (H // 8) ** 2 + 2097152 * (H // 8 - 1) ** 2 + 8192 * (H // 8 - 2) ** 2


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


def cpu_baseline(nb, lr_hw, up, threads, crop=256):
    """The CPU oracle (restatement of the reference, pinned by golden vectors) timed on this host's cores on a
    BOUNDED sample of the same workload: ONE paired G+D step of the same networks on one crop x crop LR crop
    (a full 256x256 image is (lr_hw/crop)^2 such crops; every layer is a convolution, so work scales with pixels).
    value = full-size images/s implied by the crop time."""
    import oracle
    torch.set_num_threads(threads)
    crop = min(crop, lr_hw)
    st = oracle.make_paired_state(3, 3, up, 64, nb, 32, 64, 3, seed=0)
    g = torch.Generator().manual_seed(1234)
    x = torch.rand(1, 3, crop, crop, generator=g)
    y = torch.rand(1, 3, crop * up, crop * up, generator=g)
    print(f"[bench] cpu_baseline: 1 paired step on a {crop}x{crop} crop, {threads} threads ...", file=sys.stderr, flush=True)
    t0 = time.perf_counter()
    oracle.paired_step(st, x, y)
    dt = time.perf_counter() - t0
    frac = (crop / lr_hw) ** 2
    return {"value": frac / dt, "unit": "images/s", "cores": threads, "kind": "port",
            "sample": f"1 paired G+D step of the same networks (nb={nb}) on one 3x{crop}x{crop}->3x{crop*up}x{crop*up} crop "
                      f"= {frac:.4f} of an image's pixels, fp32 torch CPU oracle, {dt:.1f} s; value = {frac:.4f}/{dt:.1f}s"}


def _ints(text):
    import re
    return [int(t) for t in re.findall(r"(?<![\w.])\d+(?![\w.])", text)]


def pmc_traffic(cls):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (FETCH_SIZE x2 gfx950
    correction + WRITE_SIZE, separate --pmc runs of this same command: scripts/profile_round.sh).  PMC counters
    cannot be read from inside the timed process, so this is the last profiled value or null.
    Matching a profiling class ("wgrad_dense<bf16,MT4,NT2,fast>") to a rocprof kernel name
    ("void wgrad_dense_fast_k<4, 2, 4>(WdP)" or a mangled "_Z..Li4ELi2E..") goes by kernel base name and the leading
    integer template arguments."""
    import glob, re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic.json")))
    if not files:
        return None
    try:
        data = json.load(open(files[-1]))["kernels"]
    except Exception:
        return None
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
            return v["hbm_bytes_per_launch"]
    return None


def launch_ranks(n):
    """Parent of a multi-GPU run started as plain `python bench.py --gpus N`: N child rank processes, one per device.
    The parent never initialises a GPU (no exec of a GPU-holding process, no fork after HIP init): children are fresh
    interpreters with the torchrun environment contract (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT)."""
    import socket
    import subprocess
    if os.environ.get("SRCGAN_LOCAL_DEVICE") is None:        # (rehearsal on one device: tests/test_gpu_dist.py)
        have = torch.cuda.device_count()                      # counting devices does not initialise HIP
        if have < n:
            raise SystemExit(f"bench.py --gpus {n}: only {have} GPU(s) visible")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    import tempfile
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=16, help="images per GPU")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--nb", type=int, default=23)
    ap.add_argument("--lr-size", type=int, default=256)
    ap.add_argument("--up", type=int, default=4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-profile", action="store_true")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args.gpus)

    from srcgan_amd import dist as sdist
    from srcgan_amd import _native as N
    from srcgan_amd.train import PairedSRGAN

    rank, local, world = sdist.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the native path has no CPU fallback")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    torch.manual_seed(0)                         # identical init on every rank, then broadcast anyway
    model = PairedSRGAN(3, 3, args.up, nf=64, nb=args.nb, gc=32, ndf=64, n_layers=3, dtype=args.dtype, device=dev)
    use_dist = world > 1 or dist.is_initialized()          # one rank + SRCGAN_FORCE_DIST=1 rehearses the RCCL path
    if use_dist:
        sdist.broadcast_module(model.netG)
        sdist.broadcast_module(model.netD)
        model.grad_sync = sdist.GradSync()
    B, h, H = args.batch, args.lr_size, args.lr_size * args.up
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    x = torch.rand(B, 3, h, h, generator=g).to(dev)           # synthetic, value range of dataset.py:131
    y = torch.rand(B, 3, H, H, generator=g).to(dev)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        model.optimize_parameters(x, y)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        model.optimize_parameters(x, y)
    barrier()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    loss_g, loss_d = float(model.loss_G), float(model.loss_D)

    # ---- per-kernel roofline: one extra step with HIP events around every conv launch (launch stream)
    roofline, kernels = None, []
    # EVERY rank runs the extra step (it contains the gradient collectives: a rank-0-only step would leave rank 0's
    # all-reduces without partners); only rank 0 brackets its launches with events.
    if not args.no_kernel_profile:
        if rank == 0:
            N.prof_enable(True)
        model.optimize_parameters(x, y)
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
                        "traffic": pmc_traffic(k["cls"]),
                        "algorithmic_bytes_per_launch": k["bytes"] / k["count"],
                        "algorithmic_gbytes_per_s": k["bytes"] / (k["ms"] * 1e-3) / 1e9}
    barrier()

    if rank == 0:
        imgs = world * B * args.steps
        per_img_mac = 3 * mac_g(args.nb, args.up, 3, 3, h * h) + 8 * mac_d3(H)
        out = {
            "metric": "paired SR-CycleGAN train images/sec at x4 256->1024",
            "value": imgs / elapsed, "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"Sat2Aerx{args.up} paired G+D step (BASELINE configs[1]): 23-block RRDB generator + 3-layer PatchGAN, "
                                   f"3x{h}x{h}->3x{H}x{H}, batch {B}/GPU" if args.nb == 23 else
                                   f"Sat2Aerx{args.up} paired G+D step, nb={args.nb}, 3x{h}x{h}->3x{H}x{H}, batch {B}/GPU",
                       "per_gpu_batch": B, "global_batch": B * world, "parallelism": f"dp{world}"},
            "algorithmic_tflop_per_image": 2 * per_img_mac / 1e12,
            "step_tflops": 2 * per_img_mac * imgs / elapsed / 1e12,
            "loss_G": loss_g, "loss_D": loss_d,
            "roofline": roofline,
            "kernels": [{"kernel": k["cls"], "launches": k["count"], "ms": round(k["ms"], 3),
                         "tflops": round(k["flops"] / (k["ms"] * 1e-3) / 1e12, 1)} for k in kernels[:8]],
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.nb, h, args.up, host_cores())
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
