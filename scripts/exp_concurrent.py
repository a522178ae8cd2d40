"""Experiment (round 3): can the dense-block backward's memory-bound gradient-slice convolutions run BESIDE its MFMA-bound weight
gradient (wgrad_dense) on a second stream, each on a share of the CUs?  One RDB's backward work at the bench size, repeated:
sequential on one stream against two streams with the grids limited to complementary CU shares
(SRCGAN_CONV_CUS / SRCGAN_WD_CUS, diagnostic build: SRCGAN_AMD_LIB=srcgan_amd/lib/variants/conc.so)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from srcgan_amd import ops

B, H, W = 16, 256, 256
torch.manual_seed(0)
A, pl = ops.make_blocked((torch.rand(B, H, W, 192, device="cuda") - 0.5).to(torch.bfloat16))
G, _ = ops.make_blocked(((torch.rand(B, H, W, 192, device="cuda") - 0.5) * 0.1).to(torch.bfloat16))
nxt, _ = ops.make_blocked(torch.zeros(B, H, W, 64, device="cuda", dtype=torch.bfloat16))
sign = torch.randint(-2**31, 2**31 - 1, (4, B, H, W), dtype=torch.int32, device="cuda")
wps = {cin: ops.pack_conv2d_fwd(torch.randn(64 if cin == 192 else 32, cin, 3, 3, device="cuda") * 0.05, "bf16") for cin in (64, 96, 128, 160, 192)}
segs = []
for m in (5, 4, 3, 2, 1):
    g0 = 0 if m == 5 else 64 + (4 - m) * 32
    co, cin = (64 if m == 5 else 32), 64 + (m - 1) * 32
    segs.append((g0, g0 + co, torch.zeros(co, cin, 3, 3, device="cuda"), torch.zeros(co, device="cuda"), cin, 1.0))


def slices():
    for cin in (64, 96, 128, 160):
        j = 4 - (cin - 64) // 32
        ops.conv_igemm(G, wps[cin], G, kh=3, kw=3, Cin=cin, Cout=32, y_coff=cin, pad=(1, 1), mslope=0.2, x_plane=pl, y_plane=pl, shape=(B, H, W), sign_in=sign[j - 1])


def block_in():
    ops.conv_igemm(G, wps[192], nxt, kh=3, kw=3, Cin=192, Cout=64, pad=(1, 1), r1=G, r1_cend=64, beta1=1.0, x_plane=pl, y_plane=pl, r1_plane=pl, shape=(B, H, W))


def wg():
    ops.wgrad_dense(G, A, segs, G=192, Cc=192, dy_plane=pl, x_plane=pl, shape=(B, H, W))


def timed(fn, n=8):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def setcus(conv, wd):
    for k, v in (("SRCGAN_CONV_CUS", conv), ("SRCGAN_WD_CUS", wd)):
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = str(v)


s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def seq():
    slices(); block_in(); wg()


def conc():
    # the slice convolutions on a CU share beside the weight gradient of the previous block; the block-input convolution
    # (MFMA-bound) full width afterwards
    ev = torch.cuda.Event()
    with torch.cuda.stream(s1):
        slices()
        ev.record()
    with torch.cuda.stream(s2):
        wg()
    torch.cuda.current_stream().wait_stream(s1); torch.cuda.current_stream().wait_stream(s2)
    full = (os.environ.get("SRCGAN_CONV_CUS"), os.environ.get("SRCGAN_WD_CUS"))
    setcus(None, None)
    block_in()
    setcus(*full)
    s1.wait_stream(torch.cuda.current_stream()); s2.wait_stream(torch.cuda.current_stream())


for rep in range(2):
    setcus(None, None)
    print(f"sequential, full grids: slices {timed(slices):7.1f} us  block-input {timed(block_in):7.1f} us  wgrad_dense {timed(wg):7.1f} us  all {timed(seq):7.1f} us", flush=True)
    for conv, wd in ((128, 128), (96, 160), (80, 176), (64, 192), (48, 208), (32, 224)):
        setcus(conv, wd)
        ts, tw = timed(slices), timed(wg)
        print(f"conv on {conv:3d} CUs + wgrad on {wd:3d}: alone slices {ts:7.1f} us, wgrad {tw:7.1f} us; concurrent (+ full-width block-input) {timed(conc):7.1f} us", flush=True)
