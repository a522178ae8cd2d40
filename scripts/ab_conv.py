"""Interleaved A/B of library variants on the dense-block 3x3 convolutions in their PRODUCTION form, one process
(cdna_hip_programming.md rule 24): blocked dense buffers, LeakyReLU sign masks written by the forward convolutions (class e16)
and read by the gradient-slice convolutions (e8), conv5 / block-input gradient with their residual operands (e1, e3).

    python scripts/ab_conv.py base=srcgan_amd/lib/libsrcgan_amd.so v1=srcgan_amd/lib/variants/v1.so [...]
env: AB_ROUNDS (5), AB_N (20 launches per timing), AB_B (16), AB_SHAPES ("f64,f96,f128,f160,f192,b64,b96,b128,b160,b192,b192r")
Prints per shape and variant the median and the minimum over the rounds (us per launch) and the TFLOP/s of the median."""
import ctypes as C
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from srcgan_amd import _native as N
from srcgan_amd import ops


def load(path):
    h = C.CDLL(os.path.abspath(path))
    for name, (res, args) in N.SIGNATURES.items():
        fn = getattr(h, name)
        fn.restype, fn.argtypes = res, args
    return h


def main():
    variants, dbg = [], {}
    for a in sys.argv[1:]:
        name, path = a.split("=", 1)
        if "@" in path:            # name=path@N: diagnostic build whose kernels read SRCGAN_DBG=N (cached per kernel class at its first launch)
            path, dbg[name] = path.split("@", 1)
        variants.append((name, load(path)))
    assert variants, __doc__
    rounds, nl = int(os.environ.get("AB_ROUNDS", "5")), int(os.environ.get("AB_N", "20"))
    B, H, W = int(os.environ.get("AB_B", "16")), 256, 256
    shapes = os.environ.get("AB_SHAPES", "f64,f96,f128,f160,f192,b64,b96,b128,b160,b192,b192r").split(",")
    dt = "bf16"
    torch.manual_seed(0)
    N._lib = variants[0][1]
    act = (torch.rand(B, H, W, 192, device="cuda") - 0.5).to(torch.bfloat16)      # forward dense buffer
    grd = ((torch.rand(B, H, W, 192, device="cuda") - 0.5) * 0.1).to(torch.bfloat16)   # gradient dense buffer
    A, pl = ops.make_blocked(act)
    G, _ = ops.make_blocked(grd)
    nxt, _ = ops.make_blocked(torch.zeros(B, H, W, 64, device="cuda", dtype=torch.bfloat16))
    sign = torch.randint(-2**31, 2**31 - 1, (4, B, H, W), dtype=torch.int32, device="cuda")
    del act, grd
    calls = {}
    for s in shapes:
        fwd, cin = s[0] == "f", int(s[1:4].rstrip("rn"))
        cout = 64 if cin == 192 else 32
        w = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
        wp = ops.pack_conv2d_fwd(w, dt)
        b = torch.randn(cout, device="cuda") * 0.1
        fl = 2.0 * B * H * W * 9 * cin * cout
        if fwd and cout == 32:          # conv1..conv4: bias + LeakyReLU, writes its slice and the sign mask
            k = (cin - 64) // 32
            f = (lambda wp=wp, b=b, cin=cin, k=k: ops.conv_igemm(A, wp, A, kh=3, kw=3, Cin=cin, Cout=32, y_coff=cin, pad=(1, 1), bias=b, act=True,
                                                                 x_plane=pl, y_plane=pl, shape=(B, H, W), sign_out=sign[k]))
        elif fwd and s.endswith("n"):    # conv5 without its residual operand (what the operand costs: compare with f192)
            f = (lambda wp=wp, b=b: ops.conv_igemm(A, wp, nxt, kh=3, kw=3, Cin=192, Cout=64, pad=(1, 1), bias=b, alpha=0.2,
                                                    x_plane=pl, y_plane=pl, shape=(B, H, W)))
        elif fwd:                        # conv5: 0.2 * (conv + bias) + x  -> channels [0, 64) of the next buffer
            f = (lambda wp=wp, b=b: ops.conv_igemm(A, wp, nxt, kh=3, kw=3, Cin=192, Cout=64, pad=(1, 1), bias=b, alpha=0.2, r1=A, r1_cend=64, beta1=1.0,
                                                    x_plane=pl, y_plane=pl, r1_plane=pl, shape=(B, H, W)))
        elif cout == 32:                 # gradient of slice j from the prefix [dy5 .. dy_{j+1}] (cin channels), times LeakyReLU'(x_j) by sign mask
            j = 4 - (cin - 64) // 32
            f = (lambda wp=wp, cin=cin, j=j: ops.conv_igemm(G, wp, G, kh=3, kw=3, Cin=cin, Cout=32, y_coff=cin, pad=(1, 1), mslope=0.2,
                                                             x_plane=pl, y_plane=pl, shape=(B, H, W), sign_in=sign[j - 1]))
        elif s.endswith("r"):            # block-input gradient of RDB1: + d(out) + RRDB skip (two residual operands)
            f = (lambda wp=wp: ops.conv_igemm(G, wp, nxt, kh=3, kw=3, Cin=192, Cout=64, pad=(1, 1), r1=G, r1_cend=64, beta1=1.0, r2=nxt, r2_cend=64, beta2=1.0,
                                               x_plane=pl, y_plane=pl, r1_plane=pl, r2_plane=pl, shape=(B, H, W)))
        else:                            # block-input gradient with one residual operand
            f = (lambda wp=wp: ops.conv_igemm(G, wp, nxt, kh=3, kw=3, Cin=192, Cout=64, pad=(1, 1), r1=G, r1_cend=64, beta1=1.0,
                                               x_plane=pl, y_plane=pl, r1_plane=pl, shape=(B, H, W)))
        calls[s] = (f, fl)
    times = {(s, v): [] for s in shapes for v, _ in variants}
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    for s in shapes:                     # warm-up: every variant, every shape once
        for v, h in variants:
            N._lib = h
            os.environ["SRCGAN_DBG"] = dbg.get(v, "0")
            calls[s][0]()
    torch.cuda.synchronize()
    for r in range(rounds):
        for s in shapes:
            for v, h in (variants if r % 2 == 0 else variants[::-1]):
                N._lib = h
                f = calls[s][0]
                f()
                e0.record()
                for _ in range(nl):
                    f()
                e1.record()
                torch.cuda.synchronize()
                times[(s, v)].append(e0.elapsed_time(e1) / nl * 1e3)
    print(f"{'shape':8s}" + "".join(f"{v:>26s}" for v, _ in variants))
    tot = {v: 0.0 for v, _ in variants}
    for s in shapes:
        row = f"{s:8s}"
        for v, _ in variants:
            t = times[(s, v)]
            med = statistics.median(t)
            tot[v] += med
            row += f"  {med:7.1f} us (min {min(t):6.1f}) {calls[s][1] / med / 1e6:5.0f}T"
        print(row)
    print(f"{'sum':8s}" + "".join(f"  {tot[v]:7.1f} us{'':20s}" for v, _ in variants))


if __name__ == "__main__":
    main()
