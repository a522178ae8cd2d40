"""Micro-benchmark of the generic implicit-GEMM kernel on the discriminator's shapes (B=16, bf16): the four stride-2 parity
gradients of a 4x4 s2 layer (2x2 s1 sub-convolutions with a strided store), the 4x4 s2 forward and the 4x4 s1 layer.
With a -DSG_IG_DIAG variant (scripts/build_variant.sh ... -DSG_IG_DIAG) SRCGAN_DBG removes one cost at a time: 1 MFMAs, 2 operand loads after the
first chunk, 8 weight loads after the first chunk, 16 LDS writes after the first chunk, 32 fragment reads, 4 epilogue."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from srcgan_amd import ops
B = int(os.environ.get("MB_B", "16"))
bf = torch.bfloat16
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
torch.manual_seed(0)
rows = []
# parity gradient of Conv2d(C, Co, 4, 2, 1): dy [B,h,w,Co] -> one parity of dx [B,2h,2w,C]
for C, Co, h in ((128, 256, 128), (64, 128, 256)):
    dy = (torch.rand(B, h, h, Co, device="cuda") - 0.5).to(bf)
    dx = torch.zeros(B, 2 * h, 2 * h, C, device="cuda", dtype=bf)
    mz = (torch.rand(B, 2 * h, 2 * h, C, device="cuda") - 0.5).to(bf)
    w = torch.randn(C, Co, 2, 2, device="cuda") * 0.05
    wp = ops.pack_conv2d_fwd(w, "bf16")
    for name, m in (("plain", None), ("mask", mz)):
        f = lambda: ops.conv_igemm(dy, wp, dx, kh=2, kw=2, Cin=Co, Cout=C, OH=h, OW=h, pad=(1, 1), os=2, oa=0, ob=0, mz=m)
        ms = timeit(f)
        rows.append((f"parity 2x2 s1 {Co}->{C} @{h}^2 {name}", ms, 2.0 * B * h * h * 4 * Co * C))
for C, Co, H in ((64, 128, 512), (128, 256, 256)):
    x = (torch.rand(B, H, H, C, device="cuda") - 0.5).to(bf)
    y = torch.zeros(B, H // 2, H // 2, Co, device="cuda", dtype=bf)
    wp = ops.pack_conv2d_fwd(torch.randn(Co, C, 4, 4, device="cuda") * 0.05, "bf16")
    f = lambda: ops.conv_igemm(x, wp, y, kh=4, kw=4, stride=2, Cin=C, Cout=Co, pad=(1, 1))
    rows.append((f"fwd 4x4 s2 {C}->{Co} @{H}^2", timeit(f), 2.0 * B * (H // 2) ** 2 * 16 * C * Co))
x = (torch.rand(B, 128, 128, 256, device="cuda") - 0.5).to(bf)
y = torch.zeros(B, 127, 127, 512, device="cuda", dtype=bf)
wp = ops.pack_conv2d_fwd(torch.randn(512, 256, 4, 4, device="cuda") * 0.05, "bf16")
f = lambda: ops.conv_igemm(x, wp, y, kh=4, kw=4, Cin=256, Cout=512, pad=(1, 1))
rows.append(("fwd 4x4 s1 256->512 @128^2", timeit(f), 2.0 * B * 127 * 127 * 16 * 256 * 512))
for name, ms, fl in rows:
    print(f"dbg={os.environ.get('SRCGAN_DBG', '0'):>3s} {name:40s} {ms * 1e3:8.1f} us {fl / ms / 1e9:8.1f} TFLOP/s")
