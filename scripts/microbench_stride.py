"""Does the pixel stride (dense-buffer width) limit the 3x3 conv?  64->32 conv reading a buffer of Cs channels."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from srcgan_amd import ops
B, H, W = 16, 256, 256
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for cin, cout in [(64, 32), (128, 32)]:
    w = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
    wp = ops.pack_conv2d_fwd(w, "bf16"); b = torch.zeros(cout, device="cuda")
    for cs in (cin, cin + 32, 192, 256):
        if cs < cin: continue
        x = (torch.rand(B, H, W, cs, device="cuda") - 0.5).bfloat16()
        for ocs in (32, 192):
            y = torch.zeros(B, H, W, ocs, device="cuda", dtype=torch.bfloat16)
            ms = timeit(lambda: ops.conv_igemm(x, wp, y, kh=3, kw=3, Cin=cin, Cout=cout, pad=(1, 1), bias=b, act=True))
            print(f"{cin}->{cout} in_cs={cs:3d} out_cs={ocs:3d}: {ms*1e3:7.1f} us  {2.0*B*H*W*9*cin*cout/ms/1e9:7.1f} TFLOP/s")
