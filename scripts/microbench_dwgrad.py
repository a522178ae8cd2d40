"""Weight gradients of the discriminator's 4x4 layers at the bench size (batch 16): 256->512 s1 @128, 64->128 s2 @512, 128->256 s2 @256."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from srcgan_amd import ops
torch.manual_seed(0)
for (H, cin, cout, s) in ((128, 256, 512, 1), (512, 64, 128, 2), (256, 128, 256, 2)):
    B, W = 16, H
    OH = (H + 2 - 4) // s + 1
    x = (torch.rand(B, H, W, cin, device="cuda") - 0.5).to(torch.bfloat16)
    dy = (torch.rand(B, OH, OH, cout, device="cuda") - 0.5).to(torch.bfloat16)
    gw = torch.zeros(cout, cin, 4, 4, device="cuda")
    f = lambda: ops.conv_wgrad(dy, x, gw, kh=4, kw=4, stride=s, Cout=cout, Cin=cin, pad=(1, 1), layout=(cin * 16, 16, 4, 1, 0))
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"wgrad 4x4 s{s} {cin}->{cout} @{H}: {ms*1e3:8.1f} us  {2.0*B*OH*OH*16*cin*cout/ms/1e9:7.1f} TFLOP/s")
