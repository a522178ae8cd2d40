"""Weight gradient of the discriminator's 256->512 4x4 stride-1 layer at the bench size (16 x 128 x 128 -> 127 x 127)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from srcgan_amd import ops
B, H, W, cin, cout = 16, 128, 128, 256, 512
torch.manual_seed(0)
x = (torch.rand(B, H, W, cin, device="cuda") - 0.5).to(torch.bfloat16)
dy = (torch.rand(B, H - 1, W - 1, cout, device="cuda") - 0.5).to(torch.bfloat16)
gw = torch.zeros(cout, cin, 4, 4, device="cuda")
f = lambda: ops.conv_wgrad(dy, x, gw, kh=4, kw=4, Cout=cout, Cin=cin, pad=(1, 1), layout=(cin * 16, 16, 4, 1, 0))
f(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
e0.record()
for _ in range(10): f()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(f"wgrad 4x4 s1 {cin}->{cout}: {ms*1e3:8.1f} us  {2.0*B*(H-1)*(W-1)*16*cin*cout/ms/1e9:7.1f} TFLOP/s")
