"""One PatchGAN forward + backward (input gradient requested, as in the G-step) at the bench size; run under
rocprofv3 --kernel-trace by scripts/trace_disc.sh to list the launches of the last iteration in order."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from srcgan_amd.model import NLayerDiscriminator

B = int(os.environ.get("MB_B", "16"))
torch.manual_seed(0)
net = NLayerDiscriminator(3, 64, 3, dtype="bf16").to("cuda")
x = torch.rand(B, 3, 1024, 1024, device="cuda", requires_grad=True)
for it in range(3):
    y = net(x)
    y.square().mean().backward()
    torch.cuda.synchronize()
print("done")
