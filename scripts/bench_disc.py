"""PatchGAN forward + backward at the bench size (B=16, 3x1024x1024, bf16): ms per pass and per-kernel-class times from the
in-library HIP-event profile.  Modes as in the paired step: 'train' (parameter gradients, no dx), 'frozen' (dx only)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from srcgan_amd.model import NLayerDiscriminator
from srcgan_amd import _native as N
from srcgan_amd.train import set_requires_grad

B = int(os.environ.get("MB_B", "16"))
torch.manual_seed(0)
net = NLayerDiscriminator(3, 64, 3, dtype=os.environ.get("MB_DT", "bf16")).to("cuda")
x = torch.rand(B, 3, 1024, 1024, device="cuda")

def one(mode):
    if mode == "frozen":
        set_requires_grad(net, False)
        xi = x.clone().requires_grad_(True)
    else:
        set_requires_grad(net, True)
        xi = x
        for p in net.parameters(): p.grad = None
    net(xi).square().mean().backward()

for mode in ("train", "frozen"):
    for _ in range(2): one(mode)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): one(mode)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    N.prof_enable(True); one(mode); torch.cuda.synchronize(); N.prof_enable(False)
    ks = sorted(N.prof_collect(), key=lambda k: -k["ms"])
    print(f"{mode}: {ms:.2f} ms per fwd+bwd; profiled kernel classes sum {sum(k['ms'] for k in ks):.2f} ms")
    for k in ks:
        print(f"   {k['cls']:40s} x{k['count']:2d} {k['ms']:7.3f} ms  {k['flops'] / (k['ms'] * 1e-3) / 1e12:7.1f} TFLOP/s")
