import sys; sys.path.insert(0,'/root/repo')
import torch
from srcgan_amd.train import StackedSR
for dt in ("bf16","fp16"):
    torch.manual_seed(0)
    m = StackedSR(ups=(4,2), nf=64, nb=23, gc=32, dtype=dt, device="cuda")
    g = torch.Generator().manual_seed(1234)
    x = torch.rand(2,3,128,128,generator=g).cuda(); y = torch.rand(2,3,1024,1024,generator=g).cuda()
    o1 = m.nets[0](x); o2 = m.nets[1](o1)
    print(dt, "stage1 max", float(o1.abs().max()), "finite", bool(torch.isfinite(o1).all()), "stage2 max", float(o2.abs().max()), bool(torch.isfinite(o2).all()))
    for S in (1.0, 64.0, 1024.0):
        for p in m.parameters(): p.grad=None
        o2 = m.nets[1](m.nets[0](x))
        (m.criterion(o2,y)*S).backward()
        gm = max(float(p.grad.abs().max()) for p in m.parameters())
        fin = all(bool(torch.isfinite(p.grad).all()) for p in m.parameters())
        print("   scale", S, "max |grad|/S", gm/S, "finite", fin)
