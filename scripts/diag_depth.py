"""Error growth with generator depth (fp32 and bf16): native vs the f32 oracle (and the bf16-storage oracle)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch, oracle
from conftest import rel_l2, rel_err
from srcgan_amd import RDDBNet, MSELoss
hw = int(sys.argv[1]) if len(sys.argv) > 1 else 64
for nb in (1, 2, 4, 8, 16, 23):
    sd = oracle.rddbnet_state(3, 3, 4, 64, nb, 32, seed=7)
    g = torch.Generator().manual_seed(11)
    x = torch.rand(1, 3, hw, hw, generator=g); t = torch.rand(1, 3, 4 * hw, 4 * hw, generator=g)
    def ref(store, dtype=torch.float32):
        p = {k: v.clone().to(dtype).requires_grad_(True) for k, v in sd.items()}
        xr = x.clone().to(dtype).requires_grad_(True)
        if store:
            with oracle.storage(torch.bfloat16): y = oracle.rddbnet_forward(p, xr, 4)
        else: y = oracle.rddbnet_forward(p, xr, 4)
        oracle.mse_loss(y, t.to(dtype)).backward()
        return y.detach(), xr.grad, {k: v.grad for k, v in p.items()}
    yr, dxr, gr = ref(False)
    y64, dx64, g64 = ref(False, torch.float64)
    print(f"nb={nb}: f32 oracle vs f64 oracle: y {rel_err(yr, y64):.2e} dx {rel_err(dxr, dx64):.2e} (L2 {rel_l2(dxr, dx64):.2e}) worst grad {max(rel_err(gr[k], g64[k]) for k in gr):.2e}")
    for dt in ("fp32", "bf16"):
        net = RDDBNet(3, 3, 4, nf=64, nb=nb, gc=32, dtype=dt); net.load_state_dict(sd); net.cuda()
        xg = x.cuda().requires_grad_(True)
        y = net(xg)
        MSELoss()(y, t.cuda()).backward()
        gw = {k: p.grad.cpu() for k, p in net.named_parameters()}
        w = max((rel_err(gw[k], g64[k]), k) for k in gw)
        print(f"   native {dt} vs f64: y max {rel_err(y.cpu(), y64):.2e} L2 {rel_l2(y.cpu(), y64):.2e} | dx max {rel_err(xg.grad.cpu(), dx64):.2e} L2 {rel_l2(xg.grad.cpu(), dx64):.2e} | worst grad max {w[0]:.2e} {w[1]} L2 {max(rel_l2(gw[k], g64[k]) for k in gw):.2e}")
