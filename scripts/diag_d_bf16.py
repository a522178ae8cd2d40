import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch, oracle
from conftest import rel_err, rel_l2
from srcgan_amd import NLayerDiscriminator, GANLoss
sd = oracle.nlayer_d_state(3, 64, 3, seed=5)
for dt in ("fp32", "bf16"):
    net = NLayerDiscriminator(3, 64, 3, dtype=dt); net.load_state_dict(sd); net.cuda()
    torch.manual_seed(1)
    x = torch.rand(2, 3, 96, 128)
    ref_sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone()) for k, v in sd.items()}
    xr = x.clone().requires_grad_(True)
    yr = oracle.nlayer_d_forward(ref_sd, xr, True)
    oracle.gan_loss(yr, False).backward()
    xg = x.cuda().requires_grad_(True)
    y = net(xg)
    GANLoss("lsgan", device="cuda")(y, False).backward()
    print(dt, "y", rel_l2(y.cpu(), yr), "dx", rel_l2(xg.grad.cpu(), xr.grad))
    for k, p in net.named_parameters():
        print("   ", k, rel_l2(p.grad.cpu(), ref_sd[k].grad), rel_err(p.grad.cpu(), ref_sd[k].grad))
