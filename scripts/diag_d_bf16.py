"""Discriminator gradient fidelity in bf16: native vs (a) the f32 oracle, (b) the oracle that stores activations and conv weights in
bf16 (oracle.storage: f32 arithmetic and backward).  (b)-vs-(a) is what the FORMAT costs (amplified by the BatchNorm backward
cancellation behind a constant lsgan label); native-vs-(b) is what the kernels add."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch, oracle
from conftest import rel_l2
from srcgan_amd import NLayerDiscriminator, GANLoss
sd = oracle.nlayer_d_state(3, 64, 3, seed=5)
torch.manual_seed(1)
x = torch.rand(2, 3, 96, 128)
def ref(store):
    p = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone()) for k, v in sd.items()}
    xr = x.clone().requires_grad_(True)
    if store:
        with oracle.storage(torch.bfloat16): y = oracle.nlayer_d_forward(p, xr, True)
    else: y = oracle.nlayer_d_forward(p, xr, True)
    oracle.gan_loss(y, False).backward()
    return y.detach(), xr.grad, p
yr, dxr, pr = ref(False); ye, dxe, pe = ref(True)
for dt in ("fp32", "bf16"):
    net = NLayerDiscriminator(3, 64, 3, dtype=dt); net.load_state_dict(sd); net.cuda()
    xg = x.cuda().requires_grad_(True)
    y = net(xg)
    GANLoss("lsgan", device="cuda")(y, False).backward()
    print(f"{dt}: y vs f32 {rel_l2(y.cpu(), yr):.2e} vs bf16-storage {rel_l2(y.cpu(), ye):.2e} | dx {rel_l2(xg.grad.cpu(), dxr):.2e} {rel_l2(xg.grad.cpu(), dxe):.2e}")
    for k, p in net.named_parameters():
        print(f"    {k:18s} native-vs-f32 {rel_l2(p.grad.cpu(), pr[k].grad):.2e}  native-vs-bf16storage {rel_l2(p.grad.cpu(), pe[k].grad):.2e}  format {rel_l2(pe[k].grad, pr[k].grad):.2e}")
