"""ResDeconv gradient fidelity in bf16: native vs (a) the f32 oracle, (b) the bf16-storage oracle (oracle.storage)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch, oracle
from conftest import rel_l2
from srcgan_amd import ResDeconv, MSELoss
torch.manual_seed(3)
net = ResDeconv(1, 3, dtype="bf16").cuda()
x, t = torch.rand(2, 1, 64, 48), torch.rand(2, 3, 64, 48)
def ref(store):
    sd = {k: p.detach().cpu().clone().requires_grad_(True) for k, p in net.named_parameters()}
    if store:
        with oracle.storage(torch.bfloat16): yr = oracle.resdeconv_forward(sd, x)
    else: yr = oracle.resdeconv_forward(sd, x)
    oracle.mse_loss(yr, t).backward()
    return yr.detach(), sd
(yr, sd), (ye, se) = ref(False), ref(True)
y = net(x.cuda())
MSELoss()(y, t.cuda()).backward()
print("y vs f32", rel_l2(y.cpu(), yr), "vs bf16-storage", rel_l2(y.cpu(), ye), "format", rel_l2(ye, yr))
rows = [(k, rel_l2(p.grad.cpu(), sd[k].grad), rel_l2(se[k].grad, sd[k].grad), rel_l2(p.grad.cpu(), se[k].grad)) for k, p in net.named_parameters()]
for k, a, b, c in rows:
    if k.endswith("conv1.weight") or k.endswith("bn1.weight") or "deconv" in k or k.startswith("pred"):
        print(f"  {k:28s} native-vs-f32 {a:.3f}  format {b:.3f}  native-vs-bf16storage {c:.3f}")
