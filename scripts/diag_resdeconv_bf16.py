import sys; sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch, oracle
from conftest import rel_l2
from srcgan_amd import ResDeconv, MSELoss
torch.manual_seed(3)
net = ResDeconv(1, 3, dtype="bf16").cuda()
sd = {k: p.detach().cpu().clone().requires_grad_(True) for k, p in net.named_parameters()}
x, t = torch.rand(2, 1, 64, 48), torch.rand(2, 3, 64, 48)
yr = oracle.resdeconv_forward(sd, x)
oracle.mse_loss(yr, t).backward()
y = net(x.cuda())
MSELoss()(y, t.cuda()).backward()
print("y", rel_l2(y.cpu(), yr))
errs = sorted(((rel_l2(p.grad.cpu(), sd[k].grad), k) for k, p in net.named_parameters()), reverse=True)
print(errs[:8]); print(errs[len(errs)//2]); print(errs[-3:])
