"""Diagnostic: two default-cascade steps, native vs the CPU oracle restatement, parameter by parameter."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import oracle
from conftest import load_golden, rel_l2
from srcgan_amd import train as T

g = load_golden("cas_default")
opt = T.CasParams(device="cuda", SRModel="ESPCN", CModel="ResDeconv", up=2)
torch.manual_seed(0)
m = T.CasSRC(opt)
sr = {k: p.detach().cpu().clone().requires_grad_(True) for k, p in m.netG_A2C.named_parameters()}
cn = {k: p.detach().cpu().clone().requires_grad_(True) for k, p in m.netG_C2B.named_parameters()}
o_sr, o_c = torch.optim.Adam(sr.values(), lr=1e-4), torch.optim.Adam(cn.values(), lr=1e-4)
realA, realB = torch.from_numpy(g["realA"]), torch.from_numpy(g["realB"])
for step in range(2):
    m.optimize_parameters(realA.cuda(), realB.cuda())
    bc = oracle.rgb_to_gray(realB); ba = oracle.bilinear_down(bc, 2)
    o_sr.zero_grad(); l = oracle.l1_loss(oracle.espcn_forward(sr, ba, 2), bc); l.backward(); o_sr.step()
    o_c.zero_grad(); l2 = oracle.l1_loss(oracle.resdeconv_forward(cn, bc), realB); l2.backward()
    if step == 0:
        gerr = sorted(((rel_l2(p.grad.cpu(), cn[k].grad), k) for k, p in m.netG_C2B.named_parameters()), reverse=True)
        print("step0 grad errs (native vs oracle) worst:", gerr[:5])
    o_c.step()
    errs = sorted(((rel_l2(p.detach().cpu(), cn[k].detach()), k, float((p.detach().cpu() - cn[k].detach()).abs().max())) for k, p in m.netG_C2B.named_parameters()), reverse=True)
    print("step", step, "loss", float(l), float(m.loss_SR), float(l2), float(m.loss_C), "worst param rel_l2:", errs[:4])
with torch.no_grad():
    ra = oracle.bilinear_down(realA, 2)
    fab = oracle.resdeconv_forward(cn, oracle.espcn_forward(sr, ra, 2))
print("fake_AB native vs oracle", rel_l2(m.fake_AB.cpu(), fab), " oracle vs reference", rel_l2(fab, torch.from_numpy(g["fake_AB"])), " native vs reference", rel_l2(m.fake_AB.cpu(), torch.from_numpy(g["fake_AB"])))
