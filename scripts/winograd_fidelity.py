"""Stretch item: would Winograd F(2x2, 3x3) be usable for the 64-output-channel dense-block convolutions?
It cuts their MFMA work 2.25x, but the transformed operands V = B^T d B and U = G g G^T must be stored in 16 bits to feed the
matrix cores, and the transforms amplify rounding.  This script measures that on the CPU, against the yardstick the bf16 tests use
(oracle.storage: 16-bit storage, exact arithmetic):
    direct   : x, w rounded to the storage type, exact accumulation, output rounded           (what conv3x3_ls computes)
    winograd : x, w rounded; V and U computed exactly from them, ROUNDED to the storage type; 16 exact batched products;
               exact output transform; output rounded
both compared with the exact convolution of the unrounded operands (relative L2).  Run: python scripts/winograd_fidelity.py"""
import torch
import torch.nn.functional as F

BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)


def rnd(t, dt):
    return t.to(dt).to(torch.float64)


def winograd(x, w, dt):
    """x [B,C,H,W] (H, W even), w [K,C,3,3], pad 1 -> [B,K,H,W]; V and U rounded to dt."""
    B, C, H, W = x.shape
    xp = F.pad(x, (1, 1, 1, 1))
    tiles = xp.unfold(2, 4, 2).unfold(3, 4, 2)                      # [B,C,H/2,W/2,4,4]
    V = rnd(torch.einsum("ij,bcthjk,lk->bcthil", BT, tiles, BT), dt)
    U = rnd(torch.einsum("ij,kcjl,ml->kcim", G, w, G), dt)
    M = torch.einsum("kcim,bcthim->bkthim", U, V)
    Y = torch.einsum("ij,bkthjl,ml->bkthim", AT, M, AT)            # [B,K,H/2,W/2,2,2]
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(B, w.shape[0], H, W)


def rel(a, b):
    return float((a - b).norm() / b.norm())


def main():
    torch.manual_seed(0)
    print(f"{'case':34s} {'storage':8s} {'direct':>9s} {'winograd':>9s} {'ratio':>6s}")
    for cin, cout in ((64, 64), (192, 64), (64, 32)):
        for name, mk in (("uniform(-0.5,0.5) activations", lambda s: torch.rand(s, dtype=torch.float64) - 0.5),
                         ("LeakyReLU(normal) activations", lambda s: F.leaky_relu(torch.randn(s, dtype=torch.float64), 0.2))):
            x = mk((2, cin, 32, 32))
            w = torch.randn(cout, cin, 3, 3, dtype=torch.float64) * (2.0 / (cout * 9)) ** 0.5       # kaiming-normal, fan_out (rddb.py:100-102)
            exact = F.conv2d(x, w, None, 1, 1)
            for dt in (torch.bfloat16, torch.float16):
                xq, wq = rnd(x, dt), rnd(w, dt)
                direct = rnd(F.conv2d(xq, wq, None, 1, 1), dt)
                wino = rnd(winograd(xq, wq, dt), dt)
                ed, ew = rel(direct, exact), rel(wino, exact)
                print(f"{cin:3d}->{cout:2d} {name:26s} {str(dt)[6:]:8s} {ed:9.2e} {ew:9.2e} {ew / ed:6.2f}")


if __name__ == "__main__":
    main()
