"""Micro-benchmark of the dense-block weight-gradient kernel at the headline size (B=16, 256x256, nf=64, gc=32)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from srcgan_amd import ops

B, H, W, nf, gc = int(os.environ.get("MB_B", "16")), 256, 256, 64, 32
Cc = nf + 4 * gc
torch.manual_seed(0)
# MB_DATA: random (default) | const (every element 0.25) | sparse (half of the elements zero, like a ReLU's output) -- the matrix
# pipe's power, hence its clock, follows the toggling of the operand bits
mode = os.environ.get("MB_DATA", "random")
def data():
    t = torch.rand(B, H, W, Cc, device="cuda") - 0.5
    if mode == "const": t = torch.full_like(t, 0.25)
    if mode == "sparse": t = torch.relu(t)
    return t.to(torch.bfloat16)
A, Gd = data(), data()
segs, fl = [], 0.0
for m in (5, 4, 3, 2, 1):
    g0 = 0 if m == 5 else nf + (4 - m) * gc
    co = nf if m == 5 else gc
    cin = nf + (m - 1) * gc
    segs.append((g0, g0 + co, torch.zeros(co, cin, 3, 3, device="cuda"), torch.zeros(co, device="cuda"), cin, 1.0))
    fl += 2.0 * B * H * W * 9 * cin * co
if os.environ.get("MB_BLOCKED") == "1":
    Ab, pl = ops.make_blocked(A)
    Gb, _ = ops.make_blocked(Gd)
    f = lambda: ops.wgrad_dense(Gb, Ab, segs, G=Cc, Cc=Cc, dy_plane=pl, x_plane=pl, shape=(B, H, W))
else:
    f = lambda: ops.wgrad_dense(Gd, A, segs)
f(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
e0.record()
for _ in range(10): f()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(f"wgrad_dense block ({mode} operands): {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TFLOP/s (useful)")
