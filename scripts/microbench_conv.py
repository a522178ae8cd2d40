"""Micro-benchmark of the RDB conv shapes (B=16, 256x256): forward igemm and wgrad TFLOP/s."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from srcgan_amd import ops

dt = sys.argv[1] if len(sys.argv) > 1 else "bf16"
which = sys.argv[2] if len(sys.argv) > 2 else "all"
blocked = os.environ.get("MB_BLOCKED") == "1"
B, H, W = int(os.environ.get("MB_B", "16")), 256, 256
tdt = torch.bfloat16 if dt == "bf16" else torch.float32
torch.manual_seed(0)
dense = (torch.rand(B, H, W, 192, device="cuda") - 0.5).to(tdt)
out64 = torch.zeros(B, H, W, 64, device="cuda", dtype=tdt)
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
if which in ("all", "igemm"):
    shapes = [(64, 32), (96, 32), (128, 32), (160, 32), (192, 64), (64, 64), (32, 64)]
    if os.environ.get("MB_SHAPES"):                  # e.g. MB_SHAPES="192:64,64:32"
        shapes = [tuple(int(v) for v in t.split(":")) for t in os.environ["MB_SHAPES"].split(",")]
    for cin, cout in shapes:
        w = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
        wp = ops.pack_conv2d_fwd(w, dt)
        b = torch.zeros(cout, device="cuda")
        if blocked and cin + cout <= 192:
            db, pl = ops.make_blocked(dense)
            f = lambda: ops.conv_igemm(db, wp, db, kh=3, kw=3, Cin=cin, Cout=cout, y_coff=cin, pad=(1, 1), bias=b, act=True,
                                       x_plane=pl, y_plane=pl, shape=(B, H, W))
        elif blocked:
            db, pl = ops.make_blocked(dense)
            f = lambda: ops.conv_igemm(db, wp, out64, kh=3, kw=3, Cin=cin, Cout=cout, pad=(1, 1), bias=b, alpha=0.2, r1=db, r1_cend=64, beta1=1.0,
                                       x_plane=pl, r1_plane=pl, shape=(B, H, W))
        elif cin + cout <= 192:
            f = lambda: ops.conv_igemm(dense, wp, dense, kh=3, kw=3, Cin=cin, Cout=cout, y_coff=cin, pad=(1, 1), bias=b, act=True)
        else:
            f = lambda: ops.conv_igemm(dense, wp, out64, kh=3, kw=3, Cin=cin, Cout=cout, pad=(1, 1), bias=b, alpha=0.2, r1=dense, r1_cend=64, beta1=1.0)
        ms = timeit(f)
        fl = 2.0 * B * H * W * 9 * cin * cout
        by = B * H * W * (cin + cout) * dense.element_size()
        print(f"igemm {cin:3d}->{cout:3d}: {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TFLOP/s  {by/ms/1e6:7.0f} GB/s(alg)")
if which in ("all", "wgrad"):
    for cin, cout in [(64, 32), (96, 32), (128, 32), (160, 32), (192, 64)]:
        gw = torch.zeros(cout, cin, 3, 3, device="cuda"); gb = torch.zeros(cout, device="cuda")
        dy = (torch.rand(B, H, W, cout, device="cuda") - 0.5).to(tdt)
        f = lambda: ops.conv_wgrad(dy, dense, gw, kh=3, kw=3, Cout=cout, Cin=cin, pad=(1, 1), layout=(cin * 9, 9, 3, 1, 0), bias_grad=gb)
        ms = timeit(f)
        fl = 2.0 * B * H * W * 9 * cin * cout
        print(f"wgrad {cin:3d}->{cout:3d}: {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TFLOP/s")
