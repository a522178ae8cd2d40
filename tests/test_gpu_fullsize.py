"""GPU: BASELINE-size checks through size-independent properties (the CPU oracle cannot run these sizes in seconds):
adjoint identities <conv(x),dy> = <x,dgrad(dy)> = <w,wgrad(dy,x)>, crop-equivalence of the full-size generator
against the oracle on a crop, and empty/ragged edge handling."""
import pytest
import torch
import torch.nn.functional as F

import oracle
from conftest import rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from srcgan_amd import ops as o
    return o


@pytest.mark.parametrize("dt,tol", [("fp32", 2e-4), ("bf16", 2e-2)])
def test_adjoint_identities_full_size(ops, dt, tol):
    """RDB conv5 shape at config-2 size (B=16, 256x256, 192->64): forward, dgrad and wgrad must be mutually adjoint."""
    torch.manual_seed(0)
    B, H, W, cin, cout = 16, 256, 256, 192, 64
    tdt = torch.bfloat16 if dt == "bf16" else torch.float32
    x = (torch.rand(B, H, W, cin, device="cuda") - 0.5).to(tdt)
    dy = (torch.rand(B, H, W, cout, device="cuda") - 0.5).to(tdt)
    w = (torch.randn(cout, cin, 3, 3, device="cuda") * 0.05).to(tdt).float()
    y = torch.zeros(B, H, W, cout, device="cuda", dtype=tdt)
    dx = torch.zeros(B, H, W, cin, device="cuda", dtype=tdt)
    gw = torch.zeros(cout, cin, 3, 3, device="cuda")
    ops.conv_igemm(x, ops.pack_conv2d_fwd(w, dt), y, kh=3, kw=3, Cout=cout, pad=(1, 1))
    ops.conv_igemm(dy, ops.pack_conv2d_dgrad_s1(w, dt), dx, kh=3, kw=3, Cout=cin, pad=(1, 1))
    ops.conv_wgrad(dy, x, gw, kh=3, kw=3, Cout=cout, Cin=cin, pad=(1, 1), layout=(cin * 9, 9, 3, 1, 0))
    a = float((y.double() * dy.double()).sum())
    b = float((x.double() * dx.double()).sum())
    c = float((w.double() * gw.double()).sum())
    scale = max(abs(a), 1.0)
    assert abs(a - b) / scale < tol and abs(a - c) / scale < tol, (a, b, c)


def test_full_size_generator_crop_equivalence():
    """3x256x256 -> 3x1024x1024 through the native generator (f32, nb=1) == oracle on a 96x96 crop, compared on the
    region outside the crop's receptive-field margin (19 LR pixels)."""
    from srcgan_amd import RDDBNet
    sd = oracle.rddbnet_state(3, 3, 4, 64, 1, 32, seed=11)
    net = RDDBNet(3, 3, 4, nf=64, nb=1, gc=32, dtype="fp32")
    net.load_state_dict(sd)
    net.cuda()
    torch.manual_seed(1)
    x = torch.rand(2, 3, 256, 256)
    with torch.no_grad():
        y = net(x.cuda()).cpu()
        y0, x0, m = 80, 120, 20
        ref = oracle.rddbnet_forward(sd, x[:, :, y0:y0 + 96, x0:x0 + 96], 4)
    got = y[:, :, 4 * (y0 + m):4 * (y0 + 96 - m), 4 * (x0 + m):4 * (x0 + 96 - m)]
    assert rel_err(got, ref[:, :, 4 * m:4 * (96 - m), 4 * m:4 * (96 - m)]) < 1e-3


def test_ragged_and_tiny_shapes(ops):
    """1x1 image, single row, width not a multiple of the 32-pixel tile, batch 1."""
    torch.manual_seed(2)
    for (h, w_) in ((1, 1), (1, 37), (5, 33), (17, 1)):
        x = torch.rand(1, 16, h, w_) - 0.5
        wt = torch.randn(8, 16, 3, 3) * 0.1
        y = torch.zeros(1, h, w_, 8, device="cuda")
        ops.conv_igemm(ops.to_nhwc(x.cuda(), 16, "fp32"), ops.pack_conv2d_fwd(wt.cuda(), "fp32"), y, kh=3, kw=3, Cout=8, pad=(1, 1))
        assert rel_err(ops.to_nchw(y).cpu(), F.conv2d(x, wt, None, 1, 1)) < 1e-3, (h, w_)


def test_bad_arguments_are_rejected_before_launch(ops):
    x = torch.zeros(1, 4, 4, 12, device="cuda", dtype=torch.bfloat16)          # 12 channels: not a multiple of 8
    y = torch.zeros(1, 4, 4, 8, device="cuda", dtype=torch.bfloat16)
    wp = torch.zeros(4096, device="cuda", dtype=torch.uint8)
    with pytest.raises(RuntimeError, match="multiples of 8"):
        ops.conv_igemm(x, wp, y, kh=3, kw=3, Cout=8, pad=(1, 1))
    with pytest.raises(RuntimeError, match="unsupported kernel"):
        ops.conv_igemm(torch.zeros(1, 4, 4, 8, device="cuda", dtype=torch.bfloat16), wp, y, kh=6, kw=6, Cout=8, pad=(2, 2))
