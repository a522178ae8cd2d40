"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/srcgan_amd.h declares;
the ctypes table in srcgan_amd/_native.py covers exactly the same set.  No compute call is made."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "srcgan_amd.h")


def header_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(srcgan_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib_path():
    from srcgan_amd import build
    return build.build(verbose=False)


def test_header_declares_the_hot_path_entry_points():
    syms = header_symbols()
    for must in ("srcgan_conv_igemm", "srcgan_conv_wgrad", "srcgan_rddbnet_forward", "srcgan_rddbnet_backward",
                 "srcgan_nlayerd_forward", "srcgan_nlayerd_backward", "srcgan_loss_fwd", "srcgan_loss_bwd",
                 "srcgan_pack_weight", "srcgan_bn_apply_lrelu", "srcgan_rgb_to_gray", "srcgan_bilinear_down"):
        assert must in syms


def test_ctypes_table_matches_header():
    from srcgan_amd import _native
    assert sorted(_native.SIGNATURES) == header_symbols()


def test_library_exports_every_declared_symbol(lib_path):
    out = subprocess.run(["nm", "-D", "--defined-only", lib_path], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r"\bT (srcgan_[a-z0-9_]+)", out))
    missing = [s for s in header_symbols() if s not in exported]
    assert not missing, missing


def test_library_loads_and_answers_host_only_queries(lib_path):
    import ctypes as C
    from srcgan_amd import _native as N
    lib = N.lib()
    assert lib.srcgan_version() >= 100
    assert lib.srcgan_dtype_size(N.F32) == 4 and lib.srcgan_dtype_size(N.BF16) == 2
    # planners are pure host code: BASELINE config 2 shapes
    cfg = N.RddbCfg(3, 3, 4, 64, 23, 32, 16, 256, 256, N.BF16, 0)
    assert lib.srcgan_rddbnet_num_params(C.byref(cfg)) == 697
    assert 20e9 < lib.srcgan_rddbnet_ws_bytes(C.byref(cfg)) < 60e9
    d = N.NLayerDCfg(3, 64, 3, 16, 1024, 1024, N.BF16, 1)
    oh, ow = C.c_int(), C.c_int()
    assert lib.srcgan_nlayerd_out_hw(C.byref(d), C.byref(oh), C.byref(ow)) == 0
    assert (oh.value, ow.value) == (126, 126)
    assert lib.srcgan_nlayerd_num_params(C.byref(d)) == 13
    # rejected configurations report an error string instead of launching
    bad = N.RddbCfg(3, 3, 4, 60, 1, 32, 1, 8, 8, N.BF16, 0)
    assert lib.srcgan_rddbnet_ws_bytes(C.byref(bad)) == 0
    assert b"multiples of 8" in lib.srcgan_last_error()


def test_code_object_targets_gfx950(lib_path):
    data = open(lib_path, "rb").read()
    assert b"gfx950" in data
