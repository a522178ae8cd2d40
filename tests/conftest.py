import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return {k: z[k] for k in z.files}


def sub(d, prefix):
    """{'sd/a': x} -> {'a': tensor(x)} for keys under prefix."""
    return {k[len(prefix):]: torch.from_numpy(np.asarray(v)) for k, v in d.items() if k.startswith(prefix)}


def rel_err(a, b):
    """The relative measure of the 1e-3 fp32 gate: the LARGER of max |a-b| / max |b| (tensor-max-normalised) and the relative
    L2 error ||a-b|| / ||b|| -- the first alone would let an error pattern hide in a tensor dominated by a few large entries."""
    a = torch.as_tensor(a).detach().to(torch.float64).cpu()
    b = torch.as_tensor(b).detach().to(torch.float64).cpu()
    mx = float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
    l2 = float((a - b).norm() / b.norm().clamp_min(1e-30))
    return max(mx, l2)


@pytest.fixture(scope="session")
def golden():
    return load_golden


def rel_l2(a, b):
    """||a-b|| / ||b|| -- used for the bf16 perf mode, where single elements carry 8-bit rounding noise."""
    a = torch.as_tensor(a).detach().to(torch.float64).cpu()
    b = torch.as_tensor(b).detach().to(torch.float64).cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))
