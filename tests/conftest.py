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
    """max |a-b| / max|b| -- the relative measure used for the 1e-3 fp32 gate."""
    a = torch.as_tensor(a).detach().to(torch.float64).cpu()
    b = torch.as_tensor(b).detach().to(torch.float64).cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.fixture(scope="session")
def golden():
    return load_golden


def rel_l2(a, b):
    """||a-b|| / ||b|| -- used for the bf16 perf mode, where single elements carry 8-bit rounding noise."""
    a = torch.as_tensor(a).detach().to(torch.float64).cpu()
    b = torch.as_tensor(b).detach().to(torch.float64).cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))
