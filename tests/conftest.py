"""Shared test plumbing.  GPU tests are marked ``gpu``; everything else must pass on a CPU-only box."""
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
    config.addinivalue_line("markers", "gpu: needs a ROCm device (run on the MI355X box with -m gpu)")
    config.addinivalue_line("filterwarnings", r"ignore:index_reduce\(\) is in beta")     # partition.propagate_sum, once per step


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def golden_names(prefix):
    return sorted(f[:-4] for f in os.listdir(GOLDEN) if f.startswith(prefix) and f.endswith(".npz"))


def t(a):
    """numpy -> torch (scalars stay Python numbers)."""
    if isinstance(a, np.ndarray) and a.ndim == 0:
        return a.item()
    return torch.from_numpy(np.ascontiguousarray(a))


def rel_fro(a, b):
    """||a-b||_F / ||b||_F"""
    a, b = a.double(), b.double()
    return ((a - b).norm() / b.norm().clamp_min(1e-300)).item()


def worst_row_rel(a, b):
    """max over rows of ||a_r - b_r||_2 / ||b_r||_2 (rows with zero reference norm must match exactly)."""
    a, b = a.double(), b.double()
    num, den = (a - b).norm(dim=1), b.norm(dim=1)
    zero = den == 0
    if zero.any():
        assert (num[zero] == 0).all(), "rows that are exactly zero in the reference must be exactly zero"
    if (~zero).any():
        return (num[~zero] / den[~zero]).max().item()
    return 0.0


@pytest.fixture(scope="session")
def device():
    if not torch.cuda.is_available():
        pytest.skip("no ROCm device")
    return torch.device("cuda:0")
