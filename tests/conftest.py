import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


@pytest.fixture(scope="session")
def golden():
    return load_golden


def rel_err(a, b):
    """per-QP relative L2 error of a against b (rows = QPs)"""
    a = np.asarray(a, np.float64).reshape(len(a), -1)
    b = np.asarray(b, np.float64).reshape(len(b), -1)
    return np.linalg.norm(a - b, axis=1) / np.maximum(np.linalg.norm(b, axis=1), 1e-300)


@pytest.fixture(autouse=True)
def poison_uninitialised_tensors(request, monkeypatch):
    """Every floating-point tensor the code under test obtains from torch.empty starts as NaN (QPX_TEST_POISON=0 turns it
    off).  The package allocates outputs, factor blobs and workspaces with torch.empty; the large-QP family deliberately
    leaves parts of its blob unwritten (the upper triangles of its factors since round 5).  With poisoned memory a kernel
    that READS such a word, or a test that COMPARES it, fails deterministically instead of depending on what the caching
    allocator hands back (round 5: a mis-sized GPU test shape had compared unwritten words for two rounds and passed)."""
    if os.environ.get("QPX_TEST_POISON", "1") == "0":
        yield
        return
    import torch
    real_empty = torch.empty

    def empty(*args, **kwargs):
        t = real_empty(*args, **kwargs)
        if t.is_floating_point() and t.numel():
            t.fill_(float("nan"))
        return t

    monkeypatch.setattr(torch, "empty", empty)
    yield
