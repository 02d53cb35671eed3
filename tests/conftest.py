import os
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    # the CPU oracle legs of the GPU tests (32 oracle frames in the free-running trajectory test alone) run on torch's intra-op pool: on
    # the GPU boxes' 256 hardware threads torch takes 128 and the pool thrashes across both packages — that ONE test: 295 s with the
    # default pool, 43 s at 32 threads, 36 s at 16 (scripts/r05y.sh; the whole suite 500 - 630 s -> under 300).  SGAM_TEST_THREADS overrides.
    import torch
    want = int(os.environ.get("SGAM_TEST_THREADS", "16"))
    if (os.cpu_count() or 1) > 64 and torch.get_num_threads() > want:
        torch.set_num_threads(want)
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")
    config.addinivalue_line("markers", "experimental: kernels outside the default plans (SGAM_TEST_EXPERIMENTAL=1 to run)")


def pytest_collection_modifyitems(config, items):
    import torch
    if os.environ.get("SGAM_TEST_EXPERIMENTAL") != "1":
        # kernels that are built and reachable by an explicit plan / environment switch but are NOT in the default plans
        # (DESIGN.md 5.5 / 5.5b "what did not pay"): their tests stay in the tree and run on request
        # (SGAM_TEST_EXPERIMENTAL=1 pytest -m experimental), not in the driver's `-m gpu` pass (budget: < 600 s)
        skip_x = pytest.mark.skip(reason="experimental kernel, not in the default plans (SGAM_TEST_EXPERIMENTAL=1 to run)")
        for item in items:
            if "experimental" in item.keywords:
                item.add_marker(skip_x)
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return load


def bits_equal(a, b):
    """bit-for-bit equality of two float32 arrays, NaN == NaN."""
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    return a.shape == b.shape and bool(((a.view(np.uint32) == b.view(np.uint32)) | ((a != a) & (b != b))).all())
