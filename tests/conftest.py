import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "python-visual-similarity_amd"), os.path.join(REPO, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(REPO, "tests", "golden")
if GOLDEN not in sys.path:
    sys.path.insert(0, GOLDEN)        # config3_inputs.py: seeded inputs shared with the fixture generator


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name: str):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def tables():
    return load_golden("tables_k256_d128")


@pytest.fixture(scope="session")
def gpu_ctx():
    """One engine context on cuda:0 for the whole GPU session (fails loudly if the HIP
    library is missing or no device is present -- there is no CPU fallback)."""
    import pvsim
    ctx = pvsim.Context(0)
    yield ctx
    ctx.close()
