import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    return oracle_lib.load()


@pytest.fixture(scope="session")
def gpu_store():
    """One ChunkStore (sq_ctx) per test session on cuda:0; fails loudly if the HIP library or the
    device is missing (no CPU fallback exists)."""
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import squishrs_amd
    st = squishrs_amd.ChunkStore(device=0, dedup_capacity=1 << 20)
    yield st
    st.close()
