import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "grail-rs_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Build (or reuse) libgrail_hip.so and liboracle.so once per session."""
    import __graft_entry__ as ge
    ge.build()
    return True


@pytest.fixture(scope="session")
def gpu_ctx(built):
    import grail_hip as G
    if G.device_count() < 1:
        pytest.fail("no HIP device visible: -m gpu tests must run on the GPU box "
                    "(the product has no CPU fallback)")
    ctx = G.Context(0)
    yield ctx
    ctx.close()
