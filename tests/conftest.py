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


def _settable_options():
    """The names of the header's option block that grail_set_option accepts and grail_get_option reads back."""
    import re
    header = open(os.path.join(ROOT, "include", "grail_hip.h")).read()
    start = header.index("/* Options (grail_set_option / grail_get_option")
    block = header[start:header.index("*/", start)].split("Read-only (grail_get_option)")[0]
    return sorted(set(re.findall(r'"([a-z0-9_]+)"', block)) - {"last_launch_fast", "scan_debug"})


@pytest.fixture(autouse=True)
def _options_as_found(request):
    """The GPU tests share one context: a test that leaves an option changed decides which kernels the tests after it
    exercise (and fails them, or — worse — lets them pass on another family).  Every test must hand the options back
    as it found them; a leak fails the test that leaked, after the options have been put back for the next one."""
    if "gpu_ctx" not in request.fixturenames:
        yield
        return
    ctx = request.getfixturevalue("gpu_ctx")
    names = _settable_options()
    before = {k: ctx.get_option(k) for k in names}
    yield
    after = {k: ctx.get_option(k) for k in names}
    leaked = {k: (before[k], after[k]) for k in names if before[k] != after[k]}
    for k, (b, _) in leaked.items():
        ctx.set_option(k, b)
    assert not leaked, f"options left changed (was, is): {leaked}"
