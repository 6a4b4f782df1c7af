import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "grail-rs_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "perf: carries a wall-clock assertion (skipped, with the evidence, where the device's clocks "
                                       "do not hold still; deselect with -m 'gpu and not perf')")


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


def kernel_time_spread(ctx, launches=6):
    """How far identical launches differ on this device right now: (max - min) / min of the kernel time of one small
    pinned launch repeated.  The wall-clock guards of the GPU suite (planner choice, lone-stream pace) mean something only
    on a device whose clocks hold still; on a shared or throttled one a guard that misses is SKIPPED with this figure
    instead of failing the parity suite for a reason that has nothing to do with correctness."""
    import grail_hip as G                                   # noqa: F401
    from grail_hip import workload as W
    saved = {k: ctx.get_option(k) for k in ("arithmetic", "lanes_per_utterance")}
    ctx.set_voices(W.single_voice())
    segs, offs, vids, seeds = W.make_batch(8192, length=0.0625, blend_length=0.0625)
    stride = (W.max_samples(length=0.0625) + 63) // 64 * 64
    batch = ctx.upload(segs, offs, vids, seeds)
    d_out, d_len = ctx.device_alloc(8192 * stride * 4), ctx.device_alloc(8192 * 4)
    try:
        ctx.set_option("arithmetic", 0)
        ctx.set_option("lanes_per_utterance", 1)
        ms = []
        for i in range(launches + 1):
            batch.synthesize_async(d_out, stride, d_len)
            ctx.sync()
            if i:
                ms.append(ctx.last_kernel_ms())
    finally:
        for k, v in saved.items():
            ctx.set_option(k, v)
        ctx.device_free(d_out)
        ctx.device_free(d_len)
        batch.free()
    return (max(ms) - min(ms)) / min(ms)


def skip_if_clocks_unstable(ctx, what, limit=0.05):
    """Called where a wall-clock guard has MISSED: skip (with the evidence) if identical launches differ by more than `limit`."""
    spread = kernel_time_spread(ctx)
    if spread > limit:
        pytest.skip(f"{what}; identical launches differ by {100 * spread:.1f} % on this device right now "
                    f"(> {100 * limit:.0f} %): a timing guard cannot be judged here")


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
