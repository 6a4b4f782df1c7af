"""The workgroup dispatcher's model and the packed launch order (grail_dispatch_model, grail_packed_launch_order; option
"packed_launch_order"), without a GPU.

tests/golden/dispatch_r06.npz holds four launches RECORDED on an MI355X by tools/dispatch_order.hip (2 048 and 2 500
one-wave workgroups that hold a SIMD alone and spin for given times — the waves of 131 072 / 160 000 speech-like rows,
scaled — launched longest first and in an early packed order): per workgroup its time, the XCC and shader engine it ran on
and when it started, and the launch's makespan.  The model must give those makespans; what the model assumes about the
hardware is checked on the records themselves."""
import ctypes as C
import importlib.util
import os

import numpy as np
import pytest

import grail_hip as G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LAUNCHES = ["131072_rows_longest_first", "131072_rows_packed_order", "160000_rows_longest_first", "160000_rows_packed_order"]


@pytest.fixture(scope="module")
def recorded():
    return np.load(os.path.join(ROOT, "tests", "golden", "dispatch_r06.npz"))


def _reference_model():
    spec = importlib.util.spec_from_file_location("packed_order_experiment", os.path.join(ROOT, "tools", "packed_order_experiment.py"))
    mod = importlib.util.module_from_spec(spec)
    try:
        spec.loader.exec_module(mod)
    except Exception as e:                                  # noqa: BLE001 (the tool imports the binding: fine here)
        pytest.skip(f"tools/packed_order_experiment.py does not import: {e}")
    return mod.dispatch_makespan


@pytest.mark.parametrize("launch", LAUNCHES)
def test_the_records_show_what_the_model_assumes(recorded, launch):
    xcc, se, start = recorded[launch + "/xcc"], recorded[launch + "/se"], recorded[launch + "/start_us"]
    n = len(xcc)
    assert np.array_equal(xcc, np.arange(n) % 8)                       # workgroup b on XCC b mod 8
    for x in range(8):
        mine = np.arange(x, n, 8)
        pattern = se[mine][:4]
        assert sorted(pattern.tolist()) == [0, 1, 2, 3]
        assert np.array_equal(se[mine], np.tile(pattern, len(mine) // 4 + 1)[:len(mine)])   # a static round robin
        st = start[mine]
        assert np.all(st >= np.maximum.accumulate(st) - 5.0)           # in launch order (5 us of clock skew)


@pytest.mark.parametrize("launch", LAUNCHES)
def test_the_model_gives_the_recorded_makespans(built, recorded, launch):
    cost, measured = recorded[launch + "/cost_us"], float(recorded[launch + "/makespan_us"])
    model = G.dispatch_model(cost)
    assert abs(model - measured) <= 1e-3 * measured, (model, measured)
    assert abs(model - _reference_model()(cost)) <= 1e-9 * model       # the tool's Python model is the same arithmetic


@pytest.mark.parametrize("rows", ["131072", "160000"])
def test_the_packed_order_evens_the_simds_out(built, recorded, rows):
    cost = recorded[rows + "_rows_longest_first/cost_us"]
    plain = G.dispatch_model(cost)
    order = G.packed_launch_order(cost)
    assert sorted(order.tolist()) == list(range(len(cost)))
    packed = G.dispatch_model(cost, order)
    ideal = max(cost.sum() / 1024.0, cost.max())
    assert packed <= 0.92 * plain and packed <= 1.06 * ideal, (plain / ideal, packed / ideal)
    # workgroups are dealt to the 32 pools by cost, in turn: positions p, p + 32, p + 64 ... hold pool p's share
    rank = np.empty(len(cost), dtype=np.int64)
    rank[np.argsort(-cost, kind="stable")] = np.arange(len(cost))
    for p in (0, 7, 31):
        assert np.all(rank[order[p::32]] % 32 == p)


def test_one_pool_when_the_device_is_not_whole_xccs_and_workgroups_of_four_waves(built):
    rng = np.random.default_rng(3)
    cost = np.sort(rng.uniform(1.0, 4.0, size=500))[::-1].copy()
    # 18 compute units (a test's "assume_compute_units"): one pool of 72 SIMDs, plain greedy list scheduling
    import heapq
    free = [0.0] * 72
    for c in cost:
        heapq.heappush(free, heapq.heappop(free) + c)
    assert abs(G.dispatch_model(cost, compute_units=18) - max(free)) < 1e-9
    order = G.packed_launch_order(cost, compute_units=18)
    assert sorted(order.tolist()) == list(range(500))
    assert G.dispatch_model(cost, order, compute_units=18) <= G.dispatch_model(cost, compute_units=18) * 1.0001
    # four waves per workgroup: a compute unit each, 8 per shader engine
    few = cost[:300]
    assert G.dispatch_model(few, waves_per_workgroup=4) >= G.dispatch_model(few, waves_per_workgroup=1)
    o4 = G.packed_launch_order(few, waves_per_workgroup=4)
    assert sorted(o4.tolist()) == list(range(300))
    # fewer workgroups than the device holds: everything starts at once, whatever the order
    assert G.dispatch_model(cost[:100]) == cost[0]


def test_bad_arguments(built):
    L = G.load()
    c = (C.c_double * 4)(1.0, 2.0, 3.0, 4.0)
    out = C.c_double()
    o = (C.c_uint32 * 4)(0, 1, 2, 9)
    assert L.grail_dispatch_model(256, 1, c, o, 4, C.byref(out)) == G.ERR_INVALID_ARG
    assert L.grail_dispatch_model(256, 2, c, None, 4, C.byref(out)) == G.ERR_INVALID_ARG
    assert L.grail_dispatch_model(0, 1, c, None, 4, C.byref(out)) == G.ERR_INVALID_ARG
    assert L.grail_dispatch_model(256, 1, c, None, 4, None) == G.ERR_INVALID_ARG
    assert L.grail_packed_launch_order(256, 1, c, 4, None) == G.ERR_INVALID_ARG
    assert L.grail_dispatch_model(256, 1, None, None, 0, C.byref(out)) == G.OK and out.value == 0.0
