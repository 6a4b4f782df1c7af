"""Vectors produced by the REAL reference crate, when somebody has generated them.

grail-rs_amd/rust/reference-golden is a small Rust program (source only: this image has no
rustc/cargo) that runs grail-rs itself on the cases of tests/golden/make_golden.py and writes the
samples to tests/golden/reference/.  When those files exist, the CPU oracle must reproduce them
bit for bit — that pins the oracle to the reference.  Until then these tests skip, and the
oracle stays pinned by hand-derived known answers only (parity unpinned)."""
import os

import numpy as np
import pytest

import oracle_lib as O

HERE = os.path.dirname(os.path.abspath(__file__))
REF_DIR = os.environ.get("GRAIL_REFERENCE_GOLDEN_DIR") or os.path.join(HERE, "golden", "reference")
sys_path_golden = os.path.join(HERE, "golden")

pytestmark = pytest.mark.skipif(
    not os.path.exists(os.path.join(REF_DIR, "manifest.txt")),
    reason="tests/golden/reference/ not generated (needs cargo: grail-rs_amd/rust/reference-golden)")


def manifest():
    path = os.path.join(REF_DIR, "manifest.txt")
    if not os.path.exists(path):
        return []
    return [(l.split()[0], int(l.split()[1])) for l in open(path) if l.strip()]


def golden_cases():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(sys_path_golden, "make_golden.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return {name: (rate, segs, seed) for name, rate, segs, seed in m.cases()}


@pytest.mark.parametrize("name,length", manifest())
def test_oracle_reproduces_the_reference_crate(name, length):
    rate, segs, seed = golden_cases()[name]
    want = np.fromfile(os.path.join(REF_DIR, name + ".f32"), dtype="<f4")
    assert len(want) == length
    pcm, n = O.synthesize_phonemes(O.voice_generic(rate), O.segments(segs), seed)
    assert n == length
    assert np.array_equal(pcm.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("tag,rate", [("44k", None), ("48k", 48000.0)])
def test_voice_tables_match_the_reference_crate(tag, rate):
    want = np.fromfile(os.path.join(REF_DIR, f"voice_{tag}.f32"), dtype="<f4")
    got = np.frombuffer(bytes(O.voice_generic(rate)), dtype=np.float32)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
