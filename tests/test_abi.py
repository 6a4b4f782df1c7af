"""CPU-side checks: the C-ABI library loads, exports what include/grail_hip.h declares,
its host-side parameter algebra agrees with the oracle, and compute calls fail loudly
without a device (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import grail_hip as G
import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "grail_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(grail_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(built):
    lib = G.load()
    declared = header_functions()
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in grail_hip.h but not exported"
    assert sorted(G.EXPORTS) == declared


def test_header_option_list_equals_what_the_library_accepts():
    """Option names live in three places — the header's option block (the contract), grail_set_option and
    grail_get_option (csrc/grail_api.cpp): they must name the same options; what the header calls read-only is exactly
    what only grail_get_option knows; the block stays a contract (no measured milliseconds, under 60 lines)."""
    header = open(os.path.join(ROOT, "include", "grail_hip.h")).read()
    start = header.index("/* Options (grail_set_option / grail_get_option")
    block = header[start:header.index("*/", start)]
    settable_text, readonly_text = block.split("Read-only (grail_get_option)")
    documented_set = set(re.findall(r'"([a-z0-9_]+)"', settable_text))
    documented_ro = set(re.findall(r'"([a-z0-9_]+)"', readonly_text)) - {"arithmetic"}      # (named in a description)
    api = open(os.path.join(ROOT, "grail-rs_amd", "csrc", "grail_api.cpp")).read()
    set_body = api[api.index("int grail_set_option("):api.index("int grail_get_option(")]
    get_start = api.index("int grail_get_option(")
    get_body = api[get_start:api.index("\n}\n", get_start)]
    names = lambda body: set(re.findall(r'strcmp\(name, "([a-z0-9_]+)"\)', body))
    accepted_set, accepted_get = names(set_body), names(get_body)
    assert documented_set - {"last_launch_fast"} == accepted_set, (documented_set ^ accepted_set)
    assert documented_ro == accepted_get - accepted_set, (documented_ro ^ (accepted_get - accepted_set))
    assert accepted_set <= accepted_get | {"scan_debug"}, accepted_set - accepted_get      # every option can be read back
    assert len(block.splitlines()) < 60
    assert not re.search(r"\d\s*(ms|us)\b", block), "measurements belong in DESIGN.md section 4"


def test_abi_version_and_status_strings(built):
    lib = G.load()
    assert lib.grail_abi_version() == G.ABI_VERSION == 3
    hdr = open(os.path.join(ROOT, "include", "grail_hip.h")).read()
    assert re.search(r"#define GRAIL_ABI_VERSION (\d+)", hdr).group(1) == "3"
    sys_rs = open(os.path.join(ROOT, "grail-rs_amd", "rust", "grail-hip-sys", "src", "lib.rs")).read()
    assert "pub const GRAIL_ABI_VERSION: c_int = 3;" in sys_rs
    for st in range(0, -8, -1):
        assert lib.grail_status_string(st)
    assert b"CPU fallback" in lib.grail_status_string(G.ERR_NO_DEVICE)


def test_struct_layouts():
    assert C.sizeof(G.SynthesisElem) == 196 == C.sizeof(O.SynthesisElem)
    assert C.sizeof(G.Voice) == 4 + 2 * 196 + 20 == C.sizeof(O.Voice)
    assert C.sizeof(G.PhonemeElem) == 16 == C.sizeof(O.PhonemeElem)
    assert C.sizeof(G.SequenceElem) == 4 + 196 + 8 == C.sizeof(O.SequenceElem)


@pytest.mark.parametrize("rate", [None, 48000.0, 22050.0, 8000.0, 96000.0])
def test_voice_generic_matches_oracle_bytes(built, rate):
    assert bytes(G.voice_generic(rate)) == bytes(O.voice_generic(rate))


def test_generic_voice_known_values(built):
    v = G.voice_generic()
    f32 = np.float32
    scale = f32(1.0) / f32(44100.0)
    # reference src/lib.rs:420-430: multiply by the f32 reciprocal, not divide
    assert f32(v.phonemes[0].formant_freq[0]) == f32(910.0) * scale
    assert f32(v.phonemes[0].formant_freq[0]) != f32(910.0) / f32(44100.0)
    assert f32(v.center_frequency) == f32(120.0) / f32(44100.0)
    amps = np.array(v.phonemes[0].formant_amp[:], dtype=np.float32)
    total = f32(0)
    for a in [0.3, 0.3, 0.2, 0.1, 0, 0, 0, 0]:
        total = f32(total + f32(a))
    assert amps[0] == f32(0.3) / total
    assert np.all(amps[4:] == 0)


def test_resample_drops_formants_above_nyquist(built):
    v = G.voice_generic(6000.0)  # 4000 Hz formant -> 0.667 > 0.5
    e = v.phonemes[0]
    assert e.formant_freq[7] == 0.5 and e.formant_amp[7] == 0.0
    assert bytes(v) == bytes(O.voice_generic(6000.0))


def test_elem_helpers_match_oracle(built):
    rng = np.random.default_rng(1)
    L = O.lib()
    for _ in range(20):
        a = rng.uniform(0, 0.5, 49).astype(np.float32)
        b = rng.uniform(0, 0.5, 49).astype(np.float32)
        alpha = float(np.float32(rng.uniform(-0.2, 1.2)))
        got = G.elem_blend(G.SynthesisElem.from_np(a), G.SynthesisElem.from_np(b), alpha).as_np()
        oa = O.SynthesisElem.from_buffer_copy(a.tobytes())
        ob = O.SynthesisElem.from_buffer_copy(b.tobytes())
        oo = O.SynthesisElem()
        L.orc_elem_blend(C.byref(oo), C.byref(oa), C.byref(ob), C.c_float(alpha))
        assert np.array_equal(got.view(np.uint32), oo.as_np().view(np.uint32))
    s = G.elem_silent().as_np()
    os_ = O.SynthesisElem()
    L.orc_elem_silent(C.byref(os_))
    assert np.array_equal(s, os_.as_np())


def test_shard_range_partitions_exactly(built):
    for n in [0, 1, 7, 8, 65536, 524288, 1000003, 2**40 + 5]:
        for world in [1, 2, 3, 8]:
            prev = 0
            for r in range(world):
                b, e = G.shard_range(n, r, world)
                assert b == prev and e >= b
                prev = e
            assert prev == n
    assert G.shard_range(524288, 3, 8) == (3 * 65536, 4 * 65536)


def test_voice_blob_roundtrip(built):
    from grail_hip import workload as W
    voices = W.preset_voices(8)
    blob = G.voices_blob(voices)
    assert len(blob) == 8 * C.sizeof(G.Voice)
    back = G.voices_from_blob(blob)
    assert all(bytes(a) == bytes(b) for a, b in zip(voices, back))


def test_workload_is_deterministic_and_shardable(built):
    from grail_hip import workload as W
    segs, offs, vids, seeds = W.make_batch(64, n_voices=8)
    segs2, offs2, vids2, seeds2 = W.make_batch(16, first_utt=32, n_voices=8)
    assert np.array_equal(segs[32 * 4:48 * 4], segs2)
    assert np.array_equal(vids[32:48], vids2) and np.array_equal(seeds[32:48], seeds2)
    assert np.all(segs["phoneme"][::4] == G.PH_SILENCE)
    hz = segs["frequency"] * np.float32(48000.0)
    assert hz.min() >= 99.9 and hz.max() <= 200.1


def test_compute_calls_fail_loudly_without_a_device(built):
    if G.device_count() > 0:
        pytest.skip("a device is present")
    with pytest.raises(G.GrailError) as ei:
        G.Context(0)
    assert ei.value.status == G.ERR_NO_DEVICE


def test_rust_sys_crate_mirrors_the_header():
    """grail-hip-sys (source only: no rustc in the image) declares exactly the header's functions."""
    src = open(os.path.join(ROOT, "grail-rs_amd", "rust", "grail-hip-sys", "src", "lib.rs")).read()
    rust = sorted(set(re.findall(r"pub fn (grail_[a-z0-9_]+)\s*\(", src)))
    assert rust == header_functions()


def test_cpp_facade_example_builds(built):
    exe = os.path.join(ROOT, "grail-rs_amd", "lib", "grail_say")
    assert os.path.exists(exe)
    import subprocess
    if G.device_count() == 0:   # no CPU fallback: the example must fail loudly, not fake audio
        r = subprocess.run([exe, "a"], capture_output=True, text=True)
        assert r.returncode == 1 and "no HIP device" in r.stderr


def test_bench_gpus_n_without_a_launcher_never_reports_one_gpu(built):
    """`python bench.py --gpus 2` started directly must run two ranks or fail — never time one GPU and
    print n_gpus: 1 (round-1 hole).  Without a device the ranks fail and the launcher must say so."""
    if G.device_count() > 0:
        pytest.skip("a device is present: covered by tests/test_bench_multirank_gpu.py")
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1",
                        "--warmup", "0", "--utts", "64"], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=300)
    assert p.returncode != 0
    assert "n_gpus" not in p.stdout
