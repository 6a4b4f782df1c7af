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
    assert lib.grail_abi_version() == G.ABI_VERSION == 4
    hdr = open(os.path.join(ROOT, "include", "grail_hip.h")).read()
    assert re.search(r"#define GRAIL_ABI_VERSION (\d+)", hdr).group(1) == "4"
    sys_rs = open(os.path.join(ROOT, "grail-rs_amd", "rust", "grail-hip-sys", "src", "lib.rs")).read()
    assert "pub const GRAIL_ABI_VERSION: c_int = 4;" in sys_rs
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


# ---- the Rust -sys crate against the header, signature by signature --------------------------------------------------
# There is no rustc in the image, so nothing compiles the crate against the header: this parser is what keeps a swapped
# u32 / u64 or a dropped parameter out of it.  Types are brought to one canonical form on both sides: a base name and the
# constness of what each pointer level points to, outermost first ("*const *const c_char" == "const char *const *").
_C_BASE = {"uint32_t": "u32", "int32_t": "i32", "uint64_t": "u64", "int64_t": "i64", "uint8_t": "u8", "int16_t": "i16",
           "float": "f32", "double": "f64", "size_t": "usize", "int": "c_int", "char": "c_char", "void": "c_void"}


def _c_type(tokens):
    """['const', 'char', '*', 'const', '*'] -> ('c_char', ('const', 'const')) — pointer levels outermost first."""
    toks = list(tokens)
    base_const = False
    while toks and toks[0] == "const":
        base_const, toks = True, toks[1:]
    base, toks = toks[0], toks[1:]
    if toks and toks[0] == "const":
        base_const, toks = True, toks[1:]
    levels, pointee_const = [], base_const
    while toks:
        assert toks[0] == "*", tokens
        levels.append("const" if pointee_const else "mut")
        toks = toks[1:]
        pointee_const = bool(toks) and toks[0] == "const"
        if pointee_const:
            toks = toks[1:]
    return _C_BASE.get(base, base), tuple(reversed(levels))


def _c_decl(decl):
    """One C parameter or struct field -> (name, canonical type, array length or None)."""
    m = re.match(r"^(.*?)(\w+)\s*(?:\[(\w+)\])?$", decl.strip(), re.S)
    ty, name, arr = m.group(1), m.group(2), m.group(3)
    return name, _c_type(re.findall(r"\w+|\*", ty)), arr


def c_header_model():
    src = open(os.path.join(ROOT, "include", "grail_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    structs = {}
    for body, name in re.findall(r"typedef struct \w+ \{(.*?)\}\s*(\w+);", src, flags=re.S):
        fields = []
        for f in body.split(";"):
            if f.strip():
                fname, ty, arr = _c_decl(f)
                fields.append((fname, ty, arr))
        structs[name] = fields
    src = re.sub(r"typedef (struct|enum) \w+ \{.*?\}\s*\w+;", "", src, flags=re.S)
    src = re.sub(r"^\s*#.*$", "", src, flags=re.M)                       # preprocessor lines
    src = re.sub(r"typedef struct \w+\s+\w+;", "", src)                  # the opaque handles
    src = src.replace('extern "C" {', "")
    funcs = {}
    for ret, name, params in re.findall(r"([\w\s\*]+?)\b(grail_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", src):
        ret_ty = _c_type(re.findall(r"\w+|\*", ret))
        plist = []
        if params.strip() != "void":
            for prm in params.split(","):
                _, ty, arr = _c_decl(prm)
                if arr:                                   # an array parameter is a pointer to its element
                    ty = (ty[0], (("const" if "const" in prm.split() else "mut"),) + ty[1])
                plist.append(ty)
        funcs[name] = (None if ret_ty == ("c_void", ()) else ret_ty, plist)
    return funcs, structs


def _rust_type(text):
    toks = text.replace("std::ffi::c_void", "c_void").split()
    levels = []
    while toks[0] in ("*const", "*mut"):
        levels.append(toks[0][1:])
        toks = toks[1:]
    assert len(toks) == 1, text
    return toks[0], tuple(levels)


def rust_sys_model(src):
    ext = src[src.index('extern "C" {'):]
    funcs = {}
    for name, params, ret in re.findall(r"pub fn (grail_[a-z0-9_]+)\s*\(([^)]*)\)\s*(?:->\s*([^;]+?))?\s*;", ext, flags=re.S):
        plist = [_rust_type(prm.split(":", 1)[1].strip()) for prm in params.split(",") if prm.strip()]
        funcs[name] = (_rust_type(ret.strip()) if ret else None, plist)
    structs = {}
    for name, body in re.findall(r"pub struct (grail_\w+) \{(.*?)\n\}", src, flags=re.S):
        fields = []
        for fname, ty in re.findall(r"pub (\w+):\s*([^,\n]+),", body):
            m = re.match(r"\[(.+);\s*(\w+)\]", ty.strip())
            fields.append((fname, _rust_type(m.group(1).strip() if m else ty.strip()), m.group(2) if m else None))
        if fields:
            structs[name] = fields
    sizes = {n: int(v) for n, v in re.findall(r"assert!\(std::mem::size_of::<(grail_\w+)>\(\) == (\d+)\)", src)}
    return funcs, structs, sizes


def ffi_differences(rust_src):
    c_funcs, c_structs = c_header_model()
    r_funcs, r_structs, _ = rust_sys_model(rust_src)
    diffs = []
    for name in sorted(set(c_funcs) | set(r_funcs)):
        if name not in c_funcs or name not in r_funcs:
            diffs.append(f"{name}: declared on one side only")
        elif c_funcs[name] != r_funcs[name]:
            diffs.append(f"{name}: header {c_funcs[name]} != crate {r_funcs[name]}")
    for name in sorted(c_structs):
        if name not in r_structs:
            diffs.append(f"struct {name}: missing in the crate")
        elif c_structs[name] != r_structs[name]:
            diffs.append(f"struct {name}: header {c_structs[name]} != crate {r_structs[name]}")
    return diffs


def _sys_source():
    return open(os.path.join(ROOT, "grail-rs_amd", "rust", "grail-hip-sys", "src", "lib.rs")).read()


def test_rust_sys_crate_mirrors_the_header():
    """grail-hip-sys (source only: no rustc in the image) declares exactly the header's functions — names, arity, every
    parameter type, every return type — and its #[repr(C)] structs have the header's fields in the header's order."""
    src = _sys_source()
    rust = sorted(set(re.findall(r"pub fn (grail_[a-z0-9_]+)\s*\(", src)))
    assert rust == header_functions()
    c_funcs, c_structs = c_header_model()
    assert sorted(c_funcs) == header_functions() and len(c_structs) >= 7
    assert ffi_differences(src) == []


def test_the_signature_check_catches_a_swapped_width_a_dropped_parameter_and_a_lost_const():
    src = _sys_source()
    cases = [
        ("out: *mut f32, out_stride: u64, out_len: *mut u32, flags: u32) -> c_int;",
         "out: *mut f32, out_stride: u32, out_len: *mut u32, flags: u32) -> c_int;", "grail_synthesize_batch"),
        ("pub fn grail_shard_range(n_utt: u64, rank: u32, world: u32, begin: *mut u64, end: *mut u64);",
         "pub fn grail_shard_range(n_utt: u64, rank: u32, begin: *mut u64, end: *mut u64);", "grail_shard_range"),
        ("pub fn grail_batch_size(batch: *const grail_batch) -> u32;",
         "pub fn grail_batch_size(batch: *mut grail_batch) -> u32;", "grail_batch_size"),
        ("pub fn grail_length_bound(segment_lengths: *const f32, n_segments: u32, sample_rate: f32) -> u64;",
         "pub fn grail_length_bound(segment_lengths: *const f32, n_segments: u32, sample_rate: f32) -> u32;",
         "grail_length_bound"),
        ("    pub first_seg: u32,\n    pub n_segs: u32,", "    pub n_segs: u32,\n    pub first_seg: u32,", "struct grail_node_shard"),
    ]
    for good, bad, where in cases:
        assert src.count(good) >= 1, good
        diffs = ffi_differences(src.replace(good, bad, 1))
        assert len(diffs) == 1 and diffs[0].startswith(where), (where, diffs)


def test_rust_struct_size_asserts_hold_for_the_header(tmp_path):
    """The crate's `const _: () = assert!(size_of::<T>() == N)` literals against what the C compiler gives the header's
    structs (and against the ctypes mirrors the GPU tests call through)."""
    import subprocess
    _, _, sizes = rust_sys_model(_sys_source())
    names = ["grail_synthesis_elem", "grail_voice", "grail_phoneme_elem", "grail_sequence_elem", "grail_plan_block",
             "grail_node_shard"]
    assert sorted(sizes) == sorted(names)
    prog = tmp_path / "sizes.c"
    prog.write_text('#include <stdio.h>\n#include "grail_hip.h"\nint main(void) {\n' +
                    "".join(f'    printf("{n} %zu\\n", sizeof({n}));\n' for n in names) + "    return 0;\n}\n")
    exe = tmp_path / "sizes"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(prog), "-o", str(exe)])
    got = dict(line.split() for line in subprocess.check_output([str(exe)], text=True).splitlines())
    assert {n: int(v) for n, v in got.items()} == sizes
    mirrors = {"grail_synthesis_elem": G.SynthesisElem, "grail_voice": G.Voice, "grail_phoneme_elem": G.PhonemeElem,
               "grail_sequence_elem": G.SequenceElem, "grail_plan_block": G.PlanBlock, "grail_node_shard": G.NodeShard}
    assert {n: C.sizeof(t) for n, t in mirrors.items()} == sizes


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
