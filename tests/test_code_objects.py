"""The gfx950 code objects inside libgrail_hip.so, read without a GPU: every kernel the library can launch is there,
compiled for gfx950 only, free of scratch where the design says so — and ONE WAVE PER SIMD holds by construction:
the lane kernels claim more than half of a SIMD's 512 registers (profiles/r04_dispatch.txt), so the
dispatcher cannot put two of their waves on one SIMD whatever launch came before."""
import os
import re
import struct
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "grail-rs_amd", "lib", "libgrail_hip.so")
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _code_objects():
    """(triple, bytes) of every device entry of the clang offload bundles in the library's .hip_fatbin."""
    blob = open(LIB, "rb").read()
    out = []
    for m in re.finditer(re.escape(MAGIC), blob):
        p = m.start()
        (n,) = struct.unpack_from("<Q", blob, p + 24)
        q = p + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, q)
            q += 24
            triple = blob[q:q + tl].decode()
            q += tl
            if size and not triple.startswith("host-"):
                out.append((triple, blob[p + off:p + off + size]))
    return out


@pytest.fixture(scope="module")
def kernels(tmp_path_factory, built):
    """{kernel symbol: metadata dict} over all gfx950 code objects (AMDGPU metadata note, via llvm-readelf)."""
    if not os.path.exists(READELF):
        pytest.skip("llvm-readelf not found")
    tmp = tmp_path_factory.mktemp("co")
    found = {}
    for i, (triple, data) in enumerate(_code_objects()):
        assert triple.endswith("gfx950"), f"a code object for {triple}: this library is gfx950 only"
        path = tmp / f"co{i}.elf"
        path.write_bytes(data)
        text = subprocess.run([READELF, "--notes", str(path)], capture_output=True, text=True, check=True).stdout
        for block in text.split("- .agpr_count:")[1:]:
            block = ".agpr_count:" + block
            meta = {}
            for key in ("agpr_count", "vgpr_count", "sgpr_count", "private_segment_fixed_size", "group_segment_fixed_size",
                        "max_flat_workgroup_size", "wavefront_size", "vgpr_spill_count", "sgpr_spill_count"):
                mm = re.search(r"\.%s:\s+(\d+)" % key, block)
                if mm:
                    meta[key] = int(mm.group(1))
            name = re.search(r"\.name:\s+(\S+)", block).group(1)
            found[name] = meta
    return found


def _synth(kernels):
    """[(template arguments as a tuple of ints / bools, metadata)] of the synth_kernel instantiations.
    <L, T, WAVES, MINW, STREAM, HALF, ANYBL, NFA, PIPE, FAST, PQP, SPLIT, MID>"""
    out = []
    for name, meta in kernels.items():
        m = re.search(r"synth_kernelI((?:L[ib]\d+E)+)E", name)
        if not m:
            continue
        args = tuple(int(x) for x in re.findall(r"L[ib](\d+)E", m.group(1)))
        assert len(args) == 13, name
        out.append((args, meta))
    return out


def test_every_kernel_family_is_in_the_library(kernels):
    synth = _synth(kernels)
    assert len(synth) >= 60, len(synth)
    have = {(a[0], bool(a[4]), a[7], bool(a[8]), bool(a[9]), bool(a[11]), bool(a[12])) for a, _ in synth}
    for L in (1, 2, 4, 8):
        for nfa in ((4, 8) if L <= 4 else (8,)):
            assert (L, False, nfa, False, False, False, False) in have, ("exact one-shot", L, nfa)
            assert (L, False, nfa, False, True, False, False) in have, ("fast one-shot", L, nfa)
        assert (L, True, 8, False, False, False, False) in have, ("exact stream", L)
        assert (L, True, 8, False, True, False, False) in have, ("fast stream", L)
    for nfa in (4, 8):
        assert (1, False, nfa, False, True, True, False) in have, ("time-split", nfa)
        assert (1, False, nfa, False, True, False, True) in have, ("second tier", nfa)
        assert (1, False, nfa, False, True, True, True) in have, ("second tier, time-split", nfa)
    assert (4, False, 4, True, False, False, False) in have and (8, False, 8, True, False, False, False) in have   # pipelined
    # ... and for blend lengths that are not powers of two (ANYBL): the four-formant lane kernels, the lean stream
    # kernels, the pipelined workgroups with rounds of 32 and of 16 samples
    anybl = {(a[0], bool(a[4]), bool(a[6]), a[7], bool(a[8]), bool(a[9]), a[10]) for a, _ in synth}
    for L in (1, 2, 4):
        assert (L, False, True, 4, False, False, 2) in anybl, ("exact, four formants, any blend", L)
        assert (L, True, True, 4, False, False, 2) in anybl, ("exact stream, four formants, any blend", L)
        assert (L, True, True, 4, False, True, 2) in anybl, ("fast stream, four formants, any blend", L)
    for L, nfa in ((4, 4), (8, 8)):
        for pqp in (8, 4):
            assert (L, False, True, nfa, True, False, pqp) in anybl, ("pipelined, any blend", L, pqp)
    names = " ".join(kernels)
    for other in ("scan_kernel", "ring_append_kernel", "lengths_kernel"):
        assert other in names, other
    for meta in kernels.values():
        assert meta.get("wavefront_size", 64) == 64


def test_lane_kernels_hold_one_wave_per_simd_by_construction(kernels):
    """A SIMD has 512 registers per lane (VGPRs + AGPRs, one file on gfx950): a kernel that holds more than 256 cannot
    share a SIMD with a second wave of itself.  Every lane kernel (all but the pipelined four-wave workgroups, which
    are placed by their LDS footprint) must, because the host sizes every launch for exactly that."""
    two_wave_kernels = []
    for args, meta in _synth(kernels):
        L, T, waves, minw, stream, half, anybl, nfa, pipe = args[:9]
        total = meta["vgpr_count"]          # (the metadata's vgpr_count is the unified total: arch VGPRs + AGPRs)
        assert total <= 512, (args, meta)
        assert meta["agpr_count"] <= total
        if pipe:
            # four waves of a workgroup on the four SIMDs of a CU; LDS decides how many workgroups a CU takes
            assert meta["group_segment_fixed_size"] >= 64 * 1024, (args, meta)
            continue
        if minw == 2:
            # the lane kernels built FOR two waves per SIMD (launches of more waves than the device has SIMDs:
            # csrc/launch_plan.cpp family_cohabits): one-shot, 2 / 4 / 8 lanes per utterance (exact: 2 / 4), and they must
            # really fit twice
            fast = args[9]
            assert not stream and L >= 2 and (L > 2 or nfa == 4) and (fast or L <= 4), args
            assert total <= 256 and meta["agpr_count"] == 0, ("built for two waves per SIMD, does not fit twice", args, meta)
            two_wave_kernels.append(args)
            continue
        assert total > 256, ("two waves of this kernel would fit one SIMD", args, meta)
        assert meta["max_flat_workgroup_size"] == 64 * waves, (args, meta)


def test_two_wave_instantiations_exist(kernels):
    fast = {(a[0], a[7]) for a, _ in _synth(kernels) if a[3] == 2 and a[9]}
    assert fast == {(2, 4), (4, 4), (4, 8), (8, 8)}, fast
    exact = {(a[0], a[7]) for a, _ in _synth(kernels) if a[3] == 2 and not a[9]}
    assert exact == {(2, 4), (4, 4), (4, 8)}, exact


def test_no_kernel_has_a_scratch_segment(kernels):
    """Every synth_kernel instantiation keeps everything in registers (arch VGPRs + AGPRs): a private segment would be HBM
    traffic the roofline does not count — and, where the register allocator puts it into the per-tile path, time (the
    eight-formant one-lane fast kernels held up to 2 KB per lane while the tolerance-mode loop had three code paths, and
    config 4 fast took 41.9 ms instead of 23; with one path none does: DESIGN.md section 4.3)."""
    checked = 0
    for args, meta in _synth(kernels):
        assert meta.get("private_segment_fixed_size", 0) == 0, (args, meta)
        checked += 1
    assert checked >= 60, checked
