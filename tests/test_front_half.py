"""The host front half of the product (csrc/text_front.cpp): Transcriber, Intonator,
languages::generic(), the RIFF writer — against the reference's six literal Transcriber tests
(src/lib.rs:1210-1358), against the oracle on random inputs, and the header spec of
examples/cli.rs:28-67."""
import os
import struct

import numpy as np
import pytest

import grail_hip as G
import oracle_lib as O

A, E, SIL = G.PH_A, G.PH_E, G.PH_SILENCE

REFERENCE_TESTS = [   # (name, text, rules, expected) — literal from the reference
    ("transcribe_unique", "abc", [("ab", [A]), ("c", [E])], [A, E]),                       # :1210
    ("transcribe_same_start", "abacab", [("ab", [A]), ("ac", [E])], [A, E, A]),            # :1233
    ("transcribe_same_char_different_length", "aaa", [("a", [A]), ("aa", [E])], [E, A]),   # :1257
    ("transcribe_same_char_different_length_cutoff", "ae",
     [("a", [A]), ("aa", [E]), ("e", [E])], [A, E]),                                        # :1282
    ("transcribe_skip_no_matches", "abuac", [("ab", [A]), ("ac", [E])], [A, SIL, E]),      # :1310
    ("transcribe_skip_partial_match_at_end", "abaca", [("ab", [A]), ("ac", [E])], [A, E, SIL]),  # :1335
]


@pytest.mark.parametrize("name,text,rules,want", REFERENCE_TESTS, ids=[t[0] for t in REFERENCE_TESTS])
def test_reference_transcriber_tests(built, name, text, rules, want):
    assert G.transcribe(text, rules) == want


def test_language_generic_table(built):
    rules, cs = G.language_generic()
    assert cs is False
    assert rules == [("a", [A]), ("e", [E]), ("i", [A]), ("ii", [E, A]), ("oui", [A, E, A]),
                     ("p", [SIL])]
    assert [r[0] for r in rules] == sorted(r[0] for r in rules)  # binary search needs sorted rules


def test_transcriber_matches_oracle_on_random_text(built):
    rng = np.random.default_rng(3)
    rules, _ = G.language_generic()
    extra = [("ab", [A]), ("abc", [E, E]), ("b", [SIL]), ("ba", [A, E]), ("c", [E]), ("cab", [A])]
    for rs in (rules, sorted(extra)):
        for _ in range(300):
            n = int(rng.integers(0, 12))
            text = "".join(rng.choice(list("aeioupbcAEI x"), n))
            for lead in (False, True):
                got = G.transcribe(text, rs, leading_silence=lead)
                want = O.transcribe(text, rs, leading_silence=lead)
                assert got == want, (text, lead, got, want)


def test_text_to_phoneme_elems_is_transcribe_then_intonate(built):
    v = G.voice_generic()
    pe = G.text_to_phoneme_elems(v, "aEi oui")
    rules, _ = G.language_generic()
    want = G.transcribe("aEi oui", rules, leading_silence=True)
    assert list(pe["phoneme"]) == want and want[0] == SIL          # src/lib.rs:1201
    assert np.all(pe["length"] == np.float32(0.5)) and np.all(pe["blend_length"] == np.float32(0.5))
    assert np.all(pe["frequency"] == np.float32(v.center_frequency))  # src/lib.rs:1068-1073
    assert len(G.text_to_phoneme_elems(v, "")) == 1                # just the leading Silence
    assert list(G.text_to_phoneme_elems(v, "éa")["phoneme"]) == [SIL, SIL, A]  # non-ASCII char: no rule


def test_wav_header_matches_cli_save_wav(built, tmp_path):
    pcm = np.array([0, 1, -1, 32767, -32768, 1234], dtype=np.int16)
    path = str(tmp_path / "t.wav")
    G.wav_write_i16(path, pcm, 44100)
    raw = open(path, "rb").read()
    assert len(raw) == 44 + 2 * len(pcm)
    assert raw[:4] == b"RIFF" and raw[8:16] == b"WAVEfmt " and raw[36:40] == b"data"
    size, = struct.unpack("<I", raw[4:8])
    fmt_len, fmt, ch, rate, brate, align, bits = struct.unpack("<IHHIIHH", raw[16:36])
    dlen, = struct.unpack("<I", raw[40:44])
    assert (size, fmt_len, fmt, ch, rate, brate, align, bits, dlen) == (
        36 + 2 * len(pcm), 16, 1, 1, 44100, 88200, 2, 16, 2 * len(pcm))
    assert np.array_equal(np.frombuffer(raw[44:], dtype="<i2"), pcm)


@pytest.mark.gpu
def test_say_batch_equals_the_cli_chain(gpu_ctx):
    """text -> PCM through the C ABI == the oracle's examples/cli.rs:175-184 chain."""
    v = G.voice_generic()
    gpu_ctx.set_voices([v])
    texts = ["a", "ae", "", "oui", "xyz"]
    out, out_len = gpu_ctx.say(texts)
    ov = O.voice_generic()
    for i, t in enumerate(texts):
        ref = O.say(ov, t)
        assert out_len[i] == len(ref), (t, out_len[i], len(ref))
        assert np.array_equal(out[i, :len(ref)].view(np.uint32), ref.view(np.uint32)), t


@pytest.mark.gpu
def test_pcm16_kernel_matches_rust_as_cast(gpu_ctx):
    rng = np.random.default_rng(9)
    n_utt, stride = 5, 4104
    x = (rng.standard_normal((n_utt, stride)) * 0.6).astype(np.float32)
    x[0, :12] = [0.0, -0.0, 1.0, -1.0, 2.0, -2.0, np.nan, np.inf, -np.inf, 0.99999, 1e-9, -3e-5]
    lens = np.array([4104, 4097, 1, 0, 2049], dtype=np.uint32)
    d_in = gpu_ctx.device_alloc(x.nbytes)
    d_len = gpu_ctx.device_alloc(lens.nbytes)
    d_out = gpu_ctx.device_alloc(n_utt * stride * 2)
    try:
        gpu_ctx.h2d(d_in, x, x.nbytes)
        gpu_ctx.h2d(d_len, lens, lens.nbytes)
        gpu_ctx.memset(d_out, 0x55, n_utt * stride * 2)
        gpu_ctx.pcm16(d_in, stride, d_len, n_utt, int(lens.max()), d_out, stride)
        gpu_ctx.sync()
        got = np.zeros((n_utt, stride), dtype=np.int16)
        gpu_ctx.d2h(got, d_out, got.nbytes)
    finally:
        for p in (d_in, d_len, d_out):
            gpu_ctx.device_free(p)
    L = O.lib()
    for u in range(n_utt):
        want = np.array([L.orc_pcm16(float(v)) if not np.isnan(v) else 0 for v in x[u, :lens[u]]],
                        dtype=np.int16)
        assert np.array_equal(got[u, :lens[u]], want), u
        assert np.all(got[u, lens[u]:] == 0x5555)   # beyond the row's length: untouched


@pytest.mark.gpu
def test_cpp_cli_example_writes_the_reference_wav(gpu_ctx, tmp_path):
    """examples/grail_say.cpp (include/grail.hpp) — text in, WAV out — byte-for-byte what the
    reference's CLI would write: save_wav header (examples/cli.rs:28-67) + `as i16` samples of
    the oracle's rendering of the same text."""
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                       "grail-rs_amd", "lib", "grail_say")
    path = str(tmp_path / "a.wav")
    r = subprocess.run([exe, "-o", path, "ae"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "seconds of audio, generated in" in r.stdout
    raw = open(path, "rb").read()
    ref = O.say(O.voice_generic(), "ae")
    L = O.lib()
    want = np.array([L.orc_pcm16(float(v)) for v in ref], dtype="<i2")
    assert len(raw) == 44 + 2 * len(want)
    assert struct.unpack("<I", raw[24:28])[0] == 44100
    assert np.array_equal(np.frombuffer(raw[44:], dtype="<i2"), want)


@pytest.mark.gpu
def test_one_call_pcm16_rows(gpu_ctx):
    from grail_hip import workload as W
    voices = W.single_voice()
    gpu_ctx.set_voices(voices)
    segs, offs, vids, seeds = W.make_batch(40, length=0.02, blend_length=0.02)
    stride = W.max_samples(length=0.02)
    f32, n = gpu_ctx.synthesize(segs, offs, vids, seeds, out_stride=stride)
    i16, n2 = gpu_ctx.synthesize_pcm16(segs, offs, vids, seeds, out_stride=stride)
    assert np.array_equal(n, n2)
    L = O.lib()
    for u in range(40):
        want = np.array([L.orc_pcm16(float(v)) for v in f32[u, :n[u]]], dtype=np.int16)
        assert np.array_equal(i16[u, :n[u]], want), u
        assert np.all(i16[u, n[u]:] == 0)


@pytest.mark.gpu
@pytest.mark.parametrize("n_voices", [8, 1])
@pytest.mark.parametrize("lanes", [1, 2, 4, 8])
def test_fused_pcm16_store_every_lane_mapping(gpu_ctx, lanes, n_voices):
    """grail_batch_synthesize_pcm16_async: the i16 rows written by the synthesis kernel's own flush
    are the examples/cli.rs:49 conversion of the oracle's f32 samples; ragged lengths, a row
    stride that forces the scalar tail, and a truncating stride."""
    import grail_hip as G
    from grail_hip import workload as W
    voices = W.preset_voices(8) if n_voices == 8 else W.single_voice()     # (eight live formants / the four-formant kernels)
    gpu_ctx.set_voices(voices)
    n_utt = 70
    segs, offs, vids, seeds = W.make_batch(n_utt, n_voices=n_voices, length=0.013, blend_length=0.008)
    ov = [O.Voice.from_buffer_copy(bytes(v)) for v in voices]
    full = W.max_samples(length=0.013)
    ref, ref_len = O.synthesize_batch(ov, segs, offs, vids, seeds, full)
    L = O.lib()
    batch = gpu_ctx.upload(segs, offs, vids, seeds)
    gpu_ctx.set_option("lanes_per_utterance", lanes)
    try:
        for stride in (full, full + 2, full + 1, 1001):          # aligned / unaligned rows (even, odd) / truncating
            d_out = gpu_ctx.device_alloc(n_utt * stride * 2 + 16)
            d_len = gpu_ctx.device_alloc(n_utt * 4)
            try:
                gpu_ctx.memset(d_out, 0x5A, n_utt * stride * 2)
                batch.synthesize_pcm16_async(d_out, stride, d_len)
                try:
                    gpu_ctx.sync()
                except G.GrailError as e:
                    assert stride == 1001 and e.status == G.ERR_BUFFER_TOO_SMALL
                got = np.zeros((n_utt, stride), dtype=np.int16)
                lens = np.zeros(n_utt, dtype=np.uint32)
                gpu_ctx.d2h(got, d_out, got.nbytes)
                gpu_ctx.d2h(lens, d_len, lens.nbytes)
            finally:
                gpu_ctx.device_free(d_out)
                gpu_ctx.device_free(d_len)
            for u in range(n_utt):
                n = min(int(ref_len[u]), stride)
                assert lens[u] == n, (stride, u)
                want = np.array([L.orc_pcm16(float(v)) for v in ref[u, :n]], dtype=np.int16)
                assert np.array_equal(got[u, :n], want), (stride, u)
                assert np.all(got[u, n:] == 0x5A5A), (stride, u)   # nothing past the row's end
    finally:
        gpu_ctx.set_option("lanes_per_utterance", 0)
        batch.free()


# (examples/grail_interactive.cpp — ONE chain per session on a live stream — is checked against the oracle in
# tests/test_live_stream_gpu.py::test_interactive_example_is_one_chain_for_the_whole_session)
