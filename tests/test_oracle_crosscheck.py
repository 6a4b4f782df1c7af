"""Two independent restatements of reference src/lib.rs — the C oracle and the numpy
generator model (tests/np_model.py) — must agree bit for bit.  Two readings of the same
source catch transcription slips that one reading cannot."""
import numpy as np
import pytest

import np_model as M
import oracle_lib as O

A, E, S, ST, GL = O.PH_A, O.PH_E, O.PH_SILENCE, O.PH_STOP, O.PH_GLIDE


def both(voice, segs, seed):
    out, n = O.synthesize_phonemes(voice, O.segments(segs), seed)
    ref = M.render(voice, segs, seed)
    assert n == len(out)
    return out, ref


CASES = {
    "fade_in_out": [(S, .01, .01, 0.0027), (A, .01, .01, 0.0027)],
    "a_e_a": [(A, .012, .012, 0.003), (E, .012, .012, 0.0035), (A, .012, .012, 0.0025)],
    "all_silent": [(S, .005, .005, 0.1), (ST, .005, .005, 0.1), (GL, .005, .005, 0.1)],
    "blend_ne_length": [(A, .02, .004, 0.003), (E, .01, .03, 0.004)],
    "zero_blend": [(A, .005, 0.0, 0.003), (E, .005, .005, 0.003)],
    "short_segments": [(A, 1e-6, .01, 0.003), (E, 1e-6, .01, 0.003), (A, .005, .005, 0.003)],
    "negative_length": [(A, -1.0, .01, 0.003), (E, .01, .01, 0.003)],
    "pitch_clamp": [(A, .005, .005, 0.7), (E, .005, .005, 0.5)],
    "zero_pitch": [(A, .005, .005, 0.0), (E, .005, .005, 1e-9)],
    "single": [(E, .01, .01, 0.004)],
    "empty": [],
}


@pytest.mark.parametrize("name", sorted(CASES))
@pytest.mark.parametrize("rate", [None, 48000.0])
def test_c_oracle_equals_numpy_model(name, rate):
    v = O.voice_generic(rate)
    for seed in (0, 977):
        with np.errstate(all="ignore"):
            out, ref = both(v, CASES[name], seed)
        assert len(out) == len(ref), (name, len(out), len(ref))
        a, b = out.view(np.uint32), ref.view(np.uint32)
        if not np.array_equal(a, b):
            i = int(np.argmax(a != b))
            raise AssertionError(f"{name} seed {seed}: first difference at sample {i}: "
                                 f"{out[i]!r} vs {ref[i]!r}")


def test_jitter_wrap_is_exercised_and_agrees():
    # jitter_frequency 16/44100 wraps the noise phase every ~2756 samples (:245, :294)
    v = O.voice_generic()
    segs = [(A, .08, .08, v.center_frequency), (E, .08, .08, v.center_frequency)]
    with np.errstate(all="ignore"):
        out, ref = both(v, segs, 42)
    assert len(out) > 2 * 2756
    assert np.array_equal(out.view(np.uint32), ref.view(np.uint32))


def test_per_sample_elems_agree():
    v = O.voice_generic(48000.0)
    segs = [(S, .006, .006, 0.003), (A, .006, .003, 0.003), (E, .006, .009, 0.004)]
    for stage in (0, 1):
        got = O.trace_elems(v, O.segments(segs), 5, stage)
        ref = M.trace(v, segs, 5, stage)
        assert got.shape == ref.shape
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


def random_voice(rng, rate):
    """A voice with random phoneme tables (through the crate's own constructor arithmetic,
    SynthesisElem::new_phoneme :381-401, resample :418-440) and random jitter settings."""
    import ctypes as C
    L = O.lib()
    L.orc_elem_new_phoneme.restype = None
    L.orc_elem_new_phoneme.argtypes = [C.POINTER(O.SynthesisElem)] + [C.POINTER(C.c_float)] * 6
    v = O.voice_generic(rate)
    for p in range(2):
        amp = rng.uniform(0, 1, 8) * (rng.uniform(0, 1, 8) > 0.3)
        amp[0] = max(amp[0], 0.1)
        arrs = [np.ascontiguousarray(a, dtype=np.float32) for a in (
            rng.uniform(200, 5000, 8), rng.uniform(30, 400, 8), rng.uniform(500, 6000, 8),
            rng.uniform(0, 1, 8), rng.uniform(0, 1, 8), amp)]      # freq, bw, smooth, turb, breath, amp
        e = O.SynthesisElem()
        L.orc_elem_new_phoneme(C.byref(e), *[a.ctypes.data_as(C.POINTER(C.c_float)) for a in arrs])
        if float(v.sample_rate) != 44100.0:
            L.orc_elem_resample(C.byref(e), 44100.0, float(v.sample_rate))
        v.phonemes[p] = e
    sr = np.float32(v.sample_rate)
    v.jitter_frequency = float(np.float32(rng.uniform(4, 400)) / sr)
    v.jitter_delta_frequency = float(np.float32(rng.uniform(0, 20)) / sr)
    v.jitter_delta_formant_frequency = float(np.float32(rng.uniform(0, 60)) / sr)
    v.jitter_delta_amplitude = float(np.float32(rng.uniform(0, 0.9)))
    return v


@pytest.mark.parametrize("seed", range(6))
def test_random_voices_and_segments_agree(seed):
    """Randomised: random phoneme tables, jitter settings, segment lists (every phoneme kind,
    lengths and blend lengths, pitches) — the C oracle and the numpy model read the reference
    independently and must still agree on every bit."""
    rng = np.random.default_rng(1000 + seed)
    rate = float(rng.choice([44100.0, 48000.0, 22050.0]))
    v = random_voice(rng, rate)
    segs = []
    for _ in range(int(rng.integers(1, 6))):
        ph = int(rng.choice([A, E, S, ST, GL], p=[.4, .35, .15, .05, .05]))
        blend = float(rng.choice([rng.uniform(0.0005, 0.02), 2.0 ** -7]))
        segs.append((ph, float(rng.uniform(0.0005, 0.012)), blend,
                     float(np.float32(rng.uniform(60, 500)) / np.float32(rate))))
    with np.errstate(all="ignore"):
        out, ref = both(v, segs, int(rng.integers(0, 2 ** 32)))
    assert len(out) == len(ref) and len(out) > 0
    a, b = out.view(np.uint32), ref.view(np.uint32)
    if not np.array_equal(a, b):
        i = int(np.argmax(a != b))
        raise AssertionError(f"seed {seed}: first difference at sample {i}: {out[i]!r} vs {ref[i]!r}")
