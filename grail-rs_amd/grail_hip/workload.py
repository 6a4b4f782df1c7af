"""Deterministic synthetic workloads of SURVEY.md §8d / BASELINE.json configs.

Inputs only (segment lists, seeds, voice presets) — no synthesis arithmetic.
Used by bench.py and the parity tests so both sides see identical inputs.
"""
import numpy as np

from . import (NUM_FORMANTS, PH_A, PH_E, PH_SILENCE, PH_STOP, PHONEME_DTYPE, elem_new_phoneme,
               elem_resample, shard_range, voice_generic)

SAMPLE_RATE = 48000.0
SEGMENTS_PER_UTT = 4


def _lcg(s):
    # the crate's own generator state update, reference src/lib.rs:40
    return (s * np.uint32(16807) + np.uint32(1)).astype(np.uint32)


def make_batch(n_utt, first_utt=0, n_voices=1, segments=SEGMENTS_PER_UTT, sample_rate=SAMPLE_RATE,
               length=0.5, blend_length=0.5):
    """Utterances [first_utt, first_utt + n_utt) of the synthetic corpus.

    Utterance u: `segments` segments; segment 0 is Silence (what .transcribe()
    always emits first, reference src/lib.rs:1201), the others draw from
    (A, E, Silence); length/blend_length are the Intonator's constants
    (src/lib.rs:1070-1071); pitch 100..200 Hz; jitter seed = u; voice = u mod n_voices.
    Returns (segs, seg_offsets, voice_ids, jitter_seeds).
    """
    with np.errstate(over="ignore"):
        u = (np.arange(n_utt, dtype=np.uint64) + np.uint64(first_utt)).astype(np.uint32)
        s = (u ^ np.uint32(0x9E3779B9)).astype(np.uint32)
        segs = np.zeros((n_utt, segments), dtype=PHONEME_DTYPE)
        choices = np.array([PH_A, PH_E, PH_SILENCE], dtype=np.int32)
        for i in range(segments):
            s = _lcg(s)
            ph = choices[(s >> np.uint32(16)) % np.uint32(3)]
            if i == 0:
                ph = np.full(n_utt, PH_SILENCE, dtype=np.int32)
            s = _lcg(s)
            hz = (np.uint32(100) + (s >> np.uint32(16)) % np.uint32(101)).astype(np.float32)
            segs["phoneme"][:, i] = ph
            segs["length"][:, i] = np.float32(length)
            segs["blend_length"][:, i] = np.float32(blend_length)
            segs["frequency"][:, i] = hz / np.float32(sample_rate)
    seg_offsets = (np.arange(n_utt + 1, dtype=np.uint64) * segments).astype(np.uint32)
    voice_ids = (u % np.uint32(max(n_voices, 1))).astype(np.uint32)
    jitter_seeds = u.copy()
    return segs.reshape(-1), seg_offsets, voice_ids, jitter_seeds


def speech_like_batch(n_utt, rng, n_voices=1, scale=1.0, blend_is_length=False, long_tail=False,
                      sample_rate=SAMPLE_RATE):
    """A SPEECH-LIKE corpus next to make_batch's four aligned segments: utterances of 8 - 32 phonemes of 40 - 160 ms
    (blends of 30 - 80 ms, any length; pitches of 90 - 220 Hz per phoneme; A / E / Silence / Stop 40 / 40 / 12 / 8 %,
    a leading Silence as .transcribe() emits it, reference src/lib.rs:1201), 2.0 s on average, 0.5 - 3.8 s: the
    utterances differ in length by a factor of seven and every one has a segment boundary and the kink of
    alpha = min(time / blend_length, 1) (src/lib.rs:899) every few thousand samples at times of its own.
    `scale` multiplies every length and blend length (0.1: phonemes of 4 - 16 ms); `blend_is_length`: no flat
    stretch of alpha and no kink; `long_tail`: one utterance in a hundred of 60 - 80 phonemes among utterances of 4 - 12.
    `rng`: a numpy Generator (the draws, in this order: counts, phonemes, lengths, blend lengths, pitches).
    Returns (segs, seg_offsets, voice_ids, jitter_seeds, out_stride)."""
    counts = rng.integers(8, 33, n_utt)
    if long_tail:
        counts = np.where(rng.random(n_utt) < 0.01, rng.integers(60, 81, n_utt), rng.integers(4, 13, n_utt))
    offs = np.zeros(n_utt + 1, dtype=np.uint32)
    offs[1:] = np.cumsum(counts)
    k = int(offs[-1])
    segs = np.zeros(k, dtype=PHONEME_DTYPE)
    segs["phoneme"] = rng.choice([PH_A, PH_E, PH_SILENCE, PH_STOP], k, p=[.4, .4, .12, .08])
    segs["phoneme"][offs[:-1]] = PH_SILENCE
    segs["length"] = (rng.uniform(0.04, 0.16, k) * scale).astype(np.float32)
    segs["blend_length"] = (rng.uniform(0.03, 0.08, k) * scale).astype(np.float32)
    if blend_is_length:
        segs["blend_length"] = segs["length"]
    segs["frequency"] = (rng.uniform(90, 220, k) / sample_rate).astype(np.float32)
    vids = (np.arange(n_utt) % max(n_voices, 1)).astype(np.uint32)
    seeds = np.arange(n_utt, dtype=np.uint32)
    stride = (int(int(counts.max()) * 0.16 * scale * sample_rate) + 64 + 63) // 64 * 64
    return segs, offs, vids, seeds, stride


def shard_inputs(utts_per_rank, rank, world, n_voices, **kw):
    """This rank's slice of the global synthetic corpus of utts_per_rank*world utterances
    (contiguous shards, no data-path collective)."""
    first, last = shard_range(utts_per_rank * world, rank, world)
    return (first, last) + make_batch(last - first, first_utt=first, n_voices=n_voices, **kw)


# BASELINE.json config 4: 8 presets with divergent formant coefficients.  The
# reference ships one voice (src/voices/mod.rs:17-20); these are the build's own:
# generic()'s phoneme tables with the formant frequencies scaled (vocal-tract
# length), a different centre pitch, and all eight formant amplitudes live.
PRESET_FREQ_SCALE = [0.80, 0.87, 0.94, 1.00, 1.07, 1.14, 1.21, 1.28]
PRESET_PITCH_HZ = [90.0, 110.0, 120.0, 140.0, 165.0, 190.0, 220.0, 250.0]
PRESET_AMP = [0.3, 0.25, 0.15, 0.1, 0.08, 0.06, 0.04, 0.02]

# src/voices/generic.rs:9-32 raw tables (Hz), MKPHON order freq, bw, smooth, turb, breath
_GENERIC_RAW = {
    "a": dict(freq=[910.0, 1271.0, 2851.0, 3213.0, 1200.0, 2000.0, 3000.0, 4000.0],
              bw=[60.0, 160.0, 180.0, 200.0, 100.0, 100.0, 100.0, 100.0],
              smooth=[1600.0] * 8,
              turb=[0.2, 0.2, 0.1, 0.0, 0.0, 0.0, 0.0, 0.0],
              breath=[0.5, 0.2, 0.05, 0.0, 0.0, 0.0, 0.0, 0.0]),
    "e": dict(freq=[910.0, 1871.0, 2851.0, 3213.0, 1200.0, 2000.0, 3000.0, 4000.0],
              bw=[80.0, 180.0, 180.0, 200.0, 100.0, 100.0, 100.0, 100.0],
              smooth=[1600.0] * 8,
              turb=[0.2, 0.4, 0.4, 0.4, 0.4, 0.4, 0.4, 0.4],
              breath=[1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 0.1, 0.1]),
}


def preset_voices(n=8, sample_rate=SAMPLE_RATE):
    voices = []
    for k in range(n):
        v = voice_generic(sample_rate)
        scale = np.float32(PRESET_FREQ_SCALE[k % 8])
        for p, name in enumerate(("a", "e")):
            raw = _GENERIC_RAW[name]
            freq = (np.array(raw["freq"], dtype=np.float32) * scale).astype(np.float32)
            e = elem_new_phoneme(freq, raw["bw"], raw["smooth"], raw["turb"], raw["breath"],
                                 PRESET_AMP)
            v.phonemes[p] = elem_resample(e, 44100.0, sample_rate)
        v.center_frequency = float(np.float32(PRESET_PITCH_HZ[k % 8]) / np.float32(sample_rate))
        voices.append(v)
    return voices


def single_voice(sample_rate=SAMPLE_RATE):
    return [voice_generic(sample_rate)]


def max_samples(segments=SEGMENTS_PER_UTT, length=0.5, sample_rate=SAMPLE_RATE):
    """A safe out_stride for make_batch's utterances: the f32 Sequencer clock yields a few
    samples more than length*rate per segment.  Rounded up to 64 samples (256 B) so every
    64-sample tile the kernel flushes is one aligned 256-B run in HBM (no partial lines)."""
    n = int(np.ceil(segments * length * sample_rate)) + 4 * segments + 8
    return (n + 63) // 64 * 64


assert NUM_FORMANTS == 8


def tame_voice(voice, limit=None):
    """Widens every bandwidth of `voice` (x 1.25 at a time) until fast arithmetic is served for it
    (grail_fast_sharpness <= the limit): random tables for the fuzz tests that sit just below the limit."""
    from . import fast_sharpness, FAST_SHARPNESS_LIMIT
    limit = FAST_SHARPNESS_LIMIT if limit is None else limit
    for _ in range(64):
        if fast_sharpness(voice) <= limit:
            break
        for p in range(len(voice.phonemes)):
            for i in range(NUM_FORMANTS):
                voice.phonemes[p].formant_bw[i] = np.float32(voice.phonemes[p].formant_bw[i] * 1.25)
    return voice
