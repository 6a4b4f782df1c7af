"""grail_length_bound (include/grail_hip.h): an upper bound of an utterance's length in samples from its segment lengths
and its voice's sample rate.  The time-split kernels skip a chunk whose utterance cannot reach it by this bound, so it must
hold for the reference's f32 clock as it is — a long segment at a high sample rate lasts noticeably longer than
length * sample_rate samples (the clock's rounding) — against the oracle's own count, on every kind of row the fuzz tests
build plus the hostile ones: zero, sub-sample and two-sample segments, segments of 2 - 20 s, sample rates of 8 - 192 kHz."""
import numpy as np
import pytest

import grail_hip as G
import oracle_lib as O
from grail_hip import workload as W
from test_fuzz_gpu import random_batch


def _bounds(segs, offs, rate):
    return np.array([G.length_bound(segs["length"][offs[u]:offs[u + 1]], rate) for u in range(len(offs) - 1)], dtype=np.float64)


@pytest.mark.parametrize("rate", [8000.0, 44100.0, 48000.0, 192000.0])
def test_the_bound_covers_the_oracles_lengths(built, rate):
    voices = [O.Voice.from_buffer_copy(bytes(G.voice_generic(rate)))]
    tightest, loosest = 1e18, 0.0
    for seed in range(4):
        rng = np.random.default_rng(seed)
        for kind in ("fuzz", "speech", "tiny", "long"):
            if kind == "fuzz":
                segs, offs, _, seeds = random_batch(rng, 200, 1, rate)
            elif kind == "speech":
                segs, offs, _, seeds, _ = W.speech_like_batch(120, rng, scale=float(rng.choice([1.0, 0.3, 0.1])), sample_rate=rate)
            elif kind == "tiny":
                segs, offs, _, seeds = random_batch(rng, 200, 1, rate)
                segs["length"] = rng.choice([0.0, 1e-7, 0.5 / rate, 1.0 / rate, 1.5 / rate, 2.0 / rate, 3.3 / rate], len(segs)).astype(np.float32)
            else:
                segs, offs, _, seeds = W.make_batch(4, segments=3, length=8.0)
                segs["length"] = rng.uniform(2.0, 20.0, len(segs)).astype(np.float32)
            true = O.count_batch(voices, segs, offs, None, seeds).astype(np.float64)
            bound = _bounds(segs, offs, rate)
            assert np.all(bound >= true), (kind, seed, float((bound - true).min()))
            used = true > 0
            if used.any():
                tightest = min(tightest, float((bound[used] - true[used]).min()))
                loosest = max(loosest, float((bound[used] / true[used]).max()))
    print(f"{rate:.0f} Hz: bound - true >= {tightest:.0f} samples, bound / true <= {loosest:.3f}")


def test_rows_without_a_bound(built):
    assert G.length_bound([0.5, float("nan")], 48000.0) is None
    assert G.length_bound([0.5, float("inf")], 48000.0) is None
    assert G.length_bound([0.5], 0.0) is None
    assert G.length_bound([3.0e9], 48000.0) is None          # (a clock that may not move: dt <= ulp(length))
    assert G.length_bound([], 48000.0) == 0
    assert G.length_bound([-1.0, 0.0], 48000.0) == 4         # (two samples per segment that leaves the clock negative)
    # a second at 48 kHz: 48 000 samples and a little
    assert 48000 <= G.length_bound([1.0], 48000.0) <= 48150
