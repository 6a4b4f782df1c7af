"""Host logic of the fast kernels (no GPU): the warm-up length of a voice and the chunk grid of the time-split
kernels, and the sharpness up to which fast arithmetic is served at all.

grail_time_split_warmup / grail_time_split_grid are the functions the library itself plans its launches with
(voice_analysis.cpp: voice_warmup, split_grid); here they are checked against a numpy restatement of the filter decay of
Synthesize::next (src/lib.rs:530-575) and against the cost model the grid is meant to balance.
"""
import math

import numpy as np
import pytest

import grail_hip as gh

pytestmark = pytest.mark.skipif(not gh.lib_exists(), reason="libgrail_hip.so not built")


def tan_approx(f):
    # src/lib.rs:595-601
    return ((1 - f) * f * (5 - 4 * (f + 0.5) * (0.5 - f))) / ((f + 0.5) * (5 - 4 * (1 - f) * f) * (0.5 - f))


def slowest_decay(voice):
    """Smallest per-sample decay rate (-ln |pole|) over the audible formants of every phoneme."""
    slow = math.inf
    for i in range(8):
        if all(voice.phonemes[p].formant_amp[i] == 0.0 for p in range(len(voice.phonemes))):
            continue
        for p in range(len(voice.phonemes)):
            e = voice.phonemes[p]
            f, w, sm = e.formant_freq[i], e.formant_bw[i], e.formant_smooth[i]
            g, k = tan_approx(f), w / f
            # the SVF's transition matrix (trapezoidal, :555-571): spectral radius by numpy
            a1 = 1.0 / (1.0 + g * (g + k))
            a2 = g * a1
            a3 = g * a2
            m = np.array([[2 * a1 - 1, -2 * a2], [2 * a2, 1 - 2 * a3]])
            slow = min(slow, -math.log(max(abs(np.linalg.eigvals(m)))))
            slow = min(slow, -5.0 * math.log1p(-sm))
    return slow


@pytest.mark.parametrize("rate", [16000.0, 22050.0, 44100.0, 48000.0])
def test_warmup_covers_the_slowest_decay(rate):
    v = gh.voice_generic(rate)
    w = gh.time_split_warmup(v)
    assert w > 0 and w % 64 == 0 and w <= 16384
    slow = slowest_decay(v)
    # after w samples the slowest mode has decayed below 2^-21, with the 5 % margin and no more than a tile over
    assert math.exp(-slow * w) <= 2.0 ** -21
    need = math.log(2.0 ** 21) / (0.95 * slow)
    assert need <= w < need + 64


def test_warmup_of_the_headline_voice():
    assert gh.time_split_warmup(gh.voice_generic(48000.0)) == 3904     # include/grail_hip.h, DESIGN.md 4.5


def test_warmup_scales_with_the_sample_rate():
    ws = [gh.time_split_warmup(gh.voice_generic(r)) for r in (16000.0, 32000.0, 48000.0)]
    assert ws[0] < ws[1] < ws[2]


def test_warmup_rejects_unqualified_voices():
    v = gh.voice_generic(48000.0)
    v.phonemes[1].formant_bw[1] = 0.0          # a band-pass that never forgets
    assert gh.time_split_warmup(v) == 0
    v = gh.voice_generic(48000.0)
    v.phonemes[0].formant_freq[0] = 0.6        # outside (0, 0.5)
    assert gh.time_split_warmup(v) == 0
    v = gh.voice_generic(48000.0)
    v.phonemes[1].formant_bw[0] = 1e-7         # qualifies in kind, but the warm-up would exceed 16384 samples
    assert gh.time_split_warmup(v) == 0
    v = gh.voice_generic(48000.0)
    v.phonemes[0].formant_smooth[2] = float("nan")
    assert gh.time_split_warmup(v) == 0


def test_warmup_of_a_silent_voice():
    v = gh.voice_generic(48000.0)
    for p in range(len(v.phonemes)):
        for i in range(8):
            v.phonemes[p].formant_amp[i] = 0.0
    assert gh.time_split_warmup(v) == 64


def lane_cost(bounds, end, warmup, r):
    out = []
    for k, b in enumerate(bounds):
        nxt = bounds[k + 1] if k + 1 < len(bounds) else end
        before = r * max(b - warmup, 0) + min(warmup, b) if k else 0.0
        out.append(before + (nxt - b))
    return out


@pytest.mark.parametrize("span,warmup,chunks,permille", [
    (96064, 3904, 16, 165), (96064, 3904, 8, 165), (96064, 3904, 4, 132), (96064, 3904, 2, 165),
    (48000, 1344, 8, 165), (400000, 3904, 24, 165), (96064, 0, 16, 0), (1 << 22, 64, 64, 10),
])
def test_grid_balances_the_lanes(span, warmup, chunks, permille):
    b = gh.time_split_grid(span, warmup, chunks, permille)
    assert len(b) == chunks and b[0] == 0
    assert all(x % 64 == 0 for x in b)
    assert all(b[k] < b[k + 1] for k in range(chunks - 1)) and b[-1] < span
    cost = lane_cost(b, span, warmup, permille * 1e-3)
    # equal within the rounding of the bounds to tiles of 64 (each bound moves a lane by at most 32 + 32)
    assert max(cost) - min(cost) <= 2 * 64 + 1
    # and the chunks shrink along the utterance while fast-forwarding costs anything
    lens = [y - x for x, y in zip(b, b[1:] + [span])]
    if permille and chunks > 2:
        assert lens[1] >= lens[-1]


def test_grid_without_costs_is_even():
    b = gh.time_split_grid(65536, 0, 16, 0)
    assert b == [4096 * k for k in range(16)]


def test_grid_refuses_what_does_not_fit():
    with pytest.raises(gh.GrailError):
        gh.time_split_grid(4096, 3904, 16, 165)       # the warm-up alone outweighs a chunk
    with pytest.raises(gh.GrailError):
        gh.time_split_grid(8192, 64, 2, 1000)         # fast-forwarding as dear as rendering: nothing to gain
    with pytest.raises(gh.GrailError):
        gh.time_split_grid(96064, 3904, 1, 165)
    with pytest.raises(gh.GrailError):
        gh.time_split_grid(96064, 3904, 65, 165)
    with pytest.raises(gh.GrailError):
        gh.time_split_grid(96064, 3904, 16, 1001)


def test_grid_is_what_the_headline_mid_range_launch_uses():
    # 4096 utterances x 2 s at 48 kHz: 16 chunks (65536 lanes) over 96000 + 64 samples
    b = gh.time_split_grid(96064, 3904, 16, 165)
    assert b[1] > 96064 // 16          # the first lane has nothing to fast-forward: the longest chunk
    # the last lane fast-forwards over 19 of 20 parts and warms up: about a thousand samples are left to render
    assert 512 <= 96064 - b[-1] <= 2048
    # more chunks buy little (the fast-forward of the last lane is the floor) and 32 no longer fit
    assert gh.time_split_grid(96064, 3904, 24, 165)[1] > 0.95 * b[1]
    with pytest.raises(gh.GrailError):
        gh.time_split_grid(96064, 3904, 32, 165)


# ---- grail_fast_sharpness: which voices fast arithmetic is served for

def sharpness(voice):
    """include/grail_hip.h: E_i = share_i (0.0709 / bw_i) (1 + (f_i / 0.075)^2), S = sqrt(sum E_i^2)."""
    n = len(voice.phonemes)
    amps = np.array([[abs(voice.phonemes[p].formant_amp[i]) for i in range(8)] for p in range(n)], dtype=np.float64)
    share = (amps / amps.sum(axis=1, keepdims=True)).max(axis=0)
    f = np.array([[voice.phonemes[p].formant_freq[i] for i in range(8)] for p in range(n)], dtype=np.float64)
    w = np.array([[voice.phonemes[p].formant_bw[i] for i in range(8)] for p in range(n)], dtype=np.float64)
    sens = ((0.0709 / w) * (1.0 + (f / 0.075) ** 2)).max(axis=0)
    return float(np.sqrt(((share * sens)[share > 0] ** 2).sum()))


def test_sharpness_of_the_shipped_voices_is_below_the_limit():
    from grail_hip import workload as W
    assert 23.0 < gh.fast_sharpness(gh.voice_generic(48000.0)) < 26.0
    assert 21.0 < gh.fast_sharpness(gh.voice_generic()) < 24.0
    for v in W.preset_voices(8):
        assert gh.fast_sharpness(v) <= 0.8 * gh.FAST_SHARPNESS_LIMIT


def test_sharpness_is_the_formula_of_the_header():
    from grail_hip import workload as W
    for v in [gh.voice_generic(48000.0), gh.voice_generic()] + W.preset_voices(8):
        assert abs(gh.fast_sharpness(v) - sharpness(v)) < 1e-4 * sharpness(v)
    last = gh.fast_sharpness(gh.voice_generic(48000.0))
    for div in (2.0, 4.0, 8.0):
        w = gh.voice_generic(48000.0)
        for p in range(2):
            for i in range(8):
                w.phonemes[p].formant_bw[i] /= div
        s = gh.fast_sharpness(w)
        assert abs(s - sharpness(w)) < 1e-4 * s
        assert 1.9 * last < s < 2.1 * last                          # goes as 1 / bandwidth
        last = s
    assert last > gh.FAST_SHARPNESS_LIMIT                           # bandwidths / 8: exact kernels
    hi = gh.voice_generic(48000.0)
    for p in range(2):
        hi.phonemes[p].formant_freq[0] = 0.3                        # the strongest formant at 14.4 kHz
    assert gh.fast_sharpness(hi) > 10.0 * gh.fast_sharpness(gh.voice_generic(48000.0))


def test_sharpness_ignores_formants_that_are_never_audible_and_rejects_bad_parameters():
    v = gh.voice_generic(48000.0)
    base = gh.fast_sharpness(v)
    for p in range(2):
        v.phonemes[p].formant_bw[7] = 1e-6          # amplitude 0 in both phonemes: nothing rings
    assert gh.fast_sharpness(v) == base
    v.phonemes[0].formant_amp[7] = 0.1              # audible in one phoneme: both phonemes' parameters count
    assert gh.fast_sharpness(v) > 1000.0
    v = gh.voice_generic(48000.0)
    v.phonemes[1].formant_freq[0] = 0.5
    assert math.isinf(gh.fast_sharpness(v))
    v = gh.voice_generic(48000.0)
    v.phonemes[1].formant_bw[2] = 0.0
    assert math.isinf(gh.fast_sharpness(v))
    v = gh.voice_generic(48000.0)
    for p in range(2):
        for i in range(8):
            v.phonemes[p].formant_amp[i] = 0.0
    assert gh.fast_sharpness(v) == 0.0              # a silent voice: whatever arithmetic


def _residual_after(f, w, sm, n, drive_seed=1):
    """float64 model of one formant of Synthesize::next (:531-571): the one-pole low-pass feeding the trapezoidal
    band-pass, run twice on the same input — from its true (long-running) state and from zero state — and the largest
    difference of any state after n samples, relative to the largest state seen (what the warm-up has to undo)."""
    rng = np.random.default_rng(drive_seed)
    g, k = tan_approx(f), w / f
    a1 = 1.0 / (1.0 + g * (g + k))
    a2, a3 = g * a1, g * g * a1
    lp = (1.0 - sm) ** 5

    def step(state, x):
        a, b, c = state
        a = a + (1.0 - lp) * (x - a)
        v3 = a - c
        v1 = a1 * b + a2 * v3
        v2 = c + a2 * b + a3 * v3
        return (a, 2.0 * v1 - b, 2.0 * v2 - c)

    true = (0.0, 0.0, 0.0)
    pre = int(12.0 / min(-math.log(math.sqrt((1 - g * k + g * g) / (1 + g * k + g * g))) if k < 2 else 1.0, -5.0 * math.log1p(-sm))) + 2000
    xs = rng.uniform(-1.0, 1.0, pre + n)
    peak = 0.0
    for x in xs[:pre]:
        true = step(true, x)
        peak = max(peak, abs(true[1]), abs(true[2]), abs(true[0]))
    zero = (0.0, 0.0, 0.0)
    for x in xs[pre:]:
        true, zero = step(true, x), step(zero, x)
        peak = max(peak, abs(true[1]), abs(true[2]))
    return max(abs(true[i] - zero[i]) for i in range(3)) / peak


def _one_formant_voice(f, w, sm, jitter=0.0):
    v = gh.voice_generic(48000.0)
    for p in range(2):
        e = v.phonemes[p]
        for i in range(8):
            e.formant_amp[i] = 1.0 if i == 0 else 0.0
        e.formant_freq[0], e.formant_bw[0], e.formant_smooth[0] = f, w, sm
    v.jitter_delta_formant_frequency = jitter
    return v


def test_warmup_residual_by_simulation_including_the_cascade_case():
    """where the low-pass and the band-pass decay at nearly the same rate the residual of a wrong start
    decays like n r^n, not r^n.  A float64 model of the chain, started from zero, must be within 2^-21 of the true
    state after grail_time_split_warmup samples — far-apart rates (the shipped voices) and equal ones."""
    cases = []
    for f, bw_hz in ((910.0 / 48000.0, 60.0), (0.05, 120.0), (0.2, 300.0), (0.01, 40.0)):
        w = bw_hz / 48000.0
        g, k = tan_approx(f), w / f
        l_bp = -0.5 * math.log((1 - g * k + g * g) / (1 + g * k + g * g))
        sm_equal = 1.0 - math.exp(-l_bp / 5.0)            # the low-pass decaying exactly as fast as the band-pass
        cases += [(f, w, 1600.0 / 48000.0), (f, w, sm_equal), (f, w, sm_equal * 1.1), (f, w, sm_equal * 0.9)]
    for f, w, sm in cases:
        n = gh.time_split_warmup(_one_formant_voice(f, w, sm))
        assert 0 < n <= 16384, (f, w, sm)
        res = _residual_after(f, w, sm, n)
        assert res <= 2.0 ** -21, (f, w, sm, n, res * 2 ** 21)
    # where the two rates meet the length is that of n r^n: longer than the plain r^n length by more than a tile
    f, w = 0.05, 120.0 / 48000.0
    g, k = tan_approx(f), w / f
    l_bp = -0.5 * math.log((1 - g * k + g * g) / (1 + g * k + g * g))
    sm = 1.0 - math.exp(-l_bp / 5.0)
    plain = int(math.ceil(math.log(2.0 ** 21) / (0.95 * l_bp)))
    assert gh.time_split_warmup(_one_formant_voice(f, w, sm)) > plain + 64


def test_warmup_covers_the_range_of_the_formant_frequency_jitter():
    """The band-pass decays most slowly at one end of the range the jitter moves its frequency through (:764)."""
    f, w, sm = 0.02, 30.0 / 48000.0, 0.2
    quiet = gh.time_split_warmup(_one_formant_voice(f, w, sm, 0.0))
    jittered = gh.time_split_warmup(_one_formant_voice(f, w, sm, 0.015))
    assert jittered >= quiet
    for ff in (f - 0.015, f, f + 0.015):
        assert _residual_after(ff, w, sm, jittered) <= 2.0 ** -21
