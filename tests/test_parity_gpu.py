"""GPU parity: the HIP path (through the C ABI) against the CPU oracle, bit for bit.

Integer/bit-exact bar: the kernel is IEEE binary32 without FMA contraction, so every
sample must equal the oracle's as a 32-bit pattern (tolerance 0 ULP)."""
import ctypes as C

import numpy as np
import pytest

import grail_hip as G
import oracle_lib as O
from grail_hip import workload as W

pytestmark = pytest.mark.gpu


def ovoices(voices):
    return [O.Voice.from_buffer_copy(bytes(v)) for v in voices]


def assert_bit_identical(out, out_len, ref, ref_len, what=""):
    assert np.array_equal(out_len, ref_len), f"{what}: lengths differ {out_len[:8]} vs {ref_len[:8]}"
    for u in range(len(out_len)):
        n = int(out_len[u])
        a = out[u, :n].view(np.uint32)
        b = ref[u, :n].view(np.uint32)
        if not np.array_equal(a, b):
            i = int(np.argmax(a != b))
            raise AssertionError(f"{what}: utterance {u} first differs at sample {i}: "
                                 f"{out[u, i]!r} vs {ref[u, i]!r} ({(a != b).sum()} of {n})")


def run_both(ctx, voices, segs, offs, vids, seeds, stride, lanes=0):
    ctx.set_voices(voices)
    ctx.set_option("lanes_per_utterance", lanes)
    try:
        out, out_len = ctx.synthesize(segs, offs, vids, seeds, out_stride=stride)
    finally:
        ctx.set_option("lanes_per_utterance", 0)
    ref, ref_len = O.synthesize_batch(ovoices(voices), segs, offs, vids, seeds, stride)
    return out, out_len, ref, ref_len


@pytest.mark.parametrize("lanes", [1, 2, 4, 8])
def test_short_batch_bit_exact_every_lane_mapping(gpu_ctx, lanes):
    n_utt = 150  # not a multiple of any utterances-per-wave count
    voices = W.single_voice()
    segs, offs, vids, seeds = W.make_batch(n_utt, length=0.03, blend_length=0.03)
    stride = W.max_samples(length=0.03)
    out, out_len, ref, ref_len = run_both(gpu_ctx, voices, segs, offs, vids, seeds, stride, lanes)
    assert_bit_identical(out, out_len, ref, ref_len, f"L={lanes}")
    assert out_len.min() > 5000


@pytest.mark.parametrize("lanes", [0, 8])
def test_config1_text_a_one_second(gpu_ctx, lanes):
    """BASELINE config 1: text "a" => [Silence, A] (src/lib.rs:1201), default Voice, 44.1 kHz."""
    v = G.voice_generic()
    f = v.center_frequency
    segs = G.segments([(G.PH_SILENCE, .5, .5, f), (G.PH_A, .5, .5, f)])
    out, out_len, ref, ref_len = run_both(gpu_ctx, [v], segs, [0, 2], None, None, 44104, lanes)
    assert out_len[0] == 44095
    assert_bit_identical(out, out_len, ref, ref_len, "text a")
    said = O.say(O.voice_generic(), "a")
    assert np.array_equal(said.view(np.uint32), out[0, :44095].view(np.uint32))


def test_eight_voice_presets_divergent_coefficients(gpu_ctx):
    """BASELINE config 4 shape: 8 presets, all eight formants live."""
    voices = W.preset_voices(8)
    n_utt = 64
    segs, offs, vids, seeds = W.make_batch(n_utt, n_voices=8, length=0.05, blend_length=0.05)
    stride = W.max_samples(length=0.05)
    for lanes in (2, 8):
        out, out_len, ref, ref_len = run_both(gpu_ctx, voices, segs, offs, vids, seeds, stride, lanes)
        assert_bit_identical(out, out_len, ref, ref_len, f"presets L={lanes}")
    assert len(set(vids.tolist())) == 8


def edge_case_batch(sr):
    f = np.float32(120.0) / np.float32(sr)
    A, E, S, ST, GL = G.PH_A, G.PH_E, G.PH_SILENCE, G.PH_STOP, G.PH_GLIDE
    utts = [
        [],                                                     # empty: no samples
        [(A, .01, .01, f)],                                     # single segment: fades out
        [(S, .01, .01, f)],                                     # a lone silence
        [(S, .01, .01, f), (ST, .01, .01, f), (GL, .01, .01, f)],  # all None elems
        [(A, .02, .005, f), (E, .02, .04, f), (A, .01, .01, f)],   # blend_length != length
        [(A, .01, 0.0, f), (E, .01, .01, f)],                   # blend_length 0: time/0 = inf -> 1
        [(A, 0.0, .01, f), (E, .01, .01, f)],                   # zero-length segment
        [(A, -1.0, .01, f), (E, .01, .01, f)],                  # negative length: ends early
        [(A, 1e-6, .01, f), (E, 1e-6, .01, f), (A, .01, .01, f)],  # shorter than one sample
        [(A, .01, .01, 0.7), (E, .01, .01, 0.5)],               # pitch above Nyquist: clamp .min(0.5)
        [(A, .01, .01, 0.0), (E, .01, .01, 1e-9)],              # zero pitch: phase/0
        [(E, .01, .01, f), (S, .01, .01, f), (A, .01, .01, f), (GL, .01, .01, f), (E, .01, .01, f)],
        [(A, .013, -.01, f), (E, .01, .01, f)],                 # negative blend length
        [(A, .01, .01, float("nan")), (E, .01, .01, f)],        # NaN pitch propagates
        [(A, .05, .05, f)] * 7,                                 # many segments
    ]
    segs, offs = [], [0]
    for u in utts:
        segs += u
        offs.append(len(segs))
    return G.segments(segs), np.array(offs, dtype=np.uint32)


@pytest.mark.parametrize("lanes", [1, 2, 4, 8])
def test_edge_cases_ragged_empty_nan(gpu_ctx, lanes):
    voices = W.single_voice()
    segs, offs = edge_case_batch(48000.0)
    n_utt = len(offs) - 1
    seeds = np.arange(n_utt, dtype=np.uint32) * 977
    with np.errstate(all="ignore"):
        out, out_len, ref, ref_len = run_both(gpu_ctx, voices, segs, offs, None, seeds, 20000, lanes)
    assert out_len[0] == 0
    assert_bit_identical(out, out_len, ref, ref_len, f"edge L={lanes}")


def test_lengths_prepass_matches_oracle(gpu_ctx):
    voices = W.single_voice()
    gpu_ctx.set_voices(voices)
    segs, offs = edge_case_batch(48000.0)
    b = gpu_ctx.upload(segs, offs)
    lens = b.lengths()
    capped = b.lengths(max_len=100)
    b.free()
    want = O.count_batch(ovoices(voices), segs, offs, np.zeros(len(offs) - 1), np.zeros(len(offs) - 1))
    assert np.array_equal(lens, want)
    assert np.array_equal(capped, np.minimum(want, 100))


def test_truncation_is_reported_not_silent(gpu_ctx):
    voices = W.single_voice()
    gpu_ctx.set_voices(voices)
    segs, offs, vids, seeds = W.make_batch(10, length=0.02, blend_length=0.02)
    full, full_len = gpu_ctx.synthesize(segs, offs, vids, seeds, out_stride=W.max_samples(length=0.02))
    with pytest.raises(G.GrailError) as ei:
        gpu_ctx.synthesize(segs, offs, vids, seeds, out_stride=1000)
    assert ei.value.status == G.ERR_BUFFER_TOO_SMALL
    out, out_len = gpu_ctx.synthesize(segs, offs, vids, seeds, out_stride=1000, allow_truncation=True)
    assert np.all(out_len == 1000)
    assert np.array_equal(out.view(np.uint32), full[:, :1000].view(np.uint32))
    # unaligned stride takes the scalar store path
    out, out_len = gpu_ctx.synthesize(segs, offs, vids, seeds, out_stride=4001)
    assert np.array_equal(out_len, full_len)
    for u in range(10):
        assert np.array_equal(out[u, :out_len[u]].view(np.uint32), full[u, :out_len[u]].view(np.uint32))


def test_batch_invariance(gpu_ctx):
    """SURVEY §8b determinism contract: utterance u's samples do not depend on batch size,
    position in the batch or the lane mapping."""
    voices = W.preset_voices(8)
    gpu_ctx.set_voices(voices)
    stride = W.max_samples(length=0.02)
    segs, offs, vids, seeds = W.make_batch(200, n_voices=8, length=0.02, blend_length=0.02)
    gpu_ctx.set_option("lanes_per_utterance", 2)
    full, full_len = gpu_ctx.synthesize(segs, offs, vids, seeds, out_stride=stride)
    for first, n, lanes in [(37, 1, 8), (100, 33, 4), (150, 50, 1), (0, 64, 0)]:
        s2, o2, v2, j2 = W.make_batch(n, first_utt=first, n_voices=8, length=0.02, blend_length=0.02)
        gpu_ctx.set_option("lanes_per_utterance", lanes)
        part, part_len = gpu_ctx.synthesize(s2, o2, v2, j2, out_stride=stride)
        assert np.array_equal(part_len, full_len[first:first + n])
        assert np.array_equal(part.view(np.uint32), full[first:first + n].view(np.uint32))
    gpu_ctx.set_option("lanes_per_utterance", 0)


def test_explicit_sequence_elems_variant(gpu_ctx):
    """.sequence(v).jitter(seed, v).synthesize() over caller-built SequenceElems."""
    v = G.voice_generic(48000.0)
    rng = np.random.default_rng(5)
    gsegs, osegs, offs = [], [], [0]
    for u in range(40):
        for _ in range(int(rng.integers(0, 5))):
            has = bool(rng.integers(0, 4))
            e = np.zeros(49, dtype=np.float32)
            e[0] = rng.uniform(0.001, 0.01)
            e[1:9] = rng.uniform(0.01, 0.2, 8)
            e[9:17] = rng.uniform(0.001, 0.01, 8)
            e[17:25] = rng.uniform(0.01, 0.1, 8)
            e[25:33] = rng.uniform(0, 1, 8)
            e[33:41] = rng.uniform(0, 1, 8)
            amp = rng.uniform(0, 1, 8)
            e[41:49] = amp / amp.sum()
            ln, bl = float(rng.uniform(0.002, 0.02)), float(rng.uniform(0.002, 0.02))
            gs = G.SequenceElem(int(has), G.SynthesisElem.from_np(e), ln, bl)
            os_ = O.SequenceElem(int(has), O.SynthesisElem.from_buffer_copy(e.tobytes()), ln, bl)
            gsegs.append(gs)
            osegs.append(os_)
        offs.append(len(gsegs))
    seeds = np.arange(40, dtype=np.uint32) + 11
    gpu_ctx.set_voices([v])
    for lanes in (1, 8):
        gpu_ctx.set_option("lanes_per_utterance", lanes)
        out, out_len = gpu_ctx.synthesize_elems(gsegs, offs, None, seeds, out_stride=4096)
        ov = O.Voice.from_buffer_copy(bytes(v))
        for u in range(40):
            ref = O.synthesize_sequence(ov, osegs[offs[u]:offs[u + 1]], int(seeds[u]))
            assert out_len[u] == len(ref)
            assert np.array_equal(out[u, :len(ref)].view(np.uint32), ref.view(np.uint32)), u
    gpu_ctx.set_option("lanes_per_utterance", 0)


def test_config2_full_size_properties_and_sampled_parity(gpu_ctx):
    """BASELINE config 2 (4096 utterances x 2 s, single Voice, 48 kHz) at full size:
    size-independent properties on everything, bit parity on a sample of utterances."""
    n_utt = 4096
    voices = W.single_voice()
    gpu_ctx.set_voices(voices)
    gpu_ctx.set_option("lanes_per_utterance", 0)
    segs, offs, vids, seeds = W.make_batch(n_utt)
    stride = W.max_samples()
    b = gpu_ctx.upload(segs, offs, vids, seeds)
    d_out = gpu_ctx.device_alloc(n_utt * stride * 4)
    d_len = gpu_ctx.device_alloc(n_utt * 4)
    try:
        gpu_ctx.memset(d_out, 0xFF, n_utt * stride * 4)  # NaN canary
        b.synthesize_async(d_out, stride, d_len)
        gpu_ctx.sync()
        assert gpu_ctx.last_kernel_ms() > 0
        out_len = np.zeros(n_utt, dtype=np.uint32)
        gpu_ctx.d2h(out_len, d_len, n_utt * 4)
        out = np.zeros((n_utt, stride), dtype=np.float32)
        gpu_ctx.d2h(out, d_out, n_utt * stride * 4)
        lens = b.lengths()
    finally:
        gpu_ctx.device_free(d_out)
        gpu_ctx.device_free(d_len)
        b.free()
    # the clock pre-pass and the synthesis agree, and match the f32 clock KAT
    assert np.array_equal(out_len, lens)
    assert np.all(out_len == 96006)
    valid = out[:, :96006]
    assert np.isfinite(valid).all()
    # synthesize_normalized (src/lib.rs:602): peaks stay inside [-1, 1]
    assert np.abs(valid).max() <= 1.0
    # samples past the end are untouched (canary still NaN)
    assert np.isnan(out[:, 96006:]).all()
    # segment 0 is Silence: identical fade-in start for everyone who shares pitch/phoneme/seed? no —
    # seeds differ; but the very first sample is the same function of (phoneme 1, pitch 1)
    # bit parity on a sample of utterances across the batch (first, last, wave boundaries)
    pick = [0, 1, 7, 8, 31, 32, 63, 64, 2047, 2048, 4064, 4095]
    sub = np.concatenate([segs[offs[u]:offs[u + 1]] for u in pick])
    sub_offs = np.arange(len(pick) + 1, dtype=np.uint32) * 4
    ref, ref_len = O.synthesize_batch(ovoices(voices), sub, sub_offs, vids[pick], seeds[pick], stride)
    assert_bit_identical(out[pick], out_len[pick], ref, ref_len, "config 2 sample")


def test_rccl_voice_broadcast_single_rank(gpu_ctx):
    """grail_comm_* / grail_broadcast_voices on a 1-rank communicator: the RCCL library loads,
    the communicator forms, ncclBroadcast runs on the context's stream and the table survives."""
    voices = W.preset_voices(8)
    gpu_ctx.set_voices(voices)
    uid = G.Context.comm_unique_id()
    assert len(uid) == G.UNIQUE_ID_BYTES
    gpu_ctx.comm_init(uid, 0, 1)
    gpu_ctx.broadcast_voices(8, root=0)
    got = gpu_ctx.get_voices()
    assert all(bytes(a) == bytes(b) for a, b in zip(got, voices))
    with pytest.raises(G.GrailError):
        gpu_ctx.broadcast_voices(3, root=0)   # root's table holds 8 voices, not 3
    G._check(G.load().grail_comm_destroy(gpu_ctx.handle))


@pytest.mark.parametrize("n_voices", [1, 8])
def test_config3_and_4_full_size_on_device_digest(gpu_ctx, n_voices):
    """BASELINE configs 3 and 4 at FULL size (65536 utterances x 2 s, 25 GB of PCM left in HBM):
    lengths, finiteness and the normalisation bound over all 6.3e9 samples via the on-device digest;
    a checksum of checksums between two lane mappings (batch invariance at scale); bit parity of
    138 utterances spread over the batch against the oracle through their digests."""
    n_utt = 65536
    voices = W.single_voice() if n_voices == 1 else W.preset_voices(8)
    gpu_ctx.set_voices(voices)
    segs, offs, vids, seeds = W.make_batch(n_utt, n_voices=n_voices)
    stride = W.max_samples()
    b = gpu_ctx.upload(segs, offs, vids, seeds)
    d_out = gpu_ctx.device_alloc(n_utt * stride * 4)
    d_len = gpu_ctx.device_alloc(n_utt * 4)
    digests = {}
    try:
        for lanes in (0, 2):
            gpu_ctx.set_option("lanes_per_utterance", lanes)
            b.synthesize_async(d_out, stride, d_len)
            gpu_ctx.sync()
            out_len = np.zeros(n_utt, dtype=np.uint32)
            gpu_ctx.d2h(out_len, d_len, n_utt * 4)
            assert np.all(out_len == 96006)
            sums, maxabs, bad = gpu_ctx.digest(d_out, stride, d_len, n_utt)
            assert bad.sum() == 0                       # every sample finite
            assert maxabs.max() <= 1.0                  # synthesize_normalized (src/lib.rs:602)
            assert (maxabs > 0.01).mean() > 0.9         # (an all-Silence utterance is exactly 0)
            digests[lanes] = sums
    finally:
        gpu_ctx.set_option("lanes_per_utterance", 0)
        gpu_ctx.device_free(d_out)
        gpu_ctx.device_free(d_len)
        b.free()
    assert np.array_equal(digests[0], digests[2])       # 65536 checksums agree across lane mappings
    pick = sorted(set([0, 1, 63, 64, 1023, 1024, 32767, 32768, 65534, 65535] +
                      [int(u) for u in np.random.default_rng(7).integers(0, n_utt, 128)]))
    sub = np.concatenate([segs[offs[u]:offs[u + 1]] for u in pick])
    sub_offs = np.arange(len(pick) + 1, dtype=np.uint32) * 4
    ref, ref_len = O.synthesize_batch(ovoices(voices), sub, sub_offs, vids[pick], seeds[pick], stride)
    for k, u in enumerate(pick):
        want = int(ref[k, :ref_len[k]].view(np.uint32).astype(np.uint64).sum())
        assert int(digests[0][u]) == want, u


def _elem(rng, amp_mask):
    e = np.zeros(49, dtype=np.float32)
    e[0] = rng.uniform(0.002, 0.006)
    e[1:9] = rng.uniform(0.01, 0.2, 8)
    e[9:17] = rng.uniform(0.001, 0.01, 8)
    e[17:25] = rng.uniform(0.01, 0.1, 8)
    e[25:33] = rng.uniform(0, 1, 8)
    e[33:41] = rng.uniform(0, 1, 8)
    amp = rng.uniform(0.1, 1, 8) * np.asarray(amp_mask, dtype=np.float64)
    e[41:49] = amp / amp.sum()
    return e


@pytest.mark.parametrize("lanes", [1, 2])
def test_silent_formant_skip_is_bit_exact_when_formants_wake_up(gpu_ctx, lanes):
    """"skip_silent_formants" may only skip what is provably +0: utterances whose upper formants
    are silent, become live, fall silent again (ringing state != 0), against the oracle, with the
    option on and off."""
    rng = np.random.default_rng(11)
    lo, all8 = [1, 1, 1, 1, 0, 0, 0, 0], [1] * 8
    plans = [[lo, lo, lo], [lo, all8, lo], [all8, lo, lo], [lo, lo, all8, lo, lo], [all8, all8],
             [lo, None, lo, all8], [None, lo, lo]]
    gsegs, osegs, offs = [], [], [0]
    for plan in plans * 3:
        for mask in plan:
            has = mask is not None
            e = _elem(rng, mask if has else all8)
            ln = float(rng.uniform(0.004, 0.012))
            gsegs.append(G.SequenceElem(int(has), G.SynthesisElem.from_np(e), ln, 0.0078125))
            osegs.append(O.SequenceElem(int(has), O.SynthesisElem.from_buffer_copy(e.tobytes()), ln, 0.0078125))
        offs.append(len(gsegs))
    n = len(offs) - 1
    seeds = np.arange(n, dtype=np.uint32) * 31 + 5
    v = G.voice_generic(48000.0)
    gpu_ctx.set_voices([v])
    gpu_ctx.set_option("lanes_per_utterance", lanes)
    ov = O.Voice.from_buffer_copy(bytes(v))
    try:
        outs = {}
        for skip in (1, 0):
            gpu_ctx.set_option("skip_silent_formants", skip)
            outs[skip] = gpu_ctx.synthesize_elems(gsegs, offs, None, seeds, out_stride=4096)
        assert np.array_equal(outs[0][1], outs[1][1])
        assert np.array_equal(outs[0][0].view(np.uint32), outs[1][0].view(np.uint32))
        for u in range(n):
            ref = O.synthesize_sequence(ov, osegs[offs[u]:offs[u + 1]], int(seeds[u]))
            assert outs[1][1][u] == len(ref)
            assert np.array_equal(outs[1][0][u, :len(ref)].view(np.uint32), ref.view(np.uint32)), u
    finally:
        gpu_ctx.set_option("skip_silent_formants", 1)
        gpu_ctx.set_option("lanes_per_utterance", 0)


def test_silent_formant_skip_respects_the_amplitude_jitter_bound(gpu_ctx):
    """0.5*jitter_delta_amplitude > 1/4 could make 0*(1-delta) a -0: such voices never skip."""
    v = G.voice_generic(48000.0)
    v.jitter_delta_amplitude = 1.7
    segs, offs, vids, seeds = W.make_batch(70, length=0.02, blend_length=0.02)
    stride = W.max_samples(length=0.02)
    gpu_ctx.set_option("lanes_per_utterance", 1)
    try:
        out, out_len, ref, ref_len = run_both(gpu_ctx, [v], segs, offs, vids, seeds, stride, 1)
        assert_bit_identical(out, out_len, ref, ref_len, "wide amplitude jitter")
    finally:
        gpu_ctx.set_option("lanes_per_utterance", 0)


@pytest.mark.parametrize("lanes", [1, 2, 8])
def test_non_uniform_smoothness_takes_the_vector_path(gpu_ctx, lanes):
    """The quiet loop evaluates 1-exp_approx(smooth) once when every formant shares one
    smoothness (voices::generic()); voices with per-formant smoothness must take the vector form."""
    v = G.voice_generic(48000.0)
    for p in range(2):
        for i in range(8):
            v.phonemes[p].formant_smooth[i] = float(np.float32(v.phonemes[p].formant_smooth[i]) *
                                                    np.float32(1.0 + 0.07 * i + 0.01 * p))
    segs, offs, vids, seeds = W.make_batch(70, length=0.02, blend_length=0.02)
    stride = W.max_samples(length=0.02)
    out, out_len, ref, ref_len = run_both(gpu_ctx, [v], segs, offs, vids, seeds, stride, lanes)
    gpu_ctx.set_option("lanes_per_utterance", 0)
    assert_bit_identical(out, out_len, ref, ref_len, f"per-formant smoothness L={lanes}")


@pytest.mark.parametrize("lanes", [1, 2, 4, 8])
def test_calm_tile_boundaries(gpu_ctx, lanes):
    """The kernel runs whole tiles (32 or 64 steps) without per-step event checks when no lane can
    have an event inside the tile.  Segment ends placed on every offset around the tile edges,
    jitter wraps every 37 / 64 / 65 steps, rows that fill up inside a tile, utterances of a wave
    that end at different samples: all must still equal the oracle bit for bit."""
    voices = []
    for every in (37.0, 64.0, 65.0, 3000.0):
        v = G.voice_generic(48000.0)
        v.jitter_frequency = float(np.float32(1.0) / np.float32(every))
        voices.append(v)
    rate = np.float32(48000.0)
    segs, offs, vids, seeds = [], [0], [], []
    for k in range(20, 150):                       # segment ends sweep across two tile sizes
        length = float((np.float32(k) + np.float32(0.5)) / rate)
        n_seg = 2 + k % 3
        for i in range(n_seg):
            ph = (G.PH_A, G.PH_E, G.PH_SILENCE)[(k + i) % 3]
            # blend lengths: two powers of two and one that is not (the kernel with the short
            # exact clk / blend_length division, its clk floor included: length < blend length)
            blend = (2.0 ** -8, 2.0 ** -10, 0.0031)[(k + i) % 3] if k % 5 else (2.0 ** -8, 2.0 ** -10)[i % 2]
            segs.append((ph, length, blend, float(np.float32(90 + k) / rate)))
        offs.append(len(segs))
        vids.append(k % len(voices))
        seeds.append(k * 7919)
    segs = G.segments(segs)
    offs = np.array(offs, dtype=np.uint32)
    vids = np.array(vids, dtype=np.uint32)
    seeds = np.array(seeds, dtype=np.uint32)
    try:
        for stride in (1024, 333):                 # 333: rows fill up inside a tile
            gpu_ctx.set_voices(voices)
            gpu_ctx.set_option("lanes_per_utterance", lanes)
            out, out_len = gpu_ctx.synthesize(segs, offs, vids, seeds, out_stride=stride,
                                              allow_truncation=True)
            ref, ref_len = O.synthesize_batch(ovoices(voices), segs, offs, vids, seeds, stride)
            ref_len = np.minimum(ref_len, stride)
            assert ref_len.max() == stride or stride == 1024
            assert_bit_identical(out, out_len, ref, ref_len, f"tile edges L={lanes} stride={stride}")
    finally:
        gpu_ctx.set_option("lanes_per_utterance", 0)


def _upper_silent_voice(mutate=None):
    v = G.voice_generic(48000.0)
    if mutate:
        mutate(v)
    return v


@pytest.mark.parametrize("lanes", [1, 2, 4])
def test_four_formant_kernels_and_their_gate(gpu_ctx, lanes):
    """voices::generic() qualifies for the kernels that lay out formants 1-4 only.  The gate must
    refuse tables or batches for which formants 5-8 could matter (a dead formant with frequency 0
    makes the reference itself emit NaN; a segment shorter than two samples lets the clock go
    negative) and the result must equal the oracle either way, NaN patterns included."""
    def freq0(v):
        v.phonemes[1].formant_freq[6] = 0.0

    def breathy(v):
        v.phonemes[0].formant_breath[5] = 1.5

    def wild_amp_jitter(v):
        v.jitter_delta_amplitude = 0.9

    n_utt = 70
    cases = [("generic", None, 0.02, 4), ("dead formant at frequency 0", freq0, 0.02, 8),
             ("dead formant breath 1.5", breathy, 0.02, 8), ("amplitude jitter 0.9", wild_amp_jitter, 0.02, 8),
             ("a one-sample segment", None, None, 8), ("a pitch of 1e-30", None, -1.0, 8)]
    # (the last two are about ONE row of the batch: with row groups that row is planned apart and the others keep their
    # four formants — here the gate itself is under test, so the whole batch is planned as one)
    gpu_ctx.set_option("row_groups", 0)
    try:
        for what, mutate, length, want_formants in cases:
            voices = [_upper_silent_voice(mutate)]
            segs, offs, vids, seeds = W.make_batch(n_utt, length=0.02, blend_length=2.0 ** -6)
            if length is None:
                segs["length"][5] = np.float32(1.0 / 48000.0)
            elif length < 0:      # the polyBLEP quotient overflows: +-inf reaches every formant
                segs["frequency"][9] = np.float32(1e-30)
            stride = W.max_samples(length=0.02)
            with np.errstate(all="ignore"):
                out, out_len, ref, ref_len = run_both(gpu_ctx, voices, segs, offs, vids, seeds, stride, lanes)
            assert gpu_ctx.get_option("last_launch_formants") == want_formants, what
            assert_bit_identical(out, out_len, ref, ref_len, f"{what} L={lanes}")
    finally:
        gpu_ctx.set_option("row_groups", 1)
        gpu_ctx.set_option("lanes_per_utterance", 0)


@pytest.mark.parametrize("n_utt", [1, 17, 300])
def test_small_batch_pipeline_is_bit_exact(gpu_ctx, n_utt):
    """Small qualifying batches run four-wave workgroups: one wave renders 16 utterances, one carries
    the per-utterance chain, two prepare the filter coefficients, LDS rings and barriers in between.
    Whole 2-second utterances (hundreds of calm tiles, segment boundaries, jitter wraps), a partly
    filled last workgroup, and the same batch with the pipeline switched off: all equal the oracle."""
    voices = W.single_voice()
    segs, offs, vids, seeds = W.make_batch(n_utt)              # config-3 utterances: 4 x 0.5 s
    stride = W.max_samples()
    ref, ref_len = O.synthesize_batch(ovoices(voices), segs, offs, vids, seeds, stride)
    try:
        for pipeline in (1, 0):
            gpu_ctx.set_option("small_batch_pipeline", pipeline)
            gpu_ctx.set_voices(voices)
            out, out_len = gpu_ctx.synthesize(segs, offs, vids, seeds, out_stride=stride)
            assert gpu_ctx.get_option("last_launch_pipelined") == pipeline
            assert_bit_identical(out, out_len, ref, ref_len, f"pipeline={pipeline} n={n_utt}")
    finally:
        gpu_ctx.set_option("small_batch_pipeline", 1)


def test_small_batch_pipeline_with_two_workgroups_per_cu_is_bit_exact(gpu_ctx):
    """4 097 - 8 192 utterances with four live formants keep the pipelined workgroups, two per CU
    ("pipeline4_max_groups" = 512): 4 200 short utterances against the oracle."""
    voices = W.single_voice()
    n_utt = 4200
    segs, offs, vids, seeds = W.make_batch(n_utt, length=0.06, blend_length=0.0625)
    stride = W.max_samples(length=0.06)
    ref, ref_len = O.synthesize_batch(ovoices(voices), segs, offs, vids, seeds, stride)
    gpu_ctx.set_voices(voices)
    out, out_len = gpu_ctx.synthesize(segs, offs, vids, seeds, out_stride=stride)
    assert gpu_ctx.get_option("last_launch_pipelined") == 1
    assert_bit_identical(out, out_len, ref, ref_len, "pipeline, 263 workgroups")


@pytest.mark.parametrize("n_utt", [13, 40])
def test_small_batch_pipeline_with_eight_live_formants_is_bit_exact(gpu_ctx, n_utt):
    """Tables whose eight formants are all audible (config 4's presets) take the same four-wave
    pipeline with eight lanes per utterance (8 utterances per workgroup): bit-identical to the oracle,
    pipeline on and off, partly filled last workgroup included."""
    voices = W.preset_voices(8)
    segs, offs, vids, seeds = W.make_batch(n_utt, n_voices=8)
    stride = W.max_samples()
    ref, ref_len = O.synthesize_batch(ovoices(voices), segs, offs, vids, seeds, stride)
    try:
        for pipeline in (1, 0):
            gpu_ctx.set_option("small_batch_pipeline", pipeline)
            gpu_ctx.set_voices(voices)
            out, out_len = gpu_ctx.synthesize(segs, offs, vids, seeds, out_stride=stride)
            assert gpu_ctx.get_option("last_launch_pipelined") == pipeline
            assert gpu_ctx.get_option("last_launch_formants") == 8
            assert_bit_identical(out, out_len, ref, ref_len, f"8 formants pipeline={pipeline} n={n_utt}")
    finally:
        gpu_ctx.set_option("small_batch_pipeline", 1)
        gpu_ctx.set_voices(W.single_voice())


@pytest.mark.parametrize("lanes", [0, 1, 2, 4, 8])
def test_length_sorted_slot_assignment_is_invisible(gpu_ctx, lanes):
    """Ragged batches fill the launch slots in order of decreasing length (grail_api.cpp
    upload_length_order); every row still belongs to the caller's utterance and every sample is the
    oracle's, with the option on and off, one-shot and streamed."""
    rng = np.random.default_rng(7)
    voices = W.single_voice()
    n_utt = 150
    segs, offs, vids, seeds = W.make_batch(n_utt, length=0.02, blend_length=0.015625)
    segs["length"] = rng.uniform(0.004, 0.03, len(segs)).astype(np.float32)
    stride = 8192
    ref, ref_len = O.synthesize_batch(ovoices(voices), segs, offs, vids, seeds, stride)
    assert len(set(ref_len.tolist())) > 50
    gpu_ctx.set_voices(voices)
    try:
        for sort in (1, 0):
            gpu_ctx.set_option("sort_by_length", sort)
            gpu_ctx.set_option("lanes_per_utterance", lanes)
            out, out_len = gpu_ctx.synthesize(segs, offs, vids, seeds, out_stride=stride)
            assert_bit_identical(out, out_len, ref, ref_len, f"sort={sort} lanes={lanes}")
    finally:
        gpu_ctx.set_option("sort_by_length", 1)
        gpu_ctx.set_option("lanes_per_utterance", 0)


def test_host_destinations_pinned_and_pageable_get_the_same_rows(gpu_ctx):
    """GRAIL_OUT_HOST renders row blocks while the previous block leaves over PCIe: a pinned destination
    (direct copies) and a pageable one (staging ring + copier threads) receive the oracle's rows, zero tails
    included, also when the batch spans several blocks and when rows are cut at out_stride."""
    voices = W.single_voice()
    gpu_ctx.set_voices(voices)
    n_utt = 4096 + 700                                     # two row blocks (4096 + a partly filled one)
    segs, offs, vids, seeds = W.make_batch(n_utt, length=0.004, blend_length=0.00390625)
    stride = W.max_samples(length=0.004)
    ref, ref_len = O.synthesize_batch(ovoices(voices), segs, offs, vids, seeds, stride)
    out_len = np.zeros(n_utt, dtype=np.uint32)
    pinned = gpu_ctx.host_alloc((n_utt, stride), np.float32)
    try:
        for dst in (pinned, np.full((n_utt, stride), 7.0, dtype=np.float32)):
            dst[:] = 7.0
            gpu_ctx.synthesize_into(dst, out_len, segs, offs, vids, seeds)
            assert np.array_equal(out_len, ref_len)
            assert np.array_equal(dst.view(np.uint32), ref.view(np.uint32))      # samples, then zeros
    finally:
        gpu_ctx.host_free(pinned)
    # rows cut at out_stride: reported, lengths capped, the samples that fit are the oracle's
    short = 256
    dst = np.zeros((n_utt, short), dtype=np.float32)
    with pytest.raises(G.GrailError) as ei:
        gpu_ctx.synthesize_into(dst, out_len, segs, offs, vids, seeds)
    assert ei.value.status == G.ERR_BUFFER_TOO_SMALL
    assert np.array_equal(out_len, np.minimum(ref_len, short))
    assert np.array_equal(dst.view(np.uint32), ref[:, :short].view(np.uint32))


HOSTILE_SCALARS = [
    ("jitter_frequency", 0.0), ("jitter_frequency", 1.0), ("jitter_frequency", 1.5), ("jitter_frequency", -0.25),
    ("jitter_frequency", float("nan")), ("jitter_frequency", float("inf")), ("jitter_frequency", 1e-30),
    ("jitter_delta_frequency", 0.0), ("jitter_delta_frequency", -0.01), ("jitter_delta_frequency", 0.6),
    ("jitter_delta_frequency", float("nan")), ("jitter_delta_frequency", float("inf")),
    ("jitter_delta_formant_frequency", 0.0), ("jitter_delta_formant_frequency", 0.3),
    ("jitter_delta_formant_frequency", float("nan")), ("jitter_delta_formant_frequency", -float("inf")),
    ("jitter_delta_amplitude", 0.0), ("jitter_delta_amplitude", -3.0), ("jitter_delta_amplitude", 1e30),
    ("jitter_delta_amplitude", float("nan")),
    ("sample_rate", 1.0), ("sample_rate", 1.0e6), ("sample_rate", 1.0e-3), ("sample_rate", float("inf")),
    ("sample_rate", -48000.0), ("sample_rate", float("nan")),
    ("center_frequency", float("nan")),
]


@pytest.mark.parametrize("lanes", [0, 1, 8])
def test_hostile_voice_scalars_propagate_as_in_the_reference(gpu_ctx, lanes):
    """The reference sanitises nothing on this path ("Errors"): a jitter rate of 0, 1, above 1, negative,
    NaN or Inf, jitter depths of any size or sign, a sample rate of 1, 1e6, 1e-3, Inf, negative or NaN (dt = 1 / rate
    decides whether the clock ever runs down: rows end by exhaustion or fill up) all give SOME sequence of bits and
    lengths, and the kernels must give the same — every case is a voice of one table, rendered in one batch."""
    voices = []
    for field, value in HOSTILE_SCALARS:
        v = G.voice_generic(48000.0)
        setattr(v, field, value)
        voices.append(v)
    n_utt = 3 * len(voices)
    segs, offs, vids, seeds = W.make_batch(n_utt, n_voices=len(voices), length=0.004, blend_length=2.0 ** -8)
    vids = (np.arange(n_utt) % len(voices)).astype(np.uint32)
    stride = 1088                       # rows of voices whose clock never runs down fill up: truncation is the expected end
    try:
        gpu_ctx.set_voices(voices)
        gpu_ctx.set_option("lanes_per_utterance", lanes)
        with np.errstate(all="ignore"):
            out, out_len = gpu_ctx.synthesize(segs, offs, vids, seeds, out_stride=stride, allow_truncation=True)
            ref, ref_len = O.synthesize_batch(ovoices(voices), segs, offs, vids, seeds, stride)
    finally:
        gpu_ctx.set_option("lanes_per_utterance", 0)
    ref_len = np.minimum(ref_len, stride)
    assert (ref_len == stride).any() and (ref_len < stride).any()
    for u in range(n_utt):
        field, value = HOSTILE_SCALARS[vids[u]]
        assert out_len[u] == ref_len[u], (field, value, u, int(out_len[u]), int(ref_len[u]))
        n = int(ref_len[u])
        a, b = out[u, :n].view(np.uint32).copy(), ref[u, :n].view(np.uint32).copy()
        # a NaN that comes IN through a parameter keeps the sign and payload rules of the machine it travels on
        # (x86 hands on the first operand's, the GPU its canonical one): NaN where the oracle has NaN is all IEEE-754
        # promises; everything else bit for bit
        both_nan = np.isnan(out[u, :n]) & np.isnan(ref[u, :n])
        a[both_nan] = 0
        b[both_nan] = 0
        assert np.array_equal(a, b), (field, value, u, int(np.argmax(a != b)))
    # fast arithmetic on the same table: same lengths, NaN / Inf where the reference has them, the tolerance elsewhere
    # (relative to the utterance's peak: some of these voices reach 1e30)
    try:
        gpu_ctx.set_option("lanes_per_utterance", lanes)
        gpu_ctx.set_option("arithmetic", 1)
        with np.errstate(all="ignore"):
            fast, fast_len = gpu_ctx.synthesize(segs, offs, vids, seeds, out_stride=stride, allow_truncation=True)
    finally:
        gpu_ctx.set_option("arithmetic", 0)
        gpu_ctx.set_option("lanes_per_utterance", 0)
    for u in range(n_utt):
        field, value = HOSTILE_SCALARS[vids[u]]
        assert fast_len[u] == ref_len[u], ("fast", field, value, u)
        n = int(ref_len[u])
        r, f = ref[u, :n].astype(np.float64), fast[u, :n].astype(np.float64)
        fin = np.isfinite(r)
        assert np.array_equal(np.isnan(r), np.isnan(f)), ("fast", field, value, u)
        assert np.array_equal(r[~fin & ~np.isnan(r)], f[~fin & ~np.isnan(r)]), ("fast", field, value, u)
        if fin.any():
            peak = max(1.0, float(np.abs(r[fin]).max()))
            assert float(np.abs(r[fin] - f[fin]).max()) <= G.FAST_TOLERANCE * peak, ("fast", field, value, u)


HOSTILE_VALUES = (0.0, -0.0, float("nan"), float("inf"), -1.0, 0.5, 0.75, 1e-39, 3e38)
ELEM_ARRAYS = ("formant_freq", "formant_bw", "formant_smooth", "formant_breath", "formant_turb", "formant_amp")


def _same_but_for_nan_payloads(out_row, ref_row):
    a, b = out_row.view(np.uint32).copy(), ref_row.view(np.uint32).copy()
    both_nan = np.isnan(out_row) & np.isnan(ref_row)
    a[both_nan] = 0
    b[both_nan] = 0
    return np.array_equal(a, b), int(np.argmax(a != b))


@pytest.mark.parametrize("lanes", [0, 1, 2, 8])
def test_hostile_formant_parameters_propagate_as_in_the_reference(gpu_ctx, lanes):
    """One entry of one formant array of phoneme A set to 0, -0, NaN, Inf, -1, 1/2 (tan_approx's pole), 3/4 (beyond it),
    a denormal, 3e38 — in an audible formant (2) and in one voices::generic() leaves silent (7): 108 voices of one table,
    three utterances each.  Whatever the reference's arithmetic makes of it (k = bw / 0, a band-pass that blows up, NaN
    from the first sample on) the kernels make the same of it: safe-window checks, the four-formant gate and the silent-
    formant skip must all refuse what they cannot reproduce."""
    voices, what = [], []
    for arr in ELEM_ARRAYS:
        for value in HOSTILE_VALUES:
            for formant in (1, 6):
                v = G.voice_generic(48000.0)
                getattr(v.phonemes[0], arr)[formant] = value
                voices.append(v)
                what.append((arr, value, formant))
    n_utt = 3 * len(voices)
    segs, offs, vids, seeds = W.make_batch(n_utt, n_voices=len(voices), length=0.005, blend_length=2.0 ** -8)
    vids = (np.arange(n_utt) % len(voices)).astype(np.uint32)
    stride = W.max_samples(length=0.005)
    try:
        gpu_ctx.set_voices(voices)
        gpu_ctx.set_option("lanes_per_utterance", lanes)
        with np.errstate(all="ignore"):
            out, out_len = gpu_ctx.synthesize(segs, offs, vids, seeds, out_stride=stride)
            ref, ref_len = O.synthesize_batch(ovoices(voices), segs, offs, vids, seeds, stride)
    finally:
        gpu_ctx.set_option("lanes_per_utterance", 0)
    assert np.array_equal(out_len, ref_len)
    nonfinite = 0
    for u in range(n_utt):
        n = int(ref_len[u])
        same, at = _same_but_for_nan_payloads(out[u, :n], ref[u, :n])
        assert same, (what[vids[u]], u, at, out[u, at], ref[u, at])
        nonfinite += int(not np.isfinite(ref[u, :n]).all())
    assert nonfinite > 20          # (the table does produce NaN / Inf rows: the comparison is not vacuous)


def _as_sequence_elems(voices, segs, offs, vids):
    """The PhonemeElems of a batch as the SequenceElems the Selector would hand on (src/lib.rs:990-1005)."""
    import ctypes as C
    arr = (G.SequenceElem * max(len(segs), 1))()
    for u in range(len(offs) - 1):
        v = voices[int(vids[u])]
        for i in range(int(offs[u]), int(offs[u + 1])):
            ph = int(segs["phoneme"][i])
            arr[i].has_elem = 1 if ph >= G.PH_A else 0
            if ph >= G.PH_A:
                arr[i].elem = v.phonemes[ph - G.PH_A]
            arr[i].elem.frequency = min(float(segs["frequency"][i]), 0.5)
            arr[i].length = float(segs["length"][i])
            arr[i].blend_length = float(segs["blend_length"][i])
    return arr


@pytest.mark.parametrize("n_utt,blend", [(300, 2.0 ** -6), (3000, 2.0 ** -6), (20000, 2.0 ** -6), (3000, 0.013)])
def test_caller_built_elems_get_the_four_formant_kernels(gpu_ctx, n_utt, blend):
    """grail_synthesize_batch_elems with the elems of voices::generic(): formants 5-8 are silent in every elem of the
    batch, which the host establishes at upload over the batch's distinct elems (as it does for a voice table), so the
    four-formant kernels and pipelines serve caller-built elems too — the same bits as the PhonemeElem batch the elems
    were made from (the Selector does nothing else), and the oracle's."""
    import ctypes as C
    voices = W.single_voice()
    gpu_ctx.set_voices(voices)
    segs, offs, vids, seeds = W.make_batch(n_utt, length=0.02, blend_length=blend)
    stride = W.max_samples(length=0.02)
    arr = _as_sequence_elems(voices, segs, offs, vids)
    h = C.c_void_p()
    G._check(G.load().grail_batch_upload_elems(gpu_ctx.handle, C.cast(arr, C.c_void_p), offs.ctypes.data, vids.ctypes.data,
                                               seeds.ctypes.data, n_utt, C.byref(h)))
    ebatch = G.Batch(gpu_ctx, h, n_utt)
    pbatch = gpu_ctx.upload(segs, offs, vids, seeds)
    d_e = gpu_ctx.device_alloc(n_utt * stride * 4)
    d_p = gpu_ctx.device_alloc(n_utt * stride * 4)
    d_le = gpu_ctx.device_alloc(n_utt * 4)
    d_lp = gpu_ctx.device_alloc(n_utt * 4)
    try:
        pbatch.synthesize_async(d_p, stride, d_lp)
        gpu_ctx.sync()
        p_name, p_formants = gpu_ctx.last_kernel_name(), gpu_ctx.get_option("last_launch_formants")
        ebatch.synthesize_async(d_e, stride, d_le)
        gpu_ctx.sync()
        e_name, e_formants = gpu_ctx.last_kernel_name(), gpu_ctx.get_option("last_launch_formants")
        md, _, bad = gpu_ctx.compare(d_e, d_p, stride, d_le, d_lp, n_utt)
        rows = np.zeros((8, stride), dtype=np.float32)
        lens = np.zeros(n_utt, dtype=np.uint32)
        gpu_ctx.d2h(rows, d_e, rows.nbytes)
        gpu_ctx.d2h(lens, d_le, lens.nbytes)
    finally:
        for p in (d_e, d_p, d_le, d_lp):
            gpu_ctx.device_free(p)
        ebatch.free()
        pbatch.free()
    assert e_formants == p_formants == 4, (e_name, p_name)      # (power-of-two blend lengths or not)
    assert e_name == p_name, (e_name, p_name)
    assert float(md.max()) == 0.0 and int(bad.sum()) == 0
    ref, ref_len = O.synthesize_batch(ovoices(voices), segs[:offs[8]], offs[:9], vids[:8], seeds[:8], stride)
    assert np.array_equal(lens[:8], ref_len)
    for u in range(8):
        assert np.array_equal(rows[u, :ref_len[u]].view(np.uint32), ref[u, :ref_len[u]].view(np.uint32)), u


@pytest.mark.parametrize("n_voices", [1, 8])
@pytest.mark.parametrize("n_utt", [17, 300])
def test_small_batch_pipeline_with_any_blend_length(gpu_ctx, n_utt, n_voices):
    """Blend lengths that are not powers of two (0.3 s, 0.013 s, 1/3 s among powers of two): the pipelined workgroups
    take them too — their chain wave divides the clock by the blend length (the short exact division) where it
    multiplied by 2^-k — with four live formants and with eight, rounds of 32 and of 16 samples.  Whole 2-second
    utterances and short ones against the oracle, pipeline on and off."""
    voices = W.single_voice() if n_voices == 1 else W.preset_voices(8)
    rng = np.random.default_rng(70 + n_utt + n_voices)
    try:
        for length, r32 in ((0.5, 1), (0.5, 0), (0.037, 1)):
            segs, offs, vids, seeds = W.make_batch(n_utt, n_voices=n_voices, length=length)
            segs["blend_length"] = rng.choice([0.3, 0.013, 1.0 / 3.0, 0.25, 0.5, 0.0625], len(segs)).astype(np.float32)
            stride = W.max_samples(length=length)
            ref, ref_len = O.synthesize_batch(ovoices(voices), segs, offs, vids, seeds, stride)
            gpu_ctx.set_option("pipeline_round32", r32)
            for pipeline in (1, 0):
                gpu_ctx.set_option("small_batch_pipeline", pipeline)
                gpu_ctx.set_voices(voices)
                out, out_len = gpu_ctx.synthesize(segs, offs, vids, seeds, out_stride=stride)
                name = gpu_ctx.last_kernel_name()
                assert gpu_ctx.get_option("last_launch_pipelined") == pipeline, name
                assert "ANYBL" in name, name
                assert gpu_ctx.get_option("last_launch_formants") == (4 if n_voices == 1 and pipeline else 8), name
                assert_bit_identical(out, out_len, ref, ref_len, f"{name} length={length}")
    finally:
        gpu_ctx.set_option("small_batch_pipeline", 1)
        gpu_ctx.set_option("pipeline_round32", 1)
        gpu_ctx.set_voices(W.single_voice())


@pytest.mark.parametrize("lanes", [1, 2, 4])
def test_four_formant_lane_kernels_with_any_blend_length(gpu_ctx, lanes):
    """The four-formant lane kernels (voices::generic(): formants 5-8 never laid out) for blend lengths that are not
    powers of two: random batches against the oracle, which evaluates all eight."""
    rng = np.random.default_rng(90 + lanes)
    voices = W.single_voice()
    gpu_ctx.set_voices(voices)
    n_utt = 200
    segs, offs, vids, seeds = W.make_batch(n_utt, length=0.03)
    segs["length"] = rng.uniform(0.004, 0.05, len(segs)).astype(np.float32)
    segs["blend_length"] = rng.choice([0.3, 0.013, 1.0 / 3.0, 0.007, 0.0625, 0.05], len(segs)).astype(np.float32)
    stride = 9984
    ref, ref_len = O.synthesize_batch(ovoices(voices), segs, offs, vids, seeds, stride)
    assert ref_len.max() < stride
    gpu_ctx.set_option("lanes_per_utterance", lanes)
    try:
        out, out_len = gpu_ctx.synthesize(segs, offs, vids, seeds, out_stride=stride)
        name = gpu_ctx.last_kernel_name()
    finally:
        gpu_ctx.set_option("lanes_per_utterance", 0)
    assert "ANYBL" in name and "NFA=4" in name, name
    assert_bit_identical(out, out_len, ref, ref_len, name)


@pytest.mark.parametrize("n_utt", [200, 20000])
def test_a_batch_is_judged_by_the_voices_it_names(gpu_ctx, n_utt):
    """A table of voices::generic() next to presets with eight live formants, a voice with a formant at frequency 0 (no
    kernel family but the general one can reproduce its NaN) and one with a bandwidth of 1 Hz (no time-split warm-up):
    batches that name only voices::generic() keep the four-formant kernels, the pipelined workgroups and — in fast mode —
    the scan and time-split kernels; a batch that names one of the others gets what that voice allows.  All against the
    oracle."""
    bad = G.voice_generic(48000.0)
    bad.phonemes[1].formant_freq[6] = 0.0
    narrow = G.voice_generic(48000.0)
    narrow.phonemes[0].formant_bw[1] = 1.0 / 48000.0
    voices = [G.voice_generic(48000.0)] + W.preset_voices(8)[1:4] + [bad, narrow]
    gpu_ctx.set_voices(voices)
    segs, offs, _, seeds = W.make_batch(n_utt, length=0.03, blend_length=0.03)
    stride = W.max_samples(length=0.03)
    sample = np.arange(0, n_utt, max(1, n_utt // 24))[:24]

    def check(vids, fast):
        gpu_ctx.set_option("arithmetic", fast)
        try:
            with np.errstate(all="ignore"):
                out, out_len = gpu_ctx.synthesize(segs, offs, vids, seeds, out_stride=stride)
        finally:
            gpu_ctx.set_option("arithmetic", 0)
        name, formants = gpu_ctx.last_kernel_name(), gpu_ctx.get_option("last_launch_formants")
        sub_offs = np.zeros(len(sample) + 1, dtype=np.uint32)
        sub = []
        for i, u in enumerate(sample):
            sub.append(segs[offs[u]:offs[u + 1]])
            sub_offs[i + 1] = sub_offs[i] + offs[u + 1] - offs[u]
        with np.errstate(all="ignore"):
            ref, ref_len = O.synthesize_batch(ovoices(voices), np.concatenate(sub), sub_offs, vids[sample], seeds[sample], stride)
        for i, u in enumerate(sample):
            n = int(ref_len[i])
            assert out_len[u] == n, (name, u)
            if fast:
                fin = np.isfinite(ref[i, :n])
                peak = max(1.0, float(np.abs(ref[i, :n][fin]).max())) if fin.any() else 1.0
                assert np.array_equal(np.isnan(ref[i, :n]), np.isnan(out[u, :n])), (name, u)
                assert float(np.abs(out[u, :n][fin].astype(np.float64) - ref[i, :n][fin]).max(initial=0.0)) <= G.FAST_TOLERANCE * peak, (name, u)
            else:
                same, at = _same_but_for_nan_payloads(out[u, :n], ref[i, :n])
                assert same, (name, u, at)
        return name, formants

    only_generic = np.zeros(n_utt, dtype=np.uint32)
    name, formants = check(only_generic, 0)
    assert formants == 4, name                                    # (the presets in the table do not matter)
    name, formants = check(only_generic, 1)
    assert formants == 4 and ("scan_kernel" in name or "SPLIT" in name), name
    with_presets = (np.arange(n_utt) % 4).astype(np.uint32)
    name, formants = check(with_presets, 0)
    assert formants == 8, name
    name, formants = check(with_presets, 1)
    assert formants == 8 and ("scan_kernel" in name or "SPLIT" in name), name
    with_bad = (np.arange(n_utt) % 5).astype(np.uint32)           # names the voice with a formant at frequency 0
    name, formants = check(with_bad, 0)
    assert formants == 8, name
    name, formants = check(with_bad, 1)
    assert "scan_kernel" not in name, name
    with_narrow = np.where(np.arange(n_utt) % 7 == 0, 5, 0).astype(np.uint32)
    name, formants = check(with_narrow, 1)
    assert "SPLIT" not in name or "scan" in name, name            # (1 Hz of bandwidth: no warm-up within 16 384 samples)


@pytest.mark.parametrize("n_voices", [1, 8])
@pytest.mark.parametrize("lanes", [1, 2, 4, 8])
def test_runs_between_the_events_of_a_tile(gpu_ctx, lanes, n_voices):
    """A speech-like corpus: every lane of a wave has segment boundaries at times of its own, so most tiles hold an
    event of some lane.  The 2 / 4 / 8-lane kernels render the samples between two events by the calm tile's loops,
    as many at once as every lane's clock, jitter phase and row still allow (synth_kernel.h MIXED_RUNS) — the same
    bits as the oracle's sample-by-sample chain, with rows that end in the middle of a run (a short out_stride cuts the
    long ones) and utterances that finish while their wave goes on.
    ... and on a device of one and of two compute units ("assume_compute_units"): more waves than SIMDs, so the 2- and
    4-lane mappings take the instantiations built for two waves per SIMD, and where the launch is at most two rounds of
    the device its second round takes the launch slots in reverse order (synth_kernel.h FOLD)."""
    if lanes == 8 and n_voices == 1:
        pytest.skip("eight lanes per utterance: eight-formant layout only")
    ctx = gpu_ctx
    voices = W.single_voice() if n_voices == 1 else W.preset_voices(8)
    rng = np.random.default_rng(100 + lanes)
    segs, offs, vids, seeds, stride = W.speech_like_batch(200, rng, n_voices=n_voices, scale=0.12)
    stride = min(stride, 9000 + 4 * 13)        # cuts the longest third of the rows, not at a tile boundary
    ref, ref_len = O.synthesize_batch(ovoices(voices), segs, offs, vids, seeds, stride)
    ref_len = np.minimum(ref_len, stride)               # (the oracle counts a cut row to its end)
    saved = ctx.get_option("small_batch_pipeline")
    ctx.set_option("small_batch_pipeline", 0)           # (200 utterances would take the pipelined workgroups)
    try:
        ctx.set_voices(voices)
        ctx.set_option("lanes_per_utterance", lanes)
        two_wave = 0
        for cus in (0, 1, 2):
            ctx.set_option("assume_compute_units", cus)
            out, out_len = ctx.synthesize(segs, offs, vids, seeds, out_stride=stride, allow_truncation=True)
            name = ctx.last_kernel_name()
            assert ("L=%d" % lanes) in name and "PIPE" not in name, name
            two_wave += ",2," in name
            assert_bit_identical(out, out_len, ref, ref_len, f"speech-like L={lanes}, {n_voices} voice(s), {cus} CUs: {name}")
        # (two lanes with eight formants laid out have no two-wave instantiation; one and eight lanes never)
        # 7 waves on two lanes, 13 on four: more than the 4 SIMDs of one compute unit, the 13 more than the 8 of two
        assert two_wave == (2 if lanes == 4 else 1 if lanes == 2 and n_voices == 1 else 0), (lanes, n_voices, two_wave)
        assert (out_len == stride).sum() > 20 and (out_len < stride).sum() > 50
    finally:
        ctx.set_option("assume_compute_units", 0)
        ctx.set_option("small_batch_pipeline", saved)
        ctx.set_option("lanes_per_utterance", 0)


@pytest.mark.parametrize("round32", [2, 1, 0])
@pytest.mark.parametrize("n_voices", [1, 8])
def test_pipelined_rounds_between_the_events_of_a_tile(gpu_ctx, n_voices, round32):
    """The same corpus on the pipelined workgroups (a small batch left to the library): in a tile that holds an event of
    one of the workgroup's utterances, the stretches in which nobody has one go through the pipeline in whole rounds of 32
    or 16 samples (synth_kernel.h pipe_rounds), what is left of a stretch in 8-sample blocks by all four waves, and only
    the samples next to an event one by one — bit-identical to the oracle, rows cut in the middle of a round included."""
    ctx = gpu_ctx
    voices = W.single_voice() if n_voices == 1 else W.preset_voices(8)
    rng = np.random.default_rng(300 + n_voices)
    segs, offs, vids, seeds, stride = W.speech_like_batch(200, rng, n_voices=n_voices, scale=0.12)
    stride = min(stride, 9000 + 4 * 13)
    ref, ref_len = O.synthesize_batch(ovoices(voices), segs, offs, vids, seeds, stride)
    ref_len = np.minimum(ref_len, stride)
    try:
        ctx.set_voices(voices)
        ctx.set_option("ragged_plan", 0)            # (by its events the batch might go to a lane mapping)
        ctx.set_option("pipeline_round32", round32)
        # (spread thinly — 200 utterances on 256 compute units: one per workgroup, whose tiles then hold no event but its
        # own — and packed sixteen / eight to a workgroup)
        for spread in (1, 0):
            ctx.set_option("pipeline_spread", spread)
            out, out_len = ctx.synthesize(segs, offs, vids, seeds, out_stride=stride, allow_truncation=True)
            name = ctx.last_kernel_name()
            # (rows that differ in length take rounds of 16 unless the option insists: 2)
            assert "PIPE" in name and ("R32" if round32 == 2 else "R16") in name, name
            assert_bit_identical(out, out_len, ref, ref_len, f"speech-like, pipelined, {n_voices} voice(s), spread {spread}: {name}")
    finally:
        ctx.set_option("ragged_plan", 1)
        ctx.set_option("pipeline_round32", 1)
        ctx.set_option("pipeline_spread", 1)
