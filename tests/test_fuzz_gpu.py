"""Randomised parity: random utterances (phonemes incl. Silence/Stop/Glide, lengths, blend
lengths — power-of-two and not — pitches, voices, seeds, ragged segment counts) through the HIP
path vs the oracle, bit for bit, for every lane mapping; the streaming path on the same inputs."""
import os

import numpy as np
import pytest

import grail_hip as G
import oracle_lib as O
from grail_hip import workload as W

pytestmark = pytest.mark.gpu


def random_batch(rng, n_utt, n_voices, rate):
    segs, offs, vids, seeds = [], [0], [], []
    pow2 = [2.0 ** -k for k in range(5, 9)]
    for _ in range(n_utt):
        for _ in range(int(rng.integers(0, 7))):
            ph = int(rng.choice([G.PH_SILENCE, G.PH_STOP, G.PH_GLIDE, G.PH_A, G.PH_E], p=[.15, .05, .05, .4, .35]))
            length = float(rng.choice([rng.uniform(0.0005, 0.03), rng.choice(pow2), 0.0]))
            blend = float(rng.choice([rng.uniform(0.0005, 0.04), rng.choice(pow2), rng.choice(pow2)]))
            hz = float(rng.uniform(60, 400))
            segs.append((ph, length, blend, np.float32(hz) / np.float32(rate)))
        offs.append(len(segs))
        vids.append(int(rng.integers(0, n_voices)))
        seeds.append(int(rng.integers(0, 2 ** 32)))
    return (G.segments(segs), np.array(offs, dtype=np.uint32), np.array(vids, dtype=np.uint32),
            np.array(seeds, dtype=np.uint32))


# GRAIL_FUZZ_EXTRA=n adds n more seeds (soak runs; the default suite stays short)
EXTRA_SEEDS = list(range(100, 100 + int(os.environ.get("GRAIL_FUZZ_EXTRA", "0"))))


@pytest.mark.parametrize("seed", [1, 2, 3] + EXTRA_SEEDS)
def test_random_batches_every_lane_mapping(gpu_ctx, seed):
    rng = np.random.default_rng(seed)
    voices = W.preset_voices(8) if seed != 2 else [G.voice_generic(48000.0), G.voice_generic(44100.0)]
    gpu_ctx.set_voices(voices)
    segs, offs, vids, seeds = random_batch(rng, 130, len(voices), 48000.0)
    ov = [O.Voice.from_buffer_copy(bytes(v)) for v in voices]
    stride = 10048
    ref, ref_len = O.synthesize_batch(ov, segs, offs, vids, seeds, stride)
    assert ref_len.max() < stride and ref_len.max() > 3000
    try:
        for lanes in (1, 2, 4, 8):
            gpu_ctx.set_option("lanes_per_utterance", lanes)
            out, out_len = gpu_ctx.synthesize(segs, offs, vids, seeds, out_stride=stride)
            assert np.array_equal(out_len, ref_len), lanes
            for u in range(len(ref_len)):
                assert np.array_equal(out[u, :ref_len[u]].view(np.uint32),
                                      ref[u, :ref_len[u]].view(np.uint32)), (lanes, u)
        # rows with zero-length segments planned apart from the others whatever the cost model says ("row_groups" = 2)
        gpu_ctx.set_option("row_groups", 2)
        for lanes in (0, 1, 4):
            gpu_ctx.set_option("lanes_per_utterance", lanes)
            out, out_len = gpu_ctx.synthesize(segs, offs, vids, seeds, out_stride=stride)
            assert gpu_ctx.get_option("last_launch_blocks") >= 2, lanes
            assert np.array_equal(out_len, ref_len), lanes
            for u in range(len(ref_len)):
                assert np.array_equal(out[u, :ref_len[u]].view(np.uint32),
                                      ref[u, :ref_len[u]].view(np.uint32)), ("row groups", lanes, u)
        gpu_ctx.set_option("row_groups", 1)
        # the batch pre-pass agrees with what was rendered
        b = gpu_ctx.upload(segs, offs, vids, seeds)
        try:
            assert np.array_equal(b.lengths(), ref_len)
        finally:
            b.free()
    finally:
        gpu_ctx.set_option("lanes_per_utterance", 0)
        gpu_ctx.set_option("row_groups", 1)


@pytest.mark.parametrize("seed", [11, 12] + EXTRA_SEEDS)
def test_random_voice_tables_every_lane_mapping(gpu_ctx, seed):
    """Random phoneme tables (per-formant smoothness, random breath / turbulence, some amplitudes
    exactly zero) and random jitter settings: the vector-smoothness loops, partly silent formants
    and fast jitter wraps, against the oracle bit for bit."""
    from test_oracle_crosscheck import random_voice
    rng = np.random.default_rng(seed)
    ovoices = [random_voice(rng, 48000.0) for _ in range(6)]
    voices = [G.Voice.from_buffer_copy(bytes(v)) for v in ovoices]
    gpu_ctx.set_voices(voices)
    segs, offs, vids, seeds = random_batch(rng, 100, len(voices), 48000.0)
    stride = 10048
    ref, ref_len = O.synthesize_batch(ovoices, segs, offs, vids, seeds, stride)
    assert ref_len.max() < stride
    try:
        for lanes in (1, 2, 4, 8):
            gpu_ctx.set_option("lanes_per_utterance", lanes)
            out, out_len = gpu_ctx.synthesize(segs, offs, vids, seeds, out_stride=stride)
            assert np.array_equal(out_len, ref_len), lanes
            for u in range(len(ref_len)):
                assert np.array_equal(out[u, :ref_len[u]].view(np.uint32),
                                      ref[u, :ref_len[u]].view(np.uint32)), (lanes, u)
    finally:
        gpu_ctx.set_option("lanes_per_utterance", 0)


def four_formant_voice(rng, rate):
    """A random voice that qualifies for the four-formant kernels: formants 5-8 silent in every
    phoneme with ordinary parameters, random formants 1-4 and jitter settings."""
    from test_oracle_crosscheck import random_voice
    v = random_voice(rng, rate)
    g = O.voice_generic(rate)
    for p in range(2):
        for i in range(4, 8):
            v.phonemes[p].formant_amp.v[i] = 0.0          # oracle_lib's Array wraps `v`
            for field in ("formant_freq", "formant_bw", "formant_smooth", "formant_breath", "formant_turb"):
                getattr(v.phonemes[p], field).v[i] = getattr(g.phonemes[p], field).v[i]
    v.jitter_delta_amplitude = float(np.float32(rng.uniform(0, 0.5)))
    v.jitter_delta_formant_frequency = float(np.float32(rng.uniform(0, 20)) / np.float32(rate))
    return v


@pytest.mark.parametrize("seed", [21, 22, 23] + EXTRA_SEEDS)
def test_random_batches_four_formant_path(gpu_ctx, seed):
    """Random batches that pass the four-formant gate (power-of-two blend lengths, segments of at
    least two samples, ordinary pitches): the kernels that lay out formants 1-4 only, for 1, 2 and 4
    lanes per utterance, against the oracle (which evaluates all eight) bit for bit."""
    rng = np.random.default_rng(seed)
    ovoices = [four_formant_voice(rng, 48000.0) for _ in range(3)]
    voices = [G.Voice.from_buffer_copy(bytes(v)) for v in ovoices]
    gpu_ctx.set_voices(voices)
    segs, offs, vids, seeds = [], [0], [], []
    for _ in range(90):
        for _ in range(int(rng.integers(0, 6))):
            ph = int(rng.choice([G.PH_SILENCE, G.PH_STOP, G.PH_GLIDE, G.PH_A, G.PH_E], p=[.15, .05, .05, .4, .35]))
            length = float(rng.choice([rng.uniform(0.0005, 0.03), 2.0 ** -int(rng.integers(5, 9)), 2.0 / 48000.0]))
            segs.append((ph, length, 2.0 ** -int(rng.integers(4, 11)), np.float32(rng.uniform(60, 400)) / np.float32(48000.0)))
        offs.append(len(segs))
        vids.append(int(rng.integers(0, len(voices))))
        seeds.append(int(rng.integers(0, 2 ** 32)))
    segs = G.segments(segs)
    offs, vids, seeds = (np.array(a, dtype=np.uint32) for a in (offs, vids, seeds))
    stride = 10048
    ref, ref_len = O.synthesize_batch(ovoices, segs, offs, vids, seeds, stride)
    assert ref_len.max() < stride
    try:
        # (segments of 0.5 - 30 ms: weighed by its events the batch would go to a lane mapping — this test wants the pipeline)
        gpu_ctx.set_option("ragged_plan", 0)
        for lanes in (0, 1, 2, 4):          # 0 = auto: the small-batch pipeline (four-wave workgroups)
            gpu_ctx.set_option("lanes_per_utterance", lanes)
            out, out_len = gpu_ctx.synthesize(segs, offs, vids, seeds, out_stride=stride)
            assert gpu_ctx.get_option("last_launch_formants") == 4, lanes
            assert gpu_ctx.get_option("last_launch_pipelined") == (1 if lanes == 0 else 0), lanes
            assert np.array_equal(out_len, ref_len), lanes
            for u in range(len(ref_len)):
                assert np.array_equal(out[u, :ref_len[u]].view(np.uint32),
                                      ref[u, :ref_len[u]].view(np.uint32)), (lanes, u)
    finally:
        gpu_ctx.set_option("lanes_per_utterance", 0)
        gpu_ctx.set_option("ragged_plan", 1)


def pow2_blend_batch(rng, n_utt, n_voices):
    """Random utterances whose blend lengths are powers of two (what the pipelined workgroups take)."""
    segs, offs, vids, seeds = [], [0], [], []
    for _ in range(n_utt):
        for _ in range(int(rng.integers(0, 6))):
            ph = int(rng.choice([G.PH_SILENCE, G.PH_STOP, G.PH_GLIDE, G.PH_A, G.PH_E], p=[.15, .05, .05, .4, .35]))
            length = float(rng.choice([rng.uniform(0.0005, 0.03), 2.0 ** -int(rng.integers(5, 9)), 2.0 / 48000.0]))
            segs.append((ph, length, 2.0 ** -int(rng.integers(4, 11)), np.float32(rng.uniform(60, 400)) / np.float32(48000.0)))
        offs.append(len(segs))
        vids.append(int(rng.integers(0, n_voices)))
        seeds.append(int(rng.integers(0, 2 ** 32)))
    return (G.segments(segs),) + tuple(np.array(a, dtype=np.uint32) for a in (offs, vids, seeds))


@pytest.mark.parametrize("seed", [31, 32] + EXTRA_SEEDS)
def test_random_batches_eight_formant_pipeline(gpu_ctx, seed):
    """Random voice tables with all eight formants live and power-of-two blends: the pipelined workgroups with
    eight utterances each, rounds of 32 and of 16 samples, against the oracle bit for bit."""
    from test_oracle_crosscheck import random_voice
    rng = np.random.default_rng(seed)
    ovoices = [random_voice(rng, 48000.0) for _ in range(4)]
    for v in ovoices:                     # (random_voice zeroes some amplitudes: keep the eight-formant path)
        for p in range(2):
            v.phonemes[p].formant_amp.v[7] = max(float(v.phonemes[p].formant_amp.v[7]), 0.01)
    voices = [G.Voice.from_buffer_copy(bytes(v)) for v in ovoices]
    gpu_ctx.set_voices(voices)
    segs, offs, vids, seeds = pow2_blend_batch(rng, 75, len(voices))
    stride = 10048
    ref, ref_len = O.synthesize_batch(ovoices, segs, offs, vids, seeds, stride)
    assert ref_len.max() < stride
    try:
        gpu_ctx.set_option("ragged_plan", 0)         # (as above: by its events the batch would go to a lane mapping)
        for round32 in (2, 0):            # (2: rounds of 32 although the rows differ in length)
            gpu_ctx.set_option("pipeline_round32", round32)
            out, out_len = gpu_ctx.synthesize(segs, offs, vids, seeds, out_stride=stride)
            assert gpu_ctx.get_option("last_launch_pipelined") == 1 and gpu_ctx.get_option("last_launch_formants") == 8
            assert ("R32" if round32 else "R16") in gpu_ctx.last_kernel_name()
            assert np.array_equal(out_len, ref_len), round32
            for u in range(len(ref_len)):
                assert np.array_equal(out[u, :ref_len[u]].view(np.uint32),
                                      ref[u, :ref_len[u]].view(np.uint32)), (round32, u)
    finally:
        gpu_ctx.set_option("pipeline_round32", 1)
        gpu_ctx.set_option("ragged_plan", 1)


@pytest.mark.parametrize("seed", [41, 42] + EXTRA_SEEDS)
def test_random_batches_streamed_in_random_chunks(gpu_ctx, seed):
    """grail_stream_*: the random batches pulled in chunks of random sizes (1 .. 3000 samples, a new size every
    call), every lane mapping: the concatenation is the oracle's rendering bit for bit."""
    from test_stream_gpu import stream_all
    rng = np.random.default_rng(seed)
    voices = W.preset_voices(8) if seed % 2 else [G.voice_generic(48000.0), G.voice_generic(44100.0)]
    gpu_ctx.set_voices(voices)
    segs, offs, vids, seeds = random_batch(rng, 60, len(voices), 48000.0)
    ov = [O.Voice.from_buffer_copy(bytes(v)) for v in voices]
    ref, ref_len = O.synthesize_batch(ov, segs, offs, vids, seeds, 10048)
    try:
        for lanes in (0, 1, 2, 4, 8):
            gpu_ctx.set_option("lanes_per_utterance", lanes)
            chunks = [int(c) for c in rng.choice([1, 2, 63, 64, 65, 500, 1024, 3000], 24)] + \
                     [int(c) for c in rng.integers(1, 3001, 24)]
            b = gpu_ctx.upload(segs, offs, vids, seeds)
            try:
                got = stream_all(gpu_ctx, b, len(ref_len), chunks, stride=3008)
            finally:
                b.free()
            for u in range(len(ref_len)):
                assert len(got[u]) == ref_len[u], (lanes, u)
                assert np.array_equal(got[u].view(np.uint32), ref[u, :ref_len[u]].view(np.uint32)), (lanes, u)
        # the same in fast arithmetic: the lengths are the oracle's, every sample within the tolerance
        gpu_ctx.set_option("arithmetic", 1)
        for lanes in (0, 1, 4):
            gpu_ctx.set_option("lanes_per_utterance", lanes)
            chunks = [int(c) for c in rng.integers(1, 3001, 32)]
            b = gpu_ctx.upload(segs, offs, vids, seeds)
            try:
                got = stream_all(gpu_ctx, b, len(ref_len), chunks, stride=3008)
            finally:
                b.free()
            for u in range(len(ref_len)):
                assert len(got[u]) == ref_len[u], (lanes, u)
                if ref_len[u]:
                    want = ref[u, :ref_len[u]]
                    d = float(np.abs(got[u].astype(np.float64) - want).max())
                    assert d <= G.FAST_TOLERANCE * max(1.0, float(np.abs(want).max())), (lanes, u, d * 2.0 ** 23)
    finally:
        gpu_ctx.set_option("arithmetic", 0)
        gpu_ctx.set_option("lanes_per_utterance", 0)


def random_sequence_elems(rng, n_utt, tame):
    """Caller-built SequenceElems, every segment with an elem of its own (blends between arbitrary parameter sets).
    tame: resonances fast arithmetic is served for (formants below 0.05 fs, bandwidths of 0.008 fs and more: the
    sharpness of a batch of elems takes the worst of every formant over all of them)."""
    gsegs, osegs, offs = [], [], [0]
    for _ in range(n_utt):
        for _ in range(int(rng.integers(0, 6))):
            has = bool(rng.integers(0, 5))
            e = np.zeros(49, dtype=np.float32)
            e[0] = rng.uniform(0.0015, 0.009)
            e[1:9] = rng.uniform(0.004, 0.05 if tame else 0.3, 8)
            e[9:17] = rng.uniform(0.008, 0.02, 8) if tame else rng.uniform(0.002, 0.012, 8)
            e[17:25] = rng.uniform(0.01, 0.1) if rng.integers(0, 2) else rng.uniform(0.01, 0.1, 8)
            e[25:33] = rng.uniform(0, 1, 8)
            e[33:41] = rng.uniform(0, 1, 8)
            amp = rng.uniform(0, 1, 8) * (rng.uniform(0, 1, 8) > 0.25)
            e[41:49] = amp / max(float(amp.sum()), 1e-3)
            ln = float(rng.choice([rng.uniform(0.002, 0.05), 2.0 ** -int(rng.integers(5, 8))]))
            bl = float(rng.choice([rng.uniform(0.002, 0.05), 2.0 ** -int(rng.integers(5, 9))]))
            gsegs.append(G.SequenceElem(int(has), G.SynthesisElem.from_np(e), ln, bl))
            osegs.append(O.SequenceElem(int(has), O.SynthesisElem.from_buffer_copy(e.tobytes()), ln, bl))
        offs.append(len(gsegs))
    return gsegs, osegs, offs


@pytest.mark.parametrize("seed", [51, 52] + EXTRA_SEEDS)
def test_random_sequence_elems_both_arithmetics(gpu_ctx, seed):
    """grail_synthesize_batch_elems on random caller-built elems: exact arithmetic bit for bit for every lane
    mapping; fast arithmetic within the tolerance (even seeds: tame elems, the fast kernels run; odd seeds: any,
    and what is too sharp for them goes to the second tier: the reference's own filter coefficients)."""
    rng = np.random.default_rng(seed)
    tame = seed % 2 == 0
    v = G.voice_generic(48000.0)
    gpu_ctx.set_voices([v])
    ov = O.Voice.from_buffer_copy(bytes(v))
    n_utt = 60
    gsegs, osegs, offs = random_sequence_elems(rng, n_utt, tame)
    seeds = rng.integers(0, 2 ** 32, n_utt, dtype=np.uint64).astype(np.uint32)
    refs = [O.synthesize_sequence(ov, osegs[offs[u]:offs[u + 1]], int(seeds[u])) for u in range(n_utt)]
    stride = max(max(len(r) for r in refs) + 64, 64)
    try:
        for lanes in (0, 1, 2, 4, 8):
            gpu_ctx.set_option("lanes_per_utterance", lanes)
            out, out_len = gpu_ctx.synthesize_elems(gsegs, offs, None, seeds, out_stride=stride)
            for u in range(n_utt):
                assert out_len[u] == len(refs[u]), (lanes, u)
                assert np.array_equal(out[u, :len(refs[u])].view(np.uint32), refs[u].view(np.uint32)), (lanes, u)
        gpu_ctx.set_option("arithmetic", 1)
        ran_fast = False
        for lanes in (0, 1, 4):
            gpu_ctx.set_option("lanes_per_utterance", lanes)
            out, out_len = gpu_ctx.synthesize_elems(gsegs, offs, None, seeds, out_stride=stride)
            ran_fast = ran_fast or "FAST" in gpu_ctx.last_kernel_name()
            for u in range(n_utt):
                assert out_len[u] == len(refs[u]), (lanes, u)
                if len(refs[u]):
                    d = float(np.abs(out[u, :len(refs[u])].astype(np.float64) - refs[u]).max())
                    assert d <= G.FAST_TOLERANCE * max(1.0, float(np.abs(refs[u]).max())), (lanes, u, d * 2.0 ** 23)
        assert ran_fast or not tame
    finally:
        gpu_ctx.set_option("arithmetic", 0)
        gpu_ctx.set_option("lanes_per_utterance", 0)


@pytest.mark.parametrize("seed", [61, 62] + EXTRA_SEEDS)
def test_random_row_capacities(gpu_ctx, seed):
    """Rows too short for some utterances, aligned to 64 samples and not: every kernel family reports the cut
    (GRAIL_ERR_BUFFER_TOO_SMALL), stores min(length, capacity) samples per row and those are the reference's —
    bit for bit in exact arithmetic, within the tolerance in fast arithmetic (lane kernels, scan kernel, time-split
    kernels: there the lane of the chunk the capacity falls into ends the row)."""
    rng = np.random.default_rng(seed)
    voices = W.preset_voices(8) if seed % 2 else W.single_voice()
    gpu_ctx.set_voices(voices)
    n_utt = 90
    segs, offs, vids, seeds = W.make_batch(n_utt, n_voices=len(voices))
    k = len(segs)
    segs["length"] = rng.uniform(0.01, 0.12, k).astype(np.float32)
    segs["blend_length"] = rng.choice([0.03125, 0.0625, 0.02, 0.05], k).astype(np.float32)
    ov = [O.Voice.from_buffer_copy(bytes(v)) for v in voices]
    ref, ref_len = O.synthesize_batch(ov, segs, offs, vids, seeds, 4 * 5760 + 64)
    assert ref_len.max() > 12000
    try:
        for cap in (int(rng.integers(3000, 12000)) // 64 * 64, int(rng.integers(3000, 12000)) | 1, int(rng.integers(40, 900))):
            want_len = np.minimum(ref_len, cap).astype(np.uint32)
            cut = bool((ref_len > cap).any())
            for fast, opts in ((0, {"lanes_per_utterance": 0}), (0, {"lanes_per_utterance": 1}), (0, {"lanes_per_utterance": 4}),
                               (0, {"lanes_per_utterance": 8}), (1, {"lanes_per_utterance": 1}), (1, {"lanes_per_utterance": 4}),
                               (1, {"time_split": 0}), (1, {"time_split_chunks": 3})):
                gpu_ctx.set_option("arithmetic", fast)
                for name, val in opts.items():
                    gpu_ctx.set_option(name, val)
                if cut:
                    with pytest.raises(G.GrailError) as ei:
                        gpu_ctx.synthesize(segs, offs, vids, seeds, out_stride=cap)
                    assert ei.value.status == G.ERR_BUFFER_TOO_SMALL, (cap, fast, opts)
                out, out_len = gpu_ctx.synthesize(segs, offs, vids, seeds, out_stride=cap, allow_truncation=True)
                kernel = gpu_ctx.last_kernel_name()
                for name in opts:
                    gpu_ctx.set_option(name, 1 if name == "time_split" else 0)
                assert np.array_equal(out_len, want_len), (cap, fast, opts, kernel)
                for u in range(n_utt):
                    n = int(want_len[u])
                    if fast == 0:
                        assert np.array_equal(out[u, :n].view(np.uint32), ref[u, :n].view(np.uint32)), (cap, opts, kernel, u)
                    elif n:
                        d = float(np.abs(out[u, :n].astype(np.float64) - ref[u, :n]).max())
                        assert d <= G.FAST_TOLERANCE * max(1.0, float(np.abs(ref[u, :n]).max())), (cap, opts, kernel, u, d * 2 ** 23)
    finally:
        gpu_ctx.set_option("arithmetic", 0)
        for name, val in (("lanes_per_utterance", 0), ("time_split", 1), ("time_split_chunks", 0)):
            gpu_ctx.set_option(name, val)
