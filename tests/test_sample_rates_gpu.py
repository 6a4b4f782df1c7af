"""Voices of different sample rates in one batch.  Voice::sample_rate is per voice in the reference (src/lib.rs:696-717,
SynthesisElem::resample :418-440): the Sequencer's dt, the jitter increment and every normalised frequency follow the
utterance's voice, so the lanes of one wave tick at different rates, reach their segment boundaries at different samples
and end at different lengths.  Exact arithmetic against the oracle bit for bit on every lane mapping and on the
pipelined workgroups; fast arithmetic (lane kernels, time-split, scan) within the tolerance."""
import numpy as np
import pytest

import grail_hip as G
import oracle_lib as O

pytestmark = pytest.mark.gpu

RATES = (8000.0, 11025.0, 16000.0, 22050.0, 44100.0, 48000.0, 96000.0, 192000.0)


def _batch(rng, n_utt, seconds, ragged, rates=RATES):
    """Utterances of ~`seconds` s (in each voice's own time: lengths are seconds, frequencies per sample of ITS rate)."""
    segs, offs, vids, seeds = [], [0], [], []
    for u in range(n_utt):
        v = int(rng.integers(0, len(rates)))
        rate = rates[v]
        n_seg = int(rng.integers(2, 6)) if ragged else 4
        for i in range(n_seg):
            ph = G.PH_SILENCE if i == 0 else int(rng.choice([G.PH_A, G.PH_E, G.PH_SILENCE], p=[.45, .45, .1]))
            length = float(rng.uniform(0.6, 1.4)) * seconds / n_seg if ragged else seconds / 4
            blend = float(rng.choice([length, 2.0 ** -6, 2.0 ** -7, rng.uniform(0.004, 0.02)]))
            segs.append((ph, length, blend, np.float32(rng.uniform(90, 260)) / np.float32(rate)))
        offs.append(len(segs))
        vids.append(v)
        seeds.append(int(rng.integers(0, 2 ** 32)))
    return (G.segments(segs), np.array(offs, dtype=np.uint32), np.array(vids, dtype=np.uint32),
            np.array(seeds, dtype=np.uint32))


def _voices(rates=RATES):
    return [G.voice_generic(r) for r in rates]


def _oracle(voices, segs, offs, vids, seeds, stride):
    ov = [O.Voice.from_buffer_copy(bytes(v)) for v in voices]
    return O.synthesize_batch(ov, segs, offs, vids, seeds, stride)


@pytest.mark.parametrize("lanes", [0, 1, 2, 4, 8])
def test_mixed_sample_rates_exact_every_lane_mapping(gpu_ctx, lanes):
    rng = np.random.default_rng(31)
    voices = _voices()
    gpu_ctx.set_voices(voices)
    segs, offs, vids, seeds = _batch(rng, 160, 0.05, ragged=True)
    stride = 14016                                                    # 0.07 s at 192 kHz and some
    ref, ref_len = _oracle(voices, segs, offs, vids, seeds, stride)
    assert ref_len.max() < stride and ref_len.min() >= 200 and ref_len.max() > 8000
    # lengths follow the voice's rate: the same seconds are 24 x as many samples at 192 kHz as at 8 kHz
    assert ref_len[vids == 7].mean() > 15 * ref_len[vids == 0].mean()
    gpu_ctx.set_option("lanes_per_utterance", lanes)
    try:
        out, out_len = gpu_ctx.synthesize(segs, offs, vids, seeds, out_stride=stride)
        b = gpu_ctx.upload(segs, offs, vids, seeds)
        try:
            assert np.array_equal(b.lengths(), ref_len)
        finally:
            b.free()
    finally:
        gpu_ctx.set_option("lanes_per_utterance", 0)
    assert np.array_equal(out_len, ref_len)
    for u in range(len(ref_len)):
        assert np.array_equal(out[u, :ref_len[u]].view(np.uint32), ref[u, :ref_len[u]].view(np.uint32)), (lanes, u, vids[u])


def test_mixed_sample_rates_streamed_in_chunks(gpu_ctx):
    rng = np.random.default_rng(32)
    voices = _voices()
    gpu_ctx.set_voices(voices)
    segs, offs, vids, seeds = _batch(rng, 96, 0.04, ragged=True)
    stride = 12032
    ref, ref_len = _oracle(voices, segs, offs, vids, seeds, stride)
    assert ref_len.max() < stride
    batch = gpu_ctx.upload(segs, offs, vids, seeds)
    n = len(ref_len)
    chunk = 1000
    cstride = 1024
    d_out = gpu_ctx.device_alloc(n * cstride * 4)
    d_len = gpu_ctx.device_alloc(n * 4)
    got = np.zeros((n, stride), dtype=np.float32)
    got_len = np.zeros(n, dtype=np.uint32)
    try:
        st = G.Stream(batch)
        piece = np.zeros((n, cstride), dtype=np.float32)
        lens = np.zeros(n, dtype=np.uint32)
        for _ in range(stride // chunk + 1):
            st.next_async(chunk, d_out, cstride, d_len)
            gpu_ctx.sync()
            gpu_ctx.d2h(piece, d_out, n * cstride * 4)
            gpu_ctx.d2h(lens, d_len, n * 4)
            for u in range(n):
                got[u, got_len[u]:got_len[u] + lens[u]] = piece[u, :lens[u]]
            got_len += lens
            if not lens.any():
                break
        st.close()
    finally:
        gpu_ctx.device_free(d_out)
        gpu_ctx.device_free(d_len)
        batch.free()
    assert np.array_equal(got_len, ref_len)
    for u in range(n):
        assert np.array_equal(got[u, :ref_len[u]].view(np.uint32), ref[u, :ref_len[u]].view(np.uint32)), (u, vids[u])


@pytest.mark.parametrize("rates,tier", [(RATES[:6], 1), (RATES, 2)])
@pytest.mark.parametrize("n_utt", [96, 900, 3000])
def test_mixed_sample_rates_fast_within_the_tolerance(gpu_ctx, n_utt, rates, tier):
    """Aligned utterances of 0.25 s in six or eight sample rates (2 000 ... 48 000 samples each) through whatever the
    cost model picks for the size in fast mode.  Up to 48 kHz voices::generic() is inside the first tier (sharpness
    17 - 26); at 96 and 192 kHz its resonances are narrower per sample (45, 89): such a batch gets the second tier, or
    the exact kernels where those are the cheaper way to the same tolerance."""
    rng = np.random.default_rng(33 + n_utt + tier)
    voices = _voices(rates)
    assert (max(G.fast_sharpness(v) for v in voices) <= G.FAST_SHARPNESS_LIMIT) == (tier == 1)
    gpu_ctx.set_voices(voices)
    segs, offs, vids, seeds = _batch(rng, n_utt, 0.25, ragged=False, rates=rates)
    stride = 48128
    sample = rng.choice(n_utt, size=40, replace=False)
    sub_offs = np.zeros(len(sample) + 1, dtype=np.uint32)
    sub_segs = []
    for i, u in enumerate(sample):
        sub_segs.append(segs[offs[u]:offs[u + 1]])
        sub_offs[i + 1] = sub_offs[i] + (offs[u + 1] - offs[u])
    ref, ref_len = _oracle(voices, np.concatenate(sub_segs), sub_offs, vids[sample], seeds[sample], stride)
    gpu_ctx.set_option("arithmetic", 1)
    try:
        out, out_len = gpu_ctx.synthesize(segs, offs, vids, seeds, out_stride=stride)
        name = gpu_ctx.last_kernel_name()
        served = gpu_ctx.get_option("last_launch_fast")
    finally:
        gpu_ctx.set_option("arithmetic", 0)
    if tier == 1:
        assert served == 1, name
    worst = 0.0
    for i, u in enumerate(sample):
        assert out_len[u] == ref_len[i], (u, vids[u], name)
        n = int(ref_len[i])
        peak = max(1.0, float(np.abs(ref[i, :n]).max()))
        worst = max(worst, float(np.abs(out[u, :n] - ref[i, :n]).max()) / peak)
    print(f"mixed sample rates up to {rates[-1]:.0f} Hz, {n_utt} utterances (ran {name}, arithmetic {served}): "
          f"worst |fast - oracle| = {worst * 2 ** 23:.1f} * 2^-23")
    assert worst <= G.FAST_TOLERANCE


def test_long_segments_cross_the_binades_of_the_clock(gpu_ctx):
    """Segments of 2 - 9 s with blends of 0.25 - 8 s (the Sequencer clock starts each segment at its length and walks
    down through the binades 8, 4, 2, 1, ... in steps of dt: where the scan kernel's closed-form clock and the
    time-split fast-forward have to reproduce the serial f32 accumulation), at 16 and 22.05 kHz so that half a minute is
    half a million samples: exact kernels bit for bit, every fast family within the tolerance
    (tools/long_utterance_check.py is the same at 48 kHz, by hand)."""
    rng = np.random.default_rng(41)
    rates = (16000.0, 22050.0)
    voices = _voices(rates)
    gpu_ctx.set_voices(voices)
    n_utt = 12
    segs, offs, vids, seeds = _batch(rng, n_utt, 1.0, ragged=False, rates=rates)
    k = len(segs)
    segs["length"] = rng.uniform(2.0, 9.0, k).astype(np.float32)
    segs["blend_length"] = rng.choice([0.5, 2.0, 0.25, 4.0, 8.0, 3.3], k).astype(np.float32)
    stride = (int(4 * 9.0 * 22050) + 64 + 63) // 64 * 64
    ref, ref_len = _oracle(voices, segs, offs, vids, seeds, stride)
    assert ref_len.max() < stride and ref_len.min() > 60000
    ran = []
    try:
        for fast, opts in ((0, {}), (0, {"lanes_per_utterance": 1}), (0, {"lanes_per_utterance": 8}), (1, {}),
                           (1, {"time_split": 0, "time_parallel_scan": 0, "lanes_per_utterance": 1}),
                           (1, {"time_parallel_scan": 0}), (1, {"time_split_chunks": 16}), (1, {"time_split_chunks": 64})):
            gpu_ctx.set_option("arithmetic", fast)
            for name, value in opts.items():
                gpu_ctx.set_option(name, value)
            try:
                out, out_len = gpu_ctx.synthesize(segs, offs, vids, seeds, out_stride=stride)
                kernel = gpu_ctx.last_kernel_name()
            finally:
                for name in opts:
                    gpu_ctx.set_option(name, 1 if name in ("time_split", "time_parallel_scan") else 0)
            assert np.array_equal(out_len, ref_len), (fast, opts, kernel)
            worst = 0.0
            for u in range(n_utt):
                n = int(ref_len[u])
                if fast:
                    peak = max(1.0, float(np.abs(ref[u, :n]).max()))
                    worst = max(worst, float(np.abs(out[u, :n] - ref[u, :n]).max()) / peak)
                else:
                    assert np.array_equal(out[u, :n].view(np.uint32), ref[u, :n].view(np.uint32)), (opts, kernel, u)
            assert worst <= G.FAST_TOLERANCE, (opts, kernel, worst * 2 ** 23)
            ran.append(f"{kernel}: {'%.1f * 2^-23' % (worst * 2 ** 23) if fast else 'bit-exact'}")
    finally:
        gpu_ctx.set_option("arithmetic", 0)
    print("long segments:", "; ".join(ran))
    assert any("scan_kernel" in r for r in ran) and any("SPLIT" in r for r in ran)


def test_mixed_sample_rates_on_the_time_split_grid_of_rows_that_differ(gpu_ctx):
    """Rows of different lengths AND sample rates on the time-split kernels: the length bound that lets a chunk's lane skip
    an utterance which ends before the chunk begins is taken at the table's HIGHEST rate (an utterance of a slower voice is
    shorter in samples than its bound says: it is skipped later than it could be, never too early).  Lengths equal to the
    oracle's for every row, sampled rows within the tolerance."""
    rates = RATES[:6]
    rng = np.random.default_rng(77)
    voices = _voices(rates)
    gpu_ctx.set_voices(voices)
    n_utt = 1500
    segs, offs, vids, seeds = _batch(rng, n_utt, 0.5, ragged=True, rates=rates)
    stride = 48128
    try:
        gpu_ctx.set_option("assume_compute_units", 18)          # 24 waves per chunk on 72 SIMDs: a coarse grid
        gpu_ctx.set_option("time_parallel_scan", 0)
        gpu_ctx.set_option("ragged_plan", 0)
        gpu_ctx.set_option("time_split_min_utterances", 0)
        gpu_ctx.set_option("arithmetic", 1)
        out, out_len = gpu_ctx.synthesize(segs, offs, vids, seeds, out_stride=stride)
        name, chunks = gpu_ctx.last_kernel_name(), gpu_ctx.get_option("last_launch_chunks")
    finally:
        for k, v in (("arithmetic", 0), ("time_split_min_utterances", -1), ("ragged_plan", 1), ("time_parallel_scan", 1),
                     ("assume_compute_units", 0)):
            gpu_ctx.set_option(k, v)
    assert "SPLIT" in name and chunks >= 3, (name, chunks)
    ref, ref_len = _oracle(voices, segs, offs, vids, seeds, stride)
    assert np.array_equal(out_len, ref_len)
    worst = 0.0
    for u in rng.choice(n_utt, size=120, replace=False):
        n = int(ref_len[u])
        peak = max(1.0, float(np.abs(ref[u, :n]).max()))
        worst = max(worst, float(np.abs(out[u, :n] - ref[u, :n]).max()) / peak)
    assert 0.0 < worst <= G.FAST_TOLERANCE, worst * 2 ** 23


def test_a_batch_uploaded_on_a_48_khz_context_and_rendered_on_a_192_khz_one(built):
    """One batch, two contexts (a batch may be rendered by several contexts): uploaded where the table's highest rate is
    48 kHz — the per-utterance length bound of the time-split kernels is computed for THAT rate — and rendered by a context
    whose table is at 192 kHz, where every utterance is four times as long in samples.  Both contexts have installed
    exactly one table: a per-context epoch counter would call the bound current ("1 == 1"), the chunks beyond it would be
    skipped and the rows would come out truncated.  Voice-table epochs are unique in the process, so the second context
    renders without the bound: lengths equal to the oracle's at 192 kHz, rows within the tolerance."""
    rng = np.random.default_rng(4242)
    slow, quick = _voices((48000.0,)), _voices((192000.0,))
    n_utt = 1500
    segs, offs, vids, seeds = _batch(rng, n_utt, 0.125, ragged=True, rates=(192000.0,))
    stride = 48128
    L = G.load()
    with G.Context(0) as a, G.Context(0) as b:
        a.set_voices(slow)
        b.set_voices(quick)
        batch = a.upload(segs, offs, vids, seeds)
        d_out, d_len = b.device_alloc(n_utt * stride * 4), b.device_alloc(n_utt * 4)
        try:
            for k, v in (("assume_compute_units", 18), ("time_parallel_scan", 0), ("ragged_plan", 0),
                         ("time_split_min_utterances", 0), ("arithmetic", 1)):
                b.set_option(k, v)
            b.memset(d_out, 0, n_utt * stride * 4)
            G._check(L.grail_batch_synthesize_async(b.handle, batch.handle, d_out, stride, d_len))
            b.sync()
            name, chunks = b.last_kernel_name(), b.get_option("last_launch_chunks")
            out = np.zeros((n_utt, stride), dtype=np.float32)
            out_len = np.zeros(n_utt, dtype=np.uint32)
            b.d2h(out, d_out, out.nbytes)
            b.d2h(out_len, d_len, out_len.nbytes)
            # ... and the uploading context still uses its bound, on its own table
            for k, v in (("assume_compute_units", 18), ("time_parallel_scan", 0), ("ragged_plan", 0),
                         ("time_split_min_utterances", 0), ("arithmetic", 1)):
                a.set_option(k, v)
            own_len = np.zeros(n_utt, dtype=np.uint32)
            d_out_a, d_len_a = a.device_alloc(n_utt * stride * 4), a.device_alloc(n_utt * 4)
            batch.synthesize_async(d_out_a, stride, d_len_a)
            a.sync()
            a.d2h(own_len, d_len_a, own_len.nbytes)
            a.device_free(d_out_a)
            a.device_free(d_len_a)
        finally:
            b.device_free(d_out)
            b.device_free(d_len)
            batch.free()
    assert "SPLIT" in name and chunks >= 3, (name, chunks)
    ref, ref_len = _oracle(quick, segs, offs, vids, seeds, stride)
    assert np.array_equal(out_len, ref_len)
    assert np.array_equal(own_len, _oracle(slow, segs, offs, vids, seeds, stride)[1])
    worst = 0.0
    for u in rng.choice(n_utt, size=60, replace=False):
        n = int(ref_len[u])
        peak = max(1.0, float(np.abs(ref[u, :n]).max()))
        worst = max(worst, float(np.abs(out[u, :n] - ref[u, :n]).max()) / peak)
    assert 0.0 < worst <= G.FAST_TOLERANCE, worst * 2 ** 23
