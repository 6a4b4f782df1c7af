"""Live streams (grail_stream_open_live / grail_stream_append): the lazy source of examples/interactive.rs:31-48.
ONE chain per utterance runs for the whole session; segments are appended while samples are being pulled, and a
Sequencer that needs a segment which is not there yet pauses (src/lib.rs:866-888 pulls iter.next() on demand).  Whatever
the interleaving of appends and pulls, the samples must be those of the oracle's one-shot rendering of everything that
was appended — bit for bit in exact arithmetic, for every lane mapping."""
import ctypes as C

import numpy as np
import pytest

import grail_hip as G
import oracle_lib as O
from grail_hip import workload as W
from test_parity_gpu import _elem

pytestmark = pytest.mark.gpu
ULP = 2.0 ** -23


def _ovoices(voices):
    return [O.Voice.from_buffer_copy(bytes(v)) for v in voices]


def _script(rng, n_utt, n_voices, max_segs=9):
    """Per utterance a list of segments (mixed phonemes, 4 - 30 ms, blend lengths of 2^-k s and a few odd ones)."""
    per_utt = []
    for u in range(n_utt):
        k = int(rng.integers(1, max_segs + 1))
        ph = rng.choice([G.PH_A, G.PH_E, G.PH_SILENCE, G.PH_STOP], size=k, p=[0.4, 0.4, 0.15, 0.05])
        ln = rng.uniform(0.004, 0.03, size=k).astype(np.float32)
        bl = np.where(rng.random(k) < 0.8, 2.0 ** -rng.integers(5, 9, size=k), rng.uniform(0.003, 0.02, size=k)).astype(np.float32)
        hz = (rng.uniform(90.0, 220.0, size=k) / 48000.0).astype(np.float32)
        per_utt.append(G.segments(list(zip(ph.tolist(), ln.tolist(), bl.tolist(), hz.tolist()))))
    return per_utt


def _drive(ctx, stream, per_utt, rng, stride, chunk_choices, finish_early=False, pcm16_every=0):
    """Append the script piece by piece at random points between pulls of random sizes; returns the rows."""
    n_utt = len(per_utt)
    fed = [0] * n_utt
    rows = [[] for _ in range(n_utt)]
    d_out = ctx.device_alloc(n_utt * stride * 4)
    d_len = ctx.device_alloc(n_utt * 4)
    finished = False
    try:
        for it in range(100000):
            # feed: each utterance gets 0..3 of its remaining segments, with probability 1/2 per round
            if not finished:
                segs, offs = [], [0]
                for u in range(n_utt):
                    k = int(rng.integers(0, 4)) if rng.random() < 0.5 else 0
                    k = min(k, len(per_utt[u]) - fed[u])
                    segs.append(per_utt[u][fed[u]:fed[u] + k])
                    fed[u] += k
                    offs.append(offs[-1] + k)
                if offs[-1]:
                    try:
                        stream.append(np.concatenate(segs), offs)
                    except G.GrailError as e:
                        # a ring is full (nothing was appended): take it back, pull first, feed again next round
                        assert e.status == G.ERR_BUFFER_TOO_SMALL
                        for u in range(n_utt):
                            fed[u] -= offs[u + 1] - offs[u]
                if all(fed[u] == len(per_utt[u]) for u in range(n_utt)):
                    stream.finish()
                    finished = True
                elif finish_early and it % 7 == 3:
                    # utterances whose script is exhausted end now, the others stay open
                    which = np.array([fed[u] == len(per_utt[u]) for u in range(n_utt)], dtype=np.uint8)
                    stream.finish(which)
            q = int(rng.choice(chunk_choices))
            stream.next_async(q, d_out, stride, d_len)
            ctx.sync()
            lens = np.zeros(n_utt, dtype=np.uint32)
            ctx.d2h(lens, d_len, n_utt * 4)
            assert lens.max(initial=0) <= q
            if lens.max(initial=0):
                buf = np.zeros((n_utt, stride), dtype=np.float32)
                ctx.d2h(buf, d_out, buf.nbytes)
                for u in range(n_utt):
                    rows[u].append(buf[u, :lens[u]].copy())
            elif finished:
                break
        else:
            raise AssertionError("the live stream never ended")
    finally:
        ctx.device_free(d_out)
        ctx.device_free(d_len)
    return [np.concatenate(r) if r else np.zeros(0, dtype=np.float32) for r in rows]


@pytest.mark.parametrize("lanes", [0, 1, 2, 4, 8])
def test_appends_between_pulls_equal_the_one_shot_rendering(gpu_ctx, lanes):
    voices = W.preset_voices(8)
    gpu_ctx.set_voices(voices)
    gpu_ctx.set_option("lanes_per_utterance", lanes)
    rng = np.random.default_rng(100 + lanes)
    n_utt = 37
    per_utt = _script(rng, n_utt, 8)
    vids = (np.arange(n_utt) % 8).astype(np.uint32)
    seeds = (np.arange(n_utt) * 2654435761 % (1 << 32)).astype(np.uint32)
    st = G.LiveStream(gpu_ctx, n_utt, vids, seeds, ring_segments=16)
    try:
        got = _drive(gpu_ctx, st, per_utt, rng, 2048, [1, 17, 64, 333, 1000, 2048], finish_early=(lanes in (0, 4)))
    finally:
        st.close()
        gpu_ctx.set_option("lanes_per_utterance", 0)
    ov = _ovoices(voices)
    total = 0
    for u in range(n_utt):
        ref, n = O.synthesize_phonemes(ov[vids[u]], per_utt[u], int(seeds[u]))
        assert len(got[u]) == n, (u, len(got[u]), n)
        assert np.array_equal(got[u].view(np.uint32), ref.view(np.uint32)), u
        total += n
    assert total > 100000
    gpu_ctx.set_voices(W.single_voice())


@pytest.mark.parametrize("lanes,ring", [(1, 4), (2, 8), (8, 4)])
def test_rings_wrap_around_many_times(gpu_ctx, lanes, ring):
    """Scripts of up to 60 segments through rings of 4 / 8: every slot is reused many times, appends that do not fit are
    refused whole (GRAIL_ERR_BUFFER_TOO_SMALL) and repeated after a pull."""
    voices = W.preset_voices(8)
    gpu_ctx.set_voices(voices)
    gpu_ctx.set_option("lanes_per_utterance", lanes)
    rng = np.random.default_rng(900 + lanes)
    n_utt = 21
    per_utt = _script(rng, n_utt, 8, max_segs=60)
    vids = (np.arange(n_utt) % 8).astype(np.uint32)
    seeds = (np.arange(n_utt) * 7919 + 1).astype(np.uint32)
    st = G.LiveStream(gpu_ctx, n_utt, vids, seeds, ring_segments=ring)
    try:
        got = _drive(gpu_ctx, st, per_utt, rng, 2048, [64, 500, 2048])
    finally:
        st.close()
        gpu_ctx.set_option("lanes_per_utterance", 0)
    ov = _ovoices(voices)
    assert max(len(p) for p in per_utt) > 6 * ring
    for u in range(n_utt):
        ref, n = O.synthesize_phonemes(ov[vids[u]], per_utt[u], int(seeds[u]))
        assert len(got[u]) == n, (u, len(got[u]), n)
        assert np.array_equal(got[u].view(np.uint32), ref.view(np.uint32)), u
    gpu_ctx.set_voices(W.single_voice())


def test_a_starved_stream_pauses_and_a_fresh_one_waits_for_two_segments(gpu_ctx):
    """The reference's Sequencer starts by pulling two segments (src/lib.rs:877-878) and then one per segment end: with
    fewer in the ring an open stream renders nothing (or stops at the segment's last sample) and reports a short row."""
    v = G.voice_generic(48000.0)
    gpu_ctx.set_voices([v])
    f = v.center_frequency
    segs = G.segments([(G.PH_SILENCE, .01, .0078125, f), (G.PH_A, .02, .0078125, f), (G.PH_E, .02, .015625, f), (G.PH_A, .01, .0078125, f)])
    ref, n_ref = O.synthesize_phonemes(O.Voice.from_buffer_copy(bytes(v)), segs, 5)
    d_out = gpu_ctx.device_alloc(8192 * 4)
    d_len = gpu_ctx.device_alloc(4)
    st = G.LiveStream(gpu_ctx, 1, None, [5], ring_segments=4)

    def pull(q):
        st.next_async(q, d_out, 8192, d_len)
        gpu_ctx.sync()
        n = np.zeros(1, dtype=np.uint32)
        gpu_ctx.d2h(n, d_len, 4)
        buf = np.zeros(8192, dtype=np.float32)
        gpu_ctx.d2h(buf, d_out, buf.nbytes)
        return buf[:int(n[0])].copy()

    try:
        got = [pull(500)]
        assert len(got[0]) == 0                               # nothing appended: nothing rendered, nothing consumed
        st.append(segs[:1], [0, 1])
        got.append(pull(500))
        assert len(got[-1]) == 0 and st.pending()[0] == 1     # one segment: the Sequencer wants two
        st.append(segs[1:2], [0, 1])
        got.append(pull(8192))
        first = len(got[-1])
        assert 0 < first < 8192 and st.pending()[0] == 0      # renders segment 0 and stops where it needs the third
        # the f32 Sequencer clock (:861-873): the first step takes it below zero and adds segment 0's length; the
        # stream stops before the step that takes it below zero again
        dt = np.float32(1.0) / np.float32(48000.0)
        clk, want = (np.float32(0.0) - dt) + np.float32(.01), 1
        while clk - dt >= 0:
            clk, want = clk - dt, want + 1
        assert first == want and 478 <= want <= 482
        got.append(pull(100))
        assert len(got[-1]) == 0                              # still starved: the same step is taken again later
        # a ring of four holds the two segments in use and two pending: a third does not fit
        st.append(segs[2:4], [0, 2])
        with pytest.raises(G.GrailError) as ei:
            st.append(segs[3:4], [0, 1])
        assert ei.value.status == G.ERR_BUFFER_TOO_SMALL
        got.append(pull(700))
        got.append(pull(8192))                                # ... pauses again at the end of what it has
        st.finish()
        got.append(pull(8192))                                # the last segment fades out and the row ends
        got.append(pull(8192))
        assert len(got[-1]) == 0
        with pytest.raises(G.GrailError):
            st.append(segs[:1], [0, 1])                       # finished: nothing can be appended
    finally:
        st.close()
        gpu_ctx.device_free(d_out)
        gpu_ctx.device_free(d_len)
    out = np.concatenate(got)
    assert len(out) == n_ref
    assert np.array_equal(out.view(np.uint32), ref.view(np.uint32))


@pytest.mark.parametrize("lanes", [1, 2])
def test_live_stream_of_caller_built_elems(gpu_ctx, lanes):
    """SequenceElems (no Selector): the elems travel through the ring with their segments; formants fall silent and
    wake up again across appends."""
    rng = np.random.default_rng(77 + lanes)
    lo, all8 = [1, 1, 1, 1, 0, 0, 0, 0], [1] * 8
    n_utt = 9
    g_per, o_per = [], []
    for u in range(n_utt):
        gs, os_ = [], []
        for i in range(int(rng.integers(2, 20))):
            mask = [lo, all8, None][int(rng.integers(0, 3))]
            has = mask is not None
            e = _elem(rng, mask if has else all8)
            ln = float(rng.uniform(0.004, 0.012))
            gs.append(G.SequenceElem(int(has), G.SynthesisElem.from_np(e), ln, 0.0078125))
            os_.append(O.SequenceElem(int(has), O.SynthesisElem.from_buffer_copy(e.tobytes()), ln, 0.0078125))
        g_per.append(gs)
        o_per.append(os_)
    v = G.voice_generic(48000.0)
    gpu_ctx.set_voices([v])
    gpu_ctx.set_option("lanes_per_utterance", lanes)
    seeds = (np.arange(n_utt) * 17 + 3).astype(np.uint32)
    st = G.LiveStream(gpu_ctx, n_utt, None, seeds, ring_segments=4, elems=True)      # (up to 19 segments: the rings wrap)
    d_out = gpu_ctx.device_alloc(n_utt * 1024 * 4)
    d_len = gpu_ctx.device_alloc(n_utt * 4)
    rows = [[] for _ in range(n_utt)]
    fed = [0] * n_utt
    try:
        finished = False
        for it in range(10000):
            if not finished:
                segs, offs = [], [0]
                for u in range(n_utt):
                    k = min(int(rng.integers(0, 3)), len(g_per[u]) - fed[u])
                    segs += g_per[u][fed[u]:fed[u] + k]
                    fed[u] += k
                    offs.append(len(segs))
                if segs:
                    try:
                        st.append(segs, offs)
                    except G.GrailError as e:        # a ring is full: nothing was appended, feed again after a pull
                        assert e.status == G.ERR_BUFFER_TOO_SMALL
                        for u in range(n_utt):
                            fed[u] -= offs[u + 1] - offs[u]
                if all(fed[u] == len(g_per[u]) for u in range(n_utt)):
                    st.finish()
                    finished = True
            q = int(rng.choice([64, 257, 1000]))
            st.next_async(q, d_out, 1024, d_len)
            gpu_ctx.sync()
            lens = np.zeros(n_utt, dtype=np.uint32)
            gpu_ctx.d2h(lens, d_len, n_utt * 4)
            buf = np.zeros((n_utt, 1024), dtype=np.float32)
            gpu_ctx.d2h(buf, d_out, buf.nbytes)
            for u in range(n_utt):
                rows[u].append(buf[u, :lens[u]].copy())
            if finished and lens.max() == 0:
                break
    finally:
        st.close()
        gpu_ctx.device_free(d_out)
        gpu_ctx.device_free(d_len)
        gpu_ctx.set_option("lanes_per_utterance", 0)
    ov = O.Voice.from_buffer_copy(bytes(v))
    for u in range(n_utt):
        ref = O.synthesize_sequence(ov, o_per[u], int(seeds[u]))
        got = np.concatenate(rows[u])
        assert len(got) == len(ref), u
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), u


@pytest.mark.parametrize("lanes", [0, 4])
def test_live_stream_in_fast_arithmetic(gpu_ctx, lanes):
    """Fast mode: the chain state is carried exactly (same lengths, same pauses), samples within the tolerance."""
    voices = W.single_voice()
    gpu_ctx.set_voices(voices)
    gpu_ctx.set_option("lanes_per_utterance", lanes)
    gpu_ctx.set_option("arithmetic", 1)
    rng = np.random.default_rng(5)
    n_utt = 20
    per_utt = []
    for u in range(n_utt):
        k = int(rng.integers(2, 6))
        ph = rng.choice([G.PH_A, G.PH_E, G.PH_SILENCE], size=k)
        hz = (rng.uniform(100.0, 200.0, size=k) / 48000.0).astype(np.float32)
        per_utt.append(G.segments([(int(ph[i]), 0.125, 0.125, float(hz[i])) for i in range(k)]))
    seeds = np.arange(n_utt, dtype=np.uint32) + 9
    st = G.LiveStream(gpu_ctx, n_utt, None, seeds)
    try:
        got = _drive(gpu_ctx, st, per_utt, rng, 4096, [480, 960, 4096])
    finally:
        st.close()
        gpu_ctx.set_option("arithmetic", 0)
        gpu_ctx.set_option("lanes_per_utterance", 0)
    ov = _ovoices(voices)[0]
    worst = 0.0
    for u in range(n_utt):
        ref, n = O.synthesize_phonemes(ov, per_utt[u], int(seeds[u]))
        assert len(got[u]) == n, u
        worst = max(worst, float(np.max(np.abs(got[u].astype(np.float64) - ref.astype(np.float64)))))
    print(f"live stream, fast arithmetic, lanes={lanes}: max |d| = {worst / ULP:.1f} * 2^-23")
    # (a few live streams on the library's own mapping take the pipelined exact workgroups: the reference's bits)
    assert worst <= G.FAST_TOLERANCE and (worst > 0.0 or lanes == 0)


def test_live_stream_argument_checks(gpu_ctx):
    gpu_ctx.set_voices(W.single_voice())
    with pytest.raises(G.GrailError):
        G.LiveStream(gpu_ctx, 0)
    with pytest.raises(G.GrailError):
        G.LiveStream(gpu_ctx, 1, ring_segments=6)
    st = G.LiveStream(gpu_ctx, 2)
    try:
        with pytest.raises(G.GrailError):                      # SequenceElems into a PhonemeElem stream
            G._check(G.load().grail_stream_append_elems(gpu_ctx.handle, st.handle, None, np.zeros(3, dtype=np.uint32).ctypes.data))
        with pytest.raises(G.GrailError):                      # offsets must be non-decreasing
            st.append(G.segments([(G.PH_A, .01, .01, .002)]), [0, 1, 0])
        with pytest.raises(G.GrailError):                      # a phoneme discriminant out of range
            st.append(G.segments([(9, .01, .01, .002)]), [0, 1, 1])
        assert list(st.pending()) == [0, 0]
    finally:
        st.close()
    b = gpu_ctx.upload(G.segments([(G.PH_A, .01, .01, .002)]), [0, 1])
    plain = G.Stream(b)
    try:
        with pytest.raises(G.GrailError):                      # not a live stream
            G._check(G.load().grail_stream_finish(gpu_ctx.handle, plain.handle, None))
    finally:
        plain.close()
        b.free()


def test_interactive_example_is_one_chain_for_the_whole_session(gpu_ctx, tmp_path):
    """examples/grail_interactive.cpp on grail::LiveStream: lines arriving at different times, Silence fed whenever the
    Sequencer asks and nothing is waiting ('  ' in the reference, examples/interactive.rs:31).  Its output must be the
    oracle's rendering of the SAME phoneme list in one piece — phase, noise, jitter and filters carried across lines —
    and the list must be what the reference's lazy source yields for this timing."""
    import os
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "grail-rs_amd", "lib", "grail_interactive")
    assert os.path.exists(exe)
    script = "ae\n@2.2 oui a\n@2.3 e\n"
    r = subprocess.run([exe, "441", "0.7"], input=script.encode(), capture_output=True, timeout=300)
    assert r.returncode == 0, r.stderr.decode()
    got = np.frombuffer(r.stdout, dtype="<f4")
    names = {"Silence": G.PH_SILENCE, "Stop": G.PH_STOP, "Glide": G.PH_GLIDE, "A": G.PH_A, "E": G.PH_E}
    fed = [(names[m.group(1)], int(m.group(2))) for m in re.finditer(r"fed (\w+) at (\d+)", r.stderr.decode())]
    phon = [p for p, _ in fed]
    # what the reference's source delivers: the leading Silence of .transcribe() (src/lib.rs:1201), "ae" + ' ', Silences while
    # nothing is waiting, then "oui a" + ' ' and "e" + ' ' back to back, Silences until the session is closed
    rules, cs = G.language_generic()
    first = G.transcribe("ae ", rules, cs, leading_silence=True)
    later = G.transcribe("oui a ", rules, cs) + G.transcribe("e ", rules, cs)
    assert phon[:len(first)] == first == [G.PH_SILENCE, G.PH_A, G.PH_E, G.PH_SILENCE]
    k = len(first)
    while phon[k] == G.PH_SILENCE:
        k += 1
    assert k > len(first)                                    # at least one Silence was fed while nothing was waiting
    assert phon[k:k + len(later)] == later
    assert all(p == G.PH_SILENCE for p in phon[k + len(later):])
    # a line that arrives at 2.2 s is first fed at a pull after 2.2 s of audio (pulls happen when a segment ends)
    assert fed[k][1] >= int(2.2 * 44100) - 441
    # one chain: the oracle renders the whole list in one piece (Intonator constants, src/lib.rs:1068-1073)
    ov = O.voice_generic()
    segs = G.segments([(p, 0.5, 0.5, ov.center_frequency) for p in phon])
    ref, n = O.synthesize_phonemes(ov, segs, 0)
    assert len(got) == n and n > 5 * 44100 // 2
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    # and NOT what separate chains per line would give: the second line restarted from phase 0 / seed 0 differs
    alone, _ = O.synthesize_phonemes(ov, G.segments([(p, 0.5, 0.5, ov.center_frequency) for p in [G.PH_SILENCE] + later]), 0)
    at = fed[k][1]
    assert not np.array_equal(got[at:at + 20000].view(np.uint32), alone[:20000].view(np.uint32))
