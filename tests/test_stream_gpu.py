"""Resumable synthesis: chunks pulled through grail_stream_* must
concatenate to exactly the one-shot rendering — for any chunk sizes, any lane mapping, ragged
and edge-case utterances included."""
import numpy as np
import pytest

import grail_hip as G
from grail_hip import workload as W
from test_parity_gpu import edge_case_batch

pytestmark = pytest.mark.gpu


def stream_all(ctx, batch, n_utt, chunk_sizes, stride):
    st = G.Stream(batch)
    d_out = ctx.device_alloc(n_utt * stride * 4)
    d_len = ctx.device_alloc(n_utt * 4)
    rows = [[] for _ in range(n_utt)]
    try:
        k = 0
        while True:
            q = chunk_sizes[k % len(chunk_sizes)]
            k += 1
            st.next_async(q, d_out, stride, d_len)
            ctx.sync()
            lens = np.zeros(n_utt, dtype=np.uint32)
            ctx.d2h(lens, d_len, n_utt * 4)
            assert lens.max(initial=0) <= q
            if lens.max(initial=0) == 0:
                break
            buf = np.zeros((n_utt, stride), dtype=np.float32)
            ctx.d2h(buf, d_out, buf.nbytes)
            for u in range(n_utt):
                rows[u].append(buf[u, :lens[u]].copy())
            assert k < 10000
    finally:
        st.close()
        ctx.device_free(d_out)
        ctx.device_free(d_len)
    return [np.concatenate(r) if r else np.zeros(0, dtype=np.float32) for r in rows]


@pytest.mark.parametrize("lanes", [0, 1, 2, 4, 8])
def test_chunked_stream_equals_one_shot(gpu_ctx, lanes):
    voices = W.preset_voices(8)
    gpu_ctx.set_voices(voices)
    gpu_ctx.set_option("lanes_per_utterance", lanes)
    n_utt = 70
    segs, offs, vids, seeds = W.make_batch(n_utt, n_voices=8, length=0.03, blend_length=0.03)
    full, full_len = gpu_ctx.synthesize(segs, offs, vids, seeds, out_stride=W.max_samples(length=0.03))
    b = gpu_ctx.upload(segs, offs, vids, seeds)
    try:
        got = stream_all(gpu_ctx, b, n_utt, [1000, 37, 2048, 64, 1], stride=2048)
    finally:
        b.free()
        gpu_ctx.set_option("lanes_per_utterance", 0)
    for u in range(n_utt):
        assert len(got[u]) == full_len[u], (u, len(got[u]), full_len[u])
        assert np.array_equal(got[u].view(np.uint32), full[u, :full_len[u]].view(np.uint32)), u


@pytest.mark.parametrize("n_voices,n_utt,kernel", [(1, 300, "NFA=4,PIPE,R32"), (1, 5000, "NFA=4,PIPE,R16"),
                                                    (8, 300, "NFA=8,PIPE,R32"), (8, 3000, "NFA=8,PIPE,R16")])
def test_mid_size_streams_take_the_pipelined_workgroups(gpu_ctx, n_voices, n_utt, kernel):
    """A few hundred to a few thousand streams leave most SIMDs idle on the lane kernels; they run the resumable form of
    the four-wave pipelined workgroups (synth_kernel<..., STREAM, PIPE>: every wave loads the shared utterances' state, the
    rendering wave saves it).  Chunks of any size — a pause inside a calm run, inside an event tile, at a tile's first
    sample — concatenate to the oracle's rendering, bit for bit, and a speech-like script (events at every lane's own
    times) does too."""
    import oracle_lib as O
    voices = W.single_voice() if n_voices == 1 else W.preset_voices(8)
    gpu_ctx.set_voices(voices)
    for corpus in ("aligned", "speech-like"):
        if corpus == "aligned":
            segs, offs, vids, seeds = W.make_batch(n_utt, n_voices=n_voices, length=0.03, blend_length=0.03125)
            total = W.max_samples(length=0.03)
        else:
            segs, offs, vids, seeds, total = W.speech_like_batch(n_utt, np.random.default_rng(n_utt), n_voices=n_voices, scale=0.05)
        b = gpu_ctx.upload(segs, offs, vids, seeds)
        try:
            got = stream_all(gpu_ctx, b, n_utt, [1000, 37, 2048, 64, 1, 129], stride=2048)
            name = gpu_ctx.last_kernel_name()
        finally:
            b.free()
        assert "STREAM" in name and kernel in name, name
        pick = sorted(set([0, 1, 15, 16, n_utt - 1] + [int(u) for u in np.random.default_rng(1).integers(0, n_utt, 40)]))
        sub = np.concatenate([segs[offs[u]:offs[u + 1]] for u in pick])
        sub_offs = np.zeros(len(pick) + 1, dtype=np.uint32)
        sub_offs[1:] = np.cumsum([offs[u + 1] - offs[u] for u in pick])
        ref, ref_len = O.synthesize_batch([O.Voice.from_buffer_copy(bytes(v)) for v in voices], sub, sub_offs, vids[pick], seeds[pick], total)
        for k, u in enumerate(pick):
            assert len(got[u]) == ref_len[k], (corpus, u)
            assert np.array_equal(got[u].view(np.uint32), ref[k, :ref_len[k]].view(np.uint32)), (corpus, u)


@pytest.mark.perf
def test_a_lone_stream_runs_at_the_pace_of_a_full_workgroup(gpu_ctx):
    """One stream in a pipelined workgroup laid out for sixteen: the fifteen slots without an utterance count as finished
    — they ride along in calm tiles like ended utterances — instead of keeping their wave out of every calm tile (a pull of
    100 ms took 1.67 ms for one stream and 0.40 ms for 256: now 0.37 / 0.38).  Kernel times, best of five pulls."""
    gpu_ctx.set_voices(W.single_voice())
    ms = {}
    for n in (1, 256):
        segs, offs, vids, seeds = W.make_batch(n)
        b = gpu_ctx.upload(segs, offs, vids, seeds)
        st = G.Stream(b)
        d_out = gpu_ctx.device_alloc(n * 4800 * 4)
        d_len = gpu_ctx.device_alloc(n * 4)
        try:
            t = []
            for _ in range(6):
                st.next_async(4800, d_out, 4800, d_len)
                gpu_ctx.sync()
                t.append(gpu_ctx.last_kernel_ms())
            assert "PIPE" in gpu_ctx.last_kernel_name(), gpu_ctx.last_kernel_name()
            ms[n] = min(t[1:])
        finally:
            st.close()
            gpu_ctx.device_free(d_out)
            gpu_ctx.device_free(d_len)
            b.free()
    if ms[1] > 1.5 * ms[256]:
        from conftest import skip_if_clocks_unstable
        skip_if_clocks_unstable(gpu_ctx, f"a lone stream took {ms[1]:.3f} ms against {ms[256]:.3f} for 256")
    assert ms[1] <= 1.5 * ms[256], ms


@pytest.mark.parametrize("lanes", [1, 8])
def test_stream_edge_cases_and_ragged_ends(gpu_ctx, lanes):
    gpu_ctx.set_voices(W.single_voice())
    gpu_ctx.set_option("lanes_per_utterance", lanes)
    segs, offs = edge_case_batch(48000.0)
    n_utt = len(offs) - 1
    seeds = np.arange(n_utt, dtype=np.uint32) * 977
    with np.errstate(all="ignore"):
        full, full_len = gpu_ctx.synthesize(segs, offs, None, seeds, out_stride=20032)
        b = gpu_ctx.upload(segs, offs, None, seeds)
        try:
            got = stream_all(gpu_ctx, b, n_utt, [333, 4096], stride=4096)
        finally:
            b.free()
            gpu_ctx.set_option("lanes_per_utterance", 0)
    for u in range(n_utt):
        assert len(got[u]) == full_len[u], (u, len(got[u]), full_len[u])
        assert np.array_equal(got[u].view(np.uint32), full[u, :full_len[u]].view(np.uint32)), u


def test_stream_silent_formants_wake_up(gpu_ctx):
    """The resumable kernel keeps the silent formants' low-pass state and may skip only their
    band-pass: utterances whose upper formants fall silent and wake up again, streamed in
    chunks, against the one-shot rendering and the oracle."""
    import oracle_lib as O
    from test_parity_gpu import _elem
    rng = np.random.default_rng(21)
    lo, all8 = [1, 1, 1, 1, 0, 0, 0, 0], [1] * 8
    plans = [[lo, lo, lo], [lo, all8, lo], [all8, lo, lo], [lo, lo, all8, lo], [lo, None, lo, all8]]
    gsegs, osegs, offs = [], [], [0]
    for plan in plans * 4:
        for mask in plan:
            has = mask is not None
            e = _elem(rng, mask if has else all8)
            ln = float(rng.uniform(0.004, 0.012))
            gsegs.append(G.SequenceElem(int(has), G.SynthesisElem.from_np(e), ln, 0.0078125))
            osegs.append(O.SequenceElem(int(has), O.SynthesisElem.from_buffer_copy(e.tobytes()), ln, 0.0078125))
        offs.append(len(gsegs))
    n = len(offs) - 1
    seeds = np.arange(n, dtype=np.uint32) * 17 + 3
    v = G.voice_generic(48000.0)
    gpu_ctx.set_voices([v])
    ov = O.Voice.from_buffer_copy(bytes(v))
    for lanes in (1, 2):
        gpu_ctx.set_option("lanes_per_utterance", lanes)
        b = gpu_ctx.upload_elems(gsegs, offs, None, seeds)
        try:
            got = stream_all(gpu_ctx, b, n, [257, 64, 1000], stride=1024)
        finally:
            b.free()
            gpu_ctx.set_option("lanes_per_utterance", 0)
        for u in range(n):
            ref = O.synthesize_sequence(ov, osegs[offs[u]:offs[u + 1]], int(seeds[u]))
            assert len(got[u]) == len(ref), (lanes, u)
            assert np.array_equal(got[u].view(np.uint32), ref.view(np.uint32)), (lanes, u)


@pytest.mark.parametrize("lanes", [0, 1, 8])
def test_stream_pcm16_chunks_mixed_with_f32(gpu_ctx, lanes):
    """grail_stream_next_pcm16_async: chunks come out as i16 PCM (examples/cli.rs:49 fused into the
    store); f32 and i16 calls alternate on one stream and together give the one-shot rendering."""
    import oracle_lib as O
    voices = W.preset_voices(8)
    gpu_ctx.set_voices(voices)
    gpu_ctx.set_option("lanes_per_utterance", lanes)
    n_utt = 66
    segs, offs, vids, seeds = W.make_batch(n_utt, n_voices=8, length=0.02, blend_length=0.02)
    full, full_len = gpu_ctx.synthesize(segs, offs, vids, seeds, out_stride=W.max_samples(length=0.02))
    L = O.lib()
    b = gpu_ctx.upload(segs, offs, vids, seeds)
    st = G.Stream(b)
    stride, q = 512, 500
    d_f = gpu_ctx.device_alloc(n_utt * stride * 4)
    d_i = gpu_ctx.device_alloc(n_utt * stride * 2)
    d_len = gpu_ctx.device_alloc(n_utt * 4)
    pos = np.zeros(n_utt, dtype=np.int64)
    try:
        for k in range(40):
            as_i16 = k % 2 == 1
            if as_i16:
                st.next_pcm16_async(q, d_i, stride, d_len)
            else:
                st.next_async(q, d_f, stride, d_len)
            gpu_ctx.sync()
            lens = np.zeros(n_utt, dtype=np.uint32)
            gpu_ctx.d2h(lens, d_len, lens.nbytes)
            if lens.max() == 0:
                break
            buf = np.zeros((n_utt, stride), dtype=np.int16 if as_i16 else np.float32)
            gpu_ctx.d2h(buf, d_i if as_i16 else d_f, buf.nbytes)
            for u in range(n_utt):
                want = full[u, pos[u]:pos[u] + lens[u]]
                if as_i16:
                    want = np.array([L.orc_pcm16(float(v)) for v in want], dtype=np.int16)
                    assert np.array_equal(buf[u, :lens[u]], want), (k, u)
                else:
                    assert np.array_equal(buf[u, :lens[u]].view(np.uint32), want.view(np.uint32)), (k, u)
                pos[u] += lens[u]
        assert np.array_equal(pos, full_len.astype(np.int64))
    finally:
        st.close()
        b.free()
        for d in (d_f, d_i, d_len):
            gpu_ctx.device_free(d)
        gpu_ctx.set_option("lanes_per_utterance", 0)


@pytest.mark.parametrize("lanes", [1, 2, 4])
@pytest.mark.parametrize("arithmetic", [0, 1, 2])
def test_lean_stream_kernels_take_any_blend_length(gpu_ctx, lanes, arithmetic):
    """Streams of voices::generic() (four live formants) with blend lengths that are not powers of two: the lean
    resumable kernels — four formants laid out — in exact arithmetic (chunks concatenate to the oracle's bits), in fast
    arithmetic and in its second tier (within the tolerance of the oracle; L = 1 only for the second tier, the exact
    lean kernels otherwise)."""
    import oracle_lib as O
    rng = np.random.default_rng(17 + lanes)
    voices = W.single_voice()
    gpu_ctx.set_voices(voices)
    n_utt = 40
    segs, offs, vids, seeds = W.make_batch(n_utt, length=0.03)
    segs["blend_length"] = rng.choice([0.3, 0.013, 1.0 / 3.0, 0.007, 0.0625], len(segs)).astype(np.float32)
    stride = 2048
    ov = [O.Voice.from_buffer_copy(bytes(v)) for v in voices]
    ref, ref_len = O.synthesize_batch(ov, segs, offs, vids, seeds, W.max_samples(length=0.03))
    gpu_ctx.set_option("lanes_per_utterance", lanes)
    gpu_ctx.set_option("arithmetic", arithmetic)
    b = gpu_ctx.upload(segs, offs, vids, seeds)
    try:
        got = stream_all(gpu_ctx, b, n_utt, [700, 64, 33, 2048], stride)
        name = gpu_ctx.last_kernel_name()
    finally:
        b.free()
        gpu_ctx.set_option("arithmetic", 0)
        gpu_ctx.set_option("lanes_per_utterance", 0)
    assert "STREAM" in name and "ANYBL" in name and "NFA=4" in name, name
    assert ("MID" in name) == (arithmetic == 2 and lanes == 1), name
    for u in range(n_utt):
        n = int(ref_len[u])
        assert len(got[u]) == n, (u, name)
        if arithmetic == 0 or (arithmetic == 2 and lanes > 1):
            assert np.array_equal(got[u].view(np.uint32), ref[u, :n].view(np.uint32)), (u, name)
        else:
            peak = max(1.0, float(np.abs(ref[u, :n]).max()))
            assert float(np.abs(got[u].astype(np.float64) - ref[u, :n]).max()) <= G.FAST_TOLERANCE * peak, (u, name)
