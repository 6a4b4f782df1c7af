"""Threading contract of the boundary: the reference chain is `Send` — examples/interactive.rs:42-48
moves it into the audio callback thread — and has no globals; here a `grail_ctx` is usable from one thread at a
time, distinct contexts are independent, and a stream may be opened on one thread and pulled on another."""
import threading

import numpy as np
import pytest

import grail_hip as G
import oracle_lib as O
from grail_hip import workload as W

pytestmark = pytest.mark.gpu


def _ov(voices):
    return [O.Voice.from_buffer_copy(bytes(v)) for v in voices]


def _same_bits(out, out_len, ref, ref_len, what):
    assert np.array_equal(out_len, ref_len), what
    for u in range(len(ref_len)):
        n = int(ref_len[u])
        assert np.array_equal(out[u, :n].view(np.uint32), ref[u, :n].view(np.uint32)), (what, u)


def test_two_contexts_driven_from_two_threads(built):
    """Two contexts on device 0, each on a thread of its own, different voice tables, different batches,
    different lane mappings and arithmetic modes, many launches each, all in flight together: every
    result is the oracle's (exact) or within the tolerance (fast)."""
    jobs = []
    for i, (n_voices, lanes, n_utt, first) in enumerate([(1, 0, 130, 0), (8, 2, 97, 1000)]):
        voices = W.single_voice() if n_voices == 1 else W.preset_voices(8)
        segs, offs, vids, seeds = W.make_batch(n_utt, first_utt=first, n_voices=n_voices, length=0.03, blend_length=0.03125)
        stride = W.max_samples(length=0.03)
        ref, ref_len = O.synthesize_batch(_ov(voices), segs, offs, vids, seeds, stride)
        jobs.append(dict(voices=voices, lanes=lanes, args=(segs, offs, vids, seeds), stride=stride, ref=ref, ref_len=ref_len))
    errors = []
    start = threading.Barrier(2)

    def drive(job):
        try:
            with G.Context(0) as ctx:
                ctx.set_voices(job["voices"])
                ctx.set_option("lanes_per_utterance", job["lanes"])
                start.wait()
                for it in range(12):
                    ctx.set_option("arithmetic", it & 1)
                    out, out_len = ctx.synthesize(*job["args"], out_stride=job["stride"])
                    if it & 1:
                        assert np.array_equal(out_len, job["ref_len"])
                        worst = max(float(np.max(np.abs(out[u, :n].astype(np.float64) - job["ref"][u, :n])))
                                    for u, n in enumerate(map(int, job["ref_len"])) if n)
                        assert worst <= G.FAST_TOLERANCE
                    else:
                        _same_bits(out, out_len, job["ref"], job["ref_len"], f"iteration {it}")
        except BaseException as e:          # noqa: BLE001 — reported by the main thread
            errors.append(e)
            try:
                start.abort()
            except Exception:               # noqa: BLE001
                pass

    threads = [threading.Thread(target=drive, args=(j,)) for j in jobs]
    for t in threads:
        t.start()
    for t in threads:
        t.join(300)
    assert not errors, errors


def test_a_stream_opened_on_one_thread_is_pulled_on_another(built):
    """examples/interactive.rs builds the chain on the main thread and moves it into the audio thread."""
    voices = W.single_voice()
    segs, offs, vids, seeds = W.make_batch(70, length=0.05, blend_length=0.0625)
    stride = W.max_samples(length=0.05)
    ref, ref_len = O.synthesize_batch(_ov(voices), segs, offs, vids, seeds, stride)
    ctx = G.Context(0)
    ctx.set_voices(voices)
    batch = ctx.upload(segs, offs, vids, seeds)
    stream = G.Stream(batch)                      # opened here ...
    chunk = 1000
    d_out = ctx.device_alloc(70 * chunk * 4)
    d_len = ctx.device_alloc(70 * 4)
    got = [[] for _ in range(70)]
    errors = []

    def audio_thread():                           # ... pulled there
        try:
            while True:
                stream.next_async(chunk, d_out, chunk, d_len)
                ctx.sync()
                lens = np.zeros(70, dtype=np.uint32)
                ctx.d2h(lens, d_len, lens.nbytes)
                rows = np.zeros((70, chunk), dtype=np.float32)
                ctx.d2h(rows, d_out, rows.nbytes)
                for u in range(70):
                    got[u].append(rows[u, :lens[u]].copy())
                if not lens.any():
                    break
        except BaseException as e:          # noqa: BLE001
            errors.append(e)

    t = threading.Thread(target=audio_thread)
    t.start()
    t.join(300)
    assert not errors, errors
    for u in range(70):
        row = np.concatenate(got[u])
        assert len(row) == ref_len[u]
        assert np.array_equal(row.view(np.uint32), ref[u, :len(row)].view(np.uint32)), u
    stream.close()
    ctx.device_free(d_out)
    ctx.device_free(d_len)
    batch.free()
    ctx.close()


def test_eight_contexts_in_one_process(built):
    """The single-process shape of SURVEY §8e (one context per GPU; here eight on the one GPU of the test
    box), every context driven by its own thread, each rendering its shard of one corpus: the rows are those
    of the corpus rendered in one piece."""
    voices = W.preset_voices(8)
    n_utt, world = 8 * 24, 8
    stride = W.max_samples(length=0.03)
    segs, offs, vids, seeds = W.make_batch(n_utt, n_voices=8, length=0.03, blend_length=0.03125)
    ref, ref_len = O.synthesize_batch(_ov(voices), segs, offs, vids, seeds, stride)
    ctxs = [G.Context(0) for _ in range(world)]
    out = [None] * world
    errors = []

    def drive(r):
        try:
            ctxs[r].set_voices(voices)
            first, last, s, o, v, j = W.shard_inputs(n_utt // world, r, world, 8, length=0.03, blend_length=0.03125)
            for _ in range(3):
                out[r] = (first, last) + ctxs[r].synthesize(s, o, v, j, out_stride=stride)
        except BaseException as e:          # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=drive, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(300)
    assert not errors, errors
    for r in range(world):
        first, last, rows, lens = out[r]
        _same_bits(rows, lens, ref[first:last], ref_len[first:last], f"context {r}")
    for c in ctxs:
        c.close()
