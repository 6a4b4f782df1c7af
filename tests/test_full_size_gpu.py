"""The launch planner's work at FULL size on the real device (no "assume_compute_units"): a batch larger than one round
of the one-lane kernels is cut into blocks (launch_plan.cpp plan_blocks), a speech-like corpus is laid out by the rows'
lengths and events (ragged_plan).  Exact arithmetic is mapping-invariant, so whatever the plan every row must carry the
same bits — checked for all rows through on-device digests, against the oracle on rows spread over the batch — and the
same batches in tolerance arithmetic stay within GRAIL_FAST_TOLERANCE of the exact rows, every sample (on-device compare).
tests/test_composite_gpu.py checks the same machinery sample by sample on a device made small."""
import numpy as np
import pytest

import grail_hip as G
import oracle_lib as O
from grail_hip import workload as W

pytestmark = pytest.mark.gpu


def _ovoices(voices):
    return [O.Voice.from_buffer_copy(bytes(v)) for v in voices]


def _oracle_digests(voices, segs, offs, vids, seeds, pick, stride):
    """bit-pattern sums (the device digest's definition) and lengths of the oracle's rendering of the rows `pick`"""
    sub = np.concatenate([segs[offs[u]:offs[u + 1]] for u in pick])
    sub_offs = np.zeros(len(pick) + 1, dtype=np.uint32)
    sub_offs[1:] = np.cumsum([offs[u + 1] - offs[u] for u in pick])
    ref, ref_len = O.synthesize_batch(_ovoices(voices), sub, sub_offs, vids[pick], seeds[pick], stride)
    return [int(ref[k, :ref_len[k]].view(np.uint32).astype(np.uint64).sum()) for k in range(len(pick))], ref_len


def _render(ctx, batch, d_out, stride, d_len, n):
    batch.synthesize_async(d_out, stride, d_len)
    ctx.sync()
    lens = np.zeros(n, dtype=np.uint32)
    ctx.d2h(lens, d_len, n * 4)
    sums, maxabs, bad = ctx.digest(d_out, stride, d_len, n)
    return lens, sums, maxabs, bad


def test_a_batch_larger_than_one_round_is_cut_into_blocks_at_full_size(gpu_ctx):
    """70 000 aligned utterances x 2 s, voices::generic(): the whole device holds 65 536 on one lane each, the rest goes to
    another kernel family in the same call.  Digests of the composite launch == digests of the one-launch rendering for all
    70 000 rows, 64 rows spread over both blocks == the oracle, the fast rendering within the tolerance everywhere."""
    n = 70000
    voices = W.single_voice()
    gpu_ctx.set_voices(voices)
    segs, offs, vids, seeds = W.make_batch(n)
    stride = W.max_samples()
    b = gpu_ctx.upload(segs, offs, vids, seeds)
    d_out = gpu_ctx.device_alloc(n * stride * 4)
    d_fast = gpu_ctx.device_alloc(n * stride * 4)
    d_len = gpu_ctx.device_alloc(n * 4)
    d_len_fast = gpu_ctx.device_alloc(n * 4)
    try:
        gpu_ctx.set_option("composite_launches", 0)
        lens1, sums1, _, bad1 = _render(gpu_ctx, b, d_out, stride, d_len, n)
        assert gpu_ctx.get_option("last_launch_blocks") == 1
        gpu_ctx.set_option("composite_launches", 1)
        lens, sums, maxabs, bad = _render(gpu_ctx, b, d_out, stride, d_len, n)
        assert gpu_ctx.get_option("last_launch_blocks") >= 2, "70 000 utterances are more than one round of the one-lane kernel"
        assert np.all(lens == 96006) and np.array_equal(lens, lens1)
        assert bad.sum() == 0 and bad1.sum() == 0 and maxabs.max() <= 1.0
        assert np.array_equal(sums, sums1)                     # 70 000 checksums: the cut changes no bit
        pick = sorted(set([0, 1, 63, 64, 65535, 65536, 65537, n - 1] +
                          [int(u) for u in np.random.default_rng(3).integers(0, n, 44)] +
                          [int(u) for u in np.random.default_rng(4).integers(65536, n, 12)]))
        want, want_len = _oracle_digests(voices, segs, offs, vids, seeds, pick, stride)
        for k, u in enumerate(pick):
            assert lens[u] == want_len[k] and int(sums[u]) == want[k], u
        # the same batch in tolerance arithmetic (composite too: 65 536 + the rest)
        gpu_ctx.set_option("arithmetic", 1)
        b.synthesize_async(d_fast, stride, d_len_fast)
        gpu_ctx.sync()
        assert gpu_ctx.get_option("last_launch_fast") == 1
        maxdiff, _, structural = gpu_ctx.compare(d_out, d_fast, stride, d_len, d_len_fast, n)
        assert structural.sum() == 0                           # same lengths, nothing non-finite
        assert maxdiff.max() <= G.FAST_TOLERANCE, maxdiff.max() * 2.0 ** 23
    finally:
        gpu_ctx.set_option("arithmetic", 0)
        gpu_ctx.set_option("composite_launches", 1)
        for p in (d_out, d_fast, d_len, d_len_fast):
            gpu_ctx.device_free(p)
        b.free()


@pytest.mark.parametrize("n_voices", [1, 8])
def test_the_speech_like_corpus_at_full_size(gpu_ctx, n_voices):
    """65 536 utterances of 8 - 32 phonemes of 40 - 160 ms (workload.speech_like_batch), 0.5 - 3.8 s: option "ragged_plan"
    on and off give the same 65 536 digests in exact arithmetic, 48 rows spread over the batch == the oracle, and the fast
    rendering (whatever mapping the planner takes for it) stays within the tolerance of the exact rows, every sample."""
    n = 65536
    voices = W.single_voice() if n_voices == 1 else W.preset_voices(8)
    gpu_ctx.set_voices(voices)
    segs, offs, vids, seeds, _ = W.speech_like_batch(n, np.random.default_rng(7), n_voices=n_voices)
    b = gpu_ctx.upload(segs, offs, vids, seeds)
    stride = (int(b.lengths().max()) + 64 + 63) // 64 * 64
    d_out = gpu_ctx.device_alloc(n * stride * 4)
    d_fast = gpu_ctx.device_alloc(n * stride * 4)
    d_len = gpu_ctx.device_alloc(n * 4)
    d_len_fast = gpu_ctx.device_alloc(n * 4)
    try:
        gpu_ctx.set_option("ragged_plan", 0)
        lens0, sums0, _, bad0 = _render(gpu_ctx, b, d_out, stride, d_len, n)
        lanes0 = gpu_ctx.get_option("last_launch_lanes")
        gpu_ctx.set_option("ragged_plan", 1)
        lens, sums, maxabs, bad = _render(gpu_ctx, b, d_out, stride, d_len, n)
        assert gpu_ctx.get_option("last_launch_lanes") != lanes0, "the ragged plan lays this corpus out on a wider mapping"
        assert np.array_equal(lens, lens0) and np.array_equal(sums, sums0)
        assert bad.sum() == 0 and bad0.sum() == 0 and maxabs.max() <= 1.0
        assert lens.min() > 0.4 * 48000 and lens.max() > 3.0 * 48000
        pick = sorted(set([0, 1, 63, 64, n - 1, int(np.argmax(lens)), int(np.argmin(lens))] +
                          [int(u) for u in np.random.default_rng(9).integers(0, n, 41)]))
        want, want_len = _oracle_digests(voices, segs, offs, vids, seeds, pick, stride)
        for k, u in enumerate(pick):
            assert lens[u] == want_len[k] and int(sums[u]) == want[k], u
        gpu_ctx.set_option("arithmetic", 1)
        b.synthesize_async(d_fast, stride, d_len_fast)
        gpu_ctx.sync()
        maxdiff, _, structural = gpu_ctx.compare(d_out, d_fast, stride, d_len, d_len_fast, n)
        assert structural.sum() == 0
        assert maxdiff.max() <= G.FAST_TOLERANCE, maxdiff.max() * 2.0 ** 23
        # ... and once more pinned to the fast kernels of two lane mappings (the planner may have taken exact ones)
        for lanes in (1, 2):
            gpu_ctx.set_option("lanes_per_utterance", lanes)
            b.synthesize_async(d_fast, stride, d_len_fast)
            gpu_ctx.sync()
            assert "FAST" in gpu_ctx.last_kernel_name()
            maxdiff, _, structural = gpu_ctx.compare(d_out, d_fast, stride, d_len, d_len_fast, n)
            assert structural.sum() == 0
            assert maxdiff.max() <= G.FAST_TOLERANCE, (lanes, maxdiff.max() * 2.0 ** 23)
    finally:
        gpu_ctx.set_option("arithmetic", 0)
        gpu_ctx.set_option("lanes_per_utterance", 0)
        gpu_ctx.set_option("ragged_plan", 1)
        for p in (d_out, d_fast, d_len, d_len_fast):
            gpu_ctx.device_free(p)
        b.free()


@pytest.mark.perf
def test_two_rounds_of_the_device_in_packed_launch_order(gpu_ctx):
    """131 072 speech-like utterances (phonemes of 10 - 40 ms, rows of 0.12 - 0.95 s) on the REAL device: two one-wave
    workgroups per SIMD on one lane per utterance, launched in the packed order (launch_plan.cpp, "The workgroup
    dispatcher").  The order of a launch's workgroups moves no bit: all 131 072 digests equal those of the longest-first
    order, 40 rows spread over the batch == the oracle; and the packed order is the faster one on this device (kernel
    time, best of three each; skipped, not failed, where the device's clocks do not hold still)."""
    from conftest import skip_if_clocks_unstable
    n = 131072
    voices = W.single_voice()
    gpu_ctx.set_voices(voices)
    segs, offs, vids, seeds, _ = W.speech_like_batch(n, np.random.default_rng(7), scale=0.25)
    b = gpu_ctx.upload(segs, offs, vids, seeds)
    stride = (int(b.lengths().max()) + 64 + 63) // 64 * 64
    d_out = gpu_ctx.device_alloc(n * stride * 4)
    d_len = gpu_ctx.device_alloc(n * 4)
    ms = {}
    try:
        gpu_ctx.set_option("lanes_per_utterance", 1)
        for packed in (0, 1):
            gpu_ctx.set_option("packed_launch_order", packed)
            t = []
            for _ in range(3):
                lens, sums, maxabs, bad = _render(gpu_ctx, b, d_out, stride, d_len, n)
                t.append(gpu_ctx.last_kernel_ms())
            assert gpu_ctx.get_option("last_launch_packed") == packed and gpu_ctx.get_option("last_launch_blocks") == 1
            assert bad.sum() == 0 and maxabs.max() <= 1.0
            ms[packed] = (min(t), lens, sums)
        assert np.array_equal(ms[0][1], ms[1][1]) and np.array_equal(ms[0][2], ms[1][2])
        lens, sums = ms[1][1], ms[1][2]
        pick = sorted(set([0, 1, 63, 64, 65535, 65536, n - 1, int(np.argmax(lens)), int(np.argmin(lens))] +
                          [int(u) for u in np.random.default_rng(12).integers(0, n, 32)]))
        want, want_len = _oracle_digests(voices, segs, offs, vids, seeds, pick, stride)
        for k, u in enumerate(pick):
            assert lens[u] == want_len[k] and int(sums[u]) == want[k], u
        print(f"\n131 072 speech-like rows (phonemes x 0.25), one lane per utterance: longest first {ms[0][0]:.2f} ms, packed order "
              f"{ms[1][0]:.2f} ms = {ms[1][0] / ms[0][0]:.3f} x")
        if ms[1][0] > 0.97 * ms[0][0]:
            skip_if_clocks_unstable(gpu_ctx, f"packed order {ms[1][0]:.2f} ms against {ms[0][0]:.2f} longest first")
        assert ms[1][0] <= 0.97 * ms[0][0], ms
    finally:
        gpu_ctx.set_option("lanes_per_utterance", 0)
        gpu_ctx.set_option("packed_launch_order", 1)
        gpu_ctx.device_free(d_out)
        gpu_ctx.device_free(d_len)
        b.free()
