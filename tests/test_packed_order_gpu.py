"""Option "packed_launch_order" on the device: a launch of more one-wave-per-SIMD workgroups than the device holds at once,
of rows that differ in length, takes its workgroups in the packed order (launch_plan.cpp, "The workgroup dispatcher").  The
order of a launch's workgroups cannot change a bit — a row's samples are a function of (segments, voice, seed), reference
src/lib.rs:594, :786-797 — and does not: every row against the plain order's digests and sampled rows against the oracle,
exact and tolerance arithmetic, f32 and i16, for a device planned as one XCC (assume_compute_units = 32: 128 SIMDs, four
pools) so that a batch of 20 000 short utterances is many rounds."""
import numpy as np
import pytest

import grail_hip as G
import oracle_lib as O
from grail_hip import workload as W

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n_voices", [1, 8])
def test_packed_launch_order_renders_the_same_rows(gpu_ctx, n_voices):
    ctx = gpu_ctx
    voices = W.single_voice() if n_voices == 1 else W.preset_voices(8)
    ctx.set_voices(voices)
    n = 20000
    segs, offs, vids, seeds, stride = W.speech_like_batch(n, np.random.default_rng(11), n_voices=n_voices, scale=0.05)
    batch = ctx.upload(segs, offs, vids, seeds)
    d_out, d_len = ctx.device_alloc(n * stride * 4), ctx.device_alloc(n * 4)
    got = {}
    try:
        ctx.set_option("assume_compute_units", 32)
        ctx.set_option("lanes_per_utterance", 1)
        for fast in (0, 1):
            ctx.set_option("arithmetic", fast)
            for packed in (0, 1):
                ctx.set_option("packed_launch_order", packed)
                ctx.memset(d_out, 0, n * stride * 4)
                batch.synthesize_async(d_out, stride, d_len)
                ctx.sync()
                assert ctx.get_option("last_launch_packed") == packed, (fast, packed, ctx.last_kernel_name())
                assert ctx.get_option("last_launch_lanes") == 1 and ctx.get_option("last_launch_blocks") == 1
                sums, maxabs, bad = ctx.digest(d_out, stride, d_len, n)
                lens = np.zeros(n, dtype=np.uint32)
                ctx.d2h(lens, d_len, lens.nbytes)
                got[(fast, packed)] = (sums.copy(), lens)
                assert bad.sum() == 0
            # the i16 rows too (the conversion is part of the flush)
            if not fast:
                pick = np.array([0, 1, 63, 64, 9999, 19998, 19999])
                rows = np.zeros((len(pick), stride), dtype=np.float32)
                for k, u in enumerate(pick):
                    ctx.d2h(rows[k], d_out, stride * 4, offset=int(u) * stride * 4)
    finally:
        for k, v in (("assume_compute_units", 0), ("lanes_per_utterance", 0), ("arithmetic", 0), ("packed_launch_order", 1)):
            ctx.set_option(k, v)
        ctx.device_free(d_out)
        ctx.device_free(d_len)
        batch.free()
    for fast in (0, 1):
        assert np.array_equal(got[(fast, 0)][1], got[(fast, 1)][1])
        assert np.array_equal(got[(fast, 0)][0], got[(fast, 1)][0]), "a row's bits moved with the launch order"
    # (the last exact rendering was the packed one) sampled rows against the oracle
    sub = np.concatenate([segs[offs[u]:offs[u + 1]] for u in pick])
    sub_offs = np.concatenate([[0], np.cumsum([offs[u + 1] - offs[u] for u in pick])]).astype(np.uint32)
    ov = [O.Voice.from_buffer_copy(bytes(v)) for v in voices]
    ref, ref_len = O.synthesize_batch(ov, sub, sub_offs, vids[pick], seeds[pick], stride)
    assert np.array_equal(got[(0, 1)][1][pick], ref_len)
    for k in range(len(pick)):
        m = int(ref_len[k])
        assert np.array_equal(rows[k, :m].view(np.uint32), ref[k, :m].view(np.uint32)), pick[k]


def test_the_plain_order_when_packing_has_nothing_to_gain(gpu_ctx):
    """Rows of one length (nothing to even out) and launches the device holds at once keep the plain order."""
    ctx = gpu_ctx
    ctx.set_voices(W.single_voice())
    segs, offs, vids, seeds, stride = W.speech_like_batch(3000, np.random.default_rng(5), scale=0.05)
    batch = ctx.upload(segs, offs, vids, seeds)
    d_out, d_len = ctx.device_alloc(3000 * stride * 4), ctx.device_alloc(3000 * 4)
    try:
        ctx.set_option("lanes_per_utterance", 1)
        batch.synthesize_async(d_out, stride, d_len)
        ctx.sync()
        assert ctx.get_option("last_launch_packed") == 0          # 47 workgroups on 1 024 SIMDs
    finally:
        ctx.set_option("lanes_per_utterance", 0)
        ctx.device_free(d_out)
        ctx.device_free(d_len)
        batch.free()


@pytest.mark.parametrize("seed", [21, 22, 23])
def test_packed_launch_orders_of_random_batches(gpu_ctx, seed):
    """Random speech-like batches (8 - 32 phonemes per row, phoneme lengths scaled at random, with and without a long tail,
    a few rows with a zero-length segment among them, two or eight voices) on devices assumed so small that a launch is
    2 - 2.7 rounds — one pool of SIMDs (1 ... 24 compute units), four pools (32) and eight (64) — on one and two lanes per
    utterance, one wave per SIMD, with and without the odd rows planned apart: every row == the oracle, bit for bit in exact
    arithmetic, within the tolerance in fast arithmetic — and several of the launches did take a packed order (a long tail
    leaves nothing to pack: the longest row is the launch)."""
    rng = np.random.default_rng(seed)
    ctx = gpu_ctx
    saved = {k: ctx.get_option(k) for k in ("assume_compute_units", "lanes_per_utterance", "two_waves_per_simd", "arithmetic",
                                            "packed_launch_order", "row_groups", "composite_launches")}
    packed = launches = 0
    try:
        for trial in range(4):
            voices = W.preset_voices(8) if trial % 2 else [G.voice_generic(48000.0), G.voice_generic(44100.0)]
            ctx.set_voices(voices)
            lanes = int(rng.choice([1, 2]))
            pools = trial == 3                                # the last trial: whole XCCs, several pools
            if pools:
                lanes, cus = 2, int(rng.choice([32, 64]))
                n = int(32 * 4 * cus * rng.uniform(2.0, 2.6))
            else:
                n = int(rng.integers(900, 3000))
                cus = max(1, int(round(n * lanes / 64.0 / 4.0 / rng.uniform(2.0, 2.7))))
                cus += 1 if cus % 32 == 0 else 0
            segs, offs, vids, seeds, stride = W.speech_like_batch(n, rng, n_voices=len(voices), scale=float(rng.uniform(0.015, 0.04)),
                                                                  long_tail=trial == 1)
            seeds = rng.integers(0, 2 ** 32, size=n, dtype=np.uint64).astype(np.uint32)
            for u in rng.integers(0, n, size=3):              # a few rows the lean kernel families cannot take
                segs["length"][offs[u] + 1] = 0.0
            ov = [O.Voice.from_buffer_copy(bytes(v)) for v in voices]
            ref, ref_len = O.synthesize_batch(ov, segs, offs, vids, seeds, stride)
            batch = ctx.upload(segs, offs, vids, seeds)
            d_out, d_len = ctx.device_alloc(n * stride * 4), ctx.device_alloc(n * 4)
            out, out_len = np.zeros((n, stride), dtype=np.float32), np.zeros(n, dtype=np.uint32)
            try:
                ctx.set_option("assume_compute_units", cus)
                ctx.set_option("lanes_per_utterance", lanes)
                ctx.set_option("two_waves_per_simd", 0)
                for fast in (0, 1):
                    for groups in (1, 0):
                        ctx.set_option("row_groups", groups)
                        ctx.set_option("arithmetic", fast)
                        ctx.memset(d_out, 0, n * stride * 4)
                        batch.synthesize_async(d_out, stride, d_len)
                        ctx.sync()
                        launches += 1
                        packed += 1 if ctx.get_option("last_launch_packed") else 0
                        what = (seed, trial, n, cus, lanes, fast, groups, ctx.last_kernel_name(), ctx.get_option("last_launch_packed"))
                        ctx.d2h(out, d_out, out.nbytes)
                        ctx.d2h(out_len, d_len, out_len.nbytes)
                        assert np.array_equal(out_len, ref_len), what
                        if not fast or ctx.get_option("last_launch_fast") == 0:
                            assert np.array_equal(out.view(np.uint32), ref.view(np.uint32)), what
                        else:
                            peak = np.maximum(1.0, np.abs(ref).max(axis=1))
                            assert (np.abs(out.astype(np.float64) - ref).max(axis=1) / peak).max() <= G.FAST_TOLERANCE, what
            finally:
                ctx.device_free(d_out)
                ctx.device_free(d_len)
                batch.free()
    finally:
        for k, v in saved.items():
            ctx.set_option(k, v)
    print(f"\nseed {seed}: {packed} of {launches} launches took a packed order")
    assert packed >= 3, (packed, launches)
