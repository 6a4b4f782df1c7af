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
