"""grail_node_*: one call, every GPU of the node (SURVEY §8e "one process, 8 devices"; the reference host makes one call
for its whole job, examples/cli.rs:175-184).  The box has one GPU: `devices = [0]` runs the real thing — communicator
formed in-process, the voice table carried by ncclBroadcast, one shard — and `devices = [0, 0, 0, 0]` runs four contexts
and four host threads on that GPU, with the table installed per context under the explicit test-only option (RCCL refuses
a communicator that names a GPU twice).  Every row must be the oracle's bits and the single-context call's bits."""
import os

import numpy as np
import pytest

import grail_hip as G
import oracle_lib as O
from grail_hip import workload as W

pytestmark = pytest.mark.gpu

os.environ.setdefault("NCCL_IB_DISABLE", "1")
os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")


def _ov(voices):
    return [O.Voice.from_buffer_copy(bytes(v)) for v in voices]


def _same_bits(out, out_len, ref, ref_len, what):
    assert np.array_equal(out_len, ref_len), what
    for u in range(len(ref_len)):
        n = int(ref_len[u])
        assert np.array_equal(out[u, :n].view(np.uint32), ref[u, :n].view(np.uint32)), (what, u)
        assert not out[u, n:].any(), (what, u, "rows end in zeros")


def _ragged(n_utt, seed, n_voices=8):
    rng = np.random.default_rng(seed)
    counts = rng.integers(0, 7, size=n_utt)
    offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint32)
    n_segs = int(offs[-1])
    segs = np.zeros(n_segs, dtype=G.PHONEME_DTYPE)
    segs["phoneme"] = rng.choice([G.PH_SILENCE, G.PH_STOP, G.PH_GLIDE, G.PH_A, G.PH_E], size=n_segs)
    segs["length"] = rng.uniform(0.002, 0.02, size=n_segs).astype(np.float32)
    segs["blend_length"] = rng.uniform(0.001, 0.02, size=n_segs).astype(np.float32)
    segs["frequency"] = (rng.uniform(90, 220, size=n_segs) / 48000.0).astype(np.float32)
    vids = rng.integers(0, n_voices, size=n_utt).astype(np.uint32)
    seeds = rng.integers(0, 2 ** 32, size=n_utt, dtype=np.uint64).astype(np.uint32)
    return segs, offs, vids, seeds


def test_one_device_over_rccl(built, gpu_ctx):
    """devices = [0]: ncclCommInitAll inside the process, slot 0's table broadcast on its stream, one shard."""
    voices = W.preset_voices(8)
    segs, offs, vids, seeds = W.make_batch(133, n_voices=8, length=0.03, blend_length=0.03125)
    stride = W.max_samples(length=0.03)
    ref, ref_len = O.synthesize_batch(_ov(voices), segs, offs, vids, seeds, stride)
    with G.Node([0]) as node:
        assert node.size() == 1 and node.get_option("node_devices") == 1
        assert node.get_option("node_rccl_ranks") == 0          # no communicator before the first table
        node.set_voices(voices)
        assert node.get_option("node_rccl_ranks") == 1          # what RCCL itself reports (ncclCommCount)
        ctx = node.context(0)
        assert ctx.comm_info() == (1, 0)
        assert all(bytes(a) == bytes(b) for a, b in zip(ctx.get_voices(), voices))
        out, out_len = node.synthesize(segs, offs, vids, seeds, out_stride=stride)
        _same_bits(out, out_len, ref, ref_len, "node [0]")
        # the table may be replaced: the communicator is kept
        node.set_voices(voices[:3] + voices[:5])
        assert node.get_option("node_rccl_ranks") == 1
    gpu_ctx.set_voices(voices)
    one, one_len = gpu_ctx.synthesize(segs, offs, vids, seeds, out_stride=stride)
    _same_bits(out, out_len, one, one_len, "node [0] against one context")


def test_duplicate_devices_are_refused_by_the_broadcast_and_served_under_the_option(built):
    voices = W.preset_voices(8)
    with G.Node([0, 0, 0, 0]) as node:
        with pytest.raises(G.GrailError) as e:
            node.set_voices(voices)
        assert e.value.status == G.ERR_RCCL and "twice" in str(e.value)
        # nothing was installed silently
        with pytest.raises(G.GrailError) as e:
            node.synthesize(*W.make_batch(8, length=0.01, blend_length=0.01), out_stride=W.max_samples(length=0.01))
        assert e.value.status == G.ERR_NO_VOICES and "device[0] = 0" in str(e.value)
        node.set_option("node_voices_without_rccl", 1)
        assert node.get_option("node_voices_without_rccl") == 1
        node.set_voices(voices)
        assert node.get_option("node_rccl_ranks") == 0
        for i in range(4):
            assert all(bytes(a) == bytes(b) for a, b in zip(node.context(i).get_voices(), voices))


@pytest.mark.parametrize("n_devices", [2, 3, 4, 8])
@pytest.mark.parametrize("n_utt", [0, 1, 3, 131])
def test_shards_on_one_gpu_give_the_single_context_rows(built, gpu_ctx, n_devices, n_utt):
    """Ragged row counts (fewer rows than devices included), rows of 0 - 6 segments, eight voices: every row in place,
    bit-identical to the oracle and to one grail_synthesize_batch."""
    voices = W.preset_voices(8)
    segs, offs, vids, seeds = _ragged(n_utt, 5 + n_utt)
    stride = 6 * 1024
    ref, ref_len = O.synthesize_batch(_ov(voices), segs, offs, vids, seeds, stride)
    with G.Node([0] * n_devices, voices_without_rccl=True) as node:
        node.set_voices(voices)
        out, out_len = node.synthesize(segs, offs, vids, seeds, out_stride=stride)
        assert np.array_equal(node.lengths(segs, offs, vids), ref_len)
        ms = node.last_shard_ms()
        assert len(ms) == n_devices and sum(1 for x in ms if x > 0.0) == min(n_devices, n_utt)
    _same_bits(out, out_len, ref, ref_len, f"node of {n_devices}")
    gpu_ctx.set_voices(voices)
    one, one_len = gpu_ctx.synthesize(segs, offs, vids, seeds, out_stride=stride)
    _same_bits(out, out_len, one, one_len, "against one context")


def test_a_batch_that_fills_four_shards_exact_and_fast(built, gpu_ctx):
    """6 000 utterances of 0.4 s over four contexts (1 500 rows each, launches in flight together), pinned destination:
    exact rows == one context's; fast rows with the lane mapping pinned ("lanes_per_utterance" = 1, the batch-invariant
    family) == one context's fast rows bit for bit, and within the tolerance of the exact ones."""
    voices = W.single_voice()
    n_utt = 6000
    segs, offs, vids, seeds = W.make_batch(n_utt, length=0.1, blend_length=0.125)
    stride = (W.max_samples(length=0.1) + 63) // 64 * 64
    gpu_ctx.set_voices(voices)
    one, one_len = gpu_ctx.synthesize(segs, offs, vids, seeds, out_stride=stride)
    pick = [0, 1, 63, 64, 1499, 1500, 2999, 3000, 4499, 4500, 5998, 5999]
    sub = np.concatenate([segs[offs[u]:offs[u + 1]] for u in pick])
    ref, ref_len = O.synthesize_batch(_ov(voices), sub, np.arange(len(pick) + 1, dtype=np.uint32) * 4, vids[pick],
                                      seeds[pick], stride)
    with G.Node([0, 0, 0, 0], voices_without_rccl=True) as node:
        node.set_voices(voices)
        dst = node.host_alloc((n_utt, stride), np.float32)
        try:
            out, out_len = node.synthesize(segs, offs, vids, seeds, out=dst)
            assert np.array_equal(out_len, one_len)
            assert np.array_equal(out.view(np.uint32), one.view(np.uint32))
            _same_bits(out[pick], out_len[pick], ref, ref_len, "sampled rows against the oracle")
            node.set_option("arithmetic", 1)
            node.set_option("lanes_per_utterance", 1)
            fast, fast_len = node.synthesize(segs, offs, vids, seeds, out=dst)
            fast = fast.copy()
            assert node.context(3).get_option("last_launch_fast") == 1
        finally:
            node.host_free(dst)
    try:
        gpu_ctx.set_option("arithmetic", 1)
        gpu_ctx.set_option("lanes_per_utterance", 1)
        one_fast, one_fast_len = gpu_ctx.synthesize(segs, offs, vids, seeds, out_stride=stride)
        assert gpu_ctx.get_option("last_launch_fast") == 1
    finally:
        gpu_ctx.set_option("arithmetic", 0)
        gpu_ctx.set_option("lanes_per_utterance", 0)
    assert np.array_equal(fast_len, one_fast_len) and np.array_equal(fast_len, one_len)
    assert np.array_equal(fast.view(np.uint32), one_fast.view(np.uint32))
    worst = float(np.max(np.abs(fast.astype(np.float64) - one)))
    assert 0.0 < worst <= G.FAST_TOLERANCE


def test_pcm16_elems_and_say_over_the_node(built, gpu_ctx):
    voices = W.preset_voices(8)
    segs, offs, vids, seeds = W.make_batch(50, n_voices=8, length=0.02, blend_length=0.02)
    stride = W.max_samples(length=0.02)
    gpu_ctx.set_voices(voices)
    with G.Node([0, 0, 0], voices_without_rccl=True) as node:
        node.set_voices(voices)
        pcm, pcm_len = node.synthesize(segs, offs, vids, seeds, out_stride=stride, pcm16=True)
        one, one_len = gpu_ctx.synthesize_pcm16(segs, offs, vids, seeds, out_stride=stride)
        assert np.array_equal(pcm_len, one_len) and np.array_equal(pcm, one)
        # caller-built SequenceElems (no Selector)
        elems = []
        for s in segs[: offs[10]]:
            e = G.SequenceElem()
            v = voices[0]
            e.has_elem = 1 if s["phoneme"] >= G.PH_A else 0
            if e.has_elem:
                e.elem = G.SynthesisElem.from_buffer_copy(bytes(v.phonemes[s["phoneme"] - G.PH_A]))
                e.elem.frequency = min(float(s["frequency"]), 0.5)
            e.length, e.blend_length = float(s["length"]), float(s["blend_length"])
            elems.append(e)
        got, got_len = node.synthesize_elems(elems, offs[:11], None, seeds[:10], out_stride=stride)
        want, want_len = gpu_ctx.synthesize_elems(elems, offs[:11], None, seeds[:10], out_stride=stride)
        assert np.array_equal(got_len, want_len) and np.array_equal(got.view(np.uint32), want.view(np.uint32))
        # text in, PCM out (examples/cli.rs:175-184) for seven texts over three devices
        texts = ["a", "ae", "", "e a", "aaa", "ea ea", "a e"]
        said, said_len = node.say(texts, voice_ids=np.arange(7) % 8, jitter_seeds=np.arange(7), out_stride=8 * 48000)
        ref, ref_len = gpu_ctx.say(texts, voice_ids=np.arange(7) % 8, jitter_seeds=np.arange(7), out_stride=8 * 48000)
        assert np.array_equal(said_len, ref_len) and np.array_equal(said.view(np.uint32), ref.view(np.uint32))


def test_failures_name_the_device_slot(built):
    voices = W.single_voice()
    segs, offs, vids, seeds = W.make_batch(9, length=0.01, blend_length=0.01)
    stride = W.max_samples(length=0.01)
    with G.Node([0, 0, 0], voices_without_rccl=True) as node:
        node.set_voices(voices)
        bad = vids.copy()
        bad[7] = 5                                        # slot 2's shard (rows 6 - 8) names a voice the table lacks
        with pytest.raises(G.GrailError) as e:
            node.synthesize(segs, offs, bad, seeds, out_stride=stride)
        assert e.value.status == G.ERR_INVALID_ARG and "device[2] = 0" in str(e.value)
        # a row that does not fit: the status of the one-context call, the other rows complete
        out, out_len = node.synthesize(segs, offs, vids, seeds, out_stride=64, allow_truncation=True)
        assert (out_len == 64).all()
        L = G.load()
        st = L.grail_node_synthesize_batch(node.handle, segs.ctypes.data, offs.ctypes.data, None, None, 9,
                                           out.ctypes.data, 64, out_len.ctypes.data, G.OUT_HOST)
        assert st == G.ERR_BUFFER_TOO_SMALL
        st = L.grail_node_synthesize_batch(node.handle, segs.ctypes.data, offs.ctypes.data, None, None, 9,
                                           out.ctypes.data, 64, out_len.ctypes.data, G.OUT_DEVICE)
        assert st == G.ERR_INVALID_ARG
        # still usable
        out, out_len = node.synthesize(segs, offs, vids, seeds, out_stride=stride)
        ref, ref_len = O.synthesize_batch(_ov(voices), segs, offs, vids, seeds, stride)
        _same_bits(out, out_len, ref, ref_len, "after the failures")
    with pytest.raises(G.GrailError) as e:
        G.Node([0, 99])
    assert e.value.status == G.ERR_NO_DEVICE and "device[1] = 99" in str(e.value)


def test_cpp_facade_node_example(built):
    """grail::Node (include/grail.hpp) end to end: examples/grail_node_say.cpp renders its arguments over the node and
    compares every row with one device's; over RCCL with one device, and over three contexts on it under the option."""
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "grail-rs_amd", "lib", "grail_node_say")
    assert os.path.exists(exe)
    texts = ["a", "ae", "e a e", "aa", "eee a"]
    r = subprocess.run([exe, "--devices", "0"] + texts, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "5 utterances over 1 device slots (RCCL ranks 1)" in r.stdout and "differ from one device's: 0" in r.stdout
    r = subprocess.run([exe, "--devices", "0,0,0", "--without-rccl"] + texts, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "5 utterances over 3 device slots (RCCL ranks 0)" in r.stdout and "differ from one device's: 0" in r.stdout
    r = subprocess.run([exe, "--devices", "0,0"] + texts, capture_output=True, text=True, timeout=300)
    assert r.returncode == 1 and "twice" in r.stderr and "status -6" in r.stderr


def test_rows_left_in_hbm(built, gpu_ctx):
    """grail_node_synthesize_batch_device: every slot's shard stays on its GPU; the rows' on-device digests are those of one
    context's rendering, an empty shard needs no buffer."""
    voices = W.preset_voices(8)
    n_utt = 1000
    segs, offs, vids, seeds = W.make_batch(n_utt, n_voices=8, length=0.05, blend_length=0.0625)
    stride = (W.max_samples(length=0.05) + 63) // 64 * 64
    gpu_ctx.set_voices(voices)
    b = gpu_ctx.upload(segs, offs, vids, seeds)
    d_out, d_len = gpu_ctx.device_alloc(n_utt * stride * 4), gpu_ctx.device_alloc(n_utt * 4)
    try:
        b.synthesize_async(d_out, stride, d_len)
        gpu_ctx.sync()
        want, _, _ = gpu_ctx.digest(d_out, stride, d_len, n_utt)
        want_len = np.zeros(n_utt, dtype=np.uint32)
        gpu_ctx.d2h(want_len, d_len, want_len.nbytes)
    finally:
        gpu_ctx.device_free(d_out)
        gpu_ctx.device_free(d_len)
        b.free()
    with G.Node([0, 0, 0], voices_without_rccl=True) as node:
        node.set_voices(voices)
        shards = [G.node_shard_of(offs, i, 3)[0] for i in range(3)]
        ctxs = [node.context(i) for i in range(3)]
        bufs = [c.device_alloc(int(s.rows) * stride * 4) for c, s in zip(ctxs, shards)]
        try:
            out_len = node.synthesize_device(segs, offs, vids, seeds, bufs, stride)
            assert np.array_equal(out_len, want_len)
            for c, s, buf in zip(ctxs, shards, bufs):
                rows = int(s.rows)
                d_l = c.device_alloc(rows * 4)
                c.h2d(d_l, np.ascontiguousarray(out_len[s.first_row:s.first_row + rows]), rows * 4)
                got, _, bad = c.digest(buf, stride, d_l, rows)
                c.device_free(d_l)
                assert bad.sum() == 0 and np.array_equal(got, want[s.first_row:s.first_row + rows])
            # two rows over three slots (grail_shard_range: [0, 0), [0, 1), [1, 2)): the first slot has nothing to render
            # and needs no buffer
            assert [int(G.node_shard_of(offs[:3], i, 3)[0].rows) for i in range(3)] == [0, 1, 1]
            few = node.synthesize_device(segs[:8], offs[:3], vids[:2], seeds[:2], [None, bufs[1], bufs[2]], stride)
            assert np.array_equal(few, want_len[:2])
            with pytest.raises(G.GrailError) as e:
                node.synthesize_device(segs, offs, vids, seeds, [bufs[0], None, bufs[2]], stride)
            assert e.value.status == G.ERR_INVALID_ARG and "out_dev[1]" in str(e.value)
        finally:
            for c, buf in zip(ctxs, bufs):
                c.device_free(buf)
