"""The node call's partition and gather, without a GPU (contiguous shards, no exchange step).

grail_node_synthesize_batch hands device slot i the view grail_node_shard_of computes — segs + first_seg, the rows'
seg_offsets rebased to it, voice_ids / jitter_seeds / out / out_len advanced by first_row — and nothing else.  Here that
view is rendered by the ORACLE, shard by shard, into slices of one buffer, and must give the oracle's rendering of the
whole batch: every row present once, in place, with its own segments, voice and seed (per-utterance state is
self-contained, reference src/lib.rs:470-488, 724-748, 839-854)."""
import ctypes as C

import numpy as np
import pytest

import grail_hip as G
import oracle_lib as O
from grail_hip import workload as W


def _ragged_batch(n_utt, seed):
    """Rows of 0 - 5 segments (empty rows included), three voices, seeds of their own."""
    rng = np.random.default_rng(seed)
    counts = rng.integers(0, 6, size=n_utt)
    offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint32)
    n_segs = int(offs[-1])
    segs = np.zeros(n_segs, dtype=G.PHONEME_DTYPE)
    segs["phoneme"] = rng.choice([G.PH_SILENCE, G.PH_STOP, G.PH_GLIDE, G.PH_A, G.PH_E], size=n_segs)
    segs["length"] = rng.uniform(0.001, 0.004, size=n_segs).astype(np.float32)
    segs["blend_length"] = rng.uniform(0.0005, 0.004, size=n_segs).astype(np.float32)
    segs["frequency"] = (rng.uniform(90, 220, size=n_segs) / 48000.0).astype(np.float32)
    vids = rng.integers(0, 3, size=n_utt).astype(np.uint32)
    seeds = rng.integers(0, 2 ** 32, size=n_utt, dtype=np.uint64).astype(np.uint32)
    return segs, offs, vids, seeds


@pytest.mark.parametrize("n_devices", [1, 2, 3, 8])
@pytest.mark.parametrize("n_utt", [0, 1, 5, 7, 64, 203])
def test_shards_partition_the_rows_and_the_segments(built, n_devices, n_utt):
    segs, offs, vids, seeds = _ragged_batch(n_utt, 1000 + n_utt)
    next_row, next_seg = 0, 0
    for i in range(n_devices):
        sh, rebased = G.node_shard_of(offs, i, n_devices)
        b, e = G.shard_range(n_utt, i, n_devices)
        assert (sh.first_row, sh.rows) == (b, e - b) == (next_row, e - b)
        assert sh.first_seg == next_seg == int(offs[b]) and sh.n_segs == int(offs[e]) - int(offs[b])
        assert len(rebased) == sh.rows + 1 and rebased[0] == 0 and rebased[-1] == sh.n_segs
        assert np.array_equal(rebased, offs[b:e + 1] - offs[b])
        next_row, next_seg = e, int(offs[e])
    assert next_row == n_utt and next_seg == int(offs[-1])
    # sizes differ by at most one row
    sizes = [G.node_shard_of(offs, i, n_devices)[0].rows for i in range(n_devices)]
    assert max(sizes) - min(sizes) <= 1


@pytest.mark.parametrize("n_devices", [1, 2, 3, 8])
@pytest.mark.parametrize("n_utt", [5, 7, 37])
def test_rendering_the_shard_views_gives_the_whole_batch(built, n_devices, n_utt):
    voices = [O.Voice.from_buffer_copy(bytes(v)) for v in W.preset_voices(3)]
    segs, offs, vids, seeds = _ragged_batch(n_utt, 77 + n_utt)
    stride = 1024
    whole, whole_len = O.synthesize_batch(voices, segs, offs, vids, seeds, stride)
    out = np.full((n_utt, stride), np.float32(np.nan), dtype=np.float32)
    out_len = np.full(n_utt, 0xFFFFFFFF, dtype=np.uint32)
    for i in range(n_devices):
        sh, rebased = G.node_shard_of(offs, i, n_devices)
        if sh.rows == 0:
            continue
        r0, r1 = sh.first_row, sh.first_row + sh.rows
        part, part_len = O.synthesize_batch(voices, segs[sh.first_seg:sh.first_seg + sh.n_segs], rebased, vids[r0:r1],
                                            seeds[r0:r1], stride)
        out[r0:r1] = part
        out_len[r0:r1] = part_len
    assert np.array_equal(out_len, whole_len)
    for u in range(n_utt):
        n = int(whole_len[u])
        assert np.array_equal(out[u, :n].view(np.uint32), whole[u, :n].view(np.uint32)), u


def test_shard_of_refuses_bad_arguments(built):
    L = G.load()
    offs = np.array([0, 1, 2], dtype=np.uint32)
    u32p = C.POINTER(C.c_uint32)
    sh = G.NodeShard()
    assert L.grail_node_shard_of(offs.ctypes.data_as(u32p), 2, 2, 2, C.byref(sh), None, 0) == G.ERR_INVALID_ARG
    assert L.grail_node_shard_of(offs.ctypes.data_as(u32p), 2, 0, 0, C.byref(sh), None, 0) == G.ERR_INVALID_ARG
    assert L.grail_node_shard_of(None, 2, 0, 1, C.byref(sh), None, 0) == G.ERR_INVALID_ARG
    small = np.zeros(2, dtype=np.uint32)
    assert L.grail_node_shard_of(offs.ctypes.data_as(u32p), 2, 0, 1, C.byref(sh), small.ctypes.data_as(u32p),
                                 2) == G.ERR_BUFFER_TOO_SMALL
    # an empty batch needs no offsets
    assert L.grail_node_shard_of(None, 0, 0, 4, C.byref(sh), None, 0) == G.OK and sh.rows == 0 and sh.n_segs == 0


def test_a_node_needs_a_device(built):
    """No CPU fallback: on a host without a GPU the node cannot be created, and says which slot failed."""
    if G.device_count() > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(G.GrailError) as e:
        G.Node([0, 1])
    assert e.value.status == G.ERR_NO_DEVICE and "device[0] = 0" in str(e.value)
