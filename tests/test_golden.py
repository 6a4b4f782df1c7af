"""Committed golden vectors (tests/golden/hotpath_v1.npz, made by tests/golden/make_golden.py):
the oracle must still produce them (CPU), and so must the HIP path (GPU)."""
import os

import numpy as np
import pytest

import oracle_lib as O

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "hotpath_v1.npz"))
NAMES = sorted({k.split("/")[0] for k in GOLD.files if "/" in k})


def case(name):
    rate = float(GOLD[name + "/rate"])
    return (None if rate == 0.0 else rate, GOLD[name + "/segs"], int(GOLD[name + "/seed"]),
            int(GOLD[name + "/len"]), GOLD[name + "/pcm"], int(GOLD[name + "/sum"]))


def check(name, pcm, n):
    rate, segs, seed, want_len, want_pcm, want_sum = case(name)
    assert n == want_len
    assert np.array_equal(pcm[:len(want_pcm)].view(np.uint32), want_pcm.view(np.uint32))
    assert int(pcm[:n].view(np.uint32).astype(np.uint64).sum()) == want_sum


def test_golden_file_has_the_expected_cases():
    assert len(NAMES) == 6 and "text_a_head" in NAMES


@pytest.mark.parametrize("name", NAMES)
def test_oracle_reproduces_golden(name):
    rate, segs, seed, *_ = case(name)
    pcm, n = O.synthesize_phonemes(O.voice_generic(rate), segs, seed)
    check(name, pcm, n)


def test_voice_tables_match_golden(built):
    import grail_hip as G
    for key, rate in (("voice_44k", None), ("voice_48k", 48000.0)):
        want = GOLD[key]
        assert np.array_equal(np.frombuffer(bytes(O.voice_generic(rate)), dtype=np.float32), want)
        assert np.array_equal(np.frombuffer(bytes(G.voice_generic(rate)), dtype=np.float32), want)


@pytest.mark.gpu
@pytest.mark.parametrize("lanes", [0, 1, 2, 4, 8])
def test_hip_path_reproduces_golden(gpu_ctx, lanes):
    import grail_hip as G
    gpu_ctx.set_option("lanes_per_utterance", lanes)
    for name in NAMES:
        rate, segs, seed, want_len, *_ = case(name)
        gpu_ctx.set_voices([G.voice_generic(rate)])
        stride = (want_len + 63) // 64 * 64
        out, out_len = gpu_ctx.synthesize(segs, [0, len(segs)], None, [seed], out_stride=stride)
        check(name, out[0], int(out_len[0]))
    gpu_ctx.set_option("lanes_per_utterance", 0)
