"""Soak (opt-in: GRAIL_SOAK=1): EVERY utterance of the full-size batch — 65536 x 96006 = 6.29e9
samples — against the oracle, through per-utterance checksums (sum of the samples' bit patterns
mod 2^64, computed on the device for the HIP rows and on the host for the oracle's).  The oracle
renders the batch in chunks on all host cores; a few minutes of CPU, so it is not part of the
default suite.  GRAIL_SOAK_UTTS picks another batch size (another kernel family), GRAIL_SOAK_FIRST the first
utterance of the synthetic corpus (k * 65536: shard k of BASELINE config 5, what rank k of an 8-GPU run renders),
GRAIL_SOAK_SPEECH=<scale> the speech-like corpus instead of the bench corpus.
Last runs: profiles/r03_full_parity.txt, r05_full_parity.txt, r05_full_parity_speech_like.txt, r06_full_parity_speech_like_131072.txt."""
import os

import numpy as np
import pytest

import grail_hip as G
import oracle_lib as O
from grail_hip import workload as W

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(os.environ.get("GRAIL_SOAK") != "1", reason="opt-in soak: GRAIL_SOAK=1")]


@pytest.mark.parametrize("n_voices", [1, 8])
def test_every_utterance_of_the_full_batch(gpu_ctx, n_voices):
    n_utt = int(os.environ.get("GRAIL_SOAK_UTTS", "65536"))
    voices = W.single_voice() if n_voices == 1 else W.preset_voices(8)
    ov = [O.Voice.from_buffer_copy(bytes(v)) for v in voices]
    gpu_ctx.set_voices(voices)
    first_utt = int(os.environ.get("GRAIL_SOAK_FIRST", "0"))
    segs, offs, vids, seeds = W.make_batch(n_utt, first_utt=first_utt, n_voices=n_voices)
    stride = W.max_samples()
    if os.environ.get("GRAIL_SOAK_RANDOM"):
        # another structure than the bench corpus': random segment lengths, one blend length (GRAIL_SOAK_BLEND, a power
        # of two keeps the pipelined workgroups in play), pitches jumping between 70 and 400 Hz, Silence / A / E at random
        rng = np.random.default_rng(int(os.environ["GRAIL_SOAK_RANDOM"]))
        k = len(segs)
        segs["length"] = rng.uniform(0.03, 0.3, k).astype(np.float32)
        segs["blend_length"] = np.float32(float(os.environ.get("GRAIL_SOAK_BLEND", "0.0625")))
        segs["frequency"] = (rng.choice([70.0, 110.0, 200.0, 400.0], k) / 48000.0).astype(np.float32)
        segs["phoneme"] = rng.choice([G.PH_SILENCE, G.PH_A, G.PH_E, G.PH_A], k)
        stride = 4 * 14400 + 64
    if os.environ.get("GRAIL_SOAK_SPEECH"):
        # the speech-like corpus (rows of 0.5 - 3.8 s; GRAIL_SOAK_SPEECH = the scale of its phoneme lengths): the launch plan
        # by the rows' lengths and events — wider mappings in several rounds, two waves per SIMD, the runs between events
        segs, offs, vids, seeds, stride = W.speech_like_batch(n_utt, np.random.default_rng(7), n_voices=n_voices,
                                                              scale=float(os.environ["GRAIL_SOAK_SPEECH"]))
    b = gpu_ctx.upload(segs, offs, vids, seeds)
    d_out = gpu_ctx.device_alloc(n_utt * stride * 4)
    d_len = gpu_ctx.device_alloc(n_utt * 4)
    try:
        b.synthesize_async(d_out, stride, d_len)
        gpu_ctx.sync()
        kernel = gpu_ctx.last_kernel_name() + (", packed launch order" if gpu_ctx.get_option("last_launch_packed") else "")
        out_len = np.zeros(n_utt, dtype=np.uint32)
        gpu_ctx.d2h(out_len, d_len, n_utt * 4)
        sums, _, bad = gpu_ctx.digest(d_out, stride, d_len, n_utt)
    finally:
        gpu_ctx.device_free(d_out)
        gpu_ctx.device_free(d_len)
        b.free()
    assert bad.sum() == 0
    threads = len(os.sched_getaffinity(0))
    chunk = 1024
    checked = 0
    for first in range(0, n_utt, chunk):
        last = min(first + chunk, n_utt)
        sub = segs[offs[first]:offs[last]]
        sub_offs = (offs[first:last + 1] - offs[first]).astype(np.uint32)
        ref, ref_len, _ = O.synthesize_batch_threads(ov, sub, sub_offs, vids[first:last], seeds[first:last],
                                                     stride, threads)
        assert np.array_equal(ref_len, out_len[first:last]), first
        bits = ref.view(np.uint32)
        mask = np.arange(stride, dtype=np.uint32)[None, :] < ref_len[:, None]
        want = np.where(mask, bits, 0).astype(np.uint64).sum(axis=1)
        bad_rows = np.nonzero(want != sums[first:last])[0]
        assert len(bad_rows) == 0, (first, bad_rows[:8])
        checked += int(ref_len.astype(np.uint64).sum())
    print(f"\nfull parity: utterances {first_utt} .. {first_utt + n_utt}, {checked} samples, {n_voices} voice(s): "
          f"every per-utterance checksum equals the oracle's  [{kernel}]")
