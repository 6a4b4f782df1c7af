#!/usr/bin/env python3
"""Regenerates tests/golden/hotpath_v1.npz from the CPU oracle (oracle/liboracle.so).

The reference is Rust and cannot run in this image (no rustc/cargo), and its own tests for
this path are empty, so these vectors are produced by the oracle — itself pinned by the
known answers in tests/test_oracle_kat.py and by the independent numpy model
(tests/test_oracle_crosscheck.py).  They freeze today's bits so that any later change to
oracle, host algebra or kernels that moves a single bit is caught on CPU and GPU alike.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as O  # noqa: E402

A, E, S, ST, GL = O.PH_A, O.PH_E, O.PH_SILENCE, O.PH_STOP, O.PH_GLIDE


def cases():
    f44 = float(np.float32(120.0) / np.float32(44100.0))
    f48 = float(np.float32(120.0) / np.float32(48000.0))
    return [
        # name, sample rate (None = 44100 generic()), segments, jitter seed
        ("text_a_head", None, [(S, .5, .5, f44), (A, .5, .5, f44)], 0),  # config 1, first 2048 samples
        ("a_e_48k", 48000.0, [(A, .01, .01, f48), (E, .01, .01, f48)], 1),
        ("fade_in_out_48k", 48000.0, [(S, .008, .008, f48), (E, .008, .008, f48)], 2),
        ("mixed_44k", None, [(E, .006, .003, 0.004), (S, .004, .004, 0.1), (A, .006, .012, 0.002),
                             (GL, .002, .002, 0.1), (E, .005, .005, 0.003)], 12345),
        ("pitch_clamp_48k", 48000.0, [(A, .004, .004, 0.7), (E, .004, .004, 0.5)], 3),
        ("wrap_48k", 48000.0, [(A, .07, .07, f48), (E, .07, .07, 0.0031)], 4242),  # jitter wraps twice
    ]


def main():
    out = {}
    for name, rate, segs, seed in cases():
        v = O.voice_generic(rate)
        sa = O.segments(segs)
        pcm, n = O.synthesize_phonemes(v, sa, seed)
        keep = min(n, 2048) if name != "wrap_48k" else n
        out[name + "/rate"] = np.float32(0.0 if rate is None else rate)
        out[name + "/segs"] = sa
        out[name + "/seed"] = np.uint32(seed)
        out[name + "/len"] = np.uint32(n)
        out[name + "/pcm"] = pcm[:keep]
        # a checksum of the whole track: sum of bit patterns mod 2^64
        out[name + "/sum"] = np.uint64(pcm.view(np.uint32).astype(np.uint64).sum())
    for rate in (None, 48000.0):
        v = O.voice_generic(rate)
        out["voice_%s" % ("44k" if rate is None else "48k")] = np.frombuffer(bytes(v), dtype=np.float32).copy()
    np.savez_compressed(os.path.join(HERE, "hotpath_v1.npz"), **out)
    print("wrote", os.path.join(HERE, "hotpath_v1.npz"), os.path.getsize(os.path.join(HERE, "hotpath_v1.npz")), "bytes")


if __name__ == "__main__":
    main()
