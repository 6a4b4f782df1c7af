"""ctypes binding of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
NF = 8

PH_SILENCE, PH_STOP, PH_GLIDE, PH_A, PH_E = range(5)


class Array(C.Structure):
    _fields_ = [("v", C.c_float * NF)]


class SynthesisElem(C.Structure):
    _fields_ = [
        ("frequency", C.c_float),
        ("formant_freq", Array),
        ("formant_bw", Array),
        ("formant_smooth", Array),
        ("formant_breath", Array),
        ("formant_turb", Array),
        ("formant_amp", Array),
    ]

    def as_np(self):
        return np.frombuffer(bytes(self), dtype=np.float32).copy()


class Voice(C.Structure):
    _fields_ = [
        ("sample_rate", C.c_float),
        ("phonemes", SynthesisElem * 2),
        ("center_frequency", C.c_float),
        ("jitter_frequency", C.c_float),
        ("jitter_delta_frequency", C.c_float),
        ("jitter_delta_formant_frequency", C.c_float),
        ("jitter_delta_amplitude", C.c_float),
    ]


class PhonemeElem(C.Structure):
    _fields_ = [
        ("phoneme", C.c_int32),
        ("length", C.c_float),
        ("blend_length", C.c_float),
        ("frequency", C.c_float),
    ]


class SequenceElem(C.Structure):
    _fields_ = [
        ("has_elem", C.c_int32),
        ("elem", SynthesisElem),
        ("length", C.c_float),
        ("blend_length", C.c_float),
    ]


class Rule(C.Structure):
    _fields_ = [
        ("string", C.POINTER(C.c_uint32)),
        ("string_len", C.c_uint32),
        ("phonemes", C.POINTER(C.c_int32)),
        ("n_phonemes", C.c_uint32),
    ]


PHONEME_DTYPE = np.dtype(
    [("phoneme", "<i4"), ("length", "<f4"), ("blend_length", "<f4"), ("frequency", "<f4")]
)

_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "liboracle.so"])


def lib():
    global _lib
    if _lib is not None:
        return _lib
    path = os.path.join(ORACLE_DIR, "liboracle.so")
    if not os.path.exists(path):
        build()
    L = C.CDLL(path)
    L.orc_random_f32.restype = C.c_float
    L.orc_random_f32.argtypes = [C.POINTER(C.c_uint32)]
    L.orc_tan_approx.restype = C.c_float
    L.orc_tan_approx.argtypes = [C.c_float]
    L.orc_exp_approx.restype = C.c_float
    L.orc_exp_approx.argtypes = [C.c_float]
    L.orc_array_sum.restype = C.c_float
    L.orc_array_sum.argtypes = [C.POINTER(Array)]
    L.orc_elem_silent.argtypes = [C.POINTER(SynthesisElem)]
    L.orc_elem_resample.argtypes = [C.POINTER(SynthesisElem), C.c_float, C.c_float]
    L.orc_elem_blend.argtypes = [C.POINTER(SynthesisElem)] * 3 + [C.c_float]
    L.orc_voice_generic.argtypes = [C.POINTER(Voice)]
    L.orc_voice_generic_at.argtypes = [C.POINTER(Voice), C.c_float]
    L.orc_synthesize_phonemes.restype = C.c_uint64
    L.orc_synthesize_phonemes.argtypes = [
        C.POINTER(Voice), C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint64]
    L.orc_synthesize_sequence.restype = C.c_uint64
    L.orc_synthesize_sequence.argtypes = [
        C.POINTER(Voice), C.POINTER(SequenceElem), C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint64]
    L.orc_trace_elems.restype = C.c_uint64
    L.orc_trace_elems.argtypes = [
        C.POINTER(Voice), C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p, C.c_uint64]
    L.orc_synthesize_batch.restype = None
    L.orc_synthesize_batch.argtypes = [
        C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
        C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p]
    L.orc_transcribe.restype = C.c_uint32
    L.orc_transcribe.argtypes = [
        C.POINTER(C.c_uint32), C.c_uint32, C.POINTER(Rule), C.c_uint32, C.c_int, C.c_int,
        C.POINTER(C.c_int32), C.c_uint32]
    L.orc_language_generic.restype = C.c_uint32
    L.orc_language_generic.argtypes = [C.POINTER(C.POINTER(Rule)), C.POINTER(C.c_int)]
    L.orc_intonate.argtypes = [C.POINTER(Voice), C.POINTER(C.c_int32), C.c_uint32, C.c_void_p]
    L.orc_say.restype = C.c_uint64
    L.orc_say.argtypes = [
        C.POINTER(Voice), C.POINTER(C.c_uint32), C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint64]
    L.orc_pcm16.restype = C.c_int16
    L.orc_pcm16.argtypes = [C.c_float]
    _lib = L
    return L


def voice_generic(sample_rate=None):
    v = Voice()
    if sample_rate is None:
        lib().orc_voice_generic(C.byref(v))
    else:
        lib().orc_voice_generic_at(C.byref(v), C.c_float(sample_rate))
    return v


def voices_array(voices):
    arr = (Voice * len(voices))()
    for i, v in enumerate(voices):
        C.memmove(C.byref(arr[i]), C.byref(v), C.sizeof(Voice))
    return arr


def segments(seq):
    """seq: iterable of (phoneme, length, blend_length, frequency) -> structured ndarray."""
    a = np.zeros(len(seq), dtype=PHONEME_DTYPE)
    for i, s in enumerate(seq):
        a[i] = tuple(s)
    return a


def synthesize_phonemes(voice, segs, jitter_seed=0, cap=None):
    segs = np.ascontiguousarray(segs, dtype=PHONEME_DTYPE)
    L = lib()
    if cap is None:
        cap = int(L.orc_synthesize_phonemes(C.byref(voice), segs.ctypes.data, len(segs),
                                            jitter_seed, None, 0))
    out = np.zeros(max(cap, 1), dtype=np.float32)
    n = int(L.orc_synthesize_phonemes(C.byref(voice), segs.ctypes.data, len(segs),
                                      jitter_seed, out.ctypes.data, cap))
    return out[: min(n, cap)], n


def synthesize_sequence(voice, seq_elems, jitter_seed=0):
    L = lib()
    arr = (SequenceElem * max(len(seq_elems), 1))(*seq_elems)
    n = int(L.orc_synthesize_sequence(C.byref(voice), arr, len(seq_elems), jitter_seed, None, 0))
    out = np.zeros(max(n, 1), dtype=np.float32)
    L.orc_synthesize_sequence(C.byref(voice), arr, len(seq_elems), jitter_seed,
                              out.ctypes.data, n)
    return out[:n]


def trace_elems(voice, segs, jitter_seed, stage):
    segs = np.ascontiguousarray(segs, dtype=PHONEME_DTYPE)
    L = lib()
    n = int(L.orc_trace_elems(C.byref(voice), segs.ctypes.data, len(segs), jitter_seed, stage,
                              None, 0))
    out = np.zeros((max(n, 1), 49), dtype=np.float32)
    L.orc_trace_elems(C.byref(voice), segs.ctypes.data, len(segs), jitter_seed, stage,
                      out.ctypes.data, n)
    return out[:n]


def _zeros_if_none(a, n):
    return np.zeros(n, dtype=np.uint32) if a is None else a


def synthesize_batch(voices, segs, seg_offsets, voice_ids, jitter_seeds, out_stride):
    """voices: list of Voice.  Returns (out[n_utt, out_stride], out_len[n_utt])."""
    L = lib()
    varr = voices_array(voices)
    segs = np.ascontiguousarray(segs, dtype=PHONEME_DTYPE)
    seg_offsets = np.ascontiguousarray(seg_offsets, dtype=np.uint32)
    n_utt = len(seg_offsets) - 1
    voice_ids = np.ascontiguousarray(_zeros_if_none(voice_ids, n_utt), dtype=np.uint32)
    jitter_seeds = np.ascontiguousarray(_zeros_if_none(jitter_seeds, n_utt), dtype=np.uint32)
    out = np.zeros((n_utt, out_stride), dtype=np.float32)
    out_len = np.zeros(n_utt, dtype=np.uint32)
    L.orc_synthesize_batch(C.cast(varr, C.c_void_p), len(voices), segs.ctypes.data,
                           seg_offsets.ctypes.data, voice_ids.ctypes.data,
                           jitter_seeds.ctypes.data, n_utt, out.ctypes.data, out_stride,
                           out_len.ctypes.data)
    return out, out_len


def synthesize_batch_threads(voices, segs, seg_offsets, voice_ids, jitter_seeds, out_stride,
                             n_threads, keep_output=True):
    """synthesize_batch over n_threads pthreads (bench baseline).  keep_output=False only counts
    (no PCM buffer), keep_output=True returns (out, out_len)."""
    L = lib()
    varr = voices_array(voices)
    segs = np.ascontiguousarray(segs, dtype=PHONEME_DTYPE)
    seg_offsets = np.ascontiguousarray(seg_offsets, dtype=np.uint32)
    n_utt = len(seg_offsets) - 1
    voice_ids = np.ascontiguousarray(_zeros_if_none(voice_ids, n_utt), dtype=np.uint32)
    jitter_seeds = np.ascontiguousarray(_zeros_if_none(jitter_seeds, n_utt), dtype=np.uint32)
    out = np.zeros((n_utt, out_stride), dtype=np.float32) if keep_output else None
    out_len = np.zeros(n_utt, dtype=np.uint32)
    L.orc_synthesize_batch_threads.restype = C.c_int
    L.orc_synthesize_batch_threads.argtypes = [
        C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32,
        C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32]
    started = L.orc_synthesize_batch_threads(
        C.cast(varr, C.c_void_p), len(voices), segs.ctypes.data, seg_offsets.ctypes.data,
        voice_ids.ctypes.data, jitter_seeds.ctypes.data, n_utt,
        out.ctypes.data if keep_output else None, out_stride, out_len.ctypes.data, n_threads)
    return out, out_len, started


def count_batch(voices, segs, seg_offsets, voice_ids, jitter_seeds):
    L = lib()
    varr = voices_array(voices)
    segs = np.ascontiguousarray(segs, dtype=PHONEME_DTYPE)
    seg_offsets = np.ascontiguousarray(seg_offsets, dtype=np.uint32)
    n_utt = len(seg_offsets) - 1
    voice_ids = np.ascontiguousarray(_zeros_if_none(voice_ids, n_utt), dtype=np.uint32)
    jitter_seeds = np.ascontiguousarray(_zeros_if_none(jitter_seeds, n_utt), dtype=np.uint32)
    out_len = np.zeros(n_utt, dtype=np.uint32)
    L.orc_synthesize_batch(C.cast(varr, C.c_void_p), len(voices), segs.ctypes.data,
                           seg_offsets.ctypes.data, voice_ids.ctypes.data,
                           jitter_seeds.ctypes.data, n_utt, None, 0, out_len.ctypes.data)
    return out_len


def make_rules(rule_list):
    """rule_list: [(string, [phonemes...]), ...] -> (Rule array, keepalive)."""
    keep = []
    arr = (Rule * len(rule_list))()
    for i, (s, ph) in enumerate(rule_list):
        cps = (C.c_uint32 * max(len(s), 1))(*[ord(ch) for ch in s])
        pp = (C.c_int32 * max(len(ph), 1))(*ph)
        keep += [cps, pp]
        arr[i].string = C.cast(cps, C.POINTER(C.c_uint32))
        arr[i].string_len = len(s)
        arr[i].phonemes = C.cast(pp, C.POINTER(C.c_int32))
        arr[i].n_phonemes = len(ph)
    return arr, keep


def transcribe(text, rule_list, case_sensitive=False, leading_silence=False):
    arr, keep = make_rules(rule_list)
    cps = (C.c_uint32 * max(len(text), 1))(*[ord(ch) for ch in text])
    cap = 4 * len(text) + 8
    out = (C.c_int32 * cap)()
    n = lib().orc_transcribe(cps, len(text), arr, len(rule_list), int(case_sensitive),
                             int(leading_silence), out, cap)
    return list(out[:n])


def say(voice, text, jitter_seed=0):
    cps = (C.c_uint32 * max(len(text), 1))(*[ord(ch) for ch in text])
    n = int(lib().orc_say(C.byref(voice), cps, len(text), jitter_seed, None, 0))
    out = np.zeros(max(n, 1), dtype=np.float32)
    lib().orc_say(C.byref(voice), cps, len(text), jitter_seed, out.ctypes.data, n)
    return out[:n]


def set_precise(on):
    """Per-formant arithmetic of the oracle in double precision (same f32 parameter track): the
    yardstick for the fast mode's tolerance — NOT the parity target."""
    L = lib()
    L.orc_set_precise.argtypes = [C.c_int]
    L.orc_set_precise.restype = None
    L.orc_set_precise(int(on))       # (True -> 1; 2 / 3: the middle-tier experiment's modes, oracle/grail_oracle.c)
