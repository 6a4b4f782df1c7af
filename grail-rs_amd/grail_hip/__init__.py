"""ctypes binding of libgrail_hip.so — the C ABI declared in include/grail_hip.h.

This is plumbing for tests and bench.py (the reference's own host language,
Rust, is not in this image; INTEGRATION.md has the Rust binding).  It holds no
arithmetic: every sample comes from the HIP kernels behind the C ABI, and
anything that needs the GPU raises GrailError when the library or a device is
missing — there is no CPU fallback.

Reference API mirrored (reference file:line):
  Voice src/lib.rs:696, SynthesisElem :316, PhonemeElem :961, SequenceElem :814,
  Phoneme :632, voices::generic() src/voices/generic.rs:5,
  .select().sequence().jitter().synthesize() src/lib.rs:1013/941/786/587.
"""
import ctypes as C
import os

import numpy as np

NUM_FORMANTS = 8
DEFAULT_SAMPLE_RATE = 44100.0

PH_SILENCE, PH_STOP, PH_GLIDE, PH_A, PH_E = range(5)
PH_COUNT = 5
NUM_VOICED = 2

OK = 0
ERR_INVALID_ARG = -1
ERR_NO_DEVICE = -2
ERR_HIP = -3
ERR_BUFFER_TOO_SMALL = -4
ERR_OUT_OF_MEMORY = -5
ERR_RCCL = -6
ERR_NO_VOICES = -7

OUT_HOST = 0
OUT_DEVICE = 1
# "arithmetic" = 1 (fast mode): GRAIL_FAST_TOLERANCE of include/grail_hip.h
FAST_TOLERANCE_ULPS = 64
FAST_TOLERANCE = FAST_TOLERANCE_ULPS * 2.0 ** -23
FAST_SHARPNESS_LIMIT = 28.0      # GRAIL_FAST_SHARPNESS_LIMIT
FAST_TOLERANCE_NOTE = (f"fast mode: max |fast - exact| <= {FAST_TOLERANCE_ULPS} * 2^-23 = {FAST_TOLERANCE:.3g} of "
                       "full scale vs the oracle (tests/test_fast_gpu.py asserts it on configs 2, 3, 4 "
                       "and a fuzz corpus); clock, phases, wraps and LCGs stay exact")
UNIQUE_ID_BYTES = 128
ABI_VERSION = 4                  # GRAIL_ABI_VERSION of the header this binding mirrors

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GRAIL_HIP_LIB") or os.path.join(os.path.dirname(_HERE), "lib",
                                                         "libgrail_hip.so")

# every symbol include/grail_hip.h declares (checked by tests/test_abi.py)
EXPORTS = [
    "grail_abi_version", "grail_status_string", "grail_last_error",
    "grail_elem_silent", "grail_elem_new_phoneme", "grail_elem_new", "grail_elem_resample",
    "grail_elem_blend", "grail_voice_generic", "grail_voice_generic_at", "grail_voice_get",
    "grail_create", "grail_destroy", "grail_device_count", "grail_device_pci_bus_id", "grail_time_split_warmup", "grail_length_bound",
    "grail_time_split_grid", "grail_fast_sharpness", "grail_plan_blocks", "grail_plan_ragged_blocks", "grail_dispatch_model",
    "grail_packed_launch_order", "grail_set_voices",
    "grail_get_voices", "grail_set_option", "grail_get_option",
    "grail_batch_upload", "grail_batch_upload_elems", "grail_batch_free", "grail_batch_size",
    "grail_batch_lengths", "grail_batch_synthesize_async", "grail_sync",
    "grail_last_kernel_ms", "grail_last_kernel_name", "grail_synthesize_batch", "grail_synthesize_batch_elems",
    "grail_stream_open", "grail_stream_next_async", "grail_stream_close",
    "grail_stream_open_live", "grail_stream_append", "grail_stream_append_elems", "grail_stream_finish", "grail_stream_pending",
    "grail_language_generic", "grail_transcribe", "grail_intonate", "grail_text_to_phoneme_elems",
    "grail_synthesize_batch_pcm16", "grail_batch_synthesize_pcm16_async", "grail_stream_next_pcm16_async", "grail_say_batch", "grail_pcm16_async", "grail_batch_digest", "grail_batch_compare", "grail_wav_write_i16",
    "grail_device_alloc", "grail_device_free", "grail_host_alloc", "grail_host_free", "grail_memcpy_d2h", "grail_memcpy_h2d",
    "grail_memset_d", "grail_shard_range", "grail_comm_unique_id", "grail_comm_init",
    "grail_broadcast_voices", "grail_comm_info", "grail_comm_destroy",
    "grail_node_create", "grail_node_destroy", "grail_node_size", "grail_node_context", "grail_node_set_voices",
    "grail_node_set_option", "grail_node_get_option", "grail_node_shard_of", "grail_node_synthesize_batch",
    "grail_node_synthesize_batch_elems", "grail_node_synthesize_batch_pcm16", "grail_node_synthesize_batch_device",
    "grail_node_say_batch",
    "grail_node_lengths", "grail_node_last_shard_ms", "grail_node_host_alloc", "grail_node_host_free",
]


class GrailError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"grail_hip status {status}: {message}")
        self.status = status


class SynthesisElem(C.Structure):
    _fields_ = [
        ("frequency", C.c_float),
        ("formant_freq", C.c_float * NUM_FORMANTS),
        ("formant_bw", C.c_float * NUM_FORMANTS),
        ("formant_smooth", C.c_float * NUM_FORMANTS),
        ("formant_breath", C.c_float * NUM_FORMANTS),
        ("formant_turb", C.c_float * NUM_FORMANTS),
        ("formant_amp", C.c_float * NUM_FORMANTS),
    ]

    def as_np(self):
        return np.frombuffer(bytes(self), dtype=np.float32).copy()

    @classmethod
    def from_np(cls, a):
        a = np.ascontiguousarray(a, dtype=np.float32)
        assert a.size == 49
        return cls.from_buffer_copy(a.tobytes())


class Voice(C.Structure):
    _fields_ = [
        ("sample_rate", C.c_float),
        ("phonemes", SynthesisElem * NUM_VOICED),
        ("center_frequency", C.c_float),
        ("jitter_frequency", C.c_float),
        ("jitter_delta_frequency", C.c_float),
        ("jitter_delta_formant_frequency", C.c_float),
        ("jitter_delta_amplitude", C.c_float),
    ]

    def copy(self):
        return Voice.from_buffer_copy(bytes(self))


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


class PlanBlock(C.Structure):
    """grail_plan_block: one kernel launch of a batch's plan (grail_plan_blocks)."""
    _fields_ = [
        ("rows", C.c_uint32),
        ("lanes_per_utterance", C.c_uint32),
        ("pipelined", C.c_uint32),
        ("chunks", C.c_uint32),
        ("scan", C.c_uint32),
        ("fast", C.c_uint32),
        ("formants", C.c_uint32),
        ("model_ms", C.c_float),
    ]

    def family(self):
        if self.scan:
            return f"scan{self.scan + 1}"
        if self.chunks:
            return f"split{self.chunks}"
        if self.pipelined:
            return f"pipe{self.formants}r{16 * self.pipelined}"
        return ("fast" if self.fast else "exact") + f"L{self.lanes_per_utterance}"


class NodeShard(C.Structure):
    """grail_node_shard: the rows and segments one device of a node renders (grail_node_shard_of)."""
    _fields_ = [
        ("first_row", C.c_uint64),
        ("rows", C.c_uint64),
        ("first_seg", C.c_uint32),
        ("n_segs", C.c_uint32),
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


def lib_exists():
    return os.path.exists(LIB_PATH)


def load():
    """Load libgrail_hip.so; fails loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GrailError(ERR_NO_DEVICE, f"{LIB_PATH} is missing: run __graft_entry__.build() "
                                        "(there is no CPU fallback)")
    L = C.CDLL(LIB_PATH)
    vp, u32p, u64 = C.c_void_p, C.POINTER(C.c_uint32), C.c_uint64
    L.grail_abi_version.restype = C.c_int
    if L.grail_abi_version() != ABI_VERSION:     # before any other symbol is touched
        raise GrailError(ERR_INVALID_ARG, f"{LIB_PATH} has ABI version {L.grail_abi_version()}, this binding "
                                          f"mirrors version {ABI_VERSION} of include/grail_hip.h: rebuild the library")
    L.grail_status_string.restype = C.c_char_p
    L.grail_status_string.argtypes = [C.c_int]
    L.grail_last_error.restype = C.c_char_p
    L.grail_elem_silent.restype = None
    L.grail_elem_silent.argtypes = [C.POINTER(SynthesisElem)]
    fp = C.POINTER(C.c_float)
    L.grail_elem_new_phoneme.restype = None
    L.grail_elem_new_phoneme.argtypes = [C.POINTER(SynthesisElem)] + [fp] * 6
    L.grail_elem_new.restype = None
    L.grail_elem_new.argtypes = [C.POINTER(SynthesisElem), C.c_float, C.c_float] + [fp] * 6
    L.grail_elem_resample.restype = None
    L.grail_elem_resample.argtypes = [C.POINTER(SynthesisElem), C.c_float, C.c_float]
    L.grail_elem_blend.restype = None
    L.grail_elem_blend.argtypes = [C.POINTER(SynthesisElem)] * 3 + [C.c_float]
    L.grail_voice_generic.restype = None
    L.grail_voice_generic.argtypes = [C.POINTER(Voice)]
    L.grail_voice_generic_at.restype = None
    L.grail_voice_generic_at.argtypes = [C.POINTER(Voice), C.c_float]
    L.grail_voice_get.argtypes = [C.POINTER(Voice), C.c_int32, C.POINTER(SynthesisElem)]
    L.grail_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.grail_destroy.argtypes = [vp]
    L.grail_device_count.argtypes = [C.POINTER(C.c_int)]
    L.grail_device_pci_bus_id.argtypes = [vp, C.c_char_p, C.c_size_t]
    L.grail_length_bound.argtypes = [C.POINTER(C.c_float), C.c_uint32, C.c_float]
    L.grail_length_bound.restype = C.c_uint64
    L.grail_time_split_warmup.argtypes = [C.POINTER(Voice)]
    L.grail_time_split_warmup.restype = C.c_uint32
    L.grail_fast_sharpness.argtypes = [C.POINTER(Voice)]
    L.grail_fast_sharpness.restype = C.c_float
    L.grail_time_split_grid.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]
    L.grail_plan_blocks.argtypes = [C.c_uint32, C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32,
                                    C.POINTER(PlanBlock), C.c_uint32, u32p]
    L.grail_plan_ragged_blocks.argtypes = [C.c_uint32, C.c_int, C.c_int, C.c_uint32, C.c_uint32, u32p, u32p, u32p,
                                           C.POINTER(PlanBlock), C.c_uint32, u32p]
    L.grail_dispatch_model.argtypes = [C.c_uint32, C.c_uint32, C.POINTER(C.c_double), u32p, C.c_uint32, C.POINTER(C.c_double)]
    L.grail_packed_launch_order.argtypes = [C.c_uint32, C.c_uint32, C.POINTER(C.c_double), C.c_uint32, u32p]
    L.grail_set_voices.argtypes = [vp, vp, C.c_uint32]
    L.grail_get_voices.argtypes = [vp, vp, C.c_uint32, u32p]
    L.grail_set_option.argtypes = [vp, C.c_char_p, C.c_int64]
    L.grail_get_option.argtypes = [vp, C.c_char_p, C.POINTER(C.c_int64)]
    L.grail_batch_upload.argtypes = [vp, vp, vp, vp, vp, C.c_uint32, C.POINTER(vp)]
    L.grail_batch_upload_elems.argtypes = [vp, vp, vp, vp, vp, C.c_uint32, C.POINTER(vp)]
    L.grail_batch_free.argtypes = [vp, vp]
    L.grail_batch_size.restype = C.c_uint32
    L.grail_batch_size.argtypes = [vp]
    L.grail_batch_lengths.argtypes = [vp, vp, C.c_uint32, vp]
    L.grail_batch_synthesize_async.argtypes = [vp, vp, vp, u64, vp]
    L.grail_sync.argtypes = [vp]
    L.grail_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_float)]
    L.grail_last_kernel_name.restype = C.c_char_p
    L.grail_last_kernel_name.argtypes = [vp]
    L.grail_synthesize_batch.argtypes = [vp, vp, vp, vp, vp, C.c_uint32, vp, u64, vp, C.c_uint32]
    L.grail_synthesize_batch_elems.argtypes = [vp, vp, vp, vp, vp, C.c_uint32, vp, u64, vp,
                                               C.c_uint32]
    L.grail_stream_open.argtypes = [vp, vp, C.POINTER(vp)]
    L.grail_stream_next_async.argtypes = [vp, vp, C.c_uint32, vp, u64, vp]
    L.grail_stream_next_pcm16_async.argtypes = [vp, vp, C.c_uint32, vp, u64, vp]
    L.grail_stream_close.argtypes = [vp, vp]
    L.grail_stream_open_live.argtypes = [vp, C.c_uint32, vp, vp, C.c_uint32, C.c_int, C.POINTER(vp)]
    L.grail_stream_append.argtypes = [vp, vp, vp, vp]
    L.grail_stream_append_elems.argtypes = [vp, vp, vp, vp]
    L.grail_stream_finish.argtypes = [vp, vp, vp]
    L.grail_stream_pending.argtypes = [vp, vp, vp]
    L.grail_language_generic.restype = C.c_uint32
    L.grail_language_generic.argtypes = [C.POINTER(C.POINTER(Rule)), C.POINTER(C.c_int)]
    L.grail_transcribe.argtypes = [C.POINTER(C.c_uint32), C.c_uint32, C.POINTER(Rule), C.c_uint32,
                                   C.c_int, C.c_int, C.POINTER(C.c_int32), C.c_uint32, u32p]
    L.grail_intonate.argtypes = [C.POINTER(Voice), C.POINTER(C.c_int32), C.c_uint32, vp]
    L.grail_text_to_phoneme_elems.argtypes = [C.POINTER(Voice), C.c_char_p, vp, C.c_uint32, u32p]
    L.grail_say_batch.argtypes = [vp, C.POINTER(C.c_char_p), C.c_uint32, vp, vp, vp, u64, vp,
                                  C.c_uint32]
    L.grail_pcm16_async.argtypes = [vp, vp, u64, vp, C.c_uint32, C.c_uint32, vp, u64]
    L.grail_synthesize_batch_pcm16.argtypes = [vp, vp, vp, vp, vp, C.c_uint32, vp, u64, vp,
                                               C.c_uint32]
    L.grail_batch_synthesize_pcm16_async.argtypes = [vp, vp, vp, u64, vp]
    L.grail_batch_digest.argtypes = [vp, vp, u64, vp, C.c_uint32, vp, vp, vp]
    L.grail_batch_compare.argtypes = [vp, vp, vp, u64, vp, vp, C.c_uint32, vp, vp, vp]
    L.grail_wav_write_i16.argtypes = [C.c_char_p, vp, C.c_uint32, C.c_uint32]
    L.grail_device_alloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
    L.grail_device_free.argtypes = [vp, vp]
    L.grail_host_alloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
    L.grail_host_free.argtypes = [vp, vp]
    L.grail_memcpy_d2h.argtypes = [vp, vp, vp, C.c_size_t]
    L.grail_memcpy_h2d.argtypes = [vp, vp, vp, C.c_size_t]
    L.grail_memset_d.argtypes = [vp, vp, C.c_int, C.c_size_t]
    L.grail_shard_range.restype = None
    L.grail_shard_range.argtypes = [u64, C.c_uint32, C.c_uint32, C.POINTER(u64), C.POINTER(u64)]
    L.grail_comm_unique_id.argtypes = [vp]
    L.grail_comm_init.argtypes = [vp, vp, C.c_uint32, C.c_uint32]
    L.grail_broadcast_voices.argtypes = [vp, C.c_uint32, C.c_uint32]
    L.grail_comm_info.argtypes = [vp, u32p, u32p]
    L.grail_comm_destroy.argtypes = [vp]
    L.grail_node_create.argtypes = [C.POINTER(C.c_int), C.c_uint32, C.POINTER(vp)]
    L.grail_node_destroy.argtypes = [vp]
    L.grail_node_size.restype = C.c_uint32
    L.grail_node_size.argtypes = [vp]
    L.grail_node_context.argtypes = [vp, C.c_uint32, C.POINTER(vp)]
    L.grail_node_set_voices.argtypes = [vp, vp, C.c_uint32]
    L.grail_node_set_option.argtypes = [vp, C.c_char_p, C.c_int64]
    L.grail_node_get_option.argtypes = [vp, C.c_char_p, C.POINTER(C.c_int64)]
    L.grail_node_shard_of.argtypes = [u32p, u64, C.c_uint32, C.c_uint32, C.POINTER(NodeShard), u32p, u64]
    L.grail_node_synthesize_batch.argtypes = [vp, vp, vp, vp, vp, C.c_uint32, vp, u64, vp, C.c_uint32]
    L.grail_node_synthesize_batch_elems.argtypes = [vp, vp, vp, vp, vp, C.c_uint32, vp, u64, vp, C.c_uint32]
    L.grail_node_synthesize_batch_pcm16.argtypes = [vp, vp, vp, vp, vp, C.c_uint32, vp, u64, vp, C.c_uint32]
    L.grail_node_synthesize_batch_device.argtypes = [vp, vp, vp, vp, vp, C.c_uint32, C.POINTER(vp), u64, vp]
    L.grail_node_say_batch.argtypes = [vp, C.POINTER(C.c_char_p), C.c_uint32, vp, vp, vp, u64, vp, C.c_uint32]
    L.grail_node_lengths.argtypes = [vp, vp, vp, vp, C.c_uint32, C.c_uint32, vp]
    L.grail_node_last_shard_ms.argtypes = [vp, C.POINTER(C.c_float), C.c_uint32]
    L.grail_node_host_alloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
    L.grail_node_host_free.argtypes = [vp, vp]
    _lib = L
    return L


def _check(status):
    if status != OK:
        raise GrailError(status, load().grail_last_error().decode() or
                         load().grail_status_string(status).decode())


# ---- host-side parameter algebra -------------------------------------------
def voice_generic(sample_rate=None):
    """voices::generic() (src/voices/generic.rs:5); sample_rate != None gives the
    resampled variant of SURVEY.md §8d."""
    v = Voice()
    if sample_rate is None:
        load().grail_voice_generic(C.byref(v))
    else:
        load().grail_voice_generic_at(C.byref(v), C.c_float(sample_rate))
    return v


def elem_new_phoneme(freq, bw, smooth, turb, breath, amp):
    """SynthesisElem::new_phoneme == MKPHON (src/lib.rs:381, src/voices/mod.rs:7)."""
    e = SynthesisElem()
    arrs = [(C.c_float * NUM_FORMANTS)(*[float(x) for x in a])
            for a in (freq, bw, smooth, turb, breath, amp)]
    load().grail_elem_new_phoneme(C.byref(e), *arrs)
    return e


def elem_resample(elem, old_rate, new_rate):
    e = SynthesisElem.from_buffer_copy(bytes(elem))
    load().grail_elem_resample(C.byref(e), C.c_float(old_rate), C.c_float(new_rate))
    return e


def elem_silent():
    e = SynthesisElem()
    load().grail_elem_silent(C.byref(e))
    return e


def elem_blend(a, b, alpha):
    e = SynthesisElem()
    load().grail_elem_blend(C.byref(e), C.byref(a), C.byref(b), C.c_float(alpha))
    return e


def shard_range(n_utt, rank, world):
    b, e = C.c_uint64(), C.c_uint64()
    load().grail_shard_range(n_utt, rank, world, C.byref(b), C.byref(e))
    return b.value, e.value


def node_shard_of(seg_offsets, index, n_devices):
    """(NodeShard, rebased seg_offsets) of device slot `index` out of n_devices: the view a node call hands to that
    device's grail_synthesize_batch (grail_node_shard_of; pure host arithmetic, no GPU)."""
    offs = np.ascontiguousarray(seg_offsets, dtype=np.uint32)
    n_utt = len(offs) - 1
    u32p = C.POINTER(C.c_uint32)
    sh = NodeShard()
    _check(load().grail_node_shard_of(offs.ctypes.data_as(u32p), n_utt, index, n_devices, C.byref(sh), None, 0))
    rebased = np.zeros(sh.rows + 1, dtype=np.uint32)
    _check(load().grail_node_shard_of(offs.ctypes.data_as(u32p), n_utt, index, n_devices, C.byref(sh),
                                      rebased.ctypes.data_as(u32p), len(rebased)))
    return sh, rebased


def voices_blob(voices):
    """The broadcast payload: the raw grail_voice[] bytes."""
    return b"".join(bytes(v) for v in voices)


def voices_from_blob(blob):
    n = len(blob) // C.sizeof(Voice)
    assert n * C.sizeof(Voice) == len(blob)
    return [Voice.from_buffer_copy(blob[i * C.sizeof(Voice):(i + 1) * C.sizeof(Voice)])
            for i in range(n)]


def segments(seq):
    a = np.zeros(len(seq), dtype=PHONEME_DTYPE)
    for i, s in enumerate(seq):
        a[i] = tuple(s)
    return a


# ---- text front half ---------------------------------------------------------
def make_rules(rule_list):
    """[(string, [phonemes...]), ...] -> (Rule array, keepalive list)."""
    keep = []
    arr = (Rule * len(rule_list))()
    for i, (st, ph) in enumerate(rule_list):
        cps = (C.c_uint32 * max(len(st), 1))(*[ord(ch) for ch in st])
        pp = (C.c_int32 * max(len(ph), 1))(*ph)
        keep += [cps, pp]
        arr[i].string = C.cast(cps, C.POINTER(C.c_uint32))
        arr[i].string_len = len(st)
        arr[i].phonemes = C.cast(pp, C.POINTER(C.c_int32))
        arr[i].n_phonemes = len(ph)
    return arr, keep


def transcribe(text, rule_list, case_sensitive=False, leading_silence=False):
    """Transcriber (src/lib.rs:1116); leading_silence=True is .transcribe() (:1201)."""
    arr, keep = make_rules(rule_list)
    cps = (C.c_uint32 * max(len(text), 1))(*[ord(ch) for ch in text])
    cap = 4 * len(text) + 8
    out = (C.c_int32 * cap)()
    n = C.c_uint32()
    _check(load().grail_transcribe(cps, len(text), arr, len(rule_list), int(case_sensitive),
                                   int(leading_silence), out, cap, C.byref(n)))
    return list(out[: n.value])


def language_generic():
    """languages::generic() as [(string, [phonemes])], case_sensitive."""
    rules = C.POINTER(Rule)()
    cs = C.c_int()
    n = load().grail_language_generic(C.byref(rules), C.byref(cs))
    out = []
    for i in range(n):
        r = rules[i]
        out.append(("".join(chr(r.string[k]) for k in range(r.string_len)),
                    [r.phonemes[k] for k in range(r.n_phonemes)]))
    return out, bool(cs.value)


def text_to_phoneme_elems(voice, text):
    n = C.c_uint32()
    _check(load().grail_text_to_phoneme_elems(C.byref(voice), text.encode("utf-8"), None, 0,
                                              C.byref(n)))
    a = np.zeros(max(n.value, 1), dtype=PHONEME_DTYPE)
    _check(load().grail_text_to_phoneme_elems(C.byref(voice), text.encode("utf-8"), a.ctypes.data,
                                              n.value, C.byref(n)))
    return a[: n.value]


def wav_write_i16(path, pcm, sample_rate):
    pcm = np.ascontiguousarray(pcm, dtype=np.int16)
    _check(load().grail_wav_write_i16(path.encode(), pcm.ctypes.data, len(pcm), int(sample_rate)))


def length_bound(segment_lengths, sample_rate):
    """An upper bound of an utterance's length in samples (grail_length_bound); None: no bound."""
    a = np.ascontiguousarray(segment_lengths, dtype=np.float32)
    v = int(load().grail_length_bound(a.ctypes.data_as(C.POINTER(C.c_float)), len(a), float(sample_rate)))
    return None if v == 2 ** 64 - 1 else v


def time_split_warmup(voice):
    """Warm-up length of `voice` for the time-split fast kernels, in samples (0: does not qualify)."""
    return int(load().grail_time_split_warmup(C.byref(voice)))


def fast_sharpness(voice):
    """Predicted bound on |fast - reference| for `voice`, units of 2^-23 (served up to FAST_SHARPNESS_LIMIT)."""
    return float(load().grail_fast_sharpness(C.byref(voice)))


def time_split_grid(span_samples, warmup, chunks, ff_cost_permille=165):
    """Chunk starts of a time-split launch (GrailError when so many chunks do not fit)."""
    out = (C.c_uint32 * max(int(chunks), 1))()
    _check(load().grail_time_split_grid(span_samples, warmup, chunks, ff_cost_permille, out))
    return [int(x) for x in out]


def plan_blocks(rows, span_samples, arithmetic=0, live_formants=4, warmup=3904, compute_units=256):
    """How a batch of `rows` utterances (longest: span_samples) is cut into kernel launches: [PlanBlock]."""
    arr = (PlanBlock * 16)()
    n = C.c_uint32()
    _check(load().grail_plan_blocks(compute_units, arithmetic, live_formants, warmup, rows, span_samples, arr, 16,
                                    C.byref(n)))
    return [PlanBlock.from_buffer_copy(bytes(arr[i])) for i in range(min(n.value, 16))]


def plan_ragged_blocks(row_samples, row_segments=None, row_kinks=None, arithmetic=0, live_formants=4, warmup=3904,
                       compute_units=256):
    """... of a batch whose utterances differ in length: row_samples descending (launch order), the rows' segments and
    kinks of alpha (option "ragged_plan"): [PlanBlock]."""
    def u32(a):
        return None if a is None else np.ascontiguousarray(a, dtype=np.uint32)
    rs, sg, kk = u32(row_samples), u32(row_segments), u32(row_kinks)
    u32p = C.POINTER(C.c_uint32)
    arr = (PlanBlock * 16)()
    n = C.c_uint32()
    _check(load().grail_plan_ragged_blocks(compute_units, arithmetic, live_formants, warmup, len(rs),
                                           rs.ctypes.data_as(u32p), None if sg is None else sg.ctypes.data_as(u32p),
                                           None if kk is None else kk.ctypes.data_as(u32p), arr, 16, C.byref(n)))
    return [PlanBlock.from_buffer_copy(bytes(arr[i])) for i in range(min(n.value, 16))]


def dispatch_model(workgroup_ms, order=None, compute_units=256, waves_per_workgroup=1):
    """The makespan of workgroups that take workgroup_ms[b], launched in `order`, by the library's model of the workgroup
    dispatcher (grail_dispatch_model; pure host arithmetic)."""
    c = np.ascontiguousarray(workgroup_ms, dtype=np.float64)
    o = None if order is None else np.ascontiguousarray(order, dtype=np.uint32)
    out = C.c_double()
    _check(load().grail_dispatch_model(compute_units, waves_per_workgroup, c.ctypes.data_as(C.POINTER(C.c_double)),
                                       None if o is None else o.ctypes.data_as(C.POINTER(C.c_uint32)), len(c), C.byref(out)))
    return out.value


def packed_launch_order(workgroup_ms, compute_units=256, waves_per_workgroup=1):
    """order[position] = workgroup: what option "packed_launch_order" launches such workgroups in (grail_packed_launch_order)."""
    c = np.ascontiguousarray(workgroup_ms, dtype=np.float64)
    o = np.zeros(len(c), dtype=np.uint32)
    _check(load().grail_packed_launch_order(compute_units, waves_per_workgroup, c.ctypes.data_as(C.POINTER(C.c_double)), len(c),
                                            o.ctypes.data_as(C.POINTER(C.c_uint32))))
    return o


def device_count():
    n = C.c_int(0)
    st = load().grail_device_count(C.byref(n))
    return n.value if st == OK else 0


def _ptr(a):
    return None if a is None else a.ctypes.data


class Batch:
    def __init__(self, ctx, handle, n_utt):
        self.ctx, self.handle, self.n_utt = ctx, handle, n_utt

    def lengths(self, max_len=0xFFFFFFFF):
        out = np.zeros(max(self.n_utt, 1), dtype=np.uint32)
        _check(load().grail_batch_lengths(self.ctx.handle, self.handle, max_len, out.ctypes.data))
        return out[: self.n_utt]

    def synthesize_async(self, out_dev, out_stride, out_len_dev=None):
        _check(load().grail_batch_synthesize_async(self.ctx.handle, self.handle, out_dev,
                                                   out_stride, out_len_dev))

    def synthesize_pcm16_async(self, out_dev, out_stride, out_len_dev=None):
        """Rows of i16 PCM: the WAV sink's conversion fused into the kernel's store."""
        _check(load().grail_batch_synthesize_pcm16_async(self.ctx.handle, self.handle, out_dev,
                                                         out_stride, out_len_dev))

    def free(self):
        if self.handle:
            load().grail_batch_free(self.ctx.handle, self.handle)
            self.handle = None


class Stream:
    """grail_stream: resumable synthesis of a Batch (chunks of samples per call)."""

    def __init__(self, batch):
        self.batch, self.ctx = batch, batch.ctx
        h = C.c_void_p()
        _check(load().grail_stream_open(self.ctx.handle, batch.handle, C.byref(h)))
        self.handle = h

    def next_async(self, max_samples, out_dev, out_stride, out_len_dev=None):
        _check(load().grail_stream_next_async(self.ctx.handle, self.handle, max_samples, out_dev,
                                              out_stride, out_len_dev))

    def next_pcm16_async(self, max_samples, out_dev, out_stride, out_len_dev=None):
        _check(load().grail_stream_next_pcm16_async(self.ctx.handle, self.handle, max_samples,
                                                    out_dev, out_stride, out_len_dev))

    def close(self):
        if self.handle:
            load().grail_stream_close(self.ctx.handle, self.handle)
            self.handle = None


class LiveStream(Stream):
    """grail_stream_open_live: n_utt chains whose sources deliver while they run (examples/interactive.rs:31-48)."""

    def __init__(self, ctx, n_utt, voice_ids=None, jitter_seeds=None, ring_segments=0, elems=False):
        self.ctx, self.n_utt, self.elems, self.batch = ctx, n_utt, elems, None
        if voice_ids is not None:
            voice_ids = np.ascontiguousarray(voice_ids, dtype=np.uint32)
            assert len(voice_ids) == n_utt
        if jitter_seeds is not None:
            jitter_seeds = np.ascontiguousarray(jitter_seeds, dtype=np.uint32)
            assert len(jitter_seeds) == n_utt
        h = C.c_void_p()
        _check(load().grail_stream_open_live(ctx.handle, n_utt, _ptr(voice_ids), _ptr(jitter_seeds), ring_segments,
                                             1 if elems else 0, C.byref(h)))
        self.handle = h

    def append(self, segs, seg_offsets):
        """Utterance u receives segs[seg_offsets[u]:seg_offsets[u + 1]] behind what it already has."""
        seg_offsets = np.ascontiguousarray(seg_offsets, dtype=np.uint32)
        assert len(seg_offsets) == self.n_utt + 1
        if self.elems:
            arr = (SequenceElem * max(len(segs), 1))(*segs)
            _check(load().grail_stream_append_elems(self.ctx.handle, self.handle, C.cast(arr, C.c_void_p),
                                                    seg_offsets.ctypes.data))
        else:
            segs = np.ascontiguousarray(segs, dtype=PHONEME_DTYPE)
            _check(load().grail_stream_append(self.ctx.handle, self.handle, segs.ctypes.data, seg_offsets.ctypes.data))

    def finish(self, which=None):
        if which is not None:
            which = np.ascontiguousarray(which, dtype=np.uint8)
            assert len(which) == self.n_utt
        _check(load().grail_stream_finish(self.ctx.handle, self.handle, _ptr(which)))

    def pending(self):
        out = np.zeros(self.n_utt, dtype=np.uint32)
        _check(load().grail_stream_pending(self.ctx.handle, self.handle, out.ctypes.data))
        return out


class Context:
    """grail_ctx: one per (process, GPU)."""

    def __init__(self, device=0):
        h = C.c_void_p()
        _check(load().grail_create(device, C.byref(h)))
        self.handle = h
        self.device = device

    def close(self):
        if self.handle:
            load().grail_destroy(self.handle)
            self.handle = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def set_voices(self, voices):
        arr = (Voice * len(voices))(*[v.copy() for v in voices])
        _check(load().grail_set_voices(self.handle, C.cast(arr, C.c_void_p), len(voices)))

    def get_voices(self):
        n = C.c_uint32()
        _check(load().grail_get_voices(self.handle, None, 0, C.byref(n)))
        arr = (Voice * max(n.value, 1))()
        _check(load().grail_get_voices(self.handle, C.cast(arr, C.c_void_p), n.value, C.byref(n)))
        return [arr[i].copy() for i in range(n.value)]

    def pci_bus_id(self):
        buf = C.create_string_buffer(32)
        _check(load().grail_device_pci_bus_id(self.handle, buf, 32))
        return buf.value.decode()

    def set_option(self, name, value):
        _check(load().grail_set_option(self.handle, name.encode(), value))

    def get_option(self, name):
        v = C.c_int64()
        _check(load().grail_get_option(self.handle, name.encode(), C.byref(v)))
        return v.value

    @staticmethod
    def _prep(segs_dtype, segs, seg_offsets, voice_ids, jitter_seeds):
        seg_offsets = np.ascontiguousarray(seg_offsets, dtype=np.uint32)
        n_utt = len(seg_offsets) - 1
        if voice_ids is not None:
            voice_ids = np.ascontiguousarray(voice_ids, dtype=np.uint32)
            assert len(voice_ids) == n_utt
        if jitter_seeds is not None:
            jitter_seeds = np.ascontiguousarray(jitter_seeds, dtype=np.uint32)
            assert len(jitter_seeds) == n_utt
        return seg_offsets, n_utt, voice_ids, jitter_seeds

    def upload(self, segs, seg_offsets, voice_ids=None, jitter_seeds=None):
        segs = np.ascontiguousarray(segs, dtype=PHONEME_DTYPE)
        seg_offsets, n_utt, voice_ids, jitter_seeds = self._prep(None, segs, seg_offsets,
                                                                 voice_ids, jitter_seeds)
        h = C.c_void_p()
        _check(load().grail_batch_upload(self.handle, segs.ctypes.data, seg_offsets.ctypes.data,
                                         _ptr(voice_ids), _ptr(jitter_seeds), n_utt, C.byref(h)))
        return Batch(self, h, n_utt)

    def upload_elems(self, seq_elems, seg_offsets, voice_ids=None, jitter_seeds=None):
        arr = (SequenceElem * max(len(seq_elems), 1))(*seq_elems)
        seg_offsets, n_utt, voice_ids, jitter_seeds = self._prep(None, None, seg_offsets,
                                                                 voice_ids, jitter_seeds)
        h = C.c_void_p()
        _check(load().grail_batch_upload_elems(self.handle, C.cast(arr, C.c_void_p),
                                               seg_offsets.ctypes.data, _ptr(voice_ids),
                                               _ptr(jitter_seeds), n_utt, C.byref(h)))
        return Batch(self, h, n_utt)

    def synthesize(self, segs, seg_offsets, voice_ids=None, jitter_seeds=None, out_stride=None,
                   allow_truncation=False):
        """One-call form over host buffers.  Returns (out[n_utt, out_stride], out_len)."""
        segs = np.ascontiguousarray(segs, dtype=PHONEME_DTYPE)
        seg_offsets, n_utt, voice_ids, jitter_seeds = self._prep(None, segs, seg_offsets,
                                                                 voice_ids, jitter_seeds)
        if out_stride is None:
            b = self.upload(segs, seg_offsets, voice_ids, jitter_seeds)
            try:
                lens = b.lengths()
            finally:
                b.free()
            out_stride = int((max(int(lens.max()) if n_utt else 0, 1) + 3) // 4 * 4)
        out = np.zeros((max(n_utt, 1), out_stride), dtype=np.float32)
        out_len = np.zeros(max(n_utt, 1), dtype=np.uint32)
        st = load().grail_synthesize_batch(self.handle, segs.ctypes.data, seg_offsets.ctypes.data,
                                           _ptr(voice_ids), _ptr(jitter_seeds), n_utt,
                                           out.ctypes.data, out_stride, out_len.ctypes.data,
                                           OUT_HOST)
        if not (allow_truncation and st == ERR_BUFFER_TOO_SMALL):
            _check(st)
        return out[:n_utt], out_len[:n_utt]

    def synthesize_pcm16(self, segs, seg_offsets, voice_ids=None, jitter_seeds=None, out_stride=4096):
        """One-call form with i16 PCM rows (examples/cli.rs:49 on the device)."""
        segs = np.ascontiguousarray(segs, dtype=PHONEME_DTYPE)
        seg_offsets, n_utt, voice_ids, jitter_seeds = self._prep(None, segs, seg_offsets,
                                                                 voice_ids, jitter_seeds)
        out = np.zeros((max(n_utt, 1), out_stride), dtype=np.int16)
        out_len = np.zeros(max(n_utt, 1), dtype=np.uint32)
        _check(load().grail_synthesize_batch_pcm16(
            self.handle, segs.ctypes.data, seg_offsets.ctypes.data, _ptr(voice_ids),
            _ptr(jitter_seeds), n_utt, out.ctypes.data, out_stride, out_len.ctypes.data, OUT_HOST))
        return out[:n_utt], out_len[:n_utt]

    def synthesize_elems(self, seq_elems, seg_offsets, voice_ids=None, jitter_seeds=None,
                         out_stride=4096):
        arr = (SequenceElem * max(len(seq_elems), 1))(*seq_elems)
        seg_offsets, n_utt, voice_ids, jitter_seeds = self._prep(None, None, seg_offsets,
                                                                 voice_ids, jitter_seeds)
        out = np.zeros((max(n_utt, 1), out_stride), dtype=np.float32)
        out_len = np.zeros(max(n_utt, 1), dtype=np.uint32)
        _check(load().grail_synthesize_batch_elems(
            self.handle, C.cast(arr, C.c_void_p), seg_offsets.ctypes.data, _ptr(voice_ids),
            _ptr(jitter_seeds), n_utt, out.ctypes.data, out_stride, out_len.ctypes.data, OUT_HOST))
        return out[:n_utt], out_len[:n_utt]

    def say(self, texts, voice_ids=None, jitter_seeds=None, out_stride=None, seconds_per_char=1.6):
        """examples/cli.rs:175-184 for a list of texts. Returns (out[n, stride], out_len)."""
        n = len(texts)
        arr = (C.c_char_p * max(n, 1))(*[t.encode("utf-8") for t in texts])
        if voice_ids is not None:
            voice_ids = np.ascontiguousarray(voice_ids, dtype=np.uint32)
        if jitter_seeds is not None:
            jitter_seeds = np.ascontiguousarray(jitter_seeds, dtype=np.uint32)
        if out_stride is None:
            rate = max(v.sample_rate for v in self.get_voices())
            longest = max([len(t) for t in texts] + [1])
            out_stride = (int((longest * seconds_per_char + 1.0) * rate) + 63) // 64 * 64
        out = np.zeros((max(n, 1), out_stride), dtype=np.float32)
        out_len = np.zeros(max(n, 1), dtype=np.uint32)
        _check(load().grail_say_batch(self.handle, arr, n, _ptr(voice_ids), _ptr(jitter_seeds),
                                      out.ctypes.data, out_stride, out_len.ctypes.data, OUT_HOST))
        return out[:n], out_len[:n]

    def pcm16(self, in_dev, in_stride, len_dev, n_utt, max_len, out_dev, out_stride):
        _check(load().grail_pcm16_async(self.handle, in_dev, in_stride, len_dev, n_utt, max_len,
                                        out_dev, out_stride))

    def digest(self, in_dev, in_stride, len_dev, n_utt):
        """(bit-pattern sums mod 2^64, max |x|, non-finite counts) per row, computed on the device."""
        sums = np.zeros(max(n_utt, 1), dtype=np.uint64)
        maxabs = np.zeros(max(n_utt, 1), dtype=np.float32)
        bad = np.zeros(max(n_utt, 1), dtype=np.uint32)
        _check(load().grail_batch_digest(self.handle, in_dev, in_stride, len_dev, n_utt,
                                         sums.ctypes.data, maxabs.ctypes.data, bad.ctypes.data))
        return sums[:n_utt], maxabs[:n_utt], bad[:n_utt]

    def compare(self, a_dev, b_dev, stride, len_a_dev, len_b_dev, n_utt):
        """(max |a-b|, sum (a-b)^2, structural mismatches) per row of two device renderings."""
        md = np.zeros(max(n_utt, 1), dtype=np.float32)
        sq = np.zeros(max(n_utt, 1), dtype=np.float64)
        bad = np.zeros(max(n_utt, 1), dtype=np.uint32)
        _check(load().grail_batch_compare(self.handle, a_dev, b_dev, stride, len_a_dev, len_b_dev, n_utt,
                                          md.ctypes.data, sq.ctypes.data, bad.ctypes.data))
        return md[:n_utt], sq[:n_utt], bad[:n_utt]

    def h2d(self, dst_dev, src, nbytes):
        _check(load().grail_memcpy_h2d(self.handle, dst_dev, src.ctypes.data, nbytes))

    def sync(self):
        _check(load().grail_sync(self.handle))

    def last_kernel_ms(self):
        ms = C.c_float()
        _check(load().grail_last_kernel_ms(self.handle, C.byref(ms)))
        return ms.value

    def last_kernel_name(self):
        return load().grail_last_kernel_name(self.handle).decode()

    def device_alloc(self, nbytes):
        p = C.c_void_p()
        _check(load().grail_device_alloc(self.handle, nbytes, C.byref(p)))
        return p

    def device_free(self, p):
        _check(load().grail_device_free(self.handle, p))

    def host_alloc(self, shape, dtype):
        """A numpy array in pinned host memory (grail_host_alloc); free it with host_free(array)."""
        dtype = np.dtype(dtype)
        n = int(np.prod(shape)) * dtype.itemsize
        p = C.c_void_p()
        _check(load().grail_host_alloc(self.handle, n, C.byref(p)))
        buf = (C.c_char * max(n, 1)).from_address(p.value)
        arr = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)
        self._pinned = getattr(self, "_pinned", {})
        self._pinned[arr.ctypes.data] = p
        return arr

    def host_free(self, arr):
        p = self._pinned.pop(arr.ctypes.data)
        _check(load().grail_host_free(self.handle, p))

    def synthesize_into(self, out, out_len, segs, seg_offsets, voice_ids=None, jitter_seeds=None):
        """One-call form into a caller-owned host array out[n_utt, out_stride] (f32 or i16)."""
        segs = np.ascontiguousarray(segs, dtype=PHONEME_DTYPE)
        seg_offsets, n_utt, voice_ids, jitter_seeds = self._prep(None, segs, seg_offsets, voice_ids,
                                                                 jitter_seeds)
        assert out.shape[0] >= n_utt and out.flags.c_contiguous
        fn = load().grail_synthesize_batch if out.dtype == np.float32 else load().grail_synthesize_batch_pcm16
        _check(fn(self.handle, segs.ctypes.data, seg_offsets.ctypes.data, _ptr(voice_ids), _ptr(jitter_seeds),
                  n_utt, out.ctypes.data, out.shape[1], out_len.ctypes.data, OUT_HOST))

    def d2h(self, dst, src_dev, nbytes, offset=0):
        src = C.c_void_p(src_dev.value + offset)
        _check(load().grail_memcpy_d2h(self.handle, dst.ctypes.data, src, nbytes))

    def memset(self, dst_dev, value, nbytes):
        _check(load().grail_memset_d(self.handle, dst_dev, value, nbytes))

    # RCCL
    @staticmethod
    def comm_unique_id():
        buf = (C.c_uint8 * UNIQUE_ID_BYTES)()
        _check(load().grail_comm_unique_id(buf))
        return bytes(buf)

    def comm_init(self, unique_id, rank, world):
        buf = (C.c_uint8 * UNIQUE_ID_BYTES).from_buffer_copy(unique_id)
        _check(load().grail_comm_init(self.handle, buf, rank, world))

    def broadcast_voices(self, n_voices, root=0):
        _check(load().grail_broadcast_voices(self.handle, n_voices, root))

    def comm_info(self):
        """(ncclCommCount, ncclCommUserRank) of this context's communicator; (0, 0) without one."""
        n, r = C.c_uint32(), C.c_uint32()
        _check(load().grail_comm_info(self.handle, C.byref(n), C.byref(r)))
        return n.value, r.value


class _BorrowedContext(Context):
    """A node's context (grail_node_context): owned by the node, never destroyed from here."""

    def __init__(self, handle, device):       # noqa: super().__init__ would create a context
        self.handle, self.device = handle, device

    def close(self):
        self.handle = None


class Node:
    """grail_node: one process, several GPUs — a context and a host thread per device; one call renders a batch over all
    of them (contiguous shards, no data-path collective; the voice table travels by one ncclBroadcast)."""

    def __init__(self, devices, voices_without_rccl=False):
        self.devices = [int(d) for d in devices]
        arr = (C.c_int * len(self.devices))(*self.devices)
        h = C.c_void_p()
        _check(load().grail_node_create(arr, len(self.devices), C.byref(h)))
        self.handle = h
        if voices_without_rccl:
            self.set_option("node_voices_without_rccl", 1)

    def close(self):
        if self.handle:
            load().grail_node_destroy(self.handle)
            self.handle = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def size(self):
        return int(load().grail_node_size(self.handle))

    def context(self, index):
        h = C.c_void_p()
        _check(load().grail_node_context(self.handle, index, C.byref(h)))
        return _BorrowedContext(h, self.devices[index])

    def set_voices(self, voices):
        arr = (Voice * len(voices))(*[v.copy() for v in voices])
        _check(load().grail_node_set_voices(self.handle, C.cast(arr, C.c_void_p), len(voices)))

    def set_option(self, name, value):
        _check(load().grail_node_set_option(self.handle, name.encode(), value))

    def get_option(self, name):
        v = C.c_int64()
        _check(load().grail_node_get_option(self.handle, name.encode(), C.byref(v)))
        return v.value

    def lengths(self, segs, seg_offsets, voice_ids=None, max_len=0xFFFFFFFF):
        segs = np.ascontiguousarray(segs, dtype=PHONEME_DTYPE)
        seg_offsets, n_utt, voice_ids, _ = Context._prep(None, segs, seg_offsets, voice_ids, None)
        out = np.zeros(max(n_utt, 1), dtype=np.uint32)
        _check(load().grail_node_lengths(self.handle, segs.ctypes.data, seg_offsets.ctypes.data, _ptr(voice_ids), n_utt,
                                         max_len, out.ctypes.data))
        return out[:n_utt]

    def synthesize(self, segs, seg_offsets, voice_ids=None, jitter_seeds=None, out_stride=None, out=None,
                   allow_truncation=False, pcm16=False):
        """grail_node_synthesize_batch(_pcm16) over host buffers.  Returns (out[n_utt, out_stride], out_len)."""
        segs = np.ascontiguousarray(segs, dtype=PHONEME_DTYPE)
        seg_offsets, n_utt, voice_ids, jitter_seeds = Context._prep(None, segs, seg_offsets, voice_ids, jitter_seeds)
        if out is not None:
            out_stride = out.shape[1]
            assert out.shape[0] >= n_utt and out.flags.c_contiguous
            pcm16 = out.dtype == np.int16
        elif out_stride is None:
            lens = self.lengths(segs, seg_offsets, voice_ids)
            out_stride = int((max(int(lens.max()) if n_utt else 0, 1) + 63) // 64 * 64)
        if out is None:
            out = np.zeros((max(n_utt, 1), out_stride), dtype=np.int16 if pcm16 else np.float32)
        out_len = np.zeros(max(n_utt, 1), dtype=np.uint32)
        fn = load().grail_node_synthesize_batch_pcm16 if pcm16 else load().grail_node_synthesize_batch
        st = fn(self.handle, segs.ctypes.data, seg_offsets.ctypes.data, _ptr(voice_ids), _ptr(jitter_seeds), n_utt,
                out.ctypes.data, out_stride, out_len.ctypes.data, OUT_HOST)
        if not (allow_truncation and st == ERR_BUFFER_TOO_SMALL):
            _check(st)
        return out[:n_utt], out_len[:n_utt]

    def synthesize_device(self, segs, seg_offsets, voice_ids, jitter_seeds, out_dev, out_stride):
        """grail_node_synthesize_batch_device: slot i's shard into out_dev[i] (device memory of its GPU).  Returns out_len."""
        segs = np.ascontiguousarray(segs, dtype=PHONEME_DTYPE)
        seg_offsets, n_utt, voice_ids, jitter_seeds = Context._prep(None, segs, seg_offsets, voice_ids, jitter_seeds)
        ptrs = (C.c_void_p * self.size())(*[p if p is None or isinstance(p, C.c_void_p) else C.c_void_p(p) for p in out_dev])
        out_len = np.zeros(max(n_utt, 1), dtype=np.uint32)
        _check(load().grail_node_synthesize_batch_device(self.handle, segs.ctypes.data, seg_offsets.ctypes.data, _ptr(voice_ids),
                                                         _ptr(jitter_seeds), n_utt, ptrs, out_stride, out_len.ctypes.data))
        return out_len[:n_utt]

    def synthesize_elems(self, seq_elems, seg_offsets, voice_ids=None, jitter_seeds=None, out_stride=4096):
        arr = (SequenceElem * max(len(seq_elems), 1))(*seq_elems)
        seg_offsets, n_utt, voice_ids, jitter_seeds = Context._prep(None, None, seg_offsets, voice_ids, jitter_seeds)
        out = np.zeros((max(n_utt, 1), out_stride), dtype=np.float32)
        out_len = np.zeros(max(n_utt, 1), dtype=np.uint32)
        _check(load().grail_node_synthesize_batch_elems(
            self.handle, C.cast(arr, C.c_void_p), seg_offsets.ctypes.data, _ptr(voice_ids), _ptr(jitter_seeds), n_utt,
            out.ctypes.data, out_stride, out_len.ctypes.data, OUT_HOST))
        return out[:n_utt], out_len[:n_utt]

    def say(self, texts, voice_ids=None, jitter_seeds=None, out_stride=4096):
        n = len(texts)
        arr = (C.c_char_p * max(n, 1))(*[t.encode("utf-8") for t in texts])
        if voice_ids is not None:
            voice_ids = np.ascontiguousarray(voice_ids, dtype=np.uint32)
        if jitter_seeds is not None:
            jitter_seeds = np.ascontiguousarray(jitter_seeds, dtype=np.uint32)
        out = np.zeros((max(n, 1), out_stride), dtype=np.float32)
        out_len = np.zeros(max(n, 1), dtype=np.uint32)
        _check(load().grail_node_say_batch(self.handle, arr, n, _ptr(voice_ids), _ptr(jitter_seeds), out.ctypes.data,
                                           out_stride, out_len.ctypes.data, OUT_HOST))
        return out[:n], out_len[:n]

    def last_shard_ms(self):
        ms = (C.c_float * self.size())()
        _check(load().grail_node_last_shard_ms(self.handle, ms, self.size()))
        return [float(x) for x in ms]

    def host_alloc(self, shape, dtype):
        """A numpy array in pinned host memory every device of the node can copy into; free it with host_free(array)."""
        dtype = np.dtype(dtype)
        n = int(np.prod(shape)) * dtype.itemsize
        p = C.c_void_p()
        _check(load().grail_node_host_alloc(self.handle, n, C.byref(p)))
        buf = (C.c_char * max(n, 1)).from_address(p.value)
        arr = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)
        self._pinned = getattr(self, "_pinned", {})
        self._pinned[arr.ctypes.data] = p
        return arr

    def host_free(self, arr):
        p = self._pinned.pop(arr.ctypes.data)
        _check(load().grail_node_host_free(self.handle, p))
