//! Raw mirror of `include/grail_hip.h` (ABI version 4: `GRAIL_ABI_VERSION`; compare it with
//! `grail_abi_version()` before the first call, as `grail_hip::Context::new` does).  Field orders follow grail-rs:
//! `SynthesisElem` src/lib.rs:316-337, `Voice` :696-717, `PhonemeElem` :961-973,
//! `SequenceElem` :814-824, `Phoneme` :632-649.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_int};

pub const GRAIL_NUM_FORMANTS: usize = 8;
pub const GRAIL_NUM_VOICED: usize = 2;
pub const GRAIL_UNIQUE_ID_BYTES: usize = 128;

pub const GRAIL_ABI_VERSION: c_int = 4;

pub const GRAIL_OK: c_int = 0;
pub const GRAIL_ERR_INVALID_ARG: c_int = -1;
pub const GRAIL_ERR_NO_DEVICE: c_int = -2;
pub const GRAIL_ERR_HIP: c_int = -3;
pub const GRAIL_ERR_BUFFER_TOO_SMALL: c_int = -4;
pub const GRAIL_ERR_OUT_OF_MEMORY: c_int = -5;
pub const GRAIL_ERR_RCCL: c_int = -6;
pub const GRAIL_ERR_NO_VOICES: c_int = -7;
pub const GRAIL_OUT_HOST: u32 = 0;
pub const GRAIL_OUT_DEVICE: u32 = 1;

#[repr(C)]
#[derive(Copy, Clone, Debug, PartialEq)]
pub struct grail_synthesis_elem {
    pub frequency: f32,
    pub formant_freq: [f32; GRAIL_NUM_FORMANTS],
    pub formant_bw: [f32; GRAIL_NUM_FORMANTS],
    pub formant_smooth: [f32; GRAIL_NUM_FORMANTS],
    pub formant_breath: [f32; GRAIL_NUM_FORMANTS],
    pub formant_turb: [f32; GRAIL_NUM_FORMANTS],
    pub formant_amp: [f32; GRAIL_NUM_FORMANTS],
}

#[repr(C)]
#[derive(Copy, Clone, Debug, PartialEq)]
pub struct grail_voice {
    pub sample_rate: f32,
    pub phonemes: [grail_synthesis_elem; GRAIL_NUM_VOICED],
    pub center_frequency: f32,
    pub jitter_frequency: f32,
    pub jitter_delta_frequency: f32,
    pub jitter_delta_formant_frequency: f32,
    pub jitter_delta_amplitude: f32,
}

#[repr(C)]
#[derive(Copy, Clone, Debug, PartialEq)]
pub struct grail_phoneme_elem {
    pub phoneme: i32,
    pub length: f32,
    pub blend_length: f32,
    pub frequency: f32,
}

#[repr(C)]
#[derive(Copy, Clone, Debug, PartialEq)]
pub struct grail_sequence_elem {
    pub has_elem: i32,
    pub elem: grail_synthesis_elem,
    pub length: f32,
    pub blend_length: f32,
}

#[repr(C)]
pub struct grail_rule {
    pub string: *const u32,
    pub string_len: u32,
    pub phonemes: *const i32,
    pub n_phonemes: u32,
}

/// One kernel launch of a batch's plan (`grail_plan_blocks`).
#[repr(C)]
#[derive(Copy, Clone, Debug, Default, PartialEq)]
pub struct grail_plan_block {
    pub rows: u32,
    pub lanes_per_utterance: u32,
    pub pipelined: u32,
    pub chunks: u32,
    pub scan: u32,
    pub fast: u32,
    pub formants: u32,
    pub model_ms: f32,
}

/// The rows and segments one device of a node renders (`grail_node_shard_of`).
#[repr(C)]
#[derive(Copy, Clone, Debug, Default, PartialEq)]
pub struct grail_node_shard {
    pub first_row: u64,
    pub rows: u64,
    pub first_seg: u32,
    pub n_segs: u32,
}

// The layouts this crate mirrors, checked at compile time against the C header's (tests/test_abi.py compares these
// literals with the sizes the C compiler gives the header's structs).
const _: () = assert!(std::mem::size_of::<grail_synthesis_elem>() == 196);
const _: () = assert!(std::mem::size_of::<grail_voice>() == 416);
const _: () = assert!(std::mem::size_of::<grail_phoneme_elem>() == 16);
const _: () = assert!(std::mem::size_of::<grail_sequence_elem>() == 208);
const _: () = assert!(std::mem::size_of::<grail_plan_block>() == 32);
const _: () = assert!(std::mem::size_of::<grail_node_shard>() == 24);

#[repr(C)]
pub struct grail_ctx {
    _private: [u8; 0],
}
#[repr(C)]
pub struct grail_batch {
    _private: [u8; 0],
}
#[repr(C)]
pub struct grail_stream {
    _private: [u8; 0],
}
#[repr(C)]
pub struct grail_node {
    _private: [u8; 0],
}

extern "C" {
    pub fn grail_abi_version() -> c_int;
    pub fn grail_status_string(status: c_int) -> *const c_char;
    pub fn grail_last_error() -> *const c_char;

    pub fn grail_elem_silent(out: *mut grail_synthesis_elem);
    pub fn grail_elem_new_phoneme(out: *mut grail_synthesis_elem, freq: *const f32, bw: *const f32,
        smooth: *const f32, turb: *const f32, breath: *const f32, amp: *const f32);
    pub fn grail_elem_new(out: *mut grail_synthesis_elem, sample_rate: f32, frequency: f32,
        freq: *const f32, smooth: *const f32, bw: *const f32, breath: *const f32, turb: *const f32,
        amp: *const f32);
    pub fn grail_elem_resample(elem: *mut grail_synthesis_elem, old_rate: f32, new_rate: f32);
    pub fn grail_elem_blend(out: *mut grail_synthesis_elem, a: *const grail_synthesis_elem,
        b: *const grail_synthesis_elem, alpha: f32);
    pub fn grail_voice_generic(out: *mut grail_voice);
    pub fn grail_voice_generic_at(out: *mut grail_voice, sample_rate: f32);
    pub fn grail_voice_get(voice: *const grail_voice, phoneme: i32, out: *mut grail_synthesis_elem) -> c_int;

    pub fn grail_create(device: c_int, out: *mut *mut grail_ctx) -> c_int;
    pub fn grail_destroy(ctx: *mut grail_ctx) -> c_int;
    pub fn grail_device_count(count: *mut c_int) -> c_int;
    pub fn grail_device_pci_bus_id(ctx: *mut grail_ctx, out: *mut c_char, cap: usize) -> c_int;
    pub fn grail_time_split_warmup(voice: *const grail_voice) -> u32;
    pub fn grail_length_bound(segment_lengths: *const f32, n_segments: u32, sample_rate: f32) -> u64;
    pub fn grail_fast_sharpness(voice: *const grail_voice) -> f32;
    pub fn grail_time_split_grid(span_samples: u32, warmup: u32, chunks: u32, ff_cost_permille: u32,
                                 bounds: *mut u32) -> c_int;
    pub fn grail_plan_blocks(compute_units: u32, arithmetic: c_int, live_formants: c_int, warmup: u32, rows: u32,
                             span_samples: u32, blocks: *mut grail_plan_block, cap: u32, n_blocks: *mut u32) -> c_int;
    pub fn grail_plan_ragged_blocks(compute_units: u32, arithmetic: c_int, live_formants: c_int, warmup: u32, rows: u32,
                                    row_samples: *const u32, row_segments: *const u32, row_kinks: *const u32,
                                    blocks: *mut grail_plan_block, cap: u32, n_blocks: *mut u32) -> c_int;
    pub fn grail_dispatch_model(compute_units: u32, waves_per_workgroup: u32, workgroup_ms: *const f64, order: *const u32,
                                n: u32, makespan_ms: *mut f64) -> c_int;
    pub fn grail_packed_launch_order(compute_units: u32, waves_per_workgroup: u32, workgroup_ms: *const f64, n: u32,
                                     order: *mut u32) -> c_int;
    pub fn grail_set_voices(ctx: *mut grail_ctx, voices: *const grail_voice, n: u32) -> c_int;
    pub fn grail_get_voices(ctx: *mut grail_ctx, voices: *mut grail_voice, cap: u32, n: *mut u32) -> c_int;
    pub fn grail_set_option(ctx: *mut grail_ctx, name: *const c_char, value: i64) -> c_int;
    pub fn grail_get_option(ctx: *mut grail_ctx, name: *const c_char, value: *mut i64) -> c_int;

    pub fn grail_batch_upload(ctx: *mut grail_ctx, segs: *const grail_phoneme_elem,
        seg_offsets: *const u32, voice_ids: *const u32, jitter_seeds: *const u32, n_utt: u32,
        out: *mut *mut grail_batch) -> c_int;
    pub fn grail_batch_upload_elems(ctx: *mut grail_ctx, segs: *const grail_sequence_elem,
        seg_offsets: *const u32, voice_ids: *const u32, jitter_seeds: *const u32, n_utt: u32,
        out: *mut *mut grail_batch) -> c_int;
    pub fn grail_batch_free(ctx: *mut grail_ctx, batch: *mut grail_batch) -> c_int;
    pub fn grail_batch_size(batch: *const grail_batch) -> u32;
    pub fn grail_batch_lengths(ctx: *mut grail_ctx, batch: *const grail_batch, max_len: u32,
        out_len: *mut u32) -> c_int;
    pub fn grail_batch_synthesize_async(ctx: *mut grail_ctx, batch: *const grail_batch,
        out_dev: *mut f32, out_stride: u64, out_len_dev: *mut u32) -> c_int;
    pub fn grail_stream_open(ctx: *mut grail_ctx, batch: *const grail_batch,
        out: *mut *mut grail_stream) -> c_int;
    pub fn grail_stream_next_async(ctx: *mut grail_ctx, stream: *mut grail_stream, max_samples: u32,
        out_dev: *mut f32, out_stride: u64, out_len_dev: *mut u32) -> c_int;
    pub fn grail_stream_next_pcm16_async(ctx: *mut grail_ctx, stream: *mut grail_stream,
        max_samples: u32, out_dev: *mut i16, out_stride: u64, out_len_dev: *mut u32) -> c_int;
    pub fn grail_stream_close(ctx: *mut grail_ctx, stream: *mut grail_stream) -> c_int;
    pub fn grail_stream_open_live(ctx: *mut grail_ctx, n_utt: u32, voice_ids: *const u32, jitter_seeds: *const u32,
        ring_segments: u32, caller_built_elems: c_int, out: *mut *mut grail_stream) -> c_int;
    pub fn grail_stream_append(ctx: *mut grail_ctx, stream: *mut grail_stream, segs: *const grail_phoneme_elem,
        seg_offsets: *const u32) -> c_int;
    pub fn grail_stream_append_elems(ctx: *mut grail_ctx, stream: *mut grail_stream, segs: *const grail_sequence_elem,
        seg_offsets: *const u32) -> c_int;
    pub fn grail_stream_finish(ctx: *mut grail_ctx, stream: *mut grail_stream, which: *const u8) -> c_int;
    pub fn grail_stream_pending(ctx: *mut grail_ctx, stream: *mut grail_stream, pending: *mut u32) -> c_int;
    pub fn grail_sync(ctx: *mut grail_ctx) -> c_int;
    pub fn grail_last_kernel_ms(ctx: *mut grail_ctx, ms: *mut f32) -> c_int;
    pub fn grail_last_kernel_name(ctx: *mut grail_ctx) -> *const c_char;
    pub fn grail_synthesize_batch(ctx: *mut grail_ctx, segs: *const grail_phoneme_elem,
        seg_offsets: *const u32, voice_ids: *const u32, jitter_seeds: *const u32, n_utt: u32,
        out: *mut f32, out_stride: u64, out_len: *mut u32, flags: u32) -> c_int;
    pub fn grail_synthesize_batch_elems(ctx: *mut grail_ctx, segs: *const grail_sequence_elem,
        seg_offsets: *const u32, voice_ids: *const u32, jitter_seeds: *const u32, n_utt: u32,
        out: *mut f32, out_stride: u64, out_len: *mut u32, flags: u32) -> c_int;

    pub fn grail_language_generic(rules: *mut *const grail_rule, case_sensitive: *mut c_int) -> u32;
    pub fn grail_transcribe(text: *const u32, text_len: u32, rules: *const grail_rule, n_rules: u32,
        case_sensitive: c_int, leading_silence: c_int, out: *mut i32, cap: u32, n_out: *mut u32) -> c_int;
    pub fn grail_intonate(voice: *const grail_voice, phonemes: *const i32, n: u32,
        out: *mut grail_phoneme_elem) -> c_int;
    pub fn grail_text_to_phoneme_elems(voice: *const grail_voice, text_utf8: *const c_char,
        out: *mut grail_phoneme_elem, cap: u32, n_out: *mut u32) -> c_int;
    pub fn grail_say_batch(ctx: *mut grail_ctx, texts: *const *const c_char, n_texts: u32,
        voice_ids: *const u32, jitter_seeds: *const u32, out: *mut f32, out_stride: u64,
        out_len: *mut u32, flags: u32) -> c_int;
    pub fn grail_pcm16_async(ctx: *mut grail_ctx, in_dev: *const f32, in_stride: u64,
        len_dev: *const u32, n_utt: u32, max_len: u32, out_dev: *mut i16, out_stride: u64) -> c_int;
    pub fn grail_batch_synthesize_pcm16_async(ctx: *mut grail_ctx, b: *const grail_batch,
        out_dev: *mut i16, out_stride: u64, out_len_dev: *mut u32) -> c_int;
    pub fn grail_synthesize_batch_pcm16(ctx: *mut grail_ctx, segs: *const grail_phoneme_elem,
        seg_offsets: *const u32, voice_ids: *const u32, jitter_seeds: *const u32, n_utt: u32,
        out: *mut i16, out_stride: u64, out_len: *mut u32, flags: u32) -> c_int;
    pub fn grail_batch_digest(ctx: *mut grail_ctx, in_dev: *const f32, in_stride: u64,
        len_dev: *const u32, n_utt: u32, sums: *mut u64, maxabs: *mut f32, nonfinite: *mut u32) -> c_int;
    pub fn grail_batch_compare(ctx: *mut grail_ctx, a_dev: *const f32, b_dev: *const f32, stride: u64,
                               len_a_dev: *const u32, len_b_dev: *const u32, n_utt: u32,
                               maxdiff: *mut f32, sumsq: *mut f64, mismatches: *mut u32) -> c_int;
    pub fn grail_wav_write_i16(path: *const c_char, pcm: *const i16, n: u32, sample_rate: u32) -> c_int;

    pub fn grail_device_alloc(ctx: *mut grail_ctx, bytes: usize, out: *mut *mut std::ffi::c_void) -> c_int;
    pub fn grail_device_free(ctx: *mut grail_ctx, ptr: *mut std::ffi::c_void) -> c_int;
    pub fn grail_host_alloc(ctx: *mut grail_ctx, bytes: usize, out: *mut *mut std::ffi::c_void) -> c_int;
    pub fn grail_host_free(ctx: *mut grail_ctx, ptr: *mut std::ffi::c_void) -> c_int;
    pub fn grail_memcpy_d2h(ctx: *mut grail_ctx, dst: *mut std::ffi::c_void, src: *const std::ffi::c_void, bytes: usize) -> c_int;
    pub fn grail_memcpy_h2d(ctx: *mut grail_ctx, dst: *mut std::ffi::c_void, src: *const std::ffi::c_void, bytes: usize) -> c_int;
    pub fn grail_memset_d(ctx: *mut grail_ctx, dst: *mut std::ffi::c_void, value: c_int, bytes: usize) -> c_int;

    pub fn grail_shard_range(n_utt: u64, rank: u32, world: u32, begin: *mut u64, end: *mut u64);
    pub fn grail_comm_unique_id(id: *mut u8) -> c_int;
    pub fn grail_comm_init(ctx: *mut grail_ctx, id: *const u8, rank: u32, world: u32) -> c_int;
    pub fn grail_broadcast_voices(ctx: *mut grail_ctx, n_voices: u32, root: u32) -> c_int;
    pub fn grail_comm_info(ctx: *mut grail_ctx, ranks: *mut u32, rank: *mut u32) -> c_int;
    pub fn grail_comm_destroy(ctx: *mut grail_ctx) -> c_int;

    pub fn grail_node_create(devices: *const c_int, n_devices: u32, out: *mut *mut grail_node) -> c_int;
    pub fn grail_node_destroy(node: *mut grail_node) -> c_int;
    pub fn grail_node_size(node: *const grail_node) -> u32;
    pub fn grail_node_context(node: *mut grail_node, index: u32, ctx: *mut *mut grail_ctx) -> c_int;
    pub fn grail_node_set_voices(node: *mut grail_node, voices: *const grail_voice, n_voices: u32) -> c_int;
    pub fn grail_node_set_option(node: *mut grail_node, name: *const c_char, value: i64) -> c_int;
    pub fn grail_node_get_option(node: *mut grail_node, name: *const c_char, value: *mut i64) -> c_int;
    pub fn grail_node_shard_of(seg_offsets: *const u32, n_utt: u64, index: u32, n_devices: u32,
        shard: *mut grail_node_shard, rebased_offsets: *mut u32, cap: u64) -> c_int;
    pub fn grail_node_synthesize_batch(node: *mut grail_node, segs: *const grail_phoneme_elem,
        seg_offsets: *const u32, voice_ids: *const u32, jitter_seeds: *const u32, n_utt: u32,
        out: *mut f32, out_stride: u64, out_len: *mut u32, flags: u32) -> c_int;
    pub fn grail_node_synthesize_batch_elems(node: *mut grail_node, segs: *const grail_sequence_elem,
        seg_offsets: *const u32, voice_ids: *const u32, jitter_seeds: *const u32, n_utt: u32,
        out: *mut f32, out_stride: u64, out_len: *mut u32, flags: u32) -> c_int;
    pub fn grail_node_synthesize_batch_pcm16(node: *mut grail_node, segs: *const grail_phoneme_elem,
        seg_offsets: *const u32, voice_ids: *const u32, jitter_seeds: *const u32, n_utt: u32,
        out: *mut i16, out_stride: u64, out_len: *mut u32, flags: u32) -> c_int;
    pub fn grail_node_synthesize_batch_device(node: *mut grail_node, segs: *const grail_phoneme_elem,
        seg_offsets: *const u32, voice_ids: *const u32, jitter_seeds: *const u32, n_utt: u32,
        out_dev: *const *mut f32, out_stride: u64, out_len: *mut u32) -> c_int;
    pub fn grail_node_say_batch(node: *mut grail_node, texts: *const *const c_char, n_texts: u32,
        voice_ids: *const u32, jitter_seeds: *const u32, out: *mut f32, out_stride: u64,
        out_len: *mut u32, flags: u32) -> c_int;
    pub fn grail_node_lengths(node: *mut grail_node, segs: *const grail_phoneme_elem, seg_offsets: *const u32,
        voice_ids: *const u32, n_utt: u32, max_len: u32, out_len: *mut u32) -> c_int;
    pub fn grail_node_last_shard_ms(node: *mut grail_node, ms: *mut f32, cap: u32) -> c_int;
    pub fn grail_node_host_alloc(node: *mut grail_node, bytes: usize, out: *mut *mut std::ffi::c_void) -> c_int;
    pub fn grail_node_host_free(node: *mut grail_node, ptr: *mut std::ffi::c_void) -> c_int;
}
