//! `utterances.synthesize_batch(&gpu)` / `utterances.synthesize_batch(&node)` — the batched counterpart of grail-rs's
//! `.select(v).sequence(v).jitter(seed, v).synthesize()` (reference src/lib.rs:1013, 941, 786,
//! 587; trait pattern of `IntoSynthesize`, src/lib.rs:582-600).  Per-utterance results are
//! bit-identical to the CPU iterator chain.  SOURCE ONLY: not compiled in the build image.
use grail_hip_sys as sys;
use grail_rs::{PhonemeElem, Voice};
use std::ffi::CStr;

#[derive(Debug)]
pub struct Error {
    pub status: i32,
    pub message: String,
}

fn check(status: i32) -> Result<(), Error> {
    if status == sys::GRAIL_OK {
        return Ok(());
    }
    let message = unsafe { CStr::from_ptr(sys::grail_last_error()) }.to_string_lossy().into_owned();
    Err(Error { status, message })
}

/// One GPU + a voice table (grail_ctx).
pub struct Gpu {
    ctx: *mut sys::grail_ctx,
}

// grail-rs keeps its struct layouts private to Rust; convert field by field.  `Array` needs a
// `pub fn to_array(self) -> [f32; NUM_FORMANTS]` accessor in grail-rs (one line).
fn elem_to_c(e: &grail_rs::SynthesisElem) -> sys::grail_synthesis_elem {
    sys::grail_synthesis_elem {
        frequency: e.frequency,
        formant_freq: e.formant_freq.to_array(),
        formant_bw: e.formant_bw.to_array(),
        formant_smooth: e.formant_smooth.to_array(),
        formant_breath: e.formant_breath.to_array(),
        formant_turb: e.formant_turb.to_array(),
        formant_amp: e.formant_amp.to_array(),
    }
}

fn voice_to_c(v: &Voice) -> sys::grail_voice {
    sys::grail_voice {
        sample_rate: v.sample_rate,
        phonemes: [elem_to_c(&v.phonemes.a), elem_to_c(&v.phonemes.e)],
        center_frequency: v.center_frequency,
        jitter_frequency: v.jitter_frequency,
        jitter_delta_frequency: v.jitter_delta_frequency,
        jitter_delta_formant_frequency: v.jitter_delta_formant_frequency,
        jitter_delta_amplitude: v.jitter_delta_amplitude,
    }
}

impl Gpu {
    pub fn new(device: i32, voices: &[Voice]) -> Result<Self, Error> {
        // the library on the loader's path may be older or newer than the header this crate mirrors
        let have = unsafe { sys::grail_abi_version() };
        if have != sys::GRAIL_ABI_VERSION {
            return Err(Error { status: sys::GRAIL_ERR_INVALID_ARG,
                               message: format!("libgrail_hip.so has ABI version {have}, grail-hip-sys mirrors {}", sys::GRAIL_ABI_VERSION) });
        }
        let mut ctx = std::ptr::null_mut();
        check(unsafe { sys::grail_create(device, &mut ctx) })?;
        let gpu = Gpu { ctx };
        let table: Vec<_> = voices.iter().map(voice_to_c).collect();
        check(unsafe { sys::grail_set_voices(gpu.ctx, table.as_ptr(), table.len() as u32) })?;
        Ok(gpu)
    }
}

impl Drop for Gpu {
    fn drop(&mut self) {
        unsafe { sys::grail_destroy(self.ctx) };
    }
}

pub struct Utterance {
    pub phonemes: Vec<PhonemeElem>,
    pub voice: u32,
    pub jitter_seed: u32,
}

/// A batch in the flat form the C ABI takes (`grail_synthesize_batch`): utterance u is
/// `segs[offs[u]..offs[u + 1]]` with voice `vids[u]` and jitter seed `seeds[u]`.
pub struct FlatBatch {
    segs: Vec<sys::grail_phoneme_elem>,
    offs: Vec<u32>,
    vids: Vec<u32>,
    seeds: Vec<u32>,
}

impl FlatBatch {
    fn new<I: IntoIterator<Item = Utterance>>(utterances: I) -> Self {
        let mut b = FlatBatch { segs: vec![], offs: vec![0u32], vids: vec![], seeds: vec![] };
        for u in utterances {
            b.segs.extend(u.phonemes.iter().map(|p| sys::grail_phoneme_elem {
                phoneme: p.phoneme as i32, // Silence=0 Stop=1 Glide=2 A=3 E=4 (src/lib.rs:632-649)
                length: p.length,
                blend_length: p.blend_length,
                frequency: p.frequency,
            }));
            b.offs.push(b.segs.len() as u32);
            b.vids.push(u.voice);
            b.seeds.push(u.jitter_seed);
        }
        b
    }

    pub fn len(&self) -> u32 {
        self.vids.len() as u32
    }
}

/// What a batch can be rendered on: one GPU ([`Gpu`]) or every GPU of the node ([`Node`]).
pub trait SynthesisTarget {
    fn render(&self, batch: &FlatBatch) -> Result<Vec<Vec<f32>>, Error>;
}

/// `utterances.synthesize_batch(&gpu)` / `utterances.synthesize_batch(&node)`: the batched counterpart of
/// `IntoSynthesize` (src/lib.rs:582-600).
pub trait IntoSynthesizeBatch {
    fn synthesize_batch<T: SynthesisTarget>(self, target: &T) -> Result<Vec<Vec<f32>>, Error>;
}

impl<I: IntoIterator<Item = Utterance>> IntoSynthesizeBatch for I {
    fn synthesize_batch<T: SynthesisTarget>(self, target: &T) -> Result<Vec<Vec<f32>>, Error> {
        target.render(&FlatBatch::new(self))
    }
}

fn rows_of(out: &[f32], stride: usize, lens: &[u32]) -> Vec<Vec<f32>> {
    lens.iter().enumerate().map(|(u, &l)| out[u * stride..][..l as usize].to_vec()).collect()
}

impl SynthesisTarget for Gpu {
    fn render(&self, b: &FlatBatch) -> Result<Vec<Vec<f32>>, Error> {
        let gpu = self;
        let n = b.len();
        let mut lens = vec![0u32; n as usize];
        unsafe {
            let mut h = std::ptr::null_mut();
            check(sys::grail_batch_upload(gpu.ctx, b.segs.as_ptr(), b.offs.as_ptr(), b.vids.as_ptr(),
                                          b.seeds.as_ptr(), n, &mut h))?;
            let r = check(sys::grail_batch_lengths(gpu.ctx, h, u32::MAX, lens.as_mut_ptr()));
            sys::grail_batch_free(gpu.ctx, h);
            r?;
        }
        let stride = (*lens.iter().max().unwrap_or(&0) as u64 + 63) / 64 * 64;
        // Results of up to PINNED_LIMIT bytes land in PINNED host memory: grail_synthesize_batch renders rows in
        // blocks of up to 4096 utterances and copies each block out on a second stream while the next one
        // renders; a pinned destination receives those copies directly (measured 53 GB/s end to end = 93 %
        // of a plain pinned hipMemcpy, profiles/r04_host_output.txt; the hipHostMalloc / hipHostFree of the
        // block itself is not in that figure and costs about a second per 10 GB).  Larger results, and hosts
        // that refuse the pinned allocation (locked-memory limit, little free RAM), take a plain Vec — the
        // library then feeds it through its own ring of pinned staging buffers and copier threads (~49 GB/s).
        const PINNED_LIMIT: usize = 2 << 30;
        let floats = n as usize * stride as usize;
        let mut pinned: *mut std::ffi::c_void = std::ptr::null_mut();
        let have_pinned = floats * 4 <= PINNED_LIMIT
            && unsafe { sys::grail_host_alloc(gpu.ctx, floats * 4, &mut pinned) } == 0
            && !pinned.is_null();
        let mut pageable: Vec<f32> = if have_pinned { Vec::new() } else { vec![0f32; floats] };
        let dst = if have_pinned { pinned as *mut f32 } else { pageable.as_mut_ptr() };
        let r = check(unsafe {
            sys::grail_synthesize_batch(gpu.ctx, b.segs.as_ptr(), b.offs.as_ptr(), b.vids.as_ptr(),
                                        b.seeds.as_ptr(), n, dst, stride,
                                        lens.as_mut_ptr(), sys::GRAIL_OUT_HOST)
        });
        let result = r.map(|_| rows_of(unsafe { std::slice::from_raw_parts(dst as *const f32, floats) }, stride as usize, &lens));
        if have_pinned {
            unsafe { sys::grail_host_free(gpu.ctx, pinned) };
        }
        drop(pageable);
        result
    }
}

/// Every GPU of the node behind the same call (grail_node_*): one context and one host thread per device inside the
/// library, the voice table carried to the other GPUs' HBM by one ncclBroadcast over xGMI, the batch cut into contiguous
/// shards (`grail_shard_range`) that render concurrently into slices of one host buffer — the reference's single call
/// (`examples/cli.rs:175-184`) for a host that holds 524 288 utterances and eight MI355X.  No data-path collective:
/// per-utterance state is self-contained (src/lib.rs:470-488, 724-748, 839-854).
pub struct Node {
    node: *mut sys::grail_node,
}

impl Node {
    /// `Node::new(&[0, 1, 2, 3, 4, 5, 6, 7], &[voice])`
    pub fn new(devices: &[i32], voices: &[Voice]) -> Result<Self, Error> {
        let have = unsafe { sys::grail_abi_version() };
        if have != sys::GRAIL_ABI_VERSION {
            return Err(Error { status: sys::GRAIL_ERR_INVALID_ARG,
                               message: format!("libgrail_hip.so has ABI version {have}, grail-hip-sys mirrors {}", sys::GRAIL_ABI_VERSION) });
        }
        let mut node = std::ptr::null_mut();
        check(unsafe { sys::grail_node_create(devices.as_ptr(), devices.len() as u32, &mut node) })?;
        let n = Node { node };
        let table: Vec<_> = voices.iter().map(voice_to_c).collect();
        check(unsafe { sys::grail_node_set_voices(n.node, table.as_ptr(), table.len() as u32) })?;
        Ok(n)
    }

    pub fn devices(&self) -> u32 {
        unsafe { sys::grail_node_size(self.node) }
    }

    /// What RCCL itself reports for the node's communicator (`ncclCommCount`): the number of GPUs that met.
    pub fn rccl_ranks(&self) -> Result<u32, Error> {
        let mut v = 0i64;
        check(unsafe { sys::grail_node_get_option(self.node, b"node_rccl_ranks\0".as_ptr() as *const _, &mut v) })?;
        Ok(v as u32)
    }

    pub fn set_arithmetic(&self, a: Arithmetic) -> Result<(), Error> {
        let v = match a { Arithmetic::Exact => 0, Arithmetic::Fast => 1 };
        check(unsafe { sys::grail_node_set_option(self.node, b"arithmetic\0".as_ptr() as *const _, v) })
    }
}

impl Drop for Node {
    fn drop(&mut self) {
        unsafe { sys::grail_node_destroy(self.node) };
    }
}

/// `utterances.synthesize_batch(&node)`: the same result as the `Gpu` form, rows in the caller's order.
impl SynthesisTarget for Node {
    fn render(&self, b: &FlatBatch) -> Result<Vec<Vec<f32>>, Error> {
        let node = self;
        let n = b.len();
        let mut lens = vec![0u32; n as usize];
        // the Sequencer clock pre-pass (src/lib.rs:861-888), sharded like the synthesis: sizes the rows
        check(unsafe { sys::grail_node_lengths(node.node, b.segs.as_ptr(), b.offs.as_ptr(), b.vids.as_ptr(), n, u32::MAX,
                                               lens.as_mut_ptr()) })?;
        let stride = (*lens.iter().max().unwrap_or(&0) as u64 + 63) / 64 * 64;
        let floats = n as usize * stride as usize;
        // pinned memory every device can copy into (hipHostMallocPortable) while it fits; else a plain Vec, fed through
        // each context's ring of pinned staging buffers
        const PINNED_LIMIT: usize = 16 << 30;
        let mut pinned: *mut std::ffi::c_void = std::ptr::null_mut();
        let have_pinned = floats * 4 <= PINNED_LIMIT
            && unsafe { sys::grail_node_host_alloc(node.node, floats * 4, &mut pinned) } == 0
            && !pinned.is_null();
        let mut pageable: Vec<f32> = if have_pinned { Vec::new() } else { vec![0f32; floats] };
        let dst = if have_pinned { pinned as *mut f32 } else { pageable.as_mut_ptr() };
        let r = check(unsafe {
            sys::grail_node_synthesize_batch(node.node, b.segs.as_ptr(), b.offs.as_ptr(), b.vids.as_ptr(),
                                             b.seeds.as_ptr(), n, dst, stride, lens.as_mut_ptr(), sys::GRAIL_OUT_HOST)
        });
        let result = r.map(|_| rows_of(unsafe { std::slice::from_raw_parts(dst as *const f32, floats) }, stride as usize, &lens));
        if have_pinned {
            unsafe { sys::grail_node_host_free(node.node, pinned) };
        }
        drop(pageable);
        result
    }
}

/// Arithmetic of the synthesis kernels: `Exact` (default) is bit-identical to the CPU iterator chain;
/// `Fast` is the stated-tolerance mode (|fast - exact| <= GRAIL_FAST_TOLERANCE = 64 * 2^-23 of full
/// scale, measured 18 * 2^-23; clock, phases, wraps and noise generators stay exact): 2.3x the
/// throughput on large batches, 2x on batches of a few thousand utterances (time-split kernels), 5x on batches
/// of a few hundred (time-parallel scan kernel).
pub enum Arithmetic { Exact, Fast }

impl Gpu {
    pub fn set_arithmetic(&self, a: Arithmetic) -> Result<(), Error> {
        let v = match a { Arithmetic::Exact => 0, Arithmetic::Fast => 1 };
        check(unsafe { sys::grail_set_option(self.ctx, b"arithmetic\0".as_ptr() as *const _, v) })
    }

    /// Whether `Arithmetic::Fast` is served for the current voice table (see [`fast_sharpness`]); sharper
    /// tables are rendered by the exact kernels whatever `set_arithmetic` says.
    pub fn fast_arithmetic_served(&self) -> Result<bool, Error> {
        let mut v = 0i64;
        check(unsafe { sys::grail_get_option(self.ctx, b"fast_arithmetic_served\0".as_ptr() as *const _, &mut v) })?;
        Ok(v != 0)
    }
}

/// Predicted |fast - reference| of `voice` in units of 2^-23 of max(1, peak) (grail_fast_sharpness): narrow and
/// high formants amplify rounding-level differences of the filter coefficients.  Fast arithmetic is served up to
/// GRAIL_FAST_SHARPNESS_LIMIT = 28 (`voices::generic()`: 24).  Pure host function, no GPU.
pub fn fast_sharpness(voice: &Voice) -> f32 {
    unsafe { sys::grail_fast_sharpness(&voice_to_c(voice)) }
}

/// Warm-up length of `voice` for the time-split fast kernels in samples (0: the voice does not qualify).
pub fn time_split_warmup(voice: &Voice) -> u32 {
    unsafe { sys::grail_time_split_warmup(&voice_to_c(voice)) }
}

/// The lazy source of `examples/interactive.rs:31-48` on the GPU: ONE chain for a whole session.  Segments are
/// appended while samples are pulled; a Sequencer that needs a segment which has not been appended yet pauses
/// (`src/lib.rs:866-888` pulls `iter.next()` on demand) and `next` comes back short — the front end then feeds it, a
/// `Phoneme::Silence` when no text is waiting, as the reference's `repeat_with(|| receiver.try_recv().unwrap_or(' '))`
/// does.  Carrier phase, noise seed, jitter and filter state carry across everything appended: the samples are those
/// of the CPU iterator chain over the concatenated list, bit for bit.
pub struct LiveStream<'a> {
    gpu: &'a Gpu,
    stream: *mut sys::grail_stream,
    d_out: *mut std::ffi::c_void,
    d_len: *mut std::ffi::c_void,
    chunk: u32,
    stride: u64,
}

impl<'a> LiveStream<'a> {
    /// `.sequence(voice).jitter(jitter_seed, voice).synthesize()` over a source that is still being written.
    pub fn new(gpu: &'a Gpu, chunk: u32, voice: u32, jitter_seed: u32) -> Result<Self, Error> {
        let stride = (chunk as u64 + 63) / 64 * 64;
        let mut s = LiveStream { gpu, stream: std::ptr::null_mut(), d_out: std::ptr::null_mut(),
                                 d_len: std::ptr::null_mut(), chunk, stride };
        unsafe {
            check(sys::grail_stream_open_live(gpu.ctx, 1, &voice, &jitter_seed, 0, 0, &mut s.stream))?;
            check(sys::grail_device_alloc(gpu.ctx, stride as usize * 4, &mut s.d_out))?;
            check(sys::grail_device_alloc(gpu.ctx, 4, &mut s.d_len))?;
        }
        Ok(s)
    }

    /// The source delivers: these segments follow what the chain already has.
    pub fn append(&mut self, phonemes: &[PhonemeElem]) -> Result<(), Error> {
        let segs: Vec<_> = phonemes.iter().map(|p| sys::grail_phoneme_elem {
            phoneme: p.phoneme as i32, length: p.length, blend_length: p.blend_length, frequency: p.frequency,
        }).collect();
        let offs = [0u32, segs.len() as u32];
        check(unsafe { sys::grail_stream_append(self.gpu.ctx, self.stream, segs.as_ptr(), offs.as_ptr()) })
    }

    /// The source has ended: what is pending is spoken, the last segment fades out, `next` then returns nothing.
    pub fn finish(&mut self) -> Result<(), Error> {
        check(unsafe { sys::grail_stream_finish(self.gpu.ctx, self.stream, std::ptr::null()) })
    }

    /// Segments appended that the Sequencer has not pulled yet.
    pub fn pending(&mut self) -> Result<u32, Error> {
        let mut n = 0u32;
        check(unsafe { sys::grail_stream_pending(self.gpu.ctx, self.stream, &mut n) })?;
        Ok(n)
    }

    /// `Iterator::next`, up to `chunk` samples at a time: fewer when the Sequencer waits for its source (or the
    /// chain has ended).
    pub fn next(&mut self) -> Result<Vec<f32>, Error> {
        let mut n = 0u32;
        unsafe {
            check(sys::grail_stream_next_async(self.gpu.ctx, self.stream, self.chunk, self.d_out as *mut f32,
                                               self.stride, self.d_len as *mut u32))?;
            check(sys::grail_sync(self.gpu.ctx))?;
            check(sys::grail_memcpy_d2h(self.gpu.ctx, &mut n as *mut u32 as *mut _, self.d_len, 4))?;
            let mut out = vec![0f32; n as usize];
            if n > 0 {
                check(sys::grail_memcpy_d2h(self.gpu.ctx, out.as_mut_ptr() as *mut _, self.d_out, n as usize * 4))?;
            }
            Ok(out)
        }
    }
}

impl Drop for LiveStream<'_> {
    fn drop(&mut self) {
        unsafe {
            if !self.stream.is_null() { sys::grail_stream_close(self.gpu.ctx, self.stream); }
            if !self.d_out.is_null() { sys::grail_device_free(self.gpu.ctx, self.d_out); }
            if !self.d_len.is_null() { sys::grail_device_free(self.gpu.ctx, self.d_len); }
        }
    }
}
