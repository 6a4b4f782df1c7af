//! Writes, for every parity case of tests/golden/make_golden.py and for the first six utterances of the
//! bench corpus at full length (BASELINE config 3), what the reference crate itself renders: `<case>.f32` (little-endian f32 samples of the whole track) and `voice_44k.f32` /
//! `voice_48k.f32` (the voice tables in the field order of `grail_voice`, include/grail_hip.h),
//! plus `manifest.txt` (`name length` per line) and `toolchain.txt`.
//!
//! The chain is the one of examples/cli.rs:175-184 after the text front end:
//! `.select(v).sequence(v).jitter(seed, v).synthesize()`.
use std::cell::RefCell;
use std::fs;
use std::io::Write;
use std::path::Path;

use grail_rs::{
    Array, IntoJitter, IntoSelector, IntoSequencer, IntoSynthesize, Phoneme, PhonemeElem,
    SynthesisElem, Voice, DEFAULT_SAMPLE_RATE,
};

/// `voices::generic()` at another sample rate, as SURVEY.md §8d defines the 48 kHz voice: every
/// phoneme elem `.resample(44100, rate)` (src/lib.rs:418) and the five scalars recomputed the way
/// src/voices/generic.rs:34-38 computes them.
fn voice_at(rate: f32) -> Voice {
    let mut v = grail_rs::voices::generic();
    if rate != DEFAULT_SAMPLE_RATE {
        v.phonemes.a = v.phonemes.a.resample(DEFAULT_SAMPLE_RATE, rate);
        v.phonemes.e = v.phonemes.e.resample(DEFAULT_SAMPLE_RATE, rate);
        v.sample_rate = rate;
        v.center_frequency = 120.0 / rate;
        v.jitter_frequency = 16.0 / rate;
        v.jitter_delta_frequency = 6.0 / rate;
        v.jitter_delta_formant_frequency = 6.0 / rate;
        v.jitter_delta_amplitude = 0.2;
    }
    v
}

/// `Array` keeps its storage private; `map` visits the elements in index order.
fn array_values(a: Array) -> Vec<f32> {
    let out = RefCell::new(Vec::with_capacity(8));
    a.map(|x| {
        out.borrow_mut().push(x);
        x
    });
    out.into_inner()
}

fn elem_values(e: SynthesisElem, out: &mut Vec<f32>) {
    out.push(e.frequency);
    for a in [e.formant_freq, e.formant_bw, e.formant_smooth, e.formant_breath, e.formant_turb, e.formant_amp] {
        out.extend(array_values(a));
    }
}

/// the layout of `grail_voice`: sample_rate, phonemes a and e (49 floats each), five scalars
fn voice_values(v: Voice) -> Vec<f32> {
    let mut out = vec![v.sample_rate];
    elem_values(v.phonemes.a, &mut out);
    elem_values(v.phonemes.e, &mut out);
    out.extend([
        v.center_frequency,
        v.jitter_frequency,
        v.jitter_delta_frequency,
        v.jitter_delta_formant_frequency,
        v.jitter_delta_amplitude,
    ]);
    out
}

fn write_f32(path: &Path, data: &[f32]) {
    let mut f = fs::File::create(path).expect("create output file");
    for x in data {
        f.write_all(&x.to_le_bytes()).expect("write");
    }
}

fn seg(phoneme: Phoneme, length: f32, blend_length: f32, frequency: f32) -> PhonemeElem {
    PhonemeElem { phoneme, length, blend_length, frequency }
}

fn main() {
    let dir = std::env::args().nth(1).expect("usage: reference-golden <output directory>");
    let dir = Path::new(&dir);
    fs::create_dir_all(dir).expect("create output directory");

    use Phoneme::{Glide as GL, Silence as S, A, E};
    let f44: f32 = 120.0 / 44100.0;
    let f48: f32 = 120.0 / 48000.0;
    // name, sample rate, segments, jitter seed — tests/golden/make_golden.py::cases()
    let cases: Vec<(&str, f32, Vec<PhonemeElem>, u32)> = vec![
        ("text_a_head", 44100.0, vec![seg(S, 0.5, 0.5, f44), seg(A, 0.5, 0.5, f44)], 0),
        ("a_e_48k", 48000.0, vec![seg(A, 0.01, 0.01, f48), seg(E, 0.01, 0.01, f48)], 1),
        ("fade_in_out_48k", 48000.0, vec![seg(S, 0.008, 0.008, f48), seg(E, 0.008, 0.008, f48)], 2),
        (
            "mixed_44k",
            44100.0,
            vec![
                seg(E, 0.006, 0.003, 0.004),
                seg(S, 0.004, 0.004, 0.1),
                seg(A, 0.006, 0.012, 0.002),
                seg(GL, 0.002, 0.002, 0.1),
                seg(E, 0.005, 0.005, 0.003),
            ],
            12345,
        ),
        ("pitch_clamp_48k", 48000.0, vec![seg(A, 0.004, 0.004, 0.7), seg(E, 0.004, 0.004, 0.5)], 3),
        ("wrap_48k", 48000.0, vec![seg(A, 0.07, 0.07, f48), seg(E, 0.07, 0.07, 0.0031)], 4242),
    ];

    // the first six utterances of the bench corpus at full length (BASELINE config 3: 4 x 0.5 s at
    // 48 kHz, grail_hip/workload.py::make_batch): utterance u hashes with the crate's own LCG step
    // (src/lib.rs:40) from 0x9E3779B9 ^ u; segment 0 is Silence, the others draw from (A, E, Silence);
    // pitch 100..200 Hz; jitter seed = u.
    let lcg = |s: u32| s.wrapping_mul(16807).wrapping_add(1);
    let mut bench_cases: Vec<(String, f32, Vec<PhonemeElem>, u32)> = Vec::new();
    for u in 0u32..6 {
        let mut s = u ^ 0x9E37_79B9;
        let mut segs = Vec::new();
        for i in 0..4 {
            s = lcg(s);
            let ph = if i == 0 { S } else { [A, E, S][((s >> 16) % 3) as usize] };
            s = lcg(s);
            let hz = (100 + (s >> 16) % 101) as f32;
            segs.push(seg(ph, 0.5, 0.5, hz / 48000.0));
        }
        bench_cases.push((format!("bench_u{u}"), 48000.0, segs, u));
    }

    let mut manifest = String::new();
    for (name, rate, segs, seed) in bench_cases {
        let v = voice_at(rate);
        let pcm: Vec<f32> = segs.into_iter().select(v).sequence(v).jitter(seed, v).synthesize().collect();
        write_f32(&dir.join(format!("{name}.f32")), &pcm);
        manifest.push_str(&format!("{name} {}\n", pcm.len()));
    }
    for (name, rate, segs, seed) in cases {
        let v = voice_at(rate);
        let pcm: Vec<f32> = segs.into_iter().select(v).sequence(v).jitter(seed, v).synthesize().collect();
        write_f32(&dir.join(format!("{name}.f32")), &pcm);
        manifest.push_str(&format!("{name} {}\n", pcm.len()));
    }
    write_f32(&dir.join("voice_44k.f32"), &voice_values(voice_at(44100.0)));
    write_f32(&dir.join("voice_48k.f32"), &voice_values(voice_at(48000.0)));
    fs::write(dir.join("manifest.txt"), manifest).expect("write manifest");
    // Iterator::sum::<f32>() starts from +0.0 up to Rust 1.82 and from -0.0 since 1.83
    let toolchain = std::process::Command::new("rustc")
        .arg("--version")
        .output()
        .ok()
        .map(|o| String::from_utf8_lossy(&o.stdout).into_owned())
        .unwrap_or_else(|| "unknown\n".to_string());
    fs::write(dir.join("toolchain.txt"), toolchain).expect("write toolchain note");
    println!("wrote {}", dir.display());
}
