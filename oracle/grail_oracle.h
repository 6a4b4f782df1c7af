/*
 * grail_oracle.h — CPU ORACLE for the grail-rs synthesis hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load liboracle.so; the product library (libgrail_hip.so) never links,
 * includes or calls it.
 *
 * PARITY UNPINNED BY THE REFERENCE: grail-rs is Rust, rustc/cargo are absent
 * from this image, and the reference's three hot-path tests are empty bodies
 * (src/lib.rs:603-608, 804-805).  The oracle is therefore pinned by
 *   (1) hand-derivable known answers (tests/test_oracle_kat.py),
 *   (2) an independent numpy-float32 restatement (tests/np_model.py),
 *   (3) the six literal Transcriber tests of the reference (src/lib.rs:1210-1358)
 *       for the text front half,
 * and not by outputs of the reference itself.
 *
 * All citations are file:line in the reference tree (/root/reference).
 * Arithmetic: IEEE-754 binary32, round-to-nearest-even, NO fused multiply-add,
 * no reassociation (build with -ffp-contract=off -fno-fast-math).
 */
#ifndef GRAIL_ORACLE_H
#define GRAIL_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_NUM_FORMANTS 8            /* src/lib.rs:24 */
#define ORC_DEFAULT_SAMPLE_RATE 44100.0f /* src/lib.rs:21 */

/* Phoneme enum order: src/lib.rs:632-649 with make_phonemes!(A, E) src/lib.rs:686-689 */
enum {
    ORC_PH_SILENCE = 0,
    ORC_PH_STOP    = 1,
    ORC_PH_GLIDE   = 2,
    ORC_PH_A       = 3,
    ORC_PH_E       = 4,
    ORC_PH_COUNT   = 5
};
#define ORC_NUM_VOICED 2 /* VoiceStorage fields a, e  (src/lib.rs:653-659) */

/* Array: src/lib.rs:88 */
typedef struct { float v[ORC_NUM_FORMANTS]; } orc_array;

/* SynthesisElem: src/lib.rs:316-337 (field order as declared) */
typedef struct {
    float     frequency;
    orc_array formant_freq;
    orc_array formant_bw;
    orc_array formant_smooth;
    orc_array formant_breath;
    orc_array formant_turb;
    orc_array formant_amp;
} orc_synthesis_elem;

/* Voice: src/lib.rs:696-717; VoiceStorage {a, e}: src/lib.rs:653-659 */
typedef struct {
    float sample_rate;
    orc_synthesis_elem phonemes[ORC_NUM_VOICED]; /* [0]=a, [1]=e */
    float center_frequency;
    float jitter_frequency;
    float jitter_delta_frequency;
    float jitter_delta_formant_frequency;
    float jitter_delta_amplitude;
} orc_voice;

/* PhonemeElem: src/lib.rs:961-973 */
typedef struct {
    int32_t phoneme;
    float   length;
    float   blend_length;
    float   frequency;
} orc_phoneme_elem;

/* SequenceElem: src/lib.rs:814-824 (Option<SynthesisElem> as has_elem + elem) */
typedef struct {
    int32_t            has_elem;
    orc_synthesis_elem elem;
    float              length;
    float              blend_length;
} orc_sequence_elem;

/* TranscriptionRule: src/lib.rs:1030-1036; strings are UTF-32 code points */
typedef struct {
    const uint32_t *string;
    uint32_t        string_len;
    const int32_t  *phonemes;
    uint32_t        n_phonemes;
} orc_rule;

/* ---- numeric helpers ---------------------------------------------------- */
float orc_random_f32(uint32_t *state);          /* src/lib.rs:36-55 */
float orc_tan_approx(float x);                  /* src/lib.rs:63-70 */
float orc_exp_approx(float x);                  /* src/lib.rs:75-82 */
void orc_set_sum_identity(int negative_zero);   /* tests: Iterator::sum identity of Rust >= 1.83 */
/* Tests only: per-formant arithmetic in double precision on the same f32 parameter track (the
 * rounding-noise yardstick for the product's tolerance mode; never the parity target). */
void orc_set_precise(int on);
float orc_array_sum(const orc_array *a);        /* src/lib.rs:123-125 */

/* ---- SynthesisElem algebra --------------------------------------------- */
void orc_elem_silent(orc_synthesis_elem *out);  /* src/lib.rs:367-377 */
void orc_elem_new_phoneme(orc_synthesis_elem *out,
                          const float *freq, const float *bw, const float *smooth,
                          const float *turb, const float *breath,
                          const float *amp);    /* src/lib.rs:381-401 */
void orc_elem_resample(orc_synthesis_elem *e, float old_rate, float new_rate); /* :418-440 */
void orc_elem_blend(orc_synthesis_elem *out, const orc_synthesis_elem *self,
                    const orc_synthesis_elem *other, float alpha);             /* :404-414 */

/* ---- voices -------------------------------------------------------------- */
void orc_voice_generic(orc_voice *out);         /* src/voices/generic.rs:5-40 */
/* The build's own 48 kHz (or any-rate) variant of generic(): SURVEY.md §8d.
 * Each phoneme elem .resample(44100, rate) (src/lib.rs:418 via for_all :674),
 * sample_rate = rate, scalars recomputed as 120/rate, 16/rate, 6/rate, 6/rate, 0.2
 * (cf. src/voices/generic.rs:34-38). */
void orc_voice_generic_at(orc_voice *out, float sample_rate);

/* ---- the hot path: select -> sequence -> jitter -> synthesize ---------- */
/* Equivalent of
 *   segs.into_iter().select(voice).sequence(voice).jitter(seed, voice).synthesize().collect()
 * (src/lib.rs:1013, 941, 786, 587).  Writes at most cap samples to out (out may
 * be NULL with cap 0 to count only); returns the number of samples the
 * iterator chain produces. */
uint64_t orc_synthesize_phonemes(const orc_voice *voice,
                                 const orc_phoneme_elem *segs, uint32_t n_segs,
                                 uint32_t jitter_seed, float *out, uint64_t cap);

/* Same, starting from explicit SequenceElems (skips Selector). */
uint64_t orc_synthesize_sequence(const orc_voice *voice,
                                 const orc_sequence_elem *segs, uint32_t n_segs,
                                 uint32_t jitter_seed, float *out, uint64_t cap);

/* Debug taps for tests: the per-sample SynthesisElem after Sequencer
 * (stage 0) or after Jitter (stage 1), 49 floats per sample. */
uint64_t orc_trace_elems(const orc_voice *voice,
                         const orc_phoneme_elem *segs, uint32_t n_segs,
                         uint32_t jitter_seed, int stage,
                         float *out49, uint64_t cap_samples);

/* Batch driver used by bench.py's cpu_baseline leg and the parity tests:
 * utterance u uses segs[seg_offsets[u] .. seg_offsets[u+1]), voices[voice_ids[u]],
 * jitter_seeds[u]; writes out + u*out_stride, out_len[u]. Single thread. */
void orc_synthesize_batch(const orc_voice *voices, uint32_t n_voices,
                          const orc_phoneme_elem *segs, const uint32_t *seg_offsets,
                          const uint32_t *voice_ids, const uint32_t *jitter_seeds,
                          uint32_t n_utt, float *out, uint64_t out_stride,
                          uint32_t *out_len);
/* Same, utterances handed out to n_threads pthreads (bench baseline only; returns threads started). */
int orc_synthesize_batch_threads(const orc_voice *voices, uint32_t n_voices,
                                 const orc_phoneme_elem *segs, const uint32_t *seg_offsets,
                                 const uint32_t *voice_ids, const uint32_t *jitter_seeds,
                                 uint32_t n_utt, float *out, uint64_t out_stride,
                                 uint32_t *out_len, uint32_t n_threads);

/* ---- text front half --------------------------------- */
/* Transcriber::next loop, src/lib.rs:1116-1191.  leading_silence != 0 seeds
 * the buffer with SILENCE as .transcribe() does (src/lib.rs:1201); 0 starts
 * with an empty buffer as the reference's unit tests do (src/lib.rs:1212-1225). */
uint32_t orc_transcribe(const uint32_t *text, uint32_t text_len,
                        const orc_rule *rules, uint32_t n_rules,
                        int case_sensitive, int leading_silence,
                        int32_t *out_phonemes, uint32_t cap);
/* languages::generic(), src/languages/mod.rs:4-34 */
uint32_t orc_language_generic(const orc_rule **rules, int *case_sensitive);
/* Intonator::next, src/lib.rs:1057-1075 */
void orc_intonate(const orc_voice *voice, const int32_t *phonemes, uint32_t n,
                  orc_phoneme_elem *out);
/* Full chain of examples/cli.rs:175-184 on ASCII/UTF-32 text. */
uint64_t orc_say(const orc_voice *voice, const uint32_t *text, uint32_t text_len,
                 uint32_t jitter_seed, float *out, uint64_t cap);
/* f32 -> i16 as examples/cli.rs:49 (saturating `as`, NaN -> 0). */
int16_t orc_pcm16(float x);

#ifdef __cplusplus
}
#endif
#endif
