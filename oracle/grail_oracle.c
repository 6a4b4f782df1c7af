/*
 * grail_oracle.c — CPU ORACLE (test infrastructure only; see grail_oracle.h).
 *
 * A plain-C restatement of the grail-rs iterator chain
 *     Selector -> Sequencer -> Jitter -> Synthesize
 * (reference src/lib.rs), one pull-style `next` function per adapter, same
 * struct fields, same evaluation order.  PARITY UNPINNED by the reference's
 * own tests (they are empty for this path) — see the header.
 *
 * Build: gcc -O3 -ffp-contract=off -fno-fast-math -fexcess-precision=standard
 * (rustc never contracts a*b+c to an FMA and never reassociates).
 */
#include <pthread.h>
#include "grail_oracle.h"

#include <math.h>
#include <string.h>

#define NF ORC_NUM_FORMANTS

/* ------------------------------------------------------------------------ */
/* helpers: src/lib.rs:31-82                                                */
/* ------------------------------------------------------------------------ */

/* src/lib.rs:36-55 */
float orc_random_f32(uint32_t *state)
{
    /* :40  wrapping_mul(16807).wrapping_add(1) */
    *state = (*state) * 16807u + 1u;
    /* :50 */
    uint32_t res = (*state >> 9) | 0x3F800000u;
    float f;
    memcpy(&f, &res, sizeof f);
    /* :54 */
    return (f - 1.5f) * 2.0f;
}

/* src/lib.rs:63-70 */
float orc_tan_approx(float x)
{
    /* Rust's `*` is left-associative: ((1-x)*x)*(...),  (4*(x+.5))*(.5-x),
     * ((x+.5)*(5 - (4*(1-x))*x))*(.5-x) */
    return ((1.0f - x) * x * (5.0f - 4.0f * (x + 0.5f) * (0.5f - x)))
         / ((x + 0.5f) * (5.0f - 4.0f * (1.0f - x) * x) * (0.5f - x));
}

/* src/lib.rs:75-82 */
float orc_exp_approx(float x)
{
    float o = 1.0f - x;
    float o2 = o * o;
    return o2 * o2 * o;
}

/* src/lib.rs:123-125: self.0.iter().sum::<f32>() — a sequential left fold.
 * The fold's identity is +0.0 up to Rust 1.82 and -0.0 from 1.83; the two
 * differ only when every term is -0.0, and then only in the sign of zero. */
static float g_sum_identity = 0.0f;

/* Tests only: fold from -0.0 (Rust >= 1.83) instead of +0.0, to show the rendering does not depend
 * on which toolchain built the crate. */
void orc_set_sum_identity(int negative_zero) { g_sum_identity = negative_zero ? -0.0f : 0.0f; }

float orc_array_sum(const orc_array *a)
{
    float s = g_sum_identity;
    for (int i = 0; i < NF; ++i) s = s + a->v[i];
    return s;
}

/* f32::min (core): if one argument is NaN the other is returned == C fminf */
static inline float f32_min(float a, float b) { return fminf(a, b); }

/* Array::blend src/lib.rs:135-137 : a*(1-alpha) + b*alpha */
static inline float blend1(float a, float b, float alpha)
{
    return a * (1.0f - alpha) + b * alpha;
}

static void array_blend(orc_array *out, const orc_array *a, const orc_array *b, float alpha)
{
    for (int i = 0; i < NF; ++i) out->v[i] = blend1(a->v[i], b->v[i], alpha);
}

static void array_splat(orc_array *out, float x)
{
    for (int i = 0; i < NF; ++i) out->v[i] = x;
}

/* ------------------------------------------------------------------------ */
/* SynthesisElem: src/lib.rs:341-460                                        */
/* ------------------------------------------------------------------------ */

/* src/lib.rs:367-377 */
void orc_elem_silent(orc_synthesis_elem *e)
{
    e->frequency = 0.25f;
    array_splat(&e->formant_freq, 0.25f);
    array_splat(&e->formant_bw, 0.25f);
    array_splat(&e->formant_smooth, 0.25f);
    array_splat(&e->formant_breath, 0.0f);
    array_splat(&e->formant_turb, 0.0f);
    array_splat(&e->formant_amp, 0.0f);
}

/* src/lib.rs:418-440 */
void orc_elem_resample(orc_synthesis_elem *e, float old_rate, float new_rate)
{
    /* :420 */
    float scale = old_rate / new_rate;
    orc_synthesis_elem r = *e; /* ..self : breath, turb untouched */
    for (int i = 0; i < NF; ++i) {
        /* :423 the un-clamped scaled frequency decides the amp drop */
        float ff = e->formant_freq.v[i] * scale;
        /* :428 */
        r.formant_freq.v[i] = f32_min(e->formant_freq.v[i] * scale, 0.5f);
        /* :429-430 */
        r.formant_bw.v[i] = e->formant_bw.v[i] * scale;
        r.formant_smooth.v[i] = e->formant_smooth.v[i] * scale;
        /* :433-435 */
        r.formant_amp.v[i] = (ff > 0.5f) ? 0.0f : e->formant_amp.v[i];
    }
    /* :427 */
    r.frequency = f32_min(e->frequency * scale, 0.5f);
    *e = r;
}

/* src/lib.rs:381-401.  Argument order (freq, bw, smooth, turb, breath, amp). */
void orc_elem_new_phoneme(orc_synthesis_elem *out,
                          const float *freq, const float *bw, const float *smooth,
                          const float *turb, const float *breath, const float *amp)
{
    orc_array a;
    memcpy(a.v, amp, sizeof a.v);
    float total = orc_array_sum(&a); /* :398 */
    out->frequency = 0.0f;           /* :390 */
    for (int i = 0; i < NF; ++i) {
        out->formant_freq.v[i] = freq[i];
        out->formant_bw.v[i] = bw[i];
        out->formant_smooth.v[i] = smooth[i];
        out->formant_breath.v[i] = breath[i];
        out->formant_turb.v[i] = turb[i];
        out->formant_amp.v[i] = amp[i] / total; /* :398 */
    }
    orc_elem_resample(out, 1.0f, ORC_DEFAULT_SAMPLE_RATE); /* :400 */
}

/* src/lib.rs:404-414 */
void orc_elem_blend(orc_synthesis_elem *out, const orc_synthesis_elem *self,
                    const orc_synthesis_elem *other, float alpha)
{
    orc_synthesis_elem r;
    r.frequency = blend1(self->frequency, other->frequency, alpha);
    array_blend(&r.formant_freq, &self->formant_freq, &other->formant_freq, alpha);
    array_blend(&r.formant_smooth, &self->formant_smooth, &other->formant_smooth, alpha);
    array_blend(&r.formant_bw, &self->formant_bw, &other->formant_bw, alpha);
    array_blend(&r.formant_turb, &self->formant_turb, &other->formant_turb, alpha);
    array_blend(&r.formant_breath, &self->formant_breath, &other->formant_breath, alpha);
    array_blend(&r.formant_amp, &self->formant_amp, &other->formant_amp, alpha);
    *out = r;
}

/* src/lib.rs:445-450 */
static orc_synthesis_elem elem_copy_with_frequency(orc_synthesis_elem e, float frequency)
{
    e.frequency = f32_min(frequency, 0.5f);
    return e;
}

/* src/lib.rs:454-459 */
static orc_synthesis_elem elem_copy_silent(orc_synthesis_elem e)
{
    array_splat(&e.formant_amp, 0.0f);
    return e;
}

/* ------------------------------------------------------------------------ */
/* voices: src/voices/generic.rs:5-40, src/voices/mod.rs:7-14               */
/* ------------------------------------------------------------------------ */

void orc_voice_generic(orc_voice *v)
{
    /* MKPHON(freq, bw, smooth, turb, breath, amp) */
    static const float a_freq[NF] = {910.0f, 1271.0f, 2851.0f, 3213.0f, 1200.0f, 2000.0f, 3000.0f, 4000.0f};
    static const float a_bw[NF] = {60.0f, 160.0f, 180.0f, 200.0f, 100.0f, 100.0f, 100.0f, 100.0f};
    static const float a_smooth[NF] = {1600.0f, 1600.0f, 1600.0f, 1600.0f, 1600.0f, 1600.0f, 1600.0f, 1600.0f};
    static const float a_turb[NF] = {0.2f, 0.2f, 0.1f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    static const float a_breath[NF] = {0.5f, 0.2f, 0.05f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    static const float a_amp[NF] = {0.3f, 0.3f, 0.2f, 0.1f, 0.0f, 0.0f, 0.0f, 0.0f};

    static const float e_freq[NF] = {910.0f, 1871.0f, 2851.0f, 3213.0f, 1200.0f, 2000.0f, 3000.0f, 4000.0f};
    static const float e_bw[NF] = {80.0f, 180.0f, 180.0f, 200.0f, 100.0f, 100.0f, 100.0f, 100.0f};
    static const float e_smooth[NF] = {1600.0f, 1600.0f, 1600.0f, 1600.0f, 1600.0f, 1600.0f, 1600.0f, 1600.0f};
    static const float e_turb[NF] = {0.2f, 0.4f, 0.4f, 0.4f, 0.4f, 0.4f, 0.4f, 0.4f};
    static const float e_breath[NF] = {1.0f, 1.0f, 1.0f, 1.0f, 1.0f, 1.0f, 0.1f, 0.1f};
    static const float e_amp[NF] = {0.5f, 0.4f, 0.3f, 0.2f, 0.0f, 0.0f, 0.0f, 0.0f};

    v->sample_rate = ORC_DEFAULT_SAMPLE_RATE;
    orc_elem_new_phoneme(&v->phonemes[0], a_freq, a_bw, a_smooth, a_turb, a_breath, a_amp);
    orc_elem_new_phoneme(&v->phonemes[1], e_freq, e_bw, e_smooth, e_turb, e_breath, e_amp);
    v->center_frequency = 120.0f / ORC_DEFAULT_SAMPLE_RATE;               /* generic.rs:34 */
    v->jitter_frequency = 16.0f / ORC_DEFAULT_SAMPLE_RATE;                /* :35 */
    v->jitter_delta_frequency = 6.0f / ORC_DEFAULT_SAMPLE_RATE;           /* :36 */
    v->jitter_delta_formant_frequency = 6.0f / ORC_DEFAULT_SAMPLE_RATE;   /* :37 */
    v->jitter_delta_amplitude = 0.2f;                                     /* :38 */
}

void orc_voice_generic_at(orc_voice *v, float sample_rate)
{
    orc_voice_generic(v);
    if (sample_rate == ORC_DEFAULT_SAMPLE_RATE) return;
    for (int p = 0; p < ORC_NUM_VOICED; ++p)
        orc_elem_resample(&v->phonemes[p], ORC_DEFAULT_SAMPLE_RATE, sample_rate);
    v->sample_rate = sample_rate;
    v->center_frequency = 120.0f / sample_rate;
    v->jitter_frequency = 16.0f / sample_rate;
    v->jitter_delta_frequency = 6.0f / sample_rate;
    v->jitter_delta_formant_frequency = 6.0f / sample_rate;
    v->jitter_delta_amplitude = 0.2f;
}

/* VoiceStorage::get src/lib.rs:664-671 */
static int voice_get(const orc_voice *v, int32_t phoneme, orc_synthesis_elem *out)
{
    switch (phoneme) {
    case ORC_PH_A: *out = v->phonemes[0]; return 1;
    case ORC_PH_E: *out = v->phonemes[1]; return 1;
    default: return 0; /* Silence | Stop | Glide => None */
    }
}

/* ------------------------------------------------------------------------ */
/* iterator sources                                                          */
/* ------------------------------------------------------------------------ */

/* The source of SequenceElems feeding the Sequencer: either a Selector over
 * PhonemeElems (src/lib.rs:987-1005) or a plain slice of SequenceElems. */
typedef struct {
    const orc_voice *voice;
    const orc_phoneme_elem *ph; /* selector mode when non-NULL */
    const orc_sequence_elem *sq;
    uint32_t n, pos;
} seq_source;

/* Option<SequenceElem> */
typedef struct { int some; orc_sequence_elem e; } opt_seq;

/* Selector::next src/lib.rs:990-1005 */
static opt_seq source_next(seq_source *s)
{
    opt_seq r;
    r.some = 0;
    if (s->pos >= s->n) return r; /* iter.next()? */
    if (s->ph) {
        const orc_phoneme_elem *p = &s->ph[s->pos++];
        orc_synthesis_elem e;
        int has = voice_get(s->voice, p->phoneme, &e);      /* :996 */
        r.some = 1;
        r.e.has_elem = has;
        if (has) r.e.elem = elem_copy_with_frequency(e, p->frequency); /* :1001 */
        else memset(&r.e.elem, 0, sizeof r.e.elem);
        r.e.length = p->length;                               /* :1002 */
        r.e.blend_length = p->blend_length;                   /* :1003 */
    } else {
        r.some = 1;
        r.e = s->sq[s->pos++];
    }
    return r;
}

/* ------------------------------------------------------------------------ */
/* Sequencer: src/lib.rs:839-933                                            */
/* ------------------------------------------------------------------------ */
typedef struct {
    seq_source iter;
    opt_seq cur_elem;   /* :844 */
    opt_seq next_elem;  /* :847 */
    float time;         /* :850 */
    float delta_time;   /* :853 */
} sequencer;

/* IntoSequencer::sequence src/lib.rs:941-949 */
static void sequencer_init(sequencer *s, seq_source src, const orc_voice *voice)
{
    s->iter = src;
    s->delta_time = 1.0f / voice->sample_rate; /* :944 */
    s->cur_elem.some = 0;
    s->next_elem.some = 0;
    s->time = 0.0f;
}

/* Sequencer::next src/lib.rs:859-932; returns 0 for None */
static int sequencer_next(sequencer *s, orc_synthesis_elem *out)
{
    s->time -= s->delta_time; /* :861 */

    if (s->time < 0.0f) { /* :864 */
        if (s->cur_elem.some && s->next_elem.some) {          /* :868 */
            orc_sequence_elem a = s->next_elem.e;
            s->cur_elem = s->next_elem;                       /* :869 */
            s->next_elem = source_next(&s->iter);             /* :870 */
            s->time += a.length;                              /* :873 */
        } else if (!s->cur_elem.some && !s->next_elem.some) { /* :876 */
            s->cur_elem = source_next(&s->iter);              /* :877 */
            s->next_elem = source_next(&s->iter);             /* :878 */
            if (s->cur_elem.some) s->time += s->cur_elem.e.length; /* :881-883 */
        } else {
            return 0;                                         /* :886 */
        }
    }

    /* :891-895 */
    if (!s->cur_elem.some) return 0;                          /* :930 */
    const orc_sequence_elem *a = &s->cur_elem.e;
    int has_b = a->has_elem;
    int has_c = s->next_elem.some && s->next_elem.e.has_elem;

    if (has_b && has_c) {                                     /* :897-903 */
        float alpha = f32_min(s->time / a->blend_length, 1.0f);
        orc_elem_blend(out, &s->next_elem.e.elem, &a->elem, alpha); /* c.blend(b, alpha) */
        return 1;
    }
    if (has_b && !has_c) {                                    /* :906-912 */
        float alpha = f32_min(s->time / a->blend_length, 1.0f);
        orc_synthesis_elem sil = elem_copy_silent(a->elem);
        orc_elem_blend(out, &sil, &a->elem, alpha);           /* b.copy_silent().blend(b, alpha) */
        return 1;
    }
    if (!has_b && has_c) {                                    /* :915-921 */
        float alpha = f32_min(s->time / a->blend_length, 1.0f);
        orc_synthesis_elem c = s->next_elem.e.elem;
        orc_synthesis_elem sil = elem_copy_silent(c);
        orc_elem_blend(out, &c, &sil, alpha);                 /* c.blend(c.copy_silent(), alpha) */
        return 1;
    }
    orc_elem_silent(out);                                     /* :924-927 */
    return 1;
}

/* ------------------------------------------------------------------------ */
/* value noise: src/lib.rs:217-307                                          */
/* ------------------------------------------------------------------------ */
typedef struct { float current, next, phase; uint32_t state; } value_noise;
typedef struct { orc_array current, next; float phase; uint32_t state; } array_value_noise;

/* ValueNoise::new src/lib.rs:227-237 */
static void value_noise_new(value_noise *n, uint32_t *state)
{
    n->current = orc_random_f32(state);
    n->next = orc_random_f32(state);
    n->phase = 0.0f;
    n->state = *state;
}

/* ValueNoise::next src/lib.rs:240-255 */
static float value_noise_next(value_noise *n, float increment)
{
    n->phase += increment;
    if (n->phase > 1.0f) {
        n->phase -= 1.0f;
        n->current = n->next;
        n->next = orc_random_f32(&n->state);
    }
    return n->current * (1.0f - n->phase) + n->next * n->phase;
}

/* ArrayValueNoise::new src/lib.rs:270-286 */
static void array_value_noise_new(array_value_noise *n, uint32_t *state)
{
    for (int i = 0; i < NF; ++i) {
        n->current.v[i] = orc_random_f32(state); /* :276 */
        n->next.v[i] = orc_random_f32(state);    /* :277 */
    }
    n->phase = 0.0f;
    n->state = *state;
}

/* ArrayValueNoise::next src/lib.rs:289-306 */
static void array_value_noise_next(array_value_noise *n, float increment, orc_array *out)
{
    n->phase += increment;
    if (n->phase > 1.0f) {
        n->phase -= 1.0f;
        n->current = n->next;
        for (int i = 0; i < NF; ++i) n->next.v[i] = orc_random_f32(&n->state); /* :301 */
    }
    float a = 1.0f - n->phase;
    for (int i = 0; i < NF; ++i)
        out->v[i] = n->current.v[i] * a + n->next.v[i] * n->phase;             /* :305 */
}

/* ------------------------------------------------------------------------ */
/* Jitter: src/lib.rs:724-798                                               */
/* ------------------------------------------------------------------------ */
typedef struct {
    sequencer iter;
    value_noise freq_noise;
    array_value_noise formant_freq_noise;
    array_value_noise formant_amp_noise;
    float frequency, delta_frequency, delta_formant_freq, delta_amplitude;
} jitter;

/* IntoJitter::jitter src/lib.rs:786-797 */
static void jitter_init(jitter *j, uint32_t seed, const orc_voice *voice)
{
    value_noise_new(&j->freq_noise, &seed);                /* :789 */
    array_value_noise_new(&j->formant_freq_noise, &seed);  /* :790 */
    array_value_noise_new(&j->formant_amp_noise, &seed);   /* :791 */
    j->frequency = voice->jitter_frequency;
    j->delta_frequency = voice->jitter_delta_frequency;
    j->delta_formant_freq = voice->jitter_delta_formant_frequency;
    j->delta_amplitude = voice->jitter_delta_amplitude;
}

/* Jitter::next src/lib.rs:753-777 */
static int jitter_next(jitter *j, orc_synthesis_elem *elem)
{
    if (!sequencer_next(&j->iter, elem)) return 0;                     /* :755 */

    float freq = value_noise_next(&j->freq_noise, j->frequency);       /* :758 */
    orc_array formant_freq, formant_amp;
    array_value_noise_next(&j->formant_freq_noise, j->frequency, &formant_freq); /* :759 */
    array_value_noise_next(&j->formant_amp_noise, j->frequency, &formant_amp);   /* :760 */

    elem->frequency += freq * j->delta_frequency;                      /* :763 */
    float amp_scale = 0.5f * j->delta_amplitude;                       /* :769 */
    for (int i = 0; i < NF; ++i) {
        elem->formant_freq.v[i] += formant_freq.v[i] * j->delta_formant_freq; /* :764 */
        float delta = (formant_amp.v[i] + 1.0f) * amp_scale;           /* :768-769 */
        float mul = 1.0f - delta;                                      /* :772 */
        elem->formant_amp.v[i] = elem->formant_amp.v[i] * mul;         /* :773 */
    }
    return 1;
}

/* ------------------------------------------------------------------------ */
/* Synthesize: src/lib.rs:470-597                                           */
/* ------------------------------------------------------------------------ */
typedef struct {
    jitter iter;
    float phase;
    orc_array filter_state_a, filter_state_b, filter_state_c;
    uint32_t seed;
    double pa[NF], pb[NF], pc[NF]; /* filter states of the double-precision variant (orc_set_precise) */
} synthesize;

/* Tests only: evaluate the per-formant arithmetic of Synthesize::next (src/lib.rs:531-574) in
 * double precision on the SAME f32 parameter track, saw and noise.  The distance between this and
 * the binary32 rendering is the rounding noise of the reference itself — the yardstick the fast
 * (tolerance) mode of the product is measured against.  Never the parity target. */
/* on = 2, 3 (round 4, VERDICT r3 item 4 — "a middle arithmetic tier"): what would a tolerance arithmetic deviate by
 * that rounds the band-pass coefficients g, k, a1, a2 = g a1, a3 = g a2 (src/lib.rs:555-562) EXACTLY as the reference does
 * at every sample and is free everywhere else?  2: everything else in double precision (the best any such tier can do);
 * 3: everything else in binary32 with fused multiply-adds and the factored state update, as a kernel would do it.
 * tools/middle_tier_experiment.py runs both over the random voice tables of profiles/r03_sharpness.txt. */
static int g_precise = 0;
void orc_set_precise(int on) { g_precise = on; }

/* IntoSynthesize::synthesize src/lib.rs:587-596 */
static void synthesize_init(synthesize *s)
{
    s->phase = 0.0f;
    array_splat(&s->filter_state_a, 0.0f);
    array_splat(&s->filter_state_b, 0.0f);
    array_splat(&s->filter_state_c, 0.0f);
    s->seed = 0;
    for (int i = 0; i < NF; ++i) s->pa[i] = s->pb[i] = s->pc[i] = 0.0;
}

/* the arithmetic of Synthesize::next src/lib.rs:501-577 on one elem */
static float synthesize_step(synthesize *s, const orc_synthesis_elem *elem)
{
    /* :503-514 */
    float polyblep;
    if (s->phase < elem->frequency) {
        float t = s->phase / elem->frequency;
        polyblep = 2.0f * t - (t * t) - 1.0f;
    } else if (s->phase > (1.0f - elem->frequency)) {
        float t = (s->phase - 1.0f) / elem->frequency;
        polyblep = (t * t) + 2.0f * t + 1.0f;
    } else {
        polyblep = 0.0f;
    }

    /* :517 */
    float saw = (2.0f * s->phase - 1.0f) - polyblep;

    /* :520-525 */
    s->phase += elem->frequency;
    if (s->phase >= 1.0f) s->phase -= 1.0f;

    /* :528 */
    float noise = orc_random_f32(&s->seed);

    if (g_precise == 2) {
        double sum = 0.0;
        for (int i = 0; i < NF; ++i) {
            double breath = elem->formant_breath.v[i], turb = elem->formant_turb.v[i];
            double nw = (double)saw * (1.0 - breath) + (double)noise * breath;
            double o = 1.0 - (double)elem->formant_smooth.v[i];
            double lp = (o * o) * (o * o) * o;
            s->pa[i] = s->pa[i] + (1.0 - lp) * (nw - s->pa[i]);
            double v0 = s->pa[i] * ((1.0 - turb) + (double)noise * turb) * (double)elem->formant_amp.v[i];
            /* the reference's own binary32 coefficients :555-562 */
            float g = orc_tan_approx(elem->formant_freq.v[i]);
            float k = elem->formant_bw.v[i] / elem->formant_freq.v[i];
            float a1 = 1.0f / (1.0f + g * (g + k));
            float a2 = g * a1;
            float a3 = g * a2;
            double b = s->pb[i], c = s->pc[i];
            double v3 = v0 - c;
            double v1i = (double)a1 * b + (double)a2 * v3;
            double v2 = c + (double)a2 * b + (double)a3 * v3;
            s->pb[i] = 2.0 * v1i - b;
            s->pc[i] = 2.0 * v2 - c;
            sum += v1i;
        }
        return (float)(sum * 0.5);
    }
    if (g_precise == 3) {
        float sum = 0.0f;
        for (int i = 0; i < NF; ++i) {
            float breath = elem->formant_breath.v[i], turb = elem->formant_turb.v[i];
            float nw = fmaf(breath, noise - saw, saw);
            float o = 1.0f - elem->formant_smooth.v[i];
            float o2 = o * o;
            float oml = 1.0f - (o2 * o2) * o;
            float a = s->filter_state_a.v[i];
            a = fmaf(oml, nw - a, a);
            s->filter_state_a.v[i] = a;
            float v0 = a * (elem->formant_amp.v[i] * fmaf(turb, noise - 1.0f, 1.0f));
            float g = orc_tan_approx(elem->formant_freq.v[i]);
            float k = elem->formant_bw.v[i] / elem->formant_freq.v[i];
            float a1 = 1.0f / (1.0f + g * (g + k));
            float a2 = g * a1;
            float a3 = g * a2;
            float b = s->filter_state_b.v[i], c = s->filter_state_c.v[i];
            float v3 = v0 - c;
            float v1i = fmaf(a2, v3, a1 * b);
            float v2 = fmaf(a3, v3, fmaf(a2, b, c));
            s->filter_state_b.v[i] = fmaf(2.0f, v1i, -b);
            s->filter_state_c.v[i] = fmaf(2.0f, v2, -c);
            sum += v1i;
        }
        return sum * 0.5f;
    }
    if (g_precise) {
        double sum = 0.0;
        for (int i = 0; i < NF; ++i) {
            double breath = elem->formant_breath.v[i], turb = elem->formant_turb.v[i];
            double x = elem->formant_freq.v[i], w = elem->formant_bw.v[i];
            double nw = (double)saw * (1.0 - breath) + (double)noise * breath;
            double o = 1.0 - (double)elem->formant_smooth.v[i];
            double lp = (o * o) * (o * o) * o;
            s->pa[i] = s->pa[i] + (1.0 - lp) * (nw - s->pa[i]);
            double v0 = s->pa[i] * ((1.0 - turb) + (double)noise * turb) * (double)elem->formant_amp.v[i];
            double g = ((1.0 - x) * x * (5.0 - 4.0 * (x + 0.5) * (0.5 - x)))
                     / ((x + 0.5) * (5.0 - 4.0 * (1.0 - x) * x) * (0.5 - x));
            double k = w / x;
            double a1 = 1.0 / (1.0 + g * (g + k)), a2 = g * a1, a3 = g * a2;
            double b = s->pb[i], c = s->pc[i];
            double v3 = v0 - c;
            double v1i = a1 * b + a2 * v3;
            double v2 = c + a2 * b + a3 * v3;
            s->pb[i] = 2.0 * v1i - b;
            s->pc[i] = 2.0 * v2 - c;
            sum += v1i;
        }
        return (float)(sum * 0.5);
    }

    orc_array v1;
    for (int i = 0; i < NF; ++i) {
        /* :531 blend_multiple: self*(1-alpha) + other*alpha */
        float breath = elem->formant_breath.v[i];
        float noise_wave = saw * (1.0f - breath) + noise * breath;

        /* :535 */
        float alpha = orc_exp_approx(elem->formant_smooth.v[i]);

        /* :538 */
        s->filter_state_a.v[i] =
            s->filter_state_a.v[i] + (1.0f - alpha) * (noise_wave - s->filter_state_a.v[i]);

        /* :541 */
        float glottal = s->filter_state_a.v[i];

        /* :544-545  splat(1.0).blend_multiple(noise, turb) = 1*(1-turb) + noise*turb */
        float turb = elem->formant_turb.v[i];
        float turbulence = glottal * (1.0f * (1.0f - turb) + noise * turb);

        /* :550 */
        float v0 = turbulence * elem->formant_amp.v[i];

        /* :555 */
        float g = orc_tan_approx(elem->formant_freq.v[i]);
        /* :558 */
        float k = elem->formant_bw.v[i] / elem->formant_freq.v[i];
        /* :560-562 */
        float a1 = 1.0f / (1.0f + g * (g + k));
        float a2 = g * a1;
        float a3 = g * a2;

        /* :565-567 */
        float b = s->filter_state_b.v[i];
        float c = s->filter_state_c.v[i];
        float v3 = v0 - c;
        float v1i = a1 * b + a2 * v3;
        float v2 = c + a2 * b + a3 * v3;

        /* :570-571 */
        s->filter_state_b.v[i] = 2.0f * v1i - b;
        s->filter_state_c.v[i] = 2.0f * v2 - c;
        v1.v[i] = v1i;
    }
    /* :574 */
    return orc_array_sum(&v1) * 0.5f;
}

/* Synthesize::next src/lib.rs:497-578 */
static int synthesize_next(synthesize *s, float *out)
{
    orc_synthesis_elem elem;
    if (!jitter_next(&s->iter, &elem)) return 0; /* :499 */
    *out = synthesize_step(s, &elem);
    return 1;
}

/* ------------------------------------------------------------------------ */
/* chains                                                                    */
/* ------------------------------------------------------------------------ */
static uint64_t run_chain(const orc_voice *voice, seq_source src, uint32_t jitter_seed,
                          float *out, uint64_t cap)
{
    synthesize s;
    sequencer_init(&s.iter.iter, src, voice);
    jitter_init(&s.iter, jitter_seed, voice);
    synthesize_init(&s);
    uint64_t n = 0;
    float x;
    while (synthesize_next(&s, &x)) {
        if (n < cap) out[n] = x;
        ++n;
    }
    return n;
}

uint64_t orc_synthesize_phonemes(const orc_voice *voice, const orc_phoneme_elem *segs,
                                 uint32_t n_segs, uint32_t jitter_seed, float *out,
                                 uint64_t cap)
{
    seq_source src = {voice, segs, NULL, n_segs, 0};
    return run_chain(voice, src, jitter_seed, out, cap);
}

uint64_t orc_synthesize_sequence(const orc_voice *voice, const orc_sequence_elem *segs,
                                 uint32_t n_segs, uint32_t jitter_seed, float *out,
                                 uint64_t cap)
{
    seq_source src = {voice, NULL, segs, n_segs, 0};
    return run_chain(voice, src, jitter_seed, out, cap);
}

static void elem_to_floats(const orc_synthesis_elem *e, float *o)
{
    memcpy(o, e, sizeof *e); /* 49 floats, declared order */
}

uint64_t orc_trace_elems(const orc_voice *voice, const orc_phoneme_elem *segs,
                         uint32_t n_segs, uint32_t jitter_seed, int stage, float *out49,
                         uint64_t cap_samples)
{
    seq_source src = {voice, segs, NULL, n_segs, 0};
    jitter j;
    sequencer_init(&j.iter, src, voice);
    jitter_init(&j, jitter_seed, voice);
    uint64_t n = 0;
    orc_synthesis_elem e;
    for (;;) {
        int ok = stage == 0 ? sequencer_next(&j.iter, &e) : jitter_next(&j, &e);
        if (!ok) break;
        if (n < cap_samples) elem_to_floats(&e, out49 + 49 * n);
        ++n;
    }
    return n;
}

void orc_synthesize_batch(const orc_voice *voices, uint32_t n_voices,
                          const orc_phoneme_elem *segs, const uint32_t *seg_offsets,
                          const uint32_t *voice_ids, const uint32_t *jitter_seeds,
                          uint32_t n_utt, float *out, uint64_t out_stride, uint32_t *out_len)
{
    for (uint32_t u = 0; u < n_utt; ++u) {
        uint32_t vid = voice_ids ? voice_ids[u] : 0;
        if (vid >= n_voices) vid = 0;
        uint32_t s0 = seg_offsets[u], s1 = seg_offsets[u + 1];
        uint64_t n = orc_synthesize_phonemes(&voices[vid], segs + s0, s1 - s0,
                                             jitter_seeds ? jitter_seeds[u] : 0,
                                             out ? out + (uint64_t)u * out_stride : NULL,
                                             out ? out_stride : 0);
        if (out_len) out_len[u] = (uint32_t)n;
    }
}

/* The reference has no threads anywhere in src/; this only exists so the bench can also report
 * "every host core, one utterance per work item" next to the single-thread baseline. */
typedef struct {
    const orc_voice *voices; uint32_t n_voices;
    const orc_phoneme_elem *segs; const uint32_t *seg_offsets, *voice_ids, *jitter_seeds;
    uint32_t n_utt; float *out; uint64_t out_stride; uint32_t *out_len;
    uint32_t next;                       /* work counter, __atomic */
} batch_job;

static void *batch_worker(void *p)
{
    batch_job *j = (batch_job *)p;
    for (;;) {
        uint32_t u = __atomic_fetch_add(&j->next, 1u, __ATOMIC_RELAXED);
        if (u >= j->n_utt) return NULL;
        orc_synthesize_batch(j->voices, j->n_voices, j->segs, j->seg_offsets + u,
                             j->voice_ids ? j->voice_ids + u : NULL,
                             j->jitter_seeds ? j->jitter_seeds + u : NULL, 1,
                             j->out ? j->out + (uint64_t)u * j->out_stride : NULL, j->out_stride,
                             j->out_len ? j->out_len + u : NULL);
    }
}

int orc_synthesize_batch_threads(const orc_voice *voices, uint32_t n_voices,
                                 const orc_phoneme_elem *segs, const uint32_t *seg_offsets,
                                 const uint32_t *voice_ids, const uint32_t *jitter_seeds,
                                 uint32_t n_utt, float *out, uint64_t out_stride,
                                 uint32_t *out_len, uint32_t n_threads)
{
    batch_job j = { voices, n_voices, segs, seg_offsets, voice_ids, jitter_seeds,
                    n_utt, out, out_stride, out_len, 0 };
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 1024) n_threads = 1024;
    pthread_t tid[1024];
    uint32_t started = 0;
    for (; started < n_threads; ++started)
        if (pthread_create(&tid[started], NULL, batch_worker, &j) != 0) break;
    if (started == 0) batch_worker(&j);
    for (uint32_t i = 0; i < started; ++i) pthread_join(tid[i], NULL);
    return (int)started;
}

/* ------------------------------------------------------------------------ */
/* text front half: src/lib.rs:1098-1205, src/languages/mod.rs              */
/* ------------------------------------------------------------------------ */

static uint32_t ascii_lower(uint32_t c) { return (c >= 'A' && c <= 'Z') ? c + 32 : c; }

/* slice::partition_point over rules[lo..hi) for the predicate of :1140-1150.
 * mode 0: chars().nth(index).map_or(true,  |x| x <  character)
 * mode 1: chars().nth(index).map_or(false, |x| x <= character) */
static int rule_pred(const orc_rule *r, uint32_t index, uint32_t character, int mode)
{
    if (index >= r->string_len) return mode == 0;
    uint32_t x = r->string[index];
    return mode == 0 ? (x < character) : (x <= character);
}

static uint32_t partition_point(const orc_rule *rules, uint32_t lo, uint32_t hi,
                                uint32_t index, uint32_t character, int mode)
{
    /* binary search for the first element where pred is false, relative to lo */
    uint32_t left = 0, right = hi - lo;
    while (left < right) {
        uint32_t mid = left + (right - left) / 2;
        if (rule_pred(&rules[lo + mid], index, character, mode)) left = mid + 1;
        else right = mid;
    }
    return left;
}

/* Transcriber::next in a loop: src/lib.rs:1118-1190 */
uint32_t orc_transcribe(const uint32_t *text, uint32_t text_len, const orc_rule *rules,
                        uint32_t n_rules, int case_sensitive, int leading_silence,
                        int32_t *out, uint32_t cap)
{
    static const int32_t SILENCE[1] = {ORC_PH_SILENCE}; /* :1114 */
    const int32_t *buffer = leading_silence ? SILENCE : NULL;
    uint32_t buffer_len = leading_silence ? 1 : 0;
    uint32_t pos = 0; /* Peekable<T> cursor */
    uint32_t n_out = 0;

    for (;;) {
        /* :1120-1122 */
        uint32_t search_min = 0, search_max = n_rules, index = 0;
        int exhausted = 0;
        while (buffer_len == 0) { /* :1125 */
            /* :1127-1133  peek()? */
            if (pos >= text_len) { exhausted = 1; break; }
            uint32_t character = case_sensitive ? text[pos] : ascii_lower(text[pos]);

            /* :1140-1150 */
            uint32_t new_min = partition_point(rules, search_min, search_max, index, character, 0) + search_min;
            uint32_t new_max = partition_point(rules, search_min, search_max, index, character, 1) + search_min;

            if (new_min >= new_max && rules[search_min].string_len == index) { /* :1153 */
                buffer = rules[search_min].phonemes;
                buffer_len = rules[search_min].n_phonemes;
            } else if (new_min >= new_max) { /* :1156 */
                buffer = SILENCE;
                buffer_len = 1;
                pos++; /* :1161 */
            } else {
                search_min = new_min; /* :1164-1166 */
                search_max = new_max;
                index += 1;
                pos++; /* :1169 */
                if (pos >= text_len && rules[search_min].string_len == index) { /* :1172-1175 */
                    buffer = rules[search_min].phonemes;
                    buffer_len = rules[search_min].n_phonemes;
                } else if (pos >= text_len) { /* :1176-1178 */
                    buffer = SILENCE;
                    buffer_len = 1;
                }
            }
        }
        if (exhausted) break; /* the `?` at :1133 returns None */
        /* :1183-1189 */
        if (buffer_len == 0) break; /* buffer.get(0) == None (a rule with no phonemes) */
        int32_t result = buffer[0];
        buffer++;
        buffer_len--;
        if (n_out < cap) out[n_out] = result;
        n_out++;
    }
    return n_out;
}

/* languages::generic(): src/languages/mod.rs:4-34 */
uint32_t orc_language_generic(const orc_rule **rules_out, int *case_sensitive)
{
    static const uint32_t s_a[] = {'a'}, s_e[] = {'e'}, s_i[] = {'i'}, s_ii[] = {'i', 'i'},
                          s_oui[] = {'o', 'u', 'i'}, s_p[] = {'p'};
    static const int32_t p_a[] = {ORC_PH_A}, p_e[] = {ORC_PH_E}, p_i[] = {ORC_PH_A},
                         p_ii[] = {ORC_PH_E, ORC_PH_A}, p_oui[] = {ORC_PH_A, ORC_PH_E, ORC_PH_A},
                         p_p[] = {ORC_PH_SILENCE};
    static const orc_rule rules[] = {
        {s_a, 1, p_a, 1}, {s_e, 1, p_e, 1}, {s_i, 1, p_i, 1},
        {s_ii, 2, p_ii, 2}, {s_oui, 3, p_oui, 3}, {s_p, 1, p_p, 1},
    };
    *rules_out = rules;
    *case_sensitive = 0; /* mod.rs:6 */
    return 6;
}

/* Intonator::next src/lib.rs:1059-1074 */
void orc_intonate(const orc_voice *voice, const int32_t *phonemes, uint32_t n,
                  orc_phoneme_elem *out)
{
    for (uint32_t i = 0; i < n; ++i) {
        out[i].phoneme = phonemes[i];
        out[i].length = 0.5f;                        /* :1070 */
        out[i].blend_length = 0.5f;                  /* :1071 */
        out[i].frequency = voice->center_frequency;  /* :1072 */
    }
}

/* examples/cli.rs:175-184 */
uint64_t orc_say(const orc_voice *voice, const uint32_t *text, uint32_t text_len,
                 uint32_t jitter_seed, float *out, uint64_t cap)
{
    const orc_rule *rules;
    int cs;
    uint32_t n_rules = orc_language_generic(&rules, &cs);
    /* every char yields at most max-rule-phonemes (3) phonemes, plus the leading Silence */
    enum { MAXP = 4096 };
    static int32_t ph[MAXP];
    static orc_phoneme_elem pe[MAXP];
    uint32_t n = orc_transcribe(text, text_len, rules, n_rules, cs, 1, ph, MAXP);
    if (n > MAXP) n = MAXP;
    orc_intonate(voice, ph, n, pe);
    return orc_synthesize_phonemes(voice, pe, n, jitter_seed, out, cap);
}

/* examples/cli.rs:49  ((x * i16::MAX as f32) as i16): Rust float->int `as`
 * truncates toward zero, saturates, and maps NaN to 0. */
int16_t orc_pcm16(float x)
{
    float y = x * 32767.0f;
    if (y != y) return 0;
    if (y >= 32767.0f) return 32767;
    if (y <= -32768.0f) return -32768;
    return (int16_t)y;
}
