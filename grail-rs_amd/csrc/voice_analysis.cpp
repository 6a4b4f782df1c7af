// voice_analysis.cpp — what the host derives from a voice table before any launch: which kernel families its voices
// qualify for (four-formant kernels, the scan kernel's window, the warm-up length of the time-split kernels), the
// predicted deviation of fast arithmetic (sharpness) and with it the tier a batch is served in, and the chunk grid
// of a time-split launch.
#include "api_internal.hpp"

using namespace grail;
using namespace grail::host;

namespace grail {
namespace host {

// Can formants 5-8 of this voice be left out of a one-shot render altogether?  They must contribute
// exactly +0.0 to every sample of the reference's own arithmetic, whatever the segments are (given
// alpha in [0,1], i.e. no segment shorter than two samples — checked per batch):
//   amplitude exactly +0 in every phoneme, 0 <= jitter_delta_amplitude/2 <= 1/4  => v0 = tw * (+0) = +-0
//   breath, turbulence, smoothness in [0,1]                                      => the low-pass state and tw stay finite
//   frequency and bandwidth inside pair_is_safe's window with the jitter margin  => finite g, k and 0 < a1, a2, a3 < inf
// and then w1 = a1*(+0) + a2*(+-0) = +0 and the band-pass state never leaves +0 ("Silent
// formants").  SynthesisElem::silent() (0.25 / 0.25 / 0.25 / 0 / 0 / 0) satisfies all of it.
bool live4_ok(const grail_voice &v)
{
    const float amp_scale = 0.5f * v.jitter_delta_amplitude;
    const float jm = 1.002f * std::fabs(v.jitter_delta_formant_frequency);
    const bool ok = (amp_scale >= 0.0f) && (amp_scale <= 0.25f) && (jm <= 1.0f) &&
                    (v.jitter_frequency >= 0.0f) && (v.jitter_frequency <= 1.0f) &&
                    (v.sample_rate > 0.0f) && std::isfinite(v.sample_rate) &&
                    std::isfinite(v.jitter_delta_frequency);
    return ok && live4_elems_ok(v.phonemes, NUM_VOICED, v.jitter_delta_formant_frequency);
}

// ... the part of it that concerns the elems (a voice's phonemes, or the caller-built elems of a batch with the largest
// |jitter_delta_formant_frequency| of the voices it names)
bool live4_elems_ok(const grail_synthesis_elem *elems, size_t n_elems, float jitter_delta_formant_frequency)
{
    constexpr float X_LO = 9.5367431640625e-07f, X_HI = 0.5f - 9.5367431640625e-07f;
    constexpr float W_LO = 1.8189894035458565e-12f, W_HI = 512.0f;
    const float jm = 1.002f * std::fabs(jitter_delta_formant_frequency);
    bool ok = jm <= 1.0f;
    for (size_t p = 0; p < n_elems && ok; ++p) {
        const grail_synthesis_elem &e = elems[p];
        for (int i = NF / 2; i < NF && ok; ++i) {
            uint32_t bits;
            std::memcpy(&bits, &e.formant_amp[i], sizeof bits);
            const float f = e.formant_freq[i], w = e.formant_bw[i];
            ok = bits == 0u && e.formant_breath[i] >= 0.0f && e.formant_breath[i] <= 1.0f &&
                 e.formant_turb[i] >= 0.0f && e.formant_turb[i] <= 1.0f &&
                 e.formant_smooth[i] >= 0.0f && e.formant_smooth[i] <= 1.0f &&
                 (f * 0.999f - jm >= X_LO) && (f * 1.001f + jm <= X_HI) && (w >= W_LO) && (w <= W_HI);
        }
    }
    return ok;
}

// Can this voice go through the time-parallel scan kernel (fast arithmetic, small batches)?  That path
// has no IEEE-division fallback: every formant of every phoneme must sit inside pair_is_safe's window
// with the jitter margin, and the jitter parameters must be sane.  SynthesisElem::silent() qualifies.
bool scan_voice_ok(const grail_voice &v)
{
    const float amp_scale = 0.5f * v.jitter_delta_amplitude;
    const float jm = 1.002f * std::fabs(v.jitter_delta_formant_frequency);
    bool ok = std::isfinite(amp_scale) && (jm <= 1.0f) && (v.jitter_frequency >= 0.0f) &&
              (v.jitter_frequency <= 0.25f) && (v.sample_rate > 0.0f) && std::isfinite(v.sample_rate) &&
              std::isfinite(v.jitter_delta_frequency);
    return ok && scan_elems_ok(v.phonemes, NUM_VOICED, v.jitter_delta_formant_frequency);
}

// ... the part of it that concerns the elems (a voice's phonemes, or the caller-built elems of a batch with the
// largest |jitter_delta_formant_frequency| of the voices it names)
bool scan_elems_ok(const grail_synthesis_elem *elems, size_t n_elems, float jitter_delta_formant_frequency)
{
    constexpr float X_LO = 9.5367431640625e-07f, X_HI = 0.5f - 9.5367431640625e-07f;
    constexpr float W_LO = 1.8189894035458565e-12f, W_HI = 512.0f;
    const float jm = 1.002f * std::fabs(jitter_delta_formant_frequency);
    bool ok = jm <= 1.0f;
    for (size_t p = 0; p < n_elems && ok; ++p) {
        const grail_synthesis_elem &e = elems[p];
        for (int i = 0; i < NF && ok; ++i) {
            const float f = e.formant_freq[i], w = e.formant_bw[i];
            ok = std::isfinite(e.formant_amp[i]) && std::isfinite(e.formant_breath[i]) &&
                 std::isfinite(e.formant_turb[i]) && e.formant_smooth[i] >= 0.0f && e.formant_smooth[i] <= 1.0f &&
                 (f * 0.999f - jm >= X_LO) && (f * 1.001f + jm <= X_HI) && (w >= W_LO) && (w <= W_HI);
        }
    }
    return ok;
}

// Time-split fast kernels: how many samples until a filter state that started from zero is within 2^-21 of the
// state the reference would have (relative to the state's size, which is below full scale)?  The chain of
// Synthesize::next per formant is a one-pole low-pass with factor exp_approx(smooth) = (1 - smooth)^5 (:535-538)
// and the trapezoidal state-variable band-pass (:555-571), whose poles are the bilinear images
// z = (1 + s) / (1 - s) of s = g (-k/2 +- sqrt(k^2/4 - 1)), g = tan_approx(freq), k = bw / freq; for k < 2,
// |z|^2 = (1 - g k + g^2) / (1 + g k + g^2) ~ exp(-2 pi bw).  The slowest of them over every phoneme (blends
// move the parameters between phonemes and towards silent()'s 0.25 / 0.25 / 0.25, which decays at once) over the range
// the formant-frequency jitter moves the band-pass through, with a 5 % margin, gives the length; where the low-pass and
// the band-pass decay at nearly the same rate the cascade's n rho^n is solved for instead of rho^n.  Formants that are silent in
// every phoneme have nothing to converge.  0: the voice does not qualify (a parameter outside the window, or a
// warm-up longer than 16384 samples).
// (elems_warmup: the same over any set of elems — a voice's phonemes, or the caller-built elems of a batch, between
// consecutive ones of which the parameters blend — with jd = the largest |jitter_delta_formant_frequency| that applies)
uint32_t elems_warmup(const grail_synthesis_elem *elems, size_t n_elems, double jd)
{
    // per-sample decay rate of the band-pass envelope at formant frequency f, bandwidth w (0: not a decaying filter)
    auto svf_rate = [](double f, double w) -> double {
        if (!(f > 0.0 && f < 0.5)) return 0.0;
        const double g = ((1 - f) * f * (5 - 4 * (f + 0.5) * (0.5 - f))) / ((f + 0.5) * (5 - 4 * (1 - f) * f) * (0.5 - f));
        const double k = w / f;
        double z;
        if (k < 2.0) {
            z = std::sqrt((1 - g * k + g * g) / (1 + g * k + g * g));
        } else {
            const double root = std::sqrt(k * k / 4 - 1);
            const double s1 = g * (-k / 2 + root), s2 = g * (-k / 2 - root);
            z = std::fmax(std::fabs((1 + s1) / (1 - s1)), std::fabs((1 + s2) / (1 - s2)));
        }
        return (z > 0.0 && z < 1.0) ? -std::log(z) : 0.0;
    };
    const double eps = 1.0 / 2097152.0;                     // 2^-21
    jd = std::fabs(jd);
    if (!std::isfinite(jd)) return 0;
    double longest = 0.0;                                   // samples
    bool any = false;
    for (int i = 0; i < NF; ++i) {
        bool audible = false;
        for (size_t p = 0; p < n_elems; ++p) audible = audible || !(elems[p].formant_amp[i] == 0.0f);
        if (!audible) continue;
        any = true;
        for (size_t p = 0; p < n_elems; ++p) {
            const grail_synthesis_elem &e = elems[p];
            const double f = e.formant_freq[i], w = e.formant_bw[i], sm = e.formant_smooth[i];
            if (!(f > 0.0 && f < 0.5 && w > 0.0 && sm > 0.0 && sm < 1.0) || !std::isfinite(w)) return 0;
            // the formant-frequency jitter moves the band-pass by up to +-jitter_delta_formant_frequency (Jitter::next
            // :764 adds noise in [-1, 1] times it): the slowest decay over that range
            double l_bp = svf_rate(f, w);
            for (const double ff : {f - jd, f + jd})
                if (ff > 0.0 && ff < 0.5) l_bp = std::fmin(l_bp, svf_rate(ff, w));
            const double l_lp = -5.0 * std::log1p(-sm);                          // (1 - smooth)^5 per sample
            if (!(l_bp > 0.0) || !(l_lp > 0.0)) return 0;
            const double slow = std::fmin(l_bp, l_lp), gap = std::fabs(l_bp - l_lp);
            // The low-pass feeds the band-pass: what is left of a wrong start after n samples is bounded by
            // rho^n + sum_j rho_bp^(n-1-j) rho_lp^j, i.e. by (1 + m) rho^n with m = min(n, 1 / |rate difference|).  Far
            // apart (every shipped voice: 0.17 against 0.004 per sample) m is a few samples' worth and the 5 % margin
            // covers it; when the two rates are within a fifth of each other the residual decays like n rho^n and the
            // length is solved for that.
            double n = std::log(1.0 / eps) / (0.95 * slow);
            if (gap <= 0.2 * std::fmax(l_bp, l_lp))
                for (int it = 0; it < 4; ++it) n = std::log((1.0 + std::fmin(n, 1.0 / std::fmax(gap, 1e-12))) / eps) / (0.95 * slow);
            longest = std::fmax(longest, n);
        }
    }
    if (!any) return 64;                                    // nothing audible: any state is the right one
    if (!(longest <= 16384.0)) return 0;
    return ((uint32_t)std::ceil(longest) + 63u) / 64u * 64u;
}

uint32_t voice_warmup(const grail_voice &v)
{
    return elems_warmup(v.phonemes, NUM_VOICED, (double)v.jitter_delta_formant_frequency);
}

// Fast arithmetic and sharp resonances.  The fast kernels interpolate the filter coefficients of Synthesize::next
// (:555-562) between points evaluated with fused and reordered operations; the reference rounds every operation
// anew at every sample.  A rounding-level difference of a coefficient that lasts for a sub-tile moves the
// resonance of a band-pass by that much of its centre frequency and its damping by that much of one, i.e. the
// output by (difference) x Q resp. x (ring time) of the formant's amplitude — the same amplification the
// reference's own rounding gets (its binary32 rendering is about a third as far from its formulas in double
// precision).  Measured (tools/q_sweep.py: bandwidth sweeps of the shipped voices; a frequency x bandwidth grid of
// single formants; tools/sharpness_data.py: 1 400 random tables; profiles/r03_sharpness.txt): the deviation of a
// single formant that carries all of the amplitude is ~ 2 500 / bandwidth [Hz at 48 kHz] * 2^-23 up to 2.4 kHz
// and grows with the square of the frequency above that; it is proportional to the formant's share of the
// amplitudes, and the formants add up in quadrature.  Hence
//     E_i = share_i * (0.0709 / bw_i) * (1 + (f_i / 0.075)^2)     (f, bw in cycles per sample, as in the elems;
//     S   = sqrt(sum_i E_i^2)                                       share, bw, f: the worst of the phonemes)
// which, scaled as it is, lies above 99.5 % of the random tables' measured deviations and within a factor 1.45
// below the rest; voices::generic() has S = 24 (measured 13 - 20), the bench presets 20 - 22 (11 - 20).
// GRAIL_FAST_TOLERANCE = 64 * 2^-23 is therefore a promise the fast kernels can keep only up to a sharpness: the
// host serves fast arithmetic for S <= GRAIL_FAST_SHARPNESS_LIMIT = 28 (worst measured among those, 3 000 random tables:
// 24; at 32 one table in 3 000 reached 56) and
// renders sharper tables with the exact kernels (their bits satisfy the tolerance trivially).
// Returns S in units of 2^-23 of max(1, peak); +inf for parameters outside the window of the formulas.
double elems_sharpness(const grail_synthesis_elem *elems, size_t n)
{
    double share[NF] = {0}, sens[NF] = {0};
    for (size_t p = 0; p < n; ++p) {
        double total = 0.0;
        for (int i = 0; i < NF; ++i) total += std::fabs((double)elems[p].formant_amp[i]);
        if (!std::isfinite(total)) return INFINITY;
        for (int i = 0; i < NF; ++i)
            if (total > 0.0) share[i] = std::fmax(share[i], std::fabs((double)elems[p].formant_amp[i]) / total);
    }
    double sum = 0.0;
    for (int i = 0; i < NF; ++i) {
        if (share[i] == 0.0) continue;             // never audible: nothing rings
        for (size_t p = 0; p < n; ++p) {
            const double f = elems[p].formant_freq[i], w = elems[p].formant_bw[i];
            if (!(f > 0.0 && f < 0.5 && w > 0.0) || !std::isfinite(w)) return INFINITY;
            sens[i] = std::fmax(sens[i], (0.0709 / w) * (1.0 + (f / 0.075) * (f / 0.075)));
        }
        sum += (share[i] * sens[i]) * (share[i] * sens[i]);
    }
    return std::sqrt(sum);
}
// Is fast arithmetic served for this batch?  Caller-built elems are judged themselves; a phoneme batch by the sharpest of
// the voices IT USES (one sharp voice in the table does not take fast arithmetic away from batches that never name it);
// without a batch: by the whole table.
double batch_sharpness(const grail_ctx *ctx, const grail_batch *batch)
{
    if (!batch) return ctx->voices_sharpness;
    if (!batch->phoneme_mode) return batch->elems_sharpness;
    double s = 0.0;
    for (const uint32_t v : batch->used_voices) s = std::fmax(s, v < ctx->voice_sharpness.size() ? ctx->voice_sharpness[v] : INFINITY);
    return s;
}
// Which arithmetic a batch is rendered in when "arithmetic" asks for a tolerance mode: 1 = the interpolating tier (up
// to "fast_sharpness_limit"), 2 = the reference's own band-pass coefficients at every sample (MID; sharper voices, up to
// "fast_sharpness_limit_exact_coefficients"), 0 = the exact kernels (sharper still, or the tier switched off).
int fast_tier_for(const grail_ctx *ctx, const grail_batch *batch, int arithmetic)
{
    if (!arithmetic) return 0;
    const double s = batch_sharpness(ctx, batch);
    if (arithmetic == 1 && s <= (double)ctx->fast_limit) return 1;
    if ((ctx->mid_option || arithmetic == 2) && s <= (double)ctx->mid_limit) return 2;
    return 0;
}
int fast_tier(const grail_ctx *ctx, const grail_batch *batch) { return fast_tier_for(ctx, batch, ctx->fast_option); }
// The chunk grid of a time-split launch: K chunks over `span` samples.  Chunk k's lane fast-forwards the chain over
// b[k] - W samples (cost r per sample, in units of a rendered sample), warms up over W and renders b[k+1] - b[k]:
// the bounds are spaced so that all lanes take the same time (T below, by bisection).  Bounds are multiples
// of 64; the last one is left to the caller (the row capacity).  false: K chunks do not fit.
bool split_grid(uint32_t span, uint32_t warmup, int K, double r, uint32_t *b)
{
    auto lay = [&](double T, double *out) {
        double at = 0.0;
        for (int k = 0; k < K; ++k) {
            out[k] = at;
            const double before = k ? r * std::fmax(at - warmup, 0.0) + std::fmin((double)warmup, at) : 0.0;
            const double len = T - before;
            if (len < 64.0) return -1.0;
            at += len;
        }
        return at;
    };
    double lo = 0.0, hi = (double)span + warmup + 64.0, pos[SPLIT_MAX_CHUNKS + 1];
    if (lay(hi, pos) < (double)span) return false;
    for (int it = 0; it < 60; ++it) {
        const double mid = 0.5 * (lo + hi);
        const double end = lay(mid, pos);
        if (end < 0.0 || end < (double)span) lo = mid;
        else hi = mid;
    }
    if (lay(hi, pos) < 0.0) return false;
    b[0] = 0;
    for (int k = 1; k < K; ++k) {
        b[k] = ((uint32_t)pos[k] + 32u) / 64u * 64u;
        if (b[k] <= b[k - 1]) return false;
    }
    return b[K - 1] < span;
}

}  // namespace host
}  // namespace grail

extern "C" {

float grail_fast_sharpness(const grail_voice *voice)
{
    return voice ? (float)elems_sharpness(voice->phonemes, NUM_VOICED) : INFINITY;
}

uint32_t grail_time_split_warmup(const grail_voice *voice) { return voice ? voice_warmup(*voice) : 0u; }

int grail_time_split_grid(uint32_t span_samples, uint32_t warmup, uint32_t chunks, uint32_t ff_cost_permille,
                          uint32_t *bounds)
{
    if (!bounds) return fail(GRAIL_ERR_INVALID_ARG, "bounds is NULL");
    if (chunks < 2u || chunks > (uint32_t)SPLIT_MAX_CHUNKS)
        return fail(GRAIL_ERR_INVALID_ARG, "chunks must be 2..64");
    if (ff_cost_permille > 1000u) return fail(GRAIL_ERR_INVALID_ARG, "ff_cost_permille must be 0..1000");
    uint32_t b[SPLIT_MAX_CHUNKS + 1];
    if (!split_grid(span_samples, warmup, (int)chunks, 1e-3 * (double)ff_cost_permille, b))
        return fail(GRAIL_ERR_INVALID_ARG, "so many chunks do not fit the span");
    std::memcpy(bounds, b, sizeof(uint32_t) * chunks);
    return GRAIL_OK;
}

}  // extern "C"
