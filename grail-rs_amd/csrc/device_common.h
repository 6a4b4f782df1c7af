// device_common.h — small device-side helpers shared by the kernels (internal).
// Reference: random_f32 src/lib.rs:36-55, tan_approx :63-70, exp_approx :75-82, Selector::next :990-1005.
#pragma once

#include "kernels.h"

namespace grail {
namespace {

typedef float f2 __attribute__((ext_vector_type(2)));

// ---- scalar / packed helpers: V is float (1 formant) or f2 (2 formants) --------
template <int W> struct VecOf;
template <> struct VecOf<1> { typedef float type; };
template <> struct VecOf<2> { typedef f2 type; };

__device__ __forceinline__ float vfma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ f2 vfma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ float vrcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ f2 vrcp(f2 x)
{
    f2 r;
    r.x = __builtin_amdgcn_rcpf(x.x);
    r.y = __builtin_amdgcn_rcpf(x.y);
    return r;
}
__device__ __forceinline__ float vsplat(float x, float) { return x; }
__device__ __forceinline__ f2 vsplat(float x, f2) { f2 r; r.x = x; r.y = x; return r; }
__device__ __forceinline__ float vget(float v, int) { return v; }
__device__ __forceinline__ float vget(f2 v, int c) { return c == 0 ? v.x : v.y; }
__device__ __forceinline__ void vset(float &v, int, float x) { v = x; }
__device__ __forceinline__ void vset(f2 &v, int c, float x) { if (c == 0) v.x = x; else v.y = x; }

// s -> s*16807 + 1 applied n times is s*mul[n] + add[n] (mod 2^32), n = 0..64
struct LcgSkip {
    uint32_t mul[65], add[65];
    constexpr LcgSkip() : mul(), add()
    {
        mul[0] = 1u;
        add[0] = 0u;
        for (int n = 1; n <= 64; ++n) {
            mul[n] = mul[n - 1] * 16807u;
            add[n] = add[n - 1] * 16807u + 1u;
        }
    }
};
__device__ const LcgSkip LCG_SKIP{};

// random_f32, src/lib.rs:36-55
__device__ __forceinline__ float lcg_f32(uint32_t &s)
{
    s = s * 16807u + 1u;
    return (__uint_as_float((s >> 9) | 0x3F800000u) - 1.5f) * 2.0f;
}

// Correctly rounded f32 division without the v_div_scale / v_div_fmas / v_div_fixup
// wrapper.  SAFE = true requires both operands finite, normal and within
// [2^-60, 2^60] in magnitude (then no intermediate can over- or underflow and the
// rounding depends on the significands only).  For that window the sequence is
// PROVEN equal to IEEE division by exhaustion over all 2^46 significand pairs with
// an exact integer remainder check (tools/div_exhaustive.hip,
// profiles/r01_div_exhaustive.txt), so the result is bit-identical to `a / b`.
template <bool SAFE, typename V>
__device__ __forceinline__ V div_exact(V a, V b)
{
    if constexpr (SAFE) {
        const V one = vsplat(1.0f, a);
        V y = vrcp(b);
        const V e = vfma(-b, y, one);
        y = vfma(e, y, y);               // RN(1/b)
        const V q = a * y;
        const V r = vfma(-b, q, a);      // exact remainder
        return vfma(r, y, q);
    } else {
        return a / b;                    // hipcc's IEEE sequence, per component
    }
}

// 1/x: v_rcp_f32 + one Newton step equals the correctly rounded reciprocal for every
// float with |x| in [2^-60, 2^61) (exhaustive, tools/div_check.hip,
// profiles/r01_div_check.txt).
template <bool SAFE, typename V>
__device__ __forceinline__ V rcp_exact(V x)
{
    const V one = vsplat(1.0f, x);
    if constexpr (SAFE) {
        const V y = vrcp(x);
        const V e = vfma(-x, y, one);
        return vfma(e, y, y);
    } else {
        return one / x;
    }
}

// tan_approx, src/lib.rs:63-70 (tan(pi x), Bhaskara-style rational)
template <bool SAFE, typename V>
__device__ __forceinline__ V tan_approx(V x)
{
    const V omx = 1.0f - x;
    const V xph = x + 0.5f;
    const V hmx = 0.5f - x;
    const V num = (omx * x) * (5.0f - (4.0f * xph) * hmx);
    const V den = (xph * (5.0f - (4.0f * omx) * x)) * hmx;
    return div_exact<SAFE>(num, den);
}

// exp_approx, src/lib.rs:75-82 ((1-x)^5)
template <typename V>
__device__ __forceinline__ V exp_approx(V x)
{
    const V o = 1.0f - x;
    const V o2 = o * o;
    return (o2 * o2) * o;
}

// Option<SequenceElem> held in registers
struct Seg {
    bool some;
    int elem;  // table row, or -1 for None
    float length, blend_length, frequency;
};

// iter.next() of the Sequencer's source.  Phoneme mode folds in Selector::next
// (src/lib.rs:990-1005): VoiceStorage::get (:664-671) and
// copy_with_frequency (:445-450: frequency.min(0.5)).
// Live streams keep an utterance's segments in a ring: segment `pos` sits at ring_base + (pos & ring_mask); every
// other launch passes ring_base = 0, ring_mask = ~0 (pos indexes segs directly).
__device__ __forceinline__ void fetch_seg(Seg &s, const DevSeg *__restrict__ segs,
                                          uint32_t &pos, uint32_t end, bool phoneme_mode,
                                          uint32_t elem_base, uint32_t ring_base = 0u, uint32_t ring_mask = 0xFFFFFFFFu)
{
    if (pos < end) {
        const DevSeg d = segs[ring_base + (pos & ring_mask)];
        ++pos;
        s.some = true;
        s.length = d.length;
        s.blend_length = d.blend_length;
        if (phoneme_mode) {
            const int ph = d.elem;
            const bool voiced = ph >= PH_FIRST_VOICED && ph < PH_FIRST_VOICED + NUM_VOICED;
            s.elem = voiced ? (int)elem_base + (ph - PH_FIRST_VOICED) : -1;
            s.frequency = __builtin_fminf(d.frequency, 0.5f);
        } else {
            s.elem = d.elem;
            s.frequency = d.frequency;
        }
    } else {
        s.some = false;
        s.elem = -1;
        s.length = 0.0f;
        s.blend_length = 1.0f;
        s.frequency = 0.0f;
    }
}

}  // namespace
}  // namespace grail
