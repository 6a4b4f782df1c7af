// pcm16.h — the sample conversion of the reference's WAV sink, examples/cli.rs:49:
// `(x * std::i16::MAX as f32) as i16` (Rust `as`: truncate toward zero, saturate, NaN -> 0).
#pragma once
#include <hip/hip_runtime.h>

namespace grail {

__device__ __forceinline__ int pcm16_from_f32(float x)
{
    // v_cvt_i32_f32 truncates toward zero, saturates and maps NaN to 0 — Rust's `as` for
    // f32 -> integer — and the i16 range is then a clamp of that i32.
    const int v = __float2int_rz(x * 32767.0f);
    return v < -32768 ? -32768 : (v > 32767 ? 32767 : v);
}

}  // namespace grail
