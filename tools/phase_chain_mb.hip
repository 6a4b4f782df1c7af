// phase_chain_mb.hip — the serial carrier-phase chain p_j = fract(p_{j-1} + f_{j-1}) over the 64 lanes of a
// wave as the scan kernel runs it (the value travels down the lanes: v_add_f32_dpp wave_shr:1 + v_fract_f32 +
// the DPP hazard's s_nop): cycles per sample of a lone wave = the floor of the time per utterance.
// (Tried against it: every lane running the same chain with EXEC shifted left once per round and the pitch
// of each round fetched by v_readlane into SGPRs — four instructions per round, 30 cycles instead of 19.)
// hipcc --offload-arch=gfx950 -O3 -o tools/phase_chain_mb.bin tools/phase_chain_mb.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ float chain_dpp(float phase, float f, int lane)
{
    const float f_below = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(f), 0x138, 0xF, 0xF, false));
    const float addend = lane == 0 ? phase : f_below;
    float ph = phase;
#pragma unroll
    for (int r = 0; r < 64; ++r)
        ph = __builtin_amdgcn_fractf(__int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(ph), 0x138, 0xF, 0xF, true)) + addend);
    return ph;
}
// variant 1: the same dependent add + fract chain without the lane shift (not the recurrence: the cost of DPP)
__device__ __forceinline__ float chain_plain(float phase, float f)
{
    float ph = phase;
#pragma unroll
    for (int r = 0; r < 64; ++r) ph = __builtin_amdgcn_fractf(ph + f);
    return ph;
}
// variant 2: the shift inside 16-lane rows only (row_shr:1; not the recurrence either)
__device__ __forceinline__ float chain_row(float phase, float f, int lane)
{
    float ph = phase;
#pragma unroll
    for (int r = 0; r < 64; ++r)
        ph = __builtin_amdgcn_fractf(__int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(ph), 0x111, 0xF, 0xF, true)) + f);
    return ph;
}

template <int V>
__global__ void bench(const float *fin, float *out, long long *cycles, int tiles)
{
    const int lane = threadIdx.x;
    float phase = 0.0f, acc = 0.0f;
    const long long t0 = clock64();
    for (int t = 0; t < tiles; ++t) {
        const float f = fin[(t & 63) * 64 + lane];
        const float ph = V == 0 ? chain_dpp(phase, f, lane) : V == 1 ? chain_plain(phase, f) : chain_row(phase, f, lane);
        const float ph_l = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ph), 63));
        const float f_l = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(f), 63));
        phase = __builtin_amdgcn_fractf(ph_l + f_l);
        acc += ph;
        if (t < 64) out[t * 64 + lane] = ph;
    }
    const long long t1 = clock64();
    out[64 * 64 + lane] = acc;
    if (lane == 0) cycles[blockIdx.x] = t1 - t0;
}

int main(int argc, char **argv)
{
    const int tiles = 4096;
    const int blocks = argc > 1 ? atoi(argv[1]) : 1;   // more than one: the same chain on many CUs at once (clock under load)
    std::vector<float> f(64 * 64);
    uint32_t s = 12345u;
    for (auto &x : f) { s = s * 1664525u + 1013904223u; x = 0.002f + 0.004f * (float)(s >> 8) / 16777216.0f; }
    float *d_f, *d_o; long long *d_c;
    hipMalloc(&d_f, f.size() * 4); hipMemcpy(d_f, f.data(), f.size() * 4, hipMemcpyHostToDevice);
    hipMalloc(&d_c, 8 * (size_t)blocks);
    hipMalloc(&d_o, 65 * 64 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int v = 2; v >= 0; --v) {
        float ms = 0.0f;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0, 0);
            if (v == 0) hipLaunchKernelGGL(bench<0>, dim3(blocks), dim3(64), 0, 0, d_f, d_o, d_c, tiles);
            else if (v == 1) hipLaunchKernelGGL(bench<1>, dim3(blocks), dim3(64), 0, 0, d_f, d_o, d_c, tiles);
            else hipLaunchKernelGGL(bench<2>, dim3(blocks), dim3(64), 0, 0, d_f, d_o, d_c, tiles);
            hipEventRecord(e1, 0);
            hipDeviceSynchronize();
            hipEventElapsedTime(&ms, e0, e1);
        }
        long long cv; hipMemcpy(&cv, d_c, 8, hipMemcpyDeviceToHost);
        printf("%s: %.2f ns per sample (%.2f s_memtime ticks)\n", v == 0 ? "wave_shr:1 + fract (the recurrence)" : v == 1 ? "add + fract without a lane shift" : "row_shr:1 + fract",
               ms * 1e6 / ((double)tiles * 64), (double)cv / tiles / 64);
    }
    long long c; hipMemcpy(&c, d_c, 8, hipMemcpyDeviceToHost);
    std::vector<float> o(65 * 64); hipMemcpy(o.data(), d_o, 65 * 64 * 4, hipMemcpyDeviceToHost);
    // the same chain on the host
    int bad = 0;
    float phase = 0.0f;
    for (int t = 0; t < 64; ++t) {
        float p = phase;
        for (int j = 0; j < 64; ++j) {
            if (o[t * 64 + j] != p) ++bad;
            float q = p + f[t * 64 + j];
            p = q >= 1.0f ? q - 1.0f : q;
        }
        phase = p;
    }
    printf("the recurrence: %d of 4096 values differ from the serial loop\n", bad);
    return bad != 0;
}
