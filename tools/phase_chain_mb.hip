// phase_chain_mb.hip — the serial carrier-phase chain p_j = fract(p_{j-1} + f_{j-1}) over the 64 lanes of a
// wave as the scan kernel runs it (the value travels down the lanes: v_add_f32_dpp wave_shr:1 + v_fract_f32 +
// the DPP hazard's s_nop): cycles per sample of a lone wave = the floor of the time per utterance.
// (Tried against it: every lane running the same chain with EXEC shifted left once per round and the pitch
// of each round fetched by v_readlane into SGPRs — four instructions per round, 30 cycles instead of 19.)
// hipcc --offload-arch=gfx950 -O3 -o tools/phase_chain_mb.bin tools/phase_chain_mb.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
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
template <int V>
__global__ void bench(const float *fin, float *out, long long *cycles, int tiles)
{
    const int lane = threadIdx.x;
    float phase = 0.0f, acc = 0.0f;
    const long long t0 = clock64();
    for (int t = 0; t < tiles; ++t) {
        const float f = fin[(t & 63) * 64 + lane];
        const float ph = chain_dpp(phase, f, lane);
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

int main()
{
    const int tiles = 4096;
    std::vector<float> f(64 * 64);
    uint32_t s = 12345u;
    for (auto &x : f) { s = s * 1664525u + 1013904223u; x = 0.002f + 0.004f * (float)(s >> 8) / 16777216.0f; }
    float *d_f, *d_o; long long *d_c;
    hipMalloc(&d_f, f.size() * 4); hipMemcpy(d_f, f.data(), f.size() * 4, hipMemcpyHostToDevice);
    hipMalloc(&d_c, 8);
    hipMalloc(&d_o, 65 * 64 * 4);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(bench<0>, dim3(1), dim3(64), 0, 0, d_f, d_o, d_c, tiles);
        hipDeviceSynchronize();
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
    printf("carrier-phase chain: %.1f s_memtime ticks per tile of 64 samples, %.2f per sample; %d of 4096 values differ from the serial loop\n",
           (double)c / tiles, (double)c / tiles / 64, bad);
    return bad != 0;
}
