// valu_microbench.hip — per-instruction VALU issue cost on gfx950, as a function of
// waves per SIMD.  Each kernel runs N iterations of an unrolled block of UNROLL
// independent-ish instructions of one kind; reports cycles per wave-instruction per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_microbench.hip -o /tmp/valu_mb && /tmp/valu_mb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int ITERS = 20000;

// 8 independent chains so dependent-issue latency is not the limit
#define REP8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)

template <int KIND>
__global__ void kern(float *out, float seedf, unsigned long long *cycles)
{
    float a[8], b[8];
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p[8], q[8];
    for (int i = 0; i < 8; ++i) {
        a[i] = seedf + threadIdx.x * 0.001f + i;
        b[i] = 1.0f + i * 0.125f + seedf;
        p[i] = f2{a[i], b[i]};
        q[i] = f2{b[i], a[i]};
    }
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; ++it) {
        if constexpr (KIND == 0) {
#define OP(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            REP8(OP) REP8(OP)
#undef OP
        } else if constexpr (KIND == 1) {
#define OP(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b[i]));
            REP8(OP) REP8(OP)
#undef OP
        } else if constexpr (KIND == 2) {
#define OP(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(q[i]));
            REP8(OP) REP8(OP)
#undef OP
        } else if constexpr (KIND == 3) {
#define OP(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(q[i]));
            REP8(OP) REP8(OP)
#undef OP
        } else if constexpr (KIND == 4) {
#define OP(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
            REP8(OP) REP8(OP)
#undef OP
        } else if constexpr (KIND == 5) {
#define OP(i) asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0" : "+v"(a[i]) : "v"(b[i]) : "vcc");
            REP8(OP) REP8(OP)
#undef OP
        } else if constexpr (KIND == 6) {
#define OP(i) asm volatile("v_div_fmas_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b[i]) : "vcc");
            REP8(OP) REP8(OP)
#undef OP
        } else if constexpr (KIND == 7) {
#define OP(i) asm volatile("v_div_fixup_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b[i]));
            REP8(OP) REP8(OP)
#undef OP
        } else if constexpr (KIND == 8) {
#define OP(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            REP8(OP) REP8(OP)
#undef OP
        } else if constexpr (KIND == 9) {
#define OP(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b[i]));
            REP8(OP) REP8(OP)
#undef OP
        } else if constexpr (KIND == 10) {
#define OP(i) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a[i]) : "v"(b[i]));
            REP8(OP) REP8(OP)
#undef OP
        } else if constexpr (KIND == 11) {
#define OP(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(q[i]));
            REP8(OP) REP8(OP)
#undef OP
        } else if constexpr (KIND == 12) {
            // plain mul and add alternating (the unfused a*b+c pattern)
#define OP(i) asm volatile("v_mul_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            REP8(OP)
#undef OP
        } else if constexpr (KIND == 13) {
#define OP(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            REP8(OP) REP8(OP)
#undef OP
        } else if constexpr (KIND == 14) {
#define OP(i) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            REP8(OP) REP8(OP)
#undef OP
        } else if constexpr (KIND == 15) {
#define OP(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(a[i]), "v"(b[i]) : "vcc");
            REP8(OP) REP8(OP)
#undef OP
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int KIND>
void run(const char *name)
{
    // blocks of 64 threads (one wave); waves_per_simd w -> 256 CUs * 4 SIMDs * w blocks
    float *out; unsigned long long *cyc;
    const int maxblocks = 256 * 4 * 8;
    CHECK(hipMalloc(&out, maxblocks * 64 * sizeof(float)));
    CHECK(hipMalloc(&cyc, maxblocks * sizeof(unsigned long long)));
    printf("%-16s", name);
    for (int w : {1, 2, 4, 8}) {
        int blocks = 256 * 4 * w;
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(kern<KIND>, dim3(blocks), dim3(64), 0, 0, out, 0.5f, cyc);  // warm
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern<KIND>, dim3(blocks), dim3(64), 0, 0, out, 0.5f, cyc);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> h(blocks);
        CHECK(hipMemcpy(h.data(), cyc, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        double avg = 0; for (auto c : h) avg += (double)c; avg /= blocks;
        const double insts = (double)ITERS * 16;
        // s_memtime ticks at 100 MHz constant? report both in-kernel ticks and wall-derived cycles
        double wall_cycles_per_inst_per_simd = (ms * 1e-3 * 2.4e9) / (insts * w);
        printf("  w=%d: %.2f cyc/inst/SIMD@2.4GHz (%.3f ms, memtime %.0f/inst-wave)", w,
               wall_cycles_per_inst_per_simd, ms, avg / insts);
    }
    printf("\n");
    CHECK(hipFree(out)); CHECK(hipFree(cyc));
}

int main()
{
    run<0>("v_mul_f32");
    run<8>("v_add_f32");
    run<1>("v_fma_f32");
    run<12>("mul+add pair/2");
    run<2>("v_pk_mul_f32");
    run<3>("v_pk_add_f32");
    run<11>("v_pk_fma_f32");
    run<4>("v_rcp_f32");
    run<5>("v_div_scale_f32");
    run<6>("v_div_fmas_f32");
    run<7>("v_div_fixup_f32");
    run<9>("v_cndmask_b32");
    run<10>("v_mov_dpp");
    run<13>("v_mul_lo_u32");
    run<14>("v_min_f32");
    run<15>("v_cmp_lt_f32");
    return 0;
}
