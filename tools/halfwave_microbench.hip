// Does a wave64 whose upper 32 lanes are inactive issue VALU in half the cycles on gfx950?
// Same instruction stream, 64- vs 32-thread workgroups (one wave each), 1 and 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
constexpr int ITERS = 20000;
typedef float f2 __attribute__((ext_vector_type(2)));
template <int KIND>
__global__ void kern(float *out, float seedf)
{
    f2 p[8], q[8]; float a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = seedf + threadIdx.x * 0.001f + i; b[i] = 1.0f + i * 0.125f + seedf; p[i] = f2{a[i], b[i]}; q[i] = f2{b[i], a[i]}; }
    for (int it = 0; it < ITERS; ++it) {
#define REP8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
        if constexpr (KIND == 0) {
#define OP(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(q[i]));
            REP8(OP) REP8(OP)
#undef OP
        } else if constexpr (KIND == 1) {
#define OP(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            REP8(OP) REP8(OP)
#undef OP
        } else {
#define OP(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
            REP8(OP) REP8(OP)
#undef OP
        }
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int KIND> void run(const char *name)
{
    float *out; CHECK(hipMalloc(&out, 256 * 4 * 8 * 64 * sizeof(float)));
    for (int threads : {64, 32}) for (int w : {1, 2}) {
        int blocks = 256 * 4 * w;
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(kern<KIND>, dim3(blocks), dim3(threads), 0, 0, out, 0.5f); CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0)); hipLaunchKernelGGL(kern<KIND>, dim3(blocks), dim3(threads), 0, 0, out, 0.5f); CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-14s active lanes/wave %2d, waves/SIMD %d: %.3f ms  (%.2f cyc/inst/wave @2.4GHz)\n", name, threads, w, ms, ms * 1e-3 * 2.4e9 / (ITERS * 16.0));
    }
    CHECK(hipFree(out));
}
int main() { run<0>("v_pk_mul_f32"); run<1>("v_mul_f32"); run<2>("v_rcp_f32"); return 0; }
