// div_exhaustive.hip — proof by exhaustion that the short f32 division sequence
//     y = v_rcp_f32(b); e = fma(-b,y,1); y = fma(e,y,y); q = a*y; r = fma(-b,q,a); q = fma(r,y,q)
// returns the correctly rounded quotient on gfx950 for EVERY pair of 24-bit significands
// (2^46 pairs; rounding of a/b depends only on the significands while no intermediate
// under/overflows).  Correct rounding is decided with exact integer arithmetic:
// with a = A*2^-23, b = B*2^-23 in [1,2) and q = Q*2^-s, |A*2^s - B*Q| * 2 < B.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/div_exhaustive.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int STEPS>
__global__ void exhaust(unsigned b_first, unsigned long long *bad, unsigned *example)
{
    const unsigned Bm = b_first + blockIdx.x * blockDim.x + threadIdx.x;  // 23-bit mantissa of b
    const unsigned B = 0x800000u | Bm;
    const float b = __uint_as_float(0x3F800000u | Bm);
    float y = __builtin_amdgcn_rcpf(b);
    const float e = __builtin_fmaf(-b, y, 1.0f);
    y = __builtin_fmaf(e, y, y);
    unsigned long long nbad = 0;
    for (unsigned Am = 0; Am < 0x800000u; ++Am) {
        const float a = __uint_as_float(0x3F800000u | Am);
        float q = a * y;
        float r = __builtin_fmaf(-b, q, a);
        if (STEPS >= 1) q = __builtin_fmaf(r, y, q);
        if (STEPS == 2) {
            r = __builtin_fmaf(-b, q, a);
            q = __builtin_fmaf(r, y, q);
        }
        const unsigned qb = __float_as_uint(q);
        const unsigned Q = 0x800000u | (qb & 0x7FFFFFu);
        const unsigned s = 23u + (127u - (qb >> 23));       // q in [1,2): s=23; q in [.5,1): s=24
        const long long rem = (long long)((unsigned long long)(0x800000u | Am) << s) - (long long)((unsigned long long)B * Q);
        const unsigned long long ar = rem < 0 ? -rem : rem;
        const bool ok = (s == 23u || s == 24u) && (2 * ar < B);
        if (!ok) { if (!nbad) { example[0] = Am; example[1] = Bm; } ++nbad; }
    }
    if (nbad) atomicAdd(bad, nbad);
}

int main(int argc, char **argv)
{
    unsigned long long *bad; unsigned *ex;
    CHECK(hipMalloc(&bad, sizeof(*bad))); CHECK(hipMalloc(&ex, 8));
    {   // negative control: with no correction step the check must find wrong roundings
        CHECK(hipMemset(bad, 0, sizeof(*bad))); CHECK(hipMemset(ex, 0, 8));
        hipLaunchKernelGGL(exhaust<0>, dim3((1u << 16) / 256), dim3(256), 0, 0, 0x123456u & ~0xFFFFu, bad, ex);
        CHECK(hipDeviceSynchronize());
        unsigned long long h; unsigned he[2];
        CHECK(hipMemcpy(&h, bad, sizeof h, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(he, ex, 8, hipMemcpyDeviceToHost));
        printf("control, q = a*y only, 2^39 pairs: incorrectly rounded %llu (first: A=0x%06x B=0x%06x)\n", h, he[0], he[1]);
    }
    for (int steps = 1; steps <= 2; ++steps) {
        CHECK(hipMemset(bad, 0, sizeof(*bad))); CHECK(hipMemset(ex, 0, 8));
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        CHECK(hipEventRecord(e0));
        // 2^23 b-mantissas in slices so that no single dispatch runs for very long
        const unsigned slice = 1u << 20;
        for (unsigned first = 0; first < (1u << 23); first += slice) {
            if (steps == 1) hipLaunchKernelGGL(exhaust<1>, dim3(slice / 256), dim3(256), 0, 0, first, bad, ex);
            else hipLaunchKernelGGL(exhaust<2>, dim3(slice / 256), dim3(256), 0, 0, first, bad, ex);
            CHECK(hipDeviceSynchronize());
        }
        CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long h; unsigned he[2];
        CHECK(hipMemcpy(&h, bad, sizeof h, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(he, ex, 8, hipMemcpyDeviceToHost));
        printf("%d quotient correction step(s): 2^46 = 70368744177664 significand pairs checked in %.1f s, "
               "incorrectly rounded: %llu (first: A=0x%06x B=0x%06x)\n", steps, ms * 1e-3, h, he[0], he[1]);
        fflush(stdout);
    }
    return 0;
}
