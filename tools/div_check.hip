// div_check.hip — which shorter f32 division / reciprocal sequences are bit-identical to
// hipcc's correctly rounded IEEE division on gfx950, for operands in a "safe" exponent
// window (no scaling / fix-up needed)?   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// LLVM's sequence without v_div_scale / v_div_fmas scaling / v_div_fixup
__device__ __forceinline__ float div_core(float a, float b)
{
    float y = rcp(b);
    float e = fma_(-b, y, 1.0f);
    y = fma_(e, y, y);
    float q = a * y;
    float r = fma_(-b, q, a);
    q = fma_(r, y, q);
    r = fma_(-b, q, a);
    return fma_(r, y, q);
}
// one quotient correction only
__device__ __forceinline__ float div_short(float a, float b)
{
    float y = rcp(b);
    float e = fma_(-b, y, 1.0f);
    y = fma_(e, y, y);
    float q = a * y;
    float r = fma_(-b, q, a);
    return fma_(r, y, q);
}
__device__ __forceinline__ float rcp_nr1(float b)
{
    float y = rcp(b);
    float e = fma_(-b, y, 1.0f);
    return fma_(e, y, y);
}
__device__ __forceinline__ float rcp_nr2(float b)
{
    float y = rcp_nr1(b);
    float e = fma_(-b, y, 1.0f);
    return fma_(e, y, y);
}
// 1/b through the division core specialised to a = 1
__device__ __forceinline__ float rcp_core(float b) { return div_core(1.0f, b); }

__global__ void rcp_exhaustive(unsigned lo_exp, unsigned hi_exp, unsigned long long *bad)
{
    // all mantissas, exponent fields [lo_exp, hi_exp], both signs
    const unsigned long long n_exp = hi_exp - lo_exp + 1;
    const unsigned long long total = n_exp << 24;  // sign + 23 mantissa bits
    unsigned long long b1 = 0, b2 = 0, b3 = 0, b0 = 0;
    for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < total;
         i += (unsigned long long)gridDim.x * blockDim.x) {
        unsigned sign = (unsigned)(i & 1), man = (unsigned)((i >> 1) & 0x7FFFFF), ex = lo_exp + (unsigned)(i >> 24);
        float x = __uint_as_float((sign << 31) | (ex << 23) | man);
        unsigned want = __float_as_uint(1.0f / x);
        b0 += __float_as_uint(rcp(x)) != want;
        b1 += __float_as_uint(rcp_nr1(x)) != want;
        b2 += __float_as_uint(rcp_nr2(x)) != want;
        b3 += __float_as_uint(rcp_core(x)) != want;
    }
    atomicAdd(&bad[0], b0); atomicAdd(&bad[1], b1); atomicAdd(&bad[2], b2); atomicAdd(&bad[3], b3);
}

__device__ __forceinline__ unsigned long long splitmix(unsigned long long &s)
{
    unsigned long long z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ float make(unsigned long long r, int lo, int hi)
{
    unsigned man = (unsigned)(r & 0x7FFFFF);
    unsigned kind = (unsigned)((r >> 23) & 7);
    if (kind == 0) man = 0x7FFFFF;            // hard mantissas
    else if (kind == 1) man = 0;
    else if (kind == 2) man &= 0x7FF000 | 0xFFF * ((r >> 40) & 1);
    unsigned ex = 127 + lo + (unsigned)((r >> 26) % (unsigned)(hi - lo + 1));
    unsigned sign = (unsigned)(r >> 63);
    return __uint_as_float((sign << 31) | (ex << 23) | man);
}
__global__ void div_random(unsigned long long seed, int iters, int lo, int hi, unsigned long long *bad, float *ex)
{
    unsigned long long s = seed + (blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x) * 0x632BE59BD9B4E019ull;
    unsigned long long bc = 0, bs = 0;
    for (int i = 0; i < iters; ++i) {
        float a = make(splitmix(s), lo, hi), b = make(splitmix(s), lo, hi);
        unsigned want = __float_as_uint(a / b);
        if (__float_as_uint(div_core(a, b)) != want) { if (!bc) { ex[0] = a; ex[1] = b; } ++bc; }
        if (__float_as_uint(div_short(a, b)) != want) { if (!bs) { ex[2] = a; ex[3] = b; } ++bs; }
    }
    atomicAdd(&bad[0], bc); atomicAdd(&bad[1], bs);
}

int main(int argc, char **argv)
{
    unsigned long long *bad; float *ex;
    CHECK(hipMalloc(&bad, 8 * sizeof(*bad))); CHECK(hipMalloc(&ex, 4 * sizeof(float)));
    unsigned long long h[8];
    // reciprocals: exponents 2^-60..2^60 exhaustively (121 * 2^24 = 2.0e9 values)
    CHECK(hipMemset(bad, 0, 8 * sizeof(*bad)));
    hipLaunchKernelGGL(rcp_exhaustive, dim3(4096), dim3(256), 0, 0, 127 - 60, 127 + 60, bad);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(h, bad, sizeof h, hipMemcpyDeviceToHost));
    printf("reciprocal, all 2.03e9 floats with |x| in [2^-60,2^61): mismatches vs 1.0f/x: v_rcp %llu, rcp+1NR %llu, rcp+2NR %llu, div_core(1,x) %llu\n", h[0], h[1], h[2], h[3]);
    CHECK(hipMemset(bad, 0, 8 * sizeof(*bad)));
    hipLaunchKernelGGL(rcp_exhaustive, dim3(4096), dim3(256), 0, 0, 127, 127 + 60, bad);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(h, bad, sizeof h, hipMemcpyDeviceToHost));
    printf("reciprocal, |x| in [1,2^61): v_rcp %llu, rcp+1NR %llu, rcp+2NR %llu, div_core(1,x) %llu\n", h[0], h[1], h[2], h[3]);
    int iters = argc > 1 ? atoi(argv[1]) : 20000;
    for (int w : {60, 40, 20, 2}) {
        CHECK(hipMemset(bad, 0, 8 * sizeof(*bad))); CHECK(hipMemset(ex, 0, 16));
        hipLaunchKernelGGL(div_random, dim3(8192), dim3(256), 0, 0, 1234567ull + w, iters, -w, w, bad, ex);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(h, bad, sizeof h, hipMemcpyDeviceToHost));
        float he[4]; CHECK(hipMemcpy(he, ex, 16, hipMemcpyDeviceToHost));
        printf("a/b random, exponents in [-%d,%d], %.3g pairs: div_core mismatches %llu (e.g. %a / %a), div_short mismatches %llu (e.g. %a / %a)\n",
               w, w, 8192.0 * 256 * iters, h[0], he[0], he[1], h[1], he[2], he[3]);
    }
    return 0;
}
