// hbm_stream.hip — what the HBM of this box sustains: a write-only fill (the synthesis kernel's
// traffic pattern: 16-B stores, nothing read) and a copy, float4 per lane, grid-stride.
// hipcc --offload-arch=gfx950 -O3 tools/hbm_stream.hip -o tools/hbm_stream.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void fill(float4 *__restrict__ dst, size_t n, float v)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        dst[i] = make_float4(v, v, v, v);
}

__global__ void copy(float4 *__restrict__ dst, const float4 *__restrict__ src, size_t n)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = src[i];
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main()
{
    const size_t bytes = (size_t)8 << 30;                  // 8 GiB per buffer
    const size_t n = bytes / sizeof(float4);
    float4 *a = nullptr, *b = nullptr;
    CK(hipMalloc((void **)&a, bytes));
    CK(hipMalloc((void **)&b, bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int blocks : {1024, 4096, 16384, 65536}) {
        float ms_fill = 1e9f, ms_copy = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            float ms;
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(fill, dim3(blocks), dim3(256), 0, 0, a, n, 1.0f);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep && ms < ms_fill) ms_fill = ms;
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(copy, dim3(blocks), dim3(256), 0, 0, b, a, n);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep && ms < ms_copy) ms_copy = ms;
        }
        std::printf("blocks %6d x 256: fill %7.3f ms = %6.0f GB/s written;  copy %7.3f ms = %6.0f GB/s (read + written)\n",
                    blocks, ms_fill, bytes / ms_fill / 1e6, ms_copy, 2.0 * bytes / ms_copy / 1e6);
    }
    float ms;
    CK(hipEventRecord(e0));
    CK(hipMemsetAsync(a, 0, bytes, 0));
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::printf("hipMemsetAsync: %7.3f ms = %6.0f GB/s\n", ms, bytes / ms / 1e6);
    return 0;
}
