// wg_placement.hip — do the waves of one workgroup land on different SIMDs of the CU?
// Each wave runs the same issue-bound VALU loop.  If a 256-thread workgroup (4 waves) takes as long as
// a 64-thread one, its waves ran side by side on four SIMDs; if it takes ~2-4x, they shared SIMDs.
// hipcc --offload-arch=gfx950 -O3 tools/wg_placement.hip -o tools/wg_placement.bin
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void spin(float *out, int iters)
{
    float a = threadIdx.x * 1e-3f, b = 1.0001f, c = 0.5f, d = 0.25f;
    for (int i = 0; i < iters; ++i) {
        a = a * b + c; c = c * b + d; d = d * b + a; b = b * 0.99999f + 1e-6f;
        a = a * b + c; c = c * b + d; d = d * b + a; b = b * 0.99999f + 1e-6f;
    }
    if (a + b + c + d == 12345.678f) out[threadIdx.x] = a;
}

int main()
{
    float *d = nullptr;
    hipMalloc((void **)&d, 4096);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 200000;
    for (int blocks : {256, 512, 1024}) {
        for (int threads : {64, 128, 256, 512, 1024}) {
            float best = 1e9f;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(spin, dim3(blocks), dim3(threads), 0, 0, d, iters);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            std::printf("blocks %5d x %4d threads (%2d waves each): %7.3f ms\n", blocks, threads, threads / 64, best);
        }
    }
    return 0;
}
