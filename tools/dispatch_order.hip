// dispatch_order.hip — how the hardware hands workgroups to SIMDs when a launch has more one-wave workgroups than the device
// has SIMDs and each holds a SIMD alone (512 registers, like the one-lane synthesis kernels).  Workgroup b spins for
// cost_us[b] microseconds and records where and when it ran: XCC, SE / CU / SIMD (HW_ID), start and end (s_memrealtime,
// 100 MHz).  tools/dispatch_order.py feeds it launch orders and checks the list-scheduling model of launch_plan.cpp.
// build: hipcc --offload-arch=gfx950 -O2 tools/dispatch_order.hip -o tools/dispatch_order.bin
// usage: dispatch_order.bin costs.u32 records.u64 [waves per workgroup: 1 (default) or 4]   (records: blocks x {start, end, hw_id, xcc_id})
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

#define CHECK(x)                                                                       \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));               \
            return 1;                                                                  \
        }                                                                              \
    } while (0)

template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void spin(const uint32_t *cost_us, uint64_t *rec)
{
    // all 512 registers of the wave's budget: one wave per SIMD, as the kernels this stands for
    asm volatile("v_mov_b32 v255, 0\n\tv_accvgpr_write_b32 a255, 0" ::: "v255", "a255");
    const uint64_t t0 = __builtin_readcyclecounter() * 0 + wall_clock64();
    const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID
    const uint32_t xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // HW_REG_XCC_ID
    const uint64_t ticks = (uint64_t)cost_us[blockIdx.x] * 100ull;
    uint64_t t1 = t0;
    while (t1 - t0 < ticks) {
        __builtin_amdgcn_s_sleep(8);
        t1 = wall_clock64();
    }
    if (threadIdx.x == 0) {
        uint64_t *r = rec + 4ull * blockIdx.x;
        r[0] = t0;
        r[1] = t1;
        r[2] = hw;
        r[3] = xcc;
    }
}

int main(int argc, char **argv)
{
    if (argc < 3) {
        std::fprintf(stderr, "usage: %s costs.u32 records.u64\n", argv[0]);
        return 2;
    }
    std::FILE *f = std::fopen(argv[1], "rb");
    if (!f) return 2;
    std::vector<uint32_t> cost;
    uint32_t v;
    while (std::fread(&v, 4, 1, f) == 1) cost.push_back(v);
    std::fclose(f);
    const size_t n = cost.size();
    uint32_t *d_cost = nullptr;
    uint64_t *d_rec = nullptr;
    CHECK(hipMalloc((void **)&d_cost, n * 4));
    CHECK(hipMalloc((void **)&d_rec, n * 32));
    CHECK(hipMemcpy(d_cost, cost.data(), n * 4, hipMemcpyHostToDevice));
    for (int rep = 0; rep < 2; ++rep) {                     // (the second launch is the one recorded: clocks up)
        if (argc > 3 && argv[3][0] == '4') hipLaunchKernelGGL(spin<4>, dim3((unsigned)n), dim3(256), 0, 0, d_cost, d_rec);
        else hipLaunchKernelGGL(spin<1>, dim3((unsigned)n), dim3(64), 0, 0, d_cost, d_rec);
        CHECK(hipDeviceSynchronize());
    }
    std::vector<uint64_t> rec(n * 4);
    CHECK(hipMemcpy(rec.data(), d_rec, n * 32, hipMemcpyDeviceToHost));
    f = std::fopen(argv[2], "wb");
    if (!f) return 2;
    std::fwrite(rec.data(), 8, rec.size(), f);
    std::fclose(f);
    return 0;
}
