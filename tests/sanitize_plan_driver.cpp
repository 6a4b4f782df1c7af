// sanitize_plan_driver.cpp — the host launch policy under AddressSanitizer + UBSan: voice analysis (warm-up lengths,
// sharpness), time-split grids and the block planner, fed random and extreme arguments through their C entry points.
// Built by tests/test_sanitizers.py from grail-rs_amd/csrc/{launch_plan,voice_analysis,voice_host}.cpp with g++ (those
// units make no HIP call); the three error helpers of grail_api.cpp are defined here.  Invariants checked: a plan's
// rows add up, grids are increasing multiples of 64 inside the span, sharpness / warm-up never trap on NaN or Inf.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../include/grail_hip.h"
#include <hip/hip_runtime_api.h>

namespace grail {
namespace host {
static thread_local std::string g_err;
int fail(int status, const std::string &msg)
{
    g_err = msg;
    return status;
}
int hip_fail(hipError_t, const char *what) { return fail(GRAIL_ERR_HIP, what); }
std::string &last_error() { return g_err; }
}  // namespace host
}  // namespace grail

static int fails = 0;
#define CHECK(c) do { if (!(c)) { std::printf("FAIL line %d: %s\n", __LINE__, #c); ++fails; } } while (0)

static uint32_t rng_state = 12345u;
static uint32_t rnd()
{
    rng_state = rng_state * 1664525u + 1013904223u;
    return rng_state >> 8;
}
static float special(int k)
{
    switch (k % 8) {
    case 0: return 0.0f;
    case 1: return -0.0f;
    case 2: return NAN;
    case 3: return INFINITY;
    case 4: return -INFINITY;
    case 5: return 1e-38f;
    case 6: return 3e38f;
    default: return -1.0f;
    }
}

int main()
{
    // ---- time-split grids
    int grids = 0;
    for (int t = 0; t < 20000; ++t) {
        const uint32_t span = t % 7 == 0 ? (rnd() % 8u) * 0x10000000u + rnd() : 1u + rnd() % 3000000u;
        const uint32_t warm = t % 5 == 0 ? rnd() : (rnd() % 300u) * 64u;
        const uint32_t K = t % 11 == 0 ? rnd() % 100u : 2u + rnd() % 63u;
        const uint32_t ff = t % 13 == 0 ? rnd() % 2000u : rnd() % 1001u;
        uint32_t b[64 + 2];
        std::memset(b, 0xEE, sizeof b);
        const int rc = grail_time_split_grid(span, warm, K, ff, b);
        if (K < 2u || K > 64u || ff > 1000u) CHECK(rc == GRAIL_ERR_INVALID_ARG);
        if (rc == GRAIL_OK) {
            ++grids;
            CHECK(b[0] == 0u);
            for (uint32_t k = 1; k < K; ++k) CHECK(b[k] > b[k - 1] && b[k] % 64u == 0u && b[k] < span);
            CHECK(b[K] == 0xEEEEEEEEu);               // nothing written past the K bounds
        }
    }
    CHECK(grids > 1000);
    CHECK(grail_time_split_grid(96006, 3904, 16, 165, nullptr) == GRAIL_ERR_INVALID_ARG);

    // ---- the block planner
    int plans = 0;
    std::vector<grail_plan_block> blocks(64);
    for (int t = 0; t < 12000; ++t) {
        const uint32_t cus = t % 17 == 0 ? rnd() % 5000u : 1u + rnd() % 512u;
        const int arith = t % 19 == 0 ? (int)(rnd() % 5u) - 1 : (int)(rnd() % 3u);
        const int formants = t % 23 == 0 ? (int)(rnd() % 10u) : (rnd() & 1u ? 4 : 8);
        const uint32_t warm = rnd() & 1u ? 0u : (1u + rnd() % 256u) * 64u;
        const uint32_t rows = t % 29 == 0 ? rnd() : rnd() % 300000u;
        const uint32_t span = t % 31 == 0 ? rnd() * 16u : 1u + rnd() % 2000000u;
        uint32_t n = 0xFFFFFFFFu;
        const uint32_t cap = rnd() % 65u;
        const int rc = grail_plan_blocks(cus, arith, formants, warm, rows, span, blocks.data(), cap, &n);
        const bool bad = cus == 0u || cus > 4096u || (formants != 4 && formants != 8) || arith < 0 || arith > 2;
        CHECK((rc == GRAIL_ERR_INVALID_ARG) == bad);
        if (rc != GRAIL_OK) continue;
        ++plans;
        CHECK((n == 0u) == (rows == 0u));
        if (n <= cap) {
            uint64_t sum = 0;
            for (uint32_t i = 0; i < n; ++i) {
                sum += blocks[i].rows;
                CHECK(blocks[i].rows > 0u && (blocks[i].formants == 4u || blocks[i].formants == 8u));
                CHECK(blocks[i].fast <= 2u && blocks[i].chunks <= 64u && std::isfinite(blocks[i].model_ms));
                if (!arith) CHECK(blocks[i].fast == 0u && blocks[i].chunks == 0u && blocks[i].scan == 0u);
            }
            CHECK(sum == rows);
        }
    }
    CHECK(plans > 1000);
    uint32_t n = 0;
    CHECK(grail_plan_blocks(256, 0, 4, 0, 70000, 96006, nullptr, 0, &n) == GRAIL_OK && n == 2u);
    CHECK(grail_plan_blocks(256, 0, 4, 0, 70000, 96006, nullptr, 0, nullptr) == GRAIL_ERR_INVALID_ARG);

    // ---- ... of batches whose utterances differ in length (grail_plan_ragged_blocks): random, unsorted and extreme rows
    int ragged = 0;
    std::vector<uint32_t> samples, segs, kinks;
    for (int t = 0; t < 400; ++t) {
        const uint32_t cus = 1u + rnd() % 300u;
        const uint32_t rows = t % 9 == 0 ? rnd() % 9u : 1u + rnd() % 200000u;
        samples.resize(rows);
        segs.resize(rows);
        kinks.resize(rows);
        uint32_t len = t % 7 == 0 ? 0xFFFFFFFFu : 64u + rnd() % 400000u;
        for (uint32_t r = 0; r < rows; ++r) {
            samples[r] = t % 5 == 0 ? rnd() * (rnd() % 300u) : len;       // (every fifth: unsorted, up to 2^32)
            len -= len > 8u ? rnd() % 8u : 0u;
            segs[r] = t % 11 == 0 ? rnd() * 256u : rnd() % 64u;
            kinks[r] = t % 13 == 0 ? 0xFFFFFFFFu : rnd() % 64u;
        }
        uint32_t nb = 0xFFFFFFFFu;
        const uint32_t cap = rnd() % 65u;
        const int rc = grail_plan_ragged_blocks(cus, (int)(rnd() % 3u), rnd() & 1u ? 4 : 8, rnd() & 1u ? 0u : 3904u, rows,
                                                samples.data(), t % 3 ? segs.data() : nullptr, t % 4 ? kinks.data() : nullptr,
                                                blocks.data(), cap, &nb);
        CHECK(rc == GRAIL_OK);
        CHECK((nb == 0u) == (rows == 0u));
        if (nb <= cap) {
            uint64_t sum = 0;
            for (uint32_t i = 0; i < nb; ++i) {
                sum += blocks[i].rows;
                CHECK(blocks[i].rows > 0u && !std::isnan(blocks[i].model_ms) && blocks[i].model_ms >= 0.0f);
            }
            CHECK(sum == rows);
        }
        ++ragged;
    }
    CHECK(grail_plan_ragged_blocks(256, 0, 4, 0, 10, nullptr, nullptr, nullptr, blocks.data(), 64, &n) == GRAIL_ERR_INVALID_ARG);
    CHECK(ragged == 400);

    // ---- the dispatcher's model and the packed launch order: random, equal, hostile costs; every device shape
    int orders = 0;
    std::vector<double> cost;
    std::vector<uint32_t> order;
    for (int t = 0; t < 600; ++t) {
        const uint32_t cus = t % 7 == 0 ? 1u + rnd() % 4096u : (t % 3 == 0 ? 32u * (1u + rnd() % 16u) : 256u);
        const uint32_t wpw = rnd() & 1u ? 1u : 4u;
        const uint32_t n_wg = t % 11 == 0 ? rnd() % 3u : 1u + rnd() % 6000u;
        cost.resize(n_wg);
        order.assign(n_wg, 0xFFFFFFFFu);
        for (uint32_t b = 0; b < n_wg; ++b)
            cost[b] = t % 13 == 0 ? (double)special((int)rnd()) : (t % 5 == 0 ? 1.0 : 0.001 * (double)(1u + rnd() % 100000u));
        double plain = -1.0, packed = -1.0;
        CHECK(grail_dispatch_model(cus, wpw, cost.data(), nullptr, n_wg, &plain) == GRAIL_OK);
        CHECK(grail_packed_launch_order(cus, wpw, cost.data(), n_wg, order.data()) == GRAIL_OK);
        std::vector<uint8_t> seen(n_wg, 0);
        for (uint32_t b = 0; b < n_wg; ++b) {
            CHECK(order[b] < n_wg);
            if (order[b] < n_wg) {
                CHECK(!seen[order[b]]);
                seen[order[b]] = 1;
            }
        }
        CHECK(grail_dispatch_model(cus, wpw, cost.data(), order.data(), n_wg, &packed) == GRAIL_OK);
        CHECK(plain >= 0.0 && packed >= 0.0 && !std::isnan(plain) && !std::isnan(packed));
        if (t % 13 != 0 && n_wg) {
            double sum = 0.0, longest = 0.0;
            for (double c : cost) { sum += c; longest = std::fmax(longest, c); }
            CHECK(plain >= longest - 1e-9 && packed >= longest - 1e-9 && packed <= sum + 1e-6);
        }
        ++orders;
    }
    double ms = 0.0;
    CHECK(grail_dispatch_model(256, 3, cost.data(), nullptr, 1, &ms) == GRAIL_ERR_INVALID_ARG);
    CHECK(grail_packed_launch_order(0, 1, cost.data(), 1, order.data()) == GRAIL_ERR_INVALID_ARG);
    CHECK(orders == 600);

    // ---- voice analysis on sane, random and hostile tables
    grail_voice v;
    grail_voice_generic_at(&v, 48000.0f);
    const float s0 = grail_fast_sharpness(&v);
    CHECK(s0 > 20.0f && s0 < 28.0f);
    CHECK(grail_time_split_warmup(&v) == 3904u);
    CHECK(std::isinf(grail_fast_sharpness(nullptr)) && grail_time_split_warmup(nullptr) == 0u);
    for (int t = 0; t < 20000; ++t) {
        grail_voice w = v;
        float *f = (float *)&w;
        const size_t nf = sizeof w / sizeof(float);
        const int edits = 1 + (int)(rnd() % 6u);
        for (int e = 0; e < edits; ++e) {
            const size_t at = rnd() % nf;
            f[at] = rnd() & 1u ? special((int)rnd()) : f[at] * (0.01f + (float)(rnd() % 4000u) * 0.001f);
        }
        const float s = grail_fast_sharpness(&w);
        CHECK(!(s < 0.0f));                               // a number >= 0, +inf or NaN-free by construction
        CHECK(!std::isnan(s));
        const uint32_t wu = grail_time_split_warmup(&w);
        CHECK(wu <= 16384u + 64u && wu % 64u == 0u);
    }
    if (fails) {
        std::printf("sanitize plan driver: %d failure(s)\n", fails);
        return 1;
    }
    std::printf("sanitize plan driver: ok (%d grids, %d plans)\n", grids, plans);
    return 0;
}
