// pcm_kernels.hip — f32 -> i16 PCM, the sample conversion of the reference's WAV sink
// (examples/cli.rs:49: `(x * std::i16::MAX as f32) as i16`: truncate toward zero, saturate,
// NaN -> 0).  Elementwise and HBM-bound: 6 B per sample (4 read + 2 written); each lane moves
// 8 samples (two 16-B loads, one 16-B store) so every wave instruction is a full 1-KiB/512-B run.
#include "kernels.h"
#include "pcm16.h"

namespace grail {

namespace {

__global__ __launch_bounds__(256) void pcm16_kernel(const float *__restrict__ in, uint64_t in_stride,
                                                    const uint32_t *__restrict__ len,
                                                    int16_t *__restrict__ out, uint64_t out_stride,
                                                    uint32_t chunks_per_row)
{
    const uint32_t u = blockIdx.x / chunks_per_row;
    const uint32_t chunk = blockIdx.x % chunks_per_row;
    const uint32_t n = len[u];
    const uint32_t t0 = (chunk * 256u + threadIdx.x) * 8u;
    if (t0 >= n) return;
    const float *src = in + (uint64_t)u * in_stride + t0;
    int16_t *dst = out + (uint64_t)u * out_stride + t0;
    const bool vec = (t0 + 8u <= n) && ((reinterpret_cast<uintptr_t>(src) & 15u) == 0) &&
                     ((reinterpret_cast<uintptr_t>(dst) & 15u) == 0);
    if (vec) {
        const float4 a = *reinterpret_cast<const float4 *>(src);
        const float4 b = *reinterpret_cast<const float4 *>(src + 4);
        uint4 o;
        o.x = (uint32_t)(pcm16_from_f32(a.x) & 0xFFFF) | ((uint32_t)pcm16_from_f32(a.y) << 16);
        o.y = (uint32_t)(pcm16_from_f32(a.z) & 0xFFFF) | ((uint32_t)pcm16_from_f32(a.w) << 16);
        o.z = (uint32_t)(pcm16_from_f32(b.x) & 0xFFFF) | ((uint32_t)pcm16_from_f32(b.y) << 16);
        o.w = (uint32_t)(pcm16_from_f32(b.z) & 0xFFFF) | ((uint32_t)pcm16_from_f32(b.w) << 16);
        *reinterpret_cast<uint4 *>(dst) = o;
    } else {
        for (uint32_t i = 0; i < 8u && t0 + i < n; ++i) dst[i] = (int16_t)pcm16_from_f32(src[i]);
    }
}

// Per-row digest of a rendered batch: the sum of the samples' bit patterns (mod 2^64), the
// largest magnitude, and the number of non-finite samples.  Lets a caller (and the full-size
// tests) compare 25 GB of output on the device instead of copying it over PCIe.
__global__ __launch_bounds__(256) void digest_kernel(const float *__restrict__ in, uint64_t in_stride,
                                                     const uint32_t *__restrict__ len,
                                                     unsigned long long *__restrict__ sums,
                                                     float *__restrict__ maxabs,
                                                     uint32_t *__restrict__ nonfinite)
{
    const uint32_t u = blockIdx.x;
    const uint32_t n = len[u];
    const float *row = in + (uint64_t)u * in_stride;
    unsigned long long s = 0;
    float m = 0.0f;
    uint32_t bad = 0;
    for (uint32_t t = threadIdx.x; t < n; t += 256u) {
        const float x = row[t];
        s += __float_as_uint(x);
        const float a = __builtin_fabsf(x);
        if (!(a <= 3.4028234663852886e38f)) ++bad;     // NaN or Inf
        else m = a > m ? a : m;
    }
    __shared__ unsigned long long ss[256];
    __shared__ float sm[256];
    __shared__ uint32_t sb[256];
    ss[threadIdx.x] = s;
    sm[threadIdx.x] = m;
    sb[threadIdx.x] = bad;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) {
            ss[threadIdx.x] += ss[threadIdx.x + k];
            sm[threadIdx.x] = sm[threadIdx.x] > sm[threadIdx.x + k] ? sm[threadIdx.x] : sm[threadIdx.x + k];
            sb[threadIdx.x] += sb[threadIdx.x + k];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        sums[u] = ss[0];
        maxabs[u] = sm[0];
        nonfinite[u] = sb[0];
    }
}

// Per-row distance between two renderings of the same batch (tolerance mode against exact mode at
// sizes no CPU oracle reaches): the largest |a - b|, the sum of squared differences, and the number of
// samples at which exactly one of the two is non-finite or the lengths disagree.
__global__ __launch_bounds__(256) void compare_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                      uint64_t stride, const uint32_t *__restrict__ len_a,
                                                      const uint32_t *__restrict__ len_b,
                                                      float *__restrict__ maxdiff, double *__restrict__ sumsq,
                                                      uint32_t *__restrict__ bad)
{
    const uint32_t u = blockIdx.x;
    const uint32_t n = len_a[u];
    const float *ra = a + (uint64_t)u * stride;
    const float *rb = b + (uint64_t)u * stride;
    float m = 0.0f;
    double q = 0.0;
    uint32_t nb = (threadIdx.x == 0 && len_b[u] != n) ? 1u : 0u;
    for (uint32_t t = threadIdx.x; t < n; t += 256u) {
        const float d = __builtin_fabsf(ra[t] - rb[t]);
        if (!(d <= 3.4028234663852886e38f)) {
            // both NaN / both the same infinity count as equal; anything else is a mismatch
            if (__float_as_uint(ra[t]) != __float_as_uint(rb[t]) && !(ra[t] != ra[t] && rb[t] != rb[t])) ++nb;
        } else {
            m = d > m ? d : m;
            q += (double)d * (double)d;
        }
    }
    __shared__ float sm[256];
    __shared__ double sq[256];
    __shared__ uint32_t sb[256];
    sm[threadIdx.x] = m;
    sq[threadIdx.x] = q;
    sb[threadIdx.x] = nb;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) {
            sm[threadIdx.x] = sm[threadIdx.x] > sm[threadIdx.x + k] ? sm[threadIdx.x] : sm[threadIdx.x + k];
            sq[threadIdx.x] += sq[threadIdx.x + k];
            sb[threadIdx.x] += sb[threadIdx.x + k];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        maxdiff[u] = sm[0];
        sumsq[u] = sq[0];
        bad[u] = sb[0];
    }
}

}  // namespace

hipError_t launch_compare(const float *a, const float *b, uint64_t stride, const uint32_t *len_a,
                          const uint32_t *len_b, uint32_t n_utt, float *maxdiff, double *sumsq, uint32_t *bad,
                          hipStream_t stream)
{
    if (n_utt == 0) return hipSuccess;
    hipLaunchKernelGGL(compare_kernel, dim3(n_utt), dim3(256), 0, stream, a, b, stride, len_a, len_b, maxdiff,
                       sumsq, bad);
    return hipGetLastError();
}

hipError_t launch_digest(const float *in, uint64_t in_stride, const uint32_t *len, uint32_t n_utt,
                         unsigned long long *sums, float *maxabs, uint32_t *nonfinite,
                         hipStream_t stream)
{
    if (n_utt == 0) return hipSuccess;
    hipLaunchKernelGGL(digest_kernel, dim3(n_utt), dim3(256), 0, stream, in, in_stride, len, sums,
                       maxabs, nonfinite);
    return hipGetLastError();
}

hipError_t launch_pcm16(const float *in, uint64_t in_stride, const uint32_t *len, uint32_t n_utt,
                        uint32_t max_len, int16_t *out, uint64_t out_stride, hipStream_t stream)
{
    if (n_utt == 0 || max_len == 0) return hipSuccess;
    const uint32_t chunks = (max_len + 2047u) / 2048u;
    hipLaunchKernelGGL(pcm16_kernel, dim3(n_utt * chunks), dim3(256), 0, stream, in, in_stride, len,
                       out, out_stride, chunks);
    return hipGetLastError();
}

}  // namespace grail
