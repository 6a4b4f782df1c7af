// live_stream.hip — the segment rings of live streams (grail_stream_open_live / grail_stream_append).
// The reference's chain is lazy: Sequencer::next pulls iter.next() when a segment runs out (src/lib.rs:866-888), and
// in examples/interactive.rs:31-38 that source never ends — text arrives while the chain is running.  A live stream
// keeps every utterance's pending segments in a ring in HBM; this kernel appends to the rings of a whole batch of
// streams in one launch (the scatter happens on the device: the host keeps no mirror of the rings).
#include "kernels.h"

namespace grail {
namespace {

// one thread per (utterance, new segment) would need a prefix sum over the batch; appends are short (a few segments
// per utterance at a time), so: one thread per utterance, a loop over its new segments
__global__ __launch_bounds__(64) void ring_append_kernel(DevSeg *ring, float *ring_elems, uint32_t *counts,
                                                         const uint32_t ring_cap, const DevSeg *__restrict__ new_segs,
                                                         const float *__restrict__ new_elems,
                                                         const uint32_t *__restrict__ new_offsets, const uint32_t n_utt)
{
    const uint32_t u = blockIdx.x * 64u + threadIdx.x;
    if (u >= n_utt) return;
    const uint32_t lo = new_offsets[u], hi = new_offsets[u + 1];
    if (hi <= lo) return;
    uint32_t have = counts[u];
    for (uint32_t i = lo; i < hi; ++i, ++have) {
        const uint32_t row = u * ring_cap + (have & (ring_cap - 1u));
        DevSeg s = new_segs[i];
        if (new_elems) {
            // caller-built SequenceElems: the elem travels with the segment, `elem` names its row (or -1 = None, :817)
            if (s.elem >= 0) {
                s.elem = (int32_t)row;
                for (int k = 0; k < ELEM_FLOATS; ++k) ring_elems[(size_t)row * ELEM_FLOATS + k] = new_elems[(size_t)i * ELEM_FLOATS + k];
            }
        }
        ring[row] = s;
    }
    counts[u] = have;
}

}  // namespace

hipError_t launch_ring_append(DevSeg *ring, float *ring_elems, uint32_t *counts, uint32_t ring_cap, const DevSeg *new_segs,
                              const float *new_elems, const uint32_t *new_offsets, uint32_t n_utt, hipStream_t stream)
{
    if (n_utt == 0) return hipSuccess;
    hipLaunchKernelGGL(ring_append_kernel, dim3((n_utt + 63u) / 64u), dim3(64), 0, stream, ring, ring_elems, counts, ring_cap,
                       new_segs, new_elems, new_offsets, n_utt);
    return hipGetLastError();
}

}  // namespace grail
