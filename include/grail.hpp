// grail.hpp — header-only C++ facade over the C ABI (grail_hip.h) that keeps the call shape of
// the grail-rs crate for this path:
//
//   Rust (reference examples/cli.rs:175-184)            C++ (this header)
//   text.chars().transcribe(lang).intonate(lang, v)  -> grail::phoneme_elems(v, text)
//       .select(v).sequence(v).jitter(seed, v)
//       .synthesize().collect::<Vec<f32>>()           -> gpu.synthesize({Utterance{...}})
//   voices::generic()                                 -> grail::voices::generic()
//   pulling the iterator a buffer at a time             -> grail::Stream(gpu, utterances, chunk).next(...)
//       (examples/interactive.rs:31-48)
//   a chain whose source delivers while it runs         -> grail::LiveStream(gpu, chunk).append(...) / .next()
//       (repeat_with(|| receiver.try_recv()...), interactive.rs:31)
//   the same one call over every GPU of the node        -> grail::Node({0, 1, ...}, voices).synthesize({...})
//
// No arithmetic lives here; everything forwards to libgrail_hip.so.
#pragma once

#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "grail_hip.h"

namespace grail {

using SynthesisElem = grail_synthesis_elem;  // src/lib.rs:316
using Voice = grail_voice;                   // src/lib.rs:696
using PhonemeElem = grail_phoneme_elem;      // src/lib.rs:961
using SequenceElem = grail_sequence_elem;    // src/lib.rs:814

enum class Phoneme : int32_t {               // src/lib.rs:632-649
    Silence = GRAIL_PH_SILENCE, Stop = GRAIL_PH_STOP, Glide = GRAIL_PH_GLIDE,
    A = GRAIL_PH_A, E = GRAIL_PH_E,
};

struct Error : std::runtime_error {
    int status;
    Error(int s, const std::string &what) : std::runtime_error(what), status(s) {}
};

inline void check(int status)
{
    if (status != GRAIL_OK) {
        const char *msg = grail_last_error();
        throw Error(status, (msg && *msg) ? msg : grail_status_string(status));
    }
}

namespace voices {
inline Voice generic() { Voice v; grail_voice_generic(&v); return v; }             // generic.rs:5
// predicted |fast - reference| of a voice, units of 2^-23 of max(1, peak); served up to GRAIL_FAST_SHARPNESS_LIMIT
inline float fast_sharpness(const Voice &voice) { return grail_fast_sharpness(&voice); }
inline Voice generic_at(float sample_rate) { Voice v; grail_voice_generic_at(&v, sample_rate); return v; }
}  // namespace voices

// text.chars().transcribe(languages::generic()).intonate(languages::generic(), voice)
inline std::vector<PhonemeElem> phoneme_elems(const Voice &voice, const std::string &text_utf8)
{
    uint32_t n = 0;
    check(grail_text_to_phoneme_elems(&voice, text_utf8.c_str(), nullptr, 0, &n));
    std::vector<PhonemeElem> out(n);
    if (n) check(grail_text_to_phoneme_elems(&voice, text_utf8.c_str(), out.data(), n, &n));
    return out;
}

struct Utterance {
    std::vector<PhonemeElem> phonemes;
    uint32_t voice = 0;
    uint32_t jitter_seed = 0;   // examples/cli.rs:182 uses 0
};

// One GPU and its voice table.
class Gpu {
public:
    Gpu(int device, const std::vector<Voice> &voices) : voices_(voices)
    {
        check(grail_create(device, &ctx_));
        const int rc = grail_set_voices(ctx_, voices_.data(), (uint32_t)voices_.size());
        if (rc != GRAIL_OK) {
            grail_destroy(ctx_);
            check(rc);
        }
    }
    ~Gpu() { grail_destroy(ctx_); }
    Gpu(const Gpu &) = delete;
    Gpu &operator=(const Gpu &) = delete;

    grail_ctx *ctx() const { return ctx_; }
    const std::vector<Voice> &voices() const { return voices_; }

    // Exact (default): every sample bit-identical to the reference arithmetic.  Fast: the stated-tolerance
    // mode (|fast - exact| <= GRAIL_FAST_TOLERANCE; clocks, phases, wraps and noise generators stay exact).
    enum class Arithmetic { Exact = 0, Fast = 1 };
    void set_arithmetic(Arithmetic a) const { check(grail_set_option(ctx_, "arithmetic", (int64_t)a)); }
    // Fast is served for voice tables up to a sharpness of their resonances (grail_fast_sharpness); sharper tables
    // are rendered by the exact kernels whatever set_arithmetic says.
    bool fast_arithmetic_served() const
    {
        int64_t v = 0;
        check(grail_get_option(ctx_, "fast_arithmetic_served", &v));
        return v != 0;
    }

    // utterances.map(|u| u.phonemes.select(v).sequence(v).jitter(seed, v).synthesize().collect())
    std::vector<std::vector<float>> synthesize(const std::vector<Utterance> &utts) const
    {
        std::vector<PhonemeElem> segs;
        std::vector<uint32_t> offs(1, 0u), vids, seeds;
        for (const Utterance &u : utts) {
            segs.insert(segs.end(), u.phonemes.begin(), u.phonemes.end());
            offs.push_back((uint32_t)segs.size());
            vids.push_back(u.voice);
            seeds.push_back(u.jitter_seed);
        }
        const uint32_t n = (uint32_t)utts.size();
        std::vector<uint32_t> lens(n ? n : 1);
        grail_batch *b = nullptr;
        check(grail_batch_upload(ctx_, segs.data(), offs.data(), vids.data(), seeds.data(), n, &b));
        const int rc = grail_batch_lengths(ctx_, b, 0xFFFFFFFFu, lens.data());
        grail_batch_free(ctx_, b);
        check(rc);
        uint64_t stride = 64;
        for (uint32_t i = 0; i < n; ++i) stride = lens[i] > stride ? lens[i] : stride;
        stride = (stride + 63) / 64 * 64;
        std::vector<float> flat((size_t)n * stride);
        check(grail_synthesize_batch(ctx_, segs.data(), offs.data(), vids.data(), seeds.data(), n,
                                     flat.data(), stride, lens.data(), GRAIL_OUT_HOST));
        std::vector<std::vector<float>> out(n);
        for (uint32_t i = 0; i < n; ++i)
            out[i].assign(flat.begin() + (size_t)i * stride, flat.begin() + (size_t)i * stride + lens[i]);
        return out;
    }

    // the whole chain of examples/cli.rs:175-184 for several texts, voice 0, seed 0
    std::vector<std::vector<float>> say(const std::vector<std::string> &texts) const
    {
        std::vector<Utterance> utts;
        for (const std::string &t : texts) utts.push_back(Utterance{phoneme_elems(voices_.at(0), t), 0, 0});
        return synthesize(utts);
    }

    // examples/cli.rs:49 on the device, then save_wav (cli.rs:28-67)
    void save_wav(const std::string &path, const std::vector<float> &pcm, uint32_t sample_rate) const
    {
        const uint32_t n = (uint32_t)pcm.size();
        void *d_in = nullptr, *d_out = nullptr, *d_len = nullptr;
        std::vector<int16_t> i16(n);
        check(grail_device_alloc(ctx_, (size_t)n * 4 + 4, &d_in));
        check(grail_device_alloc(ctx_, (size_t)n * 2 + 16, &d_out));
        check(grail_device_alloc(ctx_, 4, &d_len));
        int rc = grail_memcpy_h2d(ctx_, d_in, pcm.data(), (size_t)n * 4);
        if (!rc) rc = grail_memcpy_h2d(ctx_, d_len, &n, 4);
        if (!rc) rc = grail_pcm16_async(ctx_, (const float *)d_in, n, (const uint32_t *)d_len, 1, n,
                                        (int16_t *)d_out, n);
        if (!rc) rc = grail_sync(ctx_);
        if (!rc && n) rc = grail_memcpy_d2h(ctx_, i16.data(), d_out, (size_t)n * 2);
        grail_device_free(ctx_, d_in);
        grail_device_free(ctx_, d_out);
        grail_device_free(ctx_, d_len);
        check(rc);
        check(grail_wav_write_i16(path.c_str(), i16.data(), n, sample_rate));
    }

private:
    grail_ctx *ctx_ = nullptr;
    std::vector<Voice> voices_;
};

// Every GPU of the node behind the same call (grail_node_*): one context and one host thread per device, the voice table
// carried to the others' HBM by one ncclBroadcast, the batch cut into contiguous shards that render concurrently into
// slices of one host buffer.  Results are those of Gpu::synthesize, row for row, bit for bit (Exact).
class Node {
public:
    // voices_without_rccl: tests on a box whose `devices` name one GPU more than once (RCCL refuses such a communicator)
    Node(const std::vector<int> &devices, const std::vector<Voice> &voices, bool voices_without_rccl = false) : voices_(voices)
    {
        check(grail_node_create(devices.data(), (uint32_t)devices.size(), &node_));
        int rc = voices_without_rccl ? grail_node_set_option(node_, "node_voices_without_rccl", 1) : GRAIL_OK;
        if (!rc) rc = grail_node_set_voices(node_, voices_.data(), (uint32_t)voices_.size());
        if (rc != GRAIL_OK) {
            grail_node_destroy(node_);
            check(rc);
        }
    }
    ~Node() { grail_node_destroy(node_); }
    Node(const Node &) = delete;
    Node &operator=(const Node &) = delete;

    grail_node *node() const { return node_; }
    uint32_t size() const { return grail_node_size(node_); }
    const std::vector<Voice> &voices() const { return voices_; }
    void set_arithmetic(Gpu::Arithmetic a) const { check(grail_node_set_option(node_, "arithmetic", (int64_t)a)); }
    // ranks RCCL reports for the node's communicator (ncclCommCount; 0 under voices_without_rccl)
    uint32_t rccl_ranks() const
    {
        int64_t v = 0;
        check(grail_node_get_option(node_, "node_rccl_ranks", &v));
        return (uint32_t)v;
    }

    std::vector<std::vector<float>> synthesize(const std::vector<Utterance> &utts) const
    {
        std::vector<PhonemeElem> segs;
        std::vector<uint32_t> offs(1, 0u), vids, seeds;
        for (const Utterance &u : utts) {
            segs.insert(segs.end(), u.phonemes.begin(), u.phonemes.end());
            offs.push_back((uint32_t)segs.size());
            vids.push_back(u.voice);
            seeds.push_back(u.jitter_seed);
        }
        const uint32_t n = (uint32_t)utts.size();
        std::vector<uint32_t> lens(n ? n : 1);
        check(grail_node_lengths(node_, segs.data(), offs.data(), vids.data(), n, 0xFFFFFFFFu, lens.data()));
        uint64_t stride = 64;
        for (uint32_t i = 0; i < n; ++i) stride = lens[i] > stride ? lens[i] : stride;
        stride = (stride + 63) / 64 * 64;
        std::vector<float> flat((size_t)n * stride);
        check(grail_node_synthesize_batch(node_, segs.data(), offs.data(), vids.data(), seeds.data(), n, flat.data(),
                                          stride, lens.data(), GRAIL_OUT_HOST));
        std::vector<std::vector<float>> out(n);
        for (uint32_t i = 0; i < n; ++i)
            out[i].assign(flat.begin() + (size_t)i * stride, flat.begin() + (size_t)i * stride + lens[i]);
        return out;
    }

    std::vector<std::vector<float>> say(const std::vector<std::string> &texts) const
    {
        std::vector<Utterance> utts;
        for (const std::string &t : texts) utts.push_back(Utterance{phoneme_elems(voices_.at(0), t), 0, 0});
        return synthesize(utts);
    }

private:
    grail_node *node_ = nullptr;
    std::vector<Voice> voices_;
};

// The lazy use of the chain: the crate's iterator is pulled a buffer at a time
// (examples/interactive.rs:31-48).  Here every utterance of the batch yields its next `chunk`
// samples per call; the iterator state stays in HBM between calls (grail_stream_*), and the
// chunks concatenate to exactly what Gpu::synthesize returns.
class Stream {
public:
    Stream(const Gpu &gpu, const std::vector<Utterance> &utts, uint32_t chunk)
        : ctx_(gpu.ctx()), n_((uint32_t)utts.size()), chunk_(chunk), stride_(((uint64_t)chunk + 63) / 64 * 64)
    {
        std::vector<PhonemeElem> segs;
        std::vector<uint32_t> offs(1, 0u), vids, seeds;
        for (const Utterance &u : utts) {
            segs.insert(segs.end(), u.phonemes.begin(), u.phonemes.end());
            offs.push_back((uint32_t)segs.size());
            vids.push_back(u.voice);
            seeds.push_back(u.jitter_seed);
        }
        check(grail_batch_upload(ctx_, segs.data(), offs.data(), vids.data(), seeds.data(), n_, &batch_));
        int rc = grail_stream_open(ctx_, batch_, &stream_);
        if (!rc) rc = grail_device_alloc(ctx_, (size_t)(n_ ? n_ : 1) * stride_ * sizeof(float), &d_out_);
        if (!rc) rc = grail_device_alloc(ctx_, (size_t)(n_ ? n_ : 1) * sizeof(uint32_t), &d_len_);
        if (rc) {
            release();
            check(rc);
        }
        host_.resize((size_t)n_ * stride_);
        lens_.resize(n_);
    }
    ~Stream() { release(); }
    Stream(const Stream &) = delete;
    Stream &operator=(const Stream &) = delete;

    // Iterator::next, `chunk` samples at a time: false once every chain has returned None
    bool next(std::vector<std::vector<float>> &chunks)
    {
        chunks.assign(n_, {});
        if (n_ == 0 || chunk_ == 0) return false;
        check(grail_stream_next_async(ctx_, stream_, chunk_, (float *)d_out_, stride_, (uint32_t *)d_len_));
        check(grail_sync(ctx_));
        check(grail_memcpy_d2h(ctx_, lens_.data(), d_len_, (size_t)n_ * sizeof(uint32_t)));
        check(grail_memcpy_d2h(ctx_, host_.data(), d_out_, host_.size() * sizeof(float)));
        bool any = false;
        for (uint32_t u = 0; u < n_; ++u) {
            chunks[u].assign(host_.begin() + (size_t)u * stride_, host_.begin() + (size_t)u * stride_ + lens_[u]);
            any = any || lens_[u] != 0;
        }
        return any;
    }

private:
    void release()
    {
        if (stream_) grail_stream_close(ctx_, stream_);
        if (batch_) grail_batch_free(ctx_, batch_);
        if (d_out_) grail_device_free(ctx_, d_out_);
        if (d_len_) grail_device_free(ctx_, d_len_);
        stream_ = nullptr; batch_ = nullptr; d_out_ = nullptr; d_len_ = nullptr;
    }
    grail_ctx *ctx_;
    uint32_t n_, chunk_;
    uint64_t stride_;
    grail_batch *batch_ = nullptr;
    grail_stream *stream_ = nullptr;
    void *d_out_ = nullptr, *d_len_ = nullptr;
    std::vector<float> host_;
    std::vector<uint32_t> lens_;
};

// The live use of the chain (examples/interactive.rs:31-48): ONE chain runs for the whole session, its source delivers
// while the audio callback is pulling, and carrier phase, noise seed, jitter and filter state carry across everything that
// is ever said.  Segments are appended whenever they are known; next() yields the next `chunk` samples, or fewer when
// the Sequencer is waiting for a segment that has not been appended yet (src/lib.rs:866-888 pulls iter.next() on demand)
// — the front end then feeds it, a Silence when no text is waiting, exactly as the reference's source hands over ' '.
class LiveStream {
public:
    LiveStream(const Gpu &gpu, uint32_t chunk, uint32_t voice = 0, uint32_t jitter_seed = 0, uint32_t ring_segments = 0)
        : ctx_(gpu.ctx()), chunk_(chunk), stride_(((uint64_t)chunk + 63) / 64 * 64)
    {
        check(grail_stream_open_live(ctx_, 1, &voice, &jitter_seed, ring_segments, 0, &stream_));
        int rc = grail_device_alloc(ctx_, (size_t)stride_ * sizeof(float), &d_out_);
        if (!rc) rc = grail_device_alloc(ctx_, sizeof(uint32_t), &d_len_);
        if (rc) {
            release();
            check(rc);
        }
    }
    ~LiveStream() { release(); }
    LiveStream(const LiveStream &) = delete;
    LiveStream &operator=(const LiveStream &) = delete;

    // the source delivers: these segments follow what the chain already has
    void append(const std::vector<PhonemeElem> &segs)
    {
        const uint32_t offs[2] = {0u, (uint32_t)segs.size()};
        check(grail_stream_append(ctx_, stream_, segs.data(), offs));
    }
    // the source has ended: what is pending is spoken, the last segment fades out, next() then returns nothing
    void finish() { check(grail_stream_finish(ctx_, stream_, nullptr)); }
    // segments appended that the Sequencer has not pulled yet
    uint32_t pending()
    {
        uint32_t n = 0;
        check(grail_stream_pending(ctx_, stream_, &n));
        return n;
    }
    // Iterator::next, up to `chunk` samples: fewer when the Sequencer waits for the source (or the chain has ended)
    std::vector<float> next()
    {
        uint32_t n = 0;
        check(grail_stream_next_async(ctx_, stream_, chunk_, (float *)d_out_, stride_, (uint32_t *)d_len_));
        check(grail_sync(ctx_));
        check(grail_memcpy_d2h(ctx_, &n, d_len_, sizeof n));
        std::vector<float> out(n);
        if (n) check(grail_memcpy_d2h(ctx_, out.data(), d_out_, (size_t)n * sizeof(float)));
        return out;
    }
    uint32_t chunk() const { return chunk_; }

private:
    void release()
    {
        if (stream_) grail_stream_close(ctx_, stream_);
        if (d_out_) grail_device_free(ctx_, d_out_);
        if (d_len_) grail_device_free(ctx_, d_len_);
        stream_ = nullptr; d_out_ = nullptr; d_len_ = nullptr;
    }
    grail_ctx *ctx_;
    uint32_t chunk_;
    uint64_t stride_;
    grail_stream *stream_ = nullptr;
    void *d_out_ = nullptr, *d_len_ = nullptr;
};

}  // namespace grail
