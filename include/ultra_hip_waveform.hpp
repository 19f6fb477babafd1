// ultra_hip_waveform.hpp — header-only C++ host side over the C-ABI (include/ultra_hip.h).
//
// Classes with the reference's own method names, argument meaning and failure behaviour:
//   ultra_hip::HipLDPCDecoder      ~ ultra::LDPCDecoder      (include/ultra/fec.hpp:48-77)
//   ultra_hip::HipOfdmDemodulator  ~ ultra::OFDMDemodulator  (include/ultra/ofdm.hpp:58-127): the whole state machine of
//                                    process() / processPresynced() / getSoftBits() on the device, as a LIVE stream
//   ultra_hip::HipOfdmCoxWaveform  ~ ultra::OFDMNvisWaveform (src/waveform/ofdm_cox_waveform.cpp), WaveformMode::OFDM_COX
//   ultra_hip::HipOfdmWaveform     ~ ultra::OFDMChirpWaveform (src/waveform/ofdm_chirp_waveform.cpp), OFDM_CHIRP, with the
//                                    dual-chirp detection on the device too
//   ultra_hip::HipRxFrameDecoder   ~ the decode half of ultra::gui::RxPipeline
//
// Compiled inside the reference tree (-DULTRA_HIP_WITH_REFERENCE, -I<ref>/include -I<ref>/src) the waveforms derive from
// ultra::IWaveform and use the reference's own types, so RxPipeline / ModemEngine / the Monte-Carlo tools can hold them
// through a WaveformPtr unchanged; projectultra_amd/host/*.cpp turn HipOfdmDemodulator and HipLDPCDecoder
// into link-time replacements of ultra::OFDMDemodulator and ultra::LDPCDecoder themselves.  Compiled stand-alone the
// header uses the small mirror types below.  Either way it links only against libultra_hip.so; no HIP headers are needed
// by the caller.
#pragma once

#include <algorithm>
#include <atomic>
#include <cmath>
#include <complex>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <span>
#include <stdexcept>
#include <string>
#include <vector>

#include "ultra_hip.h"

#ifdef ULTRA_HIP_WITH_REFERENCE
#include "ultra/types.hpp"
#include "waveform/waveform_interface.hpp"
#endif

namespace ultra_hip {

#ifdef ULTRA_HIP_WITH_REFERENCE
using ultra::Bytes;
using ultra::CodeRate;
using ultra::ModemConfig;
using ultra::Modulation;
using ultra::SampleSpan;
using ultra::Samples;
using ultra::SyncResult;
using ultra::WaveformCapabilities;
#else
using Bytes = std::vector<uint8_t>;
using Samples = std::vector<float>;
using SampleSpan = std::span<const float>;
enum class Modulation : uint8_t { DBPSK = 0, BPSK = 1, DQPSK = 2, QPSK = 3, D8PSK = 4, QAM8 = 5, QAM16 = 6,
                                  QAM32 = 7, QAM64 = 8, QAM256 = 10 };
enum class CodeRate : uint8_t { R1_4, R1_3, R1_2, R2_3, R3_4, R5_6, R7_8 };
enum class CyclicPrefixMode : uint8_t { SHORT = 0, MEDIUM = 1, LONG = 2 };
struct ModemConfig {                       // receive-path fields of ultra::ModemConfig, same defaults
    uint32_t sample_rate = 48000, center_freq = 1500, fft_size = 512, num_carriers = 30;
    CyclicPrefixMode cp_mode = CyclicPrefixMode::MEDIUM;
    uint32_t symbol_guard = 4, pilot_spacing = 2;
    bool use_pilots = true;
    Modulation modulation = Modulation::QPSK;
    CodeRate code_rate = CodeRate::R1_2;
    bool adaptive_eq_enabled = false, adaptive_eq_use_rls = false;   // types.hpp:170-174
    float lms_mu = 0.05f, rls_lambda = 0.99f;
    bool decision_directed = true;
    float sync_threshold = 0.80f;                                    // types.hpp:188
    float tx_cfo_hz = 0.0f;                                          // transmit side only
};
struct SyncResult {
    bool detected = false; int start_sample = -1; float correlation = 0.0f; float cfo_hz = 0.0f;
    float snr_estimate = 0.0f; bool has_training = false;
};
#endif

inline ultra_hip_config to_c_config(const ModemConfig& c, uint32_t entry, uint32_t n_data_symbols,
                                    uint32_t training_symbols, uint32_t max_iterations = 50) {
    ultra_hip_config k{};
    k.sample_rate = c.sample_rate; k.center_freq = c.center_freq; k.fft_size = c.fft_size;
    k.num_carriers = c.num_carriers; k.cp_mode = static_cast<uint32_t>(c.cp_mode);
    k.symbol_guard = c.symbol_guard; k.pilot_spacing = c.pilot_spacing; k.use_pilots = c.use_pilots ? 1u : 0u;
    k.modulation = static_cast<uint32_t>(c.modulation); k.code_rate = static_cast<uint32_t>(c.code_rate);
    k.max_iterations = max_iterations; k.n_data_symbols = n_data_symbols; k.entry = entry;
    k.training_symbols = (entry == ULTRA_ENTRY_PRESYNCED) ? training_symbols : 0;
    k.adaptive_eq_enabled = c.adaptive_eq_enabled ? 1u : 0u; k.adaptive_eq_use_rls = c.adaptive_eq_use_rls ? 1u : 0u;
    k.decision_directed = c.decision_directed ? 1u : 0u; k.lms_mu = c.lms_mu; k.rls_lambda = c.rls_lambda;
    k.sync_threshold = c.sync_threshold;
    return k;
}

namespace detail {
inline void check(int rc, const char* what) {
    if (rc != ULTRA_HIP_OK) throw std::runtime_error(std::string(what) + ": " + ultra_hip_strerror(rc));
}
// The reference's receive interface reports failure as false / empty and never throws ("Errors"); its callers —
// RxPipeline on the audio thread, ModemEngine's decode thread — have no handler, so an exception there is std::terminate.  Every
// entry the reference's callers reach (the IWaveform overrides, the pimpl classes' members) runs through this: a failing C-ABI
// call (device lost, out of memory) is reported on stderr — loudly, every time — and becomes the interface's failure value.
template <class R, class F>
inline R guarded(const char* what, R failure, F&& body) noexcept {
    try { return body(); }
    catch (const std::exception& e) { std::fprintf(stderr, "ultra_hip: %s FAILED: %s\n", what, e.what()); }
    catch (...) { std::fprintf(stderr, "ultra_hip: %s FAILED\n", what); }
    return failure;
}
template <class F>
inline void guarded_void(const char* what, F&& body) noexcept { (void)guarded<int>(what, 0, [&] { body(); return 0; }); }
struct Ctx {                               // RAII owner of an ultra_hip_ctx
    ultra_hip_ctx* p = nullptr;
    Ctx() = default;
    Ctx(const ultra_hip_config& c, int device) { check(ultra_hip_create(&c, device, nullptr, &p), "ultra_hip_create"); }
    Ctx(const Ctx&) = delete;
    Ctx& operator=(const Ctx&) = delete;
    Ctx(Ctx&& o) noexcept : p(o.p) { o.p = nullptr; }
    Ctx& operator=(Ctx&& o) noexcept { if (this != &o) { reset(); p = o.p; o.p = nullptr; } return *this; }
    ~Ctx() { reset(); }
    void reset() { if (p) ultra_hip_destroy(p); p = nullptr; }
};
struct DevBuf {                            // device allocation tied to a context
    ultra_hip_ctx* ctx; void* d = nullptr;
    DevBuf(ultra_hip_ctx* c, size_t bytes) : ctx(c) {
        if (ultra_hip_malloc(ctx, bytes, &d) != ULTRA_HIP_OK) throw std::bad_alloc();
    }
    ~DevBuf() { if (d) ultra_hip_free(ctx, d); }
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
};
// a device buffer that lives with its owner and only ever grows (per-call scratch; the content does not survive growth)
struct GrowBuf {
    std::unique_ptr<DevBuf> b; size_t cap = 0; ultra_hip_ctx* owner = nullptr;
    void* get(ultra_hip_ctx* ctx, size_t bytes) {
        if (!b || owner != ctx || bytes > cap) { b.reset(); cap = std::max<size_t>(bytes, 2 * cap); b = std::make_unique<DevBuf>(ctx, cap); owner = ctx; }
        return b->d;
    }
    void drop() { b.reset(); cap = 0; owner = nullptr; }
};

// A context together with the device scratch its user keeps between calls.  Building a context is table construction, a
// dozen device allocations and a probe launch; the reference's harnesses construct a fresh demodulator and a fresh
// decoder PER TRIAL (tools/test_nvis_mode.cpp:44-46, tools/test_mode_snr.cpp:34-41), so the host classes below recycle
// contexts by configuration instead of building one per object: a recycled slot starts every stream afresh
// (ultra_hip_demod_stream_batch at first_symbol 0, zeroed resume words), so nothing of its previous user shows.
struct Slot {
    ultra_hip_config cfg{}; int device = 0;
    ultra_hip_ctx* ctx = nullptr;
    GrowBuf buf[4];
    // The answer of a latency-bound call (one live stream: a SYNCED process(), a single-codeword decodeSoft) comes back through a
    // pinned block the kernels write directly and ONE posted word the host polls for — no device-to-host copy command, no
    // hipStreamSynchronize (ultra_hip.h: ultra_hip_host_block / ultra_hip_stream_post / ultra_hip_host_wait).  Layout: word 0 the
    // posted sequence number, payload from byte kMailHead on.  The block lives and dies with the context.
    static constexpr size_t kMailBytes = size_t(256) << 10, kMailHead = 256;
    char* mail_h = nullptr; char* mail_d = nullptr; uint32_t mail_seq = 0; bool mail_tried = false;
    bool mail(size_t payload_bytes) {                                   // true: the payload fits the block and the block exists
        if (payload_bytes + kMailHead > kMailBytes) return false;
        if (!mail_tried) {
            mail_tried = true;
            void* h = nullptr; void* d = nullptr;
            if (ultra_hip_host_block(ctx, kMailBytes, &h, &d) == ULTRA_HIP_OK) { mail_h = static_cast<char*>(h); mail_d = static_cast<char*>(d); }
        }
        return mail_h != nullptr;
    }
    char* mailDev() const { return mail_d + kMailHead; }
    const char* mailHost() const { return mail_h + kMailHead; }
    // behind everything issued so far: post the next sequence number, wait for it (20 ms of spinning, then a stream synchronisation)
    void mailWait() {
        ++mail_seq;
        check(ultra_hip_stream_post(ctx, reinterpret_cast<uint32_t*>(mail_d), mail_seq), "stream_post");
        check(ultra_hip_host_wait(ctx, reinterpret_cast<const volatile uint32_t*>(mail_h), mail_seq, 20000), "host_wait");
    }
    Slot(const ultra_hip_config& c, int dev) : cfg(c), device(dev) { check(ultra_hip_create(&c, dev, nullptr, &ctx), "ultra_hip_create"); }
    Slot(const Slot&) = delete;
    Slot& operator=(const Slot&) = delete;
    ~Slot() { for (auto& g : buf) g.drop(); if (ctx) ultra_hip_destroy(ctx); }
};
class SlotPool {
public:
    // never destroyed: objects with static storage may still hand slots back while the process exits, and tearing GPU
    // contexts down from a static destructor races the HIP runtime's own; clear() is the orderly way to give them back
    static SlotPool& instance() { static SlotPool* p = new SlotPool(); return *p; }
    std::unique_ptr<Slot> acquire(const ultra_hip_config& c, int device) {
        {
            std::lock_guard<std::mutex> lock(m_);
            auto it = idle_.find(Key{c, device});
            if (it != idle_.end()) {
                auto s = std::move(it->second); idle_.erase(it);
                // a recycled context starts like a new one: nothing sticky of its previous user (fused deinterleaver, profiling
                // brackets, status bits) — host-side settings, no launch
                (void)ultra_hip_set_deinterleave(s->ctx, 0);
                (void)ultra_hip_set_deinterleave_table(s->ctx, nullptr, 0);
                (void)ultra_hip_profile_enable(s->ctx, 0);
                (void)ultra_hip_clear_status(s->ctx);
                return s;
            }
        }
        return std::make_unique<Slot>(c, device);
    }
    void release(std::unique_ptr<Slot> s) {
        if (!s) return;
        std::lock_guard<std::mutex> lock(m_);
        const Key k{s->cfg, s->device};
        if (idle_.count(k) < kMaxIdlePerKey && idle_.size() < kMaxIdle) idle_.emplace(k, std::move(s));
    }                                                               // (otherwise the slot is destroyed here)
    void clear() { std::lock_guard<std::mutex> lock(m_); idle_.clear(); }
private:
    static constexpr size_t kMaxIdlePerKey = 4, kMaxIdle = 64;
    struct Key {
        ultra_hip_config c; int device;
        bool operator<(const Key& o) const { const int r = std::memcmp(&c, &o.c, sizeof(c)); return r < 0 || (r == 0 && device < o.device); }
    };
    std::mutex m_;
    std::multimap<Key, std::unique_ptr<Slot>> idle_;
};
struct PooledSlot {                        // RAII: back to the pool instead of destroyed
    std::unique_ptr<Slot> s;
    PooledSlot() = default;
    PooledSlot(const ultra_hip_config& c, int device) : s(SlotPool::instance().acquire(c, device)) {}
    PooledSlot(PooledSlot&&) noexcept = default;
    PooledSlot& operator=(PooledSlot&& o) noexcept { if (this != &o) { give_back(); s = std::move(o.s); } return *this; }
    ~PooledSlot() { give_back(); }
    void give_back() { if (s) SlotPool::instance().release(std::move(s)); }
    explicit operator bool() const { return bool(s); }
    ultra_hip_ctx* ctx() const { return s->ctx; }
    void* buf(int i, size_t bytes) { return s->buf[i].get(s->ctx, bytes); }
};
}  // namespace detail

// ---------------------------------------------------------------------------------------------
class HipLDPCDecoder {
public:
    explicit HipLDPCDecoder(CodeRate rate, int device = 0) : rate_(rate), device_(device) {}

    // Decode from soft bits: bit-level multi-block semantics of LDPCDecoder::decodeSoft
    // (src/fec/ldpc_decoder.cpp:283-428): <= 648 LLRs one block (zero padded), more -> every full block
    // contributes exactly k bits, a zero-padded tail block is decoded too, bits packed once at the end.
    Bytes decodeSoft(std::span<const float> llrs) {
        last_success_ = false;                                           // (stays false if the launch below fails)
        if (llrs.empty()) return {};
        const size_t n = 648, nblocks = (llrs.size() + n - 1) / n;
        const ultra_hip_geometry& g = geometry();
        padded_.assign(nblocks * n, 0.0f);
        std::memcpy(padded_.data(), llrs.data(), llrs.size() * sizeof(float));
        bytes_.resize(nblocks * g.decoded_bytes); ok_.resize(nblocks); iters_.resize(nblocks);
        decodeBatch(padded_.data(), nblocks, bytes_.data(), iters_.data(), ok_.data());
        const bool has_tail = (llrs.size() % n) != 0 && nblocks > 1;
        bool all = true; for (auto v : ok_) all = all && v;
        last_success_ = (nblocks == 1 || has_tail) ? ok_.back() != 0 : all;
        last_iters_ = iters_.back();
        if (nblocks == 1) return Bytes(bytes_.begin(), bytes_.end());
        std::vector<uint8_t> bits;
        bits.reserve(nblocks * g.ldpc_k);
        for (size_t b = 0; b < nblocks; ++b)
            for (uint32_t j = 0; j < g.ldpc_k; ++j)
                bits.push_back((bytes_[b * g.decoded_bytes + j / 8] >> (7 - j % 8)) & 1);
        Bytes out((bits.size() + 7) / 8, 0);
        for (size_t i = 0; i < bits.size(); ++i) if (bits[i]) out[i / 8] |= uint8_t(1u << (7 - i % 8));
        return out;
    }
    // RxPipeline::setInterleaverConfig + deinterleaveCodewords (src/gui/modem/rx_pipeline.cpp:24-31,475-491):
    // every codeword goes through ChannelInterleaver(bits_per_symbol, 648)::deinterleave before it is
    // decoded — on the GPU, fused into the decoder's LLR load.  0 switches it off.
    void setDeinterleave(size_t bits_per_symbol) {
        deinterleave_ = (uint32_t)bits_per_symbol;
        if (slot_) detail::check(ultra_hip_set_deinterleave(slot_.ctx(), deinterleave_), "ultra_hip_set_deinterleave");
    }
    Bytes decode(std::span<const uint8_t> coded) {                     // ldpc_decoder.cpp:267-281
        std::vector<float> llrs; llrs.reserve(coded.size() * 8);
        for (uint8_t byte : coded) for (int b = 7; b >= 0; --b) llrs.push_back(((byte >> b) & 1) ? -6.0f : 6.0f);
        return decodeSoft(llrs);
    }
    // n_cw independent codewords, host buffers: llr [n_cw][648] -> bytes [n_cw][ceil(k/8)], iters, ok.
    // One upload, one launch, ONE download: the three outputs share a device block behind the soft bits.
    void decodeBatch(const float* llr, size_t n_cw, uint8_t* bytes, int32_t* iters, uint8_t* ok) {
        if (n_cw == 0) return;
        const ultra_hip_geometry& g = geometry();
        ensure();
        const size_t llr_bytes = n_cw * 648 * sizeof(float), it_bytes = n_cw * sizeof(int32_t), by_bytes = n_cw * g.decoded_bytes;
        const size_t out_bytes = it_bytes + by_bytes + n_cw;
        if (n_cw <= kMailCodewords && slot_.s->mail(out_bytes)) {
            // a handful of codewords (RxPipeline decodes one at a time): the kernel reads the soft bits from the pinned staging
            // ring and writes its three outputs into the pinned result block — two launches, no copy command, one polled word
            void* d_in = nullptr;
            detail::check(ultra_hip_stage_input(slot_.ctx(), llr, llr_bytes, &d_in), "stage_input");
            char* d_out = slot_.s->mailDev();
            detail::check(ultra_hip_ldpc_decode_batch(slot_.ctx(), static_cast<const float*>(d_in), n_cw, reinterpret_cast<uint8_t*>(d_out + it_bytes),
                                                      reinterpret_cast<int32_t*>(d_out), reinterpret_cast<uint8_t*>(d_out + it_bytes + by_bytes), nullptr),
                          "ldpc_decode_batch");
            slot_.s->mailWait();
            const char* h = slot_.s->mailHost();
            std::memcpy(iters, h, it_bytes);
            std::memcpy(bytes, h + it_bytes, by_bytes);
            std::memcpy(ok, h + it_bytes + by_bytes, n_cw);
            return;
        }
        char* d = static_cast<char*>(slot_.buf(0, llr_bytes + out_bytes));
        char* d_out = d + llr_bytes;
        detail::check(ultra_hip_memcpy_h2d_async(slot_.ctx(), d, llr, llr_bytes), "h2d");
        detail::check(ultra_hip_ldpc_decode_batch(slot_.ctx(), reinterpret_cast<const float*>(d), n_cw,
                                                  reinterpret_cast<uint8_t*>(d_out + it_bytes), reinterpret_cast<int32_t*>(d_out),
                                                  reinterpret_cast<uint8_t*>(d_out + it_bytes + by_bytes), nullptr), "ldpc_decode_batch");
        stage_.resize(out_bytes);
        detail::check(ultra_hip_memcpy_d2h(slot_.ctx(), stage_.data(), d_out, out_bytes), "d2h");
        std::memcpy(iters, stage_.data(), it_bytes);
        std::memcpy(bytes, stage_.data() + it_bytes, by_bytes);
        std::memcpy(ok, stage_.data() + it_bytes + by_bytes, n_cw);
    }
    bool lastDecodeSuccess() const { return last_success_; }
    int lastIterations() const { return last_iters_; }
    void setRate(CodeRate rate) { rate_ = rate; slot_ = detail::PooledSlot(); have_geo_ = false; }
    CodeRate getRate() const { return rate_; }
    void setMaxIterations(int max_iter) { max_iter_ = max_iter; slot_ = detail::PooledSlot(); have_geo_ = false; }

private:
    ultra_hip_config cConfig() const {
        ModemConfig c; c.code_rate = rate_;
        return to_c_config(c, ULTRA_ENTRY_SYNCED, 44, 0, static_cast<uint32_t>(max_iter_ < 0 ? 0 : max_iter_));
    }
    const ultra_hip_geometry& geometry() {                               // code parameters: host arithmetic, no GPU needed
        if (!have_geo_) { const ultra_hip_config k = cConfig(); detail::check(ultra_hip_geometry_for(&k, &geo_), "geometry"); have_geo_ = true; }
        return geo_;
    }
    void ensure() {                                                      // the context comes from the pool on first use
        if (slot_) return;
        slot_ = detail::PooledSlot(cConfig(), device_);
        detail::check(ultra_hip_set_deinterleave(slot_.ctx(), deinterleave_), "ultra_hip_set_deinterleave");   // a recycled context keeps its last user's
    }
    static constexpr size_t kMailCodewords = 16;                        // 41 KB of soft bits: well inside the staging ring's per-call share
    CodeRate rate_; int device_; int max_iter_ = 50; bool last_success_ = false; int last_iters_ = 0;
    uint32_t deinterleave_ = 0;
    bool have_geo_ = false; ultra_hip_geometry geo_{};
    detail::PooledSlot slot_;
    std::vector<float> padded_; std::vector<uint8_t> bytes_, ok_, stage_; std::vector<int32_t> iters_;
};

// ---------------------------------------------------------------------------------------------
// ultra::OFDMDemodulator (include/ultra/ofdm.hpp:58-127; src/ofdm/demodulator.cpp:461-1017) on the device, as a LIVE stream —
// the state machine every caller of the reference drives, one object per stream, one process() call at a time:
//   SEARCHING  every process() call is ONE launch of the chunk-fed Schmidl-Cox search that continues from the state the
//              previous call left (ultra_hip_acquire_stream_batch: start of rx_buffer, samples fed, the energy gate's
//              noise floor) — only the new samples are uploaded, nothing is searched twice;
//   SYNCED     whole symbols are demodulated as they arrive, the tracker continuing on the device
//              (ultra_hip_demod_stream_batch); no frame length is needed up front; the three ways out of SYNCED are
//              the reference's: more than MAX_SYMBOLS_BEFORE_TIMEOUT symbols (:683-691), more than
//              MAX_IDLE_CALLS_BEFORE_RESET calls without a new soft bit (:704-716), an empty call with nothing left
//              to demodulate and less than a codeword buffered ("frame complete", :720-731).
//              A new preamble arriving while SYNCED (:605-657: symbols demodulated, two calls or more without a soft
//              bit, six preamble symbols buffered) abandons the old frame: ultra_hip_resync_stream_batch scans the
//              buffer's first two symbols, and on a hit the demodulation restarts at the new data start.
//   processPresynced  an external synchroniser (the chirp) provides timing and CFO: training symbols + data symbols in
//              one launch chain on a context of the PRESYNCED entry; the object is SYNCED afterwards and process()
//              continues the frame there.
// What a frame inherits from the one before it on the same object is the reference's too: a
// SEARCHING -> SYNCED transition without reset() in between carries the tracker (ULTRA_STREAM_START_SYNC), and
// timing_offset_samples survives reset(), processPresynced's reset block and the mid-frame preamble
// (ULTRA_STREAM_START_TIMING).  One gap: an object that mixes both entries WITHOUT reset() starts the Schmidl-Cox frame
// that follows a presynced one from a fresh tracker (the two entries' records live in two contexts); every caller in the
// reference resets between them (modem_rx_decode.cpp:673,1240; rx_pipeline.cpp:172-287).
// Device buffers come from a pool and go back to it (detail::SlotPool): constructing one per trial is cheap.
// projectultra_amd/host/hip_ofdm_demodulator.cpp makes this class the Impl of ultra::OFDMDemodulator itself.
struct HipChannelQuality { float snr_db = 0.0f, doppler_hz = 0.0f, delay_spread_ms = 0.0f, ber_estimate = 0.0f; };   // ultra::ChannelQuality
class HipOfdmDemodulator {
public:
    static constexpr int kMaxSymbolsBeforeTimeout = 250, kMaxIdleCallsBeforeReset = 10;   // demodulator_constants.hpp:37-38
    static constexpr size_t kLdpcBlock = 648;                                             // LDPC_BLOCK_SIZE (:14)
    static constexpr size_t kMaxConstellationSymbols = 500;                               // MAX_CONSTELLATION_SYMBOLS (:122)
    explicit HipOfdmDemodulator(const ModemConfig& config, int device = 0) : config_(config), device_(device) {
        const ultra_hip_config k = to_c_config(config_, ULTRA_ENTRY_SYNCED, kMaxSymbolsBeforeTimeout + 1, 0);
        detail::check(ultra_hip_geometry_for(&k, &geo_), "ultra_hip_geometry_for");       // host arithmetic: no GPU yet
    }
    const ModemConfig& config() const { return config_; }
    const ultra_hip_geometry& geometry() const { return geo_; }

    // OFDMDemodulator::process (demodulator.cpp:461-741)
    bool process(SampleSpan samples) {
        ensure();
        appendSamples(samples);
        if (!synced_) {
            uint32_t* w = small();
            detail::check(ultra_hip_acquire_stream_batch(slot_.ctx(), rxDev(), rx_cap_, d_origin_, fed_, 1,
                                                         w, w + 4, w + 5, reinterpret_cast<float*>(w + 6), w + 7), "acquire_stream");
            uint32_t h[8];
            detail::check(ultra_hip_memcpy_d2h(slot_.ctx(), h, w, sizeof(h)), "d2h");
            noise_floor_bits_ = h[2];
            if (h[4]) {                                                  // SEARCHING -> SYNCED (:533-591)
                std::memcpy(&coarse_cfo_, &h[6], sizeof(float));
                freq_offset_hz_ = coarse_cfo_; freq_correction_phase_ = 0.0f; last_sync_offset_ = h[7];
                consumeTo(clampIndex(int64_t(h[5]) + manual_timing_offset_));   // consume = refined_lts + 2 preamble symbols + manual offset (:572)
                synced_ = true; synced_symbols_ = 0; pending_cfo_ = false; live_ps_ = false;
                timing_ = 0.0f;                                          // timing_offset_samples = 0 (:588)
                // a used demodulator carries its tracker into the new frame (:533-591) — from the SYNCED context's own records, or,
                // when the frame before came through processPresynced, from the PRESYNCED context's (one Impl, two contexts here)
                if (!carry_ && ps_carry_ && ps_slot_) {
                    detail::check(ultra_hip_stream_adopt(slot_.ctx(), ps_slot_.ctx(), 1), "stream_adopt");
                    carry_ = true;
                }
                ps_carry_ = false;
                start_mode_ = carry_ ? ULTRA_STREAM_START_SYNC : ULTRA_STREAM_START_FRESH;
            } else if (h[0] > origin_) {
                consumeTo(h[0]);                                         // what the search trimmed off the buffer
            }
        }
        if (!synced_) return false;
        // a new preamble while SYNCED (:605-657): symbols were demodulated, the last two calls or more brought no soft bit,
        // and six preamble symbols are buffered — the scan runs on the device, the two counters live here
        const uint32_t preamble_total = 6u * (static_cast<uint32_t>(config_.fft_size) + geo_.cp_len);
        if (synced_symbols_ > 0 && idle_calls_ >= 2 && fed_ - origin_ >= preamble_total) {
            restartSearch();                                             // word 0 of the record: where rx_buffer starts
            uint32_t* w = small();
            detail::check(ultra_hip_resync_stream_batch(slot_.ctx(), rxDev(), rx_cap_, d_origin_, fed_, 1,
                                                        w, w + 4, w + 5, reinterpret_cast<float*>(w + 6), w + 7), "resync_stream");
            uint32_t h[8];
            detail::check(ultra_hip_memcpy_d2h(slot_.ctx(), h, w, sizeof(h)), "d2h");
            if (h[4]) {                                                  // the old frame is abandoned: the constructor's tracker, but for
                std::memcpy(&coarse_cfo_, &h[6], sizeof(float));         // timing_offset_samples, which :626-655 does not touch
                freq_offset_hz_ = coarse_cfo_; freq_correction_phase_ = 0.0f;
                consumeTo(h[5]);
                demod_soft_.clear();
                synced_symbols_ = 0; idle_calls_ = 0; pending_cfo_ = false; live_ps_ = false;
                state_[ULTRA_HIP_STATE_SNR_LINEAR] = 1.0f;
                start_mode_ = (timing_ != 0.0f) ? ULTRA_STREAM_START_TIMING : ULTRA_STREAM_START_FRESH;
            }
        }
        const uint32_t sym = symbolSamples();
        uint32_t n_new = (fed_ - origin_) / sym;
        const uint32_t room = uint32_t(kMaxSymbolsBeforeTimeout + 1) - std::min<uint32_t>(synced_symbols_, kMaxSymbolsBeforeTimeout + 1);
        if (n_new > room) n_new = room;
        const size_t soft_before = demod_soft_.size();
        if (n_new > 0) {
            demodulate(n_new);
            if (synced_symbols_ > uint32_t(kMaxSymbolsBeforeTimeout)) {   // sync timeout (:683-691)
                toSearching();
                return demod_soft_.size() >= kLdpcBlock;
            }
        }
        if (demod_soft_.size() == soft_before) {                         // idle calls (:704-716)
            if (++idle_calls_ > kMaxIdleCallsBeforeReset) { toSearching(); return demod_soft_.size() >= kLdpcBlock; }
        } else {
            idle_calls_ = 0;
        }
        const bool has_codeword = demod_soft_.size() >= kLdpcBlock;
        if (!has_codeword && synced_symbols_ > 0 && samples.empty() && n_new == 0) {   // frame complete (:720-731)
            toSearching();
            demod_soft_.clear();
        }
        return has_codeword;
    }

    // OFDMDemodulator::processPresynced (demodulator.cpp:854-985): samples start at the first of `training_symbols` LTS symbols
    bool processPresynced(SampleSpan samples, int training_symbols = 2) {
        const uint32_t sym = symbolSamples();
        if (samples.size() < sym) return false;
        ensure();
        const uint32_t n_train = training_symbols > 0 ? uint32_t(training_symbols) : 0u;
        if (samples.size() / sym < n_train) {
            // fewer whole symbols than training symbols: the reference reads past the span here (:935-941) — undefined there; here
            // the call is refused before it touches the object's state, like the too-short call above
            return false;
        }
        if (!ps_slot_ || ps_train_ != n_train) {
            ps_slot_ = detail::PooledSlot(to_c_config(config_, ULTRA_ENTRY_PRESYNCED, kMaxSymbolsBeforeTimeout + 1, n_train), device_);
            ps_train_ = n_train;
        }
        // the reset block (:868-905): soft bits, buffer, counters, the tracker (the stream starts at symbol 0) — the preset
        // frequency offset and phase stay, and so does timing_offset_samples
        demod_soft_.clear();
        rx_.clear(); origin_ = fed_ = d_origin_ = 0; ++epoch_;         // sample indices start over (ultra_hip.h: the resume record's epoch)
        synced_symbols_ = 0; idle_calls_ = 0; pending_cfo_ = false;
        state_[ULTRA_HIP_STATE_SNR_LINEAR] = 1.0f;
        synced_ = true; live_ps_ = true; carry_ = false; ps_carry_ = true;   // this frame's tracker lives in the PRESYNCED context
        appendSamples(samples);
        uint32_t n_sym = static_cast<uint32_t>(samples.size() / sym);
        if (n_sym == 0) return false;
        if (n_sym > n_train + uint32_t(kMaxSymbolsBeforeTimeout + 1)) {
            std::fprintf(stderr, "ultra_hip: processPresynced: %u data symbols in one call, %d demodulated (context capacity)\n",
                         n_sym - n_train, kMaxSymbolsBeforeTimeout + 1);
            n_sym = n_train + uint32_t(kMaxSymbolsBeforeTimeout + 1);
        }
        // :918-928 — an offset set from outside is trusted; one that was never set comes from the training symbols (NaN is the
        // C-ABI's "never set"); anything else on the object is a preset
        float cp[3] = {freq_offset_hz_, freq_correction_phase_, timing_};
        if (!chirp_cfo_estimated_ && n_train >= 2 && std::fabs(freq_offset_hz_) < 0.1f) { cp[0] = std::nanf(""); cp[1] = 0.0f; }
        float* d_in = reinterpret_cast<float*>(small() + 16);
        detail::check(ultra_hip_memcpy_h2d_async(slot_.ctx(), d_in, cp, sizeof(cp)), "h2d");
        if (timing_ != 0.0f) detail::check(ultra_hip_demod_stream_start(ps_slot_.ctx(), ULTRA_STREAM_START_TIMING, d_in + 2), "stream_start");
        const uint32_t n_data = n_sym - n_train;
        const size_t n_llr_ps = size_t(n_data) * geo_.llrs_per_symbol;
        float* d_out = outDev(n_llr_ps, n_data);
        detail::check(ultra_hip_demod_stream_batch_eq(ps_slot_.ctx(), rxDev(), size_t(n_sym) * sym, d_in, d_in + 1, 1, 0, n_sym,
                                                      d_out + ULTRA_HIP_STATE_FLOATS, d_out, d_out + eqOffset(n_llr_ps)), "demod_stream");
        if (n_data > 0) fetch(n_llr_ps, n_data);
        consumeTo(origin_ + n_sym * sym);
        synced_symbols_ = n_sym;
        return demod_soft_.size() >= kLdpcBlock;
    }

    // 648 at a time (demodulator.cpp:766-791)
    std::vector<float> getSoftBits() {
        if (demod_soft_.size() <= kLdpcBlock) { std::vector<float> out = std::move(demod_soft_); demod_soft_.clear(); return out; }
        std::vector<float> out(demod_soft_.begin(), demod_soft_.begin() + kLdpcBlock);
        demod_soft_.erase(demod_soft_.begin(), demod_soft_.begin() + kLdpcBlock);
        return out;
    }
    // hard decisions of everything buffered, eight to a byte, a trailing partial byte dropped (:745-764; bit = llr > 0 there)
    Bytes getData() {
        Bytes data; uint8_t byte = 0; int n = 0;
        for (float llr : demod_soft_) {
            byte = uint8_t((byte << 1) | (llr > 0 ? 1 : 0));
            if (++n == 8) { data.push_back(byte); byte = 0; n = 0; }
        }
        demod_soft_.clear();
        return data;
    }
    HipChannelQuality getChannelQuality() const { std::lock_guard<std::mutex> l(gui_m_); return quality_; }   // Impl::updateQuality (:437-451) after every symbol
    float getEstimatedSNR() const { return 10.0f * std::log10(state_[ULTRA_HIP_STATE_SNR_LINEAR]); }   // :797-799
    float getFrequencyOffset() const { return freq_offset_hz_; }
    float coarseCFO() const { return coarse_cfo_; }                     // Impl::estimateCoarseCFO at the last sync
    void setFrequencyOffset(float cfo_hz) { setFrequencyOffsetWithPhase(cfo_hz, 0.0f); }      // :805-814
    void setFrequencyOffsetWithPhase(float cfo_hz, float initial_phase_rad) {                  // :816-825
        freq_offset_hz_ = cfo_hz; freq_correction_phase_ = initial_phase_rad; chirp_cfo_estimated_ = true;
        if (synced_ && synced_symbols_ > 0)                              // mid-frame: the tracker on the device takes it from the next symbol on
            detail::check(ultra_hip_demod_stream_set_cfo_phase(liveCtx(), 0, cfo_hz, initial_phase_rad), "stream_set_cfo");
        else pending_cfo_ = true;                                       // SYNCED before the first symbol: symbol 0 starts from it (a sync found later overwrites it: :535-537)
    }
    // the GUI's scatter plot (:827-830): the newest MAX_CONSTELLATION_SYMBOLS equalized data carriers; never cleared, as in the reference
    std::vector<std::complex<float>> getConstellationSymbols() const { std::lock_guard<std::mutex> l(gui_m_); return constellation_; }
    bool isSynced() const { return synced_.load(); }
    bool hasPendingData() const {                                       // :836-844
        return synced_ && (!demod_soft_.empty() || fed_ - origin_ >= symbolSamples());
    }
    size_t getLastSyncOffset() const { return last_sync_offset_; }
    void setTimingOffset(int offset) { manual_timing_offset_ = offset; }
    void reset() {                                                      // :987-1017; the energy gate's noise floor, timing_offset_samples,
        synced_ = false; synced_symbols_ = 0; idle_calls_ = 0;          // last_sync_offset and the quality report are not among what it clears
        rx_.clear(); origin_ = fed_; d_origin_ = fed_; demod_soft_.clear();
        freq_offset_hz_ = 0.0f; freq_correction_phase_ = 0.0f; chirp_cfo_estimated_ = false; pending_cfo_ = false;
        state_[ULTRA_HIP_STATE_SNR_LINEAR] = 1.0f; state_[ULTRA_HIP_STATE_FREQ_OFFSET_HZ] = 0.0f;
        carry_ = false; ps_carry_ = false; live_ps_ = false; start_mode_ = ULTRA_STREAM_START_FRESH;
        if (slot_) restartSearch();
    }
    uint32_t symbolSamples() const { return geo_.symbol_samples; }

private:
    void ensure() {
        if (slot_) return;
        slot_ = detail::PooledSlot(to_c_config(config_, ULTRA_ENTRY_SYNCED, kMaxSymbolsBeforeTimeout + 1, 0), device_);
        (void)slot_.buf(0, 32 * sizeof(uint32_t));   // resume[4], found, data_start, cfo, sync offset | state[8] | cfo, phase, timing in
        rx_cap_ = std::max<size_t>(slot_.s->buf[1].cap / sizeof(float), size_t(1) << 16);
        (void)slot_.buf(1, rx_cap_ * sizeof(float));
        restartSearch();
    }
    uint32_t* small() { return static_cast<uint32_t*>(slot_.buf(0, 32 * sizeof(uint32_t))); }
    const float* rxDev() { return static_cast<const float*>(slot_.buf(1, rx_cap_ * sizeof(float))); }
    // [tracker state (8 floats) | soft bits of this call]: one download brings both
    // ... and behind them (8-byte aligned) the equalized data carriers of the call's symbols, ULTRA_HIP_MAX_CARRIERS pairs each
    static size_t eqOffset(size_t n_llr) { return (ULTRA_HIP_STATE_FLOATS + n_llr + 1) & ~size_t(1); }
    // ... in the context's pinned result block when the call's answer fits it (every live call does: a symbol or a few), in
    // device scratch otherwise (a whole frame handed to processPresynced at once)
    static size_t outFloats(size_t n_llr, size_t n_sym) { return eqOffset(n_llr) + n_sym * 2 * ULTRA_HIP_MAX_CARRIERS; }
    float* outDev(size_t n_llr, size_t n_sym = 0) {
        out_in_mail_ = slot_.s->mail(outFloats(n_llr, n_sym) * sizeof(float));
        if (out_in_mail_) return reinterpret_cast<float*>(slot_.s->mailDev());
        return static_cast<float*>(slot_.buf(2, (eqOffset(std::max<size_t>(n_llr, 4096)) + std::max<size_t>(n_sym, 8) * 2 * ULTRA_HIP_MAX_CARRIERS) * sizeof(float)));
    }
    ultra_hip_ctx* liveCtx() { return live_ps_ ? ps_slot_.ctx() : slot_.ctx(); }
    uint32_t clampIndex(int64_t i) const { return uint32_t(std::min<int64_t>(std::max<int64_t>(i, origin_), fed_)); }
    // the search restarts on whatever is still buffered: rx_buffer = [origin_, fed_)
    void restartSearch() {
        const uint32_t r[4] = {origin_, fed_, noise_floor_bits_, epoch_};
        detail::check(ultra_hip_memcpy_h2d_async(slot_.ctx(), small(), r, sizeof(r)), "h2d");
    }
    // sample indices are 32-bit and absolute: long before they run out (6 h of audio) the origin moves to rx_buffer's start
    void rebase() {
        uint32_t r[4];
        detail::check(ultra_hip_memcpy_d2h(slot_.ctx(), r, small(), sizeof(r)), "d2h");
        const uint32_t shift = origin_;
        r[0] -= shift; r[1] -= shift; r[3] = ++epoch_;                  // the indices move: the library's metric cache of the stream starts over
        detail::check(ultra_hip_memcpy_h2d(slot_.ctx(), small(), r, sizeof(r)), "h2d");
        fed_ -= shift; origin_ = 0; d_origin_ = 0;
        if (!rx_.empty()) detail::check(ultra_hip_memcpy_h2d(slot_.ctx(), const_cast<float*>(rxDev()), rx_.data(), rx_.size() * sizeof(float)), "h2d");
    }
    void appendSamples(SampleSpan samples) {
        if (!synced_ && origin_ > 0 && fed_ > (1u << 29)) rebase();
        rx_.insert(rx_.end(), samples.begin(), samples.end());
        const size_t need = size_t(fed_ - d_origin_) + samples.size();
        if (need > rx_cap_ || (d_origin_ < origin_ && size_t(origin_ - d_origin_) > rx_cap_ / 2)) {
            // grow, or drop the consumed front: the device window restarts at origin_ from the host's copy
            const size_t live = rx_.size();
            if (live > rx_cap_) rx_cap_ = std::max<size_t>(2 * live, size_t(1) << 16);
            d_origin_ = origin_;
            if (live) detail::check(ultra_hip_memcpy_h2d(slot_.ctx(), const_cast<float*>(rxDev()), rx_.data(), live * sizeof(float)), "h2d");
        } else if (!samples.empty()) {
            detail::check(ultra_hip_memcpy_h2d_async(slot_.ctx(), const_cast<float*>(rxDev()) + (fed_ - d_origin_), samples.data(),
                                                     samples.size() * sizeof(float)), "h2d");
        }
        fed_ += static_cast<uint32_t>(samples.size());
    }
    void consumeTo(uint32_t abs_index) {                                 // rx_buffer.erase(begin, begin + n)
        rx_.erase(rx_.begin(), rx_.begin() + (abs_index - origin_));
        origin_ = abs_index;
    }
    void toSearching() { synced_ = false; synced_symbols_ = 0; idle_calls_ = 0; live_ps_ = false; restartSearch(); }
    // the next n_new whole symbols of rx_buffer through the tracker of the frame in flight
    void demodulate(uint32_t n_new) {
        const uint32_t sym = symbolSamples();
        const size_t n_llr = size_t(n_new) * geo_.llrs_per_symbol;
        float* d_out = outDev(n_llr, n_new);
        float* d_in = reinterpret_cast<float*>(small() + 16);
        ultra_hip_ctx* ctx = liveCtx();
        if (synced_symbols_ == 0) {                                      // symbol 0 of a Schmidl-Cox frame
            const float cp[3] = {pending_cfo_ ? freq_offset_hz_ : coarse_cfo_, pending_cfo_ ? freq_correction_phase_ : 0.0f, timing_};
            detail::check(ultra_hip_memcpy_h2d_async(ctx, d_in, cp, sizeof(cp)), "h2d");
            if (start_mode_ != ULTRA_STREAM_START_FRESH) detail::check(ultra_hip_demod_stream_start(ctx, start_mode_, d_in + 2), "stream_start");
            start_mode_ = ULTRA_STREAM_START_FRESH;
        }
        detail::check(ultra_hip_demod_stream_batch_eq(ctx, rxDev() + (origin_ - d_origin_), size_t(n_new) * sym, d_in, d_in + 1, 1,
                                                      synced_symbols_, n_new, d_out + ULTRA_HIP_STATE_FLOATS, d_out, d_out + eqOffset(n_llr)),
                      "demod_stream");
        fetch(n_llr, n_new);
        consumeTo(origin_ + n_new * sym);
        synced_symbols_ += n_new; pending_cfo_ = false;
        if (!live_ps_) carry_ = true;                                   // the SYNCED context's records now hold this object's tracker
    }
    // [state | soft bits] of the call just issued; the host's copies of what the reference reads back from Impl
    void fetch(size_t n_llr, size_t n_sym) {
        stage_.resize(outFloats(n_llr, n_sym));
        if (out_in_mail_) {                                              // the kernels wrote into the pinned block: wait for the posted word
            slot_.s->mailWait();
            std::memcpy(stage_.data(), slot_.s->mailHost(), stage_.size() * sizeof(float));
        } else {
            detail::check(ultra_hip_memcpy_d2h(slot_.ctx(), stage_.data(), outDev(n_llr, n_sym), stage_.size() * sizeof(float)), "d2h");
        }
        std::memcpy(state_, stage_.data(), sizeof(state_));
        demod_soft_.insert(demod_soft_.end(), stage_.begin() + ULTRA_HIP_STATE_FLOATS, stage_.begin() + ULTRA_HIP_STATE_FLOATS + n_llr);
        // demodulateSymbol (demodulator.cpp:199-208): every data symbol appends its equalized carriers; the newest 500 stay
        std::lock_guard<std::mutex> l(gui_m_);
        for (size_t s = 0; s < n_sym; ++s) {
            const float* row = stage_.data() + eqOffset(n_llr) + s * 2 * ULTRA_HIP_MAX_CARRIERS;
            for (uint32_t i = 0; i < geo_.n_data_carriers; ++i) constellation_.emplace_back(row[2 * i], row[2 * i + 1]);
            if (constellation_.size() > kMaxConstellationSymbols)
                constellation_.erase(constellation_.begin(), constellation_.begin() + (constellation_.size() - kMaxConstellationSymbols));
        }
        freq_offset_hz_ = state_[ULTRA_HIP_STATE_FREQ_OFFSET_HZ];
        freq_correction_phase_ = state_[ULTRA_HIP_STATE_CFO_PHASE];
        timing_ = state_[ULTRA_HIP_STATE_TIMING_OFFSET];
        quality_.snr_db = 10.0f * std::log10(state_[ULTRA_HIP_STATE_SNR_LINEAR]);
        quality_.doppler_hz = 0.0f; quality_.delay_spread_ms = 0.0f;
        quality_.ber_estimate = quality_.snr_db > 15 ? 1e-6f : quality_.snr_db > 10 ? 1e-5f : quality_.snr_db > 5 ? 1e-3f : 1e-1f;
    }

    ModemConfig config_;
    int device_;
    ultra_hip_geometry geo_{};
    detail::PooledSlot slot_, ps_slot_;      // SYNCED entry (+ the device buffers), PRESYNCED entry (context only)
    uint32_t ps_train_ = 0;
    size_t rx_cap_ = 0;
    std::vector<float> rx_;                  // rx_buffer = samples [origin_, fed_)
    uint32_t origin_ = 0, fed_ = 0, d_origin_ = 0;   // the device window holds samples from d_origin_ <= origin_ on
    uint32_t noise_floor_bits_ = 0, last_sync_offset_ = 0, synced_symbols_ = 0, epoch_ = 0;
    int idle_calls_ = 0, manual_timing_offset_ = 0, start_mode_ = ULTRA_STREAM_START_FRESH;
    // read by the GUI thread while the audio thread runs process() (ModemEngine::isSynced / getChannelQuality /
    // getConstellationSymbols, modem_engine.cpp:812-827); the reference keeps its sync state atomic and its constellation buffer
    // behind a mutex (demodulator_impl.hpp:28-32,54) — so does this
    std::atomic<bool> synced_{false};
    mutable std::mutex gui_m_;               // constellation_, quality_
    bool pending_cfo_ = false, chirp_cfo_estimated_ = false;
    bool carry_ = false;                     // the SYNCED context holds a frame's tracker of THIS object (no reset() since)
    bool ps_carry_ = false;                  // ... the PRESYNCED context does: the last frame came through processPresynced
    bool live_ps_ = false;                   // the frame in flight lives in the PRESYNCED context
    bool out_in_mail_ = false;               // the call in flight writes its answer into the slot's pinned result block
    float coarse_cfo_ = 0.0f, freq_offset_hz_ = 0.0f, freq_correction_phase_ = 0.0f, timing_ = 0.0f;
    std::vector<float> demod_soft_, stage_;
    std::vector<std::complex<float>> constellation_;
    HipChannelQuality quality_{};
    float state_[ULTRA_HIP_STATE_FLOATS] = {0, 0, 1, 0, 0, 0, 0, 0};
};

// ---------------------------------------------------------------------------------------------
// Chirp flavour: ultra::OFDMChirpWaveform (src/waveform/ofdm_chirp_waveform.cpp) — what WaveformFactory::create(OFDM_CHIRP)
// returns (src/waveform/waveform_factory.cpp:20-21,56-57).  detectSync is the dual-chirp detection on the device (scope row f4),
// process() the presynced entry of HipOfdmDemodulator with the CFO phase accumulated up to the training symbols.
// Inside the reference tree (-DULTRA_HIP_WITH_REFERENCE) it derives from ultra::IWaveform, and the transmit half — not on the
// hot path — is the reference's own: a held ultra::OFDMChirpWaveform generates the preamble and modulates (built on first use).
#ifdef ULTRA_HIP_WITH_REFERENCE
}  // namespace ultra_hip
#include "ultra/ofdm.hpp"
#include "waveform/ofdm_chirp_waveform.hpp"
namespace ultra_hip {
#endif
class HipOfdmWaveform
#ifdef ULTRA_HIP_WITH_REFERENCE
    : public ultra::IWaveform
#endif
{
public:
    // OFDMChirpWaveform::OFDMChirpWaveform(config) (src/waveform/ofdm_chirp_waveform.cpp:20-31): the chirp mode is
    // differential and pilot-free whatever the configuration says
    explicit HipOfdmWaveform(const ModemConfig& config = ModemConfig(), int device = 0)
        : config_(config), device_(device) {
        if (!isDifferential(config_.modulation)) config_.modulation = Modulation::DQPSK;
        config_.use_pilots = false;
        initComponents();
    }

    std::string getName() const { return "OFDM_HIP"; }
    void configure(Modulation mod, CodeRate rate) {                     // OFDMChirpWaveform::configure (:67-84)
        if (!isDifferential(mod)) mod = Modulation::DQPSK;
        config_.modulation = mod; config_.code_rate = rate;
        config_.use_pilots = false;
        initComponents();
    }
    void setFrequencyOffset(float cfo_hz) {                            // :86-91
        cfo_hz_ = cfo_hz;
        detail::guarded_void("HipOfdmWaveform::setFrequencyOffset", [&] { demod_->setFrequencyOffset(cfo_hz); });
    }
    Modulation getModulation() const { return config_.modulation; }
    CodeRate getCodeRate() const { return config_.code_rate; }
    float getFrequencyOffset() const { return cfo_hz_; }

    // OFDMChirpWaveform::detectSync (src/waveform/ofdm_chirp_waveform.cpp:129-172): dual-chirp detection on the
    // device (scope row f4, ultra_hip_chirp_sync_batch); start_sample = where the two training symbols start.
    // No exception leaves an IWaveform: a failing C-ABI call is reported on stderr and reads as "not detected" / "not ready".
    bool detectSync(SampleSpan samples, SyncResult& result, float threshold = 0.15f) {
        return detail::guarded<bool>("HipOfdmWaveform::detectSync", false, [&] { return detectSyncImpl(samples, result, threshold); });
    }
    bool process(SampleSpan samples) {
        return detail::guarded<bool>("HipOfdmWaveform::process", false, [&] { return processImpl(samples); });
    }
private:
    bool detectSyncImpl(SampleSpan samples, SyncResult& result, float threshold) {
        if (!sync_slot_) sync_slot_ = detail::PooledSlot(to_c_config(config_, ULTRA_ENTRY_PRESYNCED, 1, 2), device_);
        char* d = static_cast<char*>(sync_slot_.buf(0, 4 * sizeof(uint32_t) + std::max<size_t>(samples.size(), 1) * sizeof(float)));
        uint32_t* o = reinterpret_cast<uint32_t*>(d);
        float* d_a = reinterpret_cast<float*>(d + 4 * sizeof(uint32_t));
        detail::check(ultra_hip_memcpy_h2d(sync_slot_.ctx(), d_a, samples.data(), samples.size() * sizeof(float)), "h2d");
        detail::check(ultra_hip_chirp_sync_batch(sync_slot_.ctx(), d_a, samples.size(),
                                                 static_cast<uint32_t>(samples.size()), 1, threshold, o,
                                                 reinterpret_cast<int32_t*>(o + 1), reinterpret_cast<float*>(o + 2),
                                                 reinterpret_cast<float*>(o + 3), nullptr, nullptr), "chirp_sync_batch");
        uint32_t h[4];
        detail::check(ultra_hip_memcpy_d2h(sync_slot_.ctx(), h, o, sizeof(h)), "d2h");
        result.detected = h[0] != 0;
        std::memcpy(&result.cfo_hz, &h[2], sizeof(float));
        std::memcpy(&result.correlation, &h[3], sizeof(float));
        result.has_training = true;
        if (result.detected) {
            int32_t start; std::memcpy(&start, &h[1], sizeof(start));
            result.start_sample = start;
            synced_ = true; last_cfo_ = result.cfo_hz; training_start_ = start > 0 ? start : 0;
        }
        last_sync_ = result;
        return result.detected;
    }
public:
    // ... or an external synchroniser hands in what detectSync would have filled
    void acceptSync(const SyncResult& r) {
        last_sync_ = r; synced_ = r.detected; cfo_hz_ = r.cfo_hz; training_start_ = r.start_sample > 0 ? r.start_sample : 0;
        detail::guarded_void("HipOfdmWaveform::acceptSync", [&] { demod_->setFrequencyOffset(r.cfo_hz); });
    }

private:
    // samples start at the first of two training symbols (OFDMChirpWaveform::process, :174-215)
    bool processImpl(SampleSpan samples) {
        // float initial_phase_rad = -2.0f * M_PI * cfo_hz_ * training_start_sample_ / sample_rate (double expr)
        float phase = static_cast<float>((((-2.0 * M_PI) * double(cfo_hz_)) * double(training_start_)) /
                                         double(config_.sample_rate));
        while (double(phase) > M_PI) phase = static_cast<float>(double(phase) - 2.0 * M_PI);
        while (double(phase) < -M_PI) phase = static_cast<float>(double(phase) + 2.0 * M_PI);
        demod_->setFrequencyOffsetWithPhase(cfo_hz_, phase);
        const bool ready = demod_->processPresynced(samples, 2);
        if (ready) {                                                     // ALL the demodulator's soft bits, 648 at a time (:203-213)
            soft_bits_.clear();
            while (demod_->hasPendingData()) {
                std::vector<float> chunk = demod_->getSoftBits();
                if (chunk.empty()) break;
                soft_bits_.insert(soft_bits_.end(), chunk.begin(), chunk.end());
            }
        }
        return ready;
    }
public:
    std::vector<float> getSoftBits() { return std::move(soft_bits_); }
    void reset() {                                                        // :221-230; the preset CFO survives
        detail::guarded_void("HipOfdmWaveform::reset", [&] { demod_->reset(); });
        soft_bits_.clear(); synced_ = false;
    }
    bool isSynced() const { return synced_ || demod_->isSynced(); }    // :232-234
    bool hasData() const { return !soft_bits_.empty() || demod_->hasPendingData(); }   // :236-238
    float estimatedSNR() const { return demod_->getEstimatedSNR(); }
    float estimatedCFO() const {                                        // :244-252
        return std::fabs(last_cfo_) > 0.1f ? last_cfo_ : demod_->getFrequencyOffset();
    }
    std::vector<std::complex<float>> getConstellationSymbols() const { return demod_->getConstellationSymbols(); }

    std::string getStatusString() const { return "OFDM-HIP " + std::to_string(config_.num_carriers) + " carriers"; }
    int getCarrierCount() const { return static_cast<int>(config_.num_carriers); }
    int getSamplesPerSymbol() const { return static_cast<int>(demod_->symbolSamples()); }
    // the dual chirp [up][gap][down][gap] (ChirpSync::getTotalSamples, src/sync/chirp_sync.hpp:534-544, with the waveform's
    // 500 ms / 100 ms: ofdm_chirp_waveform.cpp:39-49) + two training symbols (:304-309)
    int getPreambleSamples() const {
        const float fs = static_cast<float>(config_.sample_rate);
        const size_t chirp = static_cast<size_t>(fs * 500.0f / 1000.0f), gap = static_cast<size_t>(fs * 100.0f / 1000.0f);
        return static_cast<int>(2 * chirp + 2 * gap) + 2 * getSamplesPerSymbol();
    }
    int getMinSamplesForFrame() const {                                   // ofdm_chirp_waveform.cpp:311-331: every carrier is data
        const int bits_per_symbol = static_cast<int>(config_.num_carriers) * bitsPerCarrier();
        const int data_symbols = (648 + bits_per_symbol - 1) / bits_per_symbol;
        return (2 + data_symbols) * getSamplesPerSymbol();
    }
    float getThroughput(CodeRate rate) const {                            // :266-296
        static const float ratio[] = {0.25f, 0.333f, 0.5f, 0.667f, 0.75f, 0.833f, 0.5f};
        const float symbol_rate = static_cast<float>(config_.sample_rate) / getSamplesPerSymbol();
        const float raw_bps = symbol_rate * static_cast<int>(config_.num_carriers) * bitsPerCarrier();
        return raw_bps * ratio[static_cast<int>(rate) <= 5 ? static_cast<int>(rate) : 6];
    }

#ifdef ULTRA_HIP_WITH_REFERENCE
    ultra::protocol::WaveformMode getMode() const override { return ultra::protocol::WaveformMode::OFDM_CHIRP; }
    WaveformCapabilities getCapabilities() const override {               // :51-65
        WaveformCapabilities c;
        c.supports_cfo_correction = true; c.supports_doppler_correction = true; c.requires_pilots = false;
        c.supports_differential = true; c.min_snr_db = 10.0f; c.max_snr_db = 20.0f;
        c.max_throughput_bps = getThroughput(CodeRate::R2_3);
        c.preamble_duration_ms = float(getPreambleSamples() - 2 * getSamplesPerSymbol()) * 1000.0f / config_.sample_rate;
        return c;
    }
    // The transmit half is not on the hot path: the reference's own chirp generator and modulator behind its own waveform
    // class, configured like this one (:93-127).  No exception leaves an IWaveform ("Errors").
    void setTxFrequencyOffset(float cfo_hz) override { config_.tx_cfo_hz = cfo_hz; tx_.reset(); }
    Samples generatePreamble() override { return tx().generatePreamble(); }
    Samples modulate(const Bytes& encoded) override { return tx().modulate(encoded); }
#else
    void setTxFrequencyOffset(float) {}
#endif

private:
    static bool isDifferential(Modulation m) { return m == Modulation::DBPSK || m == Modulation::DQPSK || m == Modulation::D8PSK; }
    int bitsPerCarrier() const { return config_.modulation == Modulation::DBPSK ? 1 : config_.modulation == Modulation::D8PSK ? 3 : 2; }
    void initComponents() {                                              // :33-37
        demod_ = std::make_unique<HipOfdmDemodulator>(config_, device_);
#ifdef ULTRA_HIP_WITH_REFERENCE
        tx_.reset();
#endif
    }
#ifdef ULTRA_HIP_WITH_REFERENCE
    ultra::OFDMChirpWaveform& tx() { if (!tx_) tx_ = std::make_unique<ultra::OFDMChirpWaveform>(config_); return *tx_; }
    std::unique_ptr<ultra::OFDMChirpWaveform> tx_;
#endif
    ModemConfig config_;
    int device_;
    std::unique_ptr<HipOfdmDemodulator> demod_;
    detail::PooledSlot sync_slot_;               // dual-chirp detection: templates + scratch
    float last_cfo_ = 0.0f, cfo_hz_ = 0.0f;
    int training_start_ = 0;
    bool synced_ = false;
    SyncResult last_sync_{};
    std::vector<float> soft_bits_;
};

// ---------------------------------------------------------------------------------------------
// The v2 wire format behind getSoftBits(): mirrors the decode half of ultra::gui::RxPipeline
// (src/gui/modem/rx_pipeline.hpp:39-48,76-84; rx_pipeline.cpp:283-346,348-444) on ultra_hip_decode_frames_batch.
struct HipRxFrameResult {                      // gui::RxFrameResult
    bool success = false; Bytes frame_data; int frame_type = 0x10; int codewords_ok = 0; int codewords_failed = 0;
    float snr_estimate = 0.0f; float cfo_estimate = 0.0f; bool is_ping = false;
};
class HipRxFrameDecoder {
public:
    explicit HipRxFrameDecoder(int device = 0) : device_(device) {}
    void setDataMode(CodeRate rate, bool connected) { rate_ = rate; connected_ = connected; dropScratch(); ctx_.reset(); }
    void setInterleavingEnabled(bool enabled) { interleaving_ = enabled; dropScratch(); ctx_.reset(); }
    void setInterleaverConfig(size_t bits_per_symbol) {            // rx_pipeline.cpp:24-31
        if (bits_per_symbol != bits_per_symbol_) { bits_per_symbol_ = bits_per_symbol; dropScratch(); ctx_.reset(); }
    }
    int getExpectedCodewords() const { return expected_; }
    bool isAccumulating() const { return expected_ > 0; }
    // processFrame from `auto soft_bits = waveform->getSoftBits()` on
    HipRxFrameResult decodeSoftBits(std::span<const float> soft_bits) {
        HipRxFrameResult r;
        if (soft_bits.empty()) return r;
        if (!ctx_.p) {
            ModemConfig c; c.code_rate = connected_ ? rate_ : CodeRate::R1_4;      // rx_pipeline.cpp:356-366
            ctx_ = detail::Ctx(to_c_config(c, ULTRA_ENTRY_SYNCED, 44, 0), device_);
            detail::check(ultra_hip_set_deinterleave(ctx_.p, interleaving_ ? uint32_t(bits_per_symbol_) : 0u),
                          "ultra_hip_set_deinterleave");
        }
        ultra_hip_geometry g; detail::check(ultra_hip_get_geometry(ctx_.p, &g), "geometry");
        const size_t n = soft_bits.size(), stride = std::max<size_t>((n / 648) * (g.ldpc_k / 8), 1);
        void* d_s = s_soft_.get(ctx_.p, n * sizeof(float));
        void* d_r = s_res_.get(ctx_.p, sizeof(ultra_hip_frame_result));
        void* d_d = s_data_.get(ctx_.p, stride);
        detail::check(ultra_hip_memcpy_h2d(ctx_.p, d_s, soft_bits.data(), n * sizeof(float)), "h2d");
        detail::check(ultra_hip_decode_frames_batch(ctx_.p, static_cast<const float*>(d_s), n, static_cast<uint32_t>(n), 1,
                                                    static_cast<ultra_hip_frame_result*>(d_r),
                                                    static_cast<uint8_t*>(d_d), stride), "decode_frames_batch");
        ultra_hip_frame_result h;
        detail::check(ultra_hip_memcpy_d2h(ctx_.p, &h, d_r, sizeof(h)), "d2h");
        r.success = h.success != 0; r.is_ping = h.is_ping != 0; r.frame_type = h.frame_type;
        r.codewords_ok = h.codewords_ok; r.codewords_failed = h.codewords_failed; expected_ = h.expected_codewords;
        r.frame_data.resize(size_t(h.frame_len));
        if (h.frame_len > 0) detail::check(ultra_hip_memcpy_d2h(ctx_.p, r.frame_data.data(), d_d, size_t(h.frame_len)), "d2h");
        return r;
    }
private:
    int device_; CodeRate rate_ = CodeRate::R1_4; bool connected_ = false, interleaving_ = true;
    size_t bits_per_symbol_ = 60;           // the constructor's ChannelInterleaver(60, 648): rx_pipeline.cpp:13-18
    int expected_ = 0;
    detail::GrowBuf s_soft_, s_res_, s_data_;
    detail::Ctx ctx_;
    void dropScratch() { s_soft_.drop(); s_res_.drop(); s_data_.drop(); }
public:
    ~HipRxFrameDecoder() { dropScratch(); }
};


// ---------------------------------------------------------------------------------------------
// Schmidl-Cox flavour: ultra::OFDMNvisWaveform (src/waveform/ofdm_cox_waveform.cpp) — what
// WaveformFactory::create(WaveformMode::OFDM_COX) returns (src/waveform/waveform_factory.cpp:16-17,52-53) and the waveform
// the headline configuration runs on.  Its receive half is OFDMDemodulator::process behind detectSync / process / getSoftBits:
// here HipOfdmDemodulator, the same state machine on the device.
// Inside the reference tree (-DULTRA_HIP_WITH_REFERENCE) it derives from ultra::IWaveform and the transmit half is the
// reference's own OFDMModulator, so it can stand wherever an OFDMNvisWaveform stands.
class HipOfdmCoxWaveform
#ifdef ULTRA_HIP_WITH_REFERENCE
    : public ultra::IWaveform
#endif
{
public:
    static constexpr int kMaxSymbolsBeforeTimeout = HipOfdmDemodulator::kMaxSymbolsBeforeTimeout,
                         kMaxIdleCallsBeforeReset = HipOfdmDemodulator::kMaxIdleCallsBeforeReset;
    explicit HipOfdmCoxWaveform(const ModemConfig& config = defaultConfig(), int device = 0) : config_(config), device_(device) {
        initComponents();
    }
    static ModemConfig defaultConfig() {                                // OFDMNvisWaveform::OFDMNvisWaveform()
        ModemConfig c; c.fft_size = 512; c.num_carriers = 30; c.modulation = Modulation::QPSK; c.code_rate = CodeRate::R1_2;
        c.use_pilots = true; return c;
    }

    std::string getName() const { return "OFDM-COX-HIP"; }
    void configure(Modulation mod, CodeRate rate) {                     // ofdm_cox_waveform.cpp:52-68
        config_.modulation = mod; config_.code_rate = rate;
        config_.use_pilots = !(mod == Modulation::DBPSK || mod == Modulation::DQPSK || mod == Modulation::D8PSK);
        initComponents();
    }
    void setFrequencyOffset(float cfo_hz) {                            // :70-75
        cfo_hz_ = cfo_hz;
        detail::guarded_void("HipOfdmCoxWaveform::setFrequencyOffset", [&] { demod_->setFrequencyOffset(cfo_hz); });
    }
    Modulation getModulation() const { return config_.modulation; }
    CodeRate getCodeRate() const { return config_.code_rate; }
    float getFrequencyOffset() const { return cfo_hz_; }

    // OFDMNvisWaveform::detectSync (:98-120): feed the demodulator, report whether it is synced
    // No exception leaves an IWaveform: a failing C-ABI call is reported on stderr and reads as "not detected" / "not ready".
    bool detectSync(SampleSpan samples, SyncResult& result, float /*threshold*/ = 0.3f) {
        detail::guarded_void("HipOfdmCoxWaveform::detectSync", [&] { demod_->process(samples); });
        if (!demod_->isSynced()) return false;
        result.detected = true;
        result.start_sample = static_cast<int>(demod_->getLastSyncOffset());
        result.cfo_hz = demod_->getFrequencyOffset();
        result.snr_estimate = demod_->getEstimatedSNR();
        result.has_training = true;
        return true;
    }
    bool process(SampleSpan samples) {                                  // :122-134
        const bool ready = detail::guarded<bool>("HipOfdmCoxWaveform::process", false, [&] { return demod_->process(samples); });
        if (ready) soft_bits_ = demod_->getSoftBits();
        return ready;
    }
    std::vector<float> getSoftBits() { return std::move(soft_bits_); }
    void reset() {                                                      // :140-147 -> OFDMDemodulator::reset (:987-1017)
        detail::guarded_void("HipOfdmCoxWaveform::reset", [&] { demod_->reset(); });
        soft_bits_.clear();
    }
    bool isSynced() const { return demod_->isSynced(); }
    bool hasData() const { return !soft_bits_.empty() || demod_->hasPendingData(); }   // :153-155
    float estimatedSNR() const { return demod_->getEstimatedSNR(); }
    float estimatedCFO() const { return demod_->getFrequencyOffset(); }
    float coarseCFO() const { return demod_->coarseCFO(); }            // Impl::estimateCoarseCFO at the last sync
    size_t getLastSyncOffset() const { return demod_->getLastSyncOffset(); }
    std::vector<std::complex<float>> getConstellationSymbols() const { return demod_->getConstellationSymbols(); }
    HipOfdmDemodulator& demodulator() { return *demod_; }

    std::string getStatusString() const {
        return "OFDM-COX " + std::to_string(config_.num_carriers) + " carriers (HIP)" + (config_.use_pilots ? " (pilots)" : "");
    }
    int getCarrierCount() const { return static_cast<int>(config_.num_carriers); }
    int getSamplesPerSymbol() const { return static_cast<int>(demod_->symbolSamples()); }
    int getPreambleSamples() const { return 2 * getSamplesPerSymbol(); }
    int getMinSamplesForFrame() const {                                   // ofdm_cox_waveform.cpp:231-258
        const int bits_per_symbol = dataCarriers() * bitsPerCarrier();
        return (2 + (648 + bits_per_symbol - 1) / bits_per_symbol) * getSamplesPerSymbol();
    }
    float getThroughput(CodeRate rate) const {                            // :184-220
        static const float ratio[] = {0.25f, 0.333f, 0.5f, 0.667f, 0.75f, 0.833f, 0.5f};
        return float(config_.sample_rate) / float(getSamplesPerSymbol()) * float(dataCarriers()) * float(bitsPerCarrier()) *
               ratio[static_cast<int>(rate) <= 6 ? static_cast<int>(rate) : 2];
    }

#ifdef ULTRA_HIP_WITH_REFERENCE
    ultra::protocol::WaveformMode getMode() const override { return ultra::protocol::WaveformMode::OFDM_COX; }
    WaveformCapabilities getCapabilities() const override {               // :33-50
        WaveformCapabilities c;
        c.supports_cfo_correction = true; c.supports_doppler_correction = true; c.requires_pilots = config_.use_pilots;
        c.supports_differential = true;
        c.min_snr_db = config_.use_pilots ? 17.0f : 12.0f; c.max_snr_db = 35.0f;
        c.max_throughput_bps = getThroughput(CodeRate::R3_4);
        c.preamble_duration_ms = 2.0f * getSamplesPerSymbol() * 1000.0f / config_.sample_rate;
        return c;
    }
    void setTxFrequencyOffset(float cfo_hz) override { config_.tx_cfo_hz = cfo_hz; initComponents(); }   // :77-83
    // the transmit half is not on the hot path: the reference's own modulator (:85-96)
    Samples generatePreamble() override { return modulator_->generatePreamble(); }
    Samples modulate(const Bytes& encoded) override {
        return modulator_->modulate(ultra::ByteSpan(encoded.data(), encoded.size()), config_.modulation);
    }
#else
    void setTxFrequencyOffset(float) {}
#endif

private:
    void initComponents() {                                              // :25-28
#ifdef ULTRA_HIP_WITH_REFERENCE
        modulator_ = std::make_unique<ultra::OFDMModulator>(config_);
#endif
        demod_ = std::make_unique<HipOfdmDemodulator>(config_, device_);
        soft_bits_.clear();
    }
    int dataCarriers() const {
        int d = static_cast<int>(config_.num_carriers);
        if (config_.use_pilots && config_.pilot_spacing > 0) d -= config_.num_carriers / config_.pilot_spacing;
        return d;
    }
    int bitsPerCarrier() const {
        static const int bpc[] = {1, 1, 2, 2, 3, 3, 4, 5, 6, 2, 2};
        const int m = static_cast<int>(config_.modulation);
        return bpc[m <= 10 ? m : 3];
    }

    ModemConfig config_;
    int device_;
#ifdef ULTRA_HIP_WITH_REFERENCE
    std::unique_ptr<ultra::OFDMModulator> modulator_;
#endif
    std::unique_ptr<HipOfdmDemodulator> demod_;
    float cfo_hz_ = 0.0f;
    std::vector<float> soft_bits_;
};

}  // namespace ultra_hip
