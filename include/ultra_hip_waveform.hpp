// ultra_hip_waveform.hpp — header-only C++ adapter over the C-ABI (include/ultra_hip.h).
//
// Two classes with the reference's own method names, argument meaning and failure behaviour:
//   ultra_hip::HipLDPCDecoder   ~ ultra::LDPCDecoder    (include/ultra/fec.hpp:48-77)
//   ultra_hip::HipOfdmWaveform  ~ the receive half of ultra::IWaveform
//                                 (src/waveform/waveform_interface.hpp:47-157), shaped like
//                                 OFDMChirpWaveform::process (src/waveform/ofdm_chirp_waveform.cpp:174-215):
//                                 an external synchroniser provides timing + CFO, process() runs the
//                                 presynced entry on the GPU and getSoftBits() hands back the LLRs.
//
// Compiled inside the reference tree (-DULTRA_HIP_WITH_REFERENCE, -I<ref>/include -I<ref>/src) the
// waveform derives from ultra::IWaveform and uses the reference's own types, so RxPipeline /
// ModemEngine / the Monte-Carlo tools can hold it through a WaveformPtr unchanged.
// Compiled stand-alone it uses the small mirror types below.  Either way it links only against
// libultra_hip.so; no HIP headers are needed by the caller.
#pragma once

#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdint>
#include <cstring>
#include <memory>
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
    return k;
}

namespace detail {
struct Ctx {                               // RAII owner of an ultra_hip_ctx
    ultra_hip_ctx* p = nullptr;
    Ctx() = default;
    Ctx(const ultra_hip_config& c, int device) {
        int rc = ultra_hip_create(&c, device, nullptr, &p);
        if (rc != ULTRA_HIP_OK) throw std::runtime_error(std::string("ultra_hip_create: ") + ultra_hip_strerror(rc));
    }
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
// a device buffer that lives with its owner and only ever grows (the adapters' per-call scratch)
struct GrowBuf {
    std::unique_ptr<DevBuf> b; size_t cap = 0; ultra_hip_ctx* owner = nullptr;
    void* get(ultra_hip_ctx* ctx, size_t bytes) {
        if (!b || owner != ctx || bytes > cap) { b.reset(); cap = std::max<size_t>(bytes, 2 * cap); b = std::make_unique<DevBuf>(ctx, cap); owner = ctx; }
        return b->d;
    }
    void drop() { b.reset(); cap = 0; owner = nullptr; }
};
inline void check(int rc, const char* what) {
    if (rc != ULTRA_HIP_OK) throw std::runtime_error(std::string(what) + ": " + ultra_hip_strerror(rc));
}
}  // namespace detail

// ---------------------------------------------------------------------------------------------
class HipLDPCDecoder {
public:
    explicit HipLDPCDecoder(CodeRate rate, int device = 0) : rate_(rate), device_(device) { rebuild(); }

    // Decode from soft bits: bit-level multi-block semantics of LDPCDecoder::decodeSoft
    // (src/fec/ldpc_decoder.cpp:283-428): <= 648 LLRs one block (zero padded), more -> every full block
    // contributes exactly k bits, a zero-padded tail block is decoded too, bits packed once at the end.
    Bytes decodeSoft(std::span<const float> llrs) {
        if (llrs.empty()) { last_success_ = false; return {}; }
        ultra_hip_geometry g; detail::check(ultra_hip_get_geometry(ctx_.p, &g), "geometry");
        const size_t n = 648, nblocks = (llrs.size() + n - 1) / n;
        std::vector<float> padded(nblocks * n, 0.0f);
        std::memcpy(padded.data(), llrs.data(), llrs.size() * sizeof(float));
        std::vector<uint8_t> bytes(nblocks * g.decoded_bytes), ok(nblocks);
        std::vector<int32_t> iters(nblocks);
        decodeBatch(padded.data(), nblocks, bytes.data(), iters.data(), ok.data());
        const bool has_tail = (llrs.size() % n) != 0 && nblocks > 1;
        bool all = true; for (auto v : ok) all = all && v;
        last_success_ = (nblocks == 1 || has_tail) ? ok.back() != 0 : all;
        last_iters_ = iters.back();
        if (nblocks == 1) return Bytes(bytes.begin(), bytes.end());
        std::vector<uint8_t> bits;
        bits.reserve(nblocks * g.ldpc_k);
        for (size_t b = 0; b < nblocks; ++b)
            for (uint32_t j = 0; j < g.ldpc_k; ++j)
                bits.push_back((bytes[b * g.decoded_bytes + j / 8] >> (7 - j % 8)) & 1);
        Bytes out((bits.size() + 7) / 8, 0);
        for (size_t i = 0; i < bits.size(); ++i) if (bits[i]) out[i / 8] |= uint8_t(1u << (7 - i % 8));
        return out;
    }
    // RxPipeline::setInterleaverConfig + deinterleaveCodewords (src/gui/modem/rx_pipeline.cpp:24-31,475-491):
    // every codeword goes through ChannelInterleaver(bits_per_symbol, 648)::deinterleave before it is
    // decoded — on the GPU, fused into the decoder's LLR load.  0 switches it off.
    void setDeinterleave(size_t bits_per_symbol) {
        deinterleave_ = (uint32_t)bits_per_symbol;
        detail::check(ultra_hip_set_deinterleave(ctx_.p, deinterleave_), "ultra_hip_set_deinterleave");
    }
    Bytes decode(std::span<const uint8_t> coded) {                     // ldpc_decoder.cpp:267-281
        std::vector<float> llrs; llrs.reserve(coded.size() * 8);
        for (uint8_t byte : coded) for (int b = 7; b >= 0; --b) llrs.push_back(((byte >> b) & 1) ? -6.0f : 6.0f);
        return decodeSoft(llrs);
    }
    // n_cw independent codewords, host buffers: llr [n_cw][648] -> bytes [n_cw][ceil(k/8)], iters, ok
    void decodeBatch(const float* llr, size_t n_cw, uint8_t* bytes, int32_t* iters, uint8_t* ok) {
        ultra_hip_geometry g; detail::check(ultra_hip_get_geometry(ctx_.p, &g), "geometry");
        void* d_llr = s_llr_.get(ctx_.p, n_cw * 648 * sizeof(float));
        void* d_b = s_bytes_.get(ctx_.p, n_cw * g.decoded_bytes);
        void* d_i = s_iters_.get(ctx_.p, n_cw * sizeof(int32_t));
        void* d_o = s_ok_.get(ctx_.p, n_cw);
        detail::check(ultra_hip_memcpy_h2d(ctx_.p, d_llr, llr, n_cw * 648 * sizeof(float)), "h2d");
        detail::check(ultra_hip_ldpc_decode_batch(ctx_.p, static_cast<const float*>(d_llr), n_cw,
                                                  static_cast<uint8_t*>(d_b), static_cast<int32_t*>(d_i),
                                                  static_cast<uint8_t*>(d_o), nullptr), "ldpc_decode_batch");
        detail::check(ultra_hip_memcpy_d2h(ctx_.p, bytes, d_b, n_cw * g.decoded_bytes), "d2h");
        detail::check(ultra_hip_memcpy_d2h(ctx_.p, iters, d_i, n_cw * sizeof(int32_t)), "d2h");
        detail::check(ultra_hip_memcpy_d2h(ctx_.p, ok, d_o, n_cw), "d2h");
    }
    bool lastDecodeSuccess() const { return last_success_; }
    int lastIterations() const { return last_iters_; }
    void setRate(CodeRate rate) { rate_ = rate; rebuild(); }
    CodeRate getRate() const { return rate_; }
    void setMaxIterations(int max_iter) { max_iter_ = max_iter; rebuild(); }

private:
    void rebuild() {
        s_llr_.drop(); s_bytes_.drop(); s_iters_.drop(); s_ok_.drop();   // they belong to the context that goes away
        ModemConfig c; c.code_rate = rate_;
        ctx_ = detail::Ctx(to_c_config(c, ULTRA_ENTRY_SYNCED, 44, 0, static_cast<uint32_t>(max_iter_)), device_);
        if (deinterleave_) detail::check(ultra_hip_set_deinterleave(ctx_.p, deinterleave_), "ultra_hip_set_deinterleave");
    }
    CodeRate rate_; int device_; int max_iter_ = 50; bool last_success_ = false; int last_iters_ = 0;
    uint32_t deinterleave_ = 0;
    detail::GrowBuf s_llr_, s_bytes_, s_iters_, s_ok_;   // declared before ctx_: destroyed after... see ~HipLDPCDecoder
    detail::Ctx ctx_;
public:
    ~HipLDPCDecoder() { s_llr_.drop(); s_bytes_.drop(); s_iters_.drop(); s_ok_.drop(); }   // buffers first, then the context
};

// ---------------------------------------------------------------------------------------------
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
    }

    std::string getName() const { return "OFDM_HIP"; }
    void configure(Modulation mod, CodeRate rate) {                     // OFDMChirpWaveform::configure (:67-84)
        if (!isDifferential(mod)) mod = Modulation::DQPSK;
        config_.modulation = mod; config_.code_rate = rate;
        config_.use_pilots = false;
        s_audio_.drop(); s_cp_.drop(); s_llr_.drop(); s_state_.drop();
        ctx_.reset();
        demodReset();
    }
    void setFrequencyOffset(float cfo_hz) { cfo_hz_ = cfo_hz; }
    void setTxFrequencyOffset(float) {}
    Modulation getModulation() const { return config_.modulation; }
    CodeRate getCodeRate() const { return config_.code_rate; }
    float getFrequencyOffset() const { return cfo_hz_; }

    // OFDMChirpWaveform::detectSync (src/waveform/ofdm_chirp_waveform.cpp:129-172): dual-chirp detection on the
    // device (scope row f4, ultra_hip_chirp_sync_batch); start_sample = where the two training symbols start.
    bool detectSync(SampleSpan samples, SyncResult& result, float threshold = 0.3f) {
        if (!sync_ctx_.p) sync_ctx_ = detail::Ctx(to_c_config(config_, ULTRA_ENTRY_PRESYNCED, 1, 2), device_);
        void* d_a = s_sync_audio_.get(sync_ctx_.p, std::max<size_t>(samples.size(), 1) * sizeof(float));
        void* d_o = s_sync_out_.get(sync_ctx_.p, 4 * sizeof(uint32_t));
        detail::check(ultra_hip_memcpy_h2d(sync_ctx_.p, d_a, samples.data(), samples.size() * sizeof(float)), "h2d");
        uint32_t* o = static_cast<uint32_t*>(d_o);
        detail::check(ultra_hip_chirp_sync_batch(sync_ctx_.p, static_cast<const float*>(d_a), samples.size(),
                                                 static_cast<uint32_t>(samples.size()), 1, threshold, o,
                                                 reinterpret_cast<int32_t*>(o + 1), reinterpret_cast<float*>(o + 2),
                                                 reinterpret_cast<float*>(o + 3), nullptr, nullptr), "chirp_sync_batch");
        uint32_t h[4];
        detail::check(ultra_hip_memcpy_d2h(sync_ctx_.p, h, d_o, sizeof(h)), "d2h");
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
    // ... or an external synchroniser hands in what detectSync would have filled
    void acceptSync(const SyncResult& r) {
        last_sync_ = r; synced_ = r.detected; cfo_hz_ = r.cfo_hz; training_start_ = r.start_sample > 0 ? r.start_sample : 0;
    }

    // samples start at the first of two training symbols (OFDMChirpWaveform::process)
    bool process(SampleSpan samples) {
        const uint32_t sym = symbolSamples();
        if (samples.size() < size_t(3) * sym) return false;
        const uint32_t n_data = static_cast<uint32_t>(samples.size() / sym) - 2;
        if (!ctx_.p || n_data != n_data_) {
            s_audio_.drop(); s_cp_.drop(); s_llr_.drop(); s_state_.drop();
            ctx_ = detail::Ctx(to_c_config(config_, ULTRA_ENTRY_PRESYNCED, n_data, 2), device_);
            n_data_ = n_data;
        }
        ultra_hip_geometry g; detail::check(ultra_hip_get_geometry(ctx_.p, &g), "geometry");
        // float initial_phase_rad = -2.0f * M_PI * cfo_hz_ * training_start_sample_ / sample_rate (double expr)
        float phase = static_cast<float>((((-2.0 * M_PI) * double(cfo_hz_)) * double(training_start_)) /
                                         double(config_.sample_rate));
        while (double(phase) > M_PI) phase = static_cast<float>(double(phase) - 2.0 * M_PI);
        while (double(phase) < -M_PI) phase = static_cast<float>(double(phase) + 2.0 * M_PI);
        void* d_a = s_audio_.get(ctx_.p, g.frame_samples * sizeof(float));
        void* d_c = s_cp_.get(ctx_.p, 2 * sizeof(float));
        void* d_l = s_llr_.get(ctx_.p, g.llrs_per_frame * sizeof(float));
        void* d_s = s_state_.get(ctx_.p, ULTRA_HIP_STATE_FLOATS * sizeof(float));
        const float cp[2] = {cfo_hz_, phase};
        detail::check(ultra_hip_memcpy_h2d(ctx_.p, d_a, samples.data(), g.frame_samples * sizeof(float)), "h2d");
        detail::check(ultra_hip_memcpy_h2d(ctx_.p, d_c, cp, sizeof(cp)), "h2d");
        detail::check(ultra_hip_demod_batch(ctx_.p, static_cast<const float*>(d_a), g.frame_samples,
                                            static_cast<const float*>(d_c), static_cast<const float*>(d_c) + 1, 1,
                                            static_cast<float*>(d_l), static_cast<float*>(d_s)), "demod_batch");
        pending_.resize(g.llrs_per_frame);
        detail::check(ultra_hip_memcpy_d2h(ctx_.p, pending_.data(), d_l, g.llrs_per_frame * sizeof(float)), "d2h");
        detail::check(ultra_hip_memcpy_d2h(ctx_.p, state_, d_s, sizeof(state_)), "d2h");
        demod_synced_ = true;                                            // processPresynced: state = SYNCED (demodulator.cpp:886)
        // ready = soft_bits.size() >= LDPC_BLOCK_SIZE (:985); only then does the waveform take the demodulator's bits (:203-213)
        const bool ready = pending_.size() >= 648;
        if (ready) { soft_bits_ = std::move(pending_); pending_.clear(); }
        return ready;
    }
    std::vector<float> getSoftBits() { return std::move(soft_bits_); }
    void reset() {                                                        // ofdm_chirp_waveform.cpp:221-230; the preset CFO survives
        demodReset();
        soft_bits_.clear(); synced_ = false;
    }
    ~HipOfdmWaveform() {                                                  // buffers before the contexts they came from
        s_audio_.drop(); s_cp_.drop(); s_llr_.drop(); s_state_.drop(); s_sync_audio_.drop(); s_sync_out_.drop();
    }
    bool isSynced() const { return synced_ || demod_synced_; }          // :232-234
    bool hasData() const { return !soft_bits_.empty() || !pending_.empty(); }   // :236-238 (OFDMDemodulator::hasPendingData)
    float estimatedSNR() const { return 10.0f * std::log10(state_[ULTRA_HIP_STATE_SNR_LINEAR]); }
    float estimatedCFO() const {                                        // ofdm_chirp_waveform.cpp:244-252
        return std::fabs(last_cfo_) > 0.1f ? last_cfo_ : state_[ULTRA_HIP_STATE_FREQ_OFFSET_HZ];
    }
    std::vector<std::complex<float>> getConstellationSymbols() const { return {}; }   // GUI ring: not produced

    std::string getStatusString() const { return "OFDM-HIP " + std::to_string(config_.num_carriers) + " carriers"; }
    int getCarrierCount() const { return static_cast<int>(config_.num_carriers); }
    int getSamplesPerSymbol() const { return static_cast<int>(symbolSamples()); }
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
    // remaining pure virtuals of ultra::IWaveform: transmit side and capability report are not part
    // of the receive hot path; they delegate to nothing and say so.
    ultra::protocol::WaveformMode getMode() const override { return ultra::protocol::WaveformMode::OFDM_CHIRP; }
    WaveformCapabilities getCapabilities() const override {
        WaveformCapabilities c; c.supports_cfo_correction = true; c.requires_pilots = config_.use_pilots; return c;
    }
    Samples generatePreamble() override { throw std::logic_error("HipOfdmWaveform is receive-only"); }
    Samples modulate(const Bytes&) override { throw std::logic_error("HipOfdmWaveform is receive-only"); }
#endif

private:
    static bool isDifferential(Modulation m) { return m == Modulation::DBPSK || m == Modulation::DQPSK || m == Modulation::D8PSK; }
    int bitsPerCarrier() const { return config_.modulation == Modulation::DBPSK ? 1 : config_.modulation == Modulation::D8PSK ? 3 : 2; }
    // what OFDMDemodulator::reset (demodulator.cpp:987-1017) leaves behind, as far as this adapter shows it: not synced, no
    // soft bits, SNR 1.0 (0 dB), CFO 0 (the waveform's own cfo_hz_ is kept and handed over again by process())
    void demodReset() {
        demod_synced_ = false; pending_.clear();
        const float fresh[ULTRA_HIP_STATE_FLOATS] = {0, 0, 1, 0, 0, 0, 0, 0};
        std::memcpy(state_, fresh, sizeof(state_));
    }
    uint32_t symbolSamples() const {
        const uint32_t base = config_.cp_mode == decltype(config_.cp_mode)(0) ? 32u
                            : config_.cp_mode == decltype(config_.cp_mode)(2) ? 64u : 48u;
        return config_.fft_size + base * (config_.fft_size / 512) + config_.symbol_guard;
    }
    ModemConfig config_;
    int device_;
    bool demod_synced_ = false;
    std::vector<float> pending_;                 // soft bits the demodulator holds while a call produced fewer than one codeword
    detail::Ctx ctx_, sync_ctx_;
    detail::GrowBuf s_audio_, s_cp_, s_llr_, s_state_, s_sync_audio_, s_sync_out_;   // persistent per-call scratch
    float last_cfo_ = 0.0f;
    uint32_t n_data_ = 0;
    float cfo_hz_ = 0.0f;
    int training_start_ = 0;
    bool synced_ = false;
    SyncResult last_sync_{};
    std::vector<float> soft_bits_;
    float state_[ULTRA_HIP_STATE_FLOATS] = {0, 0, 1, 0, 0, 0, 0, 0};
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
// the headline configuration runs on.  Its receive half is OFDMDemodulator::process (src/ofdm/demodulator.cpp:461-741)
// behind detectSync / process / getSoftBits; this class is that state machine on the device, as a LIVE stream:
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
// Device buffers live as long as the object.
// Inside the reference tree (-DULTRA_HIP_WITH_REFERENCE) it derives from ultra::IWaveform and the transmit half is the
// reference's own OFDMModulator, so it can stand wherever an OFDMNvisWaveform stands.
#ifdef ULTRA_HIP_WITH_REFERENCE
}  // namespace ultra_hip
#include "ultra/ofdm.hpp"
namespace ultra_hip {
#endif
class HipOfdmCoxWaveform
#ifdef ULTRA_HIP_WITH_REFERENCE
    : public ultra::IWaveform
#endif
{
public:
    static constexpr int kMaxSymbolsBeforeTimeout = 250, kMaxIdleCallsBeforeReset = 10;   // demodulator_constants.hpp:37-38
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
    void setFrequencyOffset(float cfo_hz) {                             // :70-75 -> OFDMDemodulator::setFrequencyOffset
        cfo_hz_ = cfo_hz; freq_offset_hz_ = cfo_hz;
        if (synced_ && synced_symbols_ > 0) detail::check(ultra_hip_demod_stream_set_cfo(ctx_.p, 0, cfo_hz), "stream_set_cfo");
        else pending_cfo_ = true;                                       // the next symbol 0 starts from it
    }
    Modulation getModulation() const { return config_.modulation; }
    CodeRate getCodeRate() const { return config_.code_rate; }
    float getFrequencyOffset() const { return cfo_hz_; }

    // OFDMNvisWaveform::detectSync (:98-120): feed the demodulator, report whether it is synced
    bool detectSync(SampleSpan samples, SyncResult& result, float /*threshold*/ = 0.3f) {
        demodProcess(samples);
        if (!synced_) return false;
        result.detected = true;
        result.start_sample = static_cast<int>(last_sync_offset_);
        result.cfo_hz = freq_offset_hz_;
        result.snr_estimate = estimatedSNR();
        result.has_training = true;
        return true;
    }
    bool process(SampleSpan samples) {                                  // :122-134
        const bool ready = demodProcess(samples);
        if (ready) soft_bits_ = demodGetSoftBits();
        return ready;
    }
    std::vector<float> getSoftBits() { return std::move(soft_bits_); }
    void reset() {                                                      // :140-147 -> OFDMDemodulator::reset (:987-1017)
        synced_ = false; synced_symbols_ = 0; idle_calls_ = 0;
        rx_.clear(); origin_ = fed_; d_origin_ = fed_; demod_soft_.clear(); soft_bits_.clear();
        freq_offset_hz_ = 0.0f; pending_cfo_ = false;
        state_[ULTRA_HIP_STATE_SNR_LINEAR] = 1.0f; state_[ULTRA_HIP_STATE_FREQ_OFFSET_HZ] = 0.0f;
        restartSearch();
    }
    bool isSynced() const { return synced_; }
    bool hasData() const {                                              // :153-155, OFDMDemodulator::hasPendingData
        return !soft_bits_.empty() || (synced_ && (!demod_soft_.empty() || fed_ - origin_ >= symbolSamples()));
    }
    float estimatedSNR() const { return 10.0f * std::log10(state_[ULTRA_HIP_STATE_SNR_LINEAR]); }
    float estimatedCFO() const { return freq_offset_hz_; }
    float coarseCFO() const { return coarse_cfo_; }                     // Impl::estimateCoarseCFO at the last sync
    size_t getLastSyncOffset() const { return last_sync_offset_; }
    std::vector<std::complex<float>> getConstellationSymbols() const { return {}; }   // GUI ring: not produced

    std::string getStatusString() const {
        return "OFDM-COX " + std::to_string(config_.num_carriers) + " carriers (HIP)" + (config_.use_pilots ? " (pilots)" : "");
    }
    int getCarrierCount() const { return static_cast<int>(config_.num_carriers); }
    int getSamplesPerSymbol() const { return static_cast<int>(symbolSamples()); }
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
    void setTxFrequencyOffset(float cfo_hz) override { config_.tx_cfo_hz = cfo_hz; modulator_ = std::make_unique<ultra::OFDMModulator>(config_); }
    // the transmit half is not on the hot path: the reference's own modulator (:85-96)
    Samples generatePreamble() override { return modulator_->generatePreamble(); }
    Samples modulate(const Bytes& encoded) override {
        return modulator_->modulate(ultra::ByteSpan(encoded.data(), encoded.size()), config_.modulation);
    }
#else
    void setTxFrequencyOffset(float) {}
#endif

private:
    void initComponents() {
#ifdef ULTRA_HIP_WITH_REFERENCE
        modulator_ = std::make_unique<ultra::OFDMModulator>(config_);
#endif
        d_small_.reset(); d_rx_.reset(); d_llr_.reset();               // they belong to the context that goes away
        // the longest frame process() will ever see: MAX_SYMBOLS_BEFORE_TIMEOUT + 1 symbols
        ctx_ = detail::Ctx(to_c_config(config_, ULTRA_ENTRY_SYNCED, kMaxSymbolsBeforeTimeout + 1, 0), device_);
        detail::check(ultra_hip_get_geometry(ctx_.p, &geo_), "geometry");
        d_small_ = std::make_unique<detail::DevBuf>(ctx_.p, 32 * sizeof(uint32_t));   // resume[4], found, data_start, cfo, sync offset, state[8], cfo in
        rx_cap_ = 1u << 16; d_rx_ = std::make_unique<detail::DevBuf>(ctx_.p, rx_cap_ * sizeof(float));
        d_llr_.reset(); llr_cap_ = 0;
        synced_ = false; synced_symbols_ = 0; idle_calls_ = 0;
        rx_.clear(); origin_ = fed_ = d_origin_ = 0; demod_soft_.clear(); soft_bits_.clear(); noise_floor_bits_ = 0;
        restartSearch();
    }
    uint32_t* small() const { return static_cast<uint32_t*>(d_small_->d); }
    // the search restarts on whatever is still buffered: rx_buffer = [origin_, fed_)
    void restartSearch() {
        const uint32_t r[4] = {origin_, fed_, noise_floor_bits_, 0u};
        detail::check(ultra_hip_memcpy_h2d(ctx_.p, small(), r, sizeof(r)), "h2d");
    }
    // sample indices are 32-bit and absolute: long before they run out (6 h of audio) the origin moves to rx_buffer's start
    void rebase() {
        uint32_t r[4];
        detail::check(ultra_hip_memcpy_d2h(ctx_.p, r, small(), sizeof(r)), "d2h");
        const uint32_t shift = origin_;
        r[0] -= shift; r[1] -= shift;
        detail::check(ultra_hip_memcpy_h2d(ctx_.p, small(), r, sizeof(r)), "h2d");
        fed_ -= shift; origin_ = 0; d_origin_ = 0;
        if (!rx_.empty()) detail::check(ultra_hip_memcpy_h2d(ctx_.p, d_rx_->d, rx_.data(), rx_.size() * sizeof(float)), "h2d");
    }
    void appendSamples(SampleSpan samples) {
        if (!synced_ && fed_ > (1u << 29)) rebase();
        rx_.insert(rx_.end(), samples.begin(), samples.end());
        const size_t need = size_t(fed_ - d_origin_) + samples.size();
        if (need > rx_cap_ || (d_origin_ < origin_ && size_t(origin_ - d_origin_) > rx_cap_ / 2)) {
            // grow, or drop the consumed front: the device window restarts at origin_ from the host's copy
            const size_t live = rx_.size();
            if (live > rx_cap_ || !d_rx_) { rx_cap_ = std::max<size_t>(2 * live, 1u << 16); d_rx_ = std::make_unique<detail::DevBuf>(ctx_.p, rx_cap_ * sizeof(float)); }
            d_origin_ = origin_;
            if (live) detail::check(ultra_hip_memcpy_h2d(ctx_.p, d_rx_->d, rx_.data(), live * sizeof(float)), "h2d");
        } else if (!samples.empty()) {
            detail::check(ultra_hip_memcpy_h2d(ctx_.p, static_cast<float*>(d_rx_->d) + (fed_ - d_origin_), samples.data(),
                                               samples.size() * sizeof(float)), "h2d");
        }
        fed_ += static_cast<uint32_t>(samples.size());
    }
    void consumeTo(uint32_t abs_index) {                                 // rx_buffer.erase(begin, begin + n)
        rx_.erase(rx_.begin(), rx_.begin() + (abs_index - origin_));
        origin_ = abs_index;
    }
    void toSearching() { synced_ = false; synced_symbols_ = 0; idle_calls_ = 0; restartSearch(); }

    // OFDMDemodulator::process
    bool demodProcess(SampleSpan samples) {
        appendSamples(samples);
        if (!synced_) {
            uint32_t* w = small();
            detail::check(ultra_hip_acquire_stream_batch(ctx_.p, static_cast<const float*>(d_rx_->d), rx_cap_, d_origin_, fed_, 1,
                                                         w, w + 4, w + 5, reinterpret_cast<float*>(w + 6), w + 7), "acquire_stream");
            uint32_t h[8];
            detail::check(ultra_hip_memcpy_d2h(ctx_.p, h, w, sizeof(h)), "d2h");
            noise_floor_bits_ = h[2];
            if (h[4]) {                                                  // SEARCHING -> SYNCED (:533-591)
                std::memcpy(&coarse_cfo_, &h[6], sizeof(float));
                freq_offset_hz_ = coarse_cfo_; last_sync_offset_ = h[7];
                consumeTo(h[5]);
                synced_ = true; synced_symbols_ = 0; pending_cfo_ = false;
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
            detail::check(ultra_hip_resync_stream_batch(ctx_.p, static_cast<const float*>(d_rx_->d), rx_cap_, d_origin_, fed_, 1,
                                                        w, w + 4, w + 5, reinterpret_cast<float*>(w + 6), w + 7), "resync_stream");
            uint32_t h[8];
            detail::check(ultra_hip_memcpy_d2h(ctx_.p, h, w, sizeof(h)), "d2h");
            if (h[4]) {                                                  // the old frame is abandoned, the tracker starts afresh
                std::memcpy(&coarse_cfo_, &h[6], sizeof(float));
                freq_offset_hz_ = coarse_cfo_;
                consumeTo(h[5]);
                demod_soft_.clear();
                synced_symbols_ = 0; idle_calls_ = 0; pending_cfo_ = false;
            }
        }
        const uint32_t sym = symbolSamples();
        uint32_t n_new = (fed_ - origin_) / sym;
        const uint32_t room = uint32_t(kMaxSymbolsBeforeTimeout + 1) - synced_symbols_;
        if (n_new > room) n_new = room;
        const size_t soft_before = demod_soft_.size();
        if (n_new > 0) {
            const size_t n_llr = size_t(n_new) * geo_.llrs_per_symbol;
            if (n_llr > llr_cap_) { llr_cap_ = std::max<size_t>(2 * n_llr, 4096); d_llr_ = std::make_unique<detail::DevBuf>(ctx_.p, llr_cap_ * sizeof(float)); }
            float* d_state = reinterpret_cast<float*>(small() + 8);
            float* d_cfo = reinterpret_cast<float*>(small() + 16);
            const float cfo0 = pending_cfo_ ? cfo_hz_ : coarse_cfo_;
            if (synced_symbols_ == 0) detail::check(ultra_hip_memcpy_h2d(ctx_.p, d_cfo, &cfo0, sizeof(float)), "h2d");
            detail::check(ultra_hip_demod_stream_batch(ctx_.p, static_cast<const float*>(d_rx_->d) + (origin_ - d_origin_),
                                                       size_t(n_new) * sym, d_cfo, nullptr, 1, synced_symbols_, n_new,
                                                       static_cast<float*>(d_llr_->d), d_state), "demod_stream");
            const size_t at = demod_soft_.size();
            demod_soft_.resize(at + n_llr);
            detail::check(ultra_hip_memcpy_d2h(ctx_.p, demod_soft_.data() + at, d_llr_->d, n_llr * sizeof(float)), "d2h");
            detail::check(ultra_hip_memcpy_d2h(ctx_.p, state_, d_state, sizeof(state_)), "d2h");
            freq_offset_hz_ = state_[ULTRA_HIP_STATE_FREQ_OFFSET_HZ];
            consumeTo(origin_ + n_new * sym);
            synced_symbols_ += n_new; pending_cfo_ = false;
            if (synced_symbols_ > uint32_t(kMaxSymbolsBeforeTimeout)) {   // sync timeout (:683-691)
                toSearching();
                return demod_soft_.size() >= 648;
            }
        }
        if (demod_soft_.size() == soft_before) {                         // idle calls (:704-716)
            if (++idle_calls_ > kMaxIdleCallsBeforeReset) { toSearching(); return demod_soft_.size() >= 648; }
        } else {
            idle_calls_ = 0;
        }
        const bool has_codeword = demod_soft_.size() >= 648;
        if (!has_codeword && synced_symbols_ > 0 && samples.empty() && n_new == 0) {   // frame complete (:720-731)
            toSearching();
            demod_soft_.clear();
        }
        return has_codeword;
    }
    std::vector<float> demodGetSoftBits() {                              // 648 at a time (demodulator.cpp:766-791)
        if (demod_soft_.size() <= 648) { std::vector<float> out = std::move(demod_soft_); demod_soft_.clear(); return out; }
        std::vector<float> out(demod_soft_.begin(), demod_soft_.begin() + 648);
        demod_soft_.erase(demod_soft_.begin(), demod_soft_.begin() + 648);
        return out;
    }
    uint32_t symbolSamples() const { return geo_.symbol_samples; }
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
    detail::Ctx ctx_;
    ultra_hip_geometry geo_{};
#ifdef ULTRA_HIP_WITH_REFERENCE
    std::unique_ptr<ultra::OFDMModulator> modulator_;
#endif
    std::unique_ptr<detail::DevBuf> d_small_, d_rx_, d_llr_;
    size_t rx_cap_ = 0, llr_cap_ = 0;
    std::vector<float> rx_;                  // rx_buffer = samples [origin_, fed_)
    uint32_t origin_ = 0, fed_ = 0, d_origin_ = 0;   // d_rx_[0] holds sample d_origin_ <= origin_
    uint32_t noise_floor_bits_ = 0, last_sync_offset_ = 0, synced_symbols_ = 0;
    int idle_calls_ = 0;
    bool synced_ = false, pending_cfo_ = false;
    float cfo_hz_ = 0.0f, coarse_cfo_ = 0.0f, freq_offset_hz_ = 0.0f;
    std::vector<float> demod_soft_, soft_bits_;
    float state_[ULTRA_HIP_STATE_FLOATS] = {0, 0, 1, 0, 0, 0, 0, 0};
};

}  // namespace ultra_hip
