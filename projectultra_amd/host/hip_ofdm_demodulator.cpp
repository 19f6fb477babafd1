// hip_ofdm_demodulator.cpp — link-time replacement of ultra::OFDMDemodulator by the MI355X path.
//
// ultra::OFDMDemodulator is a pimpl class (include/ultra/ofdm.hpp:58-127: `struct Impl; std::unique_ptr<Impl> impl_;`), and every
// caller in the reference constructs it directly — the Monte-Carlo tools (tools/test_nvis_mode.cpp:44, tools/test_mode_snr.cpp:34-41,
// tools/test_otfs_vs_ofdm.cpp:109-119), the legacy Modem (src/modem/modem.cpp:79-112), ModemEngine
// (src/gui/modem/modem_engine.hpp:192-193), and the two OFDM waveforms (src/waveform/ofdm_cox_waveform.cpp:25-28,
// ofdm_chirp_waveform.cpp:33-37).  This translation unit DEFINES that class: Impl is ultra_hip::HipOfdmDemodulator
// (include/ultra_hip_waveform.hpp — the process() / processPresynced() state machine over the C-ABI of libultra_hip.so), and
// every public member forwards to it.  Compile it inside the reference tree
//
//     g++ -std=c++20 -DULTRA_HIP_WITH_REFERENCE -I<ref>/include -I<ref>/src -I<repo>/include -c hip_ofdm_demodulator.cpp
//
// and link it INSTEAD OF src/ofdm/demodulator.cpp, src/ofdm/channel_equalizer.cpp and src/ofdm/ofdm_sync.cpp (the three files
// that define the reference's Impl), with -lultra_hip.  No caller changes: tools, waveforms, RxPipeline and ModemEngine run on
// the GPU as they are — INTEGRATION.md 0 / 0b; tests/test_gpu_pimpl.py and tests/test_gpu_ref_programs.py run 34 of the
// reference's own programs built that way (ModemEngine's among them: tools/test_iwaveform.cpp, test_modem_engine_loopback.cpp,
// cli_simulator.cpp, threaded_simulator.cpp, and a scripted harness with feeder / GUI-poll / mode-change threads)
// and compare their output with the reference build's.
//
// ultra::ChannelEstimator — a small stand-alone host class that shares demodulator.cpp (:1019-1066) and the header with the
// demodulator — is defined here as well so that the replaced file leaves no undefined symbol behind; it is host arithmetic
// on a handful of pilots and nothing in the reference constructs it.
//
// Device: ULTRA_HIP_DEVICE in the environment (default 0) — the reference's constructor has no such argument.
#include <cstdlib>

#include "ultra/ofdm.hpp"
#include "ultra_hip_waveform.hpp"

namespace ultra {

namespace {
int hip_device() {
    const char* e = std::getenv("ULTRA_HIP_DEVICE");
    return (e && *e) ? std::atoi(e) : 0;
}
}  // namespace

struct OFDMDemodulator::Impl {
    ultra_hip::HipOfdmDemodulator d;
    explicit Impl(const ModemConfig& config) : d(config, hip_device()) {}
};

OFDMDemodulator::OFDMDemodulator(const ModemConfig& config) : impl_(std::make_unique<Impl>(config)) {}
OFDMDemodulator::~OFDMDemodulator() = default;

// No exception leaves the class (the reference's members do not throw, and its callers run them on threads without a handler):
// a failing C-ABI call is reported on stderr and becomes the member's failure value (ultra_hip::detail::guarded).
using ultra_hip::detail::guarded;
using ultra_hip::detail::guarded_void;

bool OFDMDemodulator::process(SampleSpan samples) {
    return guarded<bool>("OFDMDemodulator::process", false, [&] { return impl_->d.process(samples); });
}
bool OFDMDemodulator::processPresynced(SampleSpan samples, int training_symbols) {
    return guarded<bool>("OFDMDemodulator::processPresynced", false, [&] { return impl_->d.processPresynced(samples, training_symbols); });
}
Bytes OFDMDemodulator::getData() { return impl_->d.getData(); }
std::vector<float> OFDMDemodulator::getSoftBits() { return impl_->d.getSoftBits(); }

ChannelQuality OFDMDemodulator::getChannelQuality() const {
    const ultra_hip::HipChannelQuality q = impl_->d.getChannelQuality();
    ChannelQuality out;
    out.snr_db = q.snr_db; out.doppler_hz = q.doppler_hz; out.delay_spread_ms = q.delay_spread_ms; out.ber_estimate = q.ber_estimate;
    return out;
}
float OFDMDemodulator::getEstimatedSNR() const { return impl_->d.getEstimatedSNR(); }
float OFDMDemodulator::getFrequencyOffset() const { return impl_->d.getFrequencyOffset(); }
void OFDMDemodulator::setFrequencyOffset(float cfo_hz) {
    guarded_void("OFDMDemodulator::setFrequencyOffset", [&] { impl_->d.setFrequencyOffset(cfo_hz); });
}
void OFDMDemodulator::setFrequencyOffsetWithPhase(float cfo_hz, float initial_phase_rad) {
    guarded_void("OFDMDemodulator::setFrequencyOffsetWithPhase", [&] { impl_->d.setFrequencyOffsetWithPhase(cfo_hz, initial_phase_rad); });
}
Symbol OFDMDemodulator::getConstellationSymbols() const { return impl_->d.getConstellationSymbols(); }
bool OFDMDemodulator::isSynced() const { return impl_->d.isSynced(); }
bool OFDMDemodulator::hasPendingData() const { return impl_->d.hasPendingData(); }
size_t OFDMDemodulator::getLastSyncOffset() const { return impl_->d.getLastSyncOffset(); }
void OFDMDemodulator::setTimingOffset(int offset) { impl_->d.setTimingOffset(offset); }
void OFDMDemodulator::reset() { guarded_void("OFDMDemodulator::reset", [&] { impl_->d.reset(); }); }

// ---------------------------------------------------------------------------------------------
// ultra::ChannelEstimator (include/ultra/ofdm.hpp:134-156): per-index one-tap estimate, h <- (h + rx / expected) / 2 where the
// expected pilot has energy, divide-through equalisation where the estimate has.
struct ChannelEstimator::Impl {
    std::vector<Complex> h;
    ChannelQuality quality{};
    explicit Impl(size_t n) : h(n, Complex(1.0f, 0.0f)) {}
    static bool usable(const Complex& v) { return std::norm(v) > 1e-10f; }
};

ChannelEstimator::ChannelEstimator(const ModemConfig& config) : impl_(std::make_unique<Impl>(config.fft_size)) {}
ChannelEstimator::~ChannelEstimator() = default;

void ChannelEstimator::updateFromPilots(const Symbol& received_pilots, const Symbol& expected_pilots) {
    const size_t n = std::min(std::min(received_pilots.size(), expected_pilots.size()), impl_->h.size());
    for (size_t k = 0; k < n; ++k) {
        if (!Impl::usable(expected_pilots[k])) continue;
        const Complex ls = received_pilots[k] / expected_pilots[k];
        impl_->h[k] = 0.5f * ls + 0.5f * impl_->h[k];
    }
}

Symbol ChannelEstimator::equalize(const Symbol& received) {
    Symbol out(received);
    const size_t n = std::min(out.size(), impl_->h.size());
    for (size_t k = 0; k < n; ++k)
        if (Impl::usable(impl_->h[k])) out[k] = received[k] / impl_->h[k];
    return out;
}

ChannelQuality ChannelEstimator::getQuality() const { return impl_->quality; }
void ChannelEstimator::interpolate() {}                                // the reference's is empty too (demodulator.cpp:1064-1066)

}  // namespace ultra
