// rx_pipeline_harness.cpp — TEST INFRASTRUCTURE (never product): the reference's own, unmodified caller of the plugin
// surface — ultra::gui::RxPipeline (/root/reference/src/gui/modem/rx_pipeline.hpp:56-73, rx_pipeline.cpp:55-78,104-269) —
// driven over either
//   ref   the reference's waveform (ultra::OFDMNvisWaveform = WaveformFactory's OFDM_COX, ultra::OFDMChirpWaveform =
//         OFDM_CHIRP: /root/reference/src/waveform/waveform_factory.cpp:16-17,52-53), or
//   hip   this repository's adapter over the C-ABI (ultra_hip::HipOfdmCoxWaveform / HipOfdmWaveform,
//         include/ultra_hip_waveform.hpp compiled with -DULTRA_HIP_WITH_REFERENCE so that it IS an ultra::IWaveform)
// held through the raw, non-owning IWaveform* RxPipeline keeps (rx_pipeline.hpp:65-68).  A recording decorator between the
// pipeline and the waveform writes every call the pipeline makes and everything the waveform answers (floats as bit
// patterns); the frame and ping callbacks, the frame queue and the buffer size after every feedAudio() go to the same log.
// tests/test_gpu_rx_pipeline.py runs it twice on the same audio and requires the two logs to be IDENTICAL.
//
// Built by oracle/Makefile from the reference's sources where they lie (nothing is copied); the binary lives in
// oracle/_ref/ (git-ignored, shipped to the GPU box like libultra_ref.so).
//
//   rx_pipeline_harness <cox|chirp> <ref|hip> audio.f32 out.log fft carriers cp_mode guard pilot_spacing use_pilots
//                       modulation code_rate connected interleave_bps chunk
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "ultra_hip_waveform.hpp"                 // -DULTRA_HIP_WITH_REFERENCE: derives from ultra::IWaveform
#include "gui/modem/rx_pipeline.hpp"
#include "waveform/ofdm_chirp_waveform.hpp"
#include "waveform/ofdm_cox_waveform.hpp"

namespace {

unsigned bits(float v) { unsigned u; std::memcpy(&u, &v, 4); return u; }

struct Recorder final : ultra::IWaveform {         // forwards everything, records the RX half
    ultra::IWaveform* in;
    FILE* log;
    Recorder(ultra::IWaveform* w, FILE* f) : in(w), log(f) {}
    std::string getName() const override { return in->getName(); }
    ultra::protocol::WaveformMode getMode() const override { return in->getMode(); }
    ultra::WaveformCapabilities getCapabilities() const override { return in->getCapabilities(); }
    void configure(ultra::Modulation m, ultra::CodeRate r) override { std::fprintf(log, "configure %d %d\n", (int)m, (int)r); in->configure(m, r); }
    void setFrequencyOffset(float c) override { std::fprintf(log, "setFrequencyOffset %08x\n", bits(c)); in->setFrequencyOffset(c); }
    void setTxFrequencyOffset(float c) override { in->setTxFrequencyOffset(c); }
    ultra::Modulation getModulation() const override { return in->getModulation(); }
    ultra::CodeRate getCodeRate() const override { return in->getCodeRate(); }
    float getFrequencyOffset() const override { return in->getFrequencyOffset(); }
    ultra::Samples generatePreamble() override { return in->generatePreamble(); }
    ultra::Samples modulate(const ultra::Bytes& e) override { return in->modulate(e); }
    bool detectSync(ultra::SampleSpan s, ultra::SyncResult& r, float thr) override {
        const bool ok = in->detectSync(s, r, thr);
        std::fprintf(log, "detectSync n=%zu thr=%08x -> %d detected=%d start=%d corr=%08x cfo=%08x training=%d synced=%d\n", s.size(), bits(thr),
                     (int)ok, (int)r.detected, r.start_sample, bits(r.correlation), bits(r.cfo_hz), (int)r.has_training, (int)in->isSynced());
        return ok;
    }
    bool process(ultra::SampleSpan s) override {
        const bool ok = in->process(s);
        std::fprintf(log, "process n=%zu -> %d synced=%d hasData=%d\n", s.size(), (int)ok, (int)in->isSynced(), (int)in->hasData());
        return ok;
    }
    std::vector<float> getSoftBits() override {
        std::vector<float> v = in->getSoftBits();
        std::fprintf(log, "getSoftBits -> %zu:", v.size());
        for (float x : v) std::fprintf(log, " %08x", bits(x));
        std::fprintf(log, "\n");
        return v;
    }
    void reset() override { std::fprintf(log, "reset\n"); in->reset(); }
    bool isSynced() const override { return in->isSynced(); }
    bool hasData() const override { return in->hasData(); }
    float estimatedSNR() const override { const float v = in->estimatedSNR(); std::fprintf(log, "estimatedSNR -> %08x\n", bits(v)); return v; }
    float estimatedCFO() const override { const float v = in->estimatedCFO(); std::fprintf(log, "estimatedCFO -> %08x\n", bits(v)); return v; }
    std::vector<std::complex<float>> getConstellationSymbols() const override { return in->getConstellationSymbols(); }
    std::string getStatusString() const override { return in->getStatusString(); }
    int getCarrierCount() const override { return in->getCarrierCount(); }
    float getThroughput(ultra::CodeRate r) const override { return in->getThroughput(r); }
    int getSamplesPerSymbol() const override { const int v = in->getSamplesPerSymbol(); std::fprintf(log, "getSamplesPerSymbol -> %d\n", v); return v; }
    int getPreambleSamples() const override { const int v = in->getPreambleSamples(); std::fprintf(log, "getPreambleSamples -> %d\n", v); return v; }
    int getMinSamplesForFrame() const override { const int v = in->getMinSamplesForFrame(); std::fprintf(log, "getMinSamplesForFrame -> %d\n", v); return v; }
};

void log_bytes(FILE* log, const ultra::Bytes& b) {
    for (unsigned char c : b) std::fprintf(log, "%02x", c);
}

}  // namespace

int main(int argc, char** argv) {
    if (argc < 16) { std::fprintf(stderr, "usage: see the head of oracle/rx_pipeline_harness.cpp\n"); return 64; }
    const std::string kind = argv[1], impl = argv[2];
    std::vector<float> audio;
    {
        FILE* f = std::fopen(argv[3], "rb");
        if (!f) return 65;
        std::fseek(f, 0, SEEK_END); const long bytes = std::ftell(f); std::fseek(f, 0, SEEK_SET);
        audio.resize((size_t)bytes / 4);
        if (std::fread(audio.data(), 4, audio.size(), f) != audio.size()) return 66;
        std::fclose(f);
    }
    FILE* log = std::fopen(argv[4], "w");
    if (!log) return 67;
    ultra::ModemConfig c;
    c.fft_size = std::stoul(argv[5]); c.num_carriers = std::stoul(argv[6]);
    c.cp_mode = static_cast<ultra::CyclicPrefixMode>(std::stoi(argv[7])); c.symbol_guard = std::stoul(argv[8]);
    c.pilot_spacing = std::stoul(argv[9]); c.use_pilots = std::stoi(argv[10]) != 0;
    c.modulation = static_cast<ultra::Modulation>(std::stoi(argv[11])); c.code_rate = static_cast<ultra::CodeRate>(std::stoi(argv[12]));
    const bool connected = std::stoi(argv[13]) != 0;
    const size_t bps = std::stoul(argv[14]), chunk = std::stoul(argv[15]);

    ultra::WaveformPtr wave;                       // the plugin pointer type of the reference (waveform_interface.hpp:160)
    if (kind == "cox" && impl == "ref") wave = std::make_unique<ultra::OFDMNvisWaveform>(c);
    else if (kind == "cox" && impl == "hip") wave = std::make_unique<ultra_hip::HipOfdmCoxWaveform>(c);
    else if (kind == "chirp" && impl == "ref") wave = std::make_unique<ultra::OFDMChirpWaveform>(c);
    else if (kind == "chirp" && impl == "hip") wave = std::make_unique<ultra_hip::HipOfdmWaveform>(c);
    else return 68;
    std::fprintf(log, "waveform mode=%d carriers=%d modulation=%d rate=%d\n", (int)wave->getMode(), wave->getCarrierCount(),
                 (int)wave->getModulation(), (int)wave->getCodeRate());
    Recorder rec(wave.get(), log);

    int rc = 0;
    {
        ultra::gui::RxPipeline rx("RX");
        rx.setWaveform(&rec);                      // raw, non-owning: the waveform outlives the pipeline (this scope)
        rx.setDataMode(c.code_rate, connected);
        rx.setInterleavingEnabled(bps != 0);
        if (bps) rx.setInterleaverConfig(bps);
        rx.setFrameCallback([&](const ultra::Bytes& data, ultra::protocol::v2::FrameType type) {
            std::fprintf(log, "FRAME_CALLBACK type=%d bytes=%zu ", (int)type, data.size()); log_bytes(log, data); std::fprintf(log, "\n");
        });
        rx.setPingCallback([&](float snr) { std::fprintf(log, "PING_CALLBACK snr=%08x\n", bits(snr)); });
        size_t last_buf = 0;
        for (size_t i = 0; i < audio.size(); i += chunk) {
            const size_t len = std::min(chunk, audio.size() - i);
            rx.feedAudio(audio.data() + i, len);
            const size_t b = rx.getBufferSize();
            if (b != last_buf + len) std::fprintf(log, "buffer after sample %zu: %zu\n", i + len, b);
            last_buf = b;
            while (rx.hasFrame()) {
                const ultra::gui::RxFrameResult r = rx.getFrame();
                std::fprintf(log, "QUEUE success=%d type=%d ok=%d failed=%d snr=%08x cfo=%08x ping=%d bytes=%zu ", (int)r.success, (int)r.frame_type,
                             r.codewords_ok, r.codewords_failed, bits(r.snr_estimate), bits(r.cfo_estimate), (int)r.is_ping, r.frame_data.size());
                log_bytes(log, r.frame_data); std::fprintf(log, "\n");
            }
        }
        std::fprintf(log, "end buffer=%zu accumulating=%d expected=%d\n", rx.getBufferSize(), (int)rx.isAccumulating(), rx.getExpectedCodewords());
    }
    std::fclose(log);
    return rc;
}
