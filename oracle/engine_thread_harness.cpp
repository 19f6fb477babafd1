// engine_thread_harness.cpp — TEST INFRASTRUCTURE (never product): the usage pattern of the reference's gui::ModemEngine that
// no Monte-Carlo tool has — receive objects created on one thread and driven from others, two engines alive at once, a GUI
// thread polling the demodulator while audio arrives — scripted so that WHAT is decoded is deterministic although WHEN is not.
//
// /root/reference/src/gui/modem/modem_engine.hpp:183-206 (the engine owns an OFDMDemodulator, an LDPCDecoder, three IWaveforms
// from WaveformFactory::create and an RxPipeline), modem_rx.cpp:18-36,153-256 (acquisition thread + decode thread),
// modem_rx.cpp:262-285 (feedAudio: connected -> RxPipeline on the CALLER's thread), modem_engine.cpp:812-827 (isSynced /
// getChannelQuality / getConstellationSymbols read the demodulator from the GUI thread), modem_mode.cpp:133-240 (setConnected /
// setDataMode re-create the demodulator from whichever thread calls them).
//
// One source; oracle/Makefile links it .ref (the reference throughout), .pimpl (the reference's factory and waveforms over the
// link-time drop-ins) and .hip (the product's factory too).  stdout must be identical across the three
// (tests/test_gpu_ref_programs.py): every delivered frame in hex in order of arrival, counts, and the final demodulator-side
// answers as bit patterns; so must the RX engine's RxPipeline log lines on stderr (timestamps cut).  Nothing that depends on
// thread timing is printed to stdout.
// What the reference's engine can and cannot receive is the reference's business: its connected-mode OFDM_COX waveform is built
// once from the constructor's configuration (modem_engine.cpp:92: 64QAM, pilots off) and never configure()d, so data frames in
// another mode synchronise, demodulate and fail to decode — on every build alike, which is what the comparison checks.
//
//   engine_thread_harness <scenario> <seed> [snr_db]
//     cox     connected OFDM_COX: 16QAM R3/4 frames, then (from a third thread) DQPSK R1/2, then QPSK R2/3
//     chirp   connected OFDM_CHIRP DQPSK R1/2 frames, then D8PSK R2/3
//     dpsk    disconnected: MC-DPSK CONNECT frames through the acquisition thread and the decode thread (LDPC on the decode thread)
//     all     the three in sequence on the SAME two engines
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "gui/modem/modem_engine.hpp"
#include "protocol/frame_v2.hpp"
#include "ultra/logging.hpp"

using namespace ultra;
using namespace ultra::gui;
namespace v2 = protocol::v2;

namespace {
struct Sink {
    std::mutex m;
    std::vector<Bytes> frames;
    void push(const Bytes& b) { std::lock_guard<std::mutex> l(m); frames.push_back(b); }
    size_t count() { std::lock_guard<std::mutex> l(m); return frames.size(); }
    std::vector<Bytes> take() { std::lock_guard<std::mutex> l(m); auto f = std::move(frames); frames.clear(); return f; }
};

void add_noise(std::vector<float>& s, float snr_db, std::mt19937& rng) {
    double p = 0; size_t n = 0;
    for (float v : s) if (std::fabs(v) > 1e-6f) { p += double(v) * v; ++n; }
    if (n == 0) return;
    const float sigma = std::sqrt(float(p / n) / std::pow(10.0f, snr_db / 10.0f));
    std::normal_distribution<float> g(0.0f, sigma);
    for (float& v : s) v += g(rng);
}

// [lead silence][frame][gap][frame]...[tail], noise over all of it
std::vector<float> stream_of(ModemEngine& tx, const std::vector<Bytes>& frames, float snr_db, std::mt19937& rng, size_t lead, size_t gap, size_t tail) {
    std::vector<float> audio(lead, 0.0f);
    for (const Bytes& f : frames) {
        const std::vector<float> a = tx.transmit(f);
        audio.insert(audio.end(), a.begin(), a.end());
        audio.resize(audio.size() + gap, 0.0f);
    }
    audio.resize(audio.size() + tail, 0.0f);
    add_noise(audio, snr_db, rng);
    return audio;
}

// the audio arrives on a FEEDER thread in 960-sample chunks while this thread polls the engine the way the GUI does
void feed_while_polling(ModemEngine& rx, const std::vector<float>& audio, size_t chunk) {
    std::atomic<bool> done{false};
    std::thread feeder([&] {
        for (size_t at = 0; at < audio.size(); at += chunk) rx.feedAudio(audio.data() + at, std::min(chunk, audio.size() - at));
        done = true;
    });
    unsigned long long polls = 0; size_t most = 0;
    while (!done) {
        (void)rx.isSynced(); (void)rx.getCurrentSNR(); (void)rx.getChannelQuality(); (void)rx.getStats();
        const auto c = rx.getConstellationSymbols();
        most = std::max(most, c.size());
        ++polls;
        std::this_thread::sleep_for(std::chrono::microseconds(300));
    }
    feeder.join();
    std::fprintf(stderr, "[harness] %llu GUI polls while feeding, constellation up to %zu points\n", polls, most);
}

void wait_for(Sink& sink, size_t want, int max_ms) {
    for (int waited = 0; waited < max_ms && sink.count() < want; waited += 20) std::this_thread::sleep_for(std::chrono::milliseconds(20));
}

void report(const char* what, Sink& sink, size_t sent) {
    const auto frames = sink.take();
    std::printf("%s: sent %zu, delivered %zu\n", what, sent, frames.size());
    for (const Bytes& f : frames) {
        std::printf("  frame %zu bytes:", f.size());
        for (uint8_t b : f) std::printf(" %02x", b);
        std::printf("\n");
    }
}

std::vector<Bytes> data_frames(int n, int first_seq, CodeRate rate, const char* tag) {
    std::vector<Bytes> out;
    for (int i = 0; i < n; ++i) {
        const std::string payload = std::string(tag) + " payload number " + std::to_string(first_seq + i) + std::string(size_t(7 * i), char('a' + i));
        out.push_back(v2::DataFrame::makeData("ALPHA", "BRAVO", uint16_t(first_seq + i), payload, rate).serialize());
    }
    return out;
}

void connected_round(ModemEngine& tx, ModemEngine& rx, Sink& sink, protocol::WaveformMode wf, Modulation mod, CodeRate rate, int n, int seq,
                     float snr_db, std::mt19937& rng, const char* tag, bool from_other_thread) {
    auto configure = [&] {
        for (ModemEngine* e : {&tx, &rx}) {
            // the order the protocol layer produces (tools/cli_simulator.cpp:286-302): the pipeline takes its code rate when the
            // engine ENTERS the connected state (modem_mode.cpp:160-165), not from a later setDataMode
            e->setConnected(false);                          // re-creates the demodulator (DQPSK R1/4), clears the pipeline
            e->setDataMode(mod, rate);
            e->setWaveformMode(wf);
            e->setConnected(true);                           // re-creates modulator, demodulator; decoder_->setRate; switchRxWaveform
            e->setHandshakeComplete(true);
        }
    };
    if (from_other_thread) { std::thread t(configure); t.join(); } else configure();
    const auto frames = data_frames(n, seq, rate, tag);
    const bool chirp = (wf == protocol::WaveformMode::OFDM_CHIRP);
    const auto audio = stream_of(tx, frames, snr_db, rng, 9600, chirp ? 48000 : 14400, 48000);
    feed_while_polling(rx, audio, 960);
    wait_for(sink, frames.size(), 2000);
    report(tag, sink, frames.size());
    std::printf("%s: engine demodulator synced %d\n", tag, rx.isSynced() ? 1 : 0);   // (its quality report is uninitialised memory in the reference until a symbol is demodulated)
}

void dpsk_round(ModemEngine& tx, ModemEngine& rx, Sink& sink, int n, float snr_db, std::mt19937& rng) {
    for (ModemEngine* e : {&tx, &rx}) {
        e->setConnected(false);
        e->setWaveformMode(protocol::WaveformMode::MC_DPSK);
        e->setConnectWaveform(protocol::WaveformMode::MC_DPSK);
        e->setInterleavingEnabled(false);
    }
    (void)tx.transmit(Bytes{0x00});                                   // consumes use_connected_waveform_once_ a disconnect leaves behind
    size_t delivered = 0;
    for (int i = 0; i < n; ++i) {
        v2::ConnectFrame f = v2::ConnectFrame::makeConnect("CALL" + std::to_string(i), "DEST", protocol::ModeCapabilities::ALL,
                                                           static_cast<uint8_t>(protocol::WaveformMode::MC_DPSK));
        f.seq = uint16_t(100 + i);
        // one frame at a time and in ONE feedAudio call: the acquisition thread takes a snapshot of the buffer whenever it wakes, so
        // a frame arriving in chunks would make it a matter of timing whether a snapshot holds the chirp without its data (which
        // the acquisition loop takes for a PING, modem_rx.cpp:84-141)
        const auto audio = stream_of(tx, {f.serialize()}, snr_db, rng, 24000, 0, 72000);
        feed_while_polling(rx, audio, audio.size());
        wait_for(sink, delivered + 1, 20000);
        delivered = sink.count();
        rx.reset();
    }
    report("dpsk", sink, size_t(n));
}
}  // namespace

int main(int argc, char** argv) {
    if (argc < 3) { std::fprintf(stderr, "usage: %s cox|chirp|dpsk|all seed [snr_db]\n", argv[0]); return 2; }
    setLogLevel(LogLevel::INFO);                                    // the pipeline's decisions ("Sync detected at N, CFO=x Hz, corr=y", decode results) go to stderr
    const std::string sc = argv[1];
    const uint32_t seed = uint32_t(std::strtoul(argv[2], nullptr, 10));
    const float snr_db = argc > 3 ? float(std::atof(argv[3])) : 28.0f;
    std::mt19937 rng(seed);

    ModemEngine tx, rx;                                             // two engines alive: 2 demodulators + 2 decoders + 4 OFDM waveforms + 2 pipelines
    tx.setLogPrefix("TX"); rx.setLogPrefix("RX");
    tx.setFilterEnabled(false); rx.setFilterEnabled(false);
    Sink sink;
    rx.setRawDataCallback([&](const Bytes& b) { sink.push(b); });
    std::this_thread::sleep_for(std::chrono::milliseconds(100));   // let the four engine threads start

    if (sc == "cox" || sc == "all") {
        connected_round(tx, rx, sink, protocol::WaveformMode::OFDM_COX, Modulation::QAM16, CodeRate::R3_4, 4, 1, snr_db, rng, "cox-16qam-r34", false);
        connected_round(tx, rx, sink, protocol::WaveformMode::OFDM_COX, Modulation::DQPSK, CodeRate::R1_2, 3, 10, snr_db - 8.0f, rng, "cox-dqpsk-r12", true);
        connected_round(tx, rx, sink, protocol::WaveformMode::OFDM_COX, Modulation::QPSK, CodeRate::R2_3, 3, 20, snr_db - 6.0f, rng, "cox-qpsk-r23", true);
    }
    if (sc == "chirp" || sc == "all") {
        connected_round(tx, rx, sink, protocol::WaveformMode::OFDM_CHIRP, Modulation::DQPSK, CodeRate::R1_2, 3, 30, snr_db - 8.0f, rng, "chirp-dqpsk-r12", false);
        connected_round(tx, rx, sink, protocol::WaveformMode::OFDM_CHIRP, Modulation::D8PSK, CodeRate::R2_3, 2, 40, snr_db - 4.0f, rng, "chirp-d8psk-r23", true);
    }
    if (sc == "dpsk" || sc == "all") dpsk_round(tx, rx, sink, 2, 12.0f, rng);

    const LoopbackStats st = rx.getStats();
    std::printf("final: frames_received %d frames_failed %d\n", int(st.frames_received), int(st.frames_failed));
    return 0;
}
