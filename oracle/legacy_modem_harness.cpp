// legacy_modem_harness.cpp — TEST INFRASTRUCTURE (never product): the reference's legacy facade ultra::Modem
// (/root/reference/include/ultra/modem.hpp:20-88, src/modem/modem.cpp:79-112 constructs an OFDMDemodulator + an LDPCDecoder,
// :153-194 drives them: process -> getSoftBits -> Interleaver(32,32)::deinterleave -> decodeSoft -> lastDecodeSuccess ->
// getChannelQuality -> recommendMode -> decoder->setRate) through its PUBLIC interface only.  One source, linked by
// oracle/Makefile against the compiled reference (.ref) and against the product's link-time drop-ins (.hip); the two outputs
// must be identical (tests/test_gpu_ref_programs.py).  Floats are printed as bit patterns.
//
//   legacy_modem_harness <snr_db> <seed> [fft modulation code_rate]
//
// Station A connects and sends; its transmit audio goes through AWGN into station B in 960-sample chunks; whatever B answers
// goes back into A the same way.  After every chunk that changed a station's statistics a line is printed, so WHEN the
// demodulator delivered each codeword, what the decoder made of it and what the rate adaptation did with the reported
// quality are all part of the comparison.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "ultra/logging.hpp"
#include "ultra/modem.hpp"
#include "ultra/types.hpp"

using namespace ultra;

namespace {
uint32_t bits(float f) { uint32_t u; std::memcpy(&u, &f, sizeof(u)); return u; }

Samples drain(Modem& m) {
    Samples all, buf(4096);
    while (m.txPending()) {
        const size_t n = m.getTxSamples(MutableSampleSpan(buf.data(), buf.size()));
        if (n == 0) break;
        all.insert(all.end(), buf.begin(), buf.begin() + n);
    }
    return all;
}
void add_noise(Samples& s, float snr_db, std::mt19937& rng) {
    double p = 0; for (float v : s) p += double(v) * v;
    if (s.empty() || p == 0) return;
    const float sigma = std::sqrt(float(p / s.size()) / std::pow(10.0f, snr_db / 10.0f));
    std::normal_distribution<float> n(0.0f, sigma);
    for (float& v : s) v += n(rng);
}
void print_stats(const char* who, size_t chunk, const ModemStats& st, const Modem& m) {
    const ChannelQuality q = m.getChannelQuality();
    std::printf("%s chunk %zu: rx_frames %llu rx_bytes %llu tx_frames %llu retx %llu snr %08x mod %d rate %d connected %d quality %08x %08x\n", who, chunk,
                (unsigned long long)st.frames_received, (unsigned long long)st.bytes_received, (unsigned long long)st.frames_sent,
                (unsigned long long)st.frames_retransmitted, bits(st.current_snr_db), int(st.current_modulation), int(st.current_code_rate),
                m.isConnected() ? 1 : 0, bits(q.snr_db), bits(q.ber_estimate));
}
// feed `audio` (+ one second of silence) into `rx` in 960-sample chunks
void feed(const char* who, Modem& rx, Samples audio) {
    audio.resize(audio.size() + 48000, 0.0f);
    ModemStats last = rx.getStats();
    for (size_t at = 0, k = 0; at < audio.size(); at += 960, ++k) {
        const size_t n = std::min<size_t>(960, audio.size() - at);
        rx.rxSamples(SampleSpan(audio.data() + at, n));
        const ModemStats st = rx.getStats();
        if (std::memcmp(&st, &last, sizeof(st)) != 0) { print_stats(who, k, st, rx); last = st; }
    }
    print_stats(who, size_t(-1), rx.getStats(), rx);
}
}  // namespace

int main(int argc, char** argv) {
    if (argc < 3) { std::fprintf(stderr, "usage: %s snr_db seed [fft modulation code_rate]\n", argv[0]); return 2; }
    setLogLevel(LogLevel::ERROR);
    const float snr_db = float(std::atof(argv[1]));
    const uint32_t seed = uint32_t(std::strtoul(argv[2], nullptr, 10));
    ModemConfig cfg;
    if (argc > 5) {
        if (std::atoi(argv[3]) == 1024) cfg = presets::nvis_mode();
        cfg.modulation = static_cast<Modulation>(std::atoi(argv[4]));
        cfg.code_rate = static_cast<CodeRate>(std::atoi(argv[5]));
    }
    std::mt19937 rng(seed);
    Modem a(cfg), b(cfg);
    a.start(); b.start();
    size_t delivered = 0;
    b.setDataCallback([&](Bytes d) {
        std::printf("B delivered %zu bytes:", d.size());
        for (size_t i = 0; i < d.size() && i < 48; ++i) std::printf(" %02x", d[i]);
        std::printf("\n");
        ++delivered;
    });
    std::printf("rate A %08x\n", bits(a.getDataRate()));

    a.connect();
    Bytes payload(96);
    for (size_t i = 0; i < payload.size(); ++i) payload[i] = uint8_t(rng());
    a.send(ByteSpan(payload.data(), payload.size()));
    Samples to_b = drain(a);
    std::printf("A -> B: %zu samples\n", to_b.size());
    add_noise(to_b, snr_db, rng);
    feed("B", b, to_b);

    Samples to_a = drain(b);
    std::printf("B -> A: %zu samples\n", to_a.size());
    add_noise(to_a, snr_db, rng);
    feed("A", a, to_a);

    // a second exchange on the same objects with the modes forced (setModulation / setCodeRate: decoder->setRate on a used decoder)
    b.setCodeRate(CodeRate::R1_4); a.setCodeRate(CodeRate::R1_4);
    b.setModulation(Modulation::QPSK); a.setModulation(Modulation::QPSK);
    for (size_t i = 0; i < payload.size(); ++i) payload[i] = uint8_t(rng());
    a.send(ByteSpan(payload.data(), 40));
    to_b = drain(a);
    std::printf("A -> B (forced QPSK R1/4): %zu samples\n", to_b.size());
    add_noise(to_b, snr_db, rng);
    feed("B", b, to_b);
    a.disconnect();
    to_b = drain(a);
    add_noise(to_b, snr_db, rng);
    feed("B", b, to_b);
    std::printf("delivered %zu, A connected %d, B connected %d\n", delivered, a.isConnected() ? 1 : 0, b.isConnected() ? 1 : 0);
    return 0;
}
