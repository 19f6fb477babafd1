// live_latency.cpp — MEASUREMENT HARNESS (test infrastructure, never product): what the receive path costs at the batch size
// the reference actually runs it at — ONE live stream, one process() call per 960-sample audio chunk (20 ms), the shape of
// /root/reference/src/gui/modem/rx_pipeline.cpp:55-78 and of tools/profile_acquisition.cpp:1-35.  One source, through the two
// pimpl classes' public interface (ultra::OFDMDemodulator, ultra::LDPCDecoder), linked by oracle/Makefile against the compiled
// reference (`.ref`: the CPU column) and against the product's link-time drop-ins (`.hip`: the MI355X column).
//
//   live_latency <fft> <modulation> <code_rate> <frames> [snr_db] [chunk]
//
// Prints, for the calls made while SEARCHING, the calls made while SYNCED and LDPCDecoder::decodeSoft: count, median, p90, p99
// and maximum wall time per call; the whole stream's wall time against its audio duration; frames decoded; and — when linked
// against libultra_hip.so — the library's blocking host-on-device waits per call (ultra_hip_host_sync_count).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "ultra/fec.hpp"
#include "ultra/logging.hpp"
#include "ultra/ofdm.hpp"
#include "ultra/types.hpp"

using namespace ultra;
using Clock = std::chrono::steady_clock;

extern "C" unsigned long long ultra_hip_host_sync_count(void) __attribute__((weak));

namespace {
double us_since(Clock::time_point t0) { return std::chrono::duration<double, std::micro>(Clock::now() - t0).count(); }
struct Stat {
    std::vector<double> v;
    void add(double x) { v.push_back(x); }
    void print(const char* name) {
        if (v.empty()) { std::printf("%-28s n=0\n", name); return; }
        std::sort(v.begin(), v.end());
        auto q = [&](double p) { return v[std::min(v.size() - 1, size_t(p * v.size()))]; };
        double sum = 0; for (double x : v) sum += x;
        std::printf("%-28s n=%-6zu median %8.1f us   p90 %8.1f   p99 %8.1f   max %9.1f   mean %8.1f\n", name, v.size(), q(0.5), q(0.9), q(0.99), v.back(), sum / v.size());
    }
};
size_t info_bytes(CodeRate r) {
    switch (r) { case CodeRate::R1_4: return 20; case CodeRate::R1_3: return 27; case CodeRate::R1_2: return 40; case CodeRate::R2_3: return 54;
                 case CodeRate::R3_4: return 60; case CodeRate::R5_6: return 67; default: return 40; }
}
}  // namespace

int main(int argc, char** argv) {
    if (argc < 5) { std::fprintf(stderr, "usage: %s fft modulation code_rate frames [snr_db] [chunk]\n", argv[0]); return 2; }
    setLogLevel(LogLevel::ERROR);
    const int fft = std::atoi(argv[1]);
    const Modulation mod = static_cast<Modulation>(std::atoi(argv[2]));
    const CodeRate rate = static_cast<CodeRate>(std::atoi(argv[3]));
    const int n_frames = std::atoi(argv[4]);
    const float snr_db = argc > 5 ? float(std::atof(argv[5])) : 30.0f;
    const size_t chunk = argc > 6 ? size_t(std::atoi(argv[6])) : 960;
    ModemConfig c = (fft == 1024) ? presets::nvis_mode() : ModemConfig();
    const bool diff = (mod == Modulation::DQPSK || mod == Modulation::D8PSK || mod == Modulation::DBPSK);
    c.use_pilots = !diff;
    if (fft == 1024 && c.use_pilots) c.pilot_spacing = 4;
    c.modulation = mod; c.code_rate = rate;

    // the stream: [noise][preamble + one codeword]... one transmission every ~1.2 s
    std::mt19937 rng(2024);
    OFDMModulator modulator(c); LDPCEncoder enc(rate);
    std::vector<float> audio;
    std::vector<Bytes> payloads;
    auto noise = [&](size_t n, float sd) { std::normal_distribution<float> d(0.0f, sd); for (size_t i = 0; i < n; ++i) audio.push_back(d(rng)); };
    noise(24000, 0.005f);
    for (int f = 0; f < n_frames; ++f) {
        Bytes p(info_bytes(rate)); for (auto& b : p) b = rng() & 0xFF;
        payloads.push_back(p);
        Bytes coded = enc.encode(p);
        Samples pre = modulator.generatePreamble(), body = modulator.modulate(coded, mod);
        Samples s(pre); s.insert(s.end(), body.begin(), body.end());
        float mx = 0; for (float v : s) mx = std::max(mx, std::abs(v));
        float pw = 0; for (float& v : s) { v *= 0.5f / mx; pw += v * v; } pw /= s.size();
        std::normal_distribution<float> d(0.0f, std::sqrt(pw / std::pow(10.0f, snr_db / 10.0f)));
        for (float v : s) audio.push_back(v + d(rng));
        noise(40000 + (rng() % 20000), 0.005f);
    }

    OFDMDemodulator demod(c); LDPCDecoder dec(rate);
    {   // warm-up outside the clock: the first call builds tables and device contexts
        std::vector<float> z(chunk, 0.0f); demod.process(SampleSpan(z.data(), z.size())); demod.reset();
        std::vector<float> l(648, 1.0f); dec.decodeSoft(l);
    }
    Stat searching, synced, decode;
    const unsigned long long syncs0 = ultra_hip_host_sync_count ? ultra_hip_host_sync_count() : 0ull;
    size_t calls = 0, decoded = 0, next_payload = 0;
    const auto t_all = Clock::now();
    for (size_t i = 0; i < audio.size(); i += chunk) {
        const size_t n = std::min(chunk, audio.size() - i);
        const bool was_synced = demod.isSynced();
        const auto t0 = Clock::now();
        const bool ready = demod.process(SampleSpan(audio.data() + i, n));
        (was_synced ? synced : searching).add(us_since(t0));
        ++calls;
        if (ready) {
            std::vector<float> soft = demod.getSoftBits();
            if (soft.size() >= 648) {
                const auto t1 = Clock::now();
                Bytes out = dec.decodeSoft(std::span<const float>(soft.data(), 648));
                decode.add(us_since(t1));
                if (dec.lastDecodeSuccess() && next_payload < payloads.size() && out.size() >= payloads[next_payload].size() &&
                    std::equal(payloads[next_payload].begin(), payloads[next_payload].end(), out.begin())) ++decoded;
                ++next_payload;
            }
            demod.reset();                                           // the pipeline's pattern: one codeword per frame here, then reset
        }
    }
    const double wall_ms = us_since(t_all) / 1000.0, audio_ms = audio.size() / 48.0;
    std::printf("config: fft %d modulation %d rate %d, %d frames at %.0f dB, %zu-sample chunks (%.1f ms of audio each)\n", fft, (int)mod, (int)rate,
                n_frames, snr_db, chunk, chunk / 48.0);
    searching.print("process() while SEARCHING");
    synced.print("process() while SYNCED");
    decode.print("LDPCDecoder::decodeSoft");
    std::printf("stream: %.1f ms of wall time for %.1f ms of audio (%.3f of real time), %zu calls, frames decoded %zu / %d\n", wall_ms, audio_ms,
                wall_ms / audio_ms, calls, decoded, n_frames);
    if (ultra_hip_host_sync_count)
        std::printf("blocking host-on-device waits: %.2f per process() / decodeSoft() call\n",
                    double(ultra_hip_host_sync_count() - syncs0) / double(calls + decode.v.size()));
    return 0;
}
