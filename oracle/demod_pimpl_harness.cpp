// demod_pimpl_harness.cpp — TEST INFRASTRUCTURE (never product): scripted use of the reference's two pimpl classes,
// ultra::OFDMDemodulator (/root/reference/include/ultra/ofdm.hpp:58-127) and ultra::LDPCDecoder (include/ultra/fec.hpp:48-77),
// and of the interleavers that share their header, through their PUBLIC interface only.  oracle/Makefile links this one source
// twice: against the compiled reference (`.ref`) and against the product's link-time drop-ins (`.hip`:
// projectultra_amd/host/hip_ofdm_demodulator.cpp + hip_ldpc_decoder.cpp over libultra_hip.so).  Every answer of every call goes
// to stdout with floats as bit patterns; tests/test_gpu_pimpl.py requires the two outputs to be identical.
//
// What the reference's own tools (tests/test_gpu_pimpl.py, first half) do not reach and this does:
//   carry      several frames through ONE demodulator without reset() (the legacy Modem's pattern, src/modem/modem.cpp:153-166):
//              the SEARCHING -> SYNCED transition carries channel estimate, noise, SNR, pilot history into the next frame
//   timing     setTimingOffset(+-n) (demodulator.cpp:572)
//   setcfo     setFrequencyOffset / setFrequencyOffsetWithPhase before sync, between sync and first symbol, mid-frame
//   presynced  processPresynced with / without a preset offset, 0 / 1 / 2 / 3 training symbols, the rest of the frame arriving
//              through process() afterwards, a second frame on the same object (timing offset survives), reset() in between
//   midframe   a new preamble while SYNCED (:605-657)
//   exits      sync timeout (250 symbols), idle timeout, "frame complete"
//   getdata    getData(), getChannelQuality(), hasPendingData()
//   decoder    decodeSoft with 1..2000 soft bits, decode(bytes), setRate, setMaxIterations(0, 1, 7), NaN / inf LLRs
//   interleave Interleaver / ChannelInterleaver, bytes and soft bits, short and long inputs
//
//   demod_pimpl_harness <scenario> <fft> <modulation> <code_rate> [seed]
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "ultra/dsp.hpp"
#include "ultra/fec.hpp"
#include "ultra/logging.hpp"
#include "ultra/ofdm.hpp"
#include "ultra/types.hpp"

using namespace ultra;

namespace {

unsigned bits(float v) { unsigned u; std::memcpy(&u, &v, 4); return u; }

ModemConfig make_config(int fft, Modulation mod, CodeRate rate) {
    ModemConfig c = (fft == 1024) ? presets::nvis_mode() : ModemConfig();
    const bool diff = (mod == Modulation::DQPSK || mod == Modulation::D8PSK || mod == Modulation::DBPSK);
    c.use_pilots = !diff;
    if (fft == 1024 && c.use_pilots) c.pilot_spacing = 4;
    c.modulation = mod; c.code_rate = rate;
    return c;
}
size_t info_bytes(CodeRate r) {
    switch (r) { case CodeRate::R1_4: return 20; case CodeRate::R1_3: return 27; case CodeRate::R1_2: return 40; case CodeRate::R2_3: return 54;
                 case CodeRate::R3_4: return 60; case CodeRate::R5_6: return 67; default: return 40; }
}

struct Tx {
    ModemConfig cfg; OFDMModulator mod; LDPCEncoder enc; std::mt19937 rng;
    Tx(const ModemConfig& c, unsigned seed) : cfg(c), mod(c), enc(c.code_rate), rng(seed) {}
    Samples noise(size_t n, float sigma) { std::normal_distribution<float> d(0.0f, sigma); Samples s(n); for (auto& v : s) v = d(rng); return s; }
    // [preamble | n_cw codewords of random payload], peak 0.5, + AWGN at snr_db, shifted by cfo_hz (Hilbert + rotate)
    Samples frame(int n_cw, float snr_db, float cfo_hz, bool presynced = false, int training = 2) {
        Bytes payload(info_bytes(cfg.code_rate) * n_cw);
        for (auto& b : payload) b = rng() & 0xFF;
        Bytes coded = enc.encode(payload);
        Samples pre = presynced ? mod.generateTrainingSymbols(training) : mod.generatePreamble();
        Samples body = mod.modulate(coded, cfg.modulation);
        Samples s(pre); s.insert(s.end(), body.begin(), body.end());
        float mx = 0; for (float v : s) mx = std::max(mx, std::abs(v));
        for (float& v : s) v *= 0.5f / mx;
        if (std::abs(cfo_hz) > 0.001f) {
            HilbertTransform h(255);
            auto a = h.process(SampleSpan(s.data(), s.size()));
            float ph = 0.0f; const float inc = 2.0f * float(M_PI) * cfo_hz / 48000.0f;
            for (size_t i = 0; i < s.size(); ++i) {
                s[i] = std::real(a[i] * Complex(std::cos(ph), std::sin(ph)));
                ph += inc; while (ph > float(M_PI)) ph -= 2.0f * float(M_PI);
            }
        }
        float p = 0; for (float v : s) p += v * v; p /= s.size();
        const float sd = std::sqrt(p / std::pow(10.0f, snr_db / 10.0f));
        std::normal_distribution<float> d(0.0f, sd);
        for (float& v : s) v += d(rng);
        return s;
    }
};

struct Rx {                                     // logs every answer of the demodulator
    OFDMDemodulator d; LDPCDecoder dec; int call = 0; bool demodulated = false;
    Rx(const ModemConfig& c) : d(c), dec(c.code_rate) {}
    void status(const char* what, bool r) {
        // Impl::quality is uninitialised memory in the reference until the first symbol has been demodulated (updateQuality)
        demodulated = demodulated || r || d.hasPendingData();
        const ChannelQuality q = d.getChannelQuality();
        // getConstellationSymbols() (demodulator.cpp:827-830): the ring's size and an FNV-1a digest of its bit patterns
        const Symbol ring = d.getConstellationSymbols();
        unsigned long long fnv = 1469598103934665603ull;
        for (const Complex& z : ring)
            for (float v : {z.real(), z.imag()}) { fnv ^= bits(v); fnv *= 1099511628211ull; }
        std::printf("%s#%d -> %d synced=%d pending=%d snr=%08x cfo=%08x sync_off=%zu q=%08x,%08x ring=%zu:%016llx\n", what, call++, (int)r,
                    (int)d.isSynced(), (int)d.hasPendingData(), bits(d.getEstimatedSNR()), bits(d.getFrequencyOffset()), d.getLastSyncOffset(),
                    demodulated ? bits(q.snr_db) : 0u, demodulated ? bits(q.ber_estimate) : 0u, ring.size(), fnv);
    }
    void drain(bool decode = true) {
        std::vector<float> s = d.getSoftBits();
        std::printf("  soft %zu:", s.size());
        for (float v : s) std::printf(" %08x", bits(v));
        std::printf("\n");
        if (decode && s.size() >= 648) {
            Bytes out = dec.decodeSoft(std::span<const float>(s.data(), 648));
            std::printf("  decoded ok=%d iters=%d n=%zu:", (int)dec.lastDecodeSuccess(), dec.lastIterations(), out.size());
            for (uint8_t b : out) std::printf(" %02x", b);
            std::printf("\n");
        }
    }
    void feed(const Samples& audio, size_t chunk = 960, bool drain_ready = true) {
        for (size_t i = 0; i < audio.size(); i += chunk) {
            const size_t n = std::min(chunk, audio.size() - i);
            const bool r = d.process(SampleSpan(audio.data() + i, n));
            status("process", r);
            if (r && drain_ready) drain();
        }
    }
    void idle(int n) { for (int i = 0; i < n; ++i) { const bool r = d.process(SampleSpan()); status("empty", r); if (r) drain(); } }
};

void append(Samples& a, const Samples& b) { a.insert(a.end(), b.begin(), b.end()); }

// ---- scenarios ------------------------------------------------------------------------------
void carry(const ModemConfig& c, unsigned seed) {
    Tx tx(c, seed); Rx rx(c);
    Samples audio = tx.noise(9000, 0.01f);
    const float snr[4] = {30.0f, 22.0f, 26.0f, 18.0f}, cfo[4] = {0.0f, 6.5f, -14.0f, 2.0f};
    for (int f = 0; f < 4; ++f) { append(audio, tx.frame(1 + f % 2, snr[f], cfo[f])); append(audio, tx.noise(20000 + 3000 * f, 0.01f)); }
    rx.feed(audio);                                            // NO reset() between the frames
    rx.idle(3);
    rx.d.reset(); std::printf("reset\n");
    rx.feed(tx.noise(5000, 0.01f)); rx.feed(tx.frame(1, 28.0f, 3.0f)); rx.idle(14);
}

void timing(const ModemConfig& c, unsigned seed) {
    for (int off : {0, 7, -5}) {
        Tx tx(c, seed); Rx rx(c);
        rx.d.setTimingOffset(off); std::printf("setTimingOffset %d\n", off);
        Samples audio = tx.noise(7000, 0.01f); append(audio, tx.frame(1, 30.0f, 0.0f)); append(audio, tx.noise(15000, 0.01f));
        rx.feed(audio);
    }
}

void setcfo(const ModemConfig& c, unsigned seed) {
    Tx tx(c, seed); Rx rx(c);
    rx.d.setFrequencyOffset(11.0f); std::printf("setFrequencyOffset 11 (searching)\n"); rx.status("after", false);
    Samples f1 = tx.noise(6000, 0.01f); append(f1, tx.frame(2, 28.0f, 9.0f)); append(f1, tx.noise(9000, 0.01f));
    // feed until synced, then set the offset between symbols of the frame
    size_t i = 0; bool did = false;
    for (; i < f1.size(); i += 960) {
        const size_t n = std::min<size_t>(960, f1.size() - i);
        const bool r = rx.d.process(SampleSpan(f1.data() + i, n));
        rx.status("process", r);
        if (r) rx.drain();
        if (!did && rx.d.isSynced()) { rx.d.setFrequencyOffset(8.25f); std::printf("setFrequencyOffset 8.25 (synced)\n"); rx.status("after", false); did = true; }
    }
    rx.idle(3);
    rx.d.reset(); std::printf("reset\n");
    Samples f2 = tx.frame(1, 25.0f, -20.0f); append(f2, tx.noise(12000, 0.01f));
    did = false;
    for (i = 0; i < f2.size(); i += 960) {
        const size_t n = std::min<size_t>(960, f2.size() - i);
        const bool r = rx.d.process(SampleSpan(f2.data() + i, n));
        rx.status("process", r);
        if (r) rx.drain();
        if (!did && rx.d.isSynced()) { rx.d.setFrequencyOffsetWithPhase(-19.5f, 0.7f); std::printf("setFrequencyOffsetWithPhase -19.5 0.7\n"); did = true; }
    }
}

void presynced(const ModemConfig& c, unsigned seed) {
    Tx tx(c, seed);
    const size_t sym = c.getSymbolDuration();
    {   // preset offset + phase, whole frame in one call; then a second frame on the same object, then reset and a third
        Rx rx(c);
        for (int f = 0; f < 3; ++f) {
            Samples fr = tx.frame(1 + (f == 1), 24.0f - 3 * f, 4.0f * f, true, 2);
            if (f != 1) { rx.d.setFrequencyOffsetWithPhase(4.0f * f, 0.3f * f); std::printf("setFrequencyOffsetWithPhase %d\n", f); }
            const bool r = rx.d.processPresynced(SampleSpan(fr.data(), fr.size()), 2);
            rx.status("presynced", r);
            while (rx.d.hasPendingData()) { rx.drain(); }
            rx.idle(2);
            if (f == 1) { rx.d.reset(); std::printf("reset\n"); rx.status("after", false); }
        }
    }
    {   // never-set offset: estimated from the training symbols; the tail of the frame arrives through process()
        Rx rx(c);
        Samples fr = tx.frame(2, 27.0f, 5.5f, true, 2);
        const size_t head = 5 * sym + 100;
        const bool r = rx.d.processPresynced(SampleSpan(fr.data(), head), 2);
        rx.status("presynced-head", r);
        Samples tail(fr.begin() + head, fr.end()); append(tail, tx.noise(3000, 0.01f));
        rx.feed(tail, 700);
        rx.idle(12);
    }
    for (int tr : {0, 1, 3}) {   // other training counts
        Rx rx(c);
        Samples fr = tx.frame(1, 30.0f, 0.0f, true, tr);
        const bool r = rx.d.processPresynced(SampleSpan(fr.data(), fr.size()), tr);
        std::printf("training=%d\n", tr); rx.status("presynced", r);
        while (rx.d.hasPendingData()) rx.drain(false);
    }
    {   // too short
        Rx rx(c);
        Samples fr = tx.frame(1, 30.0f, 0.0f, true, 2);
        const bool r = rx.d.processPresynced(SampleSpan(fr.data(), sym - 1), 2);
        rx.status("presynced-short", r);
    }
}

// processPresynced and process() on ONE object without reset() in between: the SEARCHING -> SYNCED transition of the Schmidl-Cox
// frame (demodulator.cpp:533-591) carries what the presynced frame's tracker left (channel estimate, noise variance, SNR, pilot
// history, equaliser weights); then a presynced frame again (its reset block, :868-905), then a reset() and a last Schmidl-Cox frame
void mixed(const ModemConfig& c, unsigned seed) {
    Tx tx(c, seed); Rx rx(c);
    for (int round = 0; round < 2; ++round) {
        Samples fr = tx.frame(1, 24.0f - 2 * round, 3.0f, true, 2);
        rx.d.setFrequencyOffsetWithPhase(3.0f, 0.2f); std::printf("setFrequencyOffsetWithPhase\n");
        const bool r = rx.d.processPresynced(SampleSpan(fr.data(), fr.size()), 2);
        rx.status("presynced", r);
        while (rx.d.hasPendingData()) rx.drain();
        rx.idle(14);
        Samples audio = tx.noise(8000, 0.01f); append(audio, tx.frame(2 - round, 26.0f, -5.0f + 9.0f * round)); append(audio, tx.noise(20000, 0.01f));
        rx.feed(audio); rx.idle(3);
    }
    rx.d.reset(); std::printf("reset\n");
    Samples audio = tx.noise(6000, 0.01f); append(audio, tx.frame(1, 28.0f, 1.5f)); append(audio, tx.noise(16000, 0.01f));
    rx.feed(audio); rx.idle(3);
}

void midframe(const ModemConfig& c, unsigned seed) {
    Tx tx(c, seed); Rx rx(c);
    Samples f1 = tx.frame(3, 28.0f, 2.0f);
    Samples audio = tx.noise(7000, 0.01f);
    audio.insert(audio.end(), f1.begin(), f1.begin() + f1.size() / 3);          // the first frame breaks off ...
    append(audio, tx.noise(4 * 960, 0.0005f));                                    // ... near-silence (idle calls) ...
    append(audio, tx.frame(1, 28.0f, -6.0f));                                    // ... and a new preamble arrives
    append(audio, tx.noise(20000, 0.01f));
    rx.feed(audio);
    rx.idle(3);
}

void exits(const ModemConfig& c, unsigned seed) {
    Tx tx(c, seed);
    {   // sync timeout: a "frame" that never ends — 260 symbols of signal after the preamble
        Rx rx(c);
        Samples audio = tx.noise(7000, 0.01f);
        Samples f = tx.frame(1, 30.0f, 0.0f);
        append(audio, f);
        const size_t sym = c.getSymbolDuration();
        while (audio.size() < 7000 + 270 * sym) { Samples g = tx.frame(1, 30.0f, 0.0f); audio.insert(audio.end(), g.begin() + 6 * sym, g.end()); }
        rx.feed(audio, 960, false);
        std::printf("final soft=%zu\n", rx.d.getSoftBits().size());
    }
    {   // idle timeout and frame complete
        Rx rx(c);
        Samples audio = tx.noise(7000, 0.01f); append(audio, tx.frame(1, 30.0f, 0.0f));
        rx.feed(audio); rx.idle(14);
        rx.feed(tx.noise(300, 0.01f), 100); rx.idle(14);
    }
}

void getdata(const ModemConfig& c, unsigned seed) {
    Tx tx(c, seed); Rx rx(c);
    Samples audio = tx.noise(7000, 0.01f); append(audio, tx.frame(1, 30.0f, 0.0f)); append(audio, tx.noise(9000, 0.01f));
    for (size_t i = 0; i < audio.size(); i += 960) {
        const bool r = rx.d.process(SampleSpan(audio.data() + i, std::min<size_t>(960, audio.size() - i)));
        rx.status("process", r);
        if (r) {
            Bytes b = rx.d.getData();
            std::printf("  getData %zu:", b.size()); for (uint8_t v : b) std::printf(" %02x", v); std::printf("\n");
            rx.status("after-getData", false);
        }
    }
}

void decoder(CodeRate rate, unsigned seed) {
    std::mt19937 rng(seed);
    LDPCEncoder enc(rate); LDPCDecoder dec(rate);
    auto show = [&](const char* what, const Bytes& out) {
        std::printf("%s ok=%d iters=%d rate=%d n=%zu:", what, (int)dec.lastDecodeSuccess(), dec.lastIterations(), (int)dec.getRate(), out.size());
        for (uint8_t b : out) std::printf(" %02x", b);
        std::printf("\n");
    };
    auto llrs_for = [&](size_t n_cw, float amp, float sigma) {
        Bytes payload(info_bytes(rate) * n_cw); for (auto& b : payload) b = rng() & 0xFF;
        Bytes coded = enc.encode(payload);
        std::normal_distribution<float> d(0.0f, sigma);
        std::vector<float> l;
        for (uint8_t b : coded) for (int k = 7; k >= 0; --k) l.push_back((((b >> k) & 1) ? -amp : amp) + d(rng));
        return l;
    };
    for (size_t n : {size_t(1), size_t(100), size_t(647), size_t(648), size_t(649), size_t(1000), size_t(1296), size_t(2000)}) {
        std::vector<float> l = llrs_for(4, 2.0f, 1.2f); l.resize(n);
        std::printf("decodeSoft n=%zu\n", n); show(" ", dec.decodeSoft(l));
    }
    show("empty", dec.decodeSoft(std::span<const float>()));
    { std::vector<float> l = llrs_for(1, 1.0f, 1.6f); for (int it : {0, 1, 7, 50}) { dec.setMaxIterations(it); std::printf("max_iter=%d\n", it); show(" ", dec.decodeSoft(l)); } }
    { Bytes payload(info_bytes(rate)); for (auto& b : payload) b = rng() & 0xFF; Bytes coded = enc.encode(payload); coded[3] ^= 0x10; show("decode(bytes)", dec.decode(coded)); }
    { std::vector<float> l = llrs_for(1, 2.0f, 0.5f); l[5] = NAN; l[77] = INFINITY; l[300] = -INFINITY; l[9] = -0.0f; show("special", dec.decodeSoft(l)); }
    for (CodeRate r : {CodeRate::R1_4, CodeRate::R1_2, CodeRate::R5_6, rate}) {
        dec.setRate(r); enc.setRate(r);
        Bytes payload(20); for (auto& b : payload) b = rng() & 0xFF;
        Bytes coded = enc.encode(payload);
        std::vector<float> l; std::normal_distribution<float> d(0.0f, 0.9f);
        for (uint8_t b : coded) for (int k = 7; k >= 0; --k) l.push_back((((b >> k) & 1) ? -2.0f : 2.0f) + d(rng));
        std::printf("setRate %d\n", (int)r); show(" ", dec.decodeSoft(l));
    }
}

void interleave(unsigned seed) {
    std::mt19937 rng(seed);
    auto fl = [&](size_t n) { std::vector<float> v(n); for (auto& x : v) x = float(int(rng() % 2001) - 1000) / 64.0f; return v; };
    auto by = [&](size_t n) { Bytes v(n); for (auto& x : v) x = rng() & 0xFF; return v; };
    auto pf = [&](const char* w, const std::vector<float>& v) { std::printf("%s %zu:", w, v.size()); for (float x : v) std::printf(" %08x", bits(x)); std::printf("\n"); };
    auto pb = [&](const char* w, const Bytes& v) { std::printf("%s %zu:", w, v.size()); for (uint8_t x : v) std::printf(" %02x", x); std::printf("\n"); };
    for (auto rc : {std::pair<size_t, size_t>{6, 108}, {4, 10}, {27, 24}}) {
        Interleaver il(rc.first, rc.second);
        std::printf("Interleaver %zu x %zu perm[5]=%zu\n", il.getRows(), il.getCols(), il.getPermutation(5));
        const size_t n = rc.first * rc.second;
        // (soft-bit inputs shorter than rows x cols make the reference write and read past its vectors: not a case)
        for (size_t len : {n, n + 40}) { auto v = fl(len); pf(" fi", il.interleave(v)); pf(" fd", il.deinterleave(v)); }
        for (size_t len : {(n + 7) / 8, size_t(3), n}) { auto v = by(len); pb(" bi", il.interleave(v)); pb(" bd", il.deinterleave(v)); }
    }
    for (size_t bps : {size_t(60), size_t(30), size_t(90), size_t(116), size_t(176), size_t(220), size_t(700)}) {
        ChannelInterleaver ci(bps);
        std::printf("ChannelInterleaver %zu sep=%zu\n", bps, ci.getSymbolSeparation());
        for (size_t len : {size_t(648), size_t(100), size_t(900)}) { auto v = fl(len); pf(" fi", ci.interleave(v)); pf(" fd", ci.deinterleave(v)); }
        for (size_t len : {size_t(81), size_t(10), size_t(100)}) { auto v = by(len); pb(" bi", ci.interleave(v)); pb(" bd", ci.deinterleave(v)); }
    }
    { ChannelInterleaver ci(48, 256); std::printf("ChannelInterleaver 48/256 sep=%zu\n", ci.getSymbolSeparation()); auto v = fl(256); pf(" fi", ci.interleave(v)); pf(" fd", ci.deinterleave(v)); }
}

}  // namespace

int main(int argc, char** argv) {
    if (argc < 5) { std::fprintf(stderr, "usage: %s scenario fft modulation code_rate [seed]\n", argv[0]); return 2; }
    setLogLevel(LogLevel::ERROR);
    const std::string sc = argv[1];
    const int fft = std::atoi(argv[2]);
    const Modulation mod = static_cast<Modulation>(std::atoi(argv[3]));
    const CodeRate rate = static_cast<CodeRate>(std::atoi(argv[4]));
    const unsigned seed = argc > 5 ? (unsigned)std::atoi(argv[5]) : 1u;
    const ModemConfig c = make_config(fft, mod, rate);
    if (sc == "carry") carry(c, seed);
    else if (sc == "timing") timing(c, seed);
    else if (sc == "setcfo") setcfo(c, seed);
    else if (sc == "presynced") presynced(c, seed);
    else if (sc == "midframe") midframe(c, seed);
    else if (sc == "mixed") mixed(c, seed);
    else if (sc == "exits") exits(c, seed);
    else if (sc == "getdata") getdata(c, seed);
    else if (sc == "decoder") decoder(rate, seed);
    else if (sc == "interleave") interleave(seed);
    else { std::fprintf(stderr, "unknown scenario %s\n", sc.c_str()); return 2; }
    return 0;
}
