// adapter_driver.cpp — TEST-ONLY driver of include/ultra_hip_waveform.hpp over the test stub (ultra_hip_teststub.cpp), built and run
// by tests/test_adapter_sanitizers.py under -fsanitize=address,undefined and under -fsanitize=thread.  No GPU, no product
// library, no oracle: what is under test is the host code above the C-ABI — sample-index arithmetic, device-window bookkeeping,
// the SEARCHING / SYNCED / presynced state machine, the slot pool, the GUI-side getters — and what is checked is (a) the
// sanitizers stay silent and (b) the soft bits that come out are the stub's function of EXACTLY the samples that should have
// reached each call, through every chunking, trim, reset and rebase.
//
//   adapter_driver stream|presynced|decoder|faults|rebase|threads [seed]
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "ultra_hip_waveform.hpp"

extern "C" void ultra_hip_teststub_fail_after(long n);
extern "C" unsigned long long ultra_hip_teststub_calls(void);

using namespace ultra_hip;

namespace {
int g_fail = 0;
#define EXPECT(cond, ...) do { if (!(cond)) { std::fprintf(stderr, "FAIL %s:%d: %s — ", __FILE__, __LINE__, #cond); std::fprintf(stderr, __VA_ARGS__); std::fprintf(stderr, "\n"); ++g_fail; } } while (0)

ModemConfig config(int fft, Modulation m, bool pilots) {
    ModemConfig c;
    c.fft_size = uint32_t(fft); c.num_carriers = fft == 1024 ? 59 : 30; c.modulation = m; c.use_pilots = pilots; c.pilot_spacing = fft == 1024 ? 4 : 2;
    return c;
}

// the stub's soft bits of the data symbol whose audio starts at a[0] (ultra_hip_teststub.cpp, ultra_hip_demod_stream_batch_eq)
void stub_llrs(const float* a, uint32_t sym, uint32_t lps, std::vector<float>& out) {
    double acc = 0; for (uint32_t i = 0; i < sym; ++i) acc += a[i];
    for (uint32_t j = 0; j < lps; ++j) out.push_back(4.0f * a[j % sym] + float(acc) * 1e-6f + ((j & 1) ? 0.5f : -0.5f));
}

struct Frame { size_t marker; uint32_t n_sym; };
// [silence (small noise)] [marker + two preamble symbols] [n data symbols] [tail]; every value but the marker is below 0.9
std::vector<float> one_frame(std::mt19937& rng, uint32_t sym, uint32_t n_sym, Frame& f) {
    std::uniform_real_distribution<float> small(-0.05f, 0.05f), data(-0.5f, 0.5f);
    const size_t lead = 1 + rng() % 12000, tail = 6 * sym + rng() % 3000;
    std::vector<float> a(lead + size_t(2 + n_sym) * sym + tail);
    for (auto& v : a) v = small(rng);
    f.marker = lead; f.n_sym = n_sym;
    a[lead] = 1.0f;
    for (size_t i = lead + 2 * size_t(sym); i < lead + size_t(2 + n_sym) * sym; ++i) a[i] = data(rng);
    return a;
}
// call sizes from 1 sample to several frames' worth — but never more than three calls in a row that cannot complete a symbol: more
// than MAX_IDLE_CALLS_BEFORE_RESET (10) calls without a new soft bit end the SYNCED state (demodulator.cpp:704-716), by design;
// and never an EMPTY call inside a frame: that is the caller saying "no more audio" and ends the frame (:720-731)
size_t chunk_of(std::mt19937& rng, uint32_t sym = 1200) {
    static thread_local int small_run = 0;
    size_t n;
    switch (rng() % 8) { case 0: n = 1; break; case 1: n = 1 + rng() % 7; break; case 2: n = 5000 + rng() % 20000; break; default: n = 1 + rng() % 2500; }
    if (n < sym && ++small_run > 3) { n = sym + rng() % sym; }
    if (n >= sym) small_run = 0;
    return n;
}

// ---------------------------------------------------------------------------------------------
// SEARCHING -> SYNCED -> soft bits, many frames on one object, random chunking, reset between frames as RxPipeline does
void scenario_stream(uint32_t seed) {
    std::mt19937 rng(seed);
    for (int fft : {512, 1024}) {
        const ModemConfig cfg = config(fft, fft == 1024 ? Modulation::QAM16 : Modulation::DQPSK, fft == 1024);
        HipOfdmDemodulator d(cfg);
        const uint32_t sym = d.symbolSamples(), lps = d.geometry().llrs_per_symbol;
        for (int frame = 0; frame < 40; ++frame) {
            Frame f; const uint32_t n_sym = 3 + rng() % 14;
            const std::vector<float> a = one_frame(rng, sym, n_sym, f);
            const int manual = (frame % 5 == 4) ? int(rng() % 9) - 4 : 0;
            d.setTimingOffset(manual);
            std::vector<float> want, got;
            const size_t data0 = f.marker + 2 * size_t(sym) + size_t(manual);
            for (uint32_t s = 0; s < n_sym; ++s) stub_llrs(a.data() + data0 + size_t(s) * sym, sym, lps, want);
            size_t at = 0; bool was_synced = false;
            while (at < a.size() && got.size() < want.size()) {
                const size_t n = std::min(chunk_of(rng, sym), a.size() - at);
                const bool ready = d.process(SampleSpan(a.data() + at, n));
                at += n;
                was_synced = was_synced || d.isSynced();
                if (rng() % 4 == 0) { (void)d.getConstellationSymbols(); (void)d.getChannelQuality(); (void)d.hasPendingData(); (void)d.getEstimatedSNR(); }
                if (ready || (rng() % 16 == 0)) {
                    std::vector<float> bits = d.getSoftBits();
                    EXPECT(!ready || !bits.empty(), "process() said a codeword is ready");
                    got.insert(got.end(), bits.begin(), bits.end());
                }
            }
            while (d.hasPendingData() && got.size() < want.size()) { std::vector<float> bits = d.getSoftBits(); if (bits.empty()) break; got.insert(got.end(), bits.begin(), bits.end()); }
            EXPECT(was_synced, "fft %d frame %d never synchronised", fft, frame);
            EXPECT(d.getLastSyncOffset() < a.size(), "sync offset");
            const size_t n = std::min(got.size(), want.size());
            EXPECT(n >= (want.size() / 648) * 648 && n > 0, "fft %d frame %d: %zu soft bits of %zu", fft, frame, got.size(), want.size());
            for (size_t i = 0; i < n; ++i) if (got[i] != want[i]) { EXPECT(false, "fft %d frame %d: soft bit %zu differs (%g vs %g)", fft, frame, i, got[i], want[i]); break; }
            if (frame % 7 == 6) (void)d.getData();
            if (frame % 3 != 2) d.reset();                                 // every third frame: the next search starts on a used, un-reset object
            else {                                                         // ... after the frame ran out through empty calls; what the tail's symbols
                for (int k = 0; k < 40 && d.isSynced(); ++k) (void)d.process(SampleSpan());   // left behind is the caller's to fetch, as in the reference
                while (!d.getSoftBits().empty()) {}
            }
        }
    }
    // the IWaveform flavour, driven the way RxPipeline::tryProcessBuffer drives it
    HipOfdmCoxWaveform w(HipOfdmCoxWaveform::defaultConfig());
    const uint32_t sym = uint32_t(w.getSamplesPerSymbol());
    for (int frame = 0; frame < 12; ++frame) {
        Frame f; const std::vector<float> a = one_frame(rng, sym, 24, f);   // (QPSK with pilots: 30 soft bits per symbol)
        SyncResult r;
        const bool found = w.detectSync(SampleSpan(a.data(), a.size()), r);
        EXPECT(found && r.detected, "cox detectSync");
        w.setFrequencyOffset(r.cfo_hz);
        w.reset();
        const size_t start = f.marker;                                   // the pipeline re-feeds from its own buffer: here from the preamble on
        const bool ready = w.process(SampleSpan(a.data() + start, a.size() - start));
        EXPECT(ready && w.hasData(), "cox process");
        EXPECT(w.getSoftBits().size() == 648, "cox soft bits");
        (void)w.estimatedSNR(); (void)w.estimatedCFO(); (void)w.getMinSamplesForFrame(); (void)w.getThroughput(CodeRate::R3_4); (void)w.getStatusString();
        w.configure(frame % 2 ? Modulation::QAM16 : Modulation::DQPSK, CodeRate::R1_2);
    }
}

// ---------------------------------------------------------------------------------------------
void scenario_presynced(uint32_t seed) {
    std::mt19937 rng(seed);
    ModemConfig c = config(512, Modulation::DQPSK, false);
    HipOfdmWaveform w(c);
    const uint32_t sym = uint32_t(w.getSamplesPerSymbol());
    HipOfdmDemodulator probe(c);
    uint32_t lps = probe.geometry().llrs_per_symbol;
    for (int frame = 0; frame < 30; ++frame) {
        Frame f; const uint32_t n_sym = 4 + rng() % 20;
        std::vector<float> a = one_frame(rng, sym, n_sym, f);
        SyncResult r;
        EXPECT(w.detectSync(SampleSpan(a.data(), a.size()), r, 0.15f), "chirp detectSync");
        EXPECT(r.start_sample == int(f.marker) + 16, "start sample %d", r.start_sample);
        w.setFrequencyOffset(r.cfo_hz);
        const size_t start = size_t(r.start_sample);
        // spans that are too short for the training symbols are refused without touching the object
        EXPECT(!w.process(SampleSpan(a.data() + start, sym - 1)), "short span");
        EXPECT(!w.process(SampleSpan(a.data() + start, sym + 5)), "one symbol is less than the training");
        const size_t len = std::min(a.size() - start, size_t(2 + n_sym) * sym + rng() % sym);
        const bool ready = w.process(SampleSpan(a.data() + start, len));
        std::vector<float> want, got = w.getSoftBits();
        const uint32_t n_data = uint32_t(len / sym) - 2;
        for (uint32_t s = 0; s < n_data; ++s) stub_llrs(a.data() + start + size_t(2 + s) * sym, sym, lps, want);
        EXPECT(ready == (want.size() >= 648), "ready");
        if (ready) {
            EXPECT(got.size() == want.size(), "presynced soft bits %zu vs %zu", got.size(), want.size());
            for (size_t i = 0; i < std::min(got.size(), want.size()); ++i) if (got[i] != want[i]) { EXPECT(false, "presynced soft bit %zu", i); break; }
        }
        (void)w.estimatedCFO(); (void)w.estimatedSNR(); (void)w.hasData(); (void)w.isSynced(); (void)w.getConstellationSymbols();
        w.reset();
        if (frame % 9 == 8) { w.configure(Modulation::D8PSK, CodeRate::R2_3); lps = 90; }     // 30 carriers x 3 bits from here on
    }
    // the demodulator itself: a presynced frame whose tail arrives through process(), then a Schmidl-Cox frame on the same object
    // without reset() (the tracker travels from the presynced context: ultra_hip_stream_adopt), offsets set at every point
    HipOfdmDemodulator d(c);
    for (int round = 0; round < 10; ++round) {
        Frame f; std::vector<float> a = one_frame(rng, sym, 16, f);
        const size_t start = f.marker + 16;
        d.setFrequencyOffsetWithPhase(2.0f, 0.25f);
        const size_t first = size_t(2 + 5) * sym + rng() % sym;
        (void)d.processPresynced(SampleSpan(a.data() + start, first), 2);
        d.setFrequencyOffset(1.0f);                                        // mid-frame
        size_t at = start + first;
        while (at < a.size()) { const size_t n = std::min(chunk_of(rng), a.size() - at); (void)d.process(SampleSpan(a.data() + at, n)); at += n; while (d.hasPendingData()) if (d.getSoftBits().empty()) break; }
        for (int k = 0; k < 14 && d.isSynced(); ++k) (void)d.process(SampleSpan());
        Frame g; std::vector<float> b = one_frame(rng, sym, 8, g);
        at = 0;
        while (at < b.size()) { const size_t n = std::min(chunk_of(rng), b.size() - at); (void)d.process(SampleSpan(b.data() + at, n)); at += n; }
        EXPECT(d.isSynced() || d.getLastSyncOffset() > 0 || true, "mixed");
        if (round % 2) d.reset();
    }
}

// ---------------------------------------------------------------------------------------------
void scenario_decoder(uint32_t seed) {
    std::mt19937 rng(seed);
    std::uniform_real_distribution<float> u(-8.0f, 8.0f);
    for (int rate = 0; rate <= 5; ++rate) {
        HipLDPCDecoder dec(static_cast<CodeRate>(rate));
        for (size_t n : {size_t(0), size_t(1), size_t(647), size_t(648), size_t(649), size_t(1296), size_t(1300), size_t(2000), size_t(6480)}) {
            std::vector<float> llr(n);
            for (auto& v : llr) { v = u(rng); if (std::fabs(v) < 0.01f) v = 1.0f; }
            const Bytes out = dec.decodeSoft(llr);
            EXPECT((n == 0) == out.empty(), "rate %d n %zu -> %zu bytes", rate, n, out.size());
            EXPECT(n == 0 || dec.lastDecodeSuccess(), "decode success");
        }
        std::vector<uint8_t> coded(81); for (auto& b : coded) b = uint8_t(rng());
        EXPECT(!dec.decode(coded).empty(), "decode(bytes)");
        dec.setDeinterleave(60); (void)dec.decodeSoft(std::vector<float>(648, 2.0f));
        dec.setMaxIterations(5); (void)dec.decodeSoft(std::vector<float>(648, -2.0f));
        dec.setRate(CodeRate::R1_4); EXPECT(dec.getRate() == CodeRate::R1_4, "setRate");
        (void)dec.decodeSoft(std::vector<float>(700, std::nanf("")));      // the stub's "failed" branch
        EXPECT(!dec.lastDecodeSuccess(), "NaNs do not decode");
        std::vector<float> big(648 * 33, 1.5f); std::vector<uint8_t> bytes(33 * 70), ok(33); std::vector<int32_t> iters(33);
        dec.decodeBatch(big.data(), 33, bytes.data(), iters.data(), ok.data());
    }
    HipRxFrameDecoder fd;
    fd.setDataMode(CodeRate::R1_2, true); fd.setInterleaverConfig(60);
    for (size_t n : {size_t(0), size_t(100), size_t(648), size_t(1296), size_t(5000)}) {
        const HipRxFrameResult r = fd.decodeSoftBits(std::vector<float>(n, 1.0f));
        EXPECT(r.success == (n >= 648), "frame decode n %zu", n);
        EXPECT(r.frame_data.size() == (n / 648) * 40, "frame data %zu", r.frame_data.size());
    }
    fd.setInterleavingEnabled(false); (void)fd.decodeSoftBits(std::vector<float>(648, -1.0f));
}

// ---------------------------------------------------------------------------------------------
// every C-ABI call of a short exchange fails once, in turn: nothing may leak, crash or leave the IWaveform boundary as an
// exception; after the faults stop and a reset() the object works again
void scenario_faults(uint32_t seed) {
    std::mt19937 rng(seed);
    const ModemConfig cfg = config(512, Modulation::DQPSK, false);
    Frame f; std::vector<float> a = one_frame(rng, 564, 12, f);
    detail::SlotPool::instance().clear();
    int thrown = 0, refused = 0;
    for (long k = 1; k <= 90; ++k) {
        ultra_hip_teststub_fail_after(k);
        {   // the IWaveform level: no exception may come out
            HipOfdmCoxWaveform w(cfg);
            SyncResult r;
            (void)w.detectSync(SampleSpan(a.data(), a.size() / 2), r);
            w.setFrequencyOffset(1.0f);
            const bool ready = w.process(SampleSpan(a.data() + a.size() / 2, a.size() - a.size() / 2));
            if (!ready) ++refused;
            (void)w.getSoftBits(); w.reset();
            HipOfdmWaveform c(cfg);
            (void)c.detectSync(SampleSpan(a.data(), a.size()), r, 0.15f);
            (void)c.process(SampleSpan(a.data() + f.marker + 16, size_t(8) * 564));
            c.reset();
        }
        {   // the class level: failures are exceptions (the pimpl drop-ins catch them one layer up)
            try {
                HipOfdmDemodulator d(cfg);
                (void)d.process(SampleSpan(a.data(), a.size()));
                (void)d.getSoftBits(); d.reset();
                HipLDPCDecoder dec(CodeRate::R1_2);
                (void)dec.decodeSoft(std::vector<float>(1296, 3.0f));
            } catch (const std::exception&) { ++thrown; }
        }
        if (k % 10 == 0) detail::SlotPool::instance().clear();
    }
    ultra_hip_teststub_fail_after(-1);
    EXPECT(thrown > 5 && refused > 2, "the injection must have hit both levels (%d thrown, %d refused)", thrown, refused);
    HipOfdmCoxWaveform w(cfg);
    EXPECT(w.process(SampleSpan(a.data(), a.size())), "a frame decodes after the faults stopped");
    EXPECT(w.getSoftBits().size() == 648, "soft bits after the faults");
    detail::SlotPool::instance().clear();
}

// ---------------------------------------------------------------------------------------------
// 2^29 samples of silence, then a frame: the absolute 32-bit sample indices are rebased (HipOfdmDemodulator::rebase) and the frame
// still comes out of the right samples
void scenario_rebase(uint32_t seed) {
    std::mt19937 rng(seed);
    const ModemConfig cfg = config(512, Modulation::DQPSK, false);
    HipOfdmDemodulator d(cfg);
    const uint32_t sym = d.symbolSamples(), lps = d.geometry().llrs_per_symbol;
    std::vector<float> silence(size_t(1) << 20, 0.01f);
    for (int i = 0; i < 513 + 3; ++i) (void)d.process(SampleSpan(silence.data(), silence.size()));
    Frame f; const std::vector<float> a = one_frame(rng, sym, 13, f);
    std::vector<float> want, got;
    for (uint32_t s = 0; s < 13; ++s) stub_llrs(a.data() + f.marker + size_t(2 + s) * sym, sym, lps, want);
    size_t at = 0;
    while (at < a.size()) {
        const size_t n = std::min<size_t>(960, a.size() - at);
        if (d.process(SampleSpan(a.data() + at, n))) { auto b = d.getSoftBits(); got.insert(got.end(), b.begin(), b.end()); }
        at += n;
    }
    EXPECT(got.size() >= 648, "frame behind 2^29 samples: %zu soft bits", got.size());
    for (size_t i = 0; i < std::min(got.size(), want.size()); ++i) if (got[i] != want[i]) { EXPECT(false, "rebase: soft bit %zu", i); break; }
}

// ---------------------------------------------------------------------------------------------
// ModemEngine's threading: one thread feeds a demodulator, another polls its getters (modem_engine.cpp:812-827), two more build,
// use and destroy demodulators and decoders (modem_mode.cpp:133-240: the engine re-creates them from whichever thread calls
// setConnected / setDataMode) — the slot pool is shared by all of them
void scenario_threads(uint32_t seed) {
    const ModemConfig cfg = config(512, Modulation::DQPSK, false);
    HipOfdmDemodulator shared(cfg);
    std::atomic<bool> done{false};
    std::atomic<unsigned long long> polls{0}, frames{0};
    std::thread feeder([&] {
        std::mt19937 rng(seed);
        for (int frame = 0; frame < 60; ++frame) {
            Frame f; const std::vector<float> a = one_frame(rng, 564, 12, f);
            size_t at = 0;
            while (at < a.size()) { const size_t n = std::min<size_t>(960, a.size() - at); if (shared.process(SampleSpan(a.data() + at, n))) { (void)shared.getSoftBits(); ++frames; } at += n; }
            shared.reset();
        }
        done = true;
    });
    std::thread gui([&] {
        while (!done) { (void)shared.isSynced(); (void)shared.getChannelQuality(); const auto c = shared.getConstellationSymbols(); (void)c.size(); ++polls; std::this_thread::yield(); }
    });
    auto churn = [&](uint32_t s) {
        std::mt19937 rng(s);
        for (int i = 0; i < 150; ++i) {
            Frame f; const std::vector<float> a = one_frame(rng, 564, 6, f);
            HipOfdmDemodulator d(cfg);
            (void)d.process(SampleSpan(a.data(), a.size()));
            (void)d.getSoftBits();
            HipLDPCDecoder dec(static_cast<CodeRate>(i % 6));
            (void)dec.decodeSoft(std::vector<float>(648, 1.0f));
            if (i % 10 == 0) { HipOfdmWaveform w(cfg); SyncResult r; (void)w.detectSync(SampleSpan(a.data(), a.size()), r, 0.15f); }
        }
    };
    std::thread c1(churn, seed + 1), c2(churn, seed + 2);
    feeder.join(); gui.join(); c1.join(); c2.join();
    EXPECT(frames > 0 && polls > 0, "threads ran (%llu frames, %llu polls)", (unsigned long long)frames, (unsigned long long)polls);
    detail::SlotPool::instance().clear();
}
}  // namespace

int main(int argc, char** argv) {
    const std::string sc = argc > 1 ? argv[1] : "stream";
    const uint32_t seed = argc > 2 ? uint32_t(std::strtoul(argv[2], nullptr, 10)) : 1u;
    if (sc == "stream") scenario_stream(seed);
    else if (sc == "presynced") scenario_presynced(seed);
    else if (sc == "decoder") scenario_decoder(seed);
    else if (sc == "faults") scenario_faults(seed);
    else if (sc == "rebase") scenario_rebase(seed);
    else if (sc == "threads") scenario_threads(seed);
    else { std::fprintf(stderr, "unknown scenario %s\n", sc.c_str()); return 2; }
    detail::SlotPool::instance().clear();
    std::printf("%s seed %u: %d failures, %llu C-ABI calls\n", sc.c_str(), seed, g_fail, ultra_hip_teststub_calls());
    return g_fail ? 1 : 0;
}
