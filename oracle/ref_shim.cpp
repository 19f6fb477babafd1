// oracle/ref_shim.cpp — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// Thin extern "C" shim over the *compiled reference* (secup/ProjectUltra built
// from the sources where they lie under /root/reference by oracle/Makefile,
// output oracle/_ref/libultra_ref.so).  It exists so the tests can
//   (1) pin the C restatement in oracle/ultra_oracle.c stage by stage, and
//   (2) generate the golden fixtures under tests/golden/ (tests/golden/make_golden.py).
// Nothing here is copied from the reference: this file only *calls* it.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
// the resulting library.
//
// Stage-level access: OFDMDemodulator::Impl is fully declared in the
// reference's private header src/ofdm/demodulator_impl.hpp:18-168; we reach
// impl_ by compiling this TU with `private` redefined around the public header.

#include <algorithm>
#include <atomic>
#include <cmath>
#include <complex>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <thread>
#include <memory>
#include <mutex>
#include <random>
#include <span>
#include <vector>
#include <unistd.h>
#include <fcntl.h>

#define private public
#include "ultra/ofdm.hpp"
#undef private
#include "ultra/dsp.hpp"
#include "ultra/fec.hpp"
#include "ofdm/demodulator_impl.hpp"
#include "sim/hf_channel.hpp"
#define private public
#include "sync/chirp_sync.hpp"
#include "gui/modem/rx_pipeline.hpp"
#undef private
#include "protocol/frame_v2.hpp"
#include "ultra/logging.hpp"

#include "../include/ultra_hip.h"

using namespace ultra;

namespace {

ModemConfig to_cfg(const ultra_hip_config* c) {
    ModemConfig m;
    m.sample_rate = c->sample_rate;
    m.center_freq = c->center_freq;
    m.fft_size = c->fft_size;
    m.num_carriers = c->num_carriers;
    m.cp_mode = static_cast<CyclicPrefixMode>(c->cp_mode);
    m.symbol_guard = c->symbol_guard;
    m.pilot_spacing = c->pilot_spacing;
    m.use_pilots = c->use_pilots != 0;
    m.modulation = static_cast<Modulation>(c->modulation);
    m.code_rate = static_cast<CodeRate>(c->code_rate);
    m.adaptive_eq_enabled = c->adaptive_eq_enabled != 0;
    m.adaptive_eq_use_rls = c->adaptive_eq_use_rls != 0;
    if (c->adaptive_eq_enabled) { m.decision_directed = c->decision_directed != 0; m.lms_mu = c->lms_mu; m.rls_lambda = c->rls_lambda; }
    if (c->sync_threshold != 0.0f) m.sync_threshold = c->sync_threshold;
    return m;
}

// The reference prints unconditionally to stderr inside the hot path
// (channel_equalizer.cpp:26-32,96-135,...).  Silence it around calls.
struct StderrMute {
    int saved = -1;
    StderrMute() {
        fflush(stderr);
        saved = dup(2);
        int nul = open("/dev/null", O_WRONLY);
        if (nul >= 0) { dup2(nul, 2); close(nul); }
    }
    ~StderrMute() {
        fflush(stderr);
        if (saved >= 0) { dup2(saved, 2); close(saved); }
    }
};

}  // namespace

extern "C" {

// ---------------------------------------------------------------- primitives

// FFT::forward (src/dsp/fft.cpp:124-133), built-in radix-2 path.
int ref_fft_forward(uint32_t n, const float* in_ri, float* out_ri) {
    FFT fft(n);
    std::vector<Complex> in(n), out;
    for (uint32_t i = 0; i < n; ++i) in[i] = Complex(in_ri[2 * i], in_ri[2 * i + 1]);
    fft.forward(in, out);
    for (uint32_t i = 0; i < n; ++i) { out_ri[2 * i] = out[i].real(); out_ri[2 * i + 1] = out[i].imag(); }
    return 0;
}

int ref_fft_inverse(uint32_t n, const float* in_ri, float* out_ri) {
    FFT fft(n);
    std::vector<Complex> in(n), out;
    for (uint32_t i = 0; i < n; ++i) in[i] = Complex(in_ri[2 * i], in_ri[2 * i + 1]);
    fft.inverse(in, out);
    for (uint32_t i = 0; i < n; ++i) { out_ri[2 * i] = out[i].real(); out_ri[2 * i + 1] = out[i].imag(); }
    return 0;
}

// NCO::next sequence (src/dsp/filters.cpp:228-238).
int ref_nco(float freq, float fs, uint32_t n, float* out_ri) {
    NCO nco(freq, fs);
    for (uint32_t i = 0; i < n; ++i) {
        Complex c = nco.next();
        out_ri[2 * i] = c.real();
        out_ri[2 * i + 1] = c.imag();
    }
    return 0;
}

// ---------------------------------------------------------------------- FEC

// LDPCEncoder::encode (src/fec/ldpc_encoder.cpp:193-257).
int ref_ldpc_encode(uint32_t rate, const uint8_t* data, uint32_t n, uint8_t* out, uint32_t cap) {
    LDPCEncoder enc(static_cast<CodeRate>(rate));
    Bytes r = enc.encode(ByteSpan(data, n));
    if (r.size() > cap) return -1;
    std::memcpy(out, r.data(), r.size());
    return static_cast<int>(r.size());
}

// LDPCDecoder::decodeSoft (src/fec/ldpc_decoder.cpp:283-428).
int ref_ldpc_decode_soft(uint32_t rate, int max_iters, const float* llr, uint32_t n_llr,
                         uint8_t* out, uint32_t cap, int* success, int* iters) {
    LDPCDecoder dec(static_cast<CodeRate>(rate));
    dec.setMaxIterations(max_iters);
    Bytes r = dec.decodeSoft(std::span<const float>(llr, n_llr));
    if (r.size() > cap) return -1;
    std::memcpy(out, r.data(), r.size());
    *success = dec.lastDecodeSuccess() ? 1 : 0;
    *iters = dec.lastIterations();
    return static_cast<int>(r.size());
}

// Batched decode of n_cw independent 648-LLR codewords with ONE decoder
// object (the tools hold one decoder per run; rx_pipeline.cpp:499-511 rebuilds
// it per codeword — results are identical, H is deterministic).
int ref_ldpc_decode_batch(uint32_t rate, int max_iters, const float* llr, uint32_t n_cw,
                          uint8_t* out, uint32_t bytes_per_cw, int32_t* iters, uint8_t* ok) {
    LDPCDecoder dec(static_cast<CodeRate>(rate));
    dec.setMaxIterations(max_iters);
    for (uint32_t c = 0; c < n_cw; ++c) {
        Bytes r = dec.decodeSoft(std::span<const float>(llr + 648ull * c, 648));
        if (r.size() != bytes_per_cw) return -1;
        std::memcpy(out + (size_t)bytes_per_cw * c, r.data(), r.size());
        iters[c] = dec.lastIterations();
        ok[c] = dec.lastDecodeSuccess() ? 1 : 0;
    }
    return 0;
}

// Interleaver / ChannelInterleaver soft-bit permutations
// (src/fec/ldpc_decoder.cpp:454-540,547-620).
int ref_interleaver_deinterleave(uint32_t rows, uint32_t cols, const float* in, uint32_t n, float* out) {
    Interleaver il(rows, cols);
    auto r = il.deinterleave(std::span<const float>(in, n));
    std::memcpy(out, r.data(), r.size() * sizeof(float));
    return (int)r.size();
}
int ref_channel_interleaver_deinterleave(uint32_t bits_per_symbol, uint32_t total, const float* in,
                                         uint32_t n, float* out) {
    ChannelInterleaver il(bits_per_symbol, total);
    auto r = il.deinterleave(std::span<const float>(in, n));
    std::memcpy(out, r.data(), r.size() * sizeof(float));
    return (int)r.size();
}
int ref_channel_interleaver_interleave(uint32_t bits_per_symbol, uint32_t total, const float* in,
                                       uint32_t n, float* out) {
    ChannelInterleaver il(bits_per_symbol, total);
    auto r = il.interleave(std::span<const float>(in, n));
    std::memcpy(out, r.data(), r.size() * sizeof(float));
    return (int)r.size();
}

// ---------------------------------------------------------------- modulator

// preamble + modulate(encoded) exactly as the harness does
// (tools/test_nvis_mode.cpp:60-71).  Returns total samples; *preamble_len set.
int ref_modulate_frame(const ultra_hip_config* c, const uint8_t* encoded, uint32_t n_enc,
                       float* out, uint32_t cap, uint32_t* preamble_len) {
    StderrMute mute;
    ModemConfig cfg = to_cfg(c);
    OFDMModulator mod(cfg);
    Samples pre = mod.generatePreamble();
    Samples dat = mod.modulate(ByteSpan(encoded, n_enc), cfg.modulation);
    if (pre.size() + dat.size() > cap) return -1;
    std::memcpy(out, pre.data(), pre.size() * sizeof(float));
    std::memcpy(out + pre.size(), dat.data(), dat.size() * sizeof(float));
    *preamble_len = (uint32_t)pre.size();
    return (int)(pre.size() + dat.size());
}

// generateTrainingSymbols(count) + modulate(encoded) — the chirp-synced TX
// order (src/ofdm/modulator.cpp:534-580, ofdm_chirp_waveform.cpp).
int ref_modulate_presynced(const ultra_hip_config* c, const uint8_t* encoded, uint32_t n_enc,
                           float* out, uint32_t cap) {
    StderrMute mute;
    ModemConfig cfg = to_cfg(c);
    OFDMModulator mod(cfg);
    Samples tr = mod.generateTrainingSymbols((int)c->training_symbols);
    Samples dat = mod.modulate(ByteSpan(encoded, n_enc), cfg.modulation);
    if (tr.size() + dat.size() > cap) return -1;
    std::memcpy(out, tr.data(), tr.size() * sizeof(float));
    std::memcpy(out + tr.size(), dat.data(), dat.size() * sizeof(float));
    return (int)(tr.size() + dat.size());
}

// ------------------------------------------------------------------ channel

// WattersonChannel::process (src/sim/hf_channel.hpp:106-168).
int ref_watterson(float snr_db, float delay_ms, float doppler_hz, float g1, float g2,
                  int fading, int multipath, int noise, uint32_t seed,
                  const float* in, uint32_t n, float* out) {
    sim::WattersonChannel::Config cc;
    cc.snr_db = snr_db;
    cc.delay_spread_ms = delay_ms;
    cc.doppler_spread_hz = doppler_hz;
    cc.path1_gain = g1;
    cc.path2_gain = g2;
    cc.sample_rate = 48000;
    cc.fading_enabled = fading != 0;
    cc.multipath_enabled = multipath != 0;
    cc.noise_enabled = noise != 0;
    sim::WattersonChannel ch(cc, seed);
    Samples r = ch.process(SampleSpan(in, n));
    std::memcpy(out, r.data(), n * sizeof(float));
    return 0;
}

// WattersonChannel::applyCFO (src/sim/hf_channel.hpp:161-232) of a freshly constructed channel, in place.
int ref_channel_apply_cfo(float cfo_hz, float* samples, uint32_t n) {
    sim::WattersonChannel::Config cc;
    cc.cfo_hz = cfo_hz;
    cc.sample_rate = 48000;
    sim::WattersonChannel ch(cc, 1);
    Samples s(samples, samples + n);
    ch.applyCFO(s);
    std::memcpy(samples, s.data(), n * sizeof(float));
    return 0;
}

// -------------------------------------------------------------- demodulator

// Full reference receive: fresh demodulator, feed `chunk`-sample pieces
// through OFDMDemodulator::process (Schmidl-Cox search + SYNCED loop), then
// getSoftBits() once (tools/test_nvis_mode.cpp:88-101).
int ref_demod_process(const ultra_hip_config* c, const float* audio, uint32_t n, uint32_t chunk,
                      float* llr_out, uint32_t cap, uint32_t* sync_offset, float* cfo_hz) {
    StderrMute mute;
    ModemConfig cfg = to_cfg(c);
    OFDMDemodulator demod(cfg);
    for (uint32_t i = 0; i < n; i += chunk) {
        uint32_t len = std::min(chunk, n - i);
        demod.process(SampleSpan(audio + i, len));
    }
    if (sync_offset) *sync_offset = (uint32_t)demod.getLastSyncOffset();
    if (cfo_hz) *cfo_hz = demod.getFrequencyOffset();
    // all soft bits accumulated (not only the first 648) — callers slice
    auto& sb = demod.impl_->soft_bits;
    uint32_t m = (uint32_t)std::min<size_t>(sb.size(), cap);
    std::memcpy(llr_out, sb.data(), m * sizeof(float));
    return (int)sb.size();
}

// The whole receive of a batch of raw streams on worker threads: per stream a FRESH OFDMDemodulator fed `chunk` samples per
// process() call (search, sync, SYNCED demodulation: src/ofdm/demodulator.cpp:461-760) and LDPCDecoder::decodeSoft of the
// first 648 soft bits — the reference's CPU path for bench.py's raw-audio line.  found[f] = 0 (and cleared results) where
// the stream produced fewer than 648 soft bits.
int ref_receive_batch_mt(const ultra_hip_config* c, const float* audio, size_t stream_stride, uint32_t n_samples, uint32_t chunk,
                         uint32_t n_streams, int n_threads, uint8_t* bytes_out, uint32_t bytes_per_frame, int32_t* iters_out,
                         uint8_t* ok_out, uint8_t* found_out, uint32_t* sync_offset_out) {
    StderrMute mute;                                   // once, around all threads (it redirects the process's fd 2)
    const LogLevel saved_level = g_log_level;
    setLogLevel(LogLevel::WARN);                       // as the reference's harnesses do before they run trials
    if (n_threads < 1) n_threads = 1;
    std::vector<std::thread> th;
    std::vector<int> rc((size_t)n_threads, 0);
    for (int t = 0; t < n_threads; ++t) {
        const uint32_t n0 = (uint32_t)((uint64_t)n_streams * t / n_threads), n1 = (uint32_t)((uint64_t)n_streams * (t + 1) / n_threads);
        th.emplace_back([=, &rc] {
            ModemConfig cfg = to_cfg(c);
            LDPCDecoder dec(cfg.code_rate);
            dec.setMaxIterations((int)c->max_iterations);
            for (uint32_t f = n0; f < n1; ++f) {
                const float* a = audio + (size_t)f * stream_stride;
                OFDMDemodulator demod(cfg);
                for (uint32_t i = 0; i < n_samples; i += chunk) demod.process(SampleSpan(a + i, std::min(chunk, n_samples - i)));
                auto& sb = demod.impl_->soft_bits;
                if (sync_offset_out) sync_offset_out[f] = (uint32_t)demod.getLastSyncOffset();
                if (sb.size() < 648) {
                    found_out[f] = 0; ok_out[f] = 0; iters_out[f] = 0;
                    std::memset(bytes_out + (size_t)f * bytes_per_frame, 0, bytes_per_frame);
                    continue;
                }
                Bytes r = dec.decodeSoft(std::span<const float>(sb.data(), 648));
                if (r.size() != bytes_per_frame) { rc[(size_t)t] = -1; return; }
                std::memcpy(bytes_out + (size_t)f * bytes_per_frame, r.data(), r.size());
                iters_out[f] = dec.lastIterations();
                ok_out[f] = dec.lastDecodeSuccess() ? 1 : 0;
                found_out[f] = 1;
            }
        });
    }
    for (auto& x : th) x.join();
    setLogLevel(saved_level);
    for (int v : rc) if (v) return v;
    return 0;
}

// As ref_demod_process, and additionally recovers the COARSE CFO the search stage set on
// the sync transition (demodulator.cpp:533-537).  getFrequencyOffset() after the run is the
// tracked value; the coarse one is a pure function of the buffered audio at the moment of
// sync (estimateCoarseCFO, ofdm_sync.cpp:229-261) — the buffer is untrimmed for frames this
// short — so it is re-evaluated on a second demodulator holding the same samples.
int ref_demod_process_coarse(const ultra_hip_config* c, const float* audio, uint32_t n, uint32_t chunk,
                             float* llr_out, uint32_t cap, uint32_t* sync_offset, float* coarse_cfo,
                             float* final_cfo, uint32_t* fed_at_sync) {
    StderrMute mute;
    ModemConfig cfg = to_cfg(c);
    OFDMDemodulator demod(cfg);
    uint32_t fed = 0, fed_sync = 0;
    bool was = false;
    for (uint32_t i = 0; i < n; i += chunk) {
        uint32_t len = std::min(chunk, n - i);
        demod.process(SampleSpan(audio + i, len));
        fed = i + len;
        if (!was && demod.isSynced()) { was = true; fed_sync = fed; }
    }
    *sync_offset = (uint32_t)demod.getLastSyncOffset();
    *final_cfo = demod.getFrequencyOffset();
    *fed_at_sync = fed_sync;
    *coarse_cfo = 0.0f;
    if (was) {
        OFDMDemodulator probe(cfg);
        probe.impl_->rx_buffer.assign(audio, audio + fed_sync);
        *coarse_cfo = probe.impl_->estimateCoarseCFO(*sync_offset);
    }
    auto& sb = demod.impl_->soft_bits;
    uint32_t m = (uint32_t)std::min<size_t>(sb.size(), cap);
    std::memcpy(llr_out, sb.data(), m * sizeof(float));
    return (int)sb.size();
}

// A LIVE stream through the reference's Schmidl-Cox waveform, call by call: what OFDMNvisWaveform::process
// (src/waveform/ofdm_cox_waveform.cpp:122-134) does — demodulator.process(chunk); when it reports a codeword,
// getSoftBits() hands out 648 soft bits — for a list of chunk lengths (0 = an empty call).  Records per call the
// return value, isSynced() afterwards and how many soft bits were handed out; the soft bits are concatenated.
// Exercises the ways out of SYNCED (timeout, idle calls, frame complete: demodulator.cpp:683-732) and re-acquisition.
int ref_demod_stream(const ultra_hip_config* c, const float* audio, const uint32_t* chunks, uint32_t n_calls,
                     uint8_t* ready_out, uint8_t* synced_out, uint32_t* drained_out, float* soft_out, uint32_t cap) {
    StderrMute mute;
    ModemConfig cfg = to_cfg(c);
    OFDMDemodulator demod(cfg);
    size_t pos = 0, total = 0;
    for (uint32_t i = 0; i < n_calls; ++i) {
        const bool ready = demod.process(SampleSpan(audio + pos, chunks[i]));
        pos += chunks[i];
        uint32_t drained = 0;
        if (ready) {
            std::vector<float> sb = demod.getSoftBits();
            drained = (uint32_t)sb.size();
            for (float v : sb) { if (total < cap) soft_out[total] = v; ++total; }
        }
        ready_out[i] = ready ? 1 : 0;
        synced_out[i] = demod.isSynced() ? 1 : 0;
        drained_out[i] = drained;
    }
    return (int)total;
}

// The preamble check of the SYNCED state (src/ofdm/demodulator.cpp:605-657) on rx_buffer = audio[0, n): a demodulator is
// put into the state that arms it (SYNCED, symbols demodulated, two idle calls) and process() is called ONCE with the
// whole buffer.  found / consume are read off what process() did: a detection resets synced_symbol_count (set to 100
// beforehand) and erases `consume` samples before the symbol loop takes whole symbols.  The Schmidl-Cox offset and the
// coarse CFO it was declared with are re-derived on a probe demodulator holding the same buffer with the reference's own
// Impl::measureCorrelation / refineLTSTiming / estimateCoarseCFO (the symbol loop has moved freq_offset_hz on by then).
int ref_midframe_search(const ultra_hip_config* c, const float* audio, uint32_t n, uint32_t* found, uint32_t* sts_start,
                        uint32_t* refined_lts, uint32_t* consume, float* coarse_cfo) {
    StderrMute mute;
    ModemConfig cfg = to_cfg(c);
    *found = 0; *sts_start = 0; *refined_lts = 0; *consume = 0; *coarse_cfo = 0;
    OFDMDemodulator demod(cfg);
    demod.impl_->state.store(OFDMDemodulator::Impl::State::SYNCED);
    demod.impl_->synced_symbol_count.store(100);
    demod.impl_->idle_call_count.store(2);
    demod.process(SampleSpan(audio, n));
    const int count = demod.impl_->synced_symbol_count.load();
    const size_t sym = demod.impl_->symbol_samples, psl = cfg.fft_size + cfg.getCyclicPrefix();
    if (count >= 100) return 0;                                       // no detection: the loop went on counting
    *found = 1;
    *consume = (uint32_t)(n - demod.impl_->rx_buffer.size() - (size_t)count * sym);
    *refined_lts = *consume - (uint32_t)(2 * psl);
    OFDMDemodulator probe(cfg);
    probe.impl_->rx_buffer.assign(audio, audio + n);
    for (size_t offset = 0; offset + 6 * psl <= n; offset += 8) {
        if (probe.impl_->measureCorrelation(offset) > probe.impl_->sync_threshold &&
            probe.impl_->refineLTSTiming(offset) == (size_t)*refined_lts) {
            *sts_start = (uint32_t)offset;
            *coarse_cfo = probe.impl_->estimateCoarseCFO(offset);
            break;
        }
    }
    return 0;
}

// Acquisition (scope row f1): OFDMDemodulator::process in the SEARCHING state, fed in `chunk`-sample
// calls (src/ofdm/demodulator.cpp:461-600).  Reports, per stream: whether sync was declared, after how
// many fed samples, the Schmidl-Cox offset (last_sync_offset, relative to the buffer at that call), the
// coarse CFO and the refined LTS start it was declared with, and the absolute data_start (what process()
// erases the buffer up to, :572).  The buffer's absolute start is tracked from rx_buffer.size() after
// every SEARCHING call (trims of :524-531,:592-597 and the overflow rule :482-487); coarse CFO and
// refined LTS start are re-derived on a probe demodulator holding the same buffer, calling the
// reference's own Impl::estimateCoarseCFO / refineLTSTiming.
int ref_demod_acquire(const ultra_hip_config* c, const float* audio, uint32_t n, uint32_t chunk,
                      uint32_t* found, uint32_t* fed_at_sync, uint32_t* sync_offset, float* coarse_cfo,
                      uint32_t* refined_lts, uint32_t* data_start, float* noise_floor) {
    StderrMute mute;
    ModemConfig cfg = to_cfg(c);
    OFDMDemodulator demod(cfg);
    *found = 0; *fed_at_sync = 0; *sync_offset = 0; *coarse_cfo = 0; *refined_lts = 0; *data_start = 0;
    size_t base = 0;                                   // absolute index of rx_buffer[0] before the next call
    for (uint32_t i = 0; i < n; i += chunk) {
        uint32_t len = std::min(chunk, n - i);
        const size_t fed = (size_t)i + len;
        demod.process(SampleSpan(audio + i, len));
        if (demod.isSynced()) {
            *found = 1; *fed_at_sync = (uint32_t)fed;
            if (fed - base > 240000) base = fed - 20000;           // MAX_BUFFER_SAMPLES / OVERLAP_SAMPLES at the top of the call
            break;
        }
        base = fed - demod.impl_->rx_buffer.size();
    }
    *noise_floor = demod.impl_->noise_floor_energy;
    if (*found) {
        *sync_offset = (uint32_t)demod.getLastSyncOffset();
        OFDMDemodulator probe(cfg);
        probe.impl_->rx_buffer.assign(audio + base, audio + *fed_at_sync);
        *coarse_cfo = probe.impl_->estimateCoarseCFO(*sync_offset);
        size_t r = probe.impl_->refineLTSTiming(*sync_offset);
        *refined_lts = (uint32_t)r;
        *data_start = (uint32_t)(base + r + 2 * (cfg.fft_size + cfg.getCyclicPrefix()));
    }
    return 0;
}

// One Schmidl-Cox metric (Impl::measureSchmidlCoxCorrelation, ofdm_sync.cpp:120-163) and the energy
// gate (Impl::hasMinimumEnergy, :20-50, stateful: noise_floor in/out) at `offset` of a buffer.
int ref_sc_metric(const ultra_hip_config* c, const float* audio, uint32_t n, uint32_t offset,
                  float* corr, float* p_re, float* p_im, float* energy, float* noise_floor_io, uint32_t* has_energy) {
    StderrMute mute;
    ModemConfig cfg = to_cfg(c);
    OFDMDemodulator probe(cfg);
    probe.impl_->rx_buffer.assign(audio, audio + n);
    Complex P(0, 0); float e = 0;
    *corr = probe.impl_->measureSchmidlCoxCorrelation(offset, &P, &e);
    *p_re = P.real(); *p_im = P.imag(); *energy = e;
    probe.impl_->noise_floor_energy = *noise_floor_io;
    size_t win = 2 * (cfg.fft_size + cfg.getCyclicPrefix());
    *has_energy = probe.impl_->hasMinimumEnergy(offset, win) ? 1 : 0;
    *noise_floor_io = probe.impl_->noise_floor_energy;
    return 0;
}

// LTS passband templates of the constructor (demodulator.cpp:100-133)
int ref_lts_templates(const ultra_hip_config* c, float* I, float* Q, uint32_t cap) {
    StderrMute mute;
    OFDMDemodulator probe(to_cfg(c));
    uint32_t m = (uint32_t)probe.impl_->lts_passband_I.size();
    if (m > cap) return -1;
    std::memcpy(I, probe.impl_->lts_passband_I.data(), m * sizeof(float));
    std::memcpy(Q, probe.impl_->lts_passband_Q.data(), m * sizeof(float));
    return (int)m;
}

// Chirp synchronisation (scope row f4): sync::ChirpSync with OFDMChirpWaveform's configuration
// (ofdm_chirp_waveform.cpp:39-49) and OFDMChirpWaveform::detectSync's start_sample arithmetic (:129-172).
static sync::ChirpConfig chirp_cfg(float sample_rate, float tx_cfo) {
    sync::ChirpConfig cfg;
    cfg.sample_rate = sample_rate; cfg.f_start = 300.0f; cfg.f_end = 2700.0f; cfg.duration_ms = 500.0f;
    cfg.gap_ms = 100.0f; cfg.use_dual_chirp = true; cfg.tx_cfo_hz = tx_cfo;
    return cfg;
}
int ref_chirp_detect(float sample_rate, const float* x, uint32_t n, float threshold, int32_t* out, float* fout) {
    StderrMute mute;
    fflush(stdout); int saved = dup(1); int nul = open("/dev/null", O_WRONLY); dup2(nul, 1); close(nul);   // detectDualChirp printf()s
    sync::ChirpSync cs(chirp_cfg(sample_rate, 0.0f));
    auto r = cs.detectDualChirp(SampleSpan(x, n), threshold);
    fflush(stdout); dup2(saved, 1); close(saved);
    out[0] = r.success ? 1 : 0; out[1] = r.success ? r.up_chirp_start : -1; out[2] = r.success ? r.down_chirp_start : -1;
    out[3] = -1; out[4] = -1; out[5] = -1;
    fout[0] = r.cfo_hz; fout[1] = r.up_correlation; fout[2] = r.down_correlation;
    if (r.success) {
        size_t chirp_samples = cs.getChirpSamples();
        size_t gap_samples = static_cast<size_t>((uint32_t)sample_rate * 100.0f / 1000.0f);
        out[3] = (int)(r.down_chirp_start + chirp_samples + gap_samples);
    }
    return 0;
}
int ref_chirp_generate(float sample_rate, float tx_cfo_hz, float* out, uint32_t cap) {
    StderrMute mute;
    sync::ChirpSync cs(chirp_cfg(sample_rate, tx_cfo_hz));
    Samples s = cs.generate();
    if (s.size() > cap) return -1;
    std::memcpy(out, s.data(), s.size() * sizeof(float));
    return (int)s.size();
}
int ref_chirp_templates(float sample_rate, float* up_s, float* up_c, float* dn_s, float* dn_c, float* energies, uint32_t cap) {
    StderrMute mute;
    sync::ChirpSync cs(chirp_cfg(sample_rate, 0.0f));
    size_t m = cs.up_chirp_template_.size();
    if (m > cap) return -1;
    std::memcpy(up_s, cs.up_chirp_template_.data(), 4 * m); std::memcpy(up_c, cs.up_chirp_template_cos_.data(), 4 * m);
    std::memcpy(dn_s, cs.down_chirp_template_.data(), 4 * m); std::memcpy(dn_c, cs.down_chirp_template_cos_.data(), 4 * m);
    energies[0] = cs.template_energy_; energies[1] = cs.down_template_energy_;
    return (int)m;
}

// Per-symbol stage dump layout (floats), see ref_demod_synced:
//   bb      [symbol_samples][2]
//   freq    [fft][2]
//   H       [fft][2]   channel_estimate after updateChannelEstimate
//   eq      [n_data][2]
//   nv      [n_data]   carrier_noise_var
//   scal    [8]        freq_offset_hz, noise_variance, estimated_snr_linear,
//                      timing_offset_samples, freq_correction_phase,
//                      pilot_phase_correction.re/.im, snr_symbol_count

// SYNCED-entry symbol loop driven stage by stage on a freshly constructed
// demodulator put into the state process() leaves it in on sync
// (src/ofdm/demodulator.cpp:533-591): freq_offset_hz=filtered=cfo,
// freq_correction_phase=0, symbols_since_sync=0, mixer.reset(), state=SYNCED.
// The loop body is the reference's own (demodulator.cpp:672-697), calling the
// reference's Impl member functions.
int ref_demod_synced(const ultra_hip_config* c, const float* audio, uint32_t n_symbols, float cfo_hz,
                     float* llr_out, uint32_t llr_cap, float* stage_out /*nullable*/) {
    StderrMute mute;
    ModemConfig cfg = to_cfg(c);
    OFDMDemodulator demod(cfg);
    auto* im = demod.impl_.get();
    im->freq_offset_hz = cfo_hz;
    im->freq_offset_filtered = cfo_hz;
    im->freq_correction_phase = 0.0f;
    im->symbols_since_sync = 0;
    im->state.store(OFDMDemodulator::Impl::State::SYNCED);
    im->synced_symbol_count.store(0);
    im->mixer.reset();
    im->dbpsk_prev_equalized.clear();
    im->carrier_phase_initialized = false;
    im->carrier_phase_correction = Complex(1, 0);
    im->dqpsk_skip_first_symbol = false;
    im->timing_offset_samples = 0.0f;

    const size_t S = im->symbol_samples;
    const size_t N = cfg.fft_size;
    const size_t nd = im->data_carrier_indices.size();
    float* st = stage_out;
    for (uint32_t s = 0; s < n_symbols; ++s) {
        SampleSpan sym(audio + (size_t)s * S, S);
        auto bb = im->toBaseband(sym);
        auto fd = im->extractSymbol(bb, 0);
        im->updateChannelEstimate(fd);
        auto eq = im->equalize(fd);
        im->demodulateSymbol(eq, cfg.modulation);
        if (st) {
            for (size_t i = 0; i < S; ++i) { *st++ = bb[i].real(); *st++ = bb[i].imag(); }
            for (size_t i = 0; i < N; ++i) { *st++ = fd[i].real(); *st++ = fd[i].imag(); }
            for (size_t i = 0; i < N; ++i) { *st++ = im->channel_estimate[i].real(); *st++ = im->channel_estimate[i].imag(); }
            for (size_t i = 0; i < nd; ++i) { *st++ = eq[i].real(); *st++ = eq[i].imag(); }
            for (size_t i = 0; i < nd; ++i) *st++ = im->carrier_noise_var[i];
            *st++ = im->freq_offset_hz;
            *st++ = im->noise_variance;
            *st++ = im->estimated_snr_linear;
            *st++ = im->timing_offset_samples;
            *st++ = im->freq_correction_phase;
            *st++ = im->pilot_phase_correction.real();
            *st++ = im->pilot_phase_correction.imag();
            *st++ = (float)im->snr_symbol_count;
        }
    }
    auto& sb = im->soft_bits;
    uint32_t m = (uint32_t)std::min<size_t>(sb.size(), llr_cap);
    std::memcpy(llr_out, sb.data(), m * sizeof(float));
    return (int)sb.size();
}

// CPU baseline of kind "reference": the reference's own classes over a batch of SYNCED-entry
// frames on ONE thread — one OFDMDemodulator and one LDPCDecoder for the whole run (the tools
// construct them per trial, tools/test_nvis_mode.cpp:43-46; reusing them only removes
// constructor time), state re-forced per frame as in ref_demod_synced, then the first 648 soft
// bits decoded (tools/test_nvis_mode.cpp:96-103).
int ref_demod_decode_batch(const ultra_hip_config* c, const float* audio, size_t frame_stride,
                           const float* cfo_hz, uint32_t n_frames, uint8_t* bytes_out, uint32_t bytes_per_frame,
                           int32_t* iters_out, uint8_t* ok_out) {
    StderrMute mute;
    ModemConfig cfg = to_cfg(c);
    OFDMDemodulator demod(cfg);
    LDPCDecoder dec(cfg.code_rate);
    dec.setMaxIterations((int)c->max_iterations);
    auto* im = demod.impl_.get();
    const size_t S = im->symbol_samples;
    for (uint32_t f = 0; f < n_frames; ++f) {
        const float* a = audio + (size_t)f * frame_stride;
        float cfo = cfo_hz ? cfo_hz[f] : 0.0f;
        demod.reset();
        im->freq_offset_hz = cfo; im->freq_offset_filtered = cfo; im->freq_correction_phase = 0.0f;
        im->symbols_since_sync = 0;
        im->state.store(OFDMDemodulator::Impl::State::SYNCED);
        im->carrier_phase_initialized = false; im->carrier_phase_correction = Complex(1, 0);
        im->timing_offset_samples = 0.0f;
        for (uint32_t s = 0; s < c->n_data_symbols; ++s) {
            auto bb = im->toBaseband(SampleSpan(a + (size_t)s * S, S));
            auto fd = im->extractSymbol(bb, 0);
            im->updateChannelEstimate(fd);
            auto eq = im->equalize(fd);
            im->demodulateSymbol(eq, cfg.modulation);
        }
        if (im->soft_bits.size() < 648) { ok_out[f] = 0; iters_out[f] = 0; std::memset(bytes_out + (size_t)f * bytes_per_frame, 0, bytes_per_frame); continue; }
        Bytes r = dec.decodeSoft(std::span<const float>(im->soft_bits.data(), 648));
        if (r.size() != bytes_per_frame) return -1;
        std::memcpy(bytes_out + (size_t)f * bytes_per_frame, r.data(), r.size());
        iters_out[f] = dec.lastIterations();
        ok_out[f] = dec.lastDecodeSuccess() ? 1 : 0;
    }
    return 0;
}

// The two batch entries over worker threads — one OFDMDemodulator / LDPCDecoder per thread, disjoint frame ranges: the
// reference's CPU path timed on all host cores (bench.py's cpu_baseline, kind "reference").
int ref_demod_decode_batch_mt(const ultra_hip_config* c, const float* audio, size_t frame_stride, const float* cfo_hz,
                              uint32_t n_frames, int n_threads, uint8_t* bytes_out, uint32_t bytes_per_frame,
                              int32_t* iters_out, uint8_t* ok_out) {
    StderrMute mute;                                   // once, around all threads (it redirects the process's fd 2)
    // as the reference's own harnesses do before they run trials (tools/test_mode_snr.cpp:123, tools/test_nvis_mode.cpp:
    // 132): without it every LOG_* call formats its message under one global mutex (include/ultra/logging.hpp:58-66)
    const LogLevel saved_level = g_log_level;
    setLogLevel(LogLevel::WARN);
    if (n_threads < 1) n_threads = 1;
    std::vector<std::thread> th;
    std::vector<int> rc((size_t)n_threads, 0);
    for (int t = 0; t < n_threads; ++t) {
        const uint32_t n0 = (uint32_t)((uint64_t)n_frames * t / n_threads), n1 = (uint32_t)((uint64_t)n_frames * (t + 1) / n_threads);
        th.emplace_back([=, &rc] {
            ModemConfig cfg = to_cfg(c);
            OFDMDemodulator demod(cfg);
            LDPCDecoder dec(cfg.code_rate);
            dec.setMaxIterations((int)c->max_iterations);
            auto* im = demod.impl_.get();
            const size_t S = im->symbol_samples;
            for (uint32_t f = n0; f < n1; ++f) {
                const float* a = audio + (size_t)f * frame_stride;
                float cfo = cfo_hz ? cfo_hz[f] : 0.0f;
                demod.reset();
                im->freq_offset_hz = cfo; im->freq_offset_filtered = cfo; im->freq_correction_phase = 0.0f;
                im->symbols_since_sync = 0;
                im->state.store(OFDMDemodulator::Impl::State::SYNCED);
                im->carrier_phase_initialized = false; im->carrier_phase_correction = Complex(1, 0);
                im->timing_offset_samples = 0.0f;
                for (uint32_t s = 0; s < c->n_data_symbols; ++s) {
                    auto bb = im->toBaseband(SampleSpan(a + (size_t)s * S, S));
                    auto fd = im->extractSymbol(bb, 0);
                    im->updateChannelEstimate(fd);
                    auto eq = im->equalize(fd);
                    im->demodulateSymbol(eq, cfg.modulation);
                }
                if (im->soft_bits.size() < 648) { ok_out[f] = 0; iters_out[f] = 0; std::memset(bytes_out + (size_t)f * bytes_per_frame, 0, bytes_per_frame); continue; }
                Bytes r = dec.decodeSoft(std::span<const float>(im->soft_bits.data(), 648));
                if (r.size() != bytes_per_frame) { rc[(size_t)t] = -1; return; }
                std::memcpy(bytes_out + (size_t)f * bytes_per_frame, r.data(), r.size());
                iters_out[f] = dec.lastIterations();
                ok_out[f] = dec.lastDecodeSuccess() ? 1 : 0;
            }
        });
    }
    for (auto& x : th) x.join();
    setLogLevel(saved_level);
    for (int v : rc) if (v) return v;
    return 0;
}

int ref_ldpc_decode_batch_mt(uint32_t rate, int max_iters, const float* llr, uint32_t n_cw, int n_threads,
                             uint8_t* out, uint32_t bytes_per_cw, int32_t* iters, uint8_t* ok) {
    if (n_threads < 1) n_threads = 1;
    std::vector<std::thread> th;
    std::vector<int> rc((size_t)n_threads, 0);
    for (int t = 0; t < n_threads; ++t) {
        const uint32_t n0 = (uint32_t)((uint64_t)n_cw * t / n_threads), n1 = (uint32_t)((uint64_t)n_cw * (t + 1) / n_threads);
        th.emplace_back([=, &rc] {
            rc[(size_t)t] = ref_ldpc_decode_batch(rate, max_iters, llr + 648ull * n0, n1 - n0, out + (size_t)bytes_per_cw * n0,
                                                  bytes_per_cw, iters + n0, ok + n0);
        });
    }
    for (auto& x : th) x.join();
    for (int v : rc) if (v) return v;
    return 0;
}

// Same entry but through the public API only: state forced to SYNCED, then
// ONE process() call with all data symbols (cross-check of the stage driver).
int ref_demod_synced_public(const ultra_hip_config* c, const float* audio, uint32_t n_symbols,
                            float cfo_hz, float* llr_out, uint32_t llr_cap) {
    StderrMute mute;
    ModemConfig cfg = to_cfg(c);
    OFDMDemodulator demod(cfg);
    auto* im = demod.impl_.get();
    im->freq_offset_hz = cfo_hz;
    im->freq_offset_filtered = cfo_hz;
    im->state.store(OFDMDemodulator::Impl::State::SYNCED);
    demod.process(SampleSpan(audio, (size_t)n_symbols * im->symbol_samples));
    auto& sb = im->soft_bits;
    uint32_t m = (uint32_t)std::min<size_t>(sb.size(), llr_cap);
    std::memcpy(llr_out, sb.data(), m * sizeof(float));
    return (int)sb.size();
}

// OFDMDemodulator::setFrequencyOffset (src/ofdm/demodulator.cpp:805-814) BETWEEN two process() calls of a frame in the
// SYNCED state: symbols [0, set_at) with the offset the frame started with (none when has_cfo0 == 0), then the new offset
// from symbol set_at on (correction phase restarts at 0).
int ref_demod_synced_setcfo(const ultra_hip_config* c, const float* audio, uint32_t n_symbols, int has_cfo0, float cfo0_hz,
                            uint32_t set_at, float cfo_new_hz, float* llr_out, uint32_t llr_cap) {
    StderrMute mute;
    ModemConfig cfg = to_cfg(c);
    OFDMDemodulator demod(cfg);
    auto* im = demod.impl_.get();
    if (has_cfo0) { im->freq_offset_hz = cfo0_hz; im->freq_offset_filtered = cfo0_hz; }
    im->state.store(OFDMDemodulator::Impl::State::SYNCED);
    if (set_at > n_symbols) set_at = n_symbols;
    if (set_at > 0) demod.process(SampleSpan(audio, (size_t)set_at * im->symbol_samples));
    demod.setFrequencyOffset(cfo_new_hz);
    if (n_symbols > set_at)
        demod.process(SampleSpan(audio + (size_t)set_at * im->symbol_samples, (size_t)(n_symbols - set_at) * im->symbol_samples));
    auto& sb = im->soft_bits;
    uint32_t m = (uint32_t)std::min<size_t>(sb.size(), llr_cap);
    std::memcpy(llr_out, sb.data(), m * sizeof(float));
    return (int)sb.size();
}

// OFDMDemodulator::processPresynced (src/ofdm/demodulator.cpp:854-985) after
// setFrequencyOffsetWithPhase (:816-825) when has_cfo != 0.
int ref_demod_presynced(const ultra_hip_config* c, const float* audio, uint32_t n_samples,
                        int has_cfo, float cfo_hz, float cfo_phase,
                        float* llr_out, uint32_t llr_cap, float* H_out /*[fft][2] nullable*/,
                        float* scal_out /*[8] nullable*/) {
    StderrMute mute;
    ModemConfig cfg = to_cfg(c);
    OFDMDemodulator demod(cfg);
    if (has_cfo) demod.setFrequencyOffsetWithPhase(cfo_hz, cfo_phase);
    demod.processPresynced(SampleSpan(audio, n_samples), (int)c->training_symbols);
    auto* im = demod.impl_.get();
    if (H_out) {
        for (size_t i = 0; i < cfg.fft_size; ++i) {
            H_out[2 * i] = im->channel_estimate[i].real();
            H_out[2 * i + 1] = im->channel_estimate[i].imag();
        }
    }
    if (scal_out) {
        scal_out[0] = im->freq_offset_hz;
        scal_out[1] = im->noise_variance;
        scal_out[2] = im->estimated_snr_linear;
        scal_out[3] = im->timing_offset_samples;
        scal_out[4] = im->freq_correction_phase;
        scal_out[5] = im->pilot_phase_correction.real();
        scal_out[6] = im->pilot_phase_correction.imag();
        scal_out[7] = (float)im->snr_symbol_count;
    }
    auto& sb = im->soft_bits;
    uint32_t m = (uint32_t)std::min<size_t>(sb.size(), llr_cap);
    std::memcpy(llr_out, sb.data(), m * sizeof(float));
    return (int)sb.size();
}

// Constant tables of a constructed demodulator (carrier map, pilot signs,
// interpolation table, Zadoff-Chu) for pinning the restated constructors.
int ref_demod_tables(const ultra_hip_config* c, int32_t* data_idx, int32_t* pilot_idx,
                     float* pilot_seq_ri, int32_t* interp_i /*[nd][3]*/, float* interp_alpha,
                     float* sync_seq_ri, uint32_t* counts /*[4]: nd, np, n_interp, n_sync*/) {
    StderrMute mute;
    ModemConfig cfg = to_cfg(c);
    OFDMDemodulator demod(cfg);
    auto* im = demod.impl_.get();
    counts[0] = (uint32_t)im->data_carrier_indices.size();
    counts[1] = (uint32_t)im->pilot_carrier_indices.size();
    counts[2] = (uint32_t)im->interp_table.size();
    counts[3] = (uint32_t)im->sync_sequence.size();
    for (size_t i = 0; i < counts[0]; ++i) data_idx[i] = im->data_carrier_indices[i];
    for (size_t i = 0; i < counts[1]; ++i) {
        pilot_idx[i] = im->pilot_carrier_indices[i];
        pilot_seq_ri[2 * i] = im->pilot_sequence[i].real();
        pilot_seq_ri[2 * i + 1] = im->pilot_sequence[i].imag();
    }
    for (size_t i = 0; i < counts[2]; ++i) {
        interp_i[3 * i] = im->interp_table[i].fft_idx;
        interp_i[3 * i + 1] = im->interp_table[i].lower_pilot;
        interp_i[3 * i + 2] = im->interp_table[i].upper_pilot;
        interp_alpha[i] = im->interp_table[i].alpha;
    }
    for (size_t i = 0; i < counts[3]; ++i) {
        sync_seq_ri[2 * i] = im->sync_sequence[i].real();
        sync_seq_ri[2 * i + 1] = im->sync_sequence[i].imag();
    }
    return 0;
}

// End-to-end harness body of tools/test_nvis_mode.cpp:35-114 restated as
// calls (payload in → ok/iters/decoded out), AWGN drawn exactly as there.
int ref_harness_awgn(const ultra_hip_config* c, const uint8_t* payload, uint32_t n_payload,
                     float snr_db, uint32_t noise_seed, float* audio_out, uint32_t cap,
                     uint32_t* n_audio) {
    StderrMute mute;
    ModemConfig cfg = to_cfg(c);
    OFDMModulator mod(cfg);
    LDPCEncoder enc(cfg.code_rate);
    Bytes encoded = enc.encode(ByteSpan(payload, n_payload));
    Samples pre = mod.generatePreamble();
    Samples dat = mod.modulate(encoded, cfg.modulation);
    Samples signal;
    signal.insert(signal.end(), pre.begin(), pre.end());
    signal.insert(signal.end(), dat.begin(), dat.end());
    float max_val = 0;
    for (float s : signal) max_val = std::max(max_val, std::abs(s));
    for (float& s : signal) s *= 0.5f / max_val;
    if (snr_db < 100.0f) {
        std::mt19937 rng(noise_seed);
        float sp = 0;
        for (float s : signal) sp += s * s;
        sp /= signal.size();
        float noise_std = std::sqrt(sp / std::pow(10.0f, snr_db / 10.0f));
        std::normal_distribution<float> noise(0.0f, noise_std);
        for (float& s : signal) s += noise(rng);
    }
    if (signal.size() > cap) return -1;
    std::memcpy(audio_out, signal.data(), signal.size() * sizeof(float));
    *n_audio = (uint32_t)signal.size();
    return (int)pre.size();
}


// ---- v2 wire format (scope row f4): RxPipeline::processFrame from the soft bits on ----
uint16_t ref_crc16(const uint8_t* d, uint32_t n) { return protocol::v2::ControlFrame::calculateCRC(d, n); }

int ref_v2_parse_header(const uint8_t* d, uint32_t n, int32_t* out) {
    auto h = protocol::v2::parseHeader(Bytes(d, d + n));
    out[0] = (int)h.type; out[1] = h.total_cw; out[2] = h.payload_len; out[3] = h.is_control ? 1 : 0;
    return h.valid ? 1 : 0;
}

// the statements of processFrame (rx_pipeline.cpp:283-346) after waveform->getSoftBits(), on a fresh pipeline
int ref_v2_decode_frame(uint32_t rate, uint32_t deint_bps, int max_iters, const float* soft, uint32_t n_soft,
                        int32_t* res, uint8_t* frame_data, uint32_t cap) {
    StderrMute mute;
    namespace v2 = protocol::v2;
    (void)max_iters;                                             // RxPipeline uses LDPCDecoder's default (50)
    gui::RxPipeline rx("T");
    rx.setDataMode(static_cast<CodeRate>(rate), true);           // connected: data_code_rate_ for every codeword
    // deint_bps = 0xffffffff leaves the pipeline as constructed (ChannelInterleaver(60, 648), interleaving on:
    // rx_pipeline.cpp:13-18, rx_pipeline.hpp:177,182); otherwise exactly the two public setters
    if (deint_bps != 0xffffffffu) {
        rx.setInterleavingEnabled(deint_bps != 0);
        if (deint_bps) rx.setInterleaverConfig(deint_bps);
    }
    for (int i = 0; i < 8; ++i) res[i] = 0;
    gui::RxFrameResult result;
    res[2] = (int)result.frame_type;
    std::vector<float> soft_bits(soft, soft + n_soft);
    if (soft_bits.empty()) return 0;
    if (rx.detectPing(soft_bits)) { res[0] = 1; res[1] = 1; res[2] = (int)v2::FrameType::PING; res[7] = 5; return 0; }
    if (rx.interleaving_enabled_) soft_bits = rx.deinterleaveCodewords(soft_bits);
    int num_codewords = static_cast<int>(soft_bits.size() / v2::LDPC_CODEWORD_BITS);
    if (num_codewords == 0) return 0;
    result = rx.decodeFrame(soft_bits, num_codewords);
    res[0] = result.success; res[1] = result.is_ping; res[2] = (int)result.frame_type;
    res[3] = result.codewords_ok; res[4] = result.codewords_failed; res[5] = rx.getExpectedCodewords();
    res[6] = (int)result.frame_data.size();
    res[7] = result.success ? 4 : res[5] > 0 ? 2 : result.codewords_ok == 0 ? 0 : (result.codewords_failed > 0 ? 3 : 1);
    if (result.frame_data.size() > cap) return -1;
    std::memcpy(frame_data, result.frame_data.data(), result.frame_data.size());
    return 0;
}

int ref_v2_build_frame(uint32_t rate, uint8_t type, uint8_t flags, uint16_t seq, uint32_t src_hash, uint32_t dst_hash,
                       const uint8_t* payload, uint32_t payload_len, int total_cw_override, uint8_t* codewords,
                       uint32_t cap_cw) {
    namespace v2 = protocol::v2;
    Bytes frame;
    if (v2::isControlFrame(static_cast<v2::FrameType>(type))) {
        v2::ControlFrame f;
        f.type = static_cast<v2::FrameType>(type); f.flags = flags; f.seq = seq; f.src_hash = src_hash; f.dst_hash = dst_hash;
        std::memset(f.payload, 0, sizeof(f.payload));
        std::memcpy(f.payload, payload, std::min<size_t>(payload_len, sizeof(f.payload)));
        frame = f.serialize();
    } else {
        v2::DataFrame f;
        f.type = static_cast<v2::FrameType>(type); f.flags = flags; f.seq = seq; f.src_hash = src_hash; f.dst_hash = dst_hash;
        f.payload.assign(payload, payload + payload_len);
        f.payload_len = static_cast<uint16_t>(payload_len);
        f.total_cw = total_cw_override >= 0 ? static_cast<uint8_t>(total_cw_override)
                                            : v2::DataFrame::calculateCodewords(payload_len, static_cast<CodeRate>(rate));
        frame = f.serialize();
    }
    auto cws = v2::encodeFrameWithLDPC(frame, static_cast<CodeRate>(rate));
    if (cws.size() > cap_cw) return -1;
    for (size_t i = 0; i < cws.size(); ++i) {
        if (cws[i].size() != 81) return -2;
        std::memcpy(codewords + i * 81, cws[i].data(), 81);
    }
    return (int)cws.size();
}

}  // extern "C"
