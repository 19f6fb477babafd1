// stimulus_kernel.h — batched transmit-side stimulus on the GPU (scope row f2).
//
// What one Monte-Carlo trial of the harnesses does before the receiver runs
// (tools/test_nvis_mode.cpp:35-93): random payload -> LDPCEncoder::encode
// (src/fec/ldpc_encoder.cpp:193-257) -> OFDMModulator::generatePreamble + modulate
// (src/ofdm/modulator.cpp:202-283,348-532) -> scale the whole signal to a 0.5 peak -> channel ->
// hand the receiver the samples from the first data symbol on.  One wavefront per frame.
//
// Bit-exact parts (checked against the test oracle's uo_make_batch, which is pinned to the
// compiled reference's modulator and encoder): the payload bytes (counter-based splitmix64, the
// oracle's generator), the encoded codewords, every transmitted sample and the peak scaling — so
// with channel "none" the audio equals the oracle's bit for bit.
// Statistical parts: the channels.  The reference draws its noise and fading from one serial
// mt19937 + normal_distribution stream per trial (src/sim/hf_channel.hpp:106-168,258-275); here
// every sample has its own counter-based generator and the first-order fading filters are
// evaluated chunk-parallel, so the realisations differ from any CPU run while the model (two
// taps, delay, per-tap Rayleigh magnitude from a one-pole filter restarted at (1,0) per frame,
// noise level from the RMS of the whole signal) is the same.  tests/test_gpu_stimulus.py checks
// noise power, fading statistics and receive-path error rates against the oracle's generator.
#ifndef ULTRA_STIMULUS_KERNEL_H
#define ULTRA_STIMULUS_KERNEL_H

#include <hip/hip_runtime.h>
#include "device_types.h"
#include "demod_kernel.h"
#include "acquire_kernel.h"

namespace ultra_hip {
namespace dev {

constexpr int kStimMaxCw = 8;                 // codewords modulated per frame
constexpr float kOutputScale = 40.0f;         // ModemConfig::output_scale default (types.hpp)

__device__ __forceinline__ unsigned long long splitmix_at(unsigned long long s0, unsigned long long n) {
    unsigned long long z = s0 + (n + 1ull) * 0x9E3779B97F4A7C15ull;       // state after n + 1 calls
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

template <int LOG2N>
struct StimShared {
    AcqShared<LOG2N> fft;                      // exchange buffer + twiddle runs of the inverse FFT
    unsigned char bits[kStimMaxCw * kLdpcN];   // encoded bits of the frame, codeword after codeword
    unsigned char raw[kStimMaxCw * 72];        // payload bytes
};

// mapBits (src/ofdm/modulator.cpp:13-108)
__device__ __forceinline__ c32 stim_map_bits(unsigned bits, int mod) {
    switch (mod) {
        case ULTRA_MOD_BPSK: return (bits & 1u) ? mk(1.0f, 0.0f) : mk(-1.0f, 0.0f);
        case ULTRA_MOD_QAM16: {
            const float S = 0.3162277660168379f;
            const unsigned a = (bits >> 2) & 3u, b = bits & 3u;     // levels {-3,-1,3,1}
            const float la = (a == 0) ? -3.0f : (a == 1) ? -1.0f : (a == 2) ? 3.0f : 1.0f;
            const float lb = (b == 0) ? -3.0f : (b == 1) ? -1.0f : (b == 2) ? 3.0f : 1.0f;
            return mk(la * S, lb * S);
        }
        case ULTRA_MOD_QAM32: {
            const float S = 0.1961161351381840f;
            const unsigned qb = (bits >> 2) & 7u, ib = bits & 3u;
            // inverse Gray maps: I_GRAY = {0,1,3,2}, Q_GRAY = {0,1,3,2,6,7,5,4}
            const int ii = (ib == 0) ? 0 : (ib == 1) ? 1 : (ib == 3) ? 2 : 3;
            const int qi = (qb == 0) ? 0 : (qb == 1) ? 1 : (qb == 3) ? 2 : (qb == 2) ? 3 : (qb == 6) ? 4 : (qb == 7) ? 5 : (qb == 5) ? 6 : 7;
            return mk((float)(2 * ii - 3) * S, (float)(2 * qi - 7) * S);
        }
        case ULTRA_MOD_QAM64: {
            const float S = 0.1543033499620919f;
            const float lv[8] = {-7, -5, -1, -3, 7, 5, 1, 3};
            float la = 0, lb = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) { if (((bits >> 3) & 7u) == (unsigned)i) la = lv[i]; if ((bits & 7u) == (unsigned)i) lb = lv[i]; }
            return mk(la * S, lb * S);
        }
        case ULTRA_MOD_QAM256: {
            const float S = 0.0645497224367903f;
            const float lv[16] = {-15, -13, -9, -11, -1, -3, -7, -5, 15, 13, 9, 11, 1, 3, 7, 5};
            float la = 0, lb = 0;
#pragma unroll
            for (int i = 0; i < 16; ++i) { if (((bits >> 4) & 15u) == (unsigned)i) la = lv[i]; if ((bits & 15u) == (unsigned)i) lb = lv[i]; }
            return mk(la * S, lb * S);
        }
        case ULTRA_MOD_QPSK:
        default: {
            const float QS = 0.7071067811865476f;
            return mk((bits & 2u) ? QS : -QS, (bits & 1u) ? QS : -QS);
        }
    }
}

// createOFDMSymbol + complexToReal (modulator.cpp:202-283): the frequency-domain symbol in
// sh.fft.X (natural order, padded) -> inverse FFT -> cyclic prefix + N samples, each multiplied by
// the running TX oscillator (real part) and by output_scale.  out (nullable) receives the cp + N
// samples; mx / sq accumulate max |x| and sum x^2 per lane.
template <int LOG2N>
__device__ __forceinline__ void stim_emit(AcqShared<LOG2N>& sh, const AcqLaneTw<LOG2N>& ltw, int cp,
                                          const c32* __restrict__ nco_tx, int nco_base, float* __restrict__ out,
                                          float& mx, float& sq) {
    using S = AcqShared<LOG2N>;
    constexpr int N = S::N, P = S::P, A = S::A;
    const int lane = threadIdx.x;
    const int rl = (int)(__brev((unsigned)lane) >> 26);
    c32 v[P];
#pragma unroll
    for (int q = 0; q < P; ++q) { const int i = rl + 64 * bitrev_small<A>(q); v[q] = sh.X[i + (i >> A)]; }
    wave_sync();
    acq_fft<LOG2N, true>(sh, ltw, v);                        // v[t] = time-domain sample lane + 64*t
#pragma unroll
    for (int t = 0; t < P; ++t) {
        const int i = lane + 64 * t;
        {
            const int o = cp + i;
            const c32 osc = nco_tx[nco_base + o];
            const float x = (v[t].re * osc.re - v[t].im * osc.im) * kOutputScale;     // real part of td * osc
            if (out) out[o] = x;
            mx = fmaxf(mx, fabsf(x));
            sq += x * x;
        }
        if (i >= N - cp) {                                   // cyclic prefix: the last cp samples first
            const int o = i - (N - cp);
            const c32 osc = nco_tx[nco_base + o];
            const float x = (v[t].re * osc.re - v[t].im * osc.im) * kOutputScale;
            if (out) out[o] = x;
            mx = fmaxf(mx, fabsf(x));
            sq += x * x;
        }
    }
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <int LOG2N>
__device__ __forceinline__ void stim_load_tables(AcqShared<LOG2N>& sh, AcqLaneTw<LOG2N>& ltw, const c32* __restrict__ twiddle) {
    constexpr int P = AcqShared<LOG2N>::P, A = AcqShared<LOG2N>::A;
    const int lane = threadIdx.x;
    ltw.table = twiddle;
    for (int idx = lane; idx < AcqShared<LOG2N>::kTwB; idx += kWave) {
        const int sA = 31 - __clz(idx / P + 1);
        const int k = idx - P * ((1 << sA) - 1);
        sh.twB[idx] = twiddle[k << (LOG2N - 1 - (A + sA))];
    }
    if (lane < AcqShared<LOG2N>::kTwA) {
        const int s0 = 31 - __clz(lane + 1);
        sh.twA[lane] = twiddle[(lane - ((1 << s0) - 1)) << (LOG2N - 1 - s0)];
    }
    wave_sync();
}

// OFDMModulator::generatePreamble (modulator.cpp:479-532): N + cp zeros, the Schmidl-Cox STS four
// times (one pass through the mixer, copied), the LTS twice.  One wavefront, once per context.
// out: 7 * (N + cp) samples; stats[0] = max |x|, stats[1] = sum x^2 over the whole preamble.
template <int LOG2N>
__global__ __launch_bounds__(kWave) void preamble_kernel(const DemodConst* __restrict__ Dp, const c32* __restrict__ twiddle,
                                                         const c32* __restrict__ nco_tx, float* __restrict__ out,
                                                         float* __restrict__ stats) {
    using S = AcqShared<LOG2N>;
    constexpr int N = S::N, A = S::A;
    __shared__ S sh;
    const DemodConst& D = *Dp;
    const int lane = threadIdx.x;
    AcqLaneTw<LOG2N> ltw;
    stim_load_tables<LOG2N>(sh, ltw, twiddle);
    const int psl = N + D.cp;
    for (int i = lane; i < psl; i += kWave) out[i] = 0.0f;
    float mx = 0.0f, sq = 0.0f;
    // STS: data carriers on even bins carry the sync sequence (index = data carrier number)
    for (int i = lane; i < N; i += kWave) sh.X[i + (i >> A)] = mk(0.0f, 0.0f);
    wave_sync();
    if (lane < D.n_data) {
        const int bin = D.bin[D.data_slot[lane]];
        if (bin % 2 == 0) sh.X[bin + (bin >> A)] = D.sync_seq[lane % D.n_carriers];
    }
    wave_sync();
    float m1 = 0.0f, s1 = 0.0f;
    stim_emit<LOG2N>(sh, ltw, D.cp, nco_tx, 0, out + psl, m1, s1);
    // LTS: sync sequence on the data carriers + pilots
    for (int i = lane; i < N; i += kWave) sh.X[i + (i >> A)] = mk(0.0f, 0.0f);
    wave_sync();
    if (lane < D.n_data) { const int bin = D.bin[D.data_slot[lane]]; sh.X[bin + (bin >> A)] = D.sync_seq[lane % D.n_carriers]; }
    if (lane < D.n_pilot) { const int bin = D.bin[D.pilot_slot[lane]]; sh.X[bin + (bin >> A)] = D.pilot_seq[lane]; }
    wave_sync();
    float m2 = 0.0f, s2 = 0.0f;
    stim_emit<LOG2N>(sh, ltw, D.cp, nco_tx, psl, out + 5 * psl, m2, s2);
    __threadfence_block();
    wave_sync();
    for (int r = 1; r < 4; ++r) for (int i = lane; i < psl; i += kWave) out[(1 + r) * psl + i] = out[psl + i];
    for (int i = lane; i < psl; i += kWave) out[6 * psl + i] = out[5 * psl + i];
    mx = fmaxf(wave_max(m1), wave_max(m2));
    sq = 4.0f * wave_sum(s1) + 2.0f * wave_sum(s2);
    if (lane == 0) { stats[0] = mx; stats[1] = sq; }
}

// payload -> encode -> modulate; writes the unscaled samples of the first n_data_sym symbols to
// audio[frame] and (max |x|, sum x^2) over everything the modulator emitted for the frame to fstats.
template <int LOG2N>
__global__ __launch_bounds__(kWave, 2) void stimulus_kernel(
    const DemodConst* __restrict__ Dp, const LdpcPlan* __restrict__ Pp, const c32* __restrict__ twiddle,
    const c32* __restrict__ nco_tx, unsigned long long seed, unsigned long long f0, int n_frames, int nraw,
    int ncw, int payload_bytes, int n_tx_symbols, float* __restrict__ audio, size_t frame_stride,
    unsigned char* __restrict__ payload_out, float* __restrict__ fstats) {
    using S = AcqShared<LOG2N>;
    constexpr int N = S::N, A = S::A;
    __shared__ StimShared<LOG2N> sh;
    const DemodConst& D = *Dp;
    const LdpcPlan& L = *Pp;
    const int lane = threadIdx.x;
    AcqLaneTw<LOG2N> ltw;
    stim_load_tables<LOG2N>(sh.fft, ltw, twiddle);
    const int k = L.k, m = L.m;
    const int total_bits = ncw * kLdpcN;                      // ncw = codewords the encoder makes of the nraw bytes
    const int cps = D.n_data, bpc = D.bits;
    for (int frame = blockIdx.x; frame < n_frames; frame += gridDim.x) {
        const unsigned long long f = f0 + (unsigned long long)frame;
        // ---- payload bytes: the oracle's counter-based stream (uo_make_batch) ----
        const unsigned long long s0 = (seed ^ f) * 0xD1342543DE82EF95ull + 0x5EEDull;
        for (int b = lane; b < nraw; b += kWave) {
            const unsigned char byte = (unsigned char)(splitmix_at(s0, (unsigned long long)b) >> 56);
            sh.raw[b] = byte;
            if (b < payload_bytes) payload_out[(size_t)frame * payload_bytes + b] = byte;
        }
        wave_sync();
        // ---- LDPCEncoder::encode: codeword c takes bits [c*k, c*k + k) of the byte stream (zeros beyond),
        //      parity i = xor of the row's information bits (H = [H_data | I]) ----
        for (int c = 0; c < ncw; ++c) {
            for (int j = lane; j < k; j += kWave) {
                const int b = c * k + j;
                sh.bits[c * kLdpcN + j] = (b < nraw * 8) ? (unsigned char)((sh.raw[b >> 3] >> (7 - (b & 7))) & 1) : 0;
            }
        }
        wave_sync();
        for (int c = 0; c < ncw; ++c) {
            for (int i = lane; i < L.row_rounds * 64; i += kWave) {          // row slots (may have gaps)
                if (L.row_deg[i] == 0) continue;
                unsigned char s = 0;
#pragma unroll
                for (int t = 0; t < 6; ++t) { const unsigned col = L.row_col[i * 6 + t]; if (col != 0xFFFFu) s ^= sh.bits[c * kLdpcN + col]; }
                sh.bits[c * kLdpcN + k + L.row_id[i]] = s;                   // plan rows are slots: parity bit of check row_id[i]
            }
        }
        wave_sync();
        // ---- OFDMModulator::modulate ----
        c32 dprev = mk(1.0f, 0.0f);                           // differential state of data carrier `lane`
        float mx = 0.0f, sq = 0.0f;
        float* frame_out = audio + (size_t)frame * frame_stride;
        for (int s = 0; s < n_tx_symbols; ++s) {
            for (int i = lane; i < N; i += kWave) sh.fft.X[i + (i >> A)] = mk(0.0f, 0.0f);
            wave_sync();
            if (lane < cps) {
                const int first = (s * cps + lane) * bpc;      // first bit of this carrier
                c32 sym = mk(0.0f, 0.0f);
                if (first < total_bits) {                     // else: data exhausted, carrier stays empty
                    unsigned bits = 0;
                    for (int b = 0; b < bpc; ++b) {
                        const int bi = first + b;
                        bits = (bits << 1) | ((bi < total_bits) ? (unsigned)sh.bits[bi] : 0u);
                    }
                    if (D.modulation == ULTRA_MOD_DBPSK) {
                        sym = cmul(dprev, (bits & 1u) ? mk(-1.0f, 0.0f) : mk(1.0f, 0.0f)); dprev = sym;
                    } else if (D.modulation == ULTRA_MOD_DQPSK) {
                        const unsigned q = bits & 3u;
                        const c32 ph = (q == 0) ? mk(1.0f, 0.0f) : (q == 1) ? mk(0.0f, 1.0f) : (q == 2) ? mk(-1.0f, 0.0f) : mk(0.0f, -1.0f);
                        sym = cmul(dprev, ph); dprev = sym;
                    } else if (D.modulation == ULTRA_MOD_D8PSK) {
                        const float pi = 3.14159265358979f;
                        const float angle = (float)(bits & 7u) * (pi / 4.0f) + pi / 8.0f;
                        sym = cmul(dprev, mk(um::cosf_(angle), um::sinf_(angle))); dprev = sym;
                    } else {
                        sym = stim_map_bits(bits, D.modulation);
                    }
                }
                const int bin = D.bin[D.data_slot[lane]];
                sh.fft.X[bin + (bin >> A)] = sym;
            }
            if (lane < D.n_pilot) { const int bin = D.bin[D.pilot_slot[lane]]; sh.fft.X[bin + (bin >> A)] = D.pilot_seq[lane]; }
            wave_sync();
            const bool keep = s < D.n_data_sym;
            float* out = keep ? frame_out + (size_t)s * D.sym_len : nullptr;
            stim_emit<LOG2N>(sh.fft, ltw, D.cp, nco_tx, 2 * (N + D.cp) + s * D.sym_len, out, mx, sq);
            if (keep) for (int g = N + D.cp + lane; g < D.sym_len; g += kWave) out[g] = 0.0f;     // symbol guard
        }
        mx = wave_max(mx);
        sq = wave_sum(sq);
        if (lane == 0) { fstats[2 * (size_t)frame] = mx; fstats[2 * (size_t)frame + 1] = sq; }
        wave_sync();
    }
}

// ---- channels --------------------------------------------------------------------------------
// counter-based standard normal pair for (key, n): Box-Muller on two 24-bit uniforms
__device__ __forceinline__ void gauss_pair(unsigned long long key, unsigned long long n, float* g0, float* g1) {
    const unsigned long long z = splitmix_at(key, n);
    const float u1 = ((float)((z >> 40) & 0xFFFFFFu) + 1.0f) * (1.0f / 16777216.0f);     // (0, 1]
    const float u2 = (float)((z >> 8) & 0xFFFFFFu) * (1.0f / 16777216.0f);               // [0, 1)
    const float rad = sqrtf(-2.0f * __logf(u1));
    float sn, cs;
    __sincosf(6.283185307179586f * u2, &sn, &cs);
    *g0 = rad * cs;
    *g1 = rad * sn;
}

// kind 0: scale only.  kind 1: AWGN at snr_db relative to the mean power of the whole signal
// (tools/test_nvis_mode.cpp:78-86).  kind 2: WattersonChannel::process (src/sim/hf_channel.hpp:
// 106-168,258-275): out = x*g1*|f1| + x_delayed*g2*|f2| + noise, f_k one-pole filtered complex
// Gaussians restarted at (1,0), noise from the RMS of the whole input.  The signal the channel sees
// starts at the preamble; only the frame part is written.
__global__ __launch_bounds__(kWave) void channel_kernel(
    const DemodConst* __restrict__ Dp, int kind, float snr_db, int delay_samples, float fading_alpha, float g1, float g2,
    unsigned long long seed, unsigned long long f0, int n_frames, int pre_len, int total_len,
    const float* __restrict__ preamble, const float* __restrict__ pre_stats, const float* __restrict__ fstats,
    float* __restrict__ audio, size_t frame_stride) {
    extern __shared__ float s_in[];                           // the frame's scaled input (Watterson: the delayed tap reads it)
    const DemodConst& D = *Dp;
    const int lane = threadIdx.x;
    const int frame_len = D.frame_samples;
    for (int frame = blockIdx.x; frame < n_frames; frame += gridDim.x) {
        const unsigned long long f = f0 + (unsigned long long)frame;
        const unsigned long long key = (seed ^ (f * 0x100000001B3ull)) * 0x9E3779B97F4A7C15ull + 0xC4A77E1ull;
        float* a = audio + (size_t)frame * frame_stride;
        const float mx = fmaxf(pre_stats[0], fstats[2 * (size_t)frame]);
        const float scale = 0.5f / mx;                        // sig[i] *= 0.5f / max_val
        if (kind == 0) {
            for (int i = lane; i < frame_len; i += kWave) a[i] *= scale;
            continue;
        }
        const float power = (pre_stats[1] + fstats[2 * (size_t)frame + 1]) * scale * scale / (float)total_len;
        if (kind == 1) {
            const float nstd = sqrtf(power / powf(10.0f, snr_db / 10.0f));
            for (int i = lane; i < frame_len; i += 2 * kWave) {
                float n0, n1;
                gauss_pair(key, (unsigned long long)(pre_len + i), &n0, &n1);
                a[i] = a[i] * scale + nstd * n0;
                if (i + kWave < frame_len) a[i + kWave] = a[i + kWave] * scale + nstd * n1;
            }
            continue;
        }
        for (int i = lane; i < frame_len; i += kWave) s_in[i] = a[i] * scale;
        wave_sync();
        // Watterson.  Fading filters f[n] = (1-al) f[n-1] + al * ns * g[n], ns = sqrt(1/al), over samples
        // 0 .. pre_len + frame_len - 1, chunk-parallel: lane l owns samples [l*C, (l+1)*C).
        const int T = pre_len + frame_len;
        const int C = (T + kWave - 1) / kWave;
        const float al = fading_alpha, dec = 1.0f - al, gain = al * sqrtf(1.0f / al);
        const int lo = lane * C, hi = (lo + C < T) ? lo + C : T;
        c32 p1 = mk(0.0f, 0.0f), p2 = mk(0.0f, 0.0f);        // chunk response from a zero state
        float decay = 1.0f;
        for (int i = lo; i < hi; ++i) {
            float a0, a1, b0, b1;
            gauss_pair(key ^ 0x1111ull, (unsigned long long)i, &a0, &a1);
            gauss_pair(key ^ 0x2222ull, (unsigned long long)i, &b0, &b1);
            p1 = mk(dec * p1.re + gain * a0, dec * p1.im + gain * a1);
            p2 = mk(dec * p2.re + gain * b0, dec * p2.im + gain * b1);
            decay *= dec;
        }
        // state at the start of each chunk: s_0 = (1,0), s_{l+1} = decay_l * s_l + p_l
        c32 s1 = mk(1.0f, 0.0f), s2 = mk(1.0f, 0.0f);
        c32 my1 = s1, my2 = s2;
        for (int l = 0; l < kWave; ++l) {
            if (lane == l) { my1 = s1; my2 = s2; }
            const float dl = lane_f(decay, l);
            const c32 q1 = mk(lane_f(p1.re, l), lane_f(p1.im, l)), q2 = mk(lane_f(p2.re, l), lane_f(p2.im, l));
            s1 = mk(dl * s1.re + q1.re, dl * s1.im + q1.im);
            s2 = mk(dl * s2.re + q2.re, dl * s2.im + q2.im);
        }
        const float eff_noise = sqrtf(power) * powf(10.0f, -snr_db / 20.0f);
        c32 c1 = my1, c2 = my2;
        for (int i = lo; i < hi; ++i) {
            float a0, a1, b0, b1;
            gauss_pair(key ^ 0x1111ull, (unsigned long long)i, &a0, &a1);
            gauss_pair(key ^ 0x2222ull, (unsigned long long)i, &b0, &b1);
            c1 = mk(dec * c1.re + gain * a0, dec * c1.im + gain * a1);
            c2 = mk(dec * c2.re + gain * b0, dec * c2.im + gain * b1);
            if (i < pre_len) continue;
            const int j = i - pre_len;
            auto sig = [&](int n) -> float {                   // scaled input sample n of the whole signal
                if (n < 0) return 0.0f;
                return (n < pre_len) ? preamble[n] * scale : s_in[n - pre_len];
            };
            const float h1 = sqrtf(c1.re * c1.re + c1.im * c1.im), h2 = sqrtf(c2.re * c2.re + c2.im * c2.im);
            // multipath: the reference's delay line holds delay_samples + 1 entries, so the second tap is
            // delay_samples + 1 samples late; without a delay there is a single faded path and no tap gains
            float o = (delay_samples > 0) ? sig(i) * g1 * h1 + sig(i - delay_samples - 1) * g2 * h2 : sig(i) * h1;
            float n0, n1;
            gauss_pair(key ^ 0x3333ull, (unsigned long long)i, &n0, &n1);
            o += eff_noise * n0;
            a[j] = o;
        }
        wave_sync();
    }
}

// Raw-audio streams for the end-to-end receive path (ultra_hip_receive_batch): [lead samples of noise][preamble]
// [the frame's data symbols, written unscaled by stimulus_kernel at offset lead + pre_len][tail samples of noise].
// The whole signal is scaled to a 0.5 peak and AWGN at snr_db relative to the mean power of preamble + modulator
// output is added to EVERY sample of the stream (kind 1; kind 0: scaling only) — the harness of
// tools/test_nvis_mode.cpp:62-86 with silence around the transmission.
// kind 2: the Watterson channel of channel_kernel over the transmission (same keys, same chunk-parallel fading filters: the
// frame of stream f fades exactly like frame f of ultra_hip_make_batch), its noise on every sample of the stream.
__global__ __launch_bounds__(kWave) void raw_stream_kernel(
    const DemodConst* __restrict__ Dp, int kind, float snr_db, int delay_samples, float fading_alpha, float g1, float g2,
    unsigned long long seed, unsigned long long f0, int n_frames,
    int lead, int pre_len, int tail, int total_len, const float* __restrict__ preamble, const float* __restrict__ pre_stats,
    const float* __restrict__ fstats, float* __restrict__ audio, size_t stream_stride) {
    extern __shared__ float s_tx[];                           // kind 2: the scaled transmission (the delayed tap reads it)
    const DemodConst& D = *Dp;
    const int lane = threadIdx.x;
    const int frame_len = D.frame_samples;
    const int n_out = lead + pre_len + frame_len + tail;
    for (int frame = blockIdx.x; frame < n_frames; frame += gridDim.x) {
        const unsigned long long f = f0 + (unsigned long long)frame;
        const unsigned long long key = (seed ^ (f * 0x100000001B3ull)) * 0x9E3779B97F4A7C15ull + 0xC4A77E1ull;
        float* a = audio + (size_t)frame * stream_stride;
        const float mx = fmaxf(pre_stats[0], fstats[2 * (size_t)frame]);
        const float scale = 0.5f / mx;
        const float power = (pre_stats[1] + fstats[2 * (size_t)frame + 1]) * scale * scale / (float)total_len;
        if (kind == 2) {
            const int T = pre_len + frame_len;
            for (int i = lane; i < T; i += kWave) s_tx[i] = ((i < pre_len) ? preamble[i] : a[lead + i]) * scale;
            wave_sync();
            const int C = (T + kWave - 1) / kWave;
            const float al = fading_alpha, dec = 1.0f - al, gain = al * sqrtf(1.0f / al);
            const int lo = lane * C, hi = (lo + C < T) ? lo + C : T;
            c32 p1 = mk(0.0f, 0.0f), p2 = mk(0.0f, 0.0f);
            float decay = 1.0f;
            for (int i = lo; i < hi; ++i) {
                float a0, a1, b0, b1;
                gauss_pair(key ^ 0x1111ull, (unsigned long long)i, &a0, &a1);
                gauss_pair(key ^ 0x2222ull, (unsigned long long)i, &b0, &b1);
                p1 = mk(dec * p1.re + gain * a0, dec * p1.im + gain * a1);
                p2 = mk(dec * p2.re + gain * b0, dec * p2.im + gain * b1);
                decay *= dec;
            }
            c32 s1 = mk(1.0f, 0.0f), s2 = mk(1.0f, 0.0f);
            c32 my1 = s1, my2 = s2;
            for (int l = 0; l < kWave; ++l) {
                if (lane == l) { my1 = s1; my2 = s2; }
                const float dl = lane_f(decay, l);
                const c32 q1 = mk(lane_f(p1.re, l), lane_f(p1.im, l)), q2 = mk(lane_f(p2.re, l), lane_f(p2.im, l));
                s1 = mk(dl * s1.re + q1.re, dl * s1.im + q1.im);
                s2 = mk(dl * s2.re + q2.re, dl * s2.im + q2.im);
            }
            const float eff_noise = sqrtf(power) * powf(10.0f, -snr_db / 20.0f);
            c32 c1 = my1, c2 = my2;
            for (int i = lo; i < hi; ++i) {
                float a0, a1, b0, b1;
                gauss_pair(key ^ 0x1111ull, (unsigned long long)i, &a0, &a1);
                gauss_pair(key ^ 0x2222ull, (unsigned long long)i, &b0, &b1);
                c1 = mk(dec * c1.re + gain * a0, dec * c1.im + gain * a1);
                c2 = mk(dec * c2.re + gain * b0, dec * c2.im + gain * b1);
                const float h1 = sqrtf(c1.re * c1.re + c1.im * c1.im), h2 = sqrtf(c2.re * c2.re + c2.im * c2.im);
                const int d = i - delay_samples - 1;
                float o = (delay_samples > 0) ? s_tx[i] * g1 * h1 + ((d >= 0) ? s_tx[d] : 0.0f) * g2 * h2 : s_tx[i] * h1;
                float n0, n1;
                gauss_pair(key ^ 0x3333ull, (unsigned long long)i, &n0, &n1);
                a[lead + i] = o + eff_noise * n0;
            }
            // the silence around the transmission carries the channel's noise alone
            for (int i = lane; i < lead + tail; i += kWave) {
                float n0, n1;
                gauss_pair(key ^ 0x4444ull, (unsigned long long)i, &n0, &n1);
                a[(i < lead) ? i : T + i] = eff_noise * n0;
            }
            wave_sync();
            continue;
        }
        const float nstd = (kind == 1) ? sqrtf(power / powf(10.0f, snr_db / 10.0f)) : 0.0f;
        for (int i = lane; i < n_out; i += 2 * kWave) {
            float n0 = 0.0f, n1 = 0.0f;
            if (kind == 1) gauss_pair(key, (unsigned long long)i, &n0, &n1);
            auto sig = [&](int j) -> float {
                if (j < lead || j >= lead + pre_len + frame_len) return 0.0f;
                return ((j < lead + pre_len) ? preamble[j - lead] : a[j]) * scale;
            };
            a[i] = sig(i) + nstd * n0;
            if (i + kWave < n_out) a[i + kWave] = sig(i + kWave) + nstd * n1;
        }
    }
}

// ---- BPSK-over-AWGN LLR stimulus (BASELINE configs[3]: the LDPC-only SNR sweep) -----------------
// Counter-based standard normal pair for (key, n) with libm-exact arithmetic: Box-Muller on two 24-bit uniforms,
// logf / sincosf from pinned_math.h (bit-identical to glibc 2.35's), correctly rounded sqrtf, every other step one
// IEEE float operation.  The test oracle's twin (uo_make_llr_batch) calls libm and reproduces every LLR bit for bit,
// so any codeword of any sweep point can be regenerated and decoded on the host.
__device__ __forceinline__ void gauss_pair_exact(unsigned long long key, unsigned long long n, float* g0, float* g1) {
    const unsigned long long z = splitmix_at(key, n);
    const float u1 = ((float)((z >> 40) & 0xFFFFFFull) + 1.0f) * (1.0f / 16777216.0f);     // (0, 1]
    const float u2 = (float)((z >> 8) & 0xFFFFFFull) * (1.0f / 16777216.0f);               // [0, 1)
    const float rad = sqrtf(-2.0f * um::logf_(u1));
    float sn, cs;
    um::sincosf_(6.283185307179586f * u2, &sn, &cs);
    *g0 = rad * cs;
    *g1 = rad * sn;
}

// One wavefront per codeword c = c0 + i: payload bytes (the stream of stimulus_kernel / uo_make_batch for frame c)
// -> LDPCEncoder::encode (src/fec/ldpc_encoder.cpp:193-257: k information bits, zeros behind the payload, parity
// i = xor of the row's information bits) -> BPSK x = 1 - 2 bit (LLR > 0 <=> bit 0, docs/INVARIANTS.md:213-222)
// -> y = x + sigma * n -> LLR = 2 y / sigma^2.  sigma, sigma2: floats computed by the host from Es/N0.
__global__ __launch_bounds__(kWave) void llr_stimulus_kernel(const LdpcPlan* __restrict__ Pp, unsigned long long seed,
                                                             unsigned long long c0, int n_cw, int payload_bytes, float sigma,
                                                             float sigma2, float* __restrict__ llr,
                                                             unsigned char* __restrict__ payload_out) {
    __shared__ unsigned char bits[kLdpcN + 8];
    const LdpcPlan& L = *Pp;
    const int lane = threadIdx.x;
    const int k = L.k;
    for (int cw = blockIdx.x; cw < n_cw; cw += gridDim.x) {
        const unsigned long long c = c0 + (unsigned long long)cw;
        const unsigned long long s0 = (seed ^ c) * 0xD1342543DE82EF95ull + 0x5EEDull;
        for (int j = lane; j < k; j += kWave) bits[j] = 0;
        wave_sync();
        for (int b = lane; b < payload_bytes; b += kWave) {
            const unsigned char byte = (unsigned char)(splitmix_at(s0, (unsigned long long)b) >> 56);
            payload_out[(size_t)cw * payload_bytes + b] = byte;
#pragma unroll
            for (int t = 0; t < 8; ++t) bits[8 * b + t] = (unsigned char)((byte >> (7 - t)) & 1);
        }
        wave_sync();
        for (int i = lane; i < L.row_rounds * 64; i += kWave) {              // row slots (may have gaps)
            if (L.row_deg[i] == 0) continue;
            unsigned char s = 0;
#pragma unroll
            for (int t = 0; t < 6; ++t) { const unsigned col = L.row_col[i * 6 + t]; if (col != 0xFFFFu) s ^= bits[col]; }
            bits[k + L.row_id[i]] = s;
        }
        wave_sync();
        const unsigned long long key = ((seed ^ (c * 0x100000001B3ull)) * 0x9E3779B97F4A7C15ull + 0xC4A77E1ull) ^ 0x4444ull;
        float* out = llr + (size_t)cw * kLdpcN;
        for (int p = lane; p < kLdpcN / 2; p += kWave) {                     // noise pair p -> bits 2p, 2p + 1
            float g0, g1;
            gauss_pair_exact(key, (unsigned long long)p, &g0, &g1);
            const float x0 = bits[2 * p] ? -1.0f : 1.0f, x1 = bits[2 * p + 1] ? -1.0f : 1.0f;
            const float y0 = x0 + sigma * g0, y1 = x1 + sigma * g1;
            out[2 * p] = (2.0f * y0) / sigma2;
            out[2 * p + 1] = (2.0f * y1) / sigma2;
        }
        wave_sync();
    }
}

// ---- WattersonChannel::applyCFO (src/sim/hf_channel.hpp:161-232): the radio's tuning error as the harnesses model it ----
// Mix the passband signal down from 1500 Hz, 48-tap running-mean lowpass, rotate at baseband, mix back up.  Every
// frame sees a freshly constructed channel (phase 0, sample index from 0), so the two oscillator sequences — the
// 1500 Hz mixer and the CFO rotator with its serial float phase recurrence — are the same for every frame of a call:
// cfo_tables_kernel writes them once (the rotator by one lane, in the reference's order), cfo_shift_kernel then walks
// one frame per lane with the running sums in the reference's order.  Same operations on the same operands as the
// reference, libm through pinned_math.h: bit-identical to the oracle's uo_channel_apply_cfo, which is pinned to the
// compiled reference.
__global__ __launch_bounds__(256) void cfo_tables_kernel(float phase_inc, float fs, double two_pi_fc, int n,
                                                         c32* __restrict__ mixer, c32* __restrict__ rotator) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const float t = (float)i / fs;
        const float mix_phase = (float)(two_pi_fc * (double)t);
        float sn, cs;
        um::sincosf_(mix_phase, &sn, &cs);
        mixer[i] = mk(cs, sn);
    }
    if (i == 0) {
        float phase = 0.0f;
        for (int j = 0; j < n; ++j) {
            float sn, cs;
            um::sincosf_(phase, &sn, &cs);
            rotator[j] = mk(cs, sn);
            phase += phase_inc;
            if ((double)phase > kTwoPi) phase = (float)((double)phase - kTwoPi);
        }
    }
}

__global__ __launch_bounds__(64) void cfo_shift_kernel(const float* __restrict__ in, size_t in_stride, float* __restrict__ out,
                                                       size_t out_stride, int n, int n_frames,
                                                       const c32* __restrict__ mixer, const c32* __restrict__ rotator) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n_frames) return;
    const float* x = in + (size_t)f * in_stride;
    float* y = out + (size_t)f * out_stride;
    constexpr int kWin = 48;
    float i_sum = 0.0f, q_sum = 0.0f;
    for (int i = 0; i < n; ++i) {
        const c32 m = mixer[i];
        const float xi = x[i];
        i_sum += xi * m.re;
        q_sum += xi * m.im;
        if (i >= kWin) {
            const c32 mo = mixer[i - kWin];
            const float xo = x[i - kWin];
            i_sum -= xo * mo.re;
            q_sum -= xo * mo.im;
        }
        const float cnt = (float)((i + 1 < kWin) ? i + 1 : kWin);
        const float i_filt = i_sum / cnt, q_filt = q_sum / cnt;
        const c32 r = rotator[i];
        const float i_cfo = i_filt * r.re - q_filt * r.im;
        const float q_cfo = i_filt * r.im + q_filt * r.re;
        y[i] = 2.0f * (i_cfo * m.re - q_cfo * m.im);
    }
}

}  // namespace dev
}  // namespace ultra_hip
#endif
