// acquire_kernel.h — batched preamble acquisition for gfx950: one wavefront per audio stream.
//
// Scope row f1: the SEARCHING state of OFDMDemodulator::process
// (src/ofdm/demodulator.cpp:461-600) for a batch of independent streams, each received by a fresh
// demodulator that is fed `chunk` samples per call.  Restated functions:
//   Impl::hasMinimumEnergy               src/ofdm/ofdm_sync.cpp:20-50     (stateful noise floor)
//   Impl::toAnalytic                     :56-84   FFT -> zero negative bins -> inverse FFT
//   Impl::measureSchmidlCoxCorrelation   :120-163 half-symbol autocorrelation of the analytic signal
//   Impl::estimateCoarseCFO              :230-261
//   Impl::refineLTSTiming                :386-461 passband matched filter over 3.5 preamble symbols
//   the search / plateau / trim logic of process() itself
// The search is sequential by construction (it restarts from offset 0 on every call, the energy gate
// carries state from candidate to candidate, the first plateau wins), so a stream is one wavefront
// walking the reference's control flow with wave-uniform decisions; the 64 lanes share the work
// inside each step: the 1024-point FFT pair of the analytic signal (radix-2 DIT exactly as
// FFT::fft_impl, src/dsp/fft.cpp:89-121, in LDS), the per-index terms of the correlation sums, 64
// matched-filter offsets at a time.  Every float sum the reference accumulates serially is
// accumulated in the same order (independent chains run in different lanes).
//
// Outputs per stream: found, data_start (absolute sample index: what process() consumes the buffer
// up to, demodulator.cpp:572-575), the coarse CFO, the Schmidl-Cox offset and the number of samples
// fed when sync was declared.  (data_start, coarse CFO) is exactly the SYNCED entry of the
// demodulator kernels.
#ifndef ULTRA_ACQUIRE_KERNEL_H
#define ULTRA_ACQUIRE_KERNEL_H

#include <hip/hip_runtime.h>
#include "device_types.h"
#include "demod_kernel.h"

namespace ultra_hip {
namespace dev {

// demodulator_constants.hpp:41-53
constexpr unsigned kAcqMinSearch = 4000u, kAcqMaxBuffer = 240000u, kAcqOverlap = 20000u, kAcqStep = 8u,
                   kAcqPlateauWindow = 300u, kAcqMinPlateau = 15u;
constexpr int kAcqCache = 256;
// Diagnostic build only (-DUH_ACQ_STAMPS, tools/acquire_stalls.py): shader-clock time of one stream's search split over its
// phases — record of kAcqStampWords 64-bit words per stream: [0] whole search, [1] energy-gate groups, [2] DC groups,
// [3] window metrics of the search and the plateau scans (analytic-signal transform pair + half-symbol sums), [4] the CFO
// metric, [5] LTS matched filter, [6] metrics evaluated, [7] gate groups, [8] DC groups, [9] candidates visited (turns of the
// state machine), [10] process() calls, [11] metric-cache hits.  The product build contains none of it.
constexpr int kAcqStampWords = 12;
#ifdef UH_ACQ_STAMPS
__device__ unsigned long long* g_acq_stamps = nullptr;
#define UH_AQ_T0() const unsigned long long aq_t_ = __builtin_readcyclecounter()
#define UH_AQ_ADD(k) aq_acc[k] += __builtin_readcyclecounter() - aq_t_
#define UH_AQ_CNT(k) ++aq_acc[k]
#else
#define UH_AQ_T0() do {} while (0)
#define UH_AQ_ADD(k) do {} while (0)
#define UH_AQ_CNT(k) do {} while (0)
#endif
// LIVE streams (ultra_hip_acquire_stream_batch, one process() call per launch): the metric cache of a stream lives in HBM
// between the launches, direct-mapped by absolute window start / 8 like the LDS cache of the batch kernel but covering
// 262,144 samples — more than rx_buffer can ever hold (MAX_BUFFER_SAMPLES = 240,000; between two trims of a chunk-fed stream
// it holds 2 x OVERLAP_SAMPLES, and RxPipeline hands detectSync up to two seconds = 96,000 at once), so nothing the search can
// still reach is ever evicted.  Per stream: [tags kAcqGCache u32][values kAcqGCache f32][header kAcqGHeader u32: samples fed after the
// last launch].  The reference re-evaluates every candidate of its buffer on every call (the search restarts at offset 0:
// demodulator.cpp:497); here a call evaluates only the candidates whose window the new chunk completed — in parallel, one
// wavefront each (acq_prepass_kernel) — and the sequential walk reads everything else back, 64 entries at a time.
constexpr int kAcqGCache = 32768, kAcqGHeader = 16, kAcqGWords = 2 * kAcqGCache + kAcqGHeader;

template <int LOG2N>
struct AcqShared {
    static constexpr int N = 1 << LOG2N;
    static constexpr int P = N / kWave;                     // points per lane: 8 / 16
    static constexpr int A = (P == 16) ? 4 : 3;             // log2(P)
    static constexpr int kTwB = P * ((1 << A) - 1);
    static constexpr int kTwA = (1 << A) - 1;               // wave-uniform twiddles of stages 0..A-1, one run per stage
    union {
        c32 X[N + N / P];             // FFT exchange buffer, 1 pad per P entries
        float terms[N / 2][4];        // per-index terms of the four correlation sums (analytic signal in registers)
        float lts_win[2 * (N + N / P)];   // audio window of one matched-filter pass (same bytes as X)
        float grp[(N + 8 * kWave) / 8 * 9];   // samples of 64 candidate windows 8 apart, 1 pad per 8 (acq_group_dc)
        float gsq[kWave + 2 * ((2 * (N + N / 2) + 15) / 16)];   // (cp <= N / 2) squares of every 8th sample under 64 gate windows (acq_group_energy)
    };
    c32 twB[kTwB];                    // twiddles of stages A..2A-1, one contiguous run per stage (as in mix_fft_kernel)
    c32 twA[kTwA + 1];
    // Metric cache of the stream (direct-mapped by absolute window offset / 8): the search restarts at
    // offset 0 on every process() call, so most candidates of a call were already evaluated by the call before.
    unsigned ctag[kAcqCache];
    float cval[kAcqCache];
};

// per-lane twiddles of stages 2A..LOG2N-1 (k = lane + 64*(t & (ht-1))): read from the L2-resident table at
// every use, as mix_fft_kernel does (kept in registers they cost 24 VGPRs for the whole kernel, and the kernel
// spilled: 692 B of scratch per lane, reloaded inside the FFTs)
template <int LOG2N>
struct AcqLaneTw {
    const c32* __restrict__ table;     // twiddle[k], k < N/2
    __device__ __forceinline__ c32 at(int s, int c) const { return table[((int)threadIdx.x + 64 * c) << (LOG2N - 1 - s)]; }
};

// DC sums of 64 candidate windows in ONE chain pass.  measureSchmidlCoxCorrelation (ofdm_sync.cpp:131-140)
// sums the N samples of its window in order before anything else; the candidates the search and the
// plateau scan visit lie 8 samples apart, so lane g walks the window that starts 8 g samples behind
// `first` — 64 chains side by side for the price (N dependent adds) the single broadcast walk paid per
// candidate.  The N + 504 samples sit in LDS with one pad word per 8 samples: sample 8 g + n lies at word
// 9 g + n + n / 8, so the lanes of a step read words 9 apart (all 32 banks), and a run of 16 terms is 17
// consecutive words (immediate offsets).  Windows may run past the samples fed so far (the stream is
// all in memory; a candidate is only ever USED once its window is inside the fed part, and its value does
// not depend on when it was summed); past the end of the stream they read zeros and are never used.
template <int LOG2N>
__device__ __forceinline__ void acq_gload16(const float* q, float (&r)[16]) {
#pragma unroll
    for (int u = 0; u < 16; ++u) r[u] = q[u + (u >> 3)];
    asm volatile("" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                      "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15]));
}
template <int LOG2N>
__device__ __forceinline__ float acq_group_dc(AcqShared<LOG2N>& sh, const float* __restrict__ all, unsigned first,
                                              unsigned n_samples) {
    constexpr int N = 1 << LOG2N;
    const int lane = threadIdx.x;
    for (unsigned s0 = 0; s0 < (unsigned)(N + 8 * kWave); s0 += kWave) {
        const unsigned sidx = s0 + (unsigned)lane, g = first + sidx;
        sh.grp[sidx + (sidx >> 3)] = (g < n_samples) ? all[g] : 0.0f;
    }
    wave_sync();
    const float* q = sh.grp + 9 * lane;
    float s = 0.0f, ra[16], rb[16];
    acq_gload16<LOG2N>(q, ra);
    for (int i = 0; i < N; i += 32) {
        acq_gload16<LOG2N>(q + 18, rb);
#pragma unroll
        for (int u = 0; u < 16; ++u) s += ra[u];
        acq_gload16<LOG2N>((i + 32 < N) ? q + 36 : q, ra);    // the last round re-reads a block it does not use
#pragma unroll
        for (int u = 0; u < 16; ++u) s += rb[u];
        q += 36;
    }
    wave_sync();                                             // grp aliases the FFT exchange buffer
    return s;
}

// Radix-2 DIT exactly as FFT::fft_impl (src/dsp/fft.cpp:89-121), register-resident like the FFT of
// mix_fft_kernel: in: v[q] = element P*lane + q of the bit-reversed input; three groups of stages on
// P register-resident points with two LDS transposes in between; out: v[t] = X[lane + 64*t].
// INVERSE: conjugated twiddles and the 1/N scaling.
template <int LOG2N, bool INVERSE>
__device__ __forceinline__ void acq_fft(AcqShared<LOG2N>& sh, const AcqLaneTw<LOG2N>& ltw, c32 (&v)[AcqShared<LOG2N>::P]) {
    using S = AcqShared<LOG2N>;
    constexpr int N = S::N, P = S::P, A = S::A;
    const int lane = threadIdx.x;
    auto tw = [](c32 w) { return INVERSE ? cconj(w) : w; };
#pragma unroll
    for (int s = 0; s < A; ++s) {                            // stages 0..A-1, wave-uniform twiddles
        const int half = 1 << s;
#pragma unroll
        for (int q = 0; q < P; ++q) {
            if (q & half) continue;
            const int k = q & (half - 1);                    // compile-time (unrolled): twiddle[k << (LOG2N - 1 - s)]
            const c32 w = tw(sh.twA[(half - 1) + k]);
            // twiddle[0] = (1, -0) and twiddle[N/4] = (cos(-pi/2), -1) (host_tables.h checks both): two products instead of
            // four, the same values bit for bit — see the group A of symbol_to_freq2 (demod_kernel.h)
            if (k == 0) {
                const c32 b_ = v[q + half];
                const c32 t_ = mk(b_.re - w.im * b_.im, b_.im + w.im * b_.re);
                v[q + half] = csub(v[q], t_); v[q] = cadd(v[q], t_);
            } else if (2 * k == half) {
                const c32 b_ = v[q + half];
                const c32 t_ = INVERSE ? mk(w.re * b_.re - b_.im, w.re * b_.im + b_.re) : mk(w.re * b_.re + b_.im, w.re * b_.im - b_.re);
                v[q + half] = csub(v[q], t_); v[q] = cadd(v[q], t_);
            } else {
                UH_BUTTERFLY(v[q], v[q + half], w);
            }
        }
    }
#pragma unroll
    for (int q = 0; q < P; ++q) { const int i = P * lane + q; sh.X[i + (i >> A)] = v[q]; }
    wave_sync();
    {
        const int blk = lane / P, r = lane % P;              // stages A..2A-1 on X[blk*P*P + r + P*j]
#pragma unroll
        for (int j = 0; j < P; ++j) { const int i = blk * P * P + r + P * j; v[j] = sh.X[i + (i >> A)]; }
#pragma unroll
        for (int s = A; s < 2 * A; ++s) {
            const int hj = 1 << (s - A);
#pragma unroll
            for (int j = 0; j < P; ++j) {
                if (j & hj) continue;
                const int k = r + P * (j & (hj - 1));
                const c32 w = tw(sh.twB[P * (hj - 1) + k]);
                UH_BUTTERFLY(v[j], v[j + hj], w);
            }
        }
#pragma unroll
        for (int j = 0; j < P; ++j) { const int i = blk * P * P + r + P * j; sh.X[i + (i >> A)] = v[j]; }
    }
    wave_sync();
#pragma unroll
    for (int t = 0; t < P; ++t) { const int i = lane + 64 * t; v[t] = sh.X[i + (i >> A)]; }
#pragma unroll
    for (int s = 2 * A; s < LOG2N; ++s) {                    // stages 2A..LOG2N-1 on X[lane + 64*t]
        const int ht = 1 << (s - 6);
#pragma unroll
        for (int t = 0; t < P; ++t) {
            if (t & ht) continue;
            const c32 w = tw(ltw.at(s, t & (ht - 1)));
            UH_BUTTERFLY(v[t], v[t + ht], w);
        }
    }
    if (INVERSE) {
        const float scale = 1.0f / (float)N;
#pragma unroll
        for (int t = 0; t < P; ++t) v[t] = cscale(v[t], scale);
    }
    wave_sync();                                             // X is free again
}

// Impl::toAnalytic for len == fft_size.  in: xs[qp] = sample rl + 64*qp of the window (rl =
// bitrev6(lane)), dc subtracted here; out: v[t] = analytic[lane + 64*t].
template <int LOG2N>
__device__ __forceinline__ void acq_analytic(AcqShared<LOG2N>& sh, const AcqLaneTw<LOG2N>& ltw,
                                             const float (&xs)[AcqShared<LOG2N>::P], float dc,
                                             c32 (&v)[AcqShared<LOG2N>::P]) {
    using S = AcqShared<LOG2N>;
    constexpr int N = S::N, P = S::P, A = S::A;
    const int lane = threadIdx.x;
    const int rl = (int)(__brev((unsigned)lane) >> 26);
#pragma unroll
    for (int qp = 0; qp < P; ++qp) v[bitrev_small<A>(qp)] = mk(xs[qp] - dc, 0.0f);
    acq_fft<LOG2N, false>(sh, ltw, v);
    // freq[1..N/2) *= 2, freq(N/2..N) = 0; then the inverse transform's bit reversal: its input element
    // P*lane + q is freq[bitrev(P*lane + q)] = freq[64*bitrev_A(q) + rl], i.e. register bitrev_A(q) of lane rl
#pragma unroll
    for (int t = 0; t < P; ++t) {
        const int i = lane + 64 * t;
        c32 f = v[t];
        if (i >= 1 && i < N / 2) f = cscale(f, 2.0f);
        if (i > N / 2) f = mk(0.0f, 0.0f);
        sh.X[i + (i >> A)] = f;
    }
    wave_sync();
#pragma unroll
    for (int q = 0; q < P; ++q) { const int i = rl + 64 * bitrev_small<A>(q); v[q] = sh.X[i + (i >> A)]; }
    wave_sync();
    acq_fft<LOG2N, true>(sh, ltw, v);
}

// P = sum conj(a[i]) a[i+half], R1 = sum |a[i]|^2, R2 = sum |a[i+half]|^2 in index order; a[lane + 64 t]
// = v[t], so a[i] and a[i + N/2] sit in the same lane (registers t and t + P/2).
template <int LOG2N>
__device__ __forceinline__ void acq_half_sums(AcqShared<LOG2N>& sh, const c32 (&v)[AcqShared<LOG2N>::P], c32* Pout,
                                              float* R1, float* R2) {
    using S = AcqShared<LOG2N>;
    constexpr int N = S::N, P = S::P, H = N / 2;
    const int lane = threadIdx.x;
#pragma unroll
    for (int t = 0; t < P / 2; ++t) {
        const c32 x = v[t], y = v[t + P / 2];
        const c32 m = cmul(cconj(x), y);
        *reinterpret_cast<float4*>(&sh.terms[lane + 64 * t][0]) = make_float4(m.re, m.im, cnorm(x), cnorm(y));
    }
    wave_sync();
    const int col = lane & 3;                       // four chains, lane l follows chain l & 3 (broadcast reads)
    float acc = 0.0f;
    {
        float cur[8], nxt[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) cur[u] = sh.terms[u][col];
        for (int i = 0; i < H; i += 8) {
            const int j = (i + 8 < H) ? i + 8 : i;
#pragma unroll
            for (int u = 0; u < 8; ++u) nxt[u] = sh.terms[j + u][col];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += cur[u];
#pragma unroll
            for (int u = 0; u < 8; ++u) cur[u] = nxt[u];
        }
    }
    *Pout = mk(lane_f(acc, 0), lane_f(acc, 1));
    *R1 = lane_f(acc, 2);
    *R2 = lane_f(acc, 3);
    wave_sync();
}

// The half-symbol autocorrelation of the analytic signal of the N samples at `win`: P, R1, R2.
// dc_sum: the in-order sum of the window's samples (measureSchmidlCoxCorrelation subtracts the mean first,
// ofdm_sync.cpp:131-140; it comes from acq_group_dc), +0 for estimateCoarseCFO (:241), which takes the raw samples.  ONE instance of this code serves the search, the
// plateau scan and the CFO estimate (see the state machine in acquire_kernel): three inlined copies
// cost 239 VGPRs.
template <int LOG2N>
__device__ __forceinline__ void acq_window_metric(AcqShared<LOG2N>& sh, const AcqLaneTw<LOG2N>& ltw,
                                                  const float* __restrict__ win, float dc_sum, c32* P_out, float* R1,
                                                  float* R2) {
    constexpr int N = 1 << LOG2N;
    constexpr int P = AcqShared<LOG2N>::P;
    const int lane = threadIdx.x;
    const int rl = (int)(__brev((unsigned)lane) >> 26);
    float xs[P];
#pragma unroll
    for (int qp = 0; qp < P; ++qp) xs[qp] = win[rl + 64 * qp];
    const float dc = dc_sum / (float)N;                      // dc_sum = +0 when no dc is removed: x - 0.0f == x for every float
    c32 v[P];
    acq_analytic<LOG2N>(sh, ltw, xs, dc, v);
    acq_half_sums<LOG2N>(sh, v, P_out, R1, R2);
}

// Impl::hasMinimumEnergy's sum for 64 candidates at once.  The gate squares every 16th sample of its window
// and adds the squares in order; the candidates of the search lie 8 samples apart, so candidate g needs the
// squares of samples first + 8 (g + 2 t), t = 0 .. count-1: every square is formed ONCE (one multiplication
// per sample instead of one per sample and candidate), parked in LDS, and lane g walks g, g + 2, g + 4, ..
// — 64 in-order sums for count additions (consecutive lanes read consecutive words; the DPP chain spent 63
// dependent adds of ~17 cycles per 64 terms of ONE candidate).  Lane g returns candidate g's sum.
template <int LOG2N>
__device__ __forceinline__ float acq_group_energy(AcqShared<LOG2N>& sh, const float* __restrict__ all, unsigned first,
                                                  int count, unsigned n_samples) {
    const int lane = threadIdx.x;
    const int n_sq = kWave + 2 * count;
    for (int k = lane; k < n_sq; k += kWave) {
        const unsigned g = first + 8u * (unsigned)k;
        const float v = (g < n_samples) ? all[g] : 0.0f;          // windows past the end of the stream are never used
        sh.gsq[k] = v * v;
    }
    wave_sync();
    const float* q = sh.gsq + lane;
    float sum = 0.0f;
    int t = 0;
    for (; t + 4 <= count; t += 4) {
        const float a = q[2 * t], b = q[2 * t + 2], c = q[2 * t + 4], d = q[2 * t + 6];
        sum += a; sum += b; sum += c; sum += d;
    }
    for (; t < count; ++t) sum += q[2 * t];
    wave_sync();                                             // gsq aliases the FFT exchange buffer
    return sum;
}

// Impl::hasMinimumEnergy (the sum comes from acq_group_energy)
__device__ __forceinline__ bool acq_energy_gate(float sum_sq, int count, float& noise_floor) {
    const float energy = sum_sq / (float)count;
    if (noise_floor < 1e-20f) noise_floor = energy * 0.1f;
    if (energy < noise_floor) noise_floor = energy;
    else if (energy < noise_floor * 3.0f) noise_floor = (1.0f - 0.01f) * noise_floor + 0.01f * energy;
    const float threshold = noise_floor * 4.0f;
    return energy >= threshold;
}

// Impl::refineLTSTiming; returns 0xffffffff on failure.  Lane l evaluates offsets first + l + 64 r.
template <int LOG2N>
__device__ __forceinline__ unsigned acq_refine_lts(AcqShared<LOG2N>& sh, const DemodConst& D, const float* __restrict__ lts_I,
                                                   const float* __restrict__ lts_Q, float energy_ref,
                                                   const float* __restrict__ buf, unsigned size, unsigned sts_start) {
    constexpr int N = 1 << LOG2N;
    const int lane = threadIdx.x;
    const unsigned psl = (unsigned)(N + D.cp);
    const unsigned lts_len = psl;
    const unsigned coarse = sts_start + 4u * psl;
    const int back = (int)(3u * psl), fwd = (int)(psl / 2u);
    if (coarse < (unsigned)back || coarse + (unsigned)fwd + lts_len > size) return coarse;
    // The search window goes through LDS in passes of kR*64 offsets; lane l evaluates offsets
    // l, l + 64, .., l + 64*(kR-1) of a pass together, so one (wave-uniform, scalar-loaded) template
    // pair serves kR accumulations and its load latency disappears behind them (one offset per lane
    // per round spent 64 cycles per tap, most of them waiting for the template loads: 4.5 of the
    // 15 M cycles of a stream).  Each offset's three sums still run over the taps in order.
    constexpr int kR = 8;
    typedef float v2f __attribute__((ext_vector_type(2)));
    const unsigned win0 = coarse - (unsigned)back;
    const int n_off = back + fwd + 1;
    float best_corr = 0.0f;
    unsigned best_off = coarse;
    for (int p0 = 0; p0 < n_off; p0 += kR * 64) {
        const int n_here = (n_off - p0 < kR * 64) ? n_off - p0 : kR * 64;
        const unsigned need = (unsigned)(kR * 64 - 1) + lts_len;                     // <= sizeof(lts_win): checked on the host
        wave_sync();
        for (unsigned i = lane; i < need; i += kWave) {
            const unsigned g = win0 + (unsigned)p0 + i;
            sh.lts_win[i] = (g < size) ? buf[g] : 0.0f;                               // beyond the last offset of the last pass
        }
        wave_sync();
        v2f ciq[kR], er2[kR / 2];                                 // er2[r / 2] = (energy_rx of offset r, of offset r + 1)
#pragma unroll
        for (int r = 0; r < kR; ++r) ciq[r] = v2f{0.0f, 0.0f};
#pragma unroll
        for (int r = 0; r < kR / 2; ++r) er2[r] = v2f{0.0f, 0.0f};
        const float* p = sh.lts_win + lane;
        for (unsigned i = 0; i < lts_len; ++i) {
            const v2f t = {lts_I[i], lts_Q[i]};
#pragma unroll
            for (int r = 0; r < kR; r += 2) {
                // corr_I += rx * I[i]; corr_Q += rx * Q[i]; energy_rx += rx * rx — packed, never contracted.  The two
                // samples of a register pair (one ds_read2st64_b32) multiply the scalar template pair through operand
                // selection: (I, Q) * (lo, lo) and (I, Q) * (hi, hi) — the compiler copied `hi` into a fresh pair first
                // (4 of the 28 VALU instructions per tap).
                const v2f rx = {p[i + 64 * r], p[i + 64 * (r + 1)]};
                v2f m0, m1;
                asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(m0) : "s"(t), "v"(rx));
                asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(m1) : "s"(t), "v"(rx));
                ciq[r] = ciq[r] + m0;
                ciq[r + 1] = ciq[r + 1] + m1;
                er2[r / 2] = er2[r / 2] + rx * rx;
            }
        }
        float er[kR];
#pragma unroll
        for (int r = 0; r < kR; r += 2) { er[r] = er2[r / 2].x; er[r + 1] = er2[r / 2].y; }
#pragma unroll
        for (int r = 0; r < kR; ++r) {                        // increasing offsets within a lane
            const int idx = lane + 64 * r;
            const float corr_mag = sqrtf(ciq[r].x * ciq[r].x + ciq[r].y * ciq[r].y);
            const float norm = sqrtf(er[r] * energy_ref);
            const float corr = (norm > 1e-6f) ? corr_mag / norm : 0.0f;
            if (idx < n_here && corr > best_corr) { best_corr = corr; best_off = win0 + (unsigned)(p0 + idx); }
        }
    }
    wave_sync();
    // first occurrence of the maximum over all offsets = max corr, smallest offset among equals
    float m = best_corr;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    const unsigned cand = (best_corr == m) ? best_off : 0xffffffffu;
    unsigned w = cand;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const unsigned x = (unsigned)__shfl_xor((int)w, o, 64); w = (x < w) ? x : w; }
    const float thr = (N >= 1024) ? 0.05f : 0.35f;
    if (m < thr) return 0xffffffffu;
    return (m > 0.0f) ? w : coarse;                      // no offset beat 0: best_offset stays at coarse_lts_start
}

// Out-of-line entries of the search phases.  The 1024-point instance inlined into ONE function needs far more
// than the 168 VGPRs that three wavefronts per SIMD leave (81 spilled, reloaded inside the FFTs); as separate
// functions every phase gets its own register allocation (a handful spilled) and the calls — a few hundred per
// stream, each worth thousands of cycles — cost nothing measurable: 27.5 -> 25.0 ms per 16384 cfg3 streams.
// The 512-point instance fits inlined and is faster that way (6.8 against 7.7 ms).
// (TAG: one copy per kernel instance — with two kernels calling ONE copy the callee is compiled for an unknown caller and
// the search kernel lost 9 %: 128.7 -> 140.4 ms per 65,536 streams)
template <int LOG2N, int TAG>
__device__ __attribute__((noinline)) float acq_group_energy_call(AcqShared<LOG2N>& sh, const float* __restrict__ all,
                                                                 unsigned first, int count, unsigned n_samples) {
    return acq_group_energy<LOG2N>(sh, all, first, count, n_samples);
}
template <int LOG2N, int TAG>
__device__ __attribute__((noinline)) float acq_group_dc_call(AcqShared<LOG2N>& sh, const float* __restrict__ all,
                                                             unsigned first, unsigned n_samples) {
    return acq_group_dc<LOG2N>(sh, all, first, n_samples);
}
template <int LOG2N, int TAG>
__device__ __attribute__((noinline)) void acq_window_metric_call(AcqShared<LOG2N>& sh, const AcqLaneTw<LOG2N>& ltw,
                                                                 const float* __restrict__ win, float dc_sum, c32* P_out,
                                                                 float* R1, float* R2) {
    acq_window_metric<LOG2N>(sh, ltw, win, dc_sum, P_out, R1, R2);
}
template <int LOG2N, bool MIDFRAME, bool GC = false>
__global__ __launch_bounds__(kWave, (LOG2N == 10) ? 3 : 4) void acquire_kernel(
    const DemodConst* __restrict__ Dp, const c32* __restrict__ twiddle, const float* __restrict__ lts_I,
    const float* __restrict__ lts_Q, float energy_ref, float sync_threshold, const float* __restrict__ audio,
    size_t stream_stride, unsigned n_samples, unsigned chunk, int n_streams, unsigned* __restrict__ found_out,
    unsigned* __restrict__ data_start_out, float* __restrict__ cfo_out, unsigned* __restrict__ sync_offset_out,
    unsigned* __restrict__ fed_out, unsigned origin, unsigned* __restrict__ resume, unsigned* __restrict__ gcache = nullptr) {
    // GC (live streams only): the metric cache of stream s is gcache + s * kAcqGWords in HBM (see kAcqGCache) instead of the
    // launch's LDS; acq_prepass_kernel has filled it for every candidate of the buffer before this kernel starts.
    // origin / resume (ultra_hip_acquire_stream_batch): the search of a LIVE stream, one process() call per launch.
    // Sample index i of stream s lives at audio[s * stream_stride + i - origin] (the caller keeps only the part the
    // search can still look at), and resume[s] = {base, fed, noise floor, -} is what OFDMDemodulator::Impl carries
    // from one process() call to the next in the SEARCHING state: where rx_buffer starts, how much was fed, and the
    // energy gate's noise floor — the only state of the search (everything else is recomputed from the samples).
    // MIDFRAME (an instance of its own, ultra_hip_resync_stream_batch): the preamble check of the SYNCED state (demodulator.cpp:605-657) on the
    // buffer [resume[4 s], n_samples): offsets 0, 8, .. <= min(size - 6 preamble symbols, 2 data symbols), no energy gate,
    // no plateau test — the first offset above the threshold whose LTS confirmation holds wins; resume is only read.
    constexpr int N = 1 << LOG2N;
    constexpr bool midframe = MIDFRAME;
    constexpr bool kCalls = (LOG2N == 10);                   // phases as out-of-line functions, see above
    __shared__ AcqShared<LOG2N> sh;
    const DemodConst& D = *Dp;
    const int lane = threadIdx.x;
    AcqLaneTw<LOG2N> ltw;
    ltw.table = twiddle;
    {
        constexpr int P = AcqShared<LOG2N>::P, A = AcqShared<LOG2N>::A;
        for (int idx = lane; idx < AcqShared<LOG2N>::kTwB; idx += kWave) {
            const int sA = 31 - __clz(idx / P + 1);          // stage - A: runs start at P*(2^sA - 1)
            const int k = idx - P * ((1 << sA) - 1);
            sh.twB[idx] = twiddle[k << (LOG2N - 1 - (A + sA))];
        }
        if (lane < AcqShared<LOG2N>::kTwA) {                  // stage s < A: k < 2^s, run starts at 2^s - 1
            const int s0 = 31 - __clz(lane + 1);
            sh.twA[lane] = twiddle[(lane - ((1 << s0) - 1)) << (LOG2N - 1 - s0)];
        }
        wave_sync();
    }
    const unsigned psl = (unsigned)(N + D.cp), preamble_total = psl * 6u, corr_win = psl * 2u;
    const int gate_count = (int)((corr_win + 15u) / 16u);    // i = 0, 16, .. < window_len
    for (int stream = blockIdx.x; stream < n_streams; stream += gridDim.x) {
        const float* all = audio + (size_t)stream * stream_stride - origin;
        unsigned base = 0, fed = 0, found = 0, so_out = 0, ds_out = 0, fed_at = 0;
        float cfo = 0.0f, noise_floor = 0.0f;
        if (resume) { base = resume[4 * stream]; fed = resume[4 * stream + 1]; noise_floor = __uint_as_float(resume[4 * stream + 2]); }
#ifdef UH_ACQ_STAMPS
        unsigned long long aq_acc[kAcqStampWords] = {};
        const unsigned long long aq_start = __builtin_readcyclecounter();
#endif
        unsigned gate_first = 0xffffffffu;                    // window start of lane 0 of the current energy-gate group
        float gate_sum = 0.0f;                                // lane g: sum of squares of the gate window at gate_first + 8 g
        unsigned grp_first = 0xffffffffu;                     // window start of lane 0 of the current DC group
        float grp_dc = 0.0f;                                  // lane g: dc sum of the window at grp_first + 8 g
        if constexpr (!GC) { for (int c = lane; c < kAcqCache; c += kWave) sh.ctag[c] = 0xffffffffu; }
        wave_sync();
        // GC: 64 consecutive entries of the stream's cache in registers — lane g holds the entry of the window that starts
        // 8 g samples behind gc_first (one coalesced load per 64 candidates of the walk)
        unsigned* gtags = GC ? gcache + (size_t)stream * kAcqGWords : nullptr;
        float* gvals = GC ? reinterpret_cast<float*>(gtags + kAcqGCache) : nullptr;
        unsigned gc_first = 0xffffffffu, gc_tag = 0xffffffffu;
        float gc_val = 0.0f;
        if (midframe) fed = base;                             // one call over everything buffered
        while (fed < n_samples && !found) {
            fed += (midframe || n_samples - fed < chunk) ? (n_samples - fed) : chunk;
            UH_AQ_CNT(10);
            unsigned size = fed - base;
            if (midframe) {
                if (size < preamble_total) break;
            } else {
                if (size < kAcqMinSearch) continue;
                if (size > kAcqMaxBuffer) { base = fed - kAcqOverlap; size = kAcqOverlap; }
            }
            const float* buf = all + base;
            // The search (demodulator.cpp:497-531) as a state machine with ONE metric evaluation per turn:
            //   SEARCH   candidate i passes the energy gate -> metric at i; above threshold -> PLATEAU
            //   PLATEAU  metrics at i + j, j = 0, 8, .. <= 300 while i + j + preamble < size; then the verdict
            //   CFO      estimateCoarseCFO's correlation at the chosen offset (no dc removal), then done
            bool found_sync = false;
            unsigned so = 0;
            float c0 = 0.0f;
            unsigned search_end = (size > preamble_total + corr_win) ? size - preamble_total - corr_win : 0u;
            if (midframe) {                                           // offset <= search_limit (:611-615)
                const unsigned a = size - preamble_total, b = 2u * (unsigned)D.sym_len;
                search_end = ((a < b) ? a : b) + 1u;
            }
            enum { kSearch, kPlateau, kCfo };
            int mode = kSearch;
            unsigned i = 0, j = 0, plateau = 0, peak_pos = 0;
            float peak = 0.0f;
          rescan:
            found_sync = false;
            for (;;) {
                unsigned off;
                if (mode == kSearch) {
                    if (i >= search_end) break;
                    bool energetic = midframe;
                    if (!midframe && i + corr_win <= size) {         // hasMinimumEnergy: window inside the buffer
                        const unsigned gabs = base + i, d = gabs - gate_first;
                        if (gabs < gate_first || (d & 7u) != 0u || d >= 8u * kWave) {
                            gate_first = gabs;
                            UH_AQ_T0();
                            if constexpr (kCalls) gate_sum = acq_group_energy_call<LOG2N, (int)MIDFRAME + 2 * (int)GC>(sh, all, gabs, gate_count, n_samples);
                            else gate_sum = acq_group_energy<LOG2N>(sh, all, gabs, gate_count, n_samples);
                            UH_AQ_ADD(1); UH_AQ_CNT(7);
                        }
                        energetic = acq_energy_gate(lane_f(gate_sum, (int)((gabs - gate_first) >> 3)), gate_count, noise_floor);
                    }
                    if (!energetic) { i += corr_win / 2u; continue; }
                    off = i;
                } else if (mode == kPlateau) {
                    if (!(j <= kAcqPlateauWindow && i + j + preamble_total < size)) {
                        if (plateau >= kAcqMinPlateau) { so = peak_pos; mode = kCfo; continue; }
                        mode = kSearch; i += kAcqStep; continue;
                    }
                    off = i + j;
                } else {
                    off = so;
                }
                UH_AQ_CNT(9);
                // measureSchmidlCoxCorrelation / estimateCoarseCFO: window starts cp after the offset
                const bool in_range = off + (unsigned)D.cp + (unsigned)N <= size;
                c32 Pm = mk(0.0f, 0.0f);
                float R1 = 0.0f, R2 = 0.0f;
                const unsigned wabs = base + off + (unsigned)D.cp;       // absolute first sample of the window
                const unsigned slot = (wabs >> 3) & (unsigned)(kAcqCache - 1);
                bool cached = false;
                float corr = 0.0f;
                if (in_range && mode != kCfo) {                          // wave-uniform (broadcast reads made scalar)
                    if constexpr (GC) {
                        const unsigned d = wabs - gc_first;
                        if (wabs < gc_first || (d & 7u) != 0u || d >= 8u * kWave) {
                            gc_first = wabs;
                            const unsigned e = ((wabs >> 3) + (unsigned)lane) & (unsigned)(kAcqGCache - 1);
                            gc_tag = gtags[e]; gc_val = gvals[e];
                        }
                        const int l = (int)((wabs - gc_first) >> 3);
                        const unsigned tag = (unsigned)__builtin_amdgcn_readlane((int)gc_tag, l);
                        if (tag == wabs) { cached = true; corr = lane_f(gc_val, l); }
                    } else {
                        const unsigned tag = (unsigned)__builtin_amdgcn_readfirstlane((int)sh.ctag[slot]);
                        const float val = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(sh.cval[slot])));
                        if (tag == wabs) { cached = true; corr = val; }
                    }
                }
#ifdef UH_ACQ_STAMPS
                if (cached) UH_AQ_CNT(11);
#endif
                if (in_range && !cached) {
                    float dc_sum = 0.0f;
                    if (mode != kCfo) {
                        const unsigned d = wabs - grp_first;              // candidate (d / 8) of the current group?
                        if (wabs < grp_first || (d & 7u) != 0u || d >= 8u * kWave) {
                            grp_first = wabs;
                            UH_AQ_T0();
                            if constexpr (kCalls) grp_dc = acq_group_dc_call<LOG2N, (int)MIDFRAME + 2 * (int)GC>(sh, all, wabs, n_samples);
                            else grp_dc = acq_group_dc<LOG2N>(sh, all, wabs, n_samples);
                            UH_AQ_ADD(2); UH_AQ_CNT(8);
                        }
                        dc_sum = lane_f(grp_dc, (int)((wabs - grp_first) >> 3));
                    }
                    UH_AQ_T0();
                    if constexpr (kCalls) acq_window_metric_call<LOG2N, (int)MIDFRAME + 2 * (int)GC>(sh, ltw, all + wabs, dc_sum, &Pm, &R1, &R2);
                    else acq_window_metric<LOG2N>(sh, ltw, all + wabs, dc_sum, &Pm, &R1, &R2);
#ifdef UH_ACQ_STAMPS
                    if (mode == kCfo) { UH_AQ_ADD(4); } else { UH_AQ_ADD(3); UH_AQ_CNT(6); }
#endif
                }
                if (mode == kCfo) {
                    if (in_range) {
                        const float phase = um::atan2f_(Pm.im, Pm.re);
                        // float cfo_hz = phase * config.sample_rate / (M_PI * fft_len): float * uint32 -> float, / double
                        const float cfo_hz = (float)((double)(phase * D.sample_rate) / (kPi * (double)N));
                        const float max_cfo = (float)((unsigned)D.sample_rate / (unsigned)N);   // integer division in the reference
                        c0 = fmax_std(-max_cfo, fmin_std(max_cfo, cfo_hz));
                    }
                    found_sync = true;
                    break;
                }
                if (in_range && !cached) {
                    const float normalization = sqrtf(R1 * R2);
                    corr = (normalization < 1e-10f) ? 0.0f : cabs_(Pm) / normalization;
                    if constexpr (GC) {
                        // a candidate the prepass did not cover (a buffer longer than the cache's span): into the cache and,
                        // if it belongs to the group in registers, into its lane
                        const unsigned e = (wabs >> 3) & (unsigned)(kAcqGCache - 1);
                        if (lane == 0) { gtags[e] = wabs; gvals[e] = corr; }
                        const unsigned d = wabs - gc_first;
                        if (wabs >= gc_first && (d & 7u) == 0u && d < 8u * kWave && lane == (int)(d >> 3)) { gc_tag = wabs; gc_val = corr; }
                    } else {
                        wave_sync();
                        if (lane == 0) { sh.ctag[slot] = wabs; sh.cval[slot] = corr; }
                        wave_sync();
                    }
                }
                if (mode == kSearch) {
                    if (corr > sync_threshold) {
                        if (midframe) { so = i; mode = kCfo; }
                        else { mode = kPlateau; j = 0; plateau = 0; peak = corr; peak_pos = i; }
                    } else i += kAcqStep;
                } else {
                    if (corr >= 0.90f) plateau++;
                    if (corr > peak) { peak = corr; peak_pos = i + j; }
                    j += 8u;
                }
            }
            if (found_sync) {
                // inlined in both instances: its templates then come through scalar loads (kernel arguments), out of
                // line they are per-lane flat loads the loop waits for at every tap (22.2 against 20.1 ms)
                UH_AQ_T0();
                const unsigned refined = acq_refine_lts<LOG2N>(sh, D, lts_I, lts_Q, energy_ref, buf, size, so);
                UH_AQ_ADD(5);
                if (refined == 0xffffffffu) {
                    if (midframe) { i = so + kAcqStep; mode = kSearch; goto rescan; }     // "continue" of :621-624
                    if (size > kAcqOverlap * 2u) {
                        unsigned trim = so + psl;
                        if (trim > size - kAcqOverlap) trim = size - kAcqOverlap;
                        base += trim;
                    }
                } else {
                    found = 1; fed_at = fed; so_out = so; cfo = c0;
                    ds_out = base + refined + 2u * psl;
                }
            } else if (!midframe && size > kAcqOverlap * 2u) {
                base += size - kAcqOverlap;
            }
        }
#ifdef UH_ACQ_STAMPS
        if (g_acq_stamps != nullptr && lane == 0) {
            aq_acc[0] = __builtin_readcyclecounter() - aq_start;
            for (int q = 0; q < kAcqStampWords; ++q) g_acq_stamps[(size_t)stream * kAcqStampWords + q] = aq_acc[q];
        }
#endif
        if (lane == 0 && resume && !midframe) { resume[4 * stream] = base; resume[4 * stream + 1] = fed; resume[4 * stream + 2] = __float_as_uint(noise_floor); }
        if (lane == 0) {
            found_out[stream] = found;
            data_start_out[stream] = ds_out;
            cfo_out[stream] = cfo;
            if (sync_offset_out) sync_offset_out[stream] = so_out;
            if (fed_out) fed_out[stream] = fed_at;
        }
    }
}

// Guard of the live streams' metric caches, one wavefront per stream, in front of every launch of the pair below: a stream
// that starts afresh (resume fed == 0: a new demodulator, or a recycled context) or whose sample indices went backwards (the
// host rebased them) must not find the previous occupant's entries — tags are absolute sample indices.
__global__ __launch_bounds__(kWave) void acq_cache_guard_kernel(const unsigned* __restrict__ resume, unsigned n_samples,
                                                                int n_streams, unsigned* __restrict__ gcache) {
    const int lane = threadIdx.x;
    for (int stream = blockIdx.x; stream < n_streams; stream += gridDim.x) {
        unsigned* g = gcache + (size_t)stream * kAcqGWords;
        unsigned* hdr = g + 2 * kAcqGCache;
        // ... or whose owner says so: word 3 of the resume record is an epoch the owner changes whenever it restarts its sample
        // indices for any other reason (ultra_hip.h)
        // (resume == nullptr: the batch entry — every stream is a fresh demodulator)
        const unsigned fed = resume ? resume[4 * stream + 1] : 0u, epoch = resume ? resume[4 * stream + 3] : 0u, last = hdr[0], last_epoch = hdr[1];
        if (fed == 0u || fed < last || last == 0u || epoch != last_epoch)
            for (int e = lane; e < kAcqGCache; e += kWave) g[e] = 0xffffffffu;
        wave_sync();
        if (lane == 0) { hdr[0] = (n_samples > 0u) ? n_samples : 1u; hdr[1] = epoch; }
    }
}

// The metric of every candidate of the buffers that the caches do not hold yet — in practice those whose window the call's
// new samples completed — one wavefront per candidate, in parallel (the walk would evaluate them one after the other):
// candidate c of stream s = the window that starts at base + 8 c + cp, in range iff it ends inside the samples fed
// (acquire_kernel: in_range).  Same functions, same operands as the walk's own evaluation — the in-order DC sum
// (acq_group_dc, lane 0), Impl::toAnalytic, the half-symbol sums — so the cached value IS the value the walk would compute.
// Candidates the energy gate would have skipped are evaluated too: the price of not walking.
template <int LOG2N>
__global__ __launch_bounds__(kWave, (LOG2N == 10) ? 3 : 4) void acq_prepass_kernel(
    const DemodConst* __restrict__ Dp, const c32* __restrict__ twiddle, const float* __restrict__ audio, size_t stream_stride,
    unsigned n_samples, unsigned origin, const unsigned* __restrict__ resume, unsigned* __restrict__ gcache, unsigned cand_per_stream) {
    constexpr int N = 1 << LOG2N;
    __shared__ AcqShared<LOG2N> sh;
    const DemodConst& D = *Dp;
    const int lane = threadIdx.x;
    const unsigned stream = blockIdx.x / cand_per_stream, c = blockIdx.x % cand_per_stream;
    const unsigned base = resume[4 * stream];
    unsigned size = n_samples - base;
    if (size > kAcqMaxBuffer) return;                                  // the walk trims first (:478-483): it evaluates what it needs itself
    const unsigned off = 8u * c;
    if (size < kAcqMinSearch || off + (unsigned)D.cp + (unsigned)N > size) return;
    const unsigned wabs = base + off + (unsigned)D.cp;
    unsigned* gtags = gcache + (size_t)stream * kAcqGWords;
    float* gvals = reinterpret_cast<float*>(gtags + kAcqGCache);
    const unsigned e = (wabs >> 3) & (unsigned)(kAcqGCache - 1);
    if (gtags[e] == wabs) return;                                      // evaluated by an earlier call (wave-uniform)
    AcqLaneTw<LOG2N> ltw;
    ltw.table = twiddle;
    {
        constexpr int P = AcqShared<LOG2N>::P, A = AcqShared<LOG2N>::A;
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
    const float* all = audio + (size_t)stream * stream_stride - origin;
    const float dc_sum = lane_f(acq_group_dc<LOG2N>(sh, all, wabs, n_samples), 0);
    c32 Pm; float R1, R2;
    acq_window_metric<LOG2N>(sh, ltw, all + wabs, dc_sum, &Pm, &R1, &R2);
    const float normalization = sqrtf(R1 * R2);
    const float corr = (normalization < 1e-10f) ? 0.0f : cabs_(Pm) / normalization;
    if (lane == 0) { gvals[e] = corr; gtags[e] = wabs; }
}

}  // namespace dev
}  // namespace ultra_hip
#endif
