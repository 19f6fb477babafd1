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

template <int LOG2N>
struct AcqShared {
    static constexpr int N = 1 << LOG2N;
    union {
        c32 X[N];                 // FFT work buffer (in place)
        float samp[N];            // window samples for the dc sum (before the FFT input is written)
        float terms[N / 2][4];    // per-index terms of the four correlation sums (after the analytic signal is read)
    };
    c32 tw[N / 2];                // twiddle table
};

// s = 0; s += a[0]; s += a[1]; ... in order, every lane (broadcast reads); n a multiple of 16.
// The next 16 terms are requested before the current 16 are added, so the serial chain of adds
// (the floor: one add per term) runs without waiting for LDS.
__device__ __forceinline__ float acq_ordered_sum(const float* a, int n) {
    float s = 0.0f;
    float4 cur[4], nxt[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) cur[u] = *reinterpret_cast<const float4*>(a + 4 * u);
    for (int i = 0; i < n; i += 16) {
        const int j = (i + 16 < n) ? i + 16 : i;            // last round re-reads its own block (unused)
#pragma unroll
        for (int u = 0; u < 4; ++u) nxt[u] = *reinterpret_cast<const float4*>(a + j + 4 * u);
#pragma unroll
        for (int u = 0; u < 4; ++u) { s += cur[u].x; s += cur[u].y; s += cur[u].z; s += cur[u].w; }
#pragma unroll
        for (int u = 0; u < 4; ++u) cur[u] = nxt[u];
    }
    return s;
}

// In-place radix-2 DIT on bit-reversed input (FFT::fft_impl after its permutation), INVERSE: conj
// twiddles and 1/N scaling.  Lane handles butterflies b = lane + 64 q of every stage.
template <int LOG2N, bool INVERSE>
__device__ __forceinline__ void acq_fft_stages(AcqShared<LOG2N>& sh) {
    constexpr int N = 1 << LOG2N;
    const int lane = threadIdx.x;
    for (int s = 0; s < LOG2N; ++s) {
        const int half = 1 << s;
#pragma unroll
        for (int q = 0; q < N / 128; ++q) {
            const int b = lane + 64 * q;
            const int k = b & (half - 1);
            const int i0 = ((b >> s) << (s + 1)) + k, i1 = i0 + half;
            c32 w = sh.tw[k << (LOG2N - 1 - s)];
            if (INVERSE) w = cconj(w);
            const c32 a = sh.X[i0], t = cmul(w, sh.X[i1]);
            sh.X[i1] = csub(a, t);
            sh.X[i0] = cadd(a, t);
        }
        wave_sync();
    }
    if (INVERSE) {
        const float scale = 1.0f / (float)N;
#pragma unroll
        for (int q = 0; q < N / 64; ++q) { const int i = lane + 64 * q; sh.X[i] = cscale(sh.X[i], scale); }
        wave_sync();
    }
}

// Impl::toAnalytic for len == fft_size: samples (minus dc) -> analytic signal in sh.X (natural order)
template <int LOG2N>
__device__ __forceinline__ void acq_analytic(AcqShared<LOG2N>& sh, const float* __restrict__ win, float dc) {
    constexpr int N = 1 << LOG2N;
    const int lane = threadIdx.x;
#pragma unroll
    for (int q = 0; q < N / 64; ++q) {
        const int i = lane + 64 * q;
        const int r = (int)(__brev((unsigned)i) >> (32 - LOG2N));
        sh.X[r] = mk(win[i] - dc, 0.0f);
    }
    wave_sync();
    acq_fft_stages<LOG2N, false>(sh);
    // freq[1..N/2) *= 2, freq(N/2..N) = 0, then the inverse transform's bit reversal
    c32 v[N / 64];
#pragma unroll
    for (int q = 0; q < N / 64; ++q) {
        const int i = lane + 64 * q;
        c32 f = sh.X[i];
        if (i >= 1 && i < N / 2) f = cscale(f, 2.0f);
        if (i > N / 2) f = mk(0.0f, 0.0f);
        v[q] = f;
    }
    wave_sync();
#pragma unroll
    for (int q = 0; q < N / 64; ++q) {
        const int i = lane + 64 * q;
        sh.X[(int)(__brev((unsigned)i) >> (32 - LOG2N))] = v[q];
    }
    wave_sync();
    acq_fft_stages<LOG2N, true>(sh);
}

// P = sum conj(a[i]) a[i+half], R1 = sum |a[i]|^2, R2 = sum |a[i+half]|^2 in index order.
// The terms overwrite the analytic signal (same LDS): all of it is read into registers first.
template <int LOG2N>
__device__ __forceinline__ void acq_half_sums(AcqShared<LOG2N>& sh, c32* P, float* R1, float* R2) {
    constexpr int N = 1 << LOG2N, H = N / 2;
    const int lane = threadIdx.x;
    float4 t4[H / 64];
#pragma unroll
    for (int q = 0; q < H / 64; ++q) {
        const int i = lane + 64 * q;
        const c32 x = sh.X[i], y = sh.X[i + H];
        const c32 t = cmul(cconj(x), y);
        t4[q] = make_float4(t.re, t.im, cnorm(x), cnorm(y));
    }
    wave_sync();
#pragma unroll
    for (int q = 0; q < H / 64; ++q) *reinterpret_cast<float4*>(&sh.terms[lane + 64 * q][0]) = t4[q];
    wave_sync();
    const int col = lane & 3;                       // four chains in four lanes
    float acc = 0.0f;
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
    *P = mk(lane_f(acc, 0), lane_f(acc, 1));
    *R1 = lane_f(acc, 2);
    *R2 = lane_f(acc, 3);
    wave_sync();
}

// Impl::measureSchmidlCoxCorrelation (buf = rx_buffer view, size = its length)
template <int LOG2N>
__device__ __forceinline__ float acq_sc(AcqShared<LOG2N>& sh, const float* __restrict__ buf, unsigned size,
                                        unsigned offset, int cp) {
    constexpr int N = 1 << LOG2N;
    const int lane = threadIdx.x;
    if (offset + (unsigned)cp + (unsigned)N > size) return 0.0f;
    const float* win = buf + offset + cp;
#pragma unroll
    for (int q = 0; q < N / 64; ++q) sh.samp[lane + 64 * q] = win[lane + 64 * q];
    wave_sync();
    const float dc_sum = acq_ordered_sum(sh.samp, N);
    const float dc = dc_sum / (float)N;
    wave_sync();
    acq_analytic<LOG2N>(sh, win, dc);
    c32 P; float R1, R2;
    acq_half_sums<LOG2N>(sh, &P, &R1, &R2);
    const float normalization = sqrtf(R1 * R2);
    if (normalization < 1e-10f) return 0.0f;
    return cabs_(P) / normalization;
}

// Impl::hasMinimumEnergy
template <int LOG2N>
__device__ __forceinline__ bool acq_has_energy(AcqShared<LOG2N>& sh, const float* __restrict__ buf, unsigned size,
                                               unsigned offset, unsigned window_len, float& noise_floor) {
    if (offset + window_len > size) return false;
    const int lane = threadIdx.x;
    const int count = (int)((window_len + 15u) / 16u);          // i = 0, 16, ... < window_len
    const int padded = (count + 15) & ~15;
    for (int t = lane; t < padded; t += 64) {
        float sq = -0.0f;                                           // pad: exact no-op in the sum
        if (t < count) { const float s = buf[offset + 16u * (unsigned)t]; sq = s * s; }
        sh.samp[t] = sq;
    }
    wave_sync();
    const float sum_sq = acq_ordered_sum(sh.samp, padded);
    wave_sync();
    const float energy = sum_sq / (float)count;
    if (noise_floor < 1e-20f) noise_floor = energy * 0.1f;
    if (energy < noise_floor) noise_floor = energy;
    else if (energy < noise_floor * 3.0f) noise_floor = (1.0f - 0.01f) * noise_floor + 0.01f * energy;
    const float threshold = noise_floor * 4.0f;
    return energy >= threshold;
}

// Impl::estimateCoarseCFO
template <int LOG2N>
__device__ __forceinline__ float acq_coarse_cfo(AcqShared<LOG2N>& sh, const DemodConst& D, const float* __restrict__ buf,
                                                unsigned size, unsigned sync_offset) {
    constexpr int N = 1 << LOG2N;
    const unsigned ds = sync_offset + (unsigned)D.cp;
    if (ds + (unsigned)N > size) return 0.0f;
    acq_analytic<LOG2N>(sh, buf + ds, 0.0f);          // x - 0.0f == x for every float (also -0.0f)
    c32 P; float R1, R2;
    acq_half_sums<LOG2N>(sh, &P, &R1, &R2);
    const float phase = um::atan2f_(P.im, P.re);
    // float cfo_hz = phase * config.sample_rate / (M_PI * fft_len): float * uint32 -> float, then / double
    const float cfo_hz = (float)((double)(phase * D.sample_rate) / (kPi * (double)N));
    const float max_cfo = (float)((unsigned)D.sample_rate / (unsigned)N);     // integer division in the reference
    return fmax_std(-max_cfo, fmin_std(max_cfo, cfo_hz));
}

// Impl::refineLTSTiming; returns 0xffffffff on failure.  Lane l evaluates offsets first + l + 64 r.
template <int LOG2N>
__device__ __forceinline__ unsigned acq_refine_lts(const DemodConst& D, const float* __restrict__ lts_I,
                                                   const float* __restrict__ lts_Q, float energy_ref,
                                                   const float* __restrict__ buf, unsigned size, unsigned sts_start) {
    constexpr int N = 1 << LOG2N;
    const int lane = threadIdx.x;
    const unsigned psl = (unsigned)(N + D.cp);
    const unsigned lts_len = psl;
    const unsigned coarse = sts_start + 4u * psl;
    const int back = (int)(3u * psl), fwd = (int)(psl / 2u);
    if (coarse < (unsigned)back || coarse + (unsigned)fwd + lts_len > size) return coarse;
    float best_corr = 0.0f;
    unsigned best_off = coarse;
    const int n_off = back + fwd + 1;
    for (int r0 = 0; r0 < n_off; r0 += 64) {
        const int idx = r0 + lane;
        const bool on = idx < n_off;
        const unsigned offset = coarse - (unsigned)back + (unsigned)(on ? idx : 0);
        float ci = 0.0f, cq = 0.0f, er = 0.0f;
        const float* p = buf + offset;
        for (unsigned i = 0; i < lts_len; i += 4) {
            float rx[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) rx[u] = p[i + u];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float ti = lts_I[i + u], tq = lts_Q[i + u];
                ci += rx[u] * ti;
                cq += rx[u] * tq;
                er += rx[u] * rx[u];
            }
        }
        const float corr_mag = sqrtf(ci * ci + cq * cq);
        const float norm = sqrtf(er * energy_ref);
        const float corr = (norm > 1e-6f) ? corr_mag / norm : 0.0f;
        if (on && corr > best_corr) { best_corr = corr; best_off = offset; }
    }
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

template <int LOG2N>
__global__ __launch_bounds__(kWave, 3) void acquire_kernel(
    const DemodConst* __restrict__ Dp, const c32* __restrict__ twiddle, const float* __restrict__ lts_I,
    const float* __restrict__ lts_Q, float energy_ref, float sync_threshold, const float* __restrict__ audio,
    size_t stream_stride, unsigned n_samples, unsigned chunk, int n_streams, unsigned* __restrict__ found_out,
    unsigned* __restrict__ data_start_out, float* __restrict__ cfo_out, unsigned* __restrict__ sync_offset_out,
    unsigned* __restrict__ fed_out) {
    constexpr int N = 1 << LOG2N;
    __shared__ AcqShared<LOG2N> sh;
    const DemodConst& D = *Dp;
    const int lane = threadIdx.x;
    for (int i = lane; i < N / 2; i += kWave) sh.tw[i] = twiddle[i];
    wave_sync();
    const unsigned psl = (unsigned)(N + D.cp), preamble_total = psl * 6u, corr_win = psl * 2u;
    for (int stream = blockIdx.x; stream < n_streams; stream += gridDim.x) {
        const float* all = audio + (size_t)stream * stream_stride;
        unsigned base = 0, fed = 0, found = 0, so_out = 0, ds_out = 0, fed_at = 0;
        float cfo = 0.0f, noise_floor = 0.0f;
        while (fed < n_samples && !found) {
            fed += (n_samples - fed < chunk) ? (n_samples - fed) : chunk;
            unsigned size = fed - base;
            if (size < kAcqMinSearch) continue;
            if (size > kAcqMaxBuffer) { base = fed - kAcqOverlap; size = kAcqOverlap; }
            const float* buf = all + base;
            bool found_sync = false;
            unsigned so = 0;
            const unsigned search_end = (size > preamble_total + corr_win) ? size - preamble_total - corr_win : 0u;
            for (unsigned i = 0; i < search_end; i += kAcqStep) {
                if (!acq_has_energy<LOG2N>(sh, buf, size, i, corr_win, noise_floor)) { i += corr_win / 2u - kAcqStep; continue; }
                const float corr = acq_sc<LOG2N>(sh, buf, size, i, D.cp);
                if (corr > sync_threshold) {
                    unsigned plateau = 0, peak_pos = i;
                    float peak = corr;
                    for (unsigned j = 0; j <= kAcqPlateauWindow && i + j + preamble_total < size; j += 8u) {
                        const float rc = acq_sc<LOG2N>(sh, buf, size, i + j, D.cp);
                        if (rc >= 0.90f) plateau++;
                        if (rc > peak) { peak = rc; peak_pos = i + j; }
                    }
                    if (plateau >= kAcqMinPlateau) { found_sync = true; so = peak_pos; break; }
                }
            }
            if (found_sync) {
                const float c0 = acq_coarse_cfo<LOG2N>(sh, D, buf, size, so);
                const unsigned refined = acq_refine_lts<LOG2N>(D, lts_I, lts_Q, energy_ref, buf, size, so);
                if (refined == 0xffffffffu) {
                    if (size > kAcqOverlap * 2u) {
                        unsigned trim = so + psl;
                        if (trim > size - kAcqOverlap) trim = size - kAcqOverlap;
                        base += trim;
                    }
                } else {
                    found = 1; fed_at = fed; so_out = so; cfo = c0;
                    ds_out = base + refined + 2u * psl;
                }
            } else if (size > kAcqOverlap * 2u) {
                base += size - kAcqOverlap;
            }
        }
        if (lane == 0) {
            found_out[stream] = found;
            data_start_out[stream] = ds_out;
            cfo_out[stream] = cfo;
            if (sync_offset_out) sync_offset_out[stream] = so_out;
            if (fed_out) fed_out[stream] = fed_at;
        }
    }
}

}  // namespace dev
}  // namespace ultra_hip
#endif
