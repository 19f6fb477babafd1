// chirp_kernel.h — batched dual-chirp synchronisation for gfx950: one wavefront per audio stream.
//
// Scope row f4: sync::ChirpSync::detectDualChirp (src/sync/chirp_sync.hpp:349-505) as
// OFDMChirpWaveform::detectSync uses it (src/waveform/ofdm_chirp_waveform.cpp:129-172): complex
// template correlation of the 500 ms up chirp over the whole buffer (coarse step 48, fine +-48,
// parabolic refinement), the same for the down chirp in a window behind it, CFO from the distance
// of the two peaks, CFO-corrected positions, and the sample at which the training symbols start —
// the PRESYNCED entry of the demodulator kernels.
//
// computeComplexTemplateCorrelation (:532-556) accumulates three float sums over the 24000 template
// taps serially; each lane evaluates one candidate position and keeps exactly that order.  The 64
// coarse candidates of a round are 48 samples apart: their samples sit in an LDS ring that is
// refilled as the taps advance (each sample is read from memory once per round), padded by one word
// per 48 so the lanes of a tap hit different banks; the template pair of a tap is wave-uniform and
// arrives in scalar registers.
#ifndef ULTRA_CHIRP_KERNEL_H
#define ULTRA_CHIRP_KERNEL_H

#include <hip/hip_runtime.h>
#include "device_types.h"
#include "demod_kernel.h"

namespace ultra_hip {
namespace dev {

constexpr int kChirpStep = 48;                         // COARSE_STEP; also the pad period of the sample ring
constexpr int kChirpStage = 2 * kChirpStep;            // taps per template stage
constexpr int kChirpBlock = 2 * kChirpStage;           // taps between two refills of the ring
constexpr int kChirpRing = 4096;                       // ring words (power of two)
constexpr int kChirpMirror = 64;                       // ring[0..64) repeated behind the ring: runs never wrap
constexpr int kChirpSpanCoarse = kChirpStep * 63;      // distance of the first and the last coarse candidate

struct ChirpShared {
    float ring[kChirpRing + kChirpMirror];
};

struct ChirpTemplates {                                // device pointers, `len` (cos, sin) pairs each
    const float2* up; const float2* dn;
    float e_up, e_dn;
    int len, gap;                                      // chirp samples, gap samples (detectDualChirp)
    int start_extra;                                   // chirp_samples + gap_samples of OFDMChirpWaveform::detectSync
    float cfo_to_samples;                              // sample_rate / chirp_rate
};

typedef float chirp_v2f __attribute__((ext_vector_type(2)));

// one tap of computeComplexTemplateCorrelation: corr_I += s * cos; corr_Q += s * sin; sig_energy += s * s
// (packed multiply, packed add: two roundings each, no contraction)
__device__ __forceinline__ void chirp_tap(float s, float2 t, chirp_v2f& ciq, float& se) {
    const chirp_v2f tt = {t.x, t.y}, ss = {s, s};
    ciq = ciq + ss * tt;
    se += s * s;
}

__device__ __forceinline__ float chirp_normalise(chirp_v2f ciq, float se, float energy) {
    const float denom = sqrtf(se * energy);
    if (denom < 1e-10f) return 0.0f;
    return sqrtf(ciq.x * ciq.x + ciq.y * ciq.y) / denom;
}

// Sixteen taps: template pairs in scalar registers (wave-uniform, read through the constant address space),
// the lane's samples in vector registers.
constexpr int kChirpGroup = 16;
typedef float chirp_f4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) chirp_f4 chirp_cquad;
struct ChirpGroup {
    chirp_f4 t[kChirpGroup / 2];                       // (cos, sin, cos, sin) of two taps each
    float s[kChirpGroup];
    __device__ __forceinline__ void load(const float* ps, const float2* __restrict__ tmpl, int tap0) {
        chirp_cquad* pt = (chirp_cquad*)(tmpl + tap0);
#pragma unroll
        for (int k = 0; k < kChirpGroup / 2; ++k) t[k] = pt[k];
#pragma unroll
        for (int r = 0; r < kChirpGroup; ++r) s[r] = ps[r];
    }
    __device__ __forceinline__ void accumulate(chirp_v2f& ciq, float& se) const {
#pragma unroll
        for (int r = 0; r < kChirpGroup; ++r) {
            const chirp_f4 q = t[r / 2];
            chirp_tap(s[r], (r & 1) ? make_float2(q.z, q.w) : make_float2(q.x, q.y), ciq, se);
        }
    }
};

// Correlation at x[first + STRIDE * lane ..] for every lane < n_pos: STRIDE 48 (coarse round; first is a
// multiple of 48) or 1 (fine round).
// Coarse: sample j (relative to first) lives at ring word (j + j / 48) & 4095: the lanes of a tap read words 49
// apart (conflict-free), a run of 48 taps starting at a multiple of 48 is 48 consecutive words, and the window
// of 63 * 48 + 2 * kChirpBlock samples (3479 words with padding) never overlaps itself.  Fine: word j & 4095.
// Every sample is fetched from memory once per round, one block ahead of its first use.  The loop is a software
// pipeline over groups of 16 taps: wait for group g + 1 (scalar loads share the LDS counter and return out of
// order, so the wait is for everything), issue the loads of group g + 2, accumulate group g + 1.
template <int STRIDE>
__device__ __forceinline__ float chirp_round(ChirpShared& sh, const float* __restrict__ x, int n, int first, int n_pos,
                                             const float2* __restrict__ tmpl, int len, float energy) {
    const int lane = threadIdx.x;
    constexpr int kSpan = STRIDE * (kWave - 1);
    auto fetch = [&](int j) { const int g = first + j; return (g < n) ? x[g] : 0.0f; };
    auto slot = [](int j) { return ((STRIDE == 1) ? j : j + j / kChirpStep) & (kChirpRing - 1); };
    auto put = [&](int j, float v) {
        const int a = slot(j);
        sh.ring[a] = v;
        if (a < kChirpMirror) sh.ring[a + kChirpRing] = v;
    };
    // the lane's samples of taps [48 run, 48 run + 48)
    auto run_base = [&](int run) {
        return sh.ring + ((STRIDE == 1) ? ((lane + kChirpStep * run) & (kChirpRing - 1))
                                        : (((kChirpStep + 1) * (lane + run)) & (kChirpRing - 1)));
    };
    wave_sync();                                       // the previous round's reads of the ring are done
    for (int j = lane; j < kSpan + kChirpBlock; j += kWave) put(j, fetch(j));
    wave_sync();
    chirp_v2f ciq = {0.0f, 0.0f};
    float se = 0.0f;
    constexpr int kNew = kChirpBlock / kWave;          // samples per lane a block adds to the window
    static_assert(kChirpBlock % kWave == 0 && kChirpStep % kChirpGroup == 0 && (kChirpBlock / kChirpGroup) % 2 == 0,
                  "whole lanes, whole groups, pairs of groups");
    constexpr int kGroupsPerBlock = kChirpBlock / kChirpGroup;
    const int full = len - len % kChirpBlock;
    ChirpGroup ga, gb;                                 // ping-pong: one is accumulated while the other is in flight
    auto group_ptr = [&](int tap) { return run_base(tap / kChirpStep) + tap % kChirpStep; };
    if (full > 0) ga.load(group_ptr(0), tmpl, 0);
    for (int i0 = 0; i0 < full; i0 += kChirpBlock) {
        float nx[kNew];
        const int jn = i0 + kChirpBlock + kSpan;
        const bool more = i0 + kChirpBlock < len;
#pragma unroll
        for (int k = 0; k < kNew; ++k) nx[k] = more ? fetch(jn + lane + kWave * k) : 0.0f;
#pragma unroll 1
        for (int g = 0; g < kGroupsPerBlock; g += 2) {
            int tap = i0 + g * kChirpGroup;
            // The empty asm statements pin the order the pipeline needs: [wait] < [issue next loads] and
            // [accumulate] < [next wait]; the compiler is otherwise free to sink the arithmetic below the wait.
            __builtin_amdgcn_s_waitcnt(0xc07f);        // lgkmcnt(0): group ga has arrived
            asm volatile("" : "+s"(tap));
            gb.load(group_ptr(tap + kChirpGroup), tmpl, tap + kChirpGroup);
            ga.accumulate(ciq, se);
            asm volatile("" : "+v"(se), "+v"(ciq));
            __builtin_amdgcn_s_waitcnt(0xc07f);        // group gb has arrived
            asm volatile("" : "+s"(tap));
            if (g == kGroupsPerBlock - 2) {            // the next group belongs to the next block: widen the window first
                if (more) {
#pragma unroll
                    for (int k = 0; k < kNew; ++k) put(jn + lane + kWave * k, nx[k]);
                }
                wave_sync();
            }
            // (the templates are padded by one group behind `len`)
            ga.load(group_ptr(tap + 2 * kChirpGroup), tmpl, tap + 2 * kChirpGroup);
            gb.accumulate(ciq, se);
            asm volatile("" : "+v"(se), "+v"(ciq));
        }
    }
    // taps behind the last whole block (none at 48 kHz: 24000 = 125 * 192)
    for (int t = full; t < len; ++t)
        chirp_tap(run_base(t / kChirpStep)[t % kChirpStep], tmpl[t], ciq, se);
    return (lane < n_pos) ? chirp_normalise(ciq, se, energy) : 0.0f;
}

// first occurrence of the largest value that beats `floor_corr` (strictly): (corr, pos) or (floor_corr, floor_pos)
__device__ __forceinline__ void chirp_pick(float corr, int pos, bool valid, float& best_corr, int& best_pos) {
    float m = (valid && corr > best_corr) ? corr : -1.0f;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if (m < 0.0f) return;                              // nobody beat the running best
    unsigned w = (valid && corr == m) ? (unsigned)pos : 0xffffffffu;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const unsigned y = (unsigned)__shfl_xor((int)w, o, 64); w = (y < w) ? y : w; }
    best_corr = m;
    best_pos = (int)w;
}

// detectChirpTemplate (chirp_sync.hpp:560-632) on x[0..n): returns position or -1, *corr_out = best correlation
__device__ __forceinline__ int chirp_detect_template(ChirpShared& sh, const float* __restrict__ x, int n,
                                                     const float2* __restrict__ tmpl, int len, float energy,
                                                     float threshold, float* corr_out) {
    const int lane = threadIdx.x;
    if (n < len) { *corr_out = 0.0f; return -1; }
    const int search_len = n - len;
    float best_corr = 0.0f;
    int best_pos = -1;
    // coarse search: pos = 0, 48, 96, .. < search_len; the running best is updated in position order
    const int n_coarse = (search_len + kChirpStep - 1) / kChirpStep;
    for (int c0 = 0; c0 < n_coarse; c0 += kWave) {
        const int cnt = (n_coarse - c0 < kWave) ? n_coarse - c0 : kWave;
        const float corr = chirp_round<kChirpStep>(sh, x, n, c0 * kChirpStep, cnt, tmpl, len, energy);
        chirp_pick(corr, (c0 + lane) * kChirpStep, lane < cnt, best_corr, best_pos);
    }
    if (best_pos < 0 || best_corr < threshold * 0.3f) { *corr_out = best_corr; return -1; }
    // fine search around the coarse peak (at most 97 positions: two rounds, their correlations are kept)
    const int fine_start = (best_pos - kChirpStep > 0) ? best_pos - kChirpStep : 0;
    const int fine_end = (best_pos + kChirpStep < search_len) ? best_pos + kChirpStep : search_len;
    float fine[2] = {0.0f, 0.0f};
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int p0 = fine_start + k * kWave;
        if (p0 > fine_end) break;
        const int cnt = (fine_end - p0 + 1 < kWave) ? fine_end - p0 + 1 : kWave;
        fine[k] = chirp_round<1>(sh, x, n, p0, cnt, tmpl, len, energy);
        chirp_pick(fine[k], p0 + lane, lane < cnt, best_corr, best_pos);
    }
    // parabolic interpolation: the neighbours' correlations were computed by the fine search unless the peak
    // sits on the edge of its range (the value at a position does not depend on the lane that computes it)
    if (best_pos > 0 && best_pos < search_len - 1) {
        float c0, c2;
        if (best_pos - 1 >= fine_start && best_pos + 1 <= fine_end) {
            const int r0 = best_pos - 1 - fine_start, r2 = best_pos + 1 - fine_start;
            c0 = (r0 < kWave) ? lane_f(fine[0], r0) : lane_f(fine[1], r0 - kWave);
            c2 = (r2 < kWave) ? lane_f(fine[0], r2) : lane_f(fine[1], r2 - kWave);
        } else {
            const float corr = chirp_round<1>(sh, x, n, best_pos - 1, 3, tmpl, len, energy);
            c0 = lane_f(corr, 0); c2 = lane_f(corr, 2);
        }
        const float c1 = best_corr;
        const float denom = 2.0f * (c0 - 2.0f * c1 + c2);
        if (fabsf(denom) > 1e-10f) {
            float delta = (c0 - c2) / denom;
            delta = fmax_std(-1.0f, fmin_std(1.0f, delta));
            best_pos = (int)roundf((float)best_pos + delta);
        }
    }
    *corr_out = best_corr;
    return (best_corr >= threshold) ? best_pos : -1;
}

// out per stream: detected, start_sample (training start), cfo_hz, correlation = max(up, down);
// optional: up_chirp_start, down_chirp_start
__global__ __launch_bounds__(kWave, 2) void chirp_sync_kernel(
    ChirpTemplates T, const float* __restrict__ audio, size_t stream_stride, int n_samples, int n_streams, float threshold,
    unsigned* __restrict__ detected, int* __restrict__ start_sample, float* __restrict__ cfo_out,
    float* __restrict__ corr_out, int* __restrict__ up_start_out, int* __restrict__ down_start_out) {
    __shared__ ChirpShared sh;
    const int lane = threadIdx.x;
    for (int stream = blockIdx.x; stream < n_streams; stream += gridDim.x) {
        const float* x = audio + (size_t)stream * stream_stride;
        const int n = n_samples, len = T.len, gap = T.gap;
        unsigned ok = 0;
        int start = -1, up_s = -1, dn_s = -1;
        float cfo = 0.0f, up_corr = 0.0f, dn_corr = 0.0f;
        if (n >= 2 * len + gap) {
            const int up_pos = chirp_detect_template(sh, x, n, T.up, len, T.e_up, threshold, &up_corr);
            if (up_pos >= 0) {
                const int ds = up_pos + len / 2;
                const int expected_down = up_pos + len + gap;
                int de = expected_down + 2 * len; if (de > n) de = n;
                if (ds < n) {
                    if (de <= ds + len) { de = ds + 2 * len; if (de > n) de = n; }
                    float dn_best = 0.0f;
                    const int dn_rel = chirp_detect_template(sh, x + ds, de - ds, T.dn, len, T.e_dn, threshold, &dn_best);
                    if (dn_rel >= 0) {
                        dn_corr = dn_best;               // DualChirpResult::down_correlation is only set once found
                        const int dn_pos = dn_rel + ds;
                        const int expected_gap = len + gap, actual_gap = dn_pos - up_pos;
                        const float gap_error = (float)(actual_gap - expected_gap);
                        cfo = gap_error / (2.0f * T.cfo_to_samples);
                        if (!(fabsf(cfo) > 100.0f)) {
                            const float up_correction = cfo * T.cfo_to_samples, down_correction = -cfo * T.cfo_to_samples;
                            up_s = (int)roundf((float)up_pos + up_correction);
                            dn_s = (int)roundf((float)dn_pos + down_correction);
                            start = dn_s + T.start_extra;
                            ok = 1;
                        }
                    }
                }
            }
        }
        if (lane == 0) {
            detected[stream] = ok;
            start_sample[stream] = start;
            cfo_out[stream] = cfo;                        // DualChirpResult::cfo_hz (set even when the sanity check rejects)
            corr_out[stream] = fmax_std(up_corr, dn_corr);
            if (up_start_out) up_start_out[stream] = up_s;
            if (down_start_out) down_start_out[stream] = dn_s;
        }
    }
}

// What the caller of IWaveform does between detectSync and process (tools/test_nvis_mode.cpp; OFDMChirpWaveform::
// process, src/waveform/ofdm_chirp_waveform.cpp:174-190): the frame is demodulated from start_sample with
// setFrequencyOffsetWithPhase(cfo, -2 pi cfo start / fs wrapped to [-pi, pi]).  The phase expression is evaluated
// in double (M_PI) and rounded to float; each wrap step is one double subtraction rounded to float.
//   entry[s]  = start sample, or 0xffffffff when no chirp pair was found or the frame does not fit the buffer
//   offset[s] = the same with 0 for unusable streams (demodulated from sample 0, outputs cleared afterwards)
__global__ __launch_bounds__(256) void chirp_entry_kernel(const unsigned* __restrict__ detected,
                                                          const int* __restrict__ start_sample,
                                                          const float* __restrict__ cfo_hz, unsigned frame_samples,
                                                          unsigned n_samples, unsigned sample_rate, int n_streams,
                                                          unsigned* __restrict__ entry, unsigned* __restrict__ offset,
                                                          float* __restrict__ cfo_use, float* __restrict__ phase_out) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_streams) return;
    const int start = start_sample[s];
    const bool ok = detected[s] != 0 && start >= 0 && (unsigned)start + frame_samples <= n_samples;
    entry[s] = ok ? (unsigned)start : 0xffffffffu;
    offset[s] = ok ? (unsigned)start : 0u;
    const float cfo = ok ? cfo_hz[s] : 0.0f;
    const double kPi = 3.14159265358979323846;
    float ph = (float)((((double)(-2.0f) * kPi) * (double)cfo) * (double)(unsigned)(ok ? start : 0) / (double)sample_rate);
    while ((double)ph > kPi) ph = (float)((double)ph - (double)2.0f * kPi);
    while ((double)ph < -kPi) ph = (float)((double)ph + (double)2.0f * kPi);
    cfo_use[s] = cfo;
    phase_out[s] = ph;
}

}  // namespace dev
}  // namespace ultra_hip
#endif
