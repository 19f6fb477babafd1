// demod_kernel.h — batched OFDM demodulation for gfx950: ONE WAVEFRONT PER FRAME, two kernels per symbol.
//
// Restates, with the reference's exact float/double operation order, the
// per-symbol chain of OFDMDemodulator (SYNCED loop src/ofdm/demodulator.cpp:
// 672-697, processPresynced :854-985):
//   toBaseband            src/ofdm/channel_equalizer.cpp:19-57
//   extractSymbol + FFT   src/ofdm/channel_equalizer.cpp:59-71, src/dsp/fft.cpp:89-121
//   estimateChannelFromLTS src/ofdm/channel_equalizer.cpp:77-328 (presynced entry)
//   updateChannelEstimate src/ofdm/channel_equalizer.cpp:330-595
//   interpolateChannel    src/ofdm/channel_equalizer.cpp:601-631
//   equalize              src/ofdm/channel_equalizer.cpp:728-840
//   demodulateSymbol      src/ofdm/demodulator.cpp:199-435 + src/ofdm/soft_demap.hpp
//
// Mapping (MI355X, wave64): frames are independent, symbols inside a frame are
// sequential (CFO / channel / noise tracking feed forward).  Each symbol is processed by
// two launches over the whole batch, one 64-lane wavefront per frame in both:
//   mix_fft_kernel   audio -> the 59/30 used FFT bins (streaming, no per-frame state in registers:
//                    high occupancy, the FFT and the double-precision sincos hide each other's latency)
//   track_kernel     pilots -> tracker update -> equalise -> LLRs (tiny, state record in HBM)
// The per-frame tracker state (channel estimate, previous pilots, scalars: < 1 KB) lives in a
// workspace record between the two; splitting the monolithic per-frame kernel this way removed
// its register spills and nearly doubled the resident waves per CU.
//   * mixing: every lane loads the 8/16 time samples it needs for the first FFT
//     stages straight from HBM (the wave reads 64 consecutive floats per load; the
//     cyclic prefix and guard are never fetched), multiplies by the NCO table and
//     by the CFO rotation.  The CFO phase of sample i comes from the exact closed
//     form of the reference's serial float recurrence (phase_table.h).
//   * FFT: the reference's radix-2 DIT butterflies with its float twiddle table,
//     executed in three register-resident groups of stages with two LDS transposes
//     in between; only the two output bins per lane that the carriers use are
//     produced by the last stages (pruning changes no surviving value).
//   * tracking: one lane per pilot / carrier; every float sum the reference forms
//     serially is accumulated in the same order from lane to lane with
//     v_readlane (all lanes hold the identical running sum) — bit-identical.
//   * demap: one lane per data carrier, LLRs stored straight to HBM.
// Device code is compiled with -ffp-contract=off.
#ifndef ULTRA_DEMOD_KERNEL_H
#define ULTRA_DEMOD_KERNEL_H

#include <hip/hip_runtime.h>
#include <utility>
#include "device_types.h"
#include "pinned_math.h"
#include "phase_table.h"

namespace ultra_hip {
namespace dev {

constexpr int kWave = 64;
constexpr int kPhaseCap = 48;      // phase segments per table round (typical symbols need 3..25)

// ---- complex helpers (std::complex<float> semantics of the reference build) ----
__device__ __forceinline__ c32 mk(float re, float im) { c32 r; r.re = re; r.im = im; return r; }
__device__ __forceinline__ c32 cadd(c32 a, c32 b) { return mk(a.re + b.re, a.im + b.im); }
__device__ __forceinline__ c32 csub(c32 a, c32 b) { return mk(a.re - b.re, a.im - b.im); }
__device__ __forceinline__ c32 cconj(c32 a) { return mk(a.re, -a.im); }
__device__ __forceinline__ c32 cmul(c32 x, c32 y) {
    float ac = x.re * y.re, bd = x.im * y.im, ad = x.re * y.im, bc = x.im * y.re;
    return mk(ac - bd, ad + bc);
}
// libgcc __divsc3 of the reference's runtime (libgcc_s 12): evaluated in double.
__device__ __forceinline__ c32 cdiv(c32 x, c32 y) {
    double a = x.re, b = x.im, c = y.re, d = y.im;
    double denom = (c * c) + (d * d);
    double xr = ((a * c) + (b * d)) / denom;
    double yi = ((b * c) - (a * d)) / denom;
    return mk((float)xr, (float)yi);
}
// Division by a BPSK pilot (+-1, +0): the double formula has denom == 1.0 and exact products, so
// x = a*c + b*d and y = b*c - a*d evaluated in float (products exact, one exact sum each, same
// signed-zero behaviour) give the very same bits without a double-precision divide.
__device__ __forceinline__ c32 cdiv_pilot(c32 x, c32 y) {
    if (y.im == 0.0f && !(__float_as_uint(y.im) >> 31) && fabsf(y.re) == 1.0f)
        return mk(x.re * y.re + x.im * y.im, x.im * y.re - x.re * y.im);
    return cdiv(x, y);
}
__device__ __forceinline__ c32 cscale(c32 a, float s) { return mk(a.re * s, a.im * s); }
__device__ __forceinline__ c32 cdivf(c32 a, float s) { return mk(a.re / s, a.im / s); }
__device__ __forceinline__ float cnorm(c32 a) { return a.re * a.re + a.im * a.im; }
__device__ __forceinline__ float cabs_(c32 a) { return um::hypotf_(a.re, a.im); }
__device__ __forceinline__ float carg_(c32 a) { return um::atan2f_(a.im, a.re); }
__device__ __forceinline__ c32 cexpj(float t) { float sn, cs; um::sincosf_(t, &sn, &cs); return mk(cs, sn); }
// the same values for |t| < 120 (or NaN), without the range branches of sincosf: the tracker's angles are an
// atan2f result or a timing phase 2 pi k tau / N with |tau| <= 50 N / 512 (|t| < 20)
__device__ __forceinline__ c32 cexpj_bounded(float t) { float sn, cs; um::sincosf_bounded_(t, &sn, &cs); return mk(cs, sn); }
__device__ __forceinline__ float fmin_std(float a, float b) { return (b < a) ? b : a; }  // std::min(a, b)
__device__ __forceinline__ float fmax_std(float a, float b) { return (a < b) ? b : a; }  // std::max(a, b)

// A workgroup is ONE wavefront: lanes exchange data through LDS in program order (the LDS
// pipeline executes a wave's instructions in order), so the only thing needed between a store
// and a dependent load by another lane is that the COMPILER keeps the order.  Unlike
// __syncthreads() this emits no s_barrier and — important for the asynchronous HBM->LDS
// prefetch — no s_waitcnt vmcnt(0).
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

constexpr double kPi = 3.14159265358979323846;
constexpr double kTwoPi = 2.0 * kPi;

__device__ __forceinline__ float clip_llr(float llr) {  // soft_demap.hpp:22-29
    float c = fmax_std(-10.0f, fmin_std(10.0f, llr));
    if (fabsf(c) < 0.5f) c = (c >= 0) ? 0.5f : -0.5f;
    return c;
}

// float timing_phase = 2.0f * M_PI * k * timing_offset_samples / config.fft_size  (double expr).
// fft_size is a power of two: dividing the double by it is an exact exponent shift (v_ldexp_f64;
// the quotient of two float-derived factors is far from the subnormal range), not a 30-instruction
// double-precision division.
__device__ __forceinline__ float timing_phase_of(int k, float timing, int log2_fft) {
    return (float)ldexp((kTwoPi * (double)k) * (double)timing, -log2_fft);
}

// ---- in-order reductions across lanes: every lane ends with the same sum ----
// The reference accumulates serially (s = 0; s += x[0]; s += x[1]; ...), and float addition is not
// associative, so the order is kept.  Terms go through LDS: every lane reads them back with
// wave-uniform (broadcast) addresses and runs the serial chain itself — one v_add per term instead
// of a v_readlane + v_add pair, and chains of different sums run in different lanes (see
// track_pilot_kernel).
__device__ __forceinline__ float lane_f(float v, int i) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), i));
}
// buf: 64 floats of LDS, 16-byte aligned
__device__ __forceinline__ float ordered_sum(float* buf, float v, int n) {
    buf[threadIdx.x] = v;
    wave_sync();
    float s = 0.0f;
    int i = 0;
    for (; i + 4 <= n; i += 4) {
        const float4 q = *reinterpret_cast<const float4*>(buf + i);
        s += q.x; s += q.y; s += q.z; s += q.w;
    }
    for (; i < n; ++i) s += buf[i];
    wave_sync();
    return s;
}
// sum over the 64 lanes in NO particular order (DPP operands inside the 16-lane rows, then four readlanes), wave-uniform:
// for screens whose exact value is settled elsewhere
__device__ __forceinline__ float wave_sum_unordered(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));   // row_half_mirror
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true));   // row_mirror
    const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0)), b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32)), d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return (a + b) + (c + d);
}
// buf: 64 c32 of LDS
__device__ __forceinline__ c32 ordered_csum(c32* buf, c32 v, int n) {
    buf[threadIdx.x] = v;
    wave_sync();
    c32 s = mk(0.0f, 0.0f);
    for (int i = 0; i < n; ++i) s = cadd(s, buf[i]);
    wave_sync();
    return s;
}

// per-frame tracker scalars, identical in all lanes (src/ofdm/demodulator_impl.hpp:18-119)
struct Track {
    float freq_offset_hz, freq_offset_filtered, cfo_phase;
    float noise_variance, snr_linear, timing;
    c32 ppc, cpc;
    int cpc_init, snr_symbol_count, symbols_since_sync, has_prev, has_dprev;
};

// per-lane constants of the configuration (loaded once per workgroup)
struct LaneConst {
    int pilot_slot, pilot_fq, pilot_k;    // lane = pilot index
    c32 pilot_seq;
    int data_slot, data_fq, data_k;       // lane = data-carrier index
    int i_dst, i_lo, i_hi; float i_alpha; // lane = interpolation-table entry
    int slot_k;                           // lane = carrier slot
    c32 zc;                               // sync_sequence[lane % n_carriers] for data carrier `lane`
};

struct TrackShared {                                        // track_kernel
    c32 H[kMaxCarriers];                                    // channel_estimate by slot
    union {
        float terms[kMaxCarriers][8];                       // per-pilot terms of the eight serial sums
        c32 cbuf[kMaxCarriers];                             // staging of ordered_csum
        float fbuf[kMaxCarriers];                           // staging of ordered_sum
    } __attribute__((aligned(16)));
};

// Per-frame record between the kernels (floats).  Arrays are c32 indexed by lane.
constexpr int kStScal = 0;          // 16 scalars, see st_* below
constexpr int kStH = 16;            // c32 H[64]
constexpr int kStPrev = 16 + 128;   // c32 prev_pilot_phases[64]
constexpr int kStDprev = 16 + 256;  // c32 dbpsk_prev_equalized[64]
constexpr int kStLts = 16 + 384;    // c32 h_sum_pilot[<= 32 pilots] (presynced)
// adaptive equaliser (coherent layouts only, so the differential reference's slot is free): lms_weights by data carrier,
// rls_P by data carrier in the unused half of the LTS sums' slot
constexpr int kStLms = kStDprev;    // c32 lms_weights[64]
constexpr int kStRlsP = kStLts + 64; // float rls_P[64]
// Coherent layouts with pilots: between symbols only the PILOTS' channel estimates are state (every other carrier is
// interpolated afresh per symbol, channel_equalizer.cpp:515-567), so they travel by pilot index in one cache line of
// their own — the pilot half then reads and writes 128 contiguous bytes instead of every fourth entry of H[64], and the
// carrier half does not move the 512-byte H array at all.  (The two tracking kernels run at the memory system's
// practical ceiling; this takes a third of their traffic.)  Records are 19 cache lines; the pilots' region starts on a line.
constexpr int kStHp = 16 + 512 + 16;                    // c32 Hp[32] at byte 2176 = line 17 of the record
constexpr int kStFloats = kStHp + 64;                   // 2432 bytes = 19 lines
enum { st_cfo = 0, st_cfo_filt, st_cfo_phase, st_noise, st_snr, st_timing, st_ppc_re, st_ppc_im, st_cpc_re, st_cpc_im,
       st_flags, st_count, st_since };
// deferred carrier half (track_all_kernel): per (symbol, frame) record from track_pilot_kernel
constexpr int kPwPilots = 32;                               // pilots per frame the record holds
constexpr int kTrkRecFloats = 96;                           // three cache lines: 8 scalars (below), then c32 Hp_derotated[<= 32]
constexpr int kTrkRecHp = 8;                                // (<= 12 pilots touch one line, the headline's 15 two)
enum { tk_noise = 0, tk_timing, tk_cfo, tk_snr, tk_phase, tk_count };
// Up to 15 pilots (the 59-carrier presets) the record is ONE cache line: the two scalars the carrier half reads (noise
// variance, timing) and the pilots' estimates, 2 + 2 x 15 = 32 floats — the other scalars only ever fed state_out, which
// track_all_kernel now takes from the tracker record.  One line less written per (symbol, frame) by the pilot half and one
// less read by the carrier half: 1.1 GB of the headline step's 36.
constexpr int kTrkRecFloatsCompact = 32, kTrkRecHpCompact = 2, kTrkRecCompactPilots = 15;
__host__ __device__ inline int trk_rec_floats(int n_pilot) { return n_pilot <= kTrkRecCompactPilots ? kTrkRecFloatsCompact : kTrkRecFloats; }
__host__ __device__ inline int trk_rec_hp(int n_pilot) { return n_pilot <= kTrkRecCompactPilots ? kTrkRecHpCompact : kTrkRecHp; }
// Fq row of a frame (and symbol): c32[2 * D.fq_half] = bins [0, fq_half) then [N - fq_half, N), fq_half = 32 or 64
__device__ __forceinline__ int fq_natural(const DemodConst& D, int bin) { return (bin < D.fq_half) ? bin : D.fq_half + (bin - (D.fft - D.fq_half)); }
__device__ __forceinline__ int fq_index(const DemodConst& D, int bin) { return D.fq_pos[fq_natural(D, bin)]; }      // pilots first: DemodConst::fq_pos
// Per-frame phase table of the next symbol's CFO rotation (cfo_walk_kernel -> mix_fft2_kernel),
// 32-bit words: [0] number of segments | samples covered << 8, [1] the tracker's CFO in Hz (float bits), [2] phase after
// the covered samples (float bits), [3] phase the symbol starts with; then {start, base, step} per segment.  It carries
// everything the transform needs to know about the frame: mix_fft never reads the tracker record.
constexpr int kSegTabWords = 4 + 3 * kPhaseCap;

__device__ __forceinline__ bool compact_pilot_state(const DemodConst& D) { return !D.differential && D.n_pilot > 0; }

template <int A> __device__ __forceinline__ constexpr int bitrev_small(int q) {
    int r = 0;
    for (int b = 0; b < A; ++b) r |= ((q >> b) & 1) << (A - 1 - b);
    return r;
}

#define UH_BUTTERFLY(a, b, w)            \
    do {                                 \
        const c32 t_ = cmul((w), (b));   \
        (b) = csub((a), t_);             \
        (a) = cadd((a), t_);             \
    } while (0)

// ---------------------------------------------------------------------------
// Diagnostic build only (-DUH_MIXFFT_STAMPS, tools/mix_fft_stalls.py): shader-clock stamps at the phase boundaries of
// every work item of mix_fft2_kernel, one record of kStampWords 64-bit words per wavefront and item:
// [0..kStampPhases] clock, then HW_ID (wave slot / SIMD / CU / SE) and XCC_ID.  The product build contains none of it.
constexpr int kStampPhases = 10, kStampExtra = 4, kStampWords = 16;   // extra: sub-stamps inside the lookup phase (mix_fft2)
#ifdef UH_MIXFFT_STAMPS
__device__ unsigned long long* g_mix_stamps = nullptr;
struct Stamps {
    unsigned long long t[kStampPhases + 1 + kStampExtra] = {};
    // s_memtime, followed by a marker comment in the ISA so that the tool can cut the static code into the same phases
    template <int K> __device__ __forceinline__ void at() { t[K] = __builtin_readcyclecounter(); asm volatile("; UHSTAMP %0" ::"n"(K)); }
    __device__ __forceinline__ void store(size_t record, int lane) {
        if (g_mix_stamps == nullptr || lane != 0) return;
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        unsigned long long* r = g_mix_stamps + record * kStampWords;
        for (int k = 0; k <= kStampPhases; ++k) r[k] = t[k];
        r[kStampPhases + 1] = ((unsigned long long)xcc << 32) | hw;
        for (int k = 0; k < kStampExtra; ++k) r[kStampPhases + 2 + k] = t[kStampPhases + 1 + k];
    }
};
#define UH_STAMP(k) stamps.template at<k>()
#else
struct Stamps { __device__ __forceinline__ void store(size_t, int) {} };
#define UH_STAMP(k) do {} while (0)
#endif

// minimum / maximum over the 64 lanes, wave-uniform result: DPP operands inside the 16-lane rows, then four readlanes
__device__ __forceinline__ float wave_fmin(float v) {
    v = fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true)));    // quad_perm [1,0,3,2]
    v = fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true)));    // quad_perm [2,3,0,1]
    v = fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true)));   // row_half_mirror
    v = fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true)));   // row_mirror
    const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0)), b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32)), d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return fminf(fminf(a, b), fminf(c, d));
}
__device__ __forceinline__ float wave_fmax(float v) { return -wave_fmin(-v); }

// ---------------------------------------------------------------------------
// TWO WAVEFRONTS PER FRAME (N = 1024).  The radix-2 decimation-in-time network of fft_impl (fft.cpp:89-121) works
// on the bit-reversed input: positions [0, N/2) hold the EVEN time samples, [N/2, N) the odd ones, and stages
// 0 .. log2(N)-2 never cross that line — they are two independent N/2-point transforms (with the twiddles
// W_N^(2k) = twiddle[k << (LOG2N-1-s)], the table entries a one-wavefront transform of all N points reads).  Only the last
// stage pairs element k of the even half with element k of the odd half.  So wavefront h of a 128-thread workgroup
// takes the samples of parity h — 8 points per lane instead of 16: the lane's oscillator values, rotation phases,
// sincos temporaries and butterfly registers all halve (<= 96 VGPRs instead of 154, five or six wavefronts per SIMD
// instead of three) — runs mixing, CFO rotation and nine stages on its own, exactly like the 512-point instance, and
// meets its partner ONCE per frame: the last stage needs E[k] + w O[k] for the bins k < 64 (wavefront 0 computes them
// from its own E and the partner's O) and E[k] - w O[k] for k >= 448 (bins N-64 .. N-1, wavefront 1).  Same operations
// on the same operands in the same order as a transform of all N points on one wavefront (round 2's kernel, removed in round 4): bit-identical bins.
template <int LOG2N, bool ROT = true>
struct Fft2Shared {
    static_assert(LOG2N == 10 || LOG2N == 9, "512 points per wavefront: N = 1024 on two wavefronts, N = 512 on one");
    static constexpr int W = (LOG2N == 10) ? 2 : 1;         // wavefronts per work item; W = 1: the same code on all N points
    static constexpr int N = 1 << LOG2N, M = N / W;         // M-point transform per wavefront
    static constexpr int P = M / kWave;                     // 8 points per lane
    static constexpr int A = 3;                             // log2(P); 3 groups of 3 stages + the joint last stage
    c32 X[W][M + M / P];                                    // per-wavefront exchange buffer, 1 pad per P entries
    float stage[W][M];                                      // landing zone of the NEXT item's samples (asynchronous copy)
    static constexpr int kTwB = P * ((1 << A) - 1);
    c32 twB[kTwB];                                          // twiddles of stages A..2A-1 (as in FftShared)
    // the rotation's lookup tables: not in the instance without it (15.8 KB instead of 17.9: ten workgroups per CU)
    um::PhaseSeg seg[W][ROT ? kPhaseCap + 2 : 1];           // two entries behind the last segment: start = INT_MAX
    int seg_start[W][ROT ? kPhaseCap + 4 : 4] __attribute__((aligned(16)));
    c32 xch[W == 2 ? 2 : 1][2][W == 2 ? kWave : 1];         // [frame parity][writer][lane]: E_hi from wavefront 0, O_lo from 1
    // The rotating instance keeps the oscillator at the lane's eight samples HERE instead of in sixteen registers it does not
    // have (the rotation factors stay in registers since round 4): osc[h][64 q + lane], written and read by the same lane.
    // 8 KB (N = 1024): six workgroups per CU still fit, which is what its registers allow anyway.
    c32 osc[ROT ? W : 1][ROT ? M : 1];
};

// What an item needs from memory besides its samples: the tracker's CFO and phase and this lane's entry of the frame's
// phase table.  Requested one item AHEAD (together with the samples), so that no memory round trip is left on the
// path of an item: the stamps of the first version (profiles/r03_mix_fft_stalls_two_wave_v1.txt) showed 60 % of a
// wavefront's time in four of them — staged audio, table, oscillator values, last twiddles.
struct MixItem {
    // Held as PER-LANE words until the item starts: a uniform value the compiler moves to a scalar register with
    // v_readfirstlane at once, i.e. it waits for the load — behind the asynchronous copy of the samples that was issued
    // just before it (loads return in order) — right where the request was meant to be fire-and-forget.
    unsigned hw;                // lanes 0..3: table header (segments | covered << 8, CFO, phase after them, start phase)
    int tab_start;              // lane k: segment k
    float tab_base, tab_step;
};
__device__ __forceinline__ void request_item(MixItem& it, const unsigned* __restrict__ tab, int lane) {
    it.hw = 0u; it.tab_start = 0x7fffffff; it.tab_base = 0.0f; it.tab_step = 0.0f;
    if (tab) {                  // no table <=> the CFO is zero for every frame of the launch (launch_demod): nothing to read
        it.hw = tab[lane & 3];
        const unsigned* e = tab + 4 + 3 * ((lane < kPhaseCap) ? lane : 0);      // entries behind the last segment: stale, masked at use
        it.tab_start = (int)e[0]; it.tab_base = __uint_as_float(e[1]); it.tab_step = __uint_as_float(e[2]);
    }
}

// compile-time loop: f(std::integral_constant<int, 0>{}), ..., f(<N - 1>)
template <class F, int... Is>
__device__ __forceinline__ void static_for_n_(F&& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for_n(F&& f) { static_for_n_(f, std::make_integer_sequence<int, N>{}); }

template <int W, int... Q>
__device__ __forceinline__ void prefetch_copies(const float* g0, unsigned zone, std::integer_sequence<int, Q...>) {
    using lds_f32 = __attribute__((address_space(3))) float;
    (__builtin_amdgcn_global_load_lds(g0, (lds_f32*)(size_t)(zone + 256u * (unsigned)Q - 256u * (unsigned)(W * Q)), 4, 256 * W * Q, 0), ...);
}

// staging of the samples of parity h of one symbol's FFT window: stage[64 q + l] = window[2 (64 q + l) + h] (W = 1: all
// samples, window[64 q + l])
template <int LOG2N, bool ROT>
__device__ __forceinline__ void prefetch_symbol2(Fft2Shared<LOG2N, ROT>& sh, const int cp, int h, int lane,
                                                 const float* __restrict__ audio_sym) {
    constexpr int P = Fft2Shared<LOG2N>::P, W = Fft2Shared<LOG2N>::W;
    float* stage = sh.stage[h];
    // ONE base kept alive across the items: left to itself the compiler precomputes the eight LDS addresses (M0 values) of the
    // copies once per workgroup, holds them in eight scalar registers the rotating instance does not have, spills them into
    // VGPR lanes and reads each back with a v_readlane per copy and item.  Behind this barrier an address is base + constant.
    using lds_f32 = __attribute__((address_space(3))) float;
    unsigned zone = (unsigned)(size_t)(lds_f32*)stage;      // the zone's 32-bit LDS address
    asm volatile("" : "+s"(zone));
    // Copy q reads 256 W bytes behind copy q - 1 and lands 256 bytes behind it.  The instruction's immediate offset moves
    // BOTH addresses, so it carries the global stride (256 W q <= 3584: fits the 13-bit field) and M0 takes the difference
    // back (the zone lies behind the exchange buffer: zone - 256 (W - 1) q stays a valid LDS address): one global address
    // for the eight copies instead of eight 64-bit vector additions per item.
    prefetch_copies<W>(audio_sym + cp + W * lane + h, zone, std::make_integer_sequence<int, P>{});
}

// per-lane values that do not change from item to item: the oscillator at the lane's 8 samples (for one symbol index)
// and the twiddles of the last four stages
template <int LOG2N>
struct Mix2Lane {
    static constexpr int P = Fft2Shared<LOG2N>::P;
    c32 os[P];
    c32 w6, w7[2], w8[2], w_last;
    c32 wA[P / 2];              // group A (wave-uniform): twiddle[j << (LOG2N - A)], j < P/2
};

// ROT = false: the launch has no phase table, i.e. the CFO of every frame is zero (launch_demod) — the rotation, its
// lookup and the sixteen sincosf per frame-lane are compiled out, and with them most of the kernel's registers.
template <int LOG2N, bool ROT, class NextFn>
__device__ __forceinline__ void symbol_to_freq2(Fft2Shared<LOG2N, ROT>& sh, const DemodConst& D, const int h, const int lane,
                                                MixItem& it, const Mix2Lane<LOG2N>& lc,
                                                const c32* __restrict__ twiddle, c32& bin_out, c32& bin_hi, const int par,
                                                const int cp, const int sym_len, NextFn request_next, Stamps& stamps) {
    // W = 2: the lane's bin (of its wavefront's side) in bin_out.  W = 1: bin `lane` in bin_out, bin N - 64 + lane in bin_hi.
    constexpr int P = Fft2Shared<LOG2N>::P, A = Fft2Shared<LOG2N>::A, W = Fft2Shared<LOG2N>::W;
    UH_STAMP(0);
    const int rl = (int)(__brev((unsigned)lane) >> 26);      // bitrev6(lane)
    const unsigned hw0 = (unsigned)__builtin_amdgcn_readlane((int)it.hw, 0);
    const float freq_offset_hz = __int_as_float(__builtin_amdgcn_readlane((int)it.hw, 1));
    float cfo_phase = __int_as_float(__builtin_amdgcn_readlane((int)it.hw, 3));
    const bool cfo_on = ROT && fabsf(freq_offset_hz) > 0.01f;
    const int it_ns = (int)(hw0 & 0xffu), it_covered = (int)((hw0 >> 8) & 0x7fffffu);
    const float it_pnext = __int_as_float(__builtin_amdgcn_readlane((int)it.hw, 2));
    c32* X = sh.X[h];
    um::PhaseSeg* seg = sh.seg[h];
    int* seg_start = sh.seg_start[h];
    c32 v[P];
    const float* stage = sh.stage[h];
    const int tab_ns = cfo_on ? it_ns : 0;
    // This item's samples have landed.  Nothing younger is in flight: the bins of the previous item are stored by
    // request_next() below, IN FRONT of the requests (a store at the end of the item would be the youngest operation here,
    // and vector-memory operations complete in order).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wave_sync();
    float xs[P];
#pragma unroll
    for (int qp = 0; qp < P; ++qp) xs[qp] = stage[64 * qp + rl];               // window sample W (rl + 64 qp) + h
    wave_sync();
    UH_STAMP(1);
    UH_STAMP(2);

    // ---- CFO rotation factors: lane l evaluates the P samples of its parity at local indices m = P l .. P l + P - 1
    //      (window positions 2 m + h: a run of 2 P positions, almost always inside one segment) ----
    c32* rot = X;                                          // rot[m + (m >> A)]
    float ph[P];
#pragma unroll
    for (int j = 0; j < P; ++j) ph[j] = 0.0f;
    bool bounded = true;
    if (cfo_on) bounded = (hw0 >> 31) != 0u;                 // walk_to_table: |start phase| <= 4 and |increment| <= 1
    // The common symbol: its whole window lies in ONE or TWO segments of the phase table (a CFO of a few hertz moves the
    // phase by a fraction of a radian per symbol: at most one binade boundary or wrap).  Then every lane's run is "segment 0
    // up to the start of segment 1, segment 1 from there on" with the SAME two segments — read once from the owners' lanes
    // as scalars; no table in LDS, no claims, no ballots, no cross-lane fetches.
    const bool two_segments = cfo_on && it_covered >= sym_len && tab_ns >= 1 && tab_ns <= 2;       // wave-uniform
    float range_lo = 0.0f, range_hi = 0.0f;               // two_segments: bounds of every phase of the window (see there)
    if (two_segments) {
        um::PhaseSeg s0, s1;
        s0.start = __builtin_amdgcn_readlane(it.tab_start, 0);
        s0.base = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(it.tab_base), 0));
        s0.step = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(it.tab_step), 0));
        s1.start = (tab_ns == 2) ? __builtin_amdgcn_readlane(it.tab_start, 1) : 0x7fffffff;
        s1.base = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(it.tab_base), 1));
        s1.step = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(it.tab_step), 1));
        // ... and with only two segments ANY eight ascending positions are "segment 0 up to a point, segment 1 from there":
        // the lane takes the phases of the very samples it mixes (window positions W (rl + 64 q) + h, 64 W apart) — the
        // rotation factors then never leave its registers: no exchange through LDS between the evaluation and the mixing.
        const int i0 = cp + W * rl + h, ilast = i0 + W * 64 * (P - 1);
        float a[P], b[P];
        const float step_a = s0.step * (float)(64 * W), step_b = s1.step * (float)(64 * W);      // exact (powers of two)
        a[0] = um::phase_table_eval(s0, i0);               // forward from the lane's first position: right while in segment 0
        b[P - 1] = um::phase_table_eval(s1, ilast);        // backward from its last: right while in segment 1
#pragma unroll
        for (int j = 1; j < P; ++j) { a[j] = a[j - 1] + step_a; b[P - 1 - j] = b[P - j] - step_b; }
#pragma unroll
        for (int j = 0; j < P; ++j) ph[j] = (i0 + W * 64 * j < s1.start) ? a[j] : b[j];
        // Inside a segment the phases are an arithmetic progression: over the window's positions cp .. cp + N - 1 (both
        // parities: a superset of this wavefront's, which is all the one-quadrant test below needs) they are bounded by the
        // values at the ends of the segments' parts — four scalar evaluations instead of two reductions over 512 phases.
        {
            constexpr int N = Fft2Shared<LOG2N>::N;
            const int first = cp, last = cp + N - 1;
            float lo = 3.402823466e+38f, hi = -3.402823466e+38f;
            if (s1.start > first) {                              // part of the window in segment 0
                const int e = (s1.start - 1 < last) ? s1.start - 1 : last;
                const float x = um::phase_table_eval(s0, first), y = um::phase_table_eval(s0, e);
                lo = fminf(x, y); hi = fmaxf(x, y);
            }
            if (s1.start <= last) {                              // part in segment 1
                const int b0 = (s1.start > first) ? s1.start : first;
                const float x = um::phase_table_eval(s1, b0), y = um::phase_table_eval(s1, last);
                lo = fminf(lo, fminf(x, y)); hi = fmaxf(hi, fmaxf(x, y));
            }
            range_lo = lo; range_hi = hi;
        }
#ifdef UH_MIXFFT_STAMPS
        stamps.t[kStampPhases + 4] = 1ull;
#endif
        UH_STAMP(11);
        UH_STAMP(12);
    } else if (cfo_on) {
        int done = 0;
        float pcur = cfo_phase;
        const bool have_tab = it_covered > 0;
        while (done < sym_len) {                           // one round unless a table overflows
            int covered;
            float pnext;
            int my_start = 0x7fffffff;
            float my_base = 0.0f, my_step = 0.0f;
            int ns;
            if (done == 0 && have_tab) {
                ns = tab_ns; covered = it_covered; pnext = it_pnext;
                if (lane < tab_ns) { my_start = it.tab_start; my_base = it.tab_base; my_step = it.tab_step; }
            } else {
                const float inc = (float)(((-kTwoPi) * (double)freq_offset_hz) / (double)D.sample_rate);
                ns = um::phase_table_walk(pcur, inc, sym_len - done, kPhaseCap, &covered, &pnext,
                                          [&](int k, int start, float base, float step) {
                                              const bool mine = (lane == k);
                                              my_start = mine ? start : my_start;
                                              my_base = mine ? base : my_base;
                                              my_step = mine ? step : my_step;
                                          });
            }
            if (lane < kPhaseCap + 2) { seg[lane].start = my_start; seg[lane].base = my_base; seg[lane].step = my_step; }
            if (lane < kPhaseCap + 4) seg_start[lane] = my_start;            // INT_MAX beyond the last segment
            wave_sync();
            UH_STAMP(11);
            const int i0 = cp + W * P * lane + h - done;  // position of the lane's first sample inside this round
            const int ilast = i0 + W * (P - 1);
            if (done == 0) {
                // The table round (positions i0 >= 0): the segment of the lane's first position WITHOUT a search.  Starts
                // ascend with the segment index k and first positions with the lane, so segment k claims the lanes from
                // r_k = ceil((start_k - first position of lane 0) / 2P) on, and lane l is in the LAST segment that claimed
                // a lane <= l: segment k pushes k + 1 to lane r_k (several segments may start inside one run around a
                // zero crossing: the largest arrives), lane l finds the highest claimed lane at or below l in the ballot
                // of the arrivals and fetches that lane's word.  Its own, the next and the next-but-one segment come straight out of the owners'
                // registers (ds_bpermute).  A run with at most one boundary is then two evaluations and a select per
                // sample; lanes whose run holds more boundaries (or leaves the round) walk on through the LDS table.
                const int first0 = cp + h;
                const bool in_tab = lane < ns;
                const int rk = (my_start - first0 + W * P - 1) >> (A + W - 1);
                const bool claims = in_tab && my_start > first0 && rk <= 63;
                // ds_permute: every lane PUSHES one word to a lane of its choice and, of several pushes to one lane, the
                // highest pushing lane's arrives — the largest k, as wanted.  Lanes without a claim push a zero to lane 0,
                // which no segment claims (rk >= 1).  No LDS memory, no atomics, no fence.
                const int slot = __builtin_amdgcn_ds_permute((claims ? rk : 0) << 2, claims ? lane + 1 : 0);
                const int base = __popcll(__ballot(in_tab && my_start <= first0));     // segment 0 starts at 0: base >= 1
                const unsigned long long below = __ballot(slot != 0) & ((2ull << lane) - 1ull);
                const int from = below ? 63 - __clzll(below) : lane;
                const int got = __builtin_amdgcn_ds_bpermute(from << 2, slot);
                int sg = below ? got - 1 : base - 1;
                UH_STAMP(12);
                auto from_lane = [&](int v, int src) { return __builtin_amdgcn_ds_bpermute(src << 2, v); };
                um::PhaseSeg cur, nxt;
                cur.start = from_lane(my_start, sg); cur.base = __int_as_float(from_lane(__float_as_int(my_base), sg));
                cur.step = __int_as_float(from_lane(__float_as_int(my_step), sg));
                nxt.start = from_lane(my_start, sg + 1); nxt.base = __int_as_float(from_lane(__float_as_int(my_base), sg + 1));
                nxt.step = __int_as_float(from_lane(__float_as_int(my_step), sg + 1));
                const int n2 = from_lane(my_start, sg + 2);                 // lanes >= ns hold INT_MAX (sg + 2 <= kPhaseCap + 1 < 64)
                // Inside a segment every phase lies on the grid of ONE binade and so does the step (phase_table.h): the
                // neighbour W positions on is value +- W step EXACTLY, a single addition instead of subtract / convert / fma.
                // `cur` holds the lane's first position: its chain runs forward from there (values behind the segment's end
                // are wrong and not selected); `nxt` is entered somewhere inside the run, if at all: its chain runs
                // BACKWARD from the run's last position (values in front of its start are wrong and not selected).
                {
                    float a[P], b[P];
                    const float step_a = cur.step * (float)W, step_b = nxt.step * (float)W;     // exact (W = 1 or 2)
                    a[0] = um::phase_table_eval(cur, i0);
                    b[P - 1] = um::phase_table_eval(nxt, ilast);
#pragma unroll
                    for (int j = 1; j < P; ++j) { a[j] = a[j - 1] + step_a; b[P - 1 - j] = b[P - j] - step_b; }
#pragma unroll
                    for (int j = 0; j < P; ++j) ph[j] = (i0 + W * j < nxt.start) ? a[j] : b[j];
                }
                const bool more = !(ilast < n2 && ilast < covered);
#ifdef UH_MIXFFT_STAMPS
                stamps.t[kStampPhases + 4] = __any(more) ? 2ull : 1ull;
#endif
                if (__builtin_expect(__any(more), 0)) {
                    if (more) {
                        int nstart = nxt.start;
#pragma unroll
                        for (int j = 0; j < P; ++j) {
                            const int i = i0 + W * j;
                            if (i < covered) {
                                while (i >= nstart) { ++sg; cur = seg[sg]; nstart = seg_start[sg + 1]; }
                                ph[j] = um::phase_table_eval(cur, i);
                            }
                        }
                    }
                }
            } else if (ilast >= 0 && i0 < covered) {
                // a round behind a table overflow (more than kPhaseCap segments in one symbol): the search
                const int ifirst = (i0 >= 0) ? i0 : (W == 2 ? (i0 & 1) : 0);
                int cnt = 1;                                 // segments starting at or before ifirst (segment 0 starts at 0)
                for (int k = 1; k < ns; ++k) cnt += (__builtin_amdgcn_readlane(my_start, k) <= ifirst) ? 1 : 0;
                int sg = cnt - 1;
                um::PhaseSeg cur = seg[sg];
                int nstart = seg_start[sg + 1];
#pragma unroll
                for (int j = 0; j < P; ++j) {
                    const int i = i0 + W * j;
                    if (i >= 0 && i < covered) {
                        while (i >= nstart) { ++sg; cur = seg[sg]; nstart = seg_start[sg + 1]; }
                        ph[j] = um::phase_table_eval(cur, i);
                    }
                }
            }
            wave_sync();
            done += covered;
            pcur = pnext;
        }
        cfo_phase = pcur;
    }
    // The next item's samples, CFO, start phase and table entry — requested HERE, behind the last use of this item's
    // request registers (`it` is overwritten in place: no copy at the loop end, which would have to wait for the loads)
    // and three quarters of an item ahead of their use.  Nothing else is loaded in the rest of the item.
    UH_STAMP(13);
    request_next();
    UH_STAMP(2);
    // The rotation factors of the lane's eight phases, handed one by one to `sink(j, factor)` (j a compile-time index).
    // Two sinks: with one or two segments the phases are those of the lane's OWN samples (see above) and the factor is mixed
    // in at once — it never leaves the registers; otherwise the phases are a run of consecutive samples and the factors go
    // through LDS to the lanes that mix them.
    auto for_each_factor = [&](auto sink) {
        if (bounded) {
            // Do all 512 phases of this wavefront-item reduce to the same quadrant?  reduce_fast is monotone in y, so the
            // quadrants of the smallest and of the largest phase settle it —
            // true for most items: a symbol's phases span |2 pi CFO 1024 / fs|, a fraction of a radian.  Every per-lane
            // select of sincosf_bounded_ (sign of the reduced argument, sign of the cosine, swap, the |y| < 2^-12 case) is then a
            // scalar choice made once (um::sincosf_quadrant_: the same fused operations on the same operands) — 17
            // instead of 61 vector instructions per sample.
            static_assert(P == 8, "eight phases per lane");
            float wlo = range_lo, whi = range_hi;
            if (!two_segments) {
                const float plo = fminf(fminf(fminf(ph[0], ph[1]), fminf(ph[2], ph[3])), fminf(fminf(ph[4], ph[5]), fminf(ph[6], ph[7])));
                const float phi = fmaxf(fmaxf(fmaxf(ph[0], ph[1]), fmaxf(ph[2], ph[3])), fmaxf(fmaxf(ph[4], ph[5]), fmaxf(ph[6], ph[7])));
                wlo = wave_fmin(plo); whi = wave_fmax(phi);
            }
            int n_lo, n_hi;
            (void)um::reduce_fast((double)wlo, &n_lo);
            (void)um::reduce_fast((double)whi, &n_hi);
            n_lo = __builtin_amdgcn_readfirstlane(n_lo); n_hi = __builtin_amdgcn_readfirstlane(n_hi);
            bool one_quadrant = n_lo == n_hi && fabsf(wlo) < 100.0f && fabsf(whi) < 100.0f;
            if (one_quadrant && wlo <= 0.0f && whi >= 0.0f) {
                // the one input the quadrant path must not see is -0.0 (pinned_math.h); zeros only occur in a range that
                // contains zero — the first rotating symbol, whose phase starts at +0.0
                bool negzero = false;
#pragma unroll
                for (int j = 0; j < P; ++j) negzero |= __float_as_uint(ph[j]) == 0x80000000u;
                one_quadrant = !__any(negzero);
            }
#ifdef UH_MIXFFT_STAMPS
            stamps.t[kStampPhases + 4] |= one_quadrant ? 16ull : 32ull;
#endif
            if (__builtin_expect(one_quadrant, 1)) {
                double m_n; float y_sign, cos_sign; bool swap;
                um::sincosf_quadrant_setup(n_lo, &m_n, &y_sign, &cos_sign, &swap);
                (void)y_sign; (void)cos_sign; (void)swap;              // compile-time in the four instances below (Q = n & 3)
                auto block = [&](auto q_) {
                    constexpr int Q = decltype(q_)::value;
                    static_for_n<P>([&](auto jc) {
                        constexpr int j = decltype(jc)::value;
                        float sn, cs;
                        um::sincosf_quadrant_q<Q>(ph[j], m_n, &sn, &cs);
                        sink(jc, mk(cs, sn));
                    });
                };
                switch (n_lo & 3) {                                    // wave-uniform
                    case 0: block(std::integral_constant<int, 0>{}); break;
                    case 1: block(std::integral_constant<int, 1>{}); break;
                    case 2: block(std::integral_constant<int, 2>{}); break;
                    default: block(std::integral_constant<int, 3>{}); break;
                }
            } else {
                static_for_n<P>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    float sn, cs;
                    um::sincosf_bounded_(ph[j], &sn, &cs);
                    sink(jc, mk(cs, sn));
                });
            }
        } else {
            static_for_n<P>([&](auto jc) { sink(jc, cexpj(ph[decltype(jc)::value])); });
        }
    };
    // ---- mix: samples[i] * conj(osc) (* rotation) ----
    if (cfo_on && two_segments) {
        c32 rv[P];
        for_each_factor([&](auto jc, c32 r) { rv[decltype(jc)::value] = r; });
        UH_STAMP(3);
#pragma unroll
        for (int qp = 0; qp < P; ++qp) {
            const c32 o = ROT ? sh.osc[ROT ? h : 0][ROT ? 64 * qp + lane : 0] : lc.os[qp];
            const c32 mixed = mk(o.re * xs[qp], (-o.im) * xs[qp]);
            v[bitrev_small<A>(qp)] = cmul(mixed, rv[qp]);
        }
    } else {
        if (cfo_on) {
            for_each_factor([&](auto jc, c32 r) { const int m = P * lane + decltype(jc)::value; rot[m + (m >> A)] = r; });
            wave_sync();
        }
        UH_STAMP(3);
#pragma unroll
        for (int qp = 0; qp < P; ++qp) {
            const c32 o = ROT ? sh.osc[ROT ? h : 0][ROT ? 64 * qp + lane : 0] : lc.os[qp];
            c32 mixed = mk(o.re * xs[qp], (-o.im) * xs[qp]);
            if (cfo_on) { const int m = rl + 64 * qp; mixed = cmul(mixed, rot[m + (m >> A)]); }
            v[bitrev_small<A>(qp)] = mixed;
        }
    }
    if (cfo_on) wave_sync();
    UH_STAMP(4);

    // ---- group A: stages 0..A-1 on the lane's P consecutive (bit-reversed) positions of its half ----
#pragma unroll
    for (int s = 0; s < A; ++s) {
        const int half = 1 << s;
#pragma unroll
        for (int q = 0; q < P; ++q) {
            if (q & half) continue;
            const int j = (q & (half - 1)) << (A - 1 - s);                  // compile-time: the loops are unrolled
            const c32 w = lc.wA[j];                                         // twiddle[k << (LOG2N-1-s)], wave-uniform
            // Ten of the group's twelve butterflies multiply by twiddle[0] = (1, -0) or twiddle[N/4] = (cos(-pi/2), -1)
            // (build_demod_tables checks both): 1 * x and -1 * x are exact, so w * b keeps two products instead of four —
            // (b.re - w.im b.im, b.im + w.im b.re) and (w.re b.re + b.im, w.re b.im - b.re), the SAME values bit for bit
            // (x - (-y) == x + y, x + (-y) == x - y), signed zeros, infinities and NaNs included.
            if (j == 0) {
                const c32 b_ = v[q + half];
                const c32 t_ = mk(b_.re - w.im * b_.im, b_.im + w.im * b_.re);
                v[q + half] = csub(v[q], t_); v[q] = cadd(v[q], t_);
            } else if (j == (P / 4)) {
                const c32 b_ = v[q + half];
                const c32 t_ = mk(w.re * b_.re + b_.im, w.re * b_.im - b_.re);
                v[q + half] = csub(v[q], t_); v[q] = cadd(v[q], t_);
            } else {
                UH_BUTTERFLY(v[q], v[q + half], w);
            }
        }
    }
#pragma unroll
    for (int q = 0; q < P; ++q) { const int i = P * lane + q; X[i + (i >> A)] = v[q]; }
    wave_sync();
    UH_STAMP(5);
    // ---- group B: stages A..2A-1, lane (blk, r) holds X[blk*P*P + r + P*j] ----
    {
        const int blk = lane / P, r = lane % P;
#pragma unroll
        for (int j = 0; j < P; ++j) { const int i = blk * P * P + r + P * j; v[j] = X[i + (i >> A)]; }
#pragma unroll
        for (int s = A; s < 2 * A; ++s) {
            const int hj = 1 << (s - A);
#pragma unroll
            for (int j = 0; j < P; ++j) {
                if (j & hj) continue;
                const int k = r + P * (j & (hj - 1));
                const c32 w = sh.twB[P * (hj - 1) + k];
                UH_BUTTERFLY(v[j], v[j + hj], w);
            }
        }
#pragma unroll
        for (int j = 0; j < P; ++j) { const int i = blk * P * P + r + P * j; X[i + (i >> A)] = v[j]; }
    }
    wave_sync();
    UH_STAMP(6);
    // ---- group C: stages 2A..3A-1 = LOG2N-2, lane holds X[lane + 64*t]; only t = 0 (elements 0..63 of the half) and
    //      t = P-1 (elements M-64..M-1) feed bins that are used, so only their ancestors are computed:
    //      stage 6 pairs (t, t+1) with w6 = twiddle[lane << 3]; stage 7 pairs (t, t+2) with twiddle[(lane + 64 (t&1)) << 2];
    //      stage 8 pairs (0, 4) and (3, 7) with twiddle[(lane + 64 (t&3)) << 1], t = 0 and 3 ----
    {
#pragma unroll
        for (int t = 0; t < P; ++t) { const int i = lane + 64 * t; v[t] = X[i + (i >> A)]; }
#pragma unroll
        for (int t = 0; t < P; t += 2) UH_BUTTERFLY(v[t], v[t + 1], lc.w6);
        UH_BUTTERFLY(v[0], v[2], lc.w7[0]);
        UH_BUTTERFLY(v[4], v[6], lc.w7[0]);
        UH_BUTTERFLY(v[1], v[3], lc.w7[1]);
        UH_BUTTERFLY(v[5], v[7], lc.w7[1]);
        UH_BUTTERFLY(v[0], v[4], lc.w8[0]);
        UH_BUTTERFLY(v[3], v[7], lc.w8[1]);
    }
    UH_STAMP(7);
    if constexpr (W == 2) {
        // ---- last stage (LOG2N-1): element k of the even half with element k of the odd half, w = twiddle[k] ----
        sh.xch[par][h][lane] = h ? v[0] : v[P - 1];          // what the partner needs: O_lo from 1, E_hi from 0
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        UH_STAMP(8);
        const c32 other = sh.xch[par][1 - h][lane];
        c32 a = h ? other : v[0];                             // even half's element
        c32 b = h ? v[P - 1] : other;                         // odd half's element
        UH_BUTTERFLY(a, b, lc.w_last);
        bin_out = h ? b : a;                                  // bin `lane` (a + t) / N-64+lane (a - t): stored with the next request
        bin_hi = bin_out;
    } else {
        UH_STAMP(8);
        bin_out = v[0];                                       // the transform is complete: bins `lane` and N - 64 + lane
        bin_hi = v[P - 1];
    }
    wave_sync();
    UH_STAMP(9);
}

// interpolateChannel (channel_equalizer.cpp:601-631): one lane per table entry
__device__ __forceinline__ void interpolate_channel(TrackShared& sh, const DemodConst& D, const LaneConst& lc) {
    if ((int)threadIdx.x < D.n_interp) {
        const int lo = lc.i_lo, hi = lc.i_hi, dst = lc.i_dst;
        if (lo >= 0 && hi >= 0) {
            const c32 H1 = sh.H[lo], H2 = sh.H[hi];
            const c32 pd = cmul(H2, cconj(H1));
            // std::abs(std::arg(H2 * conj(H1))) > 1.5708f: decided from the signs except in a sliver (pinned_math.h)
            if (um::atan2f_beyond_right_angle(pd.im, pd.re)) sh.H[dst] = (lc.i_alpha < 0.5f) ? H1 : H2;
            else sh.H[dst] = cadd(cscale(H1, 1.0f - lc.i_alpha), cscale(H2, lc.i_alpha));
        } else if (lo >= 0) {
            sh.H[dst] = sh.H[lo];
        } else if (hi >= 0) {
            sh.H[dst] = sh.H[hi];
        }
    }
}

// The carrier half of updateChannelEstimate when the pilot half ran in track_pilot_kernel: interpolation
// between the (already updated, already de-rotated) pilots and the timing phase re-applied to every carrier
// (:514-567)
__device__ __forceinline__ void finish_channel_estimate(TrackShared& sh, const DemodConst& D, const LaneConst& lc,
                                                        const Track& tr) {
    const int lane = threadIdx.x;
    const bool fix = !D.differential && fabsf(tr.timing) > 0.1f;
    interpolate_channel(sh, D, lc);
    wave_sync();
    if (fix && lane < D.n_carriers) {
        const float tp = timing_phase_of(lc.slot_k, tr.timing, D.log2_fft);
        sh.H[lane] = cmul(sh.H[lane], cexpj_bounded(tp));
    }
    wave_sync();
}

// The pilot half of updateChannelEstimate (channel_equalizer.cpp:330-513,583-592) for 64 / G frames per
// wavefront: G lanes per frame (G >= number of pilots), lane `sub` of a group owns pilot `sub`, and every lane
// carries its frame's tracker scalars.  Everything here is work on <= G pilots or on per-frame scalars — in the
// one-frame-per-wavefront layout it kept 15 of 64 lanes busy for half of the instructions of a symbol.  Every
// serial sum of the reference keeps its order; the record in HBM carries the result to track_kernel, which
// interpolates, equalises and demaps.
template <int G, bool FRESH>
__global__ __launch_bounds__(kWave, 5) void track_pilot_kernel(const DemodConst* __restrict__ Dp, int n_frames,
                                                               float* __restrict__ state, const c32* __restrict__ fq_all,
                                                               float* __restrict__ trk_rec) {
    constexpr bool fresh = FRESH;
    // FRESH: the frame's first symbol on a demodulator as its constructor leaves it and without initial offsets (launch_demod:
    // no init_state_kernel ran) — the record is not read, every field of it is written.
    constexpr int FPW = kWave / G;
    constexpr int kRow = G * 8 + 8;                            // floats per group: 8 terms per pilot, padded (banks)
    __shared__ __attribute__((aligned(16))) float s_terms[FPW][kRow];   // also the staging of the carrier-phase sum
    __shared__ __attribute__((aligned(16))) float s_sums[FPW][8];
    const DemodConst& D = *Dp;
    const int lane = threadIdx.x, sub = lane % G, grp = lane / G;
    const int np = D.n_pilot;
    const bool is_pilot = sub < np;
    auto fq_of = [&](int bin) { return fq_index(D, bin); };
    const int ps = is_pilot ? D.pilot_slot[sub] : 0;
    const int pilot_fq = fq_of(D.bin[ps]), pilot_k = D.k_of[ps];
    const c32 pilot_seq = is_pilot ? D.pilot_seq[sub] : mk(1.0f, 0.0f);
    const unsigned long long gmask = (G == 64) ? ~0ull : ((1ull << G) - 1ull);
    for (int base = blockIdx.x * FPW; base < n_frames; base += gridDim.x * FPW) {
        const bool act = base + grp < n_frames;
        const int frame = act ? base + grp : n_frames - 1;     // idle groups shadow the last frame, stores masked
        float* st = state + (size_t)frame * kStFloats;
        const c32* fq = fq_all + (size_t)frame * (2 * D.fq_half);
        Track tr;
        const bool compact = compact_pilot_state(D);
        c32 h_old = mk(0.0f, 0.0f), prev = mk(0.0f, 0.0f);
        if (fresh) {                                            // init_state_kernel's values (demodulator.cpp:26-43)
            tr.freq_offset_hz = 0.0f; tr.freq_offset_filtered = 0.0f; tr.cfo_phase = 0.0f;
            tr.noise_variance = 0.1f; tr.snr_linear = 1.0f; tr.timing = 0.0f;
            tr.ppc = mk(1.0f, 0.0f); tr.cpc = mk(1.0f, 0.0f);
            tr.cpc_init = 0; tr.has_prev = 0; tr.has_dprev = 0; tr.snr_symbol_count = 0; tr.symbols_since_sync = 0;
            if (is_pilot) h_old = mk(1.0f, 0.0f);
        } else {
            tr.freq_offset_hz = st[st_cfo]; tr.freq_offset_filtered = st[st_cfo_filt]; tr.cfo_phase = st[st_cfo_phase];
            tr.noise_variance = st[st_noise]; tr.snr_linear = st[st_snr]; tr.timing = st[st_timing];
            tr.ppc = mk(st[st_ppc_re], st[st_ppc_im]); tr.cpc = mk(st[st_cpc_re], st[st_cpc_im]);
            const int flags = (int)st[st_flags];
            tr.cpc_init = flags & 1; tr.has_prev = (flags >> 1) & 1; tr.has_dprev = (flags >> 2) & 1;
            tr.snr_symbol_count = (int)st[st_count]; tr.symbols_since_sync = (int)st[st_since];
            if (is_pilot) {
                h_old = compact ? reinterpret_cast<const c32*>(st + kStHp)[sub] : reinterpret_cast<const c32*>(st + kStH)[ps];
                prev = reinterpret_cast<const c32*>(st + kStPrev)[sub];
            }
        }
        const float alpha = (tr.snr_symbol_count == 0) ? 1.0f : 0.9f;

        c32 h = mk(0.0f, 0.0f);
        if (is_pilot) h = cdiv_pilot(fq[pilot_fq], pilot_seq);
        if (np != 0 && __any(!tr.cpc_init)) {                  // carrier phase recovery (:348-357)
            c32* cb = reinterpret_cast<c32*>(&s_terms[grp][0]);
            wave_sync();
            cb[sub] = h;
            wave_sync();
            c32 h_sum = mk(0.0f, 0.0f);
            for (int i = 0; i < np; ++i) h_sum = cadd(h_sum, cb[i]);
            if (!tr.cpc_init) {
                const c32 h_avg = cdivf(h_sum, (float)np);
                const float avg_mag = cabs_(h_avg);
                if (avg_mag > 0.01f) { tr.cpc = cdivf(cconj(h_avg), avg_mag); tr.cpc_init = 1; }
            }
        }
        h = cmul(h, tr.cpc);
        const float n2 = cnorm(h);

        float nd = 0.0f, ph = 0.0f;
        c32 unit = mk(0.0f, 0.0f);
        bool f_noise = false, f_cfo = false, f_tim = false;
        c32 h_new = h_old;
        if (is_pilot) {
            if (tr.has_prev) {
                const float pn = cnorm(prev);
                if (pn > 1e-6f && n2 > 1e-6f) {
                    nd = cnorm(csub(h, prev));
                    f_noise = true;
                    const c32 diff = cmul(h, cconj(prev));
                    const float mag = cabs_(diff);
                    if (mag > 1e-6f) { unit = cdivf(diff, mag); f_cfo = true; }
                }
            }
            if (tr.snr_symbol_count >= 3 && !(n2 < 1e-6f)) { ph = carg_(h); f_tim = true; }
            h_new = cadd(cscale(h, alpha), cscale(h_old, 1.0f - alpha));
        }
        const int shift = grp * G;
        const int noise_hits = __popcll((__ballot(f_noise) >> shift) & gmask),
                  cfo_hits = __popcll((__ballot(f_cfo) >> shift) & gmask),
                  tim_hits = __popcll((__ballot(f_tim) >> shift) & gmask);
        wave_sync();
        if (is_pilot) {
            const float kf = (float)pilot_k;
            float4 ta, tb;
            ta.x = n2;
            ta.y = f_noise ? nd : -0.0f;
            ta.z = f_cfo ? unit.re : -0.0f;
            ta.w = f_cfo ? unit.im : -0.0f;
            tb.x = f_tim ? kf : -0.0f;
            tb.y = f_tim ? (float)(pilot_k * pilot_k) : -0.0f;
            tb.z = f_tim ? ph : -0.0f;
            tb.w = f_tim ? kf * ph : -0.0f;
            *reinterpret_cast<float4*>(&s_terms[grp][sub * 8]) = ta;
            *reinterpret_cast<float4*>(&s_terms[grp][sub * 8 + 4]) = tb;
        }
        wave_sync();
        if (sub < 8) {                                         // the eight serial sums, one lane each, pilot order
            float acc = 0.0f;
            for (int i = 0; i < np; ++i) acc += s_terms[grp][i * 8 + sub];
            s_sums[grp][sub] = acc;
        }
        wave_sync();
        const float4 sa = *reinterpret_cast<const float4*>(&s_sums[grp][0]);
        const float4 sb = *reinterpret_cast<const float4*>(&s_sums[grp][4]);
        const float s_sig = sa.x, s_nd = sa.y, s_ur = sa.z, s_ui = sa.w, sum_k = sb.x, sum_k2 = sb.y, sum_phase = sb.z,
                    sum_k_phase = sb.w;
        const float signal_power = s_sig / (float)np;                   // NaN when np == 0 (reference quirk)
        int noise_count = noise_hits;
        float noise_power_sum = s_nd;
        if (noise_count == 0) { noise_power_sum = signal_power / 31.6f; noise_count = 1; }

        if (tr.has_prev && np != 0) {                               // CFO from pilot phase differences
            const int valid = cfo_hits;
            if (valid > 0) {
                const c32 avg = cdivf(mk(s_ur, s_ui), (float)valid);
                const float apd = um::atan2f_(avg.im, avg.re);
                tr.ppc = cexpj_bounded(-apd);
                const float residual = (float)((double)apd / D.two_pi_symbol_duration);
                const float total = tr.freq_offset_hz + residual;
                float a = 0.3f;
                if (tr.symbols_since_sync < 10) {
                    const float progress = (float)tr.symbols_since_sync / 10;
                    a = 0.9f * (1.0f - progress) + 0.3f * progress;
                }
                if (fabsf(residual) > 10.0f) a = fmax_std(a, 0.9f);
                tr.symbols_since_sync++;
                tr.freq_offset_filtered = a * total + (1.0f - a) * tr.freq_offset_filtered;
                tr.freq_offset_hz = fmax_std(-90.0f, fmin_std(90.0f, tr.freq_offset_filtered));
            }
        } else {
            tr.ppc = mk(1.0f, 0.0f);
        }

        if (tr.snr_symbol_count >= 3) {                             // timing from the pilot phase slope
            const int tv = tim_hits;
            if (tv >= 3) {
                const float n = (float)tv;
                const float denom = n * sum_k2 - sum_k * sum_k;
                if (fabsf(denom) > 1e-6f) {
                    const float slope = (n * sum_k_phase - sum_k * sum_phase) / denom;
                    const float inst = (float)((double)(slope * D.fft_f) / kTwoPi);
                    tr.timing = 0.3f * inst + (1.0f - 0.3f) * tr.timing;
                    tr.timing = fmax_std(-D.max_timing, fmin_std(D.max_timing, tr.timing));
                }
            }
        }
        tr.has_prev = (np != 0);

        // coherent timing fix: the pilots are de-rotated before the interpolation (:514-530)
        const bool fix = !D.differential && fabsf(tr.timing) > 0.1f;
        if (fix && is_pilot) h_new = cmul(h_new, cexpj_bounded(-timing_phase_of(pilot_k, tr.timing, D.log2_fft)));
        // deferred carrier half (track_all_kernel): it gets the de-rotated estimate in its own record, and the rotation back
        // that it would have applied to the pilots after the interpolation (:561-567) happens here
        const c32 h_derot = h_new;
        if (trk_rec && fix && is_pilot) h_new = cmul(h_new, cexpj_bounded(timing_phase_of(pilot_k, tr.timing, D.log2_fft)));
        if (noise_count > 1 && noise_power_sum > 0.0f) {            // (:583-592)
            float nv = noise_power_sum / (float)(noise_count - 1);
            if (nv < 1e-6f) nv = 1e-6f;
            tr.noise_variance = nv;
            float inst_snr = signal_power / nv;
            inst_snr = fmax_std(0.1f, fmin_std(10000.0f, inst_snr));
            tr.snr_linear = 0.3f * inst_snr + (1.0f - 0.3f) * tr.snr_linear;
        }
        tr.snr_symbol_count++;

        if (act) {
            // (whole 128-byte lines of Hp and of prev_pilot_phases: lanes behind the last pilot write zeros nobody reads)
            if (compact && sub < ((np + 15) & ~15)) reinterpret_cast<c32*>(st + kStHp)[sub] = is_pilot ? h_new : mk(0.0f, 0.0f);
            if (compact && !is_pilot && sub < ((np + 15) & ~15)) reinterpret_cast<c32*>(st + kStPrev)[sub] = mk(0.0f, 0.0f);
            if (is_pilot) {
                if (!compact) reinterpret_cast<c32*>(st + kStH)[ps] = h_new;
                reinterpret_cast<c32*>(st + kStPrev)[sub] = h;
            }
            // Every lane of the group carries the frame's scalars: lane `sub` stores scalar `sub` — one store
            // instruction per wavefront writing 52 contiguous bytes per record, where lane 0 storing all twelve was
            // twelve instructions of four 4-byte writes each.  st_cfo_phase belongs to cfo_walk_kernel.
            static_assert(st_since == 12 && st_cfo_phase == 2 && G >= 13, "scalar block layout");
            // (the lane index made opaque HERE: left to itself the compiler computes the seventeen lane masks `sub == k` of
            // the two select chains below once per workgroup, holds them in 34 scalar registers it does not have, spills them
            // into VGPR lanes and reads every one back with two v_readlane per group of frames — a compare is cheaper)
            int subw = sub;
            asm volatile("" : "+v"(subw));
            float mine = tr.freq_offset_hz;
            mine = (subw == st_cfo_filt) ? tr.freq_offset_filtered : mine;
            mine = (subw == st_noise) ? tr.noise_variance : mine;
            mine = (subw == st_snr) ? tr.snr_linear : mine;
            mine = (subw == st_timing) ? tr.timing : mine;
            mine = (subw == st_ppc_re) ? tr.ppc.re : mine;
            mine = (subw == st_ppc_im) ? tr.ppc.im : mine;
            mine = (subw == st_cpc_re) ? tr.cpc.re : mine;
            mine = (subw == st_cpc_im) ? tr.cpc.im : mine;
            mine = (subw == st_flags) ? (float)(tr.cpc_init | (tr.has_prev << 1) | (tr.has_dprev << 2)) : mine;
            mine = (subw == st_count) ? (float)tr.snr_symbol_count : mine;
            mine = (subw == st_since) ? (float)tr.symbols_since_sync : mine;
            mine = (subw == st_cfo_phase) ? tr.cfo_phase : mine;    // cfo_walk_kernel's, but for the record's first write
            if (sub <= st_since && (fresh || sub != st_cfo_phase)) st[sub] = mine;
            if (trk_rec) {
                float* rec = trk_rec + (size_t)frame * trk_rec_floats(np);
                if (is_pilot) reinterpret_cast<c32*>(rec + trk_rec_hp(np))[sub] = h_derot;
                float v = tr.noise_variance;                    // lane sub stores scalar sub of the record's tail
                v = (subw == tk_timing - tk_noise) ? tr.timing : v;
                v = (subw == tk_cfo - tk_noise) ? tr.freq_offset_hz : v;
                v = (subw == tk_snr - tk_noise) ? tr.snr_linear : v;
                v = (subw == tk_phase - tk_noise) ? tr.cfo_phase : v;
                v = (subw == tk_count - tk_noise) ? (float)tr.snr_symbol_count : v;
                if (sub < ((np <= kTrkRecCompactPilots) ? 2 : 8)) rec[tk_noise + sub] = (sub <= tk_count - tk_noise) ? v : 0.0f;
            }
        }
    }
}

// one carrier's LLRs (soft_demap.hpp), stored to out[0..bits)
// sym_abs / prev_abs: cabs_(sym) and cabs_(prev) where the caller has them already (the differential demappers multiply
// them: the reference calls std::abs on both, the previous symbol's |sym| is this symbol's |prev|); negative = not known
template <int MOD>
__device__ __forceinline__ void demap_carrier(c32 sym, c32 prev, float nv, float* __restrict__ out, float sym_abs = -1.0f,
                                              float prev_abs = -1.0f) {
    const bool known = sym_abs >= 0.0f && prev_abs >= 0.0f;
    switch (MOD) {
        case ULTRA_MOD_DBPSK: {
            const c32 diff = cmul(sym, cconj(prev));
            const float pd = um::atan2f_(diff.im, diff.re);
            const float sp = known ? sym_abs * prev_abs : cabs_(sym) * cabs_(prev);
            out[0] = (sp < 1e-6f) ? 0.0f : clip_llr(2.0f * sp * um::cosf_(pd) / nv);
            break;
        }
        case ULTRA_MOD_DQPSK: {
            const c32 diff = cmul(sym, cconj(prev));
            const float phase = um::atan2f_(diff.im, diff.re);
            const float sp = known ? sym_abs * prev_abs : cabs_(sym) * cabs_(prev);
            if (sp < 1e-6f) { out[0] = 0.0f; out[1] = 0.0f; break; }
            const float scale = 2.0f * sp / nv;
            const float pi = 3.14159265358979f;
            out[0] = clip_llr(scale * um::sinf_(phase + pi / 4));
            out[1] = clip_llr(scale * um::cosf_(2 * phase));
            break;
        }
        case ULTRA_MOD_D8PSK: {
            const c32 diff = cmul(sym, cconj(prev));
            const float pd = um::atan2f_(diff.im, diff.re);
            const float sp = known ? sym_abs * prev_abs : cabs_(sym) * cabs_(prev);
            if (sp < 1e-6f) { out[0] = 0.0f; out[1] = 0.0f; out[2] = 0.0f; break; }
            const float conf = sp / nv;
            out[0] = clip_llr(conf * um::sinf_(pd));
            out[1] = clip_llr(conf * um::sinf_(2.0f * pd));
            out[2] = clip_llr(conf * um::sinf_(4.0f * pd));
            break;
        }
        case ULTRA_MOD_BPSK:
            out[0] = clip_llr(-2.0f * sym.re / nv);
            break;
        case ULTRA_MOD_QAM16: {
            const float I = sym.re, Q = sym.im, scale = 2.0f / nv;
            out[0] = clip_llr(-scale * I);
            out[1] = clip_llr(scale * (fabsf(I) - 0.6324555320336759f));
            out[2] = clip_llr(-scale * Q);
            out[3] = clip_llr(scale * (fabsf(Q) - 0.6324555320336759f));
            break;
        }
        case ULTRA_MOD_QAM32: {
            const float S = 0.1961161351381840f;
            const float sf = 2.0f / nv;
            for (int b = 0; b < 5; ++b) {
                const int mask = 1 << (4 - b);
                float m0 = 1e10f, m1 = 1e10f;
                for (int qi = 0; qi < 8; ++qi) {
                    const float ql = (float)(2 * qi - 7) * S;       // Q_LEVELS[qi] * QAM32_SCALE
                    const int qg = qi ^ (qi >> 1);                   // Q_GRAY[qi]
                    for (int ii = 0; ii < 4; ++ii) {
                        const float il = (float)(2 * ii - 3) * S;   // I_LEVELS[ii] * QAM32_SCALE
                        const int bits = (qg << 2) | (ii ^ (ii >> 1));
                        const float dr = sym.re - il, di = sym.im - ql;
                        const float dist = dr * dr + di * di;
                        if (bits & mask) { if (dist < m1) m1 = dist; }
                        else { if (dist < m0) m0 = dist; }
                    }
                }
                out[b] = clip_llr(sf * (m1 - m0));
            }
            break;
        }
        case ULTRA_MOD_QAM64: {
            const float I = sym.re, Q = sym.im, scale = 2.0f / nv;
            const float D2 = 0.3086067f, D4 = 0.6172134f;
            out[0] = clip_llr(-scale * I);
            out[1] = clip_llr(scale * (fabsf(I) - D4));
            out[2] = clip_llr(scale * (fabsf(fabsf(I) - D4) - D2));
            out[3] = clip_llr(-scale * Q);
            out[4] = clip_llr(scale * (fabsf(Q) - D4));
            out[5] = clip_llr(scale * (fabsf(fabsf(Q) - D4) - D2));
            break;
        }
        case ULTRA_MOD_QAM256: {
            const float I = sym.re, Q = sym.im, scale = 2.0f / nv;
            const float D2 = 0.1290994f, D4 = 0.2581989f, D8 = 0.5163978f;
            out[0] = clip_llr(-scale * I);
            out[1] = clip_llr(scale * (fabsf(I) - D8));
            out[2] = clip_llr(scale * (fabsf(fabsf(I) - D8) - D4));
            out[3] = clip_llr(scale * (fabsf(fabsf(fabsf(I) - D8) - D4) - D2));
            out[4] = clip_llr(-scale * Q);
            out[5] = clip_llr(scale * (fabsf(Q) - D8));
            out[6] = clip_llr(scale * (fabsf(fabsf(Q) - D8) - D4));
            out[7] = clip_llr(scale * (fabsf(fabsf(fabsf(Q) - D8) - D4) - D2));
            break;
        }
        case ULTRA_MOD_QPSK:
        default: {
            const float scale = (-2.0f * 0.7071067811865476f) / nv;
            out[0] = clip_llr(sym.re * scale);
            out[1] = clip_llr(sym.im * scale);
            break;
        }
    }
}

// Impl::hardDecision (channel_equalizer.cpp:637-700): the slicer of the adaptive equaliser's decisions
__device__ __forceinline__ float slice8(float x, float d) {
    if (x < -6 * d) return -7 * d;
    if (x < -4 * d) return -5 * d;
    if (x < -2 * d) return -3 * d;
    if (x < 0) return -d;
    if (x < 2 * d) return d;
    if (x < 4 * d) return 3 * d;
    if (x < 6 * d) return 5 * d;
    return 7 * d;
}
template <int MOD>
__device__ __forceinline__ c32 hard_decision(c32 sym) {
    if (MOD == ULTRA_MOD_BPSK) return mk(sym.re > 0 ? 1.0f : -1.0f, 0.0f);
    if (MOD == ULTRA_MOD_QAM16) {
        auto slice = [](float x) { return (x < -0.4f) ? -0.9487f : (x < 0.0f) ? -0.3162f : (x < 0.4f) ? 0.3162f : 0.9487f; };
        return mk(slice(sym.re), slice(sym.im));
    }
    if (MOD == ULTRA_MOD_QAM32) {
        const float d = 0.1961161351381840f;                // QAM32_SCALE, demodulator_constants.hpp:90
        const float I = (sym.re < -2 * d) ? -3 * d : (sym.re < 0) ? -d : (sym.re < 2 * d) ? d : 3 * d;
        return mk(I, slice8(sym.im, d));
    }
    if (MOD == ULTRA_MOD_QAM64) return mk(slice8(sym.re, 0.1543f), slice8(sym.im, 0.1543f));
    return mk(sym.re > 0 ? 0.7071f : -0.7071f, sym.im > 0 ? 0.7071f : -0.7071f);      // QPSK and every other value
}
// The adaptive equaliser's per-carrier state, held by the lane of the data carrier (track_kernel)
struct AdaptiveEq { c32 w; float P; };

// equalize (channel_equalizer.cpp:728-840) + demodulateSymbol
// (demodulator.cpp:199-435) for one symbol; dprev = dbpsk_prev_equalized[lane]
template <int MOD>
__device__ __forceinline__ void equalize_demap(TrackShared& sh, const DemodConst& D, const LaneConst& lc, Track& tr,
                                               c32& dprev, const c32* __restrict__ fq, float* __restrict__ llr_sym,
                                               float* dprev_abs = nullptr, const c32* tc_fixed = nullptr,
                                               const c32* received_in = nullptr, AdaptiveEq* ad = nullptr, c32* eq_out = nullptr) {
    // eq_out (nullable; live streams): the symbol's row of equalized data carriers — what demodulateSymbol appends to
    // constellation_symbols (demodulator.cpp:199-208), the GUI's scatter plot
    // ad (coherent layouts, nullable): equalise against lms_weights instead of channel_estimate and update them from the
    // hard decisions (use_adaptive branch, :779-805; lmsUpdate / rlsUpdate :705-722)
    // received_in (nullable): this lane's bin, where the caller has requested it ahead of time (track_all_kernel)
    // dprev_abs (differential layouts, nullable): |dprev| carried from symbol to symbol by the caller (negative: not known)
    // — the previous symbol's |sym| is this symbol's |prev|.  tc_fixed (nullable): the timing-phase factor of this lane
    // where the timing estimate cannot move between symbols (no pilots).  Both only skip repeated evaluations.
    constexpr bool kDiff = (MOD == ULTRA_MOD_DBPSK || MOD == ULTRA_MOD_DQPSK || MOD == ULTRA_MOD_D8PSK);
    constexpr int kBits = (MOD == ULTRA_MOD_DBPSK || MOD == ULTRA_MOD_BPSK) ? 1 : (MOD == ULTRA_MOD_DQPSK || MOD == ULTRA_MOD_QPSK) ? 2
                        : (MOD == ULTRA_MOD_D8PSK) ? 3 : (MOD == ULTRA_MOD_QAM16) ? 4 : (MOD == ULTRA_MOD_QAM32) ? 5
                        : (MOD == ULTRA_MOD_QAM64) ? 6 : 8;
    const int lane = threadIdx.x;
    const int nd = D.n_data;
    const bool is_data = lane < nd;
    c32 eq = mk(0.0f, 0.0f);
    float nv = 100.0f;
    if (kDiff) {
        if (is_data) {
            const c32 received = fq[lc.data_fq], h = sh.H[lc.data_slot];
            const float h_power = cnorm(h);
            const c32 tc = tc_fixed ? *tc_fixed : cexpj_bounded(timing_phase_of(lc.data_k, tr.timing, D.log2_fft));
            if (h_power > 1e-6f) {
                const c32 t = cdivf(cmul(received, cconj(h)), h_power);
                eq = cmul(cmul(t, tr.ppc), tc);
                nv = tr.noise_variance / h_power;
            } else {
                eq = cmul(cmul(received, tr.ppc), tc);
                nv = 100.0f;
            }
            nv = fmax_std(1e-6f, fmin_std(100.0f, nv));
        }
    } else {
        float h_power = 0.0f;
        if (is_data) {
            const c32 received = received_in ? *received_in : fq[lc.data_fq], h = sh.H[lc.data_slot];
            h_power = cnorm(h);
            if (ad) {
                const c32 w = ad->w;
                const float w_power = cnorm(w), denom = w_power + tr.noise_variance;
                if (denom < 1e-10f) {
                    eq = mk(0.0f, 0.0f);
                    nv = 100.0f;
                } else {
                    eq = cdivf(cmul(cconj(w), received), denom);
                    nv = tr.noise_variance / (w_power + 1e-6f);           // not clamped on this branch (:789-791)
                }
                if (D.decision_directed) {
                    const c32 ref = hard_decision<MOD>(eq);
                    const c32 err = csub(received, cmul(w, ref));
                    if (D.adaptive_eq == 2) {
                        const float lambda = D.rls_lambda, P = ad->P, ref_norm = cnorm(ref);
                        const float k = P / (lambda + P * ref_norm);
                        ad->w = cadd(w, cmul(cscale(cconj(ref), k), err));
                        const float Pn = (P - k * ref_norm * P) / lambda;
                        ad->P = fmax_std(0.001f, fmin_std(1000.0f, Pn));    // ADAPTIVE_EQ_P_MIN / _MAX
                    } else {
                        ad->w = cadd(w, cmul(cscale(cconj(ref), D.lms_mu), err));
                    }
                }
            } else {
            const float mmse_denom = h_power + tr.noise_variance;
            if (mmse_denom < 1e-10f) {
                eq = mk(0.0f, 0.0f);
                nv = 100.0f;
            } else {
                eq = cdivf(cmul(cconj(h), received), mmse_denom);
                nv = tr.noise_variance / (h_power + 1e-6f);
                nv = fmax_std(1e-6f, fmin_std(100.0f, nv));
            }
            }
        }
        // Deep-fade soft erasure (:822-837): nv = 100 where |H|^2 < 0.1f * (avg_h_power / nd), avg_h_power the SERIAL float
        // sum of the nd values — 44 dependent additions that every lane would repeat, to decide a comparison that is almost
        // never close.  Screen: T = the same nd non-negative values summed in any order differs from the serial sum S by
        // |S - T| <= 2 gamma_63 T (1 + gamma) < 2^-17 T (both are within gamma_63 = 63 u / (1 - 63 u) of the true sum, no
        // underflow in additions); the reference's threshold fl(0.1f * fl(S / nd)) and thr = fl(T * fl(0.1f / nd)) add
        // four roundings: they differ by less than 2^-16 relatively, so the threshold lies strictly inside
        // (thr (1 - 2^-14), thr (1 + 2^-14)) even with those two products rounded.  A carrier outside that band is decided
        // by the screen exactly as the reference decides it; if any carrier of the item is inside it (or T is not a
        // comfortable normal number: zero, denormal, infinite, NaN) the serial sum settles the whole item.
        const float T = wave_sum_unordered(is_data ? h_power : 0.0f);
        const float thr = T * D.fade_k, lo = thr * 0.99993896484375f, hi = thr * 1.00006103515625f;    // 1 -+ 2^-14
        const bool t_plain = T >= 0x1p-60f && T <= 0x1p60f;                                        // wave-uniform
        const bool close = is_data && !(h_power < lo) && !(h_power >= hi);
        if (__builtin_expect(t_plain && !__any(close), 1)) {
            if (is_data && h_power < lo) nv = 100.0f;
        } else {
            float avg = ordered_sum(sh.fbuf, h_power, nd);
            avg /= (float)nd;
            if (is_data && h_power < 0.1f * avg) nv = 100.0f;
        }
    }

    if (eq_out && is_data) eq_out[lane] = eq;
    if (kDiff && !tr.has_dprev) { dprev = mk(1.0f, 0.0f); if (dprev_abs) *dprev_abs = -1.0f; }   // (1,0) reference on every path
    float eq_abs = -1.0f;
    if (is_data) {
        if (kDiff) {
            eq_abs = cabs_(eq);
            const float pa = (dprev_abs && *dprev_abs >= 0.0f) ? *dprev_abs : cabs_(dprev);
            demap_carrier<MOD>(eq, dprev, nv * D.ce_margin, llr_sym + lane * kBits, eq_abs, pa);
            dprev = eq;
            if (dprev_abs) *dprev_abs = eq_abs;
        } else {
            demap_carrier<MOD>(eq, dprev, nv * D.ce_margin, llr_sym + lane * kBits);
        }
    }
    if ((MOD == ULTRA_MOD_DQPSK || MOD == ULTRA_MOD_D8PSK) && tr.snr_symbol_count >= 1) {
        // Decision-directed block (demodulator.cpp:362-434).  dbpsk_prev_equalized[i] was just
        // overwritten with equalized[i], so diff = eq * conj(eq) has phase +0 exactly: quadrant 0,
        // phase_error 0, phase_correction = (cos(-0), sin(-0)) = (1, -0); phase_error_sum =
        // (sum of signal_power, +0) -> correction (1, -0), pow(|correction|, a) = 1, rotation
        // (1, -0).  The multiplies are kept literally (they only touch the sign of exact zeros);
        // the oracle runs the block with the libm calls and the parity tests compare.
        bool strong = false;
        if (is_data) {
            const float a = eq_abs;                            // cabs_(eq), taken above
            strong = (a * a) > 0.1f;
            if (strong) sh.H[lc.data_slot] = cmul(sh.H[lc.data_slot], mk(1.0f, -0.0f));
        }
        const int valid = __popcll(__ballot(strong));
        if (valid >= 5) {
            c32 p = cmul(cscale(tr.ppc, 1.0f), mk(1.0f, -0.0f));
            const float mag = cabs_(p);
            if (mag > 0.01f) p = cdivf(p, mag);
            tr.ppc = p;
        }
    }
    if (kDiff) tr.has_dprev = 1;
    wave_sync();
}

// estimateChannelFromLTS (channel_equalizer.cpp:77-328), one training symbol
__device__ __forceinline__ void lts_symbol(TrackShared& sh, const DemodConst& D, const LaneConst& lc, int sym,
                                           int n_train, c32& lts_acc, const c32* __restrict__ fq) {
    const int lane = threadIdx.x;
    if (lane < D.n_data && sym == n_train - 1) {
        c32 h = mk(0.0f, 0.0f);
        if (cabs_(lc.zc) > 0.01f) h = cdiv(fq[lc.data_fq], lc.zc);
        sh.H[lc.data_slot] = h;                                   // last training symbol's estimate
    }
    if (lane < D.n_pilot) {
        if (sym == 0) lts_acc = mk(0.0f, 0.0f);
        if (cabs_(lc.pilot_seq) > 0.01f) lts_acc = cadd(lts_acc, cdiv_pilot(fq[lc.pilot_fq], lc.pilot_seq));
    }
    wave_sync();
}

__device__ __forceinline__ void lts_finish(TrackShared& sh, const DemodConst& D, const LaneConst& lc, Track& tr,
                                           int n_train, c32 lts_acc) {
    const int lane = threadIdx.x;
    const float inv_count = 1.0f / (float)n_train;
    if (lane < D.n_pilot) sh.H[lc.pilot_slot] = cscale(lts_acc, inv_count);
    wave_sync();
    float mag = 0.0f;
    if (lane < D.n_data) mag = cabs_(sh.H[lc.data_slot]);
    const float h_mag_avg = ordered_sum(sh.fbuf, mag, D.n_data) / (float)D.n_data;
    if (h_mag_avg > 1e-6f && tr.noise_variance > 1e-10f) {
        const float s = (h_mag_avg * h_mag_avg) / tr.noise_variance;
        tr.snr_linear = fmax_std(0.1f, fmin_std(10000.0f, s));
    }
    tr.snr_symbol_count = n_train;
    wave_sync();
}

// ---------------------------------------------------------------------------
// Kernels.  All three use one 64-lane workgroup per frame with a grid-stride loop over frames.
//   state   [n_frames][kStFloats] f32 workspace records
//   fq      [n_frames][2 * fq_half] c32 workspace: the used FFT bins of the symbol in flight

// Fresh demodulator per frame (demodulator.cpp:26-43 + SYNCED transition :533-591, or the reset
// block of processPresynced :868-905).
// `timing` (nullable): Impl::timing_offset_samples the frame starts with — the one tracker value neither
// OFDMDemodulator::reset (:987-1017), nor the reset block of processPresynced, nor the mid-frame preamble (:626-655) clears
// (ultra_hip_demod_stream_start, ULTRA_STREAM_START_TIMING).
__global__ __launch_bounds__(kWave) void init_state_kernel(const float* __restrict__ cfo_hz,
                                                           const float* __restrict__ cfo_phase, int n_frames,
                                                           float* __restrict__ state, int compact, int adaptive,
                                                           const float* __restrict__ timing = nullptr) {
    const int lane = threadIdx.x;
    for (int frame = blockIdx.x; frame < n_frames; frame += gridDim.x) {
        float* st = state + (size_t)frame * kStFloats;
        if (adaptive) {                                     // lms_weights = (1, 0), rls_P = 1 (demodulator.cpp:35-38,894-897)
            reinterpret_cast<c32*>(st + kStLms)[lane] = mk(1.0f, 0.0f);
            st[kStRlsP + lane] = 1.0f;
        }
        // channel_estimate = (1, 0) everywhere: the pilots' line for the compact layouts (the training kernel, where there
        // is one, works on the full array and fills both), the whole array otherwise
        if (!compact) reinterpret_cast<c32*>(st + kStH)[lane] = mk(1.0f, 0.0f);
        if (lane < 32) reinterpret_cast<c32*>(st + kStHp)[lane] = mk(1.0f, 0.0f);
        if (lane < 16) {
            float v = 0.0f;
            const float cfo = cfo_hz ? cfo_hz[frame] : 0.0f;
            if (lane == st_cfo || lane == st_cfo_filt) v = cfo;
            else if (lane == st_cfo_phase) v = cfo_phase ? cfo_phase[frame] : 0.0f;
            else if (lane == st_noise) v = 0.1f;
            else if (lane == st_snr) v = 1.0f;
            else if (lane == st_timing) v = timing ? timing[frame] : 0.0f;
            else if (lane == st_ppc_re || lane == st_cpc_re) v = 1.0f;
            st[lane] = v;
        }
    }
}

// SEARCHING -> SYNCED on a demodulator that has demodulated frames before and was not reset() in between
// (demodulator.cpp:533-591; the legacy Modem never resets: src/modem/modem.cpp:153-166): the transition sets the
// frequency offset (and its filter) to the coarse estimate, the correction phase, symbols_since_sync and
// timing_offset_samples to 0, restarts the oscillator (the symbol index does that here), forgets the differential
// reference and — unless the layout is differential without pilots — the carrier phase correction.  Everything else the
// tracker holds is carried into the new frame: channel_estimate, noise_variance, estimated_snr_linear, snr_symbol_count,
// prev_pilot_phases, pilot_phase_correction, the adaptive equaliser's weights.
// (cfo_phase: OFDMDemodulator::setFrequencyOffsetWithPhase between the transition and the first symbol; NULL = 0)
__global__ __launch_bounds__(kWave) void resync_state_kernel(const float* __restrict__ cfo_hz, const float* __restrict__ cfo_phase,
                                                             int n_frames, float* __restrict__ state, int reset_carrier_phase) {
    const int lane = threadIdx.x;
    for (int frame = blockIdx.x; frame < n_frames; frame += gridDim.x) {
        float* st = state + (size_t)frame * kStFloats;
        if (lane < 16) {
            float v = st[lane];
            const float cfo = cfo_hz ? cfo_hz[frame] : 0.0f;
            if (lane == st_cfo || lane == st_cfo_filt) v = cfo;
            else if (lane == st_cfo_phase) v = cfo_phase ? cfo_phase[frame] : 0.0f;
            else if (lane == st_timing || lane == st_since) v = 0.0f;
            else if (lane == st_flags) {
                int flags = (int)v & ~4;                        // dbpsk_prev_equalized.clear()
                if (reset_carrier_phase) flags &= ~1;           // carrier_phase_initialized = false
                v = (float)flags;
            } else if (reset_carrier_phase && lane == st_cpc_re) v = 1.0f;
            else if (reset_carrier_phase && lane == st_cpc_im) v = 0.0f;
            st[lane] = v;
        }
    }
}

// OFDMDemodulator::Impl::estimateCFOFromTraining (src/ofdm/ofdm_sync.cpp:278-380, called with coarse_cfo = 0 from
// processPresynced when the frequency offset was never set: demodulator.cpp:920-925) for the frames whose initial CFO is
// NaN ("never set", ultra_hip.h): the first two training symbols times conj(NCO) — the oscillator table starts at phase 0
// like the estimator's fresh NCO —, then P = sum conj(z1) z2, E1, E2 over the FFT parts, summed serially in the
// reference's order: one LANE per frame (a rare entry; the frame's samples are read through the cache).
__global__ __launch_bounds__(256) void train_cfo_kernel(const DemodConst* __restrict__ Dp, const c32* __restrict__ nco,
                                                        const float* __restrict__ audio, size_t frame_stride,
                                                        const unsigned* __restrict__ frame_offset, const float* __restrict__ cfo_in,
                                                        int n_frames, float* __restrict__ state) {
    const DemodConst& D = *Dp;
    const int frame = blockIdx.x * blockDim.x + threadIdx.x;
    if (frame >= n_frames || !cfo_in) return;
    const float given = cfo_in[frame];
    if (given == given) return;                                        // a preset CFO is trusted (:918-919)
    float* st = state + (size_t)frame * kStFloats;
    float cfo = 0.0f;
    if (D.n_train >= 2) {
        const float* x = audio + (size_t)frame * frame_stride + (frame_offset ? frame_offset[frame] : 0u);
        const int N = D.fft, cp = D.cp, S = D.sym_len;
        c32 P = mk(0.0f, 0.0f);
        float E1 = 0.0f, E2 = 0.0f;
        for (int i = 0; i < N; ++i) {
            const c32 o1 = nco[cp + i], o2 = nco[S + cp + i];
            const float a1 = x[cp + i], a2 = x[S + cp + i];
            const c32 z1 = mk(a1 * o1.re, a1 * -o1.im), z2 = mk(a2 * o2.re, a2 * -o2.im);   // float * conj(osc)
            const c32 pr = cmul(mk(z1.re, -z1.im), z2);
            P = mk(P.re + pr.re, P.im + pr.im);
            E1 += z1.re * z1.re + z1.im * z1.im;
            E2 += z2.re * z2.re + z2.im * z2.im;
        }
        const float corr = um::hypotf_(P.re, P.im) / sqrtf(E1 * E2 + 1e-10f);
        if (!(corr < 0.3f)) {
            const float phase = um::atan2f_(P.im, P.re);
            cfo = (float)((double)(phase * D.sample_rate) / ((double)2.0f * 3.14159265358979323846 * (double)S));
            const float max_cfo = D.sample_rate / (2.0f * (float)S);
            cfo = fmaxf(-max_cfo, fminf(max_cfo, cfo));               // std::max(-m, std::min(m, x)); a NaN x ends at +m in both
        }
    }
    st[st_cfo] = cfo; st[st_cfo_filt] = cfo; st[st_cfo_phase] = 0.0f;
}

// toBaseband + extractSymbol/FFT of symbol `sym` of every frame.
// The serial part of the CFO rotation — the exact jump table of the float phase recurrence for the coming
// symbol (phase_table.h) — is scalar work per frame: here one LANE per frame (in mix_fft_kernel the whole
// wavefront of a frame would wait for it: 3.8 k cycles median, 19 k at the 90th percentile of a 27 k-cycle
// frame).  Reads the tracker's CFO and the phase the previous symbol ended on, writes the frame's table.
// Phase table of one frame's coming symbol (layout: kSegTabWords above) from the tracker's CFO and the phase the symbol
// starts with; returns the phase it ends with.
__device__ __forceinline__ float walk_to_table(const DemodConst& D, float cfo, float phase, unsigned* __restrict__ tab) {
    tab[3] = __float_as_uint(phase);
    tab[1] = __float_as_uint(cfo);
    if (!(fabsf(cfo) > 0.01f)) { tab[0] = 0u; tab[2] = __float_as_uint(phase); return phase; }
    const float inc = (float)(((-kTwoPi) * (double)cfo) / (double)D.sample_rate);
    int covered;
    float pnext;
    const int ns = um::phase_table_walk(phase, inc, D.sym_len, kPhaseCap, &covered, &pnext,
                                        [&](int k, int start, float base, float step) {
                                            tab[4 + 3 * k] = (unsigned)start;
                                            tab[5 + 3 * k] = __float_as_uint(base);
                                            tab[6 + 3 * k] = __float_as_uint(step);
                                        });
    // bit 31: start phase and increment are inside the range of the bounded sincosf (symbol_to_freq2 then has no use for
    // the increment — a double-precision division per item — unless a table overflows)
    const unsigned bounded = (fabsf(phase) <= 4.0f && fabsf(inc) <= 1.0f) ? 0x80000000u : 0u;
    tab[0] = (unsigned)ns | ((unsigned)covered << 8) | bounded; tab[2] = __float_as_uint(pnext);
    // a table overflow (more than kPhaseCap segments in one symbol) leaves the rest to mix_fft; the end phase is
    // walked on here with the same function
    for (int done = covered; done < D.sym_len;) {
        int c2;
        float p2;
        um::phase_table_walk(pnext, inc, D.sym_len - done, kPhaseCap, &c2, &p2, [](int, int, float, float) {});
        done += c2;
        pnext = p2;
    }
    return pnext;
}

__global__ __launch_bounds__(256) void cfo_walk_kernel(const DemodConst* __restrict__ Dp, int n_frames,
                                                       float* __restrict__ state, unsigned* __restrict__ seg_tab) {
    const DemodConst& D = *Dp;
    const int frame = blockIdx.x * blockDim.x + threadIdx.x;
    if (frame >= n_frames) return;
    float* st = state + (size_t)frame * kStFloats;
    // word 3: the phase the symbol STARTS with (what mix_fft reads); the record is advanced to the phase the symbol ENDS
    // with right here — the walk knows it, and the transform kernels then have no store but their bins
    st[st_cfo_phase] = walk_to_table(D, st[st_cfo], st[st_cfo_phase], seg_tab + (size_t)frame * kSegTabWords);
}

// One work item = one symbol of one frame, 512 points per wavefront (symbol_to_freq2: two wavefronts per frame for
// N = 1024, one for N = 512 — the same code minus the joint last stage), software-pipelined: everything item i+1 needs
// from memory is requested while item i computes.
// UH_MIX2_WAVES = wavefronts per SIMD the register allocator is told to reach (tools/build_variants.sh builds the others).
#ifndef UH_MIX2_WAVES
#define UH_MIX2_WAVES 3
#endif
#ifndef UH_MIX2_WAVES_NOROT
#define UH_MIX2_WAVES_NOROT 4
#endif
template <int LOG2N, bool ROT>
__global__ __launch_bounds__(Fft2Shared<LOG2N>::W * kWave) __attribute__((amdgpu_waves_per_eu(ROT ? UH_MIX2_WAVES : UH_MIX2_WAVES_NOROT, 8))) void mix_fft2_kernel(
    const DemodConst* __restrict__ Dp, const c32* __restrict__ nco, const c32* __restrict__ twiddle,
    const float* __restrict__ audio, size_t frame_stride, const unsigned* __restrict__ frame_offset, int n_frames,
    int sym, c32* __restrict__ fq, const unsigned* __restrict__ seg_tab, int n_sym_batch) {
    __shared__ Fft2Shared<LOG2N, ROT> sh;
    const DemodConst& D = *Dp;
    constexpr int P = Fft2Shared<LOG2N>::P, A = Fft2Shared<LOG2N>::A, M = Fft2Shared<LOG2N>::M, W = Fft2Shared<LOG2N>::W;
    const int h = (W == 2) ? __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6) : 0, lane = threadIdx.x & 63;
    const int rl = (int)(__brev((unsigned)lane) >> 26);
    for (int idx = threadIdx.x; idx < Fft2Shared<LOG2N>::kTwB; idx += W * kWave) {
        const int sA = 31 - __clz(idx / P + 1);
        const int k = idx - P * ((1 << sA) - 1);
        sh.twB[idx] = twiddle[k << (LOG2N - 1 - (A + sA))];
    }
    __syncthreads();
    const int total = n_frames * n_sym_batch;
    // What every item needs of the layout and of the launch, read ONCE and made opaque: left to itself the compiler
    // re-reads D.sym_len, D.cp and the grid size from memory inside every request (scalar registers are short here, a
    // load looks cheaper to it than a spill) and waits for them, ~1000 cycles per item in the stamps of that version.
    int sym_len = D.sym_len, cp = D.cp, grid_step = (int)gridDim.x;
    asm volatile("" : "+s"(sym_len), "+s"(cp), "+s"(grid_step));
    // (the rotating instance is launched for ONE symbol index at a time — its symbols depend on the tracker, launch_demod —,
    // so it carries neither the division nor its constants)
    auto ds_of = [&](int w) { return (!ROT && n_sym_batch > 1) ? w / n_frames : 0; };
    auto request = [&](MixItem& it, int w) {
        const int ds = ds_of(w), f = w - ds * n_frames;
        const float* audio_sym = audio + (size_t)f * frame_stride + (frame_offset ? frame_offset[f] : 0u) + (size_t)(sym + ds) * sym_len;
        prefetch_symbol2<LOG2N, ROT>(sh, cp, h, lane, audio_sym);
        request_item(it, (ROT && seg_tab) ? seg_tab + (size_t)f * kSegTabWords : nullptr, lane);
    };
    Mix2Lane<LOG2N> lc;
    lc.w6 = twiddle[lane << (LOG2N - 7)];                  // stage s of the M-point transform: twiddle[k << (LOG2N - 1 - s)]
    lc.w7[0] = twiddle[lane << (LOG2N - 8)]; lc.w7[1] = twiddle[(lane + 64) << (LOG2N - 8)];
    lc.w8[0] = twiddle[lane << (LOG2N - 9)]; lc.w8[1] = twiddle[(lane + 192) << (LOG2N - 9)];
    lc.w_last = (W == 2) ? twiddle[h ? (M - 64 + lane) : lane] : mk(0.0f, 0.0f);
#pragma unroll
    for (int j = 0; j < P / 2; ++j) lc.wA[j] = twiddle[j << (LOG2N - A)];
    asm volatile("" ::"v"(lc.w6.re), "v"(lc.w6.im), "v"(lc.w7[0].re), "v"(lc.w7[0].im), "v"(lc.w7[1].re), "v"(lc.w7[1].im),
                 "v"(lc.w8[0].re), "v"(lc.w8[0].im), "v"(lc.w8[1].re), "v"(lc.w8[1].im), "v"(lc.w_last.re), "v"(lc.w_last.im));
    int os_ds = -1;
    MixItem cur;
    c32 pending = mk(0.0f, 0.0f), pending_hi = mk(0.0f, 0.0f);   // the previous item's bin(s) of this lane, stored with the next request
    // wavefront 0 holds bins `lane`, wavefront 1 bins N - 64 + lane (W = 1: the one wavefront holds both); the row keeps
    // the fq_half of each side next to DC
    const int fh = D.fq_half;
    const bool fq_mine = h ? (lane >= 64 - fh) : (lane < fh);
    const bool fq_mine_hi = W == 1 && lane >= 64 - fh;
    // positions inside the row: the pilots' bins first (DemodConst::fq_pos)
    const int fq_slot = fq_mine ? (int)D.fq_pos[h ? fh + lane - (64 - fh) : lane] : 0;
    const int fq_slot_hi = fq_mine_hi ? (int)D.fq_pos[fh + lane - (64 - fh)] : 0;
    auto store_bins = [&](int w_item) {
        c32* row = fq + (size_t)w_item * (2 * fh);
        if (fq_mine) row[fq_slot] = pending;
        if (fq_mine_hi) row[fq_slot_hi] = pending_hi;
    };
    int w_stored = -1;
    int w = blockIdx.x;
    if (w < total) request(cur, w);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the first item has no bin store in front of it (symbol_to_freq2)
    for (int par = 0; w < total; w += grid_step, par ^= 1) {
        const int ds = ds_of(w);
        if (ds != os_ds) {                                   // the oscillator at this lane's samples: once per symbol index
            const c32* nco_sym = nco + (size_t)(sym + ds) * sym_len;
#pragma unroll
            for (int qp = 0; qp < P; ++qp) lc.os[qp] = nco_sym[cp + W * (rl + 64 * qp) + h];
            os_ds = ds;
            // the compiler's wait for these loads belongs HERE and not at their first use in the item, where it would
            // also wait for the requests of the next item issued in between
#pragma unroll
            for (int qp = 0; qp < P; ++qp) asm volatile("" ::"v"(lc.os[qp].re), "v"(lc.os[qp].im));
            if constexpr (ROT) {                               // ... into the lane's LDS slots; the registers are free again
#pragma unroll
                for (int qp = 0; qp < P; ++qp) { sh.osc[h][64 * qp + lane] = lc.os[qp]; lc.os[qp] = mk(0.0f, 0.0f); }
            }
        }
        const int next = w + grid_step;
        Stamps stamps;
        c32 bin, bin_hi;
        symbol_to_freq2<LOG2N, ROT>(sh, D, h, lane, cur, lc, twiddle, bin, bin_hi, par, cp, sym_len,
                               [&]() {
                                   if (w_stored >= 0) store_bins(w_stored);
                                   if (next < total) request(cur, next);
                               }, stamps);
        pending = bin; pending_hi = bin_hi;
        w_stored = w;
        UH_STAMP(10);
        stamps.store((size_t)w * W + h, lane);
    }
    if (w_stored >= 0) store_bins(w_stored);
}

// mode 0: data symbol (updateChannelEstimate + equalize + demodulateSymbol)
// mode 1: training symbol `sym` of n_train (estimateChannelFromLTS), finishing on the last one
__device__ __forceinline__ void load_track(const float* __restrict__ st, Track& tr) {
    tr.freq_offset_hz = st[st_cfo]; tr.freq_offset_filtered = st[st_cfo_filt]; tr.cfo_phase = st[st_cfo_phase];
    tr.noise_variance = st[st_noise]; tr.snr_linear = st[st_snr]; tr.timing = st[st_timing];
    tr.ppc = mk(st[st_ppc_re], st[st_ppc_im]); tr.cpc = mk(st[st_cpc_re], st[st_cpc_im]);
    const int flags = (int)st[st_flags];
    tr.cpc_init = flags & 1; tr.has_prev = (flags >> 1) & 1; tr.has_dprev = (flags >> 2) & 1;
    tr.snr_symbol_count = (int)st[st_count]; tr.symbols_since_sync = (int)st[st_since];
}
__device__ __forceinline__ void store_track(float* __restrict__ st, const Track& tr) {
    st[st_cfo] = tr.freq_offset_hz; st[st_cfo_filt] = tr.freq_offset_filtered;
    st[st_noise] = tr.noise_variance; st[st_snr] = tr.snr_linear; st[st_timing] = tr.timing;
    st[st_ppc_re] = tr.ppc.re; st[st_ppc_im] = tr.ppc.im; st[st_cpc_re] = tr.cpc.re; st[st_cpc_im] = tr.cpc.im;
    st[st_flags] = (float)(tr.cpc_init | (tr.has_prev << 1) | (tr.has_dprev << 2));
    st[st_count] = (float)tr.snr_symbol_count; st[st_since] = (float)tr.symbols_since_sync;
}
__device__ __forceinline__ LaneConst lane_constants(const DemodConst& D) {
    const int lane = threadIdx.x;
    LaneConst lc;
    auto fq_of = [&](int bin) { return fq_index(D, bin); };
    const int ps = (lane < D.n_pilot) ? D.pilot_slot[lane] : 0;
    lc.pilot_slot = ps; lc.pilot_fq = fq_of(D.bin[ps]); lc.pilot_k = D.k_of[ps];
    lc.pilot_seq = (lane < D.n_pilot) ? D.pilot_seq[lane] : mk(1.0f, 0.0f);
    const int dsl = (lane < D.n_data) ? D.data_slot[lane] : 0;
    lc.data_slot = dsl; lc.data_fq = fq_of(D.bin[dsl]); lc.data_k = D.k_of[dsl];
    const int q = (lane < D.n_interp) ? lane : 0;
    lc.i_dst = D.interp_slot[q]; lc.i_lo = D.interp_lo[q]; lc.i_hi = D.interp_hi[q]; lc.i_alpha = D.interp_alpha[q];
    lc.slot_k = D.k_of[(lane < D.n_carriers) ? lane : 0];
    lc.zc = D.sync_seq[lane % D.n_carriers];
    return lc;
}

// Training symbol `sym` of the presynced entry (estimateChannelFromLTS): one wavefront per frame
__global__ __launch_bounds__(kWave, 4) void train_kernel(const DemodConst* __restrict__ Dp, int n_frames, int sym,
                                                         float* __restrict__ state, const c32* __restrict__ fq_all) {
    __shared__ TrackShared sh;
    const DemodConst& D = *Dp;
    const int lane = threadIdx.x;
    const LaneConst lc = lane_constants(D);
    for (int frame = blockIdx.x; frame < n_frames; frame += gridDim.x) {
        float* st = state + (size_t)frame * kStFloats;
        const c32* fq = fq_all + (size_t)frame * (2 * D.fq_half);
        Track tr;
        load_track(st, tr);
        sh.H[lane] = reinterpret_cast<const c32*>(st + kStH)[lane];
        wave_sync();
        c32 lts_acc = (sym > 0 && lane < D.n_pilot) ? reinterpret_cast<const c32*>(st + kStLts)[lane] : mk(0.0f, 0.0f);
        lts_symbol(sh, D, lc, sym, D.n_train, lts_acc, fq);
        if (sym == D.n_train - 1) lts_finish(sh, D, lc, tr, D.n_train, lts_acc);
        else if (lane < D.n_pilot) reinterpret_cast<c32*>(st + kStLts)[lane] = lts_acc;
        wave_sync();
        reinterpret_cast<c32*>(st + kStH)[lane] = sh.H[lane];
        if (lane < D.n_pilot) reinterpret_cast<c32*>(st + kStHp)[lane] = sh.H[lc.pilot_slot];      // compact_pilot_state
        if (lane == 0) store_track(st, tr);
        wave_sync();
    }
}

// Data symbol, carrier half: the pilot half of updateChannelEstimate already ran in track_pilot_kernel; here
// the interpolation between the pilots, the equaliser and the demapper — one wavefront per frame, one lane per
// carrier.
// EQ (live streams that feed a constellation display, ultra_hip_demod_stream_batch_eq): also write each symbol's equalized
// data carriers, eq_out[frame * eq_stride + ds * kMaxCarriers + carrier] — an instance of its own, so that the batch path's
// kernels stay what they were.
template <int MOD, bool EQ = false>
__global__ __launch_bounds__(kWave, 6) void track_kernel(
    const DemodConst* __restrict__ Dp, int n_frames, int data_sym, float* __restrict__ state,
    const c32* __restrict__ fq_all, float* __restrict__ llr, size_t llr_stride, float* __restrict__ state_out,
    int n_sym_batch, int synced_loop, c32* __restrict__ eq_out = nullptr, size_t eq_stride = 0) {
    // synced_loop: these symbols reach the demodulator through process()'s SYNCED loop (demodulator.cpp:672-697), which calls
    // updateChannelEstimate for EVERY layout, although the context is of the presynced entry — the tail of a frame whose head
    // went through processPresynced (whose own loop skips the update without pilots, :954-960).
    // n_sym_batch > 1 (zero-CFO layouts, whose bins of ALL symbols are already there: launch_demod): the wavefront walks
    // data symbols data_sym .. data_sym + n_sym_batch - 1 of its frame with the tracker state in registers and LDS —
    // one record read and one record write per frame instead of one of each per symbol, one launch instead of n_sym.
    __shared__ TrackShared sh;
    const DemodConst& D = *Dp;
    const int lane = threadIdx.x;
    const LaneConst lc = lane_constants(D);
    for (int frame = blockIdx.x; frame < n_frames; frame += gridDim.x) {
        float* st = state + (size_t)frame * kStFloats;
        Track tr;
        load_track(st, tr);
        // A layout without pilots leaves the pilot half of updateChannelEstimate three scalar effects (no pilot phase
        // differences: pilot_phase_correction = 1, :421-470; prev_pilot_phases stays empty; ++snr_symbol_count, :595) —
        // taken here instead of in a track_pilot_kernel launch of their own (SYNCED entry; the presynced entry without
        // pilots never ran the estimate, demodulator.cpp:936-960).
        const bool presynced_loop = D.presynced && !synced_loop;
        const bool scalar_pilot_half = D.n_pilot == 0 && !presynced_loop;
        const bool compact = compact_pilot_state(D);
        if (!compact) sh.H[lane] = reinterpret_cast<const c32*>(st + kStH)[lane];
        else if (lane < D.n_pilot) sh.H[lc.pilot_slot] = reinterpret_cast<const c32*>(st + kStHp)[lane];   // the rest is interpolated before it is read
        c32 dprev = D.differential ? reinterpret_cast<const c32*>(st + kStDprev)[lane] : mk(1.0f, 0.0f);
        // adaptive equaliser (coherent layouts): the data carrier's weight and RLS gain live in its lane between symbols
        const bool adaptive = D.adaptive_eq != 0 && !D.differential;
        AdaptiveEq ad;
        ad.w = mk(1.0f, 0.0f); ad.P = 1.0f;
        if (adaptive) { ad.w = reinterpret_cast<const c32*>(st + kStLms)[lane]; ad.P = st[kStRlsP + lane]; }
        wave_sync();
        // differential layouts: |dprev| travels with dprev through the symbols of this launch; without pilots the timing
        // estimate never moves, so its phase factor is evaluated once per frame (equalize_demap)
        float dprev_abs = -1.0f;
        const bool tc_is_fixed = D.differential && D.n_pilot == 0 && n_sym_batch > 1;
        c32 tc0 = mk(1.0f, 0.0f);
        if (tc_is_fixed) tc0 = cexpj_bounded(timing_phase_of(lc.data_k, tr.timing, D.log2_fft));
        for (int ds = 0; ds < n_sym_batch; ++ds) {
            const c32* fq = fq_all + ((size_t)ds * n_frames + frame) * (2 * D.fq_half);
            if (scalar_pilot_half) { tr.ppc = mk(1.0f, 0.0f); tr.has_prev = 0; tr.snr_symbol_count++; }
            if (!presynced_loop || D.n_pilot != 0) {
                finish_channel_estimate(sh, D, lc, tr);
                // updateChannelEstimate seeds the weights from its estimate while snr_symbol_count < 3 (:569-581) — the
                // count BEFORE its increment at the end of the update, which the pilot half has already taken
                if (adaptive && tr.snr_symbol_count - 1 < 3 && lane < D.n_data) ad.w = sh.H[lc.data_slot];
            }
            equalize_demap<MOD>(sh, D, lc, tr, dprev, fq,
                                llr + (size_t)frame * llr_stride + (size_t)(data_sym + ds) * D.llrs_per_symbol,
                                &dprev_abs, tc_is_fixed ? &tc0 : nullptr, nullptr, adaptive ? &ad : nullptr,
                                EQ ? eq_out + (size_t)frame * eq_stride + (size_t)ds * kMaxCarriers : nullptr);
            wave_sync();
        }
        // write the record back
        if (!compact) reinterpret_cast<c32*>(st + kStH)[lane] = sh.H[lane];
        else if (lane < ((D.n_pilot + 15) & ~15))               // whole 128-byte lines: no partial-sector writes
            reinterpret_cast<c32*>(st + kStHp)[lane] = (lane < D.n_pilot) ? sh.H[lc.pilot_slot] : mk(0.0f, 0.0f);
        if (D.differential) reinterpret_cast<c32*>(st + kStDprev)[lane] = dprev;
        if (adaptive) { reinterpret_cast<c32*>(st + kStLms)[lane] = ad.w; st[kStRlsP + lane] = ad.P; }
        if (lane == 0) {
            // only what the carrier half owns (equalize_demap: pilot_phase_correction, has_dprev); the rest of the
            // record's scalars belong to the pilot half and to cfo_walk_kernel — one partial store instead of the line
            st[st_ppc_re] = tr.ppc.re; st[st_ppc_im] = tr.ppc.im;
            st[st_flags] = (float)(tr.cpc_init | (tr.has_prev << 1) | (tr.has_dprev << 2));
            if (scalar_pilot_half) st[st_count] = (float)tr.snr_symbol_count;
            if (state_out) {
                float* so = state_out + (size_t)frame * ULTRA_HIP_STATE_FLOATS;
                so[ULTRA_HIP_STATE_FREQ_OFFSET_HZ] = tr.freq_offset_hz;
                so[ULTRA_HIP_STATE_NOISE_VARIANCE] = tr.noise_variance;
                so[ULTRA_HIP_STATE_SNR_LINEAR] = tr.snr_linear;
                so[ULTRA_HIP_STATE_TIMING_OFFSET] = tr.timing;
                so[ULTRA_HIP_STATE_CFO_PHASE] = tr.cfo_phase;
                so[ULTRA_HIP_STATE_MIXER_PHASE] = D.mixer_phase_end;
                so[ULTRA_HIP_STATE_SYMBOLS] = (float)tr.snr_symbol_count;
                so[ULTRA_HIP_STATE_RESERVED] = 0.0f;
            }
        }
        wave_sync();
    }
}

// ---------------------------------------------------------------------------
// Differential layouts without pilots, SYNCED entry, at most 32 carriers (the 512-point presets carry 30): TWO frames per
// wavefront.  What track_kernel does for such a layout is per-carrier work only — equalise against the lane's own
// channel_estimate entry (never updated without pilots, but for the sign of its zeros in the decision-directed block),
// the demapper's atan2f / hypotf / sinf / cosf, the lane's differential reference — plus three scalars per frame; with
// one frame per wavefront 34 of 64 lanes idle through all of it (track_kernel<DQPSK>: 93 % of the vector issue slots
// busy, profiles/r03_issue_model.txt).  Lane l works on carrier l % 32 of frame 2 p + l / 32; everything per frame that
// track_kernel keeps wave-uniform is per lane here, the one ballot (strong carriers, demodulator.cpp:362-434) is masked
// to the lane's half.  Same operations on the same operands: bit-identical records and soft bits.
template <int MOD>
__global__ __launch_bounds__(kWave, 6) void track_diff_pair_kernel(
    const DemodConst* __restrict__ Dp, int n_frames, int data_sym, float* __restrict__ state,
    const c32* __restrict__ fq_all, float* __restrict__ llr, size_t llr_stride, float* __restrict__ state_out,
    int n_sym_batch) {
    static_assert(MOD == ULTRA_MOD_DBPSK || MOD == ULTRA_MOD_DQPSK || MOD == ULTRA_MOD_D8PSK, "differential layouts");
    constexpr int kBits = (MOD == ULTRA_MOD_DBPSK) ? 1 : (MOD == ULTRA_MOD_DQPSK) ? 2 : 3;
    const DemodConst& D = *Dp;
    const int lane = threadIdx.x, g = lane >> 5, sub = lane & 31;
    const bool is_data = sub < D.n_data;
    const int dsl = is_data ? D.data_slot[sub] : 0;
    const int data_fq = fq_index(D, D.bin[dsl]), data_k = D.k_of[dsl];
    const unsigned long long half_mask = g ? 0xffffffff00000000ull : 0x00000000ffffffffull;
    const int n_pairs = (n_frames + 1) / 2;
    for (int pair = blockIdx.x; pair < n_pairs; pair += gridDim.x) {
        // an odd frame count: the upper half of the last wavefront repeats the last frame and stores nothing
        const bool live = 2 * pair + g < n_frames;
        const int frame = live ? 2 * pair + g : n_frames - 1;
        float* st = state + (size_t)frame * kStFloats;
        Track tr;
        load_track(st, tr);
        c32 H = reinterpret_cast<const c32*>(st + kStH)[dsl];
        c32 dprev = reinterpret_cast<const c32*>(st + kStDprev)[sub];
        if (!tr.has_dprev) dprev = mk(1.0f, 0.0f);             // (1,0) reference before the first symbol
        // Two of the three hypotf per carrier and symbol are repeats: |prev| is the |sym| of the symbol before, and the
        // decision-directed block takes |sym| again; and the timing phase does not move without pilots.  Each is evaluated
        // once — the same function on the same operand.
        float prev_abs = cabs_(dprev);
        const c32 tc = cexpj_bounded(timing_phase_of(data_k, tr.timing, D.log2_fft));
        for (int ds = 0; ds < n_sym_batch; ++ds) {
            const c32* fq = fq_all + ((size_t)ds * n_frames + frame) * (2 * D.fq_half);
            // the pilot half of a layout without pilots (track_kernel: scalar_pilot_half)
            tr.ppc = mk(1.0f, 0.0f); tr.has_prev = 0; tr.snr_symbol_count++;
            // equalize, differential branch (equalize_demap)
            c32 eq = mk(0.0f, 0.0f);
            float nv = 100.0f;
            if (is_data) {
                const c32 received = fq[data_fq];
                const float h_power = cnorm(H);
                if (h_power > 1e-6f) {
                    const c32 t = cdivf(cmul(received, cconj(H)), h_power);
                    eq = cmul(cmul(t, tr.ppc), tc);
                    nv = tr.noise_variance / h_power;
                } else {
                    eq = cmul(cmul(received, tr.ppc), tc);
                    nv = 100.0f;
                }
                nv = fmax_std(1e-6f, fmin_std(100.0f, nv));
            }
            const float eq_abs = cabs_(eq);
            if (is_data) {
                float out[kBits];
                demap_carrier<MOD>(eq, dprev, nv * D.ce_margin, out, eq_abs, prev_abs);
                if (live) {
                    float* dst = llr + (size_t)frame * llr_stride + (size_t)(data_sym + ds) * D.llrs_per_symbol + sub * kBits;
#pragma unroll
                    for (int b = 0; b < kBits; ++b) dst[b] = out[b];
                }
                dprev = eq;
                prev_abs = eq_abs;
            }
            if ((MOD == ULTRA_MOD_DQPSK || MOD == ULTRA_MOD_D8PSK) && tr.snr_symbol_count >= 1) {
                // decision-directed block: see equalize_demap (the multiplies only touch the sign of exact zeros)
                bool strong = false;
                if (is_data) {
                    const float a = eq_abs;
                    strong = (a * a) > 0.1f;
                    if (strong) H = cmul(H, mk(1.0f, -0.0f));
                }
                const int valid = __popcll(__ballot(strong) & half_mask);
                if (valid >= 5) {
                    c32 p = cmul(cscale(tr.ppc, 1.0f), mk(1.0f, -0.0f));
                    const float mag = cabs_(p);
                    if (mag > 0.01f) p = cdivf(p, mag);
                    tr.ppc = p;
                }
            }
            tr.has_dprev = 1;
        }
        if (live && is_data) {
            reinterpret_cast<c32*>(st + kStH)[dsl] = H;
            reinterpret_cast<c32*>(st + kStDprev)[sub] = dprev;
        }
        if (live && sub == 0) {
            st[st_ppc_re] = tr.ppc.re; st[st_ppc_im] = tr.ppc.im;
            st[st_flags] = (float)(tr.cpc_init | (tr.has_prev << 1) | (tr.has_dprev << 2));
            st[st_count] = (float)tr.snr_symbol_count;
            if (state_out) {
                float* so = state_out + (size_t)frame * ULTRA_HIP_STATE_FLOATS;
                so[ULTRA_HIP_STATE_FREQ_OFFSET_HZ] = tr.freq_offset_hz;
                so[ULTRA_HIP_STATE_NOISE_VARIANCE] = tr.noise_variance;
                so[ULTRA_HIP_STATE_SNR_LINEAR] = tr.snr_linear;
                so[ULTRA_HIP_STATE_TIMING_OFFSET] = tr.timing;
                so[ULTRA_HIP_STATE_CFO_PHASE] = tr.cfo_phase;
                so[ULTRA_HIP_STATE_MIXER_PHASE] = D.mixer_phase_end;
                so[ULTRA_HIP_STATE_SYMBOLS] = (float)tr.snr_symbol_count;
                so[ULTRA_HIP_STATE_RESERVED] = 0.0f;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// The DEFERRED carrier half of the coherent layouts with pilots (SYNCED entry: the headline configuration).
//
// For these layouts the carrier half of a symbol (interpolate, equalise, demap) does not feed back into the tracker: no
// decision-directed block, and channel_estimate at the data carriers is rebuilt from the pilots every symbol.  The only
// thing it used to hand back is the pilots' estimates re-rotated by the timing phase (:561-567) — per-pilot work that
// track_pilot_kernel now does itself.  So the per-symbol chain is  cfo_walk -> mix_fft -> track_pilot  and the carrier
// half of ALL symbols runs afterwards in ONE launch over (symbol, frame) items (track_all_kernel), from the bins of every
// symbol and a 256-byte record per item that track_pilot_kernel leaves behind: the de-rotated pilots' estimates and the
// scalars the carrier half reads.  (Measured and not kept, profiles/r03_lane_per_frame_pilot.txt: the whole pilot half
// + the walk as ONE LANE per frame — sums as plain loops, nothing replicated, 16 x fewer wavefronts — is 0.14-0.19 ms
// per launch against 0.084 + 0.03: the per-pilot terms (hypot, two divisions, atan2f) dominate and were already spread
// over 16 lanes; one lane per frame turns them into a serial chain that four wavefronts per SIMD cannot cover.)
// The carrier half of every data symbol of every frame in ONE launch (coherent layouts of the lean chain): item
// w = s * n_frames + f reads the bins Fq[w] and the record trk_rec[w] and writes the LLRs of symbol sym0 + s of frame f.
template <int MOD>
__global__ __launch_bounds__(kWave, 6) void track_all_kernel(const DemodConst* __restrict__ Dp, int n_frames, int sym0, int n_sym_batch,
                                                             const float* __restrict__ trk_rec, const c32* __restrict__ fq_all,
                                                             float* __restrict__ llr, size_t llr_stride, float* __restrict__ state_out,
                                                             const float* __restrict__ state) {
    __shared__ TrackShared sh;
    const DemodConst& D = *Dp;
    const int lane = threadIdx.x;
    const LaneConst lc = lane_constants(D);
    const int total = n_frames * n_sym_batch;
    const int rec_floats = trk_rec_floats(D.n_pilot), rec_hp = trk_rec_hp(D.n_pilot);
    const int rec_mask = (D.n_pilot <= kTrkRecCompactPilots) ? 1 : 7;
    // What an item reads — the record's scalars, the pilots' estimates, this lane's bin — is requested one item ahead, as
    // per-lane words (a uniform load would be waited for on the spot): the item then starts on data that has arrived.
    float nx_rec = 0.0f;
    c32 nx_hp = mk(0.0f, 0.0f), nx_fq = mk(0.0f, 0.0f);
    auto request = [&](int w) {
        const float* rec = trk_rec + (size_t)w * rec_floats;
        nx_rec = rec[lane & rec_mask];
        if (lane < D.n_pilot) nx_hp = reinterpret_cast<const c32*>(rec + rec_hp)[lane];
        if (lane < D.n_data) nx_fq = fq_all[(size_t)w * (2 * D.fq_half) + lc.data_fq];
    };
    if ((int)blockIdx.x < total) request((int)blockIdx.x);
    for (int w = blockIdx.x; w < total; w += gridDim.x) {
        const int frame = w % n_frames, ds = w / n_frames;
        Track tr;
        const float my_rec = nx_rec;
        const c32 my_hp = nx_hp, my_fq = nx_fq;
        if (w + (int)gridDim.x < total) request(w + (int)gridDim.x);
        tr.noise_variance = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_rec), tk_noise));
        tr.timing = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_rec), tk_timing));
        tr.ppc = mk(1.0f, 0.0f); tr.cpc = mk(1.0f, 0.0f);
        tr.cpc_init = 1; tr.has_prev = 1; tr.has_dprev = 0; tr.snr_symbol_count = 0; tr.symbols_since_sync = 0;
        if (lane < D.n_pilot) sh.H[lc.pilot_slot] = my_hp;      // the rest is interpolated before it is read
        c32 dprev = mk(1.0f, 0.0f);
        wave_sync();
        finish_channel_estimate(sh, D, lc, tr);
        equalize_demap<MOD>(sh, D, lc, tr, dprev, fq_all + (size_t)w * (2 * D.fq_half),
                            llr + (size_t)frame * llr_stride + (size_t)(sym0 + ds) * D.llrs_per_symbol, nullptr, nullptr, &my_fq);
        if (state_out && lane == 0 && ds == n_sym_batch - 1) {   // the tracker after the last symbol of the launch:
            // the frame's tracker record as the last pilot launch left it (every pilot launch of the call ran before this one)
            const float* st = state + (size_t)frame * kStFloats;
            float* so = state_out + (size_t)frame * ULTRA_HIP_STATE_FLOATS;
            so[ULTRA_HIP_STATE_FREQ_OFFSET_HZ] = st[st_cfo];
            so[ULTRA_HIP_STATE_NOISE_VARIANCE] = st[st_noise];
            so[ULTRA_HIP_STATE_SNR_LINEAR] = st[st_snr];
            so[ULTRA_HIP_STATE_TIMING_OFFSET] = st[st_timing];
            so[ULTRA_HIP_STATE_CFO_PHASE] = st[st_cfo_phase];
            so[ULTRA_HIP_STATE_MIXER_PHASE] = D.mixer_phase_end;
            so[ULTRA_HIP_STATE_SYMBOLS] = st[st_count];
            so[ULTRA_HIP_STATE_RESERVED] = 0.0f;
        }
        wave_sync();
    }
}

}  // namespace dev
}  // namespace ultra_hip
#endif
