// phase_table.h — exact closed form of the reference's per-sample CFO phase recurrence.
//
// Impl::toBaseband (src/ofdm/channel_equalizer.cpp:43-50) advances a float phase
// once per audio sample:
//     phase += inc;                       (f32 add, one rounding)
//     if (phase >  M_PI) phase -= 2*M_PI; (compare and subtract in f64, narrowed)
//     else if (phase < -M_PI) phase += 2*M_PI;
// 1120 dependent steps per OFDM symbol — a serial chain that would leave 63 of 64
// lanes idle.  Float addition is not associative, so the chain cannot be re-bracketed;
// but it can be jumped EXACTLY: while the phase stays inside one binade (same sign
// and exponent) it lives on the grid m*u (u = ulp of the binade) and
//     fl(m*u + inc) = (m + q)*u,   q = RN(inc/u)
// with the same integer q for every m, except in the one binade where inc/u has
// fractional part exactly 1/2 (ties-to-even): there the first step fixes the
// parity of m and every later step is again constant.  So the sequence is
// piecewise affine: phase[i0 + t] = base + t*step, exactly, for all t that keep
// the value in the binade and inside (-pi, pi].  build() walks the (few) segments
// with real float steps at every boundary — binade changes, zero crossing, +-pi
// wraps — and lookups evaluate  base + (float)t * step  in single precision, which
// is exact because the product and the sum are representable (they lie on the grid).
//
// Checked against the serial recurrence position by position on the host
// (tools/phase_table_check.cpp: random starts/increments, ties, zero crossings,
// wraps, denormals) and on the GPU through the parity tests.
#ifndef ULTRA_PHASE_TABLE_H
#define ULTRA_PHASE_TABLE_H

#include <stdint.h>
#include "pinned_math.h"

namespace um {

struct PhaseSeg { int start; float base; float step; };

constexpr double kPiD = 3.14159265358979323846;
constexpr double kTwoPiD = 2.0 * 3.14159265358979323846;

// one step of the reference recurrence.  The reference compares in double against M_PI; pi is not
// a float, so for a float p: (double)p > M_PI  <=>  p > kPiMax (the largest float below pi).
UM_FN float phase_step(float p, float inc) {
    const float pi_max = as_f32(0x40490FDAu);
    p += inc;
    if (p > pi_max) p = (float)((double)p - kTwoPiD);
    else if (p < -pi_max) p = (float)((double)p + kTwoPiD);
    return p;
}

UM_FN bool same_binade(float a, float b) {
    const uint32_t ua = as_u32(a), ub = as_u32(b);
    const uint32_t ea = (ua >> 23) & 0xff;
    return ((ua ^ ub) < 0x00800000u) && ea != 0 && ea != 0xff;
}

// v >= 0, a multiple of the grid 2^(E-150) of the binade with biased exponent E (E >= 1), and
// smaller than 2^24 grid units: the multiple, by integer manipulation of the bits (no dependence
// on the denormal mode of the float pipeline).
UM_FN uint32_t grid_units(float v, uint32_t E) {
    const uint32_t b = as_u32(v);
    uint32_t e = b >> 23, m = b & 0x007fffffu;
    if (e != 0) m |= 0x00800000u; else e = 1;
    const uint32_t sh = E - e;
    return (sh < 32u) ? (m >> sh) : 0u;
}

// float approximation of 1/q for the quotient estimate (any error is removed by the integer fix-up)
UM_FN float approx_rcp(float q) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcpf(q);
#else
    return 1.0f / q;
#endif
}

// Build segments for positions [0, n) starting from phase p0 (= phase[0]).
// Writes at most `cap` segments (only when `write`: on the GPU every lane walks the same
// uniform control flow and lane 0 stores); returns their count, the number of positions they
// cover in *covered (== n unless cap was hit) and phase[*covered] in *p_next.
//
// This is a serial chain (every segment starts where the previous one ends) that a whole
// wavefront waits for, so it is written for latency: single-precision and integer operations
// only — every product and sum below is exact because its result lies on the binade's grid —
// and the segment length floor(room/|d|) comes from an approximate reciprocal plus an exact
// integer remainder fix-up instead of a correctly rounded division.
UM_FN int phase_table_build(float p0, float inc, int n, PhaseSeg* seg, int cap, int* covered, float* p_next,
                             bool write = true) {
    const float pi_max = as_f32(0x40490FDAu);   // largest float <= pi: no wrap while |phase| <= pi_max
    int i = 0, ns = 0;
    float p = p0;
    while (i < n && ns < cap) {
        const float p1 = phase_step(p, inc);
        const float p2 = phase_step(p1, inc);
        int len = 1;
        float d = 0.0f;
        if (same_binade(p, p1) && same_binade(p1, p2)) {
            const float d1 = p1 - p, d2 = p2 - p1;     // exact: same exponent
            if (d1 == d2) {
                d = d1;
                if (d1 == 0.0f) {
                    len = n - i;                        // increment absorbed: constant from here on
                } else {
                    const uint32_t up = as_u32(p);
                    const float ap = as_f32(up & 0x7fffffffu);                 // |p|
                    const bool growing = ((up >> 31) != 0) == (d1 < 0.0f);     // |phase| increases
                    float room;                                                // exact differences inside one binade
                    if (growing) {
                        float top = as_f32((up & 0x7fffffffu) | 0x007fffffu);  // largest value of the binade
                        if (top > pi_max) top = pi_max;
                        room = top - ap;
                    } else {
                        // stay strictly above 2^e: a sum that lands just below the binade is rounded
                        // on the finer grid of the binade underneath, so exactly 2^e is not safe
                        const float bottom = as_f32((up & 0x7f800000u) + 1u);   // 2^e + ulp
                        room = ap - bottom;
                    }
                    // t = floor(room / |d|), both integers (< 2^24) times the binade's ulp
                    int t = 0;
                    if (room > 0.0f) {
                        const uint32_t E = (up >> 23) & 0xffu;
                        const int R = (int)grid_units(room, E);
                        const int Q = (int)grid_units(fabsf(d1), E);           // >= 1
                        t = (int)((float)R * approx_rcp((float)Q));             // within a few units of the floor
                        int r = R - t * Q;                                      // |t*Q| < 2^26: no overflow
                        while (r < 0) { --t; r += Q; }
                        while (r >= Q) { ++t; r -= Q; }
                    }
                    const int lim = n - i - 1;
                    if (t > lim) t = lim;
                    len = t + 1;                       // positions t = 0 .. floor(room/|d|)
                }
            }
        }
        if (write) { seg[ns].start = i; seg[ns].base = p; seg[ns].step = d; }   // one lane stores on the GPU
        ++ns;
        if (len == 1) {
            p = p1;
        } else {
            const float last = p + (float)(len - 1) * d;    // exact: (len-1)*|d| <= room, result on the binade's grid
            p = phase_step(last, inc);
        }
        i += len;
    }
    *covered = i;
    *p_next = p;
    return ns;
}

// phase at position i (seg = the segment with start <= i < next start); both operations are exact
UM_FN float phase_table_eval(const PhaseSeg& s, int i) {
    return s.base + (float)(i - s.start) * s.step;
}

}  // namespace um
#endif
