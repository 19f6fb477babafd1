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
// a float, so for a float s: (double)s > M_PI  <=>  s > pi_max (the largest float below pi), and
// s + 2*pi == s - (-2*pi): one rarely taken branch covers both wraps.
UM_FN float phase_step(float p, float inc) {
    const float pi_max = as_f32(0x40490FDAu);
    float s = p + inc;
    if (fabsf(s) > pi_max) s = (float)((double)s - ((s < 0.0f) ? -kTwoPiD : kTwoPiD));
    return s;
}

// v >= 0, a multiple of the grid 2^(E-150) of the binade with biased exponent E (E >= 1), and
// smaller than 2^24 grid units: the multiple, by integer manipulation of the bits (no dependence
// on the denormal mode of the float pipeline).  Anything else yields some harmless integer.
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

// Walk the segments of positions [0, n) starting from phase p0 (= phase[0]); emit(k, start, base,
// step) receives segment k.  At most `cap` segments; returns their count, the number of positions
// they cover in *covered (== n unless cap was hit) and phase[*covered] in *p_next.
//
// This is a serial chain (every segment starts where the previous one ends) that a whole
// wavefront waits for, so it is written for latency: single-precision and integer operations
// only — every product and sum below is exact because its result lies on the binade's grid —,
// selects instead of branches (the only branches left are the rare +-pi wraps), and the segment
// length floor(room/|d|) from an approximate reciprocal plus an exact integer remainder fix-up
// instead of a correctly rounded division.
template <class Emit>
UM_FN int phase_table_walk(float p0, float inc, int n, int cap, int* covered, float* p_next, Emit emit) {
    const float pi_max = as_f32(0x40490FDAu);   // largest float <= pi: no wrap while |phase| <= pi_max
    int i = 0, ns = 0;
    float p = p0;
    while (i < n && ns < cap) {
        const float p1 = phase_step(p, inc);
        const float p2 = phase_step(p1, inc);
        const float d1 = p1 - p, d2 = p2 - p1;         // exact when the three share a binade
        const uint32_t up = as_u32(p), u1 = as_u32(p1), u2 = as_u32(p2);
        const uint32_t E = (up >> 23) & 0xffu;
        // p, p1, p2 in one binade (same sign and exponent, normal) and equally spaced: from here the
        // recurrence is the arithmetic progression p + t*d1 for as long as it stays in the binade
        const bool reg = ((up ^ u1) < 0x00800000u) & ((u1 ^ u2) < 0x00800000u) & (E != 0u) & (E != 0xffu) & (d1 == d2);
        const float ap = as_f32(up & 0x7fffffffu);                          // |p|
        const bool growing = ((up >> 31) != 0) == (d1 < 0.0f);              // |phase| increases
        float top = as_f32((up & 0x7fffffffu) | 0x007fffffu);               // largest value of the binade
        top = (top > pi_max) ? pi_max : top;
        // shrinking: stay strictly above 2^e — a sum that lands just below the binade is rounded on
        // the finer grid of the binade underneath, so exactly 2^e is not safe
        const float bottom = as_f32((up & 0x7f800000u) + 1u);               // 2^e + ulp
        float room = growing ? (top - ap) : (ap - bottom);                  // exact differences inside one binade
        room = (reg && room > 0.0f) ? room : 0.0f;
        // t = floor(room / |d1|), both integers (< 2^24) times the binade's ulp
        const int R = (int)grid_units(room, E);
        int Q = (int)grid_units(as_f32(as_u32(d1) & 0x7fffffffu), E);
        Q = (Q > 0) ? Q : 1;                                                // d1 == 0 or irregular: unused
        const float rq = approx_rcp((float)Q);
        int t = (int)((float)R * rq);                                       // within a few units of the floor
        int r = R - t * Q;                                                  // |t*Q| < 2^26: no overflow
        const int c = (int)((float)r * rq);                                 // |r| is a few Q at most
        t += c;
        r -= c * Q;                                                         // now within one Q of [0, Q)
        if (r < 0) { --t; r += Q; }
        if (r >= Q) { ++t; r -= Q; }
        const int lim = n - i - 1;
        t = (t > lim) ? lim : t;
        int len = (d1 == 0.0f) ? (n - i) : (t + 1);                        // increment absorbed: constant from here on
        len = reg ? len : 1;
        const float d = reg ? d1 : 0.0f;
        emit(ns, i, p, d);
        ++ns;
        // exact: (len-1)*|d| <= room and the sum lies on the binade's grid; len == 1 gives p1 again
        p = phase_step(p + (float)(len - 1) * d, inc);
        i += len;
    }
    *covered = i;
    *p_next = p;
    return ns;
}

// Array form: seg[k] = {start, base, step}.
UM_FN int phase_table_build(float p0, float inc, int n, PhaseSeg* seg, int cap, int* covered, float* p_next) {
    return phase_table_walk(p0, inc, n, cap, covered, p_next, [seg](int k, int start, float base, float step) {
        seg[k].start = start; seg[k].base = base; seg[k].step = step;
    });
}

// phase at position i (seg = the segment with start <= i < next start); both operations are exact
UM_FN float phase_table_eval(const PhaseSeg& s, int i) {
    return s.base + (float)(i - s.start) * s.step;
}

}  // namespace um
#endif
