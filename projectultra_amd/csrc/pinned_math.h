// pinned_math.h — bit-exact device restatement of the libm calls on the
// reference's receive path.
//
// The reference (secup/ProjectUltra) calls, through std::cos/sin/abs/arg/exp on
// float and std::complex<float>:
//     cosf, sinf, sincosf (via cexpf), atan2f (-> atanf), hypotf (via cabsf).
// These live in a third-party dependency that is not part of /root/reference:
// GNU libc 2.35 libm (Ubuntu GLIBC 2.35-0ubuntu3.11, the libm.so.6 of this
// image and of the GPU box).  To give bit-identical LLRs the HIP kernels must
// reproduce those functions bit for bit, so this header restates their
// published algorithms:
//   * sinf/cosf/sincosf: ARM optimized-routines single-precision sin/cos as
//     adopted by glibc >= 2.28 (sysdeps/ieee754/flt-32/s_sincosf.h): reduce in
//     double, degree-7/8 polynomials in double, ONE rounding to float.  On
//     x86-64 CPUs with FMA glibc dispatches the `__sinf_fma` build, whose
//     multiply-adds are fused; the fma() placement below follows that build
//     instruction for instruction (checked against the disassembly of this
//     image's libm and exhaustively against libm over all 2^32 inputs:
//     tests/test_pinned_math.py, tools/pinned_math_check.cpp).
//   * atan2f/atanf: fdlibm float code (sysdeps/ieee754/flt-32/e_atan2f.c,
//     s_atanf.c), pure float arithmetic, no FMA.
//   * hypotf: glibc 2.35 e_hypotf.c — (float)sqrt((double)x*x + (double)y*y).
// Constants were read from the image's libm.so.6 .rodata and agree with the
// published sources.
//
// The functions are plain C++ (no HIP types) so the same text compiles for the
// device (hipcc) and for the host exhaustive check (g++).  Compile with
// -ffp-contract=off: every fused operation is written as an explicit fma().
#ifndef ULTRA_PINNED_MATH_H
#define ULTRA_PINNED_MATH_H

#include <stdint.h>
#include <math.h>

#ifdef __HIPCC__
#define UM_FN __host__ __device__ __forceinline__
#else
#define UM_FN static inline
#endif

namespace um {

UM_FN uint32_t as_u32(float f) { union { float f; uint32_t u; } v; v.f = f; return v.u; }
UM_FN float as_f32(uint32_t u) { union { float f; uint32_t u; } v; v.u = u; return v.f; }
UM_FN uint32_t abstop12(float x) { return (as_u32(x) >> 20) & 0x7ff; }

// __sincosf_table[0] / [1] (s_sincosf_data.c): cosine c0..c4, sine s1..s3.
struct SinCosTab { double c0, c1, c2, c3, c4, s1, s2, s3; };

#define UM_TAB0 { 0x1p0, -0x1.ffffffd0c621cp-2, 0x1.55553e1068f19p-5, -0x1.6c087e89a359dp-10, \
                  0x1.99343027bf8c3p-16, -0x1.555545995a603p-3, 0x1.1107605230bc4p-7, -0x1.994eb3774cf24p-13 }
#define UM_TAB1 { -0x1p0, 0x1.ffffffd0c621cp-2, -0x1.55553e1068f19p-5, 0x1.6c087e89a359dp-10, \
                  -0x1.99343027bf8c3p-16, -0x1.555545995a603p-3, 0x1.1107605230bc4p-7, -0x1.994eb3774cf24p-13 }

// sinf_poly (s_sincosf.h), FMA build: n even -> sine polynomial, n odd -> cosine.
UM_FN float sinf_poly(double x, double x2, const SinCosTab& p, int n) {
    if ((n & 1) == 0) {
        double x3 = x * x2;
        double s1 = fma(x2, p.s3, p.s2);
        double x7 = x3 * x2;
        double s = fma(x3, p.s1, x);
        return (float)fma(x7, s1, s);
    } else {
        double x4 = x2 * x2;
        double c2 = fma(x2, p.c4, p.c3);
        double c1 = fma(x2, p.c1, p.c0);
        double x6 = x4 * x2;
        double c = fma(x4, p.c2, c1);
        return (float)fma(x6, c2, c);
    }
}

// reduce_fast (|x| < 120): hpi_inv prescaled by 2^24 (TOINT_INTRINSICS == 0 on x86-64).
UM_FN double reduce_fast(double x, int* np) {
    double r = x * 0x1.45F306DC9C883p+23;
    int n = ((int32_t)r + 0x800000) >> 24;
    *np = n;
    return fma(-(double)n, 0x1.921FB54442D18p0, x);
}

// reduce_large (|x| >= 120), __inv_pio4 table.
UM_FN double reduce_large(uint32_t xi, int* np) {
    const uint32_t inv_pio4[24] = {
        0xa2, 0xa2f9, 0xa2f983, 0xa2f9836e, 0xf9836e4e, 0x836e4e44, 0x6e4e4415, 0x4e441529,
        0x441529fc, 0x1529fc27, 0x29fc2757, 0xfc2757d1, 0x2757d1f5, 0x57d1f534, 0xd1f534dd, 0xf534ddc0,
        0x34ddc0db, 0xddc0db62, 0xc0db6295, 0xdb629599, 0x6295993c, 0x95993c43, 0x993c4390, 0x3c439041};
    const uint32_t* arr = &inv_pio4[(xi >> 26) & 15];
    int shift = (xi >> 23) & 7;
    uint64_t n, res0, res1, res2;
    xi = (xi & 0xffffff) | 0x800000;
    xi <<= shift;
    res0 = (uint32_t)(xi * arr[0]);
    res1 = (uint64_t)xi * arr[4];
    res2 = (uint64_t)xi * arr[8];
    res0 = (res2 >> 32) | (res0 << 32);
    res0 += res1;
    n = (res0 + (1ULL << 61)) >> 62;
    res0 -= n << 62;
    double x = (double)(int64_t)res0;
    *np = (int)n;
    return x * 0x1.921FB54442D18p-62;
}

UM_FN double quadrant_sign(int n) { return ((n + 1) & 2) ? -1.0 : 1.0; }  // {1,-1,-1,1}[n&3]

// glibc sinf (s_sinf.c)
UM_FN float sinf_(float y) {
    const SinCosTab t0 = UM_TAB0, t1 = UM_TAB1;
    double x = y;
    int n;
    if (abstop12(y) < 0x3f4) {  // |y| < pi/4   (abstop12(pio4) == 0x3f4)
        double s = x * x;
        if (abstop12(y) < 0x398) return y;  // |y| < 2^-12
        return sinf_poly(x, s, t0, 0);
    } else if (abstop12(y) < 0x42f) {  // |y| < 120
        x = reduce_fast(x, &n);
        double s = quadrant_sign(n);
        return sinf_poly(x * s, x * x, (n & 2) ? t1 : t0, n);
    } else if (abstop12(y) < 0x7f8) {
        uint32_t xi = as_u32(y);
        int sign = (int)(xi >> 31);
        x = reduce_large(xi, &n);
        double s = quadrant_sign(n + sign);
        return sinf_poly(x * s, x * x, ((n + sign) & 2) ? t1 : t0, n);
    }
    return y - y;  // inf/NaN -> NaN
}

// glibc cosf (s_cosf.c)
UM_FN float cosf_(float y) {
    const SinCosTab t0 = UM_TAB0, t1 = UM_TAB1;
    double x = y;
    int n;
    if (abstop12(y) < 0x3f4) {
        double x2 = x * x;
        if (abstop12(y) < 0x398) return 1.0f;
        return sinf_poly(x, x2, t0, 1);
    } else if (abstop12(y) < 0x42f) {
        x = reduce_fast(x, &n);
        double s = quadrant_sign(n);
        return sinf_poly(x * s, x * x, (n & 2) ? t1 : t0, n ^ 1);
    } else if (abstop12(y) < 0x7f8) {
        uint32_t xi = as_u32(y);
        int sign = (int)(xi >> 31);
        x = reduce_large(xi, &n);
        double s = quadrant_sign(n + sign);
        return sinf_poly(x * s, x * x, ((n + sign) & 2) ? t1 : t0, n ^ 1);
    }
    return y - y;
}

// glibc sincosf (s_sincosf.c, sincosf_poly): ONE reduction, the sine polynomial of (x*s, x2) and
// the cosine polynomial of x2 (which does not depend on the sign s), swapped when the quadrant is
// odd.  Bit-identical to (sinf_(y), cosf_(y)) — the separate functions evaluate exactly these two
// polynomials, one each — but branch-free and half the work; cexpf(0 + i*t) = (cos t, sin t)
// goes through it (s_cexp_template.c).
UM_FN void sincosf_(float y, float* sp, float* cp) {
    const SinCosTab t0 = UM_TAB0, t1 = UM_TAB1;
    double x = y, s = 1.0;
    int n = 0;
    bool second = false;
    const uint32_t top = abstop12(y);
    if (top < 0x3f4) {                       // |y| < pi/4
        if (top < 0x398) { *sp = y; *cp = 1.0f; return; }   // |y| < 2^-12
    } else if (top < 0x42f) {                // |y| < 120
        x = reduce_fast(x, &n);
        s = quadrant_sign(n);
        second = (n & 2) != 0;
    } else if (top < 0x7f8) {
        const uint32_t xi = as_u32(y);
        const int sign = (int)(xi >> 31);
        x = reduce_large(xi, &n);
        s = quadrant_sign(n + sign);         // sign and table include the input's sign,
        second = ((n + sign) & 2) != 0;      // the sin/cos swap below uses n alone (s_sincosf.c)
    } else {
        *sp = y - y; *cp = y - y; return;    // inf/NaN -> NaN
    }
    const SinCosTab& p = second ? t1 : t0;
    const double x2 = x * x, xs = x * s;
    // sine polynomial (sinf_poly, n even)
    const double x3 = xs * x2;
    const double s1 = fma(x2, p.s3, p.s2);
    const double x7 = x3 * x2;
    const double sa = fma(x3, p.s1, xs);
    const float sine = (float)fma(x7, s1, sa);
    // cosine polynomial (sinf_poly, n odd)
    const double x4 = x2 * x2;
    const double c2 = fma(x2, p.c4, p.c3);
    const double c1 = fma(x2, p.c1, p.c0);
    const double x6 = x4 * x2;
    const double ca = fma(x4, p.c2, c1);
    const float cosine = (float)fma(x6, c2, ca);
    if (n & 1) { *sp = cosine; *cp = sine; } else { *sp = sine; *cp = cosine; }
}

// sincosf_ restricted to |y| < 120 (abstop12(y) < 0x42f), without branches.  Same values:
//   * |y| < pi/4: glibc skips the reduction; reduce_fast yields n = 0 and fma(-0.0, hpi, x) = x,
//     so the shared path computes exactly what the short path does;
//   * table 1 is table 0 with the cosine coefficients negated (sine coefficients equal), and every
//     operation of the cosine polynomial is odd in its coefficients, so "table 1" = negate the
//     cosine of table 0; multiplying x by the quadrant sign +-1.0 = flipping its sign bit;
//   * |y| < 2^-12 returns (y, 1.0f) as glibc does (the polynomial would round to the same cosine,
//     but not to the same sine for y = -0.0).
// Checked against libm over every float of the range (tools/pinned_math_check.cpp).
UM_FN void sincosf_bounded_(float y, float* sp, float* cp) {
    const SinCosTab p = UM_TAB0;
    int n;
    const double x = reduce_fast((double)y, &n);
    const double xs = ((n + 1) & 2) ? -x : x;
    const double x2 = x * x;
    const double x3 = xs * x2;
    const double s1 = fma(x2, p.s3, p.s2);
    const double x7 = x3 * x2;
    const double sa = fma(x3, p.s1, xs);
    float sine = (float)fma(x7, s1, sa);
    const double x4 = x2 * x2;
    const double c2 = fma(x2, p.c4, p.c3);
    const double c1 = fma(x2, p.c1, p.c0);
    const double x6 = x4 * x2;
    const double ca = fma(x4, p.c2, c1);
    float cosine = (float)fma(x6, c2, ca);
    if (n & 2) cosine = -cosine;
    if (n & 1) { const float t = sine; sine = cosine; cosine = t; }
    if (abstop12(y) < 0x398) { sine = y; cosine = 1.0f; }
    *sp = sine;
    *cp = cosine;
}

// sincosf_bounded_ for a caller that KNOWS the quadrant: n = the n reduce_fast gives for y (the caller's y values share it:
// reduce_fast is monotone in y, so n(min) == n(max) settles a whole set), and y is not -0.0.  With n wave-uniform every
// per-lane select of sincosf_bounded_ becomes a scalar choice:
//   * x = fma(-n, hpi, y), then xs = +-x by the quadrant:  fma(-n s, hpi, s y) with s = +-1 is that value exactly (negating
//     product and addend negates the exact sum, rounding to nearest is symmetric; s y is exact);
//   * the cosine's sign for n & 2 is a multiplication by +-1.0f (exact), the swap for n & 1 is the caller's choice of
//     which result goes where (SWAP);
//   * glibc's |y| < 2^-12 shortcut (y, 1.0f) is what the polynomials round to anyway (|y^3 / 6| and y^2 / 2 are below half
//     an ulp of y and of 1) — for every such y but -0.0, whose sine the polynomial turns into +0.0: the caller keeps -0.0 out.
// The same fused operations on the same operands in the same order otherwise.  tools/pinned_math_check.cpp ("quadrant")
// compares it with sincosf_bounded_ and glibc over every float of the range.
template <bool SWAP>
UM_FN void sincosf_quadrant_(float y, double neg_n_signed, float y_sign, float cos_sign, float* sp, float* cp) {
    const SinCosTab p = UM_TAB0;
    const double xs = fma(neg_n_signed, 0x1.921FB54442D18p0, (double)(y * y_sign));
    const double x2 = xs * xs;
    const double x3 = xs * x2;
    const double s1 = fma(x2, p.s3, p.s2);
    const double x7 = x3 * x2;
    const double sa = fma(x3, p.s1, xs);
    const float sine = (float)fma(x7, s1, sa);
    const double x4 = x2 * x2;
    const double c2 = fma(x2, p.c4, p.c3);
    const double c1 = fma(x2, p.c1, p.c0);
    const double x6 = x4 * x2;
    const double ca = fma(x4, p.c2, c1);
    const float cosine = (float)fma(x6, c2, ca) * cos_sign;
    *sp = SWAP ? cosine : sine;
    *cp = SWAP ? sine : cosine;
}
// ... and with the quadrant's signs as compile-time constants (Q = n & 3): the two multiplications by +-1.0f fold into the
// sign modifiers of the conversions (x * -1.0f == -x, (float)(-d) == -(float)d: exact) — 17 instead of 19 operations per
// evaluation.  The caller switches once per wavefront-item on n & 3 and still passes neg_n_signed (it holds n itself).
template <int Q>
UM_FN void sincosf_quadrant_q(float y, double neg_n_signed, float* sp, float* cp) {
    constexpr bool flip = ((Q + 1) & 2) != 0;
    sincosf_quadrant_<(Q & 1) != 0>(y, neg_n_signed, flip ? -1.0f : 1.0f, (Q & 2) ? -1.0f : 1.0f, sp, cp);
}
// the scalars of sincosf_quadrant_ for quadrant n
UM_FN void sincosf_quadrant_setup(int n, double* neg_n_signed, float* y_sign, float* cos_sign, bool* swap) {
    const bool flip = ((n + 1) & 2) != 0;
    *y_sign = flip ? -1.0f : 1.0f;
    *neg_n_signed = flip ? (double)n : -(double)n;
    *cos_sign = (n & 2) ? -1.0f : 1.0f;
    *swap = (n & 1) != 0;
}

// fdlibm atanf (s_atanf.c)
UM_FN float atanf_(float x) {
    // atanhi / atanlo as selects (a dynamically indexed local array would live in memory on the GPU)
    const float hi0 = as_f32(0x3eed6338), hi1 = as_f32(0x3f490fda), hi2 = as_f32(0x3f7b985e), hi3 = as_f32(0x3fc90fda);
    const float lo0 = as_f32(0x31ac3769), lo1 = as_f32(0x33222168), lo2 = as_f32(0x33140fb4), lo3 = as_f32(0x33a22168);
    const float aT0 = as_f32(0x3eaaaaab), aT1 = as_f32(0xbe4ccccd), aT2 = as_f32(0x3e124925),
                aT3 = as_f32(0xbde38e38), aT4 = as_f32(0x3dba2e6e), aT5 = as_f32(0xbd9d8795),
                aT6 = as_f32(0x3d886b35), aT7 = as_f32(0xbd6ef16b), aT8 = as_f32(0x3d4bda59),
                aT9 = as_f32(0xbd15a221), aT10 = as_f32(0x3c8569d7);
    // Same operations on the same operands as s_atanf.c, written without divergent branches: the four
    // argument reductions are four (numerator, denominator) pairs and ONE correctly rounded division
    // (|x| < 0.4375 divides x by 1.0f, which is exact), the range cases are selects.  A wavefront
    // whose lanes fall into different ranges then runs one division instead of up to four.
    const int32_t hx = (int32_t)as_u32(x);
    const int32_t ix = hx & 0x7fffffff;
    const float ax = as_f32((uint32_t)ix);
    const bool r_small = ix < 0x3ee00000;                  // |x| < 0.4375        id = -1
    const bool r0 = ix < 0x3f300000;                       // |x| < 11/16         id = 0
    const bool r1 = ix < 0x3f980000;                       // |x| < 19/16         id = 1
    const bool r2 = ix < 0x401c0000;                       // |x| < 39/16         id = 2, else id = 3
    const float num = r_small ? x : r0 ? (2.0f * ax - 1.0f) : r1 ? (ax - 1.0f) : r2 ? (ax - 1.5f) : -1.0f;
    const float den = r_small ? 1.0f : r0 ? (2.0f + ax) : r1 ? (ax + 1.0f) : r2 ? (1.0f + 1.5f * ax) : ax;
    const float xr = num / den;
    const float z = xr * xr;
    const float w = z * z;
    const float s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
    const float s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
    const float ahi = r0 ? hi0 : r1 ? hi1 : r2 ? hi2 : hi3;
    const float alo = r0 ? lo0 : r1 ? lo1 : r2 ? lo2 : lo3;
    const float zb = ahi - ((xr * (s1 + s2) - alo) - xr);
    float r = r_small ? (xr - xr * (s1 + s2)) : ((hx < 0) ? -zb : zb);
    if (ix < 0x31000000) r = x;                            // |x| < 2^-29 (huge + x > one always holds)
    if (ix >= 0x4c000000)                                  // |x| >= 2^25, inf, NaN
        r = (ix > 0x7f800000) ? (x + x) : ((hx > 0) ? (hi3 + lo3) : (-hi3 - lo3));
    return r;
}

// fdlibm __ieee754_atan2f (e_atan2f.c); the errno wrapper adds nothing numerically.
UM_FN float atan2f_(float y, float x) {
    const float tiny = 1.0e-30f;
    const float pi_o_4 = as_f32(0x3f490fdb), pi_o_2 = as_f32(0x3fc90fdb), pi = as_f32(0x40490fdb),
                pi_lo = as_f32(0xb3bbbd2e);
    int32_t hx = (int32_t)as_u32(x), hy = (int32_t)as_u32(y);
    int32_t ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
    if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;
    if (hx == 0x3f800000) return atanf_(y);
    int32_t m = ((hy >> 31) & 1) | ((hx >> 30) & 2);
    if (iy == 0) {
        switch (m) {
            case 0: case 1: return y;
            case 2: return pi + tiny;
            default: return -pi - tiny;
        }
    }
    if (ix == 0) return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;
    if (ix == 0x7f800000) {
        if (iy == 0x7f800000) {
            switch (m) {
                case 0: return pi_o_4 + tiny;
                case 1: return -pi_o_4 - tiny;
                case 2: return 3.0f * pi_o_4 + tiny;
                default: return -3.0f * pi_o_4 - tiny;
            }
        } else {
            switch (m) {
                case 0: return 0.0f;
                case 1: return -0.0f;
                case 2: return pi + tiny;
                default: return -pi - tiny;
            }
        }
    }
    if (iy == 0x7f800000) return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;
    int32_t k = (iy - ix) >> 23;
    float z;
    if (k > 60) z = pi_o_2 + 0.5f * pi_lo;
    else if (hx < 0 && k < -60) z = 0.0f;
    else z = atanf_(fabsf(y / x));
    switch (m) {
        case 0: return z;
        case 1: return as_f32(as_u32(z) ^ 0x80000000u);
        case 2: return pi - (z - pi_lo);
        default: return (z - pi_lo) - pi;
    }
}

// glibc 2.35 hypotf (e_hypotf.c): exact products, one double rounding of the
// sum, correctly rounded sqrt, one narrowing.
UM_FN float hypotf_(float x, float y) {
    uint32_t ax = as_u32(x) & 0x7fffffff, ay = as_u32(y) & 0x7fffffff;
    if (ax >= 0x7f800000 || ay >= 0x7f800000) {
        if (ax == 0x7f800000 || ay == 0x7f800000) return as_f32(0x7f800000);
        return x + y;
    }
    double dx = x, dy = y;
    return (float)sqrt(dx * dx + dy * dy);
}

// glibc 2.35 logf (sysdeps/ieee754/flt-32/e_logf.c = ARM optimized-routines logf: 16-entry table of (1/c, log c),
// degree-3 polynomial in double, one rounding to float), the FMA build x86-64 dispatches (`__logf_fma`; operation
// order read from the disassembly of this image's libm.so.6, table and coefficients from its .rodata — they agree
// with the published logf_data.c).  NOT on the receive path: the Monte-Carlo stimulus generators' Box-Muller
// transform uses it (stimulus_kernel.h, ultra_hip_make_llr_batch) so that the test oracle, which calls libm's logf,
// reproduces the device's noise bit for bit.  Checked against libm over every float (tools/pinned_math_check.cpp).
UM_FN float logf_(float x) {
    const double invc[16] = {0x1.661ec79f8f3bep+0, 0x1.571ed4aaf883dp+0, 0x1.49539f0f010bp+0, 0x1.3c995b0b80385p+0,
                             0x1.30d190c8864a5p+0, 0x1.25e227b0b8eap+0, 0x1.1bb4a4a1a343fp+0, 0x1.12358f08ae5bap+0,
                             0x1.0953f419900a7p+0, 0x1p+0, 0x1.e608cfd9a47acp-1, 0x1.ca4b31f026aap-1,
                             0x1.b2036576afce6p-1, 0x1.9c2d163a1aa2dp-1, 0x1.886e6037841edp-1, 0x1.767dcf5534862p-1};
    const double logc[16] = {-0x1.57bf7808caadep-2, -0x1.2bef0a7c06ddbp-2, -0x1.01eae7f513a67p-2, -0x1.b31d8a68224e9p-3,
                             -0x1.6574f0ac07758p-3, -0x1.1aa2bc79c81p-3, -0x1.a4e76ce8c0e5ep-4, -0x1.1973c5a611cccp-4,
                             -0x1.252f438e10c1ep-5, 0x0p+0, 0x1.aa5aa5df25984p-5, 0x1.c5e53aa362eb4p-4,
                             0x1.526e57720db08p-3, 0x1.bc2860d22477p-3, 0x1.1058bc8a07ee1p-2, 0x1.4043057b6ee09p-2};
    const double ln2 = 0x1.62e42fefa39efp-1;
    const double A0 = -0x1.00ea348b88334p-2, A1 = 0x1.5575b0be00b6ap-2, A2 = -0x1.ffffef20a4123p-2;
    uint32_t ix = as_u32(x);
    if (ix == 0x3f800000u) return 0.0f;                          // log(1) = +0 in every rounding mode
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {         // zero, subnormal, negative, inf, NaN
        if (ix * 2u == 0u) return -INFINITY;                     // log(+-0) = -inf (errno aside)
        if (ix == 0x7f800000u) return x;                         // log(inf) = inf
        if ((ix & 0x80000000u) || ix * 2u >= 0xff000000u) return (x - x) / (x - x);   // negative or NaN -> NaN
        ix = as_u32(x * 0x1p23f);                                // subnormal: normalise
        ix -= 23u << 23;
    }
    const uint32_t tmp = ix - 0x3f330000u;
    const int i = (int)((tmp >> 19) & 15u);
    const int k = (int32_t)tmp >> 23;
    const uint32_t iz = ix - (tmp & 0xff800000u);
    const double z = (double)as_f32(iz);
    // table lookup as selects (a dynamically indexed local array would live in scratch memory on the GPU)
    double ic = invc[0], lc = logc[0];
    for (int t = 1; t < 16; ++t) { if (i == t) { ic = invc[t]; lc = logc[t]; } }
    const double r = fma(z, ic, -1.0);
    const double y0 = fma((double)k, ln2, lc);
    const double r2 = r * r;
    double y = fma(A1, r, A2);
    y = fma(A0, r2, y);
    y = fma(y, r2, y0 + r);
    return (float)y;
}

// fabsf(atan2f(y, x)) > 1.5708f — the test of interpolateChannel (channel_equalizer.cpp:613-617) — without the
// arctangent in all but a sliver of cases.  1.5708f = 1.57080006... lies 3.7e-6 above pi/2:
//   * x with a clear sign bit (and not NaN): the angle is in [-pi/2, pi/2], atan2f returns at most the float
//     next to pi/2 (1.57079637), the test is false whatever y is (a NaN result compares false as well);
//   * x negative with |x| * 2^16 > |y|: the angle is at least pi/2 + atan(2^-16) = pi/2 + 1.5e-5 from zero,
//     a hundred float steps beyond the constant: true;
//   * everything else (x = -0, |x| tiny against |y|, infinities against infinities, NaN): the arctangent.
UM_FN bool atan2f_beyond_right_angle(float y, float x) {
    const uint32_t ux = as_u32(x);
    if (ux <= 0x7f800000u) return false;
    if ((ux & 0x7fffffffu) <= 0x7f800000u && fabsf(x) * 65536.0f > fabsf(y)) return true;
    return fabsf(atan2f_(y, x)) > 1.5708f;
}

}  // namespace um
#endif
