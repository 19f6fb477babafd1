// tools/pinned_math_check.cpp — host check of projectultra_amd/csrc/pinned_math.h
// against the libm of this machine (the functions the reference binary calls).
//   g++ -O2 -std=c++17 -ffp-contract=off -mfma -pthread tools/pinned_math_check.cpp -o /tmp/pmc -lm
//   /tmp/pmc full      # all 2^32 floats for sinf/cosf/sincosf/atanf/logf, 2^31 random pairs atan2f/hypotf
//   /tmp/pmc quick     # strided subset (used by tests/test_pinned_math.py)
//   /tmp/pmc pairs     # only the two-argument functions and the right-angle test, 2^31 pairs
// Prints one line per function: "<name> checked=<n> mismatches=<m>".  Exit code 1 on any mismatch.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>
#include <atomic>
#include "../projectultra_amd/csrc/pinned_math.h"

static inline bool same(float a, float b) {
    uint32_t x, y; memcpy(&x, &a, 4); memcpy(&y, &b, 4);
    if (x == y) return true;
    return (a != a) && (b != b);  // any NaN == any NaN
}
static inline uint64_t splitmix(uint64_t& s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31);
}

int main(int argc, char** argv) {
    bool full = argc > 1 && !strcmp(argv[1], "full");
    const bool pairs_only = argc > 1 && !strcmp(argv[1], "pairs");   // 2^31 pairs, no single-argument sweep
    const unsigned T = std::max(1u, std::thread::hardware_concurrency());
    const uint64_t stride = full ? 1 : 1021;            // prime stride for the quick subset
    const uint64_t pairs = (full || pairs_only) ? (1ull << 31) : (1ull << 24);
    std::atomic<uint64_t> bad[10]; for (auto& b : bad) b = 0;
    std::atomic<uint64_t> cnt[10]; for (auto& c : cnt) c = 0;
    std::vector<std::thread> th;
    for (unsigned t = 0; t < T; ++t) th.emplace_back([&, t] {
        uint64_t lb[10] = {0}, lc[10] = {0};
        for (uint64_t u = t * stride; !pairs_only && u < (1ull << 32); u += (uint64_t)T * stride) {
            float x = um::as_f32((uint32_t)u);
            float s, c; sincosf(x, &s, &c);
            lc[0]++; if (!same(um::sinf_(x), sinf(x))) { if (lb[0]++ < 3) fprintf(stderr, "sinf %08x\n", (unsigned)u); }
            lc[1]++; if (!same(um::cosf_(x), cosf(x))) { if (lb[1]++ < 3) fprintf(stderr, "cosf %08x\n", (unsigned)u); }
            float ms, mc; um::sincosf_(x, &ms, &mc);
            lc[2]++; if (!same(ms, s) || !same(mc, c)) { if (lb[2]++ < 3) fprintf(stderr, "sincosf %08x\n", (unsigned)u); }
            if (um::abstop12(x) < 0x42f && (uint32_t)u != 0x80000000u) {   // the quadrant variant: |x| < 120, every float but -0.0 (its sine comes out +0.0)
                int n; (void)um::reduce_fast((double)x, &n);
                double m; float ys, csn; bool sw; um::sincosf_quadrant_setup(n, &m, &ys, &csn, &sw);
                float qs, qc;
                if (sw) um::sincosf_quadrant_<true>(x, m, ys, csn, &qs, &qc); else um::sincosf_quadrant_<false>(x, m, ys, csn, &qs, &qc);
                lc[9]++; if (!same(qs, s) || !same(qc, c)) { if (lb[9]++ < 3) fprintf(stderr, "sincosf_quadrant %08x\n", (unsigned)u); }
            }
            if (um::abstop12(x) < 0x42f) {            // the branch-free variant's domain: |x| < 120
                float bs, bc; um::sincosf_bounded_(x, &bs, &bc);
                lc[6]++; if (!same(bs, s) || !same(bc, c)) { if (lb[6]++ < 3) fprintf(stderr, "sincosf_bounded %08x\n", (unsigned)u); }
            }
            lc[3]++; if (!same(um::atanf_(x), atanf(x))) { if (lb[3]++ < 3) fprintf(stderr, "atanf %08x\n", (unsigned)u); }
            lc[8]++; if (!same(um::logf_(x), logf(x))) { if (lb[8]++ < 3) fprintf(stderr, "logf %08x\n", (unsigned)u); }
        }
        uint64_t seed = 0x1234 + t;
        for (uint64_t i = t; i < pairs; i += T) {
            uint64_t r = splitmix(seed);
            uint32_t a = (uint32_t)r, b = (uint32_t)(r >> 32);
            int mode = (int)(i % 4);
            float y, x;
            if (mode == 0) { y = um::as_f32(a); x = um::as_f32(b); }              // any bit patterns
            else if (mode == 1) { y = ((int32_t)a) * 0x1p-27f; x = ((int32_t)b) * 0x1p-27f; }  // moderate magnitudes
            else if (mode == 2) { y = ((int32_t)a) * 0x1p-31f; x = ((int32_t)b) * 0x1p-24f; }
            else { y = um::as_f32((a & 0x807fffffu) | 0x3f000000u); x = um::as_f32((b & 0x807fffffu) | 0x3f800000u); }
            lc[4]++; if (!same(um::atan2f_(y, x), atan2f(y, x))) { if (lb[4]++ < 3) fprintf(stderr, "atan2f %a %a\n", y, x); }
            lc[5]++; if (!same(um::hypotf_(y, x), hypotf(y, x))) { if (lb[5]++ < 3) fprintf(stderr, "hypotf %a %a\n", y, x); }
            // the right-angle test: the same pairs, and pairs on the sliver around -x/y = 2^-16 .. 2^-19 and at x = -0
            float yy = y, xx = x;
            if (i % 8 >= 4) { xx = -fabsf(y) * um::as_f32(0x35800000u + (b >> 6) % 0x04000000u); if (i % 16 >= 12) xx = (b & 1) ? -0.0f : 0.0f; }
            lc[7]++; if (um::atan2f_beyond_right_angle(yy, xx) != (fabsf(atan2f(yy, xx)) > 1.5708f)) { if (lb[7]++ < 3) fprintf(stderr, "right-angle %a %a\n", yy, xx); }
        }
        // atan2f/hypotf special values grid
        if (t == 0) {
            const float sp[] = {0.0f, -0.0f, 1.0f, -1.0f, INFINITY, -INFINITY, NAN, 1e-45f, -1e-45f, 1e30f, -1e30f,
                                1e-30f, 3.4e38f, 0.5f, 2.0f, 1.17549435e-38f};
            for (float y : sp) for (float x : sp) {
                lc[4]++; if (!same(um::atan2f_(y, x), atan2f(y, x))) lb[4]++;
                lc[5]++; if (!same(um::hypotf_(y, x), hypotf(y, x))) lb[5]++;
                lc[7]++; if (um::atan2f_beyond_right_angle(y, x) != (fabsf(atan2f(y, x)) > 1.5708f)) lb[7]++;
            }
        }
        for (int k = 0; k < 10; ++k) { bad[k] += lb[k]; cnt[k] += lc[k]; }
    });
    for (auto& x : th) x.join();
    const char* names[10] = {"sinf", "cosf", "sincosf", "atanf", "atan2f", "hypotf", "sincosf_bounded", "right_angle_test", "logf", "sincosf_quadrant"};
    int rc = 0;
    for (int k = 0; k < 10; ++k) {
        printf("%s checked=%llu mismatches=%llu\n", names[k], (unsigned long long)cnt[k].load(), (unsigned long long)bad[k].load());
        if (bad[k].load()) rc = 1;
    }
    return rc;
}
