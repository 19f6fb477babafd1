// tools/phase_table_check.cpp — host check of projectultra_amd/csrc/phase_table.h against the
// serial recurrence of Impl::toBaseband (src/ofdm/channel_equalizer.cpp:43-50), position by position.
//   g++ -O2 -std=c++17 -ffp-contract=off -pthread tools/phase_table_check.cpp -o /tmp/ptc && /tmp/ptc [cases]
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include <atomic>
#include "../projectultra_amd/csrc/phase_table.h"

static inline uint64_t splitmix(uint64_t& s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31);
}
static inline double u01(uint64_t& s) { return (splitmix(s) >> 11) * (1.0 / 9007199254740992.0); }

static bool check_case(float p0, float inc, int n, int cap, long* segs_out) {
    std::vector<float> ref(n + 1);
    float p = p0;
    for (int i = 0; i < n; ++i) { ref[i] = p; p = um::phase_step(p, inc); }
    ref[n] = p;
    std::vector<um::PhaseSeg> seg(cap);
    int done = 0; float cur = p0; long nseg_total = 0;
    while (done < n) {                      // multi-round, as the kernel does when the table fills
        int covered; float pn;
        int ns = um::phase_table_build(cur, inc, n - done, seg.data(), cap, &covered, &pn);
        nseg_total += ns;
        if (covered <= 0) return false;
        int s = 0;
        for (int i = 0; i < covered; ++i) {
            while (s + 1 < ns && seg[s + 1].start <= i) ++s;
            float v = um::phase_table_eval(seg[s], i);
            if (memcmp(&v, &ref[done + i], 4) != 0) {
                fprintf(stderr, "MISMATCH p0=%a inc=%a pos=%d got=%a want=%a (seg start %d base %a step %a)\n", p0, inc,
                        done + i, v, ref[done + i], seg[s].start, seg[s].base, seg[s].step);
                return false;
            }
        }
        done += covered; cur = pn;
        if (memcmp(&cur, &ref[done], 4) != 0) { fprintf(stderr, "p_next mismatch p0=%a inc=%a at %d\n", p0, inc, done); return false; }
    }
    *segs_out = nseg_total;
    return true;
}

int main(int argc, char** argv) {
    long cases = argc > 1 ? atol(argv[1]) : 400000;
    unsigned T = std::max(1u, std::thread::hardware_concurrency());
    std::atomic<long> bad{0}, total{0}, segsum{0}, segmax{0}, seg4sum{0}, seg4max{0}, n4{0};
    std::vector<std::thread> th;
    for (unsigned t = 0; t < T; ++t) th.emplace_back([&, t] {
        uint64_t s = 0xC0FFEE + t;
        for (long c = t; c < cases; c += T) {
            int mode = (int)(c % 8);
            float p0, inc;
            int n = 1120;
            double mag = std::exp(std::log(1e-7) + u01(s) * (std::log(0.6) - std::log(1e-7)));
            inc = (float)((u01(s) < 0.5 ? -1 : 1) * mag);
            p0 = (float)((u01(s) * 2 - 1) * 3.3);
            if (mode == 1) p0 = 0.0f;
            if (mode == 2) { int e = (int)(u01(s) * 24) - 26; inc = std::ldexp(1.0f + (float)((int)(u01(s) * 8)) / 8.0f, e); if (u01(s) < .5) inc = -inc; }  // few mantissa bits -> ties
            if (mode == 3) { p0 = (float)((u01(s) < .5 ? -1 : 1) * (3.14159 - u01(s) * 0.01)); }     // wraps
            if (mode == 4) { inc = (float)(-2.0 * M_PI * ((u01(s) * 2 - 1) * 90.0) / 48000.0); }   // realistic CFO range
            if (mode == 5) { p0 = (float)(-mag * (1 + (int)(u01(s) * 40))); inc = (float)mag; }  // zero crossings
            if (mode == 6) { p0 = um::as_f32((uint32_t)splitmix(s)); if (!(std::fabs(p0) < 1e6f)) p0 = 1.0f; n = 300; }
            if (mode == 7) { inc = um::as_f32((uint32_t)(splitmix(s) % 0x00800000u + 1)); p0 = um::as_f32((uint32_t)(splitmix(s) % 0x01000000u)); n = 200; }  // denormals
            long segs = 0;
            bool ok = check_case(p0, inc, n, (c % 3 == 0) ? 4 : 64, &segs);
            total++; if (!ok) bad++;
            if (mode == 4 && c % 3 != 0) { seg4sum += segs; n4++; long m4 = seg4max.load(); while (segs > m4 && !seg4max.compare_exchange_weak(m4, segs)) {} }
            segsum += segs; long m = segmax.load(); while (segs > m && !segmax.compare_exchange_weak(m, segs)) {}
        }
    });
    for (auto& x : th) x.join();
    printf("phase_table checked=%ld mismatches=%ld mean_segments=%.1f max_segments=%ld\n", total.load(), bad.load(),
           (double)segsum.load() / (double)total.load(), segmax.load());
    printf("realistic CFO range (|cfo| <= 90 Hz, random start): mean_segments=%.1f max_segments=%ld\n",
           (double)seg4sum.load() / (double)std::max(1L, n4.load()), seg4max.load());
    return bad.load() ? 1 : 0;
}
