// live_call_breakdown.cpp — where the microseconds of ONE live SYNCED process() call go (host side), over the C-ABI alone.
//
// The call HipOfdmDemodulator::demodulate() makes for one arriving symbol (include/ultra_hip_waveform.hpp): stage the chunk
// (ultra_hip_memcpy_h2d_async), the per-symbol launch chain (ultra_hip_demod_stream_batch_eq, answers written into a pinned block),
// post one word behind it (ultra_hip_stream_post), spin on it (ultra_hip_host_wait), copy the answer out.  Each step is timed with
// steady_clock on the calling thread; "wait" is what is left of the GPU's work when the submissions are done.
//
//   g++ -O2 -std=c++20 -Iinclude tools/live_call_breakdown.cpp -Lprojectultra_amd -lultra_hip -Wl,-rpath,$PWD/projectultra_amd \
//       -o build/live_call_breakdown && build/live_call_breakdown [1024|512] [symbols per call]
#include "ultra_hip_waveform.hpp"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

using namespace ultra_hip;
using Clock = std::chrono::steady_clock;

static void must(int rc, const char* what) {
    if (rc != ULTRA_HIP_OK) { std::fprintf(stderr, "%s: %s\n", what, ultra_hip_strerror(rc)); std::exit(1); }
}
static double us(Clock::time_point a, Clock::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); }

int main(int argc, char** argv) {
    const int fft = argc > 1 ? std::atoi(argv[1]) : 1024;
    const uint32_t per_call = argc > 2 ? uint32_t(std::atoi(argv[2])) : 1u;
    ModemConfig c;
    if (fft == 1024) { c.fft_size = 1024; c.num_carriers = 59; c.symbol_guard = 0; c.modulation = Modulation::QAM16; c.code_rate = CodeRate::R3_4; c.use_pilots = true; c.pilot_spacing = 4; }
    else { c.modulation = Modulation::DQPSK; c.code_rate = CodeRate::R1_2; c.use_pilots = false; }
    const ultra_hip_config cfg = to_c_config(c, ULTRA_ENTRY_SYNCED, 251, 0);
    ultra_hip_ctx* ctx = nullptr;
    must(ultra_hip_create(&cfg, 0, nullptr, &ctx), "create");
    ultra_hip_geometry g;
    must(ultra_hip_get_geometry(ctx, &g), "geometry");
    const uint32_t sym = g.symbol_samples;
    void *d_audio, *d_in, *mh, *md;
    must(ultra_hip_malloc(ctx, size_t(per_call) * sym * sizeof(float), &d_audio), "malloc");
    must(ultra_hip_malloc(ctx, 64, &d_in), "malloc");
    must(ultra_hip_host_block(ctx, 256 << 10, &mh, &md), "host_block");
    std::vector<float> audio(size_t(per_call) * sym);
    std::mt19937 rng(1);
    std::normal_distribution<float> n(0.0f, 0.1f);
    for (auto& v : audio) v = n(rng);
    const float cp[3] = {1.5f, 0.0f, 0.0f};
    const size_t n_llr = size_t(per_call) * g.llrs_per_symbol;
    float* d_out = reinterpret_cast<float*>(static_cast<char*>(md) + 256);
    const size_t eq_off = (ULTRA_HIP_STATE_FLOATS + n_llr + 1) & ~size_t(1);
    std::vector<float> answer(eq_off + size_t(per_call) * 2 * ULTRA_HIP_MAX_CARRIERS);
    std::vector<double> t_stage, t_launch, t_post, t_wait, t_copy, t_all;
    uint32_t seq = 0, first = 0;
    for (int it = 0; it < 600; ++it) {
        if (first + per_call > 250) first = 0;
        const auto t0 = Clock::now();
        if (first == 0) must(ultra_hip_memcpy_h2d_async(ctx, d_in, cp, sizeof(cp)), "h2d");
        must(ultra_hip_memcpy_h2d_async(ctx, d_audio, audio.data(), audio.size() * sizeof(float)), "h2d");
        const auto t1 = Clock::now();
        must(ultra_hip_demod_stream_batch_eq(ctx, static_cast<const float*>(d_audio), audio.size(), static_cast<const float*>(d_in),
                                             static_cast<const float*>(d_in) + 1, 1, first, per_call, d_out + ULTRA_HIP_STATE_FLOATS, d_out, d_out + eq_off), "demod_stream");
        const auto t2 = Clock::now();
        must(ultra_hip_stream_post(ctx, static_cast<uint32_t*>(md), ++seq), "post");
        const auto t3 = Clock::now();
        must(ultra_hip_host_wait(ctx, static_cast<const volatile uint32_t*>(mh), seq, 20000), "wait");
        const auto t4 = Clock::now();
        std::memcpy(answer.data(), static_cast<char*>(mh) + 256, answer.size() * sizeof(float));
        const auto t5 = Clock::now();
        first += per_call;
        if (it >= 100) { t_stage.push_back(us(t0, t1)); t_launch.push_back(us(t1, t2)); t_post.push_back(us(t2, t3)); t_wait.push_back(us(t3, t4)); t_copy.push_back(us(t4, t5)); t_all.push_back(us(t0, t5)); }
    }
    auto med = [](std::vector<double>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    std::printf("fft %d, %u symbol(s) per call (%u samples, %zu soft bits): median microseconds over %zu calls\n", fft, per_call, per_call * sym, n_llr, t_all.size());
    std::printf("  stage the chunk (pinned ring + async copy command)   %6.1f\n", med(t_stage));
    std::printf("  launch chain (ultra_hip_demod_stream_batch_eq)        %6.1f\n", med(t_launch));
    std::printf("  post the word                                         %6.1f\n", med(t_post));
    std::printf("  wait for it (what is left of the GPU's work)          %6.1f\n", med(t_wait));
    std::printf("  copy the answer out of the pinned block               %6.1f\n", med(t_copy));
    std::printf("  whole call                                            %6.1f\n", med(t_all));
    ultra_hip_destroy(ctx);
    return 0;
}
