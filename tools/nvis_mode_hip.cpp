// nvis_mode_hip.cpp — the Monte-Carlo loop of the reference's tools/test_nvis_mode.cpp:35-114,169-232
// (eight NVIS modes x `--trials` frames at `--snr` dB AWGN, success %) as ONE batch per mode through the
// C-ABI: stimulus (payload -> LDPC encode -> preamble + modulate -> 0.5 peak -> AWGN), post-sync
// demodulation, LDPC decode and the error counters all run on the GPU; the host only prints.
//
//   g++ -O2 -std=c++20 -Iinclude tools/nvis_mode_hip.cpp -Lprojectultra_amd -lultra_hip \
//       -Wl,-rpath,$PWD/projectultra_amd -o nvis_mode_hip && ./nvis_mode_hip --snr 30 --trials 65536
//
// Differences from the reference tool, by construction of the batch path: genie timing (the frame enters at
// its first data symbol; the Schmidl-Cox search is ultra_hip_receive_batch's business), counter-based
// payload / noise generators instead of one serial mt19937 stream.  Machine-readable lines start with "MODE".
#include "ultra_hip_waveform.hpp"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

using namespace ultra_hip;

static void must(int rc, const char* what) {
    if (rc != ULTRA_HIP_OK) { std::fprintf(stderr, "%s: %s\n", what, ultra_hip_strerror(rc)); std::exit(1); }
}

int main(int argc, char** argv) {
    float snr_db = 30.0f;
    size_t trials = 65536;
    unsigned long long seed = 42;
    for (int i = 1; i + 1 < argc; i += 2) {
        if (!std::strcmp(argv[i], "--snr")) snr_db = std::stof(argv[i + 1]);
        else if (!std::strcmp(argv[i], "--trials")) trials = std::stoul(argv[i + 1]);
        else if (!std::strcmp(argv[i], "--seed")) seed = std::stoull(argv[i + 1]);
    }
    struct Mode { Modulation mod; CodeRate rate; bool pilots; const char* name; };
    const Mode modes[] = {                                      // tools/test_nvis_mode.cpp:172-185
        {Modulation::DQPSK, CodeRate::R1_2, false, "DQPSK R1/2"}, {Modulation::DQPSK, CodeRate::R3_4, false, "DQPSK R3/4"},
        {Modulation::D8PSK, CodeRate::R1_2, false, "D8PSK R1/2"}, {Modulation::D8PSK, CodeRate::R3_4, false, "D8PSK R3/4"},
        {Modulation::QAM16, CodeRate::R1_2, true, "16QAM R1/2"},  {Modulation::QAM16, CodeRate::R3_4, true, "16QAM R3/4"},
        {Modulation::QAM32, CodeRate::R1_2, true, "32QAM R1/2"},  {Modulation::QAM32, CodeRate::R3_4, true, "32QAM R3/4"},
    };
    std::printf("NVIS modes, %zu trials per mode, AWGN %.1f dB, seed %llu\n", trials, snr_db, seed);
    std::printf("%-14s %9s %10s %12s %10s %12s\n", "Mode", "Carriers", "Success %", "BP iters", "BER", "frames/s");
    for (const Mode& m : modes) {
        ModemConfig c;                                          // presets::nvis_mode(), types.hpp:342-355
        c.fft_size = 1024; c.num_carriers = 59; c.cp_mode = CyclicPrefixMode::MEDIUM; c.symbol_guard = 0;
        c.modulation = m.mod; c.code_rate = m.rate; c.use_pilots = m.pilots; c.pilot_spacing = m.pilots ? 4 : 2;
        ultra_hip_config probe = to_c_config(c, ULTRA_ENTRY_SYNCED, 1, 0);
        ultra_hip_geometry g;
        must(ultra_hip_geometry_for(&probe, &g), "geometry");
        const uint32_t n_sym = (648 + g.llrs_per_symbol - 1) / g.llrs_per_symbol;
        const ultra_hip_config cfg = to_c_config(c, ULTRA_ENTRY_SYNCED, n_sym, 0);
        ultra_hip_ctx* ctx = nullptr;
        must(ultra_hip_create(&cfg, 0, nullptr, &ctx), "create");
        must(ultra_hip_get_geometry(ctx, &g), "geometry");
        const size_t payload_bytes = g.ldpc_k / 8;
        void *d_audio, *d_payload, *d_bytes, *d_iters, *d_ok, *d_cnt;
        must(ultra_hip_malloc(ctx, trials * g.frame_samples * sizeof(float), &d_audio), "malloc");
        must(ultra_hip_malloc(ctx, trials * payload_bytes, &d_payload), "malloc");
        must(ultra_hip_malloc(ctx, trials * g.decoded_bytes, &d_bytes), "malloc");
        must(ultra_hip_malloc(ctx, trials * sizeof(int32_t), &d_iters), "malloc");
        must(ultra_hip_malloc(ctx, trials, &d_ok), "malloc");
        must(ultra_hip_malloc(ctx, sizeof(ultra_hip_counters), &d_cnt), "malloc");
        must(ultra_hip_memset(ctx, d_cnt, 0, sizeof(ultra_hip_counters)), "memset");
        must(ultra_hip_make_batch(ctx, seed, 0, trials, 1 /* AWGN */, snr_db, 0.0f, 0.0f, static_cast<float*>(d_audio),
                                  g.frame_samples, static_cast<uint8_t*>(d_payload)), "make_batch");
        // one untimed pass first: the first launch of a kernel instance pays for loading it
        must(ultra_hip_demod_decode_batch(ctx, static_cast<const float*>(d_audio), g.frame_samples, nullptr, nullptr, trials,
                                          nullptr, static_cast<uint8_t*>(d_bytes), static_cast<int32_t*>(d_iters),
                                          static_cast<uint8_t*>(d_ok)), "demod_decode_batch");
        must(ultra_hip_synchronize(ctx), "sync");
        must(ultra_hip_timer_begin(ctx), "timer");
        must(ultra_hip_demod_decode_batch(ctx, static_cast<const float*>(d_audio), g.frame_samples, nullptr, nullptr, trials,
                                          nullptr, static_cast<uint8_t*>(d_bytes), static_cast<int32_t*>(d_iters),
                                          static_cast<uint8_t*>(d_ok)), "demod_decode_batch");
        must(ultra_hip_count_errors(ctx, static_cast<const uint8_t*>(d_bytes), static_cast<const int32_t*>(d_iters),
                                    static_cast<const uint8_t*>(d_ok), static_cast<const uint8_t*>(d_payload), payload_bytes,
                                    trials, static_cast<ultra_hip_counters*>(d_cnt)), "count_errors");
        float ms = 0.0f;
        must(ultra_hip_timer_end(ctx, &ms), "timer");
        ultra_hip_counters t;
        must(ultra_hip_memcpy_d2h(ctx, &t, d_cnt, sizeof(t)), "d2h");
        const double ok_pct = 100.0 * double(t.frames - t.frame_errors) / double(t.frames);
        std::printf("%-14s %9u %10.2f %12.2f %10.2e %12.0f\n", m.name, g.n_data_carriers, ok_pct,
                    double(t.iters_sum) / double(t.frames), double(t.bit_errors) / double(t.info_bits), trials / (ms * 1e-3));
        std::printf("MODE %u %u %llu %llu %llu %llu %llu\n", unsigned(m.mod), unsigned(m.rate), (unsigned long long)t.frames,
                    (unsigned long long)t.frame_errors, (unsigned long long)t.bit_errors, (unsigned long long)t.ldpc_fail,
                    (unsigned long long)t.iters_sum);
        for (void* p : {d_audio, d_payload, d_bytes, d_iters, d_ok, d_cnt}) ultra_hip_free(ctx, p);
        ultra_hip_destroy(ctx);
    }
    return 0;
}
