// sweep_hip.cpp — BASELINE.json configs[3] / configs[4] as a native harness: the SNR loop around the trial loop of the
// reference's tools/test_mode_snr.cpp:18-109,126-160 and the mode table of tools/test_nvis_mode.cpp:169-260, at
// Monte-Carlo scale, host side in C++ over the C-ABI only (twin of projectultra_amd/sweep.py + tools/sweep.py: same
// generators, same per-point seeds, same sharding — the counters of every point are identical).
//
//   g++ -O2 -std=c++20 -Iinclude tools/sweep_hip.cpp -Lprojectultra_amd -lultra_hip -Wl,-rpath,$PWD/projectultra_amd \
//       -ldl -pthread -o sweep_hip
//   ./sweep_hip --config cfg4 [--trials 1048576] [--gpus N] [--snr-from -11 --snr-to 30 --snr-step 1]
//   ./sweep_hip --config cfg5 [--trials 12800]   [--gpus N]
//
// Multi-GPU (one process, one host thread and one context per device): the trials of a point are the index range
// [0, n); device g owns [g n / G, (g + 1) n / G), generates exactly those trials in its own HBM, decodes them, counts on
// the device; then ONE ncclAllReduce of the eight uint64 counters per point over the G devices (RCCL bound with dlopen,
// communicators from ncclCommInitAll, the G calls inside one ncclGroupStart/End).  Nothing else crosses GPUs.
// Machine-readable lines: "POINT <label> <snr_db> frames frame_errors bit_errors info_bits ldpc_fail iters_sum undetected".
#include "ultra_hip_waveform.hpp"

#include <dlfcn.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

using namespace ultra_hip;

static void must(int rc, const char* what) {
    if (rc != ULTRA_HIP_OK) { std::fprintf(stderr, "%s: %s\n", what, ultra_hip_strerror(rc)); std::exit(1); }
}
static uint64_t point_seed(uint64_t seed, uint64_t index) { return seed ^ ((index + 1) * 0x9E3779B97F4A7C15ull); }

struct Rccl {                      // the four entry points the harness itself needs (the all-reduce lives in the library)
    int (*CommInitAll)(void**, int, const int*) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    bool load() {
        void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return false;
        CommInitAll = reinterpret_cast<int (*)(void**, int, const int*)>(dlsym(h, "ncclCommInitAll"));
        GroupStart = reinterpret_cast<int (*)()>(dlsym(h, "ncclGroupStart"));
        GroupEnd = reinterpret_cast<int (*)()>(dlsym(h, "ncclGroupEnd"));
        CommDestroy = reinterpret_cast<int (*)(void*)>(dlsym(h, "ncclCommDestroy"));
        return CommInitAll && GroupStart && GroupEnd && CommDestroy;
    }
};

// one device's share of a sweep: context + buffers, reused from point to point
struct Shard {
    int device = 0;
    bool ldpc_only = true;
    ultra_hip_ctx* ctx = nullptr;
    ultra_hip_geometry g{};
    size_t batch = 0;
    void *d_in = nullptr, *d_payload = nullptr, *d_bytes = nullptr, *d_iters = nullptr, *d_ok = nullptr, *d_cnt = nullptr;
    void open(const ultra_hip_config& cfg, int dev, bool ldpc, size_t batch_) {
        device = dev; ldpc_only = ldpc; batch = batch_;
        must(ultra_hip_create(&cfg, dev, nullptr, &ctx), "ultra_hip_create");
        must(ultra_hip_get_geometry(ctx, &g), "geometry");
        const size_t in_floats = ldpc ? 648 : g.frame_samples;
        must(ultra_hip_malloc(ctx, batch * in_floats * sizeof(float), &d_in), "malloc");
        must(ultra_hip_malloc(ctx, batch * (g.ldpc_k / 8), &d_payload), "malloc");
        must(ultra_hip_malloc(ctx, batch * g.decoded_bytes, &d_bytes), "malloc");
        must(ultra_hip_malloc(ctx, batch * sizeof(int32_t), &d_iters), "malloc");
        must(ultra_hip_malloc(ctx, batch, &d_ok), "malloc");
        must(ultra_hip_malloc(ctx, sizeof(ultra_hip_counters), &d_cnt), "malloc");
    }
    void close() {
        if (!ctx) return;
        for (void* p : {d_in, d_payload, d_bytes, d_iters, d_ok, d_cnt}) ultra_hip_free(ctx, p);
        ultra_hip_destroy(ctx); ctx = nullptr;
    }
    // trials [lo, hi) of one point; counters stay on the device (d_cnt), everything queued on the context's stream
    void run(uint64_t lo, uint64_t hi, float snr_db, uint64_t seed, int channel_kind) {
        must(ultra_hip_memset(ctx, d_cnt, 0, sizeof(ultra_hip_counters)), "memset");
        for (uint64_t f0 = lo; f0 < hi; f0 += batch) {
            const size_t n = static_cast<size_t>(std::min<uint64_t>(batch, hi - f0));
            if (ldpc_only) {
                must(ultra_hip_make_llr_batch(ctx, seed, f0, n, snr_db, static_cast<float*>(d_in), static_cast<uint8_t*>(d_payload)), "make_llr_batch");
                must(ultra_hip_ldpc_decode_batch(ctx, static_cast<const float*>(d_in), n, static_cast<uint8_t*>(d_bytes),
                                                 static_cast<int32_t*>(d_iters), static_cast<uint8_t*>(d_ok), nullptr), "ldpc_decode_batch");
            } else {
                must(ultra_hip_make_batch(ctx, seed, f0, n, channel_kind, snr_db, 0.5f, 0.1f, static_cast<float*>(d_in), g.frame_samples,
                                          static_cast<uint8_t*>(d_payload)), "make_batch");
                must(ultra_hip_demod_decode_batch(ctx, static_cast<const float*>(d_in), g.frame_samples, nullptr, nullptr, n, nullptr,
                                                  static_cast<uint8_t*>(d_bytes), static_cast<int32_t*>(d_iters), static_cast<uint8_t*>(d_ok)), "demod_decode_batch");
            }
            must(ultra_hip_count_errors(ctx, static_cast<const uint8_t*>(d_bytes), static_cast<const int32_t*>(d_iters),
                                        static_cast<const uint8_t*>(d_ok), static_cast<const uint8_t*>(d_payload), g.ldpc_k / 8, n,
                                        static_cast<ultra_hip_counters*>(d_cnt)), "count_errors");
        }
    }
};

int main(int argc, char** argv) {
    std::string config = "cfg4";
    uint64_t trials = 0, seed = 0x5EED;
    int gpus = 1, channel_kind = 1;
    double snr_from = 1e9, snr_to = 1e9, snr_step = 0;
    size_t batch = 0;
    unsigned rate_arg = 0;
    for (int i = 1; i + 1 < argc; i += 2) {
        const std::string a = argv[i];
        if (a == "--config") config = argv[i + 1];
        else if (a == "--trials") trials = std::stoull(argv[i + 1]);
        else if (a == "--seed") seed = std::stoull(argv[i + 1], nullptr, 0);
        else if (a == "--gpus") gpus = std::stoi(argv[i + 1]);
        else if (a == "--batch") batch = std::stoull(argv[i + 1]);
        else if (a == "--rate") rate_arg = static_cast<unsigned>(std::stoi(argv[i + 1]));
        else if (a == "--channel") channel_kind = (std::string(argv[i + 1]) == "watterson") ? 2 : 1;
        else if (a == "--snr-from") snr_from = std::stod(argv[i + 1]);
        else if (a == "--snr-to") snr_to = std::stod(argv[i + 1]);
        else if (a == "--snr-step") snr_step = std::stod(argv[i + 1]);
        else { std::fprintf(stderr, "unknown option %s\n", argv[i]); return 2; }
    }
    const bool cfg4 = config == "cfg4";
    if (!cfg4 && config != "cfg5") { std::fprintf(stderr, "--config cfg4|cfg5\n"); return 2; }
    if (!trials) trials = cfg4 ? (1ull << 20) : 12800;
    if (snr_from > 1e8) { snr_from = cfg4 ? -11 : -9; snr_to = cfg4 ? 30 : 21; snr_step = cfg4 ? 1 : 3; }
    if (!batch) batch = cfg4 ? (1u << 20) : (1u << 16);
    const int visible = ultra_hip_device_count();
    if (visible <= 0) { std::fprintf(stderr, "no HIP device: the receive path has no CPU fallback\n"); return 1; }
    if (gpus < 1 || gpus > visible) { std::fprintf(stderr, "--gpus %d but %d device(s) visible\n", gpus, visible); return 2; }
    std::vector<float> snrs;
    for (double s = snr_from; s <= snr_to + 1e-9; s += snr_step) snrs.push_back(static_cast<float>(s));

    Rccl rccl;
    std::vector<void*> comms(static_cast<size_t>(gpus), nullptr);
    if (!rccl.load()) { std::fprintf(stderr, "librccl not loadable\n"); return 1; }
    {
        std::vector<int> devs(static_cast<size_t>(gpus));
        for (int d = 0; d < gpus; ++d) devs[static_cast<size_t>(d)] = d;
        if (rccl.CommInitAll(comms.data(), gpus, devs.data()) != 0) { std::fprintf(stderr, "ncclCommInitAll failed\n"); return 1; }
    }

    struct Cell { Modulation mod; CodeRate rate; std::string label; };
    std::vector<Cell> cells;
    static const char* rate_names[] = {"R1_4", "R1_3", "R1_2", "R2_3", "R3_4", "R5_6"};
    if (cfg4) {
        cells.push_back({Modulation::DQPSK, static_cast<CodeRate>(rate_arg), std::string("LDPC_") + rate_names[rate_arg % 6]});
    } else {                                                   // tools/test_nvis_mode.cpp:172-185 widened to BASELINE's 5 x 6 grid
        const std::pair<Modulation, const char*> mods[] = {{Modulation::DBPSK, "DBPSK"}, {Modulation::DQPSK, "DQPSK"},
                                                           {Modulation::D8PSK, "D8PSK"}, {Modulation::QAM16, "QAM16"}, {Modulation::QAM32, "QAM32"}};
        const CodeRate rates[] = {CodeRate::R1_4, CodeRate::R1_3, CodeRate::R1_2, CodeRate::R2_3, CodeRate::R3_4, CodeRate::R5_6};
        for (const auto& m : mods) for (CodeRate r : rates)
            cells.push_back({m.first, r, std::string(m.second) + "_" + rate_names[static_cast<unsigned>(r)]});
    }
    std::printf("%s: %zu curve(s) x %zu SNR point(s) x %llu trials on %d GPU(s), seed 0x%llx\n", config.c_str(), cells.size(),
                snrs.size(), static_cast<unsigned long long>(trials), gpus, static_cast<unsigned long long>(seed));
    const auto t_all = std::chrono::steady_clock::now();
    unsigned long long total = 0;
    for (size_t ci = 0; ci < cells.size(); ++ci) {
        ModemConfig c;
        if (!cfg4) {                                            // presets::nvis_mode(), tools/test_nvis_mode.cpp:195-212
            c.fft_size = 1024; c.num_carriers = 59; c.cp_mode = CyclicPrefixMode::MEDIUM; c.symbol_guard = 0;
            c.modulation = cells[ci].mod;
            c.use_pilots = !(c.modulation == Modulation::DBPSK || c.modulation == Modulation::DQPSK || c.modulation == Modulation::D8PSK);
            c.pilot_spacing = c.use_pilots ? 4 : 2;
        }
        c.code_rate = cells[ci].rate;
        ultra_hip_config probe = to_c_config(c, ULTRA_ENTRY_SYNCED, 1, 0);
        ultra_hip_geometry g;
        must(ultra_hip_geometry_for(&probe, &g), "geometry");
        const uint32_t n_sym = (648 + g.llrs_per_symbol - 1) / g.llrs_per_symbol;
        const ultra_hip_config cfg = to_c_config(c, ULTRA_ENTRY_SYNCED, n_sym, 0);
        std::vector<Shard> shards(static_cast<size_t>(gpus));
        for (int d = 0; d < gpus; ++d)
            shards[static_cast<size_t>(d)].open(cfg, d, cfg4, static_cast<size_t>(std::min<uint64_t>(batch, (trials + gpus - 1) / gpus)));
        for (size_t si = 0; si < snrs.size(); ++si) {
            const uint64_t ps = point_seed(seed, ci * snrs.size() + si);
            const auto t0 = std::chrono::steady_clock::now();
            std::vector<std::thread> th;
            for (int d = 0; d < gpus; ++d)
                th.emplace_back([&, d] { shards[static_cast<size_t>(d)].run(trials * d / gpus, trials * (d + 1) / gpus, snrs[si], ps, channel_kind); });
            for (auto& t : th) t.join();
            // the single collective of the path: one all-reduce of the eight counters per point
            rccl.GroupStart();
            for (int d = 0; d < gpus; ++d)
                must(ultra_hip_counters_allreduce(shards[static_cast<size_t>(d)].ctx, comms[static_cast<size_t>(d)],
                                                  static_cast<ultra_hip_counters*>(shards[static_cast<size_t>(d)].d_cnt)), "counters_allreduce");
            rccl.GroupEnd();
            ultra_hip_counters t{};
            must(ultra_hip_memcpy_d2h(shards[0].ctx, &t, shards[0].d_cnt, sizeof(t)), "d2h");
            for (int d = 1; d < gpus; ++d) must(ultra_hip_synchronize(shards[static_cast<size_t>(d)].ctx), "sync");
            const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (t.frames != trials) { std::fprintf(stderr, "point counted %llu of %llu trials\n", (unsigned long long)t.frames, (unsigned long long)trials); return 1; }
            total += t.frames;
            std::printf("POINT %s %.1f %llu %llu %llu %llu %llu %llu %llu   # FER %.5f BER %.3e iters %.2f, %.2f M trials/s\n", cells[ci].label.c_str(),
                        snrs[si], (unsigned long long)t.frames, (unsigned long long)t.frame_errors, (unsigned long long)t.bit_errors,
                        (unsigned long long)t.info_bits, (unsigned long long)t.ldpc_fail, (unsigned long long)t.iters_sum,
                        (unsigned long long)t.undetected_errors, double(t.frame_errors) / double(t.frames),
                        double(t.bit_errors) / double(t.info_bits), double(t.iters_sum) / double(t.frames), trials / sec / 1e6);
        }
        for (auto& s : shards) s.close();
    }
    const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_all).count();
    std::printf("%s: %llu trials in %.2f s = %.2f M trials/s (stimulus generation included)\n", config.c_str(), total, wall, total / wall / 1e6);
    for (void* c : comms) rccl.CommDestroy(c);
    return 0;
}
