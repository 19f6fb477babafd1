// ultra_hip.hip — C-ABI (include/ultra_hip.h) over the gfx950 kernels.
//
// Host side of the drop-in boundary: builds the constant tables the way the
// reference constructors do (host_tables.h), owns the device copies, validates
// operand shapes before every launch (a faulting kernel can reset the whole
// host) and launches the hand-written kernels on the context's stream.  No
// CPU fallback exists: every entry point either runs the HIP path or fails.
#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <atomic>
#include <chrono>
#include <vector>

#include "../../include/ultra_hip.h"
#include "device_types.h"
#include "host_tables.h"
#include "demod_kernel.h"
#include "ldpc_kernel.h"
#include "ldpc_screen_kernel.h"
#include "ldpc_totals_kernel.h"
#include "acquire_kernel.h"
#include "stimulus_kernel.h"
#include "chirp_kernel.h"
#include "frame_kernel.h"

using namespace ultra_hip;

struct ultra_hip_ctx {
    ultra_hip_config cfg{};
    ultra_hip_geometry geo{};
    int device = 0;
    hipStream_t stream = nullptr;
    DemodConst h_demod{};
    LdpcConst h_ldpc{};
    DemodConst* d_demod = nullptr;
    LdpcPlan h_plan{};
    LdpcPlan* d_plan = nullptr;
    LdpcTPlan h_tplan{};                 // totals kernel (R2/3, R3/4, R5/6): valid = 0 -> the message kernel decodes
    LdpcTPlan* d_tplan = nullptr;
    unsigned int* d_work = nullptr;      // work-queue heads of the LDPC kernel (one per launch slot)
    int work_slot = 0;
    // the decoder's screen (ldpc_screen_kernel.h).  screen_mode: 1 = on where the sample says it pays (default), 0 = off
    // (ULTRA_HIP_LDPC_SCREEN=0), 2 = the full pass for every launch of any size (=2: parity tests).  d_screen_var: the rows'
    // variables (host-made); d_screen_pos: their positions under the current deinterleaver setting (device-made when stale).
    int screen_mode = 1;
    uint16_t* d_screen_var = nullptr;
    unsigned long long screen_prof = 0;  // edges per row round of the table (rows by degree, highest first), four bits each
    LdpcScreenPos* d_screen_pos = nullptr;
    bool screen_pos_stale = true;
    unsigned* d_ws_list = nullptr;       // work list of the iterating kernel: the codewords the screen did not finish
    size_t ws_list_cw = 0;
    c32* d_nco = nullptr;
    c32* d_twiddle = nullptr;
    float* d_lts = nullptr;              // LTS passband templates I then Q (acquisition)
    // stimulus generator (ultra_hip_make_batch): TX oscillator table, preamble, per-frame (max, sum of squares)
    c32* d_nco_tx = nullptr;
    float* d_preamble = nullptr;         // 7 preamble symbols + [max |x|, sum x^2]
    float* d_ws_fstats = nullptr;
    c32* d_ws_cfo = nullptr;             // channel CFO shift: mixer and rotator sequences of one call (2 x n samples)
    size_t ws_cfo_samples = 0;
    size_t ws_fstats_frames = 0;
    int stim_ncw_raw = 0, stim_ncw_enc = 0, stim_tx_symbols = 0, stim_pre_len = 0;
    uint8_t* d_ws_frame = nullptr;       // frame decode: bytes, ok, iters of every codeword
    size_t ws_frame_cw = 0;
    unsigned* d_ws_chirp = nullptr;      // chirp receive: detected, start, cfo, corr, entry, offset, cfo used, phase
    size_t ws_chirp_streams = 0;
    float* d_chirp = nullptr;            // chirp templates: up sin, up cos, down sin, down cos
    ChirpHostTables h_chirp{};
    unsigned* d_acq_gcache = nullptr;    // live streams (ultra_hip_acquire_stream_batch): metric caches, dev::kAcqGWords words per stream
    size_t acq_gcache_streams = 0;
    unsigned* d_ws_acq = nullptr;        // receive_batch workspace: found, data_start, entry, offset [n] + cfo [n]
    size_t ws_acq_frames = 0;
    uint32_t lts_len = 0;
    float lts_energy_ref = 0.0f;
    // workspace for the fused call when the caller does not want LLRs
    float* d_ws_llr = nullptr;
    size_t ws_llr_frames = 0;
    // demodulator workspace: per-frame tracker records + the used FFT bins of the symbol in flight
    float* d_ws_state = nullptr;
    c32* d_ws_fq = nullptr;
    unsigned* d_ws_seg = nullptr;       // per-frame CFO phase tables (cfo_walk_kernel -> mix_fft_kernel)
    size_t ws_demod_frames = 0;
    size_t ws_fq_rows = 0;
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
    int cu_count = 256;
    // per-kernel profiling (ultra_hip_profile_*): recorded (class, start, stop) triples + spare events
    uint32_t deint_step = 1;             // ChannelInterleaver step fused into the LDPC LLR load (1 = off)
    uint16_t* d_deint_table = nullptr;   // general gather table of the fused deinterleave (nullptr = use the step)
    int mix_wg_per_cu = 0;               // variant builds only (-DUH_AB_SWITCHES, ULTRA_HIP_MIX_WG_PER_CU): workgroups per CU of the transform's grid
    bool stream_cfo_given = false;       // launch_demod: whether the frame in flight started with caller-supplied offsets
    char* h_stage = nullptr;             // ultra_hip_memcpy_h2d_async: pinned staging ring
    char* d_stage = nullptr;             // ... and the device's view of it (ultra_hip_stage_input)
    size_t stage_cap = 0, stage_off = 0;
    std::vector<void*> host_blocks;      // ultra_hip_host_block: pinned, device-mapped result blocks
    bool post_refused = false;           // hipStreamWriteValue32 refused the block's memory once: ultra_hip_stream_post launches its kernel
    int stream_start_mode = 0;           // ultra_hip_demod_stream_start: how the next first_symbol == 0 stream call starts (consumed by it)
    const float* stream_start_timing = nullptr;
    // ULTRA_HIP_FALLBACK_CHAIN=1: the fall-back kernels for every layout — track_pilot_kernel + track_kernel per symbol instead
    // of the deferred carrier half and the pair tracker (what launch_demod drops to when the n_sym-fold workspace cannot be
    // had), and with ULTRA_HIP_LDPC_MESSAGES=1 the message-passing decoder for every rate (what the totals decoder drops to
    // when its LDS placement is refused).  Both are product paths, so both are held to the same parity tests
    // (tests/test_gpu_fallbacks.py).
    bool old_chain = false;
    float* d_ws_trk = nullptr;           // deferred carrier half: one record per (symbol, frame) from track_pilot_kernel to track_all_kernel
    size_t ws_trk_rows = 0;
    // ultra_hip_get_status: fall-back paths taken (sticky), the n_sym-fold workspace cap, the screen's last decision
    uint32_t status_flags = 0;
    size_t ws_limit_bytes = 0;
    unsigned* d_status = nullptr;        // copy of the last screened launch's control words (sample clean, dirty)
    uint32_t screen_launches = 0, last_screen_gate = 0, last_screen_sample_n = 0;
    bool profiling = false;
    struct Span { int kind; hipEvent_t e0, e1; unsigned long long items; };
    std::vector<Span> spans;
    std::vector<hipEvent_t> spare_events;
};

namespace {

// launches smaller than this are decoded without the screen: what the sample and the two launch boundaries cost when the gate
// stays shut (~13 us measured: profiles/r05_variants/r05_screen_kernel_split.txt) is 0.1 % of the headline step but would be a
// fifth of a 16,384-codeword launch of codewords that need one or two iterations each
constexpr size_t kScreenMinCodewords = 32768;

// every blocking wait of the host on the device that the library itself issues (ultra_hip_host_sync_count): what a
// latency-bound caller — one stream, one process() call at a time — pays per call beside the kernels
std::atomic<unsigned long long> g_host_syncs{0};
inline hipError_t uh_stream_sync(hipStream_t s) { g_host_syncs.fetch_add(1, std::memory_order_relaxed); return hipStreamSynchronize(s); }

#define UH_HIP(call)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess) {                                                               \
            std::fprintf(stderr, "ultra_hip: %s failed: %s (%s:%d)\n", #call,                  \
                         hipGetErrorString(e_), __FILE__, __LINE__);                          \
            return (e_ == hipErrorOutOfMemory) ? ULTRA_HIP_ERR_OOM : ULTRA_HIP_ERR_HIP;        \
        }                                                                                     \
    } while (0)

// brackets one kernel launch with events when the context is profiling
struct LaunchSpan {
    ultra_hip_ctx* ctx;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int kind;
    unsigned long long items;                      // work items of the launch (frame-symbols, codewords, streams): ultra_hip_profile_read_items
    static hipEvent_t take(ultra_hip_ctx* c) {
        hipEvent_t e = nullptr;
        if (!c->spare_events.empty()) { e = c->spare_events.back(); c->spare_events.pop_back(); }
        else if (hipEventCreate(&e) != hipSuccess) e = nullptr;
        return e;
    }
    LaunchSpan(ultra_hip_ctx* c, int k, unsigned long long n_items = 0) : ctx(c), kind(k), items(n_items) {
        if (!ctx->profiling) return;
        e0 = take(ctx); e1 = take(ctx);
        if (e0) (void)hipEventRecord(e0, ctx->stream);
    }
    ~LaunchSpan() {
        if (!ctx->profiling || !e0 || !e1) return;
        (void)hipEventRecord(e1, ctx->stream);
        ctx->spans.push_back({kind, e0, e1, items});
    }
};

struct DeviceGuard {
    int prev = -1;
    bool ok = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        ok = (hipSetDevice(dev) == hipSuccess);
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

// The per-frame workspaces grow on demand (a stream synchronisation and hipMalloc inside the call that first needs
// them); ultra_hip_reserve sizes them up front.
int ensure_demod_workspace(ultra_hip_ctx* ctx, size_t n_frames) {
    if (ctx->ws_demod_frames >= n_frames) return ULTRA_HIP_OK;
    UH_HIP(uh_stream_sync(ctx->stream));
    if (ctx->d_ws_state) { (void)hipFree(ctx->d_ws_state); ctx->d_ws_state = nullptr; }
    if (ctx->d_ws_seg) { (void)hipFree(ctx->d_ws_seg); ctx->d_ws_seg = nullptr; }
    ctx->ws_demod_frames = 0;
    UH_HIP(hipMalloc(&ctx->d_ws_state, n_frames * (size_t)dev::kStFloats * sizeof(float)));
    UH_HIP(hipMalloc(&ctx->d_ws_seg, n_frames * (size_t)dev::kSegTabWords * sizeof(unsigned)));
    ctx->ws_demod_frames = n_frames;
    return ULTRA_HIP_OK;
}
// used FFT bins: one row of 2 * fq_half (64 or 128) per frame — per frame AND symbol where all symbols are transformed in one launch
int ensure_fq_workspace(ultra_hip_ctx* ctx, size_t rows) {
    if (ctx->ws_fq_rows >= rows) return ULTRA_HIP_OK;
    UH_HIP(uh_stream_sync(ctx->stream));
    if (ctx->d_ws_fq) { (void)hipFree(ctx->d_ws_fq); ctx->d_ws_fq = nullptr; }
    ctx->ws_fq_rows = 0;
    UH_HIP(hipMalloc(&ctx->d_ws_fq, rows * (size_t)(2 * ctx->h_demod.fq_half) * sizeof(c32)));
    ctx->ws_fq_rows = rows;
    return ULTRA_HIP_OK;
}
int ensure_trk_workspace(ultra_hip_ctx* ctx, size_t rows) {
    // rows of dev::trk_rec_floats(n_pilot) floats: ONE cache line per (symbol, frame) for layouts with <= 15 pilots, three otherwise
    if (ctx->ws_trk_rows >= rows) return ULTRA_HIP_OK;
    UH_HIP(uh_stream_sync(ctx->stream));
    if (ctx->d_ws_trk) { (void)hipFree(ctx->d_ws_trk); ctx->d_ws_trk = nullptr; }
    ctx->ws_trk_rows = 0;
    UH_HIP(hipMalloc(&ctx->d_ws_trk, rows * (size_t)dev::trk_rec_floats(ctx->h_demod.n_pilot) * sizeof(float)));
    ctx->ws_trk_rows = rows;
    return ULTRA_HIP_OK;
}
int ensure_list_workspace(ultra_hip_ctx* ctx, size_t n_cw) {
    if (ctx->ws_list_cw >= n_cw) return ULTRA_HIP_OK;
    if (ctx->d_ws_list) { UH_HIP(uh_stream_sync(ctx->stream)); (void)hipFree(ctx->d_ws_list); ctx->d_ws_list = nullptr; }
    ctx->ws_list_cw = 0;
    UH_HIP(hipMalloc(&ctx->d_ws_list, n_cw * sizeof(unsigned)));
    ctx->ws_list_cw = n_cw;
    return ULTRA_HIP_OK;
}
int ensure_llr_workspace(ultra_hip_ctx* ctx, size_t n_frames) {
    if (ctx->ws_llr_frames >= n_frames) return ULTRA_HIP_OK;
    if (ctx->d_ws_llr) { UH_HIP(uh_stream_sync(ctx->stream)); (void)hipFree(ctx->d_ws_llr); ctx->d_ws_llr = nullptr; }
    ctx->ws_llr_frames = 0;
    UH_HIP(hipMalloc(&ctx->d_ws_llr, n_frames * (size_t)ctx->geo.llrs_per_frame * sizeof(float)));
    ctx->ws_llr_frames = n_frames;
    return ULTRA_HIP_OK;
}

int launch_demod(ultra_hip_ctx* ctx, const float* d_audio, size_t frame_stride, const float* d_cfo_hz,
                 const float* d_cfo_phase, size_t n_frames, float* d_llr, size_t llr_stride, float* d_state,
                 const unsigned* d_frame_offset = nullptr, int sym_begin = 0, int sym_count = -1, c32* d_eq = nullptr) {
    // d_eq (ultra_hip_demod_stream_batch_eq, nullable): [n_frames][data symbols of this call][kMaxCarriers] equalized data
    // carriers; such a call takes the per-symbol chain (track_kernel's EQ instance), whose results are the other chains'.
    // sym_begin / sym_count (ultra_hip_demod_stream_batch): symbols [sym_begin, sym_begin + sym_count) of every frame,
    // continuing from the tracker records the previous call left in the context's workspace (sym_begin > 0: no
    // initialisation; d_audio and d_llr then address the frame as if it were complete — the caller shifts its pointers)
    if (n_frames == 0) return ULTRA_HIP_OK;
    const DemodConst& D = ctx->h_demod;
    const int s_begin = sym_begin, s_end = (sym_count < 0) ? D.n_train + D.n_data_sym : sym_begin + sym_count;
    if (s_begin < 0 || s_end > D.n_train + D.n_data_sym || s_end <= s_begin) return ULTRA_HIP_ERR_INVALID_ARG;
    if (s_begin > 0 && ctx->ws_demod_frames < n_frames) return ULTRA_HIP_ERR_INVALID_ARG;      // nothing to continue from
    // ultra_hip_demod_stream_start: a frame that starts on a USED demodulator (carried tracker) or with a timing offset may
    // hold a frequency offset or a previous symbol's pilots from its first symbol on — none of the fresh-start short cuts
    const int start_mode = (s_begin == 0 && sym_count >= 0) ? ctx->stream_start_mode : ULTRA_STREAM_START_FRESH;
    const float* start_timing = (start_mode == ULTRA_STREAM_START_TIMING) ? ctx->stream_start_timing : nullptr;
    if (s_begin == 0 && sym_count >= 0) { ctx->stream_start_mode = ULTRA_STREAM_START_FRESH; ctx->stream_start_timing = nullptr; }
    if (start_mode == ULTRA_STREAM_START_SYNC && (ctx->ws_demod_frames < n_frames || D.presynced)) return ULTRA_HIP_ERR_INVALID_ARG;   // nothing to carry
    const bool cfo_given = (s_begin == 0) ? (d_cfo_hz != nullptr || start_mode != ULTRA_STREAM_START_FRESH) : ctx->stream_cfo_given;
    if (s_begin == 0) ctx->stream_cfo_given = cfo_given;
    { const int rc_ws = ensure_demod_workspace(ctx, n_frames); if (rc_ws != ULTRA_HIP_OK) return rc_ws; }
    // one wavefront per frame in every kernel; grids are capped (grid-stride over frames) so a huge
    // batch stays one launch per stage and per-workgroup constants are loaded once
    // Many short-lived workgroups (a few frames each) beat one resident set looping over the batch:
    // the hardware dispatcher rebalances CUs/XCDs that run slower (measured 7.19 -> 6.49 ms for the
    // demodulator stage at 2^18 frames; sweep in profiles/README.md; re-swept after the walk moved out of
    // mix_fft_kernel: 96..192 workgroups per CU are level, 384 is 2.5 % slower).
    // The pipelined transform wants a couple of dozen items per workgroup (its prologue — twiddles, the oscillator at the
    // lane's samples, the first item's unhidden requests — is paid once per workgroup) and enough workgroups for the
    // dispatcher to rebalance: 24 items each, between 24 and 128 workgroups per CU (sweep at 2^17 and 2^20 frames:
    // profiles/r03_variants/r03_ab_transform_grid.txt — 2^17: 0.325 ms at 128 per CU, 0.312 at 24-48; 2^20: 2.46 at 128-256,
    // 2.56 at 24).
    auto mix_grid = [&](size_t items) {
        size_t per_cu = ctx->mix_wg_per_cu > 0 ? (size_t)ctx->mix_wg_per_cu
                                               : std::min<size_t>(128, std::max<size_t>(24, items / ((size_t)ctx->cu_count * 24)));
        return (unsigned)std::min(items, (size_t)ctx->cu_count * per_cu);
    };
    const unsigned grid_fft = mix_grid(n_frames);
    const unsigned grid_trk = (unsigned)std::min(n_frames, (size_t)ctx->cu_count * 128);
    hipStream_t st = ctx->stream;
    if (D.log2_fft != 10 && D.log2_fft != 9) return ULTRA_HIP_ERR_UNSUPPORTED;
    const int n_sym = s_end - s_begin;
    // No pilots, SYNCED entry, no initial offsets: nothing on the path ever estimates a CFO (that is the pilot half's job),
    // it is 0 for every frame and every symbol, mix_fft_kernel never rotates — no phase tables to walk.
    const bool cfo_is_zero = !D.presynced && D.n_pilot == 0 && !cfo_given;
    const unsigned* seg_tab = cfo_is_zero ? nullptr : ctx->d_ws_seg;
    // ... and no symbol's transform depends on the symbol before it: ALL symbols of all frames in one launch (a grid of
    // n_frames * n_sym items instead of n_sym launches that each ramp up and drain), bins to one Fq row per frame and symbol
    bool all_symbols_at_once = cfo_is_zero && n_sym > 1 && n_frames * (size_t)n_sym < 0x7fffffffull;
    // Coherent layouts with pilots entered SYNCED: the carrier half does not feed back into the tracker, so it runs once,
    // behind the last symbol, over every (symbol, frame) — track_all_kernel, demod_kernel.h; per symbol only
    // cfo_walk -> mix_fft -> track_pilot remain.  Needs every symbol's bins and a record per (symbol, frame).
    // (not with the adaptive equaliser: its weights are per-carrier state that the carrier half carries from symbol to symbol)
    bool deferred = !ctx->old_chain && !d_eq && !D.differential && D.n_pilot > 0 && D.n_pilot <= dev::kPwPilots && D.n_train == 0 &&
                    !D.presynced && D.adaptive_eq == 0 && n_frames * (size_t)n_sym < 0x7fffffffull;
    // n_sym rows of workspace per frame instead of one: if that cannot be had, fall back to the per-symbol launches
    // (or is above the caller's cap, ultra_hip_set_workspace_limit).  Either way the caller can see it: ultra_hip_get_status.
    const size_t fold_rows = n_frames * (size_t)n_sym;
    auto over_limit = [&](size_t bytes) { return ctx->ws_limit_bytes != 0 && bytes > ctx->ws_limit_bytes; };
    if ((all_symbols_at_once || deferred) &&
        (over_limit(fold_rows * (size_t)(2 * D.fq_half) * sizeof(c32)) || ensure_fq_workspace(ctx, fold_rows) != ULTRA_HIP_OK)) {
        (void)hipGetLastError();
        all_symbols_at_once = false; deferred = false;
        ctx->status_flags |= ULTRA_HIP_ST_DEMOD_WORKSPACE_FALLBACK;
    }
    if (deferred && (over_limit(fold_rows * (size_t)dev::trk_rec_floats(D.n_pilot) * sizeof(float)) || ensure_trk_workspace(ctx, fold_rows) != ULTRA_HIP_OK)) {
        (void)hipGetLastError(); deferred = false;
        ctx->status_flags |= ULTRA_HIP_ST_DEMOD_WORKSPACE_FALLBACK;
    }
    { const int rc_fq = ensure_fq_workspace(ctx, n_frames); if (rc_fq != ULTRA_HIP_OK) return rc_fq; }
    // A batch that starts at symbol 0 of the deferred chain without initial offsets: the pilot half of symbol 0 starts from the
    // constructor's values itself (track_pilot_kernel, `fresh`) and writes the whole record — no initialisation launch, and
    // no record read in that launch.  (With offsets given the walk of symbol 0 reads them from the record first.)
    const bool fresh_pilot = deferred && s_begin == 0 && !cfo_given;
    // A presynced context's symbols behind its first call arrive through process() (ultra_hip.h, ultra_hip_demod_stream_batch):
    // the SYNCED loop updates the channel estimate for every layout, processPresynced's own loop only with pilots
    const int synced_loop = (D.presynced && s_begin > 0) ? 1 : 0;
    if (s_begin == 0 && start_mode == ULTRA_STREAM_START_SYNC) {
        LaunchSpan span(ctx, ULTRA_HIP_K_INIT_STATE);
        // if (!is_differential || config.use_pilots) the carrier phase correction starts over (demodulator.cpp:583-586)
        hipLaunchKernelGGL(dev::resync_state_kernel, dim3(grid_trk), dim3(dev::kWave), 0, st, d_cfo_hz, d_cfo_phase, (int)n_frames, ctx->d_ws_state,
                           (!D.differential || D.n_pilot > 0) ? 1 : 0);
    } else if (s_begin == 0 && !fresh_pilot) {
        LaunchSpan span(ctx, ULTRA_HIP_K_INIT_STATE);
        // compact pilot state (demod_kernel.h, kStHp) and no training symbols that read the full H array first
        const int compact = (!D.differential && D.n_pilot > 0 && D.n_train == 0) ? 1 : 0;
        hipLaunchKernelGGL(dev::init_state_kernel, dim3(grid_trk), dim3(dev::kWave), 0, st, d_cfo_hz, d_cfo_phase,
                           (int)n_frames, ctx->d_ws_state, compact, (D.adaptive_eq != 0 && !D.differential) ? 1 : 0, start_timing);
    }
    if (s_begin == 0 && D.presynced && d_cfo_hz) {
        // frames whose initial CFO is NaN ("never set"): estimateCFOFromTraining, demodulator.cpp:920-925
        LaunchSpan span(ctx, ULTRA_HIP_K_INIT_STATE);
        hipLaunchKernelGGL(dev::train_cfo_kernel, dim3((unsigned)((n_frames + 255) / 256)), dim3(256), 0, st, ctx->d_demod, ctx->d_nco,
                           d_audio, frame_stride, d_frame_offset, d_cfo_hz, (int)n_frames, ctx->d_ws_state);
    }
    // The transform of symbol `sym` (n_sym_batch symbols from it on) of every frame: 512 points per wavefront, software-
    // pipelined (mix_fft2_kernel: two wavefronts per frame at N = 1024, one at N = 512), with or without the CFO rotation.
    auto launch_mix = [&](unsigned g, int sym, c32* fq, const unsigned* tab, int n_sym_batch) {
        auto go = [&](auto kernel, unsigned threads) {
            hipLaunchKernelGGL(kernel, dim3(g), dim3(threads), 0, st, ctx->d_demod, ctx->d_nco, ctx->d_twiddle, d_audio, frame_stride,
                               d_frame_offset, (int)n_frames, sym, fq, tab, n_sym_batch);
        };
        if (tab && n_sym_batch != 1) return (int)ULTRA_HIP_ERR_UNSUPPORTED;   // mix_fft2_kernel<.., true> takes one symbol index per launch (ds_of)
        if (D.log2_fft == 10) {
            if (tab) go(dev::mix_fft2_kernel<10, true>, 2 * dev::kWave);
            else go(dev::mix_fft2_kernel<10, false>, 2 * dev::kWave);
        } else {
            if (tab) go(dev::mix_fft2_kernel<9, true>, dev::kWave);
            else go(dev::mix_fft2_kernel<9, false>, dev::kWave);
        }
        return (int)ULTRA_HIP_OK;
    };
    if (all_symbols_at_once) {
        LaunchSpan span(ctx, ULTRA_HIP_K_MIX_FFT, n_frames * (unsigned long long)n_sym);
        const unsigned g = mix_grid(n_frames * (size_t)n_sym);
        const int rc_mix = launch_mix(g, s_begin, ctx->d_ws_fq, nullptr, n_sym);       // cfo_is_zero: no table, the instance without the rotation
        if (rc_mix != ULTRA_HIP_OK) return rc_mix;
    }
    int mixed_upto = 0;                                  // symbols below this index are transformed already
    for (int s = s_begin; s < s_end; ++s) {
        c32* fq_s = ctx->d_ws_fq + ((all_symbols_at_once || deferred) ? (size_t)(s - s_begin) * n_frames * (size_t)(2 * D.fq_half) : (size_t)0);
        float* rec_s = deferred ? ctx->d_ws_trk + (size_t)(s - s_begin) * n_frames * dev::trk_rec_floats(D.n_pilot) : nullptr;
        // The first TWO symbols of a SYNCED batch without initial offsets are at CFO 0 in every frame: the tracker estimates
        // a CFO from the phase differences of the pilots between two symbols (channel_equalizer.cpp:421-470: only with
        // prev_pilot_phases from an earlier symbol), so the first estimate exists after the second symbol.  No table to
        // walk, no rotation: the transform's instance without it.
        const bool first_at_zero = s <= 1 && !D.presynced && !cfo_given;
        const unsigned* seg_tab_s = first_at_zero ? nullptr : seg_tab;
        if (!cfo_is_zero && !first_at_zero) {
            LaunchSpan span(ctx, ULTRA_HIP_K_WALK, n_frames);
            hipLaunchKernelGGL(dev::cfo_walk_kernel, dim3((unsigned)((n_frames + 255) / 256)), dim3(256), 0, st, ctx->d_demod,
                               (int)n_frames, ctx->d_ws_state, ctx->d_ws_seg);
        }
        if (!all_symbols_at_once && s >= mixed_upto) {
            const bool two = deferred && s == 0 && first_at_zero && s + 1 < s_end && n_frames * (size_t)2 < 0x7fffffffull;
            // the rotating instance (a phase table per frame) is a kernel of its own for the issue model: its own class
            LaunchSpan span(ctx, seg_tab_s ? ULTRA_HIP_K_MIX_FFT_ROT : ULTRA_HIP_K_MIX_FFT, n_frames * (two ? 2ull : 1ull));
            // Symbols 0 and 1 of a fresh batch on the deferred chain are both at CFO 0 and their bins go to consecutive row
            // blocks: ONE launch over 2 n_frames items transforms both (the tracker of symbol 0 does not feed symbol 1's
            // transform) — one ramp-up and drain less, which is what a rank's share of the strong-scaling batch notices.
            int rc_mix;
            if (two) {
                rc_mix = launch_mix(mix_grid(n_frames * 2), s, fq_s, nullptr, 2);
                mixed_upto = 2;
            } else {
                rc_mix = launch_mix(grid_fft, s, fq_s, seg_tab_s, 1);    // no table: CFO 0 in every frame, the instance without the rotation
            }
            if (rc_mix != ULTRA_HIP_OK) return rc_mix;
        }
        const bool training = s < D.n_train;
        const bool last = (s == s_end - 1);
        if (training) {
            LaunchSpan span(ctx, ULTRA_HIP_K_TRACK, n_frames);
            hipLaunchKernelGGL(dev::train_kernel, dim3(grid_trk), dim3(dev::kWave), 0, st, ctx->d_demod, (int)n_frames, s,
                               ctx->d_ws_state, fq_s);
            continue;
        }
        // Data symbols: the pilot half of the channel update runs 4 (<= 16 pilots) or 2 (<= 32: every usable
        // configuration, validate_config) frames per wavefront in its own kernel, the carrier half + equalise
        // + demap one frame per wavefront.
        if (D.n_pilot != 0) {                          // without pilots the half is three scalar updates: track_kernel takes them
            LaunchSpan span(ctx, ULTRA_HIP_K_PILOT, n_frames);
            if (D.n_pilot <= 16) {
                const unsigned g = (unsigned)std::min((n_frames + 3) / 4, (size_t)ctx->cu_count * 256);
                if (fresh_pilot && s == 0)
                    hipLaunchKernelGGL((dev::track_pilot_kernel<16, true>), dim3(g), dim3(dev::kWave), 0, st, ctx->d_demod, (int)n_frames,
                                       ctx->d_ws_state, fq_s, rec_s);
                else
                    hipLaunchKernelGGL((dev::track_pilot_kernel<16, false>), dim3(g), dim3(dev::kWave), 0, st, ctx->d_demod, (int)n_frames,
                                       ctx->d_ws_state, fq_s, rec_s);
            } else {
                const unsigned g = (unsigned)std::min((n_frames + 1) / 2, (size_t)ctx->cu_count * 512);
                if (fresh_pilot && s == 0)
                    hipLaunchKernelGGL((dev::track_pilot_kernel<32, true>), dim3(g), dim3(dev::kWave), 0, st, ctx->d_demod, (int)n_frames,
                                       ctx->d_ws_state, fq_s, rec_s);
                else
                    hipLaunchKernelGGL((dev::track_pilot_kernel<32, false>), dim3(g), dim3(dev::kWave), 0, st, ctx->d_demod, (int)n_frames,
                                       ctx->d_ws_state, fq_s, rec_s);
            }
        }
        if (deferred) {
            if (!last) continue;
            LaunchSpan span(ctx, ULTRA_HIP_K_TRACK, n_frames * (unsigned long long)n_sym);
            const unsigned g = (unsigned)std::min(n_frames * (size_t)n_sym, (size_t)ctx->cu_count * 128);
#define UH_TRACK_ALL(MOD)                                                                                                  \
    hipLaunchKernelGGL(dev::track_all_kernel<MOD>, dim3(g), dim3(dev::kWave), 0, st, ctx->d_demod, (int)n_frames, s_begin, n_sym, \
                       ctx->d_ws_trk, ctx->d_ws_fq, d_llr, llr_stride, d_state, ctx->d_ws_state)
            switch (D.modulation) {
                case ULTRA_MOD_BPSK: UH_TRACK_ALL(ULTRA_MOD_BPSK); break;
                case ULTRA_MOD_QPSK: UH_TRACK_ALL(ULTRA_MOD_QPSK); break;
                case ULTRA_MOD_QAM16: UH_TRACK_ALL(ULTRA_MOD_QAM16); break;
                case ULTRA_MOD_QAM32: UH_TRACK_ALL(ULTRA_MOD_QAM32); break;
                case ULTRA_MOD_QAM64: UH_TRACK_ALL(ULTRA_MOD_QAM64); break;
                case ULTRA_MOD_QAM256: UH_TRACK_ALL(ULTRA_MOD_QAM256); break;
                default: return ULTRA_HIP_ERR_UNSUPPORTED;
            }
#undef UH_TRACK_ALL
            break;
        }
        // zero-CFO layouts: every symbol's bins are there already — one launch walks all data symbols of a frame
        // (n_train == 0 for the SYNCED entry, so s == 0 here and the launch covers symbols 0 .. n_sym - 1)
        const int track_batch = all_symbols_at_once ? s_end - s : 1;
        const bool last_launch = last || all_symbols_at_once;
#define UH_TRACK(MOD)                                                                                            \
    do {                                                                                                         \
        if (d_eq)                                                                                                \
            hipLaunchKernelGGL((dev::track_kernel<MOD, true>), dim3(grid_trk), dim3(dev::kWave), 0, st, ctx->d_demod, (int)n_frames, \
                               s - D.n_train, ctx->d_ws_state, fq_s, d_llr, llr_stride, last_launch ? d_state : nullptr, track_batch, \
                               synced_loop, d_eq + (size_t)(s - std::max(s_begin, (int)D.n_train)) * kMaxCarriers, eq_stride);        \
        else                                                                                                     \
            hipLaunchKernelGGL((dev::track_kernel<MOD, false>), dim3(grid_trk), dim3(dev::kWave), 0, st, ctx->d_demod, (int)n_frames, \
                               s - D.n_train, ctx->d_ws_state, fq_s, d_llr, llr_stride, last_launch ? d_state : nullptr, track_batch, \
                               synced_loop, (c32*)nullptr, (size_t)0);                                           \
    } while (0)
        LaunchSpan span(ctx, ULTRA_HIP_K_TRACK, n_frames * (unsigned long long)track_batch);
        // differential layouts without pilots on at most 32 carriers (the 512-point presets): two frames per wavefront
        const size_t eq_stride = (size_t)(s_end - std::max(s_begin, (int)D.n_train)) * kMaxCarriers;   // data symbols of this call
        const bool pair_frames = !ctx->old_chain && !d_eq && D.differential && D.n_pilot == 0 && !D.presynced && D.n_carriers <= 32 &&
                                 D.n_train == 0;      // (without pilots every interpolation entry is empty: nothing to interpolate)
#define UH_TRACK_PAIR(MOD)                                                                                                         \
    hipLaunchKernelGGL(dev::track_diff_pair_kernel<MOD>, dim3((unsigned)std::min((n_frames + 1) / 2, (size_t)ctx->cu_count * 128)),  \
                       dim3(dev::kWave), 0, st, ctx->d_demod, (int)n_frames, s - D.n_train, ctx->d_ws_state, fq_s, d_llr, llr_stride, \
                       last_launch ? d_state : nullptr, track_batch)
        if (pair_frames && D.modulation == ULTRA_MOD_DBPSK) UH_TRACK_PAIR(ULTRA_MOD_DBPSK);
        else if (pair_frames && D.modulation == ULTRA_MOD_DQPSK) UH_TRACK_PAIR(ULTRA_MOD_DQPSK);
        else if (pair_frames && D.modulation == ULTRA_MOD_D8PSK) UH_TRACK_PAIR(ULTRA_MOD_D8PSK);
        else
#undef UH_TRACK_PAIR
        switch (D.modulation) {
            case ULTRA_MOD_DBPSK: UH_TRACK(ULTRA_MOD_DBPSK); break;
            case ULTRA_MOD_BPSK: UH_TRACK(ULTRA_MOD_BPSK); break;
            case ULTRA_MOD_DQPSK: UH_TRACK(ULTRA_MOD_DQPSK); break;
            case ULTRA_MOD_QPSK: UH_TRACK(ULTRA_MOD_QPSK); break;
            case ULTRA_MOD_D8PSK: UH_TRACK(ULTRA_MOD_D8PSK); break;
            case ULTRA_MOD_QAM16: UH_TRACK(ULTRA_MOD_QAM16); break;
            case ULTRA_MOD_QAM32: UH_TRACK(ULTRA_MOD_QAM32); break;
            case ULTRA_MOD_QAM64: UH_TRACK(ULTRA_MOD_QAM64); break;
            case ULTRA_MOD_QAM256: UH_TRACK(ULTRA_MOD_QAM256); break;
            default: return ULTRA_HIP_ERR_UNSUPPORTED;
        }
#undef UH_TRACK
        if (all_symbols_at_once) break;          // that launch covered symbols s .. n_sym - 1 (s == 0 here)
    }
    UH_HIP(hipGetLastError());
    return ULTRA_HIP_OK;
}

int launch_ldpc(ultra_hip_ctx* ctx, const float* d_llr, size_t llr_stride, size_t n_cw, uint8_t* d_bytes,
                int32_t* d_iters, uint8_t* d_ok, float* d_llr_total, int block_len = 0, int block_stride = 0) {
    if (n_cw == 0) return ULTRA_HIP_OK;
    // persistent workgroups (one wavefront each) pull codewords from an atomic counter; every
    // launch uses its own counter word, zeroed on the stream just before the launch
    const LdpcPlan& P = ctx->h_plan;
    unsigned int* counter = ctx->d_work + (size_t)(ctx->work_slot++ & 15) * dev::kLdpcQueueWords;
    UH_HIP(hipMemsetAsync(counter, 0, dev::kLdpcQueueWords * sizeof(unsigned int), ctx->stream));
    // The totals formulation (ldpc_totals_kernel.h: one total per variable through LDS, the row subtracts its own
    // message) where the code has a placement; same results bit for bit, two thirds of the LDS cycles.
    if (ctx->h_tplan.valid) {
        const LdpcTPlan& T = ctx->h_tplan;
        const size_t tlds = (size_t)T.lds_bytes;
        // the kernel derives its LDS offsets from its template arguments with the builder's formulas
        if (T.t_pad != T.var_rounds * 256 || T.r_base != T.t_pad + 128 || T.r_pad != T.r_base + T.n_planes * 256 ||
            T.stage_v != T.r_pad + 128 || T.stage_p != T.stage_v + T.var_rounds * 256 ||
            T.lds_bytes != T.stage_v + ldpc_stage_bytes(T.var_rounds, T.row_rounds))
            return ULTRA_HIP_ERR_UNSUPPORTED;
// one instance per code: (row rounds, variable rounds, row profile, variable profile); the plan's profiles select it
#define UH_TOTALS_LAUNCH(RR, VR, RP, VP, WV, CP)                                                                  \
    do {                                                                                                          \
        const size_t lds_wg = (CP) ? (size_t)T.stage_v + 4 * (size_t)(T.n_checked + T.m) : tlds;                  \
        const size_t per_cu = std::max<size_t>(1, std::min<size_t>(4 * (WV), (size_t)(160 * 1024) / lds_wg));     \
        const unsigned grid = (unsigned)std::min(n_cw, (size_t)ctx->cu_count * per_cu);                           \
        if (d_llr_total)                                                                                          \
            hipLaunchKernelGGL((dev::ldpc_totals_kernel<RR, VR, RP, VP, true, WV, false>), dim3(grid), dim3(dev::kLdpcThreads), tlds, \
                               ctx->stream, ctx->d_tplan, d_llr, llr_stride, (int)n_cw, d_bytes, d_iters, d_ok,   \
                               d_llr_total, counter, (int)ctx->deint_step, ctx->d_deint_table, block_len, block_stride, \
                               (const unsigned*)nullptr, 0u);                                                     \
        else                                                                                                      \
            hipLaunchKernelGGL((dev::ldpc_totals_kernel<RR, VR, RP, VP, false, WV, CP>), dim3(grid), dim3(dev::kLdpcThreads), lds_wg, \
                               ctx->stream, ctx->d_tplan, d_llr, llr_stride, (int)n_cw, d_bytes, d_iters, d_ok,   \
                               d_llr_total, counter, (int)ctx->deint_step, ctx->d_deint_table, block_len, block_stride, \
                               (const unsigned*)work_list, gate);                                                 \
    } while (0)
        auto is = [&](int rr, int vr, unsigned long long rp, unsigned long long vp) {
            return T.row_rounds == rr && T.var_rounds == vr && T.row_prof == rp && T.var_prof == vp;
        };
        const bool r34 = is(3, 6, 0x666ull, 0x333333ull), r56 = is(2, 4, 0x66ull, 0x3333ull), r23 = is(4, 7, 0x6666ull, 0x3333333ull),
                   r14 = is(8, 3, kPlaceRowProf_R1_4, kPlaceVarProf_R1_4), r13 = is(6, 6, kPlaceRowProf_R1_3, kPlaceVarProf_R1_3),
                   r12 = is(6, 6, kPlaceRowProf_R1_2, kPlaceVarProf_R1_2);
        const bool launched = r34 || r56 || r23 || r14 || r13 || r12;
        if (launched) {
            LaunchSpan span(ctx, ULTRA_HIP_K_LDPC, n_cw);
            // The screen (ldpc_screen_kernel.h): codewords whose channel values already satisfy every row are finished at
            // memory speed and the iterating kernel decodes the list of the others.  Only where a launch is large enough for
            // three more (mostly empty) launches not to show, an iteration may run at all, and the caller does not want the
            // a-posteriori values (those of a clean codeword are one iteration's, which the screen does not compute).
            unsigned* work_list = nullptr;
            unsigned gate = 0u;
            const bool screen = ctx->screen_mode != 0 && ctx->d_screen_pos && T.max_iterations > 0 && !d_llr_total &&
                                (ctx->screen_mode == 2 || n_cw >= kScreenMinCodewords) && n_cw <= 0x7fffffffull;
            if (screen && ensure_list_workspace(ctx, n_cw) != ULTRA_HIP_OK) {                            // no list: plain decode, and the caller can see it
                (void)hipGetLastError();
                ctx->status_flags |= ULTRA_HIP_ST_SCREEN_LIST_UNAVAILABLE;
            } else if (screen) {
                if (ctx->screen_pos_stale) {
                    hipLaunchKernelGGL(dev::ldpc_screen_prepare_kernel, dim3(1), dim3(512), 0, ctx->stream, ctx->d_screen_var, T.m, T.k,
                                       (int)ctx->deint_step, ctx->d_deint_table, ctx->d_screen_pos);
                    ctx->screen_pos_stale = false;
                }
                const int sample_n = (int)std::min<size_t>(n_cw, (size_t)dev::kScreenSampleMax);
                const int sample_stride = (int)(n_cw / (size_t)sample_n);
                gate = ctx->screen_mode == 2 ? 0u : (unsigned)((sample_n + 2) / 3);     // a third of the sample clean: the pass pays
                work_list = ctx->d_ws_list;
                const unsigned gs = (unsigned)((sample_n + dev::kScreenWaves - 1) / dev::kScreenWaves);
                const unsigned gf = (unsigned)((n_cw + dev::kScreenChunk - 1) / dev::kScreenChunk);
#define UH_SCREEN_LAUNCH(RR, EP)                                                                                                 \
    do {                                                                                                                         \
        hipLaunchKernelGGL((dev::ldpc_screen_kernel<RR, EP, true>), dim3(gs), dim3(dev::kScreenThreads), 0, ctx->stream, ctx->d_screen_pos, \
                           d_llr, llr_stride, (int)n_cw, block_len, block_stride, T.decoded_bytes, d_bytes, d_iters, d_ok, counter,     \
                           work_list, gate, sample_n, sample_stride);                                                            \
        hipLaunchKernelGGL((dev::ldpc_screen_kernel<RR, EP, false>), dim3(gf), dim3(dev::kScreenThreads), 0, ctx->stream, ctx->d_screen_pos, \
                           d_llr, llr_stride, (int)n_cw, block_len, block_stride, T.decoded_bytes, d_bytes, d_iters, d_ok, counter,     \
                           work_list, gate, sample_n, sample_stride);                                                            \
    } while (0)
                // one instance per (row rounds, sorted degree profile) of the reference's six codes; any other graph of the same
                // number of rounds gathers seven edge slots in every round
                const unsigned long long ep = ctx->screen_prof;
                switch (T.row_rounds) {
                    case 2: UH_SCREEN_LAUNCH(2, 0x77ull); break;
                    case 3: UH_SCREEN_LAUNCH(3, 0x777ull); break;
                    case 4: UH_SCREEN_LAUNCH(4, 0x7777ull); break;
                    case 6: if (ep == 0x244677ull) UH_SCREEN_LAUNCH(6, 0x244677ull);            // R1/3
                            else if (ep == 0x235677ull) UH_SCREEN_LAUNCH(6, 0x235677ull);       // R1/2
                            else UH_SCREEN_LAUNCH(6, 0x777777ull);
                            break;
                    case 8: if (ep == 0x33456677ull) UH_SCREEN_LAUNCH(8, 0x33456677ull);        // R1/4
                            else UH_SCREEN_LAUNCH(8, 0x77777777ull);
                            break;
                    default: work_list = nullptr; break;
                }
#undef UH_SCREEN_LAUNCH
                if (work_list) { ++ctx->screen_launches; ctx->last_screen_gate = gate; ctx->last_screen_sample_n = (uint32_t)sample_n; }
            }
            // R3/4 without a fused deinterleaver stages only the 487 values its decoder reads: 8,348 B per workgroup, 19 per CU
            // instead of 18 (ldpc_totals_kernel.h, COMPACT).
            const bool compact = r34 && !d_llr_total && ctx->deint_step == 1u && !ctx->d_deint_table && T.k == 486 && T.n_checked == 325 &&
                                 T.m == 162;            // the instance's own constants (ldpc_totals_kernel.h, kCompactK / kCompactChecked)
            if (r34 && compact) UH_TOTALS_LAUNCH(3, 6, 0x666ull, 0x333333ull, 5, true);
            else if (r34) UH_TOTALS_LAUNCH(3, 6, 0x666ull, 0x333333ull, 5, false);
            else if (r56) UH_TOTALS_LAUNCH(2, 4, 0x66ull, 0x3333ull, 6, false);
            else if (r23) UH_TOTALS_LAUNCH(4, 7, 0x6666ull, 0x3333333ull, 4, false);
            else if (r14) UH_TOTALS_LAUNCH(8, 3, kPlaceRowProf_R1_4, kPlaceVarProf_R1_4, 3, false);
            else if (r13) UH_TOTALS_LAUNCH(6, 6, kPlaceRowProf_R1_3, kPlaceVarProf_R1_3, 4, false);
            else UH_TOTALS_LAUNCH(6, 6, kPlaceRowProf_R1_2, kPlaceVarProf_R1_2, 4, false);
            // the launch's counter block is recycled sixteen launches on: what the screen decided is kept for ultra_hip_get_status
            if (work_list) (void)hipMemcpyAsync(ctx->d_status, counter, 4 * sizeof(unsigned), hipMemcpyDeviceToDevice, ctx->stream);
        }
#undef UH_TOTALS_LAUNCH
        if (launched) { UH_HIP(hipGetLastError()); return ULTRA_HIP_OK; }
    }
    ctx->status_flags |= ULTRA_HIP_ST_LDPC_MESSAGE_KERNEL;              // not the totals kernel: visible to the caller
    const size_t lds = dev::ldpc_lds_bytes(P.msg_words);
    // WV = wavefronts per SIMD the instance's registers are budgeted for; the grid is one resident set
    // (bounded by LDS: one codeword's messages + staging per workgroup)
#define UH_LDPC_LAUNCH(RR, VR, RMAX, RMIN, VMAX, VMIN, RID, LIN, WV)                                                 \
    do {                                                                                                        \
        LaunchSpan span(ctx, ULTRA_HIP_K_LDPC, n_cw);                                                           \
        const size_t per_cu = std::max<size_t>(1, std::min<size_t>(4 * (WV), (size_t)(160 * 1024) / lds));      \
        const unsigned grid = (unsigned)std::min(n_cw, (size_t)ctx->cu_count * per_cu);                         \
        if (d_llr_total)                                                                                        \
            hipLaunchKernelGGL((dev::ldpc_decode_kernel<RR, VR, RMAX, RMIN, VMAX, VMIN, RID, LIN, true, WV>),        \
                               dim3(grid), dim3(dev::kLdpcThreads), lds, ctx->stream, ctx->d_plan, d_llr,       \
                               llr_stride, (int)n_cw, d_bytes, d_iters, d_ok, d_llr_total, counter,             \
                               (int)ctx->deint_step, ctx->d_deint_table, block_len, block_stride);                                                           \
        else                                                                                                    \
            hipLaunchKernelGGL((dev::ldpc_decode_kernel<RR, VR, RMAX, RMIN, VMAX, VMIN, RID, LIN, false, WV>),       \
                               dim3(grid), dim3(dev::kLdpcThreads), lds, ctx->stream, ctx->d_plan, d_llr,       \
                               llr_stride, (int)n_cw, d_bytes, d_iters, d_ok, d_llr_total, counter,             \
                               (int)ctx->deint_step, ctx->d_deint_table, block_len, block_stride);                                                           \
    } while (0)
    // One instance per degree profile of the reference's six codes (LdpcPlan::prof_*, four bits per round, round 0
    // lowest; tools/ldpc_plan_check.cpp prints them): the kernel touches exactly the edge slots a round has.  The
    // API takes no other graph (code_rate selects one of the six); a plan that matches none is refused.
    // Register budgets (last argument: wavefronts per SIMD) measured with tools/ldpc_bench.py.
    auto is = [&](int rr, int vr, uint64_t rmax, uint64_t rmin, uint64_t vmax, uint64_t vmin, bool rid, bool lin) {
        return P.row_rounds == rr && P.var_rounds == vr && P.prof_rmax == rmax && P.prof_rmin == rmin &&
               P.prof_vmax == vmax && P.prof_vmin == vmin && (P.row_identity == 0) == rid && (P.linear != 0) == lin;
    };
    if (is(2, 4, 0x66ull, 0x66ull, 0x3333ull, 0x333ull, true, true))                                            // R5/6
        UH_LDPC_LAUNCH(2, 4, 0x66ull, 0x66ull, 0x3333ull, 0x333ull, true, true, 6);
    else if (is(3, 6, 0x666ull, 0x666ull, 0x333333ull, 0x33333ull, true, true))                                 // R3/4
        UH_LDPC_LAUNCH(3, 6, 0x666ull, 0x666ull, 0x333333ull, 0x33333ull, true, true, 5);
    else if (is(2, 4, 0x66ull, 0x66ull, 0x3333ull, 0x333ull, false, false))                                     // R5/6, arbitrary addresses
        UH_LDPC_LAUNCH(2, 4, 0x66ull, 0x66ull, 0x3333ull, 0x333ull, false, false, 5);
    else if (is(3, 6, 0x666ull, 0x666ull, 0x333333ull, 0x33333ull, false, false))                               // R3/4, arbitrary addresses
        UH_LDPC_LAUNCH(3, 6, 0x666ull, 0x666ull, 0x333333ull, 0x33333ull, false, false, 5);
    else if (is(4, 7, 0x6666ull, 0x4666ull, 0x3333333ull, 0x333333ull, true, true))                             // R2/3
        UH_LDPC_LAUNCH(4, 7, 0x6666ull, 0x4666ull, 0x3333333ull, 0x333333ull, true, true, 5);
    else if (is(4, 7, 0x6666ull, 0x4666ull, 0x3333333ull, 0x333333ull, true, false))                            // R2/3, arbitrary addresses
        UH_LDPC_LAUNCH(4, 7, 0x6666ull, 0x4666ull, 0x3333333ull, 0x333333ull, true, false, 4);
    else if (is(6, 6, 0x124566ull, 0x112456ull, 0x444445ull, 0x44444ull, true, false))                          // R1/2
        UH_LDPC_LAUNCH(6, 6, 0x124566ull, 0x112456ull, 0x444445ull, 0x44444ull, true, false, 4);
    else if (is(6, 6, 0x133566ull, 0x113356ull, 0x444446ull, 0x44444ull, true, false))                          // R1/3
        UH_LDPC_LAUNCH(6, 6, 0x133566ull, 0x113356ull, 0x444446ull, 0x44444ull, true, false, 4);
    else if (is(8, 3, 0x22345566ull, 0x12234566ull, 0xccdull, 0xccull, true, false))                            // R1/4
        UH_LDPC_LAUNCH(8, 3, 0x22345566ull, 0x12234566ull, 0xccdull, 0xccull, true, false, 3);
    else return ULTRA_HIP_ERR_UNSUPPORTED;
#undef UH_LDPC_LAUNCH
    UH_HIP(hipGetLastError());
    return ULTRA_HIP_OK;
}

// One-time stimulus tables of a context (TX oscillator, preamble, frame layout): built into locals and published only
// when every step has succeeded, so a failed allocation or copy cannot leave half an initialisation behind.
int stimulus_tables(ultra_hip_ctx* ctx) {
    if (ctx->d_nco_tx) return ULTRA_HIP_OK;
    const DemodConst& D = ctx->h_demod;
    const int N = D.fft, psl = N + D.cp;
    // frame layout of the harness (uo_make_batch): enough codewords to fill the frame's LLRs
    const int k = (int)ctx->geo.ldpc_k, pb = k / 8;
    const int ncw_raw = ((int)ctx->geo.llrs_per_frame + kLdpcN - 1) / kLdpcN;
    const int nraw = ncw_raw * pb;
    const int ncw_enc = (nraw * 8 + k - 1) / k;           // LDPCEncoder::encode: one codeword per k bits of input
    const int bps = D.n_data * D.bits;
    const int n_tx = (ncw_enc * kLdpcN + bps - 1) / bps;  // OFDMModulator::modulate: symbols until the bytes run out
    if (ncw_enc > dev::kStimMaxCw || nraw > dev::kStimMaxCw * 72 || n_tx < D.n_data_sym) return ULTRA_HIP_ERR_UNSUPPORTED;
    if (D.log2_fft != 10 && D.log2_fft != 9) return ULTRA_HIP_ERR_UNSUPPORTED;
    // TX oscillator: NCO(center_freq) from phase 0 at the first STS sample, one pass over the STS, one over
    // the LTS, then every sample of every data symbol incl. guards (modulator.cpp:479-532, quirk 6)
    const size_t total = (size_t)2 * psl + (size_t)n_tx * D.sym_len;
    std::vector<c32> nco(total);
    const double two_pi = 2.0 * M_PI;
    const float step = (float)((two_pi * (double)(float)ctx->cfg.center_freq) / (double)(float)ctx->cfg.sample_rate);
    float acc = 0.0f;
    for (size_t i = 0; i < total; ++i) {
        nco[i] = c32{cosf(acc), sinf(acc)};
        acc += step;
        if ((double)acc > two_pi) acc = (float)((double)acc - two_pi);
        if (acc < 0.0f) acc = (float)((double)acc + two_pi);
    }
    c32* d_nco_tx = nullptr;
    float* d_preamble = nullptr;
    auto fail = [&](int code) { if (d_nco_tx) (void)hipFree(d_nco_tx); if (d_preamble) (void)hipFree(d_preamble); return code; };
    if (hipMalloc(&d_nco_tx, total * sizeof(c32)) != hipSuccess) return fail(ULTRA_HIP_ERR_OOM);
    if (hipMemcpy(d_nco_tx, nco.data(), total * sizeof(c32), hipMemcpyHostToDevice) != hipSuccess) return fail(ULTRA_HIP_ERR_HIP);
    if (hipMalloc(&d_preamble, ((size_t)7 * psl + 2) * sizeof(float)) != hipSuccess) return fail(ULTRA_HIP_ERR_OOM);
    if (D.log2_fft == 10)
        hipLaunchKernelGGL(dev::preamble_kernel<10>, dim3(1), dim3(dev::kWave), 0, ctx->stream, ctx->d_demod,
                           ctx->d_twiddle, d_nco_tx, d_preamble, d_preamble + 7 * psl);
    else
        hipLaunchKernelGGL(dev::preamble_kernel<9>, dim3(1), dim3(dev::kWave), 0, ctx->stream, ctx->d_demod,
                           ctx->d_twiddle, d_nco_tx, d_preamble, d_preamble + 7 * psl);
    if (hipGetLastError() != hipSuccess) return fail(ULTRA_HIP_ERR_HIP);
    ctx->stim_ncw_raw = ncw_raw; ctx->stim_ncw_enc = ncw_enc; ctx->stim_tx_symbols = n_tx; ctx->stim_pre_len = 7 * psl;
    ctx->d_preamble = d_preamble;
    ctx->d_nco_tx = d_nco_tx;                              // published last: the guard of the next call
    return ULTRA_HIP_OK;
}

// payload -> encode -> modulate of frames first_frame .. + n_frames - 1 into d_audio rows (unscaled) + per-frame stats
int launch_stimulus(ultra_hip_ctx* ctx, uint64_t seed, uint64_t first_frame, size_t n_frames, float* d_audio, size_t frame_stride,
                    uint8_t* d_payload) {
    const DemodConst& D = ctx->h_demod;
    if (ctx->ws_fstats_frames < n_frames) {
        if (ctx->d_ws_fstats) { UH_HIP(uh_stream_sync(ctx->stream)); (void)hipFree(ctx->d_ws_fstats); ctx->d_ws_fstats = nullptr; }
        ctx->ws_fstats_frames = 0;
        UH_HIP(hipMalloc(&ctx->d_ws_fstats, n_frames * 2 * sizeof(float)));
        ctx->ws_fstats_frames = n_frames;
    }
    const int pb = (int)ctx->geo.ldpc_k / 8;
    const unsigned grid = (unsigned)std::min(n_frames, (size_t)ctx->cu_count * 32);
    if (D.log2_fft == 10)
        hipLaunchKernelGGL(dev::stimulus_kernel<10>, dim3(grid), dim3(dev::kWave), 0, ctx->stream, ctx->d_demod, ctx->d_plan,
                           ctx->d_twiddle, ctx->d_nco_tx, (unsigned long long)seed, (unsigned long long)first_frame,
                           (int)n_frames, ctx->stim_ncw_raw * pb, ctx->stim_ncw_enc, pb, ctx->stim_tx_symbols, d_audio,
                           frame_stride, d_payload, ctx->d_ws_fstats);
    else
        hipLaunchKernelGGL(dev::stimulus_kernel<9>, dim3(grid), dim3(dev::kWave), 0, ctx->stream, ctx->d_demod, ctx->d_plan,
                           ctx->d_twiddle, ctx->d_nco_tx, (unsigned long long)seed, (unsigned long long)first_frame,
                           (int)n_frames, ctx->stim_ncw_raw * pb, ctx->stim_ncw_enc, pb, ctx->stim_tx_symbols, d_audio,
                           frame_stride, d_payload, ctx->d_ws_fstats);
    UH_HIP(hipGetLastError());
    return ULTRA_HIP_OK;
}

}  // namespace

extern "C" {

int ultra_hip_abi_version(void) { return ULTRA_HIP_ABI_VERSION; }
unsigned long long ultra_hip_host_sync_count(void) { return g_host_syncs.load(std::memory_order_relaxed); }

#ifdef UH_MIXFFT_STAMPS
// diagnostic build only (tools/mix_fft_stalls.py): where mix_fft_kernel / mix_fft2_kernel leave their phase stamps
int ultra_hip_debug_set_stamps(ultra_hip_ctx* ctx, void* d_buf) {
    if (!ctx) return ULTRA_HIP_ERR_INVALID_ARG;
    DeviceGuard guard(ctx->device);
    unsigned long long* p = static_cast<unsigned long long*>(d_buf);
    return hipMemcpyToSymbol(HIP_SYMBOL(ultra_hip::dev::g_mix_stamps), &p, sizeof(p)) == hipSuccess ? ULTRA_HIP_OK : ULTRA_HIP_ERR_HIP;
}
#endif

#ifdef UH_ACQ_STAMPS
// diagnostic build only (tools/acquire_stalls.py): where acquire_kernel leaves its phase stamps
int ultra_hip_debug_set_acq_stamps(ultra_hip_ctx* ctx, void* d_buf) {
    if (!ctx) return ULTRA_HIP_ERR_INVALID_ARG;
    DeviceGuard guard(ctx->device);
    unsigned long long* p = static_cast<unsigned long long*>(d_buf);
    return hipMemcpyToSymbol(HIP_SYMBOL(ultra_hip::dev::g_acq_stamps), &p, sizeof(p)) == hipSuccess ? ULTRA_HIP_OK : ULTRA_HIP_ERR_HIP;
}
#endif
#ifdef UH_LDPC_STAMPS
// diagnostic build only (tools/ldpc_stalls.py): where ldpc_totals_kernel leaves its per-codeword phase times
int ultra_hip_debug_set_ldpc_stamps(ultra_hip_ctx* ctx, void* d_buf) {
    if (!ctx) return ULTRA_HIP_ERR_INVALID_ARG;
    DeviceGuard guard(ctx->device);
    unsigned long long* p = static_cast<unsigned long long*>(d_buf);
    return hipMemcpyToSymbol(HIP_SYMBOL(ultra_hip::dev::g_ldpc_stamps), &p, sizeof(p)) == hipSuccess ? ULTRA_HIP_OK : ULTRA_HIP_ERR_HIP;
}
#endif

const char* ultra_hip_strerror(int status) {
    switch (status) {
        case ULTRA_HIP_OK: return "ok";
        case ULTRA_HIP_ERR_INVALID_ARG: return "invalid argument";
        case ULTRA_HIP_ERR_UNSUPPORTED: return "configuration not supported by the HIP path";
        case ULTRA_HIP_ERR_NO_DEVICE: return "no usable HIP device";
        case ULTRA_HIP_ERR_HIP: return "HIP runtime call failed";
        case ULTRA_HIP_ERR_OOM: return "out of memory";
        default: return "unknown ultra_hip status";
    }
}

int ultra_hip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return ULTRA_HIP_ERR_NO_DEVICE;
    return n;
}

int ultra_hip_geometry_for(const ultra_hip_config* cfg, ultra_hip_geometry* geo) {
    if (!cfg || !geo) return ULTRA_HIP_ERR_INVALID_ARG;
    int rc = fill_geometry(*cfg, *geo);
    if (rc != ULTRA_HIP_OK) return rc;
    LdpcConst L;
    build_ldpc(cfg->code_rate, cfg->max_iterations, L);
    geo->ldpc_edges = (uint32_t)L.edges;
    return ULTRA_HIP_OK;
}

int ultra_hip_create(const ultra_hip_config* cfg, int device, void* stream, ultra_hip_ctx** out) {
    if (!cfg || !out) return ULTRA_HIP_ERR_INVALID_ARG;
    *out = nullptr;
    ultra_hip_geometry geo;
    int rc = fill_geometry(*cfg, geo);
    if (rc != ULTRA_HIP_OK) return rc;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return ULTRA_HIP_ERR_NO_DEVICE;
    if (device < 0 || device >= n) return ULTRA_HIP_ERR_NO_DEVICE;

    ultra_hip_ctx* ctx = new (std::nothrow) ultra_hip_ctx();
    if (!ctx) return ULTRA_HIP_ERR_OOM;
    ctx->cfg = *cfg;
    ctx->device = device;
    ctx->stream = static_cast<hipStream_t>(stream);

    std::vector<c32> nco, tw;
    rc = build_demod(*cfg, ctx->h_demod, nco, tw);
    if (rc != ULTRA_HIP_OK) { delete ctx; return rc; }
    build_ldpc(cfg->code_rate, cfg->max_iterations, ctx->h_ldpc);
    geo.ldpc_edges = (uint32_t)ctx->h_ldpc.edges;
    ctx->geo = geo;
    if (ctx->h_ldpc.edges > kLdpcMaxEdges || ctx->h_ldpc.m > kLdpcMaxChecks) { delete ctx; return ULTRA_HIP_ERR_UNSUPPORTED; }
    rc = build_ldpc_plan(ctx->h_ldpc, ctx->h_plan);
    if (rc != ULTRA_HIP_OK) { delete ctx; return rc; }
    // The two switches the product reads select its own FALL-BACK kernels (see ultra_hip_ctx::old_chain); grid-size sweeps
    // exist only in variant builds (tools/build_variants.sh).
#ifdef UH_AB_SWITCHES
    { const char* e = std::getenv("ULTRA_HIP_MIX_WG_PER_CU"); if (e && std::atoi(e) > 0) ctx->mix_wg_per_cu = std::atoi(e); }
#endif
    { const char* e = std::getenv("ULTRA_HIP_FALLBACK_CHAIN"); ctx->old_chain = (e && e[0] == '1'); }
    if (ctx->old_chain) ctx->status_flags |= ULTRA_HIP_ST_FORCED_FALLBACK_CHAIN;
    const char* force_messages = std::getenv("ULTRA_HIP_LDPC_MESSAGES");
    if (!(force_messages && force_messages[0] == '1')) (void)build_ldpc_tplan(ctx->h_ldpc, cfg->code_rate, ctx->h_tplan);
    else ctx->status_flags |= ULTRA_HIP_ST_FORCED_MESSAGE_KERNEL;
    ctx->h_tplan.max_iterations = ctx->h_ldpc.max_iterations;

    DeviceGuard guard(device);
    if (!guard.ok) { delete ctx; return ULTRA_HIP_ERR_NO_DEVICE; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
        ctx->cu_count = prop.multiProcessorCount;

    auto fail = [&](int code) { ultra_hip_destroy(ctx); return code; };
    if (hipMalloc(&ctx->d_demod, sizeof(DemodConst)) != hipSuccess) return fail(ULTRA_HIP_ERR_OOM);
    if (hipMalloc(&ctx->d_plan, sizeof(LdpcPlan)) != hipSuccess) return fail(ULTRA_HIP_ERR_OOM);
    if (hipMalloc(&ctx->d_tplan, sizeof(LdpcTPlan)) != hipSuccess) return fail(ULTRA_HIP_ERR_OOM);
    if (hipMalloc(&ctx->d_work, 16 * dev::kLdpcQueueWords * sizeof(unsigned int)) != hipSuccess) return fail(ULTRA_HIP_ERR_OOM);
    if (hipMalloc(&ctx->d_status, 4 * sizeof(unsigned)) != hipSuccess) return fail(ULTRA_HIP_ERR_OOM);
    if (hipMemset(ctx->d_status, 0, 4 * sizeof(unsigned)) != hipSuccess) return fail(ULTRA_HIP_ERR_HIP);
    if (hipMalloc(&ctx->d_nco, nco.size() * sizeof(c32)) != hipSuccess) return fail(ULTRA_HIP_ERR_OOM);
    if (hipMalloc(&ctx->d_twiddle, tw.size() * sizeof(c32)) != hipSuccess) return fail(ULTRA_HIP_ERR_OOM);
    {   // the decoder's screen: the rows' variables (<= kScreenEdges each, the parity bit's included); a graph that does not fit
        // decodes without it
        const char* e = std::getenv("ULTRA_HIP_LDPC_SCREEN");
        if (e && (e[0] == '0' || e[0] == '2') && e[1] == 0) { ctx->screen_mode = e[0] - '0'; ctx->status_flags |= ULTRA_HIP_ST_SCREEN_OVERRIDDEN; }
        std::vector<uint16_t> rv((size_t)kTPlanRowRounds * 64 * kScreenEdges, (uint16_t)0xFFFF);
        const LdpcConst& L = ctx->h_ldpc;
        bool fits = L.m <= kTPlanRowRounds * 64;
        // rows by degree, highest first (stable): round r of the pass then gathers as many edges as its FIRST row has
        std::vector<int> order((size_t)std::max(0, L.m));
        for (int i = 0; i < L.m; ++i) order[(size_t)i] = i;
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
            return (L.row_ptr[a + 1] - L.row_ptr[a]) > (L.row_ptr[b + 1] - L.row_ptr[b]);
        });
        for (int s = 0; fits && s < L.m; ++s) {
            const int i = order[(size_t)s], e0 = L.row_ptr[i], e1 = L.row_ptr[i + 1];
            if (e1 - e0 > kScreenEdges || e1 - e0 < 1) { fits = false; break; }
            for (int q = e0; q < e1; ++q) rv[(size_t)s * kScreenEdges + (q - e0)] = L.col[q];
            if (s % 64 == 0) ctx->screen_prof |= (unsigned long long)(e1 - e0) << (4 * (s / 64));
        }
        if (fits && ctx->screen_mode != 0) {
            if (hipMalloc(&ctx->d_screen_var, rv.size() * sizeof(uint16_t)) != hipSuccess) return fail(ULTRA_HIP_ERR_OOM);
            if (hipMalloc(&ctx->d_screen_pos, sizeof(LdpcScreenPos)) != hipSuccess) return fail(ULTRA_HIP_ERR_OOM);
            if (hipMemcpy(ctx->d_screen_var, rv.data(), rv.size() * sizeof(uint16_t), hipMemcpyHostToDevice) != hipSuccess)
                return fail(ULTRA_HIP_ERR_HIP);
        }
    }
    if (ctx->h_tplan.valid) {
        // ldpc_totals_kernel.h: dynamic LDS must start at LDS address 0 for the totals kernel; d_work doubles as the probe's word
        unsigned base = 1u;
        hipLaunchKernelGGL(dev::ldpc_lds_base_probe_kernel, dim3(1), dim3(dev::kLdpcThreads), (size_t)ctx->h_tplan.lds_bytes, ctx->stream, ctx->d_work);
        if (hipGetLastError() != hipSuccess || uh_stream_sync(ctx->stream) != hipSuccess ||
            hipMemcpy(&base, ctx->d_work, sizeof(base), hipMemcpyDeviceToHost) != hipSuccess)
            return fail(ULTRA_HIP_ERR_HIP);
        if (base != 0u) {
            std::fprintf(stderr, "ultra_hip: dynamic LDS starts at %u, not 0: this context decodes with the message-passing kernel\n", base);
            ctx->h_tplan.valid = 0;
            ctx->status_flags |= ULTRA_HIP_ST_LDS_PROBE_FAILED;
        }
    }
    if (hipMemcpy(ctx->d_demod, &ctx->h_demod, sizeof(DemodConst), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(ctx->d_plan, &ctx->h_plan, sizeof(LdpcPlan), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(ctx->d_tplan, &ctx->h_tplan, sizeof(LdpcTPlan), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(ctx->d_nco, nco.data(), nco.size() * sizeof(c32), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(ctx->d_twiddle, tw.data(), tw.size() * sizeof(c32), hipMemcpyHostToDevice) != hipSuccess)
        return fail(ULTRA_HIP_ERR_HIP);
    {
        std::vector<float> li, lq;
        build_lts_templates(*cfg, ctx->h_demod, nco, tw, li, lq, ctx->lts_energy_ref);
        ctx->lts_len = (uint32_t)li.size();
        if (hipMalloc(&ctx->d_lts, 2 * li.size() * sizeof(float)) != hipSuccess) return fail(ULTRA_HIP_ERR_OOM);
        if (hipMemcpy(ctx->d_lts, li.data(), li.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(ctx->d_lts + li.size(), lq.data(), lq.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
            return fail(ULTRA_HIP_ERR_HIP);
    }
    if (hipEventCreate(&ctx->ev_begin) != hipSuccess || hipEventCreate(&ctx->ev_end) != hipSuccess)
        return fail(ULTRA_HIP_ERR_HIP);
    *out = ctx;
    return ULTRA_HIP_OK;
}

void ultra_hip_destroy(ultra_hip_ctx* ctx) {
    if (!ctx) return;
    DeviceGuard guard(ctx->device);
    if (ctx->d_demod) (void)hipFree(ctx->d_demod);
    if (ctx->d_plan) (void)hipFree(ctx->d_plan);
    if (ctx->d_tplan) (void)hipFree(ctx->d_tplan);
    if (ctx->d_deint_table) (void)hipFree(ctx->d_deint_table);
    if (ctx->d_work) (void)hipFree(ctx->d_work);
    if (ctx->d_status) (void)hipFree(ctx->d_status);
    if (ctx->d_screen_var) (void)hipFree(ctx->d_screen_var);
    if (ctx->d_screen_pos) (void)hipFree(ctx->d_screen_pos);
    if (ctx->d_ws_list) (void)hipFree(ctx->d_ws_list);
    for (auto& sp : ctx->spans) { (void)hipEventDestroy(sp.e0); (void)hipEventDestroy(sp.e1); }
    for (auto e : ctx->spare_events) (void)hipEventDestroy(e);
    if (ctx->d_nco) (void)hipFree(ctx->d_nco);
    if (ctx->d_twiddle) (void)hipFree(ctx->d_twiddle);
    if (ctx->d_lts) (void)hipFree(ctx->d_lts);
    if (ctx->d_ws_acq) (void)hipFree(ctx->d_ws_acq);
    if (ctx->d_acq_gcache) (void)hipFree(ctx->d_acq_gcache);
    if (ctx->d_chirp) (void)hipFree(ctx->d_chirp);
    if (ctx->d_ws_chirp) (void)hipFree(ctx->d_ws_chirp);
    if (ctx->d_ws_frame) (void)hipFree(ctx->d_ws_frame);
    if (ctx->d_nco_tx) (void)hipFree(ctx->d_nco_tx);
    if (ctx->d_preamble) (void)hipFree(ctx->d_preamble);
    if (ctx->d_ws_fstats) (void)hipFree(ctx->d_ws_fstats);
    if (ctx->d_ws_cfo) (void)hipFree(ctx->d_ws_cfo);
    if (ctx->d_ws_llr) (void)hipFree(ctx->d_ws_llr);
    if (ctx->d_ws_state) (void)hipFree(ctx->d_ws_state);
    if (ctx->d_ws_trk) (void)hipFree(ctx->d_ws_trk);
    if (ctx->d_ws_fq) (void)hipFree(ctx->d_ws_fq);
    if (ctx->d_ws_seg) (void)hipFree(ctx->d_ws_seg);
    if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
    for (void* b : ctx->host_blocks) (void)hipHostFree(b);
    if (ctx->ev_begin) (void)hipEventDestroy(ctx->ev_begin);
    if (ctx->ev_end) (void)hipEventDestroy(ctx->ev_end);
    delete ctx;
}

int ultra_hip_reserve(ultra_hip_ctx* ctx, size_t n_frames) {
    if (!ctx || n_frames > 0x7fffffffull) return ULTRA_HIP_ERR_INVALID_ARG;
    if (n_frames == 0) return ULTRA_HIP_OK;
    DeviceGuard guard(ctx->device);
    int rc = ensure_demod_workspace(ctx, n_frames);
    if (rc == ULTRA_HIP_OK) {
        // One row per frame always; one per frame AND symbol where launch_demod keeps every symbol's bins (layouts without
        // pilots at CFO 0, and the deferred carrier half of the coherent layouts) — that multiple is best effort: without
        // it launch_demod runs the per-symbol launches on single-symbol buffers instead of failing.
        const DemodConst& D = ctx->h_demod;
        const size_t syms = (size_t)std::max(1, D.n_train + D.n_data_sym);
        const bool zero_cfo_layout = !D.presynced && D.n_pilot == 0;
        const bool deferred_layout = !ctx->old_chain && !D.differential && D.n_pilot > 0 && D.n_train == 0 && !D.presynced && D.adaptive_eq == 0;
        if ((zero_cfo_layout || deferred_layout) && syms > 1) {
            if (ensure_fq_workspace(ctx, n_frames * syms) != ULTRA_HIP_OK) (void)hipGetLastError();
            else if (deferred_layout && ensure_trk_workspace(ctx, n_frames * syms) != ULTRA_HIP_OK) (void)hipGetLastError();
        }
        rc = ensure_fq_workspace(ctx, n_frames);
    }
    if (rc == ULTRA_HIP_OK) rc = ensure_llr_workspace(ctx, n_frames);
    // the decoder's work list (ldpc_screen_kernel.h): best effort — without it such launches decode without the screen
    if (rc == ULTRA_HIP_OK && ctx->screen_mode != 0 && ctx->d_screen_pos && (n_frames >= kScreenMinCodewords || ctx->screen_mode == 2) &&
        ensure_list_workspace(ctx, n_frames) != ULTRA_HIP_OK)
        (void)hipGetLastError();
    return rc;
}

int ultra_hip_get_status(ultra_hip_ctx* ctx, ultra_hip_path_status* out) {
    if (!ctx || !out) return ULTRA_HIP_ERR_INVALID_ARG;
    DeviceGuard guard(ctx->device);
    unsigned ctl[4] = {0, 0, 0, 0};
    if (ctx->screen_launches) {
        UH_HIP(uh_stream_sync(ctx->stream));
        UH_HIP(hipMemcpy(ctl, ctx->d_status, sizeof(ctl), hipMemcpyDeviceToHost));
    }
    std::memset(out, 0, sizeof(*out));
    out->flags = ctx->status_flags;
    out->screen_launches = ctx->screen_launches;
    out->screen_sample_n = ctx->last_screen_sample_n;
    out->screen_sample_clean = ctl[dev::kScreenCtlSample];
    out->screen_gate = ctx->last_screen_gate;
    out->screen_gate_open = (ctx->screen_launches && ctl[dev::kScreenCtlSample] >= ctx->last_screen_gate) ? 1u : 0u;
    out->screen_dirty = out->screen_gate_open ? ctl[dev::kScreenCtlDirty] : 0u;
    return ULTRA_HIP_OK;
}

int ultra_hip_clear_status(ultra_hip_ctx* ctx) {
    if (!ctx) return ULTRA_HIP_ERR_INVALID_ARG;
    // what the environment forced when the context was created stays true of every launch: those bits are not cleared
    ctx->status_flags &= (ULTRA_HIP_ST_FORCED_FALLBACK_CHAIN | ULTRA_HIP_ST_FORCED_MESSAGE_KERNEL | ULTRA_HIP_ST_SCREEN_OVERRIDDEN | ULTRA_HIP_ST_LDS_PROBE_FAILED);
    return ULTRA_HIP_OK;
}

int ultra_hip_set_workspace_limit(ultra_hip_ctx* ctx, size_t bytes) {
    if (!ctx) return ULTRA_HIP_ERR_INVALID_ARG;
    DeviceGuard guard(ctx->device);
    ctx->ws_limit_bytes = bytes;
    // buffers already above the cap go back now, so the next batch really runs within it
    if (bytes != 0) {
        UH_HIP(uh_stream_sync(ctx->stream));
        const size_t fq_row = (size_t)(2 * ctx->h_demod.fq_half) * sizeof(c32), trk_row = (size_t)dev::trk_rec_floats(ctx->h_demod.n_pilot) * sizeof(float);
        if (ctx->ws_fq_rows * fq_row > bytes) { (void)hipFree(ctx->d_ws_fq); ctx->d_ws_fq = nullptr; ctx->ws_fq_rows = 0; }
        if (ctx->ws_trk_rows * trk_row > bytes) { (void)hipFree(ctx->d_ws_trk); ctx->d_ws_trk = nullptr; ctx->ws_trk_rows = 0; }
    }
    return ULTRA_HIP_OK;
}

int ultra_hip_get_geometry(const ultra_hip_ctx* ctx, ultra_hip_geometry* geo) {
    if (!ctx || !geo) return ULTRA_HIP_ERR_INVALID_ARG;
    *geo = ctx->geo;
    return ULTRA_HIP_OK;
}

int ultra_hip_get_tanner_graph(const ultra_hip_ctx* ctx, uint32_t* h_row_ptr, uint32_t* h_col_idx) {
    if (!ctx || !h_row_ptr || !h_col_idx) return ULTRA_HIP_ERR_INVALID_ARG;
    for (int i = 0; i <= ctx->h_ldpc.m; ++i) h_row_ptr[i] = ctx->h_ldpc.row_ptr[i];
    for (int e = 0; e < ctx->h_ldpc.edges; ++e) h_col_idx[e] = ctx->h_ldpc.col[e];
    return ULTRA_HIP_OK;
}

int ultra_hip_ldpc_decode_batch(ultra_hip_ctx* ctx, const float* d_llr, size_t n_cw, uint8_t* d_bytes,
                                int32_t* d_iters, uint8_t* d_ok, float* d_llr_total) {
    if (!ctx) return ULTRA_HIP_ERR_INVALID_ARG;
    if (n_cw == 0) return ULTRA_HIP_OK;
    if (!d_llr || !d_bytes || !d_iters || !d_ok || n_cw > 0x7fffffffull) return ULTRA_HIP_ERR_INVALID_ARG;
    DeviceGuard guard(ctx->device);
    return launch_ldpc(ctx, d_llr, kLdpcN, n_cw, d_bytes, d_iters, d_ok, d_llr_total);
}

int ultra_hip_demod_batch(ultra_hip_ctx* ctx, const float* d_audio, size_t frame_stride, const float* d_cfo_hz,
                          const float* d_cfo_phase, size_t n_frames, float* d_llr, float* d_state) {
    if (!ctx) return ULTRA_HIP_ERR_INVALID_ARG;
    if (n_frames == 0) return ULTRA_HIP_OK;
    if (!d_audio || !d_llr || frame_stride < ctx->geo.frame_samples || n_frames > 0x7fffffffull)
        return ULTRA_HIP_ERR_INVALID_ARG;
    DeviceGuard guard(ctx->device);
    return launch_demod(ctx, d_audio, frame_stride, d_cfo_hz, d_cfo_phase, n_frames, d_llr,
                        ctx->geo.llrs_per_frame, d_state);
}

int ultra_hip_demod_batch_strided(ultra_hip_ctx* ctx, const float* d_audio, size_t frame_stride, const float* d_cfo_hz,
                                  const float* d_cfo_phase, size_t n_frames, float* d_llr, size_t llr_stride, float* d_state) {
    if (!ctx) return ULTRA_HIP_ERR_INVALID_ARG;
    if (n_frames == 0) return ULTRA_HIP_OK;
    if (!d_audio || !d_llr || frame_stride < ctx->geo.frame_samples || llr_stride < ctx->geo.llrs_per_frame || n_frames > 0x7fffffffull)
        return ULTRA_HIP_ERR_INVALID_ARG;
    DeviceGuard guard(ctx->device);
    return launch_demod(ctx, d_audio, frame_stride, d_cfo_hz, d_cfo_phase, n_frames, d_llr, llr_stride, d_state);
}

int ultra_hip_ldpc_decode_blocks(ultra_hip_ctx* ctx, const float* d_llr, size_t llr_stride, size_t block_len, size_t block_stride,
                                 size_t n_blocks, uint8_t* d_bytes, int32_t* d_iters, uint8_t* d_ok) {
    if (!ctx) return ULTRA_HIP_ERR_INVALID_ARG;
    if (n_blocks == 0 || block_len == 0) return ULTRA_HIP_OK;
    if (!d_llr || !d_bytes || !d_iters || !d_ok || llr_stride < (size_t)kLdpcN || block_stride < block_len ||
        block_len > 0x7fffffffull || block_stride > 0x7fffffffull || n_blocks * block_len > 0x7fffffffull)
        return ULTRA_HIP_ERR_INVALID_ARG;
    DeviceGuard guard(ctx->device);
    return launch_ldpc(ctx, d_llr, llr_stride, n_blocks * block_len, d_bytes, d_iters, d_ok, nullptr, (int)block_len, (int)block_stride);
}

int ultra_hip_demod_stream_batch(ultra_hip_ctx* ctx, const float* d_audio, size_t frame_stride, const float* d_cfo_hz,
                                 const float* d_cfo_phase, size_t n_frames, uint32_t first_symbol, uint32_t n_symbols,
                                 float* d_llr, float* d_state) {
    return ultra_hip_demod_stream_batch_eq(ctx, d_audio, frame_stride, d_cfo_hz, d_cfo_phase, n_frames, first_symbol, n_symbols, d_llr,
                                           d_state, nullptr);
}

int ultra_hip_demod_stream_batch_eq(ultra_hip_ctx* ctx, const float* d_audio, size_t frame_stride, const float* d_cfo_hz,
                                    const float* d_cfo_phase, size_t n_frames, uint32_t first_symbol, uint32_t n_symbols,
                                    float* d_llr, float* d_state, float* d_equalized) {
    if (!ctx) return ULTRA_HIP_ERR_INVALID_ARG;
    if (n_frames == 0 || n_symbols == 0) return ULTRA_HIP_OK;
    const DemodConst& D = ctx->h_demod;
    const uint32_t total = (uint32_t)(D.n_train + D.n_data_sym);
    if (!d_audio || !d_llr || n_frames > 0x7fffffffull || first_symbol >= total || n_symbols > total - first_symbol ||
        frame_stride < (size_t)n_symbols * (size_t)D.sym_len)
        return ULTRA_HIP_ERR_INVALID_ARG;
    // the presynced entry's training symbols are one unit (estimateCFOFromTraining reads both): the first call takes them all
    if (D.n_train != 0 && ((first_symbol == 0 && n_symbols < (uint32_t)D.n_train) || (first_symbol != 0 && first_symbol < (uint32_t)D.n_train)))
        return ULTRA_HIP_ERR_INVALID_ARG;
    DeviceGuard guard(ctx->device);
    // the kernels address symbol s of a frame at row + s * sym_len and its LLRs at row + (s - n_train) * llrs_per_symbol
    const uint32_t first_data = first_symbol > (uint32_t)D.n_train ? first_symbol - (uint32_t)D.n_train : 0u;
    const uint32_t data_in_call = first_symbol + n_symbols > (uint32_t)D.n_train ? first_symbol + n_symbols - (uint32_t)D.n_train - first_data : 0u;
    const float* audio0 = d_audio - (size_t)first_symbol * (size_t)D.sym_len;
    float* llr0 = d_llr - (size_t)first_data * (size_t)D.llrs_per_symbol;
    return launch_demod(ctx, audio0, frame_stride, d_cfo_hz, d_cfo_phase, n_frames, llr0,
                        (size_t)data_in_call * (size_t)D.llrs_per_symbol, d_state, nullptr, (int)first_symbol, (int)n_symbols,
                        data_in_call ? reinterpret_cast<c32*>(d_equalized) : nullptr);
}

int ultra_hip_demod_stream_start(ultra_hip_ctx* ctx, int mode, const float* d_timing) {
    if (!ctx || mode < ULTRA_STREAM_START_FRESH || mode > ULTRA_STREAM_START_TIMING) return ULTRA_HIP_ERR_INVALID_ARG;
    if (mode == ULTRA_STREAM_START_TIMING && !d_timing) return ULTRA_HIP_ERR_INVALID_ARG;
    if (mode == ULTRA_STREAM_START_SYNC && ctx->h_demod.presynced) return ULTRA_HIP_ERR_INVALID_ARG;   // processPresynced resets the tracker (:868-905)
    ctx->stream_start_mode = mode;
    ctx->stream_start_timing = (mode == ULTRA_STREAM_START_TIMING) ? d_timing : nullptr;
    return ULTRA_HIP_OK;
}

int ultra_hip_stream_adopt(ultra_hip_ctx* dst, ultra_hip_ctx* src, size_t n_frames) {
    // One ultra::OFDMDemodulator object is ONE tracker whatever entry its frames come through; here the two entries are two
    // contexts with records of the same layout (demod_kernel.h, kStFloats: the compact pilot state does not depend on the entry).
    if (!dst || !src || dst == src || n_frames == 0 || n_frames > 0x7fffffffull) return ULTRA_HIP_ERR_INVALID_ARG;
    const DemodConst &A = dst->h_demod, &B = src->h_demod;
    if (dst->device != src->device || src->ws_demod_frames < n_frames || A.fft != B.fft || A.n_carriers != B.n_carriers ||
        A.n_pilot != B.n_pilot || A.n_data != B.n_data || A.modulation != B.modulation || A.differential != B.differential ||
        A.adaptive_eq != B.adaptive_eq || std::memcmp(A.pilot_slot, B.pilot_slot, sizeof(A.pilot_slot)) != 0)
        return ULTRA_HIP_ERR_INVALID_ARG;
    DeviceGuard guard(dst->device);
    { const int rc = ensure_demod_workspace(dst, n_frames); if (rc != ULTRA_HIP_OK) return rc; }
    if (src->stream != dst->stream) {                       // what src's stream still has in flight comes first
        UH_HIP(hipEventRecord(src->ev_end, src->stream));
        UH_HIP(hipStreamWaitEvent(dst->stream, src->ev_end, 0));
    }
    UH_HIP(hipMemcpyAsync(dst->d_ws_state, src->d_ws_state, n_frames * (size_t)dev::kStFloats * sizeof(float), hipMemcpyDeviceToDevice,
                          dst->stream));
    return ULTRA_HIP_OK;
}

int ultra_hip_demod_stream_set_cfo(ultra_hip_ctx* ctx, size_t frame, float cfo_hz) {
    return ultra_hip_demod_stream_set_cfo_phase(ctx, frame, cfo_hz, 0.0f);
}

int ultra_hip_demod_stream_set_cfo_phase(ultra_hip_ctx* ctx, size_t frame, float cfo_hz, float cfo_phase) {
    if (!ctx || frame >= ctx->ws_demod_frames) return ULTRA_HIP_ERR_INVALID_ARG;
    DeviceGuard guard(ctx->device);
    // OFDMDemodulator::setFrequencyOffset[WithPhase] (demodulator.cpp:805-825): freq_offset_hz = filtered = cfo, correction phase 0 or given
    const float v[3] = {cfo_hz, cfo_hz, cfo_phase};
    static_assert(dev::st_cfo == 0 && dev::st_cfo_filt == 1 && dev::st_cfo_phase == 2, "record layout");
    UH_HIP(hipMemcpyAsync(ctx->d_ws_state + frame * (size_t)dev::kStFloats, v, sizeof(v), hipMemcpyHostToDevice, ctx->stream));
    UH_HIP(uh_stream_sync(ctx->stream));
    // From the next symbol on some frame of the batch carries an offset nothing on the path estimated: the short cuts
    // launch_demod takes for frames that started without one (no phase table on layouts without pilots, the first two
    // symbols at CFO 0 on layouts with them) no longer hold for the rest of this frame.
    ctx->stream_cfo_given = true;
    return ULTRA_HIP_OK;
}

int ultra_hip_demod_decode_batch(ultra_hip_ctx* ctx, const float* d_audio, size_t frame_stride,
                                 const float* d_cfo_hz, const float* d_cfo_phase, size_t n_frames, float* d_llr,
                                 uint8_t* d_bytes, int32_t* d_iters, uint8_t* d_ok) {
    if (!ctx) return ULTRA_HIP_ERR_INVALID_ARG;
    if (n_frames == 0) return ULTRA_HIP_OK;
    if (!d_audio || !d_bytes || !d_iters || !d_ok || frame_stride < ctx->geo.frame_samples ||
        n_frames > 0x7fffffffull)
        return ULTRA_HIP_ERR_INVALID_ARG;
    // The harness decodes the first 648 soft bits of a frame; a frame that yields
    // fewer cannot be decoded (tools/test_nvis_mode.cpp:96-99 returns false).
    if (ctx->geo.llrs_per_frame < (uint32_t)kLdpcN) return ULTRA_HIP_ERR_UNSUPPORTED;
    DeviceGuard guard(ctx->device);
    float* llr = d_llr;
    if (!llr) {
        { const int rc_ws = ensure_llr_workspace(ctx, n_frames); if (rc_ws != ULTRA_HIP_OK) return rc_ws; }
        llr = ctx->d_ws_llr;
    }
    int rc = launch_demod(ctx, d_audio, frame_stride, d_cfo_hz, d_cfo_phase, n_frames, llr,
                          ctx->geo.llrs_per_frame, nullptr);
    if (rc != ULTRA_HIP_OK) return rc;
    return launch_ldpc(ctx, llr, ctx->geo.llrs_per_frame, n_frames, d_bytes, d_iters, d_ok, nullptr);
}

namespace {
constexpr size_t kAcqCacheMaxStreams = 4096;                 // 1 GB of metric caches: above that a batch is not a set of live adapters
int ensure_acq_gcache(ultra_hip_ctx* ctx, size_t n_streams) {
    if (ctx->acq_gcache_streams >= n_streams) return ULTRA_HIP_OK;
    if (ctx->d_acq_gcache) { UH_HIP(uh_stream_sync(ctx->stream)); (void)hipFree(ctx->d_acq_gcache); ctx->d_acq_gcache = nullptr; }
    ctx->acq_gcache_streams = 0;
    UH_HIP(hipMalloc(&ctx->d_acq_gcache, n_streams * (size_t)dev::kAcqGWords * sizeof(unsigned)));
    UH_HIP(hipMemsetAsync(ctx->d_acq_gcache, 0, n_streams * (size_t)dev::kAcqGWords * sizeof(unsigned), ctx->stream));   // header 0: "never used"
    ctx->acq_gcache_streams = n_streams;
    return ULTRA_HIP_OK;
}
int launch_acquire(ultra_hip_ctx* ctx, const float* d_audio, size_t stream_stride, uint32_t n_samples, uint32_t chunk,
                   size_t n_streams, uint32_t* d_found, uint32_t* d_data_start, float* d_cfo_hz, uint32_t* d_sync_offset,
                   uint32_t* d_fed_at_sync, uint32_t origin, uint32_t* d_resume, uint32_t midframe = 0u, unsigned* d_gcache = nullptr) {
    const unsigned grid = (unsigned)std::min(n_streams, (size_t)ctx->cu_count * 64);
    LaunchSpan span(ctx, ULTRA_HIP_K_ACQUIRE, n_streams);
    const float* lts_I = ctx->d_lts;
    const float* lts_Q = ctx->d_lts + ctx->lts_len;
    const float sync_threshold = ctx->cfg.sync_threshold != 0.0f ? ctx->cfg.sync_threshold : 0.80f;   // ModemConfig::sync_threshold (types.hpp:188)
    auto launch = [&](auto kernel) {
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(dev::kWave), 0, ctx->stream, ctx->d_demod, ctx->d_twiddle, lts_I, lts_Q,
                           ctx->lts_energy_ref, sync_threshold, d_audio, stream_stride, n_samples, chunk, (int)n_streams, d_found,
                           d_data_start, d_cfo_hz, d_sync_offset, d_fed_at_sync, origin, d_resume, d_gcache);
    };
    if (d_gcache && !midframe) {                             // live streams: the metric cache in HBM (acquire_kernel.h, GC)
        if (ctx->h_demod.log2_fft == 10) launch(dev::acquire_kernel<10, false, true>);
        else if (ctx->h_demod.log2_fft == 9) launch(dev::acquire_kernel<9, false, true>);
        else return ULTRA_HIP_ERR_UNSUPPORTED;
    }
    else if (ctx->h_demod.log2_fft == 10) { if (midframe) launch(dev::acquire_kernel<10, true>); else launch(dev::acquire_kernel<10, false>); }
    else if (ctx->h_demod.log2_fft == 9) { if (midframe) launch(dev::acquire_kernel<9, true>); else launch(dev::acquire_kernel<9, false>); }
    else
        return ULTRA_HIP_ERR_UNSUPPORTED;
    UH_HIP(hipGetLastError());
    return ULTRA_HIP_OK;
}
}  // namespace

int ultra_hip_acquire_batch(ultra_hip_ctx* ctx, const float* d_audio, size_t stream_stride, uint32_t n_samples,
                            uint32_t chunk, size_t n_streams, uint32_t* d_found, uint32_t* d_data_start,
                            float* d_cfo_hz, uint32_t* d_sync_offset, uint32_t* d_fed_at_sync) {
    if (!ctx) return ULTRA_HIP_ERR_INVALID_ARG;
    if (n_streams == 0) return ULTRA_HIP_OK;
    if (!d_audio || !d_found || !d_data_start || !d_cfo_hz || chunk == 0 || stream_stride < n_samples ||
        n_streams > 0x7fffffffull || n_samples > 0x3fffffffu)
        return ULTRA_HIP_ERR_INVALID_ARG;
    DeviceGuard guard(ctx->device);
    return launch_acquire(ctx, d_audio, stream_stride, n_samples, chunk, n_streams, d_found, d_data_start, d_cfo_hz,
                          d_sync_offset, d_fed_at_sync, 0u, nullptr);
}

int ultra_hip_acquire_stream_batch(ultra_hip_ctx* ctx, const float* d_audio, size_t stream_stride, uint32_t origin,
                                   uint32_t n_samples, size_t n_streams, uint32_t* d_resume, uint32_t* d_found,
                                   uint32_t* d_data_start, float* d_cfo_hz, uint32_t* d_sync_offset) {
    if (!ctx) return ULTRA_HIP_ERR_INVALID_ARG;
    if (n_streams == 0) return ULTRA_HIP_OK;
    if (!d_audio || !d_resume || !d_found || !d_data_start || !d_cfo_hz || n_samples < origin ||
        stream_stride < (size_t)(n_samples - origin) || n_streams > 0x7fffffffull || n_samples > 0x3fffffffu)
        return ULTRA_HIP_ERR_INVALID_ARG;
    DeviceGuard guard(ctx->device);
    // One process() call: everything fed since the last launch is one chunk.  The streams' metric caches live in HBM between
    // the calls (acquire_kernel.h, kAcqGCache): a guard launch (fresh streams start with an empty cache), the metrics of the
    // candidates the caches do not hold yet — one wavefront each, in parallel —, then the sequential walk, which finds them all.
    const size_t cand = (n_samples - origin) / 8u + 1u;                    // candidates per stream, at most (base >= origin)
    if (n_streams * cand > 0x3fffffffull)                                   // (a grid that large is not a live adapter's: the plain walk)
        return launch_acquire(ctx, d_audio, stream_stride, n_samples, 0xffffffffu, n_streams, d_found, d_data_start, d_cfo_hz,
                              d_sync_offset, nullptr, origin, d_resume);
    // the cache is 262 KB per stream and lives until destroy: best effort, like the decoder's work list — a call that needed no
    // memory before the cache existed does not fail for want of it (and says so: ULTRA_HIP_ST_ACQ_CACHE_UNAVAILABLE)
    if (n_streams > kAcqCacheMaxStreams || ensure_acq_gcache(ctx, n_streams) != ULTRA_HIP_OK) {
        (void)hipGetLastError();
        ctx->status_flags |= ULTRA_HIP_ST_ACQ_CACHE_UNAVAILABLE;
        return launch_acquire(ctx, d_audio, stream_stride, n_samples, 0xffffffffu, n_streams, d_found, d_data_start, d_cfo_hz,
                              d_sync_offset, nullptr, origin, d_resume);
    }
    {
        LaunchSpan span(ctx, ULTRA_HIP_K_ACQUIRE, n_streams);
        hipLaunchKernelGGL(dev::acq_cache_guard_kernel, dim3((unsigned)std::min(n_streams, (size_t)4096)), dim3(dev::kWave), 0, ctx->stream,
                           d_resume, n_samples, (int)n_streams, ctx->d_acq_gcache);
        const unsigned grid = (unsigned)(n_streams * cand);
        if (ctx->h_demod.log2_fft == 10)
            hipLaunchKernelGGL(dev::acq_prepass_kernel<10>, dim3(grid), dim3(dev::kWave), 0, ctx->stream, ctx->d_demod, ctx->d_twiddle, d_audio,
                               stream_stride, n_samples, origin, d_resume, ctx->d_acq_gcache, (unsigned)cand);
        else if (ctx->h_demod.log2_fft == 9)
            hipLaunchKernelGGL(dev::acq_prepass_kernel<9>, dim3(grid), dim3(dev::kWave), 0, ctx->stream, ctx->d_demod, ctx->d_twiddle, d_audio,
                               stream_stride, n_samples, origin, d_resume, ctx->d_acq_gcache, (unsigned)cand);
        else
            return ULTRA_HIP_ERR_UNSUPPORTED;
        UH_HIP(hipGetLastError());
    }
    return launch_acquire(ctx, d_audio, stream_stride, n_samples, 0xffffffffu, n_streams, d_found, d_data_start, d_cfo_hz,
                          d_sync_offset, nullptr, origin, d_resume, 0u, ctx->d_acq_gcache);
}

int ultra_hip_resync_stream_batch(ultra_hip_ctx* ctx, const float* d_audio, size_t stream_stride, uint32_t origin,
                                  uint32_t n_samples, size_t n_streams, const uint32_t* d_resume, uint32_t* d_found,
                                  uint32_t* d_data_start, float* d_cfo_hz, uint32_t* d_sync_offset) {
    if (!ctx) return ULTRA_HIP_ERR_INVALID_ARG;
    if (n_streams == 0) return ULTRA_HIP_OK;
    if (!d_audio || !d_resume || !d_found || !d_data_start || !d_cfo_hz || n_samples < origin ||
        stream_stride < (size_t)(n_samples - origin) || n_streams > 0x7fffffffull || n_samples > 0x3fffffffu)
        return ULTRA_HIP_ERR_INVALID_ARG;
    DeviceGuard guard(ctx->device);
    // the kernel only reads the records in this mode
    return launch_acquire(ctx, d_audio, stream_stride, n_samples, 0xffffffffu, n_streams, d_found, d_data_start, d_cfo_hz,
                          d_sync_offset, nullptr, origin, const_cast<uint32_t*>(d_resume), 1u);
}

int ultra_hip_receive_batch(ultra_hip_ctx* ctx, const float* d_audio, size_t stream_stride, uint32_t n_samples,
                            uint32_t chunk, size_t n_streams, float* d_llr, uint8_t* d_bytes, int32_t* d_iters,
                            uint8_t* d_ok, uint32_t* d_entry, float* d_cfo_hz) {
    if (!ctx) return ULTRA_HIP_ERR_INVALID_ARG;
    if (n_streams == 0) return ULTRA_HIP_OK;
    if (!d_audio || !d_bytes || !d_iters || !d_ok || n_samples < ctx->geo.frame_samples)
        return ULTRA_HIP_ERR_INVALID_ARG;
    if (ctx->cfg.entry != ULTRA_ENTRY_SYNCED || ctx->geo.llrs_per_frame < (uint32_t)kLdpcN) return ULTRA_HIP_ERR_UNSUPPORTED;
    DeviceGuard guard(ctx->device);
    if (ctx->ws_acq_frames < n_streams) {
        if (ctx->d_ws_acq) { UH_HIP(uh_stream_sync(ctx->stream)); (void)hipFree(ctx->d_ws_acq); ctx->d_ws_acq = nullptr; }
        ctx->ws_acq_frames = 0;
        UH_HIP(hipMalloc(&ctx->d_ws_acq, n_streams * 5 * sizeof(unsigned)));
        ctx->ws_acq_frames = n_streams;
    }
    unsigned* found = ctx->d_ws_acq;
    unsigned* data_start = found + n_streams;
    unsigned* entry = d_entry ? d_entry : data_start + n_streams;
    unsigned* offset = data_start + 2 * n_streams;
    float* cfo = d_cfo_hz ? d_cfo_hz : reinterpret_cast<float*>(data_start + 3 * n_streams);
    int rc = ultra_hip_acquire_batch(ctx, d_audio, stream_stride, n_samples, chunk, n_streams, found, data_start, cfo,
                                     nullptr, nullptr);
    if (rc != ULTRA_HIP_OK) return rc;
    const unsigned blocks = (unsigned)((n_streams + 255) / 256);
    hipLaunchKernelGGL(dev::frame_entry_kernel, dim3(blocks), dim3(256), 0, ctx->stream, found, data_start,
                       ctx->geo.frame_samples, n_samples, (int)n_streams, entry, offset);
    float* llr = d_llr;
    if (!llr) {
        { const int rc_ws = ensure_llr_workspace(ctx, n_streams); if (rc_ws != ULTRA_HIP_OK) return rc_ws; }
        llr = ctx->d_ws_llr;
    }
    rc = launch_demod(ctx, d_audio, stream_stride, cfo, nullptr, n_streams, llr, ctx->geo.llrs_per_frame, nullptr, offset);
    if (rc != ULTRA_HIP_OK) return rc;
    rc = launch_ldpc(ctx, llr, ctx->geo.llrs_per_frame, n_streams, d_bytes, d_iters, d_ok, nullptr);
    if (rc != ULTRA_HIP_OK) return rc;
    hipLaunchKernelGGL(dev::clear_unusable_kernel, dim3(blocks), dim3(256), 0, ctx->stream, entry, (int)n_streams,
                       d_bytes, (int)ctx->geo.decoded_bytes, d_iters, d_ok);
    UH_HIP(hipGetLastError());
    return ULTRA_HIP_OK;
}

int ultra_hip_chirp_receive_batch(ultra_hip_ctx* ctx, const float* d_audio, size_t stream_stride, uint32_t n_samples,
                                  size_t n_streams, float threshold, float* d_llr, uint8_t* d_bytes, int32_t* d_iters,
                                  uint8_t* d_ok, uint32_t* d_entry, float* d_cfo_hz) {
    if (!ctx) return ULTRA_HIP_ERR_INVALID_ARG;
    if (n_streams == 0) return ULTRA_HIP_OK;
    if (!d_audio || !d_bytes || !d_iters || !d_ok || n_samples < ctx->geo.frame_samples)
        return ULTRA_HIP_ERR_INVALID_ARG;
    if (ctx->cfg.entry != ULTRA_ENTRY_PRESYNCED || ctx->geo.llrs_per_frame < (uint32_t)kLdpcN) return ULTRA_HIP_ERR_UNSUPPORTED;
    DeviceGuard guard(ctx->device);
    if (ctx->ws_chirp_streams < n_streams) {
        if (ctx->d_ws_chirp) { UH_HIP(uh_stream_sync(ctx->stream)); (void)hipFree(ctx->d_ws_chirp); ctx->d_ws_chirp = nullptr; }
        ctx->ws_chirp_streams = 0;
        UH_HIP(hipMalloc(&ctx->d_ws_chirp, n_streams * 8 * sizeof(unsigned)));
        ctx->ws_chirp_streams = n_streams;
    }
    unsigned* detected = ctx->d_ws_chirp;
    int* start = reinterpret_cast<int*>(detected + n_streams);
    float* cfo_raw = reinterpret_cast<float*>(detected + 2 * n_streams);
    float* corr = reinterpret_cast<float*>(detected + 3 * n_streams);
    unsigned* entry = d_entry ? d_entry : detected + 4 * n_streams;
    unsigned* offset = detected + 5 * n_streams;
    float* cfo = d_cfo_hz ? d_cfo_hz : reinterpret_cast<float*>(detected + 6 * n_streams);
    float* phase = reinterpret_cast<float*>(detected + 7 * n_streams);
    int rc = ultra_hip_chirp_sync_batch(ctx, d_audio, stream_stride, n_samples, n_streams, threshold, detected, start,
                                        cfo_raw, corr, nullptr, nullptr);
    if (rc != ULTRA_HIP_OK) return rc;
    const unsigned blocks = (unsigned)((n_streams + 255) / 256);
    hipLaunchKernelGGL(dev::chirp_entry_kernel, dim3(blocks), dim3(256), 0, ctx->stream, detected, start, cfo_raw,
                       ctx->geo.frame_samples, n_samples, ctx->cfg.sample_rate, (int)n_streams, entry, offset, cfo, phase);
    float* llr = d_llr;
    if (!llr) {
        { const int rc_ws = ensure_llr_workspace(ctx, n_streams); if (rc_ws != ULTRA_HIP_OK) return rc_ws; }
        llr = ctx->d_ws_llr;
    }
    rc = launch_demod(ctx, d_audio, stream_stride, cfo, phase, n_streams, llr, ctx->geo.llrs_per_frame, nullptr, offset);
    if (rc != ULTRA_HIP_OK) return rc;
    rc = launch_ldpc(ctx, llr, ctx->geo.llrs_per_frame, n_streams, d_bytes, d_iters, d_ok, nullptr);
    if (rc != ULTRA_HIP_OK) return rc;
    hipLaunchKernelGGL(dev::clear_unusable_kernel, dim3(blocks), dim3(256), 0, ctx->stream, entry, (int)n_streams,
                       d_bytes, (int)ctx->geo.decoded_bytes, d_iters, d_ok);
    UH_HIP(hipGetLastError());
    return ULTRA_HIP_OK;
}

int ultra_hip_decode_frames_batch(ultra_hip_ctx* ctx, const float* d_soft, size_t frame_stride, uint32_t n_soft,
                                  size_t n_frames, ultra_hip_frame_result* d_results, uint8_t* d_frame_data,
                                  size_t frame_data_stride) {
    if (!ctx) return ULTRA_HIP_ERR_INVALID_ARG;
    if (n_frames == 0) return ULTRA_HIP_OK;
    const size_t num_cw = n_soft / (uint32_t)kLdpcN;
    // v2::getBytesPerCodeword (src/protocol/frame_v2.hpp:551-567): its own table, 216 bits for R1/3 although
    // the codec falls back to k = 324 for that rate
    static const uint32_t kV2InfoBits[6] = {162, 216, 324, 432, 486, 540};
    if (ctx->cfg.code_rate > 5) return ULTRA_HIP_ERR_UNSUPPORTED;
    const uint32_t bytes_per_cw = kV2InfoBits[ctx->cfg.code_rate] / 8;
    if (!d_soft || !d_results || !d_frame_data || frame_stride < n_soft || n_frames > 0x7fffffffull ||
        frame_data_stride < num_cw * bytes_per_cw || n_frames * std::max<size_t>(num_cw, 1) > 0x7fffffffull)
        return ULTRA_HIP_ERR_INVALID_ARG;
    DeviceGuard guard(ctx->device);
    const size_t total_cw = n_frames * num_cw, db = ctx->geo.decoded_bytes;
    if (ctx->ws_frame_cw < total_cw) {
        if (ctx->d_ws_frame) { UH_HIP(uh_stream_sync(ctx->stream)); (void)hipFree(ctx->d_ws_frame); ctx->d_ws_frame = nullptr; }
        ctx->ws_frame_cw = 0;
        UH_HIP(hipMalloc(&ctx->d_ws_frame, total_cw * (db + 1 + sizeof(int32_t)) + 16));
        ctx->ws_frame_cw = total_cw;
    }
    int32_t* iters = reinterpret_cast<int32_t*>(ctx->d_ws_frame);            // 4-byte aligned part first
    uint8_t* bytes = ctx->d_ws_frame + total_cw * sizeof(int32_t);
    uint8_t* okv = bytes + total_cw * db;
    size_t idx_frame = num_cw, idx_cw = 1;
    if (num_cw > 0) {
        if (frame_stride == num_cw * (size_t)kLdpcN) {                       // codewords back to back: one launch
            const int rc = launch_ldpc(ctx, d_soft, kLdpcN, total_cw, bytes, iters, okv, nullptr);
            if (rc != ULTRA_HIP_OK) return rc;
        } else {                                                            // one launch per codeword position
            idx_frame = 1; idx_cw = n_frames;
            for (size_t c = 0; c < num_cw; ++c) {
                const int rc = launch_ldpc(ctx, d_soft + c * kLdpcN, frame_stride, n_frames, bytes + c * n_frames * db,
                                           iters + c * n_frames, okv + c * n_frames, nullptr);
                if (rc != ULTRA_HIP_OK) return rc;
            }
        }
    }
    const unsigned blocks = (unsigned)((n_frames + 63) / 64);
    hipLaunchKernelGGL(dev::frame_assemble_kernel, dim3(blocks), dim3(64), 0, ctx->stream, d_soft, frame_stride, n_soft,
                       (int)n_frames, bytes, okv, (int)db, (int)bytes_per_cw, idx_frame, idx_cw,
                       reinterpret_cast<int32_t*>(d_results), d_frame_data, frame_data_stride);
    UH_HIP(hipGetLastError());
    return ULTRA_HIP_OK;
}

int ultra_hip_make_batch(ultra_hip_ctx* ctx, uint64_t seed, uint64_t first_frame, size_t n_frames, int channel_kind,
                         float snr_db, float delay_ms, float doppler_hz, float* d_audio, size_t frame_stride,
                         uint8_t* d_payload) {
    if (!ctx) return ULTRA_HIP_ERR_INVALID_ARG;
    if (n_frames == 0) return ULTRA_HIP_OK;
    if (!d_audio || !d_payload || frame_stride < ctx->geo.frame_samples || channel_kind < 0 || channel_kind > 2 ||
        n_frames > 0x7fffffffull)
        return ULTRA_HIP_ERR_INVALID_ARG;
    if (ctx->cfg.entry != ULTRA_ENTRY_SYNCED) return ULTRA_HIP_ERR_UNSUPPORTED;
    DeviceGuard guard(ctx->device);
    const DemodConst& D = ctx->h_demod;
    const int N = D.fft, psl = N + D.cp;
    // every check before the first launch (a kernel launched with half-initialised tables could fault)
    const size_t lds = (channel_kind == 2) ? (size_t)D.frame_samples * sizeof(float) : 0;
    if (lds > 64 * 1024) return ULTRA_HIP_ERR_UNSUPPORTED;
    { const int rc = stimulus_tables(ctx); if (rc != ULTRA_HIP_OK) return rc; }
    { const int rc = launch_stimulus(ctx, seed, first_frame, n_frames, d_audio, frame_stride, d_payload); if (rc != ULTRA_HIP_OK) return rc; }
    const unsigned grid = (unsigned)std::min(n_frames, (size_t)ctx->cu_count * 32);
    // channel
    const float fs = (float)ctx->cfg.sample_rate;
    const int delay_samples = (int)(size_t)(delay_ms * fs / 1000.0f);
    const float normalized_doppler = doppler_hz / fs;
    const float fading_alpha = (float)(1.0 - std::exp((double)-2.0f * M_PI * (double)normalized_doppler));
    const int total_len = ctx->stim_pre_len + ctx->stim_tx_symbols * D.sym_len;
    hipLaunchKernelGGL(dev::channel_kernel, dim3(grid), dim3(dev::kWave), lds, ctx->stream, ctx->d_demod, channel_kind,
                       snr_db, delay_samples, fading_alpha, 0.707f, 0.707f, (unsigned long long)seed,
                       (unsigned long long)first_frame, (int)n_frames, ctx->stim_pre_len, total_len, ctx->d_preamble,
                       ctx->d_preamble + 7 * psl, ctx->d_ws_fstats, d_audio, frame_stride);
    UH_HIP(hipGetLastError());
    return ULTRA_HIP_OK;
}

int ultra_hip_channel_cfo_batch(ultra_hip_ctx* ctx, const float* d_in, size_t in_stride, float* d_out, size_t out_stride,
                                uint32_t n_samples, size_t n_frames, float cfo_hz) {
    if (!ctx) return ULTRA_HIP_ERR_INVALID_ARG;
    if (n_frames == 0 || n_samples == 0) return ULTRA_HIP_OK;
    if (!d_in || !d_out || d_in == d_out || in_stride < n_samples || out_stride < n_samples || n_frames > 0x7fffffffull ||
        n_samples > 0x3fffffffu || !(cfo_hz == cfo_hz))
        return ULTRA_HIP_ERR_INVALID_ARG;
    DeviceGuard guard(ctx->device);
    // process() shifts only when abs(cfo) > 0.001 (hf_channel.hpp:163), applyCFO only buffers of 256 samples or more (:173)
    if (!(std::fabs(cfo_hz) > 0.001f) || n_samples < 256) {
        UH_HIP(hipMemcpy2DAsync(d_out, out_stride * sizeof(float), d_in, in_stride * sizeof(float), (size_t)n_samples * sizeof(float),
                                n_frames, hipMemcpyDeviceToDevice, ctx->stream));
        return ULTRA_HIP_OK;
    }
    if (ctx->ws_cfo_samples < n_samples) {
        if (ctx->d_ws_cfo) { UH_HIP(uh_stream_sync(ctx->stream)); (void)hipFree(ctx->d_ws_cfo); ctx->d_ws_cfo = nullptr; }
        ctx->ws_cfo_samples = 0;
        UH_HIP(hipMalloc(&ctx->d_ws_cfo, 2 * (size_t)n_samples * sizeof(c32)));
        ctx->ws_cfo_samples = n_samples;
    }
    const uint32_t fs_u = ctx->cfg.sample_rate;
    // cfo_phase_inc_ = 2.0f * M_PI * actual_cfo_hz_ / config.sample_rate (:101): double arithmetic, narrowed once
    const float phase_inc = (float)((((double)2.0f * M_PI) * (double)cfo_hz) / (double)fs_u);
    const double two_pi_fc = ((double)2.0f * M_PI) * (double)1500.0f;
    c32* mixer = ctx->d_ws_cfo;
    c32* rotator = ctx->d_ws_cfo + n_samples;
    hipLaunchKernelGGL(dev::cfo_tables_kernel, dim3((n_samples + 255) / 256), dim3(256), 0, ctx->stream, phase_inc, (float)fs_u,
                       two_pi_fc, (int)n_samples, mixer, rotator);
    hipLaunchKernelGGL(dev::cfo_shift_kernel, dim3((unsigned)((n_frames + 63) / 64)), dim3(64), 0, ctx->stream, d_in, in_stride, d_out,
                       out_stride, (int)n_samples, (int)n_frames, mixer, rotator);
    UH_HIP(hipGetLastError());
    return ULTRA_HIP_OK;
}

int ultra_hip_make_raw_batch(ultra_hip_ctx* ctx, uint64_t seed, uint64_t first_frame, size_t n_streams, int channel_kind,
                             float snr_db, uint32_t lead, uint32_t tail, float* d_audio, size_t stream_stride,
                             uint8_t* d_payload) {
    if (channel_kind < 0 || channel_kind > 1) return ULTRA_HIP_ERR_INVALID_ARG;
    return ultra_hip_make_raw_batch_channel(ctx, seed, first_frame, n_streams, channel_kind, snr_db, 0.0f, 0.0f, lead, tail, d_audio,
                                            stream_stride, d_payload);
}

int ultra_hip_make_raw_batch_channel(ultra_hip_ctx* ctx, uint64_t seed, uint64_t first_frame, size_t n_streams, int channel_kind,
                                     float snr_db, float delay_ms, float doppler_hz, uint32_t lead, uint32_t tail, float* d_audio,
                                     size_t stream_stride, uint8_t* d_payload) {
    if (!ctx) return ULTRA_HIP_ERR_INVALID_ARG;
    if (n_streams == 0) return ULTRA_HIP_OK;
    if (ctx->cfg.entry != ULTRA_ENTRY_SYNCED) return ULTRA_HIP_ERR_UNSUPPORTED;
    const DemodConst& D = ctx->h_demod;
    const int psl = D.fft + D.cp;
    const size_t n_out = (size_t)lead + (size_t)7 * psl + ctx->geo.frame_samples + tail;
    if (!d_audio || !d_payload || stream_stride < n_out || channel_kind < 0 || channel_kind > 2 || n_streams > 0x7fffffffull ||
        n_out > 0x3fffffffull)
        return ULTRA_HIP_ERR_INVALID_ARG;
    // Watterson: the scaled transmission (preamble + data symbols) sits in LDS for the delayed tap
    const size_t raw_lds = (channel_kind == 2) ? ((size_t)7 * psl + ctx->geo.frame_samples) * sizeof(float) : 0;
    if (raw_lds > 64 * 1024 || (channel_kind == 2 && !(doppler_hz > 0.0f && delay_ms >= 0.0f))) return ULTRA_HIP_ERR_UNSUPPORTED;
    DeviceGuard guard(ctx->device);
    { const int rc = stimulus_tables(ctx); if (rc != ULTRA_HIP_OK) return rc; }
    // the modulator's data symbols land where the stream has them: behind the lead and the preamble
    { const int rc = launch_stimulus(ctx, seed, first_frame, n_streams, d_audio + lead + ctx->stim_pre_len, stream_stride, d_payload);
      if (rc != ULTRA_HIP_OK) return rc; }
    const unsigned grid = (unsigned)std::min(n_streams, (size_t)ctx->cu_count * 32);
    const int total_len = ctx->stim_pre_len + ctx->stim_tx_symbols * D.sym_len;
    const float fs = (float)ctx->cfg.sample_rate;                           // the channel's constants as in ultra_hip_make_batch
    const int delay_samples = (int)(size_t)(delay_ms * fs / 1000.0f);
    const float fading_alpha = (float)(1.0 - std::exp((double)-2.0f * M_PI * (double)(doppler_hz / fs)));
    hipLaunchKernelGGL(dev::raw_stream_kernel, dim3(grid), dim3(dev::kWave), raw_lds, ctx->stream, ctx->d_demod, channel_kind, snr_db,
                       delay_samples, fading_alpha, 0.707f, 0.707f,
                       (unsigned long long)seed, (unsigned long long)first_frame, (int)n_streams, (int)lead, ctx->stim_pre_len,
                       (int)tail, total_len, ctx->d_preamble, ctx->d_preamble + 7 * psl, ctx->d_ws_fstats, d_audio, stream_stride);
    UH_HIP(hipGetLastError());
    return ULTRA_HIP_OK;
}

int ultra_hip_make_llr_batch(ultra_hip_ctx* ctx, uint64_t seed, uint64_t first_cw, size_t n_cw, float esn0_db,
                             float* d_llr, uint8_t* d_payload) {
    if (!ctx) return ULTRA_HIP_ERR_INVALID_ARG;
    if (n_cw == 0) return ULTRA_HIP_OK;
    if (!d_llr || !d_payload || n_cw > 0x7fffffffull || !(esn0_db > -100.0f && esn0_db < 100.0f))
        return ULTRA_HIP_ERR_INVALID_ARG;
    DeviceGuard guard(ctx->device);
    // sigma^2 = N0 / 2 = 1 / (2 Es/N0) for unit-energy BPSK; both as floats, from one double evaluation
    const double esn0 = std::pow(10.0, (double)esn0_db / 10.0);
    const float sigma2 = (float)(1.0 / (2.0 * esn0));
    const float sigma = (float)std::sqrt(1.0 / (2.0 * esn0));
    const unsigned grid = (unsigned)std::min(n_cw, (size_t)ctx->cu_count * 64);
    hipLaunchKernelGGL(dev::llr_stimulus_kernel, dim3(grid), dim3(dev::kWave), 0, ctx->stream, ctx->d_plan,
                       (unsigned long long)seed, (unsigned long long)first_cw, (int)n_cw, (int)ctx->geo.ldpc_k / 8, sigma,
                       sigma2, d_llr, d_payload);
    UH_HIP(hipGetLastError());
    return ULTRA_HIP_OK;
}

int ultra_hip_chirp_sync_batch(ultra_hip_ctx* ctx, const float* d_audio, size_t stream_stride, uint32_t n_samples,
                               size_t n_streams, float threshold, uint32_t* d_detected, int32_t* d_start_sample,
                               float* d_cfo_hz, float* d_correlation, int32_t* d_up_chirp_start,
                               int32_t* d_down_chirp_start) {
    if (!ctx) return ULTRA_HIP_ERR_INVALID_ARG;
    if (n_streams == 0) return ULTRA_HIP_OK;
    if (!d_audio || !d_detected || !d_start_sample || !d_cfo_hz || !d_correlation || stream_stride < n_samples ||
        n_streams > 0x7fffffffull || n_samples > 0x3fffffffu)
        return ULTRA_HIP_ERR_INVALID_ARG;
    DeviceGuard guard(ctx->device);
    if (!ctx->d_chirp) {
        build_chirp_templates(ctx->cfg.sample_rate, ctx->h_chirp);
        const size_t len = (size_t)ctx->h_chirp.len;
        const size_t padded = len + dev::kChirpGroup;         // the kernel prefetches one group of taps ahead
        std::vector<float> pairs(4 * padded, 0.0f);           // (cos, sin) per tap: up chirp, then down chirp
        for (size_t i = 0; i < len; ++i) {
            pairs[2 * i] = ctx->h_chirp.up_cos[i]; pairs[2 * i + 1] = ctx->h_chirp.up_sin[i];
            pairs[2 * (padded + i)] = ctx->h_chirp.dn_cos[i]; pairs[2 * (padded + i) + 1] = ctx->h_chirp.dn_sin[i];
        }
        UH_HIP(hipMalloc(&ctx->d_chirp, pairs.size() * sizeof(float)));
        UH_HIP(hipMemcpy(ctx->d_chirp, pairs.data(), pairs.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    dev::ChirpTemplates T;
    const size_t len = (size_t)ctx->h_chirp.len;
    T.up = reinterpret_cast<const float2*>(ctx->d_chirp); T.dn = T.up + len + dev::kChirpGroup;
    T.e_up = ctx->h_chirp.e_up; T.e_dn = ctx->h_chirp.e_dn; T.len = ctx->h_chirp.len; T.gap = ctx->h_chirp.gap;
    T.start_extra = ctx->h_chirp.start_extra; T.cfo_to_samples = ctx->h_chirp.cfo_to_samples;
    const unsigned grid = (unsigned)std::min(n_streams, (size_t)ctx->cu_count * 9);
    LaunchSpan span(ctx, ULTRA_HIP_K_CHIRP);
    hipLaunchKernelGGL(dev::chirp_sync_kernel, dim3(grid), dim3(dev::kWave), 0, ctx->stream, T, d_audio, stream_stride,
                       (int)n_samples, (int)n_streams, threshold, d_detected, d_start_sample, d_cfo_hz, d_correlation,
                       d_up_chirp_start, d_down_chirp_start);
    UH_HIP(hipGetLastError());
    return ULTRA_HIP_OK;
}

int ultra_hip_count_errors(ultra_hip_ctx* ctx, const uint8_t* d_bytes, const int32_t* d_iters, const uint8_t* d_ok,
                           const uint8_t* d_payload, size_t payload_bytes, size_t n_frames,
                           ultra_hip_counters* d_counters) {
    if (!ctx) return ULTRA_HIP_ERR_INVALID_ARG;
    if (n_frames == 0) return ULTRA_HIP_OK;
    if (!d_bytes || !d_iters || !d_ok || !d_payload || !d_counters || payload_bytes == 0 ||
        payload_bytes > ctx->geo.decoded_bytes || n_frames > 0x7fffffffull)
        return ULTRA_HIP_ERR_INVALID_ARG;
    DeviceGuard guard(ctx->device);
    const unsigned grid = (unsigned)std::min<size_t>((n_frames + 255) / 256, (size_t)ctx->cu_count * 8);
    LaunchSpan span(ctx, ULTRA_HIP_K_COUNT);
    hipLaunchKernelGGL(dev::count_errors_kernel, dim3(grid), dim3(256), 0, ctx->stream, d_bytes,
                       (size_t)ctx->geo.decoded_bytes, d_iters, d_ok, d_payload, (int)payload_bytes, (int)n_frames,
                       reinterpret_cast<unsigned long long*>(d_counters));
    UH_HIP(hipGetLastError());
    return ULTRA_HIP_OK;
}

int ultra_hip_count_errors_points(ultra_hip_ctx* ctx, const uint8_t* d_bytes, const int32_t* d_iters, const uint8_t* d_ok,
                                  const uint8_t* d_payload, size_t payload_bytes, size_t n_points, size_t frames_per_point,
                                  ultra_hip_counters* d_counters) {
    if (!ctx) return ULTRA_HIP_ERR_INVALID_ARG;
    if (n_points == 0 || frames_per_point == 0) return ULTRA_HIP_OK;
    if (!d_bytes || !d_iters || !d_ok || !d_payload || !d_counters || payload_bytes == 0 ||
        payload_bytes > ctx->geo.decoded_bytes || frames_per_point > 0x7fffffffull || n_points > 65535)
        return ULTRA_HIP_ERR_INVALID_ARG;
    DeviceGuard guard(ctx->device);
    const unsigned gx = (unsigned)std::min<size_t>((frames_per_point + 255) / 256, (size_t)ctx->cu_count * 8);
    LaunchSpan span(ctx, ULTRA_HIP_K_COUNT);
    hipLaunchKernelGGL(dev::count_errors_kernel, dim3(gx, (unsigned)n_points), dim3(256), 0, ctx->stream, d_bytes,
                       (size_t)ctx->geo.decoded_bytes, d_iters, d_ok, d_payload, (int)payload_bytes, (int)frames_per_point,
                       reinterpret_cast<unsigned long long*>(d_counters));
    UH_HIP(hipGetLastError());
    return ULTRA_HIP_OK;
}

int ultra_hip_set_deinterleave(ultra_hip_ctx* ctx, uint32_t bits_per_symbol) {
    if (!ctx) return ULTRA_HIP_ERR_INVALID_ARG;
    if (bits_per_symbol >= (uint32_t)kLdpcN) return ULTRA_HIP_ERR_INVALID_ARG;
    ctx->deint_step = bits_per_symbol ? channel_interleaver_step(bits_per_symbol, (uint32_t)kLdpcN) : 1u;
    ctx->screen_pos_stale = true;
    return ULTRA_HIP_OK;
}

int ultra_hip_channel_interleaver_step(uint32_t bits_per_symbol, uint32_t total, uint32_t* step) {
    if (!step || bits_per_symbol == 0 || total == 0) return ULTRA_HIP_ERR_INVALID_ARG;
    *step = channel_interleaver_step(bits_per_symbol, total);
    return ULTRA_HIP_OK;
}

int ultra_hip_set_deinterleave_table(ultra_hip_ctx* ctx, const uint16_t* h_index, uint32_t n) {
    if (!ctx) return ULTRA_HIP_ERR_INVALID_ARG;
    DeviceGuard guard(ctx->device);
    ctx->screen_pos_stale = true;
    if (!h_index || n == 0) {                                   // off: back to the step (ultra_hip_set_deinterleave)
        if (ctx->d_deint_table) { UH_HIP(uh_stream_sync(ctx->stream)); (void)hipFree(ctx->d_deint_table); ctx->d_deint_table = nullptr; }
        return ULTRA_HIP_OK;
    }
    if (n != (uint32_t)kLdpcN) return ULTRA_HIP_ERR_INVALID_ARG;
    for (uint32_t j = 0; j < n; ++j) if (h_index[j] >= (uint16_t)kLdpcN) return ULTRA_HIP_ERR_INVALID_ARG;   // the kernel gathers through it
    if (!ctx->d_deint_table) UH_HIP(hipMalloc(&ctx->d_deint_table, kLdpcN * sizeof(uint16_t)));
    UH_HIP(hipMemcpyAsync(ctx->d_deint_table, h_index, kLdpcN * sizeof(uint16_t), hipMemcpyHostToDevice, ctx->stream));
    UH_HIP(uh_stream_sync(ctx->stream));
    return ULTRA_HIP_OK;
}

int ultra_hip_profile_enable(ultra_hip_ctx* ctx, int enable) {
    if (!ctx) return ULTRA_HIP_ERR_INVALID_ARG;
    ctx->profiling = enable != 0;
    return ULTRA_HIP_OK;
}

int ultra_hip_profile_read(ultra_hip_ctx* ctx, float* ms, uint32_t* launches) {
    // the nine classes of ABI <= 7: the rotating transform's launches are part of ULTRA_HIP_K_MIX_FFT here
    if (!ctx || !ms || !launches) return ULTRA_HIP_ERR_INVALID_ARG;
    float m[ULTRA_HIP_K_N2]; uint32_t l[ULTRA_HIP_K_N2]; uint64_t it[ULTRA_HIP_K_N2];
    const int rc = ultra_hip_profile_read_items(ctx, m, l, it);
    if (rc != ULTRA_HIP_OK) return rc;
    for (int k = 0; k < ULTRA_HIP_K_N; ++k) { ms[k] = m[k]; launches[k] = l[k]; }
    ms[ULTRA_HIP_K_MIX_FFT] += m[ULTRA_HIP_K_MIX_FFT_ROT]; launches[ULTRA_HIP_K_MIX_FFT] += l[ULTRA_HIP_K_MIX_FFT_ROT];
    return ULTRA_HIP_OK;
}

int ultra_hip_profile_read_items(ultra_hip_ctx* ctx, float* ms, uint32_t* launches, uint64_t* items) {
    if (!ctx || !ms || !launches || !items) return ULTRA_HIP_ERR_INVALID_ARG;
    DeviceGuard guard(ctx->device);
    for (int k = 0; k < ULTRA_HIP_K_N2; ++k) { ms[k] = 0.0f; launches[k] = 0; items[k] = 0; }
    for (const auto& sp : ctx->spans) {
        float t = 0.0f;
        UH_HIP(hipEventSynchronize(sp.e1));
        UH_HIP(hipEventElapsedTime(&t, sp.e0, sp.e1));
        if (sp.kind >= 0 && sp.kind < ULTRA_HIP_K_N2) { ms[sp.kind] += t; launches[sp.kind]++; items[sp.kind] += sp.items; }
        ctx->spare_events.push_back(sp.e0);
        ctx->spare_events.push_back(sp.e1);
    }
    ctx->spans.clear();
    return ULTRA_HIP_OK;
}

// RCCL is bound at first use (dlopen): single-GPU users never load it, and the library has no link-time
// dependency on a collective runtime.
namespace {
typedef int (*rccl_allreduce_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
rccl_allreduce_fn rccl_allreduce() {
    static rccl_allreduce_fn fn = []() -> rccl_allreduce_fn {
        void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        return h ? reinterpret_cast<rccl_allreduce_fn>(dlsym(h, "ncclAllReduce")) : nullptr;
    }();
    return fn;
}
}  // namespace

int ultra_hip_counters_allreduce(ultra_hip_ctx* ctx, void* rccl_comm, ultra_hip_counters* d_counters) {
    if (!ctx || !rccl_comm || !d_counters) return ULTRA_HIP_ERR_INVALID_ARG;
    rccl_allreduce_fn fn = rccl_allreduce();
    if (!fn) return ULTRA_HIP_ERR_UNSUPPORTED;
    DeviceGuard guard(ctx->device);
    // ncclAllReduce(send, recv, count, ncclUint64 = 5, ncclSum = 0, comm, stream), in place
    const int rc = fn(d_counters, d_counters, sizeof(ultra_hip_counters) / sizeof(uint64_t), 5, 0, rccl_comm, ctx->stream);
    return rc == 0 ? ULTRA_HIP_OK : ULTRA_HIP_ERR_HIP;
}

int ultra_hip_synchronize(ultra_hip_ctx* ctx) {
    if (!ctx) return ULTRA_HIP_ERR_INVALID_ARG;
    DeviceGuard guard(ctx->device);
    UH_HIP(uh_stream_sync(ctx->stream));
    return ULTRA_HIP_OK;
}

int ultra_hip_timer_begin(ultra_hip_ctx* ctx) {
    if (!ctx) return ULTRA_HIP_ERR_INVALID_ARG;
    DeviceGuard guard(ctx->device);
    UH_HIP(hipEventRecord(ctx->ev_begin, ctx->stream));
    return ULTRA_HIP_OK;
}

int ultra_hip_timer_end(ultra_hip_ctx* ctx, float* ms) {
    if (!ctx || !ms) return ULTRA_HIP_ERR_INVALID_ARG;
    DeviceGuard guard(ctx->device);
    UH_HIP(hipEventRecord(ctx->ev_end, ctx->stream));
    UH_HIP(hipEventSynchronize(ctx->ev_end));
    UH_HIP(hipEventElapsedTime(ms, ctx->ev_begin, ctx->ev_end));
    return ULTRA_HIP_OK;
}

int ultra_hip_malloc(ultra_hip_ctx* ctx, size_t bytes, void** d_ptr) {
    if (!ctx || !d_ptr) return ULTRA_HIP_ERR_INVALID_ARG;
    DeviceGuard guard(ctx->device);
    UH_HIP(hipMalloc(d_ptr, bytes ? bytes : 1));
    return ULTRA_HIP_OK;
}

int ultra_hip_free(ultra_hip_ctx* ctx, void* d_ptr) {
    if (!ctx) return ULTRA_HIP_ERR_INVALID_ARG;
    DeviceGuard guard(ctx->device);
    if (d_ptr) UH_HIP(hipFree(d_ptr));
    return ULTRA_HIP_OK;
}

int ultra_hip_memcpy_h2d(ultra_hip_ctx* ctx, void* d_dst, const void* h_src, size_t bytes) {
    if (!ctx || (bytes && (!d_dst || !h_src))) return ULTRA_HIP_ERR_INVALID_ARG;
    DeviceGuard guard(ctx->device);
    UH_HIP(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
    UH_HIP(uh_stream_sync(ctx->stream));
    return ULTRA_HIP_OK;
}

namespace {
constexpr size_t kStageRing = size_t(1) << 20;
// a slot of the pinned staging ring holding a copy of [h_src, h_src + bytes): nullptr if the ring cannot be had
char* stage_slot(ultra_hip_ctx* ctx, const void* h_src, size_t bytes) {
    if (!ctx->h_stage) {
        void* p = nullptr; void* d = nullptr;
        if (hipHostMalloc(&p, kStageRing, hipHostMallocMapped) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        if (hipHostGetDevicePointer(&d, p, 0) != hipSuccess) { (void)hipGetLastError(); d = nullptr; }
        ctx->h_stage = static_cast<char*>(p); ctx->d_stage = static_cast<char*>(d); ctx->stage_cap = kStageRing; ctx->stage_off = 0;
    }
    const size_t need = (bytes + 255) & ~size_t(255);
    if (ctx->stage_off + need > ctx->stage_cap) {              // wrap: the ring's earlier copies (and readers) must have left it
        if (uh_stream_sync(ctx->stream) != hipSuccess) return nullptr;
        ctx->stage_off = 0;
    }
    char* slot = ctx->h_stage + ctx->stage_off;
    std::memcpy(slot, h_src, bytes);                           // the caller's buffer is free again when this call returns
    ctx->stage_off += need;
    return slot;
}
}  // namespace

int ultra_hip_memcpy_h2d_async(ultra_hip_ctx* ctx, void* d_dst, const void* h_src, size_t bytes) {
    if (!ctx || (bytes && (!d_dst || !h_src))) return ULTRA_HIP_ERR_INVALID_ARG;
    if (bytes == 0) return ULTRA_HIP_OK;
    if (bytes > kStageRing / 4) return ultra_hip_memcpy_h2d(ctx, d_dst, h_src, bytes);      // large transfers: the blocking copy
    DeviceGuard guard(ctx->device);
    char* slot = stage_slot(ctx, h_src, bytes);
    if (!slot) return ultra_hip_memcpy_h2d(ctx, d_dst, h_src, bytes);
    UH_HIP(hipMemcpyAsync(d_dst, slot, bytes, hipMemcpyHostToDevice, ctx->stream));
    return ULTRA_HIP_OK;
}

int ultra_hip_stage_input(ultra_hip_ctx* ctx, const void* h_src, size_t bytes, void** d_view) {
    if (!ctx || !h_src || !d_view || bytes == 0 || bytes > kStageRing / 4) return ULTRA_HIP_ERR_INVALID_ARG;
    DeviceGuard guard(ctx->device);
    char* slot = stage_slot(ctx, h_src, bytes);
    if (!slot || !ctx->d_stage) return ULTRA_HIP_ERR_HIP;
    *d_view = ctx->d_stage + (slot - ctx->h_stage);
    return ULTRA_HIP_OK;
}

int ultra_hip_host_block(ultra_hip_ctx* ctx, size_t bytes, void** h_ptr, void** d_ptr) {
    if (!ctx || !h_ptr || !d_ptr || bytes == 0 || bytes > (size_t(64) << 20)) return ULTRA_HIP_ERR_INVALID_ARG;
    DeviceGuard guard(ctx->device);
    void* p = nullptr; void* d = nullptr;
    UH_HIP(hipHostMalloc(&p, bytes, hipHostMallocMapped));
    if (hipHostGetDevicePointer(&d, p, 0) != hipSuccess) { (void)hipGetLastError(); (void)hipHostFree(p); return ULTRA_HIP_ERR_HIP; }
    std::memset(p, 0, bytes);
    ctx->host_blocks.push_back(p);
    *h_ptr = p; *d_ptr = d;
    return ULTRA_HIP_OK;
}

namespace ultra_hip { namespace dev {
__global__ void stream_post_kernel(unsigned* flag, unsigned value) {
    __threadfence_system();
    __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
} }

int ultra_hip_stream_post(ultra_hip_ctx* ctx, uint32_t* d_flag, uint32_t value) {
    if (!ctx || !d_flag) return ULTRA_HIP_ERR_INVALID_ARG;
    DeviceGuard guard(ctx->device);
    // A stream memory operation where the runtime has one for this memory (no launch: the command processor writes the word when
    // the stream gets there) — ULTRA_HIP_POST=kernel or a refusal falls back to one thread behind everything on the stream: the
    // kernels before it have completed (their stores to the block are out), the release store at system scope is what the host
    // polls for.
    static const bool use_kernel = [] { const char* e = std::getenv("ULTRA_HIP_POST"); return e && e[0] == 'k'; }();
    if (!use_kernel && !ctx->post_refused) {
        if (hipStreamWriteValue32(ctx->stream, d_flag, value, 0) == hipSuccess) return ULTRA_HIP_OK;
        (void)hipGetLastError();
        ctx->post_refused = true;
    }
    hipLaunchKernelGGL(dev::stream_post_kernel, dim3(1), dim3(1), 0, ctx->stream, d_flag, value);
    UH_HIP(hipGetLastError());
    return ULTRA_HIP_OK;
}

int ultra_hip_host_wait(ultra_hip_ctx* ctx, const volatile uint32_t* h_flag, uint32_t value, uint32_t timeout_us) {
    if (!ctx || !h_flag) return ULTRA_HIP_ERR_INVALID_ARG;
    g_host_syncs.fetch_add(1, std::memory_order_relaxed);       // a blocking wait of the host on the device, whatever it spins on
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0;; ++spins) {
        if (__atomic_load_n(h_flag, __ATOMIC_ACQUIRE) == value) return ULTRA_HIP_OK;
        if ((spins & 63u) == 63u &&
            std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() > (long long)timeout_us) break;
        __builtin_ia32_pause();
    }
    DeviceGuard guard(ctx->device);
    UH_HIP(hipStreamSynchronize(ctx->stream));
    return (__atomic_load_n(h_flag, __ATOMIC_ACQUIRE) == value) ? ULTRA_HIP_OK : ULTRA_HIP_ERR_HIP;
}

int ultra_hip_memcpy_d2h(ultra_hip_ctx* ctx, void* h_dst, const void* d_src, size_t bytes) {
    if (!ctx || (bytes && (!h_dst || !d_src))) return ULTRA_HIP_ERR_INVALID_ARG;
    DeviceGuard guard(ctx->device);
    UH_HIP(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    UH_HIP(uh_stream_sync(ctx->stream));
    return ULTRA_HIP_OK;
}

int ultra_hip_memset(ultra_hip_ctx* ctx, void* d_dst, int value, size_t bytes) {
    if (!ctx || (bytes && !d_dst)) return ULTRA_HIP_ERR_INVALID_ARG;
    DeviceGuard guard(ctx->device);
    UH_HIP(hipMemsetAsync(d_dst, value, bytes, ctx->stream));
    return ULTRA_HIP_OK;
}

// Device self-test of pinned_math.h: evaluates fn over n inputs on the GPU so
// the tests can compare with the host libm (not part of the reference surface).
//   fn: 0 sinf, 1 cosf, 2 atanf, 3 atan2f(a, b), 4 hypotf(a, b), 5 logf, 6 sqrtf
__global__ void pinned_math_kernel(int fn, const float* a, const float* b, float* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float r;
    switch (fn) {
        case 0: r = um::sinf_(a[i]); break;
        case 1: r = um::cosf_(a[i]); break;
        case 2: r = um::atanf_(a[i]); break;
        case 3: r = um::atan2f_(a[i], b[i]); break;
        case 5: r = um::logf_(a[i]); break;
        case 6: r = sqrtf(a[i]); break;                       // correctly rounded (-fhip-fp32-correctly-rounded-divide-sqrt)
        default: r = um::hypotf_(a[i], b[i]); break;
    }
    out[i] = r;
}

int ultra_hip_selftest_math(ultra_hip_ctx* ctx, int fn, const float* d_a, const float* d_b, float* d_out, size_t n) {
    if (!ctx || !d_a || !d_out || fn < 0 || fn > 6 || ((fn == 3 || fn == 4) && !d_b) || n > 0x7fffffffull)
        return ULTRA_HIP_ERR_INVALID_ARG;
    if (n == 0) return ULTRA_HIP_OK;
    DeviceGuard guard(ctx->device);
    hipLaunchKernelGGL(pinned_math_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, fn, d_a,
                       d_b ? d_b : d_a, d_out, (int)n);
    UH_HIP(hipGetLastError());
    return ULTRA_HIP_OK;
}

}  // extern "C"
