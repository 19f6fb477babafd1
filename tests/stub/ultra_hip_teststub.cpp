// ultra_hip_teststub.cpp — TEST-ONLY stand-in for the C-ABI of include/ultra_hip.h.  NOT a product path and NOT a CPU fallback:
// it demodulates nothing and decodes nothing.  It exists so that the HOST code above the C-ABI (include/ultra_hip_waveform.hpp:
// buffers, sample-index arithmetic, the process() / processPresynced() state machine, the slot pool, the GUI-side getters) can
// run under AddressSanitizer and ThreadSanitizer in a container that has no GPU (tests/test_adapter_sanitizers.py; VERDICT r5
// item 5b).  It lives under tests/, is compiled into a temporary directory by that test alone, is linked by name into the test
// driver only, and is never named libultra_hip: neither projectultra_amd/_lib.py nor the adapters can resolve it
// (tests/test_abi.py::test_product_never_touches_the_oracle covers "teststub" too).
//
// What makes it useful under a sanitizer:
//   * "device" memory is ordinary heap memory: every byte the adapter asks a kernel to read or write is really read or written
//     here, so a device window that is one sample short, a download past the end of a scratch buffer or a use of a freed
//     context is an ASan report instead of silent corruption on the GPU;
//   * a deterministic fake modem: a sample > 0.9 is a "preamble" (acquisition: found when 2 symbols follow; data start 2 symbols
//     behind it), soft bits are a function of the audio they were demodulated from — the driver can tell that the right
//     samples reached the right call;
//   * argument validation as strict as the library's (symbol ranges against the context's capacity, training symbols at
//     first_symbol 0): an adapter that would get ULTRA_HIP_ERR_INVALID_ARG from the product gets it here;
//   * fault injection: ultra_hip_teststub_fail_after(n) makes the n-th next entry call return ULTRA_HIP_ERR_HIP, to walk the
//     adapters' error paths (exceptions out of detail::check, the guarded IWaveform boundary) under the leak checker.
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>

#include "ultra_hip.h"

struct ultra_hip_ctx {
    uint32_t magic = 0x57ab57abu;
    ultra_hip_config cfg{};
    ultra_hip_geometry geo{};
    int device = 0;
    int start_mode = 0;
    uint32_t symbols_done = 0;             // of the stream in flight
    float cfo = 0.0f, phase = 0.0f, timing = 0.0f;
    bool have_records = false;
    std::mutex m;                          // one object, one thread at a time is the contract: a violation shows as a TSan report on `busy`
    int busy = 0;
    void* blocks[8] = {};                  // ultra_hip_host_block: here the host's and the device's view are the same heap block
    int n_blocks = 0;
    char* ring = nullptr; size_t ring_off = 0;   // ultra_hip_stage_input: slots of exactly the bytes asked for would be better for ASan; a ring is what the library has
};

namespace {
std::atomic<long> g_fail_after{-1};
std::atomic<unsigned long long> g_calls{0};

bool inject() {
    g_calls.fetch_add(1, std::memory_order_relaxed);
    long v = g_fail_after.load(std::memory_order_relaxed);
    while (v > 0 && !g_fail_after.compare_exchange_weak(v, v - 1, std::memory_order_relaxed)) {}
    return v == 1;
}
#define STUB_ENTRY(ctx)                                                        \
    if (!(ctx) || (ctx)->magic != 0x57ab57abu) return ULTRA_HIP_ERR_INVALID_ARG; \
    if (inject()) return ULTRA_HIP_ERR_HIP;                                    \
    Busy busy_guard(ctx)
struct Busy {                              // plain (non-atomic) writes on purpose: two threads inside one context are a data race TSan reports
    ultra_hip_ctx* c;
    explicit Busy(ultra_hip_ctx* x) : c(x) { c->busy = c->busy + 1; }
    ~Busy() { c->busy = c->busy - 1; }
};

int geometry(const ultra_hip_config* c, ultra_hip_geometry* g) {
    if (!c || !g) return ULTRA_HIP_ERR_INVALID_ARG;
    if ((c->fft_size != 512 && c->fft_size != 1024) || c->num_carriers == 0 || c->num_carriers > 64 || c->cp_mode > 2 || c->code_rate > 6 ||
        c->n_data_symbols == 0 || c->n_data_symbols > 252 || c->entry > 1)
        return ULTRA_HIP_ERR_UNSUPPORTED;
    static const uint32_t cp[3] = {32, 48, 64};
    static const uint32_t bits[11] = {1, 1, 2, 2, 3, 3, 4, 5, 6, 2, 8};
    static const uint32_t k[7] = {162, 216, 324, 432, 486, 540, 567};
    if (c->modulation > 10) return ULTRA_HIP_ERR_UNSUPPORTED;
    std::memset(g, 0, sizeof(*g));
    g->cp_len = cp[c->cp_mode] * (c->fft_size / 512);
    g->symbol_samples = c->fft_size + g->cp_len + c->symbol_guard;
    g->n_pilot_carriers = (c->use_pilots && c->pilot_spacing) ? c->num_carriers / c->pilot_spacing : 0;
    g->n_data_carriers = c->num_carriers - g->n_pilot_carriers;
    g->bits_per_carrier = bits[c->modulation];
    g->llrs_per_symbol = g->n_data_carriers * g->bits_per_carrier;
    g->llrs_per_frame = g->llrs_per_symbol * c->n_data_symbols;
    g->frame_samples = (c->n_data_symbols + (c->entry == ULTRA_ENTRY_PRESYNCED ? c->training_symbols : 0)) * g->symbol_samples;
    g->ldpc_n = 648; g->ldpc_k = k[c->code_rate]; g->ldpc_m = 648 - g->ldpc_k; g->ldpc_edges = 4 * g->ldpc_m;
    g->decoded_bytes = (g->ldpc_k + 7) / 8;
    return ULTRA_HIP_OK;
}

// first "preamble" (sample > 0.9) at or behind `from`, absolute index; n_samples if none
uint32_t find_marker(const float* audio, uint32_t origin, uint32_t from, uint32_t n_samples) {
    for (uint32_t i = from; i < n_samples; ++i)
        if (audio[i - origin] > 0.9f) return i;
    return n_samples;
}
}  // namespace

extern "C" {

void ultra_hip_teststub_fail_after(long n) { g_fail_after.store(n); }
unsigned long long ultra_hip_teststub_calls(void) { return g_calls.load(); }

int ultra_hip_abi_version(void) { return ULTRA_HIP_ABI_VERSION; }
const char* ultra_hip_strerror(int s) {
    switch (s) { case 0: return "ok"; case -1: return "invalid argument"; case -2: return "unsupported"; case -3: return "no device";
                 case -4: return "HIP call failed (injected)"; case -5: return "out of memory"; default: return "?"; }
}
int ultra_hip_geometry_for(const ultra_hip_config* cfg, ultra_hip_geometry* geo) { return geometry(cfg, geo); }

int ultra_hip_create(const ultra_hip_config* cfg, int device, void*, ultra_hip_ctx** out) {
    if (!cfg || !out) return ULTRA_HIP_ERR_INVALID_ARG;
    if (device != 0) return ULTRA_HIP_ERR_NO_DEVICE;
    if (inject()) return ULTRA_HIP_ERR_HIP;
    ultra_hip_geometry g;
    const int rc = geometry(cfg, &g);
    if (rc != ULTRA_HIP_OK) return rc;
    ultra_hip_ctx* c = new (std::nothrow) ultra_hip_ctx();
    if (!c) return ULTRA_HIP_ERR_OOM;
    c->cfg = *cfg; c->geo = g; c->device = device;
    *out = c;
    return ULTRA_HIP_OK;
}
void ultra_hip_destroy(ultra_hip_ctx* ctx) {
    if (!ctx) return;
    if (ctx->magic != 0x57ab57abu) std::abort();            // double destroy / foreign pointer
    ctx->magic = 0;
    for (int i = 0; i < ctx->n_blocks; ++i) std::free(ctx->blocks[i]);
    std::free(ctx->ring);
    delete ctx;
}
int ultra_hip_get_geometry(const ultra_hip_ctx* ctx, ultra_hip_geometry* geo) {
    if (!ctx || !geo || ctx->magic != 0x57ab57abu) return ULTRA_HIP_ERR_INVALID_ARG;
    *geo = ctx->geo;
    return ULTRA_HIP_OK;
}

int ultra_hip_malloc(ultra_hip_ctx* ctx, size_t bytes, void** d) {
    STUB_ENTRY(ctx);
    if (!d) return ULTRA_HIP_ERR_INVALID_ARG;
    *d = std::malloc(bytes ? bytes : 1);                    // exactly the bytes asked for: one past them is an ASan report
    return *d ? ULTRA_HIP_OK : ULTRA_HIP_ERR_OOM;
}
int ultra_hip_free(ultra_hip_ctx* ctx, void* d) {
    if (!ctx || ctx->magic != 0x57ab57abu) return ULTRA_HIP_ERR_INVALID_ARG;   // (never injected: a destructor calls it)
    std::free(d);
    return ULTRA_HIP_OK;
}
int ultra_hip_memcpy_h2d(ultra_hip_ctx* ctx, void* d, const void* h, size_t n) { STUB_ENTRY(ctx); if (n) std::memcpy(d, h, n); return ULTRA_HIP_OK; }
int ultra_hip_memcpy_h2d_async(ultra_hip_ctx* ctx, void* d, const void* h, size_t n) { STUB_ENTRY(ctx); if (n) std::memcpy(d, h, n); return ULTRA_HIP_OK; }
int ultra_hip_memcpy_d2h(ultra_hip_ctx* ctx, void* h, const void* d, size_t n) { STUB_ENTRY(ctx); if (n) std::memcpy(h, d, n); return ULTRA_HIP_OK; }

int ultra_hip_host_block(ultra_hip_ctx* ctx, size_t bytes, void** h, void** d) {
    STUB_ENTRY(ctx);
    if (!h || !d || bytes == 0 || ctx->n_blocks >= 8) return ULTRA_HIP_ERR_INVALID_ARG;
    void* p = std::calloc(1, bytes);
    if (!p) return ULTRA_HIP_ERR_OOM;
    ctx->blocks[ctx->n_blocks++] = p;
    *h = p; *d = p;
    return ULTRA_HIP_OK;
}
int ultra_hip_stage_input(ultra_hip_ctx* ctx, const void* h_src, size_t bytes, void** d_view) {
    STUB_ENTRY(ctx);
    constexpr size_t kRing = size_t(1) << 20;
    if (!h_src || !d_view || bytes == 0 || bytes > kRing / 4) return ULTRA_HIP_ERR_INVALID_ARG;
    if (!ctx->ring) { ctx->ring = static_cast<char*>(std::malloc(kRing)); if (!ctx->ring) return ULTRA_HIP_ERR_OOM; }
    const size_t need = (bytes + 255) & ~size_t(255);
    if (ctx->ring_off + need > kRing) ctx->ring_off = 0;
    std::memcpy(ctx->ring + ctx->ring_off, h_src, bytes);
    *d_view = ctx->ring + ctx->ring_off;
    ctx->ring_off += need;
    return ULTRA_HIP_OK;
}
int ultra_hip_stream_post(ultra_hip_ctx* ctx, uint32_t* d_flag, uint32_t value) {
    STUB_ENTRY(ctx);
    if (!d_flag) return ULTRA_HIP_ERR_INVALID_ARG;
    __atomic_store_n(d_flag, value, __ATOMIC_RELEASE);
    return ULTRA_HIP_OK;
}
int ultra_hip_host_wait(ultra_hip_ctx* ctx, const volatile uint32_t* h_flag, uint32_t value, uint32_t) {
    STUB_ENTRY(ctx);
    if (!h_flag) return ULTRA_HIP_ERR_INVALID_ARG;
    return __atomic_load_n(h_flag, __ATOMIC_ACQUIRE) == value ? ULTRA_HIP_OK : ULTRA_HIP_ERR_HIP;
}

int ultra_hip_set_deinterleave(ultra_hip_ctx* ctx, uint32_t bps) { if (!ctx || ctx->magic != 0x57ab57abu || bps >= 648) return ULTRA_HIP_ERR_INVALID_ARG; return ULTRA_HIP_OK; }
int ultra_hip_set_deinterleave_table(ultra_hip_ctx* ctx, const uint16_t*, uint32_t) { return (ctx && ctx->magic == 0x57ab57abu) ? ULTRA_HIP_OK : ULTRA_HIP_ERR_INVALID_ARG; }
int ultra_hip_profile_enable(ultra_hip_ctx* ctx, int) { return (ctx && ctx->magic == 0x57ab57abu) ? ULTRA_HIP_OK : ULTRA_HIP_ERR_INVALID_ARG; }
int ultra_hip_clear_status(ultra_hip_ctx* ctx) { return (ctx && ctx->magic == 0x57ab57abu) ? ULTRA_HIP_OK : ULTRA_HIP_ERR_INVALID_ARG; }

int ultra_hip_demod_stream_start(ultra_hip_ctx* ctx, int mode, const float* d_timing) {
    STUB_ENTRY(ctx);
    if (mode < 0 || mode > 2 || (mode == ULTRA_STREAM_START_TIMING && !d_timing)) return ULTRA_HIP_ERR_INVALID_ARG;
    if (mode == ULTRA_STREAM_START_SYNC && (!ctx->have_records || ctx->cfg.entry == ULTRA_ENTRY_PRESYNCED)) return ULTRA_HIP_ERR_INVALID_ARG;
    ctx->start_mode = mode;
    if (mode == ULTRA_STREAM_START_TIMING) ctx->timing = d_timing[0];
    return ULTRA_HIP_OK;
}
int ultra_hip_stream_adopt(ultra_hip_ctx* dst, ultra_hip_ctx* src, size_t n) {
    STUB_ENTRY(dst);
    if (!src || src->magic != 0x57ab57abu || n != 1 || !src->have_records) return ULTRA_HIP_ERR_INVALID_ARG;
    dst->have_records = true;
    return ULTRA_HIP_OK;
}
int ultra_hip_demod_stream_set_cfo_phase(ultra_hip_ctx* ctx, size_t frame, float cfo, float phase) {
    STUB_ENTRY(ctx);
    if (frame != 0 || ctx->symbols_done == 0) return ULTRA_HIP_ERR_INVALID_ARG;     // only between two calls of a stream in flight
    ctx->cfo = cfo; ctx->phase = phase;
    return ULTRA_HIP_OK;
}

int ultra_hip_demod_stream_batch_eq(ultra_hip_ctx* ctx, const float* d_audio, size_t frame_stride, const float* d_cfo_hz, const float* d_cfo_phase,
                                    size_t n_frames, uint32_t first_symbol, uint32_t n_symbols, float* d_llr, float* d_state, float* d_eq) {
    STUB_ENTRY(ctx);
    const uint32_t train = ctx->cfg.entry == ULTRA_ENTRY_PRESYNCED ? ctx->cfg.training_symbols : 0;
    if (!d_audio || !d_llr || n_frames != 1 || n_symbols == 0) return ULTRA_HIP_ERR_INVALID_ARG;
    if (first_symbol + n_symbols > train + ctx->cfg.n_data_symbols) return ULTRA_HIP_ERR_INVALID_ARG;         // the context's capacity
    if (first_symbol == 0 && n_symbols < train) return ULTRA_HIP_ERR_INVALID_ARG;                              // all training symbols at once
    if (first_symbol != 0 && first_symbol != ctx->symbols_done) return ULTRA_HIP_ERR_INVALID_ARG;              // the stream continues where it stopped
    const uint32_t sym = ctx->geo.symbol_samples, lps = ctx->geo.llrs_per_symbol;
    if (frame_stride < (size_t)n_symbols * sym) return ULTRA_HIP_ERR_INVALID_ARG;
    if (first_symbol == 0) {
        ctx->cfo = d_cfo_hz ? d_cfo_hz[0] : 0.0f; ctx->phase = d_cfo_phase ? d_cfo_phase[0] : 0.0f;
        if (ctx->start_mode != ULTRA_STREAM_START_TIMING) ctx->timing = 0.0f;
        ctx->start_mode = ULTRA_STREAM_START_FRESH;
    }
    uint32_t out = 0, eq_row = 0;
    for (uint32_t s = 0; s < n_symbols; ++s) {
        const float* a = d_audio + (size_t)s * sym;
        double acc = 0; for (uint32_t i = 0; i < sym; ++i) acc += a[i];                                        // every sample of the symbol is read
        if (first_symbol + s < train) continue;                                                               // training symbols give no soft bits
        for (uint32_t j = 0; j < lps; ++j) d_llr[(size_t)out * lps + j] = 4.0f * a[j % sym] + (float)acc * 1e-6f + ((j & 1) ? 0.5f : -0.5f);
        if (d_eq) for (uint32_t c = 0; c < ULTRA_HIP_MAX_CARRIERS; ++c) { d_eq[((size_t)eq_row * ULTRA_HIP_MAX_CARRIERS + c) * 2] = a[c]; d_eq[((size_t)eq_row * ULTRA_HIP_MAX_CARRIERS + c) * 2 + 1] = -a[c]; }
        ++out; ++eq_row;
    }
    ctx->symbols_done = first_symbol + n_symbols;
    ctx->have_records = true;
    if (d_state) {
        const float st[ULTRA_HIP_STATE_FLOATS] = {ctx->cfo, 0.1f, 10.0f + (float)ctx->symbols_done, ctx->timing, ctx->phase, 0.0f, (float)ctx->symbols_done, 0.0f};
        std::memcpy(d_state, st, sizeof(st));
    }
    return ULTRA_HIP_OK;
}
int ultra_hip_demod_stream_batch(ultra_hip_ctx* ctx, const float* a, size_t fs, const float* c, const float* p, size_t n, uint32_t f0, uint32_t ns, float* l, float* st) {
    return ultra_hip_demod_stream_batch_eq(ctx, a, fs, c, p, n, f0, ns, l, st, nullptr);
}

// acquisition: the first marker at or behind rx_buffer's start with two symbols behind it; otherwise the buffer is trimmed to its last
// four symbols (as the reference trims what it has searched)
int ultra_hip_acquire_stream_batch(ultra_hip_ctx* ctx, const float* d_audio, size_t stride, uint32_t origin, uint32_t n_samples, size_t n_streams,
                                   uint32_t* d_resume, uint32_t* d_found, uint32_t* d_data_start, float* d_cfo, uint32_t* d_sync_offset) {
    STUB_ENTRY(ctx);
    if (!d_audio || !d_resume || !d_found || !d_data_start || !d_cfo || !d_sync_offset || n_streams != 1) return ULTRA_HIP_ERR_INVALID_ARG;
    const uint32_t start = d_resume[0], sym = ctx->geo.symbol_samples;
    if (start < origin || start > n_samples || (size_t)(n_samples - origin) > stride) return ULTRA_HIP_ERR_INVALID_ARG;
    d_found[0] = 0; d_data_start[0] = 0; d_cfo[0] = 0.0f; d_sync_offset[0] = 0;
    const uint32_t m = find_marker(d_audio, origin, start, n_samples);
    if (m < n_samples && n_samples - m >= 2 * sym) {
        d_found[0] = 1; d_data_start[0] = m + 2 * sym; d_cfo[0] = 1.5f; d_sync_offset[0] = m - start;
    } else {
        const uint32_t keep = 4 * sym, limit = (m < n_samples) ? m : n_samples;                    // never trim a marker that is still waiting for its symbols
        const uint32_t to = (n_samples - start > keep) ? n_samples - keep : start;
        d_resume[0] = to < limit ? to : (limit > start ? limit : start);
    }
    d_resume[1] = n_samples;
    return ULTRA_HIP_OK;
}
int ultra_hip_resync_stream_batch(ultra_hip_ctx* ctx, const float* d_audio, size_t stride, uint32_t origin, uint32_t n_samples, size_t n_streams,
                                  const uint32_t* d_resume, uint32_t* d_found, uint32_t* d_data_start, float* d_cfo, uint32_t* d_sync_offset) {
    STUB_ENTRY(ctx);
    if (!d_audio || !d_resume || !d_found || !d_data_start || !d_cfo || !d_sync_offset || n_streams != 1) return ULTRA_HIP_ERR_INVALID_ARG;
    const uint32_t start = d_resume[0], sym = ctx->geo.symbol_samples;
    if (start < origin || start > n_samples || (size_t)(n_samples - origin) > stride) return ULTRA_HIP_ERR_INVALID_ARG;
    d_found[0] = 0; d_data_start[0] = 0; d_cfo[0] = 0.0f; d_sync_offset[0] = 0;
    const uint32_t end = (n_samples - start > 2 * sym) ? start + 2 * sym : n_samples;              // the first two symbols of rx_buffer
    const uint32_t m = find_marker(d_audio, origin, start, end);
    if (m < end && n_samples - m >= 6 * sym) { d_found[0] = 1; d_data_start[0] = m + 2 * sym; d_cfo[0] = -0.75f; d_sync_offset[0] = m - start; }
    return ULTRA_HIP_OK;
}
int ultra_hip_chirp_sync_batch(ultra_hip_ctx* ctx, const float* d_audio, size_t stride, uint32_t n_samples, size_t n_streams, float threshold,
                               uint32_t* d_detected, int32_t* d_start, float* d_cfo, float* d_corr, int32_t*, int32_t*) {
    STUB_ENTRY(ctx);
    if (!d_audio || !d_detected || !d_start || !d_cfo || !d_corr || n_streams != 1 || n_samples > stride) return ULTRA_HIP_ERR_INVALID_ARG;
    const uint32_t m = find_marker(d_audio, 0, 0, n_samples);
    d_detected[0] = (m < n_samples && threshold < 0.9f) ? 1u : 0u;
    d_start[0] = d_detected[0] ? (int32_t)(m + 16) : -1;
    d_cfo[0] = d_detected[0] ? 2.0f : 0.0f; d_corr[0] = d_detected[0] ? 0.95f : 0.05f;
    return ULTRA_HIP_OK;
}

int ultra_hip_ldpc_decode_batch(ultra_hip_ctx* ctx, const float* d_llr, size_t n_cw, uint8_t* d_bytes, int32_t* d_iters, uint8_t* d_ok, float* d_total) {
    STUB_ENTRY(ctx);
    if (n_cw == 0) return ULTRA_HIP_OK;
    if (!d_llr || !d_bytes || !d_iters || !d_ok) return ULTRA_HIP_ERR_INVALID_ARG;
    const uint32_t k = ctx->geo.ldpc_k, nb = ctx->geo.decoded_bytes;
    for (size_t c = 0; c < n_cw; ++c) {
        const float* l = d_llr + c * 648;
        std::memset(d_bytes + c * nb, 0, nb);
        bool weak = false;
        for (uint32_t j = 0; j < 648; ++j) weak = weak || (l[j] != l[j]);                           // every value of the row is read; a NaN "does not decode"
        for (uint32_t j = 0; j < k; ++j) if (l[j] < 0) d_bytes[c * nb + j / 8] |= (uint8_t)(0x80u >> (j % 8));
        d_iters[c] = weak ? (int32_t)ctx->cfg.max_iterations : 1;
        d_ok[c] = weak ? 0 : 1;
        if (d_total) std::memcpy(d_total + c * 648, l, 648 * sizeof(float));
    }
    return ULTRA_HIP_OK;
}
int ultra_hip_decode_frames_batch(ultra_hip_ctx* ctx, const float* d_soft, size_t stride, uint32_t n_soft, size_t n_frames, ultra_hip_frame_result* r,
                                  uint8_t* d_data, size_t data_stride) {
    STUB_ENTRY(ctx);
    if (!d_soft || !r || !d_data || n_frames != 1 || n_soft > stride) return ULTRA_HIP_ERR_INVALID_ARG;
    std::memset(r, 0, sizeof(*r));
    const uint32_t n_cw = n_soft / 648, bytes = ctx->geo.ldpc_k / 8;
    double acc = 0; for (uint32_t i = 0; i < n_soft; ++i) acc += d_soft[i];
    if (n_cw == 0) { r->status = ULTRA_HIP_FRAME_CW0_FAILED; return ULTRA_HIP_OK; }
    if ((size_t)n_cw * bytes > data_stride) return ULTRA_HIP_ERR_INVALID_ARG;
    r->success = 1; r->frame_type = 0x30; r->codewords_ok = (int32_t)n_cw; r->status = ULTRA_HIP_FRAME_COMPLETE;
    r->frame_len = (int32_t)(n_cw * bytes);
    for (uint32_t i = 0; i < n_cw * bytes; ++i) d_data[i] = (uint8_t)(i + (acc > 0 ? 1 : 0));
    return ULTRA_HIP_OK;
}

}  // extern "C"
