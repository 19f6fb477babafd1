/*
 * ultra_hip.h — C-ABI of the MI355X-native batched OFDM-demodulate + LDPC-decode
 * receive path (libultra_hip.so).
 *
 * This is the drop-in boundary for the hot path of secup/ProjectUltra.  The
 * reference has no FFI of its own: its "plugin surface" is the C++ abstract
 * class ultra::IWaveform (src/waveform/waveform_interface.hpp:47-157) plus the
 * two concrete classes every caller uses, ultra::OFDMDemodulator
 * (include/ultra/ofdm.hpp:58-127) and ultra::LDPCDecoder
 * (include/ultra/fec.hpp:48-77).  Each entry point below names the reference
 * interface it replaces.  Host C++ (include/ultra_hip_waveform.hpp) and Python
 * (projectultra_amd/) bind these symbols; INTEGRATION.md shows the binding a
 * reference maintainer would add.
 *
 * Conventions
 *   - plain pointers and sizes only; no C++/torch types cross the boundary
 *   - every function returns ULTRA_HIP_OK (0) or a negative ultra_hip_status
 *   - "d_" pointers are DEVICE pointers (HBM), "h_" pointers are HOST pointers
 *   - one context per device per host thread; a context is not thread-safe
 *     (same rule as the reference: one demodulator instance per stream,
 *     docs/INVARIANTS.md:239-258)
 *   - all launches go to the context's stream; *_batch calls are asynchronous
 *     with respect to the host unless documented otherwise
 *   - LLR sign convention: LLR > 0  <=>  bit 0 (docs/INVARIANTS.md:213-222)
 */
#ifndef ULTRA_HIP_H
#define ULTRA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ULTRA_HIP_ABI_VERSION 10

/* ultra::Modulation (include/ultra/types.hpp:27-39) — same numeric values. */
enum ultra_hip_modulation {
    ULTRA_MOD_DBPSK = 0, ULTRA_MOD_BPSK = 1, ULTRA_MOD_DQPSK = 2, ULTRA_MOD_QPSK = 3,
    ULTRA_MOD_D8PSK = 4, ULTRA_MOD_QAM8 = 5, ULTRA_MOD_QAM16 = 6, ULTRA_MOD_QAM32 = 7,
    ULTRA_MOD_QAM64 = 8, ULTRA_MOD_QAM256 = 10
};

/* ultra::CodeRate (include/ultra/types.hpp:91-100) — same numeric values. */
enum ultra_hip_code_rate {
    ULTRA_RATE_R1_4 = 0, ULTRA_RATE_R1_3 = 1, ULTRA_RATE_R1_2 = 2, ULTRA_RATE_R2_3 = 3,
    ULTRA_RATE_R3_4 = 4, ULTRA_RATE_R5_6 = 5, ULTRA_RATE_R7_8 = 6
};

/* ultra::CyclicPrefixMode (include/ultra/types.hpp:76-80). */
enum ultra_hip_cp_mode { ULTRA_CP_SHORT = 0, ULTRA_CP_MEDIUM = 1, ULTRA_CP_LONG = 2 };

/* Where in the reference's receive state machine a batch frame enters. */
enum ultra_hip_entry {
    /* State on entering SYNCED from the Schmidl-Cox search on a freshly
     * constructed demodulator (src/ofdm/demodulator.cpp:26-43,533-591): audio
     * starts at the first data symbol, H=(1,0), mixer phase 0. */
    ULTRA_ENTRY_SYNCED = 0,
    /* OFDMDemodulator::processPresynced (src/ofdm/demodulator.cpp:854-985):
     * audio starts at the first of `training_symbols` LTS symbols. */
    ULTRA_ENTRY_PRESYNCED = 1
};

enum ultra_hip_status {
    ULTRA_HIP_OK = 0,
    ULTRA_HIP_ERR_INVALID_ARG = -1,   /* bad pointer / size / enum                */
    ULTRA_HIP_ERR_UNSUPPORTED = -2,   /* configuration outside the built path      */
    ULTRA_HIP_ERR_NO_DEVICE = -3,     /* no HIP device / device index out of range */
    ULTRA_HIP_ERR_HIP = -4,           /* a HIP runtime call failed                 */
    ULTRA_HIP_ERR_OOM = -5            /* device or host allocation failed          */
};

/*
 * POD mirror of the ultra::ModemConfig fields the receive path reads
 * (include/ultra/types.hpp:139-234) + the decoder knobs of
 * ultra::LDPCDecoder (include/ultra/fec.hpp:48-77).
 */
typedef struct ultra_hip_config {
    uint32_t sample_rate;      /* ModemConfig::sample_rate (48000)                 */
    uint32_t center_freq;      /* ModemConfig::center_freq (1500)                  */
    uint32_t fft_size;         /* 512 or 1024                                      */
    uint32_t num_carriers;     /* 30 / 59                                          */
    uint32_t cp_mode;          /* ultra_hip_cp_mode                                */
    uint32_t symbol_guard;     /* guard samples after each symbol                  */
    uint32_t pilot_spacing;    /* every pilot_spacing-th carrier is a pilot (>= 2 with use_pilots) */
    uint32_t use_pilots;       /* 0/1                                              */
    uint32_t modulation;       /* ultra_hip_modulation                             */
    uint32_t code_rate;        /* ultra_hip_code_rate                              */
    uint32_t max_iterations;   /* LDPCDecoder::setMaxIterations (default 50)       */
    uint32_t n_data_symbols;   /* data symbols per frame fed to the demodulator    */
    uint32_t entry;            /* ultra_hip_entry                                  */
    uint32_t training_symbols; /* ULTRA_ENTRY_PRESYNCED only (reference default 2) */
    /* The decision-directed adaptive equaliser of the coherent modulations (ModemConfig, include/ultra/types.hpp:170-174;
     * Impl::equalize's use_adaptive branch, lmsUpdate / rlsUpdate: src/ofdm/channel_equalizer.cpp:569-581,705-722,
     * 773-805).  All zero (a zero-initialised struct) = off, the state of every preset the reference ships.  The
     * differential modulations never reach that branch (equalize returns at :769): the fields have no effect on them. */
    uint32_t adaptive_eq_enabled; /* 0/1: equalise the data carriers against lms_weights instead of channel_estimate  */
    uint32_t adaptive_eq_use_rls; /* 0 LMS (lms_mu), 1 RLS (rls_lambda)                                               */
    uint32_t decision_directed;   /* 0/1: update the weights from the hard decision of every equalised carrier        */
    float lms_mu;                 /* ModemConfig::lms_mu (0.05)                                                       */
    float rls_lambda;             /* ModemConfig::rls_lambda (0.99)                                                   */
    /* ABI 8.  ModemConfig::sync_threshold (include/ultra/types.hpp:188): the Schmidl-Cox metric a search offset must exceed
     * (Impl::sync_threshold, src/ofdm/demodulator.cpp:30,503,614).  0 (a zero-initialised struct) = the default 0.80. */
    float sync_threshold;
} ultra_hip_config;

/* Geometry derived from a config (ModemConfig::getCyclicPrefix /
 * getSymbolDuration, include/ultra/types.hpp:197-213; code parameters
 * src/fec/ldpc_decoder.cpp:22-36). */
typedef struct ultra_hip_geometry {
    uint32_t cp_len;            /* cyclic prefix samples                            */
    uint32_t symbol_samples;    /* fft + cp + guard                                 */
    uint32_t frame_samples;     /* (training + data symbols) * symbol_samples       */
    uint32_t n_data_carriers;
    uint32_t n_pilot_carriers;
    uint32_t bits_per_carrier;
    uint32_t llrs_per_symbol;   /* n_data_carriers * bits_per_carrier               */
    uint32_t llrs_per_frame;    /* llrs_per_symbol * n_data_symbols                 */
    uint32_t ldpc_n;            /* 648                                              */
    uint32_t ldpc_k;            /* info bits                                        */
    uint32_t ldpc_m;            /* parity bits / check rows                         */
    uint32_t ldpc_edges;        /* edges of H = [H_data | I]                        */
    uint32_t decoded_bytes;     /* ceil(k / 8): bytes LDPCDecoder::decodeSoft returns */
} ultra_hip_geometry;

/* Device-side Monte-Carlo counters.  Reduced on device by
 * ultra_hip_count_errors and summed across ranks by the caller's collective. */
typedef struct ultra_hip_counters {
    uint64_t frames;
    uint64_t frame_errors;      /* !success or payload mismatch                    */
    uint64_t bit_errors;        /* payload bit errors                              */
    uint64_t info_bits;         /* payload bits compared                           */
    uint64_t ldpc_fail;         /* frames whose parity check never passed          */
    uint64_t iters_sum;         /* sum of lastIterations()                         */
    uint64_t undetected_errors; /* parity passed but payload differs               */
    uint64_t reserved;
} ultra_hip_counters;

typedef struct ultra_hip_ctx ultra_hip_ctx;

/* Number of blocking host-on-device waits the library has issued so far in this process (its copies to and from host
 * memory, ultra_hip_synchronize, a workspace growing): what a latency-bound caller — one live stream, one process() call at
 * a time — pays per call beside the kernels.  Diagnostic (the live-latency harness reports it per call: INTEGRATION.md 0).  ABI 8. */
unsigned long long ultra_hip_host_sync_count(void);

/* ABI version of the loaded library (== ULTRA_HIP_ABI_VERSION). */
int ultra_hip_abi_version(void);

/* Human-readable text for a status code. */
const char* ultra_hip_strerror(int status);

/* Number of visible HIP devices, or a negative status.  Does not create a
 * HIP context on any device. */
int ultra_hip_device_count(void);

/* Pure host arithmetic: fill *geo for *cfg.  No device needed.
 * Replaces: ModemConfig::getCyclicPrefix/getSymbolDuration
 * (include/ultra/types.hpp:197-213), OFDMDemodulator::Impl::setupCarriers
 * (src/ofdm/demodulator.cpp:46-69) carrier counts, getCodeParams
 * (src/fec/ldpc_decoder.cpp:22-36). */
int ultra_hip_geometry_for(const ultra_hip_config* cfg, ultra_hip_geometry* geo);

/* Create a context on `device`: builds the per-configuration constant tables
 * on the host exactly as the reference constructors do (NCO table
 * src/dsp/filters.cpp:228-238, FFT twiddles src/dsp/fft.cpp:75-82, carrier map
 * and pilot signs src/ofdm/demodulator.cpp:46-85, interpolation table
 * :137-193, Zadoff-Chu sequence :70-78, Tanner graph
 * src/fec/ldpc_decoder.cpp:64-137) and uploads them.
 * `stream` is a hipStream_t (may be NULL for the default stream).
 * Replaces: OFDMDemodulator::OFDMDemodulator(const ModemConfig&)
 * (src/ofdm/demodulator.cpp:455-458) + LDPCDecoder::LDPCDecoder(CodeRate)
 * (src/fec/ldpc_decoder.cpp:262-265) + WaveformFactory::create
 * (src/waveform/waveform_factory.cpp:11-61). */
int ultra_hip_create(const ultra_hip_config* cfg, int device, void* stream, ultra_hip_ctx** out);

/* Replaces the destructors of the two classes above. */
void ultra_hip_destroy(ultra_hip_ctx* ctx);

/* Size the context's per-frame workspaces (tracker records, FFT bins, phase tables, the LLR workspace of the fused
 * entry) for batches of up to n_frames, so that no later call has to synchronise the stream and allocate.  The
 * workspaces otherwise grow on demand, inside the first call that needs more.  (≈ 5.5 KB + 4 * llrs_per_frame bytes
 * per frame.) */
int ultra_hip_reserve(ultra_hip_ctx* ctx, size_t n_frames);

/* Geometry of a live context. */
int ultra_hip_get_geometry(const ultra_hip_ctx* ctx, ultra_hip_geometry* geo);

/* Copy the Tanner graph the context decodes with to the host (for parity
 * tests): row_ptr[m+1], col_idx[edges] — row-major edge order of
 * LDPCDecoder::Impl::H_rows (src/fec/ldpc_decoder.cpp:64-137). */
int ultra_hip_get_tanner_graph(const ultra_hip_ctx* ctx, uint32_t* h_row_ptr, uint32_t* h_col_idx);

/*
 * Batched LDPC decode: n_cw codewords of 648 LLRs each.
 *   d_llr      [n_cw][648] f32 (device)
 *   d_bytes    [n_cw][decoded_bytes] u8 (device): info bits packed MSB-first
 *   d_iters    [n_cw] i32: LDPCDecoder::lastIterations() per codeword
 *   d_ok       [n_cw] u8 : LDPCDecoder::lastDecodeSuccess() per codeword
 *   d_llr_total[n_cw][648] f32 or NULL: final a-posteriori LLRs (parity tests)
 * Replaces: LDPCDecoder::decodeSoft (src/fec/ldpc_decoder.cpp:283-428) →
 * Impl::decodeBP (:153-259), lastDecodeSuccess (:430), lastIterations (:434).
 */
int ultra_hip_ldpc_decode_batch(ultra_hip_ctx* ctx, const float* d_llr, size_t n_cw,
                                uint8_t* d_bytes, int32_t* d_iters, uint8_t* d_ok,
                                float* d_llr_total);

/*
 * Batched OFDM demodulation: n_frames frames, each frame_samples f32 audio
 * samples starting at the configured entry point.
 *   d_audio    [n_frames] rows of `frame_stride` floats (>= frame_samples)
 *   d_cfo_hz   [n_frames] f32 or NULL (=0): initial freq_offset_hz
 *              (OFDMDemodulator::setFrequencyOffset / coarse CFO from sync).  ULTRA_ENTRY_PRESYNCED: a NaN entry
 *              means "the frequency offset was never set on this demodulator" — processPresynced then takes it
 *              from the two training symbols (Impl::estimateCFOFromTraining, src/ofdm/ofdm_sync.cpp:278-380,
 *              demodulator.cpp:920-925) and starts the correction phase at 0; NULL or a number is a preset,
 *              trusted offset (:918-919).  ULTRA_ENTRY_SYNCED has no such convention: a NaN there is the caller's
 *              error and travels through the tracker as NaN (the reference would do the same with a NaN offset)
 *   d_cfo_phase[n_frames] f32 or NULL (=0): initial freq_correction_phase
 *              (OFDMDemodulator::setFrequencyOffsetWithPhase)
 *   d_llr      [n_frames][llrs_per_frame] f32: soft bits in the order
 *              OFDMDemodulator::getSoftBits hands them out
 *   d_state    [n_frames][ULTRA_HIP_STATE_FLOATS] f32 or NULL: final tracker
 *              state (see ULTRA_HIP_STATE_* indices)
 * Replaces: OFDMDemodulator::process SYNCED loop (src/ofdm/demodulator.cpp:
 * 672-697) / processPresynced (:854-985) + getSoftBits (:766-791), i.e.
 * toBaseband, extractSymbol, updateChannelEstimate, interpolateChannel,
 * equalize (src/ofdm/channel_equalizer.cpp:19-71,330-631,728-840) and
 * demodulateSymbol (src/ofdm/demodulator.cpp:199-435).
 */
int ultra_hip_demod_batch(ultra_hip_ctx* ctx, const float* d_audio, size_t frame_stride,
                          const float* d_cfo_hz, const float* d_cfo_phase, size_t n_frames,
                          float* d_llr, float* d_state);

/* ultra_hip_demod_batch with a caller-chosen row stride of the LLR array (>= llrs_per_frame): several configurations
 * can then share one LLR array (the mode grid of BASELINE configs[4]: one demodulation per MODULATION over the frames of
 * all its code rates, rows of 768 soft bits whatever the modulation). */
int ultra_hip_demod_batch_strided(ultra_hip_ctx* ctx, const float* d_audio, size_t frame_stride, const float* d_cfo_hz,
                                  const float* d_cfo_phase, size_t n_frames, float* d_llr, size_t llr_stride, float* d_state);

/* ultra_hip_ldpc_decode_batch over n_blocks runs of block_len codewords inside a larger LLR array of rows of llr_stride
 * floats (the first 648 of a row are the codeword): codeword c = row (c / block_len) * block_stride + c % block_len.
 * One launch per CODE RATE over the soft bits of every modulation of a mode grid (decoding does not depend on the
 * modulation).  Results are dense: d_bytes [n_blocks * block_len][ceil(k/8)], d_iters, d_ok.
 * The context's channel deinterleaver (ultra_hip_set_deinterleave[_table]) applies to EVERY run alike: a step that depends on
 * the modulation's bits per symbol must not be set on a context that decodes several modulations' runs (the mode grid sets
 * none). */
int ultra_hip_ldpc_decode_blocks(ultra_hip_ctx* ctx, const float* d_llr, size_t llr_stride, size_t block_len, size_t block_stride,
                                 size_t n_blocks, uint8_t* d_bytes, int32_t* d_iters, uint8_t* d_ok);

/* The SYNCED symbol loop of OFDMDemodulator::process (src/ofdm/demodulator.cpp:672-697) as it runs on a live stream:
 * symbols are demodulated when they arrive.  Symbols [first_symbol, first_symbol + n_symbols) of every frame, the
 * tracker continuing from where the previous call on this context left it (first_symbol == 0 starts a fresh
 * demodulator from d_cfo_hz / d_cfo_phase, which are ignored otherwise).  Same results as one
 * ultra_hip_demod_batch call over all symbols: the chain is causal.
 *   d_audio   [n_frames] rows of frame_stride floats, row f starting at symbol first_symbol of frame f
 *   d_llr     [n_frames][n_data * llrs_per_symbol] f32, n_data = the data symbols among those of this call
 *   d_state   [n_frames][ULTRA_HIP_STATE_FLOATS] or NULL: the tracker after the last symbol of this call
 * ULTRA_ENTRY_PRESYNCED: the first call (first_symbol 0) is processPresynced — it takes all training symbols (n_symbols >=
 * training_symbols) and whatever data symbols the caller hands processPresynced; later calls start behind them and are the
 * rest of the frame arriving through process(), as it does in the reference (the object is SYNCED after processPresynced):
 * process()'s loop runs updateChannelEstimate for every layout (:676), processPresynced's own loop only with pilots
 * (:954-960) — without pilots the two differ (pilot_phase_correction is reset per symbol, snr_symbol_count counts), and so do
 * a frame handed over whole and a frame handed over in two pieces, exactly as in the reference.
 * The context must have been created with n_data_symbols >= the frame's length (<= 251: process() gives up after
 * MAX_SYMBOLS_BEFORE_TIMEOUT + 1 symbols) and must not run another batch in between. */
int ultra_hip_demod_stream_batch(ultra_hip_ctx* ctx, const float* d_audio, size_t frame_stride, const float* d_cfo_hz,
                                 const float* d_cfo_phase, size_t n_frames, uint32_t first_symbol, uint32_t n_symbols,
                                 float* d_llr, float* d_state);
/* The same call, which also delivers what OFDMDemodulator::Impl::demodulateSymbol appends to constellation_symbols
 * (src/ofdm/demodulator.cpp:199-208; read by getConstellationSymbols(), :827-830 — the GUI's scatter plot): the equalized
 * data carriers of every data symbol of this call (ABI 9).
 *   d_equalized  [n_frames][data symbols of this call][ULTRA_HIP_MAX_CARRIERS] (re, im) f32 pairs, the first
 *                ultra_hip_geometry::n_data_carriers of each row written; NULL = ultra_hip_demod_stream_batch.
 * Such a call runs the carrier half per symbol (the chain ULTRA_HIP_FALLBACK_CHAIN=1 selects): same soft bits, same state. */
#define ULTRA_HIP_MAX_CARRIERS 64
int ultra_hip_demod_stream_batch_eq(ultra_hip_ctx* ctx, const float* d_audio, size_t frame_stride, const float* d_cfo_hz,
                                    const float* d_cfo_phase, size_t n_frames, uint32_t first_symbol, uint32_t n_symbols,
                                    float* d_llr, float* d_state, float* d_equalized);
/* How the NEXT first_symbol == 0 call of ultra_hip_demod_stream_batch on this context starts its frames (consumed by that
 * call; ultra_hip_demod_batch and friends always start fresh).  One ultra::OFDMDemodulator object lives through many
 * frames, and not every way into a new frame resets the tracker:
 *   ULTRA_STREAM_START_FRESH   the constructor's state / after OFDMDemodulator::reset() — the default.
 *   ULTRA_STREAM_START_SYNC    SEARCHING -> SYNCED on a demodulator that demodulated frames before and was not reset in
 *                              between (src/ofdm/demodulator.cpp:533-591; the legacy Modem never calls reset():
 *                              src/modem/modem.cpp:153-166).  The transition sets freq_offset_hz = freq_offset_filtered =
 *                              the coarse CFO (d_cfo_hz of the stream call), the correction phase, symbols_since_sync and
 *                              timing_offset_samples to 0, restarts the mixer, clears dbpsk_prev_equalized and (unless
 *                              the layout is differential without pilots) the carrier phase correction.  EVERYTHING
 *                              ELSE is carried from the records the previous frame left in this context: channel_estimate,
 *                              noise_variance, estimated_snr_linear, snr_symbol_count, prev_pilot_phases,
 *                              pilot_phase_correction, the adaptive equaliser's weights.  ULTRA_ENTRY_SYNCED only; the
 *                              context must hold records of at least n_frames frames.
 *   ULTRA_STREAM_START_TIMING  a fresh tracker whose Impl::timing_offset_samples starts at d_timing[frame] — the one value
 *                              that neither reset() (:987-1017), nor the reset block of processPresynced (:868-905), nor the
 *                              mid-frame preamble (:626-655) clears.  d_timing [n_frames] f32 must stay valid until the
 *                              stream call has been issued.
 * d_timing is ignored (may be NULL) for the other two modes. */
enum ultra_hip_stream_start { ULTRA_STREAM_START_FRESH = 0, ULTRA_STREAM_START_SYNC = 1, ULTRA_STREAM_START_TIMING = 2 };
int ultra_hip_demod_stream_start(ultra_hip_ctx* ctx, int mode, const float* d_timing);
/* One ultra::OFDMDemodulator object is ONE tracker (Impl) whichever way its frames come in: a Schmidl-Cox frame found by process()
 * right after a processPresynced() frame — no reset() in between — carries that frame's channel estimate, noise variance, SNR,
 * pilot history and equaliser weights through the SEARCHING -> SYNCED transition (demodulator.cpp:533-591).  The two entries are
 * two contexts here; this call copies the tracker records of frames 0 .. n_frames - 1 from `src` (the context the previous frame
 * was demodulated in) to `dst` (same device, same carrier layout and modulation: otherwise ULTRA_HIP_ERR_INVALID_ARG), on dst's
 * stream, behind what src's stream has in flight.  Follow it with ultra_hip_demod_stream_start(dst, ULTRA_STREAM_START_SYNC, ..)
 * (ABI 9). */
int ultra_hip_stream_adopt(ultra_hip_ctx* dst, ultra_hip_ctx* src, size_t n_frames);

/* OFDMDemodulator::setFrequencyOffset (demodulator.cpp:805-815) between two ultra_hip_demod_stream_batch calls:
 * freq_offset_hz = freq_offset_filtered = cfo_hz, correction phase 0, for frame `frame` of the stream in flight, applied
 * from the next symbol on — also when the frame started without offsets (d_cfo_hz == NULL at first_symbol 0): the batch
 * then leaves the zero-offset fast paths for the rest of the frame (tests/golden/setcfo.npz). */
int ultra_hip_demod_stream_set_cfo(ultra_hip_ctx* ctx, size_t frame, float cfo_hz);
/* ... and OFDMDemodulator::setFrequencyOffsetWithPhase (demodulator.cpp:816-825): the same with freq_correction_phase =
 * cfo_phase instead of 0 (ABI 8). */
int ultra_hip_demod_stream_set_cfo_phase(ultra_hip_ctx* ctx, size_t frame, float cfo_hz, float cfo_phase);

#define ULTRA_HIP_STATE_FLOATS 8
#define ULTRA_HIP_STATE_FREQ_OFFSET_HZ 0   /* OFDMDemodulator::getFrequencyOffset     */
#define ULTRA_HIP_STATE_NOISE_VARIANCE 1   /* Impl::noise_variance                    */
#define ULTRA_HIP_STATE_SNR_LINEAR     2   /* Impl::estimated_snr_linear              */
#define ULTRA_HIP_STATE_TIMING_OFFSET  3   /* Impl::timing_offset_samples             */
#define ULTRA_HIP_STATE_CFO_PHASE      4   /* Impl::freq_correction_phase             */
#define ULTRA_HIP_STATE_MIXER_PHASE    5   /* NCO::phase_                             */
#define ULTRA_HIP_STATE_SYMBOLS        6   /* Impl::snr_symbol_count (as float)       */
#define ULTRA_HIP_STATE_RESERVED       7

/*
 * The whole receive path in one call: demodulate (one set of launches per OFDM symbol), then LDPC-decode the
 * first 648 LLRs of each frame.  Same arguments as the two calls above; d_llr may be NULL — the LLRs then
 * travel through a workspace of the context in HBM (llrs_per_frame floats per frame) instead of a caller buffer.
 * Replaces the per-frame body of the reference's Monte-Carlo harnesses
 * (tools/test_nvis_mode.cpp:88-113): demod.process → getSoftBits → first 648
 * → decoder.decodeSoft.
 */
int ultra_hip_demod_decode_batch(ultra_hip_ctx* ctx, const float* d_audio, size_t frame_stride,
                                 const float* d_cfo_hz, const float* d_cfo_phase, size_t n_frames,
                                 float* d_llr, uint8_t* d_bytes, int32_t* d_iters, uint8_t* d_ok);

/*
 * Compare decoded bytes with the transmitted payloads and accumulate the
 * Monte-Carlo counters on device (one atomic block-reduction per launch).
 *   d_payload [n_frames][payload_bytes] u8 reference payloads
 *   d_counters  device ultra_hip_counters, accumulated into (caller zeroes)
 * Frame OK iff ok && first payload_bytes bytes equal
 * (tools/test_nvis_mode.cpp:104-113).
 */
int ultra_hip_count_errors(ultra_hip_ctx* ctx, const uint8_t* d_bytes, const int32_t* d_iters,
                           const uint8_t* d_ok, const uint8_t* d_payload, size_t payload_bytes,
                           size_t n_frames, ultra_hip_counters* d_counters);

/* The same for a batch that holds several sweep points back to back (the points of one BER/FER curve share their
 * demodulate + decode launches — the receive path does not depend on the SNR, tools/test_mode_snr.cpp:126-160 is a
 * loop over it): frames [p * frames_per_point, (p + 1) * frames_per_point) accumulate into d_counters[p]. */
int ultra_hip_count_errors_points(ultra_hip_ctx* ctx, const uint8_t* d_bytes, const int32_t* d_iters,
                                  const uint8_t* d_ok, const uint8_t* d_payload, size_t payload_bytes,
                                  size_t n_points, size_t frames_per_point, ultra_hip_counters* d_counters);

/* The one collective of the path: sum the eight Monte-Carlo counters over the ranks of an
 * RCCL communicator, in place, on the context's stream.  rccl_comm is the host's ncclComm_t (one process
 * per GPU, created with ncclCommInitRank; torch.distributed users call dist.all_reduce on the tensor
 * instead).  RCCL is bound with dlopen at first use; ULTRA_HIP_ERR_UNSUPPORTED if it cannot be loaded. */
int ultra_hip_counters_allreduce(ultra_hip_ctx* ctx, void* rccl_comm, ultra_hip_counters* d_counters);

/* Block the host until everything queued on the context's stream is done. */
int ultra_hip_synchronize(ultra_hip_ctx* ctx);

/* Time the next *_batch launches with HIP events on the context's stream:
 * begin() records a start event, end() records a stop event, synchronizes it
 * and returns the elapsed milliseconds in *ms. */
int ultra_hip_timer_begin(ultra_hip_ctx* ctx);
int ultra_hip_timer_end(ultra_hip_ctx* ctx, float* ms);

/* Preamble acquisition: the SEARCHING state of OFDMDemodulator::process
 * (src/ofdm/demodulator.cpp:461-600: Schmidl-Cox search with energy gate and plateau test, coarse CFO,
 * LTS matched-filter refinement; src/ofdm/ofdm_sync.cpp:20-261,386-461) for a batch of independent
 * streams.  Stream s = d_audio[s * stream_stride .. + n_samples) is received by a fresh demodulator
 * that is fed `chunk` samples per process() call (the harnesses use 960: the search result depends on
 * the chunking, docs and SURVEY quirk 7), until sync is declared or the stream ends.
 *   d_found[s]       1 when sync was declared
 *   d_data_start[s]  absolute sample index of the first data symbol (what process() erases the buffer
 *                    up to: demodulator.cpp:572-575)
 *   d_cfo_hz[s]      Impl::estimateCoarseCFO at the Schmidl-Cox offset
 *   d_sync_offset[s] (nullable) OFDMDemodulator::getLastSyncOffset()
 *   d_fed_at_sync[s] (nullable) samples fed when sync was declared
 * (d_data_start, d_cfo_hz) is the SYNCED entry of ultra_hip_demod_batch.  Uses
 * ModemConfig::sync_threshold's default 0.80. */
int ultra_hip_acquire_batch(ultra_hip_ctx* ctx, const float* d_audio, size_t stream_stride, uint32_t n_samples,
                            uint32_t chunk, size_t n_streams, uint32_t* d_found, uint32_t* d_data_start,
                            float* d_cfo_hz, uint32_t* d_sync_offset, uint32_t* d_fed_at_sync);

/* The same search for LIVE streams, one OFDMDemodulator::process() call per launch (the streaming adapters,
 * include/ultra_hip_waveform.hpp: HipOfdmCoxWaveform).  Everything fed since the previous launch is this call's chunk.
 *   origin        sample index i of stream s lives at d_audio[s * stream_stride + i - origin]: the caller keeps only
 *                 what the search can still look at (at least everything from d_resume[4 s] on)
 *   n_samples     samples fed so far, this call's included (absolute, as are all sample indices below)
 *   d_resume      [n_streams][4] u32, in/out: {start of rx_buffer, samples fed, noise floor of the energy gate (float
 *                 bits), epoch} — what Impl carries between process() calls while SEARCHING; zero it for a fresh
 *                 demodulator, keep the noise floor across frames (OFDMDemodulator::reset does not clear it).  The library
 *                 keeps each stream's Schmidl-Cox metrics in HBM between the calls, keyed by absolute sample index, so that a
 *                 call evaluates only the windows its new samples completed (the reference re-evaluates its whole buffer on
 *                 every call: demodulator.cpp:497): a stream whose `samples fed` is 0 or smaller than at the previous call
 *                 starts with an empty cache, and an owner that restarts its sample indices any other way (a new stream laid
 *                 over the old indices) says so by CHANGING the epoch word
 *   outputs as in ultra_hip_acquire_batch; on d_found[s] = 1 the stream enters SYNCED at d_data_start[s]. */
int ultra_hip_acquire_stream_batch(ultra_hip_ctx* ctx, const float* d_audio, size_t stream_stride, uint32_t origin,
                                   uint32_t n_samples, size_t n_streams, uint32_t* d_resume, uint32_t* d_found,
                                   uint32_t* d_data_start, float* d_cfo_hz, uint32_t* d_sync_offset);

/* The preamble check of the SYNCED state (src/ofdm/demodulator.cpp:605-657: a new frame arriving while the demodulator
 * still waits for the rest of the old one).  process() runs it when symbols have been demodulated since the sync and the
 * last two calls or more brought no new soft bit; the caller keeps those two counters (HipOfdmCoxWaveform does) and calls
 * this entry instead of re-implementing the scan: Schmidl-Cox metric at offsets 0, 8, .. <= min(size - 6 preamble symbols,
 * 2 data symbols) of rx_buffer = samples [d_resume[4 s], n_samples) of stream s — no energy gate, no plateau test; the first
 * offset above sync_threshold whose LTS confirmation (refineLTSTiming) holds wins, a failed confirmation continues the
 * scan.  d_resume is only read (word 0 of each record: where rx_buffer starts); origin, n_samples and the outputs are those
 * of ultra_hip_acquire_stream_batch.  On d_found[s] = 1 the caller drops its soft bits and restarts the stream's
 * demodulation at d_data_start[s] with d_cfo_hz[s] (ultra_hip_demod_stream_batch, first_symbol 0 — the reference resets
 * the tracker to the constructor's state there, :640-655); d_found[s] = 0 (also when fewer than 6 preamble symbols are
 * buffered) leaves everything as it was. */
int ultra_hip_resync_stream_batch(ultra_hip_ctx* ctx, const float* d_audio, size_t stream_stride, uint32_t origin,
                                  uint32_t n_samples, size_t n_streams, const uint32_t* d_resume, uint32_t* d_found,
                                  uint32_t* d_data_start, float* d_cfo_hz, uint32_t* d_sync_offset);

/* Chirp synchronisation: OFDMChirpWaveform::detectSync
 * (src/waveform/ofdm_chirp_waveform.cpp:129-172) = sync::ChirpSync::detectDualChirp
 * (src/sync/chirp_sync.hpp:349-505; 300 -> 2700 Hz up chirp, 100 ms gap, down chirp, 500 ms each) for a
 * batch of independent buffers.
 *   d_detected[s]     SyncResult::detected
 *   d_start_sample[s] SyncResult::start_sample: where the two training symbols start (-1 if not detected) —
 *                     with d_cfo_hz the PRESYNCED entry of ultra_hip_demod_batch; the initial CFO phase is
 *                     -2 pi cfo start_sample / fs wrapped to [-pi, pi] (OFDMChirpWaveform::process, :174-190)
 *   d_cfo_hz[s]       SyncResult::cfo_hz (from the distance of the up and down chirp peaks)
 *   d_correlation[s]  SyncResult::correlation = max(up, down)
 *   d_up_chirp_start / d_down_chirp_start (nullable) the CFO-corrected chirp positions
 * threshold: IWaveform::detectSync's argument (detectDualChirp's default is 0.15). */
int ultra_hip_chirp_sync_batch(ultra_hip_ctx* ctx, const float* d_audio, size_t stream_stride, uint32_t n_samples,
                               size_t n_streams, float threshold, uint32_t* d_detected, int32_t* d_start_sample,
                               float* d_cfo_hz, float* d_correlation, int32_t* d_up_chirp_start,
                               int32_t* d_down_chirp_start);

/* End to end from raw audio: ultra_hip_acquire_batch, then the SYNCED demodulation of each stream from
 * its own data start with its own coarse CFO, then the LDPC decode of the first 648 soft bits — what
 * one OFDMDemodulator::process loop + getSoftBits + LDPCDecoder::decodeSoft does per trial in the
 * harnesses (tools/test_nvis_mode.cpp:88-113), for n_streams independent trials.
 *   d_entry[s]   (nullable) first data sample, 0xffffffff when the stream yields no frame: no sync, or
 *                fewer than frame_samples after the data start (the reference would still be waiting)
 *   d_cfo_hz[s]  (nullable) coarse CFO
 *   d_llr        (nullable) [n_streams][llrs_per_frame]; rows of streams without a frame are unspecified
 *   d_bytes / d_iters / d_ok  as ultra_hip_demod_decode_batch; all zero for streams without a frame
 * Requires n_samples >= frame_samples and the SYNCED entry. */
int ultra_hip_receive_batch(ultra_hip_ctx* ctx, const float* d_audio, size_t stream_stride, uint32_t n_samples,
                            uint32_t chunk, size_t n_streams, float* d_llr, uint8_t* d_bytes, int32_t* d_iters,
                            uint8_t* d_ok, uint32_t* d_entry, float* d_cfo_hz);

/* End to end for the chirp-synchronised waveform: ultra_hip_chirp_sync_batch, then what the caller of
 * IWaveform does with its result (tools/test_nvis_mode.cpp; OFDMChirpWaveform::process,
 * src/waveform/ofdm_chirp_waveform.cpp:174-215): setFrequencyOffset(cfo), process(samples from start_sample) =
 * setFrequencyOffsetWithPhase(cfo, -2 pi cfo start / fs wrapped) + processPresynced(.., 2), then the LDPC decode
 * of the first 648 soft bits.  The context must use ULTRA_ENTRY_PRESYNCED (2 training symbols).
 *   d_entry[s]  (nullable) training start sample, 0xffffffff when no chirp pair was detected or the frame does
 *               not fit in the buffer: such streams report ok = 0, iterations = 0, zero bytes
 *   d_cfo_hz[s] (nullable) the CFO handed to the demodulator (0 for unusable streams)
 *   d_llr       (nullable) [n_streams][llrs_per_frame] soft bits */
int ultra_hip_chirp_receive_batch(ultra_hip_ctx* ctx, const float* d_audio, size_t stream_stride, uint32_t n_samples,
                                  size_t n_streams, float threshold, float* d_llr, uint8_t* d_bytes, int32_t* d_iters,
                                  uint8_t* d_ok, uint32_t* d_entry, float* d_cfo_hz);

/* v2 wire format (second half): what RxPipeline::processFrame does with the soft bits of
 * a frame (src/gui/modem/rx_pipeline.cpp:283-346) for a batch of frames — detectPing (:446-472),
 * deinterleaveCodewords (:474-491; the context's ultra_hip_set_deinterleave setting), decodeFrame (:348-444):
 * CW0 -> v2::parseHeader (src/protocol/frame_v2.cpp:1175-1229) -> the remaining codewords ->
 * CodewordStatus::reassemble (:952-982,1023-1044).  The context's code rate is the rate of every codeword
 * (RxPipeline::setDataMode: R1/4 before the connection, the negotiated rate after it).
 *   d_soft        [n_frames][frame_stride] soft bits as IWaveform::getSoftBits returns them, n_soft valid per
 *                 frame (floor(n_soft / 648) codewords are available)
 *   d_results     [n_frames] RxFrameResult fields + status
 *   d_frame_data  [n_frames][frame_data_stride] RxFrameResult::frame_data of complete frames;
 *                 frame_data_stride >= floor(n_soft / 648) * floor(k / 8)
 * All available codewords are decoded in one batch (the result of a codeword does not depend on the others);
 * counters only cover the codewords the header announces, as in the reference. */
typedef struct ultra_hip_frame_result {
    int32_t success;             /* RxFrameResult::success                                              */
    int32_t is_ping;             /* RxFrameResult::is_ping                                              */
    int32_t frame_type;          /* RxFrameResult::frame_type (protocol::v2::FrameType)                 */
    int32_t codewords_ok;        /* RxFrameResult::codewords_ok                                         */
    int32_t codewords_failed;    /* RxFrameResult::codewords_failed                                     */
    int32_t expected_codewords;  /* RxPipeline::getExpectedCodewords(): total_cw while waiting, else 0  */
    int32_t frame_len;           /* RxFrameResult::frame_data.size()                                    */
    int32_t status;              /* ULTRA_HIP_FRAME_*                                                   */
} ultra_hip_frame_result;
enum ultra_hip_frame_status {
    ULTRA_HIP_FRAME_CW0_FAILED = 0,        /* no codeword, or CW0 did not decode                        */
    ULTRA_HIP_FRAME_BAD_HEADER = 1,        /* CW0 decoded, not a valid v2 header                        */
    ULTRA_HIP_FRAME_WAITING = 2,           /* header announces more codewords than were handed in       */
    ULTRA_HIP_FRAME_CODEWORDS_FAILED = 3,  /* some codeword of the frame did not decode                 */
    ULTRA_HIP_FRAME_COMPLETE = 4,          /* all codewords decoded, frame_data reassembled             */
    ULTRA_HIP_FRAME_PING = 5               /* raw "ULTR" (or its inversion) instead of a codeword       */
};
int ultra_hip_decode_frames_batch(ultra_hip_ctx* ctx, const float* d_soft, size_t frame_stride, uint32_t n_soft,
                                  size_t n_frames, ultra_hip_frame_result* d_results, uint8_t* d_frame_data,
                                  size_t frame_data_stride);

/* Transmit-side stimulus on the device: what one Monte-Carlo trial of the harnesses
 * builds before the receiver runs (tools/test_nvis_mode.cpp:35-93), for frames first_frame ..
 * first_frame + n_frames - 1: payload of floor(k/8) random bytes per codeword -> LDPCEncoder::encode
 * (src/fec/ldpc_encoder.cpp:193-257) -> OFDMModulator::generatePreamble + modulate
 * (src/ofdm/modulator.cpp:202-283,348-532) -> whole signal scaled to a 0.5 peak -> channel -> the
 * frame_samples from the first data symbol on (the SYNCED entry's input).
 *   channel_kind 0 none, 1 AWGN at snr_db, 2 Watterson two-tap (gains 0.707/0.707, delay_ms, doppler_hz,
 *                fading restarted per frame: src/sim/hf_channel.hpp:106-168,258-275)
 *   d_audio      [n_frames][frame_stride >= frame_samples] f32
 *   d_payload    [n_frames][floor(k/8)] bytes of the first codeword (what ultra_hip_count_errors compares)
 * Payload, codewords, transmitted samples and scaling are bit-identical to the test oracle's
 * uo_make_batch for the same (seed, frame index); the channels use a per-sample counter-based
 * generator and are statistically, not bitwise, equivalent to the serial CPU draws. */
int ultra_hip_make_batch(ultra_hip_ctx* ctx, uint64_t seed, uint64_t first_frame, size_t n_frames, int channel_kind,
                         float snr_db, float delay_ms, float doppler_hz, float* d_audio, size_t frame_stride,
                         uint8_t* d_payload);

/* The radio's tuning error as the harnesses model it: WattersonChannel::applyCFO (src/sim/hf_channel.hpp:161-232 — mix
 * down from 1500 Hz, 48-tap running mean, rotate by the offset, mix back up), applied by WattersonChannel::process after
 * the noise when abs(cfo_hz) > 0.001 (:163-165).  Every row is shifted by a freshly constructed channel (phase 0), as
 * the harnesses build one channel per trial; rows shorter than 256 samples and offsets within +-0.001 Hz are copied
 * unchanged, as the reference leaves them.  Out of place (d_out must not alias d_in); bit-identical to the reference
 * (the oracle's uo_channel_apply_cfo is pinned to it and the tests compare bitwise).
 *   d_in  [n_frames][in_stride >= n_samples] f32      d_out [n_frames][out_stride >= n_samples] f32 */
int ultra_hip_channel_cfo_batch(ultra_hip_ctx* ctx, const float* d_in, size_t in_stride, float* d_out, size_t out_stride,
                                uint32_t n_samples, size_t n_frames, float cfo_hz);

/* The same transmission as a RAW stream for the end-to-end entry (ultra_hip_receive_batch): `lead` samples of
 * silence, the preamble (OFDMModulator::generatePreamble, 7 symbols), the frame's data symbols, `tail` samples of
 * silence; scaled to a 0.5 peak; channel_kind 0 none, 1 AWGN on every sample of the stream at snr_db relative to the
 * mean power of the transmission (tools/test_nvis_mode.cpp:62-86).
 *   d_audio  [n_streams][stream_stride >= lead + 7 (fft + cp) + frame_samples + tail] f32 */
int ultra_hip_make_raw_batch(ultra_hip_ctx* ctx, uint64_t seed, uint64_t first_frame, size_t n_streams, int channel_kind,
                             float snr_db, uint32_t lead, uint32_t tail, float* d_audio, size_t stream_stride,
                             uint8_t* d_payload);
/* ... through a channel with parameters: channel_kind 2 = WattersonChannel::process (src/sim/hf_channel.hpp:106-168,258-275;
 * delay_ms / doppler_hz as in ultra_hip_make_batch: two paths, gains 0.707 / 0.707, fading restarted per stream) over the
 * TRANSMISSION (preamble + data symbols: the channel sees what ultra_hip_make_batch's sees, so a stream's frame fades like
 * that batch's frame), its noise on every sample of the stream; kinds 0 and 1 as above (delay_ms, doppler_hz unused). */
int ultra_hip_make_raw_batch_channel(ultra_hip_ctx* ctx, uint64_t seed, uint64_t first_frame, size_t n_streams, int channel_kind,
                                     float snr_db, float delay_ms, float doppler_hz, uint32_t lead, uint32_t tail, float* d_audio,
                                     size_t stream_stride, uint8_t* d_payload);

/* LDPC-only stimulus on the device (BASELINE.json configs[3]: "LDPC R1/4 ... SNR sweep -11..+30 dB"):
 * codewords first_cw .. first_cw + n_cw - 1 of the context's code rate as BPSK over AWGN, handed to the decoder as
 * LLRs 2y / sigma^2 with sigma^2 = 1 / (2 Es/N0).  The reference has no LDPC-only SNR harness (its decoder tests
 * build +-LLR vectors by hand, tests/test_comprehensive_modem.cpp:185-240), so the generator is this build's; it keeps
 * the harness shape of tools/test_mode_snr.cpp:18-109 (random payload of floor(k/8) bytes -> LDPCEncoder::encode,
 * src/fec/ldpc_encoder.cpp:193-257 -> noise -> decodeSoft -> compare the payload bytes) and the sign convention
 * LLR > 0 <=> bit 0.
 *   d_llr      [n_cw][648] f32
 *   d_payload  [n_cw][floor(k/8)] bytes (what ultra_hip_count_errors compares)
 * Payload bytes are those of ultra_hip_make_batch for the same (seed, index).  The noise is counter-based
 * Box-Muller with libm-exact logf / sincosf / sqrtf (csrc/pinned_math.h), so the test oracle's twin
 * (uo_make_llr_batch, plain libm calls) reproduces every LLR bit for bit: any codeword of any sweep point can be
 * regenerated and decoded on the host. */
int ultra_hip_make_llr_batch(ultra_hip_ctx* ctx, uint64_t seed, uint64_t first_cw, size_t n_cw, float esn0_db,
                             float* d_llr, uint8_t* d_payload);

/* Channel deinterleaver of the production receive path, fused into the decoder's LLR load.
 * Replaces RxPipeline::setInterleaverConfig(bits_per_symbol) + deinterleaveCodewords
 * (src/gui/modem/rx_pipeline.cpp:24-31,475-491): every 648-LLR codeword handed to
 * ultra_hip_ldpc_decode_batch / ultra_hip_demod_decode_batch is first passed through
 * ChannelInterleaver(bits_per_symbol, 648)::deinterleave (src/fec/ldpc_decoder.cpp:575-617),
 * out[j] = in[(j * step) % 648].  bits_per_symbol = 0 switches it off (the default; the
 * Monte-Carlo harnesses of SURVEY.md 3.1 do not interleave).  The LLRs returned by
 * ultra_hip_demod_batch / the llr output of the fused call stay in channel order.
 * The legacy Modem's Interleaver(32,32) (src/modem/modem.cpp:116,161) reads past the 648 soft
 * bits it is given and is not reproduced. */
int ultra_hip_set_deinterleave(ultra_hip_ctx* ctx, uint32_t bits_per_symbol);

/* The same fusion for ANY permutation of the 648 soft bits of a codeword: out[j] = in[h_index[j]], n = 648, every
 * entry < 648 (validated: the kernel gathers through the table).  Covers the reference's row-column Interleaver
 * (src/fec/ldpc_decoder.cpp:454-466; deinterleave(soft) :530-540 is out[i] = in[permutation_[i]], permutation_[i] =
 * (i % cols) * rows + i / cols — e.g. Interleaver(6, 108) of tools/test_throughput.cpp:78-134 and any other 648-entry
 * layout).  Takes precedence over ultra_hip_set_deinterleave; h_index = NULL or n = 0 switches the table off. */
int ultra_hip_set_deinterleave_table(ultra_hip_ctx* ctx, const uint16_t* h_index, uint32_t n);
/* The step of ChannelInterleaver(bits_per_symbol, total) (findCoprimeStep, src/fec/ldpc_decoder.cpp:547-573): permutation[i]
 * = (i * step) % total.  Host arithmetic, no context: what ultra_hip_set_deinterleave fuses into the decoder, handed out
 * for callers that need the permutation itself (the transmit side's interleave; projectultra_amd/host/hip_ldpc_decoder.cpp).
 * ABI 8. */
int ultra_hip_channel_interleaver_step(uint32_t bits_per_symbol, uint32_t total, uint32_t* step);

/* Per-kernel timing (diagnostics; bench.py's roofline object uses it): while enabled, every kernel
 * launch of this context is bracketed by a pair of HIP events on the context's stream.  read()
 * waits for the recorded launches, adds their elapsed milliseconds and launch counts per kernel
 * class into ms[] / launches[] (ULTRA_HIP_K_N entries each, overwritten) and forgets them. */
enum ultra_hip_kernel_class {
    ULTRA_HIP_K_INIT_STATE = 0, /* init_state_kernel */
    ULTRA_HIP_K_MIX_FFT = 1,    /* mix_fft_kernel: toBaseband + FFT, one launch per OFDM symbol */
    ULTRA_HIP_K_TRACK = 2,      /* track_kernel (carrier half: interpolate + equalize + demap) / train_kernel, one per symbol */
    ULTRA_HIP_K_LDPC = 3,       /* ldpc_decode_kernel */
    ULTRA_HIP_K_COUNT = 4,      /* count_errors_kernel */
    ULTRA_HIP_K_ACQUIRE = 5,    /* acquire_kernel */
    ULTRA_HIP_K_CHIRP = 6,      /* chirp_sync_kernel */
    ULTRA_HIP_K_PILOT = 7,      /* track_pilot_kernel: pilot half of the channel update, one per data symbol */
    ULTRA_HIP_K_WALK = 8,       /* cfo_walk_kernel: phase table of the next symbol's CFO rotation, one per symbol */
    ULTRA_HIP_K_N = 9,
    /* ABI 8, ultra_hip_profile_read_items only: the transform's ROTATING instance (mix_fft2_kernel<N, true>: a CFO phase table
     * per frame, bit-exact sincosf per sample) apart from the instance without rotation, which stays class 1 — two kernels for
     * an issue model, five times apart in instructions per item (profiles/issue.json, tools/issue_model.py). */
    ULTRA_HIP_K_MIX_FFT_ROT = 9,
    ULTRA_HIP_K_N2 = 10
};
int ultra_hip_profile_enable(ultra_hip_ctx* ctx, int enable);
int ultra_hip_profile_read(ultra_hip_ctx* ctx, float* ms, uint32_t* launches);
/* ... with ULTRA_HIP_K_N2 entries per array and the WORK ITEMS the recorded launches covered (items[]): frame-symbols for the
 * transform, the walk and the two tracking kernels (a launch may cover one symbol of every frame, two, or all), codewords for
 * the decoder, streams for the acquisition; 0 where a class has no natural item.  What an instruction count collected at one
 * batch size and launch structure is scaled by when it is quoted for another (bench.py). */
int ultra_hip_profile_read_items(ultra_hip_ctx* ctx, float* ms, uint32_t* launches, uint64_t* items);

/* Which of its own FALL-BACK paths a context has taken, and what the decoder's screen last decided.  ABI 10.
 *
 * Every path below computes the same results bit for bit (tests/test_gpu_fallbacks.py, tests/test_gpu_status.py) but not at the
 * same speed, and none of them used to be visible to the caller — the reference's convention is "failure = false / empty, never
 * silent" ("Errors"), and a batch that quietly runs on the slower chain is a silent failure of a performance
 * contract.  `flags` is sticky: a bit stays set from the first launch that took the path until ultra_hip_clear_status.
 * bench.py prints the word and does not call a line "default path" with any bit set. */
enum {
    ULTRA_HIP_ST_DEMOD_WORKSPACE_FALLBACK = 0x01, /* the per-(frame, symbol) bin / tracker workspace could not be had (allocation failed, or
                                                    * above ultra_hip_set_workspace_limit): the batch ran the per-symbol launch chain */
    ULTRA_HIP_ST_LDPC_MESSAGE_KERNEL = 0x02,      /* a decode launch ran the message-passing kernel instead of the totals kernel */
    ULTRA_HIP_ST_LDS_PROBE_FAILED = 0x04,         /* ... because ultra_hip_create's probe found dynamic LDS not at address 0 */
    ULTRA_HIP_ST_SCREEN_LIST_UNAVAILABLE = 0x08,  /* a launch that qualified for the decoder's screen ran without it: no work list */
    ULTRA_HIP_ST_ACQ_CACHE_UNAVAILABLE = 0x10,    /* ultra_hip_acquire_stream_batch searched without its per-stream metric cache */
    ULTRA_HIP_ST_FORCED_FALLBACK_CHAIN = 0x20,    /* ULTRA_HIP_FALLBACK_CHAIN=1 in the environment when the context was created */
    ULTRA_HIP_ST_FORCED_MESSAGE_KERNEL = 0x40,    /* ULTRA_HIP_LDPC_MESSAGES=1 */
    ULTRA_HIP_ST_SCREEN_OVERRIDDEN = 0x80         /* ULTRA_HIP_LDPC_SCREEN=0 or =2 */
};
typedef struct ultra_hip_path_status {
    uint32_t flags;                /* ULTRA_HIP_ST_* */
    uint32_t screen_launches;      /* decode launches that ran the screen's sample since the context was created */
    uint32_t screen_sample_n;      /* of the LAST such launch: codewords sampled, */
    uint32_t screen_sample_clean;  /*   how many of them satisfied every parity row as received, */
    uint32_t screen_gate;          /*   the count at which the full pass runs (0 = forced), */
    uint32_t screen_gate_open;     /*   whether it ran, */
    uint32_t screen_dirty;         /*   and, if so, how many codewords it left to the iterating kernel */
    uint32_t reserved;
} ultra_hip_path_status;
/* Synchronises the context's stream (the screen's counters live on the device). */
int ultra_hip_get_status(ultra_hip_ctx* ctx, ultra_hip_path_status* out);
int ultra_hip_clear_status(ultra_hip_ctx* ctx);
/* Cap, in bytes, on EACH of the two per-(frame, symbol) demodulator workspaces (bins: n_frames * n_symbols * 128 or 256 complex
 * values; tracker records likewise).  A batch that would need more runs the per-symbol chain on per-frame buffers instead —
 * same results, ULTRA_HIP_ST_DEMOD_WORKSPACE_FALLBACK set.  0 = no cap (default).  For hosts that share the card, and for the
 * test that must see the flag. */
int ultra_hip_set_workspace_limit(ultra_hip_ctx* ctx, size_t bytes);

/* Convenience for hosts without their own device allocator (the C++ adapter
 * and the ctypes tests): hipMalloc/hipFree/hipMemcpy on the context's device. */
int ultra_hip_malloc(ultra_hip_ctx* ctx, size_t bytes, void** d_ptr);
int ultra_hip_free(ultra_hip_ctx* ctx, void* d_ptr);
int ultra_hip_memcpy_h2d(ultra_hip_ctx* ctx, void* d_dst, const void* h_src, size_t bytes);
int ultra_hip_memcpy_d2h(ultra_hip_ctx* ctx, void* h_dst, const void* d_src, size_t bytes);
/* ultra_hip_memcpy_h2d without the wait: the bytes are copied into a pinned staging ring of the context before the call
 * returns (h_src is the caller's again at once) and travel to the device in stream order, ahead of every launch issued
 * after it.  For the live adapters, whose process() call then costs ONE blocking wait — the download of its answer — instead
 * of one per transfer.  Transfers above 256 KB take the blocking copy.  ABI 8. */
int ultra_hip_memcpy_h2d_async(ultra_hip_ctx* ctx, void* d_dst, const void* h_src, size_t bytes);
int ultra_hip_memset(ultra_hip_ctx* ctx, void* d_dst, int value, size_t bytes);

/* The latency path of ONE live stream (ABI 10): a call's answer without a copy command and without hipStreamSynchronize.
 *   ultra_hip_host_block   pinned host memory the device writes DIRECTLY (zero copy): *h_ptr is the host's view, *d_ptr the device's
 *                          view of the same bytes — hand d_ptr (+ offsets) to any entry point as an output buffer.  Owned by the
 *                          context, freed by ultra_hip_destroy.  Small blocks only (a live adapter's soft bits and tracker state):
 *                          a batch's outputs belong in device memory.
 *   ultra_hip_stage_input  copies `bytes` from h_src into the context's pinned staging ring and returns the DEVICE view of the slot
 *                          in *d_view: a kernel launched next reads the bytes from host memory itself — no copy command either
 *                          (for inputs read once: a codeword's 648 soft bits).  h_src is the caller's again at once; the slot stays
 *                          valid until the ring wraps (1 MB; a wrap waits for the stream first).
 *   ultra_hip_stream_post  orders the 32-bit store `*d_flag = value` (d_flag inside a host block) behind everything issued on the
 *                          context's stream so far, visible to the host when it lands.
 *   ultra_hip_host_wait    spins on the host view of that word until it equals `value`; after timeout_us microseconds without it,
 *                          falls back to a stream synchronisation (and returns ULTRA_HIP_ERR_HIP if the word still differs).
 * What a SYNCED process() call or a single-codeword decodeSoft costs beside its kernels is then one store the host polls for,
 * instead of a device-to-host copy and its completion wait (profiles/r06_live_latency.txt). */
int ultra_hip_host_block(ultra_hip_ctx* ctx, size_t bytes, void** h_ptr, void** d_ptr);
int ultra_hip_stage_input(ultra_hip_ctx* ctx, const void* h_src, size_t bytes, void** d_view);
int ultra_hip_stream_post(ultra_hip_ctx* ctx, uint32_t* d_flag, uint32_t value);
int ultra_hip_host_wait(ultra_hip_ctx* ctx, const volatile uint32_t* h_flag, uint32_t value, uint32_t timeout_us);

/* Device self-test of the pinned libm restatement (projectultra_amd/csrc/
 * pinned_math.h): out[i] = fn(a[i] [, b[i]]) evaluated on the GPU, so tests
 * can compare with the host libm the reference calls.
 * fn: 0 sinf, 1 cosf, 2 atanf, 3 atan2f(a, b), 4 hypotf(a, b), 5 logf, 6 sqrtf (the last two: stimulus only). */
int ultra_hip_selftest_math(ultra_hip_ctx* ctx, int fn, const float* d_a, const float* d_b,
                            float* d_out, size_t n);

#ifdef __cplusplus
}
#endif
#endif /* ULTRA_HIP_H */
