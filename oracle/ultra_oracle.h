/*
 * oracle/ultra_oracle.h — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C11) of the secup/ProjectUltra receive hot path and
 * of the transmit/channel pieces needed to make stimulus.  Every function in
 * ultra_oracle.c cites the reference file:line it follows.  The restatement is
 * PINNED: tests/test_oracle_vs_ref.py checks it bit-for-bit against the
 * compiled reference (oracle/_ref/libultra_ref.so, built by oracle/Makefile
 * from /root/reference) and tests/test_oracle_golden.py checks it against the
 * committed fixtures in tests/golden/ (generated from that compiled reference
 * by tests/golden/make_golden.py) and against the reference's own known-answer
 * tests (tests/test_rng.cpp:24-39, tests/test_multiblock_ldpc.cpp).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load libultra_oracle.so.  The product (projectultra_amd/, libultra_hip.so)
 * never includes, links or calls anything in this directory.
 */
#ifndef ULTRA_ORACLE_H
#define ULTRA_ORACLE_H

#include <stddef.h>
#include <stdint.h>
#include "../include/ultra_hip.h" /* ultra_hip_config / ultra_hip_geometry PODs only */

#ifdef __cplusplus
extern "C" {
#endif

/* ---- RNG -------------------------------------------------------------- */
typedef struct uo_mt19937 { uint32_t mt[624]; int idx; } uo_mt19937;
void uo_mt_seed(uo_mt19937* r, uint32_t seed);
uint32_t uo_mt_next(uo_mt19937* r);

/* ---- geometry --------------------------------------------------------- */
int uo_geometry(const ultra_hip_config* cfg, ultra_hip_geometry* geo);

/* ---- LDPC ------------------------------------------------------------- */
/* Tanner graph in CSR form, row-major edge order of H_rows. */
int uo_ldpc_graph(uint32_t rate, uint32_t* row_ptr /*[m+1]*/, uint32_t* col_idx /*[edges]*/,
                  uint32_t* k, uint32_t* m);
int uo_ldpc_encode(uint32_t rate, const uint8_t* data, uint32_t n, uint8_t* out, uint32_t cap);
int uo_ldpc_decode_soft(uint32_t rate, int max_iters, const float* llr, uint32_t n_llr,
                        uint8_t* out, uint32_t cap, int* success, int* iters);
/* n_cw independent 648-LLR codewords; llr_total_out may be NULL. */
int uo_ldpc_decode_batch(uint32_t rate, int max_iters, const float* llr, uint32_t n_cw,
                         uint8_t* out, uint32_t bytes_per_cw, int32_t* iters, uint8_t* ok,
                         float* llr_total_out);
int uo_ldpc_decode_batch_mt(uint32_t rate, int max_iters, const float* llr, uint32_t n_cw, int n_threads,
                            uint8_t* out, uint32_t bytes_per_cw, int32_t* iters, uint8_t* ok);
/* BPSK-over-AWGN LLRs 2y/sigma^2, sigma^2 = 1/(2 Es/N0), of codewords c0 .. c0+n_cw-1: twin of
 * ultra_hip_make_llr_batch, bit for bit (libm logf/sqrtf/sinf/cosf).  llr_out [n_cw][648], payload_out [n_cw][k/8]. */
int uo_make_llr_batch(uint32_t rate, uint64_t seed, uint64_t c0, uint32_t n_cw, float esn0_db,
                      float* llr_out, uint8_t* payload_out);
int uo_interleaver_deinterleave(uint32_t rows, uint32_t cols, const float* in, uint32_t n, float* out);
int uo_channel_interleaver_perm(uint32_t bits_per_symbol, uint32_t total, uint32_t* perm, uint32_t* inv);

/* ---- DSP primitives ---------------------------------------------------- */
int uo_fft_forward(uint32_t n, const float* in_ri, float* out_ri);
int uo_fft_inverse(uint32_t n, const float* in_ri, float* out_ri);
int uo_nco(float freq, float fs, uint32_t n, float* out_ri);

/* ---- acquisition (scope row f1): SEARCHING state of OFDMDemodulator::process fed in chunk-sample
 * calls; same outputs as ref_demod_acquire / ref_sc_metric / ref_lts_templates (oracle/ref_shim.cpp) */
int uo_acquire(const ultra_hip_config* c, const float* audio, uint32_t n, uint32_t chunk,
               uint32_t* found, uint32_t* fed_at_sync, uint32_t* sync_offset, float* coarse_cfo,
               uint32_t* refined_lts, uint32_t* data_start, float* noise_floor);
/* the preamble check of the SYNCED state (demodulator.cpp:605-657) on rx_buffer = audio[0, n); same outputs as
 * ref_midframe_search */
int uo_midframe_search(const ultra_hip_config* c, const float* audio, uint32_t n, uint32_t* found, uint32_t* sts_start,
                       uint32_t* refined_lts, uint32_t* consume, float* coarse_cfo);
int uo_sc_metric(const ultra_hip_config* c, const float* audio, uint32_t n, uint32_t offset,
                 float* corr, float* p_re, float* p_im, float* energy, float* noise_floor_io, uint32_t* has_energy);
int uo_lts_templates(const ultra_hip_config* c, float* I, float* Q, uint32_t cap);

/* ---- chirp synchronisation (scope row f4): sync::ChirpSync as configured by OFDMChirpWaveform; same
 * outputs as ref_chirp_detect / ref_chirp_generate / ref_chirp_templates (oracle/ref_shim.cpp) */
int uo_chirp_detect(float sample_rate, const float* x, uint32_t n, float threshold, int32_t* out, float* fout);
int uo_chirp_generate(float sample_rate, float tx_cfo_hz, float* out, uint32_t cap);
/* v2 wire format: RxPipeline::processFrame from the soft bits on, and the frame builder for stimulus */
uint16_t uo_crc16(const uint8_t* d, uint32_t n);
int uo_v2_parse_header(const uint8_t* d, uint32_t n, int32_t* out /*[4]*/);
int uo_v2_decode_frame(uint32_t rate, uint32_t deint_bps, int max_iters, const float* soft, uint32_t n_soft,
                       int32_t* res /*[8]*/, uint8_t* frame_data, uint32_t cap);
int uo_v2_build_frame(uint32_t rate, uint8_t type, uint8_t flags, uint16_t seq, uint32_t src_hash, uint32_t dst_hash,
                      const uint8_t* payload, uint32_t payload_len, int total_cw_override, uint8_t* codewords,
                      uint32_t cap_cw);
int uo_chirp_templates(float sample_rate, float* up_s, float* up_c, float* dn_s, float* dn_c, float* energies, uint32_t cap);

/* ---- demodulator ------------------------------------------------------- */
int uo_demod_tables(const ultra_hip_config* c, int32_t* data_idx, int32_t* pilot_idx,
                    float* pilot_seq_ri, int32_t* interp_i, float* interp_alpha,
                    float* sync_seq_ri, uint32_t* counts);
/* Same stage-dump layout as ref_demod_synced (oracle/ref_shim.cpp). */
int uo_demod_synced(const ultra_hip_config* c, const float* audio, uint32_t n_symbols, float cfo_hz,
                    float* llr_out, uint32_t llr_cap, float* stage_out);
int uo_demod_presynced(const ultra_hip_config* c, const float* audio, uint32_t n_samples,
                       int has_cfo, float cfo_hz, float cfo_phase,
                       float* llr_out, uint32_t llr_cap, float* H_out, float* scal_out);

/* Batched receive path = what libultra_hip.so computes; n_threads worker
 * threads over disjoint frame ranges (the timed CPU baseline).
 * state_out [n_frames][ULTRA_HIP_STATE_FLOATS] may be NULL; llr_out may be NULL;
 * bytes_out/iters_out/ok_out may be NULL (demod only). */
int uo_demod_decode_batch(const ultra_hip_config* c, const float* audio, size_t frame_stride,
                          const float* cfo_hz, const float* cfo_phase, size_t n_frames,
                          int n_threads, float* llr_out, float* state_out,
                          uint8_t* bytes_out, int32_t* iters_out, uint8_t* ok_out);

/* ---- transmit side + channel (stimulus) -------------------------------- */
int uo_modulate_frame(const ultra_hip_config* c, const uint8_t* encoded, uint32_t n_enc,
                      float* out, uint32_t cap, uint32_t* preamble_len);
int uo_modulate_presynced(const ultra_hip_config* c, const uint8_t* encoded, uint32_t n_enc,
                          float* out, uint32_t cap);
/* Watterson two-tap magnitude-fading channel (src/sim/hf_channel.hpp:106-168)
 * with a counter-based Gaussian source (statistically, not bitwise, equal to
 * the reference's mt19937 + std::normal_distribution stream). */
int uo_watterson(float snr_db, float delay_ms, float doppler_hz, float g1, float g2,
                 int fading, int multipath, int noise, uint64_t seed,
                 const float* in, uint32_t n, float* out);
/* WattersonChannel::applyCFO of a fresh channel (src/sim/hf_channel.hpp:161-232), in place. */
int uo_channel_apply_cfo(float cfo_hz, uint32_t sample_rate, float* samples, uint32_t n);
/* Synthetic batch for tests/bench: for frame f in [f0, f0+n): payload bytes
 * from a counter RNG (seed ^ f), LDPC encode, preamble+modulate, scale to 0.5
 * peak, channel (kind 0 = none, 1 = AWGN, 2 = Watterson), then keep the
 * frame_samples starting at the configured entry point.
 * audio_out [n][frame_samples], payload_out [n][payload_bytes]. */
int uo_make_batch(const ultra_hip_config* c, uint64_t seed, uint64_t f0, uint32_t n, int n_threads,
                  int channel_kind, float snr_db, float delay_ms, float doppler_hz,
                  float* audio_out, uint8_t* payload_out, uint32_t payload_bytes);

#ifdef __cplusplus
}
#endif
#endif
