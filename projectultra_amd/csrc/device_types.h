// device_types.h — PODs shared by the host table builder and the HIP kernels.
#ifndef ULTRA_DEVICE_TYPES_H
#define ULTRA_DEVICE_TYPES_H

#include <stdint.h>

namespace ultra_hip {

constexpr int kMaxCarriers = 64;   // num_carriers <= 64 (reference presets use 30 / 59)
constexpr int kMaxFft = 1024;
constexpr int kLdpcN = 648;        // LDPC block length (src/fec/ldpc_decoder.cpp:12-14)
constexpr int kLdpcMaxEdges = 2560;  // R1/4 has 2437 edges
constexpr int kLdpcMaxChecks = 486;

struct alignas(8) c32 { float re, im; };   // 8-byte aligned: one ds_read_b64 / global_load_dwordx2 per value

// Per-configuration demodulator constants.  One copy in HBM per context; every
// field is read with wave-uniform addresses (scalar loads) except the tables
// indexed by lane.  "slot" = position of a carrier in the reference's layout
// loop (src/ofdm/demodulator.cpp:46-69): slot 0 is k=-neg_limit, DC skipped.
struct DemodConst {
    int32_t fft, log2_fft, cp, sym_len;
    int32_t n_train, n_data_sym, n_carriers, n_data, n_pilot, n_interp;
    int32_t modulation, bits, differential, presynced;
    int32_t llrs_per_symbol, llrs_per_frame, frame_samples;
    int32_t fq_half;            // 32 or 64: a frame's Fq row holds bins [0, fq_half) and [fft - fq_half, fft) (every used carrier)
    float ce_margin;            // soft_demap::getCEErrorMargin(mod)
    float sample_rate;          // (float) config.sample_rate
    float symbol_duration;      // (float)getSymbolDuration() / (float)sample_rate
    float max_timing;           // 50.0f * (fft_size / 512.0f)
    float fft_f;                // (float) fft_size
    float mixer_phase_end;      // NCO::phase_ after a whole frame
    // ModemConfig::adaptive_eq_* (types.hpp:170-174): 0 off (also for the differential modulations, which never reach
    // the branch: channel_equalizer.cpp:769), 1 LMS, 2 RLS; decision_directed 0/1
    int32_t adaptive_eq, decision_directed;
    float lms_mu, rls_lambda;
    float fade_k;               // 0.1f / (float)n_data (FADE_THRESHOLD_RATIO over the carrier count): the erasure test's screen
    float _pad0;
    double two_pi_symbol_duration;  // 2.0f * M_PI * symbol_duration (double)
    int16_t bin[kMaxCarriers];        // slot -> fft bin
    int16_t k_of[kMaxCarriers];       // slot -> signed carrier number (k > fft/2 -> k - fft)
    int16_t data_slot[kMaxCarriers];  // data carrier i -> slot
    int16_t pilot_slot[kMaxCarriers]; // pilot i -> slot
    int16_t interp_slot[kMaxCarriers];
    int16_t interp_lo[kMaxCarriers];  // slot of lower pilot or -1
    int16_t interp_hi[kMaxCarriers];
    float interp_alpha[kMaxCarriers];
    c32 pilot_seq[kMaxCarriers];      // +-1 + 0j
    c32 sync_seq[kMaxCarriers];       // Zadoff-Chu
    // Where the transform puts a kept bin inside the frame's Fq row: natural slot s (bin s for s < fq_half, bin
    // fft - fq_half + (s - fq_half) above) -> position.  The PILOTS come first, in pilot order, then the data carriers in
    // data order, then the bins nobody reads: the pilot half (track_pilot_kernel) then touches one 128-byte line of the
    // row (15 pilots; two lines for up to 32) instead of all four — it runs at the memory system's ceiling.
    uint8_t fq_pos[2 * kMaxCarriers];
};

// Per-rate Tanner graph, CSR by check (row-major edge order of H_rows) and CSR
// by variable (edges of a variable in ascending check order — the order
// llr_total accumulates in, src/fec/ldpc_decoder.cpp:206-213).
struct LdpcConst {
    int32_t k, m, n, edges, max_iterations, decoded_bytes, _pad[2];
    uint16_t row_ptr[kLdpcMaxChecks + 2];
    uint16_t col[kLdpcMaxEdges];
    uint16_t var_ptr[kLdpcN + 2];
    uint16_t var_edge[kLdpcMaxEdges];
};

// Execution plan of the LDPC kernel (ldpc_kernel.h).
// Messages live in an LDS word array whose addresses are chosen by a proper 32-edge-colouring
// of the access structure (colour = LDS bank), so that every wave instruction of the check step
// (lane = row, one edge position) and of the variable step (lane = variable, one edge index)
// touches 32 distinct banks per half-wave: no bank conflicts by construction.
//   row_addr[6*i + t]   word address of information edge t of row i (0xFFFF: row has no edge t)
//   act_*               information bits that have at least one check, full-degree ones first;
//                       act_addr[a*14 + q] = word address of edge q of the variable, ascending check order
//   row_mask / act_mask random 32-bit masks of the syndrome filter (act_mask = xor of the row masks
//                       of the variable's checks)
//   row_col             variable index of each information edge (exact parity check, rare path)
// Rows are held in SLOT order: sorted by information degree, highest first (row_id[slot] = check index i, whose
// parity bit is variable k + i), and the active variables likewise by degree, so that the lanes of a round have
// (nearly) the same degree.  prof_* pack, four bits per round, the largest and the smallest degree of the
// round's lanes (rows: over the rows that exist; variables: over all 64 lanes, an idle lane counting 0): the
// kernel is instantiated on these profiles and touches exactly the slots a round has.
// linear != 0 (codes whose rows all have six information edges and whose variables all have the same degree but a
// few): the message of edge q of the variable in lane l of round r lives at word (r * dmax + q) * 64 + l, so the
// variable step reads lane-linearly and stores with ds_write_addtid_b32 (no address registers, half the store
// cost); the variables' lanes and the rows' half-waves are then chosen on the host so that the CHECK step's
// gathers still meet 32 distinct banks (build_ldpc_plan_linear).  Row slots / variable slots may then have gaps:
// a slot exists iff row_deg / act_deg is non-zero.
// The parity bit of a row never touches LDS: its message stays in a register of the row's lane.
constexpr int kLdpcPlanDmax = 14;
constexpr int kLdpcPlanMaxActive = 576;
constexpr int kLdpcPlanMaxRows = 512;
struct LdpcPlan {
    int32_t k, m, n, edges, max_iterations, decoded_bytes, n_active, row_rounds, var_rounds, dmax;
    int32_t var_rounds_full, rows_full, msg_words, _pad;
    uint8_t row_deg[kLdpcPlanMaxRows];
    uint16_t row_addr[kLdpcPlanMaxRows * 6];
    uint16_t row_col[kLdpcPlanMaxRows * 6];
    uint32_t row_mask[kLdpcPlanMaxRows];
    uint16_t act_var[kLdpcPlanMaxActive];
    uint8_t act_deg[kLdpcPlanMaxActive];
    uint32_t act_mask[kLdpcPlanMaxActive];
    uint16_t act_addr[kLdpcPlanMaxActive * kLdpcPlanDmax];
    uint16_t row_id[kLdpcPlanMaxRows];
    uint64_t prof_rmax, prof_rmin, prof_vmax, prof_vmin;
    int32_t row_identity, linear;
};

// Execution plan of the "totals" LDPC kernel (ldpc_totals_kernel.h) for the codes whose rows all have six (R2/3:
// five or six) information edges and whose variables have degree <= 4 — R2/3, R3/4, R5/6.  One total per variable
// lives in the lane-linear LDS array T[round][lane]; the check-to-variable messages live in the lane-linear array
// R[round][slot][lane]; each side gathers from the other's array.  LDS byte offsets (from the dynamic LDS base):
//   T words      [0, var_rounds * 256)                     word (round * 64 + lane) = the variable in that slot
//   T pad words  32 at t_pad (one per bank): +FLT_MAX (the phantom operand of a row with fewer than six information edges)
//   R words      [r_base, r_base + row_rounds * 6 * 256)   word ((round * 6 + t) * 64 + lane) = edge slot t of the row
//   R pad words  32 at r_pad (one per bank): -0.0f (the neutral addend of a variable with fewer than dmax edges)
//   staging      channel LLRs of the NEXT codeword: slot-indexed planes stage_v [var_rounds][64], stage_p [row_rounds][64], or —
//                ldpc_row_stage — the 648 values as they lie in memory at stage_v (ldpc_totals_kernel.h, kRowStage)
constexpr int kTPlanRowRounds = 8, kTPlanVarRounds = 7, kTPlanDmax = 13;
// Instances that stage the next codeword's row in memory order: those whose slot planes hold 648 values anyway, and R3/4
// (nine planes, 2,304 B), whose staging grows to the row's 2,592 B and still fits 18 workgroups per CU (8,992 B of
// 163,840 / 18 = 9,102: tools/ubench/lds_granule.hip — the allocation has no coarser granule than that).  R5/6's six planes
// would have to grow by 1,056 B and lose a workgroup per CU.
#ifdef UH_NO_ROW_STAGE                         // variant build for the A/B (tools/ab_ldpc.sh): slot-indexed fetch everywhere
constexpr bool ldpc_row_stage(int, int) { return false; }
#else
constexpr bool ldpc_row_stage(int var_rounds, int row_rounds) { return var_rounds + row_rounds >= 9; }
#endif
// A row-staged instance never touches the slot planes, so its staging area is the row's 2,592 B exactly — 224 to 480 B less than
// the planes of R2/3, R1/2 and R1/3 took until round 5, which is one more workgroup per CU for each of them (14 -> 15).
constexpr int ldpc_stage_bytes(int var_rounds, int row_rounds) {
    return ldpc_row_stage(var_rounds, row_rounds) ? 648 * 4 : (var_rounds + row_rounds) * 256;
}
struct LdpcTPlan {
    int32_t valid, k, m, n, max_iterations, decoded_bytes, row_rounds, var_rounds, dmax;
    int32_t t_pad, r_base, r_pad, stage_v, stage_p, lds_bytes, extra_cycles;
    int32_t n_checked;                                    // variables j < n_checked have checks, n_checked <= j < k have none (validated)
    int32_t n_planes;                                     // R planes: sum of the row profile
    // Degree profiles, four bits per round (ldpc_prof): row_prof round r = S_r, the information-edge slots of its rows (the
    // R planes of round r are plane_base[r] .. plane_base[r] + S_r - 1); var_prof round r = D_r, the edges of its variables.
    // The regular codes (R2/3, R3/4, R5/6: ldpc_totals_kernel.h) have S_r = 6 and D_r = dmax throughout.
    uint64_t row_prof, var_prof;
    int32_t plane_base[kTPlanRowRounds + 1];
    int32_t _pad[1];
    uint16_t row_check[kTPlanRowRounds * 64];             // slot -> check index i (parity bit = variable k + i); 0xFFFF: empty
    uint16_t row_taddr[kTPlanRowRounds * 64 * 6];         // byte offset of the T word gathered by edge slot t (t_pad: none)
    uint16_t var_id[kTPlanVarRounds * 64];                // slot -> variable index; 0xFFFF: empty
    uint16_t var_caddr[kTPlanVarRounds * 64 * kTPlanDmax];  // byte offset of the R word of edge q, ascending check order (r_pad: none)
    uint16_t var_slot_of[kLdpcN];                         // variable -> slot; 0xFFFF: the variable has no check
};

// The decoder's screen (ldpc_screen_kernel.h): where in a codeword's row AS IT LIES IN MEMORY (the fused channel deinterleaver
// applied) the hard bits of each parity equation and of each output bit sit.  Made on the device from the Tanner graph and the
// context's deinterleaver setting (ldpc_screen_prepare_kernel), re-made when that setting changes.
constexpr int kScreenEdges = 7;              // edges of a row, the parity bit's included (max_check_degree 6 + identity part)
constexpr int kScreenZeroBit = 660;          // a position beyond the 648 of a row: its hard bit is always 0 ("no edge")
constexpr int kScreenOutPos = 656;           // 8 positions for each of up to 82 output bytes
struct LdpcScreenPos {
    uint16_t row_pos[kTPlanRowRounds * kScreenEdges * 64];   // [round][edge][lane]: row = round * 64 + lane
    uint16_t out_pos[kScreenOutPos];                          // information bit j -> position; j >= k: kScreenZeroBit
};

}  // namespace ultra_hip
#endif
