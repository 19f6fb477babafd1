// ldpc_totals_kernel.h — LDPC(648) scaled-min-sum decode for gfx950, "totals" formulation: one wavefront per codeword, all six
// codes of the reference (round 2: R2/3, R3/4, R5/6; round 4: R1/4, R1/3, R1/2 through per-round degree profiles).
//
// Same arithmetic as LDPCDecoder::Impl::decodeBP (src/fec/ldpc_decoder.cpp:153-259) — see ldpc_kernel.h's header for
// the value-identical reformulations shared with it (minima with |.| modifiers and minNum semantics, signs as lane
// masks, the +-50 clamp deferred to the reader, parity bits in the registers of their row's lane, unchecked variables
// untouched).  What differs from the message-passing kernel there is WHO computes the variable-to-check message and what
// travels through LDS:
//
//   reference / message kernel   variable step:  v2c[e] = clamp(total[j] - c2v[e])  per EDGE, stored, read by the row
//   here                          the variable publishes ONE number, total[j]; the ROW computes total[j] - c2v[e] with
//                                 its own previous c2v[e], which it kept in a register.  Same operands, same operation.
//
// DEGREE PROFILES are template arguments (four bits per round, ldpc_prof): round r of the row phase gathers, computes and
// stores S_r information-edge slots plus the parity edge (a leave-one-out network of S_r + 1 magnitudes), round r of the
// variable phase gathers and adds D_r messages; the R planes of round r are plane_base[r] .. plane_base[r] + S_r - 1.  The
// regular codes (every row six information edges, every variable three) have S = 6 and D = 3 throughout; the low-rate codes'
// rows hold 1..6 edges and their variables 4..13, and rows / variables sit in rounds whose profile covers their degree
// (tools/ldpc_place_low.cpp -> ldpc_placement_low.h).  A shorter row's spare slots gather the T pad word FLT_MAX (positive,
// never a minimum), a variable's spare edges the R pad word -0.0f (the exact neutral addend).
//
// LDS per codeword-iteration (R3/4: 3 row rounds x 6 slots, 6 variable rounds x 3 edges):
//   row side       18 gathers of totals (ds_read_b32, 2 cycles)      18 lane-linear stores of c2v (ds_write_addtid_b32, 2)
//   variable side  18 gathers of c2v   (ds_read_b32, 2 cycles)        6 lane-linear stores of totals (add-TID, 2)
//   = 120 LDS-pipeline cycles (+ the plan's residual gather collisions, P.extra_cycles) against 180 of the message kernel,
//   whose check step stores through address registers (ds_write_b32: 4 cycles each); R1/4: 114 instructions against 70
//   read / four-cycle-write pairs.
// Both gathers are made conflict-free by WHERE variables and rows sit (tools/ldpc_place.cpp, tools/ldpc_place_low.cpp ->
// ldpc_placement.h, ldpc_placement_low.h -> build_ldpc_tplan).  The stopping test needs no extra pass: a row that has
// gathered the totals of its variables evaluates its own parity equation on them — the exact checkParity (:139-151) of the
// PREVIOUS iteration, decided at the top of the next one from the very gathers the check step needs anyway.
#ifndef ULTRA_LDPC_TOTALS_KERNEL_H
#define ULTRA_LDPC_TOTALS_KERNEL_H

#include <hip/hip_runtime.h>
#include "device_types.h"
#include "ldpc_kernel.h"
#include "ldpc_screen_kernel.h"

namespace ultra_hip {
namespace dev {
typedef float pk2 __attribute__((ext_vector_type(2)));     // two single-precision lanes of one v_pk_* instruction

// Diagnostic build only (-DUH_LDPC_STAMPS, tools/ldpc_stalls.py): shader-clock time of one codeword split over the phases of
// the loop below, summed over its iterations — record of kLdpcStampWords 64-bit words per codeword: [0] start, [1] end,
// [2] LLRs landed + staged (wait for the asynchronous copy, first totals), [3] row phases (gathers of totals, parity
// verdict, check step, c2v stores issued), [4] drain behind the row phase, [5] variable phases (gathers of c2v, sums,
// totals stored), [6] drain behind the variable phase, [7] outputs, [8] iterations executed, [9] HW_ID | XCC_ID << 32.
// Every stamp is an s_memtime whose result is waited for (lgkmcnt), i.e. each phase ends with its LDS operations
// complete; the product build contains none of this.
constexpr int kLdpcStampWords = 10;
#ifdef UH_LDPC_STAMPS
__device__ unsigned long long* g_ldpc_stamps = nullptr;
#define UH_LD_NOW() __builtin_readcyclecounter()
#define UH_LD_ACC(k) do { const unsigned long long now_ = UH_LD_NOW(); ld_acc[k] += now_ - ld_t; ld_t = now_; asm volatile("; UHLDSTAMP %0" ::"n"(k)); } while (0)
#else
#define UH_LD_ACC(k) do {} while (0)
#endif

// S lane-linear stores of c[0..S-1] to the planes at byte offsets BASE, BASE + 256, ...: ONE asm statement (M0 must not change
// between its write and the stores)
template <int S, unsigned BASE>
__device__ __forceinline__ void tprof_store_planes(const float (&c)[7]) {
    static_assert(S >= 1 && S <= 6, "information-edge slots per row round");
    if constexpr (S == 1)
        asm volatile("s_mov_b32 m0, 0\n\ts_nop 0\n\tds_write_addtid_b32 %0 offset:%1" :: "v"(c[0]), "n"(BASE) : "m0", "memory");
    else if constexpr (S == 2)
        asm volatile("s_mov_b32 m0, 0\n\ts_nop 0\n\tds_write_addtid_b32 %0 offset:%2\n\tds_write_addtid_b32 %1 offset:%3"
                     :: "v"(c[0]), "v"(c[1]), "n"(BASE), "n"(BASE + 256) : "m0", "memory");
    else if constexpr (S == 3)
        asm volatile("s_mov_b32 m0, 0\n\ts_nop 0\n\tds_write_addtid_b32 %0 offset:%3\n\tds_write_addtid_b32 %1 offset:%4\n\t"
                     "ds_write_addtid_b32 %2 offset:%5"
                     :: "v"(c[0]), "v"(c[1]), "v"(c[2]), "n"(BASE), "n"(BASE + 256), "n"(BASE + 512) : "m0", "memory");
    else if constexpr (S == 4)
        asm volatile("s_mov_b32 m0, 0\n\ts_nop 0\n\tds_write_addtid_b32 %0 offset:%4\n\tds_write_addtid_b32 %1 offset:%5\n\t"
                     "ds_write_addtid_b32 %2 offset:%6\n\tds_write_addtid_b32 %3 offset:%7"
                     :: "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]), "n"(BASE), "n"(BASE + 256), "n"(BASE + 512), "n"(BASE + 768) : "m0", "memory");
    else if constexpr (S == 5)
        asm volatile("s_mov_b32 m0, 0\n\ts_nop 0\n\tds_write_addtid_b32 %0 offset:%5\n\tds_write_addtid_b32 %1 offset:%6\n\t"
                     "ds_write_addtid_b32 %2 offset:%7\n\tds_write_addtid_b32 %3 offset:%8\n\tds_write_addtid_b32 %4 offset:%9"
                     :: "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]), "v"(c[4]), "n"(BASE), "n"(BASE + 256), "n"(BASE + 512), "n"(BASE + 768),
                        "n"(BASE + 1024) : "m0", "memory");
    else
        asm volatile("s_mov_b32 m0, 0\n\ts_nop 0\n\t"
                     "ds_write_addtid_b32 %0 offset:%6\n\tds_write_addtid_b32 %1 offset:%7\n\t"
                     "ds_write_addtid_b32 %2 offset:%8\n\tds_write_addtid_b32 %3 offset:%9\n\t"
                     "ds_write_addtid_b32 %4 offset:%10\n\tds_write_addtid_b32 %5 offset:%11"
                     :: "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]), "v"(c[4]), "v"(c[5]), "n"(BASE), "n"(BASE + 256), "n"(BASE + 512),
                        "n"(BASE + 768), "n"(BASE + 1024), "n"(BASE + 1280) : "m0", "memory");
}

// the variable phase's lane-linear stores of the VR totals to the T planes (byte offsets 0, 256, ...), one asm statement
template <int VR>
__device__ __forceinline__ void tprof_store_totals(const float (&t)[VR]) {
    static_assert(VR >= 3 && VR <= 7, "variable rounds of the six codes");
    if constexpr (VR == 3)
        asm volatile("s_mov_b32 m0, 0\n\ts_nop 0\n\tds_write_addtid_b32 %0\n\tds_write_addtid_b32 %1 offset:256\n\tds_write_addtid_b32 %2 offset:512"
                     :: "v"(t[0]), "v"(t[1]), "v"(t[2]) : "m0", "memory");
    else if constexpr (VR == 4)
        asm volatile("s_mov_b32 m0, 0\n\ts_nop 0\n\tds_write_addtid_b32 %0\n\tds_write_addtid_b32 %1 offset:256\n\t"
                     "ds_write_addtid_b32 %2 offset:512\n\tds_write_addtid_b32 %3 offset:768"
                     :: "v"(t[0]), "v"(t[1]), "v"(t[2]), "v"(t[VR > 3 ? 3 : 0]) : "m0", "memory");
    else if constexpr (VR == 5)
        asm volatile("s_mov_b32 m0, 0\n\ts_nop 0\n\tds_write_addtid_b32 %0\n\tds_write_addtid_b32 %1 offset:256\n\t"
                     "ds_write_addtid_b32 %2 offset:512\n\tds_write_addtid_b32 %3 offset:768\n\tds_write_addtid_b32 %4 offset:1024"
                     :: "v"(t[0]), "v"(t[1]), "v"(t[2]), "v"(t[VR > 3 ? 3 : 0]), "v"(t[VR > 4 ? 4 : 0]) : "m0", "memory");
    else if constexpr (VR == 6)
        asm volatile("s_mov_b32 m0, 0\n\ts_nop 0\n\tds_write_addtid_b32 %0\n\tds_write_addtid_b32 %1 offset:256\n\t"
                     "ds_write_addtid_b32 %2 offset:512\n\tds_write_addtid_b32 %3 offset:768\n\tds_write_addtid_b32 %4 offset:1024\n\t"
                     "ds_write_addtid_b32 %5 offset:1280"
                     :: "v"(t[0]), "v"(t[1]), "v"(t[2]), "v"(t[VR > 3 ? 3 : 0]), "v"(t[VR > 4 ? 4 : 0]), "v"(t[VR > 5 ? 5 : 0]) : "m0", "memory");
    else
        asm volatile("s_mov_b32 m0, 0\n\ts_nop 0\n\tds_write_addtid_b32 %0\n\tds_write_addtid_b32 %1 offset:256\n\t"
                     "ds_write_addtid_b32 %2 offset:512\n\tds_write_addtid_b32 %3 offset:768\n\tds_write_addtid_b32 %4 offset:1024\n\t"
                     "ds_write_addtid_b32 %5 offset:1280\n\tds_write_addtid_b32 %6 offset:1536"
                     :: "v"(t[0]), "v"(t[1]), "v"(t[2]), "v"(t[VR > 3 ? 3 : 0]), "v"(t[VR > 4 ? 4 : 0]), "v"(t[VR > 5 ? 5 : 0]),
                        "v"(t[VR > 6 ? 6 : 0]) : "m0", "memory");
}

// RR row rounds, VR variable rounds; RPROF / VPROF: S_r / D_r, four bits per round (LdpcTPlan::row_prof / var_prof).  WAVES:
// wavefronts per SIMD the register budget is sized for.  WANT_TOTAL: also write the final a-posteriori LLRs (parity tests).
// The totals kernel names its LDS planes by immediate offsets behind M0 = 0 (ds_write_addtid_b32), i.e. it assumes that a
// kernel whose only LDS is the dynamic allocation sees it at LDS address 0.  ultra_hip_create checks that ONCE with this
// probe (same declaration, same launch shape) and keeps the message-passing kernel for the context if it ever fails —
// a synchronous, loud decision instead of a guard inside the decoder that nobody could see.
__global__ void ldpc_lds_base_probe_kernel(unsigned* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    if (threadIdx.x == 0) { lds_raw[0] = 1; out[0] = (unsigned)(size_t)lds_raw; }
}

// COMPACT (launches without a fused deinterleaver; R3/4): only the values the decoder reads are staged — the n_checked
// information bits that have checks and the m parity bits, 487 of R3/4's 648: 1,948 B instead of 2,592, which is the
// difference between 18 and 19 workgroups per CU (8,348 B each) — measured -4 % on codewords that run all 50 iterations
// (profiles/r05_variants/r05_probe_ldpc_r34_workgroups_per_cu.txt) — and a quarter less to fetch.  Word w of the staged
// row = position w of the memory row for w < n_checked, position w + (k - n_checked) behind that.
template <int RR, int VR, unsigned long long RPROF, unsigned long long VPROF, bool WANT_TOTAL, int WAVES, bool COMPACT = false>
__global__ __launch_bounds__(kLdpcThreads, WAVES) void ldpc_totals_kernel(
    const LdpcTPlan* __restrict__ Pp, const float* __restrict__ llr, size_t llr_stride, int n_cw,
    uint8_t* __restrict__ bytes, int32_t* __restrict__ iters, uint8_t* __restrict__ okv,
    float* __restrict__ llr_total, unsigned int* __restrict__ work_counter, int llr_step,
    const uint16_t* __restrict__ llr_perm, int block_len, int block_stride, const unsigned* __restrict__ work_list, unsigned gate) {
    // work_list (nullable; ldpc_screen_kernel.h): with the screen's gate on, the codewords to decode are work_list[0 .. n_work) —
    // those whose channel values do not already satisfy every row; the screen has finished the others.  Both words were written
    // by earlier launches of the stream and no wavefront of this one changes them.
    int n_work = n_cw;
    const unsigned* lst = nullptr;
    if (work_list != nullptr && work_counter[kScreenCtlSample] >= gate) { lst = work_list; n_work = (int)work_counter[kScreenCtlDirty]; }
    if ((int)blockIdx.x >= n_work) return;         // any one wavefront drains every queue: the others are not needed
    constexpr bool kPacked = RR >= 8;          // R1/4's instance: packed single-precision subtract / multiply / add (row phase below)
    // block_len > 0 (ultra_hip_ldpc_decode_blocks): codeword c is row (c / block_len) * block_stride + c % block_len of the
    // LLR array — several equally long runs of rows inside a larger array (one code rate's share of a mode grid) decoded
    // by ONE launch; results stay dense (row c).
    auto llr_row = [&](int c) -> size_t {
        return block_len > 0 ? (size_t)(c / block_len) * (size_t)block_stride + (size_t)(c % block_len) : (size_t)c;
    };
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const LdpcTPlan& P = *Pp;
    const int lane = threadIdx.x;
    // The LDS layout is a function of the instance (build_ldpc_tplan uses the same formulas and ultra_hip.hip checks
    // them): compile-time offsets, and the plan's scalars in locals — the "memory" clobber of the store asm would
    // otherwise make the compiler reload them from the plan inside the iteration loop.
    constexpr int NPLANES = tprof_planes_before(RPROF, RR), D = ldpc_prof_max(VPROF, VR);
    constexpr unsigned T_PAD = VR * 256, R_BASE = T_PAD + 128, R_PAD = R_BASE + NPLANES * 256, STAGE_V = R_PAD + 128,
                       STAGE_P = STAGE_V + VR * 256;
    const int k = P.k, max_iterations = P.max_iterations, decoded_bytes = P.decoded_bytes, n_checked = P.n_checked;
    auto ldsf = [&](unsigned byte_off) -> float& { return *reinterpret_cast<float*>(lds_raw + byte_off); };
    const unsigned lds_base = (unsigned)(size_t)lds_raw;
    // lane-linear store: word (plane_byte_off / 4 + lane) <- x, no address register (ds_write_addtid_b32: 2 LDS cycles)
    auto store_linear = [&](unsigned plane_byte_off, float x) {
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tds_write_addtid_b32 %0" :: "v"(x), "s"(lds_base + plane_byte_off) : "m0", "memory");
    };

    // ---- per-lane slice of the plan, in registers for the whole launch ----
    bool row_on[RR], var_on[VR];
    unsigned taddr[RR][6], caddr[VR][D];
#pragma unroll
    for (int r = 0; r < RR; ++r) {
        row_on[r] = P.row_check[r * 64 + lane] != 0xFFFFu;
#pragma unroll
        for (int t = 0; t < 6; ++t) if (t < ldpc_prof(RPROF, r)) taddr[r][t] = P.row_taddr[(r * 64 + lane) * 6 + t];
    }
#pragma unroll
    for (int r = 0; r < VR; ++r) {
        var_on[r] = P.var_id[r * 64 + lane] != 0xFFFFu;
#pragma unroll
        for (int q = 0; q < D; ++q) if (q < ldpc_prof(VPROF, r)) caddr[r][q] = P.var_caddr[(r * 64 + lane) * kTPlanDmax + q];
    }
    if (lane < 32) { ldsf(T_PAD + 4u * lane) = kFltMax; ldsf(R_PAD + 4u * lane) = -0.0f; }      // one pad word per bank

    // Work queue and prefetch as in ldpc_decode_kernel: kLdpcQueues interleaved queues; the next codeword is claimed and
    // its LLRs are fetched (asynchronously, straight into the slot-indexed staging planes) while the current one decodes.
    int queue = (int)(blockIdx.x % kLdpcQueues), dry = 0;
    auto claim = [&]() -> int {
        for (;;) {
            int ticket = 0;
            if (lane == 0) ticket = (int)atomicAdd(work_counter + queue * kLdpcQueueStride, 1u);
            const int c = __builtin_amdgcn_readfirstlane(ticket) * kLdpcQueues + queue;
            if (c < n_work) return lst ? (int)lst[c] : c;
            if (++dry == kLdpcQueues) return -1;
            queue = (queue + 1) % kLdpcQueues;
        }
    };
    // channel LLR index of variable j: the production path's channel deinterleaver fused as a gather index
    // (ChannelInterleaver::deinterleave, ldpc_decoder.cpp:609-617: out[j] = in[(j * step) % 648]; step 1 = identity)
    // llr_perm (nullable): a general gather table out[j] = in[llr_perm[j]] (ultra_hip_set_deinterleave_table)
    auto src_index = [&](int j) -> unsigned { return llr_perm ? (unsigned)llr_perm[j] : (unsigned)(j * llr_step) % (unsigned)kLdpcN; };
    // Source element of each staging slot of this lane: a property of the launch, not of the codeword — computed once
    // (the plan lookup, the table lookup or the modulo would otherwise sit in front of every asynchronous copy).
    // kRowStage (round 5; the instances whose VR + RR staging planes hold a whole row anyway — R1/4, R1/3, R1/2, R2/3 — and
    // R3/4, whose nine planes grow by 288 bytes to the row's 2,592 and still fit 18 workgroups per CU: ldpc_stage_bytes; for
    // R5/6 the row would cost a workgroup per CU): the next codeword's channel values are copied AS THEY LIE IN
    // MEMORY by 11 coalesced wave-wide copies, and the permutation to slots (with the channel deinterleaver's) is the LDS read
    // that moves a value into its variable's total.  Before, every lane fetched the value of ITS variable: 11 wave-instructions
    // of 64 scattered 4-byte reads per codeword, each touching most of the row's 21 cache lines; a codeword that converges at
    // once waited 11,800 cycles for them, 9,100 now (profiles/r05_ldpc_stalls_r14*.txt).
    constexpr bool kRowStage = ldpc_row_stage(VR, RR);     // (false everywhere in the UH_NO_ROW_STAGE variant build: device_types.h)
    static_assert(!COMPACT || (kRowStage && RR == 3 && VR == 6), "the compact row is R3/4's staged row");
    // COMPACT: the memory row's positions [n_checked, k) = [325, 486) are not staged (R3/4: ldpc_decoder.cpp:64-137 connects the
    // first 325 information bits: 323 with three checks, one with two, one with one); where position p of the memory row sits in the staged row (no deinterleaver: src_index = identity)
    constexpr int kCompactK = 486, kCompactChecked = 325, kCompactGap = kCompactK - kCompactChecked;
    auto staged_word = [&](unsigned p) -> unsigned { return (COMPACT && (int)p >= kCompactChecked) ? p - (unsigned)kCompactGap : p; };
    unsigned short src_v[VR], src_p[RR];        // kRowStage: byte address of the slot's value in the staged row; else its index in the row
#pragma unroll
    for (int r = 0; r < VR; ++r) {
        const unsigned j = P.var_id[r * 64 + lane];
        if constexpr (kRowStage) src_v[r] = (unsigned short)(STAGE_V + 4u * ((j != 0xFFFFu) ? staged_word(src_index((int)j)) : 0u));
        else src_v[r] = (j != 0xFFFFu) ? (unsigned short)src_index((int)j) : (unsigned short)0xFFFFu;
    }
#pragma unroll
    for (int r = 0; r < RR; ++r) {
        const unsigned i = P.row_check[r * 64 + lane];
        if constexpr (kRowStage) src_p[r] = (unsigned short)(STAGE_V + 4u * ((i != 0xFFFFu) ? staged_word(src_index(k + (int)i)) : 0u));
        else src_p[r] = (i != 0xFFFFu) ? (unsigned short)src_index(k + (int)i) : (unsigned short)0xFFFFu;
    }
    auto fetch = [&](int c) {
        const float* src = llr + llr_row(c) * llr_stride;
        float* stage_v = reinterpret_cast<float*>(lds_raw + STAGE_V);
        if constexpr (COMPACT) {
            // a copy lands its lanes at consecutive words behind ONE base: the part of a 64-position piece in front of the gap
            // and the part behind it are copies of their own.  The instance is R3/4's: k and n_checked are ITS constants
            // (launch_ldpc checks the plan against them), so which lanes of which piece copy is decided at compile time —
            // pieces 0-4 whole, piece 5 its first five lanes, piece 6 nothing, piece 7 from lane 38 on, 8 and 9 whole, 10 eight lanes.
            ldpc_static_for(std::make_integer_sequence<int, (kLdpcN + 63) / 64>{}, [&](auto piece) {
                constexpr int i = decltype(piece)::value;
                constexpr int a_hi = kCompactChecked - 64 * i;                 // lanes [0, a_hi) lie in front of the gap
                constexpr int b_lo = kCompactK - 64 * i, b_hi = kLdpcN - 64 * i;   // lanes [b_lo, b_hi) lie behind it
                if constexpr (a_hi >= 64) __builtin_amdgcn_global_load_lds(src + lane + i * 64, stage_v + i * 64, 4, 0, 0);
                else if constexpr (a_hi > 0) { if (lane < a_hi) __builtin_amdgcn_global_load_lds(src + lane + i * 64, stage_v + i * 64, 4, 0, 0); }
                if constexpr (b_lo <= 0 && b_hi >= 64) __builtin_amdgcn_global_load_lds(src + lane + i * 64, stage_v + i * 64 - kCompactGap, 4, 0, 0);
                else if constexpr (b_lo < 64 && b_hi > 0) {
                    if (lane >= b_lo && lane < b_hi) __builtin_amdgcn_global_load_lds(src + lane + i * 64, stage_v + i * 64 - kCompactGap, 4, 0, 0);
                }
            });
        } else if constexpr (kRowStage) {
#pragma unroll
            for (int i = 0; i < (kLdpcN + 63) / 64; ++i)
                if (i * 64 + lane < kLdpcN) __builtin_amdgcn_global_load_lds(src + lane + i * 64, stage_v + i * 64, 4, 0, 0);
        } else {
            float* stage_p = reinterpret_cast<float*>(lds_raw + STAGE_P);
#pragma unroll
            for (int r = 0; r < VR; ++r)
                if (src_v[r] != 0xFFFFu) __builtin_amdgcn_global_load_lds(src + src_v[r], stage_v + r * 64, 4, 0, 0);
#pragma unroll
            for (int r = 0; r < RR; ++r)
                if (src_p[r] != 0xFFFFu) __builtin_amdgcn_global_load_lds(src + src_p[r], stage_p + r * 64, 4, 0, 0);
        }
    };

    // Lanes without a row / variable in some round run the SAME instruction stream on harmless operands (their gather
    // addresses point to the pad words, their lane-linear stores hit words nobody reads) instead of branching around
    // it: every divergent `if` costs the scalar unit an exec save, a branch and a restore, and the scalar unit — one
    // instruction per ~4 cycles per SIMD (profiles/r02_issue_table.txt) — is what this loop saturates first.  Only the
    // parity verdict needs the row masks.
    unsigned long long row_mask[RR];
#pragma unroll
    for (int r = 0; r < RR; ++r) row_mask[r] = __ballot(row_on[r]);
    // This kernel owns the whole LDS allocation of its workgroup (dynamic only), so the array starts at LDS address 0
    // and the add-TID stores can name their plane by an immediate offset behind M0 = 0.  Should a toolchain ever place
    // it elsewhere, refuse to run rather than store to the wrong words.
    if (lds_base != 0u) { if (blockIdx.x == 0 && lane == 0 && n_cw > 0) { iters[0] = -1; okv[0] = 0; } return; }

    // Tickets are drawn ONE CODEWORD AHEAD of their use: the atomic's round trip (~2,000 cycles at the head of every codeword
    // in profiles/r04_ldpc_stalls_before.txt) then runs under a decode instead of in front of one.  draw() leaves the ticket
    // in lane 0's register without waiting for it; take() turns it into a codeword of the queue it was drawn from, or — the
    // queue has run dry: the launch's tail — falls back to the synchronous claim(), which moves on to the other queues.
    int ticket_v = 0, ticket_queue = queue;
    auto draw = [&]() {
        ticket_queue = queue;
        ticket_v = 0;
        if (lane == 0) ticket_v = (int)atomicAdd(work_counter + queue * kLdpcQueueStride, 1u);
    };
    auto take = [&]() -> int {
        const int c = __builtin_amdgcn_readfirstlane(ticket_v) * kLdpcQueues + ticket_queue;
        if (c < n_work) return lst ? (int)lst[c] : c;
        if (dry >= kLdpcQueues) return -1;
        return claim();             // counts the dry queue again at worst: `dry` only has to reach kLdpcQueues eventually
    };
    int cw = claim();
    if (cw >= 0) { fetch(cw); draw(); }
    while (cw >= 0) {
#ifdef UH_LDPC_STAMPS
        unsigned long long ld_acc[8] = {};
        unsigned long long ld_t = UH_LD_NOW();
        const unsigned long long ld_t0 = ld_t;
        int ld_iters = 0;
#endif
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // staged LLRs have landed
        __syncthreads();
        float llr_v[VR], llr_p[RR], c2v[RR][7];
#pragma unroll
        for (int r = 0; r < VR; ++r) {
            llr_v[r] = ldsf(kRowStage ? (unsigned)src_v[r] : STAGE_V + (unsigned)(r * 64 + lane) * 4u);   // an empty slot reads a stale word: never used
            store_linear((unsigned)(r * 256), llr_v[r]);                       // total before any iteration = llr_in
        }
#pragma unroll
        for (int r = 0; r < RR; ++r) {
            llr_p[r] = ldsf(kRowStage ? (unsigned)src_p[r] : STAGE_P + (unsigned)(r * 64 + lane) * 4u);
#pragma unroll
            for (int t = 0; t < 7; ++t) c2v[r][t] = 0.0f;                      // check_to_var starts at 0 (:175)
        }
        __syncthreads();
        const int cw_next = take();                                           // the staging planes are free from here on
        if (cw_next >= 0) { fetch(cw_next); draw(); }

        int it = 0, ok = 0;
        float tpar[RR];                                                        // total of the row's parity bit (WANT_TOTAL)
        UH_LD_ACC(2);
        for (;;) {
            // ---- row phase: gather totals; parity equations of the previous iteration; check step ----
            const float cap = (it == 0) ? kFltMax : 50.0f;                    // clamp deferred to the reader, see ldpc_kernel.h
            // checkParity (:139-151) passes iff EVERY row's equation holds.  Round 0's rows are always evaluated; the
            // other rounds only while no failing row has been seen — far from convergence the first round settles it.
            // Iteration 0 gathers the channel values themselves (T = llr_in, check_to_var = 0): if THEIR hard decisions
            // satisfy every row, the reference converges at iteration 0 with exactly those bits — in a satisfied row the
            // product of the other edges' signs is the edge's own sign, so every check message pushes its variable
            // further the way it already points and no total changes sign (any input: a NaN counts as +, as `x < 0` does).
            // The decode then ends here, after the verdict, instead of after one full iteration and the next one's gathers.
            bool all_hold = it > 0 || (!WANT_TOTAL && max_iterations > 0);     // wave-uniform
            ldpc_static_for(std::make_integer_sequence<int, RR>{}, [&](auto round) {
                constexpr int r = decltype(round)::value;
                constexpr int S = ldpc_prof(RPROF, r);                         // information-edge slots of this round; the parity edge is edge S
                float v[7], mn[7], tot[6];
                bool ng[7], par = false;
                // total of the parity bit after the previous iteration: llr_in + c2v (:206-213; the bit has one edge)
                const float tp = llr_p[r] + c2v[r][S];
                if (WANT_TOTAL) tpar[r] = tp;
#pragma unroll
                for (int t = 0; t < S; ++t) tot[t] = ldsf(taddr[r][t]);       // a row with fewer edges reads a pad word: FLT_MAX
                if (all_hold) {                                                // uniform branch
                    bool synd = tp < 0;
#pragma unroll
                    for (int t = 0; t < S; ++t) synd ^= (tot[t] < 0);          // hard decisions of :227-230
                    all_hold = (__ballot(synd) & row_mask[r]) == 0ull;
                }
                // var_to_check = llr_total - check_to_var (:216-219).  kPacked (R1/4's instance): two edges per instruction — a
                // packed single-precision subtraction is two IEEE subtractions in one 4-cycle issue slot (two 2-cycle ones cost
                // 4.6).  Measured: R1/4 4.74 -> 4.53 ms per 2^17 codewords; the instances that already saturate the vector unit
                // at 4-5 wavefronts per SIMD lose 1-4 % with it (register pairs), so they keep the scalar form.
                {
                    float full[7];
#pragma unroll
                    for (int t = 0; t < S; ++t) full[t] = tot[t];
                    full[S] = tp;
                    if constexpr (kPacked) {
#pragma unroll
                        for (int t = 0; t + 1 <= S; t += 2) {
                            const pk2 d = pk2{full[t], full[t + 1]} - pk2{c2v[r][t], c2v[r][t + 1]};
                            v[t] = d.x; v[t + 1] = d.y;
                        }
                        if (((S + 1) & 1) != 0) v[S] = full[S] - c2v[r][S];
                    } else {
#pragma unroll
                        for (int t = 0; t <= S; ++t) v[t] = full[t] - c2v[r][t];
                    }
                }
#pragma unroll
                for (int t = 0; t <= S; ++t) { ng[t] = v[t] < 0; par ^= ng[t]; }
                leave_one_out_min<S + 1>(v, cap, mn);
                {
                    float mag[7];
                    if constexpr (kPacked) {
#pragma unroll
                        for (int t = 0; t + 1 <= S; t += 2) {
                            const pk2 m = pk2{mn[t], mn[t + 1]} * pk2{0.75f, 0.75f};
                            mag[t] = m.x; mag[t + 1] = m.y;
                        }
                        if (((S + 1) & 1) != 0) mag[S] = mn[S] * 0.75f;
                    } else {
#pragma unroll
                        for (int t = 0; t <= S; ++t) mag[t] = mn[t] * 0.75f;
                    }
#pragma unroll
                    for (int t = 0; t <= S; ++t) c2v[r][t] = (par != ng[t]) ? -mag[t] : mag[t];   // sign * min * 0.75f (:201)
                }
                // S lane-linear stores, planes named by immediate offsets behind M0 = 0
                tprof_store_planes<S, R_BASE + tprof_planes_before(RPROF, r) * 256>(c2v[r]);
            });
            UH_LD_ACC(3);
#ifdef UH_LDPC_STAMPS
            ++ld_iters;
#endif
            if (all_hold) { ok = 1; if (it > 0) --it; break; }                 // checkParity passed after iteration it - 1 (or holds at 0)
            if (it >= max_iterations) break;
            __syncthreads();                  // one wavefront per workgroup: an LDS drain (measured: no cost against leaving it out)
            UH_LD_ACC(4);
            // ---- variable phase: total = llr_in + sum of the check messages in ascending check order (:206-213) ----
            float tots[VR];
            {
                // all gathers of the phase first, then the sums: one wait for the lot instead of one per round
                float c[VR][D];
#pragma unroll
                for (int r = 0; r < VR; ++r)
#pragma unroll
                    for (int q = 0; q < D; ++q) if (q < ldpc_prof(VPROF, r)) c[r][q] = ldsf(caddr[r][q]);  // a missing edge reads a pad word: -0.0f
                // the sums of two rounds side by side where their degrees agree (packed additions), in ascending check order each
                ldpc_static_for(std::make_integer_sequence<int, (VR + 1) / 2>{}, [&](auto pair) {
                    constexpr int r = 2 * decltype(pair)::value;
                    if constexpr (kPacked && r + 1 < VR && ldpc_prof(VPROF, r) == ldpc_prof(VPROF, r + 1)) {
                        pk2 tot = pk2{llr_v[r], llr_v[r + 1]};
#pragma unroll
                        for (int q = 0; q < D; ++q) if (q < ldpc_prof(VPROF, r)) tot += pk2{c[r][q], c[r + 1][q]};
                        tots[r] = tot.x; tots[r + 1] = tot.y;
                    } else {
#pragma unroll
                        for (int rr = r; rr < r + 2 && rr < VR; ++rr) {
                            float tot = llr_v[rr];
#pragma unroll
                            for (int q = 0; q < D; ++q) if (q < ldpc_prof(VPROF, rr)) tot += c[rr][q];
                            tots[rr] = tot;
                        }
                    }
                });
            }
            tprof_store_totals<VR>(tots);
            UH_LD_ACC(5);
            __syncthreads();
            UH_LD_ACC(6);
            ++it;
        }
        const int iters_out = ok ? it : max_iterations;
        __syncthreads();

        // ---- outputs: hard decisions of the k information bits packed MSB-first (:238-258) ----
        // Totals of the checked variables (j < n_checked) are in T (those of the last completed iteration — the row phase does
        // not touch T); an unchecked variable's total is its channel LLR, read again from memory.  ONE memory round trip: the
        // eight slot words of the lane's byte and the channel values of its unchecked bits are independent loads issued
        // together (which bits are unchecked is a property of the index, LdpcTPlan::n_checked), then the eight LDS reads.
        // (Round 3's loop took the bits one at a time — slot word, wait, LDS or memory read, wait — 9,300 cycles per codeword
        // in the stamps of profiles/r04_ldpc_stalls_before.txt, as long as thirteen iterations.)
        const float* src = llr + llr_row(cw) * llr_stride;
        uint8_t* ob = bytes + (size_t)cw * decoded_bytes;
        for (int b = lane; b < decoded_bytes; b += kLdpcThreads) {
            unsigned sl[8];
            float ch[8];
            // every load unconditional, at an address that is valid for every lane (a load under a lane mask ends in a wait of
            // its own): bits that are not of the kind read word 0 of the table / element `lane` of the row and drop the value
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int j = 8 * b + t;
                sl[t] = (unsigned)P.var_slot_of[(j < n_checked) ? j : 0];
            }
            unsigned idx[8];
            if (llr_perm) {                                                       // wave-uniform: one branch around all eight
#pragma unroll
                for (int t = 0; t < 8; ++t) idx[t] = (unsigned)llr_perm[min(8 * b + t, kLdpcN - 1)];
            } else {
#pragma unroll
                for (int t = 0; t < 8; ++t) idx[t] = (unsigned)(min(8 * b + t, kLdpcN - 1) * llr_step) % (unsigned)kLdpcN;
            }
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int j = 8 * b + t;
                ch[t] = src[(j >= n_checked && j < k) ? idx[t] : (unsigned)lane];
            }
            unsigned v = 0;
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int j = 8 * b + t;
                const float in_t = ldsf(((j < n_checked) ? sl[t] : 0u) * 4u);
                const float tot = (j < n_checked) ? in_t : ((j < k) ? ch[t] : 0.0f);      // beyond k: bit 0
                v = (v << 1) | ((tot < 0) ? 1u : 0u);
            }
            ob[b] = (uint8_t)v;
        }
        if (WANT_TOTAL) {
            float* ot = llr_total + (size_t)cw * kLdpcN;
            for (int j = lane; j < k; j += kLdpcThreads) {
                const unsigned sl = P.var_slot_of[j];
                ot[j] = (sl != 0xFFFFu) ? ldsf(sl * 4u) : src[src_index(j)];
            }
#pragma unroll
            for (int r = 0; r < RR; ++r) {
                const unsigned i = P.row_check[r * 64 + lane];
                // no iteration ran (max_iterations == 0): the totals are the channel values themselves
                if (i != 0xFFFFu) ot[k + (int)i] = (max_iterations > 0) ? tpar[r] : llr_p[r];
            }
        }
        if (lane == 0) { iters[cw] = iters_out; okv[cw] = (uint8_t)ok; }
#ifdef UH_LDPC_STAMPS
        UH_LD_ACC(7);
        if (g_ldpc_stamps != nullptr && lane == 0) {
            unsigned hw, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            unsigned long long* r = g_ldpc_stamps + (size_t)cw * kLdpcStampWords;
            r[0] = ld_t0; r[1] = ld_t;
            for (int q = 2; q < 8; ++q) r[q] = ld_acc[q];
            r[8] = (unsigned long long)ld_iters; r[9] = ((unsigned long long)xcc << 32) | hw;
        }
#endif
        __syncthreads();                                                       // T is rewritten by the next codeword
        cw = cw_next;
    }
}

}  // namespace dev
}  // namespace ultra_hip
#endif
