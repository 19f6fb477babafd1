// ldpc_kernel.h — batched LDPC(648) scaled-min-sum decode for gfx950: one wavefront per codeword.
//
// Restates LDPCDecoder::Impl::decodeBP (src/fec/ldpc_decoder.cpp:153-259), flooding schedule:
//   init   v2c[e] = llr_in[col[e]]
//   iterate (<= max_iterations, 0-based index `it`):
//     check step   c2v[e] = (prod_{e'!=e} sgn(v2c[e'])) * min_{e'!=e}|v2c[e']| * 0.75f   (:181-202)
//     totals       total[j] = llr_in[j] + sum_e c2v[e], accumulated in ascending check order (:206-213)
//     var step     v2c[e] = clamp(total[col[e]] - c2v[e], -50, 50)                         (:216-224)
//     parity       all rows of H xor to 0 over (total < 0)  -> success, stop             (:227-235)
//   output first k hard bits packed MSB-first, lastIterations = it (or max on failure)
//
// Value-identical reformulations (every float operation the reference performs is performed
// here on the same operands in the same order):
//   * check step: "min over all others" of |v| as v_min3_f32 with |.| source modifiers (minNum: a NaN loses
//     against a number and an infinity against the FLT_MAX seed, exactly as in the reference's `abs < min`
//     update; the raw channel values are canonicalised where they become messages, so no signalling NaN
//     reaches a minimum); the sign is the row's sign
//     parity with edge e's own sign removed (`v < 0`: -0.0 and NaN count as positive, as in the
//     reference), applied to min*0.75f by selecting mag or -mag.
//   * var step: clamp(x, -50, 50) = std::max(-50.0f, std::min(50.0f, x)) keeps the sign of x (NaN
//     becomes +50, and `NaN < 0` is false) and turns |x| into min(|x|, 50).  The only reader of a
//     v2c message is the check step, which uses its sign and the minimum of the other edges'
//     magnitudes; so messages are stored unclamped and the minimum is capped at 50 instead (at
//     FLT_MAX in iteration 0, whose messages are the unclamped channel values).
//   * H = [H_data | I]: parity bit k+i has exactly one edge, the last one of row i.  Its message,
//     total and hard bit never leave the registers of the lane that owns row i.
//   * variables without any check (R3/4: info bits 325..485, R5/6: 217..539) keep
//     total = llr_in forever; they are never touched after the load.
//   * stopping test: the exact test "every row xors to 0" is preceded by a 32-bit linear filter
//     F = xor_i (syndrome_i ? mask_i : 0), evaluated from the variables' side with wave shuffles
//     (F = xor_j (hard_j ? xor_{i in N(j)} mask_i : 0)).  F != 0 proves the syndrome is non-zero;
//     only when F == 0 (a converged codeword, or a 2^-32 coincidence) is the exact row-by-row
//     test run.  The decision is therefore exactly the reference's.
//
// LDS: one f32 per information edge.  The word address of every edge is chosen on the host by a
// proper 32-colouring of the bipartite "instruction" graph (check-step half-waves x
// variable-step half-waves; colour = LDS bank), so every ds_read/ds_write of both steps is
// bank-conflict-free by construction (rocprof: 68 % of LDS cycles were conflicts before).  The
// Tanner graph itself lives in REGISTERS (per lane: addresses of the 6 edges of each row it owns
// and of the edges of each variable it owns, in ascending check order), loaded once per
// workgroup; workgroups are persistent and pull codewords from an atomic counter because
// iteration counts range from 0 to 50 per codeword.
#ifndef ULTRA_LDPC_KERNEL_H
#define ULTRA_LDPC_KERNEL_H

#include <hip/hip_runtime.h>
#include <type_traits>
#include <utility>
#include "device_types.h"

namespace ultra_hip {
namespace dev {

constexpr int kLdpcThreads = 64;
constexpr int kLdpcQueues = 8;        // work queues per launch
constexpr int kLdpcQueueStride = 32;  // unsigned ints between their counters (128 B)
constexpr int kLdpcQueueWords = kLdpcQueues * kLdpcQueueStride;
constexpr float kFltMax = 3.402823466e+38f;

__host__ __device__ inline size_t ldpc_lds_bytes(int msg_words) {
    return (size_t)msg_words * sizeof(float) + 656 + 648 * sizeof(float);
}

// xor over the 64 lanes (all active), result wave-uniform.  DPP cross-lane operands inside the
// 16-lane rows (no LDS round trips, unlike ds_bpermute shuffles), then four readlanes.
__device__ __forceinline__ unsigned wave_xor(unsigned v) {
    v ^= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);    // quad_perm [1,0,3,2]
    v ^= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);    // quad_perm [2,3,0,1]
    v ^= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true);   // row_half_mirror
    v ^= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true);   // row_mirror
    return (unsigned)__builtin_amdgcn_readlane((int)v, 0) ^ (unsigned)__builtin_amdgcn_readlane((int)v, 16) ^
           (unsigned)__builtin_amdgcn_readlane((int)v, 32) ^ (unsigned)__builtin_amdgcn_readlane((int)v, 48);
}

// min(|x|, |y|, |z|) and min(|x|, |y|) in one instruction each: the |.| is a source modifier of v_min3_f32 /
// v_min_f32, so the magnitudes are never formed separately.  IEEE minNum semantics: a NaN operand loses against a
// number, an infinity against the finite cap — exactly the reference's `if (abs < min) min = abs` update that
// starts from FLT_MAX (iteration 0) or from the clamp value 50 (see the check step).
__device__ __forceinline__ float fmin3abs(float x, float y, float z) {
    float r;
    asm("v_min3_f32 %0, |%1|, |%2|, |%3|" : "=v"(r) : "v"(x), "v"(y), "v"(z));
    return r;
}
__device__ __forceinline__ float fmin2abs(float x, float y) {
    float r;
    asm("v_min_f32_e64 %0, |%1|, |%2|" : "=v"(r) : "v"(x), "v"(y));
    return r;
}

// Leave-one-out minima of n magnitudes under a cap: mn[i] = min(cap, min_{j != i} |v[j]|).  Minima are exact and
// associative, so any network gives the reference's value.  n = 7 (a full row: six information edges and the
// parity bit) is the hand-counted 12-operation network; the general case runs a prefix and a suffix chain.
// The cap is wave-uniform (FLT_MAX in iteration 0, 50 afterwards): as a scalar operand of the minimum it needs no
// vector register and no per-round v_mov / v_cndmask to materialise it (one SGPR source per VOP3 instruction on gfx9).
__device__ __forceinline__ float fmin3abs_cap(float x, float y, float cap) {
    float r;
    asm("v_min3_f32 %0, |%1|, |%2|, %3" : "=v"(r) : "v"(x), "v"(y), "s"(__float_as_int(cap)));
    return r;
}
__device__ __forceinline__ float fmin2abs_cap(float x, float cap) {
    float r;
    asm("v_min_f32_e64 %0, |%1|, %2" : "=v"(r) : "v"(x), "s"(__float_as_int(cap)));
    return r;
}

template <int n>
__device__ __forceinline__ void leave_one_out_min(const float (&a)[7], float cap, float (&mn)[7]) {
    if constexpr (n == 7) {
        const float L2 = fmin3abs_cap(a[0], a[1], cap);         // min of edges 0..1
        const float L4 = fmin3abs(L2, a[2], a[3]);              // 0..3
        const float L6 = fmin3abs(L4, a[4], a[5]);              // 0..5
        const float R4 = fmin3abs_cap(a[5], a[6], cap);         // 5..6
        const float R3 = fmin2abs(R4, a[4]);                    // 4..6
        const float R2 = fmin3abs(R4, a[4], a[3]);              // 3..6
        mn[0] = fmin3abs(R2, a[2], a[1]);
        mn[1] = fmin3abs(a[0], R2, a[2]);
        mn[2] = fmin2abs(L2, R2);
        mn[3] = fmin3abs(L2, a[2], R3);
        mn[4] = fmin2abs(L4, R4);
        mn[5] = fmin3abs(L4, a[4], a[6]);
        mn[6] = L6;
    } else if constexpr (n == 2) {
        mn[0] = fmin2abs_cap(a[1], cap);
        mn[1] = fmin2abs_cap(a[0], cap);
    } else if constexpr (n == 3) {
        mn[0] = fmin3abs_cap(a[1], a[2], cap);
        mn[1] = fmin3abs_cap(a[0], a[2], cap);
        mn[2] = fmin3abs_cap(a[0], a[1], cap);
    } else {
        // pre[i] = min(cap, |a[0..i-1]|), suf[i] = min(|a[i..n-1]|); mn[i] = min(pre[i], suf[i+1])
        float pre[7], suf[7];
        pre[1] = fmin2abs_cap(a[0], cap);
#pragma unroll
        for (int i = 2; i < n; ++i) pre[i] = fmin2abs(pre[i - 1], a[i - 1]);
        suf[n - 1] = a[n - 1];
#pragma unroll
        for (int i = n - 2; i >= 1; --i) suf[i] = fmin2abs(suf[i + 1], a[i]);
        mn[0] = fmin2abs_cap(suf[1], cap);
#pragma unroll
        for (int i = 1; i < n - 1; ++i) mn[i] = fmin2abs(pre[i], suf[i + 1]);
        mn[n - 1] = pre[n - 1];
    }
}

// compile-time loop: f(std::integral_constant<int, 0>), f(<1>), ... (the round index selects template arguments)
template <int... Is, class F>
__device__ __forceinline__ void ldpc_static_for(std::integer_sequence<int, Is...>, F&& f) {
    (f(std::integral_constant<int, Is>{}), ...);
}

__host__ __device__ constexpr int ldpc_prof(unsigned long long p, int r) { return (int)((p >> (4 * r)) & 15ull); }
__host__ __device__ constexpr int ldpc_prof_max(unsigned long long p, int n) {
    int m = 0;
    for (int r = 0; r < n; ++r) m = ldpc_prof(p, r) > m ? ldpc_prof(p, r) : m;
    return m;
}

// RR = ceil(m / 64) row rounds, VR = ceil(n_active / 64) variable rounds.  RMAX / RMIN / VMAX / VMIN: the plan's
// degree profiles (LdpcPlan::prof_*, four bits per round): round r of the check step touches information-edge
// slots t < RMAX_r, without a per-lane test where t < RMIN_r; round r of the variable step touches edges
// q < VMAX_r, unconditionally where q < VMIN_r.  ROW_ID: the rows are permuted (LdpcPlan::row_id).
// WAVES = resident wavefronts per SIMD the register budget is sized for (5 -> 96 VGPRs, 4 -> 128, 3 -> 168).
// LINEAR: the plan's lane-linear layout (LdpcPlan::linear): the variable step addresses its messages by lane.
template <int RR, int VR, unsigned long long RMAX, unsigned long long RMIN, unsigned long long VMAX,
          unsigned long long VMIN, bool ROW_ID, bool LINEAR, bool WANT_TOTAL, int WAVES>
__global__ __launch_bounds__(kLdpcThreads, WAVES) void ldpc_decode_kernel(
    const LdpcPlan* __restrict__ Pp, const float* __restrict__ llr, size_t llr_stride, int n_cw,
    uint8_t* __restrict__ bytes, int32_t* __restrict__ iters, uint8_t* __restrict__ okv,
    float* __restrict__ llr_total, unsigned int* __restrict__ work_counter, int llr_step,
    const uint16_t* __restrict__ llr_perm, int block_len, int block_stride) {
    // block_len > 0 (ultra_hip_ldpc_decode_blocks): codeword c is row (c / block_len) * block_stride + c % block_len of the
    // LLR array — several equally long runs of rows inside a larger array (one code rate's share of a mode grid) decoded
    // by ONE launch; results stay dense (row c).
    auto llr_row = [&](int c) -> size_t {
        return block_len > 0 ? (size_t)(c / block_len) * (size_t)block_stride + (size_t)(c % block_len) : (size_t)c;
    };
    constexpr int DMAX = ldpc_prof_max(VMAX, VR);
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const LdpcPlan& P = *Pp;
    const int lane = threadIdx.x;
    const int m = P.m, k = P.k, n = P.n;
    float* msg = reinterpret_cast<float*>(lds_raw);                      // [msg_words]
    uint8_t* hard = reinterpret_cast<uint8_t*>(msg + P.msg_words);       // [648] (+8 pad)
    float* llr_s = reinterpret_cast<float*>(hard + 656);                 // [648] staging of the input

    // ---- per-lane slice of the Tanner graph, kept in registers for the whole launch ----
    bool row_ok[RR];                // row (slot) exists
    int raddr[RR][6];               // LDS word address of information edge t of the row, -1 if none
    unsigned rmask[RR];
#pragma unroll
    for (int r = 0; r < RR; ++r) {
        const int row = r * 64 + lane;
        row_ok[r] = P.row_deg[row] != 0;                       // slots may have gaps (linear layout)
        rmask[r] = row_ok[r] ? P.row_mask[row] : 0u;
#pragma unroll
        for (int t = 0; t < 6; ++t) {
            if (t >= ldpc_prof(RMAX, r)) continue;
            const int a = row_ok[r] ? (int)P.row_addr[row * 6 + t] : 0xFFFF;
            raddr[r][t] = (a == 0xFFFF) ? -1 : a;
        }
    }
    int var_j[VR], var_deg[VR], vaddr[VR][DMAX];
    unsigned vmask[VR];
#pragma unroll
    for (int r = 0; r < VR; ++r) {
        const int a = r * 64 + lane;
        const bool on = P.act_deg[a] != 0;
        var_j[r] = on ? P.act_var[a] : 0;
        var_deg[r] = on ? P.act_deg[a] : 0;
        vmask[r] = on ? P.act_mask[a] : 0u;
#pragma unroll
        for (int t = 0; t < DMAX; ++t)
            if (!LINEAR && t < ldpc_prof(VMAX, r)) vaddr[r][t] = on ? P.act_addr[a * kLdpcPlanDmax + t] : 0;
    }
    // Message of edge t of the variable in this lane of round r.  LINEAR: word (r * DMAX + t) * 64 + lane — the
    // load is a lane-linear ds_read_b32 with an immediate offset, the store a ds_write_addtid_b32 (address = M0 +
    // offset + 4 * lane: no address register, 2 instead of 4 LDS cycles).
    const unsigned msg_lds = (unsigned)(size_t)msg;           // LDS byte address of the message array
    auto vload = [&](int r, int t) -> float { return LINEAR ? msg[(r * DMAX + t) * 64 + lane] : msg[vaddr[r][t]]; };
    auto vstore = [&](int r, int t, float x) {
        if constexpr (LINEAR) {
            // s_nop: the wait state the hardware needs between an SALU write of M0 and an add-TID LDS instruction
            // (the compiler's hazard recogniser does not look inside the asm)
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tds_write_addtid_b32 %0"
                         :: "v"(x), "s"(msg_lds + (unsigned)((r * DMAX + t) * 256)) : "m0", "memory");
        } else {
            msg[vaddr[r][t]] = x;
        }
    };
    // parity bit of the row in slot r * 64 + lane: variable k + row_id
    auto parity_var = [&](int r) -> int { return k + (ROW_ID ? (int)P.row_id[r * 64 + lane] : r * 64 + lane); };

    // Work queue: kLdpcQueues interleaved queues (queue q serves codewords q, q + Q, q + 2Q, ...), each
    // with its own counter in its own cache line.  A single counter serialises in L2 at ~11.6 ns per
    // atomic — 3.05 ms per 2^18 codewords, as long as the decoding itself (measured: a batch that
    // converges at iteration 0 took as long as one at 20 iterations).  A wavefront starts on queue
    // blockIdx % Q and moves on to the next one when its queue runs dry.  (Claiming several
    // consecutive codewords per atomic was measured slower: 3.44 -> 3.59 ms at 4 per claim.)
    // The next codeword is claimed and its LLRs are fetched (asynchronously, straight into the LDS
    // staging area, which is idle once the messages are initialised) while the current one is being
    // decoded, so neither the atomic's round trip nor the HBM latency is paid per codeword.
    int queue = (int)(blockIdx.x % kLdpcQueues), dry = 0;
    auto claim = [&]() -> int {                           // next codeword of this wavefront, -1 when all queues are dry
        for (;;) {
            int ticket = 0;
            if (lane == 0) ticket = (int)atomicAdd(work_counter + queue * kLdpcQueueStride, 1u);
            const int c = __builtin_amdgcn_readfirstlane(ticket) * kLdpcQueues + queue;
            if (c < n_cw) return c;
            if (++dry == kLdpcQueues) return -1;
            queue = (queue + 1) % kLdpcQueues;
        }
    };
    // 648 floats -> llr_s, 64 per instruction.  The channel deinterleaver of the production receive
    // path (RxPipeline::deinterleaveCodewords, rx_pipeline.cpp:475-491 -> ChannelInterleaver::deinterleave,
    // ldpc_decoder.cpp:609-617: out[j] = in[(j * step) % 648]) is fused into this copy as the gather
    // index; llr_step == 1 is the identity.  llr_perm (nullable): a general gather table out[j] = in[llr_perm[j]]
    // (ultra_hip_set_deinterleave_table: any 648-entry permutation, e.g. the row-column Interleaver::deinterleave,
    // ldpc_decoder.cpp:454-466,530-540) takes precedence over the step.
    // The table case is a separate loop behind ONE uniform branch: with the choice made per load, every asynchronous
    // copy sat behind a conditional index load and its wait (measured on the R1/4 sweep: 10 ms of 73 per 5.5 M codewords).
    auto fetch = [&](int c) {
        const float* src = llr + llr_row(c) * llr_stride;
        if (llr_perm) {
            for (int j0 = 0; j0 < n; j0 += kLdpcThreads)
                if (j0 + lane < n) __builtin_amdgcn_global_load_lds(src + (unsigned)llr_perm[j0 + lane], llr_s + j0, 4, 0, 0);
        } else {
            for (int j0 = 0; j0 < n; j0 += kLdpcThreads)
                if (j0 + lane < n)
                    __builtin_amdgcn_global_load_lds(src + (unsigned)((j0 + lane) * llr_step) % (unsigned)kLdpcN, llr_s + j0, 4, 0, 0);
        }
    };
    // Tickets are drawn one codeword ahead of their use (see ldpc_totals_kernel.h): the atomic's round trip runs under a
    // decode; a ticket beyond the queue's end (the launch's tail) falls back to the synchronous claim().
    int ticket_v = 0, ticket_queue = queue;
    auto draw = [&]() {
        ticket_queue = queue;
        ticket_v = 0;
        if (lane == 0) ticket_v = (int)atomicAdd(work_counter + queue * kLdpcQueueStride, 1u);
    };
    auto take = [&]() -> int {
        const int c = __builtin_amdgcn_readfirstlane(ticket_v) * kLdpcQueues + ticket_queue;
        if (c < n_cw) return c;
        if (dry >= kLdpcQueues) return -1;
        return claim();
    };
    int cw = claim();
    if (cw >= 0) { fetch(cw); draw(); }
    while (cw >= 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // staged LLRs have landed
        __syncthreads();
        // ---- hard decisions of the raw channel values (decided bits of unchecked variables) ----
        for (int j = lane; j < n; j += kLdpcThreads) {
            const float v = llr_s[j];
            hard[j] = (v < 0) ? 1 : 0;
            if (WANT_TOTAL) llr_total[(size_t)cw * kLdpcN + j] = v;
        }
        __syncthreads();
        float llr_v[VR], llr_p[RR], vpar[RR];
        // Hard decisions: the totals themselves are kept (tot < 0 is evaluated where a decision is needed — the
        // rare exact parity test and the output); a bool that lives across iterations costs a 0/1 VGPR and one more
        // instruction per round to materialise it.
        float tpar[RR], tvar[VR];
#pragma unroll
        for (int r = 0; r < VR; ++r) {
            llr_v[r] = llr_s[var_j[r]];
            tvar[r] = 0.0f;
#pragma unroll
            for (int t = 0; t < DMAX; ++t) {                                   // v2c = llr_in[col]
                if (t >= ldpc_prof(VMAX, r)) continue;
                // The first messages are the raw channel values.  A signalling NaN among them would come out of
                // v_min_f32 quieted instead of being skipped (IEEE mode), so the message copy is canonicalised: a
                // quiet NaN with the same sign — every later message is an arithmetic result and quiet anyway;
                // numbers, zeros, denormals and infinities pass unchanged.
                const float first = __builtin_canonicalizef(llr_v[r]);
                if (t < ldpc_prof(VMIN, r)) vstore(r, t, first);
                else if (t < var_deg[r]) vstore(r, t, first);
            }
        }
#pragma unroll
        for (int r = 0; r < RR; ++r) {
            llr_p[r] = row_ok[r] ? llr_s[parity_var(r)] : 0.0f;
            vpar[r] = __builtin_canonicalizef(llr_p[r]);                       // v2c of the parity bit (quiet, see above)
            tpar[r] = 0.0f;
        }
        __syncthreads();
        const int cw_next = take();                        // llr_s is free from here on
        if (cw_next >= 0) { fetch(cw_next); draw(); }

        int it = 0, ok = 0;
        unsigned F = 1u;                                                       // syndrome filter of iteration it-1
        // exact parity test (checkParity :139-151) on the byte array `hard` and the given totals of the parity bits
        auto rows_hold = [&](const float (&par_total)[RR]) -> bool {
            int bad = 0;
#pragma unroll
            for (int r = 0; r < RR; ++r) {
                if (row_ok[r]) {
                    const int row = r * 64 + lane;
                    int s = (par_total[r] < 0) ? 1 : 0;
                    for (int t = 0; t < 6; ++t) {
                        const unsigned c = P.row_col[row * 6 + t];
                        if (c != 0xFFFFu) s ^= hard[c];
                    }
                    bad |= s;
                }
            }
            return __ballot(bad != 0) == 0ull;
        };
        // A codeword whose channel hard decisions already satisfy every row converges at iteration 0 with exactly those
        // bits (in a satisfied row the product of the other edges' signs is the edge's own sign: every check message
        // pushes its variable further the way it already points, no total changes sign — for any input, a NaN counting
        // as + like `x < 0` does).  So the filter is first evaluated on the channel values; when it vanishes the exact
        // test runs on them (`hard` holds their decisions), and the decode ends before its first iteration instead of
        // after it.  Kept OUT of the iteration loop: reaching the test from iteration 0 inside it cost the loop 20 %.
        if (!WANT_TOTAL && P.max_iterations > 0) {
            unsigned f0 = 0u;
#pragma unroll
            for (int r = 0; r < VR; ++r) f0 ^= (llr_v[r] < 0) ? vmask[r] : 0u;
#pragma unroll
            for (int r = 0; r < RR; ++r) f0 ^= (llr_p[r] < 0) ? rmask[r] : 0u;
            if (wave_xor(f0) == 0u && rows_hold(llr_p)) ok = 1;
        }
        while (!ok) {
            if (it > 0 && F == 0u) {
                // ---- reached once per converged codeword ----
#pragma unroll
                for (int r = 0; r < VR; ++r)
                    if (var_deg[r] > 0) hard[var_j[r]] = (tvar[r] < 0) ? 1 : 0;
                __syncthreads();
                if (rows_hold(tpar)) { ok = 1; --it; break; }
            }
            if (it >= P.max_iterations) break;

            // ---- check step + the row's own parity bit: one lane per row ----
            // The reference clamps every variable-to-check message to +-50 when it is produced
            // (:216-224), except the initial ones (= llr_in).  The clamp keeps the sign (NaN -> +50,
            // and `NaN < 0` is false) and turns |v| into min(|v|, 50), so "min over the other edges of
            // the clamped magnitudes" = min(min over the other edges of the raw magnitudes, 50): the
            // messages are stored unclamped and the two seeds of the minimum chains carry the cap
            // (FLT_MAX in iteration 0, whose inputs are the unclamped channel values).
            unsigned f = 0u;                                               // this lane's share of the syndrome filter
            const float cap = (it == 0) ? kFltMax : 50.0f;
            ldpc_static_for(std::make_integer_sequence<int, RR>{}, [&](auto round) {
                constexpr int r = decltype(round)::value;
                if (row_ok[r]) {
                    constexpr int rmax = ldpc_prof(RMAX, r), rmin = ldpc_prof(RMIN, r);   // information-edge slots of the round
                    float v[7];
#pragma unroll
                    for (int t = 0; t < 6; ++t) {
                        if (t >= rmax) continue;
                        if (t < rmin) v[t] = msg[raddr[r][t]];
                        else v[t] = (raddr[r][t] >= 0) ? msg[raddr[r][t]] : kFltMax;   // missing edge: neutral
                    }
                    v[rmax] = vpar[r];
                    // Signs as lane masks (SGPR pairs): the row parity and each edge's "all others"
                    // sign are scalar xors; the sign is applied with one select between mag and -mag.
                    // Magnitudes only ever appear as |.| source modifiers of the minima (leave_one_out_min).
                    float mn[7];
                    bool ng[7], par = false;
#pragma unroll
                    for (int t = 0; t <= rmax; ++t) {
                        ng[t] = v[t] < 0;
                        par ^= ng[t];
                    }
                    leave_one_out_min<rmax + 1>(v, cap, mn);
#pragma unroll
                    for (int t = 0; t < 6; ++t) {
                        if (t >= rmax) continue;
                        const float mag = mn[t] * 0.75f;
                        const float c = (par != ng[t]) ? -mag : mag;           // sign * min * 0.75f
                        if (t < rmin) msg[raddr[r][t]] = c;
                        else if (raddr[r][t] >= 0) msg[raddr[r][t]] = c;
                    }
                    const float mag6 = mn[rmax] * 0.75f;
                    const float c_last = (par != ng[rmax]) ? -mag6 : mag6;
                    const float total_p = llr_p[r] + c_last;               // parity bit of the row
                    tpar[r] = total_p;
                    f ^= (total_p < 0) ? rmask[r] : 0u;
                    vpar[r] = total_p - c_last;                            // clamp deferred, see `cap`
                    if (WANT_TOTAL) llr_total[(size_t)cw * kLdpcN + parity_var(r)] = total_p;
                }
            });
            __syncthreads();
            // ---- totals + variable step for the information bits that have checks ----
            ldpc_static_for(std::make_integer_sequence<int, VR>{}, [&](auto round) {
                constexpr int r = decltype(round)::value;
                constexpr int vmax = ldpc_prof(VMAX, r), vmin = ldpc_prof(VMIN, r);
                const int d = var_deg[r];
                if (vmin > 0 || d > 0) {                                   // vmin > 0: all 64 lanes hold a variable
                    float c[DMAX];
#pragma unroll
                    for (int t = 0; t < DMAX; ++t) {
                        if (t >= vmax) continue;
                        if (t < vmin) c[t] = vload(r, t);
                        else c[t] = (t < d) ? vload(r, t) : 0.0f;
                    }
                    float tot = llr_v[r];
#pragma unroll
                    for (int t = 0; t < DMAX; ++t) {                       // ascending check order
                        if (t >= vmax) continue;
                        if (t < vmin) tot += c[t];
                        else if (t < d) tot += c[t];
                    }
                    tvar[r] = tot;
                    f ^= (tot < 0) ? vmask[r] : 0u;
#pragma unroll
                    for (int t = 0; t < DMAX; ++t) {                       // clamp deferred to the reader
                        if (t >= vmax) continue;
                        if (t < vmin) vstore(r, t, tot - c[t]);
                        else if (t < d) vstore(r, t, tot - c[t]);
                    }
                    if (WANT_TOTAL) llr_total[(size_t)cw * kLdpcN + var_j[r]] = tot;
                }
            });
            F = wave_xor(f);
            __syncthreads();
            ++it;
        }
        const int iters_out = ok ? it : P.max_iterations;

        // ---- pack the k info bits MSB-first (ldpc_decoder.cpp:238-258) ----
        if (!ok && P.max_iterations > 0) {                                 // on success `hard` was just refreshed
#pragma unroll
            for (int r = 0; r < VR; ++r)
                if (var_deg[r] > 0) hard[var_j[r]] = (tvar[r] < 0) ? 1 : 0;
            __syncthreads();
        }
        uint8_t* ob = bytes + (size_t)cw * P.decoded_bytes;
        for (int b = lane; b < P.decoded_bytes; b += kLdpcThreads) {
            unsigned v = 0;
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int j = 8 * b + t;
                v = (v << 1) | ((j < k) ? (unsigned)hard[j] : 0u);
            }
            ob[b] = (uint8_t)v;
        }
        if (lane == 0) { iters[cw] = iters_out; okv[cw] = (uint8_t)ok; }
        cw = cw_next;
    }
}

// Streams without a usable frame (no sync, or the frame runs past the end of the stream): the reference
// produces no soft bits for them, so their decode results are cleared (ok = 0, iters = 0, bytes = 0).
// entry[f] = first data sample, or 0xffffffff when unusable.
__global__ __launch_bounds__(256) void clear_unusable_kernel(const unsigned* __restrict__ entry, int n_frames,
                                                             uint8_t* __restrict__ bytes, int bytes_stride,
                                                             int32_t* __restrict__ iters, uint8_t* __restrict__ okv) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n_frames || entry[f] != 0xffffffffu) return;
    for (int b = 0; b < bytes_stride; ++b) bytes[(size_t)f * bytes_stride + b] = 0;
    iters[f] = 0;
    okv[f] = 0;
}

// entry[f] = found[f] && data_start[f] + frame_samples <= n_samples ? data_start[f] : 0xffffffff;
// offset[f] = the same with 0 for unusable streams (their frame is demodulated from sample 0 and discarded)
__global__ __launch_bounds__(256) void frame_entry_kernel(const unsigned* __restrict__ found,
                                                          const unsigned* __restrict__ data_start, unsigned frame_samples,
                                                          unsigned n_samples, int n_frames, unsigned* __restrict__ entry,
                                                          unsigned* __restrict__ offset) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n_frames) return;
    const bool ok = found[f] != 0 && data_start[f] + frame_samples <= n_samples;
    entry[f] = ok ? data_start[f] : 0xffffffffu;
    offset[f] = ok ? data_start[f] : 0u;
}

// Monte-Carlo counters: frame OK iff ok && payload bytes equal
// (tools/test_nvis_mode.cpp:104-113).  One lane per frame, wave + block
// reduction, one set of atomics per block.
__global__ __launch_bounds__(256) void count_errors_kernel(
    const uint8_t* __restrict__ bytes, size_t bytes_stride, const int32_t* __restrict__ iters,
    const uint8_t* __restrict__ okv, const uint8_t* __restrict__ payload, int payload_bytes,
    int n_frames, unsigned long long* __restrict__ counters) {
    // blockIdx.y = sweep point: frames [y * n_frames, (y + 1) * n_frames) accumulate into counters[8 * y ..]
    // (ultra_hip_count_errors_points; the single-point entry launches one row)
    __shared__ unsigned long long acc[8];
    if (threadIdx.x < 8) acc[threadIdx.x] = 0;
    __syncthreads();
    const size_t first = (size_t)blockIdx.y * (size_t)n_frames;
    bytes += first * bytes_stride; iters += first; okv += first; payload += first * (size_t)payload_bytes;
    counters += 8 * (size_t)blockIdx.y;
    unsigned long long c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int f = blockIdx.x * blockDim.x + threadIdx.x; f < n_frames; f += gridDim.x * blockDim.x) {
        const uint8_t* d = bytes + (size_t)f * bytes_stride;
        const uint8_t* p = payload + (size_t)f * payload_bytes;
        // four bytes at a time (rows start at any byte: unaligned dword loads, which the memory system splits itself —
        // a quarter of the load instructions of the byte loop, each of which touched 31 cache lines per wavefront)
        unsigned biterr = 0;
        int b = 0;
        for (; b + 4 <= payload_bytes; b += 4) {
            unsigned x, y;
            __builtin_memcpy(&x, d + b, 4);
            __builtin_memcpy(&y, p + b, 4);
            biterr += __popc(x ^ y);
        }
        for (; b < payload_bytes; ++b) biterr += __popc((unsigned)(d[b] ^ p[b]));
        const int ok = okv[f];
        c[0] += 1;
        c[1] += (!ok || biterr) ? 1 : 0;
        c[2] += biterr;
        c[3] += 8ull * payload_bytes;
        c[4] += ok ? 0 : 1;
        c[5] += (unsigned long long)iters[f];
        c[6] += (ok && biterr) ? 1 : 0;
    }
    for (int q = 0; q < 7; ++q) {
        unsigned long long v = c[q];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if ((threadIdx.x & 63) == 0 && v) atomicAdd(&acc[q], v);
    }
    __syncthreads();
    if (threadIdx.x < 7 && acc[threadIdx.x]) atomicAdd(&counters[threadIdx.x], acc[threadIdx.x]);
}

}  // namespace dev
}  // namespace ultra_hip
#endif
