// ldpc_screen_kernel.h — the decoder's SCREEN: codewords whose channel values already satisfy every parity equation, found and
// finished at memory speed, so that the iterating kernel (ldpc_totals_kernel.h) only sees the codewords that need it.
//
// What it decides is the reference's own result, not a shortcut of it.  LDPCDecoder::Impl::decodeBP
// (src/fec/ldpc_decoder.cpp:153-259) stops at iteration index 0 iff the totals after ONE iteration satisfy checkParity
// (:139-151).  If the hard decisions of the channel values themselves (`llr < 0`, :227-230: -0.0 and NaN count as 0) satisfy
// every row, then in each row the product of the OTHER edges' signs equals the edge's own sign, every check message pushes
// its variable the way it already points, no total changes sign, and the reference returns exactly those bits with
// lastIterations() == 0 and lastDecodeSuccess() — the argument ldpc_totals_kernel.h already uses for its iteration-0 verdict
// (any input: the sign rule is the reference's `msg < 0`).  The totals kernel reaches that verdict after staging the row in
// LDS, eight rounds of gathers and a chain of dependent loads for the output — 20,000 cycles per codeword at three to five
// wavefronts per SIMD (profiles/r05_ldpc_stalls_r14.txt), 0.36 ms per 2^17 codewords that need no iteration at all, where
// their 340 MB take 0.06 ms to read.  Good channels are the common case of a modem, and 30 of the 42 points of BASELINE
// configs[3]'s sweep are of that kind.
//
// Three launches in front of the totals kernel, all on the context's stream, nothing read back by the host:
//   ldpc_screen_kernel<RR, true>    a SAMPLE of <= 2,048 codewords spread over the launch: how many are clean -> ctl[1]
//   ldpc_screen_kernel<RR, false>   returns at once unless the sample says the pass pays (ctl[1] >= gate); otherwise every
//                                   codeword: 11 coalesced loads -> 11 ballots = the row's 648 hard bits in LDS (21 words, one
//                                   bank each: every gather below is conflict-free) -> each lane xors the bits of its rows' edges
//                                   (positions in the row AS IT LIES IN MEMORY: the fused channel deinterleaver is part of the
//                                   table, ldpc_screen_prepare_kernel) -> clean: bytes / iters = 0 / ok = 1 written here;
//                                   dirty: appended to the work list (one atomic per 64-codeword chunk, order kept inside it)
//   ldpc_totals_kernel              with the gate on, decodes work_list[0 .. ctl[2]) instead of 0 .. n_cw
// A clean codeword costs its bytes once; a launch far from convergence (the headline batch) costs the sample and two empty
// launches, ~10 us.  Results are bit-identical with and without the screen (tests/test_gpu_ldpc_screen.py; ULTRA_HIP_LDPC_SCREEN=0
// switches it off, =2 forces the pass for every launch whatever the sample says).
#ifndef ULTRA_LDPC_SCREEN_KERNEL_H
#define ULTRA_LDPC_SCREEN_KERNEL_H

#include <hip/hip_runtime.h>
#include "device_types.h"
#include "ldpc_kernel.h"

namespace ultra_hip {
namespace dev {

__host__ __device__ constexpr int tprof_planes_before(unsigned long long p, int r) {     // edge slots of the rounds before r
    int s = 0;
    for (int i = 0; i < r; ++i) s += ldpc_prof(p, i);
    return s;
}

constexpr int kScreenWaves = 4, kScreenThreads = 64 * kScreenWaves;
#ifndef UH_SCREEN_CHUNK
#define UH_SCREEN_CHUNK 64
#endif
constexpr int kScreenChunk = UH_SCREEN_CHUNK;   // codewords per workgroup of the full pass (<= 64: one ballot appends them), the next one's loads in flight
constexpr int kScreenSampleMax = 2048;       // codewords of the sample
constexpr int kScreenLoads = (kLdpcN + 63) / 64;     // 11 wave-wide loads per row
// words of the launch's counter block (ultra_hip_ctx::d_work; word 0 is queue 0's head, zeroed with the queues before every launch)
constexpr int kScreenCtlSample = 1, kScreenCtlDirty = 2;

// The table of positions under the context's current deinterleaver setting (one workgroup; launched when the setting changed):
// row_pos[round][edge][lane] for the rows in their slots, out_pos[j] for the output bits.
__global__ __launch_bounds__(512) void ldpc_screen_prepare_kernel(const uint16_t* __restrict__ row_var, int m, int k, int llr_step,
                                                                  const uint16_t* __restrict__ llr_perm, LdpcScreenPos* __restrict__ out) {
    auto src_index = [&](int j) -> unsigned { return llr_perm ? (unsigned)llr_perm[j] : (unsigned)(j * llr_step) % (unsigned)kLdpcN; };
    for (int s = threadIdx.x; s < kTPlanRowRounds * 64; s += blockDim.x) {
        const int r = s / 64, lane = s % 64;
        for (int t = 0; t < kScreenEdges; ++t) {
            const unsigned v = (s < m) ? (unsigned)row_var[s * kScreenEdges + t] : 0xFFFFu;
            out->row_pos[(r * kScreenEdges + t) * 64 + lane] = (uint16_t)((v != 0xFFFFu) ? src_index((int)v) : (unsigned)kScreenZeroBit);
        }
    }
    for (int j = threadIdx.x; j < kScreenOutPos; j += blockDim.x)
        out->out_pos[j] = (uint16_t)((j < k) ? src_index(j) : (unsigned)kScreenZeroBit);
}

// RR row rounds; EPROF: edges per row round, four bits each (ldpc_prof) — the rows sit in the table by degree, highest first
// (ultra_hip_create), so round r gathers E_r = the degree of its first row: 41 edge slots per lane instead of 56 for R1/4,
// 30 instead of 42 for R1/3 and R1/2; a shorter row of the round gathers the always-zero bit.
template <int RR, unsigned long long EPROF, bool SAMPLE>
__global__ __launch_bounds__(kScreenThreads) void ldpc_screen_kernel(
    const LdpcScreenPos* __restrict__ Sp, const float* __restrict__ llr, size_t llr_stride, int n_cw, int block_len, int block_stride,
    int decoded_bytes, uint8_t* __restrict__ bytes, int32_t* __restrict__ iters, uint8_t* __restrict__ okv,
    unsigned* __restrict__ ctl, unsigned* __restrict__ list, unsigned gate, int sample_n, int sample_stride) {
    static_assert(RR >= 1 && RR <= kTPlanRowRounds, "row rounds of the six codes");
    __shared__ __attribute__((aligned(16))) unsigned bits[kScreenWaves][32];   // a wavefront's row as 648 hard bits (+ zeros up to bit 1023)
    __shared__ unsigned flags[kScreenChunk];                                   // full pass: 1 = needs the iterating kernel
    __shared__ unsigned n_clean;                                               // sample: clean codewords of this workgroup
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (!SAMPLE && ctl[kScreenCtlSample] < gate) return;                       // the sample says the pass does not pay (uniform)
    auto llr_row = [&](int c) -> size_t {
        return block_len > 0 ? (size_t)(c / block_len) * (size_t)block_stride + (size_t)(c % block_len) : (size_t)c;
    };

    // per-lane gather words of the rows in slots lane, lane + 64, ...: (byte offset of the word inside `bits`) << 5 | bit — 15
    // bits, two edges per register (the pass wants wavefronts in flight, not registers)
    constexpr int NE = tprof_planes_before(EPROF, RR), NQ = (NE + 1) / 2;
    unsigned qq[NQ];
#pragma unroll
    for (int e = 0; e < NQ; ++e) qq[e] = 0u;
    ldpc_static_for(std::make_integer_sequence<int, RR>{}, [&](auto round) {
        constexpr int r = decltype(round)::value;
#pragma unroll
        for (int t = 0; t < ldpc_prof(EPROF, r); ++t) {
            const int idx = tprof_planes_before(EPROF, r) + t;
            const unsigned p = (unsigned)Sp->row_pos[(r * kScreenEdges + t) * 64 + lane];
            qq[idx >> 1] |= ((((unsigned)wave * 128u + (p >> 5) * 4u) << 5) | (p & 31u)) << (16 * (idx & 1));
        }
    });
    // positions of the eight bits of output byte `lane` (16 bytes of the table per lane)
    uint4 op = make_uint4(0, 0, 0, 0);
    if (!SAMPLE) op = *reinterpret_cast<const uint4*>(Sp->out_pos + 8 * lane);
    if (threadIdx.x < kScreenChunk) flags[threadIdx.x] = 0u;
    if (threadIdx.x == 0) n_clean = 0u;
    if (lane < 32) bits[wave][lane] = 0u;
    __syncthreads();

    const unsigned char* bits_bytes = reinterpret_cast<const unsigned char*>(&bits[0][0]);
    auto word_at = [&](unsigned byte_off) -> unsigned { return *reinterpret_cast<const unsigned*>(bits_bytes + byte_off); };
    auto fetch = [&](int c, float (&x)[kScreenLoads]) {
        const float* s = llr + llr_row(c) * llr_stride;
#pragma unroll
        for (int i = 0; i < kScreenLoads; ++i) x[i] = (i * 64 + lane < kLdpcN) ? s[i * 64 + lane] : 0.0f;
    };
    // hard bits of a row -> LDS -> every lane's parity equations; true = at least one row fails
    auto is_dirty = [&](const float (&x)[kScreenLoads]) -> bool {
        unsigned long long mine = 0ull;
#pragma unroll
        for (int i = 0; i < kScreenLoads; ++i) {
            const unsigned long long mask = __ballot(x[i] < 0.0f);             // :227-230 (lanes beyond the row hold +0: bit 0)
            if (lane == i) mine = mask;
        }
        if (lane < kScreenLoads) reinterpret_cast<unsigned long long*>(&bits[wave][0])[lane] = mine;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                     // one wavefront writes, the same one reads: in order
        // (the packed words are re-read as opaque values: hoisted out of the codeword loop, their 56 unpacked addresses and
        // shifts would cost the registers the packing saved)
#pragma unroll
        for (int e = 0; e < NQ; ++e) asm volatile("" : "+v"(qq[e]));
        unsigned bad = 0u;
        ldpc_static_for(std::make_integer_sequence<int, RR>{}, [&](auto round) {
            constexpr int r = decltype(round)::value;
            unsigned acc = 0u;
#pragma unroll
            for (int t = 0; t < ldpc_prof(EPROF, r); ++t) {
                const int idx = tprof_planes_before(EPROF, r) + t;
                const unsigned q = (idx & 1) ? (qq[idx >> 1] >> 16) : (qq[idx >> 1] & 0xFFFFu);
                acc ^= word_at(q >> 5) >> (q & 31u);
            }
            bad |= acc;
        });
        return __ballot((bad & 1u) != 0u) != 0ull;
    };

    if (SAMPLE) {
        const int s = (int)blockIdx.x * kScreenWaves + wave;
        if (s < sample_n) {
            float x[kScreenLoads];
            fetch(s * sample_stride, x);
            if (!is_dirty(x) && lane == 0) atomicAdd(&n_clean, 1u);
        }
        __syncthreads();
        if (threadIdx.x == 0 && n_clean != 0u) atomicAdd(ctl + kScreenCtlSample, n_clean);
        return;
    }

    const int chunk0 = (int)blockIdx.x * kScreenChunk;
    const int c_end = min(n_cw, chunk0 + kScreenChunk);
    auto finish = [&](int c, const float (&x)[kScreenLoads]) {
        if (is_dirty(x)) {
            if (lane == 0) flags[c - chunk0] = 1u;
        } else {
            // the k information bits MSB-first (:238-258): bit j of the output is the hard bit at out_pos[j]
            uint8_t* ob = bytes + (size_t)c * decoded_bytes;
#pragma nounroll
            for (int b = lane; b < decoded_bytes; b += 64) {                   // a second trip only for R5/6's 68 bytes
                uint4 o = op;
                if (b >= 64) o = *reinterpret_cast<const uint4*>(Sp->out_pos + 8 * b);
                const unsigned pw[4] = {o.x, o.y, o.z, o.w};
                unsigned v = 0u;
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const unsigned p = (pw[t >> 1] >> (16 * (t & 1))) & 0xFFFFu;
                    v = (v << 1) | ((word_at((unsigned)wave * 128u + (p >> 5) * 4u) >> (p & 31u)) & 1u);
                }
                ob[b] = (uint8_t)v;
            }
            if (lane == 0) { iters[c] = 0; okv[c] = 1; }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                     // the reads of this row before the next row's bits
    };
    // two register sets in turn: the next codeword's loads are in flight while this one's equations are evaluated
    float xa[kScreenLoads], xb[kScreenLoads];
    int c = chunk0 + wave;
    if (c < c_end) {
        fetch(c, xa);
#pragma nounroll
        for (;;) {
            const bool has_b = c + kScreenWaves < c_end;
            if (has_b) fetch(c + kScreenWaves, xb);
            finish(c, xa);
            if (!has_b) break;
            c += kScreenWaves;
            const bool has_a = c + kScreenWaves < c_end;
            if (has_a) fetch(c + kScreenWaves, xa);
            finish(c, xb);
            if (!has_a) break;
            c += kScreenWaves;
        }
    }
    __syncthreads();
    if (wave == 0) {                                                           // append the chunk's dirty codewords, order kept
        const bool f = lane < kScreenChunk && flags[lane] != 0u;
        const unsigned long long mask = __ballot(f);
        const unsigned n = (unsigned)__popcll(mask);
        if (n != 0u) {                                                         // uniform
            unsigned base = 0u;
            if (lane == 0) base = atomicAdd(ctl + kScreenCtlDirty, n);
            base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
            if (f) list[base + (unsigned)__popcll(mask & ((1ull << lane) - 1ull))] = (unsigned)(chunk0 + lane);
        }
    }
}

}  // namespace dev
}  // namespace ultra_hip
#endif
