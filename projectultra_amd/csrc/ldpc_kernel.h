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
//   * check step: the brute-force "all others" minimum is min1 unless edge e holds the minimum,
//     then min2; the sign is the row's sign parity with edge e's own sign removed (`v < 0`, so
//     -0.0 counts as positive as in the reference); one multiply by 0.75f.
//   * H = [H_data | I]: parity variable k+i has exactly one edge, the last one of row i.  Its
//     total (llr + c2v), its new v2c and its hard bit are formed by the lane that owns row i,
//     straight from registers.
//   * variables without any check (R3/4: info bits 325..485, R5/6: 217..539) keep
//     total = llr_in forever; they are never touched after the load.
//   * the parity test of iteration it-1 runs at the top of iteration it, from one hard-bit
//     byte per edge that the variable step left next to the messages.
//
// Layout in LDS per codeword: rows padded to 8 slots (max row degree is 7):
//   msg[8*row + pos]  f32   v2c before the check step, c2v after it; pad slots hold +FLT_MAX,
//                           which is neutral for the min (`a < FLT_MAX` false) and the sign
//   hb [8*row + pos]  u8    hard bit of the variable on that edge (info edges only, else 0)
//   hard[648]         u8    hard decision per variable (for the output bytes)
// so a row is two ds_read_b128 + one ds_read_b64.  The Tanner graph lives in REGISTERS: each lane
// keeps the degrees of the rows it owns and, for the variables it owns, their slot lists in
// ascending check order (the order llr_total accumulates in), loaded once per workgroup;
// workgroups are persistent and pull codewords from an atomic counter, because iteration
// counts range from 0 to 50 per codeword.
#ifndef ULTRA_LDPC_KERNEL_H
#define ULTRA_LDPC_KERNEL_H

#include <hip/hip_runtime.h>
#include "device_types.h"

namespace ultra_hip {
namespace dev {

constexpr int kLdpcThreads = 64;
constexpr float kFltMax = 3.402823466e+38f;

// std::max(-50.0f, std::min(50.0f, v)): fminf/fmaxf give the same value for every input, NaN
// included (std::min(50, NaN) = 50 because `NaN < 50` is false; fminf(50, NaN) = 50)
__device__ __forceinline__ float clamp50(float v) { return fmaxf(-50.0f, fminf(50.0f, v)); }

__host__ __device__ inline size_t ldpc_lds_bytes(int m) {
    return (size_t)m * 8 * sizeof(float) + (size_t)m * 8 + 656 + 648 * sizeof(float);
}

// RR = ceil(m / 64) row rounds, VR = ceil(n_active / 64) variable rounds, DMAX = max variable degree
template <int RR, int VR, int DMAX, bool WANT_TOTAL>
__global__ __launch_bounds__(kLdpcThreads) void ldpc_decode_kernel(
    const LdpcPlan* __restrict__ Pp, const float* __restrict__ llr, size_t llr_stride, int n_cw,
    uint8_t* __restrict__ bytes, int32_t* __restrict__ iters, uint8_t* __restrict__ okv,
    float* __restrict__ llr_total, unsigned int* __restrict__ work_counter) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const LdpcPlan& P = *Pp;
    const int lane = threadIdx.x;
    const int m = P.m, k = P.k, n = P.n;
    float* msg = reinterpret_cast<float*>(lds_raw);                      // [m][8]
    uint8_t* hb = reinterpret_cast<uint8_t*>(msg + (size_t)m * 8);       // [m][8]
    uint8_t* hard = hb + (size_t)m * 8;                                  // [648] (+8 pad)
    float* llr_s = reinterpret_cast<float*>(hard + 656);                 // [648] staging of the input

    // ---- per-lane slice of the Tanner graph, kept in registers for the whole launch ----
    int row_deg[RR];
#pragma unroll
    for (int r = 0; r < RR; ++r) { const int row = r * 64 + lane; row_deg[r] = (row < m) ? P.row_deg[row] : 0; }
    int var_j[VR], var_deg[VR];
    int slot[VR][DMAX];          // slot index (8*row + pos) of each edge of the variable, ascending check order
#pragma unroll
    for (int r = 0; r < VR; ++r) {
        const int a = r * 64 + lane;
        const bool on = a < P.n_active;
        var_j[r] = on ? P.act_var[a] : 0;
        var_deg[r] = on ? P.act_deg[a] : 0;
#pragma unroll
        for (int t = 0; t < DMAX; ++t) slot[r][t] = on ? P.act_slot[a * kLdpcPlanDmax + t] : 0;
    }
    for (int i = lane; i < m * 2; i += kLdpcThreads) reinterpret_cast<unsigned int*>(hb)[i] = 0u;   // pad / parity slots stay 0
    __syncthreads();

    for (;;) {
        int cw = 0;
        if (lane == 0) cw = (int)atomicAdd(work_counter, 1u);
        cw = __builtin_amdgcn_readfirstlane(cw);
        if (cw >= n_cw) break;
        const float* in = llr + (size_t)cw * llr_stride;

        // ---- load: 648 LLRs once (coalesced), hard decisions of the raw channel values ----
        for (int j = lane; j < n; j += kLdpcThreads) {
            const float v = in[j];
            llr_s[j] = v;
            hard[j] = (v < 0) ? 1 : 0;
            if (WANT_TOTAL) llr_total[(size_t)cw * kLdpcN + j] = v;
        }
        __syncthreads();
        float llr_v[VR], llr_p[RR];
        int hpar[RR];
#pragma unroll
        for (int r = 0; r < VR; ++r) {
            llr_v[r] = llr_s[var_j[r]];
#pragma unroll
            for (int t = 0; t < DMAX; ++t)
                if (t < var_deg[r]) msg[slot[r][t]] = llr_v[r];               // v2c = llr_in[col]
        }
#pragma unroll
        for (int r = 0; r < RR; ++r) {
            const int row = r * 64 + lane;
            hpar[r] = 0;
            llr_p[r] = 0.0f;
            if (row < m) {
                llr_p[r] = llr_s[k + row];
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    if (t == row_deg[r] - 1) msg[row * 8 + t] = llr_p[r];
                    else if (t >= row_deg[r]) msg[row * 8 + t] = kFltMax;
                }
            }
        }
        __syncthreads();

        int it = 0, ok = 0;
        for (;;) {
            if (it > 0) {
                // ---- parity of the totals left by iteration it-1 (checkParity :139-151) ----
                int bad = 0;
#pragma unroll
                for (int r = 0; r < RR; ++r) {
                    const int row = r * 64 + lane;
                    if (row < m) {
                        const uint2 b = *reinterpret_cast<const uint2*>(hb + row * 8);
                        unsigned x = b.x ^ b.y;
                        x ^= x >> 16;
                        x ^= x >> 8;
                        bad |= (int)((x & 1u) ^ (unsigned)hpar[r]);
                    }
                }
                if (__ballot(bad != 0) == 0ull) { ok = 1; --it; break; }
            }
            if (it >= P.max_iterations) break;

            // ---- check step + the row's own parity variable: one lane per row ----
#pragma unroll
            for (int r = 0; r < RR; ++r) {
                const int row = r * 64 + lane;
                if (row < m) {
                    float4* rowp = reinterpret_cast<float4*>(msg + row * 8);
                    const float4 lo = rowp[0], hi = rowp[1];
                    float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                    // min over "all others": prefix/suffix minima of |v| (min is exact and order-free;
                    // the reference's `abs < min` update ignores NaN exactly like fminf does)
                    float a[8], pre[8], suf[8];
                    unsigned sg[8], par = 0u;
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        a[t] = fabsf(v[t]);
                        sg[t] = (v[t] < 0) ? 0x80000000u : 0u;     // `msg < 0`: -0.0 and NaN count as positive
                        par ^= sg[t];
                    }
                    pre[0] = kFltMax;
#pragma unroll
                    for (int t = 1; t < 8; ++t) pre[t] = fminf(pre[t - 1], a[t - 1]);
                    suf[7] = kFltMax;
#pragma unroll
                    for (int t = 6; t >= 0; --t) suf[t] = fminf(suf[t + 1], a[t + 1]);
                    const int d = row_deg[r];
                    float c_last = 0.0f;
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        const float mag = fminf(pre[t], suf[t]) * 0.75f;
                        const float c = __uint_as_float(__float_as_uint(mag) ^ (par ^ sg[t]));   // sign * min * 0.75f
                        if (t == d - 1) c_last = c;
                        v[t] = (t < d - 1) ? c : kFltMax;
                    }
                    const float total_p = llr_p[r] + c_last;              // parity variable k+row
                    hpar[r] = (total_p < 0) ? 1 : 0;
                    const float v2c_p = clamp50(total_p - c_last);
#pragma unroll
                    for (int t = 0; t < 8; ++t)
                        if (t == d - 1) v[t] = v2c_p;
                    rowp[0] = make_float4(v[0], v[1], v[2], v[3]);
                    rowp[1] = make_float4(v[4], v[5], v[6], v[7]);
                    if (WANT_TOTAL) llr_total[(size_t)cw * kLdpcN + k + row] = total_p;
                }
            }
            __syncthreads();
            // ---- totals + variable step for the information bits that have checks ----
#pragma unroll
            for (int r = 0; r < VR; ++r) {
                const int d = var_deg[r];
                if (d > 0) {
                    float c[DMAX];
#pragma unroll
                    for (int t = 0; t < DMAX; ++t) c[t] = (t < d) ? msg[slot[r][t]] : 0.0f;
                    float tot = llr_v[r];
#pragma unroll
                    for (int t = 0; t < DMAX; ++t)
                        if (t < d) tot += c[t];                            // ascending check order
                    const uint8_t hbit = (tot < 0) ? 1 : 0;
                    hard[var_j[r]] = hbit;
#pragma unroll
                    for (int t = 0; t < DMAX; ++t) {
                        if (t < d) {
                            msg[slot[r][t]] = clamp50(tot - c[t]);
                            hb[slot[r][t]] = hbit;
                        }
                    }
                    if (WANT_TOTAL) llr_total[(size_t)cw * kLdpcN + var_j[r]] = tot;
                }
            }
            __syncthreads();
            ++it;
        }
        const int iters_out = ok ? it : P.max_iterations;

        // ---- pack the k info bits MSB-first (ldpc_decoder.cpp:238-258) ----
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
        __syncthreads();
    }
}

// Monte-Carlo counters: frame OK iff ok && payload bytes equal
// (tools/test_nvis_mode.cpp:104-113).  One lane per frame, wave + block
// reduction, one set of atomics per block.
__global__ __launch_bounds__(256) void count_errors_kernel(
    const uint8_t* __restrict__ bytes, size_t bytes_stride, const int32_t* __restrict__ iters,
    const uint8_t* __restrict__ okv, const uint8_t* __restrict__ payload, int payload_bytes,
    int n_frames, unsigned long long* __restrict__ counters) {
    __shared__ unsigned long long acc[8];
    if (threadIdx.x < 8) acc[threadIdx.x] = 0;
    __syncthreads();
    unsigned long long c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int f = blockIdx.x * blockDim.x + threadIdx.x; f < n_frames; f += gridDim.x * blockDim.x) {
        const uint8_t* d = bytes + (size_t)f * bytes_stride;
        const uint8_t* p = payload + (size_t)f * payload_bytes;
        unsigned biterr = 0;
        for (int b = 0; b < payload_bytes; ++b) biterr += __popc((unsigned)(d[b] ^ p[b]));
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
