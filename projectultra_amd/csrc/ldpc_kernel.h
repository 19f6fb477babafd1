// ldpc_kernel.h — batched LDPC(648) scaled-min-sum decode for gfx950.
//
// Restates LDPCDecoder::Impl::decodeBP (src/fec/ldpc_decoder.cpp:153-259):
//   init   v2c[e] = llr_in[col[e]]
//   iterate (<= max_iterations, 0-based index `it`):
//     check step   c2v[e] = (prod_{e'!=e} sgn(v2c[e'])) * min_{e'!=e}|v2c[e']| * 0.75f   (:181-202)
//     totals       total[j] = llr_in[j] + sum_e c2v[e], accumulated in ascending check order (:206-213)
//     var step     v2c[e] = clamp(total[col[e]] - c2v[e], -50, 50)                         (:216-224)
//     parity       all rows of H xor to 0 over (total < 0)  -> success, stop             (:227-235)
//   output first k hard bits packed MSB-first, lastIterations = it (or max on failure)
//
// Value-identical reformulation of the check step: the brute-force "all others"
// minimum equals min1 unless edge e holds the minimum, then min2; the sign is
// the row's total sign parity with edge e's own sign removed (`msg < 0`, so
// -0.0 counts as positive, as in the reference).  min/abs/compare and the single
// multiply by 0.75f are exact or correctly rounded, so results are bit-identical.
//
// Mapping: one 64-lane wavefront per codeword, the whole message state
// (one f32 per edge, <= 9.7 KB) and the channel LLRs live in LDS; HBM sees the
// 648 input LLRs once and ceil(k/8)+5 output bytes.  Codewords exit as soon as
// their parity check passes (iteration counts vary 0..50 per codeword).
#ifndef ULTRA_LDPC_KERNEL_H
#define ULTRA_LDPC_KERNEL_H

#include <hip/hip_runtime.h>
#include "device_types.h"

namespace ultra_hip {
namespace dev {

constexpr int kLdpcThreads = 64;

// LDS carve (dynamic, sized per code rate by the host: ldpc_lds_bytes()):
//   msg[edges]  v2c before the check step, c2v after it
//   llr_in[648] channel LLRs, total[648] a-posteriori LLRs, hard[648] hard decisions
struct LdpcShared {
    float* msg;
    float* llr_in;
    float* total;
    uint8_t* hard;
};
__host__ __device__ inline size_t ldpc_lds_bytes(int edges) {
    return (size_t)((edges + 3) & ~3) * sizeof(float) + 2 * kLdpcN * sizeof(float) + ((kLdpcN + 15) & ~15);
}
__device__ __forceinline__ LdpcShared ldpc_carve(unsigned char* base, int edges) {
    LdpcShared sh;
    sh.msg = reinterpret_cast<float*>(base);
    sh.llr_in = sh.msg + ((edges + 3) & ~3);
    sh.total = sh.llr_in + kLdpcN;
    sh.hard = reinterpret_cast<uint8_t*>(sh.total + kLdpcN);
    return sh;
}

__device__ __forceinline__ float clamp50(float v) {
    const float lo = (v < 50.0f) ? v : 50.0f;        // std::min(50.0f, v)
    return (-50.0f < lo) ? lo : -50.0f;              // std::max(-50.0f, .)
}

// Decode one codeword held by this wavefront.  Returns through out params.
__device__ __forceinline__ void ldpc_decode_wave(const LdpcShared& sh, const LdpcConst& L,
                                                 const float* __restrict__ llr, int* out_iters, int* out_ok) {
    const int lane = threadIdx.x;
    const int n = L.n, m = L.m, edges = L.edges;
    for (int j = lane; j < n; j += kLdpcThreads) {
        const float v = llr[j];
        sh.llr_in[j] = v;
        sh.total[j] = v;
    }
    __syncthreads();
    for (int e = lane; e < edges; e += kLdpcThreads) sh.msg[e] = sh.llr_in[L.col[e]];
    __syncthreads();

    int it = 0, ok = 0;
    for (; it < L.max_iterations; ++it) {
        // ---- check step: one lane per check row ----
        for (int i = lane; i < m; i += kLdpcThreads) {
            const int e0 = L.row_ptr[i], e1 = L.row_ptr[i + 1];
            float min1 = 3.402823466e+38f, min2 = 3.402823466e+38f;
            int arg = -1, neg = 0;
            for (int e = e0; e < e1; ++e) {
                const float v = sh.msg[e];
                const float a = fabsf(v);
                neg ^= (v < 0) ? 1 : 0;
                if (a < min1) { min2 = min1; min1 = a; arg = e; }
                else if (a < min2) { min2 = a; }
            }
            for (int e = e0; e < e1; ++e) {
                const float v = sh.msg[e];
                const int s = neg ^ ((v < 0) ? 1 : 0);
                const float mag = (e == arg) ? min2 : min1;
                sh.msg[e] = (s ? -mag : mag) * 0.75f;
            }
        }
        __syncthreads();
        // ---- totals + variable step: one lane per variable ----
        for (int j = lane; j < n; j += kLdpcThreads) {
            const int q0 = L.var_ptr[j], q1 = L.var_ptr[j + 1];
            float t = sh.llr_in[j];
            for (int q = q0; q < q1; ++q) t += sh.msg[L.var_edge[q]];   // ascending check order
            sh.total[j] = t;
            sh.hard[j] = (t < 0) ? 1 : 0;
            for (int q = q0; q < q1; ++q) {
                const int e = L.var_edge[q];
                sh.msg[e] = clamp50(t - sh.msg[e]);
            }
        }
        __syncthreads();
        // ---- parity ----
        int bad = 0;
        for (int i = lane; i < m; i += kLdpcThreads) {
            const int e0 = L.row_ptr[i], e1 = L.row_ptr[i + 1];
            int s = 0;
            for (int e = e0; e < e1; ++e) s ^= sh.hard[L.col[e]];
            bad |= s;
        }
        if (__ballot(bad != 0) == 0ull) { ok = 1; break; }
    }
    *out_iters = it;
    *out_ok = ok;
}

// Kernel: one 64-thread workgroup per codeword (grid-stride over codewords).
//   llr        rows of llr_stride floats, the first 648 of each row are decoded
//   bytes      [n_cw][decoded_bytes], iters [n_cw], ok [n_cw]
//   llr_total  [n_cw][648] or nullptr
__global__ __launch_bounds__(kLdpcThreads) void ldpc_decode_kernel(
    const LdpcConst* __restrict__ Lp, const float* __restrict__ llr, size_t llr_stride, int n_cw,
    uint8_t* __restrict__ bytes, int32_t* __restrict__ iters, uint8_t* __restrict__ okv,
    float* __restrict__ llr_total) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ldpc_lds[];
    const LdpcConst& L = *Lp;
    const LdpcShared sh = ldpc_carve(ldpc_lds, L.edges);
    const int lane = threadIdx.x;
    for (int cw = blockIdx.x; cw < n_cw; cw += gridDim.x) {
        int it, ok;
        ldpc_decode_wave(sh, L, llr + (size_t)cw * llr_stride, &it, &ok);
        // pack the k info bits MSB-first (ldpc_decoder.cpp:238-258)
        uint8_t* ob = bytes + (size_t)cw * L.decoded_bytes;
        for (int b = lane; b < L.decoded_bytes; b += kLdpcThreads) {
            unsigned v = 0;
            for (int t = 0; t < 8; ++t) {
                const int j = 8 * b + t;
                v = (v << 1) | ((j < L.k) ? (sh.total[j] < 0 ? 1u : 0u) : 0u);
            }
            ob[b] = (uint8_t)v;
        }
        if (llr_total)
            for (int j = lane; j < L.n; j += kLdpcThreads) llr_total[(size_t)cw * kLdpcN + j] = sh.total[j];
        if (lane == 0) { iters[cw] = it; okv[cw] = (uint8_t)ok; }
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
