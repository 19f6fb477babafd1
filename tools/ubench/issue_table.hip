// issue_table.hip — per-opcode issue cost on gfx950, measured in SHADER CYCLES (s_memtime), for the
// instruction kinds the receive-path kernels are made of.  Evidence for the compute-side roofline
// (profiles/r02_issue_table.txt):
//
//   VALU kinds: W wavefronts per SIMD (1, 2, 4, 5, 6, 8) each run a stream of independent instructions of one
//   kind; cost = cycles the SIMD needs per wave-instruction = (elapsed cycles of a wavefront) / (its
//   instructions x W).  Every CU of the chip runs the same thing (clocks under load).
//   LDS kinds: cost per wave-instruction on the CU's single LDS pipeline = elapsed / (instructions x waves per CU).
//
//   hipcc --offload-arch=gfx950 -O3 -o issue_table tools/ubench/issue_table.hip && ./issue_table
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

enum Kind {
    K_ADD_F32, K_MUL_F32, K_FMA_F32, K_MIN3_ABS, K_MIN2_ABS, K_CMP_SGPR, K_CNDMASK_NEG, K_CMP_CNDMASK, K_MOV, K_ADD_U32,
    K_XOR, K_PK_ADD, K_PK_MUL, K_FMA_F64, K_MUL_F64, K_ADD_F64, K_CVT_F64_F32, K_CVT_F32_F64, K_RCP_F32, K_XOR_DPP,
    K_READLANE, K_SALU_XOR64, K_MIX_LDPC_ROW, K_BFI, K_AND_OR, K_MIN3_PLAIN, K_MIN2_E32, K_AND, K_CMP_GT_I32, K_CMP_VCC, K_CNDMASK_VCC,
    K_SDWA_ADD, K_LSHL_OR, K_SUB_F32, K_CVT_I32_F64, K_CVT_F64_I32, K_TRUNC_F64, K_FLOOR_F64,
    K_DS_READ_B32, K_DS_READ_B64, K_DS_READ2_B32, K_DS_READ_B128, K_DS_WRITE_B32, K_DS_WRITE_ADDTID, K_DS_WRITE_B64,
    K_DS_MIX_LDPC, K_N
};
static const char* kNames[K_N] = {
    "v_add_f32", "v_mul_f32", "v_fma_f32", "v_min3_f32 |x|,|y|,|z|", "v_min_f32_e64 |x|,|y|", "v_cmp_lt_f32 -> sgpr pair",
    "v_cndmask_b32 x,-x,sgpr", "v_cmp + v_cndmask (pair)", "v_mov_b32", "v_add_u32", "v_xor_b32", "v_pk_add_f32", "v_pk_mul_f32",
    "v_fma_f64", "v_mul_f64", "v_add_f64", "v_cvt_f64_f32", "v_cvt_f32_f64", "v_rcp_f32", "v_xor_b32 dpp quad_perm",
    "v_readlane_b32", "s_xor_b64", "ldpc row mix (7cmp 12min 7mul 7cnd)", "v_bfi_b32", "v_and_or_b32", "v_min3_f32 (no modifiers)",
    "v_min_f32_e32", "v_and_b32", "v_cmp_gt_i32 -> sgpr pair", "v_cmp_lt_f32 -> vcc (e32)", "v_cndmask_b32 (vcc, e32)",
    "v_add_u32_sdwa WORD_1", "v_lshl_or_b32", "v_sub_f32", "v_cvt_i32_f64", "v_cvt_f64_i32", "v_trunc_f64", "v_floor_f64",
    "ds_read_b32", "ds_read_b64", "ds_read2_b32", "ds_read_b128", "ds_write_b32", "ds_write_addtid_b32", "ds_write_b64",
    "ldpc lds mix (2 rd,1 wr,1 addtid)"};
static const int kInstrPerBlock[K_N] = {8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 33, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8,
                                        8, 8, 8, 8, 8, 8, 8, 8};

template <int KIND>
__global__ __launch_bounds__(64) void k(unsigned long long* cyc, float* sink, int iters, unsigned* arrive) {
    __shared__ float lds[1024];           // 4 KB: 32 wavefronts per CU stay resident
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (float)i;
    __syncthreads();
    float a0 = threadIdx.x + 1.0f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b0 = 0.5f, b1 = 0.25f;
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3;
    unsigned u0 = threadIdx.x, u1 = u0 * 3 + 1, u2 = u0 * 5 + 2, u3 = u0 * 7 + 3;
    const unsigned addr = (unsigned)(size_t)lds + threadIdx.x * 4;       // lane-linear, conflict-free
    const unsigned addr8 = (unsigned)(size_t)lds + threadIdx.x * 8;
    const unsigned addr16 = (unsigned)(size_t)lds + threadIdx.x * 16;
    const unsigned ldsbase = (unsigned)(size_t)lds;
    unsigned long long s0 = 0, s1 = 0;
    // Start line: workgroups are dispatched at a finite rate (a few thousand single-wavefront workgroups take longer to
    // launch than a short measurement runs), so every wavefront waits until the whole grid is resident — bounded (2 ms),
    // so a grid that cannot be resident at once still terminates; the residency column shows whether it was.
    if (arrive && threadIdx.x == 0) {
        atomicAdd(arrive, 1u);
        const unsigned long long w0 = __builtin_amdgcn_s_memrealtime();
        while (__hip_atomic_load(arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x &&
               __builtin_amdgcn_s_memrealtime() - w0 < 200000ull)
            __builtin_amdgcn_s_sleep(8);
        if (__hip_atomic_load(arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x) atomicAdd(arrive + 1, 1u);   // gave up
    }
    __syncthreads();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();      // 100 MHz
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();          // shader clock
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if constexpr (KIND == K_ADD_F32)
                asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                             "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0));
            else if constexpr (KIND == K_MUL_F32)
                asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                             "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0));
            else if constexpr (KIND == K_FMA_F32)
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1));
            else if constexpr (KIND == K_MIN3_ABS)
                asm volatile("v_min3_f32 %0, |%0|, |%1|, |%2|\n v_min3_f32 %1, |%1|, |%2|, |%3|\n v_min3_f32 %2, |%2|, |%3|, |%4|\n"
                             "v_min3_f32 %3, |%3|, |%4|, |%5|\n v_min3_f32 %4, |%4|, |%5|, |%6|\n v_min3_f32 %5, |%5|, |%6|, |%7|\n"
                             "v_min3_f32 %6, |%6|, |%7|, |%0|\n v_min3_f32 %7, |%7|, |%0|, |%1|"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            else if constexpr (KIND == K_MIN2_ABS)
                asm volatile("v_min_f32_e64 %0, |%0|, |%1|\n v_min_f32_e64 %1, |%1|, |%2|\n v_min_f32_e64 %2, |%2|, |%3|\n"
                             "v_min_f32_e64 %3, |%3|, |%4|\n v_min_f32_e64 %4, |%4|, |%5|\n v_min_f32_e64 %5, |%5|, |%6|\n"
                             "v_min_f32_e64 %6, |%6|, |%7|\n v_min_f32_e64 %7, |%7|, |%0|"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            else if constexpr (KIND == K_CMP_SGPR)
                asm volatile("v_cmp_lt_f32_e64 %0, %2, %3\n v_cmp_lt_f32_e64 %1, %3, %4\n v_cmp_lt_f32_e64 %0, %4, %5\n v_cmp_lt_f32_e64 %1, %5, %6\n"
                             "v_cmp_lt_f32_e64 %0, %6, %7\n v_cmp_lt_f32_e64 %1, %7, %8\n v_cmp_lt_f32_e64 %0, %8, %9\n v_cmp_lt_f32_e64 %1, %9, %2"
                             : "+s"(s0), "+s"(s1) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7));
            else if constexpr (KIND == K_CNDMASK_NEG)
                asm volatile("v_cndmask_b32_e64 %0, %0, -%0, %8\n v_cndmask_b32_e64 %1, %1, -%1, %8\n v_cndmask_b32_e64 %2, %2, -%2, %8\n"
                             "v_cndmask_b32_e64 %3, %3, -%3, %8\n v_cndmask_b32_e64 %4, %4, -%4, %8\n v_cndmask_b32_e64 %5, %5, -%5, %8\n"
                             "v_cndmask_b32_e64 %6, %6, -%6, %8\n v_cndmask_b32_e64 %7, %7, -%7, %8"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(s0));
            else if constexpr (KIND == K_CMP_CNDMASK)
                asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc\n v_cmp_lt_f32 vcc, %1, %2\n v_cndmask_b32 %3, %3, %0, vcc\n"
                             "v_cmp_lt_f32 vcc, %2, %3\n v_cndmask_b32 %0, %0, %1, vcc\n v_cmp_lt_f32 vcc, %3, %0\n v_cndmask_b32 %1, %1, %2, vcc"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : : "vcc");
            else if constexpr (KIND == K_MOV)
                asm volatile("v_mov_b32 %0, %4\n v_mov_b32 %1, %5\n v_mov_b32 %2, %6\n v_mov_b32 %3, %7\n"
                             "v_mov_b32 %4, %0\n v_mov_b32 %5, %1\n v_mov_b32 %6, %2\n v_mov_b32 %7, %3"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            else if constexpr (KIND == K_ADD_U32)
                asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %2\n v_add_u32 %2, %2, %3\n v_add_u32 %3, %3, %0\n"
                             "v_add_u32 %0, %0, %2\n v_add_u32 %1, %1, %3\n v_add_u32 %2, %2, %0\n v_add_u32 %3, %3, %1"
                             : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));
            else if constexpr (KIND == K_XOR)
                asm volatile("v_xor_b32 %0, %0, %1\n v_xor_b32 %1, %1, %2\n v_xor_b32 %2, %2, %3\n v_xor_b32 %3, %3, %0\n"
                             "v_xor_b32 %0, %0, %2\n v_xor_b32 %1, %1, %3\n v_xor_b32 %2, %2, %0\n v_xor_b32 %3, %3, %1"
                             : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));
            else if constexpr (KIND == K_PK_ADD)
                asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                             "v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4"
                             : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(d0));
            else if constexpr (KIND == K_PK_MUL)
                asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                             "v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4"
                             : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(d0));
            else if constexpr (KIND == K_FMA_F64)
                asm volatile("v_fma_f64 %0, %0, %0, %0\n v_fma_f64 %1, %1, %1, %1\n v_fma_f64 %2, %2, %2, %2\n v_fma_f64 %3, %3, %3, %3\n"
                             "v_fma_f64 %0, %0, %0, %0\n v_fma_f64 %1, %1, %1, %1\n v_fma_f64 %2, %2, %2, %2\n v_fma_f64 %3, %3, %3, %3"
                             : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));
            else if constexpr (KIND == K_MUL_F64)
                asm volatile("v_mul_f64 %0, %0, %1\n v_mul_f64 %1, %1, %2\n v_mul_f64 %2, %2, %3\n v_mul_f64 %3, %3, %0\n"
                             "v_mul_f64 %0, %0, %1\n v_mul_f64 %1, %1, %2\n v_mul_f64 %2, %2, %3\n v_mul_f64 %3, %3, %0"
                             : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));
            else if constexpr (KIND == K_ADD_F64)
                asm volatile("v_add_f64 %0, %0, %1\n v_add_f64 %1, %1, %2\n v_add_f64 %2, %2, %3\n v_add_f64 %3, %3, %0\n"
                             "v_add_f64 %0, %0, %1\n v_add_f64 %1, %1, %2\n v_add_f64 %2, %2, %3\n v_add_f64 %3, %3, %0"
                             : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));
            else if constexpr (KIND == K_CVT_F64_F32)
                asm volatile("v_cvt_f64_f32 %0, %4\n v_cvt_f64_f32 %1, %5\n v_cvt_f64_f32 %2, %6\n v_cvt_f64_f32 %3, %7\n"
                             "v_cvt_f64_f32 %0, %5\n v_cvt_f64_f32 %1, %6\n v_cvt_f64_f32 %2, %7\n v_cvt_f64_f32 %3, %4"
                             : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
            else if constexpr (KIND == K_CVT_F32_F64)
                asm volatile("v_cvt_f32_f64 %0, %4\n v_cvt_f32_f64 %1, %5\n v_cvt_f32_f64 %2, %6\n v_cvt_f32_f64 %3, %7\n"
                             "v_cvt_f32_f64 %0, %5\n v_cvt_f32_f64 %1, %6\n v_cvt_f32_f64 %2, %7\n v_cvt_f32_f64 %3, %4"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(d0), "v"(d1), "v"(d2), "v"(d3));
            else if constexpr (KIND == K_RCP_F32)
                asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
                             "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            else if constexpr (KIND == K_XOR_DPP)
                asm volatile("v_xor_b32_dpp %0, %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                             "v_xor_b32_dpp %1, %2, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                             "v_xor_b32_dpp %2, %3, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                             "v_xor_b32_dpp %3, %0, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                             "v_xor_b32_dpp %0, %2, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                             "v_xor_b32_dpp %1, %3, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                             "v_xor_b32_dpp %2, %0, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                             "v_xor_b32_dpp %3, %1, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
                             : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));
            else if constexpr (KIND == K_READLANE) {
                unsigned r0, r1, r2, r3;
                asm volatile("v_readlane_b32 %0, %4, 0\n v_readlane_b32 %1, %5, 16\n v_readlane_b32 %2, %6, 32\n v_readlane_b32 %3, %7, 48\n"
                             "v_readlane_b32 %0, %5, 1\n v_readlane_b32 %1, %6, 17\n v_readlane_b32 %2, %7, 33\n v_readlane_b32 %3, %4, 49"
                             : "=s"(r0), "=s"(r1), "=s"(r2), "=s"(r3) : "v"(u0), "v"(u1), "v"(u2), "v"(u3));
                s0 += r0 ^ r1 ^ r2 ^ r3;
            } else if constexpr (KIND == K_SALU_XOR64)
                asm volatile("s_xor_b64 %0, %0, %1\n s_xor_b64 %1, %1, %0\n s_xor_b64 %0, %0, %1\n s_xor_b64 %1, %1, %0\n"
                             "s_xor_b64 %0, %0, %1\n s_xor_b64 %1, %1, %0\n s_xor_b64 %0, %0, %1\n s_xor_b64 %1, %1, %0"
                             : "+s"(s0), "+s"(s1) : : "scc");
            else if constexpr (KIND == K_MIX_LDPC_ROW) {
                // the VALU mix of one row round of the decoder's check step: 7 compares into lane masks (+ scalar xors),
                // the 12-operation leave-one-out minimum network, 7 multiplies by 0.75, 7 sign selects
                unsigned long long m0, m1, m2, m3, m4, m5, m6, par;
                float l2, l4, l6, r4, r3, r2, n0, n1, n2, n3, n4, n5;
                asm volatile(
                    "v_cmp_lt_f32_e64 %0, %20, 0\n v_cmp_lt_f32_e64 %1, %21, 0\n v_cmp_lt_f32_e64 %2, %22, 0\n v_cmp_lt_f32_e64 %3, %23, 0\n"
                    "v_cmp_lt_f32_e64 %4, %24, 0\n v_cmp_lt_f32_e64 %5, %25, 0\n v_cmp_lt_f32_e64 %6, %26, 0\n"
                    "s_xor_b64 %7, %0, %1\n s_xor_b64 %7, %7, %2\n s_xor_b64 %7, %7, %3\n s_xor_b64 %7, %7, %4\n s_xor_b64 %7, %7, %5\n s_xor_b64 %7, %7, %6\n"
                    "v_min3_f32 %8, |%20|, |%21|, |%27|\n v_min3_f32 %9, |%8|, |%22|, |%23|\n v_min3_f32 %10, |%9|, |%24|, |%25|\n"
                    "v_min3_f32 %11, |%25|, |%26|, |%27|\n v_min_f32_e64 %12, |%11|, |%24|\n v_min3_f32 %13, |%11|, |%24|, |%23|\n"
                    "v_min3_f32 %14, |%13|, |%22|, |%21|\n v_min3_f32 %15, |%20|, |%13|, |%22|\n v_min_f32_e64 %16, |%8|, |%13|\n"
                    "v_min3_f32 %17, |%8|, |%22|, |%12|\n v_min_f32_e64 %18, |%9|, |%11|\n v_min3_f32 %19, |%9|, |%24|, |%26|\n"
                    "v_mul_f32 %14, 0x3f400000, %14\n v_mul_f32 %15, 0x3f400000, %15\n v_mul_f32 %16, 0x3f400000, %16\n v_mul_f32 %17, 0x3f400000, %17\n"
                    "v_mul_f32 %18, 0x3f400000, %18\n v_mul_f32 %19, 0x3f400000, %19\n v_mul_f32 %10, 0x3f400000, %10\n"
                    "s_xor_b64 %0, %0, %7\n s_xor_b64 %1, %1, %7\n s_xor_b64 %2, %2, %7\n s_xor_b64 %3, %3, %7\n s_xor_b64 %4, %4, %7\n s_xor_b64 %5, %5, %7\n s_xor_b64 %6, %6, %7\n"
                    "v_cndmask_b32_e64 %20, %14, -%14, %0\n v_cndmask_b32_e64 %21, %15, -%15, %1\n v_cndmask_b32_e64 %22, %16, -%16, %2\n"
                    "v_cndmask_b32_e64 %23, %17, -%17, %3\n v_cndmask_b32_e64 %24, %18, -%18, %4\n v_cndmask_b32_e64 %25, %19, -%19, %5\n"
                    "v_cndmask_b32_e64 %26, %10, -%10, %6"
                    : "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3), "=&s"(m4), "=&s"(m5), "=&s"(m6), "=&s"(par), "=&v"(l2), "=&v"(l4),
                      "=&v"(l6), "=&v"(r4), "=&v"(r3), "=&v"(r2), "=&v"(n0), "=&v"(n1), "=&v"(n2), "=&v"(n3), "=&v"(n4), "=&v"(n5),
                      "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6)
                    : "v"(a7) : "scc");
            } else if constexpr (KIND == K_BFI)
                asm volatile("v_bfi_b32 %0, %4, %0, %1\n v_bfi_b32 %1, %4, %1, %2\n v_bfi_b32 %2, %4, %2, %3\n v_bfi_b32 %3, %4, %3, %0\n"
                             "v_bfi_b32 %0, %4, %0, %2\n v_bfi_b32 %1, %4, %1, %3\n v_bfi_b32 %2, %4, %2, %0\n v_bfi_b32 %3, %4, %3, %1"
                             : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(0x7fffffffu));
            else if constexpr (KIND == K_AND_OR)
                asm volatile("v_and_or_b32 %0, %0, %4, %1\n v_and_or_b32 %1, %1, %4, %2\n v_and_or_b32 %2, %2, %4, %3\n v_and_or_b32 %3, %3, %4, %0\n"
                             "v_and_or_b32 %0, %0, %4, %2\n v_and_or_b32 %1, %1, %4, %3\n v_and_or_b32 %2, %2, %4, %0\n v_and_or_b32 %3, %3, %4, %1"
                             : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(0x80000000u));
            else if constexpr (KIND == K_MIN3_PLAIN)
                asm volatile("v_min3_f32 %0, %0, %1, %2\n v_min3_f32 %1, %1, %2, %3\n v_min3_f32 %2, %2, %3, %4\n"
                             "v_min3_f32 %3, %3, %4, %5\n v_min3_f32 %4, %4, %5, %6\n v_min3_f32 %5, %5, %6, %7\n"
                             "v_min3_f32 %6, %6, %7, %0\n v_min3_f32 %7, %7, %0, %1"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            else if constexpr (KIND == K_MIN2_E32)
                asm volatile("v_min_f32_e32 %0, %0, %1\n v_min_f32_e32 %1, %1, %2\n v_min_f32_e32 %2, %2, %3\n v_min_f32_e32 %3, %3, %4\n"
                             "v_min_f32_e32 %4, %4, %5\n v_min_f32_e32 %5, %5, %6\n v_min_f32_e32 %6, %6, %7\n v_min_f32_e32 %7, %7, %0"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            else if constexpr (KIND == K_AND)
                asm volatile("v_and_b32 %0, %0, %1\n v_and_b32 %1, %1, %2\n v_and_b32 %2, %2, %3\n v_and_b32 %3, %3, %0\n"
                             "v_and_b32 %0, %0, %2\n v_and_b32 %1, %1, %3\n v_and_b32 %2, %2, %0\n v_and_b32 %3, %3, %1"
                             : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));
            else if constexpr (KIND == K_CMP_GT_I32)
                asm volatile("v_cmp_gt_i32_e64 %0, 0, %2\n v_cmp_gt_i32_e64 %1, 0, %3\n v_cmp_gt_i32_e64 %0, 0, %4\n v_cmp_gt_i32_e64 %1, 0, %5\n"
                             "v_cmp_gt_i32_e64 %0, 0, %2\n v_cmp_gt_i32_e64 %1, 0, %3\n v_cmp_gt_i32_e64 %0, 0, %4\n v_cmp_gt_i32_e64 %1, 0, %5"
                             : "+s"(s0), "+s"(s1) : "v"(u0), "v"(u1), "v"(u2), "v"(u3));
            else if constexpr (KIND == K_CMP_VCC)
                asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cmp_lt_f32 vcc, %1, %2\n v_cmp_lt_f32 vcc, %2, %3\n v_cmp_lt_f32 vcc, %3, %0\n"
                             "v_cmp_lt_f32 vcc, %0, %2\n v_cmp_lt_f32 vcc, %1, %3\n v_cmp_lt_f32 vcc, %2, %0\n v_cmp_lt_f32 vcc, %3, %1"
                             :: "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "vcc");
            else if constexpr (KIND == K_CNDMASK_VCC)
                asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %0, vcc\n"
                             "v_cndmask_b32 %0, %0, %2, vcc\n v_cndmask_b32 %1, %1, %3, vcc\n v_cndmask_b32 %2, %2, %0, vcc\n v_cndmask_b32 %3, %3, %1, vcc"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : : "vcc");
            else if constexpr (KIND == K_SDWA_ADD)
                asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
                             "v_add_u32_sdwa %1, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
                             "v_add_u32_sdwa %2, %2, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
                             "v_add_u32_sdwa %3, %3, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
                             "v_add_u32_sdwa %0, %0, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
                             "v_add_u32_sdwa %1, %1, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
                             "v_add_u32_sdwa %2, %2, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
                             "v_add_u32_sdwa %3, %3, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1"
                             : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));
            else if constexpr (KIND == K_LSHL_OR)
                asm volatile("v_lshl_or_b32 %0, %0, 1, %1\n v_lshl_or_b32 %1, %1, 1, %2\n v_lshl_or_b32 %2, %2, 1, %3\n v_lshl_or_b32 %3, %3, 1, %0\n"
                             "v_lshl_or_b32 %0, %0, 1, %2\n v_lshl_or_b32 %1, %1, 1, %3\n v_lshl_or_b32 %2, %2, 1, %0\n v_lshl_or_b32 %3, %3, 1, %1"
                             : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));
            else if constexpr (KIND == K_SUB_F32)
                asm volatile("v_sub_f32 %0, %0, %8\n v_sub_f32 %1, %1, %8\n v_sub_f32 %2, %2, %8\n v_sub_f32 %3, %3, %8\n"
                             "v_sub_f32 %4, %4, %8\n v_sub_f32 %5, %5, %8\n v_sub_f32 %6, %6, %8\n v_sub_f32 %7, %7, %8"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0));
            else if constexpr (KIND == K_CVT_I32_F64)
                asm volatile("v_cvt_i32_f64 %0, %4\n v_cvt_i32_f64 %1, %5\n v_cvt_i32_f64 %2, %6\n v_cvt_i32_f64 %3, %7\n"
                             "v_cvt_i32_f64 %0, %5\n v_cvt_i32_f64 %1, %6\n v_cvt_i32_f64 %2, %7\n v_cvt_i32_f64 %3, %4"
                             : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(d0), "v"(d1), "v"(d2), "v"(d3));
            else if constexpr (KIND == K_CVT_F64_I32)
                asm volatile("v_cvt_f64_i32 %0, %4\n v_cvt_f64_i32 %1, %5\n v_cvt_f64_i32 %2, %6\n v_cvt_f64_i32 %3, %7\n"
                             "v_cvt_f64_i32 %0, %5\n v_cvt_f64_i32 %1, %6\n v_cvt_f64_i32 %2, %7\n v_cvt_f64_i32 %3, %4"
                             : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(u0), "v"(u1), "v"(u2), "v"(u3));
            else if constexpr (KIND == K_TRUNC_F64)
                asm volatile("v_trunc_f64 %0, %0\n v_trunc_f64 %1, %1\n v_trunc_f64 %2, %2\n v_trunc_f64 %3, %3\n"
                             "v_trunc_f64 %0, %0\n v_trunc_f64 %1, %1\n v_trunc_f64 %2, %2\n v_trunc_f64 %3, %3"
                             : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));
            else if constexpr (KIND == K_FLOOR_F64)
                asm volatile("v_floor_f64 %0, %0\n v_floor_f64 %1, %1\n v_floor_f64 %2, %2\n v_floor_f64 %3, %3\n"
                             "v_floor_f64 %0, %0\n v_floor_f64 %1, %1\n v_floor_f64 %2, %2\n v_floor_f64 %3, %3"
                             : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));
            else if constexpr (KIND == K_DS_READ_B32) {
                float q0, q1, q2, q3, q4, q5, q6, q7;
                asm volatile("ds_read_b32 %0, %8\n ds_read_b32 %1, %8 offset:256\n ds_read_b32 %2, %8 offset:512\n ds_read_b32 %3, %8 offset:768\n"
                             "ds_read_b32 %4, %8 offset:1024\n ds_read_b32 %5, %8 offset:1280\n ds_read_b32 %6, %8 offset:1536\n ds_read_b32 %7, %8 offset:1792\n"
                             "s_waitcnt lgkmcnt(0)"
                             : "=v"(q0), "=v"(q1), "=v"(q2), "=v"(q3), "=v"(q4), "=v"(q5), "=v"(q6), "=v"(q7) : "v"(addr) : "memory");
                a0 += q0 + q1 + q2 + q3 + q4 + q5 + q6 + q7;
            } else if constexpr (KIND == K_DS_READ_B64) {
                double q0, q1, q2, q3, q4, q5, q6, q7;
                asm volatile("ds_read_b64 %0, %8\n ds_read_b64 %1, %8 offset:512\n ds_read_b64 %2, %8 offset:1024\n ds_read_b64 %3, %8 offset:1536\n"
                             "ds_read_b64 %4, %8 offset:2048\n ds_read_b64 %5, %8 offset:2560\n ds_read_b64 %6, %8 offset:3072\n ds_read_b64 %7, %8 offset:3584\n"
                             "s_waitcnt lgkmcnt(0)"
                             : "=v"(q0), "=v"(q1), "=v"(q2), "=v"(q3), "=v"(q4), "=v"(q5), "=v"(q6), "=v"(q7) : "v"(addr8) : "memory");
                d0 += q0 + q1 + q2 + q3 + q4 + q5 + q6 + q7;
            } else if constexpr (KIND == K_DS_READ2_B32) {
                double q0, q1, q2, q3, q4, q5, q6, q7;
                asm volatile("ds_read2_b32 %0, %8 offset0:0 offset1:64\n ds_read2_b32 %1, %8 offset0:128 offset1:192\n ds_read2_b32 %2, %8 offset0:1 offset1:65\n"
                             "ds_read2_b32 %3, %8 offset0:129 offset1:193\n ds_read2_b32 %4, %8 offset0:2 offset1:66\n ds_read2_b32 %5, %8 offset0:130 offset1:194\n"
                             "ds_read2_b32 %6, %8 offset0:3 offset1:67\n ds_read2_b32 %7, %8 offset0:131 offset1:195\n s_waitcnt lgkmcnt(0)"
                             : "=v"(q0), "=v"(q1), "=v"(q2), "=v"(q3), "=v"(q4), "=v"(q5), "=v"(q6), "=v"(q7) : "v"(addr) : "memory");
                d0 += q0 + q1 + q2 + q3 + q4 + q5 + q6 + q7;
            } else if constexpr (KIND == K_DS_READ_B128) {
                float4 q0, q1, q2, q3, q4, q5, q6, q7;
                asm volatile("ds_read_b128 %0, %8\n ds_read_b128 %1, %8 offset:1024\n ds_read_b128 %2, %8 offset:2048\n ds_read_b128 %3, %8 offset:3072\n"
                             "ds_read_b128 %4, %8\n ds_read_b128 %5, %8 offset:1024\n ds_read_b128 %6, %8 offset:2048\n ds_read_b128 %7, %8 offset:3072\n"
                             "s_waitcnt lgkmcnt(0)"
                             : "=v"(q0), "=v"(q1), "=v"(q2), "=v"(q3), "=v"(q4), "=v"(q5), "=v"(q6), "=v"(q7) : "v"(addr16) : "memory");
                a0 += q0.x + q1.y + q2.z + q3.w + q4.x + q5.y + q6.z + q7.w;
            } else if constexpr (KIND == K_DS_WRITE_B32)
                asm volatile("ds_write_b32 %0, %1\n ds_write_b32 %0, %2 offset:256\n ds_write_b32 %0, %3 offset:512\n ds_write_b32 %0, %4 offset:768\n"
                             "ds_write_b32 %0, %1 offset:1024\n ds_write_b32 %0, %2 offset:1280\n ds_write_b32 %0, %3 offset:1536\n ds_write_b32 %0, %4 offset:1792\n"
                             "s_waitcnt lgkmcnt(0)"
                             :: "v"(addr), "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "memory");
            else if constexpr (KIND == K_DS_WRITE_ADDTID)
                asm volatile("s_mov_b32 m0, %0\n s_nop 0\n ds_write_addtid_b32 %1\n ds_write_addtid_b32 %2 offset:256\n ds_write_addtid_b32 %3 offset:512\n"
                             "ds_write_addtid_b32 %4 offset:768\n ds_write_addtid_b32 %1 offset:1024\n ds_write_addtid_b32 %2 offset:1280\n"
                             "ds_write_addtid_b32 %3 offset:1536\n ds_write_addtid_b32 %4 offset:1792\n s_waitcnt lgkmcnt(0)"
                             :: "s"(ldsbase), "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "memory", "m0");
            else if constexpr (KIND == K_DS_WRITE_B64)
                asm volatile("ds_write_b64 %0, %1\n ds_write_b64 %0, %2 offset:512\n ds_write_b64 %0, %3 offset:1024\n ds_write_b64 %0, %4 offset:1536\n"
                             "ds_write_b64 %0, %1 offset:2048\n ds_write_b64 %0, %2 offset:2560\n ds_write_b64 %0, %3 offset:3072\n ds_write_b64 %0, %4 offset:3584\n"
                             "s_waitcnt lgkmcnt(0)"
                             :: "v"(addr8), "v"(d0), "v"(d1), "v"(d2), "v"(d3) : "memory");
            else if constexpr (KIND == K_DS_MIX_LDPC) {
                // the decoder's LDS mix per pair of edges: 2 reads per store, half the stores lane-linear (addtid)
                float q0, q1, q2, q3;
                asm volatile("s_mov_b32 m0, %4\n ds_read_b32 %0, %5\n ds_read_b32 %1, %5 offset:256\n ds_write_b32 %5, %6 offset:512\n ds_write_addtid_b32 %7 offset:768\n"
                             "ds_read_b32 %2, %5 offset:1024\n ds_read_b32 %3, %5 offset:1280\n ds_write_b32 %5, %6 offset:1536\n ds_write_addtid_b32 %7 offset:1792\n"
                             "s_waitcnt lgkmcnt(0)"
                             : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3) : "s"(ldsbase), "v"(addr), "v"(a1), "v"(a2) : "memory", "m0");
                a0 += q0 + q1 + q2 + q3;
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { cyc[2 * blockIdx.x] = t1 - t0; cyc[2 * blockIdx.x + 1] = r1 - r0; }
    sink[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(d0 + d1 + d2 + d3) + (float)(u0 ^ u1 ^ u2 ^ u3) + (float)(s0 ^ s1) + lds[threadIdx.x];
}

static double g_clock_sum = 0.0; static int g_clock_n = 0;
static double g_resid[K_N][6], g_p50[K_N][6]; static int g_col = 0;
template <int KIND>
void run(int cus) {
    const bool is_lds = KIND >= K_DS_READ_B32;
    std::printf("%-38s", kNames[KIND]);
    for (int w : {1, 2, 4, 5, 6, 8}) {
        const int grid = cus * 4 * w;
        unsigned long long* cyc; float* sink;
        (void)hipMalloc(&cyc, (size_t)grid * 16); (void)hipMalloc(&sink, (size_t)grid * 64 * 4);
        const int iters = 600;
        unsigned* arrive; (void)hipMalloc(&arrive, 8); (void)hipMemset(arrive, 0, 8);
        hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(64), 0, 0, cyc, sink, 4, (unsigned*)nullptr);
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(64), 0, 0, cyc, sink, iters, arrive);
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        float wall_ms = 0; (void)hipEventElapsedTime(&wall_ms, e0, e1);
        if (hipGetLastError() != hipSuccess) { std::printf(" launch failed\n"); return; }
        std::vector<unsigned long long> both(2 * (size_t)grid), h(grid), rt(grid);
        (void)hipMemcpy(both.data(), cyc, (size_t)grid * 16, hipMemcpyDeviceToHost);
        for (int i = 0; i < grid; ++i) { h[i] = both[2 * i]; rt[i] = both[2 * i + 1]; }
        std::sort(h.begin(), h.end()); std::sort(rt.begin(), rt.end());
        // All wavefronts start together (start line above); the SIMD's arbiter favours older wavefronts, so lifetimes
        // differ and the MEDIAN lifetime understates the cost — the cost is the makespan (longest lifetime; the 99th
        // percentile is used so that one straggling CU cannot set it) over the instructions of all W wavefronts.
        const double med = (double)h[(size_t)grid * 99 / 100];
        g_p50[KIND][g_col % 6] = (double)h[grid / 2] / ((double)iters * 16 * kInstrPerBlock[KIND] * ((KIND >= K_DS_READ_B32) ? 4 * w : w));
        g_clock_sum += (double)h[grid / 2] / ((double)rt[grid / 2] * 10e-9) * 1e-9; g_clock_n += 1;   // shader GHz while this ran (median lifetime in both clocks)
        const double n = (double)iters * 16 * kInstrPerBlock[KIND];
        // VALU/SALU: cycles of the SIMD per wave-instruction; LDS: cycles of the CU's LDS pipeline per wave-instruction
        const double per = med / (n * (is_lds ? 4 * w : w));
        std::printf(" %6.2f", per);
        // residency check: launch wall time / median wavefront lifetime (1.0 = all wavefronts ran side by side)
        unsigned arr[2]; (void)hipMemcpy(arr, arrive, 8, hipMemcpyDeviceToHost);
        g_resid[KIND][g_col++ % 6] = (double)arr[1] / grid; (void)wall_ms;
        (void)hipFree(cyc); (void)hipFree(sink); (void)hipFree(arrive);
    }
    std::printf("\n");
}

template <int... Ks> void run_all(int cus, std::integer_sequence<int, Ks...>) { (run<Ks>(cus), ...); }

int main() {
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    std::printf("# %s, %d CUs; shader cycles (s_memtime) per wave-instruction from the makespan of a grid that starts together, every CU busy\n", p.gcnArchName, p.multiProcessorCount);
    std::printf("# VALU/SALU rows: cycles of ONE SIMD per wave-instruction with W wavefronts per SIMD issuing independent instructions\n");
    std::printf("# LDS rows: cycles of the CU's LDS pipeline per wave-instruction with W wavefronts per SIMD (4W per CU), drained every 8\n");
    std::printf("%-38s %6s %6s %6s %6s %6s %6s\n", "instruction \\ W =", "1", "2", "4", "5", "6", "8");
    run_all(p.multiProcessorCount, std::make_integer_sequence<int, K_N>{});
    std::printf("# the same from the MEDIAN wavefront lifetime (lower where the arbiter lets older wavefronts finish first)\n");
    for (int kd = 0; kd < K_N; ++kd) { std::printf("#   %-34s", kNames[kd]); for (int c = 0; c < 6; ++c) std::printf(" %6.2f", g_p50[kd][c]); std::printf("\n"); }
    std::printf("# residency check: fraction of wavefronts that gave up waiting at the start line (0 = the whole grid was resident and started together)\n");
    for (int kd = 0; kd < K_N; ++kd) { std::printf("#   %-34s", kNames[kd]); for (int c = 0; c < 6; ++c) std::printf(" %6.2f", g_resid[kd][c]); std::printf("\n"); }
    std::printf("# shader clock while measuring (s_memtime ticks per s_memrealtime second, mean over all runs): %.3f GHz\n", g_clock_sum / g_clock_n);
    return 0;
}
