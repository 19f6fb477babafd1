// issue_table2.hip — round 4 supplement to issue_table.hip: issue cost (shader cycles of one SIMD per wave-instruction, W
// wavefronts per SIMD issuing independent instructions, every CU busy) of the candidate replacements for the decoder's
// sign / minimum arithmetic.  Same method as issue_table.hip (start line, 99th-percentile lifetime).
//   hipcc --offload-arch=gfx950 -O3 -o issue_table2 tools/ubench/issue_table2.hip && ./issue_table2
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define U4(op) op " %0, %0, %1\n " op " %1, %1, %2\n " op " %2, %2, %3\n " op " %3, %3, %0\n " op " %0, %0, %2\n " op " %1, %1, %3\n " op " %2, %2, %0\n " op " %3, %3, %1"
#define U4_3(op) op " %0, %0, %1, %2\n " op " %1, %1, %2, %3\n " op " %2, %2, %3, %0\n " op " %3, %3, %0, %1\n " op " %0, %0, %2, %3\n " op " %1, %1, %3, %0\n " op " %2, %2, %0, %1\n " op " %3, %3, %1, %2"
#define U4_LIT(op, lit) op " %0, " lit ", %0\n " op " %1, " lit ", %1\n " op " %2, " lit ", %2\n " op " %3, " lit ", %3\n " op " %0, " lit ", %0\n " op " %1, " lit ", %1\n " op " %2, " lit ", %2\n " op " %3, " lit ", %3"
#define U4_S(op) op " %0, %0, %1, %4\n " op " %1, %1, %2, %4\n " op " %2, %2, %3, %4\n " op " %3, %3, %0, %4\n " op " %0, %0, %2, %4\n " op " %1, %1, %3, %4\n " op " %2, %2, %0, %4\n " op " %3, %3, %1, %4"

struct Op { const char* name; };
static const Op kOps[] = {
    {"v_xor_b32 (reference: fast class)"}, {"v_or_b32"}, {"v_min_u32"}, {"v_max_u32"}, {"v_min_i32"}, {"v_sub_u32"}, {"v_lshlrev_b32"},
    {"v_ashrrev_i32"}, {"v_max_f32_e32"}, {"v_med3_f32"}, {"v_add3_u32"}, {"v_or3_b32"}, {"v_and_b32 literal"}, {"v_xor_b32 literal"},
    {"v_mul_f32 literal"}, {"v_fmac_f32"}, {"v_cndmask_b32_e64 sgpr (no modifiers)"}, {"v_and_or_b32 sgpr mask"}, {"v_bfi_b32 sgpr mask"},
    {"v_sub_f32 e32"}, {"v_mad_u32_u24"}, {"v_bfe_u32"}, {"v_min3_u32"}, {"v_xad_u32"}, {"v_lshl_add_u32"}, {"v_mul_legacy_f32"},
    {"v_perm_b32"}, {"v_alignbit_b32"}, {"v_cmp_lt_i32 -> sgpr"}, {"v_cmp_class_f32 -> sgpr"}, {"v_xor3-like: 2 x v_xor_b32"},
};
constexpr int kNOps = sizeof(kOps) / sizeof(kOps[0]);

template <int KIND>
__global__ __launch_bounds__(64) void k(unsigned long long* cyc, float* sink, int iters, unsigned* arrive) {
    unsigned u0 = threadIdx.x + 77, u1 = u0 * 3 + 1, u2 = u0 * 5 + 2, u3 = u0 * 7 + 3;
    float a0 = threadIdx.x + 1.0f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    unsigned long long s0 = 0x5555555555555555ull, s1 = 0;
    const unsigned smask = 0x80000000u;
    if (arrive && threadIdx.x == 0) {
        atomicAdd(arrive, 1u);
        const unsigned long long w0 = __builtin_amdgcn_s_memrealtime();
        while (__hip_atomic_load(arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x && __builtin_amdgcn_s_memrealtime() - w0 < 200000ull)
            __builtin_amdgcn_s_sleep(8);
    }
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if constexpr (KIND == 0) asm volatile(U4("v_xor_b32") : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));
            else if constexpr (KIND == 1) asm volatile(U4("v_or_b32") : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));
            else if constexpr (KIND == 2) asm volatile(U4("v_min_u32") : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));
            else if constexpr (KIND == 3) asm volatile(U4("v_max_u32") : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));
            else if constexpr (KIND == 4) asm volatile(U4("v_min_i32") : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));
            else if constexpr (KIND == 5) asm volatile(U4("v_sub_u32") : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));
            else if constexpr (KIND == 6) asm volatile(U4("v_lshlrev_b32") : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));
            else if constexpr (KIND == 7) asm volatile(U4("v_ashrrev_i32") : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));
            else if constexpr (KIND == 8) asm volatile(U4("v_max_f32_e32") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
            else if constexpr (KIND == 9) asm volatile(U4_3("v_med3_f32") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
            else if constexpr (KIND == 10) asm volatile(U4_3("v_add3_u32") : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));
            else if constexpr (KIND == 11) asm volatile(U4_3("v_or3_b32") : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));
            else if constexpr (KIND == 12) asm volatile(U4_LIT("v_and_b32", "0x80000001") : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));
            else if constexpr (KIND == 13) asm volatile(U4_LIT("v_xor_b32", "0x80000001") : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));
            else if constexpr (KIND == 14) asm volatile(U4_LIT("v_mul_f32", "0x3f400000") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
            else if constexpr (KIND == 15) asm volatile(U4("v_fmac_f32") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
            else if constexpr (KIND == 16) asm volatile(U4_S("v_cndmask_b32_e64") : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "s"(s0));
            else if constexpr (KIND == 17) asm volatile("v_and_or_b32 %0, %1, %4, %0\n v_and_or_b32 %1, %2, %4, %1\n v_and_or_b32 %2, %3, %4, %2\n v_and_or_b32 %3, %0, %4, %3\n"
                                                        "v_and_or_b32 %0, %2, %4, %0\n v_and_or_b32 %1, %3, %4, %1\n v_and_or_b32 %2, %0, %4, %2\n v_and_or_b32 %3, %1, %4, %3"
                                                        : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "s"(smask));
            else if constexpr (KIND == 18) asm volatile("v_bfi_b32 %0, %4, %0, %1\n v_bfi_b32 %1, %4, %1, %2\n v_bfi_b32 %2, %4, %2, %3\n v_bfi_b32 %3, %4, %3, %0\n"
                                                        "v_bfi_b32 %0, %4, %0, %2\n v_bfi_b32 %1, %4, %1, %3\n v_bfi_b32 %2, %4, %2, %0\n v_bfi_b32 %3, %4, %3, %1"
                                                        : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "s"(~smask));
            else if constexpr (KIND == 19) asm volatile(U4("v_sub_f32_e32") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
            else if constexpr (KIND == 20) asm volatile(U4_3("v_mad_u32_u24") : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));
            else if constexpr (KIND == 21) asm volatile("v_bfe_u32 %0, %0, 3, 9\n v_bfe_u32 %1, %1, 3, 9\n v_bfe_u32 %2, %2, 3, 9\n v_bfe_u32 %3, %3, 3, 9\n"
                                                        "v_bfe_u32 %0, %0, 1, 9\n v_bfe_u32 %1, %1, 1, 9\n v_bfe_u32 %2, %2, 1, 9\n v_bfe_u32 %3, %3, 1, 9"
                                                        : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));
            else if constexpr (KIND == 22) asm volatile(U4_3("v_min3_u32") : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));
            else if constexpr (KIND == 23) asm volatile(U4_3("v_xad_u32") : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));
            else if constexpr (KIND == 24) asm volatile("v_lshl_add_u32 %0, %0, 1, %1\n v_lshl_add_u32 %1, %1, 1, %2\n v_lshl_add_u32 %2, %2, 1, %3\n v_lshl_add_u32 %3, %3, 1, %0\n"
                                                        "v_lshl_add_u32 %0, %0, 1, %2\n v_lshl_add_u32 %1, %1, 1, %3\n v_lshl_add_u32 %2, %2, 1, %0\n v_lshl_add_u32 %3, %3, 1, %1"
                                                        : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));
            else if constexpr (KIND == 25) asm volatile(U4("v_mul_legacy_f32") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
            else if constexpr (KIND == 26) asm volatile(U4_3("v_perm_b32") : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));
            else if constexpr (KIND == 27) asm volatile("v_alignbit_b32 %0, %0, %1, 7\n v_alignbit_b32 %1, %1, %2, 7\n v_alignbit_b32 %2, %2, %3, 7\n v_alignbit_b32 %3, %3, %0, 7\n"
                                                        "v_alignbit_b32 %0, %0, %2, 7\n v_alignbit_b32 %1, %1, %3, 7\n v_alignbit_b32 %2, %2, %0, 7\n v_alignbit_b32 %3, %3, %1, 7"
                                                        : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));
            else if constexpr (KIND == 28) asm volatile("v_cmp_lt_i32_e64 %0, %2, 0\n v_cmp_lt_i32_e64 %1, %3, 0\n v_cmp_lt_i32_e64 %0, %4, 0\n v_cmp_lt_i32_e64 %1, %5, 0\n"
                                                        "v_cmp_lt_i32_e64 %0, %3, 0\n v_cmp_lt_i32_e64 %1, %2, 0\n v_cmp_lt_i32_e64 %0, %5, 0\n v_cmp_lt_i32_e64 %1, %4, 0"
                                                        : "+s"(s0), "+s"(s1) : "v"(u0), "v"(u1), "v"(u2), "v"(u3));
            else if constexpr (KIND == 29) asm volatile("v_cmp_class_f32_e64 %0, %2, 0x3c\n v_cmp_class_f32_e64 %1, %3, 0x3c\n v_cmp_class_f32_e64 %0, %4, 0x3c\n v_cmp_class_f32_e64 %1, %5, 0x3c\n"
                                                        "v_cmp_class_f32_e64 %0, %3, 0x3c\n v_cmp_class_f32_e64 %1, %2, 0x3c\n v_cmp_class_f32_e64 %0, %5, 0x3c\n v_cmp_class_f32_e64 %1, %4, 0x3c"
                                                        : "+s"(s0), "+s"(s1) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
            else if constexpr (KIND == 30) asm volatile(U4("v_xor_b32") : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    sink[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + (float)(u0 ^ u1 ^ u2 ^ u3) + (float)(s0 ^ s1);
}

template <int KIND>
void run(int cus) {
    std::printf("%-40s", kOps[KIND].name);
    for (int w : {1, 2, 4, 5, 6, 8}) {
        const int grid = cus * 4 * w;
        unsigned long long* cyc; float* sink; unsigned* arrive;
        (void)hipMalloc(&cyc, (size_t)grid * 8); (void)hipMalloc(&sink, (size_t)grid * 64 * 4); (void)hipMalloc(&arrive, 8); (void)hipMemset(arrive, 0, 8);
        const int iters = 600;
        hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(64), 0, 0, cyc, sink, 4, (unsigned*)nullptr);
        hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(64), 0, 0, cyc, sink, iters, arrive);
        (void)hipDeviceSynchronize();
        if (hipGetLastError() != hipSuccess) { std::printf(" launch failed\n"); return; }
        std::vector<unsigned long long> h(grid);
        (void)hipMemcpy(h.data(), cyc, (size_t)grid * 8, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        std::printf(" %6.2f", (double)h[(size_t)grid * 99 / 100] / ((double)iters * 16 * 8 * w));
        (void)hipFree(cyc); (void)hipFree(sink); (void)hipFree(arrive);
    }
    std::printf("\n");
}
template <int... Ks> void run_all(int cus, std::integer_sequence<int, Ks...>) { (run<Ks>(cus), ...); }

int main() {
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    std::printf("# %s, %d CUs; shader cycles of ONE SIMD per wave-instruction, W wavefronts per SIMD, independent instructions (method: issue_table.hip)\n", p.gcnArchName, p.multiProcessorCount);
    std::printf("%-40s %6s %6s %6s %6s %6s %6s\n", "instruction \\ W =", "1", "2", "4", "5", "6", "8");
    run_all(p.multiProcessorCount, std::make_integer_sequence<int, kNOps>{});
    return 0;
}
