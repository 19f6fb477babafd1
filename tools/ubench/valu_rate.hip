// VALU issue-rate microbenchmark for gfx950: independent chains per wave, W waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int KIND>
__global__ __launch_bounds__(64) void k(float* out, int iters) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3;
    unsigned u0 = threadIdx.x, u1 = u0 * 3, u2 = u0 * 5, u3 = u0 * 7;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (KIND == 0) { // f32 fma x8
                asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n"
                             "v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if (KIND == 1) { // int xor x8
                asm volatile("v_xor_b32 %0, %0, %1\n v_xor_b32 %1, %1, %2\n v_xor_b32 %2, %2, %3\n v_xor_b32 %3, %3, %0\n"
                             "v_xor_b32 %0, %0, %2\n v_xor_b32 %1, %1, %3\n v_xor_b32 %2, %2, %0\n v_xor_b32 %3, %3, %1"
                             : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));
            } else if (KIND == 2) { // f64 fma x8
                asm volatile("v_fma_f64 %0, %0, %0, %0\n v_fma_f64 %1, %1, %1, %1\n v_fma_f64 %2, %2, %2, %2\n v_fma_f64 %3, %3, %3, %3\n"
                             "v_fma_f64 %0, %0, %0, %0\n v_fma_f64 %1, %1, %1, %1\n v_fma_f64 %2, %2, %2, %2\n v_fma_f64 %3, %3, %3, %3"
                             : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));
            } else if (KIND == 3) { // f32 mul/add x8 (non-fma)
                asm volatile("v_mul_f32 %0, %0, %1\n v_add_f32 %1, %1, %2\n v_mul_f32 %2, %2, %3\n v_add_f32 %3, %3, %4\n"
                             "v_mul_f32 %4, %4, %5\n v_add_f32 %5, %5, %6\n v_mul_f32 %6, %6, %7\n v_add_f32 %7, %7, %0"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if (KIND == 4) { // min3_u32 x8
                asm volatile("v_min3_u32 %0, %0, %1, %2\n v_min3_u32 %1, %1, %2, %3\n v_min3_u32 %2, %2, %3, %0\n v_min3_u32 %3, %3, %0, %1\n"
                             "v_min3_u32 %0, %0, %1, %2\n v_min3_u32 %1, %1, %2, %3\n v_min3_u32 %2, %2, %3, %0\n v_min3_u32 %3, %3, %0, %1"
                             : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));
            } else if (KIND == 5) { // f64 mul/add x8
                asm volatile("v_mul_f64 %0, %0, %1\n v_add_f64 %1, %1, %2\n v_mul_f64 %2, %2, %3\n v_add_f64 %3, %3, %0\n"
                             "v_mul_f64 %0, %0, %1\n v_add_f64 %1, %1, %2\n v_mul_f64 %2, %2, %3\n v_add_f64 %3, %3, %0"
                             : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));
            } else if (KIND == 6) { // cmp + cndmask pairs x4
                asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc\n v_cmp_lt_f32 vcc, %1, %2\n v_cndmask_b32 %3, %3, %0, vcc\n"
                             "v_cmp_lt_f32 vcc, %2, %3\n v_cndmask_b32 %0, %0, %1, vcc\n v_cmp_lt_f32 vcc, %3, %0\n v_cndmask_b32 %1, %1, %2, vcc"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : : "vcc");
            }
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(d0 + d1 + d2 + d3) + (float)(u0 ^ u1 ^ u2 ^ u3);
}
template <int KIND> void run(const char* name, int waves_per_simd) {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    const int grid = cus * 4 * waves_per_simd;
    float* out; hipMalloc(&out, (size_t)grid * 64 * 4);
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(64), 0, 0, out, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(64), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = (double)iters * 16 * 8 * waves_per_simd;
    const double clk = p.clockRate * 1e3;   // Hz (nominal)
    printf("%-14s waves/SIMD %d: %.3f ms, %.2f ns per wave-instr per SIMD (%.2f cycles @ %.2f GHz nominal)\n", name,
           waves_per_simd, ms, ms * 1e6 / instr_per_simd, ms * 1e-3 / instr_per_simd * clk, clk * 1e-9);
    hipFree(out);
}
int main() {
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_fma_f32", w); run<3>("v_mul/add_f32", w); run<1>("v_xor_b32", w); run<4>("v_min3_u32", w);
        run<6>("cmp+cndmask", w); run<2>("v_fma_f64", w); run<5>("v_mul/add_f64", w);
    }
    return 0;
}
