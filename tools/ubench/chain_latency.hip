// Latency of dependent chains on gfx950, one wavefront per SIMD: plain v_add_f32, the DPP wave_shr:1
// add used for in-order sums, and an LDS broadcast walk (ds_read_b128 + 4 adds).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(64) void k_add(float* out, int iters) {
    float s = threadIdx.x, x = 1.0f + threadIdx.x;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 64; ++r) asm volatile("v_add_f32 %0, %0, %1" : "+v"(s) : "v"(x));
    }
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
__global__ __launch_bounds__(64) void k_dpp(float* out, int iters) {
    float s = threadIdx.x, x = 1.0f + threadIdx.x;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 64; ++r) asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(s) : "v"(x));
    }
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
__global__ __launch_bounds__(64) void k_lds(float* out, int iters) {
    __shared__ float a[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) a[i] = i;
    __syncthreads();
    float s = 0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float4 q = *reinterpret_cast<const float4*>(&a[((i * 16 + r) * 4) & 1020]); s += q.x; s += q.y; s += q.z; s += q.w; }
    }
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <class K> void run(const char* name, K kern, int waves_per_simd, int per_iter) {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int grid = p.multiProcessorCount * 4 * waves_per_simd;
    float* out; hipMalloc(&out, (size_t)grid * 64 * 4);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64), 0, 0, out, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s waves/SIMD %d: %.3f ms -> %.1f cycles per chain step per wave (2.4 GHz nominal)\n", name, waves_per_simd, ms,
           ms * 1e-3 * 2.4e9 / ((double)iters * per_iter));
    hipFree(out);
}
int main() {
    for (int w : {1, 2, 4}) {
        run("dependent v_add_f32", k_add, w, 64);
        run("s_nop 1 + v_add_f32_dpp", k_dpp, w, 64);
        run("lds b128 broadcast + 4 adds", k_lds, w, 64);
    }
    return 0;
}
