// LDS allocation granule of gfx950, measured: how many 64-thread workgroups fit a CU as the dynamic LDS size grows
// (hipOccupancyMaxActiveBlocksPerMultiprocessor), printed where the count changes.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/lds_granule.hip -o build/lds_granule && build/lds_granule
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float* out) { extern __shared__ float s[]; s[threadIdx.x] = 1.0f; __syncthreads(); out[threadIdx.x] = s[(threadIdx.x + 1) & 63]; }
int main() {
    int prev = -1;
    for (size_t lds = 1024; lds <= 163840; lds += 16) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, 64, lds) != hipSuccess) { std::printf("query failed at %zu\n", lds); return 1; }
        if (n != prev) { std::printf("dynamic LDS %6zu B: %2d workgroups per CU (%zu B each if the LDS were split evenly)\n", lds, n, n ? 163840 / (size_t)n : 0); prev = n; }
        if (n <= 8) break;
    }
    return 0;
}
