// Does gfx950 execute scalar memory atomics (s_atomic_add with return)?  One wavefront per workgroup takes a ticket.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ void k(unsigned* counter, unsigned* out) {
    unsigned t = 1;
    asm volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(t) : "s"(counter) : "memory");
    if (threadIdx.x == 0) out[blockIdx.x] = t;
}
int main() {
    const int n = 4096;
    unsigned *c, *o;
    hipMalloc(&c, 4); hipMalloc(&o, n * 4); hipMemset(c, 0, 4);
    hipLaunchKernelGGL(k, dim3(n), dim3(64), 0, 0, c, o);
    if (hipDeviceSynchronize() != hipSuccess) { std::printf("kernel failed\n"); return 1; }
    std::vector<unsigned> h(n); unsigned total;
    hipMemcpy(h.data(), o, n * 4, hipMemcpyDeviceToHost); hipMemcpy(&total, c, 4, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    bool ok = total == (unsigned)n;
    for (int i = 0; i < n; ++i) ok = ok && h[i] == (unsigned)i;
    std::printf("scalar atomic tickets: counter %u, distinct 0..%d: %s\n", total, n - 1, ok ? "yes" : "NO");
    return ok ? 0 : 2;
}
