// anyorder.hip - does a kernel launched with hipExtAnyOrderLaunch (AQL packet without the barrier bit) overlap its
// predecessor in the SAME stream on gfx950 / ROCm 7.2?  Two kernels of 128 one-wave workgroups that each spin for ~50 us:
// back to back they take ~100 us, overlapped ~50.  Also a chain A, B(any), C(any), D(normal) where D checks that it sees
// the values written by A, B and C (the barrier bit of D must wait for all three).
// build: hipcc --offload-arch=gfx950 -O2 -o anyorder anyorder.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>

__global__ void k_spin(int* out, int value, long long cycles) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) out[blockIdx.x] = value;
}
__global__ void k_check(const int* a, const int* b, const int* c, int n, int va, int vb, int vc, int* bad) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && (a[i] != va || b[i] != vb || c[i] != vc)) atomicAdd(bad, 1);
}

int main() {
    const int G = 128;
    int *a, *b, *c, *bad;
    hipMalloc(&a, G * 4); hipMalloc(&b, G * 4); hipMalloc(&c, G * 4); hipMalloc(&bad, 4);
    hipStream_t s;
    hipStreamCreate(&s);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const long long cyc = 5000;  // wall_clock64 ticks at 100 MHz: 50 us
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 4; ++rep) {
            hipMemsetAsync(a, 0, G * 4, s); hipMemsetAsync(b, 0, G * 4, s); hipMemsetAsync(c, 0, G * 4, s); hipMemsetAsync(bad, 0, 4, s);
            hipStreamSynchronize(s);
            hipEventRecord(e0, s);
            const unsigned fl = mode ? hipExtAnyOrderLaunch : 0;
            hipExtLaunchKernelGGL(k_spin, dim3(G), dim3(64), 0, s, nullptr, nullptr, 0, a, rep + 1, cyc);
            hipExtLaunchKernelGGL(k_spin, dim3(G), dim3(64), 0, s, nullptr, nullptr, fl, b, rep + 11, cyc);
            hipExtLaunchKernelGGL(k_spin, dim3(G), dim3(64), 0, s, nullptr, nullptr, fl, c, rep + 21, cyc);
            hipExtLaunchKernelGGL(k_check, dim3(1), dim3(128), 0, s, nullptr, nullptr, 0, (const int*)a, (const int*)b, (const int*)c, G,
                                  rep + 1, rep + 11, rep + 21, bad);
            hipEventRecord(e1, s);
            hipStreamSynchronize(s);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            int hb = -1;
            hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
            printf("mode %s rep %d: %.1f us for three 50-us kernels + check, mismatches seen by the dependent kernel: %d\n",
                   mode ? "any-order" : "in-order", rep, ms * 1e3, hb);
        }
    }
    return 0;
}
