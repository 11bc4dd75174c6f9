// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 against KNOWN byte counts, in the access patterns of the console
// kernels (round-3 review item 3).  Every kernel moves exactly `bytes` = 1 GiB (4x the 256 MiB Infinity Cache, so that
// cache hits cannot hide traffic); run under
//     rocprofv3 --kernel-trace --pmc FETCH_SIZE -- ./hbm_counters      and      ... --pmc WRITE_SIZE -- ./hbm_counters
// tools/hbm_calib.py divides the counters by the true bytes -> per-pattern factors (profiles/round4_hbm_calibration.md).
//   build: hipcc --offload-arch=gfx950 -O3 -o hbm_counters hbm_counters.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

static size_t kBytes = 1ull << 30;  // argv[1] = MiB (default 1024; 64 = inside the Infinity Cache, every kernel then re-reads what the previous left there)
constexpr int kWG = 256;

// 16 B per lane, lane-consecutive (the STFT / loss / apply streams' basic shape)
__global__ void rd_16B(const float4* __restrict__ p, float* __restrict__ out, size_t n4) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * kWG + threadIdx.x; i < n4; i += (size_t)gridDim.x * kWG) { const float4 v = p[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 123.456f) out[0] = acc;
}
// 32 B per lane = two adjacent float4 (the compressor kernels: 8 consecutive samples per lane)
__global__ void rd_32B(const float4* __restrict__ p, float* __restrict__ out, size_t n4) {
    float acc = 0.f;
    for (size_t i = ((size_t)blockIdx.x * kWG + threadIdx.x) * 2; i + 1 < n4; i += (size_t)gridDim.x * kWG * 2) {
        const float4 a = p[i], b = p[i + 1];
        acc += a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w;
    }
    if (acc == 123.456f) out[0] = acc;
}
// EQ slab pattern: a 64-lane wave owns a 4096-float tile; slab j: thread t reads 16 B at float offset (t / 4) * 64 + j * 16 + (t % 4) * 4,
// i.e. 64-byte pieces at a 256-byte pitch, the four slabs of a tile one after the other (mst_eq.hip: slab_fetch)
__global__ void rd_slab(const float* __restrict__ p, float* __restrict__ out, size_t ntiles) {
    float acc = 0.f;
    const int t = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (size_t tile = (size_t)blockIdx.x * (kWG / 64) + w; tile < ntiles; tile += (size_t)gridDim.x * (kWG / 64)) {
        const float* base = p + tile * 4096;
        for (int j = 0; j < 4; ++j)
            for (int q0 = 0; q0 < 4; ++q0) {
                const int q = t + 64 * q0;
                const float4 v = *reinterpret_cast<const float4*>(base + (q / 4) * 64 + j * 16 + (q % 4) * 4);
                acc += v.x + v.y + v.z + v.w;
            }
    }
    if (acc == 123.456f) out[0] = acc;
}
// 4 B per lane, lane-consecutive
__global__ void rd_4B(const float* __restrict__ p, float* __restrict__ out, size_t n) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * kWG + threadIdx.x; i < n; i += (size_t)gridDim.x * kWG) acc += p[i];
    if (acc == 123.456f) out[0] = acc;
}
__global__ void wr_16B(float4* __restrict__ p, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * kWG + threadIdx.x; i < n4; i += (size_t)gridDim.x * kWG) p[i] = make_float4(1.f, 2.f, 3.f, (float)i);
}
__global__ void wr_32B(float4* __restrict__ p, size_t n4) {
    for (size_t i = ((size_t)blockIdx.x * kWG + threadIdx.x) * 2; i + 1 < n4; i += (size_t)gridDim.x * kWG * 2) {
        p[i] = make_float4(1.f, 2.f, 3.f, (float)i);
        p[i + 1] = make_float4(4.f, 5.f, 6.f, (float)i);
    }
}
__global__ void wr_slab(float* __restrict__ p, size_t ntiles) {
    const int t = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (size_t tile = (size_t)blockIdx.x * (kWG / 64) + w; tile < ntiles; tile += (size_t)gridDim.x * (kWG / 64)) {
        float* base = p + tile * 4096;
        for (int j = 0; j < 4; ++j)
            for (int q0 = 0; q0 < 4; ++q0) {
                const int q = t + 64 * q0;
                *reinterpret_cast<float4*>(base + (q / 4) * 64 + j * 16 + (q % 4) * 4) = make_float4(1.f, 2.f, 3.f, (float)tile);
            }
    }
}
// chunk-state layout: 4 B per lane at a large row pitch is not used anywhere; the [row][state][chunk] arrays are read 16 B per lane
// lane-consecutive (rd_16B) - covered.

int main(int argc, char** argv) {
    if (argc > 1) kBytes = (size_t)atoi(argv[1]) << 20;
    void *a, *o;
    if (hipMalloc(&a, kBytes) != hipSuccess || hipMalloc(&o, 256) != hipSuccess) return 1;
    (void)hipMemset(a, 0, kBytes);
    const size_t n4 = kBytes / 16, n = kBytes / 4, ntiles = kBytes / (4096 * 4);
    const dim3 g(2048), b(kWG);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(rd_16B, g, b, 0, 0, (const float4*)a, (float*)o, n4);
        hipLaunchKernelGGL(rd_32B, g, b, 0, 0, (const float4*)a, (float*)o, n4);
        hipLaunchKernelGGL(rd_slab, g, b, 0, 0, (const float*)a, (float*)o, ntiles);
        hipLaunchKernelGGL(rd_4B, g, b, 0, 0, (const float*)a, (float*)o, n);
        hipLaunchKernelGGL(wr_16B, g, b, 0, 0, (float4*)a, n4);
        hipLaunchKernelGGL(wr_32B, g, b, 0, 0, (float4*)a, n4);
        hipLaunchKernelGGL(wr_slab, g, b, 0, 0, (float*)a, ntiles);
    }
    (void)hipDeviceSynchronize();
    printf("bytes per kernel: %zu\n", kBytes);
    return 0;
}
