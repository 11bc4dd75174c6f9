// Developer micro-benchmark: issue rate of v_fma_f32 vs v_pk_fma_f32 on gfx950 (wave64), independent chains.
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
    float x[16];
    f2 p[8];
    for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 0.001f + i;
    for (int i = 0; i < 8; ++i) p[i] = f2{x[2 * i], x[2 * i + 1]};
    const f2 pa = {a, a}, pb = {b, b};
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) x[i] = __builtin_fmaf(x[i], a, b);
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) p[i] = __builtin_elementwise_fma(p[i], pa, pb);
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += x[i];
    for (int i = 0; i < 8; ++i) s += p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    float* out;
    hipMalloc(&out, 1 << 24);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int waves_per_simd = 1; waves_per_simd <= 8; waves_per_simd *= 2) {
        const int blocks = 256 * waves_per_simd;  // 256 CUs x (256 threads = 4 waves = 1 per SIMD) x waves_per_simd
        for (int mode = 0; mode < 2; ++mode) {
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
                else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double insts_per_wave = (double)iters * (mode == 0 ? 16 : 8);
            const double cyc = ms * 1e-3 * 2.4e9;  // nominal clock
            printf("%s waves/SIMD %d: %.3f ms, %.2f cycles per wave-instruction per SIMD (at 2.4 GHz), %.1f TFLOP/s\n",
                   mode == 0 ? "v_fma_f32   " : "v_pk_fma_f32", waves_per_simd, ms, cyc / (insts_per_wave * waves_per_simd),
                   (double)blocks * 256 * iters * 32 / (ms * 1e-3) / 1e12);
        }
    }
    return 0;
}
