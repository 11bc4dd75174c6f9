// Developer micro-benchmark: a 4096-point complex FFT by ONE wave - 64 points per lane in registers, two register FFT-64s with a
// single transpose through LDS between them and no workgroup barrier - against the 512-lane / 4-exchange engine of mst_fft2.h.
//   hipcc --offload-arch=gfx950 -O3 -o fft_wave fft_wave.hip && ./fft_wave
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ float2 mul_mi(float2 a) { return make_float2(a.y, -a.x); }
__device__ __forceinline__ float2 mul_pi(float2 a) { return make_float2(-a.y, a.x); }
__device__ __forceinline__ void fft4(float2& v0, float2& v1, float2& v2, float2& v3) {
    const float2 s02 = cadd(v0, v2), d02 = csub(v0, v2), s13 = cadd(v1, v3), d13 = csub(v1, v3);
    v0 = cadd(s02, s13); v1 = cadd(d02, mul_mi(d13)); v2 = csub(s02, s13); v3 = cadd(d02, mul_pi(d13));
}
__device__ __forceinline__ void bfly8(float2* v) {
    constexpr float c = 0.70710678118654752f;
    float2 e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6], o0 = v[1], o1 = v[3], o2 = v[5], o3 = v[7];
    fft4(e0, e1, e2, e3);
    fft4(o0, o1, o2, o3);
    o1 = make_float2(c * (o1.x + o1.y), c * (o1.y - o1.x));
    o2 = mul_mi(o2);
    o3 = make_float2(c * (o3.y - o3.x), -c * (o3.x + o3.y));
    v[0] = cadd(e0, o0); v[4] = csub(e0, o0); v[1] = cadd(e1, o1); v[5] = csub(e1, o1);
    v[2] = cadd(e2, o2); v[6] = csub(e2, o2); v[3] = cadd(e3, o3); v[7] = csub(e3, o3);
}
struct W64 { float c[64], s[64]; };
constexpr double kPi = 3.14159265358979323846;
// compile-time cos / sin of 2 pi m / 64 by Taylor series on a reduced argument (constexpr-friendly, exact to fp32)
constexpr double ccos(double x) { double t = 1, s = 1; for (int i = 1; i < 14; ++i) { t *= -x * x / ((2 * i - 1) * (2 * i)); s += t; } return s; }
constexpr double csin(double x) { double t = x, s = x; for (int i = 1; i < 14; ++i) { t *= -x * x / ((2 * i) * (2 * i + 1)); s += t; } return s; }
constexpr W64 make_w64() { W64 w{}; for (int m = 0; m < 64; ++m) { double a = 2 * kPi * m / 64; if (a > kPi) a -= 2 * kPi; w.c[m] = (float)ccos(a); w.s[m] = (float)(-csin(a)); } return w; }
__device__ constexpr W64 kW64 = make_w64();

// in-register FFT-64, in place: on entry v[n] natural; on exit X[ka + 8 kb] sits in v[8 ka + kb]
__device__ __forceinline__ void fft64(float2 (&v)[64]) {
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        float2 t[8];
#pragma unroll
        for (int a = 0; a < 8; ++a) t[a] = v[8 * a + b];
        bfly8(t);
#pragma unroll
        for (int ka = 0; ka < 8; ++ka) {
            const int m = (b * ka) & 63;
            v[8 * ka + b] = (m == 0) ? t[ka] : cmul(t[ka], make_float2(kW64.c[m], kW64.s[m]));
        }
    }
#pragma unroll
    for (int ka = 0; ka < 8; ++ka) bfly8(&v[8 * ka]);
}
__device__ __forceinline__ constexpr int dig(int k) { return 8 * (k & 7) + (k >> 3); }  // register of X[k] after fft64

constexpr int kPitch = 65;
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
template <int WAVES, bool HALF, int REPS>
__global__ __launch_bounds__(WAVES * 64) void k_fft4096(const float2* __restrict__ in, float2* __restrict__ out, const float2* __restrict__ tw, int nseq, int mod) {
    extern __shared__ float2 lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float2* buf = lds + wave * 64 * kPitch;
    // inter-stage twiddles W_4096^(lane k1), k1 = 8 a + b, as products of two exactly rounded per-lane factors held in registers
    float2 tA[8], tB[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        tA[i] = tw[(lane * 8 * i) & 4095];
        tB[i] = tw[(lane * i) & 4095];
    }
    for (int seq = blockIdx.x * WAVES + wave; seq < nseq; seq += gridDim.x * WAVES) {
        float2 v[64];
#pragma unroll
        for (int n1 = 0; n1 < 64; ++n1) v[n1] = in[(size_t)(seq % mod) * 4096 + 64 * n1 + lane];
        for (int rep = 0; rep < REPS; ++rep) {
            fft64(v);  // over n1; Y[k1] in v[dig(k1)]
#pragma unroll
            for (int k1 = 1; k1 < 64; ++k1) {
                const float2 w = (k1 & 7) == 0 ? tA[k1 >> 3] : ((k1 >> 3) == 0 ? tB[k1 & 7] : cmul(tA[k1 >> 3], tB[k1 & 7]));
                buf[k1 * kPitch + lane] = cmul(v[dig(k1)], w);
            }
            buf[lane] = v[dig(0)];
            wave_sync();
#pragma unroll
            for (int n2 = 0; n2 < 64; ++n2) v[n2] = buf[lane * kPitch + n2];
            wave_sync();
            fft64(v);  // over n2; X[lane + 64 k2] in v[dig(k2)]
            if (REPS > 1) {  // benchmark only: feed the (digit-swapped) result back, scaled so that it stays finite
#pragma unroll
                for (int k2 = 0; k2 < 64; ++k2) v[k2] = make_float2(v[k2].x * (1.0f / 64.0f), v[k2].y * (1.0f / 64.0f));
            }
        }
#pragma unroll
        for (int k2 = 0; k2 < 64; ++k2) out[(size_t)(seq % mod) * 4096 + lane + 64 * k2] = v[dig(k2)];
    }
}

template <int WAVES, bool HALF, int REPS>
static float run(const float2* in, float2* out, const float2* tw, int nseq, int blocks, int mod = 1 << 30) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t shm = (size_t)WAVES * 64 * kPitch * sizeof(float2);
    hipFuncSetAttribute((const void*)k_fft4096<WAVES, HALF, REPS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fft4096<WAVES, HALF, REPS>), dim3(blocks), dim3(WAVES * 64), shm, 0, in, out, tw, nseq, mod);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    return best;
}

int main() {
    const int nseq = 4096;
    std::vector<float2> h((size_t)nseq * 4096), tw(4096);
    for (size_t i = 0; i < h.size(); ++i) h[i] = make_float2((float)((i * 2654435761u >> 8) % 2001) / 1000.f - 1.f, (float)((i * 40503u >> 4) % 2001) / 1000.f - 1.f);
    for (int t = 0; t < 4096; ++t) tw[t] = make_float2((float)cos(2 * kPi * t / 4096), (float)(-sin(2 * kPi * t / 4096)));
    float2 *din, *dout, *dtw;
    hipMalloc(&din, h.size() * 8); hipMalloc(&dout, h.size() * 8); hipMalloc(&dtw, 4096 * 8);
    hipMemcpy(din, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(dtw, tw.data(), 4096 * 8, hipMemcpyHostToDevice);
    // correctness: sequence 3, a few bins against a float64 DFT
    run<1, false, 1>(din, dout, dtw, nseq, nseq);
    std::vector<float2> o(4096);
    hipMemcpy(o.data(), dout + 3 * 4096, 4096 * 8, hipMemcpyDeviceToHost);
    double worst = 0, scale = 0;
    for (int k : {0, 1, 63, 64, 65, 1000, 2048, 4095}) {
        double re = 0, im = 0;
        for (int n = 0; n < 4096; ++n) { const double a = -2 * kPi * ((long)n * k % 4096) / 4096; re += h[3 * 4096 + n].x * cos(a) - h[3 * 4096 + n].y * sin(a); im += h[3 * 4096 + n].x * sin(a) + h[3 * 4096 + n].y * cos(a); }
        worst = fmax(worst, hypot(o[k].x - re, o[k].y - im)); scale = fmax(scale, hypot(re, im));
    }
    printf("max |err| over 8 bins %.3e (|X| up to %.1f)\n", worst, scale);
    float ms;
    const int big = 65536;
    ms = run<1, false, 1>(din, dout, dtw, big, 1024, 64);  printf("L2-resident data, 1 wave per workgroup (33 KB: 4 waves per CU): %.3f ms for %d transforms -> %.2f us per transform per SIMD, %.1f TFLOP/s nominal\n", ms, big, ms * 1e3 / (big / 1024.0), 5.0 * 4096 * 12 * big / ms / 1e9);
    ms = run<4, false, 1>(din, dout, dtw, big, 256, 64);   printf("same, 4 waves per workgroup: %.3f ms -> %.2f us per transform per SIMD\n", ms, ms * 1e3 / (big / 1024.0));
    ms = run<1, false, 1>(din, dout, dtw, nseq, 1024);  printf("HBM-streamed: %.3f ms for %d transforms (%.0f GB/s)\n", ms, nseq, 2.0 * nseq * 4096 * 8 / ms / 1e6);
    return 0;
}
