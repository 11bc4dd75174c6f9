// Developer micro-benchmark: the in-wave 512-point transform (mst_wfft.h: exchanges by permlane swaps / DPP) against the round-2
// engine's 512-point transform (mst_fft2.h: exchanges through LDS).  Both iterate REPS transforms per wave on register data.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -o /tmp/wf512 tools/ubench/wf512.hip && /tmp/wf512
#include "wfft_inwave.h"
#include <cmath>
#include <cstdio>
#include <vector>
using namespace mst;

template <int REPS>
__global__ __launch_bounds__(256) void k_wf(const float2* __restrict__ in, float2* __restrict__ out, const float2* __restrict__ twg) {
    const int lane = threadIdx.x & 63, wid = blockIdx.x * 4 + (threadIdx.x >> 6);
    WfTw tw;
    tw.init<512>(twg, lane);
    float2 v[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) v[t] = in[(size_t)(wid & 63) * 512 + lane + 64 * t];
    for (int rep = 0; rep < REPS; ++rep) {
        wf512(v, tw, lane);
        if (REPS > 1) {
#pragma unroll
            for (int t = 0; t < 8; ++t) v[t] = make_float2(v[t].x * 0.05f, v[t].y * 0.05f);
        }
    }
#pragma unroll
    for (int d = 0; d < 8; ++d) out[(size_t)wid * 512 + wf_kfreq(lane, d)] = v[d];
}

template <int REPS>
__global__ __launch_bounds__(256) void k_lds(const float2* __restrict__ in, float2* __restrict__ out, const float2* __restrict__ twg) {
    using S = FftShape<512>;
    __shared__ __attribute__((aligned(16))) float2 bufs[4][S::SLOTS];
    const int lane = threadIdx.x & 63, wid = blockIdx.x * 4 + (threadIdx.x >> 6);
    float2* buf = bufs[threadIdx.x >> 6];
    LaneTw<512> tw;
    tw.init(twg, lane);
    float2 v[8], o[1][8];
#pragma unroll
    for (int t = 0; t < 8; ++t) v[t] = in[(size_t)(wid & 63) * 512 + lane + 64 * t];
    for (int rep = 0; rep < REPS; ++rep) {
        fft_run<512>(v, o, buf, tw, lane);
        group_lds_sync<64>();
#pragma unroll
        for (int t = 0; t < 8; ++t) v[t] = REPS > 1 ? make_float2(o[0][t].x * 0.05f, o[0][t].y * 0.05f) : o[0][t];
    }
#pragma unroll
    for (int d = 0; d < 8; ++d) out[(size_t)wid * 512 + lane + 64 * d] = v[d];
}

template <typename K>
static float run(K kern, const float2* in, float2* out, const float2* tw, int blocks) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(blocks / 4), dim3(256), 0, 0, in, out, tw);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    return best;
}

int main() {
    const double kPi = 3.14159265358979323846;
    const int nseq = 8192;
    std::vector<float2> h((size_t)64 * 512), tw(512);
    for (size_t i = 0; i < h.size(); ++i) h[i] = make_float2((float)((i * 2654435761u >> 8) % 2001) / 1000.f - 1.f, (float)((i * 40503u >> 4) % 2001) / 1000.f - 1.f);
    for (int t = 0; t < 512; ++t) tw[t] = make_float2((float)cos(2 * kPi * t / 512), (float)(-sin(2 * kPi * t / 512)));
    float2 *din, *dout, *dtw;
    hipMalloc(&din, h.size() * 8); hipMalloc(&dout, (size_t)nseq * 512 * 8); hipMalloc(&dtw, 512 * 8);
    hipMemcpy(din, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(dtw, tw.data(), 512 * 8, hipMemcpyHostToDevice);
    for (int which = 0; which < 2; ++which) {
        if (which == 0) run(k_wf<1>, din, dout, dtw, 64); else run(k_lds<1>, din, dout, dtw, 64);
        std::vector<float2> o(512);
        hipMemcpy(o.data(), dout + 5 * 512, 512 * 8, hipMemcpyDeviceToHost);
        double worst = 0, scale = 0;
        for (int k = 0; k < 512; ++k) {
            double re = 0, im = 0;
            for (int n = 0; n < 512; ++n) { const double a = -2 * kPi * ((long)n * k % 512) / 512; re += h[5 * 512 + n].x * cos(a) - h[5 * 512 + n].y * sin(a); im += h[5 * 512 + n].x * sin(a) + h[5 * 512 + n].y * cos(a); }
            worst = fmax(worst, hypot(o[k].x - re, o[k].y - im)); scale = fmax(scale, hypot(re, im));
        }
        printf("%s: max |err| over 512 bins %.3e (|X| up to %.1f)\n", which ? "lds engine" : "in-wave  ", worst, scale);
    }
    constexpr int R = 2048;
    for (int wps : {1, 2, 4, 8}) {
        const int blocks = 1024 * wps;  // wps waves per SIMD
        const float a = run(k_wf<R>, din, dout, dtw, blocks), b = run(k_lds<R>, din, dout, dtw, blocks);
        printf("%d waves/SIMD: in-wave %.1f us, lds engine %.1f us  (%d transforms per wave; per transform per SIMD: %.0f vs %.0f cycles at 2.4 GHz)\n",
               wps, a * 1e3, b * 1e3, R, a * 1e-3 * 2.4e9 / (R * wps), b * 1e-3 * 2.4e9 / (R * wps));
    }
    return 0;
}
