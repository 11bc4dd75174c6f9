// mst_fft3.h - round-3 FFT engine: ONE WAVE per 2048-point transform, 32 points per lane in registers.
//
// The round-2 engine (mst_fft2.h) spreads a frame over 256 lanes x 8 points: four radix passes with three exchanges through LDS and a
// workgroup barrier around each - profiles/round2_counters.md has these kernels at 33-36 % vector issue, the rest is waiting.
// Here N = 32 x 64:
//   phase 1   lane n2 (0..63) holds x[64 n1 + n2], n1 = 0..31: an FFT-32 entirely in registers (4 x 8, compile-time twiddles),
//             then the inter-stage twiddle W_N^(n2 k1) as the product of two exactly rounded per-lane factors;
//   exchange  ONE transpose through LDS (32 rows k1 x 64 columns n2) - a single wave, so no barrier, only the wave's own
//             LDS ordering;
//   phase 2   row k1 is owned by the lane pair (2 k1, 2 k1 + 1): lane q of the pair takes n2 = 32 q + m (m = 0..31, 16 contiguous
//             16-byte reads), the pair's radix-2 butterfly is one DPP swap per point, and a second register FFT-32 finishes:
//             X[k1 + 32 j + 64 kappa] ends in register kappa of lane (k1, j = q).
// Same flop count as before, one LDS round trip instead of three, no barriers, and 2000 independent instructions per wave for
// the scheduler to hide latency with.
#pragma once
#include "mst_fft2.h"

namespace mst {

constexpr int kF3Pitch = 66;  // float2 per LDS row of the exchange: 528 bytes, 16 consecutive rows start in 16 different 16-byte slots

struct W32T { float c[32], s[32]; };
constexpr double f3_pi = 3.14159265358979323846;
constexpr double f3_cos(double x) { double t = 1, s = 1; for (int i = 1; i < 16; ++i) { t *= -x * x / ((2 * i - 1) * (2 * i)); s += t; } return s; }
constexpr double f3_sin(double x) { double t = x, s = x; for (int i = 1; i < 16; ++i) { t *= -x * x / ((2 * i) * (2 * i + 1)); s += t; } return s; }
template <int D>
constexpr W32T make_wtab() {  // (cos, -sin)(2 pi m / D), m < 32, argument reduced to (-pi, pi]
    W32T w{};
    for (int m = 0; m < 32; ++m) {
        double a = 2 * f3_pi * m / D;
        if (a > f3_pi) a -= 2 * f3_pi;
        w.c[m] = (float)f3_cos(a);
        w.s[m] = (float)(-f3_sin(a));
    }
    return w;
}
__device__ constexpr W32T kW32 = make_wtab<32>();
__device__ constexpr W32T kW64h = make_wtab<64>();  // W_64^m, m < 32: the pair butterfly's twiddles

// in-register FFT-32 (4 x 8), in place: v[n] natural on entry; on exit X[ka + 4 kb] sits in v[8 ka + kb]
__device__ __forceinline__ void fft32(float2 (&v)[32]) {
#pragma unroll
    for (int b = 0; b < 8; ++b) {  // radix 4 over a (n = 8 a + b), then W_32^(b ka)
        fft4(v[b], v[8 + b], v[16 + b], v[24 + b]);
#pragma unroll
        for (int ka = 1; ka < 4; ++ka) {
            const int m = (b * ka) & 31;
            if (m != 0) v[8 * ka + b] = cmul(v[8 * ka + b], make_float2(kW32.c[m], kW32.s[m]));
        }
    }
#pragma unroll
    for (int ka = 0; ka < 4; ++ka) butterfly<8>(&v[8 * ka]);  // radix 8 over b: X[ka + 4 kb]
}
__device__ __forceinline__ constexpr int dig32(int k) { return 8 * (k & 3) + (k >> 2); }  // register of X[k] after fft32

// per-lane inter-stage twiddles W_2048^(lane k1), k1 = 8 a + b: two sets of exactly rounded table entries
struct F3Tw {
    float2 a[4], b[8];
    __device__ __forceinline__ void init(const float2* __restrict__ tw /* (cos, -sin)(2 pi t / 2048) */, int lane) {
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = tw[(lane * 8 * i) & 2047];
#pragma unroll
        for (int i = 0; i < 8; ++i) b[i] = tw[(lane * i) & 2047];
    }
};
__device__ __forceinline__ float dpp_swap1(float x) {  // value of the neighbouring lane (lane ^ 1): quad_perm [1, 0, 3, 2]
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xf, 0xf, true));
}

// 2048-point forward transform of the wave's frame.  v[t] = element lane + 64 t (t < 32).  On return bin k1 + 32 j + 64 kappa is in
// v[dig32(kappa)] of lane 2 k1 + j.  buf: 32 x kF3Pitch float2 of LDS owned by this wave; free again on return.
__device__ __forceinline__ void fft2048_wave(float2 (&v)[32], float2* __restrict__ buf, const F3Tw& tw, int lane) {
    fft32(v);  // over n1: Y[k1] in v[dig32(k1)]
#pragma unroll
    for (int k1 = 0; k1 < 32; ++k1) {
        float2 y = v[dig32(k1)];
        if (k1 != 0) {
            const float2 w = (k1 & 7) == 0 ? tw.a[k1 >> 3] : ((k1 >> 3) == 0 ? tw.b[k1 & 7] : cmul(tw.a[k1 >> 3], tw.b[k1 & 7]));
            y = cmul(y, w);
        }
        buf[k1 * kF3Pitch + lane] = y;
    }
    wave_lds_sync();
    const int k1 = lane >> 1, q = lane & 1;
#pragma unroll
    for (int m = 0; m < 32; m += 2) {
        const float4 z = *reinterpret_cast<const float4*>(&buf[k1 * kF3Pitch + 32 * q + m]);
        v[m] = make_float2(z.x, z.y);
        v[m + 1] = make_float2(z.z, z.w);
    }
    wave_lds_sync();
    // pair butterfly over q: lane j keeps u_j[m] = Z[m] + (-1)^j Z[32 + m], times W_64^(m j)
#pragma unroll
    for (int m = 0; m < 32; ++m) {
        const float2 o = make_float2(dpp_swap1(v[m].x), dpp_swap1(v[m].y));
        const float2 u = q ? csub(o, v[m]) : cadd(v[m], o);
        v[m] = (q && m != 0) ? cmul(u, make_float2(kW64h.c[m], kW64h.s[m])) : u;
    }
    fft32(v);  // over m: X[k1 + 32 (q + 2 kappa)] in v[dig32(kappa)]
}

}  // namespace mst
