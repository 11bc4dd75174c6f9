// mst_fft.h - complex FFT of a power-of-two length held in LDS, for the spectrogram losses.
//
// STATUS (round 1): NOT used by the shipped kernels.  Measured on MI355X at BASELINE cfg #2 this register
// radix-8/16 formulation was SLOWER than the radix-4 Stockham of mst_stft.hip (fwd 94/60/40 us vs
// 64/43/45 us for n_fft 8192/2048/512): 210-256 VGPRs leave 2 waves per SIMD and the LDS latency of
// each pass is no longer hidden.  Kept as the starting point for a radix-8, 4-waves/SIMD variant.
//
// Stockham autosort (natural order in and out, ping-pong between two LDS buffers) with the
// butterflies done IN REGISTERS at radix 8 / 16: a 512-point transform is three radix-8 passes by one
// 64-lane wave, 2048 = 8*16*16 and 8192 = 2*16*16*16 by N/16 lanes - three to four LDS round trips
// instead of log4(N), and one twiddle multiply per point per pass.  Inter-pass twiddles are per-lane
// constants gathered once into registers (LaneTw).  LDS indices are padded (one slot per 32) so that the power-of-two strides of
// the autosort writes spread over the banks.
#pragma once
#include "mst_common.h"

namespace mst {

__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ float2 mul_mi(float2 a) { return make_float2(a.y, -a.x); }  // a * (-i)
__device__ __forceinline__ float2 mul_pi(float2 a) { return make_float2(-a.y, a.x); }  // a * (+i)

__device__ __forceinline__ int lds_pad(int i) { return i + (i >> 5); }
constexpr int lds_padded(int n) { return n + (n >> 5); }

// ---- forward (e^{-i...}) butterflies, natural order in place ---------------------------------------
__device__ __forceinline__ void fft2(float2& a, float2& b) {
    const float2 t = csub(a, b);
    a = cadd(a, b);
    b = t;
}
__device__ __forceinline__ void fft4(float2& v0, float2& v1, float2& v2, float2& v3) {
    const float2 s02 = cadd(v0, v2), d02 = csub(v0, v2), s13 = cadd(v1, v3), d13 = csub(v1, v3);
    v0 = cadd(s02, s13);
    v1 = cadd(d02, mul_mi(d13));
    v2 = csub(s02, s13);
    v3 = cadd(d02, mul_pi(d13));
}
template <int R>
__device__ __forceinline__ void butterfly(float2* v);
template <>
__device__ __forceinline__ void butterfly<2>(float2* v) { fft2(v[0], v[1]); }
template <>
__device__ __forceinline__ void butterfly<4>(float2* v) { fft4(v[0], v[1], v[2], v[3]); }
template <>
__device__ __forceinline__ void butterfly<8>(float2* v) {
    constexpr float c = 0.70710678118654752f;
    float2 e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6], o0 = v[1], o1 = v[3], o2 = v[5], o3 = v[7];
    fft4(e0, e1, e2, e3);
    fft4(o0, o1, o2, o3);
    o1 = make_float2(c * (o1.x + o1.y), c * (o1.y - o1.x));    // * W8^1 = (c, -c)
    o2 = mul_mi(o2);                                           // * W8^2 = -i
    o3 = make_float2(c * (o3.y - o3.x), -c * (o3.x + o3.y));   // * W8^3 = (-c, -c)
    v[0] = cadd(e0, o0); v[4] = csub(e0, o0);
    v[1] = cadd(e1, o1); v[5] = csub(e1, o1);
    v[2] = cadd(e2, o2); v[6] = csub(e2, o2);
    v[3] = cadd(e3, o3); v[7] = csub(e3, o3);
}
template <>
__device__ __forceinline__ void butterfly<16>(float2* v) {
    // 4 x 4: F[n2] = DFT4 over n1 of x[4 n1 + n2]; X[k1 + 4 k2] = DFT4 over n2 of W16^(n2 k1) F[n2][k1]
    constexpr float c1 = 0.92387953251128674f, s1 = 0.38268343236508977f;  // cos, sin (pi/8)
    constexpr float c2 = 0.70710678118654752f;
#pragma unroll
    for (int n2 = 0; n2 < 4; ++n2) fft4(v[n2], v[n2 + 4], v[n2 + 8], v[n2 + 12]);  // F[n2][k1] now in v[n2 + 4 k1]
    // twiddles W16^m = (cos(pi m/8), -sin(pi m/8)) for m = n2*k1
    v[5] = cmul(v[5], make_float2(c1, -s1));    // n2 = 1, k1 = 1 : m = 1
    v[6] = cmul(v[6], make_float2(c2, -c2));    // n2 = 2, k1 = 1 : m = 2
    v[7] = cmul(v[7], make_float2(s1, -c1));    // n2 = 3, k1 = 1 : m = 3
    v[9] = cmul(v[9], make_float2(c2, -c2));    // n2 = 1, k1 = 2 : m = 2
    v[10] = mul_mi(v[10]);                      // n2 = 2, k1 = 2 : m = 4
    v[11] = cmul(v[11], make_float2(-c2, -c2)); // n2 = 3, k1 = 2 : m = 6
    v[13] = cmul(v[13], make_float2(s1, -c1));  // n2 = 1, k1 = 3 : m = 3
    v[14] = cmul(v[14], make_float2(-c2, -c2)); // n2 = 2, k1 = 3 : m = 6
    v[15] = cmul(v[15], make_float2(-c1, s1));  // n2 = 3, k1 = 3 : m = 9
    float2 out[16];
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) {
        float2 g0 = v[4 * k1], g1 = v[4 * k1 + 1], g2 = v[4 * k1 + 2], g3 = v[4 * k1 + 3];
        fft4(g0, g1, g2, g3);
        out[k1] = g0; out[k1 + 4] = g1; out[k1 + 8] = g2; out[k1 + 12] = g3;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = out[i];
}

// radix plan: N = R0 * R1 * R2 (* R3), small leading radix first (its Ns = 1 pass needs no twiddles)
template <int N> struct FftPlan;
template <> struct FftPlan<128>  { static constexpr int n = 2, r[4] = {8, 16, 1, 1},  tpf = 64; };
template <> struct FftPlan<256>  { static constexpr int n = 2, r[4] = {16, 16, 1, 1}, tpf = 64; };
template <> struct FftPlan<512>  { static constexpr int n = 3, r[4] = {8, 8, 8, 1},   tpf = 64; };
template <> struct FftPlan<1024> { static constexpr int n = 3, r[4] = {4, 16, 16, 1}, tpf = 64; };
template <> struct FftPlan<2048> { static constexpr int n = 3, r[4] = {8, 16, 16, 1}, tpf = 128; };
template <> struct FftPlan<4096> { static constexpr int n = 3, r[4] = {16, 16, 16, 1}, tpf = 256; };
template <> struct FftPlan<8192> { static constexpr int n = 4, r[4] = {2, 16, 16, 16}, tpf = 512; };

// ---- per-lane twiddles, held in registers for the lifetime of the kernel ----------------------------
// In passes 1..3 every lane owns exactly ONE butterfly, always the same one, so its R-1 inter-pass
// twiddles w^t = exp(-2 pi i t k / (Ns R)) are frame-independent: they are gathered once from the
// exactly-rounded global table (cos, -sin)(2 pi t / N) - no twiddle arithmetic and no table lookups
// inside the transform, and 0.5-ulp twiddles.
template <int N>
struct LaneTw {
    float2 w[3][15];
    __device__ __forceinline__ void init(const float2* __restrict__ tw, int lane) {
        using P = FftPlan<N>;
        int Ns = P::r[0];
#pragma unroll
        for (int p = 1; p < 4; ++p) {
            const int R = P::r[p];
            if (R > 1) {
                const int Q = N / R;
                const int k = lane & (Ns - 1);
                const int step = N / (Ns * R);
#pragma unroll
                for (int t = 1; t < 16; ++t)
                    if (t < R) w[p - 1][t - 1] = (lane < Q) ? tw[(t * k * step) & (N - 1)] : make_float2(1.f, 0.f);
                Ns *= R;
            }
        }
    }
};

// ---- one Stockham pass of radix R over an N-point signal; `lane` in [0, TPF) ---------------------------
// wl: this lane's R-1 twiddles (used when the lane owns a single butterfly of a pass with Ns > 1)
template <int N, int R, int TPF>
__device__ __forceinline__ void stockham_pass(const float2* __restrict__ in, float2* __restrict__ out, int Ns, const float2* wl,
                                              int lane) {
    constexpr int Q = N / R;
    static_assert(Q <= TPF || R <= 8, "twiddled passes must be one butterfly per lane");
#pragma unroll
    for (int j = lane; j < Q; j += TPF) {
        const int k = j & (Ns - 1);
        float2 v[R];
#pragma unroll
        for (int t = 0; t < R; ++t) v[t] = in[lds_pad(j + t * Q)];
        if (Ns > 1) {
#pragma unroll
            for (int t = 1; t < R; ++t) v[t] = cmul(v[t], wl[t - 1]);
        }
        butterfly<R>(v);
        const int base = (j - k) * R + k;
#pragma unroll
        for (int t = 0; t < R; ++t) out[lds_pad(base + t * Ns)] = v[t];
    }
}

template <int N, int P_, int TPF>
__device__ __forceinline__ void pass_dispatch(float2*& a, float2*& b, int& Ns, const LaneTw<N>& T, int lane) {
    constexpr int R = FftPlan<N>::r[P_];
    if (R > 1) {
        stockham_pass<N, (R > 1 ? R : 2), TPF>(a, b, Ns, T.w[P_ > 0 ? P_ - 1 : 0], lane);
        float2* t = a; a = b; b = t;
        Ns *= R;
        __syncthreads();
    }
}
// Forward DFT of the N points in `a` (padded indexing), ping-pong with `b`; returns the result buffer.
// All threads of the workgroup must call it (it contains workgroup barriers); `lane` = thread index inside
// the group of TPF lanes that owns this transform.
template <int N>
__device__ __forceinline__ float2* lds_fft(float2* a, float2* b, const LaneTw<N>& T, int lane) {
    constexpr int TPF = FftPlan<N>::tpf;
    int Ns = 1;
    pass_dispatch<N, 0, TPF>(a, b, Ns, T, lane);
    pass_dispatch<N, 1, TPF>(a, b, Ns, T, lane);
    pass_dispatch<N, 2, TPF>(a, b, Ns, T, lane);
    pass_dispatch<N, 3, TPF>(a, b, Ns, T, lane);
    return a;
}

}  // namespace mst
