// mst_stft.hip - multi-resolution STFT loss (auraloss 0.4.0 semantics; reference
// configuration configs/models/naive.yaml:54-68, call site mst/system.py:331).
//
//   per resolution: frames of torch.stft(center=True, reflect pad, periodic Hann), hop = n_fft/2..n_fft/4
//   mag = sqrt(max(re^2+im^2, 1e-8));  SC = ||ym - xm||_F / ||ym||_F (per row or global);
//   logmag = mean |log xm - log ym|;  loss = mean over resolutions of w_sc SC + w_log logmag + w_lin L1.
//
// Kernel shape: one workgroup owns a strip of consecutive frames of one (batch, channel) row.  A
// frame of the prediction and the same frame of the target are transformed by ONE complex FFT
// (z = w*(x + i y)) living in LDS (mixed radix-2/4 decimation-in-time, digit-reversed on load),
// the two spectra are separated by Hermitian symmetry and the magnitude / log / partial-sum
// epilogue is fused - no spectrogram ever reaches HBM.  Backward recomputes the forward tile,
// forms dL/dX, and returns two frames' time-domain cotangents per complex FFT; the overlap-add
// (incl. the reflect-padding fold-back) uses hardware float atomics into grad_pred.
#include "mst_kernels.h"
#include "mst_stft.h"
#ifndef MST_STFT_UNROLL_STAGES
#define MST_STFT_UNROLL_STAGES 1  // stage loops of the transforms unrolled so that the stage constants fold (386 -> 368 us; 0 = rolled, for A/B)
#endif

namespace mst {

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }

// Forward DFT (e^{-i...}) of NFFT complex points held in LDS: Stockham autosort, radix-4 stages
// (one radix-2 stage first when log2 NFFT is odd), natural order in AND out, ping-pong between
// `a` (input) and `b`; the result pointer is returned.  Every stage reads a[j + r n/4] for
// consecutive j (conflict-free).  Twiddles come from a two-level table staged in LDS
// (w^t = coarse[t >> 6] * fine[t & 63], 1.5 KB instead of 64 KB at n = 8192) so that no stage waits
// on a global load; w^2t and w^3t are formed by multiplication.
template <int NFFT>
struct Twiddles {
    float2 coarse[NFFT / 64];
    float2 fine[64];
};
template <int NFFT>
__device__ __forceinline__ void stage_twiddles(Twiddles<NFFT>& T, const float2* __restrict__ tw, int tid, int nthreads) {
    for (int i = tid; i < NFFT / 64; i += nthreads) T.coarse[i] = tw[i * 64];
    for (int i = tid; i < 64; i += nthreads) T.fine[i] = tw[i];
}
// R2_DONE: the caller already applied the leading radix-2 stage while loading (load_frame<..., true>)
template <int NFFT, int THREADS, bool R2_DONE = false>
__device__ __forceinline__ float2* lds_fft(float2* a, float2* b, const Twiddles<NFFT>& T, int tid) {
    constexpr int LOG2N = (NFFT == 128) ? 7 : (NFFT == 256) ? 8 : (NFFT == 512) ? 9 : (NFFT == 1024) ? 10
                        : (NFFT == 2048) ? 11 : (NFFT == 4096) ? 12 : 13;
    int Ns = ((LOG2N & 1) && R2_DONE) ? 2 : 1;
    if ((LOG2N & 1) && !R2_DONE) {
        constexpr int h = NFFT >> 1;
#pragma unroll
        for (int j = tid; j < h; j += THREADS) {
            const float2 u0 = a[j], u1 = a[j + h];
            b[2 * j] = make_float2(u0.x + u1.x, u0.y + u1.y);
            b[2 * j + 1] = make_float2(u0.x - u1.x, u0.y - u1.y);
        }
        float2* t = a; a = b; b = t;
        Ns = 2;
        __syncthreads();
    }
    constexpr int q = NFFT >> 2;
#if MST_STFT_UNROLL_STAGES
#pragma unroll
#else
#pragma unroll 1
#endif
    for (; Ns < NFFT; Ns <<= 2) {
        const int tstep = NFFT / (4 * Ns);
#pragma unroll
        for (int j = tid; j < q; j += THREADS) {
            const int k = j & (Ns - 1);
            float2 u0 = a[j], u1 = a[j + q], u2 = a[j + 2 * q], u3 = a[j + 3 * q];
            if (Ns > 1) {
                const int t = k * tstep;
                const float2 w1 = cmul(T.coarse[t >> 6], T.fine[t & 63]);
                const float2 w2 = cmul(w1, w1);
                const float2 w3 = cmul(w2, w1);
                u1 = cmul(u1, w1);
                u2 = cmul(u2, w2);
                u3 = cmul(u3, w3);
            }
            const float2 s02 = make_float2(u0.x + u2.x, u0.y + u2.y), d02 = make_float2(u0.x - u2.x, u0.y - u2.y);
            const float2 s13 = make_float2(u1.x + u3.x, u1.y + u3.y), d13 = make_float2(u1.x - u3.x, u1.y - u3.y);
            const int base = ((j - k) << 2) + k;
            b[base] = make_float2(s02.x + s13.x, s02.y + s13.y);
            b[base + Ns] = make_float2(d02.x + d13.y, d02.y - d13.x);  // d02 - i d13
            b[base + 2 * Ns] = make_float2(s02.x - s13.x, s02.y - s13.y);
            b[base + 3 * Ns] = make_float2(d02.x - d13.y, d02.y + d13.x);  // d02 + i d13
        }
        float2* t = a; a = b; b = t;
        __syncthreads();
    }
    return a;
}

// ---- in-place variant (n_fft = 8192): ONE LDS buffer, so two workgroups share a CU ------------------
// Measured on MI355X: the ping-pong 8192 kernels (1 workgroup of 1024 lanes per CU) lose only ~20 % per
// workgroup when run with 512 lanes - they wait on barriers and LDS latency, not on throughput - so two
// independent 512-lane workgroups per CU overlap each other's stalls.
// (The forward kernel did not gain - 69-78 vs 62 us - and stays on the ping-pong transform by default.)
//   fft_dif: decimation in frequency, natural order in, digit-reversed out (freq_pos() gives the slot of bin k)
//   fft_dit: decimation in time, digit-reversed in (same map), natural order out
// A leading (DIF) / trailing (DIT) radix-2 stage handles odd log2 n; the rest is radix-4.  Slots are bank-
// swizzled (lds_swz) so that digit-reversed neighbours - consecutive bins - land in different banks.
constexpr int ilog2(int n) { return n <= 1 ? 0 : 1 + ilog2(n >> 1); }
template <int LOG2N>
__device__ __forceinline__ int lds_swz(int a) { return a ^ ((a >> (LOG2N - 5)) & 31); }
template <int DIGITS>
__device__ __forceinline__ int rev4(int k) {  // reverse DIGITS base-4 digits
    int r = 0;
#pragma unroll
    for (int i = 0; i < DIGITS; ++i) {
        r = (r << 2) | (k & 3);
        k >>= 2;
    }
    return r;
}
// slot (before swizzling) of bin k after fft_dif<N>, = slot sample k must be put in before fft_dit<N>
template <int N>
__device__ __forceinline__ int freq_pos(int k) {
    constexpr int LG = ilog2(N);
    if (LG & 1) return (k & 1) * (N / 2) + rev4<LG / 2>(k >> 1);
    return rev4<LG / 2>(k);
}

template <int N, int THREADS, bool R2_DONE = false>  // R2_DONE: the loader applied the leading radix-2 stage
__device__ __forceinline__ void fft_dif(float2* buf, const Twiddles<N>& T, int tid) {
    constexpr int LG = ilog2(N);
    constexpr bool ODD = LG & 1;
    constexpr int M = ODD ? N / 2 : N;  // length of the radix-4 sub-transforms
    if (ODD && !R2_DONE) {
#pragma unroll
        for (int j = tid; j < M; j += THREADS) {
            const int p0 = lds_swz<LG>(j), p1 = lds_swz<LG>(j + M);
            const float2 a = buf[p0], b = buf[p1];
            buf[p0] = cadd(a, b);
            buf[p1] = cmul(csub(a, b), cmul(T.coarse[j >> 6], T.fine[j & 63]));
        }
        __syncthreads();
    }
#if MST_STFT_UNROLL_STAGES
#pragma unroll
#else
#pragma unroll 1
#endif
    for (int L = M / 4; L >= 1; L >>= 2) {
        const int tstep = N / (4 * L);
#pragma unroll
        for (int b = tid; b < N / 4; b += THREADS) {
            const int h = ODD ? b / (M / 4) : 0, bb = ODD ? b % (M / 4) : b;
            const int k = bb & (L - 1);
            const int i0 = h * M + ((bb - k) << 2) + k;
            const int p0 = lds_swz<LG>(i0), p1 = lds_swz<LG>(i0 + L), p2 = lds_swz<LG>(i0 + 2 * L), p3 = lds_swz<LG>(i0 + 3 * L);
            const float2 u0 = buf[p0], u1 = buf[p1], u2 = buf[p2], u3 = buf[p3];
            const float2 s02 = cadd(u0, u2), d02 = csub(u0, u2), s13 = cadd(u1, u3), d13 = csub(u1, u3);
            float2 y1 = make_float2(d02.x + d13.y, d02.y - d13.x);  // d02 - i d13
            float2 y2 = csub(s02, s13);
            float2 y3 = make_float2(d02.x - d13.y, d02.y + d13.x);  // d02 + i d13
            if (L > 1) {
                const int t = k * tstep;
                const float2 w1 = cmul(T.coarse[t >> 6], T.fine[t & 63]);
                const float2 w2 = cmul(w1, w1);
                y1 = cmul(y1, w1);
                y2 = cmul(y2, w2);
                y3 = cmul(y3, cmul(w2, w1));
            }
            buf[p0] = cadd(s02, s13);
            buf[p1] = y1;
            buf[p2] = y2;
            buf[p3] = y3;
        }
        __syncthreads();
    }
}
template <int N, int THREADS>
__device__ __forceinline__ void fft_dit(float2* buf, const Twiddles<N>& T, int tid) {
    constexpr int LG = ilog2(N);
    constexpr bool ODD = LG & 1;
    constexpr int M = ODD ? N / 2 : N;
#if MST_STFT_UNROLL_STAGES
#pragma unroll
#else
#pragma unroll 1
#endif
    for (int L = 1; L < M; L <<= 2) {
        const int tstep = N / (4 * L);
#pragma unroll
        for (int b = tid; b < N / 4; b += THREADS) {
            const int h = ODD ? b / (M / 4) : 0, bb = ODD ? b % (M / 4) : b;
            const int k = bb & (L - 1);
            const int i0 = h * M + ((bb - k) << 2) + k;
            const int p0 = lds_swz<LG>(i0), p1 = lds_swz<LG>(i0 + L), p2 = lds_swz<LG>(i0 + 2 * L), p3 = lds_swz<LG>(i0 + 3 * L);
            float2 u0 = buf[p0], u1 = buf[p1], u2 = buf[p2], u3 = buf[p3];
            if (L > 1) {
                const int t = k * tstep;
                const float2 w1 = cmul(T.coarse[t >> 6], T.fine[t & 63]);
                const float2 w2 = cmul(w1, w1);
                u1 = cmul(u1, w1);
                u2 = cmul(u2, w2);
                u3 = cmul(u3, cmul(w2, w1));
            }
            const float2 s02 = cadd(u0, u2), d02 = csub(u0, u2), s13 = cadd(u1, u3), d13 = csub(u1, u3);
            buf[p0] = cadd(s02, s13);
            buf[p1] = make_float2(d02.x + d13.y, d02.y - d13.x);
            buf[p2] = csub(s02, s13);
            buf[p3] = make_float2(d02.x - d13.y, d02.y + d13.x);
        }
        __syncthreads();
    }
    if (ODD) {
#pragma unroll
        for (int j = tid; j < M; j += THREADS) {
            const int p0 = lds_swz<LG>(j), p1 = lds_swz<LG>(j + M);
            const float2 a = buf[p0], b = cmul(buf[p1], cmul(T.coarse[j >> 6], T.fine[j & 63]));
            buf[p0] = cadd(a, b);
            buf[p1] = csub(a, b);
        }
        __syncthreads();
    }
}

__device__ __forceinline__ int64_t reflect_index(int64_t i, int64_t n) {
    if (i < 0) i = -i;
    if (i >= n) i = 2 * (n - 1) - i;
    return i;
}

// load frame f of (x, y) as z = w (x + i y), natural order.
// R2 (odd log2 n_fft): the transform's leading radix-2 stage is applied on the way in - points j and j + n/2
// are fetched together and (z_j + z_(j+n/2), z_j - z_(j+n/2)) stored at 2j, 2j+1 - one LDS pass and one barrier less.
template <bool R2>
__device__ __forceinline__ void load_frame(float2* buf, const float* __restrict__ x, const float* __restrict__ y,
                                           const float* __restrict__ win, int f, const ResInfo& r, int64_t n, int tid,
                                           int nthreads) {
    const int64_t start = (int64_t)f * r.hop - r.n_fft / 2;
    if (R2) {
        const int h = r.n_fft / 2;
        for (int j = tid; j < h; j += nthreads) {
            const int64_t i0 = reflect_index(start + j, n), i1 = reflect_index(start + j + h, n);
            const float w0 = win[j], w1 = win[j + h];
            const float2 u0 = make_float2(w0 * x[i0], w0 * y[i0]), u1 = make_float2(w1 * x[i1], w1 * y[i1]);
            *reinterpret_cast<float4*>(&buf[2 * j]) = make_float4(u0.x + u1.x, u0.y + u1.y, u0.x - u1.x, u0.y - u1.y);
        }
        return;
    }
    for (int k = tid; k < r.n_fft; k += nthreads) {
        const int64_t i = reflect_index(start + k, n);
        const float w = win[k];
        buf[k] = make_float2(w * x[i], w * y[i]);
    }
}

// bin k of a transform result: natural order (ping-pong kernels) or the swizzled digit-reversed slots of fft_dif
template <int N, bool INPLACE>
__device__ __forceinline__ float2 bin_at(const float2* buf, int k) {
    return INPLACE ? buf[lds_swz<ilog2(N)>(freq_pos<N>(k))] : buf[k];
}
// separate the two real spectra packed in one complex FFT
template <int N, bool INPLACE>
__device__ __forceinline__ void split_xy(const float2* buf, int k, float2& X, float2& Y) {
    const float2 zk = bin_at<N, INPLACE>(buf, k), zn = bin_at<N, INPLACE>(buf, (N - k) & (N - 1));
    X = make_float2(0.5f * (zk.x + zn.x), 0.5f * (zk.y - zn.y));
    Y = make_float2(0.5f * (zk.y + zn.y), -0.5f * (zk.x - zn.x));
}

#ifndef MST_STFT_T8192
#define MST_STFT_T8192 1024
#endif
constexpr int stft_threads(int n_fft) { return n_fft <= 512 ? 128 : (n_fft <= 2048 ? 512 : (n_fft <= 4096 ? 1024 : MST_STFT_T8192)); }


template <int NFFT>
__global__ __launch_bounds__(stft_threads(NFFT)) void k_stft_fwd(StftArgs a) {
    constexpr int THREADS = stft_threads(NFFT);
    constexpr bool kOddLog2 = ilog2(NFFT) & 1;
    __shared__ __attribute__((aligned(16))) float2 bufA[NFFT];
    __shared__ __attribute__((aligned(16))) float2 bufB[NFFT];
    __shared__ float red[16][4];
    __shared__ Twiddles<NFFT> twd;
    const int tid = threadIdx.x, row = blockIdx.y;
    const ResInfo r = a.r;
    stage_twiddles<NFFT>(twd, reinterpret_cast<const float2*>(a.tables + r.tw_off), tid, THREADS);
    const float* win = a.tables + r.win_off;
    const float* x = a.pred + (int64_t)row * a.n;
    const float* y = a.target + (int64_t)row * a.n;
    float s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f;
    const int f0 = blockIdx.x * r.frames_per_wg;
    for (int f = f0; f < f0 + r.frames_per_wg && f < r.n_frames; ++f) {
        load_frame<kOddLog2>(bufA, x, y, win, f, r, a.n, tid, THREADS);
        __syncthreads();
        const float2* Z = lds_fft<NFFT, THREADS, true>(bufA, bufB, twd, tid);
        for (int k = tid; k < r.n_bins; k += THREADS) {
            float2 X, Y;
            split_xy<NFFT, false>(Z, k, X, Y);
            const float xm = sqrtf(fmaxf(X.x * X.x + X.y * X.y, a.eps));
            const float ym = sqrtf(fmaxf(Y.x * Y.x + Y.y * Y.y, a.eps));
            const float d = ym - xm;
            s1 = fmaf(d, d, s1);
            s2 = fmaf(ym, ym, s2);
            s3 += fabsf(__builtin_amdgcn_logf(xm) - __builtin_amdgcn_logf(ym));  // log2; scaled by ln2 below
            s4 += fabsf(d);
        }
        __syncthreads();
    }
    s3 *= kLn2;
    const int wave = tid >> 6, lane = tid & 63;
    s1 = wave_sum(s1); s2 = wave_sum(s2); s3 = wave_sum(s3); s4 = wave_sum(s4);
    if (lane == 0) { red[wave][0] = s1; red[wave][1] = s2; red[wave][2] = s3; red[wave][3] = s4; }
    __syncthreads();
    if (tid < 4) {
        float v = 0.f;
        for (int w = 0; w < THREADS / 64; ++w) v += red[w][tid];
        a.part[((int64_t)row * gridDim.x + blockIdx.x) * 4 + tid] = v;
    }
}

// cotangent G[k] = dL/dX[k] of the prediction's half spectrum for the frame whose packed FFT is Z
template <int N, bool INPLACE>
__device__ __forceinline__ float2 spectrum_cotangent(const float2* Z, int k, const StftArgs& a, const float* coef) {
    float2 X, Y;
    split_xy<N, INPLACE>(Z, k, X, Y);
    const float p2 = X.x * X.x + X.y * X.y;
    const float xm = sqrtf(fmaxf(p2, a.eps));
    const float ym = sqrtf(fmaxf(Y.x * Y.x + Y.y * Y.y, a.eps));
    float g = coef[0] * (xm - ym);
    const float dl = __builtin_amdgcn_logf(xm) - __builtin_amdgcn_logf(ym);
    g += coef[1] * ((dl > 0.f) - (dl < 0.f)) / xm;
    g += coef[2] * ((xm > ym) - (xm < ym));
    const float s = (p2 >= a.eps) ? g / xm : 0.0f;  // through sqrt(clamp(|X|^2, eps)): zero below the clamp
    return make_float2(s * X.x, s * X.y);
}

// Backward.  The real cotangent frame is Re IDFT of the half spectrum G, i.e. the IDFT of its
// Hermitian extension He (He[k] = G[k]/2, He[N-k] = conj(G[k])/2, real at k = 0, N/2), and
// IDFT(h) = conj(FFT(conj(h))).
//   PAIR  (n_fft <= 4096, three LDS buffers): two frames share one complex inverse FFT (He_a + i He_b).
//   !PAIR (n_fft = 8192, two LDS buffers): the real-output inverse is done with a HALF-size complex FFT:
//         y_even + i y_odd = IDFT_M(A + i Bq),  A[k] = H[k] + conj(H[M-k]),  Bq[k] = (H[k] - conj(H[M-k])) conj(W_N^k).
template <int NFFT, bool PAIR>
__global__ __launch_bounds__(stft_threads(NFFT)) void k_stft_bwd(StftArgs a) {
    constexpr int THREADS = stft_threads(NFFT);
    constexpr int M = NFFT / 2;
    __shared__ __attribute__((aligned(16))) float2 bufA[NFFT];
    __shared__ __attribute__((aligned(16))) float2 bufB[NFFT];
    __shared__ __attribute__((aligned(16))) float2 bufH[PAIR ? NFFT : 1];
    __shared__ Twiddles<NFFT> twd;
    __shared__ Twiddles<PAIR ? 64 : M> twh;  // half-size transform's twiddles (!PAIR only)
    const ResInfo r = a.r;
    const int tid = threadIdx.x, row = blockIdx.y;
    const float2* twg = reinterpret_cast<const float2*>(a.tables + r.tw_off);
    stage_twiddles<NFFT>(twd, twg, tid, THREADS);
    if constexpr (!PAIR) {  // W_M^t = W_N^(2t)
        for (int i = tid; i < M / 64; i += THREADS) twh.coarse[i] = twg[2 * 64 * i];
        for (int i = tid; i < 64; i += THREADS) twh.fine[i] = twg[2 * i];
    }
    const float* win = a.tables + r.win_off;
    const float* x = a.pred + (int64_t)row * a.n;
    const float* y = a.target + (int64_t)row * a.n;
    const float gl = a.grad_loss[0];
    const float coef[3] = {a.coef[(int64_t)row * 4] * gl, a.coef[(int64_t)row * 4 + 1] * gl, a.coef[(int64_t)row * 4 + 2] * gl};
    float* gx = a.grad_pred + (int64_t)row * a.n;
    const int fa = PAIR ? 2 * blockIdx.x : blockIdx.x, fb = fa + 1;
    const bool have_b = PAIR && fb < r.n_frames;
    const int64_t sa = (int64_t)fa * r.hop - NFFT / 2, sb = (int64_t)fb * r.hop - NFFT / 2;

    constexpr bool kOddLog2 = ilog2(NFFT) & 1;
    load_frame<kOddLog2>(bufA, x, y, win, fa, r, a.n, tid, THREADS);
    __syncthreads();
    float2* Z = lds_fft<NFFT, THREADS, true>(bufA, bufB, twd, tid);
    float2* O = (Z == bufA) ? bufB : bufA;  // the buffer the forward result is NOT in
    if constexpr (PAIR) {
        float2* H = bufH;  // conj(He) is assembled here
        for (int k = tid; k <= NFFT / 2; k += THREADS) {
            const float2 G = spectrum_cotangent<NFFT, false>(Z, k, a, coef);
            const bool edge = (k == 0) || (k == NFFT / 2);
            // conj(He): He[k] = G/2 -> (Gx/2, -Gy/2); He[N-k] = conj(G)/2 -> (Gx/2, +Gy/2)
            H[k] = edge ? make_float2(G.x, 0.f) : make_float2(0.5f * G.x, -0.5f * G.y);
            if (!edge) H[NFFT - k] = make_float2(0.5f * G.x, 0.5f * G.y);
        }
        __syncthreads();
        if (have_b) {
            load_frame<kOddLog2>(bufA, x, y, win, fb, r, a.n, tid, THREADS);
            __syncthreads();
            Z = lds_fft<NFFT, THREADS, true>(bufA, bufB, twd, tid);
            for (int k = tid; k <= NFFT / 2; k += THREADS) {
                const float2 G = spectrum_cotangent<NFFT, false>(Z, k, a, coef);
                const bool edge = (k == 0) || (k == NFFT / 2);
                // conj(i He_b): i He[k] = i G/2 -> conj = (-Gy/2, -Gx/2);  i He[N-k] = i conj(G)/2 -> conj = (Gy/2, -Gx/2)
                if (edge) {
                    H[k].y -= G.x;
                } else {
                    H[k].x += -0.5f * G.y; H[k].y += -0.5f * G.x;
                    H[NFFT - k].x += 0.5f * G.y; H[NFFT - k].y += -0.5f * G.x;
                }
            }
            __syncthreads();
        }
        // FFT(conj(h)) = conj(r_a + i r_b)  =>  r_a = Re, r_b = -Im
        const float2* R = lds_fft<NFFT, THREADS>(H, bufA, twd, tid);
        for (int k = tid; k < NFFT; k += THREADS) {
            const float2 v = R[k];
            const float w = win[k];
            unsafeAtomicAdd(&gx[reflect_index(sa + k, a.n)], w * v.x);
            if (have_b) unsafeAtomicAdd(&gx[reflect_index(sb + k, a.n)], -w * v.y);
        }
    } else {
        // pair (k, M-k), k = 0..M/2, of the first-half Hermitian spectrum H[0..M] (H[k] = G[k]/2 inside, real at the ends)
        for (int k = tid; k <= M / 2; k += THREADS) {
            float2 Hk = spectrum_cotangent<NFFT, false>(Z, k, a, coef), Hm = spectrum_cotangent<NFFT, false>(Z, M - k, a, coef);
            if (k == 0) {
                // V[0] = (H0 + HM) + i (H0 - HM), both real; store conj
                O[0] = make_float2(Hk.x + Hm.x, -(Hk.x - Hm.x));
            } else {
                Hk = make_float2(0.5f * Hk.x, 0.5f * Hk.y);
                Hm = make_float2(0.5f * Hm.x, 0.5f * Hm.y);
                const float2 w = cmul(twd.coarse[k >> 6], twd.fine[k & 63]);  // W_N^k ; W_N^(M-k) = -conj(W_N^k)
                const float2 Ak = make_float2(Hk.x + Hm.x, Hk.y - Hm.y);                                  // H[k] + conj(H[M-k])
                const float2 Bk = cmul(make_float2(Hk.x - Hm.x, Hk.y + Hm.y), make_float2(w.x, -w.y));    // (..)(conj W^k)
                O[k] = make_float2(Ak.x - Bk.y, -(Ak.y + Bk.x));                                         // conj(A + i Bq)
                if (k != M / 2) {
                    const float2 Am = make_float2(Hm.x + Hk.x, Hm.y - Hk.y);
                    const float2 Bm = cmul(make_float2(Hm.x - Hk.x, Hm.y + Hk.y), make_float2(-w.x, -w.y));
                    O[M - k] = make_float2(Am.x - Bm.y, -(Am.y + Bm.x));
                }
            }
        }
        __syncthreads();
        // half-size forward FFT of conj(V): result = conj(y_even + i y_odd)
        const float2* R = lds_fft<M, THREADS>(O, Z, twh, tid);
        for (int m = tid; m < M; m += THREADS) {
            const float2 v = R[m];
            unsafeAtomicAdd(&gx[reflect_index(sa + 2 * m, a.n)], win[2 * m] * v.x);
            unsafeAtomicAdd(&gx[reflect_index(sa + 2 * m + 1, a.n)], -win[2 * m + 1] * v.y);
        }
    }
}

// ---- the same two kernels on the in-place transform (used for n_fft = 8192) -------------------------
constexpr int kIpThreads = 512;

// frame f as z = w (x + i y) with fft_dif's leading radix-2 stage applied on the way in (n_fft with odd log2):
// slot j <- z_j + z_(j+n/2),  slot j + n/2 <- (z_j - z_(j+n/2)) W_n^j   (tw = the exactly rounded global table)
template <int NFFT>
__device__ __forceinline__ void load_frame_ip(float2* buf, const float* __restrict__ x, const float* __restrict__ y,
                                              const float* __restrict__ win, const float2* __restrict__ tw, int f,
                                              const ResInfo& r, int64_t n, int tid) {
    static_assert(ilog2(NFFT) & 1, "the fused radix-2 stage needs an odd log2 n_fft");
    constexpr int LG = ilog2(NFFT), h = NFFT / 2;
    const int64_t start = (int64_t)f * r.hop - NFFT / 2;
#pragma unroll 4
    for (int j = tid; j < h; j += kIpThreads) {
        const int64_t i0 = reflect_index(start + j, n), i1 = reflect_index(start + j + h, n);
        const float w0 = win[j], w1 = win[j + h];
        const float2 u0 = make_float2(w0 * x[i0], w0 * y[i0]), u1 = make_float2(w1 * x[i1], w1 * y[i1]);
        buf[lds_swz<LG>(j)] = cadd(u0, u1);
        buf[lds_swz<LG>(j + h)] = cmul(csub(u0, u1), tw[j]);
    }
}

template <int NFFT>
__global__ __launch_bounds__(kIpThreads) void k_stft_fwd_ip(StftArgs a) {
    constexpr int THREADS = kIpThreads;
    __shared__ __attribute__((aligned(16))) float2 buf[NFFT];
    __shared__ float red[THREADS / 64][4];
    __shared__ Twiddles<NFFT> twd;
    const int tid = threadIdx.x, row = blockIdx.y;
    const ResInfo r = a.r;
    stage_twiddles<NFFT>(twd, reinterpret_cast<const float2*>(a.tables + r.tw_off), tid, THREADS);
    const float* win = a.tables + r.win_off;
    const float* x = a.pred + (int64_t)row * a.n;
    const float* y = a.target + (int64_t)row * a.n;
    float s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f;
    const int f0 = blockIdx.x * r.frames_per_wg;
    for (int f = f0; f < f0 + r.frames_per_wg && f < r.n_frames; ++f) {
        load_frame_ip<NFFT>(buf, x, y, win, reinterpret_cast<const float2*>(a.tables + r.tw_off), f, r, a.n, tid);
        __syncthreads();
        fft_dif<NFFT, THREADS, true>(buf, twd, tid);
        for (int k = tid; k < r.n_bins; k += THREADS) {
            float2 X, Y;
            split_xy<NFFT, true>(buf, k, X, Y);
            const float xm = sqrtf(fmaxf(X.x * X.x + X.y * X.y, a.eps));
            const float ym = sqrtf(fmaxf(Y.x * Y.x + Y.y * Y.y, a.eps));
            const float d = ym - xm;
            s1 = fmaf(d, d, s1);
            s2 = fmaf(ym, ym, s2);
            s3 += fabsf(__builtin_amdgcn_logf(xm) - __builtin_amdgcn_logf(ym));  // log2; scaled by ln2 below
            s4 += fabsf(d);
        }
        __syncthreads();
    }
    s3 *= kLn2;
    const int wave = tid >> 6, lane = tid & 63;
    s1 = wave_sum(s1); s2 = wave_sum(s2); s3 = wave_sum(s3); s4 = wave_sum(s4);
    if (lane == 0) { red[wave][0] = s1; red[wave][1] = s2; red[wave][2] = s3; red[wave][3] = s4; }
    __syncthreads();
    if (tid < 4) {
        float v = 0.f;
        for (int w = 0; w < THREADS / 64; ++w) v += red[w][tid];
        a.part[((int64_t)row * gridDim.x + blockIdx.x) * 4 + tid] = v;
    }
}

// one frame per workgroup; the real cotangent frame comes out of a HALF-size complex transform (see k_stft_bwd, !PAIR)
template <int NFFT>
__global__ __launch_bounds__(kIpThreads) void k_stft_bwd_ip(StftArgs a) {
    constexpr int THREADS = kIpThreads;
    constexpr int M = NFFT / 2;
    constexpr int LGM = ilog2(M);
    constexpr int PER = (M / 2 + 1 + THREADS - 1) / THREADS;  // (k, M-k) pairs per lane
    __shared__ __attribute__((aligned(16))) float2 buf[NFFT];
    __shared__ Twiddles<NFFT> twd;
    __shared__ Twiddles<M> twh;
    const ResInfo r = a.r;
    const int tid = threadIdx.x, row = blockIdx.y;
    const float2* twg = reinterpret_cast<const float2*>(a.tables + r.tw_off);
    stage_twiddles<NFFT>(twd, twg, tid, THREADS);
    for (int i = tid; i < M / 64; i += THREADS) twh.coarse[i] = twg[2 * 64 * i];  // W_M^t = W_N^(2t)
    for (int i = tid; i < 64; i += THREADS) twh.fine[i] = twg[2 * i];
    const float* win = a.tables + r.win_off;
    const float* x = a.pred + (int64_t)row * a.n;
    const float* y = a.target + (int64_t)row * a.n;
    const float gl = a.grad_loss[0];
    const float coef[3] = {a.coef[(int64_t)row * 4] * gl, a.coef[(int64_t)row * 4 + 1] * gl, a.coef[(int64_t)row * 4 + 2] * gl};
    float* gx = a.grad_pred + (int64_t)row * a.n;
    const int fa = blockIdx.x;
    const int64_t sa = (int64_t)fa * r.hop - NFFT / 2;

    load_frame_ip<NFFT>(buf, x, y, win, twg, fa, r, a.n, tid);
    __syncthreads();
    fft_dif<NFFT, THREADS, true>(buf, twd, tid);
    // conj(A + i Bq) for the pairs (k, M-k), k = 0..M/2, kept in registers until every lane has read its bins
    float2 Vk[PER], Vm[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int k = tid + i * THREADS;
        Vk[i] = Vm[i] = make_float2(0.f, 0.f);
        if (k <= M / 2) {
            float2 Hk = spectrum_cotangent<NFFT, true>(buf, k, a, coef), Hm = spectrum_cotangent<NFFT, true>(buf, M - k, a, coef);
            if (k == 0) {
                Vk[i] = make_float2(Hk.x + Hm.x, -(Hk.x - Hm.x));  // V[0] = (H0 + HM) + i (H0 - HM), both real; conj
            } else {
                Hk = make_float2(0.5f * Hk.x, 0.5f * Hk.y);
                Hm = make_float2(0.5f * Hm.x, 0.5f * Hm.y);
                const float2 w = cmul(twd.coarse[k >> 6], twd.fine[k & 63]);  // W_N^k ; W_N^(M-k) = -conj(W_N^k)
                const float2 Ak = make_float2(Hk.x + Hm.x, Hk.y - Hm.y);
                const float2 Bk = cmul(make_float2(Hk.x - Hm.x, Hk.y + Hm.y), make_float2(w.x, -w.y));
                Vk[i] = make_float2(Ak.x - Bk.y, -(Ak.y + Bk.x));
                const float2 Am = make_float2(Hm.x + Hk.x, Hm.y - Hk.y);
                const float2 Bm = cmul(make_float2(Hm.x - Hk.x, Hm.y + Hk.y), make_float2(-w.x, -w.y));
                Vm[i] = make_float2(Am.x - Bm.y, -(Am.y + Bm.x));
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int k = tid + i * THREADS;
        if (k <= M / 2) {
            buf[lds_swz<LGM>(freq_pos<M>(k))] = Vk[i];
            if (k != 0 && k != M / 2) buf[lds_swz<LGM>(freq_pos<M>(M - k))] = Vm[i];
        }
    }
    __syncthreads();
    // half-size forward transform of conj(V): natural-order result = conj(y_even + i y_odd)
    fft_dit<M, THREADS>(buf, twh, tid);
    for (int m = tid; m < M; m += THREADS) {
        const float2 v = buf[lds_swz<LGM>(m)];
        unsafeAtomicAdd(&gx[reflect_index(sa + 2 * m, a.n)], win[2 * m] * v.x);
        unsafeAtomicAdd(&gx[reflect_index(sa + 2 * m + 1, a.n)], -win[2 * m + 1] * v.y);
    }
}


// ---- tables: twiddles (cos, -sin) and the (centre-padded) periodic Hann window -------------------
__global__ void k_stft_tables(float* tables, ResInfo r, int win_length) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= r.n_fft) return;
    const double ang = 6.283185307179586476925 * (double)t / (double)r.n_fft;
    tables[r.tw_off + 2 * t] = (float)cos(ang);
    tables[r.tw_off + 2 * t + 1] = (float)(-sin(ang));
    const int off = (r.n_fft - win_length) / 2;
    float w = 0.0f;
    if (t >= off && t < off + win_length) {
        // torch.hann_window(win_length) (periodic), evaluated in fp32 like torch does
        const float ph = 6.283185307179586f * (float)(t - off) / (float)win_length;
        w = 0.5f - 0.5f * (float)cos((double)ph);
    }
    tables[r.win_off + t] = w;
}

// (One merged 1024-lane launch measured 15.5-21.6 us against 5.0 + 6.7 us for these two: sixteen waves on one CU walking three
// pairs each lose more to their serial fp64 chains than the second launch costs.)
// stage 1: one 64-lane workgroup per (row, resolution) folds that row's strip partials (fp64, fixed order)
__global__ __launch_bounds__(64) void k_mrstft_rowsums(LossArgs a) { mrstft_rowsum<false>(a, blockIdx.x, blockIdx.y, threadIdx.x); }
// stage 2: loss scalar + per-row backward coefficients (without dL/dloss, applied by k_scale_coef)
__global__ __launch_bounds__(64) void k_mrstft_final(LossArgs a) {
    __shared__ double rs[kMaxRes][4];
    __shared__ double ratio[kMaxRes * 64];  // per (resolution, row): sqrt(sum |d|^2) / sqrt(sum |Y|^2)
    __shared__ float4 srow[kMaxRes * 64];   // the row sums, fetched by all lanes at once
    mrstft_final_body(a, threadIdx.x, rs, ratio, srow, true);
}
// stages 1 + 2 in one launch: every workgroup folds its (row, resolution) pair; the one that finishes LAST (ticket, zeroed by the
// call's first transform launch) goes on to the loss and the coefficients.  The fences are the textbook last-block pattern: agent
// scope, because the row sums of the other workgroups may sit in another XCD's L2 (cheap here - the transform kernels have ended,
// the L2s are clean; the same fences at the end of a transform kernel cost it 70 us).
__global__ __launch_bounds__(64) void k_mrstft_finish(LossArgs a, unsigned* ticket) {
    __shared__ double rs[kMaxRes][4];
    __shared__ double ratio[kMaxRes * 64];
    __shared__ float4 srow[kMaxRes * 64];
    __shared__ unsigned mine;
#ifndef MST_FINISH_FENCES
#define MST_FINISH_FENCES 0  // 1: the textbook __threadfence() pair around the ticket (rounds 4: two L2 write-back / L1 invalidate
                             // rounds, ~2 x 2-3.5 us of a 9.5 us launch)
#endif
#if !MST_FINISH_FENCES && defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "k_mrstft_finish: the fence-free hand-over below relies on gfx942 / gfx950 cache behaviour (sc1 write-through stores, L1-bypassing agent-scope loads); build any other target with -DMST_FINISH_FENCES=1"
#endif
#if MST_FINISH_FENCES
    mrstft_rowsum(a, blockIdx.x, blockIdx.y, threadIdx.x);
    __threadfence();
    if (threadIdx.x == 0) mine = atomicAdd(ticket, 1u);
    __syncthreads();
    if (mine != gridDim.x * gridDim.y - 1) return;
    __threadfence();
    mrstft_final_body(a, threadIdx.x, rs, ratio, srow, true);
#else
    // Round 5, no fence: the row sums travel as 8-byte agent-scope atomics on BOTH sides (write-through stores, L1-bypassing loads:
    // one of the valid hand-over forms of MI355X_MICROARCH.md), the writer drains its stores before it takes the ticket.  This is NOT
    // expressed in the language's memory model (relaxed atomics only): it is correct by the cache behaviour of the targets named in the
    // #error above, and only built for them - the portable form is the MST_FINISH_FENCES=1 branch.
    mrstft_rowsum<true>(a, blockIdx.x, blockIdx.y, threadIdx.x);
    if (threadIdx.x == 0) {
#if defined(__clang__)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        mine = atomicAdd(ticket, 1u);
    }
    __syncthreads();
    if (mine != gridDim.x * gridDim.y - 1) return;
    mrstft_final_body<true, true>(a, threadIdx.x, rs, ratio, srow, true);
#endif
}
// sharded evaluation only: this rank's totals per resolution, fixed order over the rows
__global__ __launch_bounds__(64) void k_mrstft_totals(LossArgs a) {
    const int tid = threadIdx.x;
    if (tid < a.n_res * 4) {
        const int res = tid >> 2, q = tid & 3;
        double t = 0.0;
        for (int row = 0; row < a.rows; ++row) t += (double)a.sums[((int64_t)res * a.rows + row) * 4 + q];
        a.totals[tid] = t;
    }
}
}  // namespace mst

// ======================================================================================= C ABI
using namespace mst;

namespace {
struct Plan {
    ResInfo res[kMaxRes];
    int log2n[kMaxRes], n_groups[kMaxRes], win[kMaxRes];
    bool engine2[kMaxRes];  // this resolution runs on the round-2 kernels (mst_fft2.h)
    int64_t part_off[kMaxRes];
    int64_t tables_floats;
    // workspace (floats): part | sums | coef | coef_scaled
    int64_t part_total, sums_off, coef_off, coefs_off, seam_off, tick_off, ws_floats;
    int64_t ymag_off[kMaxRes];  // round-2 kernels: the target's magnitudes kept by the forward for the backward (-1: none)
    int64_t xspec_off[kMaxRes];  // 8192-point resolution on the round-2 kernels: the prediction's spectrum kept for the backward (-1: none)
    int seam_res;  // index of the one seam-mode resolution whose seams are handed to a halo-mode launch, or -1
    bool ok;
};
Plan make_plan(const mst_mrstft_desc* d) {
    Plan p{};
    p.ok = false;
    if (!d || d->rows <= 0 || d->n_res <= 0 || d->n_res > kMaxRes) return p;
    int64_t t = 0, po = 0;
    for (int i = 0; i < d->n_res; ++i) {
        const int nf = d->fft_size[i];
        int lg = 0;
        while ((1 << lg) < nf) ++lg;
        if ((1 << lg) != nf || nf < 128 || nf > 8192) return p;
        if (d->hop_size[i] <= 0 || d->win_length[i] <= 0 || d->win_length[i] > nf) return p;
        if (d->n_samples <= nf / 2) return p;  // reflect padding needs n > n_fft/2 (torch.stft raises)
        ResInfo& r = p.res[i];
        r.n_fft = nf;
        r.hop = d->hop_size[i];
        r.n_frames = 1 + (int)(d->n_samples / r.hop);
        r.n_bins = nf / 2 + 1;
        r.tw_off = t;
        t += 2 * (int64_t)nf;
        r.win_off = t;
        t += nf;
        p.log2n[i] = lg;
        p.win[i] = d->win_length[i];
        // strips of 2 frames measured best on MI355X at cfg #2 (1: 423, 2: 411, 4: 443, 8: 455, 16: 566 us per fwd+bwd)
        r.frames_per_wg = ((int64_t)r.n_frames * d->rows >= 1024) ? 2 : 1;
#ifdef MST_STFT_FPW
        r.frames_per_wg = MST_STFT_FPW;
#endif
        p.n_groups[i] = (r.n_frames + r.frames_per_wg - 1) / r.frames_per_wg;
        // round-2 kernels: the reference's shape of resolution (hop = n_fft / 2, full-length window), rows long enough
        // that no frame reflects at both ends, whole hops per row (the owner-computes overlap-add assumes it)
        p.engine2[i] = (nf == 512 || nf == 2048 || nf == 8192) && r.hop * 2 == nf && d->win_length[i] == nf &&
                       d->n_samples >= 2 * (int64_t)nf && d->n_samples % r.hop == 0;
#ifdef MST_STFT_ROUND1
        p.engine2[i] = false;
#endif
        if (p.engine2[i]) {
            // balanced strips of ~8 / 4 / 2 frames (one workgroup each): consecutive frames share half their samples
            // fused forward with the kept spectra (one box, us): (4, 4) 83.8; (3, 4) 80.0; (3, 6) 79.6; (3, 7) 80.0; (4, 6) 78.6 vs 81.5; (6, 4) 83.6 vs
            // 81.5; (2, 4) 81.5 vs 85.0 - short one-wave 512-point strips fill the launch's tail, long 2048-point strips amortise their prologue
#ifndef MST_STFT2_STRIP_512
#define MST_STFT2_STRIP_512 3
#endif
#ifndef MST_STFT2_STRIP_2048
#define MST_STFT2_STRIP_2048 6
#endif
            // at cfg #2 (16 rows x 262144): 4100 one-wave / 1024 four-wave / 512 eight-wave workgroups = one resident round each
#ifndef MST_STFT2_STRIP_8192
// Frames per 8192-point forward strip.  Without the kept spectra one frame per workgroup was best (fused forward 72.6 -> 68.5 us: 1040 workgroups
// walk the 512 resident slots without the stragglers of two-frame strips); with them - a heavier epilogue, the strips of a row on one XCD - the
// prologue is worth amortising again (one box, us): 1: 78.3; 2: 73.9; 3: 73.6; 4: 71.0; 5: 70.7 / 69.3; 6: 72.6; 8: 83.3; 11: 114
#define MST_STFT2_STRIP_8192 5
#endif
            const int target = nf == 512 ? MST_STFT2_STRIP_512 : (nf == 2048 ? MST_STFT2_STRIP_2048 : MST_STFT2_STRIP_8192);
            p.n_groups[i] = r.n_frames / target > 0 ? r.n_frames / target : 1;
        }
        p.part_off[i] = po;
        po += (int64_t)d->rows * p.n_groups[i] * 4;
    }
    p.tables_floats = t;
    p.part_total = round_up(po, 64);
    p.sums_off = p.part_total;
    p.coef_off = p.sums_off + round_up((int64_t)d->n_res * d->rows * 4, 64);
    p.coefs_off = p.coef_off + round_up((int64_t)d->n_res * d->rows * 4, 64);
    p.tick_off = p.coefs_off + round_up((int64_t)d->n_res * d->rows * 4, 64);  // rows + 1 tickets of the fused reduction (mst_stft.h)
    p.seam_off = p.tick_off + 64;
    p.ws_floats = p.seam_off;
    // backward seam hand-over (mst_stft.h): exactly one seam-mode resolution, at least one halo-mode one, all on the round-2 kernels
    p.seam_res = -1;
    {
        int n_seam = 0, n_halo = 0, which = -1;
        bool all2 = true;
        for (int i = 0; i < d->n_res; ++i) {
            all2 = all2 && p.engine2[i];
            if (stft2_bwd_needs_zero(p.res[i].n_fft)) { ++n_seam; which = i; } else ++n_halo;
        }
        if (all2 && n_seam == 1 && n_halo >= 1) {
            p.seam_res = which;
            p.ws_floats += round_up((int64_t)d->rows * d->n_samples, 64);
        }
    }
    for (int i = 0; i < d->n_res; ++i) {
        p.ymag_off[i] = -1;
        p.xspec_off[i] = -1;
        if (p.engine2[i]) {
            p.ymag_off[i] = p.ws_floats;
            p.ws_floats += round_up((int64_t)d->rows * p.res[i].n_frames * p.res[i].n_bins, 64);
            if (stft2_keeps_spectrum(p.res[i].n_fft)) {
                p.xspec_off[i] = p.ws_floats;
                p.ws_floats += round_up((int64_t)d->rows * p.res[i].n_frames * p.res[i].n_bins * 2, 64);
            }
        }
    }
    p.ok = true;
    return p;
}
// n_fft = 8192: the backward runs on the in-place kernel (two workgroups per CU: 105 vs 128 us at cfg #2), the
// forward stays on the two-buffer kernel (in place it measured 69-78 vs 62 us).  Build-time A/B switches:
// -DMST_STFT_PINGPONG -> two-buffer kernels everywhere; -DMST_STFT_INPLACE_FWD -> in-place forward as well.
static constexpr bool inplace_8192(bool forward) {
#if defined(MST_STFT_PINGPONG)
    return false;
#elif defined(MST_STFT_INPLACE_FWD)
    return true;
#else
    return !forward;
#endif
}
#define MST_FOR_NFFT(nf, CALL) \
    switch (nf) {               \
        case 128: CALL(128); break;   \
        case 256: CALL(256); break;   \
        case 512: CALL(512); break;   \
        case 1024: CALL(1024); break; \
        case 2048: CALL(2048); break; \
        case 4096: CALL(4096); break; \
        case 8192: CALL(8192); break; \
        default: break;               \
    }
}  // namespace

extern "C" size_t mst_mrstft_tables_bytes(const mst_mrstft_desc* d) {
    const Plan p = make_plan(d);
    return p.ok ? (size_t)p.tables_floats * sizeof(float) : 0;
}
extern "C" size_t mst_mrstft_workspace_bytes(const mst_mrstft_desc* d) {
    const Plan p = make_plan(d);
    return p.ok ? (size_t)p.ws_floats * sizeof(float) : 0;
}
extern "C" int mst_mrstft_init_tables(const mst_mrstft_desc* d, void* tables, void* stream_) {
    const Plan p = make_plan(d);
    if (!p.ok || !tables) return hipErrorInvalidValue;
    hipStream_t stream = (hipStream_t)stream_;
    for (int i = 0; i < d->n_res; ++i)
        hipLaunchKernelGGL(k_stft_tables, dim3((p.res[i].n_fft + 255) / 256), dim3(256), 0, stream, (float*)tables, p.res[i], p.win[i]);
    return (int)hipGetLastError();
}

// stages: 1 = transforms + row sums (+ totals when asked for), 2 = loss + backward coefficients
static int mrstft_forward_stages(const mst_mrstft_desc* d, const float* pred, const float* target, const void* tables, float* loss,
                                 double* totals, const double* gtotals, int world, int stages, void* workspace,
                                 size_t workspace_bytes, void* stream_, bool keep = true) {
    Plan p = make_plan(d);
    if (!keep) {  // evaluation only (mst_mrstft_forward_eval): nothing is kept for a backward, the planes are not written
        for (int i = 0; i < kMaxRes; ++i) p.ymag_off[i] = p.xspec_off[i] = -1;
    }
    if (!p.ok || !workspace) return hipErrorInvalidValue;
    if ((stages & 1) && (!pred || !target || !tables)) return hipErrorInvalidValue;
    if ((stages & 2) && (!loss || world < 1)) return hipErrorInvalidValue;
    if (workspace_bytes < (size_t)p.ws_floats * sizeof(float)) return hipErrorInvalidValue;
    hipStream_t stream = (hipStream_t)stream_;
    float* ws = (float*)workspace;
    LossArgs la{};
    la.part = ws;
    la.sums = ws + p.sums_off;
    la.coef = ws + p.coef_off;
    la.loss = loss;
    la.n_res = d->n_res;
    la.rows = d->rows;
    la.w_sc = d->w_sc;
    la.w_log = d->w_log_mag;
    la.w_lin = d->w_lin_mag;
    la.sc_per_example = d->sc_per_example;
    la.totals = totals;
    la.gtotals = gtotals;
    la.world = gtotals ? world : 1;
#ifndef MST_MRSTFT_FUSE_FINISH
#define MST_MRSTFT_FUSE_FINISH 1
#endif
    // single-rank call: row sums, loss and backward coefficients in ONE launch (k_mrstft_finish) whose ticket the first transform
    // launch zeroes - that launch has to be a round-2 kernel
    const bool fuse = MST_MRSTFT_FUSE_FINISH && stages == 3 && !totals && !gtotals && p.engine2[0];
    for (int i = 0; i < d->n_res; ++i) {
        la.n_groups[i] = p.n_groups[i];
        la.part_off[i] = p.part_off[i];
        la.count[i] = (float)((double)d->rows * p.res[i].n_bins * p.res[i].n_frames);
    }
#ifndef MST_STFT3_FUSE
#define MST_STFT3_FUSE 1  // A/B switch: 0 = one forward launch per resolution
#endif
    // the reference's three resolutions, all on the round-2 kernels: ONE forward launch (k_stft3_fwd, mst_stft2.hip)
    int role[3] = {-1, -1, -1};
    bool fuse3 = MST_STFT3_FUSE && (stages & 1) && d->n_res == 3;
    for (int i = 0; i < d->n_res && fuse3; ++i) {
        const int which = p.res[i].n_fft == 8192 ? 0 : (p.res[i].n_fft == 2048 ? 1 : (p.res[i].n_fft == 512 ? 2 : -1));
        if (which < 0 || !p.engine2[i] || role[which] >= 0) fuse3 = false;
        else role[which] = i;
    }
    if (fuse3) {
        Stft3Args q{};
        for (int w = 0; w < 3; ++w) {
            const int i = role[w];
            StftArgs& a = q.a[w];
            a.pred = pred;
            a.target = target;
            a.tables = (const float*)tables;
            a.part = ws + p.part_off[i];
            a.r = p.res[i];
            a.log2n = p.log2n[i];
            a.n = d->n_samples;
            a.eps = d->eps;
            a.ymag = p.ymag_off[i] >= 0 ? ws + p.ymag_off[i] : nullptr;
            a.xspec = p.xspec_off[i] >= 0 ? ws + p.xspec_off[i] : nullptr;
            q.groups[w] = p.n_groups[i];
        }
        q.rows = d->rows;
        q.wg_end[0] = q.groups[0] * d->rows;
        q.wg_end[1] = q.wg_end[0] + q.groups[1] * ((d->rows + 1) / 2);
        q.wg_end[2] = q.wg_end[1] + (q.groups[2] * d->rows + 7) / 8;
        q.tickets = fuse ? reinterpret_cast<unsigned*>(ws + p.tick_off) : nullptr;
        launch_stft3_fwd(q, stream);
    }
    for (int i = 0; i < d->n_res; ++i) {
        if (!(stages & 1) || fuse3) continue;
        StftArgs a{};
        a.pred = pred;
        a.target = target;
        a.tables = (const float*)tables;
        a.part = ws + p.part_off[i];
        a.r = p.res[i];
        a.log2n = p.log2n[i];
        a.n = d->n_samples;
        a.eps = d->eps;
        a.ymag = p.ymag_off[i] >= 0 ? ws + p.ymag_off[i] : nullptr;
        a.xspec = p.xspec_off[i] >= 0 ? ws + p.xspec_off[i] : nullptr;
        if (fuse && i == 0) a.tickets = reinterpret_cast<unsigned*>(ws + p.tick_off);
        const dim3 grid(p.n_groups[i], d->rows);
#define MST_LAUNCH_FWD(NF) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_stft_fwd<NF>), grid, dim3(stft_threads(NF)), 0, stream, a)
        if (p.engine2[i]) {
            launch_stft2_fwd(a, p.n_groups[i], d->rows, stream);
        } else if (a.r.n_fft == 8192 && inplace_8192(true))
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_stft_fwd_ip<8192>), grid, dim3(kIpThreads), 0, stream, a);
        else
            MST_FOR_NFFT(a.r.n_fft, MST_LAUNCH_FWD)
    }
    if (fuse) {
        hipLaunchKernelGGL(k_mrstft_finish, dim3(d->rows, d->n_res), dim3(64), 0, stream, la, reinterpret_cast<unsigned*>(ws + p.tick_off));
        return (int)hipGetLastError();
    }
    if (stages & 1) {
        hipLaunchKernelGGL(k_mrstft_rowsums, dim3(d->rows, d->n_res), dim3(64), 0, stream, la);
        if (totals) hipLaunchKernelGGL(k_mrstft_totals, dim3(1), dim3(64), 0, stream, la);
    }
    if (stages & 2) hipLaunchKernelGGL(k_mrstft_final, dim3(1), dim3(64), 0, stream, la);
    return (int)hipGetLastError();
}

extern "C" int mst_mrstft_forward(const mst_mrstft_desc* d, const float* pred, const float* target, const void* tables,
                                  float* loss, void* workspace, size_t workspace_bytes, void* stream) {
    return mrstft_forward_stages(d, pred, target, tables, loss, nullptr, nullptr, 1, 3, workspace, workspace_bytes, stream);
}
extern "C" int mst_mrstft_forward_eval(const mst_mrstft_desc* d, const float* pred, const float* target, const void* tables,
                                       float* loss, void* workspace, size_t workspace_bytes, void* stream) {
    return mrstft_forward_stages(d, pred, target, tables, loss, nullptr, nullptr, 1, 3, workspace, workspace_bytes, stream, false);
}
extern "C" int mst_mrstft_forward_partial(const mst_mrstft_desc* d, const float* pred, const float* target, const void* tables,
                                          double* totals, void* workspace, size_t workspace_bytes, void* stream) {
    if (!totals) return hipErrorInvalidValue;
    return mrstft_forward_stages(d, pred, target, tables, nullptr, totals, nullptr, 1, 1, workspace, workspace_bytes, stream);
}
extern "C" int mst_mrstft_forward_finish(const mst_mrstft_desc* d, const double* global_totals, int32_t world, float* loss,
                                         void* workspace, size_t workspace_bytes, void* stream) {
    return mrstft_forward_stages(d, nullptr, nullptr, nullptr, loss, nullptr, global_totals, world, 2, workspace, workspace_bytes, stream);
}

extern "C" int mst_mrstft_backward(const mst_mrstft_desc* d, const float* pred, const float* target, const void* tables,
                                   const float* grad_loss, float* grad_pred, void* workspace, size_t workspace_bytes,
                                   void* stream_) {
    const Plan p = make_plan(d);
    if (!p.ok || !pred || !target || !tables || !grad_loss || !grad_pred || !workspace) return hipErrorInvalidValue;
    if (workspace_bytes < (size_t)p.ws_floats * sizeof(float)) return hipErrorInvalidValue;
    hipStream_t stream = (hipStream_t)stream_;
    float* ws = (float*)workspace;
    bool all2 = true;
    for (int i = 0; i < d->n_res; ++i) all2 = all2 && p.engine2[i];
    if (all2) {
        // round-2 kernels: owner-computes overlap-add.  Seam-mode resolutions (8192) go first, onto a zeroed buffer; the
        // halo-mode ones follow and add to it; with no seam-mode resolution the first launch owns the buffer (no memset).
        // With exactly one seam-mode resolution and a halo-mode one behind it the seams are handed over through a scratch slab
        // instead (no memset, no atomics: every sample is stored once by the seam launch and read-modified once per later launch).
        bool zero = false;
        for (int i = 0; i < d->n_res; ++i) zero = zero || stft2_bwd_needs_zero(p.res[i].n_fft);
        const bool handover = p.seam_res >= 0;
        if (zero && !handover) (void)hipMemsetAsync(grad_pred, 0, (size_t)d->rows * d->n_samples * sizeof(float), stream);
        bool written = zero;
        bool pending = handover;  // the parked seam halves still wait for a halo-mode launch
#ifndef MST_STFT2_BWD_FUSE
#define MST_STFT2_BWD_FUSE 1  // A/B switch: 0 = one backward launch per resolution
#endif
        // the 512- and the 2048-point resolution (halo mode both) in one launch: k_stft2_bwd_512_2048, mst_stft2.hip
        int i512 = -1, i2048 = -1;
        for (int i = 0; i < d->n_res; ++i) {
            if (p.res[i].n_fft == 512) i512 = i512 < 0 ? i : -2;
            if (p.res[i].n_fft == 2048) i2048 = i2048 < 0 ? i : -2;
        }
        const bool fuse2 = MST_STFT2_BWD_FUSE && i512 >= 0 && i2048 >= 0 && stft2_bwd_can_fuse(d->n_samples);
        StftArgs held{};  // the 512-point launch's arguments, kept until the 2048-point resolution comes up
        for (int pass = 0; pass < 2; ++pass) {
            for (int k = 0; k < d->n_res; ++k) {
                // fused pair: the 512-point resolution is visited first whatever the caller's order (it owns the samples first)
                const int i = (fuse2 && pass == 1) ? (k == (i512 < i2048 ? i512 : i2048) ? i512 : (k == (i512 < i2048 ? i2048 : i512) ? i2048 : k)) : k;
                if (stft2_bwd_needs_zero(p.res[i].n_fft) != (pass == 0)) continue;
                StftArgs a{};
                a.pred = pred;
                a.target = target;
                a.tables = (const float*)tables;
                a.sums = ws + p.sums_off + (int64_t)i * d->rows * 4;
                a.coef = ws + p.coef_off + (int64_t)i * d->rows * 4;
                a.grad_loss = grad_loss;
                a.grad_pred = grad_pred;
                a.r = p.res[i];
                a.log2n = p.log2n[i];
                a.n = d->n_samples;
                a.eps = d->eps;
                a.ymag = ws + p.ymag_off[i];
                a.xspec = p.xspec_off[i] >= 0 ? ws + p.xspec_off[i] : nullptr;
                a.accumulate = (pass == 1 && written) ? 1 : 0;
                if (handover && (pass == 0 || pending)) {
                    const ResInfo& sr = p.res[p.seam_res];
                    a.seam = ws + p.seam_off;
                    a.seam_frames = sr.n_frames;
                    a.seam_groups = stft2_bwd_groups(sr.n_fft, sr.n_frames, d->rows);
                    a.seam_hop = sr.n_fft / 2;
                    if (pass == 1) pending = false;
                }
                if (fuse2 && i == i512) {
                    held = a;  // launched together with the 2048-point resolution below
                } else if (fuse2 && i == i2048) {
                    launch_stft2_bwd_512_2048(held, a, d->rows, stream);
                } else {
                    launch_stft2_bwd(a, stft2_bwd_groups(a.r.n_fft, a.r.n_frames, d->rows), d->rows, stream);
                }
                written = true;
            }
        }
        return (int)hipGetLastError();
    }
    (void)hipMemsetAsync(grad_pred, 0, (size_t)d->rows * d->n_samples * sizeof(float), stream);
    for (int i = 0; i < d->n_res; ++i) {
        StftArgs a{};
        a.pred = pred;
        a.target = target;
        a.tables = (const float*)tables;
        a.sums = ws + p.sums_off + (int64_t)i * d->rows * 4;
        a.coef = ws + p.coef_off + (int64_t)i * d->rows * 4;
        a.grad_loss = grad_loss;
        a.grad_pred = grad_pred;
        a.r = p.res[i];
        a.log2n = p.log2n[i];
        a.n = d->n_samples;
        a.eps = d->eps;
        // three LDS buffers (pairing two frames per inverse FFT) fit up to n_fft = 4096; 8192 runs one frame per workgroup
        const bool pair = a.r.n_fft <= 4096;
        const dim3 grid(pair ? (a.r.n_frames + 1) / 2 : a.r.n_frames, d->rows);
#define MST_LAUNCH_BWD(NF)                                                                                             \
    if (NF <= 4096) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_stft_bwd<NF, (NF <= 4096)>), grid, dim3(stft_threads(NF)), 0, stream, a); \
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_stft_bwd<NF, false>), grid, dim3(stft_threads(NF)), 0, stream, a)
        if (a.r.n_fft == 8192 && inplace_8192(false))
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_stft_bwd_ip<8192>), grid, dim3(kIpThreads), 0, stream, a);
        else
            MST_FOR_NFFT(a.r.n_fft, MST_LAUNCH_BWD)
    }
    return (int)hipGetLastError();
}
