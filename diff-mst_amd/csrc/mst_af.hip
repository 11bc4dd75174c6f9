// mst_af.hip - AudioFeatureLoss (reference mst/loss.py:198-260 and the five transforms :62-195):
//   rms, crest factor, stereo width, stereo imbalance  (closed-form reductions over the stereo mix)
//   Bark spectrum: mid/side -> STFT(32768, hop 8192, periodic Hann, reflect) -> |X| -> mean over
//   frames -> (24 x 16385) filterbank -> log(. + 1e-8)
// each compared with the target by MSE and weighted.  Forward and reverse-mode.
//
// The 32768-point real transform of a frame is one 16384-point complex FFT of (even + i odd)
// samples living entirely in LDS (128 KiB of the CU's 160 KiB): in-place radix-4 decimation in
// frequency (natural in, base-4 digit-reversed out) with an XOR bank swizzle, the real-input
// untangling done on the digit-reversed image; the adjoint writes its Hermitian-packed input back
// to the same slots and runs the mirror decimation-in-time network (digit-reversed in, natural
// out), so nothing but partial magnitude sums and the final gradient ever reaches HBM.
#include "mst_common.h"

namespace mst {

constexpr int kAfFft = 32768;       // reference default fft_size (mst/loss.py:64)
constexpr int kAfM = kAfFft / 2;    // complex points
constexpr int kAfHop = kAfFft / 4;  // reference hop_length = fft_size // 4 (mst/loss.py:106)
constexpr int kAfBins = kAfM + 1;
constexpr int kAfBands = 24;
constexpr int kAfThreads = 1024;

// tables (floats): twM: kAfM float2 (W_M^t) | twN: (kAfM + 1) float2 (W_N^k) | win: kAfFft floats
constexpr int64_t kAfTwM = 0, kAfTwN = 2 * kAfM, kAfWin = kAfTwN + 2 * (kAfM + 1), kAfTablesFloats = kAfWin + kAfFft;

__device__ __forceinline__ float2 cmulf(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ float2 cconj(float2 a) { return make_float2(a.x, -a.y); }
// bank swizzle: fold the top index bits into the low 5 so that digit-reversed neighbours spread over banks
// SWZ = false: plain slots.  Measured: the forward kernel is 35 % faster WITHOUT the swizzle (273 -> 178 us at bs 8:
// its address arithmetic costs more than the epilogue's bank conflicts), the backward kernel is not (345 vs 358 us).
template <bool SWZ>
__device__ __forceinline__ int swzT(int a) { return SWZ ? a ^ ((a >> 9) & 31) : a; }
__device__ __forceinline__ int swz(int a) { return swzT<true>(a); }
__device__ __forceinline__ int rev4_7(int k) {  // reverse the 7 base-4 digits of a 14-bit index
    int r = 0;
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        r = (r << 2) | (k & 3);
        k >>= 2;
    }
    return r;
}

struct AfTw {
    float2 coarse[kAfM / 64];
    float2 fine[64];
    __device__ __forceinline__ float2 get(int t) const { return cmulf(coarse[t >> 6], fine[t & 63]); }
};

// in-place radix-4 DIF, forward sign: natural order in, digit-reversed out
template <bool SWZ>
__device__ void fft16k_dif(float2* buf, const AfTw& T, int tid) {
    for (int L = kAfM / 4; L >= 1; L >>= 2) {
        const int tstep = kAfM / (4 * L);
#pragma unroll
        for (int b = tid; b < kAfM / 4; b += kAfThreads) {
            const int k = b & (L - 1);
            const int i0 = ((b - k) << 2) + k;
            const int p0 = swzT<SWZ>(i0), p1 = swzT<SWZ>(i0 + L), p2 = swzT<SWZ>(i0 + 2 * L), p3 = swzT<SWZ>(i0 + 3 * L);
            const float2 u0 = buf[p0], u1 = buf[p1], u2 = buf[p2], u3 = buf[p3];
            const float2 s02 = make_float2(u0.x + u2.x, u0.y + u2.y), d02 = make_float2(u0.x - u2.x, u0.y - u2.y);
            const float2 s13 = make_float2(u1.x + u3.x, u1.y + u3.y), d13 = make_float2(u1.x - u3.x, u1.y - u3.y);
            float2 y0 = make_float2(s02.x + s13.x, s02.y + s13.y);
            float2 y1 = make_float2(d02.x + d13.y, d02.y - d13.x);
            float2 y2 = make_float2(s02.x - s13.x, s02.y - s13.y);
            float2 y3 = make_float2(d02.x - d13.y, d02.y + d13.x);
            if (k) {
                const float2 w1 = T.get(k * tstep);
                const float2 w2 = cmulf(w1, w1);
                y1 = cmulf(y1, w1);
                y2 = cmulf(y2, w2);
                y3 = cmulf(y3, cmulf(w2, w1));
            }
            buf[p0] = y0; buf[p1] = y1; buf[p2] = y2; buf[p3] = y3;
        }
        __syncthreads();
    }
}
// in-place radix-4 DIT, forward sign: digit-reversed in, natural order out
__device__ void fft16k_dit(float2* buf, const AfTw& T, int tid) {
    for (int L = 1; L < kAfM; L <<= 2) {
        const int tstep = kAfM / (4 * L);
#pragma unroll
        for (int b = tid; b < kAfM / 4; b += kAfThreads) {
            const int k = b & (L - 1);
            const int i0 = ((b - k) << 2) + k;
            const int p0 = swz(i0), p1 = swz(i0 + L), p2 = swz(i0 + 2 * L), p3 = swz(i0 + 3 * L);
            float2 u0 = buf[p0], u1 = buf[p1], u2 = buf[p2], u3 = buf[p3];
            if (k) {
                const float2 w1 = T.get(k * tstep);
                const float2 w2 = cmulf(w1, w1);
                u1 = cmulf(u1, w1);
                u2 = cmulf(u2, w2);
                u3 = cmulf(u3, cmulf(w2, w1));
            }
            const float2 s02 = make_float2(u0.x + u2.x, u0.y + u2.y), d02 = make_float2(u0.x - u2.x, u0.y - u2.y);
            const float2 s13 = make_float2(u1.x + u3.x, u1.y + u3.y), d13 = make_float2(u1.x - u3.x, u1.y - u3.y);
            buf[p0] = make_float2(s02.x + s13.x, s02.y + s13.y);
            buf[p1] = make_float2(d02.x + d13.y, d02.y - d13.x);
            buf[p2] = make_float2(s02.x - s13.x, s02.y - s13.y);
            buf[p3] = make_float2(d02.x - d13.y, d02.y + d13.x);
        }
        __syncthreads();
    }
}

__device__ __forceinline__ int64_t af_reflect(int64_t i, int64_t n) {
    if (i < 0) i = -i;
    if (i >= n) i = 2 * (n - 1) - i;
    return i;
}

struct AfArgs {
    const float* pred;    // (bs, 2, n)
    const float* target;  // (bs, 2, n)
    const float* tables;
    const float* fb;      // (kAfBins, 24) filterbank, row-major like the reference's (n_freqs, n_barks)
    float* magpart;       // (4*bs, n_groups, kAfBins) partial sums of |X| over a strip of frames
    float* meanmag;       // (4*bs, kAfBins)
    float* bark;          // (4*bs, 24) log band energies; (4*bs, 24) linear band energies follow
    double* stats;        // (2*bs, 8) reduced statistics per (signal set, b), see k_af_stats / k_af_stats_reduce
    float* statpart;      // partials of the above
    float* bandpart;      // (4*bs, kAfBinSlices, 24) partial band energies
    float* losses;        // 5 weighted loss scalars out
    float* coef;          // backward coefficients
    const float* grad_losses;  // (5) upstream dL/d(loss_k)
    float* grad_pred;     // (bs, 2, n)
    float* yframes;       // (2*bs, n_frames, kAfFft) windowed adjoint frames of the prediction's mid / side signals
    float weights[5];
    int bs, n_frames, n_groups, n_statblk;
    int64_t n;
};

// signal index s in [0, 4*bs): which = s / bs (0 pred mid, 1 pred side, 2 target mid, 3 target side), b = s % bs
__device__ __forceinline__ void af_signal(const AfArgs& a, int s, const float*& l, const float*& r, float& sign) {
    const int which = s / a.bs, b = s % a.bs;
    const float* base = (which < 2 ? a.pred : a.target) + (int64_t)b * 2 * a.n;
    l = base;
    r = base + a.n;
    sign = (which & 1) ? -1.0f : 1.0f;
}

// pack frame f of (L + sign R) as z[m] = w[2m] x[2m] + i w[2m+1] x[2m+1]
template <bool SWZ>
__device__ __forceinline__ void af_load_frame(float2* buf, const float* l, const float* r, float sign, const float* win, int f,
                                              int64_t n, int tid) {
    const int64_t start = (int64_t)f * kAfHop - kAfFft / 2;
    // interior frames of 16-byte aligned rows (all but the two reflected ends): eight samples per lane and step
    if (start >= 0 && start + kAfFft <= n && !(((uintptr_t)l | (uintptr_t)r) & 15) && !(n & 3)) {
        for (int q = tid; q < kAfM / 4; q += kAfThreads) {
            const int m = 4 * q;
            const float4 l0 = *reinterpret_cast<const float4*>(l + start + 2 * m), l1 = *reinterpret_cast<const float4*>(l + start + 2 * m + 4);
            const float4 r0 = *reinterpret_cast<const float4*>(r + start + 2 * m), r1 = *reinterpret_cast<const float4*>(r + start + 2 * m + 4);
            const float2* w2 = reinterpret_cast<const float2*>(win + 2 * m);  // the window table is 8-byte aligned
            const float2 wa = w2[0], wb = w2[1], wc = w2[2], wd = w2[3];
            buf[swzT<SWZ>(m)] = make_float2(wa.x * (l0.x + sign * r0.x), wa.y * (l0.y + sign * r0.y));
            buf[swzT<SWZ>(m + 1)] = make_float2(wb.x * (l0.z + sign * r0.z), wb.y * (l0.w + sign * r0.w));
            buf[swzT<SWZ>(m + 2)] = make_float2(wc.x * (l1.x + sign * r1.x), wc.y * (l1.y + sign * r1.y));
            buf[swzT<SWZ>(m + 3)] = make_float2(wd.x * (l1.z + sign * r1.z), wd.y * (l1.w + sign * r1.w));
        }
        return;
    }
    for (int m = tid; m < kAfM; m += kAfThreads) {
        const int64_t i0 = af_reflect(start + 2 * m, n), i1 = af_reflect(start + 2 * m + 1, n);
        const float x0 = l[i0] + sign * r[i0], x1 = l[i1] + sign * r[i1];
        buf[swzT<SWZ>(m)] = make_float2(win[2 * m] * x0, win[2 * m + 1] * x1);
    }
}

// X[k] and X[M-k] of the real frame from the digit-reversed half-size spectrum
__device__ __forceinline__ int rev4_6(int k) {  // reverse the 6 base-4 digits of a 12-bit index
    int r = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        r = (r << 2) | (k & 3);
        k >>= 2;
    }
    return r;
}
// X[k], X[M-k] from the half-size spectrum values z[k], z[M-k] and w = W_N^k
__device__ __forceinline__ void af_untangle_core(float2 zk, float2 zm, float2 w, float2& Xk, float2& Xm) {
    const float2 E = make_float2(0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y));
    const float2 O = make_float2(0.5f * (zk.y + zm.y), -0.5f * (zk.x - zm.x));
    const float2 wo = cmulf(w, O);
    Xk = make_float2(E.x + wo.x, E.y + wo.y);
    Xm = make_float2(E.x - wo.x, -(E.y - wo.y));
}
template <bool SWZ>
__device__ __forceinline__ void af_untangle(const float2* buf, const float2* twN, int k, float2& Xk, float2& Xm) {
    const float2 zk = buf[swzT<SWZ>(rev4_7(k))], zm = buf[swzT<SWZ>(rev4_7((kAfM - k) & (kAfM - 1)))];
    const float2 E = make_float2(0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y));      // even-sample spectrum
    const float2 O = make_float2(0.5f * (zk.y + zm.y), -0.5f * (zk.x - zm.x));     // odd-sample spectrum
    const float2 wo = cmulf(twN[k], O);
    Xk = make_float2(E.x + wo.x, E.y + wo.y);
    Xm = make_float2(E.x - wo.x, -(E.y - wo.y));  // X[M-k] = conj(E - W^k O)
}

__global__ __launch_bounds__(kAfThreads) void k_af_bark_fwd(AfArgs a) {
    __shared__ __attribute__((aligned(16))) float2 buf[kAfM];
    __shared__ AfTw T;
    const int tid = threadIdx.x, s = blockIdx.y, grp = blockIdx.x;
    const float2* twM = reinterpret_cast<const float2*>(a.tables + kAfTwM);
    const float2* twN = reinterpret_cast<const float2*>(a.tables + kAfTwN);
    const float* win = a.tables + kAfWin;
    for (int i = tid; i < kAfM / 64; i += kAfThreads) T.coarse[i] = twM[i * 64];
    for (int i = tid; i < 64; i += kAfThreads) T.fine[i] = twM[i];
    const float *l, *r;
    float sign;
    af_signal(a, s, l, r, sign);
    // The spectrum sits digit-reversed in plain (unswizzled) slots.  Lane tid owns the slot groups m = tid + 1024 j
    // (j < 4): slots 4m, 4m+1 hold bins k = d 4096 + r (d = 0, 1; r = rev4_6(m)) and their mirrors M - k sit in slots
    // 4m'+3, 4m'+2 with m' = rev4_6(4096 - r) - two 16-byte LDS reads per group, consecutive lanes on consecutive
    // groups.  (m = 0 pairs bins 0 / M and 4096 / 12288 in slots 0, 1, 3; lane 0 also owns bin M/2.)
    int rr[4], mp[4];
    float2 w0[4], w1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = tid + kAfThreads * j;
        rr[j] = rev4_6(m);
        mp[j] = rev4_6((4096 - rr[j]) & 4095);
        w0[j] = twN[rr[j]];
        w1[j] = twN[4096 + rr[j]];
    }
    float acc_lo[8], acc_hi[8], acc_mid = 0.0f;  // [2 j + d]: bin k = d 4096 + rr[j]  and its mirror
#pragma unroll
    for (int j = 0; j < 8; ++j) acc_lo[j] = acc_hi[j] = 0.0f;
    // strips of near-equal length: frames [grp F / G, (grp+1) F / G)
    const int f0 = (int)(((int64_t)grp * a.n_frames) / a.n_groups), f1 = (int)(((int64_t)(grp + 1) * a.n_frames) / a.n_groups);
    for (int f = f0; f < f1; ++f) {
        __syncthreads();
        af_load_frame<false>(buf, l, r, sign, win, f, a.n, tid);
        __syncthreads();
        fft16k_dif<false>(buf, T, tid);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = tid + kAfThreads * j;
            const float4 zk2 = *reinterpret_cast<const float4*>(&buf[4 * m]);
            float4 zm2 = *reinterpret_cast<const float4*>(&buf[4 * mp[j] + 2]);
            if (m == 0) zm2 = make_float4(buf[3].x, buf[3].y, buf[0].x, buf[0].y);  // mirrors of bins 4096 and 0
            float2 Xk, Xm;
            af_untangle_core(make_float2(zk2.x, zk2.y), make_float2(zm2.z, zm2.w), w0[j], Xk, Xm);
            acc_lo[2 * j] += sqrtf(Xk.x * Xk.x + Xk.y * Xk.y);
            acc_hi[2 * j] += sqrtf(Xm.x * Xm.x + Xm.y * Xm.y);  // k = 0 -> bin M (Nyquist)
            af_untangle_core(make_float2(zk2.z, zk2.w), make_float2(zm2.x, zm2.y), w1[j], Xk, Xm);
            acc_lo[2 * j + 1] += sqrtf(Xk.x * Xk.x + Xk.y * Xk.y);
            acc_hi[2 * j + 1] += sqrtf(Xm.x * Xm.x + Xm.y * Xm.y);
        }
        if (tid == 0) {
            float2 Xk, Xm;
            af_untangle<false>(buf, twN, kAfM / 2, Xk, Xm);
            acc_mid += sqrtf(Xk.x * Xk.x + Xk.y * Xk.y);
        }
    }
    float* out = a.magpart + ((int64_t)s * a.n_groups + grp) * kAfBins;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            const int k = d * 4096 + rr[j];
            out[k] = acc_lo[2 * j + d];
            out[kAfM - k] = acc_hi[2 * j + d];  // k = 0 writes bin M
        }
    }
    if (tid == 0) out[kAfM / 2] = acc_mid;
}

// mean over frames + filterbank, one slice of the bins per workgroup.  grid (kAfBinSlices, 4*bs), 256 lanes.
constexpr int kAfStatSpan = 256 * 16;  // samples per k_af_stats workgroup
constexpr int kAfBinSlices = 16;
constexpr int kAfSliceBins = (kAfBins + kAfBinSlices - 1) / kAfBinSlices;
__global__ __launch_bounds__(256) void k_af_bark_reduce(AfArgs a) {
    __shared__ float red[4][kAfBands];
    const int tid = threadIdx.x, s = blockIdx.y, sl = blockIdx.x;
    float band[kAfBands];
#pragma unroll
    for (int j = 0; j < kAfBands; ++j) band[j] = 0.0f;
    const float invF = 1.0f / (float)a.n_frames;
    const int k1 = (sl + 1) * kAfSliceBins < kAfBins ? (sl + 1) * kAfSliceBins : kAfBins;
    for (int k = sl * kAfSliceBins + tid; k < k1; k += 256) {
        float m = 0.0f;
        for (int g = 0; g < a.n_groups; ++g) m += a.magpart[((int64_t)s * a.n_groups + g) * kAfBins + k];
        m *= invF;
        a.meanmag[(int64_t)s * kAfBins + k] = m;
        const float* fr = a.fb + (int64_t)k * kAfBands;
#pragma unroll
        for (int j = 0; j < kAfBands; ++j) band[j] = fmaf(fr[j], m, band[j]);
    }
#pragma unroll
    for (int j = 0; j < kAfBands; ++j) {
        const float v = wave_sum(band[j]);
        if ((tid & 63) == 0) red[tid >> 6][j] = v;
    }
    __syncthreads();
    if (tid < kAfBands)
        a.bandpart[((int64_t)s * kAfBinSlices + sl) * kAfBands + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
}
// slices -> band energies (log and linear) of signal s, and the statistics of (set, b) = blockIdx.x < 2*bs.
// grid (4*bs), 64 lanes: fixed-order sums.
__global__ __launch_bounds__(64) void k_af_finish(AfArgs a) {
    const int tid = threadIdx.x, s = blockIdx.x;
    if (tid < kAfBands) {
        float lin = 0.0f;
        for (int sl = 0; sl < kAfBinSlices; ++sl) lin += a.bandpart[((int64_t)s * kAfBinSlices + sl) * kAfBands + tid];
        a.bark[(int64_t)s * kAfBands + tid] = logf(lin + 1e-8f);
        a.bark[(int64_t)(4 * a.bs + s) * kAfBands + tid] = lin;
    }
    if (s < 2 * a.bs) {  // statistics of signal set / batch item sb = s: lanes stride over the time blocks
        double sum[4] = {0, 0, 0, 0};
        float ml = -1.f, mr = -1.f;
        int64_t il = 0x7fffffffffffLL, ir = 0x7fffffffffffLL;
        for (int k = tid; k < a.n_statblk; k += 64) {
            const float* p = a.statpart + ((int64_t)s * a.n_statblk + k) * 8;
            const float4 v0 = *reinterpret_cast<const float4*>(p), v1 = *reinterpret_cast<const float4*>(p + 4);
            sum[0] += (double)v0.x; sum[1] += (double)v0.y; sum[2] += (double)v0.z; sum[3] += (double)v0.w;
            // first maximum in time order: ascending k inside a lane, ties across lanes broken by the index below
            if (v1.x > ml) { ml = v1.x; il = (int64_t)k * kAfStatSpan + __float_as_int(v1.y); }
            if (v1.z > mr) { mr = v1.z; ir = (int64_t)k * kAfStatSpan + __float_as_int(v1.w); }
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
#pragma unroll
            for (int q = 0; q < 4; ++q) sum[q] += __shfl_xor(sum[q], m);
            float ov = __shfl_xor(ml, m);
            int64_t oi = __shfl_xor(il, m);
            if (ov > ml || (ov == ml && oi < il)) { ml = ov; il = oi; }
            ov = __shfl_xor(mr, m);
            oi = __shfl_xor(ir, m);
            if (ov > mr || (ov == mr && oi < ir)) { mr = ov; ir = oi; }
        }
        if (tid == 0) {
            double* o = a.stats + (int64_t)s * 8;
            const double N = (double)a.n;
            o[0] = sum[0] / N; o[1] = sum[1] / N; o[2] = sum[2] / N; o[3] = sum[3] / N;  // mean L^2, R^2, (L+R)^2, (L-R)^2
            o[4] = ml; o[5] = (double)il; o[6] = mr; o[7] = (double)ir;
        }
    }
}

// ---- closed-form features: per (set in {pred,target}, b) partial reductions over a slice of time
// stat slots: 0 sum L^2, 1 sum R^2, 2 sum (L+R)^2, 3 sum (L-R)^2, 4 max|L|, 5 argmax L, 6 max|R|, 7 argmax R
__global__ __launch_bounds__(256) void k_af_stats(AfArgs a) {
    __shared__ float sv[4][8];
    const int tid = threadIdx.x, sb = blockIdx.y;  // sb = set * bs + b
    const float* base = (sb < a.bs ? a.pred : a.target) + (int64_t)(sb % a.bs) * 2 * a.n;
    const float *l = base, *r = base + a.n;
    const int64_t i0 = (int64_t)blockIdx.x * kAfStatSpan;
    float s0 = 0, s1 = 0, s2 = 0, s3 = 0, ml = -1.0f, mr = -1.0f;
    int il = 0x7fffffff, ir = 0x7fffffff;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int64_t i = i0 + ((int64_t)q * 256 + tid) * 4;
        const float4 lv = load4(l, i, a.n), rv = load4(r, i, a.n);
        const float le[4] = {lv.x, lv.y, lv.z, lv.w}, re[4] = {rv.x, rv.y, rv.z, rv.w};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (i + t < a.n) {
                s0 = fmaf(le[t], le[t], s0);
                s1 = fmaf(re[t], re[t], s1);
                const float p = le[t] + re[t], m = le[t] - re[t];
                s2 = fmaf(p, p, s2);
                s3 = fmaf(m, m, s3);
                if (fabsf(le[t]) > ml) { ml = fabsf(le[t]); il = (int)(i + t - i0); }
                if (fabsf(re[t]) > mr) { mr = fabsf(re[t]); ir = (int)(i + t - i0); }
            }
        }
    }
    s0 = wave_sum(s0); s1 = wave_sum(s1); s2 = wave_sum(s2); s3 = wave_sum(s3);
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        float ov = __shfl_xor(ml, m); int oi = __shfl_xor(il, m);
        if (ov > ml || (ov == ml && oi < il)) { ml = ov; il = oi; }
        ov = __shfl_xor(mr, m); oi = __shfl_xor(ir, m);
        if (ov > mr || (ov == mr && oi < ir)) { mr = ov; ir = oi; }
    }
    if ((tid & 63) == 0) {
        float* o = sv[tid >> 6];
        o[0] = s0; o[1] = s1; o[2] = s2; o[3] = s3; o[4] = ml; o[5] = __int_as_float(il); o[6] = mr; o[7] = __int_as_float(ir);
    }
    __syncthreads();
    if (tid == 0) {
        float* o = a.statpart + ((int64_t)sb * a.n_statblk + blockIdx.x) * 8;
        o[0] = (sv[0][0] + sv[1][0]) + (sv[2][0] + sv[3][0]);
        o[1] = (sv[0][1] + sv[1][1]) + (sv[2][1] + sv[3][1]);
        o[2] = (sv[0][2] + sv[1][2]) + (sv[2][2] + sv[3][2]);
        o[3] = (sv[0][3] + sv[1][3]) + (sv[2][3] + sv[3][3]);
        float bl = sv[0][4], br = sv[0][6];
        int bil = __float_as_int(sv[0][5]), bir = __float_as_int(sv[0][7]);
        for (int w = 1; w < 4; ++w) {
            const int wil = __float_as_int(sv[w][5]), wir = __float_as_int(sv[w][7]);
            if (sv[w][4] > bl || (sv[w][4] == bl && wil < bil)) { bl = sv[w][4]; bil = wil; }
            if (sv[w][6] > br || (sv[w][6] == br && wir < bir)) { br = sv[w][6]; bir = wir; }
        }
        o[4] = bl; o[5] = __int_as_float(bil); o[6] = br; o[7] = __int_as_float(bir);
    }
}

// ---- closed-form features of batch item b from the reduced statistics.
// part[0..3] += squared feature differences (rms, crest, width, imbalance);  when `coef` is given the
// cotangent of the prediction is written as  gL = c0 L + c1 R, gR = c2 L + c3 R  plus the crest-factor
// deltas at the two arg-max samples (c4/c5 = value/index for L, c6/c7 for R), scaled by the upstream
// gradients g[0..3] of the four weighted losses.
__device__ void af_features(const AfArgs& a, int b, const float* g, double* part, float* coef) {
    const double N = (double)a.n;
    const double c20 = 8.685889638065035;  // 20 / ln 10
    double st[2][8];
    for (int set = 0; set < 2; ++set)
        for (int q = 0; q < 8; ++q) st[set][q] = a.stats[((int64_t)set * a.bs + b) * 8 + q];  // k_af_finish
    const double g0 = g ? g[0] : 0.0, g1 = g ? g[1] : 0.0, g2 = g ? g[2] : 0.0, g3 = g ? g[3] : 0.0;
    double cLL = 0, cLR = 0, cRL = 0, cRR = 0, dl = 0, dr = 0;
    for (int ch = 0; ch < 2; ++ch) {  // rms + crest factor
        const double mp = st[0][ch], mt = st[1][ch];
        const double rp = sqrt(fmax(mp, 1e-8)), rt = sqrt(fmax(mt, 1e-8));
        part[0] += (rp - rt) * (rp - rt);
        const double pk = st[0][4 + 2 * ch], pkt = st[1][4 + 2 * ch];
        const double ratp = pk / fmax(rp, 1e-8), ratt = pkt / fmax(rt, 1e-8);
        const double cfp = 20.0 * log10(fmax(ratp, 1e-8)), cft = 20.0 * log10(fmax(ratt, 1e-8));
        part[1] += (cfp - cft) * (cfp - cft);
        double d_rms = g0 * (double)a.weights[0] * 2.0 * (rp - rt) / (2.0 * a.bs);
        const double d_cf = g1 * (double)a.weights[1] * 2.0 * (cfp - cft) / (2.0 * a.bs);
        double d_pk = 0.0;
        if (ratp >= 1e-8) {
            d_pk = d_cf * c20 / pk;
            d_rms += -d_cf * c20 / rp;
        }
        const double e = (mp >= 1e-8) ? d_rms / (N * rp) : 0.0;  // d rms / d x = x / (N rms) when unclamped
        if (ch == 0) { cLL += e; dl = d_pk; } else { cRR += e; dr = d_pk; }
    }
    {  // stereo width = D / clamp(S),  D = mean (L-R)^2, S = mean (L+R)^2
        const double Sp = st[0][2], Dp = st[0][3], St = st[1][2], Dt = st[1][3];
        const double wp = Dp / fmax(Sp, 1e-8), wt = Dt / fmax(St, 1e-8);
        part[2] += (wp - wt) * (wp - wt);
        const double dw = g2 * (double)a.weights[2] * 2.0 * (wp - wt) / a.bs;
        const double dD = dw / fmax(Sp, 1e-8), dS = (Sp >= 1e-8) ? -dw * Dp / (Sp * Sp) : 0.0;
        cLL += (2.0 / N) * (dD + dS); cLR += (2.0 / N) * (-dD + dS);
        cRL += (2.0 / N) * (-dD + dS); cRR += (2.0 / N) * (dD + dS);
    }
    {  // stereo imbalance = (ER - EL) / clamp(ER + EL)
        const double ELp = st[0][0], ERp = st[0][1], ELt = st[1][0], ERt = st[1][1];
        const double Tp = ERp + ELp, Tt = ERt + ELt;
        const double ip = (ERp - ELp) / fmax(Tp, 1e-8), it = (ERt - ELt) / fmax(Tt, 1e-8);
        part[3] += (ip - it) * (ip - it);
        const double di = g3 * (double)a.weights[3] * 2.0 * (ip - it) / a.bs;
        const double Tc = fmax(Tp, 1e-8);
        const double dT = (Tp >= 1e-8) ? -di * (ERp - ELp) / (Tc * Tc) : 0.0;
        cLL += (-di / Tc + dT) * 2.0 / N;
        cRR += (di / Tc + dT) * 2.0 / N;
    }
    if (coef) {
        coef[0] = (float)cLL; coef[1] = (float)cLR; coef[2] = (float)cRL; coef[3] = (float)cRR;
        coef[4] = (float)dl; coef[5] = __int_as_float((int)st[0][5]);
        coef[6] = (float)dr; coef[7] = __int_as_float((int)st[0][7]);
    }
}

// ---- final: the five weighted MSE losses.  One 64-lane workgroup.
__global__ __launch_bounds__(64) void k_af_final(AfArgs a) {
    __shared__ double acc[5];
    const int tid = threadIdx.x;
    double part[5] = {0, 0, 0, 0, 0};
    for (int b = tid; b < a.bs; b += 64) af_features(a, b, nullptr, part, nullptr);
    // bark MSE over (bs, 24, 2): signals 0..bs-1 pred mid, bs..2bs-1 pred side, then the target's
    for (int i = tid; i < 2 * a.bs * kAfBands; i += 64) {
        const double d = (double)a.bark[i] - (double)a.bark[2 * a.bs * kAfBands + i];
        part[4] += d * d;
    }
    for (int q = 0; q < 5; ++q) {
        double v = part[q];
        for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
        if (tid == 0) acc[q] = v;
    }
    __syncthreads();
    if (tid == 0) {
        a.losses[0] = (float)(a.weights[0] * acc[0] / (2.0 * a.bs));
        a.losses[1] = (float)(a.weights[1] * acc[1] / (2.0 * a.bs));
        a.losses[2] = (float)(a.weights[2] * acc[2] / a.bs);
        a.losses[3] = (float)(a.weights[3] * acc[3] / a.bs);
        a.losses[4] = (float)(a.weights[4] * acc[4] / (2.0 * a.bs * kAfBands));
    }
}
__global__ __launch_bounds__(64) void k_af_coef(AfArgs a) {
    double part[4] = {0, 0, 0, 0};
    for (int b = threadIdx.x; b < a.bs; b += 64) af_features(a, b, a.grad_losses, part, a.coef + (int64_t)b * 16);
}

// ---- backward ---------------------------------------------------------------------------------------
// dM[s][k] = sum_j (w4 * 2 (B - Bt) / (2 bs 24)) / (lin_j + 1e-8) * fb[k][j] / F   for the 2*bs prediction signals
__global__ __launch_bounds__(256) void k_af_bark_dmag(AfArgs a) {
    __shared__ float cj[kAfBands];
    const int tid = threadIdx.x, s = blockIdx.y;  // s < 2*bs
    if (tid < kAfBands) {
        const float B = a.bark[(int64_t)s * kAfBands + tid], Bt = a.bark[(int64_t)(2 * a.bs + s) * kAfBands + tid];
        const float lin = a.bark[(int64_t)(4 * a.bs + s) * kAfBands + tid];
        cj[tid] = a.grad_losses[4] * a.weights[4] * 2.0f * (B - Bt) / (2.0f * a.bs * kAfBands) / (lin + 1e-8f) / (float)a.n_frames;
    }
    __syncthreads();
    const int k = blockIdx.x * 256 + tid;
    if (k < kAfBins) {
        const float* fr = a.fb + (int64_t)k * kAfBands;
        float v = 0.0f;
#pragma unroll
        for (int j = 0; j < kAfBands; ++j) v = fmaf(cj[j], fr[j], v);
        a.meanmag[(int64_t)(4 * a.bs + s) * kAfBins + k] = v;  // stored after the 4*bs mean-magnitude rows
    }
}

// bark adjoint: one frame of one prediction signal per workgroup
__global__ __launch_bounds__(kAfThreads) void k_af_bark_bwd(AfArgs a) {
    __shared__ __attribute__((aligned(16))) float2 buf[kAfM];
    __shared__ AfTw T;
    const int tid = threadIdx.x, s = blockIdx.y, f = blockIdx.x;  // s < 2*bs
    const float2* twM = reinterpret_cast<const float2*>(a.tables + kAfTwM);
    const float2* twN = reinterpret_cast<const float2*>(a.tables + kAfTwN);
    const float* win = a.tables + kAfWin;
    for (int i = tid; i < kAfM / 64; i += kAfThreads) T.coarse[i] = twM[i * 64];
    for (int i = tid; i < 64; i += kAfThreads) T.fine[i] = twM[i];
    const float *l, *r;
    float sign;
    af_signal(a, s, l, r, sign);
    const float* dM = a.meanmag + (int64_t)(4 * a.bs + s) * kAfBins;
    __syncthreads();
    af_load_frame<true>(buf, l, r, sign, win, f, a.n, tid);
    __syncthreads();
#ifndef MST_AF_ABLATE
#define MST_AF_ABLATE 0  // timing ablations (wrong results): 2 no inverse transform, 3 no transforms
#endif
    if (MST_AF_ABLATE < 3) fft16k_dif<true>(buf, T, tid);
    // For each mirror pair (k, M-k): G = dM * X / |X|, Hermitian extension H (H[k] = G[k]/2 inside,
    // real at 0 and M), then the half-size packing  A[k] = H[k] + conj(H[M-k]),
    // Bq[k] = (H[k] - conj(H[M-k])) conj(W^k);  slot(k) <- conj(A + i Bq)  (inverse = conj FFT conj).
    for (int j = 0; j <= 8; ++j) {
        const int k = tid + kAfThreads * j;
        if (k > kAfM / 2) break;
        float2 Xk, Xm;
        af_untangle<true>(buf, twN, k, Xk, Xm);
        const float ak = sqrtf(Xk.x * Xk.x + Xk.y * Xk.y), am = sqrtf(Xm.x * Xm.x + Xm.y * Xm.y);
        const float gk = ak > 0.f ? dM[k] / ak : 0.f, gm = am > 0.f ? dM[kAfM - k] / am : 0.f;
        float2 Hk = make_float2(gk * Xk.x, gk * Xk.y), Hm = make_float2(gm * Xm.x, gm * Xm.y);  // G[k], G[M-k]
        if (k == 0) {
            Hk = make_float2(Hk.x, 0.f);  // H[0] = Re G[0]
            Hm = make_float2(Hm.x, 0.f);  // H[M] = Re G[M]
        } else {
            Hk = make_float2(0.5f * Hk.x, 0.5f * Hk.y);
            Hm = make_float2(0.5f * Hm.x, 0.5f * Hm.y);
        }
        const float2 w = twN[k];             // W^k ; W^(M-k) = -conj(W^k)
        // index k:    A = Hk + conj(Hm) ;  Bq = (Hk - conj(Hm)) conj(W^k)
        const float2 Ak = make_float2(Hk.x + Hm.x, Hk.y - Hm.y);
        const float2 Bk = cmulf(make_float2(Hk.x - Hm.x, Hk.y + Hm.y), cconj(w));
        // index M-k:  A = Hm + conj(Hk) ;  Bq = (Hm - conj(Hk)) conj(W^(M-k)) = (Hm - conj(Hk)) (-W^k)
        const float2 Am = make_float2(Hm.x + Hk.x, Hm.y - Hk.y);
        const float2 Bm = cmulf(make_float2(Hm.x - Hk.x, Hm.y + Hk.y), make_float2(-w.x, -w.y));
        // value = A + i Bq = (A.x - B.y, A.y + B.x); store its conjugate
        const int pk = swz(rev4_7(k & (kAfM - 1))), pm = swz(rev4_7((kAfM - k) & (kAfM - 1)));
        if (k == 0) {
            // slot 0 combines H[0] and H[M]: A[0] = H[0] + H[M], Bq[0] = H[0] - H[M] (both real)
            buf[pk] = make_float2(Hk.x + Hm.x, -(Hk.x - Hm.x));
        } else {
            buf[pk] = make_float2(Ak.x - Bk.y, -(Ak.y + Bk.x));
            if (k != kAfM / 2) buf[pm] = make_float2(Am.x - Bm.y, -(Am.y + Bm.x));
        }
    }
    __syncthreads();
    if (MST_AF_ABLATE < 2) fft16k_dit(buf, T, tid);
    // conj(result) = y_even + i y_odd; the windowed frame goes to its own slot of `yframes` (coalesced 8-byte stores) -
    // k_af_bwd_gather overlap-adds the frames, owner-computes (no atomics: the gradient is bitwise reproducible)
    float2* yf = reinterpret_cast<float2*>(a.yframes + ((int64_t)s * a.n_frames + f) * kAfFft);
    for (int m = tid; m < kAfM; m += kAfThreads) {
        const float2 v = buf[swz(m)];
        const float2 w = *reinterpret_cast<const float2*>(win + 2 * m);
        yf[m] = make_float2(w.x * v.x, -w.y * v.y);
    }
}

// grad_pred = closed-form features (energy-type + crest, elementwise) + overlap-add of the Bark adjoint frames.
// Sample i of the reflect-padded signal sits at padded position p = i + N/2; frame f covers [f hop, f hop + N).  A sample
// collects every frame over its own position and, within N/2 of an end, over its mirror position (torch reflect padding).
// grid (blocks, bs), 4 samples per lane.
__device__ __forceinline__ void af_frames_over(int64_t p, int n_frames, int& f_lo, int& f_hi) {
    const int64_t lo = p - kAfFft + 1;
    f_lo = lo <= 0 ? 0 : (int)((lo + kAfHop - 1) / kAfHop);
    const int64_t hi = p / kAfHop;
    f_hi = hi > n_frames - 1 ? n_frames - 1 : (int)hi;
}
__global__ __launch_bounds__(256) void k_af_bwd_gather(AfArgs a) {
    const int b = blockIdx.y;
    const float* c = a.coef + (int64_t)b * 16;
    const float* l = a.pred + (int64_t)b * 2 * a.n;
    const float* r = l + a.n;
    float* gl = a.grad_pred + (int64_t)b * 2 * a.n;
    float* gr = gl + a.n;
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= a.n) return;
    const float4 lv = load4(l, i, a.n), rv = load4(r, i, a.n);
    const float le[4] = {lv.x, lv.y, lv.z, lv.w}, re[4] = {rv.x, rv.y, rv.z, rv.w};
    float ol[4], orr[4];
    const int al = __float_as_int(c[5]), ar = __float_as_int(c[7]);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        ol[t] = c[0] * le[t] + c[1] * re[t];
        orr[t] = c[2] * le[t] + c[3] * re[t];
        if (i + t == al) ol[t] += c[4] * ((le[t] > 0.f) - (le[t] < 0.f));
        if (i + t == ar) orr[t] += c[6] * ((re[t] > 0.f) - (re[t] < 0.f));
    }
    const float* ym = a.yframes + (int64_t)b * a.n_frames * kAfFft;            // mid  = L + R
    const float* ys = a.yframes + (int64_t)(a.bs + b) * a.n_frames * kAfFft;   // side = L - R
    float m[4] = {0.f, 0.f, 0.f, 0.f}, sd[4] = {0.f, 0.f, 0.f, 0.f};
    {   // own position: the four samples share their frames (hop and frame length are multiples of 4)
        const int64_t p = i + kAfFft / 2;
        int f_lo, f_hi;
        af_frames_over(p, a.n_frames, f_lo, f_hi);
        for (int f = f_lo; f <= f_hi; ++f) {
            const int64_t at = (int64_t)f * kAfFft + (p - (int64_t)f * kAfHop);
            const float4 vm = *reinterpret_cast<const float4*>(ym + at), vs = *reinterpret_cast<const float4*>(ys + at);
            m[0] += vm.x; m[1] += vm.y; m[2] += vm.z; m[3] += vm.w;
            sd[0] += vs.x; sd[1] += vs.y; sd[2] += vs.z; sd[3] += vs.w;
        }
    }
    if (i <= kAfFft / 2 || i + 3 >= a.n - 1 - kAfFft / 2) {  // mirror positions near the two ends
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int64_t it = i + t;
            if (it >= a.n) break;
            for (int side = 0; side < 2; ++side) {
                int64_t p;
                if (side == 0) {
                    if (it < 1 || it > kAfFft / 2) continue;
                    p = kAfFft / 2 - it;
                } else {
                    if (it < a.n - 1 - kAfFft / 2 || it > a.n - 2) continue;
                    p = kAfFft / 2 + 2 * (a.n - 1) - it;
                }
                int f_lo, f_hi;
                af_frames_over(p, a.n_frames, f_lo, f_hi);
                for (int f = f_lo; f <= f_hi; ++f) {
                    const int64_t at = (int64_t)f * kAfFft + (p - (int64_t)f * kAfHop);
                    m[t] += ym[at];
                    sd[t] += ys[at];
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        ol[t] += m[t] + sd[t];
        orr[t] += m[t] - sd[t];
    }
    store4(gl, i, a.n, make_float4(ol[0], ol[1], ol[2], ol[3]));
    store4(gr, i, a.n, make_float4(orr[0], orr[1], orr[2], orr[3]));
}

__global__ void k_af_tables(float* tables) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < kAfM) {
        const double ang = 6.283185307179586476925 * (double)t / (double)kAfM;
        tables[kAfTwM + 2 * t] = (float)cos(ang);
        tables[kAfTwM + 2 * t + 1] = (float)(-sin(ang));
    }
    if (t <= kAfM) {
        const double ang = 6.283185307179586476925 * (double)t / (double)kAfFft;
        tables[kAfTwN + 2 * t] = (float)cos(ang);
        tables[kAfTwN + 2 * t + 1] = (float)(-sin(ang));
    }
    if (t < kAfFft) {
        const float ph = 6.283185307179586f * (float)t / (float)kAfFft;  // torch.hann_window(32768), periodic
        tables[kAfWin + t] = 0.5f - 0.5f * (float)cos((double)ph);
    }
}

struct AfPlan {
    int n_frames, n_groups, n_statblk;
    int64_t magpart, meanmag, bark, statpart, bandpart, stats, coef, yframes, total;
    bool ok;
};
// frames of a signal are cut into G strips (one workgroup each, one 128 KiB workgroup per CU, 256 CUs): choose G to
// minimise  rounds x longest strip  = ceil(G S / 256) x ceil(F / G);  e.g. F = 33, S = 32: G = 8 -> 1 x 5 (G = 9: 2 x 4)
static int af_groups(int n_frames, int n_signals) {
    int best = 1;
    int64_t best_cost = INT64_MAX;
    for (int g = 1; g <= n_frames; ++g) {
        const int64_t rounds = ((int64_t)g * n_signals + 255) / 256, len = (n_frames + g - 1) / g;
        const int64_t cost = rounds * len * 1024 + g;  // ties: fewer partial-sum rows
        if (cost < best_cost) { best_cost = cost; best = g; }
    }
    return best;
}
static AfPlan af_plan(int bs, int64_t n) {
    AfPlan p{};
    p.ok = bs > 0 && n > kAfFft / 2;
    if (!p.ok) return p;
    p.n_frames = 1 + (int)(n / kAfHop);
    p.n_groups = af_groups(p.n_frames, 4 * bs);
    p.n_statblk = (int)((n + kAfStatSpan - 1) / kAfStatSpan);
    int64_t o = 0;
    auto take = [&](int64_t k) { int64_t at = o; o += round_up(k, 64); return at; };
    p.magpart = take((int64_t)4 * bs * p.n_groups * kAfBins);
    p.meanmag = take((int64_t)6 * bs * kAfBins);  // 4*bs mean magnitudes + 2*bs bark cotangents
    p.bark = take((int64_t)8 * bs * kAfBands);    // log energies, then linear energies
    p.statpart = take((int64_t)2 * bs * p.n_statblk * 8);
    p.bandpart = take((int64_t)4 * bs * kAfBinSlices * kAfBands);
    p.stats = take((int64_t)2 * bs * 8 * 2);  // doubles
    p.coef = take((int64_t)bs * 16);
    p.yframes = take((int64_t)2 * bs * p.n_frames * kAfFft);  // backward only
    p.total = o;
    return p;
}
static AfArgs af_args(const AfPlan& p, int bs, int64_t n, const float* pred, const float* target, const float* tables,
                      const float* fb, const float* weights, float* ws) {
    AfArgs a{};
    a.pred = pred; a.target = target; a.tables = tables; a.fb = fb;
    a.magpart = ws + p.magpart; a.meanmag = ws + p.meanmag; a.bark = ws + p.bark;
    a.statpart = ws + p.statpart; a.coef = ws + p.coef;
    a.yframes = ws + p.yframes;
    a.bandpart = ws + p.bandpart; a.stats = reinterpret_cast<double*>(ws + p.stats);
    for (int i = 0; i < 5; ++i) a.weights[i] = weights[i];
    a.bs = bs; a.n_frames = p.n_frames; a.n_groups = p.n_groups; a.n_statblk = p.n_statblk; a.n = n;
    return a;
}
}  // namespace mst

using namespace mst;

extern "C" size_t mst_afloss_tables_bytes(void) { return (size_t)kAfTablesFloats * sizeof(float); }
extern "C" int mst_afloss_init_tables(void* tables, void* stream_) {
    if (!tables) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_af_tables, dim3((kAfFft + 255) / 256), dim3(256), 0, (hipStream_t)stream_, (float*)tables);
    return (int)hipGetLastError();
}
extern "C" size_t mst_afloss_workspace_bytes(int32_t bs, int64_t n_samples) {
    const AfPlan p = af_plan(bs, n_samples);
    return p.ok ? (size_t)p.total * sizeof(float) : 0;
}
extern "C" int mst_afloss_forward(const float* pred, const float* target, int32_t bs, int64_t n_samples, const float* weights5,
                                  const void* tables, const float* filterbank, float* losses5, void* workspace,
                                  size_t workspace_bytes, void* stream_) {
    const AfPlan p = af_plan(bs, n_samples);
    if (!p.ok || !pred || !target || !weights5 || !tables || !filterbank || !losses5 || !workspace) return hipErrorInvalidValue;
    if (workspace_bytes < (size_t)p.total * sizeof(float)) return hipErrorInvalidValue;
    hipStream_t stream = (hipStream_t)stream_;
    AfArgs a = af_args(p, bs, n_samples, pred, target, (const float*)tables, filterbank, weights5, (float*)workspace);
    a.losses = losses5;
    hipLaunchKernelGGL(k_af_stats, dim3(p.n_statblk, 2 * bs), dim3(256), 0, stream, a);
    hipLaunchKernelGGL(k_af_bark_fwd, dim3(p.n_groups, 4 * bs), dim3(kAfThreads), 0, stream, a);
    hipLaunchKernelGGL(k_af_bark_reduce, dim3(kAfBinSlices, 4 * bs), dim3(256), 0, stream, a);
    hipLaunchKernelGGL(k_af_finish, dim3(4 * bs), dim3(64), 0, stream, a);
    hipLaunchKernelGGL(k_af_final, dim3(1), dim3(64), 0, stream, a);
    return (int)hipGetLastError();
}
extern "C" int mst_afloss_backward(const float* pred, const float* target, int32_t bs, int64_t n_samples, const float* weights5,
                                   const void* tables, const float* filterbank, const float* grad_losses5, float* grad_pred,
                                   void* workspace, size_t workspace_bytes, void* stream_) {
    const AfPlan p = af_plan(bs, n_samples);
    if (!p.ok || !pred || !target || !weights5 || !tables || !filterbank || !grad_losses5 || !grad_pred || !workspace)
        return hipErrorInvalidValue;
    if (workspace_bytes < (size_t)p.total * sizeof(float)) return hipErrorInvalidValue;
    hipStream_t stream = (hipStream_t)stream_;
    AfArgs a = af_args(p, bs, n_samples, pred, target, (const float*)tables, filterbank, weights5, (float*)workspace);
    a.grad_losses = grad_losses5;
    a.grad_pred = grad_pred;
    hipLaunchKernelGGL(k_af_coef, dim3(1), dim3(64), 0, stream, a);
    hipLaunchKernelGGL(k_af_bark_dmag, dim3((kAfBins + 255) / 256, 2 * bs), dim3(256), 0, stream, a);
    hipLaunchKernelGGL(k_af_bark_bwd, dim3(p.n_frames, 2 * bs), dim3(kAfThreads), 0, stream, a);
    hipLaunchKernelGGL(k_af_bwd_gather, dim3((unsigned)((n_samples + 1023) / 1024), bs), dim3(256), 0, stream, a);
    return (int)hipGetLastError();
}
