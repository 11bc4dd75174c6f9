// mst_af2.hip - Bark-spectrum transforms of AudioFeatureLoss (reference mst/loss.py:62-115: mid/side -> STFT(32768, hop 8192,
// periodic Hann, reflect) -> |X| -> mean over frames) and their adjoint, on the register-radix engine of mst_fft2.h.
//
// The 32768-point real transform of a frame is the 16384-point complex transform of z[m] = w x[2m] + i w x[2m+1]; that one is
// split once more by decimation in frequency,
//     Z[2q]     = FFT_8192( z[i] + z[i + 8192] )[q]                 (half 0: the even bins)
//     Z[2q + 1] = FFT_8192( (z[i] - z[i + 8192]) W_16384^i )[q]     (half 1: the odd bins)
// and the real-input untangling pairs bin k with M - k, which have the same parity - so the two halves never meet: each is
// one 8192-point transform of the engine (512 lanes, 64 KiB of LDS, two workgroups per CU) with its own bins, its own
// magnitude sums and its own cotangents.  The adjoint runs the same split as decimation in time:
//     Y[m], Y[m + 8192] = P0[m] +- W_16384^m P1[m],   Ph = FFT_8192( V[2q + h] )
// one workgroup per frame does half 0, parks P0 in the frame's slab of `yframes`, does half 1 and writes the windowed frame.
// (Round 1 ran a 7-pass radix-4 transform of the whole 16384 points in 128 KiB of LDS, one workgroup per CU: 36 us per frame.)
#include "mst_af.h"
#include "mst_fft2.h"

namespace mst {

using AfS = FftShape<8192>;
constexpr int kAf2Lanes = 512;
#ifndef MST_AF2_W
#define MST_AF2_W 4      // min waves per SIMD asked of the forward kernel (4 = two workgroups per CU)
#endif
#ifndef MST_AF2_W_BWD
#define MST_AF2_W_BWD 4  // one half per launch: 121-124 registers, two workgroups per CU
#endif

// |X| and g / |X| on the hardware's 1-ulp v_sqrt_f32 / v_rsq_f32 (the correctly rounded sqrtf and '/' are 10 + 11 instructions per bin
// with a denormal guard; -DMST_AF_PRECISE_MAG=1 restores them).  |X|^2 below 1e-30 - |X| < 1e-15, far under fp32 round-off of any
// audible frame - counts as zero in the gradient, as |X| = 0 did.
#ifndef MST_AF_PRECISE_MAG
#define MST_AF_PRECISE_MAG 0
#endif
__device__ __forceinline__ float af_mag(float2 X) {
    const float p2 = X.x * X.x + X.y * X.y;
    return MST_AF_PRECISE_MAG ? sqrtf(p2) : __builtin_amdgcn_sqrtf(p2);
}
__device__ __forceinline__ float af_over_mag(float g, float2 X) {
    const float p2 = X.x * X.x + X.y * X.y;
    if (MST_AF_PRECISE_MAG) {
        const float m = sqrtf(p2);
        return m > 0.f ? g / m : 0.f;
    }
    return p2 > 1e-30f ? g * __builtin_amdgcn_rsqf(p2) : 0.f;
}
// W_32^t = (cos, -sin)(2 pi t / 32), t < 16: W_16384^(lane + 512 t) = W_16384^lane W_32^t
__device__ __forceinline__ float2 w32(int t) {
    constexpr float c[16] = {1.0f, 0.98078528040323043f, 0.92387953251128674f, 0.83146961230254524f, 0.70710678118654752f,
                             0.55557023301960218f, 0.38268343236508977f, 0.19509032201612825f, 0.0f, -0.19509032201612825f,
                             -0.38268343236508977f, -0.55557023301960218f, -0.70710678118654752f, -0.83146961230254524f,
                             -0.92387953251128674f, -0.98078528040323043f};
    constexpr float sn[16] = {0.0f, 0.19509032201612825f, 0.38268343236508977f, 0.55557023301960218f, 0.70710678118654752f,
                              0.83146961230254524f, 0.92387953251128674f, 0.98078528040323043f, 1.0f, 0.98078528040323043f,
                              0.92387953251128674f, 0.83146961230254524f, 0.70710678118654752f, 0.55557023301960218f,
                              0.38268343236508977f, 0.19509032201612825f};
    return make_float2(c[t], -sn[t]);
}

__device__ __forceinline__ int af2_reflect(int i, int n) {
    i = i < 0 ? -i : i;
    return i >= n ? 2 * (n - 1) - i : i;
}
// element q of a half's 8192-point sequence in the engine's output layout (fft8192_from): even q in buf[0], odd q in buf[1]
__device__ __forceinline__ float2& af2_at(float2 (*buf)[AfS::SLOTS], int q) { return buf[q & 1][AfS::slot(q >> 1)]; }

// X[k], X[M-k] of the real frame from the packed spectrum values z[k], z[M-k] and w = W_N^k
__device__ __forceinline__ void af2_untangle(float2 zk, float2 zm, float2 w, float2& Xk, float2& Xm) {
    const float2 E = make_float2(0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y));   // even-sample spectrum
    const float2 O = make_float2(0.5f * (zk.y + zm.y), -0.5f * (zk.x - zm.x));  // odd-sample spectrum
    const float2 wo = cmul(w, O);
    Xk = make_float2(E.x + wo.x, E.y + wo.y);
    Xm = make_float2(E.x - wo.x, -(E.y - wo.y));  // X[M-k] = conj(E - W^k O)
}

// Forward transform of half HALF of frame f of the signal l + sign r: afterwards (the caller adds one barrier) the packed
// spectrum value Z[2q + HALF] is af2_at(buf, q).  wm = W_16384^lane, wl = W_8192^lane.
// INSIDE (whole frame inside the row, 8-byte aligned rows - all but the two reflected frames at each end): straight-line code,
// every load is (uniform base of t) + lane; a branch per element instead makes the register allocator spill across 16 diamonds.
template <int HALF, bool INSIDE>
__device__ __forceinline__ void af2_forward_path(float2 (*buf)[AfS::SLOTS], const float* __restrict__ l, const float* __restrict__ r,
                                                 float sign, const float* __restrict__ win, int start, int n,
                                                 const LaneTw<8192>& tw, float2 wl, float2 wm, int lane) {
    const float2* lp = reinterpret_cast<const float2*>(l + (INSIDE ? start : 0));
    const float2* rp = reinterpret_cast<const float2*>(r + (INSIDE ? start : 0));
    const float2* wp = reinterpret_cast<const float2*>(win);
    fft8192_from<false>([&](int t) {
        // element i = lane + 512 t of the half's sequence: z[i] +- z[i + 8192]
        const unsigned ul = (unsigned)lane, i = ul + (unsigned)(kAf2Lanes * t);
        const float2 w = (wp + kAf2Lanes * t)[ul];  // hann(j + N/2) = 1 - hann(j)
        float2 xa, xb;
        if (INSIDE) {
            const float2 la = (lp + kAf2Lanes * t)[ul], ra = (rp + kAf2Lanes * t)[ul];
            const float2 lb = (lp + kAf2Lanes * t + kAfHalf)[ul], rb = (rp + kAf2Lanes * t + kAfHalf)[ul];
            xa = make_float2(la.x + sign * ra.x, la.y + sign * ra.y);
            xb = make_float2(lb.x + sign * rb.x, lb.y + sign * rb.y);
        } else {
            const int j = start + 2 * (int)i;
            const unsigned a0 = af2_reflect(j, n), a1 = af2_reflect(j + 1, n);
            const unsigned b0 = af2_reflect(j + kAfM, n), b1 = af2_reflect(j + kAfM + 1, n);
            xa = make_float2(l[a0] + sign * r[a0], l[a1] + sign * r[a1]);
            xb = make_float2(l[b0] + sign * r[b0], l[b1] + sign * r[b1]);
        }
        const float2 za = make_float2(w.x * xa.x, w.y * xa.y), zb = make_float2((1.0f - w.x) * xb.x, (1.0f - w.y) * xb.y);
        return HALF == 0 ? cadd(za, zb) : cmul(csub(za, zb), cmul(wm, w32(t)));
    }, buf[0], buf[1], tw, wl, lane);
}
template <int HALF>
__device__ __forceinline__ void af2_forward(float2 (*buf)[AfS::SLOTS], const float* __restrict__ l, const float* __restrict__ r,
                                            float sign, const float* __restrict__ win, int f, int n, bool fast,
                                            const LaneTw<8192>& tw, float2 wl, float2 wm, int lane) {
    const int start = f * kAfHop - kAfFft / 2;
    if (fast && start >= 0 && start + kAfFft <= n) af2_forward_path<HALF, true>(buf, l, r, sign, win, start, n, tw, wl, wm, lane);
    else af2_forward_path<HALF, false>(buf, l, r, sign, win, start, n, tw, wl, wm, lane);
}

// ---- forward: magnitude sums over a strip of frames, one half of the bins per workgroup -----------------------------------------
template <int HALF>
__device__ __forceinline__ void af2_fwd_body(const AfArgs& a, float2 (*buf)[AfS::SLOTS], int grp, int s) {
    const int lane = threadIdx.x;
    const float2* twH = reinterpret_cast<const float2*>(a.tables + kAfTwH);
    const float2* twN = reinterpret_cast<const float2*>(a.tables + kAfTwN);
    const float* win = a.tables + kAfWin;
    LaneTw<8192> tw;
    tw.init(twH, lane);
    const float *l, *r;
    float sign;
    af_signal(a, s, l, r, sign);
    const int n = (int)a.n;  // a row is < 2^31 samples
    const bool fast = !(((uintptr_t)l | (uintptr_t)r) & 7);
    // lane owns the bin pairs (k, M - k), k = 2 q + HALF, q = lane + 512 j (j < 8); half 0, lane 0 also owns k = M/2
    float acc_lo[8], acc_hi[8], acc_mid = 0.0f;
#pragma unroll
    for (int j = 0; j < 8; ++j) acc_lo[j] = acc_hi[j] = 0.0f;
    const int f0 = (int)(((int64_t)grp * a.n_frames) / a.n_groups), f1 = (int)(((int64_t)(grp + 1) * a.n_frames) / a.n_groups);
    for (int f = f0; f < f1; ++f) {
        int li = lane;
        asm volatile("" : "+v"(li));  // the two per-lane twiddles are re-fetched per frame (L1 hits) instead of living in registers
        const float2 wl = twH[li], wm = reinterpret_cast<const float2*>(a.tables + kAfTwM)[li];
        lds_barrier();  // the previous frame's spectrum has been read
        af2_forward<HALF>(buf, l, r, sign, win, f, n, fast, tw, wl, wm, li);
        lds_barrier();
        // (li, not lane: the bin addresses are rebuilt per frame - hoisted out of the loop they would pin 32 registers)
#pragma unroll 2
        for (int j = 0; j < 8; ++j) {
            const int q = li + kAf2Lanes * j;
            const int qm = HALF == 0 ? ((kAfHalf - q) & (kAfHalf - 1)) : kAfHalf - 1 - q;
            float2 Xk, Xm;
            af2_untangle(af2_at(buf, q), af2_at(buf, qm), twN[2 * q + HALF], Xk, Xm);
            acc_lo[j] += af_mag(Xk);
            acc_hi[j] += af_mag(Xm);  // k = 0 -> bin M (Nyquist)
        }
        if (HALF == 0 && lane == 0) {
            const float2 z = af2_at(buf, kAfHalf / 2);  // k = M/2 pairs with itself
            float2 Xk, Xm;
            af2_untangle(z, z, twN[kAfM / 2], Xk, Xm);
            acc_mid += af_mag(Xk);
        }
    }
    float* out = a.magpart + ((int64_t)s * a.n_groups + grp) * kAfBins;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = 2 * (lane + kAf2Lanes * j) + HALF;
        out[k] = acc_lo[j];
        out[kAfM - k] = acc_hi[j];  // k = 0 writes bin M
    }
    if (HALF == 0 && lane == 0) out[kAfM / 2] = acc_mid;
}
__global__ __launch_bounds__(kAf2Lanes, MST_AF2_W) void k_af2_bark_fwd(AfArgs a) {
    __shared__ __attribute__((aligned(16))) float2 buf[2][AfS::SLOTS];
    // The four workgroups that read the same stereo rows (mid / side x even / odd bins of one strip of frames) are walked onto ONE XCD,
    // next to each other: workgroup id L lands on XCD L % 8 (mst_common.h: row_block_xcd), and with the plain (strip, signal, half) grid
    // the four sat on four XCDs - every frame (256 KB) crossed the fabric four times, 2.2 GB per launch at bs 32, against an L2 that
    // 64 resident workgroups x 256 KB overflow anyway.  Any other batch size keeps the plain walk; the result does not depend on it.
#ifndef MST_AF2_XCD
#define MST_AF2_XCD 1
#endif
    int grp = blockIdx.x, sgn = blockIdx.y, half = blockIdx.z;
    if (MST_AF2_XCD && (2 * a.bs) % 8 == 0) {
        const int L = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), xcd = L & 7, k = L >> 3;
        const int W = 4 * a.n_groups, src = (k / W) * 8 + xcd, rem = k % W, var = rem & 3;  // src = (pred | target, batch item)
        grp = rem >> 2;
        half = var >> 1;
        sgn = (2 * (src / a.bs) + (var & 1)) * a.bs + src % a.bs;
    }
    if (half == 0) af2_fwd_body<0>(a, buf, grp, sgn);
    else af2_fwd_body<1>(a, buf, grp, sgn);
}

// ---- backward: one frame of one prediction signal per workgroup -----------------------------------------------------------------
// For each mirror pair (k, M-k): G = dM X / |X|, Hermitian extension H (H[k] = G[k]/2 inside, real at 0 and M), then the packing
// of the half-size inverse  A[k] = H[k] + conj(H[M-k]),  Bq[k] = (H[k] - conj(H[M-k])) conj(W_N^k);  V[k] = conj(A + i Bq)
// (inverse = conj FFT conj).  Y = FFT_16384(V);  conj(Y[m]) = y[2m] + i y[2m+1] is the frame before the window.
template <int HALF>
__device__ __forceinline__ void af2_cotangent(const AfArgs& a, float2 (*buf)[AfS::SLOTS], const float* __restrict__ dM,
                                              const float2* __restrict__ twN, int lane) {
#pragma unroll 2
    for (int j = 0; j < (HALF == 0 ? 9 : 8); ++j) {
        if (j == 8 && lane != 0) break;
        const int q = j == 8 ? kAfHalf / 2 : lane + kAf2Lanes * j;
        const int qm = HALF == 0 ? ((kAfHalf - q) & (kAfHalf - 1)) : kAfHalf - 1 - q;
        const int k = 2 * q + HALF;
        const float2 w = twN[k];  // W^k ; W^(M-k) = -conj(W^k)
        float2 Xk, Xm;
        af2_untangle(af2_at(buf, q), af2_at(buf, qm), w, Xk, Xm);
        const float gk = af_over_mag(dM[k], Xk), gm = af_over_mag(dM[kAfM - k], Xm);
        float2 Hk = make_float2(gk * Xk.x, gk * Xk.y), Hm = make_float2(gm * Xm.x, gm * Xm.y);  // G[k], G[M-k]
        if (HALF == 0 && k == 0) {
            // slot 0 combines H[0] = Re G[0] and H[M] = Re G[M]: A[0] = H[0] + H[M], Bq[0] = H[0] - H[M] (both real)
            af2_at(buf, 0) = make_float2(Hk.x + Hm.x, -(Hk.x - Hm.x));
            continue;
        }
        Hk = make_float2(0.5f * Hk.x, 0.5f * Hk.y);
        Hm = make_float2(0.5f * Hm.x, 0.5f * Hm.y);
        // index k:    A = Hk + conj(Hm) ;  Bq = (Hk - conj(Hm)) conj(W^k)
        const float2 Ak = make_float2(Hk.x + Hm.x, Hk.y - Hm.y);
        const float2 Bk = cmul(make_float2(Hk.x - Hm.x, Hk.y + Hm.y), cconj(w));
        // index M-k:  A = Hm + conj(Hk) ;  Bq = (Hm - conj(Hk)) conj(W^(M-k)) = (Hm - conj(Hk)) (-W^k)
        const float2 Am = make_float2(Hm.x + Hk.x, Hm.y - Hk.y);
        const float2 Bm = cmul(make_float2(Hm.x - Hk.x, Hm.y + Hk.y), make_float2(-w.x, -w.y));
        // value = A + i Bq = (A.x - B.y, A.y + B.x); store its conjugate.  (q, qm) belong to this lane alone: in place.
        af2_at(buf, q) = make_float2(Ak.x - Bk.y, -(Ak.y + Bk.x));
        if (q != qm) af2_at(buf, qm) = make_float2(Am.x - Bm.y, -(Am.y + Bm.x));
    }
}

template <int HALF>
__device__ __forceinline__ void af2_bwd_half(const AfArgs& a, float2 (*buf)[AfS::SLOTS], const float* l, const float* r, float sign,
                                             int f, int s, int n, bool fast, const LaneTw<8192>& tw, int lane) {
    const float2* twH = reinterpret_cast<const float2*>(a.tables + kAfTwH);
    const float2* twN = reinterpret_cast<const float2*>(a.tables + kAfTwN);
    const float* win = a.tables + kAfWin;
    const float* dM = a.meanmag + (int64_t)(4 * a.bs + s) * kAfBins;
    int li = lane;
    asm volatile("" : "+v"(li));
    const float2 wl = twH[li], wm = reinterpret_cast<const float2*>(a.tables + kAfTwM)[li];
    af2_forward<HALF>(buf, l, r, sign, win, f, n, fast, tw, wl, wm, lane);
    lds_barrier();
    af2_cotangent<HALF>(a, buf, dM, twN, lane);
    lds_barrier();
    // the second transform reads its inputs straight from the buffers it is about to overwrite (barrier inside)
    fft8192_from<false, true>([&](int t) { return af2_at(buf, lane + kAf2Lanes * t); }, buf[0], buf[1], tw, wl, lane);
    lds_barrier();
    // Ph[m] = af2_at(buf, m).  Half 0 parks P0 in the frame's slab; half 1 reads it back (the same lane wrote it), combines
    // Y[m], Y[m + 8192] = P0 +- W_16384^m P1 and stores the windowed frame  y[2m] = w Re Y[m],  y[2m+1] = -w Im Y[m].
    float2* yf = reinterpret_cast<float2*>(a.yframes + ((int64_t)s * a.n_frames + f) * kAfFft);
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const int m = lane + kAf2Lanes * t;
        const float2 p = af2_at(buf, m);
        if (HALF == 0) {
            yf[m] = p;
        } else {
            const float2 p0 = yf[m], tp = cmul(cmul(wm, w32(t)), p);
            const float2 ya = cadd(p0, tp), yb = csub(p0, tp);
            const float2 w = *reinterpret_cast<const float2*>(win + 2 * m);
            yf[m] = make_float2(w.x * ya.x, -w.y * ya.y);
            yf[m + kAfHalf] = make_float2((1.0f - w.x) * yb.x, -(1.0f - w.y) * yb.y);
        }
    }
}
// One launch per half (half 0 parks P0 in the frame's slab, half 1 reads it back).  Both halves in one kernel take 229 registers
// (85 spills when capped at 128): the compiler interleaves them, half 1's 80 frame loads float up into half 0's transforms.  One
// half alone fits 128 registers without a spill = two workgroups per CU, and 2 x ceil(frames / 512) half-length rounds beat
// ceil(frames / 256) full ones (bs 32: 2112 frames = 2 x 5 half rounds instead of 9 whole ones).
template <int HALF>
__global__ __launch_bounds__(kAf2Lanes, MST_AF2_W_BWD) void k_af2_bark_bwd(AfArgs a) {
    __shared__ __attribute__((aligned(16))) float2 buf[2][AfS::SLOTS];
    int f = blockIdx.x, s = blockIdx.y;  // s < 2*bs
    if (MST_AF2_XCD && a.bs % 8 == 0) {
        // the 2 x n_frames workgroups of one batch item (mid and side of every frame: frames overlap 4x) on one XCD, frames ascending: the
        // resident set of an XCD is then a window of ~32 consecutive frames of one stereo pair = 2 MB, which its L2 holds
        const int L = blockIdx.x + gridDim.x * blockIdx.y, xcd = L & 7, k = L >> 3, per = 2 * a.n_frames;
        const int b = (k / per) * 8 + xcd, rem = k % per;
        f = rem >> 1;
        s = (rem & 1) * a.bs + b;
    }
    const int lane = threadIdx.x;
    LaneTw<8192> tw;
    tw.init(reinterpret_cast<const float2*>(a.tables + kAfTwH), lane);
    const float *l, *r;
    float sign;
    af_signal(a, s, l, r, sign);
    const int n = (int)a.n;
    const bool fast = !(((uintptr_t)l | (uintptr_t)r) & 7);
    af2_bwd_half<HALF>(a, buf, l, r, sign, f, s, n, fast, tw, lane);
}

void launch_af2_bark_fwd(const AfArgs& a, hipStream_t stream) {
    hipLaunchKernelGGL(k_af2_bark_fwd, dim3(a.n_groups, 4 * a.bs, 2), dim3(kAf2Lanes), 0, stream, a);
}
void launch_af2_bark_bwd(const AfArgs& a, hipStream_t stream) {
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_af2_bark_bwd<0>), dim3(a.n_frames, 2 * a.bs), dim3(kAf2Lanes), 0, stream, a);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_af2_bark_bwd<1>), dim3(a.n_frames, 2 * a.bs), dim3(kAf2Lanes), 0, stream, a);
}

}  // namespace mst
