// mst_fx.hip - the fx bus of the console: noise-shaped reverberation on a send bus.
//
// Replaces dasp-pytorch's `stereo_bus` + `noise_shaped_reverberation` (reference call sites mst/modules.py:275-284;
// algorithm SURVEY A.2 / A.6, restated in oracle/dasp_restated.py):
//   fx_in[b,ch,n]  = sum_t 10^(send_db[b,t]/20) * mixed_tracks[b,ch,t,n]            (send bus; accumulated by k_apply_tracks)
//   wnf[r,k,n]     = sum_j f_k[j] * noise[r,k,n+j]        r = 2b + ch, 12 octave bands, `taps`-tap FIR band-passes
//   ir[b,ch,n]     = 1/12 sum_k gain[b,k] * exp(-(10 decay[b,k] + 1) * n/(S-1)) * wnf[r,k,n]      n < S (65536)
//   wet[b,ch,n]    = sum_j ir[b,ch,j] * fx_in[b,ch,n-j]   (causal)            master_bus += wet   (the console forces mix = 1)
// The noise is an INPUT (the reference draws torch.randn inside the op; the host wrapper draws it here, tests pass a fixed one).
//
// The 65536-tap convolution is a uniformly partitioned overlap-save convolution on the 8192-point register-radix engine
// (mst_fft2.h): partitions and hops of 4096 samples, left and right channel packed into one complex transform,
//   Xs[b][m]  = FFT8192(fx_in[b, L + iR, (m-1) 4096 .. (m+1) 4096))         one frame per 4096-sample block m
//   Hs[b][p]  = FFT8192(ir[b, L + iR, p 4096 .. (p+1) 4096) ++ 4096 zeros)   one per partition p < S / 4096
//   Ys[b][m]  = sum_p Xs[b][m-p] (.) Hs[b][p]     per channel (the packed spectra are separated by Hermitian symmetry)
//   wet block m = last 4096 samples of IFFT8192(Ys[b][m]):  real part left, imaginary part right.
// Backward = the adjoints of the same four steps (MAC with conj(H) / conj(X), frames placed / cropped on the other side).
#include "mst_kernels.h"
#include "mst_fft2.h"

namespace mst {

constexpr int kFxN = 8192, kFxHop = 4096, kFxLanes = FftPlan<8192>::LG;

// ---- band-pass filtering of the noise: wnf[(r,k)][n] = sum_j f[k][j] noise[(r,k)][n + j] ------------------------------
// Overlap-save correlation on the 8192-point engine (the direct form was the most expensive kernel of the bus: 25.7 GFLOP at
// bs 8, 277 us at 57 % of the fp32 VALU peak; this way it is two transforms per 8192 - (taps - 1) outputs of a stereo pair).
// The left and right noise rows of a band travel as the real and imaginary part of one complex sequence: the filter is real,
// so the filtered pair comes back the same way and nothing has to be separated.
//   Hc[k]   = conj(FFT8192(f[k] ++ zeros))                                   k_fx_filt_spec, 12 workgroups
//   block c = first V = 8192 - (taps - 1) samples of IFFT8192(FFT8192(z[c V .. c V + 8192)) (.) Hc[k])
__global__ __launch_bounds__(kFxLanes, 4) void k_fx_filt_spec(const float* __restrict__ filt, float2* __restrict__ Hc, const float* tables, int taps) {
    using S = FftShape<kFxN>;
    __shared__ __attribute__((aligned(16))) float2 buf[2][S::SLOTS];
    const int lane = threadIdx.x, k = blockIdx.x;
    const float2* twg = reinterpret_cast<const float2*>(tables);
    LaneTw<kFxN> tw;
    tw.init(twg, lane);
    fft8192_from<false>([&](int t) {
        const int e = lane + kFxLanes * t;
        return make_float2(e < taps ? filt[k * taps + e] : 0.0f, 0.0f);
    }, buf[0], buf[1], tw, twg[lane], lane);
    lds_barrier();
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const int q = lane + kFxLanes * t;
        const float2 z = buf[q & 1][S::slot(q >> 1)];
        Hc[(int64_t)k * kFxN + q] = make_float2(z.x, -z.y);
    }
}
// grid (ceil(S / V), 12, bs), 512 lanes
__global__ __launch_bounds__(kFxLanes, 2) void k_fx_fir(const float* __restrict__ noise, const float2* __restrict__ Hc, float* __restrict__ wnf,
                                                        const float* tables, int S, int taps) {
    using Sh = FftShape<kFxN>;
    __shared__ __attribute__((aligned(16))) float2 buf[2][Sh::SLOTS];
    const int lane = threadIdx.x, k = blockIdx.y, b = blockIdx.z;
    const int V = kFxN - (taps - 1), n0 = blockIdx.x * V, in_len = S + taps - 1;
    const float2* twg = reinterpret_cast<const float2*>(tables);
    LaneTw<kFxN> tw;
    tw.init(twg, lane);
    const float2 wl = twg[lane];
    const float* xl = noise + ((int64_t)(2 * b) * 12 + k) * in_len;
    const float* xr = noise + ((int64_t)(2 * b + 1) * 12 + k) * in_len;
    fft8192_from<false>([&](int t) {
        const int i = n0 + lane + kFxLanes * t;
        return i < in_len ? make_float2(xl[i], xr[i]) : make_float2(0.f, 0.f);
    }, buf[0], buf[1], tw, wl, lane);
    lds_barrier();
    // product with the filter spectrum, conjugated for the inverse (IFFT(P) = conj(FFT(conj P)) / N), formed while the second
    // transform gathers its inputs from the buffers it is about to overwrite (barrier inside)
    const float2* h = Hc + (int64_t)k * kFxN;
    fft8192_from<false, true>([&](int t) {
        const int q = lane + kFxLanes * t;
        const float2 pz = cmul(buf[q & 1][Sh::slot(q >> 1)], h[q]);
        return make_float2(pz.x, -pz.y);
    }, buf[0], buf[1], tw, wl, lane);
    lds_barrier();
    float* dl = wnf + ((int64_t)(2 * b) * 12 + k) * S;
    float* dr = wnf + ((int64_t)(2 * b + 1) * 12 + k) * S;
    constexpr float inv = 1.0f / (float)kFxN;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const int m = lane + kFxLanes * t;
        if (m < V && n0 + m < S) {
            const float2 y = buf[m & 1][Sh::slot(m >> 1)];
            dl[n0 + m] = inv * y.x;
            dr[n0 + m] = -inv * y.y;
        }
    }
}

// ---- impulse response: ir[r][n] = 1/12 sum_k gain[b,k] exp(-rate[b,k] n/(S-1)) wnf[r,k,n];  rcfx[b] = {gain[12], rate[12]}
__global__ __launch_bounds__(256) void k_fx_ir(const float* __restrict__ wnf, const float* __restrict__ rcfx, float* __restrict__ ir, int S) {
    const int r = blockIdx.y, b = r >> 1, n = blockIdx.x * 256 + threadIdx.x;
    if (n >= S) return;
    const float t = (float)n / (float)(S - 1);
    const float* c = rcfx + (int64_t)b * 24;
    float acc = 0.0f;
#pragma unroll
    for (int k = 0; k < 12; ++k) acc += c[k] * __builtin_amdgcn_exp2f(-1.4426950408889634f * c[12 + k] * t) * wnf[((int64_t)r * 12 + k) * S + n];
    ir[(int64_t)r * S + n] = acc * (1.0f / 12.0f);
}

// ---- 8192-point transforms of packed stereo frames -------------------------------------------------------------------
// MODE_SIG  : frame m of a signal (bs, 2, stride): samples (m-1) 4096 .. (m+1) 4096, zero outside [0, n)
// MODE_PART : partition p of an impulse response (bs, 2, S): 4096 samples then 4096 zeros
// MODE_GRAD : cotangent frame m: 4096 zeros then samples m 4096 .. (m+1) 4096  (adjoint of "keep the last 4096 outputs")
enum { FX_SIG = 0, FX_PART = 1, FX_GRAD = 2 };
struct FxFftArgs {
    const float* src;   // (bs, 2, stride)
    int64_t stride;     // samples between channel rows
    int64_t n;          // valid samples per row
    float2* spec;       // (bs, frames, 8192)
    const float* tables;  // twiddles (cos, -sin)(2 pi t / 8192), t < 8192
    int frames;
};
template <int MODE>
__global__ __launch_bounds__(kFxLanes, 4) void k_fx_fft(FxFftArgs a) {
    using S = FftShape<kFxN>;
    __shared__ __attribute__((aligned(16))) float2 buf[2][S::SLOTS];
    const int lane = threadIdx.x, m = blockIdx.x, b = blockIdx.y;
    const float2* twg = reinterpret_cast<const float2*>(a.tables);
    LaneTw<kFxN> tw;
    tw.init(twg, lane);
    const float2 wl = twg[lane];
    const float* xl = a.src + (int64_t)(2 * b) * a.stride;
    const float* xr = xl + a.stride;
    const int64_t start = MODE == FX_SIG ? (int64_t)(m - 1) * kFxHop : (MODE == FX_PART ? (int64_t)m * kFxHop : (int64_t)(m - 1) * kFxHop);
    fft8192_from<false>([&](int t) {
        const int e = lane + kFxLanes * t;            // element of the frame
        const int64_t i = start + e;
        bool live = i >= 0 && i < a.n;
        if (MODE == FX_PART) live = live && e < kFxHop;
        if (MODE == FX_GRAD) live = live && e >= kFxHop;
        return live ? make_float2(xl[i], xr[i]) : make_float2(0.f, 0.f);
    }, buf[0], buf[1], tw, wl, lane);
    lds_barrier();
    float2* out = a.spec + ((int64_t)b * a.frames + m) * kFxN;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const int q = lane + kFxLanes * t;  // bins 2q (even sequence) and 2q + 1 (odd sequence)
        const float2 e = buf[0][S::slot(q)], o = buf[1][S::slot(q)];
        *reinterpret_cast<float4*>(out + 2 * q) = make_float4(e.x, e.y, o.x, o.y);
    }
}

// ---- frequency-domain multiply-accumulate over the partitions -----------------------------------------------------------
// packed spectra Z = ZL + i ZR of two real signals: ZL[k] = (Z[k] + conj Z[N-k]) / 2, ZR[k] = (Z[k] - conj Z[N-k]) / (2i)
__device__ __forceinline__ void unpack_lr(float2 zk, float2 zn, float2& L, float2& R) {
    L = make_float2(0.5f * (zk.x + zn.x), 0.5f * (zk.y - zn.y));
    R = make_float2(0.5f * (zk.y + zn.y), -0.5f * (zk.x - zn.x));
}
// FX_MAC_Y : Y[m]  = sum_p X[m - p]   (.) H[p]          (forward)
// FX_MAC_DX: dX[m] = sum_p dY[m + p]  (.) conj(H[p])    (cotangent of the signal frames)
// FX_MAC_DH: dH[p] = sum_m dY[m]      (.) conj(X[m - p]) (cotangent of the partitions; `out` indexed by p)
enum { FX_MAC_Y = 0, FX_MAC_DX = 1, FX_MAC_DH = 2 };
struct FxMacArgs {
    const float2* a;  // (bs, na, 8192): X (Y), dY (DX, DH)
    const float2* h;  // (bs, nh, 8192): H (Y, DX), X (DH)
    float2* out;      // (bs, nout, 8192)
    int na, nh, nout;
};
template <int MODE>
__global__ __launch_bounds__(256) void k_fx_mac(FxMacArgs g) {
    const int k = blockIdx.x * 256 + threadIdx.x, mo = blockIdx.y, b = blockIdx.z;  // bin pair (k, N - k), output frame
    if (k > kFxN / 2) return;
    const int kn = (kFxN - k) & (kFxN - 1);
    const float2* A = g.a + (int64_t)b * g.na * kFxN;
    const float2* Hh = g.h + (int64_t)b * g.nh * kFxN;
    float2 yl = make_float2(0.f, 0.f), yr = make_float2(0.f, 0.f);
    const int np = MODE == FX_MAC_DH ? g.na : g.nh;
    for (int p = 0; p < np; ++p) {
        // (ia, ih): frame of `a` and frame of `h` that meet in this term
        const int ia = MODE == FX_MAC_Y ? mo - p : (MODE == FX_MAC_DX ? mo + p : p);
        const int ih = MODE == FX_MAC_DH ? p - mo : p;
        if (ia < 0 || ia >= g.na || ih < 0 || ih >= g.nh) continue;
        float2 al, ar, hl, hr;
        unpack_lr(A[(int64_t)ia * kFxN + k], A[(int64_t)ia * kFxN + kn], al, ar);
        unpack_lr(Hh[(int64_t)ih * kFxN + k], Hh[(int64_t)ih * kFxN + kn], hl, hr);
        if (MODE != FX_MAC_Y) {
            hl.y = -hl.y;
            hr.y = -hr.y;
        }
        yl = cadd(yl, cmul(al, hl));
        yr = cadd(yr, cmul(ar, hr));
    }
    float2* O = g.out + ((int64_t)b * g.nout + mo) * kFxN;
    // repack: Z[k] = L[k] + i R[k];  Z[N - k] = conj(L[k]) + i conj(R[k])
    O[k] = make_float2(yl.x - yr.y, yl.y + yr.x);
    if (kn != k) O[kn] = make_float2(yl.x + yr.y, -yl.y + yr.x);
}

// The same three products for K = 16 partitions (the reference's 65536-sample response) with every spectrum bin read ONCE:
// a lane owns the bin pair (k, N - k) of one batch item and walks the frames keeping the last 16 unpacked (L, R) values of
// the sliding operand in a register ring (static indices: the walk is unrolled by 16) next to the 16 resident ones.  The
// generic kernel above re-reads both operands for every term (1.07 GB of L2 traffic per launch at cfg #2: 80-110 us);
// this one moves 67 MB.
//   FX_MAC_Y : resident H_p, ring of X;  frames ascending in chunks of 16 (grid.y), 15 frames of run-in per chunk
//   FX_MAC_DX: resident H_p, ring of dY; frames DESCENDING
//   FX_MAC_DH: resident accumulators dH_p, ring of X, one walk over all frames
template <int MODE>
__global__ __launch_bounds__(256) void k_fx_mac16(FxMacArgs g) {
    constexpr int K = 16;
    const int k = blockIdx.x * 256 + threadIdx.x, chunk = blockIdx.y, b = blockIdx.z;
    if (k > kFxN / 2) return;
    const int kn = (kFxN - k) & (kFxN - 1);
    const float2* A = g.a + (int64_t)b * g.na * kFxN;
    const float2* Hh = g.h + (int64_t)b * g.nh * kFxN;
    float2* O = g.out + (int64_t)b * g.nout * kFxN;
    float2 hl[K], hr[K], rl[K], rr[K];  // resident operand (H, or the dH accumulators) and the ring
#pragma unroll
    for (int p = 0; p < K; ++p) {
        rl[p] = rr[p] = make_float2(0.f, 0.f);
        if (MODE == FX_MAC_DH) hl[p] = hr[p] = make_float2(0.f, 0.f);
        else unpack_lr(Hh[(int64_t)p * kFxN + k], Hh[(int64_t)p * kFxN + kn], hl[p], hr[p]);
    }
    auto store = [&](int m, float2 yl, float2 yr) {
        O[(int64_t)m * kFxN + k] = make_float2(yl.x - yr.y, yl.y + yr.x);
        if (kn != k) O[(int64_t)m * kFxN + kn] = make_float2(yl.x + yr.y, -yl.y + yr.x);
    };
    if (MODE == FX_MAC_Y) {
        const int m0 = chunk * K, m1 = m0 + K < g.nout ? m0 + K : g.nout;
        for (int base = m0 - K; base < m1; base += K) {  // one block of run-in, then the chunk (both multiples of 16)
#pragma unroll
            for (int j = 0; j < K; ++j) {
                const int m = base + j;
                float2 xl = make_float2(0.f, 0.f), xr = xl;
                if (m >= 0 && m < g.na) unpack_lr(A[(int64_t)m * kFxN + k], A[(int64_t)m * kFxN + kn], xl, xr);
                rl[j] = xl;
                rr[j] = xr;
                if (m >= m0 && m < m1) {
                    float2 yl = make_float2(0.f, 0.f), yr = yl;
#pragma unroll
                    for (int p = 0; p < K; ++p) {  // X[m - p] sits in slot (j - p) mod 16
                        yl = cadd(yl, cmul(rl[(j - p) & (K - 1)], hl[p]));
                        yr = cadd(yr, cmul(rr[(j - p) & (K - 1)], hr[p]));
                    }
                    store(m, yl, yr);
                }
            }
        }
    } else if (MODE == FX_MAC_DX) {
        const int m0 = chunk * K, m1 = m0 + K < g.nout ? m0 + K : g.nout;
        for (int base = m0 + K; base >= m0; base -= K) {  // run-in from above: dY[m + p], p < 16
#pragma unroll
            for (int j = K - 1; j >= 0; --j) {
                const int m = base + j;
                float2 xl = make_float2(0.f, 0.f), xr = xl;
                if (m < g.na) unpack_lr(A[(int64_t)m * kFxN + k], A[(int64_t)m * kFxN + kn], xl, xr);
                rl[j] = xl;
                rr[j] = xr;
                if (m >= m0 && m < m1) {
                    float2 yl = make_float2(0.f, 0.f), yr = yl;
#pragma unroll
                    for (int p = 0; p < K; ++p) {  // dY[m + p] sits in slot (j + p) mod 16
                        yl = cadd(yl, cmul(rl[(j + p) & (K - 1)], cconj(hl[p])));
                        yr = cadd(yr, cmul(rr[(j + p) & (K - 1)], cconj(hr[p])));
                    }
                    store(m, yl, yr);
                }
            }
        }
    } else {
        // the frames are dealt to gridDim.y workgroups in runs of whole 16-blocks (one block of run-in fills the ring); each
        // writes its own partial dH, and the inverse transform behind it adds the partials (k_fx_ifft<FX_CROP>, nsum)
        const int nb16 = (g.na + K - 1) / K, per = (nb16 + (int)gridDim.y - 1) / (int)gridDim.y;
        const int m0 = chunk * per * K, m1 = m0 + per * K < g.na ? m0 + per * K : g.na;
        for (int base = m0 - K; base < m1; base += K) {
#pragma unroll
            for (int j = 0; j < K; ++j) {
                const int m = base + j;
                if (m >= m1) break;
                if (m < 0) continue;  // the ring slots still hold zeros
                unpack_lr(Hh[(int64_t)m * kFxN + k], Hh[(int64_t)m * kFxN + kn], rl[j], rr[j]);  // X[m] enters the ring
                if (m < m0) continue;
                float2 dl, dr;
                unpack_lr(A[(int64_t)m * kFxN + k], A[(int64_t)m * kFxN + kn], dl, dr);   // dY[m]
#pragma unroll
                for (int p = 0; p < K; ++p) {  // dH[p] += dY[m] conj(X[m - p]); slots of frames < 0 still hold zeros
                    hl[p] = cadd(hl[p], cmul(dl, cconj(rl[(j - p) & (K - 1)])));
                    hr[p] = cadd(hr[p], cmul(dr, cconj(rr[(j - p) & (K - 1)])));
                }
            }
        }
#pragma unroll
        for (int p = 0; p < K; ++p) store(chunk * K + p, hl[p], hr[p]);
    }
}

// ---- inverse transforms of spectrum frames ---------------------------------------------------------------------------
// FX_OUT : dst[b, :, m 4096 .. (m+1) 4096) += last 4096 samples of IFFT(spec[b][m])          (wet signal onto the bus)
// FX_SCAT: dst[b, :, m 4096 .. (m+1) 4096)  = last 4096 samples of IFFT(spec[b][m]) + first 4096 samples of IFFT(spec[b][m+1])
//          (cotangent of the send bus: frame m came from samples (m-1) 4096 .. (m+1) 4096; owner-computes, one transform per block)
// FX_CROP: dst[b, :, m 4096 .. (m+1) 4096)  = first 4096 samples of IFFT(spec[b][m])         (cotangent of the impulse response)
enum { FX_OUT = 0, FX_SCAT = 1, FX_CROP = 2 };
struct FxIfftArgs {
    const float2* spec;  // (bs, frames, 8192)
    float* dst;          // (bs, 2, stride)
    int64_t stride, n;
    const float* tables;
    int frames;
    int nsum;            // FX_CROP: the spectrum is the sum of nsum partials, `frames` frames apart (the frame axis holds nsum * frames)
    // wet/dry mix (forward_mix_console may hand over mix != 1; AdvancedMixConsole.forward always runs with 1):
    const float* mixv;   // (bs) or null
    const float* dry;    // FX_OUT: the send bus fx_in, FX_SCAT: the bus cotangent dbus   (bs, 2, dry_stride)
    int64_t dry_stride;
    const float* dry2;   // FX_SCAT: fx_in (bs, 2, dry2_stride) for the mix gradient <dbus, fx_in>
    int64_t dry2_stride;
    float* dry_part;     // FX_SCAT: (bs, frames) partial sums of <dbus, fx_in>
};
template <int MODE>
__global__ __launch_bounds__(kFxLanes, 4) void k_fx_ifft(FxIfftArgs a) {
    using S = FftShape<kFxN>;
    __shared__ __attribute__((aligned(16))) float2 buf[2][S::SLOTS];
    const int lane = threadIdx.x, m = blockIdx.x, b = blockIdx.y;
    const float2* twg = reinterpret_cast<const float2*>(a.tables);
    LaneTw<kFxN> tw;
    tw.init(twg, lane);
    const float2 wl = twg[lane];
    constexpr float inv = 1.0f / (float)kFxN;
    // half-frame values of this lane: samples 2q, 2q + 1 for q = lane + 512 t, t = 0..3 (first half) or 4..7 (second half),
    // i.e. offsets 2 lane + 1024 (t mod 4) + {0, 1} inside a 4096-sample block; {L even, R even, L odd, R odd} per t
    float acc[4][4];
    const int ns = MODE == FX_CROP ? a.nsum : 1;
    const float2* in = a.spec + ((int64_t)b * a.frames * ns + m) * kFxN;
    const bool next = MODE == FX_SCAT && m + 1 < a.frames;
    // IDFT(Z) = conj(FFT(conj(Z))) / N.  FX_SCAT: the last half of IFFT(Z_m) is the first half of IFFT((-1)^k Z_m) (a circular
    // shift by N/2), so block m = first half of ONE inverse transform of (-1)^k Z_m + Z_(m+1)
    fft8192_from<false>([&](int t) {
        const int q = lane + kFxLanes * t;
        float2 z = in[q];
        for (int c = 1; c < ns; ++c) z = cadd(z, in[(int64_t)c * a.frames * kFxN + q]);
        if (MODE == FX_SCAT) {
            if (lane & 1) z = make_float2(-z.x, -z.y);  // q = lane + 512 t has the parity of lane
            if (next) z = cadd(z, in[kFxN + q]);
        }
        return make_float2(z.x, -z.y);
    }, buf[0], buf[1], tw, wl, lane);
    lds_barrier();
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int q = lane + kFxLanes * (MODE == FX_OUT ? t + 4 : t);
        const float2 e = buf[0][S::slot(q)], o = buf[1][S::slot(q)];
        acc[t][0] = inv * e.x;
        acc[t][1] = -inv * e.y;
        acc[t][2] = inv * o.x;
        acc[t][3] = -inv * o.y;
    }
    float* dl = a.dst + (int64_t)(2 * b) * a.stride + (int64_t)m * kFxHop;
    float* dr = dl + a.stride;
    if (MODE != FX_CROP && a.mixv) {
        // y = (1 - mix) fx_in + mix wet: the wet share rides on the band gains (k_prep); the dry share and its adjoints are added here
        const float dryw = 1.0f - a.mixv[b];
        float dot = 0.0f;
        if (dryw != 0.0f) {  // uniform per workgroup; mix = 1 (every AdvancedMixConsole.forward call) touches nothing
            const float* sl = a.dry + (int64_t)(2 * b) * a.dry_stride + (int64_t)m * kFxHop;
            const float* sr = sl + a.dry_stride;
            const float* xl = MODE == FX_SCAT ? a.dry2 + (int64_t)(2 * b) * a.dry2_stride + (int64_t)m * kFxHop : nullptr;
            const float* xr = MODE == FX_SCAT ? xl + a.dry2_stride : nullptr;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int off = 2 * lane + 1024 * t;
                const int64_t i = (int64_t)m * kFxHop + off;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (i + j < a.n) {
                        const float vl = sl[off + j], vr = sr[off + j];
                        acc[t][2 * j] = fmaf(dryw, vl, acc[t][2 * j]);
                        acc[t][2 * j + 1] = fmaf(dryw, vr, acc[t][2 * j + 1]);
                        if (MODE == FX_SCAT) dot = fmaf(vl, xl[off + j], fmaf(vr, xr[off + j], dot));
                    }
                }
            }
        }
        if (MODE == FX_SCAT) {  // fixed-order block sum (lanes -> waves -> workgroup): the mix gradient stays reproducible
            float* red = reinterpret_cast<float*>(&buf[1][0]);
            const float w = wave_sum(dot);
            lds_barrier();
            if ((lane & 63) == 0) red[lane >> 6] = w;
            lds_barrier();
            if (lane == 0) {
                float s = 0.0f;
                for (int i = 0; i < kFxLanes / 64; ++i) s += red[i];
                a.dry_part[(int64_t)b * a.frames + m] = s;
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int off = 2 * lane + 1024 * t;
        const int64_t i = (int64_t)m * kFxHop + off;
        if (i < a.n) {
            dl[off] = MODE == FX_OUT ? dl[off] + acc[t][0] : acc[t][0];
            dr[off] = MODE == FX_OUT ? dr[off] + acc[t][1] : acc[t][1];
        }
        if (i + 1 < a.n) {
            dl[off + 1] = MODE == FX_OUT ? dl[off + 1] + acc[t][2] : acc[t][2];
            dr[off + 1] = MODE == FX_OUT ? dr[off + 1] + acc[t][3] : acc[t][3];
        }
    }
}

// ---- backward of the impulse-response synthesis: partial sums over n of dL/dgain[b,k], dL/drate[b,k] -----------------------
// part[b][blk][24]; grid (ceil(S / 1024), bs), 256 lanes x 4 samples x both channels
__global__ __launch_bounds__(256) void k_fx_ir_bwd(const float* __restrict__ wnf, const float* __restrict__ rcfx, const float* __restrict__ dir,
                                                   float* __restrict__ part, int S) {
    __shared__ float red[4][24];
    const int tid = threadIdx.x, b = blockIdx.y;
    const float* c = rcfx + (int64_t)b * 24;
    float dg[12], dr[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) dg[k] = dr[k] = 0.0f;
    for (int q = 0; q < 4; ++q) {
        const int n = blockIdx.x * 1024 + q * 256 + tid;
        if (n >= S) continue;
        const float t = (float)n / (float)(S - 1);
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
            const int r = 2 * b + ch;
            const float d = dir[(int64_t)r * S + n] * (1.0f / 12.0f);
#pragma unroll
            for (int k = 0; k < 12; ++k) {
                const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * c[12 + k] * t) * wnf[((int64_t)r * 12 + k) * S + n];
                dg[k] = fmaf(d, e, dg[k]);
                dr[k] = fmaf(d * c[k] * (-t), e, dr[k]);
            }
        }
    }
    const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        const float a = wave_sum(dg[k]), r = wave_sum(dr[k]);
        if (lane == 0) {
            red[wave][k] = a;
            red[wave][12 + k] = r;
        }
    }
    lds_barrier();
    if (tid < 24) part[((int64_t)b * gridDim.x + blockIdx.x) * 24 + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
}

// ---- launch sequences (called from mst_console.hip) ------------------------------------------------------------------------
void launch_fx_forward(const FxPlan& p, const float* noise, const float* filters, const float* tables, float* ws, float* bus,
                       int64_t bus_stride, hipStream_t stream) {
    const int rows = 2 * p.bs;
    float2* Hc = reinterpret_cast<float2*>(ws + p.Hf);
    const int V = kFxN - (p.taps - 1);
    hipLaunchKernelGGL(k_fx_filt_spec, dim3(12), dim3(kFxLanes), 0, stream, filters, Hc, tables, p.taps);
    hipLaunchKernelGGL(k_fx_fir, dim3((p.S + V - 1) / V, 12, p.bs), dim3(kFxLanes), 0, stream, noise, Hc, ws + p.wnf, tables, p.S, p.taps);
    hipLaunchKernelGGL(k_fx_ir, dim3((p.S + 255) / 256, rows), dim3(256), 0, stream, ws + p.wnf, ws + p.rcfx, ws + p.ir, p.S);
    FxFftArgs fs{ws + p.fx_in, p.Ns, p.n, reinterpret_cast<float2*>(ws + p.Xs), tables, p.nblk};
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fx_fft<FX_SIG>), dim3(p.nblk, p.bs), dim3(kFxLanes), 0, stream, fs);
    FxFftArgs fh{ws + p.ir, p.S, p.S, reinterpret_cast<float2*>(ws + p.Hs), tables, p.K};
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fx_fft<FX_PART>), dim3(p.K, p.bs), dim3(kFxLanes), 0, stream, fh);
    FxMacArgs mc{reinterpret_cast<const float2*>(ws + p.Xs), reinterpret_cast<const float2*>(ws + p.Hs), reinterpret_cast<float2*>(ws + p.Ys),
                 p.nblk, p.K, p.nblk};
    if (p.K == 16) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fx_mac16<FX_MAC_Y>), dim3((kFxN / 2 + 256) / 256, (p.nblk + 15) / 16, p.bs), dim3(256), 0, stream, mc);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fx_mac<FX_MAC_Y>), dim3((kFxN / 2 + 256) / 256, p.nblk, p.bs), dim3(256), 0, stream, mc);
    FxIfftArgs io{reinterpret_cast<const float2*>(ws + p.Ys), bus, bus_stride, p.n, tables, p.nblk, 1,
                  ws + p.mixv, ws + p.fx_in, p.Ns, nullptr, 0, nullptr};
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fx_ifft<FX_OUT>), dim3(p.nblk, p.bs), dim3(kFxLanes), 0, stream, io);
}

// dbus: cotangent of the bus the wet signal was added to (bs, 2, dbus_stride).  Leaves dfx_in (cotangent of the send bus) and
// the partial sums of the reverberation parameters in the workspace.
void launch_fx_backward(const FxPlan& p, const float* dbus, int64_t dbus_stride, const float* tables, float* ws, hipStream_t stream) {
    FxFftArgs fg{dbus, dbus_stride, p.n, reinterpret_cast<float2*>(ws + p.Ys), tables, p.nblk};  // dY overwrites Y
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fx_fft<FX_GRAD>), dim3(p.nblk, p.bs), dim3(kFxLanes), 0, stream, fg);
    // cotangent of the send bus: dX[m] = sum_p dY[m + p] conj(H[p]), frames scattered back over (m-1) 4096 .. (m+1) 4096
    FxMacArgs mx{reinterpret_cast<const float2*>(ws + p.Ys), reinterpret_cast<const float2*>(ws + p.Hs), reinterpret_cast<float2*>(ws + p.dXs),
                 p.nblk, p.K, p.nblk};
    if (p.K == 16) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fx_mac16<FX_MAC_DX>), dim3((kFxN / 2 + 256) / 256, (p.nblk + 15) / 16, p.bs), dim3(256), 0, stream, mx);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fx_mac<FX_MAC_DX>), dim3((kFxN / 2 + 256) / 256, p.nblk, p.bs), dim3(256), 0, stream, mx);
    FxIfftArgs ix{reinterpret_cast<const float2*>(ws + p.dXs), ws + p.dfx_in, p.Ns, p.n, tables, p.nblk, 1,
                  ws + p.mixv, dbus, dbus_stride, ws + p.fx_in, p.Ns, ws + p.dry};
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fx_ifft<FX_SCAT>), dim3(p.nblk, p.bs), dim3(kFxLanes), 0, stream, ix);
    // cotangent of the impulse response: dH[p] = sum_m dY[m] conj(X[m - p]), first 4096 samples of each inverse
    // K = 16: the frame walk of the dH product is dealt to up to kFxDhChunks workgroups per bin slice (partials summed by the
    // inverse transform); the generic kernel writes one spectrum per partition
    const int nb16 = (p.nblk + 15) / 16, chunks = p.K == 16 ? (nb16 < kFxDhChunks ? nb16 : kFxDhChunks) : 1;
    FxMacArgs mh{reinterpret_cast<const float2*>(ws + p.Ys), reinterpret_cast<const float2*>(ws + p.Xs), reinterpret_cast<float2*>(ws + p.dHs),
                 p.nblk, p.nblk, p.K * chunks};
    if (p.K == 16) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fx_mac16<FX_MAC_DH>), dim3((kFxN / 2 + 256) / 256, chunks, p.bs), dim3(256), 0, stream, mh);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fx_mac<FX_MAC_DH>), dim3((kFxN / 2 + 256) / 256, p.K, p.bs), dim3(256), 0, stream, mh);
    FxIfftArgs ih{reinterpret_cast<const float2*>(ws + p.dHs), ws + p.dir, p.S, p.S, tables, p.K, chunks, nullptr, nullptr, 0, nullptr, 0, nullptr};
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fx_ifft<FX_CROP>), dim3(p.K, p.bs), dim3(kFxLanes), 0, stream, ih);
    hipLaunchKernelGGL(k_fx_ir_bwd, dim3(p.nblk_ir, p.bs), dim3(256), 0, stream, ws + p.wnf, ws + p.rcfx, ws + p.dir, ws + p.fxpart, p.S);
}

// twiddle table of the 8192-point engine: (cos, -sin)(2 pi t / 8192), fp64 evaluation rounded once
__global__ void k_fx_tables(float* tables) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= kFxN) return;
    const double ang = 6.283185307179586476925 * (double)t / (double)kFxN;
    tables[2 * t] = (float)cos(ang);
    tables[2 * t + 1] = (float)(-sin(ang));
}
void launch_fx_tables(float* tables, hipStream_t stream) {
    hipLaunchKernelGGL(k_fx_tables, dim3(kFxN / 256), dim3(256), 0, stream, tables);
}

}  // namespace mst
