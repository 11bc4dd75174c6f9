// mst_stft2.hip - round-2 kernels of the multi-resolution STFT loss on the register-radix FFT engine (mst_fft2.h).
// Same semantics as mst_stft.hip (auraloss 0.4.0; reference configs/models/naive.yaml:54-68); see mst_stft.h
// for the split between the two files.
#include "mst_stft.h"
#include "mst_fft2.h"
#ifndef MST_STFT2_ABLATE
#define MST_STFT2_ABLATE 0  // timing diagnostics only (wrong results): 1 = no epilogue math, 2 = no global sample loads, 4 = first pass only
#endif
#ifndef MST_STFT2_HALF_REUSE
#define MST_STFT2_HALF_REUSE 1  // prefetching kernels (n_fft <= 2048): a frame's first half is taken from the previous frame's registers, not re-fetched
#endif
#ifndef MST_STFT2_W8192
#define MST_STFT2_W8192 4  // min waves per SIMD asked of the 8192 FORWARD kernel: 2 workgroups of 512 lanes per CU (128 VGPRs, no spill)
#endif
#ifndef MST_STFT2_W8192_BWD
// the recomputing backward needs ~170 registers: at 128 it spills 71 (measured 102 us vs 77 us with one workgroup per CU); the
// saved-spectrum backward spills 47 at 128 as well (72.7 us against 30.2 at one workgroup per CU)
#define MST_STFT2_W8192_BWD 2
#endif

namespace mst {

__device__ __forceinline__ int reflect_i32(int i, int n) {
    i = i < 0 ? -i : i;
    return i >= n ? 2 * (n - 1) - i : i;
}

// Forward transform of ONE frame (pred + i target, windowed) by the LG lanes of a workgroup; afterwards the spectrum
// Z sits in `buf` in natural order (padded slots): NSEQ = 1: Z[k] = buf[0][slot(k)]; NSEQ = 2 (n_fft = 8192):
// Z[2m] = buf[0][slot(m)], Z[2m+1] = buf[1][slot(m)].  Ends with a barrier (the spectrum is readable by every lane).
template <int N>
struct FrameLoader {
    using S = FftShape<N>;
    static constexpr int LG = S::LG, PTS = N / LG, H = N / 2;
    // raw(t) -> unwindowed (pred, target) sample pair of element lane + LG t; win[t] = its window value (NSEQ = 1);
    // the 8192-point transform forms its window from the radix-2 twiddle it needs anyway:
    //   W_N^i = wl W_16^t (i = lane + 512 t),  hann(i) = 0.5 - 0.5 Re W_N^i,  hann(i + N/2) = 0.5 + 0.5 Re W_N^i
    // HALF: the window values handed in (NSEQ = 1) / formed here (8192) carry a factor 1/2: split<true> then skips its own
    template <bool HALF = false, typename F>
    __device__ static __forceinline__ void transform(F&& raw, const float* win, float2 (*buf)[S::SLOTS],
                                                     const LaneTw<N>& tw, float2 wl, int lane) {
        if constexpr (S::NSEQ == 1) {
            float2 v[8], o[S::NBL][S::RL];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const float2 z = raw(t);
                v[t] = make_float2(win[t] * z.x, win[t] * z.y);
            }
            fft_run<N>(v, o, buf[0], tw, lane);
            group_lds_sync<LG>();
#pragma unroll
            for (int u = 0; u < S::NBL; ++u)
#pragma unroll
                for (int t = 0; t < S::RL; ++t) buf[0][S::slot(lane + u * LG + t * (S::M / S::RL))] = o[u][t];
        } else {
            fft8192_from<true, false, HALF>(raw, buf[0], buf[1], tw, wl, lane);
        }
        group_lds_sync<LG>();
    }
    // spectrum bin k of the frame transformed last
    __device__ static __forceinline__ float2 bin(const float2 (*buf)[S::SLOTS], int k) {
        if constexpr (S::NSEQ == 1) return buf[0][S::slot(k)];
        else return buf[k & 1][S::slot(k >> 1)];
    }
    // HALF: the transform ran on a window scaled by 1/2 (transform<true>): Z is half the packed spectrum - exactly, a power-of-two
    // scaling commutes with every rounding - and the four multiplications by 1/2 below are not needed: the same X, Y to the bit
    template <bool HALF = false>
    __device__ static __forceinline__ void split(const float2 (*buf)[S::SLOTS], int k, float2& X, float2& Y) {
        const float2 zk = bin(buf, k), zn = bin(buf, (N - k) & (N - 1));
        if constexpr (HALF) {
            X = make_float2(zk.x + zn.x, zk.y - zn.y);
            Y = make_float2(zk.y + zn.y, zn.x - zk.x);
        } else {
            X = make_float2(0.5f * (zk.x + zn.x), 0.5f * (zk.y - zn.y));
            Y = make_float2(0.5f * (zk.y + zn.y), -0.5f * (zk.x - zn.x));
        }
    }
};

// |X| and 1 / |X| of the loss epilogues: the hardware's 1-ulp v_sqrt_f32 / v_rcp_f32 instead of the correctly rounded library
// sequences (10 and 11 instructions each; the arguments are clamped to >= eps, far from the denormal range those sequences guard)
#ifndef MST_STFT2_PRECISE_MAG
#define MST_STFT2_PRECISE_MAG 0
#endif
__device__ __forceinline__ float mag_sqrt(float v) { return MST_STFT2_PRECISE_MAG ? sqrtf(v) : __builtin_amdgcn_sqrtf(v); }
__device__ __forceinline__ float mag_rcp(float v) { return MST_STFT2_PRECISE_MAG ? 1.0f / v : __builtin_amdgcn_rcpf(v); }

// periodic Hann window value of element lane + LG t.  n_fft <= 2048: from the table (same fp32 values as torch.hann_window);
// 8192: 0.5 - 0.5 cos(2 pi i / N) with the cosine taken from the product W_N^lane W_32^t that the radix-2 split needs anyway
template <int N>
__device__ __forceinline__ void load_window(float* win, const float* __restrict__ wtab, int lane) {
    constexpr int LG = FftShape<N>::LG;
#pragma unroll
    for (int t = 0; t < N / LG; ++t) win[t] = wtab[lane + LG * t];
}

// balanced strips: strip g of G covers frames [g F / G, (g + 1) F / G)
__device__ __forceinline__ void strip_range(int g, int G, int F, int& f0, int& f1) {
    f0 = (int)(((unsigned)g * (unsigned)F) / (unsigned)G);  // g F < 2^31: at most 2^20 frames per row in 2^11 strips
    f1 = (int)(((unsigned)(g + 1) * (unsigned)F) / (unsigned)G);
}

// Forward of one strip of one row by a group of LG lanes (lane = 0 .. LG-1).  `buf` / `red`: the group's LDS (one transform buffer
// set, LG / 64 x 4 floats).  strip / nstrips: balanced strips over the row's frames.  live = false: a filler unit of the fused
// launch (k_stft3_fwd) - it runs the same barriers as its neighbour group and writes nothing.
template <int N>
__device__ __forceinline__ void stft2_fwd_body(const StftArgs& a, const int lane, const int strip, const int nstrips, const int row,
                                               const bool live, float2 (*buf)[FftShape<N>::SLOTS], float (*red)[4]) {
    using S = FftShape<N>;
    using L = FrameLoader<N>;
    constexpr int LG = S::LG, PTS = N / LG, H = N / 2;
    const ResInfo r = a.r;
    const float2* twg = reinterpret_cast<const float2*>(a.tables + r.tw_off);
    LaneTw<N> tw;
    tw.init(twg, lane);
    const float2 wl = twg[lane];
    float win[S::NSEQ == 1 ? PTS : 1];
    if constexpr (S::NSEQ == 1) load_window<N>(win, a.tables + r.win_off, lane);
#ifndef MST_STFT2_FWD_HALF_WINDOW
#define MST_STFT2_FWD_HALF_WINDOW 1  // the forward's window carries the 1/2 of the Hermitian split (FrameLoader::split<true>): four multiplications per bin less, same bits
#endif
    constexpr bool HW = MST_STFT2_FWD_HALF_WINDOW;
    if constexpr (S::NSEQ == 1 && HW) {
#pragma unroll
        for (int t = 0; t < PTS; ++t) win[t] *= 0.5f;
    }
    const float* x = a.pred + (int64_t)row * a.n;
    const float* y = a.target + (int64_t)row * a.n;
    int f0, f1;
    strip_range(strip, nstrips, r.n_frames, f0, f1);
    float s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f;
    const int nrow = (int)a.n;  // 32-bit sample indices: a row is < 2^31 samples
    // Every frame is fetched whole (consecutive frames share half their samples: the re-read hits L2); the reflection costs
    // three integer ops per sample and only ever acts in a row's first and last frame.  n_fft <= 2048: the NEXT frame's
    // samples are requested before the current frame is transformed, so their latency hides behind ~700 instructions.
    constexpr bool PREFETCH = S::NSEQ == 1;
    float2 nxt[PREFETCH ? PTS : 1];
    auto fetch = [&](int f, int t) {
        const int i = reflect_i32(f * H - H + lane + LG * t, nrow);
        if (MST_STFT2_ABLATE & 2) return make_float2((float)i, 1.0f);
        return make_float2(x[i], y[i]);
    };
    if constexpr (PREFETCH) {
#pragma unroll
        for (int t = 0; t < PTS; ++t) nxt[t] = fetch(f0, t);
    }
    for (int f = f0; f < f1; ++f) {
        if constexpr (PREFETCH) {
            float2 cur[PTS];
#pragma unroll
            for (int t = 0; t < PTS; ++t) cur[t] = nxt[t];
            if (f + 1 < f1) {
                // hop = n_fft / 2: the next frame's first half IS this frame's second half, element for element in the same lane
                // (the reflection at the row ends is a function of the sample index, so it agrees too): four new pairs per frame
#pragma unroll
                for (int t = 0; t < PTS; ++t) nxt[t] = (MST_STFT2_HALF_REUSE && t < PTS / 2) ? cur[t + PTS / 2] : fetch(f + 1, t);
            }
            L::template transform<HW>([&](int t) { return cur[t]; }, win, buf, tw, wl, lane);
        } else {
            // 8192: W_N^lane is re-fetched per frame (an L1 hit) instead of living in two registers across the epilogue - the
            // kernel sits exactly at the 128-register edge that lets two 512-lane workgroups share a CU, and a spilled register
            // means scratch memory for every wave of the launch
            int li = lane;
            asm volatile("" : "+v"(li));
            const float2 wlf = twg[li];
            L::template transform<HW>([&](int t) { return fetch(f, t); }, win, buf, tw, wlf, lane);
        }
#pragma unroll 2
        for (int k = lane; k <= ((MST_STFT2_ABLATE & 1) ? lane : N / 2); k += LG) {
            float2 X, Y;
            L::template split<HW>(buf, k, X, Y);
            const float xm = mag_sqrt(fmaxf(X.x * X.x + X.y * X.y, a.eps));
            const float ym = mag_sqrt(fmaxf(Y.x * Y.x + Y.y * Y.y, a.eps));
            // round 5: the target's magnitudes are what the backward needs of the target - kept, so that it transforms the prediction alone
            // (non-temporal stores of the two kept planes were measured: forward 86.5 -> 85.8 us, but the backward then reads them from HBM
            // instead of the Infinity Cache - 43.2 -> 49.0 and 27.5 -> 30.3 us; plain stores)
            if (a.ymag) a.ymag[((int64_t)row * r.n_frames + f) * (N / 2 + 1) + k] = ym;
            if constexpr (stft2_keeps_spectrum(N)) {  // ... and of the prediction its spectrum: the backward runs the inverse only
                if (a.xspec) reinterpret_cast<float2*>(a.xspec)[((int64_t)row * r.n_frames + f) * (N / 2 + 1) + k] = X;
            }
            const float d = ym - xm;
            s1 = fmaf(d, d, s1);
            s2 = fmaf(ym, ym, s2);
            s3 += fabsf(__builtin_amdgcn_logf(xm) - __builtin_amdgcn_logf(ym));  // log2; scaled by ln2 below
            s4 += fabsf(d);
        }
        group_lds_sync<LG>();  // the next frame's first pass overwrites the spectrum
    }
    s3 *= kLn2;
    const int wave = lane >> 6, wl_ = lane & 63;
    s1 = wave_sum(s1); s2 = wave_sum(s2); s3 = wave_sum(s3); s4 = wave_sum(s4);
    if (wl_ == 0) { red[wave][0] = s1; red[wave][1] = s2; red[wave][2] = s3; red[wave][3] = s4; }
    group_lds_sync<LG>();
    if (lane < 4 && live) {
        float v = 0.f;
        for (int w = 0; w < LG / 64; ++w) v += red[w][lane];
        a.part[((int64_t)row * nstrips + strip) * 4 + lane] = v;
    }
}

template <int N>
__global__ __launch_bounds__(FftPlan<N>::LG, (N == 8192 ? MST_STFT2_W8192 : 1)) void k_stft2_fwd(StftArgs a) {
    using S = FftShape<N>;
    __shared__ __attribute__((aligned(16))) float2 buf[S::NSEQ][S::SLOTS];
    __shared__ float red[S::LG / 64][4];
    stft2_fwd_body<N>(a, threadIdx.x, blockIdx.x, gridDim.x, blockIdx.y, true, buf, red);
    if (a.tickets && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) a.tickets[0] = 0u;  // armed for k_mrstft_finish (mst_stft.hip)
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 5: the three forward transforms of the reference's resolutions in ONE launch.  Separate launches cost a drain + ramp
// each (~2 x 5 us with their table prologues) and each leaves the chip mostly idle in its last round - the 8192-point launch is
// 2.03 rounds of its 512 resident workgroups (65 frames per row over 32 strips: sixteen workgroups walk a third frame while
// 496 slots sit empty), the others end ragged as well.  Here every workgroup is 512 lanes and takes one ROLE from its index:
//   role 0  one strip of the 8192-point transform (512 lanes per frame, as before)
//   role 1  two strips of the 2048-point transform, one per 256-lane half: the same strip index of two different rows, i.e. the
//           same frame range, so both halves run the same number of workgroup barriers (an odd row count gives the last half a
//           filler that repeats a row and writes nothing)
//   role 2  eight strips of the 512-point transform, one per wave (no workgroup barrier in that role)
// LDS and registers are the 8192-point kernel's (64 KB, 128: two workgroups per CU = four waves per SIMD, which is what each of
// the three kernels ran at alone).  Heavy roles first, so that the short ones fill the tail.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512, MST_STFT2_W8192) void k_stft3_fwd(Stft3Args p) {
    __shared__ __attribute__((aligned(16))) float2 lds[2 * FftShape<8192>::SLOTS];
    __shared__ float red[8][4];
    const int tid = threadIdx.x, b = blockIdx.x;
    if (p.tickets && b == 0 && tid == 0) p.tickets[0] = 0u;  // armed for k_mrstft_finish (mst_stft.hip), which runs after this launch
#ifndef MST_STFT3_ROLES
#define MST_STFT3_ROLES 7  // diagnostics: bit mask of the roles compiled in
#endif
    if (!(MST_STFT3_ROLES & 1) && b < p.wg_end[0]) return;
    if (!(MST_STFT3_ROLES & 2) && b >= p.wg_end[0] && b < p.wg_end[1]) return;
    if (!(MST_STFT3_ROLES & 4) && b >= p.wg_end[1]) return;
#ifndef MST_STFT3_XCD
#define MST_STFT3_XCD 1  // neighbouring strips of a row on ONE XCD (they share half a frame of samples): see xcd_chunk below
#endif
    // Workgroup ids go round-robin over the eight XCDs, each with an L2 of its own: strips u, u + 1 of a row - which read the same half
    // frame - landed on two L2s and the overlap was fetched twice from the fabric (FETCH_SIZE of this launch: 4.3x its unique input).
    // Within a role, XCD j takes the j-th contiguous eighth of the (row, strip) sequence instead.
    auto xcd_chunk = [](int u, int n) {
        if (!MST_STFT3_XCD) return u;
        const int q = n >> 3, r = n & 7, x = u & 7, idx = u >> 3;
        return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + idx;
    };
    if (b < p.wg_end[0]) {
        const int G = p.groups[0], u = xcd_chunk(b, p.wg_end[0]);
        stft2_fwd_body<8192>(p.a[0], tid, u % G, G, u / G, true, reinterpret_cast<float2(*)[FftShape<8192>::SLOTS]>(lds), red);
    } else if (b < p.wg_end[1]) {
        const int u = xcd_chunk(b - p.wg_end[0], p.wg_end[1] - p.wg_end[0]), G = p.groups[1], sub = __builtin_amdgcn_readfirstlane(tid >> 8);  // wave-uniform: keep it scalar
        const int row = 2 * (u / G) + sub;
        const bool live = row < p.rows;
        stft2_fwd_body<2048>(p.a[1], tid & 255, u % G, G, live ? row : p.rows - 1, live,
                             reinterpret_cast<float2(*)[FftShape<2048>::SLOTS]>(lds + sub * FftShape<2048>::SLOTS), red + 4 * sub);
    } else {
        const int G = p.groups[2], wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int u = xcd_chunk(b - p.wg_end[1], p.wg_end[2] - p.wg_end[1]) * 8 + wave;
        if (u < G * p.rows)
            stft2_fwd_body<512>(p.a[2], tid & 63, u % G, G, u / G, true,
                                reinterpret_cast<float2(*)[FftShape<512>::SLOTS]>(lds + wave * FftShape<512>::SLOTS), red + wave);
    }
}


// =====================================================================================================================
// Backward.  Per frame: forward transform (recomputed), cotangent of the prediction's half spectrum, inverse transform of
// its Hermitian extension, window, overlap-add.  The overlap-add is OWNER-COMPUTES - no float atomics on interior samples
// and a fixed summation order everywhere, so the gradient is bitwise reproducible:
//   hop = n_fft / 2: hop block b = samples [b h, (b+1) h) receives the second half of frame b and the first half of frame
//   b + 1, and the lane that holds output element i of one frame holds element i of the next.  A workgroup walks
//   consecutive frames keeping the previous frame's second half in registers (`carry`); a block is complete when the next
//   frame's first half arrives and is written once.
//   HALO mode (512 / 2048): a strip owns whole blocks [b0, b1) and transforms frames b0 .. b1, i.e. it recomputes the one
//   frame it shares with the next strip (1 / strip-length extra work); every sample of grad_pred is written by exactly one
//   lane of the launch (plain store, or read-modify-write when another resolution has already written: a.accumulate).
//   SEAM mode (8192, where frames are few and a recomputed frame costs 1/3 more): every frame is transformed once; a
//   strip's leading first half and trailing second half are seams, added with atomics onto a ZEROED buffer - exactly two
//   contributions per seam sample, and x + y is commutative, so the result does not depend on their order.  Runs first.
//   Row ends: frame 0's first half and the last frame's second half reflect back into the row (torch.stft centre
//   padding); they are mirrored through LDS into the owning lanes' registers before the block is written.
// IDFT(h) = conj(FFT(conj(h))).  512 / 2048: two frames share one complex inverse (He_a + i He_b, real part = frame a,
// imaginary part = frame b).  8192: one frame per inverse of HALF size: y_even + i y_odd = IDFT_M(A + i Bq),
// A[k] = H[k] + conj(H[M-k]), Bq[k] = (H[k] - conj(H[M-k])) conj(W_N^k), M = 4096 - the same 8 x 8 x 8 x 8 plan as the
// even / odd halves of the forward transform.
// =====================================================================================================================
// cotangent of the prediction's bin X given the target's saved magnitude ym (= sqrt(clamp(|Y|^2, eps)), written by the forward)
__device__ __forceinline__ float2 cotangent_xy(float2 X, float ym, float eps, const float* coef) {
    const float p2 = X.x * X.x + X.y * X.y;
    const float xm = mag_sqrt(fmaxf(p2, eps));
    float g = coef[0] * (xm - ym);
    const float dl = __builtin_amdgcn_logf(xm) - __builtin_amdgcn_logf(ym);
    const float rx = mag_rcp(xm);
    g += coef[1] * ((dl > 0.f) - (dl < 0.f)) * rx;
    g += coef[2] * ((xm > ym) - (xm < ym));
    const float s = (p2 >= eps) ? g * rx : 0.0f;  // through sqrt(clamp(|X|^2, eps)): zero below the clamp
    return make_float2(s * X.x, s * X.y);
}
template <int N>
__device__ __forceinline__ float2 cotangent(const float2 (*buf)[FftShape<N>::SLOTS], int k, float eps, const float* coef) {
    float2 X, Y;
    FrameLoader<N>::split(buf, k, X, Y);
    const float p2 = X.x * X.x + X.y * X.y;
    const float xm = mag_sqrt(fmaxf(p2, eps));
    const float ym = mag_sqrt(fmaxf(Y.x * Y.x + Y.y * Y.y, eps));
    float g = coef[0] * (xm - ym);
    const float dl = __builtin_amdgcn_logf(xm) - __builtin_amdgcn_logf(ym);
    const float rx = mag_rcp(xm);
    g += coef[1] * ((dl > 0.f) - (dl < 0.f)) * rx;
    g += coef[2] * ((xm > ym) - (xm < ym));
    const float s = (p2 >= eps) ? g * rx : 0.0f;  // through sqrt(clamp(|X|^2, eps)): zero below the clamp
    return make_float2(s * X.x, s * X.y);
}

#ifndef MST_STFT2_BWD_L512
#define MST_STFT2_BWD_L512 4  // 5 frames per workgroup, 25 % recomputed: 4096 one-wave workgroups = exactly the 4 waves per SIMD that 128 registers allow
#endif
#ifndef MST_STFT2_BWD_L2048
#define MST_STFT2_BWD_L2048 4  // 5 frames per 256-lane workgroup, 25 % recomputed: 1024 workgroups = one round at 127 registers (55.7 -> 46.8 us)
#endif

#ifndef MST_STFT2_W2048_BWD
#define MST_STFT2_W2048_BWD 1  // min waves per SIMD asked of the 2048-point backward (A/B switch; uncapped it takes 144 registers)
#endif
#ifndef MST_STFT2_BWD8192_SLOTS
#define MST_STFT2_BWD8192_SLOTS 256
#endif
constexpr int kStft2Bwd8192Slots = MST_STFT2_BWD8192_SLOTS;  // resident workgroups of k_stft2_bwd<8192> on the 256 CUs
// one strip of one row by a group of LG lanes (lane = 0 .. LG - 1); buf / hb: the group's LDS (hb: n_fft <= 2048 only)
template <int N>
__device__ __forceinline__ void stft2_bwd_body(const StftArgs& a, const int lane, const int strip, const int nstrips, const int row,
                                               float2 (*buf)[FftShape<N>::SLOTS], float2* hb, const int blk0 = -1, const int blk1 = -1) {
    using S = FftShape<N>;
    using L = FrameLoader<N>;
    constexpr int LG = S::LG, H = N / 2;
    constexpr bool PAIR = S::NSEQ == 1, SEAMS = !PAIR;
    constexpr int K = PAIR ? 4 : 8;  // floats of one half frame per lane
    const ResInfo r = a.r;
    const float2* twg = reinterpret_cast<const float2*>(a.tables + r.tw_off);
    LaneTw<N> tw;
    tw.init(twg, lane);
    const float2 wl = twg[lane];
    float win[PAIR ? 8 : 1];
    if constexpr (PAIR) load_window<N>(win, a.tables + r.win_off, lane);
    const float* x = a.pred + (int64_t)row * a.n;
    const float* y = a.target + (int64_t)row * a.n;
    float* gx = a.grad_pred + (int64_t)row * a.n;
    const float gl = a.grad_loss[0];
    const float coef[3] = {a.coef[(int64_t)row * 4] * gl, a.coef[(int64_t)row * 4 + 1] * gl, a.coef[(int64_t)row * 4 + 2] * gl};
    const int nrow = (int)a.n, B = nrow / H;  // hop blocks per row; frames 0 .. B
    int F0, F1;  // frames [F0, F1) of this strip
    if constexpr (SEAMS) {
        strip_range(strip, nstrips, B + 1, F0, F1);
    } else {
        int b0, b1;
        strip_range(strip, nstrips, B, b0, b1);
        if (blk0 >= 0) { b0 = blk0; b1 = blk1; }  // the caller's own split of the row (k_stft2_bwd_512_2048)
        F0 = b0;
        F1 = b1 + 1;
    }
    // offset, inside a hop block, of half-frame value q of this lane
    auto pos = [&](int q) { return PAIR ? lane + LG * q : 2 * lane + 1024 * (q >> 1) + (q & 1); };
    // halo-mode launch that follows a seam-mode one: does the seam launch's block holding sample `at` have its second
    // contribution parked in a.seam?  (frame t opens a strip of the seam launch iff t = floor(j F / G) for some 1 <= j < G)
    auto parked = [&](int at) {
        if (SEAMS || !a.seam) return false;
        const int t = at / a.seam_hop + 1, F = a.seam_frames, G = a.seam_groups;
        const int j = (int)(((unsigned)t * (unsigned)G + (unsigned)F - 1u) / (unsigned)F);
        return j >= 1 && j < G && (int)(((unsigned)j * (unsigned)F) / (unsigned)G) == t;
    };
    auto put = [&](int block, const float* v, bool rmw) {  // write one complete block
        float* g = gx + block * H;
        // 8192-point kernel (round 6): every load of the block before its first store.  Written as  *p = *p + v  per element the compiler
        // cannot prove the addresses distinct and emits load - wait - store four times, each wait (vmcnt(0): loads and stores share the
        // counter on gfx9) also draining the previous store - four dependent round trips per block instead of one (27.3 -> 25.9 us).
        // Same sums, same order.
        if constexpr (PAIR) {
            // (uncapped, the K temporaries of the batched form take the fused 512 / 2048-point kernel from 128 to 144 registers = three waves
            // per SIMD instead of four, 43.7 -> 51.4 us: it is built with a 128-register cap - MST_STFT2_W512_2048_BWD - and fits without spills)
#ifndef MST_STFT2_PUT_BATCH
#define MST_STFT2_PUT_BATCH 1  // A/B: 0 = the element-wise form (43.5 against 43.0 us for the fused 512 / 2048-point launch)
#endif
            if (parked(block * H)) {  // workgroup-uniform
                const float* sc = a.seam + (int64_t)row * a.n + block * H;
                if (MST_STFT2_PUT_BATCH) {
                    float o[K], p[K];
#pragma unroll
                    for (int q = 0; q < K; ++q) {
                        o[q] = rmw ? g[pos(q)] : 0.0f;
                        p[q] = sc[pos(q)];
                    }
#pragma unroll
                    for (int q = 0; q < K; ++q) g[pos(q)] = o[q] + p[q] + v[q];
                    return;
                }
#pragma unroll
                for (int q = 0; q < K; ++q) g[pos(q)] = (rmw ? g[pos(q)] : 0.0f) + sc[pos(q)] + v[q];
                return;
            }
            if (MST_STFT2_PUT_BATCH && rmw) {
                float o[K];
#pragma unroll
                for (int q = 0; q < K; ++q) o[q] = g[pos(q)];
#pragma unroll
                for (int q = 0; q < K; ++q) g[pos(q)] = o[q] + v[q];
                return;
            }
#pragma unroll
            for (int q = 0; q < K; ++q) g[pos(q)] = rmw ? g[pos(q)] + v[q] : v[q];
        } else {
            float2 o[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) o[t] = rmw ? *reinterpret_cast<const float2*>(g + 2 * lane + 1024 * t) : make_float2(0.f, 0.f);
#pragma unroll
            for (int t = 0; t < 4; ++t)
                *reinterpret_cast<float2*>(g + 2 * lane + 1024 * t) = make_float2(o[t].x + v[2 * t], o[t].y + v[2 * t + 1]);
        }
    };
    // SEAM mode: one of the two contributions to a block shared with a neighbour strip.  `opening` = the first half of this
    // strip's first frame (the other one is the previous strip's trailing half).
    auto seam = [&](int block, const float* v, bool opening) {
        if (a.seam) {  // hand-over: each half is stored once, the next launch adds the parked one
            if (opening) {
                float* sc = a.seam + (int64_t)row * a.n + block * H;
#pragma unroll
                for (int t = 0; t < K / 2; ++t)
                    *reinterpret_cast<float2*>(sc + 2 * lane + 1024 * t) = make_float2(v[2 * t], v[2 * t + 1]);
            } else {
                put(block, v, a.accumulate != 0);
            }
            return;
        }
        float* g = gx + block * H;
#pragma unroll
        for (int q = 0; q < K; ++q) unsafeAtomicAdd(&g[pos(q)], v[q]);
    };
    float* fold = reinterpret_cast<float*>(&buf[0][0]);  // H floats of scratch for the row-end mirrors (the transform buffer is idle then)
    float carry[K];
    bool have_carry = false;
    // one finished frame: `first` -> block f - 1, `second` -> block f
    auto emit = [&](int f, float* first, float* second) {
        if (f == 0) {  // the first half lies before the row: sample -p reflects to +p, i.e. element i = H - p lands on block 0, offset p
            group_lds_sync<LG>();
#pragma unroll
            for (int q = 0; q < K; ++q) fold[pos(q)] = first[q];
            group_lds_sync<LG>();
#pragma unroll
            for (int q = 0; q < K; ++q)
                if (pos(q) >= 1) second[q] += fold[H - pos(q)];
            group_lds_sync<LG>();
        } else if (f == B) {  // the second half lies beyond the row: element H + i' reflects to sample n - 2 - i'
            group_lds_sync<LG>();
#pragma unroll
            for (int q = 0; q < K; ++q) fold[pos(q)] = second[q];
            group_lds_sync<LG>();
#pragma unroll
            for (int q = 0; q < K; ++q)
                if (pos(q) <= H - 2) first[q] += fold[H - 2 - pos(q)];
        }
        if (f > 0) {
            if (have_carry) {
#pragma unroll
                for (int q = 0; q < K; ++q) first[q] += carry[q];
                put(f - 1, first, a.accumulate != 0);
            } else if (SEAMS) {
                seam(f - 1, first, true);
            }
        }
        if (f == B) {
            // i' = H - 1 reflects to sample n - H - 1 = the last sample of block B - 2, which this lane stored one frame ago
            if (pos(K - 1) == H - 1) gx[nrow - H - 1] += fold[H - 1];
            group_lds_sync<LG>();
            have_carry = false;
        } else {
#pragma unroll
            for (int q = 0; q < K; ++q) carry[q] = second[q];
            have_carry = true;
        }
    };
    auto fetch = [&](int f, int t) {
        const int i = reflect_i32(f * H - H + lane + LG * t, nrow);
        return make_float2(x[i], y[i]);
    };

#ifndef MST_STFT2_BWD_PAIR_8192
#define MST_STFT2_BWD_PAIR_8192 0  // ... and for the 8192-point resolution as pairs of frames in the full-size transform.  Built and NOT taken: 46.1 us against
                                   // 49.6 (the two inverses of a pair run one after the other, each with its own set of barriers: 1.5 instead of 2 barrier
                                   // sets per frame), and the interior log-magnitude adjoint moves from 0.8e-3 to 2.1e-3 from float64 (fp32 reference
                                   // 0.8e-3; test_mrstft_log_magnitude_adjoint_away_from_the_clamp[2-65536] fails its 2x bound)
#endif
#ifndef MST_STFT2_BWD_SAVED_MAG_8192
#define MST_STFT2_BWD_SAVED_MAG_8192 0  // ... for the 8192-point resolution as well (the frame as a REAL transform through one 4096-point complex one).
                                        // Built, -17 us more (8192-point backward 49 -> ~27 us: two sequences per frame instead of three, eight 8-byte loads,
                                        // two workgroups per CU) - and NOT taken: X[k] = E[k] + W^k O[k] cancels two terms of the size of X[k + 4096],
                                        // and on the near-zero bins of frame 0 the log-magnitude adjoint lands 5-15x further from float64 than the
                                        // fp32 reference does on half of the seeds (tools/dbg_logmag2.py); the paired transform below does not
                                        // (equal to rounds 2-4 on targets independent of the prediction)
#endif
#ifndef MST_STFT2_BWD_SAVED_MAG
#define MST_STFT2_BWD_SAVED_MAG 1  // 0: rounds 2-4 - every frame's (prediction + i target) transform is recomputed in the backward
#endif
    // Round 5: the forward keeps |Y| (a.ymag: one float per bin and frame), which is all the cotangent needs of the target.  The
    // backward then transforms the PREDICTION alone, and two real frames share one complex transform exactly as they share the
    // inverse: z = w (x_a + i x_b), X_a / X_b by the Hermitian split.  Per pair of frames: one forward + one inverse transform instead
    // of two + one; consecutive frames overlap by half, so a pair needs two new half frames of ONE signal (8 loads per lane, was 32).
    if constexpr (PAIR && stft2_keeps_spectrum(N)) {
        // Round 5, second step: no forward transform in the backward at all - the forward launch kept the prediction's spectrum next to the
        // target's magnitudes (12 bytes per bin and frame).  Per pair of frames: cotangents of both -> conj(He_a + i He_b) -> ONE inverse.
        constexpr int NB = N / 2 + 1, NK = (NB + LG - 1) / LG;  // bins per frame / trips per lane
        const float2* xrow = reinterpret_cast<const float2*>(a.xspec) + (int64_t)row * r.n_frames * NB;
        const float* ymrow = a.ymag + (int64_t)row * r.n_frames * NB;
        for (int fa = F0; fa < F1; fa += 2) {
            const bool have_b = fa + 1 < F1;
            float2 xa[NK], xb[NK];
            float ya[NK], yb[NK];
#pragma unroll
            for (int i = 0; i < NK; ++i) {
                const int k = lane + LG * i;
                const bool in = k < NB;
                xa[i] = in ? xrow[(int64_t)fa * NB + k] : make_float2(0.f, 0.f);
                ya[i] = in ? ymrow[(int64_t)fa * NB + k] : 1.0f;
                xb[i] = (in && have_b) ? xrow[(int64_t)(fa + 1) * NB + k] : make_float2(0.f, 0.f);
                yb[i] = (in && have_b) ? ymrow[(int64_t)(fa + 1) * NB + k] : 1.0f;
            }
#pragma unroll
            for (int i = 0; i < NK; ++i) {
                const int k = lane + LG * i;
                if (k < NB) {
                    const float2 Ga = cotangent_xy(xa[i], ya[i], a.eps, coef);
                    const float2 Gb = have_b ? cotangent_xy(xb[i], yb[i], a.eps, coef) : make_float2(0.f, 0.f);
                    const int sk = S::slot(k), sn = S::slot((N - k) & (N - 1));
                    // conj(He_a + i He_b):  He[k] = G / 2, He[N - k] = conj(G) / 2; the real bins 0 and N / 2 carry G.x whole
                    if (k == 0 || k == N / 2) {
                        hb[sk] = make_float2(Ga.x, -Gb.x);
                    } else {
                        hb[sk] = make_float2(0.5f * (Ga.x - Gb.y), -0.5f * (Ga.y + Gb.x));
                        hb[sn] = make_float2(0.5f * (Ga.x + Gb.y), 0.5f * (Ga.y - Gb.x));
                    }
                }
            }
            group_lds_sync<LG>();  // hb is complete
            float2 v[8], o[S::NBL][S::RL];
#pragma unroll
            for (int t = 0; t < 8; ++t) v[t] = hb[S::slot(lane + LG * t)];
            fft_run<N>(v, o, buf[0], tw, lane);
            float fa1[4], fa2[4], fb1[4], fb2[4];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float2 R = S::RL == 8 ? o[0][q] : o[q & 1][q >> 1];  // element lane + LG q
                const float ra = win[q] * R.x, rb = -win[q] * R.y;
                if (q < 4) { fa1[q] = ra; fb1[q] = rb; }
                else { fa2[q - 4] = ra; fb2[q - 4] = rb; }
            }
            group_lds_sync<LG>();  // the inverse has left buf[0] (emit() may use it as mirror scratch); every lane has read hb
            emit(fa, fa1, fa2);
            if (have_b) emit(fa + 1, fb1, fb2);
        }
    } else if constexpr (PAIR && MST_STFT2_BWD_SAVED_MAG) {
        constexpr int NB = N / 2 + 1, NK = (NB + LG - 1) / LG;  // bins per frame / cotangent-loop trips per lane
        const float* ymrow = a.ymag + (int64_t)row * r.n_frames * NB;
        auto fetchx = [&](int f, int t) { return x[reflect_i32(f * H - H + lane + LG * t, nrow)]; };
        // Frame 0 of a row goes ALONE (its partner is an all-zero virtual frame -1): it is even about its centre (reflect padding under a
        // symmetric window), so its spectrum is (-1)^k R[k] with R real - it crosses zero between bins, and those bins' 1 / |X| carries
        // the rounding error of the whole log-magnitude adjoint.  Sharing a transform with frame 1 would put frame 1's round-off onto
        // exactly those bins; what fp32 leaves in the imaginary part of frame 0's spectrum is error only and is dropped (the exact value).
        const int fstart = F0 == 0 ? -1 : F0;
        float xa1[4];  // first half of frame fa (elements lane + LG t, t < 4)
#pragma unroll
        for (int t = 0; t < 4; ++t) xa1[t] = fstart >= 0 ? fetchx(fstart, t) : 0.0f;
        for (int fa = fstart; fa < F1; fa += 2) {
            const bool have_a = fa >= 0, have_b = fa + 1 < F1;
            float xa2[4], xb2[4], ya[NK], yb[NK];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                xa2[t] = fetchx(fa + 1, t);                       // second half of frame fa = first half of frame fa + 1 (fa + 1 <= B always)
                xb2[t] = have_b ? fetchx(fa + 1, t + 4) : 0.0f;   // = the first half of frame fa + 2
            }
#pragma unroll
            for (int i = 0; i < NK; ++i) {  // requested here, used behind the transform
                const int k = lane + LG * i;
                ya[i] = (k < NB && have_a) ? ymrow[(int64_t)fa * NB + k] : 1.0f;
                yb[i] = (k < NB && have_b) ? ymrow[(int64_t)(fa + 1) * NB + k] : 1.0f;
            }
            {
                float2 v[8], o[S::NBL][S::RL];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    v[t] = make_float2(have_a ? win[t] * xa1[t] : 0.0f, have_b ? win[t] * xa2[t] : 0.0f);
                    v[t + 4] = make_float2(have_a ? win[t + 4] * xa2[t] : 0.0f, win[t + 4] * xb2[t]);
                }
                fft_run<N>(v, o, buf[0], tw, lane);
                group_lds_sync<LG>();
#pragma unroll
                for (int u = 0; u < S::NBL; ++u)
#pragma unroll
                    for (int t = 0; t < S::RL; ++t) buf[0][S::slot(lane + u * LG + t * (S::M / S::RL))] = o[u][t];
                group_lds_sync<LG>();
            }
#pragma unroll
            for (int i = 0; i < NK; ++i) {
                const int k = lane + LG * i;
                if (k < NB) {
                    float2 Xa, Xb;
                    L::split(buf, k, Xa, Xb);
                    if (fa == -1) Xb.y = 0.0f;  // frame 0: real spectrum (above)
                    const float2 Ga = have_a ? cotangent_xy(Xa, ya[i], a.eps, coef) : make_float2(0.f, 0.f);
                    const float2 Gb = have_b ? cotangent_xy(Xb, yb[i], a.eps, coef) : make_float2(0.f, 0.f);
                    const int sk = S::slot(k), sn = S::slot((N - k) & (N - 1));
                    // conj(He_a + i He_b):  He[k] = G / 2, He[N - k] = conj(G) / 2; the real bins 0 and N / 2 carry G.x whole
                    if (k == 0 || k == N / 2) {
                        hb[sk] = make_float2(Ga.x, -Gb.x);
                    } else {
                        hb[sk] = make_float2(0.5f * (Ga.x - Gb.y), -0.5f * (Ga.y + Gb.x));
                        hb[sn] = make_float2(0.5f * (Ga.x + Gb.y), 0.5f * (Ga.y - Gb.x));
                    }
                }
            }
            group_lds_sync<LG>();  // hb is complete, the spectrum has been consumed
            float2 v[8], o[S::NBL][S::RL];
#pragma unroll
            for (int t = 0; t < 8; ++t) v[t] = hb[S::slot(lane + LG * t)];
            fft_run<N>(v, o, buf[0], tw, lane);
            float fa1[4], fa2[4], fb1[4], fb2[4];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float2 R = S::RL == 8 ? o[0][q] : o[q & 1][q >> 1];  // element lane + LG q
                const float ra = win[q] * R.x, rb = -win[q] * R.y;
                if (q < 4) { fa1[q] = ra; fb1[q] = rb; }
                else { fa2[q - 4] = ra; fb2[q - 4] = rb; }
            }
            group_lds_sync<LG>();  // the inverse has left buf[0]: emit() may use it as mirror scratch, the next transform as work space
            if (have_a) emit(fa, fa1, fa2);
            if (have_b) emit(fa + 1, fb1, fb2);
#pragma unroll
            for (int t = 0; t < 4; ++t) xa1[t] = xb2[t];
        }
    } else if constexpr (PAIR) {
#ifndef MST_STFT2_BWD2048_PREFETCH
#define MST_STFT2_BWD2048_PREFETCH 0  // the 2048-point backward fetches each frame when it needs it: 16 registers less = 127, i.e. four
                                      // workgroups per CU instead of three, which is what lets 1024 five-frame strips run in one round
#endif
        constexpr bool PREF = N != 2048 || MST_STFT2_BWD2048_PREFETCH;
        float2 nxt[PREF ? 8 : 1];
        if (PREF) {
#pragma unroll
            for (int t = 0; t < 8; ++t) nxt[t] = fetch(F0, t);
        }
        for (int fa = F0; fa < F1; fa += 2) {
            const bool have_b = fa + 1 < F1;
#pragma unroll 1
            for (int which = 0; which < 2; ++which) {
                if (which == 1 && !have_b) break;
                const int f = fa + which;
                float2 cur[8];
                if (PREF) {
#pragma unroll
                    for (int t = 0; t < 8; ++t) cur[t] = nxt[t];
                    if (f + 1 < F1) {
#pragma unroll
                        for (int t = 0; t < 8; ++t) nxt[t] = (MST_STFT2_HALF_REUSE && t < 4) ? cur[t + 4] : fetch(f + 1, t);  // see stft2_fwd_body
                    }
                } else {
#pragma unroll
                    for (int t = 0; t < 8; ++t) cur[t] = fetch(f, t);
                }
                L::transform([&](int t) { return cur[t]; }, win, buf, tw, wl, lane);
                for (int k = lane; k <= N / 2; k += LG) {
                    const float2 G = cotangent<N>(buf, k, a.eps, coef);
                    const bool edge = (k == 0) || (k == N / 2);
                    const int sk = S::slot(k), sn = S::slot((N - k) & (N - 1));
                    if (which == 0) {
                        // conj(He): He[k] = G/2 -> (Gx/2, -Gy/2); He[N-k] = conj(G)/2 -> (Gx/2, +Gy/2)
                        hb[sk] = edge ? make_float2(G.x, 0.f) : make_float2(0.5f * G.x, -0.5f * G.y);
                        if (!edge) hb[sn] = make_float2(0.5f * G.x, 0.5f * G.y);
                    } else if (edge) {
                        hb[sk].y -= G.x;  // conj(i He_b[k]) = (0, -Gx) at the real bins
                    } else {
                        // conj(i He_b): i G/2 -> (-Gy/2, -Gx/2);  i conj(G)/2 -> (Gy/2, -Gx/2)
                        const float2 h0 = hb[sk], h1 = hb[sn];
                        hb[sk] = make_float2(h0.x - 0.5f * G.y, h0.y - 0.5f * G.x);
                        hb[sn] = make_float2(h1.x + 0.5f * G.y, h1.y - 0.5f * G.x);
                    }
                }
                group_lds_sync<LG>();  // the spectrum has been consumed (the next transform overwrites it); hb is complete
            }
            // FFT(conj(h)) = conj(r_a + i r_b)  =>  r_a = Re, r_b = -Im
            float2 v[8], o[S::NBL][S::RL];
#pragma unroll
            for (int t = 0; t < 8; ++t) v[t] = hb[S::slot(lane + LG * t)];
            fft_run<N>(v, o, buf[0], tw, lane);
            float fa1[4], fa2[4], fb1[4], fb2[4];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float2 R = S::RL == 8 ? o[0][q] : o[q & 1][q >> 1];  // element lane + LG q
                const float ra = win[q] * R.x, rb = -win[q] * R.y;
                if (q < 4) { fa1[q] = ra; fb1[q] = rb; }
                else { fa2[q - 4] = ra; fb2[q - 4] = rb; }
            }
            group_lds_sync<LG>();  // the inverse has left buf[0]: emit() may use it as mirror scratch, the next transform as work space
            emit(fa, fa1, fa2);
            if (have_b) emit(fa + 1, fb1, fb2);
        }
    } else if constexpr (MST_STFT2_BWD_SAVED_MAG && MST_STFT2_BWD_SAVED_MAG_8192) {
        // 8192, round 5: the prediction's frame alone, as a REAL transform - z[m] = x[2m] + i x[2m+1] through one 4096-point complex
        // transform, X[k] = E[k] + W_N^k O[k], X[M - k] = conj(E[k] - W_N^k O[k]) with E / O the Hermitian split of Z at (k, M - k) - the
        // exact mirror of the half-size inverse below.  Two 4096-point transforms per frame instead of three, eight 8-byte loads per lane
        // instead of 32 four-byte ones, one sequence in flight instead of two (registers: two workgroups per CU).
        constexpr int M = S::M, NB = N / 2 + 1;                    // 4096, 4097
        const float2 we = twg[2 * lane], wo = twg[2 * lane + 1];
        const float* ymrow = a.ymag + (int64_t)row * r.n_frames * NB;
        // periodic Hann window of samples 2m, 2m + 1 (m = lane + 512 t): 0.5 - 0.5 Re(W_N^(2 lane + c) W_8^t)
        auto hann2 = [&](int t, float& he, float& ho) {
            const float2 w8 = t == 0 ? make_float2(1.f, 0.f) : (t == 1 ? make_float2(0.70710678118654752f, -0.70710678118654752f)
                            : (t == 2 ? make_float2(0.f, -1.f) : (t == 3 ? make_float2(-0.70710678118654752f, -0.70710678118654752f)
                            : (t == 4 ? make_float2(-1.f, 0.f) : (t == 5 ? make_float2(-0.70710678118654752f, 0.70710678118654752f)
                            : (t == 6 ? make_float2(0.f, 1.f) : make_float2(0.70710678118654752f, 0.70710678118654752f)))))));
            he = 0.5f - 0.5f * (we.x * w8.x - we.y * w8.y);
            ho = 0.5f - 0.5f * (wo.x * w8.x - wo.y * w8.y);
#ifdef MST_STFT2_REAL8192_TABLE_WINDOW  // experiment: the exactly rounded table values (torch.hann_window's) instead of the twiddle products
            const float2 wt = *reinterpret_cast<const float2*>(a.tables + r.win_off + 2 * (lane + LG * t));
            he = wt.x;
            ho = wt.y;
#endif
        };
        for (int f = F0; f < F1; ++f) {
            const int base = f * H - H;
            const bool interior = base >= 0 && base + N <= nrow;  // no reflection in this frame (workgroup-uniform)
            const float* ymf = ymrow + (int64_t)f * NB;
            float2 v[8], o[1][8];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int m2 = base + 2 * (lane + LG * t);
                float2 p;
                if (interior) p = *reinterpret_cast<const float2*>(x + m2);
                else p = make_float2(x[reflect_i32(m2, nrow)], x[reflect_i32(m2 + 1, nrow)]);
                float he, ho;
                hann2(t, he, ho);
                v[t] = make_float2(he * p.x, ho * p.y);
            }
            float yk[4], ymk[4];  // the target's magnitudes at this lane's bin pairs (k, M - k), requested ahead of the transform
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                yk[i] = ymf[lane + LG * i];
                ymk[i] = ymf[M - (lane + LG * i)];
            }
            const float yh = ymf[M / 2];
            fft_run<N>(v, o, buf[0], tw, lane);  // Z[lane + 512 t]
            group_lds_sync<LG>();
#pragma unroll
            for (int t = 0; t < 8; ++t) buf[0][S::slot(lane + LG * t)] = o[0][t];
            group_lds_sync<LG>();
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                if (i == 4 && lane != 0) break;
                const int k = i == 4 ? M / 2 : lane + LG * i;
                const float2 Zk = buf[0][S::slot(k)], Zm = buf[0][S::slot((M - k) & (M - 1))];
                const float2 w = twg[k];  // W_N^k ; W_N^(M-k) = -conj(W_N^k)
                float2 Xk, Xm;
                if (k == 0) {  // bins 0 and N / 2: both real
                    Xk = make_float2(Zk.x + Zk.y, 0.f);
                    Xm = make_float2(Zk.x - Zk.y, 0.f);
                } else {
                    const float2 E = make_float2(0.5f * (Zk.x + Zm.x), 0.5f * (Zk.y - Zm.y));
                    const float2 O = make_float2(0.5f * (Zk.y + Zm.y), -0.5f * (Zk.x - Zm.x));
                    const float2 WO = cmul(w, O);
                    Xk = cadd(E, WO);
                    Xm = cconj(csub(E, WO));
                }
                // Frame 0 of a row is x[-p] = x[p] under a window that is symmetric about the same point: X[k] = (-1)^k R[k] with R REAL, which
                // crosses zero between bins - the bins whose 1 / |X| carries the whole rounding error of the log-magnitude adjoint.  The
                // imaginary part any fp32 transform leaves there is error only; it is dropped (the exact value), as the real-even structure says.
                if (f == 0) { Xk.y = 0.0f; Xm.y = 0.0f; }
                float2 Hk = cotangent_xy(Xk, i == 4 ? yh : yk[i < 4 ? i : 0], a.eps, coef);
                float2 Hm = cotangent_xy(Xm, i == 4 ? yh : ymk[i < 4 ? i : 0], a.eps, coef);
                float2 vk, vm;
                if (k == 0) {
                    vk = make_float2(Hk.x + Hm.x, -(Hk.x - Hm.x));  // V[0] = (H0 + HM) + i (H0 - HM), both real; conj
                    vm = vk;
                } else {
                    Hk = make_float2(0.5f * Hk.x, 0.5f * Hk.y);
                    Hm = make_float2(0.5f * Hm.x, 0.5f * Hm.y);
                    const float2 Ak = make_float2(Hk.x + Hm.x, Hk.y - Hm.y);
                    const float2 Bk = cmul(make_float2(Hk.x - Hm.x, Hk.y + Hm.y), make_float2(w.x, -w.y));
                    vk = make_float2(Ak.x - Bk.y, -(Ak.y + Bk.x));
                    const float2 Am = make_float2(Hm.x + Hk.x, Hm.y - Hk.y);
                    const float2 Bm = cmul(make_float2(Hm.x - Hk.x, Hm.y + Hk.y), make_float2(-w.x, -w.y));
                    vm = make_float2(Am.x - Bm.y, -(Am.y + Bm.x));
                }
                buf[1][S::slot(k)] = vk;
                if (k != 0 && k != M / 2) buf[1][S::slot(M - k)] = vm;
            }
            group_lds_sync<LG>();
#pragma unroll
            for (int t = 0; t < 8; ++t) v[t] = buf[1][S::slot(lane + LG * t)];
            // (the first pass of the inverse stores into buf[0], which nobody reads any more: no barrier needed here)
            fft_run<N>(v, o, buf[0], tw, lane);  // = conj(y_even + i y_odd) at m = lane + 512 t
            float h1[8], h2[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                float he, ho;
                hann2(t, he, ho);
                const float ye = he * o[0][t].x, yo = -ho * o[0][t].y;
                if (t < 4) { h1[2 * t] = ye; h1[2 * t + 1] = yo; }
                else { h2[2 * (t - 4)] = ye; h2[2 * (t - 4) + 1] = yo; }
            }
            group_lds_sync<LG>();
            emit(f, h1, h2);
        }
    } else if constexpr (MST_STFT2_BWD_SAVED_MAG && MST_STFT2_BWD_PAIR_8192) {
        // 8192, round 5: two PREDICTION frames per 8192-point complex transform (as for 512 / 2048 above; the target is present through
        // its saved magnitudes), each with its own half-size inverse: two 4096-point sequences per frame instead of three, 16 new
        // 4-byte loads per pair of frames instead of 64.  The transform buffers are destroyed by the first inverse, so the second
        // frame's inverse input waits in registers (28: the kernel runs at two waves per SIMD, where 256 are free).
        constexpr int M = S::M, NB = N / 2 + 1;                    // 4096, 4097
        const float2 we = twg[2 * lane], wo = twg[2 * lane + 1];
        const float* ymrow = a.ymag + (int64_t)row * r.n_frames * NB;
        auto fetchx = [&](int f, int t) { return x[reflect_i32(f * H - H + lane + LG * t, nrow)]; };
        // V[k], V[M - k] of the half-size inverse from the cotangents at bins k and M - k (the algebra of the one-frame path below)
        auto v_pair = [&](int k, float2 Hk, float2 Hm, float2& vk, float2& vm) {
            if (k == 0) {
                vk = make_float2(Hk.x + Hm.x, -(Hk.x - Hm.x));  // V[0] = (H0 + HM) + i (H0 - HM), both real; conj
                vm = vk;
                return;
            }
            Hk = make_float2(0.5f * Hk.x, 0.5f * Hk.y);
            Hm = make_float2(0.5f * Hm.x, 0.5f * Hm.y);
            const float2 w = twg[k];  // W_N^k ; W_N^(M-k) = -conj(W_N^k)
            const float2 Ak = make_float2(Hk.x + Hm.x, Hk.y - Hm.y);
            const float2 Bk = cmul(make_float2(Hk.x - Hm.x, Hk.y + Hm.y), make_float2(w.x, -w.y));
            vk = make_float2(Ak.x - Bk.y, -(Ak.y + Bk.x));
            const float2 Am = make_float2(Hm.x + Hk.x, Hm.y - Hk.y);
            const float2 Bm = cmul(make_float2(Hm.x - Hk.x, Hm.y + Hk.y), make_float2(-w.x, -w.y));
            vm = make_float2(Am.x - Bm.y, -(Am.y + Bm.x));
        };
        const int fstart = F0 == 0 ? -1 : F0;  // frame 0 alone (virtual all-zero partner -1): see the 512 / 2048 path
        float xa1[8];  // first half of frame fa (elements lane + 512 t, t < 8)
#pragma unroll
        for (int t = 0; t < 8; ++t) xa1[t] = fstart >= 0 ? fetchx(fstart, t) : 0.0f;
        for (int fa = fstart; fa < F1; fa += 2) {
            const bool have_a = fa >= 0, have_b = fa + 1 < F1;
            const float* ya = ymrow + (int64_t)(have_a ? fa : 0) * NB;
            const float* yb = ymrow + (int64_t)(have_b ? fa + 1 : 0) * NB;
            float xa2[8], xb2[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                xa2[t] = fetchx(fa + 1, t);                       // second half of frame fa = first half of frame fa + 1
                xb2[t] = have_b ? fetchx(fa + 1, t + 8) : 0.0f;   // = the first half of frame fa + 2
            }
            int li = lane;
            asm volatile("" : "+v"(li));
            const float2 wlf = twg[li];
            L::transform([&](int t) {
                return make_float2(have_a ? (t < 8 ? xa1[t < 8 ? t : 0] : xa2[t >= 8 ? t - 8 : 0]) : 0.0f,
                                   have_b ? (t < 8 ? xa2[t < 8 ? t : 0] : xb2[t >= 8 ? t - 8 : 0]) : 0.0f);
            }, win, buf, tw, wlf, lane);
#pragma unroll
            for (int t = 0; t < 8; ++t) xa1[t] = xb2[t];
            // cotangents of both frames at the bin pair (k, M - k) -> the two frames' (V[k], V[M - k])
            auto both = [&](int k, float2& ak, float2& am, float2& bk, float2& bm) {
                float2 Xa, Xb, Ya, Yb;
                L::split(buf, k, Xa, Xb);
                L::split(buf, M - k, Ya, Yb);
                if (fa == -1) { Xb.y = 0.0f; Yb.y = 0.0f; }  // frame 0: real spectrum
                const float2 z0 = make_float2(0.f, 0.f);
                v_pair(k, have_a ? cotangent_xy(Xa, ya[k], a.eps, coef) : z0, have_a ? cotangent_xy(Ya, ya[M - k], a.eps, coef) : z0, ak, am);
                v_pair(k, have_b ? cotangent_xy(Xb, yb[k], a.eps, coef) : z0, have_b ? cotangent_xy(Yb, yb[M - k], a.eps, coef) : z0, bk, bm);
            };
            static_assert(M / 4 == 2 * LG, "two odd and two even pairs per lane (plus k = M/2 on lane 0)");
            // odd pairs first (their bins live in buf[1], which then becomes the inverse's input buffer), both frames into registers
            float2 Oak[2], Oam[2], Pbk[5], Pbm[5];  // Pb: frame b's pairs, parked until frame a's inverse has run (0, 1 odd; 2 .. 4 even)
#pragma unroll
            for (int i = 0; i < 2; ++i) both(2 * (lane + i * LG) + 1, Oak[i], Oam[i], Pbk[i], Pbm[i]);
            group_lds_sync<LG>();  // every lane has read its odd bins: buf[1] is free
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                Pbk[2 + i] = Pbm[2 + i] = make_float2(0.f, 0.f);
                if (i == 2 && lane != 0) break;
                const int k = i == 2 ? M / 2 : 2 * (lane + i * LG);
                float2 vk, vm;
                both(k, vk, vm, Pbk[2 + i], Pbm[2 + i]);
                buf[1][S::slot(k)] = vk;
                if (k != 0 && k != M / 2) buf[1][S::slot(M - k)] = vm;
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int k = 2 * (lane + i * LG) + 1;
                buf[1][S::slot(k)] = Oak[i];
                buf[1][S::slot(M - k)] = Oam[i];
            }
            // half-size inverse of what sits in buf[1] (passes in buf[0]), window, overlap-add
            auto inverse_emit = [&](int f) {
                group_lds_sync<LG>();
                float2 v[8], o[1][8];
#pragma unroll
                for (int t = 0; t < 8; ++t) v[t] = buf[1][S::slot(lane + LG * t)];
                fft_run<N>(v, o, buf[0], tw, lane);  // = conj(y_even + i y_odd) at m = lane + 512 t
                float h1[8], h2[8];
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const float2 w8 = t == 0 ? make_float2(1.f, 0.f) : (t == 1 ? make_float2(0.70710678118654752f, -0.70710678118654752f)
                                    : (t == 2 ? make_float2(0.f, -1.f) : (t == 3 ? make_float2(-0.70710678118654752f, -0.70710678118654752f)
                                    : (t == 4 ? make_float2(-1.f, 0.f) : (t == 5 ? make_float2(-0.70710678118654752f, 0.70710678118654752f)
                                    : (t == 6 ? make_float2(0.f, 1.f) : make_float2(0.70710678118654752f, 0.70710678118654752f)))))));
                    const float ce = we.x * w8.x - we.y * w8.y, co = wo.x * w8.x - wo.y * w8.y;  // cos(2 pi i / N), i = 2m, 2m + 1
                    const float ye = (0.5f - 0.5f * ce) * o[0][t].x, yo = -(0.5f - 0.5f * co) * o[0][t].y;
                    if (t < 4) { h1[2 * t] = ye; h1[2 * t + 1] = yo; }
                    else { h2[2 * (t - 4)] = ye; h2[2 * (t - 4) + 1] = yo; }
                }
                group_lds_sync<LG>();
                emit(f, h1, h2);
            };
            if (have_a) inverse_emit(fa);
            if (have_b) {
                group_lds_sync<LG>();  // (frame a's inverse input has been read by every lane)
#pragma unroll
                for (int i = 0; i < 5; ++i) {
                    if (i == 4 && lane != 0) break;
                    const int k = i < 2 ? 2 * (lane + i * LG) + 1 : (i == 4 ? M / 2 : 2 * (lane + (i - 2) * LG));
                    buf[1][S::slot(k)] = Pbk[i];
                    if (k != 0 && k != M / 2) buf[1][S::slot(M - k)] = Pbm[i];
                }
                inverse_emit(fa + 1);
            }
        }
    } else if constexpr (MST_STFT2_BWD_SAVED_SPEC_8192) {
        // 8192, round 5: NO forward transform in the backward.  The forward launch kept the prediction's spectrum X and the target's clamped
        // magnitudes (a.xspec, a.ymag: 12 bytes per bin and frame, 51 MB at cfg #2) - the very values the recomputation below would
        // produce - so a frame is: cotangents at the bin pairs (k, M - k) -> V of the half-size inverse -> one 4096-point transform ->
        // window -> overlap-add.  One sequence per frame instead of three, no sample loads; 128 registers = two workgroups per CU.
        constexpr int M = S::M, NB = N / 2 + 1;                    // 4096, 4097
        const float2 we = twg[2 * lane], wo = twg[2 * lane + 1];
        const float2* xrow = reinterpret_cast<const float2*>(a.xspec) + (int64_t)row * r.n_frames * NB;
        const float* ymrow = a.ymag + (int64_t)row * r.n_frames * NB;
        static_assert(M / 2 == 4 * LG, "four bin pairs per lane (plus k = M/2 on lane 0)");
        float2 xk[4], xm[4];
        float yk[4], ymk[4];
        auto request = [&](int f) {
            const float2* xf = xrow + (int64_t)f * NB;
            const float* yf = ymrow + (int64_t)f * NB;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int k = lane + LG * i;
                xk[i] = xf[k]; xm[i] = xf[M - k];
                yk[i] = yf[k]; ymk[i] = yf[M - k];
            }
        };
#ifndef MST_STFT2_BWD_SPEC_PREFETCH
#define MST_STFT2_BWD_SPEC_PREFETCH 1  // the next frame's spectrum is requested behind the current frame's cotangents: 27.2 against 30.6 us (24 registers, free at two waves per SIMD)
#endif
        if (MST_STFT2_BWD_SPEC_PREFETCH) request(F0);
        for (int f = F0; f < F1; ++f) {
            if (!MST_STFT2_BWD_SPEC_PREFETCH) request(f);
            const float2 xh = xrow[(int64_t)f * NB + M / 2];
            const float yh = ymrow[(int64_t)f * NB + M / 2];
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                if (i == 4 && lane != 0) break;
                const int k = i == 4 ? M / 2 : lane + LG * i;
                float2 Hk = cotangent_xy(i == 4 ? xh : xk[i < 4 ? i : 0], i == 4 ? yh : yk[i < 4 ? i : 0], a.eps, coef);
                float2 Hm = cotangent_xy(i == 4 ? xh : xm[i < 4 ? i : 0], i == 4 ? yh : ymk[i < 4 ? i : 0], a.eps, coef);
                float2 vk, vm;
                if (k == 0) {
                    vk = make_float2(Hk.x + Hm.x, -(Hk.x - Hm.x));  // V[0] = (H0 + HM) + i (H0 - HM), both real; conj
                    vm = vk;
                } else {
                    Hk = make_float2(0.5f * Hk.x, 0.5f * Hk.y);
                    Hm = make_float2(0.5f * Hm.x, 0.5f * Hm.y);
                    const float2 w = twg[k];  // W_N^k ; W_N^(M-k) = -conj(W_N^k)
                    const float2 Ak = make_float2(Hk.x + Hm.x, Hk.y - Hm.y);
                    const float2 Bk = cmul(make_float2(Hk.x - Hm.x, Hk.y + Hm.y), make_float2(w.x, -w.y));
                    vk = make_float2(Ak.x - Bk.y, -(Ak.y + Bk.x));
                    const float2 Am = make_float2(Hm.x + Hk.x, Hm.y - Hk.y);
                    const float2 Bm = cmul(make_float2(Hm.x - Hk.x, Hm.y + Hk.y), make_float2(-w.x, -w.y));
                    vm = make_float2(Am.x - Bm.y, -(Am.y + Bm.x));
                }
                buf[1][S::slot(k)] = vk;
                if (k != 0 && k != M / 2) buf[1][S::slot(M - k)] = vm;
            }
            if (MST_STFT2_BWD_SPEC_PREFETCH && f + 1 < F1) request(f + 1);  // the next frame's 48 bytes per pair fly behind this frame's transform
            group_lds_sync<LG>();
            float2 v[8], o[1][8];
#pragma unroll
            for (int t = 0; t < 8; ++t) v[t] = buf[1][S::slot(lane + LG * t)];
            fft_run<N>(v, o, buf[0], tw, lane);  // = conj(y_even + i y_odd) at m = lane + 512 t
            float h1[8], h2[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const float2 w8 = t == 0 ? make_float2(1.f, 0.f) : (t == 1 ? make_float2(0.70710678118654752f, -0.70710678118654752f)
                                : (t == 2 ? make_float2(0.f, -1.f) : (t == 3 ? make_float2(-0.70710678118654752f, -0.70710678118654752f)
                                : (t == 4 ? make_float2(-1.f, 0.f) : (t == 5 ? make_float2(-0.70710678118654752f, 0.70710678118654752f)
                                : (t == 6 ? make_float2(0.f, 1.f) : make_float2(0.70710678118654752f, 0.70710678118654752f)))))));
                const float ce = we.x * w8.x - we.y * w8.y, co = wo.x * w8.x - wo.y * w8.y;  // cos(2 pi i / N), i = 2m, 2m + 1
                const float ye = (0.5f - 0.5f * ce) * o[0][t].x, yo = -(0.5f - 0.5f * co) * o[0][t].y;
                if (t < 4) { h1[2 * t] = ye; h1[2 * t + 1] = yo; }
                else { h2[2 * (t - 4)] = ye; h2[2 * (t - 4) + 1] = yo; }
            }
            group_lds_sync<LG>();
            emit(f, h1, h2);
        }
    } else {
        constexpr int M = S::M;                                    // 4096
        // window of samples 2m, 2m + 1 (m = lane + 512 t): 0.5 - 0.5 Re(W_N^(2 lane + c) W_8^t)
        const float2 we = twg[2 * lane], wo = twg[2 * lane + 1];
#ifndef MST_STFT2_BWD8192_PREFETCH
#define MST_STFT2_BWD8192_PREFETCH 0  // (measured neutral: 49.2 vs 49.5 us) the next frame's 16 sample pairs are requested before the current frame is transformed: 32 more live
                                      // registers, free here - the kernel sits at 184 of the 256 that two waves per SIMD allow
#endif
        constexpr bool PREF8 = MST_STFT2_BWD8192_PREFETCH;
        float2 nxt8[PREF8 ? 16 : 1];
        if (PREF8) {
#pragma unroll
            for (int t = 0; t < 16; ++t) nxt8[t] = fetch(F0, t);
        }
        for (int f = F0; f < F1; ++f) {
            int li = lane;
            asm volatile("" : "+v"(li));
            const float2 wlf = twg[li];
            if (PREF8) {
                float2 cur8[16];
#pragma unroll
                for (int t = 0; t < 16; ++t) cur8[t] = nxt8[t];
                if (f + 1 < F1) {
#pragma unroll
                    for (int t = 0; t < 16; ++t) nxt8[t] = fetch(f + 1, t);
                }
                L::transform([&](int t) { return cur8[t]; }, win, buf, tw, wlf, lane);
            } else {
                L::transform([&](int t) { return fetch(f, t); }, win, buf, tw, wlf, lane);
            }
            // Pair (k, M - k) of the half-size inverse reads the 8192-point bins k, M - k, M + k, N - k: all of k's parity, i.e.
            // all in ONE of the two spectrum buffers (even bins in buf[0], odd in buf[1]).  The odd pairs go first and wait in
            // registers (2 per lane); once every lane has read its odd bins buf[1] is free, and the even pairs (which read
            // buf[0] only) write their values - and the waiting odd ones - straight into buf[1]: 8 staging registers
            // instead of 40 (all pairs parked across one barrier).
            auto pair_v = [&](int k, float2& vk, float2& vm) {
                float2 Hk = cotangent<N>(buf, k, a.eps, coef), Hm = cotangent<N>(buf, M - k, a.eps, coef);
                if (k == 0) {
                    vk = make_float2(Hk.x + Hm.x, -(Hk.x - Hm.x));  // V[0] = (H0 + HM) + i (H0 - HM), both real; conj
                    vm = vk;
                    return;
                }
                Hk = make_float2(0.5f * Hk.x, 0.5f * Hk.y);
                Hm = make_float2(0.5f * Hm.x, 0.5f * Hm.y);
                const float2 w = twg[k];  // W_N^k ; W_N^(M-k) = -conj(W_N^k)
                const float2 Ak = make_float2(Hk.x + Hm.x, Hk.y - Hm.y);
                const float2 Bk = cmul(make_float2(Hk.x - Hm.x, Hk.y + Hm.y), make_float2(w.x, -w.y));
                vk = make_float2(Ak.x - Bk.y, -(Ak.y + Bk.x));
                const float2 Am = make_float2(Hm.x + Hk.x, Hm.y - Hk.y);
                const float2 Bm = cmul(make_float2(Hm.x - Hk.x, Hm.y + Hk.y), make_float2(-w.x, -w.y));
                vm = make_float2(Am.x - Bm.y, -(Am.y + Bm.x));
            };
            static_assert(M / 4 == 2 * LG, "two odd and two even pairs per lane (plus k = M/2 on lane 0)");
            float2 Ok[2], Om[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) pair_v(2 * (lane + i * LG) + 1, Ok[i], Om[i]);
            group_lds_sync<LG>();  // every lane has read its odd bins: buf[1] is free
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                if (i == 2 && lane != 0) break;
                const int k = i == 2 ? M / 2 : 2 * (lane + i * LG);
                float2 vk, vm;
                pair_v(k, vk, vm);
                buf[1][S::slot(k)] = vk;
                if (k != 0 && k != M / 2) buf[1][S::slot(M - k)] = vm;
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int k = 2 * (lane + i * LG) + 1;
                buf[1][S::slot(k)] = Ok[i];
                buf[1][S::slot(M - k)] = Om[i];
            }
            group_lds_sync<LG>();
            float2 v[8], o[1][8];
#pragma unroll
            for (int t = 0; t < 8; ++t) v[t] = buf[1][S::slot(lane + LG * t)];
            // (the first pass of the inverse stores into buf[0], which nobody reads any more: no barrier needed here)
            fft_run<N>(v, o, buf[0], tw, lane);  // = conj(y_even + i y_odd) at m = lane + 512 t
            float h1[8], h2[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const float2 w8 = t == 0 ? make_float2(1.f, 0.f) : (t == 1 ? make_float2(0.70710678118654752f, -0.70710678118654752f)
                                : (t == 2 ? make_float2(0.f, -1.f) : (t == 3 ? make_float2(-0.70710678118654752f, -0.70710678118654752f)
                                : (t == 4 ? make_float2(-1.f, 0.f) : (t == 5 ? make_float2(-0.70710678118654752f, 0.70710678118654752f)
                                : (t == 6 ? make_float2(0.f, 1.f) : make_float2(0.70710678118654752f, 0.70710678118654752f)))))));
                const float ce = we.x * w8.x - we.y * w8.y, co = wo.x * w8.x - wo.y * w8.y;  // cos(2 pi i / N), i = 2m, 2m + 1
                const float ye = (0.5f - 0.5f * ce) * o[0][t].x, yo = -(0.5f - 0.5f * co) * o[0][t].y;
                if (t < 4) { h1[2 * t] = ye; h1[2 * t + 1] = yo; }
                else { h2[2 * (t - 4)] = ye; h2[2 * (t - 4) + 1] = yo; }
            }
            group_lds_sync<LG>();
            emit(f, h1, h2);
        }
    }
    // trailing second half of a strip that does not end the row: the seam shared with the next strip (halo mode: the next
    // strip recomputes this frame and owns the block)
    if (SEAMS && have_carry) seam(F1 - 1, carry, false);
}

template <int N>
__global__ __launch_bounds__(FftPlan<N>::LG, (N == 8192 ? MST_STFT2_W8192_BWD : (N == 2048 ? MST_STFT2_W2048_BWD : 1))) void k_stft2_bwd(StftArgs a) {
    using S = FftShape<N>;
    __shared__ __attribute__((aligned(16))) float2 buf[S::NSEQ][S::SLOTS];
    __shared__ __attribute__((aligned(16))) float2 hb[S::NSEQ == 1 ? S::SLOTS : 1];  // conj(He_a + i He_b)
    stft2_bwd_body<N>(a, threadIdx.x, blockIdx.x, gridDim.x, blockIdx.y, buf, hb);
}

// Round 5: the 512- and the 2048-point backward in ONE launch.  Both walk strips of four hop blocks (L512 = L2048 = 4), and 2048-point
// blocks are four times as long: the 4096 samples of 2048-point strip g are exactly the 512-point strips 4 g .. 4 g + 3.  A 256-lane
// workgroup first runs those four 512-point strips, one per wave (no workgroup barrier in that part), then - behind one barrier that
// also drains its stores - the 2048-point strip, which read-modify-writes the samples the same workgroup has just written: no second
// launch, and the re-read comes out of this CU's own cache levels.  LDS and registers are the 2048-point kernel's (4 x 9 KB = 36 KB,
// <= 128): four workgroups per CU as before.  Strips are balanced (lengths may differ by one block); other strip lengths measured with
// the kept spectra (us): L = 3: 51.9, 4: 44.4, 5: 47.9, 6: 51.0 - a strip is a dependent chain, fewer inverses per row do not pay for a longer one.
#ifndef MST_STFT2_W512_2048_BWD
#define MST_STFT2_W512_2048_BWD 4  // waves per SIMD asked of the fused launch (128 registers, no spills)
#endif
__global__ __launch_bounds__(256, MST_STFT2_W512_2048_BWD) void k_stft2_bwd_512_2048(StftArgs a512, StftArgs a2048) {
    using S5 = FftShape<512>;
    using S2 = FftShape<2048>;
    __shared__ __attribute__((aligned(16))) float2 lds[2 * S2::SLOTS];
    static_assert(8 * S5::SLOTS <= 2 * S2::SLOTS, "four waves' 512-point buffers fit the 2048-point kernel's");
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // the 2048-point strip's hop blocks [B0, B1) are the 512-point blocks [4 B0, 4 B1): a quarter of them per wave (balanced strips may differ
    // in length by one block - the quarters follow their own strip, not a split of the whole row)
    int B0, B1;
    strip_range(blockIdx.x, gridDim.x, (int)(a2048.n / 1024), B0, B1);
    const int q = B1 - B0;
    stft2_bwd_body<512>(a512, tid & 63, 4 * blockIdx.x + wave, 4 * gridDim.x, blockIdx.y,
                        reinterpret_cast<float2(*)[S5::SLOTS]>(lds + 2 * wave * S5::SLOTS), lds + (2 * wave + 1) * S5::SLOTS,
                        4 * B0 + wave * q, 4 * B0 + (wave + 1) * q);
    __syncthreads();  // s_waitcnt vmcnt(0) + barrier: the four strips' stores have left before any lane of the workgroup reads them back
    stft2_bwd_body<2048>(a2048, tid, blockIdx.x, gridDim.x, blockIdx.y, reinterpret_cast<float2(*)[S2::SLOTS]>(lds), lds + S2::SLOTS);
}
bool stft2_bwd_can_fuse(int64_t n) { return MST_STFT2_BWD_L512 == MST_STFT2_BWD_L2048 && n % 1024 == 0 && n / 1024 >= 2 * MST_STFT2_BWD_L2048; }
void launch_stft2_bwd_512_2048(const StftArgs& a512, const StftArgs& a2048, int rows, hipStream_t stream) {
    const int G = stft2_bwd_groups(2048, a2048.r.n_frames, rows);
    hipLaunchKernelGGL(k_stft2_bwd_512_2048, dim3(G, rows), dim3(256), 0, stream, a512, a2048);
}

int stft2_bwd_groups(int n_fft, int n_frames, int rows) {
    const int B = n_frames - 1;
    if (n_fft == 8192) {
        // seam mode: strips of >= 2 frames, the last one of >= 3 (it finishes block B - 2 itself).  The kernel takes 192 registers:
        // one 512-lane workgroup per CU, 256 resident.  When two-frame strips would need more than one round, the rows are cut
        // into 256 / rows longer strips instead - the same frame-times per CU, but one prologue and one pair of seams per
        // strip less (cfg #2: 512 x 2 frames = 66.8 us, 256 x 4 frames = 56.9 us; 3 frames = 1.4 rounds = 75.8 us)
        int G = n_frames / 2;
        if ((int64_t)rows * G > kStft2Bwd8192Slots) {
            const int fit = kStft2Bwd8192Slots / (rows > 0 ? rows : 1);
            G = fit >= 1 ? (fit < G ? fit : G) : 1;
        }
        while (G > 1 && n_frames - (int)(((int64_t)(G - 1) * n_frames) / G) < 3) --G;
        return G > 0 ? G : 1;
    }
    const int L = n_fft == 512 ? MST_STFT2_BWD_L512 : MST_STFT2_BWD_L2048;  // halo mode: >= 2 blocks per strip
    const int G = B / L;
    return G > 0 ? G : 1;
}
bool stft2_bwd_needs_zero(int n_fft) { return n_fft == 8192; }

void launch_stft2_bwd(const StftArgs& a, int n_groups, int rows, hipStream_t stream) {
    const dim3 grid(n_groups, rows);
    if (a.r.n_fft == 512) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_stft2_bwd<512>), grid, dim3(FftPlan<512>::LG), 0, stream, a);
    else if (a.r.n_fft == 2048) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_stft2_bwd<2048>), grid, dim3(FftPlan<2048>::LG), 0, stream, a);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_stft2_bwd<8192>), grid, dim3(FftPlan<8192>::LG), 0, stream, a);
}

void launch_stft3_fwd(const Stft3Args& p, hipStream_t stream) {
    hipLaunchKernelGGL(k_stft3_fwd, dim3(p.wg_end[2]), dim3(512), 0, stream, p);
}

void launch_stft2_fwd(const StftArgs& a, int n_groups, int rows, hipStream_t stream) {
    const dim3 grid(n_groups, rows);
    if (a.r.n_fft == 512) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_stft2_fwd<512>), grid, dim3(FftPlan<512>::LG), 0, stream, a);
    else if (a.r.n_fft == 2048) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_stft2_fwd<2048>), grid, dim3(FftPlan<2048>::LG), 0, stream, a);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_stft2_fwd<8192>), grid, dim3(FftPlan<8192>::LG), 0, stream, a);
}

}  // namespace mst
