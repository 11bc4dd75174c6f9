// mst_cnn_net.hip - the spectrogram encoder around the MFMA convolutions of mst_cnn.hip: STFT front end, first layer,
// BatchNorm / ReLU / average pooling and their adjoints, the pooling head, and the launch sequences behind the C ABI
// (mst_spectrogram_forward, mst_cnn14_forward, mst_cnn14_backward; include/diffmst_hip.h).
//
// Reference: SpectrogramEncoder.forward (mst/modules.py:772-806): torch.stft(n_fft 2048, hop 512, Hann, centre-padded) ->
// (|X| + 1e-8)^0.3 -> Cnn14 (mst/panns.py:126-209): six ConvBlocks (conv3x3 - BN - ReLU - conv3x3 - BN - ReLU - avg pool;
// :27-85) with pools (2,2) (4,4) (4,2) (4,2) (4,2) (2,2) over (bins, frames), mean over bins, max + mean over frames, Linear.
// Image convention here: H = frames, W = bins (mst_cnn.h).
#include "mst_cnn.h"
#include "mst_fft2.h"

namespace mst {

__device__ __forceinline__ float bf2f_(bf16_t v) { return __uint_as_float((uint32_t)v << 16); }
__device__ __forceinline__ bf16_t f2bf_(float f) {
    uint32_t u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
// eight consecutive channels of an NHWC tensor
__device__ __forceinline__ void load8(const bf16_t* p, float* v) {
    const uint4 u = *reinterpret_cast<const uint4*>(p);
    const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[2 * i] = __uint_as_float(w[i] << 16);
        v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
}
__device__ __forceinline__ void load8(const float* p, float* v) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void store8(bf16_t* p, const float* v) {
    uint4 u;
    u.x = (uint32_t)f2bf_(v[0]) | ((uint32_t)f2bf_(v[1]) << 16);
    u.y = (uint32_t)f2bf_(v[2]) | ((uint32_t)f2bf_(v[3]) << 16);
    u.z = (uint32_t)f2bf_(v[4]) | ((uint32_t)f2bf_(v[5]) << 16);
    u.w = (uint32_t)f2bf_(v[6]) | ((uint32_t)f2bf_(v[7]) << 16);
    *reinterpret_cast<uint4*>(p) = u;
}
__device__ __forceinline__ void store8(float* p, const float* v) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
template <typename T> __device__ __forceinline__ T cvt_out(float v);
template <> __device__ __forceinline__ float cvt_out<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t cvt_out<bf16_t>(float v) { return f2bf_(v); }

// =====================================================================================================================
// STFT front end: spec[row][frame][bin] = (|STFT(x)| + 1e-8)^0.3, n_fft 2048, periodic Hann, reflect-padded centre frames.
// Two consecutive frames share one complex transform (frame f real, f + 1 imaginary) on the register-radix engine.
constexpr int kSpecN = 2048, kSpecLanes = FftPlan<2048>::LG;
__global__ void k_spec_tables(float* tables) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= kSpecN) return;
    const double ang = 6.283185307179586476925 * (double)t / (double)kSpecN;
    tables[2 * t] = (float)cos(ang);
    tables[2 * t + 1] = (float)(-sin(ang));
    const float ph = 6.283185307179586f * (float)t / (float)kSpecN;  // torch.hann_window: fp32 phase, then the cosine
    tables[2 * kSpecN + t] = 0.5f - 0.5f * (float)cos((double)ph);
}
struct SpecArgs {
    const float* x;
    float* spec;
    const float* tables;
    int64_t n;
    int hop, frames, pairs_per_block;
};
__global__ __launch_bounds__(kSpecLanes) void k_spectrogram(SpecArgs a) {
    using S = FftShape<kSpecN>;
    constexpr int LG = kSpecLanes, PTS = kSpecN / LG, BINS = kSpecN / 2 + 1;
    __shared__ __attribute__((aligned(16))) float2 buf[S::SLOTS];
    const int lane = threadIdx.x, row = blockIdx.y;
    const float2* twg = reinterpret_cast<const float2*>(a.tables);
    LaneTw<kSpecN> tw;
    tw.init(twg, lane);
    float win[PTS];
#pragma unroll
    for (int t = 0; t < PTS; ++t) win[t] = a.tables[2 * kSpecN + lane + LG * t];
    const float* x = a.x + (int64_t)row * a.n;
    const int nrow = (int)a.n;
    auto refl = [&](int i) {
        i = i < 0 ? -i : i;
        return i >= nrow ? 2 * (nrow - 1) - i : i;
    };
    const int pairs = (a.frames + 1) / 2;
    for (int pp = 0; pp < a.pairs_per_block; ++pp) {
        const int pair = blockIdx.x * a.pairs_per_block + pp;
        if (pair >= pairs) break;
        const int f0 = 2 * pair, f1 = f0 + 1;
        const bool has1 = f1 < a.frames;
        float2 v[8], o[S::NBL][S::RL];
#pragma unroll
        for (int t = 0; t < PTS; ++t) {
            const int i = lane + LG * t - kSpecN / 2;
            const float xa = x[refl(f0 * a.hop + i)], xb = has1 ? x[refl(f1 * a.hop + i)] : 0.f;
            v[t] = make_float2(win[t] * xa, win[t] * xb);
        }
        fft_run<kSpecN>(v, o, buf, tw, lane);
        group_lds_sync<LG>();
#pragma unroll
        for (int u = 0; u < S::NBL; ++u)
#pragma unroll
            for (int t = 0; t < S::RL; ++t) buf[S::slot(lane + u * LG + t * (S::M / S::RL))] = o[u][t];
        group_lds_sync<LG>();
        float* oa = a.spec + ((int64_t)row * a.frames + f0) * BINS;
        float* ob = oa + BINS;
        for (int k = lane; k < BINS; k += LG) {
            const float2 zk = buf[S::slot(k)], zn = buf[S::slot((kSpecN - k) & (kSpecN - 1))];
            const float xr = 0.5f * (zk.x + zn.x), xi = 0.5f * (zk.y - zn.y);
            const float yr = 0.5f * (zk.y + zn.y), yi = -0.5f * (zk.x - zn.x);
            const float ma = sqrtf(xr * xr + xi * xi) + 1e-8f, mb = sqrtf(yr * yr + yi * yi) + 1e-8f;
            oa[k] = __builtin_amdgcn_exp2f(0.3f * __builtin_amdgcn_logf(ma));
            if (has1) ob[k] = __builtin_amdgcn_exp2f(0.3f * __builtin_amdgcn_logf(mb));
        }
        group_lds_sync<LG>();
    }
}

// =====================================================================================================================
// weights: torch (co, ci, kh, kw) fp32 -> forward (co, tap, ci) T and data-gradient (ci, 8 - tap, co) T with tap = a * 3 + b,
// a = offset along frames = kw, b = offset along bins = kh (transposed image)
template <typename T>
__global__ void k_prep_weights(const float* __restrict__ w, T* __restrict__ wf, T* __restrict__ wd, int Cin, int Cout) {
    const int64_t total = (int64_t)Cout * Cin * 9;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int ci = (int)(e % Cin), tap = (int)((e / Cin) % 9), co = (int)(e / ((int64_t)Cin * 9));
        const int a_ = tap / 3, b_ = tap % 3;
        const float v = w[(((int64_t)co * Cin + ci) * 3 + b_) * 3 + a_];
        wf[e] = cvt_out<T>(v);
        if (wd) wd[((int64_t)ci * 9 + (8 - tap)) * Cout + co] = cvt_out<T>(v);
    }
}
// the same for Cin, Cout multiples of 32, through a (32 co x 32 ci x 9) LDS tile: 1152-byte runs in, 32-element rows out in
// both layouts (the element-wise kernel scatters the data-gradient layout in 2-byte writes: 1.4 ms per step for 80 M weights)
template <typename T>
__global__ __launch_bounds__(256) void k_prep_weights_tiled(const float* __restrict__ w, T* __restrict__ wf, T* __restrict__ wd, int Cin, int Cout) {
    constexpr int ROW = 32 * 9 + 2;  // odd dword pitch for T = bf16 and for T = float
    __shared__ T tile[32 * ROW];
    const int ci0 = blockIdx.x * 32, co0 = blockIdx.y * 32, tid = threadIdx.x;
    for (int i = tid; i < 32 * 288; i += 256) {
        const int co = i / 288, j = i % 288;  // j = ci * 9 + kh * 3 + kw
        tile[co * ROW + j] = cvt_out<T>(w[((int64_t)(co0 + co) * Cin + ci0) * 9 + j]);
    }
    __syncthreads();
    const int l32 = tid & 31;
    for (int r = tid >> 5; r < 32 * 9; r += 8) {
        const int o = r / 9, tap = r % 9, inner = (tap % 3) * 3 + tap / 3;  // source index kh * 3 + kw of tap = kw * 3 + kh
        wf[((int64_t)(co0 + o) * 9 + tap) * Cin + ci0 + l32] = tile[o * ROW + l32 * 9 + inner];
        wd[((int64_t)(ci0 + o) * 9 + (8 - tap)) * Cout + co0 + l32] = tile[l32 * ROW + o * 9 + inner];
    }
}
// partial sums (splits, 9, Cout, Cin) -> torch layout (co, ci, kh, kw), fixed order.  A workgroup folds 64 consecutive sums (256-byte
// rows of the slabs); its four 64-lane groups take every fourth slab, four loads in flight each, and meet in LDS (one thread per sum
// walking all slabs: 360 us for the 64 -> 64 layer's 1024 slabs)
__global__ __launch_bounds__(256) void k_wgrad_reduce(const float* __restrict__ part, float* __restrict__ gw, int Cin, int Cout, int splits) {
    // workgroup = (output channel co, 64 input channels): nine 256-byte rows of every slab in, 576 CONSECUTIVE floats of the torch layout
    // out (one thread per sum wrote 4 bytes every 36: the re-layout of the deep layers' single slab was a 150 MB scatter)
    __shared__ float red[4][9][64];
    __shared__ float outv[9 * 64];
    const int64_t total = (int64_t)Cout * Cin * 9;
    const int l = threadIdx.x & 63, sg = threadIdx.x >> 6, cib = Cin / 64;
    for (int64_t blk = blockIdx.x; blk < (int64_t)Cout * cib; blk += gridDim.x) {
        const int co = (int)(blk / cib), ci0 = (int)(blk % cib) * 64;
        float acc[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[t] = 0.f;
        for (int s = sg; s < splits; s += 4) {  // the four lane groups take every fourth slab; nine loads in flight per lane
            const float* p = part + (int64_t)s * total + (int64_t)co * Cin + ci0 + l;
            float v[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) v[t] = p[(int64_t)t * Cout * Cin];  // slab order (tap, co, ci)
#pragma unroll
            for (int t = 0; t < 9; ++t) acc[t] += v[t];
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) red[sg][t][l] = acc[t];
        __syncthreads();
        for (int i = threadIdx.x; i < 9 * 64; i += 256) {
            const int t = i / 64, c = i % 64, a_ = t / 3, b_ = t % 3;  // tap = a_ * 3 + b_  ->  torch index kh * 3 + kw = b_ * 3 + a_
            outv[c * 9 + b_ * 3 + a_] = (red[0][t][c] + red[1][t][c]) + (red[2][t][c] + red[3][t][c]);
        }
        __syncthreads();
        float* o = gw + ((int64_t)co * Cin + ci0) * 9;
        for (int i = threadIdx.x; i < 9 * 64; i += 256) o[i] = outv[i];
        __syncthreads();
    }
}
// first layer (Cin = 1, partial sums (splits, Cout, 16)): 576 sums over ~2000 pixel splits each - one workgroup per output channel,
// 16 lanes per tap take every 16th split, folded in a fixed order (one thread per output walking all splits: 520 us)
__global__ __launch_bounds__(256) void k_wgrad_reduce_first(const float* __restrict__ part, float* __restrict__ gw, int Cout, int splits) {
    __shared__ float red[16][16];
    const int co = blockIdx.x, tap = threadIdx.x & 15, lane16 = threadIdx.x >> 4;
    float acc = 0.f;
    for (int s = lane16; s < splits; s += 16) acc += part[((int64_t)s * Cout + co) * 16 + tap];
    red[lane16][tap] = acc;
    __syncthreads();
    if (threadIdx.x < 9) {
        float v = 0.f;
        for (int j = 0; j < 16; ++j) v += red[j][threadIdx.x];
        const int a_ = threadIdx.x / 3, b_ = threadIdx.x % 3;
        gw[((int64_t)co * 3 + b_) * 3 + a_] = v;
    }
}

// =====================================================================================================================
// first layer: one input channel (the fp32 spectrogram), 64 outputs; direct form.  Eight lanes per pixel, eight channels per lane:
// a wave instruction stores eight whole 128-byte pixels (one pixel x 64 channels per lane strode the stores by 128 bytes - 64
// separate 16-byte segments per instruction, 780 us for 1.08 GB); the nine taps are fetched by all eight lanes of a pixel (L1 hits).
constexpr int kConv1Pix = 2048;
template <typename T>
__global__ __launch_bounds__(256) void k_conv1(const float* __restrict__ spec, const float* __restrict__ w /* (64, tap) */, T* __restrict__ out,
                                               float* __restrict__ part, int N, int H, int W) {
    __shared__ float red[4][64][2];
    const int tid = threadIdx.x, oct = tid & 7, pl = tid >> 3;
    float wr[8][9];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int t = 0; t < 9; ++t) wr[j][t] = w[(oct * 8 + j) * 9 + t];
    const int HW = H * W;
    const int64_t P = (int64_t)N * HW;
    float s1[8], s2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s1[j] = s2[j] = 0.f;
    int64_t p = (int64_t)blockIdx.x * kConv1Pix + pl;
    int r = (int)(p % HW);
    for (int it = 0; it < kConv1Pix / 32; ++it, p += 32) {
        if (p < P) {
            const int h = r / W, ww = r - h * W;
            float x[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int dh = t / 3 - 1, dw = t % 3 - 1;
                const bool ok = (unsigned)(h + dh) < (unsigned)H && (unsigned)(ww + dw) < (unsigned)W;
                x[t] = ok ? spec[p + (int64_t)dh * W + dw] : 0.f;
            }
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float acc = 0.f;
#pragma unroll
                for (int t = 0; t < 9; ++t) acc = fmaf(wr[j][t], x[t], acc);
                v[j] = acc;
                s1[j] += acc;
                s2[j] = fmaf(acc, acc, s2[j]);
            }
            store8(out + p * 64 + oct * 8, v);
        }
        r += 32;
        while (r >= HW) r -= HW;
    }
    if (part) {
        const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float a = s1[j], b = s2[j];
#pragma unroll
            for (int m = 8; m < 64; m <<= 1) {
                a += __shfl_xor(a, m);
                b += __shfl_xor(b, m);
            }
            if (lane < 8) {
                red[wave][oct * 8 + j][0] = a;
                red[wave][oct * 8 + j][1] = b;
            }
        }
        __syncthreads();
        if (tid < 128) {
            const int c = tid >> 1, q = tid & 1;
            part[((int64_t)blockIdx.x * 64 + c) * 2 + q] = (red[0][c][q] + red[1][c][q]) + (red[2][c][q] + red[3][c][q]);
        }
    }
}

// =====================================================================================================================
// BatchNorm2d (training: batch statistics, biased variance; eval: running statistics).  stat = {mean, invstd, var} x C.
// column sums of a (tiles, C, 2) fp32 partial array, first stage: workgroup (channel group of 32, slice s of S) folds its
// slice of the tiles in fp64, fixed order -> (S, C, 2) doubles.  (One thread per channel walking every tile was 60 % of the
// encoder step: 65 k tiles per layer at 16 signals, each read 8 bytes out of a 512-byte-strided line.)
constexpr int kColSlices = 64;
__global__ __launch_bounds__(256) void k_colsum(const float* __restrict__ part, int tiles, int C, double* __restrict__ out) {
    __shared__ double red[8][32][2];
    const int cl = threadIdx.x & 31, tl = threadIdx.x >> 5, c = blockIdx.x * 32 + cl, S = gridDim.y, s = blockIdx.y;
    const int per = (tiles + S - 1) / S, t0 = s * per, t1 = (t0 + per < tiles) ? t0 + per : tiles;
    double s1 = 0.0, s2 = 0.0;
    for (int t = t0 + tl; t < t1; t += 8) {
        const float2 v = *reinterpret_cast<const float2*>(part + ((int64_t)t * C + c) * 2);
        s1 += (double)v.x;
        s2 += (double)v.y;
    }
    red[tl][cl][0] = s1;
    red[tl][cl][1] = s2;
    __syncthreads();
    if (tl == 0) {
        for (int q = 1; q < 8; ++q) {
            s1 += red[q][cl][0];
            s2 += red[q][cl][1];
        }
        out[((int64_t)s * C + c) * 2] = s1;
        out[((int64_t)s * C + c) * 2 + 1] = s2;
    }
}
static int colsum_slices(int tiles) {
    const int s = (tiles + 63) / 64;
    return s < 1 ? 1 : (s > kColSlices ? kColSlices : s);
}
// fold of the slice sums of channel c in ascending order, eight 16-byte loads in flight (one dependent L2 round trip per slice made the
// two finalize kernels 18-20 us each, 24 of them per step)
__device__ __forceinline__ void colsum_fold(const double* __restrict__ part, int slices, int C, int c, double& s1, double& s2) {
    for (int t0 = 0; t0 < slices; t0 += 8) {
        double2 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            v[u] = t0 + u < slices ? *reinterpret_cast<const double2*>(part + ((int64_t)(t0 + u) * C + c) * 2) : make_double2(0.0, 0.0);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (t0 + u < slices) {
                s1 += v[u].x;
                s2 += v[u].y;
            }
        }
    }
}
__global__ void k_bn_finalize(const double* __restrict__ part, int tiles, int C, double count, const float* __restrict__ run_mean,
                              const float* __restrict__ run_var, int training, float eps, float* __restrict__ stat, float* __restrict__ batch_stats) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float mean, var;
    if (training) {
        double s1 = 0.0, s2 = 0.0;
        colsum_fold(part, tiles, C, c, s1, s2);
        const double m = s1 / count;
        double v = s2 / count - m * m;
        v = v < 0.0 ? 0.0 : v;
        mean = (float)m;
        var = (float)v;
        if (batch_stats) {
            batch_stats[c] = mean;
            batch_stats[2048 + c] = var;
        }
    } else {
        mean = run_mean[c];
        var = run_var[c];
    }
    stat[c] = mean;
    stat[C + c] = (float)(1.0 / sqrt((double)var + (double)eps));
    stat[2 * C + c] = var;
}

template <typename T>
__global__ void k_bn_relu(const T* __restrict__ raw, T* __restrict__ out, const float* __restrict__ stat, const float* __restrict__ gamma,
                          const float* __restrict__ beta, int64_t chunks, int C) {
    // the grid stride (a multiple of 256) is a multiple of C / 8 <= 256: a thread stays on ONE channel group - its constants are loaded once
    const int cg = C / 8, c0 = (int)((blockIdx.x * blockDim.x + threadIdx.x) % cg) * 8;
    float sc[8], mu[8], be[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        sc[j] = gamma[c0 + j] * stat[C + c0 + j];
        mu[j] = stat[c0 + j];
        be[j] = beta[c0 + j];
    }
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < chunks; e += (int64_t)gridDim.x * blockDim.x) {
        float v[8];
        load8(raw + e * 8, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = fmaxf(fmaf(v[j] - mu[j], sc[j], be[j]), 0.f);
        store8(out + e * 8, v);
    }
}
template <typename T>
__global__ void k_bn_relu_pool(const T* __restrict__ raw, T* __restrict__ out, const float* __restrict__ stat, const float* __restrict__ gamma,
                               const float* __restrict__ beta, int N, int H, int W, int C, int ph, int pw) {
    const int cg = C / 8, Ho = H / ph, Wo = W / pw;
    const int64_t chunks = (int64_t)N * Ho * Wo * cg;
    const float inv = 1.0f / (float)(ph * pw);
    const int c0 = (int)((blockIdx.x * blockDim.x + threadIdx.x) % cg) * 8;  // constant per thread (see k_bn_relu)
    float sc[8], mu[8], be[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        sc[j] = gamma[c0 + j] * stat[C + c0 + j];
        mu[j] = stat[c0 + j];
        be[j] = beta[c0 + j];
    }
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < chunks; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = e / cg;  // pooled pixel (n, ho, wo); cg is a power of two
        int n, ho, wo;
        if (r < 0x7fffffff) {  // 32-bit divisions (a 64-bit one is ~100 instructions)
            const unsigned ru = (unsigned)r, q = ru / (unsigned)Wo;
            wo = (int)(ru - q * (unsigned)Wo);
            n = (int)(q / (unsigned)Ho);
            ho = (int)(q - (unsigned)n * (unsigned)Ho);
        } else {
            wo = (int)(r % Wo);
            ho = (int)((r / Wo) % Ho);
            n = (int)(r / Wo / Ho);
        }
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
        for (int i = 0; i < ph; ++i)
            for (int k = 0; k < pw; ++k) {
                float v[8];
                load8(raw + (((int64_t)n * H + ho * ph + i) * W + wo * pw + k) * C + c0, v);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += fmaxf(fmaf(v[j] - mu[j], sc[j], be[j]), 0.f);
            }
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] *= inv;
        store8(out + e * 8, acc);
    }
}

struct BnBwdArgs {
    const void* raw;   // (N, H, W, C) conv output
    const void* g_up;
    void* d_raw;       // out of the apply pass
    const float* stat; // mean, invstd
    const float* gamma;
    const float* beta;
    float* part;       // (strips, C, 2)
    const float* coef; // apply pass: (C, 3) = gamma invstd, mean g, mean g xhat
    int N, H, W, C, ph, pw, strips;
    int64_t pix_per_strip;
};
template <typename T, bool POOL>
__device__ __forceinline__ void bn_bwd_pixel(const BnBwdArgs& a, int64_t p, int c0, const float* mu, const float* is, const float* ga, const float* be,
                                             float* g, float* xh) {
    const T* raw = reinterpret_cast<const T*>(a.raw);
    const T* gu = reinterpret_cast<const T*>(a.g_up);
    float v[8];
    load8(raw + p * a.C + c0, v);
    bool inside = true;
    float scale = 1.0f;
    int64_t gp = p;
    if (POOL) {
        const int Ho = a.H / a.ph, Wo = a.W / a.pw, HW = a.H * a.W;
        int r, n;
        if (p < 0x7fffffff) {  // 32-bit divisions (a 64-bit one is ~100 instructions, and every 16-byte chunk pays it)
            n = (int)((unsigned)p / (unsigned)HW);
            r = (int)((unsigned)p - (unsigned)n * (unsigned)HW);
        } else {
            n = (int)(p / HW);
            r = (int)(p - (int64_t)n * HW);
        }
        const int h = r / a.W, ho = h / a.ph, wo = (r - h * a.W) / a.pw;
        inside = ho < Ho && wo < Wo;
        gp = ((int64_t)n * Ho + ho) * Wo + wo;
        scale = 1.0f / (float)(a.ph * a.pw);
    }
    if (inside) load8(gu + gp * a.C + c0, g);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        xh[j] = (v[j] - mu[j]) * is[j];
        const float y = fmaf(v[j] - mu[j], ga[j] * is[j], be[j]);  // the forward kernels' expression, bit for bit: the mask is the one of OUR forward
        g[j] = (inside && y > 0.f) ? g[j] * scale : 0.f;
    }
}
template <typename T, bool POOL>
__global__ __launch_bounds__(256) void k_bn_bwd_reduce(BnBwdArgs a) {
    __shared__ float red[256][17];
    const int tid = threadIdx.x, cg = a.C / 8;
    const int cgw = cg < 256 ? cg : 256, pl = 256 / cgw;
    const int cc = blockIdx.y * cgw + tid % cgw, lane_p = tid / cgw, c0 = cc * 8;
    float mu[8], is[8], ga[8], be[8], s1[8], s2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        mu[j] = a.stat[c0 + j];
        is[j] = a.stat[a.C + c0 + j];
        ga[j] = a.gamma[c0 + j];
        be[j] = a.beta[c0 + j];
        s1[j] = s2[j] = 0.f;
    }
    const int64_t P = (int64_t)a.N * a.H * a.W;
    const int64_t pb = (int64_t)blockIdx.x * a.pix_per_strip;
    int64_t pe = pb + a.pix_per_strip;
    pe = pe < P ? pe : P;
    for (int64_t p = pb + lane_p; p < pe; p += pl) {
        float g[8], xh[8];
        bn_bwd_pixel<T, POOL>(a, p, c0, mu, is, ga, be, g, xh);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            s1[j] += g[j];
            s2[j] = fmaf(g[j], xh[j], s2[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        red[tid][j] = s1[j];
        red[tid][8 + j] = s2[j];
    }
    __syncthreads();
    if (tid < cgw) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float t1 = 0.f, t2 = 0.f;
            for (int q = 0; q < pl; ++q) {
                t1 += red[q * cgw + tid][j];
                t2 += red[q * cgw + tid][8 + j];
            }
            float* o = a.part + ((int64_t)blockIdx.x * a.C + c0 + j) * 2;
            o[0] = t1;
            o[1] = t2;
        }
    }
}
__global__ void k_bn_bwd_finalize(const double* __restrict__ part, int strips, int C, double count, const float* __restrict__ stat,
                                  const float* __restrict__ gamma, int training, float* __restrict__ coef, float* __restrict__ g_gamma,
                                  float* __restrict__ g_beta, double grad_scale) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s1 = 0.0, s2 = 0.0;
    colsum_fold(part, strips, C, c, s1, s2);
    // grad_scale = 1 / ranks when the sums span every rank: (global sum) / world is what an averaging DistributedDataParallel
    // leaves of the per-rank sums torch.nn.SyncBatchNorm hands it
    g_beta[c] = (float)(s1 * grad_scale);
    g_gamma[c] = (float)(s2 * grad_scale);
    coef[c * 3] = gamma[c] * stat[C + c];
    coef[c * 3 + 1] = training ? (float)(s1 / count) : 0.f;  // eval mode: the statistics are constants
    coef[c * 3 + 2] = training ? (float)(s2 / count) : 0.f;
}
template <typename T, bool POOL>
__global__ void k_bn_bwd_apply(BnBwdArgs a) {
    const int cg = a.C / 8;
    const int64_t chunks = (int64_t)a.N * a.H * a.W * cg;
    T* d = reinterpret_cast<T*>(a.d_raw);
    const int c0 = (int)((blockIdx.x * blockDim.x + threadIdx.x) % cg) * 8;  // constant per thread (see k_bn_relu)
    float mu[8], is[8], ga[8], be[8], k0[8], k1[8], k2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        mu[j] = a.stat[c0 + j];
        is[j] = a.stat[a.C + c0 + j];
        ga[j] = a.gamma[c0 + j];
        be[j] = a.beta[c0 + j];
        k0[j] = a.coef[(c0 + j) * 3];
        k1[j] = a.coef[(c0 + j) * 3 + 1];
        k2[j] = a.coef[(c0 + j) * 3 + 2];
    }
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < chunks; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = e / cg;
        float g[8], xh[8];
        bn_bwd_pixel<T, POOL>(a, p, c0, mu, is, ga, be, g, xh);
#pragma unroll
        for (int j = 0; j < 8; ++j) g[j] = k0[j] * (g[j] - k1[j] - xh[j] * k2[j]);
        store8(d + e * 8, g);
    }
}

// =====================================================================================================================
// head: feat[n][c] = max_h m[h] + mean_h m[h], m[h] = mean_w x6[n][h][w][c]  (panns.py:198-203); embed = feat fc_w^T + fc_b
template <typename T>
__global__ void k_head_feat(const T* __restrict__ x, float* __restrict__ feat, int* __restrict__ arg, int N, int H, int W, int C) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * C) return;
    const int n = i / C, c = i % C;
    float best = -INFINITY, sum = 0.f;
    int bi = 0;
    for (int h = 0; h < H; ++h) {
        float m = 0.f;
        for (int w = 0; w < W; ++w) {
            const T v = x[(((int64_t)n * H + h) * W + w) * C + c];
            m += sizeof(T) == 2 ? bf2f_((bf16_t)v) : (float)v;
        }
        m /= (float)W;
        sum += m;
        if (m > best) {  // first maximum, like torch.max
            best = m;
            bi = h;
        }
    }
    feat[i] = best + sum / (float)H;
    arg[i] = bi;
}
__global__ __launch_bounds__(256) void k_fc_fwd(const float* __restrict__ feat, const float* __restrict__ w, const float* __restrict__ b,
                                                float* __restrict__ out, int N, int E, int C) {
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (o >= N * E) return;
    const int n = o / E, e = o % E;
    float acc = 0.f;
    for (int c = lane; c < C; c += 64) acc = fmaf(feat[(int64_t)n * C + c], w[(int64_t)e * C + c], acc);
    acc = wave_sum(acc);
    if (lane == 0) out[o] = acc + b[e];
}
// g_feat[n][c] = sum_e g[n][e] w[e][c]
// workgroup = 64 features of one signal; its four 64-lane groups take every fourth embedding row, eight loads in flight each, and meet
// in LDS in a fixed order (one thread per feature walking all E rows of fc_w: 115 us of L2 latency)
__global__ __launch_bounds__(256) void k_fc_bwd_feat(const float* __restrict__ g, const float* __restrict__ w, float* __restrict__ g_feat, int N, int E, int C) {
    __shared__ float red[4][64];
    const int l = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int blocks_per_n = C / 64, n = blockIdx.x / blocks_per_n, c = (blockIdx.x % blocks_per_n) * 64 + l;
    float acc = 0.f;
    for (int e0 = q; e0 < E; e0 += 32) {
        float wv[8], gv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = e0 + 4 * u;
            wv[u] = e < E ? w[(int64_t)e * C + c] : 0.f;
            gv[u] = e < E ? g[(int64_t)n * E + e] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = fmaf(gv[u], wv[u], acc);
    }
    red[q][l] = acc;
    __syncthreads();
    if (q == 0) g_feat[(int64_t)n * C + c] = (red[0][l] + red[1][l]) + (red[2][l] + red[3][l]);
}
// g_w[e][c] = sum_n g[n][e] feat[n][c];  g_b[e] = sum_n g[n][e]
__global__ void k_fc_bwd_w(const float* __restrict__ g, const float* __restrict__ feat, float* __restrict__ g_w, float* __restrict__ g_b, int N,
                           int E, int C) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= E * C) return;
    const int e = i / C, c = i % C;
    float acc = 0.f, accb = 0.f;
    for (int n = 0; n < N; ++n) {
        const float ge = g[(int64_t)n * E + e];
        acc = fmaf(ge, feat[(int64_t)n * C + c], acc);
        accb += ge;
    }
    g_w[i] = acc;
    if (c == 0) g_b[e] = accb;
}
template <typename T>
__global__ void k_head_scatter(const float* __restrict__ g_feat, const int* __restrict__ arg, T* __restrict__ gx, int N, int H, int W, int C) {
    const int64_t total = (int64_t)N * H * W * C;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(e % C);
        int64_t r = e / C;
        r /= W;
        const int h = (int)(r % H), n = (int)(r / H);
        const float gf = g_feat[(int64_t)n * C + c];
        gx[e] = cvt_out<T>(gf * (1.0f / (float)(H * W) + (arg[(int64_t)n * C + c] == h ? 1.0f / (float)W : 0.f)));
    }
}

// =====================================================================================================================
// plan: buffer offsets (bytes) of one forward / backward pass
constexpr int kBlocks = 6;
static const int kChan[kBlocks + 1] = {1, 64, 128, 256, 512, 1024, 2048};
static const int kPoolH[kBlocks] = {2, 4, 2, 2, 2, 2};  // along frames (reference pool_size[1])
static const int kPoolW[kBlocks] = {2, 4, 4, 4, 4, 2};  // along bins   (reference pool_size[0])
struct CnnPlan {
    bool ok;
    int n, esz;
    int H[kBlocks + 1], W[kBlocks + 1];
    size_t raw1[kBlocks], a1[kBlocks], raw2[kBlocks], out[kBlocks];
    size_t wf[2 * kBlocks], wd[2 * kBlocks], stat[2 * kBlocks];
    size_t w1, bnpart, colsum, coef, feat, arg, gfeat, ga, gb, wgpart;
    size_t bnpart_bytes, wgpart_bytes, kpart, kpart_bytes;
    size_t total;
};
static int64_t wgrad_part_floats(int precision, int W, int64_t P, int Cin, int Cout, int* splits, int* steps) {
    const int64_t ksteps = (P + kWgradPix - 1) / kWgradPix;
    int tiles;
#ifndef MST_WGRAD4_WGS
#define MST_WGRAD4_WGS 512  // workgroups of the nine-tap kernel (two per CU, one round; 1024: +4 % kernel time and twice the slabs): its 9 x Cout x Cin fp32 slab per split is written and re-read
#endif
    if (conv_wgrad_nine_taps(precision, W, Cin, Cout)) tiles = (2048 / MST_WGRAD4_WGS) * (Cout / 64) * (Cin / 64);
    else if (Cin == 1) tiles = 1;
    else if (Cin % 128 == 0 && Cout % 128 == 0) tiles = (Cout / 128) * (Cin / 128) * 9;
    else tiles = (Cout / 64) * (Cin / 64) * 9;
    int64_t s = (2048 + tiles - 1) / tiles;
    const int64_t smax = (ksteps + 7) / 8;  // at least 8 K steps per split
    s = s > smax ? smax : s;
    s = s < 1 ? 1 : s;
    const int64_t per = (ksteps + s - 1) / s;
    s = (ksteps + per - 1) / per;
    *splits = (int)s;
    *steps = (int)per;
    return Cin == 1 ? s * Cout * 16 : s * 9 * (int64_t)Cout * Cin;
}
static CnnPlan cnn_plan(const mst_cnn14_desc* d) {
    CnnPlan p{};
    if (!d || d->n <= 0 || d->frames < 64 || d->bins < 256 || d->embed_dim <= 0 || (d->precision < 0 || d->precision > 3)) return p;
    p.n = d->n;
    p.esz = d->precision == 0 ? 2 : 4;
    p.H[0] = d->frames;
    p.W[0] = d->bins;
    for (int b = 0; b < kBlocks; ++b) {
        p.H[b + 1] = p.H[b] / kPoolH[b];
        p.W[b + 1] = p.W[b] / kPoolW[b];
        if (p.H[b + 1] < 1 || p.W[b + 1] < 1) return p;
    }
    size_t o = 0;
    auto take = [&](size_t bytes) {
        const size_t at = o;
        o += (bytes + 255) / 256 * 256;
        return at;
    };
    size_t gmax = 0, bnmax = 0, wgmax = 0, kmax = 0;
    for (int b = 0; b < kBlocks; ++b) {
        const int64_t P = (int64_t)p.n * p.H[b] * p.W[b];
        const size_t full = (size_t)P * kChan[b + 1] * p.esz;
        p.raw1[b] = take(full);
        p.a1[b] = take(full);
        p.raw2[b] = take(full);
        p.out[b] = take((size_t)p.n * p.H[b + 1] * p.W[b + 1] * kChan[b + 1] * p.esz);
        gmax = full > gmax ? full : gmax;
        for (int k = 0; k < 2; ++k) {
            const int Cin = k == 0 ? kChan[b] : kChan[b + 1], Cout = kChan[b + 1];
            p.stat[2 * b + k] = take((size_t)3 * Cout * 4);
            if (Cin > 1) {
                p.wf[2 * b + k] = take((size_t)9 * Cin * Cout * p.esz);
                p.wd[2 * b + k] = take((size_t)9 * Cin * Cout * p.esz);
            }
            const size_t tiles = Cin == 1 ? (size_t)((P + kConv1Pix - 1) / kConv1Pix) : (size_t)conv_pixel_tiles(p.n, p.H[b], p.W[b]);
            const size_t strips = 4096;
            size_t bn = (tiles > strips ? tiles : strips) * Cout * 2 * 4;
            bnmax = bn > bnmax ? bn : bnmax;
            int sp, st;
            const size_t wg = (size_t)wgrad_part_floats(p.esz == 2 ? 0 : 1, p.W[b], P, Cin, Cout, &sp, &st) * 4;
            wgmax = wg > wgmax ? wg : wgmax;
            if (Cin > 1) {  // split-K scratch of the forward convolution and of its data gradient (channel roles swapped)
                const size_t k1 = conv_splitk_bytes(p.n, p.H[b], p.W[b], Cin, Cout), k2 = conv_splitk_bytes(p.n, p.H[b], p.W[b], Cout, Cin);
                kmax = k1 > kmax ? k1 : kmax;
                kmax = k2 > kmax ? k2 : kmax;
            }
        }
    }
    p.w1 = take(64 * 9 * 4);
    p.bnpart_bytes = bnmax;
    p.bnpart = take(bnmax);
    p.colsum = take((size_t)kColSlices * 2048 * 2 * 8);
    p.coef = take((size_t)2048 * 3 * 4);
    p.feat = take((size_t)p.n * 2048 * 4);
    p.arg = take((size_t)p.n * 2048 * 4);
    p.gfeat = take((size_t)p.n * 2048 * 4);
    p.ga = take(gmax);
    p.gb = take(gmax);
    p.wgpart_bytes = wgmax;
    p.wgpart = take(wgmax);
    p.kpart_bytes = kmax;
    p.kpart = take(kmax);
    p.total = o;
    p.ok = true;
    return p;
}
static int ew_grid(int64_t items) {
    int64_t g = (items + 255) / 256;
    return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g));
}

template <typename T>
static int cnn_forward_t(const mst_cnn14_desc* d, const CnnPlan& p, const float* spec, const mst_cnn14_params* prm, float* embed, float* batch_stats,
                         char* ws, hipStream_t s, mst_sync_fn sync, void* user) {
    const int prec = d->precision;
    const bool synced = sync && d->training && d->world > 1;  // statistics over every rank's signals (SyncBatchNorm)
    const double ranks = synced ? (double)d->world : 1.0;
    // weights of this call, in both operand layouts (the backward reuses them)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_prep_weights<float>), dim3(3), dim3(256), 0, s, prm->conv_w[0], (float*)(ws + p.w1), (float*)nullptr, 1, 64);
    for (int l = 1; l < 2 * kBlocks; ++l) {
        const int b = l / 2, Cin = (l & 1) ? kChan[b + 1] : kChan[b], Cout = kChan[b + 1];
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_prep_weights_tiled<T>), dim3(Cin / 32, Cout / 32), dim3(256), 0, s, prm->conv_w[l],
                           (T*)(ws + p.wf[l]), (T*)(ws + p.wd[l]), Cin, Cout);
    }
    const void* x_in = spec;
    for (int b = 0; b < kBlocks; ++b) {
        const int H = p.H[b], W = p.W[b], C = kChan[b + 1];
        const int64_t P = (int64_t)p.n * H * W;
        for (int k = 0; k < 2; ++k) {
            const int l = 2 * b + k, Cin = k == 0 ? kChan[b] : C;
            T* raw = (T*)(ws + (k == 0 ? p.raw1[b] : p.raw2[b]));
            float* part = (float*)(ws + p.bnpart);
            int tiles;
            if (Cin == 1) {
                tiles = (int)((P + kConv1Pix - 1) / kConv1Pix);
                hipLaunchKernelGGL(HIP_KERNEL_NAME(k_conv1<T>), dim3(tiles), dim3(256), 0, s, (const float*)x_in, (const float*)(ws + p.w1), raw,
                                   d->training ? part : nullptr, p.n, H, W);
            } else {
                ConvArgs ca{x_in, ws + p.wf[l], raw, d->training ? part : nullptr, p.n, H, W, Cin, C, 0, nullptr};
                tiles = launch_conv3x3(prec, ca, s, (float*)(ws + p.kpart), p.kpart_bytes);
            }
            float* stat = (float*)(ws + p.stat[l]);
            double* cs = (double*)(ws + p.colsum);
            const int slices = colsum_slices(tiles);
            if (d->training) hipLaunchKernelGGL(k_colsum, dim3(C / 32, slices), dim3(256), 0, s, part, tiles, C, cs);
            if (synced) sync(user, cs, (size_t)slices * C * 2, (void*)s);  // slice sums of every rank added element-wise; the fold below sees the global sums
            hipLaunchKernelGGL(k_bn_finalize, dim3((C + 255) / 256), dim3(256), 0, s, cs, slices, C, (double)P * ranks, prm->bn_mean[l], prm->bn_var[l],
                               d->training, d->bn_eps, stat, batch_stats ? batch_stats + (size_t)l * 2 * 2048 : nullptr);
            if (k == 0) {
                T* a1 = (T*)(ws + p.a1[b]);
                hipLaunchKernelGGL(HIP_KERNEL_NAME(k_bn_relu<T>), dim3(ew_grid(P * C / 8)), dim3(256), 0, s, raw, a1, stat, prm->bn_gamma[l],
                                   prm->bn_beta[l], P * C / 8, C);
                x_in = a1;
            } else {
                T* out = (T*)(ws + p.out[b]);
                hipLaunchKernelGGL(HIP_KERNEL_NAME(k_bn_relu_pool<T>), dim3(ew_grid((int64_t)p.n * p.H[b + 1] * p.W[b + 1] * C / 8)), dim3(256), 0, s,
                                   raw, out, stat, prm->bn_gamma[l], prm->bn_beta[l], p.n, H, W, C, kPoolH[b], kPoolW[b]);
                x_in = out;
            }
        }
    }
    const int H6 = p.H[kBlocks], W6 = p.W[kBlocks];
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_head_feat<T>), dim3((p.n * 2048 + 255) / 256), dim3(256), 0, s, (const T*)x_in, (float*)(ws + p.feat),
                       (int*)(ws + p.arg), p.n, H6, W6, 2048);
    hipLaunchKernelGGL(k_fc_fwd, dim3((p.n * d->embed_dim + 3) / 4), dim3(256), 0, s, (const float*)(ws + p.feat), prm->fc_w, prm->fc_b, embed, p.n,
                       d->embed_dim, 2048);
    return (int)hipGetLastError();
}

template <typename T>
static int cnn_backward_t(const mst_cnn14_desc* d, const CnnPlan& p, const float* spec, const mst_cnn14_params* prm, const float* g_embed,
                          const mst_cnn14_grads* gr, char* ws, hipStream_t s, mst_sync_fn sync, void* user) {
    const int prec = d->precision, E = d->embed_dim;
    const bool synced = sync && d->training && d->world > 1;
    const double ranks = synced ? (double)d->world : 1.0;
    const int H6 = p.H[kBlocks], W6 = p.W[kBlocks];
    hipLaunchKernelGGL(k_fc_bwd_feat, dim3(p.n * (2048 / 64)), dim3(256), 0, s, g_embed, prm->fc_w, (float*)(ws + p.gfeat), p.n, E, 2048);
    hipLaunchKernelGGL(k_fc_bwd_w, dim3((E * 2048 + 255) / 256), dim3(256), 0, s, g_embed, (const float*)(ws + p.feat), gr->fc_w, gr->fc_b, p.n, E, 2048);
    T* GA = (T*)(ws + p.ga);
    T* GB = (T*)(ws + p.gb);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_head_scatter<T>), dim3(ew_grid((int64_t)p.n * H6 * W6 * 2048)), dim3(256), 0, s, (const float*)(ws + p.gfeat),
                       (const int*)(ws + p.arg), GB, p.n, H6, W6, 2048);
    float* part = (float*)(ws + p.bnpart);
    float* coef = (float*)(ws + p.coef);
    float* wgpart = (float*)(ws + p.wgpart);
    for (int b = kBlocks - 1; b >= 0; --b) {
        const int H = p.H[b], W = p.W[b], C = kChan[b + 1];
        const int64_t P = (int64_t)p.n * H * W;
        for (int k = 1; k >= 0; --k) {
            const int l = 2 * b + k, Cin = k == 0 ? kChan[b] : C;
            // (pool) - ReLU - BatchNorm adjoint: cotangent of the conv output into GA
            const int cg = C / 8, cgw = cg < 256 ? cg : 256, pl = 256 / cgw;
            int64_t strips = 4096 / (cg / cgw);
            int64_t per = (P + strips - 1) / strips;
            per = (per + pl - 1) / pl * pl;
            per = per < pl ? pl : per;
            strips = (P + per - 1) / per;
            BnBwdArgs ba{ws + (k == 0 ? p.raw1[b] : p.raw2[b]), GB, GA, (const float*)(ws + p.stat[l]), prm->bn_gamma[l], prm->bn_beta[l], part, coef,
                         p.n, H, W, C, kPoolH[b], kPoolW[b], (int)strips, per};
            const dim3 rgrid((unsigned)strips, cg / cgw);
            if (k == 1) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_bn_bwd_reduce<T, true>), rgrid, dim3(256), 0, s, ba);
            else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_bn_bwd_reduce<T, false>), rgrid, dim3(256), 0, s, ba);
            double* cs = (double*)(ws + p.colsum);
            const int slices = colsum_slices((int)strips);
            hipLaunchKernelGGL(k_colsum, dim3(C / 32, slices), dim3(256), 0, s, part, (int)strips, C, cs);
            if (synced) sync(user, cs, (size_t)slices * C * 2, (void*)s);  // sum g, sum g xhat over every rank (the adjoint of the shared statistics)
            hipLaunchKernelGGL(k_bn_bwd_finalize, dim3((C + 255) / 256), dim3(256), 0, s, cs, slices, C, (double)P * ranks, (const float*)(ws + p.stat[l]),
                               prm->bn_gamma[l], d->training, coef, gr->bn_gamma[l], gr->bn_beta[l], 1.0 / ranks);
            if (k == 1) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_bn_bwd_apply<T, true>), dim3(ew_grid(P * cg)), dim3(256), 0, s, ba);
            else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_bn_bwd_apply<T, false>), dim3(ew_grid(P * cg)), dim3(256), 0, s, ba);
            // weight gradient
            const void* x_in = k == 1 ? (const void*)(ws + p.a1[b]) : (b == 0 ? (const void*)spec : (const void*)(ws + p.out[b - 1]));
            int splits, steps;
            wgrad_part_floats(prec, W, P, Cin, C, &splits, &steps);
            WgradArgs wa{GA, x_in, wgpart, p.n, H, W, Cin, C, splits, steps};
            launch_conv_wgrad(prec, wa, s);
            if (Cin == 1) hipLaunchKernelGGL(k_wgrad_reduce_first, dim3(C), dim3(256), 0, s, wgpart, gr->conv_w[l], C, splits);
            else hipLaunchKernelGGL(k_wgrad_reduce, dim3(ew_grid((int64_t)C * (Cin / 64) * 256)), dim3(256), 0, s, wgpart, gr->conv_w[l], Cin, C, splits);
            // data gradient (not for the spectrogram itself)
            if (Cin > 1) {
                ConvArgs ca{GA, ws + p.wd[l], GB, nullptr, p.n, H, W, C, Cin, 0, nullptr};
                launch_conv3x3(prec, ca, s, (float*)(ws + p.kpart), p.kpart_bytes);
            }
        }
    }
    return (int)hipGetLastError();
}

}  // namespace mst

using namespace mst;

extern "C" size_t mst_spectrogram_tables_bytes(void) { return (size_t)3 * kSpecN * sizeof(float); }
extern "C" int mst_spectrogram_init_tables(void* tables, void* stream) {
    if (!tables) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_spec_tables, dim3(kSpecN / 256), dim3(256), 0, (hipStream_t)stream, (float*)tables);
    return (int)hipGetLastError();
}
extern "C" int mst_spectrogram_forward(const float* x, int32_t rows, int64_t n_samples, int32_t n_fft, int32_t hop, const void* tables, float* spec,
                                       void* stream) {
    if (!x || !tables || !spec || rows <= 0 || n_fft != kSpecN || hop <= 0 || n_samples <= n_fft / 2 || n_samples >= (1ll << 30)) return hipErrorInvalidValue;
    const int frames = 1 + (int)(n_samples / hop), pairs = (frames + 1) / 2;
    int ppb = (int)(((int64_t)pairs * rows + 2047) / 2048);
    ppb = ppb < 1 ? 1 : ppb;
    SpecArgs a{x, spec, (const float*)tables, n_samples, hop, frames, ppb};
    hipLaunchKernelGGL(k_spectrogram, dim3((pairs + ppb - 1) / ppb, rows), dim3(kSpecLanes), 0, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}

extern "C" size_t mst_cnn14_workspace_bytes(const mst_cnn14_desc* d) {
    const CnnPlan p = cnn_plan(d);
    return p.ok ? p.total : 0;
}
extern "C" int mst_cnn14_forward(const mst_cnn14_desc* d, const float* spec, const mst_cnn14_params* params, float* embed, float* batch_stats,
                                 void* workspace, size_t workspace_bytes, void* stream) {
    const CnnPlan p = cnn_plan(d);
    if (!p.ok || !spec || !params || !embed || !workspace || workspace_bytes < p.total || ((uintptr_t)workspace & 255)) return hipErrorInvalidValue;
    return mst_cnn14_forward_sync(d, spec, params, embed, batch_stats, workspace, workspace_bytes, stream, nullptr, nullptr);
}
extern "C" int mst_cnn14_forward_sync(const mst_cnn14_desc* d, const float* spec, const mst_cnn14_params* params, float* embed, float* batch_stats,
                                      void* workspace, size_t workspace_bytes, void* stream, mst_sync_fn sync, void* user) {
    const CnnPlan p = cnn_plan(d);
    if (!p.ok || !spec || !params || !embed || !workspace || workspace_bytes < p.total || ((uintptr_t)workspace & 255)) return hipErrorInvalidValue;
    if (d->precision == 0) return cnn_forward_t<bf16_t>(d, p, spec, params, embed, batch_stats, (char*)workspace, (hipStream_t)stream, sync, user);
    return cnn_forward_t<float>(d, p, spec, params, embed, batch_stats, (char*)workspace, (hipStream_t)stream, sync, user);
}
extern "C" int mst_cnn14_backward(const mst_cnn14_desc* d, const float* spec, const mst_cnn14_params* params, const float* grad_embed,
                                  const mst_cnn14_grads* grads, void* workspace, size_t workspace_bytes, void* stream) {
    const CnnPlan p = cnn_plan(d);
    if (!p.ok || !spec || !params || !grad_embed || !grads || !workspace || workspace_bytes < p.total) return hipErrorInvalidValue;
    return mst_cnn14_backward_sync(d, spec, params, grad_embed, grads, workspace, workspace_bytes, stream, nullptr, nullptr);
}
extern "C" int mst_cnn14_backward_sync(const mst_cnn14_desc* d, const float* spec, const mst_cnn14_params* params, const float* grad_embed,
                                       const mst_cnn14_grads* grads, void* workspace, size_t workspace_bytes, void* stream, mst_sync_fn sync,
                                       void* user) {
    const CnnPlan p = cnn_plan(d);
    if (!p.ok || !spec || !params || !grad_embed || !grads || !workspace || workspace_bytes < p.total) return hipErrorInvalidValue;
    if (d->precision == 0) return cnn_backward_t<bf16_t>(d, p, spec, params, grad_embed, grads, (char*)workspace, (hipStream_t)stream, sync, user);
    return cnn_backward_t<float>(d, p, spec, params, grad_embed, grads, (char*)workspace, (hipStream_t)stream, sync, user);
}
