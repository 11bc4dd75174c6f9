// mst_af.hip - AudioFeatureLoss (reference mst/loss.py:198-260 and the five transforms :62-195):
//   rms, crest factor, stereo width, stereo imbalance  (closed-form reductions over the stereo mix)
//   Bark spectrum: mid/side -> STFT(32768, hop 8192, periodic Hann, reflect) -> |X| -> mean over
//   frames -> (24 x 16385) filterbank -> log(. + 1e-8)
// each compared with the target by MSE and weighted.  Forward and reverse-mode.
//
// The transforms live in mst_af2.hip (register-radix engine); this file holds the closed-form features, the reductions
// around the Bark spectrum, the overlap-add gather of the adjoint frames and the C ABI.
#include "mst_af.h"

namespace mst {

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
    if (t < kAfHalf) {
        const double ang = 6.283185307179586476925 * (double)t / (double)kAfHalf;
        tables[kAfTwH + 2 * t] = (float)cos(ang);
        tables[kAfTwH + 2 * t + 1] = (float)(-sin(ang));
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
// frames of a (signal, half) unit are cut into G strips (one workgroup each, kAf2Slots co-resident): choose G to
// minimise  rounds x longest strip  = ceil(G U / slots) x ceil(F / G);  e.g. F = 33, U = 64: G = 8 -> 1 x 5 (G = 9: 2 x 4)
static int af_groups(int n_frames, int n_signals) {
    int best = 1;
    int64_t best_cost = INT64_MAX;
    for (int g = 1; g <= n_frames; ++g) {
        const int64_t rounds = ((int64_t)g * n_signals + kAf2Slots - 1) / kAf2Slots, len = (n_frames + g - 1) / g;
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
    p.n_groups = af_groups(p.n_frames, 8 * bs);
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
    launch_af2_bark_fwd(a, stream);
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
    launch_af2_bark_bwd(a, stream);
    hipLaunchKernelGGL(k_af_bwd_gather, dim3((unsigned)((n_samples + 1023) / 1024), bs), dim3(256), 0, stream, a);
    return (int)hipGetLastError();
}
