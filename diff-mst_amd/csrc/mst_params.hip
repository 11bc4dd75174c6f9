// mst_params.hip - parameter kernels of the console:
//   k_prep      normalised params -> range check, denormalise, RBJ biquad design, compressor /
//               pan constants, and the S-step transition-matrix power tables the carry scans use
//   k_prep_bwd  reduce the per-workgroup partial sums and chain them back to the NORMALISED params
//
// Reference semantics restated: denormalize / range check mst/modules.py:71-97; index maps
// :353-460; biquad design + compressor constants + pan law = dasp-pytorch 0.0.1.
#include "mst_kernels.h"

#ifndef MST_PREP_STOP
#define MST_PREP_STOP 0  // timing diagnostics only (wrong results): k_prep returns after stage 1 / 2 / 3
#endif
namespace mst {

// ---------------------------------------------------------------------------------------------
// fp32-faithful design: the same sequence of fp32 roundings the reference's torch-CPU ops make
// (no FMA contraction); transcendental values come from fp64 libm rounded once to fp32.
// kind: 0 low shelf, 1 peaking, 2 high shelf.  out = {b0,b1,b2,a1,a2}/a0.
// ---------------------------------------------------------------------------------------------
__device__ void design_section_f32(int kind, float gain_db, float freq, float q, float sr, float* out) {
#pragma clang fp contract(off)
    // 10^x = 2^(x log2 10) in fp64: far below one fp32 ulp from the correctly rounded value, and much cheaper than pow()
    const float A = (float)exp2((double)(gain_db / 40.0f) * 3.321928094887362);
    const float w0 = 6.283185307179586f * (freq / sr);
    double sn64, cw64;
    sincos((double)w0, &sn64, &cw64);
    const float sn = (float)sn64;
    const float cw = (float)cw64;
    const float alpha = sn / (2.0f * q);
    const float sA = sqrtf(A);
    float b0, b1, b2, a0, a1, a2;
    if (kind == 1) {
        const float aA = alpha * A;
        const float aoA = alpha / A;
        b0 = 1.0f + aA;
        b1 = -2.0f * cw;
        b2 = 1.0f - aA;
        a0 = 1.0f + aoA;
        a1 = -2.0f * cw;
        a2 = 1.0f - aoA;
    } else {
        const float Ap1 = A + 1.0f, Am1 = A - 1.0f;
        const float t = (2.0f * sA) * alpha;
        const float m1 = Am1 * cw, p1 = Ap1 * cw;
        if (kind == 0) {
            b0 = A * ((Ap1 - m1) + t);
            b1 = (2.0f * A) * (Am1 - p1);
            b2 = A * ((Ap1 - m1) - t);
            a0 = (Ap1 + m1) + t;
            a1 = -2.0f * (Am1 + p1);
            a2 = (Ap1 + m1) - t;
        } else {
            b0 = A * ((Ap1 + m1) + t);
            b1 = (-2.0f * A) * (Am1 + p1);
            b2 = A * ((Ap1 + m1) - t);
            a0 = (Ap1 - m1) + t;
            a1 = 2.0f * (Am1 - p1);
            a2 = (Ap1 - m1) - t;
        }
    }
    out[0] = b0 / a0;
    out[1] = b1 / a0;
    out[2] = b2 / a0;
    out[3] = a1 / a0;
    out[4] = a2 / a0;
}

// forward-mode dual numbers (value + d/d{gain_db, freq, q}) for the design Jacobian
struct D3 {
    double v, d[3];
};
__device__ inline D3 dconst(double c) { return D3{c, {0, 0, 0}}; }
__device__ inline D3 operator+(D3 a, D3 b) { return D3{a.v + b.v, {a.d[0] + b.d[0], a.d[1] + b.d[1], a.d[2] + b.d[2]}}; }
__device__ inline D3 operator-(D3 a, D3 b) { return D3{a.v - b.v, {a.d[0] - b.d[0], a.d[1] - b.d[1], a.d[2] - b.d[2]}}; }
__device__ inline D3 operator*(D3 a, D3 b) {
    return D3{a.v * b.v, {a.d[0] * b.v + a.v * b.d[0], a.d[1] * b.v + a.v * b.d[1], a.d[2] * b.v + a.v * b.d[2]}};
}
__device__ inline D3 operator/(D3 a, D3 b) {
    const double r = 1.0 / b.v, q = a.v * r;
    return D3{q, {(a.d[0] - q * b.d[0]) * r, (a.d[1] - q * b.d[1]) * r, (a.d[2] - q * b.d[2]) * r}};
}
__device__ inline D3 dscale(D3 a, double s) { return D3{a.v * s, {a.d[0] * s, a.d[1] * s, a.d[2] * s}}; }
__device__ inline D3 dfun(D3 a, double f, double fp) { return D3{f, {a.d[0] * fp, a.d[1] * fp, a.d[2] * fp}}; }

__device__ void design_section_dual(int kind, double gain_db, double freq, double q, double sr, D3* out) {
    D3 g{gain_db, {1, 0, 0}}, f{freq, {0, 1, 0}}, qq{q, {0, 0, 1}};
    const double Av = exp2(gain_db / 40.0 * 3.321928094887362);
    D3 A = dfun(g, Av, Av * 2.302585092994046 / 40.0);
    D3 w0 = dscale(f, 6.283185307179586 / sr);
    double snv, csv;
    sincos(w0.v, &snv, &csv);
    D3 sn = dfun(w0, snv, csv);
    D3 cw = dfun(w0, csv, -snv);
    D3 alpha = sn / dscale(qq, 2.0);
    const double sAv = sqrt(Av);
    D3 sA = dfun(A, sAv, 0.5 / sAv);
    D3 one = dconst(1.0);
    D3 b0, b1, b2, a0, a1, a2;
    if (kind == 1) {
        b0 = one + alpha * A;
        b1 = dscale(cw, -2.0);
        b2 = one - alpha * A;
        a0 = one + alpha / A;
        a1 = dscale(cw, -2.0);
        a2 = one - alpha / A;
    } else {
        D3 Ap1 = A + one, Am1 = A - one, t = dscale(sA * alpha, 2.0);
        if (kind == 0) {
            b0 = A * (Ap1 - Am1 * cw + t);
            b1 = dscale(A * (Am1 - Ap1 * cw), 2.0);
            b2 = A * (Ap1 - Am1 * cw - t);
            a0 = Ap1 + Am1 * cw + t;
            a1 = dscale(Am1 + Ap1 * cw, -2.0);
            a2 = Ap1 + Am1 * cw - t;
        } else {
            b0 = A * (Ap1 + Am1 * cw + t);
            b1 = dscale(A * (Am1 + Ap1 * cw), -2.0);
            b2 = A * (Ap1 + Am1 * cw - t);
            a0 = Ap1 - Am1 * cw + t;
            a1 = dscale(Am1 - Ap1 * cw, 2.0);
            a2 = Ap1 - Am1 * cw - t;
        }
    }
    out[0] = b0 / a0;
    out[1] = b1 / a0;
    out[2] = b2 / a0;
    out[3] = a1 / a0;
    out[4] = a2 / a0;
}

__device__ __forceinline__ float denorm(float v, float lo, float hi) {
#pragma clang fp contract(off)
    return v * (hi - lo) + lo;
}
__device__ __forceinline__ int section_kind(int k) { return k == 0 ? 0 : (k == 5 ? 2 : 1); }

// LDS matrix product C = A*B for 12x12 doubles, one element per lane (lanes 0..143 of the group)
__device__ __forceinline__ void mat12_mul(const double* A, const double* B, double* C, int e) {
    const int i = e / 12, j = e % 12;
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < 12; ++k) acc = fma(A[i * 12 + k], B[k * 12 + j], acc);  // explicit: this file is built with -ffp-contract=off
    C[e] = acc;
}

// Two workgroups (320 lanes) per filter row: rows [0,R) are tracks, [R,R+bs) master buses.  Both design the
// row (identical values, identical stores); workgroup y = 0 then builds the 12x12 cascade tables, y = 1 the
// all-pole tables - two serial fp64 chains that would otherwise run back to back.
__global__ __launch_bounds__(320) void k_prep(PrepArgs a) {
    __shared__ double mats[2][3][144];  // [fwd|adj][cur, tmp, acc]
    __shared__ double wv[2][kEqChunk][12];  // A^t b, t < 64: response of the chunk's end state to an impulse t samples before its end
    __shared__ double dblk[2][2][kSections][4];  // diagonal 2x2 blocks of M and M^64, [set][fwd|adj][section]
    __shared__ float coef[32];
    const int row = blockIdx.x, tid = threadIdx.x;
    if (blockIdx.y >= 2) {
        // Round 5, prefetch riders: this launch is two serial fp64 chains per filter row on 2 (R + bs) workgroups - the memory system idles
        // for its ~20 us - and the launch after it opens with a pass over the track rows that is bound by HBM (the zero-state map of the
        // EQ: 67 MB at cfg #2).  The extra workgroups (dispatched behind the real ones) pull the track rows through the Infinity Cache
        // meanwhile: plain loads whose values are dropped.  Purely a hint: nothing reads what they "produce".
        if (row < a.R && a.pf_src) {
            const float4* src = reinterpret_cast<const float4*>(a.pf_src + (int64_t)row * a.pf_stride);
            const int nseg = gridDim.y - 2, seg = blockIdx.y - 2;
            const int64_t nv = a.pf_n >> 2, per = (nv + nseg - 1) / nseg, v0 = seg * per, v1 = v0 + per < nv ? v0 + per : nv;
            for (int64_t g0 = v0 + tid; g0 < v1; g0 += 320 * 8) {
                float4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = g0 + 320 * u < v1 ? src[g0 + 320 * u] : make_float4(0.f, 0.f, 0.f, 0.f);
#if defined(__clang__)
#pragma unroll
                for (int u = 0; u < 8; ++u) asm volatile("" ::"v"(v[u].x), "v"(v[u].y), "v"(v[u].z), "v"(v[u].w));
#endif
            }
        }
        return;
    }
    // arm the granules through which later launches of this call exchange block aggregates (mst_common.h)
    for (int64_t i = ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 320 + tid; i < a.gran_n; i += (int64_t)gridDim.x * 2 * 320) a.gran[i] = 0ull;
    const bool is_master = row >= a.R;
    const int mrow = row - a.R;
    const mst_console_desc& d = a.d;
    const float sr = d.sample_rate;
    const float* p = is_master ? a.master_params + (int64_t)mrow * MST_NUM_MASTER_PARAMS
                               : a.track_params + (int64_t)row * MST_NUM_TRACK_PARAMS;
    const float* lo = is_master ? d.master_lo : d.track_lo;
    const float* hi = is_master ? d.master_hi : d.track_hi;
    float* rc = is_master ? a.rc_m + (int64_t)mrow * RC_STRIDE : a.rc_t + (int64_t)row * RC_STRIDE;
    const int eq0 = is_master ? 0 : 1;     // index of low_shelf_gain_db
    const int cmp0 = is_master ? 18 : 19;  // index of threshold_db
    const int np = is_master ? MST_NUM_MASTER_PARAMS : MST_NUM_TRACK_PARAMS;

    // ---- range check (reference mst/modules.py:86-89), first offender in dictionary order wins
    const bool check = !(d.flags & MST_NO_RANGE_CHECK);  // forward_mix_console hands over denormalised values unchecked
    if (check && tid < np) {
        const float v = p[tid];
        if (v < 0.0f || v > 1.0f) atomicMax(a.status, 1000 - (1 + (is_master ? 52 : 0) + tid));
    }
    if (check && is_master && tid >= 32 && tid < 32 + 24) {  // fx-bus band gains/decays; "mix" is forced to 1
        const float v = a.fx_params[(int64_t)mrow * MST_NUM_FX_PARAMS + (tid - 32)];
        if (v < 0.0f || v > 1.0f) atomicMax(a.status, 1000 - (1 + 27 + (tid - 32)));
    }

    const bool eq_on = is_master ? (d.flags & MST_USE_MASTER_BUS) : (d.flags & MST_USE_TRACK_EQ);
    const bool comp_on = is_master ? (d.flags & MST_USE_MASTER_BUS) : (d.flags & MST_USE_TRACK_COMPRESSOR);
    const bool gin_on = is_master ? (d.flags & MST_USE_MASTER_BUS) : (d.flags & MST_USE_TRACK_INPUT_FADER);

    if (tid < kSections) {
        float c[5] = {1.f, 0.f, 0.f, 0.f, 0.f};
        if (eq_on) {
            const int i = eq0 + 3 * tid;
            design_section_f32(section_kind(tid), denorm(p[i], lo[i], hi[i]), denorm(p[i + 1], lo[i + 1], hi[i + 1]),
                               denorm(p[i + 2], lo[i + 2], hi[i + 2]), sr, c);
        }
        for (int j = 0; j < 5; ++j) coef[5 * tid + j] = c[j];
    } else if (tid == 64) {  // one wave per divergent fp64 branch: they run side by side
#pragma clang fp contract(off)
        float thr = 0.f, kappa = 0.f, knee = 1.f, alpha = 0.f, mk = 0.f;
        if (comp_on) {
            thr = denorm(p[cmp0], lo[cmp0], hi[cmp0]);
            const float ratio = denorm(p[cmp0 + 1], lo[cmp0 + 1], hi[cmp0 + 1]);
            const float att = denorm(p[cmp0 + 2], lo[cmp0 + 2], hi[cmp0 + 2]);
            knee = denorm(p[cmp0 + 4], lo[cmp0 + 4], hi[cmp0 + 4]);
            mk = denorm(p[cmp0 + 5], lo[cmp0 + 5], hi[cmp0 + 5]);
            kappa = (1.0f / ratio) - 1.0f;
            const float nat = sr * (att / 1000.0f);
            alpha = (float)exp((double)(-2.1972245773362196f / nat));
        }
        rc[RC_THR] = thr;
        rc[RC_KAPPA] = kappa;
        rc[RC_KNEE] = knee;
        rc[RC_ALPHA] = alpha;
        rc[RC_MAKEUP] = mk;
        rc[RC_ALPHA_C] = (float)pow((double)alpha, (double)kCompChunk);
        rc[RC_LOG2A_C] = alpha > 0.0f ? (float)((double)kCompChunk * log2((double)alpha)) : -1.0e30f;
    } else if (tid == 128) {
#pragma clang fp contract(off)
        float gin = 1.0f;
        const int gi = is_master ? 25 : 0;
        if (gin_on) gin = (float)exp2((double)(denorm(p[gi], lo[gi], hi[gi]) / 20.0f) * 3.321928094887362);
        rc[RC_GIN] = gin;
        coef[30] = gin;
        if (is_master) {
            float gout = 1.0f;
            if (d.flags & MST_USE_OUTPUT_FADER) gout = (float)exp2((double)(denorm(p[24], lo[24], hi[24]) / 20.0f) * 3.321928094887362);
            rc[RC_PANL] = gout;
            rc[RC_PANR] = gout;
        } else {
            const float half_pi = 1.5707963267948966f, two_over_pi = 0.6366197723675814f;
            const float theta = denorm(p[25], lo[25], hi[25]) * half_pi;
            rc[RC_PANL] = sqrtf(((half_pi - theta) * two_over_pi) * (float)cos((double)theta));
            rc[RC_PANR] = sqrtf((theta * two_over_pi) * (float)sin((double)theta));
            // fx send gain 10^(send_db / 20) (reference stereo_bus, mst/modules.py:276)
            rc[RC_SEND] = (d.flags & MST_USE_FX_BUS) ? (float)exp2((double)(denorm(p[26], lo[26], hi[26]) / 20.0f) * 3.321928094887362) : 0.0f;
        }
    } else if (tid == 192 && is_master && a.rc_fx) {
        // reverberation: band gains and decay rates 10 d + 1 (dasp noise_shaped_reverberation)
        const float* fp = a.fx_params + (int64_t)mrow * MST_NUM_FX_PARAMS;
        float* o = a.rc_fx + (int64_t)mrow * 24;
        // wet/dry mix: AdvancedMixConsole.forward forces it to 1 (mst/modules.py:420); forward_mix_console applies the value it is
        // given (MST_NO_RANGE_CHECK).  mix * wet = fx_in * (mix * ir): the mix rides on the band gains, the dry share
        // (1 - mix) * fx_in is added where the wet block is written (k_fx_ifft<FX_OUT>).
        const float mix = check ? 1.0f : denorm(fp[24], d.fx_lo[24], d.fx_hi[24]);
        a.fx_mix[mrow] = mix;
        for (int k = 0; k < 12; ++k) {
            o[k] = denorm(fp[k], d.fx_lo[k], d.fx_hi[k]) * mix;
            o[12 + k] = denorm(fp[12 + k], d.fx_lo[12 + k], d.fx_hi[12 + k]) * 10.0f + 1.0f;
        }
    }
    __syncthreads();
    if (tid < 30) {
        float v = coef[tid];
        if (tid < 3) v *= coef[30];  // fold the input fader (this row's gin, parked in LDS by lane 128) into section 0's numerator
        rc[RC_SOS + tid] = v;
        coef[tid] = v;  // the table chains below read the folded coefficients from LDS, not back from HBM
    }
    __syncthreads();
    if (tid < kSections) {  // all-pole constants of section tid (fp32 divisions, as the kernels did them)
        const float b0 = coef[5 * tid];
        rc[RC_AP + 3 * tid] = coef[5 * tid + 1] / b0;
        rc[RC_AP + 3 * tid + 1] = coef[5 * tid + 2] / b0;
        rc[RC_AP + 3 * tid + 2] = 1.0f / b0;
    }

#if MST_PREP_STOP == 1
    return;
#endif
    // ---- one-sample transition matrices of the forward and the adjoint cascade (zero input)
    double c64[30];
    for (int i = 0; i < 30; ++i) c64[i] = (double)coef[i];
    const int part = blockIdx.y;
    if (part == 0) {
    if (tid < 24) {
        const int which = tid / 12, col = tid % 12;
        double st[12];
        for (int i = 0; i < 12; ++i) st[i] = (i == col) ? 1.0 : 0.0;
        if (which == 0) cascade_step<double>(0.0, c64, st);
        else cascade_adj_step<double>(0.0, c64, st);
        for (int i = 0; i < 12; ++i) mats[which][0][i * 12 + col] = st[i];
    }
    if (tid >= 32 && tid < 34) {  // b = the state one unit input sample leaves behind
        const int which = tid - 32;
        double st[12];
        for (int i = 0; i < 12; ++i) st[i] = 0.0;
        if (which == 0) cascade_step<double>(1.0, c64, st);
        else cascade_adj_step<double>(1.0, c64, st);
        for (int i = 0; i < 12; ++i) wv[which][0][i] = st[i];
    }
    __syncthreads();

    const int grp = tid / 144, e = tid % 144;  // lanes 0..287 own one matrix element each
    const bool mat_lane = tid < 288;
    float* powF = (is_master ? a.powF_m + (int64_t)mrow * kPow * 144 : a.powF_t + (int64_t)row * kPow * 144);
    float* powA = (is_master ? a.powA_m + (int64_t)mrow * kPow * 144 : a.powA_t + (int64_t)row * kPow * 144);
    float* pw = grp == 0 ? powF : powA;
    // cur = A^(kEqChunk) by log2(kEqChunk) squarings
    int cur = 0;
    for (int s = 1; s < kEqChunk; s <<= 1) {
        // zero-state map by doubling: cur = A^s, so A^(t+s) b = cur (A^t b) for every t < s already made
        if (a.eq1) {
            for (int item = tid; item < 2 * s * 12; item += 320) {
                const int g2 = item / (s * 12), t = (item / 12) % s, dd = item % 12;
                double acc = 0.0;
#pragma unroll
                for (int q = 0; q < 12; ++q) acc = fma(mats[g2][cur][dd * 12 + q], wv[g2][t][q], acc);
                wv[g2][t + s][dd] = acc;
            }
        }
        if (mat_lane) mat12_mul(mats[grp][cur], mats[grp][cur], mats[grp][cur ^ 1], e);
        __syncthreads();
        cur ^= 1;
    }
    if (mat_lane) pw[0 * 144 + e] = (float)mats[grp][cur][e];
#if MST_PREP_STOP == 2
    return;
#endif
    if (a.eq1) {
        // forward: an impulse at sample j of the chunk is 63 - j steps from its end; adjoint (reverse time): j steps
        float* wzF = is_master ? a.wzF_m + (int64_t)mrow * kWz : a.wzF_t + (int64_t)row * kWz;
        float* wzA = is_master ? a.wzA_m + (int64_t)mrow * kWz : a.wzA_t + (int64_t)row * kWz;
        for (int item = tid; item < 2 * kWz; item += 320) {
            const int g2 = item / kWz, j = (item % kWz) / 16, dd = item % 16;
            const float v = dd < 12 ? (float)wv[g2][g2 == 0 ? kEqChunk - 1 - j : j][dd] : 0.0f;
            (g2 == 0 ? wzF : wzA)[j * 16 + dd] = v;
        }
    }
    if (a.eq1) {
        // in-wave scans: M^(2^j), j = 0..kPow1-1 (M = one-chunk transition), by repeated squaring; what is kept of each power is
        // its six diagonal 2x2 blocks (the diagonal blocks of a power of a block-triangular matrix are the powers of its diagonal
        // blocks) and, of M and M^64 themselves, the 15 blocks below the diagonal (layout: mst_mat.h, two sets of kTriFloats)
        float* pw1 = grp == 0 ? (is_master ? a.pow1F_m + (int64_t)mrow * kTri2 : a.pow1F_t + (int64_t)row * kTri2)
                              : (is_master ? a.pow1A_m + (int64_t)mrow * kTri2 : a.pow1A_t + (int64_t)row * kTri2);
        const int ei = e / 12, ec = e % 12, bk = ei >> 1, bj = ec >> 1, sub = (ei & 1) * 2 + (ec & 1);
        constexpr int kPow1Full = 6;  // last power formed as a full matrix: M^(2^6) = the tile transition
        for (int j = 0; j < kPow1; ++j) {
            if (mat_lane) {
                float* set = pw1 + (j / 6) * (kTri2 / 2);
                const float val = (float)mats[grp][cur][e];
                if (bk == bj) set[(6 * bk + j % 6) * 4 + sub] = val;
                else if (j % 6 == 0 && bj < bk) set[144 + (bk * (bk - 1) / 2 + bj) * 4 + sub] = val;
            }
            if (j % 6 == 0 && tid >= 288 && tid < 300) {  // park the diagonal blocks of M (j = 0) and M^64 (j = 6) for the Dp tables
                const int g2 = (tid - 288) / 6, k = (tid - 288) % 6;
                const double* Mm = mats[g2][cur];
                double* o = dblk[j / 6][g2][k];
                o[0] = Mm[(2 * k) * 12 + 2 * k];
                o[1] = Mm[(2 * k) * 12 + 2 * k + 1];
                o[2] = Mm[(2 * k + 1) * 12 + 2 * k];
                o[3] = Mm[(2 * k + 1) * 12 + 2 * k + 1];
            }
            // Full 12x12 squarings only up to M^64 (j = 6), whose off-diagonal blocks the tile-level scan needs; of the five powers
            // beyond it only the diagonal blocks are kept, and those are powers of M^64's own 2x2 diagonal blocks: formed below, by
            // lanes that square a 2x2 block, instead of five more 12x12 products with their workgroup barriers (round 5: -2.5 us)
            if (j == kPow1Full) break;
            if (mat_lane) mat12_mul(mats[grp][cur], mats[grp][cur], mats[grp][cur ^ 1], e);
            __syncthreads();
            cur ^= 1;
        }
#if MST_PREP_STOP == 3
        return;
#endif
        __syncthreads();  // the parked diagonal blocks of M^64
        for (int item = tid; item < 12 * (kPow1 - 1 - kPow1Full); item += 320) {
            const int nj = kPow1 - 1 - kPow1Full, blk = item / nj, jj = item % nj, g2 = blk / 6, k = blk % 6;
            const double* d = dblk[1][g2][k];
            double b0 = d[0], b1 = d[1], b2 = d[2], b3 = d[3];
            for (int sq = 0; sq <= jj; ++sq) {  // the products of the 12x12 chain in its order: the k-th column pair is all that is non-zero
                const double n0 = fma(b1, b2, b0 * b0), n1 = fma(b1, b3, b0 * b1), n2 = fma(b3, b2, b2 * b0), n3 = fma(b3, b3, b2 * b1);
                b0 = n0; b1 = n1; b2 = n2; b3 = n3;
            }
            float* base = g2 == 0 ? (is_master ? a.pow1F_m + (int64_t)mrow * kTri2 : a.pow1F_t + (int64_t)row * kTri2)
                                  : (is_master ? a.pow1A_m + (int64_t)mrow * kTri2 : a.pow1A_t + (int64_t)row * kTri2);
            const int j = kPow1Full + 1 + jj;
            *reinterpret_cast<float4*>(base + (j / 6) * (kTri2 / 2) + (6 * k + j % 6) * 4) = make_float4((float)b0, (float)b1, (float)b2, (float)b3);
        }
        // Dp[k][q] = D_k^(q+1), q < 16 (the in-row fix-up of the scans): 24 (set, cascade, section) blocks x 16 powers, one
        // (block, power) per lane and pass, by binary exponentiation in fp64 (at most 7 products of 2x2 matrices)
        __syncthreads();
        for (int item = tid; item < 24 * 16; item += 320) {
            const int blk = item >> 4, q = item & 15, set = blk / 12, g2 = (blk / 6) & 1, k = blk % 6;
            const double* d = dblk[set][g2][k];
            double b0 = d[0], b1 = d[1], b2 = d[2], b3 = d[3];      // running square
            double r0 = 1.0, r1 = 0.0, r2 = 0.0, r3 = 1.0;          // result
            for (int ebit = q + 1; ebit > 0; ebit >>= 1) {
                if (ebit & 1) {
                    const double n0 = fma(r0, b0, r1 * b2), n1 = fma(r0, b1, r1 * b3), n2 = fma(r2, b0, r3 * b2), n3 = fma(r2, b1, r3 * b3);
                    r0 = n0; r1 = n1; r2 = n2; r3 = n3;
                }
                const double s0_ = fma(b0, b0, b1 * b2), s1_ = fma(b0, b1, b1 * b3), s2_ = fma(b2, b0, b3 * b2), s3_ = fma(b2, b1, b3 * b3);
                b0 = s0_; b1 = s1_; b2 = s2_; b3 = s3_;
            }
            float* base = g2 == 0 ? (is_master ? a.pow1F_m + (int64_t)mrow * kTri2 : a.pow1F_t + (int64_t)row * kTri2)
                                  : (is_master ? a.pow1A_m + (int64_t)mrow * kTri2 : a.pow1A_t + (int64_t)row * kTri2);
            *reinterpret_cast<float4*>(base + set * (kTri2 / 2) + 208 + (16 * k + q) * 4) = make_float4((float)r0, (float)r1, (float)r2, (float)r3);
        }
    } else {
    // acc (slot 2) = cur^KE by binary exponentiation; `cur`/`cur^1` ping-pong the running square
    if (mat_lane) mats[grp][2][e] = (e / 12 == e % 12) ? 1.0 : 0.0;
    __syncthreads();
    for (int k = a.KE; k > 0; k >>= 1) {
        if (k & 1) {
            double tmpv = 0.0;
            if (mat_lane) {
                const int i = e / 12, j = e % 12;
                for (int q = 0; q < 12; ++q) tmpv = fma(mats[grp][2][i * 12 + q], mats[grp][cur][q * 12 + j], tmpv);
            }
            __syncthreads();
            if (mat_lane) mats[grp][2][e] = tmpv;
            __syncthreads();
        }
        if (k > 1) {
            if (mat_lane) mat12_mul(mats[grp][cur], mats[grp][cur], mats[grp][cur ^ 1], e);
            __syncthreads();
            cur ^= 1;
        }
    }
    // tables 1..kScanLevels: (M^KE)^(2^j)
    int src = 2;  // accumulator slot holds M^KE
    for (int j = 0; j < kScanLevels; ++j) {
        if (mat_lane) pw[(1 + j) * 144 + e] = (float)mats[grp][src][e];
        if (j + 1 < kScanLevels) {
            const int dst = (src == 2) ? 0 : (src == 0 ? 1 : 0);
            if (mat_lane) mat12_mul(mats[grp][src], mats[grp][src], mats[grp][dst], e);
            __syncthreads();
            src = dst;
        }
    }
    }

    }  // part 0

    // ---- all-pole filters used by the coefficient-gradient pass: f = 2k (1/A_k), 2k+1 (1/B_k)
    if (part == 1 && tid < 12) {
        const int f = tid, k = f >> 1;
        double c1, c2;
        if (f & 1) {
            c1 = c64[5 * k + 1] / c64[5 * k + 0];
            c2 = c64[5 * k + 2] / c64[5 * k + 0];
        } else {
            c1 = c64[5 * k + 3];
            c2 = c64[5 * k + 4];
        }
        double m[4] = {-c1, -c2, 1.0, 0.0}, t[4];
        auto mul = [](const double* x, const double* y, double* o) {
            o[0] = fma(x[0], y[0], x[1] * y[2]);
            o[1] = fma(x[0], y[1], x[1] * y[3]);
            o[2] = fma(x[2], y[0], x[3] * y[2]);
            o[3] = fma(x[2], y[1], x[3] * y[3]);
        };
        for (int s = 1; s < kEqChunk; s <<= 1) {
            mul(m, m, t);
            for (int i = 0; i < 4; ++i) m[i] = t[i];
        }
        float* pp = (is_master ? a.powP_m + ((int64_t)mrow * 12 + f) * kPow * 4 : a.powP_t + ((int64_t)row * 12 + f) * kPow * 4);
        for (int i = 0; i < 4; ++i) pp[i] = (float)m[i];
        double acc[4] = {1, 0, 0, 1}, base[4] = {m[0], m[1], m[2], m[3]};
        for (int k2 = a.KE; k2 > 0; k2 >>= 1) {
            if (k2 & 1) {
                mul(acc, base, t);
                for (int i = 0; i < 4; ++i) acc[i] = t[i];
            }
            mul(base, base, t);
            for (int i = 0; i < 4; ++i) base[i] = t[i];
        }
        for (int j = 0; j < kScanLevels; ++j) {
            for (int i = 0; i < 4; ++i) pp[(1 + j) * 4 + i] = (float)acc[i];
            mul(acc, acc, t);
            for (int i = 0; i < 4; ++i) acc[i] = t[i];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// backward of the parameter maps.  One workgroup (4 waves) per filter row.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_prep_bwd(PrepBwdArgs a) {
    __shared__ double dsos[EP_COUNT];
    __shared__ double dcp[CP_COUNT];
    __shared__ double lanesum[64][EP_COUNT + CP_COUNT + 1];  // per-lane fp64 sums of wave 0 (odd row length: conflict-free columns)
    __shared__ double Jd[kSections][5][3];                   // design Jacobians d{b0 b1 b2 a1 a2} / d{gain, freq, q}
    const int row = blockIdx.x, tid = threadIdx.x;
    // every compressor launch of this backward has run: re-arm its granules, so that a second backward over the same forward works
    for (int64_t i = (int64_t)blockIdx.x * 256 + tid; i < a.gran_n; i += (int64_t)gridDim.x * 256) a.gran[i] = 0ull;
    const bool is_master = row >= a.R;
    const int mrow = row - a.R;
    const mst_console_desc& d = a.d;
    const double sr = d.sample_rate;
    const float* p = is_master ? a.master_params + (int64_t)mrow * MST_NUM_MASTER_PARAMS
                               : a.track_params + (int64_t)row * MST_NUM_TRACK_PARAMS;
    const float* lo = is_master ? d.master_lo : d.track_lo;
    const float* hi = is_master ? d.master_hi : d.track_hi;
    const float* rc = is_master ? a.rc_m + (int64_t)mrow * RC_STRIDE : a.rc_t + (int64_t)row * RC_STRIDE;
    float* g = is_master ? a.grad_master_params + (int64_t)mrow * MST_NUM_MASTER_PARAMS
                         : a.grad_track_params + (int64_t)row * MST_NUM_TRACK_PARAMS;
    const int np = is_master ? MST_NUM_MASTER_PARAMS : MST_NUM_TRACK_PARAMS;
    const int eq0 = is_master ? 0 : 1, cmp0 = is_master ? 18 : 19;
    const bool eq_on = is_master ? (d.flags & MST_USE_MASTER_BUS) : (d.flags & MST_USE_TRACK_EQ);
    const bool comp_on = is_master ? (d.flags & MST_USE_MASTER_BUS) : (d.flags & MST_USE_TRACK_COMPRESSOR);
    const bool gin_on = is_master ? (d.flags & MST_USE_MASTER_BUS) : (d.flags & MST_USE_TRACK_INPUT_FADER);
    const bool chain_on = is_master ? (d.flags & MST_USE_MASTER_BUS) : true;  // does the EQ/gain stage exist

    // deterministic reduction of the partial sums.  Wave 0: lane l takes workgroup-partials l, l+64, ... as fp64, every load of a
    // lane issued before anything is summed (one memory round trip), and parks its 38 sums in LDS; then lane q < 38 adds the 64
    // lane values of sum q in lane order (a fp64 shuffle tree here was 456 ds_bpermute).  Wave 1 meanwhile runs the fp64
    // dual-number biquad design (exp2, sincos) whose Jacobian the chain rule below needs - it used to wait for the reduction.
    if (tid < 64) {
        const int nsig = is_master ? 2 : 1;
        const float* ep = is_master ? a.ep_m + ((int64_t)(mrow * 2) * a.nblkEt) * EP_COUNT
                                    : a.ep_t + ((int64_t)row * a.nblkEt) * EP_COUNT;
        const int nE = chain_on ? nsig * a.nblkEt : 0;  // the two channels of a master row are adjacent
        double se[EP_COUNT];
#pragma unroll
        for (int q = 0; q < EP_COUNT; ++q) se[q] = 0.0;
        for (int b = tid; b < nE; b += 64) {
            const float* pb = ep + (int64_t)b * EP_COUNT;  // 30 floats, 8-byte aligned
            float2 v[EP_COUNT / 2];
#pragma unroll
            for (int q = 0; q < EP_COUNT / 2; ++q) v[q] = *reinterpret_cast<const float2*>(pb + 2 * q);
#pragma unroll
            for (int q = 0; q < EP_COUNT / 2; ++q) {
                se[2 * q] += (double)v[q].x;
                se[2 * q + 1] += (double)v[q].y;
            }
        }
        const float* cp = is_master ? a.cp_m + ((int64_t)mrow * a.nblkC) * CP_COUNT : a.cp_t + ((int64_t)row * a.nblkC) * CP_COUNT;
        double sc[CP_COUNT];
#pragma unroll
        for (int q = 0; q < CP_COUNT; ++q) sc[q] = 0.0;
        for (int b = tid; b < a.nblkC; b += 64) {
            const float4 v0 = *reinterpret_cast<const float4*>(cp + (int64_t)b * CP_COUNT);
            const float4 v1 = *reinterpret_cast<const float4*>(cp + (int64_t)b * CP_COUNT + 4);
            sc[0] += (double)v0.x; sc[1] += (double)v0.y; sc[2] += (double)v0.z; sc[3] += (double)v0.w;
            sc[4] += (double)v1.x; sc[5] += (double)v1.y; sc[6] += (double)v1.z; sc[7] += (double)v1.w;
        }
#pragma unroll
        for (int q = 0; q < EP_COUNT; ++q) lanesum[tid][q] = se[q];
#pragma unroll
        for (int q = 0; q < CP_COUNT; ++q) lanesum[tid][EP_COUNT + q] = sc[q];
    } else if (tid < 64 + kSections) {
        const int k = tid - 64, i = eq0 + 3 * k;
        if (eq_on) {
            D3 J[5];
            design_section_dual(section_kind(k), (double)denorm(p[i], lo[i], hi[i]), (double)denorm(p[i + 1], lo[i + 1], hi[i + 1]),
                                (double)denorm(p[i + 2], lo[i + 2], hi[i + 2]), sr, J);
            for (int j = 0; j < 5; ++j)
                for (int v = 0; v < 3; ++v) Jd[k][j][v] = J[j].d[v];
        }
    }
    if (tid < np) g[tid] = 0.0f;
    __syncthreads();
    if (tid < EP_COUNT + CP_COUNT) {
        double s = 0.0;
        for (int l = 0; l < 64; ++l) s += lanesum[l][tid];
        if (tid < EP_COUNT) dsos[tid] = s;
        else dcp[tid - EP_COUNT] = s;
    }
    __syncthreads();

    const double gin = rc[RC_GIN];
    if (tid < kSections) {
        if (eq_on) {
            const int k = tid, i = eq0 + 3 * k;
            const double bs0 = (k == 0) ? gin : 1.0;  // section 0's numerator was scaled by the fader
            for (int v = 0; v < 3; ++v) {
                double acc = 0.0;
                for (int j = 0; j < 5; ++j) acc += dsos[5 * k + j] * Jd[k][j][v] * (j < 3 ? bs0 : 1.0);
                g[i + v] = (float)(acc * (double)(hi[i + v] - lo[i + v]));
            }
        }
    } else if (tid == 64) {  // one wave per divergent fp64 branch
        if (gin_on) {
            // d/d gin: section-0 numerator b' = gin*b  =>  sum_j dL/db'_j * b_j ;  b_j = b'_j / gin
            double acc = 0.0;
            for (int j = 0; j < 3; ++j) acc += dsos[j] * (double)rc[RC_SOS + j];
            const int gi = is_master ? 25 : 0;
            // acc = gin * dL/dgin ; d gin / d gain_db = gin*ln10/20
            g[gi] = (float)(acc * (double)kLn10Over20 * (double)(hi[gi] - lo[gi]));
        }
    } else if (tid == 128) {
        if (comp_on) {
            const double ratio = (double)denorm(p[cmp0 + 1], lo[cmp0 + 1], hi[cmp0 + 1]);
            const double att = (double)denorm(p[cmp0 + 2], lo[cmp0 + 2], hi[cmp0 + 2]);
            const double alpha = rc[RC_ALPHA];
            g[cmp0 + 0] = (float)(dcp[CP_THR] * (double)(hi[cmp0] - lo[cmp0]));
            g[cmp0 + 1] = (float)(dcp[CP_KAPPA] * (-1.0 / (ratio * ratio)) * (double)(hi[cmp0 + 1] - lo[cmp0 + 1]));
            g[cmp0 + 2] = (float)(dcp[CP_ALPHA] * alpha * 2.1972245773362196 * 1000.0 / (sr * att * att) *
                                  (double)(hi[cmp0 + 2] - lo[cmp0 + 2]));
            g[cmp0 + 3] = 0.0f;  // release_ms is accepted but unused by the reference op
            g[cmp0 + 4] = (float)(dcp[CP_KNEE] * (double)(hi[cmp0 + 4] - lo[cmp0 + 4]));
            g[cmp0 + 5] = (float)(dcp[CP_MAKEUP] * (double)(hi[cmp0 + 5] - lo[cmp0 + 5]));
        }
    } else if (tid == 192) {
        if (is_master) {
            if (d.flags & MST_USE_OUTPUT_FADER) {
                // CP_PANL holds sum(grad_out * out_before_fader); d gout/d db = gout*ln10/20
                g[24] = (float)(dcp[CP_PANL] * (double)rc[RC_PANL] * (double)kLn10Over20 * (double)(hi[24] - lo[24]));
            }
        } else {
            const double half_pi = 1.5707963267948966, two_over_pi = 0.6366197723675814;
            const double theta = (double)denorm(p[25], lo[25], hi[25]) * half_pi;
            const double L = rc[RC_PANL], Rr = rc[RC_PANR];
            double dLdth = 0.0, dRdth = 0.0;
            if (L > 0.0) dLdth = (-two_over_pi * cos(theta) - (half_pi - theta) * two_over_pi * sin(theta)) / (2.0 * L);
            if (Rr > 0.0) dRdth = (two_over_pi * sin(theta) + theta * two_over_pi * cos(theta)) / (2.0 * Rr);
            g[25] = (float)((dcp[CP_PANL] * dLdth + dcp[CP_PANR] * dRdth) * half_pi * (double)(hi[25] - lo[25]));
            // fx send: d send / d send_db = send ln10 / 20
            if (d.flags & MST_USE_FX_BUS)
                g[26] = (float)(dcp[CP_SEND] * (double)rc[RC_SEND] * (double)kLn10Over20 * (double)(hi[26] - lo[26]));
        }
    }
    // reverberation parameters: fixed-order reduction of k_fx_ir_bwd's partial sums (wave 1 of the master rows)
    if (is_master && a.grad_fx_params && tid >= 64 && tid < 64 + MST_NUM_FX_PARAMS) {
        const int k = tid - 64;
        float* gf = a.grad_fx_params + (int64_t)mrow * MST_NUM_FX_PARAMS;
        const double mix = (double)a.fx_mix[mrow];
        if (k == 24) {
            // forward(): "mix" is forced to 1 by the reference (mst/modules.py:420), no gradient reaches it.  forward_mix_console:
            // y = (1 - mix) fx_in + mix wet  =>  dL/dmix = <dbus, wet> - <dbus, fx_in> = sum_k gain_k dL/d(mix gain_k) - dry sums
            double acc = 0.0;
            if (d.flags & MST_NO_RANGE_CHECK) {
                for (int j = 0; j < 12; ++j) {
                    double aj = 0.0;
                    for (int bk = 0; bk < a.nblkF; ++bk) aj += (double)a.fx_part[((int64_t)mrow * a.nblkF + bk) * 24 + j];
                    acc += aj * (double)denorm(a.fx_params[(int64_t)mrow * MST_NUM_FX_PARAMS + j], d.fx_lo[j], d.fx_hi[j]);
                }
                for (int bk = 0; bk < a.nblkX; ++bk) acc -= (double)a.fx_dry[(int64_t)mrow * a.nblkX + bk];
                acc *= (double)(d.fx_hi[24] - d.fx_lo[24]);
            }
            gf[k] = (float)acc;
        } else {
            double acc = 0.0;
            for (int bk = 0; bk < a.nblkF; ++bk) acc += (double)a.fx_part[((int64_t)mrow * a.nblkF + bk) * 24 + k];
            const double scale = (double)(d.fx_hi[k] - d.fx_lo[k]) * (k >= 12 ? 10.0 : mix);  // rate = 10 decay + 1; gains carry the mix
            gf[k] = (float)(acc * scale);
        }
    }
}

// =====================================================================================================================
// BASELINE cfg #1: gain + pan + bus sum only (BasicMixConsole; contract inferred from reference mst/mixing.py:122-164,
// :935-945: two parameters per track).  The general path would run k_prep's table chains, two EQ launches over identity
// sections, the apply kernel and the whole backward chain (0.28 ms per step at 2 x 4 x 65536); here the call is ONE forward
// launch and TWO backward launches with the same arithmetic per sample: y = gin x (the identity cascade's only product),
// bus_c = sum_t fma(pan_c, y, bus_c) in track order.  Built in this file for its contraction-free denormalisation.
struct BasicConst { float gin, pl, pr; };
__device__ __forceinline__ BasicConst basic_consts(const float* p, const mst_console_desc& d) {
#pragma clang fp contract(off)
    BasicConst k;
    k.gin = 1.0f;
    if (d.flags & MST_USE_TRACK_INPUT_FADER) k.gin = (float)exp2((double)(denorm(p[0], d.track_lo[0], d.track_hi[0]) / 20.0f) * 3.321928094887362);
    const float half_pi = 1.5707963267948966f, two_over_pi = 0.6366197723675814f;
    const float theta = denorm(p[25], d.track_lo[25], d.track_hi[25]) * half_pi;
    k.pl = sqrtf(((half_pi - theta) * two_over_pi) * (float)cos((double)theta));
    k.pr = sqrtf((theta * two_over_pi) * (float)sin((double)theta));
    return k;
}
constexpr int kBasicSpan = 8 * 256;  // samples per workgroup: 8 per lane, like the compressor kernels
__global__ __launch_bounds__(256) void k_basic_fwd(BasicArgs a) {
    __shared__ BasicConst kc[kBasicMaxTracks];
    const int b = blockIdx.y, tid = threadIdx.x, T = a.d.n_tracks;
    const float* tp = a.track_params + (int64_t)b * T * MST_NUM_TRACK_PARAMS;
    if (tid < T) kc[tid] = basic_consts(tp + (int64_t)tid * MST_NUM_TRACK_PARAMS, a.d);
    if (blockIdx.x == 0 && !(a.d.flags & MST_NO_RANGE_CHECK)) {  // reference mst/modules.py:86-89, same status codes as k_prep
        for (int i = tid; i < T * MST_NUM_TRACK_PARAMS; i += 256) {
            const float v = tp[i];
            if (v < 0.0f || v > 1.0f) atomicMax(a.status, 1000 - (1 + i % MST_NUM_TRACK_PARAMS));
        }
        if (tid < MST_NUM_FX_PARAMS - 1) {
            const float v = a.fx_params[(int64_t)b * MST_NUM_FX_PARAMS + tid];
            if (v < 0.0f || v > 1.0f) atomicMax(a.status, 1000 - (1 + 27 + tid));
        }
        if (tid >= 64 && tid < 64 + MST_NUM_MASTER_PARAMS) {
            const float v = a.master_params[(int64_t)b * MST_NUM_MASTER_PARAMS + tid - 64];
            if (v < 0.0f || v > 1.0f) atomicMax(a.status, 1000 - (1 + 52 + tid - 64));
        }
    }
    __syncthreads();
    const int64_t i0 = (int64_t)blockIdx.x * kBasicSpan + tid * 8, n = a.d.n_samples;
    float4 L[2] = {make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0)}, R[2] = {L[0], L[0]};
    for (int t = 0; t < T; ++t) {
        const float* row = a.tracks + ((int64_t)b * T + t) * a.d.track_row_stride;
        const BasicConst k = kc[t];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float4 x = load4(row, i0 + 4 * h, n);
            const float4 y = make_float4(k.gin * x.x, k.gin * x.y, k.gin * x.z, k.gin * x.w);
            L[h] = make_float4(fmaf(k.pl, y.x, L[h].x), fmaf(k.pl, y.y, L[h].y), fmaf(k.pl, y.z, L[h].z), fmaf(k.pl, y.w, L[h].w));
            R[h] = make_float4(fmaf(k.pr, y.x, R[h].x), fmaf(k.pr, y.y, R[h].y), fmaf(k.pr, y.z, R[h].z), fmaf(k.pr, y.w, R[h].w));
            if (a.mixed) {
                store4(a.mixed + (((int64_t)b * 2 + 0) * T + t) * n, i0 + 4 * h, n, make_float4(k.pl * y.x, k.pl * y.y, k.pl * y.z, k.pl * y.w));
                store4(a.mixed + (((int64_t)b * 2 + 1) * T + t) * n, i0 + 4 * h, n, make_float4(k.pr * y.x, k.pr * y.y, k.pr * y.z, k.pr * y.w));
            }
        }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        store4(a.mix + ((int64_t)b * 2 + 0) * n, i0 + 4 * h, n, L[h]);
        store4(a.mix + ((int64_t)b * 2 + 1) * n, i0 + 4 * h, n, R[h]);
    }
}
// backward, stage 1: per (mix, block) and track the sums <gL, y>, <gR, y> (y = gin x); grad_tracks = gin (pl gL + pr gR) (+ the
// mixed_tracks cotangent) when asked for.  part: (bs, nblk, T, 2)
__global__ __launch_bounds__(256) void k_basic_bwd_part(BasicArgs a) {
    __shared__ BasicConst kc[kBasicMaxTracks];
    __shared__ float red[4][2];
    const int b = blockIdx.y, tid = threadIdx.x, T = a.d.n_tracks;
    if (tid < T) kc[tid] = basic_consts(a.track_params + ((int64_t)b * T + tid) * MST_NUM_TRACK_PARAMS, a.d);
    __syncthreads();
    const int64_t i0 = (int64_t)blockIdx.x * kBasicSpan + tid * 8, n = a.d.n_samples;
    float gl[8], gr[8];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const float4 l = load4(a.grad_mix + ((int64_t)b * 2 + 0) * n, i0 + 4 * h, n), r = load4(a.grad_mix + ((int64_t)b * 2 + 1) * n, i0 + 4 * h, n);
        gl[4 * h] = l.x; gl[4 * h + 1] = l.y; gl[4 * h + 2] = l.z; gl[4 * h + 3] = l.w;
        gr[4 * h] = r.x; gr[4 * h + 1] = r.y; gr[4 * h + 2] = r.z; gr[4 * h + 3] = r.w;
    }
    for (int t = 0; t < T; ++t) {
        const float* row = a.tracks + ((int64_t)b * T + t) * a.d.track_row_stride;
        const BasicConst k = kc[t];
        float sl = 0.0f, sr = 0.0f;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float4 x4 = load4(row, i0 + 4 * h, n);
            const float x[4] = {x4.x, x4.y, x4.z, x4.w};
            float ml[4] = {0, 0, 0, 0}, mr[4] = {0, 0, 0, 0};
            if (a.grad_mixed) {
                const float4 u = load4(a.grad_mixed + (((int64_t)b * 2 + 0) * T + t) * n, i0 + 4 * h, n);
                const float4 v = load4(a.grad_mixed + (((int64_t)b * 2 + 1) * T + t) * n, i0 + 4 * h, n);
                ml[0] = u.x; ml[1] = u.y; ml[2] = u.z; ml[3] = u.w;
                mr[0] = v.x; mr[1] = v.y; mr[2] = v.z; mr[3] = v.w;
            }
            float gx[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float y = k.gin * x[e], cl = gl[4 * h + e] + ml[e], cr = gr[4 * h + e] + mr[e];
                sl = fmaf(cl, y, sl);
                sr = fmaf(cr, y, sr);
                gx[e] = k.gin * fmaf(k.pr, cr, k.pl * cl);
            }
            if (a.grad_tracks) store4(a.grad_tracks + ((int64_t)b * T + t) * n, i0 + 4 * h, n, make_float4(gx[0], gx[1], gx[2], gx[3]));
        }
        sl = wave_sum(sl);
        sr = wave_sum(sr);
        if ((tid & 63) == 0) { red[tid >> 6][0] = sl; red[tid >> 6][1] = sr; }
        __syncthreads();
        if (tid < 2) a.part[(((int64_t)b * gridDim.x + blockIdx.x) * T + t) * 2 + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
        __syncthreads();
    }
}
// stage 2: one lane per track row folds the blocks in order (fp64) and applies the chain rule of k_prep_bwd (gain: d gin / d dB =
// gin ln10 / 20; constant-power pan law); every other column of the (bs, T, 27) gradient is zero, like autograd's unused leaves
__global__ void k_basic_bwd_final(BasicArgs a, int nblk) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x, T = a.d.n_tracks;
    if (r >= a.d.bs * T) return;
    const int b = r / T, t = r % T;
    double sl = 0.0, sr = 0.0;
    for (int k = 0; k < nblk; ++k) {
        sl += (double)a.part[(((int64_t)b * nblk + k) * T + t) * 2];
        sr += (double)a.part[(((int64_t)b * nblk + k) * T + t) * 2 + 1];
    }
    const float* p = a.track_params + (int64_t)r * MST_NUM_TRACK_PARAMS;
    const BasicConst k = basic_consts(p, a.d);
    float* g = a.grad_track_params + (int64_t)r * MST_NUM_TRACK_PARAMS;
    for (int i = 0; i < MST_NUM_TRACK_PARAMS; ++i) g[i] = 0.0f;
    const float* lo = a.d.track_lo;
    const float* hi = a.d.track_hi;
    if (a.d.flags & MST_USE_TRACK_INPUT_FADER)  // gin dL/dgin = sum (pl gL + pr gR) y
        g[0] = (float)(((double)k.pl * sl + (double)k.pr * sr) * (double)kLn10Over20 * (double)(hi[0] - lo[0]));
    const double half_pi = 1.5707963267948966, two_over_pi = 0.6366197723675814;
    const double theta = (double)denorm(p[25], lo[25], hi[25]) * half_pi, L = k.pl, Rr = k.pr;
    double dLdth = 0.0, dRdth = 0.0;
    if (L > 0.0) dLdth = (-two_over_pi * cos(theta) - (half_pi - theta) * two_over_pi * sin(theta)) / (2.0 * L);
    if (Rr > 0.0) dRdth = (two_over_pi * sin(theta) + theta * two_over_pi * cos(theta)) / (2.0 * Rr);
    g[25] = (float)((sl * dLdth + sr * dRdth) * half_pi * (double)(hi[25] - lo[25]));
    if (t == 0 && a.grad_master_params)
        for (int i = 0; i < MST_NUM_MASTER_PARAMS; ++i) a.grad_master_params[(int64_t)b * MST_NUM_MASTER_PARAMS + i] = 0.0f;
}
void launch_basic_forward(const BasicArgs& a, hipStream_t stream) {
    const int nblk = (int)((a.d.n_samples + kBasicSpan - 1) / kBasicSpan);
    hipLaunchKernelGGL(k_basic_fwd, dim3(nblk, a.d.bs), dim3(256), 0, stream, a);
}
void launch_basic_backward(const BasicArgs& a, hipStream_t stream) {
    const int nblk = (int)((a.d.n_samples + kBasicSpan - 1) / kBasicSpan), rows = a.d.bs * a.d.n_tracks;
    hipLaunchKernelGGL(k_basic_bwd_part, dim3(nblk, a.d.bs), dim3(256), 0, stream, a);
    hipLaunchKernelGGL(k_basic_bwd_final, dim3((rows + 63) / 64), dim3(64), 0, stream, a, nblk);
}

void launch_prep(const PrepArgs& a, hipStream_t stream) {
#ifndef MST_PREP_PREFETCH_SEGS
#define MST_PREP_PREFETCH_SEGS 12  // rider workgroups per track row (0: none); ~17 sixteen-byte loads per lane at 262144 samples
#endif
    const int segs = (a.pf_src && MST_PREP_PREFETCH_SEGS > 0) ? MST_PREP_PREFETCH_SEGS : 0;
    hipLaunchKernelGGL(k_prep, dim3(a.R + a.bs, 2 + segs), dim3(320), 0, stream, a);
}
void launch_prep_bwd(const PrepBwdArgs& a, hipStream_t stream) {
    hipLaunchKernelGGL(k_prep_bwd, dim3(a.R + a.bs), dim3(256), 0, stream, a);
}

}  // namespace mst
