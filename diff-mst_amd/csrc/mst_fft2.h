// mst_fft2.h - the register-radix FFT engine of the spectrogram loss (round 2).
//
// One frame is transformed by a group of LG lanes that each hold 8 points per sequence IN REGISTERS and run radix-8
// butterflies there; between passes the points are exchanged through ONE padded LDS buffer (Stockham autosort: natural
// order in, natural order out).  Compared with the round-1 radix-4 LDS transform (4.5-6.5 LDS round trips per frame,
// twiddles rebuilt from a two-level table inside every butterfly, 31-47 % of the LDS cycles lost to bank conflicts -
// profiles/round2_counters.md) a frame makes 3-4 round trips, the inter-pass twiddles come from three per-lane constants
// per pass fetched once per kernel from the exactly rounded table, and the layout below is conflict-free on the
// (expensive) store side.
//
//   n_fft   lanes/frame  passes
//   512     64           8 x 8 x 8
//   2048    256          8 x 8 x 8 x 4      (the radix-4 pass: two butterflies per lane)
//   8192    512          radix-2 decimation in frequency on the way in (x[i] +- x[i + 4096], the difference times
//                        W_8192^i), then TWO independent 4096-point transforms 8 x 8 x 8 x 8: the even and the odd bins
//   4096    512          8 x 8 x 8 x 8      (the half-size inverse of the 8192 backward)
//
// Pass p (radix R, Ns = product of the earlier radices) handles butterfly j: inputs j + t M/R, twiddles W_(Ns R)^(k t)
// with k = j mod Ns, outputs (j - k) R + k + t Ns.  LDS slot of element i: i + i / 8 (one pad slot per 8 elements):
//   pass-1 stores  lane j -> 9 j + t        ds_write_b64, 18 dwords between lanes: 16 lanes hit the 32 banks once
//   Ns = 8 stores  8 consecutive lanes -> 8 consecutive slots of one padded row, the next 8 lanes 8 rows = 144 = 16 (mod 32)
//                  dwords later: conflict-free as well
//   Ns >= 64 stores and all loads: lane-consecutive elements; the pad slot every 8 elements makes two banks 2-way
//                  (loads are 3x cheaper than stores on this LDS, MI355X_MICROARCH.md)
#pragma once
#include "mst_common.h"
#ifndef MST_FFT2_SWIZZLE
#define MST_FFT2_SWIZZLE 0
#endif
#ifndef MST_FFT2_PAD64
#define MST_FFT2_PAD64 0  // A/B switch: 1 = the load-conflict-free images described at FftShape::slot1 (round 5).  Measured and NOT taken: the
                          // bench kernels are unchanged (k_stft3_fwd 72.4 vs 72.4 us, k_stft2_bwd_512_2048 71.7 vs 71.7) and the engine alone
                          // (tools/ubench/wf512.hip, 4 / 8 waves per SIMD) runs 846 / 776 cycles per transform against 823 / 743 - the two-way
                          // load conflicts the counter reports are not what these kernels wait for
#endif

namespace mst {

// A/B switch MST_FFT_PK: complex values as float pairs on the packed fp32 instructions (a complex add = one v_pk_add_f32, a product =
// v_pk_mul_f32 + v_pk_fma_f32 with the swapped, half-negated operand on the op_sel / neg modifiers).  Measured and NOT taken: the
// instruction count falls 3-10 % only (the pair assembly costs a v_mov for most of what it saves) while the even-aligned pairs raise the
// register need - 8192 forward spills 49 registers at its 128 cap (33.6 -> 64.1 us), 512 / 2048 backward leave the four-waves-per-SIMD
// budget (37.2 -> 47.5, 42.7 -> 51.7 us), and even the spill-free 512 forward is slower (20.9 -> 22.4 us).
#ifndef MST_FFT_PK
#define MST_FFT_PK 0
#endif
#if MST_FFT_PK && defined(__clang__)
__device__ __forceinline__ f2 c2v(float2 a) { return f2{a.x, a.y}; }
__device__ __forceinline__ float2 v2c(f2 a) { return make_float2(a.x, a.y); }
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return v2c(c2v(a) + c2v(b)); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return v2c(c2v(a) - c2v(b)); }
__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return v2c(f2_fma(f2{a.y, a.y}, f2{-b.y, b.x}, f2{a.x, a.x} * c2v(b)));
}
__device__ __forceinline__ float2 cscale(float c, float2 a) { return v2c(f2{c, c} * c2v(a)); }
#else
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ float2 cscale(float c, float2 a) { return make_float2(c * a.x, c * a.y); }
#endif
__device__ __forceinline__ float2 cconj(float2 a) { return make_float2(a.x, -a.y); }
__device__ __forceinline__ float2 mul_mi(float2 a) { return make_float2(a.y, -a.x); }  // a * (-i)
__device__ __forceinline__ float2 mul_pi(float2 a) { return make_float2(-a.y, a.x); }  // a * (+i)

// ---- forward (e^{-i...}) butterflies, natural order in place ---------------------------------------
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
__device__ __forceinline__ void butterfly<4>(float2* v) { fft4(v[0], v[1], v[2], v[3]); }
template <>
__device__ __forceinline__ void butterfly<8>(float2* v) {
    constexpr float c = 0.70710678118654752f;
    float2 e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6], o0 = v[1], o1 = v[3], o2 = v[5], o3 = v[7];
    fft4(e0, e1, e2, e3);
    fft4(o0, o1, o2, o3);
    o1 = cscale(c, cadd(o1, mul_mi(o1)));   // * W8^1 = (c, -c):  c (x + y, y - x)
    o2 = mul_mi(o2);                        // * W8^2 = -i
    o3 = cscale(c, csub(mul_mi(o3), o3));   // * W8^3 = (-c, -c):  c (y - x, -(x + y))
    v[0] = cadd(e0, o0); v[4] = csub(e0, o0);
    v[1] = cadd(e1, o1); v[5] = csub(e1, o1);
    v[2] = cadd(e2, o2); v[6] = csub(e2, o2);
    v[3] = cadd(e3, o3); v[7] = csub(e3, o3);
}

constexpr int ilog2c(int n) { return n <= 1 ? 0 : 1 + ilog2c(n >> 1); }

// M = length the passes run on, NSEQ sequences of it per frame, LG lanes per frame, NP passes of radix 8 (the last one
// of radix RL <= 8)
template <int N> struct FftPlan;
template <> struct FftPlan<512>  { static constexpr int M = 512,  NSEQ = 1, LG = 64,  NP = 3, RL = 8; };
template <> struct FftPlan<2048> { static constexpr int M = 2048, NSEQ = 1, LG = 256, NP = 4, RL = 4; };
template <> struct FftPlan<4096> { static constexpr int M = 4096, NSEQ = 1, LG = 512, NP = 4, RL = 8; };
template <> struct FftPlan<8192> { static constexpr int M = 4096, NSEQ = 2, LG = 512, NP = 4, RL = 8; };

template <int N>
struct FftShape {
    using P = FftPlan<N>;
    static constexpr int M = P::M, NSEQ = P::NSEQ, LG = P::LG, NP = P::NP, RL = P::RL;
    static constexpr int NBL = (M / RL) / LG;      // butterflies per lane in the last pass (1, or 2 for the radix-4 pass)
    // two slot maps (measured on MI355X, both conflict-free on the store side): padding i + i/8 costs no address arithmetic
    // (offsets fold into the ds instructions) and wins for the small transforms; the XOR swizzle makes the loads conflict-free
    // too and needs no padding - it wins for the 4096-point sequences (8192 forward 40.5 -> 37.3 us, backward 77 -> 72 us) and
    // loses below (2048 forward 27.9 -> 32.7 us, 512 backward 48.9 -> 58.4 us)
    static constexpr bool SWZ = MST_FFT2_SWIZZLE ? true : (M >= 4096);
    static constexpr bool PAD64 = MST_FFT2_PAD64;
    static constexpr int SLOTS = SWZ ? M : M + M / 8;
    static constexpr int TWSCALE = N / M;          // W_M^e = W_N^(TWSCALE e)
    static_assert(M / 8 == LG, "one radix-8 butterfly per lane and sequence in every pass but the last");
    // swizzle: element bits [2:0] ^= bits [6:4], bit 3 ^= bit 6 - rows of 8 elements are permuted by (row >> 1) & 7 and pairs
    // of rows swap halves with row bit 3, so that every access pattern of the passes (a column of 16 rows, 8 + 8 lanes into
    // rows 8 apart, 16 / 32 consecutive elements) touches each bank once
    __device__ static __forceinline__ constexpr int slot(int i) {
        return SWZ ? (i ^ (((i >> 4) & 7) | ((i >> 3) & 8))) : (PAD64 ? i + ((i >> 6) << 3) : i + (i >> 3));
    }
    // Round 5 (PAD64, the padded maps only): the pad sits behind every 64 elements (8 slots) instead of behind every 8 (1 slot).  With
    // one pad slot per 8 elements a run of 32 lane-consecutive elements - every LOAD of the passes, every bin read of the epilogues -
    // spans 35-36 slots = more than the 64 banks a ds_read_b64 group covers: three lanes wrap onto busy banks and the group takes two
    // LDS cycles instead of one (SQ_LDS_BANK_CONFLICT 35-40 % of the LDS cycles of the 512- / 2048-point kernels, rounds 2-4).  Runs of 32
    // never cross a multiple of 64, so now they are 32 consecutive slots.  The stores stay conflict-free: Ns = 8 stores put eight
    // consecutive lanes on eight consecutive slots and the next eight lanes 72 slots = 8 (mod 16) eight-byte bank units later; the first
    // exchange, whose stores the one-slot pad was made for (lane j -> 8 j + t), gets its own image - slot1() below.
    // slot1: the first exchange (pass-1 outputs 8 j + t, read back as j' + t' M / 8) as an 8 x M/8 matrix with rows of M/8 + 4 slots:
    // stores are lane-linear inside row t, loads take row j' & 7, column (j' >> 3) + t' M / 64 - 32 lanes = 8 rows x 4 columns on
    // 4 (j' & 7) + (j' >> 3) (mod 32) = 32 distinct bank units.  No address arithmetic beyond a per-lane base either way.
    static constexpr int ROW1 = M / 8 + 4;
    __device__ static __forceinline__ constexpr int slot1(int i) { return (SWZ || !PAD64) ? slot(i) : (i & 7) * ROW1 + (i >> 3); }
    static_assert(SWZ || !PAD64 || 8 * ROW1 <= SLOTS, "the first exchange's image fits the buffer");
};

// ---- per-lane twiddles, held in registers for the lifetime of the kernel ------------------------------------------
// A butterfly with base exponent e multiplies input t by W^(e t), t = 1..R-1.
//   default: only W^e, W^2e, W^4e are stored (exactly rounded table entries) and the other four are single products of two
//        of them - 12 complex multiplies per radix-8 butterfly, 5 of them on twiddles alone;
//   MST_FFT2_FULL_TW=1 (n_fft <= 2048 only): all R-1 factors fetched from the table, 7 multiplies.  Measured: no gain
//        (2048 forward 27.9 -> 30.2 us, 512 backward 51.8 -> 52.4 us) - these passes wait on LDS and barriers, not on the VALU.
#ifndef MST_FFT2_FULL_TW
#define MST_FFT2_FULL_TW 0
#endif
template <int N>
struct LaneTw {
    using S = FftShape<N>;
    static constexpr bool FULL = MST_FFT2_FULL_TW && N <= 2048;
    static constexpr int LB = ilog2c(S::RL);                       // bases of the last pass
    float2 mid[S::NP - 2][FULL ? 7 : 3];                           // passes 2 .. NP-1 (radix 8, Ns = 8^(p-1))
    float2 last[S::NBL][FULL ? S::RL - 1 : LB];                    // pass NP (radix RL, Ns = M / RL, k = j)
    // tw = (cos, -sin)(2 pi t / N), t < N
    __device__ __forceinline__ void init(const float2* __restrict__ tw, int lane) {
        int Ns = 8;
#pragma unroll
        for (int p = 0; p < S::NP - 2; ++p) {
            const int e = (lane & (Ns - 1)) * (S::M / (Ns * 8));  // W_(8 Ns)^(k t) = W_M^(k t M / (8 Ns))
            if constexpr (FULL) {
#pragma unroll
                for (int t = 1; t < 8; ++t) mid[p][t - 1] = tw[(S::TWSCALE * (e * t)) & (N - 1)];
            } else {
#pragma unroll
                for (int b = 0; b < 3; ++b) mid[p][b] = tw[(S::TWSCALE * (e << b)) & (N - 1)];
            }
            Ns *= 8;
        }
#pragma unroll
        for (int u = 0; u < S::NBL; ++u) {
            const int j = lane + u * S::LG;
            if constexpr (FULL) {
#pragma unroll
                for (int t = 1; t < S::RL; ++t) last[u][t - 1] = tw[(S::TWSCALE * (j * t)) & (N - 1)];
            } else {
#pragma unroll
                for (int b = 0; b < LB; ++b) last[u][b] = tw[(S::TWSCALE * (j << b)) & (N - 1)];
            }
        }
    }
};
template <int R, bool FULL>
__device__ __forceinline__ void tw_apply(float2* v, const float2* base) {  // v[t] *= W^(e t)
    if constexpr (FULL) {
#pragma unroll
        for (int t = 1; t < R; ++t) v[t] = cmul(v[t], base[t - 1]);
    } else {
        v[1] = cmul(v[1], base[0]);
        v[2] = cmul(v[2], base[1]);
        v[3] = cmul(v[3], cmul(base[0], base[1]));
        if (R == 8) {
            v[4] = cmul(v[4], base[2]);
            v[5] = cmul(v[5], cmul(base[0], base[2]));
            v[6] = cmul(v[6], cmul(base[1], base[2]));
            v[7] = cmul(v[7], cmul(cmul(base[0], base[1]), base[2]));
        }
    }
}

// ---- passes.  `buf` = one sequence's SLOTS float2 in LDS; every lane of the frame's group calls them; the group barrier
// is __syncthreads() (one frame group per workgroup) and is placed by the caller.
// pass 1: v[t] = input element lane + t LG; leaves its result in buf.
template <int N>
__device__ __forceinline__ void fft_first(float2* v, float2* __restrict__ buf, int lane) {
    using S = FftShape<N>;
    butterfly<8>(v);
#pragma unroll
    for (int t = 0; t < 8; ++t) buf[S::slot1(lane * 8 + t)] = v[t];
}
// FIRST: the loads of pass 2, which read the first exchange's image (slot1)
template <int N, bool FIRST = false>
__device__ __forceinline__ void fft_load8(float2* v, const float2* __restrict__ buf, int lane) {
    using S = FftShape<N>;
#pragma unroll
    for (int t = 0; t < 8; ++t) v[t] = buf[FIRST ? S::slot1(lane + t * (S::M / 8)) : S::slot(lane + t * (S::M / 8))];
}
// middle pass P (2 <= P < NP), Ns = 8^(P-1): twiddle + butterfly + store (after fft_load8 and a barrier)
template <int N, int P>
__device__ __forceinline__ void fft_mid_store(float2* v, float2* __restrict__ buf, const LaneTw<N>& tw, int lane) {
    using S = FftShape<N>;
    constexpr int Ns = P == 2 ? 8 : 64;
    static_assert(P == 2 || P == 3, "middle passes");
    tw_apply<8, LaneTw<N>::FULL>(v, tw.mid[P - 2]);
    butterfly<8>(v);
    const int k = lane & (Ns - 1);
    const int base = (lane - k) * 8 + k;  // element base + t Ns
#pragma unroll
    for (int t = 0; t < 8; ++t) buf[S::slot(base + t * Ns)] = v[t];
}
// last pass, butterfly u of this lane: afterwards v[t] = X[lane + u LG + t M / RL] (natural order, in registers)
template <int N>
__device__ __forceinline__ void fft_last(float2* v, const float2* __restrict__ buf, const LaneTw<N>& tw, int lane, int u) {
    using S = FftShape<N>;
    const int j = lane + u * S::LG;
#pragma unroll
    for (int t = 0; t < S::RL; ++t) v[t] = buf[S::slot(j + t * (S::M / S::RL))];
    tw_apply<S::RL, LaneTw<N>::FULL>(v, tw.last[u]);
    butterfly<S::RL>(v);
}

// Whole transform of one sequence whose first-pass inputs are in v[8]: on return X[lane + u LG + t M / RL] = o[u][t].
// Contains NP*2 - 2 barriers; buf may be overwritten by the caller after ONE more barrier.
template <int N>
__device__ __forceinline__ void fft_run(float2* v, float2 (*o)[FftShape<N>::RL], float2* __restrict__ buf, const LaneTw<N>& tw, int lane) {
    using S = FftShape<N>;
    fft_first<N>(v, buf, lane);
    group_lds_sync<S::LG>();
#ifdef MST_FFT2_FIRST_PASS_ONLY
    for (int u = 0; u < S::NBL; ++u) for (int t = 0; t < S::RL; ++t) o[u][t] = buf[S::slot1(lane + t)];
    return;
#endif
    fft_load8<N, true>(v, buf, lane);
    group_lds_sync<S::LG>();
    fft_mid_store<N, 2>(v, buf, tw, lane);
    group_lds_sync<S::LG>();
    if constexpr (S::NP == 4) {
        fft_load8<N>(v, buf, lane);
        group_lds_sync<S::LG>();
        fft_mid_store<N, 3>(v, buf, tw, lane);
        group_lds_sync<S::LG>();
    }
#pragma unroll
    for (int u = 0; u < S::NBL; ++u) fft_last<N>(o[u], buf, tw, lane, u);
}

// two sequences side by side (the even / odd halves of the 8192-point transform): same barriers as one
template <int N>
__device__ __forceinline__ void fft_run2(float2* va, float2* vb, float2 (*oa)[FftShape<N>::RL], float2 (*ob)[FftShape<N>::RL],
                                         float2* __restrict__ bufa, float2* __restrict__ bufb, const LaneTw<N>& tw, int lane) {
    using S = FftShape<N>;
    fft_first<N>(va, bufa, lane);
    fft_first<N>(vb, bufb, lane);
    group_lds_sync<S::LG>();
    fft_load8<N, true>(va, bufa, lane);
    fft_load8<N, true>(vb, bufb, lane);
    group_lds_sync<S::LG>();
    fft_mid_store<N, 2>(va, bufa, tw, lane);
    fft_mid_store<N, 2>(vb, bufb, tw, lane);
    group_lds_sync<S::LG>();
    if constexpr (S::NP == 4) {
        fft_load8<N>(va, bufa, lane);
        fft_load8<N>(vb, bufb, lane);
        group_lds_sync<S::LG>();
        fft_mid_store<N, 3>(va, bufa, tw, lane);
        fft_mid_store<N, 3>(vb, bufb, tw, lane);
        group_lds_sync<S::LG>();
    }
#pragma unroll
    for (int u = 0; u < S::NBL; ++u) {
        fft_last<N>(oa[u], bufa, tw, lane, u);
        fft_last<N>(ob[u], bufb, tw, lane, u);
    }
}

// W_16^t = (cos, -sin)(2 pi t / 16), t < 8 (the radix-2 split of the 8192-point transform: lane l owns i = l + 512 t)
__device__ __forceinline__ float2 w16(int t) {
    constexpr float c[8] = {1.0f, 0.92387953251128674f, 0.70710678118654752f, 0.38268343236508977f, 0.0f,
                            -0.38268343236508977f, -0.70710678118654752f, -0.92387953251128674f};
    constexpr float sn[8] = {0.0f, 0.38268343236508977f, 0.70710678118654752f, 0.92387953251128674f, 1.0f,
                             0.92387953251128674f, 0.70710678118654752f, 0.38268343236508977f};
    return make_float2(c[t], -sn[t]);
}


// 8192-point transform of a complex sequence z by 512 lanes: radix-2 decimation in frequency on the way in (even bins from
// z[i] + z[i + 4096], odd bins from the twiddled difference), then two independent 4096-point transforms side by side.
// raw(t) -> element lane + 512 t (t < 16).  WINDOW: multiply by the periodic Hann window, formed from the radix-2 twiddle the
// split needs anyway (W_N^i = wl W_16^t for i = lane + 512 t;  hann(i) = 0.5 - 0.5 Re W_N^i,  hann(i + N/2) = 0.5 + 0.5 Re W_N^i).
// On return (the caller adds ONE barrier) Z[2m] = bufe[slot(m)], Z[2m + 1] = bufo[slot(m)].  tw = LaneTw<8192>, wl = W_8192^lane.
// RAW_IN_LDS: raw() reads the transform's own LDS buffers (the previous transform's output, modified in place) - a barrier
// separates the last read from the first pass's stores, so that callers need not park 16 values in registers.
// HALF: the window carries a factor 1/2 (an exact scaling: the spectrum is the same bits, halved - see FrameLoader::split)
template <bool WINDOW, bool RAW_IN_LDS = false, bool HALF = false, typename F>
__device__ __forceinline__ void fft8192_from(F&& raw, float2* __restrict__ bufe, float2* __restrict__ bufo, const LaneTw<8192>& tw,
                                             float2 wl, int lane) {
    using S = FftShape<8192>;
    float2 e[8], d[8], oe[1][8], od[1][8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const float2 w = cmul(wl, w16(t));
        constexpr float hc = HALF ? 0.25f : 0.5f;
        const float h0 = WINDOW ? hc - hc * w.x : 1.0f, h1 = WINDOW ? hc + hc * w.x : 1.0f;
        const float2 za = raw(t), zb = raw(t + 8);
        const float2 a = make_float2(h0 * za.x, h0 * za.y), b = make_float2(h1 * zb.x, h1 * zb.y);
        e[t] = cadd(a, b);
        d[t] = cmul(csub(a, b), w);
    }
    if (RAW_IN_LDS) group_lds_sync<FftShape<8192>::LG>();
    fft_run2<8192>(e, d, oe, od, bufe, bufo, tw, lane);
    group_lds_sync<FftShape<8192>::LG>();
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        bufe[S::slot(lane + t * (S::M / 8))] = oe[0][t];
        bufo[S::slot(lane + t * (S::M / 8))] = od[0][t];
    }
}

}  // namespace mst
