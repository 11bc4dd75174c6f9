// mst_cnn.hip - the 3x3 convolutions of the Cnn14 spectrogram encoder on the MI355X matrix cores.
//
// Replaces the twelve nn.Conv2d (3x3, stride 1, padding 1, no bias) of the reference's encoder (mst/panns.py:27-85, 126-209;
// MIOpen in the reference) in forward, data-gradient and weight-gradient form.  gfx950 only:
//   bf16 operands  v_mfma_f32_16x16x32_bf16  (8 bf16 of K per lane and operand, fp32 accumulate)      - the production path
//   fp32 operands  v_mfma_f32_16x16x4_f32    (exact fp32 FMA chain at the vector rate)                - the parity path
//   bf16x3         fp32 operands in memory and LDS, split on the way into the matrix pipe: x = hi + lo with hi = bf16(x),
//                  lo = bf16(x - hi) (both round-to-nearest-even, v_cvt_pk_bf16_f32), x y ~ hi hi + hi lo + lo hi in the fp32
//                  accumulators - three v_mfma_f32_16x16x32_bf16 (48 matrix-pipe cycles per 16 x 16 x 32 block) where the fp32
//                  instruction needs eight v_mfma_f32_16x16x4_f32 (256).  Dropped: lo lo and the rounding of lo, ~2^-17 of a
//                  product - below the fp32 rounding of a K-term accumulation from K ~ 100 on.  (precision code 2)
//
// k_conv_igemm   implicit GEMM  D[co][pixel] = sum_(tap, ci) W[co][tap][ci] X[pixel + tap][ci]:
//                the WEIGHTS are the MFMA's A operand (rows = output channels) and the activations its B operand (columns =
//                pixels), so that a lane's four accumulator registers are four CONSECUTIVE channels of one pixel - an 8-byte
//                (bf16) / 16-byte (fp32) NHWC store, no LDS transpose in the epilogue.  Both operands are K-contiguous in
//                memory (NHWC activations, (co, tap, ci) weights): every fragment is one 16-byte LDS read.  Workgroup tile
//                BC channels x 128 pixels x 32 of K; four waves as 2 x 2, each (BC/2) x 64.  The per-channel sum / sum of
//                squares that BatchNorm needs ride in the epilogue (fp32 accumulators, before any rounding).
//                The data gradient is the same kernel on (dY, W^T rotated by 180 degrees) - see k_prep_weights.
// k_conv_igemm3 / k_conv_igemm4   the bf16 production form of that GEMM: tiles fetched by LDS-direct loads into a ring with counted
//                waits (3), the three taps of an image row sharing one activation segment (4).  k_conv_igemm stays for fp32 operands
//                and as the register-staged A/B baseline (-DMST_CONV_NO_GLDS).
// k_conv_wgrad3 / k_conv_wgrad4   bf16 weight gradient on the same loader + ds_read_b64_tr_b16 fragments; one tap (3) or all nine (4)
//                per workgroup.
// k_conv_wgrad   dW[co][tap][ci] = sum_pixel dY[pixel][co] X[pixel + tap][ci]: K = pixels, which is the STRIDED axis of both
//                NHWC operands; the tiles are staged pixel-major as they lie in memory and the fragments are gathered column
//                by column (bf16: eight 2-byte LDS reads per fragment on a 260-byte pitch that spreads the four k-groups
//                over the banks).  Split-K over pixel ranges, partial sums reduced in a fixed order by k_wgrad_reduce.
// k_conv1 / k_conv1_wgrad   the first layer (one input channel): 9 taps are no GEMM K - direct VALU forward; its weight
//                gradient (64 x 9 sums over every pixel) does go through the MFMA with the nine shifted spectrogram values of
//                a pixel as the "channels" of the B operand.
#include "mst_cnn.h"

namespace mst {

#ifndef MST_CONV_ABLATE
#define MST_CONV_ABLATE 0
#endif
#ifndef MST_CONV_X3_STAGES
#define MST_CONV_X3_STAGES 2  // LDS stages of the bf16x3 implicit GEMM
#endif
#ifndef MST_CONV_X3_BK
#define MST_CONV_X3_BK 32     // K per staged tile of the bf16x3 implicit GEMM (32 | 64)
#endif
#ifndef MST_CONV_LDS_STAGES
#define MST_CONV_LDS_STAGES 2  // A/B switch: 1 = one LDS stage, two barriers per K step
#endif
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float((uint32_t)v << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {  // round to nearest even
    uint32_t u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return f2bf(v); }
template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<bf16_t>(bf16_t v) { return bf2f(v); }

// bf16x3 operand split of eight consecutive-K fp32 values (2.5 VALU instructions per value: two packed converts, the widening
// shift / mask of hi, one packed subtract per pair)
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split8(const float (&x)[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const f32x2_t v = {x[2 * j], x[2 * j + 1]};
        const bf16x2_t h = __builtin_convertvector(v, bf16x2_t);
        const f32x2_t r = v - __builtin_convertvector(h, f32x2_t);
        const bf16x2_t l = __builtin_convertvector(r, bf16x2_t);
        hi[2 * j] = h[0]; hi[2 * j + 1] = h[1];
        lo[2 * j] = l[0]; lo[2 * j + 1] = l[1];
    }
}
// three-term split x = hi + mid + lo (24 bits of significand: every fp32 value exactly, up to the rounding of lo at 2^-26) for bf16x6
__device__ __forceinline__ void split8(const float (&x)[8], bf16x8& hi, bf16x8& mid, bf16x8& lo) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const f32x2_t v = {x[2 * j], x[2 * j + 1]};
        const bf16x2_t h = __builtin_convertvector(v, bf16x2_t);
        const f32x2_t r1 = v - __builtin_convertvector(h, f32x2_t);
        const bf16x2_t m = __builtin_convertvector(r1, bf16x2_t);
        const f32x2_t r2 = r1 - __builtin_convertvector(m, f32x2_t);
        const bf16x2_t l = __builtin_convertvector(r2, bf16x2_t);
        hi[2 * j] = h[0]; hi[2 * j + 1] = h[1];
        mid[2 * j] = m[0]; mid[2 * j + 1] = m[1];
        lo[2 * j] = l[0]; lo[2 * j + 1] = l[1];
    }
}
// the operand pieces of one fragment: TERMS = 3 (hi, lo) | 6 (hi, mid, lo)
template <int TERMS> struct Pieces {
    bf16x8 h, m, l;  // m unused for TERMS = 3
    __device__ __forceinline__ void set(const float (&x)[8]) {
        if constexpr (TERMS == 6) split8(x, h, m, l);
        else split8(x, h, l);
    }
};
#define MST_MMA(A, B, C) C = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B, C, 0, 0, 0)
// acc += a b over the kept terms, small terms first.  3: hi lo + lo hi + hi hi (dropped from 2^-16 of the product on);
// 6: every term down to 2^-16 x 2^-8 (lo hi, hi lo, mid mid, mid hi, hi mid, hi hi; dropped: 2^-24 of the product and below)
template <int TERMS>
__device__ __forceinline__ f32x4 mma_split(const Pieces<TERMS>& a, const Pieces<TERMS>& b, f32x4 acc) {
    if constexpr (TERMS == 6) {
        MST_MMA(a.l, b.h, acc);
        MST_MMA(a.h, b.l, acc);
        MST_MMA(a.m, b.m, acc);
        MST_MMA(a.m, b.h, acc);
        MST_MMA(a.h, b.m, acc);
    } else {
        MST_MMA(a.l, b.h, acc);
        MST_MMA(a.h, b.l, acc);
    }
    MST_MMA(a.h, b.h, acc);
    return acc;
}
template <typename T, int BK> struct Mma;
// EPC: elements per 16-byte chunk; BK: K extent of one staged tile; LDK: LDS row pitch (elements); off(): element offset of chunk
// kc of row `row`.  bf16: 64 of K per tile (two MFMA K steps between barriers) in 128-byte rows whose eight 16-byte slots are
// XOR-swizzled with (row >> 1) & 7 - the 16 rows x one slot of a fragment read, and the 8 slots x 2 rows of a staging write, each
// cover the 64 banks exactly once.  fp32: 32 of K in 144-byte rows (16 rows x 4 k of a fragment read hit 64 banks once).
// The 64-channel layers (64 -> 64 at full resolution: bound by their 2 GB of activations, not by the matrix pipe) keep 32 of K
// per tile - half the staging registers, five waves per SIMD (measured: 1.20 ms vs 1.80 ms with 64) - in 64-byte rows with the
// same XOR swizzle over their four slots (dense rows: 1.6e9 conflict cycles per step; ds_read_b128 is served in the lane groups
// {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... of MI355X_MICROARCH.md, not in 16 consecutive lanes).
template <int BK> struct Mma<bf16_t, BK> {
    static constexpr int EPC = 8, LDK = BK;
    __device__ static __forceinline__ int off(int row, int kc) { return row * LDK + ((kc ^ ((row >> 1) & (BK / 8 - 1))) << 3); }
};
template <int BK> struct Mma<float, BK> {
    static constexpr int EPC = 4, LDK = BK + 4;
    static_assert(BK == 32 || BK == 64, "fp32 tiles hold 32 of K (64: the bf16x3 form, two matrix-pipe steps per staged tile)");
    __device__ static __forceinline__ int off(int row, int kc) { return row * LDK + kc * 4; }
};

__device__ __forceinline__ void store4(bf16_t* p, f32x4 v) {
    uint2 u;
    u.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
    u.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
    *reinterpret_cast<uint2*>(p) = u;
}
__device__ __forceinline__ void store4(float* p, f32x4 v) { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }

int conv_pixel_tiles(int N, int H, int W) { return (int)(((int64_t)N * H * W + kConvPix - 1) / kConvPix); }

// =====================================================================================================================
template <typename T, int BC, int BK, int X3 = 0>
__global__ __launch_bounds__(256, 2) void k_conv_igemm(ConvArgs a) {  // 2: up to 256 registers - at the default budget hipcc parks the staging registers in AGPRs (104 v_accvgpr moves per 32 MFMAs)
    using MM = Mma<T, BK>;
    constexpr int BP = kConvPix, EPC = MM::EPC, LDK = MM::LDK, RC = BK / EPC;
    constexpr int WCH = BC * RC / 256, XCH = BP * RC / 256, MT = BC / 32, NT = 4;
    // two LDS stages: tile s + 1 is stored while tile s is multiplied - ONE barrier per K step
    constexpr int NB = X3 ? MST_CONV_X3_STAGES : MST_CONV_LDS_STAGES;
    __shared__ __attribute__((aligned(16))) T sWb[NB][BC * LDK];
    __shared__ __attribute__((aligned(16))) T sXb[NB][BP * LDK];
    __shared__ float red[2][BC][2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wy = wave >> 1, wx = wave & 1;
    const int g = lane >> 4, li = lane & 15;
    const int H = a.H, W = a.W, Cin = a.Cin, Cout = a.Cout;
    const int64_t P = (int64_t)a.N * H * W, p0 = (int64_t)blockIdx.x * BP;
    const int co0 = blockIdx.y * BC;
    const T* __restrict__ in = reinterpret_cast<const T*>(a.in);
    const T* __restrict__ w = reinterpret_cast<const T*>(a.w);

    int xoff[XCH];
    unsigned xmask[XCH];  // bit t: tap t of this chunk's pixel lies inside the image (nine compares here, one bit test per tile)
    int64_t xbase[XCH];
#pragma unroll
    for (int q = 0; q < XCH; ++q) {
        const int c = tid + q * 256, row = c / RC, kc = c % RC;
        const int64_t p = p0 + row;
        const bool ok = p < P;
        const int64_t pp = ok ? p : 0;
        const int r = (int)(pp % ((int64_t)H * W)), xh = r / W, xw = r % W;
        unsigned m = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t)
            if (ok && (unsigned)(xh + t / 3 - 1) < (unsigned)H && (unsigned)(xw + t % 3 - 1) < (unsigned)W) m |= 1u << t;
        xmask[q] = m;
        xbase[q] = pp * Cin + kc * EPC;
        xoff[q] = MM::off(row, kc);
    }
    int64_t wbase[WCH];
    int woff[WCH];
#pragma unroll
    for (int q = 0; q < WCH; ++q) {
        const int c = tid + q * 256, row = c / RC, kc = c % RC;
        wbase[q] = (int64_t)(co0 + row) * 9 * Cin + kc * EPC;
        woff[q] = MM::off(row, kc);
    }
    uint4 rw[WCH], rx[XCH];
    auto gload_to = [&](uint4* rx, uint4* rw, int tap, int c0) {
#if MST_CONV_ABLATE & 1  // timing diagnostics only (wrong results): no global loads inside the K loop
        if (tap > 0 || c0 > 0) return;
#endif
        const int dh = tap / 3 - 1, dw = tap % 3 - 1;
        const int64_t shift = ((int64_t)dh * W + dw) * Cin + c0;
#pragma unroll
        for (int q = 0; q < XCH; ++q)
            rx[q] = (xmask[q] >> tap) & 1u ? *reinterpret_cast<const uint4*>(in + xbase[q] + shift) : make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < WCH; ++q) rw[q] = *reinterpret_cast<const uint4*>(w + wbase[q] + (int64_t)tap * Cin + c0);
    };
    f32x4 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    int ksteps = Cin / BK, steps = 9 * ksteps, tap = 0, kq = 0, kq0 = 0;
    if (a.csplit) {  // this workgroup's slice of K
        tap = blockIdx.z / a.csplit;
        ksteps /= a.csplit;
        kq0 = (blockIdx.z % a.csplit) * ksteps;
        steps = ksteps;
    }
    auto gload = [&](int tap, int c0) { gload_to(rx, rw, tap, c0); };
    auto sstore = [&](T* sX, T* sW) {
#pragma unroll
        for (int q = 0; q < XCH; ++q) *reinterpret_cast<uint4*>(&sX[xoff[q]]) = rx[q];
#pragma unroll
        for (int q = 0; q < WCH; ++q) *reinterpret_cast<uint4*>(&sW[woff[q]]) = rw[q];
    };
    auto multiply = [&](const T* sX, const T* sW) {
#if MST_CONV_ABLATE & 2  // no LDS reads, no MFMAs
        return;
#endif
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int ks = 0; ks < BK / 32; ++ks) {
                bf16x8 af[MT], bfr[NT];
#pragma unroll
                for (int m = 0; m < MT; ++m) af[m] = *reinterpret_cast<const bf16x8*>(&sW[MM::off(wy * (BC / 2) + m * 16 + li, ks * 4 + g)]);
#pragma unroll
                for (int n = 0; n < NT; ++n) bfr[n] = *reinterpret_cast<const bf16x8*>(&sX[MM::off(wx * 64 + n * 16 + li, ks * 4 + g)]);
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < NT; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[m], bfr[n], acc[m][n], 0, 0, 0);
            }
        } else if constexpr (X3) {
            // bf16x3: lane (g, li) holds K = 8 g .. 8 g + 7 of row li - two 16-byte LDS reads per fragment (the 144-byte pitch puts the
            // sixteen rows of a read on sixteen different 16-byte bank groups), split into (hi, lo), three MFMAs per 16 x 16 block
            auto frag = [&](const float* p, Pieces<X3>& f) {
                const float4 v0 = *reinterpret_cast<const float4*>(p), v1 = *reinterpret_cast<const float4*>(p + 4);
                const float x[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                f.set(x);
            };
#pragma unroll
            for (int ks = 0; ks < BK / 32; ++ks) {
                Pieces<X3> af[MT], bf[NT];
#pragma unroll
                for (int m = 0; m < MT; ++m) frag(&sW[(wy * (BC / 2) + m * 16 + li) * LDK + ks * 32 + g * 8], af[m]);
#pragma unroll
                for (int n = 0; n < NT; ++n) frag(&sX[(wx * 64 + n * 16 + li) * LDK + ks * 32 + g * 8], bf[n]);
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < NT; ++n) acc[m][n] = mma_split<X3>(af[m], bf[n], acc[m][n]);
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < BK / 4; ++kk) {
                float af[MT], bfr[NT];
#pragma unroll
                for (int m = 0; m < MT; ++m) af[m] = sW[(wy * (BC / 2) + m * 16 + li) * LDK + kk * 4 + g];
#pragma unroll
                for (int n = 0; n < NT; ++n) bfr[n] = sX[(wx * 64 + n * 16 + li) * LDK + kk * 4 + g];
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < NT; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m], bfr[n], acc[m][n], 0, 0, 0);
            }
        }
    };
    auto advance = [&]() { if (++kq == ksteps) { kq = 0; ++tap; } };
    gload(tap, kq0 * BK);
    if constexpr (NB == 1) {
        for (int s = 0; s < steps; ++s) {
            sstore(sXb[0], sWb[0]);
            __syncthreads();
            advance();
            if (s + 1 < steps) gload(tap, (kq0 + kq) * BK);  // next tile's global loads fly while this one is multiplied
            multiply(sXb[0], sWb[0]);
            __syncthreads();
        }
    } else {
        sstore(sXb[0], sWb[0]);
        advance();
        if (steps > 1) gload(tap, (kq0 + kq) * BK);
        __syncthreads();
        for (int s = 0; s < steps; ++s) {
            const int cur = s & 1;
            multiply(sXb[cur], sWb[cur]);
            if (s + 1 < steps) {
                sstore(sXb[cur ^ 1], sWb[cur ^ 1]);  // tile s + 1 (in registers since the previous step) into the idle stage
                advance();
                if (s + 2 < steps) gload(tap, (kq0 + kq) * BK);
            }
            __syncthreads();
        }
    }
    // ---- epilogue: D row = channel (lane >> 4) * 4 + r, D column = pixel lane & 15
    if (a.csplit) {
        float* __restrict__ kp = a.kpart + (int64_t)blockIdx.z * P * Cout;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int64_t p = p0 + wx * 64 + n * 16 + li;
            if (p < P) {
#pragma unroll
                for (int m = 0; m < MT; ++m) store4(kp + p * Cout + co0 + wy * (BC / 2) + m * 16 + g * 4, acc[m][n]);
            }
        }
        return;
    }
    T* __restrict__ out = reinterpret_cast<T*>(a.out);
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int64_t p = p0 + wx * 64 + n * 16 + li;
        if (p < P) {
#pragma unroll
            for (int m = 0; m < MT; ++m) store4(out + p * Cout + co0 + wy * (BC / 2) + m * 16 + g * 4, acc[m][n]);
        }
    }
    if (a.part) {  // rows of pixels beyond P multiplied zeros: they add nothing
#pragma unroll
        for (int m = 0; m < MT; ++m) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    s1 += acc[m][n][r];
                    s2 = fmaf(acc[m][n][r], acc[m][n][r], s2);
                }
#pragma unroll
                for (int msk = 1; msk < 16; msk <<= 1) {
                    s1 += __shfl_xor(s1, msk);
                    s2 += __shfl_xor(s2, msk);
                }
                if (li == 0) {
                    red[wx][wy * (BC / 2) + m * 16 + g * 4 + r][0] = s1;
                    red[wx][wy * (BC / 2) + m * 16 + g * 4 + r][1] = s2;
                }
            }
        }
        __syncthreads();
        if (tid < BC) {
            float* o = a.part + ((int64_t)blockIdx.x * Cout + co0 + tid) * 2;
            o[0] = red[0][tid][0] + red[1][tid][0];
            o[1] = red[0][tid][1] + red[1][tid][1];
        }
    }
}

// =====================================================================================================================
// k_conv_igemm3 (bf16): k_conv_igemm's GEMM with its tiles brought in by LDS-direct loads (global_load_lds_dwordx4: 1 KiB per wave
// instruction, no staging registers, no ds_write pass) into a ring of NS stages, so that NS - 2 whole K tiles stay in flight
// ACROSS the barrier of the tile being multiplied (k_conv_igemm: one tile, parked in 32 registers; the ablation that showed the
// load path - not the matrix pipe - binding it is in DESIGN 9.3).
//   * The LDS image of such a load is lane-linear (wave-uniform base + 16 lane): a tile is rows of 32 of K = four 16-byte slots,
//     lane l of a wave instruction fills slot l & 3 of row base + (l >> 2).  The XOR swizzle of the fragment reads therefore sits on
//     the SOURCE address: slot s of a row receives K chunk s ^ ((row >> 1) & 3), the involution Mma<bf16_t, 32>::off() reads with.
//   * Taps outside the image fetch from a 64-byte zero page instead of the activations.
//   * hipcc orders every ds_read it can see behind ALL outstanding LDS-direct loads (s_waitcnt vmcnt(0)), which would empty the ring
//     each step; the fragment reads are therefore asm statements with the counted waits written out: vmcnt((NS - 2) G) (G loads per
//     lane and tile: this wave's share of tile s has landed) -> s_barrier (everyone's share has; everyone is done reading tile
//     s - 1) -> issue tile s + NS - 1 into the slot of tile s - 1 -> read fragments -> lgkmcnt(0) -> MFMAs.  One barrier per K tile.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;
__device__ uint4 g_conv_zero[4];  // never written

template <int OFF> __device__ __forceinline__ u32x4 lds_read16(uint32_t addr) {
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
template <int N, int STRIDE, int I = 0> __device__ __forceinline__ void lds_read_frags(u32x4 (&r)[N], uint32_t addr) {
    r[I] = lds_read16<I * STRIDE>(addr);
    if constexpr (I + 1 < N) lds_read_frags<N, STRIDE, I + 1>(r, addr);
}
// after the s_waitcnt statement: the consumers of r must not be scheduled above it (volatile asm statements keep their order)
template <int N> __device__ __forceinline__ void lds_frags_landed(u32x4 (&r)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("" : "+v"(r[i]));
}
#ifndef MST_CONV_GLDS_STAGES
#define MST_CONV_GLDS_STAGES 3
#endif
#ifndef MST_CONV_XCD_REMAP
#define MST_CONV_XCD_REMAP 1  // pixel tiles in eight contiguous ranges, one per XCD (neighbouring image rows share an L2)
#endif
template <int BC, int BP, int NS>
__global__ __launch_bounds__(256, 2) void k_conv_igemm3(ConvArgs a) {
    constexpr int BK = 32, MT = BC / 32, NT = BP / 32;
    constexpr int XG = BP * 4 / 256, WG = BC * 4 / 256, G = XG + WG;  // 1 KiB wave instructions per lane and tile
    constexpr int SW = BC * BK, SX = BP * BK;                           // elements per stage
    static_assert(NS >= 3 && NS <= 6, "ring depth");
    __shared__ __attribute__((aligned(1024))) bf16_t sW[NS * SW];
    __shared__ __attribute__((aligned(1024))) bf16_t sX[NS * SX];
    __shared__ float red[2][BC][2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wy = wave >> 1, wx = wave & 1;
    const int g = lane >> 4, li = lane & 15;
    const int H = a.H, W = a.W, Cin = a.Cin, Cout = a.Cout;
    int bx = blockIdx.x;
    if (MST_CONV_XCD_REMAP) {  // workgroup ids go round-robin over the 8 XCDs: give XCD j the j-th contiguous range of tiles
        const int gx = gridDim.x, q = gx >> 3, r = gx & 7, xcd = bx & 7, idx = bx >> 3;
        bx = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int64_t P = (int64_t)a.N * H * W, p0 = (int64_t)bx * BP;
    const int co0 = blockIdx.y * BC;
    const bf16_t* __restrict__ in = reinterpret_cast<const bf16_t*>(a.in);
    const bf16_t* __restrict__ w = reinterpret_cast<const bf16_t*>(a.w);
    const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_conv_zero);

    // loader: wave instruction (q, wave) fills rows [(4 q + wave) 16, +16) of the tile
    unsigned xmask[XG];
    int64_t xbase[XG], wbase[WG];
#pragma unroll
    for (int q = 0; q < XG; ++q) {
        const int row = (q * 4 + wave) * 16 + (lane >> 2), kc = (lane & 3) ^ ((row >> 1) & 3);
        const int64_t p = p0 + row;
        const bool ok = p < P;
        const int64_t pp = ok ? p : 0;
        const int r = (int)(pp % ((int64_t)H * W)), xh = r / W, xw = r % W;
        unsigned m = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t)
            if (ok && (unsigned)(xh + t / 3 - 1) < (unsigned)H && (unsigned)(xw + t % 3 - 1) < (unsigned)W) m |= 1u << t;
        xmask[q] = m;
        xbase[q] = pp * Cin + kc * 8;
    }
#pragma unroll
    for (int q = 0; q < WG; ++q) {
        const int row = (q * 4 + wave) * 16 + (lane >> 2), kc = (lane & 3) ^ ((row >> 1) & 3);
        wbase[q] = (int64_t)(co0 + row) * 9 * Cin + kc * 8;
    }
    int ksteps = Cin / BK, steps = 9 * ksteps, itap = 0, ikq = 0, kq0 = 0;
    if (a.csplit) {  // this workgroup's slice of K
        itap = blockIdx.z / a.csplit;
        ksteps /= a.csplit;
        kq0 = (blockIdx.z % a.csplit) * ksteps;
        steps = ksteps;
    }
    auto issue = [&](int stage) {  // tile (itap, ikq) -> ring slot `stage`; advances the issue cursor
        const int c0 = (kq0 + ikq) * BK;
        const int64_t shift = ((int64_t)(itap / 3 - 1) * W + (itap % 3 - 1)) * Cin + c0;
#pragma unroll
        for (int q = 0; q < XG; ++q) {
            const bf16_t* src = (xmask[q] >> itap) & 1u ? in + xbase[q] + shift : zero;
            __builtin_amdgcn_global_load_lds((glb_void_t*)src, (lds_void_t*)(uintptr_t)(sX + stage * SX + (q * 4 + wave) * 512), 16, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < WG; ++q) {
            const bf16_t* src = w + wbase[q] + (int64_t)itap * Cin + c0;
            __builtin_amdgcn_global_load_lds((glb_void_t*)src, (lds_void_t*)(uintptr_t)(sW + stage * SW + (q * 4 + wave) * 512), 16, 0, 0);
        }
        if (++ikq == ksteps) { ikq = 0; ++itap; }
    };
    // fragment addresses (bytes): row = wy BC/2 + 16 m + li resp. wx 64 + 16 n + li; (row >> 1) & 3 = (li >> 1) & 3 for every m, n
    const uint32_t fsw = (uint32_t)(((g ^ ((li >> 1) & 3)) << 3) * 2);
    const uint32_t aaddr0 = (uint32_t)(uintptr_t)sW + (uint32_t)((wy * (BC / 2) + li) * BK * 2) + fsw;
    const uint32_t baddr0 = (uint32_t)(uintptr_t)sX + (uint32_t)((wx * (BP / 2) + li) * BK * 2) + fsw;
    f32x4 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
        if (s < steps) issue(s);
    int cur = 0;
    for (int s = 0; s < steps; ++s) {
        // tiles issued beyond s: min(NS - 2, steps - 1 - s); the short tail drains everything
        if (steps - 1 - s >= NS - 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * G) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (s + NS - 1 < steps) issue(cur == 0 ? NS - 1 : cur - 1);
        const uint32_t aaddr = aaddr0 + (uint32_t)(cur * SW * 2), baddr = baddr0 + (uint32_t)(cur * SX * 2);
        u32x4 rb[NT], ra[MT];
        lds_read_frags<NT, 16 * BK * 2>(rb, baddr);
        lds_read_frags<MT, 16 * BK * 2>(ra, aaddr);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        lds_frags_landed(rb);
        lds_frags_landed(ra);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n)
                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ra[m]), __builtin_bit_cast(bf16x8, rb[n]), acc[m][n], 0, 0, 0);
        cur = cur + 1 == NS ? 0 : cur + 1;
    }
    // ---- epilogue (as k_conv_igemm): D row = channel (lane >> 4) * 4 + r, D column = pixel lane & 15
    if (a.csplit) {
        float* __restrict__ kp = a.kpart + (int64_t)blockIdx.z * P * Cout;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int64_t p = p0 + wx * (BP / 2) + n * 16 + li;
            if (p < P) {
#pragma unroll
                for (int m = 0; m < MT; ++m) store4(kp + p * Cout + co0 + wy * (BC / 2) + m * 16 + g * 4, acc[m][n]);
            }
        }
        return;
    }
    bf16_t* __restrict__ out = reinterpret_cast<bf16_t*>(a.out);
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int64_t p = p0 + wx * (BP / 2) + n * 16 + li;
        if (p < P) {
#pragma unroll
            for (int m = 0; m < MT; ++m) store4(out + p * Cout + co0 + wy * (BC / 2) + m * 16 + g * 4, acc[m][n]);
        }
    }
    if (a.part) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    s1 += acc[m][n][r];
                    s2 = fmaf(acc[m][n][r], acc[m][n][r], s2);
                }
#pragma unroll
                for (int msk = 1; msk < 16; msk <<= 1) {
                    s1 += __shfl_xor(s1, msk);
                    s2 += __shfl_xor(s2, msk);
                }
                if (li == 0) {
                    red[wx][wy * (BC / 2) + m * 16 + g * 4 + r][0] = s1;
                    red[wx][wy * (BC / 2) + m * 16 + g * 4 + r][1] = s2;
                }
            }
        }
        __syncthreads();
        if (tid < BC) {
            float* o = a.part + ((int64_t)bx * Cout + co0 + tid) * 2;
            o[0] = red[0][tid][0] + red[1][tid][0];
            o[1] = red[0][tid][1] + red[1][tid][1];
        }
    }
}

// =====================================================================================================================
// k_conv_igemm4 (bf16, no split K): k_conv_igemm3 with the activation bytes cut to a third.  k_conv_igemm3 fills LDS at ~20 bytes per
// clock and CU (13 TB/s over the chip) and that, not the matrix pipe, is its limit on the large layers.  Here the three taps of an
// image row share ONE activation segment: rows j = 0 .. BP + 1 of the segment hold input pixels p0 - 1 + j (+ dh W), staged once per
// (dh, channel block) into a double buffer, and tap dw reads its B fragments from rows shifted by dw + 1 (the slot swizzle
// (row >> 1) & 3 is conflict-free for all three shifts: tools/lds_conflicts.py).  What a shift drags in across a row end, an image
// border or the end of the batch is zeroed per lane (a lane's B fragment is eight channels of ONE pixel).  Weight tiles (one tap x 32
// channels) run through a three-slot ring; a step = one tap = MT x NT MFMAs per wave between two barriers.  Counted waits: at step
// t only what step t - 1 issued may still be in flight (the next weight tile, plus the next segment when t - 1 opened a tap row).
template <int BC, int BP>
__global__ __launch_bounds__(256, 2) void k_conv_igemm4(ConvArgs a) {
    constexpr int BK = 32, MT = BC / 32, NT = BP / 32;
    constexpr int XR = BP + 16, XI = XR / 16;              // segment rows (BP + 2 used), 1 KiB wave instructions per segment
    constexpr int XG0 = (XI + 3) / 4, XG1 = XI / 4;        // of which wave 0 / waves 1-3 issue
    constexpr int WGc = BC / 64;                           // weight instructions per wave and tap
    constexpr int SW = BC * BK, SX = XR * BK;              // elements per weight slot / segment buffer
    static_assert(XI % 4 == 1, "wave 0 takes the odd segment instruction");
    __shared__ __attribute__((aligned(1024))) bf16_t sW[3 * SW];
    __shared__ __attribute__((aligned(1024))) bf16_t sX[2 * SX];
    __shared__ float red[2][BC][2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wy = wave >> 1, wx = wave & 1;
    const int g = lane >> 4, li = lane & 15;
    const int H = a.H, W = a.W, Cin = a.Cin, Cout = a.Cout, HW = H * W;
    int bx = blockIdx.x;
    if (MST_CONV_XCD_REMAP) {
        const int gx = gridDim.x, q = gx >> 3, r = gx & 7, xcd = bx & 7, idx = bx >> 3;
        bx = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int64_t P = (int64_t)a.N * HW, p0 = (int64_t)bx * BP;
    const int co0 = blockIdx.y * BC;
    const bf16_t* __restrict__ in = reinterpret_cast<const bf16_t*>(a.in);
    const bf16_t* __restrict__ w = reinterpret_cast<const bf16_t*>(a.w);
    const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_conv_zero);

    // segment loader: instruction i = wave + 4 q fills rows [16 i, +16); row j <-> centre pixel c = p0 - 1 + j
    // (the K chunk of a lane, (lane & 3) ^ ((row >> 1) & 3), is the same for all its instructions: rows differ by multiples of 64)
    const int kc = (lane & 3) ^ ((lane >> 3) & 3);
    unsigned xmask = 0;  // bit 3 q + d: row h + d - 1 of the image of instruction q's centre pixel exists
#pragma unroll
    for (int q = 0; q < XG0; ++q) {
        const int row = (wave + 4 * q) * 16 + (lane >> 2);
        const int64_t c = p0 - 1 + row;
        if (c >= 0 && c < P && row < BP + 2) {
            const int h = (int)(c % HW) / W;
#pragma unroll
            for (int d = 0; d < 3; ++d)
                if ((unsigned)(h + d - 1) < (unsigned)H) xmask |= 1u << (3 * q + d);
        }
    }
    const bf16_t* xsrc0 = in + (p0 - 1 + wave * 16 + (lane >> 2)) * Cin + kc * 8;  // + 64 q Cin per instruction; dereferenced only where xmask allows
    const bf16_t* wsrc0 = w + (int64_t)(co0 + wave * 16 + (lane >> 2)) * 9 * Cin + kc * 8;
    // horizontal validity of this lane's output pixels: bit n = the left neighbour exists, bit 8 + n = the right one does
    unsigned hmask = 0;
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int64_t p = p0 + wx * (BP / 2) + n * 16 + li;
        if (p < P) {
            const int ww = (int)(p % HW) % W;
            if (ww > 0) hmask |= 1u << n;
            if (ww < W - 1) hmask |= 1u << (8 + n);
        }
    }
    constexpr unsigned kAll = (1u << NT) - 1u;
    const bool edgeL = __builtin_amdgcn_ballot_w64((hmask & kAll) != kAll) != 0;          // wave-uniform: some lane needs zeroing
    const bool edgeR = __builtin_amdgcn_ballot_w64(((hmask >> 8) & kAll) != kAll) != 0;

    const int chunks = Cin / BK, K = 3 * chunks;  // tap-row groups k = dh_idx * chunks + chunk; steps t = 3 k + dw_idx
    int wk_dh = 0, wk_ch = 0, wk_d = 0, wslot = 0;  // weight issue cursor
    auto issue_w = [&]() {
        const int tap = wk_dh * 3 + wk_d, c0 = wk_ch * BK;
#pragma unroll
        for (int q = 0; q < WGc; ++q) {
            const bf16_t* src = wsrc0 + ((int64_t)q * 64 * 9 + tap) * Cin + c0;
            __builtin_amdgcn_global_load_lds((glb_void_t*)src, (lds_void_t*)(uintptr_t)(sW + wslot * SW + (q * 4 + wave) * 512), 16, 0, 0);
        }
        wslot = wslot == 2 ? 0 : wslot + 1;
        if (++wk_d == 3) { wk_d = 0; if (++wk_ch == chunks) { wk_ch = 0; ++wk_dh; } }
    };
    int xk_dh = 0, xk_ch = 0, xbuf = 0;  // segment issue cursor
    auto issue_x = [&]() {
        const int64_t shift = (int64_t)(xk_dh - 1) * W * Cin + xk_ch * BK;
#pragma unroll
        for (int q = 0; q < XG0; ++q) {
            if (q < XG1 || wave == 0) {
                const bf16_t* src = (xmask >> (3 * q + xk_dh)) & 1u ? xsrc0 + (int64_t)q * 64 * Cin + shift : zero;
                __builtin_amdgcn_global_load_lds((glb_void_t*)src, (lds_void_t*)(uintptr_t)(sX + xbuf * SX + (wave + 4 * q) * 512), 16, 0, 0);
            }
        }
        xbuf ^= 1;
        if (++xk_ch == chunks) { xk_ch = 0; ++xk_dh; }
    };
    const uint32_t fsw = (uint32_t)(((li >> 1) & 3));
    const uint32_t aaddr0 = (uint32_t)(uintptr_t)sW + (uint32_t)((wy * (BC / 2) + li) * BK * 2) + (((uint32_t)g ^ fsw) << 4);
    // B rows are shifted by d = 0, 1, 2: the slot swizzle follows the row, (row >> 1) & 3 with row = li + d (the 16 n and wx BP/2 terms vanish)
    uint32_t baddr0[3];
#pragma unroll
    for (int d = 0; d < 3; ++d)
        baddr0[d] = (uint32_t)(uintptr_t)sX + (uint32_t)((wx * (BP / 2) + li + d) * BK * 2) + (((uint32_t)g ^ (uint32_t)(((li + d) >> 1) & 3)) << 4);
    f32x4 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    issue_x();
    issue_w();
    issue_w();
    int rslot = 0;  // weight slot of the current step
    for (int k = 0; k < K; ++k) {
        const bool last = k == K - 1;
        const uint32_t xoff = (uint32_t)((k & 1) * SX * 2);
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            // in flight may stay: what the previous step issued
            if (d == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WGc) : "memory");
            else if (d == 1) {
                if (last) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WGc) : "memory");
                else if (wave == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WGc + XG0) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WGc + XG1) : "memory");
            } else {
                if (last) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WGc) : "memory");
            }
            __builtin_amdgcn_s_barrier();
            if (!(last && d >= 1)) issue_w();          // tile t + 2 into the slot of tile t - 1
            if (d == 0 && !last) issue_x();            // next segment into the buffer the previous tap row used
            const uint32_t aaddr = aaddr0 + (uint32_t)(rslot * SW * 2), baddr = baddr0[d] + xoff;
            u32x4 ra[MT];
            lds_read_frags<MT, 16 * BK * 2>(ra, aaddr);
#pragma unroll
            for (int nh = 0; nh < NT / 4; ++nh) {  // four pixel columns at a time (all eight: 32 more registers, spills)
                u32x4 rb[4];
                lds_read_frags<4, 16 * BK * 2>(rb, baddr + (uint32_t)(nh * 64 * BK * 2));
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                lds_frags_landed(rb);
                if (nh == 0) lds_frags_landed(ra);
                if (d == 0 && edgeL) {
#pragma unroll
                    for (int n = 0; n < 4; ++n)
                        if (!((hmask >> (nh * 4 + n)) & 1u)) rb[n] = u32x4{0u, 0u, 0u, 0u};
                }
                if (d == 2 && edgeR) {
#pragma unroll
                    for (int n = 0; n < 4; ++n)
                        if (!((hmask >> (8 + nh * 4 + n)) & 1u)) rb[n] = u32x4{0u, 0u, 0u, 0u};
                }
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < 4; ++n)
                        acc[m][nh * 4 + n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ra[m]), __builtin_bit_cast(bf16x8, rb[n]), acc[m][nh * 4 + n], 0, 0, 0);
            }
            rslot = rslot == 2 ? 0 : rslot + 1;
        }
    }
    // ---- epilogue (as k_conv_igemm3)
    bf16_t* __restrict__ out = reinterpret_cast<bf16_t*>(a.out);
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int64_t p = p0 + wx * (BP / 2) + n * 16 + li;
        if (p < P) {
#pragma unroll
            for (int m = 0; m < MT; ++m) store4(out + p * Cout + co0 + wy * (BC / 2) + m * 16 + g * 4, acc[m][n]);
        }
    }
    if (a.part) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    s1 += acc[m][n][r];
                    s2 = fmaf(acc[m][n][r], acc[m][n][r], s2);
                }
#pragma unroll
                for (int msk = 1; msk < 16; msk <<= 1) {
                    s1 += __shfl_xor(s1, msk);
                    s2 += __shfl_xor(s2, msk);
                }
                if (li == 0) {
                    red[wx][wy * (BC / 2) + m * 16 + g * 4 + r][0] = s1;
                    red[wx][wy * (BC / 2) + m * 16 + g * 4 + r][1] = s2;
                }
            }
        }
        __syncthreads();
        if (tid < BC) {
            float* o = a.part + ((int64_t)bx * Cout + co0 + tid) * 2;
            o[0] = red[0][tid][0] + red[1][tid][0];
            o[1] = red[0][tid][1] + red[1][tid][1];
        }
    }
}

// =====================================================================================================================
template <typename T>
__global__ __launch_bounds__(256) void k_conv_splitk_reduce(const float* __restrict__ kpart, T* __restrict__ out, float* __restrict__ part, int64_t P,
                                                            int Cout, int nz) {
    // workgroup = (pixel tile, 32 channels): 8 lanes x 4 channels cover a 128-byte line, the 32 lane groups share the pixels; eight
    // slabs are requested before the first is added (one 4-byte load per slab and a dependent add was 0.45 TB/s on 75 MB)
    __shared__ float red[32][32][2];
    const int c4 = threadIdx.x & 7, pl = threadIdx.x >> 3, c = blockIdx.y * 32 + c4 * 4;
    const int64_t p0 = (int64_t)blockIdx.x * kConvPix;
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = pl; i < kConvPix; i += 32) {
        const int64_t p = p0 + i;
        if (p >= P) break;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        const float* src = kpart + p * Cout + c;
        const int64_t zs = P * Cout;
        for (int z0 = 0; z0 < nz; z0 += 8) {
            float4 t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = z0 + u < nz ? *reinterpret_cast<const float4*>(src + (z0 + u) * zs) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int u = 0; u < 8; ++u) {  // slabs in ascending order (fixed summation order)
                if (z0 + u < nz) { v[0] += t[u].x; v[1] += t[u].y; v[2] += t[u].z; v[3] += t[u].w; }
            }
        }
        store4(out + p * Cout + c, f32x4{v[0], v[1], v[2], v[3]});
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            s1[q] += v[q];
            s2[q] = fmaf(v[q], v[q], s2[q]);
        }
    }
    if (part) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            red[pl][c4 * 4 + q][0] = s1[q];
            red[pl][c4 * 4 + q][1] = s2[q];
        }
        __syncthreads();
        if (threadIdx.x < 64) {
            const int cl = threadIdx.x >> 1, w = threadIdx.x & 1;
            float acc = 0.f;
            for (int q = 0; q < 32; ++q) acc += red[q][cl][w];
            part[((int64_t)blockIdx.x * Cout + blockIdx.y * 32 + cl) * 2 + w] = acc;
        }
    }
}
static constexpr bool conv3_enabled() {
#ifdef MST_CONV_NO_GLDS
    return false;  // A/B switch: register-staged k_conv_igemm
#else
    return true;
#endif
}
static int conv_csplit(int N, int H, int W, int Cin, int Cout) {
    const int tiles = conv_pixel_tiles(N, H, W), bc = Cout % 128 == 0 ? 128 : 64, wgs = tiles * (Cout / bc);
    if (wgs >= 512) return 0;
    int cs = 1;
    while (wgs * 9 * cs < 1024 && Cin / (cs * 2) >= 128) cs *= 2;
    return cs;
}
size_t conv_splitk_bytes(int N, int H, int W, int Cin, int Cout) {
    const int cs = conv_csplit(N, H, W, Cin, Cout);
    return cs ? (size_t)9 * cs * N * H * W * Cout * sizeof(float) : 0;
}
int launch_conv3x3(int precision, ConvArgs a, hipStream_t s, float* kpart, size_t kpart_bytes) {
    const int tiles = conv_pixel_tiles(a.N, a.H, a.W);
    a.csplit = 0;
    a.kpart = nullptr;
    const size_t need = conv_splitk_bytes(a.N, a.H, a.W, a.Cin, a.Cout);
    if (need && kpart && need <= kpart_bytes) {
        a.csplit = conv_csplit(a.N, a.H, a.W, a.Cin, a.Cout);
        a.kpart = kpart;
    }
    const unsigned gz = a.csplit ? 9 * a.csplit : 1;
    constexpr int kConvPix2 = 256;  // pixels per workgroup of the fat tiles
    const int tiles2 = (int)(((int64_t)a.N * a.H * a.W + kConvPix2 - 1) / kConvPix2);
#ifndef MST_CONV_FAT_MIN
#define MST_CONV_FAT_MIN 1024  // workgroups from which the 256-pixel tile is used
#endif
    const bool fat_ok = precision == 0 && !a.csplit && a.Cout % 128 == 0 && (int64_t)tiles2 * (a.Cout / 128) >= MST_CONV_FAT_MIN;
#ifndef MST_CONV_TAPROW
#define MST_CONV_TAPROW 1  // A/B switch: 0 = k_conv_igemm3 everywhere
#endif
    if (MST_CONV_TAPROW && conv3_enabled() && precision == 0 && !a.csplit && (int64_t)tiles2 * (a.Cout / (a.Cout % 128 == 0 ? 128 : 64)) >= MST_CONV_FAT_MIN) {
        if (a.Cout % 128 == 0) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_conv_igemm4<128, 256>), dim3(tiles2, a.Cout / 128), dim3(256), 0, s, a);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_conv_igemm4<64, 256>), dim3(tiles2, a.Cout / 64), dim3(256), 0, s, a);
        return tiles2;
    }
    if (fat_ok && conv3_enabled()) {
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_conv_igemm3<128, 256, MST_CONV_GLDS_STAGES>), dim3(tiles2, a.Cout / 128), dim3(256), 0, s, a);
        return tiles2;
    }
    if (a.Cout % 128 == 0) {
        const dim3 grid(tiles, a.Cout / 128, gz);
        if (precision == 0 && conv3_enabled()) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_conv_igemm3<128, 128, MST_CONV_GLDS_STAGES>), grid, dim3(256), 0, s, a);
        else if (precision == 0) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_conv_igemm<bf16_t, 128, 64>), grid, dim3(256), 0, s, a);
        else if (precision == 2) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_conv_igemm<float, 128, MST_CONV_X3_BK, 3>), grid, dim3(256), 0, s, a);
        else if (precision == 3) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_conv_igemm<float, 128, MST_CONV_X3_BK, 6>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_conv_igemm<float, 128, 32>), grid, dim3(256), 0, s, a);
    } else {
        const dim3 grid(tiles, a.Cout / 64, gz);
        if (precision == 0 && conv3_enabled()) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_conv_igemm3<64, 128, MST_CONV_GLDS_STAGES>), grid, dim3(256), 0, s, a);
        else if (precision == 0) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_conv_igemm<bf16_t, 64, 32>), grid, dim3(256), 0, s, a);
        else if (precision == 2) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_conv_igemm<float, 64, MST_CONV_X3_BK, 3>), grid, dim3(256), 0, s, a);
        else if (precision == 3) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_conv_igemm<float, 64, MST_CONV_X3_BK, 6>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_conv_igemm<float, 64, 32>), grid, dim3(256), 0, s, a);
    }
    if (a.csplit) {
        const dim3 grid(tiles, a.Cout / 32);
        const int64_t P = (int64_t)a.N * a.H * a.W;
        if (precision == 0) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_conv_splitk_reduce<bf16_t>), grid, dim3(256), 0, s, a.kpart, (bf16_t*)a.out, a.part, P, a.Cout, (int)gz);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_conv_splitk_reduce<float>), grid, dim3(256), 0, s, a.kpart, (float*)a.out, a.part, P, a.Cout, (int)gz);
    }
    return tiles;
}

// =====================================================================================================================
// Weight gradient.  grid (channel tiles, 9 taps, splits); FIRST: the input is the fp32 spectrogram, "channel" j < 9 of the B
// operand is its value at tap j (BCI = 16, grid.y = 1).
template <typename T> struct WgPitch;
template <> struct WgPitch<bf16_t> { static constexpr int PAD = 2; };   // (C + 2) * 2 bytes = 4 (mod 16): k-groups 8 rows apart land 32 bytes apart
template <> struct WgPitch<float> { static constexpr int PAD = 16; };   // C * 4 + 64 bytes: consecutive rows fill the two halves of the 32 banks

template <typename T, int BCO, int BCI, bool FIRST, int X3 = 0>
__global__ __launch_bounds__(256, 2) void k_conv_wgrad(WgradArgs a) {
    constexpr int BKP = kWgradPix, EPC = Mma<T, 32>::EPC, PY = BCO + WgPitch<T>::PAD, PX = BCI + WgPitch<T>::PAD;
    constexpr int YCH = BKP * (BCO / EPC), XCH = FIRST ? 0 : BKP * (BCI / EPC);  // 16-byte chunks per tile
    constexpr int YQ = (YCH + 255) / 256, XQ = FIRST ? 1 : (XCH + 255) / 256;
    // waves: 2 x 2 over (co, ci); FIRST: 4 x 1 (every wave all 16 taps)
    constexpr int WCO = FIRST ? BCO / 4 : BCO / 2, WCI = FIRST ? BCI : BCI / 2, MT = WCO / 16, NT = WCI / 16;
    __shared__ __attribute__((aligned(16))) T sY[BKP * PY];
    __shared__ __attribute__((aligned(16))) T sX[BKP * PX];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wy = FIRST ? wave : wave >> 1, wx = FIRST ? 0 : wave & 1;
    const int g = lane >> 4, li = lane & 15;
    const int H = a.H, W = a.W, Cin = a.Cin, Cout = a.Cout, HW = H * W;
    const int64_t P = (int64_t)a.N * HW;
    const int ci_tiles = FIRST ? 1 : Cin / BCI;
    const int co0 = (blockIdx.x / ci_tiles) * BCO, ci0 = (blockIdx.x % ci_tiles) * BCI;
    const int tap = blockIdx.y, dh = tap / 3 - 1, dw = tap % 3 - 1;
    const T* __restrict__ dy = reinterpret_cast<const T*>(a.dy);
    const int64_t pstart = (int64_t)blockIdx.z * a.steps_per_split * BKP;

    f32x4 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    uint4 ry[YQ], rx[XQ];
    float rs = 0.f;  // FIRST: one gathered spectrogram value per thread and pass
    float rs2 = 0.f;
    auto gload = [&](int64_t pb) {
#pragma unroll
        for (int q = 0; q < YQ; ++q) {
            const int c = tid + q * 256, row = c / (BCO / EPC), col = c % (BCO / EPC);
            const int64_t p = pb + row;
            ry[q] = (c < YCH && p < P) ? *reinterpret_cast<const uint4*>(dy + p * Cout + co0 + col * EPC) : make_uint4(0, 0, 0, 0);
        }
        if constexpr (FIRST) {
            const float* __restrict__ sp = reinterpret_cast<const float*>(a.x);
            // thread -> (pixel row tid % 32, tap tid / 32) and a second pass for tap 8 (threads < 32)
            auto tapval = [&](int row, int t) -> float {
                const int64_t p = pb + row;
                if (p >= P) return 0.f;
                const int r = (int)(p % HW), h = r / W, ww = r % W, th = t / 3 - 1, tw = t % 3 - 1;
                if ((unsigned)(h + th) >= (unsigned)H || (unsigned)(ww + tw) >= (unsigned)W) return 0.f;
                return sp[p + (int64_t)th * W + tw];
            };
            rs = tapval(tid & 31, tid >> 5);
            rs2 = tid < 32 ? tapval(tid, 8) : 0.f;
        } else {
            const T* __restrict__ x = reinterpret_cast<const T*>(a.x);
#pragma unroll
            for (int q = 0; q < XQ; ++q) {
                const int c = tid + q * 256, row = c / (BCI / EPC), col = c % (BCI / EPC);
                const int64_t p = pb + row;
                bool ok = c < XCH && p < P;
                if (ok) {
                    const int r = (int)(p % HW), h = r / W, ww = r % W;
                    ok = (unsigned)(h + dh) < (unsigned)H && (unsigned)(ww + dw) < (unsigned)W;
                }
                rx[q] = ok ? *reinterpret_cast<const uint4*>(x + (p + (int64_t)dh * W + dw) * Cin + ci0 + col * EPC) : make_uint4(0, 0, 0, 0);
            }
        }
    };
    auto put16 = [&](T* dst, uint4 v) {  // the pitches are multiples of 4 bytes, not of 16
        uint32_t* d = reinterpret_cast<uint32_t*>(dst);
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    };
    gload(pstart);
    for (int s = 0; s < a.steps_per_split; ++s) {
#pragma unroll
        for (int q = 0; q < YQ; ++q) {
            const int c = tid + q * 256, row = c / (BCO / EPC), col = c % (BCO / EPC);
            if (c < YCH) put16(&sY[row * PY + col * EPC], ry[q]);
        }
        if constexpr (FIRST) {
            sX[(tid & 31) * PX + (tid >> 5)] = from_f32<T>(rs);
            if (tid < 32) sX[tid * PX + 8] = from_f32<T>(rs2);
            if (tid >= 32 && tid < 32 + 32 * 7) sX[((tid - 32) & 31) * PX + 9 + ((tid - 32) >> 5)] = from_f32<T>(0.f);
        } else {
#pragma unroll
            for (int q = 0; q < XQ; ++q) {
                const int c = tid + q * 256, row = c / (BCI / EPC), col = c % (BCI / EPC);
                if (c < XCH) put16(&sX[row * PX + col * EPC], rx[q]);
            }
        }
        __syncthreads();
        if (s + 1 < a.steps_per_split) gload(pstart + (int64_t)(s + 1) * BKP);
        if constexpr (sizeof(T) == 2) {
            bf16x8 af[MT], bfr[NT];
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const bf16_t* c = reinterpret_cast<const bf16_t*>(sY) + (8 * g) * PY + wy * WCO + m * 16 + li;
                union { bf16x8 v; bf16_t e[8]; } u;
#pragma unroll
                for (int j = 0; j < 8; ++j) u.e[j] = c[j * PY];
                af[m] = u.v;
            }
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const bf16_t* c = reinterpret_cast<const bf16_t*>(sX) + (8 * g) * PX + wx * WCI + n * 16 + li;
                union { bf16x8 v; bf16_t e[8]; } u;
#pragma unroll
                for (int j = 0; j < 8; ++j) u.e[j] = c[j * PX];
                bfr[n] = u.v;
            }
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[m], bfr[n], acc[m][n], 0, 0, 0);
        } else if constexpr (X3) {
            // bf16x3 (see k_conv_igemm): K = the 32 staged pixels, lane (g, li) gathers pixels 8 g .. 8 g + 7 of its channel column
            static_assert(BKP == 32, "one 16x16x32 step per staged tile");
            Pieces<X3> af[MT], bf[NT];
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                float x[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = sY[(8 * g + j) * PY + wy * WCO + m * 16 + li];
                af[m].set(x);
            }
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                float x[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = sX[(8 * g + j) * PX + wx * WCI + n * 16 + li];
                bf[n].set(x);
            }
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n) acc[m][n] = mma_split<X3>(af[m], bf[n], acc[m][n]);
        } else {
#pragma unroll
            for (int kk = 0; kk < BKP / 4; ++kk) {
                float af[MT], bfr[NT];
#pragma unroll
                for (int m = 0; m < MT; ++m) af[m] = sY[(kk * 4 + g) * PY + wy * WCO + m * 16 + li];
#pragma unroll
                for (int n = 0; n < NT; ++n) bfr[n] = sX[(kk * 4 + g) * PX + wx * WCI + n * 16 + li];
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < NT; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m], bfr[n], acc[m][n], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    // D row = co (lane >> 4) * 4 + r, D column = ci (or tap) lane & 15
    const int ncol = FIRST ? 16 : Cin;
    float* __restrict__ o = a.part + ((int64_t)blockIdx.z * (FIRST ? 1 : 9) + (FIRST ? 0 : tap)) * Cout * ncol;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                o[(int64_t)(co0 + wy * WCO + m * 16 + g * 4 + r) * ncol + ci0 + wx * WCI + n * 16 + li] = acc[m][n][r];
}

// =====================================================================================================================
// k_conv_wgrad3 (bf16, Cin > 1): the weight-gradient GEMM on the loader of k_conv_igemm3 and gfx950's transposing LDS read.
// K = pixels is the strided axis of both NHWC operands.  A stage holds 32 pixels of dY and of the (tap-shifted) X as they lie in
// memory, cut into [32 pixels][16 channels] subtiles of 1 KiB - one LDS-direct wave instruction each (lane l brings half a row:
// pixel l >> 1, channels 8 (l & 1) .. +7).  ds_read_b64_tr_b16 then hands lane (i, kg) of a fragment the four pixels 4 kg .. 4 kg + 3
// of channel i (each lane of a 16-lane group addresses one 8-byte piece of a [4][16] block, the hardware transposes:
// tools/ubench/tr_read.hip); a second read 16 pixels further on fills K slots 4 .. 7.  Which pixel sits in which K slot is free as
// long as both operands agree - with (4 kg + j, 16 + 4 kg + j) every read sweeps a contiguous 512 bytes.  Two 8-byte reads per
// fragment instead of eight 2-byte gathers and their packing; ring, waits and barrier as in k_conv_igemm3.
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
template <int OFF> __device__ __forceinline__ u32x2 lds_read_tr8(uint32_t addr) {
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
template <int N, int I = 0> __device__ __forceinline__ void lds_read_tr_frags(u32x2 (&lo)[N], u32x2 (&hi)[N], uint32_t addr) {
    lo[I] = lds_read_tr8<I * 1024>(addr);
    hi[I] = lds_read_tr8<I * 1024 + 512>(addr);
    if constexpr (I + 1 < N) lds_read_tr_frags<N, I + 1>(lo, hi, addr);
}
template <int N> __device__ __forceinline__ void lds_frags_landed(u32x2 (&r)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("" : "+v"(r[i]));
}
template <int BCO, int BCI, int NS>
__global__ __launch_bounds__(256, 2) void k_conv_wgrad3(WgradArgs a) {
    constexpr int BKP = kWgradPix, WCO = BCO / 2, WCI = BCI / 2, MT = WCO / 16, NT = WCI / 16;
    constexpr int YG = BCO / 64, XG = BCI / 64, G = YG + XG;  // 1 KiB subtiles per lane and stage
    constexpr int SY = BCO * BKP, SX = BCI * BKP;             // elements per stage
    static_assert(BKP == 32 && NS >= 3 && NS <= 6, "stage shape");
    __shared__ __attribute__((aligned(1024))) bf16_t sY[NS * SY];
    __shared__ __attribute__((aligned(1024))) bf16_t sX[NS * SX];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wy = wave >> 1, wx = wave & 1;
    const int g = lane >> 4, li = lane & 15;
    const int H = a.H, W = a.W, Cin = a.Cin, Cout = a.Cout, HW = H * W;
    const int64_t P = (int64_t)a.N * HW;
    // 1-D grid, decoded so that the nine taps (and the channel tiles) of one pixel range run on ONE XCD at about the same time
    // (workgroup ids go round-robin over the 8 XCDs): with (tiles, 9, splits) grids the nine readers of a range of dY sat on nine
    // different L2s and the 64 -> 64 layer pulled 19 GB through HBM at 7 TB/s
    const int ci_tiles = Cin / BCI, tiles = ci_tiles * (Cout / BCO);
    const int xcd = blockIdx.x & 7, kseq = blockIdx.x >> 3;
    const int tap = kseq % 9, tile = (kseq / 9) % tiles, split = (kseq / (9 * tiles)) * 8 + xcd;
    if (split >= a.splits) return;
    const int co0 = (tile / ci_tiles) * BCO, ci0 = (tile % ci_tiles) * BCI;
    const int dh = tap / 3 - 1, dw = tap % 3 - 1;
    const bf16_t* __restrict__ dy = reinterpret_cast<const bf16_t*>(a.dy);
    const bf16_t* __restrict__ x = reinterpret_cast<const bf16_t*>(a.x);
    const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_conv_zero);
    const int64_t pstart = (int64_t)split * a.steps_per_split * BKP;

    // loader: this lane's pixel of the stage being issued; its position inside the image and its two source pointers are kept
    // incrementally (selects between two ready pointers: hipcc turns a select over a 64-bit multiply into divergent branches
    // with one exec-masked load each - harmless for the counted waits, which only need >= G loads per stage, but slower)
    int64_t ip = pstart + (lane >> 1);
    int ir = (int)(ip % HW);
    const bf16_t* pdy = dy + ip * Cout + co0 + 8 * (lane & 1);
    const bf16_t* px = x + (ip + (int64_t)dh * W + dw) * Cin + ci0 + 8 * (lane & 1);
    const int64_t dy_step = (int64_t)BKP * Cout, x_step = (int64_t)BKP * Cin;
    auto issue = [&](int stage) {
        const bool live = ip < P;
        const int h = ir / W, ww = ir - h * W;
        const bool xok = live && (unsigned)(h + dh) < (unsigned)H && (unsigned)(ww + dw) < (unsigned)W;
        const bf16_t* sdy = live ? pdy : zero;
        const bf16_t* sx = xok ? px : zero;
        const int dsub = live ? 16 : 0, xsub = xok ? 16 : 0;
#pragma unroll
        for (int q = 0; q < YG; ++q) {
            const int sub = q * 4 + wave;
            __builtin_amdgcn_global_load_lds((glb_void_t*)(sdy + dsub * sub), (lds_void_t*)(uintptr_t)(sY + stage * SY + sub * 512), 16, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < XG; ++q) {
            const int sub = q * 4 + wave;
            __builtin_amdgcn_global_load_lds((glb_void_t*)(sx + xsub * sub), (lds_void_t*)(uintptr_t)(sX + stage * SX + sub * 512), 16, 0, 0);
        }
        ip += BKP;
        pdy += dy_step;
        px += x_step;
        ir += BKP;
        while (ir >= HW) ir -= HW;
    };
    // fragment piece of lane (kg = g, t = li): row 4 g + (t >> 2), columns 4 (t & 3) .. +3 of its subtile (32-byte rows)
    const uint32_t piece = (uint32_t)((4 * g + (li >> 2)) * 32 + 8 * (li & 3));
    const uint32_t aaddr0 = (uint32_t)(uintptr_t)sY + (uint32_t)(wy * (WCO / 16) * 1024) + piece;
    const uint32_t baddr0 = (uint32_t)(uintptr_t)sX + (uint32_t)(wx * (WCI / 16) * 1024) + piece;
    f32x4 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int steps = a.steps_per_split;
#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
        if (s < steps) issue(s);
    int cur = 0;
    for (int s = 0; s < steps; ++s) {
        if (steps - 1 - s >= NS - 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * G) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (s + NS - 1 < steps) issue(cur == 0 ? NS - 1 : cur - 1);
        const uint32_t aaddr = aaddr0 + (uint32_t)(cur * SY * 2), baddr = baddr0 + (uint32_t)(cur * SX * 2);
        u32x2 alo[MT], ahi[MT], blo[NT], bhi[NT];
        lds_read_tr_frags<NT>(blo, bhi, baddr);
        lds_read_tr_frags<MT>(alo, ahi, aaddr);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        lds_frags_landed(blo);
        lds_frags_landed(bhi);
        lds_frags_landed(alo);
        lds_frags_landed(ahi);
        bf16x8 af[MT], bfr[NT];
#pragma unroll
        for (int m = 0; m < MT; ++m) af[m] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(alo[m], ahi[m], 0, 1, 2, 3));
#pragma unroll
        for (int n = 0; n < NT; ++n) bfr[n] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(blo[n], bhi[n], 0, 1, 2, 3));
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[m], bfr[n], acc[m][n], 0, 0, 0);
        cur = cur + 1 == NS ? 0 : cur + 1;
    }
    // D row = co (lane >> 4) * 4 + r, D column = ci lane & 15
    float* __restrict__ o = a.part + ((int64_t)split * 9 + tap) * Cout * Cin;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                o[(int64_t)(co0 + wy * WCO + m * 16 + g * 4 + r) * Cin + ci0 + wx * WCI + n * 16 + li] = acc[m][n][r];
}

// =====================================================================================================================
// k_conv_wgrad4 (bf16, 64 x 64 channel tiles, W >= 64): ALL NINE taps of a pixel range in one workgroup.  k_conv_wgrad3 runs one
// tap per workgroup, so every 32-pixel slab of dY and X crosses the L2 -> LDS path nine times (19 GB for the 64 -> 64 layer at
// ~12 TB/s: that path, not the matrix pipe, is its limit).  Here a stage holds the 32 pixels of dY once and, per vertical offset dh,
// one segment of X: rows j = 0 .. 33 <-> pixels pb - 1 + j (+ dh W).  Tap (dh, dw) reads its B fragments from segment dh shifted
// by dw + 1 rows (32 bytes: the transposing read's pieces stay 8-byte aligned and sweep a contiguous 512 bytes); nine accumulator
// sets of 2 x 2 MFMA tiles per wave.  17.5 KB per stage for 36 MFMAs per wave instead of 72 KB.
//   * vertical validity (row h + dh of the centre pixel's image) acts at load time (zero page);
//   * horizontal validity is per PIXEL = per K slot of a fragment: in the stages that contain a row end (wave-uniform test on the
//     column of pb; every 32nd stage at W = 1025) the fragments of the dw = -1 / +1 taps are ANDed with a per-lane mask.
// Segment subtiles are [34 rows][16 channels] on a 1152-byte pitch: one full LDS-direct instruction (rows 0 .. 31) and one with four
// active lanes (rows 32, 33).  Seven instructions per wave and stage, the same for every wave: counted waits as in k_conv_wgrad3.
constexpr int kWg4Sub = 1152;  // bytes per segment subtile
template <int NS>
__global__ __launch_bounds__(256, 2) void k_conv_wgrad4(WgradArgs a) {
    constexpr int BKP = kWgradPix, G = 7;
    constexpr int SYB = 4 * 1024, SEGB = 4 * kWg4Sub, SXB = 3 * SEGB;  // bytes per stage: dY, one segment, three segments
    static_assert(BKP == 32 && NS >= 3 && NS <= 4, "stage shape");
    __shared__ __attribute__((aligned(1024))) char sY[NS * SYB];
    __shared__ __attribute__((aligned(1024))) char sX[NS * SXB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wy = wave >> 1, wx = wave & 1;
    const int g = lane >> 4, li = lane & 15;
    const int H = a.H, W = a.W, Cin = a.Cin, Cout = a.Cout, HW = H * W;
    const int64_t P = (int64_t)a.N * HW;
    const int ci_tiles = Cin / 64, tiles = ci_tiles * (Cout / 64);
    const int xcd = blockIdx.x & 7, kseq = blockIdx.x >> 3;
    const int tile = kseq % tiles, split = (kseq / tiles) * 8 + xcd;
    if (split >= a.splits) return;
    const int co0 = (tile / ci_tiles) * 64, ci0 = (tile % ci_tiles) * 64;
    const bf16_t* __restrict__ dy = reinterpret_cast<const bf16_t*>(a.dy);
    const bf16_t* __restrict__ x = reinterpret_cast<const bf16_t*>(a.x);
    const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_conv_zero);
    const int64_t pstart = (int64_t)split * a.steps_per_split * BKP;

    // loader: wave w owns channel subtile w of dY and of every segment; lane l brings half a row (pixel row l >> 1, channels 8 (l & 1) ..)
    int64_t ip = pstart + (lane >> 1);                 // dY pixel of the stage being issued; segment row l >> 1 is pixel ip - 1
    int irc = (int)((ip - 1 + HW) % HW);               // position of pixel ip - 1 inside its image
    const bf16_t* pdy = dy + ip * Cout + co0 + 16 * wave + 8 * (lane & 1);
    const bf16_t* px = x + (ip - 1) * Cin + ci0 + 16 * wave + 8 * (lane & 1);
    const int64_t dy_step = (int64_t)BKP * Cout, x_step = (int64_t)BKP * Cin, x_tail = (int64_t)32 * Cin, x_row = (int64_t)W * Cin;
    auto issue = [&](int stage) {
        const bool live = ip < P;
        __builtin_amdgcn_global_load_lds((glb_void_t*)(live ? pdy : zero), (lds_void_t*)(uintptr_t)(sY + stage * SYB + wave * 1024), 16, 0, 0);
        // centre pixels: c = ip - 1 (rows 0 .. 31) and c + 32 (rows 32, 33: lanes 0 .. 3)
        const int64_t c = ip - 1;
        const bool okc = c >= 0 && c < P, okt = c + 32 < P;  // c + 32 >= 31
        const int h = irc / W;
        int irt = irc + 32;
        while (irt >= HW) irt -= HW;
        const int ht = irt / W;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const bool v = okc && (unsigned)(h + d - 1) < (unsigned)H;
            const bf16_t* src = v ? px + (d - 1) * x_row : zero;
            __builtin_amdgcn_global_load_lds((glb_void_t*)src, (lds_void_t*)(uintptr_t)(sX + stage * SXB + d * SEGB + wave * kWg4Sub), 16, 0, 0);
        }
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const bool v = okt && (unsigned)(ht + d - 1) < (unsigned)H;
            const bf16_t* src = v ? px + x_tail + (d - 1) * x_row : zero;
            if (lane < 4)
                __builtin_amdgcn_global_load_lds((glb_void_t*)src, (lds_void_t*)(uintptr_t)(sX + stage * SXB + d * SEGB + wave * kWg4Sub + 1024), 16, 0, 0);
        }
        ip += BKP;
        pdy += dy_step;
        px += x_step;
        irc += BKP;
        while (irc >= HW) irc -= HW;
    };
    // fragment piece of lane (kg = g, t = li): row 4 g + (t >> 2), columns 4 (t & 3) .. +3 of its subtile (32-byte rows)
    const uint32_t piece = (uint32_t)((4 * g + (li >> 2)) * 32 + 8 * (li & 3));
    const uint32_t aaddr0 = (uint32_t)(uintptr_t)sY + (uint32_t)(wy * 2 * 1024) + piece;
    const uint32_t baddr0 = (uint32_t)(uintptr_t)sX + (uint32_t)(wx * 2 * kWg4Sub) + piece;
    f32x4 acc[9][2][2];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n) acc[t][m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int steps = a.steps_per_split;
    int wp = (int)(pstart % HW) % W;  // column of the stage's first pixel (wave-uniform)
#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
        if (s < steps) issue(s);
    int cur = 0;
    for (int s = 0; s < steps; ++s) {
        if (steps - 1 - s >= NS - 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * G) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (s + NS - 1 < steps) issue(cur == 0 ? NS - 1 : cur - 1);
        const uint32_t aaddr = aaddr0 + (uint32_t)(cur * SYB), baddr = baddr0 + (uint32_t)(cur * SXB);
        // K slot s of this lane's fragments <-> pixel offset 4 g + (s & 3) + 16 (s >> 2); masks of the slots whose left / right
        // neighbour lies in another image row (only in stages that contain a row end)
        const bool edge = wp == 0 || wp + 31 >= W - 1;
        uint32_t mL[4] = {~0u, ~0u, ~0u, ~0u}, mR[4] = {~0u, ~0u, ~0u, ~0u};
        if (edge) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                int col = wp + 4 * g + (q & 3) + 16 * (q >> 2);
                if (col >= W) col -= W;
                const uint32_t keep = (q & 1) ? 0x0000ffffu : 0xffff0000u;  // what stays of the dword when slot q goes
                if (col == 0) mL[q >> 1] &= keep;
                if (col == W - 1) mR[q >> 1] &= keep;
            }
        }
        u32x2 alo[2], ahi[2];
        lds_read_tr_frags<2>(alo, ahi, aaddr);
#pragma unroll
        for (int d = 0; d < 3; ++d) {       // vertical offset dh = d - 1: one segment
            u32x2 blo[3][2], bhi[3][2];
#pragma unroll
            for (int e = 0; e < 3; ++e) {   // horizontal offset dw = e - 1: rows shifted by e
                blo[e][0] = lds_read_tr8<0>(baddr + (uint32_t)(d * SEGB + e * 32));
                bhi[e][0] = lds_read_tr8<512>(baddr + (uint32_t)(d * SEGB + e * 32));
                blo[e][1] = lds_read_tr8<kWg4Sub>(baddr + (uint32_t)(d * SEGB + e * 32));
                bhi[e][1] = lds_read_tr8<kWg4Sub + 512>(baddr + (uint32_t)(d * SEGB + e * 32));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (d == 0) {
                lds_frags_landed(alo);
                lds_frags_landed(ahi);
            }
#pragma unroll
            for (int e = 0; e < 3; ++e) {
                lds_frags_landed(blo[e]);
                lds_frags_landed(bhi[e]);
            }
            if (edge) {
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    blo[0][n][0] &= mL[0]; blo[0][n][1] &= mL[1]; bhi[0][n][0] &= mL[2]; bhi[0][n][1] &= mL[3];
                    blo[2][n][0] &= mR[0]; blo[2][n][1] &= mR[1]; bhi[2][n][0] &= mR[2]; bhi[2][n][1] &= mR[3];
                }
            }
#pragma unroll
            for (int e = 0; e < 3; ++e)
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n)
                        acc[d * 3 + e][m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            __builtin_bit_cast(bf16x8, __builtin_shufflevector(alo[m], ahi[m], 0, 1, 2, 3)),
                            __builtin_bit_cast(bf16x8, __builtin_shufflevector(blo[e][n], bhi[e][n], 0, 1, 2, 3)), acc[d * 3 + e][m][n], 0, 0, 0);
        }
        cur = cur + 1 == NS ? 0 : cur + 1;
        wp += BKP;
        if (wp >= W) wp -= W;
    }
    // D row = co (lane >> 4) * 4 + r, D column = ci lane & 15
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        float* __restrict__ o = a.part + ((int64_t)split * 9 + t) * Cout * Cin;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    o[(int64_t)(co0 + wy * 32 + m * 16 + g * 4 + r) * Cin + ci0 + wx * 32 + n * 16 + li] = acc[t][m][n][r];
    }
}
bool conv_wgrad_nine_taps(int precision, int W, int Cin, int Cout) {
#ifdef MST_CONV_NO_WGRAD4
    return false;  // A/B switch
#else
    return precision == 0 && conv3_enabled() && W >= 64 && Cin % 64 == 0 && Cout % 64 == 0;
#endif
}

void launch_conv_wgrad(int precision, const WgradArgs& a, hipStream_t s) {
    if (conv_wgrad_nine_taps(precision, a.W, a.Cin, a.Cout)) {
        const dim3 grid4((a.Cout / 64) * (a.Cin / 64) * ((a.splits + 7) / 8) * 8);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_conv_wgrad4<3>), grid4, dim3(256), 0, s, a);
        return;
    }
    if (a.Cin == 1) {
        const dim3 grid(a.Cout / 64, 1, a.splits);
        if (precision == 0) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_conv_wgrad<bf16_t, 64, 16, true>), grid, dim3(256), 0, s, a);
        else if (precision == 2) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_conv_wgrad<float, 64, 16, true, 3>), grid, dim3(256), 0, s, a);
        else if (precision == 3) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_conv_wgrad<float, 64, 16, true, 6>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_conv_wgrad<float, 64, 16, true>), grid, dim3(256), 0, s, a);
    } else if (a.Cin % 128 == 0 && a.Cout % 128 == 0) {
        const dim3 grid((a.Cout / 128) * (a.Cin / 128), 9, a.splits);
        const dim3 grid3((a.Cout / 128) * (a.Cin / 128) * 9 * ((a.splits + 7) / 8) * 8);
        if (precision == 0 && conv3_enabled()) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_conv_wgrad3<128, 128, MST_CONV_GLDS_STAGES>), grid3, dim3(256), 0, s, a);
        else if (precision == 0) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_conv_wgrad<bf16_t, 128, 128, false>), grid, dim3(256), 0, s, a);
        else if (precision == 2) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_conv_wgrad<float, 128, 128, false, 3>), grid, dim3(256), 0, s, a);
        else if (precision == 3) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_conv_wgrad<float, 128, 128, false, 6>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_conv_wgrad<float, 128, 128, false>), grid, dim3(256), 0, s, a);
    } else {
        const dim3 grid((a.Cout / 64) * (a.Cin / 64), 9, a.splits);
        const dim3 grid3((a.Cout / 64) * (a.Cin / 64) * 9 * ((a.splits + 7) / 8) * 8);
        if (precision == 0 && conv3_enabled()) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_conv_wgrad3<64, 64, MST_CONV_GLDS_STAGES>), grid3, dim3(256), 0, s, a);
        else if (precision == 0) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_conv_wgrad<bf16_t, 64, 64, false>), grid, dim3(256), 0, s, a);
        else if (precision == 2) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_conv_wgrad<float, 64, 64, false, 3>), grid, dim3(256), 0, s, a);
        else if (precision == 3) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_conv_wgrad<float, 64, 64, false, 6>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_conv_wgrad<float, 64, 64, false>), grid, dim3(256), 0, s, a);
    }
}

}  // namespace mst
