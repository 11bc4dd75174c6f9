// mst_mat.h - small dense mat-vec helpers shared by the carry scans (mst_scan.hip) and the
// single-pass cascade kernels (mst_eq1.hip).
#pragma once
#include "mst_common.h"

namespace mst {

// number of leading columns of row i that can be non-zero (cascade matrices are block lower-triangular)
template <int D>
__device__ __forceinline__ constexpr int row_cols(int i) { return D == 12 ? 2 * (i / 2 + 1) : D; }

// acc += M v ; M row-major D x D (16-byte aligned), read with explicit 16-byte loads.  The address is
// wave-uniform: from LDS every read is a broadcast, from global memory the compiler emits scalar loads.
template <int D>
__device__ __forceinline__ void matvec_acc(const float* M, const float* v, float* acc) {
    if (D == 2) {
        const float4 m = *reinterpret_cast<const float4*>(M);
        acc[0] = fmaf(m.x, v[0], fmaf(m.y, v[1], acc[0]));
        acc[1] = fmaf(m.z, v[0], fmaf(m.w, v[1], acc[1]));
        return;
    }
#pragma unroll
    for (int i = 0; i < D; ++i) {
        float s = acc[i];
#pragma unroll
        for (int c = 0; c < row_cols<D>(i); c += 4) {
            const float4 m = *reinterpret_cast<const float4*>(M + i * D + c);
            s = fmaf(m.x, v[c], s);
            s = fmaf(m.y, v[c + 1], s);
            if (c + 2 < row_cols<D>(i)) {
                s = fmaf(m.z, v[c + 2], s);
                s = fmaf(m.w, v[c + 3], s);
            }
        }
        acc[i] = s;
    }
}

// ---- in-wave carry scans of the 12-state cascade, section by section -------------------------------------------------
// The chunk transition P of a cascade is block lower-triangular in 2x2 blocks (section k sees sections j <= k only), so
//     x[pos] = P x[pos-1] + v[pos]
// splits into six 2-state scans run one after the other: section k scans  f = v_k + sum_{j<k} P_kj x_j[pos-1]  (x_j: the finished
// scan of an earlier section, one position back) with its own diagonal block D_k = P_kk:
//     x_k[pos] = D_k x_k[pos-1] + f[pos].
// Each 2-state scan runs on DPP data movement alone (no ds_bpermute: an ablation build without the scans was 31 us per step
// faster while they used 82 shuffles each): a Kogge-Stone pass inside every 16-lane row (row shifts by 1, 2, 4, 8 with the
// wave-uniform D^(2^j)), the three row totals fetched with v_readlane and chained with D^16 into wave-uniform carries, and one
// lane-dependent fix-up  x += D^(pin+1) carry  (pin = position inside the row).  "One position back" is a wave shift by 1.
// One table set (kTriFloats floats, staged through LDS):
//   D[k][j][4]   at (6 k + j) 4                    D_k^(2^j), j = 0..5, row-major 2x2 (j <= 4 used)
//   C[k][jj][4]  at 144 + (k (k-1) / 2 + jj) 4     P_k,jj for jj < k
//   Dp[k][q][4]  at 208 + (16 k + q) 4             D_k^(q+1), q = 0..15
constexpr int kTriFloats = 592;
#ifndef MST_SCAN_PACKED
#define MST_SCAN_PACKED 1  // the LDS image of the tables holds every 2x2 block column-major (wave_scan_tri below)
#endif
struct TabRegs {
    float4 v[3];
};
__device__ __forceinline__ void tab_fetch(TabRegs& r, const float* __restrict__ g, int lane) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int q = lane + 64 * k;
        r.v[k] = q < kTriFloats / 4 ? *reinterpret_cast<const float4*>(g + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}
__device__ __forceinline__ void tab_stash(const TabRegs& r, float* __restrict__ lds, int lane) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int q = lane + 64 * k;
        if (q < kTriFloats / 4) *reinterpret_cast<float4*>(lds + 4 * q) = MST_SCAN_PACKED ? make_float4(r.v[k].x, r.v[k].z, r.v[k].y, r.v[k].w) : r.v[k];
    }
}
// value of the lane CTRL points at; 0 where that lane does not exist (row / wave edge)
template <int CTRL>
__device__ __forceinline__ float dpp_get(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float lane_get(float v, int lane) {  // wave-uniform
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
// In-place inclusive scan over the 64 lanes of one wave (pos = position in recurrence order; lane = pos, or 63 - pos when REV).
// Positions that hold no data must carry zeros (they only receive).
// The LDS image holds every 2x2 block COLUMN-major (tab_stash swaps the middle components on the way in): (m00, m10) and (m01, m11) are
// register pairs, so  F += M o  is two packed multiply-adds (v_pk_fma_f32) on the state pair F = (f0, f1) - the same roundings, in
// the same order, as the four scalar ones.
using scan_f2 = f2;
struct Mat2c { scan_f2 c0, c1; };  // columns
__device__ __forceinline__ Mat2c mat_at(const float* __restrict__ tab, int at) {
    const float4 m = *reinterpret_cast<const float4*>(tab + at);
    if (MST_SCAN_PACKED) return Mat2c{scan_f2{m.x, m.y}, scan_f2{m.z, m.w}};
    return Mat2c{scan_f2{m.x, m.z}, scan_f2{m.y, m.w}};  // row-major image
}
__device__ __forceinline__ scan_f2 mat_acc(const Mat2c& m, float o0, float o1, scan_f2 f) {  // f + M (o0, o1)
    return f2_fma(m.c0, scan_f2{o0, o0}, f2_fma(m.c1, scan_f2{o1, o1}, f));
}
template <bool REV>
__device__ __forceinline__ void wave_scan_tri(float* v, const float* __restrict__ tab, int pos) {
    // one position back / d positions back inside the row: towards lower pos = lower lanes, or higher lanes when REV
    constexpr int kBack1 = REV ? 0x130 : 0x138;   // wave_shl:1 / wave_shr:1
    constexpr int kRow = REV ? 0x100 : 0x110;     // row_shl:d / row_shr:d
    const int prow = pos >> 4, pin = pos & 15;
    float prev[kStates];
#pragma unroll
    for (int k = 0; k < kSections; ++k) {
        scan_f2 f = {v[2 * k], v[2 * k + 1]};
#pragma unroll
        for (int jj = 0; jj < k; ++jj) f = mat_acc(mat_at(tab, 144 + (k * (k - 1) / 2 + jj) * 4), prev[2 * jj], prev[2 * jj + 1], f);
        // inside the rows
        f = mat_acc(mat_at(tab, (6 * k + 0) * 4), dpp_get<kRow + 1>(f.x), dpp_get<kRow + 1>(f.y), f);
        f = mat_acc(mat_at(tab, (6 * k + 1) * 4), dpp_get<kRow + 2>(f.x), dpp_get<kRow + 2>(f.y), f);
        f = mat_acc(mat_at(tab, (6 * k + 2) * 4), dpp_get<kRow + 4>(f.x), dpp_get<kRow + 4>(f.y), f);
        f = mat_acc(mat_at(tab, (6 * k + 3) * 4), dpp_get<kRow + 8>(f.x), dpp_get<kRow + 8>(f.y), f);
        // across the rows: E_r = value at the last position of row r with everything before it included (wave-uniform)
        const Mat2c d16 = mat_at(tab, (6 * k + 4) * 4);
        const float e00 = lane_get(f.x, REV ? 48 : 15), e01 = lane_get(f.y, REV ? 48 : 15);
        const float r10 = lane_get(f.x, REV ? 32 : 31), r11 = lane_get(f.y, REV ? 32 : 31);
        const float r20 = lane_get(f.x, REV ? 16 : 47), r21 = lane_get(f.y, REV ? 16 : 47);
        const scan_f2 e1 = mat_acc(d16, e00, e01, scan_f2{r10, r11});
        const scan_f2 e2 = mat_acc(d16, e1.x, e1.y, scan_f2{r20, r21});
        const float c0 = prow == 1 ? e00 : (prow == 2 ? e1.x : e2.x), c1 = prow == 1 ? e01 : (prow == 2 ? e1.y : e2.y);
        const Mat2c dp = mat_at(tab, 208 + (16 * k + pin) * 4);
        if (prow >= 1) f = mat_acc(dp, c0, c1, f);
        v[2 * k] = f.x;
        v[2 * k + 1] = f.y;
        if (k + 1 < kSections) {
            prev[2 * k] = dpp_get<kBack1>(f.x);
            prev[2 * k + 1] = dpp_get<kBack1>(f.y);
        }
    }
}

}  // namespace mst
