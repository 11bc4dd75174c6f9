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

// Six 12x12 matrices (864 floats) of a scan table travel global -> registers -> LDS: the loads are issued
// early (4 x 16 B per lane of a 64-lane wave) and parked in LDS right before the scan that uses them, where
// every matrix read is a broadcast.  (Scalar loads of the same data measured 3-5x slower per scan level.)
constexpr int kTabFloats = 6 * 144;
struct TabRegs {
    float4 v[4];
};
__device__ __forceinline__ void tab_fetch(TabRegs& r, const float* __restrict__ g, int lane) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int q = lane + 64 * k;
        r.v[k] = q < kTabFloats / 4 ? *reinterpret_cast<const float4*>(g + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}
__device__ __forceinline__ void tab_stash(const TabRegs& r, float* __restrict__ lds, int lane) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int q = lane + 64 * k;
        if (q < kTabFloats / 4) *reinterpret_cast<float4*>(lds + 4 * q) = r.v[k];
    }
}

// In-place inclusive scan over the lanes of one wave of  s[pos] = P s[pos-1] + v[pos]  (pos = position in
// recurrence order; lane = pos, or 63 - pos when REV).  tab[j] = P^(2^j), wave-uniform.  Levels whose
// stride reaches `limit` (wave-uniform: number of populated positions) are skipped.
template <bool REV>
__device__ __forceinline__ void wave_scan12(float* v, const float* __restrict__ tab, int pos, int limit) {
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        if ((1 << j) >= limit) break;
        float o[kStates];
#pragma unroll
        for (int d = 0; d < kStates; ++d) o[d] = REV ? __shfl_down(v[d], 1u << j) : __shfl_up(v[d], 1u << j);
        if (pos >= (1 << j)) matvec_acc<kStates>(tab + j * 144, o, v);
    }
}
// v <- P^pos v for a per-lane exponent pos in [0, 64): six conditional applications of tab[j] = P^(2^j)
__device__ __forceinline__ void apply_pow12(float* v, const float* __restrict__ tab, int pos) {
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        float nv[kStates];
#pragma unroll
        for (int d = 0; d < kStates; ++d) nv[d] = 0.0f;
        matvec_acc<kStates>(tab + j * 144, v, nv);
        const bool take = (pos >> j) & 1;
#pragma unroll
        for (int d = 0; d < kStates; ++d) v[d] = take ? nv[d] : v[d];
    }
}

}  // namespace mst
