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
// 204 FMAs and 82 shuffles per scan instead of six 12x12 triangular mat-vecs (504 FMAs, 144 16-byte LDS reads, 72 shuffles);
// measured at cfg #2: the scans were 46 us of the step (ablation MST_DBG_NOSCAN).
// One table set (kTriFloats floats, wave-uniform, staged through LDS):
//   D[k][j][4]   at (6 k + j) 4          D_k^(2^j), j = 0..5, row-major 2x2
//   C[k][jj][4]  at 144 + (k (k-1) / 2 + jj) 4     P_k,jj for jj < k
constexpr int kTriFloats = 208;  // 204 used, padded to whole 16-byte vectors
struct TabRegs {
    float4 v;
};
__device__ __forceinline__ void tab_fetch(TabRegs& r, const float* __restrict__ g, int lane) {
    r.v = lane < kTriFloats / 4 ? *reinterpret_cast<const float4*>(g + 4 * lane) : make_float4(0.f, 0.f, 0.f, 0.f);
}
__device__ __forceinline__ void tab_stash(const TabRegs& r, float* __restrict__ lds, int lane) {
    if (lane < kTriFloats / 4) *reinterpret_cast<float4*>(lds + 4 * lane) = r.v;
}
// In-place inclusive scan over the lanes of one wave (pos = position in recurrence order; lane = pos, or 63 - pos when REV).
// Levels whose stride reaches `limit` (wave-uniform: number of populated positions) are skipped.
template <bool REV>
__device__ __forceinline__ void wave_scan_tri(float* v, const float* __restrict__ tab, int pos, int limit) {
    float prev[kStates];
#pragma unroll
    for (int k = 0; k < kSections; ++k) {
        float f0 = v[2 * k], f1 = v[2 * k + 1];
#pragma unroll
        for (int jj = 0; jj < k; ++jj) {
            const float4 c = *reinterpret_cast<const float4*>(tab + 144 + (k * (k - 1) / 2 + jj) * 4);
            f0 = fmaf(c.x, prev[2 * jj], fmaf(c.y, prev[2 * jj + 1], f0));
            f1 = fmaf(c.z, prev[2 * jj], fmaf(c.w, prev[2 * jj + 1], f1));
        }
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            if ((1 << j) >= limit) break;
            const float o0 = REV ? __shfl_down(f0, 1u << j) : __shfl_up(f0, 1u << j);
            const float o1 = REV ? __shfl_down(f1, 1u << j) : __shfl_up(f1, 1u << j);
            if (pos >= (1 << j)) {
                const float4 d = *reinterpret_cast<const float4*>(tab + (6 * k + j) * 4);
                f0 = fmaf(d.x, o0, fmaf(d.y, o1, f0));
                f1 = fmaf(d.z, o0, fmaf(d.w, o1, f1));
            }
        }
        v[2 * k] = f0;
        v[2 * k + 1] = f1;
        if (k + 1 < kSections) {
            const float p0 = REV ? __shfl_down(f0, 1u) : __shfl_up(f0, 1u);
            const float p1 = REV ? __shfl_down(f1, 1u) : __shfl_up(f1, 1u);
            prev[2 * k] = pos >= 1 ? p0 : 0.0f;
            prev[2 * k + 1] = pos >= 1 ? p1 : 0.0f;
        }
    }
}

}  // namespace mst
