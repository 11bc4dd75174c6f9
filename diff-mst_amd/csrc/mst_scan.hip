// mst_scan.hip - carry scans that turn per-chunk zero-state end states into true start states:
//      s0[c+1] = M s0[c] + z[c]      (forward)       s0[c-1] = M s0[c] + z[c]   (reverse)
// with a constant D x D matrix per row (D = 12 cascade, D = 2 all-pole, D = 1 envelope smoother).
// One 512-lane workgroup per row: each lane folds K consecutive chunks sequentially, a
// Hillis-Steele scan over the 512 lane aggregates uses the precomputed powers M^(K 2^j)
// (staged in LDS, every read a broadcast), then each lane replays its K chunks from its true start.
// The EQ cascades of rows up to 262144 samples do NOT come here: their carries are scanned inside the
// zs / run kernels (mst_eq.hip, SCAN1); this kernel serves the all-pole bank and longer rows.
#include "mst_kernels.h"
#include "mst_mat.h"

namespace mst {

// the single-chunk matrix kept in registers for the sequential fold / replay loops
template <int D>
struct RegMat {
    float m[D][D];
    __device__ __forceinline__ void load(const float* M) {
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int c = 0; c < D; ++c) m[i][c] = (c < row_cols<D>(i)) ? M[i * D + c] : 0.0f;
    }
    __device__ __forceinline__ void acc(const float* v, float* out) const {
#pragma unroll
        for (int i = 0; i < D; ++i) {
            float s = out[i];
#pragma unroll
            for (int c = 0; c < row_cols<D>(i); ++c) s = fmaf(m[i][c], v[c], s);
            out[i] = s;
        }
    }
};

// Eight consecutive chunk states (scan positions i0 .. i0+7 of a row; memory order is descending when REVERSE)
// <-> registers, with 16-byte accesses whenever the span is whole and aligned.
template <int D, bool REVERSE>
__device__ __forceinline__ void span8_load(const float* __restrict__ zr, int nc, int nc_pad, int i0, float (*zl)[D]) {
    const int lo = REVERSE ? nc - 8 - i0 : i0;  // lowest chunk index of the span
    if (lo >= 0 && lo + 8 <= nc && (lo & 3) == 0) {
#pragma unroll
        for (int d = 0; d < D; ++d)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const float4 v = *reinterpret_cast<const float4*>(zr + (int64_t)d * nc_pad + lo + 4 * q);
                const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int t = 0; t < 4; ++t) zl[REVERSE ? 7 - (4 * q + t) : 4 * q + t][d] = e[t];
            }
    } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int i = i0 + k, c = REVERSE ? nc - 1 - i : i;
#pragma unroll
            for (int d = 0; d < D; ++d) zl[k][d] = (i < nc) ? zr[(int64_t)d * nc_pad + c] : 0.0f;
        }
    }
}
template <int D, bool REVERSE>
__device__ __forceinline__ void span8_store(float* __restrict__ sr, int nc, int nc_pad, int i0, int count, float (*so)[D]) {
    const int lo = REVERSE ? nc - 8 - i0 : i0;
    if (count == 8 && lo >= 0 && lo + 8 <= nc && (lo & 3) == 0) {
#pragma unroll
        for (int d = 0; d < D; ++d)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                float e[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) e[t] = so[REVERSE ? 7 - (4 * q + t) : 4 * q + t][d];
                *reinterpret_cast<float4*>(sr + (int64_t)d * nc_pad + lo + 4 * q) = make_float4(e[0], e[1], e[2], e[3]);
            }
    } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int i = i0 + k, c = REVERSE ? nc - 1 - i : i;
            if (k < count && i < nc) {
#pragma unroll
                for (int d = 0; d < D; ++d) sr[(int64_t)d * nc_pad + c] = so[k][d];
            }
        }
    }
}

// z, s0: [row][D][nc_pad].  tab: [table_row][kPow][D*D]; row = sig * sub + f, table_row = filter_row(sig, split) * sub + f
// KT > 0: K == KT known at compile time - all of a lane's chunk states are fetched up front (one
// memory round trip instead of K dependent ones) and the fold / replay loops are unrolled.
template <int D, bool REVERSE, int KT>
__global__ __launch_bounds__(kScanThreads) void k_scan(const float* __restrict__ z, float* __restrict__ s0,
                                                       const float* __restrict__ tab, int sub, int split,
                                                       int nc, int nc_pad, int Krt) {
    __shared__ float buf[2][D][kScanThreads];
    __shared__ __attribute__((aligned(16))) float T[kPow * D * D];  // this row's power table, staged once
    const int tid = threadIdx.x, row = blockIdx.x;
    const int K = KT > 0 ? KT : Krt;
    {
        const float* Tg = tab + ((int64_t)filter_row(row / sub, split) * sub + (row % sub)) * kPow * D * D;
        for (int i = tid; i < kPow * D * D; i += kScanThreads) T[i] = Tg[i];
    }
    const float* zr = z + (int64_t)row * D * nc_pad;
    float* sr = s0 + (int64_t)row * D * nc_pad;
    auto cidx = [&](int i) { return REVERSE ? nc - 1 - i : i; };

    // 1. fold my K chunks from zero
    constexpr int KL = KT > 0 ? KT : 1;
    float zl[KL][D];
    // A lane's KT chunks are contiguous in memory (ascending, or descending when REVERSE): fetch them
    // with 16-byte accesses whenever the span is whole and aligned (it is for every production length).
    const int span0 = REVERSE ? nc - KT * (tid + 1) : KT * tid;  // lowest chunk index of my span
    const bool vec = (KT % 4 == 0) && span0 >= 0 && span0 + KT <= nc && (span0 & 3) == 0;
    if (KT > 0) {
        if (vec) {
#pragma unroll
            for (int d = 0; d < D; ++d)
#pragma unroll
                for (int q = 0; q < KL / 4; ++q) {
                    const float4 v = *reinterpret_cast<const float4*>(zr + (int64_t)d * nc_pad + span0 + 4 * q);
                    const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int m = 4 * q + t;                      // memory order inside the span
                        zl[REVERSE ? KL - 1 - m : m][d] = e[t];       // scan order
                    }
                }
        } else {
#pragma unroll
            for (int k = 0; k < KL; ++k) {
                const int i = tid * KT + k;
#pragma unroll
                for (int d = 0; d < D; ++d) zl[k][d] = (i < nc) ? zr[(int64_t)d * nc_pad + cidx(i)] : 0.0f;
            }
        }
    }
    lds_barrier();  // table staged
    RegMat<D> M1;
    M1.load(T);
    float agg[D];
#pragma unroll
    for (int d = 0; d < D; ++d) agg[d] = 0.0f;
    if (KT > 0) {
#pragma unroll
        for (int k = 0; k < KL; ++k) {
            float nv[D];
#pragma unroll
            for (int d = 0; d < D; ++d) nv[d] = zl[k][d];
            M1.acc(agg, nv);
#pragma unroll
            for (int d = 0; d < D; ++d) agg[d] = nv[d];
        }
    } else {
        // any K: sub-spans of eight chunks, each fetched in one go (a lane's K chunks are contiguous)
        for (int k0 = 0; k0 < K; k0 += 8) {
            float z8[8][D];
            span8_load<D, REVERSE>(zr, nc, nc_pad, tid * K + k0, z8);
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                if (k0 + kk < K) {
                    float nv[D];
#pragma unroll
                    for (int d = 0; d < D; ++d) nv[d] = z8[kk][d];
                    M1.acc(agg, nv);
#pragma unroll
                    for (int d = 0; d < D; ++d) agg[d] = nv[d];
                }
            }
        }
    }
    // 2. inclusive Hillis-Steele over lanes (levels beyond the populated lanes are skipped)
    const int active = (nc + K - 1) / K;
    int cur = 0;
#pragma unroll
    for (int d = 0; d < D; ++d) buf[0][d][tid] = agg[d];
    lds_barrier();
    for (int j = 0; j < kScanLevels && (1 << j) < active; ++j) {
        const int off = 1 << j;
        if (tid >= off) {
            float o[D];
#pragma unroll
            for (int d = 0; d < D; ++d) o[d] = buf[cur][d][tid - off];
            matvec_acc<D>(T + (1 + j) * D * D, o, agg);
        }
#pragma unroll
        for (int d = 0; d < D; ++d) buf[cur ^ 1][d][tid] = agg[d];
        lds_barrier();
        cur ^= 1;
    }
    // 3. exclusive start of my span = inclusive value of the previous lane
    float st[D];
#pragma unroll
    for (int d = 0; d < D; ++d) st[d] = (tid > 0) ? buf[cur][d][tid - 1] : 0.0f;
    if (KT > 0) {
        float so[KL][D];
#pragma unroll
        for (int k = 0; k < KL; ++k) {
#pragma unroll
            for (int d = 0; d < D; ++d) so[k][d] = st[d];
            float nv[D];
#pragma unroll
            for (int d = 0; d < D; ++d) nv[d] = zl[k][d];
            M1.acc(st, nv);
#pragma unroll
            for (int d = 0; d < D; ++d) st[d] = nv[d];
        }
        if (vec) {
#pragma unroll
            for (int d = 0; d < D; ++d)
#pragma unroll
                for (int q = 0; q < KL / 4; ++q) {
                    float e[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int m = 4 * q + t;
                        e[t] = so[REVERSE ? KL - 1 - m : m][d];
                    }
                    *reinterpret_cast<float4*>(sr + (int64_t)d * nc_pad + span0 + 4 * q) = make_float4(e[0], e[1], e[2], e[3]);
                }
        } else {
#pragma unroll
            for (int k = 0; k < KL; ++k) {
                const int i = tid * KT + k;
                if (i < nc) {
#pragma unroll
                    for (int d = 0; d < D; ++d) sr[(int64_t)d * nc_pad + cidx(i)] = so[k][d];
                }
            }
        }
    } else {
        for (int k0 = 0; k0 < K; k0 += 8) {
            if (tid * K + k0 >= nc) break;
            float z8[8][D], so8[8][D];
            span8_load<D, REVERSE>(zr, nc, nc_pad, tid * K + k0, z8);
            const int count = K - k0 < 8 ? K - k0 : 8;
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
#pragma unroll
                for (int d = 0; d < D; ++d) so8[kk][d] = st[d];
                if (kk < count) {
                    float nv[D];
#pragma unroll
                    for (int d = 0; d < D; ++d) nv[d] = z8[kk][d];
                    M1.acc(st, nv);
#pragma unroll
                    for (int d = 0; d < D; ++d) st[d] = nv[d];
                }
            }
            span8_store<D, REVERSE>(sr, nc, nc_pad, tid * K + k0, count, so8);
        }
    }
}

template <int D, bool REV>
static void launch_scan_k(dim3 grid, hipStream_t stream, const float* z, float* s0, const float* tab, int sub, int split,
                          int nc, int nc_pad, int K) {
    const dim3 block(kScanThreads);
    if (K == 1) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_scan<D, REV, 1>), grid, block, 0, stream, z, s0, tab, sub, split, nc, nc_pad, K);
    else if (K == 2) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_scan<D, REV, 2>), grid, block, 0, stream, z, s0, tab, sub, split, nc, nc_pad, K);
    else if (K == 4) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_scan<D, REV, 4>), grid, block, 0, stream, z, s0, tab, sub, split, nc, nc_pad, K);
    else if (K == 8) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_scan<D, REV, 8>), grid, block, 0, stream, z, s0, tab, sub, split, nc, nc_pad, K);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_scan<D, REV, 0>), grid, block, 0, stream, z, s0, tab, sub, split, nc, nc_pad, K);
}
void launch_scan12(bool reverse, const float* z, float* s0, const float* tab, int split, int nc, int nc_pad, int K, int nsig,
                   hipStream_t stream) {
    if (reverse) launch_scan_k<12, true>(dim3(nsig), stream, z, s0, tab, 1, split, nc, nc_pad, K);
    else launch_scan_k<12, false>(dim3(nsig), stream, z, s0, tab, 1, split, nc, nc_pad, K);
}
void launch_scan2(const float* z, float* s0, const float* tab, int split, int nc, int nc_pad, int K, int nsig, hipStream_t stream) {
    launch_scan_k<2, false>(dim3(nsig * 12), stream, z, s0, tab, 12, split, nc, nc_pad, K);
}

}  // namespace mst
