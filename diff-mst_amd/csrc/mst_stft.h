// mst_stft.h - argument blocks shared by the two translation units of the spectrogram loss:
//   mst_stft.hip   C ABI, tables, loss reduction, the round-1 radix-4 LDS transform kernels (any power-of-two n_fft,
//                  any hop / window length - the generic path)
//   mst_stft2.hip  the round-2 kernels on the register-radix engine (mst_fft2.h) for the reference's shape of
//                  resolution (n_fft in {512, 2048, 8192}, hop = n_fft / 2, full-length window); compiled with
//                  -fno-slp-vectorize (packed-fp32 pairing costs this code 40 % more registers and 60 v_mov per frame)
#pragma once
#include "mst_common.h"

namespace mst {

constexpr int kMaxRes = 8;

struct ResInfo {
    int n_fft, hop, n_frames, n_bins;
    int64_t tw_off, win_off;  // float offsets into the tables buffer (tw: n_fft float2, win: n_fft floats)
    int frames_per_wg;                  // forward strip length
};

// ---- reduction of the partial sums + loss + backward coefficients ---------------------------------
struct LossArgs {
    const float* part;   // concatenated per resolution: (rows, n_groups[res], 4)
    float* sums;         // (n_res, rows, 4)
    float* coef;         // (n_res, rows, 4)
    float* loss;         // scalar out
    int n_res, rows;
    int n_groups[kMaxRes];
    int64_t part_off[kMaxRes];
    float count[kMaxRes];  // rows * n_bins * n_frames
    float w_sc, w_log, w_lin;
    int sc_per_example;
    // sharded evaluation (rows of the batch split over ranks): per-resolution totals of THIS rank's rows out (k_mrstft_totals),
    // all-reduced totals in (k_mrstft_final) for the batch-global spectral-convergence ratio
    double* totals;         // (n_res, 4) out, or null
    const double* gtotals;  // (n_res, 4) in, or null
    int world;              // ranks that contributed to gtotals (1 when null)
};
// fold of one row's strip partials of one resolution (fp64, fixed order) by one wave; lane = 0..63
// AGENT: the four sums are written as two 8-byte agent-scope (write-through) stores - the publishing half of the fence-free hand-over to
// the last-arriving workgroup of k_mrstft_finish (mst_stft.hip); later LAUNCHES read them with plain loads either way
template <bool AGENT = false>
__device__ __forceinline__ void mrstft_rowsum(const LossArgs& a, int row, int res, int tid) {
    const float* p = a.part + a.part_off[res] + (int64_t)row * a.n_groups[res] * 4;
    double s[4] = {0, 0, 0, 0};
    const int ng = a.n_groups[res];
    for (int g0 = tid; g0 < ng; g0 += 256) {  // four strips in flight per lane, folded in ascending order
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int g = g0 + 64 * u;
            v[u] = g < ng ? *reinterpret_cast<const float4*>(p + (int64_t)g * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (g0 + 64 * u < ng) { s[0] += v[u].x; s[1] += v[u].y; s[2] += v[u].z; s[3] += v[u].w; }
    }
    // (six DPP additions per value; the butterfly of __shfl_xor it replaces was 48 ds_bpermute round trips, ~2 us of k_mrstft_finish)
#pragma unroll
    for (int q = 0; q < 4; ++q) s[q] = wave_sum_f64(s[q]);
    if (tid == 0) {
        float* o = a.sums + ((int64_t)res * a.rows + row) * 4;
        if (AGENT) {
            unsigned long long* o8 = reinterpret_cast<unsigned long long*>(o);
            const unsigned long long lo = ((unsigned long long)(unsigned)__float_as_int((float)s[1]) << 32) | (unsigned)__float_as_int((float)s[0]);
            const unsigned long long hi = ((unsigned long long)(unsigned)__float_as_int((float)s[3]) << 32) | (unsigned)__float_as_int((float)s[2]);
            __hip_atomic_store(o8, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(o8 + 1, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            o[0] = (float)s[0]; o[1] = (float)s[1]; o[2] = (float)s[2]; o[3] = (float)s[3];
        }
    }
}
// one (resolution, row) entry of the row sums; AGENT: with 8-byte agent-scope loads (see mrstft_rowsum)
template <bool AGENT>
__device__ __forceinline__ float4 mrstft_load_sums(const float* sums, int64_t i) {
    if (!AGENT) return *reinterpret_cast<const float4*>(sums + i * 4);
    const unsigned long long* p = reinterpret_cast<const unsigned long long*>(sums + i * 4);
    const unsigned long long lo = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long hi = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_float4(__int_as_float((int)(unsigned)lo), __int_as_float((int)(unsigned)(lo >> 32)), __int_as_float((int)(unsigned)hi),
                       __int_as_float((int)(unsigned)(hi >> 32)));
}
__device__ __forceinline__ void write_coef(const LossArgs& a, int i, int res, double c_sc) {
    float* c = a.coef + (int64_t)i * 4;
    c[0] = (float)(c_sc / a.n_res);
    c[1] = (float)(a.w_log / a.count[res] / a.n_res);
    c[2] = (float)(a.w_lin / a.count[res] / a.n_res);
    c[3] = 0.f;
}
// loss scalar + per-row backward coefficients by ONE wave (tid = 0..63; BLOCK_SYNC: that wave is the whole workgroup and the
// phases are separated by __syncthreads, otherwise by the wave's own LDS ordering).  rs / ratio / srow: LDS scratch; can_stage:
// ratio and srow hold kMaxRes * 64 entries each.
template <bool BLOCK_SYNC = true, bool AGENT = false>
__device__ __forceinline__ void mrstft_final_body(const LossArgs& a, int tid, double (*rs)[4], double* ratio, float4* srow, bool can_stage) {
    auto sync = [] {
        if (BLOCK_SYNC) __syncthreads();
        else wave_lds_sync();
    };
    // A lane per resolution walking its rows paid one L2 round trip per row plus a ~150-instruction fp64 sqrt / sqrt / divide chain
    // (7.2 us for 16 rows).  One lane per (resolution, row) fetches and takes the roots side by side; the fold over the rows below
    // reads LDS and keeps its fixed order (bitwise the same loss).
    const bool staged = can_stage && a.n_res * a.rows <= kMaxRes * 64;
    for (int i = tid; i < a.n_res * a.rows; i += 64) {
        const float4 sm = mrstft_load_sums<AGENT>(a.sums, i);
        if (staged) srow[i] = sm;
        if (a.sc_per_example) {
            const double s0 = sqrt((double)sm.x), s1 = sqrt((double)sm.y);
            if (staged) ratio[i] = s0 / s1;
            double c_sc = a.w_sc / ((double)a.rows * s0 * s1);
            if (!(c_sc == c_sc) || c_sc > 1e30) c_sc = 0.0;  // identical signals: 0/0 -> no SC gradient
            write_coef(a, i, i / a.rows, c_sc);
        }
    }
    sync();
    if (tid < a.n_res) {
        const int res = tid;
        double tot[4] = {0, 0, 0, 0}, sc_acc = 0.0;
        for (int row = 0; row < a.rows; ++row) {
            const int i = res * a.rows + row;
            const float4 sm = staged ? srow[i] : mrstft_load_sums<AGENT>(a.sums, i);
            tot[0] += (double)sm.x; tot[1] += (double)sm.y; tot[2] += (double)sm.z; tot[3] += (double)sm.w;
            if (a.sc_per_example) sc_acc += staged ? ratio[i] : sqrt((double)sm.x) / sqrt((double)sm.y);  // same roots, same quotient
        }
        for (int q = 0; q < 4; ++q) rs[res][q] = tot[q];
        if (a.gtotals) {  // batch-global ratio over the rows of every rank
            rs[res][0] = a.gtotals[res * 4 + 0];
            rs[res][1] = a.gtotals[res * 4 + 1];
        }
        const double sc = a.sc_per_example ? sc_acc / a.rows : sqrt(rs[res][0]) / sqrt(rs[res][1]);
        rs[res][3] = a.w_sc * sc + a.w_log * tot[2] / a.count[res] + a.w_lin * tot[3] / a.count[res];
    }
    sync();
    if (tid == 0) {
        double total = 0.0;
        for (int res = 0; res < a.n_res; ++res) total += rs[res][3];
        a.loss[0] = (float)(total / a.n_res);
    }
    if (a.sc_per_example) return;  // coefficients written in the staging loop
    for (int i = tid; i < a.n_res * a.rows; i += 64) {
        const int res = i / a.rows;
        double c_sc = a.w_sc * (double)a.world / (sqrt(rs[res][0]) * sqrt(rs[res][1]));  // world: see mst_mrstft_forward_finish
        if (!(c_sc == c_sc) || c_sc > 1e30) c_sc = 0.0;  // identical signals: 0/0 -> no SC gradient
        write_coef(a, i, res, c_sc);
    }
}

struct StftArgs {
    const float* pred;     // (rows, n)
    const float* target;   // (rows, n)
    const float* tables;
    float* part;           // forward: (rows, n_groups, 4) partial sums {S1, S2, S3, S4}
    const float* sums;     // backward: (rows, 4) reduced sums of this resolution
    const float* coef;     // backward: (rows, 4) per-row gradient coefficients {c_sc, c_log, c_lin, -}
    const float* grad_loss; // backward: upstream dL/dloss (one device float), folded into the coefficients
    float* grad_pred;      // backward: (rows, n), accumulated with float atomics
    ResInfo r;
    int log2n;
    int64_t n;
    float eps;
    int accumulate;        // round-2 backward: 0 = this launch owns grad_pred (plain stores), 1 = it adds to what is there
    // Seam hand-over between launches (null: a seam-mode launch adds both seam halves with atomics onto a zeroed buffer).
    // A seam-mode launch (8192) stores every block once - a strip's trailing half frame goes to grad_pred, the half frame that
    // opens the NEXT strip goes to the same samples of `seam` - and the first halo-mode launch after it adds `seam` back at
    // exactly those blocks (seam_frames / seam_groups describe the seam launch's strips; seam_hop its block length).
    float* seam;           // (rows, n) scratch, only seam blocks are ever touched
    int seam_frames, seam_groups, seam_hop;
    // forward, first launch of a call: workgroup (0, 0) zeroes the ticket of k_mrstft_finish (null: nothing to arm)
    unsigned* tickets;
    // round-2 kernels, round 5: the target's clamped magnitudes sqrt(max(|Y|^2, eps)), (rows, n_frames, n_fft / 2 + 1) - written by the
    // forward, read by the backward, which then transforms the prediction alone (null in the forward: not kept)
    float* ymag;
    // 8192-point resolution, round 5: the prediction's spectrum X (float2 per bin and frame, (rows, n_frames, n_fft / 2 + 1)) - written by
    // the forward, read by the backward, which then runs the inverse transform only (MST_STFT2_BWD_SAVED_SPEC_8192; null: not kept)
    float* xspec;
};
#ifndef MST_STFT2_BWD_SAVED_SPEC_8192
#define MST_STFT2_BWD_SAVED_SPEC_8192 1  // 0: the 8192-point backward recomputes every frame's (prediction + i target) transform (rounds 2-4)
#endif
#ifndef MST_STFT2_BWD_SAVED_SPEC_SMALL
#define MST_STFT2_BWD_SAVED_SPEC_SMALL 1  // the same for the 512- / 2048-point resolutions (0: their backward transforms the prediction's frames in pairs)
#endif
__host__ __device__ constexpr bool stft2_keeps_spectrum(int n_fft) {
    return n_fft == 8192 ? MST_STFT2_BWD_SAVED_SPEC_8192 != 0 : ((n_fft == 512 || n_fft == 2048) && MST_STFT2_BWD_SAVED_SPEC_SMALL != 0);
}

// the three forward transforms of the reference's resolutions in one launch (mst_stft2.hip: k_stft3_fwd).  a[0] / a[1] / a[2] =
// the 8192- / 2048- / 512-point resolution; groups = strips per row; wg_end = running workgroup counts of the three roles
struct Stft3Args {
    StftArgs a[3];
    int groups[3];
    int wg_end[3];
    int rows;
    unsigned* tickets;  // workgroup 0 zeroes the ticket of k_mrstft_finish (null: nothing to arm)
};
void launch_stft3_fwd(const Stft3Args& p, hipStream_t stream);

constexpr float kLn2 = 0.6931471805599453f;

// round-2 launchers (mst_stft2.hip); grid.x = strips per row, grid.y = rows
void launch_stft2_fwd(const StftArgs& a, int n_groups, int rows, hipStream_t stream);
// backward: n_fft = 8192 runs in seam mode (every frame once; a strip's two half-frame seams are added atomically, exactly two
// contributions per sample onto a zeroed buffer, hence order-independent); 512 / 2048 in halo mode (a strip owns whole hop
// blocks and recomputes the one frame it shares with its neighbour).  n_groups from stft2_bwd_groups().
int stft2_bwd_groups(int n_fft, int n_frames, int rows);
bool stft2_bwd_needs_zero(int n_fft);
void launch_stft2_bwd(const StftArgs& a, int n_groups, int rows, hipStream_t stream);
// the 512- and the 2048-point backward in one launch (mst_stft2.hip: k_stft2_bwd_512_2048); a512 is the first to touch its samples
// after any seam-mode launch (accumulate / seam fields as for a stand-alone launch), a2048 adds to what a512 wrote (accumulate = 1)
bool stft2_bwd_can_fuse(int64_t n_samples);
void launch_stft2_bwd_512_2048(const StftArgs& a512, const StftArgs& a2048, int rows, hipStream_t stream);

}  // namespace mst
