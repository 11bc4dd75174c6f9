// mst_af.h - shared declarations of the AudioFeatureLoss kernels (mst_af.hip: closed-form features, reductions, the C ABI;
// mst_af2.hip: the Bark-spectrum transforms on the register-radix FFT engine).
#pragma once
#include "mst_common.h"

namespace mst {

constexpr int kAfFft = 32768;       // reference default fft_size (mst/loss.py:64)
constexpr int kAfM = kAfFft / 2;    // complex points of the packed real transform
constexpr int kAfHalf = kAfM / 2;   // the 16384-point transform runs as two 8192-point ones (even / odd bins)
constexpr int kAfHop = kAfFft / 4;  // reference hop_length = fft_size // 4 (mst/loss.py:106)
constexpr int kAfBins = kAfM + 1;
constexpr int kAfBands = 24;

// tables (floats): twM: kAfM float2 (W_M^t) | twN: (kAfM + 1) float2 (W_N^k) | win: kAfFft floats | twH: kAfHalf float2 (W_(M/2)^t)
constexpr int64_t kAfTwM = 0, kAfTwN = 2 * kAfM, kAfWin = kAfTwN + 2 * (kAfM + 1), kAfTwH = kAfWin + kAfFft,
                  kAfTablesFloats = kAfTwH + 2 * kAfHalf;

struct AfArgs {
    const float* pred;    // (bs, 2, n)
    const float* target;  // (bs, 2, n)
    const float* tables;
    const float* fb;      // (kAfBins, 24) filterbank, row-major like the reference's (n_freqs, n_barks)
    float* magpart;       // (4*bs, n_groups, kAfBins) partial sums of |X| over a strip of frames
    float* meanmag;       // (4*bs, kAfBins)
    float* bark;          // (4*bs, 24) log band energies; (4*bs, 24) linear band energies follow
    double* stats;        // (2*bs, 8) reduced statistics per (signal set, b), see k_af_stats / k_af_finish
    float* statpart;      // partials of the above
    float* bandpart;      // (4*bs, kAfBinSlices, 24) partial band energies
    float* losses;        // 5 weighted loss scalars out
    float* coef;          // backward coefficients
    const float* grad_losses;  // (5) upstream dL/d(loss_k)
    float* grad_pred;     // (bs, 2, n)
    float* yframes;       // (2*bs, n_frames, kAfFft) windowed adjoint frames of the prediction's mid / side signals
    float weights[5];
    int bs, n_frames, n_groups, n_statblk;
    int64_t n;
};

// signal index s in [0, 4*bs): which = s / bs (0 pred mid, 1 pred side, 2 target mid, 3 target side), b = s % bs
__device__ __forceinline__ void af_signal(const AfArgs& a, int s, const float*& l, const float*& r, float& sign) {
    const int which = s / a.bs, b = s % a.bs;
    const float* base = (which < 2 ? a.pred : a.target) + (int64_t)b * 2 * a.n;
    l = base;
    r = base + a.n;
    sign = (which & 1) ? -1.0f : 1.0f;
}

// mst_af2.hip
constexpr int kAf2Slots = 512;  // co-resident 512-lane workgroups (64 KiB of LDS each: two per CU)
void launch_af2_bark_fwd(const AfArgs& a, hipStream_t stream);  // grid (n_groups, 4 bs, 2 halves)
void launch_af2_bark_bwd(const AfArgs& a, hipStream_t stream);  // grid (n_frames, 2 bs)

}  // namespace mst
