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
};

constexpr float kLn2 = 0.6931471805599453f;

// round-2 launchers (mst_stft2.hip); grid.x = strips per row, grid.y = rows
void launch_stft2_fwd(const StftArgs& a, int n_groups, int rows, hipStream_t stream);
// backward: n_fft = 8192 runs in seam mode (every frame once; a strip's two half-frame seams are added atomically, exactly two
// contributions per sample onto a zeroed buffer, hence order-independent); 512 / 2048 in halo mode (a strip owns whole hop
// blocks and recomputes the one frame it shares with its neighbour).  n_groups from stft2_bwd_groups().
int stft2_bwd_groups(int n_fft, int n_frames, int rows);
bool stft2_bwd_needs_zero(int n_fft);
void launch_stft2_bwd(const StftArgs& a, int n_groups, int rows, hipStream_t stream);

}  // namespace mst
