// mst_cnn.h - shared declarations of the spectrogram encoder kernels (mst_cnn.hip: MFMA convolutions, mst_cnn_net.hip: the
// Cnn14 launch sequences behind mst_cnn14_forward / _backward).  SURVEY 8f rank 2; reference mst/panns.py:27-209,
// mst/modules.py:740-806.
//
// Image convention: the reference's spectrogram is (bs, 1, bins, frames); the STFT kernel emits (frames, bins) rows (bins
// contiguous: coalesced stores), so the whole network runs on the TRANSPOSED image - H = frames, W = bins, every 3x3 weight
// transposed (tap (dh, dw) here = torch's [kw][kh]) and every pool size swapped.  Activations are NHWC, T = bf16 (uint16_t
// storage) or fp32; accumulation, BatchNorm statistics and all parameter gradients are fp32 (partials folded in fp64).
#pragma once
#include "mst_common.h"

namespace mst {

using bf16_t = uint16_t;

struct ConvArgs {
    const void* in;   // (N, H, W, Cin) T
    const void* w;    // (Cout, 9, Cin) T, tap = (dh + 1) * 3 + (dw + 1)
    void* out;        // (N, H, W, Cout) T
    float* part;      // (pixel tiles, Cout, 2) per-tile sum / sum of squares of the fp32 results, or null
    int N, H, W, Cin, Cout;
    // split K (layers with few pixels and a long K = 9 Cin: 32 workgroups at 16 signals otherwise): workgroup z of
    // 9 * csplit multiplies tap z / csplit, input channels [(z % csplit) Cin / csplit, +Cin / csplit) and stores its fp32 partial
    // product to kpart (9 csplit, pixels, Cout); k_conv_splitk_reduce folds them in a fixed order (and makes the statistics)
    int csplit;       // 0: no split
    float* kpart;
};
struct WgradArgs {
    const void* dy;   // (N, H, W, Cout) T
    const void* x;    // (N, H, W, Cin) T   (first layer: (N, H, W) fp32 spectrogram)
    float* part;      // (splits, 9, Cout, Cin) fp32 partial sums  (first layer: (splits, Cout, 16))
    int N, H, W, Cin, Cout;
    int splits;
    int steps_per_split;  // K steps (of kWgradPix pixels) per split
};
constexpr int kConvPix = 128;   // pixels per workgroup of the implicit-GEMM convolution
constexpr int kWgradPix = 32;   // pixels per K step of the weight-gradient GEMM

// all launchers: precision 0 = bf16 operands, 1 = fp32 operands, 2 / 3 = fp32 storage with the bf16x3 / bf16x6 operand split (mst_cnn14_desc::precision)
// returns the number of pixel tiles whose statistics it wrote to a.part (the kernels differ in tile size)
int launch_conv3x3(int precision, ConvArgs a, hipStream_t s, float* kpart, size_t kpart_bytes);
size_t conv_splitk_bytes(int N, int H, int W, int Cin, int Cout);
void launch_conv_wgrad(int precision, const WgradArgs& a, hipStream_t s);
// true: launch_conv_wgrad runs all nine taps in one workgroup (k_conv_wgrad4) - the split policy counts a ninth of the workgroups
bool conv_wgrad_nine_taps(int precision, int W, int Cin, int Cout);
int conv_pixel_tiles(int N, int H, int W);

}  // namespace mst
