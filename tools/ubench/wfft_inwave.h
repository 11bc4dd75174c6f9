// wfft_inwave.h - round-5 EXPERIMENT, measured and NOT taken: a 512-point transform whose exchanges never touch LDS.
// Result (tools/ubench/wf512.hip, MI355X, 4 / 8 waves per SIMD): 1370 / 1344 cycles per transform per SIMD against 816 / 735 for
// the LDS-exchange engine of mst_fft2.h - correct (8.7e-6 against a float64 DFT) and 1.7x SLOWER.  tools/ubench/xlane_rate.hip says
// why: the cross-lane VALU forms are not full rate on gfx950 - v_mov_b32_dpp (any control) issues at ~1.75x the cycles of v_fma_f32,
// v_permlane32_swap / v_permlane16_swap at ~3.3x (two floats moved, so the same per float), ds_bpermute_b32 at 24 cycles per
// wave-instruction per SIMD; the 128 exchange instructions of a transform (64 DPP moves, 16 swaps, 32 selects, 16 copies) cost as
// much as its 228 butterfly / twiddle instructions, while the LDS engine's 16 + 16 DS instructions run beside the VALU.
// Kept as the record of the measurement; the product engine stays mst_fft2.h.
//
// Original design note:
//
// ONE wave transforms 512 complex points that never leave its registers: lane l holds 8 points, three radix-8 passes run in
// registers, and the two exchanges between the passes - the transposition of the 3-bit register index with three bits of the
// lane index - are done with the cross-lane data paths of the vector ALU instead of a round trip through LDS:
//   lane bits 5, 4   v_permlane32_swap / v_permlane16_swap (gfx950): one instruction swaps the upper half (odd rows) of one
//                    register with the lower half (even rows) of another - half an instruction per moved float
//   lane bits 3, 2   v_mov_b32_dpp row_ror:8 / row_shr:4 / row_shl:4 with a bank mask that enables exactly the receiving lanes
//   lane bits 1, 0   v_mov_b32_dpp quad_perm + v_cndmask
// ~90 VALU instructions per transform replace the 16 ds_write_b64 + 16 ds_read_b64 (+ waits, + bank conflicts: 35-40 % of the LDS
// cycles of the round-2 engine, profiles/round4_counters.md) of the two exchanges of mst_fft2.h - and every twiddle of the
// transform is a per-lane register constant (14 + 14 floats), which saves the 10 twiddle-by-twiddle products per transform
// that engine spends.  No LDS, no barrier, no s_waitcnt inside a transform: a wave's transform is one dependent-free VALU stream.
//
// Index maps (n = input sample, k = output bin; l = lane):
//   in :  v[t] = x[l + 64 t]                                   (lane-consecutive: coalesced global / conflict-free LDS reads)
//   out:  v[d] = X[k],  k = (l >> 3) + 8 (l & 7) + 64 d         (digit-reversed over the lane: kfreq())
// Larger transforms are built from it with ONE exchange through LDS (mst_stft3.hip): R waves each transform the decimated
// sequence x[s + R m], a radix-R pass over s combines them - and because that pass is the last one, the lane that forms
// X[k + 512 q] (all q) also forms X[(512 - k) + 512 q], i.e. it holds every bin TOGETHER with its mirror bin N - k: the
// untangling of the two real signals packed into one complex transform, and the Hermitian packing of the inverse, need no
// further exchange.
#pragma once
#include "../../diff-mst_amd/csrc/mst_fft2.h"

namespace mst {

template <int CTRL, int BANK_MASK>
__device__ __forceinline__ float dpp_sel(float old, float src) {  // lanes enabled by BANK_MASK: src of the lane CTRL names; others: old
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(src), CTRL, 0xf, BANK_MASK, false));
}
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float src) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(src), CTRL, 0xf, 0xf, true));
}
// a = register whose index has the exchanged bit clear, b = set.  Lanes with the lane bit clear keep a and receive the partner's a
// into b; lanes with the lane bit set keep b and receive the partner's b into a.
__device__ __forceinline__ void xch32(float& a, float& b) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]);
    b = __uint_as_float(r[1]);
}
__device__ __forceinline__ void xch16(float& a, float& b) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]);
    b = __uint_as_float(r[1]);
}
__device__ __forceinline__ void xch8(float& a, float& b) {  // lane bit 3: partner = lane ^ 8 = row_ror:8
    const float na = dpp_sel<0x128, 0xc>(a, b), nb = dpp_sel<0x128, 0x3>(b, a);
    a = na;
    b = nb;
}
__device__ __forceinline__ void xch4(float& a, float& b) {  // lane bit 2: lanes 4-7, 12-15 read lane - 4 (row_shr:4), the others lane + 4 (row_shl:4)
    const float na = dpp_sel<0x114, 0xa>(a, b), nb = dpp_sel<0x104, 0x5>(b, a);
    a = na;
    b = nb;
}
__device__ __forceinline__ void xch2(float& a, float& b, bool hi) {  // lane bit 1: quad_perm [2, 3, 0, 1]
    const float pa = dpp_mov<0x4e>(a), pb = dpp_mov<0x4e>(b);
    a = hi ? pb : a;
    b = hi ? b : pa;
}
__device__ __forceinline__ void xch1(float& a, float& b, bool hi) {  // lane bit 0: quad_perm [1, 0, 3, 2]
    const float pa = dpp_mov<0xb1>(a), pb = dpp_mov<0xb1>(b);
    a = hi ? pb : a;
    b = hi ? b : pa;
}

// per-lane twiddles of the 512-point in-wave transform, from the exactly rounded table tw[t] = (cos, -sin)(2 pi t / NT), NT a
// multiple of 512
struct WfTw {
    float2 t1[7];  // W_512^(l k1), k1 = 1..7
    float2 t2[7];  // W_64^((l & 7) c), c = 1..7
    template <int NT>
    __device__ __forceinline__ void init(const float2* __restrict__ tw, int lane) {
#pragma unroll
        for (int k = 1; k < 8; ++k) {
            t1[k - 1] = tw[((NT / 512) * (lane * k)) & (NT - 1)];
            t2[k - 1] = tw[((NT / 64) * ((lane & 7) * k)) & (NT - 1)];
        }
    }
};

// bin held by register d of lane l after wf512
__device__ __forceinline__ constexpr int wf_kfreq(int l, int d) { return (l >> 3) + 8 * (l & 7) + 64 * d; }

__device__ __forceinline__ void wf512(float2* v, const WfTw& tw, int lane) {
    butterfly<8>(v);  // over t -> k1
#pragma unroll
    for (int k = 1; k < 8; ++k) v[k] = cmul(v[k], tw.t1[k - 1]);
    // register index k1 <-> lane bits 5..3 (a)
#pragma unroll
    for (int r = 0; r < 4; ++r) { xch32(v[r].x, v[r + 4].x); xch32(v[r].y, v[r + 4].y); }
#pragma unroll
    for (int r = 0; r < 8; ++r)
        if (!(r & 2)) { xch16(v[r].x, v[r + 2].x); xch16(v[r].y, v[r + 2].y); }
#pragma unroll
    for (int r = 0; r < 8; r += 2) { xch8(v[r].x, v[r + 1].x); xch8(v[r].y, v[r + 1].y); }
    butterfly<8>(v);  // over a -> c
#pragma unroll
    for (int k = 1; k < 8; ++k) v[k] = cmul(v[k], tw.t2[k - 1]);
    // register index c <-> lane bits 2..0 (b)
    const bool h1 = (lane & 2) != 0, h0 = (lane & 1) != 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) { xch4(v[r].x, v[r + 4].x); xch4(v[r].y, v[r + 4].y); }
#pragma unroll
    for (int r = 0; r < 8; ++r)
        if (!(r & 2)) { xch2(v[r].x, v[r + 2].x, h1); xch2(v[r].y, v[r + 2].y, h1); }
#pragma unroll
    for (int r = 0; r < 8; r += 2) { xch1(v[r].x, v[r + 1].x, h0); xch1(v[r].y, v[r + 1].y, h0); }
    butterfly<8>(v);  // over b -> d
}

}  // namespace mst
