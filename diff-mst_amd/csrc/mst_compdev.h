// mst_compdev.h - device helpers of the feed-forward compressor shared by mst_comp.hip and the fused
// EQ-run + gain-computer pass of mst_eq.hip (static curve of dasp-pytorch's compressor).
#pragma once
#include "mst_common.h"

namespace mst {

struct CompK {
    float thr, kappa, knee, hw, inv2w, invw, alpha, oma, mk;
};
__device__ __forceinline__ CompK load_comp(const float* rc) {
    CompK k;
    k.thr = rc[RC_THR];
    k.kappa = rc[RC_KAPPA];
    k.knee = rc[RC_KNEE];
    k.hw = 0.5f * k.knee;
    k.invw = 1.0f / k.knee;
    k.inv2w = 0.5f * k.invw;
    k.alpha = rc[RC_ALPHA];
    k.oma = 1.0f - k.alpha;
    k.mk = rc[RC_MAKEUP];
    return k;
}
// static curve: returns g_c = kappa * f(x_db - thr); d = x_db - thr is handed back
__device__ __forceinline__ float gain_computer(float side, const CompK& k, float& d) {
    const float ax = fmaxf(fabsf(side), kCompEps);
    d = kDbPerLog2 * __builtin_amdgcn_logf(ax) - k.thr;
    float f = 0.0f;
    if (d > k.hw) f = d;
    else if (d >= -k.hw) {
        const float t = d + k.hw;
        f = t * t * k.inv2w;
    }
    return k.kappa * f;
}
__device__ __forceinline__ float lin_gain(float gs, const CompK& k) {
    return __builtin_amdgcn_exp2f((gs + k.mk) * kLog2PerDb);
}

}  // namespace mst
