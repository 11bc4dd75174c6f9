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
    // a zero knee (reachable through forward_mix_console / MST_NO_RANGE_CHECK, which apply denormalised values unchecked) is the hard-knee
    // curve: kept finite as a knee of 1e-6 dB - the branch-free form below divides by the width (0 x inf = NaN for every sample),
    // while the reference's torch.where form is finite everywhere but at d == 0
    k.knee = fmaxf(rc[RC_KNEE], 1e-6f);
    k.hw = 0.5f * k.knee;
    k.invw = 1.0f / k.knee;
    k.inv2w = 0.5f * k.invw;
    k.alpha = rc[RC_ALPHA];
    k.oma = 1.0f - k.alpha;
    k.mk = rc[RC_MAKEUP];
    return k;
}
// static curve of the soft knee, branch-free.  With t = x_db - thr + knee / 2 (the level above the knee's lower edge) and
// tc = clamp(t, 0, knee):   f = tc^2 / (2 knee) + max(t - knee, 0)   = 0 below the knee, t^2 / 2w inside, d above it;
//   df/dd = tc / knee   (0 | t / w | 1),     df/dknee = tc (knee - tc) / (2 knee^2)   (0 | t (hw - d) / 2w^2 | 0).
// One v_med3, no compare / select / exec-mask branch: the kernels that evaluate it are bound by instruction issue.
__device__ __forceinline__ float curve_t(float side, const CompK& k) {
    return fmaf(kDbPerLog2, __builtin_amdgcn_logf(fmaxf(fabsf(side), kCompEps)), k.hw - k.thr);
}
__device__ __forceinline__ float curve_f(float t, const CompK& k, float& tc) {
    tc = __builtin_amdgcn_fmed3f(t, 0.0f, k.knee);
    return fmaf(tc * k.inv2w, tc, fmaxf(t - k.knee, 0.0f));
}
// returns g_c = kappa * f(x_db - thr); d = x_db - thr is handed back
__device__ __forceinline__ float gain_computer(float side, const CompK& k, float& d) {
    const float t = curve_t(side, k);
    float tc;
    d = t - k.hw;
    return k.kappa * curve_f(t, k, tc);
}
__device__ __forceinline__ float lin_gain(float gs, const CompK& k) {
    return __builtin_amdgcn_exp2f((gs + k.mk) * kLog2PerDb);
}

}  // namespace mst
