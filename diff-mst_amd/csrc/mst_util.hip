// mst_util.hip - batch_stereo_peak_normalize (reference mst/utils.py:14-29):
//   y = x / clamp(max over (channel, time) of |x|, 1e-8)   per batch item, x (bs, 2, n) dense.
// Two launches: per-workgroup max|x| (+ first index), then every workgroup re-reduces the few
// partials of its batch item in a fixed order and scales its slice.  Backward is the exact
// reverse-mode (gradient also flows to the arg-max sample through the peak).
#include "mst_common.h"

namespace mst {
constexpr int kPeakSpan = 256 * 16;  // samples per workgroup

__global__ __launch_bounds__(256) void k_peak_partial(const float* __restrict__ x, float* __restrict__ part, int64_t len) {
    __shared__ float sv[4];
    __shared__ int si[4];
    const int b = blockIdx.y, tid = threadIdx.x;
    const float* row = x + (int64_t)b * len;
    const int64_t base = (int64_t)blockIdx.x * kPeakSpan;
    float best = -1.0f;
    int bi = 0x7fffffff;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int64_t i = base + ((int64_t)r * 256 + tid) * 4;
        const float4 v = load4(row, i, len);
        const float a[4] = {fabsf(v.x), fabsf(v.y), fabsf(v.z), fabsf(v.w)};
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (i + q < len && a[q] > best) {  // strict: keeps the first occurrence inside the lane
                best = a[q];
                bi = (int)(i + q - base);
            }
    }
    // wave arg-max (ties -> smaller index), then across the 4 waves
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const float ov = __shfl_xor(best, m);
        const int oi = __shfl_xor(bi, m);
        if (ov > best || (ov == best && oi < bi)) {
            best = ov;
            bi = oi;
        }
    }
    if ((tid & 63) == 0) {
        sv[tid >> 6] = best;
        si[tid >> 6] = bi;
    }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; ++w)
            if (sv[w] > best || (sv[w] == best && si[w] < bi)) {
                best = sv[w];
                bi = si[w];
            }
        float* p = part + ((int64_t)b * gridDim.x + blockIdx.x) * 2;
        p[0] = best;
        p[1] = __int_as_float(bi);
    }
}

// reduce the partials of batch item b (fixed order), return (peak, global arg index)
__device__ __forceinline__ void peak_of(const float* __restrict__ part, int b, int nblk, float& peak, int64_t& arg) {
    peak = -1.0f;
    arg = 0;
    for (int k = 0; k < nblk; ++k) {
        const float v = part[((int64_t)b * nblk + k) * 2];
        if (v > peak) {
            peak = v;
            arg = (int64_t)k * kPeakSpan + __float_as_int(part[((int64_t)b * nblk + k) * 2 + 1]);
        }
    }
    if (peak < 0.0f) peak = 0.0f;
}

__global__ __launch_bounds__(256) void k_peak_scale(const float* __restrict__ x, const float* __restrict__ part,
                                                    float* __restrict__ y, int64_t len) {
    const int b = blockIdx.y, tid = threadIdx.x;
    float peak;
    int64_t arg;
    peak_of(part, b, gridDim.x, peak, arg);
    const float inv = 1.0f / fmaxf(peak, 1e-8f);
    const int64_t base = (int64_t)blockIdx.x * kPeakSpan;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int64_t i = base + ((int64_t)r * 256 + tid) * 4;
        float4 v = load4(x + (int64_t)b * len, i, len);
        // division, not multiplication by the reciprocal, to round like the reference's x / peak
        const float pc = fmaxf(peak, 1e-8f);
        v.x /= pc; v.y /= pc; v.z /= pc; v.w /= pc;
        (void)inv;
        store4(y + (int64_t)b * len, i, len, v);
    }
}

__global__ __launch_bounds__(256) void k_peak_bwd_partial(const float* __restrict__ x, const float* __restrict__ g,
                                                          float* __restrict__ dots, int64_t len) {
    __shared__ float sv[4];
    const int b = blockIdx.y, tid = threadIdx.x;
    const int64_t base = (int64_t)blockIdx.x * kPeakSpan;
    float acc = 0.0f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int64_t i = base + ((int64_t)r * 256 + tid) * 4;
        const float4 xv = load4(x + (int64_t)b * len, i, len), gv = load4(g + (int64_t)b * len, i, len);
        acc = fmaf(xv.x, gv.x, fmaf(xv.y, gv.y, fmaf(xv.z, gv.z, fmaf(xv.w, gv.w, acc))));
    }
    acc = wave_sum(acc);
    if ((tid & 63) == 0) sv[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) dots[(int64_t)b * gridDim.x + blockIdx.x] = (sv[0] + sv[1]) + (sv[2] + sv[3]);
}

__global__ __launch_bounds__(256) void k_peak_bwd(const float* __restrict__ x, const float* __restrict__ g,
                                                  const float* __restrict__ part, const float* __restrict__ dots,
                                                  float* __restrict__ dx, int64_t len) {
    const int b = blockIdx.y, tid = threadIdx.x, nblk = gridDim.x;
    float peak;
    int64_t arg;
    peak_of(part, b, nblk, peak, arg);
    const float pc = fmaxf(peak, 1e-8f);
    double dot = 0.0;
    for (int k = 0; k < nblk; ++k) dot += (double)dots[(int64_t)b * nblk + k];
    const float corr = (peak >= 1e-8f) ? (float)(dot / ((double)pc * (double)pc)) : 0.0f;
    const int64_t base = (int64_t)blockIdx.x * kPeakSpan;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int64_t i = base + ((int64_t)r * 256 + tid) * 4;
        const float4 gv = load4(g + (int64_t)b * len, i, len);
        float o[4] = {gv.x / pc, gv.y / pc, gv.z / pc, gv.w / pc};
        if (arg >= i && arg < i + 4) {
            const float xa = x[(int64_t)b * len + arg];
            o[arg - i] -= corr * ((xa > 0.f) - (xa < 0.f));
        }
        store4(dx + (int64_t)b * len, i, len, make_float4(o[0], o[1], o[2], o[3]));
    }
}
}  // namespace mst

using namespace mst;

extern "C" size_t mst_peak_normalize_workspace_bytes(int32_t bs, int64_t n_samples) {
    if (bs <= 0 || n_samples <= 0) return 0;
    const int64_t nblk = (2 * n_samples + kPeakSpan - 1) / kPeakSpan;
    return (size_t)(bs * nblk * 3) * sizeof(float);
}
extern "C" int mst_peak_normalize_forward(const float* x, float* y, int32_t bs, int64_t n_samples, void* workspace,
                                          size_t workspace_bytes, void* stream_) {
    if (!x || !y || !workspace || workspace_bytes < mst_peak_normalize_workspace_bytes(bs, n_samples)) return hipErrorInvalidValue;
    hipStream_t stream = (hipStream_t)stream_;
    const int64_t len = 2 * n_samples;
    const int nblk = (int)((len + kPeakSpan - 1) / kPeakSpan);
    float* part = (float*)workspace;
    hipLaunchKernelGGL(k_peak_partial, dim3(nblk, bs), dim3(256), 0, stream, x, part, len);
    hipLaunchKernelGGL(k_peak_scale, dim3(nblk, bs), dim3(256), 0, stream, x, part, y, len);
    return (int)hipGetLastError();
}
extern "C" int mst_peak_normalize_backward(const float* x, const float* grad_y, float* grad_x, int32_t bs, int64_t n_samples,
                                           void* workspace, size_t workspace_bytes, void* stream_) {
    if (!x || !grad_y || !grad_x || !workspace || workspace_bytes < mst_peak_normalize_workspace_bytes(bs, n_samples))
        return hipErrorInvalidValue;
    hipStream_t stream = (hipStream_t)stream_;
    const int64_t len = 2 * n_samples;
    const int nblk = (int)((len + kPeakSpan - 1) / kPeakSpan);
    float* part = (float*)workspace;
    float* dots = part + (int64_t)bs * nblk * 2;
    hipLaunchKernelGGL(k_peak_bwd_partial, dim3(nblk, bs), dim3(256), 0, stream, x, grad_y, dots, len);
    hipLaunchKernelGGL(k_peak_bwd, dim3(nblk, bs), dim3(256), 0, stream, x, grad_y, part, dots, grad_x, len);
    return (int)hipGetLastError();
}
