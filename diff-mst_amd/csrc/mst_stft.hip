// mst_stft.hip - multi-resolution STFT loss (auraloss 0.4.0 semantics; reference
// configuration configs/models/naive.yaml:54-68, call site mst/system.py:331).
//
//   per resolution: frames of torch.stft(center=True, reflect pad, periodic Hann), hop = n_fft/2..n_fft/4
//   mag = sqrt(max(re^2+im^2, 1e-8));  SC = ||ym - xm||_F / ||ym||_F (per row or global);
//   logmag = mean |log xm - log ym|;  loss = mean over resolutions of w_sc SC + w_log logmag + w_lin L1.
//
// Kernel shape: one workgroup owns a strip of consecutive frames of one (batch, channel) row.  A
// frame of the prediction and the same frame of the target are transformed by ONE complex FFT
// (z = w*(x + i y)) living in LDS (mixed radix-2/4 decimation-in-time, digit-reversed on load),
// the two spectra are separated by Hermitian symmetry and the magnitude / log / partial-sum
// epilogue is fused - no spectrogram ever reaches HBM.  Backward recomputes the forward tile,
// forms dL/dX, and returns two frames' time-domain cotangents per complex FFT; the overlap-add
// (incl. the reflect-padding fold-back) uses hardware float atomics into grad_pred.
#include "mst_fft.h"

namespace mst {

constexpr int kMaxRes = 8;

struct ResInfo {
    int n_fft, hop, n_frames, n_bins;
    int64_t tw_off, win_off;  // float offsets into the tables buffer (tw: n_fft float2, win: n_fft floats)
    int frames_per_slot;      // forward: consecutive frames handled by one slot of a workgroup
};

// Workgroup geometry: a transform is owned by TPF = FftPlan<N>::tpf lanes (one radix-16 butterfly per
// lane and pass); small transforms run several frames ("slots") side by side in one workgroup.
constexpr int stft_slots(int n_fft) { return n_fft <= 512 ? 4 : (n_fft <= 1024 ? 2 : 1); }
template <int N> constexpr int stft_threads() { return FftPlan<N>::tpf * stft_slots(N); }

__device__ __forceinline__ int64_t reflect_index(int64_t i, int64_t n) {
    if (i < 0) i = -i;
    if (i >= n) i = 2 * (n - 1) - i;
    return i;
}

// frame f of (x, y) as z = w (x + i y) in natural order (padded LDS indexing); zeros when !valid
template <int N, int TPF>
__device__ __forceinline__ void load_frame(float2* buf, const float* __restrict__ x, const float* __restrict__ y,
                                           const float* __restrict__ win, int f, int hop, int64_t n, bool valid, int lane) {
    const int64_t start = (int64_t)f * hop - N / 2;
#pragma unroll 4
    for (int k = lane; k < N; k += TPF) {
        float2 v = make_float2(0.f, 0.f);
        if (valid) {
            const int64_t i = reflect_index(start + k, n);
            const float w = win[k];
            v = make_float2(w * x[i], w * y[i]);
        }
        buf[lds_pad(k)] = v;
    }
}

struct StftArgs {
    const float* pred;     // (rows, n)
    const float* target;   // (rows, n)
    const float* tables;
    float* part;           // forward: (rows, n_groups, 4) partial sums {S1, S2, S3, S4}
    const float* sums;     // backward: (rows, 4) reduced sums of this resolution
    const float* coef;     // backward: (rows, 4) per-row gradient coefficients {c_sc, c_log, c_lin, -}
    float* grad_pred;      // backward: (rows, n), accumulated with float atomics
    ResInfo r;
    int log2n;
    int64_t n;
    float eps;
};

// separate the two real spectra packed in one complex FFT
template <int N>
__device__ __forceinline__ void split_xy(const float2* buf, int k, float2& X, float2& Y) {
    const float2 zk = buf[lds_pad(k)], zn = buf[lds_pad((N - k) & (N - 1))];
    X = make_float2(0.5f * (zk.x + zn.x), 0.5f * (zk.y - zn.y));
    Y = make_float2(0.5f * (zk.y + zn.y), -0.5f * (zk.x - zn.x));
}

constexpr float kLn2 = 0.6931471805599453f;

template <int N>
__global__ __launch_bounds__(stft_threads<N>()) void k_stft_fwd(StftArgs a) {
    constexpr int TPF = FftPlan<N>::tpf, SLOTS = stft_slots(N), THREADS = TPF * SLOTS, PN = lds_padded(N);
    __shared__ __attribute__((aligned(16))) float2 bufs[SLOTS * 2 * PN];
    __shared__ float red[THREADS / 64][4];
    const int tid = threadIdx.x, row = blockIdx.y, slot = tid / TPF, lane = tid % TPF;
    const ResInfo r = a.r;
    LaneTw<N> twd;
    twd.init(reinterpret_cast<const float2*>(a.tables + r.tw_off), lane);
    const float* win = a.tables + r.win_off;
    const float* x = a.pred + (int64_t)row * a.n;
    const float* y = a.target + (int64_t)row * a.n;
    float2* A = bufs + slot * 2 * PN;
    float2* B = A + PN;
    float s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f;
    const int f0 = (blockIdx.x * SLOTS + slot) * r.frames_per_slot;
    for (int it = 0; it < r.frames_per_slot; ++it) {
        const int f = f0 + it;
        const bool valid = f < r.n_frames;
        load_frame<N, TPF>(A, x, y, win, f, r.hop, a.n, valid, lane);
        __syncthreads();
        const float2* Z = lds_fft<N>(A, B, twd, lane);
        if (valid) {
            for (int k = lane; k < r.n_bins; k += TPF) {
                float2 X, Y;
                split_xy<N>(Z, k, X, Y);
                const float xm = __builtin_amdgcn_sqrtf(fmaxf(X.x * X.x + X.y * X.y, a.eps));
                const float ym = __builtin_amdgcn_sqrtf(fmaxf(Y.x * Y.x + Y.y * Y.y, a.eps));
                const float d = ym - xm;
                s1 = fmaf(d, d, s1);
                s2 = fmaf(ym, ym, s2);
                s3 += fabsf(__builtin_amdgcn_logf(xm) - __builtin_amdgcn_logf(ym));  // log2; scaled by ln2 below
                s4 += fabsf(d);
            }
        }
        __syncthreads();
    }
    s3 *= kLn2;
    const int wave = tid >> 6, wl = tid & 63;
    s1 = wave_sum(s1); s2 = wave_sum(s2); s3 = wave_sum(s3); s4 = wave_sum(s4);
    if (wl == 0) { red[wave][0] = s1; red[wave][1] = s2; red[wave][2] = s3; red[wave][3] = s4; }
    __syncthreads();
    if (tid < 4) {
        float v = 0.f;
        for (int w = 0; w < THREADS / 64; ++w) v += red[w][tid];
        a.part[((int64_t)row * gridDim.x + blockIdx.x) * 4 + tid] = v;
    }
}

// cotangent G[k] = dL/dX[k] of the prediction's half spectrum for the frame whose packed FFT is Z
template <int N>
__device__ __forceinline__ float2 spectrum_cotangent(const float2* Z, int k, const StftArgs& a, const float* coef) {
    float2 X, Y;
    split_xy<N>(Z, k, X, Y);
    const float p2 = X.x * X.x + X.y * X.y;
    const float xm = __builtin_amdgcn_sqrtf(fmaxf(p2, a.eps));
    const float ym = __builtin_amdgcn_sqrtf(fmaxf(Y.x * Y.x + Y.y * Y.y, a.eps));
    const float rx = __builtin_amdgcn_rcpf(xm);
    float g = coef[0] * (xm - ym);
    const float dl = __builtin_amdgcn_logf(xm) - __builtin_amdgcn_logf(ym);
    g += coef[1] * ((dl > 0.f) - (dl < 0.f)) * rx;
    g += coef[2] * ((xm > ym) - (xm < ym));
    const float s = (p2 >= a.eps) ? g * rx : 0.0f;  // through sqrt(clamp(|X|^2, eps)): zero below the clamp
    return make_float2(s * X.x, s * X.y);
}

// Backward.  The real cotangent frame is Re IDFT of the half spectrum G, i.e. the IDFT of its
// Hermitian extension He (He[k] = G[k]/2, He[N-k] = conj(G[k])/2, real at k = 0, N/2), and
// IDFT(h) = conj(FFT(conj(h))).  PAIR: two frames share one complex inverse FFT (He_a + i He_b).
template <int N, bool PAIR>
__global__ __launch_bounds__(stft_threads<N>()) void k_stft_bwd(StftArgs a) {
    constexpr int TPF = FftPlan<N>::tpf, SLOTS = stft_slots(N), THREADS = TPF * SLOTS, PN = lds_padded(N);
    constexpr int NBUF = PAIR ? 3 : 2;
    __shared__ __attribute__((aligned(16))) float2 bufs[SLOTS * NBUF * PN];
    const ResInfo r = a.r;
    const int tid = threadIdx.x, row = blockIdx.y, slot = tid / TPF, lane = tid % TPF;
    LaneTw<N> twd;
    twd.init(reinterpret_cast<const float2*>(a.tables + r.tw_off), lane);
    const float* win = a.tables + r.win_off;
    const float* x = a.pred + (int64_t)row * a.n;
    const float* y = a.target + (int64_t)row * a.n;
    const float* coef = a.coef + (int64_t)row * 4;
    float* gx = a.grad_pred + (int64_t)row * a.n;
    float2* A = bufs + slot * NBUF * PN;
    float2* B = A + PN;
    const int unit = blockIdx.x * SLOTS + slot;  // frame (or frame pair) index
    const int fa = PAIR ? 2 * unit : unit, fb = fa + 1;
    const bool have_a = fa < r.n_frames, have_b = PAIR && fb < r.n_frames;
    const int64_t sa = (int64_t)fa * r.hop - N / 2, sb = (int64_t)fb * r.hop - N / 2;

    load_frame<N, TPF>(A, x, y, win, fa, r.hop, a.n, have_a, lane);
    __syncthreads();
    float2* Z = lds_fft<N>(A, B, twd, lane);
    float2* O = (Z == A) ? B : A;         // the buffer the forward result is NOT in
    float2* H = PAIR ? A + 2 * PN : O;    // where conj(He) is assembled
    for (int k = lane; k <= N / 2; k += TPF) {
        const float2 G = spectrum_cotangent<N>(Z, k, a, coef);
        const bool edge = (k == 0) || (k == N / 2);
        // conj(He): He[k] = G/2 -> (Gx/2, -Gy/2); He[N-k] = conj(G)/2 -> (Gx/2, +Gy/2)
        H[lds_pad(k)] = edge ? make_float2(G.x, 0.f) : make_float2(0.5f * G.x, -0.5f * G.y);
        if (!edge) H[lds_pad(N - k)] = make_float2(0.5f * G.x, 0.5f * G.y);
    }
    __syncthreads();
    if (PAIR) {
        load_frame<N, TPF>(A, x, y, win, fb, r.hop, a.n, have_b, lane);
        __syncthreads();
        Z = lds_fft<N>(A, B, twd, lane);
        if (have_b) {
            for (int k = lane; k <= N / 2; k += TPF) {
                const float2 G = spectrum_cotangent<N>(Z, k, a, coef);
                const bool edge = (k == 0) || (k == N / 2);
                // conj(i He_b): i He[k] = i G/2 -> conj = (-Gy/2, -Gx/2);  i He[N-k] = i conj(G)/2 -> conj = (Gy/2, -Gx/2)
                if (edge) {
                    H[lds_pad(k)].y -= G.x;
                } else {
                    float2 h = H[lds_pad(k)];
                    H[lds_pad(k)] = make_float2(h.x - 0.5f * G.y, h.y - 0.5f * G.x);
                    h = H[lds_pad(N - k)];
                    H[lds_pad(N - k)] = make_float2(h.x + 0.5f * G.y, h.y - 0.5f * G.x);
                }
            }
        }
        __syncthreads();
        O = A;  // both work buffers are free again
    } else {
        O = Z;  // forward result no longer needed
    }
    // FFT(conj(h)) = conj(r_a + i r_b)  =>  r_a = Re, r_b = -Im
    const float2* R = lds_fft<N>(H, O, twd, lane);
    if (have_a) {
#pragma unroll 4
        for (int k = lane; k < N; k += TPF) {
            const float2 v = R[lds_pad(k)];
            const float w = win[k];
            unsafeAtomicAdd(&gx[reflect_index(sa + k, a.n)], w * v.x);
            if (have_b) unsafeAtomicAdd(&gx[reflect_index(sb + k, a.n)], -w * v.y);
        }
    }
}

// ---- tables: twiddles (cos, -sin) and the (centre-padded) periodic Hann window -------------------
__global__ void k_stft_tables(float* tables, ResInfo r, int win_length) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= r.n_fft) return;
    const double ang = 6.283185307179586476925 * (double)t / (double)r.n_fft;
    tables[r.tw_off + 2 * t] = (float)cos(ang);
    tables[r.tw_off + 2 * t + 1] = (float)(-sin(ang));
    const int off = (r.n_fft - win_length) / 2;
    float w = 0.0f;
    if (t >= off && t < off + win_length) {
        // torch.hann_window(win_length) (periodic), evaluated in fp32 like torch does
        const float ph = 6.283185307179586f * (float)(t - off) / (float)win_length;
        w = 0.5f - 0.5f * (float)cos((double)ph);
    }
    tables[r.win_off + t] = w;
}

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
};
// stage 1: one 64-lane workgroup per (row, resolution) folds that row's strip partials (fp64, fixed order)
__global__ __launch_bounds__(64) void k_mrstft_rowsums(LossArgs a) {
    const int tid = threadIdx.x, row = blockIdx.x, res = blockIdx.y;
    const float* p = a.part + a.part_off[res] + (int64_t)row * a.n_groups[res] * 4;
    double s[4] = {0, 0, 0, 0};
    for (int g = tid; g < a.n_groups[res]; g += 64) {
        const float4 v = *reinterpret_cast<const float4*>(p + (int64_t)g * 4);
        s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
        for (int m = 32; m >= 1; m >>= 1) s[q] += __shfl_xor(s[q], m);
    if (tid == 0) {
        float* o = a.sums + ((int64_t)res * a.rows + row) * 4;
        o[0] = (float)s[0]; o[1] = (float)s[1]; o[2] = (float)s[2]; o[3] = (float)s[3];
    }
}
// stage 2: loss scalar + per-row backward coefficients (without dL/dloss, applied by k_scale_coef)
__global__ __launch_bounds__(64) void k_mrstft_final(LossArgs a) {
    __shared__ double rs[kMaxRes][4];
    const int tid = threadIdx.x;
    if (tid < a.n_res) {
        const int res = tid;
        double tot[4] = {0, 0, 0, 0}, sc_acc = 0.0;
        for (int row = 0; row < a.rows; ++row) {
            const float* sm = a.sums + ((int64_t)res * a.rows + row) * 4;
            for (int q = 0; q < 4; ++q) tot[q] += (double)sm[q];
            sc_acc += sqrt((double)sm[0]) / sqrt((double)sm[1]);
        }
        for (int q = 0; q < 4; ++q) rs[res][q] = tot[q];
        const double sc = a.sc_per_example ? sc_acc / a.rows : sqrt(tot[0]) / sqrt(tot[1]);
        rs[res][3] = a.w_sc * sc + a.w_log * tot[2] / a.count[res] + a.w_lin * tot[3] / a.count[res];
    }
    __syncthreads();
    if (tid == 0) {
        double total = 0.0;
        for (int res = 0; res < a.n_res; ++res) total += rs[res][3];
        a.loss[0] = (float)(total / a.n_res);
    }
    for (int i = tid; i < a.n_res * a.rows; i += 64) {
        const int res = i / a.rows;
        const float* sm = a.sums + (int64_t)i * 4;
        double c_sc;
        if (a.sc_per_example) c_sc = a.w_sc / ((double)a.rows * sqrt((double)sm[0]) * sqrt((double)sm[1]));
        else c_sc = a.w_sc / (sqrt(rs[res][0]) * sqrt(rs[res][1]));
        if (!(c_sc == c_sc) || c_sc > 1e30) c_sc = 0.0;  // identical signals: 0/0 -> no SC gradient
        float* c = a.coef + (int64_t)i * 4;
        c[0] = (float)(c_sc / a.n_res);
        c[1] = (float)(a.w_log / a.count[res] / a.n_res);
        c[2] = (float)(a.w_lin / a.count[res] / a.n_res);
        c[3] = 0.f;
    }
}
__global__ void k_scale_coef(const float* coef, const float* grad_loss, float* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = coef[i] * grad_loss[0];
}

}  // namespace mst

// ======================================================================================= C ABI
using namespace mst;

namespace {
struct Plan {
    ResInfo res[kMaxRes];
    int log2n[kMaxRes], n_groups[kMaxRes], win[kMaxRes];
    int64_t part_off[kMaxRes];
    int64_t tables_floats;
    // workspace (floats): part | sums | coef | coef_scaled
    int64_t part_total, sums_off, coef_off, coefs_off, ws_floats;
    bool ok;
};
Plan make_plan(const mst_mrstft_desc* d) {
    Plan p{};
    p.ok = false;
    if (!d || d->rows <= 0 || d->n_res <= 0 || d->n_res > kMaxRes) return p;
    int64_t t = 0, po = 0;
    for (int i = 0; i < d->n_res; ++i) {
        const int nf = d->fft_size[i];
        int lg = 0;
        while ((1 << lg) < nf) ++lg;
        if ((1 << lg) != nf || nf < 128 || nf > 8192) return p;
        if (d->hop_size[i] <= 0 || d->win_length[i] <= 0 || d->win_length[i] > nf) return p;
        if (d->n_samples <= nf / 2) return p;  // reflect padding needs n > n_fft/2 (torch.stft raises)
        ResInfo& r = p.res[i];
        r.n_fft = nf;
        r.hop = d->hop_size[i];
        r.n_frames = 1 + (int)(d->n_samples / r.hop);
        r.n_bins = nf / 2 + 1;
        r.tw_off = t;
        t += 2 * (int64_t)nf;
        r.win_off = t;
        t += nf;
        p.log2n[i] = lg;
        p.win[i] = d->win_length[i];
        // aim for >= ~2000 workgroups per launch (256 CUs x several resident) but cap the strip at 8 frames
        const int slots = stft_slots(nf);
        int fps = (int)(((int64_t)r.n_frames * d->rows) / (2048 * slots));
        r.frames_per_slot = fps < 1 ? 1 : (fps > 8 ? 8 : fps);
        p.n_groups[i] = (r.n_frames + r.frames_per_slot * slots - 1) / (r.frames_per_slot * slots);
        p.part_off[i] = po;
        po += (int64_t)d->rows * p.n_groups[i] * 4;
    }
    p.tables_floats = t;
    p.part_total = round_up(po, 64);
    p.sums_off = p.part_total;
    p.coef_off = p.sums_off + round_up((int64_t)d->n_res * d->rows * 4, 64);
    p.coefs_off = p.coef_off + round_up((int64_t)d->n_res * d->rows * 4, 64);
    p.ws_floats = p.coefs_off + round_up((int64_t)d->n_res * d->rows * 4, 64);
    p.ok = true;
    return p;
}
#define MST_FOR_NFFT(nf, CALL) \
    switch (nf) {               \
        case 128: CALL(128); break;   \
        case 256: CALL(256); break;   \
        case 512: CALL(512); break;   \
        case 1024: CALL(1024); break; \
        case 2048: CALL(2048); break; \
        case 4096: CALL(4096); break; \
        case 8192: CALL(8192); break; \
        default: break;               \
    }
}  // namespace

extern "C" size_t mst_mrstft_tables_bytes(const mst_mrstft_desc* d) {
    const Plan p = make_plan(d);
    return p.ok ? (size_t)p.tables_floats * sizeof(float) : 0;
}
extern "C" size_t mst_mrstft_workspace_bytes(const mst_mrstft_desc* d) {
    const Plan p = make_plan(d);
    return p.ok ? (size_t)p.ws_floats * sizeof(float) : 0;
}
extern "C" int mst_mrstft_init_tables(const mst_mrstft_desc* d, void* tables, void* stream_) {
    const Plan p = make_plan(d);
    if (!p.ok || !tables) return hipErrorInvalidValue;
    hipStream_t stream = (hipStream_t)stream_;
    for (int i = 0; i < d->n_res; ++i)
        hipLaunchKernelGGL(k_stft_tables, dim3((p.res[i].n_fft + 255) / 256), dim3(256), 0, stream, (float*)tables, p.res[i], p.win[i]);
    return (int)hipGetLastError();
}

extern "C" int mst_mrstft_forward(const mst_mrstft_desc* d, const float* pred, const float* target, const void* tables,
                                  float* loss, void* workspace, size_t workspace_bytes, void* stream_) {
    const Plan p = make_plan(d);
    if (!p.ok || !pred || !target || !tables || !loss || !workspace) return hipErrorInvalidValue;
    if (workspace_bytes < (size_t)p.ws_floats * sizeof(float)) return hipErrorInvalidValue;
    hipStream_t stream = (hipStream_t)stream_;
    float* ws = (float*)workspace;
    LossArgs la{};
    la.part = ws;
    la.sums = ws + p.sums_off;
    la.coef = ws + p.coef_off;
    la.loss = loss;
    la.n_res = d->n_res;
    la.rows = d->rows;
    la.w_sc = d->w_sc;
    la.w_log = d->w_log_mag;
    la.w_lin = d->w_lin_mag;
    la.sc_per_example = d->sc_per_example;
    for (int i = 0; i < d->n_res; ++i) {
        StftArgs a{};
        a.pred = pred;
        a.target = target;
        a.tables = (const float*)tables;
        a.part = ws + p.part_off[i];
        a.r = p.res[i];
        a.log2n = p.log2n[i];
        a.n = d->n_samples;
        a.eps = d->eps;
        const dim3 grid(p.n_groups[i], d->rows);
#define MST_LAUNCH_FWD(NF) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_stft_fwd<NF>), grid, dim3(stft_threads<NF>()), 0, stream, a)
        MST_FOR_NFFT(a.r.n_fft, MST_LAUNCH_FWD)
        la.n_groups[i] = p.n_groups[i];
        la.part_off[i] = p.part_off[i];
        la.count[i] = (float)((double)d->rows * p.res[i].n_bins * p.res[i].n_frames);
    }
    hipLaunchKernelGGL(k_mrstft_rowsums, dim3(d->rows, d->n_res), dim3(64), 0, stream, la);
    hipLaunchKernelGGL(k_mrstft_final, dim3(1), dim3(64), 0, stream, la);
    return (int)hipGetLastError();
}

extern "C" int mst_mrstft_backward(const mst_mrstft_desc* d, const float* pred, const float* target, const void* tables,
                                   const float* grad_loss, float* grad_pred, void* workspace, size_t workspace_bytes,
                                   void* stream_) {
    const Plan p = make_plan(d);
    if (!p.ok || !pred || !target || !tables || !grad_loss || !grad_pred || !workspace) return hipErrorInvalidValue;
    if (workspace_bytes < (size_t)p.ws_floats * sizeof(float)) return hipErrorInvalidValue;
    hipStream_t stream = (hipStream_t)stream_;
    float* ws = (float*)workspace;
    const int nc = d->n_res * d->rows * 4;
    hipLaunchKernelGGL(k_scale_coef, dim3((nc + 255) / 256), dim3(256), 0, stream, ws + p.coef_off, grad_loss, ws + p.coefs_off, nc);
    hipMemsetAsync(grad_pred, 0, (size_t)d->rows * d->n_samples * sizeof(float), stream);
    for (int i = 0; i < d->n_res; ++i) {
        StftArgs a{};
        a.pred = pred;
        a.target = target;
        a.tables = (const float*)tables;
        a.sums = ws + p.sums_off + (int64_t)i * d->rows * 4;
        a.coef = ws + p.coefs_off + (int64_t)i * d->rows * 4;
        a.grad_pred = grad_pred;
        a.r = p.res[i];
        a.log2n = p.log2n[i];
        a.n = d->n_samples;
        a.eps = d->eps;
        // three LDS buffers (pairing two frames per inverse FFT) fit up to n_fft = 4096; 8192 runs one frame per slot
        const bool pair = a.r.n_fft <= 4096;
        const int units = pair ? (a.r.n_frames + 1) / 2 : a.r.n_frames;
        const int slots = stft_slots(a.r.n_fft);
        const dim3 grid((units + slots - 1) / slots, d->rows);
#define MST_LAUNCH_BWD(NF)                                                                                             \
    if (NF <= 4096) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_stft_bwd<NF, (NF <= 4096)>), grid, dim3(stft_threads<NF>()), 0, stream, a); \
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_stft_bwd<NF, false>), grid, dim3(stft_threads<NF>()), 0, stream, a)
        MST_FOR_NFFT(a.r.n_fft, MST_LAUNCH_BWD)
    }
    return (int)hipGetLastError();
}
