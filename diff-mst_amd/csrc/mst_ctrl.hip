// mst_ctrl.hip - the TransformerController's encoder stack (reference mst/modules.py:809-914: torch.nn.TransformerEncoder of
// post-norm TransformerEncoderLayer(d_model, nhead, dim_feedforward = 2048, relu, dropout 0, batch_first), forward and backward.
//
// Why it exists: at the reference's sizes (one mix of 32 tracks = 36 tokens of width 512, 12 layers) the stack is ~580 library
// kernels of a few microseconds each per training step; on torch (rocBLAS + SDPA) it took 5.3 ms of the 25.6 ms cfg #5 step,
// hipGraph replay included.  M = bs x tokens is tiny, so every GEMM is a WEIGHT-STREAMING problem: 12.6 MB of fp32
// weights per layer are read once by the forward, once by the data gradient, and 12.6 MB of weight gradient are written.  The
// kernels are therefore organised around "every weight element crosses the memory system once, 16 bytes per lane":
//   k_lin_nt  C = act(A W^T + b) (+ R)   one workgroup per 16 output columns, its 8 waves split K, fragments straight from
//                                        global memory (both operands K-contiguous: float4 per lane, the K order inside a
//                                        16-step is permuted identically for A and W, which a dot product does not see)
//   k_lin_nn  C = (A W) (.) mask (+ R)   data gradient: one workgroup per 16 output columns, 8 waves split the reduction
//   k_lin_tn_batch  dW = dY^T X, db      (every weight gradient of a backward call in one launch) one wave per 64 x 64 tile (row and column permutations
//                                        make both operand loads and the stores 16 bytes per lane), bias gradient on the way
// all on v_mfma_f32_16x16x4_f32 (fp32 operands, exact fp32 FMA chains: parity with the reference's fp32 stack is rounding-level).
// Attention (<= 128 tokens, head width <= 64) and LayerNorm are small vector-ALU kernels.  No atomics: run-to-run deterministic.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/diffmst_hip.h"

namespace mst {
namespace ctrl {

typedef float f32x4 __attribute__((vector_size(16)));
#define MFMA4(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float comp(const float4& v, int j) { return j == 0 ? v.x : (j == 1 ? v.y : (j == 2 ? v.z : v.w)); }

// ---- C[m][n] = act(sum_k A[m][k] W[n][k] + bias[n]) (+ R[m][n]);  grid (N / 16, ceil(M / 64)), 512 lanes -------------------
template <bool RELU>
__global__ __launch_bounds__(512) void k_lin_nt(const float* __restrict__ A, int lda, const float* __restrict__ W, int ldw,
                                                const float* __restrict__ bias, const float* __restrict__ R, int ldr,
                                                float* __restrict__ C, int ldc, int M, int K, size_t part_stride) {
    __shared__ float red[8][4][4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.x * 16, m0 = blockIdx.y * 64;
    const int nmt = min(4, (M - m0 + 15) >> 4);
    // gridDim.z > 1: workgroup z multiplies the z-th slice of K into partial output z (the consuming LayerNorm kernel folds the
    // partials in a fixed order; bias and residual ride in partial 0) - a K = 2048 product on 32 workgroups is bound by the
    // fp32 matrix pipes of 32 CUs
    const int Kz = K / (int)gridDim.z;
    const int kq = Kz >> 3, kb = blockIdx.z * Kz + wave * kq;  // Kz % 128 == 0: every wave owns a multiple of 16
    C += blockIdx.z * part_stride;
    if (blockIdx.z) { bias = nullptr; R = nullptr; }
    f32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* wp = W + (int64_t)(n0 + i) * ldw + kb + 4 * g;
    const float* ap[4];
    bool ok[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int row = m0 + 16 * t + i;
        ok[t] = t < nmt && row < M;
        ap[t] = A + (int64_t)(ok[t] ? row : m0) * lda + kb + 4 * g;
    }
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < kq; k += 64) {  // four 16-steps per trip, every load of the trip issued before the first product
        float4 b4[4], a4[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool in = k + 16 * u < kq;
            b4[u] = in ? ld4(wp + k + 16 * u) : zero4;
#pragma unroll
            for (int t = 0; t < 4; ++t) a4[u][t] = (in && ok[t]) ? ld4(ap[t] + k + 16 * u) : zero4;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (t < nmt) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[t] = MFMA4(comp(a4[u][t], j), comp(b4[u], j), acc[t]);
                }
            }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave][t][r][lane] = acc[t][r];
    __syncthreads();
    const int t = wave >> 1;  // waves 2 t, 2 t + 1 finish row tile t (two accumulator rows each)
    if (t < nmt) {
        const float bv = bias ? bias[n0 + i] : 0.f;
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int r = 2 * (wave & 1) + rr;
            const int row = m0 + 16 * t + 4 * g + r;
            if (row < M) {
                float v = 0.f;
#pragma unroll
                for (int w = 0; w < 8; ++w) v += red[w][t][r][lane];
                v += bv;
                if (RELU) v = fmaxf(v, 0.f);
                if (R) v += R[(int64_t)row * ldr + n0 + i];
                C[(int64_t)row * ldc + n0 + i] = v;
            }
        }
    }
}

// ---- C[m][c] = (sum_n A[m][n] W[n][c]) (. [Hm[m][c] > 0]) (+ R[m][c]);  grid (cols / 16, ceil(M / 64)), 512 lanes ------------
__global__ __launch_bounds__(512) void k_lin_nn(const float* __restrict__ A, int lda, const float* __restrict__ W, int ldw,
                                                const float* __restrict__ Hm, int ldh, const float* __restrict__ R, int ldr,
                                                float* __restrict__ C, int ldc, int M, int N, size_t part_stride) {
    __shared__ float red[8][4][4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, g = lane >> 4;
    const int c0 = blockIdx.x * 16, m0 = blockIdx.y * 64;
    const int nmt = min(4, (M - m0 + 15) >> 4);
    const int Nz = N / (int)gridDim.z;  // reduction slices over grid.z -> partial outputs, as in k_lin_nt (no mask with partials)
    const int nq = Nz >> 3, nb = blockIdx.z * Nz + wave * nq;
    C += blockIdx.z * part_stride;
    if (blockIdx.z) R = nullptr;
    f32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* wp = W + (int64_t)(nb + 4 * g) * ldw + c0 + i;
    const float* ap[4];
    bool ok[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int row = m0 + 16 * t + i;
        ok[t] = t < nmt && row < M;
        ap[t] = A + (int64_t)(ok[t] ? row : m0) * lda + nb + 4 * g;
    }

    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int n = 0; n < nq; n += 64) {
        float b[4][4];
        float4 a4[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool in = n + 16 * u < nq;
#pragma unroll
            for (int j = 0; j < 4; ++j) b[u][j] = in ? wp[(int64_t)(n + 16 * u + j) * ldw] : 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t) a4[u][t] = (in && ok[t]) ? ld4(ap[t] + n + 16 * u) : zero4;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (t < nmt) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[t] = MFMA4(comp(a4[u][t], j), b[u][j], acc[t]);
                }
            }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave][t][r][lane] = acc[t][r];
    __syncthreads();
    const int t = wave >> 1;
    if (t < nmt) {
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int r = 2 * (wave & 1) + rr;
            const int row = m0 + 16 * t + 4 * g + r;
            if (row < M) {
                float v = 0.f;
#pragma unroll
                for (int w = 0; w < 8; ++w) v += red[w][t][r][lane];
                if (Hm && !(Hm[(int64_t)row * ldh + c0 + i] > 0.f)) v = 0.f;
                if (R) v += R[(int64_t)row * ldr + c0 + i];
                C[(int64_t)row * ldc + c0 + i] = v;
            }
        }
    }
}

// ---- dW[n][k] = sum_m dY[m][n] X[m][k],  db[n] = sum_m dY[m][n];  grid (Kd / 64, N / 64), one wave --------------------------
// Row tile ii of the wave holds the rows n0 + 4 r + ii (r = 0..15), column tile t the columns k0 + 4 c + t: a lane's float4 of
// dY is one element of each of the four row tiles, its float4 of X one element of each column tile.
__device__ __forceinline__ void lin_tn_tile(const float* __restrict__ dY, int ldy, const float* __restrict__ X, int ldx,
                                            float* __restrict__ dW, int ldw, float* __restrict__ db, int M, int bx, int by) {
    const int lane = threadIdx.x, c = lane & 15, g = lane >> 4;
    const int k0 = bx * 64, n0 = by * 64;
    f32x4 acc[4][4];
#pragma unroll
    for (int ii = 0; ii < 4; ++ii)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[ii][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int m = 0; m < M; m += 16) {  // four 4-row steps per trip, loads first
        float4 a4[4], b4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int row = m + 4 * u + g;
            a4[u] = b4[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < M) {
                a4[u] = ld4(dY + (int64_t)row * ldy + n0 + 4 * c);
                b4[u] = ld4(X + (int64_t)row * ldx + k0 + 4 * c);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            bsum.x += a4[u].x; bsum.y += a4[u].y; bsum.z += a4[u].z; bsum.w += a4[u].w;
#pragma unroll
            for (int ii = 0; ii < 4; ++ii)
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[ii][t] = MFMA4(comp(a4[u], ii), comp(b4[u], t), acc[ii][t]);
        }
    }
#pragma unroll
    for (int ii = 0; ii < 4; ++ii)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int n = n0 + 4 * (4 * g + q) + ii;
            *reinterpret_cast<float4*>(dW + (int64_t)n * ldw + k0 + 4 * c) =
                make_float4(acc[ii][0][q], acc[ii][1][q], acc[ii][2][q], acc[ii][3][q]);
        }
    if (db && bx == 0) {
        float v[4] = {bsum.x, bsum.y, bsum.z, bsum.w};
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
            v[ii] += __shfl_xor(v[ii], 16);
            v[ii] += __shfl_xor(v[ii], 32);
        }
        if (g == 0) *reinterpret_cast<float4*>(db + n0 + 4 * c) = make_float4(v[0], v[1], v[2], v[3]);
    }
}

// Every weight gradient of a backward call in ONE launch: the 4 x n_layers products are off the critical path (nothing in the
// backward reads a weight gradient), so they wait until every dY has been formed and then fill the chip together
// (12 layers: 9216 tiles) instead of 48 launches of 64-256 tiles.  Jobs ride in the kernel arguments.
constexpr int kMaxTnJobs = 64;
struct TnJob {
    const float* dY;
    const float* X;
    float* dW;
    float* db;
    int ldy, ldx, ldw, tiles_x, tile0;  // tile0 = first workgroup of the job
};
struct TnBatch {
    TnJob job[kMaxTnJobs];
    int n_jobs, M;
};
__global__ __launch_bounds__(64) void k_lin_tn_batch(const TnBatch b) {
    int j = 0, hi = b.n_jobs - 1;  // wave-uniform binary search over the job table (kernel arguments: scalar loads)
    while (j < hi) {
        const int mid = (j + hi + 1) >> 1;
        if ((int)blockIdx.x >= b.job[mid].tile0) j = mid;
        else hi = mid - 1;
    }
    const TnJob& J = b.job[j];
    const int t = blockIdx.x - J.tile0;
    lin_tn_tile(J.dY, J.ldy, J.X, J.ldx, J.dW, J.ldw, J.db, b.M, t % J.tiles_x, t / J.tiles_x);
}

// ---- attention ----------------------------------------------------------------------------------------------------------
constexpr int kMaxS = 128, kMaxDh = 64, kPitch = kMaxDh + 1;
constexpr int kRowsPerWg = 8;  // query rows (key rows in k_attn_bwd2) of one workgroup: grid.z = ceil(S / 8)

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// qkv (bs S, 3 d) -> P (bs, H, S, S) softmax probabilities, ctx (bs S, d).  grid (H, bs, ceil(S / 8)), 256 lanes; a wave per query row.
__global__ __launch_bounds__(256) void k_attn_fwd(const float* __restrict__ qkv, const uint8_t* __restrict__ mask, float* __restrict__ P,
                                                  float* __restrict__ ctx, int S, int d, int dh, float scale) {
    extern __shared__ float sm[];
    float* q = sm;
    float* k = q + S * kPitch;
    float* v = k + S * kPitch;
    float* pw = v + S * kPitch;  // [4][kMaxS]
    const int h = blockIdx.x, b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* base = qkv + (int64_t)b * S * 3 * d + h * dh;
    for (int e = threadIdx.x; e < S * dh; e += 256) {
        const int r = e / dh, c = e - r * dh;
        const float* p = base + (int64_t)r * 3 * d + c;
        q[r * kPitch + c] = p[0] * scale;  // torch scales q before the product
        k[r * kPitch + c] = p[d];
        v[r * kPitch + c] = p[2 * d];
    }
    __syncthreads();
    const uint8_t* mk = mask ? mask + (int64_t)b * S : nullptr;
    for (int r = blockIdx.z * kRowsPerWg + wave; r < min(S, (int)(blockIdx.z + 1) * kRowsPerWg); r += 4) {
        float s[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int j = lane + 64 * u;
            float a = -INFINITY;
            if (j < S && !(mk && mk[j])) {
                a = 0.f;
                for (int c = 0; c < dh; ++c) a = fmaf(q[r * kPitch + c], k[j * kPitch + c], a);
            }
            s[u] = a;
        }
        const float mx = wave_max(fmaxf(s[0], s[1]));
        const float e0 = s[0] == -INFINITY ? 0.f : __expf(s[0] - mx), e1 = s[1] == -INFINITY ? 0.f : __expf(s[1] - mx);
        const float inv = 1.0f / wave_sum(e0 + e1);
        float* prow = P + (((int64_t)b * gridDim.x + h) * S + r) * S;
        if (lane < S) { pw[wave * kMaxS + lane] = e0 * inv; prow[lane] = e0 * inv; }
        if (lane + 64 < S) { pw[wave * kMaxS + lane + 64] = e1 * inv; prow[lane + 64] = e1 * inv; }
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the wave's own LDS writes have landed (wave-local data, no barrier)
        __builtin_amdgcn_wave_barrier();
        if (lane < dh) {
            float o = 0.f;
            for (int j = 0; j < S; ++j) o = fmaf(pw[wave * kMaxS + j], v[j * kPitch + lane], o);
            ctx[((int64_t)b * S + r) * d + h * dh + lane] = o;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// dS = P (.) (dP - rowsum(dP (.) P)) with dP = dO V^T, written unscaled.  grid (H, bs, ceil(S / 8)), 256 lanes
__global__ __launch_bounds__(256) void k_attn_bwd1(const float* __restrict__ qkv, const float* __restrict__ dctx, const float* __restrict__ P,
                                                   float* __restrict__ dS, int S, int d, int dh) {
    extern __shared__ float sm[];
    float* v = sm;
    float* go = v + S * kPitch;
    const int h = blockIdx.x, b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int e = threadIdx.x; e < S * dh; e += 256) {
        const int r = e / dh, c = e - r * dh;
        v[r * kPitch + c] = qkv[((int64_t)b * S + r) * 3 * d + 2 * d + h * dh + c];
        go[r * kPitch + c] = dctx[((int64_t)b * S + r) * d + h * dh + c];
    }
    __syncthreads();
    for (int r = blockIdx.z * kRowsPerWg + wave; r < min(S, (int)(blockIdx.z + 1) * kRowsPerWg); r += 4) {
        const int64_t off = (((int64_t)b * gridDim.x + h) * S + r) * S;
        float dp[2], p[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int j = lane + 64 * u;
            dp[u] = 0.f;
            p[u] = 0.f;
            if (j < S) {
                p[u] = P[off + j];
                float a = 0.f;
                for (int c = 0; c < dh; ++c) a = fmaf(go[r * kPitch + c], v[j * kPitch + c], a);
                dp[u] = a;
            }
        }
        const float delta = wave_sum(dp[0] * p[0] + dp[1] * p[1]);
        if (lane < S) dS[off + lane] = p[0] * (dp[0] - delta);
        if (lane + 64 < S) dS[off + lane + 64] = p[1] * (dp[1] - delta);
    }
}

// dQ = scale dS K, dK = scale dS^T Q, dV = P^T dO -> dqkv (bs S, 3 d).  grid (H, bs, ceil(S / 8)), 256 lanes; lane = head column
__global__ __launch_bounds__(256) void k_attn_bwd2(const float* __restrict__ qkv, const float* __restrict__ dctx, const float* __restrict__ P,
                                                   const float* __restrict__ dS, float* __restrict__ dqkv, int S, int d, int dh, float scale) {
    extern __shared__ float sm[];
    float* q = sm;
    float* k = q + S * kPitch;
    float* go = k + S * kPitch;
    float* pw = go + S * kPitch;  // [4][2][kMaxS]
    const int h = blockIdx.x, b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* base = qkv + (int64_t)b * S * 3 * d + h * dh;
    for (int e = threadIdx.x; e < S * dh; e += 256) {
        const int r = e / dh, c = e - r * dh;
        q[r * kPitch + c] = base[(int64_t)r * 3 * d + c];
        k[r * kPitch + c] = base[(int64_t)r * 3 * d + d + c];
        go[r * kPitch + c] = dctx[((int64_t)b * S + r) * d + h * dh + c];
    }
    __syncthreads();
    const int64_t hb = ((int64_t)b * gridDim.x + h) * S * S;
    float* w0 = pw + wave * 2 * kMaxS;
    float* w1 = w0 + kMaxS;
    for (int r = blockIdx.z * kRowsPerWg + wave; r < min(S, (int)(blockIdx.z + 1) * kRowsPerWg); r += 4) {
        // row r of dS -> dQ[r]; column r of dS -> dK[r]; column r of P -> dV[r]
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int j = lane + 64 * u;
            if (j < S) w0[j] = dS[hb + (int64_t)r * S + j];
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        float dq = 0.f;
        if (lane < dh)
            for (int j = 0; j < S; ++j) dq = fmaf(w0[j], k[j * kPitch + lane], dq);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int j = lane + 64 * u;
            if (j < S) {
                w0[j] = dS[hb + (int64_t)j * S + r];
                w1[j] = P[hb + (int64_t)j * S + r];
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        if (lane < dh) {
            float dk = 0.f, dv = 0.f;
            for (int j = 0; j < S; ++j) {
                dk = fmaf(w0[j], q[j * kPitch + lane], dk);
                dv = fmaf(w1[j], go[j * kPitch + lane], dv);
            }
            float* o = dqkv + ((int64_t)b * S + r) * 3 * d + h * dh + lane;
            o[0] = dq * scale;
            o[d] = dk * scale;
            o[2 * d] = dv;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- LayerNorm (rows of width d <= 1024, one wave per row) --------------------------------------------------------------
constexpr int kLnMax = 16;  // columns per lane
constexpr int kMaxSplit = 4;  // partial buffers a LayerNorm kernel folds on load

// s = the sum of `np` partial buffers `part_stride` floats apart (np > 1: folded here, in order, and written to s_out)
__global__ __launch_bounds__(256) void k_ln_fwd(const float* __restrict__ s, int np, size_t part_stride, float* __restrict__ s_out,
                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                float* __restrict__ y, float* __restrict__ stats, int M, int d, float eps) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* x = s + (int64_t)row * d;
    float v[kLnMax];
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < kLnMax; ++t) {
        const int c = lane + 64 * t;
        v[t] = c < d ? x[c] : 0.f;
    }
    if (np > 1) {  // partials 1 .. np - 1: a whole row of independent loads per partial, added in order
#pragma unroll
        for (int q = 1; q < kMaxSplit; ++q) {
            if (q < np) {
                float u[kLnMax];
#pragma unroll
                for (int t = 0; t < kLnMax; ++t) {
                    const int c = lane + 64 * t;
                    u[t] = c < d ? x[q * part_stride + c] : 0.f;
                }
#pragma unroll
                for (int t = 0; t < kLnMax; ++t) v[t] += u[t];
            }
        }
#pragma unroll
        for (int t = 0; t < kLnMax; ++t) {
            const int c = lane + 64 * t;
            if (c < d) s_out[(int64_t)row * d + c] = v[t];
        }
    }
#pragma unroll
    for (int t = 0; t < kLnMax; ++t) sum += v[t];
    const float mean = wave_sum(sum) / (float)d;
    float sq = 0.f;
#pragma unroll
    for (int t = 0; t < kLnMax; ++t) {
        const int c = lane + 64 * t;
        const float u = c < d ? v[t] - mean : 0.f;
        sq = fmaf(u, u, sq);
    }
    const float rstd = 1.0f / sqrtf(wave_sum(sq) / (float)d + eps);
#pragma unroll
    for (int t = 0; t < kLnMax; ++t) {
        const int c = lane + 64 * t;
        if (c < d) y[(int64_t)row * d + c] = (v[t] - mean) * rstd * gamma[c] + beta[c];
    }
    if (lane == 0) {
        stats[2 * row] = mean;
        stats[2 * row + 1] = rstd;
    }
}

__device__ __forceinline__ float fold_parts(const float* __restrict__ p, int np, size_t part_stride) {
    float u[kMaxSplit];
#pragma unroll
    for (int q = 0; q < kMaxSplit; ++q) u[q] = q < np ? p[q * part_stride] : 0.f;  // independent loads, fixed order of addition
    return ((u[0] + u[1]) + u[2]) + u[3];
}
__device__ __forceinline__ void ln_bwd_dx(const float* __restrict__ dy, int np, size_t part_stride, const float* __restrict__ s,
                                          const float* __restrict__ stats, const float* __restrict__ gamma, float* __restrict__ dx, int M,
                                          int d, int blk) {
    const int lane = threadIdx.x & 63, row = blk * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float mean = stats[2 * row], rstd = stats[2 * row + 1];
    float xh[kLnMax], gh[kLnMax];
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int t = 0; t < kLnMax; ++t) {
        const int c = lane + 64 * t;
        xh[t] = gh[t] = 0.f;
        if (c < d) {
            xh[t] = (s[(int64_t)row * d + c] - mean) * rstd;
            gh[t] = fold_parts(dy + (int64_t)row * d + c, np, part_stride) * gamma[c];
        }
        c1 += gh[t];
        c2 = fmaf(gh[t], xh[t], c2);
    }
    c1 = wave_sum(c1) / (float)d;
    c2 = wave_sum(c2) / (float)d;
#pragma unroll
    for (int t = 0; t < kLnMax; ++t) {
        const int c = lane + 64 * t;
        if (c < d) dx[(int64_t)row * d + c] = rstd * (gh[t] - c1 - xh[t] * c2);
    }
}

// dgamma[c] = sum_rows dy xhat, dbeta[c] = sum_rows dy: 64 columns per workgroup, 4 row groups folded through LDS in a fixed order
__device__ __forceinline__ void ln_bwd_gb(const float* __restrict__ dy, int np, size_t part_stride, const float* __restrict__ s,
                                          const float* __restrict__ stats, float* __restrict__ dgamma, float* __restrict__ dbeta, int M, int d,
                                          int blk) {
    __shared__ float part[2][4][64];
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6, c = blk * 64 + lane;
    float a = 0.f, b = 0.f;
    if (c < d)
        for (int row = grp; row < M; row += 4) {
            const float g = fold_parts(dy + (int64_t)row * d + c, np, part_stride);
            a = fmaf(g, (s[(int64_t)row * d + c] - stats[2 * row]) * stats[2 * row + 1], a);
            b += g;
        }
    part[0][grp][lane] = a;
    part[1][grp][lane] = b;
    __syncthreads();
    if (grp == 0 && c < d) {
        dgamma[c] = (part[0][0][lane] + part[0][1][lane]) + (part[0][2][lane] + part[0][3][lane]);
        dbeta[c] = (part[1][0][lane] + part[1][1][lane]) + (part[1][2][lane] + part[1][3][lane]);
    }
}

// one launch: workgroups [0, ceil(M / 4)) form dx (a wave per row), the next ceil(d / 64) the column sums
// (dy = the sum of `np` partial buffers, like k_ln_fwd's input)
__global__ __launch_bounds__(256) void k_ln_bwd(const float* __restrict__ dy, int np, size_t part_stride, const float* __restrict__ s,
                                                const float* __restrict__ stats, const float* __restrict__ gamma, float* __restrict__ dx,
                                                float* __restrict__ dgamma, float* __restrict__ dbeta, int M, int d) {
    const int nrow = (M + 3) >> 2;
    if ((int)blockIdx.x < nrow) ln_bwd_dx(dy, np, part_stride, s, stats, gamma, dx, M, d, blockIdx.x);
    else ln_bwd_gb(dy, np, part_stride, s, stats, dgamma, dbeta, M, d, blockIdx.x - nrow);
}

// ---- host side ------------------------------------------------------------------------------------------------------------
// reduction slices of a product whose consumer folds partials: 512 per slice from 1024 up (2048 -> 4, 1536 -> 3)
static int split_of(int K) { return (K >= 1024 && K % 512 == 0 && K / 512 <= kMaxSplit) ? K / 512 : 1; }

struct Plan {
    int M, S, d, H, dh, ff, L;
    size_t qkv, P, ctx, s1, st1, x1, h, s2, st2, x2, b_ds2, b_dh, b_ds1, b_dqkv, per_layer;  // float offsets inside one layer's slab
                                                                                  // (b_*: the backward's dY operands, kept for the batched weight gradient)
    size_t t_part, t_g0, t_g1, t_dx1, t_dctx, t_dS, part, total;  // temporaries (floats, after the slabs); t_part / t_g* / t_dx1 hold
                                                                  // up to kMaxSplit partial buffers `part` floats apart
};

static bool make_plan(const mst_ctrl_desc* d, Plan& p) {
    if (!d || d->bs < 1 || d->seq < 1 || d->seq > kMaxS || d->n_layers < 1 || 4 * d->n_layers > kMaxTnJobs || d->nhead < 1) return false;
    if (d->d_model % 128 || d->d_ff % 128 || d->d_model > 64 * kLnMax || d->d_model % d->nhead) return false;
    p.dh = d->d_model / d->nhead;
    if (p.dh > kMaxDh) return false;
    p.M = d->bs * d->seq; p.S = d->seq; p.d = d->d_model; p.H = d->nhead; p.ff = d->d_ff; p.L = d->n_layers;
    auto up = [](size_t v) { return (v + 63) & ~(size_t)63; };
    size_t o = 0;
    const size_t M = p.M, dm = p.d;
    p.qkv = o; o += up(M * 3 * dm);
    p.P = o; o += up((size_t)d->bs * p.H * p.S * p.S);
    p.ctx = o; o += up(M * dm);
    p.s1 = o; o += up(M * dm);
    p.st1 = o; o += up(2 * M);
    p.x1 = o; o += up(M * dm);
    p.h = o; o += up(M * p.ff);
    p.s2 = o; o += up(M * dm);
    p.st2 = o; o += up(2 * M);
    p.x2 = o; o += up(M * dm);
    p.b_ds2 = o; o += up(M * dm);
    p.b_dh = o; o += up(M * p.ff);
    p.b_ds1 = o; o += up(M * dm);
    p.b_dqkv = o; o += up(M * 3 * dm);
    p.per_layer = o;
    o = p.per_layer * p.L;
    p.part = up(M * dm);
    p.t_part = o; o += kMaxSplit * p.part;
    p.t_g0 = o; o += kMaxSplit * p.part;
    p.t_g1 = o; o += kMaxSplit * p.part;
    p.t_dx1 = o; o += kMaxSplit * p.part;
    p.t_dctx = o; o += up(M * dm);
    p.t_dS = o; o += up((size_t)d->bs * p.H * p.S * p.S);
    p.total = o;
    return true;
}

static void lin_nt(bool relu, const float* A, int lda, const float* W, int ldw, const float* bias, const float* R, int ldr, float* C, int ldc,
                   int M, int N, int K, hipStream_t st, int split = 1, size_t part_stride = 0) {
    const dim3 grid(N / 16, (M + 63) / 64, split);
    if (relu) hipLaunchKernelGGL(k_lin_nt<true>, grid, dim3(512), 0, st, A, lda, W, ldw, bias, R, ldr, C, ldc, M, K, part_stride);
    else hipLaunchKernelGGL(k_lin_nt<false>, grid, dim3(512), 0, st, A, lda, W, ldw, bias, R, ldr, C, ldc, M, K, part_stride);
}
static void lin_nn(const float* A, int lda, const float* W, int ldw, const float* Hm, int ldh, const float* R, int ldr, float* C, int ldc, int M,
                   int N, int cols, hipStream_t st, int split = 1, size_t part_stride = 0) {
    hipLaunchKernelGGL(k_lin_nn, dim3(cols / 16, (M + 63) / 64, split), dim3(512), 0, st, A, lda, W, ldw, Hm, ldh, R, ldr, C, ldc, M, N, part_stride);
}
static void add_tn(TnBatch& b, int& tiles, const float* dY, int ldy, const float* X, int ldx, float* dW, int ldw, float* db, int N, int Kd) {
    TnJob& J = b.job[b.n_jobs++];
    J.dY = dY; J.X = X; J.dW = dW; J.db = db;
    J.ldy = ldy; J.ldx = ldx; J.ldw = ldw; J.tiles_x = Kd / 64; J.tile0 = tiles;
    tiles += (Kd / 64) * (N / 64);
}

}  // namespace ctrl
}  // namespace mst

using namespace mst::ctrl;

extern "C" size_t mst_ctrl_workspace_bytes(const mst_ctrl_desc* d) {
    Plan p;
    return make_plan(d, p) ? p.total * sizeof(float) : 0;
}

extern "C" int mst_ctrl_forward(const mst_ctrl_desc* d, const float* tokens, const uint8_t* key_padding_mask, const mst_ctrl_layer* layers,
                                float* out, void* workspace, size_t workspace_bytes, void* stream_) {
    Plan p;
    if (!make_plan(d, p) || !tokens || !layers || !out || !workspace) return hipErrorInvalidValue;
    if (workspace_bytes < p.total * sizeof(float)) return hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream_;
    float* ws = (float*)workspace;
    const int M = p.M, dm = p.d, ff = p.ff;
    const float scale = 1.0f / sqrtf((float)p.dh);
    const size_t lds_attn = ((size_t)3 * p.S * kPitch + 4 * kMaxS) * sizeof(float);
    if (lds_attn > 64 * 1024)  // sequences past ~80 tokens: raise the kernel's dynamic-LDS cap (idempotent; nothing is cached here)
        (void)hipFuncSetAttribute((const void*)k_attn_fwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_attn);
    const float* x = tokens;
    for (int l = 0; l < p.L; ++l) {
        float* L = ws + p.per_layer * l;
        const mst_ctrl_layer& w = layers[l];
        float* x2 = l == p.L - 1 ? out : L + p.x2;
        lin_nt(false, x, dm, w.in_proj_weight, dm, w.in_proj_bias, nullptr, 0, L + p.qkv, 3 * dm, M, 3 * dm, dm, st);
        hipLaunchKernelGGL(k_attn_fwd, dim3(p.H, d->bs, (p.S + kRowsPerWg - 1) / kRowsPerWg), dim3(256), lds_attn, st, L + p.qkv, key_padding_mask, L + p.P, L + p.ctx, p.S, dm, p.dh, scale);
        lin_nt(false, L + p.ctx, dm, w.out_proj_weight, dm, w.out_proj_bias, x, dm, L + p.s1, dm, M, dm, dm, st);
        hipLaunchKernelGGL(k_ln_fwd, dim3((M + 3) / 4), dim3(256), 0, st, L + p.s1, 1, (size_t)0, (float*)nullptr, w.norm1_weight, w.norm1_bias, L + p.x1, L + p.st1, M, dm, d->ln_eps);
        lin_nt(true, L + p.x1, dm, w.linear1_weight, dm, w.linear1_bias, nullptr, 0, L + p.h, ff, M, ff, dm, st);
        const int sp = split_of(ff);  // feed-forward output: K slices into partials, folded (and stored as s2) by the LayerNorm kernel
        float* s2p = sp > 1 ? ws + p.t_part : L + p.s2;
        lin_nt(false, L + p.h, ff, w.linear2_weight, ff, w.linear2_bias, L + p.x1, dm, s2p, dm, M, dm, ff, st, sp, p.part);
        hipLaunchKernelGGL(k_ln_fwd, dim3((M + 3) / 4), dim3(256), 0, st, s2p, sp, p.part, L + p.s2, w.norm2_weight, w.norm2_bias, x2, L + p.st2, M, dm, d->ln_eps);
        x = x2;
    }
    return (int)hipGetLastError();
}

extern "C" int mst_ctrl_backward(const mst_ctrl_desc* d, const float* tokens, const mst_ctrl_layer* layers, const float* grad_out,
                                 const mst_ctrl_layer_grads* grads, float* grad_tokens, void* workspace, size_t workspace_bytes, void* stream_) {
    Plan p;
    if (!make_plan(d, p) || !tokens || !layers || !grad_out || !grads || !grad_tokens || !workspace) return hipErrorInvalidValue;
    if (workspace_bytes < p.total * sizeof(float)) return hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream_;
    float* ws = (float*)workspace;
    const int M = p.M, dm = p.d, ff = p.ff;
    const float scale = 1.0f / sqrtf((float)p.dh);
    const size_t lds1 = (size_t)2 * p.S * kPitch * sizeof(float), lds2 = ((size_t)3 * p.S * kPitch + 8 * kMaxS) * sizeof(float);
    const dim3 lnb((M + 3) / 4 + (dm + 63) / 64);
    if (lds2 > 64 * 1024) {
        (void)hipFuncSetAttribute((const void*)k_attn_bwd1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1);
        (void)hipFuncSetAttribute((const void*)k_attn_bwd2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
    }
    const float* g = grad_out;
    int g_parts = 1;  // partial buffers behind `g`
    TnBatch tb;  // filled here, passed by value (3.6 KB of kernel arguments)
    tb.n_jobs = 0;
    tb.M = M;
    int tiles = 0;
    for (int l = p.L - 1; l >= 0; --l) {
        float* L = ws + p.per_layer * l;
        const mst_ctrl_layer& w = layers[l];
        const mst_ctrl_layer_grads& gw = grads[l];
        const float* x = l == 0 ? tokens : ws + p.per_layer * (l - 1) + p.x2;
        float* gout = l == 0 ? grad_tokens : ws + ((l & 1) ? p.t_g1 : p.t_g0);
        const int sp_out = l == 0 ? 1 : split_of(3 * dm), sp_ff = split_of(ff);  // partials of this layer's dx and of dx1
        float *ds2 = L + p.b_ds2, *dh = L + p.b_dh, *ds1 = L + p.b_ds1, *dqkv = L + p.b_dqkv;
        float *dx1 = ws + p.t_dx1, *dctx = ws + p.t_dctx, *dS = ws + p.t_dS;
        // LayerNorm 2 (input s2 = x1 + ffn)
        hipLaunchKernelGGL(k_ln_bwd, lnb, dim3(256), 0, st, g, g_parts, p.part, L + p.s2, L + p.st2, w.norm2_weight, ds2, gw.norm2_weight, gw.norm2_bias, M, dm);
        // feed-forward
        lin_nn(ds2, dm, w.linear2_weight, ff, L + p.h, ff, nullptr, 0, dh, ff, M, dm, ff, st);
        lin_nn(dh, ff, w.linear1_weight, dm, nullptr, 0, ds2, dm, dx1, dm, M, ff, dm, st, sp_ff, p.part);
        // LayerNorm 1 (input s1 = x + attention)
        hipLaunchKernelGGL(k_ln_bwd, lnb, dim3(256), 0, st, dx1, sp_ff, p.part, L + p.s1, L + p.st1, w.norm1_weight, ds1, gw.norm1_weight, gw.norm1_bias, M, dm);
        // attention
        lin_nn(ds1, dm, w.out_proj_weight, dm, nullptr, 0, nullptr, 0, dctx, dm, M, dm, dm, st);
        hipLaunchKernelGGL(k_attn_bwd1, dim3(p.H, d->bs, (p.S + kRowsPerWg - 1) / kRowsPerWg), dim3(256), lds1, st, L + p.qkv, dctx, L + p.P, dS, p.S, dm, p.dh);
        hipLaunchKernelGGL(k_attn_bwd2, dim3(p.H, d->bs, (p.S + kRowsPerWg - 1) / kRowsPerWg), dim3(256), lds2, st, L + p.qkv, dctx, L + p.P, dS, dqkv, p.S, dm, p.dh, scale);
        lin_nn(dqkv, 3 * dm, w.in_proj_weight, dm, nullptr, 0, ds1, dm, gout, dm, M, 3 * dm, dm, st, sp_out, p.part);
        g_parts = sp_out;
        // the layer's weight gradients: queued for the batched launch
        add_tn(tb, tiles, ds2, dm, L + p.h, ff, gw.linear2_weight, ff, gw.linear2_bias, dm, ff);
        add_tn(tb, tiles, dh, ff, L + p.x1, dm, gw.linear1_weight, dm, gw.linear1_bias, ff, dm);
        add_tn(tb, tiles, ds1, dm, L + p.ctx, dm, gw.out_proj_weight, dm, gw.out_proj_bias, dm, dm);
        add_tn(tb, tiles, dqkv, 3 * dm, x, dm, gw.in_proj_weight, dm, gw.in_proj_bias, 3 * dm, dm);
        g = gout;
    }
    hipLaunchKernelGGL(k_lin_tn_batch, dim3(tiles), dim3(64), 0, st, tb);
    return (int)hipGetLastError();
}

// =====================================================================================================================
// The controller's own ends (reference mst/modules.py:841-859, :866-914): token sequence in, three sigmoid heads out.
// Tiny, launch-bound work that used to be ~40 torch kernels and 7 rocBLAS GEMMs per step (profiles/round3_cfg5.md):
//   tokens  = cat(track_embeds + track_embedding, mix_embeds + mix_embedding, fx_bus_embedding, master_bus_embedding)   (+ the mask
//             extended by four always-attended tokens)
//   heads   = sigmoid(Linear) of the track tokens / the fx token (seq - 2) / the master token (seq - 1)
// fp32 vector arithmetic, a wave per token row, fixed summation orders (deterministic).
namespace mst {
namespace ctrl {

__global__ __launch_bounds__(128) void k_ctrl_tokens(const float* __restrict__ te, const float* __restrict__ me, const uint8_t* __restrict__ mask,
                                                     mst_ctrl_io io, float* __restrict__ tokens, uint8_t* __restrict__ mask_out, int T, int D) {
    const int s = blockIdx.x, b = blockIdx.y, S = T + 4;
    float* o = tokens + ((size_t)b * S + s) * D;
    for (int k = threadIdx.x; k < D; k += 128) {
        float v;
        if (s < T) v = te[((size_t)b * T + s) * D + k] + io.track_embedding[k];
        else if (s < T + 2) v = me[((size_t)b * 2 + (s - T)) * D + k] + io.mix_embedding[(s - T) * D + k];
        else v = (s == T + 2 ? io.fx_bus_embedding : io.master_bus_embedding)[k];
        o[k] = v;
    }
    if (mask_out && threadIdx.x == 0) mask_out[(size_t)b * S + s] = (mask && s < T) ? (mask[(size_t)b * T + s] ? 1 : 0) : 0;
}
// head of row r: rows [0, bs T) track tokens, [bs T, bs T + bs) fx tokens, then master tokens
struct HeadSel { const float* W; const float* B; float* out; const float* g; int n; size_t tok; size_t orow; };
__device__ __forceinline__ HeadSel head_of(int r, int bs, int T, const mst_ctrl_io& io, int nt, int nf, int nm, float* ot, float* of, float* om,
                                           const float* gt, const float* gf, const float* gm) {
    HeadSel h;
    const int S = T + 4;
    if (r < bs * T) {
        const int b = r / T, t = r % T;
        h = {io.track_w, io.track_b, ot, gt, nt, (size_t)b * S + t, (size_t)r};
    } else if (r < bs * T + bs) {
        const int b = r - bs * T;
        h = {io.fx_w, io.fx_b, of, gf, nf, (size_t)b * S + T + 2, (size_t)b};
    } else {
        const int b = r - bs * T - bs;
        h = {io.master_w, io.master_b, om, gm, nm, (size_t)b * S + T + 3, (size_t)b};
    }
    return h;
}
constexpr int kHeadK = 16;  // d_model <= 1024 = 64 lanes x 16
// one wave per (row, output): 16 independent loads per operand, one wave sum (a loop over the 27 outputs inside one wave was 27 serial
// memory round trips)
__global__ __launch_bounds__(64) void k_ctrl_heads_fwd(const float* __restrict__ z, mst_ctrl_io io, int nt, int nf, int nm, float* ot, float* of,
                                                       float* om, int bs, int T, int D) {
    const int lane = threadIdx.x, o = blockIdx.y;
    const HeadSel h = head_of(blockIdx.x, bs, T, io, nt, nf, nm, ot, of, om, nullptr, nullptr, nullptr);
    if (o >= h.n) return;
    const float* x = z + h.tok * D;
    const float* w = h.W + (size_t)o * D;
    float acc = 0.0f;
#pragma unroll
    for (int j = 0; j < kHeadK; ++j)
        if (lane + 64 * j < D) acc = fmaf(w[lane + 64 * j], x[lane + 64 * j], acc);
    acc = wave_sum(acc) + h.B[o];
    if (lane == 0) h.out[h.orow * h.n + o] = 1.0f / (1.0f + __expf(-acc));
}
// grad_z rows (every token row of the sequence is written: the mix tokens get zeros) and the pre-activation cotangents
// dpre (rows_total x 32) that the weight-gradient kernel sums over
__global__ __launch_bounds__(64) void k_ctrl_heads_bwd_dz(mst_ctrl_io io, int nt, int nf, int nm, const float* ot, const float* of, const float* om,
                                                          const float* gt, const float* gf, const float* gm, float* __restrict__ dz,
                                                          float* __restrict__ dpre, int bs, int T, int D) {
    const int lane = threadIdx.x, S = T + 4, r = blockIdx.x, rows = bs * (T + 2);
    if (r >= rows) {  // the 2 bs mix-token rows: no head reads them
        const int q = r - rows, b = q / 2, s = T + (q & 1);
        for (int k = lane; k < D; k += 64) dz[((size_t)b * S + s) * D + k] = 0.0f;
        return;
    }
    const HeadSel h = head_of(r, bs, T, io, nt, nf, nm, const_cast<float*>(ot), const_cast<float*>(of), const_cast<float*>(om), gt, gf, gm);
    float dp = 0.0f;
    if (lane < h.n && h.g) {
        const float y = h.out[h.orow * h.n + lane];
        dp = h.g[h.orow * h.n + lane] * y * (1.0f - y);
    }
    if (lane < 32) dpre[(size_t)r * 32 + lane] = dp;
    float acc[kHeadK];
#pragma unroll
    for (int j = 0; j < kHeadK; ++j) acc[j] = 0.0f;
    // grid.y splits the 64-float column slices of the row: lane (y, l) owns columns y 256 + l + 64 j, j < 4; the outputs go four at a time
    // with all sixteen weight loads in flight (one output per iteration was 27 serial round trips: 57 us for 36 rows)
    for (int o0 = 0; o0 < h.n; o0 += 4) {
        float wv[4][kHeadK];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < kHeadK; ++j) wv[q][j] = (o0 + q < h.n && lane + 64 * j < D) ? h.W[(size_t)(o0 + q) * D + lane + 64 * j] : 0.0f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float d = __shfl(dp, o0 + q < h.n ? o0 + q : 0);
#pragma unroll
            for (int j = 0; j < kHeadK; ++j) acc[j] = fmaf(o0 + q < h.n ? d : 0.0f, wv[q][j], acc[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < kHeadK; ++j)
        if (lane + 64 * j < D) dz[h.tok * D + lane + 64 * j] = acc[j];
}
// dW[o][k] = sum_rows dpre[row][o] z[token(row)][k], db[o] = sum_rows dpre[row][o]; grid (n_t + n_f + n_m), rows folded in order
__global__ __launch_bounds__(256) void k_ctrl_heads_bwd_dw(const float* __restrict__ z, const float* __restrict__ dpre, mst_ctrl_io_grads g, int nt,
                                                           int nf, int nm, int have_f, int have_m, int bs, int T, int D) {
    const int S = T + 4;
    int o = blockIdx.x, head = 0;
    if (o >= nt) { o -= nt; head = 1; }
    if (head == 1 && o >= nf) { o -= nf; head = 2; }
    if ((head == 1 && !have_f) || (head == 2 && !have_m)) return;  // unused head: its gradients stay None on the Python side
    float* dW = head == 0 ? g.track_w : (head == 1 ? g.fx_w : g.master_w);
    float* dB = head == 0 ? g.track_b : (head == 1 ? g.fx_b : g.master_b);
    const int r0 = head == 0 ? 0 : (head == 1 ? bs * T : bs * T + bs), nr = head == 0 ? bs * T : bs;
    for (int k = threadIdx.x; k < D; k += 256) {
        float acc = 0.0f;
        for (int q0 = 0; q0 < nr; q0 += 8) {  // eight rows' operands in flight, folded in row order
            float dv[8], zv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int q = q0 + u;
                const size_t tok = head == 0 ? (size_t)(q / T) * S + (q % T) : (size_t)q * S + T + 1 + head;
                dv[u] = q < nr ? dpre[(size_t)(r0 + q) * 32 + o] : 0.0f;
                zv[u] = q < nr ? z[tok * D + k] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = fmaf(dv[u], zv[u], acc);
        }
        dW[(size_t)o * D + k] = acc;
    }
    if (threadIdx.x == 0) {
        float acc = 0.0f;
        for (int q = 0; q < nr; ++q) acc += dpre[(size_t)(r0 + q) * 32 + o];
        dB[o] = acc;
    }
}
// gradients of the four type embeddings: sums of the token cotangents over the batch (and the tracks)
__global__ __launch_bounds__(256) void k_ctrl_tokens_bwd(const float* __restrict__ gtok, mst_ctrl_io_grads g, int bs, int T, int D) {
    const int k = blockIdx.x * 256 + threadIdx.x, S = T + 4;
    if (k >= D) return;
    float a = 0.0f, m0 = 0.0f, m1 = 0.0f, f = 0.0f, ms = 0.0f;
    for (int b = 0; b < bs; ++b) {
        const float* row = gtok + (size_t)b * S * D + k;
        for (int t0 = 0; t0 < T; t0 += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = t0 + u < T ? row[(size_t)(t0 + u) * D] : 0.0f;
#pragma unroll
            for (int u = 0; u < 8; ++u) a += v[u];
        }
        m0 += row[(size_t)T * D];
        m1 += row[(size_t)(T + 1) * D];
        f += row[(size_t)(T + 2) * D];
        ms += row[(size_t)(T + 3) * D];
    }
    g.track_embedding[k] = a;
    g.mix_embedding[k] = m0;
    g.mix_embedding[D + k] = m1;
    g.fx_bus_embedding[k] = f;
    g.master_bus_embedding[k] = ms;
}

}  // namespace ctrl
}  // namespace mst

static bool io_ok(const mst_ctrl_desc* d, int T, int nt, int nf, int nm) {
    return d && d->bs > 0 && T > 0 && d->seq == T + 4 && d->d_model > 0 && d->d_model <= 64 * kHeadK && nt > 0 && nt <= 32 && nf > 0 && nf <= 32 && nm > 0 &&
           nm <= 32;
}
extern "C" int mst_ctrl_tokens_forward(const mst_ctrl_desc* d, int32_t n_tracks, const float* track_embeds, const float* mix_embeds,
                                       const uint8_t* track_padding_mask, const mst_ctrl_io* io, float* tokens, uint8_t* key_padding_mask_out,
                                       void* stream) {
    if (!io_ok(d, n_tracks, 1, 1, 1) || !track_embeds || !mix_embeds || !io || !tokens) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_ctrl_tokens, dim3(d->seq, d->bs), dim3(128), 0, (hipStream_t)stream, track_embeds, mix_embeds, track_padding_mask, *io, tokens,
                       key_padding_mask_out, n_tracks, d->d_model);
    return (int)hipGetLastError();
}
extern "C" int mst_ctrl_heads_forward(const mst_ctrl_desc* d, int32_t n_tracks, const float* z, const mst_ctrl_io* io, int32_t n_t, int32_t n_f,
                                      int32_t n_m, float* out_t, float* out_f, float* out_m, void* stream) {
    if (!io_ok(d, n_tracks, n_t, n_f, n_m) || !z || !io || !out_t || !out_f || !out_m) return hipErrorInvalidValue;
    const int nmax = n_t > n_f ? (n_t > n_m ? n_t : n_m) : (n_f > n_m ? n_f : n_m);
    hipLaunchKernelGGL(k_ctrl_heads_fwd, dim3(d->bs * (n_tracks + 2), nmax), dim3(64), 0, (hipStream_t)stream, z, *io, n_t, n_f, n_m, out_t, out_f, out_m,
                       d->bs, n_tracks, d->d_model);
    return (int)hipGetLastError();
}
extern "C" size_t mst_ctrl_heads_scratch_bytes(const mst_ctrl_desc* d, int32_t n_tracks) {
    return (d && n_tracks > 0) ? (size_t)d->bs * (n_tracks + 2) * 32 * sizeof(float) : 0;
}
extern "C" int mst_ctrl_heads_backward(const mst_ctrl_desc* d, int32_t n_tracks, const float* z, const mst_ctrl_io* io, int32_t n_t, int32_t n_f,
                                       int32_t n_m, const float* out_t, const float* out_f, const float* out_m, const float* g_t, const float* g_f,
                                       const float* g_m, const mst_ctrl_io_grads* grads, float* grad_z, void* scratch, void* stream) {
    if (!io_ok(d, n_tracks, n_t, n_f, n_m) || !z || !io || !out_t || !out_f || !out_m || !grads || !grad_z || !scratch) return hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_ctrl_heads_bwd_dz, dim3(d->bs * (n_tracks + 4)), dim3(64), 0, st, *io, n_t, n_f, n_m, out_t, out_f, out_m, g_t, g_f, g_m, grad_z,
                       (float*)scratch, d->bs, n_tracks, d->d_model);
    hipLaunchKernelGGL(k_ctrl_heads_bwd_dw, dim3(n_t + n_f + n_m), dim3(256), 0, st, z, (const float*)scratch, *grads, n_t, n_f, n_m, g_f ? 1 : 0,
                       g_m ? 1 : 0, d->bs, n_tracks, d->d_model);
    return (int)hipGetLastError();
}
extern "C" int mst_ctrl_tokens_backward(const mst_ctrl_desc* d, int32_t n_tracks, const float* grad_tokens, const mst_ctrl_io_grads* grads, void* stream) {
    if (!io_ok(d, n_tracks, 1, 1, 1) || !grad_tokens || !grads) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_ctrl_tokens_bwd, dim3((d->d_model + 255) / 256), dim3(256), 0, (hipStream_t)stream, grad_tokens, *grads, d->bs, n_tracks, d->d_model);
    return (int)hipGetLastError();
}
