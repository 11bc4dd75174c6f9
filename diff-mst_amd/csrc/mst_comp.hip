// mst_comp.hip - feed-forward compressor, pan and bus-sum kernels (forward and backward).
//
// Replaces dasp-pytorch's `compressor` / `stereo_panner` + `tracks.sum(dim=2)` (reference call
// sites mst/modules.py:246, 263, 272, 300; algorithm SURVEY A.2/A.5):
//   side = sum_ch x ; x_db = 20 log10(max(|side|,1e-8)) ; soft-knee static curve -> g_c ;
//   g_s[n] = (1-a) g_c[n] + a g_s[n-1] ; y[n] = x[n-L] * 10^((g_s[n]+makeup)/20) .
// Every lane owns kCompChunk = 8 consecutive samples (two 16-byte accesses per stream); the
// one-pole smoother is a first-order linear recurrence handled as zs -> k_scan1 -> run, exactly
// like the EQ states.  The track "apply" kernel loops over the tracks of one mix so the stereo
// bus is accumulated in registers and written once (no (bs,2,T,N) intermediate unless asked for).
#include "mst_kernels.h"
#include "mst_compdev.h"

namespace mst {

constexpr int CC = kCompChunk;

// developer timeline (-DMST_CBR_STAMPS, tools/cbr_timeline.py): wave 0 of every track workgroup of k_comp_bwd_run stamps the 100 MHz
// wall clock at its phase boundaries; mst_debug_read_cbr_stamps copies the table out
#ifdef MST_CBR_STAMPS
constexpr int kStampSlots = 12, kStampWGs = 16384;
__device__ unsigned long long g_cbr_stamps[kStampWGs * kStampSlots];
#define CBR_STAMP(k)                                                                                         \
    do {                                                                                                     \
        if (!MASTER && threadIdx.x == 0) {                                                                   \
            const int wg_ = blockIdx.x + gridDim.x * blockIdx.y;                                             \
            if (wg_ < kStampWGs) g_cbr_stamps[wg_ * kStampSlots + (k)] = wall_clock64();                     \
        }                                                                                                    \
    } while (0)
#define CBM_STAMP(k)                                                                                         \
    do {                                                                                                     \
        if (threadIdx.x == 0) {                                                                              \
            const int wg_ = blockIdx.x + gridDim.x * blockIdx.y;                                             \
            if (wg_ < kStampWGs) g_cbr_stamps[wg_ * kStampSlots + (k)] = wall_clock64();                     \
        }                                                                                                    \
    } while (0)
#else
#define CBR_STAMP(k) do {} while (0)
#define CBM_STAMP(k) do {} while (0)
#endif

// the value is materialised HERE: without it LLVM sinks a whole unrolled loop below the next spin-wait (into the block that uses its
// results), which serialises the arithmetic behind the wait and keeps every operand of the loop alive across it
#if defined(__clang__)
__device__ __forceinline__ void pin(float& x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ void pin2(f2& x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ void pin_int(int& x) { asm volatile("" : "+v"(x)); }
#else  // host-only g++ build of these sources (the repository's CPU test harness): no code motion to guard against
inline void pin(float&) {}
inline void pin2(f2&) {}
inline void pin_int(int&) {}
#endif
__device__ __forceinline__ void ld8(const float* row, int64_t i, int64_t n, float* v) {
    const float4 a = load4(row, i, n), b = load4(row, i + 4, n);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void ld8s(const float* row, int64_t i, int64_t n, float* v) {
    const float4 a = load4_shift(row, i, n), b = load4_shift(row, i + 4, n);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void st8(float* row, int64_t i, int64_t n, const float* v);
// Unguarded variants for interior lanes of aligned rows (the overwhelmingly common case): no bounds or
// alignment tests, two plain 16-byte accesses.
__device__ __forceinline__ void ld8f(const float* row, int64_t i, float* v) {
#ifdef MST_NOMEM  // timing diagnostics only (wrong results): the interior blocks' operands are synthesised from the index - no global loads
    const float base = 0.001f * (float)((int)i & 2047) + 0.01f + 1e-6f * (float)((int)(uintptr_t)row & 0xffff);
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = base + 0.0007f * (float)q;
    return;
#endif
    const float4 a = *reinterpret_cast<const float4*>(row + i), b = *reinterpret_cast<const float4*>(row + i + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void st8f(float* row, int64_t i, const float* v) {
    *reinterpret_cast<float4*>(row + i) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(row + i + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
// FAST selects them at compile time inside the kernels: LD8(fast, ...) etc.
template <bool FAST> __device__ __forceinline__ void LD8(const float* row, int64_t i, int64_t n, float* v) {
    if (FAST) ld8f(row, i, v); else ld8(row, i, n, v);
}
template <bool FAST> __device__ __forceinline__ void LD8S(const float* row, int64_t i, int64_t n, float* v) {
    if (FAST) ld8f(row, i, v); else ld8s(row, i, n, v);
}
template <bool FAST> __device__ __forceinline__ void ST8(float* row, int64_t i, int64_t n, const float* v) {
    if (FAST) st8f(row, i, v); else st8(row, i, n, v);
}
__device__ __forceinline__ void st8(float* row, int64_t i, int64_t n, const float* v) {
    store4(row, i, n, make_float4(v[0], v[1], v[2], v[3]));
    store4(row, i + 4, n, make_float4(v[4], v[5], v[6], v[7]));
}

// sum of one value per wave over the workgroup's waves (fixed order), through four LDS floats at `slot`; trailing barrier so that the
// slots may be reused at once.  One-wave workgroups (MST_COMP_WG = 64): the identity, no LDS, no barrier.
__device__ __forceinline__ float waves_sum(float v, float* slot, int tid) {
    if (kCompWaves == 1) return v;
    if ((tid & 63) == 0) slot[tid >> 6] = v;
    lds_barrier();
    float s = 0.0f;
    if constexpr (kCompWaves == 4) s = (slot[0] + slot[1]) + (slot[2] + slot[3]);
    else
        for (int w = 0; w < kCompWaves; ++w) s += slot[w];
    lds_barrier();
    return s;
}

// ---- first-order carry machinery -------------------------------------------------------------------
// Per lane chunk the smoother is  s_out = a s_in + z  (a = alpha^8, z = zero-state end value).
// block_enter<REV>() returns the state ENTERING this lane's chunk given the state S entering the
// workgroup's 2048-sample block: wave-level Hillis-Steele with shuffles, wave aggregates through LDS.
// Scan order is lane-ascending (forward recursion) or lane-descending (REV, the adjoint recursion).
template <bool REV>
__device__ __forceinline__ float block_enter(float z, float a, float log2a, float S, float* lds, int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    const int rl = REV ? 63 - lane : lane, rw = REV ? kCompWaves - 1 - wave : wave;
    float v = z, p = a;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const float o = REV ? __shfl_down(v, d) : __shfl_up(v, d);
        if (rl >= d) v = fmaf(p, o, v);
        p *= p;
    }
    float sw = S;  // state entering this wave
    if (kCompWaves > 1) {
        if (rl == 63) lds[rw] = v;  // zero-entry aggregate of this wave
        lds_barrier();
        for (int w = 0; w < rw; ++w) sw = fmaf(p, sw, lds[w]);  // p == a^64
    }
    float ex = REV ? __shfl_down(v, 1) : __shfl_up(v, 1);
    if (rl == 0) ex = 0.0f;
    if (kCompWaves > 1) lds_barrier();  // lds may be reused by the caller's next call
    return fmaf(__builtin_amdgcn_exp2f((float)rl * log2a), sw, ex);
}
// The same scan with the entering state S left open: the state entering this lane's chunk is Q0 + W S.  Everything here is known
// before S is (for the in-launch exchange: before the other workgroups' aggregates have arrived).
template <bool REV>
__device__ __forceinline__ void block_enter_split(float z, float a, float log2a, float* lds, int tid, float& Q0, float& W) {
    const int lane = tid & 63, wave = tid >> 6;
    const int rl = REV ? 63 - lane : lane, rw = REV ? kCompWaves - 1 - wave : wave;
    float v = z, p = a;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const float o = REV ? __shfl_down(v, d) : __shfl_up(v, d);
        if (rl >= d) v = fmaf(p, o, v);
        p *= p;
    }
    float sw = 0.0f;  // zero-entry state entering this wave
    if (kCompWaves > 1) {
        if (rl == 63) lds[rw] = v;
        lds_barrier();
        for (int w = 0; w < rw; ++w) sw = fmaf(p, sw, lds[w]);  // p == a^64
    }
    float ex = REV ? __shfl_down(v, 1) : __shfl_up(v, 1);
    if (rl == 0) ex = 0.0f;
    if (kCompWaves > 1) lds_barrier();
    Q0 = fmaf(__builtin_amdgcn_exp2f((float)rl * log2a), sw, ex);
    W = __builtin_amdgcn_exp2f((float)(rl + 64 * rw) * log2a);
}
// state entering block `blk` from the aggregates of the blocks before it (after it when REV):
//   S = sum_j A^(dist-1) agg[j],  A = a^256, computed in a fixed order by the whole workgroup
template <bool REV>
__device__ __forceinline__ float block_carry(const float* __restrict__ agg, int blk, int nblk, float log2a, float* lds, int tid) {
    float acc = 0.0f;
    if (REV) {
        for (int j = blk + 1 + tid; j < nblk; j += kWG) acc += __builtin_amdgcn_exp2f((float)(j - blk - 1) * (float)kWG * log2a) * agg[j];
    } else {
        for (int j = blk - 1 - tid; j >= 0; j -= kWG) acc += __builtin_amdgcn_exp2f((float)(blk - 1 - j) * (float)kWG * log2a) * agg[j];
    }
    acc = wave_sum(acc);
    return waves_sum(acc, lds + 4, tid);
}

// the same from granules that the other workgroups of THIS launch publish (mst_common.h: gran_publish / gran_wait).  The first poll
// is split off (carry_peek) so that a kernel can request it early, do work that does not need the carry, and only then look at it.
template <bool REV>
__device__ __forceinline__ int carry_first(int blk, int tid) { return REV ? blk + 1 + tid : blk - 1 - tid; }
template <bool REV>
__device__ __forceinline__ gran_t carry_peek(const gran_t* __restrict__ agg, int64_t near_off, int blk, int nblk, int tid) {
    const int j = carry_first<REV>(blk, tid);
    return (j >= 0 && j < nblk) ? gran_load(agg + j + (MST_GRAN_NEAR ? near_off : 0)) : 0ull;
}
template <bool REV>
__device__ __forceinline__ float block_carry_g(const gran_t* __restrict__ agg, int64_t near_off, gran_t peek, int blk, int nblk, float log2a, float* lds, int tid,
                                               int32_t* status = nullptr) {
    float acc = 0.0f;
    bool first = true;
    if (REV) {
        for (int j = blk + 1 + tid; j < nblk; j += kWG, first = false) {
            const float v = (first && (peek >> 32) == 1) ? __int_as_float((int)(unsigned)peek) : gran_wait(agg + j, near_off, status);
            acc += __builtin_amdgcn_exp2f((float)(j - blk - 1) * (float)kWG * log2a) * v;
        }
    } else {
        for (int j = blk - 1 - tid; j >= 0; j -= kWG, first = false) {
            const float v = (first && (peek >> 32) == 1) ? __int_as_float((int)(unsigned)peek) : gran_wait(agg + j, near_off, status);
            acc += __builtin_amdgcn_exp2f((float)(blk - 1 - j) * (float)kWG * log2a) * v;
        }
    }
    acc = wave_sum(acc);
    return waves_sum(acc, lds + 4, tid);
}

// zero-entry aggregate of the workgroup's block alone (what the zero-state passes publish): a weighted SUM, not a scan -
//   forward: sum_t a^(255 - t) z_t,   REV: sum_t a^t z_t   (t = lane chunk index in the block, a = alpha^8)
// one exp2, six DPP additions and one barrier instead of block_enter's seven ds_bpermute and two barriers.
template <bool REV>
__device__ __forceinline__ float block_aggregate(float z, float log2a, float* lds, int tid) {
    const float w = __builtin_amdgcn_exp2f((float)(REV ? tid : kWG - 1 - tid) * log2a);
    const float v = wave_sum(w * z);
    return waves_sum(v, lds, tid);
}

// ---- forward: zero-state end value of the smoother per 2048-sample block ---------------------------
// u: [(row*NCH+ch)][stride]; zs: [row][nc_pad]
template <int NCH, bool FAST>
__device__ __forceinline__ void comp_zs_body(const float* __restrict__ u, int64_t stride, const float* __restrict__ rc,
                                             float* __restrict__ zs, int nc_pad, int64_t n) {
    const int row = blockIdx.y, chunk = blockIdx.x * kWG + threadIdx.x;
    const int64_t i0 = (int64_t)chunk * CC;
    const CompK k = load_comp(rc + (int64_t)row * RC_STRIDE);
    float side[CC];
    LD8<FAST>(u + (int64_t)(row * NCH) * stride, i0, n, side);
    if (NCH == 2) {
        float o[CC];
        LD8<FAST>(u + (int64_t)(row * NCH + 1) * stride, i0, n, o);
#pragma unroll
        for (int i = 0; i < CC; ++i) side[i] += o[i];
    }
    __shared__ float lds[8];
    float acc = 0.0f;
#pragma unroll
    for (int i = 0; i < CC; ++i) {
        float d;
        const float gc = gain_computer(side[i], k, d);
        // samples past the end contribute nothing (their state is never consumed)
        acc = fmaf(k.alpha, acc, k.oma * gc);
    }
    const float agg = block_aggregate<false>(acc, rc[(int64_t)row * RC_STRIDE + RC_LOG2A_C], lds, threadIdx.x);
    if (threadIdx.x == 0) zs[(int64_t)row * gridDim.x + blockIdx.x] = agg;
}
template <int NCH>
__global__ __launch_bounds__(kWG) void k_comp_zs(const float* __restrict__ u, int64_t stride, const float* __restrict__ rc,
                                                 float* __restrict__ zs, int nc_pad, int64_t n) {
    // whole-workgroup decision: every sample this block touches is in range (rows are 16-byte aligned)
    if ((int64_t)(blockIdx.x + 1) * kWG * CC <= n) comp_zs_body<NCH, true>(u, stride, rc, zs, nc_pad, n);
    else comp_zs_body<NCH, false>(u, stride, rc, zs, nc_pad, n);
}

// ---- forward: tracks.  grid (nblk, bs).  Accumulates the stereo bus over the T tracks of mix b.
// The loop over the tracks is software-pipelined: track t + 1's EQ output (at both offsets) is requested before track t's block scans, so
// its memory round trip hides behind four barriers and the static-curve arithmetic instead of opening every iteration (round-3 counters:
// 19 % VALU issue, 66 % of the wave cycles waiting).  FX (fx send bus) is a template parameter: its two accumulators cost 16 registers
// that the prefetch needs (123 of 128 before).
template <bool FAST, bool FX>
__device__ __forceinline__ void apply_tracks_body(const TrackApplyArgs& a, int b, int blk) {
    __shared__ float lds[8];
    const int chunk = blk * kWG + threadIdx.x;
    const int64_t i0 = (int64_t)chunk * CC;
    float accL[CC], accR[CC], fxL[FX ? CC : 1], fxR[FX ? CC : 1];
#pragma unroll
    for (int i = 0; i < CC; ++i) accL[i] = accR[i] = 0.0f;
    if (FX) {
#pragma unroll
        for (int i = 0; i < CC; ++i) fxL[i] = fxR[i] = 0.0f;
    }
    float xn[CC], xdn[CC];  // the NEXT track's samples (comp_on: at i0 and at i0 - lookahead)
    {
        const float* u0 = a.u + (int64_t)(b * a.T) * a.stride;
        LD8<FAST>(u0, i0, a.n, xn);
        if (a.comp_on) LD8S<FAST>(u0, i0 - a.lookahead, a.n, xdn);
    }
    for (int t = 0; t < a.T; ++t) {
        const int row = b * a.T + t;
        const float* rc = a.rc + (int64_t)row * RC_STRIDE;
        const float pl = rc[RC_PANL], pr = rc[RC_PANR];
        const float sl = FX ? pl * rc[RC_SEND] : 0.0f, sr = FX ? pr * rc[RC_SEND] : 0.0f;  // fx send bus: sum_t send_t * panned track
        float y[CC], x[CC], xd[CC];
#pragma unroll
        for (int i = 0; i < CC; ++i) {
            x[i] = xn[i];
            xd[i] = xdn[i];
        }
        if (t + 1 < a.T) {
            const float* un = a.u + (int64_t)(row + 1) * a.stride;
            LD8<FAST>(un, i0, a.n, xn);
            if (a.comp_on) LD8S<FAST>(un, i0 - a.lookahead, a.n, xdn);
        }
        if (a.comp_on) {
            const CompK k = load_comp(rc);
            float g[CC];
            float z = 0.0f;
#pragma unroll
            for (int i = 0; i < CC; ++i) {
                float d;
                g[i] = k.oma * gain_computer(x[i], k, d);
                z = fmaf(k.alpha, z, g[i]);
            }
            const float ac = rc[RC_ALPHA_C], l2a = rc[RC_LOG2A_C];
            const float S = block_carry<false>(a.s0 + (int64_t)row * gridDim.x, blk, gridDim.x, l2a, lds, threadIdx.x);
            float s = block_enter<false>(z, ac, l2a, S, lds, threadIdx.x);
#pragma unroll
            for (int i = 0; i < CC; ++i) {
                s = fmaf(k.alpha, s, g[i]);
                g[i] = s;
                y[i] = xd[i] * lin_gain(s, k);
            }
            if (a.gs) ST8<FAST>(a.gs + (int64_t)row * a.stride, i0, a.n, g);
        } else {
#pragma unroll
            for (int i = 0; i < CC; ++i) y[i] = x[i];
        }
#pragma unroll
        for (int i = 0; i < CC; ++i) {
            accL[i] = fmaf(pl, y[i], accL[i]);
            accR[i] = fmaf(pr, y[i], accR[i]);
        }
        if (FX) {
#pragma unroll
            for (int i = 0; i < CC; ++i) {
                fxL[i] = fmaf(sl, y[i], fxL[i]);
                fxR[i] = fmaf(sr, y[i], fxR[i]);
            }
        }
        if (a.mixed) {
            float ml[CC], mr[CC];
#pragma unroll
            for (int i = 0; i < CC; ++i) {
                ml[i] = pl * y[i];
                mr[i] = pr * y[i];
            }
            ST8<FAST>(a.mixed + (((int64_t)b * 2 + 0) * a.T + t) * a.n, i0, a.n, ml);
            ST8<FAST>(a.mixed + (((int64_t)b * 2 + 1) * a.T + t) * a.n, i0, a.n, mr);
        }
    }
    ST8<FAST>(a.bus + ((int64_t)b * 2 + 0) * a.bus_stride, i0, a.n, accL);
    ST8<FAST>(a.bus + ((int64_t)b * 2 + 1) * a.bus_stride, i0, a.n, accR);
    if (FX) {  // (bs, 2, stride) rows of the workspace: always 16-byte aligned
        ST8<FAST>(a.fx + ((int64_t)b * 2 + 0) * a.stride, i0, a.n, fxL);
        ST8<FAST>(a.fx + ((int64_t)b * 2 + 1) * a.stride, i0, a.n, fxR);
    }
}
__device__ __forceinline__ bool block_interior(int64_t n, int lookahead, int aligned, int blk = -1) {
    const int64_t lo = (int64_t)(blk < 0 ? (int)blockIdx.x : blk) * kWG * CC, hi = lo + (int64_t)kWG * CC;
    return aligned && lo - lookahead >= 0 && hi + lookahead <= n;
}
template <bool FX>
__global__ __launch_bounds__(kWG) void k_apply_tracks(TrackApplyArgs a) {
    // consecutive blocks of one mix on one XCD (mst_common.h: row_block_xcd): the look-ahead read of block b (samples 2048 earlier) is what
    // block b - 1 has just streamed through the same L2 - on the plain walk it was a second trip over the fabric for the whole of u
    int b, blk;
    row_block_xcd(b, blk);
    if (block_interior(a.n, a.lookahead, a.aligned, blk)) apply_tracks_body<true, FX>(a, b, blk);
    else apply_tracks_body<false, FX>(a, b, blk);
}

// ---- forward: master bus.  grid (nblk, bs).  out = delay(v) * G * gout  (stereo-linked)
template <bool FAST>
__device__ __forceinline__ void apply_master_body(const MasterApplyArgs& a, int b, int blk) {
    __shared__ float lds[8];
    const int chunk = blk * kWG + threadIdx.x;
    const int64_t i0 = (int64_t)chunk * CC;
    const float* rc = a.rc + (int64_t)b * RC_STRIDE;
    const float gout = rc[RC_PANL];
    const float* v0 = a.v + (int64_t)(b * 2) * a.stride;
    const float* v1 = v0 + a.stride;
    float yl[CC], yr[CC];
    if (a.comp_on) {
        const CompK k = load_comp(rc);
        float l[CC], r[CC], g[CC];
        LD8<FAST>(v0, i0, a.n, l);
        LD8<FAST>(v1, i0, a.n, r);
        float z = 0.0f;
#pragma unroll
        for (int i = 0; i < CC; ++i) {
            float d;
            g[i] = k.oma * gain_computer(l[i] + r[i], k, d);
            z = fmaf(k.alpha, z, g[i]);
        }
        const float ac = rc[RC_ALPHA_C], l2a = rc[RC_LOG2A_C];
        // no k_comp_zs launch (a.gran): this block's aggregate is published here, the earlier blocks' are picked up as they appear
        gran_t* gr = a.gran ? a.gran + (int64_t)b * gridDim.x : nullptr;
        if (gr) {
            const float agg = block_aggregate<false>(z, l2a, lds, threadIdx.x);
            if (threadIdx.x == 0) gran_publish(gr + blk, a.gran_near, agg);
        }
        LD8S<FAST>(v0, i0 - a.lookahead, a.n, yl);  // requested before the wait for the other blocks
        LD8S<FAST>(v1, i0 - a.lookahead, a.n, yr);
        const float S = gr ? block_carry_g<false>(gr, a.gran_near, carry_peek<false>(gr, a.gran_near, blk, gridDim.x, threadIdx.x), blk, gridDim.x, l2a, lds, threadIdx.x, a.status)
                           : block_carry<false>(a.s0 + (int64_t)b * gridDim.x, blk, gridDim.x, l2a, lds, threadIdx.x);
        float s = block_enter<false>(z, ac, l2a, S, lds, threadIdx.x);
#pragma unroll
        for (int i = 0; i < CC; ++i) {
            s = fmaf(k.alpha, s, g[i]);
            g[i] = s;
            const float G = lin_gain(s, k) * gout;
            yl[i] *= G;
            yr[i] *= G;
        }
        if (a.gs) ST8<FAST>(a.gs + (int64_t)b * a.stride, i0, a.n, g);
    } else {
        LD8<FAST>(v0, i0, a.n, yl);
        LD8<FAST>(v1, i0, a.n, yr);
#pragma unroll
        for (int i = 0; i < CC; ++i) {
            yl[i] *= gout;
            yr[i] *= gout;
        }
    }
    ST8<FAST>(a.out + ((int64_t)b * 2 + 0) * a.out_stride, i0, a.n, yl);
    ST8<FAST>(a.out + ((int64_t)b * 2 + 1) * a.out_stride, i0, a.n, yr);
}
__global__ __launch_bounds__(kWG) void k_apply_master(MasterApplyArgs a) {
    int b = blockIdx.y, blk = blockIdx.x;
    if (a.gran) row_block_xcd(b, blk);  // the blocks of one mix on one XCD, earlier blocks dispatched first (mst_common.h)
    if (block_interior(a.n, a.lookahead, a.aligned, blk)) apply_master_body<true>(a, b, blk);
    else apply_master_body<false>(a, b, blk);
}

// ---- backward ------------------------------------------------------------------------------------
// upstream cotangent of the compressor output y for 8 samples starting at i (may run past either end)
template <bool MASTER, bool FAST>
__device__ __forceinline__ void load_gy(const CompBwdArgs& a, int row, const float* rc, int64_t i, float* gl, float* gr) {
    // raw upstream cotangents per stereo channel: grad_mix (MASTER) or grad_bus (+ grad_mixed_tracks)
    const int b = MASTER ? row : row / a.T;
    float l[CC], r[CC];
    LD8S<FAST>(a.gup + ((int64_t)b * 2 + 0) * a.gup_stride, i, a.n, l);
    LD8S<FAST>(a.gup + ((int64_t)b * 2 + 1) * a.gup_stride, i, a.n, r);
    if (!MASTER && a.gmixed) {
        const int t = row % a.T;
        float ml[CC], mr[CC];
        LD8S<FAST>(a.gmixed + (((int64_t)b * 2 + 0) * a.T + t) * a.n, i, a.n, ml);
        LD8S<FAST>(a.gmixed + (((int64_t)b * 2 + 1) * a.T + t) * a.n, i, a.n, mr);
#pragma unroll
        for (int q = 0; q < CC; ++q) {
            l[q] += ml[q];
            r[q] += mr[q];
        }
    }
    if (!MASTER && a.gfx) {  // the panned track also feeds the fx send bus with gain `send`
        const float send = rc[RC_SEND];
        float fl[CC], fr[CC];
        LD8S<FAST>(a.gfx + ((int64_t)b * 2 + 0) * a.gfx_stride, i, a.n, fl);
        LD8S<FAST>(a.gfx + ((int64_t)b * 2 + 1) * a.gfx_stride, i, a.n, fr);
#pragma unroll
        for (int q = 0; q < CC; ++q) {
            l[q] = fmaf(send, fl[q], l[q]);
            r[q] = fmaf(send, fr[q], r[q]);
        }
    }
#pragma unroll
    for (int q = 0; q < CC; ++q) {
        gl[q] = l[q];
        gr[q] = r[q];
    }
}

// zero-state (from the right) end value of the adjoint smoother q[n] = dgs[n] + a q[n+1]
template <bool MASTER, bool FAST>
__device__ __forceinline__ void comp_bwd_zs_body(const CompBwdArgs& a) {
    constexpr int NCH = MASTER ? 2 : 1;
    __shared__ float lds[8];
    const int row = blockIdx.y, chunk = blockIdx.x * kWG + threadIdx.x;
    const int64_t i0 = (int64_t)chunk * CC;
    const float* rc = a.rc + (int64_t)row * RC_STRIDE;
    const CompK k = load_comp(rc);
    float gl[CC], gr[CC], xd0[CC], xd1[CC], g[CC];
    load_gy<MASTER, FAST>(a, row, rc, i0, gl, gr);
    LD8S<FAST>(a.u + (int64_t)(row * NCH) * a.stride, i0 - a.lookahead, a.n, xd0);
    if (MASTER) LD8S<FAST>(a.u + (int64_t)(row * NCH + 1) * a.stride, i0 - a.lookahead, a.n, xd1);
    LD8<FAST>(a.gs + (int64_t)row * a.stride, i0, a.n, g);
    const float pl = rc[RC_PANL], pr = rc[RC_PANR];  // master: both = output-fader gain
    float acc = 0.0f;
#pragma unroll
    for (int i = CC - 1; i >= 0; --i) {
        const float G = lin_gain(g[i], k);
        const float dot = MASTER ? pl * (gl[i] * xd0[i] + gr[i] * xd1[i]) : (pl * gl[i] + pr * gr[i]) * xd0[i];
        const float dgs = (FAST || i0 + i < a.n) ? dot * G * kLn10Over20 : 0.0f;
        acc = fmaf(k.alpha, acc, dgs);
    }
    const float agg = block_aggregate<true>(acc, rc[RC_LOG2A_C], lds, threadIdx.x);
    if (threadIdx.x == 0) a.zq[(int64_t)row * gridDim.x + blockIdx.x] = agg;
}
template <bool MASTER>
__global__ __launch_bounds__(kWG) void k_comp_bwd_zs(CompBwdArgs a) {
    if (block_interior(a.n, a.lookahead, a.aligned)) comp_bwd_zs_body<MASTER, true>(a);
    else comp_bwd_zs_body<MASTER, false>(a);
}

// ---- coefficient-gradient sums of the block, formed where du and u are in registers (MST_FUSE_COEFGRAD) ----------------------
// k_coefgrad's arithmetic (all-pole bank 1/A_k, b0/B_k on u from the saved chunk-entry states, five inner products with the
// cotangent per section) on this workgroup's 2048 samples = 32 chunks of 64: the block's du and u go to LDS in the chunk-per-row
// layout, lane (section s = tid >> 5, chunk c = tid & 31) walks its chunk with ONE section (k_coefgrad: three per lane), the 32
// chunk lanes of a section meet in a half-wave shuffle sum.  192 of the 256 lanes work; the cotangent du crosses HBM only if
// someone downstream wants it, u is not fetched a second time.
#ifndef MST_CG_ABLATE
#define MST_CG_ABLATE 0  // timing diagnostics only (wrong results): 1 = no state loads, 2 = four samples instead of 64
#endif
#ifndef MST_CG_PITCH
#define MST_CG_PITCH (kEqChunk + 4)
#endif
constexpr int kCgPitch = MST_CG_PITCH, kCgChunks = kWG * CC / kEqChunk, kCgTile = kCgChunks * kCgPitch;
static_assert((kCgChunks == 32 || kCgChunks == 8) && kSections * kCgChunks <= kWG, "one section x kCgChunks chunks per lane group");
// the four all-pole states entering this lane's (section, chunk) walk.  They depend on nothing the kernel computes: the compressor
// adjoint requests them with its first loads, so that their latency is not exposed behind the transposition barrier (MST_CBR_EARLY)
struct CgStates { float wa1, wa2, wb1, wb2; };
__device__ __forceinline__ int cg_chunk_of(int tid) {
#ifndef MST_CG_ROT
#define MST_CG_ROT 0  // A/B: chunk rotation of the upper half wave (lanes l and l + 32 then read different LDS rows)
#endif
    return kCgChunks == 32 ? (((tid & 31) + ((tid & 32) ? MST_CG_ROT : 0)) & 31) : tid % kCgChunks;
}
__device__ __forceinline__ CgStates coefgrad_states(const CompBwdArgs& a, int blk, int sig) {
    const int tid = threadIdx.x, s = tid / kCgChunks, c = cg_chunk_of(tid);
    CgStates w = {0.f, 0.f, 0.f, 0.f};
    if (s < kSections) {
        const int64_t base = ((int64_t)sig * 24 + 4 * s) * a.ap_nc_pad + (int64_t)blk * kCgChunks + c;
        w.wa1 = a.ap_s0[base];
        w.wa2 = a.ap_s0[base + a.ap_nc_pad];
        w.wb1 = a.ap_s0[base + 2 * (int64_t)a.ap_nc_pad];
        w.wb2 = a.ap_s0[base + 3 * (int64_t)a.ap_nc_pad];
    }
    return w;
}
template <bool FAST>
__device__ __forceinline__ void coefgrad_fused(const CompBwdArgs& a, int blk, int sig, const float* __restrict__ rc, int64_t i0, const float* xu,
                                               const float* du, float* __restrict__ cg_u, float* __restrict__ cg_g,
                                               bool have_pre = false, CgStates pre = {0.f, 0.f, 0.f, 0.f}) {
    const int tid = threadIdx.x;
    {
        const int c = tid >> 3, off = (tid & 7) * CC;  // 8 lanes x 8 samples = one chunk
        float g[CC];
#pragma unroll
        for (int i = 0; i < CC; ++i) g[i] = (FAST || i0 + i < a.n) ? du[i] : 0.0f;
        // chunk row image: sample 8 j + 4 h + e sits at float h 32 + 4 j + e (j = lane of the chunk, h = half of its eight samples), so that
        // the eight lanes of a ds_write_b128 group cover 32 consecutive banks (sample order 8 j + e put lanes j and j + 4 on the same banks:
        // a 2-way conflict on every store of the transposition - round-3 counters: 43 % of this kernel's LDS cycles)
        (void)off;
        float* pu = &cg_u[c * kCgPitch + 4 * (tid & 7)];
        float* pg = &cg_g[c * kCgPitch + 4 * (tid & 7)];
        *reinterpret_cast<float4*>(pu) = make_float4(xu[0], xu[1], xu[2], xu[3]);
        *reinterpret_cast<float4*>(pu + 32) = make_float4(xu[4], xu[5], xu[6], xu[7]);
        *reinterpret_cast<float4*>(pg) = make_float4(g[0], g[1], g[2], g[3]);
        *reinterpret_cast<float4*>(pg + 32) = make_float4(g[4], g[5], g[6], g[7]);
    }
    if (kCompWaves > 1) lds_barrier();
    else wave_lds_sync();
    const int s = tid / kCgChunks, c = cg_chunk_of(tid);
    if (s < kSections) {
        const float ka1 = rc[RC_SOS + 5 * s + 3], ka2 = rc[RC_SOS + 5 * s + 4];
        const float kc1 = rc[RC_AP + 3 * s], kc2 = rc[RC_AP + 3 * s + 1], kib0 = rc[RC_AP + 3 * s + 2];
        const CgStates w0 = have_pre ? pre : coefgrad_states(a, blk, sig);
        float wa1 = w0.wa1, wa2 = w0.wa2, wb1 = w0.wb1, wb2 = w0.wb2;
        if (MST_CG_ABLATE & 1) wa1 = wa2 = wb1 = wb2 = (float)c;
#ifdef MST_CG_F64  // diagnostic: the five inner products (and the recurrences feeding them) in double
        double db0 = 0., db1 = 0., db2 = 0., da1 = 0., da2 = 0.;
        double Wa1 = wa1, Wa2 = wa2, Wb1 = wb1, Wb2 = wb2;
        for (int i = 0; i < kEqChunk; ++i) {
            const int at = ((i >> 2) & 1) * 32 + (i >> 3) * 4 + (i & 3);
            const double x = cg_u[c * kCgPitch + at], gp = cg_g[c * kCgPitch + at];
            const double wa = -(double)ka2 * Wa2 - (double)ka1 * Wa1 + x, wb = -(double)kc2 * Wb2 - (double)kc1 * Wb1 + x;
            db0 += gp * wb; db1 += gp * Wb1; db2 += gp * Wb2; da1 -= gp * Wa1; da2 -= gp * Wa2;
            Wa2 = Wa1; Wa1 = wa; Wb2 = Wb1; Wb1 = wb;
        }
        float acc[5] = {(float)(db0 * kib0), (float)(db1 * kib0), (float)(db2 * kib0), (float)da1, (float)da2};
#else
        // The two recurrences (1 / B_s and 1 / A_s driven by the same input) and the lag-1 / lag-2 inner products are the same arithmetic
        // on two values: held as float pairs they take one packed instruction each (v_pk_fma_f32) - five instructions per sample
        // instead of nine.  Lane x of a pair = the 1 / B_s side (numerator sums), lane y = the 1 / A_s side (its sums enter negated).
#ifndef MST_CG_PACKED
#define MST_CG_PACKED 1
#endif
        float db0 = 0.f, db1, db2, da1, da2;
        const float* mu = &cg_u[c * kCgPitch];
        const float* mg = &cg_g[c * kCgPitch];
#if MST_CG_PACKED
        f2 w1 = {wb1, wa1}, w2 = {wb2, wa2}, acc1 = {0.f, 0.f}, acc2 = {0.f, 0.f};
        const f2 nk1 = {-kc1, -ka1}, nk2 = {-kc2, -ka2};
#pragma unroll 2
        for (int i4 = 0; i4 < ((MST_CG_ABLATE & 2) ? 4 : kEqChunk); i4 += 4) {
            const int at = ((i4 >> 2) & 1) * 32 + (i4 >> 3) * 4;  // samples i4 .. i4 + 3 in the row image (see the stores above)
            const float4 xv = *reinterpret_cast<const float4*>(&mu[at]);
            const float4 gv = *reinterpret_cast<const float4*>(&mg[at]);
            const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, gs[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const f2 xx = {xs[t], xs[t]}, gg = {gs[t], -gs[t]};
                const f2 wn = f2_fma(nk2, w2, f2_fma(nk1, w1, xx));
                db0 = fmaf(gs[t], wn.x, db0);
                acc1 = f2_fma(gg, w1, acc1);
                acc2 = f2_fma(gg, w2, acc2);
                w2 = w1;
                w1 = wn;
            }
        }
        db1 = acc1.x; da1 = acc1.y; db2 = acc2.x; da2 = acc2.y;
#else
        db1 = db2 = da1 = da2 = 0.f;
#pragma unroll 2
        for (int i4 = 0; i4 < ((MST_CG_ABLATE & 2) ? 4 : kEqChunk); i4 += 4) {
            const int at = ((i4 >> 2) & 1) * 32 + (i4 >> 3) * 4;  // samples i4 .. i4 + 3 in the row image (see the stores above)
            const float4 xv = *reinterpret_cast<const float4*>(&mu[at]);
            const float4 gv = *reinterpret_cast<const float4*>(&mg[at]);
            const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, gs[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float x = xs[t], gp = gs[t], gm = -gs[t];
                const float wa = fmaf(-ka2, wa2, fmaf(-ka1, wa1, x));
                const float wb = fmaf(-kc2, wb2, fmaf(-kc1, wb1, x));
                db0 = fmaf(gp, wb, db0);
                db1 = fmaf(gp, wb1, db1);
                db2 = fmaf(gp, wb2, db2);
                da1 = fmaf(gm, wa1, da1);
                da2 = fmaf(gm, wa2, da2);
                wa2 = wa1;
                wa1 = wa;
                wb2 = wb1;
                wb1 = wb;
            }
        }
#endif
        float acc[5] = {db0 * kib0, db1 * kib0, db2 * kib0, da1, da2};
#endif
        // sum over the 32 chunk lanes of the section (one half wave), fixed order: four DPP row shifts leave every 16-lane row's total in
        // its last lane, row_bcast:15 adds row 0's into row 1 (row 2's into row 3) - lane 31 / 63 holds the half-wave sum.  (The five
        // ds_bpermute levels this replaces were 50 of the kernel's 64 LDS-crossbar round trips and what rocprof reported as 40 % "bank
        // conflict" cycles: the LDS images themselves are conflict-free, tools/lds_conflicts.py.)
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            float v = acc[i];
            if (kCgChunks == 32) {
                v = dpp_add<0x111, 0xf>(v);
                v = dpp_add<0x112, 0xf>(v);
                v = dpp_add<0x114, 0xf>(v);
                v = dpp_add<0x118, 0xf>(v);
                v = dpp_add<0x142, 0xa>(v);
            } else {  // eight chunk lanes per section: xor 1, xor 2 (quad permutes), then the other quad of the 8-lane group (row_half_mirror)
                v = dpp_add<0xb1, 0xf>(v);   // quad_perm [1,0,3,2]
                v = dpp_add<0x4e, 0xf>(v);   // quad_perm [2,3,0,1]
                v = dpp_add<0x141, 0xf>(v);  // row_half_mirror
            }
            acc[i] = v;
        }
        if ((tid % kCgChunks) == kCgChunks - 1) {
            float* o = a.ep + ((int64_t)sig * gridDim.x + blk) * EP_COUNT + 5 * s;
#pragma unroll
            for (int i = 0; i < 5; ++i) o[i] = acc[i];
        }
    }
    if (kCompWaves > 1) lds_barrier();  // red[] below and the next use of the tiles
    else wave_lds_sync();
}

// FXS: the fx send bus is on (tracks only) - its cotangent rows are read and the send-gain sum is formed
template <bool MASTER, bool FAST, bool FXS>
__device__ __forceinline__ void comp_bwd_run_body(const CompBwdArgs& a, int row, int blk, float* __restrict__ cg_u, float* __restrict__ cg_g) {
    constexpr int NCH = MASTER ? 2 : 1;
    __shared__ float red[kCompWaves > 1 ? kCompWaves : 1][CP_COUNT];  // red[0] doubles as the 8-float scan scratch
    const int tid = threadIdx.x, chunk = blk * kWG + tid;
    const int64_t i0 = (int64_t)chunk * CC;
    const float* rc = a.rc + (int64_t)row * RC_STRIDE;
    const float* u0 = a.u + (int64_t)(row * NCH) * a.stride;
    const float* u1 = u0 + a.stride;
    float p[CP_COUNT];
#pragma unroll
    for (int i = 0; i < CP_COUNT; ++i) p[i] = 0.0f;
    float gl[CC], gr[CC], du0[CC], du1[CC];
    float xu[CC], xu1[MASTER ? CC : 1];  // the compressor input of this lane's samples, kept for the fused coefficient-gradient pass
    CBR_STAMP(0);
    load_gy<MASTER, FAST>(a, row, rc, i0, gl, gr);
    CgStates cgs = {0.f, 0.f, 0.f, 0.f}, cgs1 = {0.f, 0.f, 0.f, 0.f};  // all-pole entry states of the coefficient walk, when requested early
    bool have_cgs = false;
    const float pl = rc[RC_PANL], pr = rc[RC_PANR];  // master: both = output-fader gain
    // cotangent of the fx send gain: sum_n (pl fL + pr fR)[n] y[n]  (fL, fR = cotangent of the send bus)
    float fsum[FXS ? CC : 1];
    if (FXS) {
        const int bb = row / a.T;
        float fl[CC], fr[CC];
        LD8S<FAST>(a.gfx + ((int64_t)bb * 2 + 0) * a.gfx_stride, i0, a.n, fl);
        LD8S<FAST>(a.gfx + ((int64_t)bb * 2 + 1) * a.gfx_stride, i0, a.n, fr);
#pragma unroll
        for (int i = 0; i < CC; ++i) fsum[i] = pl * fl[i] + pr * fr[i];
    }

    if (a.comp_on) {
        const CompK k = load_comp(rc);
        float x0[CC], x1[CC], xd0[CC], xd1[CC], g[CC];
        // what the adjoint smoother's zero-state value needs comes first: with a.gran the block's aggregate is published as early as
        // possible (no k_comp_bwd_zs launch) and the later blocks' aggregates are awaited only after every other load has been requested
        LD8S<FAST>(u0, i0 - a.lookahead, a.n, xd0);
        if (MASTER) LD8S<FAST>(u1, i0 - a.lookahead, a.n, xd1);
        LD8<FAST>(a.gs + (int64_t)row * a.stride, i0, a.n, g);
#ifndef MST_CBR_EARLY
#define MST_CBR_EARLY 0  // bit mask (1 look-ahead operands, 2 x / g[-1], 4 all-pole entry states): every other global load of the block is requested here too, in front of the zero-state loop: one memory latency
                         // per workgroup instead of two (the loads behind the aggregate's barrier could not start before it)
#endif
        float gF[CC], glF[CC], grF[CC];
        float g_prev0 = 0.0f;
        if (MST_CBR_EARLY & 1) {
            LD8S<FAST>(a.gs + (int64_t)row * a.stride, i0 + a.lookahead, a.n, gF);
            load_gy<MASTER, FAST>(a, row, rc, i0 + a.lookahead, glF, grF);
        }
        if (MST_CBR_EARLY & 2) {
            LD8<FAST>(u0, i0, a.n, x0);
            if (MASTER) LD8<FAST>(u1, i0, a.n, x1);
            g_prev0 = (i0 > 0 && i0 - 1 < a.n) ? a.gs[(int64_t)row * a.stride + i0 - 1] : 0.0f;
        }
        if ((MST_CBR_EARLY & 4) && a.ep) {
            cgs = coefgrad_states(a, blk, row * NCH);
            if (MASTER) cgs1 = coefgrad_states(a, blk, row * NCH + 1);
            have_cgs = true;
        }
#ifndef MST_CBR_SPLIT
#define MST_CBR_SPLIT 1  // 1: the adjoint smoother's state enters every q-dependent quantity as  zero-state part + Q x homogeneous part
                         // (q is linear in the state Q entering the lane's chunk), both parts formed BEFORE the other blocks' aggregates
                         // are awaited: behind the wait are one multiply-add per sample and per sum instead of the q recurrence, and
                         // the five per-sample coefficient arrays the recurrence needed are not kept (the kernel spilled 23 registers)
#endif
        float dgsv[CC];
        float zq = 0.0f;
#pragma unroll
        for (int i = CC - 1; i >= 0; --i) {
            const float Gi = lin_gain(g[i], k);
            const float dot = MASTER ? pl * (gl[i] * xd0[i] + gr[i] * xd1[i]) : (pl * gl[i] + pr * gr[i]) * xd0[i];
            dgsv[i] = (FAST || i0 + i < a.n) ? dot * Gi * kLn10Over20 : 0.0f;
            zq = fmaf(k.alpha, zq, dgsv[i]);
#if MST_CBR_SPLIT
            // the pan / make-up / send sums need exactly these operands: formed here, the upstream cotangents and the delayed input
            // are dead before the static-curve loop starts
            if (FAST || i0 + i < a.n) {
                p[CP_MAKEUP] += dgsv[i];
                if (MASTER) {
                    p[CP_PANL] = fmaf(gl[i] * xd0[i] + gr[i] * xd1[i], Gi, p[CP_PANL]);  // output-fader gain: sum(grad_mix * out_before_fader)
                } else {
                    const float yv = xd0[i] * Gi;
                    p[CP_PANL] = fmaf(gl[i], yv, p[CP_PANL]);
                    p[CP_PANR] = fmaf(gr[i], yv, p[CP_PANR]);
                    if (FXS) p[CP_SEND] = fmaf(fsum[i], yv, p[CP_SEND]);
                }
            }
#endif
        }
        const float ac = rc[RC_ALPHA_C], l2a = rc[RC_LOG2A_C];
        gran_t* gq = a.gran ? a.gran + (int64_t)row * gridDim.x : nullptr;
        CBR_STAMP(1);
        if (gq) {
            const float agg = block_aggregate<true>(zq, l2a, red[0], tid);
            if (tid == 0) gran_publish(gq + blk, a.gran_near, agg);
        }
        CBR_STAMP(2);
        // look-ahead branch: du[i] += gy[i+L] * G[i+L].  Folded to one (two: master) value per sample right after the loads -
        // three arrays less are alive across the block scan (the kernel ran at 152 registers = 3 waves per SIMD)
        float fwd0[CC], fwd1[MASTER ? CC : 1];
        {
            if (!(MST_CBR_EARLY & 1)) {
                LD8S<FAST>(a.gs + (int64_t)row * a.stride, i0 + a.lookahead, a.n, gF);
                load_gy<MASTER, FAST>(a, row, rc, i0 + a.lookahead, glF, grF);
            }
#pragma unroll
            for (int i = 0; i < CC; ++i) {
                const bool liveF = FAST || i0 + i + a.lookahead < a.n;
                const float GF = lin_gain(gF[i], k);
                fwd0[i] = liveF ? (MASTER ? pl * glF[i] : pl * glF[i] + pr * grF[i]) * GF : 0.0f;
                if (MASTER) fwd1[i] = liveF ? pr * grF[i] * GF : 0.0f;
            }
        }
        CBR_STAMP(3);
        if (!(MST_CBR_EARLY & 2)) {
            LD8<FAST>(u0, i0, a.n, x0);
            if (MASTER) LD8<FAST>(u1, i0, a.n, x1);
            g_prev0 = (i0 > 0 && i0 - 1 < a.n) ? a.gs[(int64_t)row * a.stride + i0 - 1] : 0.0f;
        }
#pragma unroll
        for (int i = 0; i < CC; ++i) xu[i] = x0[i];
        if (MASTER) {
#pragma unroll
            for (int i = 0; i < CC; ++i) xu1[i] = x1[i];
        }
#if MST_CBR_SPLIT
#ifndef MST_CBR_SCHED
#define MST_CBR_SCHED 3  // scheduling fences (1: between the phases, 2: between the samples of the static-curve loop): unfenced the
                         // compiler interleaves all eight samples and needs 172 registers
#endif
        if (MST_CBR_SCHED & 1) __builtin_amdgcn_sched_barrier(0);
        // state entering this lane's chunk = Q0 + W S (S = state entering the block, known after the wait)
        float Q0 = 0.0f, W = 0.0f;
        block_enter_split<true>(zq, ac, l2a, red[0], tid, Q0, W);
        const gran_t peek = gq ? carry_peek<true>(gq, a.gran_near, blk, gridDim.x, tid) : 0ull;
        CBR_STAMP(4);
        // q[i] = zs[i] + pw[i] Q with zs the zero-entry recurrence over this lane's samples and pw[i] = alpha^(CC - i):
        //   alpha += q A;  kappa += oma q Fv;  thr -= oma q Kp;  knee += oma q Kw;  du = oma q E + look-ahead branch
        // pairs (zero-state part, homogeneous part): one packed multiply-add per sum and sample
        f2 zp = {0.0f, 1.0f}, sA = {0.f, 0.f}, sF = {0.f, 0.f}, sP = {0.f, 0.f}, sW = {0.f, 0.f};
        const float kinvw = k.kappa * k.invw, kw2 = k.kappa * k.inv2w * k.invw, ke = k.oma * 8.685889638065035f;
        float db[CC];  // du0 / du1 hold the zero-state part until Q is known
#pragma unroll
        for (int i = CC - 1; i >= 0; --i) {
            const bool live = FAST || i0 + i < a.n;
            const float side = MASTER ? x0[i] + x1[i] : x0[i];
            // static curve and its derivatives (mst_compdev.h: curve_f): f, kappa df/dd, kappa df/dknee
            float tc;
#ifdef MST_CBR_ABLATE  // timing diagnostics only (wrong results): no static curve
            tc = side;
            const float fval = side;
#else
            const float fval = curve_f(curve_t(side, k), k, tc);
#endif
            const float gprev = (i > 0) ? g[i - 1] : g_prev0;
            const float kp = tc * kinvw;
            const float cKw = tc * (k.knee - tc) * kw2;
            // beyond the row's end the guarded loads return 0: side = 0 gives tc = 0 (f and both derivatives vanish) and the clamp below
            // kills the side chain; only g[i-1] - g_c has to be masked
            const float cA = live ? fmaf(-k.kappa, fval, gprev) : 0.0f;
            // side chain: d x_db / d side = (20/ln10) / side, clamp kills it below eps
#ifdef MST_CBR_ABLATE
            const float cE = kp;
#else
            const float cE = (fabsf(side) >= kCompEps) ? kp * ke * __builtin_amdgcn_rcpf(side) : 0.0f;
#endif
            zp.x = fmaf(k.alpha, zp.x, dgsv[i]);  // zs: zero-entry recurrence
            zp.y *= k.alpha;                      // pw = alpha^(CC - i)
            sA = f2_fma(zp, f2{cA, cA}, sA);
            sF = f2_fma(zp, f2{fval, fval}, sF);
            sP = f2_fma(zp, f2{kp, kp}, sP);
            sW = f2_fma(zp, f2{cKw, cKw}, sW);
            du0[i] = fmaf(zp.x, cE, fwd0[i]);
            if (MASTER) du1[i] = fmaf(zp.x, cE, fwd1[i]);
            db[i] = zp.y * cE;
#ifndef MST_CBR_PIN
#define MST_CBR_PIN 1  // 1: the static-curve loop is pinned in front of the granule wait (tools/cbr_timeline.py showed it sunk BEHIND the wait,
                       // into the block that uses its results: the wait then overlapped nothing)
#endif
            if (MST_CBR_PIN) {
                pin(du0[i]);
                if (MASTER) pin(du1[i]);
                pin(db[i]);
            }
            if (MST_CBR_SCHED & 2) __builtin_amdgcn_sched_barrier(0);
        }
        if (MST_CBR_PIN) {
            pin2(sA);
            pin2(sF);
            pin2(sP);
            pin2(sW);
        }
        CBR_STAMP(5);
        const float S = gq ? block_carry_g<true>(gq, a.gran_near, peek, blk, gridDim.x, l2a, red[0], tid, a.status)
                           : block_carry<true>(a.s0 + (int64_t)row * gridDim.x, blk, gridDim.x, l2a, red[0], tid);
        CBR_STAMP(6);
        const float Q = fmaf(W, S, Q0);
#pragma unroll
        for (int i = 0; i < CC; ++i) {
            du0[i] = fmaf(Q, db[i], du0[i]);
            if (MASTER) du1[i] = fmaf(Q, db[i], du1[i]);
        }
        p[CP_ALPHA] = fmaf(Q, sA.y, sA.x);
        p[CP_KAPPA] = k.oma * fmaf(Q, sF.y, sF.x);
        p[CP_THR] = -k.oma * fmaf(Q, sP.y, sP.x);
        p[CP_KNEE] = k.oma * fmaf(Q, sW.y, sW.x);
#else
        // first look at the later blocks' aggregates: requested here, examined after the arithmetic below, which does not need them
        const gran_t peek = gq ? carry_peek<true>(gq, a.gran_near, blk, gridDim.x, tid) : 0ull;
        // Everything that does not involve the adjoint smoother's state q (static curve, knee derivatives, the 1 / side of the side
        // chain, the pan / make-up sums) - five values per sample are kept for the short q-dependent loop behind the wait:
        //   dgc = oma q;  alpha += q A;  kappa += dgc Fv;  thr -= dgc Kp;  knee += dgc Kw;  du = dgc E + look-ahead branch
        float cA[CC], cFv[CC], cKp[CC], cKw[CC], cE[CC];
#pragma unroll
        for (int i = CC - 1; i >= 0; --i) {
            const bool live = FAST || i0 + i < a.n;
            const float G = lin_gain(g[i], k);  // recomputed (one exp2): eight registers less across the zero-state loop
            const float side = MASTER ? x0[i] + x1[i] : x0[i];
            float d;
            const float gc = gain_computer(side, k, d);
            const float gprev = (i > 0) ? g[i - 1] : g_prev0;
            float fval = 0.0f, fp = 0.0f, fw = 0.0f;
            if (d > k.hw) {
                fval = d;
                fp = 1.0f;
            } else if (d >= -k.hw) {
                const float t = d + k.hw;
                fval = t * t * k.inv2w;
                fp = t * k.invw;
                fw = t * (k.hw - d) * k.inv2w * k.invw;
            }
            const float kp = k.kappa * fp;
            cA[i] = live ? gprev - gc : 0.0f;
            cFv[i] = live ? fval : 0.0f;
            cKp[i] = live ? kp : 0.0f;
            cKw[i] = live ? k.kappa * fw : 0.0f;
            // side chain: d x_db / d side = (20/ln10) / side, clamp kills it below eps
            cE[i] = (fabsf(side) >= kCompEps) ? kp * 8.685889638065035f * __builtin_amdgcn_rcpf(side) : 0.0f;
            if (live) {
                p[CP_MAKEUP] += dgsv[i];
                if (MASTER) {
                    // cotangent of the output-fader gain: sum(grad_mix * out_before_fader)
                    p[CP_PANL] = fmaf(gl[i] * xd0[i] + gr[i] * xd1[i], G, p[CP_PANL]);
                } else {
                    const float yv = xd0[i] * G;
                    p[CP_PANL] = fmaf(gl[i], yv, p[CP_PANL]);
                    p[CP_PANR] = fmaf(gr[i], yv, p[CP_PANR]);
                    if (FXS) p[CP_SEND] = fmaf(fsum[i], yv, p[CP_SEND]);
                }
            }
        }
        const float S = gq ? block_carry_g<true>(gq, a.gran_near, peek, blk, gridDim.x, l2a, red[0], tid, a.status)
                           : block_carry<true>(a.s0 + (int64_t)row * gridDim.x, blk, gridDim.x, l2a, red[0], tid);
        float q = block_enter<true>(zq, ac, l2a, S, red[0], tid);
#pragma unroll
        for (int i = CC - 1; i >= 0; --i) {
            q = fmaf(k.alpha, q, dgsv[i]);
            const float dgc = k.oma * q;
            p[CP_ALPHA] = fmaf(q, cA[i], p[CP_ALPHA]);
            p[CP_KAPPA] = fmaf(dgc, cFv[i], p[CP_KAPPA]);
            p[CP_THR] = fmaf(-dgc, cKp[i], p[CP_THR]);
            p[CP_KNEE] = fmaf(dgc, cKw[i], p[CP_KNEE]);
            const float ds = dgc * cE[i];
            du0[i] = ds + fwd0[i];
            if (MASTER) du1[i] = ds + fwd1[i];
        }
#endif
    } else {
        float x0[CC], x1[CC];
        LD8<FAST>(u0, i0, a.n, x0);
        if (MASTER) LD8<FAST>(u1, i0, a.n, x1);
#pragma unroll
        for (int i = 0; i < CC; ++i) {
            xu[i] = x0[i];
            if (MASTER) xu1[i] = x1[i];
        }
#pragma unroll
        for (int i = 0; i < CC; ++i) {
            du0[i] = MASTER ? pl * gl[i] : pl * gl[i] + pr * gr[i];
            if (MASTER) du1[i] = pr * gr[i];
            if (i0 + i < a.n) {
                if (MASTER) p[CP_PANL] = fmaf(gl[i], x0[i], fmaf(gr[i], x1[i], p[CP_PANL]));
                else {
                    p[CP_PANL] = fmaf(gl[i], x0[i], p[CP_PANL]);
                    p[CP_PANR] = fmaf(gr[i], x0[i], p[CP_PANR]);
                    if (FXS) p[CP_SEND] = fmaf(fsum[i], x0[i], p[CP_SEND]);
                }
            }
        }
    }
    if (a.du) {
        ST8<FAST>(a.du + (int64_t)(row * NCH) * a.stride, i0, a.n, du0);
        if (MASTER) ST8<FAST>(a.du + (int64_t)(row * NCH + 1) * a.stride, i0, a.n, du1);
    }
    CBR_STAMP(7);
    if (a.ep) {  // signal rows: tracks = row, master = 2 row + channel
        coefgrad_fused<FAST>(a, blk, row * NCH, rc, i0, xu, du0, cg_u, cg_g, have_cgs, cgs);
        if (MASTER) coefgrad_fused<FAST>(a, blk, row * NCH + 1, rc, i0, xu1, du1, cg_u, cg_g, have_cgs, cgs1);
    }

    CBR_STAMP(8);
    const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
    for (int i = 0; i < CP_COUNT; ++i) {
        const float v = wave_sum(p[i]);
        if (lane == 0) red[wave][i] = v;
    }
    if (kCompWaves > 1) lds_barrier();
    else wave_lds_sync();
    if (tid < CP_COUNT) {
        float t = 0.0f;
        if constexpr (kCompWaves == 4) t = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
        else
            for (int w = 0; w < kCompWaves; ++w) t += red[w][tid];
        a.part[((int64_t)row * gridDim.x + blk) * CP_COUNT + tid] = t;
    }
    CBR_STAMP(9);
}
#ifndef MST_COMP_BWD_W
#define MST_COMP_BWD_W 1  // min waves per SIMD asked of the compressor backward (A/B switch)
#endif
template <bool MASTER, bool FXS = false>
#ifndef MST_COMP_BWD_WT
#define MST_COMP_BWD_WT 4
#endif
#ifndef MST_COMP_BWD_WM
#define MST_COMP_BWD_WM 4  // master rows (round 6): 137 registers uncapped = three waves per SIMD = 1.33 rounds of its 1024 workgroups at cfg #2;
                           // capped at 128 (8 spilled, also in the path with the coefficient-gradient walks that cfg #2 does not take) one round: 18.5 -> 17.9 us
#endif
__global__ __launch_bounds__(kWG, (!MASTER && !FXS) ? MST_COMP_BWD_WT : (MASTER ? MST_COMP_BWD_WM : MST_COMP_BWD_W)) void k_comp_bwd_run(CompBwdArgs a) {  // tracks without fx: 130 registers uncapped, two short of four waves per SIMD
    __shared__ __attribute__((aligned(16))) float cg_u[kCgTile], cg_g[kCgTile];  // one copy for both bodies
    // a.gran: the blocks a workgroup waits for (LATER in time: the adjoint smoother runs backwards) must have been dispatched before
    // it, so the grid walks the row from its end
    int row = blockIdx.y, blk = blockIdx.x;
    const int extra = MASTER ? 0 : a.cg2_rows, main_rows = (int)gridDim.y - extra;
    // Coefficient-gradient-only rows (one channel of a master bus each; its compressor adjoint ran in the master launch) are light
    // workgroups.  Behind the track rows they were a 21 us tail of half-empty rounds; they come FIRST in the grid (+13 us).  A/B switch
    // MST_CG2_INTERLEAVE: eight of them after every 8 x ratio track workgroups (a track workgroup's index keeps its residue mod 8 = its
    // XCD and the dispatch order of the track workgroups is kept) - measured WORSE than in front, 98.3 against 93.6 us.
#ifndef MST_CG2_INTERLEAVE
#define MST_CG2_INTERLEAVE 0
#endif
    int lin = -1;
    bool light = !MASTER && row < extra;
    if (!MASTER && MST_CG2_INTERLEAVE && extra > 0 && a.gran && main_rows % extra == 0 && ((int)gridDim.x * extra) % 8 == 0) {
        const int ratio = main_rows / extra, period = 8 * (ratio + 1);
        const int L = blockIdx.x + (int)gridDim.x * blockIdx.y, q = L / period, r = L % period;
        light = r >= 8 * ratio;
        if (light) {
            const int lj = q * 8 + (r - 8 * ratio);
            row = lj / (int)gridDim.x;
            blk = lj % (int)gridDim.x;
        } else {
            lin = q * 8 * ratio + r;
        }
    }
    if (light) {
        CBR_STAMP(10);
        const int j = row;
        const int64_t i0 = ((int64_t)blk * kWG + threadIdx.x) * CC;
        const bool fast = a.aligned && (int64_t)(blk + 1) * kWG * CC <= a.n;
        float xu[CC], du[CC];
        if (fast) {
            ld8f(a.cg2_u + (int64_t)j * a.stride, i0, xu);
            ld8f(a.cg2_du + (int64_t)j * a.stride, i0, du);
            coefgrad_fused<true>(a, blk, main_rows + j, a.cg2_rc + (int64_t)(j >> 1) * RC_STRIDE, i0, xu, du, cg_u, cg_g);
        } else {
            ld8(a.cg2_u + (int64_t)j * a.stride, i0, a.n, xu);
            ld8(a.cg2_du + (int64_t)j * a.stride, i0, a.n, du);
            coefgrad_fused<false>(a, blk, main_rows + j, a.cg2_rc + (int64_t)(j >> 1) * RC_STRIDE, i0, xu, du, cg_u, cg_g);
        }
        CBR_STAMP(11);
        return;
    }
    if (a.gran) {
        row_block_xcd(row, blk, MASTER ? 1 : a.T, main_rows, extra, lin);  // the blocks of one row - and the tracks of one mix - on one XCD (mst_common.h)
        blk = gridDim.x - 1 - blk;
    } else {
        row -= extra;
    }
    if (block_interior(a.n, a.lookahead, a.aligned, blk)) comp_bwd_run_body<MASTER, true, FXS>(a, row, blk, cg_u, cg_g);
    else comp_bwd_run_body<MASTER, false, FXS>(a, row, blk, cg_u, cg_g);
}

// ---- backward, tracks, MIX-WISE (round 6): one workgroup = one 2048-sample block of one MIX, looping over (half of) the mix's tracks ----
// k_comp_bwd_run<tracks> runs one workgroup per (track row, block): 9 workgroup barriers, five dependent load phases and 16 operand tiles
// per block, of which the bus cotangent (4 tiles) is the same for all the tracks of the mix.  tools/cbr_timeline.py (wall-clock stamps of
// every workgroup): of a workgroup's 8.6 us, 3.3 us are spent in the barrier / exchange phases, 1.6 us in load phases, and the SIMDs issue
// 44 % of the time.  Here the tracks of a mix are walked by ONE workgroup (as k_apply_tracks does in the forward):
//   * the bus cotangent of the block (at i and at i + look-ahead) is loaded once and parked in LDS (lane-private slots);
//   * the NEXT track's first operands (u at i - L, g_s, the all-pole entry states) are requested while the current track computes;
//   * ONE workgroup barrier per track: the waves meet only to exchange their zero-entry aggregates of the adjoint smoother (block
//     aggregate to publish + state entering each wave).  The state entering the block is summed from the other blocks' granules by
//     every wave on its own (two granules per lane), and every wave walks the coefficient-gradient bank over ITS 512 samples (8 chunks x
//     6 sections = 48 lanes; tiles are wave-private, wave-level LDS hand-over); the waves' partial sums meet behind the next track's
//     barrier (double-buffered slots);
//   * the master channels' coefficient-gradient walks (2048 light workgroups in front of the row-wise grid, 11 us) are one more
//     short iteration of the workgroups of their mix.
// Registers: 168 = three waves per SIMD (the row-wise kernel is at 128 without any of the above; with the prefetch it spills 65).  768
// resident workgroups: the tracks of a mix are dealt to TWO workgroups (halves), 2048 items at cfg #2 = 2.67 rounds of four tracks.
// Same arithmetic per sample as comp_bwd_run_body (static curve, split adjoint state, packed all-pole walk); the order of the
// block-level sums differs (fixed, reproducible).  Handles what the bench / training path needs - compressor on, in-launch granules,
// fused coefficient gradients, no grad_tracks / grad_mixed_tracks / fx bus, look-ahead = one block; everything else keeps the row-wise kernel.
// MEASURED AND LEFT OFF (round 6, one box, tools/ab_bench.sh): 119 us against 97-99 us for the row-wise kernel.  The premise was
// wrong: with every bulk operand synthesised in registers instead of loaded (-DMST_NOMEM, wrong results) the row-wise kernel still takes
// 84 us (64 us without the 64-sample walks) and this one 116 us - both are bound by what they execute, not by what they wait for, and
// this form executes more (per-wave walks at 48 of 64 lanes, the ride, LDS parking) on three waves per SIMD instead of four.
#ifndef MST_CBM
#define MST_CBM 0  // A/B switch: 1 = the mix-wise kernel where it applies, 0 = k_comp_bwd_run<tracks> for every configuration
#endif
#if MST_CBM  // the whole kernel is compiled only into -DMST_CBM=1 builds (make BUILD=build_cbm OUT=../lib/cbm.so EXTRA=-DMST_CBM=1)
#ifndef MST_CBM_W
#define MST_CBM_W 3  // waves per SIMD asked of the mix-wise kernel (3: register cap 168)
#endif
#ifndef MST_CBM_PREFETCH
#define MST_CBM_PREFETCH 1  // 0: the next track's early operands are requested at the END of the track (fewer live registers, exposed latency)
#endif
constexpr int kMwChunks = 64 * CC / kEqChunk;     // 64-sample chunks per wave (8)
constexpr int kMwTile = kMwChunks * kCgPitch;     // floats of one wave's u (or du) tile
static_assert(kMwChunks == 8 && kCompWaves * kMwTile == kCgTile, "per-wave tiles fill the row-wise kernel's tiles");

__device__ __forceinline__ void park8(float* slot, const float* v) {
    *reinterpret_cast<float4*>(slot) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(slot + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
// a lane's eight samples into a chunk row of the walk's tile image (sample 8 j + 4 h + e at float 32 h + 4 j + e: coefgrad_fused)
__device__ __forceinline__ void tile8(float* at, const float* v) {
    *reinterpret_cast<float4*>(at) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(at + 32) = make_float4(v[4], v[5], v[6], v[7]);
}
__device__ __forceinline__ void fetch8(const float* slot, float* v) {
    const float4 a = *reinterpret_cast<const float4*>(slot), b = *reinterpret_cast<const float4*>(slot + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
// the all-pole walk of one (section, chunk) lane over its wave's tiles; sums over the eight chunk lanes of a section land in sums[5 cs ..]
__device__ __forceinline__ void mix_walk(const float* __restrict__ rc, const float* __restrict__ my_u, const float* __restrict__ my_g, const CgStates ws,
                                         const int cs, const int cc, float* __restrict__ sums) {
    const float ka1 = rc[RC_SOS + 5 * cs + 3], ka2 = rc[RC_SOS + 5 * cs + 4];
    const float kc1 = rc[RC_AP + 3 * cs], kc2 = rc[RC_AP + 3 * cs + 1], kib0 = rc[RC_AP + 3 * cs + 2];
    const float* mu = &my_u[cc * kCgPitch];
    const float* mg = &my_g[cc * kCgPitch];
    float db0 = 0.f;
    f2 w1 = {ws.wb1, ws.wa1}, w2 = {ws.wb2, ws.wa2}, acc1 = {0.f, 0.f}, acc2 = {0.f, 0.f};
    const f2 nk1 = {-kc1, -ka1}, nk2 = {-kc2, -ka2};
#pragma unroll 2
    for (int i4 = 0; i4 < kEqChunk; i4 += 4) {
        const int at = ((i4 >> 2) & 1) * 32 + (i4 >> 3) * 4;
        const float4 xv = *reinterpret_cast<const float4*>(&mu[at]);
        const float4 gv = *reinterpret_cast<const float4*>(&mg[at]);
        const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, gsv[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f2 xx = {xs[q], xs[q]}, gg = {gsv[q], -gsv[q]};
            const f2 wn = f2_fma(nk2, w2, f2_fma(nk1, w1, xx));
            db0 = fmaf(gsv[q], wn.x, db0);
            acc1 = f2_fma(gg, w1, acc1);
            acc2 = f2_fma(gg, w2, acc2);
            w2 = w1;
            w1 = wn;
        }
    }
    float e[5] = {db0 * kib0, acc1.x * kib0, acc2.x * kib0, acc1.y, acc2.y};
#pragma unroll
    for (int i = 0; i < 5; ++i) {  // sum over the eight chunk lanes of the section
        float s = e[i];
        s = dpp_add<0xb1, 0xf>(s);   // quad_perm [1,0,3,2]
        s = dpp_add<0x4e, 0xf>(s);   // quad_perm [2,3,0,1]
        s = dpp_add<0x141, 0xf>(s);  // row_half_mirror
        e[i] = s;
    }
    if (cc == 0) {
#pragma unroll
        for (int i = 0; i < 5; ++i) sums[5 * cs + i] = e[i];
    }
}

struct MixSlots {  // small LDS exchange slots of the mix-wise kernel (one copy for both template instantiations of the body)
    float agg[2][kCompWaves];
    float p[2][kCompWaves][CP_COUNT];
    float e[2][kCompWaves][EP_COUNT];
    float r[2][kCompWaves][EP_COUNT];
};
// tracks [t_lo, t_hi) of mix b; ride = the master channels whose coefficient-gradient walk this workgroup carries: [ride_lo, ride_hi) of {0, 1}
template <bool FAST>
__device__ __forceinline__ void comp_bwd_mix_body(const CompBwdArgs& a, const int b, const int blk, const int t_lo, const int t_hi, const int ride_lo,
                                                  const int ride_hi, const int sig0, float* __restrict__ cg_u, float* __restrict__ cg_g,
                                                  float* __restrict__ park, MixSlots& sl) {
    float (*sc_agg)[kCompWaves] = sl.agg;             // zero-entry wave aggregates of the adjoint smoother (by track parity)
    float (*sc_p)[kCompWaves][CP_COUNT] = sl.p;       // per-wave compressor partial sums
    float (*sc_e)[kCompWaves][EP_COUNT] = sl.e;       // per-wave coefficient-gradient partial sums
    float (*sc_r)[kCompWaves][EP_COUNT] = sl.r;       // ... of the master channels that ride here
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nblk = gridDim.x;
    const int64_t i0 = ((int64_t)blk * kWG + tid) * CC;
    const int L = a.lookahead;
    CBM_STAMP(9);
    // ---- once per workgroup: the bus cotangent of this block and of the next block, parked in lane-private LDS slots
    float* pkL = park + tid * CC;
    float* pkR = pkL + kWG * CC;
    float* pkLF = pkR + kWG * CC;
    float* pkRF = pkLF + kWG * CC;
    {
        const float* gL = a.gup + ((int64_t)b * 2 + 0) * a.gup_stride;
        const float* gR = a.gup + ((int64_t)b * 2 + 1) * a.gup_stride;
        float v0[CC], v1[CC], v2[CC], v3[CC];
        LD8S<FAST>(gL, i0, a.n, v0);
        LD8S<FAST>(gR, i0, a.n, v1);
        LD8S<FAST>(gL, i0 + L, a.n, v2);
        LD8S<FAST>(gR, i0 + L, a.n, v3);
        park8(pkL, v0);
        park8(pkR, v1);
        park8(pkLF, v2);
        park8(pkRF, v3);
    }
    // coefficient-gradient walk of this lane: section cs, chunk cc of the wave's eight
    const int cs = lane >> 3, cc = lane & 7;
    const bool walker = cs < kSections;
    auto load_states = [&](int sig) {
        CgStates w = {0.f, 0.f, 0.f, 0.f};
        if (walker) {
            const int64_t base = ((int64_t)sig * 24 + 4 * cs) * a.ap_nc_pad + (int64_t)blk * kCgChunks + wave * kMwChunks + cc;
            w.wa1 = a.ap_s0[base];
            w.wa2 = a.ap_s0[base + a.ap_nc_pad];
            w.wb1 = a.ap_s0[base + 2 * (int64_t)a.ap_nc_pad];
            w.wb2 = a.ap_s0[base + 3 * (int64_t)a.ap_nc_pad];
        }
        return w;
    };
    const int row0 = b * a.T;
    float xd[CC], g[CC], gprev;
    CgStates ws;
    {   // first track's early operands
        const float* u0 = a.u + (int64_t)(row0 + t_lo) * a.stride;
        const float* gs0 = a.gs + (int64_t)(row0 + t_lo) * a.stride;
        LD8S<FAST>(u0, i0 - L, a.n, xd);
        LD8<FAST>(gs0, i0, a.n, g);
        gprev = (i0 > 0 && i0 - 1 < a.n) ? gs0[i0 - 1] : 0.0f;
        ws = load_states(row0 + t_lo);
    }
    float* my_u = cg_u + wave * kMwTile;
    float* my_g = cg_g + wave * kMwTile;
    float* tile_at_u = &my_u[(lane >> 3) * kCgPitch + 4 * (lane & 7)];  // row image of coefgrad_fused: sample 8 j + 4 h + e at float 32 h + 4 j + e
    float* tile_at_g = &my_g[(lane >> 3) * kCgPitch + 4 * (lane & 7)];
    const int rl = 63 - lane, rw = kCompWaves - 1 - wave;  // position in the adjoint recursion's order (it runs from the row's end)

    // ---- the master channels that ride here: all that is left of their backward is the all-pole walk over (master EQ output, its
    // cotangent).  Their waves' sums are folded behind the first track's barrier.
    for (int ch = ride_lo; ch < ride_hi; ++ch) {
        const int j = 2 * b + ch, sig = sig0 + j;
        float xu[CC], dv[CC];
        LD8<FAST>(a.cg2_u + (int64_t)j * a.stride, i0, a.n, xu);
        LD8<FAST>(a.cg2_du + (int64_t)j * a.stride, i0, a.n, dv);
        const CgStates wm = load_states(sig);
        if (!FAST) {
#pragma unroll
            for (int i = 0; i < CC; ++i) dv[i] = (i0 + i < a.n) ? dv[i] : 0.0f;
        }
        tile8(tile_at_u, xu);
        tile8(tile_at_g, dv);
        wave_lds_sync();
        if (walker) mix_walk(a.cg2_rc + (int64_t)b * RC_STRIDE, my_u, my_g, wm, cs, cc, sc_r[ch][wave]);
        wave_lds_sync();
    }

    CBM_STAMP(11);
    for (int t = t_lo; t < t_hi; ++t) {
        const int row = row0 + t, par = (t - t_lo) & 1;
        const bool stamp = t == t_lo + 1;
        if (stamp) CBM_STAMP(0);
        const float* rc = a.rc + (int64_t)row * RC_STRIDE;
        const float* urow = a.u + (int64_t)row * a.stride;
        const float* gsrow = a.gs + (int64_t)row * a.stride;
        const CompK k = load_comp(rc);
        const float pl = rc[RC_PANL], pr = rc[RC_PANR];
        const float ac = rc[RC_ALPHA_C], l2a = rc[RC_LOG2A_C];
        // (1) this track's late operands: compressor input of the block, smoothed gain one block ahead
        float x0[CC], gF[CC];
        LD8<FAST>(urow, i0, a.n, x0);
        LD8S<FAST>(gsrow, i0 + L, a.n, gF);
        // (2) cotangent of the smoothed gain, zero-entry end value of the adjoint smoother over the lane's samples, pan / make-up sums
        float p[CP_COUNT];
#pragma unroll
        for (int i = 0; i < CP_COUNT; ++i) p[i] = 0.0f;
        float dgsv[CC];
        float zq = 0.0f;
        {
            // (the slot offset is laundered once per track: the parked values are loop-invariant, and hoisted out of the loop they are
            // the 32 registers the parking was for)
            int po = 0;
            pin_int(po);
            float gl[CC], gr[CC];
            fetch8(pkL + po, gl);
            fetch8(pkR + po, gr);
#pragma unroll
            for (int i = CC - 1; i >= 0; --i) {
                const float Gi = lin_gain(g[i], k);
                const float yv = xd[i] * Gi;
                dgsv[i] = (FAST || i0 + i < a.n) ? (pl * gl[i] + pr * gr[i]) * yv * kLn10Over20 : 0.0f;
                pin(dgsv[i]);
                zq = fmaf(k.alpha, zq, dgsv[i]);
                if (FAST || i0 + i < a.n) {
                    p[CP_MAKEUP] += dgsv[i];
                    p[CP_PANL] = fmaf(gl[i], yv, p[CP_PANL]);
                    p[CP_PANR] = fmaf(gr[i], yv, p[CP_PANR]);
                }
            }
            pin(p[CP_MAKEUP]);
            pin(p[CP_PANL]);
            pin(p[CP_PANR]);
        }
        if (stamp) CBM_STAMP(1);
        // (3) in-wave scan (from the last lane down), wave aggregates through LDS: the ONE workgroup barrier of the track
        float v = zq, pw = ac;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const float o = __shfl_down(v, d);
            if (rl >= d) v = fmaf(pw, o, v);
            pw *= pw;
        }
        if (lane == 0) sc_agg[par][wave] = v;  // zero-entry aggregate of this wave (pw == a^64 now)
        float ex = __shfl_down(v, 1);
        if (rl == 0) ex = 0.0f;
        lds_barrier();
        if (stamp) CBM_STAMP(2);
        float sw = 0.0f;  // zero-entry state entering this wave: the waves behind it in time come first
        for (int w = kCompWaves - 1; w > wave; --w) sw = fmaf(pw, sw, sc_agg[par][w]);
        gran_t* gq = a.gran + (int64_t)row * nblk;
        if (tid == 0) {  // block aggregate = state leaving the block towards earlier time with zero entering it
            float agg = 0.0f;
            for (int w = kCompWaves - 1; w >= 0; --w) agg = fmaf(pw, agg, sc_agg[par][w]);
            gran_publish(gq + blk, a.gran_near, agg);
        }
        const float Q0 = fmaf(__builtin_amdgcn_exp2f((float)rl * l2a), sw, ex);
        const float W = __builtin_amdgcn_exp2f((float)(rl + 64 * rw) * l2a);
        // the partial sums of the PREVIOUS track (first track: of the riding master channels): complete behind the barrier above
        if (t == t_lo) {
            for (int ch = ride_lo; ch < ride_hi; ++ch) {
                if (tid >= 64 && tid < 64 + EP_COUNT) {
                    const int q = tid - 64;
                    const float s = (sc_r[ch][0][q] + sc_r[ch][1][q]) + (sc_r[ch][2][q] + sc_r[ch][3][q]);
                    a.ep[((int64_t)(sig0 + 2 * b + ch) * nblk + blk) * EP_COUNT + q] = s;
                }
            }
        }
        if (t > t_lo) {
            if (tid < CP_COUNT) {
                const float s = (sc_p[par ^ 1][0][tid] + sc_p[par ^ 1][1][tid]) + (sc_p[par ^ 1][2][tid] + sc_p[par ^ 1][3][tid]);
                a.part[((int64_t)(row - 1) * nblk + blk) * CP_COUNT + tid] = s;
            } else if (tid >= 64 && tid < 64 + EP_COUNT) {
                const int q = tid - 64;
                const float s = (sc_e[par ^ 1][0][q] + sc_e[par ^ 1][1][q]) + (sc_e[par ^ 1][2][q] + sc_e[par ^ 1][3][q]);
                a.ep[((int64_t)(row - 1) * nblk + blk) * EP_COUNT + q] = s;
            }
        }
        // (4) the next track's early operands, in flight behind everything below
        float xdn[CC], gn[CC], gprevn = 0.0f;
        CgStates wsn = {0.f, 0.f, 0.f, 0.f};
        if (MST_CBM_PREFETCH && t + 1 < t_hi) {
            const float* un = urow + a.stride;
            const float* gsn = gsrow + a.stride;
            LD8S<FAST>(un, i0 - L, a.n, xdn);
            LD8<FAST>(gsn, i0, a.n, gn);
            gprevn = (i0 > 0 && i0 - 1 < a.n) ? gsn[i0 - 1] : 0.0f;
            wsn = load_states(row + 1);
        }
        // first look at the later blocks' aggregates of this row (one per lane now, a second round of 64 behind the arithmetic)
        const int j0 = blk + 1 + lane;
        const gran_t peek0 = j0 < nblk ? gran_load(gq + j0 + (MST_GRAN_NEAR ? a.gran_near : 0)) : 0ull;
        // (5) look-ahead branch: du[i] += (pl gl + pr gr)[i + L] G[i + L]
        float fwd[CC];
        {
            int po = 0;
            pin_int(po);
            float fl[CC], fr[CC];
            fetch8(pkLF + po, fl);
            fetch8(pkRF + po, fr);
#pragma unroll
            for (int i = 0; i < CC; ++i) {
                const bool liveF = FAST || i0 + i + L < a.n;
                fwd[i] = liveF ? (pl * fl[i] + pr * fr[i]) * lin_gain(gF[i], k) : 0.0f;
#if defined(MST_CBM_DBG) && MST_CBM_DBG == 1
                fwd[i] = 0.0f;
#endif
                pin(fwd[i]);
            }
        }
        if (stamp) CBM_STAMP(3);
        // (6) the compressor input goes to the wave's u tile
        tile8(tile_at_u, x0);
        __builtin_amdgcn_sched_barrier(0);
        // (7) static curve; every q-dependent quantity as (zero-state part, homogeneous part) of q[i] = zs[i] + alpha^(CC - i) Q
        f2 zp = {0.0f, 1.0f}, sA = {0.f, 0.f}, sF = {0.f, 0.f}, sP = {0.f, 0.f}, sW = {0.f, 0.f};
        const float kinvw = k.kappa * k.invw, kw2 = k.kappa * k.inv2w * k.invw, ke = k.oma * 8.685889638065035f;
        float du[CC], db[CC];
#pragma unroll
        for (int i = CC - 1; i >= 0; --i) {
            const bool live = FAST || i0 + i < a.n;
            const float side = x0[i];
            float tc;
            const float fval = curve_f(curve_t(side, k), k, tc);
            const float gp = (i > 0) ? g[i - 1] : gprev;
            const float kp = tc * kinvw;
            const float cKw = tc * (k.knee - tc) * kw2;
            const float cA = live ? fmaf(-k.kappa, fval, gp) : 0.0f;
            const float cE = (fabsf(side) >= kCompEps) ? kp * ke * __builtin_amdgcn_rcpf(side) : 0.0f;
            zp.x = fmaf(k.alpha, zp.x, dgsv[i]);
            zp.y *= k.alpha;
            sA = f2_fma(zp, f2{cA, cA}, sA);
            sF = f2_fma(zp, f2{fval, fval}, sF);
            sP = f2_fma(zp, f2{kp, kp}, sP);
            sW = f2_fma(zp, f2{cKw, cKw}, sW);
            du[i] = fmaf(zp.x, cE, fwd[i]);
            db[i] = zp.y * cE;
            pin(du[i]);
            pin(db[i]);
            __builtin_amdgcn_sched_barrier(0);
        }
        pin2(sA);
        pin2(sF);
        pin2(sP);
        pin2(sW);
        if (stamp) CBM_STAMP(4);
        // (8) state entering the block from later time: sum of the later blocks' aggregates, formed by every wave on its own
        float acc = 0.0f;
        {
            bool first = true;
            for (int j = j0; j < nblk; j += 64, first = false) {
                const float vj = (first && (peek0 >> 32) == 1) ? __int_as_float((int)(unsigned)peek0) : gran_wait(gq + j, a.gran_near, a.status);
                acc += __builtin_amdgcn_exp2f((float)(j - blk - 1) * (float)kWG * l2a) * vj;
            }
        }
        const float S = wave_sum(acc);
        if (stamp) CBM_STAMP(5);
        const float Q = fmaf(W, S, Q0);
#pragma unroll
        for (int i = 0; i < CC; ++i) du[i] = (FAST || i0 + i < a.n) ? fmaf(Q, db[i], du[i]) : 0.0f;
        p[CP_ALPHA] = fmaf(Q, sA.y, sA.x);
        p[CP_KAPPA] = k.oma * fmaf(Q, sF.y, sF.x);
        p[CP_THR] = -k.oma * fmaf(Q, sP.y, sP.x);
        p[CP_KNEE] = k.oma * fmaf(Q, sW.y, sW.x);
        // (9) cotangent of the EQ output to the wave's tile, then the all-pole walk of (section cs, chunk cc)
#if defined(MST_CBM_DBG) && MST_CBM_DBG == 2
#pragma unroll
        for (int i = 0; i < CC; ++i) du[i] = 1.0f;
#endif
        tile8(tile_at_g, du);
        wave_lds_sync();
        if (stamp) CBM_STAMP(6);
        if (walker) mix_walk(rc, my_u, my_g, ws, cs, cc, sc_e[par][wave]);
        if (stamp) CBM_STAMP(7);
        // (10) compressor partial sums of the wave
#pragma unroll
        for (int i = 0; i < CP_COUNT; ++i) {
            const float s = wave_sum(p[i]);
            if (lane == 0) sc_p[par][wave][i] = s;
        }
        wave_lds_sync();  // the tiles are rewritten by the next track's stores
        if (!MST_CBM_PREFETCH && t + 1 < t_hi) {
            const float* un = urow + a.stride;
            const float* gsn = gsrow + a.stride;
            LD8S<FAST>(un, i0 - L, a.n, xdn);
            LD8<FAST>(gsn, i0, a.n, gn);
            gprevn = (i0 > 0 && i0 - 1 < a.n) ? gsn[i0 - 1] : 0.0f;
            wsn = load_states(row + 1);
        }
        if (stamp) CBM_STAMP(8);
        // rotate
#pragma unroll
        for (int i = 0; i < CC; ++i) {
            xd[i] = xdn[i];
            g[i] = gn[i];
        }
        gprev = gprevn;
        ws = wsn;
    }
    lds_barrier();
    CBM_STAMP(10);
    {   // the last track's sums
        const int par = (t_hi - 1 - t_lo) & 1, row = row0 + t_hi - 1;
        if (tid < CP_COUNT) {
            const float s = (sc_p[par][0][tid] + sc_p[par][1][tid]) + (sc_p[par][2][tid] + sc_p[par][3][tid]);
            a.part[((int64_t)row * nblk + blk) * CP_COUNT + tid] = s;
        } else if (tid >= 64 && tid < 64 + EP_COUNT) {
            const int q = tid - 64;
            const float s = (sc_e[par][0][q] + sc_e[par][1][q]) + (sc_e[par][2][q] + sc_e[par][3][q]);
            a.ep[((int64_t)row * nblk + blk) * EP_COUNT + q] = s;
        }
    }
}

// grid (nblk, bs x mw_split): one workgroup per (mix, block, part of the tracks); the blocks of a mix on one XCD and walked from the row's
// end (a block waits for LATER blocks of its row only: they carry lower workgroup ids)
__global__ __launch_bounds__(kWG, MST_CBM_W) void k_comp_bwd_mix(CompBwdArgs a) {
    static_assert(kCompWaves == 4, "the mix-wise kernel is written for four-wave workgroups");
    __shared__ __attribute__((aligned(16))) float cg_u[kCgTile], cg_g[kCgTile];
    __shared__ __attribute__((aligned(16))) float park[4 * kWG * CC];
    __shared__ MixSlots sl;
    const int nsplit = a.mw_split, bs = (int)gridDim.y / nsplit;
    int rid, step;
    row_block_xcd(rid, step, 1, (int)gridDim.y, 0);
    const int b = rid % bs, h = rid / bs;  // part h of mix b: with bs % 8 == 0 both parts of a mix sit on XCD b % 8
    const int blk = gridDim.x - 1 - step;
    const int per = (a.T + nsplit - 1) / nsplit, t_lo = h * per, t_hi = t_lo + per < a.T ? t_lo + per : a.T;
    int ride_lo = 0, ride_hi = 0;
    if (a.cg2_rows) {
        ride_lo = nsplit == 2 ? h : 0;
        ride_hi = nsplit == 2 ? h + 1 : 2;
    }
    if (t_lo >= t_hi) return;
#ifdef MST_CBM_FASTONLY  // register-pressure diagnostics: the interior body alone
    comp_bwd_mix_body<true>(a, b, blk, t_lo, t_hi, ride_lo, ride_hi, bs * a.T, cg_u, cg_g, park, sl);
#else
    if (block_interior(a.n, a.lookahead, a.aligned, blk)) comp_bwd_mix_body<true>(a, b, blk, t_lo, t_hi, ride_lo, ride_hi, bs * a.T, cg_u, cg_g, park, sl);
    else comp_bwd_mix_body<false>(a, b, blk, t_lo, t_hi, ride_lo, ride_hi, bs * a.T, cg_u, cg_g, park, sl);
#endif
}

#endif  // MST_CBM

// ---- launch helpers ---------------------------------------------------------------------------------
void launch_comp_zs(int nch, const float* u, int64_t stride, const float* rc, float* zs, int nc_pad, int64_t n, int rows,
                    hipStream_t stream) {
    dim3 grid(nc_pad / kWG, rows), block(kWG);
    if (nch == 1) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_comp_zs<1>), grid, block, 0, stream, u, stride, rc, zs, nc_pad, n);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_comp_zs<2>), grid, block, 0, stream, u, stride, rc, zs, nc_pad, n);
}
void launch_apply_tracks(const TrackApplyArgs& a, int bs, hipStream_t stream) {
    if (a.fx) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_apply_tracks<true>), dim3(a.nc_pad / kWG, bs), dim3(kWG), 0, stream, a);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_apply_tracks<false>), dim3(a.nc_pad / kWG, bs), dim3(kWG), 0, stream, a);
}
void launch_apply_master(const MasterApplyArgs& a, int bs, hipStream_t stream) {
    hipLaunchKernelGGL(k_apply_master, dim3(a.nc_pad / kWG, bs), dim3(kWG), 0, stream, a);
}
void launch_comp_bwd(bool master, bool run, const CompBwdArgs& a, int rows, hipStream_t stream) {
    dim3 grid(a.nc_pad / kWG, rows + ((!master && run) ? a.cg2_rows : 0)), block(kWG);
    if (master && !run) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_comp_bwd_zs<true>), grid, block, 0, stream, a);
    else if (master && run) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_comp_bwd_run<true>), grid, block, 0, stream, a);
    else if (!master && !run) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_comp_bwd_zs<false>), grid, block, 0, stream, a);
    else if (a.gfx) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_comp_bwd_run<false, true>), grid, block, 0, stream, a);
#if MST_CBM
    else if (MST_CBM && kWG == 256 && a.comp_on && a.gran && a.ep && !a.du && !a.gmixed && a.T > 0 && rows % a.T == 0 && a.lookahead == kWG * CC &&
             (a.cg2_rows == 0 || a.cg2_rows == 2 * (rows / a.T))) {
        CompBwdArgs m = a;  // one workgroup per (mix, block, half of the tracks)
        m.mw_split = (a.T >= 4 && a.T % 2 == 0) ? 2 : 1;
        hipLaunchKernelGGL(k_comp_bwd_mix, dim3(a.nc_pad / kWG, (rows / a.T) * m.mw_split), block, 0, stream, m);
    }
#endif
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_comp_bwd_run<false>), grid, block, 0, stream, a);
}

}  // namespace mst

#ifdef MST_CBR_STAMPS
extern "C" int mst_debug_read_cbr_stamps(void* host, size_t bytes) {
    if (bytes > sizeof(mst::g_cbr_stamps)) bytes = sizeof(mst::g_cbr_stamps);
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(mst::g_cbr_stamps), bytes, 0, hipMemcpyDeviceToHost);
}
#endif
