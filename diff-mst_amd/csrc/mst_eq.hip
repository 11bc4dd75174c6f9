// mst_eq.hip - the cascaded-biquad (parametric EQ) kernels.
//
// Replaces dasp-pytorch's frequency-sampling `parametric_eq` (reference call sites
// mst/modules.py:237, 293; SURVEY A.3-A.4: twelve 2^19-point coefficient rFFTs + signal
// rFFT/irFFT per row) with a time-domain, chunk-parallel linear recurrence:
//
//   every lane owns kEqChunk consecutive samples of one signal row; the 64 lanes of a (one-wave) workgroup
//   cover a 4096-sample tile, staged through LDS in 16-sample slabs (next slab prefetched into registers) so that HBM sees only
//   coalesced 16-byte accesses while each lane reads its own contiguous slab.
//     *_zs   : run the chunk from ZERO state, keep only the 12-float end state  (z)
//     scan   : (mst_scan.hip) s0[c+1] = M s0[c] + z[c] gives every chunk's true start state
//     *_run  : re-run the chunk from its true start state and emit the output
//   SCAN1 (rows of <= 64 tiles): no scan kernel.  The zs kernel scans its 64 lane states in-wave with
//   M^(2^j) and writes every chunk's end state GIVEN A ZERO STATE AT THE TILE START; the run kernel scans
//   the <= 63 preceding tile aggregates of its row in-wave with M^(64 2^j) for the state S entering its
//   tile, and starts lane l from  (end state of lane l-1) + M^l S.
//   The adjoint (reverse-time) cascade uses the same skeleton with time reversed.
//   Coefficient gradients use the forward-only identity  dL/db_kj = <g, S^j (1/B_k) u>,
//   dL/da_kj = -<g, S^j (1/A_k) u>  (u = EQ output, g = its cotangent): twelve independent 2-state
//   all-pole recurrences on u, same zs / scan / run structure (k_allpole_zs, k_coefgrad).
#include "mst_kernels.h"
#include "mst_mat.h"
#include "mst_compdev.h"

namespace mst {

#ifndef MST_EQ_SLAB
#define MST_EQ_SLAB 16
#endif
constexpr int kSlab = MST_EQ_SLAB;  // samples per lane per LDS stage
constexpr int kLdw = kSlab + 4;   // padded LDS row (5 x 16 B, odd): conflict-free ds_read_b128 / ds_write_b128
constexpr int kNSlab = kEqChunk / kSlab;
constexpr int kSlabVec = kSlab / 4;              // float4 per lane per slab
constexpr int kFetch = kEqWG * kSlabVec / kEqWG;     // float4 each thread moves per slab (= kSlabVec)

// A slab (kEqWG lanes x kSlab samples) travels HBM -> registers -> LDS in two steps so that the NEXT slab's
// global loads are in flight while the current one is being filtered (all workgroups of these kernels
// are resident at once and would otherwise alternate, in lockstep, between a pure-memory and a
// pure-ALU phase).
struct SlabRegs {
    float4 v[kFetch];
};
// FAST (whole tile inside the row, row base 16-byte aligned - every tile but a ragged last one): plain 16-byte
// accesses; otherwise the guarded load4 / store4, whose bounds + alignment tests cost ~10 scalar/branch
// instructions per access.
template <bool FAST>
__device__ __forceinline__ void slab_fetch(SlabRegs& r, const float* __restrict__ row, int64_t tile_base, int j, int64_t n, int tid) {
#pragma unroll
    for (int q0 = 0; q0 < kFetch; ++q0) {
        const int q = tid + kEqWG * q0;      // float4 index inside the slab image
        const int lane = q / kSlabVec;     // owning lane
        const int i = (q % kSlabVec) * 4;
        const int64_t at = tile_base + (int64_t)lane * kEqChunk + j * kSlab + i;
        r.v[q0] = FAST ? *reinterpret_cast<const float4*>(row + at) : load4(row, at, n);
    }
}
__device__ __forceinline__ void slab_stash(const SlabRegs& r, float* __restrict__ tile, int tid) {
#pragma unroll
    for (int q0 = 0; q0 < kFetch; ++q0) {
        const int q = tid + kEqWG * q0;
        *reinterpret_cast<float4*>(&tile[(q / kSlabVec) * kLdw + (q % kSlabVec) * 4]) = r.v[q0];
    }
}
template <bool FAST>
__device__ __forceinline__ void slab_store(const float* __restrict__ tile, float* __restrict__ row, int64_t tile_base,
                                           int j, int64_t n, int tid) {
#pragma unroll
    for (int q0 = 0; q0 < kFetch; ++q0) {
        const int q = tid + kEqWG * q0;
        const int lane = q / kSlabVec;
        const int i = (q % kSlabVec) * 4;
        const int64_t at = tile_base + (int64_t)lane * kEqChunk + j * kSlab + i;
        const float4 v = *reinterpret_cast<const float4*>(&tile[lane * kLdw + i]);
        if (FAST) *reinterpret_cast<float4*>(row + at) = v;
        else store4(row, at, n, v);
    }
}

// Build-time variants of the staging pipeline (A/B on the GPU; defaults are the measured best):
//   MST_EQ_PREFETCH  1: the next slab's global loads are issued before the current slab is filtered
//                    0: each slab is fetched when it is needed (fewer live registers)
//   MST_EQ_SCHEDBAR  1: a scheduling barrier closes every slab iteration (single-wave workgroups have no
//                       real barrier, and the compiler otherwise hoists the staging of ALL slabs above the math)
#ifndef MST_EQ_PREFETCH
#define MST_EQ_PREFETCH 1
#endif
#ifndef MST_EQ_SCHEDBAR
#define MST_EQ_SCHEDBAR 0
#endif
template <bool FAST>
__device__ __forceinline__ void slab_first(SlabRegs& r, const float* __restrict__ row, int64_t tile_base, int j, int64_t n, int tid) {
    if (MST_EQ_PREFETCH) slab_fetch<FAST>(r, row, tile_base, j, n, tid);
}
template <bool FAST>
__device__ __forceinline__ void slab_enter(SlabRegs& r, const float* __restrict__ row, int64_t tile_base, int j, int64_t n, int tid) {
    if (!MST_EQ_PREFETCH) slab_fetch<FAST>(r, row, tile_base, j, n, tid);
}
template <bool FAST>
__device__ __forceinline__ void slab_next(SlabRegs& r, const float* __restrict__ row, int64_t tile_base, int j, bool valid, int64_t n, int tid) {
    if (MST_EQ_PREFETCH && valid) slab_fetch<FAST>(r, row, tile_base, j, n, tid);
}
// whole-workgroup decision: the tile lies inside the row and the row base allows 16-byte accesses
__device__ __forceinline__ bool tile_fast(const float* row, int64_t tile_base, int64_t n) {
    return tile_base + kTile <= n && !((uintptr_t)row & 15);
}
__device__ __forceinline__ void slab_fence() {
#if MST_EQ_SCHEDBAR
    __builtin_amdgcn_sched_barrier(0);
#endif
}

// ---- zero-state chunk end states of one tile on the matrix pipe (the arithmetic of k_eq_zs_mfma below, which explains it) ----------
// On return st[0 .. 12) = end state of chunk `lane` run from zero state.  `w` = the filter row's zero-state map (64 x 16).
// `tile`: >= 64 x 13 floats of LDS; ends with a wave-level LDS fence (the buffer may be reused at once).
typedef float f32x4_t __attribute__((vector_size(16)));
template <bool FAST>
__device__ __forceinline__ void zs_inline(const float* __restrict__ row, int64_t tile_base, int64_t n, const float* __restrict__ w,
                                          float* __restrict__ tile, float* st, int lane) {
    constexpr int kPitch = kStates + 1;
    const int li = lane & 15, g = lane >> 4;
    float4 x[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int64_t at = tile_base + (int64_t)(16 * m + li) * kEqChunk + 16 * s + 4 * g;
            x[m][s] = FAST ? *reinterpret_cast<const float4*>(row + at) : load4(row, at, n);
        }
    float wb[16];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int e = 0; e < 4; ++e) wb[4 * s + e] = w[(16 * s + 4 * g + e) * 16 + li];
    f32x4_t acc[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        acc[m] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const float xs[4] = {x[m][s].x, x[m][s].y, x[m][s].z, x[m][s].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(xs[e], wb[4 * s + e], acc[m], 0, 0, 0);
        }
    }
    if (li < kStates) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) tile[(16 * m + 4 * g + r) * kPitch + li] = acc[m][r];
    }
    wave_lds_sync();
#pragma unroll
    for (int dd = 0; dd < kStates; ++dd) st[dd] = tile[lane * kPitch + dd];
    wave_lds_sync();
}

// MODE_RUN = false: zero-state pass, writes z[sig][12][nc_pad]
// MODE_RUN = true : true pass from s0[sig][12][nc_pad], writes out
// FUSE_GC (forward run of mono rows only): the compressor's static curve is evaluated on the fresh EQ
// output and the zero-state envelope end value of every 2048-sample compressor block (= 32 lanes) is
// written to zs_comp[sig][block] - this replaces the separate k_comp_zs pass over the EQ output.
// FUSE_AP (forward run, when the call saves for backward): the all-pole bank of the coefficient-gradient pass
// (k_allpole_zs) advances on the fresh EQ output too and its zero-state chunk end states go to zp - the backward then
// starts at the all-pole carry scan, one pass over u less.
template <int DIR, bool MODE_RUN, bool FUSE_GC, bool SCAN1, bool FAST, bool FUSE_AP, bool ZSIN = false>
__device__ __forceinline__ void cascade_body(const float* __restrict__ in, int64_t in_stride,
                                                 float* __restrict__ out, int64_t out_stride,
                                                 const float* __restrict__ rc, int split,
                                                 const float* __restrict__ s0, float* __restrict__ z,
                                                 int nc_pad, int64_t n, float* __restrict__ zs_comp,
                                                 int nblk_comp, const float* __restrict__ pw1, int ntiles,
                                                 float* __restrict__ agg, float* __restrict__ tile, float* __restrict__ zp,
                                                 const int sig, const int bx, const ZsIn zi = ZsIn{}) {
    static_assert(!ZSIN || (MODE_RUN && SCAN1), "the in-launch zero-state pass belongs to the SCAN1 run");
    const int tid = threadIdx.x;
    const int64_t tile_base = (int64_t)bx * kEqWG * kEqChunk;
    const int chunk = bx * kEqWG + tid;
    const float* inrow = in + (int64_t)sig * in_stride;
    float* outrow = MODE_RUN ? out + (int64_t)sig * out_stride : nullptr;
    auto order = [](int jj) { return (DIR == EQ_FWD) ? jj : kNSlab - 1 - jj; };
    const int pos = DIR == EQ_FWD ? tid : kEqWG - 1 - tid;  // position of my chunk in recurrence order inside the tile
    const float* tab = SCAN1 ? pw1 + (int64_t)filter_row(sig, split) * kTri2 : nullptr;
    TabRegs tlo, thi;  // table sets of M (lanes of a tile) and M^64 (tiles of a row), mst_mat.h
    if (SCAN1) tab_fetch(tlo, tab, tid);
    float st[kStates];
    const int wt = DIR == EQ_FWD ? bx : ntiles - 1 - bx;  // my tile in recurrence order
    if (ZSIN) {
        // Round 5: no zero-state launch in front of this one.  The tile's zero-state chunk end states come off the matrix pipe right
        // here (the arithmetic of k_eq_zs_mfma below), its aggregate is published as 12 granules, and the aggregates of the tiles
        // before it are picked up as their workgroups publish them (all of them are resident or gone: they were dispatched earlier).
        zs_inline<FAST>(inrow, tile_base, n, zi.wz + (int64_t)filter_row(sig, split) * kWz, tile, st, tid);
        gran_t* gq = zi.gran + ((int64_t)sig * kMaxTiles1) * kStates;
        {   // st = inclusive scan of the chunk end states: the value at the last position is the tile aggregate
            float sc[kStates];
#pragma unroll
            for (int i = 0; i < kStates; ++i) sc[i] = st[i];
            tab_stash(tlo, tile, tid);
            wave_lds_sync();
            wave_scan_tri<DIR == EQ_ADJ>(sc, tile, pos);
            wave_lds_sync();
            gran_publish_vec<kStates>(gq + (int64_t)wt * kStates, zi.gran_near, sc, DIR == EQ_FWD ? kEqWG - 1 : 0, tid);
        }
        // forcing of the seeded scan below: the end state of the chunk one position back (nothing at position 0)
#pragma unroll
        for (int i = 0; i < kStates; ++i) st[i] = dpp_get<(DIR == EQ_ADJ) ? 0x130 : 0x138>(st[i]);
    }
    SlabRegs pre;
    slab_first<FAST>(pre, inrow, tile_base, order(0), n, tid);  // first slab in flight while constants load

    const float* coef = rc + (int64_t)filter_row(sig, split) * RC_STRIDE + RC_SOS;
    float c[5 * kSections];
#pragma unroll
    for (int i = 0; i < 5 * kSections; ++i) c[i] = coef[i];
    if (MODE_RUN && SCAN1) {
        tab_fetch(thi, tab + kTriFloats, tid);
        // s0 = the zs kernel's zero-state chunk end states.  The state entering chunk `pos` is the inclusive scan, over the
        // positions of the tile, of the forcing  { S at position 0, end state of chunk pos-1 elsewhere }  with S = the state
        // entering the tile = the scan over the aggregates of the preceding tiles.
        const int nb = DIR == EQ_FWD ? chunk - 1 : chunk + 1;
        if (!ZSIN) {
#pragma unroll
            for (int i = 0; i < kStates; ++i) st[i] = pos > 0 ? s0[((int64_t)sig * kStates + i) * nc_pad + nb] : 0.0f;
        }
#ifndef MST_DBG_NOSCAN
#define MST_DBG_NOSCAN 0  // timing diagnostics only: 1 skips the in-wave carry scans (wrong results)
#endif
        if (wt > 0 && !MST_DBG_NOSCAN) {
            float zz[kStates];  // lane q looks at the tile at recurrence position q
            if (ZSIN) {
                gran_read_vec<kStates>(zi.gran + ((int64_t)sig * kMaxTiles1 + tid) * kStates, zi.gran_near, zz, tid < wt, zi.status);
            } else {
#pragma unroll
                for (int d = 0; d < kStates; ++d) zz[d] = tid < wt ? agg[((int64_t)sig * kStates + d) * kMaxTiles1 + tid] : 0.0f;
            }
            tab_stash(thi, tile, tid);
            wave_lds_sync();
            wave_scan_tri<false>(zz, tile, tid);
#pragma unroll
            for (int d = 0; d < kStates; ++d) {
                const float S = __shfl(zz[d], wt - 1);
                if (pos == 0) st[d] = S;
            }
            wave_lds_sync();
        }
        if (!MST_DBG_NOSCAN) {
            tab_stash(tlo, tile, tid);
            wave_lds_sync();
            wave_scan_tri<DIR == EQ_ADJ>(st, tile, pos);
            wave_lds_sync();  // the slab image overwrites the table next
        }
    } else {
#pragma unroll
        for (int i = 0; i < kStates; ++i)
            st[i] = MODE_RUN ? s0[((int64_t)sig * kStates + i) * nc_pad + chunk] : 0.0f;
    }
    float* mine = &tile[tid * kLdw];
    CompK ck{};
    float zacc = 0.0f;
    if (FUSE_GC) ck = load_comp(rc + (int64_t)filter_row(sig, split) * RC_STRIDE);
    // the all-pole bank of k_coefgrad, zero state: 1/A_k on (wa1, wa2), b0/B_k on (wb1, wb2); a1, a2 are c[5s+3], c[5s+4]
    // held as pairs (1/A_k side, b0/B_k side): the two recurrences of a section are one packed multiply-add each (v_pk_fma_f32)
    f2 nk1[kSections], nk2[kSections], w1[kSections], w2[kSections];
    if (FUSE_AP) {
#pragma unroll
        for (int s = 0; s < kSections; ++s) {
            nk1[s] = f2{-c[5 * s + 3], -coef[RC_AP - RC_SOS + 3 * s]};
            nk2[s] = f2{-c[5 * s + 4], -coef[RC_AP - RC_SOS + 3 * s + 1]};
            w1[s] = w2[s] = f2{0.0f, 0.0f};
        }
    }

#ifndef MST_EQ_PACKED
#define MST_EQ_PACKED 0  // A/B: forward sections on state pairs (s1, s2) - y = b0 x + s1, then the two state updates as two packed
                         // multiply-adds on (b1, b2) x + (s2, 0), bit-identical to the scalar form: 47.4 us against 47.0 (the pair
                         // assembly costs the instruction it saves).  The variant s1' = (b1 x - a1 y) + s2 measured 44.7 against 46.5
                         // but moved the most ill-conditioned golden fixture from 2.13x to 2.57x the fp32 reference's distance from
                         // float64 - not taken
#endif
    f2 sp[kSections], cb[kSections], can[kSections];
    if (DIR == EQ_FWD && MST_EQ_PACKED) {
#pragma unroll
        for (int s = 0; s < kSections; ++s) {
            sp[s] = f2{st[2 * s], st[2 * s + 1]};
            cb[s] = f2{c[5 * s + 1], c[5 * s + 2]};
            can[s] = f2{-c[5 * s + 3], -c[5 * s + 4]};
        }
    }
    auto step_fwd = [&](float x) {
        if (!MST_EQ_PACKED) return cascade_step<float>(x, c, st);
#pragma unroll
        for (int s = 0; s < MST_DBG_SECTIONS; ++s) {
            const float y = fmaf(c[5 * s], x, sp[s].x);
            // same roundings as biquad_step: (n1, t) = (b1, b2) x + (s2, 0);  (s1', s2') = -(a1, a2) y + (n1, t)
            sp[s] = f2_fma(can[s], f2{y, y}, f2_fma(cb[s], f2{x, x}, f2{sp[s].y, 0.0f}));
            x = y;
        }
        return x;
    };
    for (int jj = 0; jj < kNSlab; ++jj) {
        const int j = order(jj);
        slab_enter<FAST>(pre, inrow, tile_base, j, n, tid);
        slab_stash(pre, tile, tid);
        wave_lds_sync();
        slab_next<FAST>(pre, inrow, tile_base, order(jj + 1 < kNSlab ? jj + 1 : jj), jj + 1 < kNSlab, n, tid);
        if (DIR == EQ_FWD) {
#pragma unroll
            for (int i4 = 0; i4 < kSlab; i4 += 4) {
                float4 v = *reinterpret_cast<float4*>(&mine[i4]);
                v.x = step_fwd(v.x);
                v.y = step_fwd(v.y);
                v.z = step_fwd(v.z);
                v.w = step_fwd(v.w);
                if (MODE_RUN) *reinterpret_cast<float4*>(&mine[i4]) = v;
                if (FUSE_AP) {
                    const float ys[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
#pragma unroll
                        for (int s = 0; s < kSections; ++s) {
                            const f2 wn = f2_fma(nk2[s], w2[s], f2_fma(nk1[s], w1[s], f2{ys[t], ys[t]}));
                            w2[s] = w1[s];
                            w1[s] = wn;
                        }
                    }
                }
                if (FUSE_GC) {
                    float dd;
                    zacc = fmaf(ck.alpha, zacc, ck.oma * gain_computer(v.x, ck, dd));
                    zacc = fmaf(ck.alpha, zacc, ck.oma * gain_computer(v.y, ck, dd));
                    zacc = fmaf(ck.alpha, zacc, ck.oma * gain_computer(v.z, ck, dd));
                    zacc = fmaf(ck.alpha, zacc, ck.oma * gain_computer(v.w, ck, dd));
                }
            }
        } else {
#pragma unroll
            for (int i4 = kSlab - 4; i4 >= 0; i4 -= 4) {
                float4 v = *reinterpret_cast<float4*>(&mine[i4]);
                v.w = cascade_adj_step<float>(v.w, c, st);
                v.z = cascade_adj_step<float>(v.z, c, st);
                v.y = cascade_adj_step<float>(v.y, c, st);
                v.x = cascade_adj_step<float>(v.x, c, st);
                if (MODE_RUN) *reinterpret_cast<float4*>(&mine[i4]) = v;
            }
        }
        wave_lds_sync();
        if (MODE_RUN) {
            slab_store<FAST>(tile, outrow, tile_base, j, n, tid);
            wave_lds_sync();  // the image is read by other lanes' stores before the next stash overwrites it
        }
        slab_fence();
    }
    if (DIR == EQ_FWD && MST_EQ_PACKED) {
#pragma unroll
        for (int s = 0; s < kSections; ++s) {
            st[2 * s] = sp[s].x;
            st[2 * s + 1] = sp[s].y;
        }
    }
    if (!MODE_RUN) {
#pragma unroll
        for (int i = 0; i < kStates; ++i) z[((int64_t)sig * kStates + i) * nc_pad + chunk] = st[i];
        if (SCAN1 && !MST_DBG_NOSCAN) {  // tile aggregate = the last position of the scan over the tile's chunk end states
            tab_stash(tlo, tile, tid);  // the slab buffer is free now
            wave_lds_sync();
            wave_scan_tri<DIR == EQ_ADJ>(st, tile, pos);
            if (pos == kEqWG - 1) {  // indexed by the tile's position in recurrence order
#pragma unroll
                for (int i = 0; i < kStates; ++i) agg[((int64_t)sig * kStates + i) * kMaxTiles1 + wt] = st[i];
            }
        }
    }
    if (FUSE_AP) {
#pragma unroll
        for (int s = 0; s < kSections; ++s) {
            const int64_t base = ((int64_t)sig * 24 + 4 * s) * nc_pad + chunk;
            zp[base] = w1[s].x;
            zp[base + nc_pad] = w2[s].x;
            zp[base + 2 * (int64_t)nc_pad] = w1[s].y;
            zp[base + 3 * (int64_t)nc_pad] = w2[s].y;
        }
    }
    if (FUSE_GC) {
        // fold the lane values of each compressor block (kBlkLanes EQ lanes = kWG * kCompChunk samples): v_l += a64^d v_(l-d) inside the
        // lane segments
        constexpr int kBlkLanes = kWG * kCompChunk / kEqChunk;
        const float l2a64 = 8.0f * rc[(int64_t)filter_row(sig, split) * RC_STRIDE + RC_LOG2A_C];  // log2(alpha^64)
        float p = __builtin_amdgcn_exp2f(l2a64);
        const int lseg = tid & (kBlkLanes - 1);
#pragma unroll
        for (int d = 1; d < kBlkLanes; d <<= 1) {
            const float o = __shfl_up(zacc, d);
            if (lseg >= d) zacc = fmaf(p, o, zacc);
            p *= p;
        }
        const int blk = chunk / kBlkLanes;
        if (lseg == kBlkLanes - 1 && blk < nblk_comp) zs_comp[(int64_t)sig * nblk_comp + blk] = zacc;
    }
}

template <int DIR, bool MODE_RUN, bool FUSE_GC = false, bool SCAN1 = false, bool FUSE_AP = false>
// The track rows' forward run (compressor curve + all-pole bank riding along) takes 152 registers uncapped = 3 waves per SIMD =
// 1.33 rounds of its 4096 one-wave workgroups at cfg #2; capped at 128 (28 spilled) it runs in one round: 57.8 -> 52.8 us.
// The 16-row master kernels are lone waves and lose 0.2-0.4 us to the same cap, so they stay uncapped.
#ifndef MST_EQ_RUN_W
#define MST_EQ_RUN_W 4
#endif
__global__ __launch_bounds__(kEqWG, (MODE_RUN && FUSE_GC && FUSE_AP) ? MST_EQ_RUN_W : 1) void k_cascade(const float* __restrict__ in, int64_t in_stride,
                                                 float* __restrict__ out, int64_t out_stride,
                                                 const float* __restrict__ rc, int split,
                                                 const float* __restrict__ s0, float* __restrict__ z,
                                                 int nc_pad, int64_t n, float* __restrict__ zs_comp = nullptr,
                                                 int nblk_comp = 0, const float* __restrict__ pw1 = nullptr, int ntiles = 0,
                                                 float* __restrict__ agg = nullptr, float* __restrict__ zp = nullptr) {
    static_assert(!FUSE_AP || (MODE_RUN && DIR == EQ_FWD), "the all-pole bank rides on the forward run only");
    // the slab image; a scan table set (kTriFloats) is staged through the same buffer
    __shared__ __attribute__((aligned(16))) float tile[kEqWG * kLdw > kTriFloats ? kEqWG * kLdw : kTriFloats];
    const int64_t tile_base = (int64_t)blockIdx.x * kTile;
    const bool fast = tile_fast(in + (int64_t)blockIdx.y * in_stride, tile_base, n) &&
                      (!MODE_RUN || !((uintptr_t)(out + (int64_t)blockIdx.y * out_stride) & 15));
    if (fast) cascade_body<DIR, MODE_RUN, FUSE_GC, SCAN1, true, FUSE_AP>(in, in_stride, out, out_stride, rc, split, s0, z, nc_pad, n, zs_comp, nblk_comp, pw1, ntiles, agg, tile, zp, blockIdx.y, blockIdx.x);
    else cascade_body<DIR, MODE_RUN, FUSE_GC, SCAN1, false, FUSE_AP>(in, in_stride, out, out_stride, rc, split, s0, z, nc_pad, n, zs_comp, nblk_comp, pw1, ntiles, agg, tile, zp, blockIdx.y, blockIdx.x);
}

// The SCAN1 run with the zero-state pass inside (round 5, ZsIn in mst_kernels.h): one launch instead of k_eq_zs_mfma + k_cascade.  The
// grid is walked so that the tiles of a row share an XCD and a tile's predecessors - in recurrence order: the adjoint cascade runs
// backwards in time - carry lower workgroup ids, i.e. were dispatched before it (row_block_xcd, mst_common.h).
template <int DIR, bool FUSE_GC, bool FUSE_AP>
__global__ __launch_bounds__(kEqWG, (FUSE_GC && FUSE_AP) ? MST_EQ_RUN_W : 1) void k_cascade_zsin(const float* __restrict__ in, int64_t in_stride,
                                                 float* __restrict__ out, int64_t out_stride,
                                                 const float* __restrict__ rc, int split, int nc_pad, int64_t n,
                                                 float* __restrict__ zs_comp, int nblk_comp, const float* __restrict__ pw1, int ntiles,
                                                 float* __restrict__ zp, ZsIn zi) {
    __shared__ __attribute__((aligned(16))) float tile[kEqWG * kLdw > kTriFloats ? kEqWG * kLdw : kTriFloats];
    int sig, step;
    row_block_xcd(sig, step);
    const int bx = DIR == EQ_FWD ? step : ntiles - 1 - step;
    const int64_t tile_base = (int64_t)bx * kTile;
    const bool fast = tile_fast(in + (int64_t)sig * in_stride, tile_base, n) && !((uintptr_t)(out + (int64_t)sig * out_stride) & 15);
    if (fast) cascade_body<DIR, true, FUSE_GC, true, true, FUSE_AP, true>(in, in_stride, out, out_stride, rc, split, nullptr, nullptr, nc_pad, n, zs_comp, nblk_comp, pw1, ntiles, nullptr, tile, zp, sig, bx, zi);
    else cascade_body<DIR, true, FUSE_GC, true, false, FUSE_AP, true>(in, in_stride, out, out_stride, rc, split, nullptr, nullptr, nc_pad, n, zs_comp, nblk_comp, pw1, ntiles, nullptr, tile, zp, sig, bx, zi);
}

// ---- zero-state pass on the matrix pipe ------------------------------------------------------------------------------
// The end state of a chunk run from zero state is a fixed linear map of its 64 samples: z = W^T x with W (64 x 12) the row's
// impulse-to-end-state responses (k_prep, fp64).  For the 64 chunks of a tile that is Z (64 x 12) = X (64 x 64) W - a GEMM the
// otherwise idle matrix pipe runs as 64 v_mfma_f32_16x16x4_f32 (exact fp32 FMA chains, k-ordered) instead of 64 x 30
// DEPENDENT vector FMAs per lane: the tile goes straight from HBM into A fragments (lane (i, g) of the MFMA takes four
// 16-byte pieces of row 16 m + i; the K order - sample 16 s + 4 g + e at step (s, e) - is the W fragments' order too, so no
// LDS staging and no shuffles), the results are written out in the chunk-state layout the run kernel reads, transposed
// through LDS to one chunk per lane, and scanned across the tile like before for the tile aggregate.
template <int DIR>
__global__ __launch_bounds__(kEqWG) void k_eq_zs_mfma(const float* __restrict__ in, int64_t in_stride, const float* __restrict__ wz, int split,
                                                     float* __restrict__ z, int nc_pad, int64_t n, const float* __restrict__ pw1, int ntiles,
                                                     float* __restrict__ agg) {
    constexpr int kPitch = kStates + 1;
    __shared__ __attribute__((aligned(16))) float tile[kEqWG * kPitch > kTriFloats ? kEqWG * kPitch : kTriFloats];
    const int lane = threadIdx.x, sig = blockIdx.y;
    const int64_t tile_base = (int64_t)blockIdx.x * kTile;
    const float* row = in + (int64_t)sig * in_stride;
    const int frow = filter_row(sig, split);
    TabRegs tlo;
    tab_fetch(tlo, pw1 + (int64_t)frow * kTri2, lane);
    float st[kStates];
    if (tile_fast(row, tile_base, n)) zs_inline<true>(row, tile_base, n, wz + (int64_t)frow * kWz, tile, st, lane);
    else zs_inline<false>(row, tile_base, n, wz + (int64_t)frow * kWz, tile, st, lane);
    // the chunk-state layout the run kernel reads: z[sig][state][chunk]
    const int chunk0 = blockIdx.x * kEqWG;
#pragma unroll
    for (int dd = 0; dd < kStates; ++dd) z[((int64_t)sig * kStates + dd) * nc_pad + chunk0 + lane] = st[dd];
    tab_stash(tlo, tile, lane);
    wave_lds_sync();
    const int pos = DIR == EQ_FWD ? lane : kEqWG - 1 - lane;
    wave_scan_tri<DIR == EQ_ADJ>(st, tile, pos);
    if (pos == kEqWG - 1) {
        const int wt = DIR == EQ_FWD ? (int)blockIdx.x : ntiles - 1 - (int)blockIdx.x;
#pragma unroll
        for (int dd = 0; dd < kStates; ++dd) agg[((int64_t)sig * kStates + dd) * kMaxTiles1 + wt] = st[dd];
    }
}
void launch_eq_zs_mfma(int dir, const float* in, int64_t in_stride, const float* wz, int split, float* z, int nc_pad, int64_t n, int nsig,
                       hipStream_t stream, const float* pw1, int ntiles, float* agg) {
    const dim3 grid(ntiles, nsig), block(kEqWG);
    if (dir == EQ_FWD) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_eq_zs_mfma<EQ_FWD>), grid, block, 0, stream, in, in_stride, wz, split, z, nc_pad, n, pw1, ntiles, agg);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_eq_zs_mfma<EQ_ADJ>), grid, block, 0, stream, in, in_stride, wz, split, z, nc_pad, n, pw1, ntiles, agg);
}

// ---- the all-pole bank's carry scan by ONE wave, riding on the master-bus forward run (round 4) -------------------------------
// k_scan<2> (mst_scan.hip) opened every backward with 15 us of its own; all it needs is what the TRACK rows' forward run left
// (zero-state chunk end states zp), and the launch that follows - the 16-row master EQ run - keeps one wave per SIMD busy and the
// memory system idle.  So the scan of the track rows' 12 R two-state systems rides there as extra one-wave workgroups: a row of
// <= 4096 chunk states is exactly one "tile" of the slab machinery above (lane l owns chunks 64 l .. 64 l + 63, staged through LDS
// in 16-element slabs, HBM sees coalesced 16-byte accesses):
//   pass 1  fold the lane's 64 chunks from zero:  s <- P s + z_c                      (P = A_f^64, table entry 0)
//   scan    inclusive Hillis-Steele over the lanes with P^(64 2^j) = table entry 1 + sh + j, 64 = KE 2^sh (the tables k_prep makes
//           for k_scan: (P^KE)^(2^j)), exclusive shift -> the state entering the lane's first chunk
//   pass 2  replay from there, emitting the state ENTERING every chunk in the layout k_coefgrad / coefgrad_fused read.
// Same recurrences, same fp32 products as k_scan; the association of the lane-level scan differs (64 x 64 instead of 512 x 8).
template <bool FAST>
__device__ __forceinline__ void allpole_scan_tile(const float* __restrict__ z0, float* __restrict__ o0, const float* __restrict__ tab, int nc,
                                                  int nc_pad, int sh, float* __restrict__ tileA, float* __restrict__ tileB, int tid) {
    const float* z1 = z0 + nc_pad;
    float* o1 = o0 + nc_pad;
    // the whole row in ONE memory round trip (2 x 4 slabs = 32 float4 per lane; the host kernel's register allocation - the cascade's -
    // is the budget) and kept for the replay: a lone wave has nothing else to hide a second trip behind
    SlabRegs ra[kNSlab], rb[kNSlab];
#pragma unroll
    for (int j = 0; j < kNSlab; ++j) {
        slab_fetch<FAST>(ra[j], z0, 0, j, nc, tid);
        slab_fetch<FAST>(rb[j], z1, 0, j, nc, tid);
    }
    const float p00 = tab[0], p01 = tab[1], p10 = tab[2], p11 = tab[3];
    float a = 0.0f, b = 0.0f;
    float* mineA = &tileA[tid * kLdw];
    float* mineB = &tileB[tid * kLdw];
#pragma unroll
    for (int j = 0; j < kNSlab; ++j) {
        slab_stash(ra[j], tileA, tid);
        slab_stash(rb[j], tileB, tid);
        wave_lds_sync();
#pragma unroll
        for (int i4 = 0; i4 < kSlab; i4 += 4) {
            const float4 va = *reinterpret_cast<const float4*>(&mineA[i4]), vb = *reinterpret_cast<const float4*>(&mineB[i4]);
            const float xa[4] = {va.x, va.y, va.z, va.w}, xb[4] = {vb.x, vb.y, vb.z, vb.w};
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float na = fmaf(p01, b, fmaf(p00, a, xa[t])), nb = fmaf(p11, b, fmaf(p10, a, xb[t]));
                a = na;
                b = nb;
            }
        }
        wave_lds_sync();
    }
    const int lanes = (nc + kEqChunk - 1) / kEqChunk;
    for (int j = 0; (1 << j) < lanes; ++j) {
        const float* T = tab + (1 + sh + j) * 4;
        const float oa = __shfl_up(a, 1 << j), ob = __shfl_up(b, 1 << j);
        if (tid >= (1 << j)) {
            a = fmaf(T[1], ob, fmaf(T[0], oa, a));
            b = fmaf(T[3], ob, fmaf(T[2], oa, b));
        }
    }
    float ea = __shfl_up(a, 1), eb = __shfl_up(b, 1);
    if (tid == 0) ea = eb = 0.0f;
#pragma unroll
    for (int j = 0; j < kNSlab; ++j) {
        slab_stash(ra[j], tileA, tid);
        slab_stash(rb[j], tileB, tid);
        wave_lds_sync();
#pragma unroll
        for (int i4 = 0; i4 < kSlab; i4 += 4) {
            const float4 va = *reinterpret_cast<const float4*>(&mineA[i4]), vb = *reinterpret_cast<const float4*>(&mineB[i4]);
            const float xa[4] = {va.x, va.y, va.z, va.w}, xb[4] = {vb.x, vb.y, vb.z, vb.w};
            float ya[4], yb[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                ya[t] = ea;
                yb[t] = eb;
                const float na = fmaf(p01, eb, fmaf(p00, ea, xa[t])), nb = fmaf(p11, eb, fmaf(p10, ea, xb[t]));
                ea = na;
                eb = nb;
            }
            *reinterpret_cast<float4*>(&mineA[i4]) = make_float4(ya[0], ya[1], ya[2], ya[3]);
            *reinterpret_cast<float4*>(&mineB[i4]) = make_float4(yb[0], yb[1], yb[2], yb[3]);
        }
        wave_lds_sync();
        slab_store<FAST>(tileA, o0, 0, j, nc, tid);
        slab_store<FAST>(tileB, o1, 0, j, nc, tid);
        wave_lds_sync();
    }
}

struct ApScanArgs {       // the scan jobs riding on a launch: job q = two-state system (signal row q / 12, filter q % 12)
    const float* z;       // (jobs, 2, nc_pad) zero-state chunk end states
    float* s0;            // (jobs, 2, nc_pad) out: state entering every chunk
    const float* tab;     // (filter rows x 12, kPow, 4) power tables (mst_params.hip)
    int jobs, nc, nc_pad, sh;
    int stereo;           // 0: job q uses table row q (track rows); 1: signal rows are L/R pairs sharing filter row (q / 12) / 2 (master buses)
};
// master-bus run (cascade role: blockIdx.y < nsig) + all-pole carry-scan jobs (blockIdx.y >= nsig).  DIR = EQ_FWD: the forward run
// (all-pole bank riding along) carries the TRACK rows' scans; DIR = EQ_ADJ: the adjoint run of the backward carries the MASTER rows' own
// (their coefficient-gradient walk happens later, in the tracks' compressor-backward launch)
template <int DIR, bool ZSIN>
__global__ __launch_bounds__(kEqWG) void k_master_run_apscan(const float* __restrict__ in, int64_t in_stride, float* __restrict__ out,
                                                           int64_t out_stride, const float* __restrict__ rc, const float* __restrict__ s0,
                                                           int nc_pad, int64_t n, const float* __restrict__ pw1, int ntiles,
                                                           float* __restrict__ agg, float* __restrict__ zp, int nsig, ApScanArgs sc, ZsIn zi) {
    __shared__ __attribute__((aligned(16))) float tile[2 * kEqWG * kLdw > kTriFloats ? 2 * kEqWG * kLdw : kTriFloats];
#ifndef MST_DBG_APSCAN
#define MST_DBG_APSCAN 0  // timing diagnostics only (wrong results): 1 = the scan role returns at once, 2 = the cascade role does
#endif
    if ((int)blockIdx.y >= nsig) {
        const int job = ((int)blockIdx.y - nsig) * gridDim.x + blockIdx.x;
        if (job >= sc.jobs || MST_DBG_APSCAN == 1) return;
        const float* z0 = sc.z + (int64_t)job * 2 * sc.nc_pad;
        float* o0 = sc.s0 + (int64_t)job * 2 * sc.nc_pad;
        const int trow = sc.stereo ? ((job / 12) >> 1) * 12 + job % 12 : job;
        const float* tab = sc.tab + (int64_t)trow * kPow * 4;
        if (sc.nc == kTile) allpole_scan_tile<true>(z0, o0, tab, sc.nc, sc.nc_pad, sc.sh, tile, tile + kEqWG * kLdw, threadIdx.x);
        else allpole_scan_tile<false>(z0, o0, tab, sc.nc, sc.nc_pad, sc.sh, tile, tile + kEqWG * kLdw, threadIdx.x);
        return;
    }
    if (MST_DBG_APSCAN == 2) return;
    int sig = blockIdx.y, bx = blockIdx.x;
    if (ZSIN) {  // the cascade workgroups (ids 0 .. ntiles nsig - 1, ahead of the scan jobs) exchange tile aggregates: see k_cascade_zsin
        int step;
        row_block_xcd(sig, step, 1, nsig);
        bx = DIR == EQ_FWD ? step : ntiles - 1 - step;
    }
    const int64_t tile_base = (int64_t)bx * kTile;
    const bool fast = tile_fast(in + (int64_t)sig * in_stride, tile_base, n) && !((uintptr_t)(out + (int64_t)sig * out_stride) & 15);
    constexpr bool AP = DIR == EQ_FWD;
    if (fast) cascade_body<DIR, true, false, true, true, AP, ZSIN>(in, in_stride, out, out_stride, rc, 0, s0, nullptr, nc_pad, n, nullptr, 0, pw1, ntiles, agg, tile, zp, sig, bx, zi);
    else cascade_body<DIR, true, false, true, false, AP, ZSIN>(in, in_stride, out, out_stride, rc, 0, s0, nullptr, nc_pad, n, nullptr, 0, pw1, ntiles, agg, tile, zp, sig, bx, zi);
}
void launch_master_run_apscan(const float* in, int64_t in_stride, float* out, int64_t out_stride, const float* rc, const float* s0, int nc_pad,
                              int64_t n, int nsig, hipStream_t stream, const float* pw1, int ntiles, float* agg, float* zp,
                              const float* sc_z, float* sc_s0, const float* sc_tab, int sc_jobs, int sc_nc, int sc_sh, int dir, const ZsIn* zi) {
    const ApScanArgs sc{sc_z, sc_s0, sc_tab, sc_jobs, sc_nc, nc_pad, sc_sh, dir == EQ_ADJ ? 1 : 0};
    const dim3 grid(ntiles, nsig + (sc_jobs + ntiles - 1) / ntiles), block(kEqWG);
    const ZsIn z0 = zi ? *zi : ZsIn{};
#define MST_LAUNCH_MRA(D, Z) \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_master_run_apscan<D, Z>), grid, block, 0, stream, in, in_stride, out, out_stride, rc, s0, nc_pad, n, pw1, ntiles, agg, zp, nsig, sc, z0)
    if (dir == EQ_FWD) {
        if (z0.wz) MST_LAUNCH_MRA(EQ_FWD, true);
        else MST_LAUNCH_MRA(EQ_FWD, false);
    } else {
        if (z0.wz) MST_LAUNCH_MRA(EQ_ADJ, true);
        else MST_LAUNCH_MRA(EQ_ADJ, false);
    }
#undef MST_LAUNCH_MRA
}

// ---- all-pole bank for the coefficient gradients ------------------------------------------------
// filter f = 2k : w = u - a1 w1 - a2 w2 (1/A_k);  f = 2k+1 : w = u - (b1/b0) w1 - (b2/b0) w2 (b0/B_k: the 1/b0 of 1/B_k is
// applied once, to the finished inner products - one multiply per sample and section less in all three kernels)
struct ApCoef {
    float a1[kSections], a2[kSections], ib0[kSections], c1[kSections], c2[kSections];
};
__device__ __forceinline__ void load_ap(const float* rcrow, ApCoef& k) {  // rcrow = the filter row's constants
#pragma unroll
    for (int s = 0; s < kSections; ++s) {
        k.a1[s] = rcrow[RC_SOS + 5 * s + 3];
        k.a2[s] = rcrow[RC_SOS + 5 * s + 4];
        k.c1[s] = rcrow[RC_AP + 3 * s];
        k.c2[s] = rcrow[RC_AP + 3 * s + 1];
        k.ib0[s] = rcrow[RC_AP + 3 * s + 2];
    }
}

// zero-state end states of the 12 all-pole filters per lane chunk: z[sig][24][nc_pad]
template <bool FAST>
__device__ __forceinline__ void allpole_zs_body(const float* __restrict__ u, int64_t u_stride,
                                                const float* __restrict__ rc, int split, float* __restrict__ z,
                                                int nc_pad, int64_t n, float* __restrict__ tile) {
    const int tid = threadIdx.x, sig = blockIdx.y;
    const int64_t tile_base = (int64_t)blockIdx.x * kEqWG * kEqChunk;
    const int chunk = blockIdx.x * kEqWG + tid;
    ApCoef k;
    load_ap(rc + (int64_t)filter_row(sig, split) * RC_STRIDE, k);
    float wa1[kSections], wa2[kSections], wb1[kSections], wb2[kSections];
#pragma unroll
    for (int s = 0; s < kSections; ++s) wa1[s] = wa2[s] = wb1[s] = wb2[s] = 0.0f;
    const float* urow = u + (int64_t)sig * u_stride;
    float* mine = &tile[tid * kLdw];
    SlabRegs pre;
    slab_first<FAST>(pre, urow, tile_base, 0, n, tid);
    for (int j = 0; j < kNSlab; ++j) {
        slab_enter<FAST>(pre, urow, tile_base, j, n, tid);
        slab_stash(pre, tile, tid);
        wave_lds_sync();
        slab_next<FAST>(pre, urow, tile_base, j + 1, j + 1 < kNSlab, n, tid);
#pragma unroll 4
        for (int i = 0; i < kSlab; ++i) {
            const float x = mine[i];
#pragma unroll
            for (int s = 0; s < kSections; ++s) {
                const float wa = fmaf(-k.a2[s], wa2[s], fmaf(-k.a1[s], wa1[s], x));
                wa2[s] = wa1[s];
                wa1[s] = wa;
                const float wb = fmaf(-k.c2[s], wb2[s], fmaf(-k.c1[s], wb1[s], x));
                wb2[s] = wb1[s];
                wb1[s] = wb;
            }
        }
        wave_lds_sync();
        slab_fence();
    }
#pragma unroll
    for (int s = 0; s < kSections; ++s) {
        const int64_t base = ((int64_t)sig * 24 + 4 * s) * nc_pad + chunk;
        z[base] = wa1[s];
        z[base + nc_pad] = wa2[s];
        z[base + 2 * (int64_t)nc_pad] = wb1[s];
        z[base + 3 * (int64_t)nc_pad] = wb2[s];
    }
}

__global__ __launch_bounds__(kEqWG) void k_allpole_zs(const float* __restrict__ u, int64_t u_stride,
                                                    const float* __restrict__ rc, int split, float* __restrict__ z,
                                                    int nc_pad, int64_t n) {
    __shared__ __attribute__((aligned(16))) float tile[kEqWG * kLdw];
    if (tile_fast(u + (int64_t)blockIdx.y * u_stride, (int64_t)blockIdx.x * kTile, n)) allpole_zs_body<true>(u, u_stride, rc, split, z, nc_pad, n, tile);
    else allpole_zs_body<false>(u, u_stride, rc, split, z, nc_pad, n, tile);
}

// coefficient-gradient partial sums: part[sig][block][30] = {db0 db1 db2 da1 da2} x 6 sections
// TWO waves per tile: both walk the same 64 chunks, wave h owns sections 3h .. 3h+2 (the bank's sections are independent:
// every one filters the same u and is correlated with the same g).  Wave 0 stages the u slabs, wave 1 the g slabs, each lane
// reads both images.  One wave per tile needed 144 registers (3 waves per SIMD, 5120 workgroups = 1.67 rounds at cfg #2);
// half the sections per wave halve the arithmetic per wave and the register count with it.
constexpr int kCgSec = kSections / 2;
template <bool FAST>
__device__ __forceinline__ void coefgrad_body(const float* __restrict__ u, int64_t u_stride,
                                              const float* __restrict__ g, int64_t g_stride,
                                              const float* __restrict__ rc, int split,
                                              const float* __restrict__ s0, int nc_pad,
                                              float* __restrict__ part, int64_t n, float* __restrict__ tile_u,
                                              float* __restrict__ tile_g, float (*red)[EP_COUNT / 2]) {
    const int tid = threadIdx.x & 63, half = threadIdx.x >> 6, sig = blockIdx.y;
    const int64_t tile_base = (int64_t)blockIdx.x * kEqWG * kEqChunk;
    const int chunk = blockIdx.x * kEqWG + tid;
    // Per sample and section: 1/A_k and b0/B_k advance (2 FMAs each) and their five inner products with the cotangent
    // accumulate (5 FMAs) - 9 plain FMAs.  (Round 1 packed the filter pair into v_pk_fma_f32: 6 packed operations, which on
    // gfx950 issue at HALF rate - 12 issue slots where these take 9, and a lane of one product was idle.)
    const float* rcrow = rc + (int64_t)filter_row(sig, split) * RC_STRIDE;
    float ka1[kCgSec], ka2[kCgSec], kc1[kCgSec], kc2[kCgSec], kib0[kCgSec];
    float wa1[kCgSec], wa2[kCgSec], wb1[kCgSec], wb2[kCgSec];
    float db0[kCgSec], db1[kCgSec], db2[kCgSec], da1[kCgSec], da2[kCgSec];
#pragma unroll
    for (int j = 0; j < kCgSec; ++j) {
        const int s = kCgSec * half + j;
        ka1[j] = rcrow[RC_SOS + 5 * s + 3];
        ka2[j] = rcrow[RC_SOS + 5 * s + 4];
        kc1[j] = rcrow[RC_AP + 3 * s];
        kc2[j] = rcrow[RC_AP + 3 * s + 1];
        kib0[j] = rcrow[RC_AP + 3 * s + 2];
        const int64_t base = ((int64_t)sig * 24 + 4 * s) * nc_pad + chunk;
        wa1[j] = s0[base];
        wa2[j] = s0[base + nc_pad];
        wb1[j] = s0[base + 2 * (int64_t)nc_pad];
        wb2[j] = s0[base + 3 * (int64_t)nc_pad];
        db0[j] = db1[j] = db2[j] = da1[j] = da2[j] = 0.0f;
    }
    const float* srow = half == 0 ? u + (int64_t)sig * u_stride : g + (int64_t)sig * g_stride;  // the stream this wave stages
    float* simg = half == 0 ? tile_u : tile_g;
    float* mu = &tile_u[tid * kLdw];
    float* mg = &tile_g[tid * kLdw];
    SlabRegs pre;
    slab_first<FAST>(pre, srow, tile_base, 0, n, tid);
    for (int j = 0; j < kNSlab; ++j) {
        slab_enter<FAST>(pre, srow, tile_base, j, n, tid);
        slab_stash(pre, simg, tid);
        lds_barrier();
        slab_next<FAST>(pre, srow, tile_base, j + 1, j + 1 < kNSlab, n, tid);
#pragma unroll 1
        for (int i4 = 0; i4 < kSlab; i4 += 4) {
            const float4 xv = *reinterpret_cast<const float4*>(&mu[i4]);
            const float4 gv = *reinterpret_cast<const float4*>(&mg[i4]);
            const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, gs[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float x = xs[t], gp = gs[t], gm = -gs[t];
#pragma unroll
                for (int q = 0; q < kCgSec; ++q) {
                    const float wa = fmaf(-ka2[q], wa2[q], fmaf(-ka1[q], wa1[q], x));
                    const float wb = fmaf(-kc2[q], wb2[q], fmaf(-kc1[q], wb1[q], x));
                    db0[q] = fmaf(gp, wb, db0[q]);
                    db1[q] = fmaf(gp, wb1[q], db1[q]);
                    db2[q] = fmaf(gp, wb2[q], db2[q]);
                    da1[q] = fmaf(gm, wa1[q], da1[q]);
                    da2[q] = fmaf(gm, wa2[q], da2[q]);
                    wa2[q] = wa1[q];
                    wa1[q] = wa;
                    wb2[q] = wb1[q];
                    wb1[q] = wb;
                }
            }
        }
        lds_barrier();
        slab_fence();
    }
    float acc[EP_COUNT / 2];
#pragma unroll
    for (int q = 0; q < kCgSec; ++q) {
        acc[5 * q + 0] = db0[q] * kib0[q];
        acc[5 * q + 1] = db1[q] * kib0[q];
        acc[5 * q + 2] = db2[q] * kib0[q];
        acc[5 * q + 3] = da1[q];
        acc[5 * q + 4] = da2[q];
    }
    // deterministic reduction: one DPP sum per value, lane 0 of each wave parks its 15
#pragma unroll
    for (int i = 0; i < EP_COUNT / 2; ++i) {
        const float v = wave_sum(acc[i]);
        if (tid == 0) red[half][i] = v;
    }
    lds_barrier();
    if (threadIdx.x < EP_COUNT)
        part[((int64_t)sig * gridDim.x + blockIdx.x) * EP_COUNT + threadIdx.x] = red[threadIdx.x / (EP_COUNT / 2)][threadIdx.x % (EP_COUNT / 2)];
}

#ifndef MST_COEFGRAD_W
#define MST_COEFGRAD_W 1  // min waves per SIMD asked of k_coefgrad (A/B switch)
#endif
__global__ __launch_bounds__(2 * kEqWG, MST_COEFGRAD_W) void k_coefgrad(const float* __restrict__ u, int64_t u_stride,
                                                  const float* __restrict__ g, int64_t g_stride,
                                                  const float* __restrict__ rc, int split,
                                                  const float* __restrict__ s0, int nc_pad,
                                                  float* __restrict__ part, int64_t n) {
    __shared__ __attribute__((aligned(16))) float tile_u[kEqWG * kLdw];
    __shared__ __attribute__((aligned(16))) float tile_g[kEqWG * kLdw];
    __shared__ float red[2][EP_COUNT / 2];
    const int64_t tile_base = (int64_t)blockIdx.x * kTile;
    if (tile_fast(u + (int64_t)blockIdx.y * u_stride, tile_base, n) && !((uintptr_t)(g + (int64_t)blockIdx.y * g_stride) & 15))
        coefgrad_body<true>(u, u_stride, g, g_stride, rc, split, s0, nc_pad, part, n, tile_u, tile_g, red);
    else
        coefgrad_body<false>(u, u_stride, g, g_stride, rc, split, s0, nc_pad, part, n, tile_u, tile_g, red);
}

// ---- host-side launch helpers (called from mst_console.hip) --------------------------------------
// pw1 != nullptr selects the SCAN1 kernels (rows of ntiles <= kMaxTiles1 tiles): `z` then receives, and `s0` must be,
// the zs launch's in-tile end states, and no carry-scan launch goes in between.
void launch_cascade(int dir, bool run, const float* in, int64_t in_stride, float* out, int64_t out_stride, const float* rc,
                    int split, const float* s0, float* z, int nc_pad, int64_t n, int nsig, hipStream_t stream, const float* pw1,
                    int ntiles, float* agg, float* zp) {
    const dim3 grid(pw1 ? ntiles : nc_pad / kEqWG, nsig), block(kEqWG);
    float* const nozs = nullptr;
    if (zp && dir == EQ_FWD && run) {  // forward run + all-pole bank
        if (pw1)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_cascade<EQ_FWD, true, false, true, true>), grid, block, 0, stream, in, in_stride, out,
                               out_stride, rc, split, s0, z, nc_pad, n, nozs, 0, pw1, ntiles, agg, zp);
        else
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_cascade<EQ_FWD, true, false, false, true>), grid, block, 0, stream, in, in_stride, out,
                               out_stride, rc, split, s0, z, nc_pad, n, nozs, 0, pw1, ntiles, agg, zp);
        return;
    }
#define MST_LAUNCH_CASCADE(D, R, S) \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_cascade<D, R, false, S>), grid, block, 0, stream, in, in_stride, out, out_stride, rc, \
                       split, s0, z, nc_pad, n, nozs, 0, pw1, ntiles, agg, nozs)
    if (pw1) {
        if (dir == EQ_FWD && !run) MST_LAUNCH_CASCADE(EQ_FWD, false, true);
        else if (dir == EQ_FWD) MST_LAUNCH_CASCADE(EQ_FWD, true, true);
        else if (!run) MST_LAUNCH_CASCADE(EQ_ADJ, false, true);
        else MST_LAUNCH_CASCADE(EQ_ADJ, true, true);
    } else {
        if (dir == EQ_FWD && !run) MST_LAUNCH_CASCADE(EQ_FWD, false, false);
        else if (dir == EQ_FWD) MST_LAUNCH_CASCADE(EQ_FWD, true, false);
        else if (!run) MST_LAUNCH_CASCADE(EQ_ADJ, false, false);
        else MST_LAUNCH_CASCADE(EQ_ADJ, true, false);
    }
#undef MST_LAUNCH_CASCADE
}

void launch_cascade_run_gc(const float* in, int64_t in_stride, float* out, int64_t out_stride, const float* rc, int split,
                           const float* s0, int nc_pad, int64_t n, int nsig, float* zs_comp, int nblk_comp, hipStream_t stream,
                           const float* pw1, int ntiles, float* agg, float* zp, const ZsIn* zi) {
    static_assert(kWG * kCompChunk % kEqChunk == 0 && kEqWG % (kWG * kCompChunk / kEqChunk) == 0, "a compressor block must be a power-of-two group of EQ lanes");
    const dim3 grid(pw1 ? ntiles : nc_pad / kEqWG, nsig), block(kEqWG);
    if (zi && zi->wz && pw1) {  // zero-state pass inside the launch
        if (zp) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_cascade_zsin<EQ_FWD, true, true>), grid, block, 0, stream, in, in_stride, out, out_stride, rc, split, nc_pad, n, zs_comp, nblk_comp, pw1, ntiles, zp, *zi);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_cascade_zsin<EQ_FWD, true, false>), grid, block, 0, stream, in, in_stride, out, out_stride, rc, split, nc_pad, n, zs_comp, nblk_comp, pw1, ntiles, zp, *zi);
        return;
    }
    if (zp) {
        if (pw1)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_cascade<EQ_FWD, true, true, true, true>), grid, block, 0, stream, in, in_stride, out, out_stride,
                               rc, split, s0, (float*)nullptr, nc_pad, n, zs_comp, nblk_comp, pw1, ntiles, agg, zp);
        else
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_cascade<EQ_FWD, true, true, false, true>), grid, block, 0, stream, in, in_stride, out, out_stride,
                               rc, split, s0, (float*)nullptr, nc_pad, n, zs_comp, nblk_comp, pw1, ntiles, agg, zp);
        return;
    }
    if (pw1)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_cascade<EQ_FWD, true, true, true>), grid, block, 0, stream, in, in_stride, out, out_stride,
                           rc, split, s0, (float*)nullptr, nc_pad, n, zs_comp, nblk_comp, pw1, ntiles, agg, (float*)nullptr);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_cascade<EQ_FWD, true, true, false>), grid, block, 0, stream, in, in_stride, out, out_stride,
                           rc, split, s0, (float*)nullptr, nc_pad, n, zs_comp, nblk_comp, pw1, ntiles, agg, (float*)nullptr);
}

void launch_allpole_zs(const float* u, int64_t u_stride, const float* rc, int split, float* z, int nc_pad, int64_t n,
                       int nsig, hipStream_t stream) {
    dim3 grid(nc_pad / kEqWG, nsig), block(kEqWG);
    hipLaunchKernelGGL(k_allpole_zs, grid, block, 0, stream, u, u_stride, rc, split, z, nc_pad, n);
}

void launch_coefgrad(const float* u, int64_t u_stride, const float* g, int64_t g_stride, const float* rc, int split,
                     const float* s0, int nc_pad, float* part, int64_t n, int nsig, hipStream_t stream) {
    dim3 grid(nc_pad / kEqWG, nsig), block(2 * kEqWG);
    hipLaunchKernelGGL(k_coefgrad, grid, block, 0, stream, u, u_stride, g, g_stride, rc, split, s0, nc_pad, part, n);
}

}  // namespace mst
