// mst_common.h - shared constants, row-constant layout and small device helpers for the
// MI355X (gfx950) Diff-MST mix-console kernels.  Wave = 64 lanes, workgroup = 256 lanes
// unless a kernel says otherwise.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/diffmst_hip.h"

namespace mst {

#ifndef MST_FUSE_COEFGRAD
#define MST_FUSE_COEFGRAD 1  // the track rows' coefficient-gradient sums are made inside k_comp_bwd_run (A/B switch: 0 = k_coefgrad for every row)
#endif
constexpr int kSections = 6;           // low shelf, 4 peaking, high shelf (reference mst/modules.py:125-143)
constexpr int kStates = 2 * kSections; // DF2T state of the whole cascade
#ifndef MST_COMP_WG
#define MST_COMP_WG 256
#endif
constexpr int kWG = MST_COMP_WG;       // lanes per workgroup in the compressor kernels (A/B: 64 = one wave per block, no workgroup barrier)
constexpr int kCompWaves = kWG / 64;
constexpr int kEqWG = 64;              // lanes per workgroup in the EQ kernels: ONE wave per 4096-sample tile, so
                                       // a CU hosts many independent tiles at different phases (no lockstep)
constexpr int kEqChunk = 64;           // samples one lane filters sequentially (EQ kernels)
constexpr int kCompChunk = 8;          // samples one lane owns in the compressor kernels
constexpr int kScanThreads = 512;      // lanes per row in the carry-scan kernels
constexpr int kScanLevels = 9;         // log2(kScanThreads)
constexpr int kPow = 1 + kScanLevels;  // matrices per scan table: M, then M^(K*2^j)
constexpr int kTile = kEqWG * kEqChunk;  // samples one single-wave workgroup of the EQ kernels covers (4096)
constexpr int kPow1 = 12;              // in-wave scans use M^(2^j), j = 0..5 (lanes of a tile) and 6..11 (tiles of a row)
constexpr int kTri2 = 2 * 592;         // ... stored as two block-triangular table sets per row (mst_mat.h: kTriFloats each)
constexpr int kFxDhChunks = 4;          // fx bus backward: frame chunks of the dH product (partials summed by the inverse transform)
constexpr int kWz = kEqChunk * 16;      // floats of one zero-state map: 64 samples x (12 states + 4 pad)
constexpr int kMaxTiles1 = 64;         // rows of up to 64 tiles (262144 samples) scan in-wave (no carry-scan kernel)

// ---- per-filter-row constants ("rc"), written by k_prep, floats -------------------------------
constexpr int RC_SOS = 0;      // 6 x {b0 b1 b2 a1 a2}; section 0's b carries the input-fader gain
constexpr int RC_THR = 30;     // compressor threshold dB
constexpr int RC_KAPPA = 31;   // 1/ratio - 1
constexpr int RC_KNEE = 32;    // knee width dB
constexpr int RC_ALPHA = 33;   // one-pole smoother coefficient
constexpr int RC_MAKEUP = 34;  // make-up gain dB
constexpr int RC_ALPHA_C = 35; // alpha^kCompChunk
constexpr int RC_PANL = 36;    // tracks: left pan gain   | master: output-fader linear gain
constexpr int RC_PANR = 37;    // tracks: right pan gain  | master: output-fader linear gain
constexpr int RC_GIN = 38;     // input-fader linear gain (already folded into section 0)
constexpr int RC_LOG2A_C = 39; // log2(alpha^kCompChunk), from fp64
constexpr int RC_SEND = 40;    // tracks: linear fx-bus send gain 10^(send_db/20)
constexpr int RC_AP = 44;      // 6 x {b1/b0, b2/b0, 1/b0}: the all-pole bank's constants, made by k_prep so that every kernel
                               // fetches them as wave-uniform scalars (a division in the kernel puts them into 18 vector registers)
constexpr int RC_STRIDE = 64;

// partial-sum slots of the compressor backward kernel
constexpr int CP_THR = 0, CP_KAPPA = 1, CP_KNEE = 2, CP_ALPHA = 3, CP_MAKEUP = 4, CP_PANL = 5, CP_PANR = 6, CP_SEND = 7;
constexpr int CP_COUNT = 8;
constexpr int EP_COUNT = 30;  // coefficient-gradient partial sums: 6 x {b0 b1 b2 a1 a2}

constexpr float kDbPerLog2 = 6.02059991327962390f;   // 20*log10(2)
constexpr float kLog2PerDb = 0.16609640474436813f;   // log2(10)/20
constexpr float kLn10Over20 = 0.11512925464970229f;  // d/dg 10^(g/20) = that * 10^(g/20)
constexpr float kCompEps = 1e-8f;                     // clamp of |side chain|

__host__ __device__ inline int64_t round_up(int64_t a, int64_t b) { return (a + b - 1) / b * b; }
// filter (parameter) row of signal row `sig`: rows below `split` are mono tracks, the rest stereo pairs
__host__ __device__ inline int filter_row(int sig, int split) { return sig < split ? sig : split + ((sig - split) >> 1); }

// ---- float pairs for the packed fp32 instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) ----
// Kernels bound by instruction issue hold values that share their arithmetic as pairs: one packed instruction does both halves.
// hipcc: a native two-float vector.  A host-only g++ build of these sources (the repository's CPU test harness compiles them that way)
// has no such type: there a plain struct with the same member names and operators.
#if defined(__clang__)
using f2 = float __attribute__((ext_vector_type(2)));
__host__ __device__ __forceinline__ f2 f2_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
#else
struct f2 {
    float x, y;
};
inline f2 operator*(f2 a, f2 b) { return f2{a.x * b.x, a.y * b.y}; }
inline f2 operator+(f2 a, f2 b) { return f2{a.x + b.x, a.y + b.y}; }
inline f2 f2_fma(f2 a, f2 b, f2 c) { return f2{fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y)}; }
#endif

// ---- one DF2T biquad step: y = b0 x + s1; s1' = b1 x - a1 y + s2; s2' = b2 x - a2 y ---------
template <typename T>
__device__ __forceinline__ T biquad_step(T x, const T* c, T& s1, T& s2) {
    T y = c[0] * x + s1;
    T n1 = c[1] * x + s2;
    s1 = n1 - c[3] * y;
    s2 = c[2] * x - c[4] * y;
    return y;
}
template <>
__device__ __forceinline__ float biquad_step<float>(float x, const float* c, float& s1, float& s2) {
    float y = fmaf(c[0], x, s1);
    float n1 = fmaf(c[1], x, s2);
    s1 = fmaf(-c[3], y, n1);
    s2 = fmaf(-c[4], y, c[2] * x);
    return y;
}

// forward cascade, state st[2k], st[2k+1] for section k
#ifndef MST_DBG_SECTIONS
#define MST_DBG_SECTIONS kSections  // timing diagnostics only: fewer sections = less math, wrong results
#endif
template <typename T>
__device__ __forceinline__ T cascade_step(T x, const T* c, T* st) {
#pragma unroll
    for (int k = 0; k < MST_DBG_SECTIONS; ++k) x = biquad_step<T>(x, c + 5 * k, st[2 * k], st[2 * k + 1]);
    return x;
}

// adjoint of one section, run in REVERSE time: p = g - a1 r1 - a2 r2; xbar = b0 p + b1 r1 + b2 r2
template <typename T>
__device__ __forceinline__ T biquad_adj_step(T g, const T* c, T& r1, T& r2) {
    T p = g - c[3] * r1 - c[4] * r2;
    T xb = c[0] * p + c[1] * r1 + c[2] * r2;
    r2 = r1;
    r1 = p;
    return xb;
}
// adjoint cascade: sections 5..0; state slot j belongs to section 5-j (processing order)
template <typename T>
__device__ __forceinline__ T cascade_adj_step(T g, const T* c, T* st) {
#pragma unroll
    for (int j = 0; j < kSections; ++j) {
        const int k = kSections - 1 - j;
        g = biquad_adj_step<T>(g, c + 5 * k, st[2 * j], st[2 * j + 1]);
    }
    return g;
}

// ---- guarded 4-wide global access (rows are 16-byte aligned; only the tail is ragged) ----------
__device__ __forceinline__ float4 load4(const float* __restrict__ row, int64_t i, int64_t n) {
    if (i + 3 < n && !((uintptr_t)(row + i) & 15)) return *reinterpret_cast<const float4*>(row + i);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n) v.x = row[i];
    if (i + 1 < n) v.y = row[i + 1];
    if (i + 2 < n) v.z = row[i + 2];
    if (i + 3 < n) v.w = row[i + 3];
    return v;
}
__device__ __forceinline__ void store4(float* __restrict__ row, int64_t i, int64_t n, float4 v) {
    if (i + 3 < n && !((uintptr_t)(row + i) & 15)) {
        *reinterpret_cast<float4*>(row + i) = v;
        return;
    }
    if (i < n) row[i] = v.x;
    if (i + 1 < n) row[i + 1] = v.y;
    if (i + 2 < n) row[i + 2] = v.z;
    if (i + 3 < n) row[i + 3] = v.w;
}
// same, with a signed offset that may run off either end (look-ahead delay lines)
__device__ __forceinline__ float4 load4_shift(const float* __restrict__ row, int64_t i, int64_t n) {
    if (i >= 0 && i + 3 < n && !((uintptr_t)(row + i) & 15)) return *reinterpret_cast<const float4*>(row + i);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i >= 0 && i < n) v.x = row[i];
    if (i + 1 >= 0 && i + 1 < n) v.y = row[i + 1];
    if (i + 2 >= 0 && i + 2 < n) v.z = row[i + 2];
    if (i + 3 >= 0 && i + 3 < n) v.w = row[i + 3];
    return v;
}

// LDS hand-over between the lanes of ONE wave (the 64-lane workgroups of the EQ kernels): a wave's DS instructions execute in
// order, so the lanes see each other's LDS writes without waiting - all that is needed is that the compiler keeps the order.
// __syncthreads() here costs `s_waitcnt vmcnt(0)` as well: it drains the global prefetch of the NEXT slab at every slab, which is
// exactly the latency the prefetch was issued early to hide.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Workgroup barrier that orders LDS only: `s_waitcnt lgkmcnt(0); s_barrier`.  __syncthreads() is a fence over ALL address spaces -
// it adds `vmcnt(0)`, i.e. every global load in flight (the next frame / slab, requested early on purpose) is drained at
// every barrier.  Use where the waves hand each other LDS data and nothing else.
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
// ... for a group of LANES threads that is the whole workgroup
template <int LANES>
__device__ __forceinline__ void group_lds_sync() {
    if (LANES <= 64) wave_lds_sync();
    else lds_barrier();
}

// Sum over the 64 lanes of a wave, the same value returned to every lane.  Six DPP additions (row shifts by 1, 2, 4, 8 leave
// each 16-lane row's total in its last lane, two row broadcasts carry the totals into lane 63) and one v_readlane, all
// full-rate VALU work - `v += __shfl_xor(v, m)` compiles to six ds_bpermute_b32 round trips through the LDS crossbar
// (180 of them closed every k_coefgrad tile).  Fixed order: results are reproducible.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, true));
}
__device__ __forceinline__ float wave_sum(float v) {
    v = dpp_add<0x111, 0xf>(v);  // row_shr:1
    v = dpp_add<0x112, 0xf>(v);  // row_shr:2
    v = dpp_add<0x114, 0xf>(v);  // row_shr:4
    v = dpp_add<0x118, 0xf>(v);  // row_shr:8
    v = dpp_add<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
    v = dpp_add<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// the same for a double (its two halves travel as two DPP moves): fixed order, every lane gets lane 63's total
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add_f64(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), CTRL, ROW_MASK, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, ROW_MASK, 0xf, true);
    return v + __longlong_as_double(((long long)hi << 32) | (long long)(unsigned)lo);
}
__device__ __forceinline__ double wave_sum_f64(double v) {
    v = dpp_add_f64<0x111, 0xf>(v);
    v = dpp_add_f64<0x112, 0xf>(v);
    v = dpp_add_f64<0x114, 0xf>(v);
    v = dpp_add_f64<0x118, 0xf>(v);
    v = dpp_add_f64<0x142, 0xa>(v);
    v = dpp_add_f64<0x143, 0xc>(v);
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), 63), hi = __builtin_amdgcn_readlane((int)(b >> 32), 63);
    return __longlong_as_double(((long long)hi << 32) | (long long)(unsigned)lo);
}

// ---- block aggregates handed from workgroup to workgroup INSIDE a launch (round 4) --------------------------------------------
// A zero-state pass used to be its own launch: every workgroup published the aggregate of its block and the run launch read the
// aggregates of the blocks before it.  The aggregate of block b does not depend on any other block, so the run kernel can publish
// it itself and pick up its predecessors' as soon as they appear: one launch and one pass over the saved signals less.
// Protocol = recipe R2 of cdna_hip_programming.md Guideline 16: ONE naturally aligned 8-byte {tag = 1, value} granule written by one
// agent-scope (sc1, write-through) store and polled with agent-scope loads - the data is the flag, no fence on either side.  The
// granules are zeroed by an EARLIER kernel of the same call (k_prep: every granule array; k_prep_bwd re-arms the backward's).
// No serial chain: a workgroup waits only for values that are computed from saved signals, never for another workgroup's wait.
// Deadlock freedom: block b waits for blocks that were dispatched before it (the grid is walked so that predecessors have lower
// workgroup ids; an XCD hands out its share of the ids in order), and every spin is bounded (MST_GRAN_SPINS: the wait gives up,
// returns NaN - which poisons the outputs that depend on it - and raises kStatusExchangeTimeout in the call's status word: the launch
// ends and the host side turns the status into an error, include/diffmst_hip.h).
typedef unsigned long long gran_t;
#ifndef MST_GRAN_SPINS
#define MST_GRAN_SPINS (1 << 22)
#endif
// Two copies per granule.  The FAR copy is the protocol above: a write-through (sc1) store that any
// CU of the chip will see - after a trip through the fabric, 3-5 us under load (measured: the run kernels lost 10-16 us to it).  The
// NEAR copy is a plain store: it stays in the writer's XCD L2, where an L1-bypassing load of a workgroup ON THE SAME XCD finds it
// within an L2 hit.  A reader polls both; a copy that shows the tag is the value (one 8-byte store each), whichever path it took,
// so the result never depends on where the workgroups were placed - only the latency does, and the grids that exchange granules
// are walked so that the blocks of one row share an XCD (row_block_xcd).
#ifndef MST_GRAN_NEAR
#define MST_GRAN_NEAR 1
#endif
// Status code a kernel raises (atomic max into the call's status word) when a bounded wait gives up: the values the launch wrote are
// poisoned (NaN) and the host side turns the code into an error (diffmst_hip/_desc.py: status_to_error) - larger than every
// range-check code (1000 - index - 1), so it survives the max.
constexpr int kStatusExchangeTimeout = 2000;
__device__ __forceinline__ void gran_give_up(int32_t* status) {
    if (status) atomicMax(status, kStatusExchangeTimeout);
}
__device__ __forceinline__ gran_t gran_load(const gran_t* g) { return __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// g: the FAR copies (rows x nblk); the NEAR copies follow `near_off` granules later (0: none)
__device__ __forceinline__ void gran_publish(gran_t* g, int64_t near_off, float v) {
#ifdef MST_GRAN_DROP_PUBLISH  // test build only (tests/test_exchange_timeout_gpu.py): nothing is ever published, every wait gives up
    return;
#endif
    const gran_t x = ((gran_t)1 << 32) | (gran_t)(unsigned)__float_as_int(v);
    if (MST_GRAN_NEAR && near_off) __hip_atomic_store(g + near_off, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_store(g, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float gran_wait(const gran_t* g, int64_t near_off, int32_t* status = nullptr) {
    const bool near = MST_GRAN_NEAR && near_off;
    gran_t x = gran_load(near ? g + near_off : g);
    for (int spins = 0; (x >> 32) != 1; ++spins) {
        if (spins >= MST_GRAN_SPINS) {
            gran_give_up(status);
            return __int_as_float(0x7fc00000);
        }
        if (near && !(spins & 1)) x = gran_load(g);  // alternate: far, near, far, ...
        else {
            __builtin_amdgcn_s_sleep(1);
            x = gran_load(near ? g + near_off : g);
        }
    }
    return __int_as_float((int)(unsigned)x);
}
// the same exchange for a small VECTOR (the 12-state aggregate of an EQ tile): NV consecutive granules.  Lane `src_lane` holds
// v[0 .. NV); lanes 0 .. NV-1 store one granule each (one store instruction per copy).
template <int NV>
__device__ __forceinline__ void gran_publish_vec(gran_t* g, int64_t near_off, const float* v, int src_lane, int lane) {
    float mine = 0.0f;
#pragma unroll
    for (int d = 0; d < NV; ++d) {
        const float t = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v[d]), src_lane));
        mine = lane == d ? t : mine;
    }
    if (lane < NV) gran_publish(g + lane, near_off, mine);
}
// lane-private read of NV granules at g (all NV loads of a poll are in flight together; near and far copies are polled in turn).
// Inactive lanes return zeros.  A wait that gives up returns NaN and raises kStatusExchangeTimeout in *status.
template <int NV>
__device__ __forceinline__ void gran_read_vec(const gran_t* g, int64_t near_off, float* out, bool active, int32_t* status) {
#pragma unroll
    for (int d = 0; d < NV; ++d) out[d] = 0.0f;
    if (!active) return;
    const bool has_near = MST_GRAN_NEAR && near_off;
    bool use_near = has_near;
    gran_t x[NV];
    for (int spins = 0;; ++spins) {
        const gran_t* p = use_near ? g + near_off : g;
        bool ok = true;
#pragma unroll
        for (int d = 0; d < NV; ++d) x[d] = gran_load(p + d);
#pragma unroll
        for (int d = 0; d < NV; ++d) ok = ok && (x[d] >> 32) == 1;
        if (ok) break;
        if (spins >= MST_GRAN_SPINS) {
            gran_give_up(status);
#pragma unroll
            for (int d = 0; d < NV; ++d) x[d] = 0x7fc00000ull;
            break;
        }
        if (has_near) use_near = !use_near;
        if (!has_near || use_near) __builtin_amdgcn_s_sleep(1);
    }
#pragma unroll
    for (int d = 0; d < NV; ++d) out[d] = __int_as_float((int)(unsigned)x[d]);
}

// Grid walk of the kernels that exchange granules.  Workgroup id L = blockIdx.x + gridDim.x blockIdx.y lands on XCD L % 8 and an XCD
// hands out its ids in ascending order (observed, MI355X_MICROARCH.md; nothing below is WRONG if it changes, see above): with rows % 8 == 0
// row r lives on XCD r % 8, and within an XCD the blocks of a row are walked in ascending `step` - the order in which a block's
// predecessors are dispatched before it.  Other row counts keep the plain (x = step, y = row) walk.
// `group` > 1: rows come in groups (the tracks of one mix) that read the same shared data (the bus cotangent of that mix): the whole
// group is kept on one XCD and its rows are interleaved block by block, so that the shared block is fetched once per XCD instead of
// once per row (with row r on XCD r % 8 the eight tracks of a mix sat on eight XCDs: 8 x 16 MB of bus cotangent over the fabric).
__device__ __forceinline__ void row_block_xcd(int& row, int& step, int group = 1, int rows = 0, int skip_rows = 0, int lin = -1) {
    // rows: 0 = the whole grid; skip_rows: grid rows in FRONT of the mapped ones that belong to another role of the launch;
    // lin >= 0: the workgroup's index among the mapped ones, handed over by a caller that interleaves another role (same residue mod 8
    // as its workgroup id, so that it still names the XCD)
    const int nblk = gridDim.x;
    if (rows <= 0) rows = gridDim.y - skip_rows;
    if (group >= 1 && rows % (8 * group) == 0 && (lin >= 0 || (nblk * skip_rows) % 8 == 0)) {
        const int L = lin >= 0 ? lin : blockIdx.x + nblk * ((int)blockIdx.y - skip_rows), xcd = L & 7, k = L >> 3;
        const int per_group = group * nblk, gi = k / per_group, rem = k % per_group;
        row = (gi * 8 + xcd) * group + rem % group;
        step = rem / group;
    } else if (lin >= 0) {
        row = lin / nblk;
        step = lin % nblk;
    } else {
        row = blockIdx.y - skip_rows;
        step = blockIdx.x;
    }
}

// ---- workspace layout (element offsets in floats), computed on the host ------------------------
struct Layout {
    int bs, T, R;            // R = bs*T track rows
    int64_t N;
    int ncE, ncE_pad;        // EQ lane-chunks per signal row (pad to kWG)
    int ncC, ncC_pad;        // compressor lane-chunks per row
    int nblkE, nblkC;        // workgroups per row in EQ / compressor kernels
    int nblkEt;              // coefficient-gradient partial rows per signal row: nblkC when k_comp_bwd_run makes them (MST_FUSE_COEFGRAD), else nblkE
    int KE, KC;              // chunks per scan thread
    int ntE;                 // 4096-sample EQ tiles per row
    int eq1;                 // 1: EQ carries scanned inside the zs / run kernels, 0: separate carry-scan kernel
    int apscan_fwd;          // 1: the track rows' all-pole carry scan rides on the master-bus forward run (mst_eq.hip), the backward scans the master rows only
    int apscan_sh;           // 64 = KE 2^apscan_sh
    // offsets
    int64_t rc_t, rc_m;                  // row constants
    int64_t powF_t, powF_m;              // forward cascade scan tables  rows x kPow x 144
    int64_t powA_t, powA_m;              // adjoint cascade scan tables
    int64_t powP_t, powP_m;              // all-pole scan tables rows x 12 x kPow x 4
    int64_t u_t, gs_t, bus, v_m, gs_m;   // saved signals
    int64_t zE_t, sE_t, zE_m, sE_m;      // EQ chunk states (z = zero-state end, s = true start)
    int64_t zS_t, sS_t, zS_m, sS_m;      // smoother chunk states
    int64_t du_m, dbus, du_t;            // backward signals
    int64_t zQ_t, sQ_t, zQ_m, sQ_m;      // smoother-adjoint chunk states
    int64_t zA_t, sA_t, zA_m, sA_m;      // EQ-adjoint chunk states
    int64_t zP_t, sP_t, zP_m, sP_m;      // all-pole (coefficient-gradient) chunk states
    int64_t cp_t, cp_m, ep_t, ep_m;      // partial sums
    int64_t pow1F_t, pow1F_m, pow1A_t, pow1A_m;  // in-wave scan tables rows x kTri2
    int64_t aggF_t, aggF_m, aggA_t, aggA_m;      // tile aggregates sigrows x 12 x kMaxTiles1 (forward / adjoint cascade)
    int64_t wzF_t, wzF_m, wzA_t, wzA_m;          // zero-state maps rows x 64 x 16: chunk end state = W^T chunk (mst_eq.hip, k_eq_zs_mfma)
    // granules (8 bytes each, float offsets here): block aggregates exchanged inside a launch.  Forward arrays first, then the
    // backward's: k_prep zeroes [gran_f, gran_f + 2 (gran_nf + gran_nb) floats), k_prep_bwd re-arms the backward part
    int64_t gran_f, gran_b;                      // gran_f: master smoother (bs x nblkC, twice: far + near copies); gran_b: adjoint smoother, tracks (R x nblkC, twice) then master (bs x nblkC, twice)
    int64_t gran_nf, gran_nb;                    // granule counts
    // round 5: tile aggregates of the EQ runs that carry their own zero-state pass (ZsIn, mst_kernels.h): (signal rows, kMaxTiles1, 12)
    // granules, far copies then near copies.  eqg_f (inside the forward block): track rows then master rows; eqg_b (inside the
    // backward block, re-armed by k_prep_bwd): the master rows' adjoint run
    int64_t eqg_f, eqg_b;
    int64_t eqg_nf, eqg_nb;                      // granules per copy
    // fx bus (only laid out when MST_USE_FX_BUS is set)
    int fxS, fxTaps, fxK, fxBlk, fxBlkIr;         // impulse-response samples, band-pass taps, partitions, signal blocks, ir-bwd blocks
    int64_t fx_rc, fx_in, fx_wnf, fx_ir, fx_Xs, fx_Hs, fx_Ys, fx_dXs, fx_dHs, fx_dir, fx_din, fx_part, fx_Hf, fx_mix, fx_dry;
    int64_t total;                       // floats
};

// floats between consecutive signal rows of the workspace arrays (u, g_s, bus, du ...): the row length rounded to 16 bytes, plus
// MST_ROW_PAD floats (A/B switch: rows exactly 2^k bytes apart put the same block of every row on the same memory channel)
#ifndef MST_ROW_PAD
#define MST_ROW_PAD 0
#endif
inline int64_t row_stride(int64_t n) { return round_up(n, 4) + MST_ROW_PAD; }

inline Layout make_layout(const mst_console_desc* d) {
    Layout L{};
    L.bs = d->bs;
    L.T = d->n_tracks;
    L.R = d->bs * d->n_tracks;
    L.N = d->n_samples;
    L.ncE = (int)((L.N + kEqChunk - 1) / kEqChunk);
    L.ncE_pad = (int)round_up(L.ncE, 256);
    L.ncC = (int)((L.N + kCompChunk - 1) / kCompChunk);
    L.ncC_pad = (int)round_up(L.ncC, kWG);
    L.nblkE = L.ncE_pad / kEqWG;
    L.nblkC = L.ncC_pad / kWG;
    L.nblkEt = MST_FUSE_COEFGRAD ? L.nblkC : L.nblkE;
    L.KE = (L.ncE + kScanThreads - 1) / kScanThreads;
    if (L.KE > 8) L.KE = (int)round_up(L.KE, 8);  // whole 8-chunk sub-spans: aligned 16-byte state accesses in k_scan
    L.KC = (L.ncC + kScanThreads - 1) / kScanThreads;
    L.ntE = (int)((L.N + kTile - 1) / kTile);
    L.eq1 = (L.ntE <= kMaxTiles1 && !(d->flags & MST_DEV_MULTIPASS_EQ)) ? 1 : 0;
    {   // needs: one tile of chunk states per row (= eq1), KE a power of two <= 64 (the k_scan tables hold (P^KE)^(2^j)), a master-bus run to ride on
        int sh = 0;
        while ((L.KE << sh) < 64) ++sh;
        const bool pow2 = (L.KE << sh) == 64;
#ifdef MST_APSCAN_SEPARATE
        const bool on = false;
#else
        const bool on = true;
#endif
        L.apscan_fwd = (on && L.eq1 && pow2 && (d->flags & MST_USE_MASTER_BUS) && (d->flags & MST_SAVE_FOR_BACKWARD)) ? 1 : 0;
        L.apscan_sh = sh;
    }
    int64_t o = 0;
    auto take = [&](int64_t n) {
        int64_t at = o;
        o += round_up(n, 64);  // keep every array 256-byte aligned
        return at;
    };
    const int64_t R = L.R, B = L.bs, N = row_stride(L.N);
    // Track rows come first and the master rows follow IN THE SAME ARRAY wherever a kernel can serve
    // both in one launch (signal rows [0,R) = tracks, [R, R+2bs) = master L/R; filter rows [0,R), [R,R+bs)).
    L.rc_t = take((R + B) * RC_STRIDE);
    L.rc_m = L.rc_t + R * RC_STRIDE;
    L.powF_t = take((R + B) * kPow * 144);
    L.powF_m = L.powF_t + R * kPow * 144;
    L.powA_t = take((R + B) * kPow * 144);
    L.powA_m = L.powA_t + R * kPow * 144;
    L.powP_t = take((R + B) * 12 * kPow * 4);
    L.powP_m = L.powP_t + R * 12 * kPow * 4;
    L.u_t = take((R + 2 * B) * N);
    L.v_m = L.u_t + R * N;
    L.gs_t = take(R * N);
    L.bus = take(B * 2 * N);
    L.gs_m = take(B * N);
    L.zE_t = take(R * 12 * L.ncE_pad);
    L.sE_t = take(R * 12 * L.ncE_pad);
    L.zE_m = take(B * 2 * 12 * L.ncE_pad);
    L.sE_m = take(B * 2 * 12 * L.ncE_pad);
    L.zS_t = take(R * L.ncC_pad);
    L.sS_t = take(R * L.ncC_pad);
    L.zS_m = take(B * L.ncC_pad);
    L.sS_m = take(B * L.ncC_pad);
    L.du_t = take((R + 2 * B) * N);
    L.du_m = L.du_t + R * N;
    L.dbus = take(B * 2 * N);
    L.zQ_t = take(R * L.ncC_pad);
    L.sQ_t = take(R * L.ncC_pad);
    L.zQ_m = take(B * L.ncC_pad);
    L.sQ_m = take(B * L.ncC_pad);
    L.zA_t = take(R * 12 * L.ncE_pad);
    L.sA_t = take(R * 12 * L.ncE_pad);
    L.zA_m = take(B * 2 * 12 * L.ncE_pad);
    L.sA_m = take(B * 2 * 12 * L.ncE_pad);
    L.zP_t = take((R + 2 * B) * 24 * L.ncE_pad);
    L.zP_m = L.zP_t + R * 24 * L.ncE_pad;
    L.sP_t = take((R + 2 * B) * 24 * L.ncE_pad);
    L.sP_m = L.sP_t + R * 24 * L.ncE_pad;
    L.cp_t = take(R * L.nblkC * CP_COUNT);
    L.cp_m = take(B * L.nblkC * CP_COUNT);
    L.ep_t = take((R + 2 * B) * L.nblkEt * EP_COUNT);
    L.ep_m = L.ep_t + R * L.nblkEt * EP_COUNT;
    L.pow1F_t = take((R + B) * kTri2);
    L.pow1F_m = L.pow1F_t + R * kTri2;
    L.pow1A_t = take((R + B) * kTri2);
    L.pow1A_m = L.pow1A_t + R * kTri2;
    L.aggF_t = take((R + 2 * B) * kStates * kMaxTiles1);
    L.aggF_m = L.aggF_t + R * kStates * kMaxTiles1;
    L.aggA_t = take((R + 2 * B) * kStates * kMaxTiles1);
    L.aggA_m = L.aggA_t + R * kStates * kMaxTiles1;
    L.wzF_t = take((R + B) * kWz);
    L.wzF_m = L.wzF_t + R * kWz;
    L.wzA_t = take((R + B) * kWz);
    L.wzA_m = L.wzA_t + R * kWz;
    L.eqg_nf = (R + 2 * B) * kMaxTiles1 * kStates;
    L.eqg_nb = 2 * B * kMaxTiles1 * kStates;
    L.gran_nf = 2 * B * L.nblkC + 2 * L.eqg_nf;         // far copies, then near copies (mst_common.h: gran_publish)
    L.gran_nb = 2 * (R + B) * L.nblkC + 2 * L.eqg_nb;
    L.gran_f = take(2 * (L.gran_nf + L.gran_nb));
    L.gran_b = L.gran_f + 2 * L.gran_nf;
    L.eqg_f = L.gran_f + 2 * (2 * B * L.nblkC);
    L.eqg_b = L.gran_b + 2 * (2 * (R + B) * L.nblkC);
    if (d->flags & MST_USE_FX_BUS) {
        L.fxS = d->fx_ir_samples;
        L.fxTaps = d->fx_bandpass_taps;
        L.fxK = L.fxS / 4096;
        L.fxBlk = (int)((L.N + 4095) / 4096);
        L.fxBlkIr = (L.fxS + 1023) / 1024;
        L.fx_rc = take(B * 24);
        L.fx_in = take(B * 2 * N);
        L.fx_wnf = take(B * 2 * 12 * (int64_t)L.fxS);
        L.fx_ir = take(B * 2 * (int64_t)L.fxS);
        L.fx_Xs = take(B * (int64_t)L.fxBlk * 8192 * 2);
        L.fx_Hs = take(B * (int64_t)L.fxK * 8192 * 2);
        L.fx_Ys = take(B * (int64_t)L.fxBlk * 8192 * 2);
        L.fx_dXs = take(B * (int64_t)L.fxBlk * 8192 * 2);
        L.fx_dHs = take(B * (int64_t)L.fxK * 8192 * 2 * kFxDhChunks);  // partial dH spectra, one set per frame chunk
        L.fx_dir = take(B * 2 * (int64_t)L.fxS);
        L.fx_din = take(B * 2 * N);
        L.fx_part = take(B * (int64_t)L.fxBlkIr * 24);
        L.fx_Hf = take((int64_t)12 * 8192 * 2);  // conjugated spectra of the twelve band-pass filters
        L.fx_mix = take(B);                       // wet/dry mix per batch item (1 unless forward_mix_console hands one over)
        L.fx_dry = take(B * (int64_t)L.fxBlk);    // per-block partial sums <dbus, fx_in> of the mix gradient
    }
    L.total = o;
    return L;
}

}  // namespace mst
