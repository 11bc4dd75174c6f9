// tests/hostsim - a tiny host-side SIMULATOR of the HIP execution model.
//
// TEST INFRASTRUCTURE ONLY.  It lets `pytest -m "not gpu"` run the *unchanged*
// kernel sources of diff-mst_amd/csrc on the CPU (g++ sees this header instead
// of ROCm's <hip/hip_runtime.h>) so that indexing / scan / carry logic is checked
// against the oracle in the build container, which has no GPU.  It is never
// linked into the product library (libdiffmst_hip.so is built by hipcc for
// gfx950 only) and the mst package refuses to load it.
//
// Model: one OS thread per HIP thread of a block, blocks executed one after the
// other; __syncthreads() = std::barrier over the block; wave64 shuffles through
// a per-wave exchange buffer + per-wave barrier.  `__shared__` maps to `static`
// (safe because blocks run sequentially).
#pragma once
#include <atomic>
#include <barrier>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <thread>
#include <tuple>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __launch_bounds__(...)
#define __restrict__ __restrict

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float2 { float x, y; };
struct alignas(16) float4 { float x, y, z, w; };
struct double2 { double x, y; };
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline float2 make_float2(float x, float y) { return float2{x, y}; }

typedef int hipError_t;
typedef void* hipStream_t;
#define hipSuccess 0
#define hipErrorInvalidValue 1
static inline hipError_t hipGetLastError() { return 0; }
static inline hipError_t hipPeekAtLastError() { return 0; }
static inline const char* hipGetErrorString(hipError_t) { return "hostsim"; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return 0; }
enum { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) { memcpy(d, s, n); return 0; }
#define hipMemcpyDeviceToDevice 3
typedef void* hipEvent_t;
#define hipEventDisableTiming 2
#define hipStreamNonBlocking 1
static inline hipError_t hipGetDevice(int* d) { *d = 0; return 0; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = nullptr; return 0; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return 0; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return 0; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return 0; }

namespace hostsim {
struct BlockCtx {
    unsigned nthreads = 0;
    std::unique_ptr<std::barrier<>> block_bar;
    std::vector<std::unique_ptr<std::barrier<>>> wave_bar;
    std::vector<uint64_t> xchg;  // one 64-bit slot per thread
};
extern thread_local dim3 t_threadIdx, t_blockIdx, t_blockDim, t_gridDim;
extern thread_local BlockCtx* t_ctx;
extern thread_local unsigned t_tid;

template <typename T>
inline T shfl_idx(T v, unsigned src_lane) {
    static_assert(sizeof(T) <= 8, "shuffle payload");
    BlockCtx* c = t_ctx;
    unsigned wave = t_tid >> 6;
    uint64_t bits = 0;
    memcpy(&bits, &v, sizeof(T));
    c->xchg[t_tid] = bits;
    c->wave_bar[wave]->arrive_and_wait();
    unsigned src = (wave << 6) + (src_lane & 63u);
    uint64_t got = (src < c->nthreads) ? c->xchg[src] : bits;
    c->wave_bar[wave]->arrive_and_wait();
    T out;
    memcpy(&out, &got, sizeof(T));
    return out;
}

template <typename K, typename... Args>
void launch(K kernel, dim3 grid, dim3 block, Args... args) {
    unsigned nt = block.x * block.y * block.z;
    BlockCtx ctx;
    ctx.nthreads = nt;
    ctx.block_bar = std::make_unique<std::barrier<>>(nt);
    unsigned nw = (nt + 63) / 64;
    for (unsigned w = 0; w < nw; ++w) {
        unsigned cnt = (w == nw - 1) ? nt - 64 * w : 64;
        ctx.wave_bar.emplace_back(std::make_unique<std::barrier<>>(cnt));
    }
    ctx.xchg.assign(nt, 0);
    auto worker = [&](unsigned tid) {
        t_ctx = &ctx;
        t_tid = tid;
        t_blockDim = block;
        t_gridDim = grid;
        t_threadIdx = dim3(tid % block.x, (tid / block.x) % block.y, tid / (block.x * block.y));
        for (unsigned bz = 0; bz < grid.z; ++bz)
            for (unsigned by = 0; by < grid.y; ++by)
                for (unsigned bx = 0; bx < grid.x; ++bx) {
                    t_blockIdx = dim3(bx, by, bz);
                    kernel(args...);
                    ctx.block_bar->arrive_and_wait();
                }
    };
    std::vector<std::thread> th;
    th.reserve(nt);
    for (unsigned t = 0; t < nt; ++t) th.emplace_back(worker, t);
    for (auto& t : th) t.join();
}
}  // namespace hostsim

#define threadIdx (hostsim::t_threadIdx)
#define blockIdx (hostsim::t_blockIdx)
#define blockDim (hostsim::t_blockDim)
#define gridDim (hostsim::t_gridDim)

#define HIP_KERNEL_NAME(...) __VA_ARGS__
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    hostsim::launch(kernel, dim3(grid), dim3(block), ##__VA_ARGS__)

static inline void __syncthreads() { hostsim::t_ctx->block_bar->arrive_and_wait(); }
// wave-scope fence / barrier: the simulator's lanes are OS threads, so "the wave runs in lockstep" has to be a real rendezvous
static inline void __builtin_amdgcn_wave_barrier() { hostsim::t_ctx->wave_bar[hostsim::t_tid >> 6]->arrive_and_wait(); }
#define __builtin_amdgcn_fence(order, ...) __atomic_thread_fence(order)
static inline void __builtin_amdgcn_s_barrier() { hostsim::t_ctx->block_bar->arrive_and_wait(); }
template <typename T> inline T __shfl(T v, int src, int = 64) { return hostsim::shfl_idx(v, (unsigned)src); }
template <typename T> inline T __shfl_xor(T v, int m, int = 64) { return hostsim::shfl_idx(v, (hostsim::t_tid & 63u) ^ (unsigned)m); }
template <typename T> inline T __shfl_up(T v, unsigned d, int = 64) {
    unsigned lane = hostsim::t_tid & 63u;
    return hostsim::shfl_idx(v, lane >= d ? lane - d : lane);
}
template <typename T> inline T __shfl_down(T v, unsigned d, int = 64) {
    unsigned lane = hostsim::t_tid & 63u;
    return hostsim::shfl_idx(v, lane + d < 64 ? lane + d : lane);
}

static inline int atomicMin(int* p, int v) {
    int old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (old > v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return old;
}
static inline int atomicMax(int* p, int v) {
    int old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (old < v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return old;
}
static inline float unsafeAtomicAdd(float* p, float v) {
    uint32_t o = __atomic_load_n((uint32_t*)p, __ATOMIC_RELAXED);  // CAS loop on the bit pattern
    for (;;) {
        float old;
        memcpy(&old, &o, 4);
        const float want = old + v;
        uint32_t w;
        memcpy(&w, &want, 4);
        if (__atomic_compare_exchange_n((uint32_t*)p, &o, w, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) return old;
    }
}
static inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
static inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }

// agent-scope atomics, wave votes and scalar helpers of the single-pass kernels
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __hip_atomic_store(p, v, order, scope) __atomic_store_n((p), (v), (order))
#define __hip_atomic_load(p, order, scope) __atomic_load_n((p), (order))
static inline int __all(int pred) {
    int v = pred ? 1 : 0;
    for (int m = 32; m >= 1; m >>= 1) v &= hostsim::shfl_idx(v, (hostsim::t_tid & 63u) ^ (unsigned)m);
    return v;
}
static inline int __builtin_amdgcn_readfirstlane(int v) { return hostsim::shfl_idx(v, 0u); }
static inline int __builtin_amdgcn_readlane(int v, int lane) { return hostsim::shfl_idx(v, (unsigned)lane); }
// DPP data movement of one 64-lane wave (the controls the kernels use: row_shr:n = 0x110 + n, row_bcast:15 = 0x142,
// row_bcast:31 = 0x143).  Lanes outside row_mask / bank_mask keep `old`; enabled lanes without a source get 0 when
// bound_ctrl is set and `old` otherwise.
static inline int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
    const unsigned lane = hostsim::t_tid & 63u, row = lane >> 4, in_row = lane & 15u;
    bool valid = false;
    unsigned from = lane;
    if (ctrl > 0x110 && ctrl <= 0x11f) {          // row_shr:n
        const unsigned n = (unsigned)ctrl - 0x110u;
        valid = in_row >= n;
        from = valid ? lane - n : lane;
    } else if (ctrl > 0x100 && ctrl <= 0x10f) {   // row_shl:n
        const unsigned n = (unsigned)ctrl - 0x100u;
        valid = in_row + n < 16u;
        from = valid ? lane + n : lane;
    } else if (ctrl == 0x130) {                   // wave_shl:1
        valid = lane < 63u;
        from = valid ? lane + 1u : lane;
    } else if (ctrl == 0x138) {                   // wave_shr:1
        valid = lane >= 1u;
        from = valid ? lane - 1u : lane;
    } else if (ctrl == 0x142) {
        valid = row >= 1;
        from = valid ? 16u * row - 1u : lane;
    } else if (ctrl == 0x143) {
        valid = row >= 2;
        from = valid ? 31u : lane;
    }
    const int got = hostsim::shfl_idx(src, from);
    const bool enabled = ((row_mask >> row) & 1) && ((bank_mask >> (in_row >> 2)) & 1);
    if (!enabled) return old;
    return valid ? got : (bound_ctrl ? 0 : old);
}
static inline void __builtin_amdgcn_s_sleep(int) { std::this_thread::yield(); }
static inline void __builtin_amdgcn_sched_barrier(int) {}

// raw gfx950 transcendental builtins used by the kernels
// v_mfma_f32_16x16x4_f32: D = A B + C for one wave; lane l supplies A[l & 15][l >> 4] and B[l >> 4][l & 15] and holds
// D[(l >> 4) * 4 + r][l & 15], r < 4; a k-ordered fp32 FMA chain (cdna_hip_programming.md)
typedef float hostsim_f32x4 __attribute__((vector_size(16)));
static inline hostsim_f32x4 __builtin_amdgcn_mfma_f32_16x16x4f32(float a, float b, hostsim_f32x4 c, int, int, int) {
    // one exchange for the whole instruction (a, b packed into the lane's 64-bit slot): two rendezvous instead of 64
    hostsim::BlockCtx* ctx = hostsim::t_ctx;
    const unsigned wave = hostsim::t_tid >> 6, lane = hostsim::t_tid & 63u, base = wave << 6;
    float ab[2] = {a, b};
    uint64_t bits;
    memcpy(&bits, ab, 8);
    ctx->xchg[hostsim::t_tid] = bits;
    ctx->wave_bar[wave]->arrive_and_wait();
    for (unsigned r = 0; r < 4; ++r)
        for (unsigned k = 0; k < 4; ++k) {
            float pa[2], pb[2];
            memcpy(pa, &ctx->xchg[base + k * 16u + (lane >> 4) * 4u + r], 8);
            memcpy(pb, &ctx->xchg[base + k * 16u + (lane & 15u)], 8);
            c[r] = fmaf(pa[0], pb[1], c[r]);
        }
    ctx->wave_bar[wave]->arrive_and_wait();
    return c;
}
static inline float __builtin_amdgcn_logf(float x) { return log2f(x); }
static inline float __builtin_amdgcn_exp2f(float x) { return exp2f(x); }
static inline float __builtin_amdgcn_rcpf(float x) { return 1.0f / x; }
static inline float __builtin_amdgcn_fmed3f(float a, float b, float c) { return fmaxf(fminf(a, b), fminf(fmaxf(a, b), c)); }
static inline float __builtin_amdgcn_sqrtf(float x) { return sqrtf(x); }
static inline float __builtin_amdgcn_rsqf(float x) { return 1.0f / sqrtf(x); }
static inline float __fdividef(float a, float b) { return a / b; }
static inline long long __double_as_longlong(double d) { long long v; memcpy(&v, &d, 8); return v; }
static inline double __longlong_as_double(long long v) { double d; memcpy(&d, &v, 8); return d; }
static inline float __int_as_float(int v) { float f; memcpy(&f, &v, 4); return f; }
static inline int __float_as_int(float f) { int v; memcpy(&v, &f, 4); return v; }
