// Developer micro-benchmark: issue cost of the cross-lane VALU forms on gfx950 (cycles per wave-instruction per SIMD with W waves
// per SIMD): v_mov_b32_dpp (quad_perm / row_shr + bank mask), DPP folded into v_add_f32, v_permlane32_swap, v_permlane16_swap,
// v_cndmask, ds_bpermute, ds_swizzle, plain v_fma.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/ubench/bin/xlane_rate tools/ubench/xlane_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>

template <int OP>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = (float)(threadIdx.x * 16 + i) * 1e-3f;
    const bool hi = threadIdx.x & 1;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (OP == 0) v[i] = fmaf(v[i], 0.999f, 0.001f);
            if (OP == 1) v[i] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v[i]), 0xb1, 0xf, 0xf, true));
            if (OP == 2) v[i] = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v[i]), __float_as_int(v[(i + 1) & 15]), 0x114, 0xf, 0xa, false));
            if (OP == 3) v[i] = v[i] + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v[i]), 0xb1, 0xf, 0xf, true));
            if (OP == 4 && !(i & 1)) {
                const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[i]), __float_as_uint(v[i + 1]), false, false);
                v[i] = __uint_as_float(r[0]); v[i + 1] = __uint_as_float(r[1]);
            }
            if (OP == 5 && !(i & 1)) {
                const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[i]), __float_as_uint(v[i + 1]), false, false);
                v[i] = __uint_as_float(r[0]); v[i + 1] = __uint_as_float(r[1]);
            }
            if (OP == 6) v[i] = hi ? v[(i + 1) & 15] : v[i];
            if (OP == 7) v[i] = __int_as_float(__builtin_amdgcn_ds_bpermute((threadIdx.x ^ 5) << 2, __float_as_int(v[i])));
            if (OP == 8) v[i] = __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v[i]), 0x041f));  // xor 1 within 32
            if (OP == 9) v[i] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v[i]), 0x128, 0xf, 0xf, true));  // row_ror:8
            if (OP == 10) v[i] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v[i]), 0x141, 0xf, 0xf, true));  // row_half_mirror
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(v[i]));
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int OP>
static void bench(const char* name, float* out, int per_iter) {
    const int iters = 16384;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int wps : {1, 2, 4}) {
        float best = 1e9f;
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(k<OP>, dim3(256 * wps), dim3(256), 0, 0, out, iters);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best;
        }
        printf("%-34s %d waves/SIMD: %.2f cycles per wave-instruction per SIMD\n", name, wps, best * 1e-3 * 2.4e9 / ((double)iters * per_iter * wps));
    }
}

int main() {
    float* out;
    (void)hipMalloc(&out, 1024 * 8 * 64 * 4);
    bench<0>("v_fma_f32", out, 16);
    bench<1>("v_mov_dpp quad_perm", out, 16);
    bench<9>("v_mov_dpp row_ror:8", out, 16);
    bench<10>("v_mov_dpp row_half_mirror", out, 16);
    bench<2>("v_mov_dpp row_shr:4 bank_mask", out, 16);
    bench<3>("v_add_dpp quad_perm (or mov+add)", out, 16);
    bench<4>("v_permlane32_swap", out, 8);
    bench<5>("v_permlane16_swap", out, 8);
    bench<6>("v_cndmask", out, 16);
    bench<7>("ds_bpermute_b32", out, 16);
    bench<8>("ds_swizzle_b32", out, 16);
    return 0;
}
