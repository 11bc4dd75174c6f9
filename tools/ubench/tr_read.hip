// ds_read_b64_tr_b16 semantics probe (gfx950): hipcc --offload-arch=gfx950 -O2 tools/ubench/tr_read.hip -o /tmp/tr_read && /tmp/tr_read
// Hypothesis: in every 16-lane group, lane t supplies the address of an 8-byte piece = row t >> 2, columns 4 (t & 3) .. +3 of a
// 4 x 16 matrix of 16-bit elements; the result of lane i is column i of that matrix (rows 0..3).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
__global__ void k(uint16_t* out, int row_stride) {
    __shared__ __attribute__((aligned(16))) uint16_t s[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) s[i] = (uint16_t)i;
    __syncthreads();
    const int l = threadIdx.x, grp = l >> 4, t = l & 15;
    // group g reads rows 4g .. 4g+3 of a [16][row_stride] image, columns 0..15
    const uint32_t addr = (uint32_t)(uintptr_t)s + 2u * (uint32_t)((4 * grp + (t >> 2)) * row_stride + 4 * (t & 3));
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr));
    out[l * 4 + 0] = v[0] & 0xffff; out[l * 4 + 1] = v[0] >> 16; out[l * 4 + 2] = v[1] & 0xffff; out[l * 4 + 3] = v[1] >> 16;
}
int main() {
    uint16_t* d; hipMalloc(&d, 512);
    int bad = 0;
    for (int stride : {16, 24, 128}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, stride);
        uint16_t h[256]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
        for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 4; ++j) {
                const int expect = (4 * (l >> 4) + j) * stride + (l & 15);
                if (h[l * 4 + j] != expect) { if (bad < 8) printf("stride %d lane %d elem %d: got %d expect %d\n", stride, l, j, h[l * 4 + j], expect); ++bad; }
            }
    }
    printf(bad ? "tr_read: MISMATCH (%d)\n" : "tr_read: hypothesis holds (%d mismatches)\n", bad);
    return bad != 0;
}
