// Probe of ds_read_b64_tr_b16 semantics on gfx950 (used to design sq_conv_wgrad_bf16.hip):
// LDS holds a [64 rows][16 cols] int16 image with value row*100+col (32-byte rows).  Lane group g
// reads the 4x16 block of rows 4g..4g+3: lane 4q+p supplies the address of (row 4g+q, cols 4p..4p+3).
// Expected: lane i of the group receives column i of the 4 rows (element q = row 4g+q).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(short* out) {
    __shared__ __attribute__((aligned(16))) short lds[64 * 16];
    for (int i = threadIdx.x; i < 64 * 16; i += 64) lds[i] = (short)((i / 16) * 100 + i % 16);
    __syncthreads();
    const int lane = threadIdx.x, g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    auto ptr = (__attribute__((address_space(3))) s16x4*)(&lds[(4 * g + q) * 16 + 4 * p]);
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(ptr);
    for (int j = 0; j < 4; ++j) out[lane * 4 + j] = v[j];
}
int main() {
    short* d; hipMalloc(&d, 64 * 4 * 2);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane) for (int j = 0; j < 4; ++j) {
        int g = lane >> 4, i = lane & 15;
        int expect = (4 * g + j) * 100 + i;
        if (h[lane * 4 + j] != expect) { if (bad < 8) printf("lane %d elem %d: got %d expect %d\n", lane, j, h[lane*4+j], expect); ++bad; }
    }
    printf("tr16 probe: %d mismatches\n", bad);
    return bad != 0;
}
