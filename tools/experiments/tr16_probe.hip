// What does ds_read_b64_tr_b16 deliver?  LDS image [32 rows][64 cols] of fp16, value = 100 row + col.  Group g of 16 lanes
// addresses the 4 x 16 block at rows 4g .. 4g+3, columns 16 .. 31: lane 4q+p of the group gives the address of row 4g+q,
// columns 16 + 4p .. +3.  Prints what every lane receives.   hipcc --offload-arch=gfx950 tr16_probe.hip -o build/tr16_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __fp16 fh4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
__global__ void k(float *o)
{
    __shared__ __attribute__((aligned(16))) __fp16 img[32][64];
    for (int i = threadIdx.x; i < 32 * 64; i += 64)
        img[i / 64][i % 64] = (__fp16)(float)(100 * (i / 64) + (i % 64));
    __syncthreads();
    const int l = threadIdx.x, g = l >> 4, q = (l & 15) >> 2, p = l & 3;
    fh4 v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fh4 *)&img[4 * g + q][16 + 4 * p]);
    for (int e = 0; e < 4; ++e)
        o[l * 4 + e] = (float)v[e];
}
int main()
{
    float *d, h[256];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l)
        printf("lane %2d: %6.0f %6.0f %6.0f %6.0f\n", l, h[4 * l], h[4 * l + 1], h[4 * l + 2], h[4 * l + 3]);
    return 0;
}
