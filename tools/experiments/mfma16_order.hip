// Does a chain of v_mfma_f32_16x16x4_f32 (k ascending) reproduce the sequential fmaf chain bit for bit, as the
// 32x32x2 form does (the property the exact kernel's parity with the oracle rests on)?
// Build and run on the GPU box: hipcc --offload-arch=gfx950 -O2 mfma16_order.hip -o /tmp/m16 && /tmp/m16
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void k16(const float *A, const float *B, float *C, int K)
{
    // A [16][K] row-major, B [16][K] row-major (column n of the MFMA's B = row n here), C [16][16]
    const int lane = threadIdx.x, i = lane & 15, kq = lane >> 4;
    f32x4 acc = {0, 0, 0, 0};
    for (int s = 0; s < K / 4; ++s)
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[i * K + 4 * s + kq], B[i * K + 4 * s + kq], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r)
        C[(4 * kq + r) * 16 + i] = acc[r]; // row = 4 (lane>>4) + r, col = lane & 15
}

int main()
{
    const int K = 256;
    std::vector<float> A(16 * K), B(16 * K), C(256);
    srand(1);
    for (auto &x : A) x = (rand() / (float)RAND_MAX - 0.5f) * 0.2f;
    for (auto &x : B) x = (rand() / (float)RAND_MAX - 0.5f) * 0.2f;
    float *dA, *dB, *dC;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 1024);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k16, dim3(1), dim3(64), 0, 0, dA, dB, dC, K);
    hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost);
    int diff = 0;
    double maxd = 0;
    for (int m = 0; m < 16; ++m)
        for (int n = 0; n < 16; ++n) {
            float acc = 0.0f;
            for (int k = 0; k < K; ++k)
                acc = fmaf(A[m * K + k], B[n * K + k], acc);
            if (acc != C[m * 16 + n]) { ++diff; maxd = fmax(maxd, fabs((double)acc - C[m * 16 + n])); }
        }
    printf("16x16x4 chain vs sequential fmaf chain: %d of 256 differ, max |diff| %.3g\n", diff, maxd);
    return 0;
}
