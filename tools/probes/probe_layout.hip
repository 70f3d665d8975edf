// dev probe: operand / result layout of v_mfma_f32_32x32x16_f16 on gfx950 (A[i][k], B[k][j], D[i][j])
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(const float *A, const float *B, float *D)
{
    const int l = threadIdx.x;
    f16x8 a, b;
    for (int v = 0; v < 8; ++v) {
        a[v] = (_Float16)A[(l & 31) * 16 + 8 * (l >> 5) + v];      // A[i = l&31][k = 8*(l>>5)+v]
        b[v] = (_Float16)B[(8 * (l >> 5) + v) * 32 + (l & 31)];    // B[k][j = l&31]
    }
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
        D[row * 32 + col] = c[r];
    }
}
int main()
{
    float hA[32 * 16], hB[16 * 32], hD[32 * 32], *dA, *dB, *dD;
    for (int i = 0; i < 512; ++i) { hA[i] = (float)((i * 7 + 3) % 11 - 5); hB[i] = (float)((i * 5 + 1) % 13 - 6); }
    hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dD, sizeof(hD));
    hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(hD, dD, sizeof(hD), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
        float s = 0; for (int kk = 0; kk < 16; ++kk) s += hA[i * 16 + kk] * hB[kk * 32 + j];
        if (s != hD[i * 32 + j]) ++bad;
    }
    printf("32x32x16 f16 layout check: %d mismatches of 1024\n", bad);
    return 0;
}
