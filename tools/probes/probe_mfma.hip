// dev probe: issue rate of the bf16 / f32 MFMA shapes on gfx950, 1/2/4 waves per SIMD, independent accumulators
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
template <int OP>
__global__ void k(float *out, int iters)
{
    bf16x8 a8, b8; s16x4 a4, b4; float af = threadIdx.x, bf = 1.f;
    for (int i = 0; i < 8; ++i) { a8[i] = (__bf16)(float)(threadIdx.x + i); b8[i] = (__bf16)(float)(i); }
    for (int i = 0; i < 4; ++i) { a4[i] = (short)(threadIdx.x + i); b4[i] = (short)i; }
    f32x4 acc[8]; f32x16 big[2];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
    for (int i = 0; i < 2; ++i) for (int q = 0; q < 16; ++q) big[i][q] = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, acc[i], 0, 0, 0);
            if (OP == 1) acc[i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, acc[i], 0, 0, 0);
            if (OP == 2) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf, acc[i], 0, 0, 0);
            if (OP == 3 && i < 2) big[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, big[i], 0, 0, 0);
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
    s += big[0][0] + big[1][5];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP>
void run(const char *name, float *out, int w, int per_iter)
{
    const int iters = 2000, nthreads = 256 * w;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(nthreads), 0, 0, out, iters);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(nthreads), 0, 0, out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s %d waves/SIMD: %.2f ns per MFMA per SIMD\n", name, w, ms * 1e6 / ((double)iters * per_iter * w));
}
int main()
{
    float *out; (void)hipMalloc(&out, 256 * 1024 * 4);
    for (int w : {1, 2, 4}) {
        run<0>("16x16x32 bf16", out, w, 8); run<1>("16x16x16 bf16 (1k)", out, w, 8); run<2>("16x16x4 f32", out, w, 8); run<3>("32x32x16 bf16", out, w, 2);
    }
    return 0;
}
