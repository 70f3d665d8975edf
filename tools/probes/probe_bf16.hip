// dev probe: does bf16 MFMA overlap with VALU work on gfx950?  (not part of the product)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));

template <int MODE, int NV>   // MODE 0 both, 1 mfma only, 2 valu only; NV = VALU ops per MFMA
__global__ __launch_bounds__(256, 2) void k(float *out, int iters)
{
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (short)(threadIdx.x + i); b[i] = (short)(threadIdx.x * 3 + i); }
    float best[4] = {1e30f, 1e30f, 1e30f, 1e30f};
    float v = threadIdx.x * 0.001f;
    f32x4 keep = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            if (MODE != 2) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
            else { acc[0] = v; acc[1] = v + 1; acc[2] = v + 2; acc[3] = v + 3; }
            if (MODE == 1) { asm volatile("" :: "v"(acc)); keep = acc; continue; }
            // NV dependent-ish VALU ops on the result
            float m = __builtin_fminf(__builtin_fminf(best[t], acc[0]), acc[1]);
            m = __builtin_fminf(__builtin_fminf(m, acc[2]), acc[3]);
#pragma unroll
            for (int q = 0; q < NV - 2; ++q) m = __builtin_fmaf(m, 1.0000001f, v);
            best[t] = m;
            a[t & 7] += 1;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = best[0] + best[1] + best[2] + best[3] + keep[0];
}

template <int MODE, int NV>
void run(float *out, const char *name)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int nblk = 1024, iters = 64;        // 1024 blocks x 4 waves x 64 iters x 4 MFMA = same tile count as the VQ kernel at B=64
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<MODE, NV>), dim3(nblk), dim3(256), 0, 0, out, iters);
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k<MODE, NV>), dim3(nblk), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-40s %8.2f us\n", name, ms * 1e3 / 20);
}

int main()
{
    float *out; hipMalloc(&out, 1024 * 256 * 4);
    run<1, 8>(out, "bf16 16x16x32 MFMA only");
    run<2, 8>(out, "VALU only, 8 ops per tile");
    run<0, 8>(out, "MFMA + 8 VALU per tile");
    run<2, 12>(out, "VALU only, 12 ops per tile");
    run<0, 12>(out, "MFMA + 12 VALU per tile");
    run<2, 4>(out, "VALU only, 4 ops per tile");
    run<0, 4>(out, "MFMA + 4 VALU per tile");
    return 0;
}
