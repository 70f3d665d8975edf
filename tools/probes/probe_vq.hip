// dev probe: ablate the VQ inner loop (not part of the product)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

typedef float f32x2 __attribute__((ext_vector_type(2)));
__global__ void clk(long long *o, int iters)
{
    f32x4 acc = {0, 0, 0, 0};
    long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, 2.0f, acc, 0, 0, 0);
    long long c1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) { o[0] = c1 - c0; o[1] = w1 - w0; o[2] = (long long)acc[0]; }
}

template <int ZT, int MODE, int WPE>   // MODE 0 = mfma+epilogue, 1 = mfma only, 2 = epilogue only, 3 = mfma + packed epilogue, 4 = pure VALU
__global__ __launch_bounds__(256, WPE) void k(const float *z, const float *cb, int K, float *out)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *cbT = smem, *ee = smem + 4 * K;
    for (int i = threadIdx.x; i < K; i += 256) {
        float4 e = ((const float4 *)cb)[i];
        cbT[i] = e.x; cbT[K + i] = e.y; cbT[2 * K + i] = e.z; cbT[3 * K + i] = e.w;
        ee[i] = e.x * e.x + e.y * e.y;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
    float zv[ZT], zz[ZT], best[ZT]; int bt[ZT];
    for (int t = 0; t < ZT; ++t) { zv[t] = z[(blockIdx.x * 256 + threadIdx.x) * ZT + t]; zz[t] = zv[t] * zv[t]; best[t] = 1e30f; bt[t] = 0; }
    const int ntile = K >> 4;
    f32x4 keep = {0, 0, 0, 0};
    for (int ct = 0; ct < ntile; ++ct) {
        const float a = cbT[g * K + 16 * ct + j];
        const f32x4 e4 = *(const f32x4 *)&ee[16 * ct + 4 * g];
#pragma unroll
        for (int t = 0; t < ZT; ++t) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            if (MODE != 2) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, zv[t], acc, 0, 0, 0);
            else { acc[0] = a; acc[1] = zv[t]; acc[2] = a + 1.f; acc[3] = zv[t] + 1.f; }
            if (MODE == 1) { asm volatile("" :: "v"(acc)); keep = acc; continue; }
            if (MODE == 4) {
                // pure VALU contraction for this lane's (z, 4 codes): mul + 3 fma each
                const float4 c0 = ((const float4 *)cbT)[(16 * ct + 4 * g) & 1023];
                acc[0] = __builtin_fmaf(zv[t], c0.w, __builtin_fmaf(zv[t], c0.z, __builtin_fmaf(zz[t], c0.y, zv[t] * c0.x)));
                acc[1] = __builtin_fmaf(zv[t], c0.x, __builtin_fmaf(zv[t], c0.w, __builtin_fmaf(zz[t], c0.z, zv[t] * c0.y)));
                acc[2] = __builtin_fmaf(zv[t], c0.y, __builtin_fmaf(zv[t], c0.x, __builtin_fmaf(zz[t], c0.w, zv[t] * c0.z)));
                acc[3] = __builtin_fmaf(zv[t], c0.z, __builtin_fmaf(zv[t], c0.y, __builtin_fmaf(zz[t], c0.x, zv[t] * c0.w)));
            }
            if (MODE == 3) {
                f32x2 zz2 = {zz[t], zz[t]}, m2c = {-2.0f, -2.0f};
                f32x2 s01 = zz2 + f32x2{e4[0], e4[1]}, s23 = zz2 + f32x2{e4[2], e4[3]};
                f32x2 d01 = __builtin_elementwise_fma(m2c, f32x2{acc[0], acc[1]}, s01);
                f32x2 d23 = __builtin_elementwise_fma(m2c, f32x2{acc[2], acc[3]}, s23);
                const float m1 = __builtin_fminf(__builtin_fminf(best[t], d01[0]), d01[1]);
                const float m2 = __builtin_fminf(__builtin_fminf(m1, d23[0]), d23[1]);
                bt[t] = m2 < best[t] ? ct : bt[t];
                best[t] = m2;
                continue;
            }
            const float d0 = __builtin_fmaf(-2.0f, acc[0], zz[t] + e4[0]);
            const float d1 = __builtin_fmaf(-2.0f, acc[1], zz[t] + e4[1]);
            const float d2 = __builtin_fmaf(-2.0f, acc[2], zz[t] + e4[2]);
            const float d3 = __builtin_fmaf(-2.0f, acc[3], zz[t] + e4[3]);
            const float m1 = __builtin_fminf(__builtin_fminf(best[t], d0), d1);
            const float m2 = __builtin_fminf(__builtin_fminf(m1, d2), d3);
            bt[t] = m2 < best[t] ? ct : bt[t];
            best[t] = m2;
        }
    }
    float r = keep[0];
    for (int t = 0; t < ZT; ++t) r += best[t] + bt[t];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int ZT, int MODE, int WPE>
float run(const float *z, const float *cb, float *out, int nblk, const char *name)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    size_t lds = 1024 * 5 * 4;
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<ZT, MODE, WPE>), dim3(nblk), dim3(256), lds, 0, z, cb, 1024, out);
    hipEventRecord(a);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k<ZT, MODE, WPE>), dim3(nblk), dim3(256), lds, 0, z, cb, 1024, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double tiles = (double)nblk * 4 * ZT * 64;   // MFMA tiles
    double us = ms * 1e3 / 20;
    printf("%-34s nblk %5d  %8.2f us   %6.1f ns/tile/SIMD-equivalent cycles@2.4GHz: %6.1f\n", name, nblk, us,
           us * 1e3 / (tiles / 1024), us * 1e3 / (tiles / 1024) * 2.4);
    return us;
}

int main()
{
    const int N = 262144;
    float *z, *cb, *out;
    hipMalloc(&z, N * 8 * 4); hipMalloc(&cb, 1024 * 16); hipMalloc(&out, (size_t)N * 4 * 4);
    float *h = (float *)malloc(N * 8 * 4);
    for (int i = 0; i < N * 8; ++i) h[i] = (float)rand() / RAND_MAX - 0.5f;
    hipMemcpy(z, h, N * 8 * 4, hipMemcpyHostToDevice); hipMemcpy(cb, h, 1024 * 16, hipMemcpyHostToDevice);
    {
        long long *o; hipMalloc(&o, 64); long long h3[3];
        hipLaunchKernelGGL(clk, dim3(1024), dim3(256), 0, 0, o, 20000);
        hipLaunchKernelGGL(clk, dim3(1024), dim3(256), 0, 0, o, 20000);
        hipMemcpy(h3, o, 24, hipMemcpyDeviceToHost);
        printf("shader clock: %.3f GHz (clock64 %lld / wall_clock64 %lld @100MHz); cycles per 16x16x4 f32 MFMA (1 wave/SIMD): %.1f\n",
               (double)h3[0] / ((double)h3[1] / 100e6) / 1e9, h3[0], h3[1], (double)h3[0] / 20000);
        hipLaunchKernelGGL(clk, dim3(256 * 8), dim3(256), 0, 0, o, 20000);
        hipMemcpy(h3, o, 24, hipMemcpyDeviceToHost);
        printf("  with 8 waves/SIMD-ish: %.3f GHz, cycles per MFMA per wave %.1f\n", (double)h3[0] / ((double)h3[1] / 100e6) / 1e9, (double)h3[0] / 20000);
    }
    // same total work (N vectors x 1024 codes) in every row
    run<8, 0, 2>(z, cb, out, N / 512, "ZT8 mfma+epi wpe2");
    run<8, 1, 2>(z, cb, out, N / 512, "ZT8 mfma only");
    run<8, 2, 2>(z, cb, out, N / 512, "ZT8 epilogue only");
    run<4, 0, 2>(z, cb, out, N / 256, "ZT4 mfma+epi wpe2");
    run<4, 1, 2>(z, cb, out, N / 256, "ZT4 mfma only");
    run<4, 2, 2>(z, cb, out, N / 256, "ZT4 epilogue only");
    run<2, 0, 2>(z, cb, out, N / 128, "ZT2 mfma+epi");
    run<2, 0, 4>(z, cb, out, N / 128, "ZT2 mfma+epi wpe4");
    run<1, 0, 4>(z, cb, out, N / 64, "ZT1 mfma+epi wpe4");
    run<4, 0, 4>(z, cb, out, N / 256, "ZT4 mfma+epi wpe4");
    run<8, 0, 1>(z, cb, out, N / 512, "ZT8 mfma+epi wpe1 (AGPR form)");
    run<4, 3, 2>(z, cb, out, N / 256, "ZT4 mfma + PACKED epilogue");
    run<8, 3, 2>(z, cb, out, N / 512, "ZT8 mfma + PACKED epilogue");
    run<4, 4, 2>(z, cb, out, N / 256, "ZT4 pure VALU contraction + epi");
    return 0;
}
