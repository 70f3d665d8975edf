// dev probe (round 6): the VQ filter's scan loop by itself -- fp16 MFMAs + the min3 / and_or / med3 / min bookkeeping -- in the
// forms that were candidates for the kernel: the shipping 32x32x16 ping-pong (chains / trees / three accumulator sets / one
// vector tile per wave) and 16x16 tiles (v_mfma_f32_16x16x16_f16 and 16x16x32_f16, 4-register accumulators, rotation depth D).
// Every kernel: 256 workgroups, NT threads, one per CU; a wave scans 1024 codes (np tiles) for 64 vectors per repetition.
//   hipcc -O3 --offload-arch=gfx950 probe_scan3.hip -o probe_scan3 && ./probe_scan3
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float vmin3(float a, float b, float c) { return __builtin_fminf(__builtin_fminf(a, b), c); }
__device__ __forceinline__ void book(float &m1, float &m2, float u, unsigned int T)
{
    u = __uint_as_float((__float_as_uint(u) & ~31u) | T);
    m2 = __builtin_amdgcn_fmed3f(m1, m2, u);
    asm("v_min_f32 %0, %1, %2" : "=v"(m1) : "v"(m1), "v"(u));
}

// ---- 32x32x16: ZT vector tiles of 32 per wave (the kernel: ZT = 2); VAR 0 chains, 1 trees, 2 three accumulator sets
template <int ZT, int NT, int VAR, int MODE, int CAP>
__global__ __launch_bounds__(NT, CAP) void scan32(const uint4 *__restrict__ tab, float *out, int reps)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint4 *ldsA = reinterpret_cast<uint4 *>(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 32 * 64; i += NT) ldsA[i] = tab[i];
    __syncthreads();
    constexpr int ntile = 32;
    f16x8 bop[ZT];
    float m1[ZT], m2[ZT];
    for (int t = 0; t < ZT; ++t) { bop[t] = __builtin_bit_cast(f16x8, tab[(lane + t * 64) & 2047]); m1[t] = m2[t] = __builtin_inff(); }
    f32x16 zero16;
    for (int i = 0; i < 16; ++i) zero16[i] = 0.f;
    f32x16 X[ZT], Y[ZT], Z[ZT];
    for (int t = 0; t < ZT; ++t) { X[t] = zero16 + (float)lane; Y[t] = X[t]; Z[t] = X[t]; }
    auto issue = [&](int T, f32x16 (&D)[ZT]) {
        if (MODE == 2) return;
        const int TT = T < ntile ? T : ntile - 1;
        const f16x8 av = __builtin_bit_cast(f16x8, ldsA[TT * 64 + lane]);
#pragma unroll
        for (int t = 0; t < ZT; ++t) D[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bop[t], zero16, 0, 0, 0);
    };
    auto digest = [&](int T, f32x16 (&D)[ZT]) {
        if (MODE == 1) {
#pragma unroll
            for (int t = 0; t < ZT; ++t) for (int i = 0; i < 16; ++i) asm volatile("" :: "v"(D[t][i]));
            return;
        }
#pragma unroll
        for (int t = 0; t < ZT; ++t) {
            float u;
            if (VAR == 1) {
                const float a = vmin3(D[t][0], D[t][1], D[t][2]), b = vmin3(D[t][3], D[t][4], D[t][5]), c = vmin3(D[t][6], D[t][7], D[t][8]);
                const float d = vmin3(D[t][9], D[t][10], D[t][11]), e = vmin3(D[t][12], D[t][13], D[t][14]);
                u = vmin3(vmin3(a, b, c), vmin3(d, e, D[t][15]), __builtin_inff());
            } else {
                u = __builtin_inff();
#pragma unroll
                for (int r = 0; r < 16; r += 2) u = vmin3(u, D[t][r], D[t][r + 1]);
            }
            book(m1[t], m2[t], u, (unsigned int)T);
            if (MODE == 2) D[t][T & 15] += m1[t] * 1e-30f;
        }
    };
    for (int r = 0; r < reps; ++r) {
        if (VAR == 2) {
            issue(0, X); issue(1, Y);
            for (int T = 0; T < ntile - 2; T += 3) {
                issue(T + 2, Z); digest(T, X);
                issue(T + 3, X); digest(T + 1, Y);
                issue(T + 4, Y); digest(T + 2, Z);
            }
            digest(30, X); digest(31, Y);
        } else {
            issue(0, X);
            for (int T = 0; T < ntile; T += 2) {
                issue(T + 1, Y);
                digest(T, X);
                issue(T + 2, X);
                digest(T + 1, Y);
            }
        }
    }
    float acc = 0.f;
    for (int t = 0; t < ZT; ++t) acc += m1[t] + m2[t] + X[t][0] + Y[t][1] + Z[t][2];
    if (acc == 12345.678f) out[tid] = acc;
}

// ---- 16x16 tiles: 4 vector blocks of 16 per wave (64 vectors), a code tile of 16 = 4 MFMAs (one per block), rotation depth D;
// bookkeeping once per 64 codes (4 code tiles).  K16: v_mfma_f32_16x16x16_f16 (legacy rate), else v_mfma_f32_16x16x32_f16.
template <int NT, int D, bool K16, int MODE, int CAP>
__global__ __launch_bounds__(NT, CAP) void scan16(const uint4 *__restrict__ tab, float *out, int reps)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint4 *ldsA = reinterpret_cast<uint4 *>(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 64 * 64; i += NT) ldsA[i] = tab[i & 2047];
    __syncthreads();
    constexpr int ntile = 64;          // code tiles of 16
    f16x8 bop[4];
    float m1[4], m2[4], u[4];
    for (int b = 0; b < 4; ++b) { bop[b] = __builtin_bit_cast(f16x8, tab[(lane + b * 64) & 2047]); m1[b] = m2[b] = u[b] = __builtin_inff(); }
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 S[D][4];
    for (int d = 0; d < D; ++d) for (int b = 0; b < 4; ++b) S[d][b] = zero4 + (float)(lane + d);
    auto issue = [&](int T, f32x4 (&A)[4]) {
        if (MODE == 2) return;
        const int TT = T < ntile ? T : ntile - 1;
        if (K16) {
            const uint2 raw = reinterpret_cast<const uint2 *>(ldsA)[TT * 64 + lane];
            const f16x4 av = __builtin_bit_cast(f16x4, raw);
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const f16x4 bv = {bop[b][0], bop[b][1], bop[b][2], bop[b][3]};
                A[b] = __builtin_amdgcn_mfma_f32_16x16x16f16(av, bv, zero4, 0, 0, 0);
            }
        } else {
            const f16x8 av = __builtin_bit_cast(f16x8, ldsA[TT * 64 + lane]);
#pragma unroll
            for (int b = 0; b < 4; ++b) A[b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bop[b], zero4, 0, 0, 0);
        }
    };
    auto digest = [&](int T, f32x4 (&A)[4]) {
        if (MODE == 1) {
#pragma unroll
            for (int b = 0; b < 4; ++b) for (int i = 0; i < 4; ++i) asm volatile("" :: "v"(A[b][i]));
            return;
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            u[b] = vmin3(u[b], A[b][0], A[b][1]);
            u[b] = vmin3(u[b], A[b][2], A[b][3]);
            if ((T & 3) == 3) { book(m1[b], m2[b], u[b], (unsigned int)(T >> 2)); u[b] = __builtin_inff(); }
            if (MODE == 2) A[b][T & 3] += m1[b] * 1e-30f;
        }
    };
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int d = 0; d < D - 1; ++d) issue(d, S[d]);
        for (int T = 0; T < ntile; T += D) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                issue(T + d + D - 1, S[(d + D - 1) % D]);
                digest(T + d, S[d]);
            }
        }
    }
    float acc = 0.f;
    for (int b = 0; b < 4; ++b) acc += m1[b] + m2[b] + u[b];
    for (int d = 0; d < D; ++d) acc += S[d][0][0] + S[d][3][1];
    if (acc == 12345.678f) out[tid] = acc;
}

template <typename F>
static float timeit(F launch)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    launch(); hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < 5; ++i) launch();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1e3f / 5;
}

int main()
{
    uint4 *tab; float *out;
    hipMalloc(&tab, 4096 * 16); hipMalloc(&out, 4096 * 4);
    {
        static uint32_t h[4096 * 4];
        for (int i = 0; i < 4096 * 4; ++i) { uint32_t x = 0x3C003C00u ^ ((i * 2654435761u) & 0x03FF03FFu); h[i] = x; }
        hipMemcpy(tab, h, sizeof(h), hipMemcpyHostToDevice);
    }
    const int reps = 64;
    // per repetition a wave covers 64 vectors x 1024 codes = 16 "iterations" of 64 codes; a SIMD hosts NT/256 waves
#define REPORT(name, us, NT, vecs)                                                                                                      \
    printf("%-46s %d waves/SIMD: %8.1f us per launch, %7.1f ns per 64x64 iteration per wave, %6.1f ns per SIMD\n", name, NT / 256, us,                           \
           us * 1e3 / (reps * 16.0) * (64.0 / vecs), us * 1e3 / (reps * 16.0) * (64.0 / vecs) / (NT / 256));
#define RUN32(ZT, NT, VAR, MODE, CAP, name)                                                                                            \
    {                                                                                                                                   \
        hipFuncSetAttribute((const void *)scan32<ZT, NT, VAR, MODE, CAP>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);          \
        float us = timeit([&] { hipLaunchKernelGGL((scan32<ZT, NT, VAR, MODE, CAP>), dim3(256), dim3(NT), 65536, 0, tab, out, reps); }); \
        REPORT(name, us, NT, 32 * ZT)                                                                                                   \
    }
#define RUN16(NT, D, K16, MODE, CAP, name)                                                                                             \
    {                                                                                                                                   \
        hipFuncSetAttribute((const void *)scan16<NT, D, K16, MODE, CAP>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);           \
        float us = timeit([&] { hipLaunchKernelGGL((scan16<NT, D, K16, MODE, CAP>), dim3(256), dim3(NT), 65536, 0, tab, out, reps); }); \
        REPORT(name, us, NT, 64)                                                                                                        \
    }
    printf("---- forward\n");
    RUN32(2, 512, 0, 0, 4, "32x32x16 ZT2 chains (shipping), cap 128")
    RUN32(2, 512, 0, 1, 4, "32x32x16 ZT2 MFMA only")
    RUN32(2, 512, 0, 2, 4, "32x32x16 ZT2 VALU only")
    RUN32(2, 512, 1, 0, 4, "32x32x16 ZT2 trees, cap 128")
    RUN32(2, 512, 2, 0, 2, "32x32x16 ZT2 three sets, cap 256")
    RUN32(1, 512, 0, 0, 4, "32x32x16 ZT1 chains")
    RUN32(1, 512, 2, 0, 4, "32x32x16 ZT1 three sets")
    RUN32(1, 768, 2, 0, 6, "32x32x16 ZT1 three sets, cap 80")
    RUN32(1, 1024, 0, 0, 4, "32x32x16 ZT1 chains")
    RUN32(2, 768, 0, 0, 6, "32x32x16 ZT2 chains cap 80 (spills?)")
    RUN16(512, 2, true, 0, 4, "16x16x16 D2")
    RUN16(512, 4, true, 0, 4, "16x16x16 D4")
    RUN16(512, 4, true, 1, 4, "16x16x16 D4 MFMA only")
    RUN16(512, 4, true, 2, 4, "16x16x16 D4 VALU only")
    RUN16(512, 8, true, 0, 4, "16x16x16 D8")
    RUN16(768, 4, true, 0, 6, "16x16x16 D4 cap 80")
    RUN16(768, 8, true, 0, 6, "16x16x16 D8 cap 80")
    RUN16(1024, 4, true, 0, 4, "16x16x16 D4")
    RUN16(512, 4, false, 0, 4, "16x16x32 D4")
    RUN16(512, 4, false, 1, 4, "16x16x32 D4 MFMA only")
    RUN16(768, 4, false, 0, 6, "16x16x32 D4 cap 80")
    printf("---- reversed\n");
    RUN16(768, 4, false, 0, 6, "16x16x32 D4 cap 80")
    RUN16(512, 4, false, 1, 4, "16x16x32 D4 MFMA only")
    RUN16(512, 4, false, 0, 4, "16x16x32 D4")
    RUN16(1024, 4, true, 0, 4, "16x16x16 D4")
    RUN16(768, 8, true, 0, 6, "16x16x16 D8 cap 80")
    RUN16(768, 4, true, 0, 6, "16x16x16 D4 cap 80")
    RUN16(512, 8, true, 0, 4, "16x16x16 D8")
    RUN16(512, 4, true, 2, 4, "16x16x16 D4 VALU only")
    RUN16(512, 4, true, 1, 4, "16x16x16 D4 MFMA only")
    RUN16(512, 4, true, 0, 4, "16x16x16 D4")
    RUN16(512, 2, true, 0, 4, "16x16x16 D2")
    RUN32(2, 768, 0, 0, 6, "32x32x16 ZT2 chains cap 80 (spills?)")
    RUN32(1, 1024, 0, 0, 4, "32x32x16 ZT1 chains")
    RUN32(1, 768, 2, 0, 6, "32x32x16 ZT1 three sets, cap 80")
    RUN32(1, 512, 2, 0, 4, "32x32x16 ZT1 three sets")
    RUN32(1, 512, 0, 0, 4, "32x32x16 ZT1 chains")
    RUN32(2, 512, 2, 0, 2, "32x32x16 ZT2 three sets, cap 256")
    RUN32(2, 512, 1, 0, 4, "32x32x16 ZT2 trees, cap 128")
    RUN32(2, 512, 0, 2, 4, "32x32x16 ZT2 VALU only")
    RUN32(2, 512, 0, 1, 4, "32x32x16 ZT2 MFMA only")
    RUN32(2, 512, 0, 0, 4, "32x32x16 ZT2 chains (shipping), cap 128")
    return 0;
}
