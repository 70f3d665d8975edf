// dev probe (round 2): issue-rate ceiling of the VQ filter scan loop on one SIMD -- 16 bf16 MFMAs + the min3 / and_or / med3 / min
// bookkeeping per quad iteration -- by waves per SIMD and by what is left in the loop.
//   hipcc -O3 --offload-arch=gfx950 probe_scan.hip -o probe_scan && ./probe_scan
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// MODE: 0 = MFMA + VALU (the real loop), 1 = MFMA only, 2 = VALU only
template <int ZT, int MODE, int NT, int VAR = 0>
__global__ __launch_bounds__(NT, NT / 256) void scan16(const uint4 *__restrict__ tab, float *out, int reps, int np)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint4 *ldsA = reinterpret_cast<uint4 *>(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 64 * 64; i += NT) ldsA[i] = tab[i];
    __syncthreads();
    bf16x8 bop[ZT];
    float m1[ZT], m2[ZT], uq[ZT];
    for (int t = 0; t < ZT; ++t) {
        uint4 b = tab[(lane + t * 64) & 4095];
        bop[t] = __builtin_bit_cast(bf16x8, b);
        m1[t] = __builtin_inff(); m2[t] = __builtin_inff();
    }
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 X0[ZT], X1[ZT], Y0[ZT], Y1[ZT];
    for (int t = 0; t < ZT; ++t) { X0[t] = zero4 + (float)lane; X1[t] = zero4 + (float)t; Y0[t] = X0[t]; Y1[t] = X1[t]; }
    auto issue = [&](int p, f32x4 (&A0)[ZT], f32x4 (&A1)[ZT]) {
        if (MODE == 2) return;
        const int pp = p < np ? p : np - 1;
        const bf16x8 a0 = __builtin_bit_cast(bf16x8, ldsA[(2 * pp) * 64 + lane]);
        const bf16x8 a1 = __builtin_bit_cast(bf16x8, ldsA[(2 * pp + 1) * 64 + lane]);
#pragma unroll
        for (int t = 0; t < ZT; ++t) {
            A0[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, bop[t], zero4, 0, 0, 0);
            A1[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, bop[t], zero4, 0, 0, 0);
        }
    };
    auto chain = [&](float u, const f32x4 &A0, const f32x4 &A1) -> float {
        u = __builtin_fminf(__builtin_fminf(u, A0[0]), A0[1]);
        u = __builtin_fminf(__builtin_fminf(u, A0[2]), A0[3]);
        u = __builtin_fminf(__builtin_fminf(u, A1[0]), A1[1]);
        return __builtin_fminf(__builtin_fminf(u, A1[2]), A1[3]);
    };
    auto fetch = [&](int p, bf16x8 &a0, bf16x8 &a1) {
        const int pp = p < np ? p : np - 1;
        a0 = __builtin_bit_cast(bf16x8, ldsA[(2 * pp) * 64 + lane]);
        a1 = __builtin_bit_cast(bf16x8, ldsA[(2 * pp + 1) * 64 + lane]);
    };
    auto mm = [&](const bf16x8 &a0, const bf16x8 &a1, f32x4 (&A0)[ZT], f32x4 (&A1)[ZT]) {
#pragma unroll
        for (int t = 0; t < ZT; ++t) {
            A0[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, bop[t], zero4, 0, 0, 0);
            A1[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, bop[t], zero4, 0, 0, 0);
        }
    };
    if (VAR == 0 || VAR == 3) {
    for (int r = 0; r < reps; ++r) {
        issue(0, X0, X1);
        for (int p = 0; p < np; p += 2) {
            if (VAR == 3) __builtin_amdgcn_s_setprio(1);
            issue(p + 1, Y0, Y1);
            if (VAR == 3) __builtin_amdgcn_s_setprio(0);
#pragma unroll
            for (int t = 0; t < ZT; ++t) uq[t] = chain(__builtin_inff(), X0[t], X1[t]);
            if (VAR == 3) __builtin_amdgcn_s_setprio(1);
            issue(p + 2, X0, X1);
            if (VAR == 3) __builtin_amdgcn_s_setprio(0);
#pragma unroll
            for (int t = 0; t < ZT; ++t) {
                const float u = __uint_as_float((__float_as_uint(chain(uq[t], Y0[t], Y1[t])) & ~15u) | (unsigned int)(p >> 1));
                m2[t] = __builtin_amdgcn_fmed3f(m1[t], m2[t], u);
                asm("v_min_f32 %0, %1, %2" : "=v"(m1[t]) : "v"(m1[t]), "v"(u));
            }
        }
    }
    } else {
    for (int r = 0; r < reps; ++r) {
        bf16x8 ax0, ax1, ay0, ay1;
        fetch(0, ax0, ax1); fetch(1, ay0, ay1);
        mm(ax0, ax1, X0, X1);
        for (int p = 0; p < np; p += 2) {
            fetch(p + 2, ax0, ax1);
            __builtin_amdgcn_sched_barrier(0);
            mm(ay0, ay1, Y0, Y1);
#pragma unroll
            for (int t = 0; t < ZT; ++t) uq[t] = chain(__builtin_inff(), X0[t], X1[t]);
            if (VAR == 2) {
                for (int k = 0; k < 2 * ZT; ++k) { __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x2, 2, 0); }
            }
            fetch(p + 3, ay0, ay1);
            __builtin_amdgcn_sched_barrier(0);
            mm(ax0, ax1, X0, X1);
#pragma unroll
            for (int t = 0; t < ZT; ++t) {
                const float u = __uint_as_float((__float_as_uint(chain(uq[t], Y0[t], Y1[t])) & ~15u) | (unsigned int)(p >> 1));
                m2[t] = __builtin_amdgcn_fmed3f(m1[t], m2[t], u);
                asm("v_min_f32 %0, %1, %2" : "=v"(m1[t]) : "v"(m1[t]), "v"(u));
            }
            if (VAR == 2) {
                for (int k = 0; k < 2 * ZT; ++k) { __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x2, 3, 0); }
            }
        }
    }
    }
    float acc = 0.f;
    for (int t = 0; t < ZT; ++t) acc += m1[t] + m2[t] + X0[t][0] + Y0[t][1];
    if (acc == 12345.678f) out[tid] = acc;
}

// 32x32x16 variant: one MFMA = 32 codes x 32 vectors, 16 outputs per lane; ZT = vector tiles of 32
template <int ZT, int MODE, int NT>
__global__ __launch_bounds__(NT, NT / 256) void scan32(const uint4 *__restrict__ tab, float *out, int reps, int np)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint4 *ldsA = reinterpret_cast<uint4 *>(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 32 * 64; i += NT) ldsA[i] = tab[i];
    __syncthreads();
    bf16x8 bop[ZT];
    float m1[ZT], m2[ZT];
    for (int t = 0; t < ZT; ++t) {
        uint4 b = tab[(lane + t * 64) & 4095];
        bop[t] = __builtin_bit_cast(bf16x8, b);
        m1[t] = __builtin_inff(); m2[t] = __builtin_inff();
    }
    f32x16 zero16; for (int i = 0; i < 16; ++i) zero16[i] = 0.f;
    f32x16 X[ZT], Y[ZT];
    for (int t = 0; t < ZT; ++t) { X[t] = zero16 + (float)lane; Y[t] = X[t]; }
    auto issue = [&](int p, f32x16 (&A)[ZT]) {     // p = 32-code tile
        if (MODE == 2) return;
        const int pp = p < np ? p : np - 1;
        const bf16x8 a0 = __builtin_bit_cast(bf16x8, ldsA[pp * 64 + lane]);
#pragma unroll
        for (int t = 0; t < ZT; ++t) A[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bop[t], zero16, 0, 0, 0);
    };
    auto chain = [&](float u, const f32x16 &A) -> float {
#pragma unroll
        for (int i = 0; i < 16; i += 2) u = __builtin_fminf(__builtin_fminf(u, A[i]), A[i + 1]);
        return u;
    };
    for (int r = 0; r < reps; ++r) {
        issue(0, X);
        for (int p = 0; p < np; p += 2) {      // 64 codes per iteration: two 32-code tiles
            issue(p + 1, Y);
            float uq[ZT];
            if (MODE != 1) {
#pragma unroll
                for (int t = 0; t < ZT; ++t) uq[t] = chain(__builtin_inff(), X[t]);
            } else {
#pragma unroll
                for (int t = 0; t < ZT; ++t) for (int i = 0; i < 16; ++i) asm volatile("" :: "v"(X[t][i]));
            }
            issue(p + 2, X);
            if (MODE != 1) {
#pragma unroll
                for (int t = 0; t < ZT; ++t) {
                    const float u = __uint_as_float((__float_as_uint(chain(uq[t], Y[t])) & ~15u) | (unsigned int)(p >> 1));
                    m2[t] = __builtin_amdgcn_fmed3f(m1[t], m2[t], u);
                    asm("v_min_f32 %0, %1, %2" : "=v"(m1[t]) : "v"(m1[t]), "v"(u));
                    if (MODE == 2) { X[t][0] += m1[t] * 1e-30f; Y[t][3] += m2[t] * 1e-30f; }
                }
            } else {
#pragma unroll
                for (int t = 0; t < ZT; ++t) for (int i = 0; i < 16; ++i) asm volatile("" :: "v"(Y[t][i]));
            }
        }
    }
    float acc = 0.f;
    for (int t = 0; t < ZT; ++t) acc += m1[t] + m2[t] + X[t][0] + Y[t][1];
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
        uint32_t h[4096 * 4];
        for (int i = 0; i < 4096 * 4; ++i) { uint32_t x = 0x3F803F80u ^ ((i * 2654435761u) & 0x007F007Fu); h[i] = x; }
        hipMemcpy(tab, h, sizeof(h), hipMemcpyHostToDevice);
    }
    const int reps = 64;      // 64 x (np/2 = 16 quad iterations) per wave
#define RUN16(ZT, MODE, NT, name)                                                                              \
    {                                                                                                          \
        hipFuncSetAttribute((const void *)scan16<ZT, MODE, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);                  \
        float us = timeit([&] { hipLaunchKernelGGL((scan16<ZT, MODE, NT>), dim3(256), dim3(NT), 65536, 0, tab, out, reps, 32); });  \
        double per = us * 1e3 / (reps * 16.0);                                                                \
        printf("16x16x32 ZT=%d %-10s %2d waves/SIMD: %7.1f ns per quad iteration per wave (%d MFMA), %6.1f ns per 64-vector-equivalent iteration per SIMD\n", ZT, name, NT / 256, per, 4 * ZT, per / (NT / 256) * (4.0 / ZT)); \
    }
    RUN16(4, 0, 512, "mfma+valu") RUN16(4, 1, 512, "mfma") RUN16(4, 2, 512, "valu")
#define RUNV(ZT, NT, VAR, name)                                                                              \
    {                                                                                                          \
        hipFuncSetAttribute((const void *)scan16<ZT, 0, NT, VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);                  \
        float us = timeit([&] { hipLaunchKernelGGL((scan16<ZT, 0, NT, VAR>), dim3(256), dim3(NT), 65536, 0, tab, out, reps, 32); });  \
        double per = us * 1e3 / (reps * 16.0);                                                                \
        printf("16x16x32 ZT=%d %-22s %2d waves/SIMD: %7.1f ns per quad iteration per wave, %6.1f ns per 64-vector-equivalent iteration per SIMD\n", ZT, name, NT / 256, per, per / (NT / 256) * (4.0 / ZT)); \
    }
    RUNV(4, 512, 1, "prefetch+schedbarrier") RUNV(4, 512, 2, "prefetch+group-interleave") RUNV(4, 512, 3, "setprio around mfma")
    RUNV(4, 256, 1, "prefetch+schedbarrier") RUNV(4, 256, 2, "prefetch+group-interleave")
    RUNV(2, 1024, 1, "prefetch+schedbarrier") RUNV(2, 1024, 2, "prefetch+group-interleave") RUNV(2, 1024, 3, "setprio around mfma")
    RUNV(4, 768, 1, "prefetch+schedbarrier") RUNV(4, 768, 2, "prefetch+group-interleave")
#define RUN32(ZT, MODE, NT, name)                                                                              \
    {                                                                                                          \
        hipFuncSetAttribute((const void *)scan32<ZT, MODE, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);                  \
        float us = timeit([&] { hipLaunchKernelGGL((scan32<ZT, MODE, NT>), dim3(256), dim3(NT), 65536, 0, tab, out, reps, 32); });  \
        double per = us * 1e3 / (reps * 16.0);                                                                \
        printf("32x32x16 ZT=%d %-10s %2d waves/SIMD: %7.1f ns per quad iteration per wave (%d MFMA), %6.1f ns per 64-vector-equivalent iteration per SIMD\n", ZT, name, NT / 256, per, 2 * ZT, per / (NT / 256) * (2.0 / ZT)); \
    }
    RUN32(2, 0, 512, "mfma+valu") RUN32(1, 0, 1024, "mfma+valu")
    return 0;
}
