// dev probe (round 3): can the LDS atomic unit take part of the VQ filter scan's min bookkeeping off the VALU?
// The scan digests 16 fp32 scores per lane and MFMA with 8 half-rate v_min3 (+ and/or, med3, min): VALU-issue-bound.
// Variant L routes L of the 16 scores through a per-lane LDS slot (ds_write_b32, L-1 x ds_min_f32, ds_read_b32 one tile later)
// and keeps ceil((17-L)/2) v_min3 on the VALU.
//   hipcc -O3 --offload-arch=gfx950 probe_scan_lds.hip -o probe_scan_lds && ./probe_scan_lds
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int L, int NT>
__global__ __launch_bounds__(NT, 1) void scan(const uint4 *__restrict__ tab, float *out, int reps, int np)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint4 *ldsA = reinterpret_cast<uint4 *>(smem);                       // 32 tiles x 64 lanes x 16 B = 32 KB
    float *slots = reinterpret_cast<float *>(smem + 32768);             // [2 parity][2 t][NT]
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 32 * 64; i += NT) ldsA[i] = tab[i];
    __syncthreads();
    f16x8 bop[2];
    float m1[2], m2[2];
    for (int t = 0; t < 2; ++t) {
        uint4 b = tab[(lane + t * 64) & 4095];
        bop[t] = __builtin_bit_cast(f16x8, b);
        m1[t] = __builtin_inff(); m2[t] = __builtin_inff();
    }
    f32x16 zero16; for (int i = 0; i < 16; ++i) zero16[i] = 0.f;
    f32x16 X[2], Y[2];
    for (int t = 0; t < 2; ++t) { X[t] = zero16 + (float)lane; Y[t] = X[t]; }
    auto issue = [&](int p, f32x16 (&A)[2]) {
        const int pp = p < np ? p : np - 1;
        const f16x8 a0 = __builtin_bit_cast(f16x8, ldsA[pp * 64 + lane]);
#pragma unroll
        for (int t = 0; t < 2; ++t) A[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, bop[t], zero16, 0, 0, 0);
    };
    const unsigned int slot_addr = 32768u + (unsigned int)tid * 4u;      // byte address in LDS; + (parity * 2 + t) * NT * 4
    auto digest = [&](int T, const f32x16 (&D)[2], int parity) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float u = __builtin_inff();
            if (L == 0) {
#pragma unroll
                for (int r = 0; r < 16; r += 2) u = __builtin_fminf(__builtin_fminf(u, D[t][r]), D[t][r + 1]);
            } else {
                // the previous tile of this parity left its routed minimum in the slot: fetch it, then restart the slot with this tile
                const unsigned int a = slot_addr + (unsigned int)((parity * 2 + t) * NT * 4);
                float prev;
                asm volatile("ds_read_b32 %0, %1" : "=v"(prev) : "v"(a) : "memory");
                asm volatile("ds_write_b32 %0, %1" :: "v"(a), "v"(D[t][16 - L]) : "memory");
#pragma unroll
                for (int r = 17 - L; r < 16; ++r) asm volatile("ds_min_f32 %0, %1" :: "v"(a), "v"(D[t][r]) : "memory");
                // VALU share: scores 0 .. 15-L of THIS tile
                constexpr int NV = 16 - L;
#pragma unroll
                for (int r = 0; r + 1 < NV; r += 2) u = __builtin_fminf(__builtin_fminf(u, D[t][r]), D[t][r + 1]);
                asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(prev) : "n"(L) : "memory");      // the read is older than this tile's L stores
                if (NV & 1) u = __builtin_fminf(__builtin_fminf(u, D[t][NV - 1]), prev);
                else u = __builtin_fminf(u, prev);
                // (the real kernel would merge `prev` into the PREVIOUS tile's minimum; for issue-rate purposes the count is the same)
            }
            u = __uint_as_float((__float_as_uint(u) & ~31u) | (unsigned int)T);
            m2[t] = __builtin_amdgcn_fmed3f(m1[t], m2[t], u);
            asm("v_min_f32 %0, %1, %2" : "=v"(m1[t]) : "v"(m1[t]), "v"(u));
        }
    };
    if (L) { for (int k = 0; k < 4; ++k) slots[k * NT + tid] = __builtin_inff(); }
    for (int r = 0; r < reps; ++r) {
        issue(0, X);
        for (int p = 0; p < np; p += 2) {
            issue(p + 1, Y);
            digest(p, X, 0);
            issue(p + 2, X);
            digest(p + 1, Y, 1);
        }
    }
    float acc = 0.f;
    for (int t = 0; t < 2; ++t) acc += m1[t] + m2[t] + X[t][0] + Y[t][1];
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
#define RUN(L, NT)                                                                                                          \
    {                                                                                                                       \
        hipFuncSetAttribute((const void *)scan<L, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);                 \
        float us = timeit([&] { hipLaunchKernelGGL((scan<L, NT>), dim3(256), dim3(NT), 32768 + 4 * NT * 4, 0, tab, out, reps, 32); }); \
        double per = us * 1e3 / (reps * 16.0);                                                                              \
        printf("L=%d routed via LDS, %d waves/SIMD: %7.1f ns per 64-code iteration per wave, %6.1f ns per SIMD\n", L, NT / 256, per, per / (NT / 256)); \
    }
    RUN(0, 512) RUN(3, 512) RUN(5, 512) RUN(7, 512) RUN(9, 512)
    RUN(0, 768) RUN(5, 768) RUN(7, 768)
    RUN(0, 256) RUN(5, 256)
    return 0;
}
