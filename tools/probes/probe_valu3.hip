// dev probe (round 6): issue cost of the candidate VALU instructions for the VQ scan's running minimum, per wave64 instruction,
// at 1 / 2 waves per SIMD: 8 independent accumulator chains, 64 instructions per loop trip, operands in distinct / equal VGPR banks.
//   hipcc -O3 --offload-arch=gfx950 probe_valu3.hip -o probe_valu3 && ./probe_valu3
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
// one trip = 8 x (8 chains) = 64 instructions; a[k] = accumulators, b / c = sources
#define KERNEL(NAME, BODY)                                                                                      \
    __global__ __launch_bounds__(512) void NAME(float *out, int trips, float seed)                            \
    {                                                                                                           \
        float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        float b0 = a0 * 0.5f, b1 = a1 * 0.5f, b2 = a2 * 0.5f, b3 = a3 * 0.5f, c0 = a0 * 0.25f, c1 = a1 * 0.25f, c2 = a2 * 0.25f, c3 = a3 * 0.25f; \
        for (int t = 0; t < trips; ++t) {                                                                       \
            _Pragma("unroll") for (int r = 0; r < 8; ++r) { BODY }                                            \
        }                                                                                                       \
        const float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + b0 + b1 + b2 + b3 + c0 + c1 + c2 + c3;                                         \
        if (s == 12345.678f) out[threadIdx.x] = s;                                                              \
    }
#define OP3(ins, k, x, y) asm volatile(ins " %0, %0, %1, %2" : "+v"(a##k) : "v"(x), "v"(y));
#define OP2(ins, k, x) asm volatile(ins " %0, %0, %1" : "+v"(a##k) : "v"(x));

KERNEL(k_fma, OP3("v_fma_f32", 0, b0, c0) OP3("v_fma_f32", 1, b1, c1) OP3("v_fma_f32", 2, b2, c2) OP3("v_fma_f32", 3, b3, c3) OP3("v_fma_f32", 4, b0, c1) OP3("v_fma_f32", 5, b1, c2) OP3("v_fma_f32", 6, b2, c3) OP3("v_fma_f32", 7, b3, c0))
KERNEL(k_min, OP2("v_min_f32", 0, b0) OP2("v_min_f32", 1, b1) OP2("v_min_f32", 2, b2) OP2("v_min_f32", 3, b3) OP2("v_min_f32", 4, c0) OP2("v_min_f32", 5, c1) OP2("v_min_f32", 6, c2) OP2("v_min_f32", 7, c3))
KERNEL(k_add, OP2("v_add_f32", 0, b0) OP2("v_add_f32", 1, b1) OP2("v_add_f32", 2, b2) OP2("v_add_f32", 3, b3) OP2("v_add_f32", 4, c0) OP2("v_add_f32", 5, c1) OP2("v_add_f32", 6, c2) OP2("v_add_f32", 7, c3))
KERNEL(k_min3, OP3("v_min3_f32", 0, b0, c0) OP3("v_min3_f32", 1, b1, c1) OP3("v_min3_f32", 2, b2, c2) OP3("v_min3_f32", 3, b3, c3) OP3("v_min3_f32", 4, b0, c1) OP3("v_min3_f32", 5, b1, c2) OP3("v_min3_f32", 6, b2, c3) OP3("v_min3_f32", 7, b3, c0))
KERNEL(k_minimum3, OP3("v_minimum3_f32", 0, b0, c0) OP3("v_minimum3_f32", 1, b1, c1) OP3("v_minimum3_f32", 2, b2, c2) OP3("v_minimum3_f32", 3, b3, c3) OP3("v_minimum3_f32", 4, b0, c1) OP3("v_minimum3_f32", 5, b1, c2) OP3("v_minimum3_f32", 6, b2, c3) OP3("v_minimum3_f32", 7, b3, c0))
KERNEL(k_min3i, OP3("v_min3_i32", 0, b0, c0) OP3("v_min3_i32", 1, b1, c1) OP3("v_min3_i32", 2, b2, c2) OP3("v_min3_i32", 3, b3, c3) OP3("v_min3_i32", 4, b0, c1) OP3("v_min3_i32", 5, b1, c2) OP3("v_min3_i32", 6, b2, c3) OP3("v_min3_i32", 7, b3, c0))
KERNEL(k_min3u, OP3("v_min3_u32", 0, b0, c0) OP3("v_min3_u32", 1, b1, c1) OP3("v_min3_u32", 2, b2, c2) OP3("v_min3_u32", 3, b3, c3) OP3("v_min3_u32", 4, b0, c1) OP3("v_min3_u32", 5, b1, c2) OP3("v_min3_u32", 6, b2, c3) OP3("v_min3_u32", 7, b3, c0))
KERNEL(k_mini, OP2("v_min_i32", 0, b0) OP2("v_min_i32", 1, b1) OP2("v_min_i32", 2, b2) OP2("v_min_i32", 3, b3) OP2("v_min_i32", 4, c0) OP2("v_min_i32", 5, c1) OP2("v_min_i32", 6, c2) OP2("v_min_i32", 7, c3))
KERNEL(k_minu, OP2("v_min_u32", 0, b0) OP2("v_min_u32", 1, b1) OP2("v_min_u32", 2, b2) OP2("v_min_u32", 3, b3) OP2("v_min_u32", 4, c0) OP2("v_min_u32", 5, c1) OP2("v_min_u32", 6, c2) OP2("v_min_u32", 7, c3))
KERNEL(k_med3, OP3("v_med3_f32", 0, b0, c0) OP3("v_med3_f32", 1, b1, c1) OP3("v_med3_f32", 2, b2, c2) OP3("v_med3_f32", 3, b3, c3) OP3("v_med3_f32", 4, b0, c1) OP3("v_med3_f32", 5, b1, c2) OP3("v_med3_f32", 6, b2, c3) OP3("v_med3_f32", 7, b3, c0))
KERNEL(k_andor, OP3("v_and_or_b32", 0, b0, c0) OP3("v_and_or_b32", 1, b1, c1) OP3("v_and_or_b32", 2, b2, c2) OP3("v_and_or_b32", 3, b3, c3) OP3("v_and_or_b32", 4, b0, c1) OP3("v_and_or_b32", 5, b1, c2) OP3("v_and_or_b32", 6, b2, c3) OP3("v_and_or_b32", 7, b3, c0))
KERNEL(k_perm, OP3("v_perm_b32", 0, b0, c0) OP3("v_perm_b32", 1, b1, c1) OP3("v_perm_b32", 2, b2, c2) OP3("v_perm_b32", 3, b3, c3) OP3("v_perm_b32", 4, b0, c1) OP3("v_perm_b32", 5, b1, c2) OP3("v_perm_b32", 6, b2, c3) OP3("v_perm_b32", 7, b3, c0))
KERNEL(k_pkmin3h, OP3("v_pk_minimum3_f16", 0, b0, c0) OP3("v_pk_minimum3_f16", 1, b1, c1) OP3("v_pk_minimum3_f16", 2, b2, c2) OP3("v_pk_minimum3_f16", 3, b3, c3) OP3("v_pk_minimum3_f16", 4, b0, c1) OP3("v_pk_minimum3_f16", 5, b1, c2) OP3("v_pk_minimum3_f16", 6, b2, c3) OP3("v_pk_minimum3_f16", 7, b3, c0))
KERNEL(k_pkminh, OP2("v_pk_min_f16", 0, b0) OP2("v_pk_min_f16", 1, b1) OP2("v_pk_min_f16", 2, b2) OP2("v_pk_min_f16", 3, b3) OP2("v_pk_min_f16", 4, c0) OP2("v_pk_min_f16", 5, c1) OP2("v_pk_min_f16", 6, c2) OP2("v_pk_min_f16", 7, c3))
KERNEL(k_pkmini16, OP2("v_pk_min_i16", 0, b0) OP2("v_pk_min_i16", 1, b1) OP2("v_pk_min_i16", 2, b2) OP2("v_pk_min_i16", 3, b3) OP2("v_pk_min_i16", 4, c0) OP2("v_pk_min_i16", 5, c1) OP2("v_pk_min_i16", 6, c2) OP2("v_pk_min_i16", 7, c3))
KERNEL(k_cvtpk, OP2("v_cvt_pk_f16_f32", 0, b0) OP2("v_cvt_pk_f16_f32", 1, b1) OP2("v_cvt_pk_f16_f32", 2, b2) OP2("v_cvt_pk_f16_f32", 3, b3) OP2("v_cvt_pk_f16_f32", 4, c0) OP2("v_cvt_pk_f16_f32", 5, c1) OP2("v_cvt_pk_f16_f32", 6, c2) OP2("v_cvt_pk_f16_f32", 7, c3))
KERNEL(k_cvtrtz, OP2("v_cvt_pkrtz_f16_f32", 0, b0) OP2("v_cvt_pkrtz_f16_f32", 1, b1) OP2("v_cvt_pkrtz_f16_f32", 2, b2) OP2("v_cvt_pkrtz_f16_f32", 3, b3) OP2("v_cvt_pkrtz_f16_f32", 4, c0) OP2("v_cvt_pkrtz_f16_f32", 5, c1) OP2("v_cvt_pkrtz_f16_f32", 6, c2) OP2("v_cvt_pkrtz_f16_f32", 7, c3))
KERNEL(k_minsdwa, asm volatile("v_min_f32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD" : "+v"(a0) : "v"(b0)); asm volatile("v_min_f32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD" : "+v"(a1) : "v"(b1)); asm volatile("v_min_f32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD" : "+v"(a2) : "v"(b2)); asm volatile("v_min_f32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD" : "+v"(a3) : "v"(b3)); asm volatile("v_min_f32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD" : "+v"(a4) : "v"(c0)); asm volatile("v_min_f32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD" : "+v"(a5) : "v"(c1)); asm volatile("v_min_f32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD" : "+v"(a6) : "v"(c2)); asm volatile("v_min_f32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD" : "+v"(a7) : "v"(c3));)

// packed fp32: operands are register PAIRS
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define KERNEL2(NAME, INS)                                                                                      \
    __global__ __launch_bounds__(512) void NAME(float *out, int trips, float seed)                            \
    {                                                                                                           \
        f32x2 a[8], b[4];                                                                                        \
        for (int k = 0; k < 8; ++k) a[k] = f32x2{seed + threadIdx.x + k, seed - k};                              \
        for (int k = 0; k < 4; ++k) b[k] = f32x2{seed * 0.5f + k, seed * 0.25f - k};                             \
        for (int t = 0; t < trips; ++t) {                                                                       \
            _Pragma("unroll") for (int r = 0; r < 8; ++r) {                                                   \
                _Pragma("unroll") for (int k = 0; k < 8; ++k) asm volatile(INS " %0, %0, %1" : "+v"(a[k]) : "v"(b[k & 3])); \
            }                                                                                                   \
        }                                                                                                       \
        float s = 0.f;                                                                                          \
        for (int k = 0; k < 8; ++k) s += a[k][0] + a[k][1];                                                      \
        if (s == 12345.678f) out[threadIdx.x] = s;                                                              \
    }
KERNEL2(k_pkadd, "v_pk_add_f32")
KERNEL2(k_pkmul, "v_pk_mul_f32")
KERNEL2(k_minf64, "v_min_f64")

template <typename K>
static void run(const char *name, K kern, float *out, int threads)
{
    const int trips = 4096;
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, out, trips, 1.5f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, out, trips, 1.5f);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    const double ns = ms * 1e6 / 3;
    const int waves = threads / 256;            // per SIMD
    // a SIMD issued waves x trips x 64 instructions
    printf("%-22s %d waves/SIMD: %6.2f ns = %5.2f cycles @2.4GHz per wave64 instruction\n", name, waves, ns / ((double)waves * trips * 64), ns / ((double)waves * trips * 64) * 2.4);
}

int main()
{
    float *out;
    (void)hipMalloc(&out, 4096);
    for (int threads : {256, 512}) {
        run("v_fma_f32", k_fma, out, threads);
        run("v_add_f32", k_add, out, threads);
        run("v_min_f32", k_min, out, threads);
        run("v_min3_f32", k_min3, out, threads);
        run("v_minimum3_f32", k_minimum3, out, threads);
        run("v_min3_i32", k_min3i, out, threads);
        run("v_min3_u32", k_min3u, out, threads);
        run("v_min_i32", k_mini, out, threads);
        run("v_min_u32", k_minu, out, threads);
        run("v_med3_f32", k_med3, out, threads);
        run("v_and_or_b32", k_andor, out, threads);
        run("v_perm_b32", k_perm, out, threads);
        run("v_pk_minimum3_f16", k_pkmin3h, out, threads);
        run("v_pk_min_f16", k_pkminh, out, threads);
        run("v_pk_min_i16", k_pkmini16, out, threads);
        run("v_cvt_pk_f16_f32", k_cvtpk, out, threads);
        run("v_cvt_pkrtz_f16_f32", k_cvtrtz, out, threads);
        run("v_min_f32_sdwa", k_minsdwa, out, threads);
        run("v_pk_add_f32", k_pkadd, out, threads);
        run("v_pk_mul_f32", k_pkmul, out, threads);
        run("v_min_f64", k_minf64, out, threads);
    }
    return 0;
}
