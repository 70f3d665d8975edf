// dev probe: VALU issue cost per instruction type on gfx950 (cycles per wave64 instruction per SIMD)
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int OP>
__global__ void k(float *out, int iters, float seed)
{
    float a[8], b = seed + threadIdx.x, c = seed * 3.f;
    int sel[8];
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 pk[8], pkb = {seed, seed + 1.f}, pkc = {seed * 2.f, seed};
    for (int i = 0; i < 8; ++i) pk[i] = f2{seed * i, seed + i};
    for (int i = 0; i < 8; ++i) { a[i] = seed * (i + 1) + threadIdx.x; sel[i] = i; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 1) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            if (OP == 2) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            if (OP == 3) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            if (OP == 4) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 5) asm volatile("v_cmp_lt_f32 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %3, vcc" : "+v"(sel[i]) : "v"(a[i]), "v"(b), "v"(it) : "vcc");
            if (OP == 6) asm volatile("v_min3_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 7) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(pk[i]) : "v"(pkb));
            if (OP == 8) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(pk[i]) : "v"(pkb), "v"(pkc));
            if (OP == 9) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(pk[i]) : "v"(pkb));
            if (OP == 10) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
            if (OP == 11) asm volatile("ds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(0)" : "+v"(a[i]) : "v"(sel[i] * 4));
            if (OP == 12) { auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a[i]), __float_as_uint(a[i]), false, false); a[i] = __uint_as_float(r[0] ^ r[1]); }
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i] + sel[i] + pk[i][0] + pk[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP>
void run(const char *name, float *out, int waves_per_simd)
{
    const int iters = 4000, nthreads = 64 * 4 * waves_per_simd;   // one block per CU
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(nthreads), 0, 0, out, iters, 1.5f);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(nthreads), 0, 0, out, iters, 1.5f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)iters * 8 * waves_per_simd * (OP == 5 ? 2 : 1);
    printf("%-28s %d waves/SIMD: %.2f ns per instr per SIMD (%.2f cycles at 2.4 GHz)\n", name, waves_per_simd, ms * 1e6 / n, ms * 1e6 / n * 2.4);
}
int main()
{
    float *out; (void)hipMalloc(&out, 256 * 1024 * 4);
    for (int w : {1, 2, 4}) {
        run<0>("v_min_f32", out, w); run<1>("v_min3_f32 (3 vgpr)", out, w); run<6>("v_min3_f32 (2 distinct)", out, w);
        run<2>("v_med3_f32", out, w); run<3>("v_fma_f32", out, w); run<4>("v_add_f32", out, w); run<5>("v_cmp+v_cndmask", out, w);
        run<7>("v_pk_add_f32", out, w); run<8>("v_pk_fma_f32", out, w); run<9>("v_pk_mul_f32", out, w); run<10>("v_exp_f32", out, w);
        run<11>("ds_bpermute+wait", out, w); run<12>("permlane32_swap(+2 mov+xor)", out, w);
    }
    return 0;
}
