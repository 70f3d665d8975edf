// dev probe: issue cost of individual VALU opcodes on gfx950 at 2 and 4 waves/SIMD
#include <hip/hip_runtime.h>
#include <stdio.h>
#define OPS(X) \
  X(0, "v_add_f32 %0, %0, %1") X(1, "v_mul_f32 %0, %0, %1") X(2, "v_sub_f32 %0, %0, %1") X(3, "v_max_f32 %0, %0, %1") \
  X(4, "v_min_f32 %0, %0, %1") X(5, "v_and_b32 %0, %0, %1") X(6, "v_or_b32 %0, %0, %1") X(7, "v_add_u32 %0, %0, %1") \
  X(8, "v_min_u32 %0, %0, %1") X(9, "v_min_i32 %0, %0, %1") X(10, "v_mov_b32 %0, %1") X(11, "v_lshlrev_b32 %0, 1, %0") \
  X(12, "v_min3_u32 %0, %0, %1, %2") X(13, "v_min3_f32 %0, %0, %1, %2") X(14, "v_med3_f32 %0, %0, %1, %2") X(15, "v_med3_i32 %0, %0, %1, %2") \
  X(16, "v_and_or_b32 %0, %0, %1, %2") X(17, "v_fma_f32 %0, %0, %1, %2") X(18, "v_lshl_add_u32 %0, %0, 1, %1") X(19, "v_cndmask_b32 %0, %0, %1, vcc") \
  X(20, "v_cmp_lt_f32 vcc, %0, %1") X(21, "v_cmp_lt_u32 vcc, %0, %1") X(22, "v_fmac_f32 %0, %1, %2") X(23, "v_mul_f32 %0, %1, %2") \
  X(24, "v_exp_f32 %0, %0") X(25, "v_rcp_f32 %0, %0") X(26, "v_mul_legacy_f32 %0, %0, %1") X(27, "v_add3_u32 %0, %0, %1, %2") \
  X(28, "v_max3_f32 %0, %0, %1, %2") X(29, "v_perm_b32 %0, %0, %1, %2") X(30, "v_xor_b32 %0, %0, %1") X(31, "v_min_f32 %0, %1, %2") \
  X(32, "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1") X(33, "v_add_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1") \
  X(34, "v_add_f32_dpp %0, %1, %0 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1") X(35, "s_nop 1\n\tv_permlane16_swap_b32 %0, %1") X(36, "v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1")
template <int OP>
__global__ void k(float *out, int iters, float seed)
{
    float a[8], b = seed + threadIdx.x, c = seed * 3.f;
    for (int i = 0; i < 8; ++i) a[i] = seed * (i + 1) + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
#define X(n, s) if (OP == n) asm volatile(s : "+v"(a[i]), "+v"(b) : "v"(c) : "vcc");
            OPS(X)
#undef X
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP>
double run(float *out, int w)
{
    const int iters = 4000, nthreads = 256 * w;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(nthreads), 0, 0, out, iters, 1.5f);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(nthreads), 0, 0, out, iters, 1.5f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e6 / ((double)iters * 8 * w);
}
int main()
{
    float *out; (void)hipMalloc(&out, 256 * 1024 * 4);
#define X(n, s) printf("%-34s  1w %.2f  2w %.2f  4w %.2f  8w %.2f ns/instr/SIMD\n", s, run<n>(out, 1), run<n>(out, 2), run<n>(out, 4), run<n>(out, 4));
    OPS(X)
#undef X
    return 0;
}
