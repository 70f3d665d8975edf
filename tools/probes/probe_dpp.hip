// dev probe: DPP/permlane-swap butterfly sum vs ds_bpermute butterfly on gfx950
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v)
{
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ float dpp_get(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__global__ void k(float *out)
{
    float x = (float)(threadIdx.x * threadIdx.x % 17) * 0.37f + threadIdx.x;
    float a = x;
    for (int off = 1; off < 32; off <<= 1) a += __shfl_xor(a, off, 64);
    float v = x;
    v = dpp_add<0xB1>(v); v = dpp_add<0x4E>(v); v = dpp_add<0x141>(v); v = dpp_add<0x140>(v);
    float va = v, vb = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(va), "+v"(vb));
    v = va + vb;
    float id = (float)threadIdx.x;
    out[threadIdx.x] = a; out[64 + threadIdx.x] = v;
    out[128 + threadIdx.x] = dpp_get<0xB1>(id); out[192 + threadIdx.x] = dpp_get<0x4E>(id);
    out[256 + threadIdx.x] = dpp_get<0x141>(id); out[320 + threadIdx.x] = dpp_get<0x140>(id);
}
int main()
{
    float *d, h[384];
    (void)hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64; ++i) bad += h[i] != h[64 + i];
    printf("butterfly mismatches: %d (lane0 %.6f vs %.6f)\n", bad, h[0], h[64]);
    const char *n[4] = {"quad_perm[1,0,3,2]", "quad_perm[2,3,0,1]", "row_half_mirror", "row_mirror"};
    for (int a = 0; a < 4; ++a) { printf("%-20s:", n[a]); for (int i = 0; i < 16; ++i) printf(" %g", h[128 + 64 * a + i]); printf("\n"); }
    return 0;
}
