// dev probe: lane semantics of v_permlane16_swap / v_permlane32_swap on gfx950
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned *out)
{
    unsigned x = threadIdx.x;
    auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false);
    auto q = __builtin_amdgcn_permlane32_swap(x, x, false, false);
    out[threadIdx.x] = r[0]; out[64 + threadIdx.x] = r[1]; out[128 + threadIdx.x] = q[0]; out[192 + threadIdx.x] = q[1];
}
int main()
{
    unsigned *d, h[256];
    (void)hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char *names[4] = {"swap16[0]", "swap16[1]", "swap32[0]", "swap32[1]"};
    for (int a = 0; a < 4; ++a) { printf("%s:", names[a]); for (int i = 0; i < 64; i += 8) printf(" %u", h[64 * a + i]); printf("\n"); }
    return 0;
}
