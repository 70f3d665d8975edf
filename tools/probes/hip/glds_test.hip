// dev probe: LDS-DMA (global_load_lds_dwordx4) from inline asm with counted vmcnt waits across raw barriers
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ void glds16(const void *gsrc, unsigned int lds_dst)
{
    unsigned int keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

extern "C" __global__ __launch_bounds__(512) void k(const uint4 *src, uint4 *dst, int npieces, long long *clk)
{
    extern __shared__ unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned int base = (unsigned int)(size_t)(__attribute__((address_space(3))) unsigned char *)smem;
    long long t0 = clock64();
    // wave w loads pieces w, w + 8, ... (1 KB each)
    for (int p = wave; p < npieces; p += 8)
        glds16(src + p * 64 + lane, __builtin_amdgcn_readfirstlane(base + p * 1024));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    long long t1 = clock64();
    const uint4 *l = reinterpret_cast<const uint4 *>(smem);
    for (int i = tid; i < npieces * 64; i += 512) dst[blockIdx.x * npieces * 64 + i] = l[i];
    if (tid == 0) clk[blockIdx.x] = t1 - t0;
}

int main()
{
    const int np = 54, nb = 256;
    std::vector<unsigned int> h(np * 256);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned int)i * 2654435761u;
    uint4 *s, *d; long long *c;
    hipMalloc(&s, np * 1024); hipMalloc(&d, (size_t)nb * np * 1024); hipMalloc(&c, nb * 8);
    hipMemcpy(s, h.data(), np * 1024, hipMemcpyHostToDevice);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k, dim3(nb), dim3(512), np * 1024, 0, s, d, np, c);
    hipDeviceSynchronize();
    std::vector<unsigned int> o((size_t)nb * np * 256);
    hipMemcpy(o.data(), d, o.size() * 4, hipMemcpyDeviceToHost);
    size_t bad = 0;
    for (size_t i = 0; i < o.size(); ++i) bad += o[i] != h[i % h.size()];
    std::vector<long long> cl(nb);
    hipMemcpy(cl.data(), c, nb * 8, hipMemcpyDeviceToHost);
    long long mx = 0, mn = 1ll << 60;
    for (auto v : cl) { mx = v > mx ? v : mx; mn = v < mn ? v : mn; }
    printf("mismatches %zu of %zu; copy cycles min %lld max %lld (%s)\n", bad, o.size(), mn, mx, hipGetErrorString(hipGetLastError()));
    return bad != 0;
}
