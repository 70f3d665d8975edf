// dev probe (round 6): what a refinement UNIT costs (64 pixels of a patch in the reference's own arithmetic: window, five exp,
// chunk sums in torch's order; cgic_entropy_dev.h) by variant, in throughput terms (2 / 4 waves per SIMD, every wave evaluating
// units back to back) -- and an EXHAUSTIVE check of the fast exp against the fp64 library exp over every fp32 argument in [-104, 0].
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -I../../control-gic_amd/csrc probe_unit.hip -o probe_unit
#include "cgic_entropy_dev.h"
using namespace cgic;

// ---- candidate: exp(a) rounded once to fp32, a in [-104, 0], from a short fp64 evaluation + a Ziv test (fallback: the library exp)
__device__ __forceinline__ float fast_exp_rn(float a, bool *slow)
{
    const double x = (double)a;
    const double k = __builtin_rint(x * 1.4426950408889634);            // a / ln 2
    const double r = __builtin_fma(-k, 1.9082149292705877e-10, __builtin_fma(-k, 6.93147180369123816490e-01, x));   // |r| <= 0.3466 (+ a hair)
    // Taylor to degree 13 (remainder 0.35^14 / 14! = 4.7e-18 relative), Horner in fp64
    double p = 1.0 / 6227020800.0;
    p = __builtin_fma(p, r, 1.0 / 479001600.0);
    p = __builtin_fma(p, r, 1.0 / 39916800.0);
    p = __builtin_fma(p, r, 1.0 / 3628800.0);
    p = __builtin_fma(p, r, 1.0 / 362880.0);
    p = __builtin_fma(p, r, 1.0 / 40320.0);
    p = __builtin_fma(p, r, 1.0 / 5040.0);
    p = __builtin_fma(p, r, 1.0 / 720.0);
    p = __builtin_fma(p, r, 1.0 / 120.0);
    p = __builtin_fma(p, r, 1.0 / 24.0);
    p = __builtin_fma(p, r, 1.0 / 6.0);
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    const double y = __builtin_ldexp(p, (int)k);
    const float f = (float)y;
    // Ziv: is y within 2^-40 (relative) of the midpoint between f and a neighbour?  then the last bit is not certain
    const float fn = __uint_as_float(__float_as_uint(f) + 1u);
    const double half = 0.5 * ((double)fn - (double)f);                  // half an ulp of f (subnormals included)
    const double d = __builtin_fabs(y - (double)f);
    *slow = __builtin_fabs(d - half) <= y * 9.094947017729282e-13;        // 2^-40
    return f;
}


// ---- candidate: chunk sums of a NARROW unit (all 64 window starts within one bin of each other: six active bins) from a dense
// [pixel][6] matrix in the record area -- 3 writes + 16 reads + 16 additions per lane instead of 64 reads + ~250 VALU
__device__ __forceinline__ unsigned int wave_or_u32(unsigned int v)
{
    v |= (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);
    v |= (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);
    v |= (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true);
    v |= (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true);
    unsigned int a = v, b = v;
    swap16(a, b);
    v = a | b;
    a = v; b = v;
    swap32(a, b);
    return a | b;
}
__device__ __forceinline__ void unit_chunks_fast(float *rec, int j0, const float v[kRefWin], float *T)
{
    const int lane = lane_id();
    const unsigned int mask = (unsigned int)__builtin_amdgcn_readfirstlane((int)wave_or_u32(1u << j0));
    const int jmin = __builtin_ctz(mask), jmax = 31 - __builtin_clz(mask);
    if (jmax - jmin > 1) { ref_unit_chunks(rec, j0, v, T); return; }            // (wave-uniform)
    const bool up = j0 != jmin;                                                // this pixel's window starts one bin later
    float2 w0, w1, w2;
    w0.x = up ? 0.f : v[0];  w0.y = up ? v[0] : v[1];
    w1.x = up ? v[1] : v[2]; w1.y = up ? v[2] : v[3];
    w2.x = up ? v[3] : v[4]; w2.y = up ? v[4] : 0.f;
    float2 *row = reinterpret_cast<float2 *>(rec + lane * 6);
    row[0] = w0; row[1] = w1; row[2] = w2;
    T[(lane >> 5) * kRefRow + (lane & 31)] = 0.f;
    T[(2 + (lane >> 5)) * kRefRow + (lane & 31)] = 0.f;
    __builtin_amdgcn_wave_barrier();
    if (lane < 24) {
        const int c = lane / 6, b = lane - 6 * c;
        const float *col = rec + (16 * c) * 6 + b;
        float x[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = col[6 * i];
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc = acc + x[i];
        if (jmin + b < kBins) T[c * kRefRow + jmin + b] = acc;
    }
    __builtin_amdgcn_wave_barrier();
}

template <int VAR>
__device__ __forceinline__ float bin_of(const float *bins, int j) { return (VAR >= 1 && VAR <= 3) ? linspace_bin(j) : bins[j]; }

template <int VAR>
__device__ __forceinline__ void pixel_var(const float *bins, float sigma, float gray, int &j0, float v[kRefWin])
{
    {
        const float t = gray - 0.1445f;
        float g = floorf((t - bin_of<VAR>(bins, 0)) * 15.5f) + 1.0f;
        g = g >= 0.f ? g : 0.f;
        g = g > (float)kBins ? (float)kBins : g;
        int j = (int)g;
        while (j > 0 && !(bin_of<VAR>(bins, j - 1) < t)) --j;
        while (j < kBins && bin_of<VAR>(bins, j) < t) ++j;
        j0 = j > kBins - kRefWin ? kBins - kRefWin : j;
    }
    const bool fast = sigma == 0.01f;
    bool any_slow = false;
    float a_k[kRefWin];
#pragma unroll
    for (int k = 0; k < kRefWin; ++k) {
        const float res = gray - bin_of<VAR>(bins, j0 + k);
        float t = div_by_sigma001(res);
        if (!(fast && fabsf(res) < 8.0f)) t = res / sigma;
        const float t2 = t * t;
        const float a = -0.5f * t2;
        a_k[k] = a;
        if (VAR == 3) v[k] = a < -104.0f ? 0.f : a * 0.001f;                         // no exp at all: the ceiling
        else if (VAR == 2) {
            bool slow = false;
            const float f = fast_exp_rn(a, &slow);
            v[k] = (a < -104.0f) ? 0.f : f;
            any_slow |= slow && !(a < -104.0f);
            if (!(a == a)) v[k] = a;                                                // NaN stays NaN
        } else v[k] = (a < -104.0f) ? 0.f : (float)exp((double)a);
    }
    if (VAR == 2 && any_slow) {
#pragma unroll
        for (int k = 0; k < kRefWin; ++k) v[k] = (a_k[k] < -104.0f) ? 0.f : (float)exp((double)a_k[k]);
    }
}

template <int VAR, int NT>
__global__ __launch_bounds__(NT) void unit_kernel(const float *__restrict__ grays, float *out, int units)
{
    constexpr int NW = NT / 64;
    __shared__ float sT[NW][kRefUnitRows * kRefRow];
    __shared__ float sRec[NW][kRefRecFloats];
    __shared__ float sP[NW][2 * kBins];
    __shared__ float sBins[kBins];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < kBins) sBins[tid] = linspace_bin(tid);
    __syncthreads();
    float total = 0.f;
    const float *g = grays + ((size_t)blockIdx.x * NW + wave) * 64 * 4;
    for (int u = 0; u < units; u += 4) {
        float acc = 0.f;
        for (int q = 0; q < 4; ++q) {
            const float gray = g[((u + q) & 3) * 64 + lane] + (float)(u >> 2) * 1e-4f;
            int j0;
            float v[kRefWin];
            pixel_var<VAR>(sBins, 0.01f, gray, j0, v);
            if (VAR == 4) unit_chunks_fast(sRec[wave], j0, v, sT[wave]); else ref_unit_chunks(sRec[wave], j0, v, sT[wave]);
            acc = ref_add_rows(acc, sT[wave]);
        }
        total += ref_finalize(acc, 256, sP[wave]);
    }
    if (total == 12345.678f) out[tid] = total;
}

// every fp32 a in [-104, 0] (sign bit set, magnitude bits 0 .. bits(104.0f)): fast == library?
__global__ __launch_bounds__(256) void exp_check_kernel(unsigned long long *bad, unsigned long long *slow_count, unsigned int first_bad[4])
{
    const unsigned int top = __float_as_uint(104.0f);
    unsigned long long nbad = 0, nslow = 0;
    for (unsigned long long m = (unsigned long long)blockIdx.x * 256 + threadIdx.x; m <= top; m += (unsigned long long)gridDim.x * 256) {
        const float a = __uint_as_float(0x80000000u | (unsigned int)m);
        bool slow = false;
        const float f = fast_exp_rn(a, &slow);
        const float ref = (float)exp((double)a);
        nslow += slow ? 1 : 0;
        if (!slow && __float_as_uint(f) != __float_as_uint(ref)) { ++nbad; first_bad[0] = (unsigned int)m; first_bad[1] = __float_as_uint(f); first_bad[2] = __float_as_uint(ref); }
    }
    if (nbad) atomicAdd(bad, nbad);
    if (nslow) atomicAdd(slow_count, nslow);
}


// chunk sums: the narrow form against the general one, bit for bit, on smooth / noisy / NaN grays
__global__ __launch_bounds__(64) void chunk_check_kernel(const float *__restrict__ grays, int nunits, unsigned long long *bad, unsigned long long *narrow)
{
    __shared__ float sT[2][kRefUnitRows * kRefRow];
    __shared__ float sRec[kRefRecFloats];
    __shared__ float sBins[kBins];
    const int lane = threadIdx.x;
    if (lane < kBins) sBins[lane] = linspace_bin(lane);
    __syncthreads();
    for (int u = blockIdx.x; u < nunits; u += gridDim.x) {
        const float gray = grays[(size_t)u * 64 + lane];
        int j0;
        float v[kRefWin];
        ref_pixel(sBins, 0.01f, gray, j0, v);
        ref_unit_chunks(sRec, j0, v, sT[0]);
        unit_chunks_fast(sRec, j0, v, sT[1]);
        const unsigned int mask = (unsigned int)__builtin_amdgcn_readfirstlane((int)wave_or_u32(1u << j0));
        if (lane == 0 && 31 - __builtin_clz(mask) - __builtin_ctz(mask) <= 1) atomicAdd(narrow, 1ull);
        for (int k = lane; k < kRefUnitRows * kRefRow; k += 64) {
            if (k % kRefRow == kBins) continue;
            if (__float_as_uint(sT[0][k]) != __float_as_uint(sT[1][k])) atomicAdd(bad, 1ull);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

template <typename F>
static float timeit(F launch)
{
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    launch(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    for (int i = 0; i < 3; ++i) launch();
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    return ms * 1e3f / 3;
}

int main()
{
    const int nwaves = 256 * 16;
    float *grays, *out;
    (void)hipMalloc(&grays, sizeof(float) * nwaves * 256); (void)hipMalloc(&out, 4096 * 4);
    {
        static float h[256 * 16 * 256];
        // smooth 8-bit content: a gradient plus +-1 level of noise, like oracle/content_families.py smooth8
        unsigned int s = 12345u;
        for (int i = 0; i < nwaves * 256; ++i) {
            s = s * 1664525u + 1013904223u;
            const int base = 60 + (i / 256) % 120, n = (int)((s >> 16) % 3) - 1;
            h[i] = (float)(base + n) / 255.0f * 0.97f;
        }
        (void)hipMemcpy(grays, h, sizeof(h), hipMemcpyHostToDevice);
    }
    const int units = 256;
#define RUN(VAR, NT, name)                                                                                                      \
    {                                                                                                                             \
        float us = timeit([&] { hipLaunchKernelGGL((unit_kernel<VAR, NT>), dim3(256), dim3(NT), 0, 0, grays, out, units); });     \
        printf("%-34s %d waves/SIMD: %8.1f us, %6.3f us per unit per wave, %6.3f us of SIMD time per unit\n", name, NT / 256, us,  \
               us / units, us / units / (NT / 256));                                                                               \
    }
    for (int rep = 0; rep < 2; ++rep) {
        RUN(0, 512, "current")  RUN(1, 512, "closed-form bins")  RUN(2, 512, "closed-form bins + fast exp")  RUN(3, 512, "closed-form bins, no exp") RUN(4, 512, "narrow chunk sums")
        RUN(0, 1024, "current") RUN(1, 1024, "closed-form bins") RUN(2, 1024, "closed-form bins + fast exp") RUN(3, 1024, "closed-form bins, no exp") RUN(4, 1024, "narrow chunk sums")
    }
    {
        // check: the timing grays (smooth) + noise + a few NaN / out-of-range ones
        const int nu = 16384;
        float *cg;
        (void)hipMalloc(&cg, sizeof(float) * nu * 64);
        static float hc[16384 * 64];
        unsigned int s2 = 777u;
        for (int u = 0; u < nu; ++u)
            for (int l = 0; l < 64; ++l) {
                s2 = s2 * 1664525u + 1013904223u;
                const int kind = u & 3;
                float g;
                if (kind == 0) g = (float)(40 + (u >> 2) % 180 + (int)((s2 >> 16) % 3) - 1) / 255.0f;                 // smooth: +-1 level
                else if (kind == 1) g = (float)((40 + (u >> 2) % 180) + (int)((s2 >> 16) % 9) - 4) / 255.0f * 0.9f;     // +-4 levels
                else if (kind == 2) g = (float)((s2 >> 8) % 256) / 255.0f;                                             // noise
                else g = (float)(100 + (int)((s2 >> 16) % 2)) / 255.0f * ((u & 255) == 3 && l == 5 ? __builtin_nanf("") : 1.0f);
                hc[u * 64 + l] = g;
            }
        (void)hipMemcpy(cg, hc, sizeof(hc), hipMemcpyHostToDevice);
        unsigned long long *cb, hcb[2];
        (void)hipMalloc(&cb, 16); (void)hipMemset(cb, 0, 16);
        hipLaunchKernelGGL(chunk_check_kernel, dim3(1024), dim3(64), 0, 0, cg, nu, cb, cb + 1);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(hcb, cb, 16, hipMemcpyDeviceToHost);
        printf("narrow chunk sums vs the general form on %d units: %llu differing chunk sums, %llu units took the narrow path\n", nu, hcb[0], hcb[1]);
    }
    unsigned long long *bad, hb[2];
    unsigned int *fb, hfb[4] = {0, 0, 0, 0};
    (void)hipMalloc(&bad, 16); (void)hipMemset(bad, 0, 16); (void)hipMalloc(&fb, 16); (void)hipMemset(fb, 0, 16);
    hipLaunchKernelGGL(exp_check_kernel, dim3(4096), dim3(256), 0, 0, bad, bad + 1, fb);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(hb, bad, 16, hipMemcpyDeviceToHost); (void)hipMemcpy(hfb, fb, 16, hipMemcpyDeviceToHost);
    unsigned int top; { float t = 104.0f; memcpy(&top, &t, 4); }
    printf("fast exp vs (float)exp((double)a) over all %u fp32 arguments in [-104, -0]: %llu differ outside the Ziv band, %llu take the slow path (%.2e)\n",
           top + 1, hb[0], hb[1], (double)hb[1] / (top + 1.0));
    if (hb[0]) printf("  e.g. a bits 0x%08x: fast 0x%08x library 0x%08x\n", 0x80000000u | hfb[0], hfb[1], hfb[2]);
    return 0;
}
