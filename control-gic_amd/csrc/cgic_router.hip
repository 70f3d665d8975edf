// cgic_router.hip -- TripleGrainFixedEntropyRouter.forward
// (reference: CGIC/modules/vqvae/RouterTriple.py:15-95).
//
// The reference does two full torch.sort()s plus ~15 small elementwise
// launches per call just to read ONE order statistic per granularity
// (sorted[k-1]).  Here one 1024-thread block per segment (an image, or the whole
// batch for the reference's flatten-across-batch semantics) finds each
// threshold with a 4-pass radix select on order-preserving keys (LDS histogram,
// no sort, no temp arrays), keeps the coarse gate as a bitset in LDS, and writes
// the three int32 masks (+ the optional fp32 gate tensor) in the same launch.
//
// Exactness: thresholds are the exact k-th smallest fp32 values (NaN last, like
// torch.sort), comparisons are strict '<' on the original values, k comes from
// the host in float64 with Python's round-half-even.  Integer/compare work only.
#include "cgic_common.h"

#include <math.h>

namespace cgic {

constexpr int kRouterThreads = 1024;

__device__ __forceinline__ uint32_t f2key(float f)
{
    if (f != f) return 0xFFFFFFFFu;                       // NaN sorts last
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key2f(uint32_t k)
{
    if (k == 0xFFFFFFFFu) return __uint_as_float(0x7FC00000u);
    uint32_t u = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
    return __uint_as_float(u);
}

struct RouterShared {
    unsigned int hist[256];
    unsigned int prefix;
    unsigned int rank;
};

// k-th smallest (0-based rank) of n values produced by val(i); all threads call.
template <typename F>
__device__ float radix_select(F val, int64_t n, unsigned int rank0, RouterShared *sh)
{
    const int tid = threadIdx.x;
    if (tid == 0) { sh->prefix = 0; sh->rank = rank0; }
    unsigned int himask = 0;
#pragma unroll 1
    for (int shift = 24; shift >= 0; shift -= 8) {
        if (tid < 256) sh->hist[tid] = 0;
        __syncthreads();
        const unsigned int prefix = sh->prefix;
        // (a wave-aggregated variant -- one ballot per distinct digit per wave -- was measured 2x
        // SLOWER than plain LDS atomics here, even though entropy values crowd into 2-3 bins of the
        // first pass: 5.3 + 10.6 us vs 2.8 + 5.8 us for the two selects of a 256x256 image)
        for (int64_t i = tid; i < n; i += kRouterThreads) {
            uint32_t key = f2key(val(i));
            if ((key & himask) == prefix) atomicAdd(&sh->hist[(key >> shift) & 0xFF], 1u);
        }
        __syncthreads();
        if (tid < kWave) {
            // lane handles 4 consecutive digits; find the digit holding `rank`
            const unsigned int rank = sh->rank;
            unsigned int c0 = sh->hist[4 * tid], c1 = sh->hist[4 * tid + 1];
            unsigned int c2 = sh->hist[4 * tid + 2], c3 = sh->hist[4 * tid + 3];
            unsigned int s = c0 + c1 + c2 + c3;
            unsigned int incl = wave_inclusive_scan(s);
            unsigned int excl = incl - s;
            if (excl <= rank && rank < incl) {
                unsigned int r = rank - excl, d = 4 * tid;
                if (r >= c0) { r -= c0; ++d; if (r >= c1) { r -= c1; ++d; if (r >= c2) { r -= c2; ++d; } } }
                sh->prefix = prefix | (d << shift);
                sh->rank = r;
            }
        }
        himask |= 0xFFu << shift;
        __syncthreads();
    }
    const float thr = key2f(sh->prefix);
    __syncthreads();   // everyone has read prefix before a later call resets it
    return thr;
}

struct RouterArgs {
    const float *e16;
    const float *e8;
    int32_t *mask_c, *mask_m, *mask_f;
    float *gate;
    int64_t per;      // images per segment
    int64_t h16, w16;
    int mode;
    unsigned int rank_c;   // 0-based rank of the coarse threshold in the segment
    unsigned int rank_m;
    int stage;             // 1: the segment's e16/e8 are copied to LDS once (all select passes read LDS)
};

__global__ __launch_bounds__(kRouterThreads) void router_kernel(RouterArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
    RouterShared *sh = reinterpret_cast<RouterShared *>(dyn);
    unsigned long long *gc_bits = reinterpret_cast<unsigned long long *>(dyn + 1040);  // [ceil(N16/64)]

    const int tid = threadIdx.x;
    const int lane = lane_id();
    CGIC_STAMP(0);
    const int64_t h16 = a.h16, w16 = a.w16, h8 = 2 * h16, w8 = 2 * w16, h4 = 4 * h16, w4 = 4 * w16;
    const int64_t n16 = h16 * w16, n8 = h8 * w8, n4 = h4 * w4;
    const int64_t N16 = a.per * n16, N8 = a.per * n8, N4 = a.per * n4;
    const int64_t seg = blockIdx.x;
    const float *e16 = a.e16 + seg * N16;
    const float *e8 = a.e8 + seg * N8;
    if (a.stage) {
        // one round trip to HBM/L2 instead of one per radix pass (8 passes + 3 elementwise sweeps)
        float *l16 = reinterpret_cast<float *>(gc_bits + ((N16 + 63) >> 6));
        float *l8 = l16 + N16;
        for (int64_t i = tid; i < N16; i += kRouterThreads) l16[i] = e16[i];
        for (int64_t i = tid; i < N8; i += kRouterThreads) l8[i] = e8[i];
        e16 = l16;
        e8 = l8;
        __syncthreads();
    }
    CGIC_STAMP(1);
    int32_t *mc = a.mask_c + seg * N16;
    int32_t *mm = a.mask_m + seg * N8;
    int32_t *mf = a.mask_f + seg * N4;
    const int mode = a.mode;
    const bool has_thr_c = mode == 0 || mode == 2 || mode == 3;

    // ---- coarse gate (RouterTriple.py:21-25 / 52-56 / 63-66)
    float thr_c = 0.f;
    if (has_thr_c) thr_c = radix_select([&](int64_t i) { return e16[i]; }, N16, a.rank_c, sh);
    CGIC_STAMP(2);
    const int64_t N16r = (N16 + 63) & ~(int64_t)63;
    for (int64_t i = tid; i < N16r; i += kRouterThreads) {
        bool g = false;
        if (i < N16) g = has_thr_c ? (e16[i] < thr_c) : (mode == 4);
        unsigned long long bal = __ballot(g);
        if (lane == 0) gc_bits[i >> 6] = bal;
        if (i < N16) mc[i] = g ? 1 : 0;
    }
    __syncthreads();
    // 32-bit index math throughout (N8 < 2^31 is checked on the host): a 64-bit divide is ~100 instructions
    const int n8i = (int)n8, w8i = (int)w8, n16i = (int)n16, w16i = (int)w16;
    auto gc_of8 = [&](int64_t i) -> bool {   // coarse gate of the parent of medium element i
        const int ii = (int)i;
        const int b = ii / n8i, r = ii - b * n8i;
        const int y = r / w8i, x = r - y * w8i;
        const int c = b * n16i + (y >> 1) * w16i + (x >> 1);
        return (gc_bits[c >> 6] >> (c & 63)) & 1ull;
    };

    CGIC_STAMP(3);
    // ---- medium gate
    float thr_m = 0.f;
    if (mode == 0) {      // :27-31: sort e8 * (1 - up2(gate_coarse))
        if (a.stage) {
            // materialise the masked values once (LDS), so the four radix passes are plain LDS sweeps
            float *l8m = const_cast<float *>(e8) + N8;
            for (int64_t i = tid; i < N8; i += kRouterThreads) l8m[i] = e8[i] * (1.0f - (gc_of8(i) ? 1.0f : 0.0f));
            __syncthreads();
            thr_m = radix_select([&](int64_t i) { return l8m[i]; }, N8, a.rank_m, sh);
        } else {
            thr_m = radix_select([&](int64_t i) { return e8[i] * (1.0f - (gc_of8(i) ? 1.0f : 0.0f)); }, N8, a.rank_m, sh);
        }
    }
    if (mode == 1)        // :40-43
        thr_m = radix_select([&](int64_t i) { return e8[i]; }, N8, a.rank_m, sh);
    auto gm_of8 = [&](int64_t i) -> bool {
        switch (mode) {
        case 0: return (e8[i] < thr_m) && !gc_of8(i);      // :32
        case 1: return e8[i] < thr_m;                       // :44
        case 3: return !gc_of8(i);                          // :68
        case 5: return true;                                // :81
        default: return false;
        }
    };
    CGIC_STAMP(4);
    for (int64_t i = tid; i < N8; i += kRouterThreads) mm[i] = gm_of8(i) ? 1 : 0;
    CGIC_STAMP(5);

    // ---- fine gate + optional gate tensor (:34,47,58,69,77,83,87,93): 4 consecutive x per thread
    // (w4 is a multiple of 4, so a quad never straddles a row, a medium pair or a coarse cell)
    float *gate = a.gate ? a.gate + seg * N4 * 3 : nullptr;
    const int W4 = (int)w4, W8 = (int)w8, W16 = (int)w16, NQ = (int)(N4 >> 2), n4i = (int)n4, qrow = W4 >> 2;
    for (int q = tid; q < NQ; q += kRouterThreads) {
        const int i = q << 2;
        const int b = i / n4i, r = i - b * n4i;
        const int y = r / W4, x = r - y * W4;
        const int64_t c = (int64_t)b * n16 + (y >> 2) * W16 + (x >> 2);
        const bool gc = (gc_bits[c >> 6] >> (c & 63)) & 1ull;
        const int64_t m0 = (int64_t)b * n8 + (y >> 1) * W8 + (x >> 1);
        const bool gm0 = gm_of8(m0), gm1 = gm_of8(m0 + 1);
        bool gf0, gf1;
        switch (mode) {
        case 0: gf0 = !gc && !gm0; gf1 = !gc && !gm1; break;
        case 1: gf0 = !gm0; gf1 = !gm1; break;
        case 2: gf0 = gf1 = !gc; break;
        case 6: gf0 = gf1 = true; break;
        default: gf0 = gf1 = false; break;
        }
        *reinterpret_cast<int4 *>(mf + i) = make_int4(gf0, gf0, gf1, gf1);
        if (gate) {
            float *row = gate + ((int64_t)b * h4 + y) * 3 * w4;
            const float c1 = gc ? 1.f : 0.f, a0 = gm0 ? 1.f : 0.f, a1 = gm1 ? 1.f : 0.f;
            *reinterpret_cast<float4 *>(row + x) = make_float4(c1, c1, c1, c1);
            *reinterpret_cast<float4 *>(row + w4 + x) = make_float4(a0, a0, a1, a1);
            *reinterpret_cast<float4 *>(row + 2 * w4 + x) = make_float4(gf0 ? 1.f : 0.f, gf0 ? 1.f : 0.f, gf1 ? 1.f : 0.f, gf1 ? 1.f : 0.f);
        }
        (void)qrow;
    }
    CGIC_STAMP(6);
}

}  // namespace cgic

using namespace cgic;

extern "C" int cgic_router_mode(double c, double m)
{
    // RouterTriple.py:13: fine = 1 - coarse - medium in float64; :19,36,72
    volatile double f = 1.0 - c - m;
    int nz = (f == 0) + (m == 0) + (c == 0);
    if (nz == 0) return 0;
    if (nz == 1) return c == 0 ? 1 : (m == 0 ? 2 : 3);
    return c != 0 ? 4 : (m != 0 ? 5 : 6);
}

extern "C" int cgic_router_f32(const float *e16, const float *e8, int64_t B, int64_t h16, int64_t w16,
                               double c_ratio, double m_ratio, int per_image, int32_t *mask_c,
                               int32_t *mask_m, int32_t *mask_f, float *gate, int *mode_out,
                               cgic_stream_t stream)
{
    CGIC_REQUIRE(e16 && e8 && mask_c && mask_m && mask_f, CGIC_ERR_INVALID, "router: NULL tensor");
    CGIC_REQUIRE(B >= 0 && h16 > 0 && w16 > 0, CGIC_ERR_INVALID, "router: bad shape");
    const int mode = cgic_router_mode(c_ratio, m_ratio);
    if (mode_out) *mode_out = mode;
    if (B == 0) return CGIC_OK;
    const int64_t per = per_image ? 1 : B;
    const int64_t nseg = per_image ? B : 1;
    const int64_t N16 = per * h16 * w16, N8 = 4 * N16;
    CGIC_REQUIRE(N8 < (int64_t)1 << 31, CGIC_ERR_UNSUPPORTED, "router: segment too large");
    // Python round() == round-half-even on the float64 product (:23,30,42,54,65)
    long k_c = 0, k_m = 0;
    if (mode == 0 || mode == 2 || mode == 3) k_c = (long)nearbyint((double)N16 * c_ratio);
    if (mode == 0) k_m = (long)nearbyint((double)(4 * N16) * c_ratio + (double)N8 * m_ratio);
    if (mode == 1) k_m = (long)nearbyint((double)N8 * m_ratio);
    CGIC_REQUIRE(k_c >= 0 && k_c <= N16 && k_m >= 0 && k_m <= N8, CGIC_ERR_INVALID,
                 "router: k out of range (k_coarse=%ld of %lld, k_medium=%ld of %lld); the reference raises IndexError",
                 k_c, (long long)N16, k_m, (long long)N8);
    RouterArgs a;
    a.e16 = e16; a.e8 = e8; a.mask_c = mask_c; a.mask_m = mask_m; a.mask_f = mask_f; a.gate = gate;
    a.per = per; a.h16 = h16; a.w16 = w16; a.mode = mode;
    a.rank_c = (unsigned int)(k_c != 0 ? k_c - 1 : 0);      // sorted[k-1 if k != 0 else k]
    a.rank_m = (unsigned int)(k_m != 0 ? k_m - 1 : 0);
    size_t lds = 1040 + 8 * (size_t)((N16 + 63) / 64);
    CGIC_REQUIRE(lds <= 150 * 1024, CGIC_ERR_UNSUPPORTED, "router: segment of %lld coarse patches exceeds LDS", (long long)N16);
    a.stage = lds + 4 * (size_t)(N16 + 2 * N8) <= 96 * 1024 ? 1 : 0;      // e16, e8 and the masked copy of e8
    if (a.stage) lds += 4 * (size_t)(N16 + 2 * N8);
    if (lds > 64 * 1024)
        CGIC_HIP_TRY(hipFuncSetAttribute((const void *)router_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(router_kernel, dim3((unsigned)nseg), dim3(kRouterThreads), lds, (hipStream_t)stream, a);
    return launch_check("router_kernel");
}
