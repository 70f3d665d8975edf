// cgic_router_dev.h -- device code of the granularity router (RouterTriple.py:15-95), shared by the
// stand-alone launch (cgic_router.hip) and the horizontally fused VQ+router launch (cgic_vq.hip).
#pragma once
#include "cgic_common.h"

#include <math.h>

namespace cgic {

constexpr int kRouterThreads = 1024;   // stand-alone launch; the VQ-fused launch runs the same body with 256

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
    unsigned int hist[3][256];      // rotating: pass p counts into [p % 3] while [(p + 1) % 3] is being cleared
};

// k-th smallest (0-based rank) of n values produced by val(i); all NT threads of the block call.
// 4 passes of 8 bits, ONE barrier per pass: every wave finds the digit holding the rank by itself from the
// finished histogram (a 256-bin scan is 4 loads + one wave scan), so there is nothing to broadcast, and the
// histogram of pass p+1 was cleared during pass p (three rotating buffers: a slow wave may still be reading
// pass p-1's while a fast one clears).  The previous 3-barriers-per-pass version cost 2.8 + 5.8 us for the two
// selects of a 256x256 image.
template <int NT, typename F>
__device__ float radix_select(F val, int64_t n, unsigned int rank0, RouterShared *sh)
{
    const int tid = threadIdx.x;
    const int lane = lane_id();
    for (int i = tid; i < 512; i += NT) (&sh->hist[0][0])[i] = 0;      // buffers 0 and 1
    __syncthreads();
    unsigned int prefix = 0, rank = rank0, himask = 0;
    int pass = 0;
#pragma unroll 1
    for (int shift = 24; shift >= 0; shift -= 8, ++pass) {
        unsigned int *h = sh->hist[pass % 3];
        // Same-address LDS atomics serialise across the whole workgroup, and entropy values crowd into 2-3 bins of a pass
        // (all of them in the exponent byte): at 9216 values per 768x768 tile a pass spent ~4 us incrementing ONE
        // counter.  Every lane therefore merges equal digits of its own consecutive slots before it touches LDS: a
        // hot pass costs one or two atomics per lane instead of n / NT, a spread-out pass the same as before.
        // (Ballot-based wave aggregation was measured no faster: its ~25 extra instructions per slot cost what it saved.)
        {
            unsigned int run_d = 0, run_n = 0;
            for (int64_t i = tid; i < n; i += NT) {
                const uint32_t key = f2key(val(i));
                if ((key & himask) != prefix) continue;
                const unsigned int digit = (key >> shift) & 0xFF;
                if (digit != run_d && run_n) { atomicAdd(&h[run_d], run_n); run_n = 0; }
                run_d = digit;
                ++run_n;
            }
            if (run_n) atomicAdd(&h[run_d], run_n);
        }
        if (pass >= 1) {
            unsigned int *hz = sh->hist[(pass + 1) % 3];
            for (int i = tid; i < 256; i += NT) hz[i] = 0;
        }
        __syncthreads();
        // every wave: lane handles 4 consecutive digits; find the digit holding `rank`
        const unsigned int c0 = h[4 * lane], c1 = h[4 * lane + 1], c2 = h[4 * lane + 2], c3 = h[4 * lane + 3];
        const unsigned int s = c0 + c1 + c2 + c3;
        const unsigned int incl = wave_inclusive_scan(s);
        const unsigned int excl = incl - s;
        const bool mine = excl <= rank && rank < incl;
        unsigned int r = rank - excl, d = 4 * lane;
        if (r >= c0) { r -= c0; ++d; if (r >= c1) { r -= c1; ++d; if (r >= c2) { r -= c2; ++d; } } }
        const unsigned long long who = __ballot(mine);
        const int src = who ? __builtin_ctzll(who) : 0;            // (rank >= n cannot happen: checked on the host)
        d = __shfl(d, src, kWave);
        r = __shfl(r, src, kWave);
        prefix |= d << shift;
        rank = r;
        himask |= 0xFFu << shift;
    }
    __syncthreads();   // all waves are done with the histograms before a later call clears them
    return key2f(prefix);
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
    int stage;             // 1: the segment's e16/e8 are copied to LDS once (all select passes read LDS); 2: only e8 (router_lds_bytes)
    int bands;             // workgroups per segment (per-image segments only): every one finds the thresholds, each writes
                           // only its band of rows of the masks (one CU per 768x768 tile spent 10 us writing 200 KB of masks)
};

// The whole router for segment `seg`, executed by a block of NT threads; `dyn` = dynamic LDS of at least
// router_lds_bytes() bytes.
template <int NT>
__device__ __forceinline__ void router_body(const RouterArgs &a, int64_t blk, unsigned char *dyn)
{
    const int nb = a.bands > 1 ? a.bands : 1;
    const int64_t seg = blk / nb;
    const int band = (int)(blk - seg * nb);
    RouterShared *sh = reinterpret_cast<RouterShared *>(dyn);
    unsigned long long *gc_bits = reinterpret_cast<unsigned long long *>(dyn + 3072);  // [ceil(N16/64)]

    const int tid = threadIdx.x;
    const int lane = lane_id();
    CGIC_STAMP(0);
    const int64_t h16 = a.h16, w16 = a.w16, h8 = 2 * h16, w8 = 2 * w16, h4 = 4 * h16, w4 = 4 * w16;
    const int64_t n16 = h16 * w16, n8 = h8 * w8, n4 = h4 * w4;
    const int64_t N16 = a.per * n16, N8 = a.per * n8, N4 = a.per * n4;
    const float *e16 = a.e16 + seg * N16;
    const float *e8 = a.e8 + seg * N8;
    if (a.stage) {
        // one round trip to HBM/L2 instead of one per radix pass (8 passes + 3 elementwise sweeps)
        float *l16 = reinterpret_cast<float *>(gc_bits + ((N16 + 63) >> 6));
        float *l8 = a.stage == 1 ? l16 + N16 : l16;               // stage 2: e16 stays in global memory
        if (a.stage == 1) {
            for (int64_t i = tid; i < N16; i += NT) l16[i] = e16[i];
            e16 = l16;
        }
        for (int64_t i = tid; i < N8; i += NT) l8[i] = e8[i];
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
    if (has_thr_c) thr_c = radix_select<NT>([&](int64_t i) { return e16[i]; }, N16, a.rank_c, sh);
    CGIC_STAMP(2);
    const int64_t N16r = (N16 + 63) & ~(int64_t)63;
    for (int64_t i = tid; i < N16r; i += NT) {
        bool g = false;
        if (i < N16) g = has_thr_c ? (e16[i] < thr_c) : (mode == 4);
        unsigned long long bal = __ballot(g);
        if (lane == 0) gc_bits[i >> 6] = bal;
        if (i < N16 && band == 0) mc[i] = g ? 1 : 0;
    }
    __syncthreads();
    // 32-bit index math throughout (N8 < 2^31 is checked on the host): a 64-bit divide is ~100 instructions
    const int n8i = (int)n8, w8i = (int)w8, n16i = (int)n16, w16i = (int)w16;
    auto gc_at = [&](int b, int y8, int x8) -> bool {   // coarse gate of the parent of medium element (y8, x8) of image b
        const int c = b * n16i + (y8 >> 1) * w16i + (x8 >> 1);
        return (gc_bits[c >> 6] >> (c & 63)) & 1ull;
    };
    auto gc_of8 = [&](int64_t i) -> bool {
        const int ii = (int)i;
        const int b = ii / n8i, r = ii - b * n8i;
        const int y = r / w8i, x = r - y * w8i;
        return gc_at(b, y, x);
    };
    // One image per segment (per-image routing: every batched compress): walk rows by wave and columns by lane -- the
    // flat loops below spend ~4 integer divisions (~25 VALU instructions each) per element, which at 9216 + 36864
    // elements per 768x768 tile on ONE CU was 11 of the router's 39 us.
    const bool rows2d = a.per == 1 && (w8 >= 64 || nb > 1);      // (narrow rows leave most lanes of a wave idle: flat loops there)
    constexpr int NWV = NT / 64;
    const int wv = tid >> 6;
    // this workgroup's band of coarse rows [cy0, cy1) (medium rows x2, fine rows x4); bands > 1 only with rows2d
    const int cper = ((int)h16 + nb - 1) / nb;
    const int cy0 = band * cper < (int)h16 ? band * cper : (int)h16;
    const int cy1 = cy0 + cper < (int)h16 ? cy0 + cper : (int)h16;

    CGIC_STAMP(3);
    // ---- medium gate
    float thr_m = 0.f;
    if (mode == 0) {      // :27-31: sort e8 * (1 - up2(gate_coarse))
        if (a.stage) {
            // materialise the masked values once (LDS), so the four radix passes are plain LDS sweeps
            float *l8m = const_cast<float *>(e8) + N8;
            if (rows2d) {
                for (int y = wv; y < (int)h8; y += NWV)
                    for (int x = lane; x < w8i; x += 64) l8m[y * w8i + x] = e8[y * w8i + x] * (1.0f - (gc_at(0, y, x) ? 1.0f : 0.0f));
            } else {
                for (int64_t i = tid; i < N8; i += NT) l8m[i] = e8[i] * (1.0f - (gc_of8(i) ? 1.0f : 0.0f));
            }
            __syncthreads();
            thr_m = radix_select<NT>([&](int64_t i) { return l8m[i]; }, N8, a.rank_m, sh);
        } else {
            thr_m = radix_select<NT>([&](int64_t i) { return e8[i] * (1.0f - (gc_of8(i) ? 1.0f : 0.0f)); }, N8, a.rank_m, sh);
        }
    }
    if (mode == 1)        // :40-43
        thr_m = radix_select<NT>([&](int64_t i) { return e8[i]; }, N8, a.rank_m, sh);
    auto gm_rule = [&](float v, bool gc) -> bool {
        switch (mode) {
        case 0: return (v < thr_m) && !gc;                  // :32
        case 1: return v < thr_m;                           // :44
        case 3: return !gc;                                 // :68
        case 5: return true;                                // :81
        default: return false;
        }
    };
    auto gm_of8 = [&](int64_t i) -> bool { return gm_rule(e8[i], (mode == 0 || mode == 3) ? gc_of8(i) : false); };
    CGIC_STAMP(4);
    if (rows2d) {
        for (int y = 2 * cy0 + wv; y < 2 * cy1; y += NWV)
            for (int x = lane; x < w8i; x += 64)
                mm[y * w8i + x] = gm_rule(e8[y * w8i + x], (mode == 0 || mode == 3) ? gc_at(0, y, x) : false) ? 1 : 0;
    } else {
        for (int64_t i = tid; i < N8; i += NT) mm[i] = gm_of8(i) ? 1 : 0;
    }
    CGIC_STAMP(5);

    // ---- fine gate + optional gate tensor (:34,47,58,69,77,83,87,93): 4 consecutive x per thread
    // (w4 is a multiple of 4, so a quad never straddles a row, a medium pair or a coarse cell)
    float *gate = a.gate ? a.gate + seg * N4 * 3 : nullptr;
    const int W4 = (int)w4, W8 = (int)w8, W16 = (int)w16, NQ = (int)(N4 >> 2), n4i = (int)n4, qrow = W4 >> 2;
    auto fine_quad = [&](int b, int y, int x) {
        const int i = b * n4i + y * W4 + x;
        const int64_t c = (int64_t)b * n16 + (y >> 2) * W16 + (x >> 2);
        const bool gc = (gc_bits[c >> 6] >> (c & 63)) & 1ull;
        const int64_t m0 = (int64_t)b * n8 + (y >> 1) * W8 + (x >> 1);
        const bool gcm = (mode == 0 || mode == 3) ? gc : false;          // (the medium pair's coarse parent is this quad's)
        const bool gm0 = gm_rule(e8[m0], gcm), gm1 = gm_rule(e8[m0 + 1], gcm);
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
    };
    if (rows2d) {
        for (int y = 4 * cy0 + wv; y < 4 * cy1; y += NWV)
            for (int xq = lane; xq < qrow; xq += 64) fine_quad(0, y, xq << 2);
    } else {
        for (int q = tid; q < NQ; q += NT) {
            const int i = q << 2;
            const int b = i / n4i, r = i - b * n4i;
            const int y = r / W4, x = r - y * W4;
            fine_quad(b, y, x);
        }
    }
    CGIC_STAMP(6);
}


// stage 1: e16, e8 and the masked copy of e8 live in LDS; stage 2: only e8 and its masked copy (e16 is read from global memory
// by the coarse select's four passes); 0: nothing staged.  `budget`: the fused VQ + router launch keeps a router workgroup
// under half a CU's LDS so that it can share the CU with a VQ workgroup (a 768x768 tile: 86 KB full, 77 KB at stage 2).
__host__ __device__ inline size_t router_lds_bytes(int64_t N16, int64_t N8, int *stage, size_t budget = 96 * 1024)
{
    size_t lds = 3072 + 8 * (size_t)((N16 + 63) / 64);
    int st = 0;
    if (lds + 4 * (size_t)(N16 + 2 * N8) <= budget) { st = 1; lds += 4 * (size_t)(N16 + 2 * N8); }
    else if (lds + 4 * (size_t)(2 * N8) <= budget) { st = 2; lds += 4 * (size_t)(2 * N8); }
    if (stage) *stage = st;
    return lds;
}

// host-side argument preparation shared by the stand-alone and the VQ-fused launch
int router_prepare(const float *e16, const float *e8, int64_t B, int64_t h16, int64_t w16, double c_ratio,
                   double m_ratio, int per_image, int32_t *mask_c, int32_t *mask_m, int32_t *mask_f, float *gate,
                   RouterArgs *out, int64_t *nseg, size_t *lds, size_t lds_budget = 96 * 1024);

}  // namespace cgic
