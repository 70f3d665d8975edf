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
#include "cgic_router_dev.h"

namespace cgic {

__global__ __launch_bounds__(kRouterThreads) void router_kernel(RouterArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
    router_body<kRouterThreads, true>(a, blockIdx.x, dyn);
}

int router_prepare(const float *e16, const float *e8, int64_t B, int64_t h16, int64_t w16, double c_ratio,
                   double m_ratio, int per_image, int32_t *mask_c, int32_t *mask_m, int32_t *mask_f, float *gate,
                   RouterArgs *out, int64_t *nseg_out, size_t *lds_out, size_t lds_budget, const cgic_pixels *refine, hipStream_t stream, bool queues)
{
    CGIC_REQUIRE(e16 && e8 && mask_c && mask_m && mask_f, CGIC_ERR_INVALID, "router: NULL tensor");
    CGIC_REQUIRE(B > 0 && h16 > 0 && w16 > 0, CGIC_ERR_INVALID, "router: bad shape");
    const int mode = cgic_router_mode(c_ratio, m_ratio);
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
    {
        // magic multipliers of the index divisions (cgic_router_dev.h: fdiv): exact while (largest dividend) x (divisor) < 2^32
        const int64_t n8 = 4 * h16 * w16, n4 = 16 * h16 * w16, w8 = 2 * w16, w4 = 4 * w16;
        auto magic = [](int64_t nmax, int64_t d) -> unsigned int {
            return (d > 1 && nmax * d < ((int64_t)1 << 32)) ? (unsigned int)((((uint64_t)1 << 32) + (uint64_t)d - 1) / (uint64_t)d) : 0u;
        };
        a.mg_n8 = magic(N8, n8); a.mg_w8 = magic(n8, w8);
        a.mg_n4 = magic(4 * N8, n4); a.mg_w4 = magic(n4, w4);
    }
    a.rf.x = nullptr;
    a.rq.hdr = nullptr; a.rq.board = nullptr; a.rq.scratch = nullptr; a.rq.nq = 0; a.rq.pad = 0;
    if (refine && refine->x && (mode <= 3)) {          // (modes 4-6 compare nothing)
        CGIC_REQUIRE(refine->bins && refine->nbins == kBins, CGIC_ERR_UNSUPPORTED, "router: refinement needs the 32 bin centres (model.py:480)");
        CGIC_REQUIRE(refine->sigma > 0.f && refine->sigma <= 0.0105f, CGIC_ERR_UNSUPPORTED,
                     "router: sigma=%g; the five-bin window assumes the reference's sigma=0.01 (model.py:481)", refine->sigma);
        for (int i = 0; i < kBins; ++i)        // the kernel recomputes them (24-byte RefineSrc): they must be THE linspace, to the bit
            CGIC_REQUIRE(refine->bins[i] == linspace_bin(i), CGIC_ERR_UNSUPPORTED,
                         "router: bins[%d]=%.9g is not torch.linspace(-1, 1, 32)[%d]=%.9g", i, refine->bins[i], i, linspace_bin(i));
        CGIC_REQUIRE(!refine->is_u8 || ((uintptr_t)refine->x & 3u) == 0, CGIC_ERR_INVALID, "router: the uint8 frame must be 4-byte aligned");
        a.rf.x = refine->x;
        a.rf.u8 = refine->is_u8 ? 1 : 0;
        CGIC_REQUIRE(16 * h16 < ((int64_t)1 << 30) && 16 * w16 < ((int64_t)1 << 30), CGIC_ERR_UNSUPPORTED, "router: image too large");
        a.rf.H = (int)(16 * h16);
        a.rf.W = (int)(16 * w16);
        a.rf.sigma = refine->sigma;
        a.rf.flat8 = refine->flat8;
    }
    // (with refinement every segment must fit the FUSED launch's budget, so that the stand-alone and the fused launch accept
    // the same shapes)
    const size_t lds = router_lds_bytes(N16, N8, &a.stage, a.rf.x ? (lds_budget < kRouterFusedLds ? lds_budget : kRouterFusedLds) : lds_budget,
                                        a.rf.x != nullptr);
    CGIC_REQUIRE(a.stage >= 0, CGIC_ERR_UNSUPPORTED,
                 "router: threshold refinement needs a segment whose maps fit the workgroup's LDS (%lld + %lld patches here; "
                 "cgic_router_refine_supported): route per image / per tile of at most 768x768, or pass no pixels",
                 (long long)N16, (long long)N8);
    CGIC_REQUIRE(!a.rf.x || N8 <= 64 * (int64_t)kRefBitWords, CGIC_ERR_UNSUPPORTED, "router: refinement of a segment of %lld patches", (long long)N8);
    // large per-image segments: several workgroups per image share the mask writing (every one repeats the selects, which
    // costs nothing while most CUs are idle): up to 8, while the launch stays within ~a quarter of the chip
    a.bands = 1;
    if (per_image && h16 * w16 >= 32 * 32) {
        int64_t nb = 64 / nseg;
        if (nb > 8) nb = 8;
        if (nb > h16) nb = h16;
        a.bands = nb >= 2 ? (int)nb : 1;
    }
    CGIC_REQUIRE(lds <= 150 * 1024, CGIC_ERR_UNSUPPORTED, "router: segment of %lld coarse patches exceeds LDS", (long long)N16);
    // the launch's refinement queues: one header per (segment, select) + the board, in library-owned slots (a pool of their own); payload in the caller's scratch
    // (the fused launch, queues == false, uses headers + scratch only for the row bands' exchange: segments with bands > 1)
    if ((queues || a.bands > 1) && a.rf.x && refine->scratch && 2 * nseg <= 4096) {
        const size_t need = (size_t)nseg * refine_scratch_bytes_per_segment(N16, N8);
        CGIC_REQUIRE(refine->scratch_bytes >= need, CGIC_ERR_INVALID, "router: refinement scratch of %zu bytes, %zu needed (cgic_router_refine_scratch_bytes)",
                     refine->scratch_bytes, need);
        CGIC_REQUIRE(((uintptr_t)refine->scratch & 15u) == 0, CGIC_ERR_INVALID, "router: the refinement scratch must be 16-byte aligned");
        int rc = acquire_tickets(stream, (int)(2 * nseg), &a.rq.hdr, 1);
        if (rc) return rc;
        rc = acquire_tickets(stream, 1, &a.rq.board, 0);
        if (rc) return rc;
        a.rq.scratch = reinterpret_cast<unsigned char *>(refine->scratch);
        a.rq.nq = (unsigned int)(2 * nseg);
    }
    *out = a; *nseg_out = nseg * a.bands; *lds_out = lds;       // workgroups of the launch
    return CGIC_OK;
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

extern "C" int cgic_router_refine_supported(int64_t B, int64_t h16, int64_t w16, int per_image)
{
    if (B <= 0 || h16 <= 0 || w16 <= 0) return 0;
    const int64_t N16 = (per_image ? 1 : B) * h16 * w16;
    int st = 0;
    router_lds_bytes(N16, 4 * N16, &st, kRouterFusedLds, true);
    return st == 1 ? 1 : 0;
}

extern "C" size_t cgic_router_refine_scratch_bytes(int64_t B, int64_t h16, int64_t w16, int per_image)
{
    if (!cgic_router_refine_supported(B, h16, w16, per_image)) return 0;
    const int64_t nseg = per_image ? B : 1, N16 = (per_image ? 1 : B) * h16 * w16;
    if (2 * nseg > 4096) return 0;
    return (size_t)nseg * refine_scratch_bytes_per_segment(N16, 4 * N16);
}

extern "C" int cgic_router_f32(const float *e16, const float *e8, int64_t B, int64_t h16, int64_t w16,
                               double c_ratio, double m_ratio, int per_image, int32_t *mask_c,
                               int32_t *mask_m, int32_t *mask_f, float *gate, int *mode_out,
                               const cgic_pixels *refine, cgic_stream_t stream)
{
    CGIC_NOT_IN_GROUP("cgic_router_f32");
    if (mode_out) *mode_out = cgic_router_mode(c_ratio, m_ratio);
    CGIC_REQUIRE(B >= 0, CGIC_ERR_INVALID, "router: bad shape");
    if (B == 0) return CGIC_OK;
    RouterArgs a;
    int64_t nseg;
    size_t lds;
    int rc = router_prepare(e16, e8, B, h16, w16, c_ratio, m_ratio, per_image, mask_c, mask_m, mask_f, gate, &a, &nseg, &lds,
                            96 * 1024, refine, (hipStream_t)stream, true);
    if (rc) return rc;
    if (lds > 64 * 1024)
        { int rc_ = ensure_dynamic_lds((const void *)router_kernel, (size_t)lds); if (rc_) return rc_; }
    hipLaunchKernelGGL(router_kernel, dim3((unsigned)nseg), dim3(kRouterThreads), lds, (hipStream_t)stream, a);
    return launch_check("router_kernel");
}
