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

// the pixels behind the maps, checked (cgic_pixels of the C ABI -> RefineSrc of the kernels)
int refine_source(const cgic_pixels *refine, int64_t h16, int64_t w16, RefineSrc *out)
{
    CGIC_REQUIRE(refine->bins && refine->nbins == kBins, CGIC_ERR_UNSUPPORTED, "router: refinement needs the 32 bin centres (model.py:480)");
    CGIC_REQUIRE(refine->sigma > 0.f && refine->sigma <= 0.0105f, CGIC_ERR_UNSUPPORTED,
                 "router: sigma=%g; the five-bin window assumes the reference's sigma=0.01 (model.py:481)", refine->sigma);
    for (int i = 0; i < kBins; ++i)        // the kernel recomputes them (24-byte RefineSrc): they must be THE linspace, to the bit
        CGIC_REQUIRE(refine->bins[i] == linspace_bin(i), CGIC_ERR_UNSUPPORTED,
                     "router: bins[%d]=%.9g is not torch.linspace(-1, 1, 32)[%d]=%.9g", i, refine->bins[i], i, linspace_bin(i));
    CGIC_REQUIRE(!refine->is_u8 || ((uintptr_t)refine->x & 3u) == 0, CGIC_ERR_INVALID, "router: the uint8 frame must be 4-byte aligned");
    out->x = refine->x;
    out->u8 = refine->is_u8 ? 1 : 0;
    CGIC_REQUIRE(16 * h16 < ((int64_t)1 << 30) && 16 * w16 < ((int64_t)1 << 30), CGIC_ERR_UNSUPPORTED, "router: image too large");
    out->H = (int)(16 * h16);
    out->W = (int)(16 * w16);
    out->sigma = refine->sigma;
    out->flat8 = refine->flat8;
    return CGIC_OK;
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
        int rc = refine_source(refine, h16, w16, &a.rf);
        if (rc) return rc;
    }
    // (with refinement every segment must fit the FUSED launch's budget, so that the stand-alone and the fused launch accept
    // the same shapes)
    const size_t lds = router_lds_bytes(N16, N8, &a.stage, a.rf.x ? (lds_budget < kRouterFusedLds ? lds_budget : kRouterFusedLds) : lds_budget,
                                        a.rf.x != nullptr);
    CGIC_REQUIRE(a.stage >= 0, CGIC_ERR_UNSUPPORTED,
                 "router: a segment of %lld + %lld patches does not fit the workgroup's LDS: its refinement is a chain of launches "
                 "(cgic_router_refine_in_lds == 0), which has no recorded form inside a launch group",
                 (long long)N16, (long long)N8);
    CGIC_REQUIRE(!a.rf.x || N8 <= 64 * (int64_t)kRefBitWords, CGIC_ERR_UNSUPPORTED, "router: refinement of a segment of %lld patches", (long long)N8);
    // large per-image segments: several workgroups per image share the mask writing (every one repeats the selects, which
    // costs nothing while most CUs are idle): up to 8, while the launch stays within ~a quarter of the chip
    // (The row bands of a tile that split a threshold band WAIT for each other (refine_select's exchange()): every band of every launch
    // in flight has to be resident.  nseg x bands <= 64 workgroups per launch = a quarter of the chip's CUs, two such workgroups fit a
    // CU: up to FOUR launches in flight -- the pipeline's four hardware queues -- are resident together whatever else runs; more
    // concurrent launches than that are outside the contract of cgic_pixels.scratch, see include/cgic_hip.h.)
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


// ---- threshold-band refinement of segments that do not fit a workgroup's LDS ---------------------------------------------------
// The reference routes over the FLATTENED batch (RouterTriple.py:21,40,52,63: encode() of B images is one segment of B x 256 +
// B x 1024 entropies) and an untiled image beyond 768x768 is one segment too: the maps of such a segment stay in global memory,
// and the refinement of cgic_router_dev.h (patch the LDS copy, select again) becomes a chain of launches over patched COPIES of
// the maps in the caller's scratch -- same argument, same band (2 refine_delta around the approximate threshold), same
// arithmetic for the band's patches (cgic_entropy_dev.h), hence the same masks as routing on cgic_entropy_maps_ref_f32's maps:
//   1. big_select<0>   one workgroup per segment: approximate coarse threshold            -> thr[seg][0]
//   2. big_patch<16>   every wave of the chip: e16x = e16, band members re-evaluated from their pixels
//   3. big_select<1>   exact coarse threshold on e16x -> thr[seg][1]; approximate medium threshold on e8 (masked by the exact
//                      coarse gate in mode 0)                                               -> thr[seg][2]
//   4. big_patch<8>    e8x = e8, ungated band members re-evaluated
//   5. router_kernel   the ordinary (unstaged) router on (e16x, e8x): thresholds by selection, masks, gate
// Latency is that of five dependent launches (~0.1 ms for 64 images of 256x256): the path of the reference's default batch
// semantics, not of the timed per-image step.
struct BigArgs {
    const float *e16, *e8;
    float *e16x, *e8x, *thr;        // thr: [nseg][4]
    int64_t per, h16, w16, nseg;
    int mode;
    unsigned int rank_c, rank_m;
    RefineSrc rf;
};

template <int PHASE>
__global__ __launch_bounds__(kRouterThreads) void big_select_kernel(BigArgs a)
{
    __shared__ RouterShared sh;
    const int tid = threadIdx.x;
    const int64_t seg = blockIdx.x;
    const int64_t n16 = a.h16 * a.w16, N16 = a.per * n16, N8 = 4 * N16;
    const int64_t w16 = a.w16, w8 = 2 * a.w16, n8 = 4 * n16;
    for (int i = tid; i < 256; i += kRouterThreads) sh.hist[0][i] = 0;
    __syncthreads();
    float *thr = a.thr + 4 * seg;
    const bool has_c = a.mode == 0 || a.mode == 2 || a.mode == 3;
    if (PHASE == 0) {
        const float *e16 = a.e16 + seg * N16;
        const float t = radix_select<kRouterThreads>([&](int64_t i) { return e16[i]; }, N16, a.rank_c, &sh);
        if (tid == 0) thr[0] = t;
        return;
    }
    float thr_c = 0.f;
    if (has_c) {
        const float *e16x = a.e16x + seg * N16;
        thr_c = radix_select<kRouterThreads>([&](int64_t i) { return e16x[i]; }, N16, a.rank_c, &sh);
        if (tid == 0) thr[1] = thr_c;
    }
    if (a.mode == 0 || a.mode == 1) {
        const float *e8 = a.e8 + seg * N8;
        const float *e16x = a.e16x + seg * N16;
        const bool masked = a.mode == 0;
        const float t = radix_select<kRouterThreads>([&](int64_t i) {
            float v = e8[i];
            if (masked) {       // RouterTriple.py:27-29: e8 * (1 - up2(gate_coarse))
                const int64_t b = i / n8, r = i - b * n8, y = r / w8, x = r - y * w8;
                const bool gc = e16x[b * n16 + (y >> 1) * w16 + (x >> 1)] < thr_c;
                v = v * (1.0f - (gc ? 1.0f : 0.0f));
            }
            return v; }, N8, a.rank_m, &sh);
        if (tid == 0) thr[2] = t;
    }
}

constexpr int kBigPatchThreads = 256;
template <int P>
__global__ __launch_bounds__(kBigPatchThreads) void big_patch_kernel(BigArgs a)
{
    constexpr int NW = kBigPatchThreads / 64;
    __shared__ float sT[NW][kRefUnitRows * kRefRow];
    __shared__ float sRec[NW][kRefRecFloats];
    __shared__ float sP[NW][2 * kBins];
    __shared__ float sBins[kBins];
    const int tid = threadIdx.x, lane = lane_id(), wave = tid >> 6;
    if (tid < kBins) sBins[tid] = linspace_bin(tid);
    __syncthreads();
    const int64_t n16 = a.h16 * a.w16, N16 = a.per * n16;
    const int64_t nP = P == 16 ? n16 : 4 * n16, NP = a.per * nP;           // patches per image / per segment at this granularity
    const int wP = (int)(P == 16 ? a.w16 : 2 * a.w16);
    const int64_t total = a.nseg * NP, nchunk = (total + 63) >> 6;
    const float *src = P == 16 ? a.e16 : a.e8;
    float *dst = P == 16 ? a.e16x : a.e8x;
    for (int64_t ch = (int64_t)blockIdx.x * NW + wave; ch < nchunk; ch += (int64_t)gridDim.x * NW) {
        const int64_t i = ch * 64 + lane;
        float v = 0.f;
        bool in = false;
        int64_t seg = 0, e = 0;
        if (i < total) {
            v = src[i];
            seg = i / NP;
            e = i - seg * NP;
            const float t = a.thr[4 * seg + (P == 16 ? 0 : 2)];
            const float w = 2.f * refine_delta(t + kRefineBand);
            in = fabsf(v - t) <= w;                                         // (NaN: false)
            if (P == 8 && a.mode == 0 && in) {
                // a gated element's masked value is an exact 0: never re-evaluated (refine_select's ExactGate)
                const int64_t b = e / nP, r = e - b * nP, y = r / wP, x = r - y * wP;
                in = !(a.e16x[seg * N16 + b * n16 + (y >> 1) * a.w16 + (x >> 1)] < a.thr[4 * seg + 1]);
            }
        }
        unsigned long long todo = __ballot(in);
        while (todo) {
            const int m = __builtin_ctzll(todo);
            todo &= todo - 1;
            const int64_t sg = __shfl(seg, m, kWave), ee = __shfl(e, m, kWave);
            float ent;
            if (P == 8) {
                refine_unit<8>(a.rf, sBins, sg * a.per, wP, (int)nP, (int)ee, 0, sRec[wave], sT[wave]);
                ent = ref_finalize(ref_add_rows(0.f, sT[wave]), 64, sP[wave]);
            } else {
                float acc = 0.f;
                for (int q = 0; q < 4; ++q) {
                    refine_unit<16>(a.rf, sBins, sg * a.per, wP, (int)nP, (int)ee, q, sRec[wave], sT[wave]);
                    acc = ref_add_rows(acc, sT[wave]);
                }
                ent = ref_finalize(acc, 256, sP[wave]);
            }
            if (lane == m) v = ent;
        }
        if (i < total) dst[i] = v;
    }
}

size_t router_big_scratch_bytes(int64_t B, int64_t h16, int64_t w16, int per_image)
{
    const int64_t nseg = per_image ? B : 1;
    return 20 * (size_t)(B * h16 * w16) + 16 * (size_t)nseg + 64;
}

bool router_refine_in_lds(int64_t B, int64_t h16, int64_t w16, int per_image)
{
    const int64_t N16 = (per_image ? 1 : B) * h16 * w16;
    int st = 0;
    router_lds_bytes(N16, 4 * N16, &st, kRouterFusedLds, true);
    return st == 1 && 4 * N16 <= 64 * (int64_t)kRefBitWords;
}

int router_big(const float *e16, const float *e8, int64_t B, int64_t h16, int64_t w16, double c_ratio, double m_ratio, int per_image,
               int32_t *mask_c, int32_t *mask_m, int32_t *mask_f, float *gate, const cgic_pixels *refine, hipStream_t stream)
{
    CGIC_REQUIRE(!group_recording(), CGIC_ERR_UNSUPPORTED,
                 "router: the refinement of a segment that does not fit the LDS is a chain of launches (cgic_router_refine_in_lds == 0): not inside a launch group");
    const size_t need = router_big_scratch_bytes(B, h16, w16, per_image);
    CGIC_REQUIRE(refine->scratch && refine->scratch_bytes >= need, CGIC_ERR_INVALID,
                 "router: a routing segment of %lld x %lldx%lld pixels is refined through patched copies of the maps: cgic_pixels.scratch of %zu bytes needed "
                 "(cgic_router_refine_scratch_bytes), %zu given", (long long)(per_image ? 1 : B), (long long)(16 * h16), (long long)(16 * w16), need,
                 refine->scratch ? refine->scratch_bytes : (size_t)0);
    CGIC_REQUIRE(((uintptr_t)refine->scratch & 15u) == 0, CGIC_ERR_INVALID, "router: the refinement scratch must be 16-byte aligned");
    const int64_t T16 = B * h16 * w16, T8 = 4 * T16;
    float *e16x = reinterpret_cast<float *>(refine->scratch), *e8x = e16x + T16, *thr = e8x + T8;
    // the final launch's arguments (ranks, mode, index magic) -- made first: it validates the shapes and ratios
    RouterArgs fa;
    int64_t fseg;
    size_t flds;
    int rc = router_prepare(e16x, e8x, B, h16, w16, c_ratio, m_ratio, per_image, mask_c, mask_m, mask_f, gate, &fa, &fseg, &flds, 96 * 1024, nullptr, stream, false);
    if (rc) return rc;
    BigArgs a;
    a.e16 = e16; a.e8 = e8; a.e16x = e16x; a.e8x = e8x; a.thr = thr;
    a.per = per_image ? 1 : B; a.h16 = h16; a.w16 = w16; a.nseg = per_image ? B : 1;
    a.mode = fa.mode; a.rank_c = fa.rank_c; a.rank_m = fa.rank_m;
    rc = refine_source(refine, h16, w16, &a.rf);
    if (rc) return rc;
    const bool has_c = a.mode == 0 || a.mode == 2 || a.mode == 3, has_m = a.mode == 0 || a.mode == 1;
    auto patch_grid = [&](int64_t total) { int64_t g = (total + 255) / 256; return dim3((unsigned)(g < 1 ? 1 : g > 2048 ? 2048 : g)); };
    if (has_c) {
        hipLaunchKernelGGL(big_select_kernel<0>, dim3((unsigned)a.nseg), dim3(kRouterThreads), 0, stream, a);
        rc = launch_check("big_select_kernel<0>");
        if (rc) return rc;
        hipLaunchKernelGGL(big_patch_kernel<16>, patch_grid(T16), dim3(kBigPatchThreads), 0, stream, a);
        rc = launch_check("big_patch_kernel<16>");
        if (rc) return rc;
    } else {
        CGIC_HIP_TRY(hipMemcpyAsync(e16x, e16, sizeof(float) * (size_t)T16, hipMemcpyDeviceToDevice, stream));
    }
    if (has_m) {
        hipLaunchKernelGGL(big_select_kernel<1>, dim3((unsigned)a.nseg), dim3(kRouterThreads), 0, stream, a);
        rc = launch_check("big_select_kernel<1>");
        if (rc) return rc;
        hipLaunchKernelGGL(big_patch_kernel<8>, patch_grid(T8), dim3(kBigPatchThreads), 0, stream, a);
        rc = launch_check("big_patch_kernel<8>");
        if (rc) return rc;
    } else {
        CGIC_HIP_TRY(hipMemcpyAsync(e8x, e8, sizeof(float) * (size_t)T8, hipMemcpyDeviceToDevice, stream));
    }
    if (flds > 64 * 1024) { rc = ensure_dynamic_lds((const void *)router_kernel, flds); if (rc) return rc; }
    hipLaunchKernelGGL(router_kernel, dim3((unsigned)fseg), dim3(kRouterThreads), flds, stream, fa);
    return launch_check("router_kernel");
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
    // (ABI 8: every segment is refined -- in the router's workgroup when its maps fit the LDS, else through patched copies)
    if (B <= 0 || h16 <= 0 || w16 <= 0) return 0;
    const int64_t N16 = (per_image ? 1 : B) * h16 * w16;
    return 4 * N16 < ((int64_t)1 << 31) ? 1 : 0;
}

extern "C" int cgic_router_refine_in_lds(int64_t B, int64_t h16, int64_t w16, int per_image)
{
    if (B <= 0 || h16 <= 0 || w16 <= 0) return 0;
    return router_refine_in_lds(B, h16, w16, per_image) ? 1 : 0;
}

extern "C" size_t cgic_router_refine_scratch_bytes(int64_t B, int64_t h16, int64_t w16, int per_image)
{
    if (!cgic_router_refine_supported(B, h16, w16, per_image)) return 0;
    if (!router_refine_in_lds(B, h16, w16, per_image)) return router_big_scratch_bytes(B, h16, w16, per_image);      // REQUIRED there
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
    CGIC_REQUIRE(h16 > 0 && w16 > 0, CGIC_ERR_INVALID, "router: bad shape");
    if (refine && refine->x && cgic_router_mode(c_ratio, m_ratio) <= 3 && !router_refine_in_lds(B, h16, w16, per_image))
        return router_big(e16, e8, B, h16, w16, c_ratio, m_ratio, per_image, mask_c, mask_m, mask_f, gate, refine, (hipStream_t)stream);
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
