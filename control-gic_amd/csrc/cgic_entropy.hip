// cgic_entropy.hip -- per-patch soft-histogram Shannon entropy maps for patch
// sizes 8 and 16 in ONE pass over the image (reference: Entropy,
// CGIC/models/model.py:433-483; CGIC.encode runs it twice, model.py:100-101,
// each time materialising a [patches, p*p, 32] fp32 temp).
//
// Layout: a 256-thread block owns a 16-row x 64-column strip of one image =
// four 16x16 patches side by side; each wave owns one of them.  The strip is
// read once with coalesced float4 loads (256 B per row per channel), turned
// into gray in registers and parked in LDS.  Per 8x8 sub-patch a wave does
// lane = pixel: only the bins within +-2 of the pixel's own bin can be non-zero
// in fp32 (exp underflows to exactly 0 beyond 14.42 sigma = 2.24 bin widths,
// in the reference too) and only +-1 can matter (beyond: < 3e-17 per pixel, see
// subpatch_sum), so 3 exps per pixel instead of 32; the 64x32 kernel
// values are deposited in a wave-private LDS tile and summed per bin in a
// fixed order (lane = bin) -- deterministic, no float atomics.  The four 8x8
// sums accumulate into the 16x16 patch's histogram.
//
// fp32 throughout, denormals kept (epsilon = 1e-40 is an fp32 denormal, model.py:451).
// The Gaussian is evaluated as exp2(c * r^2) with c = -0.5*log2(e)/sigma^2 folded on the host
// (one v_exp_f32 instead of an IEEE divide + OCML expf: 3 instructions per bin instead of ~45).
// This op cannot be bit-exact against the CPU reference anyway (SLEEF vs GPU transcendentals,
// torch.mean's summation tree); it is held to 2e-5 absolute, and measures ~1e-6.
#include "cgic_common.h"

#include <stdlib.h>

namespace cgic {

constexpr int kEntThreads = 256;
constexpr int kEntStrips = 4;     // 64-column strips per workgroup (a 256-wide image row: one workgroup per 16 rows)
constexpr int kBins = 32;
constexpr int kWin = 3;           // bins evaluated per pixel: the nearest one and its two neighbours
constexpr int kTileStride = 68;  // 64 pixels + 4 pad dwords: conflict-free b128 column reads

struct BinsArg { float v[kBins]; };   // passed by value in the kernarg segment

// Butterfly over the 32 bins held by lanes {0..31} (and, mirrored, {32..63}); every lane ends with
// the same total, summed in the same fixed order => deterministic.  Offsets 1, 2, 4, 8 are DPP operands of
// the adds (quad_perm / row_half_mirror / row_mirror on values that are already uniform inside the smaller
// group = xor), offset 16 is a v_permlane16_swap: no LDS round trip (five ds_bpermute + waits cost ~55 ns per
// sum, six sums per 16x16 patch).  Same association as the ds_bpermute butterfly it replaces, bit for bit.
// (An earlier DPP + v_readlane variant was measured SLOWER: 48.7 vs 39.6 us at B=64.)
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v)
{
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float sum32(float v)
{
    v = dpp_add<0xB1>(v);        // quad_perm [1,0,3,2]: xor 1
    v = dpp_add<0x4E>(v);        // quad_perm [2,3,0,1]: xor 2
    v = dpp_add<0x141>(v);       // row_half_mirror: xor 4 for quad-uniform values
    v = dpp_add<0x140>(v);       // row_mirror: xor 8 for values uniform over 8 lanes
    unsigned int a = __float_as_uint(v), b = a;
    swap16(a, b);
    return __uint_as_float(a) + __uint_as_float(b);      // xor 16
}

// entropy of one histogram: lane (b = lane&31) holds sum over pixels of bin b.
// pdf / norm as pdf * v_rcp_f32(norm) and log as v_log_f32 * ln 2 (1 ulp each) instead of the IEEE divide and
// OCML logf (~30 instructions): this op is held to 2e-5 absolute against the CPU reference and measures ~1e-6
// either way.  v_log_f32 does not take denormals, and an empty bin is pdf = eps = 1e-40: its term
// eps * log(eps) = -9e-39 is dropped (contributes < 3e-37 over 32 bins).
__device__ __forceinline__ float patch_entropy(float s, float inv_npix)
{
    const float eps = 1e-40f;
    float pdf = s * inv_npix;                   // torch.mean over pixels (1/64, 1/256: exact) (:456)
    float norm = sum32(pdf) + eps;              // sum over bins + epsilon     (:457)
    pdf = pdf * __builtin_amdgcn_rcpf(norm) + eps;                          // (:458)
    float t = pdf > 1e-30f ? pdf * (__builtin_amdgcn_logf(pdf) * 0.6931471805599453f) : 0.f;
    return -sum32(t);                           //                             (:459)
}

__global__ __launch_bounds__(kEntThreads) void entropy_maps_kernel(
    const float *__restrict__ x, int64_t H, int64_t W, float exp2_scale, float *__restrict__ e8,
    float *__restrict__ e16, BinsArg bins_arg)
{
    __shared__ __attribute__((aligned(16))) float gray[16][64 + 4];
    __shared__ __attribute__((aligned(16))) float tile[4][kBins * kTileStride];
    __shared__ float bins[kBins];

    const int64_t b = blockIdx.z;
    const int64_t row0 = (int64_t)blockIdx.y * 16;
    const int tid = threadIdx.x;
    const int lane = lane_id();
    const int wave = tid >> 6;
    // A workgroup walks up to kEntStrips consecutive 64-column strips of its 16 rows; the pixels of strip s+1 are
    // requested (into registers) before strip s is computed.  One strip per workgroup spent half of every
    // workgroup's life waiting for its 12 KB of HBM (2.2 of 4.3 us, four workgroups per CU): 22.6 us per launch.
    const int64_t nsx = (W + 63) / 64;
    const int64_t s_lo = (int64_t)blockIdx.x * kEntStrips;
    const int64_t s_hi = s_lo + kEntStrips < nsx ? s_lo + kEntStrips : nsx;

    CGIC_STAMP(16);
#ifdef CGIC_PHASE_CLOCKS      // dev: per-workgroup (start, end) for tools/probe_entropy.py
    const unsigned int dbg_lin = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    if (threadIdx.x == 0 && dbg_lin < 4096) g_blk_t[2 * dbg_lin] = wall_clock64();
    struct DbgEnd { unsigned int lin; __device__ ~DbgEnd() { if (threadIdx.x == 0 && lin < 4096) g_blk_t[2 * lin + 1] = wall_clock64(); } } dbg_end{dbg_lin};
#endif
    if (tid < kBins) bins[tid] = bins_arg.v[tid];
    for (int i = tid; i < 4 * kBins * kTileStride; i += kEntThreads) (&tile[0][0])[i] = 0.f;

    const int lr = tid >> 4;          // row 0..15 of the strip this thread loads
    const int lc4 = (tid & 15) * 4;   // column 0..60
    const int64_t plane = H * W;
    float4 pR = {0.f, 0.f, 0.f, 0.f}, pG = pR, pB = pR;
    auto request = [&](int64_t strip) {
        const int64_t col = strip * 64 + lc4;
        pR = pG = pB = float4{0.f, 0.f, 0.f, 0.f};
        if (strip < s_hi && col < W) {   // W % 16 == 0 => a float4 is entirely inside or outside
            const float *p = x + (b * 3) * plane + (row0 + lr) * W + col;
            pR = *reinterpret_cast<const float4 *>(p);
            pG = *reinterpret_cast<const float4 *>(p + plane);
            pB = *reinterpret_cast<const float4 *>(p + 2 * plane);
        }
    };
    request(s_lo);

    float *T = tile[wave];
    const int bin = lane & 31;
    const int half = lane >> 5;
    const float inv_step = 15.5f;        // (nbins-1)/2 bins per unit; only picks the candidate window

#pragma unroll 1
    for (int64_t strip = s_lo; strip < s_hi; ++strip) {
    const int64_t col0 = strip * 64;
    {
        // gray = 0.2989 R + 0.5870 G + 0.1140 B  (:471)
        float4 g4;
        g4.x = (0.2989f * pR.x + 0.5870f * pG.x) + 0.1140f * pB.x;
        g4.y = (0.2989f * pR.y + 0.5870f * pG.y) + 0.1140f * pB.y;
        g4.z = (0.2989f * pR.z + 0.5870f * pG.z) + 0.1140f * pB.z;
        g4.w = (0.2989f * pR.w + 0.5870f * pG.w) + 0.1140f * pB.w;
        CGIC_STAMP(17);
        *reinterpret_cast<float4 *>(&gray[lr][lc4]) = g4;
    }
    CGIC_STAMP(18);
    __syncthreads();
    CGIC_STAMP(19);
    request(strip + 1);                  // in flight during this strip's arithmetic
    const float bin0 = bins[0];
    float s16 = 0.f;

    if (col0 + wave * 16 < W) {          // else this wave's 16x16 patch is outside the image (whole wave)
    // sum over the 64 pixels of 8x8 sub-patch `sp` of every bin's kernel value; every lane returns the
    // total of bin (lane & 31) (both half-waves hold the same 32 totals)
    auto subpatch_sum = [&](int sp) -> float {
        const int sy = sp >> 1, sx = sp & 1;
        // lane = pixel (row-major inside the 8x8 patch, like nn.Unfold)
        const int py = lane >> 3, px = lane & 7;
        const float gv = gray[sy * 8 + py][wave * 16 + sx * 8 + px];
        // candidate window: nearest bin +-1.  A bin further than 1.5 bin widths holds exp(-0.5 (0.0968/sigma)^2) <= 3e-17
        // (sigma <= 0.0111; 4.5e-21 at the reference's 0.01): nonzero in fp32 and summed by the reference, but worth
        // < 1e-15 of entropy -- far below this op's 2e-5 tolerance and its ~1e-6 transcendental noise.
        float fc = rintf((gv - bin0) * inv_step);
        fc = fminf(fmaxf(fc, 0.f), 31.f);
        int jc = (gv == gv) ? (int)fc : 0;
        int jlo = jc - 1 < 0 ? 0 : jc - 1;
        jlo = jlo > kBins - kWin ? kBins - kWin : jlo;
#pragma unroll
        for (int q = 0; q < kWin; ++q) {
            const int jb = jlo + q;
            const float res = gv - bins[jb];                       // residuals          (:453)
            const float kv = __builtin_amdgcn_exp2f(exp2_scale * (res * res));   // exp(-0.5 (res/sigma)^2) (:454)
            T[jb * kTileStride + lane] = kv;
        }
        __builtin_amdgcn_wave_barrier();
        // lane = (bin, half): sum 32 pixels in a fixed order, then the two halves
        const float4 *row = reinterpret_cast<const float4 *>(&T[bin * kTileStride + half * 32]);
        float4 a4 = row[0];
#pragma unroll
        for (int q = 1; q < 8; ++q) {            // four independent accumulators: 8-deep chains, fixed order
            const float4 v = row[q];
            a4.x += v.x; a4.y += v.y; a4.z += v.z; a4.w += v.w;
        }
        float s = (a4.x + a4.y) + (a4.z + a4.w);
        s += __shfl_xor(s, 32, kWave);     // (a v_permlane32_swap here instead was measured 18 us SLOWER per launch)
        __builtin_amdgcn_wave_barrier();
        // clear what this pixel deposited, ready for the next sub-patch
#pragma unroll
        for (int q = 0; q < kWin; ++q) T[(jlo + q) * kTileStride + lane] = 0.f;
        __builtin_amdgcn_wave_barrier();
        return s;
    };

#pragma unroll 1
    for (int pr = 0; pr < 2; ++pr) {     // sub-patch pairs (0,0)(0,1) then (1,0)(1,1), row-major like nn.Unfold
        const float sA = subpatch_sum(2 * pr);
        const float sB = subpatch_sum(2 * pr + 1);
        s16 += sA;
        s16 += sB;
        if (e8) {
            // finalise BOTH 8x8 patches in one pass: half-wave 0 takes A, half-wave 1 takes B (the
            // butterfly offsets 1..16 of sum32 stay inside a 32-lane half)
            const float ent = patch_entropy(half ? sB : sA, 1.0f / 64.0f);
            if ((lane & 31) == 0) {
                const int64_t h8 = H / 8, w8 = W / 8;
                e8[(b * h8 + (row0 / 8 + pr)) * w8 + (col0 / 8 + wave * 2 + half)] = ent;
            }
        }
        CGIC_STAMP(20 + pr);
    }
    if (e16) {
        float ent = patch_entropy(s16, 1.0f / 256.0f);
        if (lane == 0) {
            const int64_t h16 = H / 16, w16 = W / 16;
            e16[(b * h16 + row0 / 16) * w16 + (col0 / 16 + wave)] = ent;
        }
    }
    }   // wave inside the image
    CGIC_STAMP(24);
    __syncthreads();                     // everybody is done with gray[] before the next strip overwrites it
    }   // strips
}

}  // namespace cgic

using namespace cgic;

extern "C" int cgic_entropy_maps_f32(const float *x, int64_t B, int64_t H, int64_t W, const float *bins,
                                     int nbins, float sigma, float *e8, float *e16, cgic_stream_t stream)
{
    CGIC_REQUIRE(x && bins, CGIC_ERR_INVALID, "entropy: x and bins must not be NULL");
    CGIC_REQUIRE(nbins == kBins, CGIC_ERR_UNSUPPORTED, "entropy: nbins=%d; the reference uses 32 (model.py:480)", nbins);
    CGIC_REQUIRE(B >= 0 && H > 0 && W > 0 && H % 16 == 0 && W % 16 == 0, CGIC_ERR_INVALID,
                 "entropy: H=%lld W=%lld must be positive multiples of 16", (long long)H, (long long)W);
    CGIC_REQUIRE(B <= 65535 && H / 16 <= 65535, CGIC_ERR_UNSUPPORTED, "entropy: batch/height exceed the grid limits");
    // The +-1-bin candidate window drops kernel values <= exp(-0.5 (1.5 * (2/31) / sigma)^2): 3e-17 at the bound
    // below, 4.5e-21 at the reference's sigma
    CGIC_REQUIRE(sigma > 0.f && sigma <= 0.0111f, CGIC_ERR_UNSUPPORTED,
                 "entropy: sigma=%g; the 3-bin window assumes the reference's sigma=0.01 (model.py:481)", sigma);
    for (int i = 1; i < kBins; ++i)
        CGIC_REQUIRE(fabsf((bins[i] - bins[i - 1]) - 2.0f / 31.0f) < 1e-5f, CGIC_ERR_UNSUPPORTED,
                     "entropy: bins are not linspace(-1, 1, 32)");
    if (B == 0 || (!e8 && !e16)) return CGIC_OK;

    BinsArg ba;
    memcpy(ba.v, bins, sizeof(ba.v));
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((unsigned)(((W + 63) / 64 + kEntStrips - 1) / kEntStrips), (unsigned)(H / 16), (unsigned)B);
    // exp(-0.5 (r/sigma)^2) = exp2(c r^2), c = -0.5 log2(e) / sigma^2 (float64 on the host, rounded once)
    const float exp2_scale = (float)(-0.5 * 1.4426950408889634 / ((double)sigma * (double)sigma));
#ifdef CGIC_DEV_KNOBS
    // dev: pad the workgroup's LDS so that fewer of them fit a CU (co-residency experiments)
    static const int pad = getenv("CGIC_ENT_PAD") ? atoi(getenv("CGIC_ENT_PAD")) : 0;
    if (pad > 0) {
        CGIC_HIP_TRY(hipFuncSetAttribute((const void *)entropy_maps_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, pad));
        hipLaunchKernelGGL(entropy_maps_kernel, grid, dim3(kEntThreads), (size_t)pad, s, x, H, W, exp2_scale, e8, e16, ba);
        return launch_check("entropy_maps_kernel");
    }
#endif
    hipLaunchKernelGGL(entropy_maps_kernel, grid, dim3(kEntThreads), 0, s, x, H, W, exp2_scale, e8, e16, ba);
    return launch_check("entropy_maps_kernel");
}
