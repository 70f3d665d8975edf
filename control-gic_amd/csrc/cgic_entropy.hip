// cgic_entropy.hip -- per-patch soft-histogram Shannon entropy maps for patch
// sizes 8 and 16 in ONE pass over the image (reference: Entropy,
// CGIC/models/model.py:433-483; CGIC.encode runs it twice, model.py:100-101,
// each time materialising a [patches, p*p, 32] fp32 temp).
//
// Layout: a wave owns one 16x16 patch at a time (lane = row, 4 consecutive columns: a 64-byte quad of lanes per
// row, read as float4 per channel) and walks the patches of its 256-thread workgroup's row band; the pixels of the
// next patch are requested before the current one is computed.  Waves never synchronise with each other.
//
// Per pixel only the bins within +-2 of its own bin can be non-zero in fp32 (exp underflows to exactly 0 beyond
// 14.42 sigma = 2.24 bin widths, in the reference too) and only the two that bracket it matter (every other bin is
// >= 6.45 sigma away: <= 9e-10 per pixel, < 1e-7 of entropy), so 2 exps per pixel instead of 32.  The histogram of an 8x8 sub-patch is the sum of those kernel values per bin.
// Round 2 (second half): the values are DEPOSITED, not gathered -- each one is converted to fixed point (quantum
// 2^-26) and added with a return-less 32-bit LDS atomic into a wave-private table [bin][sub-patch][replica]; integer
// addition is associative, so the result does not depend on the order in which the LDS serves the lanes
// (deterministic, run to run and across layouts).  Eight replicas per (bin, sub-patch), chosen by the pixel's row,
// keep same-address collisions at <= 2 lanes per instruction (a smooth patch puts all its pixels into one bin) and a
// replica's total at <= 8 pixels * 2^26 = 2^29.  Lane = (bin, half-wave) then reads its 8 replicas (two b128
// loads), converts and adds them in a fixed tree.  Before: every value was written into a [bin][pixel] LDS tile and
// each bin summed all 64 columns (8 b128 reads + 31 adds per lane and sub-patch for 192 non-zero values):
// ~420 VALU instructions per patch against ~200 now.
//
// fp32 throughout.  The Gaussian is exp2(c * r^2) with c = -0.5*log2(e)/sigma^2 folded on the host (one v_exp_f32
// instead of an IEEE divide + OCML expf).  The entropy of a histogram only depends on its normalised form, so the
// mean over pixels (:456) is not materialised.  (Evaluating it as ln S - (sum s ln s)/S, which needs both sums in one
// lane only, was measured 5x less accurate -- 5.9e-6 against the CPU reference -- because the two terms cancel.)
// This op cannot be bit-exact against the CPU reference anyway (SLEEF vs GPU transcendentals, torch.mean's
// summation tree); it is held to 2e-5 absolute, and measures ~1e-6.
#include "cgic_entropy_dev.h"

#include <stdlib.h>

namespace cgic {

constexpr int kEntThreads = 256;
constexpr int kEntWaves = kEntThreads / kWave;
constexpr int kWin = 2;            // bins evaluated per pixel: the two that bracket it
constexpr int kHistStride = 36;    // dwords per bin: 4 sub-patches x 8 replicas + 4 pad (conflict-free b128 rows)
constexpr float kFixScale = 67108864.f;           // 2^26: a replica collects <= 8 pixels with values <= 1, four replicas < 2^31

template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp_add(float v)
{
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, true));
}
// Sum over the 32 lanes of each half-wave; the total is valid in the UPPER row of the half only (lanes 16..31 and
// 48..63).  Offsets 1, 2, 4, 8 are DPP operands of the adds (quad_perm / row_half_mirror / row_mirror on values that
// are already uniform inside the smaller group = xor butterfly), then row_bcast:15 hands row 0's total to row 1 and
// row 2's to row 3.  Fixed association => deterministic.  (The all-lanes form needs a v_permlane16_swap instead of
// the last step: 4x the issue time of a DPP add, tools/probes/probe_valu2.)
__device__ __forceinline__ float sum32_upper(float v)
{
    v = dpp_add<0xB1>(v);        // quad_perm [1,0,3,2]: xor 1
    v = dpp_add<0x4E>(v);        // quad_perm [2,3,0,1]: xor 2
    v = dpp_add<0x141>(v);       // row_half_mirror: xor 4 for quad-uniform values
    v = dpp_add<0x140>(v);       // row_mirror: xor 8 for values uniform over 8 lanes
    return dpp_add<0x142, 0xA>(v);   // row_bcast:15 into rows 1 and 3
}

// Same sum, valid in all 64 lanes: the last step is a ds_swizzle (lane ^ 16 through the LDS crossbar, no memory):
// one LDS-port instruction instead of a v_permlane16_swap that holds the VALU port for four issue slots.
__device__ __forceinline__ float sum32_all(float v)
{
    v = dpp_add<0xB1>(v);
    v = dpp_add<0x4E>(v);
    v = dpp_add<0x141>(v);
    v = dpp_add<0x140>(v);
    return v + __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x401F));   // xor 0x10
}

// entropy of the histogram whose bin b total `s` (>= 0, any common scale: the mean over pixels of :456 drops out)
// sits in lane b of a half-wave; valid in lanes 16..31 / 48..63.  pdf / norm as pdf * v_rcp_f32(norm) and log as
// v_log_f32 * ln 2 (1 ulp each) instead of the IEEE divide and OCML logf.  An empty bin is p = eps = 1e-40 in the
// reference: its term eps * log(eps) = -9e-39 is dropped (< 3e-37 over 32 bins; an all-zero histogram gives 0 where
// the reference gives 2.9e-37).  NaN totals stay NaN.
__device__ __forceinline__ float hist_entropy(float s)
{
    // (+ 1e-30 instead of max(., 1e-30): a full-rate add where v_max_f32 holds the issue port for two slots; it changes
    // nothing above 1e-23 and turns an empty histogram / an empty bin into 0 * finite)
    const float S = sum32_all(s);                                   // sum over bins          (:457)
    const float p = s * __builtin_amdgcn_rcpf(S + 1e-30f);          //                        (:458)
    const float t = p * __builtin_amdgcn_logf(p + 1e-30f);          // p log2 p, 0 for an empty bin
    return -0.6931471805599453f * sum32_upper(t);                   //                        (:459)
}

__device__ __forceinline__ int cvt_floor_i32(float v)
{
    int r;
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(r) : "v"(v));      // floor, saturating, NaN -> 0
    return r;
}
__device__ __forceinline__ unsigned int cvt_round_u32(float v)
{
    int r;
    asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(v));      // floor(v + 0.5); v in [0, 2^26]; NaN -> 0
    return (unsigned int)r;
}

// U8: x is a uint8 [B, H, W, 3] frame (PIL / decoder layout) and the kernel is ToTensor + Entropy in one pass: a lane reads the
// 12 bytes of its 4 pixels, converts them to the fp32 values T.ToTensor() produces (bit for bit) and, if x_out is given, writes
// them as the fp32 [B, 3, H, W] tensor the conv encoder takes (inference.py:50-59 + model.py:99-101): 3 B read + 12 B written
// per pixel instead of 3 + 12 (ToTensor) and 12 again (Entropy)
// WIN (cgic_entropy_maps_tiles): the batch is the tiles of ONE shape cut out of unpadded images -- pad + crop of the tiling driver
// (inference_high_resolution.py:145-173, :236-244) and the maps in one pass: image b = (source image b / T, tile b % T); a lane
// reads its 4 pixels from the SOURCE window (zeros where the tile reaches into the centred pad), the tile itself is written to
// x_out as a by-product (the conv encoder's input and the router's refinement pixels).  33 MB read + 33 MB written for a
// 2040x1356 image instead of (33 + 33) for the cut and 33 again for the maps.
constexpr int kEntMaxTiles = 48;
struct EntWindow {
    const void *src;          // fp32 [N,3,srcH,srcW] or uint8 [N,srcH,srcW,3]
    int srcH, srcW, T;
    int org[kEntMaxTiles][2]; // (y0, x0) of tile k in unpadded source coordinates (negative inside the pad)
};

template <bool U8, bool WIN = false>
__device__ __forceinline__ void entropy_maps_body(
    const void *__restrict__ xin, int64_t H, int64_t W, float exp2_scale, float *__restrict__ e8,
    float *__restrict__ e16, const BinsArg &bins_arg, int patches_per_wave, float *__restrict__ x_out, float *__restrict__ flat8,
    const Blk blk, const EntWindow *win = nullptr)
{
    const float *__restrict__ x = reinterpret_cast<const float *>(xin);
    __shared__ __attribute__((aligned(16))) unsigned int hist_all[kEntWaves][kBins * kHistStride];
    __shared__ float bins[kBins + 4];

    const int64_t b = blk.z;
    const int64_t row0 = (int64_t)blk.y * 16;
    const int tid = threadIdx.x;
    const int lane = lane_id();
    const int wave = tid >> 6;
    // the workgroup's band of 16-pixel-wide patches; wave w takes patches w, w + 4, ... of it
    const int64_t npx = W / 16;
    const int64_t p_lo = (int64_t)blk.x * (kEntWaves * patches_per_wave);
    const int64_t p_end = p_lo + kEntWaves * patches_per_wave < npx ? p_lo + kEntWaves * patches_per_wave : npx;

    CGIC_STAMP(16);
#ifdef CGIC_PHASE_CLOCKS      // dev: per-workgroup (start, end) for tools/probes/probe_entropy.py
    const unsigned int dbg_lin = (blk.z * blk.ny + blk.y) * blk.nx + blk.x;
    if (threadIdx.x == 0 && dbg_lin < 4096) g_blk_t[2 * dbg_lin] = wall_clock64();
    struct DbgEnd { unsigned int lin; __device__ ~DbgEnd() { if (threadIdx.x == 0 && lin < 4096) g_blk_t[2 * lin + 1] = wall_clock64(); } } dbg_end{dbg_lin};
#endif

    // pixel role: lane = (row of the patch, 4 consecutive columns)
    const int prow = lane >> 2;
    const int pc4 = (lane & 3) * 4;
    const int64_t plane = H * W;
    float4 pR = {0.f, 0.f, 0.f, 0.f}, pG = pR, pB = pR;
    unsigned int raw0 = 0, raw1 = 0, raw2 = 0;      // U8: the 12 bytes R0 G0 B0 R1 | G1 B1 R2 G2 | B2 R3 G3 B3 of the lane's 4 pixels
    // WIN: this tile's window of its source image (all scalar)
    int wn = 0, wy0 = 0, wx0 = 0, sH = 0, sW = 0;
    bool wvec = false;
    if constexpr (WIN) {
        wn = (int)(b / win->T);
        const int wk = (int)b - wn * win->T;
        wy0 = win->org[wk][0]; wx0 = win->org[wk][1]; sH = win->srcH; sW = win->srcW;
        // whole 4-pixel units can be fetched as vectors when the window is shifted by a multiple of 4 pixels in rows of a multiple of 4
        wvec = ((wx0 | sW) & 3) == 0 && (reinterpret_cast<uintptr_t>(win->src) & 15) == 0;
    }
    auto request = [&](int64_t patch) {
        if constexpr (WIN) {
            if (patch < p_end) {
                const int sy = wy0 + (int)(row0 + prow), sx = wx0 + (int)(patch * 16 + pc4);
                const bool row_ok = sy >= 0 && sy < sH, all_in = sx >= 0 && sx + 3 < sW;
                if (U8) {
                    raw0 = raw1 = raw2 = 0u;
                    if (row_ok) {
                        const unsigned char *q = reinterpret_cast<const unsigned char *>(win->src) + (((int64_t)wn * sH + sy) * (int64_t)sW + sx) * 3;
                        if (all_in && wvec) {
                            const unsigned int *qw = reinterpret_cast<const unsigned int *>(q);
                            raw0 = qw[0]; raw1 = qw[1]; raw2 = qw[2];
                        } else {
#pragma unroll
                            for (int j = 0; j < 12; ++j) {
                                const int sxj = sx + j / 3;
                                const unsigned int v = (sxj >= 0 && sxj < sW) ? q[j] : 0u;
                                if (j < 4) raw0 |= v << (8 * j); else if (j < 8) raw1 |= v << (8 * (j - 4)); else raw2 |= v << (8 * (j - 8));
                            }
                        }
                    }
                } else {
                    const float4 zero = {0.f, 0.f, 0.f, 0.f};
                    pR = zero; pG = zero; pB = zero;
                    if (row_ok) {
                        const int64_t splane = (int64_t)sH * sW;
                        const float *p = reinterpret_cast<const float *>(win->src) + ((int64_t)wn * 3) * splane + (int64_t)sy * sW + sx;
                        if (all_in && wvec) {
                            pR = *reinterpret_cast<const float4 *>(p);
                            pG = *reinterpret_cast<const float4 *>(p + splane);
                            pB = *reinterpret_cast<const float4 *>(p + 2 * splane);
                        } else {
                            float r[4], g4[4], bl[4];
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const bool ok = sx + j >= 0 && sx + j < sW;
                                r[j] = ok ? p[j] : 0.f; g4[j] = ok ? p[splane + j] : 0.f; bl[j] = ok ? p[2 * splane + j] : 0.f;
                            }
                            pR = {r[0], r[1], r[2], r[3]}; pG = {g4[0], g4[1], g4[2], g4[3]}; pB = {bl[0], bl[1], bl[2], bl[3]};
                        }
                    }
                }
            }
            return;
        }
        if (patch < p_end) {
            if (U8) {
                const unsigned int *q = reinterpret_cast<const unsigned int *>(
                    reinterpret_cast<const unsigned char *>(xin) + (((b * H + row0 + prow) * W + patch * 16 + pc4) * 3));
                raw0 = q[0]; raw1 = q[1]; raw2 = q[2];
            } else {
                const float *p = x + (b * 3) * plane + (row0 + prow) * W + patch * 16 + pc4;
                pR = *reinterpret_cast<const float4 *>(p);
                pG = *reinterpret_cast<const float4 *>(p + plane);
                pB = *reinterpret_cast<const float4 *>(p + 2 * plane);
            }
        }
    };
    auto byte_f = [](unsigned int w, int k) { return (float)((w >> (8 * k)) & 0xFFu); };      // v_cvt_f32_ubyteK
    request(p_lo + wave);

    if (tid < kBins) bins[tid] = bins_arg.v[tid];
    // sub-patch (sy, sx) of this lane's pixels and its replica: slot = sub * 8 + (row & 7)
    const int slot = ((lane >> 5) << 4) | (((lane >> 1) & 1) << 3) | (prow & 7);
    // histogram role: lane = (bin, half); over the two read-back rounds it owns sub-patches `half` and 2 + `half`
    const int bin = lane & 31;
    const int half = lane >> 5;
    unsigned int *Hw = hist_all[wave];
    uint4 *own0 = reinterpret_cast<uint4 *>(Hw + bin * kHistStride + half * 8);
    uint4 *own1 = reinterpret_cast<uint4 *>(Hw + bin * kHistStride + 16 + half * 8);
    const uint4 zero4 = {0u, 0u, 0u, 0u};
    own0[0] = zero4; own0[1] = zero4; own1[0] = zero4; own1[1] = zero4;
    __syncthreads();                                      // bins[]; the only workgroup-wide barrier
    const float k0 = -bins[0] * 15.5f;                    // floor((g - bins[0]) * 15.5) = the bin below the pixel
    // flat8 by-product: crossbar source of a lane (first lane of its 8x8 sub-patch) and where the sub-patch's flag goes
    const int flat_src = (lane & 0x22) * 4;
    const int64_t flat_off = (int64_t)(lane >> 5) * (W / 8) + ((lane >> 1) & 1);
    float *__restrict__ flat_row = flat8 ? flat8 + (b * (H / 8) + row0 / 8) * (W / 8) : nullptr;

#pragma unroll 1
    for (int64_t patch = p_lo + wave; patch < p_end; patch += kEntWaves) {
        if (U8) {
            pR = {unit_of_byte(byte_f(raw0, 0)), unit_of_byte(byte_f(raw0, 3)), unit_of_byte(byte_f(raw1, 2)), unit_of_byte(byte_f(raw2, 1))};
            pG = {unit_of_byte(byte_f(raw0, 1)), unit_of_byte(byte_f(raw1, 0)), unit_of_byte(byte_f(raw1, 3)), unit_of_byte(byte_f(raw2, 2))};
            pB = {unit_of_byte(byte_f(raw0, 2)), unit_of_byte(byte_f(raw1, 1)), unit_of_byte(byte_f(raw2, 0)), unit_of_byte(byte_f(raw2, 3))};
        }
        if constexpr (U8 || WIN) if (x_out) {
            float *o = x_out + (b * 3) * plane + (row0 + prow) * W + patch * 16 + pc4;
            *reinterpret_cast<float4 *>(o) = pR;
            *reinterpret_cast<float4 *>(o + plane) = pG;
            *reinterpret_cast<float4 *>(o + 2 * plane) = pB;
        }
        // gray = 0.2989 R + 0.5870 G + 0.1140 B  (:471)
        float g[4];
        g[0] = (0.2989f * pR.x + 0.5870f * pG.x) + 0.1140f * pB.x;
        g[1] = (0.2989f * pR.y + 0.5870f * pG.y) + 0.1140f * pB.y;
        g[2] = (0.2989f * pR.z + 0.5870f * pG.z) + 0.1140f * pB.z;
        g[3] = (0.2989f * pR.w + 0.5870f * pG.w) + 0.1140f * pB.w;
        request(patch + kEntWaves);            // in flight during this patch's arithmetic
        CGIC_STAMP(17);
        if (flat8) {
            // flat8[sub-patch] = its gray value if all 64 pixels of the 8x8 sub-patch carry the SAME gray (bit for bit), else NaN:
            // lets the router's refinement evaluate constant patches -- blown-out sky, letterbox bars, flat graphics: the big tie
            // groups of real content -- once per distinct gray instead of once per patch (cgic_router_dev.h).  Sub-patch (sy, sx) =
            // the lanes with bit 5 == sy and bit 1 == sx; its first lane is lane & 0x22.  Alone this kernel is bound by VALU issue,
            // so the test is kept to one crossbar read, four compares and one select per lane: everything else is 64-bit mask
            // arithmetic on the scalar unit (the first version -- per-lane masks and shifts -- cost 2 of 14 us in the one-lane loop).
            const float first = __int_as_float(__builtin_amdgcn_ds_bpermute(flat_src, __float_as_int(g[0])));
            const unsigned long long eq = __ballot((g[0] == first) & (g[1] == first) & (g[2] == first) & (g[3] == first));   // (NaN: never)
            constexpr unsigned long long M00 = 0x0000000033333333ull, M01 = 0x00000000CCCCCCCCull, M10 = M00 << 32, M11 = M01 << 32;
            const unsigned long long ok = ((eq & M00) == M00 ? 1ull << 0 : 0ull) | ((eq & M01) == M01 ? 1ull << 2 : 0ull) |
                                          ((eq & M10) == M10 ? 1ull << 32 : 0ull) | ((eq & M11) == M11 ? 1ull << 34 : 0ull);
            if (__builtin_amdgcn_inverse_ballot_w64(0x0000000500000005ull))       // lanes 0, 2, 32, 34: one per sub-patch
                flat_row[patch * 2 + flat_off] = __builtin_amdgcn_inverse_ballot_w64(ok) ? first : __builtin_nanf("");
        }
        unsigned long long nan_lanes = 0;      // lanes that hold a NaN pixel (the reference's histogram turns NaN)
        // candidate window: the two bins that bracket the pixel.  Every other bin is at least one bin width = 6.45 sigma away
        // and holds <= exp(-0.5 * 6.45^2) = 9e-10 (6.5e-9 at the largest sigma accepted): nonzero in fp32 and summed by the
        // reference, but worth < 1e-7 of entropy against this op's ~1e-6 transcendental noise (a third bin was evaluated until
        // the kernel turned out to carry 31 % of the step's VALU instructions: 12 -> 8 values per lane and patch).  All eight
        // bin centres are fetched before the first deposit (the compiler does not move LDS reads across the atomics).
        int jlo[4];
        float bc[4][kWin];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            nan_lanes |= __ballot(g[i] != g[i]);
            jlo[i] = min(max(cvt_floor_i32(fmaf(g[i], 15.5f, k0)), 0), kBins - kWin);       // v_med3_i32
#pragma unroll
            for (int q = 0; q < kWin; ++q) bc[i][q] = bins[jlo[i] + q];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            unsigned int *hp = Hw + jlo[i] * kHistStride + slot;
#pragma unroll
            for (int q = 0; q < kWin; ++q) {
                const float res = g[i] - bc[i][q];                                     // residuals (:453)
                const float kv = __builtin_amdgcn_exp2f(exp2_scale * (res * res));     // exp(-0.5 (res/sigma)^2) (:454)
                __hip_atomic_fetch_add(hp + q * kHistStride, cvt_round_u32(kv * kFixScale), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_WAVEFRONT);
            }
        }
        __builtin_amdgcn_wave_barrier();       // LDS operations of one wave execute in order
        CGIC_STAMP(18);

        // read back: round k = sub-patch pair (k,0) | (k,1) in the two half-waves, row-major like nn.Unfold
        float s8[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            uint4 *own = k ? own1 : own0;
            const uint4 a = own[0], c = own[1];
            own[0] = zero4; own[1] = zero4;    // ready for the next patch
            // four replicas add up as integers (<= 32 pixels * 2^26 < 2^32), the two halves as floats: 6 + 3 instructions
            s8[k] = (float)((a.x + a.y) + (a.z + a.w)) + (float)((c.x + c.y) + (c.z + c.w));
        }
        __builtin_amdgcn_wave_barrier();
        float s16 = s8[0] + s8[1];
        s16 += __shfl_xor(s16, 32, kWave);     // both halves: (sub 0 + sub 2) + (sub 1 + sub 3), commutative
        if (nan_lanes) {                       // wave-uniform, never taken on real images
            const unsigned long long sx0 = 0x3333333333333333ull, lo = 0xFFFFFFFFull;
            const float qnan = __builtin_nanf("");
#pragma unroll
            for (int k = 0; k < 2; ++k) {      // sub-patch (k, half): rows 8k..8k+7 = lanes 32k..32k+31, sx = bit 1 of the lane
                const unsigned long long rows = k ? nan_lanes >> 32 : nan_lanes & lo;
                const bool badA = (rows & sx0 & lo) != 0, badB = (rows & ~sx0 & lo) != 0;
                if (half ? badB : badA) s8[k] = qnan;
            }
            s16 = qnan;
        }
        CGIC_STAMP(19);

        const int64_t pcol = patch;
        if (e8) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const float ent = hist_entropy(s8[k]);
                if (bin == 16) {
                    const int64_t h8 = H / 8, w8 = W / 8;
                    e8[(b * h8 + (row0 / 8 + k)) * w8 + (pcol * 2 + half)] = ent;
                }
            }
        }
        if (e16) {
            const float ent = hist_entropy(s16);
            if (lane == 16) {
                const int64_t h16 = H / 16, w16 = W / 16;
                e16[(b * h16 + row0 / 16) * w16 + pcol] = ent;
            }
        }
        CGIC_STAMP(24);
    }
}

struct EntArgs {
    const void *x;
    int64_t H, W;
    float exp2_scale;
    int ppw;
    float *e8, *e16, *x_out, *flat8;
    BinsArg bins;
};

template <bool U8>
__global__ __launch_bounds__(kEntThreads) void entropy_maps_kernel(EntArgs a)
{
    entropy_maps_body<U8>(a.x, a.H, a.W, a.exp2_scale, a.e8, a.e16, a.bins, a.ppw, a.x_out, a.flat8, own_blk());
}

// several shape groups in one launch (cgic_common.h: launch groups)
template <bool U8>
__global__ __launch_bounds__(kEntThreads) void entropy_maps_grouped_kernel(Grouped<EntArgs> g)
{
    Blk blk;
    const EntArgs &a = g.a[group_locate(g, &blk)];
    entropy_maps_body<U8>(a.x, a.H, a.W, a.exp2_scale, a.e8, a.e16, a.bins, a.ppw, a.x_out, a.flat8, blk);
}

struct EntWinArgs {
    EntArgs e;
    EntWindow w;
};
template <bool U8>
__global__ __launch_bounds__(kEntThreads) void entropy_tiles_kernel(EntWinArgs a)
{
    entropy_maps_body<U8, true>(nullptr, a.e.H, a.e.W, a.e.exp2_scale, a.e.e8, a.e.e16, a.e.bins, a.e.ppw, a.e.x_out, a.e.flat8, own_blk(), &a.w);
}
template <bool U8>
__global__ __launch_bounds__(kEntThreads) void entropy_tiles_grouped_kernel(Grouped<EntWinArgs> g)
{
    Blk blk;
    const EntWinArgs &a = g.a[group_locate(g, &blk)];
    entropy_maps_body<U8, true>(nullptr, a.e.H, a.e.W, a.e.exp2_scale, a.e.e8, a.e.e16, a.e.bins, a.e.ppw, a.e.x_out, a.e.flat8, blk, &a.w);
}

// =====================================================================================================
// Reference-arithmetic variant (opt-in, cgic_entropy_maps_ref_f32): the reference's own fp32 operation sequence and
// summation ORDER, with exp / log correctly rounded (evaluated in fp64, rounded once).
//
// The router's thresholds are k-th smallest entropies with a strict '<' (RouterTriple.py:21-34): on tie-heavy content
// (8-bit, flat, blocky images) which patches fall under a threshold is decided by the last bits of the maps.  The
// kernel above is accurate to ~1e-6 and order-free, which flips a handful of mask elements in ~1 of 64 such images (and
// 2 of 8 768x768 tiles: profiles / bench `mask_mismatch.tie_heavy_content`).  This variant reproduces how torch's CPU
// operators round (measured against the real Entropy class: 98.6-100 % of all values bit-identical, the rest within
// 5e-7, no mask element flipped on any family -- torch's exp / log are MKL's, not correctly rounded in ~1 % of the
// arguments, which is what is left):
//   gray   = (0.2989 R + 0.5870 G) + 0.1140 B                       three products, two sums, no fma (model.py:471)
//   kv     = exp(-0.5 * ((gray - bin) / sigma)^2)                    IEEE divide, fp32 square and product (:452-454);
//            exactly 0 beyond 14.42 sigma in fp32 (a < -104): at most five consecutive bins per pixel are evaluated
//   pdf    = mean over the patch's pixels in row-major order         torch's cascade sum of an outer reduction: chunks of
//            16 consecutive pixels summed one after the other from 0, the chunk sums added one after the other (:456)
//   norm   = sum over the 32 bins + 1e-40                            torch's inner reduction of 32 contiguous floats: eight
//            strided partials p_k = ((x_k + x_{8+k}) + x_{16+k}) + x_{24+k}, then p_0 + p_1 + ... + p_7 (:457)
//   q      = pdf / norm + 1e-40;  H = -sum(q log q), same 8-partial order (:458-459)
// One 256-thread workgroup per 16x16 block of the image (thread = pixel): the kernel values of a pixel's five-bin window go
// to LDS once and serve both patch sizes; 1024 chunk sums (16 pixels each) are dealt four to a thread, 160 threads
// combine them, 160 finalise.  ~8x the instructions of the kernel above: an option for when bit-level agreement of the
// masks with the CPU reference is wanted, not the default.
// =====================================================================================================
__global__ __launch_bounds__(256) void entropy_ref_kernel(const float *__restrict__ x, int64_t H, int64_t W, float sigma,
                                                          float *__restrict__ e8, float *__restrict__ e16, BinsArg bins_arg)
{
    __shared__ float s_bins[kBins];
    __shared__ float s_val[256][kBins + 1];            // kernel values of a pixel for all 32 bins: zero outside its window (+1: bank padding)
    __shared__ float s_chunk[5][kBins][16];            // [patch: 0..3 = the 8x8 ones, 4 = 16x16][bin][chunk]
    __shared__ float s_pdf[5][kBins];
    __shared__ float s_norm[5];
    const int tid = threadIdx.x;
    const int64_t b = blockIdx.z, by = blockIdx.y, bx = blockIdx.x;
    if (tid < kBins) s_bins[tid] = bins_arg.v[tid];
    const int r = tid >> 4, c = tid & 15;
    const int64_t o = ((b * 3) * H + by * 16 + r) * W + bx * 16 + c;
    const float R = x[o], G = x[o + H * W], Bc = x[o + 2 * H * W];
    const float gray = ref_gray_of(R, G, Bc);
    __syncthreads();
    {
        int j0;
        float v[kRefWin];
        ref_pixel(s_bins, sigma, gray, j0, v);                                  // cgic_entropy_dev.h: the window and its five values
#pragma unroll
        for (int j = 0; j < kBins; ++j) s_val[tid][j] = 0.f;                    // (exactly what exp rounds to out there)
#pragma unroll
        for (int k = 0; k < kRefWin; ++k) s_val[tid][j0 + k] = v[k];
    }
    __syncthreads();
    // chunk sums: task = (patch, bin, chunk): 4 x 32 x 4 for the 8x8 patches (chunk = two rows of 8), 32 x 16 for the 16x16
    // patch (chunk = one row of 16); 1024 tasks of 16 sequential additions, four per thread
    for (int task = tid; task < 1024; task += 256) {
        float acc = 0.f;
        int pt, bin, ch;
        if (task < 512) {
            pt = task >> 7; bin = (task >> 2) & 31; ch = task & 3;
            const int pr = (pt >> 1) * 8, pc = (pt & 1) * 8;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int k = 16 * ch + i, px = (pr + (k >> 3)) * 16 + pc + (k & 7);
                acc = acc + s_val[px][bin];
            }
        } else {
            const int t2 = task - 512;
            pt = 4; bin = t2 >> 4; ch = t2 & 15;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc = acc + s_val[ch * 16 + i][bin];
        }
        s_chunk[pt][bin][ch] = acc;
    }
    __syncthreads();
    if (tid < 160) {
        const int pt = tid >> 5, bin = tid & 31;
        const int nch = pt < 4 ? 4 : 16;
        float acc = 0.f;
        for (int ch = 0; ch < nch; ++ch) acc = acc + s_chunk[pt][bin][ch];
        s_pdf[pt][bin] = acc / (pt < 4 ? 64.0f : 256.0f);                    // torch.mean: the sum times... divided by the count (exact: a power of two)
    }
    __syncthreads();
    const float eps = 1e-40f;
    if (tid < 5) s_norm[tid] = sum32_lanes8(s_pdf[tid]) + eps;
    __syncthreads();
    if (tid < 160) {
        const int pt = tid >> 5, bin = tid & 31;
        const float q = s_pdf[pt][bin] / s_norm[pt] + eps;
        s_pdf[pt][bin] = q * (float)log((double)q);
    }
    __syncthreads();
    if (tid < 5) {
        const float ent = -sum32_lanes8(s_pdf[tid]);
        if (tid < 4) {
            if (e8) e8[(b * (H / 8) + by * 2 + (tid >> 1)) * (W / 8) + bx * 2 + (tid & 1)] = ent;
        } else if (e16) {
            e16[(b * (H / 16) + by) * (W / 16) + bx] = ent;
        }
    }
}

}  // namespace cgic

using namespace cgic;

static int entropy_maps_launch(const void *x, bool u8, int64_t B, int64_t H, int64_t W, const float *bins, int nbins, float sigma,
                               float *x_out, float *e8, float *e16, float *flat8, cgic_stream_t stream)
{
    CGIC_REQUIRE(x && bins, CGIC_ERR_INVALID, "entropy: x and bins must not be NULL");
    CGIC_REQUIRE(nbins == kBins, CGIC_ERR_UNSUPPORTED, "entropy: nbins=%d; the reference uses 32 (model.py:480)", nbins);
    CGIC_REQUIRE(B >= 0 && H > 0 && W > 0 && H % 16 == 0 && W % 16 == 0, CGIC_ERR_INVALID,
                 "entropy: H=%lld W=%lld must be positive multiples of 16", (long long)H, (long long)W);
    CGIC_REQUIRE(B <= 65535 && H / 16 <= 65535, CGIC_ERR_UNSUPPORTED, "entropy: batch/height exceed the grid limits");
    // The two-bin window drops kernel values <= exp(-0.5 ((2/31) / sigma)^2): 6.5e-9 at the bound below, 9e-10 at the
    // reference's sigma
    CGIC_REQUIRE(sigma > 0.f && sigma <= 0.0105f, CGIC_ERR_UNSUPPORTED,
                 "entropy: sigma=%g; the 2-bin window assumes the reference's sigma=0.01 (model.py:481)", sigma);
    for (int i = 1; i < kBins; ++i)
        CGIC_REQUIRE(fabsf((bins[i] - bins[i - 1]) - 2.0f / 31.0f) < 1e-5f, CGIC_ERR_UNSUPPORTED,
                     "entropy: bins are not linspace(-1, 1, 32)");
    if (B == 0 || (!e8 && !e16 && !flat8 && !(u8 && x_out))) return CGIC_OK;

    BinsArg ba;
    memcpy(ba.v, bins, sizeof(ba.v));
    hipStream_t s = (hipStream_t)stream;
    // a wave walks `ppw` patches of its row band: 4 for a 256-wide image (one workgroup per 16 rows)
#ifdef CGIC_ENT_PPW
    const int ppw = CGIC_ENT_PPW;       // dev: tools/build_variants.sh
#else
    const int ppw = 4;
#endif
    const int64_t per_wg = (int64_t)kEntWaves * ppw;
    dim3 grid((unsigned)((W / 16 + per_wg - 1) / per_wg), (unsigned)(H / 16), (unsigned)B);
    // exp(-0.5 (r/sigma)^2) = exp2(c r^2), c = -0.5 log2(e) / sigma^2 (float64 on the host, rounded once)
    const float exp2_scale = (float)(-0.5 * 1.4426950408889634 / ((double)sigma * (double)sigma));
    EntArgs a;
    a.x = x; a.H = H; a.W = W; a.exp2_scale = exp2_scale; a.ppw = ppw; a.e8 = e8; a.e16 = e16; a.x_out = u8 ? x_out : nullptr; a.flat8 = flat8;
    a.bins = ba;
#ifdef CGIC_DEV_KNOBS
    // dev: pad the workgroup's LDS so that fewer of them fit a CU (co-residency experiments)
    static const int pad = getenv("CGIC_ENT_PAD") ? atoi(getenv("CGIC_ENT_PAD")) : 0;
    if (pad > 0 && !u8 && !group_recording()) {
        CGIC_HIP_TRY(hipFuncSetAttribute((const void *)entropy_maps_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, pad));
        hipLaunchKernelGGL(entropy_maps_kernel<false>, grid, dim3(kEntThreads), (size_t)pad, s, a);
        return launch_check("entropy_maps_kernel");
    }
#endif
    if (u8)
        return launch_or_record(KID_ENTROPY_U8, grid, dim3(kEntThreads), 0, a, s, [=] {
            hipLaunchKernelGGL(entropy_maps_kernel<true>, grid, dim3(kEntThreads), 0, s, a);
            return launch_check("entropy_maps_kernel"); });
    return launch_or_record(KID_ENTROPY_F32, grid, dim3(kEntThreads), 0, a, s, [=] {
        hipLaunchKernelGGL(entropy_maps_kernel<false>, grid, dim3(kEntThreads), 0, s, a);
        return launch_check("entropy_maps_kernel"); });
}

template <bool U8>
static int entropy_grouped_launch(const GroupRec *const *recs, int n, hipStream_t s)
{
    Grouped<EntArgs> g;
    size_t lds;
    int rc = fill_grouped(recs, n, &g, &lds);
    if (rc) return rc;
    hipLaunchKernelGGL(entropy_maps_grouped_kernel<U8>, dim3(g.start[kMaxGroups]), dim3(kEntThreads), 0, s, g);
    return launch_check("entropy_maps_grouped_kernel");
}
static GroupedRegistrar reg_ent_f32(KID_ENTROPY_F32, entropy_grouped_launch<false>);
static GroupedRegistrar reg_ent_u8(KID_ENTROPY_U8, entropy_grouped_launch<true>);

template <bool U8>
static int entropy_tiles_grouped_launch(const GroupRec *const *recs, int n, hipStream_t s)
{
    Grouped<EntWinArgs> g;
    size_t lds;
    int rc = fill_grouped(recs, n, &g, &lds);
    if (rc) return rc;
    hipLaunchKernelGGL(entropy_tiles_grouped_kernel<U8>, dim3(g.start[kMaxGroups]), dim3(kEntThreads), 0, s, g);
    return launch_check("entropy_tiles_grouped_kernel");
}
static GroupedRegistrar reg_ent_win_f32(KID_ENTROPY_WIN_F32, entropy_tiles_grouped_launch<false>);
static GroupedRegistrar reg_ent_win_u8(KID_ENTROPY_WIN_U8, entropy_tiles_grouped_launch<true>);

extern "C" int cgic_entropy_maps_tiles(const void *src, int is_u8, int64_t N, int64_t H, int64_t W, int T, const int *origins,
                                       int64_t th, int64_t tw, const float *bins, int nbins, float sigma, float *x_out,
                                       float *e8, float *e16, float *flat8, cgic_stream_t stream)
{
    CGIC_REQUIRE(src && bins && origins && x_out, CGIC_ERR_INVALID, "entropy_maps_tiles: src, bins, origins and x_out must not be NULL");
    CGIC_REQUIRE(nbins == kBins, CGIC_ERR_UNSUPPORTED, "entropy: nbins=%d; the reference uses 32 (model.py:480)", nbins);
    CGIC_REQUIRE(N >= 0 && H > 0 && W > 0 && H < (1 << 30) && W < (1 << 30), CGIC_ERR_INVALID, "entropy_maps_tiles: bad source shape");
    CGIC_REQUIRE(T >= 1 && T <= kEntMaxTiles, CGIC_ERR_UNSUPPORTED, "entropy_maps_tiles: %d tiles per image in this group (1..%d): cut them with cgic_cut_tiles",
                 T, kEntMaxTiles);
    CGIC_REQUIRE(th > 0 && tw > 0 && th % 16 == 0 && tw % 16 == 0, CGIC_ERR_INVALID,
                 "entropy_maps_tiles: tile %lldx%lld must be positive multiples of 16", (long long)th, (long long)tw);
    CGIC_REQUIRE(N * T <= 65535 && th / 16 <= 65535, CGIC_ERR_UNSUPPORTED, "entropy: batch/height exceed the grid limits");
    CGIC_REQUIRE(sigma > 0.f && sigma <= 0.0105f, CGIC_ERR_UNSUPPORTED,
                 "entropy: sigma=%g; the 2-bin window assumes the reference's sigma=0.01 (model.py:481)", sigma);
    for (int i = 1; i < kBins; ++i)
        CGIC_REQUIRE(fabsf((bins[i] - bins[i - 1]) - 2.0f / 31.0f) < 1e-5f, CGIC_ERR_UNSUPPORTED, "entropy: bins are not linspace(-1, 1, 32)");
    CGIC_REQUIRE((reinterpret_cast<uintptr_t>(x_out) & 15) == 0 && (!is_u8 || (reinterpret_cast<uintptr_t>(src) & 3) == 0), CGIC_ERR_INVALID,
                 "entropy_maps_tiles: x_out must be 16-byte aligned (uint8 frames 4-byte aligned)");
    if (N == 0) return CGIC_OK;
    EntWinArgs a;
    memcpy(a.e.bins.v, bins, sizeof(a.e.bins.v));
#ifndef CGIC_ENT_TILES_PPW
#define CGIC_ENT_TILES_PPW 4
#endif
    a.e.x = nullptr; a.e.H = th; a.e.W = tw; a.e.ppw = CGIC_ENT_TILES_PPW; a.e.e8 = e8; a.e.e16 = e16; a.e.x_out = x_out; a.e.flat8 = flat8;
    a.e.exp2_scale = (float)(-0.5 * 1.4426950408889634 / ((double)sigma * (double)sigma));
    a.w.src = src; a.w.srcH = (int)H; a.w.srcW = (int)W; a.w.T = T;
    for (int k = 0; k < kEntMaxTiles; ++k) {
        const int kk = k < T ? k : T - 1;
        a.w.org[k][0] = origins[2 * kk]; a.w.org[k][1] = origins[2 * kk + 1];
        CGIC_REQUIRE(abs(a.w.org[k][0]) < (1 << 29) && abs(a.w.org[k][1]) < (1 << 29), CGIC_ERR_INVALID, "entropy_maps_tiles: tile origin out of range");
    }
    const int64_t per_wg = (int64_t)kEntWaves * a.e.ppw;
    const dim3 grid((unsigned)((tw / 16 + per_wg - 1) / per_wg), (unsigned)(th / 16), (unsigned)(N * T));
    hipStream_t s = (hipStream_t)stream;
    if (is_u8)
        return launch_or_record(KID_ENTROPY_WIN_U8, grid, dim3(kEntThreads), 0, a, s, [=] {
            hipLaunchKernelGGL(entropy_tiles_kernel<true>, grid, dim3(kEntThreads), 0, s, a);
            return launch_check("entropy_tiles_kernel"); });
    return launch_or_record(KID_ENTROPY_WIN_F32, grid, dim3(kEntThreads), 0, a, s, [=] {
        hipLaunchKernelGGL(entropy_tiles_kernel<false>, grid, dim3(kEntThreads), 0, s, a);
        return launch_check("entropy_tiles_kernel"); });
}

extern "C" int cgic_entropy_maps_f32(const float *x, int64_t B, int64_t H, int64_t W, const float *bins,
                                     int nbins, float sigma, float *e8, float *e16, float *flat8, cgic_stream_t stream)
{
    return entropy_maps_launch(x, false, B, H, W, bins, nbins, sigma, nullptr, e8, e16, flat8, stream);
}

extern "C" int cgic_entropy_maps_u8(const unsigned char *x_hwc, int64_t B, int64_t H, int64_t W, const float *bins,
                                    int nbins, float sigma, float *x_out, float *e8, float *e16, float *flat8, cgic_stream_t stream)
{
    CGIC_REQUIRE(((uintptr_t)x_hwc & 3u) == 0, CGIC_ERR_INVALID, "entropy: the uint8 frame must be 4-byte aligned");
    return entropy_maps_launch(x_hwc, true, B, H, W, bins, nbins, sigma, x_out, e8, e16, flat8, stream);
}

extern "C" int cgic_entropy_maps_ref_f32(const float *x, int64_t B, int64_t H, int64_t W, const float *bins,
                                         int nbins, float sigma, float *e8, float *e16, cgic_stream_t stream)
{
    CGIC_NOT_IN_GROUP("cgic_entropy_maps_ref_f32");
    CGIC_REQUIRE(x && bins, CGIC_ERR_INVALID, "entropy: x and bins must not be NULL");
    CGIC_REQUIRE(nbins == kBins, CGIC_ERR_UNSUPPORTED, "entropy: nbins=%d; the reference uses 32 (model.py:480)", nbins);
    CGIC_REQUIRE(B >= 0 && H > 0 && W > 0 && H % 16 == 0 && W % 16 == 0, CGIC_ERR_INVALID,
                 "entropy: H=%lld W=%lld must be positive multiples of 16", (long long)H, (long long)W);
    CGIC_REQUIRE(B <= 65535 && H / 16 <= 65535, CGIC_ERR_UNSUPPORTED, "entropy: batch/height exceed the grid limits");
    CGIC_REQUIRE(sigma > 0.f && sigma <= 0.0105f, CGIC_ERR_UNSUPPORTED,
                 "entropy: sigma=%g; the five-bin window assumes the reference's sigma=0.01 (model.py:481)", sigma);
    for (int i = 1; i < kBins; ++i)
        CGIC_REQUIRE(fabsf((bins[i] - bins[i - 1]) - 2.0f / 31.0f) < 1e-5f, CGIC_ERR_UNSUPPORTED,
                     "entropy: bins are not linspace(-1, 1, 32)");
    if (B == 0 || (!e8 && !e16)) return CGIC_OK;
    BinsArg ba;
    memcpy(ba.v, bins, sizeof(ba.v));
    dim3 grid((unsigned)(W / 16), (unsigned)(H / 16), (unsigned)B);
    hipLaunchKernelGGL(entropy_ref_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, H, W, sigma, e8, e16, ba);
    return launch_check("entropy_ref_kernel");
}

