// cgic_entropy_dev.h -- the reference's OWN entropy arithmetic (Entropy, CGIC/models/model.py:433-483) as device building
// blocks, shared by the opt-in whole-map kernel (cgic_entropy.hip: entropy_ref_kernel) and the router's threshold-band
// refinement (cgic_router_dev.h): both must produce the same bits for the same patch.
//
//   gray   = (0.2989 R + 0.5870 G) + 0.1140 B                       three products, two sums, no fma (model.py:471)
//   kv     = exp(-0.5 * ((gray - bin) / sigma)^2)                    IEEE divide, fp32 square and product (:452-454);
//            exactly 0 beyond 14.42 sigma in fp32 (a < -104): at most five consecutive bins per pixel are evaluated
//   pdf    = mean over the patch's pixels in row-major order         torch's cascade sum of an outer reduction: chunks of
//            16 consecutive pixels summed one after the other from 0, the chunk sums added one after the other (:456)
//   norm   = sum over the 32 bins + 1e-40                            torch's inner reduction of 32 contiguous floats: eight
//            strided partials p_k = ((x_k + x_{8+k}) + x_{16+k}) + x_{24+k}, then p_0 + p_1 + ... + p_7 (:457)
//   q      = pdf / norm + 1e-40;  H = -sum(q log q), same 8-partial order (:458-459)
// exp / log are evaluated in fp64 and rounded once (torch's are MKL's: off the correctly rounded value in ~1 % of the
// arguments, which is what is left against the real class: 98.6-100 % of the values bit-identical, the rest <= 5e-7).
#pragma once
#include "cgic_common.h"

#include <math.h>

namespace cgic {

constexpr int kBins = 32;
constexpr int kRefWin = 5;         // bins that can be non-zero for one pixel in fp32 (exp underflows beyond 14.42 sigma)
constexpr int kRefRow = kBins + 1; // LDS row stride of a chunk-sum row (+1: bank padding)

struct BinsArg { float v[kBins]; };   // passed by value in the kernarg segment

// Pixels behind a pair of entropy maps, for re-evaluating single patches (cgic_pixels of the C ABI, checked on the host)
// (kept to 32 bytes: it rides in the kernarg segment of the fused VQ + router launch, whose VQ half is at its register cap;
// the bin centres are recomputed in the kernel -- linspace_bin() -- and checked against the caller's on the host)
struct RefineSrc {
    const void *x;      // nullptr = off.  fp32 [B,3,H,W], or uint8 [B,H,W,3] (u8 != 0)
    int u8;
    int H, W;
    float sigma;
    const float *flat8; // nullptr, or [B, H/8, W/8]: gray of an 8x8 patch whose 64 pixels all carry the same one, else NaN
};

// torch.linspace(-1, 1, 32)[i] as its CPU kernel computes it (model.py:480): step = (end - start) / (steps - 1) in fp32; the
// first half start + step * i, the second half end - step * (steps - 1 - i) (pinned against torch by the oracle's restatement)
__host__ __device__ inline float linspace_bin(int i)
{
    const float step = 2.0f / 31.0f;
    return i < kBins / 2 ? -1.0f + step * (float)i : 1.0f - step * (float)(kBins - 1 - i);
}

// byte / 255 as torch's `.div(255)` rounds it (T.ToTensor(), inference.py:50-53): q = b * fl(1/255) corrected once by the exact
// remainder -- equal to the IEEE quotient for all 256 bytes (checked exhaustively, tests/test_host_logic.py), three full-rate
// instructions instead of the ~10 of a division
__device__ __forceinline__ float unit_of_byte(float b)
{
    const float r = 0.00392156886f;                 // fl(1 / 255)
    const float q = b * r;
    return __builtin_fmaf(__builtin_fmaf(-q, 255.0f, b), r, q);
}

// torch's sum over 32 contiguous floats (see above)
__device__ __forceinline__ float sum32_lanes8(const float *v)
{
    float p[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) p[k] = ((v[k] + v[8 + k]) + v[16 + k]) + v[24 + k];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) s = s + p[k];
    return s;
}

__device__ __forceinline__ float ref_gray_of(float R, float G, float B) { return (0.2989f * R + 0.5870f * G) + 0.1140f * B; }

// gray value of pixel (b, y, x) of the source
__device__ __forceinline__ float ref_gray_at(const RefineSrc &s, int64_t b, int64_t y, int64_t x)
{
    if (s.u8) {
        const unsigned char *p = reinterpret_cast<const unsigned char *>(s.x) + ((b * s.H + y) * s.W + x) * 3;
        return ref_gray_of(unit_of_byte((float)p[0]), unit_of_byte((float)p[1]), unit_of_byte((float)p[2]));
    }
    const int64_t plane = (int64_t)s.H * s.W;
    const float *p = reinterpret_cast<const float *>(s.x) + (b * 3) * plane + y * s.W + x;
    return ref_gray_of(p[0], p[plane], p[2 * plane]);
}

// The pixel's window: the first bin that is not below gray - 0.1445 (14.42 sigma = 0.1442: exp is exactly 0 in fp32 beyond) and
// the four after it = the number of bins below that value, clamped.  `bins` is increasing, so a guess from the bin pitch fixed
// up against the bin values themselves gives exactly that count (NaN: every comparison is false -> window 0, which the NaN
// then poisons; +-inf end up clamped).
__device__ __forceinline__ int ref_window(const float *bins, float gray)
{
    const float t = gray - 0.1445f;
    float g = floorf((t - bins[0]) * 15.5f) + 1.0f;
    g = g >= 0.f ? g : 0.f;                                  // (NaN -> 0)
    g = g > (float)kBins ? (float)kBins : g;
    int j = (int)g;
    while (j > 0 && !(bins[j - 1] < t)) --j;
    while (j < kBins && bins[j] < t) ++j;
    return j > kBins - kRefWin ? kBins - kRefWin : j;
}

// x / 0.01f, correctly rounded, for 2^-100 <= |x| <= 8 in three full-rate instructions instead of the ~15 of an IEEE divide:
// q = x * fl(1/sigma), corrected once by the exact remainder (fma).  Checked EXHAUSTIVELY against x / sigma over every fp32 in that
// range, both signs (oracle/cgic_oracle.c: cgic_oracle_check_fast_div; 2 x 864 026 625 values, no mismatch).  Below 2^-100 the
// square of the quotient is 0 either way -- the only thing the caller uses -- and beyond 8 (or non-finite) the caller divides.
__host__ __device__ inline float div_by_sigma001(float x)
{
    const float sigma = 0.01f, r = 100.0f;               // fl(1 / 0.01f) == 100.0f
    const float q = x * r;
    return __builtin_fmaf(__builtin_fmaf(-q, sigma, x), r, q);
}

// the five kernel values of a pixel (model.py:452-454); v[k] belongs to bin j0 + k
__device__ __forceinline__ void ref_pixel(const float *bins, float sigma, float gray, int &j0, float v[kRefWin])
{
    j0 = ref_window(bins, gray);
    const bool fast = sigma == 0.01f;                    // (the reference's sigma, model.py:481)
#pragma unroll
    for (int k = 0; k < kRefWin; ++k) {
        const float res = gray - bins[j0 + k];
        float t = div_by_sigma001(res);
        if (!(fast && fabsf(res) < 8.0f)) t = res / sigma;
        const float t2 = t * t;
        const float a = -0.5f * t2;
        // exp(a) < 2^-150 rounds to 0 in fp32 (a < -103.98); NaN takes the exp
        v[k] = (a < -104.0f) ? 0.f : (float)exp((double)a);      // correctly rounded but for ~1e-9 of the arguments
    }
}

// ---- wave-level evaluation of 64 pixels = 4 chunks of 16 (an 8x8 patch, or a quarter -- four rows -- of a 16x16 one) -----------
constexpr int kRefRecStride = 6;                     // per pixel: window start + five values
constexpr int kRefRecFloats = 64 * kRefRecStride;    // record area of a wave
constexpr int kRefUnitRows = 4;                      // chunk-sum rows a unit produces

// Lane = pixel `lane` of the unit in row-major order of its patch (chunk = 16 consecutive lanes).  Leaves the four chunk sums
// per bin in T[chunk * kRefRow + bin]: every (chunk, bin) adds its 16 pixels one after the other from 0 -- zeros outside a
// pixel's window are added like the reference adds them (x + 0 == x).  `rec`: kRefRecFloats floats of wave-private LDS.
// OR over the 64 lanes on DPP operands + the 16- / 32-lane swaps (no LDS crossbar); every lane gets the result
__device__ __forceinline__ unsigned int wave_or_u32(unsigned int v)
{
    v |= (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);       // quad_perm [1,0,3,2]
    v |= (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);       // quad_perm [2,3,0,1]
    v |= (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true);      // row_half_mirror
    v |= (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true);      // row_mirror
    unsigned int a = v, b = v;
    swap16(a, b);
    v = a | b;
    a = v; b = v;
    swap32(a, b);
    return a | b;
}

__device__ __forceinline__ void ref_unit_chunks(float *rec, int j0, const float v[kRefWin], float *T)
{
    const int lane = lane_id();
    // NARROW unit (round 6): all 64 windows start within one bin of each other -- the nearly constant patches a threshold band of
    // smooth content is made of -- so only six bins are ever non-zero.  Their values go into a dense [pixel][6] matrix in the record
    // area (zeros where a pixel's window does not reach: exactly what the general form adds there), 24 lanes add their (chunk, bin)
    // column in pixel order, every other chunk sum is the 0 the general form arrives at by adding sixteen zeros.  Same additions in
    // the same order: the same bits (tools/probes/probe_unit.hip compares the two forms); 3 writes + 16 reads + 16 additions per
    // lane instead of 64 reads and ~250 VALU instructions: a unit 1.13 -> 0.89 us of SIMD time.
    {
        const unsigned int mask = (unsigned int)__builtin_amdgcn_readfirstlane((int)wave_or_u32(1u << j0));
        const int jmin = __builtin_ctz(mask), jmax = 31 - __builtin_clz(mask);
        if (jmax - jmin <= 1) {                                                    // (wave-uniform)
            const bool up = j0 != jmin;                                            // this pixel's window starts one bin later
            float2 w0, w1, w2;
            w0.x = up ? 0.f : v[0];  w0.y = up ? v[0] : v[1];
            w1.x = up ? v[1] : v[2]; w1.y = up ? v[2] : v[3];
            w2.x = up ? v[3] : v[4]; w2.y = up ? v[4] : 0.f;
            float2 *row = reinterpret_cast<float2 *>(rec + lane * kRefRecStride);
            row[0] = w0; row[1] = w1; row[2] = w2;
            T[(lane >> 5) * kRefRow + (lane & 31)] = 0.f;
            T[(2 + (lane >> 5)) * kRefRow + (lane & 31)] = 0.f;
            __builtin_amdgcn_wave_barrier();
            if (lane < 24) {
                const int c = lane / 6, b = lane - 6 * c;
                const float *col = rec + (16 * c) * kRefRecStride + b;
                float x[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) x[i] = col[kRefRecStride * i];
                float acc = 0.f;
#pragma unroll
                for (int i = 0; i < 16; ++i) acc = acc + x[i];
                if (jmin + b < kBins) T[c * kRefRow + jmin + b] = acc;
            }
            __builtin_amdgcn_wave_barrier();
            return;
        }
    }
    rec[lane * kRefRecStride] = __int_as_float(j0);
#pragma unroll
    for (int k = 0; k < kRefWin; ++k) rec[lane * kRefRecStride + 1 + k] = v[k];
    __builtin_amdgcn_wave_barrier();               // LDS operations of one wave execute in order
    const int bin = lane & 31;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int ch = 2 * r + (lane >> 5);
        // (all 16 window starts first, then all 16 values, then the additions in order: one addition behind two dependent LDS
        // round trips, 32 times over, was 1.1 of a unit's 2.6 us)
        int d[16];
        float val[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) d[i] = bin - __float_as_int(rec[(16 * ch + i) * kRefRecStride]);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int dc = d[i] < 0 ? 0 : (d[i] > kRefWin - 1 ? kRefWin - 1 : d[i]);
            val[i] = rec[(16 * ch + i) * kRefRecStride + 1 + dc];
        }
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc = acc + (((unsigned int)d[i] < (unsigned int)kRefWin) ? val[i] : 0.f);
        T[ch * kRefRow + bin] = acc;
    }
    __builtin_amdgcn_wave_barrier();
}

// acc + the unit's four chunk sums of this lane's bin, in chunk order: the running sum over a patch's chunks (the second level of
// torch's cascade sum).  rows: which of T's rows to add, in order (0..3 for a unit; a constant 16x16 patch adds them four times).
__device__ __forceinline__ float ref_add_rows(float acc, const float *T, int nrows = kRefUnitRows, int rowmask = 3)
{
    const int bin = lane_id() & 31;
    for (int c = 0; c < nrows; ++c) acc = acc + T[(c & rowmask) * kRefRow + bin];
    return acc;
}

// The entropy of a patch from the sum over its chunks (lane = bin, both half-waves alike), npix = 64 or 256; all 64 lanes call,
// the result is wave-uniform.  `P`: 64 floats of wave-private LDS.
__device__ __forceinline__ float ref_finalize(float acc, int npix, float *P)
{
    const int lane = lane_id();
    const int bin = lane & 31;
    const float pdf = acc / (npix == 64 ? 64.0f : 256.0f);       // torch.mean: the sum divided by the count (exact: a power of two)
    if (lane < kBins) P[bin] = pdf;
    __builtin_amdgcn_wave_barrier();
    const float eps = 1e-40f;
    const float norm = sum32_lanes8(P) + eps;
    const float q = pdf / norm + eps;
    const float t = q * (float)log((double)q);
    if (lane < kBins) P[kBins + bin] = t;
    __builtin_amdgcn_wave_barrier();
    const float ent = -sum32_lanes8(P + kBins);
    __builtin_amdgcn_wave_barrier();
    return ent;
}

}  // namespace cgic
