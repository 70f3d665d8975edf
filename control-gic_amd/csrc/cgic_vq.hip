// cgic_vq.hip -- fused distance + argmin (+ z_q, loss, usage histogram) for
// VectorQuantize2.forward (reference: CGIC/modules/vqvae/quantize.py:69-97).
//
// The reference materialises d[N,K] = sum(z^2) + sum(e^2) - 2 z.e^T in HBM and
// argmins it.  Here the [K,4] codebook and its row norms live in LDS, the
// 4-deep contraction runs on the fp32 matrix cores (v_mfma_f32_16x16x4_f32:
// one instruction = 16 codes x 16 latent vectors x K=4, bit-for-bit an fmaf
// chain in k order, i.e. exactly the CPU reference's sgemm rounding), and the
// add / fma / running-min epilogue runs on the VALU while the next MFMA is in
// flight.  No N x K matrix ever exists.
//
// Rounding contract (checked against the reference through tests/golden/vq.npz):
//   zz = ((z0^2 + z1^2) + z2^2) + z3^2           ee likewise per codebook row
//   mm = fma(z3,e3, fma(z2,e2, fma(z1,e1, z0*e0)))
//   d  = fl(fl(zz + ee) - 2*mm)                   argmin, lowest index on ties
// Compiled with -ffp-contract=off; every fused op is an explicit fmaf / MFMA.
#include "cgic_router_dev.h"

#include <stdlib.h>

#include <map>
#include <mutex>

namespace cgic {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kVqThreads = 256;   // 4 waves
constexpr int kVqMaxK = 8192;

__device__ __forceinline__ float sumsq4(float a, float b, float c, float d)
{
    float s = a * a;
    s = s + b * b;
    s = s + c * c;
    s = s + d * d;
    return s;
}

// Stage the codebook in LDS, transposed to [4][K] so that the MFMA A-operand
// read (16 consecutive codes, fixed k) is conflict-free, plus ee[K].
__device__ __forceinline__ void stage_codebook(const float *__restrict__ cb, int K, float *cbT, float *ee)
{
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        float4 e = reinterpret_cast<const float4 *>(cb)[k];
        cbT[0 * K + k] = e.x;
        cbT[1 * K + k] = e.y;
        cbT[2 * K + k] = e.z;
        cbT[3 * K + k] = e.w;
        ee[k] = sumsq4(e.x, e.y, e.z, e.w);
    }
}

// distance of latent (z0..z3, zz) to code `c`, the reference's rounding sequence, on the VALU
__device__ __forceinline__ float dist_valu(float z0, float z1, float z2, float z3, float zz,
                                           const float *cbT, const float *ee, int K, int c)
{
    float mm = z0 * cbT[c];
    mm = __builtin_fmaf(z1, cbT[K + c], mm);
    mm = __builtin_fmaf(z2, cbT[2 * K + c], mm);
    mm = __builtin_fmaf(z3, cbT[3 * K + c], mm);
    return __builtin_fmaf(-2.0f, mm, zz + ee[c]);
}

// Last-arriving block sums the per-block partials in a fixed order (deterministic whichever
// block is last) and writes loss = m + beta*m (quantize.py:85-90).  Hand-off per
// cdna_hip_programming.md G16 form R1: the 8-byte partial is stored write-through (relaxed
// agent-scope atomic store = sc1), drained with s_waitcnt, then the ticket is taken -- no
// release fence: a per-workgroup buffer_wbl2 measured ~10 us per launch at 1024 workgroups
// (and 60 us at 4096).  The last block does one agent acquire and reads with sc1 loads.  `ticket` lives in library-owned device memory,
// zeroed once at allocation; the last block resets it, so no per-call memset node is needed.
template <int NT>
__device__ __forceinline__ void finish_loss(double block_sum, double *sq_partial, unsigned int *ticket,
                                            double count, float beta, int legacy, float *loss,
                                            unsigned int blk, unsigned int nblk)
{
    __shared__ double red[NT];
    __shared__ unsigned int s_last;
    const int tid = threadIdx.x;
    if (tid == 0) {
        __hip_atomic_store(&sq_partial[blk], block_sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        s_last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nblk - 1;
    }
    __syncthreads();
    if (!s_last) return;
    if (tid == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    __syncthreads();
    double a = 0.0;
    for (unsigned int i = tid; i < nblk; i += NT)
        a += __hip_atomic_load(&sq_partial[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    red[tid] = a;
    __syncthreads();
    for (int off = NT / 2; off > 0; off >>= 1) {
        if (tid < off) red[tid] += red[tid + off];
        __syncthreads();
    }
    if (tid == 0) {
        const float m = (float)(red[0] / count);
        *loss = legacy ? (m + beta * m) : (beta * m + m);
        __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
    }
}

// Variant for the filter path, executed by ONE wave: lane 0 publishes the workgroup's partial (write-through
// store, drained, then the ticket); the wave of the last workgroup sums all partials in a fixed order.
__device__ __forceinline__ void finish_loss_wave(double block_sum, double *sq_partial, unsigned int *ticket, double count,
                                                 float beta, int legacy, float *loss, unsigned int blk, unsigned int nblk)
{
    const int lane = lane_id();
    int last = 0;
    if (lane == 0) {
        __hip_atomic_store(&sq_partial[blk], block_sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nblk - 1;
    }
    last = __builtin_amdgcn_readfirstlane(last);
    if (!last) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    double a = 0.0;
    for (unsigned int i = lane; i < nblk; i += kWave)
        a += __hip_atomic_load(&sq_partial[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off, kWave);
    if (lane == 0) {
        const float m = (float)(a / count);
        *loss = legacy ? (m + beta * m) : (beta * m + m);
        __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
    }
}

// ZT = latent tiles (of 16 vectors) per wave; a wave owns 16*ZT vectors and scans all K codes;
// a block owns 4 * 16 * ZT vectors.  __launch_bounds__(256, 2): a <=256-VGPR budget makes hipcc
// pick the VGPR-destination MFMA form (no v_accvgpr_read per output).
//
// Epilogue per MFMA (4 outputs per lane): 4 add + 4 fma + 2 v_min3 + 1 cmp + 1 cndmask.  Only the
// running minimum VALUE and the index of the 16-code TILE where it last strictly decreased are
// tracked; which of the lane's 4 rows achieved it is resolved once at the end by recomputing that
// one tile on the VALU (bit-identical to the MFMA: same fmaf chain) and taking the first row equal
// to the minimum.  Strict '<' on tiles + first-equal on rows + (value, index) lexicographic merge
// across the 4 row groups == lowest index among exact minima, like torch.argmin.
struct VqArgs {
    const float *z;
    int64_t hw, N;
    const float *cb;
    int K;
    int64_t *idx_out;
    float *zq_out;
    double *sq_partial;
    unsigned int *ticket;
    float beta;
    int legacy;
    float *loss;
    unsigned int nblk;        // VQ workgroups (a fused launch has router workgroups beside them)
    // filter path: groups per workgroup.  Workgroups [0, n_early) own `g_early` groups each, the rest `g_late`
    // (router workgroups in front of a fused launch delay the VQ workgroups that have to wait for their CUs)
    unsigned int n_early, g_early, g_late;
};

template <int ZT>
__device__ __forceinline__ void vq_mfma_body(const VqArgs &a, float *smem)
{
    const float *__restrict__ z = a.z;
    const int64_t hw = a.hw, N = a.N;
    const float *__restrict__ cb = a.cb;
    const int K = a.K;
    int64_t *__restrict__ idx_out = a.idx_out;
    float *__restrict__ zq_out = a.zq_out;
    double *__restrict__ sq_partial = a.sq_partial;
    float *cbT = smem;                // [4][K]
    float *ee = smem + 4 * K;         // [K]

    CGIC_STAMP(0);
    CGIC_BLK_BEGIN();
    stage_codebook(cb, K, cbT, ee);
    __syncthreads();
    CGIC_STAMP(1);

    const int lane = lane_id();
    const int wave = threadIdx.x >> 6;
    const int j = lane & 15;   // column: which latent vector of the tile
    const int g = lane >> 4;   // B-operand k index / C-row group
    const int64_t wave_base = ((int64_t)blockIdx.x * 4 + wave) * (16 * ZT);

    // (image, position) of the wave's first vector: ONE 64-bit division per wave; every other
    // address follows by adding and carrying (a 64-bit divide costs ~100 instructions, and the
    // previous per-tile divides in prologue + epilogue were a quarter of the kernel's issue time)
    const int64_t b0 = wave_base / hw;
    const int64_t p0 = wave_base - b0 * hw;
    auto locate = [&](int t, int64_t *b, int64_t *p) {
        int64_t pp = p0 + 16 * t + j, bb = b0;
        while (pp >= hw) { pp -= hw; ++bb; }
        *b = bb; *p = pp;
    };
    // B operand: lane holds z[n_j][k=g] for each tile; zz per column.
    float zv[ZT], zz[ZT], best[ZT];
    int bt[ZT];
#pragma unroll
    for (int t = 0; t < ZT; ++t) {
        int64_t n = wave_base + 16 * t + j;
        float v = 0.f;
        if (n < N) {
            int64_t b, p;
            locate(t, &b, &p);
            v = z[(b * 4 + g) * hw + p];
        }
        zv[t] = v;
        float z0 = __shfl(v, j, kWave), z1 = __shfl(v, 16 + j, kWave);
        float z2 = __shfl(v, 32 + j, kWave), z3 = __shfl(v, 48 + j, kWave);
        zz[t] = sumsq4(z0, z1, z2, z3);
        best[t] = __builtin_inff();
        bt[t] = 0;
    }

    CGIC_STAMP(2);
    const int ntile = K >> 4;
    for (int ct = 0; ct < ntile; ++ct) {
        // A operand: A[i = lane&15][k = lane>>4] = e[16*ct + i][k]
        const float a = cbT[g * K + 16 * ct + j];
        // C rows held by this lane: codes 16*ct + 4*g + r, r = 0..3
        const f32x4 e4 = *reinterpret_cast<const f32x4 *>(&ee[16 * ct + 4 * g]);
#pragma unroll
        for (int t = 0; t < ZT; ++t) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, zv[t], acc, 0, 0, 0);
            const float d0 = __builtin_fmaf(-2.0f, acc[0], zz[t] + e4[0]);   // fl(fl(zz+ee) - 2*mm)
            const float d1 = __builtin_fmaf(-2.0f, acc[1], zz[t] + e4[1]);
            const float d2 = __builtin_fmaf(-2.0f, acc[2], zz[t] + e4[2]);
            const float d3 = __builtin_fmaf(-2.0f, acc[3], zz[t] + e4[3]);
            const float m1 = __builtin_fminf(__builtin_fminf(best[t], d0), d1);   // v_min3_f32
            const float m2 = __builtin_fminf(__builtin_fminf(m1, d2), d3);
            bt[t] = m2 < best[t] ? ct : bt[t];
            best[t] = m2;
        }
    }

    CGIC_STAMP(3);
    // Resolve the row inside the winning tile, then combine the 4 row groups
    // (lanes j, j+16, j+32, j+48): lexicographic (d, index).
    double sq = 0.0;
#pragma unroll
    for (int t = 0; t < ZT; ++t) {
        const float v = zv[t];
        const float z0 = __shfl(v, j, kWave), z1 = __shfl(v, 16 + j, kWave);
        const float z2 = __shfl(v, 32 + j, kWave), z3 = __shfl(v, 48 + j, kWave);
        const int c0 = 16 * bt[t] + 4 * g;
        float d = best[t];
        int i = c0;
        {
            // the lane's 4 candidate codes c0..c0+3 are contiguous in the transposed codebook:
            // five 16-byte LDS reads instead of twenty 4-byte ones
            const f32x4 e0 = *reinterpret_cast<const f32x4 *>(&cbT[c0]);
            const f32x4 e1 = *reinterpret_cast<const f32x4 *>(&cbT[K + c0]);
            const f32x4 e2 = *reinterpret_cast<const f32x4 *>(&cbT[2 * K + c0]);
            const f32x4 e3 = *reinterpret_cast<const f32x4 *>(&cbT[3 * K + c0]);
            const f32x4 en = *reinterpret_cast<const f32x4 *>(&ee[c0]);
#pragma unroll
            for (int r = 3; r >= 0; --r) {   // descending: the lowest matching row wins
                float mm = z0 * e0[r];
                mm = __builtin_fmaf(z1, e1[r], mm);
                mm = __builtin_fmaf(z2, e2[r], mm);
                mm = __builtin_fmaf(z3, e3[r], mm);
                i = __builtin_fmaf(-2.0f, mm, zz[t] + en[r]) == d ? c0 + r : i;
            }
        }
#pragma unroll
        for (int off = 16; off < 64; off <<= 1) {
            float od = __shfl_xor(d, off, kWave);
            int oi = __shfl_xor(i, off, kWave);
            // non-finite distances are outside the contract
            bool take = od < d || (od == d && oi < i);
            d = take ? od : d;
            i = take ? oi : i;
        }
        int64_t n = wave_base + 16 * t + j;
        if (n < N) {
            if (zq_out || sq_partial) {
                // this lane owns channel g of vector n
                float e = cbT[g * K + i];
                float diff = e - zv[t];
                if (zq_out) {
                    int64_t b, p;
                    locate(t, &b, &p);
                    zq_out[(b * 4 + g) * hw + p] = zv[t] + diff;
                }
                sq += (double)diff * (double)diff;
            }
            if (g == 0 && idx_out) idx_out[n] = (int64_t)i;
        }
    }

    CGIC_STAMP(4);
    if (sq_partial) {
        // deterministic block reduction: fixed shuffle tree, then waves in order
        __shared__ double wsum[4];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sq += __shfl_down(sq, off, kWave);
        if (lane == 0) wsum[wave] = sq;
        __syncthreads();
        finish_loss<kVqThreads>(((wsum[0] + wsum[1]) + wsum[2]) + wsum[3], sq_partial, a.ticket, (double)N * 4.0, a.beta, a.legacy, a.loss,
                    blockIdx.x, a.nblk);
    }
    CGIC_STAMP(5);
    CGIC_BLK_END();
}

template <int ZT>
__global__ __launch_bounds__(kVqThreads, ZT <= 4 ? 4 : 2) void vq_mfma_kernel(VqArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    vq_mfma_body<ZT>(a, smem);
}

// Horizontal fusion: the per-image router workgroups ride behind the VQ workgroups of the same launch.
// The router is latency-bound (one workgroup per image, ~10 us of barriers and LDS sweeps) and needs
// nothing from the VQ; VQ needs nothing from the router; both only need the kernels before them.  As
// extra workgroups of this grid the router runs in the shadow of the ~45 us VQ instead of occupying its
// own ~15 us slot on the stream (graph branches were measured to cost more than they hide).
template <int ZT>
__global__ __launch_bounds__(kVqThreads, ZT <= 4 ? 4 : 2) void vq_router_kernel(VqArgs a, RouterArgs r)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    if (blockIdx.x >= a.nblk) {
        router_body<kVqThreads>(r, (int64_t)(blockIdx.x - a.nblk), reinterpret_cast<unsigned char *>(smem));
        return;
    }
    vq_mfma_body<ZT>(a, smem);
}

// =====================================================================================================
// Filter path (K % 64 == 0, K <= 1024; the reference's codebook is 1024 x 4): the bf16 matrix cores find
// the candidates, fp32 decides.  The exact loop above costs ~7 issue slots per (16 codes x 16 vectors)
// tile, 4 of them the fp32 MFMA (which does not overlap VALU work on gfx950); it cannot be made cheaper
// and still deliver every distance.  But argmin only needs the distances that can win:
//
//  * score  f(k) = ee_k - 2 z.e_k  (the distance minus zz, which does not change the argmin) comes out of
//    ONE v_mfma_f32_16x16x32_bf16 per tile: z_j, -2 e_kj and ee_k are each split exactly into three bf16
//    (8+8+8 significand bits, by truncation); the 32 K-slots carry, per dimension, the six products
//    zh.wh zh.wm zm.wh zh.wl zl.wh zm.wm, plus ee's three pieces against 1.0.  Dropped terms and fp32
//    accumulation leave |f - F| <= 2^-17.6 S, S = max ee + 2 max|e| sum|z_j| (measured: ~2^-22 S).
//  * per lane, a running (smallest, second smallest) over QUADS of tiles (64 codes), the quad's index packed into the
//    low mantissa bits of its minimum: 8 v_min3 + and_or + med3 + min for 4 MFMAs -- 2.75 VALU per tile instead
//    of 12, and an 8 ns MFMA instead of 14.5.
//  * every code whose reference distance could be minimal has f <= f_min + M,
//    M = 2 (|f - F| + |d_ref - zz - F|) + packing <= 1.6e-5 S + 2.5e-7 zz  (d_ref's own rounding: 2^-23 zz + 2^-21 S).
//    S is the smaller of the codebook-maxima bound and (|z| + sqrt D)(3|z| + sqrt D), D = zz + f_min + slack (the
//    winner and whatever can beat it lie within sqrt D of z).
//    If the winner's quad is the only place holding such codes (second-smallest quad value of every
//    row group above the threshold, one hot row group), its 16 codes are evaluated with the exact fp32
//    sequence and the lowest-index minimum is the reference's argmin.  Otherwise (0.1-0.5 % of N(0,1)
//    vectors) the wave scans all K codes exactly for that vector; a group with many such vectors reruns
//    the exact fp32-MFMA loop.  Results are bit-identical to the exact kernels for every finite input.
// =====================================================================================================
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kVqfThreads = 512;               // 8 waves, one workgroup per CU (84 KB of LDS at K = 1024)
constexpr int kVqfWaves = kVqfThreads / 64;
constexpr int kVqfMaxK = 1024;
constexpr int kVqfTbFloats = 1024 + 64;        // per wave: 4 fields x 64 vectors x 4 row groups, + 64 results
constexpr int kVqfBulk = 12;                   // more flagged vectors than this in a 64-vector group: rerun it exactly on the MFMA

// x == h + m + l exactly, three bf16 by truncation (|m| < 2^-7 |x|, |l| < 2^-14 |x|)
__device__ __forceinline__ void split3(float x, unsigned int &h, unsigned int &m, unsigned int &l)
{
    const unsigned int xb = __float_as_uint(x);
    h = xb >> 16;
    const float r1 = x - __uint_as_float(xb & 0xFFFF0000u);
    const unsigned int rb = __float_as_uint(r1);
    m = rb >> 16;
    const float r2 = r1 - __uint_as_float(rb & 0xFFFF0000u);
    l = __float_as_uint(r2) >> 16;
}

// Cross-row-group exchanges on the VALU (gfx950 v_permlane16_swap / v_permlane32_swap) -- no LDS round trip.
// Lane = (column j = lane & 15, row group g = lane >> 4).
//   rows16(x): .x = x of row groups (0,0,2,2), .y = x of row groups (1,1,3,3)
//   rows32(x): .x = x of the lower 32 lanes in both halves, .y = the upper 32 lanes' in both halves
__device__ __forceinline__ uint2 rows16(unsigned int x)
{
    unsigned int a = x, b = x;
    swap16(a, b);
    return make_uint2(a, b);
}
__device__ __forceinline__ uint2 rows32(unsigned int x)
{
    unsigned int a = x, b = x;
    swap32(a, b);
    return make_uint2(a, b);
}
// reduce over the 4 row groups of each column; every lane of the column gets the result
__device__ __forceinline__ float colmin(float x)
{
    uint2 a = rows16(__float_as_uint(x));
    x = __builtin_fminf(__uint_as_float(a.x), __uint_as_float(a.y));
    a = rows32(__float_as_uint(x));
    return __builtin_fminf(__uint_as_float(a.x), __uint_as_float(a.y));
}
__device__ __forceinline__ int colmax(int x)
{
    uint2 a = rows16((unsigned int)x);
    x = (int)a.x > (int)a.y ? (int)a.x : (int)a.y;
    a = rows32((unsigned int)x);
    return (int)a.x > (int)a.y ? (int)a.x : (int)a.y;
}
__device__ __forceinline__ int colsum(int x)
{
    uint2 a = rows16((unsigned int)x);
    x = (int)(a.x + a.y);
    a = rows32((unsigned int)x);
    return (int)(a.x + a.y);
}
// lexicographic (d, i) minimum over the 4 row groups of each column
__device__ __forceinline__ void colargmin(float &d, int &i)
{
    uint2 dd = rows16(__float_as_uint(d)), ii = rows16((unsigned int)i);
    {
        const float da = __uint_as_float(dd.x), db = __uint_as_float(dd.y);
        const int ia = (int)ii.x, ib = (int)ii.y;
        const bool take = db < da || (db == da && ib < ia);
        d = take ? db : da;
        i = take ? ib : ia;
    }
    dd = rows32(__float_as_uint(d));
    ii = rows32((unsigned int)i);
    {
        const float da = __uint_as_float(dd.x), db = __uint_as_float(dd.y);
        const int ia = (int)ii.x, ib = (int)ii.y;
        const bool take = db < da || (db == da && ib < ia);
        d = take ? db : da;
        i = take ? ib : ia;
    }
}

// the reference's rounding sequence for one codebook row
__device__ __forceinline__ float dist_row(float z0, float z1, float z2, float z3, float zz, const float4 e, float ee)
{
    float mm = z0 * e.x;
    mm = __builtin_fmaf(z1, e.y, mm);
    mm = __builtin_fmaf(z2, e.z, mm);
    mm = __builtin_fmaf(z3, e.w, mm);
    return __builtin_fmaf(-2.0f, mm, zz + ee);
}

template <int ZT>
__device__ __forceinline__ void vq_filter_body(const VqArgs &a, unsigned char *smem, const unsigned int vblk)
{
    constexpr int NT = kVqfThreads, NW = kVqfWaves;
    static_assert(ZT <= 4, "the decide step maps one lane to each of the group's 16 * ZT vectors");
    const float *__restrict__ z = a.z;
    const int64_t hw = a.hw, N = a.N;
    const int K = a.K, ntile = K >> 4, np = K >> 5;
    int64_t *__restrict__ idx_out = a.idx_out;
    float *__restrict__ zq_out = a.zq_out;
    // A operands, 16 bytes per (tile, lane): {u0, u1, u0, u0} with u0 = wh | wm << 16, u1 = wl | ee piece << 16,
    // against B = {zh|zh, zh|1.0, zm|zm, zl|0}: slots wh.zh wm.zh | wl.zh ee.1 | wh.zm wm.zm | wh.zl 0.
    // (8 bytes per lane + two v_mov per tile was measured 1 us slower; LDS size is not what limits residency.)
    uint4 *ldsA = reinterpret_cast<uint4 *>(smem);                         // [K/16][64]
    float4 *cbs = reinterpret_cast<float4 *>(smem + (size_t)K * 64);       // [K] fp32 rows
    float *ees = reinterpret_cast<float *>(smem + (size_t)K * 80);         // [K] their squared norms
    float *tbuf = reinterpret_cast<float *>(smem + (size_t)K * 84);        // [NW][kVqfTbFloats] per-wave transpose buffers
    double *gsum = reinterpret_cast<double *>(smem + (size_t)K * 84 + (size_t)NW * kVqfTbFloats * 4);      // [groups of this workgroup] loss partials
    __shared__ unsigned int s_max[2];
    __shared__ unsigned int s_next;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15;   // column: which latent vector of the tile
    const int g = lane >> 4;   // K-slot group of the operands (= dimension) / row group of the result

    CGIC_STAMP(0);
    CGIC_BLK_BEGIN();
    // ---- groups of 16*ZT vectors: the workgroup owns a contiguous range, its waves take groups from a shared
    // counter.  (Static shares leave the SIMD's younger wave behind: VALU issue is arbitrated by age, the older
    // wave finishes early and the younger one then runs alone at half the issue rate.)
    const int64_t ngroups = (N + 16 * ZT - 1) / (16 * ZT);
    int64_t blk_lo, blk_hi;
    if (vblk < a.n_early) {
        blk_lo = (int64_t)vblk * a.g_early;
        blk_hi = blk_lo + a.g_early;
    } else {
        blk_lo = (int64_t)a.n_early * a.g_early + (int64_t)(vblk - a.n_early) * a.g_late;
        blk_hi = blk_lo + a.g_late;
    }
    blk_lo = blk_lo < ngroups ? blk_lo : ngroups;
    blk_hi = blk_hi < ngroups ? blk_hi : ngroups;
    auto grab = [&]() -> int64_t {
        int v = 0;
        if (lane == 0) v = (int)atomicAdd(&s_next, 1u);
        return blk_lo + __builtin_amdgcn_readfirstlane(v);
    };
    // (image, position) of a group's first vector
    auto origin = [&](int64_t grp, int64_t *b, int64_t *p) {
        const int64_t n0 = grp * (16 * ZT);
        if (((n0 | hw) >> 32) == 0) {
            const unsigned int q = (unsigned int)n0 / (unsigned int)hw;
            *b = q; *p = (int64_t)((unsigned int)n0 - q * (unsigned int)hw);
        } else {
            *b = n0 / hw; *p = n0 - *b * hw;
        }
    };
    auto locate = [&](int64_t bb, int64_t pp, int t, int64_t *b, int64_t *p) {
        pp += 16 * t + j;
        while (pp >= hw) { pp -= hw; ++bb; }
        *b = bb; *p = pp;
    };
    auto load_group = [&](int64_t grp, int64_t bb, int64_t pp, float (&out)[ZT]) {
#pragma unroll
        for (int t = 0; t < ZT; ++t) {
            const int64_t n = grp * (16 * ZT) + 16 * t + j;
            float v = 0.f;
            if (n < N) {
                int64_t b, p;
                locate(bb, pp, t, &b, &p);
                v = z[(b * 4 + g) * hw + p];
            }
            out[t] = v;
        }
    };

    // the first group's latents are requested before the codebook is staged: their HBM latency hides behind it
    float zn[ZT];
    int64_t cur = blk_lo + wave, cb0 = 0, cp0 = 0;
    if (cur < blk_hi) {
        origin(cur, &cb0, &cp0);
        load_group(cur, cb0, cp0, zn);
    }


    // ---- stage: fp32 rows, then the split A operands built from them
    if (tid < 2) s_max[tid] = 0;
    if (tid == 0) s_next = NW;              // groups 0..NW-1 of the workgroup's range are the waves' first ones
    {
        // every (tile, lane) element reads its codebook row straight from global (L2): all loads of a thread are in
        // flight together and no barrier sits between the fp32 copy and the operand build (one round trip, not two)
        float emax = 0.f, eemax = 0.f;
        constexpr int kPerThread = kVqfMaxK * 4 / NT;             // 8 elements per thread at K = 1024
        float4 rows[kPerThread];
#pragma unroll
        for (int q = 0; q < kPerThread; ++q) {
            const int i = tid + q * NT;
            rows[q] = i < ntile * 64 ? reinterpret_cast<const float4 *>(a.cb)[((i >> 6) << 4) + (i & 15)] : float4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int q = 0; q < kPerThread; ++q) {
            const int i = tid + q * NT;
            if (i >= ntile * 64) break;
            const int l = i & 63, gg = l >> 4;
            const float4 e = rows[q];
            const float ee = sumsq4(e.x, e.y, e.z, e.w);
            const float ec = gg == 0 ? e.x : gg == 1 ? e.y : gg == 2 ? e.z : e.w;
            unsigned int wh, wm, wl, eh, em, el;
            split3(-2.0f * ec, wh, wm, wl);
            split3(ee, eh, em, el);
            const unsigned int ep = gg == 0 ? eh : gg == 1 ? em : gg == 2 ? el : 0u;
            ldsA[i] = make_uint4(wh | (wm << 16), wl | (ep << 16), wh | (wm << 16), wh | (wm << 16));
            if (gg == 0) { ees[((i >> 6) << 4) + (l & 15)] = ee; cbs[((i >> 6) << 4) + (l & 15)] = e; }
            emax = fmaxf(emax, fabsf(ec));
            eemax = fmaxf(eemax, ee);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            emax = fmaxf(emax, __shfl_xor(emax, off, kWave));
            eemax = fmaxf(eemax, __shfl_xor(eemax, off, kWave));
        }
        // non-negative floats order like their bit patterns; a NaN lands above every finite value
        if (lane == 0) { atomicMax(&s_max[0], __float_as_uint(emax)); atomicMax(&s_max[1], __float_as_uint(eemax)); }
    }
    __syncthreads();
    const float Emax = __uint_as_float(s_max[0]), EEmax = __uint_as_float(s_max[1]);
    CGIC_STAMP(1);

    while (cur < blk_hi) {
        const int64_t grp = cur;
        const int64_t base = grp * (16 * ZT);
        const int64_t gb = cb0, gp = cp0;
        float zv[ZT], m1[ZT], m2[ZT];
        bf16x8 bop[ZT];
#pragma unroll
        for (int t = 0; t < ZT; ++t) {
            zv[t] = zn[t];
            unsigned int h, m, l;
            split3(zv[t], h, m, l);
            uint4 bb;
            bb.x = h | (h << 16);
            bb.y = h | (0x3F80u << 16);           // 1.0 against ee's piece
            bb.z = m | (m << 16);
            bb.w = l;                             // slot 7 = 0
            bop[t] = __builtin_bit_cast(bf16x8, bb);
            m1[t] = __builtin_inff();
            m2[t] = __builtin_inff();
        }
        // reserve the next group now: its latents are in flight during this group's scan
        cur = grab();
        if (cur < blk_hi) {
            origin(cur, &cb0, &cp0);
            load_group(cur, cb0, cp0, zn);
        }
        CGIC_STAMP(2);

        // ---- scan: QUADS of code tiles (64 codes), as two ping-pong pairs -- the MFMAs of one pair run while the
        // VALU digests the other.  One running-minimum chain per quad, one (smallest, second, where) update per quad.
        f32x4 X0[ZT], X1[ZT], Y0[ZT], Y1[ZT];
        float uq[ZT];
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
        auto issue = [&](int p, f32x4 (&A0)[ZT], f32x4 (&A1)[ZT]) {
            const int pp = p < np ? p : np - 1;
            const bf16x8 a0 = __builtin_bit_cast(bf16x8, ldsA[(2 * pp) * 64 + lane]);
            const bf16x8 a1 = __builtin_bit_cast(bf16x8, ldsA[(2 * pp + 1) * 64 + lane]);
#pragma unroll
            for (int t = 0; t < ZT; ++t) {
                A0[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, bop[t], zero4, 0, 0, 0);
                A1[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, bop[t], zero4, 0, 0, 0);
            }
        };
        auto chain = [&](float u, const f32x4 &A0, const f32x4 &A1) -> float {
            u = __builtin_fminf(__builtin_fminf(u, A0[0]), A0[1]);       // v_min3_f32 on the raw MFMA outputs
            u = __builtin_fminf(__builtin_fminf(u, A0[2]), A0[3]);
            u = __builtin_fminf(__builtin_fminf(u, A1[0]), A1[1]);
            return __builtin_fminf(__builtin_fminf(u, A1[2]), A1[3]);
        };
        issue(0, X0, X1);
        for (int p = 0; p < np; p += 2) {
            issue(p + 1, Y0, Y1);
#pragma unroll
            for (int t = 0; t < ZT; ++t) uq[t] = chain(__builtin_inff(), X0[t], X1[t]);    // seeded with a constant: no canonicalising v_max
            issue(p + 2, X0, X1);      // the last one is a harmless repeat of the final pair
#pragma unroll
            for (int t = 0; t < ZT; ++t) {
                // the quad's index rides in the low 4 mantissa bits of its minimum (one v_and_or_b32 instead of a
                // compare + select per quad); the 2^-19 relative perturbation is part of the margin
                const float u = __uint_as_float((__float_as_uint(chain(uq[t], Y0[t], Y1[t])) & ~15u) | (unsigned int)(p >> 1));
                m2[t] = __builtin_amdgcn_fmed3f(m1[t], m2[t], u);     // second smallest quad value
                m1[t] = __builtin_fminf(m1[t], u);
            }
        }
        CGIC_STAMP(3);

        // ---- decide: ONE LANE PER VECTOR.  The scan leaves (smallest, second, where) per (vector, row group) in the
        // MFMA's lane layout; a transpose through a wave-private LDS buffer gives lane t*16+j all four row groups of
        // its vector, and the whole decision -- margin, hot row group, the 16 exact distances -- is lane-local: no
        // cross-lane reductions (they were a third of the kernel's VALU instructions).
        int win[ZT];                                   // result per tile in the (column j, row group g) layout
        {
            float *tb = tbuf + wave * kVqfTbFloats;
#pragma unroll
            for (int t = 0; t < ZT; ++t) {
                const int o = (t * 16 + j) * 4 + g;
                tb[o] = m1[t];
                tb[256 + o] = m2[t];
                tb[768 + o] = zv[t];
            }
            const float4 a1 = reinterpret_cast<const float4 *>(tb)[lane];            // m1 of row groups 0..3
            const float4 a2 = reinterpret_cast<const float4 *>(tb)[64 + lane];       // m2
            const float4 az = reinterpret_cast<const float4 *>(tb)[192 + lane];      // z0..z3
            const bool valid = lane < 16 * ZT;
            const float y0 = az.x, y1 = az.y, y2 = az.z, y3 = az.w;
            const float yy = sumsq4(y0, y1, y2, y3);
            const float mt = __builtin_fminf(__builtin_fminf(a1.x, a1.y), __builtin_fminf(a1.z, a1.w));
            // S bounds ee_k + 2 sum|z_j e_kj| for every code that can matter: by the codebook maxima, and -- the winner
            // and every code that can beat it lie within sqrt(D) of z, D = zz + f_min + slack -- by
            // (|z| + sqrt D)^2 + 2 |z| (|z| + sqrt D); the smaller of the two (v_sqrt_f32 is good to 1 ulp: x 1.001)
            const float S0 = EEmax + 2.0f * Emax * (((fabsf(y0) + fabsf(y1)) + fabsf(y2)) + fabsf(y3));
            const float M0 = 1.6e-5f * S0 + 2.5e-7f * yy + 1e-30f;
            const float D = fmaxf(yy * 1.0001f + mt + 2.0f * M0, 0.f);
            const float nz = __builtin_amdgcn_sqrtf(yy) * 1.001f, sd = __builtin_amdgcn_sqrtf(D) * 1.001f;
            const float S1 = (nz + sd) * (3.0f * nz + sd);
            const float S = S1 < S0 ? S1 : S0;                  // (a NaN S1 keeps S0)
            const float M = 1.6e-5f * S + 2.5e-7f * yy + 1e-30f;     // 1.2e-5: filter + reference rounding; 0.4e-5: the packed index
            float thr = mt + M;
            thr += fabsf(thr) * 2.4e-7f;
            // anything not comparable (NaN / Inf anywhere above) must count as "flagged": test the negation
            const bool h0 = !(a1.x > thr), h1 = !(a1.y > thr), h2 = !(a1.z > thr), h3 = !(a1.w > thr);   // best quad holds a candidate
            const bool more = !(a2.x > thr) || !(a2.y > thr) || !(a2.z > thr) || !(a2.w > thr);        // ... and so does another quad
            // settled iff exactly one hot row group and no second quad anywhere
            const bool flag = valid && (((int)h0 + (int)h1 + (int)h2 + (int)h3) != 1 || more);
            const int gw = h0 ? 0 : h1 ? 1 : h2 ? 2 : 3;
            int bq = __float_as_int(gw == 0 ? a1.x : gw == 1 ? a1.y : gw == 2 ? a1.z : a1.w) & 15;     // the quad index packed by the scan
            bq = valid && bq >= 0 && bq < (K >> 6) ? bq : 0;       // (idle lanes of the ZT < 4 instantiations read stale LDS)
            // exact fp32 on the 16 codes of the winning (quad, row group); descending, the lowest index wins ties
            float d = __builtin_inff();
            int wi = 0;
#pragma unroll
            for (int q = 15; q >= 0; --q) {
                const int c = 64 * bq + 16 * (q >> 2) + 4 * gw + (q & 3);
                const float dd = dist_row(y0, y1, y2, y3, yy, cbs[c], ees[c]);
                const bool take = dd <= d;
                d = take ? dd : d;
                wi = take ? c : wi;
            }
            const unsigned long long fmask = __ballot(flag);               // one bit per vector, wave-uniform
            const int nflag = __builtin_popcountll(fmask);
            if (nflag <= kVqfBulk) {
                // a few near-ties: the whole wave scans all K codes exactly for each such vector
                unsigned long long todo = fmask;
                while (todo) {
                    const int v = __builtin_ctzll(todo);
                    todo &= todo - 1;
                    const float s0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(y0), v));
                    const float s1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(y1), v));
                    const float s2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(y2), v));
                    const float s3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(y3), v));
                    const float ss = sumsq4(s0, s1, s2, s3);
                    float bd = __builtin_inff();
                    int bi = 0;
#pragma unroll 4
                    for (int c = K - 64 + lane; c >= 0; c -= 64) {       // descending: the lowest index wins ties
                        const float dd = dist_row(s0, s1, s2, s3, ss, cbs[c], ees[c]);
                        const bool take = dd <= bd;
                        bd = take ? dd : bd;
                        bi = take ? c : bi;
                    }
#pragma unroll
                    for (int off = 1; off < 64; off <<= 1) {
                        const float od = __shfl_xor(bd, off, kWave);
                        const int oi = __shfl_xor(bi, off, kWave);
                        const bool take = od < bd || (od == bd && oi < bi);
                        bd = take ? od : bd;
                        bi = take ? oi : bi;
                    }
                    wi = lane == v ? bi : wi;
                }
                // back to the (column, row group) layout of the outputs
                int *tw = reinterpret_cast<int *>(tb + 1024);
                tw[lane] = wi;
#pragma unroll
                for (int t = 0; t < ZT; ++t) win[t] = tw[t * 16 + j];
            } else {
                // many near-ties (degenerate codebooks, zz >> ee, non-finite input): the exact fp32-MFMA scan of
                // vq_mfma_body for the whole group, operands from the fp32 rows in LDS
                float z0[ZT], z1[ZT], z2[ZT], z3[ZT], zz[ZT], best[ZT];
                int bt2[ZT];
#pragma unroll
                for (int t = 0; t < ZT; ++t) {
                    const uint2 eo = rows16(__float_as_uint(zv[t]));         // (z0,z0,z2,z2) / (z1,z1,z3,z3)
                    const uint2 e = rows32(eo.x), o = rows32(eo.y);
                    z0[t] = __uint_as_float(e.x); z2[t] = __uint_as_float(e.y);
                    z1[t] = __uint_as_float(o.x); z3[t] = __uint_as_float(o.y);
                    zz[t] = sumsq4(z0[t], z1[t], z2[t], z3[t]);
                    best[t] = __builtin_inff();
                    bt2[t] = 0;
                }
                for (int ct = 0; ct < ntile; ++ct) {
                    const float av = reinterpret_cast<const float *>(cbs)[(16 * ct + j) * 4 + g];
                    const f32x4 e4 = *reinterpret_cast<const f32x4 *>(&ees[16 * ct + 4 * g]);
#pragma unroll
                    for (int t = 0; t < ZT; ++t) {
                        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, zv[t], acc, 0, 0, 0);
                        const float d0 = __builtin_fmaf(-2.0f, acc[0], zz[t] + e4[0]);
                        const float d1 = __builtin_fmaf(-2.0f, acc[1], zz[t] + e4[1]);
                        const float d2 = __builtin_fmaf(-2.0f, acc[2], zz[t] + e4[2]);
                        const float d3 = __builtin_fmaf(-2.0f, acc[3], zz[t] + e4[3]);
                        const float q1 = __builtin_fminf(__builtin_fminf(best[t], d0), d1);
                        const float q2 = __builtin_fminf(__builtin_fminf(q1, d2), d3);
                        bt2[t] = q2 < best[t] ? ct : bt2[t];
                        best[t] = q2;
                    }
                }
#pragma unroll
                for (int t = 0; t < ZT; ++t) {
                    const int c0 = 16 * bt2[t] + 4 * g;
                    float dq = best[t];
                    int i = c0;
#pragma unroll
                    for (int r = 3; r >= 0; --r)
                        i = dist_row(z0[t], z1[t], z2[t], z3[t], zz[t], cbs[c0 + r], ees[c0 + r]) == dq ? c0 + r : i;
                    colargmin(dq, i);
                    win[t] = i;
                }
            }
        }
        CGIC_STAMP(4);

        // ---- outputs: this lane owns channel g of vector n
        double sq = 0.0;
#pragma unroll
        for (int t = 0; t < ZT; ++t) {
            const int64_t n = base + 16 * t + j;
            if (n < N) {
                if (zq_out || a.sq_partial) {
                    const float e = reinterpret_cast<const float *>(cbs)[win[t] * 4 + g];
                    const float diff = e - zv[t];
                    if (zq_out) {
                        int64_t b, p;
                        locate(gb, gp, t, &b, &p);
                        zq_out[(b * 4 + g) * hw + p] = zv[t] + diff;
                    }
                    sq += (double)diff * (double)diff;
                }
                if (g == 0 && idx_out) idx_out[n] = (int64_t)win[t];
            }
        }
        if (a.sq_partial) {
            // one partial per GROUP (not per wave), parked in LDS: whichever wave took the group, the workgroup's sum
            // has the same operands in the same order.  Fixed shuffle tree.
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) sq += __shfl_down(sq, off, kWave);
            if (lane == 0) gsum[grp - blk_lo] = sq;
        }
    }

    CGIC_STAMP(5);
    if (a.sq_partial) {
        // Only wave 0 stays for the hand-off: the other waves leave at the barrier WITHOUT draining their z_q /
        // index stores (an s_waitcnt vmcnt(0) in every wave before the barrier cost ~4 us at the end of every
        // workgroup); wave 0's own stores are long complete by the time it has waited for the others.
        __syncthreads();
        if (wave == 0) {
            const int ng = (int)(blk_hi - blk_lo);
            double bs = 0.0;
            for (int i = lane; i < ng; i += kWave) bs += gsum[i];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) bs += __shfl_down(bs, off, kWave);
            finish_loss_wave(bs, a.sq_partial, a.ticket, (double)N * 4.0, a.beta, a.legacy, a.loss, vblk, a.nblk);
        }
    }
    CGIC_STAMP(6);
    CGIC_BLK_END();
}

template <int ZT>
__global__ __launch_bounds__(kVqfThreads, 1) void vq_filter_kernel(VqArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_f[];
    vq_filter_body<ZT>(a, smem_f, blockIdx.x);
}

// The fused launch of the filter path (see vq_router_kernel).  A VQ workgroup takes the CU's LDS (84 KB) and
// 2 x 166 VGPRs per SIMD, and dynamic LDS / the VGPR budget are per launch, so a router workgroup cannot share a CU
// with one: behind the VQ workgroups it only starts when the VQ is over (30.4 + ~11 us).  The router workgroups
// therefore come FIRST (`nrouter` of them, one CU each for ~11 us); the VQ workgroups that have to wait for those
// CUs own fewer groups, the others more -- in steps of 4 groups, the unit in which a workgroup's time grows (8
// waves, 2 per SIMD).  Measured alternatives at B=64: an evenly balanced uneven split (18 / 11 groups) 43.2 us,
// 8-byte A operands + a 128-VGPR build so that both kinds share a CU 47.6 (two VQ workgroups then also share
// CUs), a device-wide chunk queue 46.3, the router on a forked graph branch +9 us per step.
template <int ZT>
__global__ __launch_bounds__(kVqfThreads, 1) void vq_filter_router_kernel(VqArgs a, RouterArgs r, unsigned int nrouter)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_f[];
    if (blockIdx.x < nrouter) {
        router_body<kVqfThreads>(r, (int64_t)blockIdx.x, smem_f);
        return;
    }
    vq_filter_body<ZT>(a, smem_f, blockIdx.x - nrouter);
}

// the same with the router workgroups BEHIND the VQ workgroups (when every VQ workgroup gets a CU at once anyway)
template <int ZT>
__global__ __launch_bounds__(kVqfThreads, 1) void vq_filter_router_behind_kernel(VqArgs a, RouterArgs r)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_f[];
    if (blockIdx.x >= a.nblk) {
        router_body<kVqfThreads>(r, (int64_t)(blockIdx.x - a.nblk), smem_f);
        return;
    }
    vq_filter_body<ZT>(a, smem_f, blockIdx.x);
}

// Plain-VALU restatement: one latent vector per thread, codebook broadcast from
// LDS.  Independent of the MFMA path; used to cross-check it on hardware.
__global__ __launch_bounds__(kVqThreads) void vq_valu_kernel(
    const float *__restrict__ z, int64_t hw, int64_t N, const float *__restrict__ cb, int K,
    int64_t *__restrict__ idx_out, float *__restrict__ zq_out, double *__restrict__ sq_partial,
    unsigned int *__restrict__ ticket, float beta, int legacy, float *__restrict__ loss)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float4 *cbs = reinterpret_cast<float4 *>(smem);  // [K]
    float *ee = smem + 4 * K;
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        float4 e = reinterpret_cast<const float4 *>(cb)[k];
        cbs[k] = e;
        ee[k] = sumsq4(e.x, e.y, e.z, e.w);
    }
    __syncthreads();
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    double sq = 0.0;
    if (n < N) {
        int64_t b = n / hw, p = n - b * hw;
        const float *zp = z + b * 4 * hw + p;
        float z0 = zp[0], z1 = zp[hw], z2 = zp[2 * hw], z3 = zp[3 * hw];
        float zz = sumsq4(z0, z1, z2, z3);
        float best = 0.f;
        int bi = 0;
        for (int k = 0; k < K; ++k) {
            float4 e = cbs[k];
            float mm = z0 * e.x;
            mm = __builtin_fmaf(z1, e.y, mm);
            mm = __builtin_fmaf(z2, e.z, mm);
            mm = __builtin_fmaf(z3, e.w, mm);
            float s = zz + ee[k];
            float d = __builtin_fmaf(-2.0f, mm, s);
            bool take = k == 0 || d < best;
            best = take ? d : best;
            bi = take ? k : bi;
        }
        if (idx_out) idx_out[n] = bi;
        if (zq_out || sq_partial) {
            float4 e = cbs[bi];
            float d0 = e.x - z0, d1 = e.y - z1, d2 = e.z - z2, d3 = e.w - z3;
            if (zq_out) {
                float *q = zq_out + b * 4 * hw + p;
                q[0] = z0 + d0; q[hw] = z1 + d1; q[2 * hw] = z2 + d2; q[3 * hw] = z3 + d3;
            }
            sq = (double)d0 * d0 + (double)d1 * d1 + (double)d2 * d2 + (double)d3 * d3;
        }
    }
    if (sq_partial) {
        __shared__ double wsum[4];
        const int lane = lane_id(), wave = threadIdx.x >> 6;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sq += __shfl_down(sq, off, kWave);
        if (lane == 0) wsum[wave] = sq;
        __syncthreads();
        finish_loss<kVqThreads>(((wsum[0] + wsum[1]) + wsum[2]) + wsum[3], sq_partial, ticket, (double)N * 4.0, beta, legacy, loss, blockIdx.x, gridDim.x);
    }
}

// Usage histogram of an index tensor (quantize.py:79-81).  A small number of fat blocks, each with a
// private LDS histogram, so a hot bin receives at most gridDim.x global atomics (not one per
// 256 vectors): same-address atomics serialise in L2.
__global__ __launch_bounds__(1024) void index_hist_kernel(const int64_t *__restrict__ idx, int64_t n, int K,
                                                          unsigned long long *__restrict__ hist)
{
    extern __shared__ unsigned int lh[];
    for (int k = threadIdx.x; k < K; k += blockDim.x) lh[k] = 0;
    __syncthreads();
    const int64_t per = (n + gridDim.x - 1) / gridDim.x;
    const int64_t lo = per * blockIdx.x, hi = lo + per < n ? lo + per : n;
    for (int64_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        int64_t v = idx[i];
        if (v >= 0 && v < K) atomicAdd(&lh[v], 1u);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < K; k += blockDim.x)
        if (lh[k]) atomicAdd(&hist[k], (unsigned long long)lh[k]);
}

static int launch_hist(const int64_t *idx, int64_t n, int K, int64_t *hist, hipStream_t s)
{
    int nblk = (int)((n + 4095) / 4096);
    if (nblk > 64) nblk = 64;
    if (nblk < 1) nblk = 1;
    hipLaunchKernelGGL(index_hist_kernel, dim3(nblk), dim3(1024), sizeof(unsigned int) * (size_t)K, s, idx, n, K,
                       (unsigned long long *)hist);
    return launch_check("index_hist_kernel");
}

static int vq_check(const float *z, int64_t B, int64_t hw, const float *cb, int K, int e_dim,
                    const float *loss, const void *ws, bool hist_needs_idx)
{
    CGIC_REQUIRE(z && cb, CGIC_ERR_INVALID, "vq: z and codebook must not be NULL");
    CGIC_REQUIRE(B >= 0 && hw >= 0, CGIC_ERR_INVALID, "vq: negative shape");
    CGIC_REQUIRE(e_dim == 4, CGIC_ERR_UNSUPPORTED,
                 "vq: e_dim=%d; this build implements embed_dim == 4 (config_inference.yaml:8)", e_dim);
    CGIC_REQUIRE(K > 0 && K % 16 == 0 && K <= kVqMaxK, CGIC_ERR_UNSUPPORTED,
                 "vq: K=%d; need K %% 16 == 0 and K <= %d", K, kVqMaxK);
    CGIC_REQUIRE(!loss || ws, CGIC_ERR_INVALID, "vq: loss requested without workspace");
    CGIC_REQUIRE(!hist_needs_idx, CGIC_ERR_INVALID, "vq: hist requires indices (the histogram is taken from them)");
    return CGIC_OK;
}

struct VqWs {
    unsigned int *ticket;   // library-owned, self-resetting
    double *partial;        // caller's workspace: double partial[nblk]
};

static int vq_ws(void *workspace, hipStream_t s, VqWs *out)
{
    out->partial = (double *)workspace;
    out->ticket = nullptr;
    if (!workspace) return CGIC_OK;
    return acquire_tickets(s, 1, &out->ticket);
}

template <int ZT>
static int launch_mfma(const float *z, int64_t hw, int64_t N, const float *cb, int K, int64_t *idx,
                       float *zq, VqWs ws, float beta, int legacy, float *loss, hipStream_t s,
                       const RouterArgs *router, int64_t router_blocks, size_t router_lds)
{
    const int64_t per_block = 4 * 16 * ZT;
    VqArgs a;
    a.z = z; a.hw = hw; a.N = N; a.cb = cb; a.K = K; a.idx_out = idx; a.zq_out = zq;
    a.sq_partial = loss ? ws.partial : nullptr; a.ticket = ws.ticket; a.beta = beta; a.legacy = legacy; a.loss = loss;
    a.nblk = (unsigned int)((N + per_block - 1) / per_block);
    size_t lds = sizeof(float) * (size_t)K * 5;
    if (!router) {
        hipLaunchKernelGGL(vq_mfma_kernel<ZT>, dim3(a.nblk), dim3(kVqThreads), lds, s, a);
        return launch_check("vq_mfma_kernel");
    }
    if (router_lds > lds) lds = router_lds;
    if (lds > 64 * 1024)
        CGIC_HIP_TRY(hipFuncSetAttribute((const void *)vq_router_kernel<ZT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(vq_router_kernel<ZT>, dim3(a.nblk + (unsigned int)router_blocks), dim3(kVqThreads), lds, s, a, *router);
    return launch_check("vq_router_kernel");
}

static int device_cu_count(int *out)
{
    static std::mutex mu;
    static std::map<int, int> cus;
    int dev = 0;
    CGIC_HIP_TRY(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    auto it = cus.find(dev);
    if (it == cus.end()) {
        int n = 0;
        CGIC_HIP_TRY(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev));
        it = cus.emplace(dev, n > 0 ? n : 256).first;
    }
    *out = it->second;
    return CGIC_OK;
}

template <int ZT>
static int launch_filter(const float *z, int64_t hw, int64_t N, const float *cb, int K, int64_t *idx, float *zq,
                         VqWs ws, float beta, int legacy, float *loss, hipStream_t s, const RouterArgs *router,
                         int64_t router_blocks, size_t router_lds)
{
    int cus = 0;
    int rc = device_cu_count(&cus);
    if (rc) return rc;
    const int64_t ngroups = (N + 16 * ZT - 1) / (16 * ZT);
    int64_t nblk = (ngroups + kVqfWaves - 1) / kVqfWaves;
    if (nblk > cus) nblk = cus;                      // one resident workgroup per CU; waves take groups from a counter
    const int64_t kMaxGroups = 6144;                 // per workgroup: 8 bytes of LDS each for the loss partials
    if ((ngroups + nblk - 1) / nblk > kMaxGroups) nblk = (ngroups + kMaxGroups - 1) / kMaxGroups;
    VqArgs a;
    a.z = z; a.hw = hw; a.N = N; a.cb = cb; a.K = K; a.idx_out = idx; a.zq_out = zq;
    a.sq_partial = loss ? ws.partial : nullptr; a.ticket = ws.ticket; a.beta = beta; a.legacy = legacy; a.loss = loss;
    a.nblk = (unsigned int)nblk;
    // groups per workgroup.  Router workgroups in front: the `late` VQ workgroups that must wait for a router's CU
    // (~11 us at 256x256, ~0.0021 * hw groups of VQ work) own `g_late` groups, the others `g_early`, a multiple of 4
    int64_t per = (ngroups + nblk - 1) / nblk, g_early = per, g_late = per, n_early = nblk;
    static const int split_off = getenv("CGIC_VQ_NOSPLIT") ? atoi(getenv("CGIC_VQ_NOSPLIT")) : 0;    // dev: A/B
    const int64_t late = router ? nblk + router_blocks - cus : 0;
    bool router_first = false;
    if (late > 0 && late < nblk && !split_off) {
        const int64_t delta = (int64_t)(0.0021 * (double)hw * (4.0 / ZT) + 0.5);
        static const int force_ge = getenv("CGIC_VQ_GE") ? atoi(getenv("CGIC_VQ_GE")) : 0;     // dev: tuning
        for (int64_t ge = force_ge ? force_ge : (per / 4 + 1) * 4; ge <= per + 12; ge += 4) {
            const int64_t rest = ngroups - (nblk - late) * ge;
            const int64_t gl = rest > 0 ? (rest + late - 1) / late : 0;
            // (gl == 0: the router outlasts the whole VQ -- the early workgroups simply take everything)
            if (gl + delta <= ge || gl == 0) { g_early = ge; g_late = gl; n_early = nblk - late; router_first = true; break; }
        }
    }
    a.n_early = (unsigned int)n_early; a.g_early = (unsigned int)g_early; a.g_late = (unsigned int)g_late;
    const int64_t gmax = g_early > g_late ? g_early : g_late;
    size_t lds = (size_t)K * 84 + (size_t)kVqfWaves * kVqfTbFloats * 4 + (loss ? 8 * (size_t)gmax : 0);
    if (!router) {
        CGIC_HIP_TRY(hipFuncSetAttribute((const void *)vq_filter_kernel<ZT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(vq_filter_kernel<ZT>, dim3(a.nblk), dim3(kVqfThreads), lds, s, a);
        return launch_check("vq_filter_kernel");
    }
    if (router_lds > lds) lds = router_lds;
    CGIC_HIP_TRY(hipFuncSetAttribute((const void *)vq_filter_router_kernel<ZT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (router_first) {
        hipLaunchKernelGGL(vq_filter_router_kernel<ZT>, dim3(a.nblk + (unsigned int)router_blocks), dim3(kVqfThreads), lds, s, a, *router,
                           (unsigned int)router_blocks);
    } else {
        CGIC_HIP_TRY(hipFuncSetAttribute((const void *)vq_filter_router_behind_kernel<ZT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(vq_filter_router_behind_kernel<ZT>, dim3(a.nblk + (unsigned int)router_blocks), dim3(kVqfThreads), lds, s, a, *router);
    }
    return launch_check("vq_filter_router_kernel");
}

static int vq_dispatch(const float *z, int64_t hw, int64_t N, const float *codebook, int K, int64_t *indices, float *z_q,
                       VqWs ws, float beta, int legacy, float *loss, hipStream_t s, const RouterArgs *router,
                       int64_t router_blocks, size_t router_lds)
{
    // per-wave tile: measured on MI355X (tools/probe_vq.hip) ZT=4 at 4 waves/SIMD is the fastest
    // for large N; smaller N shrinks the tile so that all 256 CUs get work
    static const int force_zt = getenv("CGIC_VQ_ZT") ? atoi(getenv("CGIC_VQ_ZT")) : 0;     // tuning knob (dev)
    static const int exact_only = getenv("CGIC_VQ_EXACT") ? atoi(getenv("CGIC_VQ_EXACT")) : 0;   // dev: A/B against the exact loop
    if (!exact_only && K % 64 == 0 && K <= kVqfMaxK) {
#define CGIC_VQF_LAUNCH(ZT) launch_filter<ZT>(z, hw, N, codebook, K, indices, z_q, ws, beta, legacy, loss, s, router, router_blocks, router_lds)
        if (force_zt == 2) return CGIC_VQF_LAUNCH(2);
        if (force_zt == 1) return CGIC_VQF_LAUNCH(1);
        if (force_zt == 4 || N >= (int64_t)128 * 1024) return CGIC_VQF_LAUNCH(4);
        if (N >= (int64_t)64 * 1024) return CGIC_VQF_LAUNCH(2);
        return CGIC_VQF_LAUNCH(1);
#undef CGIC_VQF_LAUNCH
    }
#define CGIC_VQ_LAUNCH(ZT) launch_mfma<ZT>(z, hw, N, codebook, K, indices, z_q, ws, beta, legacy, loss, s, router, router_blocks, router_lds)
    if (force_zt == 8) return CGIC_VQ_LAUNCH(8);
    if (force_zt == 4) return CGIC_VQ_LAUNCH(4);
    if (force_zt == 2) return CGIC_VQ_LAUNCH(2);
    if (force_zt == 1) return CGIC_VQ_LAUNCH(1);
    if (N >= (int64_t)1 << 22) return CGIC_VQ_LAUNCH(8);
    if (N >= (int64_t)256 * 512) return CGIC_VQ_LAUNCH(4);
    if (N >= (int64_t)128 * 512) return CGIC_VQ_LAUNCH(2);
    return CGIC_VQ_LAUNCH(1);
#undef CGIC_VQ_LAUNCH
}

}  // namespace cgic

using namespace cgic;

extern "C" size_t cgic_vq_workspace_bytes(int64_t n_vectors)
{
    // one double per workgroup; (n / 16 + 1) covers every tiling of both paths
    return sizeof(double) * (size_t)((n_vectors + 15) / 16 + 1);
}

extern "C" int cgic_vq_forward_f32(const float *z, int64_t B, int64_t hw, const float *codebook, int K,
                                   int e_dim, float beta, int legacy, int64_t *indices, float *z_q,
                                   float *loss, int64_t *hist, void *workspace, cgic_stream_t stream)
{
    int rc = vq_check(z, B, hw, codebook, K, e_dim, loss, workspace, hist && !indices);
    if (rc) return rc;
    const int64_t N = B * hw;
    if (N == 0) return CGIC_OK;
    hipStream_t s = (hipStream_t)stream;
    VqWs ws;
    rc = vq_ws(loss ? workspace : nullptr, s, &ws);
    if (rc) return rc;
    rc = vq_dispatch(z, hw, N, codebook, K, indices, z_q, ws, beta, legacy, loss, s, nullptr, 0, 0);
    if (rc == CGIC_OK && hist) rc = launch_hist(indices, N, K, hist, s);
    return rc;
}

extern "C" int cgic_vq_forward_route_f32(const float *z, int64_t B, int64_t hw, const float *codebook, int K, int e_dim,
                                         float beta, int legacy, int64_t *indices, float *z_q, float *loss,
                                         void *workspace, const float *e16, const float *e8, int64_t h16, int64_t w16,
                                         double coarse_ratio, double medium_ratio, int per_image, int32_t *mask_c,
                                         int32_t *mask_m, int32_t *mask_f, float *gate, int *mode_out,
                                         cgic_stream_t stream)
{
    int rc = vq_check(z, B, hw, codebook, K, e_dim, loss, workspace, false);
    if (rc) return rc;
    if (mode_out) *mode_out = cgic_router_mode(coarse_ratio, medium_ratio);
    const int64_t N = B * hw;
    if (N == 0) return CGIC_OK;
    RouterArgs r;
    int64_t nseg;
    size_t rlds;
    rc = router_prepare(e16, e8, B, h16, w16, coarse_ratio, medium_ratio, per_image, mask_c, mask_m, mask_f, gate, &r, &nseg, &rlds);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    VqWs ws;
    rc = vq_ws(loss ? workspace : nullptr, s, &ws);
    if (rc) return rc;
    return vq_dispatch(z, hw, N, codebook, K, indices, z_q, ws, beta, legacy, loss, s, &r, nseg, rlds);
}

extern "C" int cgic_vq_forward_valu_f32(const float *z, int64_t B, int64_t hw, const float *codebook, int K,
                                        int e_dim, float beta, int legacy, int64_t *indices, float *z_q,
                                        float *loss, int64_t *hist, void *workspace, cgic_stream_t stream)
{
    int rc = vq_check(z, B, hw, codebook, K, e_dim, loss, workspace, hist && !indices);
    if (rc) return rc;
    const int64_t N = B * hw;
    if (N == 0) return CGIC_OK;
    hipStream_t s = (hipStream_t)stream;
    VqWs ws;
    rc = vq_ws(loss ? workspace : nullptr, s, &ws);
    if (rc) return rc;
    int nblk = (int)((N + kVqThreads - 1) / kVqThreads);
    size_t lds = sizeof(float) * (size_t)K * 5;
    hipLaunchKernelGGL(vq_valu_kernel, dim3(nblk), dim3(kVqThreads), lds, s, z, hw, N, codebook, K, indices, z_q,
                       loss ? ws.partial : nullptr, ws.ticket, beta, legacy, loss);
    rc = launch_check("vq_valu_kernel");
    if (rc == CGIC_OK && hist) rc = launch_hist(indices, N, K, hist, s);
    return rc;
}

extern "C" int cgic_index_histogram(const int64_t *indices, int64_t n, int K, int64_t *hist, cgic_stream_t stream)
{
    CGIC_REQUIRE(indices && hist && K > 0 && K <= 16384 && n >= 0, CGIC_ERR_INVALID, "index_histogram: bad args");
    if (n == 0) return CGIC_OK;
    return launch_hist(indices, n, K, hist, (hipStream_t)stream);
}
