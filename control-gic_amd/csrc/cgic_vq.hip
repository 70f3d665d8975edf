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

#include <atomic>

#include <stdlib.h>

#include <algorithm>
#include <map>
#include <mutex>
#include <vector>
#include <utility>

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
    // no acquire fence: every partial was stored write-through (sc1) and is read below with sc1 loads, which are served by
    // L2 / memory, never by this CU's L1 (8-byte agent-scope atomics on both sides; an L1 invalidate costs ~1.7 us here)
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

// Cross-row-group exchanges on the VALU (gfx950 v_permlane16_swap / v_permlane32_swap) -- no LDS round trip.
//   rows16(x): .x = x of the 16-lane rows (0,0,2,2), .y = x of rows (1,1,3,3)
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
// sum over the 64 lanes in a fixed association (xor 1, 2, 4, 8 on DPP operands, then the 16- and 32-lane swaps):
// every lane gets the total; no LDS crossbar (a shuffle of a double is two ds_bpermute)
template <int CTRL>
__device__ __forceinline__ double dpp_add_f64(double v)
{
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned int lo = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)(unsigned int)u, CTRL, 0xF, 0xF, true);
    const unsigned int hi = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)(unsigned int)(u >> 32), CTRL, 0xF, 0xF, true);
    return v + __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double wave_sum_f64(double v)
{
    v = dpp_add_f64<0xB1>(v);        // quad_perm [1,0,3,2]: xor 1
    v = dpp_add_f64<0x4E>(v);        // quad_perm [2,3,0,1]: xor 2
    v = dpp_add_f64<0x141>(v);       // row_half_mirror: xor 4 for quad-uniform values
    v = dpp_add_f64<0x140>(v);       // row_mirror: xor 8 for values uniform over 8 lanes
    unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    uint2 a = rows16((unsigned int)u), b = rows16((unsigned int)(u >> 32));
    v = __builtin_bit_cast(double, ((unsigned long long)b.x << 32) | a.x) + __builtin_bit_cast(double, ((unsigned long long)b.y << 32) | a.y);
    u = __builtin_bit_cast(unsigned long long, v);
    a = rows32((unsigned int)u);
    b = rows32((unsigned int)(u >> 32));
    return __builtin_bit_cast(double, ((unsigned long long)b.x << 32) | a.x) + __builtin_bit_cast(double, ((unsigned long long)b.y << 32) | a.y);
}

#ifdef CGIC_VQF_DRAIN_HANDOFF      // dev A/B: round 5's hand-off (partials in the caller's workspace, drained before the ticket)
// Variant for the filter path, executed by ONE wave: lane 0 publishes the workgroup's partial (write-through
// store, drained, then the ticket); the wave of the last workgroup sums all partials in a fixed order.
__device__ __forceinline__ void finish_loss_wave(double block_sum, double *sq_partial, unsigned int *ticket, double count,
                                                 float beta, int legacy, float *loss, unsigned int blk, unsigned int nblk, unsigned int = 0)
{
    const int lane = lane_id();
    int last = 0;
    if (lane == 0) {
#ifdef CGIC_STRICT_HANDOFF
        // textbook form (make FLAGS+=-DCGIC_STRICT_HANDOFF): plain store, release at the ticket; for A/B runs against the
        // write-through shortcut below on a new ROCm / GPU
        sq_partial[blk] = block_sum;
        last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT) == nblk - 1;
#else
        __hip_atomic_store(&sq_partial[blk], block_sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nblk - 1;
#endif
    }
    last = __builtin_amdgcn_readfirstlane(last);
    if (!last) return;
#ifdef CGIC_STRICT_HANDOFF
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
    // (no acquire fence: sc1 stores on the producers' side, sc1 loads here -- see finish_loss)
    double a = 0.0;
    for (unsigned int i = lane; i < nblk; i += kWave)
        a += __hip_atomic_load(&sq_partial[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    a = wave_sum_f64(a);
    if (lane == 0) {
        const float m = (float)(a / count);
        *loss = legacy ? (m + beta * m) : (beta * m + m);
        __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
    }
}

#else
// Variant for the filter path, executed by ONE wave.  Round 6: the partials live in LIBRARY-owned, zero-on-entry slots (behind the
// ticket: acquire_tickets) instead of the caller's workspace, and a workgroup publishes -partial -- the sum of squares is never
// negative, so the sign bit says "written" (a +0.0 partial goes out as -0.0, a NaN stays a NaN with its sign set) -- with a
// write-through store that it does NOT wait for before it takes its ticket: whoever sums polls any slot that still reads zero
// (its store was issued before its workgroup left: it is on its way), sums in the fixed order as before, and hands the slots back
// zeroed.  The sum is the same additions in the same order whoever makes it: the same bits.
// loss_collect: the caller is the last to arrive (one wave): wait for every partial, sum, write the loss, hand everything back zeroed
__device__ __forceinline__ void loss_collect(unsigned long long *slots, unsigned int *ticket, double count, float beta, int legacy,
                                             float *loss, unsigned int nblk)
{
    const int lane = lane_id();
    double a = 0.0;
    for (unsigned int i = lane; i < nblk; i += kWave) {
        unsigned long long v = __hip_atomic_load(&slots[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (v == 0ull) {                                      // (still in flight, or its workgroup is still working: see above)
            __builtin_amdgcn_s_sleep(1);
            v = __hip_atomic_load(&slots[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        a += -__builtin_bit_cast(double, v);
    }
    for (unsigned int i = lane; i < nblk; i += kWave) __hip_atomic_store(&slots[i], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    a = wave_sum_f64(a);
    if (lane == 0) {
        const float m = (float)(a / count);
        *loss = legacy ? (m + beta * m) : (beta * m + m);
        __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
    }
}
__device__ __forceinline__ void finish_loss_wave(double block_sum, double *sq_partial, unsigned int *ticket, double count,
                                                 float beta, int legacy, float *loss, unsigned int blk, unsigned int nblk,
                                                 unsigned int tail_mode = 0)
{
    const int lane = lane_id();
    unsigned long long *slots = reinterpret_cast<unsigned long long *>(sq_partial);
    int last = 0;
    if (lane == 0) {
        __hip_atomic_store(&slots[blk], __builtin_bit_cast(unsigned long long, -block_sum), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tail_mode == 0) last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nblk - 1;
        if (tail_mode == 2) last = blk == 0;          // the COLLECTOR: no tickets at all -- workgroup 0 waits for everybody's partial
    }
    last = __builtin_amdgcn_readfirstlane(last);
    if (!last) return;
    loss_collect(slots, ticket, count, beta, legacy, loss, nblk);
}
#endif
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
    // loss hand-off of the filter path: 0 = the VQ workgroups take a ticket, the last one sums (finish_loss_wave); 2 (round 6) = a VQ
    // workgroup only PUBLISHES its partial and workgroup 0 waits for all of them and sums -- 256 returning atomics on one word at the
    // end of every VQ workgroup were 2.3 us of the launch; 1 = dev A/B: the router workgroups of the fused launch take the ticket among
    // themselves and the last ROUTER sums (loss_collect_by_routers: slower, the routers are the launch's tail)
    unsigned int tail_mode;
    // quant_conv fused in front of the quantiser (model.py:51,110): z = W h (+ b), 4 -> 4, or NULL
    const float *conv_w, *conv_b;
    int conv_bias_first;
    // filter path: the LDS image of the codebook (split fp16 A operands, padded fp32 rows, row norms, maxima) as
    // vq_prepare_kernel left it, or NULL: every workgroup then derives it from `cb` itself
    const void *prep;
    // telemetry (cgic_vq_stats; NULL = off): [0] vectors that took the all-K exact scan ("flagged"), [1] 64-vector groups that
    // reran the exact fp32-MFMA loop, [2] groups that evaluated a second candidate set, [3] groups seen -- touched only
    // inside the rare branches, so a launch without near-ties executes nothing for it
    unsigned int *stats;
    // margin telemetry (cgic_vq_filter_probe_f32 only; the product instantiations never read them): every approximate score
    // the filter saw, [N, K], and per vector (f_min, margin M, threshold, flagged, scale exponent q, S) in unscaled score units
    float *probe_scores, *probe_aux;
};

#ifndef CGIC_VQF_DRAIN_HANDOFF
// tail_mode 1, called by every thread of a ROUTER workgroup of the fused launch when its routing is done: the routers take the ticket
// among themselves (64 atomics spread over ~1 us instead of 256 within the VQ workgroups' last microsecond) and the last one sums
// the VQ workgroups' partials -- which were published long before in the usual launch (the routers are its tail), and are simply
// waited for otherwise (the VQ workgroups wait for nobody)
__device__ __forceinline__ void loss_collect_by_routers(const VqArgs &a, unsigned int nrouter)
{
    if (a.tail_mode != 1 || !a.sq_partial) return;
    if (threadIdx.x >= kWave) return;
    int last = 0;
    if (lane_id() == 0) last = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nrouter - 1;
    last = __builtin_amdgcn_readfirstlane(last);
    if (!last) return;
    loss_collect(reinterpret_cast<unsigned long long *>(a.sq_partial), a.ticket, (double)a.N * 4.0, a.beta, a.legacy, a.loss, a.nblk);
}
#else
__device__ __forceinline__ void loss_collect_by_routers(const VqArgs &, unsigned int) {}
#endif

// The reference's quant_conv is a torch.nn.Conv2d(4, 4, 1) on the CPU.  Its fp32 rounding sequence is an fma chain over
// the input channels in order, with the bias either seeding the accumulator or added at the end -- oneDNN picks one or
// the other by shape and thread count (measured: one thread, or a 64x64 latent, adds the bias last; 8 threads and a
// 192x192 latent seed with it).  Both orders are implemented; the caller says which one to reproduce.
struct Conv1x1 {
    float w[16], b[4];
    int bias_first, has_bias;
    __device__ __forceinline__ void load(const float *cw, const float *cb, int bf)
    {
#pragma unroll
        for (int i = 0; i < 16; ++i) w[i] = cw[i];            // wave-uniform addresses: scalar loads
#pragma unroll
        for (int i = 0; i < 4; ++i) b[i] = cb ? cb[i] : 0.f;
        bias_first = bf;
        has_bias = cb != nullptr;
    }
    __device__ __forceinline__ void apply(float (&v)[4]) const
    {
        const float h0 = v[0], h1 = v[1], h2 = v[2], h3 = v[3];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float first = w[4 * c] * h0, seeded = __builtin_fmaf(w[4 * c], h0, b[c]);
            float acc = (bias_first && has_bias) ? seeded : first;
            acc = __builtin_fmaf(w[4 * c + 1], h1, acc);
            acc = __builtin_fmaf(w[4 * c + 2], h2, acc);
            acc = __builtin_fmaf(w[4 * c + 3], h3, acc);
            v[c] = (!bias_first && has_bias) ? acc + b[c] : acc;
        }
    }
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
// Filter path (K % 64 == 0, K <= 1024; the reference's codebook is 1024 x 4): the fp16 matrix cores find
// the candidates, fp32 decides.  The exact loop above delivers every distance at the fp32 MFMA rate (1/16 of
// the 16-bit rate) plus ~3 VALU instructions per output pair; argmin only needs the distances that can win:
//
//  * score  f(k) = ee_k - 2 z.e_k  (the distance minus zz, which does not change the argmin) comes out of ONE
//    v_mfma_f32_32x32x16_f16 per (32 codes x 32 vectors): z_j and w_kj = -2 e_kj are each split into two fp16
//    pieces by round-to-nearest (11 + 11 significand bits) after an exact power-of-two scaling into fp16's range
//    (2^b for the codebook, 2^a per latent vector), ee_k into three pieces.  The 16 K-slots carry, per dimension,
//    zh.wh, zh.wm, zm.wh (12 slots), and ee's three pieces against 2^s, s = a + b - be (3 slots).  Every product
//    is exact in fp32; the dropped zm.wm terms and the fp32 accumulation leave |f - F| <= 1.8e-6 T_k,
//    T_k = ee_k + 2 sum|z_j e_kj|, plus an ABSOLUTE floor for pieces that fall into fp16's subnormal range
//    (<= 2^-7 in scaled units, where the scaled scores of a group are ~2^26; it only matters for a code whose own
//    T_k is tiny next to the codebook's scale, and then sends the vector to the exact path).
//    (Round 1 used three bf16 pieces by truncation and v_mfma_f32_16x16x32_bf16: 32 K-slots for the same
//    precision, i.e. twice the matrix-core time per (code, vector) pair.)
//  * per lane a running (smallest, second smallest) over TILES of 32 codes (the lane holds 16 of them), the tile's
//    index packed into the low 5 mantissa bits of its minimum: 8 v_min3 + and_or + med3 + min per MFMA.
//  * every code whose reference distance could be minimal has f <= f_min + M,
//    M = 2 (|f - F| + |d_ref - zz - F|) + packing <= 1.3e-5 S + 2.5e-7 zz + floor  (d_ref's own rounding:
//    2^-23 zz + 2^-21 S; packing 2^-18 |f| twice).  S is the smaller of the codebook-maxima bound and
//    (|z| + sqrt D)(3|z| + sqrt D), D = zz + f_min + slack (the winner and whatever can beat it lie within sqrt D
//    of z).  A vector's two 16-code sets (rows 4h..4h+3 mod 8 of each tile, h = lane >> 5) each contribute their
//    best and second-best tile.  If no second-best is under the threshold, the candidates are the best tile of one
//    or both halves: those 16 or 32 codes get the exact fp32 sequence, lane-local, and the lowest-index minimum is
//    the reference's argmin.  Otherwise (a third tile may hide behind the second: ~0.1 % of N(0,1) vectors) the
//    wave scans all K codes exactly for that vector; a group with many such vectors reruns the exact fp32-MFMA
//    loop.  Results are bit-identical to the exact kernels for every finite input.
// =====================================================================================================
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kVqfMaxK = 1024;
constexpr int kVqfBulk = 12;                   // more flagged vectors than this in a 64-vector group: rerun it exactly on the MFMA
constexpr int kVqfGroup = 64;                  // vectors per group: two tiles of 32 (one lane per vector in the decide step)

__device__ __forceinline__ unsigned int h16(float x)
{
    return (unsigned int)__builtin_bit_cast(unsigned short, (_Float16)x);
}
// x ~ h + m (+ l), fp16 pieces by round-to-nearest; the residuals x - h, x - h - m are exact in fp32
__device__ __forceinline__ void split2h(float x, unsigned int &h, unsigned int &m)
{
    const _Float16 hh = (_Float16)x;
    const float r = x - (float)hh;
    h = (unsigned int)__builtin_bit_cast(unsigned short, hh);
    m = h16(r);
}
__device__ __forceinline__ void split3h(float x, unsigned int &h, unsigned int &m, unsigned int &l)
{
    const _Float16 hh = (_Float16)x;
    const float r = x - (float)hh;
    const _Float16 mm = (_Float16)r;
    h = (unsigned int)__builtin_bit_cast(unsigned short, hh);
    m = (unsigned int)__builtin_bit_cast(unsigned short, mm);
    l = h16(r - (float)mm);
}
// floor(log2 |x|) of a finite x (-126 for zero / subnormals); 128 for Inf / NaN
__device__ __forceinline__ int exponent_of(float x)
{
    const int e = (int)((__float_as_uint(x) >> 23) & 255u);
    return e == 0 ? -126 : e - 127;
}

// lexicographic (d, i) minimum over the 4 row groups (lane >> 4) of each column (lane & 15)
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
// lexicographic (d, i) minimum over all 64 lanes on the VALU: xor 1, 2 (quad_perm), 4 (row_half_mirror), 8 (row_mirror) on DPP
// operands, then the 16- and 32-lane swaps of colargmin -- the minimum is idempotent, so mirrors do what xor shuffles do.  (The
// all-K scan of a flagged vector used twelve ds_bpermute round trips here: ~0.3 of its ~0.5 us, and the workgroups that meet three
// or four such vectors are the ones a launch waits for.)
template <int CTRL>
__device__ __forceinline__ void dpp_argmin_step(float &d, int &i)
{
    const float od = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(d), CTRL, 0xF, 0xF, true));
    const int oi = __builtin_amdgcn_update_dpp(0, i, CTRL, 0xF, 0xF, true);
    const bool take = od < d || (od == d && (unsigned int)oi < (unsigned int)i);
    d = take ? od : d;
    i = take ? oi : i;
}
__device__ __forceinline__ void wave_argmin(float &d, int &i)
{
    dpp_argmin_step<0xB1>(d, i);
    dpp_argmin_step<0x4E>(d, i);
    dpp_argmin_step<0x141>(d, i);
    dpp_argmin_step<0x140>(d, i);
    colargmin(d, i);
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

// Where code c's fp32 row / squared norm sits in LDS.  The decide step reads, in every lane, row 32 T + 4 h + 8 q + r of
// a lane-specific tile T: unpadded, all lanes of a row half would hit the same four banks (16-way conflicts; 67 % of the
// kernel's LDS cycles).  One extra row per tile (and four extra norms) rotate the bank with the tile.
__device__ __forceinline__ int rowpos(int c) { return c + (c >> 5); }
__device__ __forceinline__ int eepos(int c) { return c + 4 * (c >> 5); }
// LDS of the filter path: [K/32][64] x 16 B split A operands, [K + K/32] fp32 rows, [K + K/8] row norms
__host__ __device__ constexpr size_t vqf_rows_off(int K) { return (size_t)K * 32; }
__host__ __device__ constexpr size_t vqf_ees_off(int K) { return vqf_rows_off(K) + (size_t)(K + K / 32) * 16; }
__host__ __device__ constexpr size_t vqf_lds_bytes(int K) { return vqf_ees_off(K) + (size_t)(K + K / 8) * 4; }

// The codebook's LDS image of the filter path: fp32 rows (at rowpos()), row norms (at eepos()), the split fp16 A operands
// and, in s_max[0..1], the bit patterns of max |e_kj| and max ee_k (they fix the fp16 scaling).  Ends with a barrier.
constexpr unsigned int kPreparedMagic = 0x43474951u;      // "CGIQ": last word of a prepared codebook image, after (Emax, EEmax, K)
constexpr unsigned int kPreparedMagicPerm = 0x43474950u;  // "CGIP": a PERMUTED image (near-duplicate rows packed into tiles); K keys (orig << 16 | position) follow
// perm (vq_prepare_kernel only): position k of the image holds ORIGINAL row perm[k] (near-duplicate rows packed into one tile:
// cgic_vq_prepare_f32); nullptr: the identity.
template <int NT>
__device__ __forceinline__ void vqf_stage(const float *__restrict__ cb, const int K, unsigned char *smem, unsigned int *s_max,
                                          const unsigned short *__restrict__ perm = nullptr)
{
    uint4 *ldsA = reinterpret_cast<uint4 *>(smem);                                  // [K/32][64]
    float4 *cbs = reinterpret_cast<float4 *>(smem + vqf_rows_off(K));               // fp32 rows, at rowpos()
    float *ees = reinterpret_cast<float *>(smem + vqf_ees_off(K));                  // their squared norms, at eepos()
    const int tid = threadIdx.x, lane = tid & 63;
    // Phase 1: fp32 rows, row norms and the codebook maxima.
    if (tid < 2) s_max[tid] = 0;
    constexpr int kRows = (kVqfMaxK + NT - 1) / NT;          // rows per thread (2 at 512 threads)
    float4 rows[kRows];
    float rowee[kRows];
    {
        float emax = 0.f, eemax = 0.f;
#pragma unroll
        for (int q = 0; q < kRows; ++q) {
            const int k = tid + q * NT;
            rows[q] = k < K ? reinterpret_cast<const float4 *>(cb)[perm ? (int)perm[k] : k] : float4{0.f, 0.f, 0.f, 0.f};
        }
        __syncthreads();                    // s_max is zero
#pragma unroll
        for (int q = 0; q < kRows; ++q) {
            const int k = tid + q * NT;
            const float4 e = rows[q];
            rowee[q] = sumsq4(e.x, e.y, e.z, e.w);
            if (k < K) { cbs[rowpos(k)] = e; ees[eepos(k)] = rowee[q]; }
            emax = fmaxf(emax, fmaxf(fmaxf(fabsf(e.x), fabsf(e.y)), fmaxf(fabsf(e.z), fabsf(e.w))));
            eemax = fmaxf(eemax, rowee[q]);
            // fmaxf drops NaNs: route them into the maxima by hand (a NaN anywhere in the codebook disables the filter)
            if (!(rowee[q] == rowee[q])) eemax = rowee[q];
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float oe = __shfl_xor(emax, off, kWave), oee = __shfl_xor(eemax, off, kWave);
            emax = fmaxf(emax, oe);
            eemax = (oee != oee || eemax != eemax) ? __builtin_nanf("") : fmaxf(eemax, oee);
        }
        // non-negative floats order like their bit patterns; a NaN lands above every finite value
        if (lane == 0) { atomicMax(&s_max[0], __float_as_uint(emax)); atomicMax(&s_max[1], __float_as_uint(eemax)); }
    }
    __syncthreads();
    const float Emax = __uint_as_float(s_max[0]), EEmax = __uint_as_float(s_max[1]);
    // 2 Emax 2^b and EEmax 2^be in [2^13, 2^14)
    const int sb = 13 - (exponent_of(Emax) + 1), sbe = 13 - exponent_of(EEmax);
    // Phase 2: the split A operands.  Row k = 32 T + i feeds lanes i (K-slots 0..7) and 32 + i (K-slots 8..15) of tile T:
    //   slots 0..7  : wh0 wm0 wh0 | wh1 wm1 wh1 | wh2 wm2        against  zh0 zh0 zm0 | zh1 zh1 zm1 | zh2 zh2
    //   slots 8..15 : wh2 | wh3 wm3 wh3 | eh em el | 0          against  zm2 | zh3 zh3 zm3 | 2^s 2^s 2^s | 0
#pragma unroll
    for (int q = 0; q < kRows; ++q) {
        const int k = tid + q * NT;
        if (k < K) {
            const float4 e = rows[q];
            unsigned int wh0, wm0, wh1, wm1, wh2, wm2, wh3, wm3, eh, em, el;
            split2h(ldexpf(-2.0f * e.x, sb), wh0, wm0);
            split2h(ldexpf(-2.0f * e.y, sb), wh1, wm1);
            split2h(ldexpf(-2.0f * e.z, sb), wh2, wm2);
            split2h(ldexpf(-2.0f * e.w, sb), wh3, wm3);
            split3h(ldexpf(rowee[q], sbe), eh, em, el);
            uint4 *dst = ldsA + (k >> 5) * 64 + (k & 31);
            dst[0] = make_uint4(wh0 | (wm0 << 16), wh0 | (wh1 << 16), wm1 | (wh1 << 16), wh2 | (wm2 << 16));
            dst[32] = make_uint4(wh2 | (wh3 << 16), wm3 | (wh3 << 16), eh | (em << 16), el);
        }
    }
    __syncthreads();
}

// ALIGNED: hw % 64 == 0 -- a group of 64 consecutive vectors never straddles two images, so (image, position) of a
// group is wave-uniform and every address is a scalar base plus a per-lane offset that is computed once.
// PERM: the prepared image is permuted -- near-duplicate rows (a trained codebook's clusters: quantize.py:22-26,78) sit in one tile,
// so that the 32-code exact step resolves them instead of every vector running into the all-K scan; "index" below is then the key
// (original index << 16 | position): ties go to the lowest ORIGINAL index like the reference's argmin, the position finds the row.
template <int NT, bool ALIGNED, bool CONV, bool PROBE = false, bool PERM = false>
__device__ __forceinline__ void vq_filter_body(const VqArgs &a, unsigned char *smem, const unsigned int vblk)
{
    constexpr int NW = NT / 64, G = kVqfGroup;
    const float *__restrict__ z = a.z;
    const int64_t hw = a.hw, N = a.N;
    const int K = a.K, ntile = K >> 5;
    int64_t *__restrict__ idx_out = a.idx_out;
    float *__restrict__ zq_out = a.zq_out;
    uint4 *ldsA = reinterpret_cast<uint4 *>(smem);                                  // [K/32][64]
    float4 *cbs = reinterpret_cast<float4 *>(smem + vqf_rows_off(K));               // fp32 rows, at rowpos()
    float *ees = reinterpret_cast<float *>(smem + vqf_ees_off(K));                  // their squared norms, at eepos()
    unsigned int *okey = reinterpret_cast<unsigned int *>(smem + vqf_lds_bytes(K));  // PERM: [K] (original index << 16) | position
    __shared__ unsigned int s_max[4];        // [2..3]: the prepared image's (K, magic) tag
    __shared__ unsigned int s_key0;          // PERM: the key of original row 0
    __shared__ double s_wsum[NT / 64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31;    // column of the MFMA: which latent vector of the 32-vector tile
    const int hf = lane >> 5;   // K-slot half of the operands / row half of the result

    CGIC_STAMP(0);
    CGIC_BLK_BEGIN();
    // ---- groups of 64 vectors: the workgroup owns a contiguous range, wave w takes groups w, w + NW, ... of it.
    // (A shared counter balanced the waves no better -- a SIMD's total work is what it is -- and cost an LDS atomic and a
    // cross-lane loss reduction per group: with fixed shares a wave's loss partial is a fixed sum.)
    const int64_t ngroups = (N + G - 1) / G;
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
    // (image, position) of a vector
    auto divmod = [&](int64_t n, int64_t *b, int64_t *p) {
        if (((n | hw) >> 32) == 0) {
            const unsigned int q = (unsigned int)n / (unsigned int)hw;
            *b = q; *p = (int64_t)((unsigned int)n - q * (unsigned int)hw);
        } else {
            *b = n / hw; *p = n - *b * hw;
        }
    };
    // ALIGNED: (image, position) of the workgroup's first group by ONE division; a group's own follows by adding and
    // carrying on the scalar unit (there is no scalar divide, and a vector one costs ~20 instructions per group)
    int64_t wg_b = 0, wg_p = 0;
    if (ALIGNED) divmod(blk_lo * G, &wg_b, &wg_p);
    auto origin = [&](int64_t grp, int64_t *b, int64_t *p0) {
        int64_t bb = wg_b, pp = wg_p + (grp - blk_lo) * G;
        while (pp >= hw) { pp -= hw; ++bb; }
        *b = bb; *p0 = pp;
    };
    const int64_t lane_off = j;                                    // per-lane element offsets inside the group's image
    const int64_t out_off = (int64_t)(2 * hf) * hw + j;
    auto load_group = [&](int64_t grp, float (&out)[2][4]) {
        const int64_t n0 = grp * G;
        if (ALIGNED) {
            int64_t b, p0;
            origin(grp, &b, &p0);                      // wave-uniform
            const float *zb = z + b * 4 * hw + p0;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int c = 0; c < 4; ++c) out[t][c] = zb[c * hw + 32 * t + lane_off];
        } else {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int64_t n = n0 + 32 * t + j;
                int64_t b = 0, p = 0;
                const bool ok = n < N;
                if (ok) divmod(n, &b, &p);
#pragma unroll
                for (int c = 0; c < 4; ++c) out[t][c] = ok ? z[(b * 4 + c) * hw + p] : 0.f;
            }
        }
    };

    Conv1x1 conv;
    if (CONV) conv.load(a.conv_w, a.conv_b, a.conv_bias_first);
    // the first group's latents are requested before the codebook is staged: their HBM latency hides behind it
    float zn[2][4];
    int64_t cur = blk_lo + wave;
    if (cur < blk_hi) load_group(cur, zn);

    // ---- stage: the codebook's LDS image -- copied from the prepared image (cgic_vq_prepare_f32: inference, the codebook
    // does not change between launches) or derived from the fp32 rows here
    if (a.prep) {
        const uint4 *src = reinterpret_cast<const uint4 *>(a.prep);
        uint4 *dst = reinterpret_cast<uint4 *>(smem);
        const int n16 = (int)(vqf_lds_bytes(K) / 16);
#pragma unroll 8
        for (int i = tid; i < n16; i += NT) dst[i] = src[i];
        if (tid < 4) s_max[tid] = reinterpret_cast<const unsigned int *>(src + n16)[tid];
        __syncthreads();
        // the image carries (K, magic) behind the two maxima: an image of another K, or memory that is not an image at all,
        // is not trusted -- the workgroup derives its own from the fp32 rows (workgroup-uniform branch)
        if (s_max[2] != (unsigned int)K || s_max[3] != (PERM ? kPreparedMagicPerm : kPreparedMagic)) {
            __syncthreads();
            vqf_stage<NT>(a.cb, K, smem, s_max);
            if (PERM) { for (int c = tid; c < K; c += NT) okey[c] = ((unsigned int)c << 16) | (unsigned int)c; __syncthreads(); }
        } else if (PERM) {
            const unsigned int *ks = reinterpret_cast<const unsigned int *>(src + n16 + 1);
            for (int c = tid; c < K; c += NT) okey[c] = ks[c];
            __syncthreads();
        }
    } else {
        vqf_stage<NT>(a.cb, K, smem, s_max);
        if (PERM) { for (int c = tid; c < K; c += NT) okey[c] = ((unsigned int)c << 16) | (unsigned int)c; __syncthreads(); }
    }
    if (PERM) {
        for (int c = tid; c < K; c += NT) if ((okey[c] >> 16) == 0u) s_key0 = okey[c];
        __syncthreads();
    }
    const float Emax = __uint_as_float(s_max[0]), EEmax = __uint_as_float(s_max[1]);
    // the filter needs a finite, non-zero codebook (an all-zero one ties everywhere: exact path)
    const bool filter_ok = Emax > 0.f && EEmax > 0.f && EEmax < __builtin_inff() && Emax < __builtin_inff();
    // 2 Emax 2^b and EEmax 2^be in [2^13, 2^14)
    const int sb = 13 - (exponent_of(Emax) + 1), sbe = 13 - exponent_of(EEmax);
    CGIC_STAMP(1);
    // exponent window of the per-vector scale 2^a: 2^s, s = a + sb - sbe, must be a normal fp16
    const int a_cap = 15 + sbe - sb, a_min = -14 + sbe - sb;

    double sq = 0.0;                    // this lane's share of the loss, over all groups of the wave
#ifdef CGIC_PHASE_CLOCKS      // dev: per-wave (loop start, loop end) of the first 64 workgroups, slots 512 + 8 wg + wave of g_blk_t
    if (lane == 0 && vblk < 64 && wave < 8) g_blk_t[2 * (512 + 8 * vblk + wave)] = wall_clock64();
#endif
#ifndef CGIC_VQF_TURN_PRIO
#define CGIC_VQF_TURN_PRIO 1
#endif
#ifndef CGIC_VQF_NO_TURNS
    // Two waves share a SIMD (wave w and w + NW/2), and the arbiter serves the OLDER one first: waves 0..NW/2-1 ran their two
    // groups in 11.6 us and left, their younger mates then ran alone -- a lone wave's MFMAs and VALU work do not overlap -- until
    // 14.6 us (per-wave loop clocks, tools/probes/probe_vq_phases.py waves).  The mates take turns instead: group by group the one
    // that is behind gets the issue priority, so both have work until the end.
    int turn = wave >= NW / 2 ? 1 : 0;
#endif
    while (cur < blk_hi) {
#ifndef CGIC_VQF_NO_TURNS
        if (turn & 1) __builtin_amdgcn_s_setprio(CGIC_VQF_TURN_PRIO); else __builtin_amdgcn_s_setprio(0);
        ++turn;
#endif
        const int64_t grp = cur;
        const int64_t base = grp * G;
        CGIC_PHASE_T0();
        float zv[2][4];
        f16x8 bop[2];
        int qs[2];                      // scaled score = 2^qs x score
        bool unscalable[2];             // no exponent fits (or no usable codebook): exact path for that vector
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int c = 0; c < 4; ++c) zv[t][c] = zn[t][c];
            if (CONV) conv.apply(zv[t]);                // quant_conv on the four channels this lane holds anyway
            const float zmax = fmaxf(fmaxf(fabsf(zv[t][0]), fabsf(zv[t][1])), fmaxf(fabsf(zv[t][2]), fabsf(zv[t][3])));
            const int ez = exponent_of(zmax);
            int sa = 13 - ez;
            sa = sa < a_cap ? sa : a_cap;       // z small next to the codebook: keep 2^s <= 2^15 (z's pieces only lose bits that do not matter)
            // z huge next to the codebook (or not finite): no exponent fits -- exact path
            unscalable[t] = !filter_ok || sa < a_min || ez > 127 || !(zv[t][0] == zv[t][0] && zv[t][1] == zv[t][1] && zv[t][2] == zv[t][2] && zv[t][3] == zv[t][3]);
            sa = sa < a_min ? a_min : sa;
            qs[t] = sa + sb;
            const unsigned int sig = (unsigned int)(sa + sb - sbe + 15) << 10;      // fp16 bits of 2^s
            // slots 0..7 (hf 0): zh0 zh0 | zm0 zh1 | zh1 zm1 | zh2 zh2      slots 8..15 (hf 1): zm2 zh3 | zh3 zm3 | 2^s 2^s | 2^s 0
            // with (a, b) = (z0, z1) resp. (z2, z3) both halves hold P = am | bh << 16 and Q = bh | bm << 16
            unsigned int ah, am, bh, bm;
            split2h(ldexpf(hf ? zv[t][2] : zv[t][0], sa), ah, am);
            split2h(ldexpf(hf ? zv[t][3] : zv[t][1], sa), bh, bm);
            const unsigned int ch = h16(ldexpf(zv[t][2], sa));
            const unsigned int P = am | (bh << 16), Q = bh | (bm << 16);
            uint4 bb;
            bb.x = hf ? P : ah | (ah << 16);
            bb.y = hf ? Q : P;
            bb.z = hf ? sig | (sig << 16) : Q;
            bb.w = hf ? sig : ch | (ch << 16);
            bop[t] = __builtin_bit_cast(f16x8, bb);
        }
        // the next group's latents are in flight during this group's scan
        cur += NW;
        if (cur < blk_hi) load_group(cur, zn);
        CGIC_STAMP(2);
        CGIC_PHASE_ACC(0);

        // ---- scan: tiles of 32 codes, ping-pong -- the MFMAs of one tile run while the VALU digests the other.
        float m1[2], m2[2];
        m1[0] = m1[1] = m2[0] = m2[1] = __builtin_inff();
        {
            f32x16 X[2], Y[2];
            f32x16 zero16;
#pragma unroll
            for (int r = 0; r < 16; ++r) zero16[r] = 0.f;
            auto issue = [&](int T, f32x16 (&D)[2]) {
                const int TT = T < ntile ? T : ntile - 1;      // the last one is a harmless repeat of the final tile
                const f16x8 av = __builtin_bit_cast(f16x8, ldsA[TT * 64 + lane]);
                D[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bop[0], zero16, 0, 0, 0);
                D[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bop[1], zero16, 0, 0, 0);
            };
            auto digest = [&](int T, const f32x16 (&D)[2]) {
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    if (PROBE && T < ntile && base + 32 * t + j < N) {
                        // register r of this lane = row 8 (r / 4) + 4 hf + r % 4 of the tile = code 32 T + that row, column j = vector 32 t + j
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            a.probe_scores[(base + 32 * t + j) * K + 32 * T + 8 * (r >> 2) + 4 * hf + (r & 3)] = ldexpf(D[t][r], -qs[t]);
                    }
                    float u = __builtin_inff();        // seeded with a constant: a two-operand fminf() canonicalises both operands first
#ifdef CGIC_VQF_TREE
                    {
                        float ua = u, ub = u;
#pragma unroll
                        for (int r = 0; r < 8; r += 2) {
                            ua = __builtin_fminf(__builtin_fminf(ua, D[t][r]), D[t][r + 1]);
                            ub = __builtin_fminf(__builtin_fminf(ub, D[t][8 + r]), D[t][9 + r]);
                        }
                        u = __builtin_fminf(ua, ub);
                    }
#else
#pragma unroll
                    for (int r = 0; r < 16; r += 2) u = __builtin_fminf(__builtin_fminf(u, D[t][r]), D[t][r + 1]);     // v_min3_f32 on the raw MFMA outputs
#endif
                    // the tile's index rides in the low 5 mantissa bits of its minimum (one v_and_or_b32 instead of a
                    // compare + select per tile); the 2^-18 relative perturbation is part of the margin
                    u = __uint_as_float((__float_as_uint(u) & ~31u) | (unsigned int)T);
                    m2[t] = __builtin_amdgcn_fmed3f(m1[t], m2[t], u);     // second smallest tile value
                    // raw v_min_f32: fminf() adds two canonicalising v_max per call here
                    asm("v_min_f32 %0, %1, %2" : "=v"(m1[t]) : "v"(m1[t]), "v"(u));
                }
            };
            // (pinning this order with sched_barrier was measured slower: 26.1 vs 25.5 us)
            issue(0, X);
            for (int T = 0; T < ntile; T += 2) {
                issue(T + 1, Y);
                digest(T, X);
                issue(T + 2, X);
                digest(T + 1, Y);
            }
        }
        CGIC_STAMP(3);
        CGIC_PHASE_ACC(1);

        // ---- decide: ONE LANE PER VECTOR.  Lane L = 32 t + j takes vector (tile t, column j): its own scaled
        // (smallest, second) of the row half it scanned, the other half's from lane L ^ 32 (v_permlane32_swap), and its
        // four latent components, which it loaded itself.  Margin, candidate sets and the exact distances are lane-local.
        int wi = 0;
        {
            const uint2 s1a = rows32(__float_as_uint(m1[0])), s2a = rows32(__float_as_uint(m2[0]));     // tile 0: .x row half 0, .y row half 1
            const uint2 s1b = rows32(__float_as_uint(m1[1])), s2b = rows32(__float_as_uint(m2[1]));     // tile 1
            const float A1 = __uint_as_float(hf ? s1b.x : s1a.x), B1 = __uint_as_float(hf ? s1b.y : s1a.y);
            const float A2 = __uint_as_float(hf ? s2b.x : s2a.x), B2 = __uint_as_float(hf ? s2b.y : s2a.y);
            const float y0 = hf ? zv[1][0] : zv[0][0], y1 = hf ? zv[1][1] : zv[0][1];
            const float y2 = hf ? zv[1][2] : zv[0][2], y3 = hf ? zv[1][3] : zv[0][3];
            const int q = hf ? qs[1] : qs[0];
            const bool valid = base + lane < N;
            const float yy = sumsq4(y0, y1, y2, y3);
            const float mts = __builtin_fminf(A1, B1);                      // scaled
            const float mt = ldexpf(mts, -q);
            // S bounds ee_k + 2 sum|z_j e_kj| for every code that can matter: by the codebook maxima, and -- the winner
            // and every code that can beat it lie within sqrt(D) of z, D = zz + f_min + slack -- by
            // (|z| + sqrt D)^2 + 2 |z| (|z| + sqrt D); the smaller of the two (v_sqrt_f32 is good to 1 ulp: x 1.001)
            const float S0 = EEmax + 2.0f * Emax * (((fabsf(y0) + fabsf(y1)) + fabsf(y2)) + fabsf(y3));
            const float M0 = 1.3e-5f * S0 + 2.5e-7f * yy + 1e-30f;
            const float D = fmaxf(yy * 1.0001f + mt + 2.0f * M0, 0.f);
            const float nz = __builtin_amdgcn_sqrtf(yy) * 1.001f, sd = __builtin_amdgcn_sqrtf(D) * 1.001f;
            const float S1 = (nz + sd) * (3.0f * nz + sd);
            const float S = S1 < S0 ? S1 : S0;                  // (a NaN S1 keeps S0)
            const float M = 1.3e-5f * S + 2.5e-7f * yy + 1e-30f;
            // + 0.02 scaled units: pieces in fp16's subnormal range (5 x 2^-10 per score, both sides)
            float thr = ldexpf(mt + M, q) + 0.02f;
            thr += fabsf(thr) * 2.4e-7f;
            // anything not comparable (NaN / Inf anywhere above) must count as a candidate: test the negation
            const bool cA1 = !(A1 > thr), cB1 = !(B1 > thr);
            const bool deep = !(A2 > thr) || !(B2 > thr);       // a second tile of one half is a candidate: a third may be too
            const bool other = !(B1 < A1) ? cB1 : cA1;           // the half that does not hold the minimum has one as well
            const bool flag = valid && (deep || (hf ? unscalable[1] : unscalable[0]) || !(cA1 || cB1));
            if (PROBE && valid) {
                float *ax = a.probe_aux + (base + lane) * 6;
                ax[0] = mt; ax[1] = M; ax[2] = ldexpf(thr, -q); ax[3] = flag ? 1.f : 0.f; ax[4] = (float)q; ax[5] = S;
            }
            const bool firstB = B1 < A1;                        // which half holds the minimum
            auto tile_of = [&](float v) -> int {
                int T = __float_as_int(v) & 31;
                return T < ntile ? T : 0;
            };
            // exact fp32 on the 16 codes of the best (tile, row half); descending, the lowest index wins ties
            float d = __builtin_inff();
            {
                const int cb0 = 32 * tile_of(firstB ? B1 : A1) + (firstB ? 4 : 0);
                const float4 *rb = cbs + rowpos(cb0);          // (a set stays inside one tile: same padding for all 16)
                const float *eb = ees + eepos(cb0);
                int rs = 0;                                     // the winner's offset in the set: a select between inline constants
                unsigned int kb = 0xFFFFFFFFu;                  // PERM: the winner's key
#pragma unroll
                for (int r4 = 3; r4 >= 0; --r4) {
                    const float4 en = *reinterpret_cast<const float4 *>(&eb[8 * r4]);
                    uint4 k4 = make_uint4(0u, 0u, 0u, 0u);
                    if (PERM) k4 = *reinterpret_cast<const uint4 *>(&okey[cb0 + 8 * r4]);
#pragma unroll
                    for (int r = 3; r >= 0; --r) {
                        const float dd = dist_row(y0, y1, y2, y3, yy, rb[8 * r4 + r], r == 0 ? en.x : r == 1 ? en.y : r == 2 ? en.z : en.w);
                        if (PERM) {
                            const unsigned int kk = r == 0 ? k4.x : r == 1 ? k4.y : r == 2 ? k4.z : k4.w;
                            const bool take = dd < d || (dd == d && kk < kb);
                            d = take ? dd : d;
                            kb = take ? kk : kb;
                        } else {
                            const bool take = dd <= d;
                            d = take ? dd : d;
                            rs = take ? 8 * r4 + r : rs;
                        }
                    }
                }
                wi = PERM ? (int)kb : cb0 + rs;
            }
            // ... and on the other half's best tile where it is a candidate too (lexicographic merge)
            if (__ballot(valid && other && !flag)) {
                CGIC_DBG_COUNT(1, 1);
                if (a.stats && lane == 0) atomicAdd(&a.stats[2], 1u);
                const int cb1 = 32 * tile_of(firstB ? A1 : B1) + (firstB ? 0 : 4);
                const float4 *rb = cbs + rowpos(cb1);
                const float *eb = ees + eepos(cb1);
                const bool doit = other;
#pragma unroll
                for (int r4 = 3; r4 >= 0; --r4) {
                    const float4 en = *reinterpret_cast<const float4 *>(&eb[8 * r4]);
                    uint4 k4 = make_uint4(0u, 0u, 0u, 0u);
                    if (PERM) k4 = *reinterpret_cast<const uint4 *>(&okey[cb1 + 8 * r4]);
#pragma unroll
                    for (int r = 3; r >= 0; --r) {
                        const int c = PERM ? (int)(r == 0 ? k4.x : r == 1 ? k4.y : r == 2 ? k4.z : k4.w) : cb1 + 8 * r4 + r;
                        const float dd = dist_row(y0, y1, y2, y3, yy, rb[8 * r4 + r], r == 0 ? en.x : r == 1 ? en.y : r == 2 ? en.z : en.w);
                        const bool take = doit && (dd < d || (dd == d && (unsigned int)c < (unsigned int)wi));
                        d = take ? dd : d;
                        wi = take ? c : wi;
                    }
                }
            }
            const unsigned long long fmask = __ballot(flag);               // one bit per vector, wave-uniform
            const int nflag = __builtin_popcountll(fmask);
            CGIC_DBG_COUNT(0, nflag);
            if (a.stats && nflag != 0 && lane == 0) { atomicAdd(&a.stats[0], (unsigned int)nflag); if (nflag > kVqfBulk) atomicAdd(&a.stats[1], 1u); }
            if (nflag != 0 && nflag <= kVqfBulk) {
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
                    int bi = PERM ? 0x7FFFFFFF : 0;                      // (PERM: "none yet" sorts behind every key)
#pragma unroll 4
                    for (int c = K - 64 + lane; c >= 0; c -= 64) {       // descending: the lowest index wins ties
                        const float dd = dist_row(s0, s1, s2, s3, ss, cbs[rowpos(c)], ees[eepos(c)]);
                        if (PERM) {
                            const int kk = (int)okey[c];
                            const bool take = dd < bd || (dd == bd && (unsigned int)kk < (unsigned int)bi);
                            bd = take ? dd : bd;
                            bi = take ? kk : bi;
                        } else {
                            const bool take = dd <= bd;
                            bd = take ? dd : bd;
                            bi = take ? c : bi;
                        }
                    }
#ifndef CGIC_VQF_SHFLMIN
                    wave_argmin(bd, bi);
#else
#pragma unroll
                    for (int off = 1; off < 64; off <<= 1) {
                        const float od = __shfl_xor(bd, off, kWave);
                        const int oi = __shfl_xor(bi, off, kWave);
                        const bool take = od < bd || (od == bd && (unsigned int)oi < (unsigned int)bi);
                        bd = take ? od : bd;
                        bi = take ? oi : bi;
                    }
#endif
                    wi = lane == v ? bi : wi;
                }
            } else if (nflag != 0) {
                // many near-ties (degenerate codebooks, zz >> ee, non-finite input): the exact fp32-MFMA scan of
                // vq_mfma_body for the whole group.  Vector 16 tt + col of the group sits in lane 16 tt + col;
                // the 16x16x4 MFMA wants z[vector 16 tt + (lane & 15)][k = lane >> 4] as its B operand.
                const int col = lane & 15, g4 = lane >> 4;
                float zb4[4], z0[4], z1[4], z2[4], z3[4], zz[4], best[4];
                int bt2[4];
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) {
                    const int src = 16 * tt + col;
                    z0[tt] = __shfl(y0, src, kWave); z1[tt] = __shfl(y1, src, kWave);
                    z2[tt] = __shfl(y2, src, kWave); z3[tt] = __shfl(y3, src, kWave);
                    zb4[tt] = g4 == 0 ? z0[tt] : g4 == 1 ? z1[tt] : g4 == 2 ? z2[tt] : z3[tt];
                    zz[tt] = sumsq4(z0[tt], z1[tt], z2[tt], z3[tt]);
                    best[tt] = __builtin_inff();
                    bt2[tt] = PERM ? 0x7FFFFFFF : 0;                 // (PERM: the best KEY so far instead of its 16-code tile)
                }
                for (int ct = 0; ct < (K >> 4); ++ct) {
                    const float av = reinterpret_cast<const float *>(cbs)[rowpos(16 * ct + col) * 4 + g4];
                    const f32x4 e4 = *reinterpret_cast<const f32x4 *>(&ees[eepos(16 * ct + 4 * g4)]);
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) {
                        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, zb4[tt], acc, 0, 0, 0);
                        const float d0 = __builtin_fmaf(-2.0f, acc[0], zz[tt] + e4[0]);
                        const float d1 = __builtin_fmaf(-2.0f, acc[1], zz[tt] + e4[1]);
                        const float d2 = __builtin_fmaf(-2.0f, acc[2], zz[tt] + e4[2]);
                        const float d3 = __builtin_fmaf(-2.0f, acc[3], zz[tt] + e4[3]);
                        if (PERM) {
                            // ties across the whole codebook go to the lowest ORIGINAL index: the smallest key among this step's minima
                            const uint4 k4 = *reinterpret_cast<const uint4 *>(&okey[16 * ct + 4 * g4]);
                            const float m = __builtin_fminf(__builtin_fminf(d0, d1), __builtin_fminf(d2, d3));
                            unsigned int kq = 0x7FFFFFFFu;
                            kq = d0 == m && k4.x < kq ? k4.x : kq;
                            kq = d1 == m && k4.y < kq ? k4.y : kq;
                            kq = d2 == m && k4.z < kq ? k4.z : kq;
                            kq = d3 == m && k4.w < kq ? k4.w : kq;
                            const bool take = m < best[tt] || (m == best[tt] && kq < (unsigned int)bt2[tt]);
                            best[tt] = take ? m : best[tt];
                            bt2[tt] = take ? (int)kq : bt2[tt];
                        } else {
                            const float q1 = __builtin_fminf(__builtin_fminf(best[tt], d0), d1);
                            const float q2 = __builtin_fminf(__builtin_fminf(q1, d2), d3);
                            bt2[tt] = q2 < best[tt] ? ct : bt2[tt];
                            best[tt] = q2;
                        }
                    }
                }
                int w16[4];
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) {
                    float dq = best[tt];
                    int i = bt2[tt];
                    if (!PERM) {
                        const int c0 = 16 * bt2[tt] + 4 * g4;
                        i = c0;
#pragma unroll
                        for (int r = 3; r >= 0; --r)
                            i = dist_row(z0[tt], z1[tt], z2[tt], z3[tt], zz[tt], cbs[rowpos(c0 + r)], ees[eepos(c0 + r)]) == dq ? c0 + r : i;
                    }
                    colargmin(dq, i);
                    w16[tt] = i;
                }
                // lane 16 tt + col holds all four tiles' results for its own column: pick its tile
                wi = g4 == 0 ? w16[0] : g4 == 1 ? w16[1] : g4 == 2 ? w16[2] : w16[3];
            }
        }
        CGIC_STAMP(4);
        CGIC_PHASE_ACC(2);

        // ---- outputs: lane (column j, half hf) owns channels 2 hf, 2 hf + 1 of vector (tile t, column j); lane L the
        // index of vector L
        if (PERM) {
            // a key that never got set (every distance NaN: nothing wins) is original row 0, like the plain path's index 0
            if ((unsigned int)wi >= ((unsigned int)K << 16)) wi = (int)s_key0;
        }
        // indices and z_q leave as nontemporal stores: 6.3 MB per launch that nobody in this launch reads again -- as ordinary stores they
        // sat dirty in the L2s until the end-of-kernel write-back (fused launch 23.2 -> 22.4 us, same box A/B: profiles/r06_vq_ab.md)
#ifndef CGIC_VQF_PLAIN_STORES
#define CGIC_VQF_STORE(p, v) __builtin_nontemporal_store((v), (p))
#else
#define CGIC_VQF_STORE(p, v) (*(p) = (v))
#endif
        if (idx_out && base + lane < N) CGIC_VQF_STORE(&idx_out[base + lane], (int64_t)(PERM ? wi >> 16 : wi));
        if (PERM) wi &= 0xFFFF;                                   // from here on: where the row sits
        if (zq_out || a.sq_partial) {
            const uint2 wt = rows32((unsigned int)wi);           // .x: tile 0's winners (lanes 0..31), .y: tile 1's
            int64_t gb = 0, gp0 = 0;
            if (ALIGNED) origin(grp, &gb, &gp0);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int w = (int)(t ? wt.y : wt.x);
                const float2 e = *reinterpret_cast<const float2 *>(reinterpret_cast<const float *>(cbs) + rowpos(w) * 4 + 2 * hf);
                const float za = hf ? zv[t][2] : zv[t][0], zb = hf ? zv[t][3] : zv[t][1];
                const float da = e.x - za, db = e.y - zb;
                const int64_t n = base + 32 * t + j;
                if (ALIGNED || n < N) {
                    if (zq_out) {
                        if (ALIGNED) {
                            float *qb = zq_out + (gb * 4 * hw + gp0 + 32 * t) + out_off;
                            CGIC_VQF_STORE(&qb[0], za + da);
                            CGIC_VQF_STORE(&qb[hw], zb + db);
                        } else {
                            int64_t b, p;
                            divmod(n, &b, &p);
                            zq_out[(b * 4 + 2 * hf) * hw + p] = za + da;
                            zq_out[(b * 4 + 2 * hf + 1) * hw + p] = zb + db;
                        }
                    }
                    sq += (double)da * (double)da;
                    sq += (double)db * (double)db;
                }
            }
        }
        CGIC_PHASE_ACC(3);
    }

#ifndef CGIC_VQF_NO_TURNS
    __builtin_amdgcn_s_setprio(0);
#endif
#ifdef CGIC_PHASE_CLOCKS
    if (lane == 0 && vblk < 64 && wave < 8) g_blk_t[2 * (512 + 8 * vblk + wave) + 1] = wall_clock64();
#endif
    CGIC_STAMP(5);
#ifdef CGIC_VQF_NO_TAIL           // dev: no reduction, no barrier, no hand-off (the loss is garbage): what the tail costs
    if (false) {
#else
    if (a.sq_partial) {
#endif
        // Only wave 0 stays for the hand-off: the other waves leave at the barrier WITHOUT draining their z_q /
        // index stores (an s_waitcnt vmcnt(0) in every wave before the barrier cost ~4 us at the end of every
        // workgroup); wave 0's own stores are long complete by the time it has waited for the others.
        sq = wave_sum_f64(sq);              // fixed association, on the VALU
        if (lane == 0) s_wsum[wave] = sq;
        __syncthreads();
        if (wave == 0) {
            double bs = 0.0;
#pragma unroll
            for (int w = 0; w < NW; ++w) bs += s_wsum[w];
#ifndef CGIC_VQF_NO_HANDOFF      // dev: reduction + barrier, but no store / ticket / last-workgroup sum
            finish_loss_wave(bs, a.sq_partial, a.ticket, (double)N * 4.0, a.beta, a.legacy, a.loss, vblk, a.nblk, a.tail_mode);
#else
            if (bs == 12345.678) a.loss[0] = (float)bs;
#endif
        }
    }
    CGIC_STAMP(6);
    CGIC_BLK_END();
}

#ifndef CGIC_VQF_THREADS
#define CGIC_VQF_THREADS 512
#endif
constexpr int kVqfThreads = CGIC_VQF_THREADS;      // one workgroup per CU, 2 waves per SIMD.  Alone at B=64 x 64x64 latents 512 / 768 / 1024 threads are within 1 us of each other; with several batches in flight (bench.py --lanes 4) 512 leaves a third of the register file to the other batches' kernels: 86.9 vs 83.4 (768) vs 82.9 (1024) GPixel/s
// 128 registers per lane (4-10 spilled, 20-44 bytes of scratch) instead of 144-150: two 512-thread workgroups then fit a CU's
// register file, so a ROUTER workgroup of the fused launch (same launch => same allocation) shares its CU with a VQ workgroup
// instead of holding the CU to itself for ~12 us: fused launch 26.3 -> 24.1 us at B=64 (the VQ kernel alone: 23.3 -> 23.8 on the
// same GPU), no uneven split of the VQ shares needed any more.
#ifndef CGIC_VQF_VGPR_CAP
#define CGIC_VQF_VGPR_CAP 128
#endif
#if CGIC_VQF_VGPR_CAP
// (amdgpu_num_vgpr counts VGPR + AGPR on gfx950: the attribute carries half the cap)
#define CGIC_VQF_BOUNDS __launch_bounds__(kVqfThreads, kVqfThreads / 256 > 1 ? kVqfThreads / 256 : 1) __attribute__((amdgpu_num_vgpr(CGIC_VQF_VGPR_CAP / 2)))
#else
#define CGIC_VQF_BOUNDS __launch_bounds__(kVqfThreads, kVqfThreads / 256 > 1 ? kVqfThreads / 256 : 1)
#endif

template <bool ALIGNED, bool CONV>
__global__ CGIC_VQF_BOUNDS void vq_filter_kernel(VqArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_f[];
    vq_filter_body<kVqfThreads, ALIGNED, CONV>(a, smem_f, blockIdx.x);
}

// PERM instantiations (a prepared image whose rows were packed by cluster: cgic_vq_prepare_f32): kernels of their own, so that the
// plain kernels' code and register allocation are untouched
template <bool ALIGNED>
__global__ CGIC_VQF_BOUNDS void vq_filter_perm_kernel(VqArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_f[];
    vq_filter_body<kVqfThreads, ALIGNED, false, false, true>(a, smem_f, blockIdx.x);
}

// Telemetry instantiation (cgic_vq_filter_probe_f32): the SAME body, with every approximate score and every decision threshold
// written out -- what tools/stress_vq.py --telemetry and tests/test_gpu_stress.py compare with the budgeted error bound.
template <bool ALIGNED>
__global__ CGIC_VQF_BOUNDS void vq_filter_probe_kernel(VqArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_f[];
    vq_filter_body<kVqfThreads, ALIGNED, false, true>(a, smem_f, blockIdx.x);
}

// The fused launch of the filter path (see vq_router_kernel).  A VQ workgroup takes most of a CU's register file, and
// dynamic LDS / the VGPR budget are per launch, so a router workgroup does not share a CU with one: behind the VQ
// workgroups it only starts when the VQ is over.  The router workgroups therefore come FIRST (`nrouter` of them, one CU
// each for ~11 us); the VQ workgroups that have to wait for those CUs own fewer groups, the others more.
template <bool ALIGNED, bool CONV, bool SPLIT = false>
__global__ CGIC_VQF_BOUNDS void vq_filter_router_kernel(VqArgs a, RouterArgs r, unsigned int nrouter, unsigned int router_behind)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_f[];
    // router workgroups in front of the VQ workgroups (they hold their CUs first; the VQ shares are uneven) or behind them
    // (every VQ workgroup gets a CU at once and the router workgroups move in beside them)
    const unsigned int rb = router_behind ? a.nblk : 0u, vb = router_behind ? 0u : nrouter;
    if (blockIdx.x - rb < nrouter) {
#ifdef CGIC_ROUTER_PRIO
        __builtin_amdgcn_s_setprio(CGIC_ROUTER_PRIO);
#endif
        router_body<kVqfThreads, false, SPLIT>(r, (int64_t)(blockIdx.x - rb), smem_f);
        loss_collect_by_routers(a, nrouter);
        return;
    }
    vq_filter_body<kVqfThreads, ALIGNED, CONV>(a, smem_f, blockIdx.x - vb);
}

template <bool ALIGNED>
__global__ CGIC_VQF_BOUNDS void vq_filter_router_perm_kernel(VqArgs a, RouterArgs r, unsigned int nrouter, unsigned int router_behind)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_f[];
    const unsigned int rb = router_behind ? a.nblk : 0u, vb = router_behind ? 0u : nrouter;
    if (blockIdx.x - rb < nrouter) {
        router_body<kVqfThreads>(r, (int64_t)(blockIdx.x - rb), smem_f);
        loss_collect_by_routers(a, nrouter);
        return;
    }
    vq_filter_body<kVqfThreads, ALIGNED, false, false, true>(a, smem_f, blockIdx.x - vb);
}

// The same launch for several shape groups at once (cgic_common.h: launch groups): every group keeps its own VQ shares, router
// workgroups and loss ticket; a workgroup's role is read off its block index inside its group.
struct VqfrArgs {
    VqArgs a;
    RouterArgs r;
    unsigned int nrouter, router_behind;
};
template <bool ALIGNED, bool SPLIT = false>
__global__ CGIC_VQF_BOUNDS void vq_filter_router_grouped_kernel(Grouped<VqfrArgs> g)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_f[];
    Blk blk;
    const VqfrArgs &p = g.a[group_locate(g, &blk)];
    const unsigned int rb = p.router_behind ? p.a.nblk : 0u, vb = p.router_behind ? 0u : p.nrouter;
    if (blk.x - rb < p.nrouter) {
        router_body<kVqfThreads, false, SPLIT>(p.r, (int64_t)(blk.x - rb), smem_f);
        loss_collect_by_routers(p.a, p.nrouter);
        return;
    }
    vq_filter_body<kVqfThreads, ALIGNED, false>(p.a, smem_f, blk.x - vb);
}

// The codebook's LDS image, computed ONCE (cgic_vq_prepare_f32) instead of by every workgroup of every launch: inference
// runs thousands of launches against one codebook, and deriving the image (two block-wide maxima, 11 fp16 splits per row, three
// barriers) was ~3 us at the head of every ~23 us launch.  One workgroup; the image is followed by the two maxima.
__global__ __launch_bounds__(kVqfThreads) void vq_prepare_kernel(const float *__restrict__ cb, int K, uint4 *__restrict__ out,
                                                                 const unsigned short *__restrict__ perm)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_f[];
    __shared__ unsigned int s_max[4];
    const int n16 = (int)(vqf_lds_bytes(K) / 16);
    uint4 *img = reinterpret_cast<uint4 *>(smem_f);
    for (int i = threadIdx.x; i < n16; i += kVqfThreads) img[i] = make_uint4(0u, 0u, 0u, 0u);      // the padding rows: defined bytes
    __syncthreads();
    vqf_stage<kVqfThreads>(cb, K, smem_f, s_max, perm);
    for (int i = threadIdx.x; i < n16; i += kVqfThreads) out[i] = img[i];
    if (threadIdx.x == 0) out[n16] = make_uint4(s_max[0], s_max[1], (unsigned int)K, perm ? kPreparedMagicPerm : kPreparedMagic);
    // behind the tag: the keys of a permuted image, (original index << 16) | position (all zero otherwise: defined bytes)
    unsigned int *keys = reinterpret_cast<unsigned int *>(out + n16 + 1);
    for (int k = threadIdx.x; k < K; k += kVqfThreads) keys[k] = perm ? ((unsigned int)perm[k] << 16) | (unsigned int)k : 0u;
}

// Plain-VALU restatement: one latent vector per thread, codebook broadcast from
// LDS.  Independent of the MFMA path; used to cross-check it on hardware.
__global__ __launch_bounds__(kVqThreads) void vq_valu_kernel(
    const float *__restrict__ z, int64_t hw, int64_t N, const float *__restrict__ cb, int K,
    int64_t *__restrict__ idx_out, float *__restrict__ zq_out, double *__restrict__ sq_partial,
    unsigned int *__restrict__ ticket, float beta, int legacy, float *__restrict__ loss,
    const float *__restrict__ conv_w, const float *__restrict__ conv_b, int conv_bias_first)
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
        if (conv_w) {
            Conv1x1 conv;
            conv.load(conv_w, conv_b, conv_bias_first);
            float v[4] = {z0, z1, z2, z3};
            conv.apply(v);
            z0 = v[0]; z1 = v[1]; z2 = v[2]; z3 = v[3];
        }
        float zz = sumsq4(z0, z1, z2, z3);
        // lowest index among the minimal distances; a NaN distance never wins (index 0 if every distance is NaN) -- the
        // same rule as the exact scans of the filter path (torch.argmin would return the first NaN: see cgic_hip.h)
        float best = __builtin_inff();
        int bi = 0;
        bool any = false;
        for (int k = 0; k < K; ++k) {
            float4 e = cbs[k];
            float mm = z0 * e.x;
            mm = __builtin_fmaf(z1, e.y, mm);
            mm = __builtin_fmaf(z2, e.z, mm);
            mm = __builtin_fmaf(z3, e.w, mm);
            float s = zz + ee[k];
            float d = __builtin_fmaf(-2.0f, mm, s);
            bool take = d == d && (!any || d < best);
            any = any || d == d;
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
    // (inside a launch group: recorded at its position and launched on its own -- histograms of several groups add into one table)
    const size_t lds = sizeof(unsigned int) * (size_t)K;
    return launch_or_record(KID_NONE, dim3(nblk), dim3(1024), lds, n, s, [=] {
        hipLaunchKernelGGL(index_hist_kernel, dim3(nblk), dim3(1024), lds, s, idx, n, K, (unsigned long long *)hist);
        return launch_check("index_hist_kernel"); });
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

// telemetry target of the filter path's launches (cgic_vq_stats): process-wide, read when a launch is enqueued / captured
static std::atomic<unsigned int *> g_vq_stats{nullptr};

static int prepared_check(const void *prepared, int K)
{
    CGIC_REQUIRE(!prepared || (((uintptr_t)prepared & 15) == 0 && K % 64 == 0 && K <= 1024), CGIC_ERR_INVALID,
                 "vq: a prepared codebook image must be 16-byte aligned and made for this K (cgic_vq_prepare_f32)");
    return CGIC_OK;
}

static int conv_check(const cgic_conv1x1 *qc)
{
    CGIC_REQUIRE(!qc || qc->weight, CGIC_ERR_INVALID, "vq: quant_conv without a weight");
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
    a.n_early = a.g_early = a.g_late = 0; a.tail_mode = 0; a.conv_w = a.conv_b = nullptr; a.conv_bias_first = 0; a.prep = nullptr; a.stats = nullptr; a.probe_scores = a.probe_aux = nullptr;
    size_t lds = sizeof(float) * (size_t)K * 5;
    if (!router) {
        int rc = ensure_dynamic_lds((const void *)vq_mfma_kernel<ZT>, lds);
        if (rc) return rc;
        hipLaunchKernelGGL(vq_mfma_kernel<ZT>, dim3(a.nblk), dim3(kVqThreads), lds, s, a);
        return launch_check("vq_mfma_kernel");
    }
    if (router_lds > lds) lds = router_lds;
    int rc = ensure_dynamic_lds((const void *)vq_router_kernel<ZT>, lds);
    if (rc) return rc;
    // (no grouped form: inside a launch group this position is launched group by group)
    const RouterArgs r = *router;
    const dim3 grid(a.nblk + (unsigned int)router_blocks);
    return launch_or_record(KID_NONE, grid, dim3(kVqThreads), lds, a, s, [=] {
        hipLaunchKernelGGL(vq_router_kernel<ZT>, grid, dim3(kVqThreads), lds, s, a, r);
        return launch_check("vq_router_kernel"); });
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

#ifdef CGIC_DEV_KNOBS
static int dev_knob(const char *name) { const char *v = getenv(name); return v ? atoi(v) : 0; }
#else
static int dev_knob(const char *) { return 0; }      // the environment knobs exist in `make dbg` builds only
#endif

static bool prepared_is_perm(const void *prepared);      // (below, next to cgic_vq_prepare_f32)

template <bool ALIGNED, bool CONV>
static int launch_filter(const float *z, int64_t hw, int64_t N, const float *cb, int K, int64_t *idx, float *zq,
                         VqWs ws, float beta, int legacy, float *loss, hipStream_t s, const RouterArgs *router,
                         int64_t router_blocks, size_t router_lds, const cgic_conv1x1 *qc, const void *prepared)
{
    int cus = 0;
    int rc = device_cu_count(&cus);
    if (rc) return rc;
    const int64_t ngroups = (N + kVqfGroup - 1) / kVqfGroup;
    // one resident workgroup per CU; its waves take groups from a counter.  Fewer groups than CUs x waves: spread them
    // over the CUs first (a small batch then costs one staging + one group per CU, whatever the waves per workgroup)
    if (dev_knob("CGIC_VQ_WGS_PER_CU") > 1) cus *= dev_knob("CGIC_VQ_WGS_PER_CU");      // dev: several resident workgroups per CU
    if (group_recording()) {          // one shape group of a grouped launch: its share of the chip
        cus = (int)((double)cus * group_cu_share() + 0.5);
        cus = cus < 1 ? 1 : cus;
    }
    int64_t nblk = ngroups < cus ? ngroups : cus;
    VqArgs a;
    a.z = z; a.hw = hw; a.N = N; a.cb = cb; a.K = K; a.idx_out = idx; a.zq_out = zq;
    a.sq_partial = nullptr; a.ticket = ws.ticket; a.beta = beta; a.legacy = legacy; a.loss = loss;
#ifdef CGIC_VQF_DRAIN_HANDOFF
    a.sq_partial = loss ? ws.partial : nullptr;
    if (false) {
#else
    if (loss) {
#endif
        // the ticket and, behind it, one 8-byte partial per workgroup: library-owned, zero when handed out, zeroed again by the launch
        // (finish_loss_wave); the caller's workspace is not touched by this path
        unsigned int *t = nullptr;
        rc = acquire_tickets(s, 1 + (int)((nblk + 7) / 8), &t);
        if (rc) return rc;
        a.ticket = t;
        a.sq_partial = reinterpret_cast<double *>(t + kTicketStride);
    }
    a.nblk = (unsigned int)nblk;
#if defined(CGIC_VQF_ROUTER_TAIL)
    a.tail_mode = (router && loss) ? 1u : 0u;       // dev A/B: the routers sum the loss
#elif !defined(CGIC_VQF_TICKET_TAIL) && !defined(CGIC_VQF_DRAIN_HANDOFF)
    a.tail_mode = loss ? 2u : 0u;                   // workgroup 0 collects (VqArgs::tail_mode)
#else
    a.tail_mode = 0u;
#endif
    a.conv_w = CONV ? qc->weight : nullptr; a.conv_b = CONV ? qc->bias : nullptr; a.conv_bias_first = CONV ? qc->bias_first : 0;
    a.prep = prepared;
    a.stats = g_vq_stats.load(std::memory_order_relaxed);
    a.probe_scores = a.probe_aux = nullptr;
    // groups per workgroup.  Router workgroups in front: the `late` VQ workgroups that must wait for a router's CU
    // (~11 us at 256x256, ~`delta` groups of VQ work) own `g_late` groups, the others `g_early`, a multiple of 4
    int64_t per = (ngroups + nblk - 1) / nblk, g_early = per, g_late = per, n_early = nblk;
    const int64_t late = router ? nblk + router_blocks - cus : 0;
    bool router_first = false;
#if CGIC_VQF_VGPR_CAP && CGIC_VQF_VGPR_CAP <= 128
    // A router workgroup can share its CU with a VQ workgroup.  Behind the VQ workgroups in the grid (every VQ workgroup gets a
    // CU at once and keeps its even share, the routers move in beside them) the router is free as long as it ends before the VQ
    // does -- beside an issue-bound VQ workgroup it runs ~1.6x slower than alone: 64 images of 256x256 24.1 us fused against
    // 23.5 for the VQ alone (in front with even shares: 27.1 -- the VQ workgroups pair up on the free CUs).  Few large tiles
    // (8 of 768x768: router 21 us alone, VQ 26) keep the older scheme: routers in front, uneven VQ shares.
    const double t_router = 10.9 + 0.000275 * (double)hw, t_vq = 3.0 + 1.25 * (double)per;
    const bool coresident = router && 1.6 * t_router <= t_vq;
#else
    const bool coresident = false;
#endif
    if (!coresident && late > 0 && late < nblk && !dev_knob("CGIC_VQ_NOSPLIT")) {
        // how long a router workgroup holds its CU, in groups of VQ work (~1.05 us each per workgroup): measured 12 us at
        // 64x64 latents, 21 us at 192x192 (with its row bands)
        const int64_t delta = (int64_t)((10.9 + 0.000275 * (double)hw) / 1.05 + 0.5);
        const int force_ge = dev_knob("CGIC_VQ_GE");
        for (int64_t ge = force_ge ? force_ge : (per / 4 + 1) * 4; ge <= per + 28; ge += 4) {
            const int64_t rest = ngroups - (nblk - late) * ge;
            const int64_t gl = rest > 0 ? (rest + late - 1) / late : 0;
            // (gl == 0: the router outlasts the whole VQ -- the early workgroups simply take everything)
            if (gl + delta <= ge || gl == 0) { g_early = ge; g_late = gl; n_early = nblk - late; router_first = true; break; }
        }
    }
    a.n_early = (unsigned int)n_early; a.g_early = (unsigned int)g_early; a.g_late = (unsigned int)g_late;
    size_t lds = vqf_lds_bytes(K);
    // a prepared image packed by cluster takes the PERM kernels (their own instantiations; inside a launch group and with a fused
    // quant_conv the plain kernels run: they do not trust the permuted image's tag and derive their own)
    const bool perm = !CONV && !group_recording() && prepared_is_perm(prepared);
    if (perm) lds += 4 * (size_t)K;
    if (!router) {
        if (perm) {
            if constexpr (!CONV) {
                rc = ensure_dynamic_lds((const void *)vq_filter_perm_kernel<ALIGNED>, lds);
                if (rc) return rc;
                hipLaunchKernelGGL((vq_filter_perm_kernel<ALIGNED>), dim3(a.nblk), dim3(kVqfThreads), lds, s, a);
                return launch_check("vq_filter_perm_kernel");
            }
        }
        rc = ensure_dynamic_lds((const void *)vq_filter_kernel<ALIGNED, CONV>, lds);
        if (rc) return rc;
        hipLaunchKernelGGL((vq_filter_kernel<ALIGNED, CONV>), dim3(a.nblk), dim3(kVqfThreads), lds, s, a);
        return launch_check("vq_filter_kernel");
    }
    if (router_lds > lds) lds = router_lds;
    if (perm) {
        if constexpr (!CONV) {
            rc = ensure_dynamic_lds((const void *)vq_filter_router_perm_kernel<ALIGNED>, lds);
            if (rc) return rc;
            const dim3 grid_p(a.nblk + (unsigned int)router_blocks);
            const unsigned int behind = router_first ? 0u : 1u;
            hipLaunchKernelGGL((vq_filter_router_perm_kernel<ALIGNED>), grid_p, dim3(kVqfThreads), lds, s, a, *router, (unsigned int)router_blocks, behind);
            return launch_check("vq_filter_router_perm_kernel");
        }
    }
    VqfrArgs p;
    p.a = a; p.r = *router; p.nrouter = (unsigned int)router_blocks; p.router_behind = router_first ? 0u : 1u;
    const dim3 grid(a.nblk + (unsigned int)router_blocks);
    // row bands that share a threshold band's re-evaluation run the SPLIT instantiation (cgic_router_dev.h: router_body)
    // ... and images with a workgroup of their own whose band is long start over with the launch's refinement queues (round 6)
    const bool split = router->rq.nq != 0;
    if (split) {
        rc = ensure_dynamic_lds((const void *)vq_filter_router_kernel<ALIGNED, CONV, true>, lds);
        if (rc) return rc;
        return launch_or_record(CONV ? KID_NONE : ALIGNED ? KID_VQF_ROUTER_AL : KID_VQF_ROUTER_UN, grid, dim3(kVqfThreads), lds, p, s, [=] {
            hipLaunchKernelGGL((vq_filter_router_kernel<ALIGNED, CONV, true>), grid, dim3(kVqfThreads), lds, s, p.a, p.r, p.nrouter, p.router_behind);
            return launch_check("vq_filter_router_kernel(split)"); });
    }
    rc = ensure_dynamic_lds((const void *)vq_filter_router_kernel<ALIGNED, CONV>, lds);
    if (rc) return rc;
    return launch_or_record(CONV ? KID_NONE : ALIGNED ? KID_VQF_ROUTER_AL : KID_VQF_ROUTER_UN, grid, dim3(kVqfThreads), lds, p, s, [=] {
        hipLaunchKernelGGL((vq_filter_router_kernel<ALIGNED, CONV>), grid, dim3(kVqfThreads), lds, s, p.a, p.r, p.nrouter, p.router_behind);
        return launch_check("vq_filter_router_kernel"); });
}

template <bool ALIGNED>
static int vqfr_grouped_launch(const GroupRec *const *recs, int n, hipStream_t s)
{
    Grouped<VqfrArgs> g;
    size_t lds;
    int rc = fill_grouped(recs, n, &g, &lds);
    if (rc) return rc;
    bool split = false;
    for (int i = 0; i < n; ++i) split = split || g.a[i].r.rq.nq != 0;
    if (split) {
        rc = ensure_dynamic_lds((const void *)vq_filter_router_grouped_kernel<ALIGNED, true>, lds);
        if (rc) return rc;
        hipLaunchKernelGGL((vq_filter_router_grouped_kernel<ALIGNED, true>), dim3(g.start[kMaxGroups]), dim3(kVqfThreads), lds, s, g);
        return launch_check("vq_filter_router_grouped_kernel(split)");
    }
    rc = ensure_dynamic_lds((const void *)vq_filter_router_grouped_kernel<ALIGNED>, lds);
    if (rc) return rc;
    hipLaunchKernelGGL((vq_filter_router_grouped_kernel<ALIGNED>), dim3(g.start[kMaxGroups]), dim3(kVqfThreads), lds, s, g);
    return launch_check("vq_filter_router_grouped_kernel");
}
static GroupedRegistrar reg_vqfr_al(KID_VQF_ROUTER_AL, vqfr_grouped_launch<true>);
static GroupedRegistrar reg_vqfr_un(KID_VQF_ROUTER_UN, vqfr_grouped_launch<false>);

static int vq_dispatch(const float *z, int64_t hw, int64_t N, const float *codebook, int K, int64_t *indices, float *z_q,
                       VqWs ws, float beta, int legacy, float *loss, hipStream_t s, const RouterArgs *router,
                       int64_t router_blocks, size_t router_lds, const cgic_conv1x1 *qc, const void *prepared)
{
    const int force_zt = dev_knob("CGIC_VQ_ZT");          // dev: tile count of the exact loop
    if (!dev_knob("CGIC_VQ_EXACT") && K % 64 == 0 && K <= kVqfMaxK) {
#define CGIC_VQF_LAUNCH(AL, CV) launch_filter<AL, CV>(z, hw, N, codebook, K, indices, z_q, ws, beta, legacy, loss, s, router, router_blocks, router_lds, qc, prepared)
        if (hw % kVqfGroup == 0) return qc ? CGIC_VQF_LAUNCH(true, true) : CGIC_VQF_LAUNCH(true, false);
        return qc ? CGIC_VQF_LAUNCH(false, true) : CGIC_VQF_LAUNCH(false, false);
#undef CGIC_VQF_LAUNCH
    }
    CGIC_REQUIRE(!qc, CGIC_ERR_UNSUPPORTED, "vq: the fused quant_conv needs K %% 64 == 0 and K <= %d (K=%d): apply the 1x1 convolution separately", kVqfMaxK, K);
    // exact loop; per-wave tile: measured on MI355X (tools/probes/probe_vq.hip) ZT=4 at 4 waves/SIMD is the fastest
    // for large N; smaller N shrinks the tile so that all 256 CUs get work
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

extern "C" int cgic_vq_filter_probe_f32(const float *z, int64_t B, int64_t hw, const float *codebook, int K, int64_t *indices,
                                        float *scores, float *aux, cgic_stream_t stream)
{
    CGIC_NOT_IN_GROUP("cgic_vq_filter_probe_f32");
    int rc = vq_check(z, B, hw, codebook, K, 4, nullptr, nullptr, false);
    if (rc) return rc;
    CGIC_REQUIRE(K % 64 == 0 && K <= kVqfMaxK, CGIC_ERR_UNSUPPORTED, "vq_filter_probe: K=%d has no filter path", K);
    CGIC_REQUIRE(indices && scores && aux, CGIC_ERR_INVALID, "vq_filter_probe: NULL output");
    const int64_t N = B * hw;
    if (N == 0) return CGIC_OK;
    int cus = 0;
    rc = device_cu_count(&cus);
    if (rc) return rc;
    const int64_t ngroups = (N + kVqfGroup - 1) / kVqfGroup;
    const int64_t nblk = ngroups < cus ? ngroups : cus, per = (ngroups + nblk - 1) / nblk;
    VqArgs a;
    a.z = z; a.hw = hw; a.N = N; a.cb = codebook; a.K = K; a.idx_out = indices; a.zq_out = nullptr;
    a.sq_partial = nullptr; a.ticket = nullptr; a.beta = 0.f; a.legacy = 1; a.loss = nullptr;
    a.nblk = (unsigned int)nblk; a.n_early = (unsigned int)nblk; a.g_early = a.g_late = (unsigned int)per;
    a.tail_mode = 0; a.conv_w = a.conv_b = nullptr; a.conv_bias_first = 0; a.prep = nullptr; a.stats = nullptr;
    a.probe_scores = scores; a.probe_aux = aux;
    const size_t lds = vqf_lds_bytes(K);
    hipStream_t s = (hipStream_t)stream;
    if (hw % kVqfGroup == 0) {
        rc = ensure_dynamic_lds((const void *)vq_filter_probe_kernel<true>, lds);
        if (rc) return rc;
        hipLaunchKernelGGL((vq_filter_probe_kernel<true>), dim3(a.nblk), dim3(kVqfThreads), lds, s, a);
    } else {
        rc = ensure_dynamic_lds((const void *)vq_filter_probe_kernel<false>, lds);
        if (rc) return rc;
        hipLaunchKernelGGL((vq_filter_probe_kernel<false>), dim3(a.nblk), dim3(kVqfThreads), lds, s, a);
    }
    return launch_check("vq_filter_probe_kernel");
}

extern "C" int cgic_vq_stats(unsigned int *device_counters)
{
    CGIC_REQUIRE(((uintptr_t)device_counters & 3u) == 0, CGIC_ERR_INVALID, "vq_stats: the counters must be 4-byte aligned");
    g_vq_stats.store(device_counters, std::memory_order_relaxed);
    return CGIC_OK;
}

extern "C" size_t cgic_vq_prepared_bytes(int K)
{
    // the filter path's LDS image + 16 bytes (the two maxima, K, magic) + K keys of a permuted image + K uint16 (the permutation as
    // handed to the prepare kernel); 0: this K has no filter path (nothing to prepare)
    return (K > 0 && K % 64 == 0 && K <= kVqfMaxK) ? vqf_lds_bytes(K) + 16 + 6 * (size_t)K : 0;
}

namespace cgic {
// Which prepared images are permuted (the launch picks the PERM kernels for them).  A wrong answer is harmless: a kernel that finds
// another magic than the one it was built for derives its own image from the fp32 rows.
static std::mutex g_perm_mu;
static std::map<std::pair<int, const void *>, bool> g_perm_images;        // (device, image): addresses of different devices may coincide
static std::pair<int, const void *> perm_key(const void *prepared)
{
    int dev = 0;
    (void)hipGetDevice(&dev);
    return std::make_pair(dev, prepared);
}
static bool prepared_is_perm(const void *prepared)
{
    if (!prepared) return false;
    std::lock_guard<std::mutex> lock(g_perm_mu);
    auto it = g_perm_images.find(perm_key(prepared));
    return it != g_perm_images.end() && it->second;
}

// Near-duplicate rows -> one tile.  A trained codebook holds clusters of rows closer to each other than the candidate filter's
// margin (quantize.py:22-26: rows that started within +-1/K of each other and were never pulled apart, dead codes): spread over the
// tiles, every vector near such a cluster finds a runner-up TILE inside its margin and falls to the all-K exact scan (the whole
// group to the exact loop: 2.2x the step).  Packed into one 32-code tile they are resolved by the exact step that looks at the best
// tile anyway.  Greedy on the host (K <= 1024): first-fit leaders under the max-norm, clusters largest first into the tiles (never
// split unless larger than a tile), singles fill up.  Returns false when there is nothing to pack (every cluster is a single row).
static bool cluster_permutation(const float *cb, int K, std::vector<unsigned short> *perm)
{
    float emax = 0.f;
    for (int i = 0; i < 4 * K; ++i) { const float v = fabsf(cb[i]); if (!(v <= emax)) emax = v; }      // (a NaN ends up in emax)
    if (!(emax > 0.f) || !(emax < INFINITY)) return false;
    const float tau = 3e-4f * emax;
    std::vector<int> leader(K), leaders;
    for (int i = 0; i < K; ++i) {
        int L = i;
        for (int j : leaders) {
            const float *a = cb + 4 * i, *b = cb + 4 * j;
            if (fabsf(a[0] - b[0]) <= tau && fabsf(a[1] - b[1]) <= tau && fabsf(a[2] - b[2]) <= tau && fabsf(a[3] - b[3]) <= tau) { L = j; break; }
        }
        leader[i] = L;
        if (L == i) leaders.push_back(i);
    }
    if ((int)leaders.size() == K) return false;
    std::map<int, std::vector<int>> members;
    for (int i = 0; i < K; ++i) members[leader[i]].push_back(i);
    std::vector<std::vector<int>> chunks;         // clusters cut into pieces of at most one tile
    for (auto &kv : members)
        for (size_t at = 0; at < kv.second.size(); at += 32)
            chunks.emplace_back(kv.second.begin() + at, kv.second.begin() + (at + 32 < kv.second.size() ? at + 32 : kv.second.size()));
    std::stable_sort(chunks.begin(), chunks.end(), [](const std::vector<int> &a, const std::vector<int> &b) { return a.size() > b.size(); });
    const int ntile = K / 32;
    std::vector<std::vector<int>> tiles(ntile);
    // Round 6: tiles are filled to EXACTLY 32 rows wherever the cluster sizes allow.  (First fit, largest first, left gaps of a few
    // rows in many tiles and then cut the last clusters over them -- 7 of the bench's 64 clusters, 8.7 % of the vectors next to a cut
    // cluster, 4.5 % on the all-K scan.)  Per tile: the largest piece not yet placed, then a depth-first search for a subset of the
    // remaining multi-row pieces whose sizes add up to the room that is left (single rows fill any gap, so only the multi-row
    // pieces are searched; equal sizes are tried once per level; the search is cut off after 200 000 nodes and the fullest subset
    // found is taken).  What cannot be placed whole at the end is cut over the room that is left, as before.
    {
        std::vector<int> multi, singles;                  // chunk ids, multi-row ones largest first (chunks is sorted by size)
        for (int i = 0; i < (int)chunks.size(); ++i) (chunks[i].size() >= 2 ? multi : singles).push_back(i);
        std::vector<char> used(chunks.size(), 0);
        size_t nsing = singles.size();
        std::vector<int> left;
        for (int t = 0; t < ntile; ++t) {
            int first = -1;
            for (int i : multi) if (!used[i]) { first = i; break; }
            if (first < 0) break;
            used[first] = 1;
            const int target = 32 - (int)chunks[first].size();
            std::vector<int> rest;
            for (int i : multi) if (!used[i]) rest.push_back(i);
            std::vector<int> chosen, best, found;
            int best_sum = -1;
            long nodes = 0;
            bool done = false;
            std::function<void(size_t, int)> dfs = [&](size_t startk, int remaining) {
                if (done) return;
                const int fill = remaining < (int)nsing ? remaining : (int)nsing;
                const int got = (target - remaining) + fill;
                if (got > best_sum) { best_sum = got; best = chosen; }
                if (remaining - fill == 0) { found = chosen; done = true; return; }
                if (++nodes > 200000) { done = true; return; }
                int prev = -1;
                for (size_t k = startk; k < rest.size(); ++k) {
                    const int sz = (int)chunks[rest[k]].size();
                    if (sz > remaining || sz == prev) continue;
                    prev = sz;
                    chosen.push_back(rest[k]);
                    dfs(k + 1, remaining - sz);
                    chosen.pop_back();
                    if (done) return;
                }
            };
            dfs(0, target);
            const std::vector<int> &sub = (best_sum >= 0 && (int)found.size() == 0 && best_sum < target) ? best : (found.empty() ? best : found);
            tiles[t].insert(tiles[t].end(), chunks[first].begin(), chunks[first].end());
            for (int i : sub) { used[i] = 1; tiles[t].insert(tiles[t].end(), chunks[i].begin(), chunks[i].end()); }
            while (tiles[t].size() < 32 && nsing > 0) { const int i = singles[singles.size() - nsing]; --nsing; used[i] = 1; tiles[t].push_back(chunks[i][0]); }
        }
        for (int i = 0; i < (int)chunks.size(); ++i) if (!used[i]) left.push_back(i);
        // the rest: whole where it fits, else cut over the tiles with the most room (its rows then meet the all-K scan: only that cluster)
        for (int ci : left) {
            const auto &c = chunks[ci];
            size_t at = 0;
            while (at < c.size()) {
                int bestt = -1;
                for (int t = 0; t < ntile; ++t)
                    if (tiles[t].size() + (c.size() - at) <= 32) { bestt = t; break; }
                size_t take = c.size() - at;
                if (bestt < 0) {
                    size_t room = 0;
                    for (int t = 0; t < ntile; ++t)
                        if (32 - tiles[t].size() > room) { room = 32 - tiles[t].size(); bestt = t; }
                    if (bestt < 0 || room == 0) return false;          // (cannot happen: the pieces sum up to K = 32 ntile)
                    take = room;
                }
                tiles[bestt].insert(tiles[bestt].end(), c.begin() + at, c.begin() + at + take);
                at += take;
            }
        }
    }
    perm->resize(K);
    for (int t = 0; t < ntile; ++t) {
        if ((int)tiles[t].size() != 32) return false;
        for (int k = 0; k < 32; ++k) (*perm)[32 * t + k] = (unsigned short)tiles[t][k];
    }
    return true;
}
}  // namespace cgic

extern "C" int cgic_vq_cluster_permutation_host(const float *codebook_host, int K, uint16_t *perm_out)
{
    CGIC_REQUIRE(codebook_host && perm_out, CGIC_ERR_INVALID, "vq_cluster_permutation_host: NULL argument");
    CGIC_REQUIRE(cgic_vq_prepared_bytes(K) != 0, CGIC_ERR_UNSUPPORTED, "vq_cluster_permutation_host: K=%d has no filter path", K);
    std::vector<unsigned short> perm;
    if (!cluster_permutation(codebook_host, K, &perm)) return 0;
    for (int i = 0; i < K; ++i) perm_out[i] = perm[i];
    return 1;
}

extern "C" int cgic_vq_prepare_f32(const float *codebook, int K, int e_dim, void *prepared, cgic_stream_t stream)
{
    CGIC_NOT_IN_GROUP("cgic_vq_prepare_f32");
    CGIC_REQUIRE(codebook && prepared, CGIC_ERR_INVALID, "vq_prepare: NULL argument");
    CGIC_REQUIRE(e_dim == 4, CGIC_ERR_UNSUPPORTED, "vq_prepare: e_dim=%d; this build implements embed_dim == 4", e_dim);
    CGIC_REQUIRE(cgic_vq_prepared_bytes(K) != 0, CGIC_ERR_UNSUPPORTED, "vq_prepare: K=%d has no filter path (need K %% 64 == 0, K <= %d)", K, kVqfMaxK);
    CGIC_REQUIRE(((uintptr_t)prepared & 15) == 0, CGIC_ERR_INVALID, "vq_prepare: the image must be 16-byte aligned");
    const size_t lds = vqf_lds_bytes(K);
    int rc = ensure_dynamic_lds((const void *)vq_prepare_kernel, lds);
    if (rc) return rc;
    // near-duplicate rows packed into tiles (cluster_permutation): needs the rows on the host -- one 16 KB copy and a wait, once per
    // codebook; not while the stream is being captured (the image is then the plain one: same results, no packing)
    hipStream_t s = (hipStream_t)stream;
    const unsigned short *perm_dev = nullptr;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    CGIC_HIP_TRY(hipStreamIsCapturing(s, &cap));
    bool is_perm = false;
    if (cap == hipStreamCaptureStatusNone) {
        std::vector<float> host((size_t)4 * K);
        CGIC_HIP_TRY(hipMemcpyAsync(host.data(), codebook, sizeof(float) * 4 * K, hipMemcpyDeviceToHost, s));
        CGIC_HIP_TRY(hipStreamSynchronize(s));
        std::vector<unsigned short> perm;
        if (cluster_permutation(host.data(), K, &perm)) {
            unsigned short *dst = reinterpret_cast<unsigned short *>(reinterpret_cast<unsigned char *>(prepared) + lds + 16 + 4 * (size_t)K);
            CGIC_HIP_TRY(hipMemcpyAsync(dst, perm.data(), sizeof(unsigned short) * K, hipMemcpyHostToDevice, s));
            CGIC_HIP_TRY(hipStreamSynchronize(s));          // (perm is about to go out of scope)
            perm_dev = dst;
            is_perm = true;
        }
    }
    {   // (a plain image leaves no entry: the map only ever holds the permuted images that are alive or were overwritten in place)
        std::lock_guard<std::mutex> lock(g_perm_mu);
        if (is_perm) g_perm_images[perm_key(prepared)] = true; else g_perm_images.erase(perm_key(prepared));
    }
    hipLaunchKernelGGL(vq_prepare_kernel, dim3(1), dim3(kVqfThreads), lds, s, codebook, K, (uint4 *)prepared, perm_dev);
    return launch_check("vq_prepare_kernel");
}

extern "C" int cgic_vq_forward_f32(const float *z, int64_t B, int64_t hw, const float *codebook, int K,
                                   int e_dim, float beta, int legacy, int64_t *indices, float *z_q,
                                   float *loss, int64_t *hist, void *workspace, const cgic_conv1x1 *quant_conv,
                                   const void *prepared, cgic_stream_t stream)
{
    CGIC_NOT_IN_GROUP("cgic_vq_forward_f32");
    int rc = vq_check(z, B, hw, codebook, K, e_dim, loss, workspace, hist && !indices);
    if (rc) return rc;
    rc = conv_check(quant_conv);
    if (rc) return rc;
    rc = prepared_check(prepared, K);
    if (rc) return rc;
    const int64_t N = B * hw;
    if (N == 0) return CGIC_OK;
    hipStream_t s = (hipStream_t)stream;
    VqWs ws;
    rc = vq_ws(loss ? workspace : nullptr, s, &ws);
    if (rc) return rc;
    rc = vq_dispatch(z, hw, N, codebook, K, indices, z_q, ws, beta, legacy, loss, s, nullptr, 0, 0, quant_conv, prepared);
    if (rc == CGIC_OK && hist) rc = launch_hist(indices, N, K, hist, s);
    return rc;
}

extern "C" int cgic_vq_forward_route_f32(const float *z, int64_t B, int64_t hw, const float *codebook, int K, int e_dim,
                                         float beta, int legacy, int64_t *indices, float *z_q, float *loss,
                                         void *workspace, const float *e16, const float *e8, int64_t h16, int64_t w16,
                                         double coarse_ratio, double medium_ratio, int per_image, int32_t *mask_c,
                                         int32_t *mask_m, int32_t *mask_f, float *gate, int *mode_out,
                                         const cgic_conv1x1 *quant_conv, const void *prepared, const cgic_pixels *refine,
                                         cgic_stream_t stream)
{
    int rc = vq_check(z, B, hw, codebook, K, e_dim, loss, workspace, false);
    if (rc) return rc;
    rc = conv_check(quant_conv);
    if (rc) return rc;
    rc = prepared_check(prepared, K);
    if (rc) return rc;
    if (mode_out) *mode_out = cgic_router_mode(coarse_ratio, medium_ratio);
    const int64_t N = B * hw;
    if (N == 0) return CGIC_OK;
    if (refine && refine->x && cgic_router_mode(coarse_ratio, medium_ratio) <= 3 && !router_refine_in_lds(B, h16, w16, per_image)) {
        // a routing segment beyond the LDS (the reference's flattened-batch routing of encode(), an untiled large image): the VQ
        // launch by itself, then the router's chain of launches over patched copies of the maps (cgic_router.hip: router_big)
        CGIC_REQUIRE(!group_recording(), CGIC_ERR_UNSUPPORTED,
                     "vq_forward_route: the refinement of a segment that does not fit the LDS is a chain of launches: not inside a launch group");
        hipStream_t s0 = (hipStream_t)stream;
        VqWs ws0;
        rc = vq_ws(loss ? workspace : nullptr, s0, &ws0);
        if (rc) return rc;
        rc = vq_dispatch(z, hw, N, codebook, K, indices, z_q, ws0, beta, legacy, loss, s0, nullptr, 0, 0, quant_conv, prepared);
        if (rc) return rc;
        return router_big(e16, e8, B, h16, w16, coarse_ratio, medium_ratio, per_image, mask_c, mask_m, mask_f, gate, refine, s0);
    }
    RouterArgs r;
    int64_t nseg;
    size_t rlds;
    // (78 KB: a router workgroup of the fused launch shares its CU with a VQ workgroup -- two allocations per 160 KB)
    // (with a scratch: the launch's refinement queues -- images with long threshold bands publish them, the routers that are done help)
    rc = router_prepare(e16, e8, B, h16, w16, coarse_ratio, medium_ratio, per_image, mask_c, mask_m, mask_f, gate, &r, &nseg, &rlds,
                        kRouterFusedLds, refine, (hipStream_t)stream, per_image != 0 && refine && refine->scratch);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    VqWs ws;
    rc = vq_ws(loss ? workspace : nullptr, s, &ws);
    if (rc) return rc;
    return vq_dispatch(z, hw, N, codebook, K, indices, z_q, ws, beta, legacy, loss, s, &r, nseg, rlds, quant_conv, prepared);
}

extern "C" int cgic_vq_forward_valu_f32(const float *z, int64_t B, int64_t hw, const float *codebook, int K,
                                        int e_dim, float beta, int legacy, int64_t *indices, float *z_q,
                                        float *loss, int64_t *hist, void *workspace, const cgic_conv1x1 *quant_conv,
                                        cgic_stream_t stream)
{
    CGIC_NOT_IN_GROUP("cgic_vq_forward_valu_f32");
    int rc = vq_check(z, B, hw, codebook, K, e_dim, loss, workspace, hist && !indices);
    if (rc) return rc;
    rc = conv_check(quant_conv);
    if (rc) return rc;
    const int64_t N = B * hw;
    if (N == 0) return CGIC_OK;
    hipStream_t s = (hipStream_t)stream;
    VqWs ws;
    rc = vq_ws(loss ? workspace : nullptr, s, &ws);
    if (rc) return rc;
    int nblk = (int)((N + kVqThreads - 1) / kVqThreads);
    size_t lds = sizeof(float) * (size_t)K * 5;
    rc = ensure_dynamic_lds((const void *)vq_valu_kernel, lds);
    if (rc) return rc;
    hipLaunchKernelGGL(vq_valu_kernel, dim3(nblk), dim3(kVqThreads), lds, s, z, hw, N, codebook, K, indices, z_q,
                       loss ? ws.partial : nullptr, ws.ticket, beta, legacy, loss,
                       quant_conv ? quant_conv->weight : (const float *)nullptr, quant_conv ? quant_conv->bias : (const float *)nullptr,
                       quant_conv ? quant_conv->bias_first : 0);
    rc = launch_check("vq_valu_kernel");
    if (rc == CGIC_OK && hist) rc = launch_hist(indices, N, K, hist, s);
    return rc;
}

// y[n] = W x[n] (+ b) for n rows of 4 floats: post_quant_conv applied to the codebook (model.py:52,115) -- gathering rows
// of the transformed codebook is bit-identical to convolving the gathered latent, at 1024 rows instead of B*h*w
__global__ __launch_bounds__(256) void conv_rows_kernel(const float4 *__restrict__ x, int64_t n, const float *__restrict__ cw,
                                                        const float *__restrict__ cb, int bias_first, float4 *__restrict__ y)
{
    Conv1x1 conv;
    conv.load(cw, cb, bias_first);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 r = x[i];
        float v[4] = {r.x, r.y, r.z, r.w};
        conv.apply(v);
        y[i] = make_float4(v[0], v[1], v[2], v[3]);
    }
}

extern "C" int cgic_conv1x1_rows_f32(const float *rows, int64_t n, const cgic_conv1x1 *conv, float *out, cgic_stream_t stream)
{
    CGIC_NOT_IN_GROUP("cgic_conv1x1_rows_f32");
    CGIC_REQUIRE(rows && out && conv && conv->weight && n >= 0, CGIC_ERR_INVALID, "conv1x1_rows: NULL argument");
    if (n == 0) return CGIC_OK;
    int64_t nblk = (n + 255) / 256;
    if (nblk > 4096) nblk = 4096;
    hipLaunchKernelGGL(conv_rows_kernel, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, (const float4 *)rows, n, conv->weight,
                       conv->bias, conv->bias_first, (float4 *)out);
    return launch_check("conv_rows_kernel");
}

extern "C" int cgic_index_histogram(const int64_t *indices, int64_t n, int K, int64_t *hist, cgic_stream_t stream)
{
    CGIC_REQUIRE(indices && hist && K > 0 && K <= 16384 && n >= 0, CGIC_ERR_INVALID, "index_histogram: bad args");
    if (n == 0) return CGIC_OK;
    return launch_hist(indices, n, K, hist, (hipStream_t)stream);
}
